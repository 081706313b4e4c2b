// Frozen wav2vec 2.0 speech embedder (reference call site utils/wav2vec_util.py:14-32): the stages that
// are not GEMMs.  Every Linear / strided Conv1d / grouped positional conv of the model runs on conv_gemm (kernel size 1
// on overlapping-row views, see speech_decoding_amd/wav2vec2.py); this file holds
//   * w2v_conv0_kernel      first feature-encoder layer: Conv1d(1 -> C, k, stride) + LayerNorm(C) + GELU, waveform in
//   * layernorm_rows_kernel LayerNorm over the channels of each row (+ optional GELU)
//   * group_split / group_merge_add  row layout <-> per-group row layout around the grouped positional conv
//   * attention_kernel      softmax(Q K^T / sqrt(d)) V per head on the matrix cores, flash-style over 64-key blocks
//   * mean4_kernel          mean of the last four hidden states (wav2vec_util.py:18-20), fp32 out
// All activations are row-layout buffers of ONE chunk (B = 1): row SDA_ROW_PAD + t holds frame t.
#include "sd_common.h"

namespace sda {

// ------------------------------------------------------------------------------------------------
// Conv1d(1 -> C, K, stride) + bias + LayerNorm(C) + GELU: one wave per output frame, lane = channels lane + 64 i
// (HF Wav2Vec2LayerNormConvLayer, layer 0).  Weights [C][K] sit in LDS.
// ------------------------------------------------------------------------------------------------
template <typename E, int NI>
__global__ __launch_bounds__(256) void w2v_conv0_kernel(const float* __restrict__ wave, long n_samples,
                                                        const float* __restrict__ w, const float* __restrict__ bias,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        E* __restrict__ y, int T, int C, int Cp, int K, int stride, float eps) {
  extern __shared__ float wl[];                      // [C][K]
  for (int i = threadIdx.x; i < C * K; i += 256) wl[i] = w[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  float g[NI], be[NI], b0[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int c = lane + 64 * i;
    const bool ok = c < C;
    g[i] = ok ? gamma[c] : 0.f; be[i] = ok ? beta[c] : 0.f; b0[i] = (ok && bias) ? bias[c] : 0.f;
  }
  for (int t = blockIdx.x * 4 + wid; t < T; t += gridDim.x * 4) {
    float v[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) v[i] = b0[i];
    const float* xs = wave + (long)t * stride;
    for (int j = 0; j < K; ++j) {
      const float xv = xs[j];
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int c = lane + 64 * i;
        if (c < C) v[i] = fmaf(wl[c * K + j], xv, v[i]);
      }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) s += (lane + 64 * i < C) ? v[i] : 0.f;
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) { const float d = (lane + 64 * i < C) ? v[i] - mean : 0.f; q = fmaf(d, d, q); }
    const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
    E* yr = y + (size_t)(PAD + t) * Cp;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int c = lane + 64 * i;
      if (c < C) Elem<E>::st(yr + c, gelu_f<E>(fmaf((v[i] - mean) * rstd, g[i], be[i])));
    }
  }
}

// ------------------------------------------------------------------------------------------------
// LayerNorm over the C valid channels of each row (two-pass: mean, then centred variance), affine, optional GELU.
// One wave per row; a lane keeps its 16-byte chunks (lane + 64 i) in registers.
// ------------------------------------------------------------------------------------------------
template <typename E, int NI>
__global__ __launch_bounds__(256) void layernorm_rows_kernel(const E* __restrict__ x, E* __restrict__ y,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             int T, int C, int Cp, float eps, int gelu) {
  constexpr int CH = Vec16<E>::N;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int nch = Cp / CH;
  for (int t = blockIdx.x * 4 + wid; t < T; t += gridDim.x * 4) {
    const size_t row = (size_t)(PAD + t) * Cp;
    float v[NI][CH];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int q = lane + 64 * i;
      if (q < nch) {
        Vec16<E>::load(x + row + q * CH, v[i]);
#pragma unroll
        for (int j = 0; j < CH; ++j) s += (q * CH + j < C) ? v[i][j] : 0.f;
      }
    }
    const float mean = wave_sum(s) / (float)C;
    float qq = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int q = lane + 64 * i;
      if (q < nch) {
#pragma unroll
        for (int j = 0; j < CH; ++j) { const float d = (q * CH + j < C) ? v[i][j] - mean : 0.f; qq = fmaf(d, d, qq); }
      }
    }
    const float rstd = rsqrtf(wave_sum(qq) / (float)C + eps);
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int q = lane + 64 * i;
      if (q < nch) {
        float o[CH];
#pragma unroll
        for (int j = 0; j < CH; ++j) {
          const int c = q * CH + j;
          float r = 0.f;
          if (c < C) {
            r = fmaf((v[i][j] - mean) * rstd, gamma[c], beta[c]);
            if (gelu) r = gelu_f<E>(r);
          }
          o[j] = r;
        }
        Vec16<E>::store(y + row + q * CH, o);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Grouped positional conv (HF Wav2Vec2PositionalConvEmbedding: Conv1d(H, H, K, padding = K/2, groups = G)) as G GEMMs:
// group g's channels go to their own buffer [R rows][gwp] with frame t at row lead + t (zero rows around it), so that
// output frame t contracts over the CONTIGUOUS K * gwp elements starting at row lead + t - K/2.
// ------------------------------------------------------------------------------------------------
template <typename E>
__global__ __launch_bounds__(256) void group_split_kernel(const E* __restrict__ h, E* __restrict__ xg, int T, int Hp, int gw,
                                                          int gwp, int G, long group_rows, int lead) {
  constexpr int CH = Vec16<E>::N;
  const int per_row = G * (gw / CH);
  const long total = (long)T * per_row;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int t = (int)(i / per_row), r = (int)(i - (long)t * per_row);
    const int g = r / (gw / CH), q = r - g * (gw / CH);
    const uint4 u = *reinterpret_cast<const uint4*>(h + (size_t)(PAD + t) * Hp + g * gw + q * CH);
    *reinterpret_cast<uint4*>(xg + ((size_t)g * group_rows + lead + t) * gwp + q * CH) = u;
  }
}

// h_out[t][g * gw + c] = h[t][g * gw + c] + f(yg[g][PAD + t][c] + bias[g * gw + c]),  f = GELU when `gelu`
// (yg: G row-layout buffers of gwp channels, `yg_rows` rows each; bias may be NULL)
template <typename E>
__global__ __launch_bounds__(256) void group_merge_add_kernel(const E* __restrict__ h, const E* __restrict__ yg,
                                                              E* __restrict__ out, int T, int Hp, int gw, int gwp, int G,
                                                              long yg_rows, const float* __restrict__ bias, int gelu) {
  constexpr int CH = Vec16<E>::N;
  const int per_row = G * (gw / CH);
  const long total = (long)T * per_row;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int t = (int)(i / per_row), r = (int)(i - (long)t * per_row);
    const int g = r / (gw / CH), q = r - g * (gw / CH);
    float a[CH], b[CH];
    const size_t off = (size_t)(PAD + t) * Hp + g * gw + q * CH;
    Vec16<E>::load(h + off, a);
    Vec16<E>::load(yg + ((size_t)g * yg_rows + PAD + t) * gwp + q * CH, b);
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      float v = b[j] + (bias ? bias[g * gw + q * CH + j] : 0.f);
      if (gelu) v = gelu_f<E>(v);
      a[j] += v;
    }
    Vec16<E>::store(out + off, a);
  }
}

// ------------------------------------------------------------------------------------------------
// Self-attention of one chunk (HF eager_attention_forward: softmax(q k^T * scale) v, no mask, eval).
//   q, k : row layout, head h in columns [64 h, 64 h + 64) of rows with pitch `qk_pitch`
//   vt   : V TRANSPOSED, plain matrix [heads * 64][vt_pitch] (row 64 h + d = dimension d of head h over the keys);
//          columns >= T may hold anything finite (their probabilities are exactly zero)
//   out  : row layout [rows][out_pitch], head h in columns [64 h, 64 h + 64)
// One workgroup = one head x 64 queries (a wave owns 16 query rows); keys in blocks of 64, online softmax.  Operands go
// to the MFMA straight from global memory / L2 in `mma16`'s 64-byte K-step form: A = 16 rows x 16 bytes per lane group.
// S and O live in the MFMA accumulator layout (lane (lr, lq): column lr, rows 4 lq + r), so the per-row running max and
// sum are per-lane values reduced over the 16 lanes of a lane group; P goes through a per-wave LDS tile to become the A
// operand of P V.
// ------------------------------------------------------------------------------------------------
template <typename E>
__global__ __launch_bounds__(256) void attention_kernel(const E* __restrict__ q, const E* __restrict__ k,
                                                        const E* __restrict__ vt, E* __restrict__ out, int T, long qk_pitch,
                                                        long vt_pitch, long out_pitch, float scale) {
  constexpr int HD = 64;                                 // head dimension
  constexpr int ES = (int)sizeof(E);
  constexpr int NKQ = HD * ES / 64;                      // 64-byte K-steps over the head dimension (2 / 4)
  constexpr int NKP = 64 * ES / 64;                      // K-steps over a 64-key block
  constexpr int PER16 = Elem<E>::PER16;
  __shared__ __attribute__((aligned(16))) unsigned char plds[4][16 * 64 * ES];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int lr = lane & 15, lq = lane >> 4;
  const int head = blockIdx.y;
  const int q0 = blockIdx.x * 64 + wid * 16;             // this wave's first query
  // A fragments of Q (rows past T are clamped: their outputs are never stored)
  uint4 qf[NKQ];
  {
    const int qr = min(q0 + lr, T - 1);
    const E* qp = q + (size_t)(PAD + qr) * qk_pitch + head * HD;
#pragma unroll
    for (int s = 0; s < NKQ; ++s) qf[s] = *reinterpret_cast<const uint4*>(qp + s * (64 / ES) + lq * PER16);
  }
  f32x4 o[4];
#pragma unroll
  for (int n = 0; n < 4; ++n) o[n] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m[4], l[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) { m[r] = -INFINITY; l[r] = 0.f; }
  unsigned char* pw = plds[wid];
  for (int k0 = 0; k0 < T; k0 += 64) {
    // S = Q K^T for 64 keys: 4 key fragments
    f32x4 s[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      s[f] = f32x4{0.f, 0.f, 0.f, 0.f};
      const int kr = min(k0 + f * 16 + lr, T - 1);
      const E* kp = k + (size_t)(PAD + kr) * qk_pitch + head * HD;
#pragma unroll
      for (int ks = 0; ks < NKQ; ++ks) {
        const uint4 kf = *reinterpret_cast<const uint4*>(kp + ks * (64 / ES) + lq * PER16);
        s[f] = mma16<E>(qf[ks], kf, s[f]);
      }
    }
    // online softmax: this lane holds rows 4 lq + r, key column k0 + 16 f + lr
    float mx[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float v = -INFINITY;
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        s[f][r] = (k0 + f * 16 + lr < T) ? s[f][r] * scale : -INFINITY;
        v = fmaxf(v, s[f][r]);
      }
      v = fmaxf(v, __shfl_xor(v, 1)); v = fmaxf(v, __shfl_xor(v, 2)); v = fmaxf(v, __shfl_xor(v, 4)); v = fmaxf(v, __shfl_xor(v, 8));
      mx[r] = fmaxf(m[r], v);               // finite: every block has at least one valid key
    }
    float alpha[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      alpha[r] = __expf(m[r] - mx[r]);      // exp(-inf) = 0 on the first block
      float sum = 0.f;
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        const float p = Vec16<E>::round(__expf(s[f][r] - mx[r]));     // the probabilities the matrix cores will see
        s[f][r] = p;
        sum += p;
      }
      sum += __shfl_xor(sum, 1); sum += __shfl_xor(sum, 2); sum += __shfl_xor(sum, 4); sum += __shfl_xor(sum, 8);
      l[r] = l[r] * alpha[r] + sum;
      m[r] = mx[r];
    }
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) o[n][r] *= alpha[r];
    // P: accumulator layout -> row-major [16 rows][64 keys] in this wave's LDS tile -> A fragments
    E* pe = reinterpret_cast<E*>(pw);
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
      for (int r = 0; r < 4; ++r) Elem<E>::st(pe + (4 * lq + r) * 64 + f * 16 + lr, s[f][r]);
    __syncthreads();                        // (per-wave tile; every wave runs the same number of key blocks)
    uint4 pf[NKP];
#pragma unroll
    for (int ks = 0; ks < NKP; ++ks) pf[ks] = *reinterpret_cast<const uint4*>(pw + lr * 64 * ES + ks * 64 + lq * 16);
    // O += P V: B fragment = V^T rows (dimension 16 n + lr) over this block's keys
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      const E* vp = vt + (size_t)(head * HD + n * 16 + lr) * vt_pitch + k0;
#pragma unroll
      for (int ks = 0; ks < NKP; ++ks) {
        const uint4 vf = *reinterpret_cast<const uint4*>(vp + ks * (64 / ES) + lq * PER16);
        o[n] = mma16<E>(pf[ks], vf, o[n]);
      }
    }
    __syncthreads();                        // the tile is rewritten by the next block
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int qr = q0 + 4 * lq + r;
    if (qr < T) {
      const float inv = 1.f / l[r];
      E* op = out + (size_t)(PAD + qr) * out_pitch + head * HD;
#pragma unroll
      for (int n = 0; n < 4; ++n) Elem<E>::st(op + n * 16 + lr, o[n][r] * inv);
    }
  }
}

// Epilogue of a split-K GEMM (sda_conv_gemm with ksplit > 1 leaves raw fp32 slabs): y[PAD + t][c] =
// f(sum_s partial[s][t][c] + bias[c]) + res[PAD + t][c], f = GELU when `gelu`; slabs summed in order (deterministic).
// The chunks of this path are a few hundred frames: without the K split a 1024-wide Linear is 24 workgroups on 256 CUs.
template <typename E>
__global__ __launch_bounds__(256) void splitk_epilogue_kernel(const float* __restrict__ partial, int ksplit,
                                                              const float* __restrict__ bias, const E* __restrict__ res,
                                                              E* __restrict__ y, int T, int Cp, int gelu) {
  constexpr int CH = Vec16<E>::N;
  const int nch = Cp / CH;
  const long total = (long)T * nch;
  const size_t slab = (size_t)T * Cp;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int t = (int)(i / nch), q = (int)(i - (long)t * nch);
    float v[CH];
#pragma unroll
    for (int j = 0; j < CH; ++j) v[j] = bias ? bias[q * CH + j] : 0.f;
    for (int s = 0; s < ksplit; ++s) {
      const float* p = partial + s * slab + (size_t)t * Cp + q * CH;
#pragma unroll
      for (int j4 = 0; j4 < CH / 4; ++j4) {
        const float4 f = *reinterpret_cast<const float4*>(p + j4 * 4);
        v[j4 * 4 + 0] += f.x; v[j4 * 4 + 1] += f.y; v[j4 * 4 + 2] += f.z; v[j4 * 4 + 3] += f.w;
      }
    }
    if (gelu) {
#pragma unroll
      for (int j = 0; j < CH; ++j) v[j] = gelu_f<E>(v[j]);
    }
    const size_t off = (size_t)(PAD + t) * Cp + q * CH;
    if (res) {
      float r[CH];
      Vec16<E>::load(res + off, r);
#pragma unroll
      for (int j = 0; j < CH; ++j) v[j] += r[j];
    }
    Vec16<E>::store(y + off, v);
  }
}

// out[t][c] = mean of four row-layout buffers (fp32, dense [T][C])
template <typename E>
__global__ void mean4_kernel(const E* __restrict__ a, const E* __restrict__ b, const E* __restrict__ c, const E* __restrict__ d,
                             float* __restrict__ out, int T, int C, int Cp) {
  const long total = (long)T * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int t = (int)(i / C), ch = (int)(i - (long)t * C);
    const size_t off = (size_t)(PAD + t) * Cp + ch;
    // torch.stack(...).mean(0) adds in order and divides once (wav2vec_util.py:20)
    out[i] = (((Elem<E>::ld(a + off) + Elem<E>::ld(b + off)) + Elem<E>::ld(c + off)) + Elem<E>::ld(d + off)) / 4.0f;
  }
}

}  // namespace sda

using namespace sda;

#define SDA_DISPATCH(dtype, CALL)                                   \
  do {                                                              \
    if ((dtype) == SDA_F32) { using E = float; CALL; }              \
    else if ((dtype) == SDA_BF16) { using E = uint16_t; CALL; }     \
    else if ((dtype) == SDA_F16) { using E = half_t; CALL; }        \
    else { set_error("unknown dtype %d", (int)(dtype)); return -1; } \
  } while (0)

static inline int grid_for(long n, int per_block) {
  long g = (n + per_block - 1) / per_block;
  return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

extern "C" int sda_w2v_conv0(const float* wave, long n_samples, const float* w, const float* bias, const float* gamma,
                             const float* beta, void* y, int T, int C, int Cp, int K, int stride, float eps, int dtype,
                             void* stream) {
  if (!wave || !w || !gamma || !beta || !y || T < 1 || C < 1 || C > Cp || Cp % 64 || Cp > 1024 || K < 1 || stride < 1 ||
      (long)(T - 1) * stride + K > n_samples || (size_t)C * K * 4 > 64 * 1024) {
    set_error("w2v_conv0: bad arguments");
    return -1;
  }
  hipStream_t st = (hipStream_t)stream;
  const int ni = (C + 63) / 64;
  const size_t lds = (size_t)C * K * sizeof(float);
#define W2V_C0(NI) SDA_DISPATCH(dtype, hipLaunchKernelGGL((w2v_conv0_kernel<E, NI>), dim3(grid_for(T, 4)), dim3(256), lds, st, wave, \
                                                          n_samples, w, bias, gamma, beta, (E*)y, T, C, Cp, K, stride, eps))
  if (ni <= 2) W2V_C0(2); else if (ni <= 8) W2V_C0(8); else W2V_C0(16);
#undef W2V_C0
  return check_launch("w2v_conv0");
}

extern "C" int sda_layernorm_rows(const void* x, void* y, const float* gamma, const float* beta, int T, int C, int Cp,
                                  float eps, int gelu, int dtype, void* stream) {
  const int per_lane = Cp / (dtype == SDA_F32 ? 4 : 8);      // 16-byte chunks per row
  if (!x || !y || !gamma || !beta || T < 1 || C < 1 || C > Cp || Cp % 64 || per_lane > 64 * 4) {
    set_error("layernorm_rows: bad arguments (rows of at most 256 16-byte chunks)");
    return -1;
  }
  hipStream_t st = (hipStream_t)stream;
  const int ni = (per_lane + 63) / 64;
#define W2V_LN(NI) SDA_DISPATCH(dtype, hipLaunchKernelGGL((layernorm_rows_kernel<E, NI>), dim3(grid_for(T, 4)), dim3(256), 0, st, \
                                                          (const E*)x, (E*)y, gamma, beta, T, C, Cp, eps, gelu))
  if (ni <= 1) W2V_LN(1); else if (ni <= 2) W2V_LN(2); else W2V_LN(4);
#undef W2V_LN
  return check_launch("layernorm_rows");
}

extern "C" int sda_w2v_group_split(const void* h, void* xg, int T, int Hp, int gw, int gwp, int G, long group_rows, int lead,
                                   int dtype, void* stream) {
  const int ch = dtype == SDA_F32 ? 4 : 8;
  if (!h || !xg || T < 1 || gw % ch || gwp % 64 || gw > gwp || G * gw > Hp || lead < 0 || group_rows < lead + T) {
    set_error("w2v_group_split: bad arguments");
    return -1;
  }
  SDA_DISPATCH(dtype, hipLaunchKernelGGL(group_split_kernel<E>, dim3(grid_for((long)T * G * (gw / ch), 256)), dim3(256), 0,
                                         (hipStream_t)stream, (const E*)h, (E*)xg, T, Hp, gw, gwp, G, group_rows, lead));
  return check_launch("w2v_group_split");
}

extern "C" int sda_w2v_group_merge_add(const void* h, const void* yg, void* out, int T, int Hp, int gw, int gwp, int G,
                                       long yg_rows, const float* bias, int gelu, int dtype, void* stream) {
  const int ch = dtype == SDA_F32 ? 4 : 8;
  if (!h || !yg || !out || T < 1 || gw % ch || gwp % 64 || gw > gwp || G * gw > Hp || yg_rows < SDA_ROW_PAD + T) {
    set_error("w2v_group_merge_add: bad arguments");
    return -1;
  }
  SDA_DISPATCH(dtype, hipLaunchKernelGGL(group_merge_add_kernel<E>, dim3(grid_for((long)T * G * (gw / ch), 256)), dim3(256), 0,
                                         (hipStream_t)stream, (const E*)h, (const E*)yg, (E*)out, T, Hp, gw, gwp, G, yg_rows, bias, gelu));
  return check_launch("w2v_group_merge_add");
}

extern "C" int sda_w2v_attention(const void* q, const void* k, const void* vt, void* out, int T, int heads, int head_dim,
                                 long qk_pitch, long vt_pitch, long out_pitch, float scale, int dtype, void* stream) {
  const int es = dtype == SDA_F32 ? 4 : 2;
  if (!q || !k || !vt || !out || T < 1 || heads < 1 || head_dim != 64 || (qk_pitch * es) % 16 || (vt_pitch * es) % 16 ||
      vt_pitch < ((T + 63) / 64) * 64 || out_pitch < (long)heads * 64) {
    set_error("w2v_attention: bad arguments (head_dim must be 64, V^T rows must cover whole 64-key blocks)");
    return -1;
  }
  SDA_DISPATCH(dtype, hipLaunchKernelGGL(attention_kernel<E>, dim3((T + 63) / 64, heads), dim3(256), 0, (hipStream_t)stream,
                                         (const E*)q, (const E*)k, (const E*)vt, (E*)out, T, qk_pitch, vt_pitch, out_pitch, scale));
  return check_launch("w2v_attention");
}

extern "C" int sda_splitk_epilogue(const float* partial, int ksplit, const float* bias, const void* res, void* y, int T, int Cp,
                                   int gelu, int dtype, void* stream) {
  if (!partial || !y || ksplit < 1 || T < 1 || Cp % 64) { set_error("splitk_epilogue: bad arguments"); return -1; }
  SDA_DISPATCH(dtype, hipLaunchKernelGGL(splitk_epilogue_kernel<E>, dim3(grid_for((long)T * (Cp / Vec16<E>::N), 256)), dim3(256), 0,
                                         (hipStream_t)stream, partial, ksplit, bias, (const E*)res, (E*)y, T, Cp, gelu));
  return check_launch("splitk_epilogue");
}

extern "C" int sda_w2v_mean4(const void* a, const void* b, const void* c, const void* d, float* out, int T, int C, int Cp,
                             int dtype, void* stream) {
  if (!a || !b || !c || !d || !out || T < 1 || C < 1 || C > Cp) { set_error("w2v_mean4: bad arguments"); return -1; }
  SDA_DISPATCH(dtype, hipLaunchKernelGGL(mean4_kernel<E>, dim3(grid_for((long)T * C, 256)), dim3(256), 0, (hipStream_t)stream,
                                         (const E*)a, (const E*)b, (const E*)c, (const E*)d, out, T, C, Cp));
  return check_launch("w2v_mean4");
}

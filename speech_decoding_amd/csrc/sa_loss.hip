// SpatialAttention weight build / backward (models.py:45-65) and the CLIP loss tail (loss.py:64-79) with
// the retrieval ranks of Classifier (models.py:233-243).  These stages are tiny (D1 x C and B x B
// matrices): one workgroup per row or column, wavefront reductions, fp32 throughout, ordered reductions.
#include "sd_common.h"

namespace sda {

__device__ inline float block_sum(float v, float* sh) {     // 256 threads; result broadcast
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}
__device__ inline float block_max(float v, float* sh) {
  v = wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
}

// SpatialAttention weights, forward.  a[o][c] = sum_m Re z[o][m] cos[m][c] + Im z[o][m] sin[m][c].
// Stage 1: block (row group of SA_R rows, m-chunk of SA_MC) accumulates partial sums for all sensors; every
// table element it loads feeds SA_R rows (the tables are the traffic: 2 x K2 x C floats).  Stage 2: one block
// per row sums the m-chunk partials in fixed order, takes the softmax over sensors, applies the dropout mask
// and writes the packed MFMA operand.
constexpr int SA_R = 8;
constexpr int SA_MC = 128;

__global__ __launch_bounds__(256) void sa_fwd_partial_kernel(const float* __restrict__ z, const float* __restrict__ cos_t,
                                                             const float* __restrict__ sin_t, float* __restrict__ part,
                                                             int D1, int K2, int C) {
  __shared__ float zs[SA_R][SA_MC][2];
  const int o0 = blockIdx.x * SA_R, m0 = blockIdx.y * SA_MC, tid = threadIdx.x;
  for (int i = tid; i < SA_R * SA_MC; i += 256) {
    const int r = i / SA_MC, m = i - r * SA_MC;
    const bool ok = (o0 + r < D1) && (m0 + m < K2);
    zs[r][m][0] = ok ? z[((size_t)(o0 + r) * K2 + m0 + m) * 2 + 0] : 0.f;
    zs[r][m][1] = ok ? z[((size_t)(o0 + r) * K2 + m0 + m) * 2 + 1] : 0.f;
  }
  __syncthreads();
  const int mm = min(SA_MC, K2 - m0);
  for (int c = tid; c < C; c += 256) {
    float acc[SA_R];
#pragma unroll
    for (int r = 0; r < SA_R; ++r) acc[r] = 0.f;
#pragma unroll 8                                       // eight table rows in flight per thread
    for (int m = 0; m < mm; ++m) {
      const float cv = cos_t[(size_t)(m0 + m) * C + c], sv = sin_t[(size_t)(m0 + m) * C + c];
#pragma unroll
      for (int r = 0; r < SA_R; ++r) acc[r] += zs[r][m][0] * cv + zs[r][m][1] * sv;
    }
#pragma unroll
    for (int r = 0; r < SA_R; ++r)
      if (o0 + r < D1) part[((size_t)blockIdx.y * D1 + o0 + r) * C + c] = acc[r];
  }
}

template <typename E>
__global__ __launch_bounds__(256) void sa_fwd_final_kernel(const float* __restrict__ part, int nchunk, int ppitch,
                                                           const float* __restrict__ mask, float* __restrict__ W,
                                                           E* __restrict__ Wp, int D1, int C, int Cp) {
  __shared__ float sh[4];
  const int o = blockIdx.x, tid = threadIdx.x;
  if (o >= D1) {                                   // padded output rows of the packed operand
    for (int c = tid; c < Cp; c += 256) Elem<E>::st(Wp + (size_t)o * Cp + c, 0.f);
    return;
  }
  constexpr int MAXC = 2;                          // up to 512 sensors
  float a[MAXC];
  float mx = -INFINITY;
#pragma unroll
  for (int k = 0; k < MAXC; ++k) {
    const int c = tid + k * 256;
    a[k] = 0.f;
    if (c < C) {
      for (int j = 0; j < nchunk; ++j) a[k] += part[((size_t)j * D1 + o) * ppitch + c];
      mx = fmaxf(mx, a[k]);
    }
  }
  mx = block_max(mx, sh);
  float ex[MAXC], sum = 0.f;
#pragma unroll
  for (int k = 0; k < MAXC; ++k) { ex[k] = (tid + k * 256 < C) ? expf(a[k] - mx) : 0.f; sum += ex[k]; }
  sum = block_sum(sum, sh);
  const float inv = 1.f / sum;
#pragma unroll
  for (int k = 0; k < MAXC; ++k) {
    const int c = tid + k * 256;
    if (c < C) {
      const float w = ex[k] * inv;
      W[(size_t)o * C + c] = w;
      Elem<E>::st(Wp + (size_t)o * Cp + c, mask ? w * mask[c] : w);
    }
  }
  for (int c = C + tid; c < Cp; c += 256) Elem<E>::st(Wp + (size_t)o * Cp + c, 0.f);
}

// Backward: block (row group of SA_R rows, 256 Fourier modes).  Prologue: softmax backward of its rows into
// LDS (da[r][c] = W (dW - <dW, W>)); then dz[o][m] = sum_c da[o][c] * (cosT | sinT)[c][m], every transposed-table
// element feeding SA_R rows.
__global__ __launch_bounds__(256) void sa_weights_bwd_kernel(const float* __restrict__ dWd, const float* __restrict__ W,
                                                             const float* __restrict__ mask, const float* __restrict__ cosT,
                                                             const float* __restrict__ sinT, float* __restrict__ dz, int D1,
                                                             int K2, int C, int Cp) {
  __shared__ float sh[4];
  __shared__ float da[SA_R][512];
  const int o0 = blockIdx.x * SA_R, m = blockIdx.y * 256 + threadIdx.x, tid = threadIdx.x;
  for (int r = 0; r < SA_R; ++r) {
    const int o = o0 + r;
    float dot = 0.f;
    if (o < D1) {
      for (int c = tid; c < C; c += 256) {
        const float dw = dWd[(size_t)o * Cp + c] * (mask ? mask[c] : 1.f);
        da[r][c] = dw;
        dot += dw * W[(size_t)o * C + c];
      }
    }
    dot = block_sum(dot, sh);
    for (int c = tid; c < C; c += 256) da[r][c] = (o < D1) ? W[(size_t)o * C + c] * (da[r][c] - dot) : 0.f;
  }
  __syncthreads();
  if (m >= K2) return;
  float sr[SA_R], si[SA_R];
#pragma unroll
  for (int r = 0; r < SA_R; ++r) { sr[r] = 0.f; si[r] = 0.f; }
#pragma unroll 8
  for (int c = 0; c < C; ++c) {
    const float cv = cosT[(size_t)c * K2 + m], sv = sinT[(size_t)c * K2 + m];
#pragma unroll
    for (int r = 0; r < SA_R; ++r) { sr[r] += da[r][c] * cv; si[r] += da[r][c] * sv; }
  }
#pragma unroll
  for (int r = 0; r < SA_R; ++r)
    if (o0 + r < D1) {
      dz[((size_t)(o0 + r) * K2 + m) * 2 + 0] = sr[r];
      dz[((size_t)(o0 + r) * K2 + m) * 2 + 1] = si[r];
    }
}

// softmax backward only: da[o][c] = W (dW*mask - <dW*mask, W>), one block per row; padded columns of da are zeroed
__global__ __launch_bounds__(256) void sa_softmax_bwd_kernel(const float* __restrict__ dWd, int dpitch,
                                                             const float* __restrict__ W, const float* __restrict__ mask,
                                                             float* __restrict__ da, int apitch, int D1, int C) {
  __shared__ float sh[4];
  const int o = blockIdx.x, tid = threadIdx.x;
  float dot = 0.f;
  for (int c = tid; c < C; c += 256) dot += dWd[(size_t)o * dpitch + c] * (mask ? mask[c] : 1.f) * W[(size_t)o * C + c];
  dot = block_sum(dot, sh);
  for (int c = tid; c < apitch; c += 256)
    da[(size_t)o * apitch + c] = c < C ? W[(size_t)o * C + c] * (dWd[(size_t)o * dpitch + c] * (mask ? mask[c] : 1.f) - dot) : 0.f;
}

// ---------------------------------------------------------------------------------------------- loss tail
// block per row i: logits row + (max, sumexp) over the local columns; also the diagonal logit
__global__ __launch_bounds__(256) void clip_rows_kernel(const float* __restrict__ S, long s_pitch, const float* __restrict__ ysq,
                                                        const float* __restrict__ zsq, const float* __restrict__ temp,
                                                        float* __restrict__ logits, float* __restrict__ row_max,
                                                        float* __restrict__ row_sum, float* __restrict__ diag,
                                                        float* __restrict__ row_lse, int Bm, int Bn, int col0) {
  __shared__ float sh[4];
  const int i = blockIdx.x, tid = threadIdx.x;
  const float alpha = expf(temp[0]);
  const float ri = alpha / sqrtf(ysq[i]);
  float mx = -INFINITY;
  for (int j = tid; j < Bn; j += 256) {
    const float l = S[(size_t)i * s_pitch + j] * ri / sqrtf(zsq[j]);
    logits[(size_t)i * Bn + j] = l;
    mx = fmaxf(mx, l);
    if (col0 + j == i) diag[i] = l;
  }
  mx = block_max(mx, sh);
  float sum = 0.f;
  for (int j = tid; j < Bn; j += 256) sum += expf(logits[(size_t)i * Bn + j] - mx);
  sum = block_sum(sum, sh);
  if (tid == 0) {
    row_max[i] = mx; row_sum[i] = sum;
    if (row_lse) row_lse[i] = mx + logf(sum);              // the row's lse over THIS block of columns (all of them on one GPU)
    if (i < col0 || i >= col0 + Bn) diag[i] = 0.f;         // the positive lives in another rank's block
  }
}

// block per local column j: lse over all rows
__global__ __launch_bounds__(256) void clip_cols_kernel(const float* __restrict__ logits, float* __restrict__ col_lse,
                                                        int Bm, int Bn) {
  __shared__ float sh[4];
  const int j = blockIdx.x, tid = threadIdx.x;
  float mx = -INFINITY;
  for (int i = tid; i < Bm; i += 256) mx = fmaxf(mx, logits[(size_t)i * Bn + j]);
  mx = block_max(mx, sh);
  float sum = 0.f;
  for (int i = tid; i < Bm; i += 256) sum += expf(logits[(size_t)i * Bn + j] - mx);
  sum = block_sum(sum, sh);
  if (tid == 0) col_lse[j] = mx + logf(sum);
}

// block per local column j: D_ij, G_ij, r_j, the GEMM epilogue's column factor and the per-column scalar partials.
// G is an MFMA operand of the dZ product, i.e. it is stored in the compute dtype — and dL/dlogits * exp(temp) / (|Y_i||Z_j|)
// is ~1e-8 at the 8-GPU shapes (1 / (2 B_global) times 164 over norms of ~1000 each): far below fp16's range.  What is
// stored is therefore the O(1) part,  G[i][j] = (p_row + p_col - 2 delta_ij) * (ymax / |Y_i|),  ymax = max_i |Y_i|, and what
// was taken out,  cscale[j] = inv_norm * exp(temp) / (ymax |Z_j|),  multiplies the fp32 accumulator of column j's
// gradient in the GEMM's epilogue:  dZ_j = cscale_j * sum_i G_ij Y_i - rscale_j Z_j.
template <typename E>
__global__ __launch_bounds__(256) void clip_grad_kernel(const float* __restrict__ logits, const float* __restrict__ row_lse,
                                                        const float* __restrict__ col_lse, const float* __restrict__ ysq,
                                                        const float* __restrict__ zsq, const float* __restrict__ temp,
                                                        float inv_norm, int col0, E* __restrict__ G, long g_pitch,
                                                        float* __restrict__ rscale, float* __restrict__ cscale,
                                                        float* __restrict__ colpart, int Bm, int Bn) {
  __shared__ float sh[4];
  const int j = blockIdx.x, tid = threadIdx.x;
  // G has one all-zero row behind the Bm real ones (the dZ GEMM's stand-in for rows past the batch) and zero padding columns
  // up to g_pitch: written here, so the caller hands in an uninitialised buffer
  if (tid == 0) Elem<E>::st(G + (size_t)Bm * g_pitch + j, 0.f);
  if (j >= Bn) {
    for (int i = tid; i < Bm; i += 256) Elem<E>::st(G + (size_t)i * g_pitch + j, 0.f);
    return;
  }
  float ymax2 = 0.f;
  for (int i = tid; i < Bm; i += 256) ymax2 = fmaxf(ymax2, ysq[i]);
  ymax2 = block_max(ymax2, sh);
  const float ymax = sqrtf(ymax2);
  const float alpha = expf(temp[0]);
  const float cl = col_lse[j];
  float dl = 0.f;
  for (int i = tid; i < Bm; i += 256) {
    const float l = logits[(size_t)i * Bn + j];
    float d = expf(l - row_lse[i]) + expf(l - cl);
    if (i == col0 + j) d -= 2.f;
    dl += d * l;
    Elem<E>::st(G + (size_t)i * g_pitch + j, d * (ymax / sqrtf(ysq[i])));
  }
  dl = block_sum(dl, sh) * inv_norm;
  if (tid == 0) {
    const float ljj = logits[(size_t)(col0 + j) * Bn + j];
    rscale[j] = dl / zsq[j];
    cscale[j] = inv_norm * alpha / (ymax * sqrtf(zsq[j]));
    colpart[j * 2 + 0] = (row_lse[col0 + j] - ljj) + (cl - ljj);   // both CE terms of sample (col0 + j)
    colpart[j * 2 + 1] = dl;                                       // d loss / d temp share
  }
}

__global__ void clip_scalars_kernel(const float* __restrict__ colpart, float inv_norm, float* __restrict__ scalars, int Bn) {
  // one wave: lane-strided fp64 partial sums, then lane 0 adds the 64 partials in lane order (deterministic)
  __shared__ double pl[64], pd[64];
  if (blockIdx.x != 0 || threadIdx.x >= 64) return;
  double l = 0.0, dt = 0.0;
  for (int j = threadIdx.x; j < Bn; j += 64) { l += (double)colpart[j * 2]; dt += (double)colpart[j * 2 + 1]; }
  pl[threadIdx.x] = l;
  pd[threadIdx.x] = dt;
  __syncthreads();
  if (threadIdx.x != 0) return;
  l = 0.0; dt = 0.0;
  for (int k = 0; k < 64; ++k) { l += pl[k]; dt += pd[k]; }
  scalars[0] = (float)(l * (double)inv_norm);
  scalars[1] = (float)dt;
}

// block per row i: number of local columns whose logit beats the row's positive (ties: lower index wins)
__global__ __launch_bounds__(256) void clip_ranks_kernel(const float* __restrict__ logits, const float* __restrict__ diag,
                                                         int32_t* __restrict__ cnt, int Bm, int Bn, int col0) {
  __shared__ float sh[4];
  const int i = blockIdx.x, tid = threadIdx.x;
  const float d = diag[i];
  float c = 0.f;
  for (int j = tid; j < Bn; j += 256) {
    const float l = logits[(size_t)i * Bn + j];
    if (l > d || (l == d && col0 + j < i)) c += 1.f;
  }
  c = block_sum(c, sh);
  if (tid == 0) cnt[i] = (int32_t)(c + 0.5f);
}

}  // namespace sda

using namespace sda;

extern "C" int sda_sa_scratch_floats(int D1, int K2, int C) { return ((K2 + SA_MC - 1) / SA_MC) * D1 * C; }

extern "C" int sda_sa_weights_forward(const float* z, const float* cos_t, const float* sin_t, const float* mask, float* W,
                                      void* Wp, float* scratch, int D1, int K2, int C, int D1p, int Cp, int dtype, void* stream) {
  if (!z || !cos_t || !sin_t || !W || !Wp || !scratch || C > 512 || C > Cp || D1 > D1p) { set_error("sa_weights_forward: bad arguments (C <= 512)"); return -1; }
  hipStream_t st = (hipStream_t)stream;
  const int nchunk = (K2 + SA_MC - 1) / SA_MC;
  hipLaunchKernelGGL(sa_fwd_partial_kernel, dim3((D1 + SA_R - 1) / SA_R, nchunk), dim3(256), 0, st, z, cos_t, sin_t, scratch, D1, K2, C);
  if (dtype == SDA_F32)
    hipLaunchKernelGGL(sa_fwd_final_kernel<float>, dim3(D1p), dim3(256), 0, st, scratch, nchunk, C, mask, W, (float*)Wp, D1, C, Cp);
  else if (dtype == SDA_BF16)
    hipLaunchKernelGGL(sa_fwd_final_kernel<uint16_t>, dim3(D1p), dim3(256), 0, st, scratch, nchunk, C, mask, W, (uint16_t*)Wp, D1, C, Cp);
  else if (dtype == SDA_F16)
    hipLaunchKernelGGL(sa_fwd_final_kernel<half_t>, dim3(D1p), dim3(256), 0, st, scratch, nchunk, C, mask, W, (half_t*)Wp, D1, C, Cp);
  else { set_error("sa_weights_forward: unknown dtype"); return -1; }
  return check_launch("sa_weights_forward");
}

/* The two GEMMs of SpatialAttention's weight build can also run on the matrix cores (sda_conv_gemm, split-K
 * matrix mode, fp32): a = [Re z | Im z] (D1 x 2K2) . [cos | sin]^T, and dz = da . [cos ; sin] in backward.  These two
 * entry points are the non-GEMM halves: softmax over sensors + mask + operand packing, and the softmax backward. */
extern "C" int sda_sa_softmax_pack(const float* a, int a_pitch, const float* mask, float* W, void* Wp, int D1, int C,
                                   int D1p, int Cp, int dtype, void* stream) {
  if (!a || !W || !Wp || C > 512 || C > Cp || D1 > D1p || a_pitch < C) { set_error("sa_softmax_pack: bad arguments (C <= 512)"); return -1; }
  hipStream_t st = (hipStream_t)stream;
  if (dtype == SDA_F32)
    hipLaunchKernelGGL(sa_fwd_final_kernel<float>, dim3(D1p), dim3(256), 0, st, a, 1, a_pitch, mask, W, (float*)Wp, D1, C, Cp);
  else if (dtype == SDA_BF16)
    hipLaunchKernelGGL(sa_fwd_final_kernel<uint16_t>, dim3(D1p), dim3(256), 0, st, a, 1, a_pitch, mask, W, (uint16_t*)Wp, D1, C, Cp);
  else if (dtype == SDA_F16)
    hipLaunchKernelGGL(sa_fwd_final_kernel<half_t>, dim3(D1p), dim3(256), 0, st, a, 1, a_pitch, mask, W, (half_t*)Wp, D1, C, Cp);
  else { set_error("sa_softmax_pack: unknown dtype"); return -1; }
  return check_launch("sa_softmax_pack");
}

extern "C" int sda_sa_softmax_backward(const float* dWd, int dwd_pitch, const float* W, const float* mask, float* da,
                                       int da_pitch, int D1, int C, void* stream) {
  if (!dWd || !W || !da || dwd_pitch < C || da_pitch < C || D1 < 1) { set_error("sa_softmax_backward: bad arguments"); return -1; }
  hipLaunchKernelGGL(sa_softmax_bwd_kernel, dim3(D1), dim3(256), 0, (hipStream_t)stream, dWd, dwd_pitch, W, mask, da, da_pitch,
                     D1, C);
  return check_launch("sa_softmax_backward");
}

extern "C" int sda_sa_weights_backward(const float* dWd, const float* W, const float* mask, const float* cosT,
                                       const float* sinT, float* dz, int D1, int K2, int C, int Cp, void* stream) {
  if (!dWd || !W || !cosT || !sinT || !dz || C > 512) { set_error("sa_weights_backward: bad arguments"); return -1; }
  hipLaunchKernelGGL(sa_weights_bwd_kernel, dim3((D1 + SA_R - 1) / SA_R, (K2 + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                     dWd, W, mask, cosT, sinT, dz, D1, K2, C, Cp);
  return check_launch("sa_weights_backward");
}

namespace sda {
// Cross-rank merge of the loss's row statistics: every rank holds, for every GLOBAL speech row, (max, sum exp(l - max))
// over ITS block of brain columns and the positive's logit (zero where the positive lives on another rank); `all` = the
// all-gathered table [world][3][Bg].  lse[i] = M + log sum_r s_r exp(m_r - M), M = max_r m_r; diag[i] = sum_r d_r — in rank
// order, as distributed.combine_row_stats (the torch form, kept for the CPU tests) computes them.
__global__ __launch_bounds__(256) void clip_merge_rows_kernel(const float* __restrict__ all, int world, int Bg,
                                                              float* __restrict__ lse, float* __restrict__ diag) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= Bg) return;
  float M = -INFINITY;
  for (int r = 0; r < world; ++r) M = fmaxf(M, all[((size_t)r * 3 + 0) * Bg + i]);
  float S = 0.f, D = 0.f;
  for (int r = 0; r < world; ++r) {
    S += all[((size_t)r * 3 + 1) * Bg + i] * expf(all[((size_t)r * 3 + 0) * Bg + i] - M);
    D += all[((size_t)r * 3 + 2) * Bg + i];
  }
  lse[i] = M + logf(S);
  diag[i] = D;
}
}  // namespace sda

extern "C" int sda_clip_merge_rows(const float* all, int world, int Bg, float* lse, float* diag, void* stream) {
  if (!all || !lse || !diag || world < 1 || Bg < 1) { set_error("clip_merge_rows: bad arguments"); return -1; }
  hipLaunchKernelGGL(sda::clip_merge_rows_kernel, dim3((Bg + 255) / 256), dim3(256), 0, (hipStream_t)stream, all, world, Bg, lse, diag);
  return check_launch("clip_merge_rows");
}

extern "C" int sda_clip_logits_stats(const float* S, long s_pitch, const float* ysq, const float* zsq, const float* temp,
                                     float* logits, float* row_max, float* row_sum, float* col_lse, float* diag,
                                     float* row_lse, int Bm, int Bn, int col0, void* stream) {
  if (!S || !ysq || !zsq || !temp || !logits || !row_max || !row_sum || !col_lse || !diag || Bm < 1 || Bn < 1) {
    set_error("clip_logits_stats: bad arguments"); return -1;
  }
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(clip_rows_kernel, dim3(Bm), dim3(256), 0, st, S, s_pitch, ysq, zsq, temp, logits, row_max, row_sum, diag, row_lse, Bm, Bn, col0);
  hipLaunchKernelGGL(clip_cols_kernel, dim3(Bn), dim3(256), 0, st, logits, col_lse, Bm, Bn);
  return check_launch("clip_logits_stats");
}

extern "C" int sda_clip_grad(const float* logits, const float* row_lse, const float* col_lse, const float* ysq,
                             const float* zsq, const float* temp, float inv_norm, int col0, void* G, long g_pitch,
                             float* rscale, float* cscale, float* colpart, float* scalars, int Bm, int Bn, int dtype,
                             void* stream) {
  if (!logits || !row_lse || !col_lse || !ysq || !zsq || !temp || !G || !rscale || !cscale || !colpart || !scalars || g_pitch < Bn) {
    set_error("clip_grad: null argument or g_pitch < Bn"); return -1;
  }
  hipStream_t st = (hipStream_t)stream;
  if (dtype == SDA_F32)
    hipLaunchKernelGGL(clip_grad_kernel<float>, dim3((unsigned)g_pitch), dim3(256), 0, st, logits, row_lse, col_lse, ysq, zsq, temp, inv_norm, col0, (float*)G, g_pitch, rscale, cscale, colpart, Bm, Bn);
  else if (dtype == SDA_BF16)
    hipLaunchKernelGGL(clip_grad_kernel<uint16_t>, dim3((unsigned)g_pitch), dim3(256), 0, st, logits, row_lse, col_lse, ysq, zsq, temp, inv_norm, col0, (uint16_t*)G, g_pitch, rscale, cscale, colpart, Bm, Bn);
  else if (dtype == SDA_F16)
    hipLaunchKernelGGL(clip_grad_kernel<half_t>, dim3((unsigned)g_pitch), dim3(256), 0, st, logits, row_lse, col_lse, ysq, zsq, temp, inv_norm, col0, (half_t*)G, g_pitch, rscale, cscale, colpart, Bm, Bn);
  else { set_error("clip_grad: unknown dtype"); return -1; }
  hipLaunchKernelGGL(clip_scalars_kernel, dim3(1), dim3(64), 0, st, colpart, inv_norm, scalars, Bn);
  return check_launch("clip_grad");
}

extern "C" int sda_clip_ranks(const float* logits, const float* diag, int32_t* cnt, int Bm, int Bn, int col0, void* stream) {
  if (!logits || !diag || !cnt) { set_error("clip_ranks: null argument"); return -1; }
  hipLaunchKernelGGL(clip_ranks_kernel, dim3(Bm), dim3(256), 0, (hipStream_t)stream, logits, diag, cnt, Bm, Bn, col0);
  return check_launch("clip_ranks");
}

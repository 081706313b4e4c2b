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

// One block per output row o.  a[o][c] = sum_m Re z[o][m] cos[m][c] + Im z[o][m] sin[m][c]; softmax over c.
template <typename E>
__global__ __launch_bounds__(256) void sa_weights_fwd_kernel(const float* __restrict__ z, const float* __restrict__ cos_t,
                                                             const float* __restrict__ sin_t, const float* __restrict__ mask,
                                                             float* __restrict__ W, E* __restrict__ Wp, int D1, int K2,
                                                             int C, int Cp) {
  __shared__ float sh[4];
  __shared__ float zs[2][256];
  const int o = blockIdx.x, tid = threadIdx.x;
  if (o >= D1) {                                   // padded output rows of the packed operand
    for (int c = tid; c < Cp; c += 256) Elem<E>::st(Wp + (size_t)o * Cp + c, 0.f);
    return;
  }
  constexpr int MAXC = 2;                          // up to 512 sensors
  float acc[MAXC] = {0.f, 0.f};
  for (int m0 = 0; m0 < K2; m0 += 256) {
    __syncthreads();
    if (m0 + tid < K2) {
      zs[0][tid] = z[((size_t)o * K2 + m0 + tid) * 2 + 0];
      zs[1][tid] = z[((size_t)o * K2 + m0 + tid) * 2 + 1];
    }
    __syncthreads();
    const int mm = min(256, K2 - m0);
#pragma unroll
    for (int k = 0; k < MAXC; ++k) {
      const int c = tid + k * 256;
      if (c < C) {
        float s = acc[k];
        for (int m = 0; m < mm; ++m)
          s += zs[0][m] * cos_t[(size_t)(m0 + m) * C + c] + zs[1][m] * sin_t[(size_t)(m0 + m) * C + c];
        acc[k] = s;
      }
    }
  }
  float mx = -INFINITY;
#pragma unroll
  for (int k = 0; k < MAXC; ++k) if (tid + k * 256 < C) mx = fmaxf(mx, acc[k]);
  mx = block_max(mx, sh);
  float ex[MAXC], sum = 0.f;
#pragma unroll
  for (int k = 0; k < MAXC; ++k) { ex[k] = (tid + k * 256 < C) ? expf(acc[k] - mx) : 0.f; sum += ex[k]; }
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

// One block per row o: softmax backward on the row, then dz[o][m] = sum_c da[c] * (cosT | sinT)[c][m].
__global__ __launch_bounds__(256) void sa_weights_bwd_kernel(const float* __restrict__ dWd, const float* __restrict__ W,
                                                             const float* __restrict__ mask, const float* __restrict__ cosT,
                                                             const float* __restrict__ sinT, float* __restrict__ dz, int D1,
                                                             int K2, int C, int Cp) {
  __shared__ float sh[4];
  __shared__ float da[512];
  const int o = blockIdx.x, tid = threadIdx.x;
  float dot = 0.f;
  for (int c = tid; c < C; c += 256) {
    const float dw = dWd[(size_t)o * Cp + c] * (mask ? mask[c] : 1.f);
    da[c] = dw;
    dot += dw * W[(size_t)o * C + c];
  }
  dot = block_sum(dot, sh);
  for (int c = tid; c < C; c += 256) da[c] = W[(size_t)o * C + c] * (da[c] - dot);
  __syncthreads();
  for (int m = tid; m < K2; m += 256) {
    float sr = 0.f, si = 0.f;
    for (int c = 0; c < C; ++c) {
      sr += da[c] * cosT[(size_t)c * K2 + m];
      si += da[c] * sinT[(size_t)c * K2 + m];
    }
    dz[((size_t)o * K2 + m) * 2 + 0] = sr;
    dz[((size_t)o * K2 + m) * 2 + 1] = si;
  }
}

// ---------------------------------------------------------------------------------------------- loss tail
// block per row i: logits row + (max, sumexp) over the local columns; also the diagonal logit
__global__ __launch_bounds__(256) void clip_rows_kernel(const float* __restrict__ S, long s_pitch, const float* __restrict__ ysq,
                                                        const float* __restrict__ zsq, const float* __restrict__ temp,
                                                        float* __restrict__ logits, float* __restrict__ row_max,
                                                        float* __restrict__ row_sum, float* __restrict__ diag, int Bm, int Bn,
                                                        int col0) {
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
  if (tid == 0) { row_max[i] = mx; row_sum[i] = sum; }
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

// block per local column j: D_ij, G_ij, r_j and the per-column scalar partials
template <typename E>
__global__ __launch_bounds__(256) void clip_grad_kernel(const float* __restrict__ logits, const float* __restrict__ row_lse,
                                                        const float* __restrict__ col_lse, const float* __restrict__ ysq,
                                                        const float* __restrict__ zsq, const float* __restrict__ temp,
                                                        float inv_norm, int col0, E* __restrict__ G, long g_pitch,
                                                        float* __restrict__ rscale, float* __restrict__ colpart, int Bm,
                                                        int Bn) {
  __shared__ float sh[4];
  const int j = blockIdx.x, tid = threadIdx.x;
  const float alpha = expf(temp[0]);
  const float cl = col_lse[j];
  const float mj = alpha / sqrtf(zsq[j]);
  float dl = 0.f;
  for (int i = tid; i < Bm; i += 256) {
    const float l = logits[(size_t)i * Bn + j];
    float d = expf(l - row_lse[i]) + expf(l - cl);
    if (i == col0 + j) d -= 2.f;
    d *= inv_norm;
    dl += d * l;
    Elem<E>::st(G + (size_t)i * g_pitch + j, d * mj / sqrtf(ysq[i]));
  }
  dl = block_sum(dl, sh);
  if (tid == 0) {
    const float ljj = logits[(size_t)(col0 + j) * Bn + j];
    rscale[j] = dl / zsq[j];
    colpart[j * 2 + 0] = (row_lse[col0 + j] - ljj) + (cl - ljj);   // both CE terms of sample (col0 + j)
    colpart[j * 2 + 1] = dl;                                       // d loss / d temp share
  }
}

__global__ void clip_scalars_kernel(const float* __restrict__ colpart, float inv_norm, float* __restrict__ scalars, int Bn) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double l = 0.0, dt = 0.0;
  for (int j = 0; j < Bn; ++j) { l += (double)colpart[j * 2]; dt += (double)colpart[j * 2 + 1]; }
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

extern "C" int sda_sa_weights_forward(const float* z, const float* cos_t, const float* sin_t, const float* mask, float* W,
                                      void* Wp, int D1, int K2, int C, int D1p, int Cp, int dtype, void* stream) {
  if (!z || !cos_t || !sin_t || !W || !Wp || C > 512 || C > Cp || D1 > D1p) { set_error("sa_weights_forward: bad arguments (C <= 512)"); return -1; }
  hipStream_t st = (hipStream_t)stream;
  if (dtype == SDA_F32)
    hipLaunchKernelGGL(sa_weights_fwd_kernel<float>, dim3(D1p), dim3(256), 0, st, z, cos_t, sin_t, mask, W, (float*)Wp, D1, K2, C, Cp);
  else if (dtype == SDA_BF16)
    hipLaunchKernelGGL(sa_weights_fwd_kernel<uint16_t>, dim3(D1p), dim3(256), 0, st, z, cos_t, sin_t, mask, W, (uint16_t*)Wp, D1, K2, C, Cp);
  else { set_error("sa_weights_forward: unknown dtype"); return -1; }
  return check_launch("sa_weights_forward");
}

extern "C" int sda_sa_weights_backward(const float* dWd, const float* W, const float* mask, const float* cosT,
                                       const float* sinT, float* dz, int D1, int K2, int C, int Cp, void* stream) {
  if (!dWd || !W || !cosT || !sinT || !dz || C > 512) { set_error("sa_weights_backward: bad arguments"); return -1; }
  hipLaunchKernelGGL(sa_weights_bwd_kernel, dim3(D1), dim3(256), 0, (hipStream_t)stream, dWd, W, mask, cosT, sinT, dz, D1, K2, C, Cp);
  return check_launch("sa_weights_backward");
}

extern "C" int sda_clip_logits_stats(const float* S, long s_pitch, const float* ysq, const float* zsq, const float* temp,
                                     float* logits, float* row_max, float* row_sum, float* col_lse, float* diag, int Bm,
                                     int Bn, int col0, void* stream) {
  if (!S || !ysq || !zsq || !temp || !logits || !row_max || !row_sum || !col_lse || !diag || Bm < 1 || Bn < 1) {
    set_error("clip_logits_stats: bad arguments"); return -1;
  }
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(clip_rows_kernel, dim3(Bm), dim3(256), 0, st, S, s_pitch, ysq, zsq, temp, logits, row_max, row_sum, diag, Bm, Bn, col0);
  hipLaunchKernelGGL(clip_cols_kernel, dim3(Bn), dim3(256), 0, st, logits, col_lse, Bm, Bn);
  return check_launch("clip_logits_stats");
}

extern "C" int sda_clip_grad(const float* logits, const float* row_lse, const float* col_lse, const float* ysq,
                             const float* zsq, const float* temp, float inv_norm, int col0, void* G, long g_pitch,
                             float* rscale, float* colpart, float* scalars, int Bm, int Bn, int dtype, void* stream) {
  if (!logits || !row_lse || !col_lse || !ysq || !zsq || !temp || !G || !rscale || !colpart || !scalars) {
    set_error("clip_grad: null argument"); return -1;
  }
  hipStream_t st = (hipStream_t)stream;
  if (dtype == SDA_F32)
    hipLaunchKernelGGL(clip_grad_kernel<float>, dim3(Bn), dim3(256), 0, st, logits, row_lse, col_lse, ysq, zsq, temp, inv_norm, col0, (float*)G, g_pitch, rscale, colpart, Bm, Bn);
  else if (dtype == SDA_BF16)
    hipLaunchKernelGGL(clip_grad_kernel<uint16_t>, dim3(Bn), dim3(256), 0, st, logits, row_lse, col_lse, ysq, zsq, temp, inv_norm, col0, (uint16_t*)G, g_pitch, rscale, colpart, Bm, Bn);
  else { set_error("clip_grad: unknown dtype"); return -1; }
  hipLaunchKernelGGL(clip_scalars_kernel, dim3(1), dim3(64), 0, st, colpart, inv_norm, scalars, Bn);
  return check_launch("clip_grad");
}

extern "C" int sda_clip_ranks(const float* logits, const float* diag, int32_t* cnt, int Bm, int Bn, int col0, void* stream) {
  if (!logits || !diag || !cnt) { set_error("clip_ranks: null argument"); return -1; }
  hipLaunchKernelGGL(clip_ranks_kernel, dim3(Bm), dim3(256), 0, (hipStream_t)stream, logits, diag, cnt, Bm, Bn, col0);
  return check_launch("clip_ranks");
}

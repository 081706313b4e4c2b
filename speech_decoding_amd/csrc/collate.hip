// GPU collate: the per-batch preprocessing the reference runs in Python/sklearn on DataLoader
// workers — Gwilliams2022Collator.forward (gwilliams2022.py:651-661) = baseline_correction_single
// (preproc_utils.py:128-142) + scaleAndClamp (preproc_utils.py:69-90): for every (sample, channel) row of T
// samples subtract the mean of the first `nb` samples, then RobustScaler (sklearn: centre = median, scale =
// 75th - 25th percentile with linear interpolation, zero scale -> 1) over time, then clamp to +-lim.
// One wavefront per row: the row is sorted in LDS (bitonic, padded with +inf) to read the three quantiles.
#include "sd_common.h"

namespace sda {

// Row sources: either a dense (rows, T) matrix, or — segment gather, gwilliams2022.py:129-142 — per-sample
// windows of resident session recordings: sample b = rows [b*C, (b+1)*C) read win_ptr[b] + c * win_cstride[b].
template <int NPL>      // slots per lane; sorts 64*NPL values (T <= 64*NPL)
__global__ __launch_bounds__(256) void collate_rows_kernel(const float* __restrict__ src, const float* const* __restrict__ win_ptr,
                                                           const long* __restrict__ win_cstride, int C,
                                                           float* __restrict__ dst, long rows, int T, int nb, float lim,
                                                           int do_clamp) {
  constexpr int N = 64 * NPL;
  __shared__ float sbuf[4][N];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const long row = (long)blockIdx.x * 4 + wid;
  const bool live = row < rows;
  float* s = sbuf[wid];
  const long r = live ? row : 0;
  const float* x = win_ptr ? win_ptr[r / C] + (r % C) * win_cstride[r / C] : src + r * T;

  float v[NPL];
  float bsum = 0.f;
#pragma unroll
  for (int i = 0; i < NPL; ++i) {
    const int t = lane + 64 * i;
    v[i] = (t < T) ? x[t] : 0.f;
    if (t < nb) bsum += v[i];
  }
  const float base = nb > 0 ? wave_sum(bsum) / (float)nb : 0.f;
#pragma unroll
  for (int i = 0; i < NPL; ++i) {
    const int t = lane + 64 * i;
    v[i] -= base;
    s[t] = (t < T) ? v[i] : INFINITY;
  }
  __syncthreads();
  for (int k = 2; k <= N; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
#pragma unroll
      for (int i = 0; i < NPL / 2; ++i) {
        const int p = lane + 64 * i;                       // pair index 0 .. N/2-1
        const int lo = ((p & ~(j - 1)) << 1) | (p & (j - 1));
        const int hi = lo | j;
        const bool up = (lo & k) == 0;
        const float a = s[lo], b = s[hi];
        if ((a > b) == up) { s[lo] = b; s[hi] = a; }
      }
      __syncthreads();
    }
  }
  auto quantile = [&](float q) {
    const float pos = q * (float)(T - 1);
    const int i0 = (int)floorf(pos);
    const float g = pos - (float)i0;
    const float a = s[i0], b = s[min(i0 + 1, T - 1)];
    return a + (b - a) * g;                                // numpy's default "linear" interpolation
  };
  const float med = quantile(0.5f);
  float iqr = quantile(0.75f) - quantile(0.25f);
  if (iqr == 0.f) iqr = 1.f;                               // sklearn _handle_zeros_in_scale
  if (live) {
    float* y = dst + row * T;
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
      const int t = lane + 64 * i;
      if (t < T) {
        float o = (v[i] - med) / iqr;
        if (do_clamp) o = fminf(fmaxf(o, -lim), lim);
        y[t] = o;
      }
    }
  }
}

}  // namespace sda

using namespace sda;

extern "C" int sda_collate_rows(const float* src, float* dst, long rows, int T, int baseline_len, float clamp_lim, int clamp,
                                void* stream) {
  if (!src || !dst || rows < 1 || T < 2 || baseline_len < 0 || baseline_len > T) { set_error("collate_rows: bad arguments"); return -1; }
  if (T > 1024) { set_error("collate_rows: T = %d exceeds the 1024 samples one wavefront sorts", T); return -1; }
  hipStream_t st = (hipStream_t)stream;
  const unsigned grid = (unsigned)((rows + 3) / 4);
  if (T <= 512) hipLaunchKernelGGL(collate_rows_kernel<8>, dim3(grid), dim3(256), 0, st, src, nullptr, nullptr, 1, dst, rows, T, baseline_len, clamp_lim, clamp);
  else hipLaunchKernelGGL(collate_rows_kernel<16>, dim3(grid), dim3(256), 0, st, src, nullptr, nullptr, 1, dst, rows, T, baseline_len, clamp_lim, clamp);
  return check_launch("collate_rows");
}

extern "C" int sda_collate_windows(const float* const* win_ptr, const long* win_cstride, float* dst, int B, int C, int T,
                                   int baseline_len, float clamp_lim, int clamp, void* stream) {
  if (!win_ptr || !win_cstride || !dst || B < 1 || C < 1 || T < 2 || baseline_len < 0 || baseline_len > T) { set_error("collate_windows: bad arguments"); return -1; }
  if (T > 1024) { set_error("collate_windows: T = %d exceeds the 1024 samples one wavefront sorts", T); return -1; }
  hipStream_t st = (hipStream_t)stream;
  const long rows = (long)B * C;
  const unsigned grid = (unsigned)((rows + 3) / 4);
  if (T <= 512) hipLaunchKernelGGL(collate_rows_kernel<8>, dim3(grid), dim3(256), 0, st, nullptr, win_ptr, win_cstride, C, dst, rows, T, baseline_len, clamp_lim, clamp);
  else hipLaunchKernelGGL(collate_rows_kernel<16>, dim3(grid), dim3(256), 0, st, nullptr, win_ptr, win_cstride, C, dst, rows, T, baseline_len, clamp_lim, clamp);
  return check_launch("collate_windows");
}

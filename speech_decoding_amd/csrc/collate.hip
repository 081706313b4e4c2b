// GPU collate: the per-batch preprocessing the reference runs in Python/sklearn on DataLoader
// workers — Gwilliams2022Collator.forward (gwilliams2022.py:651-661) = baseline_correction_single
// (preproc_utils.py:128-142) + scaleAndClamp (preproc_utils.py:69-90): for every (sample, channel) row of T
// samples subtract the mean of the first `nb` samples, then RobustScaler (sklearn: centre = median, scale =
// 75th - 25th percentile with linear interpolation, zero scale -> 1) over time, then clamp to +-lim.
// One wavefront per row: the row is sorted in registers (bitonic over registers x lanes, padded with +inf) and laid out
// in LDS once to read the three quantiles.
#include "sd_common.h"
#include "flat_tile.h"

namespace sda {

// Row sources: either a dense (rows, T) matrix, or — segment gather, gwilliams2022.py:129-142 — per-sample
// windows of resident session recordings: sample b = rows [b*C, (b+1)*C) read win_ptr[b] + c * win_cstride[b].
template <int NPL>      // slots per lane; sorts 64*NPL values (T <= 64*NPL)
__global__ __launch_bounds__(256) void collate_rows_kernel(const float* __restrict__ src, const float* const* __restrict__ win_ptr,
                                                           const long* __restrict__ win_cstride, int C,
                                                           float* __restrict__ dst, long rows, int T, int nb, float lim,
                                                           int do_clamp) {
  constexpr int N = 64 * NPL;
  constexpr int LOG2N = NPL == 8 ? 9 : 10;
  static_assert((1 << LOG2N) == N, "NPL is 8 or 16");
  __shared__ float sbuf[4][N];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const long row = (long)blockIdx.x * 4 + wid;
  const bool live = row < rows;
  float* s = sbuf[wid];
  const long r = live ? row : 0;
  const float* x = win_ptr ? win_ptr[r / C] + (r % C) * win_cstride[r / C] : src + r * T;

  // Element t of the row lives in register t % NPL of lane t / NPL (a lane holds NPL CONSECUTIVE samples: 16-byte global
  // accesses where the row allows them).
  float v[NPL];
  const bool vec = (T % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0) && ((reinterpret_cast<uintptr_t>(dst) & 15) == 0);
  if (vec) {
#pragma unroll
    for (int q = 0; q < NPL / 4; ++q) {
      const int t = lane * NPL + 4 * q;
      float4 f = make_float4(0.f, 0.f, 0.f, 0.f);
      if (t < T) f = *reinterpret_cast<const float4*>(x + t);
      v[4 * q + 0] = f.x; v[4 * q + 1] = f.y; v[4 * q + 2] = f.z; v[4 * q + 3] = f.w;
    }
  } else {
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
      const int t = lane * NPL + i;
      v[i] = (t < T) ? x[t] : 0.f;
    }
  }
  float bsum = 0.f;
#pragma unroll
  for (int i = 0; i < NPL; ++i) {
    if (lane * NPL + i < nb) bsum += v[i];
  }
  const float base = nb > 0 ? wave_sum(bsum) / (float)nb : 0.f;
  // Bitonic sort of the wave's 64 * NPL values IN REGISTERS (padding = +inf).  A compare-exchange at distance < NPL pairs
  // two registers of one lane; beyond, lane l with another lane: DPP (quad permutations, row_half_mirror, row_mirror: no
  // LDS involved) for lane masks 1, 2, 3, 7, 15, one ds_bpermute per register otherwise.  For 512 values: 24 of the 45
  // stages stay inside the lane, 13 are DPP, 8 go through the LDS crossbar.  History (config 2, 256 x 208 rows of 360): an
  // LDS image with a workgroup barrier per stage 264 us; registers with lane-major elements (39 crossbar stages) 204 us;
  // lane-contiguous elements with per-pair direction selects 135 us.
  float w[NPL];
#pragma unroll
  for (int i = 0; i < NPL; ++i) {
    v[i] -= base;
    w[i] = (lane * NPL + i < T) ? v[i] : INFINITY;
  }
  // The network is the all-ascending form of the bitonic sort: merge step k first compares element e with e ^ (k - 1) (its
  // mirror image inside the k-block), then with e ^ j for j = k / 4 ... 1 — the element with the lower index always keeps
  // the minimum, so there is no direction to select.  Inside a lane that is v_min + v_max per pair; between lanes ONE
  // v_med3_f32 per register against a per-lane constant (-inf in the lower lane: med3(a, b, -inf) = min; +inf in the upper).
  auto shuffle = [&](float val, auto mc) -> float {          // the value of lane (lane ^ m), m = 2^n - 1 (mirror) or a power of two
    constexpr int m = decltype(mc)::value;
    const int iv = __float_as_int(val);
    if constexpr (m == 1) return __int_as_float(__builtin_amdgcn_update_dpp(iv, iv, 0xB1, 0xF, 0xF, false));        // quad_perm [1,0,3,2]
    else if constexpr (m == 2) return __int_as_float(__builtin_amdgcn_update_dpp(iv, iv, 0x4E, 0xF, 0xF, false));   // quad_perm [2,3,0,1]
    else if constexpr (m == 3) return __int_as_float(__builtin_amdgcn_update_dpp(iv, iv, 0x1B, 0xF, 0xF, false));   // quad_perm [3,2,1,0]
    else if constexpr (m == 7) return __int_as_float(__builtin_amdgcn_update_dpp(iv, iv, 0x141, 0xF, 0xF, false));  // row_half_mirror
    else if constexpr (m == 15) return __int_as_float(__builtin_amdgcn_update_dpp(iv, iv, 0x140, 0xF, 0xF, false)); // row_mirror
    else return __shfl_xor(val, m);
  };
  static_for<1, LOG2N + 1>([&](auto kc) {
    constexpr int k = 1 << decltype(kc)::value;
    // --- mirror step: e <-> e ^ (k - 1)
    if constexpr (k <= NPL) {
#pragma unroll
      for (int i = 0; i < NPL; ++i) {
        if ((i & (k >> 1)) == 0) {
          const float a = w[i], b = w[i ^ (k - 1)];
          w[i] = fminf(a, b);
          w[i ^ (k - 1)] = fmaxf(a, b);
        }
      }
    } else {
      constexpr int lm = k / NPL - 1;                        // lane mask; the partner's registers are in reverse order
      const float sel = (lane & ((lm + 1) >> 1)) == 0 ? -INFINITY : INFINITY;
      float p[NPL];
#pragma unroll
      for (int i = 0; i < NPL; ++i) p[i] = shuffle(w[NPL - 1 - i], std::integral_constant<int, lm>{});
#pragma unroll
      for (int i = 0; i < NPL; ++i) w[i] = __builtin_amdgcn_fmed3f(w[i], p[i], sel);
    }
    // --- e <-> e ^ j, j = k / 4 ... 1
    static_for<1, decltype(kc)::value>([&](auto jc) {
      constexpr int j = k >> (decltype(jc)::value + 1);
      if constexpr (j < NPL) {
#pragma unroll
        for (int i = 0; i < NPL; ++i) {
          if ((i & j) == 0) {
            const float a = w[i], b = w[i | j];
            w[i] = fminf(a, b);
            w[i | j] = fmaxf(a, b);
          }
        }
      } else {
        constexpr int d = j / NPL;
        const float sel = (lane & d) == 0 ? -INFINITY : INFINITY;
#pragma unroll
        for (int i = 0; i < NPL; ++i) w[i] = __builtin_amdgcn_fmed3f(w[i], shuffle(w[i], std::integral_constant<int, d>{}), sel);
      }
    });
  });
  // the sorted row, once, for the three quantile reads (same wave writes and reads: the barrier only orders them)
#pragma unroll
  for (int i = 0; i < NPL; ++i) s[lane * NPL + i] = w[i];
  __syncthreads();
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
    float o[NPL];
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
      o[i] = (v[i] - med) / iqr;
      if (do_clamp) o[i] = fminf(fmaxf(o[i], -lim), lim);
    }
    if (vec) {
#pragma unroll
      for (int q = 0; q < NPL / 4; ++q) {
        const int t = lane * NPL + 4 * q;
        if (t < T) *reinterpret_cast<float4*>(y + t) = make_float4(o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]);
      }
    } else {
#pragma unroll
      for (int i = 0; i < NPL; ++i) {
        const int t = lane * NPL + i;
        if (t < T) y[t] = o[i];
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

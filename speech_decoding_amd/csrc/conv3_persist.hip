// conv3_persist — the kernel-3 dilated Conv1d of conv_gemm.hip as ONE persistent workgroup per CU whose
// epilogue is taken off the critical path.
//
// Reference ops: nn.Conv1d(kernel_size=3, dilation=d, padding="same") of ConvBlock (models.py:128-150) and its
// input gradient; same sda_conv_args contract as conv_gemm (bias, residual, BatchNorm statistics, bn_x mode).
//
// Why a second kernel.  Measured on conv_gemm (SQ counters + what-if builds): its K loop is bound by
// the LDS-DMA bytes a CU can keep in flight, and its epilogue (residual read + output write, HBM-bound) runs
// with the matrix cores idle.  Here
//   * a workgroup = 8 waves computes TWO 128-row tiles against one weight slab (37 % fewer DMA bytes per FLOP)
//     with three LDS stages (two K slabs in flight behind a counted s_waitcnt vmcnt(7));
//   * the workgroup is persistent (grid = one per CU, XCD-aware unit order) and the DMA stream runs ahead
//     ACROSS tiles, so a new tile's first slabs are already in flight while the previous one finishes;
//   * MFMA operand roles are swapped (D[row = co][col = t]): a lane then owns 4 CONSECUTIVE channels of one
//     row, so the epilogue stores 8/16-byte pieces straight from registers — no LDS staging, no barriers;
//   * the finished tile's accumulators move to a second register set and its epilogue is issued in four
//     16-row slices inside the first four K iterations of the NEXT tile: the HBM traffic of tile i overlaps
//     the MFMAs of tile i+1.  Only a workgroup's last tile pays its epilogue in the open.
#include "conv_tile.h"

#include <type_traits>

namespace sda {

namespace {

constexpr int P_CO = 160;                               // output channels per workgroup
constexpr int P_NREP = 5;                               // 16-channel tiles per wave (wave tile: 80 co x 64 t)
constexpr int P_HALF = 80;
constexpr int P_TP = P_CO / 16;                         // 1 KB weight pieces per tap
constexpr int P_STAGE = 2 * XS_BYTES + 3 * P_CO * ROW_B;   // two input tiles + [3][160] weight rows = 50 KB
constexpr int P_NS = 3;
constexpr int P_RED_FLOATS = 8 * 2 * P_HALF;            // per wave [2][80] statistics partials
constexpr int P_LDS = P_NS * P_STAGE + P_RED_FLOATS * 4;
static_assert(P_LDS <= 160 * 1024, "persistent conv does not fit the CU's LDS");

// 4 consecutive elements kept packed until use
template <typename E> struct Q4;
template <> struct Q4<uint16_t> {
  typedef uint2 raw;
  __device__ static raw zero() { return make_uint2(0u, 0u); }
  __device__ static raw load(const uint16_t* p) { return *reinterpret_cast<const uint2*>(p); }
  __device__ static float4 unpack(const raw& u) {
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                       __uint_as_float(u.y & 0xffff0000u));
  }
};
template <> struct Q4<float> {
  typedef uint4 raw;
  __device__ static raw zero() { return make_uint4(0u, 0u, 0u, 0u); }
  __device__ static raw load(const float* p) { return *reinterpret_cast<const uint4*>(p); }
  __device__ static float4 unpack(const raw& u) {
    return make_float4(__uint_as_float(u.x), __uint_as_float(u.y), __uint_as_float(u.z), __uint_as_float(u.w));
  }
};

// sum over the 16 lanes of a DPP row (every lane ends with the total; fixed tree => deterministic)
__device__ inline float row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));  // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));  // row_mirror
  return v;
}

struct Unit {            // one (pair of t tiles, co tile); the fields are per 4-wave group (its own t tile)
  int co0, t0, tt;
  long row_base;         // global row of LDS input row 0 (t0 - dil)
  long out_row0;         // global row of output row t0
  bool ok;               // this group's tile exists (odd tile counts leave the last pair half empty)
};

template <typename E, bool BN>
__global__ __launch_bounds__(512, 1) void conv3_persist_kernel(const sda_conv_args a, const int n_t_tiles,
                                                               const int n_tpairs, const int units_per_xcd,
                                                               const int wg_per_xcd) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int SLAB = ROW_B / (int)sizeof(E);
  constexpr int PER16 = Elem<E>::PER16;
  typedef typename Q4<E>::raw raw4;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tsel = wid >> 2;
  const int wave_t = (wid >> 1) & 1, wave_c = wid & 1;
  const int lr = lane & 15, lq = lane >> 4;
  const int prow = lane >> 2, pchunk = lane & 3;

  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int n_co = a.Cout_p / P_CO;
  const int tiles_total = a.B * n_t_tiles;
  const int dil = a.dil;
  const int x_pieces = (TILE_T + 2 * dil + 15) >> 4;
  const int nslab = a.Cin_p / SLAB;

  // number of units of this workgroup: u = slot + k * wg_per_xcd while the unit exists
  int n_my = 0;
  for (int u = slot; u < units_per_xcd; u += wg_per_xcd) {
    if ((u / n_co) * 8 + xcd < n_tpairs) ++n_my;
  }
  if (n_my == 0) return;
  const int total_slabs = n_my * nslab;

  auto unit_of = [&](int k) __attribute__((always_inline)) {
    const int u = slot + k * wg_per_xcd;
    const int tp = (u / n_co) * 8 + xcd;
    Unit U;
    U.co0 = (u % n_co) * P_CO;
    const int tt = 2 * tp + tsel;
    U.ok = tt < tiles_total;
    U.tt = U.ok ? tt : tiles_total - 1;                 // an absent tile re-reads its neighbour; nothing is stored
    const int b = U.tt / n_t_tiles;
    U.t0 = (U.tt - b * n_t_tiles) * TILE_T;
    U.out_row0 = a.x_row0 + (long)b * a.x_sample_rows + U.t0;
    U.row_base = U.out_row0 - dil;
    return U;
  };

  const E* __restrict__ xg = reinterpret_cast<const E*>(a.x);
  const E* __restrict__ wg = reinterpret_cast<const E*>(a.w);
  E* __restrict__ yg = reinterpret_cast<E*>(a.y);
  const E* __restrict__ resg = reinterpret_cast<const E*>(a.res);
  const E* __restrict__ bnx = reinterpret_cast<const E*>(a.bn_x);
  float* red = reinterpret_cast<float*>(smem + P_NS * P_STAGE) + wid * 2 * P_HALF;

  // ---- LDS-DMA of one K slab: exactly 3 input pieces + 4 weight pieces per wave (clamped duplicates rewrite
  // identical bytes), so that the counted vmcnt(7) below retires slab g while slab g+1 stays in flight
  auto stage_x = [&](const Unit& U, int s, int buf) __attribute__((always_inline)) {
    unsigned char* xs = smem + buf * P_STAGE + tsel * XS_BYTES;
    const size_t koff = (size_t)s * SLAB;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      int p = (wid & 3) + 4 * i;
      p = p < x_pieces ? p : x_pieces - 1;
      const int r = p * 16 + prow;
      long row = U.row_base + r;
      row = row < 0 ? 0 : (row >= a.x_rows_limit ? a.x_rows_limit - 1 : row);
      const int lc = pchunk ^ sw64(r);
      lds_dma16(xg + (size_t)row * a.x_pitch + koff + lc * PER16,
                __builtin_amdgcn_readfirstlane(lds_addr(xs + p * 1024)));
    }
  };
  auto stage_w = [&](const Unit& U, int s, int buf, int i0, int i1) __attribute__((always_inline)) {
    unsigned char* ws = smem + buf * P_STAGE + 2 * XS_BYTES;
    const size_t koff = (size_t)s * SLAB;
#pragma unroll
    for (int i = i0; i < i1; ++i) {
      int q = wid + 8 * i;
      q = q < 3 * P_TP ? q : 3 * P_TP - 1;
      const int tap = q / P_TP;
      const int r = q * 16 + prow;                        // row of the [tap][co] weight image
      int co = U.co0 + (q - tap * P_TP) * 16 + prow;
      co = co < a.w_rows_limit ? co : a.w_rows_limit - 1;
      const int lc = pchunk ^ sw64(r);
      lds_dma16(wg + ((size_t)tap * a.Cout_p + co) * a.w_pitch + koff + lc * PER16,
                __builtin_amdgcn_readfirstlane(lds_addr(ws + q * 1024)));
    }
  };

  f32x4 acc[4][P_NREP], pend[4][P_NREP];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n = 0; n < P_NREP; ++n) { acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f}; pend[m][n] = f32x4{0.f, 0.f, 0.f, 0.f}; }

  auto compute_tap = [&](const unsigned char* xs, const unsigned char* ws, int tap) __attribute__((always_inline)) {
    uint4 af[4], bf[P_NREP];
    const int xrow = wave_t * 64 + lr + tap * dil;
    const int wrow = tap * P_CO + wave_c * P_HALF + lr;
#pragma unroll
    for (int m = 0; m < 4; ++m) af[m] = *reinterpret_cast<const uint4*>(xs + lds_sw64(xrow + m * 16, lq));
#pragma unroll
    for (int n = 0; n < P_NREP; ++n) bf[n] = *reinterpret_cast<const uint4*>(ws + lds_sw64(wrow + n * 16, lq));
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int n = 0; n < P_NREP; ++n) acc[m][n] = mma16<E>(bf[n], af[m], acc[m][n]);   // D[co = 4*lq + r][t = lr]
  };

  // ---- deferred epilogue of the pending tile, slice m = rows [16m, 16m + 16) of this wave's 64
  Unit P{};                                              // the pending tile
  bool pend_valid = false;
  raw4 rres[P_NREP], rbnx[BN ? P_NREP : 1];
  auto slice_row = [&](int m) __attribute__((always_inline)) { return wave_t * 64 + m * 16 + lr; };
  auto epi_loads = [&](int m) __attribute__((always_inline)) {                          // residual / BatchNorm-input rows of the slice
    const int tr = slice_row(m);
    const bool valid = P.ok && (P.t0 + tr < a.T);
    const size_t base = (size_t)(P.out_row0 + tr) * a.Cout_p + P.co0 + wave_c * P_HALF + lq * 4;
#pragma unroll
    for (int n = 0; n < P_NREP; ++n) {
      rres[n] = Q4<E>::zero();
      if (resg && valid) rres[n] = Q4<E>::load(resg + base + n * 16);
      if constexpr (BN) {
        rbnx[n] = Q4<E>::zero();
        if (valid) rbnx[n] = Q4<E>::load(bnx + base + n * 16);
      }
    }
  };
  auto epi_math = [&](auto MC) __attribute__((always_inline)) {
    constexpr int m = decltype(MC)::value;
    const int tr = slice_row(m);
    const bool valid = P.ok && (P.t0 + tr < a.T);
    const int cw = P.co0 + wave_c * P_HALF + lq * 4;     // first of this lane's 4 channels at n = 0
    const size_t base = (size_t)(P.out_row0 + tr) * a.Cout_p + cw;
#pragma unroll
    for (int n = 0; n < P_NREP; ++n) {
      float v[4] = {pend[m][n][0], pend[m][n][1], pend[m][n][2], pend[m][n][3]};
      if (a.bias) {
        const float4 bv = *reinterpret_cast<const float4*>(a.bias + cw + n * 16);
        v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
      }
      if (resg) {
        const float4 rv = Q4<E>::unpack(rres[n]);
        v[0] += rv.x; v[1] += rv.y; v[2] += rv.z; v[3] += rv.w;
      }
      if (valid && !(a.flags & 256)) store4(yg + base + n * 16, make_float4(v[0], v[1], v[2], v[3]));
      if (a.stats) {
        float s0[4], s1[4];
        if constexpr (BN) {
          const float4 ga = *reinterpret_cast<const float4*>(a.bn_coef + cw + n * 16);
          const float4 be = *reinterpret_cast<const float4*>(a.bn_coef + a.Cout_p + cw + n * 16);
          const float4 mu = *reinterpret_cast<const float4*>(a.bn_coef + 2 * a.Cout_p + cw + n * 16);
          const float4 rs = *reinterpret_cast<const float4*>(a.bn_coef + 3 * a.Cout_p + cw + n * 16);
          const float4 xv = Q4<E>::unpack(rbnx[n]);
          const float gav[4] = {ga.x, ga.y, ga.z, ga.w}, bev[4] = {be.x, be.y, be.z, be.w};
          const float muv[4] = {mu.x, mu.y, mu.z, mu.w}, rsv[4] = {rs.x, rs.y, rs.z, rs.w};
          const float xvv[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float xh = (xvv[r] - muv[r]) * rsv[r];
            const float dg = Vec16<E>::round(v[r]) * gelu_grad_f<E>(gav[r] * xh + bev[r]);
            s0[r] = valid ? dg : 0.f;
            s1[r] = valid ? dg * xh : 0.f;
          }
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float q = Vec16<E>::round(v[r]);
            s0[r] = valid ? q : 0.f;
            s1[r] = valid ? q * q : 0.f;
          }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) { s0[r] = row16_sum(s0[r]); s1[r] = row16_sum(s1[r]); }
        if (lr == 0) {                                   // this lane's 4 channels of tile n: running sums over the slices
          float4* r0 = reinterpret_cast<float4*>(red + n * 16 + lq * 4);
          float4* r1 = reinterpret_cast<float4*>(red + P_HALF + n * 16 + lq * 4);
          float4 o0 = make_float4(s0[0], s0[1], s0[2], s0[3]), o1 = make_float4(s1[0], s1[1], s1[2], s1[3]);
          if (m > 0) {
            const float4 p0 = *r0, p1 = *r1;
            o0.x += p0.x; o0.y += p0.y; o0.z += p0.z; o0.w += p0.w;
            o1.x += p1.x; o1.y += p1.y; o1.z += p1.z; o1.w += p1.w;
          }
          *r0 = o0; *r1 = o1;
        }
      }
    }
  };
  // the two waves that share a tile's channels (wave_t = 0 / 1) combine their sums in a fixed order; called
  // one workgroup barrier after slice 3
  auto epi_stats_store = [&]() __attribute__((always_inline)) {
    if (!a.stats || wave_t != 0 || !P.ok) return;
    const float* mine = red;
    const float* other = red + 2 * 2 * P_HALF;           // wave wid + 2: same tile, same channels, rows 64..127
    for (int i = lane; i < 2 * P_HALF; i += 64) {
      const int which = i / P_HALF, c = i - which * P_HALF;
      a.stats[((size_t)P.tt * 2 + which) * a.Cout_p + P.co0 + wave_c * P_HALF + c] = mine[i] + other[i];
    }
  };
  auto epi_step = [&](int s) __attribute__((always_inline)) {
    switch (s) {
      case 0: epi_math(std::integral_constant<int, 0>{}); break;
      case 1: epi_math(std::integral_constant<int, 1>{}); break;
      case 2: epi_math(std::integral_constant<int, 2>{}); break;
      case 3: epi_math(std::integral_constant<int, 3>{}); break;
      default: break;
    }
  };

  // ---- the slab stream
  Unit Lu = unit_of(0);                                  // unit the DMA front is in
  int l_unit = 0, l_s = 0;
  auto advance_front = [&]() __attribute__((always_inline)) {
    if (++l_s == nslab) { l_s = 0; ++l_unit; if (l_unit < n_my) Lu = unit_of(l_unit); }
  };
  auto issue_next = [&](int buf) __attribute__((always_inline)) {                       // stage the slab at the DMA front and advance it
    stage_x(Lu, l_s, buf);
    stage_w(Lu, l_s, buf, 0, 4);
    advance_front();
  };
  issue_next(0);
  if (total_slabs > 1) issue_next(1);

  int g = 0, cur = 0;
  // One K iteration.  PH = 0..3: slice PH of the pending tile rides along (its loads go out first, are
  // consumed after two taps of MFMAs, and only then is the next slab's DMA issued, so the compiler's own waits
  // for those loads never include DMA pieces younger than them); PH = 4: the pending tile's statistics are
  // combined and stored; PH = 5: nothing pending.  The first five iterations of a tile are PEELED with
  // PH = 0..4 so that the pending accumulators' live range ends slice by slice; the remaining iterations run
  // in a loop where only the active accumulators are live.
  auto iteration = [&](auto PHC) __attribute__((always_inline)) {
    constexpr int PH = decltype(PHC)::value;
    if (g + 1 < total_slabs) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    else                     asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                       // slab g landed for everyone; slab g-1 fully consumed
    const bool more = g + 2 < total_slabs;
    const int nxt = cur >= 1 ? cur - 1 : 2;             // (cur + 2) % 3
    const unsigned char* xs = smem + cur * P_STAGE + tsel * XS_BYTES;
    const unsigned char* ws = smem + cur * P_STAGE + 2 * XS_BYTES;
    if constexpr (PH == 5) {
      if (more) { stage_x(Lu, l_s, nxt); stage_w(Lu, l_s, nxt, 0, 1); }
      compute_tap(xs, ws, 0);
      if (more) stage_w(Lu, l_s, nxt, 1, 2);
      compute_tap(xs, ws, 1);
      if (more) { stage_w(Lu, l_s, nxt, 2, 4); advance_front(); }
      compute_tap(xs, ws, 2);
    } else {
      if constexpr (PH < 4) { if (pend_valid) epi_loads(PH); }
      compute_tap(xs, ws, 0);
      compute_tap(xs, ws, 1);
      if (pend_valid) {
        if constexpr (PH < 4) epi_math(std::integral_constant<int, PH>{});
        else epi_stats_store();
      }
      if (more) issue_next(nxt);
      compute_tap(xs, ws, 2);
    }
    cur = cur == 2 ? 0 : cur + 1;
    ++g;
  };
  for (int k = 0; k < n_my; ++k) {
    const Unit Cu = unit_of(k);
    iteration(std::integral_constant<int, 0>{});
    iteration(std::integral_constant<int, 1>{});
    iteration(std::integral_constant<int, 2>{});
    iteration(std::integral_constant<int, 3>{});
    iteration(std::integral_constant<int, 4>{});
    for (int s = 5; s < nslab; ++s) iteration(std::integral_constant<int, 5>{});
    // the finished tile becomes the pending one
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int n = 0; n < P_NREP; ++n) { pend[m][n] = acc[m][n]; acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    P = Cu;
    pend_valid = true;
  }
  // last tile: nothing left to hide behind
  for (int m = 0; m < 4; ++m) { epi_loads(m); epi_step(m); }
  __syncthreads();
  epi_stats_store();
}

template <typename E, bool BN>
int launch_persist(const sda_conv_args& a, hipStream_t st) {
  static bool attr_done = false;
  static int n_cu = 0;
  auto kern = conv3_persist_kernel<E, BN>;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, P_LDS) !=
        hipSuccess) {
      set_error("conv3_persist: cannot reserve %d bytes of LDS", P_LDS);
      return -3;
    }
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
      set_error("conv3_persist: cannot query the device");
      return -3;
    }
    n_cu = prop.multiProcessorCount;
    attr_done = true;
  }
  const int n_t = (a.T + TILE_T - 1) / TILE_T;
  const int tiles_total = a.B * n_t;
  const int n_tpairs = (tiles_total + 1) / 2;
  const int n_co = a.Cout_p / P_CO;
  const int units_per_xcd = ((n_tpairs + 7) / 8) * n_co;
  int wg_per_xcd = n_cu / 8;
  if (wg_per_xcd < 1) wg_per_xcd = 1;
  if (wg_per_xcd > units_per_xcd) wg_per_xcd = units_per_xcd;
  hipLaunchKernelGGL(kern, dim3(8 * wg_per_xcd), dim3(512), P_LDS, st, a, n_t, n_tpairs, units_per_xcd, wg_per_xcd);
  return check_launch("conv3_persist");
}

}  // namespace

bool conv3_persist_supports(const sda_conv_args& a) {
  const int slab = ROW_B / (a.dtype == SDA_F32 ? 4 : 2);
  if (a.dtype != SDA_F32 && a.dtype != SDA_BF16) return false;     // fp16 runs on the tile-per-workgroup kernel
  return a.KS == 3 && a.Cout_p % P_CO == 0 && !a.widx && a.ksplit == 1 && !a.partial && !(a.flags & SDA_EPI_GELU) &&
         !a.y_pre && a.y && a.Cin_p / slab >= 5 && (!a.bn_x || (a.bn_coef && a.stats));
}

int launch_conv3_persist(const sda_conv_args& a, hipStream_t st) {
  if (a.dtype == SDA_F32) return a.bn_x ? launch_persist<float, true>(a, st) : launch_persist<float, false>(a, st);
  return a.bn_x ? launch_persist<uint16_t, true>(a, st) : launch_persist<uint16_t, false>(a, st);
}

}  // namespace sda

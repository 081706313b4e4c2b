// conv1_flat — the kernel-size-1 Conv1d (a per-row projection) on 256/128-row x 160- or 128-channel tiles of the FLAT row
// space, persistent workgroups, two per CU: conv3_flat.hip's tiling for the 1 x 1 convs.
//
// Reference ops: conv_final1 / conv_final2 = nn.Conv1d(kernel_size=1) + GELU (models.py:194-195) and their input gradients;
// same sda_conv_args contract as conv_gemm (bias, y_pre, GELU), plus two epilogues of its own (SDA_EPI_GELU_BWD,
// SDA_EPI_ROW_SUMSQ, include/sd_amd.h).
//
// Why a third kernel.  conv_gemm's 128 x 128 tile runs 16 MFMAs per wave between barriers and spends 4.3 scalar + 4.4 vector
// instructions per MFMA on them (profiles/r04_step_sq_counters.txt: matrix-pipe utilisation 0.25); more LDS stages in flight
// did not move it and neither did two tiles per workgroup: the K loop is bound by what a wave issues per
// MFMA, not by latency.  Here a wave owns 128 rows x 80 (64) channels — 40 (32) MFMAs per barrier, one LDS fragment read per
// 5 (4) MFMAs — and every LDS-DMA piece is one scalar add + one instruction (scalar bases fixed per tile, one per-lane
// offset register shared by all pieces, conv3_flat's form):
//   * stage = input slab (256 rows x 64 B) + weight slab (160 / 128 rows x 64 B) = 26 / 24 KB, a ring of THREE (78 / 72 KB,
//     two workgroups per CU): during K-step s every wave issues its share of slab s + 2 between its MFMA rows, and ONE
//     counted `s_waitcnt vmcnt(N)` at the top of K-step s + 1 retires slab s + 1 while those stay in flight across the raw
//     s_barrier.  The stage slab s + 2 goes into is the one slab s - 1 was read from: every wave has passed this K-step's
//     barrier, i.e. has consumed it;
//   * tiles, persistent runs, the two co-resident workgroups' opposite tile order and priority hand-over: as conv3_flat.
#include "flat_tile.h"

namespace sda {

namespace {

constexpr int G_UNIT = 128;                          // rows per work unit
constexpr int G_EP_ROWS = 64;

template <int NREP> struct G1 {                      // geometry for NREP 16-channel fragments per wave
  static constexpr int CO = 32 * NREP;               // output channels per workgroup (160 / 128)
  static constexpr int WP = CO / 16;                 // 1 KB weight pieces per slab
  static constexpr int XB = 256 * ROW_B;             // 16 KB: input slab of a 256-row tile
  static constexpr int STAGE = XB + CO * ROW_B;      // 26 / 24 KB
  static constexpr int NS = 3;
  static constexpr int LDS = NS * STAGE;             // 78 / 72 KB
  static constexpr int STRIDE = CO + 4;              // floats; == 4 (mod 8): conflict-free accumulator writes
  static constexpr int EP_BYTES = G_EP_ROWS * STRIDE * 4;
  static_assert(2 * LDS <= 160 * 1024, "two workgroups per CU");
};
template <int NREP, int CH> struct G1Epi {
  static constexpr int NCH = G1<NREP>::CO / CH;      // CH-channel chunks (16 bytes of E) per row
  static constexpr int RG = 256 / NCH;               // row groups
  static constexpr int ITERS = (G_EP_ROWS + RG - 1) / RG;
  static constexpr int RED_BYTES = RG * G1<NREP>::CO * 4;
  static_assert(G1<NREP>::EP_BYTES + RED_BYTES <= G1<NREP>::LDS, "epilogue staging must fit");
};

struct G1Ctx {               // per-workgroup constants shared by all its tiles
  int tid, lane, wid, co0, co_tile, n_co, nslab, Tp;
  long total_rows;
};

// One tile: R = 32 * MREP output rows starting at flat (= buffer) row f0.
// GB (SDA_EPI_GELU_BWD): y = round(conv) * GELU'(u), u = a.bn_x (what gelu_backward_colsum does in a pass of its own from the
//   gradient as STORED: here it never goes to memory); stats row `stat_row`, plane 0 = column sums of the products (the bias
//   gradient of the layer below), plane 1 = 0.
// RSQ (SDA_EPI_ROW_SUMSQ): stats [row][n_co] = sum over this tile's 128 channels of the squares of row's values as stored.
template <typename E, int NREP, int MREP, bool GB, bool RSQ>
__device__ __forceinline__ void flat1_tile(const sda_conv_args& a, unsigned char* smem, const G1Ctx& c, const long f0, const int stat_row) {
  using P = G1<NREP>;
  constexpr int PER16 = Elem<E>::PER16;
  constexpr int CH = Vec16<E>::N;
  using G = G1Epi<NREP, CH>;
  constexpr int R = 32 * MREP;
  constexpr int XP = R / 16;                         // input pieces per slab
  constexpr int TOT = XP + P::WP;
  constexpr int NPW = (TOT + 3) / 4;                 // pieces per wave per slab (indices past the end are clamped: duplicates rewrite identical bytes)
  constexpr int PER = (NPW + MREP - 1) / MREP;       // pieces behind each MFMA row
  static_assert(!RSQ || G::NCH == 16 || G::NCH == 32, "row sums reduce over the 16 (fp32: 32) consecutive lanes of a row");
  const int co0 = c.co0;
  const E* __restrict__ xg = reinterpret_cast<const E*>(a.x);
  const E* __restrict__ wg = reinterpret_cast<const E*>(a.w);

  f32x4 acc[MREP][NREP];
#pragma unroll
  for (int m = 0; m < MREP; ++m)
#pragma unroll
    for (int n = 0; n < NREP; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

  {   // ==== K loop (a scope of its own, per-thread values re-derived from a laundered lane index: conv3_flat.hip explains)
  int lane_k = c.lane;
  asm volatile("" : "+v"(lane_k));
  const int wid = c.wid, wave_m = wid >> 1, wave_n = wid & 1, lr = lane_k & 15, lq = lane_k >> 4;
  const int k_prow = lane_k >> 2, k_pchunk = lane_k & 3;
  const uint32_t voff0 = (uint32_t)(((size_t)k_prow * a.x_pitch + (size_t)((k_pchunk ^ sw64(k_prow)) * PER16)) * sizeof(E));
  const uint32_t lds_base = __builtin_amdgcn_readfirstlane(lds_addr(smem));
  // this wave's pieces of a slab: j = wid + 4 i; j < XP: input rows [16 j, 16 j + 16) of the tile, else weight rows 16 (j - XP)
  const char* pbase[NPW];
  uint32_t pdst[NPW];
  static_for<0, NPW>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    int j = wid + 4 * i;
    j = j < TOT ? j : TOT - 1;
    if (j < XP) {
      long srow = f0 + j * 16;
      srow = srow > a.x_rows_limit - 16 ? a.x_rows_limit - 16 : srow;       // (rows past the last sample: never stored)
      pbase[i] = reinterpret_cast<const char*>(xg + (size_t)srow * a.x_pitch);
      pdst[i] = (uint32_t)j * 1024u;
    } else {
      pbase[i] = reinterpret_cast<const char*>(wg + ((size_t)co0 + (j - XP) * 16) * a.w_pitch);
      pdst[i] = (uint32_t)P::XB + (uint32_t)(j - XP) * 1024u;
    }
  });
  uint32_t kvoff = voff0;                            // per-lane offset incl. the channel offset of the slab being issued
  const int nslab = (a.flags & 512) ? 1 : c.nslab;   // flag 512 (diagnostic): one K-step only — the epilogue's time (results are garbage)
  // ---- prologue: slabs 0 and 1 into stages 0 and 1
  static_for<0, NPW>([&](auto ic) { lds_dma16_lean<true>(pbase[decltype(ic)::value], kvoff, lds_base + pdst[decltype(ic)::value]); });
  if (nslab > 1) {
    kvoff += ROW_B;
    static_for<0, NPW>([&](auto ic) { lds_dma16_lean<true>(pbase[decltype(ic)::value], kvoff, lds_base + (uint32_t)P::STAGE + pdst[decltype(ic)::value]); });
  }
  const int wrow = wave_n * (P::CO / 2) + lr;
  const int xrow0 = wave_m * (R / 2) + lr;
  const bool no_dma = a.flags & 128;                 // (diagnostic: no LDS-DMA inside the K loop — garbage results)
  uint32_t rd = 0;                                   // byte offset of the stage K-step s reads
  // One K-step.  MODE 0: issues slab s + 2; 1: the last but one (nothing left to issue); 2: the last (waits for everything).
  auto kstep = [&](auto modec) {
    constexpr int MODE = decltype(modec)::value;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    wait_vmcnt_lit<MODE == 2 ? 0 : NPW>();           // this wave's pieces of slab s have landed (those of slab s + 1 may fly on) ...
    __builtin_amdgcn_s_barrier();                    // ... everybody's have, and slab s - 1 is fully consumed
    const unsigned char* xs = smem + rd;
    const unsigned char* ws = xs + P::XB;
    uint4 bf[NREP];
#pragma unroll
    for (int n = 0; n < NREP; ++n) bf[n] = *reinterpret_cast<const uint4*>(ws + lds_sw64(wrow + n * 16, lq));
    uint4 af = *reinterpret_cast<const uint4*>(xs + lds_sw64(xrow0, lq));
    const uint32_t wr = lds_base + (rd == 0 ? 2u * P::STAGE : rd - (uint32_t)P::STAGE);   // stage (s + 2) % 3 == (s - 1) % 3
    if constexpr (MODE == 0) kvoff += ROW_B;
    static_for<0, MREP>([&](auto mc) {
      constexpr int m = decltype(mc)::value;
      uint4 af_next = af;
      if constexpr (m + 1 < MREP) af_next = *reinterpret_cast<const uint4*>(xs + lds_sw64(xrow0 + (m + 1) * 16, lq));
      mma16_row<E, NREP>(af, bf, acc[m]);
      // the DMA pieces go BETWEEN the MFMA rows: the wave's issue slot is free while the matrix pipe works through the MFMAs queued before it
      if constexpr (MODE == 0) {
        if (!no_dma) {
          static_for<m * PER, ((m + 1) * PER < NPW ? (m + 1) * PER : NPW)>([&](auto ic) {
            lds_dma16_lean<false>(pbase[decltype(ic)::value], kvoff, wr + pdst[decltype(ic)::value]);
          });
        }
      }
      af = af_next;
    });
    rd = rd == 2u * P::STAGE ? 0u : rd + (uint32_t)P::STAGE;
  };
  // The K loop runs at raised priority: a SIMD issues MFMA and ordinary vector instructions through one port, oldest wave
  // first — beside an OLDER workgroup's epilogue (a dense stream of vector instructions, GELU) a younger one's MFMAs would
  // only get the slots that stream leaves; an MFMA takes the port for one pass in four, so the epilogue loses little.
  if (!(a.flags & 2048)) __builtin_amdgcn_s_setprio(2);
  {
    int s = 0;
    for (; s + 2 < nslab; ++s) kstep(std::integral_constant<int, 0>{});
    if (s + 1 < nslab) kstep(std::integral_constant<int, 1>{});
    kstep(std::integral_constant<int, 2>{});
  }
  __builtin_amdgcn_s_setprio(0);
  }   // ==== end of the K loop's scope
  // ------------------------------------------------------------------ epilogue
  int tid_e = c.tid;
  asm volatile("" : "+v"(tid_e));
  const int tid = tid_e, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6), wave_m = wid >> 1, wave_n = wid & 1;
  const int lr = lane & 15, lq = lane >> 4;
  if (a.flags & 256) {        // diagnostic: skip the epilogue, keep the accumulators live
    float keep = 0.f;
#pragma unroll
    for (int m = 0; m < MREP; ++m)
#pragma unroll
      for (int n = 0; n < NREP; ++n) keep += acc[m][n][0] + acc[m][n][1] + acc[m][n][2] + acc[m][n][3];
    if (keep == 123.456f) reinterpret_cast<float*>(a.y)[0] = keep;
    __syncthreads();
    return;
  }
  if (a.bias) {
#pragma unroll
    for (int n = 0; n < NREP; ++n) {
      const float bv = a.bias[co0 + wave_n * (P::CO / 2) + n * 16 + lr];
#pragma unroll
      for (int m = 0; m < MREP; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[m][n][r] += bv;
    }
  }
  float* ep = reinterpret_cast<float*>(smem);
  float* red = reinterpret_cast<float*>(smem + P::EP_BYTES);
  const int chunk = tid % G::NCH, rg = tid / G::NCH;
  const bool active = rg < G::RG;
  E* __restrict__ yg = reinterpret_cast<E*>(a.y);
  E* __restrict__ ypre = reinterpret_cast<E*>(a.y_pre);
  const E* __restrict__ ug = reinterpret_cast<const E*>(a.bn_x);
  const bool gelu = (a.flags & SDA_EPI_GELU) && !(a.flags & 16);    // flags 16 / 8 (diagnostic): no GELU arithmetic / no stores
  const bool no_store = a.flags & 8;
  float ssum[GB ? CH : 1];
#pragma unroll
  for (int j = 0; j < (GB ? CH : 1); ++j) ssum[j] = 0.f;
  // position of this thread's first row inside its sample (rows p < PAD are the sample's zero padding: not stored)
  const int Tp = c.Tp;
  int pq = (int)(((unsigned)f0 + (unsigned)rg) % (unsigned)Tp);     // (flat rows fit 32 bits: supports() checks)

#pragma unroll
  for (int q = 0; q < R / G_EP_ROWS; ++q) {         // 64-row slices of the tile
    constexpr int WROWS = R / 2;                    // rows per wave
    const int owner = (q * G_EP_ROWS) / WROWS;      // the wave_m whose accumulators hold this slice
    const int m0 = (q * G_EP_ROWS - owner * WROWS) / 16;
    __syncthreads();                                // main-loop LDS reads (q == 0) / the previous slice's reads are done
    if (wave_m == owner) {
#pragma unroll
      for (int mm = 0; mm < 4; ++mm)
#pragma unroll
        for (int n = 0; n < NREP; ++n)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            ep[(mm * 16 + lq * 4 + r) * P::STRIDE + wave_n * (P::CO / 2) + n * 16 + lr] = acc[m0 + mm][n][r];
    }
    __syncthreads();
    // this thread's rows of the slice (the pre-activation rows of SDA_EPI_GELU_BWD are fetched one row ahead, packed; the
    // accumulators of the later slices are still live here: no register may spill)
    const long row_mask = (a.flags & 32) ? 1023L : ~0L;   // (diagnostic: every store lands in the first 1024 rows — no HBM write traffic)
    auto row_off = [&](int it) { return (size_t)((f0 + q * G_EP_ROWS + rg + it * G::RG) & row_mask) * a.Cout_p + co0 + chunk * CH; };
    int p = pq;
    auto row_ok = [&](int it, int pp) {
      const int row = rg + it * G::RG;
      return active && row < G_EP_ROWS && pp >= PAD && f0 + q * G_EP_ROWS + row < c.total_rows;
    };
    bool ok_cur = row_ok(0, p);
    uint4 u_cur = make_uint4(0u, 0u, 0u, 0u);
    if constexpr (GB) { if (ok_cur) u_cur = Vec16<E>::load_raw(ug + row_off(0)); }
#pragma unroll
    for (int it = 0; it < G::ITERS; ++it) {
      p += G::RG;
      if (p >= Tp) p -= Tp;
      bool ok_nxt = false;
      uint4 u_nxt = make_uint4(0u, 0u, 0u, 0u);
      if (it + 1 < G::ITERS) {
        ok_nxt = row_ok(it + 1, p);
        if constexpr (GB) { if (ok_nxt) u_nxt = Vec16<E>::load_raw(ug + row_off(it + 1)); }
      }
      float sq = 0.f;
      if (ok_cur) {
        const int row = rg + it * G::RG;
        float v[CH];
#pragma unroll
        for (int q4 = 0; q4 < CH / 4; ++q4) {
          const float4 f = *reinterpret_cast<const float4*>(ep + row * P::STRIDE + chunk * CH + q4 * 4);
          v[q4 * 4 + 0] = f.x; v[q4 * 4 + 1] = f.y; v[q4 * 4 + 2] = f.z; v[q4 * 4 + 3] = f.w;
        }
        if constexpr (GB) {
          float u8[CH];
          Vec16<E>::unpack(u_cur, u8);
#pragma unroll
          for (int j = 0; j < CH; j += 2) {          // same expression as bwd_colsum_kernel<E, 0> on the gradient as it would have been stored
            const f32x2 o = f32x2{Vec16<E>::round(v[j]), Vec16<E>::round(v[j + 1])} * gelu_grad_pair<E>(f32x2{u8[j], u8[j + 1]});
            v[j] = o.x; v[j + 1] = o.y;
            ssum[j] += o.x; ssum[j + 1] += o.y;
          }
        } else if (gelu) {
          if (ypre && !no_store) Vec16<E>::store(ypre + row_off(it), v);
#pragma unroll
          for (int j = 0; j < CH; j += 2) {
            const f32x2 gp = gelu_pair<E>(f32x2{v[j], v[j + 1]});
            v[j] = gp.x; v[j + 1] = gp.y;
          }
        }
        if (!no_store) Vec16<E>::store(yg + row_off(it), v);
        else if (v[0] == 123.456f) Vec16<E>::store(yg + row_off(it), v);
        if constexpr (RSQ) {
#pragma unroll
          for (int j = 0; j < CH; ++j) { const float qv = Vec16<E>::round(v[j]); sq = fmaf(qv, qv, sq); }
        }
      }
      if constexpr (RSQ) {
        // the 16 (fp32: 32) threads of a row are consecutive lanes: butterfly over them (every lane takes part, rows that are
        // not stored carry zeros), fixed order
#pragma unroll
        for (int o = 1; o < G::NCH; o <<= 1) sq += __shfl_xor(sq, o);
        if (ok_cur && chunk == 0) a.stats[(size_t)(f0 + q * G_EP_ROWS + rg + it * G::RG) * c.n_co + c.co_tile] = sq;
      }
      ok_cur = ok_nxt; u_cur = u_nxt;
    }
    pq += G_EP_ROWS;
    while (pq >= Tp) pq -= Tp;
  }
  if constexpr (GB) {
    __syncthreads();
    if (active) {
#pragma unroll
      for (int j = 0; j < CH; ++j) red[rg * P::CO + chunk * CH + j] = ssum[j];
    }
    __syncthreads();
    if (tid < P::CO) {
      float s0 = 0.f;
      for (int g = 0; g < G::RG; ++g) s0 += red[g * P::CO + tid];
      a.stats[((size_t)stat_row * 2 + 0) * a.Cout_p + co0 + tid] = s0;
      a.stats[((size_t)stat_row * 2 + 1) * a.Cout_p + co0 + tid] = 0.f;
      if (R > G_UNIT) {
        a.stats[((size_t)(stat_row + 1) * 2 + 0) * a.Cout_p + co0 + tid] = 0.f;
        a.stats[((size_t)(stat_row + 1) * 2 + 1) * a.Cout_p + co0 + tid] = 0.f;
      }
    }
  }
  __syncthreads();            // the next tile's LDS-DMA overwrites the staging / reduction area
}

template <typename E, int NREP, bool GB, bool RSQ>
__global__ __launch_bounds__(256, 2) void conv1_flat_kernel(const sda_conv_args a, const int n_units, const int units_per_wg,
                                                            const long total_rows, const int Tp) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  using P = G1<NREP>;
  G1Ctx c;
  c.tid = threadIdx.x;
  c.lane = c.tid & 63;
  c.wid = __builtin_amdgcn_readfirstlane(c.tid >> 6);
  c.nslab = a.Cin_p / (ROW_B / (int)sizeof(E));
  c.Tp = Tp;
  c.total_rows = total_rows;
  // XCD-aware order (blocks i and i + 8 share an XCD/L2): the n_co workgroups that read the same input rows are dealt to the
  // same XCD.  Pure speed: any placement is correct.
  int bid = blockIdx.x, co_tile, run;
  const int n_co = a.Cout_p / P::CO;
  {
    const int group = 8 * n_co, full = (int)(gridDim.x / group) * group;
    if (bid < full) {
      const int base = bid / group * group, rem = bid - base;
      co_tile = rem / 8;
      run = base / n_co + (rem & 7);
    } else {
      const int rem = bid - full;
      co_tile = rem % n_co;
      run = full / n_co + rem / n_co;
    }
  }
  c.co_tile = co_tile;
  c.n_co = n_co;
  c.co0 = co_tile * P::CO;
  int u = run * units_per_wg;                               // this workgroup's 128-row units: [u, u_end)
  const int u_end = min(n_units, u + units_per_wg);
  // the two workgroups of a CU (observed: blocks i and i + 8 * 32 of an XCD) may take their tiles in opposite order, so that
  // their epilogues fall at different times (conv3_flat.hip; flag 1024)
  const bool second = ((blockIdx.x >> 3) >> 5) & 1;
  const bool small_first = second && (a.flags & 1024);     // (measured alone: the same order is 3-8 % faster here; flag 1024 staggers)
  const int n = u_end - u;
  if (n <= 0) return;
  int lead, tail, pairs;
  if (n & 1) { pairs = n >> 1; lead = small_first ? 1 : 0; tail = 1 - lead; }
  else if (small_first && n >= 2) { pairs = (n >> 1) - 1; lead = 1; tail = 1; }
  else { pairs = n >> 1; lead = 0; tail = 0; }
  if (lead) { flat1_tile<E, NREP, 4, GB, RSQ>(a, smem, c, (long)u * G_UNIT, u); ++u; }
  for (int p = 0; p < pairs; ++p, u += 2) flat1_tile<E, NREP, 8, GB, RSQ>(a, smem, c, (long)u * G_UNIT, u);
  if (tail) flat1_tile<E, NREP, 4, GB, RSQ>(a, smem, c, (long)u * G_UNIT, u);
}

template <typename E, int NREP, bool GB, bool RSQ>
int launch_flat1(const sda_conv_args& a, hipStream_t st) {
  using P = G1<NREP>;
  static unsigned long long attr_done = 0;        // per device
  auto kern = conv1_flat_kernel<E, NREP, GB, RSQ>;
  if (first_use_on_device(attr_done)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, P::LDS) != hipSuccess) {
      set_error("conv1_flat: cannot reserve %d bytes of LDS", P::LDS);
      return -3;
    }
  }
  const long total_rows = (long)a.B * rows_tp(a.T);
  const int n_units = (int)((total_rows + G_UNIT - 1) / G_UNIT);
  const int n_co = a.Cout_p / P::CO;
  const long slots = ((a.flags & SDA_CONV_ONE_PER_CU) ? 1L : 2L) * launch_cus();
  int units_per_wg = (int)(((long)n_units * n_co + slots - 1) / slots);
  if (units_per_wg < 1) units_per_wg = 1;
  const int runs_per_co = (n_units + units_per_wg - 1) / units_per_wg;
  hipLaunchKernelGGL(kern, dim3((unsigned)(runs_per_co * n_co)), dim3(256), P::LDS, st, a, n_units, units_per_wg, total_rows, rows_tp(a.T));
  return check_launch("conv1_flat");
}

template <typename E> int launch_flat1_e(const sda_conv_args& a, hipStream_t st) {
  if (a.flags & SDA_EPI_ROW_SUMSQ) return launch_flat1<E, 4, false, true>(a, st);
  if (a.flags & SDA_EPI_GELU_BWD) return a.Cout_p % 160 == 0 ? launch_flat1<E, 5, true, false>(a, st) : launch_flat1<E, 4, true, false>(a, st);
  return a.Cout_p % 160 == 0 ? launch_flat1<E, 5, false, false>(a, st) : launch_flat1<E, 4, false, false>(a, st);
}

}  // namespace

bool conv1_flat_supports(const sda_conv_args& a) {
  const bool gb = a.flags & SDA_EPI_GELU_BWD, rsq = a.flags & SDA_EPI_ROW_SUMSQ;
  if (gb && (rsq || !a.bn_x || !a.stats || a.bias || a.y_pre || (a.flags & SDA_EPI_GELU))) return false;
  if (rsq && (!a.stats || a.Cout_p % 128)) return false;
  if (!gb && !rsq && (a.stats || a.bn_x)) return false;
  return a.KS == 1 && (a.Cout_p % 160 == 0 || a.Cout_p % 128 == 0) && !a.widx && !a.res && a.ksplit == 1 && !a.partial && a.y &&
         !(a.flags & (SDA_EPI_GLU | SDA_EPI_GLU_BWD)) && a.x_row0 == PAD && a.x_pitch == a.w_pitch && a.x_pitch == a.Cin_p &&
         a.x_sample_rows == rows_tp(a.T) && a.x_rows_limit >= (long)a.B * rows_tp(a.T) + 3 * PAD && a.x_rows_limit < (1L << 31) &&
         a.w_rows_limit >= a.Cout_p && (long)a.x_pitch * 16 * (a.dtype == SDA_F32 ? 4 : 2) < (1L << 31);
}

int launch_conv1_flat(const sda_conv_args& a, hipStream_t st) {
  if (a.dtype == SDA_F32) return launch_flat1_e<float>(a, st);
  if (a.dtype == SDA_F16) return launch_flat1_e<half_t>(a, st);
  return launch_flat1_e<uint16_t>(a, st);
}

}  // namespace sda

// conv_gemm — implicit-GEMM dilated Conv1d (kernel 1 or 3) on MFMA, channels-last rows.
//
// Reference ops this replaces: every nn.Conv1d on the path — models.py:97-109 (1x1 + per-subject 1x1),
// models.py:128-150 (k=3 dilated), models.py:188-189 (final 1x1) — their input gradients (same kernel
// on mode-1 packed weights), SpatialAttention's channel mix (models.py:65) and, with ksplit > 1, the
// similarity matmul of loss.py:68.
//
// One workgroup (256 threads = 4 waves as 2(t) x 2(co)) computes a 128-row x TILE_CO-channel output
// tile of one sample.  K loop over 128-byte slabs of input channels: the (128 + 2*dil)-row input slab
// and the KS x TILE_CO weight slab are staged into XOR-swizzled LDS; every MFMA operand is a 16-byte
// ds_read_b128 (8 bf16 / 4 fp32 along the contraction).  The three taps reuse the same input slab
// at row offsets tap*dil.  D[row = t][col = co] accumulates in fp32.
// Epilogue: bias in registers -> LDS staging -> coalesced pass adding the residual, optional GELU,
// BatchNorm partial statistics (per tile, deterministic), 8/16-byte stores.
#include "sd_common.h"
#include "flat_tile.h"

namespace sda {

constexpr int EP_ROWS = 64;

template <int TILE_CO, int CH = 4> struct EpiGeom {
  static constexpr int STRIDE = TILE_CO + 4;          // floats; == 4 (mod 8): conflict-free D writes
  static constexpr int NCH = TILE_CO / CH;            // CH-channel chunks (16 bytes of E) per row
  static constexpr int RG = 256 / NCH;                // row groups
  static constexpr int ITERS = (EP_ROWS + RG - 1) / RG;
  static constexpr int EP_BYTES = EP_ROWS * STRIDE * 4;
  static constexpr int RED_BYTES = RG * TILE_CO * 2 * 4;
};

template <int TILE_CO, int KS, int NT> constexpr int conv_stage_bytes() { return NT * XS_BYTES + KS * TILE_CO * ROW_B; }
// LDS stages of the K loop: 3 for the paired kernel-3 tiles, a RING of 4 (3 for 160-channel tiles: 60 KB, two workgroups per
// CU) for single-tile kernel-size-1 launches — a K-step there is 16 MFMAs per wave, far less than one LDS-DMA round trip, so
// two / three slabs stay in flight behind counted waits — in split-K matrix mode (SV = false), 2 otherwise.
// (measured, config-2 shapes alone: the similarity matmul 203 -> 114 us at 256 x 256 samples, 710 -> 631 us at 2048 x 256;
// the row-layout 1x1 convs, whose time is their epilogue's, gained nothing and lose co-residency: they keep two stages)
template <int TILE_CO, int NT, int KS, bool SV> constexpr int conv_stages() {
#ifdef SDA_RING_SV
  return (NT == 2 && KS == 3) ? 3 : ((NT == 1 && KS == 1) ? (TILE_CO > 128 ? 3 : 4) : 2);
#else
  return (NT == 2 && KS == 3) ? 3 : ((NT == 1 && KS == 1 && !SV) ? (TILE_CO > 128 ? 3 : 4) : 2);
#endif
}
template <int TILE_CO, int KS, int NT, bool SV = true> constexpr int conv_lds_bytes() {
  constexpr int main_b = conv_stages<TILE_CO, NT, KS, SV>() * conv_stage_bytes<TILE_CO, KS, NT>();
  constexpr int epi_b = NT * (EpiGeom<TILE_CO, 8>::EP_BYTES + EpiGeom<TILE_CO, 8>::RED_BYTES);   // CH = 8 is the larger
  return main_b > epi_b ? main_b : epi_b;
}

typedef __attribute__((address_space(1))) const void gmem_cv;
typedef __attribute__((address_space(3))) void lds_v;

template <typename T> __device__ inline float4 round_like(float4 v);
template <> __device__ inline float4 round_like<float>(float4 v) { return v; }
template <> __device__ inline float4 round_like<half_t>(float4 v) {
  return make_float4((float)(half_t)v.x, (float)(half_t)v.y, (float)(half_t)v.z, (float)(half_t)v.w);
}
template <> __device__ inline float4 round_like<uint16_t>(float4 v) {
  return make_float4(bf2f(f2bf(v.x)), bf2f(f2bf(v.y)), bf2f(f2bf(v.z)), bf2f(f2bf(v.w)));
}

// NT = number of 128-row output tiles one workgroup computes SIDE BY SIDE against the same weight slab
// (4 waves per tile).  The weight slab is 3/4 of the bytes a workgroup pulls through LDS per K-step, so
// NT = 2 cuts the L2->LDS traffic per FLOP by 37 %: the main loop is LDS-DMA bound, not MFMA bound.
// SV: row-layout operands with fully padded weights (every launch except the split-K "matrix mode"): each 1 KB LDS-DMA
// piece starts at a wave-uniform row, so its address is a scalar base + ONE per-lane offset register (lds_dma16_sv) and the
// per-piece arithmetic runs on the scalar unit — with per-lane 64-bit addresses the staging cost ~100 VALU instructions
// per K-step and wave, a third of the MFMA time beside it.
// BN: 0 = off, 1 = BatchNorm-backward statistics epilogue (bn_x), 2 = the same storing dg instead of dy (SDA_EPI_BN_STORE_DG;
// a template parameter, not a branch: with both forms in one kernel the code grew by 15 % and the step by 0.13 ms)
// (3 = SDA_EPI_GELU_BWD, kernel size 1: bn_x is the GELU's input)
template <typename E, int TILE_CO, int KS, int NT, int BN = 0, bool SV = true>
__global__ __launch_bounds__(256 * NT, 2) void conv_gemm_kernel(const sda_conv_args a, const int n_t_tiles) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int SLAB = ROW_B / (int)sizeof(E);          // input channels per LDS row / K-step
  constexpr int PER16 = Elem<E>::PER16;
  constexpr int NREP = TILE_CO / 32;
  constexpr int HALF_CO = TILE_CO / 2;
  constexpr int STAGE = conv_stage_bytes<TILE_CO, KS, NT>();
  constexpr int NW = 4 * NT;                            // waves per workgroup
  constexpr int TP = TILE_CO / 16;                      // 1 KB weight pieces per tap (16 rows x 64 B)
  constexpr int CH = Vec16<E>::N;                       // channels per thread in the epilogue: 16-byte global accesses
  using G = EpiGeom<TILE_CO, CH>;

  // SDA_CONV_WAVE_PRIO: this kernel's waves win the per-SIMD issue arbitration against the waves of a kernel co-resident from
  // another stream (the backward's data-gradient convs beside the weight-gradient GEMMs)
  if (a.flags & SDA_CONV_WAVE_PRIO) __builtin_amdgcn_s_setprio(3);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tsel = wid >> 2;                            // which of the NT tiles this wave works on
  const int wave_t = (wid >> 1) & 1, wave_c = wid & 1;
  const int lr = lane & 15, lq = lane >> 4;

  int bid = blockIdx.x;
  const int n_co = a.Cout_p / TILE_CO;
  // XCD-aware order (blocks i and i+8 share an XCD/L2): the n_co workgroups that read the same input rows
  // are dealt to the same XCD.  Pure speed: any placement is correct.
  int co_tile;
  {
    const int group = 8 * n_co, full = (int)(gridDim.x / group) * group;
    if (bid < full) {
      const int base = bid / group * group, rem = bid - base;
      co_tile = rem / 8;
      bid = base / n_co + (rem & 7);
    } else {
      co_tile = bid % n_co; bid /= n_co;
    }
  }
  const int tiles_total = a.B * n_t_tiles;
  const int groups_total = (tiles_total + NT - 1) / NT;
  const int ks = bid / groups_total;
  const int tt = (bid - ks * groups_total) * NT + tsel;   // linear (sample, t-tile) index of this wave's tile
  // (split-K matrix mode — the loss's similarity matmul — keeps this order: with ALL output tiles of a K slice dealt to one XCD
  // the column operand is fetched once instead of twice (604 -> ~400 MB) and the kernel alone is 7 % faster, 113 -> 106 us,
  // but IN THE STEP, where both operands have just been written, it was 14-18 us slower on every box tried — DESIGN.md §7)
  const bool tile_ok = tt < tiles_total;
  const int b = tile_ok ? tt / n_t_tiles : 0;
  const int t_tile = tile_ok ? tt - b * n_t_tiles : 0;
  const int co0 = co_tile * TILE_CO;
  const int t0 = t_tile * TILE_T;
  const int dil = a.dil;
  const int halo = (KS == 3) ? dil : 0;

  const E* __restrict__ xg = reinterpret_cast<const E*>(a.x);
  const int wsel = a.widx ? a.widx[b] : 0;              // per-sample weights are launched with NT == 1
  const E* __restrict__ wg = reinterpret_cast<const E*>(a.w) + (size_t)wsel * KS * a.Cout_p * a.w_pitch;
  const long row_base = a.x_row0 + (long)b * a.x_sample_rows + t0 - halo;   // LDS x row 0 of this wave's tile
  const int x_pieces = (TILE_T + 2 * halo + 15) >> 4;                       // 16-row pieces actually needed

  f32x4 acc[4][NREP];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n = 0; n < NREP; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nslab = a.Cin_p / SLAB;
  const int per_split = (nslab + a.ksplit - 1) / a.ksplit;
  const int s_begin = ks * per_split;
  const int s_end = (a.flags & 512) ? s_begin : min(nslab, s_begin + per_split);

  // LDS-DMA pieces.  A piece is one wave-instruction: 64 lanes x 16 B land lane-linearly in LDS (16 rows x
  // 64 B), so the swizzle goes on the SOURCE chunk.  Rows outside the operand (only possible in split-K
  // matrix mode) are clamped: they feed outputs that are never stored.  The 4 waves of a tile fetch that
  // tile's input rows; all NW waves share the weight pieces, issued tap by tap between MFMA groups.
  const int prow = lane >> 2;                                   // row within the piece
  const int pchunk = lane & 3;                                  // physical chunk written by this lane
  // SV: per-lane byte offsets inside a piece (the swizzle depends on bit 2 of the row: pieces start at multiples of 16
  // rows and TILE_CO is a multiple of 32, so it is the same for every piece)
  const uint32_t xvoff = (uint32_t)(((size_t)prow * a.x_pitch + (size_t)((pchunk ^ sw64(prow)) * PER16)) * sizeof(E));
  const uint32_t wvoff = (uint32_t)(((size_t)prow * a.w_pitch + (size_t)((pchunk ^ sw64(prow)) * PER16)) * sizeof(E));
  const uint32_t lds_base = __builtin_amdgcn_readfirstlane(lds_addr(smem));
  // (wave-uniform by construction; where hipcc cannot prove it — the two-tile form's tile select — this makes it provable,
  // elsewhere it folds away)
  auto uniform_u32 = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };
  auto dma_x = [&](int s, int p, uint32_t lds_off) {           // input piece p (16 rows) of K-step s
    if constexpr (SV) {
      long srow = row_base + p * 16;
      srow = srow < 0 ? 0 : (srow > a.x_rows_limit - 16 ? a.x_rows_limit - 16 : srow);   // (never taken in row layout: slack rows)
      // (lean form: the base is scalar arithmetic on kernel arguments and block indices, no VALU-written SGPR feeds the load —
      // tools/check_dma_hazard.py walks the listing)
      lds_dma16_lean<false>(xg + (size_t)srow * a.x_pitch + (size_t)s * SLAB, xvoff, uniform_u32(lds_base + lds_off));
    } else {
      const int r = p * 16 + prow;
      long row = row_base + r;
      row = row < 0 ? 0 : (row >= a.x_rows_limit ? a.x_rows_limit - 1 : row);
      lds_dma16(xg + (size_t)row * a.x_pitch + (size_t)s * SLAB + (pchunk ^ sw64(r)) * PER16, lds_base + lds_off);
    }
  };
  auto dma_w = [&](int s, int tap, int q, uint32_t lds_off) {  // weight piece q (16 output channels) of one tap
    if constexpr (SV) {
      lds_dma16_lean<false>(wg + ((size_t)tap * a.Cout_p + co0 + q * 16) * a.w_pitch + (size_t)s * SLAB, wvoff, uniform_u32(lds_base + lds_off));
    } else {
      int co = co0 + q * 16 + prow;
      co = co < a.w_rows_limit ? co : a.w_rows_limit - 1;
      lds_dma16(wg + ((size_t)tap * a.Cout_p + co) * a.w_pitch + (size_t)s * SLAB + (pchunk ^ sw64(q * 16 + prow)) * PER16,
                lds_base + lds_off);
    }
  };
  auto stage_x = [&](int s, int buf) {
    if (tile_ok) {
      for (int p = wid & 3; p < x_pieces; p += 4) dma_x(s, p, buf * STAGE + tsel * XS_BYTES + p * 1024);
    }
  };
  auto stage_w = [&](int s, int buf, int tap) {          // the TILE_CO rows of one tap
#pragma unroll
    for (int i = 0; i < (TP + NW - 1) / NW; ++i) {
      const int q = wid + i * NW;
      if (q < TP) dma_w(s, tap, q, buf * STAGE + NT * XS_BYTES + (tap * TP + q) * 1024);
    }
  };

  constexpr int NS = conv_stages<TILE_CO, NT, KS, SV>();
  auto compute_tap = [&](const unsigned char* xs, const unsigned char* ws, int tap, auto&& between) {
    uint4 af[4], bf[NREP];
    const int xrow = wave_t * 64 + lr + tap * dil;
    const int wrow = tap * TILE_CO + wave_c * HALF_CO + lr;
#pragma unroll
    for (int m = 0; m < 4; ++m) af[m] = *reinterpret_cast<const uint4*>(xs + lds_sw64(xrow + m * 16, lq));
#pragma unroll
    for (int n = 0; n < NREP; ++n) bf[n] = *reinterpret_cast<const uint4*>(ws + lds_sw64(wrow + n * 16, lq));
    between();                                          // next slab's DMA issue, hidden behind the MFMAs below
#pragma unroll
    for (int m = 0; m < 4; ++m)
      mma16_row<E, NREP>(af[m], bf, acc[m]);
  };
  // The same with the next slab's LDS-DMA pieces spread over the tap: `piece(m)` runs after the m-th row of MFMAs (an issue
  // costs ~100 cycles of the wave's issue slot; behind queued MFMAs the matrix pipe keeps working through it — one burst of
  // all pieces in front of a tap's MFMAs, as above, stalls them), and each input fragment is read one MFMA row ahead.
  // `piece(tap, m)` runs after MFMA row m of tap `tap`; the NEXT tap's weight fragments and first input fragment are read
  // during the current tap's rows (all taps of a slab sit in the same LDS stage), so only a slab's first tap waits for LDS.
  auto compute_slab_spread = [&](const unsigned char* xs, const unsigned char* ws, auto&& piece) {
    uint4 bf[2][NREP];
    const int wrow0 = wave_c * HALF_CO + lr, xrow0 = wave_t * 64 + lr;
#pragma unroll
    for (int n = 0; n < NREP; ++n) bf[0][n] = *reinterpret_cast<const uint4*>(ws + lds_sw64(wrow0 + n * 16, lq));
    uint4 af = *reinterpret_cast<const uint4*>(xs + lds_sw64(xrow0, lq));
#pragma unroll
    for (int tap = 0; tap < KS; ++tap) {
      const int xrow = xrow0 + tap * dil;
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        uint4 af_next = af;
        if (m + 1 < 4) af_next = *reinterpret_cast<const uint4*>(xs + lds_sw64(xrow + (m + 1) * 16, lq));
        else if (tap + 1 < KS) af_next = *reinterpret_cast<const uint4*>(xs + lds_sw64(xrow0 + (tap + 1) * dil, lq));
        if (m == 1 && tap + 1 < KS) {
#pragma unroll
          for (int n = 0; n < NREP; ++n)
            bf[(tap + 1) & 1][n] = *reinterpret_cast<const uint4*>(ws + lds_sw64((tap + 1) * TILE_CO + wrow0 + n * 16, lq));
        }
        mma16_row<E, NREP>(af, bf[tap & 1], acc[m]);
        piece(tap, m);
        af = af_next;
      }
    }
  };

#ifdef SDA_RING_SV
  constexpr bool RING = KS == 1 && NT == 1;
#else
  constexpr bool RING = KS == 1 && NT == 1 && !SV;
#endif
  if constexpr (RING) {
    // Ring of NS stages, D = NS - 1 slabs of LDS-DMA in flight.  Every wave issues EXACTLY PPW pieces per slab (2 input + 2 or 3
    // weight pieces; indices past the end are clamped, duplicates rewrite identical bytes), so a counted s_waitcnt retires slab
    // s while its successors stay in flight across the raw s_barrier.  Slab s + D goes into the buffer slab s - 1 was read
    // from: every wave has passed this iteration's barrier, i.e. has consumed it.
    constexpr int D = NS - 1;
    constexpr int NXW = 2, NWW = (TP + NW - 1) / NW, PPW = NXW + NWW;
    static_assert(4 * NXW * 16 == TILE_T, "two 16-row input pieces per wave cover the 128-row tile");
    auto issue_fixed = [&](int s, int buf, int j) {       // j-th of this wave's PPW pieces of slab s
      if (j < NXW) {
        const int pc = (wid & 3) + 4 * j;
        dma_x(s, pc, buf * STAGE + tsel * XS_BYTES + pc * 1024);
      } else {
        int q = wid + NW * (j - NXW);
        q = q < TP ? q : TP - 1;
        dma_w(s, 0, q, buf * STAGE + NT * XS_BYTES + q * 1024);
      }
    };
    auto wait_keep = [&](int slabs_in_flight) {           // all but this wave's youngest slabs_in_flight * PPW operations
      switch (slabs_in_flight * PPW) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
        case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
      }
    };
#pragma unroll
    for (int p = 0; p < D; ++p) {
      if (s_begin + p < s_end) {
#pragma unroll
        for (int j = 0; j < PPW; ++j) issue_fixed(s_begin + p, p, j);
      }
    }
    int cur = 0;
    for (int s = s_begin; s < s_end; ++s) {
      const int younger = s_end - 1 - s;
      wait_keep(younger < D - 1 ? younger : D - 1);
      __builtin_amdgcn_s_barrier();
      const bool more = s + D < s_end;
      const int nxt = cur == 0 ? NS - 1 : cur - 1;        // (cur + D) % NS: the buffer of slab s - 1
      const unsigned char* xs = smem + cur * STAGE + tsel * XS_BYTES;
      const unsigned char* ws = smem + cur * STAGE + NT * XS_BYTES;
      constexpr int PER = (PPW + 3) / 4;                  // pieces after each of the four MFMA rows
      compute_slab_spread(xs, ws, [&](int tap, int m) {
        if (more) {
#pragma unroll
          for (int i = 0; i < PER; ++i) {
            if (m * PER + i < PPW) issue_fixed(s + D, nxt, m * PER + i);
          }
        }
      });
      cur = cur == NS - 1 ? 0 : cur + 1;
    }
  } else if constexpr (NS == 2) {
    if (s_begin < s_end) {
      stage_x(s_begin, 0);
#pragma unroll
      for (int tap = 0; tap < KS; ++tap) stage_w(s_begin, 0, tap);
    }
    // Single tile, scalar bases: this wave's pieces of a slab are fixed for the whole K loop, so each keeps ONE per-lane byte
    // offset (its rows inside the tile / the weight block) against two running scalar bases that advance one slab per
    // K-step — an issue is an add for the LDS destination, the M0 write and the load (9-10 scalar instructions and an
    // s_nop 4 per piece before; dispatch_conv_nt checks that the offsets fit 31 bits)
    constexpr int L_NXW = 3, L_NWW = (KS * TP + NW - 1) / NW;
    constexpr bool LEAN = NT == 1 && SV;
    uint32_t xoff[LEAN ? L_NXW : 1], woff[LEAN ? L_NWW : 1];
    const E* xrun = xg;
    const E* wrun = wg;
    uint32_t lds_xw = 0, lds_ww = 0;
    if constexpr (LEAN) {
#pragma unroll
      for (int j = 0; j < L_NXW; ++j) xoff[j] = xvoff + (uint32_t)((size_t)(((wid & 3) + 4 * j) * 16) * a.x_pitch * sizeof(E));
#pragma unroll
      for (int i = 0; i < L_NWW; ++i) {
        int q = wid + NW * i;
        q = q < KS * TP ? q : KS * TP - 1;
        const int t2 = q / TP;
        woff[i] = wvoff + (uint32_t)(((size_t)t2 * a.Cout_p + (size_t)(q - t2 * TP) * 16) * a.w_pitch * sizeof(E));
      }
      xrun = xg + (size_t)row_base * a.x_pitch + (size_t)(s_begin + 1) * SLAB;       // slab s + 1 of the first K-step
      wrun = wg + (size_t)co0 * a.w_pitch + (size_t)(s_begin + 1) * SLAB;
      lds_xw = uniform_u32(lds_base + tsel * XS_BYTES + (wid & 3) * 1024);
      lds_ww = uniform_u32(lds_base + NT * XS_BYTES + wid * 1024);
    }
    for (int s = s_begin; s < s_end; ++s) {
      const int cur = (s - s_begin) & 1;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // this wave's pieces of slab s have landed
      __builtin_amdgcn_s_barrier();                       // ... everybody's have, and slab s-1 is fully consumed
      const bool more = s + 1 < s_end;
      const unsigned char* xs = smem + cur * STAGE + tsel * XS_BYTES;
      const unsigned char* ws = smem + cur * STAGE + NT * XS_BYTES;
      if constexpr (NT == 1 && SV) {
        // this wave's pieces of slab s + 1 in need order (input first), PER of them after each row of MFMAs
        constexpr int NXW = 3, NWW = (KS * TP + NW - 1) / NW, PER = (NXW + NWW + 4 * KS - 1) / (4 * KS);
        const uint32_t nxt = (uint32_t)((cur ^ 1) * STAGE);
        auto issue_j = [&](int j) {
          if (j < NXW) {
            const int pc = (wid & 3) + 4 * j;
            if (pc < x_pieces) lds_dma16_lean<false>(xrun, xoff[j], lds_xw + nxt + (uint32_t)(j * 4096));
          } else if (j < NXW + NWW) {
            const int q = wid + NW * (j - NXW);
            if (q < KS * TP) lds_dma16_lean<false>(wrun, woff[j - NXW], lds_ww + nxt + (uint32_t)((j - NXW) * NW * 1024));
          }
        };
        compute_slab_spread(xs, ws, [&](int tap, int m) {
          if (more) {
#pragma unroll
            for (int i = 0; i < PER; ++i) issue_j((tap * 4 + m) * PER + i);
          }
        });
        xrun += SLAB;
        wrun += SLAB;
      } else {
        if (more) stage_x(s + 1, cur ^ 1);                // DMA of the next slab overlaps the MFMAs below
#pragma unroll
        for (int tap = 0; tap < KS; ++tap)
          compute_tap(xs, ws, tap, [&] {
            if (more && tap == 0) {
#pragma unroll
              for (int t2 = 0; t2 < KS; ++t2) stage_w(s + 1, cur ^ 1, t2);
            }
          });
      }
    }
  } else {
    // Three LDS stages, two slabs in flight: every wave issues EXACTLY 3 input pieces + 4 weight pieces per
    // slab (indices clamped, duplicates rewrite identical bytes), so a counted s_waitcnt vmcnt(7) retires
    // slab s while slab s+1 stays in flight across the raw s_barrier (a __syncthreads() would drain it).
    constexpr int WQ = 4;                                 // weight pieces per wave per slab: KS*TP = 30 <= 8*4
    static_assert(KS * TP <= NW * WQ, "weight pieces do not fit the fixed per-wave DMA count");
    auto stage3_x = [&](int s, int buf) {
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        int p = (wid & 3) + 4 * i;
        p = p < x_pieces ? p : x_pieces - 1;
        dma_x(s, p, buf * STAGE + tsel * XS_BYTES + p * 1024);
      }
    };
    auto stage3_w = [&](int s, int buf, int i) {          // i-th of this wave's WQ weight pieces
      int q = wid + NW * i;
      q = q < KS * TP ? q : KS * TP - 1;
      const int tap = q / TP;
      dma_w(s, tap, q - tap * TP, buf * STAGE + NT * XS_BYTES + q * 1024);
    };
    auto stage3 = [&](int s, int buf) {
      stage3_x(s, buf);
#pragma unroll
      for (int i = 0; i < WQ; ++i) stage3_w(s, buf, i);
    };
    if (s_begin < s_end) stage3(s_begin, 0);
    if (s_begin + 1 < s_end) stage3(s_begin + 1, 1);
    int cur = 0;
    for (int s = s_begin; s < s_end; ++s) {
      if (s + 1 < s_end) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
      else               asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      const bool more = s + 2 < s_end;
      const int nxt = cur >= 1 ? cur - 1 : 2;             // (cur + 2) % 3
      if (more) stage3_x(s + 2, nxt);
      const unsigned char* xs = smem + cur * STAGE + tsel * XS_BYTES;
      const unsigned char* ws = smem + cur * STAGE + NT * XS_BYTES;
      compute_tap(xs, ws, 0, [&] { if (more) stage3_w(s + 2, nxt, 0); });
      compute_tap(xs, ws, 1, [&] { if (more) stage3_w(s + 2, nxt, 1); });
      compute_tap(xs, ws, 2, [&] { if (more) { stage3_w(s + 2, nxt, 2); stage3_w(s + 2, nxt, 3); } });
      cur = cur == 2 ? 0 : cur + 1;
    }
  }

  // ------------------------------------------------------------------ split-K: raw fp32 partials
  if (a.partial) {
    if (!tile_ok) return;
    float* __restrict__ pp = a.partial + (size_t)ks * a.T * a.Cout_p;
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int n = 0; n < NREP; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int t = t0 + wave_t * 64 + m * 16 + lq * 4 + r;
          const int co = co0 + wave_c * HALF_CO + n * 16 + lr;
          if (t < a.T) pp[(size_t)t * a.Cout_p + co] = acc[m][n][r];
        }
    return;
  }

  // ------------------------------------------------------------------ epilogue
  if (a.flags & 256) {        // diagnostic: skip the epilogue, keep the accumulators live
    float keep = 0.f;
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int n = 0; n < NREP; ++n) keep += acc[m][n][0] + acc[m][n][1] + acc[m][n][2] + acc[m][n][3];
    if (keep == 123.456f) reinterpret_cast<float*>(a.y)[0] = keep;
    return;
  }
  if (a.bias) {
#pragma unroll
    for (int n = 0; n < NREP; ++n) {
      const float bv = a.bias[co0 + wave_c * HALF_CO + n * 16 + lr];
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[m][n][r] += bv;
    }
  }
  // each tile's 4 waves run their own epilogue in a private LDS region (barriers are workgroup-wide)
  using G8 = EpiGeom<TILE_CO, 8>;                       // region sizes as reserved by conv_lds_bytes
  float* ep = reinterpret_cast<float*>(smem + tsel * (G8::EP_BYTES + G8::RED_BYTES));
  float* red = reinterpret_cast<float*>(smem + tsel * (G8::EP_BYTES + G8::RED_BYTES) + G8::EP_BYTES);
  const int ltid = tid & 255;
  const int chunk = ltid % G::NCH, rg = ltid / G::NCH;
  const bool active = (rg < G::RG) && tile_ok;
  float ssum[CH], ssq[CH];
#pragma unroll
  for (int j = 0; j < CH; ++j) { ssum[j] = 0.f; ssq[j] = 0.f; }
  E* __restrict__ yg = reinterpret_cast<E*>(a.y);
  E* __restrict__ ypre = reinterpret_cast<E*>(a.y_pre);
  const E* __restrict__ resg = reinterpret_cast<const E*>(a.res);
  const E* __restrict__ bnx = reinterpret_cast<const E*>(a.bn_x);
  const long out_row0 = a.x_row0 + (long)b * a.x_sample_rows + t0;
  // BatchNorm-backward statistics mode (BN): this thread's CH channels of (gamma, beta, mean, rstd)
  float bga[BN ? CH : 1], bbe[BN ? CH : 1], bmu[BN ? CH : 1], brs[BN ? CH : 1];
  if constexpr (BN == 1 || BN == 2) {
    if (active) {
#pragma unroll
      for (int q4 = 0; q4 < CH / 4; ++q4) {
        const int c = co0 + chunk * CH + q4 * 4;
        *reinterpret_cast<float4*>(bga + q4 * 4) = *reinterpret_cast<const float4*>(a.bn_coef + c);
        *reinterpret_cast<float4*>(bbe + q4 * 4) = *reinterpret_cast<const float4*>(a.bn_coef + a.Cout_p + c);
        *reinterpret_cast<float4*>(bmu + q4 * 4) = *reinterpret_cast<const float4*>(a.bn_coef + 2 * a.Cout_p + c);
        *reinterpret_cast<float4*>(brs + q4 * 4) = *reinterpret_cast<const float4*>(a.bn_coef + 3 * a.Cout_p + c);
      }
    }
  }

  for (int h = 0; h < 2; ++h) {
    // residual rows of this half: issue all loads up front so their latency hides behind the LDS staging
    // rows are prefetched PACKED (16 bytes = 4 registers each) and unpacked at use
    uint4 rv[G::ITERS], bx[BN ? G::ITERS : 1];
    if (resg && active) {
#pragma unroll
      for (int it = 0; it < G::ITERS; ++it) {
        const int row = rg + it * G::RG;
        rv[it] = make_uint4(0u, 0u, 0u, 0u);
        if (row < EP_ROWS && t0 + h * EP_ROWS + row < a.T)
          rv[it] = Vec16<E>::load_raw(resg + (size_t)(out_row0 + h * EP_ROWS + row) * a.Cout_p + co0 + chunk * CH);
      }
    }
    if constexpr (BN) {
      if (active) {
#pragma unroll
        for (int it = 0; it < G::ITERS; ++it) {
          const int row = rg + it * G::RG;
          bx[it] = make_uint4(0u, 0u, 0u, 0u);
          if (row < EP_ROWS && t0 + h * EP_ROWS + row < a.T)
            bx[it] = Vec16<E>::load_raw(bnx + (size_t)(out_row0 + h * EP_ROWS + row) * a.Cout_p + co0 + chunk * CH);
        }
      }
    }
    __syncthreads();            // main-loop LDS reads (h == 0) / previous half's reads are done
    if (wave_t == h) {
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < NREP; ++n)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            ep[(m * 16 + lq * 4 + r) * G::STRIDE + wave_c * HALF_CO + n * 16 + lr] = acc[m][n][r];
    }
    __syncthreads();
    if (active) {
#pragma unroll
      for (int it = 0; it < G::ITERS; ++it) {
        const int row = rg + it * G::RG;
        const int t = t0 + h * EP_ROWS + row;
        if (row < EP_ROWS && t < a.T) {
          float v[CH];
#pragma unroll
          for (int q4 = 0; q4 < CH / 4; ++q4) {
            const float4 f = *reinterpret_cast<const float4*>(ep + row * G::STRIDE + chunk * CH + q4 * 4);
            v[q4 * 4 + 0] = f.x; v[q4 * 4 + 1] = f.y; v[q4 * 4 + 2] = f.z; v[q4 * 4 + 3] = f.w;
          }
          const size_t off = (size_t)(out_row0 + h * EP_ROWS + row) * a.Cout_p + co0 + chunk * CH;
          if (resg) {
            float r8[CH];
            Vec16<E>::unpack(rv[it], r8);
#pragma unroll
            for (int j = 0; j < CH; ++j) v[j] += r8[j];
          }
          if (a.flags & SDA_EPI_GELU) {
            if (ypre) Vec16<E>::store(ypre + off, v);
#pragma unroll
            for (int j = 0; j < CH; j += 2) {          // on pairs: the 16-bit form's arithmetic is packed (two elements per issue slot)
              const f32x2 gp = gelu_pair<E>(f32x2{v[j], v[j + 1]});
              v[j] = gp.x; v[j + 1] = gp.y;
            }
          }
          if constexpr (!BN) {
            if (a.flags & SDA_EPI_GLU_BWD) {
              // v = the gradient entering F.glu (models.py:164): write the GLU backward instead of it (what bwd_colsum_kernel<E, 2>
              // does in a pass of its own, from dy as STORED: here dy never goes to memory) and keep the column sums of the two
              // halves in the statistics slots (the bias gradient of the conv that fed the GLU)
              float o8[CH], g8[CH], dg8[CH];
              Vec16<E>::load(reinterpret_cast<const E*>(a.glu_out) + off, o8);
              Vec16<E>::load(reinterpret_cast<const E*>(a.glu_gate) + off, g8);
#pragma unroll
              for (int j = 0; j < CH; ++j) {
                const float sg = sigmoid_f(g8[j]);
                dg8[j] = v[j] * o8[j] * (1.f - sg);
                v[j] *= sg;
                ssum[j] += Vec16<E>::round(v[j]);
                ssq[j] += Vec16<E>::round(dg8[j]);
              }
              const size_t off2 = (size_t)(out_row0 + h * EP_ROWS + row) * (2 * a.Cout_p) + co0 + chunk * CH;
              Vec16<E>::store(yg + off2, v);
              Vec16<E>::store(yg + off2 + a.Cout_p, dg8);
              continue;
            }
          }
          if constexpr (BN == 3) {
            // SDA_EPI_GELU_BWD (kernel size 1): v = the gradient entering a GELU whose input u = bn_x the forward kept — write
            // round(v) * GELU'(u), bwd_colsum_kernel<E, 0>'s expression on the gradient as it would have been stored, and keep
            // the column sums of the products (the bias gradient of the layer that fed the GELU)
            float u8[CH];
            Vec16<E>::unpack(bx[it], u8);
#pragma unroll
            for (int j = 0; j < CH; j += 2) {
              const f32x2 o = f32x2{Vec16<E>::round(v[j]), Vec16<E>::round(v[j + 1])} * gelu_grad_pair<E>(f32x2{u8[j], u8[j + 1]});
              v[j] = o.x; v[j + 1] = o.y;
              ssum[j] += o.x; ssum[j + 1] += o.y;
            }
            Vec16<E>::store(yg + off, v);
          } else if constexpr (BN) {
            // BatchNorm+GELU backward sums of the layer this gradient enters (what col_reduce_kernel<E, 1>
            // computes in a pass of its own): dg = dy * GELU'(gamma * xhat + beta) with dy as stored.
            // SDA_EPI_BN_STORE_DG: dg itself is what gets stored (and summed as stored) — the pass that finishes the
            // BatchNorm backward then needs no GELU' of its own
            constexpr bool store_dg = BN == 2;
            if (!store_dg) Vec16<E>::store(yg + off, v);
            float x8[CH];
            Vec16<E>::unpack(bx[it], x8);
            // (on PAIRS: gelu_grad_f on one element runs the packed 16-bit form on a duplicated pair — the same instructions
            // for half the work; per lane the arithmetic is the same, so are the results)
#pragma unroll
            for (int j = 0; j < CH; j += 2) {
              const float xh0 = (x8[j] - bmu[j]) * brs[j], xh1 = (x8[j + 1] - bmu[j + 1]) * brs[j + 1];
              const f32x2 gp = gelu_grad_pair<E>(f32x2{bga[j] * xh0 + bbe[j], bga[j + 1] * xh1 + bbe[j + 1]});
              float dg0 = Vec16<E>::round(v[j]) * gp.x, dg1 = Vec16<E>::round(v[j + 1]) * gp.y;
              if (store_dg) { dg0 = Vec16<E>::round(dg0); dg1 = Vec16<E>::round(dg1); v[j] = dg0; v[j + 1] = dg1; }
              ssum[j] += dg0; ssum[j + 1] += dg1;
              ssq[j] += dg0 * xh0; ssq[j + 1] += dg1 * xh1;
            }
            if (store_dg) Vec16<E>::store(yg + off, v);
          } else if (a.stats) {
            Vec16<E>::store(yg + off, v);
            // statistics of the values as stored (rounded to E), so BN normalises what it will read
#pragma unroll
            for (int j = 0; j < CH; ++j) { const float q = Vec16<E>::round(v[j]); ssum[j] += q; ssq[j] += q * q; }
          } else {
            Vec16<E>::store(yg + off, v);
          }
        }
      }
    }
  }
  if (a.stats) {
    __syncthreads();
    if (active) {
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        red[(rg * 2 + 0) * TILE_CO + chunk * CH + j] = ssum[j];
        red[(rg * 2 + 1) * TILE_CO + chunk * CH + j] = ssq[j];
      }
    }
    __syncthreads();
    if (tile_ok) {
      for (int i = ltid; i < 2 * TILE_CO; i += 256) {
        const int which = i / TILE_CO, c = i - which * TILE_CO;
        float s = 0.f;
        for (int g = 0; g < G::RG; ++g) s += red[(g * 2 + which) * TILE_CO + c];
        a.stats[((size_t)tt * 2 + which) * a.Cout_p + co0 + c] = s;
      }
    }
  }
}

template <typename E, int TILE_CO, int KS, int NT, int BN = 0, bool SV = true>
static int launch_conv(const sda_conv_args& a, hipStream_t st) {
  constexpr int lds = conv_lds_bytes<TILE_CO, KS, NT, SV>();
  static unsigned long long attr_done = 0;        // per device
  auto kern = conv_gemm_kernel<E, TILE_CO, KS, NT, BN, SV>;
  if (first_use_on_device(attr_done)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                            lds) != hipSuccess) {
      set_error("conv_gemm: cannot reserve %d bytes of LDS", lds);
      return -3;
    }
  }
  const int n_t = (a.T + TILE_T - 1) / TILE_T;
  const long groups = ((long)a.B * n_t + NT - 1) / NT;
  const long grid = (long)(a.Cout_p / TILE_CO) * groups * a.ksplit;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256 * NT), lds, st, a, n_t);
  return check_launch("conv_gemm");
}

template <typename E, int TILE_CO>
static int dispatch_conv_nt(const sda_conv_args& a, hipStream_t st) {
  const bool k3 = a.KS == 3;
  const int n_t = (a.T + TILE_T - 1) / TILE_T;
  // Two tiles per workgroup (shared weight slab, 3-stage counted-vmcnt pipeline) move 37 % fewer LDS-DMA bytes
  // per FLOP and are the faster form when the kernel has the CU to itself (75 vs 83 us at 320->320), but they
  // take the whole LDS of a CU (150 KB): the caller asks for them (SDA_CONV_PAIR_TILES) where nothing else wants
  // the CU — the forward pass — and keeps the single-tile form (two 80 KB workgroups per CU, room for one
  // weight-gradient workgroup beside them) in backward, where the paired form measured slower end to end.
  const bool pair = !a.widx && a.ksplit == 1 && (a.flags & SDA_CONV_PAIR_TILES) && !(a.flags & SDA_CONV_SINGLE_TILE);
  const bool gelu_bwd = a.flags & SDA_EPI_GELU_BWD;
  if (a.bn_x && !gelu_bwd) {    // BatchNorm-backward statistics epilogue: data-gradient convs of the k = 3 layers
    if (!k3) { set_error("conv_gemm: bn_x is built for kernel size 3 only"); return -1; }
  }
  if (gelu_bwd && (k3 || pair || a.widx)) { set_error("conv_gemm: SDA_EPI_GELU_BWD is built for kernel size 1, shared weights, one tile per workgroup"); return -1; }
  // scalar-base DMA needs whole 16-row pieces inside the operands: row-layout activations (slack rows behind the last
  // sample) and fully padded weights; the split-K matrix mode (plain matrices, ragged row counts) keeps per-lane addresses
  const bool sv = a.x_row0 >= PAD && a.x_sample_rows >= a.T + PAD && a.w_rows_limit >= a.Cout_p && !a.partial &&
                  a.x_rows_limit >= a.x_row0 + (long)(a.B - 1) * a.x_sample_rows + (long)n_t * TILE_T + 2 * PAD &&
                  (long)a.x_pitch * XROWS * (long)sizeof(E) < (1L << 31) &&
                  (long)a.w_pitch * ((long)a.KS * a.Cout_p + 16) * (long)sizeof(E) < (1L << 31);   // per-piece offsets: 31 bits
  if (!sv) {
    if (k3 || pair || gelu_bwd) { set_error("conv_gemm: kernel-3 / paired-tile / GELU-backward launches need row-layout operands"); return -1; }
    return launch_conv<E, TILE_CO, 1, 1, 0, false>(a, st);
  }
  if (gelu_bwd) return launch_conv<E, TILE_CO, 1, 1, 3>(a, st);
  if (a.bn_x && (a.flags & SDA_EPI_BN_STORE_DG)) return pair ? launch_conv<E, TILE_CO, 3, 2, 2>(a, st) : launch_conv<E, TILE_CO, 3, 1, 2>(a, st);
  if (a.bn_x) return pair ? launch_conv<E, TILE_CO, 3, 2, 1>(a, st) : launch_conv<E, TILE_CO, 3, 1, 1>(a, st);
  if (pair) return k3 ? launch_conv<E, TILE_CO, 3, 2>(a, st) : launch_conv<E, TILE_CO, 1, 2>(a, st);
  return k3 ? launch_conv<E, TILE_CO, 3, 1>(a, st) : launch_conv<E, TILE_CO, 1, 1>(a, st);
}

template <typename E>
static int dispatch_conv(const sda_conv_args& a, hipStream_t st) {
  if ((a.flags & SDA_CONV_FLAT_TILES) && a.KS == 3 && a.Cout_p % 160 == 0) {
    // (exactly the condition sda_conv_stats_rows uses: the statistics rows a caller sized must be the rows written)
    if (!conv3_flat_supports(a)) { set_error("conv_gemm: SDA_CONV_FLAT_TILES needs a plain row-layout kernel-3 convolution"); return -1; }
    return launch_conv3_flat(a, st);
  }
  if ((a.flags & SDA_CONV_WIDE_TILES) && a.KS == 1) {
    // (an error, not a fallback: the statistics rows a caller sized with sda_conv_stats_rows must be the rows written)
    if (!conv1_wide_supports(a)) { set_error("conv_gemm: SDA_CONV_WIDE_TILES needs a plain row-layout kernel-1 convolution, 16-bit storage, Cout_p %% 256 == 0 or %% 320 == 0"); return -1; }
    return launch_conv1_wide(a, st);
  }
  if ((a.flags & SDA_CONV_FLAT_TILES) && a.KS == 1 && conv1_flat_supports(a)) return launch_conv1_flat(a, st);
  if (a.flags & SDA_EPI_ROW_SUMSQ) {
    set_error("conv_gemm: SDA_EPI_ROW_SUMSQ needs SDA_CONV_FLAT_TILES (or SDA_CONV_WIDE_TILES) and a plain row-layout kernel-1 convolution with Cout_p % 128 == 0");
    return -1;
  }
  if ((a.flags & SDA_EPI_GELU_BWD) && (!a.bn_x || !a.stats || a.bias || a.y_pre || a.res || a.partial || a.KS != 1 ||
                                       (a.flags & (SDA_EPI_GELU | SDA_EPI_GLU | SDA_EPI_GLU_BWD)))) {
    set_error("conv_gemm: SDA_EPI_GELU_BWD needs kernel size 1, bn_x (the GELU's input) and stats, and no bias / y_pre / residual / other epilogue");
    return -1;
  }
  if (a.flags & SDA_EPI_GLU) { set_error("conv_gemm: SDA_EPI_GLU needs SDA_CONV_FLAT_TILES, kernel size 3 and Cout_p % 160 == 0"); return -1; }
  // 1x1 convs whose width divides both ways (640): 128-channel tiles (conv_final1 forward 125 -> 113 us, conv_final2's data
  // gradient 225 -> 195 us; the k = 3 convs and the statistics-row contract stay on 160)
  if (a.KS == 1 && a.Cout_p % 128 == 0 && !a.stats) return dispatch_conv_nt<E, 128>(a, st);
  if (a.Cout_p % 160 == 0) return dispatch_conv_nt<E, 160>(a, st);
  if (a.Cout_p % 128 == 0) return dispatch_conv_nt<E, 128>(a, st);
  return dispatch_conv_nt<E, 64>(a, st);
}

}  // namespace sda

using namespace sda;

extern "C" int sda_conv_n_t_tiles(int T) { return (T + TILE_T - 1) / TILE_T; }

extern "C" int sda_conv_gemm(const sda_conv_args* a, void* stream) {
  if (!a || !a->x || !a->w || (!a->y && !a->partial)) { set_error("conv_gemm: null argument"); return -1; }
  if (a->KS != 1 && a->KS != 3) { set_error("conv_gemm: kernel size %d not supported (1 or 3)", a->KS); return -1; }
  if (a->dil < 0 || a->dil > PAD) { set_error("conv_gemm: dilation %d outside [0, %d]", a->dil, PAD); return -1; }
  if (a->Cout_p % 64 || a->Cin_p % 64) { set_error("conv_gemm: channel extents (%d, %d) must be multiples of 64", a->Cin_p, a->Cout_p); return -1; }
  // (x_pitch < Cin_p is allowed for kernel size 1: overlapping rows = the im2col view of a strided convolution)
  if (a->x_pitch % 8 || a->w_pitch % 8 || a->x_pitch < 8 || (a->x_pitch < a->Cin_p && a->KS != 1) || a->w_pitch < a->Cin_p) {
    set_error("conv_gemm: bad pitch"); return -1;
  }
  if (a->ksplit < 1 || (a->ksplit > 1 && (!a->partial || a->B != 1))) { set_error("conv_gemm: split-K needs partial output and B == 1"); return -1; }
  if (a->partial && a->ksplit < 1) { set_error("conv_gemm: bad ksplit"); return -1; }
  if (a->B < 1 || a->T < 1) { set_error("conv_gemm: empty batch"); return -1; }
  if ((a->flags & SDA_EPI_GLU_BWD) && (!a->glu_out || !a->glu_gate || !a->stats || a->bn_x || a->partial || a->y_pre ||
                                     (a->flags & (SDA_EPI_GELU | SDA_EPI_GLU | SDA_CONV_FLAT_TILES)))) {
    set_error("conv_gemm: SDA_EPI_GLU_BWD needs glu_out, glu_gate and stats, and excludes bn_x / GELU / flat tiles / split-K"); return -1;
  }
  if (a->bn_x && !(a->flags & SDA_EPI_GELU_BWD) && (!a->bn_coef || !a->stats || a->partial || (a->flags & SDA_EPI_GELU))) {
    set_error("conv_gemm: bn_x needs bn_coef and stats, and excludes split-K / GELU epilogues"); return -1;
  }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (a->dtype == SDA_F32) return dispatch_conv<float>(*a, st);
  if (a->dtype == SDA_BF16) return dispatch_conv<uint16_t>(*a, st);
  if (a->dtype == SDA_F16) return dispatch_conv<half_t>(*a, st);
  set_error("conv_gemm: unknown dtype %d", a->dtype);
  return -1;
}

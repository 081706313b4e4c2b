// conv1_wide — the kernel-size-1 Conv1d (a per-row projection) on 256-row x 256- or 320-channel tiles of the FLAT row space:
// eight waves, one persistent workgroup per CU, sim_gemm's LDS-DMA ring, loss_gemm's tiled-dZ epilogue.
//
// Reference ops: conv_final1 / conv_final2 = nn.Conv1d(kernel_size=1) + GELU (models.py:194-195) and their input gradients;
// sda_conv_args contract of conv_gemm (bias, y_pre, GELU) plus SDA_EPI_GELU_BWD and SDA_EPI_ROW_SUMSQ (include/sd_amd.h),
// selected by SDA_CONV_WIDE_TILES.
//
// Why a fourth conv kernel.  The two projections and their gradients are 475 GFLOP of a 4 055-GFLOP step and ran at 360-630
// TFLOP/s: conv_gemm's 128 x 128 tile issues 16 MFMAs per barrier, conv1_flat's 256 x 128/160 tile (two 4-wave workgroups per
// CU) reaches 1.0 PFLOP/s in its K loop but its epilogue — two 189 MB outputs for conv_final2 — does not overlap the partner
// workgroup's K loop: the kernel takes the SUM of the two phases (LABNOTES, round 4).  The tiled dZ GEMM of round 5
// (loss_gemm.hip: same arithmetic intensity, 200 FLOP per HBM byte) reaches 902 TFLOP/s INCLUDING its epilogue with one
// 8-wave workgroup per CU that requests the next tile's first K-steps before it starts the epilogue of the current one.  This
// is that organisation for the 1 x 1 convs:
//   * tile = 256 rows x 256 channels (waves 2 (channels) x 4 (rows), wave tile 128 x 64, 128 accumulators) or 256 x 320
//     (waves 4 x 2, wave tile 80 x 128, 160 accumulators; the 640- and 320-wide layers): 128 / 142 FLOP per LDS-DMA byte;
//   * K-step = one 64-byte slab of input rows (16 KB) and weight rows (16 / 20 KB); ring of 4 / 3 stages, every wave issues
//     the same number of 1 KB pieces per K-step (indices past the end repeat the last piece: identical bytes), ONE counted
//     s_waitcnt + one raw barrier per K-step (32 / 40 MFMAs per wave); scalar base per piece + one per-lane offset register;
//   * the MFMA's first operand is the WEIGHT fragment: a lane's four accumulator registers are four consecutive channels of
//     one row; per 16-row group a wave passes 64 / 80 channels through a private LDS patch (4 / 5 KB, swizzled) and leaves
//     with 16 bytes per lane: 128 / 160 contiguous bytes per row — whole cache lines of y, y_pre and u;
//   * epilogues: bias + pre-activation + GELU (+ per-row sums of squares for the loss's norms, 256-wide tiles), or
//     SDA_EPI_GELU_BWD (y = round(conv) * GELU'(u) + per-tile column sums); rows that are a sample's padding are computed and
//     never stored.
#include "flat_tile.h"

namespace sda {

namespace {

template <int AFR_, int BFR_, int WJ_, int WI_, int NS_> struct W1 {
  static constexpr int AFR = AFR_, BFR = BFR_, WJ = WJ_, WI = WI_, NS = NS_;
  static constexpr int TN = WJ * AFR * 16;             // channels per tile
  static constexpr int TM = WI * BFR * 16;             // rows per tile
  static constexpr int XB = TM * ROW_B, WB = TN * ROW_B;
  static constexpr int STAGE = XB + WB;
  static constexpr int XP = TM / 16, WP = TN / 16, TOT = XP + WP;
  static constexpr int NPW = (TOT + 7) / 8;            // pieces per wave and K-step
  static constexpr int PB = (AFR % 4 == 0) ? 4 : AFR;  // fragments per epilogue block: 64 or 80 channels
  static constexpr int NBLK = AFR / PB;
  static constexpr int NC = PB * 2;                    // 8-channel chunks (16 bytes of E) per patch row
  static constexpr int RPP = 64 / NC;                  // patch rows per pass
  static constexpr int NPASS = (16 + RPP - 1) / RPP;
  static constexpr int PATCH = 16 * PB * 16 * 4;       // per wave: 16 rows x PB * 16 channels, fp32
  static constexpr int LDS = NS * STAGE + 8 * PATCH;
  static_assert(TM == 256 && WJ * WI == 8, "eight waves, 256-row tiles");
  static_assert(LDS <= 160 * 1024, "one workgroup per CU");
  static_assert(8 * PATCH >= 8 * 64 * NBLK * 8 * 4, "the column-sum image reuses the patches");
};
using W1A = W1<8, 4, 2, 4, 4>;     // 256 channels: 32 KB stages, ring of 4
using W1B = W1<5, 8, 4, 2, 3>;     // 320 channels: 36 KB stages, ring of 3

// byte offset of 16-byte chunk `cidx` (4 floats) of patch row `row`: rows are PB * 64 bytes apart — for PB = 4 (256 B: every
// row on the same banks) the chunk is XOR-ed with the row; for PB = 5 (320 B) rotated by row / 4: 16 rows, 16 distinct slots
template <int PB> __device__ __forceinline__ int patch_off(int row, int cidx) {
  if constexpr (PB == 4) return row * 256 + ((cidx ^ row) << 4);
  else {
    int c = cidx + (row >> 2);
    c = c >= PB * 4 ? c - PB * 4 : c;
    return row * (PB * 64) + (c << 4);
  }
}

// MODE 0: y = [GELU](conv + bias), y_pre = conv + bias (optional), RSQ: per-row sums of squares of y as stored.
// MODE 1: SDA_EPI_GELU_BWD: y = round(conv) * GELU'(u), u = a.bn_x; stats row per tile = column sums of y (plane 0), 0 (plane 1).
template <typename E, typename P, int MODE, bool RSQ>
__global__ __launch_bounds__(512, 2) void conv1_wide_kernel(const sda_conv_args a, const int n_row_tiles, const long total_rows, const int Tp) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int AFR = P::AFR, BFR = P::BFR, NPW = P::NPW, NS = P::NS, D = NS - 1, PB = P::PB;
  constexpr int CH = 8;
  static_assert(sizeof(E) == 2, "16-bit storage");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wj = wid % P::WJ, wi = wid / P::WJ;        // wave tile: channels [wj * AFR * 16, ...), rows [wi * BFR * 16, ...)
  const int lr = lane & 15, lq = lane >> 4;
  const int n_ch = a.Cout_p / P::TN;
  const int nks = a.Cin_p / 32;
  const E* __restrict__ xg = reinterpret_cast<const E*>(a.x);
  const E* __restrict__ wg = reinterpret_cast<const E*>(a.w);
  const uint32_t lds_base = __builtin_amdgcn_readfirstlane(lds_addr(smem));

  // tiles of this workgroup, round by round.  XCD-aware where the grid allows it (workgroups w and w + 8 share an XCD / L2):
  // the n_ch channel tiles of a row tile run on ONE XCD in the same round, so the input rows they share come from HBM once.
  const int G = (int)gridDim.x, w = (int)blockIdx.x;
  const bool xcd_map = (G % 8 == 0) && ((G / 8) % n_ch == 0);
  const int rows_per_round = xcd_map ? G / n_ch : 0;
  auto tile_of = [&](int k, int& row_tile, int& ch_tile) -> bool {
    if (xcd_map) {
      const int x = w & 7, j = w >> 3;
      row_tile = k * rows_per_round + x * ((G / 8) / n_ch) + j / n_ch;
      ch_tile = j % n_ch;
    } else {
      const long t = (long)k * G + w;
      row_tile = (int)(t / n_ch);
      ch_tile = (int)(t % n_ch);
    }
    return row_tile < n_row_tiles;
  };
  const int n_rounds = xcd_map ? (n_row_tiles + rows_per_round - 1) / rows_per_round : (int)(((long)n_row_tiles * n_ch + G - 1) / G);

  // LDS-DMA pieces: piece j < XP = input rows [16 j, 16 j + 16) of the tile, else weight rows 16 (j - XP); wave w takes
  // j = wid + 8 i (clamped).  One per-lane offset register (x_pitch == w_pitch) that also carries the K-step's channel offset.
  const int prow = lane >> 2, pchunk = lane & 3;
  const uint32_t voff0 = (uint32_t)(((size_t)prow * a.x_pitch + (size_t)((pchunk ^ sw64(prow)) * 8)) * sizeof(E));
  const char* pbase[NPW];
  uint32_t pdst[NPW];
  uint32_t kvoff = voff0;
  auto set_tile = [&](long f0, int ch0) {
    static_for<0, NPW>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      int j = wid + 8 * i;
      j = j < P::TOT ? j : P::TOT - 1;
      if (j < P::XP) {
        long srow = f0 + j * 16;
        srow = srow > a.x_rows_limit - 16 ? a.x_rows_limit - 16 : srow;       // (rows past the last sample: never stored)
        pbase[i] = reinterpret_cast<const char*>(xg + (size_t)srow * a.x_pitch);
        pdst[i] = (uint32_t)j * 1024u;
      } else {
        pbase[i] = reinterpret_cast<const char*>(wg + ((size_t)ch0 + (j - P::XP) * 16) * a.w_pitch);
        pdst[i] = (uint32_t)P::XB + (uint32_t)(j - P::XP) * 1024u;
      }
    });
    kvoff = voff0;
  };
  auto issue_all = [&](int buf) {                      // this wave's pieces of the K-step kvoff points at, into stage buf
    static_for<0, NPW>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      lds_dma16_lean<true>(pbase[i], kvoff, lds_base + (uint32_t)buf * P::STAGE + pdst[i]);
    });
  };
  auto prologue = [&](long f0, int ch0) {
    set_tile(f0, ch0);
#pragma unroll
    for (int p = 0; p < D; ++p) {
      if (p < nks) { issue_all(p); kvoff += ROW_B; }
    }
  };

  // fragment addresses inside a stage (fixed for the kernel)
  uint32_t a_addr[AFR], b_addr[BFR];
#pragma unroll
  for (int q = 0; q < AFR; ++q) a_addr[q] = (uint32_t)P::XB + (uint32_t)lds_sw64(wj * (AFR * 16) + q * 16 + lr, lq);
#pragma unroll
  for (int q = 0; q < BFR; ++q) b_addr[q] = (uint32_t)lds_sw64(wi * (BFR * 16) + q * 16 + lr, lq);

  unsigned char* patch = smem + NS * P::STAGE + wid * P::PATCH;
  const int it_row = lane / P::NC, it_c8 = lane % P::NC;            // this lane's item of an epilogue pass
  const bool it_on = lane < P::RPP * P::NC;

  int row_tile = 0, ch_tile = 0;
  bool have = n_rounds > 0 && tile_of(0, row_tile, ch_tile);
  if (have) prologue((long)row_tile * P::TM, ch_tile * P::TN);
  for (int k = 0; k < n_rounds && have; ++k) {
    const long f0 = (long)row_tile * P::TM;
    const int ch0 = ch_tile * P::TN;
    f32x4 acc[AFR][BFR];
#pragma unroll
    for (int q = 0; q < AFR; ++q)
#pragma unroll
      for (int r = 0; r < BFR; ++r) acc[q][r] = f32x4{0.f, 0.f, 0.f, 0.f};
    int cur = 0;
    for (int s = 0; s < nks; ++s) {
      int younger = nks - 1 - s;
      younger = younger < D - 1 ? younger : D - 1;
      if (younger == 2) wait_vmcnt_lit<(NS > 3 ? 2 * NPW : 0)>();
      else if (younger == 1) wait_vmcnt_lit<NPW>();
      else wait_vmcnt_lit<0>();
      __builtin_amdgcn_s_barrier();
      const bool more = s + D < nks;
      const int nxt = cur == 0 ? NS - 1 : cur - 1;
      const unsigned char* st = smem + cur * P::STAGE;
      uint4 af[AFR], bf[BFR];
#pragma unroll
      for (int q = 0; q < AFR; ++q) af[q] = *reinterpret_cast<const uint4*>(st + a_addr[q]);
#pragma unroll
      for (int r = 0; r < BFR; ++r) bf[r] = *reinterpret_cast<const uint4*>(st + b_addr[r]);
      if constexpr (AFR >= BFR) {
        static_for<0, AFR>([&](auto qc) {
          constexpr int q = decltype(qc)::value;
#pragma unroll
          for (int r = 0; r < BFR; ++r) acc[q][r] = mma16<E>(af[q], bf[r], acc[q][r]);
          // K-step s + D's pieces go out behind the MFMA rows: the matrix pipe works through the issue
          if constexpr (q * NPW / AFR != (q + 1) * NPW / AFR) {
            if (more) lds_dma16_lean<false>(pbase[q * NPW / AFR], kvoff, lds_base + (uint32_t)nxt * P::STAGE + pdst[q * NPW / AFR]);
          }
        });
      } else {
        static_for<0, BFR>([&](auto rc) {
          constexpr int r = decltype(rc)::value;
#pragma unroll
          for (int q = 0; q < AFR; ++q) acc[q][r] = mma16<E>(af[q], bf[r], acc[q][r]);
          if constexpr (r * NPW / BFR != (r + 1) * NPW / BFR) {
            if (more) lds_dma16_lean<false>(pbase[r * NPW / BFR], kvoff, lds_base + (uint32_t)nxt * P::STAGE + pdst[r * NPW / BFR]);
          }
        });
      }
      if (more) kvoff += ROW_B;
      cur = cur == NS - 1 ? 0 : cur + 1;
    }
    __syncthreads();                                    // every wave has read its last fragments: the ring is free
    int nrow = 0, nch = 0;
    const bool have_next = (k + 1 < n_rounds) && tile_of(k + 1, nrow, nch);
    if (have_next) prologue((long)nrow * P::TM, nch * P::TN);      // lands while the epilogue below runs

    // ---- epilogue.  acc[q][r]: channels ch0 + wj * AFR * 16 + q * 16 + 4 lq + i, row f0 + wi * BFR * 16 + r * 16 + lr
    E* __restrict__ yg = reinterpret_cast<E*>(a.y);
    E* __restrict__ ypre = reinterpret_cast<E*>(a.y_pre);
    const E* __restrict__ ug = reinterpret_cast<const E*>(a.bn_x);
    const bool do_gelu = a.flags & SDA_EPI_GELU;
    const long wrow0 = f0 + wi * (BFR * 16);
    const int wch0 = ch0 + wj * (AFR * 16);
    const int p_base = (int)((wrow0 + it_row) % Tp);
    float csum[MODE == 1 ? P::NBLK : 1][CH];
    if constexpr (MODE == 1) {
#pragma unroll
      for (int bl = 0; bl < P::NBLK; ++bl)
#pragma unroll
        for (int e = 0; e < CH; ++e) csum[bl][e] = 0.f;
    }
#pragma unroll
    for (int r = 0; r < BFR; ++r) {
      float rsq[RSQ ? P::NPASS : 1];
      if constexpr (RSQ) {
#pragma unroll
        for (int ps = 0; ps < P::NPASS; ++ps) rsq[ps] = 0.f;
      }
#pragma unroll
      for (int bl = 0; bl < P::NBLK; ++bl) {
#pragma unroll
        for (int a4 = 0; a4 < PB; ++a4)
          *reinterpret_cast<f32x4*>(patch + patch_off<PB>(lr, a4 * 4 + lq)) = acc[bl * PB + a4][r];
        const int ch = wch0 + bl * (PB * 16) + it_c8 * 8;
        float bv[CH];
        if constexpr (MODE == 0) {
#pragma unroll
          for (int e = 0; e < CH; ++e) bv[e] = (a.bias && it_on) ? a.bias[ch + e] : 0.f;
        }
#pragma unroll
        for (int ps = 0; ps < P::NPASS; ++ps) {
          const int prw = ps * P::RPP + it_row;
          const bool in_patch = it_on && prw < 16;
          const long f = wrow0 + r * 16 + prw;
          const int pos = (p_base + r * 16 + ps * P::RPP) % Tp;
          const bool ok = in_patch && f < total_rows && pos >= PAD;
          float v[CH] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          if (in_patch) {
            const f32x4 lo = *reinterpret_cast<const f32x4*>(patch + patch_off<PB>(prw, 2 * it_c8));
            const f32x4 hi = *reinterpret_cast<const f32x4*>(patch + patch_off<PB>(prw, 2 * it_c8 + 1));
            v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3]; v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
          }
          const size_t off = (size_t)f * a.Cout_p + ch;
          if constexpr (MODE == 0) {
            float q2 = 0.f;
            if (ok) {
#pragma unroll
              for (int e = 0; e < CH; ++e) v[e] += bv[e];
              if (ypre) Vec16<E>::store(ypre + off, v);
              if (do_gelu) {
#pragma unroll
                for (int e = 0; e < CH; e += 2) {
                  const f32x2 gp = gelu_pair<E>(f32x2{v[e], v[e + 1]});
                  v[e] = gp.x; v[e + 1] = gp.y;
                }
              }
              Vec16<E>::store(yg + off, v);
              if constexpr (RSQ) {
#pragma unroll
                for (int e = 0; e < CH; ++e) { const float qv = Vec16<E>::round(v[e]); q2 += qv * qv; }
              }
            }
            if constexpr (RSQ) {                            // the NC = 8 lanes of a row are neighbours
              q2 += __shfl_xor(q2, 1);
              q2 += __shfl_xor(q2, 2);
              q2 += __shfl_xor(q2, 4);
              rsq[ps] += q2;
            }
          } else {
            if (ok) {
              float u8[CH];
              Vec16<E>::load(ug + off, u8);
#pragma unroll
              for (int e = 0; e < CH; e += 2) {
                const f32x2 o = f32x2{Vec16<E>::round(v[e]), Vec16<E>::round(v[e + 1])} * gelu_grad_pair<E>(f32x2{u8[e], u8[e + 1]});
                v[e] = o.x; v[e + 1] = o.y;
                csum[bl][e] += o.x; csum[bl][e + 1] += o.y;
              }
              Vec16<E>::store(yg + off, v);
            }
          }
        }
      }
      if constexpr (RSQ) {
        // per-row partial sums of squares over this wave's 128 channels: stats [buffer row][Cout_p / 128]
#pragma unroll
        for (int ps = 0; ps < P::NPASS; ++ps) {
          const int prw = ps * P::RPP + it_row;
          const long f = wrow0 + r * 16 + prw;
          const int pos = (p_base + r * 16 + ps * P::RPP) % Tp;
          if (it_on && it_c8 == 0 && prw < 16 && f < total_rows && pos >= PAD)
            a.stats[(size_t)f * (a.Cout_p / 128) + (wch0 >> 7)] = rsq[ps];
        }
      }
    }
    if constexpr (MODE == 1) {
      // column sums of the tile: lanes' partial sums -> LDS (the patches, all waves done with them) -> one thread per channel
      __syncthreads();
      float* red = reinterpret_cast<float*>(smem + NS * P::STAGE);
#pragma unroll
      for (int bl = 0; bl < P::NBLK; ++bl)
#pragma unroll
        for (int e = 0; e < CH; e += 4)
          *reinterpret_cast<f32x4*>(red + ((size_t)(wid * 64 + lane) * P::NBLK + bl) * CH + e) =
              f32x4{csum[bl][e], csum[bl][e + 1], csum[bl][e + 2], csum[bl][e + 3]};
      __syncthreads();
      for (int c = tid; c < P::TN; c += 512) {
        const int cwj = c / (AFR * 16), cin = c % (AFR * 16), bl = cin / (PB * 16), c8 = (cin % (PB * 16)) >> 3, e = cin & 7;
        float s = 0.f;
        for (int vi = 0; vi < P::WI; ++vi)
          for (int rr = 0; rr < P::RPP; ++rr)
            s += red[((size_t)((vi * P::WJ + cwj) * 64 + rr * P::NC + c8) * P::NBLK + bl) * CH + e];
        a.stats[((size_t)row_tile * 2 + 0) * a.Cout_p + ch0 + c] = s;
        a.stats[((size_t)row_tile * 2 + 1) * a.Cout_p + ch0 + c] = 0.f;
      }
      __syncthreads();                                  // the patches are free again
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the next tile's first K-steps (and this epilogue's own traffic)
    have = have_next;
    row_tile = nrow; ch_tile = nch;
  }
}

template <typename E, typename P, int MODE, bool RSQ>
int launch_wide(const sda_conv_args& a, hipStream_t st) {
  static unsigned long long attr_done = 0;        // per device
  auto kern = conv1_wide_kernel<E, P, MODE, RSQ>;
  if (first_use_on_device(attr_done)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, P::LDS) != hipSuccess) {
      set_error("conv1_wide: cannot reserve %d bytes of LDS", P::LDS);
      return -3;
    }
  }
  const long total_rows = (long)a.B * rows_tp(a.T);
  const int n_row_tiles = (int)((total_rows + P::TM - 1) / P::TM);
  const int n_ch = a.Cout_p / P::TN;
  long grid = launch_cus();
  grid = grid / 8 * 8;
  if (grid < 8) grid = launch_cus();
  const long tiles = (long)n_row_tiles * n_ch;
  if (grid > tiles) grid = tiles;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), P::LDS, st, a, n_row_tiles, total_rows, rows_tp(a.T));
  return check_launch("conv1_wide");
}

template <typename E> int launch_wide_e(const sda_conv_args& a, hipStream_t st) {
  const bool gb = a.flags & SDA_EPI_GELU_BWD, rsq = a.flags & SDA_EPI_ROW_SUMSQ;
  if (a.Cout_p % 256 == 0) {
    if (gb) return launch_wide<E, W1A, 1, false>(a, st);
    return rsq ? launch_wide<E, W1A, 0, true>(a, st) : launch_wide<E, W1A, 0, false>(a, st);
  }
  if (gb) return launch_wide<E, W1B, 1, false>(a, st);
  return launch_wide<E, W1B, 0, false>(a, st);
}

}  // namespace

bool conv1_wide_supports(const sda_conv_args& a) {
  const bool gb = a.flags & SDA_EPI_GELU_BWD, rsq = a.flags & SDA_EPI_ROW_SUMSQ;
  if (a.KS != 1 || (a.dtype != SDA_BF16 && a.dtype != SDA_F16)) return false;
  if (!(a.Cout_p % 256 == 0 || a.Cout_p % 320 == 0) || a.Cin_p % 32 != 0 || a.Cin_p < 32) return false;
  if (rsq && (a.Cout_p % 256 != 0 || gb)) return false;
  if (a.res || a.widx || a.partial || a.ksplit != 1 || !a.y || (a.flags & (SDA_EPI_GLU | SDA_EPI_GLU_BWD))) return false;
  if (gb ? (!a.bn_x || !a.stats || a.bias || a.y_pre || (a.flags & SDA_EPI_GELU)) : (a.bn_x != nullptr)) return false;
  if (!gb && !rsq && a.stats) return false;
  if (rsq && !a.stats) return false;
  return a.x_row0 == PAD && a.x_pitch == a.w_pitch && a.x_pitch >= a.Cin_p && a.x_sample_rows == rows_tp(a.T) &&
         a.x_rows_limit >= (long)a.B * rows_tp(a.T) + PAD && a.x_rows_limit < (1L << 31) && a.w_rows_limit >= a.Cout_p &&
         (long)a.x_pitch * 16 * 2 < (1L << 31);
}

int launch_conv1_wide(const sda_conv_args& a, hipStream_t st) {
  return a.dtype == SDA_BF16 ? launch_wide_e<uint16_t>(a, st) : launch_wide_e<half_t>(a, st);
}

int conv1_wide_stat_rows(int B, int T) { return (int)(((long)B * rows_tp(T) + 255) / 256); }

}  // namespace sda

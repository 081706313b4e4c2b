// sim_gemm — the loss's similarity matmul on 256 x 256 output tiles:
//     S[i][j] = sum_k X[i][k] * W[j][k]        X = packed speech rows (M x K), W = brain embeddings (N x K), 16-bit storage
// Reference op this replaces: loss.py:68 (torch.einsum("bft,nft->bn", x, y) of the CLIP loss' fast path), at the full
// contraction length K = F * T (368 640 at config 2) and, under data parallelism, against the all-gathered speech rows
// (M = B_global: 2048 or 4096).
//
// Why a kernel of its own.  The product is a thin one — a few hundred output rows and columns over a contraction of
// hundreds of thousands — so it streams BOTH operands once from HBM and does little else: at 256 x 256 samples 378 MB for
// 48 GFLOP, at 2048 x 256 1.7 GB for 387 GFLOP.  conv_gemm's split-K matrix mode covers it with 128 x 128 tiles, i.e. every
// row of either operand is pulled through LDS for two tiles (604 MB fetched for 378 algorithmic, profiles/r03_pmc_summary)
// at 64 FLOP per LDS-DMA byte.  Here ONE workgroup owns a 256 x 256 tile over its K slice: every operand byte of the slice
// enters LDS exactly once (128 FLOP / byte), eight waves (2 x 4, wave tile 128 (j) x 64 (i)) share it, and the K loop is the
// ring conv_gemm's matrix mode uses: four 32 KB stages, three slabs of LDS-DMA in flight behind counted s_waitcnt vmcnt,
// one raw s_barrier per 32-deep K-step (32 MFMAs per wave).  All DMA addresses are a scalar base advanced by a scalar add
// per K-step plus four per-lane byte offsets computed once (rows past the operands are clamped: they feed outputs that are
// never stored).  Bound: HBM into LDS (~6 TB/s, MI355X_MICROARCH.md), then the matrix pipe.  Measured alone (SQ counters,
// tools/probes/pmc_sim.sh, 2048 x 256): 525-590 us = 6 TB/s of LDS-DMA (half of it the 1.7 GB from HBM, half re-reads of W
// from L2), matrix pipe 43 % busy, waves 33 % of their cycles in s_waitcnt — on vmcnt, i.e. on LDS-DMA landing: LDS waits
// 6 %, no bank conflicts, 0.19 vector and 1.1 scalar instructions per MFMA.
//
// Orientation: the MFMA's first operand is W (rows j), its second X (rows i), so a lane's four accumulator registers are four
// CONSECUTIVE j of one output row i: the K-split partial sums leave as 16-byte stores.  Output: fp32 partial[ks][i][j]
// (ks < ksplit; summed in fixed order by sda_reduce_slabs, deterministic).
#include "sd_common.h"
#include "flat_tile.h"

namespace sda {

namespace {

constexpr int SG_TILE = 256;                       // output rows and columns per workgroup
constexpr int SG_OP_BYTES = SG_TILE * ROW_B;       // one operand's stage: 256 rows x 64 B = 16 KB
constexpr int SG_STAGE = 2 * SG_OP_BYTES;          // 32 KB
constexpr int SG_NS = 4;                           // LDS stages; SG_NS - 1 slabs of LDS-DMA in flight (128 KB: one workgroup per CU;
                                                   // five stages, 160 KB: 570 vs 590 us at 2048 x 256, 107 vs 103 at 256 x 256)
constexpr int SG_PPW = 4;                          // 1 KB pieces per wave and slab: 2 of W + 2 of X (8 waves x 4 = 32 pieces)

// M32: the K-step on v_mfma_f32_32x32x16 (sd_common.h, mma32): per wave 4 (j) x 2 (i) blocks of 32 x 32, two MFMAs each per
// K-step; the 64-byte rows' swizzle is then chunk ^ ((row >> 2) & 3) — a 16-lane service group of ds_read_b128 (conv_tile.h)
// holds the row quads {0, 3, 5, 6} or {1, 2, 4, 7} of ONE chunk.
template <typename E, int NS, bool M32>
__global__ __launch_bounds__(512, 2) void sim_gemm_kernel(const E* __restrict__ X, const E* __restrict__ W, float* __restrict__ partial,
                                                           const int M, const int N, const int Np, const long pitch, const int nslab,
                                                           const int ksplit, const int m_tiles, const int n_tiles) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int SLAB = ROW_B / (int)sizeof(E);     // 32 elements per K-step
  constexpr int PER16 = Elem<E>::PER16;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wj = wid & 1, wi = wid >> 1;           // wave tile: columns j [wj * 128, +128), rows i [wi * 64, +64)
  const int lr = lane & 15, lq = lane >> 4;

  // XCD-aware order (blocks b and b + 8 share an XCD / L2): the output tiles of one K slice run on one XCD side by side, so
  // the operand rows two tiles share are fetched from HBM once.  Pure speed: any placement is correct.
  const int tiles = m_tiles * n_tiles;
  int ks, tile;
  {
    const int bid = blockIdx.x, group = 8 * tiles, full = (int)(gridDim.x / group) * group;
    if (bid < full) {
      const int base = bid / group, rem = bid - base * group;
      ks = base * 8 + (rem & 7);
      tile = rem >> 3;
    } else {
      const int rem = bid - full;
      ks = full / tiles + rem / tiles;
      tile = rem % tiles;
    }
  }
  const int m0 = (tile / n_tiles) * SG_TILE, n0 = (tile % n_tiles) * SG_TILE;
  const int per_split = (nslab + ksplit - 1) / ksplit;
  const int s_begin = ks * per_split;
  const int s_end = min(nslab, s_begin + per_split);

  // LDS-DMA pieces: one wave instruction = 16 rows x 64 B, landing lane-linearly; the XOR swizzle of the 64-byte rows goes on
  // the SOURCE chunk.  Wave w fetches pieces w and w + 8 of either operand.
  const int prow = lane >> 2, pchunk = lane & 3;
  uint32_t xoff[2], woff[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int r = (wid + 8 * h) * 16 + prow;                       // row of the tile
    const int sw = (pchunk ^ (M32 ? ((r >> 2) & 3) : sw64(r))) * PER16;
    const long xr = min((long)r, (long)(M - 1 - m0)), wr = min((long)r, (long)(N - 1 - n0));      // clamped: never stored
    xoff[h] = (uint32_t)((xr * pitch + sw) * (long)sizeof(E));
    woff[h] = (uint32_t)((wr * pitch + sw) * (long)sizeof(E));
  }
  const uint32_t lds_base = __builtin_amdgcn_readfirstlane(lds_addr(smem));
  // wave-uniform bases of K-step s_begin (scalar arithmetic on kernel arguments and the block index only)
  const E* xs = X + (size_t)m0 * pitch + (size_t)s_begin * SLAB;
  const E* ws = W + (size_t)n0 * pitch + (size_t)s_begin * SLAB;
  // this wave's four pieces of the slab `xs` / `ws` point at, into stage `buf`; j selects the piece (compile-time in the callers)
  auto issue = [&](int buf, int j) {
    const uint32_t dst = lds_base + buf * SG_STAGE + (wid + 8 * (j & 1)) * 1024;
    if (j < 2) lds_dma16_lean<false>(ws, woff[j & 1], dst);
    else lds_dma16_lean<false>(xs, xoff[j & 1], dst + SG_OP_BYTES);
  };
  auto advance = [&]() { xs += SLAB; ws += SLAB; };

  f32x4 acc[M32 ? 1 : 8][M32 ? 1 : 4];
  f32x16 acc32[M32 ? 4 : 1][M32 ? 2 : 1];
  if constexpr (M32) {
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc32[a][b][v] = 0.f;
  } else {
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const int l32 = lane & 31, lh = lane >> 5;

  constexpr int D = NS - 1;
#pragma unroll
  for (int p = 0; p < D; ++p) {
    if (s_begin + p < s_end) {
#pragma unroll
      for (int j = 0; j < SG_PPW; ++j) issue(p, j);
      advance();
    }
  }
  int cur = 0;
  for (int s = s_begin; s < s_end; ++s) {
    // retire slab s: all but this wave's `younger` newest slabs (4 pieces each), then everybody's
    const int younger = s_end - 1 - s;
    if (younger >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (younger == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const bool more = s + D < s_end;
    const int nxt = cur == 0 ? NS - 1 : cur - 1;            // (cur + D) % NS: the stage slab s - 1 was read from
    const unsigned char* wsm = smem + cur * SG_STAGE;          // W rows (first MFMA operand -> accumulator rows = j)
    const unsigned char* xsm = wsm + SG_OP_BYTES;              // X rows (second operand -> accumulator columns = i)
    // all twelve fragments of the K-step are requested at once, right behind the barrier
    uint4 bf[4], af[8];
    if constexpr (M32) {
      auto at = [&](int row, int chunk) { return row * ROW_B + ((chunk ^ ((row >> 2) & 3)) << 4); };
#pragma unroll
      for (int b = 0; b < 4; ++b) bf[b] = *reinterpret_cast<const uint4*>(xsm + at(wi * 64 + (b >> 1) * 32 + l32, 2 * (b & 1) + lh));
#pragma unroll
      for (int a = 0; a < 8; ++a) af[a] = *reinterpret_cast<const uint4*>(wsm + at(wj * 128 + (a >> 1) * 32 + l32, 2 * (a & 1) + lh));
      // af[2 A + h], bf[2 B + h]: block A / B, K half h; h outer, so an accumulator block comes round every eighth MFMA
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int a = 0; a < 4; ++a) {
#pragma unroll
          for (int b = 0; b < 2; ++b) acc32[a][b] = mma32<E>(af[2 * a + h], bf[2 * b + h], acc32[a][b]);
          if ((a & 1) && more) issue(nxt, 2 * h + (a >> 1));
        }
    } else {
#pragma unroll
    for (int b = 0; b < 4; ++b) bf[b] = *reinterpret_cast<const uint4*>(xsm + lds_sw64(wi * 64 + b * 16 + lr, lq));
#pragma unroll
    for (int a = 0; a < 8; ++a) af[a] = *reinterpret_cast<const uint4*>(wsm + lds_sw64(wj * 128 + a * 16 + lr, lq));
#pragma unroll
    for (int a = 0; a < 8; ++a) {
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = mma16<E>(af[a], bf[b], acc[a][b]);
      // slab s + D's pieces go out behind MFMA rows 1, 3, 5, 7: the matrix pipe works through the issue
      if ((a & 1) && more) issue(nxt, a >> 1);
    }
    }
    if (more) advance();
    cur = cur == NS - 1 ? 0 : cur + 1;
  }

  // K-split partial sums: lane (lq, lr) of block (a, b) holds S[i][j .. j + 3], i = m0 + wi * 64 + b * 16 + lr,
  // j = n0 + wj * 128 + a * 16 + 4 * lq
  float* __restrict__ P = partial + (size_t)ks * M * Np;
  if constexpr (M32) {
    // block (a, b), register v: S[i][j], i = m0 + wi * 64 + b * 32 + (lane & 31), j = n0 + wj * 128 + a * 32 + 8 * (v >> 2) +
    // 4 * (lane >> 5) + (v & 3): four consecutive j per register quad
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int i = m0 + wi * 64 + b * 32 + l32;
      if (i < M) {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int j = n0 + wj * 128 + a * 32 + 8 * g + 4 * lh;
            if (j + 3 < Np)
              *reinterpret_cast<f32x4*>(P + (size_t)i * Np + j) =
                  f32x4{acc32[a][b][4 * g], acc32[a][b][4 * g + 1], acc32[a][b][4 * g + 2], acc32[a][b][4 * g + 3]};
          }
      }
    }
    return;
  }
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    const int i = m0 + wi * 64 + b * 16 + lr;
    if (i < M) {
#pragma unroll
      for (int a = 0; a < 8; ++a) {
        const int j = n0 + wj * 128 + a * 16 + 4 * lq;
        if (j + 3 < Np) *reinterpret_cast<f32x4*>(P + (size_t)i * Np + j) = acc[a][b];
      }
    }
  }
}

template <typename E, int NS>
int launch_sim_ns(const void* X, const void* W, float* partial, int M, int N, int Np, long K, long pitch, int ksplit, hipStream_t st) {
  constexpr int lds = NS * SG_STAGE;
  static unsigned long long attr_done = 0;        // per device
  static const bool m32 = []() { const char* e = getenv("SDA_SIM_MFMA32"); return !e || atoi(e) != 0; }();
  auto kern = m32 ? sim_gemm_kernel<E, NS, true> : sim_gemm_kernel<E, NS, false>;
  if (first_use_on_device(attr_done)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
      set_error("sim_gemm: cannot reserve %d bytes of LDS", lds);
      return -3;
    }
  }
  const int m_tiles = (M + SG_TILE - 1) / SG_TILE, n_tiles = (N + SG_TILE - 1) / SG_TILE;
  const int nslab = (int)(K / (ROW_B / (long)sizeof(E)));
  const long grid = (long)m_tiles * n_tiles * ksplit;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), lds, st, reinterpret_cast<const E*>(X), reinterpret_cast<const E*>(W),
                     partial, M, N, Np, pitch, nslab, ksplit, m_tiles, n_tiles);
  return check_launch("sim_gemm");
}
template <typename E>
int launch_sim(const void* X, const void* W, float* partial, int M, int N, int Np, long K, long pitch, int ksplit, hipStream_t st) {
  return launch_sim_ns<E, SG_NS>(X, W, partial, M, N, Np, K, pitch, ksplit, st);
}

}  // namespace
}  // namespace sda

using namespace sda;

extern "C" int sda_sim_gemm_ksplit(int M, int N, long K, int dtype) {
  // K slices per output tile: one workgroup per CU (128 KB of LDS each) and a grid of about one round of them, in multiples of
  // 8 (the XCD dealing); 0 = this shape is not served (fp32 storage, a contraction that is not whole 64-byte K-steps)
  if (dtype != SDA_BF16 && dtype != SDA_F16) return 0;
  if (M < 1 || N < 1 || K < 32 || K % 32 != 0) return 0;
  const long tiles = (long)((M + SG_TILE - 1) / SG_TILE) * ((N + SG_TILE - 1) / SG_TILE);
  const long nslab = K / 32;
  long ks = (launch_cus() + tiles - 1) / tiles;
  ks = (ks + 7) / 8 * 8;
  if (ks > nslab) ks = nslab;
  return (int)(ks < 1 ? 1 : ks);
}

extern "C" int sda_sim_gemm(const void* X, const void* W, float* partial, int M, int N, int Np, long K, long pitch, int ksplit, int dtype,
                            void* stream) {
  if (!X || !W || !partial) { set_error("sim_gemm: null argument"); return -1; }
  if (dtype != SDA_BF16 && dtype != SDA_F16) { set_error("sim_gemm: 16-bit storage only (the fp32 path keeps conv_gemm's split-K mode)"); return -1; }
  if (M < 1 || N < 1 || K < 32 || K % 32 != 0 || pitch < K || pitch % 8 != 0) { set_error("sim_gemm: K must be whole 64-byte K-steps, rows 16-byte aligned"); return -1; }
  if (Np < N || Np % 4 != 0) { set_error("sim_gemm: the padded width must cover N in whole 16-byte groups"); return -1; }
  if (ksplit < 1 || ksplit > K / 32) { set_error("sim_gemm: ksplit out of range"); return -1; }
  if ((long)SG_TILE * pitch * 2 >= (1L << 32)) { set_error("sim_gemm: a 256-row tile of the operands spans more than 4 GB"); return -1; }
  if ((((uintptr_t)X | (uintptr_t)W) & 15) != 0 || (((uintptr_t)partial) & 15) != 0) { set_error("sim_gemm: operands must be 16-byte aligned"); return -1; }
  hipStream_t st = (hipStream_t)stream;
  return dtype == SDA_BF16 ? launch_sim<uint16_t>(X, W, partial, M, N, Np, K, pitch, ksplit, st)
                           : launch_sim<half_t>(X, W, partial, M, N, Np, K, pitch, ksplit, st);
}

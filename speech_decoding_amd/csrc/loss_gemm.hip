// loss_gemm — the embedding gradient of the CLIP loss on one GPU, as a STREAMING kernel.
//
// Reference op: autograd of `logits = x @ y.T` (loss.py:68) with respect to the brain embeddings y = Z:
//     dZ[j][k] = dloss * ( cscale[j] * sum_{i < Bm} G[i][j] * Y[i][k]  -  rscale[j] * Z[j][k] ),     j < Bn, k < row_elems
// (G, cscale, rscale from sda_clip_grad).  The contraction runs over the BATCH only (Bm <= 256 speech rows), the other
// extent is a whole embedding (F x T = 368 640 ... 1 024 000 elements): 50 GFLOP against 591 MB of unavoidable traffic
// at config 2 (read Y, read Z, write dZ, 197 MB each) — HBM-bound by a factor of four at any sensible MFMA rate.  The
// general kernel (wgrad_gemm, typed output) tiles this as 6 016 independent 128 x 128 problems of four K-chunks each,
// every one with its own prologue, barriers and fp32 staging epilogue: 250 us.  Here instead:
//   * the coefficient matrix never moves: every wave keeps G's fragments for its 64 columns j in REGISTERS for the whole
//     kernel (<= 8 K-steps x 4 fragments = 128 VGPRs), loaded once;
//   * persistent workgroups (two per CU) walk 64-element column tiles of the embeddings: the (256 x 128 B) slice of Y
//     arrives by LDS-DMA, double-buffered one tile ahead, is read as the TRANSPOSED MFMA operand (ds_read_b64_tr_b16:
//     the contraction index is the image's row), D[k][j] = sum_i Y[i][k] G[i][j] lands with 4 consecutive k per lane;
//   * the epilogue is per wave (no workgroup barrier): 16 x 32 accumulator blocks pass through a 2 KB private LDS patch
//     to become 16-byte row segments, Z is read and dZ written with 16 bytes per lane, each byte exactly once.
// Y, Z and dZ are each touched once; bound: HBM (591 MB per launch at config 2 -> ~100 us at 6 TB/s).
#include <cstdlib>

#include "sd_common.h"
#include "flat_tile.h"
#include "tr_operand.h"

namespace sda {

namespace {

constexpr int DZ_KT = 64;                              // embedding elements (columns) per tile: 128 bytes per row
constexpr int DZ_ROWS = 256;                           // contraction rows staged per tile (Bm <= 256)
constexpr int DZ_YB = DZ_ROWS * DZ_KT * 2;             // 32 KB per Y image
constexpr int DZ_STG = 16 * 32 * 4;                    // per-wave epilogue patch: 16 rows j x 32 columns k, fp32
constexpr int DZ_LDS = 2 * DZ_YB + 4 * DZ_STG;         // 73 728 B: two workgroups per CU

template <typename E>
__global__ __launch_bounds__(256, 2) void clip_dz_kernel(const E* __restrict__ G, long g_pitch, const E* __restrict__ Y,
                                                         const E* __restrict__ Z, E* __restrict__ out,
                                                         const float* __restrict__ cscale, const float* __restrict__ rscale,
                                                         const float* __restrict__ out_scale, int Bm, int Bn, long row_elems,
                                                         int ntiles) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lq = lane >> 4;
  const int jwave = blockIdx.y * 256 + wid * 64;       // this wave's 64 brain columns j
  const int nks = (Bm + 31) >> 5;                      // K-steps of 32 speech rows

  // G fragments (the MFMA's B operand: column j = lr, contraction i in the order the transposed reads of Y deliver it:
  // lane group lq holds i = 32 ks + {4 lq .. 4 lq + 3, 16 + 4 lq .. 16 + 4 lq + 3}); rows / columns past the matrix are zero
  uint4 gf[8][4];
#pragma unroll
  for (int ks = 0; ks < 8; ++ks)
#pragma unroll
    for (int nf = 0; nf < 4; ++nf) {
      uint32_t w[4] = {0u, 0u, 0u, 0u};
      const int j = jwave + nf * 16 + lr;
      if (ks < nks && j < Bn) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int i = ks * 32 + (e < 4 ? 4 * lq + e : 16 + 4 * lq + (e - 4));
          const uint32_t v = i < Bm ? (uint32_t)*reinterpret_cast<const uint16_t*>(G + (size_t)i * g_pitch + j) : 0u;
          w[e >> 1] |= v << (16 * (e & 1));
        }
      }
      gf[ks][nf] = make_uint4(w[0], w[1], w[2], w[3]);
    }

  // LDS-DMA of one Y tile: 32 pieces of 8 rows x 128 B, lane-linear in LDS, chunk swizzle on the source (tr_operand.h)
  const uint32_t lds_base = __builtin_amdgcn_readfirstlane(lds_addr(smem));
  const int prow = lane >> 3, pchunk = lane & 7;
  // (scalar base = the tile's first column, ONE 32-bit per-lane offset per piece, fixed for the kernel: eight registers
  // instead of eight 64-bit pointers — those were spilled, and a scratch reload inside the loop is a compiler-generated
  // vmcnt(0), i.e. a drained DMA pipeline)
  uint32_t yoff[8];
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    const int row = (wid * 8 + p) * 8 + prow;
    const int src_row = row < Bm ? row : Bm - 1;       // (rows past the batch meet zero coefficients: any finite data will do)
    yoff[p] = (uint32_t)(((size_t)src_row * row_elems + (size_t)((pchunk ^ chunk_xor<E, 128>(row)) * 8)) * sizeof(E));
  }
  auto stage = [&](int tile, int buf) {
    const E* yk = Y + (long)tile * DZ_KT;
    const uint32_t dst = lds_base + buf * DZ_YB + wid * 8 * 1024;
#pragma unroll
    for (int p = 0; p < 8; ++p)
      asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(yoff[p]), "s"(yk), "s"(dst + p * 1024) : "memory");
  };

  float* stg = reinterpret_cast<float*>(smem + 2 * DZ_YB + wid * DZ_STG);
  const float os = out_scale ? out_scale[0] : 1.f;
  // ---- the tile loop, with every vector-memory operation counted by hand.
  // hipcc knows nothing of the LDS-DMA issued from asm, so ANY load it does know of poisons the pipeline: its wait for that
  // load is "all but my own younger loads", which drains the next tile's DMA issued in between (the first version read Z with
  // ordinary loads: by the end of a tile's epilogue the prefetch had been waited for — 3.9 TB/s).  So Z, too, is loaded from
  // asm, the per-column factors are hoisted out of the loop, and the only compiler-visible memory operations left in it are
  // the stores, which nothing waits for.  Per tile and wave, in issue order:
  //     [L loads of Z (this tile)] [D = 8 LDS-DMA pieces (next tile; 0 on the last)] ... [L stores, one per epilogue pass]
  // with L = 2 * nv, nv = the wave's 16-column blocks that hold at least one valid column (a pass without one is skipped:
  // wave-uniform, so the counts are exact).  The pass that needs Z load q has seen q stores since: the operations younger
  // than that load are (L - 1 - q) + D + q = L - 1 + D — one literal per tile; at the top of a tile the D pieces issued
  // during the previous one are followed by its L stores: vmcnt(L) leaves exactly those stores in flight.
  const int jr = lane >> 2, c8 = lane & 3;             // row of an epilogue patch, which 8 of its 32 columns
  int nv = (Bn - jwave + 15) >> 4;
  nv = nv < 0 ? 0 : (nv > 4 ? 4 : nv);
  nv = __builtin_amdgcn_readfirstlane(nv);
  float csr[4], rsr[4];
  uint32_t zoff[4];                                    // this lane's byte offset into Z / dZ per 16-column block (rows past the matrix: the last row)
#pragma unroll
  for (int n = 0; n < 4; ++n) {
    const int j = jwave + n * 16 + jr;
    csr[n] = (j < Bn && cscale) ? cscale[j] : 1.f;
    rsr[n] = j < Bn ? rscale[j] : 0.f;
    zoff[n] = (uint32_t)(((size_t)(j < Bn ? j : Bn - 1) * row_elems + c8 * 8) * sizeof(E));
  }
  // (waves with a ragged last block — nv < 4, the edge of a batch that is not a multiple of 64 — take vmcnt(0): always
  // safe, and they are few; the full waves get the literals)
  auto wait_later = [&](bool more) {                    // vmcnt(2 nv - 1 + (more ? 8 : 0)) for nv == 4
    if (nv == 4) {
      if (more) asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  };
  int tile = blockIdx.x, buf = 0;
  if (tile < ntiles) stage(tile, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the first tile (and the coefficient loads above)
  bool first = true;
  for (; tile < ntiles; tile += gridDim.x, buf ^= 1) {
    if (!first) {                                       // this wave's pieces of the tile have landed; the previous tile's stores may fly on
      if (nv == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    first = false;
    __builtin_amdgcn_s_barrier();                       // ... everybody's; the other image is no longer being read
    const long k0 = (long)tile * DZ_KT;
    // this lane's 16-byte pieces of Z for the tile's epilogue passes, requested now: they arrive behind the MFMAs
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    u32x4 zraw[2][4];
#pragma unroll
    for (int half = 0; half < 2; ++half)
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        if (n < nv) {                                   // (wave-uniform)
          const E* zk = Z + k0 + half * 32;             // scalar base of the tile's half; the lane's part is zoff[n]
          asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(zraw[half][n]) : "v"(zoff[n]), "s"(zk) : "memory");
        }
      }
    const bool more = tile + (int)gridDim.x < ntiles;
    if (more) stage(tile + gridDim.x, buf ^ 1);
    const unsigned char* img = smem + buf * DZ_YB;
#pragma unroll
    for (int half = 0; half < 2; ++half) {             // 32 columns k at a time: 8 accumulator fragments
      f32x4 acc[2][4];
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        if (ks < nks) {
          const uint4 a0 = tr_operand_bf16<128>(img, ks * 32, half * 32, lane);
          const uint4 a1 = tr_operand_bf16<128>(img, ks * 32, half * 32 + 16, lane);
#pragma unroll
          for (int n = 0; n < 4; ++n) {
            acc[0][n] = mma16<E>(a0, gf[ks][n], acc[0][n]);
            acc[1][n] = mma16<E>(a1, gf[ks][n], acc[1][n]);
          }
        }
      }
      // accumulator layout: column j = lr, rows k = 16 m + 4 lq + r.  Per 16-column block nf: patch[j][k] fp32 (XOR-swizzled
      // 16-byte chunks), then lane -> (row j = lane >> 2, 8 consecutive k): 16-byte accesses to Z and dZ
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        if (n >= nv) continue;                          // (wave-uniform: no valid column in this block, nothing loaded, nothing stored)
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          const int ch = (m * 4 + lq) ^ (lr & 7);      // 16-byte chunk (4 floats) of row lr, swizzled
          *reinterpret_cast<f32x4*>(stg + lr * 32 + ch * 4) = acc[m][n];
        }
        const f32x4 lo = *reinterpret_cast<const f32x4*>(stg + jr * 32 + (((2 * c8) ^ (jr & 7)) * 4));
        const f32x4 hi = *reinterpret_cast<const f32x4*>(stg + jr * 32 + (((2 * c8 + 1) ^ (jr & 7)) * 4));
        // the counted wait for this pass's Z piece, tied to the registers it guards (the compiler may not use them above it)
        wait_later(more);
        asm volatile("" : "+v"(zraw[half][n]));           // (one 128-bit operand: nothing may read these registers above the wait;
                                                            //  tools/check_dma_hazard.py audits the listing for a touch in between)
        const int j = jwave + n * 16 + jr;
        if (j < Bn) {
          const size_t off = (size_t)(zoff[n] / sizeof(E)) + k0 + half * 32;
          const float cs = csr[n], rs = rsr[n];
          float z[8], v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          Vec16<E>::unpack(make_uint4(zraw[half][n][0], zraw[half][n][1], zraw[half][n][2], zraw[half][n][3]), z);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = os * (cs * v[e] - rs * z[e]);
          Vec16<E>::store(out + off, v);
        }
      }
    }
  }
}

template <typename E>
int launch_dz(const void* G, long g_pitch, const void* Y, const void* Z, void* out, const float* cscale, const float* rscale,
              const float* out_scale, int Bm, int Bn, long row_elems, hipStream_t st) {
  static unsigned long long attr_done = 0;        // per device
  auto kern = clip_dz_kernel<E>;
  if (first_use_on_device(attr_done)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, DZ_LDS) != hipSuccess) {
      set_error("clip_dz: cannot reserve %d bytes of LDS", DZ_LDS);
      return -3;
    }
  }
  const int cus = launch_cus();
  const int ntiles = (int)(row_elems / DZ_KT);
  const int jblocks = (Bn + 255) / 256;
  int gx = 2 * cus / jblocks;                          // two workgroups per CU in all
  if (gx < 1) gx = 1;
  if (gx > ntiles) gx = ntiles;
  hipLaunchKernelGGL(kern, dim3((unsigned)gx, (unsigned)jblocks), dim3(256), DZ_LDS, st, (const E*)G, g_pitch, (const E*)Y, (const E*)Z,
                     (E*)out, cscale, rscale, out_scale, Bm, Bn, row_elems, ntiles);
  return check_launch("clip_dz");
}

// ---------------------------------------------------------------------------------------------------------------------------
// The same gradient against MORE than 256 speech rows — a rank's block under data parallelism, where the contraction runs over
// the GLOBAL batch (2048 or 4096 rows at the 8-GPU configurations): 404 GFLOP against 2.0 GB at 2048 x 256 x 385 024, i.e. a
// matrix-core problem, and the coefficient matrix (1 MB) no longer fits a wave's registers.  One workgroup = 8 waves owns a
// 256 (columns j) x 256 (embedding elements k) tile of dZ and walks the speech rows in 32-row K-steps through sim_gemm's ring:
// four 32 KB LDS stages (the K-step's 32 x 512 B of Y and 32 x 512 B of G, both by LDS-DMA from a scalar base advanced once per
// K-step plus two per-lane offsets), three K-steps in flight behind counted waits, one raw barrier per K-step (32 MFMAs per
// wave).  Both operands are contracted over their ROW index, so all twelve fragments of a K-step are transposed reads
// (ds_read_b64_tr_b16; 512-byte image rows, 32-byte groups XOR-ed with row & 7: conflict-free).  D[k][j] leaves four
// consecutive k per lane; a 4 KB per-wave LDS patch turns 16 (j) x 64 (k) accumulator blocks into 128-byte row segments, so Z
// is read and dZ written in whole cache lines, each byte once.  Workgroups are persistent (one per CU, 160 KB of LDS: the ring
// and the patches side by side): the next tile's first three K-steps are requested BEFORE the epilogue of the current one.
// wgrad_gemm's typed-output mode (128 x 128 tiles, two stages, fp32 staging epilogue) ran this shape at 527 TFLOP/s.
constexpr int DT_TILE = 256;                          // columns j and embedding elements k per workgroup
constexpr int DT_RB = DT_TILE * 2;                    // image row bytes
constexpr int DT_OP = 32 * DT_RB;                     // 16 KB per operand and K-step
constexpr int DT_STAGE = 2 * DT_OP;                   // Y image, then G image
constexpr int DT_NS = 4;
constexpr int DT_PATCH = 16 * 64 * 4;                 // per wave: 16 rows j x 64 columns k, fp32
constexpr int DT_LDS = DT_NS * DT_STAGE + 8 * DT_PATCH;   // 163 840 B

template <typename E>
__global__ __launch_bounds__(512, 2) void clip_dz_tiles_kernel(const E* __restrict__ G, const long g_pitch, const E* __restrict__ Y,
                                                               const E* __restrict__ Z, E* __restrict__ out,
                                                               const float* __restrict__ cscale, const float* __restrict__ rscale,
                                                               const float* __restrict__ out_scale, const int Bm, const int Bn,
                                                               const long row_elems, const int ntiles) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wk = wid & 1, wj = wid >> 1;               // wave tile: k [wk * 128, +128) x j [wj * 64, +64)
  const int lr = lane & 15, lq = lane >> 4;
  const int j0 = blockIdx.y * DT_TILE;
  const int nks = Bm >> 5;                             // K-steps of 32 speech rows (Bm % 32 == 0: the launcher checks)

  // LDS-DMA pieces: one wave instruction = 2 image rows x 512 B, lane-linear; the bank swizzle goes on the SOURCE chunk.
  // Wave w fetches pieces w and w + 8 of either image.  Columns of G past its pitch (a tile wider than the matrix) are moved
  // inside: they feed outputs that are never stored.
  const int prow = lane >> 5, pchunk = lane & 31;
  uint32_t yoff[2], goff[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int r = (wid + 8 * h) * 2 + prow;
    const int sw = pchunk ^ chunk_xor<E, DT_RB>(r);
    long jc = (long)j0 + sw * 8;
    if (jc + 8 > g_pitch) jc = g_pitch - 8;
    yoff[h] = (uint32_t)(((size_t)r * row_elems + (size_t)sw * 8) * sizeof(E));
    goff[h] = (uint32_t)(((size_t)r * g_pitch + (size_t)jc) * sizeof(E));
  }
  const uint32_t lds_base = __builtin_amdgcn_readfirstlane(lds_addr(smem));
  const size_t ystep = (size_t)32 * row_elems, gstep = (size_t)32 * g_pitch;
  const E* ys = Y;
  const E* gs = G;
  auto issue = [&](int buf, int q) {                   // this wave's q-th piece of the K-step `ys` / `gs` point at, into stage `buf`
    const uint32_t dst = lds_base + buf * DT_STAGE + (wid + 8 * (q & 1)) * 1024;
    if (q < 2) lds_dma16_lean<false>(ys, yoff[q & 1], dst);
    else lds_dma16_lean<false>(gs, goff[q & 1], dst + DT_OP);
  };
  auto prologue = [&](int tile) {
    ys = Y + (size_t)tile * DT_TILE;
    gs = G;
#pragma unroll
    for (int p = 0; p < DT_NS - 1; ++p) {
      if (p < nks) {
#pragma unroll
        for (int q = 0; q < 4; ++q) issue(p, q);
        ys += ystep; gs += gstep;
      }
    }
  };
  // transposed fragment reads: per-lane offsets, fixed for the kernel (a K-step is one stage: row 0 of either image)
  uint32_t a_off[8], b_off[4];
#pragma unroll
  for (int a = 0; a < 8; ++a) a_off[a] = tr_offset_bf16<DT_RB>(0, wk * 128 + a * 16, lane);
#pragma unroll
  for (int b = 0; b < 4; ++b) b_off[b] = DT_OP + tr_offset_bf16<DT_RB>(0, wj * 64 + b * 16, lane);

  float* patch = reinterpret_cast<float*>(smem + DT_NS * DT_STAGE + wid * DT_PATCH);
  const float os = out_scale ? out_scale[0] : 1.f;
  const int jr = lane >> 3, kc = (lane & 7) * 8;       // epilogue pass: row of the patch (+ 8 on the second pass), 8 consecutive k

  int tile = blockIdx.x;
  if (tile < ntiles) prologue(tile);
  for (; tile < ntiles; tile += gridDim.x) {
    f32x4 acc[8][4];
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    int cur = 0;
    for (int s = 0; s < nks; ++s) {
      const int younger = nks - 1 - s;
      if (younger >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if (younger == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      const bool more = s + (DT_NS - 1) < nks;
      const int nxt = cur == 0 ? DT_NS - 1 : cur - 1;
      const unsigned char* img = smem + cur * DT_STAGE;
      uint4 bf[4], af[8];
#pragma unroll
      for (int b = 0; b < 4; ++b) bf[b] = tr_read_bf16<DT_RB>(img + b_off[b]);
#pragma unroll
      for (int a = 0; a < 8; ++a) af[a] = tr_read_bf16<DT_RB>(img + a_off[a]);
#pragma unroll
      for (int a = 0; a < 8; ++a) {
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = mma16<E>(af[a], bf[b], acc[a][b]);
        if ((a & 1) && more) issue(nxt, a >> 1);
      }
      if (more) { ys += ystep; gs += gstep; }
      cur = cur == DT_NS - 1 ? 0 : cur + 1;
    }
    __syncthreads();                                    // every wave has read its last fragments: the ring is free
    const int next = tile + (int)gridDim.x;
    if (next < ntiles) prologue(next);                  // lands while the epilogue below runs
    // ---- epilogue.  acc[a][b]: k = wk * 128 + a * 16 + 4 * lq + r, j = wj * 64 + b * 16 + lr.  Per (b, half): four fragments
    // -> patch[j = lr][64 k] (16-byte chunks XOR-ed with the row), then lane -> (row, 8 consecutive k): 16 bytes of Z in, 16 out
    const long kbase = (long)tile * DT_TILE + wk * 128;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
#pragma unroll
      for (int half = 0; half < 2; ++half) {
#pragma unroll
        for (int a4 = 0; a4 < 4; ++a4) {
          const int ch = (a4 * 4 + lq) ^ lr;
          *reinterpret_cast<f32x4*>(patch + lr * 64 + ch * 4) = acc[half * 4 + a4][b];
        }
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
          const int row = jr + 8 * pass;
          const int j = j0 + wj * 64 + b * 16 + row;
          const f32x4 lo = *reinterpret_cast<const f32x4*>(patch + row * 64 + (((kc >> 2)) ^ row) * 4);
          const f32x4 hi = *reinterpret_cast<const f32x4*>(patch + row * 64 + (((kc >> 2) + 1) ^ row) * 4);
          if (j < Bn) {
            const size_t off = (size_t)j * row_elems + kbase + half * 64 + kc;
            const float cs = cscale ? cscale[j] : 1.f, rs = rscale[j];
            float z[8], v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            Vec16<E>::load(Z + off, z);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = os * (cs * v[e] - rs * z[e]);
            Vec16<E>::store(out + off, v);
          }
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the next tile's first K-steps (and this epilogue's own traffic)
  }
}

template <typename E>
int launch_dz_tiles(const void* G, long g_pitch, const void* Y, const void* Z, void* out, const float* cscale, const float* rscale,
                    const float* out_scale, int Bm, int Bn, long row_elems, hipStream_t st) {
  static unsigned long long attr_done = 0;        // per device
  auto kern = clip_dz_tiles_kernel<E>;
  if (first_use_on_device(attr_done)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, DT_LDS) != hipSuccess) {
      set_error("clip_dz: cannot reserve %d bytes of LDS", DT_LDS);
      return -3;
    }
  }
  const int ntiles = (int)(row_elems / DT_TILE);
  const int jblocks = (Bn + DT_TILE - 1) / DT_TILE;
  int gx = launch_cus() / jblocks;                     // one persistent workgroup per CU in all
  if (gx < 1) gx = 1;
  if (gx > ntiles) gx = ntiles;
  hipLaunchKernelGGL(kern, dim3((unsigned)gx, (unsigned)jblocks), dim3(512), DT_LDS, st, (const E*)G, g_pitch, (const E*)Y, (const E*)Z,
                     (E*)out, cscale, rscale, out_scale, Bm, Bn, row_elems, ntiles);
  return check_launch("clip_dz(tiles)");
}

// shapes of the tiled form: whole 32-row K-steps, whole 256-element column tiles, per-lane offsets inside 32 bits
static bool dz_tiles_ok(int Bm, int Bn, long row_elems, int dtype) {
  // (SDA_DZ_TILES_MIN=<rows> in the environment moves the hand-over between the two forms: diagnostics)
  static const int min_rows = [] { const char* e = getenv("SDA_DZ_TILES_MIN"); return e ? atoi(e) : DZ_ROWS; }();
  return (dtype == SDA_BF16 || dtype == SDA_F16) && Bm >= min_rows && Bm % 32 == 0 && Bn >= 1 && row_elems >= DT_TILE &&
         row_elems % DT_TILE == 0 && row_elems / DT_TILE < 0x7fffffffL && 32L * row_elems * 2 < (1L << 32);
}

}  // namespace
}  // namespace sda

using namespace sda;

extern "C" int sda_clip_dz_supported(int Bm, int Bn, long row_elems, int dtype) {
  if (dz_tiles_ok(Bm, Bn, row_elems, dtype)) return 1;
  return (dtype == SDA_BF16 || dtype == SDA_F16) && Bm >= 1 && Bm <= DZ_ROWS && Bn >= 1 && row_elems >= DZ_KT && row_elems % DZ_KT == 0 &&
         row_elems / DZ_KT < 0x7fffffffL;
}

extern "C" int sda_clip_dz(const void* G, long g_pitch, const void* Y, const void* Z, void* out, const float* cscale,
                           const float* rscale, const float* out_scale, int Bm, int Bn, long row_elems, int dtype, void* stream) {
  if (!G || !Y || !Z || !out || !rscale || g_pitch < Bn) { set_error("clip_dz: bad arguments"); return -1; }
  if (!sda_clip_dz_supported(Bm, Bn, row_elems, dtype)) {
    set_error("clip_dz: needs a 16-bit dtype and either Bm <= %d with row_elems %% %d == 0, or Bm %% 32 == 0 with row_elems %% %d == 0 "
              "(use sda_wgrad_gemm's typed output otherwise)", DZ_ROWS, DZ_KT, DT_TILE);
    return -1;
  }
  hipStream_t st = (hipStream_t)stream;
  if (dz_tiles_ok(Bm, Bn, row_elems, dtype)) {
    if (g_pitch % 8 || g_pitch < 8) { set_error("clip_dz: the coefficient matrix's pitch must be a multiple of 8 elements"); return -1; }
    if (dtype == SDA_BF16) return launch_dz_tiles<uint16_t>(G, g_pitch, Y, Z, out, cscale, rscale, out_scale, Bm, Bn, row_elems, st);
    return launch_dz_tiles<half_t>(G, g_pitch, Y, Z, out, cscale, rscale, out_scale, Bm, Bn, row_elems, st);
  }
  if (dtype == SDA_BF16) return launch_dz<uint16_t>(G, g_pitch, Y, Z, out, cscale, rscale, out_scale, Bm, Bn, row_elems, st);
  return launch_dz<half_t>(G, g_pitch, Y, Z, out, cscale, rscale, out_scale, Bm, Bn, row_elems, st);
}

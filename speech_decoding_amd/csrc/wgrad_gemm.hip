// wgrad_gemm — contraction over ROWS (time) of two channels-last tensors on MFMA:
//     g[seg][tap][co][ci] = sum_{b in seg} sum_t dy[b, t, co] * x[b, t + (tap - KS/2) * dil, ci]
// Reference ops this replaces: the weight gradients autograd derives for every nn.Conv1d on the path
// (models.py:97-109,128-150,188-189), for SpatialAttention's mix (models.py:65) and — in typed-output
// mode — the embedding gradient of the similarity matmul (loss.py:68): dZ = G^T Y - diag(r) Z.
//
// Both operands are contracted over their slow (row) index, so the MFMA fragments are TRANSPOSED reads
// of row-major LDS images: bf16 uses ds_read_b64_tr_b16 (two per 16x16x32 operand; lane group g takes
// rows {4g..4g+3} and {16+4g..16+4g+3} for BOTH operands, which keeps a 32-lane half on 8 consecutive
// rows = 8 distinct 32-byte bank segments with a row stride = 32 (mod 64) bytes); fp32 uses ds_read_b32
// with row stride = 16 (mod 32) words.  One workgroup = 4 waves (2 x 2) owns a TILE_M x 64 output tile for
// all KS taps and streams its segment's samples in 64-row chunks; the three taps read the same staged x
// image at row offsets tap*dil.  Segments make the K split explicit: per-subject weight gradients use one
// segment per subject (final result, no reduction), shared weights use ~CU-count segments + reduce_slabs.
#include "sd_common.h"
#include "tr_operand.h"

#include <type_traits>
#include <utility>

namespace sda {

// One LDS-DMA piece, lean form (conv3_flat.hip's): wave-uniform 64-bit base the caller computed with scalar arithmetic well
// ahead (no VALU-written SGPR feeds the load: no s_nop 4), one per-lane byte offset register, M0 written in the statement
// that reads it.  tools/check_dma_hazard.py walks the listing for hazards.
__device__ __forceinline__ void wg_dma16_lean(const void* sbase, uint32_t voff, uint32_t lds_dst) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

namespace {
// compile-time loop: f(std::integral_constant<int, I>) for I in [B, E)
template <int B, int E, typename F> __device__ __forceinline__ void wg_static_for(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    wg_static_for<B + 1, E>(f);
  }
}
}  // namespace

// Rows per MFMA K-step: 32 (bf16) / 16 (fp32).  A staged K-chunk holds KM of them.
template <typename E> struct WK;
template <> struct WK<uint16_t> { static constexpr int KSTEP = 32; };
template <> struct WK<half_t> { static constexpr int KSTEP = 32; };
template <> struct WK<float> { static constexpr int KSTEP = 16; };

typedef __attribute__((address_space(1))) const void gmem_cv;
typedef __attribute__((address_space(3))) void lds_v;

// TN = ci columns per workgroup: 64 for KS = 3 (three accumulator sets), 128 for KS = 1 (wave tile
// TILE_M/2 x 64: fewer LDS bytes per MFMA and half as many re-reads of dy).
// KM = MFMA K-steps per staged chunk, NS = LDS stages (NS - 1 chunks of LDS-DMA in flight behind the MFMAs).
template <typename E, int TILE_M, int KS, int TN, int KM, int NS> struct WGeom {
  static constexpr int KT = WK<E>::KSTEP * KM;
  static constexpr int RB_M = TILE_M * (int)sizeof(E);          // dy image row bytes
  static constexpr int RB_N = TN * (int)sizeof(E);              // x image row bytes
  static constexpr int DY_BYTES = KT * RB_M;
  static constexpr int XR = KT + (KS == 3 ? 2 * PAD : 0);
  static constexpr int X_BYTES = XR * RB_N;
  static constexpr int STAGE = DY_BYTES + X_BYTES;
  static constexpr int DY_PIECES = DY_BYTES / 1024;
  static constexpr int EPI_BYTES = TILE_M * (TN + 4) * 4;
  static constexpr int LDS = (NS * STAGE > EPI_BYTES) ? NS * STAGE : EPI_BYTES;
  static_assert(DY_BYTES % 1024 == 0 && X_BYTES % 1024 == 0, "images must be whole 1 KB pieces");
};

template <typename E, int TILE_M, int KS, int TN, int KM, int NS>
__global__ __launch_bounds__(256, (WGeom<E, TILE_M, KS, TN, KM, NS>::LDS > 80 * 1024) ? 1 : 2) void wgrad_gemm_kernel(const sda_wgrad_args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  using G = WGeom<E, TILE_M, KS, TN, KM, NS>;
  constexpr int WG_TN = TN;
  constexpr int NREP = TN / 32;                             // 16-column n tiles per wave (wave tile = TILE_M/2 x TN/2)
  constexpr int PER16 = Elem<E>::PER16;
  constexpr int KSTEP = WK<E>::KSTEP, KT = G::KT;
  constexpr int MREP = TILE_M / 32;                         // 16-row m tiles per wave (wave tile = TILE_M/2 x 32)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave_m = wid >> 1, wave_n = wid & 1;
  const int lr = lane & 15, lq = lane >> 4;

  int bid = blockIdx.x;
  const int n_n = a.Cin_p / WG_TN, n_m = a.Cout_p / TILE_M;
  // XCD-aware order (blocks i and i+8 share an XCD/L2): all output tiles of one sample segment run on the
  // same XCD back to back, so the dy / x rows they all stream are fetched from HBM once and re-read from
  // that XCD's L2 (measured: 5x fewer fabric bytes than the tile-fastest order).  Pure speed.
  int seg, tile;
  {
    const int ntile = n_n * n_m, group = 8 * ntile, full = (int)(gridDim.x / group) * group;
    if (bid < full) {
      const int base = bid / group, rem = bid - base * group;
      seg = base * 8 + (rem & 7);
      tile = rem >> 3;
    } else {
      const int rem = bid - full;
      seg = full / ntile + rem / ntile;
      tile = rem % ntile;
    }
  }
  int n_tile = tile % n_n, m_tile = tile / n_n;
  if (a.nseg == 1 && n_m > 1) {
    // one segment (the typed-output matrix mode: dZ = G^T Y with thousands of column tiles): the n_m row tiles of a column
    // tile stream the same x rows — deal them to the same XCD side by side (blocks i and i + 8), so those rows come from HBM
    // once and from that XCD's L2 the other n_m - 1 times (tile-major order sent every x row to HBM n_m times, a whole
    // pass of the grid apart)
    const int grp = 8 * n_m, full2 = (int)(gridDim.x / grp) * grp;
    bid = blockIdx.x;
    if (bid < full2) {
      const int base = bid / grp, rem = bid - base * grp;
      n_tile = base * 8 + (rem & 7);
      m_tile = rem >> 3;
    } else {
      const int rem = bid - full2;
      n_tile = full2 / n_m + rem / n_m;
      m_tile = rem % n_m;
    }
  }
  const int co0 = m_tile * TILE_M, ci0 = n_tile * WG_TN;
  const int halo = (KS == 3) ? a.dil : 0;
  const int x_pieces = ((KT + 2 * halo) * G::RB_N + 1023) >> 10;

  f32x4 acc[KS][MREP][NREP];
#pragma unroll
  for (int k = 0; k < KS; ++k)
#pragma unroll
    for (int m = 0; m < MREP; ++m)
#pragma unroll
      for (int n = 0; n < NREP; ++n) acc[k][m][n] = f32x4{0, 0, 0, 0};

  const E* __restrict__ dyg = reinterpret_cast<const E*>(a.dy);
  const E* __restrict__ xg = reinterpret_cast<const E*>(a.x);
  const int s_beg = a.seg_start ? a.seg_start[seg] : 0;
  const int s_end = a.seg_start ? a.seg_start[seg + 1] : a.B;
  const int nchunk_plain = (a.T + KT - 1) / KT;
  // Two ways to cut a segment's rows into K-chunks of KT rows:
  //   per sample (the general form; `perm` may reorder samples): chunk it = (sample it / nchunk, rows [ch * KT, ch * KT + KT) of it);
  //     a sample's last chunk is partial (T = 360: 5 x 64 + 40) and takes the per-lane staging path;
  //   flat (SDA_WGRAD_FLAT_ROWS, consecutive samples): the segment is ONE run of rows — the valid rows of its samples and the
  //     layout's pad rows between them, which are zero in dy and therefore contribute nothing (they are the zero padding a
  //     "same" convolution reads, models.py:128-150) — cut into KT-row chunks regardless of sample boundaries: only the
  //     segment's very last chunk is partial, every other one takes the scalar-base staging path, and 376 / 360 rows are
  //     contracted per sample instead of 384 / 360.
  const bool flat = (a.flags & SDA_WGRAD_FLAT_ROWS) && !a.perm && s_end > s_beg;
  // With a sample permutation (per-subject segments: a segment's samples are not neighbours in memory) the same guarantee —
  // dy is zero on the pad rows around every sample — lets each SAMPLE be contracted as whole chunks: its run of T rows is
  // extended by `lead` rows of its own leading padding and the rest into the next sample's (T = 360: 16 + 360 + 8 = six
  // 64-row chunks, none partial: no per-lane staging path, 1 chunk in 6 before).
  const int pad_need = (KT - a.T % KT) % KT;
  const bool padded = (a.flags & SDA_WGRAD_FLAT_ROWS) && a.perm && pad_need <= 2 * PAD;
  const int pad_lead = padded ? (pad_need < PAD ? pad_need : PAD) : 0;
  const long seg_r0 = a.row0 + (long)s_beg * a.sample_rows;                               // first row of the run
  const long seg_r1 = a.row0 + (long)(s_end - 1) * a.sample_rows + a.T;                   // one past its last valid row
  const int nchunk = padded ? (a.T + pad_need) / KT : nchunk_plain;
  const int total = flat ? (int)((seg_r1 - seg_r0 + KT - 1) / KT) : (s_end - s_beg) * nchunk;

  // one K-chunk = KT rows.  DMA pieces are 1 KB, lane-linear in LDS; each lane derives the (row, chunk) its 16 bytes
  // belong to and fetches the swizzle-matched source chunk.  dy rows past the chunk's valid ones come from a
  // guaranteed-zero row (they must not contribute); x rows are merely clamped into the buffer.
  // (the sample index is read from `perm` once per SAMPLE: a global load in the chunk loop is waited for with vmcnt(0))
  // Chunks are visited in order, so (sample, chunk-in-sample) advance incrementally: no division in the loop.
  int cur_si = 0, cur_ch = -1, cur_it = -1, staged_b = 0;
  long c_row = 0;                 // first dy row of the chunk described last
  int c_valid = 0;                // its number of valid rows (<= KT)
  auto describe = [&](int it) {   // it == cur_it + 1 (or cur_it: idempotent)
    if (it == cur_it) return;
    cur_it = it;
    if (flat) {
      c_row = seg_r0 + (long)it * KT;
      const long left = seg_r1 - c_row;
      c_valid = left < KT ? (int)left : KT;
      return;
    }
    if (++cur_ch == nchunk || it == 0) {
      if (it != 0) { cur_ch = 0; ++cur_si; }
      staged_b = a.perm ? __builtin_amdgcn_readfirstlane(a.perm[s_beg + cur_si]) : (s_beg + cur_si);
    }
    const int t0 = cur_ch * KT - pad_lead;
    c_row = a.row0 + (long)staged_b * a.sample_rows + t0;
    c_valid = padded ? KT : (a.T - t0 < KT ? a.T - t0 : KT);
  };
  auto stage = [&](int it, int buf) {
    describe(it);
    unsigned char* dys = smem + buf * G::STAGE;
    unsigned char* xs = dys + G::DY_BYTES;
#pragma unroll
    for (int i = 0; i < (G::DY_PIECES + 3) / 4; ++i) {
      const int p = wid + i * 4;
      if (p < G::DY_PIECES) {
        const int byte = p * 1024 + lane * 16;
        const int r = byte / G::RB_M;
        const int c = ((byte - r * G::RB_M) >> 4) ^ chunk_xor<E, G::RB_M>(r);
        const long row = (r < c_valid) ? (c_row + r) : a.dy_zero_row;
        lds_dma16(dyg + (size_t)row * a.dy_pitch + co0 + c * PER16,
                  __builtin_amdgcn_readfirstlane(lds_addr(dys + p * 1024)));
      }
    }
    for (int p = wid; p < x_pieces; p += 4) {
      const int byte = p * 1024 + lane * 16;
      const int r = byte / G::RB_N;
      const int c = ((byte - r * G::RB_N) >> 4) ^ chunk_xor<E, G::RB_N>(r);
      long row = c_row - halo + r;
      row = row < 0 ? 0 : (row >= a.rows_limit ? a.rows_limit - 1 : row);
      lds_dma16(xg + (size_t)row * a.x_pitch + ci0 + c * PER16,
                  __builtin_amdgcn_readfirstlane(lds_addr(xs + p * 1024)));
    }
  };

  // Fast staging of whole chunks.  Inside a chunk the (row, 16-byte chunk) a lane fetches for the wave's i-th piece never
  // changes, so its byte offset is computed ONCE; per chunk only a wave-uniform base moves (first row of the chunk), and a
  // piece is a scalar base + one offset register (lds_dma16_sv) — no per-lane address arithmetic left in the loop.  That
  // makes an issue cheap enough to sit BETWEEN the MFMA groups of the chunk being consumed (it costs the wave's issue slot
  // while the matrix pipe drains the MFMAs queued before it) instead of in front of them.  Chunks that run past the
  // sample's T rows (dy must read the zero row there) or past the buffer keep the per-lane path, issued up front.
  constexpr int NDY = (G::DY_PIECES + 3) / 4;
  constexpr int NX = (G::X_BYTES / 1024 + 3) / 4;
  uint32_t vdy[NDY], vx[NX];
#pragma unroll
  for (int i = 0; i < NDY; ++i) {
    const int byte = (wid + 4 * i) * 1024 + lane * 16;
    const int r = byte / G::RB_M;
    const int c = ((byte - r * G::RB_M) >> 4) ^ chunk_xor<E, G::RB_M>(r);
    vdy[i] = (uint32_t)(((size_t)r * a.dy_pitch + (size_t)c * PER16) * sizeof(E));
  }
#pragma unroll
  for (int i = 0; i < NX; ++i) {
    const int byte = (wid + 4 * i) * 1024 + lane * 16;
    const int r = byte / G::RB_N;
    const int c = ((byte - r * G::RB_N) >> 4) ^ chunk_xor<E, G::RB_N>(r);
    vx[i] = (uint32_t)(((size_t)r * a.x_pitch + (size_t)c * PER16) * sizeof(E));
  }
  const uint32_t lds_base = __builtin_amdgcn_readfirstlane(lds_addr(smem));
  const E* dy_base = dyg;            // wave-uniform bases of the chunk being staged (fast path)
  const E* x_base = xg;
  uint32_t nxt_off = 0;
  const uint32_t lds_w = lds_base + (uint32_t)wid * 1024u;   // this wave's share of a piece index: a destination is lds_w + stage + a constant
  // chunk `it` -> is it a fast one?  sets the bases; the slow path stages it at once
  // Flat rows: chunk `it` starts KT rows behind chunk it - 1, so its two bases are running pointers advanced by a constant
  // stride per chunk, and "is it a fast one" is one compare — the leading n_fast chunks are (full chunk, input rows inside the
  // buffer): the general form below recomputes rows, bounds and two 64-bit products per chunk, ~70 scalar instructions at the
  // top of every chunk, in front of its first fragment reads.
  int n_fast = 0;
  const E* dy_run = dyg;
  const E* x_run = xg;
  const size_t dy_stride = (size_t)KT * a.dy_pitch, x_stride = (size_t)KT * a.x_pitch;
  if (flat && seg_r0 >= halo) {
    const long n_full = (seg_r1 - seg_r0) / KT, n_ub = (a.rows_limit - halo - seg_r0) / KT;
    n_fast = (int)(n_full < n_ub ? n_full : n_ub);
    if (n_fast < 0) n_fast = 0;
    dy_run = dyg + (size_t)(seg_r0 + KT) * a.dy_pitch + co0;              // chunk 1 (chunk 0 is staged in front of the loop)
    x_run = xg + (size_t)(seg_r0 + KT - halo) * a.x_pitch + ci0;
  }
  auto prepare = [&](int it, int buf) -> bool {
    if (flat) {
      const bool f = it < n_fast;
      if (f) { dy_base = dy_run; x_base = x_run; nxt_off = (uint32_t)(buf * G::STAGE); }
      dy_run += dy_stride;
      x_run += x_stride;
      if (f) return true;
      stage(it, buf);
      return false;
    }
    describe(it);
    const long xrow0 = c_row - halo;
    if (c_valid == KT && xrow0 >= 0 && xrow0 + KT + 2 * halo <= a.rows_limit) {
      dy_base = dyg + (size_t)c_row * a.dy_pitch + co0;
      x_base = xg + (size_t)xrow0 * a.x_pitch + ci0;
#ifdef SDA_WGRAD_FAKE_SRC      /* diagnostic build (garbage results): every chunk re-reads the segment's first rows — no memory-system load */
      dy_base = dyg + (size_t)seg_r0 * a.dy_pitch + co0;
      x_base = xg + (size_t)seg_r0 * a.x_pitch + ci0;
#endif
      nxt_off = (uint32_t)(buf * G::STAGE);
      return true;
    }
    stage(it, buf);
    return false;
  };
  auto issue_piece = [&](auto jc) {            // j-th piece of this wave of the chunk prepared last
    constexpr int j = decltype(jc)::value;
    if constexpr (j < NDY) {
      const int pc = wid + 4 * j;
      if (pc < G::DY_PIECES) wg_dma16_lean(dy_base, vdy[j], lds_w + nxt_off + (uint32_t)(j * 4096));
    } else if constexpr (j < NDY + NX) {
      const int pc = wid + 4 * (j - NDY);
      if (pc < x_pieces) wg_dma16_lean(x_base, vx[j - NDY], lds_w + nxt_off + (uint32_t)(G::DY_BYTES + (j - NDY) * 4096));
    }
  };
  constexpr int NPW = NDY + NX;                // pieces per wave and chunk (upper bound)
  constexpr int NGRP = (KT / KSTEP) * KS;      // MFMA groups per chunk
#ifndef SDA_WGRAD_FRONT
#define SDA_WGRAD_FRONT 1
#endif
  // pieces per MFMA group: SDA_WGRAD_FRONT = 1 spreads them over all groups of the chunk, 2 over its first half, 3 over its
  // first third (more time to land before the next chunk's wait; an issue is three scalar-side instructions now)
  constexpr int PER_GRP = (NPW * SDA_WGRAD_FRONT + NGRP - 1) / NGRP;

  // 16-bit types: per-lane offsets of every transposed fragment read of a stage, computed once (tr_operand.h); a read of
  // K-step kk is then `stage base + offset` plus the immediate kk * KSTEP rows
  constexpr bool FASTTR = sizeof(E) == 2;
  uint32_t a_off[FASTTR ? MREP : 1], b_off[FASTTR ? KS : 1][FASTTR ? NREP : 1];
  if constexpr (FASTTR) {
#pragma unroll
    for (int m = 0; m < MREP; ++m) a_off[m] = tr_offset_bf16<G::RB_M>(0, wave_m * (TILE_M / 2) + m * 16, lane);
#pragma unroll
    for (int tap = 0; tap < KS; ++tap)
#pragma unroll
      for (int n = 0; n < NREP; ++n) b_off[tap][n] = G::DY_BYTES + tr_offset_bf16<G::RB_N>(tap * a.dil, wave_n * (TN / 2) + n * 16, lane);
  }
  static_assert(NS == 2, "the chunk loop below is written for two LDS stages");
  if (0 < total) stage(0, 0);
  int cur = 0;
  for (int it = 0; it < total; ++it) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // own pieces of chunk `it` landed
    __builtin_amdgcn_s_barrier();                       // everyone's landed; chunk it-1 fully consumed
    const unsigned char* dys = smem + cur * G::STAGE;
    const int nxt = cur ^ 1;
    cur ^= 1;
    const unsigned char* xs = dys + G::DY_BYTES;
    // MFMA groups g = (kk, tap) of the chunk, software-pipelined IN THE SOURCE: group g + 1's x fragments (and, on a K-step's
    // last tap, the next K-step's dy fragments) are read BEFORE group g's MFMAs.  hipcc cannot do this itself — the LDS-DMA
    // asm statements between the groups are memory barriers to it — and without it every group began with its own fragment
    // reads and a full wait for them: six exposed LDS round trips per chunk on a SIMD that holds one wave of this kernel.
    constexpr int NKK = KT / KSTEP, NG = NKK * KS;
    auto read_a = [&](auto kc, uint4 (&af)[MREP]) {
      constexpr int kk = decltype(kc)::value;
#pragma unroll
      for (int m = 0; m < MREP; ++m) {
        if constexpr (FASTTR) af[m] = tr_read_bf16<G::RB_M>(dys + a_off[m] + kk * KSTEP * G::RB_M);
        else af[m] = TrOp<E, G::RB_M>::get(dys, kk * KSTEP, wave_m * (TILE_M / 2) + m * 16, lane);
      }
    };
    auto read_b = [&](auto gc, uint4 (&bf)[NREP]) {
      constexpr int g = decltype(gc)::value, kk = g / KS, tap = g % KS;
#pragma unroll
      for (int n = 0; n < NREP; ++n) {
        if constexpr (FASTTR) bf[n] = tr_read_bf16<G::RB_N>(dys + b_off[tap][n] + kk * KSTEP * G::RB_N);
        else bf[n] = TrOp<E, G::RB_N>::get(xs, kk * KSTEP + tap * a.dil, wave_n * (TN / 2) + n * 16, lane);
      }
    };
    // (kernel size 3 only: the 1 x 1 instantiations have no registers for a second fragment set — <160, 1, 128> would
    // drop to one wave per SIMD — and read each group's fragments at its start, as before)
    constexpr bool PIPE = KS == 3;
    uint4 af[MREP], bf[NREP];
    read_a(std::integral_constant<int, 0>{}, af);
    read_b(std::integral_constant<int, 0>{}, bf);
    // (the next chunk is prepared AFTER this chunk's first fragment reads are on their way: its scalar work hides behind them)
    const bool fast = (it + 1 < total) && prepare(it + 1, nxt);
    wg_static_for<0, NG>([&](auto gc) {
      constexpr int g = decltype(gc)::value, kk = g / KS, tap = g % KS;
      uint4 af_n[MREP], bf_n[NREP];
      if constexpr (!PIPE && g > 0) {
        if constexpr (tap == 0) read_a(std::integral_constant<int, kk>{}, af);
        read_b(gc, bf);
      }
      if constexpr (PIPE && g + 1 < NG) read_b(std::integral_constant<int, g + 1>{}, bf_n);
      if constexpr (PIPE && tap == KS - 1 && kk + 1 < NKK) read_a(std::integral_constant<int, kk + 1>{}, af_n);
      // (hipcc sinks the last two groups' reads below the MFMAs of the group before them; pinning them with a scheduling
      // barrier — 235 registers, every group's fragments in flight a group ahead — measured 7.05-7.08 vs 7.02-7.09 ms: not kept)
      mma16_block<E, MREP, NREP>(af, bf, acc[tap]);
      if (fast) {
        wg_static_for<g * PER_GRP, (g + 1) * PER_GRP < NPW ? (g + 1) * PER_GRP : NPW>(issue_piece);
      }
      if constexpr (PIPE && g + 1 < NG) {
#pragma unroll
        for (int n = 0; n < NREP; ++n) bf[n] = bf_n[n];
      }
      if constexpr (PIPE && tap == KS - 1 && kk + 1 < NKK) {
#pragma unroll
        for (int m = 0; m < MREP; ++m) af[m] = af_n[m];
      }
    });
  }

  if (!a.out_e) {
    float* __restrict__ gp = a.g + (size_t)seg * KS * a.Cout_p * a.Cin_p;
#pragma unroll
    for (int tap = 0; tap < KS; ++tap)
#pragma unroll
      for (int m = 0; m < MREP; ++m)
#pragma unroll
        for (int n = 0; n < NREP; ++n)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int co = co0 + wave_m * (TILE_M / 2) + m * 16 + lq * 4 + r;
            const int ci = ci0 + wave_n * (TN / 2) + n * 16 + lr;
            gp[((size_t)tap * a.Cout_p + co) * a.Cin_p + ci] = acc[tap][m][n][r];
          }
    return;
  }

  // typed output through LDS: out[co][ci] = out_scale * (acc_scale[co] * acc - rscale[co] * sub[co][ci])   (KS == 1)
  constexpr int EP_STRIDE = WG_TN + 4;
  float* ep = reinterpret_cast<float*>(smem);
  __syncthreads();
#pragma unroll
  for (int m = 0; m < MREP; ++m)
#pragma unroll
    for (int n = 0; n < NREP; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        ep[(wave_m * (TILE_M / 2) + m * 16 + lq * 4 + r) * EP_STRIDE + wave_n * (TN / 2) + n * 16 + lr] = acc[0][m][n][r];
  __syncthreads();
  E* __restrict__ og = reinterpret_cast<E*>(a.out_e);
  const E* __restrict__ sg = reinterpret_cast<const E*>(a.sub);
  for (int idx = tid; idx < TILE_M * (WG_TN / 4); idx += 256) {
    const int row = idx / (WG_TN / 4), c4 = idx - row * (WG_TN / 4);
    const int co = co0 + row;
    if (co >= a.co_valid) continue;
    float4 v = *reinterpret_cast<const float4*>(ep + row * EP_STRIDE + c4 * 4);
    const size_t off = (size_t)co * a.out_pitch + ci0 + c4 * 4;
    if (a.acc_scale) { const float cs = a.acc_scale[co]; v.x *= cs; v.y *= cs; v.z *= cs; v.w *= cs; }
    if (sg) {
      const float rs = a.rscale[co];
      const float4 s = load4(sg + off);
      v.x -= rs * s.x; v.y -= rs * s.y; v.z -= rs * s.z; v.w -= rs * s.w;
    }
    if (a.out_scale) { const float k = a.out_scale[0]; v.x *= k; v.y *= k; v.z *= k; v.w *= k; }
    store4(og + off, v);
  }
}

template <typename E, int TILE_M, int KS, int TN, int KM = 2, int NS = 2>
static int launch_wgrad(const sda_wgrad_args& a, hipStream_t st) {
  using WG_ = WGeom<E, TILE_M, KS, TN, KM, NS>;
  constexpr int lds_max = WG_::LDS;
  // the fp32 staging of the typed epilogue only where there is one: a slab-output launch that asks for less LDS leaves
  // room on the CU for the other stream's workgroups
  const int lds = a.out_e ? lds_max : NS * WG_::STAGE;
  static unsigned long long attr_done = 0;        // per device
  auto kern = wgrad_gemm_kernel<E, TILE_M, KS, TN, KM, NS>;
  if (first_use_on_device(attr_done)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                            lds_max) != hipSuccess) {
      set_error("wgrad_gemm: cannot reserve %d bytes of LDS", lds_max);
      return -3;
    }
  }
  const long grid = (long)(a.Cin_p / TN) * (a.Cout_p / TILE_M) * a.nseg;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, st, a);
  return check_launch("wgrad_gemm");
}

template <typename E, int TILE_M>
static int dispatch_wgrad_m(const sda_wgrad_args& a, hipStream_t st) {
  // KM = 2 K-steps per chunk, NS = 2 stages: deeper pipelines (NS = 3, 4) and one-step chunks were measured
  // slower at one and at two workgroups per CU
  if (a.KS == 3) return launch_wgrad<E, TILE_M, 3, 64>(a, st);
  if (a.Cin_p % 128 == 0) return launch_wgrad<E, TILE_M, 1, 128>(a, st);
  return launch_wgrad<E, TILE_M, 1, 64>(a, st);
}

template <typename E>
static int dispatch_wgrad(const sda_wgrad_args& a, hipStream_t st) {
  if (a.Cout_p % 160 == 0) return dispatch_wgrad_m<E, 160>(a, st);
  if (a.Cout_p % 128 == 0) return dispatch_wgrad_m<E, 128>(a, st);
  return dispatch_wgrad_m<E, 64>(a, st);
}

}  // namespace sda

using namespace sda;

extern "C" int sda_wgrad_gemm(const sda_wgrad_args* a, void* stream) {
  if (!a || !a->dy || !a->x || (!a->g && !a->out_e)) { set_error("wgrad_gemm: null argument"); return -1; }
  if (a->KS != 1 && a->KS != 3) { set_error("wgrad_gemm: kernel size %d not supported", a->KS); return -1; }
  if (a->dil < 0 || a->dil > PAD) { set_error("wgrad_gemm: dilation %d outside [0, %d]", a->dil, PAD); return -1; }
  if (a->Cout_p % 64 || a->Cin_p % 64) { set_error("wgrad_gemm: channel extents must be multiples of 64"); return -1; }
  if (a->dy_pitch % 8 || a->x_pitch % 8) { set_error("wgrad_gemm: pitches must be multiples of 8 elements"); return -1; }
  if (a->nseg < 1 || a->B < 1 || a->T < 1) { set_error("wgrad_gemm: empty problem"); return -1; }
  if (a->out_e && (a->KS != 1 || a->nseg != 1 || a->out_pitch % 8)) { set_error("wgrad_gemm: typed output needs KS == 1, nseg == 1"); return -1; }
  if (a->dy_zero_row < 0) { set_error("wgrad_gemm: dy_zero_row missing"); return -1; }
  if (a->out_e && a->sub && !a->rscale) { set_error("wgrad_gemm: sub needs rscale"); return -1; }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (a->dtype == SDA_F32) return dispatch_wgrad<float>(*a, st);
  if (a->dtype == SDA_BF16) return dispatch_wgrad<uint16_t>(*a, st);
  if (a->dtype == SDA_F16) return dispatch_wgrad<half_t>(*a, st);
  set_error("wgrad_gemm: unknown dtype %d", a->dtype);
  return -1;
}

// conv3_flat — the kernel-3 dilated Conv1d of conv_gemm.hip on 256/128-row x 160-channel tiles of the FLAT row space,
// one persistent workgroup per run of rows.
//
// Reference ops: nn.Conv1d(kernel_size=3, dilation=d, padding="same") of ConvBlock (models.py:128-150) and its
// input gradient; same sda_conv_args contract as conv_gemm (bias, residual, BatchNorm statistics, bn_x mode).
//
// Why a second kernel.  conv_gemm's K loop is bound by the bytes a CU pulls from L2 into LDS (three quarters of them
// weight slabs that every workgroup re-reads), its two-tile form, which halves them, owns the CU's whole
// LDS, and either way all workgroups of a round reach their epilogue together: the residual read + output write
// (HBM-bound, ~1/3 of the kernel) then runs with every matrix core idle and the main loops with HBM idle.  Here
//   * a 256-row tile = 4 waves (2 x 2) x (128 rows x 80 channels: 8 x 5 MFMA fragments, 160 accumulator registers):
//     the L2->LDS bytes per FLOP of the two-tile form (3MN/(M+3N) = 167 FLOP/B against 101) and 0.33 instead of 0.45
//     LDS fragment reads per MFMA;
//   * LDS per workgroup stays under 80 KB, so TWO workgroups share a CU: the input slab (288 rows x 64 B) is
//     double-buffered per K-step, the weights travel as per-TAP slabs (160 rows x 64 B) through a 4-slot ring —
//     2 x 18 KB + 4 x 10 KB = 76 KB; one barrier per tap phase (40 MFMAs per wave); every wave issues its share of the
//     next K-step's LDS-DMA pieces 4 per phase, in need order, so ONE counted `s_waitcnt vmcnt(N)` per phase retires
//     what the phase reads while the youngest pieces stay in flight across the raw s_barrier;
//   * tiles are cut from the flat row space (all samples back to back, 16 zero rows in front of each; rows that are a
//     sample's padding are computed and never stored): no per-sample remainder tiles;
//   * the grid is ONE round of persistent workgroups (at most two per CU), each owning a run of consecutive 128-row
//     units of one channel tile and working through it as 256-row tiles plus at most two 128-row tiles.  The two
//     workgroups that share a CU take their tiles in a different order (128 first / 128 last), so they reach their
//     epilogues at different times: one's HBM traffic runs under the other's MFMAs instead of both idling the matrix
//     cores together, and the work divides evenly whatever the shape (no partly filled last round).
#include "flat_tile.h"

namespace sda {

namespace {

constexpr int F_UNIT = 128;                          // rows per work unit
constexpr int F_CO = 160;                            // output channels per workgroup
constexpr int F_NREP = 5;                            // 16-channel fragments per wave (80 channels)
constexpr int F_XROWS_MAX = 256 + 2 * PAD;           // 288 staged input rows (256-row tile, worst-case halo)
constexpr int F_XB = F_XROWS_MAX * ROW_B;            // 18 KB
constexpr int F_WB = F_CO * ROW_B;                   // 10 KB per tap
constexpr int F_WP = F_WB / 1024;                    // 10 pieces per tap
constexpr int F_MAIN = 2 * F_XB + 4 * F_WB;          // 77824 B
constexpr int F_EP_ROWS = 64;
constexpr int F_STRIDE = F_CO + 4;                   // floats; == 4 (mod 8): conflict-free accumulator writes
constexpr int F_EP_BYTES = F_EP_ROWS * F_STRIDE * 4; // 41984 B

template <int CH> struct FEpi {
  static constexpr int NCH = F_CO / CH;              // CH-channel chunks (16 bytes of E) per row
  static constexpr int RG = 256 / NCH;               // row groups
  static constexpr int ITERS = (F_EP_ROWS + RG - 1) / RG;
  static constexpr int RED_BYTES = RG * F_CO * 2 * 4;
};
template <int CH> struct FEpiGlu {                   // SDA_EPI_GLU: a row of the tile is 80 value + 80 gate channels
  static constexpr int NCH = F_CO / 2 / CH;          // CH-channel chunks of the 80 output channels
  static constexpr int RG = 256 / NCH;
  static constexpr int ITERS = (F_EP_ROWS + RG - 1) / RG;
};
constexpr int F_LDS = F_MAIN;
static_assert(F_EP_BYTES + FEpi<4>::RED_BYTES <= F_LDS && F_EP_BYTES + FEpi<8>::RED_BYTES <= F_LDS, "epilogue staging must fit");
static_assert(2 * F_LDS <= 160 * 1024, "two workgroups per CU");

// LDS-DMA schedule of a tile with MREP 16-row fragments per wave (R = 32 * MREP rows).  Everything a tap phase reads has
// landed — and is visible to every wave — ONE PHASE EARLY, so a phase's weight fragments and its first input fragment are
// read at the end of the phase before it, in front of the barrier, and the MFMAs start the moment the barrier opens.
// The stream of pieces issued during K-step s, in need order:
//   [x(s+1): XP] [tap 1 (s+1): 10] [tap 2 (s+1): 10] [tap 0 (s+2): 10]  = 4 * NP pieces, wave w takes j = w + 4 i (i < NP),
// in groups of (G0, G1, G2) during the three tap phases, between the MFMA rows.
//   * landed one phase early: a slab read in phase p must be issued by phase p - 3 (the counted wait at the top of phase
//     p - 1 leaves only the pieces of phase p - 2 in flight): x(s+1) and tap 0 (s+1) by phase (s, 0), tap 1 (s+1) by (s, 1),
//     tap 2 (s+1) by (s, 2);
//   * write-after-read: a tap slab goes into the ring slot of the slab four phases older, whose fragments were all read
//     (and waited for, lgkmcnt(0)) in front of THAT phase's barrier — so the slot is free from that barrier on: tap 1 (s+1)
//     from phase (s, 0), tap 2 (s+1) from (s, 1), tap 0 (s+2) from (s, 2); x(s+1) overwrites x(s-1), last read in phase
//     (s-1, 2): free from (s, 0).
template <int MREP> struct FSched {
  static constexpr int R = 32 * MREP;
  static constexpr int XP = (R + 2 * PAD) / 16;
  static constexpr int NP = (XP + 3 * F_WP) / 4;
  static constexpr int G0 = (XP + 3) / 4, G1 = (XP + 2 * F_WP) / 4 - G0, G2 = NP - G0 - G1;
  static_assert((XP + 3 * F_WP) % 4 == 0, "pieces must divide evenly over the four waves");
  static_assert(4 * G0 >= XP && 4 * G0 <= XP + F_WP, "group 0: all of x(s+1), nothing of tap 2");
  static_assert(4 * (G0 + G1) >= XP + F_WP && 4 * (G0 + G1) <= XP + 2 * F_WP, "group 1: the rest of tap 1, nothing of tap 0 (s+2)");
  static_assert(G0 <= MREP && G1 <= MREP && G2 <= MREP, "one piece per MFMA row at most");
  // pieces that may still be in flight at the top of a phase = the group issued during the phase before it
  static constexpr int WAIT0 = G2, WAIT1 = G0, WAIT2 = G1;
  // tile prologue: tap 0 (0) first (pieces j = w + 4 i < 10), then the stream of "K-step -1"; P(0, 0)'s operands (x(0),
  // tap 0 (0)) are behind everything but that stream's groups 1 and 2
  static constexpr int WAITP = G1 + G2;
};

// MFMA "weights" fragment of a 16-column identity block: B[col][k] = 1 where k == 16 * half + col, as mma16<E> reads
// it (lane (lr = col, lq) holds k = 8 lq + j in element j; fp32: k = 4 lq + j, one 16-wide block per slab, half = 0).
template <typename E> __device__ inline uint4 ident_frag(int half, int lr, int lq) {
  uint32_t w[4] = {0u, 0u, 0u, 0u};
  if constexpr (sizeof(E) == 4) {
    if (lq == (lr >> 2)) w[lr & 3] = 0x3F800000u;
  } else {
    const uint32_t one = std::is_same<E, half_t>::value ? 0x3C00u : 0x3F80u;
    if (lq == 2 * half + (lr >> 3)) w[(lr & 7) >> 1] = one << (16 * (lr & 1));
  }
  return make_uint4(w[0], w[1], w[2], w[3]);
}

struct FTileCtx {            // per-workgroup constants shared by all its tiles
  int tid, lane, wid, wave_m, wave_n, lr, lq, prow, pchunk, co0, nslab, Tp;
  long total_rows;
};

// One tile: R = 32 * MREP output rows starting at flat (= buffer) row f0; statistics go to row `stat_row` (and, for a
// 256-row tile, zeros to stat_row + 1: one row per 128-row unit).
// RESX: a residual is added (y = conv(x) + res; in the model and its backward pass res is always the conv's own input,
// L2-hot when the epilogue wants it).  16-bit types: each 64-row slice of it (20 KB) arrives by LDS-DMA in the 20 KB of LDS
// the epilogue leaves free, issued one barrier ahead of its use — no registers (the accumulators of the later slices are
// live and a spilled register's reload waits for every store in flight), 1 KB per request instead of 16 B per lane.
// fp32 (a slice would be 40 KB): per-thread 16-byte loads, one row ahead.
// STAMP (diagnostic build, never the product path): s_memtime stamps split every tap phase of the K loop into
// [barrier exit -> operands in registers] [MFMA + DMA issue] [vmcnt wait] [barrier]; the four cycle sums of wave 0 go to
// st[0..3] (+ the phase count in st[4]).  Shares are meaningful, the run time of this build is not.
template <typename E, bool BN, int MREP, bool RESX, int DIAG = 0, bool GLU = false>
__device__ __forceinline__ void flat_tile(const sda_conv_args& a, unsigned char* smem, const FTileCtx& c, const long f0,
                                          const int stat_row, unsigned long long* st = nullptr) {
  constexpr int SLAB = ROW_B / (int)sizeof(E);
  constexpr int PER16 = Elem<E>::PER16;
  constexpr int CH = Vec16<E>::N;
  using G = FEpi<CH>;
  using S = FSched<MREP>;
  constexpr bool STAMP = DIAG == 1;          // DIAG (diagnostic builds): 1 stamps, 2 no LDS-DMA inside the K loop, 3 no LDS-DMA and no fragment reads either
  constexpr int R = S::R;
  const int co0 = c.co0, dil = a.dil, Tp = c.Tp;
  const long lds_row0 = f0 - dil;                            // buffer row of LDS input row 0
  const E* __restrict__ xg = reinterpret_cast<const E*>(a.x);
  const E* __restrict__ wg = reinterpret_cast<const E*>(a.w);

  f32x4 acc[MREP][F_NREP];
#pragma unroll
  for (int m = 0; m < MREP; ++m)
#pragma unroll
    for (int n = 0; n < F_NREP; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

  {   // ==== K loop (a scope of its own: none of its per-thread values lives on into the epilogue, and vice versa — below)
  // (per-thread values re-derived per tile from a lane index hipcc cannot see through: they die with the K loop instead of
  // being carried — i.e. spilled — across the epilogue and the other tile size's code)
  int lane_k = c.lane;
  asm volatile("" : "+v"(lane_k));
  const int wid = c.wid, wave_m = c.wave_m, wave_n = c.wave_n, lr = lane_k & 15, lq = lane_k >> 4;
  const int k_prow = lane_k >> 2, k_pchunk = lane_k & 3;
  // ---- LDS-DMA pieces (FSched).  A piece is one wave-instruction: 64 lanes x 16 B land lane-linearly in LDS (16 rows x 64 B);
  // the bank swizzle goes on the SOURCE chunk (bit 2 of the row: the same for every piece, pieces start at multiples of 16
  // rows).  A piece's first row is wave-uniform, so its address is a scalar base (computed once per tile, with scalar
  // arithmetic) + ONE per-lane offset register that also carries the K-step's channel offset: no per-piece address
  // arithmetic at all inside the K loop.  x_pitch == w_pitch (supports()), so input and weight pieces share that register.
  // Pieces that would start outside the buffer are moved inside as a whole: they only feed rows that are never stored
  // (the first sample's leading padding / rows past the last sample).
  const uint32_t voff0 = (uint32_t)(((size_t)k_prow * a.x_pitch + (size_t)((k_pchunk ^ sw64(k_prow)) * PER16)) * sizeof(E));
  const uint32_t lds_base = __builtin_amdgcn_readfirstlane(lds_addr(smem));
  // stream piece i of this wave (j = wid + 4 i): kind 0 = x(s+1), 1..2 = tap 1..2 of K-step s+1, 3 = tap 0 of K-step s+2
  // (its source base already carries the extra K-step).  Source bases (scalar pairs, fixed for the tile):
  const char* pbase[S::NP];
  static_for<0, S::NP>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    const int j = wid + 4 * i;
    if (j < S::XP) {
      long srow = lds_row0 + j * 16;
      srow = srow < 0 ? 0 : (srow > a.x_rows_limit - 16 ? a.x_rows_limit - 16 : srow);
      pbase[i] = reinterpret_cast<const char*>(xg + (size_t)srow * a.x_pitch);
    } else {
      const int q = j - S::XP, kind = 1 + q / F_WP, qq = q - (kind - 1) * F_WP;
      const int tap = kind == 3 ? 0 : kind;
      pbase[i] = reinterpret_cast<const char*>(wg + ((size_t)tap * a.Cout_p + co0 + qq * 16) * a.w_pitch) + (kind == 3 ? ROW_B : 0);
    }
  });
  // LDS destinations: the buffers the stream of K-step s writes — x buffer (s+1) & 1, ring slots (3 (s+1) + tap) & 3 and
  // (3 (s+2)) & 3 — as four rotating scalars that already carry this wave's share (wid * 1024) of the piece index, so a
  // piece's destination is one of them plus a compile-time constant
  uint32_t xdst, wdst1, wdst2, wdst0;
  const uint32_t lds_w = lds_base + (uint32_t)wid * 1024u;
  auto set_dst = [&](int s) {
    xdst = lds_w + (uint32_t)((s + 1) & 1) * F_XB;
    wdst1 = lds_w + 2 * F_XB + (uint32_t)((3 * s + 4) & 3) * F_WB;
    wdst2 = lds_w + 2 * F_XB + (uint32_t)((3 * s + 5) & 3) * F_WB;
    wdst0 = lds_w + 2 * F_XB + (uint32_t)((3 * s + 6) & 3) * F_WB;
  };
  uint32_t kvoff = voff0;                                  // per-lane offset incl. the channel offset of K-step s+1
  // issue stream piece i.  (The last K-step but one has no tap 0 (s+2) to fetch; it issues those ten pieces all the same —
  // they read the 64 bytes behind each weight row's last slab, i.e. the head of the next row, always inside the tap-0 block
  // of the operand, into a ring slot nothing reads any more — so that every K-step but the last is the same code with the
  // same counts.)
  auto issue = [&](auto ic, auto padc) {
    constexpr int i = decltype(ic)::value;
    constexpr bool PD = decltype(padc)::value;
    constexpr int jlo = 4 * i, jhi = 4 * i + 3;
    constexpr int b1 = S::XP, b2 = S::XP + F_WP, b3 = S::XP + 2 * F_WP;      // kind boundaries in j
    // piece index inside its buffer minus wid, times 1024, per kind (x: j; tap pieces: j - boundary)
    constexpr int cx = 4 * i * 1024, c1 = (4 * i - b1) * 1024, c2 = (4 * i - b2) * 1024, c3 = (4 * i - b3) * 1024;
    const int j = wid + 4 * i;
    if constexpr (jlo >= b3) lds_dma16_lean<PD>(pbase[i], kvoff, wdst0 + (uint32_t)c3);
    else if constexpr (jhi >= b3) lds_dma16_lean<PD>(pbase[i], kvoff, j < b3 ? wdst2 + (uint32_t)c2 : wdst0 + (uint32_t)c3);   // tap 2 | tap 0 (s+2), by wave
    else if constexpr (jlo >= b2) lds_dma16_lean<PD>(pbase[i], kvoff, wdst2 + (uint32_t)c2);
    else if constexpr (jhi >= b2) lds_dma16_lean<PD>(pbase[i], kvoff, j < b2 ? wdst1 + (uint32_t)c1 : wdst2 + (uint32_t)c2);
    else if constexpr (jlo >= b1) lds_dma16_lean<PD>(pbase[i], kvoff, wdst1 + (uint32_t)c1);
    else if constexpr (jhi >= b1) lds_dma16_lean<PD>(pbase[i], kvoff, j < b1 ? xdst + (uint32_t)cx : wdst1 + (uint32_t)c1);
    else lds_dma16_lean<PD>(pbase[i], kvoff, xdst + (uint32_t)cx);
  };

  const int nslab = (a.flags & 512) ? 1 : c.nslab;          // flag 512 (diagnostic): one K-step only — the epilogue's time (results are garbage)
  // ---- tile prologue: tap 0 of K-step 0, then the stream of "K-step -1" (x(0), taps 1 and 2 of K-step 0, tap 0 of K-step 1)
  {
    const uint32_t w00 = lds_base + 2 * F_XB;              // slot (3 * 0 + 0) & 3
#pragma unroll
    for (int i = 0; i < (F_WP + 3) / 4; ++i) {
      const int qq = wid + 4 * i;
      if (qq < F_WP) lds_dma16_lean<true>(reinterpret_cast<const char*>(wg + ((size_t)co0 + qq * 16) * a.w_pitch), voff0, w00 + (uint32_t)qq * 1024u);
    }
  }
  set_dst(-1);
  static_for<0, S::NP>([&](auto ic) { issue(ic, TrueC{}); });
  wait_vmcnt_lit<S::WAITP>();
  __builtin_amdgcn_s_barrier();
  unsigned long long tA = 0, tB = 0, tC = 0, tD = 0, sAB = 0, sBC = 0, sCD = 0, sDA = 0, nph = 0;
  auto now = [&]() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
  };
  // operands of phase (0, 0): the wave's five weight fragments and its first input fragment
  const int wrow = wave_n * (F_CO / 2) + lr;
  const int xrow0 = wave_m * (R / 2) + lr;
  uint4 bf[F_NREP];
  uint4 af;
  auto load_b = [&](int slot) {
    const unsigned char* ws = smem + 2 * F_XB + slot * F_WB;
#pragma unroll
    for (int n = 0; n < F_NREP; ++n) bf[n] = *reinterpret_cast<const uint4*>(ws + lds_sw64(wrow + n * 16, lq));
  };
  load_b(0);
  af = *reinterpret_cast<const uint4*>(smem + lds_sw64(xrow0, lq));

  // One K-step = three tap phases.  MODE 0: every K-step but the last (issues the whole stream of this K-step), 2: the last
  // (issues nothing, waits for everything).
  auto kstep = [&](const int s, auto modec) {
    constexpr int MODE = decltype(modec)::value;
    const unsigned char* xs = smem + (s & 1) * F_XB;
    if constexpr (MODE < 2) { set_dst(s); kvoff += ROW_B; }
#pragma unroll
    for (int tap = 0; tap < 3; ++tap) {
      // this phase's operands are in registers (or on their way: lgkmcnt below); what the NEXT phase reads has landed once
      // all but this wave's youngest WAIT pieces are done ...
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // ... and our reads of the slots this phase's pieces overwrite are done
      if constexpr (MODE == 2) { if (tap < 2) wait_vmcnt_lit<0>(); }
      else if (tap == 0) wait_vmcnt_lit<S::WAIT0>();
      else if (tap == 1) wait_vmcnt_lit<S::WAIT1>();
      else wait_vmcnt_lit<S::WAIT2>();
      if constexpr (STAMP) { tD = now(); if (nph) sCD += tD - tC; }
      __builtin_amdgcn_s_barrier();                          // ... for every wave
      if constexpr (STAMP) { tA = now(); if (nph) sDA += tA - tD; tB = tA; }
      const int xrow = xrow0 + tap * dil;
      // next phase: (s, tap + 1) or (s + 1, 0)
      const bool last_phase = MODE == 2 && tap == 2;
      const unsigned char* xs_n = tap == 2 ? smem + ((s + 1) & 1) * F_XB : xs;
      const int xrow_n = xrow0 + (tap == 2 ? 0 : (tap + 1) * dil);
      static_for<0, MREP>([&](auto mc) {
        constexpr int m = decltype(mc)::value;
        uint4 af_next = af;
        if constexpr (DIAG == 3) { asm volatile("" : "+v"(af_next.x)); }
        else if constexpr (m + 1 < MREP) af_next = *reinterpret_cast<const uint4*>(xs + lds_sw64(xrow + (m + 1) * 16, lq));
        else { if (!last_phase) af_next = *reinterpret_cast<const uint4*>(xs_n + lds_sw64(xrow_n, lq)); }
        mma16_row<E, F_NREP>(af, bf, acc[m]);
        // this phase's DMA pieces go BETWEEN the MFMA rows (the wave's issue slot is free while the matrix pipe works
        // through the MFMAs queued before it)
        if constexpr (MODE < 2 && DIAG < 2) {
          if (tap == 0) { if constexpr (m < S::G0) issue(std::integral_constant<int, m>{}, FalseC{}); }
          else if (tap == 1) { if constexpr (m < S::G1) issue(std::integral_constant<int, S::G0 + m>{}, FalseC{}); }
          else { if constexpr (m < S::G2) issue(std::integral_constant<int, S::G0 + S::G1 + m>{}, FalseC{}); }
        }
        af = af_next;
      });
      // the next phase's weight fragments, in front of its barrier (their slab landed a phase ago)
      // (behind a scheduling fence: the fragments go into the registers the last MFMA row has just read — without it hipcc
      // overlaps the two sets, 20 registers the kernel does not have)
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (DIAG != 3) { if (!last_phase) load_b((3 * s + tap + 1) & 3); }
      if constexpr (STAMP) { tC = now(); sBC += tC - tB; ++nph; }
    }
  };
  {
    int s = 0;
    for (; s + 1 < nslab; ++s) kstep(s, std::integral_constant<int, 0>{});
    kstep(s, std::integral_constant<int, 2>{});
  }
  if constexpr (STAMP) {
    if (st && c.tid == 0) { st[0] += sAB; st[1] += sBC; st[2] += sCD; st[3] += sDA; st[4] += nph; }
  }

  }   // ==== end of the K loop's scope
  // ------------------------------------------------------------------ epilogue
  // Every per-thread value of the epilogue is derived HERE from a thread index hipcc cannot see through: otherwise it
  // computes them at kernel entry and keeps them in registers across the K loop, where 160 accumulators + 28 fragment
  // registers leave no room — the fragments were spilled INSIDE the loop (and a compiler-generated wait for a scratch
  // reload is vmcnt(0): it drains the LDS-DMA pipeline).
  int tid_e = c.tid;
  asm volatile("" : "+v"(tid_e));
  const int tid = tid_e, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6), wave_m = wid >> 1, wave_n = wid & 1;
  const int lr = lane & 15, lq = lane >> 4, e_prow = lane >> 2, e_pchunk = lane & 3;
  const uint32_t lds_base = __builtin_amdgcn_readfirstlane(lds_addr(smem));
  if (a.flags & 256) {        // diagnostic: skip the epilogue, keep the accumulators live
    float keep = 0.f;
#pragma unroll
    for (int m = 0; m < MREP; ++m)
#pragma unroll
      for (int n = 0; n < F_NREP; ++n) keep += acc[m][n][0] + acc[m][n][1] + acc[m][n][2] + acc[m][n][3];
    if (keep == 123.456f) reinterpret_cast<float*>(a.y)[0] = keep;
    __syncthreads();
    return;
  }
  if (a.bias) {
#pragma unroll
    for (int n = 0; n < F_NREP; ++n) {
      const float bv = a.bias[co0 + wave_n * (F_CO / 2) + n * 16 + lr];
#pragma unroll
      for (int m = 0; m < MREP; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[m][n][r] += bv;
    }
  }
  float* ep = reinterpret_cast<float*>(smem);
  float* red = reinterpret_cast<float*>(smem + F_EP_BYTES);
  const int chunk = tid % G::NCH, rg = tid / G::NCH;
  const bool active = rg < G::RG;
  E* __restrict__ yg = reinterpret_cast<E*>(a.y);
  const E* __restrict__ bnx = reinterpret_cast<const E*>(a.bn_x);
  float ssum[CH], ssq[CH];
#pragma unroll
  for (int j = 0; j < CH; ++j) { ssum[j] = 0.f; ssq[j] = 0.f; }
  // BatchNorm-backward mode: u = gamma * xhat + beta = ca * x + cb with ca = gamma * rstd, cb = beta - ca * mean (two
  // registers per channel instead of four); the second sum is accumulated as sum dg * x and turned into
  // sum dg * xhat = rstd * (sum dg * x - mean * sum dg) once per tile, below
  float ca[BN ? CH : 1], cb[BN ? CH : 1];
  if constexpr (BN) {
    if (active) {
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        const int cc = co0 + chunk * CH + j;
        const float ga = a.bn_coef[cc], be = a.bn_coef[a.Cout_p + cc], mu = a.bn_coef[2 * a.Cout_p + cc], rs = a.bn_coef[3 * a.Cout_p + cc];
        ca[j] = ga * rs;
        cb[j] = be - ga * rs * mu;
      }
    }
  }
  // position of this thread's first row inside its sample (rows p < PAD are the sample's zero padding: not stored)
  int pq = (int)(((unsigned)f0 + (unsigned)rg) % (unsigned)Tp);     // (flat rows fit 32 bits: supports() checks)
  // residual slices through LDS (16-bit types): the image is five 32-channel column blocks of 64 rows x 64 B, so a 1 KB
  // piece = 16 rows of one block: its first row and block are wave-uniform (scalar base + ONE per-lane offset register)
  constexpr bool RES_LDS = RESX && sizeof(E) == 2;
  constexpr int RES_OFF = F_EP_BYTES + FEpi<8>::RED_BYTES;
  static_assert(!RES_LDS || RES_OFF + F_EP_ROWS * F_CO * 2 <= F_LDS, "residual slice must fit beside the staging area");
  const E* __restrict__ resg = reinterpret_cast<const E*>(a.res);
  const uint32_t resvoff = (uint32_t)(((size_t)e_prow * a.Cout_p + (size_t)e_pchunk * 8) * 2);
  auto issue_res = [&](int q) {
    if constexpr (RES_LDS) {
#pragma unroll
      for (int i = 0; i < 5; ++i) {
        const int pc = wid + 4 * i, blk = pc >> 2, r0 = (pc & 3) * 16;
        long frow = f0 + q * F_EP_ROWS + r0;
        frow = frow > a.x_rows_limit - 16 ? a.x_rows_limit - 16 : frow;      // (rows past the last sample: never stored)
        lds_dma16_sv(resg + (size_t)frow * a.Cout_p + co0 + blk * 32, resvoff, lds_base + RES_OFF + pc * 1024);
      }
    }
  };

#pragma unroll
  for (int q = 0; q < R / F_EP_ROWS; ++q) {         // 64-row slices of the tile
    constexpr int WROWS = R / 2;                    // rows per wave
    const int owner = (q * F_EP_ROWS) / WROWS;      // the wave_m whose accumulators hold this slice
    const int m0 = (q * F_EP_ROWS - owner * WROWS) / 16;
    __syncthreads();                                // main-loop LDS reads (q == 0) / the previous slice's reads are done
    issue_res(q);                                   // lands while the accumulators are staged below
    if (wave_m == owner) {
#pragma unroll
      for (int mm = 0; mm < 4; ++mm)
#pragma unroll
        for (int n = 0; n < F_NREP; ++n)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            ep[(mm * 16 + lq * 4 + r) * F_STRIDE + wave_n * (F_CO / 2) + n * 16 + lr] = acc[m0 + mm][n][r];
    }
    if constexpr (RES_LDS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's residual pieces have landed
    __syncthreads();
    if constexpr (GLU) {
      // F.glu (models.py:164) on the staged slice: columns [0, 80) are values, [80, 160) their gates; both are rounded to the
      // storage type first, so the result is bit-equal to conv -> store -> glu_fwd_kernel; the gate is kept for backward
      using GG = FEpiGlu<CH>;
      const int gch = tid % GG::NCH, grg = tid / GG::NCH;
      const int Hp = a.Cout_p / 2, ho0 = co0 / 2;
      E* __restrict__ gateg = reinterpret_cast<E*>(a.y_pre);
      int p = (int)(((unsigned)(f0 + q * F_EP_ROWS) + (unsigned)grg) % (unsigned)Tp);
#pragma unroll
      for (int it = 0; it < GG::ITERS; ++it) {
        const int row = grg + it * GG::RG;
        if (grg < GG::RG && row < F_EP_ROWS && p >= PAD && f0 + q * F_EP_ROWS + row < c.total_rows) {
          float v[CH], g[CH];
#pragma unroll
          for (int q4 = 0; q4 < CH / 4; ++q4) {
            const float4 f = *reinterpret_cast<const float4*>(ep + row * F_STRIDE + gch * CH + q4 * 4);
            const float4 h = *reinterpret_cast<const float4*>(ep + row * F_STRIDE + F_CO / 2 + gch * CH + q4 * 4);
            v[q4 * 4 + 0] = f.x; v[q4 * 4 + 1] = f.y; v[q4 * 4 + 2] = f.z; v[q4 * 4 + 3] = f.w;
            g[q4 * 4 + 0] = h.x; g[q4 * 4 + 1] = h.y; g[q4 * 4 + 2] = h.z; g[q4 * 4 + 3] = h.w;
          }
          const size_t off = (size_t)(f0 + q * F_EP_ROWS + row) * Hp + ho0 + gch * CH;
          if (gateg) Vec16<E>::store(gateg + off, g);
#pragma unroll
          for (int j = 0; j < CH; ++j) v[j] = Vec16<E>::round(v[j]) * sigmoid_f(Vec16<E>::round(g[j]));
          Vec16<E>::store(yg + off, v);
        }
        p += GG::RG;
        while (p >= Tp) p -= Tp;
      }
    } else
    // this thread's rows of the slice (BatchNorm-input rows are fetched one row ahead, packed; the accumulators of the
    // later slices are still live here: no register may spill — a scratch reload waits for every store in flight)
    {
      auto row_off = [&](int it) { return (size_t)(f0 + q * F_EP_ROWS + rg + it * G::RG) * a.Cout_p + co0 + chunk * CH; };
      int p = pq;
      auto row_ok = [&](int it, int pp) {
        const int row = rg + it * G::RG;
        return active && row < F_EP_ROWS && pp >= PAD && f0 + q * F_EP_ROWS + row < c.total_rows;
      };
      bool ok_cur = row_ok(0, p);
      uint4 bx_cur = make_uint4(0u, 0u, 0u, 0u), rv_cur = make_uint4(0u, 0u, 0u, 0u);
      if constexpr (BN) { if (ok_cur) bx_cur = Vec16<E>::load_raw(bnx + row_off(0)); }
      if constexpr (RESX && !RES_LDS) { if (ok_cur) rv_cur = Vec16<E>::load_raw(resg + row_off(0)); }
#pragma unroll
      for (int it = 0; it < G::ITERS; ++it) {
        p += G::RG;
        if (p >= Tp) p -= Tp;
        bool ok_nxt = false;
        uint4 bx_nxt = make_uint4(0u, 0u, 0u, 0u), rv_nxt = make_uint4(0u, 0u, 0u, 0u);
        if (it + 1 < G::ITERS) {
          ok_nxt = row_ok(it + 1, p);
          if constexpr (BN) { if (ok_nxt) bx_nxt = Vec16<E>::load_raw(bnx + row_off(it + 1)); }
          if constexpr (RESX && !RES_LDS) { if (ok_nxt) rv_nxt = Vec16<E>::load_raw(resg + row_off(it + 1)); }
        }
        if (ok_cur) {
          const int row = rg + it * G::RG;
          float v[CH];
#pragma unroll
          for (int q4 = 0; q4 < CH / 4; ++q4) {
            const float4 f = *reinterpret_cast<const float4*>(ep + row * F_STRIDE + chunk * CH + q4 * 4);
            v[q4 * 4 + 0] = f.x; v[q4 * 4 + 1] = f.y; v[q4 * 4 + 2] = f.z; v[q4 * 4 + 3] = f.w;
          }
          if constexpr (RESX) {
            if constexpr (RES_LDS) rv_cur = *reinterpret_cast<const uint4*>(smem + RES_OFF + (chunk >> 2) * 4096 + row * 64 + (chunk & 3) * 16);
            float r8[CH];
            Vec16<E>::unpack(rv_cur, r8);
#pragma unroll
            for (int j = 0; j < CH; ++j) v[j] += r8[j];
          }
          if constexpr (BN) {
            // BatchNorm+GELU backward sums of the layer this gradient enters: dg = dy * GELU'(gamma * xhat + beta), dy as stored
            // (SDA_EPI_BN_STORE_DG: dg is what gets stored, and summed as stored)
            const bool store_dg = a.flags & SDA_EPI_BN_STORE_DG;
            if (!store_dg) Vec16<E>::store(yg + row_off(it), v);
            float x8[CH];
            Vec16<E>::unpack(bx_cur, x8);
#pragma unroll
            for (int j = 0; j < CH; j += 2) {           // on pairs: the 16-bit form of GELU' is packed arithmetic
              const f32x2 gp = gelu_grad_pair<E>(f32x2{fmaf(ca[j], x8[j], cb[j]), fmaf(ca[j + 1], x8[j + 1], cb[j + 1])});
              float dg0 = Vec16<E>::round(v[j]) * gp.x, dg1 = Vec16<E>::round(v[j + 1]) * gp.y;
              if (store_dg) { dg0 = Vec16<E>::round(dg0); dg1 = Vec16<E>::round(dg1); v[j] = dg0; v[j + 1] = dg1; }
              ssum[j] += dg0; ssum[j + 1] += dg1;
              ssq[j] = fmaf(dg0, x8[j], ssq[j]); ssq[j + 1] = fmaf(dg1, x8[j + 1], ssq[j + 1]);
            }
            if (store_dg) Vec16<E>::store(yg + row_off(it), v);
          } else if (a.stats) {
            Vec16<E>::store(yg + row_off(it), v);
            // statistics of the values as stored (rounded to E), so BatchNorm normalises what it will read
#pragma unroll
            for (int j = 0; j < CH; ++j) { const float qv = Vec16<E>::round(v[j]); ssum[j] += qv; ssq[j] += qv * qv; }
          } else {
            Vec16<E>::store(yg + row_off(it), v);
          }
        }
        ok_cur = ok_nxt; bx_cur = bx_nxt; rv_cur = rv_nxt;
      }
    }
    pq += F_EP_ROWS;
    while (pq >= Tp) pq -= Tp;
  }
  if (a.stats) {
    __syncthreads();
    if (active) {
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        red[(rg * 2 + 0) * F_CO + chunk * CH + j] = ssum[j];
        red[(rg * 2 + 1) * F_CO + chunk * CH + j] = ssq[j];
      }
    }
    __syncthreads();
    if (tid < F_CO) {
      const int cc = tid;
      float s0 = 0.f, s1 = 0.f;
      for (int g = 0; g < G::RG; ++g) { s0 += red[(g * 2 + 0) * F_CO + cc]; s1 += red[(g * 2 + 1) * F_CO + cc]; }
      if constexpr (BN) s1 = a.bn_coef[3 * a.Cout_p + co0 + cc] * (s1 - a.bn_coef[2 * a.Cout_p + co0 + cc] * s0);   // sum dg * xhat
      a.stats[((size_t)stat_row * 2 + 0) * a.Cout_p + co0 + cc] = s0;
      a.stats[((size_t)stat_row * 2 + 1) * a.Cout_p + co0 + cc] = s1;
      if (R > F_UNIT) {
        a.stats[((size_t)(stat_row + 1) * 2 + 0) * a.Cout_p + co0 + cc] = 0.f;
        a.stats[((size_t)(stat_row + 1) * 2 + 1) * a.Cout_p + co0 + cc] = 0.f;
      }
    }
  }
  __syncthreads();            // the next tile's LDS-DMA overwrites the staging / reduction area
}

template <typename E, bool BN, bool RESX, int DIAG = 0, bool GLU = false>
__global__ __launch_bounds__(256, 2) void conv3_flat_kernel(const sda_conv_args a, const int n_units, const int units_per_wg,
                                                            const int runs_per_co, const long total_rows, const int Tp) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  FTileCtx c;
  c.tid = threadIdx.x;
  c.lane = c.tid & 63;
  c.wid = __builtin_amdgcn_readfirstlane(c.tid >> 6);
  c.wave_m = c.wid >> 1;
  c.wave_n = c.wid & 1;
  c.lr = c.lane & 15;
  c.lq = c.lane >> 4;
  c.prow = c.lane >> 2;
  c.pchunk = c.lane & 3;
  c.nslab = a.Cin_p / (ROW_B / (int)sizeof(E));
  c.Tp = Tp;
  c.total_rows = total_rows;

  // XCD-aware order (blocks i and i+8 share an XCD/L2): the n_co workgroups that read the same input rows are dealt
  // to the same XCD.  Pure speed: any placement is correct.
  int bid = blockIdx.x, co_tile, run;
  const int n_co = a.Cout_p / F_CO;
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
  c.co0 = co_tile * F_CO;
  int u = run * units_per_wg;                               // this workgroup's 128-row units: [u, u_end)
  const int u_end = min(n_units, u + units_per_wg);
  // The dispatcher fills an XCD's 32 CUs once before it gives any of them a second workgroup, so workgroups
  // (blockIdx / 8) 0..31 and 32..63 of an XCD are the co-resident pairs (verified with tools/flat_timeline.py; not
  // guaranteed, it only matters for speed).  With flag 1024 the second of a pair takes its 128-row tile(s) FIRST, the
  // first one LAST, so that their epilogues fall at different times; measured: no gain, default off.
  // flags 1024 / 2048 (diagnostic): a different tile order (128-row tile first) for the second workgroup of every CU / for
  // every other CU — spreads the epilogues (HBM bursts) over more distinct moments
  const bool small_first = ((((blockIdx.x >> 3) >> 5) & 1) && (a.flags & 1024)) || (((blockIdx.x >> 3) & 1) && (a.flags & 2048));
  // The two workgroups of a CU share each SIMD's matrix pipe and issue slots, arbitrated by priority, then AGE: at equal
  // priority the first-dispatched one wins every time — it runs its three units in 42 us and the second one in 54 us, the
  // last 11 us alone on the CU at a single workgroup's (poor) rate (tools/flat_timeline.py).  So the second
  // workgroup of each pair holds priority 1 for its FIRST tile and drops it afterwards: each is the winner for about half
  // of its work and both finish together.  Which workgroups share a CU is the dispatcher's business (observed: blocks i and
  // i + 8 * 32 of an XCD) — a wrong guess costs speed only.
  const bool younger = (((blockIdx.x >> 3) >> 5) & 1) && !(a.flags & 64);              // flag 64 (diagnostic): no priority hand-over
  if (younger) __builtin_amdgcn_s_setprio(1);
  int n = u_end - u;
  if (n <= 0) return;
  // tile plan: `pairs` 256-row tiles and `lead` + `tail` 128-row tiles around them
  int lead, tail, pairs;
  if (n & 1) { pairs = n >> 1; lead = small_first ? 1 : 0; tail = 1 - lead; }
  else if (small_first && n >= 2) { pairs = (n >> 1) - 1; lead = 1; tail = 1; }
  else { pairs = n >> 1; lead = 0; tail = 0; }
  // diagnostic (flag 32, `partial` = a buffer of 16 x 8 bytes per workgroup that nothing else reads): wall-clock stamps
  // (100 MHz) at the start and after every tile, plus where the workgroup ran — the timeline behind DESIGN.md's numbers
  unsigned long long* dbg = (a.flags & 32) && a.partial ? reinterpret_cast<unsigned long long*>(a.partial) + (size_t)blockIdx.x * 32 : nullptr;
  int nstamp = 0;
  auto stamp = [&]() {
    if (dbg && c.tid == 0 && nstamp < 5) {
      dbg[4 + nstamp] = __builtin_amdgcn_s_memrealtime();
      dbg[10 + nstamp] = __builtin_amdgcn_s_memtime();       // shader clock: cycles / 10 ns = the clock the chip holds
    }
    ++nstamp;
  };
  if (dbg && c.tid == 0) {
    dbg[0] = __builtin_amdgcn_s_getreg((31 << 11) | 4);      // HW_REG_HW_ID: wave / SIMD / CU / SH / SE
    dbg[1] = __builtin_amdgcn_s_getreg((31 << 11) | 20);     // HW_REG_XCC_ID
    dbg[2] = (unsigned long long)lead | ((unsigned long long)pairs << 8) | ((unsigned long long)tail << 16);
    dbg[3] = (unsigned long long)u;
  }
  stamp();
  if constexpr (DIAG == 1) {       // only 256-row tiles are stamped; sums land behind the wall-clock stamps of this workgroup
    if (dbg && c.tid == 0) { for (int i = 0; i < 5; ++i) dbg[16 + i] = 0; }
    for (int p = 0; p < pairs; ++p, u += 2) { flat_tile<E, BN, 8, RESX, 1>(a, smem, c, (long)u * F_UNIT, u, dbg ? dbg + 16 : nullptr); stamp(); }
    return;
  }
  if constexpr (sizeof(E) == 4) {
    // fp32 storage (the exact path): 128-row tiles only — 64 x 80 per wave, 80 accumulator registers.  The 256-row tile's 160
    // accumulators leave the fp32 instantiation (16-deep K-steps: twice the fragment traffic per MFMA row, 40-KB epilogue
    // slices read per thread) some 300 registers short, spilled inside the K loop.
    for (; u < u_end; ++u) { flat_tile<E, BN, 4, RESX, 0, GLU>(a, smem, c, (long)u * F_UNIT, u); __builtin_amdgcn_s_setprio(0); }
    return;
  }
  if (lead) { flat_tile<E, BN, 4, RESX, (DIAG > 1 ? DIAG : 0), GLU>(a, smem, c, (long)u * F_UNIT, u); ++u; stamp(); __builtin_amdgcn_s_setprio(0); }
  for (int p = 0; p < pairs; ++p, u += 2) { flat_tile<E, BN, 8, RESX, (DIAG > 1 ? DIAG : 0), GLU>(a, smem, c, (long)u * F_UNIT, u); stamp(); __builtin_amdgcn_s_setprio(0); }
  if (tail) { flat_tile<E, BN, 4, RESX, (DIAG > 1 ? DIAG : 0), GLU>(a, smem, c, (long)u * F_UNIT, u); stamp(); }
}

struct FlatPlan { int n_units, units_per_wg, runs_per_co, grid; };

FlatPlan flat_plan(const sda_conv_args& a) {
  FlatPlan p;
  const long total_rows = (long)a.B * rows_tp(a.T);
  p.n_units = (int)((total_rows + F_UNIT - 1) / F_UNIT);
  const int n_co = a.Cout_p / F_CO;
  const int cus = launch_cus();
  // two workgroups per CU; one with SDA_CONV_ONE_PER_CU (the caller wants the other half of every CU's LDS for a kernel on
  // another stream: in backward the weight-gradient GEMMs run beside the data-gradient convs)
  const long slots = ((a.flags & SDA_CONV_ONE_PER_CU) ? 1L : 2L) * cus;
  p.units_per_wg = (int)(((long)p.n_units * n_co + slots - 1) / slots);
  if (p.units_per_wg < 1) p.units_per_wg = 1;
  p.runs_per_co = (p.n_units + p.units_per_wg - 1) / p.units_per_wg;
  p.grid = p.runs_per_co * n_co;
  return p;
}

template <typename E, bool BN, bool RESX, int DIAG = 0, bool GLU = false>
int launch_flat(const sda_conv_args& a, hipStream_t st) {
  static unsigned long long attr_done = 0;        // per device
  auto kern = conv3_flat_kernel<E, BN, RESX, DIAG, GLU>;
  if (first_use_on_device(attr_done)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, F_LDS) != hipSuccess) {
      set_error("conv3_flat: cannot reserve %d bytes of LDS", F_LDS);
      return -3;
    }
  }
  const FlatPlan p = flat_plan(a);
  hipLaunchKernelGGL(kern, dim3((unsigned)p.grid), dim3(256), F_LDS, st, a, p.n_units, p.units_per_wg, p.runs_per_co,
                     (long)a.B * rows_tp(a.T), rows_tp(a.T));
  return check_launch("conv3_flat");
}

}  // namespace

int conv3_flat_stat_rows(int B, int T) { return (int)(((long)B * rows_tp(T) + F_UNIT - 1) / F_UNIT); }

bool conv3_flat_supports(const sda_conv_args& a) {
  const bool glu = a.flags & SDA_EPI_GLU;
  if (glu ? (a.res || a.stats || a.bn_x) : a.y_pre != nullptr) return false;
  return a.KS == 3 && a.Cout_p % F_CO == 0 && !a.widx && a.ksplit == 1 && (!a.partial || (a.flags & 32)) && !(a.flags & SDA_EPI_GELU) &&
         a.y && a.x_row0 == PAD && a.x_pitch == a.w_pitch && a.x_sample_rows == rows_tp(a.T) && (!a.bn_x || (a.bn_coef && a.stats)) &&

         a.x_rows_limit >= (long)a.B * rows_tp(a.T) + 3 * PAD && a.x_rows_limit < (1L << 31) && a.w_rows_limit >= a.Cout_p &&
         a.Cin_p / (ROW_B / (a.dtype == SDA_F32 ? 4 : 2)) >= 1;
}

template <typename E> static int launch_flat_e(const sda_conv_args& a, hipStream_t st) {
  if (a.flags & SDA_EPI_GLU) return launch_flat<E, false, false, 0, true>(a, st);
  if (a.bn_x) return a.res ? launch_flat<E, true, true>(a, st) : launch_flat<E, true, false>(a, st);
  return a.res ? launch_flat<E, false, true>(a, st) : launch_flat<E, false, false>(a, st);
}

int launch_conv3_flat(const sda_conv_args& a, hipStream_t st) {
  if ((a.flags & 128) && (a.flags & 32) && a.dtype == SDA_BF16 && !a.bn_x && !a.res) return launch_flat<uint16_t, false, false, 1>(a, st);   // diagnostic
  if ((a.flags & 24) && a.dtype == SDA_BF16 && !a.bn_x && !a.res && !(a.flags & SDA_EPI_GLU))                                        // diagnostic (garbage results)
    return (a.flags & 8) ? launch_flat<uint16_t, false, false, 3>(a, st) : launch_flat<uint16_t, false, false, 2>(a, st);
  if (a.dtype == SDA_F32) return launch_flat_e<float>(a, st);
  if (a.dtype == SDA_F16) return launch_flat_e<half_t>(a, st);
  return launch_flat_e<uint16_t>(a, st);
}

}  // namespace sda

// rows of sda_conv_args.stats one launch writes: the tile-per-workgroup kernels write B * sda_conv_n_t_tiles(T) rows
// (one per 128-row tile of a sample), the flat-tile kernel one per 128-row unit of the flat row space
extern "C" int sda_conv_stats_rows(int B, int T, int KS, int Cout_p, int flags) {
  if ((flags & SDA_CONV_FLAT_TILES) && KS == 3 && Cout_p % sda::F_CO == 0) return sda::conv3_flat_stat_rows(B, T);
  if ((flags & SDA_CONV_WIDE_TILES) && KS == 1 && (flags & SDA_EPI_GELU_BWD)) return sda::conv1_wide_stat_rows(B, T);   // conv1_wide: one row per 256-row tile
  if ((flags & SDA_CONV_FLAT_TILES) && KS == 1 && (flags & SDA_EPI_GELU_BWD)) return sda::conv3_flat_stat_rows(B, T);   // conv1_flat: same units
  return B * ((T + sda::TILE_T - 1) / sda::TILE_T);
}

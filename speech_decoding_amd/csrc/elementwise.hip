// HBM-bound stages of the path: layout packing, BatchNorm statistics/apply (+GELU), GLU, GELU
// backward, column sums, slab reduction.  All kernels touch VALID rows only (pad rows stay zero) and
// move 16 bytes (fp32) / 8-16 bytes (bf16) per lane.  Reductions are two-stage and ordered, so results
// are bitwise reproducible run to run.
#include <math.h>

#include "sd_common.h"

namespace sda {

// ------------------------------------------------------------------------------------------------
// Row streaming skeleton of the elementwise passes.  A thread OWNS one 16-byte channel chunk (its per-channel
// coefficients stay in registers) and walks the valid rows of its workgroup's contiguous row range with stride RG =
// 256 / chunks-per-row; (sample, time) advance incrementally, so the loop has no integer division and no
// coefficient loads — with both, the BatchNorm/GELU passes were VALU-bound (≈ 4.3 TB/s), not HBM-bound.
// Rows are taken U at a time: all loads of a batch are issued before the first use.
// ------------------------------------------------------------------------------------------------
// Wave priority of the backward's HBM-bound passes on the step's main chain (BatchNorm backward, GLU / GELU backward with their
// column sums and the small reductions behind them): they run beside the weight-gradient GEMMs of the other stream, and at
// s_setprio 3 their loads and stores win the SIMD's issue arbitration against those waves.  Round 5, six alternations on
// one box (ms per step): off 6.771, on 6.777; with the data-gradient convs' waves raised as well (SDA_CONV_WAVE_PRIO) 6.725
// against 6.754 for those alone — the pair is worth 0.05 ms, either one alone nothing.  -DSDA_EW_BWD_PRIO=0 turns it off.
#ifndef SDA_EW_BWD_PRIO
#define SDA_EW_BWD_PRIO 3
#endif
__device__ __forceinline__ void ew_bwd_prio() {
  if constexpr (SDA_EW_BWD_PRIO > 0) __builtin_amdgcn_s_setprio(SDA_EW_BWD_PRIO);
}

struct RowWalk {
  int r, r1, step, T, b, t;
  __device__ RowWalk(int r0, int r1_, int step_, int T_) : r(r0), r1(r1_), step(step_), T(T_) {
    b = r0 / T_;
    t = r0 - b * T_;
  }
  __device__ bool valid() const { return r < r1; }
  __device__ size_t mem_row() const { return (size_t)b * rows_tp(T) + PAD + t; }
  __device__ void next() {
    r += step; t += step;
    while (t >= T) { t -= T; ++b; }
  }
};

// The same walk reduced to what the inner loop needs: a 32-bit ELEMENT offset that advances by a constant and takes the
// pad rows in one extra add when the time index wraps (row-layout buffers stay below 2^32 elements: the launchers check),
// and the number of rows this thread owns — so the loop body has no validity predicates and no 64-bit multiplies.
struct RowCursor {
  uint32_t off, step, wrap;
  int t, T, RG, n;
  __device__ RowCursor(int r0, int r1, int rg, int RG_, int T_, int Cp, int col) : T(T_), RG(RG_) {
    const int first = r0 + rg;
    n = first < r1 ? (r1 - first + RG_ - 1) / RG_ : 0;
    const int b = first / T_;
    t = first - b * T_;
    off = (uint32_t)(b * rows_tp(T_) + PAD + t) * (uint32_t)Cp + (uint32_t)col;
    step = (uint32_t)RG_ * (uint32_t)Cp;
    wrap = (uint32_t)PAD * (uint32_t)Cp;
  }
  __device__ uint32_t take() {              // offset of the current row; moves to the next one
    const uint32_t o = off;
    off += step; t += RG;
    while (t >= T) { t -= T; off += wrap; }
    return o;
  }
};

// valid-row range of this workgroup: rows split evenly over the grid
__device__ inline void block_rows(int B, int T, int& r0, int& r1) {
  const int rows = B * T;
  const int per = (rows + (int)gridDim.x - 1) / (int)gridDim.x;
  r0 = min(rows, (int)blockIdx.x * per);
  r1 = min(rows, r0 + per);
}

// grid of a streaming pass: about 32 rows per thread-row group, at most 8 workgroups per CU
// RowCursor's offsets are 32-bit element indices
static inline bool fits_u32(int B, int T, long width) { return (unsigned long long)rows_alloc(B, T) * (unsigned long long)width < (1ull << 32); }

static inline int stream_blocks(int B, int T, int nch) {
  const long rows = (long)B * T;
  const int RG = nch >= 256 ? 1 : 256 / nch;
  long nb = (rows + (long)RG * 8 - 1) / ((long)RG * 8);
  return (int)(nb < 1 ? 1 : (nb > 2048 ? 2048 : nb));
}

// ------------------------------------------------------------------------------------------------
// (B, C, T) fp32  <->  RL rows
// ------------------------------------------------------------------------------------------------
template <typename E>
__global__ __launch_bounds__(256) void pack_rows_kernel(const float* __restrict__ src, E* __restrict__ dst,
                                                        int C, int T, int Cp, int one_ch) {
  __shared__ float tile[64][65];
  const int b = blockIdx.z, c0 = blockIdx.y * 64, t0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int cc = ty; cc < 64; cc += 4) {
    const int c = c0 + cc, t = t0 + tx;
    tile[cc][tx] = (c < C && t < T) ? src[((size_t)b * C + c) * T + t] : ((c == one_ch && t < T) ? 1.f : 0.f);
  }
  __syncthreads();
  for (int rr = ty; rr < 64; rr += 4) {
    const int t = t0 + rr;
    if (t < T) Elem<E>::st(dst + ((size_t)b * rows_tp(T) + PAD + t) * Cp + c0 + tx, tile[tx][rr]);
  }
}

// The same for T % 4 == 0 and a 16-byte aligned source: 16-byte loads along t, 16-byte stores along c (8 channels of a 16-bit
// type per lane instead of one: a wave's store covers 8 rows x 128 B instead of 128 B).  Same values, same rounding.
template <typename E>
__global__ __launch_bounds__(256) void pack_rows_vec_kernel(const float* __restrict__ src, E* __restrict__ dst,
                                                            int C, int T, int Cp, int one_ch) {
  __shared__ float tile[64][65];
  const int b = blockIdx.z, c0 = blockIdx.y * 64, t0 = blockIdx.x * 64;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int idx = threadIdx.x + 256 * k, cc = idx >> 4, t4 = (idx & 15) * 4;
    const int c = c0 + cc, t = t0 + t4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (t < T) {                                       // (T % 4 == 0: the four are inside or outside together)
      if (c < C) v = *reinterpret_cast<const float4*>(src + ((size_t)b * C + c) * T + t);
      else if (c == one_ch) v = make_float4(1.f, 1.f, 1.f, 1.f);
    }
    tile[cc][t4] = v.x; tile[cc][t4 + 1] = v.y; tile[cc][t4 + 2] = v.z; tile[cc][t4 + 3] = v.w;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int item = threadIdx.x + 256 * k, chunk = item & 7, rr = item >> 3;
    const int t = t0 + rr;
    if (t < T) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = tile[chunk * 8 + j][rr];
      E* out = dst + ((size_t)b * rows_tp(T) + PAD + t) * Cp + c0 + chunk * 8;
      if constexpr (sizeof(E) == 4) { Vec16<E>::store(out, v); Vec16<E>::store(out + 4, v + 4); }
      else Vec16<E>::store(out, v);
    }
  }
}

template <typename E>
__global__ __launch_bounds__(256) void unpack_rows_kernel(const E* __restrict__ src, float* __restrict__ dst,
                                                          int C, int T, int Cp) {
  __shared__ float tile[64][65];
  const int b = blockIdx.z, c0 = blockIdx.y * 64, t0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int rr = ty; rr < 64; rr += 4) {
    const int t = t0 + rr;
    tile[rr][tx] = (t < T) ? Elem<E>::ld(src + ((size_t)b * rows_tp(T) + PAD + t) * Cp + c0 + tx) : 0.f;
  }
  __syncthreads();
  for (int cc = ty; cc < 64; cc += 4) {
    const int c = c0 + cc, t = t0 + tx;
    if (c < C && t < T) dst[((size_t)b * C + c) * T + t] = tile[tx][cc];
  }
}

// ------------------------------------------------------------------------------------------------
// per-sample sum of squares (loss.py:64-65 norms), two-stage
// ------------------------------------------------------------------------------------------------
constexpr int SUMSQ_CHUNKS = 64;

template <typename E>
__global__ __launch_bounds__(256) void rows_sumsq_kernel(const E* __restrict__ x, float* __restrict__ scratch,
                                                         long row_elems, long pitch) {
  __shared__ float red[4];
  const int b = blockIdx.y, ch = blockIdx.x;
  const long n4 = row_elems / 4;
  const long per = (n4 + SUMSQ_CHUNKS - 1) / SUMSQ_CHUNKS;
  const long beg = ch * per, end = min(n4, beg + per);
  const E* p = x + (size_t)b * pitch;
  float s = 0.f;
  for (long i = beg + threadIdx.x; i < end; i += 256) {
    const float4 v = load4(p + i * 4);
    s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) scratch[b * SUMSQ_CHUNKS + ch] = red[0] + red[1] + red[2] + red[3];
}

__global__ void rows_sumsq_final_kernel(const float* __restrict__ scratch, float* __restrict__ out, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  double s = 0.0;
  for (int i = 0; i < SUMSQ_CHUNKS; ++i) s += (double)scratch[b * SUMSQ_CHUNKS + i];
  out[b] = (float)s;
}

// per-sample sum of squares from the per-tile channel statistics a conv epilogue already wrote
// (stats[tile][1][c] = sum over the tile's valid rows of y^2): out[b] = sum over the sample's tiles and all channels
__global__ __launch_bounds__(256) void rows_sumsq_from_stats_kernel(const float* __restrict__ stats, int tiles_per_sample,
                                                                    int Cp, float* __restrict__ out) {
  __shared__ double sh[256];
  const int b = blockIdx.x;
  double s = 0.0;
  for (int t = 0; t < tiles_per_sample; ++t) {
    const float* p = stats + (((size_t)b * tiles_per_sample + t) * 2 + 1) * Cp;
    for (int c = threadIdx.x; c < Cp; c += 256) s += (double)p[c];
  }
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {                      // fixed tree: deterministic
    if (threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[b] = (float)sh[0];
}

// per-sample norms from conv1_flat's SDA_EPI_ROW_SUMSQ partials ([buffer row][n_parts]): sample b = rows b * (T + PAD) + PAD + t
__global__ __launch_bounds__(256) void rows_sumsq_from_row_parts_kernel(const float* __restrict__ parts, int n_parts, int T,
                                                                        float* __restrict__ out) {
  __shared__ double sh[256];
  const int b = blockIdx.x;
  const float* p = parts + ((size_t)b * rows_tp(T) + PAD) * n_parts;
  double s = 0.0;
  for (int i = threadIdx.x; i < T * n_parts; i += 256) s += (double)p[i];
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {                      // fixed tree: deterministic
    if (threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[b] = (float)sh[0];
}

// ------------------------------------------------------------------------------------------------
// weight / vector packing
// ------------------------------------------------------------------------------------------------
__device__ inline int glu_unmap(int cop, int Cout, int half, int half_p, int tile = 0) {   // packed row -> source row or -1
  if (half == 0) return cop < Cout ? cop : -1;
  if (tile > 0) {                        // SDA_EPI_GLU layout: per 2*tile packed channels, `tile` values then `tile` gates
    const int j = cop / (2 * tile), w = cop - j * 2 * tile;
    const int r = j * tile + (w < tile ? w : w - tile);          // channel inside its half
    if (w < tile) return r < half ? r : -1;
    return r < Cout - half ? half + r : -1;
  }
  if (cop < half_p) return cop < half ? cop : -1;
  const int r = cop - half_p;
  return r < Cout - half ? half + r : -1;
}
__device__ inline int glu_map(int co, int half, int half_p) { return (half == 0 || co < half) ? co : half_p + co - half; }

template <typename E>
__global__ void pack_conv_weight_kernel(const float* __restrict__ w, E* __restrict__ dst, int nW, int Cout, int Cin,
                                        int KS, int Cout_p, int Cin_p, int mode, int half, int half_p) {
  const size_t total = (size_t)nW * KS * Cout_p * Cin_p;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    int co, ci, tap;
    size_t q = i;
    if (mode == 0) {
      ci = q % Cin_p; q /= Cin_p;
      co = glu_unmap((int)(q % Cout_p), Cout, half, half_p); q /= Cout_p;
      tap = q % KS; q /= KS;
    } else {
      co = glu_unmap((int)(q % Cout_p), Cout, half, half_p); q /= Cout_p;
      ci = q % Cin_p; q /= Cin_p;
      tap = KS - 1 - (int)(q % KS); q /= KS;
    }
    const int n = (int)q;
    float v = 0.f;
    if (co >= 0 && ci < Cin) v = w[(((size_t)n * Cout + co) * Cin + ci) * KS + tap];
    Elem<E>::st(dst + i, v);
  }
}

__global__ void unpack_conv_wgrad_kernel(const float* __restrict__ g, float* __restrict__ dst, int nW, int Cout,
                                         int Cin, int KS, int Cout_p, int Cin_p, int half, int half_p) {
  const size_t total = (size_t)nW * Cout * Cin * KS;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    size_t q = i;
    const int tap = q % KS; q /= KS;
    const int ci = q % Cin; q /= Cin;
    const int co = q % Cout; q /= Cout;
    const int n = (int)q;
    dst[i] = g[(((size_t)n * KS + tap) * Cout_p + glu_map(co, half, half_p)) * Cin_p + ci];
  }
}

// All operand packs of one step in ONE launch: blockIdx.y selects the descriptor (device array).
template <typename E>
__global__ void pack_multi_kernel(const sda_pack_desc* __restrict__ descs) {
  const sda_pack_desc d = descs[blockIdx.y];
  const size_t total = (size_t)d.total;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    if (d.is_vector) {
      const int c = glu_unmap((int)i, d.Cout, d.glu_half, d.glu_half_p, d.glu_tile);
      reinterpret_cast<float*>(d.dst)[i] = c >= 0 ? d.src[c] : 0.f;
      continue;
    }
    int co, ci, tap;
    size_t q = i;
    if (d.mode == 0) {
      ci = q % d.Cin_p; q /= d.Cin_p;
      co = glu_unmap((int)(q % d.Cout_p), d.Cout, d.glu_half, d.glu_half_p, d.glu_tile); q /= d.Cout_p;
      tap = q % d.KS; q /= d.KS;
    } else {
      co = glu_unmap((int)(q % d.Cout_p), d.Cout, d.glu_half, d.glu_half_p, d.glu_tile); q /= d.Cout_p;
      ci = q % d.Cin_p; q /= d.Cin_p;
      tap = d.KS - 1 - (int)(q % d.KS); q /= d.KS;
    }
    const int n = (int)q;
    float v = 0.f;
    if (co >= 0 && ci < d.Cin) v = d.src[(((size_t)n * d.Cout + co) * d.Cin + ci) * d.KS + tap];
    Elem<E>::st(reinterpret_cast<E*>(d.dst) + i, v);
  }
}

// dst[co][ci][tap] = sum_s slabs[s][tap][co'][ci]   (ordered sum over the K-split slabs + un-packing in one pass)
__global__ void reduce_unpack_wgrad_kernel(const float* __restrict__ slabs, int nslabs, float* __restrict__ dst, int Cout,
                                           int Cin, int KS, int Cout_p, int Cin_p, int half, int half_p) {
  const size_t total = (size_t)Cout * Cin * KS;
  const size_t slab = (size_t)KS * Cout_p * Cin_p;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    // thread order follows the SOURCE layout ([tap][co][ci], ci fastest) so the slab reads coalesce
    size_t q = i;
    const int ci = q % Cin; q /= Cin;
    const int co = q % Cout; q /= Cout;
    const int tap = (int)q;
    const size_t src = ((size_t)tap * Cout_p + glu_map(co, half, half_p)) * Cin_p + ci;
    // 8 independent loads in flight, then added in slab order (the sum order does not depend on the batching)
    float sum = 0.f;
    int k = 0;
    for (; k + 8 <= nslabs; k += 8) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = slabs[(size_t)(k + j) * slab + src];
#pragma unroll
      for (int j = 0; j < 8; ++j) sum += v[j];
    }
    for (; k < nslabs; ++k) sum += slabs[(size_t)k * slab + src];
    dst[((size_t)co * Cin + ci) * KS + tap] = sum;
  }
}

// The same on 16-byte loads (Cin % 4 == 0, slabs 16-byte aligned): a thread sums FOUR consecutive input channels of one
// (tap, co) — a quarter of the threads, each with 8 x 16 B in flight; identical sums (each output adds its slabs in slab order)
__global__ __launch_bounds__(256) void reduce_unpack_wgrad_vec_kernel(const float* __restrict__ slabs, int nslabs, float* __restrict__ dst,
                                                                      int Cout, int Cin, int KS, int Cout_p, int Cin_p, int half, int half_p) {
  const int cin4 = Cin >> 2;
  const size_t total = (size_t)Cout * cin4 * KS;
  const size_t slab = (size_t)KS * Cout_p * Cin_p;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    size_t q = i;
    const int c4 = q % cin4; q /= cin4;
    const int co = q % Cout; q /= Cout;
    const int tap = (int)q;
    const size_t src = ((size_t)tap * Cout_p + glu_map(co, half, half_p)) * Cin_p + 4 * c4;
    float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
    int k = 0;
    for (; k + 8 <= nslabs; k += 8) {
      float4 v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const float4*>(slabs + (size_t)(k + j) * slab + src);
#pragma unroll
      for (int j = 0; j < 8; ++j) { sum.x += v[j].x; sum.y += v[j].y; sum.z += v[j].z; sum.w += v[j].w; }
    }
    for (; k < nslabs; ++k) {
      const float4 v = *reinterpret_cast<const float4*>(slabs + (size_t)k * slab + src);
      sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
    }
    float* o = dst + ((size_t)co * Cin + 4 * c4) * KS + tap;
    o[0] = sum.x; o[KS] = sum.y; o[2 * KS] = sum.z; o[3 * KS] = sum.w;
  }
}

__global__ void pack_vector_kernel(const float* __restrict__ v, float* __restrict__ dst, int C, int Cp, int half, int half_p) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= Cp) return;
  const int c = glu_unmap(i, C, half, half_p);
  dst[i] = c >= 0 ? v[c] : 0.f;
}
__global__ void unpack_vector_kernel(const float* __restrict__ g, float* __restrict__ dst, int C, int half, int half_p) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < C) dst[i] = g[glu_map(i, half, half_p)];
}

// ------------------------------------------------------------------------------------------------
// BatchNorm1d: finalize / apply+GELU / backward   (models.py:135,143,158,161)
// ------------------------------------------------------------------------------------------------
// Sum partial[k][which][c] over k for the 8 channels of this block: 128 thread groups (2 lanes x float4
// each) stride over k, fp64 accumulation, fixed combination order (deterministic).  Result: channel
// 8*blockIdx.x + threadIdx.x on threads 0..7.
__device__ inline void block_partial_sums(const float* __restrict__ partial, int n, int Cp, int cbase, bool two,
                                          double& s0, double& s1) {
  // LDS image [which][j][thread]: every store and every load below is 64 consecutive 8-byte slots per wave (the channel-major
  // [row group][8 channels] image of rounds 1-4 put 8 lanes on one bank: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.77)
  __shared__ double sh[2][4][256];
  const int tx = threadIdx.x & 1, ty = threadIdx.x >> 1;
  double a0[4] = {0.0, 0.0, 0.0, 0.0}, a1[4] = {0.0, 0.0, 0.0, 0.0};
  const int c4 = cbase + tx * 4;
  if (c4 < Cp) {
#pragma unroll 4
    for (int k = ty; k < n; k += 128) {
      const float4 u = *reinterpret_cast<const float4*>(partial + ((size_t)k * 2 + 0) * Cp + c4);
      a0[0] += (double)u.x; a0[1] += (double)u.y; a0[2] += (double)u.z; a0[3] += (double)u.w;
      if (two) {
        const float4 v = *reinterpret_cast<const float4*>(partial + ((size_t)k * 2 + 1) * Cp + c4);
        a1[0] += (double)v.x; a1[1] += (double)v.y; a1[2] += (double)v.z; a1[3] += (double)v.w;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) { sh[0][j][threadIdx.x] = a0[j]; sh[1][j][threadIdx.x] = a1[j]; }
  __syncthreads();
  // waves 0 and 1 (which = 0, 1): lane l adds, per j, the entries of threads l, l + 64, l + 128, l + 192 (same tx = l & 1), then
  // a fixed shuffle tree over the 32 lanes of its parity: lanes 0 / 1 end with channels j / 4 + j
  __shared__ double res[2][8];
  if (threadIdx.x < 128) {
    const int l = threadIdx.x & 63, which = threadIdx.x >> 6;
    double a[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      a[j] = (sh[which][j][l] + sh[which][j][l + 64]) + (sh[which][j][l + 128] + sh[which][j][l + 192]);
      a[j] += __shfl_xor(a[j], 2);
      a[j] += __shfl_xor(a[j], 4);
      a[j] += __shfl_xor(a[j], 8);
      a[j] += __shfl_xor(a[j], 16);
      a[j] += __shfl_xor(a[j], 32);
    }
    if (l < 2) {
#pragma unroll
      for (int j = 0; j < 4; ++j) res[which][l * 4 + j] = a[j];
    }
  }
  __syncthreads();
  s0 = 0.0; s1 = 0.0;
  if (threadIdx.x < 8) { s0 = res[0][threadIdx.x]; s1 = res[1][threadIdx.x]; }
}

__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ partial, int ntiles, double count,
                                   const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                   float momentum, float* running_mean, float* running_var, float* mean_o,
                                   float* rstd_o, float* scale_o, float* shift_o, float* bwd_coef, int C, int Cp,
                                   int training, long* batches_tracked) {
  // nn.BatchNorm1d.num_batches_tracked += 1 (models.py:135,143: track_running_stats) — here, not in a launch of its own
  if (batches_tracked && blockIdx.x == 0 && threadIdx.x == 0) *batches_tracked += 1;
  const int c = blockIdx.x * 8 + (threadIdx.x & 7);
  double s = 0.0, q = 0.0;
  if (training) block_partial_sums(partial, ntiles, Cp, blockIdx.x * 8, true, s, q);
  if (threadIdx.x >= 8 || c >= Cp) return;
  if (c >= C) {
    mean_o[c] = 0.f; rstd_o[c] = 0.f; scale_o[c] = 0.f; shift_o[c] = 0.f;
    if (bwd_coef) { bwd_coef[c] = 0.f; bwd_coef[Cp + c] = 0.f; bwd_coef[2 * Cp + c] = 0.f; bwd_coef[3 * Cp + c] = 0.f; }
    return;
  }
  double mean, var;
  if (training) {
    mean = s / count;
    var = q / count - mean * mean;
    if (var < 0.0) var = 0.0;
    if (running_mean) {
      const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
      running_mean[c] = (float)((1.0 - momentum) * (double)running_mean[c] + momentum * mean);
      running_var[c] = (float)((1.0 - momentum) * (double)running_var[c] + momentum * unbiased);
    }
  } else {
    mean = running_mean[c];
    var = running_var[c];
  }
  const double rstd = 1.0 / sqrt(var + (double)eps);
  const double sc = (double)gamma[c] * rstd;
  mean_o[c] = (float)mean;
  rstd_o[c] = (float)rstd;
  scale_o[c] = (float)sc;
  shift_o[c] = (float)((double)beta[c] - mean * sc);
  if (bwd_coef) {
    bwd_coef[c] = gamma[c]; bwd_coef[Cp + c] = beta[c]; bwd_coef[2 * Cp + c] = (float)mean; bwd_coef[3 * Cp + c] = (float)rstd;
  }
}

template <typename E>
__global__ __launch_bounds__(256) void bn_gelu_fwd_kernel(const E* __restrict__ x, E* __restrict__ y,
                                                          const float* __restrict__ scale,
                                                          const float* __restrict__ shift, int B, int T, int Cp) {
  constexpr int CH = Vec16<E>::N, U = 4;
  const int nch = Cp / CH;
  const int RG = nch >= 256 ? 1 : 256 / nch;
  int r0, r1;
  block_rows(B, T, r0, r1);
  for (int c = threadIdx.x; c < RG * nch; c += 256) {       // one pass unless a row has more than 256 chunks
    const int ch = c % nch, rg = c / nch;
    float sc[CH], sh[CH];
#pragma unroll
    for (int q4 = 0; q4 < CH / 4; ++q4) {
      *reinterpret_cast<float4*>(sc + q4 * 4) = *reinterpret_cast<const float4*>(scale + ch * CH + q4 * 4);
      *reinterpret_cast<float4*>(sh + q4 * 4) = *reinterpret_cast<const float4*>(shift + ch * CH + q4 * 4);
    }
    auto one = [&](float* v) {
#pragma unroll
      for (int j = 0; j < CH; j += 2) {
        const f32x2 o = gelu_pair<E>(fma2(f32x2{v[j], v[j + 1]}, f32x2{sc[j], sc[j + 1]}, f32x2{sh[j], sh[j + 1]}));
        v[j] = o.x; v[j + 1] = o.y;
      }
    };
    RowCursor w(r0, r1, rg, RG, T, Cp, ch * CH);
    for (; w.n >= U; w.n -= U) {
      uint32_t off[U];
      float v[U][CH];
#pragma unroll
      for (int u = 0; u < U; ++u) off[u] = w.take();
#pragma unroll
      for (int u = 0; u < U; ++u) Vec16<E>::load(x + off[u], v[u]);
#pragma unroll
      for (int u = 0; u < U; ++u) { one(v[u]); Vec16<E>::store(y + off[u], v[u]); }
    }
    for (; w.n > 0; --w.n) {
      const uint32_t off = w.take();
      float v[CH];
      Vec16<E>::load(x + off, v);
      one(v);
      Vec16<E>::store(y + off, v);
    }
  }
}

constexpr int RED_MAX_BLOCKS = 1024;

// Column reductions over valid rows.  MODE 0: sum x (bias grads).  MODE 1: BN+GELU backward sums
// (dg, dg * xhat) with dg = dy * GELU'(gamma * xhat + beta).
template <typename E, int MODE>
__global__ __launch_bounds__(256) void col_reduce_kernel(const E* __restrict__ dy, const E* __restrict__ x,
                                                         const float* __restrict__ mean, const float* __restrict__ rstd,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         int C, float* __restrict__ partial, int B, int T, int Cp) {
  extern __shared__ __attribute__((aligned(16))) float red[];
  constexpr int CH = Vec16<E>::N;
  const int nch = Cp / CH;
  const int RG = 256 / nch;
  const int ch = threadIdx.x % nch, rg = threadIdx.x / nch;
  const size_t rows = (size_t)B * T;
  const size_t per = (rows + gridDim.x - 1) / gridDim.x;
  const size_t r0 = (size_t)blockIdx.x * per, r1 = min(rows, r0 + per);
  float a0[CH], a1[CH];
#pragma unroll
  for (int j = 0; j < CH; ++j) { a0[j] = 0.f; a1[j] = 0.f; }
  if (rg < RG) {
    float mu[CH], rs[CH], ga[CH], be[CH];
    if (MODE == 1) {
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        const int c = ch * CH + j;
        mu[j] = mean[c]; rs[j] = rstd[c];
        ga[j] = c < C ? gamma[c] : 0.f; be[j] = c < C ? beta[c] : 0.f;
      }
    }
#pragma unroll 2
    for (RowWalk w((int)r0 + rg, (int)r1, RG, T); w.valid(); w.next()) {
      const size_t off = w.mem_row() * Cp + ch * CH;
      float dv[CH];
      Vec16<E>::load(dy + off, dv);
      if (MODE == 0) {
#pragma unroll
        for (int j = 0; j < CH; ++j) a0[j] += dv[j];
      } else {
        float xv[CH];
        Vec16<E>::load(x + off, xv);
#pragma unroll
        for (int j = 0; j < CH; j += 2) {
          const f32x2 xh = (f32x2{xv[j], xv[j + 1]} - f32x2{mu[j], mu[j + 1]}) * f32x2{rs[j], rs[j + 1]};
          const f32x2 dg = f32x2{dv[j], dv[j + 1]} * gelu_grad_pair<E>(fma2(f32x2{ga[j], ga[j + 1]}, xh, f32x2{be[j], be[j + 1]}));
          const f32x2 dgx = dg * xh;
          a0[j] += dg.x; a0[j + 1] += dg.y;
          a1[j] += dgx.x; a1[j + 1] += dgx.y;
        }
      }
    }
  }
  // the per-row-group sums meet in LDS as [row group][which][16-byte quarter q][chunk] float4: a wave's store is 64 consecutive
  // 16-byte slots (conflict-free; channel-major rows of CH floats per lane were 2- to 4-way conflicted)
  if (rg < RG) {
#pragma unroll
    for (int q4 = 0; q4 < CH / 4; ++q4) {
      *reinterpret_cast<float4*>(red + ((((rg * 2 + 0) * (CH / 4) + q4) * nch + ch) << 2)) = make_float4(a0[4 * q4], a0[4 * q4 + 1], a0[4 * q4 + 2], a0[4 * q4 + 3]);
      *reinterpret_cast<float4*>(red + ((((rg * 2 + 1) * (CH / 4) + q4) * nch + ch) << 2)) = make_float4(a1[4 * q4], a1[4 * q4 + 1], a1[4 * q4 + 2], a1[4 * q4 + 3]);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * Cp; i += 256) {
    const int which = i / Cp, c = i - which * Cp;
    const int chc = c / CH, q4 = (c % CH) >> 2, e = c & 3;
    float s = 0.f;
    for (int g = 0; g < RG; ++g) s += red[((((g * 2 + which) * (CH / 4) + q4) * nch + chc) << 2) + e];
    partial[((size_t)blockIdx.x * 2 + which) * Cp + c] = s;
  }
}

// padding (in float4) behind a row group of bwd_colsum_kernel's LDS image: (2 * (CH / 4) * nch + pad) = nch (mod 16)
__host__ __device__ inline int bwd_colsum_pad(int nch, int CH) { return (((nch - 2 * (CH / 4) * nch) % 16) + 16) % 16; }
static inline size_t bwd_colsum_lds(int Ch, int dtype) {
  const int CH = dtype == SDA_F32 ? 4 : 8, nch = Ch / CH, RG = 256 / nch;
  return (size_t)RG * (2 * (CH / 4) * nch + bwd_colsum_pad(nch, CH)) * 16;
}

// Backward elementwise stage fused with the column sums of its own output (bias gradients of the layer
// below), same thread->channel ownership as col_reduce_kernel.
//   MODE 0 (GELU):  du = dz * GELU'(u);  partial[blk][0][c] = sum du
//   MODE 1 (GLU):   x = [a | g] (2*Ch channels), dy (Ch): da = dy*sig(g), dg = dy*a*sig(g)*(1-sig(g));
//                   partial[blk][0][c] = sum da, partial[blk][1][c] = sum dg      (c < Ch)
//   MODE 2 (GLU after a fused forward, SDA_EPI_GLU): x = out = a*sig(g) (Ch channels), gate = g (Ch channels):
//                   da = dy*sig(g), dg = dy*out*(1-sig(g)); same outputs as MODE 1 (the value half is never stored)
template <typename E, int MODE>
__global__ __launch_bounds__(256) void bwd_colsum_kernel(const E* __restrict__ x, const E* __restrict__ dy,
                                                         E* __restrict__ dx, float* __restrict__ partial, int B, int T,
                                                         int Ch, const E* __restrict__ gate = nullptr) {
  ew_bwd_prio();
  extern __shared__ __attribute__((aligned(16))) float red[];
  constexpr int CH = Vec16<E>::N;
  const int nch = Ch / CH;
  const int RG = 256 / nch;
  const int ch = threadIdx.x % nch, rg = threadIdx.x / nch;
  const size_t rows = (size_t)B * T;
  const size_t per = (rows + gridDim.x - 1) / gridDim.x;
  const size_t r0 = (size_t)blockIdx.x * per, r1 = min(rows, r0 + per);
  const int xw = MODE == 1 ? 2 * Ch : Ch, dxw = MODE == 0 ? Ch : 2 * Ch;
  float a0[CH], a1[CH];
#pragma unroll
  for (int j = 0; j < CH; ++j) { a0[j] = 0.f; a1[j] = 0.f; }
  if (rg < RG) {
    RowCursor w((int)r0, (int)r1, rg, RG, T, Ch, ch * CH);       // offsets in Ch-wide rows; 2*Ch-wide rows: 2 * off - ch * CH
#pragma unroll 2
    for (; w.n > 0; --w.n) {
      const uint32_t off = w.take(), off2 = 2 * off - ch * CH;
      float d[CH], xv[CH], o0[CH], o1[CH];
      Vec16<E>::load(dy + off, d);
      Vec16<E>::load(x + (MODE == 1 ? off2 : off), xv);
      if (MODE == 0) {
#pragma unroll
        for (int j = 0; j < CH; j += 2) {
          const f32x2 o = f32x2{d[j], d[j + 1]} * gelu_grad_pair<E>(f32x2{xv[j], xv[j + 1]});
          o0[j] = o.x; o0[j + 1] = o.y;
          a0[j] += o.x; a0[j + 1] += o.y;
        }
        Vec16<E>::store(dx + off, o0);
      } else {
        float g[CH];
        if (MODE == 1) Vec16<E>::load(x + off2 + Ch, g);
        else Vec16<E>::load(gate + off, g);
#pragma unroll
        for (int j = 0; j < CH; ++j) {
          const float sg = sigmoid_f(g[j]);
          o0[j] = d[j] * sg;
          o1[j] = MODE == 1 ? d[j] * xv[j] * sg * (1.f - sg) : d[j] * xv[j] * (1.f - sg);
          a0[j] += o0[j];
          a1[j] += o1[j];
        }
        Vec16<E>::store(dx + off2, o0);
        Vec16<E>::store(dx + off2 + Ch, o1);
      }
    }
  }
  // (LDS image as in col_reduce_kernel: [row group][which][quarter][chunk] float4; a row group's base is shifted so that
  // base(rg + 1) = base(rg) + nch (mod 16 float4): the 16 lanes of a store that straddle two row groups — 320 channels: nch = 40,
  // lanes 32..47 = chunks 32..39 of one group and 0..7 of the next — then still take 16 distinct 16-byte slots of a bank row)
  const int rgs = 2 * (CH / 4) * nch + bwd_colsum_pad(nch, CH);            // float4 per row group
  if (rg < RG) {
#pragma unroll
    for (int q4 = 0; q4 < CH / 4; ++q4) {
      *reinterpret_cast<float4*>(red + ((rg * rgs + (0 * (CH / 4) + q4) * nch + ch) << 2)) = make_float4(a0[4 * q4], a0[4 * q4 + 1], a0[4 * q4 + 2], a0[4 * q4 + 3]);
      *reinterpret_cast<float4*>(red + ((rg * rgs + (1 * (CH / 4) + q4) * nch + ch) << 2)) = make_float4(a1[4 * q4], a1[4 * q4 + 1], a1[4 * q4 + 2], a1[4 * q4 + 3]);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * Ch; i += 256) {
    const int which = i / Ch, c = i - which * Ch;
    const int chc = c / CH, q4 = (c % CH) >> 2, e = c & 3;
    float sum = 0.f;
    for (int g = 0; g < RG; ++g) sum += red[((g * rgs + (which * (CH / 4) + q4) * nch + chc) << 2) + e];
    partial[((size_t)blockIdx.x * 2 + which) * Ch + c] = sum;
  }
}

// sums[which][c] = sum over blocks (fp64, fixed order)
__global__ __launch_bounds__(256) void col_reduce_final_kernel(const float* __restrict__ partial, int nblocks,
                                                               float* __restrict__ out0, float* __restrict__ out1, int Cp) {
  ew_bwd_prio();
  const int c = blockIdx.x * 8 + (threadIdx.x & 7);
  double s0, s1;
  block_partial_sums(partial, nblocks, Cp, blockIdx.x * 8, out1 != nullptr, s0, s1);
  if (threadIdx.x >= 8 || c >= Cp) return;
  out0[c] = (float)s0;
  if (out1) out1[c] = (float)s1;
}

// per-channel coefficients of the BN+GELU backward, precomputed once per launch into a [6][Cp] table:
//   0: gamma  1: beta  2: mean  3: rstd  4: dbeta/N  5: dgamma/N
__global__ void bn_bwd_coef_kernel(const float* __restrict__ mean, const float* __restrict__ rstd,
                                   const float* __restrict__ gamma, const float* __restrict__ beta, int C,
                                   const float* __restrict__ dbeta, const float* __restrict__ dgamma, float inv_count,
                                   float* __restrict__ coef, int Cp) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= Cp) return;
  coef[0 * Cp + c] = c < C ? gamma[c] : 0.f;
  coef[1 * Cp + c] = c < C ? beta[c] : 0.f;
  coef[2 * Cp + c] = mean[c];
  coef[3 * Cp + c] = rstd[c];
  coef[4 * Cp + c] = dbeta[c] * inv_count;
  coef[5 * Cp + c] = dgamma[c] * inv_count;
}

// reduce_stats + bn_bwd_coef in one launch (single-process case: no all-reduce sits between them)
__global__ __launch_bounds__(256) void bn_bwd_stats_coef_kernel(const float* __restrict__ partial, int nrows,
                                                                const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                int C, float inv_count, float* __restrict__ dbeta,
                                                                float* __restrict__ dgamma, float* __restrict__ coef, int Cp) {
  ew_bwd_prio();
  const int c = blockIdx.x * 8 + (threadIdx.x & 7);
  double s0, s1;
  block_partial_sums(partial, nrows, Cp, blockIdx.x * 8, true, s0, s1);
  if (threadIdx.x >= 8 || c >= Cp) return;
  const float db = (float)s0, dg = (float)s1;
  dbeta[c] = db;
  dgamma[c] = dg;
  coef[0 * Cp + c] = c < C ? gamma[c] : 0.f;
  coef[1 * Cp + c] = c < C ? beta[c] : 0.f;
  coef[2 * Cp + c] = mean[c];
  coef[3 * Cp + c] = rstd[c];
  coef[4 * Cp + c] = db * inv_count;
  coef[5 * Cp + c] = dg * inv_count;
}

// DG: `dy` already holds dg = dy * GELU'(gamma * xhat + beta) (a conv epilogue with SDA_EPI_BN_STORE_DG wrote it): the affine
// part only — 5 vector instructions per element pair instead of ~25
template <typename E, bool DG = false>
__global__ __launch_bounds__(256) void bn_gelu_bwd_apply_kernel(const E* __restrict__ dy, const E* __restrict__ x,
                                                                const float* __restrict__ coef, E* __restrict__ dx,
                                                                int B, int T, int Cp) {
  ew_bwd_prio();
  constexpr int CH = Vec16<E>::N, U = 2;
  const int nch = Cp / CH;
  const int RG = nch >= 256 ? 1 : 256 / nch;
  int r0, r1;
  block_rows(B, T, r0, r1);
  for (int c = threadIdx.x; c < RG * nch; c += 256) {
    const int ch = c % nch, rg = c / nch;
    // per channel: a = gamma*rstd, b = beta - a*mean (so gamma*xhat + beta = a*x + b), and the linear part of
    //   dx = gamma*rstd*(g - dbeta/N - xhat*dgamma/N) = a*g - (p + q*x),  q = a*rstd*dgamma/N,  p = a*dbeta/N - q*mean
    float ca[CH], cb[CH], cp[CH], cq[CH];
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      const int k = ch * CH + j;
      const float ga = coef[k], be = coef[Cp + k], mu = coef[2 * Cp + k], rs = coef[3 * Cp + k];
      ca[j] = ga * rs;
      cb[j] = be - ca[j] * mu;
      cq[j] = ca[j] * rs * coef[5 * Cp + k];
      cp[j] = ca[j] * coef[4 * Cp + k] - cq[j] * mu;
    }
    auto one = [&](float* d, const float* xv) {
#pragma unroll
      for (int j = 0; j < CH; j += 2) {
        const f32x2 xx = {xv[j], xv[j + 1]}, a2 = {ca[j], ca[j + 1]};
        f32x2 g = f32x2{d[j], d[j + 1]};
        if constexpr (!DG) g = g * gelu_grad_pair<E>(fma2(a2, xx, f32x2{cb[j], cb[j + 1]}));
        const f32x2 o = fma2(a2, g, -fma2(f32x2{cq[j], cq[j + 1]}, xx, f32x2{cp[j], cp[j + 1]}));
        d[j] = o.x; d[j + 1] = o.y;
      }
    };
    RowCursor w(r0, r1, rg, RG, T, Cp, ch * CH);
    for (; w.n >= U; w.n -= U) {
      uint32_t off[U];
      float d[U][CH], xv[U][CH];
#pragma unroll
      for (int u = 0; u < U; ++u) off[u] = w.take();
#pragma unroll
      for (int u = 0; u < U; ++u) { Vec16<E>::load(dy + off[u], d[u]); Vec16<E>::load(x + off[u], xv[u]); }
#pragma unroll
      for (int u = 0; u < U; ++u) { one(d[u], xv[u]); Vec16<E>::store(dx + off[u], d[u]); }
    }
    for (; w.n > 0; --w.n) {
      const uint32_t off = w.take();
      float d[CH], xv[CH];
      Vec16<E>::load(dy + off, d);
      Vec16<E>::load(x + off, xv);
      one(d, xv);
      Vec16<E>::store(dx + off, d);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// GLU (models.py:164) and GELU backward
// ------------------------------------------------------------------------------------------------
template <typename E>
__global__ __launch_bounds__(256) void glu_fwd_kernel(const E* __restrict__ x, E* __restrict__ y, int B, int T, int Ch) {
  constexpr int CH = Vec16<E>::N, U = 2;
  const int nch = Ch / CH;
  const int RG = nch >= 256 ? 1 : 256 / nch;
  int r0, r1;
  block_rows(B, T, r0, r1);
  for (int c = threadIdx.x; c < RG * nch; c += 256) {
    const int ch = c % nch, rg = c / nch;
    RowCursor w(r0, r1, rg, RG, T, Ch, ch * CH);            // offsets in the Ch-wide output; the input row is twice as wide
    auto one = [&](uint32_t off) {
      float a[CH], g[CH];
      const uint32_t xo = 2 * off - ch * CH;                // row * 2 Ch + ch * CH
      Vec16<E>::load(x + xo, a);
      Vec16<E>::load(x + xo + Ch, g);
#pragma unroll
      for (int j = 0; j < CH; ++j) a[j] *= sigmoid_f(g[j]);
      Vec16<E>::store(y + off, a);
    };
    for (; w.n >= U; w.n -= U) {
      uint32_t off[U];
#pragma unroll
      for (int u = 0; u < U; ++u) off[u] = w.take();
#pragma unroll
      for (int u = 0; u < U; ++u) one(off[u]);
    }
    for (; w.n > 0; --w.n) one(w.take());
  }
}

template <typename E>
__global__ __launch_bounds__(256) void glu_bwd_kernel(const E* __restrict__ x, const E* __restrict__ dy,
                                                      E* __restrict__ dx, int B, int T, int Ch) {
  constexpr int CH = Vec16<E>::N;
  const int nch = Ch / CH;
  const int RG = nch >= 256 ? 1 : 256 / nch;
  int r0, r1;
  block_rows(B, T, r0, r1);
  for (int c = threadIdx.x; c < RG * nch; c += 256) {
    const int ch = c % nch, rg = c / nch;
    for (RowWalk w(r0 + rg, r1, RG, T); w.valid(); w.next()) {
      const size_t row = w.mem_row();
      float a[CH], g[CH], d[CH];
      Vec16<E>::load(x + row * 2 * Ch + ch * CH, a);
      Vec16<E>::load(x + row * 2 * Ch + Ch + ch * CH, g);
      Vec16<E>::load(dy + row * Ch + ch * CH, d);
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        const float sg = sigmoid_f(g[j]);
        g[j] = d[j] * a[j] * sg * (1.f - sg);               // same expression order as bwd_colsum_kernel<1> (bit-equal outputs)
        a[j] = d[j] * sg;
      }
      Vec16<E>::store(dx + row * 2 * Ch + ch * CH, a);
      Vec16<E>::store(dx + row * 2 * Ch + Ch + ch * CH, g);
    }
  }
}

template <typename E>
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const E* __restrict__ u, const E* __restrict__ dz,
                                                       E* __restrict__ du, int B, int T, int Cp) {
  constexpr int CH = Vec16<E>::N;
  const int nch = Cp / CH;
  const int RG = nch >= 256 ? 1 : 256 / nch;
  int r0, r1;
  block_rows(B, T, r0, r1);
  for (int c = threadIdx.x; c < RG * nch; c += 256) {
    const int ch = c % nch, rg = c / nch;
    for (RowWalk w(r0 + rg, r1, RG, T); w.valid(); w.next()) {
      const size_t off = w.mem_row() * Cp + ch * CH;
      float uv[CH], d[CH];
      Vec16<E>::load(u + off, uv);
      Vec16<E>::load(dz + off, d);
#pragma unroll
      for (int j = 0; j < CH; j += 2) {
        const f32x2 o = f32x2{d[j], d[j + 1]} * gelu_grad_pair<E>(f32x2{uv[j], uv[j + 1]});
        d[j] = o.x; d[j + 1] = o.y;
      }
      Vec16<E>::store(du + off, d);
    }
  }
}

__global__ void reduce_slabs_kernel(const float* __restrict__ src, float* __restrict__ dst, int nslabs, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float s = 0.f;
    int k = 0;
    for (; k + 8 <= nslabs; k += 8) {                     // 8 loads in flight; added in slab order
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = src[(size_t)(k + j) * n + i];
#pragma unroll
      for (int j = 0; j < 8; ++j) s += v[j];
    }
    for (; k < nslabs; ++k) s += src[(size_t)k * n + i];
    dst[i] = s;
  }
}

static inline int ew_grid(size_t total) {
  size_t g = (total + 255) / 256;
  return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

static int red_blocks(int B, int T) {
  long rows = (long)B * T;
  long nb = (rows + 31) / 32;
  if (nb > RED_MAX_BLOCKS) nb = RED_MAX_BLOCKS;
  if (nb < 1) nb = 1;
  return (int)nb;
}

}  // namespace sda

using namespace sda;

#define SDA_DISPATCH(dtype, CALL)                                   \
  do {                                                              \
    if ((dtype) == SDA_F32) { using E = float; CALL; }              \
    else if ((dtype) == SDA_BF16) { using E = uint16_t; CALL; }     \
    else if ((dtype) == SDA_F16) { using E = half_t; CALL; }        \
    else { set_error("unknown dtype %d", (int)(dtype)); return -1; } \
  } while (0)

namespace sda {
// rows of a row-layout buffer that no kernel writes but every conv reads as zeros: the PAD rows in front of each sample and
// the slack behind the last one (16-byte stores; a row is Cp * sizeof(E) bytes, a multiple of 128)
__global__ __launch_bounds__(256) void zero_pad_rows_kernel(uint4* __restrict__ buf, int B, int Tp, long rows_alloc, int row_vec) {
  const long pad_vecs = (long)B * PAD * row_vec, tail_vecs = (rows_alloc - (long)B * Tp) * row_vec;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < pad_vecs + tail_vecs; i += (long)gridDim.x * 256) {
    long v;
    if (i < pad_vecs) { const long b = i / ((long)PAD * row_vec), r = i - b * PAD * row_vec; v = b * Tp * row_vec + r; }
    else v = (long)B * Tp * row_vec + (i - pad_vecs);
    buf[v] = make_uint4(0u, 0u, 0u, 0u);
  }
}
// zero `n16` 16-byte vectors + `tail` trailing bytes
__global__ __launch_bounds__(256) void fill_zero_kernel(uint4* __restrict__ p, long n16, int tail) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long)gridDim.x * 256) p[i] = make_uint4(0u, 0u, 0u, 0u);
  if (blockIdx.x == 0 && (int)threadIdx.x < tail) reinterpret_cast<unsigned char*>(p + n16)[threadIdx.x] = 0;
}
// dst sample b = table sample idx[b]: whole samples of `vecs` 16-byte vectors (a row-layout sample is (T + PAD) * Cp contiguous
// elements, pad rows included, so a gathered batch IS a row-layout buffer); blockIdx.y = b
__global__ __launch_bounds__(256) void gather_samples_kernel(const uint4* __restrict__ table, const long* __restrict__ idx,
                                                             uint4* __restrict__ dst, long vecs) {
  const uint4* __restrict__ src = table + (size_t)idx[blockIdx.y] * vecs;
  uint4* __restrict__ out = dst + (size_t)blockIdx.y * vecs;
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  const long step = (long)gridDim.x * 256;
  for (; i + 3 * step < vecs; i += 4 * step) {            // four loads in flight per lane
    const uint4 a = src[i], b = src[i + step], c = src[i + 2 * step], d = src[i + 3 * step];
    out[i] = a; out[i + step] = b; out[i + 2 * step] = c; out[i + 3 * step] = d;
  }
  for (; i < vecs; i += step) out[i] = src[i];
}
__global__ void scalar_mul_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = a[i] * b[0];
}
}  // namespace sda

extern "C" int sda_zero_pad_rows(void* buf, int B, int T, int Cp, int dtype, void* stream) {
  if (!buf || B < 1 || T < 1 || Cp % 64) { set_error("zero_pad_rows: bad arguments"); return -1; }
  const int row_vec = Cp * (dtype == SDA_F32 ? 4 : 2) / 16;
  const long n = ((long)B * PAD + (rows_alloc(B, T) - (long)B * rows_tp(T))) * row_vec;
  const long blocks = (n + 255) / 256;
  hipLaunchKernelGGL(zero_pad_rows_kernel, dim3((unsigned)(blocks < 1024 ? blocks : 1024)), dim3(256), 0, (hipStream_t)stream,
                     (uint4*)buf, B, rows_tp(T), rows_alloc(B, T), row_vec);
  return check_launch("zero_pad_rows");
}

extern "C" int sda_fill_zero(void* p, long nbytes, void* stream) {
  if (!p || nbytes < 1 || (reinterpret_cast<uintptr_t>(p) & 15)) { set_error("fill_zero: null, empty or unaligned buffer"); return -1; }
  const long n16 = nbytes / 16;
  const long blocks = (n16 + 255) / 256;
  hipLaunchKernelGGL(fill_zero_kernel, dim3((unsigned)(blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks))), dim3(256), 0, (hipStream_t)stream,
                     (uint4*)p, n16, (int)(nbytes - n16 * 16));
  return check_launch("fill_zero");
}

extern "C" int sda_gather_samples(const void* table, const long* idx, void* dst, int B, long sample_bytes, void* stream) {
  if (!table || !idx || !dst || B < 1 || sample_bytes < 16 || sample_bytes % 16 || (reinterpret_cast<uintptr_t>(table) & 15) ||
      (reinterpret_cast<uintptr_t>(dst) & 15)) {
    set_error("gather_samples: bad arguments (samples are whole 16-byte vectors)"); return -1;
  }
  const long vecs = sample_bytes / 16;
  long bx = (vecs + 256 * 4 - 1) / (256 * 4);
  const long want = (8L * launch_cus() + B - 1) / B;      // ~8 workgroups per CU over the whole batch
  if (bx > want) bx = want;
  if (bx < 1) bx = 1;
  hipLaunchKernelGGL(gather_samples_kernel, dim3((unsigned)bx, (unsigned)B), dim3(256), 0, (hipStream_t)stream,
                     (const uint4*)table, idx, (uint4*)dst, vecs);
  return check_launch("gather_samples");
}

extern "C" int sda_scalar_mul(const float* a, const float* b, float* out, int n, void* stream) {
  if (!a || !b || !out || n < 1) { set_error("scalar_mul: bad arguments"); return -1; }
  hipLaunchKernelGGL(scalar_mul_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, a, b, out, n);
  return check_launch("scalar_mul");
}

extern "C" int sda_pack_rows(const float* src, void* dst, int B, int C, int T, int Cp, int dtype, void* stream) {
  if (!src || !dst || Cp % 64 || C > Cp || B < 1) { set_error("pack_rows: bad arguments"); return -1; }
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((T + 63) / 64, Cp / 64, B);
  if (T % 4 == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0) {
    SDA_DISPATCH(dtype, hipLaunchKernelGGL(pack_rows_vec_kernel<E>, grid, dim3(256), 0, st, src, (E*)dst, C, T, Cp, -1));
  } else {
    SDA_DISPATCH(dtype, hipLaunchKernelGGL(pack_rows_kernel<E>, grid, dim3(256), 0, st, src, (E*)dst, C, T, Cp, -1));
  }
  return check_launch("pack_rows");
}

extern "C" int sda_pack_rows_ones(const float* src, void* dst, int B, int C, int T, int Cp, int ones_channel, int dtype,
                                  void* stream) {
  if (!src || !dst || Cp % 64 || C > Cp || B < 1 || ones_channel < C || ones_channel >= Cp) {
    set_error("pack_rows_ones: bad arguments (the constant channel must be a padding channel)"); return -1;
  }
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((T + 63) / 64, Cp / 64, B);
  if (T % 4 == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0) {
    SDA_DISPATCH(dtype, hipLaunchKernelGGL(pack_rows_vec_kernel<E>, grid, dim3(256), 0, st, src, (E*)dst, C, T, Cp, ones_channel));
  } else {
    SDA_DISPATCH(dtype, hipLaunchKernelGGL(pack_rows_kernel<E>, grid, dim3(256), 0, st, src, (E*)dst, C, T, Cp, ones_channel));
  }
  return check_launch("pack_rows_ones");
}

extern "C" int sda_unpack_rows(const void* src, float* dst, int B, int C, int T, int Cp, int dtype, void* stream) {
  if (!src || !dst || Cp % 64 || C > Cp || B < 1) { set_error("unpack_rows: bad arguments"); return -1; }
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((T + 63) / 64, Cp / 64, B);
  SDA_DISPATCH(dtype, hipLaunchKernelGGL(unpack_rows_kernel<E>, grid, dim3(256), 0, st, (const E*)src, dst, C, T, Cp));
  return check_launch("unpack_rows");
}

extern "C" int sda_rows_sumsq_from_stats(const float* stats, int tiles_per_sample, int Cp, float* out, int B, void* stream) {
  if (!stats || !out || tiles_per_sample < 1 || Cp < 1 || B < 1) { set_error("rows_sumsq_from_stats: bad arguments"); return -1; }
  hipLaunchKernelGGL(rows_sumsq_from_stats_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, stats, tiles_per_sample, Cp, out);
  return check_launch("rows_sumsq_from_stats");
}

extern "C" int sda_rows_sumsq_from_row_parts(const float* parts, int n_parts, float* out, int B, int T, void* stream) {
  if (!parts || !out || n_parts < 1 || B < 1 || T < 1) { set_error("rows_sumsq_from_row_parts: bad arguments"); return -1; }
  hipLaunchKernelGGL(rows_sumsq_from_row_parts_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, parts, n_parts, T, out);
  return check_launch("rows_sumsq_from_row_parts");
}

extern "C" int sda_rows_sumsq(const void* x, float* out, float* scratch, int B, long row_elems, long pitch,
                              int dtype, void* stream) {
  if (!x || !out || !scratch || row_elems % 4 || pitch % 4 || B < 1) { set_error("rows_sumsq: bad arguments"); return -1; }
  hipStream_t st = (hipStream_t)stream;
  SDA_DISPATCH(dtype, hipLaunchKernelGGL(rows_sumsq_kernel<E>, dim3(SUMSQ_CHUNKS, B), dim3(256), 0, st,
                                         (const E*)x, scratch, row_elems, pitch));
  hipLaunchKernelGGL(rows_sumsq_final_kernel, dim3((B + 63) / 64), dim3(64), 0, st, scratch, out, B);
  return check_launch("rows_sumsq");
}

extern "C" int sda_pack_conv_weight(const float* w, void* dst, int nW, int Cout, int Cin, int KS, int Cout_p,
                                    int Cin_p, int mode, int glu_half, int glu_half_p, int dtype, void* stream) {
  if (!w || !dst || nW < 1 || Cout > Cout_p || Cin > Cin_p || (mode != 0 && mode != 1)) { set_error("pack_conv_weight: bad arguments"); return -1; }
  if (glu_half && (glu_half_p < glu_half || glu_half_p + (Cout - glu_half) > Cout_p)) { set_error("pack_conv_weight: bad GLU split"); return -1; }
  hipStream_t st = (hipStream_t)stream;
  const size_t total = (size_t)nW * KS * Cout_p * Cin_p;
  SDA_DISPATCH(dtype, hipLaunchKernelGGL(pack_conv_weight_kernel<E>, dim3(ew_grid(total)), dim3(256), 0, st, w,
                                         (E*)dst, nW, Cout, Cin, KS, Cout_p, Cin_p, mode, glu_half, glu_half_p));
  return check_launch("pack_conv_weight");
}

extern "C" int sda_unpack_conv_wgrad(const float* g, float* dst, int nW, int Cout, int Cin, int KS, int Cout_p,
                                     int Cin_p, int glu_half, int glu_half_p, void* stream) {
  if (!g || !dst) { set_error("unpack_conv_wgrad: null"); return -1; }
  const size_t total = (size_t)nW * Cout * Cin * KS;
  hipLaunchKernelGGL(unpack_conv_wgrad_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, g, dst, nW,
                     Cout, Cin, KS, Cout_p, Cin_p, glu_half, glu_half_p);
  return check_launch("unpack_conv_wgrad");
}

// Adam (train.py:161-163: torch.optim.Adam defaults, no weight decay / amsgrad) over ALL parameter tensors in
// one launch: blockIdx.y selects the tensor, complex parameters are updated through their real view.
__global__ __launch_bounds__(256) void adam_multi_kernel(const sda_adam_desc* __restrict__ descs, float lr, float beta1,
                                                         float beta2, float eps, float bc1, float bc2_sqrt) {
  const sda_adam_desc d = descs[blockIdx.y];
  const float step_size = lr / bc1;
  for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < d.n; i += (long)gridDim.x * 1024) {
    if (i + 4 <= d.n && d.aligned) {
      float4 p = *reinterpret_cast<float4*>(d.param + i);
      const float4 g = *reinterpret_cast<const float4*>(d.grad + i);
      float4 m = *reinterpret_cast<float4*>(d.exp_avg + i), v = *reinterpret_cast<float4*>(d.exp_avg_sq + i);
#define SDA_ADAM(f)                                                     \
      m.f = beta1 * m.f + (1.f - beta1) * g.f;                          \
      v.f = beta2 * v.f + (1.f - beta2) * g.f * g.f;                    \
      p.f -= step_size * (m.f / (sqrtf(v.f) / bc2_sqrt + eps));
      SDA_ADAM(x) SDA_ADAM(y) SDA_ADAM(z) SDA_ADAM(w)
#undef SDA_ADAM
      *reinterpret_cast<float4*>(d.param + i) = p;
      *reinterpret_cast<float4*>(d.exp_avg + i) = m;
      *reinterpret_cast<float4*>(d.exp_avg_sq + i) = v;
    } else {
      for (long j = i; j < min(i + 4, d.n); ++j) {
        const float g = d.grad[j];
        const float m = beta1 * d.exp_avg[j] + (1.f - beta1) * g;
        const float v = beta2 * d.exp_avg_sq[j] + (1.f - beta2) * g * g;
        d.exp_avg[j] = m;
        d.exp_avg_sq[j] = v;
        d.param[j] -= step_size * (m / (sqrtf(v) / bc2_sqrt + eps));
      }
    }
  }
}

extern "C" int sda_adam_multi(const sda_adam_desc* descs_dev, int n, long max_n, float lr, float beta1, float beta2, float eps,
                              long step, void* stream) {
  if (!descs_dev || n < 1 || max_n < 1 || step < 1) { set_error("adam_multi: bad arguments"); return -1; }
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  long gx = (max_n + 1023) / 1024;
  if (gx > 256) gx = 256;
  hipLaunchKernelGGL(adam_multi_kernel, dim3((unsigned)gx, (unsigned)n), dim3(256), 0, (hipStream_t)stream, descs_dev, lr, beta1,
                     beta2, eps, (float)bc1, (float)sqrt(bc2));
  return check_launch("adam_multi");
}

extern "C" int sda_pack_multi(const sda_pack_desc* descs_dev, int n, long max_total, int dtype, void* stream) {
  if (!descs_dev || n < 1 || max_total < 1) { set_error("pack_multi: bad arguments"); return -1; }
  hipStream_t st = (hipStream_t)stream;
  long gx = (max_total + 1023) / 1024;
  if (gx > 512) gx = 512;
  SDA_DISPATCH(dtype, hipLaunchKernelGGL(pack_multi_kernel<E>, dim3((unsigned)gx, (unsigned)n), dim3(256), 0, st, descs_dev));
  return check_launch("pack_multi");
}

extern "C" int sda_reduce_unpack_wgrad(const float* slabs, int nslabs, float* dst, int Cout, int Cin, int KS, int Cout_p,
                                       int Cin_p, int glu_half, int glu_half_p, void* stream) {
  if (!slabs || !dst || nslabs < 1) { set_error("reduce_unpack_wgrad: bad arguments"); return -1; }
  const size_t total = (size_t)Cout * Cin * KS;
  if (Cin % 4 == 0 && Cin_p % 4 == 0 && (reinterpret_cast<uintptr_t>(slabs) & 15) == 0) {
    hipLaunchKernelGGL(reduce_unpack_wgrad_vec_kernel, dim3(ew_grid(total / 4)), dim3(256), 0, (hipStream_t)stream, slabs, nslabs, dst,
                       Cout, Cin, KS, Cout_p, Cin_p, glu_half, glu_half_p);
    return check_launch("reduce_unpack_wgrad");
  }
  hipLaunchKernelGGL(reduce_unpack_wgrad_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, slabs, nslabs, dst,
                     Cout, Cin, KS, Cout_p, Cin_p, glu_half, glu_half_p);
  return check_launch("reduce_unpack_wgrad");
}

extern "C" int sda_pack_vector(const float* v, float* dst, int C, int Cp, int glu_half, int glu_half_p, void* stream) {
  if (!v || !dst || C > Cp) { set_error("pack_vector: bad arguments"); return -1; }
  hipLaunchKernelGGL(pack_vector_kernel, dim3((Cp + 255) / 256), dim3(256), 0, (hipStream_t)stream, v, dst, C, Cp,
                     glu_half, glu_half_p);
  return check_launch("pack_vector");
}

extern "C" int sda_unpack_vector(const float* g, float* dst, int C, int Cp, int glu_half, int glu_half_p, void* stream) {
  if (!g || !dst || C > Cp) { set_error("unpack_vector: bad arguments"); return -1; }
  hipLaunchKernelGGL(unpack_vector_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, g, dst, C,
                     glu_half, glu_half_p);
  return check_launch("unpack_vector");
}

extern "C" int sda_bn_finalize(const float* partial, int ntiles, double count, const float* gamma, const float* beta,
                               float eps, float momentum, float* running_mean, float* running_var, float* mean,
                               float* rstd, float* scale, float* shift, float* bwd_coef, int C, int Cp, int training,
                               long* batches_tracked, void* stream) {
  if (!gamma || !beta || !mean || !rstd || !scale || !shift || C > Cp) { set_error("bn_finalize: bad arguments"); return -1; }
  if (training && (!partial || ntiles < 1 || count < 1.0)) { set_error("bn_finalize: training mode needs partial statistics"); return -1; }
  if (!training && (!running_mean || !running_var)) { set_error("bn_finalize: eval mode needs running statistics"); return -1; }
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((Cp + 7) / 8), dim3(256), 0, (hipStream_t)stream, partial, ntiles, count,
                     gamma, beta, eps, momentum, running_mean, running_var, mean, rstd, scale, shift, bwd_coef, C, Cp,
                     training, training ? batches_tracked : nullptr);
  return check_launch("bn_finalize");
}

extern "C" int sda_bn_gelu_forward(const void* x, void* y, const float* scale, const float* shift, int B, int T,
                                   int Cp, int dtype, void* stream) {
  if (!x || !y || !scale || !shift || Cp % 64 || !fits_u32(B, T, Cp)) { set_error("bn_gelu_forward: bad arguments"); return -1; }
  SDA_DISPATCH(dtype, hipLaunchKernelGGL(bn_gelu_fwd_kernel<E>, dim3(stream_blocks(B, T, Cp / Vec16<E>::N)), dim3(256), 0, (hipStream_t)stream,
                                         (const E*)x, (E*)y, scale, shift, B, T, Cp));
  return check_launch("bn_gelu_forward");
}


extern "C" int sda_bn_gelu_backward_reduce(const void* dy, const void* x, const float* mean, const float* rstd,
                                           const float* gamma, const float* beta, int C, float* partial, float* dgamma,
                                           float* dbeta, int B, int T, int Cp, int dtype, void* stream) {
  if (!dy || !x || !mean || !rstd || !gamma || !beta || !partial || !dgamma || !dbeta || Cp % 64 || Cp > 1024) {
    set_error("bn_gelu_backward_reduce: bad arguments"); return -1;
  }
  hipStream_t st = (hipStream_t)stream;
  const int nb = red_blocks(B, T);
  const int RG = 256 / (Cp / (dtype == SDA_F32 ? 4 : 8));
  const size_t lds = (size_t)RG * 2 * Cp * sizeof(float);
  SDA_DISPATCH(dtype, hipLaunchKernelGGL((col_reduce_kernel<E, 1>), dim3(nb), dim3(256), lds, st, (const E*)dy,
                                         (const E*)x, mean, rstd, gamma, beta, C, partial, B, T, Cp));
  hipLaunchKernelGGL(col_reduce_final_kernel, dim3((Cp + 7) / 8), dim3(256), 0, st, partial, nb, dbeta, dgamma, Cp);
  return check_launch("bn_gelu_backward_reduce");
}

extern "C" int sda_reduce_stats(const float* partial, int nrows, float* out0, float* out1, int Cp, void* stream) {
  if (!partial || !out0 || nrows < 1 || Cp % 8) { set_error("reduce_stats: bad arguments"); return -1; }
  hipLaunchKernelGGL(col_reduce_final_kernel, dim3((Cp + 7) / 8), dim3(256), 0, (hipStream_t)stream, partial, nrows,
                     out0, out1, Cp);
  return check_launch("reduce_stats");
}

static int bn_backward_apply(bool dg, const void* dy, const void* x, const float* mean, const float* rstd,
                             const float* gamma, const float* beta, int C, const float* dgamma,
                             const float* dbeta, double count, float* coef, void* dx, int B, int T, int Cp,
                             int dtype, void* stream) {
  if (!dy || !x || !mean || !rstd || !gamma || !beta || !dgamma || !dbeta || !coef || !dx || Cp % 64 || count < 1.0 || !fits_u32(B, T, Cp)) {
    set_error("bn_gelu_backward_apply: bad arguments"); return -1;
  }
  const float inv_count = (float)(1.0 / count);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(bn_bwd_coef_kernel, dim3((Cp + 255) / 256), dim3(256), 0, st, mean, rstd, gamma, beta, C, dbeta,
                     dgamma, inv_count, coef, Cp);
  if (dg) {
    SDA_DISPATCH(dtype, hipLaunchKernelGGL((bn_gelu_bwd_apply_kernel<E, true>), dim3(stream_blocks(B, T, Cp / Vec16<E>::N)), dim3(256), 0, st,
                                           (const E*)dy, (const E*)x, coef, (E*)dx, B, T, Cp));
  } else {
    SDA_DISPATCH(dtype, hipLaunchKernelGGL((bn_gelu_bwd_apply_kernel<E, false>), dim3(stream_blocks(B, T, Cp / Vec16<E>::N)), dim3(256), 0, st,
                                           (const E*)dy, (const E*)x, coef, (E*)dx, B, T, Cp));
  }
  return check_launch("bn_gelu_backward_apply");
}

extern "C" int sda_bn_gelu_backward_apply(const void* dy, const void* x, const float* mean, const float* rstd,
                                          const float* gamma, const float* beta, int C, const float* dgamma,
                                          const float* dbeta, double count, float* coef, void* dx, int B, int T, int Cp,
                                          int dtype, void* stream) {
  return bn_backward_apply(false, dy, x, mean, rstd, gamma, beta, C, dgamma, dbeta, count, coef, dx, B, T, Cp, dtype, stream);
}
extern "C" int sda_bn_gelu_backward_apply_dg(const void* dg, const void* x, const float* mean, const float* rstd,
                                             const float* gamma, const float* beta, int C, const float* dgamma,
                                             const float* dbeta, double count, float* coef, void* dx, int B, int T, int Cp,
                                             int dtype, void* stream) {
  return bn_backward_apply(true, dg, x, mean, rstd, gamma, beta, C, dgamma, dbeta, count, coef, dx, B, T, Cp, dtype, stream);
}

static int bn_backward_from_stats(bool dg, const float* partial, int nrows, const void* dy, const void* x,
                                  const float* mean, const float* rstd, const float* gamma,
                                  const float* beta, int C, double count, float* dgamma, float* dbeta,
                                  float* coef, void* dx, int B, int T, int Cp, int dtype, void* stream) {
  if (!partial || nrows < 1 || !dy || !x || !mean || !rstd || !gamma || !beta || !dgamma || !dbeta || !coef || !dx ||
      Cp % 64 || count < 1.0 || !fits_u32(B, T, Cp)) {
    set_error("bn_gelu_backward_from_stats: bad arguments"); return -1;
  }
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(bn_bwd_stats_coef_kernel, dim3((Cp + 7) / 8), dim3(256), 0, st, partial, nrows, mean, rstd, gamma, beta, C,
                     (float)(1.0 / count), dbeta, dgamma, coef, Cp);
  if (dg) {
    SDA_DISPATCH(dtype, hipLaunchKernelGGL((bn_gelu_bwd_apply_kernel<E, true>), dim3(stream_blocks(B, T, Cp / Vec16<E>::N)), dim3(256), 0, st,
                                           (const E*)dy, (const E*)x, coef, (E*)dx, B, T, Cp));
  } else {
    SDA_DISPATCH(dtype, hipLaunchKernelGGL((bn_gelu_bwd_apply_kernel<E, false>), dim3(stream_blocks(B, T, Cp / Vec16<E>::N)), dim3(256), 0, st,
                                           (const E*)dy, (const E*)x, coef, (E*)dx, B, T, Cp));
  }
  return check_launch("bn_gelu_backward_from_stats");
}

extern "C" int sda_bn_gelu_backward_from_stats(const float* partial, int nrows, const void* dy, const void* x,
                                              const float* mean, const float* rstd, const float* gamma,
                                              const float* beta, int C, double count, float* dgamma, float* dbeta,
                                              float* coef, void* dx, int B, int T, int Cp, int dtype, void* stream) {
  return bn_backward_from_stats(false, partial, nrows, dy, x, mean, rstd, gamma, beta, C, count, dgamma, dbeta, coef, dx, B, T, Cp, dtype, stream);
}
extern "C" int sda_bn_gelu_backward_from_stats_dg(const float* partial, int nrows, const void* dg, const void* x,
                                                 const float* mean, const float* rstd, const float* gamma,
                                                 const float* beta, int C, double count, float* dgamma, float* dbeta,
                                                 float* coef, void* dx, int B, int T, int Cp, int dtype, void* stream) {
  return bn_backward_from_stats(true, partial, nrows, dg, x, mean, rstd, gamma, beta, C, count, dgamma, dbeta, coef, dx, B, T, Cp, dtype, stream);
}

extern "C" int sda_colsum(const void* x, float* out, float* scratch, int B, int T, int Cp, int dtype, void* stream) {
  if (!x || !out || !scratch || Cp % 64 || Cp > 1024) { set_error("colsum: bad arguments"); return -1; }
  hipStream_t st = (hipStream_t)stream;
  const int nb = red_blocks(B, T);
  const int RG = 256 / (Cp / (dtype == SDA_F32 ? 4 : 8));
  const size_t lds = (size_t)RG * 2 * Cp * sizeof(float);
  SDA_DISPATCH(dtype, hipLaunchKernelGGL((col_reduce_kernel<E, 0>), dim3(nb), dim3(256), lds, st, (const E*)x,
                                         (const E*)nullptr, nullptr, nullptr, nullptr, nullptr, 0, scratch, B, T, Cp));
  hipLaunchKernelGGL(col_reduce_final_kernel, dim3((Cp + 7) / 8), dim3(256), 0, st, scratch, nb, out, (float*)nullptr, Cp);
  return check_launch("colsum");
}

extern "C" int sda_reduce_scratch_floats(int Cp) { return RED_MAX_BLOCKS * 2 * Cp; }
extern "C" int sda_reduce_scratch_rows(int B, int T) { return red_blocks(B, T); }

extern "C" int sda_glu_forward(const void* x, void* y, int B, int T, int Ch, int dtype, void* stream) {
  if (!x || !y || Ch % 64 || !fits_u32(B, T, 2L * Ch)) { set_error("glu_forward: bad arguments"); return -1; }
  SDA_DISPATCH(dtype, hipLaunchKernelGGL(glu_fwd_kernel<E>, dim3(stream_blocks(B, T, Ch / Vec16<E>::N)), dim3(256), 0, (hipStream_t)stream,
                                         (const E*)x, (E*)y, B, T, Ch));
  return check_launch("glu_forward");
}

extern "C" int sda_glu_backward(const void* x, const void* dy, void* dx, int B, int T, int Ch, int dtype, void* stream) {
  if (!x || !dy || !dx || Ch % 64) { set_error("glu_backward: bad arguments"); return -1; }
  SDA_DISPATCH(dtype, hipLaunchKernelGGL(glu_bwd_kernel<E>, dim3(stream_blocks(B, T, Ch / Vec16<E>::N)), dim3(256), 0, (hipStream_t)stream,
                                         (const E*)x, (const E*)dy, (E*)dx, B, T, Ch));
  return check_launch("glu_backward");
}

extern "C" int sda_gelu_backward(const void* u, const void* dz, void* du, int B, int T, int Cp, int dtype, void* stream) {
  if (!u || !dz || !du || Cp % 64) { set_error("gelu_backward: bad arguments"); return -1; }
  SDA_DISPATCH(dtype, hipLaunchKernelGGL(gelu_bwd_kernel<E>, dim3(stream_blocks(B, T, Cp / Vec16<E>::N)), dim3(256), 0, (hipStream_t)stream,
                                         (const E*)u, (const E*)dz, (E*)du, B, T, Cp));
  return check_launch("gelu_backward");
}

extern "C" int sda_glu_backward_colsum(const void* x, const void* dy, void* dx, float* colsum, float* scratch, int B, int T,
                                       int Ch, int dtype, void* stream) {
  if (!x || !dy || !dx || !scratch || Ch % 64 || Ch > 1024 || !fits_u32(B, T, 2L * Ch)) { set_error("glu_backward_colsum: bad arguments"); return -1; }
  hipStream_t st = (hipStream_t)stream;
  const int nb = red_blocks(B, T);
  const size_t lds = bwd_colsum_lds(Ch, dtype);
  SDA_DISPATCH(dtype, hipLaunchKernelGGL((bwd_colsum_kernel<E, 1>), dim3(nb), dim3(256), lds, st, (const E*)x,
                                         (const E*)dy, (E*)dx, scratch, B, T, Ch));
  if (colsum) hipLaunchKernelGGL(col_reduce_final_kernel, dim3((Ch + 7) / 8), dim3(256), 0, st, scratch, nb, colsum, colsum + Ch, Ch);
  return check_launch("glu_backward_colsum");
}

extern "C" int sda_glu_backward_colsum_og(const void* out, const void* gate, const void* dy, void* dx, float* colsum,
                                          float* scratch, int B, int T, int Ch, int dtype, void* stream) {
  if (!out || !gate || !dy || !dx || !scratch || Ch % 64 || Ch > 1024 || !fits_u32(B, T, 2L * Ch)) { set_error("glu_backward_colsum_og: bad arguments"); return -1; }
  hipStream_t st = (hipStream_t)stream;
  const int nb = red_blocks(B, T);
  const size_t lds = bwd_colsum_lds(Ch, dtype);
  SDA_DISPATCH(dtype, hipLaunchKernelGGL((bwd_colsum_kernel<E, 2>), dim3(nb), dim3(256), lds, st, (const E*)out,
                                         (const E*)dy, (E*)dx, scratch, B, T, Ch, (const E*)gate));
  if (colsum) hipLaunchKernelGGL(col_reduce_final_kernel, dim3((Ch + 7) / 8), dim3(256), 0, st, scratch, nb, colsum, colsum + Ch, Ch);
  return check_launch("glu_backward_colsum_og");
}

extern "C" int sda_gelu_backward_colsum(const void* u, const void* dz, void* du, float* colsum, float* scratch, int B, int T,
                                        int Cp, int dtype, void* stream) {
  if (!u || !dz || !du || !scratch || Cp % 64 || Cp > 1024 || !fits_u32(B, T, Cp)) { set_error("gelu_backward_colsum: bad arguments"); return -1; }
  hipStream_t st = (hipStream_t)stream;
  const int nb = red_blocks(B, T);
  const size_t lds = bwd_colsum_lds(Cp, dtype);
  SDA_DISPATCH(dtype, hipLaunchKernelGGL((bwd_colsum_kernel<E, 0>), dim3(nb), dim3(256), lds, st, (const E*)u,
                                         (const E*)dz, (E*)du, scratch, B, T, Cp));
  if (colsum) hipLaunchKernelGGL(col_reduce_final_kernel, dim3((Cp + 7) / 8), dim3(256), 0, st, scratch, nb, colsum, (float*)nullptr, Cp);
  return check_launch("gelu_backward_colsum");
}

extern "C" int sda_reduce_slabs(const float* src, float* dst, int nslabs, long n, void* stream) {
  if (!src || !dst || nslabs < 1 || n < 1) { set_error("reduce_slabs: bad arguments"); return -1; }
  hipLaunchKernelGGL(reduce_slabs_kernel, dim3(ew_grid((size_t)n)), dim3(256), 0, (hipStream_t)stream, src, dst, nslabs, n);
  return check_launch("reduce_slabs");
}

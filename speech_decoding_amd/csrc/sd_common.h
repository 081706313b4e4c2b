// Common device/host helpers for the gfx950 kernels of the contrastive training path.
// CDNA4 only: 64-lane wavefronts, MFMA 16x16 tiles, 128-byte swizzled LDS rows.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/sd_amd.h"

namespace sda {

// ------------------------------------------------------------------------------------------
// Activation "row layout" (RL).  A tensor (B, C, T) lives as rows of Cp channels (channels-last):
//   row(b, t) = b * Tp + PAD + t,   Tp = T + PAD,   PAD = 16 (= the largest dilation)
// The PAD rows in front of every sample (and behind the last one) are always zero, so a dilated tap
// that leaves [0, T) reads zeros without any masking.  Channel padding (Cp - C) is zero as well.
// R_alloc adds slack behind the last sample so that a 128-row tile may overrun harmlessly.
// ------------------------------------------------------------------------------------------
constexpr int PAD = SDA_ROW_PAD;
constexpr int TILE_T = 128;               // output rows per workgroup (conv_gemm)
constexpr int ROW_SLACK = TILE_T + 2 * PAD;

__host__ __device__ inline int rows_tp(int T) { return T + PAD; }
__host__ __device__ inline long rows_alloc(int B, int T) { return (long)B * rows_tp(T) + PAD + ROW_SLACK; }

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef _Float16 half_t;                                    // fp16 storage element (bf16 storage is uint16_t)
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;

// bf16 <-> f32 (round to nearest even; NaN stays NaN through the compiler's cvt)
__device__ inline float bf2f(uint16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ inline uint16_t f2bf(float f) {
  __bf16 b = (__bf16)f;
  return *reinterpret_cast<uint16_t*>(&b);
}

// two floats -> two bf16 in one register (ONE v_cvt_pk_bf16_f32, round to nearest even; element 0 in the low half)
__device__ inline uint32_t pack_bf16x2(float a, float b) {
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  const bf16x2_t p = __builtin_convertvector(f32x2_t{a, b}, bf16x2_t);
  return __builtin_bit_cast(uint32_t, p);
}

// two fp16 values packed in one register <-> floats (v_cvt_f32_f16 / v_cvt_f16_f32, round to nearest even)
__device__ inline float2 h2_to_f2(uint32_t w) {
  const f16x2 h = __builtin_bit_cast(f16x2, w);
  return make_float2((float)h[0], (float)h[1]);
}
__device__ inline uint32_t f2_to_h2(float a, float b) {
  f16x2 h;
  h[0] = (_Float16)a;
  h[1] = (_Float16)b;
  return __builtin_bit_cast(uint32_t, h);
}

// Element traits: E = float (exact fp32 path on the f32 MFMA), uint16_t (bf16 storage) or half_t (fp16 storage).
template <typename E> struct Elem;
template <> struct Elem<float> {
  static constexpr int PER16 = 4;      // elements per 16-byte chunk
  static constexpr int SLAB = 32;      // elements per 128-byte LDS row
  __device__ static float ld(const float* p) { return *p; }
  __device__ static void st(float* p, float v) { *p = v; }
};
template <> struct Elem<uint16_t> {
  static constexpr int PER16 = 8;
  static constexpr int SLAB = 64;
  __device__ static float ld(const uint16_t* p) { return bf2f(*p); }
  __device__ static void st(uint16_t* p, float v) { *p = f2bf(v); }
};

template <> struct Elem<half_t> {
  static constexpr int PER16 = 8;
  static constexpr int SLAB = 64;
  __device__ static float ld(const half_t* p) { return (float)*p; }
  __device__ static void st(half_t* p, float v) { *p = (half_t)v; }
};

// 4 consecutive elements <-> float4 (16-byte fp32 / 8-byte bf16 accesses)
__device__ inline float4 load4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ inline float4 load4(const uint16_t* p) {
  uint2 u = *reinterpret_cast<const uint2*>(p);
  return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u),
                     __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
}
__device__ inline void store4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ inline void store4(uint16_t* p, float4 v) {
  uint2 u;
  u.x = (uint32_t)f2bf(v.x) | ((uint32_t)f2bf(v.y) << 16);
  u.y = (uint32_t)f2bf(v.z) | ((uint32_t)f2bf(v.w) << 16);
  *reinterpret_cast<uint2*>(p) = u;
}

__device__ inline float4 load4(const half_t* p) {
  const uint2 u = *reinterpret_cast<const uint2*>(p);
  const float2 a = h2_to_f2(u.x), b = h2_to_f2(u.y);
  return make_float4(a.x, a.y, b.x, b.y);
}
__device__ inline void store4(half_t* p, float4 v) {
  uint2 u;
  u.x = f2_to_h2(v.x, v.y);
  u.y = f2_to_h2(v.z, v.w);
  *reinterpret_cast<uint2*>(p) = u;
}

// 16 bytes of consecutive elements <-> floats: 4 fp32 or 8 bf16 (the widest global access per lane)
template <typename E> struct Vec16;
template <> struct Vec16<float> {
  static constexpr int N = 4;
  __device__ static void load(const float* p, float* v) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  }
  __device__ static void store(float* p, const float* v) { *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]); }
  __device__ static float round(float x) { return x; }
  __device__ static uint4 load_raw(const float* p) { return *reinterpret_cast<const uint4*>(p); }
  __device__ static void unpack(const uint4& u, float* v) {
    v[0] = __uint_as_float(u.x); v[1] = __uint_as_float(u.y); v[2] = __uint_as_float(u.z); v[3] = __uint_as_float(u.w);
  }
};
template <> struct Vec16<uint16_t> {
  static constexpr int N = 8;
  __device__ static void load(const uint16_t* p, float* v) {
    const uint4 u = *reinterpret_cast<const uint4*>(p);
    const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[2 * i] = __uint_as_float(w[i] << 16); v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
  }
  __device__ static void store(uint16_t* p, const float* v) {
    uint32_t w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = pack_bf16x2(v[2 * i], v[2 * i + 1]);
    *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
  }
  __device__ static float round(float x) { return bf2f(f2bf(x)); }
  // 16 bytes kept packed (4 registers) until use: halves the registers of prefetched rows
  __device__ static uint4 load_raw(const uint16_t* p) { return *reinterpret_cast<const uint4*>(p); }
  __device__ static void unpack(const uint4& u, float* v) {
    const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[2 * i] = __uint_as_float(w[i] << 16); v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
  }
};

template <> struct Vec16<half_t> {
  static constexpr int N = 8;
  __device__ static void unpack(const uint4& u, float* v) {
    const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) { const float2 f = h2_to_f2(w[i]); v[2 * i] = f.x; v[2 * i + 1] = f.y; }
  }
  __device__ static void load(const half_t* p, float* v) { unpack(*reinterpret_cast<const uint4*>(p), v); }
  __device__ static void store(half_t* p, const float* v) {
    *reinterpret_cast<uint4*>(p) = make_uint4(f2_to_h2(v[0], v[1]), f2_to_h2(v[2], v[3]), f2_to_h2(v[4], v[5]), f2_to_h2(v[6], v[7]));
  }
  __device__ static float round(float x) { return (float)(half_t)x; }
  __device__ static uint4 load_raw(const half_t* p) { return *reinterpret_cast<const uint4*>(p); }
};

// Byte offset of 16-byte chunk `chunk` (0..7) of 128-byte LDS row `row`, XOR-swizzled so that 16
// lanes reading the same chunk of 16 consecutive rows (the MFMA operand pattern) hit 16 distinct
// 16-byte slots of the 256-byte bank row.
__device__ inline int lds_sw(int row, int chunk) { return row * 128 + (((chunk ^ (row >> 1)) & 7) << 4); }

// One 64-byte K-step (4 lane groups x 16 bytes) of MFMA work on a 16x16 output tile.
// bf16: one v_mfma_f32_16x16x32_bf16.  fp32: four v_mfma_f32_16x16x4_f32; lane group g supplies
// k = 4g + j at step j for BOTH operands, so the contraction covers the same 16 k values.
template <typename E> __device__ inline f32x4 mma16(const uint4& a, const uint4& b, f32x4 c);
template <> __device__ inline f32x4 mma16<uint16_t>(const uint4& a, const uint4& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const s16x8*>(&a),
                                                 *reinterpret_cast<const s16x8*>(&b), c, 0, 0, 0);
}
template <> __device__ inline f32x4 mma16<half_t>(const uint4& a, const uint4& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(*reinterpret_cast<const f16x8*>(&a), *reinterpret_cast<const f16x8*>(&b), c, 0, 0, 0);
}
template <> __device__ inline f32x4 mma16<float>(const uint4& a, const uint4& b, f32x4 c) {
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), c, 0, 0, 0);
  return c;
}

// The 32 x 32 output block form: v_mfma_f32_32x32x16 (16-bit storage only).  Lane l supplies row (l & 31) of either operand,
// k = 8 * (l >> 5) + 0..7 — one 16-byte chunk; a 64-byte K-step is two of them (chunks {0, 1}, then {2, 3}).  Register v of the
// result: row 8 * (v >> 2) + 4 * (l >> 5) + (v & 3) of the FIRST operand's block, column l & 31 = row of the second operand's.
// Why: one wave issues a 16x16x32 every ~24 clocks although the instruction occupies the pipe for 16 (bare loop, one wave per
// SIMD: 1.61 PFLOP/s; two waves 2.0; four 2.2), a 32x32x16 every 32 = its pipe time (2.45 PFLOP/s from ONE wave per SIMD):
// tools/probes/sclk_probe.hip.
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8_t;
template <typename E> __device__ inline f32x16 mma32(const uint4& a, const uint4& b, f32x16 c);
template <> __device__ inline f32x16 mma32<uint16_t>(const uint4& a, const uint4& b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8_t*>(&a), *reinterpret_cast<const bf16x8_t*>(&b), c, 0, 0, 0);
}
template <> __device__ inline f32x16 mma32<half_t>(const uint4& a, const uint4& b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const f16x8*>(&a), *reinterpret_cast<const f16x8*>(&b), c, 0, 0, 0);
}

// One MFMA row: acc[n] += a x b[n] for n < N.  16-bit types: N MFMAs.  fp32: the 16-deep K-step is four 16x16x4 MFMAs per
// accumulator — issued j-OUTER, n-inner, so that consecutive MFMAs are independent and each accumulator is updated in place
// (the accumulation order per accumulator is unchanged: j = 0, 1, 2, 3 — bit-equal results).  Written n-outer (four
// back-to-back MFMAs chained through one accumulator, as mma16<float> does on its own) hipcc routes every chain through a
// temporary register quad: two of every four MFMAs read a SrcC that is not exactly the previous MFMA's vDst, and the matrix
// pipe waits out the dependency (cdna3 ISA, XDL write -> SrcC read, 8 passes) — the exact path's k = 3 convs sat at 57 % of
// the fp32 matrix peak.
template <typename E, int N> __device__ __forceinline__ void mma16_row(const uint4& a, const uint4 (&b)[N], f32x4 (&acc)[N]) {
  if constexpr (sizeof(E) == 4) {
    const uint32_t av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int n = 0; n < N; ++n) {
        const uint32_t bj = j == 0 ? b[n].x : (j == 1 ? b[n].y : (j == 2 ? b[n].z : b[n].w));
        acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(av[j]), __uint_as_float(bj), acc[n], 0, 0, 0);
      }
  } else {
#pragma unroll
    for (int n = 0; n < N; ++n) acc[n] = mma16<E>(a, b[n], acc[n]);
  }
}
// The same for an M x N block of accumulators (wgrad_gemm: M row fragments against N column fragments)
template <typename E, int M, int N> __device__ __forceinline__ void mma16_block(const uint4 (&a)[M], const uint4 (&b)[N], f32x4 (&acc)[M][N]) {
  if constexpr (sizeof(E) == 4) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int m = 0; m < M; ++m) {
        const uint32_t aj = j == 0 ? a[m].x : (j == 1 ? a[m].y : (j == 2 ? a[m].z : a[m].w));
#pragma unroll
        for (int n = 0; n < N; ++n) {
          const uint32_t bj = j == 0 ? b[n].x : (j == 1 ? b[n].y : (j == 2 ? b[n].z : b[n].w));
          acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(aj), __uint_as_float(bj), acc[m][n], 0, 0, 0);
        }
      }
  } else {
#pragma unroll
    for (int m = 0; m < M; ++m)
#pragma unroll
      for (int n = 0; n < N; ++n) acc[m][n] = mma16<E>(a[m], b[n], acc[m][n]);
  }
}

// GELU (erf form, F.gelu default — models.py:158,161,194,195) and its derivative, by storage type.
// fp32 storage: libm erff, the exact path.  16-bit storage: the normal tail 0.5*erfc(|x|/sqrt2) by Abramowitz-Stegun
// 26.2.17 (|error| <= 7.5e-8 absolute; 2^-9 / 2^-11 is the storage resolution) — one v_rcp + one v_exp shared by the
// CDF and the density.  The passes that apply it are VALU-bound, not HBM-bound, unless it stays under ~13 issue slots
// per element (measured: affine only 19.7 us, this form written per element 26.6 us for 118 MB), so it is written on
// PAIRS of elements: every multiply/add is a v_pk_*_f32 (two lanes of work per issue slot), and
// GELU = max(x,0) - |x|*tail needs no compare/select.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ inline f32x2 fma2(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ inline f32x2 splat2(float v) { return f32x2{v, v}; }
// tail = 0.5*erfc(|x|/sqrt2) = 1 - Phi(|x|),  e = exp(-x^2/2)
__device__ inline void normal_tail2(f32x2 x, f32x2& tail, f32x2& e) {
#ifdef SDA_FAKE_GELU          // diagnostic build (wrong results): what the step would cost if GELU / GELU' were free
  tail = x * 0.01f; e = x * 0.02f;
  return;
#endif
  const f32x2 y = x * 0.849321800288f;                                // sqrt(log2(e)/2): exp(-x^2/2) = 2^-(y*y)
  const f32x2 yy = y * y;
  e = f32x2{__builtin_amdgcn_exp2f(-yy.x), __builtin_amdgcn_exp2f(-yy.y)};
  const f32x2 t = {__builtin_amdgcn_rcpf(__builtin_fmaf(__builtin_fabsf(x.x), 0.2316419f, 1.0f)),
                   __builtin_amdgcn_rcpf(__builtin_fmaf(__builtin_fabsf(x.y), 0.2316419f, 1.0f))};
  f32x2 q = fma2(t, splat2(0.5f * 1.061405429f), splat2(0.5f * -1.453152027f));   // 26.2.17's b5..b1 = 7.1.26's a5..a1 halved
  q = fma2(q, t, splat2(0.5f * 1.421413741f));
  q = fma2(q, t, splat2(0.5f * -0.284496736f));
  q = fma2(q, t, splat2(0.5f * 0.254829592f));
  tail = (q * t) * e;
}
__device__ inline f32x2 gelu2(f32x2 x) {                              // x*Phi(x) = max(x,0) - |x|*tail
  f32x2 tail, e;
  normal_tail2(x, tail, e);
  return f32x2{__builtin_fmaf(-__builtin_fabsf(x.x), tail.x, __builtin_fmaxf(x.x, 0.f)),
               __builtin_fmaf(-__builtin_fabsf(x.y), tail.y, __builtin_fmaxf(x.y, 0.f))};
}
__device__ inline f32x2 gelu_grad2(f32x2 x) {                         // Phi(x) + x*phi(x)
  f32x2 tail, e;
  normal_tail2(x, tail, e);
  const f32x2 h = splat2(0.5f) - tail;                                // Phi(x) = 0.5 + sign(x)*(0.5 - tail)
  const f32x2 cdf = f32x2{__builtin_copysignf(h.x, x.x), __builtin_copysignf(h.y, x.y)} + splat2(0.5f);
  return fma2(x * 0.3989422804014327f, e, cdf);
}
template <typename E> __device__ inline float gelu_f(float x);
template <typename E> __device__ inline float gelu_grad_f(float x);
template <> __device__ inline float gelu_f<float>(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
template <> __device__ inline float gelu_grad_f<float>(float x) {
  const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752f));
  const float pdf = 0.3989422804014327f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}
template <> __device__ inline float gelu_f<uint16_t>(float x) { return gelu2(splat2(x)).x; }
template <> __device__ inline float gelu_grad_f<uint16_t>(float x) { return gelu_grad2(splat2(x)).x; }
// fp16 storage: the same form (its absolute error is below half an fp16 ulp for |GELU| > 3e-4 and below the smallest
// normal fp16 step elsewhere)
template <> __device__ inline float gelu_f<half_t>(float x) { return gelu_f<uint16_t>(x); }
template <> __device__ inline float gelu_grad_f<half_t>(float x) { return gelu_grad_f<uint16_t>(x); }
// element pairs (the elementwise passes): 16-bit storage on the packed form, fp32 storage per element on erff
template <typename E> __device__ inline f32x2 gelu_pair(f32x2 x) { return gelu2(x); }
template <typename E> __device__ inline f32x2 gelu_grad_pair(f32x2 x) { return gelu_grad2(x); }
template <> __device__ inline f32x2 gelu_pair<float>(f32x2 x) { return f32x2{gelu_f<float>(x.x), gelu_f<float>(x.y)}; }
template <> __device__ inline f32x2 gelu_grad_pair<float>(f32x2 x) { return f32x2{gelu_grad_f<float>(x.x), gelu_grad_f<float>(x.y)}; }
__device__ inline float sigmoid_f(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }   // v_rcp_f32: 1 ulp

__device__ inline float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ inline float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

// LDS-DMA (global_load_lds_dwordx4) issued through inline asm.  hipcc treats the builtin form as an LDS
// write that may alias every later ds_read of the same array and drains it with s_waitcnt vmcnt(0) before the
// first read — which serialises the copy of slab s+1 against the MFMAs of slab s.  The asm form is invisible
// to that bookkeeping: completion is enforced by OUR counted s_waitcnt vmcnt(N) + barrier.
//   gsrc: this lane's 16 source bytes; lds_dst: wave-uniform LDS byte address; lane l lands at lds_dst + 16*l.
// M0 (the LDS destination base) is written in the statement that reads it and NOT saved: hipcc keeps no value of its own
// in M0 across statements in these kernels (tools/check_dma_hazard.py also lists compiler-generated uses of M0, there are none).
__device__ inline void lds_dma16(const void* gsrc, uint32_t lds_dst_in) {
  const uint32_t lds_dst = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_dst_in);   // wave-uniform by contract
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(gsrc), "s"(lds_dst) : "memory");
}
// The same with the address split into a wave-uniform 64-bit base (SGPR pair) and a per-lane 32-bit byte offset: when
// the piece's row is wave-uniform all per-piece arithmetic is scalar and ONE offset register serves every piece.
// (s_nop 4: a base that was just produced by v_readfirstlane / v_readlane needs 5 wait states before a VMEM instruction
// reads it, and hipcc pads nothing inside an asm statement.)
__device__ inline void lds_dma16_sv(const void* sbase_in, uint32_t voff, uint32_t lds_dst_in) {
  const uint32_t lds_dst = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_dst_in);
  // the base IS wave-uniform by construction; where hipcc cannot prove it, make it provable (folds away where it can)
  const uint64_t sb64 = (uint64_t)(uintptr_t)sbase_in;
  // (the builtin returns int: go through uint32_t, or an address with bit 31 set sign-extends into the high half)
  const uint32_t sb_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(sb64 >> 32));
  const uint32_t sb_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)sb64);
  const void* sbase = (const void*)(uintptr_t)(((uint64_t)sb_hi << 32) | (uint64_t)sb_lo);
  asm volatile("s_nop 4\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
// s_waitcnt vmcnt(n) for a wave-uniform runtime n (the immediate must be a literal); anything above the
// table waits for everything, which is always safe.
__device__ inline void wait_vmcnt_dyn(int n) {
#define SDA_VMC(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
  switch (n) {
    SDA_VMC(1) SDA_VMC(2) SDA_VMC(3) SDA_VMC(4) SDA_VMC(5) SDA_VMC(6) SDA_VMC(7) SDA_VMC(8) SDA_VMC(9) SDA_VMC(10)
    SDA_VMC(11) SDA_VMC(12) SDA_VMC(13) SDA_VMC(14) SDA_VMC(15) SDA_VMC(16) SDA_VMC(17) SDA_VMC(18) SDA_VMC(19)
    SDA_VMC(20) SDA_VMC(21) SDA_VMC(22) SDA_VMC(23) SDA_VMC(24)
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
#undef SDA_VMC
}

__device__ inline uint32_t lds_addr(const void* p) {
  return (uint32_t)(uintptr_t)((__attribute__((address_space(3))) const void*)p);
}

// host-side error plumbing (capi.hip)
void set_error(const char* fmt, ...);
int check_launch(const char* what);
// Per-DEVICE launch state (a process may drive several GPUs): `done` is a static bit mask owned by the call site; true the
// first time the CURRENT device asks (hipFuncSetAttribute is per device).
bool first_use_on_device(unsigned long long& done);
// CUs a persistent grid may count on: the current device's CU count, or the limit the caller set for launches that go to
// a CU-masked stream (sda_set_cu_limit; thread-local, 0 = none)
int launch_cus();

}  // namespace sda

// param_gemm — batched fp32 matrix products on PARAMETER-sized operands with arbitrary strides, and a strided 3-d copy.
//
// What it replaces: the handful of small matrix products a training step makes in parameter space — composing the
// SubjectBlock's three consecutive linear maps (SpatialAttention mix, shared 1x1 conv, per-subject 1x1 conv;
// models.py:111-117) into one per-subject matrix, and the chain rule back through that composition — which were
// torch.matmul calls (library GEMMs) with cat / mul / copy glue around them.  Operands are what the parameters are:
// 270 x 270, 270 x 209, 27 x 270 x 270 ... with odd extents and, for the transposed uses, a non-unit stride along
// the contraction.  Nothing is padded, packed or transposed first: the kernel takes element strides for every index.
//
//     C[b][i][j] = sum_k A[b][i][k] * B[b][k][j]          (fp32 in, exact-fp32 MFMA, fp32 / bf16 / fp16 out)
//
// One workgroup = 4 waves (2 x 2) owns a TM x TM tile (TM = 128 or 64) of one batch member; the contraction runs in
// steps of 16 through LDS images stored k-major ([k][i], [k][j]) so that the operand of v_mfma_f32_16x16x4_f32 — lane
// l supplies A[i = l & 15][k = l >> 4] resp. B[k = l >> 4][j = l & 15] — is one conflict-free ds_read_b32.  Global
// loads are per-element (rows / columns outside the operand clamped, the contraction's tail zero-filled), coalesced along
// whichever index has the unit stride, and the loads of step s + 1 are in registers while step s computes.  fp32 throughout:
// results are a k-ordered fmaf chain per output element, bitwise reproducible.  Bound: fp32 matrix rate (157 TFLOP/s); these products are ~6 GFLOP a
// step, i.e. tens of microseconds.
#include "sd_common.h"

namespace sda {

namespace {

constexpr int PG_K = 16;                 // contraction step

template <int TM> struct PGeom {
  static constexpr int PITCH = TM + 4;   // floats per LDS row; +4 keeps the k-major stores of the k-fastest load pattern spread
                                         // (+16, conflict-free fragment reads at 4-way store conflicts: the same 75 us)
  static constexpr int FR = TM / 32;     // 16x16 fragments per wave and dimension (wave tile TM/2 x TM/2)
  static constexpr int PER_THREAD = TM * PG_K / 256;      // elements of one operand tile per thread and step
};

template <typename OUT> __device__ inline void pg_store(OUT* p, float v);
template <> __device__ inline void pg_store<float>(float* p, float v) { *p = v; }
template <> __device__ inline void pg_store<uint16_t>(uint16_t* p, float v) { *p = f2bf(v); }
template <> __device__ inline void pg_store<half_t>(half_t* p, float v) { *p = (half_t)v; }

template <int TM, typename OUT>
__global__ __launch_bounds__(256) void param_gemm_kernel(const sda_pgemm_args a) {
  using G = PGeom<TM>;
  __shared__ float As[PG_K * G::PITCH];
  __shared__ float Bs[PG_K * G::PITCH];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wave_m = wid >> 1, wave_n = wid & 1;
  const int lr = lane & 15, lq = lane >> 4;
  const int tiles_n = (a.N + TM - 1) / TM, tiles_m = (a.M + TM - 1) / TM;
  int bid = blockIdx.x;
  const int b = bid / (tiles_m * tiles_n);
  bid -= b * tiles_m * tiles_n;
  const int i0 = (bid / tiles_n) * TM, j0 = (bid % tiles_n) * TM;
  const float* __restrict__ Ab = a.A + (size_t)b * a.a_b;
  const float* __restrict__ Bb = a.B + (size_t)b * a.b_b;

  // element (row r of the tile, contraction index kk of the step) this thread fetches in its e-th load: consecutive threads
  // run along the index whose stride is 1 (k-fastest for K-contiguous rows, row-fastest otherwise)
  const bool a_kfast = a.a_k == 1, b_kfast = a.b_k == 1;
  auto a_coord = [&](int e, int& r, int& kk) {
    const int idx = tid + e * 256;
    if (a_kfast) { kk = idx % PG_K; r = idx / PG_K; } else { r = idx % TM; kk = idx / TM; }
  };
  auto b_coord = [&](int e, int& r, int& kk) {
    const int idx = tid + e * 256;
    if (b_kfast) { kk = idx % PG_K; r = idx / PG_K; } else { r = idx % TM; kk = idx / TM; }
  };
  float ra[G::PER_THREAD], rb[G::PER_THREAD];
  // per element: its 32-bit byte offset from a wave-uniform base that one scalar add advances per step (the host checked
  // that every offset fits) — a full step then costs one load and one 64-bit add per element: with per-element
  // 64-bit pointers, a bounds compare and a pointer bump each, the fetch was 3.6 vector instructions per MFMA on the port the
  // MFMAs issue through (profiles/r04_step_sq_counters.txt).  Rows / columns outside the operand are CLAMPED, not zero-filled:
  // they feed outputs that are never stored.  Only the contraction's partial last step (k >= K must contribute zero) tests.
  uint32_t oa[G::PER_THREAD], ob[G::PER_THREAD];
  int ka[G::PER_THREAD], kb[G::PER_THREAD];                 // the element's k inside a step
#pragma unroll
  for (int e = 0; e < G::PER_THREAD; ++e) {
    int r, kk;
    a_coord(e, r, kk);
    oa[e] = (uint32_t)(((long)min(i0 + r, a.M - 1) * a.a_i + (long)kk * a.a_k) * 4);      // BYTES: global_load saddr + 32-bit voffset
    ka[e] = kk;
    b_coord(e, r, kk);
    ob[e] = (uint32_t)(((long)kk * a.b_k + (long)min(j0 + r, a.N - 1) * a.b_j) * 4);
    kb[e] = kk;
  }
  const long step_a = (long)PG_K * a.a_k, step_b = (long)PG_K * a.b_k;
  const char* __restrict__ Ak = reinterpret_cast<const char*>(Ab);      // wave-uniform: base of the step being fetched
  const char* __restrict__ Bk = reinterpret_cast<const char*>(Bb);
  auto ld = [](const char* base, uint32_t off) { return *reinterpret_cast<const float*>(base + off); };
  auto fetch = [&](int k0) {
    const int left = a.K - k0;
    if (left >= PG_K) {
#pragma unroll
      for (int e = 0; e < G::PER_THREAD; ++e) {
        ra[e] = ld(Ak, oa[e]);
        rb[e] = ld(Bk, ob[e]);
      }
    } else {
#pragma unroll
      for (int e = 0; e < G::PER_THREAD; ++e) {
        ra[e] = ka[e] < left ? ld(Ak, oa[e]) : 0.f;
        rb[e] = kb[e] < left ? ld(Bk, ob[e]) : 0.f;
      }
    }
    Ak += step_a * 4;
    Bk += step_b * 4;
  };
  auto stash = [&]() {
#pragma unroll
    for (int e = 0; e < G::PER_THREAD; ++e) {
      int r, kk;
      a_coord(e, r, kk);
      As[kk * G::PITCH + r] = ra[e];
      b_coord(e, r, kk);
      Bs[kk * G::PITCH + r] = rb[e];
    }
  };

  f32x4 acc[G::FR][G::FR];
#pragma unroll
  for (int m = 0; m < G::FR; ++m)
#pragma unroll
    for (int n = 0; n < G::FR; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

  // (a register ring of four steps' loads, statically indexed, was tried: hipcc's own s_waitcnt in front of each stash came
  // out as vmcnt(8) — only the step fetched last may stay in flight — so the effective distance stayed one step: 75 us either way)
  fetch(0);
  for (int k0 = 0; k0 < a.K; k0 += PG_K) {
    __syncthreads();                       // the previous step's LDS reads are done
    stash();
    __syncthreads();
    if (k0 + PG_K < a.K) fetch(k0 + PG_K);  // in flight while this step computes
#pragma unroll
    for (int k4 = 0; k4 < PG_K / 4; ++k4) {
      float fa[G::FR], fb[G::FR];
#pragma unroll
      for (int m = 0; m < G::FR; ++m) fa[m] = As[(k4 * 4 + lq) * G::PITCH + wave_m * (TM / 2) + m * 16 + lr];
#pragma unroll
      for (int n = 0; n < G::FR; ++n) fb[n] = Bs[(k4 * 4 + lq) * G::PITCH + wave_n * (TM / 2) + n * 16 + lr];
#pragma unroll
      for (int m = 0; m < G::FR; ++m)
#pragma unroll
        for (int n = 0; n < G::FR; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[m], fb[n], acc[m][n], 0, 0, 0);
    }
  }

  OUT* __restrict__ Cb = reinterpret_cast<OUT*>(a.C) + (size_t)b * a.c_b;
#pragma unroll
  for (int m = 0; m < G::FR; ++m)
#pragma unroll
    for (int n = 0; n < G::FR; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = i0 + wave_m * (TM / 2) + m * 16 + lq * 4 + r;      // C/D map: row = 4 * (lane >> 4) + reg, col = lane & 15
        const int j = j0 + wave_n * (TM / 2) + n * 16 + lr;
        if (i < a.M && j < a.N) pg_store<OUT>(Cb + (size_t)i * a.c_i + (size_t)j * a.c_j, acc[m][n][r]);
      }
}

__global__ __launch_bounds__(256) void copy3d_kernel(float* __restrict__ dst, long d0, long d1, long d2,
                                                     const float* __restrict__ src, long s0, long s1, long s2, int n0, int n1,
                                                     int n2) {
  const long total = (long)n0 * n1 * n2;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int k = (int)(idx % n2);
    const long t = idx / n2;
    const int j = (int)(t % n1), i = (int)(t / n1);
    dst[i * d0 + j * d1 + k * d2] = src[i * s0 + j * s1 + k * s2];
  }
}

template <int TM> int launch_pgemm(const sda_pgemm_args& a, hipStream_t st) {
  const long grid = (long)((a.M + TM - 1) / TM) * ((a.N + TM - 1) / TM) * a.batch;
  if (a.c_dtype == SDA_F32) hipLaunchKernelGGL((param_gemm_kernel<TM, float>), dim3((unsigned)grid), dim3(256), 0, st, a);
  else if (a.c_dtype == SDA_BF16) hipLaunchKernelGGL((param_gemm_kernel<TM, uint16_t>), dim3((unsigned)grid), dim3(256), 0, st, a);
  else hipLaunchKernelGGL((param_gemm_kernel<TM, half_t>), dim3((unsigned)grid), dim3(256), 0, st, a);
  return check_launch("param_gemm");
}

}  // namespace
}  // namespace sda

using namespace sda;

extern "C" int sda_param_gemm(const sda_pgemm_args* a, void* stream) {
  if (!a || !a->A || !a->B || !a->C) { set_error("param_gemm: null argument"); return -1; }
  if (a->M < 1 || a->N < 1 || a->K < 1 || a->batch < 1) { set_error("param_gemm: empty problem"); return -1; }
  if (a->c_dtype != SDA_F32 && a->c_dtype != SDA_BF16 && a->c_dtype != SDA_F16) { set_error("param_gemm: unknown output dtype %d", a->c_dtype); return -1; }
  // the kernel addresses an operand by 32-bit element offsets from a per-batch-member base (parameter-sized by contract)
  auto span = [](long n0, long s0, long n1, long s1) { return (n0 - 1) * (s0 < 0 ? -s0 : s0) + (n1 - 1) * (s1 < 0 ? -s1 : s1); };
  if (a->a_i < 0 || a->a_k < 0 || a->b_k < 0 || a->b_j < 0) { set_error("param_gemm: negative strides are not supported"); return -1; }
  if (span(a->M, a->a_i, a->K, a->a_k) > 0x3fffffffL || span(a->K, a->b_k, a->N, a->b_j) > 0x3fffffffL) {
    set_error("param_gemm: an operand spans more than 2^30 elements per batch member (parameter-sized operands only)");
    return -1;
  }
  const long t128 = (long)((a->M + 127) / 128) * ((a->N + 127) / 128) * a->batch;
  if ((long)((a->M + 63) / 64) * ((a->N + 63) / 64) * a->batch > 0x7fffffffL) { set_error("param_gemm: grid too large"); return -1; }
  // 128-wide tiles once they fill the chip; 64-wide ones otherwise (four times the workgroups for the same work)
  return t128 >= 256 ? launch_pgemm<128>(*a, (hipStream_t)stream) : launch_pgemm<64>(*a, (hipStream_t)stream);
}

extern "C" int sda_copy3d(float* dst, long d0, long d1, long d2, const float* src, long s0, long s1, long s2, int n0, int n1,
                          int n2, void* stream) {
  if (!dst || !src || n0 < 1 || n1 < 1 || n2 < 1) { set_error("copy3d: bad arguments"); return -1; }
  const long total = (long)n0 * n1 * n2;
  const long blocks = (total + 255) / 256;
  hipLaunchKernelGGL(copy3d_kernel, dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256), 0, (hipStream_t)stream, dst, d0, d1, d2,
                     src, s0, s1, s2, n0, n1, n2);
  return check_launch("copy3d");
}

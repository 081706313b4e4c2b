// Probe: how many VALU issue slots per element does a 1:1 bf16 streaming pass (118 MB) hide on MI355X?
// Same thread->chunk mapping as the elementwise passes; per element N extra ops of one kind.
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_budget valu_budget.hip ; run: ./valu_budget
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <cmath>
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int ROWS = 92160, CP = 320, NCH = CP / 8, RG = 256 / NCH;

__device__ inline f32x2 fma2(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ inline f32x2 splat2(float v) { return f32x2{v, v}; }
__device__ inline f32x2 gelu2(f32x2 x) {        // the product's packed-pair exact GELU (sd_common.h)
  const f32x2 y = x * 0.849321800288f;
  const f32x2 yy = y * y;
  const f32x2 e = f32x2{__builtin_amdgcn_exp2f(-yy.x), __builtin_amdgcn_exp2f(-yy.y)};
  const f32x2 t = {__builtin_amdgcn_rcpf(__builtin_fmaf(__builtin_fabsf(x.x), 0.2316419f, 1.0f)),
                   __builtin_amdgcn_rcpf(__builtin_fmaf(__builtin_fabsf(x.y), 0.2316419f, 1.0f))};
  f32x2 q = fma2(t, splat2(0.5f * 1.061405429f), splat2(0.5f * -1.453152027f));
  q = fma2(q, t, splat2(0.5f * 1.421413741f));
  q = fma2(q, t, splat2(0.5f * -0.284496736f));
  q = fma2(q, t, splat2(0.5f * 0.254829592f));
  const f32x2 tail = (q * t) * e;
  return f32x2{__builtin_fmaf(-__builtin_fabsf(x.x), tail.x, __builtin_fmaxf(x.x, 0.f)),
               __builtin_fmaf(-__builtin_fabsf(x.y), tail.y, __builtin_fmaxf(x.y, 0.f))};
}
constexpr int TAB_N = 2048;                       // intervals over |x| in [0, 8): h = 1/256; entries (f, f' * h) pairs
__device__ inline float gelu_tab(float x, const float2* tab) {
  const float u = __builtin_amdgcn_fmed3f(__builtin_fabsf(x) * 256.0f, 0.0f, 2047.996f);
  const int i = (int)u;
  const float fr = u - (float)i;
  const float2 e = tab[i];
  const float tail = __builtin_fmaf(fr, e.y, e.x);
  return __builtin_fmaf(-__builtin_fabsf(x), tail, __builtin_fmaxf(x, 0.f));
}

template <int KIND, int N>
__global__ __launch_bounds__(256) void pass(const uint16_t* __restrict__ x, uint16_t* __restrict__ y, float c0, float c1,
                                            const float2* __restrict__ gtab = nullptr) {
  __shared__ float2 tab[KIND == 4 ? TAB_N : 1];
  if (KIND == 4) {
    for (int i = threadIdx.x; i < TAB_N; i += 256) tab[i] = gtab[i];
    __syncthreads();
  }
  const int per = (ROWS + gridDim.x - 1) / gridDim.x;
  const int r0 = blockIdx.x * per, r1 = min(ROWS, r0 + per);
  const int ch = threadIdx.x % NCH, rg = threadIdx.x / NCH;
  if (rg >= RG) return;
  for (int r = r0 + rg; r < r1; r += RG * 2) {
    uint4 u[2];
    bool ok[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) { ok[k] = r + k * RG < r1; if (ok[k]) u[k] = *reinterpret_cast<const uint4*>(x + (size_t)(r + k * RG) * CP + ch * 8); }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      if (!ok[k]) continue;
      uint32_t w[4] = {u[k].x, u[k].y, u[k].z, u[k].w};
      float v[8];
#pragma unroll
      for (int i = 0; i < 4; ++i) { v[2 * i] = __uint_as_float(w[i] << 16); v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
      if (KIND == 0) {                       // scalar v_fma_f32 chain
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
          for (int n = 0; n < N; ++n) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j]) : "v"(c0), "v"(c1));
      } else if (KIND == 1) {                // packed v_pk_fma_f32 (N per PAIR)
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
          f32x2 p = {v[j], v[j + 1]};
          const f32x2 a = {c0, c0}, b = {c1, c1};
#pragma unroll
          for (int n = 0; n < N; ++n) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p) : "v"(a), "v"(b));
          v[j] = p.x; v[j + 1] = p.y;
        }
      } else if (KIND == 3) {                // affine + packed exact GELU
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
          const f32x2 o = gelu2(fma2(f32x2{v[j], v[j + 1]}, splat2(c0), splat2(c1)));
          v[j] = o.x; v[j + 1] = o.y;
        }
      } else if (KIND == 4) {                // affine + LDS-table GELU (linear interpolation of the normal tail)
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = gelu_tab(__builtin_fmaf(v[j], c0, c1), tab);
      } else if (KIND == 2) {                // v_exp_f32
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
          for (int n = 0; n < N; ++n) asm volatile("v_exp_f32 %0, %0" : "+v"(v[j]));
      }
      uint32_t o[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(o[i]) : "v"(v[2 * i]), "v"(v[2 * i + 1]));
      }
      *reinterpret_cast<uint4*>(y + (size_t)(r + k * RG) * CP + ch * 8) = make_uint4(o[0], o[1], o[2], o[3]);
    }
  }
}

static float2* g_tab = nullptr;
static int g_nb = 1920;
template <int KIND, int N> float run(const uint16_t* x, uint16_t* y) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int nb = g_nb;
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((pass<KIND, N>), dim3(nb), dim3(256), 0, 0, x, y, 1.0001f, 0.001f, g_tab);
  hipEventRecord(e0);
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((pass<KIND, N>), dim3(nb), dim3(256), 0, 0, x, y, 1.0001f, 0.001f, g_tab);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / 20 * 1e3f;
}
#define ROW(K, N) printf("kind %d  N=%2d  %6.1f us\n", K, N, run<K, N>(x, y));
int main() {
  uint16_t *x, *y;
  hipMalloc(&x, (size_t)ROWS * CP * 2); hipMalloc(&y, (size_t)ROWS * CP * 2);
  hipMemset(x, 0x3f, (size_t)ROWS * CP * 2);
  printf("kind 0 = v_fma_f32 per element, 1 = v_pk_fma_f32 per PAIR, 2 = v_exp_f32 per element\n");
  ROW(0, 0) ROW(0, 4) ROW(0, 8) ROW(0, 12) ROW(0, 16) ROW(0, 24) ROW(0, 32)
  ROW(1, 4) ROW(1, 8) ROW(1, 16) ROW(1, 24) ROW(1, 32) ROW(1, 48)
  ROW(2, 1) ROW(2, 2) ROW(2, 4) ROW(2, 8) ROW(2, 12)
  {
    std::vector<float2> h(TAB_N);
    for (int i = 0; i < TAB_N; ++i) {
      const double a = i / 256.0, b = (i + 1) / 256.0;
      const double fa = 0.5 * erfc(a / sqrt(2.0)), fb = 0.5 * erfc(b / sqrt(2.0));
      h[i] = make_float2((float)fa, (float)(fb - fa));
    }
    hipMalloc(&g_tab, TAB_N * sizeof(float2));
    hipMemcpy(g_tab, h.data(), TAB_N * sizeof(float2), hipMemcpyHostToDevice);
  }
  printf("kind 3 = affine + packed exact GELU, 4 = affine + LDS table GELU (grid 1920 / 1024 / 512 blocks)\n");
  for (int nb : {1920, 1024, 512}) { g_nb = nb; printf("blocks %d: ", nb); ROW(3, 0) printf("blocks %d: ", nb); ROW(4, 0) }
  return 0;
}

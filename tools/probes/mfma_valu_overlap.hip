// Probe: do MFMAs of one wave and ordinary vector instructions of ANOTHER wave on the same SIMD overlap?
// 512-thread workgroups, one per CU: waves 0-3 (one per SIMD) run a chain of 16x16x32 bf16 MFMAs over 8 independent accumulators,
// waves 4-7 run packed-fp32 FMA chains (or v_exp/v_rcp).  Times: MFMA alone, VALU alone, both.
// Build: hipcc -O3 --offload-arch=gfx950 tools/probes/mfma_valu_overlap.hip -o gpurun_out/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND>   // VALU flavour: 0 packed fma, 1 transcendental (exp2 + rcp), 2 plain v_fma_f32
__global__ __launch_bounds__(512) void probe(float* out, int n_mfma, int n_valu, int do_mfma, int do_valu) {
  const int wid = threadIdx.x >> 6;
  if (wid < 4) {
    if (!do_mfma) return;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(1.0f + i * 0.01f); }
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < n_mfma; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 123.456f) out[threadIdx.x] = s;
  } else {
    if (!do_valu) return;
    f32x2 v[4];
    for (int i = 0; i < 4; ++i) v[i] = f32x2{threadIdx.x * 0.001f + i, 0.5f + i};
    const f32x2 c1 = {0.999f, 1.001f}, c2 = {0.001f, -0.001f};
    for (int it = 0; it < n_valu; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (KIND == 0) v[i] = __builtin_elementwise_fma(v[i], c1, c2);
        else if (KIND == 1) v[i] = f32x2{__builtin_amdgcn_exp2f(v[i].x) , __builtin_amdgcn_rcpf(v[i].y)};
        else v[i] = f32x2{__builtin_fmaf(v[i].x, 0.999f, 0.001f), __builtin_fmaf(v[i].y, 1.001f, -0.001f)};
      }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) s += v[i].x + v[i].y;
    if (s == 123.456f) out[threadIdx.x] = s;
  }
}

template <int KIND> void run(const char* name, float* out, int n_mfma, int n_valu) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  float ms[3];
  for (int mode = 0; mode < 3; ++mode) {
    const int dm = mode != 1, dv = mode != 0;
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(probe<KIND>, dim3(256), dim3(512), 0, 0, out, n_mfma, n_valu, dm, dv);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      hipEventElapsedTime(&ms[mode], e0, e1);
    }
  }
  printf("%-22s MFMA alone %8.1f us  VALU alone %8.1f us  both %8.1f us  (sum %8.1f, max %8.1f)\n", name, ms[0] * 1e3, ms[1] * 1e3,
         ms[2] * 1e3, (ms[0] + ms[1]) * 1e3, (ms[0] > ms[1] ? ms[0] : ms[1]) * 1e3);
}

int main() {
  float* out;
  hipMalloc(&out, 4096);
  const int n_mfma = 20000;                     // x 8 MFMAs per wave
  printf("per wave: %d MFMAs (16x16x32 bf16)\n", n_mfma * 8);
  run<0>("packed fma (v_pk_fma)", out, n_mfma, 160000);
  run<2>("plain v_fma_f32 x2", out, n_mfma, 80000);
  run<1>("v_exp_f32 + v_rcp_f32", out, n_mfma, 40000);
  run<0>("packed fma, half load", out, n_mfma, 80000);
  return 0;
}

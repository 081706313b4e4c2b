// Probe: the shader clock while the matrix pipe is loaded, and what a bare MFMA loop reaches.
// A one-wave kernel reads the shader-clock counter (s_memtime, clock64()) and the constant-rate counter (s_memrealtime,
// wall_clock64(), 100 MHz) before and after a fixed stretch of wall time; the ratio is the engine clock it ran at.  It runs
// alone, then beside a grid that issues dense v_mfma_f32_16x16x32_bf16 from every SIMD (8 independent accumulator quads per
// wave, no memory traffic), with 1, 2 and 4 waves per SIMD; the grid's own rate is taken from HIP events.
// Build: hipcc -O3 --offload-arch=gfx950 tools/probes/sclk_probe.hip -o gpurun_out/sclk_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4;

__global__ void clock_probe(unsigned long long* out, unsigned long long wall_ticks) {
  if (threadIdx.x != 0) return;
  const unsigned long long w0 = wall_clock64(), c0 = clock64();
  unsigned long long w1 = w0;
  while (w1 - w0 < wall_ticks) { __builtin_amdgcn_s_sleep(16); w1 = wall_clock64(); }
  const unsigned long long c1 = clock64();
  out[0] = w1 - w0; out[1] = c1 - c0;
}

__global__ __launch_bounds__(256) void mfma_load(float* sink, int iters) {
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)(float)(blockIdx.x + i); }
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 12345.678f) sink[0] = s;
}

typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;
// the same FLOPs per wave on v_mfma_f32_32x32x16_bf16 (32 768 FLOP per instruction, 4 independent accumulator blocks)
__global__ __launch_bounds__(256) void mfma_load32(float* sink, int iters) {
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)(float)(blockIdx.x + i); }
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
  if (s == 12345.678f) sink[0] = s;
}
// fp32: v_mfma_f32_16x16x4_f32 (2 048 FLOP) and v_mfma_f32_32x32x2_f32 (4 096 FLOP)
__global__ __launch_bounds__(256) void mfma_load_f32(float* sink, int iters, int big) {
  float a = (float)threadIdx.x, b = (float)blockIdx.x;
  f32x4 acc[8];
  f32x16 acc2[4];
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc2[i][j] = 0.f;
  if (!big) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
  } else {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc2[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc2[i], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) s += acc2[i][j];
  if (s == 12345.678f) sink[0] = s;
}

template <typename F>
static void timed(const char* what, int wps, double flop, hipStream_t s1, F launch) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  launch(8);
  hipStreamSynchronize(s1);
  hipEventRecord(e0, s1);
  launch(1);
  hipEventRecord(e1, s1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  printf("%s, %d wave(s) per SIMD: %.2f ms, %.0f TFLOP/s\n", what, wps, ms, flop / (ms * 1e-3) / 1e12);
}

int main() {
  int wall_khz = 0;
  hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0);
  int sclk_khz = 0;
  hipDeviceGetAttribute(&sclk_khz, hipDeviceAttributeClockRate, 0);
  printf("wall clock rate %d kHz, nominal engine clock %d kHz\n", wall_khz, sclk_khz);
  unsigned long long* out; hipMalloc(&out, 16);
  float* sink; hipMalloc(&sink, 4);
  hipStream_t s1, s2;
  hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
  hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
  const unsigned long long ticks = (unsigned long long)wall_khz * 20;          // 20 ms
  unsigned long long h[2];
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(clock_probe, dim3(1), dim3(64), 0, s2, out, ticks);
    hipStreamSynchronize(s2);
    hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
    printf("alone: %.1f ms of wall time, counter ratio %.4f -> %.0f MHz\n", h[0] / (double)wall_khz, (double)h[1] / h[0],
           (double)h[1] / h[0] * wall_khz / 1e3);
  }
  for (int wps : {1, 2, 4}) {
    const int wgs = 256 * wps;             // one 4-wave workgroup per CU and wave-per-SIMD
    const int iters = 120000 / wps;        // ~ equal MFMA count per SIMD in every variant
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(mfma_load, dim3(wgs), dim3(256), 0, s1, sink, iters / 8);        // warm-up
    hipStreamSynchronize(s1);
    hipEventRecord(e0, s1);
    hipLaunchKernelGGL(mfma_load, dim3(wgs), dim3(256), 0, s1, sink, iters);
    hipEventRecord(e1, s1);
    // the probe starts a little later, inside the load
    hipLaunchKernelGGL(clock_probe, dim3(1), dim3(64), 0, s2, out, ticks / 4);
    hipStreamSynchronize(s2);
    hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)wgs * 4 * iters * 32 * 16384.0;
    printf("%d wave(s) per SIMD: %.2f ms, %.0f TFLOP/s bf16; probe beside it: ratio %.4f -> %.0f MHz (%.1f ms)\n", wps, ms,
           flop / (ms * 1e-3) / 1e12, (double)h[1] / h[0], (double)h[1] / h[0] * wall_khz / 1e3, h[0] / (double)wall_khz);
  }
  for (int wps : {1, 2, 4}) {
    const int wgs = 256 * wps, iters = 60000 / wps;
    timed("bf16 32x32x16", wps, (double)wgs * 4 * iters * 16 * 32768.0, s1,
          [&](int div) { hipLaunchKernelGGL(mfma_load32, dim3(wgs), dim3(256), 0, s1, sink, iters / div); });
  }
  for (int big : {0, 1})
    for (int wps : {1, 2, 4}) {
      const int wgs = 256 * wps, iters = 30000 / wps;
      timed(big ? "fp32 32x32x2" : "fp32 16x16x4", wps, (double)wgs * 4 * iters * (big ? 16 * 4096.0 : 32 * 2048.0), s1,
            [&](int div) { hipLaunchKernelGGL(mfma_load_f32, dim3(wgs), dim3(256), 0, s1, sink, iters / div, big); });
    }
  return 0;
}

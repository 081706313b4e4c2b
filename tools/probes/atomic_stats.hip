// Probe: cost of accumulating per-workgroup BatchNorm partial sums with 64-bit integer atomics (order-independent, hence
// bitwise reproducible) instead of per-tile rows + a finalize launch.  G workgroups x (2 x C) atomicAdd(unsigned long long)
// onto 2*C addresses.  Build: hipcc --offload-arch=gfx950 -O3 -o atomic_stats atomic_stats.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void k(unsigned long long* acc, int C, int spin) {
  // some fake work first so that workgroups do not all arrive in the same cycle (as conv tiles would not)
  float v = threadIdx.x;
  for (int i = 0; i < spin * (1 + (blockIdx.x & 7)); ++i) v = v * 1.0001f + 0.5f;
  if (v == 12345.f) acc[0] = 1;
  for (int c = threadIdx.x; c < 2 * C; c += 256) atomicAdd(acc + c, (unsigned long long)(blockIdx.x + c));
}
int main() {
  unsigned long long* acc; hipMalloc(&acc, 2 * 1024 * 8); hipMemset(acc, 0, 2 * 1024 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int C : {160, 320, 640})
    for (int G : {512, 1536, 3072})
      for (int spin : {0, 200}) {
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(G), dim3(256), 0, 0, acc, C, spin);
        hipEventRecord(e0);
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k, dim3(G), dim3(256), 0, 0, acc, C, spin);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("C=%d workgroups=%d spin=%d: %.1f us per launch (%d atomics per address)\n", C, G, spin, ms / 20 * 1e3, G);
      }
  return 0;
}

// Probe: does the L2 -> LDS rate of LDS-DMA depend on how many contiguous bytes a piece takes per source row?
// One wave-instruction (global_load_lds_dwordx4) moves 1 KiB = 64 lanes x 16 B.  The conv / GEMM kernels of this repo stage
// 64-byte K-step slabs: a piece = 16 rows x 64 B (half a 128-byte cache line per row).  Compared here, all else equal: pieces of
// 16 x 64 B, 8 x 128 B, 4 x 256 B, 2 x 512 B and 1 x 1024 B, from a table every workgroup shares (L2-resident) and from one far
// larger than the caches, with 1 or 2 workgroups per CU and DEPTH pieces in flight per wave.
// Build: hipcc -O3 --offload-arch=gfx950 tools/probes/ldsdma_segment.hip -o gpurun_out/ldsdma_segment
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int SEG, int DEPTH>   // SEG = contiguous bytes per row and piece
__global__ __launch_bounds__(256) void probe(const unsigned char* __restrict__ table, long rows, long pitch, int iters, float* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  constexpr int LPR = SEG / 16;            // lanes per row
  constexpr int RPP = 64 / LPR;            // rows per piece
  const uint32_t voff = (uint32_t)((lane / LPR) * pitch + (lane % LPR) * 16);
  const uint32_t lds_base = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) void*)smem) + wid * DEPTH * 1024;
  // wave-uniform walk over the table: each piece starts RPP rows and (for short segments) one segment further
  long row = ((long)blockIdx.x * 4 + wid) * 977 % (rows - RPP);
  long col = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const unsigned char* base = table + row * pitch + col;
      const uint32_t dst = lds_base + d * 1024;
      asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(base), "s"(dst) : "memory");
      // next piece: the next segment of the same rows (as a K loop does), wrapping to the next row block
      col += SEG;
      if (col + SEG > pitch) { col = 0; row += RPP * 13; if (row >= rows - RPP) row -= (rows - RPP); }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  if (smem[threadIdx.x] == 123 && sink) sink[0] = 1.f;
}

template <int SEG, int DEPTH>
static double run(const unsigned char* table, long rows, long pitch, int wgs, int iters, float* sink) {
  hipFuncSetAttribute(reinterpret_cast<const void*>(probe<SEG, DEPTH>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * DEPTH * 1024);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((probe<SEG, DEPTH>), dim3(wgs), dim3(256), 4 * DEPTH * 1024, 0, table, rows, pitch, iters / 4, sink);
  hipEventRecord(e0);
  hipLaunchKernelGGL((probe<SEG, DEPTH>), dim3(wgs), dim3(256), 4 * DEPTH * 1024, 0, table, rows, pitch, iters, sink);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)wgs * 4 * iters * DEPTH * 1024;
  return bytes / (ms * 1e-3) / 1e12;
}

int main() {
  const long pitch = 1024;                                  // bytes per table row (a multiple of every SEG)
  float* sink; hipMalloc(&sink, 4);
  struct { const char* name; long rows; } tabs[] = {{"L2-resident (2 MB shared)", 2048}, {"streamed (2 GB)", 2L << 20}};
  for (auto& t : tabs) {
    unsigned char* table;
    if (hipMalloc(&table, t.rows * pitch) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(table, 1, t.rows * pitch);
    for (int per_cu = 1; per_cu <= 2; ++per_cu) {
      const int wgs = 256 * per_cu;
      const int iters = t.rows > 100000 ? 400 : 2000;
      printf("%s, %d workgroup(s) per CU, 8 pieces in flight per wave: TB/s by bytes per row and piece\n", t.name, per_cu);
      printf("   64 B: %.2f   128 B: %.2f   256 B: %.2f   512 B: %.2f   1024 B: %.2f\n",
             run<64, 8>(table, t.rows, pitch, wgs, iters, sink), run<128, 8>(table, t.rows, pitch, wgs, iters, sink),
             run<256, 8>(table, t.rows, pitch, wgs, iters, sink), run<512, 8>(table, t.rows, pitch, wgs, iters, sink),
             run<1024, 8>(table, t.rows, pitch, wgs, iters, sink));
      printf("   16 in flight:  64 B: %.2f   128 B: %.2f   512 B: %.2f\n",
             run<64, 16>(table, t.rows, pitch, wgs, iters, sink), run<128, 16>(table, t.rows, pitch, wgs, iters, sink),
             run<512, 16>(table, t.rows, pitch, wgs, iters, sink));
    }
    hipFree(table);
  }
  return 0;
}

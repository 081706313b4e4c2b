// Helpers shared by the flat-tile conv kernels (conv3_flat.hip, conv1_flat.hip).
#pragma once
#include "conv_tile.h"

#include <type_traits>
#include <utility>

namespace sda {

// compile-time loop: f(std::integral_constant<int, I>) for I in [B, E)
template <int B, int E, typename F> __device__ __forceinline__ void static_for(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_for<B + 1, E>(f);
  }
}

template <int N> __device__ __forceinline__ void wait_vmcnt_lit() {
  static_assert(N >= 0 && N <= 8, "add the literal");
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
  if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
  if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
  if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  if constexpr (N == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
  if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
}

// One LDS-DMA piece, lean form: wave-uniform 64-bit source base (an SGPR pair the caller computed with scalar arithmetic
// well ahead — no VALU-written SGPR feeds the load, so no s_nop 4), ONE per-lane byte offset register, and M0 written in
// the statement that reads it (nothing else in these kernels keeps a value in M0, so it is not saved).
// PADDED = true opens with s_nop 4: for the places (a tile's prologue) where hipcc may hand the statement a scalar it has
// just reloaded from a spill lane (v_readlane: a VALU write).  tools/check_dma_hazard.py walks the listing for unpadded ones.
template <bool PADDED = false>
__device__ __forceinline__ void lds_dma16_lean(const void* sbase, uint32_t voff, uint32_t lds_dst) {
  if constexpr (PADDED)
    asm volatile("s_nop 4\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
  else
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

using TrueC = std::integral_constant<bool, true>;
using FalseC = std::integral_constant<bool, false>;

}  // namespace sda

// Transposed MFMA operands out of row-major LDS images (ds_read_b64_tr_b16 / ds_read_b32), shared by the kernels that
// contract over the ROW index of their operands: wgrad_gemm.hip (weight gradients) and loss_gemm.hip (the loss's dZ product).
#pragma once
#include "sd_common.h"

namespace sda {

// LDS images are plain row-major copies (row = time step, RB bytes per row, no padding) filled by 1 KB
// LDS-DMA pieces; bank conflicts of the TRANSPOSED operand reads are removed by XOR-ing the 16-byte chunk
// index with a function of the row (applied to the DMA source address and to the read address alike):
//   bf16 (ds_read_b64_tr_b16, a 32-lane half reads 8 consecutive rows x 32 B):  32-byte groups
//       RB = 128: group ^= (row >> 1) & 3     RB = 256, 512: group ^= row & 7     RB = 320: group ^= (row >> 2) & 1
//   fp32 (ds_read_b32, a half reads 2 consecutive rows x 64 B):  64-byte groups, group ^= row & 1
template <typename E, int RB> __device__ inline int chunk_xor(int row) {
  if (sizeof(E) == 4) return (row & 1) << 2;
  if (RB == 128) return ((row >> 1) & 3) << 1;
  if (RB == 256) return (row & 7) << 1;
  if (RB == 320) return ((row >> 2) & 1) << 1;
  if (RB == 512) return (row & 7) << 1;                         // (row stride = 0 mod 256 B: the eight rows of a half take eight distinct 32-byte groups)
  return 0;
}

// transposed MFMA operand: 16 columns starting at col0; rows row0.. (32 for bf16, 16 for fp32).
// Lane group g supplies k = {4g..4g+3, 16+4g..16+4g+3} (bf16) / {g, 4+g, 8+g, 12+g} (fp32) for BOTH operands.
template <typename E, int RB> __device__ inline uint4 tr_operand(const unsigned char* img, int row0, int col0, int lane);

template <int RB> __device__ inline uint4 tr_operand_bf16(const unsigned char* img, int row0, int col0, int lane) {
  const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
  const int row = row0 + 4 * g + q;
  const int colb = (col0 + 4 * p) * 2;                         // byte column, multiple of 8
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  const unsigned char* a0 = img + row * RB + ((((colb >> 4) ^ chunk_xor<uint16_t, RB>(row))) << 4) + (colb & 15);
  const unsigned char* a1 = img + (row + 16) * RB + ((((colb >> 4) ^ chunk_xor<uint16_t, RB>(row + 16))) << 4) + (colb & 15);
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a1));
  uint4 r;
  r.x = (uint32_t)(uint16_t)lo[0] | ((uint32_t)(uint16_t)lo[1] << 16);
  r.y = (uint32_t)(uint16_t)lo[2] | ((uint32_t)(uint16_t)lo[3] << 16);
  r.z = (uint32_t)(uint16_t)hi[0] | ((uint32_t)(uint16_t)hi[1] << 16);
  r.w = (uint32_t)(uint16_t)hi[2] | ((uint32_t)(uint16_t)hi[3] << 16);
  return r;
}
template <int RB> __device__ inline uint4 tr_operand_f32(const unsigned char* img, int row0, int col0, int lane) {
  const int g = lane >> 4, i = lane & 15;
  const int colb = (col0 + i) * 4;
  uint4 r;
  uint32_t* rp = reinterpret_cast<uint32_t*>(&r);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int row = row0 + g + 4 * j;
    rp[j] = *reinterpret_cast<const uint32_t*>(img + row * RB + (((colb >> 4) ^ chunk_xor<float, RB>(row)) << 4) + (colb & 15));
  }
  return r;
}
// The same read split into a per-lane byte offset that is computed ONCE (tr_offset: everything that depends on the lane,
// on the operand's column and on the low bits of its first row — the swizzle only looks at row bits 0..2) and the read
// itself at `img + off` plus a compile-time row offset (a multiple of 16 rows: the swizzle does not see it): inside a
// K loop a fragment read is then ONE address add instead of ~10 vector instructions per fragment — the MFMAs share the
// vector issue port with them (wgrad_gemm's chunk loop had two VALU instructions per MFMA).
template <int RB> __device__ inline uint32_t tr_offset_bf16(int row0, int col0, int lane) {
  const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
  const int row = row0 + 4 * g + q;
  const int colb = (col0 + 4 * p) * 2;
  return (uint32_t)(row * RB + ((((colb >> 4) ^ chunk_xor<uint16_t, RB>(row))) << 4) + (colb & 15));
}
template <int RB> __device__ inline uint4 tr_read_bf16(const unsigned char* img_plus_off) {
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img_plus_off));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img_plus_off + 16 * RB));
  uint4 r;
  r.x = (uint32_t)(uint16_t)lo[0] | ((uint32_t)(uint16_t)lo[1] << 16);
  r.y = (uint32_t)(uint16_t)lo[2] | ((uint32_t)(uint16_t)lo[3] << 16);
  r.z = (uint32_t)(uint16_t)hi[0] | ((uint32_t)(uint16_t)hi[1] << 16);
  r.w = (uint32_t)(uint16_t)hi[2] | ((uint32_t)(uint16_t)hi[3] << 16);
  return r;
}
template <typename E, int RB> struct TrOp;
template <int RB> struct TrOp<uint16_t, RB> {
  __device__ static uint4 get(const unsigned char* img, int row0, int col0, int lane) { return tr_operand_bf16<RB>(img, row0, col0, lane); }
};
template <int RB> struct TrOp<half_t, RB> {     // any 16-bit element: ds_read_b64_tr_b16 moves bits
  __device__ static uint4 get(const unsigned char* img, int row0, int col0, int lane) { return tr_operand_bf16<RB>(img, row0, col0, lane); }
};
template <int RB> struct TrOp<float, RB> {
  __device__ static uint4 get(const unsigned char* img, int row0, int col0, int lane) { return tr_operand_f32<RB>(img, row0, col0, lane); }
};

}  // namespace sda

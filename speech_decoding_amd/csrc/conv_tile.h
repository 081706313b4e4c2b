// LDS tile geometry shared by the conv kernels (conv_gemm.hip, conv3_flat.hip).
#pragma once
#include "sd_common.h"

namespace sda {

// One LDS row = one MFMA K-step = 64 bytes (32 bf16 / 16 fp32 input channels).
constexpr int ROW_B = 64;
constexpr int XROWS = TILE_T + 2 * PAD;            // 160: worst-case halo
constexpr int XS_BYTES = XROWS * ROW_B;            // 10 KB

// 64-byte rows, 4 chunks of 16 bytes, chunk' = chunk ^ sw64(row) with sw64 = 2 * bit 2 of the row.
// ds_read_b128 is serviced in four 16-lane groups that are NOT contiguous ({0-3,12-15,20-27}, {4-11,16-19,
// 28-31}, ...): with lane = 16 * chunk + row the MFMA operand read puts rows {0-3,12-15} of one chunk and rows
// {4-11} of the next chunk in one group, and this XOR lands them on 16 distinct 16-byte slots of the 256-byte
// bank row for EVERY starting row (dilated taps start anywhere); measured SQ_LDS_BANK_CONFLICT = 0.
__device__ inline int sw64(int row) { return (row >> 1) & 2; }
__device__ inline int lds_sw64(int row, int chunk) { return row * ROW_B + ((chunk ^ sw64(row)) << 4); }

// conv3_flat.hip
int launch_conv3_flat(const sda_conv_args& a, hipStream_t st);
bool conv3_flat_supports(const sda_conv_args& a);
int conv3_flat_stat_rows(int B, int T);

// conv1_flat.hip
int launch_conv1_flat(const sda_conv_args& a, hipStream_t st);
bool conv1_flat_supports(const sda_conv_args& a);

// conv1_wide.hip
int launch_conv1_wide(const sda_conv_args& a, hipStream_t st);
bool conv1_wide_supports(const sda_conv_args& a);
int conv1_wide_stat_rows(int B, int T);

}  // namespace sda

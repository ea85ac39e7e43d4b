// ds_read_b64_tr_b16: the hardware-transposed LDS read of gfx950.  Per 16-lane group it delivers a 4 row x 16 column bf16
// block column-major: lane = column, registers = 4 consecutive rows.  Two of them (rows 4g .. 4g + 3 and 16 + 4g .. of a
// 32-row tile, g = lane >> 4) are one operand of v_mfma_f32_16x16x32_bf16 whose reduction index is the ROW of a tile kept in
// LDS in its natural [row][column] layout - no transposed copy, no 2-byte stores.  Requirements: EXEC all ones, 8-byte
// aligned addresses, and a row pitch of 8 x odd dwords (16 x odd elements: 48, 80, 144, 272) for conflict-free reads.
#pragma once
#include "common.h"

namespace mmft {

typedef short tr_s16x4 __attribute__((ext_vector_type(4)));
typedef short tr_s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 tr_bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) tr_s16x4 tr_lds_s16x4;

// the element a lane points at inside the 32-row tile: row tr_lane_row, column (16 x block) + tr_lane_col
__device__ __forceinline__ int tr_lane_row(int lane) { return 4 * (lane >> 4) + ((lane >> 2) & 3); }
__device__ __forceinline__ int tr_lane_col(int lane) { return 4 * (lane & 3); }

// p = tile + tr_lane_row * pitch + 16 * block + tr_lane_col; rows_16 = 16 * pitch (the second half of the 32 rows)
__device__ __forceinline__ tr_bf16x8 tr_read_pair(const unsigned short* p, int rows_16) {
  const tr_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr_lds_s16x4*)p);
  const tr_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr_lds_s16x4*)(p + rows_16));
  const tr_s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(tr_bf16x8, v);
}

}  // namespace mmft

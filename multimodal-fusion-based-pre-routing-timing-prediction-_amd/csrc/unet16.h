// bf16-STORAGE kernels of the layout U-Net (bf16 math mode; src/Unet.py:8-119): shared device helpers.
//
// In this family every activation, pre-activation and activation gradient lives in HBM as bf16 (NHWC); statistics,
// parameters, parameter gradients and Adam state stay fp32; every contraction accumulates in fp32 on the MFMA.
// Fragment conventions are those of conv_tile.h / conv_wgrad_narrow.h (v_mfma_f32_16x16x16_bf16):
//   A fragment  lane (r, q) = A[row r][k = 4q .. 4q+3]          B fragment  lane (r, q) = B[k = 4q .. 4q+3][col r]
//   D           lane (r, q) = D[row 4q .. 4q+3][col r]
// "natural" fragment of an NHWC tensor = lane (pixel r, 4 consecutive channels 4q..) = one 8-byte load.
#pragma once
#include "gemm_engine.h"

namespace mmft {

typedef unsigned short u16;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));   // 8 bf16 = one 16-byte access
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));   // 4 bf16

__device__ __forceinline__ float bf2f(unsigned h) { return __uint_as_float(h << 16); }
__device__ __forceinline__ unsigned f2bf(float v) {
  f32x2_t t = {v, 0.f};
  bf16x2_t b = __builtin_convertvector(t, bf16x2_t);
  return __builtin_bit_cast(unsigned, b) & 0xffffu;
}
// 8 bf16 <-> 8 floats
__device__ __forceinline__ void unpack8(u32x4 v, float f[8]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f[2 * i] = __uint_as_float(v[i] << 16);
    f[2 * i + 1] = __uint_as_float(v[i] & 0xffff0000u);
  }
}
__device__ __forceinline__ u32x4 pack8(const float f[8]) {
  u32x4 v;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f32x2_t t = {f[2 * i], f[2 * i + 1]};
    bf16x2_t b = __builtin_convertvector(t, bf16x2_t);
    v[i] = __builtin_bit_cast(unsigned, b);
  }
  return v;
}
__device__ __forceinline__ void unpack4(u32x2 v, float f[4]) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    f[2 * i] = __uint_as_float(v[i] << 16);
    f[2 * i + 1] = __uint_as_float(v[i] & 0xffff0000u);
  }
}
__device__ __forceinline__ s16x4 as_s16x4(u32x2 v) { return __builtin_bit_cast(s16x4, v); }
__device__ __forceinline__ u32x2 as_u32x2(s16x4 v) { return __builtin_bit_cast(u32x2, v); }

// the value a stored bf16 activation takes: relu(fma(z, scale, shift)) rounded to bf16.  The backward kernels recompute
// the ReLU mask from z with the SAME expression (t > 0), so forward and backward agree on every element.
__device__ __forceinline__ float bn_pre(float z, float scale, float shift) { return __fmaf_rn(z, scale, shift); }

}  // namespace mmft

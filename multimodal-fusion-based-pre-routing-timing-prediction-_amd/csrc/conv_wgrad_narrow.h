// Weight gradient of the narrow 3x3 / pad 1 convolutions (Ci, Co in {16, 32}: the 256x256 and 128x128 stages of
// src/Unet.py:8-25), bf16 math mode.
//
//   dw[co][ky][kx][ci] = sum over pixels p of dy[p][co] * x[p + (ky - 1, kx - 1)][ci]
//
// As an implicit GEMM this is M = Co (16 or 32), N = 9 Ci, K = every pixel of the batch: the generic engine gives it a
// 16-row tile and pulls the im2col operand - nine shifted views of x - through the vector memory path once per tap
// (300 MB for a 33 MB tensor; 108 us per layer at 256x256, a quarter of what the whole forward U-Net costs).
//
// The contraction runs over PIXELS, so both MFMA operands want "lane = channel, registers = 4 consecutive pixels", while
// NHWC memory gives "lane = pixel, registers = 4 consecutive channels".  The transposition is done by the matrix unit
// itself: for a 16 pixel x 16 channel block in the natural layout, D = A * I (one v_mfma_f32_16x16x16_bf16 against an
// identity operand) returns exactly the block with lanes and registers swapped - exact, the products are 1.0 * bf16.
// So per workgroup tile of TH x 64 pixels:
//   * x (with its one-pixel halo) is read from HBM once, rounded to bf16 and kept in LDS in its NATURAL layout (one
//     ds_write_b64 per 16-byte load; pixel pitch CI + 8 elements, which spreads the 16 pixels of a fragment read over
//     all banks); a tap's fragment is an aligned ds_read_b64 at any pixel shift;
//   * dy never touches LDS: a wave loads the 1 KB of its 16 pixels straight into the natural fragment;
//   * per 16 pixels a wave runs Co/16 + 9 Ci/16 transposing MFMAs and (Co/16) * 9 * (Ci/16) accumulating ones; the
//     accumulators stay in registers across all tiles of the (persistent) workgroup;
//   * the loads of tile t + 1 are in flight while tile t is multiplied;
//   * the four waves' partial sums are added in LDS in wave order and the workgroup writes ONE slab; slab_reduce adds
//     the slabs in a fixed order (bitwise reproducible run to run).
#pragma once
#include "gemm_engine.h"

namespace mmft {

struct ConvWgradArgs {
  const float* x;    // [N][H][W][CI]
  const float* dy;   // [N][H][W][CO]
  float* slabs;      // [gridDim.x][CO][9][CI]
  int N, H, W;
  int tiles;         // N * (H / TH) * (W / TW)
};

template <int CI, int CO, int TH, int TW>
struct ConvWgradCfg {
  static constexpr int XR = TH + 2, XC = TW + 2;
  static constexpr int PIX = CI + 8;               // LDS elements per pixel: 12 / 20 dwords - 16 pixels x 4 dwords hit 64 banks once
  static constexpr int XS_BYTES = XR * XC * PIX * 2;
  static constexpr int RED_BYTES = CO * 9 * CI * 4;
  static constexpr int LDS_BYTES = XS_BYTES > RED_BYTES ? XS_BYTES : RED_BYTES;
};

// CS: channels of x in memory (CS == CI, or the 3-channel image of the first layer with CI == 16: scalar loads, the
// other LDS channels stay zero, and only dw[co][tap][0..CS) is written to the slab).
template <int CI, int CO, int TH, int TW, int CS = CI>
__global__ void __launch_bounds__(256) conv3x3_wgrad_narrow_kernel(ConvWgradArgs a) {
  using C = ConvWgradCfg<CI, CO, TH, TW>;
  constexpr int XR = C::XR, XC = C::XC, PIX = C::PIX, CB = CI / 16, MB = CO / 16;
  constexpr int STEPS = TH * TW / 16, SPR = TW / 16, SPW = STEPS / 4;
  static_assert(TW % 16 == 0 && CI % 16 == 0 && CO % 16 == 0 && STEPS % 4 == 0, "tile / channel granularity");
  extern __shared__ __attribute__((aligned(16))) unsigned short xs[];     // [XR][XC][PIX]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int tiles_x = a.W / TW, tiles_y = a.H / TH;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  // identity operand: lane (n = r, q) holds I[4q + j][r]
  s16x4 ident;
#pragma unroll
  for (int j = 0; j < 4; ++j) ident[j] = (4 * q + j == r) ? (short)0x3F80 : (short)0;
  auto transpose = [&](s16x4 natural) { return pack_bf16x4(mfma_bf16_k16(natural, ident, zero)); };

  f32x4 acc[MB][9][CB];
#pragma unroll
  for (int m = 0; m < MB; ++m)
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int c = 0; c < CB; ++c) acc[m][t][c] = zero;

  // Staging: x item = (halo row, halo column, channel group of 4), one 16-byte load each; dy fragment = (step of this
  // wave, 16-channel block), loaded in the natural layout (lane = pixel r, channels 4q..4q+3 of the block).
  constexpr int XG = CS == CI ? CI / 4 : 1, XI = XR * XC * XG, NX = (XI + 255) / 256;
  if constexpr (CS != CI) {
    for (int e = tid; e < XR * XC * PIX / 4; e += 256) reinterpret_cast<s16x4*>(xs)[e] = s16x4{0, 0, 0, 0};
    __syncthreads();
  }
  f32x4 xr[NX], dr[SPW][MB];
  s16x4 af[SPW][MB];
  auto request = [&](int tile) {
    const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, img = tile / (tiles_x * tiles_y);
    const int x0 = tx * TW, y0 = ty * TH;
    const long long img0 = (long long)img * a.H;
#pragma unroll
    for (int k = 0; k < NX; ++k) {
      const int it = tid + k * 256;
      const int cg = it % XG, col = (it / XG) % XC, row = it / (XG * XC);
      const int yy = y0 - 1 + row, xx = x0 - 1 + col;
      const bool ok = it < XI && yy >= 0 && yy < a.H && xx >= 0 && xx < a.W;
      // out-of-image items read the tile's first pixel (always valid) and are zeroed by the select
      const float* src = a.x + ((img0 + (ok ? yy : y0)) * a.W + (ok ? xx : x0)) * CS + cg * 4;
      f32x4 v = zero;
      if constexpr (CS == CI) {
        v = *reinterpret_cast<const f32x4*>(src);
      } else {
#pragma unroll
        for (int j = 0; j < CS; ++j) v[j] = src[j];
      }
      xr[k] = ok ? v : zero;
    }
#pragma unroll
    for (int s = 0; s < SPW; ++s) {
      const int st = wave + 4 * s, row = st / SPR, c0 = (st % SPR) * 16;
#pragma unroll
      for (int m = 0; m < MB; ++m)
        dr[s][m] = *reinterpret_cast<const f32x4*>(a.dy + ((img0 + y0 + row) * a.W + x0 + c0 + r) * CO + m * 16 + 4 * q);
    }
  };
  auto deposit = [&]() {
#pragma unroll
    for (int k = 0; k < NX; ++k) {
      const int it = tid + k * 256;
      if (it < XI) {
        const int cg = it % XG, pix = it / XG;                  // pix = row * XC + col
        *reinterpret_cast<s16x4*>(xs + pix * PIX + cg * 4) = pack_bf16x4(xr[k]);
      }
    }
#pragma unroll
    for (int s = 0; s < SPW; ++s)
#pragma unroll
      for (int m = 0; m < MB; ++m) af[s][m] = transpose(pack_bf16x4(dr[s][m]));
  };

  if ((int)blockIdx.x < a.tiles) request(blockIdx.x);
  for (int tile = blockIdx.x; tile < a.tiles; tile += gridDim.x) {
    deposit();
    __syncthreads();
    if (tile + (int)gridDim.x < a.tiles) request(tile + gridDim.x);
#pragma unroll
    for (int s = 0; s < SPW; ++s) {
      const int st = wave + 4 * s, row = st / SPR, c0 = (st % SPR) * 16;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
          for (int c = 0; c < CB; ++c) {
            // natural fragment of the tap: pixel (row + ky, c0 + r + kx) of the halo tile, channels 16 c + 4 q ..
            const s16x4 nat = *reinterpret_cast<const s16x4*>(xs + ((row + ky) * XC + c0 + r + kx) * PIX + c * 16 + 4 * q);
            const s16x4 bfr = transpose(nat);
#pragma unroll
            for (int m = 0; m < MB; ++m) acc[m][ky * 3 + kx][c] = mfma_bf16_k16(af[s][m], bfr, acc[m][ky * 3 + kx][c]);
          }
    }
    __syncthreads();
  }

  // ---- the four waves' sums, added in wave order; acc[m][t][c][j] at lane (r, q) = dw[co = 16 m + 4 q + j][t][ci = 16 c + r]
  float* red = reinterpret_cast<float*>(xs);
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
          for (int c = 0; c < CB; ++c)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              float* p = red + ((m * 16 + 4 * q + j) * 9 + t) * CI + c * 16 + r;
              *p = w == 0 ? acc[m][t][c][j] : *p + acc[m][t][c][j];
            }
    }
    __syncthreads();
  }
  float* out = a.slabs + (long long)blockIdx.x * (CO * 9 * CS);
  if constexpr (CS == CI) {
    for (int e = tid * 4; e < CO * 9 * CI; e += 1024)
      *reinterpret_cast<f32x4*>(out + e) = *reinterpret_cast<const f32x4*>(red + e);
  } else {
    for (int e = tid; e < CO * 9 * CS; e += 256) out[e] = red[(e / CS) * CI + e % CS];
  }
}

template <int TH>
inline bool conv_wgrad_narrow_shape_ok(int H, int W, int Ci, int Co, int KH, int KW, int pad) {
  return KH == 3 && KW == 3 && pad == 1 && (((Ci == 16 || Ci == 32) && (Co == 16 || Co == 32)) || (Ci == 3 && Co == 16)) &&
         W % 64 == 0 && H % TH == 0;
}

// 4 x 64 pixel tiles: 19 KB of LDS at Ci = 16, 32 KB at Ci = 32
constexpr int WGN_TH = 4;
constexpr int wgn_tw(int Ci) { return 64; }

inline bool conv_wgrad_narrow_ok(int H, int W, int Ci, int Co, int KH, int KW, int pad) {
  static int off = -1;
  if (off < 0) {
    const char* e = getenv("MMFT_CONV_WGRAD_NARROW");
    off = (e && atoi(e) == 0) ? 1 : 0;               // MMFT_CONV_WGRAD_NARROW=0: implicit-GEMM path (comparison runs)
  }
  return !off && math_mode() == MMFT_MATH_BF16 && conv_wgrad_narrow_shape_ok<WGN_TH>(H, W, Ci, Co, KH, KW, pad);
}

inline int conv_wgrad_narrow_grid(int Nimg, int H, int W, int Ci) {
  const long long tiles = (long long)Nimg * (H / WGN_TH) * (W / wgn_tw(Ci));
  static int cap = 0, tpw = 0;
  if (!cap) {
    const char* e = getenv("MMFT_CONV_WGRAD_GRID");
    cap = e && atoi(e) > 0 ? atoi(e) : 512;
    e = getenv("MMFT_CONV_WGRAD_TPW");
    tpw = e && atoi(e) > 0 ? atoi(e) : 2;
  }
  long long g = (tiles + tpw - 1) / tpw;             // tiles per workgroup: the slab is written once per workgroup
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

template <int CI, int CO, int CS = CI>
inline int conv_wgrad_narrow_launch_t(const ConvWgradArgs& a, int grid, hipStream_t st) {
  constexpr int TW = wgn_tw(CI);
  using C = ConvWgradCfg<CI, CO, WGN_TH, TW>;
  const double flops = 2.0 * a.N * a.H * a.W * CO * 9.0 * CS;
  const double bytes = 4.0 * a.N * a.H * a.W * (CS + CO);
  static DynLdsOnce once;
  int rc = ensure_dyn_lds(once, reinterpret_cast<const void*>(&conv3x3_wgrad_narrow_kernel<CI, CO, WGN_TH, TW, CS>), C::LDS_BYTES,
                          "conv3x3_wgrad_narrow");
  if (rc) return rc;
  MMFT_LAUNCH_LDS("conv3x3_wgrad_narrow_kernel", flops, bytes, (conv3x3_wgrad_narrow_kernel<CI, CO, WGN_TH, TW, CS>), dim3(grid),
                  dim3(256), C::LDS_BYTES, st, a);
  return MMFT_OK;
}

// slabs: grid * Co * 9 * Ci floats.  Returns the launch status; the caller runs slab_reduce over `grid` slabs.
inline int conv_wgrad_narrow_launch(const float* x, const float* dy, float* slabs, int Nimg, int H, int W, int Ci, int Co,
                                    int grid, hipStream_t st) {
  ConvWgradArgs a{x, dy, slabs, Nimg, H, W, Nimg * (H / WGN_TH) * (W / wgn_tw(Ci))};
  int rc;
  if (Ci == 3 && Co == 16) rc = conv_wgrad_narrow_launch_t<16, 16, 3>(a, grid, st);
  else if (Ci == 16 && Co == 16) rc = conv_wgrad_narrow_launch_t<16, 16>(a, grid, st);
  else if (Ci == 16 && Co == 32) rc = conv_wgrad_narrow_launch_t<16, 32>(a, grid, st);
  else if (Ci == 32 && Co == 16) rc = conv_wgrad_narrow_launch_t<32, 16>(a, grid, st);
  else rc = conv_wgrad_narrow_launch_t<32, 32>(a, grid, st);
  return rc ? rc : check_launch("conv3x3_wgrad_narrow");
}

}  // namespace mmft

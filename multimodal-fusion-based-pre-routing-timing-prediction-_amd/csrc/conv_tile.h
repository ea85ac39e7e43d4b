// Tile-resident 3x3 / pad 1 convolution for the narrow high-resolution layers of the U-Net (Ci, Co in {16, 32}; the
// 256x256 and 128x128 stages of src/Unet.py:8-25; forward and - with flipped weights - input gradient), bf16 math mode.
//
// conv3x3_direct_kernel (conv_direct.h) feeds the MFMA straight from global memory: every output pixel requests its nine
// taps itself, so the vector memory path carries nine times the input (L1 / L2 absorb most of it, the address and tag
// work stays) and rocprofv3 counts 1.75-2x the algorithmic HBM bytes.  Here a workgroup owns a TH x 64 pixel tile:
//   * the input tile with its one-pixel halo is read ONCE (16-byte loads), rounded to bf16 and kept in LDS in the NHWC
//     layout it has in memory (pixel pitch CI + 8 elements: the 16 pixels of a fragment read cover all 64 banks once);
//   * a tap's B operand - 16 consecutive pixels of a row x 16 channels, lane = pixel, 4 channels per lane - is one
//     aligned ds_read_b64 at any (ky, kx) shift; the weights live in registers as A fragments for the whole kernel;
//   * per 16 pixels a wave issues 9 * CI / 16 LDS reads and 9 * (CI / 16) * (CO / 16) v_mfma_f32_16x16x16_bf16;
//   * the loads of tile t + 1 are in flight while tile t is multiplied (one exposed memory latency per workgroup).
//   out[p][co] = act(bias[co] + sum_{ky,kx,ci} w[co][ky][kx][ci] * in[p + (ky-1, kx-1)][ci])
#pragma once
#include "gemm_engine.h"
#include "conv_direct.h"

namespace mmft {

// CS: channels of the input tensor in memory.  CS == CI, or CS < 4 with CI == 16 (the RGB input of the first layer,
// src/Unet.py:93: its pixels are 12 bytes, loaded as scalars; the other 13 channels of the LDS tile stay zero).
template <int CI, int CO, int TH, int TW, int CS = CI>
__global__ void __launch_bounds__(256) conv3x3_tile_kernel(ConvDirectArgs a, int tiles) {
  constexpr int XR = TH + 2, XC = TW + 2, PIX = CI + 8, CB = CI / 16, MB = CO / 16;
  constexpr int STEPS = TH * TW / 16, SPR = TW / 16, SPW = STEPS / 4;
  static_assert(TW % 16 == 0 && CI % 16 == 0 && CO % 16 == 0 && STEPS % 4 == 0, "tile / channel granularity");
  __shared__ __attribute__((aligned(16))) unsigned short xs[XR * XC * PIX];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int tiles_x = a.W / TW, tiles_y = a.H / TH;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};

  // A fragments: lane (co = 16 m + r, q) holds w[co][tap][16 c + 4 q .. + 3]
  s16x4 wf[9][CB][MB];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int c = 0; c < CB; ++c)
#pragma unroll
      for (int m = 0; m < MB; ++m) {
        const int co = m * 16 + r, ci = c * 16 + 4 * q;
        f32x4 v;
        if constexpr (CS != CI) {
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = ci + j < CS ? a.w[((long long)co * 9 + t) * CS + ci + j] : 0.f;
        } else if (!a.flip) {
          v = *reinterpret_cast<const f32x4*>(a.w + ((long long)co * 9 + t) * CI + ci);
        } else {
          // dx = conv(dy, w') with w'[co][tap][ci] = w[ci][8 - tap][co]  (w: the layer's forward weight [CI][3][3][CO])
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = a.w[((long long)(ci + j) * 9 + (8 - t)) * CO + co];
        }
        wf[t][c][m] = pack_bf16x4(v);
      }
  f32x4 bias[MB];
#pragma unroll
  for (int m = 0; m < MB; ++m) bias[m] = a.bias ? *reinterpret_cast<const f32x4*>(a.bias + m * 16 + 4 * q) : zero;

  constexpr int XG = CS == CI ? CI / 4 : 1, XI = XR * XC * XG, NX = (XI + 255) / 256;
  if constexpr (CS != CI) {
    for (int e = tid; e < XR * XC * PIX / 4; e += 256) reinterpret_cast<s16x4*>(xs)[e] = s16x4{0, 0, 0, 0};
    __syncthreads();
  }
  f32x4 xr[NX];
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
  };
  auto deposit = [&]() {
#pragma unroll
    for (int k = 0; k < NX; ++k) {
      const int it = tid + k * 256;
      if (it < XI) *reinterpret_cast<s16x4*>(xs + (it / XG) * PIX + (it % XG) * 4) = pack_bf16x4(xr[k]);
    }
  };

  if ((int)blockIdx.x < tiles) request(blockIdx.x);
  for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    deposit();
    __syncthreads();
    if (tile + (int)gridDim.x < tiles) request(tile + gridDim.x);
    const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, img = tile / (tiles_x * tiles_y);
    const int x0 = tx * TW, y0 = ty * TH;
#pragma unroll
    for (int s = 0; s < SPW; ++s) {
      const int st = wave + 4 * s, row = st / SPR, c0 = (st % SPR) * 16;
      f32x4 acc[MB];
#pragma unroll
      for (int m = 0; m < MB; ++m) acc[m] = zero;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
          for (int c = 0; c < CB; ++c) {
            const s16x4 xf = *reinterpret_cast<const s16x4*>(xs + ((row + ky) * XC + c0 + r + kx) * PIX + c * 16 + 4 * q);
#pragma unroll
            for (int m = 0; m < MB; ++m) acc[m] = mfma_bf16_k16(wf[ky * 3 + kx][c][m], xf, acc[m]);
          }
      // acc[m] at lane (pixel r, q) = output channels 16 m + 4 q .. + 3 of pixel (y0 + row, x0 + c0 + r)
      float* out = a.y + (((long long)img * a.H + y0 + row) * a.W + x0 + c0 + r) * CO;
#pragma unroll
      for (int m = 0; m < MB; ++m) {
        f32x4 v = acc[m] + bias[m];
        if (a.act == ACT_RELU) {
#pragma unroll
          for (int t = 0; t < 4; ++t) v[t] = v[t] > 0.f ? v[t] : 0.f;
        } else if (a.act == ACT_LEAKY) {
#pragma unroll
          for (int t = 0; t < 4; ++t) v[t] = v[t] > 0.f ? v[t] : v[t] * a.slope;
        }
        *reinterpret_cast<f32x4*>(out + m * 16 + 4 * q) = v;
      }
    }
    __syncthreads();
  }
}

constexpr int CVT_TH = 4, CVT_TW = 64;

// the RGB first layer (forward only: nothing needs the gradient of the images)
inline bool conv_tile_rgb_shape(int Ci, int Co, int KH, int KW, int pad) { return Ci == 3 && Co == 16 && KH == 3 && KW == 3 && pad == 1; }

inline bool conv_tile_ok(int H, int W) {
  static int off = -1;
  if (off < 0) {
    const char* e = getenv("MMFT_CONV_TILE");
    off = (e && atoi(e) == 0) ? 1 : 0;               // MMFT_CONV_TILE=0: conv3x3_direct_kernel (comparison runs)
  }
  return !off && math_mode() == MMFT_MATH_BF16 && H % CVT_TH == 0 && W % CVT_TW == 0;
}

template <int CI, int CO, int CS = CI>
inline void conv_tile_launch_t(const ConvDirectArgs& a, hipStream_t st) {
  const int tiles = a.N * (a.H / CVT_TH) * (a.W / CVT_TW);
  static int cap = 0;
  if (!cap) {
    const char* e = getenv("MMFT_CONV_TILE_GRID");
    cap = e && atoi(e) > 0 ? atoi(e) : 1024;
  }
  const int grid = tiles < cap ? tiles : cap;
  const double flops = 2.0 * a.N * a.H * a.W * CO * 9.0 * CS;
  const double bytes = 4.0 * a.N * a.H * a.W * (CS + CO) + 4.0 * CO * 9 * CS;
  MMFT_LAUNCH("conv3x3_tile_kernel", flops, bytes, (conv3x3_tile_kernel<CI, CO, CVT_TH, CVT_TW, CS>), dim3(grid), dim3(256), st,
              a, tiles);
}

inline int conv_tile_launch(const float* x, const float* w, const float* bias, float* y, int Nimg, int H, int W, int Ci,
                            int Co, int act, float slope, hipStream_t st, int flip = 0) {
  ConvDirectArgs a{x, w, bias, y, Nimg, H, W, act, slope, flip};
  if (Ci == 3 && Co == 16) conv_tile_launch_t<16, 16, 3>(a, st);
  else if (Ci == 16 && Co == 16) conv_tile_launch_t<16, 16>(a, st);
  else if (Ci == 16 && Co == 32) conv_tile_launch_t<16, 32>(a, st);
  else if (Ci == 32 && Co == 16) conv_tile_launch_t<32, 16>(a, st);
  else conv_tile_launch_t<32, 32>(a, st);
  return check_launch("conv3x3_tile");
}

}  // namespace mmft

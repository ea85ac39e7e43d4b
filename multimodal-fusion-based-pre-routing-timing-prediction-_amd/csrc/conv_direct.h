// Direct 3x3 / pad 1 convolution for the narrow high-resolution layers of the U-Net (Ci, Co in {16, 32}; the 256x256
// and 128x128 stages of src/Unet.py:8-25, forward and - with flipped weights - input gradient).
//
// The implicit-GEMM path stages an im2col tile through LDS with a barrier per 16-deep K step; with 16 / 32 output
// channels a wave then has 8-16 MFMAs between barriers and the layer runs at a third of the matrix rate.  Here the
// NHWC layout is used directly: the B operand of v_mfma_f32_16x16x4_f32 for (16 consecutive pixels of an image row, one
// tap, 16 input channels) is ONE contiguous 1 KB wave load (lane = pixel, lane quarter q = channels 4q..4q+3), the
// weights sit in LDS for the life of the (persistent) workgroup, every wave owns 64 pixels of a row and runs without
// any barrier; the next (tap row, channel chunk) stage is requested in program order before the current one is
// multiplied, but pinning that order (sched_barrier) costs 100 more VGPRs and measured slower (46 vs 40 us):
// five waves per SIMD hide the loads better than a deeper pipeline in two.
//   out[p][co] = act(bias[co] + sum_{ky,kx,ci} w[co][ky][kx][ci] * in[p + (ky-1, kx-1)][ci])
#pragma once
#include "gemm_engine.h"

namespace mmft {

struct ConvDirectArgs {
  const float* x;      // [N][H][W][CI]
  const float* w;      // [CO][3][3][CI]
  const float* bias;   // [CO] or null
  float* y;            // [N][H][W][CO]
  int N, H, W;
  int act;
  float slope;
  int flip;            // 1: input-gradient use - w is the FORWARD weight [CI][3][3][CO] of the layer, read flipped
};

template <int CI, int CO, bool BF>   // BF: operands rounded to bf16 at the MFMA (MMFT_MATH_BF16)
__global__ void __launch_bounds__(256) conv3x3_direct_kernel(ConvDirectArgs a) {
  constexpr int K = 9 * CI, WS = K + 4;          // (K + 4) / 4 is odd: conflict-free b128 fragment reads
  constexpr int NS = 4, CC = CI / 16, CS = CO / 16, STAGES = 3 * CC;
  extern __shared__ __attribute__((aligned(16))) float wl0[];      // [CO][WS]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (!a.flip) {
    for (int e = tid; e < CO * K / 4; e += 256) {
      int co = e / (K / 4), k4 = e % (K / 4);
      *reinterpret_cast<f32x4*>(wl0 + co * WS + k4 * 4) = *reinterpret_cast<const f32x4*>(a.w + (long long)co * K + k4 * 4);
    }
  } else {
    // dx = conv(dy, w') with w'[co][tap][ci] = w[ci][8 - tap][co]: the 180-degree flip and the channel transpose are
    // done while staging (<= 37 KB once per persistent workgroup) instead of by a re-layout kernel per step
    for (int e = tid; e < CO * K; e += 256) {
      int co = e / K, k = e % K;
      int tap = k / CI, ci = k % CI;
      wl0[co * WS + k] = a.w[((long long)ci * 9 + (8 - tap)) * CO + co];
    }
  }
  __syncthreads();
  const int r = lane & 15, q = lane >> 4;
  const int xblocks = a.W / 64;
  const int items = a.N * a.H * xblocks;
  const int stride = gridDim.x * 4;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};

  for (int item = blockIdx.x * 4 + wave; item < items; item += stride) {
    const int xb = item % xblocks;
    const int row = item / xblocks;                    // img * H + y
    const int yrow = row % a.H;
    const long long img_row0 = (long long)(row - yrow);   // img * H
    const int x0 = xb * 64;

    f32x4 xf[2][3][NS];
    auto load_stage = [&](auto stc, auto bufc) {
      constexpr int st = decltype(stc)::value, buf = decltype(bufc)::value;
      constexpr int ky = st / CC, cc = st % CC;
      const int yy = yrow + ky - 1;
      const bool rowok = yy >= 0 && yy < a.H;            // wave-uniform
      const float* base = a.x + ((img_row0 + (rowok ? yy : yrow)) * a.W) * CI + cc * 16 + 4 * q;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int sub = 0; sub < NS; ++sub) {
          int px = x0 + sub * 16 + r + kx - 1;
          bool ok = rowok && px >= 0 && px < a.W;
          f32x4 v = *reinterpret_cast<const f32x4*>(base + (long long)(ok ? px : x0) * CI);
          xf[buf][kx][sub] = ok ? v : zero;
        }
    };
    f32x4 acc[NS][CS];
#pragma unroll
    for (int sub = 0; sub < NS; ++sub)
#pragma unroll
      for (int cs = 0; cs < CS; ++cs) acc[sub][cs] = zero;
    int woff = 0;
    asm volatile("" : "+s"(woff));                     // opaque per item: keeps the weight reads inside the loop
    const float* wl = wl0 + woff;
    auto compute_stage = [&](auto stc, auto bufc) {
      constexpr int st = decltype(stc)::value, buf = decltype(bufc)::value;
      constexpr int ky = st / CC, cc = st % CC;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int cs = 0; cs < CS; ++cs) {
          f32x4 wf = *reinterpret_cast<const f32x4*>(wl + (cs * 16 + r) * WS + ((ky * 3 + kx) * CI + cc * 16) + 4 * q);
          if constexpr (BF) {
            const s16x4 wp = pack_bf16x4(wf);
#pragma unroll
            for (int sub = 0; sub < NS; ++sub)
              acc[sub][cs] = mfma_bf16_k16(wp, pack_bf16x4(xf[buf][kx][sub]), acc[sub][cs]);
          } else {
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
              for (int sub = 0; sub < NS; ++sub)
                acc[sub][cs] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[s], xf[buf][kx][sub][s], acc[sub][cs], 0, 0, 0);
          }
        }
    };
    auto run = [&](auto self, auto stc) -> void {
      constexpr int st = decltype(stc)::value;
      if constexpr (st < STAGES) {
        if constexpr (st + 1 < STAGES)
          load_stage(std::integral_constant<int, st + 1>{}, std::integral_constant<int, (st + 1) & 1>{});
        compute_stage(std::integral_constant<int, st>{}, std::integral_constant<int, st & 1>{});
        self(self, std::integral_constant<int, st + 1>{});
      }
    };
    load_stage(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
    run(run, std::integral_constant<int, 0>{});

    // lane holds output channels 4q..4q+3 (+16 cs) of pixel x0 + 16 sub + r: one 16-byte store each
    float* out = a.y + ((long long)row * a.W + x0) * CO;
#pragma unroll
    for (int sub = 0; sub < NS; ++sub)
#pragma unroll
      for (int cs = 0; cs < CS; ++cs) {
        f32x4 v = acc[sub][cs];
        if (a.bias) v += *reinterpret_cast<const f32x4*>(a.bias + cs * 16 + 4 * q);
        if (a.act == ACT_RELU) {
#pragma unroll
          for (int t = 0; t < 4; ++t) v[t] = v[t] > 0.f ? v[t] : 0.f;
        } else if (a.act == ACT_LEAKY) {
#pragma unroll
          for (int t = 0; t < 4; ++t) v[t] = v[t] > 0.f ? v[t] : v[t] * a.slope;
        }
        *reinterpret_cast<f32x4*>(out + (long long)(sub * 16 + r) * CO + cs * 16 + 4 * q) = v;
      }
  }
}

inline bool conv_direct_ok(const float* x, const float* w, const float* bias, const float* y, int W, int Ci, int Co, int KH,
                           int KW, int pad) {
  static int off = -1;
  if (off < 0) {
    const char* e = getenv("MMFT_CONV_DIRECT");
    off = (e && atoi(e) == 0) ? 1 : 0;               // MMFT_CONV_DIRECT=0: implicit-GEMM path (comparison runs)
  }
  if (off) return false;
  return KH == 3 && KW == 3 && pad == 1 && (Ci == 16 || Ci == 32) && (Co == 16 || Co == 32) && W % 64 == 0 &&
         aligned16(x) && aligned16(w) && aligned16(y) && (!bias || aligned16(bias));
}

template <int CI, int CO>
inline void conv_direct_launch_t(const ConvDirectArgs& a, hipStream_t st) {
  const size_t lds = (size_t)CO * (9 * CI + 4) * 4;
  const long long items = (long long)a.N * a.H * (a.W / 64);
  long long wgs = (items + 3) / 4;
  // persistent workgroups: the weights are staged once per workgroup, so cap the grid at a few waves of the chip
  const long long cap = 256 * 4;
  if (wgs > cap) wgs = cap;
  const double flops = 2.0 * a.N * a.H * a.W * CO * 9.0 * CI;
  const double bytes = 4.0 * a.N * a.H * a.W * (CI + CO) + 4.0 * CO * 9 * CI;
  if (math_mode() == MMFT_MATH_BF16)
    MMFT_LAUNCH_LDS("conv3x3_direct_kernel<bf16>", flops, bytes, (conv3x3_direct_kernel<CI, CO, true>), dim3((unsigned)wgs),
                    dim3(256), lds, st, a);
  else
    MMFT_LAUNCH_LDS("conv3x3_direct_kernel", flops, bytes, (conv3x3_direct_kernel<CI, CO, false>), dim3((unsigned)wgs),
                    dim3(256), lds, st, a);
}

inline int conv_direct_launch(const float* x, const float* w, const float* bias, float* y, int Nimg, int H, int W, int Ci,
                              int Co, int act, float slope, hipStream_t st, int flip = 0) {
  ConvDirectArgs a{x, w, bias, y, Nimg, H, W, act, slope, flip};
  if (Ci == 16 && Co == 16) conv_direct_launch_t<16, 16>(a, st);
  else if (Ci == 16 && Co == 32) conv_direct_launch_t<16, 32>(a, st);
  else if (Ci == 32 && Co == 16) conv_direct_launch_t<32, 16>(a, st);
  else conv_direct_launch_t<32, 32>(a, st);
  return check_launch("conv3x3_direct");
}

}  // namespace mmft

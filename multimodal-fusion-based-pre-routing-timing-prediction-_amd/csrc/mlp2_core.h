// Shared pieces of the fused Linear-ReLU-Linear row kernels (mlp2.hip): register-resident
// weight panels and the two MFMA phases over a 32-row tile.  128 -> 256 -> 128 widths.
#pragma once
#include "gemm_engine.h"

namespace mmft {

constexpr int M2_BM = 32, M2_K1 = 128, M2_HD = 256, M2_D2 = 128, M2_BK = 32;

struct Mlp2Args {
  const float* x1;
  long long ldx1;
  const int* rows;
  int n;
  const float* w1;
  long long ldw1;
  const float* b1;
  const float* w2;
  long long ldw2;
  const float* b2;
  const float* mask;
  long long ldmask;
  float* hid_out;
  long long ldhid;
  float* out;
  long long ldout;
  int add_act;   // 1: out = act(out_old + acc + b2), 0: out = acc + b2
  int relu_out;
  const unsigned char* active;   // optional per-node flags (fan-in cone of the step): rows outside it are not computed
};

// Weight panels are read from L2 ONCE per workgroup, at kernel entry, with every 16-byte load of both layers in
// flight together (64 per thread = 256 VGPRs; the kernel runs one workgroup per CU, so the 512-entry register file
// has room).  The K loops then only move registers -> LDS -> MFMA fragments: no global round trip sits on the
// serial path of a level (the two-launch form paid one L2 round trip per 16-deep K step).
template <bool KM, int N, int KT, int NT = 256>   // tile kt of a weight with N output features; KT = number of 32-deep tiles; NT threads
struct WPanel {
  static constexpr int PER = N * M2_BK / 4 / NT;
  f32x4 r[KT][PER];
  __device__ __forceinline__ void load(const float* w, long long ldw, int tid) {
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
      for (int i = 0; i < PER; ++i) {
        int g = tid + i * NT;
        if (KM) {
          int kk = g / (N / 4), n4 = g % (N / 4);
          r[kt][i] = *reinterpret_cast<const f32x4*>(w + (long long)(kt * M2_BK + kk) * ldw + n4 * 4);
        } else {
          int rr = g / (M2_BK / 4), k4 = g % (M2_BK / 4);
          r[kt][i] = *reinterpret_cast<const f32x4*>(w + (long long)rr * ldw + kt * M2_BK + k4 * 4);
        }
      }
  }
  template <int KTI>
  __device__ __forceinline__ void to_lds(float* dst, int tid) const {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      int g = tid + i * NT;
      if (KM) {
        int kk = g / (N / 4), n4 = g % (N / 4);
        *reinterpret_cast<f32x4*>(dst + kk * (N + 4) + n4 * 4) = r[KTI][i];
      } else {
        int rr = g / (M2_BK / 4), k4 = g % (M2_BK / 4);
        *reinterpret_cast<f32x4*>(dst + rr * (M2_BK + 8) + k4 * 4) = r[KTI][i];
      }
    }
  }
};

// SINGLE = one weight tile buffer in LDS (two barriers per K step) instead of two (one barrier): the persistent
// sweep kernel keeps its LDS footprint small so that the U-Net's kernels can share the CUs it occupies.
template <bool KM, int KTI, int NK, bool SINGLE = false, int NW = 4, bool BF = false>
struct Phase1 {
  static constexpr int HC = (M2_HD / 16) / NW;       // hidden-column subtiles per wave
  template <class P>
  static __device__ __forceinline__ void run(const P& pan, const float* xs, float* wb, int wsz, int tid, int lane, int wave,
                                             f32x4 (&acc)[2][HC]) {
    constexpr int XS = M2_K1 + 8;
    constexpr int WS = KM ? (M2_HD + 4) : (M2_BK + 8);
    float* wt = wb + (SINGLE ? 0 : (KTI & 1)) * wsz;
    if (SINGLE && KTI > 0) __syncthreads();      // everyone finished reading the previous tile
    pan.template to_lds<KTI>(wt, tid);
    __syncthreads();
#pragma unroll
    for (int kb = 0; kb < M2_BK / 16; ++kb) {
      float xf[2][4], wf[HC][4];
#pragma unroll
      for (int i = 0; i < 2; ++i) read_frag<false, XS>(xs, i * 16, KTI * (M2_BK / 16) + kb, lane, xf[i]);
#pragma unroll
      for (int j = 0; j < HC; ++j) read_frag<KM, WS>(wt, wave * (HC * 16) + j * 16, kb, lane, wf[j]);
      if constexpr (BF) {
        s16x4 xp[2], wp[HC];
#pragma unroll
        for (int i = 0; i < 2; ++i) xp[i] = pack_bf16x4(xf[i][0], xf[i][1], xf[i][2], xf[i][3]);
#pragma unroll
        for (int j = 0; j < HC; ++j) wp[j] = pack_bf16x4(wf[j][0], wf[j][1], wf[j][2], wf[j][3]);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < HC; ++j) acc[i][j] = mfma_bf16_k16(wp[j], xp[i], acc[i][j]);
      } else {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < HC; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[j][s], xf[i][s], acc[i][j], 0, 0, 0);
      }
    }
    if constexpr (KTI + 1 < NK) Phase1<KM, KTI + 1, NK, SINGLE, NW, BF>::run(pan, xs, wb, wsz, tid, lane, wave, acc);
  }
};

template <bool KM, int KTI, int NK, bool SINGLE = false, int NW = 4, bool BF = false>
struct Phase2 {
  static constexpr int OC = (M2_D2 / 16) / NW;       // output-column subtiles per wave
  template <class P>
  static __device__ __forceinline__ void run(const P& pan, const float* hs, float* wb, int wsz, int tid, int lane, int wave,
                                             f32x4 (&acc)[2][OC]) {
    constexpr int HS = M2_HD + 8;
    constexpr int WS = KM ? (M2_D2 + 4) : (M2_BK + 8);
    float* wt = wb + (SINGLE ? 0 : (KTI & 1)) * wsz;
    if (SINGLE && KTI > 0) __syncthreads();
    pan.template to_lds<KTI>(wt, tid);
    __syncthreads();
#pragma unroll
    for (int kb = 0; kb < M2_BK / 16; ++kb) {
      float xf[2][4], wf[OC][4];
#pragma unroll
      for (int i = 0; i < 2; ++i) read_frag<false, HS>(hs, i * 16, KTI * (M2_BK / 16) + kb, lane, xf[i]);
#pragma unroll
      for (int j = 0; j < OC; ++j) read_frag<KM, WS>(wt, wave * (OC * 16) + j * 16, kb, lane, wf[j]);
      if constexpr (BF) {
        s16x4 xp[2], wp[OC];
#pragma unroll
        for (int i = 0; i < 2; ++i) xp[i] = pack_bf16x4(xf[i][0], xf[i][1], xf[i][2], xf[i][3]);
#pragma unroll
        for (int j = 0; j < OC; ++j) wp[j] = pack_bf16x4(wf[j][0], wf[j][1], wf[j][2], wf[j][3]);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < OC; ++j) acc[i][j] = mfma_bf16_k16(wp[j], xp[i], acc[i][j]);
      } else {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < OC; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[j][s], xf[i][s], acc[i][j], 0, 0, 0);
      }
    }
    if constexpr (KTI + 1 < NK) Phase2<KM, KTI + 1, NK, SINGLE, NW, BF>::run(pan, hs, wb, wsz, tid, lane, wave, acc);
  }
};


}  // namespace mmft

// bf16-operand / fp32-accumulate flavour of the GEMM engine (gfx950, v_mfma_f32_16x16x32_bf16).
//
// Same contract as gemm_f32_kernel:  C[m][n] = epilogue( sum_k X[m][k] * Wt[n][k] ), same pluggable operand loaders
// (DenseMK / DenseKM / Im2col*), same epilogue, same split-K slabs, same XCD-aware tile order.  What changes is the
// arithmetic: both operands are read as fp32 from HBM (the tensors of the path stay fp32), rounded to bf16
// (v_cvt_pk_bf16_f32, round to nearest even) on their way into LDS and multiplied on the bf16 matrix pipe, which runs
// at 16x the fp32-input MFMA rate; sums stay in fp32.  This is the throughput mode BASELINE.json configs[1] names
// ("bf16"); the fp32 engine stays the 1e-4 parity mode.
//
// At 16x the matrix rate every contraction of this path is bound by its fp32 operand traffic, not by the MFMA pipe
// (a 128x128 tile needs 1 KB of operands per k for 32 kFLOP), so the kernel is organised around bytes in flight:
//   * LDS holds both tiles as [row][k] bf16, k contiguous, rows padded by 16 elements (conflict-free ds_read_b128:
//     one read = the 8 k-values a lane feeds to one MFMA);
//   * k-contiguous operands (MK) are staged 8 elements at a time: two 16-byte loads -> one 16-byte LDS store;
//   * m-contiguous operands (KM: dY of a weight gradient, W read transposed) are transposed IN REGISTERS: a thread
//     loads the 4 m-values of 8 consecutive k rows (eight 16-byte loads in flight, coalesced along m across lanes)
//     and writes four 16-byte LDS rows - no transposing LDS read is needed afterwards;
//   * the bias-gradient by-product (column sums of the m-contiguous operand) is taken from those fp32 staging
//     registers, i.e. it is exact, not a sum of rounded values.
#pragma once
#include "gemm_engine.h"

namespace mmft {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
  f32x2 v = {a, b};
  bf16x2 r = __builtin_convertvector(v, bf16x2);          // v_cvt_pk_bf16_f32 (RNE; NaN stays NaN)
  return __builtin_bit_cast(unsigned, r);
}

constexpr int BF16_PAD = 16;      // bf16 elements of row padding: row strides of 96 B (BK 32) / 160 B (BK 64)

template <class CFG, class XL, class WL>
__global__ void __launch_bounds__(256) gemm_bf16_kernel(XL xl, WL wl, Epi epi, int M, int N, int K, int ksplit) {
  constexpr int BM = CFG::BM, BN = CFG::BN, BK = CFG::BK, WM = CFG::WM, WN = CFG::WN;
  static_assert(BK % 32 == 0, "bf16 MFMA consumes 32 k per instruction");
  constexpr int RT = BM / (16 * WM), FT = BN / (16 * WN);
  constexpr int SB = BK + BF16_PAD;                        // LDS row stride in bf16 elements (multiple of 8)
  constexpr int XSZ = BM * SB, WSZ = BN * SB;              // bf16 elements per tile
  // staging groups: MK = (row, 8 k) ; KM = (8 k, 4 m)
  constexpr int XG = XL::KMAJOR ? (BK / 8) * (BM / 4) : BM * (BK / 8);
  constexpr int WG = WL::KMAJOR ? (BK / 8) * (BN / 4) : BN * (BK / 8);
  constexpr int XN = (XG + 255) / 256, WNL = (WG + 255) / 256;
  constexpr int XV = XL::KMAJOR ? 8 : 2, WV = WL::KMAJOR ? 8 : 2;      // 16-byte loads per group
  // ONE LDS buffer: the multiply phase of a 64-deep step is ~0.2 us on the bf16 pipe, so a second buffer would buy
  // nothing - what hides the HBM latency is the next step's loads in flight (registers) and the other workgroups of
  // the CU (40 KB of LDS per 128x128 tile: the register budget, not LDS, sets the occupancy)
  __shared__ __attribute__((aligned(16))) unsigned short lds[XSZ + WSZ];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int gx = (M + BM - 1) / BM, gy = (N + BN - 1) / BN;
  const int lin = (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3);     // XCD-aware order
  const int ntile = lin % gy, mtile = (lin / gy) % gx, zsplit = lin / (gy * gx);
  const int nsplit = (K > 0 && ksplit > 0) ? (K + ksplit - 1) / ksplit : 1;
  if (zsplit >= nsplit) return;
  const int m0 = mtile * BM, n0 = ntile * BN;
  const int kbeg = zsplit * ksplit;
  const int kend = (kbeg + ksplit < K) ? kbeg + ksplit : K;
  const int nk = (kend - kbeg + BK - 1) / BK;

  typename XL::Ctx xc[XN];
  typename WL::Ctx wc[WNL];
  int xk[XN], xo[XN], wk[WNL], wo[WNL];
#pragma unroll
  for (int i = 0; i < XN; ++i) {
    int g = tid + i * 256;
    if (XL::KMAJOR) {
      // k8 fastest: the BK/8 lanes of one m4 group write one contiguous piece of each LDS row (conflict-free) and
      // every global load instruction still covers whole 128-byte lines
      int m4 = g / (BK / 8), k8 = g % (BK / 8);
      xc[i] = xl.ctx(m0 + m4 * 4);
      xk[i] = k8 * 8;
      xo[i] = (m4 * 4) * SB + k8 * 8;
    } else {
      int r = g / (BK / 8), k8 = g % (BK / 8);
      xc[i] = xl.ctx(m0 + r);
      xk[i] = k8 * 8;
      xo[i] = r * SB + k8 * 8;
    }
  }
#pragma unroll
  for (int i = 0; i < WNL; ++i) {
    int g = tid + i * 256;
    if (WL::KMAJOR) {
      int n4 = g / (BK / 8), k8 = g % (BK / 8);
      wc[i] = wl.ctx(n0 + n4 * 4);
      wk[i] = k8 * 8;
      wo[i] = (n4 * 4) * SB + k8 * 8;
    } else {
      int r = g / (BK / 8), k8 = g % (BK / 8);
      wc[i] = wl.ctx(n0 + r);
      wk[i] = k8 * 8;
      wo[i] = r * SB + k8 * 8;
    }
  }

  f32x4 acc[RT][FT];
#pragma unroll
  for (int a = 0; a < RT; ++a)
#pragma unroll
    for (int b = 0; b < FT; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  // exact fp32 column sums of the m-contiguous X operand (bias gradient of the same layer), from the staging registers
  const bool do_cs = XL::KMAJOR && epi.colsum != nullptr && ntile == 0;
  f32x4 cs[XN];
#pragma unroll
  for (int i = 0; i < XN; ++i) cs[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  // register sets = K steps whose loads are in flight.  Measured at config B: a second set changes nothing for the
  // 128x128 tiles (121 / 111 us either way - not latency-bound) and costs the narrow tiles their occupancy.
  constexpr int PF = 1;
  f32x4 xr[PF][XN][XV], wr[PF][WNL][WV];
  const bool x_in = has_fast<XL>::value && fast_interior(xl, m0, BM);
  const bool w_in = has_fast<WL>::value && fast_interior(wl, n0, BN);
  auto gload = [&](auto setc, int k0) {
    constexpr int S = decltype(setc)::value;
    // MK: two loads of 4 consecutive k; KM: eight loads, one per k row.  The interior / edge decision is made ONCE per
    // tile and operand (block-uniform): a per-load choice makes the compiler branch around every load and wait for
    // each one in turn.
    const bool whole = k0 + BK <= kend;
    if (x_in && whole) {
#pragma unroll
      for (int i = 0; i < XN; ++i)
        if (XG % 256 == 0 || tid + i * 256 < XG) {
#pragma unroll
          for (int v = 0; v < XV; ++v) xr[S][i][v] = fast_load(xl, xc[i], k0 + xk[i] + (XL::KMAJOR ? v : 4 * v), kend);
        }
    } else {
#pragma unroll
      for (int i = 0; i < XN; ++i)
        if (XG % 256 == 0 || tid + i * 256 < XG) {
#pragma unroll
          for (int v = 0; v < XV; ++v) xr[S][i][v] = xl.load(xc[i], k0 + xk[i] + (XL::KMAJOR ? v : 4 * v), kend);
        }
    }
    if (w_in && whole) {
#pragma unroll
      for (int i = 0; i < WNL; ++i)
        if (WG % 256 == 0 || tid + i * 256 < WG) {
#pragma unroll
          for (int v = 0; v < WV; ++v) wr[S][i][v] = fast_load(wl, wc[i], k0 + wk[i] + (WL::KMAJOR ? v : 4 * v), kend);
        }
    } else {
#pragma unroll
      for (int i = 0; i < WNL; ++i)
        if (WG % 256 == 0 || tid + i * 256 < WG) {
#pragma unroll
          for (int v = 0; v < WV; ++v) wr[S][i][v] = wl.load(wc[i], k0 + wk[i] + (WL::KMAJOR ? v : 4 * v), kend);
        }
    }
  };
  auto put = [&](unsigned short* tile, int off, const f32x4* r, bool kmajor) {
    if (!kmajor) {
      u32x4 p = {pack_bf16(r[0].x, r[0].y), pack_bf16(r[0].z, r[0].w), pack_bf16(r[1].x, r[1].y), pack_bf16(r[1].z, r[1].w)};
      *reinterpret_cast<u32x4*>(tile + off) = p;
    } else {
      // r[v] = the 4 m-values of k row v: transpose to 4 rows of 8 k
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        u32x4 p = {pack_bf16(r[0][j], r[1][j]), pack_bf16(r[2][j], r[3][j]), pack_bf16(r[4][j], r[5][j]),
                   pack_bf16(r[6][j], r[7][j])};
        *reinterpret_cast<u32x4*>(tile + off + j * SB) = p;
      }
    }
  };
  auto lstore = [&](auto setc) {
    constexpr int S = decltype(setc)::value;
    unsigned short* xs = lds;
    unsigned short* ws = xs + XSZ;
#pragma unroll
    for (int i = 0; i < XN; ++i)
      if (XG % 256 == 0 || tid + i * 256 < XG) {
        put(xs, xo[i], xr[S][i], XL::KMAJOR);
        if (XL::KMAJOR && do_cs) {
#pragma unroll
          for (int v = 0; v < 8; ++v) cs[i] += xr[S][i][v];
        }
      }
#pragma unroll
    for (int i = 0; i < WNL; ++i)
      if (WG % 256 == 0 || tid + i * 256 < WG) put(ws, wo[i], wr[S][i], WL::KMAJOR);
  };
  auto multiply = [&]() {
    const unsigned short* xs = lds;
    const unsigned short* ws = xs + XSZ;
#pragma unroll
    for (int kb = 0; kb < BK / 32; ++kb) {
      bf16x8 xf[RT], wf[FT];
#pragma unroll
      for (int a = 0; a < RT; ++a)
        xf[a] = *reinterpret_cast<const bf16x8*>(xs + ((wm * RT + a) * 16 + (lane & 15)) * SB + kb * 32 + (lane >> 4) * 8);
#pragma unroll
      for (int b = 0; b < FT; ++b)
        wf[b] = *reinterpret_cast<const bf16x8*>(ws + ((wn * FT + b) * 16 + (lane & 15)) * SB + kb * 32 + (lane >> 4) * 8);
#pragma unroll
      for (int a = 0; a < RT; ++a)
#pragma unroll
        for (int b = 0; b < FT; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[b], xf[a], acc[a][b], 0, 0, 0);
    }
  };

  // global -> registers -> LDS -> MFMA.  Step t: registers (tile t, requested a whole step ago) go to LDS, the loads of
  // tile t + 1 are issued at once and stay in flight across the multiply, the barrier and the wait of the next step.
  using S0 = std::integral_constant<int, 0>;
  if (nk > 0) gload(S0{}, kbeg);
  for (int t = 0; t < nk; ++t) {
    lstore(S0{});
    __syncthreads();
    if (t + 1 < nk) gload(S0{}, kbeg + (t + 1) * BK);
    multiply();
    __syncthreads();
  }

#pragma unroll
  for (int a = 0; a < RT; ++a)
#pragma unroll
    for (int b = 0; b < FT; ++b) {
      int m = m0 + (wm * RT + a) * 16 + (lane & 15);
      int n = n0 + (wn * FT + b) * 16 + (lane >> 4) * 4;
      epi.store(m, n, acc[a][b], M, N, zsplit);
    }
  if (XL::KMAJOR && do_cs) {
    // thread (k8, m4) holds the sums over its k rows: combine the BK/8 threads of every m4 group through LDS
    float* red = reinterpret_cast<float*>(lds);            // tiles are dead: reuse (BK/8 * BM floats <= tile bytes)
    __syncthreads();
#pragma unroll
    for (int i = 0; i < XN; ++i) {
      int g = tid + i * 256;
      if (XG % 256 == 0 || g < XG) {
        int m4 = g / (BK / 8), k8 = g % (BK / 8);
        *reinterpret_cast<f32x4*>(red + k8 * BM + m4 * 4) = cs[i];
      }
    }
    __syncthreads();
    for (int mm = tid; mm < BM; mm += 256) {
      float v = 0.f;
#pragma unroll
      for (int k8 = 0; k8 < BK / 8; ++k8) v += red[k8 * BM + mm];
      int m = m0 + mm;
      if (m < M) {
        float* q = epi.colsum + (long long)zsplit * epi.colsum_slab + m;
        *q = (epi.colsum_accum ? *q : 0.f) + v;
      }
    }
  }
}

template <class CFG, class XL, class WL>
inline const char* gemm_bf16_kernel_name() {
  static char buf[192];
  static bool init = false;
  if (!init) {
    snprintf(buf, sizeof(buf), "gemm_bf16_kernel<TileCfg<%d,%d,%d,%d,%d>,%s,%s>", CFG::BM, CFG::BN, CFG::BK, CFG::WM, CFG::WN,
             XL::name(), WL::name());
    init = true;
  }
  return buf;
}

}  // namespace mmft

// fp32 MFMA GEMM engine for gfx950:  C[m][n] = epilogue( sum_k X[m][k] * Wt[n][k] ).
//
// One engine serves every dense contraction on the hot path (level MLPs K4, fusion MLP K9, dense
// fcn K7, conv / dgrad / wgrad K11, convT K14) through pluggable operand loaders:
//   * "MK" operands are k-contiguous in memory  (activations [row][feature], weights [out][in]);
//   * "KM" operands are m-contiguous in memory  (dY for wgrad, W[out][in] read as [k=out][n=in]).
// Each operand is copied into LDS in its natural order (16-B global loads, 16-B LDS stores) and read
// back as MFMA fragments: MK tiles with one ds_read_b128 per 16-deep k block (the four dwords feed
// four consecutive v_mfma_f32_16x16x4_f32 steps; lane quarter q owns k = 4q..4q+3, the same
// permutation on both operands, so the sum over k is complete), KM tiles with ds_read_b32.
// LDS row strides are chosen conflict-free for those reads (MK: BK+8 dwords == 8 mod 16;
// KM: BM+4 dwords, so that the four k rows of a 32-lane group fall on disjoint bank halves).
//
// MFMA roles: the weight-side operand is MFMA "A" (index i), the row-side operand is MFMA "B"
// (index j).  The 16x16 accumulator then holds, per lane, FOUR CONSECUTIVE n for ONE m
// (n = 4*(lane>>4)+r, m = lane&15), i.e. every lane ends with a 16-byte store along the
// contiguous dimension of the row-major output.
//
// Exact fp32: v_mfma_f32_16x16x4_f32 is a k-ordered fmaf chain (guide §3 "FP32-input MFMA").
#pragma once
#include "common.h"
#include <stdlib.h>
#include <type_traits>

// Register prefetch depth (K steps whose global loads are in flight).  A narrow tile has 8-16 MFMAs per wave and K
// step, so its K loop is a chain of global-load round trips (~1.5 us under load): more steps in flight cost registers,
// not LDS, and keep the occupancy.  Large tiles hide the latency behind their 64 MFMAs per step (measured: depth 2
// changes nothing for 128x128).  MMFT_GEMM_PF=<n> forces one depth for every tile (tuning).
constexpr int gemm_prefetch_depth(int bm, int bn, int bk) {
#ifdef MMFT_GEMM_PF
  return MMFT_GEMM_PF;
#else
  if (bm * bn >= 128 * 128) return 1;
  int regs = ((bm * bk + 1023) / 1024 + (bn * bk + 1023) / 1024) * 4;      // VGPRs per in-flight step and thread
  return regs <= 12 ? 4 : (regs <= 24 ? 2 : 1);
#endif
}

namespace mmft {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// One 16-deep k block of a 16x16 output tile from operands held as fp32 fragments "4 consecutive k per lane quarter"
// (the fragment layout of every hand-scheduled kernel of this library):
//   BF = false: four v_mfma_f32_16x16x4_f32 (exact fp32, k-ordered fmaf chain);
//   BF = true : the fragments are rounded to bf16 (v_cvt_pk_bf16_f32) and multiplied by ONE v_mfma_f32_16x16x16_bf16
//               (lane quarter q supplies k = 4q..4q+3 on both operands - the same k placement), fp32 accumulate.
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ s16x4 pack_bf16x4(float a, float b, float c, float d) {
  f32x2_t lo = {a, b}, hi = {c, d};
  bf16x2_t l = __builtin_convertvector(lo, bf16x2_t), h = __builtin_convertvector(hi, bf16x2_t);
  unsigned ul = __builtin_bit_cast(unsigned, l), uh = __builtin_bit_cast(unsigned, h);
  s16x4 r;
  r[0] = (short)(ul & 0xffff); r[1] = (short)(ul >> 16); r[2] = (short)(uh & 0xffff); r[3] = (short)(uh >> 16);
  return r;
}
__device__ __forceinline__ s16x4 pack_bf16x4(f32x4 v) { return pack_bf16x4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ f32x4 mfma_bf16_k16(s16x4 a, s16x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0);
}
template <bool BF>
__device__ __forceinline__ f32x4 mma_k16(f32x4 a, f32x4 b, f32x4 c) {
  if constexpr (BF) {
    return mfma_bf16_k16(pack_bf16x4(a), pack_bf16x4(b), c);
  } else {
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, c, 0, 0, 0);
    return c;
  }
}

template <int BM_, int BN_, int BK_, int WM_, int WN_>
struct TileCfg {
  static constexpr int BM = BM_, BN = BN_, BK = BK_, WM = WM_, WN = WN_;
  static constexpr int PF = gemm_prefetch_depth(BM_, BN_, BK_);
  static_assert(WM_ * WN_ == 4, "4 waves per workgroup");
  static_assert(BM_ % (16 * WM_) == 0 && BN_ % (16 * WN_) == 0, "tile/wave mismatch");
};

// ------------------------------------------------------------------------------------ loaders
// MK loaders: ctx(m) once per thread-row, load(ctx, k, kend) -> X[m][k..k+3] (zero beyond kend/rows)
// KM loaders: ctx(m4) once per thread-column group, load(ctx, k, kend) -> X[k][m..m+3]

struct DenseMK {
  static const char* name() { return "DenseMK"; }
  static constexpr bool KMAJOR = false;
  const float* p;
  const int* idx;  // optional row gather
  long long ld;
  int rows;
  int vec;  // 16-B loads legal (ld % 4 == 0, base aligned)
  struct Ctx {
    const float* row;
  };
  __device__ __forceinline__ Ctx ctx(int m) const {
    Ctx c;
    if (m >= rows) {
      c.row = nullptr;
    } else {
      long long r = idx ? (long long)idx[m] : (long long)m;
      c.row = p + r * ld;
    }
    return c;
  }
  __device__ __forceinline__ f32x4 load(const Ctx& c, int k, int kend) const {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (c.row == nullptr || k >= kend) return v;
    if (vec && k + 3 < kend) return *reinterpret_cast<const f32x4*>(c.row + k);
    v.x = c.row[k];
    if (k + 1 < kend) v.y = c.row[k + 1];
    if (k + 2 < kend) v.z = c.row[k + 2];
    if (k + 3 < kend) v.w = c.row[k + 3];
    return v;
  }
  // interior tiles (all rows valid, 16-byte loads legal, whole K step inside the range): no per-load guards
  __device__ __forceinline__ bool interior(int m0, int bm) const { return vec && m0 + bm <= rows; }
  __device__ __forceinline__ f32x4 load_fast(const Ctx& c, int k) const {
    return *reinterpret_cast<const f32x4*>(c.row + k);
  }
};

struct DenseKM {
  static const char* name() { return "DenseKM"; }
  static constexpr bool KMAJOR = true;
  const float* p;
  const int* kidx;  // optional gather on the reduction rows
  long long ld;
  int cols;
  int vec;
  struct Ctx {
    int m;
  };
  __device__ __forceinline__ Ctx ctx(int m) const { return Ctx{m}; }
  __device__ __forceinline__ f32x4 load(const Ctx& c, int k, int kend) const {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (k >= kend || c.m >= cols) return v;
    long long r = kidx ? (long long)kidx[k] : (long long)k;
    const float* q = p + r * ld + c.m;
    if (vec && c.m + 3 < cols) return *reinterpret_cast<const f32x4*>(q);
    v.x = q[0];
    if (c.m + 1 < cols) v.y = q[1];
    if (c.m + 2 < cols) v.z = q[2];
    if (c.m + 3 < cols) v.w = q[3];
    return v;
  }
  __device__ __forceinline__ bool interior(int m0, int bm) const { return vec && m0 + bm <= cols; }
  __device__ __forceinline__ f32x4 load_fast(const Ctx& c, int k) const {
    long long r = kidx ? (long long)kidx[k] : (long long)k;
    return *reinterpret_cast<const f32x4*>(p + r * ld + c.m);
  }
};

// NHWC implicit-GEMM A operand: X[m = pixel][k = (tap, ci)], C % 4 == 0
// cheap index decode for the implicit-GEMM loaders: channel counts and image sizes on this path are powers of two,
// so k -> (tap, ci) and pixel -> (img, y, x) are shifts (integer division is ~20 VALU instructions per 16-byte
// load and made the small-channel convolutions VALU-bound); tap -> (ky, kx) uses a 16-bit reciprocal of KW.
struct ConvDecode {
  int cshift, wshift, hwshift, kwmagic;   // -1 where the size is not a power of two (then: plain division)
};
inline int log2_exact(int v) {
  for (int s = 0; s < 31; ++s)
    if ((1 << s) == v) return s;
  return -1;
}
inline ConvDecode make_decode(int H, int W, int C, int KW) {
  return ConvDecode{log2_exact(C), log2_exact(W), log2_exact(H * W), (65536 + KW - 1) / KW};
}

struct Im2colMK {
  static const char* name() { return "Im2colMK"; }
  static constexpr bool KMAJOR = false;
  const float* p;
  int H, W, C, KH, KW, pad;
  int rows;  // Nimg*H*W
  ConvDecode dc;
  double unique_bytes(double pixels) const { return 4.0 * pixels * C; }   // host side: the gathered NHWC tensor
  struct Ctx {
    int pix0, y, x;
  };
  __device__ __forceinline__ Ctx ctx(int m) const {
    Ctx c;
    if (m >= rows) {
      c.pix0 = 0;
      c.y = -(1 << 20);
      c.x = 0;
      return c;
    }
    int hw = H * W;
    int img = dc.hwshift >= 0 ? (m >> dc.hwshift) : m / hw;
    int rem = m - img * hw;
    c.y = dc.wshift >= 0 ? (rem >> dc.wshift) : rem / W;
    c.x = rem - c.y * W;
    c.pix0 = img * hw;
    return c;
  }
  // Branch-free: an out-of-image tap or a k beyond the range loads from a clamped (always valid) address and is
  // zeroed by a select - the guarded form cost ~20 exec-mask branches per K step on the 16/32-channel layers.
  __device__ __forceinline__ f32x4 load(const Ctx& c, int k, int kend) const {
    const bool kin = k < kend;
    const int kk = kin ? k : 0;
    int tap = dc.cshift >= 0 ? (kk >> dc.cshift) : kk / C;
    int ci = kk - tap * C;
    int ky = (tap * dc.kwmagic) >> 16;          // exact for tap < 1000 (taps <= 81 here)
    int kx = tap - ky * KW;
    int yy = c.y + ky - pad, xx = c.x + kx - pad;
    const bool ok = kin && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
    yy = ok ? yy : 0;
    xx = ok ? xx : 0;
    f32x4 v = *reinterpret_cast<const f32x4*>(p + ((long long)(c.pix0 + yy * W + xx)) * C + ci);
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    return ok ? v : z;
  }
};

// wgrad B operand: X[k = pixel][n = (tap, ci)] gathered from the NHWC input, C % 4 == 0
struct Im2colKM {
  static const char* name() { return "Im2colKM"; }
  static constexpr bool KMAJOR = true;
  const float* p;
  int H, W, C, KH, KW, pad;
  int cols;  // KH*KW*C
  ConvDecode dc;
  double unique_bytes(double pixels) const { return 4.0 * pixels * C; }   // host side: the gathered NHWC tensor
  struct Ctx {
    int dy, dx, ci;
  };
  __device__ __forceinline__ Ctx ctx(int n) const {
    Ctx c;
    if (n >= cols) {
      c.ci = -1;
      c.dy = c.dx = 0;
      return c;
    }
    int tap = n / C;
    c.ci = n - tap * C;
    int ky = tap / KW;
    c.dy = ky - pad;
    c.dx = tap - ky * KW - pad;
    return c;
  }
  __device__ __forceinline__ f32x4 load(const Ctx& c, int k, int kend) const {
    const bool kin = k < kend && c.ci >= 0;
    const int kk = kin ? k : 0;
    int hw = H * W;
    int img = dc.hwshift >= 0 ? (kk >> dc.hwshift) : kk / hw;
    int rem = kk - img * hw;
    int y = dc.wshift >= 0 ? (rem >> dc.wshift) : rem / W;
    int x = rem - y * W;
    int yy = y + c.dy, xx = x + c.dx;
    const bool ok = kin && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
    yy = ok ? yy : 0;
    xx = ok ? xx : 0;
    const int ci = c.ci >= 0 ? c.ci : 0;
    f32x4 v = *reinterpret_cast<const f32x4*>(p + ((long long)(img * hw + yy * W + xx)) * C + ci);
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    return ok ? v : z;
  }
};

// scalar variants for channel counts that are not a multiple of 4 (first conv of UNet: Ci = 3,
// LayoutNet: Ci = 2, dgrad of the 1-channel heads: "Ci" = 1); every element decodes its own tap
struct Im2colMKScalar {
  static const char* name() { return "Im2colMKScalar"; }
  static constexpr bool KMAJOR = false;
  const float* p;
  int H, W, C, KH, KW, pad;
  int rows;
  double unique_bytes(double pixels) const { return 4.0 * pixels * C; }
  typedef Im2colMK::Ctx Ctx;
  __device__ __forceinline__ Ctx ctx(int m) const {
    Im2colMK v{p, H, W, C, KH, KW, pad, rows, ConvDecode{-1, -1, -1, 0}};
    return v.ctx(m);
  }
  __device__ __forceinline__ float one(const Ctx& c, int k, int kend) const {
    if (k >= kend) return 0.f;
    int tap = k / C;
    int ci = k - tap * C;
    int ky = tap / KW;
    int kx = tap - ky * KW;
    int yy = c.y + ky - pad, xx = c.x + kx - pad;
    if ((unsigned)yy >= (unsigned)H || (unsigned)xx >= (unsigned)W) return 0.f;
    return p[((long long)(c.pix0 + yy * W + xx)) * C + ci];
  }
  __device__ __forceinline__ f32x4 load(const Ctx& c, int k, int kend) const {
    f32x4 v = {one(c, k, kend), one(c, k + 1, kend), one(c, k + 2, kend), one(c, k + 3, kend)};
    return v;
  }
};

struct Im2colKMScalar {
  static const char* name() { return "Im2colKMScalar"; }
  static constexpr bool KMAJOR = true;
  const float* p;
  int H, W, C, KH, KW, pad;
  int cols;
  double unique_bytes(double pixels) const { return 4.0 * pixels * C; }
  struct Ctx {
    int n;
  };
  __device__ __forceinline__ Ctx ctx(int n) const { return Ctx{n}; }
  __device__ __forceinline__ f32x4 load(const Ctx& c, int k, int kend) const {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (k >= kend) return v;
    int hw = H * W;
    int img = k / hw;
    int rem = k - img * hw;
    int y = rem / W;
    int x = rem - y * W;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int n = c.n + j;
      if (n >= cols) continue;
      int tap = n / C;
      int ci = n - tap * C;
      int ky = tap / KW;
      int yy = y + ky - pad, xx = x + (tap - ky * KW) - pad;
      if ((unsigned)yy >= (unsigned)H || (unsigned)xx >= (unsigned)W) continue;
      v[j] = p[((long long)(img * hw + yy * W + xx)) * C + ci];
    }
    return v;
  }
};

// ------------------------------------------------------------------------------------ epilogue
enum { EPI_STORE = 0, EPI_ACCUM = 1, EPI_ADD_ACT = 2, EPI_MASK = 3 };
enum { ACT_NONE = 0, ACT_RELU = 1, ACT_LEAKY = 2 };

struct Epi {
  float* C;
  long long ldc;
  const int* rowidx;   // optional row scatter
  const float* bias;   // optional [N]
  const float* mask;   // EPI_MASK: keep v where mask[maskrow][n] > 0
  const int* maskidx;
  long long ldmask;
  int mode, act;
  float slope;
  long long slab;      // split-K: C + z*slab
  int vec;
  // optional by-product for weight-gradient GEMMs: colsum[z][m] (+)= sum_k X[m][k], i.e. the bias gradient of the same
  // layer, accumulated from the fragments the n-tile-0 workgroups already hold (saves re-reading X in a second kernel)
  float* colsum;
  long long colsum_slab;
  int colsum_accum;

  __device__ __forceinline__ float activate(float v) const {
    if (act == ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == ACT_LEAKY) return v > 0.f ? v : v * slope;
    return v;
  }
  __device__ __forceinline__ void store(int m, int n, f32x4 v, int M, int N, int z) const {
    if (m >= M || n >= N) return;
    long long r = rowidx ? (long long)rowidx[m] : (long long)m;
    float* q = C + (long long)z * slab + r * ldc + n;
    int cnt = N - n < 4 ? N - n : 4;
    if (vec && cnt == 4) {
      // fast path: whole 16-byte group, vector loads of the old value / the mask
      if (bias) {
        v.x += bias[n]; v.y += bias[n + 1]; v.z += bias[n + 2]; v.w += bias[n + 3];
      }
      if (mode == EPI_STORE) {
        v.x = activate(v.x); v.y = activate(v.y); v.z = activate(v.z); v.w = activate(v.w);
      } else if (mode == EPI_ACCUM) {
        v += *reinterpret_cast<const f32x4*>(q);
      } else if (mode == EPI_ADD_ACT) {
        v += *reinterpret_cast<const f32x4*>(q);
        v.x = activate(v.x); v.y = activate(v.y); v.z = activate(v.z); v.w = activate(v.w);
      } else {
        long long mr = maskidx ? (long long)maskidx[m] : (long long)m;
        const float* mk = mask + mr * ldmask + n;
        f32x4 mv;
        if ((ldmask & 3) == 0 && (reinterpret_cast<uintptr_t>(mask) & 15) == 0) mv = *reinterpret_cast<const f32x4*>(mk);
        else mv = f32x4{mk[0], mk[1], mk[2], mk[3]};
        v.x = mv.x > 0.f ? v.x : 0.f; v.y = mv.y > 0.f ? v.y : 0.f;
        v.z = mv.z > 0.f ? v.z : 0.f; v.w = mv.w > 0.f ? v.w : 0.f;
      }
      *reinterpret_cast<f32x4*>(q) = v;
      return;
    }
    float vv[4] = {v.x, v.y, v.z, v.w};
    const float* mk = nullptr;
    if (mode == EPI_MASK) {
      long long mr = maskidx ? (long long)maskidx[m] : (long long)m;
      mk = mask + mr * ldmask + n;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (j < cnt) {
        float t = vv[j] + (bias ? bias[n + j] : 0.f);
        if (mode == EPI_STORE) t = activate(t);
        else if (mode == EPI_ACCUM) t += q[j];
        else if (mode == EPI_ADD_ACT) t = activate(t + q[j]);
        else t = mk[j] > 0.f ? t : 0.f;
        q[j] = t;
      }
    }
  }
};

// ------------------------------------------------------------------------------------ kernel
// loaders that offer an unguarded path for interior tiles (interior() + load_fast()); the others keep load()
template <class L, class = void>
struct has_fast : std::false_type {};
template <class L>
struct has_fast<L, std::void_t<decltype(&L::load_fast)>> : std::true_type {};
template <class L>
__device__ __forceinline__ bool fast_interior(const L& l, int m0, int bm) {
  if constexpr (has_fast<L>::value) return l.interior(m0, bm);
  else return false;
}
template <class L, class C>
__device__ __forceinline__ f32x4 fast_load(const L& l, const C& c, int k, int kend) {
  if constexpr (has_fast<L>::value) return l.load_fast(c, k);
  else return l.load(c, k, kend);
}

// MK tiles with BK = 16: a row is 4 16-byte chunks at stride BK + 8 = 24 dwords, so rows r and r + 1 written by 8
// consecutive lanes overlap in two of the eight 16-byte bank groups (33 % LDS conflict cycles measured).  Rows r and
// r + 2 do not: lanes 4-7 of every 8 take the row two further (rows 0,2,1,3 within each block of four).
template <int BK>
__device__ __forceinline__ int lds_row_swizzle(int r) {
  if (BK == 16) return (r & ~3) | ((r & 1) << 1) | ((r >> 1) & 1);
  return r;
}

template <bool KMAJOR, int S>
__device__ __forceinline__ void read_frag(const float* tile, int row0, int kb, int lane, float (&f)[4]) {
  if (!KMAJOR) {
    f32x4 v = *reinterpret_cast<const f32x4*>(tile + (row0 + (lane & 15)) * S + kb * 16 + (lane >> 4) * 4);
    f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
  } else {
    const float* q = tile + (kb * 16 + 4 * (lane >> 4)) * S + row0 + (lane & 15);
    f[0] = q[0]; f[1] = q[S]; f[2] = q[2 * S]; f[3] = q[3 * S];
  }
}

template <class CFG, class XL, class WL>
__global__ void __launch_bounds__(256) gemm_f32_kernel(XL xl, WL wl, Epi epi, int M, int N, int K, int ksplit) {
  constexpr int BM = CFG::BM, BN = CFG::BN, BK = CFG::BK, WM = CFG::WM, WN = CFG::WN;
  constexpr int RT = BM / (16 * WM), FT = BN / (16 * WN);
  constexpr int XS = XL::KMAJOR ? (BM + 4) : (BK + 8);
  constexpr int WS = WL::KMAJOR ? (BN + 4) : (BK + 8);
  constexpr int XSZ = XL::KMAJOR ? BK * XS : BM * XS;
  constexpr int WSZ = WL::KMAJOR ? BK * WS : BN * WS;
  constexpr int XG = BM * BK / 4, WG = BN * BK / 4;       // 16-byte groups per tile
  constexpr int XN = (XG + 255) / 256, WNL = (WG + 255) / 256;
  __shared__ __attribute__((aligned(16))) float lds[2 * (XSZ + WSZ)];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  // XCD-aware tile order.  Workgroup ids go round-robin over the 8 XCDs (each with its own L2), so consecutive ids
  // would put neighbouring tiles - which share halo rows in the implicit-GEMM loaders, X rows across N tiles and
  // pixel rows across adjacent split-K ranges - on eight different L2s.  The launch is 1-D, padded to a multiple of
  // 8 workgroups; id w takes position (w % 8) * (count / 8) + w / 8 of the (n fastest, then m, then k-split) order,
  // so every XCD walks one contiguous eighth of the tiles.
  const int gx = (M + BM - 1) / BM, gy = (N + BN - 1) / BN;
  const int lin = (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3);
  const int ntile = lin % gy, mtile = (lin / gy) % gx, zsplit = lin / (gy * gx);
  const int nsplit = (K > 0 && ksplit > 0) ? (K + ksplit - 1) / ksplit : 1;
  if (zsplit >= nsplit) return;                                             // padding workgroups
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
      int kk = g / (BM / 4), m4 = g % (BM / 4);
      xc[i] = xl.ctx(m0 + m4 * 4);
      xk[i] = kk;
      xo[i] = kk * XS + m4 * 4;
    } else {
      int r = g / (BK / 4), k4 = g % (BK / 4);
      r = lds_row_swizzle<BK>(r);
      xc[i] = xl.ctx(m0 + r);
      xk[i] = k4 * 4;
      xo[i] = r * XS + k4 * 4;
    }
  }
#pragma unroll
  for (int i = 0; i < WNL; ++i) {
    int g = tid + i * 256;
    if (WL::KMAJOR) {
      int kk = g / (BN / 4), n4 = g % (BN / 4);
      wc[i] = wl.ctx(n0 + n4 * 4);
      wk[i] = kk;
      wo[i] = kk * WS + n4 * 4;
    } else {
      int r = g / (BK / 4), k4 = g % (BK / 4);
      r = lds_row_swizzle<BK>(r);
      wc[i] = wl.ctx(n0 + r);
      wk[i] = k4 * 4;
      wo[i] = r * WS + k4 * 4;
    }
  }

  f32x4 acc[RT][FT];
#pragma unroll
  for (int a = 0; a < RT; ++a)
#pragma unroll
    for (int b = 0; b < FT; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  const bool do_cs = epi.colsum != nullptr && ntile == 0 && wn == 0;      // wave-uniform
  float cs[RT];
#pragma unroll
  for (int a = 0; a < RT; ++a) cs[a] = 0.f;

  // global -> register -> LDS pipeline.  PF register sets: with PF = 2 the loads of tile t+2 are issued while tile t is
  // multiplied and tile t+1 waits in the other set, so a load has two k-steps (~1.7 us at BK = 16) to come back from
  // HBM instead of one; the set index is a compile-time constant so that the arrays stay in registers.
  constexpr int PF = CFG::PF;
  f32x4 xr[PF][XN], wr[PF][WNL];
  const bool x_in = has_fast<XL>::value && fast_interior(xl, m0, BM);     // block-uniform
  const bool w_in = has_fast<WL>::value && fast_interior(wl, n0, BN);
  auto gload = [&](auto set, int k0) {
    constexpr int S = decltype(set)::value;
    const bool whole = k0 + BK <= kend;
    if (x_in && whole) {
#pragma unroll
      for (int i = 0; i < XN; ++i)
        if (XG % 256 == 0 || tid + i * 256 < XG) xr[S][i] = fast_load(xl, xc[i], k0 + xk[i], kend);
    } else {
#pragma unroll
      for (int i = 0; i < XN; ++i)
        if (XG % 256 == 0 || tid + i * 256 < XG) xr[S][i] = xl.load(xc[i], k0 + xk[i], kend);
    }
    if (w_in && whole) {
#pragma unroll
      for (int i = 0; i < WNL; ++i)
        if (WG % 256 == 0 || tid + i * 256 < WG) wr[S][i] = fast_load(wl, wc[i], k0 + wk[i], kend);
    } else {
#pragma unroll
      for (int i = 0; i < WNL; ++i)
        if (WG % 256 == 0 || tid + i * 256 < WG) wr[S][i] = wl.load(wc[i], k0 + wk[i], kend);
    }
  };
  auto lstore = [&](auto set, int buf) {
    constexpr int S = decltype(set)::value;
    float* xs = lds + buf * (XSZ + WSZ);
    float* ws = xs + XSZ;
#pragma unroll
    for (int i = 0; i < XN; ++i)
      if (XG % 256 == 0 || tid + i * 256 < XG) *reinterpret_cast<f32x4*>(xs + xo[i]) = xr[S][i];
#pragma unroll
    for (int i = 0; i < WNL; ++i)
      if (WG % 256 == 0 || tid + i * 256 < WG) *reinterpret_cast<f32x4*>(ws + wo[i]) = wr[S][i];
  };
  auto multiply = [&](int buf) {
    const float* xs = lds + buf * (XSZ + WSZ);
    const float* ws = xs + XSZ;
#pragma unroll
    for (int kb = 0; kb < BK / 16; ++kb) {
      float xf[RT][4], wf[FT][4];
#pragma unroll
      for (int a = 0; a < RT; ++a) read_frag<XL::KMAJOR, XS>(xs, (wm * RT + a) * 16, kb, lane, xf[a]);
#pragma unroll
      for (int b = 0; b < FT; ++b) read_frag<WL::KMAJOR, WS>(ws, (wn * FT + b) * 16, kb, lane, wf[b]);
      if (do_cs) {
#pragma unroll
        for (int a = 0; a < RT; ++a) cs[a] += (xf[a][0] + xf[a][1]) + (xf[a][2] + xf[a][3]);
      }
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int a = 0; a < RT; ++a)
#pragma unroll
          for (int b = 0; b < FT; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[b][s], xf[a][s], acc[a][b], 0, 0, 0);
    }
  };
  static_assert(PF == 1 || PF == 2 || PF == 4, "prefetch depth");
  // Step tau multiplies LDS buffer tau & 1.  Register set j = tau % PF held tile tau; it went to LDS during step
  // tau - 1, so step tau starts by loading tile tau + PF into it and ends by moving tile tau + 1 (set (j + 1) % PF,
  // requested PF - 1 steps ago) to the other LDS buffer.  The set index is a compile-time constant (unrolled by PF).
  auto prologue = [&](auto self, auto jc) -> void {
    constexpr int j = decltype(jc)::value;
    if constexpr (j < PF) {
      if (j < nk) gload(std::integral_constant<int, j>{}, kbeg + j * BK);
      self(self, std::integral_constant<int, j + 1>{});
    }
  };
  prologue(prologue, std::integral_constant<int, 0>{});
  if (nk > 0) lstore(std::integral_constant<int, 0>{}, 0);
  __syncthreads();
  auto substeps = [&](auto self, auto jc, int t) -> void {
    constexpr int j = decltype(jc)::value;
    if constexpr (j < PF) {
      const int tau = t + j;
      if (tau < nk) {
        if (tau + PF < nk) gload(std::integral_constant<int, j>{}, kbeg + (tau + PF) * BK);
        multiply(tau & 1);
        if (tau + 1 < nk) lstore(std::integral_constant<int, (j + 1) % PF>{}, (tau + 1) & 1);
        __syncthreads();
        self(self, std::integral_constant<int, j + 1>{}, t);
      }
    }
  };
  for (int t = 0; t < nk; t += PF) substeps(substeps, std::integral_constant<int, 0>{}, t);

#pragma unroll
  for (int a = 0; a < RT; ++a)
#pragma unroll
    for (int b = 0; b < FT; ++b) {
      int m = m0 + (wm * RT + a) * 16 + (lane & 15);
      int n = n0 + (wn * FT + b) * 16 + (lane >> 4) * 4;
      epi.store(m, n, acc[a][b], M, N, zsplit);
    }
  if (do_cs) {
#pragma unroll
    for (int a = 0; a < RT; ++a) {
      float v = cs[a];                                  // lane quarter q holds the k = 4q..4q+3 share of row lane&15
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      int m = m0 + (wm * RT + a) * 16 + lane;
      if (lane < 16 && m < M) {
        float* q = epi.colsum + (long long)zsplit * epi.colsum_slab + m;
        *q = (epi.colsum_accum ? *q : 0.f) + v;
      }
    }
  }
}

// ------------------------------------------------------------------------------------ dispatch
template <class CFG, class XL, class WL>
inline const char* gemm_kernel_name() {
  static char buf[192];
  static bool init = false;
  if (!init) {
    snprintf(buf, sizeof(buf), "gemm_f32_kernel<TileCfg<%d,%d,%d,%d,%d>,%s,%s>", CFG::BM, CFG::BN, CFG::BK, CFG::WM, CFG::WN,
             XL::name(), WL::name());
    init = true;
  }
  return buf;
}

// algorithmic bytes of one operand: a dense operand is read once (rows x depth); an implicit-GEMM operand is the NHWC
// tensor it gathers from (pixels x channels), not the KH*KW times larger virtual im2col matrix
template <class L, class = void>
struct has_unique_bytes : std::false_type {};
template <class L>
struct has_unique_bytes<L, std::void_t<decltype(&L::unique_bytes)>> : std::true_type {};
template <class L>
inline double operand_bytes(const L& l, double rows, double depth) {
  if constexpr (has_unique_bytes<L>::value) return l.unique_bytes(L::KMAJOR ? depth : rows);
  else return 4.0 * rows * depth;
}

// math mode of the MFMA-bound contractions (mmft_set_math_mode): 0 = exact fp32 MFMA, 1 = bf16 operands / fp32 sums
int math_mode();
template <class CFG, class XL, class WL>
__global__ void gemm_bf16_kernel(XL xl, WL wl, Epi epi, int M, int N, int K, int ksplit);
template <class CFG, class XL, class WL>
inline const char* gemm_bf16_kernel_name();

template <class CFG, class XL, class WL>
inline void launch_cfg(const XL& xl, const WL& wl, const Epi& epi, int M, int N, int K, int splits, hipStream_t st) {
  int ksplit = K;
  if (splits > 1) {
    // 16-row granularity whatever BK is, so that the slab count equals effective_splits() (the callers size and sum
    // the slabs with it); a K range that is not a multiple of BK ends in a guarded, zero-filled step
    int per = (K + splits - 1) / splits;
    ksplit = ((per + 15) / 16) * 16;
    splits = (K + ksplit - 1) / ksplit;
  }
  long long wgs = (long long)cdiv(M, CFG::BM) * cdiv(N, CFG::BN) * splits;
  dim3 grid((unsigned)((wgs + 7) / 8 * 8));                                  // 1-D, padded: see the XCD-aware tile order
  // algorithmic work of this launch: 2*M*N*K flops; bytes = the three matrices touched once
  const double alg_bytes = operand_bytes(xl, M, K) + operand_bytes(wl, N, K) + 4.0 * (double)M * N;
  if (math_mode() == 1) {
    // every tile runs 64-deep K steps in this mode (two bf16 MFMAs of 32 k; 16 KB of fp32 operands per 128 rows and
    // step in flight); the slab geometry is unchanged (ksplit is a multiple of 16, a K range that is not a multiple
    // of 64 ends in a zero-filled step)
    using C2 = TileCfg<CFG::BM, CFG::BN, 64, CFG::WM, CFG::WN>;
    MMFT_LAUNCH((gemm_bf16_kernel_name<C2, XL, WL>()), 2.0 * M * N * K, alg_bytes, (gemm_bf16_kernel<C2, XL, WL>), grid, dim3(256), st, xl, wl, epi, M, N, K, ksplit);
    return;
  }
  MMFT_LAUNCH((gemm_kernel_name<CFG, XL, WL>()), 2.0 * M * N * K, alg_bytes, (gemm_f32_kernel<CFG, XL, WL>), grid, dim3(256), st, xl, wl, epi, M, N, K, ksplit);
}

// K up to which short GEMMs (few output tiles) take the 64-deep K step; MMFT_GEMM_DEEPK_LIMIT overrides (tuning)
inline int gemm_deepk_limit() {
  static int lim = -1;
  if (lim < 0) {
    const char* e = getenv("MMFT_GEMM_DEEPK_LIMIT");
    lim = e ? atoi(e) : 1280;      // 3x3 convolutions with up to 128 input channels (K = 1152) included
  }
  return lim;
}

// number of split-K slabs launch_gemm will actually use for a requested count
inline int effective_splits(int K, int splits) {
  if (splits <= 1) return 1;
  int per = (K + splits - 1) / splits;
  int ksplit = ((per + 15) / 16) * 16;
  return (K + ksplit - 1) / ksplit;
}

// Tile choice.  Candidates are ordered by tile area; the first one that (a) does not exceed M / N rounded up to 16
// (Co = 16/32 weight-gradient GEMMs must not be padded to 128 rows of wasted MFMA work) and (b) still fills the
// chip is taken.  "Fills" is tried at >= 1024, then 512, then 256 workgroups (256 CUs): the level-serial GEMMs of
// the sweep (M ~ 1k-8k rows, K = 128/256) are latency-bound at 8 K-steps per tile, so thread-level parallelism
// (4 workgroups per CU) hides what a short K loop cannot.
template <class XL, class WL>
inline int launch_gemm(const XL& xl, const WL& wl, const Epi& epi, int M, int N, int K, int splits, hipStream_t st) {
  if (M <= 0 || N <= 0) return MMFT_OK;
  const int sp = splits > 1 ? splits : 1;
  const int mcap = ((M + 15) / 16) * 16, ncap = ((N + 15) / 16) * 16;
  static const int cand[][2] = {{128, 128}, {64, 128}, {128, 64}, {32, 128}, {64, 64}, {128, 32}, {32, 64},
                                {16, 128}, {64, 32}, {128, 16}, {16, 64}, {64, 16}};
  const int ncand = sizeof(cand) / sizeof(cand[0]);
  auto fits = [&](int i) {
    int bm = cand[i][0], bn = cand[i][1];
    if (bn > ncap && bn != 16) return false;            // do not pad features beyond the next multiple of 16...
    if (bm > mcap && bm != 16 && !(bm == 64 && bn <= 32)) return false;   // ...nor rows (smallest shapes excepted)
    return true;
  };
  auto wgs = [&](int i) { return (long long)cdiv(M, cand[i][0]) * cdiv(N, cand[i][1]) * sp; };
  int pick = -1;
  // split-K launches reach the workgroup count through their slabs: keep the largest tile that gives >= 256
  const long long want[3] = {splits > 1 ? 256 : 1024, splits > 1 ? 256 : 512, 256};
  for (int t = 0; t < 3 && pick < 0; ++t)
    for (int i = 0; i < ncand; ++i)
      if (fits(i) && wgs(i) >= want[t]) {
        pick = i;
        break;
      }
  if (pick < 0) {                                         // tiny problem: the fitting candidate with most workgroups
    long long best = -1;
    for (int i = 0; i < ncand; ++i)
      if (fits(i) && wgs(i) > best) {
        best = wgs(i);
        pick = i;
      }
  }
  if (pick < 0) pick = ncand - 1;
  // Short level-serial GEMMs (few rows, K <= 512): a 16-deep K step costs one global round trip (~1 us) for 4-16
  // MFMAs, so the tile's K loop is latency-bound.  BK = 64 makes a quarter of the round trips with four times the
  // bytes in flight per thread.
  if (splits <= 1 && K >= 64 && K <= gemm_deepk_limit() && (long long)M * N <= (1ll << 23) && ncap >= 64 && mcap >= 32) {
    launch_cfg<TileCfg<32, 64, 64, 2, 2>>(xl, wl, epi, M, N, K, splits, st);
    return check_launch("gemm_f32");
  }
  // tuning hook (tools/bench_gemm.py): MMFT_GEMM_FORCE=<bm>x<bn>x<bk> forces one of the wide-K-step tiles
  if (const char* f = getenv("MMFT_GEMM_FORCE")) {
    int fbm = 0, fbn = 0, fbk = 0;
    if (sscanf(f, "%dx%dx%d", &fbm, &fbn, &fbk) == 3) {
      if (fbm == 128 && fbn == 128 && fbk == 32) { launch_cfg<TileCfg<128, 128, 32, 2, 2>>(xl, wl, epi, M, N, K, splits, st); return check_launch("gemm_f32"); }
      if (fbm == 128 && fbn == 64 && fbk == 32) { launch_cfg<TileCfg<128, 64, 32, 2, 2>>(xl, wl, epi, M, N, K, splits, st); return check_launch("gemm_f32"); }
      if (fbm == 64 && fbn == 128 && fbk == 32) { launch_cfg<TileCfg<64, 128, 32, 2, 2>>(xl, wl, epi, M, N, K, splits, st); return check_launch("gemm_f32"); }
      if (fbm == 64 && fbn == 64 && fbk == 32) { launch_cfg<TileCfg<64, 64, 32, 2, 2>>(xl, wl, epi, M, N, K, splits, st); return check_launch("gemm_f32"); }
      if (fbm == 64 && fbn == 64 && fbk == 64) { launch_cfg<TileCfg<64, 64, 64, 2, 2>>(xl, wl, epi, M, N, K, splits, st); return check_launch("gemm_f32"); }
      if (fbm == 128 && fbn == 64 && fbk == 16) { launch_cfg<TileCfg<128, 64, 16, 2, 2>>(xl, wl, epi, M, N, K, splits, st); return check_launch("gemm_f32"); }
      if (fbm == 64 && fbn == 128 && fbk == 16) { launch_cfg<TileCfg<64, 128, 16, 2, 2>>(xl, wl, epi, M, N, K, splits, st); return check_launch("gemm_f32"); }
      if (fbm == 64 && fbn == 64 && fbk == 16) { launch_cfg<TileCfg<64, 64, 16, 2, 2>>(xl, wl, epi, M, N, K, splits, st); return check_launch("gemm_f32"); }
    }
  }
  // Split-K launches with a narrow tile (weight gradients of 16/32-channel convolutions: Co x 9Ci outputs over
  // 524 288 pixels) have 8-16 MFMAs per wave and 16-deep K step: the K loop is a chain of global round trips.  A
  // deeper K step makes 2-4x fewer of them with as many bytes in flight each.
  if (splits > 1) {
    static int deep = -1;                         // MMFT_GEMM_SPLITK_BK=16 switches the deep-K tiles off (tuning hook)
    if (deep < 0) {
      const char* e = getenv("MMFT_GEMM_SPLITK_BK");
      deep = e ? atoi(e) : 64;
    }
    const int key = cand[pick][0] * 1000 + cand[pick][1];
    if (deep >= 32) {
      // measured at config B (us per launch, BK = 16 / 32 / 64): 64x32 scalar 119 / - / 66, 64x16 66 / - / 38,
      // 16x128 118 / 109 / 138, 32x128 74 / 63 / -, 64x128 56 / 46 / -
      if (key == 64016) { launch_cfg<TileCfg<64, 16, 64, 4, 1>>(xl, wl, epi, M, N, K, splits, st); return check_launch("gemm_f32"); }
      if (key == 64032) { launch_cfg<TileCfg<64, 32, 64, 4, 1>>(xl, wl, epi, M, N, K, splits, st); return check_launch("gemm_f32"); }
      if (key == 16128) { launch_cfg<TileCfg<16, 128, 32, 1, 4>>(xl, wl, epi, M, N, K, splits, st); return check_launch("gemm_f32"); }
      if (key == 16064) { launch_cfg<TileCfg<16, 64, 32, 1, 4>>(xl, wl, epi, M, N, K, splits, st); return check_launch("gemm_f32"); }
      if (key == 32128) { launch_cfg<TileCfg<32, 128, 32, 2, 2>>(xl, wl, epi, M, N, K, splits, st); return check_launch("gemm_f32"); }
      if (key == 128032) { launch_cfg<TileCfg<128, 32, 32, 4, 1>>(xl, wl, epi, M, N, K, splits, st); return check_launch("gemm_f32"); }
      if (key == 128016) { launch_cfg<TileCfg<128, 16, 32, 4, 1>>(xl, wl, epi, M, N, K, splits, st); return check_launch("gemm_f32"); }
      if (key == 64128) { launch_cfg<TileCfg<64, 128, 32, 2, 2>>(xl, wl, epi, M, N, K, splits, st); return check_launch("gemm_f32"); }
    }
  }
#define MMFT_GO(BM, BN, WM, WN) launch_cfg<TileCfg<BM, BN, 16, WM, WN>>(xl, wl, epi, M, N, K, splits, st)
  switch (cand[pick][0] * 1000 + cand[pick][1]) {
    case 128128: MMFT_GO(128, 128, 2, 2); break;
    case 64128: MMFT_GO(64, 128, 2, 2); break;
    case 128064: MMFT_GO(128, 64, 2, 2); break;
    case 32128: MMFT_GO(32, 128, 2, 2); break;
    case 64064: MMFT_GO(64, 64, 2, 2); break;
    case 128032: MMFT_GO(128, 32, 4, 1); break;
    case 32064: MMFT_GO(32, 64, 2, 2); break;
    case 16128: MMFT_GO(16, 128, 1, 4); break;
    case 64032: MMFT_GO(64, 32, 4, 1); break;
    case 128016: MMFT_GO(128, 16, 4, 1); break;
    case 16064: MMFT_GO(16, 64, 1, 4); break;
    default: MMFT_GO(64, 16, 4, 1); break;
  }
#undef MMFT_GO
  return check_launch("gemm_f32");
}

// deterministic split-K combine: out[i] = (accumulate ? out[i] : 0) + sum_z slab[z][i], fixed z order
int launch_slab_reduce(const float* slabs, int splits, long long elems, float* out, int accumulate, hipStream_t st);
int launch_slab_reduce_strided(const float* slabs, int splits, long long stride, long long elems, float* out, int accumulate,
                               hipStream_t st);

}  // namespace mmft
#include "gemm_bf16.h"

// bf16-storage ConvTranspose2d(k = 2, s = 2, bias) of the U-Net's Up blocks (src/Unet.py:53): per input pixel a
// [4 Co] x [Ci] matrix product whose four Co-wide pieces land on the 2 x 2 output pixels.  Ci -> Co = Ci / 2 with
// Ci in {128, 64, 32}.  The output is written straight into its channel slice of the concatenation buffer
// (torch.cat([x2, x1], 1), src/Unet.py:67), the backward reads the slice of the concatenation's gradient in place.
//   forward : u[(n, 2y+a, 2x+b)][co] = bias[co] + sum_ci Wm[(a,b,co)][ci] x[(n,y,x)][ci]
//   dgrad   : dx[p][ci]   = sum_k gu[p][k] Wm[k][ci],   k = (a,b,co), gu[p][k] = g[(n, 2y+a, 2x+b)][co]
//   wgrad   : dWm[k][ci]  = sum_p gu[p][k] x[p][ci];    db[co] = sum_{p,a,b} gu[p][(a,b,co)]
// No LDS: the operands that are reused (weights) sit in registers as pre-packed MFMA A fragments, split over the four
// waves of a workgroup by output block; the pixel operands are 8-byte "natural" fragment loads (lane = pixel, 4 channels).
// The weight gradient contracts over pixels: both operands are transposed by the matrix unit (D = A * I), as in
// conv_wgrad_narrow.h; per-workgroup slabs are added in a fixed order.
#include "unet16.h"

namespace mmft {

struct U16ConvTArgs {
  const u16* x;        // [P][CI]           P = N * h * w input pixels
  const u16* wpk;      // packed A fragments (forward: rows (a,b,co), K = ci; dgrad: rows ci, K = (a,b,co))
  const float* bias;   // [CO]
  u16* u;              // output / gradient slice: pixel (n, Y, X) at u + ((n * 2h + Y) * 2w + X) * ldu
  int ldu;
  u16* dx;             // [P][CI]           (dgrad)
  float* slabs;        // [gridDim.x][4 CO * CI + 4 CO]   (wgrad)
  int N, h, w;
  long long P;
};

__device__ __forceinline__ long long ct_out_pixel(const U16ConvTArgs& a, long long p, int ab) {
  const int x = (int)(p % a.w), y = (int)((p / a.w) % a.h);
  const long long n = p / ((long long)a.w * a.h);
  return (n * (2 * a.h) + 2 * y + (ab >> 1)) * (2 * a.w) + 2 * x + (ab & 1);
}

template <int CI>
__global__ void __launch_bounds__(256) u16_convt_fwd_kernel(U16ConvTArgs a) {
  constexpr int CO = CI / 2, MB = 4 * CO / 16, CB = CI / 16, MBW = MB / 4;
  static_assert(MB % 4 == 0, "row blocks split over four waves");
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, q = lane >> 4;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  s16x4 wf[MBW][CB];
  f32x4 bv[MBW];
#pragma unroll
  for (int i = 0; i < MBW; ++i) {
    const int m = wave + 4 * i;
#pragma unroll
    for (int c = 0; c < CB; ++c) wf[i][c] = *reinterpret_cast<const s16x4*>(a.wpk + (((long long)m * CB + c) * 64 + lane) * 4);
    const int co = (m * 16 + 4 * q) % CO;
    bv[i] = a.bias ? *reinterpret_cast<const f32x4*>(a.bias + co) : zero;
  }
  const long long steps = (a.P + 15) / 16;
  for (long long s = blockIdx.x; s < steps; s += gridDim.x) {
    const long long p = s * 16 + r;
    const bool ok = p < a.P;
    s16x4 xf[CB];
#pragma unroll
    for (int c = 0; c < CB; ++c) {
      const u32x2 v = *reinterpret_cast<const u32x2*>(a.x + (ok ? p : 0) * CI + c * 16 + 4 * q);
      const u32x2 z2 = {0u, 0u};
      xf[c] = as_s16x4(ok ? v : z2);
    }
#pragma unroll
    for (int i = 0; i < MBW; ++i) {
      f32x4 acc = bv[i];
#pragma unroll
      for (int c = 0; c < CB; ++c) acc = mfma_bf16_k16(wf[i][c], xf[c], acc);
      // lane (pixel r, q): rows 16 m + 4 q .. + 3 = (ab, co .. co + 3)
      const int row = (wave + 4 * i) * 16 + 4 * q, ab = row / CO, co = row % CO;
      if (ok) *reinterpret_cast<s16x4*>(a.u + ct_out_pixel(a, p, ab) * a.ldu + co) = pack_bf16x4(acc);
    }
  }
}

template <int CI>
__global__ void __launch_bounds__(256) u16_convt_dgrad_kernel(U16ConvTArgs a) {
  constexpr int CO = CI / 2, MB = CI / 16, KB = 4 * CO / 16, MBW = (MB + 3) / 4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, q = lane >> 4;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  s16x4 wf[MBW][KB];
#pragma unroll
  for (int i = 0; i < MBW; ++i) {
    const int m = wave + 4 * i;
#pragma unroll
    for (int k = 0; k < KB; ++k)
      wf[i][k] = m < MB ? *reinterpret_cast<const s16x4*>(a.wpk + (((long long)m * KB + k) * 64 + lane) * 4) : s16x4{0, 0, 0, 0};
  }
  if (wave >= MB) return;                               // Ci = 32: two row blocks, waves 2 and 3 have nothing to do
  const long long steps = (a.P + 15) / 16;
  for (long long s = blockIdx.x; s < steps; s += gridDim.x) {
    const long long p = s * 16 + r;
    const bool ok = p < a.P;
    s16x4 gf[KB];
#pragma unroll
    for (int k = 0; k < KB; ++k) {
      const int kk = k * 16 + 4 * q, ab = kk / CO, co = kk % CO;
      const u32x2 v = *reinterpret_cast<const u32x2*>(a.u + ct_out_pixel(a, ok ? p : 0, ab) * a.ldu + co);
      const u32x2 z2 = {0u, 0u};
      gf[k] = as_s16x4(ok ? v : z2);
    }
#pragma unroll
    for (int i = 0; i < MBW; ++i) {
      const int m = wave + 4 * i;
      if (m < MB) {
        f32x4 acc = zero;
#pragma unroll
        for (int k = 0; k < KB; ++k) acc = mfma_bf16_k16(wf[i][k], gf[k], acc);
        if (ok) *reinterpret_cast<s16x4*>(a.dx + p * CI + m * 16 + 4 * q) = pack_bf16x4(acc);
      }
    }
  }
}

// wave w owns the k blocks w, w + 4, ... (KBW = CO / 16 of them); every workgroup writes one slab
template <int CI>
__global__ void __launch_bounds__(256) u16_convt_wgrad_kernel(U16ConvTArgs a) {
  constexpr int CO = CI / 2, KB = 4 * CO / 16, CB = CI / 16, KBW = KB / 4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, q = lane >> 4;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  s16x4 ident;
#pragma unroll
  for (int j = 0; j < 4; ++j) ident[j] = (4 * q + j == r) ? (short)0x3F80 : (short)0;
  auto transpose = [&](s16x4 natural) { return pack_bf16x4(mfma_bf16_k16(natural, ident, zero)); };
  f32x4 acc[KBW][CB];
  float cs[KBW][4];
#pragma unroll
  for (int i = 0; i < KBW; ++i) {
#pragma unroll
    for (int c = 0; c < CB; ++c) acc[i][c] = zero;
#pragma unroll
    for (int j = 0; j < 4; ++j) cs[i][j] = 0.f;
  }
  const long long steps = (a.P + 15) / 16;
  for (long long s = blockIdx.x; s < steps; s += gridDim.x) {
    const long long p = s * 16 + r;
    const bool ok = p < a.P;
    const u32x2 z2 = {0u, 0u};
    s16x4 xt[CB];
#pragma unroll
    for (int c = 0; c < CB; ++c) {
      const u32x2 v = *reinterpret_cast<const u32x2*>(a.x + (ok ? p : 0) * CI + c * 16 + 4 * q);
      xt[c] = transpose(as_s16x4(ok ? v : z2));
    }
#pragma unroll
    for (int i = 0; i < KBW; ++i) {
      const int kk = (wave + 4 * i) * 16 + 4 * q, ab = kk / CO, co = kk % CO;
      const u32x2 v = *reinterpret_cast<const u32x2*>(a.u + ct_out_pixel(a, ok ? p : 0, ab) * a.ldu + co);
      const u32x2 gv = ok ? v : z2;
      float gfl[4];
      unpack4(gv, gfl);
#pragma unroll
      for (int j = 0; j < 4; ++j) cs[i][j] += gfl[j];
      const s16x4 gt = transpose(as_s16x4(gv));
#pragma unroll
      for (int c = 0; c < CB; ++c) acc[i][c] = mfma_bf16_k16(gt, xt[c], acc[i][c]);
    }
  }
  // acc[i][c][j] at lane (r, q) = dWm[k = 16 (wave + 4 i) + 4 q + j][ci = 16 c + r]
  float* slab = a.slabs + (long long)blockIdx.x * (4 * CO * CI + 4 * CO);
#pragma unroll
  for (int i = 0; i < KBW; ++i) {
#pragma unroll
    for (int c = 0; c < CB; ++c)
#pragma unroll
      for (int j = 0; j < 4; ++j) slab[(long long)((wave + 4 * i) * 16 + 4 * q + j) * CI + c * 16 + r] = acc[i][c][j];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float v = cs[i][j];
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) v += __shfl_xor(v, o, 64);
      if (r == 0) slab[4 * CO * CI + (wave + 4 * i) * 16 + 4 * q + j] = v;
    }
  }
}

// dWm[e] (+)= sum of the slabs in order; db[co] (+)= sum over (a,b) and slabs of the column sums
__global__ void __launch_bounds__(256) u16_convt_reduce_kernel(const float* __restrict__ slabs, int nslab, int CO, int CI,
                                                               float* __restrict__ dw, float* __restrict__ db, int accumulate) {
  const long long wel = (long long)4 * CO * CI, stride = wel + 4 * CO;
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e < wel) {
    float s = 0.f;
    for (int b = 0; b < nslab; ++b) s += slabs[b * stride + e];
    dw[e] = accumulate ? dw[e] + s : s;
  } else if (e < wel + CO && db) {
    const int co = (int)(e - wel);
    float s = 0.f;
    for (int b = 0; b < nslab; ++b)
      for (int ab = 0; ab < 4; ++ab) s += slabs[b * stride + wel + ab * CO + co];
    db[co] = accumulate ? db[co] + s : s;
  }
}

static inline int u16_ct_grid(long long P, int cap) {
  long long g = (P + 15) / 16;
  if (g > cap) g = cap;
  return (int)(g < 1 ? 1 : g);
}
// weight gradient: one slab of 4 Co Ci + 4 Co floats per workgroup - at least eight 16-pixel steps per slab
static inline int u16_ct_wgrad_grid(long long P) {
  long long g = (P + 127) / 128;
  if (g > 256) g = 256;
  return (int)(g < 1 ? 1 : g);
}

}  // namespace mmft

using namespace mmft;

extern "C" {

#define U16_CT_CHECK(name)                                                                                              \
  MMFT_REQUIRE(N > 0 && h > 0 && w > 0 && (Ci == 32 || Ci == 64 || Ci == 128) && ldu >= Ci / 2 && ldu % 4 == 0, name ": Ci in {32, 64, 128}, " \
               "slice pitch >= Ci / 2");

int mmft_u16_convt_fwd(const void* x, const void* wpk, const float* bias, void* u, int ldu, int N, int h, int w, int Ci, int device,
                       void* stream) {
  MMFT_REQUIRE(x && wpk && u, "u16_convt_fwd: null pointer");
  U16_CT_CHECK("u16_convt_fwd")
  DeviceGuard dg(device);
  U16ConvTArgs a{reinterpret_cast<const u16*>(x), reinterpret_cast<const u16*>(wpk), bias, reinterpret_cast<u16*>(u), ldu, nullptr, nullptr,
                 N, h, w, (long long)N * h * w};
  const int grid = u16_ct_grid(a.P, 2048);
  const double fl = 2.0 * a.P * Ci * 2.0 * Ci, by = 2.0 * a.P * Ci * 3.0;
  if (Ci == 128) MMFT_LAUNCH("u16_convt_fwd_kernel", fl, by, u16_convt_fwd_kernel<128>, dim3(grid), dim3(256), (hipStream_t)stream, a);
  else if (Ci == 64) MMFT_LAUNCH("u16_convt_fwd_kernel", fl, by, u16_convt_fwd_kernel<64>, dim3(grid), dim3(256), (hipStream_t)stream, a);
  else MMFT_LAUNCH("u16_convt_fwd_kernel", fl, by, u16_convt_fwd_kernel<32>, dim3(grid), dim3(256), (hipStream_t)stream, a);
  return check_launch("u16_convt_fwd");
}

int mmft_u16_convt_dgrad(const void* g, int ldu, const void* wpk_t, void* dx, int N, int h, int w, int Ci, int device, void* stream) {
  MMFT_REQUIRE(g && wpk_t && dx, "u16_convt_dgrad: null pointer");
  U16_CT_CHECK("u16_convt_dgrad")
  DeviceGuard dg(device);
  U16ConvTArgs a{nullptr, reinterpret_cast<const u16*>(wpk_t), nullptr, const_cast<u16*>(reinterpret_cast<const u16*>(g)), ldu,
                 reinterpret_cast<u16*>(dx), nullptr, N, h, w, (long long)N * h * w};
  const int grid = u16_ct_grid(a.P, 2048);
  const double fl = 2.0 * a.P * Ci * 2.0 * Ci, by = 2.0 * a.P * Ci * 3.0;
  if (Ci == 128) MMFT_LAUNCH("u16_convt_dgrad_kernel", fl, by, u16_convt_dgrad_kernel<128>, dim3(grid), dim3(256), (hipStream_t)stream, a);
  else if (Ci == 64) MMFT_LAUNCH("u16_convt_dgrad_kernel", fl, by, u16_convt_dgrad_kernel<64>, dim3(grid), dim3(256), (hipStream_t)stream, a);
  else MMFT_LAUNCH("u16_convt_dgrad_kernel", fl, by, u16_convt_dgrad_kernel<32>, dim3(grid), dim3(256), (hipStream_t)stream, a);
  return check_launch("u16_convt_dgrad");
}

long long mmft_u16_convt_wgrad_workspace_bytes(int N, int h, int w, int Ci) {
  const int Co = Ci / 2;
  return (long long)u16_ct_wgrad_grid((long long)N * h * w) * (4LL * Co * Ci + 4 * Co) * 4;
}

/* slabs of 4 Co Ci + 4 Co floats: [weights in the parameter's order | column sums per (a, b)] */
int mmft_u16_convt_wgrad_slabs(int N, int h, int w) { return u16_ct_wgrad_grid((long long)N * h * w); }

/* dw: [(a,b,co)][ci] (the parameter's memory order, see Unet.Up); db: [Co] or null */
int mmft_u16_convt_wgrad(const void* x, const void* g, int ldu, float* dw, float* db, int accumulate, int N, int h, int w, int Ci,
                         float* workspace, long long workspace_bytes, int device, void* stream) {
  MMFT_REQUIRE(x && g, "u16_convt_wgrad: null pointer");
  U16_CT_CHECK("u16_convt_wgrad")
  MMFT_REQUIRE(workspace && workspace_bytes >= mmft_u16_convt_wgrad_workspace_bytes(N, h, w, Ci), "u16_convt_wgrad: workspace too small");
  DeviceGuard dg(device);
  hipStream_t st = (hipStream_t)stream;
  U16ConvTArgs a{reinterpret_cast<const u16*>(x), nullptr, nullptr, const_cast<u16*>(reinterpret_cast<const u16*>(g)), ldu, nullptr,
                 workspace, N, h, w, (long long)N * h * w};
  const int grid = u16_ct_wgrad_grid(a.P), Co = Ci / 2;
  const double fl = 2.0 * a.P * Ci * 2.0 * Ci, by = 2.0 * a.P * Ci * 3.0;
  if (Ci == 128) MMFT_LAUNCH("u16_convt_wgrad_kernel", fl, by, u16_convt_wgrad_kernel<128>, dim3(grid), dim3(256), st, a);
  else if (Ci == 64) MMFT_LAUNCH("u16_convt_wgrad_kernel", fl, by, u16_convt_wgrad_kernel<64>, dim3(grid), dim3(256), st, a);
  else MMFT_LAUNCH("u16_convt_wgrad_kernel", fl, by, u16_convt_wgrad_kernel<32>, dim3(grid), dim3(256), st, a);
  int rc = check_launch("u16_convt_wgrad");
  if (rc || !dw) return rc;          // dw == NULL: the slabs stay in `workspace` for mmft_slab_reduce_batch
  const long long el = 4LL * Co * Ci + Co;
  hipLaunchKernelGGL(u16_convt_reduce_kernel, dim3((int)((el + 255) / 256)), dim3(256), 0, st, workspace, grid, Co, Ci, dw, db,
                     accumulate ? 1 : 0);
  return check_launch("u16_convt_reduce");
}

}  // extern "C"

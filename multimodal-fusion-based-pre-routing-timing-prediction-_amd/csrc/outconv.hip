// OutConv of the layout U-Net in one kernel per direction (src/Unet.py:71-82): 1x1 convolution to ONE channel (bias),
// 2x2 pooling, ReLU.
//
// As separate operators the head of the U-Net was a GEMM with N = 1 (a 16-wide tile for one column), a pooling and an
// activation kernel forward, and five launches backward - two of them implicit GEMMs with one input / one output channel
// (50 and 63 us at 8 x 256 x 256 x 16).  The whole thing is a dot product per pixel: HBM-bound, 33 MB in, 0.5 MB out.
//   forward : out[n][y][x] = relu(pool_{2x2}(b + sum_c w[c] x[n][2y + dy][2x + dx][c]))
//   backward: g_p = dL/dout routed through the ReLU and the pooling window (torch's argmax rule: the first maximum in
//             scan order, NaN wins), dx[p][c] = g_p w[c], dw[c] = sum_p g_p x[p][c], db = sum_p g_p.
// Nothing is saved between the two: the backward recomputes the pixel values from x (it needs the argmax anyway).
// A wave covers 32 pixels of an even row (lanes 0-31) and the 32 pixels below them (lanes 32-63): every lane's load is 4 Ci
// contiguous bytes, a window's four values meet through three lane exchanges.  The arithmetic is plain fp32 in both math
// modes (the layer is 32 flops per pixel).  Weight-gradient partials: one slab per workgroup, added in a fixed order.
#include "common.h"

namespace mmft {

typedef float oc_f32x4 __attribute__((ext_vector_type(4)));
constexpr int OC_MAX_CI = 32;

struct OutConvArgs {
  const float* x;      // [N][H][W][Ci]
  const float* w;      // [Ci]
  const float* bias;   // [1] or null
  const float* gout;   // [N][H/2][W/2]       (backward)
  float* out;          // [N][H/2][W/2]       (forward)
  float* dx;           // [N][H][W][Ci]       (backward)
  float* slabs;        // [gridDim.x][Ci + 1] (backward)
  int N, H, W, mode;
  long long items;     // N * (H / 2) * (W / 32)
};

// value of this lane's pixel and the pooled value / argmax of its window; returns false for lanes outside the image
template <int CI>
__device__ __forceinline__ void oc_window(const OutConvArgs& a, long long item, int lane, float xv[CI], float& pooled,
                                          int& arg, int& me, long long& pix, long long& opix) {
  const int wx = a.W / 32;
  const int xb = (int)(item % wx);
  const long long rp = item / wx;                    // n * (H / 2) + y2
  const int y2 = (int)(rp % (a.H / 2));
  const long long n = rp / (a.H / 2);
  const int rowbit = lane >> 5, xx = xb * 32 + (lane & 31);
  pix = (n * a.H + 2 * y2 + rowbit) * a.W + xx;
  opix = rp * (a.W / 2) + (xx >> 1);
  const float* p = a.x + pix * CI;
  float v = a.bias ? a.bias[0] : 0.f;
#pragma unroll
  for (int c = 0; c < CI; c += 4) {
    const oc_f32x4 t = *reinterpret_cast<const oc_f32x4*>(p + c);
    const oc_f32x4 ww = *reinterpret_cast<const oc_f32x4*>(a.w + c);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      xv[c + j] = t[j];
      v = __fmaf_rn(t[j], ww[j], v);
    }
  }
  me = rowbit * 2 + (xx & 1);
  const float vx = __shfl_xor(v, 1, 64), vy = __shfl_xor(v, 32, 64), vd = __shfl_xor(v, 33, 64);
  // window values in scan order (0,0) (0,1) (1,0) (1,1), by selects (me differs from lane to lane)
  float q[4];
  q[0] = me == 0 ? v : me == 1 ? vx : me == 2 ? vy : vd;
  q[1] = me == 1 ? v : me == 0 ? vx : me == 3 ? vy : vd;
  q[2] = me == 2 ? v : me == 3 ? vx : me == 0 ? vy : vd;
  q[3] = me == 3 ? v : me == 2 ? vx : me == 1 ? vy : vd;
  if (a.mode == MMFT_POOL_MAX) {
    float m = q[0];
    arg = 0;
#pragma unroll
    for (int j = 1; j < 4; ++j)
      if (q[j] > m || q[j] != q[j]) {               // torch: take val if (val > max) || isnan(val)
        m = q[j];
        arg = j;
      }
    pooled = m;
  } else {
    pooled = (q[0] + q[1] + q[2] + q[3]) * 0.25f;
    arg = -1;
  }
}

template <int CI>
__global__ void __launch_bounds__(256) outconv_fwd_kernel(OutConvArgs a) {
  const int lane = threadIdx.x & 63;
  for (long long item = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); item < a.items; item += (long long)gridDim.x * 4) {
    float xv[CI], pooled;
    int arg, me;
    long long pix, opix;
    oc_window<CI>(a, item, lane, xv, pooled, arg, me, pix, opix);
    if (me == 0) a.out[opix] = pooled > 0.f ? pooled : 0.f;
  }
}

template <int CI>
__global__ void __launch_bounds__(256) outconv_bwd_kernel(OutConvArgs a) {
  __shared__ float red[4][CI + 1];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float acc[CI + 1];
#pragma unroll
  for (int c = 0; c <= CI; ++c) acc[c] = 0.f;
  for (long long item = (long long)blockIdx.x * 4 + wave; item < a.items; item += (long long)gridDim.x * 4) {
    float xv[CI], pooled;
    int arg, me;
    long long pix, opix;
    oc_window<CI>(a, item, lane, xv, pooled, arg, me, pix, opix);
    float g = a.gout[opix];
    if (!(pooled > 0.f)) g = 0.f;                                       // ReLU (mask from the output, as ActFn)
    g = a.mode == MMFT_POOL_MAX ? (me == arg ? g : 0.f) : g * 0.25f;
    float* d = a.dx + pix * CI;
#pragma unroll
    for (int c = 0; c < CI; c += 4) {
      const oc_f32x4 ww = *reinterpret_cast<const oc_f32x4*>(a.w + c);
      *reinterpret_cast<oc_f32x4*>(d + c) = ww * g;
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[c + j] = __fmaf_rn(g, xv[c + j], acc[c + j]);
    }
    acc[CI] += g;
  }
  // wave sums (butterfly: every lane ends with the total), then the four waves in order
#pragma unroll
  for (int c = 0; c <= CI; ++c) {
    float v = acc[c];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if (lane == 0) red[wave][c] = v;
  }
  __syncthreads();
  if (threadIdx.x <= CI)
    a.slabs[(long long)blockIdx.x * (CI + 1) + threadIdx.x] =
        ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
}

// dw[c] / db = sum of the slabs, fixed order: 8 partial sums per element combined through LDS
__global__ void __launch_bounds__(512) outconv_reduce_kernel(const float* __restrict__ slabs, int nslab, int CI,
                                                             float* __restrict__ dw, float* __restrict__ db, int accumulate) {
  __shared__ float part[8][64];
  const int e = threadIdx.x & 63, p = threadIdx.x >> 6;
  float s = 0.f;
  if (e <= CI)
    for (int b = p; b < nslab; b += 8) s += slabs[(long long)b * (CI + 1) + e];
  part[p][e] = s;
  __syncthreads();
  if (p == 0 && e <= CI) {
    float t = 0.f;
    for (int k = 0; k < 8; ++k) t += part[k][e];
    float* dst = e < CI ? dw + e : db;
    if (dst) *dst = accumulate ? *dst + t : t;
  }
}

static inline bool outconv_ok(int H, int W, int Ci) {
  return (Ci == 16 || Ci == 32) && H % 2 == 0 && W % 32 == 0;
}

static inline int outconv_grid(long long items) {
  long long g = (items + 3) / 4;
  if (g > 1024) g = 1024;
  return (int)(g < 1 ? 1 : g);
}

}  // namespace mmft

using namespace mmft;

extern "C" int mmft_outconv_supported(int H, int W, int Ci) { return outconv_ok(H, W, Ci) ? 1 : 0; }

extern "C" int mmft_outconv_fwd(const float* x, const float* w, const float* bias, float* out, int Nimg, int H, int W, int Ci,
                                int mode, int device, void* stream) {
  MMFT_REQUIRE(x && w && out && Nimg > 0 && (mode == MMFT_POOL_MAX || mode == MMFT_POOL_AVG), "outconv_fwd: bad arguments");
  MMFT_REQUIRE(outconv_ok(H, W, Ci), "outconv_fwd: needs Ci in {16, 32}, even H, W %% 32 == 0 (mmft_outconv_supported)");
  MMFT_REQUIRE(aligned16(x) && aligned16(w), "outconv_fwd: x / w must be 16-byte aligned");
  DeviceGuard dg(device);
  const long long items = (long long)Nimg * (H / 2) * (W / 32);
  OutConvArgs a{x, w, bias, nullptr, out, nullptr, nullptr, Nimg, H, W, mode, items};
  const double by = 4.0 * Nimg * H * W * Ci + 1.0 * Nimg * H * W;
  if (Ci == 16)
    MMFT_LAUNCH("outconv_fwd_kernel", 2.0 * Nimg * H * W * Ci, by, outconv_fwd_kernel<16>, dim3(outconv_grid(items)), dim3(256),
                (hipStream_t)stream, a);
  else
    MMFT_LAUNCH("outconv_fwd_kernel", 2.0 * Nimg * H * W * Ci, by, outconv_fwd_kernel<32>, dim3(outconv_grid(items)), dim3(256),
                (hipStream_t)stream, a);
  return check_launch("outconv_fwd");
}

extern "C" long long mmft_outconv_bwd_workspace_bytes(int Nimg, int H, int W, int Ci) {
  if (Nimg <= 0 || !outconv_ok(H, W, Ci)) return 0;
  return (long long)outconv_grid((long long)Nimg * (H / 2) * (W / 32)) * (Ci + 1) * 4;
}

extern "C" int mmft_outconv_bwd(const float* x, const float* w, const float* bias, const float* gout, float* dx, float* dw,
                                float* db, int accumulate, int Nimg, int H, int W, int Ci, int mode, float* workspace,
                                long long workspace_bytes, int device, void* stream) {
  MMFT_REQUIRE(x && w && gout && dx && dw && Nimg > 0 && (mode == MMFT_POOL_MAX || mode == MMFT_POOL_AVG),
               "outconv_bwd: bad arguments");
  MMFT_REQUIRE(outconv_ok(H, W, Ci), "outconv_bwd: needs Ci in {16, 32}, even H, W %% 32 == 0 (mmft_outconv_supported)");
  MMFT_REQUIRE(aligned16(x) && aligned16(w) && aligned16(dx), "outconv_bwd: x / w / dx must be 16-byte aligned");
  MMFT_REQUIRE(workspace && workspace_bytes >= mmft_outconv_bwd_workspace_bytes(Nimg, H, W, Ci), "outconv_bwd: workspace too small");
  DeviceGuard dg(device);
  hipStream_t st = (hipStream_t)stream;
  const long long items = (long long)Nimg * (H / 2) * (W / 32);
  const int grid = outconv_grid(items);
  OutConvArgs a{x, w, bias, gout, nullptr, dx, workspace, Nimg, H, W, mode, items};
  const double by = 8.0 * Nimg * H * W * Ci + 1.0 * Nimg * H * W;
  if (Ci == 16)
    MMFT_LAUNCH("outconv_bwd_kernel", 4.0 * Nimg * H * W * Ci, by, outconv_bwd_kernel<16>, dim3(grid), dim3(256), st, a);
  else
    MMFT_LAUNCH("outconv_bwd_kernel", 4.0 * Nimg * H * W * Ci, by, outconv_bwd_kernel<32>, dim3(grid), dim3(256), st, a);
  int rc = check_launch("outconv_bwd");
  if (rc) return rc;
  hipLaunchKernelGGL(outconv_reduce_kernel, dim3(1), dim3(512), 0, st, workspace, grid, Ci, dw, db, accumulate ? 1 : 0);
  return check_launch("outconv_reduce");
}

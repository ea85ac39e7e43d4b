// Layout-image CNN kernels: NHWC fp32 implicit-GEMM convolution (forward / dgrad / wgrad on the MFMA
// engine), train-mode BatchNorm (+ReLU) with fp64-combined statistics, 2x2 pooling, pixel shuffle for
// ConvTranspose2d(k2,s2), region copies for cat/pad, NCHW<->NHWC.
// Replaces the torch modules inside the reference's src/Unet.py:8-119 and LayoutNet (src/model.py:216-247).
#include "gemm_engine.h"
#include "conv_direct.h"
#include "conv_tile.h"
#include "conv_wgrad_narrow.h"

namespace mmft {

// ------------------------------------------------------------------ weight re-layout for dgrad
// wd[ci][kh][kw][co] = w[co][KH-1-kh][KW-1-kw][ci]
__global__ void __launch_bounds__(256) dgrad_weight_kernel(const float* __restrict__ w, float* __restrict__ wd, int Co,
                                                           int KH, int KW, int Ci) {
  int total = Co * KH * KW * Ci;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    int co = i % Co;
    int r = i / Co;
    int kw = r % KW;
    r /= KW;
    int kh = r % KH;
    int ci = r / KH;
    wd[i] = w[(((long long)co * KH + (KH - 1 - kh)) * KW + (KW - 1 - kw)) * Ci + ci];
  }
}

static int conv_fwd_launch(const float* x, const float* w, const float* bias, float* y, int Nimg, int H, int W, int Ci,
                           int Co, int KH, int KW, int pad, int act, float slope, hipStream_t st) {
  if (conv_direct_ok(x, w, bias, y, W, Ci, Co, KH, KW, pad)) {
    if (conv_tile_ok(H, W)) return conv_tile_launch(x, w, bias, y, Nimg, H, W, Ci, Co, act, slope, st);
    return conv_direct_launch(x, w, bias, y, Nimg, H, W, Ci, Co, act, slope, st);
  }
  if (conv_tile_rgb_shape(Ci, Co, KH, KW, pad) && conv_tile_ok(H, W) && aligned16(y) && (!bias || aligned16(bias)))
    return conv_tile_launch(x, w, bias, y, Nimg, H, W, Ci, Co, act, slope, st);       // the 3-channel first layer
  int M = Nimg * H * W, N = Co, K = KH * KW * Ci;
  DenseMK wl{w, nullptr, K, N, (K % 4 == 0) && aligned16(w)};
  Epi epi{y, Co, nullptr, bias, nullptr, nullptr, 0, EPI_STORE, act, slope, 0, (Co % 4 == 0) && aligned16(y)};
  if (Ci % 4 == 0 && aligned16(x)) {
    Im2colMK xl{x, H, W, Ci, KH, KW, pad, M, make_decode(H, W, Ci, KW)};
    return launch_gemm(xl, wl, epi, M, N, K, 1, st);
  }
  Im2colMKScalar xl{x, H, W, Ci, KH, KW, pad, M};
  return launch_gemm(xl, wl, epi, M, N, K, 1, st);
}

// ------------------------------------------------------------------ BatchNorm (train mode)
// stage 1: per (group, row-block) partial sums of (x - shift) and (x - shift)^2, shift = first row of
// the group (removes the mean offset from the E[x^2]-E[x]^2 cancellation); stage 2 combines in fp64.
// rows per block = ceil(rows / blocks_per_group); the host picks blocks_per_group so that a layer of any size launches
// on the order of 2048 workgroups (bn_blocks) - with a fixed 512 rows the 64x64 and 32x32 stages ran on 64 and 16.

// C % 4 == 0: a thread owns one 4-channel group (16-byte loads) and walks the rows of its block with two rows in
// flight; C % 4 != 0 falls back to one channel per thread.
template <int VEC>
__global__ void __launch_bounds__(256) bn_partial_kernel(const float* __restrict__ x, long long rows, int C,
                                                         int blocks_per_group, float* __restrict__ partial) {
  __shared__ float red[2][256 * VEC];
  int grp = blockIdx.x / blocks_per_group, blk = blockIdx.x % blocks_per_group;
  const float* xg = x + (long long)grp * rows * C;
  const long long rpb = (rows + blocks_per_group - 1) / blocks_per_group;
  long long r0 = (long long)blk * rpb;
  long long r1 = r0 + rpb < rows ? r0 + rpb : rows;
  const int cg = C / VEC;                              // channel groups per row
  int lanes_per_row = cg < 256 ? cg : 256;
  int rstep = 256 / lanes_per_row;
  int c = (threadIdx.x % lanes_per_row) * VEC, rl = threadIdx.x / lanes_per_row;
  float s[VEC], ss[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) s[j] = ss[j] = 0.f;
  if (rl < rstep) {
    float shift[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) shift[j] = xg[c + j];
#pragma unroll 4
    for (long long r = r0 + rl; r < r1; r += rstep) {
      if (VEC == 4) {
        f32x4 v = *reinterpret_cast<const f32x4*>(xg + r * C + c);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float d = v[j] - shift[j];
          s[j] += d;
          ss[j] += d * d;
        }
      } else {
        float d = xg[r * C + c] - shift[0];
        s[0] += d;
        ss[0] += d * d;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    red[0][threadIdx.x * VEC + j] = s[j];
    red[1][threadIdx.x * VEC + j] = ss[j];
  }
  __syncthreads();
  for (int cc = threadIdx.x; cc < C && cc < lanes_per_row * VEC; cc += 256) {
    float a = 0.f, bsum = 0.f;
    for (int j = 0; j < rstep; ++j) {
      a += red[0][j * lanes_per_row * VEC + cc];
      bsum += red[1][j * lanes_per_row * VEC + cc];
    }
    float* q = partial + ((long long)blockIdx.x * 2) * C;
    q[cc] = a;
    q[C + cc] = bsum;
  }
}

__device__ __forceinline__ double wave_sum(double v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// stage 2: one wave per (group, channel) combines the row-block partials in fp64
__global__ void __launch_bounds__(64) bn_finalize_kernel(const float* __restrict__ x, const float* __restrict__ partial,
                                                         long long rows, int C, int blocks_per_group, float eps,
                                                         float* __restrict__ save_mean, float* __restrict__ save_invstd,
                                                         float* __restrict__ save_var) {
  int g = blockIdx.x / C, c = blockIdx.x % C;
  double s = 0.0, ss = 0.0;
  for (int b = threadIdx.x; b < blocks_per_group; b += 64) {
    const float* q = partial + ((long long)(g * blocks_per_group + b) * 2) * C;
    s += (double)q[c];
    ss += (double)q[C + c];
  }
  s = wave_sum(s);
  ss = wave_sum(ss);
  if (threadIdx.x == 0) {
    double shift = (double)x[(long long)g * rows * C + c];
    double n = (double)rows;
    double m_sh = s / n;
    double var = ss / n - m_sh * m_sh;
    if (var < 0.0) var = 0.0;
    save_mean[g * C + c] = (float)(m_sh + shift);
    save_invstd[g * C + c] = (float)(1.0 / sqrt(var + (double)eps));
    save_var[g * C + c] = (float)var;
  }
}

// running statistics: `groups` sequential momentum updates per channel (one per image when per-sample).
// Runs in block 0 of the apply kernel (C <= 256 = its block size): one launch less per layer.
struct BnRunning {
  const float* save_var;
  float* running_mean;
  float* running_var;
  float momentum;
  int groups;
};

__device__ __forceinline__ void bn_running_update(const float* __restrict__ save_mean, const float* __restrict__ save_var,
                                                  int groups, long long rows, int C, float momentum,
                                                  float* __restrict__ running_mean, float* __restrict__ running_var) {
  int c = threadIdx.x;
  if (c >= C) return;
  double rm = (double)running_mean[c], rv = (double)running_var[c], n = (double)rows;
  for (int g = 0; g < groups; ++g) {
    double var = (double)save_var[g * C + c];
    double unbiased = rows > 1 ? var * n / (n - 1.0) : var;
    rm = (1.0 - (double)momentum) * rm + (double)momentum * (double)save_mean[g * C + c];
    rv = (1.0 - (double)momentum) * rv + (double)momentum * unbiased;
  }
  running_mean[c] = (float)rm;
  running_var[c] = (float)rv;
}

// y = (x - mean) * invstd * gamma + beta with a fixed rounding sequence: the backward kernels recompute the ReLU mask
// from x with the SAME function instead of reading y back (two of the seven tensor passes of the BatchNorm backward)
__device__ __forceinline__ float bn_affine(float x, float mean, float invstd, float gamma, float beta) {
  return __fmaf_rn(__fmul_rn(__fsub_rn(x, mean), invstd), gamma, beta);
}

// blockIdx.y = group.  The grid stride (gridDim.x * 256 * VEC elements) is a multiple of C whenever C divides 1024 - every
// layer of the U-Net - so a thread keeps ONE channel group for the whole launch: its four statistics / parameters are
// loaded once instead of per element (the per-element form spent more issue slots on index arithmetic and parameter loads
// than on the tensor: 2.5 TB/s).
template <int VEC>
__global__ void __launch_bounds__(256) bn_apply_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const float* __restrict__ mean, const float* __restrict__ invstd,
                                                       long long rows, int C, long long total, int relu, BnRunning run) {
  if (run.running_mean && blockIdx.x == 0 && blockIdx.y == 0)
    bn_running_update(mean, run.save_var, run.groups, rows, C, run.momentum, run.running_mean, run.running_var);
  const int g = blockIdx.y;
  const long long per_group = rows * C;
  const float* xg = x + (long long)g * per_group;
  float* yg = y + (long long)g * per_group;
  const long long stride = (long long)gridDim.x * blockDim.x * VEC;
  long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * VEC;
  if (VEC == 4 && stride % C == 0) {
    const int c = (int)(i % C);
    float mu[4], is[4], ga[4], be[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      mu[j] = mean[g * C + c + j]; is[j] = invstd[g * C + c + j]; ga[j] = gamma[c + j]; be[j] = beta[c + j];
    }
    for (; i + stride < per_group; i += 2 * stride) {            // two independent 16-byte streams per thread
      f32x4 v0 = *reinterpret_cast<const f32x4*>(xg + i), v1 = *reinterpret_cast<const f32x4*>(xg + i + stride);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float t0 = bn_affine(v0[j], mu[j], is[j], ga[j], be[j]), t1 = bn_affine(v1[j], mu[j], is[j], ga[j], be[j]);
        v0[j] = (relu && !(t0 > 0.f)) ? 0.f : t0;
        v1[j] = (relu && !(t1 > 0.f)) ? 0.f : t1;
      }
      *reinterpret_cast<f32x4*>(yg + i) = v0;
      *reinterpret_cast<f32x4*>(yg + i + stride) = v1;
    }
    for (; i < per_group; i += stride) {
      f32x4 v = *reinterpret_cast<const f32x4*>(xg + i);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float t = bn_affine(v[j], mu[j], is[j], ga[j], be[j]);
        v[j] = (relu && !(t > 0.f)) ? 0.f : t;
      }
      *reinterpret_cast<f32x4*>(yg + i) = v;
    }
    return;
  }
  for (; i < per_group; i += stride) {
    int c = (int)(i % C);
    if (VEC == 4) {
      f32x4 v = *reinterpret_cast<const f32x4*>(xg + i);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float t = bn_affine(v[j], mean[g * C + c + j], invstd[g * C + c + j], gamma[c + j], beta[c + j]);
        v[j] = (relu && !(t > 0.f)) ? 0.f : t;
      }
      *reinterpret_cast<f32x4*>(yg + i) = v;
    } else {
      float t = bn_affine(xg[i], mean[g * C + c], invstd[g * C + c], gamma[c], beta[c]);
      yg[i] = (relu && !(t > 0.f)) ? 0.f : t;
    }
  }
}

// backward stage 1: partial sums of g and g*xhat (g = gy masked by the ReLU output)
template <int VEC>
__global__ void __launch_bounds__(256) bn_bwd_partial_kernel(const float* __restrict__ gy, const float* __restrict__ x,
                                                             const float* __restrict__ y, const float* __restrict__ mean,
                                                             const float* __restrict__ invstd, long long rows, int C,
                                                             int blocks_per_group, int relu,
                                                             float* __restrict__ partial, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta) {
  __shared__ float red[2][256 * VEC];
  int grp = blockIdx.x / blocks_per_group, blk = blockIdx.x % blocks_per_group;
  long long base = (long long)grp * rows * C;
  const long long rpb = (rows + blocks_per_group - 1) / blocks_per_group;
  long long r0 = (long long)blk * rpb;
  long long r1 = r0 + rpb < rows ? r0 + rpb : rows;
  const int cg = C / VEC;
  int lanes_per_row = cg < 256 ? cg : 256;
  int rstep = 256 / lanes_per_row;
  int c = (threadIdx.x % lanes_per_row) * VEC, rl = threadIdx.x / lanes_per_row;
  float s[VEC], sx[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) s[j] = sx[j] = 0.f;
  if (rl < rstep) {
    float mu[VEC], is[VEC], ga[VEC], be[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      mu[j] = mean[grp * C + c + j];
      is[j] = invstd[grp * C + c + j];
      ga[j] = beta ? gamma[c + j] : 0.f;
      be[j] = beta ? beta[c + j] : 0.f;
    }
#pragma unroll 4
    for (long long r = r0 + rl; r < r1; r += rstep) {
      long long i = base + r * C + c;
      if (VEC == 4) {
        f32x4 g = *reinterpret_cast<const f32x4*>(gy + i);
        f32x4 xv = *reinterpret_cast<const f32x4*>(x + i);
        if (relu && beta) {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (!(bn_affine(xv[j], mu[j], is[j], ga[j], be[j]) > 0.f)) g[j] = 0.f;
        } else if (relu) {
          f32x4 yv = *reinterpret_cast<const f32x4*>(y + i);
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (!(yv[j] > 0.f)) g[j] = 0.f;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          s[j] += g[j];
          sx[j] += g[j] * (xv[j] - mu[j]) * is[j];
        }
      } else {
        float g = gy[i];
        if (relu && !((beta ? bn_affine(x[i], mu[0], is[0], ga[0], be[0]) : y[i]) > 0.f)) g = 0.f;
        s[0] += g;
        sx[0] += g * (x[i] - mu[0]) * is[0];
      }
    }
  }
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    red[0][threadIdx.x * VEC + j] = s[j];
    red[1][threadIdx.x * VEC + j] = sx[j];
  }
  __syncthreads();
  for (int cc = threadIdx.x; cc < C && cc < lanes_per_row * VEC; cc += 256) {
    float a = 0.f, bsum = 0.f;
    for (int j = 0; j < rstep; ++j) {
      a += red[0][j * lanes_per_row * VEC + cc];
      bsum += red[1][j * lanes_per_row * VEC + cc];
    }
    float* q = partial + ((long long)blockIdx.x * 2) * C;
    q[cc] = a;
    q[C + cc] = bsum;
  }
}

// stage 2: one wave per (group, channel): per-group means coef[g][0..1][C]; then dgamma/dbeta over groups
__global__ void __launch_bounds__(64) bn_bwd_finalize_kernel(const float* __restrict__ partial, long long rows, int C,
                                                             int blocks_per_group, float* __restrict__ coef) {
  int g = blockIdx.x / C, c = blockIdx.x % C;
  double s = 0.0, sx = 0.0;
  for (int b = threadIdx.x; b < blocks_per_group; b += 64) {
    const float* q = partial + ((long long)(g * blocks_per_group + b) * 2) * C;
    s += (double)q[c];
    sx += (double)q[C + c];
  }
  s = wave_sum(s);
  sx = wave_sum(sx);
  if (threadIdx.x == 0) {
    coef[(g * 2) * C + c] = (float)(s / (double)rows);
    coef[(g * 2 + 1) * C + c] = (float)(sx / (double)rows);
  }
}

__device__ __forceinline__ void bn_bwd_params(const float* __restrict__ coef, int groups, long long rows, int C,
                                              float* __restrict__ dgamma, float* __restrict__ dbeta) {
  int c = threadIdx.x;
  if (c >= C) return;
  double dg = 0.0, db = 0.0;
  for (int g = 0; g < groups; ++g) {
    db += (double)coef[(g * 2) * C + c];
    dg += (double)coef[(g * 2 + 1) * C + c];
  }
  dgamma[c] = (float)(dg * (double)rows);
  dbeta[c] = (float)(db * (double)rows);
}

template <int VEC>
__global__ void __launch_bounds__(256) bn_bwd_apply_kernel(const float* __restrict__ gy, const float* __restrict__ x,
                                                           const float* __restrict__ y, const float* __restrict__ gamma,
                                                           const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           const float* __restrict__ coef, float* __restrict__ dx,
                                                           long long rows, int C, long long total, int relu, int groups,
                                                           float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                           const float* __restrict__ beta) {
  if (blockIdx.x == 0 && blockIdx.y == 0) bn_bwd_params(coef, groups, rows, C, dgamma, dbeta);
  // blockIdx.y = group; a thread keeps one channel group for the whole launch when the grid stride is a multiple of C
  // (see bn_apply_kernel): its statistics, parameters and the two coefficients are loaded once
  const int g = blockIdx.y;
  const long long per_group = rows * C, base = (long long)g * per_group;
  const long long stride = (long long)gridDim.x * blockDim.x * VEC;
  long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * VEC;
  if (VEC == 4 && relu && beta && stride % C == 0) {
    const int c = (int)(i % C);
    float mu[4], is[4], ga[4], be[4], c0[4], c1[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      mu[j] = mean[g * C + c + j]; is[j] = invstd[g * C + c + j]; ga[j] = gamma[c + j]; be[j] = beta[c + j];
      c0[j] = coef[(g * 2) * C + c + j]; c1[j] = coef[(g * 2 + 1) * C + c + j];
    }
    for (; i < per_group; i += stride) {
      f32x4 gg = *reinterpret_cast<const f32x4*>(gy + base + i);
      const f32x4 xv = *reinterpret_cast<const f32x4*>(x + base + i);
      f32x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (!(bn_affine(xv[j], mu[j], is[j], ga[j], be[j]) > 0.f)) gg[j] = 0.f;
        const float xh = (xv[j] - mu[j]) * is[j];
        o[j] = ga[j] * is[j] * (gg[j] - c0[j] - xh * c1[j]);
      }
      *reinterpret_cast<f32x4*>(dx + base + i) = o;
    }
    return;
  }
  for (; i < per_group; i += stride) {
    int c = (int)(i % C);
    const long long e = base + i;
    if (VEC == 4) {
      f32x4 gg = *reinterpret_cast<const f32x4*>(gy + e);
      f32x4 xv = *reinterpret_cast<const f32x4*>(x + e);
      if (relu && beta) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (!(bn_affine(xv[j], mean[g * C + c + j], invstd[g * C + c + j], gamma[c + j], beta[c + j]) > 0.f)) gg[j] = 0.f;
      } else if (relu) {
        f32x4 yv = *reinterpret_cast<const f32x4*>(y + e);
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (!(yv[j] > 0.f)) gg[j] = 0.f;
      }
      f32x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float is = invstd[g * C + c + j];
        float xh = (xv[j] - mean[g * C + c + j]) * is;
        o[j] = gamma[c + j] * is * (gg[j] - coef[(g * 2) * C + c + j] - xh * coef[(g * 2 + 1) * C + c + j]);
      }
      *reinterpret_cast<f32x4*>(dx + e) = o;
    } else {
      float gg = gy[e];
      if (relu && !((beta ? bn_affine(x[e], mean[g * C + c], invstd[g * C + c], gamma[c], beta[c]) : y[e]) > 0.f)) gg = 0.f;
      float is = invstd[g * C + c];
      float xh = (x[e] - mean[g * C + c]) * is;
      dx[e] = gamma[c] * is * (gg - coef[(g * 2) * C + c] - xh * coef[(g * 2 + 1) * C + c]);
    }
  }
}

// ------------------------------------------------------------------ bilinear x2 up-sampling, align_corners=True
// nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True) of Up(bilinear=True) (src/Unet.py:48-51), with ATen's
// arithmetic: scale = (in - 1) / (out - 1) (0 when out == 1), src = scale * dst, i0 = (int)src, i1 = i0 + (i0 < in - 1),
// lambda1 = src - i0, lambda0 = 1 - lambda1.
__device__ __forceinline__ void up2_coord(int o, int in, float scale, int& i0, int& i1, float& l0, float& l1) {
  const float src = scale * (float)o;
  i0 = (int)src;
  i1 = i0 + (i0 < in - 1 ? 1 : 0);
  l1 = src - (float)i0;
  l0 = 1.0f - l1;
}

__global__ void __launch_bounds__(256) upsample2x_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int Nimg,
                                                             int H, int W, int C, float sy, float sx) {
  const int Ho = 2 * H, Wo = 2 * W;
  long long total = (long long)Nimg * Ho * Wo * C;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    int c = (int)(i % C);
    long long p = i / C;
    int xo = (int)(p % Wo);
    p /= Wo;
    int yo = (int)(p % Ho);
    int n = (int)(p / Ho);
    int y0, y1, x0, x1;
    float ly0, ly1, lx0, lx1;
    up2_coord(yo, H, sy, y0, y1, ly0, ly1);
    up2_coord(xo, W, sx, x0, x1, lx0, lx1);
    const float* b = x + (long long)n * H * W * C + c;
    float v00 = b[((long long)y0 * W + x0) * C], v01 = b[((long long)y0 * W + x1) * C];
    float v10 = b[((long long)y1 * W + x0) * C], v11 = b[((long long)y1 * W + x1) * C];
    y[i] = ly0 * (lx0 * v00 + lx1 * v01) + ly1 * (lx0 * v10 + lx1 * v11);
  }
}

// gather form of the backward (no atomics): input pixel (yy, xx) collects, over the few output rows / columns whose
// source interval touches it, the weight the forward gave it
__global__ void __launch_bounds__(256) upsample2x_bwd_kernel(const float* __restrict__ gy, float* __restrict__ dx, int Nimg,
                                                             int H, int W, int C, float sy, float sx) {
  const int Ho = 2 * H, Wo = 2 * W;
  long long total = (long long)Nimg * H * W * C;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    int c = (int)(i % C);
    long long p = i / C;
    int xx = (int)(p % W);
    p /= W;
    int yy = (int)(p % H);
    int n = (int)(p / H);
    // candidate outputs: src in (yy - 1, yy + 1)  <=>  o in ((yy - 1) / s, (yy + 1) / s); widened by one on both sides
    int oy_lo = sy > 0.f ? (int)floorf((float)(yy - 1) / sy) - 1 : 0, oy_hi = sy > 0.f ? (int)ceilf((float)(yy + 1) / sy) + 1 : Ho - 1;
    int ox_lo = sx > 0.f ? (int)floorf((float)(xx - 1) / sx) - 1 : 0, ox_hi = sx > 0.f ? (int)ceilf((float)(xx + 1) / sx) + 1 : Wo - 1;
    oy_lo = oy_lo < 0 ? 0 : oy_lo; ox_lo = ox_lo < 0 ? 0 : ox_lo;
    oy_hi = oy_hi > Ho - 1 ? Ho - 1 : oy_hi; ox_hi = ox_hi > Wo - 1 ? Wo - 1 : ox_hi;
    float acc = 0.f;
    const float* g = gy + (long long)n * Ho * Wo * C + c;
    for (int oy = oy_lo; oy <= oy_hi; ++oy) {
      int y0, y1;
      float ly0, ly1;
      up2_coord(oy, H, sy, y0, y1, ly0, ly1);
      float wy = (y0 == yy ? ly0 : 0.f) + (y1 == yy ? ly1 : 0.f);
      if (wy == 0.f && y0 != yy && y1 != yy) continue;
      for (int ox = ox_lo; ox <= ox_hi; ++ox) {
        int x0, x1;
        float lx0, lx1;
        up2_coord(ox, W, sx, x0, x1, lx0, lx1);
        if (x0 != xx && x1 != xx) continue;
        float wx = (x0 == xx ? lx0 : 0.f) + (x1 == xx ? lx1 : 0.f);
        acc += wy * wx * g[((long long)oy * Wo + ox) * C];
      }
    }
    dx[i] = acc;
  }
}

// ------------------------------------------------------------------ pooling
__global__ void __launch_bounds__(256) pool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int Nimg, int H,
                                                       int W, int C, int mode) {
  int Ho = H / 2, Wo = W / 2;
  long long total = (long long)Nimg * Ho * Wo * C;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    int c = (int)(i % C);
    long long p = i / C;
    int xo = (int)(p % Wo);
    p /= Wo;
    int yo = (int)(p % Ho);
    int n = (int)(p / Ho);
    const float* q = x + (((long long)n * H + 2 * yo) * W + 2 * xo) * C + c;
    float a = q[0], b = q[C], d = q[(long long)W * C], e = q[(long long)W * C + C];
    float r;
    if (mode == MMFT_POOL_MAX) {
      r = a;
      if (b > r || b != b) r = b;      // torch: take val if (val > max) || isnan(val)
      if (d > r || d != d) r = d;
      if (e > r || e != e) r = e;
    } else {
      r = (a + b + d + e) * 0.25f;
    }
    y[i] = r;
  }
}

__global__ void __launch_bounds__(256) pool_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                       float* __restrict__ dx, int Nimg, int H, int W, int C, int mode) {
  int Ho = H / 2, Wo = W / 2;
  long long total = (long long)Nimg * H * W * C;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    int c = (int)(i % C);
    long long p = i / C;
    int xx = (int)(p % W);
    p /= W;
    int yy = (int)(p % H);
    int n = (int)(p / H);
    int yo = yy >> 1, xo = xx >> 1;
    float r = 0.f;
    if (yo < Ho && xo < Wo) {
      float g = gy[(((long long)n * Ho + yo) * Wo + xo) * C + c];
      if (mode == MMFT_POOL_MAX) {
        const float* q = x + (((long long)n * H + 2 * yo) * W + 2 * xo) * C + c;
        float v[4] = {q[0], q[C], q[(long long)W * C], q[(long long)W * C + C]};
        int arg = 0;
        float m = v[0];
#pragma unroll
        for (int j = 1; j < 4; ++j)
          if (v[j] > m || v[j] != v[j]) {
            m = v[j];
            arg = j;
          }
        int me = (yy & 1) * 2 + (xx & 1);
        r = (me == arg) ? g : 0.f;
      } else {
        r = g * 0.25f;
      }
    }
    dx[i] = r;
  }
}

// ------------------------------------------------------------------ pixel shuffle / region copy / layout
__global__ void __launch_bounds__(256) pixel_shuffle_kernel(const float* __restrict__ in, const float* __restrict__ bias,
                                                            float* __restrict__ out, int Nimg, int H, int W, int Co,
                                                            int reverse, int ldc, int c_off) {
  // forward:  out[n][2y+a][2x+b][c_off + co] = in[n][y][x][(a*2+b)*Co+co] + bias[co]     (big tensor: ldc channels)
  // reverse:  out[n][y][x][(a*2+b)*Co+co] = in[n][2y+a][2x+b][c_off + co]
  long long total = (long long)Nimg * H * W * 4 * Co;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    int co = (int)(i % Co);
    long long p = i / Co;
    int ab = (int)(p % 4);
    p /= 4;
    int x = (int)(p % W);
    p /= W;
    int y = (int)(p % H);
    int n = (int)(p / H);
    long long big = (((long long)n * 2 * H + 2 * y + (ab >> 1)) * 2 * W + 2 * x + (ab & 1)) * ldc + c_off + co;
    if (!reverse) out[big] = in[i] + (bias ? bias[co] : 0.f);
    else out[i] = in[big];
  }
}

__global__ void __launch_bounds__(256) copy_region_kernel(float* __restrict__ src, int Nimg, int Hs, int Ws, int Cs,
                                                          float* __restrict__ dst, int Hd, int Wd, int Cd, int c_off,
                                                          int y_off, int x_off, int reverse) {
  long long total = (long long)Nimg * Hs * Ws * Cs;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    int c = (int)(i % Cs);
    long long p = i / Cs;
    int x = (int)(p % Ws);
    p /= Ws;
    int y = (int)(p % Hs);
    int n = (int)(p / Hs);
    long long j = (((long long)n * Hd + y + y_off) * Wd + x + x_off) * Cd + c_off + c;
    if (!reverse) dst[j] = src[i];
    else src[i] = dst[j];
  }
}

__global__ void __launch_bounds__(256) nchw_nhwc_kernel(const float* __restrict__ src, float* __restrict__ dst, int Nimg,
                                                        int C, int H, int W, int Cpad, int to_nhwc) {
  long long total = (long long)Nimg * H * W * Cpad;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    int c = (int)(i % Cpad);
    long long p = i / Cpad;
    int x = (int)(p % W);
    p /= W;
    int y = (int)(p % H);
    int n = (int)(p / H);
    long long j = (((long long)n * C + c) * H + y) * W + x;
    if (to_nhwc) dst[i] = c < C ? src[j] : 0.f;
    else if (c < C) dst[j] = src[i];
  }
}

static inline int bn_blocks(int groups, long long rows) {
  if ((long long)groups * rows < 16384) return cdiv(rows, 512);     // small layers: one or a few blocks per group
  long long rpb = 64;                                  // at least 64 rows per block, at most 512
  while (rpb < 512 && (long long)groups * cdiv(rows, rpb) > 2048) rpb *= 2;
  return cdiv(rows, rpb);
}

}  // namespace mmft

using namespace mmft;

#define CONV_CHECK(name)                                                                                    \
  MMFT_REQUIRE(Nimg > 0 && H > 0 && W > 0 && Ci > 0 && Co > 0 && KH > 0 && KW > 0, name ": bad sizes");     \
  MMFT_REQUIRE(2 * pad == KH - 1 && KH == KW, name ": only stride-1 'same' convolutions (2*pad == K-1)");   \
  MMFT_REQUIRE((long long)Nimg* H* W < (1ll << 31) && (long long)KH * KW * Ci < (1ll << 31), name ": too large")

extern "C" {

int mmft_conv2d_fwd(const float* x, const float* w, const float* bias, float* y, int Nimg, int H, int W, int Ci, int Co,
                    int KH, int KW, int pad, int act, float slope, int device, void* stream) {
  MMFT_REQUIRE(x && w && y, "conv2d_fwd: null pointer");
  CONV_CHECK("conv2d_fwd");
  DeviceGuard dg(device);
  return conv_fwd_launch(x, w, bias, y, Nimg, H, W, Ci, Co, KH, KW, pad, act, slope, (hipStream_t)stream);
}

int mmft_conv2d_dgrad(const float* dy, const float* w, float* dx, int Nimg, int H, int W, int Ci, int Co, int KH, int KW,
                      int pad, float* workspace, long long workspace_bytes, int device, void* stream) {
  MMFT_REQUIRE(dy && w && dx, "conv2d_dgrad: null pointer");
  CONV_CHECK("conv2d_dgrad");
  long long need = (long long)Co * KH * KW * Ci * 4;
  MMFT_REQUIRE(workspace && workspace_bytes >= need, "conv2d_dgrad: workspace too small");
  DeviceGuard dg(device);
  hipStream_t st = (hipStream_t)stream;
  if (conv_direct_ok(dy, w, nullptr, dx, W, Co, Ci, KH, KW, pad)) {    // narrow layers: the kernel flips w while staging it
    if (conv_tile_ok(H, W)) return conv_tile_launch(dy, w, nullptr, dx, Nimg, H, W, Co, Ci, ACT_NONE, 0.f, st, 1);
    return conv_direct_launch(dy, w, nullptr, dx, Nimg, H, W, Co, Ci, ACT_NONE, 0.f, st, 1);
  }
  hipLaunchKernelGGL(dgrad_weight_kernel, dim3(ew_grid(need / 4)), dim3(256), 0, st, w, workspace, Co, KH, KW, Ci);
  int rc = check_launch("dgrad_weight");
  if (rc) return rc;
  // dx[p][ci] = sum_{tap,co} dy[p+tap-pad][co] * wd[ci][tap][co]  : a forward conv with Ci<->Co swapped
  return conv_fwd_launch(dy, workspace, nullptr, dx, Nimg, H, W, Co, Ci, KH, KW, pad, ACT_NONE, 0.f, st);
}

static int conv_wgrad_splits(int Nimg, int H, int W, int Ci, int Co, int KH, int KW) {
  int N = KH * KW * Ci;
  long long tiles = (long long)cdiv(Co, 128) * cdiv(N, N <= 16 ? 16 : N <= 32 ? 32 : N <= 64 ? 64 : 128);
  long long K = (long long)Nimg * H * W;
  int want = (int)(768 / (tiles > 0 ? tiles : 1));
  long long maxs = K / 128;
  if (want > maxs) want = (int)maxs;
  if (want < 1) want = 1;
  if (want > 512) want = 512;
  return effective_splits((int)K, want);
}

long long mmft_conv2d_wgrad_workspace_bytes(int Nimg, int H, int W, int Ci, int Co, int KH, int KW) {
  int s = conv_wgrad_splits(Nimg, H, W, Ci, Co, KH, KW);
  long long need = s > 1 ? (long long)s * Co * KH * KW * Ci * 4 : 0;
  if (conv_wgrad_narrow_shape_ok<WGN_TH>(H, W, Ci, Co, KH, KW, (KH - 1) / 2)) {      // either math mode may be active later
    long long narrow = (long long)conv_wgrad_narrow_grid(Nimg, H, W, Ci) * Co * 9 * Ci * 4;
    if (narrow > need) need = narrow;
  }
  return need;
}

int mmft_conv2d_wgrad(const float* x, const float* dy, float* dw, int Nimg, int H, int W, int Ci, int Co, int KH, int KW,
                      int pad, float* workspace, long long workspace_bytes, int device, void* stream) {
  MMFT_REQUIRE(x && dy && dw, "conv2d_wgrad: null pointer");
  CONV_CHECK("conv2d_wgrad");
  DeviceGuard dg(device);
  hipStream_t st = (hipStream_t)stream;
  int M = Co, N = KH * KW * Ci, K = Nimg * H * W;
  if (conv_wgrad_narrow_ok(H, W, Ci, Co, KH, KW, pad) && aligned16(x) && aligned16(dy) && aligned16(dw)) {
    const int grid = conv_wgrad_narrow_grid(Nimg, H, W, Ci);
    MMFT_REQUIRE(workspace && workspace_bytes >= (long long)grid * M * N * 4, "conv2d_wgrad: workspace too small");
    int rc = conv_wgrad_narrow_launch(x, dy, workspace, Nimg, H, W, Ci, Co, grid, st);
    if (rc) return rc;
    return launch_slab_reduce(workspace, grid, (long long)M * N, dw, 0, st);
  }
  int splits = conv_wgrad_splits(Nimg, H, W, Ci, Co, KH, KW);
  long long need = splits > 1 ? (long long)splits * M * N * 4 : 0;
  MMFT_REQUIRE(splits <= 1 || (workspace && workspace_bytes >= need), "conv2d_wgrad: workspace too small");
  DenseKM xl{dy, nullptr, Co, Co, (Co % 4 == 0) && aligned16(dy)};
  float* outp = splits > 1 ? workspace : dw;
  Epi epi{outp, N, nullptr, nullptr, nullptr, nullptr, 0, EPI_STORE, ACT_NONE, 0.f, (long long)M * N,
          (N % 4 == 0) && aligned16(outp)};
  int rc;
  if (Ci % 4 == 0 && aligned16(x)) {
    Im2colKM wl{x, H, W, Ci, KH, KW, pad, N, make_decode(H, W, Ci, KW)};
    rc = launch_gemm(xl, wl, epi, M, N, K, splits, st);
  } else {
    Im2colKMScalar wl{x, H, W, Ci, KH, KW, pad, N};
    rc = launch_gemm(xl, wl, epi, M, N, K, splits, st);
  }
  if (rc || splits <= 1) return rc;
  return launch_slab_reduce(workspace, splits, (long long)M * N, dw, 0, st);
}

long long mmft_bn_workspace_bytes(int groups, long long rows, int C) {
  return ((long long)groups * bn_blocks(groups, rows) * 2 * C + (long long)groups * 2 * C) * 4;
}

int mmft_bn_train_fwd(const float* x, float* y, const float* gamma, const float* beta, float* running_mean,
                      float* running_var, float momentum, float eps, int groups, long long rows, int C, float* save_mean,
                      float* save_invstd, int relu, float* workspace, long long workspace_bytes, int device,
                      void* stream) {
  MMFT_REQUIRE(x && y && gamma && beta && save_mean && save_invstd, "bn_train_fwd: null pointer");
  MMFT_REQUIRE(groups > 0 && rows > 0 && C > 0 && C <= 256, "bn_train_fwd: bad sizes (C <= 256 supported)");
  MMFT_REQUIRE(workspace && workspace_bytes >= mmft_bn_workspace_bytes(groups, rows, C), "bn_train_fwd: workspace too small");
  DeviceGuard dg(device);
  hipStream_t st = (hipStream_t)stream;
  int bpg = bn_blocks(groups, rows);
  ProfScope ps("bn_train_fwd(3 kernels)", 0.0, 3.0 * 4.0 * groups * rows * C, st);
  const bool v4 = (C % 4 == 0) && aligned16(x) && aligned16(y);
  if (v4) hipLaunchKernelGGL(bn_partial_kernel<4>, dim3(groups * bpg), dim3(256), 0, st, x, rows, C, bpg, workspace);
  else hipLaunchKernelGGL(bn_partial_kernel<1>, dim3(groups * bpg), dim3(256), 0, st, x, rows, C, bpg, workspace);
  float* save_var = workspace + (long long)groups * bpg * 2 * C;
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(groups * C), dim3(64), 0, st, x, workspace, rows, C, bpg, eps, save_mean,
                     save_invstd, save_var);
  BnRunning run{save_var, (running_mean && running_var) ? running_mean : nullptr, running_var, momentum, groups};
  long long total = (long long)groups * rows * C;
  // blockIdx.y = group; ~2048 workgroups over all groups, two 16-byte streams per thread and iteration
  auto gx = [&](long long items) { long long b = cdiv(cdiv(items, 256 * 2), 1); long long cap = cdiv(2048, groups); return (unsigned)(b < 1 ? 1 : b > cap ? cap : b); };
  if (v4)
    hipLaunchKernelGGL(bn_apply_kernel<4>, dim3(gx(rows * C / 4), groups), dim3(256), 0, st, x, y, gamma, beta, save_mean,
                       save_invstd, rows, C, total, relu, run);
  else
    hipLaunchKernelGGL(bn_apply_kernel<1>, dim3(gx(rows * C), groups), dim3(256), 0, st, x, y, gamma, beta, save_mean,
                       save_invstd, rows, C, total, relu, run);
  return check_launch("bn_train_fwd");
}

int mmft_bn_train_bwd(const float* gy, const float* x, const float* y, const float* gamma, const float* beta,
                      const float* save_mean, const float* save_invstd, float* dx, float* dgamma, float* dbeta, int groups,
                      long long rows, int C, int relu, float* workspace, long long workspace_bytes, int device,
                      void* stream) {
  MMFT_REQUIRE(gy && x && gamma && save_mean && save_invstd && dx && dgamma && dbeta, "bn_train_bwd: null pointer");
  MMFT_REQUIRE(!relu || y || beta, "bn_train_bwd: relu backward needs beta (mask recomputed from x) or the forward output");
  MMFT_REQUIRE(groups > 0 && rows > 0 && C > 0 && C <= 256, "bn_train_bwd: bad sizes (C <= 256 supported)");
  MMFT_REQUIRE(workspace && workspace_bytes >= mmft_bn_workspace_bytes(groups, rows, C), "bn_train_bwd: workspace too small");
  DeviceGuard dg(device);
  hipStream_t st = (hipStream_t)stream;
  int bpg = bn_blocks(groups, rows);
  float* coef = workspace + (long long)groups * bpg * 2 * C;
  ProfScope ps("bn_train_bwd(3 kernels)", 0.0, (beta ? 5.0 : 7.0) * 4.0 * groups * rows * C, st);
  const bool v4 = (C % 4 == 0) && aligned16(x) && aligned16(gy) && aligned16(dx) && (!y || aligned16(y));
  if (v4)
    hipLaunchKernelGGL(bn_bwd_partial_kernel<4>, dim3(groups * bpg), dim3(256), 0, st, gy, x, y, save_mean, save_invstd, rows,
                       C, bpg, relu, workspace, gamma, beta);
  else
    hipLaunchKernelGGL(bn_bwd_partial_kernel<1>, dim3(groups * bpg), dim3(256), 0, st, gy, x, y, save_mean, save_invstd, rows,
                       C, bpg, relu, workspace, gamma, beta);
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(groups * C), dim3(64), 0, st, workspace, rows, C, bpg, coef);
  long long total = (long long)groups * rows * C;
  auto gx = [&](long long items) { long long b = cdiv(items, 256); long long cap = cdiv(2048, groups); return (unsigned)(b < 1 ? 1 : b > cap ? cap : b); };
  if (v4)
    hipLaunchKernelGGL(bn_bwd_apply_kernel<4>, dim3(gx(rows * C / 4), groups), dim3(256), 0, st, gy, x, y, gamma, save_mean,
                       save_invstd, coef, dx, rows, C, total, relu, groups, dgamma, dbeta, beta);
  else
    hipLaunchKernelGGL(bn_bwd_apply_kernel<1>, dim3(gx(rows * C), groups), dim3(256), 0, st, gy, x, y, gamma, save_mean,
                       save_invstd, coef, dx, rows, C, total, relu, groups, dgamma, dbeta, beta);
  return check_launch("bn_train_bwd");
}

static inline float up2_scale(int in) { return 2 * in > 1 ? (float)(in - 1) / (float)(2 * in - 1) : 0.f; }

int mmft_upsample_bilinear2x_fwd(const float* x, float* y, int Nimg, int H, int W, int C, int device, void* stream) {
  MMFT_REQUIRE(x && y && Nimg > 0 && H > 0 && W > 0 && C > 0, "upsample_bilinear2x_fwd: bad args");
  DeviceGuard dg(device);
  long long total = (long long)Nimg * 4 * H * W * C;
  MMFT_LAUNCH("upsample2x_fwd_kernel", 0.0, 4.0 * total * 1.25, upsample2x_fwd_kernel, dim3(ew_grid(total)), dim3(256),
              (hipStream_t)stream, x, y, Nimg, H, W, C, up2_scale(H), up2_scale(W));
  return check_launch("upsample_bilinear2x_fwd");
}

int mmft_upsample_bilinear2x_bwd(const float* gy, float* dx, int Nimg, int H, int W, int C, int device, void* stream) {
  MMFT_REQUIRE(gy && dx && Nimg > 0 && H > 0 && W > 0 && C > 0, "upsample_bilinear2x_bwd: bad args");
  DeviceGuard dg(device);
  long long total = (long long)Nimg * H * W * C;
  MMFT_LAUNCH("upsample2x_bwd_kernel", 0.0, 4.0 * total * 5.0, upsample2x_bwd_kernel, dim3(ew_grid(total)), dim3(256),
              (hipStream_t)stream, gy, dx, Nimg, H, W, C, up2_scale(H), up2_scale(W));
  return check_launch("upsample_bilinear2x_bwd");
}

int mmft_pool2x2_fwd(const float* x, float* y, int Nimg, int H, int W, int C, int mode, int device, void* stream) {
  MMFT_REQUIRE(x && y && Nimg > 0 && H >= 2 && W >= 2 && C > 0 && (mode == 0 || mode == 1), "pool2x2_fwd: bad args");
  DeviceGuard dg(device);
  long long total = (long long)Nimg * (H / 2) * (W / 2) * C;
  MMFT_LAUNCH("pool_fwd_kernel", 0.0, 5.0 * total, pool_fwd_kernel, dim3(ew_grid(total)), dim3(256), (hipStream_t)stream, x, y, Nimg, H, W, C, mode);
  return check_launch("pool2x2_fwd");
}

int mmft_pool2x2_bwd(const float* x, const float* gy, float* dx, int Nimg, int H, int W, int C, int mode, int device,
                     void* stream) {
  MMFT_REQUIRE(x && gy && dx && Nimg > 0 && H >= 2 && W >= 2 && C > 0 && (mode == 0 || mode == 1), "pool2x2_bwd: bad args");
  DeviceGuard dg(device);
  long long total = (long long)Nimg * H * W * C;
  MMFT_LAUNCH("pool_bwd_kernel", 0.0, 9.0 * total, pool_bwd_kernel, dim3(ew_grid(total)), dim3(256), (hipStream_t)stream, x, gy, dx, Nimg, H, W, C, mode);
  return check_launch("pool2x2_bwd");
}

int mmft_pixel_shuffle2_into(const float* in, const float* bias, float* out, int Nimg, int H, int W, int Co, int ldc,
                             int c_off, int device, void* stream) {
  MMFT_REQUIRE(in && out && Nimg > 0 && H > 0 && W > 0 && Co > 0, "pixel_shuffle2: bad args");
  MMFT_REQUIRE(c_off >= 0 && ldc >= c_off + Co, "pixel_shuffle2: channel slice outside the destination");
  DeviceGuard dg(device);
  long long total = (long long)Nimg * H * W * 4 * Co;
  MMFT_LAUNCH("pixel_shuffle_kernel", 0.0, 8.0 * total, pixel_shuffle_kernel, dim3(ew_grid(total)), dim3(256), (hipStream_t)stream, in, bias, out, Nimg, H, W, Co, 0, ldc, c_off);
  return check_launch("pixel_shuffle2");
}

int mmft_pixel_shuffle2(const float* in, const float* bias, float* out, int Nimg, int H, int W, int Co, int device,
                        void* stream) {
  return mmft_pixel_shuffle2_into(in, bias, out, Nimg, H, W, Co, Co, 0, device, stream);
}

int mmft_pixel_unshuffle2_from(const float* in, float* out, int Nimg, int H, int W, int Co, int ldc, int c_off, int device,
                               void* stream) {
  MMFT_REQUIRE(in && out && Nimg > 0 && H > 0 && W > 0 && Co > 0, "pixel_unshuffle2: bad args");
  MMFT_REQUIRE(c_off >= 0 && ldc >= c_off + Co, "pixel_unshuffle2: channel slice outside the source");
  DeviceGuard dg(device);
  long long total = (long long)Nimg * H * W * 4 * Co;
  MMFT_LAUNCH("pixel_shuffle_kernel", 0.0, 8.0 * total, pixel_shuffle_kernel, dim3(ew_grid(total)), dim3(256), (hipStream_t)stream, in, nullptr, out, Nimg, H, W, Co, 1, ldc, c_off);
  return check_launch("pixel_unshuffle2");
}

int mmft_pixel_unshuffle2(const float* in, float* out, int Nimg, int H, int W, int Co, int device, void* stream) {
  return mmft_pixel_unshuffle2_from(in, out, Nimg, H, W, Co, Co, 0, device, stream);
}

int mmft_copy_region_nhwc(float* src, int Nimg, int Hs, int Ws, int Cs, float* dst, int Hd, int Wd, int Cd, int c_off,
                          int y_off, int x_off, int reverse, int device, void* stream) {
  MMFT_REQUIRE(src && dst && Nimg > 0 && Hs > 0 && Ws > 0 && Cs > 0, "copy_region_nhwc: bad args");
  MMFT_REQUIRE(y_off >= 0 && x_off >= 0 && c_off >= 0 && Hs + y_off <= Hd && Ws + x_off <= Wd && Cs + c_off <= Cd,
               "copy_region_nhwc: region outside destination");
  DeviceGuard dg(device);
  long long total = (long long)Nimg * Hs * Ws * Cs;
  MMFT_LAUNCH("copy_region_kernel", 0.0, 8.0 * total, copy_region_kernel, dim3(ew_grid(total)), dim3(256), (hipStream_t)stream, src, Nimg, Hs, Ws, Cs, dst, Hd, Wd, Cd, c_off, y_off, x_off, reverse);
  return check_launch("copy_region_nhwc");
}

int mmft_nchw_to_nhwc(const float* src, float* dst, int Nimg, int C, int H, int W, int Cpad, int device, void* stream) {
  MMFT_REQUIRE(src && dst && Nimg > 0 && C > 0 && H > 0 && W > 0 && Cpad >= C, "nchw_to_nhwc: bad args");
  DeviceGuard dg(device);
  long long total = (long long)Nimg * H * W * Cpad;
  hipLaunchKernelGGL(nchw_nhwc_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, src, dst, Nimg, C, H, W,
                     Cpad, 1);
  return check_launch("nchw_to_nhwc");
}

int mmft_nhwc_to_nchw(const float* src, float* dst, int Nimg, int C, int H, int W, int Cpad, int device, void* stream) {
  MMFT_REQUIRE(src && dst && Nimg > 0 && C > 0 && H > 0 && W > 0 && Cpad >= C, "nhwc_to_nchw: bad args");
  DeviceGuard dg(device);
  long long total = (long long)Nimg * H * W * Cpad;
  hipLaunchKernelGGL(nchw_nhwc_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, src, dst, Nimg, C, H, W,
                     Cpad, 0);
  return check_launch("nhwc_to_nchw");
}

}  // extern "C"

// bf16-storage element-wise / reduction kernels of the layout U-Net (bf16 math mode): train-mode BatchNorm + ReLU
// (src/Unet.py:17-18,20-21) split as  statistics (conv epilogue, unet16_conv.hip) -> finalize -> apply (+ 2x2 pooling of
// src/Unet.py:33-36 in the same pass, + the skip half of the concatenation of src/Unet.py:67 written in place),
// its backward (partial sums -> finalize -> apply), the pooling backward fused with the skip-connection gradient add, and
// OutConv (src/Unet.py:71-82).  All tensors bf16 NHWC in HBM (half the bytes of the fp32 kernels of cnn.hip), statistics
// fp32, combined in fp64 in a fixed order (bitwise reproducible).
#include "unet16.h"

namespace mmft {

__device__ __forceinline__ double wave_sum_d(double v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ---------------------------------------------------------------------------------------------- forward finalize
// One wave per (image, channel): the tile partials [tile][2][C] written by the convolution are combined in fp64;
// bnp = [5][N][C] = mean, invstd, biased variance, scale = gamma * invstd, shift = beta - mean * scale.
__global__ void __launch_bounds__(64) u16_bn_finalize_kernel(const float* __restrict__ stats, int tiles_per_image, int N, int C,
                                                             double count, float eps, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float* __restrict__ bnp) {
  const int img = blockIdx.x / C, c = blockIdx.x % C, lane = threadIdx.x;
  const long long NC = (long long)N * C;
  double s = 0.0, ss = 0.0;
  for (int t = lane; t < tiles_per_image; t += 64) {
    const float* p = stats + ((long long)(img * tiles_per_image + t) * 2) * C;
    s += (double)p[c];
    ss += (double)p[C + c];
  }
  s = wave_sum_d(s);
  ss = wave_sum_d(ss);
  if (lane == 0) {
    const double mean = s / count;
    double var = ss / count - mean * mean;
    if (var < 0.0) var = 0.0;
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    const float scale = gamma[c] * invstd;
    const long long o = (long long)img * C + c;
    bnp[o] = (float)mean;
    bnp[NC + o] = invstd;
    bnp[2 * NC + o] = (float)var;
    bnp[3 * NC + o] = scale;
    bnp[4 * NC + o] = __fmaf_rn(-(float)mean, scale, beta[c]);
  }
}

// running statistics: N sequential momentum updates per channel (one per image: per-image statistics = the reference's
// one-image batches, src/train.py:465).  Runs in block 0 of the apply kernels (C <= 256 = their block size).
struct U16Running {
  float* mean;
  float* var;
  float momentum;
  double count;
};
__device__ __forceinline__ void u16_running_update(const float* __restrict__ bnp, int N, int C, const U16Running& run) {
  const int c = threadIdx.x;
  if (c >= C) return;
  const long long NC = (long long)N * C;
  double rm = (double)run.mean[c], rv = (double)run.var[c];
  for (int img = 0; img < N; ++img) {
    const double var = (double)bnp[2 * NC + (long long)img * C + c];
    const double unbiased = run.count > 1.0 ? var * run.count / (run.count - 1.0) : var;
    rm = (1.0 - (double)run.momentum) * rm + (double)run.momentum * (double)bnp[(long long)img * C + c];
    rv = (1.0 - (double)run.momentum) * rv + (double)run.momentum * unbiased;
  }
  run.mean[c] = (float)rm;
  run.var[c] = (float)rv;
}

// ---------------------------------------------------------------------------------------------- forward apply (+ pool)
struct U16ApplyArgs {
  const u16* z;          // [N][H][W][C]
  const float* scale;    // [N][C]
  const float* shift;    // [N][C]
  u16* a;                // [N][H][W] pixels of pitch lda elements (a channel slice of a concatenation buffer), or plain (lda = C)
  int lda;
  u16* pooled;           // [N][H/2][W/2][C] or null
  int N, H, W, C, pool_mode;
  const float* bnp;      // for the running-statistics update (block 0)
  U16Running run;
};

// a = relu(z * scale + shift) rounded to bf16; item = (pixel, group of 8 channels)
__global__ void __launch_bounds__(256) u16_bn_apply_kernel(U16ApplyArgs p) {
  if (p.run.mean && blockIdx.x == 0) u16_running_update(p.bnp, p.N, p.C, p.run);
  const int CG = p.C / 8;
  const long long per_img = (long long)p.H * p.W * CG, total = per_img * p.N;
  for (long long it = (long long)blockIdx.x * 256 + threadIdx.x; it < total; it += (long long)gridDim.x * 256) {
    const int cg = (int)(it % CG);
    const long long pix = it / CG;
    const int img = (int)(it / per_img);
    float sc[8], sh[8], f[8];
    const float* s0 = p.scale + (long long)img * p.C + cg * 8;
    const float* h0 = p.shift + (long long)img * p.C + cg * 8;
    *reinterpret_cast<f32x4*>(sc) = *reinterpret_cast<const f32x4*>(s0);
    *reinterpret_cast<f32x4*>(sc + 4) = *reinterpret_cast<const f32x4*>(s0 + 4);
    *reinterpret_cast<f32x4*>(sh) = *reinterpret_cast<const f32x4*>(h0);
    *reinterpret_cast<f32x4*>(sh + 4) = *reinterpret_cast<const f32x4*>(h0 + 4);
    unpack8(*reinterpret_cast<const u32x4*>(p.z + pix * p.C + cg * 8), f);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float t = bn_pre(f[j], sc[j], sh[j]);
      f[j] = t > 0.f ? t : 0.f;
    }
    *reinterpret_cast<u32x4*>(p.a + pix * p.lda + cg * 8) = pack8(f);
  }
}

// the same + pooled = pool2x2(a) (max: of the rounded values, exact; avg: mean of the four rounded values, rounded);
// item = (2x2 window, group of 8 channels)
__global__ void __launch_bounds__(256) u16_bn_apply_pool_kernel(U16ApplyArgs p) {
  if (p.run.mean && blockIdx.x == 0) u16_running_update(p.bnp, p.N, p.C, p.run);
  const int CG = p.C / 8, H2 = p.H / 2, W2 = p.W / 2;
  const long long per_img = (long long)H2 * W2 * CG, total = per_img * p.N;
  for (long long it = (long long)blockIdx.x * 256 + threadIdx.x; it < total; it += (long long)gridDim.x * 256) {
    const int cg = (int)(it % CG);
    const long long win = it / CG;
    const int x2 = (int)(win % W2), y2 = (int)((win / W2) % H2), img = (int)(win / ((long long)W2 * H2));
    float sc[8], sh[8];
    const float* s0 = p.scale + (long long)img * p.C + cg * 8;
    const float* h0 = p.shift + (long long)img * p.C + cg * 8;
    *reinterpret_cast<f32x4*>(sc) = *reinterpret_cast<const f32x4*>(s0);
    *reinterpret_cast<f32x4*>(sc + 4) = *reinterpret_cast<const f32x4*>(s0 + 4);
    *reinterpret_cast<f32x4*>(sh) = *reinterpret_cast<const f32x4*>(h0);
    *reinterpret_cast<f32x4*>(sh + 4) = *reinterpret_cast<const f32x4*>(h0 + 4);
    const long long p00 = ((long long)img * p.H + 2 * y2) * p.W + 2 * x2;
    u32x4 zv[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
      zv[k] = *reinterpret_cast<const u32x4*>(p.z + (p00 + (k >> 1) * p.W + (k & 1)) * p.C + cg * 8);
    float best[8];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float f[8];
      unpack8(zv[k], f);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float t = bn_pre(f[j], sc[j], sh[j]);
        f[j] = t > 0.f ? t : 0.f;
      }
      const u32x4 av = pack8(f);
      *reinterpret_cast<u32x4*>(p.a + (p00 + (k >> 1) * p.W + (k & 1)) * p.lda + cg * 8) = av;
      unpack8(av, f);                                  // the ROUNDED values are what the pool (and its backward) sees
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (p.pool_mode == MMFT_POOL_MAX) best[j] = (k == 0 || f[j] > best[j] || f[j] != f[j]) ? f[j] : best[j];
        else best[j] = k == 0 ? f[j] : best[j] + f[j];
      }
    }
    if (p.pool_mode != MMFT_POOL_MAX) {
#pragma unroll
      for (int j = 0; j < 8; ++j) best[j] *= 0.25f;
    }
    *reinterpret_cast<u32x4*>(p.pooled + (((long long)img * H2 + y2) * W2 + x2) * p.C + cg * 8) = pack8(best);
  }
}

// ---------------------------------------------------------------------------------------------- backward partial sums
// per (image, row block): sum of gm = g * [pre > 0] and of gm * xhat, xhat = (z - mean) * invstd.  A thread owns one
// group of 8 channels and walks the rows of its block; partial[(block * 2 + {0, 1}) * C + c].
__global__ void __launch_bounds__(256) u16_bn_bwd_partial_kernel(const u16* __restrict__ g, const u16* __restrict__ z,
                                                                 const float* __restrict__ bnp, long long rows, int N, int C,
                                                                 int blocks_per_image, float* __restrict__ partial) {
  __shared__ float red[2][256 * 8];
  const int img = blockIdx.x / blocks_per_image, blk = blockIdx.x % blocks_per_image;
  const long long NC = (long long)N * C;
  const long long rpb = (rows + blocks_per_image - 1) / blocks_per_image;
  const long long r0 = (long long)blk * rpb, r1 = r0 + rpb < rows ? r0 + rpb : rows;
  const int CG = C / 8;                                  // <= 16
  const int rstep = 256 / CG;
  const int cg = threadIdx.x % CG, rl = threadIdx.x / CG;
  float s[8], sx[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) s[j] = sx[j] = 0.f;
  {
    float mu[8], is[8], sc[8], sh[8];
    const long long o = (long long)img * C + cg * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      mu[j] = bnp[o + j];
      is[j] = bnp[NC + o + j];
      sc[j] = bnp[3 * NC + o + j];
      sh[j] = bnp[4 * NC + o + j];
    }
    const long long base = (long long)img * rows;
#pragma unroll 2
    for (long long rr = r0 + rl; rr < r1; rr += rstep) {
      const long long e = (base + rr) * C + cg * 8;
      float gv[8], zf[8];
      unpack8(*reinterpret_cast<const u32x4*>(g + e), gv);
      unpack8(*reinterpret_cast<const u32x4*>(z + e), zf);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float gm = bn_pre(zf[j], sc[j], sh[j]) > 0.f ? gv[j] : 0.f;
        s[j] += gm;
        sx[j] = __fmaf_rn(gm, (zf[j] - mu[j]) * is[j], sx[j]);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    red[0][threadIdx.x * 8 + j] = s[j];
    red[1][threadIdx.x * 8 + j] = sx[j];
  }
  __syncthreads();
  for (int cc = threadIdx.x; cc < C; cc += 256) {
    float a0 = 0.f, a1 = 0.f;
    for (int k = 0; k < rstep; ++k) {
      a0 += red[0][(k * CG) * 8 + cc];
      a1 += red[1][(k * CG) * 8 + cc];
    }
    partial[((long long)blockIdx.x * 2) * C + cc] = a0;
    partial[((long long)blockIdx.x * 2 + 1) * C + cc] = a1;
  }
}

// one workgroup per channel (wave w: images w, w + 4, ...): coef[2][N][C] = mean of gm, mean of gm * xhat per image;
// dgamma[c] = sum over images of sum(gm * xhat), dbeta[c] = sum over images of sum(gm) (added in image order)
__global__ void __launch_bounds__(256) u16_bn_bwd_finalize_kernel(const float* __restrict__ partial, int blocks_per_image, int N, int C,
                                                                  double count, float* __restrict__ coef, float* __restrict__ dgamma,
                                                                  float* __restrict__ dbeta, int accumulate) {
  __shared__ double tot[2][64];
  const int c = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long NC = (long long)N * C;
  for (int img = wave; img < N; img += 4) {
    double s = 0.0, sx = 0.0;
    for (int b = lane; b < blocks_per_image; b += 64) {
      const float* p = partial + ((long long)(img * blocks_per_image + b) * 2) * C;
      s += (double)p[c];
      sx += (double)p[C + c];
    }
    s = wave_sum_d(s);
    sx = wave_sum_d(sx);
    if (lane == 0) {
      coef[(long long)img * C + c] = (float)(s / count);
      coef[NC + (long long)img * C + c] = (float)(sx / count);
      if (img < 64) {
        tot[0][img] = s;
        tot[1][img] = sx;
      }
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double db = 0.0, dg = 0.0;
    const int n = N < 64 ? N : 64;
    for (int img = 0; img < n; ++img) {
      db += tot[0][img];
      dg += tot[1][img];
    }
    for (int img = 64; img < N; ++img) {                 // beyond the LDS table: from the coefficients (never on this path)
      db += (double)coef[(long long)img * C + c] * count;
      dg += (double)coef[NC + (long long)img * C + c] * count;
    }
    dgamma[c] = accumulate ? dgamma[c] + (float)dg : (float)dg;
    dbeta[c] = accumulate ? dbeta[c] + (float)db : (float)db;
  }
}

// dz = gamma * invstd * (gm - mean(gm) - xhat * mean(gm * xhat)), rounded to bf16; item = (pixel, group of 8 channels)
__global__ void __launch_bounds__(256) u16_bn_bwd_apply_kernel(const u16* __restrict__ g, const u16* __restrict__ z,
                                                               const float* __restrict__ bnp, const float* __restrict__ coef,
                                                               u16* __restrict__ dz, long long rows, int N, int C) {
  const int CG = C / 8;
  const long long NC = (long long)N * C, per_img = rows * CG, total = per_img * N;
  for (long long it = (long long)blockIdx.x * 256 + threadIdx.x; it < total; it += (long long)gridDim.x * 256) {
    const int cg = (int)(it % CG);
    const long long pix = it / CG;
    const int img = (int)(it / per_img);
    const long long o = (long long)img * C + cg * 8, e = pix * C + cg * 8;
    float gv[8], zf[8], out[8];
    unpack8(*reinterpret_cast<const u32x4*>(g + e), gv);
    unpack8(*reinterpret_cast<const u32x4*>(z + e), zf);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float mu = bnp[o + j], is = bnp[NC + o + j], sc = bnp[3 * NC + o + j], sh = bnp[4 * NC + o + j];
      const float gm = bn_pre(zf[j], sc, sh) > 0.f ? gv[j] : 0.f;
      const float xh = (zf[j] - mu) * is;
      out[j] = sc * (gm - coef[o + j] - xh * coef[NC + o + j]);
    }
    *reinterpret_cast<u32x4*>(dz + e) = pack8(out);
  }
}

// ---------------------------------------------------------------------------------------------- pooling backward + skip add
// g[pixel] = gskip[pixel] + (the window's pooled gradient routed to torch's argmax: first maximum in scan order, NaN wins
// / a quarter of it for average pooling).  a: the stored activation (a slice of the concatenation buffer, pitch lda);
// gskip: the skip half of the concatenation's gradient (pitch ldg), may be null; item = (window, group of 8 channels).
struct U16PoolBwdArgs {
  const u16* a;
  int lda;
  const u16* gskip;
  int ldg;
  const u16* gp;       // [N][H/2][W/2][C]
  u16* g;              // [N][H][W][C]
  int N, H, W, C, pool_mode;
};

__global__ void __launch_bounds__(256) u16_pool_bwd_kernel(U16PoolBwdArgs p) {
  const int CG = p.C / 8, H2 = p.H / 2, W2 = p.W / 2;
  const long long total = (long long)p.N * H2 * W2 * CG;
  for (long long it = (long long)blockIdx.x * 256 + threadIdx.x; it < total; it += (long long)gridDim.x * 256) {
    const int cg = (int)(it % CG);
    const long long win = it / CG;
    const int x2 = (int)(win % W2), y2 = (int)((win / W2) % H2), img = (int)(win / ((long long)W2 * H2));
    const long long p00 = ((long long)img * p.H + 2 * y2) * p.W + 2 * x2;
    float gpv[8];
    unpack8(*reinterpret_cast<const u32x4*>(p.gp + win * p.C + cg * 8), gpv);
    float av[4][8];
#pragma unroll
    for (int k = 0; k < 4; ++k)
      unpack8(*reinterpret_cast<const u32x4*>(p.a + (p00 + (k >> 1) * p.W + (k & 1)) * p.lda + cg * 8), av[k]);
    int arg[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float m = av[0][j];
      arg[j] = 0;
#pragma unroll
      for (int k = 1; k < 4; ++k)
        if (av[k][j] > m || av[k][j] != av[k][j]) {
          m = av[k][j];
          arg[j] = k;
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const long long pix = p00 + (k >> 1) * p.W + (k & 1);
      float gs[8];
      if (p.gskip) {
        unpack8(*reinterpret_cast<const u32x4*>(p.gskip + pix * p.ldg + cg * 8), gs);
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) gs[j] = 0.f;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j)
        gs[j] += p.pool_mode == MMFT_POOL_MAX ? (arg[j] == k ? gpv[j] : 0.f) : 0.25f * gpv[j];
      *reinterpret_cast<u32x4*>(p.g + pix * p.C + cg * 8) = pack8(gs);
    }
  }
}

// ---------------------------------------------------------------------------------------------- OutConv (bf16 activations)
// outconv.hip with bf16 loads / stores: out = relu(pool2x2(b + sum_c w[c] x[c])) fp32; backward recomputes the pixel
// values from x, writes dx (bf16) and per-workgroup slabs of dw / db.
constexpr int OC16_CI = 16;

struct U16OutConvArgs {
  const u16* x;        // [N][H][W][16]
  const float* w;      // [16]
  const float* bias;   // [1] or null
  const float* gout;   // [N][H/2][W/2]
  float* out;          // [N][H/2][W/2]
  u16* dx;             // [N][H][W][16]
  float* slabs;        // [gridDim.x][17]
  int N, H, W, mode;
  long long items;     // N * (H / 2) * (W / 32)
};

__device__ __forceinline__ void oc16_window(const U16OutConvArgs& a, long long item, int lane, float xv[OC16_CI], float& pooled,
                                            int& arg, int& me, long long& pix, long long& opix) {
  const int wx = a.W / 32;
  const int xb = (int)(item % wx);
  const long long rp = item / wx;
  const int y2 = (int)(rp % (a.H / 2));
  const long long n = rp / (a.H / 2);
  const int rowbit = lane >> 5, xx = xb * 32 + (lane & 31);
  pix = (n * a.H + 2 * y2 + rowbit) * a.W + xx;
  opix = rp * (a.W / 2) + (xx >> 1);
  const u16* p = a.x + pix * OC16_CI;
  unpack8(*reinterpret_cast<const u32x4*>(p), xv);
  unpack8(*reinterpret_cast<const u32x4*>(p + 8), xv + 8);
  float v = a.bias ? a.bias[0] : 0.f;
#pragma unroll
  for (int c = 0; c < OC16_CI; ++c) v = __fmaf_rn(xv[c], a.w[c], v);
  me = rowbit * 2 + (xx & 1);
  const float vx = __shfl_xor(v, 1, 64), vy = __shfl_xor(v, 32, 64), vd = __shfl_xor(v, 33, 64);
  float qv[4];
  qv[0] = me == 0 ? v : me == 1 ? vx : me == 2 ? vy : vd;
  qv[1] = me == 1 ? v : me == 0 ? vx : me == 3 ? vy : vd;
  qv[2] = me == 2 ? v : me == 3 ? vx : me == 0 ? vy : vd;
  qv[3] = me == 3 ? v : me == 2 ? vx : me == 1 ? vy : vd;
  if (a.mode == MMFT_POOL_MAX) {
    float m = qv[0];
    arg = 0;
#pragma unroll
    for (int j = 1; j < 4; ++j)
      if (qv[j] > m || qv[j] != qv[j]) {
        m = qv[j];
        arg = j;
      }
    pooled = m;
  } else {
    pooled = (qv[0] + qv[1] + qv[2] + qv[3]) * 0.25f;
    arg = -1;
  }
}

__global__ void __launch_bounds__(256) u16_outconv_fwd_kernel(U16OutConvArgs a) {
  const int lane = threadIdx.x & 63;
  for (long long item = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); item < a.items; item += (long long)gridDim.x * 4) {
    float xv[OC16_CI], pooled;
    int arg, me;
    long long pix, opix;
    oc16_window(a, item, lane, xv, pooled, arg, me, pix, opix);
    if (me == 0) a.out[opix] = pooled > 0.f ? pooled : 0.f;
  }
}

__global__ void __launch_bounds__(256) u16_outconv_bwd_kernel(U16OutConvArgs a) {
  __shared__ float red[4][OC16_CI + 1];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float acc[OC16_CI + 1];
#pragma unroll
  for (int c = 0; c <= OC16_CI; ++c) acc[c] = 0.f;
  for (long long item = (long long)blockIdx.x * 4 + wave; item < a.items; item += (long long)gridDim.x * 4) {
    float xv[OC16_CI], pooled;
    int arg, me;
    long long pix, opix;
    oc16_window(a, item, lane, xv, pooled, arg, me, pix, opix);
    float g = a.gout[opix];
    if (!(pooled > 0.f)) g = 0.f;
    g = a.mode == MMFT_POOL_MAX ? (me == arg ? g : 0.f) : g * 0.25f;
    float d[OC16_CI];
#pragma unroll
    for (int c = 0; c < OC16_CI; ++c) {
      d[c] = a.w[c] * g;
      acc[c] = __fmaf_rn(g, xv[c], acc[c]);
    }
    acc[OC16_CI] += g;
    u16* dst = a.dx + pix * OC16_CI;
    *reinterpret_cast<u32x4*>(dst) = pack8(d);
    *reinterpret_cast<u32x4*>(dst + 8) = pack8(d + 8);
  }
#pragma unroll
  for (int c = 0; c <= OC16_CI; ++c) {
    float v = acc[c];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if (lane == 0) red[wave][c] = v;
  }
  __syncthreads();
  if (threadIdx.x <= OC16_CI)
    a.slabs[(long long)blockIdx.x * (OC16_CI + 1) + threadIdx.x] =
        ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
}

__global__ void __launch_bounds__(512) u16_outconv_reduce_kernel(const float* __restrict__ slabs, int nslab, float* __restrict__ dw,
                                                                 float* __restrict__ db, int accumulate) {
  __shared__ float part[8][64];
  const int e = threadIdx.x & 63, p = threadIdx.x >> 6;
  float s = 0.f;
  if (e <= OC16_CI)
    for (int b = p; b < nslab; b += 8) s += slabs[(long long)b * (OC16_CI + 1) + e];
  part[p][e] = s;
  __syncthreads();
  if (p == 0 && e <= OC16_CI) {
    float t = 0.f;
    for (int k = 0; k < 8; ++k) t += part[k][e];
    float* dst = e < OC16_CI ? dw + e : db;
    if (dst) *dst = accumulate ? *dst + t : t;
  }
}

static inline int u16_oc_grid(long long items) {
  long long g = (items + 3) / 4;
  if (g > 1024) g = 1024;
  return (int)(g < 1 ? 1 : g);
}

static inline int u16_bwd_blocks(long long rows) {         // row blocks per image of the backward partial sums
  long long b = (rows + 511) / 512;
  if (b > 256) b = 256;
  return (int)(b < 1 ? 1 : b);
}

}  // namespace mmft

using namespace mmft;

extern "C" {

int mmft_u16_bn_finalize(const float* stats, int tiles_per_image, int N, int C, long long pixels_per_image, float eps,
                         const float* gamma, const float* beta, float* bnp, int device, void* stream) {
  MMFT_REQUIRE(stats && gamma && beta && bnp && tiles_per_image > 0 && N > 0 && C > 0 && pixels_per_image > 0, "u16_bn_finalize: bad arguments");
  DeviceGuard dg(device);
  MMFT_LAUNCH("u16_bn_finalize_kernel", 0.0, 8.0 * N * tiles_per_image * C, u16_bn_finalize_kernel, dim3(N * C), dim3(64), (hipStream_t)stream,
              stats, tiles_per_image, N, C, (double)pixels_per_image, eps, gamma, beta, bnp);
  return check_launch("u16_bn_finalize");
}

int mmft_u16_bn_apply(const void* z, const float* bnp, void* a, int lda, void* pooled, int N, int H, int W, int C, int pool_mode,
                      float momentum, float* running_mean, float* running_var, int device, void* stream) {
  MMFT_REQUIRE(z && bnp && a && N > 0 && H > 0 && W > 0 && C >= 8 && C % 8 == 0 && C <= 256 && lda >= C && lda % 8 == 0,
               "u16_bn_apply: bad arguments");
  MMFT_REQUIRE(!pooled || (H % 2 == 0 && W % 2 == 0), "u16_bn_apply: pooling needs even H, W");
  MMFT_REQUIRE(!running_mean == !running_var, "u16_bn_apply: running_mean and running_var come together");
  MMFT_REQUIRE(aligned16(z) && aligned16(a) && aligned16(bnp) && (!pooled || aligned16(pooled)), "u16_bn_apply: 16-byte alignment");
  DeviceGuard dg(device);
  const long long NC = (long long)N * C;
  U16ApplyArgs p{reinterpret_cast<const u16*>(z), bnp + 3 * NC, bnp + 4 * NC, reinterpret_cast<u16*>(a), lda,
                 reinterpret_cast<u16*>(pooled), N, H, W, C, pool_mode, bnp, U16Running{running_mean, running_var, momentum, (double)H * W}};
  const double by = 2.0 * N * H * W * C * (pooled ? 2.25 : 2.0);
  if (pooled) {
    const long long items = (long long)N * (H / 2) * (W / 2) * (C / 8);
    MMFT_LAUNCH("u16_bn_apply_pool_kernel", 0.0, by, u16_bn_apply_pool_kernel, dim3(ew_grid(items)), dim3(256), (hipStream_t)stream, p);
  } else {
    const long long items = (long long)N * H * W * (C / 8);
    MMFT_LAUNCH("u16_bn_apply_kernel", 0.0, by, u16_bn_apply_kernel, dim3(ew_grid(items)), dim3(256), (hipStream_t)stream, p);
  }
  return check_launch("u16_bn_apply");
}

long long mmft_u16_bn_bwd_workspace_bytes(int N, long long pixels_per_image, int C) {
  return (long long)N * u16_bwd_blocks(pixels_per_image) * 2 * C * 4 + (long long)2 * N * C * 4;
}

/* dz = BatchNorm-train backward of g through relu(bn(z)); dgamma / dbeta fp32.  workspace: partial sums + coefficients. */
int mmft_u16_bn_bwd(const void* g, const void* z, const float* bnp, void* dz, float* dgamma, float* dbeta, int accumulate, int N,
                    long long pixels_per_image, int C, float* workspace, long long workspace_bytes, int device, void* stream) {
  MMFT_REQUIRE(g && z && bnp && dz && dgamma && dbeta && N > 0 && pixels_per_image > 0 && C >= 8 && C % 8 == 0 && C <= 128,
               "u16_bn_bwd: bad arguments (C a multiple of 8, <= 128)");
  MMFT_REQUIRE(workspace && workspace_bytes >= mmft_u16_bn_bwd_workspace_bytes(N, pixels_per_image, C), "u16_bn_bwd: workspace too small");
  MMFT_REQUIRE(aligned16(g) && aligned16(z) && aligned16(dz), "u16_bn_bwd: 16-byte alignment");
  DeviceGuard dg(device);
  hipStream_t st = (hipStream_t)stream;
  const int bpi = u16_bwd_blocks(pixels_per_image);
  float* partial = workspace;
  float* coef = workspace + (long long)N * bpi * 2 * C;
  const double elems = (double)N * pixels_per_image * C;
  MMFT_LAUNCH("u16_bn_bwd_partial_kernel", 0.0, 4.0 * elems, u16_bn_bwd_partial_kernel, dim3(N * bpi), dim3(256), st,
              reinterpret_cast<const u16*>(g), reinterpret_cast<const u16*>(z), bnp, pixels_per_image, N, C, bpi, partial);
  int rc = check_launch("u16_bn_bwd_partial");
  if (rc) return rc;
  MMFT_LAUNCH("u16_bn_bwd_finalize_kernel", 0.0, 8.0 * N * bpi * C, u16_bn_bwd_finalize_kernel, dim3(C), dim3(256), st, partial, bpi, N, C,
              (double)pixels_per_image, coef, dgamma, dbeta, accumulate ? 1 : 0);
  rc = check_launch("u16_bn_bwd_finalize");
  if (rc) return rc;
  const long long items = (long long)N * pixels_per_image * (C / 8);
  MMFT_LAUNCH("u16_bn_bwd_apply_kernel", 0.0, 6.0 * elems, u16_bn_bwd_apply_kernel, dim3(ew_grid(items)), dim3(256), st,
              reinterpret_cast<const u16*>(g), reinterpret_cast<const u16*>(z), bnp, coef, reinterpret_cast<u16*>(dz), pixels_per_image, N, C);
  return check_launch("u16_bn_bwd_apply");
}

int mmft_u16_pool_bwd(const void* a, int lda, const void* gskip, int ldg, const void* gp, void* g, int N, int H, int W, int C,
                      int pool_mode, int device, void* stream) {
  MMFT_REQUIRE(a && gp && g && N > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && C >= 8 && C % 8 == 0 && lda >= C && lda % 8 == 0 &&
                   (!gskip || (ldg >= C && ldg % 8 == 0)),
               "u16_pool_bwd: bad arguments");
  MMFT_REQUIRE(aligned16(a) && aligned16(gp) && aligned16(g) && (!gskip || aligned16(gskip)), "u16_pool_bwd: 16-byte alignment");
  DeviceGuard dg(device);
  U16PoolBwdArgs p{reinterpret_cast<const u16*>(a), lda, reinterpret_cast<const u16*>(gskip), ldg, reinterpret_cast<const u16*>(gp),
                   reinterpret_cast<u16*>(g), N, H, W, C, pool_mode};
  const long long items = (long long)N * (H / 2) * (W / 2) * (C / 8);
  MMFT_LAUNCH("u16_pool_bwd_kernel", 0.0, 2.0 * N * H * W * C * (gskip ? 3.25 : 2.25), u16_pool_bwd_kernel, dim3(ew_grid(items)), dim3(256),
              (hipStream_t)stream, p);
  return check_launch("u16_pool_bwd");
}

int mmft_u16_outconv_fwd(const void* x, const float* w, const float* bias, float* out, int N, int H, int W, int mode, int device,
                         void* stream) {
  MMFT_REQUIRE(x && w && out && N > 0 && H % 2 == 0 && W % 32 == 0 && (mode == MMFT_POOL_MAX || mode == MMFT_POOL_AVG) && aligned16(x),
               "u16_outconv_fwd: needs 16 input channels, even H, W %% 32 == 0");
  DeviceGuard dg(device);
  const long long items = (long long)N * (H / 2) * (W / 32);
  U16OutConvArgs a{reinterpret_cast<const u16*>(x), w, bias, nullptr, out, nullptr, nullptr, N, H, W, mode, items};
  MMFT_LAUNCH("u16_outconv_fwd_kernel", 2.0 * N * H * W * 16, 2.0 * N * H * W * 16 + 1.0 * N * H * W, u16_outconv_fwd_kernel,
              dim3(u16_oc_grid(items)), dim3(256), (hipStream_t)stream, a);
  return check_launch("u16_outconv_fwd");
}

long long mmft_u16_outconv_bwd_workspace_bytes(int N, int H, int W) {
  return (long long)u16_oc_grid((long long)N * (H / 2) * (W / 32)) * 17 * 4;
}

/* slabs of 17 floats: [dw[16] | db] */
int mmft_u16_outconv_bwd_slabs(int N, int H, int W) { return u16_oc_grid((long long)N * (H / 2) * (W / 32)); }

int mmft_u16_outconv_bwd(const void* x, const float* w, const float* bias, const float* gout, void* dx, float* dw, float* db,
                         int accumulate, int N, int H, int W, int mode, float* workspace, long long workspace_bytes, int device,
                         void* stream) {
  MMFT_REQUIRE(x && w && gout && dx && N > 0 && H % 2 == 0 && W % 32 == 0 && (mode == MMFT_POOL_MAX || mode == MMFT_POOL_AVG) &&
                   aligned16(x) && aligned16(dx),
               "u16_outconv_bwd: needs 16 input channels, even H, W %% 32 == 0");
  MMFT_REQUIRE(workspace && workspace_bytes >= mmft_u16_outconv_bwd_workspace_bytes(N, H, W), "u16_outconv_bwd: workspace too small");
  DeviceGuard dg(device);
  hipStream_t st = (hipStream_t)stream;
  const long long items = (long long)N * (H / 2) * (W / 32);
  const int grid = u16_oc_grid(items);
  U16OutConvArgs a{reinterpret_cast<const u16*>(x), w, bias, gout, nullptr, reinterpret_cast<u16*>(dx), workspace, N, H, W, mode, items};
  MMFT_LAUNCH("u16_outconv_bwd_kernel", 4.0 * N * H * W * 16, 4.0 * N * H * W * 16 + 1.0 * N * H * W, u16_outconv_bwd_kernel, dim3(grid),
              dim3(256), st, a);
  int rc = check_launch("u16_outconv_bwd");
  if (rc || !dw) return rc;          // dw == NULL: the slabs stay in `workspace` for mmft_slab_reduce_batch
  hipLaunchKernelGGL(u16_outconv_reduce_kernel, dim3(1), dim3(512), 0, st, workspace, grid, dw, db, accumulate ? 1 : 0);
  return check_launch("u16_outconv_reduce");
}

}  // extern "C"

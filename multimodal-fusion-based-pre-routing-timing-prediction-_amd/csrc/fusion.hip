// Fusion-head and step-glue kernels: masked projection of the CNN feature map (CSR masks, never the
// dense T x P path map), MSE loss + gradient, flat fused Adam.
// Replaces reference src/train.py:500-501 + src/model.py:271-272, src/train.py:32,522 and :431-435,555.
#include "common.h"

namespace mmft {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) transpose_kernel(const float* __restrict__ src, float* __restrict__ dst, int R,
                                                        int C) {
  __shared__ float tile[32][33];
  int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
  for (int j = ty; j < 32; j += 8) {
    int r = r0 + j, c = c0 + tx;
    tile[j][tx] = (r < R && c < C) ? src[(long long)r * C + c] : 0.f;
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    int c = c0 + j, r = r0 + tx;
    if (c < C && r < R) dst[(long long)c * R + r] = tile[tx][j];
  }
}

// one thread per (path row t, 4-channel group): gathers 16-B pieces of wT rows, like the graph kernels
__global__ void __launch_bounds__(256) masked_fc_fwd_kernel(const int* __restrict__ indptr, const int* __restrict__ cols,
                                                            const int* __restrict__ paths,
                                                            const int* __restrict__ foff, int T,
                                                            const float* __restrict__ f, const float* __restrict__ wT,
                                                            const float* __restrict__ bias, float* __restrict__ out,
                                                            int Dout) {
  const int groups = Dout >> 2;
  const long long total = (long long)T * groups;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    int t = (int)(i / groups), c = (int)(i - (long long)t * groups) * 4;
    int q = paths[t];
    const float* fb = f + (foff ? foff[t] : 0);
    f32x4 acc = bias ? *reinterpret_cast<const f32x4*>(bias + c) : f32x4{0.f, 0.f, 0.f, 0.f};
    for (int e = indptr[q]; e < indptr[q + 1]; ++e) {
      int p = cols[e];
      acc += *reinterpret_cast<const f32x4*>(wT + (long long)p * Dout + c) * fb[p];
    }
    *reinterpret_cast<f32x4*>(out + (long long)t * Dout + c) = acc;
  }
}

// one thread per (t, channel): each wave-instruction adds 256 contiguous bytes of one S row
__global__ void __launch_bounds__(256) masked_fc_bwd_scatter_kernel(const int* __restrict__ indptr,
                                                                    const int* __restrict__ cols,
                                                                    const int* __restrict__ paths,
                                                                    const int* __restrict__ foff, int T,
                                                                    const float* __restrict__ gout,
                                                                    float* __restrict__ S, int Dout) {
  const long long total = (long long)T * Dout;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    int t = (int)(i / Dout), c = (int)(i - (long long)t * Dout);
    int q = paths[t];
    float g = gout[i];
    float* Sb = S + (long long)(foff ? foff[t] : 0) * Dout;
    for (int e = indptr[q]; e < indptr[q + 1]; ++e) atomicAdd(Sb + (long long)cols[e] * Dout + c, g);
  }
}

// block = 32 columns p x all channels; dw written coalesced along p, df by an LDS column reduction
__global__ void __launch_bounds__(256) masked_fc_bwd_finish_kernel(const float* __restrict__ S, const float* __restrict__ f,
                                                                   const float* __restrict__ w, float* __restrict__ dw,
                                                                   float* __restrict__ df, int B, int P, int Dout) {
  __shared__ float part[8][32];
  int p = blockIdx.x * 32 + (threadIdx.x & 31);
  int cy = threadIdx.x >> 5;                      // 8 channel lanes
  if (p < P)
    for (int c = cy; c < Dout; c += 8) dw[(long long)c * P + p] = 0.f;
  for (int b = 0; b < B; ++b) {
    float acc = 0.f;
    if (p < P) {
      float fp = f[(long long)b * P + p];
      for (int c = cy; c < Dout; c += 8) {
        float s = S[((long long)b * P + p) * Dout + c];
        dw[(long long)c * P + p] += fp * s;
        acc += w[(long long)c * P + p] * s;
      }
    }
    __syncthreads();
    part[cy][threadIdx.x & 31] = acc;
    __syncthreads();
    if (threadIdx.x < 32 && p < P) {
      float s = 0.f;
      for (int j = 0; j < 8; ++j) s += part[j][threadIdx.x];
      df[(long long)b * P + p] = s;
    }
  }
}

__global__ void __launch_bounds__(1024) mse_kernel(const float* __restrict__ pred, const float* __restrict__ target, int n,
                                                   float* __restrict__ loss, float* __restrict__ grad) {
  __shared__ double red[1024];
  double s = 0.0;
  float inv = 2.0f / (float)n;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    float d = pred[i] - target[i];
    s += (double)d * (double)d;
    if (grad) grad[i] = d * inv;
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss[0] = (float)(red[0] / (double)n);
}

__global__ void __launch_bounds__(256) adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, long long n, float step_size, float beta1,
                                                   float beta2, float eps, float wd, float bc2_sqrt, float gscale) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    float gi = g[i] * gscale, pi = p[i];
    if (wd != 0.f) gi += wd * pi;
    float mi = m[i], vi = v[i];
    mi = mi + (gi - mi) * (1.0f - beta1);            // exp_avg.lerp_(grad, 1-beta1)
    vi = vi * beta2 + (1.0f - beta2) * gi * gi;      // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1-beta2)
    float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = pi - step_size * (mi / denom);
    m[i] = mi;
    v[i] = vi;
  }
}

}  // namespace mmft

using namespace mmft;

extern "C" {

int mmft_transpose(const float* src, float* dst, int R, int C, int device, void* stream) {
  MMFT_REQUIRE(src && dst && R > 0 && C > 0, "transpose: bad args");
  DeviceGuard dg(device);
  hipLaunchKernelGGL(transpose_kernel, dim3(cdiv(C, 32), cdiv(R, 32)), dim3(256), 0, (hipStream_t)stream, src, dst, R, C);
  return check_launch("transpose");
}

int mmft_masked_fc_fwd(const int* mask_indptr, const int* mask_cols, const int* paths, const int* f_off, int T,
                       const float* f, const float* wT, const float* bias, float* out, int P, int Dout, int device, void* stream) {
  MMFT_REQUIRE(mask_indptr && paths && f && wT && out, "masked_fc_fwd: null pointer");
  MMFT_REQUIRE(T >= 0 && P > 0 && Dout > 0 && Dout % 4 == 0, "masked_fc_fwd: bad sizes (Dout %% 4 == 0)");
  MMFT_REQUIRE(aligned16(wT) && aligned16(out) && (!bias || aligned16(bias)), "masked_fc_fwd: 16-byte alignment");
  if (T == 0) return MMFT_OK;
  DeviceGuard dg(device);
  ProfScope ps("masked_fc_fwd_kernel", 0.0, 0.0, (hipStream_t)stream);
  hipLaunchKernelGGL(masked_fc_fwd_kernel, dim3(ew_grid((long long)T * (Dout / 4))), dim3(256), 0, (hipStream_t)stream,
                     mask_indptr, mask_cols, paths, f_off, T, f, wT, bias, out, Dout);
  return check_launch("masked_fc_fwd");
}

int mmft_masked_fc_bwd_scatter(const int* mask_indptr, const int* mask_cols, const int* paths, const int* f_off, int T,
                               const float* gout, float* S, int P, int Dout, int device, void* stream) {
  MMFT_REQUIRE(mask_indptr && paths && gout && S, "masked_fc_bwd_scatter: null pointer");
  MMFT_REQUIRE(T >= 0 && P > 0 && Dout > 0, "masked_fc_bwd_scatter: bad sizes");
  if (T == 0) return MMFT_OK;
  DeviceGuard dg(device);
  ProfScope ps("masked_fc_bwd_scatter_kernel", 0.0, 0.0, (hipStream_t)stream);
  hipLaunchKernelGGL(masked_fc_bwd_scatter_kernel, dim3(ew_grid((long long)T * Dout)), dim3(256), 0, (hipStream_t)stream,
                     mask_indptr, mask_cols, paths, f_off, T, gout, S, Dout);
  return check_launch("masked_fc_bwd_scatter");
}

int mmft_masked_fc_bwd_finish(const float* S, const float* f, const float* w, float* dw, float* df, int B, int P,
                              int Dout, int device, void* stream) {
  MMFT_REQUIRE(S && f && w && dw && df && B > 0 && P > 0 && Dout > 0, "masked_fc_bwd_finish: bad args");
  DeviceGuard dg(device);
  ProfScope ps("masked_fc_bwd_finish_kernel", 0.0, 4.0 * ((double)B * P * Dout + 2.0 * P * Dout), (hipStream_t)stream);
  hipLaunchKernelGGL(masked_fc_bwd_finish_kernel, dim3(cdiv(P, 32)), dim3(256), 0, (hipStream_t)stream, S, f, w, dw, df, B, P,
                     Dout);
  return check_launch("masked_fc_bwd_finish");
}

int mmft_mse_fwd_bwd(const float* pred, const float* target, int n, float* loss, float* grad, int device, void* stream) {
  MMFT_REQUIRE(pred && target && loss && n > 0 && n <= (1 << 24), "mse_fwd_bwd: bad args");
  DeviceGuard dg(device);
  hipLaunchKernelGGL(mse_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, pred, target, n, loss, grad);
  return check_launch("mse_fwd_bwd");
}

int mmft_adam_step(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2, float eps,
                   float weight_decay, float bias_correction1, float bias_correction2, float gscale, int device,
                   void* stream) {
  MMFT_REQUIRE(p && g && m && v && n >= 0, "adam_step: bad args");
  MMFT_REQUIRE(bias_correction1 > 0.f && bias_correction2 > 0.f, "adam_step: bias corrections must be positive");
  if (n == 0) return MMFT_OK;
  DeviceGuard dg(device);
  ProfScope ps("adam_kernel", 0.0, 28.0 * n, (hipStream_t)stream);
  hipLaunchKernelGGL(adam_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr / bias_correction1,
                     beta1, beta2, eps, weight_decay, sqrtf(bias_correction2), gscale);
  return check_launch("adam_step");
}

}  // extern "C"

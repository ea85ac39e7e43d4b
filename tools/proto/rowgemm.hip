// Prototype: weight-stationary streaming GEMM  C[M][N] = X[M][K] . W[N][K]^T + bias  (fp32 MFMA 16x16x4)
// W lives in LDS for the whole (persistent) workgroup; every wave streams 16-row blocks of X straight into MFMA
// operand registers (no LDS, no barriers in the main loop), prefetching the next block while it multiplies.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include <type_traits>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int N, int K, int WAVES, int EXP>
__global__ void __launch_bounds__(WAVES * 64, 1) rowgemm_kernel(const float* __restrict__ X, const float* __restrict__ W,
                                                               const float* __restrict__ bias, float* __restrict__ C,
                                                               int M, unsigned long long* clk) {
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  constexpr int WS = K + 8;                 // LDS row stride (floats): conflict-free ds_read_b128 fragments
  constexpr int NT = N / 16, KB = K / 16, JG = 4;
  extern __shared__ __attribute__((aligned(16))) float wl0[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int g = tid; g < N * K / 4; g += WAVES * 64) {
    int n = g / (K / 4), k4 = g % (K / 4);
    *reinterpret_cast<f32x4*>(wl0 + n * WS + k4 * 4) = *reinterpret_cast<const f32x4*>(W + (long long)n * K + k4 * 4);
  }
  for (int g = tid; g < N; g += WAVES * 64) wl0[N * WS + g] = bias[g];
  __syncthreads();
  const int r = lane & 15, q = lane >> 4;
  const int nfull = M / 16;
  const int wstride = gridDim.x * WAVES;
  int blk = blockIdx.x * WAVES + wave;
  constexpr int G = KB * (NT / JG);
  f32x4 x[KB], xn[KB], stage[NT], acc[NT];
  auto load = [&](f32x4 (&dst)[KB], int b) {
    const float* p = X + (long long)(b * 16 + r) * K + 4 * q;
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) dst[kb] = *reinterpret_cast<const f32x4*>(p + kb * 16);
  };
  f32x4 w[2][JG];
  auto wread = [&](f32x4 (&dst)[JG], const float* wl, int g) {
    int kb = g / (NT / JG), j0 = (g % (NT / JG)) * JG;
#pragma unroll
    for (int jj = 0; jj < JG; ++jj)
      dst[jj] = *reinterpret_cast<const f32x4*>(wl + ((j0 + jj) * 16 + r) * WS + kb * 16 + 4 * q);
  };
  auto groups = [&](const float* wl, auto G0, auto G1) {
#pragma unroll
    for (int g = decltype(G0)::value; g < decltype(G1)::value; ++g) {
      if (g + 1 < G) wread(w[(g + 1) & 1], wl, g + 1);
      __builtin_amdgcn_sched_barrier(0);
      const int kb = g / (NT / JG), j0 = (g % (NT / JG)) * JG;
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int jj = 0; jj < JG; ++jj) {
          acc[j0 + jj] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[g & 1][jj][s], x[kb][s], acc[j0 + jj], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
    }
  };
  auto store_stage = [&](int b) {
    if (EXP & 8) b = blockIdx.x * WAVES + wave;          // EXPERIMENT: always the same rows (writes stay in L2)
    float* c = C + (long long)(b * 16 + r) * N + 4 * q;
#pragma unroll
    for (int j = 0; j < NT; ++j) *reinterpret_cast<f32x4*>(c + j * 16) = stage[j];
  };
  // One loop body, all memory traffic asynchronous to the MFMA stream:
  //   top: issue the loads of the NEXT block; middle: issue the stores of the PREVIOUS block (staged in registers);
  //   bottom: the only wait - by then the loads are a block old and the stores half a block old.
  int prev = -1;
  if (blk < nfull) load(x, blk);
  __builtin_amdgcn_s_waitcnt(0x0F70);        // enter the loop with nothing pending: same state as at the loop bottom
  while (blk < nfull) {
    int nb = blk + wstride;
    load(xn, nb < nfull ? nb : nfull - 1);
    int woff = 0;
    asm volatile("" : "+s"(woff));          // opaque per block: keeps the W fragment reads inside the loop
    const float* wl = wl0 + woff;
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    wread(w[0], wl, 0);
    groups(wl, std::integral_constant<int, 0>{}, std::integral_constant<int, G / 2>{});
    if (prev >= 0) store_stage(prev);
    __builtin_amdgcn_sched_barrier(0);
    groups(wl, std::integral_constant<int, G / 2>{}, std::integral_constant<int, G>{});
#pragma unroll
    for (int j = 0; j < NT; ++j) stage[j] = acc[j] + *reinterpret_cast<const f32x4*>(wl + N * WS + j * 16 + 4 * q);
    prev = blk;
    __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): next X block (a block old) and the stores (half a block old)
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) x[kb] = xn[kb];
    blk = nb;
  }
  if (prev >= 0) store_stage(prev);
  if (clk && threadIdx.x == 0 && blockIdx.x == 7) {
    clk[0] = __builtin_amdgcn_s_memtime() - t0;
    clk[1] = __builtin_amdgcn_s_memrealtime() - r0;
  }
}

template <int N, int K, int WAVES, int EXP>
void run(int M, int ctas_per_cu) {
  std::vector<float> hx((size_t)M * K), hw((size_t)N * K), hb(N);
  for (auto& v : hx) v = (float)rand() / RAND_MAX - 0.5f;
  for (auto& v : hw) v = (float)rand() / RAND_MAX - 0.5f;
  for (auto& v : hb) v = (float)rand() / RAND_MAX - 0.5f;
  float *x, *w, *b, *c; unsigned long long* clk; hipMalloc(&clk, 16);
  hipMalloc(&x, hx.size() * 4); hipMalloc(&w, hw.size() * 4); hipMalloc(&b, hb.size() * 4); hipMalloc(&c, (size_t)M * N * 4);
  hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(b, hb.data(), hb.size() * 4, hipMemcpyHostToDevice);
  size_t lds = (size_t)N * (K + 8) * 4 + N * 4;
  auto kern = rowgemm_kernel<N, K, WAVES, EXP>;
  hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  int grid = 256 * ctas_per_cu;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVES * 64), lds, 0, x, w, b, c, M, clk);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  const int it = 20;
  for (int i = 0; i < it; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVES * 64), lds, 0, x, w, b, c, M, clk);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double us = ms / it * 1e3;
  std::vector<float> hc((size_t)64 * N);
  int rows[4] = {0, 17, M / 2 + 3, M - 1};
  double maxerr = 0;
  for (int t = 0; t < 4; ++t) {
    hipMemcpy(hc.data(), c + (size_t)rows[t] * N, N * 4, hipMemcpyDeviceToHost);
    for (int n = 0; n < N; ++n) {
      double s = hb[n];
      for (int k = 0; k < K; ++k) s += (double)hx[(size_t)rows[t] * K + k] * hw[(size_t)n * K + k];
      maxerr = fmax(maxerr, fabs(s - hc[n]));
    }
  }
  unsigned long long hclk[2]; hipMemcpy(hclk, clk, 16, hipMemcpyDeviceToHost);
  printf("[core clock %.2f GHz] ", (double)hclk[0] / (double)hclk[1] * 0.1);
  printf("EXP=%d N=%d K=%d waves=%d ctas/cu=%d lds=%zu: %.1f us  %.1f TF  maxerr %.2e  (%s)\n", EXP, N, K, WAVES, ctas_per_cu, lds, us,
         2.0 * M * N * K / us / 1e6, maxerr, hipGetErrorString(hipGetLastError()));
  hipFree(x); hipFree(w); hipFree(b); hipFree(c);
}

int main() {
  const int M = 245760;
  run<256, 128, 8, 4>(M, 1);
  run<256, 128, 8, 12>(M, 1);
  run<128, 256, 8, 4>(M, 1);
  run<128, 256, 8, 12>(M, 1);
  return 0;
}

// Dense layers on the fp32 MFMA engine: forward, dgrad, wgrad, bias-gradient column sums, activations.
// Replaces th.nn.Linear / LeakyReLU inside MLP (reference src/model.py:10-24).
#include "gemm_engine.h"
#include <string.h>

namespace mmft {

static thread_local char g_err[512] = {0};

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// ---------------------------------------------------------------- math mode (process-wide; backward runs on another thread)
static int g_math_mode = 0;
int math_mode() { return __atomic_load_n(&g_math_mode, __ATOMIC_RELAXED); }

// ---------------------------------------------------------------- launch profiling
}  // namespace mmft
#include <mutex>
#include <string>
#include <vector>
#include <map>
namespace mmft {
struct ProfRec {
  const char* name;
  double flops, bytes;
  hipEvent_t e0, e1;
};
static bool g_prof = false;
static std::mutex g_prof_mu;
static std::vector<ProfRec> g_recs;
static std::vector<hipEvent_t> g_pool;
static ProfRec g_open;
// Accounting for the NEXT launch of the calling thread whose site cannot know its own work (data-dependent gathers: the
// host knows the run / edge counts of the step): mmft_prof_hint; consumed by the first launch that records zero counts.
static thread_local double t_hint_flops = 0.0, t_hint_bytes = 0.0;
static void take_hint(double* flops, double* bytes) {
  if (*flops == 0.0 && *bytes == 0.0 && (t_hint_flops != 0.0 || t_hint_bytes != 0.0)) {
    *flops = t_hint_flops;
    *bytes = t_hint_bytes;
    t_hint_flops = t_hint_bytes = 0.0;
  }
}

bool prof_on() { return g_prof; }
static hipEvent_t prof_event() {
  if (!g_pool.empty()) {
    hipEvent_t e = g_pool.back();
    g_pool.pop_back();
    return e;
  }
  hipEvent_t e;
  (void)hipEventCreate(&e);
  return e;
}
void prof_begin(const char* name, double flops, double bytes, hipStream_t st) {
  take_hint(&flops, &bytes);
  g_prof_mu.lock();                       // held until prof_end: launches are serialised while profiling
  g_open = ProfRec{name, flops, bytes, prof_event(), prof_event()};
  (void)hipEventRecord(g_open.e0, st);
}
void prof_events(const char* name, double flops, double bytes, hipEvent_t* e0, hipEvent_t* e1) {
  take_hint(&flops, &bytes);
  g_prof_mu.lock();
  g_open = ProfRec{name, flops, bytes, prof_event(), prof_event()};
  *e0 = g_open.e0;
  *e1 = g_open.e1;
}
void prof_commit() {
  g_recs.push_back(g_open);
  g_prof_mu.unlock();
}
void prof_end(hipStream_t st) {
  (void)hipEventRecord(g_open.e1, st);
  g_recs.push_back(g_open);
  g_prof_mu.unlock();
}

// ---------------------------------------------------------------- split-K slab combine
// out[i] = (accumulate ? out[i] : 0) + sum_z slab[z][i].  256 threads = 16 slab lanes x 16 column groups of
// VEC elements: every thread sums its share of the slabs with independent loads (latency overlaps), the 16
// partials are combined through LDS in a fixed order -> bitwise reproducible for a given split count.
template <int VEC>
__global__ void __launch_bounds__(256) slab_reduce_kernel(const float* __restrict__ slabs, int splits, long long elems,
                                                          float* __restrict__ out, int accumulate, long long stride) {
  __shared__ float part[16][16 * VEC + 1];
  const int zl = threadIdx.x >> 4, cg = threadIdx.x & 15;
  const long long i0 = ((long long)blockIdx.x * 16 + cg) * VEC;
  float acc[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
  if (i0 < elems) {
    for (int z = zl; z < splits; z += 16) {
      const float* q = slabs + (long long)z * stride + i0;
      if (VEC == 4) {
        f32x4 v = *reinterpret_cast<const f32x4*>(q);
        acc[0] += v.x; acc[1] += v.y; acc[2] += v.z; acc[3] += v.w;
      } else {
        acc[0] += q[0];
      }
    }
  }
#pragma unroll
  for (int j = 0; j < VEC; ++j) part[zl][cg * VEC + j] = acc[j];
  __syncthreads();
  if (threadIdx.x < 16 * VEC) {
    long long i = (long long)blockIdx.x * 16 * VEC + threadIdx.x;
    if (i < elems) {
      float s = accumulate ? out[i] : 0.f;
#pragma unroll
      for (int z = 0; z < 16; ++z) s += part[z][threadIdx.x];
      out[i] = s;
    }
  }
}

// Several slab reductions in ONE launch (the U-Net's 18 weight-gradient reductions of a step: each of them alone is a
// handful of workgroups walking a few hundred slabs - 4-10 us of latency per launch, 50 us for the serial forms they
// replace - and nothing downstream needs any of them before the optimizer).  out[i] (+)= sum over slabs z (and over `fold`
// consecutive segments of `elems` floats inside a slab: the four (a, b) column sums of a transposed convolution's bias)
// of slabs[z * stride + f * elems + i]; per result the same 16 z-lanes / fixed LDS order as slab_reduce_kernel.
constexpr int SLAB_BATCH_MAX = 24;
struct SlabBatchDesc {
  const float* slabs;
  float* out;
  long long stride;
  int splits, elems, fold, accumulate, blk0, vec;
};
struct SlabBatchArgs {
  SlabBatchDesc d[SLAB_BATCH_MAX];
  int n;
};

__global__ void __launch_bounds__(256) slab_reduce_batch_kernel(SlabBatchArgs a) {
  __shared__ float part[16][65];
  int k = 0;
#pragma unroll 1
  for (int i = 1; i < a.n; ++i)
    if ((int)blockIdx.x >= a.d[i].blk0) k = i;
  const SlabBatchDesc& d = a.d[k];
  const int zl = threadIdx.x >> 4, cg = threadIdx.x & 15, vec = d.vec ? 4 : 1;
  const long long i0 = ((long long)((int)blockIdx.x - d.blk0) * 16 + cg) * vec;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  if (i0 < d.elems) {
    const long long zs = 16 * d.stride;
    for (int f = 0; f < d.fold; ++f) {
      const float* q = d.slabs + (long long)zl * d.stride + (long long)f * d.elems + i0;
      int z = zl;
      if (d.vec) {
        for (; z + 48 < d.splits; z += 64, q += 4 * zs) {                 // four slabs of this lane in flight
          const f32x4 v0 = *reinterpret_cast<const f32x4*>(q), v1 = *reinterpret_cast<const f32x4*>(q + zs);
          const f32x4 v2 = *reinterpret_cast<const f32x4*>(q + 2 * zs), v3 = *reinterpret_cast<const f32x4*>(q + 3 * zs);
          acc[0] += v0.x; acc[1] += v0.y; acc[2] += v0.z; acc[3] += v0.w;
          acc[0] += v1.x; acc[1] += v1.y; acc[2] += v1.z; acc[3] += v1.w;
          acc[0] += v2.x; acc[1] += v2.y; acc[2] += v2.z; acc[3] += v2.w;
          acc[0] += v3.x; acc[1] += v3.y; acc[2] += v3.z; acc[3] += v3.w;
        }
        for (; z < d.splits; z += 16, q += zs) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(q);
          acc[0] += v.x; acc[1] += v.y; acc[2] += v.z; acc[3] += v.w;
        }
      } else {
        for (; z + 48 < d.splits; z += 64, q += 4 * zs) {
          const float v0 = q[0], v1 = q[zs], v2 = q[2 * zs], v3 = q[3 * zs];
          acc[0] += v0; acc[0] += v1; acc[0] += v2; acc[0] += v3;
        }
        for (; z < d.splits; z += 16, q += zs) acc[0] += q[0];
      }
    }
  }
  for (int j = 0; j < vec; ++j) part[zl][cg * vec + j] = acc[j];
  __syncthreads();
  if ((int)threadIdx.x < 16 * vec) {
    const long long i = (long long)((int)blockIdx.x - d.blk0) * 16 * vec + threadIdx.x;
    if (i < d.elems) {
      float s = d.accumulate ? d.out[i] : 0.f;
#pragma unroll
      for (int z = 0; z < 16; ++z) s += part[z][threadIdx.x];
      d.out[i] = s;
    }
  }
}

// slab z starts at slabs + z * stride (stride >= elems: slabs that carry more than one result, e.g. [weights | column sums])
int launch_slab_reduce_strided(const float* slabs, int splits, long long stride, long long elems, float* out, int accumulate,
                               hipStream_t st) {
  const double by = 4.0 * (splits + 1.0) * elems;
  if (elems % 4 == 0 && stride % 4 == 0 && aligned16(slabs)) {
    MMFT_LAUNCH("slab_reduce_kernel", 0.0, by, slab_reduce_kernel<4>, dim3(cdiv(elems, 64)), dim3(256), st, slabs, splits,
                elems, out, accumulate, stride);
  } else {
    MMFT_LAUNCH("slab_reduce_kernel", 0.0, by, slab_reduce_kernel<1>, dim3(cdiv(elems, 16)), dim3(256), st, slabs, splits,
                elems, out, accumulate, stride);
  }
  return check_launch("slab_reduce");
}

int launch_slab_reduce(const float* slabs, int splits, long long elems, float* out, int accumulate, hipStream_t st) {
  return launch_slab_reduce_strided(slabs, splits, elems, elems, out, accumulate, st);
}

// n <= SLAB_BATCH_MAX reductions in one launch (see slab_reduce_batch_kernel); per result the summation order of
// slab_reduce_kernel, so a result does not depend on which of the two launchers produced it
int launch_slab_reduce_batch(const SlabSeg* segs, int n, hipStream_t st) {
  if (n <= 0) return MMFT_OK;
  if (n > SLAB_BATCH_MAX) {
    set_error("slab_reduce_batch: at most %d reductions per launch (got %d)", SLAB_BATCH_MAX, n);
    return MMFT_ERR_BAD_ARG;
  }
  SlabBatchArgs a;
  a.n = n;
  int blocks = 0;
  double by = 0.0;
  for (int i = 0; i < n; ++i) {
    SlabBatchDesc& d = a.d[i];
    d.slabs = segs[i].slabs;
    d.out = segs[i].out;
    d.stride = segs[i].stride;
    d.splits = segs[i].splits;
    d.elems = segs[i].elems;
    d.fold = segs[i].fold;
    d.accumulate = segs[i].accumulate ? 1 : 0;
    if (!(d.slabs && d.out && d.splits > 0 && d.elems > 0 && d.fold >= 1 && d.stride >= (long long)d.elems * d.fold)) {
      set_error("slab_reduce_batch: bad descriptor %d", i);
      return MMFT_ERR_BAD_ARG;
    }
    d.vec = (d.elems % 4 == 0 && d.stride % 4 == 0 && aligned16(d.slabs) && aligned16(d.out)) ? 1 : 0;
    d.blk0 = blocks;
    blocks += cdiv(d.elems, d.vec ? 64 : 16);
    by += 4.0 * ((double)d.splits * d.fold + 1.0) * d.elems;
  }
  MMFT_LAUNCH("slab_reduce_batch_kernel", 0.0, by, slab_reduce_batch_kernel, dim3(blocks), dim3(256), st, a);
  return check_launch("slab_reduce_batch");
}

// ---------------------------------------------------------------- column sums (bias gradients)
// stage 1: block = 16 row lanes x 16 column groups (4 columns each) over a strip of rows -> partial[strip][cols];
// stage 2 = slab_reduce (fixed order)
// The strip shrinks for short inputs: 10 800 head rows in strips of 1024 were 22 workgroups walking 64 rows per lane (24 us);
// strips of 128 rows are 170 workgroups (the launch is latency, not bandwidth).
static inline int colsum_strip(int rows) { return rows <= (1 << 16) ? 128 : 1024; }
__global__ void __launch_bounds__(256) colsum_partial_kernel(const float* __restrict__ g, const int* __restrict__ idx,
                                                             long long ld, int rows, int cols,
                                                             float* __restrict__ partial, int COLSUM_STRIP) {
  __shared__ float part[16][65];
  const int rl = threadIdx.x >> 4, cg = threadIdx.x & 15;
  const int c0 = blockIdx.x * 64 + cg * 4;
  const int r0 = blockIdx.y * COLSUM_STRIP;
  const int r1 = r0 + COLSUM_STRIP < rows ? r0 + COLSUM_STRIP : rows;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  const bool vec = (ld % 4 == 0) && (c0 + 3 < cols) && ((reinterpret_cast<uintptr_t>(g) & 15) == 0);
  if (c0 < cols) {
    for (int r = r0 + rl; r < r1; r += 16) {
      long long rr = idx ? (long long)idx[r] : (long long)r;
      const float* q = g + rr * ld + c0;
      if (vec) {
        f32x4 v = *reinterpret_cast<const f32x4*>(q);
        acc[0] += v.x; acc[1] += v.y; acc[2] += v.z; acc[3] += v.w;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (c0 + j < cols) acc[j] += q[j];
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) part[rl][cg * 4 + j] = acc[j];
  __syncthreads();
  if (threadIdx.x < 64) {
    int c = blockIdx.x * 64 + threadIdx.x;
    if (c < cols) {
      float s = 0.f;
#pragma unroll
      for (int z = 0; z < 16; ++z) s += part[z][threadIdx.x];
      partial[(long long)blockIdx.y * cols + c] = s;
    }
  }
}

static inline int colsum_blocks(int rows) {
  int nb = cdiv(rows, colsum_strip(rows));
  return nb < 1 ? 1 : nb;
}

__global__ void __launch_bounds__(256) act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                      float* __restrict__ dpre, long long n, int act, float slope) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long stride = (long long)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    float g = dy[i];
    if (act != ACT_NONE && !(y[i] > 0.f)) g = (act == ACT_LEAKY) ? g * slope : 0.f;
    dpre[i] = g;
  }
}

__global__ void __launch_bounds__(256) act_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long long n,
                                                      int act, float slope) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long stride = (long long)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    float v = x[i];
    if (act == ACT_RELU) v = v > 0.f ? v : 0.f;
    else if (act == ACT_LEAKY) v = v > 0.f ? v : v * slope;
    y[i] = v;
  }
}

// one device timestamp (100 MHz wall clock) into slots[index]: marks a point of a stream inside a captured graph, where host
// timers and the profiler's serialising kernel trace see nothing (tools/step_timeline.py)
__global__ void stamp_kernel(unsigned long long* slots, int index) { slots[index] = wall_clock64(); }

}  // namespace mmft

using namespace mmft;

extern "C" {

int mmft_version(void) { return 200; }

int mmft_set_math_mode(int mode) {
  MMFT_REQUIRE(mode == MMFT_MATH_F32 || mode == MMFT_MATH_BF16, "set_math_mode: unknown mode %d", mode);
  __atomic_store_n(&mmft::g_math_mode, mode, __ATOMIC_RELAXED);
  return MMFT_OK;
}

int mmft_get_math_mode(void) { return mmft::math_mode(); }

int mmft_prof_enable(int on) {
  std::lock_guard<std::mutex> lk(mmft::g_prof_mu);
  mmft::g_prof = on != 0;
  return MMFT_OK;
}

int mmft_prof_stamp(unsigned long long* slots, int index, int device, void* stream) {
  MMFT_REQUIRE(slots && index >= 0, "prof_stamp: bad arguments");
  DeviceGuard dg(device);
  hipLaunchKernelGGL(mmft::stamp_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, slots, index);
  return check_launch("prof_stamp");
}

int mmft_prof_hint(double flops, double bytes) {
  mmft::t_hint_flops = flops;
  mmft::t_hint_bytes = bytes;
  return MMFT_OK;
}

int mmft_prof_reset(void) {
  std::lock_guard<std::mutex> lk(mmft::g_prof_mu);
  for (auto& r : mmft::g_recs) {
    mmft::g_pool.push_back(r.e0);
    mmft::g_pool.push_back(r.e1);
  }
  mmft::g_recs.clear();
  return MMFT_OK;
}

// one line per kernel name: name \t launches \t total_ms \t flops \t bytes ; returns bytes written (or needed)
int mmft_prof_report(char* buf, int cap) {
  std::lock_guard<std::mutex> lk(mmft::g_prof_mu);
  struct Agg { long long n = 0; double ms = 0, flops = 0, bytes = 0; };
  std::map<std::string, Agg> agg;
  for (auto& r : mmft::g_recs) {
    (void)hipEventSynchronize(r.e1);
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess) ms = 0.f;
    Agg& a = agg[r.name];
    a.n++;
    a.ms += ms;
    a.flops += r.flops;
    a.bytes += r.bytes;
  }
  std::string out;
  char line[512];
  for (auto& kv : agg) {
    snprintf(line, sizeof(line), "%s\t%lld\t%.6f\t%.6e\t%.6e\n", kv.first.c_str(), kv.second.n, kv.second.ms,
             kv.second.flops, kv.second.bytes);
    out += line;
  }
  if (buf && cap > 0) {
    int n = (int)out.size() < cap - 1 ? (int)out.size() : cap - 1;
    memcpy(buf, out.data(), n);
    buf[n] = 0;
  }
  return (int)out.size() + 1;
}
const char* mmft_last_error(void) { return mmft::g_err; }

int mmft_linear_fwd(const float* x, const int* xidx, long long ldx, const float* w, long long ldw, const float* bias,
                    float* y, const int* yidx, long long ldy, int M, int N, int K, int epi_mode, int act, float slope,
                    int device, void* stream) {
  MMFT_REQUIRE(x && w && y, "linear_fwd: null pointer");
  MMFT_REQUIRE(M >= 0 && N > 0 && K > 0, "linear_fwd: bad sizes M=%d N=%d K=%d", M, N, K);
  MMFT_REQUIRE(ldx >= K && ldw >= K && ldy >= N, "linear_fwd: leading dimension smaller than row length");
  MMFT_REQUIRE(epi_mode >= 0 && epi_mode <= 2 && act >= 0 && act <= 2, "linear_fwd: bad epilogue mode/act");
  if (M == 0) return MMFT_OK;
  DeviceGuard dg(device);
  DenseMK xl{x, xidx, ldx, M, (ldx % 4 == 0) && aligned16(x)};
  DenseMK wl{w, nullptr, ldw, N, (ldw % 4 == 0) && aligned16(w)};
  Epi epi{y, ldy, yidx, bias, nullptr, nullptr, 0, epi_mode, act, slope, 0, (ldy % 4 == 0) && aligned16(y)};
  return launch_gemm(xl, wl, epi, M, N, K, 1, (hipStream_t)stream);
}

int mmft_linear_dgrad(const float* g, const int* gidx, long long ldg, const float* w, long long ldw, float* dx,
                      const int* dxidx, long long lddx, int M, int N, int K, const float* mask, const int* maskidx,
                      long long ldmask, int epi_mode, int device, void* stream) {
  MMFT_REQUIRE(g && w && dx, "linear_dgrad: null pointer");
  MMFT_REQUIRE(M >= 0 && N > 0 && K > 0, "linear_dgrad: bad sizes");
  MMFT_REQUIRE(ldg >= K && ldw >= N && lddx >= N, "linear_dgrad: leading dimension smaller than row length");
  MMFT_REQUIRE(epi_mode == MMFT_EPI_STORE || epi_mode == MMFT_EPI_ACCUM, "linear_dgrad: bad epilogue mode");
  MMFT_REQUIRE(!(mask && epi_mode != MMFT_EPI_STORE), "linear_dgrad: mask requires STORE");
  if (M == 0) return MMFT_OK;
  DeviceGuard dg(device);
  DenseMK xl{g, gidx, ldg, M, (ldg % 4 == 0) && aligned16(g)};
  DenseKM wl{w, nullptr, ldw, N, (ldw % 4 == 0) && aligned16(w)};
  Epi epi{dx, lddx, dxidx, nullptr, mask, maskidx, ldmask, mask ? EPI_MASK : epi_mode, ACT_NONE, 0.f, 0,
          (lddx % 4 == 0) && aligned16(dx)};
  return launch_gemm(xl, wl, epi, M, N, K, 1, (hipStream_t)stream);
}

static int wgrad_splits(int rows, int out, int in) {
  long long tiles = (long long)cdiv(out, 128) * cdiv(in, in <= 16 ? 16 : in <= 32 ? 32 : in <= 64 ? 64 : 128);
  int want = (int)(768 / (tiles > 0 ? tiles : 1));
  int maxs = rows / 64;
  if (want > maxs) want = maxs;
  if (want < 1) want = 1;
  if (want > 256) want = 256;
  return effective_splits(rows, want);
}

long long mmft_linear_wgrad_workspace_bytes(int rows, int out, int in) {
  int s = wgrad_splits(rows, out, in);
  return s > 1 ? (long long)s * out * in * 4 : 0;
}

long long mmft_linear_wgrad_bias_workspace_bytes(int rows, int out, int in) {
  int s = wgrad_splits(rows, out, in);
  return s > 1 ? (long long)s * ((long long)out * in + out) * 4 : 0;
}

// dw (+)= g^T x and, when db is given, db (+)= column sums of g from the same pass over g
static int wgrad_impl(const float* g, const int* gidx, long long ldg, const float* x, const int* xidx, long long ldx,
                      float* dw, long long lddw, float* db, int rows, int out, int in, int accumulate, float* workspace,
                      long long workspace_bytes, int device, void* stream) {
  MMFT_REQUIRE(g && x && dw, "linear_wgrad: null pointer");
  MMFT_REQUIRE(rows >= 0 && out > 0 && in > 0, "linear_wgrad: bad sizes");
  MMFT_REQUIRE(ldg >= out && ldx >= in && lddw >= in, "linear_wgrad: leading dimension smaller than row length");
  DeviceGuard dg(device);
  hipStream_t st = (hipStream_t)stream;
  int splits = wgrad_splits(rows, out, in);
  if (rows == 0) {
    if (!accumulate) {
      for (int o = 0; o < out; ++o) (void)hipMemsetAsync(dw + (long long)o * lddw, 0, (size_t)in * 4, st);
      if (db) (void)hipMemsetAsync(db, 0, (size_t)out * 4, st);
    }
    return MMFT_OK;
  }
  // M := out (m-contiguous in g), N := in (n-contiguous in x), K := rows
  DenseKM xl{g, gidx, ldg, out, (ldg % 4 == 0) && aligned16(g)};
  DenseKM wl{x, xidx, ldx, in, (ldx % 4 == 0) && aligned16(x)};
  if (splits <= 1 || lddw != in) {
    // single pass straight into dw (also the path for strided dw, where slabs would not line up)
    Epi epi{dw, lddw, nullptr, nullptr, nullptr, nullptr, 0, accumulate ? EPI_ACCUM : EPI_STORE, ACT_NONE, 0.f, 0,
            (lddw % 4 == 0) && aligned16(dw), db, 0, accumulate};
    return launch_gemm(xl, wl, epi, out, in, rows, 1, st);
  }
  long long need = (long long)splits * ((long long)out * in + (db ? out : 0)) * 4;
  MMFT_REQUIRE(workspace && workspace_bytes >= need, "linear_wgrad: workspace too small (%lld < %lld)", workspace_bytes,
               need);
  // db right behind dw (weight and bias of a Linear are neighbours in the flat gradient buffer): the column-sum
  // slab rides at the tail of each weight slab and ONE reduction launch finishes both
  const bool joined = db && db == dw + (long long)out * in && ((long long)out * in) % 4 == 0 && out % 4 == 0;
  const long long wslab = (long long)out * in + (joined ? out : 0);
  float* cs_slabs = !db ? nullptr : (joined ? workspace + (long long)out * in : workspace + (long long)splits * out * in);
  Epi epi{workspace, in, nullptr, nullptr, nullptr, nullptr, 0, EPI_STORE, ACT_NONE, 0.f, wslab,
          (in % 4 == 0) && aligned16(workspace), cs_slabs, joined ? wslab : (long long)out, 0};
  int rc = launch_gemm(xl, wl, epi, out, in, rows, splits, st);
  if (rc) return rc;
  if (joined) return launch_slab_reduce(workspace, splits, wslab, dw, accumulate, st);
  rc = launch_slab_reduce(workspace, splits, (long long)out * in, dw, accumulate, st);
  if (rc || !db) return rc;
  return launch_slab_reduce(cs_slabs, splits, out, db, accumulate, st);
}

int mmft_linear_wgrad(const float* g, const int* gidx, long long ldg, const float* x, const int* xidx, long long ldx,
                      float* dw, long long lddw, int rows, int out, int in, int accumulate, float* workspace,
                      long long workspace_bytes, int device, void* stream) {
  return wgrad_impl(g, gidx, ldg, x, xidx, ldx, dw, lddw, nullptr, rows, out, in, accumulate, workspace, workspace_bytes,
                    device, stream);
}

int mmft_linear_wgrad_bias(const float* g, const int* gidx, long long ldg, const float* x, const int* xidx, long long ldx,
                           float* dw, long long lddw, float* db, int rows, int out, int in, int accumulate,
                           float* workspace, long long workspace_bytes, int device, void* stream) {
  MMFT_REQUIRE(db, "linear_wgrad_bias: null bias-gradient pointer");
  return wgrad_impl(g, gidx, ldg, x, xidx, ldx, dw, lddw, db, rows, out, in, accumulate, workspace, workspace_bytes,
                    device, stream);
}

/* n <= 24 slab reductions in one launch.  table (HOST memory) = n rows of seven 64-bit integers: slabs (device pointer), out
 * (device pointer), splits, stride (floats between slabs), elems, fold (segments of `elems` floats added together, >= 1),
 * accumulate.  Each result is summed in a fixed order (bitwise reproducible). */
int mmft_slab_reduce_batch(const long long* table, int n, int device, void* stream) {
  MMFT_REQUIRE(table && n >= 0 && n <= SLAB_BATCH_MAX, "slab_reduce_batch: 0 .. %d reductions per launch", SLAB_BATCH_MAX);
  if (n == 0) return MMFT_OK;
  SlabSeg segs[SLAB_BATCH_MAX];
  for (int i = 0; i < n; ++i) {
    const long long* t = table + 7 * i;
    segs[i] = SlabSeg{reinterpret_cast<const float*>(t[0]), reinterpret_cast<float*>(t[1]), t[3], (int)t[2], (int)t[4], (int)t[5],
                      t[6] ? 1 : 0};
  }
  DeviceGuard dg(device);
  return launch_slab_reduce_batch(segs, n, (hipStream_t)stream);
}

long long mmft_colsum_workspace_bytes(int rows, int cols) { return (long long)colsum_blocks(rows) * cols * 4; }

int mmft_colsum(const float* g, const int* idx, long long ld, int rows, int cols, float* out, int accumulate,
                float* workspace, long long workspace_bytes, int device, void* stream) {
  MMFT_REQUIRE(g && out, "colsum: null pointer");
  MMFT_REQUIRE(rows >= 0 && cols > 0 && ld >= cols, "colsum: bad sizes");
  DeviceGuard dg(device);
  hipStream_t st = (hipStream_t)stream;
  if (rows == 0) {
    if (!accumulate) (void)hipMemsetAsync(out, 0, (size_t)cols * 4, st);
    return MMFT_OK;
  }
  int nb = colsum_blocks(rows);
  MMFT_REQUIRE(workspace && workspace_bytes >= (long long)nb * cols * 4, "colsum: workspace too small");
  MMFT_LAUNCH("colsum_partial_kernel", 0.0, 4.0 * rows * cols, colsum_partial_kernel, dim3(cdiv(cols, 64), nb), dim3(256), st,
              g, idx, ld, rows, cols, workspace, colsum_strip(rows));
  int rc = check_launch("colsum_partial");
  if (rc) return rc;
  return launch_slab_reduce(workspace, nb, cols, out, accumulate, st);
}

int mmft_act_bwd(const float* dy, const float* y, float* dpre, long long n, int act, float slope, int device,
                 void* stream) {
  MMFT_REQUIRE(dy && y && dpre && n >= 0, "act_bwd: bad args");
  if (n == 0) return MMFT_OK;
  DeviceGuard dg(device);
  MMFT_LAUNCH("act_bwd_kernel", 0.0, 12.0 * n, act_bwd_kernel, dim3(ew_grid(n)), dim3(256), (hipStream_t)stream, dy, y, dpre, n, act, slope);
  return check_launch("act_bwd");
}

int mmft_act_fwd(const float* x, float* y, long long n, int act, float slope, int device, void* stream) {
  MMFT_REQUIRE(x && y && n >= 0, "act_fwd: bad args");
  if (n == 0) return MMFT_OK;
  DeviceGuard dg(device);
  hipLaunchKernelGGL(act_fwd_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, x, y, n, act, slope);
  return check_launch("act_fwd");
}

}  // extern "C"

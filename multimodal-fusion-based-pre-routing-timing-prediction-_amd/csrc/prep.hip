// Design preprocessing on the GPU: the graph-side steps the reference runs in Python on networkx
// before a design can be trained on.  Integer / index work, HBM- and latency-bound; no MFMA here.
//
//   levelize            longest-path levels from the primary inputs  (src/verilog_parser_asap7.py:1452-1517,
//                       cal_topo_level: frontier expansion, then reverse de-duplication = "keep the LAST level a
//                       node appears in"; unreachable nodes get -1 and are dropped by the caller)
//   trace_critical      one critical path per endpoint               (:1433-1450, find_critical_path)
//   path_mask_*         union of the drive-sink bounding boxes along a path, as CSR rows with ascending columns
//                       (:1302-1369, masking == 'critical')
//   minmax_normalize    per-column (a - min) / (max - min)           (src/train.py:309-318)
#include "common.h"
#include <math.h>

namespace mmft {

// ---------------------------------------------------------------------------------------------- levelization
// One frontier step over up to two out-edge CSRs (net and cell edges are kept apart by PinGraph).  All writers of
// next[v] / last[v] in a step store the same values, so the races are benign and the result is deterministic.
__global__ void __launch_bounds__(256) levelize_step_kernel(int n, const int* __restrict__ p0, const int* __restrict__ i0,
                                                            const int* __restrict__ p1, const int* __restrict__ i1,
                                                            unsigned char* __restrict__ cur,
                                                            unsigned char* __restrict__ next, int* __restrict__ last,
                                                            int it, int* __restrict__ active) {
  int u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= n || !cur[u]) return;
  cur[u] = 0;
  bool any = false;
  for (int e = p0[u]; e < p0[u + 1]; ++e) {
    int v = i0[e];
    next[v] = 1;
    last[v] = it;
    any = true;
  }
  if (p1)
    for (int e = p1[u]; e < p1[u + 1]; ++e) {
      int v = i1[e];
      next[v] = 1;
      last[v] = it;
      any = true;
    }
  if (any) *active = it;
}

__global__ void __launch_bounds__(256) levelize_init_kernel(int n, unsigned char* __restrict__ cur,
                                                            unsigned char* __restrict__ next, int* __restrict__ last) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    cur[i] = 0;
    next[i] = 0;
    last[i] = -1;
  }
}

__global__ void __launch_bounds__(256) levelize_seed_kernel(const int* __restrict__ pis, int npi,
                                                            unsigned char* __restrict__ cur, int* __restrict__ last) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < npi) {
    cur[pis[i]] = 1;
    last[pis[i]] = 0;
  }
}

// ---------------------------------------------------------------------------------------------- fan-in cone
// One step of the backward closure: every marked node of the given level marks its in-neighbours (SURVEY.md 8f-1:
// only the transitive fan-in of the sampled endpoints can influence their predictions).  All writers store 1.
__global__ void __launch_bounds__(256) cone_step_kernel(const int* __restrict__ rows, int row0, int n,
                                                        const int* __restrict__ p0, const int* __restrict__ i0,
                                                        const int* __restrict__ p1, const int* __restrict__ i1,
                                                        unsigned char* __restrict__ mark) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int v = rows ? rows[i] : row0 + i;
  if (!mark[v]) return;
  for (int e = p0[v]; e < p0[v + 1]; ++e) mark[i0[e]] = 1;
  if (p1)
    for (int e = p1[v]; e < p1[v + 1]; ++e) mark[i1[e]] = 1;
}

// ---------------------------------------------------------------------------------------------- critical paths
// One thread per endpoint walks back through the in-edges: at every step the FIRST predecessor (in CSR = insertion
// order, as networkx's predecessors()) that sits exactly one level below is taken; a predecessor flagged in `stop`
// (the reference's "'clk' in name" test) met before that ends the walk.
__global__ void __launch_bounds__(256) trace_critical_kernel(int npaths, const int* __restrict__ endpoints,
                                                             const int* __restrict__ p0, const int* __restrict__ i0,
                                                             const int* __restrict__ p1, const int* __restrict__ i1,
                                                             const int* __restrict__ level,
                                                             const unsigned char* __restrict__ stop, int maxlen,
                                                             int* __restrict__ paths, int* __restrict__ lens) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= npaths) return;
  int cur = endpoints[t];
  int lv = level[cur];
  int* out = paths + (long long)t * maxlen;
  int len = 0;
  if (len < maxlen) out[len] = cur;
  ++len;
  bool halted = false;
  while (lv >= 2 && !halted) {
    int pick = -1;
    for (int pass = 0; pass < 2 && pick < 0 && !halted; ++pass) {
      const int* ptr = pass == 0 ? p0 : p1;
      const int* idx = pass == 0 ? i0 : i1;
      if (!ptr) continue;
      for (int e = ptr[cur]; e < ptr[cur + 1]; ++e) {
        int nd = idx[e];
        if (level[nd] < 0) continue;       // not reachable from a primary input: the reference removed it (:1511-1513)
        if (stop && stop[nd]) {
          halted = true;
          break;
        }
        if (level[nd] == lv - 1) {
          pick = nd;
          break;
        }
      }
    }
    if (pick < 0) break;            // malformed levels (no predecessor one level below): stop instead of spinning
    if (len < maxlen) out[len] = pick;
    ++len;
    cur = pick;
    --lv;
  }
  lens[t] = len;                    // > maxlen tells the caller the row was truncated
  for (int i = len; i < maxlen; ++i) out[i] = -1;
}

// ---------------------------------------------------------------------------------------------- path masks
// One workgroup per path: the union of bounding boxes is built as a bitmap in LDS (P bits), so duplicates cost
// nothing and the set bits come out in ascending column order.  FILL = false: count only.
template <bool FILL>
__global__ void __launch_bounds__(256) path_mask_kernel(const int* __restrict__ paths, const int* __restrict__ lens,
                                                        int maxlen, const int* __restrict__ loc_x,
                                                        const int* __restrict__ loc_y, int map_x, int map_y,
                                                        int* __restrict__ counts, const int* __restrict__ indptr,
                                                        int* __restrict__ cols) {
  extern __shared__ unsigned bits[];
  __shared__ int wave_tot[4];
  const int P = map_x * map_y, words = (P + 31) >> 5;
  const int t = blockIdx.x, tid = threadIdx.x;
  for (int w = tid; w < words; w += 256) bits[w] = 0u;
  __syncthreads();
  const int* path = paths + (long long)t * maxlen;
  int len = lens[t] < maxlen ? lens[t] : maxlen;
  for (int j = 0; j + 1 < len; ++j) {
    int a = path[j], b = path[j + 1];
    int xa = loc_x[a], ya = loc_y[a], xb = loc_x[b], yb = loc_y[b];
    int x1 = xa < xb ? xa : xb, x2 = xa < xb ? xb : xa;
    int y1 = ya < yb ? ya : yb, y2 = ya < yb ? yb : ya;
    x1 = x1 < 0 ? 0 : x1;
    y1 = y1 < 0 ? 0 : y1;
    x2 = x2 >= map_x ? map_x - 1 : x2;
    y2 = y2 >= map_y ? map_y - 1 : y2;
    int w = y2 - y1 + 1, hgt = x2 - x1 + 1;
    if (w <= 0 || hgt <= 0) continue;
    // a box row is the contiguous bit range [x * map_y + y1, x * map_y + y2]: OR whole-word masks instead of one
    // LDS atomic per cell (boxes of long nets cover thousands of cells)
    const int wpr = ((w + 30) >> 5) + 1;                      // words a row's range can touch
    for (int c = tid; c < hgt * wpr; c += 256) {
      int row = c / wpr, k = c - row * wpr;
      int lo = (x1 + row) * map_y + y1, hi = lo + w - 1;      // inclusive bit range
      int word = (lo >> 5) + k;
      if (word > (hi >> 5)) continue;
      int b0 = word == (lo >> 5) ? (lo & 31) : 0;
      int b1 = word == (hi >> 5) ? (hi & 31) : 31;
      unsigned mask = (b1 == 31 ? 0xffffffffu : ((1u << (b1 + 1)) - 1u)) & ~((1u << b0) - 1u);
      atomicOr(&bits[word], mask);
    }
  }
  __syncthreads();
  // ordered emission: per-thread popcount over a contiguous run of words, then an exclusive scan over the block
  const int per = (words + 255) / 256;
  int mine = 0;
  for (int w = tid * per; w < (tid + 1) * per && w < words; ++w) mine += __popc(bits[w]);
  int lane = tid & 63, wave = tid >> 6, incl = mine;
  for (int o = 1; o < 64; o <<= 1) {
    int up = __shfl_up(incl, o, 64);
    if (lane >= o) incl += up;
  }
  if (lane == 63) wave_tot[wave] = incl;
  __syncthreads();
  int base = 0;
  for (int w = 0; w < wave; ++w) base += wave_tot[w];
  if (!FILL) {
    if (tid == 255) counts[t] = base + incl;
    return;
  }
  int pos = indptr[t] + base + incl - mine;
  for (int w = tid * per; w < (tid + 1) * per && w < words; ++w) {
    unsigned v = bits[w];
    while (v) {
      int b = __ffs(v) - 1;
      cols[pos++] = (w << 5) + b;
      v &= v - 1;
    }
  }
}

// ---------------------------------------------------------------------------------------------- min-max scaling
constexpr int MM_ROWS = 2048;
__global__ void __launch_bounds__(256) minmax_partial_kernel(const float* __restrict__ f, long long ld, int n, int c0, int C,
                                                             float* __restrict__ part) {
  // block (strip of rows) x column: 256 threads stride the strip; partial min / max per (strip, column)
  __shared__ float smin[256], smax[256];
  int col = c0 + blockIdx.y;
  int r0 = blockIdx.x * MM_ROWS, r1 = r0 + MM_ROWS < n ? r0 + MM_ROWS : n;
  float mn = INFINITY, mx = -INFINITY;
  bool nan = false;
  for (int r = r0 + threadIdx.x; r < r1; r += 256) {
    float v = f[(long long)r * ld + col];
    nan |= (v != v);
    mn = v < mn ? v : mn;
    mx = v > mx ? v : mx;
  }
  if (nan) mn = mx = NAN;                      // th.min / th.max propagate NaN
  smin[threadIdx.x] = mn;
  smax[threadIdx.x] = mx;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) {
      float a = smin[threadIdx.x], b = smin[threadIdx.x + o];
      smin[threadIdx.x] = (a != a || b != b) ? NAN : (b < a ? b : a);
      a = smax[threadIdx.x];
      b = smax[threadIdx.x + o];
      smax[threadIdx.x] = (a != a || b != b) ? NAN : (b > a ? b : a);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    part[((long long)blockIdx.x * (C - c0) + blockIdx.y) * 2] = smin[0];
    part[((long long)blockIdx.x * (C - c0) + blockIdx.y) * 2 + 1] = smax[0];
  }
}

__global__ void __launch_bounds__(64) minmax_final_kernel(const float* __restrict__ part, int strips, int ncol,
                                                          float* __restrict__ mm) {
  int c = blockIdx.x;
  float mn = INFINITY, mx = -INFINITY;
  bool nan = false;
  for (int s = threadIdx.x; s < strips; s += 64) {
    float a = part[((long long)s * ncol + c) * 2], b = part[((long long)s * ncol + c) * 2 + 1];
    nan |= (a != a) || (b != b);
    mn = a < mn ? a : mn;
    mx = b > mx ? b : mx;
  }
  for (int o = 32; o > 0; o >>= 1) {
    float a = __shfl_xor(mn, o, 64), b = __shfl_xor(mx, o, 64);
    int nn = __shfl_xor((int)nan, o, 64);
    nan |= (nn != 0);
    mn = a < mn ? a : mn;
    mx = b > mx ? b : mx;
  }
  if (threadIdx.x == 0) {
    mm[c * 2] = nan ? NAN : mn;
    mm[c * 2 + 1] = nan ? NAN : mx;
  }
}

__global__ void __launch_bounds__(256) minmax_apply_kernel(float* __restrict__ f, long long ld, int n, int c0, int C,
                                                           const float* __restrict__ mm) {
  long long total = (long long)n * (C - c0);
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
    int r = (int)(t / (C - c0)), c = (int)(t % (C - c0));
    float mn = mm[c * 2], mx = mm[c * 2 + 1];
    float* q = f + (long long)r * ld + c0 + c;
    *q = (*q - mn) / (mx - mn);              // same two roundings as the reference's (a - min_a) / (max_a - min_a)
  }
}

}  // namespace mmft

using namespace mmft;

extern "C" {

long long mmft_levelize_workspace_bytes(int n) { return 2ll * ((n + 255) / 256 * 256) + 256; }

int mmft_levelize(const int* out_indptr0, const int* out_indices0, const int* out_indptr1, const int* out_indices1,
                  int n, const int* pis, int npi, int* level, int* num_levels, void* workspace,
                  long long workspace_bytes, int device, void* stream) {
  MMFT_REQUIRE(out_indptr0 && level && num_levels, "levelize: null pointer");
  MMFT_REQUIRE((out_indptr1 == nullptr) == (out_indices1 == nullptr), "levelize: second CSR needs both arrays");
  MMFT_REQUIRE(n >= 0 && npi >= 0 && (pis || npi == 0), "levelize: bad sizes");
  MMFT_REQUIRE(workspace && workspace_bytes >= mmft_levelize_workspace_bytes(n), "levelize: workspace too small");
  *num_levels = 0;
  if (n == 0) return MMFT_OK;
  DeviceGuard dg(device);
  hipStream_t st = (hipStream_t)stream;
  const int npad = (n + 255) / 256 * 256;
  unsigned char* cur = (unsigned char*)workspace;
  unsigned char* next = cur + npad;
  int* active = (int*)(next + npad);                       // last step that expanded at least one edge
  const int nb = cdiv(n, 256);
  hipLaunchKernelGGL(levelize_init_kernel, dim3(nb), dim3(256), 0, st, n, cur, next, level);
  (void)hipMemsetAsync(active, 0, sizeof(int), st);
  if (npi > 0) hipLaunchKernelGGL(levelize_seed_kernel, dim3(cdiv(npi, 256)), dim3(256), 0, st, pis, npi, cur, level);
  // A DAG of n nodes has at most n levels; steps are issued in chunks and the host looks at `active` between chunks
  // (this entry point synchronises the stream - it is preprocessing, not part of the training step).
  int it = 1, host_active = 0;
  const int CHUNK = 32;
  while (it <= n) {
    for (int k = 0; k < CHUNK && it <= n; ++k, ++it) {
      hipLaunchKernelGGL(levelize_step_kernel, dim3(nb), dim3(256), 0, st, n, out_indptr0, out_indices0, out_indptr1,
                         out_indices1, cur, next, level, it, active);
      unsigned char* tmp = cur;
      cur = next;
      next = tmp;
    }
    if (hipMemcpyAsync(&host_active, active, sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess) {
      set_error("levelize: device error while reading the frontier flag: %s", hipGetErrorString(hipGetLastError()));
      return MMFT_ERR_LAUNCH;
    }
    if (host_active < it - CHUNK) break;                   // no edge expanded during the whole last chunk
  }
  int rc = check_launch("levelize");
  if (rc) return rc;
  if (it > n && host_active >= n) {
    set_error("levelize: the graph has a cycle reachable from the primary inputs");
    return MMFT_ERR_BAD_ARG;
  }
  *num_levels = host_active + 1;                           // levels 0 .. host_active
  return MMFT_OK;
}

int mmft_fanin_cone_step(const int* rows, int row0, int n, const int* in_indptr0, const int* in_indices0,
                         const int* in_indptr1, const int* in_indices1, unsigned char* mark, int device, void* stream) {
  MMFT_REQUIRE(n >= 0 && row0 >= 0, "fanin_cone_step: bad sizes");
  if (n == 0) return MMFT_OK;
  MMFT_REQUIRE(in_indptr0 && mark, "fanin_cone_step: null pointer");
  MMFT_REQUIRE((in_indptr1 == nullptr) == (in_indices1 == nullptr), "fanin_cone_step: second CSR needs both arrays");
  DeviceGuard dg(device);
  hipLaunchKernelGGL(cone_step_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, rows, row0, n, in_indptr0,
                     in_indices0, in_indptr1, in_indices1, mark);
  return check_launch("fanin_cone_step");
}

int mmft_trace_critical_paths(const int* in_indptr0, const int* in_indices0, const int* in_indptr1,
                              const int* in_indices1, const int* level, const unsigned char* stop,
                              const int* endpoints, int npaths, int maxlen, int* paths, int* lens, int device,
                              void* stream) {
  MMFT_REQUIRE(npaths >= 0 && maxlen >= 1, "trace_critical_paths: bad sizes");
  if (npaths == 0) return MMFT_OK;
  MMFT_REQUIRE(in_indptr0 && level && paths && lens && endpoints, "trace_critical_paths: null pointer");
  MMFT_REQUIRE((in_indptr1 == nullptr) == (in_indices1 == nullptr), "trace_critical_paths: second CSR needs both arrays");
  DeviceGuard dg(device);
  hipLaunchKernelGGL(trace_critical_kernel, dim3(cdiv(npaths, 256)), dim3(256), 0, (hipStream_t)stream, npaths, endpoints,
                     in_indptr0, in_indices0, in_indptr1, in_indices1, level, stop, maxlen, paths, lens);
  return check_launch("trace_critical_paths");
}

static int path_mask_common(bool fill, const int* paths, const int* lens, int npaths, int maxlen, const int* loc_x,
                            const int* loc_y, int map_x, int map_y, int* counts, const int* indptr, int* cols,
                            int device, void* stream) {
  MMFT_REQUIRE(npaths >= 0 && maxlen >= 1 && map_x > 0 && map_y > 0, "path_mask: bad sizes");
  if (npaths == 0) return MMFT_OK;
  MMFT_REQUIRE(paths && lens && loc_x && loc_y, "path_mask: null pointer");
  long long P = (long long)map_x * map_y;
  MMFT_REQUIRE(P <= (1ll << 20), "path_mask: map larger than 2^20 cells (the LDS bitmap holds one bit per cell)");
  if (npaths == 0) return MMFT_OK;
  DeviceGuard dg(device);
  size_t lds = (size_t)((P + 31) / 32) * 4;
  static DynLdsOnce once[2];
  int arc = ensure_dyn_lds(once[fill ? 1 : 0], fill ? (const void*)path_mask_kernel<true> : (const void*)path_mask_kernel<false>,
                           1 << 17, "path_mask");
  if (arc) return arc;
  if (fill)
    hipLaunchKernelGGL(path_mask_kernel<true>, dim3(npaths), dim3(256), lds, (hipStream_t)stream, paths, lens, maxlen,
                       loc_x, loc_y, map_x, map_y, counts, indptr, cols);
  else
    hipLaunchKernelGGL(path_mask_kernel<false>, dim3(npaths), dim3(256), lds, (hipStream_t)stream, paths, lens, maxlen,
                       loc_x, loc_y, map_x, map_y, counts, indptr, cols);
  return check_launch("path_mask");
}

int mmft_path_mask_count(const int* paths, const int* lens, int npaths, int maxlen, const int* loc_x, const int* loc_y,
                         int map_x, int map_y, int* counts, int device, void* stream) {
  MMFT_REQUIRE(counts || npaths == 0, "path_mask_count: null pointer");
  return path_mask_common(false, paths, lens, npaths, maxlen, loc_x, loc_y, map_x, map_y, counts, nullptr, nullptr,
                          device, stream);
}

int mmft_path_mask_fill(const int* paths, const int* lens, int npaths, int maxlen, const int* loc_x, const int* loc_y,
                        int map_x, int map_y, const int* indptr, int* cols, int device, void* stream) {
  MMFT_REQUIRE((indptr && cols) || npaths == 0, "path_mask_fill: null pointer");
  return path_mask_common(true, paths, lens, npaths, maxlen, loc_x, loc_y, map_x, map_y, nullptr, indptr, cols, device,
                          stream);
}

long long mmft_minmax_workspace_bytes(int n, int ncol) {
  return ((long long)cdiv(n > 0 ? n : 1, MM_ROWS) * ncol * 2 + (long long)ncol * 2) * 4;
}

int mmft_minmax_normalize(float* feat, long long ld, int n, int C, int start_col, float* workspace,
                          long long workspace_bytes, int device, void* stream) {
  MMFT_REQUIRE(feat && n >= 0 && C > 0 && start_col >= 0 && start_col <= C && ld >= C, "minmax_normalize: bad arguments");
  int ncol = C - start_col;
  if (n == 0 || ncol == 0) return MMFT_OK;
  MMFT_REQUIRE(workspace && workspace_bytes >= mmft_minmax_workspace_bytes(n, ncol), "minmax_normalize: workspace too small");
  DeviceGuard dg(device);
  hipStream_t st = (hipStream_t)stream;
  int strips = cdiv(n, MM_ROWS);
  float* part = workspace;
  float* mm = workspace + (long long)strips * ncol * 2;
  hipLaunchKernelGGL(minmax_partial_kernel, dim3(strips, ncol), dim3(256), 0, st, feat, ld, n, start_col, C, part);
  hipLaunchKernelGGL(minmax_final_kernel, dim3(ncol), dim3(64), 0, st, part, strips, ncol, mm);
  long long total = (long long)n * ncol;
  int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(minmax_apply_kernel, dim3(grid), dim3(256), 0, st, feat, ld, n, start_col, C, mm);
  return check_launch("minmax_normalize");
}

}  // extern "C"

// Persistent forward sweep: ONE launch walks every topological level of the netlist (levels 1..L-1; level 0 and
// the *_self MLPs are pre-filled by batched GEMMs).  Replaces the per-level launches of PathConv.forward
// (reference src/model.py:158-213, called L times per mini-batch from src/train.py:490-511).
//
// Why: a level of a 64k-node design holds ~1k nodes (8k when 8 designs are merged); its whole-chip MFMA time is
// ~7 us and its HBM time ~3 us, but every separate launch costs 8-16 us of launch + dependent-load latency, and
// the fused level kernel re-reads its 256 KB of weights from L2 per level.  Here every workgroup loads the
// fc_cell_neigh weights into registers ONCE for the whole sweep (64 x 16-byte loads per thread), and the levels
// are separated by a grid barrier instead of a kernel boundary.
//
// Grid barrier (MI355X_MICROARCH.md, "barrier-counter"; cdna_hip_programming.md Guideline 16): per-XCD L2s are
// not coherent and a CU's L1 is never refreshed by other CUs' stores, so each barrier is
//   h rows stored write-through (sc1) -> every wave: s_waitcnt vmcnt(0) -> __syncthreads -> lane 0: relaxed agent
//   atomic add on a monotonic counter (no release fence needed for write-through payloads, recipe R1) -> relaxed sc1 poll with s_sleep (BOUNDED) -> agent-scope
//   ACQUIRE fence + s_waitcnt vmcnt(0) -> __syncthreads -> plain loads.
// The grid is sized to be co-resident (<= 1 workgroup per CU, 256 CUs); the spin is bounded so that a placement
// surprise ends in an error flag, never in a hang.  The counter is zeroed by a memset node ahead of the launch.
#include "mlp2_core.h"

namespace mmft {

struct SweepFwdArgs {
  float* h;            // [N][128], rows pre-filled with the *_self MLP outputs; updated in place
  float* A;            // [N][128]
  float* LSE;          // [N][128]
  float* HN;           // [N][256]
  const int* in_net_ptr;
  const int* in_net_idx;
  const int* in_cell_ptr;
  const int* in_cell_idx;
  const int* lvl_ptr;  // [L+1] offsets into lvl_rows
  const int* lvl_rows; // node ids of all levels, level-major
  int L;
  const float* w1;     // fc_cell_neigh.layers.0  [256][128]
  const float* b1;
  const float* w2;     // fc_cell_neigh.layers.2  [128][256]
  const float* b2;
  int relu;
  unsigned* counter;   // barrier counter (zeroed before the launch)
  int* error;          // set to 1 when a barrier spin runs out
};

// h rows are the only bytes another workgroup reads inside the launch.  They are stored WRITE-THROUGH (two 8-byte
// relaxed agent-scope stores = global_store_dwordx2 sc1), so no release fence (an L2 write-back of everything this
// XCD dirtied, A / LSE / HN included) is needed before the barrier: Guideline 16 recipe R1.
__device__ __forceinline__ void store_h_wt(float* p, f32x4 v) {
  typedef __attribute__((address_space(1))) unsigned long long gu64;
  unsigned long long lo = ((unsigned long long)__float_as_uint(v.y) << 32) | __float_as_uint(v.x);
  unsigned long long hi = ((unsigned long long)__float_as_uint(v.w) << 32) | __float_as_uint(v.z);
  __hip_atomic_store((gu64*)p, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store((gu64*)(p + 2), hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

constexpr long long SPIN_LIMIT = 1ll << 22;   // x s_sleep(4) ~ a fraction of a second: far above any real wait

__device__ __forceinline__ bool grid_barrier(unsigned* counter, unsigned target, int* error) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  __shared__ int ok;
  if (threadIdx.x == 0) {
    // every storing wave drained its write-through stores above (vmcnt(0)) and reached the barrier: arrive
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    long long spins = 0;
    int good = 1;
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(4);
      if (++spins > SPIN_LIMIT) {
        good = 0;
        break;
      }
    }
    if (!good) __hip_atomic_store(error, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ok = good;
  }
  __syncthreads();
  return ok != 0;
}

__global__ void __launch_bounds__(256, 1) sweep_fwd_persistent_kernel(SweepFwdArgs a) {
  constexpr int XS = M2_K1 + 8, HS = M2_HD + 8;
  constexpr int W1SZ = M2_HD * (M2_BK + 8), W2SZ = M2_D2 * (M2_BK + 8);
  constexpr int WSZ = W1SZ > W2SZ ? W1SZ : W2SZ;
  __shared__ __attribute__((aligned(16))) float lds[M2_BM * XS + M2_BM * HS + WSZ];   // 91 KB: single weight buffer
  float* xs = lds;
  float* hs = lds + M2_BM * XS;
  float* wb = hs + M2_BM * HS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = tid >> 3, c0 = (tid & 7) * 16;      // gather mapping: 8 threads per row, 16 channels each

  WPanel<false, M2_HD, M2_K1 / M2_BK> p1;
  WPanel<false, M2_D2, M2_HD / M2_BK> p2;
  p1.load(a.w1, 128, tid);
  p2.load(a.w2, 256, tid);

  for (int level = 1; level < a.L; ++level) {
    const int* rows = a.lvl_rows + a.lvl_ptr[level];
    const int n = a.lvl_ptr[level + 1] - a.lvl_ptr[level];
    for (int m0 = blockIdx.x * M2_BM; m0 < n; m0 += gridDim.x * M2_BM) {
      const bool live = m0 + r < n;
      const long long v = live ? (long long)rows[m0 + r] : 0;
      if (level & 1) {
        // net level: h[v] = act(h[v] + mean_{u->v} h[u])      (src/model.py:186-187,103-111)
        if (live) {
          f32x4 o[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) o[q] = f32x4{0.f, 0.f, 0.f, 0.f};
          int e0 = a.in_net_ptr[v], e1 = a.in_net_ptr[v + 1];
          for (int e = e0; e < e1; ++e) {
            const float* src = a.h + (long long)a.in_net_idx[e] * 128 + c0;
#pragma unroll
            for (int q = 0; q < 4; ++q) o[q] += *reinterpret_cast<const f32x4*>(src + q * 4);
          }
          float inv = e1 > e0 ? 1.0f / (float)(e1 - e0) : 0.f;
          float* dst = a.h + v * 128 + c0;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            f32x4 t = o[q] * inv + *reinterpret_cast<const f32x4*>(dst + q * 4);
            if (a.relu) {
              t.x = t.x > 0.f ? t.x : 0.f; t.y = t.y > 0.f ? t.y : 0.f;
              t.z = t.z > 0.f ? t.z : 0.f; t.w = t.w > 0.f ? t.w : 0.f;
            }
            store_h_wt(dst + q * 4, t);
          }
        }
        continue;
      }
      // ---- cell level: fan-in softmax-sum (src/model.py:113-116) -> A tile in LDS (+ A, LSE in HBM)
      {
        f32x4 o[4], mx[4], sm[4], lse[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          o[q] = f32x4{0.f, 0.f, 0.f, 0.f};
          mx[q] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
          sm[q] = f32x4{0.f, 0.f, 0.f, 0.f};
          lse[q] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        int e0 = 0, e1 = 0;
        if (live) {
          e0 = a.in_cell_ptr[v];
          e1 = a.in_cell_ptr[v + 1];
        }
        for (int e = e0; e < e1; ++e) {
          const float* src = a.h + (long long)a.in_cell_idx[e] * 128 + c0;
          f32x4 x[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) x[q] = *reinterpret_cast<const f32x4*>(src + q * 4);
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              float mn = fmaxf(mx[q][j], x[q][j]);
              float sc = expf(mx[q][j] - mn), pe = expf(x[q][j] - mn);
              sm[q][j] = sm[q][j] * sc + pe;
              o[q][j] = o[q][j] * sc + pe * x[q][j];
              mx[q][j] = mn;
            }
        }
        if (e1 > e0) {
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              o[q][j] = o[q][j] / sm[q][j];
              lse[q][j] = mx[q][j] + logf(sm[q][j]);
            }
        }
        if (live) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            *reinterpret_cast<f32x4*>(a.A + v * 128 + c0 + q * 4) = o[q];
            *reinterpret_cast<f32x4*>(a.LSE + v * 128 + c0 + q * 4) = lse[q];
          }
        }
        __syncthreads();      // previous tile's readers of xs / hs are done
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4*>(xs + r * XS + c0 + q * 4) = o[q];
      }
      // ---- HN = relu(A W1^T + b1) kept in LDS (and stored), h = act(h + HN W2^T + b2)
      f32x4 acc1[2][4];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc1[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      Phase1<false, 0, M2_K1 / M2_BK, true>::run(p1, xs, wb, 0, tid, lane, wave, acc1);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          int m = i * 16 + (lane & 15);
          int nn = wave * 64 + j * 16 + (lane >> 4) * 4;
          f32x4 t = acc1[i][j];
          t.x += a.b1[nn]; t.y += a.b1[nn + 1]; t.z += a.b1[nn + 2]; t.w += a.b1[nn + 3];
          t.x = t.x > 0.f ? t.x : 0.f; t.y = t.y > 0.f ? t.y : 0.f;
          t.z = t.z > 0.f ? t.z : 0.f; t.w = t.w > 0.f ? t.w : 0.f;
          *reinterpret_cast<f32x4*>(hs + m * HS + nn) = t;
          if (m0 + m < n) *reinterpret_cast<f32x4*>(a.HN + (long long)rows[m0 + m] * 256 + nn) = t;
        }
      __syncthreads();        // hidden tile complete, last W1 tile no longer read
      f32x4 acc2[2][2];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc2[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      Phase2<false, 0, M2_HD / M2_BK, true>::run(p2, hs, wb, 0, tid, lane, wave, acc2);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          int m = i * 16 + (lane & 15);
          if (m0 + m >= n) continue;
          int nn = wave * 32 + j * 16 + (lane >> 4) * 4;
          float* q = a.h + (long long)rows[m0 + m] * 128 + nn;
          f32x4 t = acc2[i][j] + *reinterpret_cast<const f32x4*>(q);
          t.x += a.b2[nn]; t.y += a.b2[nn + 1]; t.z += a.b2[nn + 2]; t.w += a.b2[nn + 3];
          if (a.relu) {
            t.x = t.x > 0.f ? t.x : 0.f; t.y = t.y > 0.f ? t.y : 0.f;
            t.z = t.z > 0.f ? t.z : 0.f; t.w = t.w > 0.f ? t.w : 0.f;
          }
          store_h_wt(q, t);
        }
      __syncthreads();        // wb / hs free for the next tile
    }
    if (level + 1 < a.L) {
      if (!grid_barrier(a.counter, (unsigned)level * gridDim.x, a.error)) return;   // bounded: never hangs
    }
  }
}

}  // namespace mmft

using namespace mmft;

extern "C" int mmft_sweep_fwd_persistent(float* h, float* A, float* LSE, float* HN, const int* in_net_indptr,
                                         const int* in_net_indices, const int* in_cell_indptr, const int* in_cell_indices,
                                         const int* level_ptr, const int* level_rows, int L, const float* w1, const float* b1,
                                         const float* w2, const float* b2, int relu, int D, int HD, int max_level_rows,
                                         unsigned* counter, int* error_flag, int device, void* stream) {
  MMFT_REQUIRE(h && A && LSE && HN && in_net_indptr && in_cell_indptr && level_ptr && level_rows && w1 && b1 && w2 && b2 &&
                   counter && error_flag,
               "sweep_fwd_persistent: null pointer");
  if (D != M2_K1 || HD != M2_HD) {
    set_error("sweep_fwd_persistent: only D=128, hidden=256 (got %d, %d)", D, HD);
    return MMFT_ERR_UNSUPPORTED;
  }
  MMFT_REQUIRE(L >= 1 && max_level_rows >= 0, "sweep_fwd_persistent: bad sizes");
  MMFT_REQUIRE(aligned16(h) && aligned16(A) && aligned16(LSE) && aligned16(HN) && aligned16(w1) && aligned16(w2),
               "sweep_fwd_persistent: operands must be 16-byte aligned");
  if (L == 1) return MMFT_OK;
  DeviceGuard dg(device);
  hipStream_t st = (hipStream_t)stream;
  // co-resident grid: one workgroup per CU at most (91 KB LDS, 1 wave per SIMD), never more tiles than needed
  int grid = cdiv(max_level_rows > 0 ? max_level_rows : 1, M2_BM);
  if (grid > 256) grid = 256;
  if (grid < 1) grid = 1;
  (void)hipMemsetAsync(counter, 0, sizeof(unsigned), st);
  (void)hipMemsetAsync(error_flag, 0, sizeof(int), st);
  SweepFwdArgs a{h, A, LSE, HN, in_net_indptr, in_net_indices, in_cell_indptr, in_cell_indices, level_ptr, level_rows, L,
                 w1, b1, w2, b2, relu, counter, error_flag};
  MMFT_LAUNCH("sweep_fwd_persistent_kernel", 0.0, 0.0, sweep_fwd_persistent_kernel, dim3(grid), dim3(256), st, a);
  return check_launch("sweep_fwd_persistent");
}

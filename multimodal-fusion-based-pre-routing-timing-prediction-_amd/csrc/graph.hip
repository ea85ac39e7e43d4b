// Netlist-graph aggregation kernels (HBM-bound gather / segmented reductions over CSR).
// Replaces DGL graph.pull with fn.copy_src+fn.mean (reference src/model.py:186-187), the
// degree-bucketed UDF PathConv.cell_msg_reduce (src/model.py:113-116,202-204), the activation
// write-back and the target gather (src/model.py:206-213), and their autograd mirror.
//
// Layout: node features are [N][D] fp32 rows (D % 4 == 0).  One thread owns one float4 channel group
// of one node, so the D/4 threads of a node read each neighbour row as one contiguous 16-B-per-lane
// segment (512 B for D = 128) and no cross-lane reduction is needed; a 256-thread workgroup covers
// 256/(D/4) nodes (8 for D = 128).  Degrees on this path are small (net in-degree 1, cell fan-in
// ~2), so the per-thread edge loop is short; heavy-tailed out-degrees in the reverse sweep are the
// known skew.
#include "common.h"
#include "fold_gather.h"

namespace mmft {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }

// node/channel-group assignment shared by all kernels here
#define MMFT_NODE_LOOP(n, D)                                                  \
  const int groups = (D) >> 2;                                                \
  const long long total = (long long)(n) * groups;                            \
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; \
       t += (long long)gridDim.x * blockDim.x)

__global__ void __launch_bounds__(256) seg_softmax_sum_fwd_kernel(const float* __restrict__ h, long long ldh,
                                                                  const int* __restrict__ indptr,
                                                                  const int* __restrict__ indices,
                                                                  const int* __restrict__ rows, int row0, int n,
                                                                  int D, float* __restrict__ A,
                                                                  float* __restrict__ LSE, long long lda) {
  MMFT_NODE_LOOP(n, D) {
    int i = (int)(t / groups), c = (int)(t - (long long)i * groups) * 4;
    int v = rows ? rows[i] : row0 + i;
    int e0 = indptr[v], e1 = indptr[v + 1];
    f32x4 mx = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    f32x4 s = {0.f, 0.f, 0.f, 0.f}, acc = {0.f, 0.f, 0.f, 0.f};
    for (int e = e0; e < e1; ++e) {
      f32x4 x = ld4(h + (long long)indices[e] * ldh + c);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float m_new = fmaxf(mx[j], x[j]);
        float scale = expf(mx[j] - m_new);   // exp(-inf) = 0 on the first edge
        float p = expf(x[j] - m_new);
        s[j] = s[j] * scale + p;
        acc[j] = acc[j] * scale + p * x[j];
        mx[j] = m_new;
      }
    }
    f32x4 a = {0.f, 0.f, 0.f, 0.f}, l = {0.f, 0.f, 0.f, 0.f};
    if (e1 > e0) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        a[j] = acc[j] / s[j];
        l[j] = mx[j] + logf(s[j]);
      }
    }
    st4(A + (long long)v * lda + c, a);
    if (LSE) st4(LSE + (long long)v * lda + c, l);
  }
}

__global__ void __launch_bounds__(256) seg_mean_fwd_kernel(const float* __restrict__ src, long long lds,
                                                           const int* __restrict__ indptr,
                                                           const int* __restrict__ indices,
                                                           const int* __restrict__ rows, int row0, int n, int D,
                                                           float* __restrict__ out, long long ldo, int add_self,
                                                           int relu, int do_mean) {
  MMFT_NODE_LOOP(n, D) {
    int i = (int)(t / groups), c = (int)(t - (long long)i * groups) * 4;
    int v = rows ? rows[i] : row0 + i;
    int e0 = indptr[v], e1 = indptr[v + 1];
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int e = e0; e < e1; ++e) acc += ld4(src + (long long)indices[e] * lds + c);
    if (do_mean && e1 > e0) acc = acc * (1.0f / (float)(e1 - e0));
    float* o = add_self ? out + (long long)v * ldo + c : out + (long long)i * ldo + c;
    if (add_self) acc += ld4(o);
    if (relu) {
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = acc[j] > 0.f ? acc[j] : 0.f;
    }
    st4(o, acc);
  }
}

// FAST: the hardware exponential in the cell-consumer term (bf16 math mode, see fold_gather.h)
template <bool FAST>
__global__ void __launch_bounds__(256) MMFT_NO_PACKED_F32 level_bwd_pull_kernel(
    float* __restrict__ G, const float* __restrict__ h, long long ld, const int* __restrict__ rows, int row0, int n,
    int D, const int* __restrict__ on_ptr, const int* __restrict__ on_idx, const float* __restrict__ on_w,
    const int* __restrict__ oc_ptr, const int* __restrict__ oc_idx, const float* __restrict__ A,
    const float* __restrict__ LSE, const float* __restrict__ DA, int relu, const unsigned char* __restrict__ own,
    const int* __restrict__ heavy, int nheavy, int light_blocks, int heavy_thresh,
    const unsigned char* __restrict__ active) {
  // active (optional): per-node flags of the fan-in cone of this step's endpoints; rows outside it are skipped (their G
  // stays at the zero the caller filled in) and cell consumers outside it are not read (their A / LSE rows are stale).
  // Heavy rows (drivers of clock / reset-like nets, SRAM pins: out-degree in the hundreds) get a whole workgroup each:
  // its thread groups stride over the row's out-edges and the partial sums are combined through LDS in a fixed order
  // (bitwise reproducible).  One 32-lane group walking such a row in series is what set the duration of a level's
  // launch (the light rows finish in a fraction of it).
  __shared__ f32x4 part[8][64];
  if ((int)blockIdx.x >= light_blocks) {
    const int groups = D >> 2, tgs = 256 / groups, act = tgs < 8 ? tgs : 8;
    const int tg = threadIdx.x / groups, c = (threadIdx.x - tg * groups) * 4;
    for (int hi = (int)blockIdx.x - light_blocks; hi < nheavy; hi += (int)gridDim.x - light_blocks) {
      const int v = heavy[hi];
      if (active && !active[v]) continue;                // block-uniform
      const long long off = (long long)v * ld + c;
      const f32x4 hv = ld4(h + off);
      if (tg < act) {
        f32x4 g = {0.f, 0.f, 0.f, 0.f};
        const int e1 = on_ptr[v + 1];
        int e = on_ptr[v] + tg;
        for (; e + 3 * act < e1; e += 4 * act) {           // four edges of this group in flight
          int w0 = on_idx[e], w1 = on_idx[e + act], w2 = on_idx[e + 2 * act], w3 = on_idx[e + 3 * act];
          float s0 = on_w[e], s1 = on_w[e + act], s2 = on_w[e + 2 * act], s3 = on_w[e + 3 * act];
          f32x4 r0 = ld4(G + (long long)w0 * ld + c), r1 = ld4(G + (long long)w1 * ld + c);
          f32x4 r2 = ld4(G + (long long)w2 * ld + c), r3 = ld4(G + (long long)w3 * ld + c);
          g += r0 * s0;
          g += r1 * s1;
          g += r2 * s2;
          g += r3 * s3;
        }
        for (; e < e1; e += act) g += ld4(G + (long long)on_idx[e] * ld + c) * on_w[e];
        const int c1 = oc_ptr[v + 1];
        for (e = oc_ptr[v] + tg; e < c1; e += act) {
          if (active && !active[oc_idx[e]]) continue;
          long long wo = (long long)oc_idx[e] * ld + c;
          f32x4 da = ld4(DA + wo), a = ld4(A + wo), l = ld4(LSE + wo);
#pragma unroll
          for (int j = 0; j < 4; ++j) g[j] = cell_consumer_term<FAST>(g[j], da[j], hv[j], l[j], a[j]);
        }
        part[tg][c >> 2] = g;
      }
      __syncthreads();
      if (tg == 0) {
        f32x4 s = (!own || own[v]) ? ld4(G + off) : f32x4{0.f, 0.f, 0.f, 0.f};
        for (int t = 0; t < act; ++t) s += part[t][c >> 2];           // fixed order
        if (relu) {
#pragma unroll
          for (int j = 0; j < 4; ++j) s[j] = hv[j] > 0.f ? s[j] : 0.f;
        }
        st4(G + off, s);
      }
      __syncthreads();
    }
    return;
  }
  const int groups = (D) >> 2;
  const long long total = (long long)(n)*groups;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)light_blocks * blockDim.x) {
    int i = (int)(t / groups), c = (int)(t - (long long)i * groups) * 4;
    int v = rows ? rows[i] : row0 + i;
    if (heavy && (on_ptr[v + 1] - on_ptr[v]) + (oc_ptr[v + 1] - oc_ptr[v]) > heavy_thresh) continue;
    if (active && !active[v]) continue;
    long long off = (long long)v * ld + c;
    // both edge ranges are requested up front: a node has net OR cell consumers, and the second pointer pair would
    // otherwise start its own dependent chain (pointer -> index -> row) only after the first loop
    int e = on_ptr[v], e1 = on_ptr[v + 1];
    const int c0 = oc_ptr[v], c1 = oc_ptr[v + 1];
    f32x4 hv = ld4(h + off);
    // own-row gradient: present only where a sampled endpoint put one (own == NULL: G was zero-filled by the caller)
    f32x4 g = (!own || own[v]) ? ld4(G + off) : f32x4{0.f, 0.f, 0.f, 0.f};
    // net consumers: d/dh[v] of the mean over the consumer's in-edges (weight = 1/indeg, per out-edge, static).
    // Four edges are issued together so that the dependent index -> row loads of different edges overlap.
    for (; e + 4 <= e1; e += 4) {
      int w0 = on_idx[e], w1 = on_idx[e + 1], w2 = on_idx[e + 2], w3 = on_idx[e + 3];
      float s0 = on_w[e], s1 = on_w[e + 1], s2 = on_w[e + 2], s3 = on_w[e + 3];
      f32x4 r0 = ld4(G + (long long)w0 * ld + c), r1 = ld4(G + (long long)w1 * ld + c);
      f32x4 r2 = ld4(G + (long long)w2 * ld + c), r3 = ld4(G + (long long)w3 * ld + c);
      g += r0 * s0;
      g += r1 * s1;
      g += r2 * s2;
      g += r3 * s3;
    }
    for (; e < e1; ++e) g += ld4(G + (long long)on_idx[e] * ld + c) * on_w[e];
    // cell consumers: d a_c / d m_jc = w_jc (1 + m_jc - a_c),  w_jc = exp(m_jc - LSE_c)
    e = c0;
    e1 = c1;
    for (; e + 2 <= e1; e += 2) {
      const int w0 = oc_idx[e], w1 = oc_idx[e + 1];
      long long o0 = (long long)w0 * ld + c, o1 = (long long)w1 * ld + c;
      f32x4 da0 = ld4(DA + o0), a0 = ld4(A + o0), l0 = ld4(LSE + o0);
      f32x4 da1 = ld4(DA + o1), a1 = ld4(A + o1), l1 = ld4(LSE + o1);
      const bool k0 = !active || active[w0], k1 = !active || active[w1];    // outside the cone: contributes exactly zero
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float t0 = cell_consumer_term<FAST>(g[j], da0[j], hv[j], l0[j], a0[j]);
        g[j] = k0 ? t0 : g[j];
        const float t1 = cell_consumer_term<FAST>(g[j], da1[j], hv[j], l1[j], a1[j]);
        g[j] = k1 ? t1 : g[j];
      }
    }
    for (; e < e1; ++e) {
      const int w = oc_idx[e];
      if (active && !active[w]) continue;
      long long wo = (long long)w * ld + c;
      f32x4 da = ld4(DA + wo), a = ld4(A + wo), l = ld4(LSE + wo);
#pragma unroll
      for (int j = 0; j < 4; ++j) g[j] = cell_consumer_term<FAST>(g[j], da[j], hv[j], l[j], a[j]);
    }
    if (relu) {
#pragma unroll
      for (int j = 0; j < 4; ++j) g[j] = hv[j] > 0.f ? g[j] : 0.f;
    }
    st4(G + off, g);
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Folded forward sweep: ONE gather launch per (net level l - 1, cell level l) pair.
//   part A, net rows u of level l - 1:  h[u] = act(PRE[u] + mean_{d -> u} h[d])            (src/model.py:186-187,103-111)
//   part B, cell rows v of level l:     A[v], LSE[v] = softmax-weighted sum over the in-neighbours (src/model.py:113-116),
//           where an in-neighbour u of level l - 1 is NOT read from h (part A of this same launch is writing it) but
//           recomputed from PRE[u] and its driver's row with the same instruction sequence - bitwise the value part A
//           stores.  PRE holds fc_net_self(x_net) of the net rows (a buffer of its own, so that the in-place update of
//           h cannot race with part B).  The net level must be the contiguous id range net_row0 .. net_row0 + n_net - 1.
// Cell rows with more than `heavy_thresh` in-edges (SRAM macros, config E's Zipf fan-in) get a whole workgroup: eight
// thread groups stride over the edges with an online softmax each and their (max, sum, weighted sum) triples are merged
// in a fixed order.
// ------------------------------------------------------------------------------------------------------------------
template <bool FAST>
__global__ void __launch_bounds__(256) pair_fwd_gather_kernel(
    float* __restrict__ h, const float* __restrict__ PRE, long long ld, int D, const int* __restrict__ in_ptr,
    const int* __restrict__ in_idx, const int* __restrict__ ic_ptr, const int* __restrict__ ic_idx, int net_row0, int n_net,
    const int* __restrict__ rows, int cell_row0, int n_cell, float* __restrict__ A, float* __restrict__ LSE, long long lda,
    int relu, const int* __restrict__ heavy, int nheavy, int light_blocks, int heavy_thresh,
    const unsigned char* __restrict__ active, const int* __restrict__ ic_drv) {
  __shared__ f32x4 pm[8][64], ps[8][64], pa[8][64];
  const int groups = D >> 2;
  const FoldSrc fs{h, PRE, ld, in_ptr, in_idx, ic_idx, ic_drv, net_row0, n_net, relu};
  if ((int)blockIdx.x >= light_blocks) {
    const int tgs = 256 / groups, act = tgs < 8 ? tgs : 8;
    const int tg = threadIdx.x / groups, c = (threadIdx.x - tg * groups) * 4;
    for (int hi = (int)blockIdx.x - light_blocks; hi < nheavy; hi += (int)gridDim.x - light_blocks) {
      const int v = heavy[hi];
      if (active && !active[v]) continue;                    // block-uniform: outside this step's fan-in cone
      const int e0 = ic_ptr[v], e1 = ic_ptr[v + 1];
      if (tg < act) {
        SoftAccT<FAST> sa;
        sa.init();
        fold_gather_edges(fs, e0 + tg, e1, act, c, sa);
        pm[tg][c >> 2] = sa.mx; ps[tg][c >> 2] = sa.s; pa[tg][c >> 2] = sa.acc;
      }
      __syncthreads();
      if (tg == 0) {
        f32x4 M = pm[0][c >> 2];
        for (int t = 1; t < act; ++t)
#pragma unroll
          for (int j = 0; j < 4; ++j) M[j] = fmaxf(M[j], pm[t][c >> 2][j]);
        f32x4 S = {0.f, 0.f, 0.f, 0.f}, AC = {0.f, 0.f, 0.f, 0.f};
        for (int t = 0; t < act; ++t)                        // fixed order; empty partials carry max = -inf, sum = 0
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float sc = ps[t][c >> 2][j] > 0.f ? fg_exp<FAST>(pm[t][c >> 2][j] - M[j]) : 0.f;
            S[j] += ps[t][c >> 2][j] * sc;
            AC[j] += pa[t][c >> 2][j] * sc;
          }
        f32x4 a, l;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          a[j] = AC[j] / S[j];
          l[j] = M[j] + fg_log<FAST>(S[j]);
        }
        st4(A + (long long)v * lda + c, a);
        if (LSE) st4(LSE + (long long)v * lda + c, l);
      }
      __syncthreads();
    }
    return;
  }
  const long long total = (long long)(n_net + n_cell) * groups;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)light_blocks * blockDim.x) {
    int i = (int)(t / groups), c = (int)(t - (long long)i * groups) * 4;
    if (i < n_net) {                                         // part A
      const int u = net_row0 + i;
      if (active && !active[u]) continue;
      st4(h + (long long)u * ld + c, fold_net_value(fs, u, c));
      continue;
    }
    i -= n_net;                                              // part B
    const int v = rows ? rows[i] : cell_row0 + i;
    const int e0 = ic_ptr[v], e1 = ic_ptr[v + 1];
    if (e1 - e0 > heavy_thresh && heavy) continue;
    if (active && !active[v]) continue;
    SoftAccT<FAST> sa;
    sa.init();
    fold_gather_edges(fs, e0, e1, 1, c, sa);
    f32x4 a = {0.f, 0.f, 0.f, 0.f}, l = {0.f, 0.f, 0.f, 0.f};
    if (e1 > e0) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        a[j] = sa.acc[j] / sa.s[j];
        l[j] = sa.mx[j] + fg_log<FAST>(sa.s[j]);
      }
    }
    st4(A + (long long)v * lda + c, a);
    if (LSE) st4(LSE + (long long)v * lda + c, l);
  }
}

// value = 1: zero the G rows of the sampled endpoints and flag them (the scatter-add of their gradients follows);
// value = 0: clear the flags again.  Duplicated endpoints write the same values.
__global__ void __launch_bounds__(256) target_rows_kernel(float* __restrict__ G, long long ld, const int* __restrict__ idx,
                                                          int n, int D, unsigned char* __restrict__ flags, int value) {
  MMFT_NODE_LOOP(n, D) {
    int i = (int)(t / groups), c = (int)(t - (long long)i * groups) * 4;
    int v = idx[i];
    if (G) st4(G + (long long)v * ld + c, f32x4{0.f, 0.f, 0.f, 0.f});
    if (c == 0) flags[v] = (unsigned char)value;
  }
}

__global__ void __launch_bounds__(256) gather_rows_kernel(const float* __restrict__ src, long long lds,
                                                          const int* __restrict__ idx, int n, int D,
                                                          float* __restrict__ dst, long long ldd) {
  MMFT_NODE_LOOP(n, D) {
    int i = (int)(t / groups), c = (int)(t - (long long)i * groups) * 4;
    st4(dst + (long long)i * ldd + c, ld4(src + (long long)idx[i] * lds + c));
  }
}

__global__ void __launch_bounds__(256) scatter_add_rows_kernel(float* __restrict__ dst, long long ldd,
                                                               const int* __restrict__ idx, int n, int D,
                                                               const float* __restrict__ src, long long lds) {
  MMFT_NODE_LOOP(n, D) {
    int i = (int)(t / groups), c = (int)(t - (long long)i * groups) * 4;
    f32x4 v = ld4(src + (long long)i * lds + c);
    float* q = dst + (long long)idx[i] * ldd + c;
#pragma unroll
    for (int j = 0; j < 4; ++j) atomicAdd(q + j, v[j]);
  }
}

// dst[idx[t]] += src[t] for t in the order `order` (a stable sort of idx): the first position of every run of equal
// destinations adds the run's rows one after another in batch order - no atomics, so duplicated endpoints
// (oversampling, src/train.py:377-380) give bitwise reproducible sums.
__global__ void __launch_bounds__(256) scatter_add_rows_sorted_kernel(float* __restrict__ dst, long long ldd,
                                                                      const int* __restrict__ idx,
                                                                      const int* __restrict__ order, int n, int D,
                                                                      const float* __restrict__ src, long long lds) {
  MMFT_NODE_LOOP(n, D) {
    int p = (int)(t / groups), c = (int)(t - (long long)p * groups) * 4;
    const int v = idx[order[p]];
    if (p > 0 && idx[order[p - 1]] == v) continue;            // not the first of its run
    float* q = dst + (long long)v * ldd + c;
    f32x4 acc = ld4(q);
    for (int r = p; r < n; ++r) {
      const int tr = order[r];
      if (idx[tr] != v) break;
      acc += ld4(src + (long long)tr * lds + c);
    }
    st4(q, acc);
  }
}

// out[rows[i]] (+)= sum over the CSR segment of row i of src[indices[e]] with ONE WORKGROUP per row: the thread groups
// stride over the segment and their partial sums are combined through LDS in a fixed order.  For few, long segments
// (the gradient of PathModel.mlp_alpha's level table: 64 rows x ~170 endpoints each) - one thread group per row walks
// such a segment as a serial chain of dependent loads.
// sorted_keys != NULL: segment v = the rows e of src with sorted_keys[e] == v (keys ascending: found by two binary searches,
// no index arrays at all - the gradient of a table gathered with level-ordered indices).
__device__ __forceinline__ int seg_lower_bound(const int* __restrict__ keys, int n, int v) {
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (keys[mid] < v) lo = mid + 1;
    else hi = mid;
  }
  return lo;
}

__global__ void __launch_bounds__(256) seg_sum_wg_kernel(const float* __restrict__ src, long long lds_,
                                                         const int* __restrict__ indptr, const int* __restrict__ indices,
                                                         const int* __restrict__ rows, int n, int D,
                                                         float* __restrict__ out, long long ldo, int accumulate,
                                                         const int* __restrict__ sorted_keys, int nkeys) {
  __shared__ f32x4 part[64][4];                        // up to 64 thread groups x 4 float4 channel groups (D <= 16)...
  __shared__ f32x4 part2[32][64];                      // ...or up to 32 thread groups x 64 channel groups (D <= 256)
  const int groups = D >> 2, tgs = 256 / groups;
  const int tg = threadIdx.x / groups, cg = threadIdx.x - tg * groups, c = cg * 4;
  const bool wide = groups > 4;
  const int act = wide ? (tgs < 32 ? tgs : 32) : (tgs < 64 ? tgs : 64);
  for (int i = blockIdx.x; i < n; i += gridDim.x) {
    const int v = rows ? rows[i] : i;
    const int e0 = sorted_keys ? seg_lower_bound(sorted_keys, nkeys, v) : indptr[v];
    const int e1 = sorted_keys ? seg_lower_bound(sorted_keys, nkeys, v + 1) : indptr[v + 1];
    if (tg < act) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      for (int e = e0 + tg; e < e1; e += act) acc += ld4(src + (long long)(indices ? indices[e] : e) * lds_ + c);
      if (wide) part2[tg][cg] = acc;
      else part[tg][cg] = acc;
    }
    __syncthreads();
    if (tg == 0) {
      float* o = out + (long long)v * ldo + c;
      f32x4 s = accumulate ? ld4(o) : f32x4{0.f, 0.f, 0.f, 0.f};
      for (int t = 0; t < act; ++t) s += wide ? part2[t][cg] : part[t][cg];     // fixed order
      st4(o, s);
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Attention branch of PathConv (flag_attn = True; src/model.py:56-58,119-136,190-198).
//   message_func_attn:     e_i = leaky_relu(fc_attn([fc_key(key_u) || fc_key(key_v)]))    for the in-edge u_i -> v
//   cell_msg_reduce_attn:  alpha = softmax_i(e_i);  A[v] = sum_i alpha_i h[u_i]            (one weight per EDGE)
// fc_key is Linear(1, 256, no bias) and fc_attn Linear(512, 1, no bias), so the score is e_i = leaky_relu(c1 key_u +
// c2 key_v) with c1 = <fc_attn.w[:256], fc_key.w>, c2 = <fc_attn.w[256:], fc_key.w>: the host computes the two scalars
// with the dense kernels (so autograd carries their gradient back into both parameters) and hands them over in
// device memory.  alpha is kept per in-edge (CSR position) for the reverse sweep.
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float lrelu(float x, float slope) { return x > 0.f ? x : x * slope; }

__global__ void __launch_bounds__(256) seg_attn_fwd_kernel(const float* __restrict__ h, long long ldh,
                                                           const float* __restrict__ key, const float* __restrict__ cc,
                                                           float slope, const int* __restrict__ indptr,
                                                           const int* __restrict__ indices, const int* __restrict__ rows,
                                                           int row0, int n, int D, float* __restrict__ A, long long lda,
                                                           float* __restrict__ alpha) {
  const float c1 = cc[0], c2 = cc[1];
  MMFT_NODE_LOOP(n, D) {
    int i = (int)(t / groups), c = (int)(t - (long long)i * groups) * 4;
    int v = rows ? rows[i] : row0 + i;
    const int e0 = indptr[v], e1 = indptr[v + 1];
    const float cv = c2 * key[v];
    float mx = -INFINITY;
    for (int e = e0; e < e1; ++e) mx = fmaxf(mx, lrelu(c1 * key[indices[e]] + cv, slope));
    float ssum = 0.f;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int e = e0; e < e1; ++e) {
      const int u = indices[e];
      const float p = expf(lrelu(c1 * key[u] + cv, slope) - mx);
      ssum += p;
      acc += ld4(h + (long long)u * ldh + c) * p;
    }
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    if (e1 > e0) {
      const float inv = 1.0f / ssum;
      a = acc * inv;
      if (c == 0)
        for (int e = e0; e < e1; ++e) alpha[e] = expf(lrelu(c1 * key[indices[e]] + cv, slope) - mx) * inv;
    }
    st4(A + (long long)v * lda + c, a);
  }
}

// reverse pull with per-edge attention weights: G[v] = mask( own + sum_{net out} G[w] wgt + sum_{cell out e} alpha[o2i[e]] DA[w] )
__global__ void __launch_bounds__(256) level_bwd_pull_attn_kernel(
    float* __restrict__ G, const float* __restrict__ h, long long ld, const int* __restrict__ rows, int row0, int n, int D,
    const int* __restrict__ on_ptr, const int* __restrict__ on_idx, const float* __restrict__ on_w,
    const int* __restrict__ oc_ptr, const int* __restrict__ oc_idx, const int* __restrict__ o2i,
    const float* __restrict__ alpha, const float* __restrict__ DA, int relu, const unsigned char* __restrict__ own) {
  MMFT_NODE_LOOP(n, D) {
    int i = (int)(t / groups), c = (int)(t - (long long)i * groups) * 4;
    int v = rows ? rows[i] : row0 + i;
    const long long off = (long long)v * ld + c;
    f32x4 g = (!own || own[v]) ? ld4(G + off) : f32x4{0.f, 0.f, 0.f, 0.f};
    for (int e = on_ptr[v]; e < on_ptr[v + 1]; ++e) g += ld4(G + (long long)on_idx[e] * ld + c) * on_w[e];
    for (int e = oc_ptr[v]; e < oc_ptr[v + 1]; ++e) g += ld4(DA + (long long)oc_idx[e] * ld + c) * alpha[o2i[e]];
    if (relu) {
      const f32x4 hv = ld4(h + off);
#pragma unroll
      for (int j = 0; j < 4; ++j) g[j] = hv[j] > 0.f ? g[j] : 0.f;
    }
    st4(G + off, g);
  }
}

// gradient of the edge scores of one cell level: dcp[v] = (sum_i da_i key_u, sum_i da_i key_v) with
//   de_i = alpha_i (<DA[v], h[u_i]> - <DA[v], A[v]>),  da_i = de_i * leaky_relu'(c1 key_u + c2 key_v)
// (D / 4 lanes of a node reduce their channel shares with xor shuffles: D / 4 must be a power of two <= 64)
__global__ void __launch_bounds__(256) seg_attn_bwd_scores_kernel(
    const float* __restrict__ DA, const float* __restrict__ h, const float* __restrict__ A, long long ld,
    const float* __restrict__ alpha, const float* __restrict__ key, const float* __restrict__ cc, float slope,
    const int* __restrict__ indptr, const int* __restrict__ indices, const int* __restrict__ rows, int row0, int n, int D,
    float* __restrict__ dcp) {
  const float c1 = cc[0], c2 = cc[1];
  const int groups = D >> 2;
  const long long total = (long long)n * groups;
  // every lane of a wave takes part in the shuffles: loop bounds are rounded up to whole thread groups (always true
  // here since total is a multiple of groups and groups divides the block size)
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
    int i = (int)(t / groups), c = (int)(t - (long long)i * groups) * 4;
    int v = rows ? rows[i] : row0 + i;
    const f32x4 da = ld4(DA + (long long)v * ld + c), av = ld4(A + (long long)v * ld + c);
    float tpart = da.x * av.x + da.y * av.y + da.z * av.z + da.w * av.w;
    for (int m = 1; m < groups; m <<= 1) tpart += __shfl_xor(tpart, m, 64);
    const float kv = key[v];
    float p1 = 0.f, p2 = 0.f;
    for (int e = indptr[v]; e < indptr[v + 1]; ++e) {
      const int u = indices[e];
      const f32x4 hu = ld4(h + (long long)u * ld + c);
      float sp = da.x * hu.x + da.y * hu.y + da.z * hu.z + da.w * hu.w;
      for (int m = 1; m < groups; m <<= 1) sp += __shfl_xor(sp, m, 64);
      const float ku = key[u];
      const float pre = c1 * ku + c2 * kv;
      const float dai = alpha[e] * (sp - tpart) * (pre > 0.f ? 1.0f : slope);
      p1 += dai * ku;
      p2 += dai * kv;
    }
    if (c == 0) {
      dcp[(long long)v * 2] = p1;
      dcp[(long long)v * 2 + 1] = p2;
    }
  }
}

// out[rows[i] or i][0..D) = mean over the CSR segment of src rows, any D (one thread per element): ndata['h_drive'] of the
// attention branch (src/model.py:197-198: fn.copy_src('net_feat') + fn.mean, D = net_feat_dim = 2)
__global__ void __launch_bounds__(256) seg_mean_any_kernel(const float* __restrict__ src, long long lds_,
                                                           const int* __restrict__ indptr, const int* __restrict__ indices,
                                                           const int* __restrict__ rows, int n, int D,
                                                           float* __restrict__ out, long long ldo, int scatter) {
  const long long total = (long long)n * D;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
    int i = (int)(t / D), c = (int)(t - (long long)i * D);
    int v = rows ? rows[i] : i;
    const int e0 = indptr[v], e1 = indptr[v + 1];
    float acc = 0.f;
    for (int e = e0; e < e1; ++e) acc += src[(long long)indices[e] * lds_ + c];
    if (e1 > e0) acc = acc / (float)(e1 - e0);
    out[(long long)(scatter ? v : i) * ldo + c] = acc;
  }
}

static inline int node_grid(int n, int D) { return ew_grid((long long)n * (D / 4)); }

}  // namespace mmft

using namespace mmft;

#define CHECK_ROWS(name)                                                                              \
  MMFT_REQUIRE(n >= 0 && D > 0 && (D % 4) == 0, name ": D must be a positive multiple of 4 (D=%d)", D)

extern "C" {

int mmft_seg_softmax_sum_fwd(const float* h, long long ldh, const int* in_indptr, const int* in_indices,
                             const int* rows, int row0, int n, int D, float* A, float* LSE, long long lda,
                             long long alg_bytes, int device, void* stream) {
  CHECK_ROWS("seg_softmax_sum_fwd");
  MMFT_REQUIRE(h && in_indptr && A, "seg_softmax_sum_fwd: null pointer");
  MMFT_REQUIRE(ldh >= D && lda >= D && ldh % 4 == 0 && lda % 4 == 0 && aligned16(h) && aligned16(A) &&
                   (!LSE || aligned16(LSE)),
               "seg_softmax_sum_fwd: rows must be 16-byte aligned");
  if (n == 0) return MMFT_OK;
  DeviceGuard dg(device);
  MMFT_LAUNCH("seg_softmax_sum_fwd_kernel", 0.0, alg_bytes > 0 ? (double)alg_bytes : 3.0 * 4.0 * n * D,
              seg_softmax_sum_fwd_kernel, dim3(node_grid(n, D)), dim3(256), (hipStream_t)stream, h, ldh, in_indptr,
              in_indices, rows, row0, n, D, A, LSE, lda);
  return check_launch("seg_softmax_sum_fwd");
}

int mmft_seg_mean_add_act_fwd(float* h, long long ldh, const int* in_indptr, const int* in_indices, const int* rows,
                              int row0, int n, int D, int relu, long long alg_bytes, int device, void* stream) {
  CHECK_ROWS("seg_mean_add_act_fwd");
  MMFT_REQUIRE(h && in_indptr, "seg_mean_add_act_fwd: null pointer");
  MMFT_REQUIRE(ldh >= D && ldh % 4 == 0 && aligned16(h), "seg_mean_add_act_fwd: rows must be 16-byte aligned");
  if (n == 0) return MMFT_OK;
  DeviceGuard dg(device);
  MMFT_LAUNCH("seg_mean_fwd_kernel", 0.0, alg_bytes > 0 ? (double)alg_bytes : 3.0 * 4.0 * n * D, seg_mean_fwd_kernel,
              dim3(node_grid(n, D)), dim3(256), (hipStream_t)stream, h, ldh, in_indptr, in_indices, rows, row0, n, D, h,
              ldh, 1, relu, 1);
  return check_launch("seg_mean_add_act_fwd");
}

int mmft_seg_mean_fwd(const float* src, long long lds, const int* in_indptr, const int* in_indices, const int* rows,
                      int n, int D, float* out, long long ldo, int device, void* stream) {
  CHECK_ROWS("seg_mean_fwd");
  MMFT_REQUIRE(src && in_indptr && out, "seg_mean_fwd: null pointer");
  MMFT_REQUIRE(lds >= D && ldo >= D && lds % 4 == 0 && ldo % 4 == 0 && aligned16(src) && aligned16(out),
               "seg_mean_fwd: rows must be 16-byte aligned");
  if (n == 0) return MMFT_OK;
  DeviceGuard dg(device);
  MMFT_LAUNCH("seg_mean_fwd_kernel", 0.0, 3.0 * 4.0 * n * D, seg_mean_fwd_kernel, dim3(node_grid(n, D)), dim3(256), (hipStream_t)stream, src, lds, in_indptr, in_indices, rows, 0, n, D, out, ldo, 0, 0, 1);
  return check_launch("seg_mean_fwd");
}

int mmft_seg_attn_fwd(const float* h, long long ldh, const float* key, const float* c12, float slope, const int* in_indptr,
                      const int* in_indices, const int* rows, int row0, int n, int D, float* A, long long lda, float* alpha,
                      int device, void* stream) {
  CHECK_ROWS("seg_attn_fwd");
  if (n == 0) return MMFT_OK;
  MMFT_REQUIRE(h && key && c12 && in_indptr && A && alpha, "seg_attn_fwd: null pointer");
  MMFT_REQUIRE(ldh >= D && lda >= D && ldh % 4 == 0 && lda % 4 == 0 && aligned16(h) && aligned16(A),
               "seg_attn_fwd: rows must be 16-byte aligned");
  DeviceGuard dg(device);
  MMFT_LAUNCH("seg_attn_fwd_kernel", 0.0, 3.0 * 4.0 * n * D, seg_attn_fwd_kernel, dim3(node_grid(n, D)), dim3(256),
              (hipStream_t)stream, h, ldh, key, c12, slope, in_indptr, in_indices, rows, row0, n, D, A, lda, alpha);
  return check_launch("seg_attn_fwd");
}

int mmft_level_bwd_pull_attn(float* G, const float* h, long long ld, const int* rows, int row0, int n, int D,
                             const int* out_net_indptr, const int* out_net_indices, const float* out_net_weight,
                             const int* out_cell_indptr, const int* out_cell_indices, const int* out2in_cell,
                             const float* alpha, const float* DA, int relu, const unsigned char* own_mask, int device,
                             void* stream) {
  CHECK_ROWS("level_bwd_pull_attn");
  if (n == 0) return MMFT_OK;
  MMFT_REQUIRE(G && h && out_net_indptr && out_cell_indptr && DA, "level_bwd_pull_attn: null pointer");
  MMFT_REQUIRE(ld >= D && ld % 4 == 0 && aligned16(G) && aligned16(h) && aligned16(DA),
               "level_bwd_pull_attn: rows must be 16-byte aligned");
  DeviceGuard dg(device);
  MMFT_LAUNCH("level_bwd_pull_attn_kernel", 0.0, 3.0 * 4.0 * n * D, level_bwd_pull_attn_kernel, dim3(node_grid(n, D)),
              dim3(256), (hipStream_t)stream, G, h, ld, rows, row0, n, D, out_net_indptr, out_net_indices, out_net_weight,
              out_cell_indptr, out_cell_indices, out2in_cell, alpha, DA, relu, own_mask);
  return check_launch("level_bwd_pull_attn");
}

int mmft_seg_attn_bwd_scores(const float* DA, const float* h, const float* A, long long ld, const float* alpha,
                             const float* key, const float* c12, float slope, const int* in_indptr, const int* in_indices,
                             const int* rows, int row0, int n, int D, float* dcp, int device, void* stream) {
  CHECK_ROWS("seg_attn_bwd_scores");
  if (n == 0) return MMFT_OK;
  MMFT_REQUIRE(DA && h && A && alpha && key && c12 && in_indptr && dcp, "seg_attn_bwd_scores: null pointer");
  const int groups = D / 4;
  MMFT_REQUIRE(groups <= 64 && (groups & (groups - 1)) == 0, "seg_attn_bwd_scores: D / 4 must be a power of two <= 64");
  MMFT_REQUIRE(ld >= D && ld % 4 == 0 && aligned16(DA) && aligned16(h) && aligned16(A),
               "seg_attn_bwd_scores: rows must be 16-byte aligned");
  DeviceGuard dg(device);
  MMFT_LAUNCH("seg_attn_bwd_scores_kernel", 0.0, 3.0 * 4.0 * n * D, seg_attn_bwd_scores_kernel, dim3(node_grid(n, D)),
              dim3(256), (hipStream_t)stream, DA, h, A, ld, alpha, key, c12, slope, in_indptr, in_indices, rows, row0, n, D, dcp);
  return check_launch("seg_attn_bwd_scores");
}

int mmft_seg_mean_rows_any(const float* src, long long lds, const int* indptr, const int* indices, const int* rows, int n,
                           int D, float* out, long long ldo, int scatter, int device, void* stream) {
  MMFT_REQUIRE(n >= 0 && D > 0 && lds >= D && ldo >= D, "seg_mean_rows_any: bad sizes");
  if (n == 0) return MMFT_OK;
  MMFT_REQUIRE(src && indptr && out, "seg_mean_rows_any: null pointer");
  DeviceGuard dg(device);
  hipLaunchKernelGGL(seg_mean_any_kernel, dim3(ew_grid((long long)n * D)), dim3(256), 0, (hipStream_t)stream, src, lds, indptr,
                     indices, rows, n, D, out, ldo, scatter);
  return check_launch("seg_mean_rows_any");
}

int mmft_seg_sum_rows_wg(const float* src, long long lds, const int* indptr, const int* indices, const int* rows, int n,
                         int D, float* out, long long ldo, int accumulate, int device, void* stream) {
  CHECK_ROWS("seg_sum_rows_wg");
  MMFT_REQUIRE(src && indptr && out, "seg_sum_rows_wg: null pointer");
  MMFT_REQUIRE(lds >= D && ldo >= D && lds % 4 == 0 && ldo % 4 == 0 && aligned16(src) && aligned16(out),
               "seg_sum_rows_wg: rows must be 16-byte aligned");
  MMFT_REQUIRE(D <= 256 && 256 % (D / 4) == 0, "seg_sum_rows_wg: D / 4 must divide 256 (D <= 256)");
  if (n == 0) return MMFT_OK;
  DeviceGuard dg(device);
  MMFT_LAUNCH("seg_sum_wg_kernel", 0.0, 0.0, seg_sum_wg_kernel, dim3(n < 2048 ? n : 2048), dim3(256), (hipStream_t)stream, src,
              lds, indptr, indices, rows, n, D, out, ldo, accumulate, (const int*)nullptr, 0);
  return check_launch("seg_sum_rows_wg");
}

int mmft_seg_sum_sorted(const float* src, long long lds, const int* sorted_keys, int nsrc, int R, int D, float* out, long long ldo,
                        int accumulate, int device, void* stream) {
  MMFT_REQUIRE(src && sorted_keys && out && nsrc >= 0 && R > 0, "seg_sum_sorted: bad arguments");
  MMFT_REQUIRE(D % 4 == 0 && D <= 256 && 256 % (D / 4) == 0 && lds >= D && ldo >= D && lds % 4 == 0 && ldo % 4 == 0 &&
                   aligned16(src) && aligned16(out),
               "seg_sum_sorted: D / 4 must divide 256 (D <= 256), rows 16-byte aligned");
  DeviceGuard dg(device);
  MMFT_LAUNCH("seg_sum_wg_kernel", (double)nsrc * D, 4.0 * ((double)nsrc * (D + 1) + 2.0 * R * D), seg_sum_wg_kernel,
              dim3(R < 2048 ? R : 2048), dim3(256), (hipStream_t)stream, src, lds, (const int*)nullptr, (const int*)nullptr,
              (const int*)nullptr, R, D, out, ldo, accumulate, sorted_keys, nsrc);
  return check_launch("seg_sum_sorted");
}

int mmft_seg_sum_fwd(const float* src, long long lds, const int* indptr, const int* indices, const int* rows, int n,
                     int D, float* out, long long ldo, int accumulate, int device, void* stream) {
  CHECK_ROWS("seg_sum_fwd");
  MMFT_REQUIRE(src && indptr && out, "seg_sum_fwd: null pointer");
  MMFT_REQUIRE(lds >= D && ldo >= D && lds % 4 == 0 && ldo % 4 == 0 && aligned16(src) && aligned16(out),
               "seg_sum_fwd: rows must be 16-byte aligned");
  if (n == 0) return MMFT_OK;
  DeviceGuard dg(device);
  MMFT_LAUNCH("seg_mean_fwd_kernel", 0.0, 3.0 * 4.0 * n * D, seg_mean_fwd_kernel, dim3(node_grid(n, D)), dim3(256), (hipStream_t)stream, src, lds, indptr, indices, rows, 0, n, D, out, ldo, accumulate ? 1 : 0, 0, 0);
  return check_launch("seg_sum_fwd");
}

int mmft_level_bwd_pull(float* G, const float* h, long long ld, const int* rows, int row0, int n, int D,
                        const int* out_net_indptr, const int* out_net_indices, const float* out_net_weight,
                        const int* out_cell_indptr, const int* out_cell_indices, const float* A, const float* LSE,
                        const float* DA, int relu, const unsigned char* own_mask, const int* heavy_rows, int nheavy,
                        int heavy_thresh, const unsigned char* active, long long alg_bytes, int device, void* stream) {
  CHECK_ROWS("level_bwd_pull");
  MMFT_REQUIRE(nheavy >= 0 && (nheavy == 0 || heavy_rows) && heavy_thresh >= 0, "level_bwd_pull: heavy row list");
  MMFT_REQUIRE(nheavy == 0 || (D <= 256 && 256 % (D / 4) == 0), "level_bwd_pull: heavy rows need D / 4 to divide 256");
  MMFT_REQUIRE(G && h && out_net_indptr && out_cell_indptr && A && LSE && DA,
               "level_bwd_pull: null pointer");
  MMFT_REQUIRE(ld >= D && ld % 4 == 0 && aligned16(G) && aligned16(h) && aligned16(A) && aligned16(LSE) &&
                   aligned16(DA),
               "level_bwd_pull: rows must be 16-byte aligned");
  if (n == 0) return MMFT_OK;
  DeviceGuard dg(device);
  const int light = node_grid(n, D);
  const int hb = nheavy < 1024 ? nheavy : 1024;
  const double by = alg_bytes > 0 ? (double)alg_bytes : 3.0 * 4.0 * n * D;
  if (math_mode() == MMFT_MATH_BF16)
    MMFT_LAUNCH("level_bwd_pull_kernel", 0.0, by, level_bwd_pull_kernel<true>, dim3(light + hb), dim3(256), (hipStream_t)stream, G, h, ld,
                rows, row0, n, D, out_net_indptr, out_net_indices, out_net_weight, out_cell_indptr, out_cell_indices, A, LSE, DA, relu,
                own_mask, nheavy ? heavy_rows : nullptr, nheavy, light, heavy_thresh, active);
  else
    MMFT_LAUNCH("level_bwd_pull_kernel", 0.0, by, level_bwd_pull_kernel<false>, dim3(light + hb), dim3(256), (hipStream_t)stream, G, h, ld,
                rows, row0, n, D, out_net_indptr, out_net_indices, out_net_weight, out_cell_indptr, out_cell_indices, A, LSE, DA, relu,
                own_mask, nheavy ? heavy_rows : nullptr, nheavy, light, heavy_thresh, active);
  return check_launch("level_bwd_pull");
}

int mmft_pair_fwd_gather(float* h, const float* pre, long long ld, int D, const int* in_net_indptr,
                         const int* in_net_indices, const int* in_cell_indptr, const int* in_cell_indices, int net_row0,
                         int n_net, const int* cell_rows, int cell_row0, int n_cell, float* A, float* LSE, long long lda,
                         int relu, const int* heavy_rows, int nheavy, int heavy_thresh, const unsigned char* active,
                         const int* in_cell_driver, long long alg_bytes, int device, void* stream) {
  const int n = n_net + n_cell;
  CHECK_ROWS("pair_fwd_gather");
  MMFT_REQUIRE(n_net >= 0 && n_cell >= 0 && net_row0 >= 0 && cell_row0 >= 0, "pair_fwd_gather: negative row count / offset");
  if (n == 0) return MMFT_OK;
  MMFT_REQUIRE(h && pre && in_net_indptr && in_cell_indptr && (A || n_cell == 0), "pair_fwd_gather: null pointer");
  MMFT_REQUIRE(ld >= D && ld % 4 == 0 && aligned16(h) && aligned16(pre) && (!A || (aligned16(A) && lda >= D && lda % 4 == 0)) &&
                   (!LSE || aligned16(LSE)),
               "pair_fwd_gather: rows must be 16-byte aligned");
  MMFT_REQUIRE(D <= 256 && 256 % (D / 4) == 0, "pair_fwd_gather: D / 4 must divide 256 (D <= 256)");
  MMFT_REQUIRE(nheavy >= 0 && (nheavy == 0 || heavy_rows) && heavy_thresh >= 0, "pair_fwd_gather: heavy row list");
  DeviceGuard dg(device);
  const int light = node_grid(n, D);
  const int hb = nheavy < 512 ? nheavy : 512;
  const double gby = alg_bytes > 0 ? (double)alg_bytes : 3.0 * 4.0 * n * D;
  if (math_mode() == MMFT_MATH_BF16)             // the hardware exp / log of the bf16 mode's level kernels (fold_gather.h)
    MMFT_LAUNCH("pair_fwd_gather_kernel", 0.0, gby, pair_fwd_gather_kernel<true>, dim3(light + hb), dim3(256), (hipStream_t)stream, h,
                pre, ld, D, in_net_indptr, in_net_indices, in_cell_indptr, in_cell_indices, net_row0, n_net, cell_rows, cell_row0,
                n_cell, A, LSE, lda, relu, nheavy ? heavy_rows : nullptr, nheavy, light, heavy_thresh, active, in_cell_driver);
  else
    MMFT_LAUNCH("pair_fwd_gather_kernel", 0.0, gby, pair_fwd_gather_kernel<false>, dim3(light + hb), dim3(256), (hipStream_t)stream, h,
                pre, ld, D, in_net_indptr, in_net_indices, in_cell_indptr, in_cell_indices, net_row0, n_net, cell_rows, cell_row0,
                n_cell, A, LSE, lda, relu, nheavy ? heavy_rows : nullptr, nheavy, light, heavy_thresh, active, in_cell_driver);
  return check_launch("pair_fwd_gather");
}

int mmft_target_rows_begin(float* G, long long ld, const int* idx, int n, int D, unsigned char* flags, int device,
                           void* stream) {
  CHECK_ROWS("target_rows_begin");
  if (n == 0) return MMFT_OK;
  MMFT_REQUIRE(G && idx && flags && ld >= D && ld % 4 == 0 && aligned16(G), "target_rows_begin: bad arguments");
  DeviceGuard dg(device);
  hipLaunchKernelGGL(target_rows_kernel, dim3(node_grid(n, D)), dim3(256), 0, (hipStream_t)stream, G, ld, idx, n, D, flags, 1);
  return check_launch("target_rows_begin");
}

int mmft_target_rows_end(const int* idx, int n, unsigned char* flags, int device, void* stream) {
  MMFT_REQUIRE(n >= 0, "target_rows_end: negative count");
  if (n == 0) return MMFT_OK;
  MMFT_REQUIRE(idx && flags, "target_rows_end: null pointer");
  DeviceGuard dg(device);
  hipLaunchKernelGGL(target_rows_kernel, dim3(node_grid(n, 4)), dim3(256), 0, (hipStream_t)stream, nullptr, 0, idx, n, 4, flags, 0);
  return check_launch("target_rows_end");
}

int mmft_mark_rows(const int* idx, int n, unsigned char* flags, int value, int device, void* stream) {
  MMFT_REQUIRE(n >= 0, "mark_rows: negative count");
  if (n == 0) return MMFT_OK;
  MMFT_REQUIRE(idx && flags, "mark_rows: null pointer");
  DeviceGuard dg(device);
  hipLaunchKernelGGL(target_rows_kernel, dim3(node_grid(n, 4)), dim3(256), 0, (hipStream_t)stream, nullptr, 0, idx, n, 4, flags,
                     value ? 1 : 0);
  return check_launch("mark_rows");
}

int mmft_gather_rows(const float* src, long long lds, const int* idx, int n, int D, float* dst, long long ldd,
                     int device, void* stream) {
  CHECK_ROWS("gather_rows");
  MMFT_REQUIRE(src && dst && (idx || n == 0), "gather_rows: null pointer");
  MMFT_REQUIRE(lds >= D && ldd >= D && lds % 4 == 0 && ldd % 4 == 0 && aligned16(src) && aligned16(dst),
               "gather_rows: rows must be 16-byte aligned");
  if (n == 0) return MMFT_OK;
  DeviceGuard dg(device);
  MMFT_LAUNCH("gather_rows_kernel", 0.0, 2.0 * 4.0 * n * D, gather_rows_kernel, dim3(node_grid(n, D)), dim3(256), (hipStream_t)stream, src, lds, idx, n, D, dst, ldd);
  return check_launch("gather_rows");
}

int mmft_scatter_add_rows(float* dst, long long ldd, const int* idx, int n, int D, const float* src, long long lds,
                          int device, void* stream) {
  CHECK_ROWS("scatter_add_rows");
  MMFT_REQUIRE(src && dst && (idx || n == 0), "scatter_add_rows: null pointer");
  MMFT_REQUIRE(lds >= D && ldd >= D && lds % 4 == 0 && ldd % 4 == 0 && aligned16(src) && aligned16(dst),
               "scatter_add_rows: rows must be 16-byte aligned");
  if (n == 0) return MMFT_OK;
  DeviceGuard dg(device);
  MMFT_LAUNCH("scatter_add_rows_kernel", 0.0, 2.0 * 4.0 * n * D, scatter_add_rows_kernel, dim3(node_grid(n, D)), dim3(256), (hipStream_t)stream, dst, ldd, idx, n, D, src, lds);
  return check_launch("scatter_add_rows");
}

int mmft_scatter_add_rows_sorted(float* dst, long long ldd, const int* idx, const int* order, int n, int D,
                                 const float* src, long long lds, int device, void* stream) {
  CHECK_ROWS("scatter_add_rows_sorted");
  MMFT_REQUIRE(src && dst && ((idx && order) || n == 0), "scatter_add_rows_sorted: null pointer");
  MMFT_REQUIRE(lds >= D && ldd >= D && lds % 4 == 0 && ldd % 4 == 0 && aligned16(src) && aligned16(dst),
               "scatter_add_rows_sorted: rows must be 16-byte aligned");
  if (n == 0) return MMFT_OK;
  DeviceGuard dg(device);
  MMFT_LAUNCH("scatter_add_rows_sorted_kernel", 0.0, 3.0 * 4.0 * n * D, scatter_add_rows_sorted_kernel,
              dim3(node_grid(n, D)), dim3(256), (hipStream_t)stream, dst, ldd, idx, order, n, D, src, lds);
  return check_launch("scatter_add_rows_sorted");
}

}  // extern "C"

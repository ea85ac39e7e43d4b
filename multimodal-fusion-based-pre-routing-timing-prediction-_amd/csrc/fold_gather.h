// Folded forward gather of one (net level l - 1, cell level l) pair, shared by mmft_pair_fwd_gather (graph.hip) and the
// fused bf16 level kernel (mlp2_bf16.hip).
//
// A cell row v of level l takes the softmax-weighted sum (src/model.py:100-117) over its in-edges; the source of an edge
// is a net u whose own value relu(mean_in(h) + PRE[u]) (src/model.py:92-98) is recomputed on the fly when u belongs to
// the folded level l - 1 (its row of h is being written by the same launch) and read from h otherwise.
//
// Walking that in series is a five-deep chain of dependent loads per edge (ic_ptr -> ic_idx -> in_ptr -> in_idx -> rows)
// and one thread group has the row's edges in series behind it: the launch of a 7 000-row level took 30 us with the
// card almost idle.  Two things shorten it:
//   * `ic_drv` (optional, one int per cell in-edge, built on the host next to the CSR): the single driver of the net
//     behind the edge, so that the chain is ic_ptr -> (ic_idx, ic_drv) -> rows;
//   * FOUR edges are requested together, their index loads first, then their row loads, and the softmax recurrence
//     consumes them in edge order (same arithmetic, same order: bitwise equal to the serial form).
#pragma once
#include "common.h"

namespace mmft {

typedef float fg_f32x4 __attribute__((ext_vector_type(4)));

struct FoldSrc {
  const float* h;
  const float* pre;
  long long ld;
  const int *in_ptr, *in_idx;     // in-CSR of the net edges (driver -> net)
  const int *ic_idx, *ic_drv;     // in-CSR indices of the cell edges (net -> cell); ic_drv optional
  int net_row0, n_net;
  int relu;
};

__device__ __forceinline__ fg_f32x4 fg_ld4(const float* p) { return *reinterpret_cast<const fg_f32x4*>(p); }

__device__ __forceinline__ fg_f32x4 fg_finish_net(fg_f32x4 acc, fg_f32x4 pre, int relu) {
  acc += pre;
  if (relu) {
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = acc[j] > 0.f ? acc[j] : 0.f;
  }
  return acc;
}

// value of net u, channel group c: relu(mean over in-edges of h + PRE[u])
__device__ __forceinline__ fg_f32x4 fold_net_value(const FoldSrc& s, int u, int c) {
  const int e0 = s.in_ptr[u], e1 = s.in_ptr[u + 1];
  fg_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int e = e0; e < e1; ++e) acc += fg_ld4(s.h + (long long)s.in_idx[e] * s.ld + c);
  if (e1 > e0) acc = acc * (1.0f / (float)(e1 - e0));
  return fg_finish_net(acc, fg_ld4(s.pre + (long long)u * s.ld + c), s.relu);
}

// Reverse sweep, one cell consumer c of a net row: g += DA[c] * exp(h - LSE[c]) * (1 + h - A[c])   (d a_c / d m of the
// softmax-weighted sum, src/model.py:113-116).  Every operation is rounded on its own (no fused multiply-add), so that the
// kernels that share this term - level_bwd_pull (graph.hip) and the pair kernel (mlp2_bf16.hip) - agree bit for bit
// whatever the compiler would contract around them.  FAST (bf16 math mode): the exponential is the hardware's
// v_exp_f32(x log2 e) - two instructions instead of ~15, relative error ~|x| 6e-8 on a factor in (0, 1]; the reverse sweep's
// sink phase is bound by VALU issue, not by memory, and 16 of these per lane per sink were most of it.
template <bool FAST>
__device__ __forceinline__ float cell_consumer_term(float g, float da, float hv, float lse, float a) {
  const float x = __fsub_rn(hv, lse);
  const float ex = FAST ? __builtin_amdgcn_exp2f(__fmul_rn(x, 1.44269504088896340736f)) : expf(x);
  const float w = __fmul_rn(da, ex);
  return __fadd_rn(g, __fmul_rn(w, __fsub_rn(__fadd_rn(1.0f, hv), a)));
}

// exp / log of the online softmax.  FAST (bf16 math mode): the hardware's v_exp_f32 / v_log_f32 (base 2, one multiply for the
// base change, each rounded on its own so that every kernel that shares these helpers produces the same bits): the gather of a
// level is 32 exponentials per lane and row, ~20 VALU instructions each in the accurate form - a third of the kernel's time.
template <bool FAST>
__device__ __forceinline__ float fg_exp(float x) {
  return FAST ? __builtin_amdgcn_exp2f(__fmul_rn(x, 1.44269504088896340736f)) : expf(x);
}
template <bool FAST>
__device__ __forceinline__ float fg_log(float x) {
  return FAST ? __fmul_rn(__builtin_amdgcn_logf(x), 0.69314718055994530942f) : logf(x);
}

template <bool FAST>
struct SoftAccT {
  fg_f32x4 mx, s, acc;
  __device__ __forceinline__ void init() {
    mx = fg_f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    s = fg_f32x4{0.f, 0.f, 0.f, 0.f};
    acc = s;
  }
  __device__ __forceinline__ void add(fg_f32x4 x) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float m_new = fmaxf(mx[j], x[j]);
      float scale = fg_exp<FAST>(mx[j] - m_new);      // exp(-inf) = 0 on the first edge
      float p = fg_exp<FAST>(x[j] - m_new);
      s[j] = s[j] * scale + p;
      acc[j] = acc[j] * scale + p * x[j];
      mx[j] = m_new;
    }
  }
};
typedef SoftAccT<false> SoftAcc;

// Adds the edges e0, e0 + stride, ... < e1 of one cell row to `sa`, in that order.
template <bool FAST>
__device__ __forceinline__ void fold_gather_edges(const FoldSrc& s, int e0, int e1, int stride, int c, SoftAccT<FAST>& sa) {
  if (!s.ic_drv) {
    for (int e = e0; e < e1; e += stride) {
      const int u = s.ic_idx[e];
      sa.add((unsigned)(u - s.net_row0) < (unsigned)s.n_net ? fold_net_value(s, u, c) : fg_ld4(s.h + (long long)u * s.ld + c));
    }
    return;
  }
  // Branch-free request phase: a missing edge of the last batch repeats the row's last edge (its loads hit the same
  // lines) and is dropped by a select in the recurrence - conditional loads would sit in separate basic blocks and the
  // compiler then waits for each edge's rows before it requests the next one's.
  for (int e = e0; e < e1; e += 4 * stride) {
    int u[4], d[4];
    bool ok[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int ee = e + k * stride;
      ok[k] = ee < e1;
      const int es = ok[k] ? ee : e;
      u[k] = s.ic_idx[es];
      d[k] = s.ic_drv[es];
    }
    fg_f32x4 a[4], p[4];
    bool folded[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      folded[k] = (unsigned)(u[k] - s.net_row0) < (unsigned)s.n_net;
      a[k] = fg_ld4(s.h + (long long)((folded[k] && d[k] >= 0) ? d[k] : u[k]) * s.ld + c);
      p[k] = fg_ld4(s.pre + (long long)u[k] * s.ld + c);        // read (and ignored) for the rare edge from an older net level
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      fg_f32x4 x = folded[k] ? fg_finish_net(a[k], p[k], s.relu) : a[k];   // mean over ONE in-edge = the driver's row itself
      if (folded[k] && d[k] < 0) x = fold_net_value(s, u[k], c);           // net with no or several drivers (not in a folded schedule)
      SoftAccT<FAST> nx = sa;
      nx.add(x);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        sa.mx[j] = ok[k] ? nx.mx[j] : sa.mx[j];
        sa.s[j] = ok[k] ? nx.s[j] : sa.s[j];
        sa.acc[j] = ok[k] ? nx.acc[j] : sa.acc[j];
      }
    }
  }
}

}  // namespace mmft

// Fused backward of the FIRST Linear of a Linear-ReLU-Linear MLP (the reference's MLP(in, 256, out),
// src/model.py:10-24, as used for fc_cell_self / fc_net_self, src/model.py:66-67):
//
//   dH  = (G . W2) * (H > 0)          G:[rows, D2]  W2:[D2, HD]  H = saved hidden activations [rows, HD]
//   dW1 = dH^T . X                    X:[rows, Fin]
//   db1 = column sums of dH
//
// The unfused path wrote dH (rows x 256 fp32 = 268 MB at the benchmark size) with one GEMM and read it back with
// a second GEMM; here a 16-row block of dH never leaves the registers of the four waves that computed it.
//
// Weight-stationary streaming design (study: tools/proto/rowgemm.hip):
//   * W2^T sits in LDS for the life of the (persistent) workgroup; the main loop has no barrier.
//   * A quad of waves owns one 16-row block at a time; wave w of the quad computes hidden columns [64w, 64w+64).
//   * MFMA #1 (A = G fragment, rows on the i axis; B = W2^T fragment from LDS) leaves dH in the accumulator layout
//     "lane holds rows 4q..4q+3 of column lane&15", which is exactly the B-operand layout of MFMA #2 when its
//     reduction index (the block's 16 rows) is ordered k(q, s) = 4q + s - the same k permutation gemm_engine.h
//     uses.  MFMA #2 (A = X^T fragment) accumulates dW1^T[feature][hidden column] in registers over all blocks.
//   * G of the NEXT block is prefetched while the current one is multiplied; mask and X of the current block are
//     requested at the top of the block and consumed ~4000 cycles later.  No stores inside the loop.
//   * Per-quad partial results go to slabs that slab_reduce sums in a fixed order (bitwise reproducible).
#include "gemm_engine.h"

namespace mmft {

int launch_slab_reduce(const float* slabs, int splits, long long elems, float* out, int accumulate, hipStream_t st);

constexpr int MG_HD = 256, MG_D2 = 128, MG_WS = MG_D2 + 8, MG_WAVES = 8;

struct MlpGradArgs {
  const float* g;
  long long ldg;
  const float* hid;
  long long ldh;
  const float* x;
  long long ldx;
  const int* rows;
  int row0, n, fin;
  const float* w2;
  long long ldw2;
  float* slab_w;    // [2 * grid] slabs of [HD][fin], `wstride` floats apart
  float* slab_b;    // [2 * grid] slabs of [HD], `bstride` floats apart (interleaved with slab_w when dw1 | db1 are neighbours)
  long long wstride, bstride;
};

template <int FT, bool BF>   // input-feature subtiles of 16 (fin <= 16 * FT); BF: bf16 operands at the MFMAs (MMFT_MATH_BF16)
__global__ void __launch_bounds__(MG_WAVES * 64, 1) mlp_first_layer_grads_kernel(MlpGradArgs a) {
  constexpr int KB = MG_D2 / 16;           // 8 K chunks of 16 for MFMA #1
  extern __shared__ __attribute__((aligned(16))) float w2t[];      // [HD][D2 + 8]: W2^T, conflict-free b128 reads
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // W2 is [D2][HD] row-major; W2^T goes to LDS once per workgroup.  Lanes run along k so that the four transposed
  // LDS writes of a thread hit consecutive banks across the wave (the 16-byte global reads are then 1 KB apart, which
  // only matters for these 128 KB that stay in L2).
  for (int e = tid; e < MG_D2 * MG_HD / 4; e += MG_WAVES * 64) {
    int k = e % MG_D2, n4 = e / MG_D2;
    f32x4 v = *reinterpret_cast<const f32x4*>(a.w2 + (long long)k * a.ldw2 + n4 * 4);
    w2t[(n4 * 4 + 0) * MG_WS + k] = v.x;
    w2t[(n4 * 4 + 1) * MG_WS + k] = v.y;
    w2t[(n4 * 4 + 2) * MG_WS + k] = v.z;
    w2t[(n4 * 4 + 3) * MG_WS + k] = v.w;
  }
  __syncthreads();

  const int r = lane & 15, q = lane >> 4;
  const int quad = wave >> 2, wq = wave & 3;             // two quads per workgroup, 64 hidden columns per wave
  const int nblocks = (a.n + 15) / 16;
  const int stride = gridDim.x * 2;
  int blk = blockIdx.x * 2 + quad;
  const int ncol0 = wq * 64;

  auto node = [&](int i) -> long long {                  // i-th row of the set -> node id (clamped: callers mask)
    int ii = i < a.n ? i : a.n - 1;
    return a.rows ? (long long)a.rows[ii] : (long long)(a.row0 + ii);
  };

  f32x4 gc[KB], gn[KB];
  auto load_g = [&](f32x4 (&dst)[KB], int b) {
    int i = b * 16 + r;
    const float* p = a.g + node(i) * a.ldg + 4 * q;
    const bool live = i < a.n;                           // rows past the end contribute zeros
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      f32x4 v = *reinterpret_cast<const f32x4*>(p + kb * 16);
      dst[kb] = live ? v : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };

  f32x4 acc2[FT][4];
  float cs[4];
#pragma unroll
  for (int f = 0; f < FT; ++f)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc2[f][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 4; ++j) cs[j] = 0.f;

  if (blk < nblocks) load_g(gc, blk);
  __builtin_amdgcn_s_waitcnt(0x0F70);                    // vmcnt(0): enter the loop with nothing pending
  while (blk < nblocks) {
    // ---- requests of this block: mask and X rows 4q..4q+3, then G of the next block
    long long nd[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) nd[t] = node(blk * 16 + 4 * q + t);
    float mk[4][4], xt[FT][4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int t = 0; t < 4; ++t) mk[j][t] = a.hid[nd[t] * a.ldh + ncol0 + j * 16 + r];
#pragma unroll
    for (int f = 0; f < FT; ++f)
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        int feat = f * 16 + r;
        xt[f][s] = feat < a.fin ? a.x[nd[s] * a.ldx + feat] : 0.f;
      }
    int nb = blk + stride;
    load_g(gn, nb < nblocks ? nb : nblocks - 1);

    // ---- MFMA #1: dH block (16 rows x 64 columns of this wave), W2^T fragments pipelined through two register sets
    int woff = 0;
    asm volatile("" : "+s"(woff));                       // opaque per block: keeps the LDS reads inside the loop
    const float* wl = w2t + woff;
    f32x4 acc1[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc1[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 w[2][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) w[0][j] = *reinterpret_cast<const f32x4*>(wl + (ncol0 + j * 16 + r) * MG_WS + 4 * q);
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      if (kb + 1 < KB) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          w[(kb + 1) & 1][j] = *reinterpret_cast<const f32x4*>(wl + (ncol0 + j * 16 + r) * MG_WS + (kb + 1) * 16 + 4 * q);
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (BF) {
        const s16x4 gp = pack_bf16x4(gc[kb]);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc1[j] = mfma_bf16_k16(gp, pack_bf16x4(w[kb & 1][j]), acc1[j]);
      } else {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            // i axis = rows (A = G), j axis = hidden columns (B = W2^T): lane gets rows 4q..4q+3 of column lane&15
            acc1[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(gc[kb][s], w[kb & 1][j][s], acc1[j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
          }
      }
    }

    // ---- ReLU mask, bias-gradient partials, MFMA #2 with the masked dH straight from the accumulators
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        float v = mk[j][t] > 0.f ? acc1[j][t] : 0.f;
        acc1[j][t] = v;
        cs[j] += v;
      }
    }
    if constexpr (BF) {
      s16x4 dp[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) dp[j] = pack_bf16x4(acc1[j]);
#pragma unroll
      for (int f = 0; f < FT; ++f) {
        const s16x4 xp = pack_bf16x4(xt[f][0], xt[f][1], xt[f][2], xt[f][3]);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc2[f][j] = mfma_bf16_k16(xp, dp[j], acc2[f][j]);
      }
    } else {
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int f = 0; f < FT; ++f)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc2[f][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(xt[f][s], acc1[j][s], acc2[f][j], 0, 0, 0);
    }

    __builtin_amdgcn_s_waitcnt(0x0F70);                  // next G block: requested a whole block ago
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) gc[kb] = gn[kb];
    blk = nb;
  }

  // ---- partial results of this quad: dW1[hidden][feature] (lane: features 4q..4q+3 of hidden column lane&15)
  const int z = blockIdx.x * 2 + quad;
  float* sw = a.slab_w + (long long)z * a.wstride;
#pragma unroll
  for (int f = 0; f < FT; ++f)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int hcol = ncol0 + j * 16 + r;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        int feat = f * 16 + 4 * q + t;
        if (feat < a.fin) sw[(long long)hcol * a.fin + feat] = acc2[f][j][t];
      }
    }
  float* sb = a.slab_b + (long long)z * a.bstride;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float v = cs[j];                                     // rows 4q..4q+3 of every block: sum the four quarters
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    if (q == 0) sb[ncol0 + j * 16 + r] = v;
  }
}

static int mlp_grads_grid() { return 256; }             // one persistent workgroup per CU (139 KB of LDS each)

}  // namespace mmft

using namespace mmft;

extern "C" long long mmft_mlp2_first_layer_grads_workspace_bytes(int fin, int HD) {
  return (long long)2 * mlp_grads_grid() * ((long long)HD * fin + HD) * 4;
}

extern "C" int mmft_mlp2_first_layer_grads(const float* g, long long ldg, const float* hid, long long ldh, const float* x,
                                           long long ldx, const int* rows, int row0, int n, const float* w2,
                                           long long ldw2, float* dw1, float* db1, int fin, int HD, int D2,
                                           int accumulate, float* workspace, long long workspace_bytes, int device,
                                           void* stream) {
  MMFT_REQUIRE(g && hid && x && w2 && dw1 && db1, "mlp2_first_layer_grads: null pointer");
  if (HD != MG_HD || D2 != MG_D2 || fin < 1 || fin > 48) {
    set_error("mlp2_first_layer_grads: only fin <= 48, hidden %d, out %d is fused (got %d, %d, %d)", MG_HD, MG_D2, fin,
              HD, D2);
    return MMFT_ERR_UNSUPPORTED;
  }
  MMFT_REQUIRE(n >= 0 && row0 >= 0, "mlp2_first_layer_grads: negative row count / offset");
  MMFT_REQUIRE(ldg >= D2 && ldh >= HD && ldx >= fin && ldw2 >= HD, "mlp2_first_layer_grads: leading dimensions");
  MMFT_REQUIRE(ldg % 4 == 0 && ldw2 % 4 == 0 && aligned16(g) && aligned16(w2),
               "mlp2_first_layer_grads: g and w2 must be 16-byte aligned with strides that are multiples of 4");
  DeviceGuard dg(device);
  hipStream_t st = (hipStream_t)stream;
  if (n == 0) {
    if (!accumulate) {
      (void)hipMemsetAsync(dw1, 0, (size_t)HD * fin * 4, st);
      (void)hipMemsetAsync(db1, 0, (size_t)HD * 4, st);
    }
    return MMFT_OK;
  }
  const int grid = mlp_grads_grid();
  long long need = mmft_mlp2_first_layer_grads_workspace_bytes(fin, HD);
  MMFT_REQUIRE(workspace && workspace_bytes >= need, "mlp2_first_layer_grads: workspace too small (%lld < %lld)",
               workspace_bytes, need);
  // db1 right behind dw1 (neighbours in the flat gradient buffer): interleave the slabs and reduce both in one launch
  const bool joined = db1 == dw1 + (long long)HD * fin && ((long long)HD * fin) % 4 == 0;
  const long long wslab = (long long)HD * fin + (joined ? HD : 0);
  float* slab_w = workspace;
  float* slab_b = joined ? workspace + (long long)HD * fin : workspace + (long long)2 * grid * HD * fin;
  MlpGradArgs a{g, ldg, hid, ldh, x, ldx, rows, row0, n, fin, w2, ldw2, slab_w, slab_b, wslab, joined ? wslab : (long long)HD};
  const size_t lds = (size_t)MG_HD * MG_WS * 4;
  const int ft = (fin + 15) / 16;
  const bool bf = math_mode() == MMFT_MATH_BF16;
  static DynLdsOnce once[2][4];
  int arc = MMFT_OK;
  auto set_attr = [&](const void* k) { arc = ensure_dyn_lds(once[bf][ft < 3 ? ft : 3], k, (int)lds, "mlp2_first_layer_grads"); };
  const double fl = 2.0 * n * ((double)D2 * HD + (double)HD * ft * 16), by = 4.0 * n * ((double)D2 + HD + fin);
  const dim3 gr(grid), bl(MG_WAVES * 64);
#define MMFT_MG(FTV, BFV, NAME)                                                                       \
  do {                                                                                                \
    set_attr((const void*)mlp_first_layer_grads_kernel<FTV, BFV>);                                    \
    if (arc) return arc;                                                                              \
    MMFT_LAUNCH_LDS(NAME, fl, by, (mlp_first_layer_grads_kernel<FTV, BFV>), gr, bl, lds, st, a);      \
  } while (0)
  if (bf) {
    if (ft == 1) MMFT_MG(1, true, "mlp_first_layer_grads_kernel<bf16>");
    else if (ft == 2) MMFT_MG(2, true, "mlp_first_layer_grads_kernel<bf16>");
    else MMFT_MG(3, true, "mlp_first_layer_grads_kernel<bf16>");
  } else {
    if (ft == 1) MMFT_MG(1, false, "mlp_first_layer_grads_kernel");
    else if (ft == 2) MMFT_MG(2, false, "mlp_first_layer_grads_kernel");
    else MMFT_MG(3, false, "mlp_first_layer_grads_kernel");
  }
#undef MMFT_MG
  int rc = check_launch("mlp2_first_layer_grads");
  if (rc) return rc;
  if (joined) return launch_slab_reduce(slab_w, 2 * grid, wslab, dw1, accumulate, st);
  rc = launch_slab_reduce(slab_w, 2 * grid, (long long)HD * fin, dw1, accumulate, st);
  if (rc) return rc;
  return launch_slab_reduce(slab_b, 2 * grid, HD, db1, accumulate, st);
}

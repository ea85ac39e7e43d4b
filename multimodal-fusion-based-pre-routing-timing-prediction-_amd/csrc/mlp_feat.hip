// The feature MLPs of the sweep, fc_cell_self / fc_net_self (Linear(fin, 256) - ReLU - Linear(256, 128) over ALL cell /
// net nodes, src/model.py:48-51,66-67,148-153,186-189), bf16 math mode, WITHOUT storing the hidden activations.
//
// As two GEMMs per MLP the hidden tensor H (rows x 256 fp32 = 268 MB per MLP at config B) was written once and read three
// times per step (second forward GEMM, weight gradient of layer 2, fused first-layer gradients): 4 x 111 us forward,
// 2 x (120 + 103) us backward, all of it HBM time.  fin is 36 / 2: recomputing H costs 1-5 % of the MLP's flops, so
//   * forward  (mmft_mlp2_feat_fwd_bf16): one kernel per MLP reads X, keeps the 64 x 256 hidden tile in LDS as bf16 and
//     writes only the output rows;
//   * backward (mmft_mlp2_feat_bwd_bf16): one kernel per MLP reads G (gradient of the output rows) and X, recomputes
//     H = relu(X W1^T + b1) with the forward's instruction sequence (bitwise the forward's H, so the ReLU mask agrees),
//     forms dH = (G W2) * (H > 0) in registers and accumulates all four gradients
//         dW2 += G^T H,  db2 += sum G,  dW1 += dH^T X,  db1 += sum dH
//     in MFMA accumulators across the 32-row tiles of a persistent workgroup (the contractions over rows take their
//     operands from transposed bf16 copies of the tile in LDS); one slab per workgroup, summed in a fixed order.
// Lane layouts are those of mlp2_bf16.hip: A = weights / transposed tiles (lane = output feature, 8 consecutive k),
// B = row tiles (lane = row or feature, 8 consecutive k), D[m][n] at lane (n = lane & 15, q = lane >> 4) = rows 4q..4q+3.
#include <stdlib.h>
#include "gemm_bf16.h"
#include "tr_read.h"

namespace mmft {

int launch_slab_reduce(const float* slabs, int splits, long long elems, float* out, int accumulate, hipStream_t st);

constexpr int MF_HD = 256, MF_D2 = 128;

__device__ __forceinline__ unsigned short mf_bf16(float v) { return (unsigned short)(pack_bf16(v, 0.f) & 0xffff); }

__device__ __forceinline__ bf16x8 mf_pack8(const float* v) {
  u32x4 p = {pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]), pack_bf16(v[4], v[5]), pack_bf16(v[6], v[7])};
  return __builtin_bit_cast(bf16x8, p);
}

// A fragment of W1 [HD][fin] for hidden column `col`, k = k0 .. k0 + 7 (zero beyond fin)
__device__ __forceinline__ bf16x8 mf_w1_frag(const float* __restrict__ w1, int fin, int col, int k0) {
  float v[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) v[t] = k0 + t < fin ? w1[(long long)col * fin + k0 + t] : 0.f;
  return mf_pack8(v);
}

struct FeatFwdArgs {
  const float* x;
  long long ldx;
  int row0, n, fin;
  const float *w1, *b1, *w2, *b2;
  float* out;
  long long ldout;
  int relu_out;
};

template <int KS>   // K steps of 32 for the first layer (fin <= 32 KS)
__global__ void __launch_bounds__(512) mlp2_feat_fwd_kernel(FeatFwdArgs a) {
  constexpr int BM = 64, RT = BM / 16, XS = KS * 32 + 8, HS = MF_HD + 8;
  __shared__ __attribute__((aligned(16))) unsigned short xs[BM * XS];
  __shared__ __attribute__((aligned(16))) unsigned short hs[BM * HS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, q = lane >> 4;
  bf16x8 w1f[2][KS], w2f[8];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) w1f[j][ks] = mf_w1_frag(a.w1, a.fin, wave * 32 + j * 16 + r16, ks * 32 + q * 8);
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) w2f[ks] = mf_pack8(a.w2 + (long long)(wave * 16 + r16) * MF_HD + ks * 32 + q * 8);
  const int nn2 = wave * 16 + q * 4;
  const f32x4 b2v = *reinterpret_cast<const f32x4*>(a.b2 + nn2);
  f32x4 b1v[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) b1v[j] = *reinterpret_cast<const f32x4*>(a.b1 + wave * 32 + j * 16 + q * 4);
  // The weights (164 KB of fp32) are fetched ONCE per workgroup: the grid is a few workgroups per CU and each walks
  // tiles of 64 rows; the X elements of tile t + 1 are requested before tile t is multiplied.
  constexpr int NXE = BM * KS * 32 / 512;
  float xr[NXE];
  auto request = [&](int tile) {
#pragma unroll
    for (int k = 0; k < NXE; ++k) {
      const int e = tid + k * 512, r = e / (KS * 32), kk = e % (KS * 32);
      const int m = tile * BM + r;
      xr[k] = (m < a.n && kk < a.fin) ? a.x[(long long)(a.row0 + m) * a.ldx + kk] : 0.f;
    }
  };
  const int ntiles = (a.n + BM - 1) / BM;
  if ((int)blockIdx.x < ntiles) request(blockIdx.x);
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int m0 = tile * BM;
#pragma unroll
    for (int k = 0; k < NXE; ++k) {
      const int e = tid + k * 512;
      xs[(e / (KS * 32)) * XS + e % (KS * 32)] = mf_bf16(xr[k]);
    }
    __syncthreads();
    if (tile + (int)gridDim.x < ntiles) request(tile + gridDim.x);
    f32x4 acc1[RT][2];
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc1[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int i = 0; i < RT; ++i) {
        const bf16x8 xf = *reinterpret_cast<const bf16x8*>(xs + (i * 16 + r16) * XS + ks * 32 + q * 8);
#pragma unroll
        for (int j = 0; j < 2; ++j) acc1[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1f[j][ks], xf, acc1[i][j], 0, 0, 0);
      }
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int m = i * 16 + r16, nn = wave * 32 + j * 16 + q * 4;
        f32x4 v = acc1[i][j] + b1v[j];
        v.x = v.x > 0.f ? v.x : 0.f; v.y = v.y > 0.f ? v.y : 0.f;
        v.z = v.z > 0.f ? v.z : 0.f; v.w = v.w > 0.f ? v.w : 0.f;
        const unsigned lo = pack_bf16(v.x, v.y), hi = pack_bf16(v.z, v.w);
        *reinterpret_cast<unsigned long long*>(hs + m * HS + nn) = ((unsigned long long)hi << 32) | lo;
      }
    __syncthreads();
    f32x4 acc2[RT];
#pragma unroll
    for (int i = 0; i < RT; ++i) acc2[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
#pragma unroll
      for (int i = 0; i < RT; ++i) {
        const bf16x8 hf = *reinterpret_cast<const bf16x8*>(hs + (i * 16 + r16) * HS + ks * 32 + q * 8);
        acc2[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2f[ks], hf, acc2[i], 0, 0, 0);
      }
#pragma unroll
    for (int i = 0; i < RT; ++i) {
      const int m = m0 + i * 16 + r16;
      if (m >= a.n) continue;
      f32x4 v = acc2[i] + b2v;
      if (a.relu_out) {
        v.x = v.x > 0.f ? v.x : 0.f; v.y = v.y > 0.f ? v.y : 0.f;
        v.z = v.z > 0.f ? v.z : 0.f; v.w = v.w > 0.f ? v.w : 0.f;
      }
      *reinterpret_cast<f32x4*>(a.out + (long long)(a.row0 + m) * a.ldout + nn2) = v;
    }
    // the next deposit overwrites xs (last read before the barrier above) and the next epilogue hs (last read here)
    __syncthreads();
  }
}

struct FeatBwdArgs {
  const float* g;
  long long ldg;
  const float* x;
  long long ldx;
  int row0, n, fin;
  const float *w1, *b1, *w2;
  float* slabs;        // [gridDim.x][slab]: dw1 [HD][fin] | db1 [HD] | dw2 [D2][HD] | db2 [D2]
  long long slab;
};

template <int KS>
__global__ void __launch_bounds__(512) mlp2_feat_bwd_kernel(FeatBwdArgs a) {
  // every tile lives in LDS in its natural [row][column] layout; the contractions over the tile's rows read their operands
  // with the hardware-transposed read (tr_read.h) - pitches of 16 x odd elements
  constexpr int BM = 32, KP = KS * 32, XS = KP + 16, GS = MF_D2 + 16, HS = MF_HD + 16, KB = KP / 16;
  extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
  unsigned short* xs = lds;                       // X tile  [row][k]
  unsigned short* gs = xs + BM * XS;              // G tile  [row][d2]
  unsigned short* hs = gs + BM * GS;              // H tile  [row][col]
  unsigned short* ds = hs + BM * HS;              // dH tile [row][col]
  f32x4(*gred)[32] = reinterpret_cast<f32x4(*)[32]>(hs);      // end of the kernel only (16 x 32 x 16 B <= hs)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, q = lane >> 4;
  const int trr = tr_lane_row(lane), trc = tr_lane_col(lane);
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};

  // weights in registers for the life of the workgroup
  bf16x8 w1f[2][KS], w2tf[2][4];
  f32x4 b1v[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = wave * 32 + j * 16 + r16;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) w1f[j][ks] = mf_w1_frag(a.w1, a.fin, col, ks * 32 + q * 8);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {         // A fragment of W2^T: W2T[col][d2] = w2[d2][col], d2 = 32 ks + 8 q ..
      float v[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) v[t] = a.w2[(long long)(ks * 32 + q * 8 + t) * MF_HD + col];
      w2tf[j][ks] = mf_pack8(v);
    }
    b1v[j] = *reinterpret_cast<const f32x4*>(a.b1 + wave * 32 + j * 16 + q * 4);
  }
  f32x4 dw2[2][8], dw1[2][KB], dsum[2], gsum = zero;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    dsum[j] = zero;
#pragma unroll
    for (int d = 0; d < 8; ++d) dw2[j][d] = zero;
#pragma unroll
    for (int k = 0; k < KB; ++k) dw1[j][k] = zero;
  }

  const int ntiles = (a.n + BM - 1) / BM;
  // G (two 16-byte groups per thread) and X (BM * KP / 512 scalars) of tile t + 1 are requested before tile t is multiplied
  constexpr int NXE = BM * KP / 512;
  f32x4 gr[2];
  float xr[NXE];
  auto request = [&](int tile) {
    const int m0 = tile * BM;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int e = tid + it * 512, r = e >> 5, c = (e & 31) * 4;
      gr[it] = m0 + r < a.n ? *reinterpret_cast<const f32x4*>(a.g + (long long)(a.row0 + m0 + r) * a.ldg + c) : zero;
    }
#pragma unroll
    for (int k = 0; k < NXE; ++k) {
      const int e = tid + k * 512, r = e / KP, kk = e % KP;
      xr[k] = (m0 + r < a.n && kk < a.fin) ? a.x[(long long)(a.row0 + m0 + r) * a.ldx + kk] : 0.f;
    }
  };
  if ((int)blockIdx.x < ntiles) request(blockIdx.x);
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    // ---- deposit G and X; rows past the end are zero
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int e = tid + it * 512, r = e >> 5, c = (e & 31) * 4;          // (row, 4 consecutive d2): c is the same for both
      const f32x4 v = gr[it];
      gsum += v;
      const unsigned lo = pack_bf16(v.x, v.y), hi = pack_bf16(v.z, v.w);
      *reinterpret_cast<unsigned long long*>(gs + r * GS + c) = ((unsigned long long)hi << 32) | lo;
    }
#pragma unroll
    for (int k = 0; k < NXE; ++k) {
      const int e = tid + k * 512, r = e / KP, kk = e % KP;
      xs[r * XS + kk] = mf_bf16(xr[k]);
    }
    __syncthreads();
    if (tile + (int)gridDim.x < ntiles) request(tile + gridDim.x);
    __syncthreads();
    // ---- H = relu(X W1^T + b1), dH = (G W2) * (H > 0): hidden columns [32 wave, 32 wave + 32) of the 32 rows
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      f32x4 acc1[2] = {zero, zero}, acc3[2] = {zero, zero};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const bf16x8 xf = *reinterpret_cast<const bf16x8*>(xs + (i * 16 + r16) * XS + ks * 32 + q * 8);
#pragma unroll
        for (int j = 0; j < 2; ++j) acc1[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1f[j][ks], xf, acc1[j], 0, 0, 0);
      }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const bf16x8 gf = *reinterpret_cast<const bf16x8*>(gs + (i * 16 + r16) * GS + ks * 32 + q * 8);
#pragma unroll
        for (int j = 0; j < 2; ++j) acc3[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2tf[j][ks], gf, acc3[j], 0, 0, 0);
      }
      const int m = i * 16 + r16;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int nn = wave * 32 + j * 16 + q * 4;
        f32x4 hv = acc1[j] + b1v[j], dh = acc3[j];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          hv[t] = hv[t] > 0.f ? hv[t] : 0.f;
          dh[t] = hv[t] > 0.f ? dh[t] : 0.f;
        }
        dsum[j] += dh;
        const unsigned hl = pack_bf16(hv.x, hv.y), hh = pack_bf16(hv.z, hv.w);
        const unsigned dl = pack_bf16(dh.x, dh.y), dhh = pack_bf16(dh.z, dh.w);
        *reinterpret_cast<unsigned long long*>(hs + m * HS + nn) = ((unsigned long long)hh << 32) | hl;
        *reinterpret_cast<unsigned long long*>(ds + m * HS + nn) = ((unsigned long long)dhh << 32) | dl;
      }
    }
    __syncthreads();
    // ---- contractions over the tile's 32 rows (one K step): dW2^T[col][d2] += H^T G, dW1^T[col][k] += dH^T X
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col0 = wave * 32 + j * 16;
      const bf16x8 hf = __builtin_bit_cast(bf16x8, tr_read_pair(hs + trr * HS + col0 + trc, 16 * HS));
      const bf16x8 df = __builtin_bit_cast(bf16x8, tr_read_pair(ds + trr * HS + col0 + trc, 16 * HS));
#pragma unroll
      for (int d = 0; d < 8; ++d) {
        const bf16x8 gf = __builtin_bit_cast(bf16x8, tr_read_pair(gs + trr * GS + d * 16 + trc, 16 * GS));
        dw2[j][d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(hf, gf, dw2[j][d], 0, 0, 0);
      }
#pragma unroll
      for (int k = 0; k < KB; ++k) {
        const bf16x8 xf = __builtin_bit_cast(bf16x8, tr_read_pair(xs + trr * XS + k * 16 + trc, 16 * XS));
        dw1[j][k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(df, xf, dw1[j][k], 0, 0, 0);
      }
    }
    __syncthreads();
  }

  // ---- the workgroup's slab
  float* sl = a.slabs + (long long)blockIdx.x * a.slab;
  float* s_dw1 = sl;
  float* s_db1 = s_dw1 + (long long)MF_HD * a.fin;
  float* s_dw2 = s_db1 + MF_HD;
  float* s_db2 = s_dw2 + (long long)MF_D2 * MF_HD;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int colq = wave * 32 + j * 16 + q * 4;           // D[m = col][n]: lane (n = r16, q) holds cols colq .. colq + 3
#pragma unroll
    for (int d = 0; d < 8; ++d) *reinterpret_cast<f32x4*>(s_dw2 + (long long)(d * 16 + r16) * MF_HD + colq) = dw2[j][d];
#pragma unroll
    for (int k = 0; k < KB; ++k) {
      const int kk = k * 16 + r16;
      if (kk < a.fin) {
#pragma unroll
        for (int t = 0; t < 4; ++t) s_dw1[(long long)(colq + t) * a.fin + kk] = dw1[j][k][t];
      }
    }
    // db1: sum over the 16 row lanes (rows live on r16), fixed butterfly order
    f32x4 v = dsum[j];
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) {
#pragma unroll
      for (int t = 0; t < 4; ++t) v[t] += __shfl_xor(v[t], o, 64);
    }
    if (r16 == 0) *reinterpret_cast<f32x4*>(s_db1 + colq) = v;
  }
  // db2: thread t summed d2 group (t & 31) over its rows; 16 threads share a group
  __syncthreads();
  gred[tid >> 5][tid & 31] = gsum;
  __syncthreads();
  if (tid < 32) {
    f32x4 v = gred[0][tid];
    for (int p = 1; p < 16; ++p) v += gred[p][tid];
    *reinterpret_cast<f32x4*>(s_db2 + tid * 4) = v;
  }
}

template <int KS>
constexpr int feat_bwd_lds() {
  constexpr int BM = 32, KP = KS * 32, XS = KP + 16, GS = MF_D2 + 16, HS = MF_HD + 16;
  return (BM * XS + BM * GS + 2 * BM * HS) * 2;
}

static inline long long feat_slab(int fin) { return (long long)MF_HD * fin + MF_HD + (long long)MF_D2 * MF_HD + MF_D2; }
// One workgroup per CU (256 VGPRs) - on 192 of the 256 CUs: the two launches close the reverse sweep on its side stream while
// the U-Net's backward, the stream that finishes last, still runs on the main one; a grid that takes every CU for 2 x 118 us
// stalls it.  Measured on the replayed config-B step (ms): 256 -> 3.37, 224 -> 3.28, 192 -> 3.23, 160 -> 3.26, 128 -> 3.29.
// MMFT_FEAT_BWD_GRID overrides (read once).
static inline int feat_bwd_grid(int n) {
  int tiles = cdiv(n, 32);
  static int cap = getenv("MMFT_FEAT_BWD_GRID") ? atoi(getenv("MMFT_FEAT_BWD_GRID")) : 192;
  if (cap < 1) cap = 1;
  return tiles < cap ? (tiles < 1 ? 1 : tiles) : cap;
}

}  // namespace mmft

using namespace mmft;

extern "C" int mmft_mlp2_feat_fwd_bf16(const float* x, long long ldx, int row0, int n, int fin, const float* w1, const float* b1,
                                       const float* w2, const float* b2, float* out, long long ldout, int relu_out, int device,
                                       void* stream) {
  MMFT_REQUIRE(x && w1 && b1 && w2 && b2 && out, "mlp2_feat_fwd_bf16: null pointer");
  MMFT_REQUIRE(n >= 0 && row0 >= 0 && fin >= 1 && fin <= 64 && ldx >= fin, "mlp2_feat_fwd_bf16: bad sizes (fin <= 64)");
  MMFT_REQUIRE(ldout % 4 == 0 && aligned16(out) && aligned16(w2) && aligned16(b1) && aligned16(b2),
               "mlp2_feat_fwd_bf16: out / w2 / biases must be 16-byte aligned");
  if (n == 0) return MMFT_OK;
  DeviceGuard dg(device);
  FeatFwdArgs a{x, ldx, row0, n, fin, w1, b1, w2, b2, out, ldout, relu_out};
  const double fl = 2.0 * n * ((double)fin * MF_HD + (double)MF_HD * MF_D2), by = 4.0 * n * ((double)fin + MF_D2);
  int grid = cdiv(n, 64);
  if (grid > 768) grid = 768;                      // three workgroups per CU (43 KB of LDS each)
  if (fin <= 32)
    MMFT_LAUNCH("mlp2_feat_fwd_kernel", fl, by, mlp2_feat_fwd_kernel<1>, dim3(grid), dim3(512), (hipStream_t)stream, a);
  else
    MMFT_LAUNCH("mlp2_feat_fwd_kernel", fl, by, mlp2_feat_fwd_kernel<2>, dim3(grid), dim3(512), (hipStream_t)stream, a);
  return check_launch("mlp2_feat_fwd_bf16");
}

extern "C" long long mmft_mlp2_feat_bwd_workspace_bytes(int n, int fin) {
  if (n <= 0 || fin < 1 || fin > 64) return 0;
  return (long long)feat_bwd_grid(n) * feat_slab(fin) * 4;
}

extern "C" int mmft_mlp2_feat_bwd_bf16(const float* g, long long ldg, const float* x, long long ldx, int row0, int n, int fin,
                                       const float* w1, const float* b1, const float* w2, float* dw1, float* db1, float* dw2,
                                       float* db2, int accumulate, float* workspace, long long workspace_bytes, int device,
                                       void* stream) {
  MMFT_REQUIRE(g && x && w1 && b1 && w2 && dw1 && db1 && dw2 && db2, "mlp2_feat_bwd_bf16: null pointer");
  MMFT_REQUIRE(n > 0 && row0 >= 0 && fin >= 1 && fin <= 64 && ldx >= fin && ldg >= MF_D2, "mlp2_feat_bwd_bf16: bad sizes (fin <= 64)");
  MMFT_REQUIRE(ldg % 4 == 0 && aligned16(g) && aligned16(b1), "mlp2_feat_bwd_bf16: g / b1 must be 16-byte aligned");
  MMFT_REQUIRE(workspace && aligned16(workspace) && workspace_bytes >= mmft_mlp2_feat_bwd_workspace_bytes(n, fin),
               "mlp2_feat_bwd_bf16: workspace too small");
  DeviceGuard dg(device);
  hipStream_t st = (hipStream_t)stream;
  const int grid = feat_bwd_grid(n);
  const long long slab = feat_slab(fin);
  FeatBwdArgs a{g, ldg, x, ldx, row0, n, fin, w1, b1, w2, workspace, slab};
  const double fl = 2.0 * n * (2.0 * fin * MF_HD + 2.0 * MF_HD * MF_D2), by = 4.0 * n * ((double)fin + MF_D2);
  static DynLdsOnce once[2];
  int arc = fin <= 32 ? ensure_dyn_lds(once[0], reinterpret_cast<const void*>(&mlp2_feat_bwd_kernel<1>), feat_bwd_lds<1>(), "mlp2_feat_bwd_bf16")
                      : ensure_dyn_lds(once[1], reinterpret_cast<const void*>(&mlp2_feat_bwd_kernel<2>), feat_bwd_lds<2>(), "mlp2_feat_bwd_bf16");
  if (arc) return arc;
  if (fin <= 32)
    MMFT_LAUNCH_LDS("mlp2_feat_bwd_kernel", fl, by, mlp2_feat_bwd_kernel<1>, dim3(grid), dim3(512), feat_bwd_lds<1>(), st, a);
  else
    MMFT_LAUNCH_LDS("mlp2_feat_bwd_kernel", fl, by, mlp2_feat_bwd_kernel<2>, dim3(grid), dim3(512), feat_bwd_lds<2>(), st, a);
  int rc = check_launch("mlp2_feat_bwd_bf16");
  if (rc) return rc;
  // The four gradients are the four segments of a slab.  In FlatAdam's gradient buffer they are neighbours in exactly that
  // order (weight, bias, weight, bias of one MLP), so ONE parallel slab reduction finishes all of them.
  if (db1 == dw1 + (long long)MF_HD * fin && dw2 == db1 + MF_HD && db2 == dw2 + (long long)MF_D2 * MF_HD && aligned16(dw1))
    return launch_slab_reduce(workspace, grid, slab, dw1, accumulate, st);
  // scattered outputs: the same four reductions as one batched launch - per element the summation order of the joined form,
  // so the gradients do not depend on where the caller's (or the allocator's) tensors happen to lie
  const long long o1 = (long long)MF_HD * fin, o2 = o1 + MF_HD, o3 = o2 + (long long)MF_D2 * MF_HD;
  const SlabSeg segs[4] = {{workspace, dw1, slab, grid, (int)o1, 1, accumulate ? 1 : 0},
                           {workspace + o1, db1, slab, grid, MF_HD, 1, accumulate ? 1 : 0},
                           {workspace + o2, dw2, slab, grid, MF_D2 * MF_HD, 1, accumulate ? 1 : 0},
                           {workspace + o3, db2, slab, grid, MF_D2, 1, accumulate ? 1 : 0}};
  return launch_slab_reduce_batch(segs, 4, st);
}

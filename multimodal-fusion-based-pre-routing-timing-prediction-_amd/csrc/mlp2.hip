// Fused two-layer row kernel for the level-serial chain of the netlist sweep (K4 in SURVEY.md §2.4):
//
//   hid = mask ? (x1[rows] . W1) * (mask[rows] > 0)          (reverse sweep: dHn = (G W2g) * relu'(HN))
//              : relu(x1[rows] . W1^T + b1)                  (forward: HN = relu(fc_cell_neigh.0(A)))
//   out[rows] = epi(hid . W2 (+ b2))                         (forward: h = relu(h + ...); reverse: DA = ...)
//
// One workgroup owns 32 rows; the 32 x 256 hidden tile never leaves the CU (LDS), so a cell level costs one
// launch instead of two GEMM launches and the hidden activations are not re-read from HBM.  Replaces the pair of
// th.nn.Linear calls inside PathConv.apply_cell_func (reference src/model.py:138-146) and its autograd mirror.
// Same fp32 MFMA fragments, LDS strides and k permutation as gemm_engine.h; 128 -> 256 -> 128 widths only
// (the reference's PathConv defaults, src/model.py:48-51); other widths use the two-launch path.
// 8 waves per workgroup (two per SIMD): a level is one round of <= 256 workgroups, so its duration is one workgroup's
// latency; halving every wave's share of the MFMAs and of the register-resident weight panels took 24 -> 19 us
// (MMFT_MLP2_WAVES=4 restores the four-wave form for comparison).
#include "mlp2_core.h"

namespace mmft {

// BF: operands rounded to bf16 at the MFMA (v_mfma_f32_16x16x16_bf16, fp32 accumulate) - MMFT_MATH_BF16
template <bool KM, int NW, bool BF>   // NW waves per workgroup: each owns 256 / NW hidden columns and 128 / NW output columns
__global__ void __launch_bounds__(NW * 64, 1) mlp2_rows_kernel(Mlp2Args a) {
  constexpr int NT = NW * 64, HC = (M2_HD / 16) / NW, OC = (M2_D2 / 16) / NW;
  constexpr int XS = M2_K1 + 8;                       // 136: x1 tile, whole K resident
  constexpr int HS = M2_HD + 8;                       // 264: hidden tile
  constexpr int W1S = KM ? (M2_HD + 4) : (M2_BK + 8);
  constexpr int W2S = KM ? (M2_D2 + 4) : (M2_BK + 8);
  constexpr int W1SZ = KM ? M2_BK * W1S : M2_HD * W1S;
  constexpr int W2SZ = KM ? M2_BK * W2S : M2_D2 * W2S;
  constexpr int WSZ = W1SZ > W2SZ ? W1SZ : W2SZ;
  __shared__ __attribute__((aligned(16))) float lds[M2_BM * XS + M2_BM * HS + 2 * WSZ];
  float* xs = lds;
  float* hs = lds + M2_BM * XS;
  float* wb = hs + M2_BM * HS;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = blockIdx.x * M2_BM;
  auto live_row = [&](int r) -> bool { return m0 + r < a.n && (!a.active || a.active[a.rows[m0 + r]]); };
  if (a.active) {
    // a tile without a single row inside the step's fan-in cone has nothing to do
    int any = 0;
    if (tid < M2_BM) any = live_row(tid) ? 1 : 0;
    if (!__syncthreads_or(any)) return;
  }

  // ---- every global load of the kernel is issued here: both weight panels and the gathered x1 rows
  WPanel<KM, M2_HD, M2_K1 / M2_BK, NT> p1;
  WPanel<KM, M2_D2, M2_HD / M2_BK, NT> p2;
  p1.load(a.w1, a.ldw1, tid);
  f32x4 xr[M2_BM * M2_K1 / 4 / NT];
#pragma unroll
  for (int i = 0; i < M2_BM * M2_K1 / 4 / NT; ++i) {
    int g = tid + i * NT;
    int r = g / (M2_K1 / 4), k4 = g % (M2_K1 / 4);
    xr[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (live_row(r)) xr[i] = *reinterpret_cast<const f32x4*>(a.x1 + (long long)a.rows[m0 + r] * a.ldx1 + k4 * 4);
  }
  p2.load(a.w2, a.ldw2, tid);
#pragma unroll
  for (int i = 0; i < M2_BM * M2_K1 / 4 / NT; ++i) {
    int g = tid + i * NT;
    int r = g / (M2_K1 / 4), k4 = g % (M2_K1 / 4);
    *reinterpret_cast<f32x4*>(xs + r * XS + k4 * 4) = xr[i];
  }

  // ---- phase 1: hid[32 x 256]; wave w owns hidden columns [16 HC w, 16 HC (w + 1))
  f32x4 acc1[2][HC];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < HC; ++j) acc1[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  Phase1<KM, 0, M2_K1 / M2_BK, false, NW, BF>::run(p1, xs, wb, WSZ, tid, lane, wave, acc1);

  // ---- epilogue 1: bias + ReLU, or ReLU mask from the saved forward hidden activations -> LDS (+ HBM)
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < HC; ++j) {
      int m = i * 16 + (lane & 15);
      int nn = wave * (HC * 16) + j * 16 + (lane >> 4) * 4;
      f32x4 v = acc1[i][j];
      bool live = live_row(m);
      long long row = live ? (long long)a.rows[m0 + m] : 0;
      if (a.mask) {
        f32x4 mk = {0.f, 0.f, 0.f, 0.f};
        if (live) mk = *reinterpret_cast<const f32x4*>(a.mask + row * a.ldmask + nn);
        v.x = mk.x > 0.f ? v.x : 0.f; v.y = mk.y > 0.f ? v.y : 0.f;
        v.z = mk.z > 0.f ? v.z : 0.f; v.w = mk.w > 0.f ? v.w : 0.f;
      } else {
        if (a.b1) {
          v.x += a.b1[nn]; v.y += a.b1[nn + 1]; v.z += a.b1[nn + 2]; v.w += a.b1[nn + 3];
        }
        v.x = v.x > 0.f ? v.x : 0.f; v.y = v.y > 0.f ? v.y : 0.f;
        v.z = v.z > 0.f ? v.z : 0.f; v.w = v.w > 0.f ? v.w : 0.f;
      }
      *reinterpret_cast<f32x4*>(hs + m * HS + nn) = v;
      if (a.hid_out && live) *reinterpret_cast<f32x4*>(a.hid_out + row * a.ldhid + nn) = v;
    }
  __syncthreads();   // hidden tile complete; phase 1's last weight tile no longer read

  // ---- phase 2: out[32 x 128]; wave w owns output columns [16 OC w, 16 OC (w + 1))
  f32x4 acc2[2][OC];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < OC; ++j) acc2[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  Phase2<KM, 0, M2_HD / M2_BK, false, NW, BF>::run(p2, hs, wb, WSZ, tid, lane, wave, acc2);

  // ---- epilogue 2: row scatter by node id
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < OC; ++j) {
      int m = i * 16 + (lane & 15);
      if (!live_row(m)) continue;
      int nn = wave * (OC * 16) + j * 16 + (lane >> 4) * 4;
      float* q = a.out + (long long)a.rows[m0 + m] * a.ldout + nn;
      f32x4 v = acc2[i][j];
      if (a.b2) {
        v.x += a.b2[nn]; v.y += a.b2[nn + 1]; v.z += a.b2[nn + 2]; v.w += a.b2[nn + 3];
      }
      if (a.add_act) v += *reinterpret_cast<const f32x4*>(q);
      if (a.relu_out) {
        v.x = v.x > 0.f ? v.x : 0.f; v.y = v.y > 0.f ? v.y : 0.f;
        v.z = v.z > 0.f ? v.z : 0.f; v.w = v.w > 0.f ? v.w : 0.f;
      }
      *reinterpret_cast<f32x4*>(q) = v;
    }
}

}  // namespace mmft

using namespace mmft;

extern "C" int mmft_mlp2_rows(const float* x1, long long ldx1, const int* rows, int n, const float* w1, long long ldw1,
                              const float* b1, const float* w2, long long ldw2, const float* b2, int weights_kmajor,
                              const float* mask, long long ldmask, float* hid_out, long long ldhid, float* out,
                              long long ldout, int add_act, int relu_out, int K1, int HD, int D2,
                              const unsigned char* active, int device, void* stream) {
  MMFT_REQUIRE(x1 && rows && w1 && w2 && out, "mlp2_rows: null pointer");
  if (K1 != M2_K1 || HD != M2_HD || D2 != M2_D2) {
    set_error("mlp2_rows: only %d -> %d -> %d is fused (got %d -> %d -> %d)", M2_K1, M2_HD, M2_D2, K1, HD, D2);
    return MMFT_ERR_UNSUPPORTED;
  }
  MMFT_REQUIRE(n >= 0, "mlp2_rows: negative row count");
  MMFT_REQUIRE(ldx1 % 4 == 0 && ldw1 % 4 == 0 && ldw2 % 4 == 0 && ldout % 4 == 0 && aligned16(x1) && aligned16(w1) &&
                   aligned16(w2) && aligned16(out),
               "mlp2_rows: operands must be 16-byte aligned");
  MMFT_REQUIRE(!mask || (ldmask % 4 == 0 && aligned16(mask)), "mlp2_rows: mask alignment");
  MMFT_REQUIRE(!hid_out || (ldhid % 4 == 0 && aligned16(hid_out)), "mlp2_rows: hid_out alignment");
  MMFT_REQUIRE(weights_kmajor ? (ldw1 >= HD && ldw2 >= D2) : (ldw1 >= K1 && ldw2 >= HD), "mlp2_rows: weight strides");
  if (n == 0) return MMFT_OK;
  DeviceGuard dg(device);
  hipStream_t st = (hipStream_t)stream;
  Mlp2Args a{x1, ldx1, rows, n, w1, ldw1, b1, w2, ldw2, b2, mask, ldmask, hid_out, ldhid, out, ldout, add_act, relu_out, active};
  const double fl = 2.0 * n * ((double)K1 * HD + (double)HD * D2), by = 4.0 * n * ((double)K1 + 2.0 * HD + 2.0 * D2);
  static int nw = -1;
  if (nw < 0) {
    const char* e = getenv("MMFT_MLP2_WAVES");         // tuning hook: 4 or 8 waves per workgroup
    nw = (e && atoi(e) == 4) ? 4 : 8;
  }
  const bool bf = math_mode() == MMFT_MATH_BF16;
#define MMFT_M2(KM, NWV, BFV, NAME) \
  MMFT_LAUNCH(NAME, fl, by, (mlp2_rows_kernel<KM, NWV, BFV>), dim3(cdiv(n, M2_BM)), dim3(NWV * 64), st, a)
  if (weights_kmajor) {
    if (bf) { if (nw == 8) MMFT_M2(true, 8, true, "mlp2_rows_kernel<KM,bf16>"); else MMFT_M2(true, 4, true, "mlp2_rows_kernel<KM,bf16>"); }
    else { if (nw == 8) MMFT_M2(true, 8, false, "mlp2_rows_kernel<KM>"); else MMFT_M2(true, 4, false, "mlp2_rows_kernel<KM>"); }
  } else {
    if (bf) { if (nw == 8) MMFT_M2(false, 8, true, "mlp2_rows_kernel<MK,bf16>"); else MMFT_M2(false, 4, true, "mlp2_rows_kernel<MK,bf16>"); }
    else { if (nw == 8) MMFT_M2(false, 8, false, "mlp2_rows_kernel<MK>"); else MMFT_M2(false, 4, false, "mlp2_rows_kernel<MK>"); }
  }
#undef MMFT_M2
  return check_launch("mlp2_rows");
}

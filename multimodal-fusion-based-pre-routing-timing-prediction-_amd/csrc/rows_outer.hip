// dW[o][i] = sum over rows r of g[r][o] * x[r][i]   (+ db[o] = sum_r g[r][o]) for the two weight gradients of fc_cell_neigh
// (Linear(128, 256) - ReLU - Linear(256, 128) over ALL cell nodes of the batch, src/model.py:100-117,138-146): out x in =
// 256 x 128 and 128 x 256 with ~250 000 rows - a product whose reduction index is the ROW, so both operands are "k-major".
// bf16 math mode.
//
// The generic engine (gemm_bf16.h, DenseKM x DenseKM) transposes such operands in registers while staging (8 loads per thread
// per 16-byte column, 100-125 us per launch, bound by load issue).  Here a 32-row chunk of g and x is copied to LDS in its
// natural [row][column] layout (fp32 -> bf16 at staging, 16-byte loads, 8-byte LDS stores) and every MFMA operand is one
// pair of ds_read_b64_tr_b16 - the hardware transposed read: per 16-lane group a 4 row x 16 column block delivered
// column-major, i.e. "lane = output / input feature, registers = 4 consecutive rows", which is what
// v_mfma_f32_16x16x32_bf16 wants on both sides (k slots (g, 0..3) = rows 4g.., (g, 4..7) = rows 16 + 4g..).
// A workgroup keeps the whole out x in product in its accumulators (4 waves as 2 x 2, 32 fragments each), walks its share
// of the row chunks with the next chunk's loads in flight, and writes ONE slab; slabs are added in a fixed order.
#include "unet16.h"

namespace mmft {

constexpr int RO_KT = 32;          // rows per step

struct RowsOuterArgs {
  const float* g;
  long long ldg;
  const float* x;
  long long ldx;
  float* slabs;      // [gridDim.x][OUT * IN + OUT]
  long long rows;
  int chunks;
};

typedef __attribute__((address_space(3))) s16x4 ro_lds_s16x4;
typedef __bf16 ro_bf16x8 __attribute__((ext_vector_type(8)));
typedef short ro_s16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ ro_bf16x8 ro_tr_pair(const u16* p, int second_off) {
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ro_lds_s16x4*)p);
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ro_lds_s16x4*)(p + second_off));
  ro_s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(ro_bf16x8, v);
}

// GB / XB: the operand is stored as bf16 (the level MLP's hidden activations / hidden gradients, mlp2_bf16.hip hid16): 8-byte
// loads go to LDS as they are
template <int OUT, int IN, bool GB, bool XB>
__global__ void __launch_bounds__(256) rows_outer_kernel(RowsOuterArgs a) {
  constexpr int PG = OUT + 16, PX = IN + 16;                      // LDS row pitches (elements): 8 * odd dwords -> conflict-free tr reads
  constexpr int MB = OUT / 16, NB = IN / 16, MW = MB / 2, NW = NB / 2;
  constexpr int GI = RO_KT * OUT / 4, XI = RO_KT * IN / 4;        // float4 items per chunk
  constexpr int NG = GI / 256, NX = XI / 256;
  static_assert(GI % 256 == 0 && XI % 256 == 0 && (256 % (OUT / 4)) == 0, "staging granularity");
  __shared__ __attribute__((aligned(16))) u16 gs[RO_KT * PG];
  __shared__ __attribute__((aligned(16))) u16 xs[RO_KT * PX];
  __shared__ float csum[256 / (OUT / 4)][OUT];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int tr_row = 4 * (lane >> 4) + ((lane >> 2) & 3), tr_col = 4 * (lane & 3);
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  f32x4 acc[MW][NW];
#pragma unroll
  for (int i = 0; i < MW; ++i)
#pragma unroll
    for (int j = 0; j < NW; ++j) acc[i][j] = zero;
  f32x4 cs = zero;                                               // column sums of g: this thread always stages the same 4 columns
  f32x4 gr[NG], xr[NX];                  // fp32 operands: four floats per item; bf16 operands: the 8 bytes in .x / .y
  auto load4 = [&](const float* base, long long row, long long ld, int col, bool b16) -> f32x4 {
    if (b16) {
      const u32x2 v = *reinterpret_cast<const u32x2*>(reinterpret_cast<const u16*>(base) + row * ld + col);
      return f32x4{__uint_as_float(v.x), __uint_as_float(v.y), 0.f, 0.f};
    }
    return *reinterpret_cast<const f32x4*>(base + row * ld + col);
  };
  auto request = [&](int chunk) {
    const long long r0 = (long long)chunk * RO_KT;
#pragma unroll
    for (int k = 0; k < NG; ++k) {
      const int it = tid + k * 256, col = (it % (OUT / 4)) * 4, row = it / (OUT / 4);
      const bool ok = r0 + row < a.rows;
      gr[k] = ok ? load4(a.g, r0 + row, a.ldg, col, GB) : zero;
    }
#pragma unroll
    for (int k = 0; k < NX; ++k) {
      const int it = tid + k * 256, col = (it % (IN / 4)) * 4, row = it / (IN / 4);
      const bool ok = r0 + row < a.rows;
      xr[k] = ok ? load4(a.x, r0 + row, a.ldx, col, XB) : zero;
    }
  };
  auto deposit = [&]() {
#pragma unroll
    for (int k = 0; k < NG; ++k) {
      const int it = tid + k * 256, col = (it % (OUT / 4)) * 4, row = it / (OUT / 4);
      if (GB) {
        const u32x2 v = {__float_as_uint(gr[k].x), __float_as_uint(gr[k].y)};
        *reinterpret_cast<u32x2*>(gs + row * PG + col) = v;
        float f[4];
        unpack4(v, f);
        cs += f32x4{f[0], f[1], f[2], f[3]};
      } else {
        *reinterpret_cast<s16x4*>(gs + row * PG + col) = pack_bf16x4(gr[k]);
        cs += gr[k];
      }
    }
#pragma unroll
    for (int k = 0; k < NX; ++k) {
      const int it = tid + k * 256, col = (it % (IN / 4)) * 4, row = it / (IN / 4);
      if (XB) *reinterpret_cast<u32x2*>(xs + row * PX + col) = u32x2{__float_as_uint(xr[k].x), __float_as_uint(xr[k].y)};
      else *reinterpret_cast<s16x4*>(xs + row * PX + col) = pack_bf16x4(xr[k]);
    }
  };
  if ((int)blockIdx.x < a.chunks) request(blockIdx.x);
  for (int chunk = blockIdx.x; chunk < a.chunks; chunk += gridDim.x) {
    deposit();
    __syncthreads();
    if (chunk + (int)gridDim.x < a.chunks) request(chunk + gridDim.x);
    ro_bf16x8 af[MW];
#pragma unroll
    for (int i = 0; i < MW; ++i) af[i] = ro_tr_pair(gs + tr_row * PG + (wm * MW + i) * 16 + tr_col, 16 * PG);
#pragma unroll
    for (int j = 0; j < NW; ++j) {
      const ro_bf16x8 bfr = ro_tr_pair(xs + tr_row * PX + (wn * NW + j) * 16 + tr_col, 16 * PX);
#pragma unroll
      for (int i = 0; i < MW; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr, acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }
  // acc[i][j][t] at lane (r, q) = dW[o = 16 (wm MW + i) + 4 q + t][in = 16 (wn NW + j) + r]
  const int r = lane & 15, q = lane >> 4;
  float* slab = a.slabs + (long long)blockIdx.x * (OUT * IN + OUT);
#pragma unroll
  for (int i = 0; i < MW; ++i)
#pragma unroll
    for (int j = 0; j < NW; ++j)
#pragma unroll
      for (int t = 0; t < 4; ++t) slab[(long long)((wm * MW + i) * 16 + 4 * q + t) * IN + (wn * NW + j) * 16 + r] = acc[i][j][t];
  // column sums: threads tid, tid + OUT / 4, ... staged the same columns; added in thread order
  constexpr int TPR = OUT / 4;
  *reinterpret_cast<f32x4*>(&csum[tid / TPR][(tid % TPR) * 4]) = cs;
  __syncthreads();
  if (tid < OUT) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 256 / TPR; ++k) s += csum[k][tid];
    slab[OUT * IN + tid] = s;
  }
}

static inline int rows_outer_grid(int chunks) {
  int g = chunks / 4;                                 // at least four 32-row chunks per slab
  if (g > 256) g = 256;
  return g < 1 ? 1 : g;
}

}  // namespace mmft

using namespace mmft;

extern "C" {

int mmft_rows_outer_supported(int out, int in) { return ((out == 128 && in == 256) || (out == 256 && in == 128)) ? 1 : 0; }

long long mmft_rows_outer_workspace_bytes(long long rows, int out, int in) {
  const int chunks = (int)((rows + RO_KT - 1) / RO_KT);
  return (long long)rows_outer_grid(chunks) * ((long long)out * in + out) * 4;
}

/* dw [out][in] (+)= g^T x over `rows` rows, db [out] (+)= column sums of g (db may be NULL); g [rows][ldg >= out],
 * x [rows][ldx >= in] fp32 (rounded to bf16 at staging) or - g_bf16 / x_bf16 - stored as bf16 (ld in elements), fp32 accumulation (bf16 math mode's weight gradient of
 * fc_cell_neigh, src/model.py:48-51).  (out, in) in {(128, 256), (256, 128)}. */
int mmft_rows_outer_bf16(const float* g, long long ldg, const float* x, long long ldx, float* dw, float* db, long long rows, int out,
                         int in, int accumulate, float* workspace, long long workspace_bytes, int g_bf16, int x_bf16, int device,
                         void* stream) {
  MMFT_REQUIRE(g && x && dw && rows > 0, "rows_outer_bf16: bad arguments");
  MMFT_REQUIRE(mmft_rows_outer_supported(out, in), "rows_outer_bf16: (out, in) must be (128, 256) or (256, 128)");
  MMFT_REQUIRE(ldg >= out && ldx >= in && ldg % 4 == 0 && ldx % 4 == 0 && aligned16(g) && aligned16(x) && aligned16(dw),
               "rows_outer_bf16: rows must be 16-byte aligned");
  MMFT_REQUIRE(workspace && workspace_bytes >= mmft_rows_outer_workspace_bytes(rows, out, in) && aligned16(workspace),
               "rows_outer_bf16: workspace too small");
  DeviceGuard dg(device);
  hipStream_t st = (hipStream_t)stream;
  const int chunks = (int)((rows + RO_KT - 1) / RO_KT), grid = rows_outer_grid(chunks);
  RowsOuterArgs a{g, ldg, x, ldx, workspace, rows, chunks};
  const double fl = 2.0 * rows * out * in, by = (g_bf16 ? 2.0 : 4.0) * rows * out + (x_bf16 ? 2.0 : 4.0) * rows * in;
#define RO_LAUNCH(O, I, GBV, XBV) \
  MMFT_LAUNCH("rows_outer_kernel<" #O "," #I ">", fl, by, (rows_outer_kernel<O, I, GBV, XBV>), dim3(grid), dim3(256), st, a)
  if (out == 128) {
    if (g_bf16 && x_bf16) RO_LAUNCH(128, 256, true, true);
    else if (g_bf16) RO_LAUNCH(128, 256, true, false);
    else if (x_bf16) RO_LAUNCH(128, 256, false, true);
    else RO_LAUNCH(128, 256, false, false);
  } else {
    if (g_bf16 && x_bf16) RO_LAUNCH(256, 128, true, true);
    else if (g_bf16) RO_LAUNCH(256, 128, true, false);
    else if (x_bf16) RO_LAUNCH(256, 128, false, true);
    else RO_LAUNCH(256, 128, false, false);
  }
#undef RO_LAUNCH
  int rc = check_launch("rows_outer_bf16");
  if (rc) return rc;
  const long long wel = (long long)out * in;
  // the slabs interleave [weights | column sums]: two strided reductions
  rc = launch_slab_reduce_strided(workspace, grid, wel + out, wel, dw, accumulate, st);
  if (rc || !db) return rc;
  return launch_slab_reduce_strided(workspace + wel, grid, wel + out, out, db, accumulate, st);
}

}  // extern "C"

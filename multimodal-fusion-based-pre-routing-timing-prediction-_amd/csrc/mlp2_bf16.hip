// Lean bf16 form of the fused Linear-ReLU-Linear level kernel (MMFT_MATH_BF16), 128 -> 256 -> 128:
//
//   hid = mask ? (x1[rows] . W1^T) * (mask[rows] > 0) : relu(x1[rows] . W1^T + b1)
//   out[rows] = add_act ? act(out[rows] + hid . W2^T + b2) : hid . W2^T + b2
//
// The fp32 kernel (mlp2.hip) is bound by its 2 x 128 KB fp32 weight panels: every workgroup pulls them from L2, parks
// them in registers and stages them through LDS tile by tile (one or two barriers per 32-deep K tile), and its MFMA phase
// alone is ~7 us.  On the bf16 pipe the arithmetic is 0.25 us, so this kernel is organised around what is left:
//   * the weights arrive PRE-PACKED as bf16 in [n][k] order (mmft_pack_bf16, once per sweep: 4 x 64 KB), so a lane's
//     MFMA A fragment - 8 consecutive k of one output feature - is ONE 16-byte global load; all 16 fragment loads of a
//     wave (both layers) are issued at kernel entry and never touch LDS;
//   * only the row operands go through LDS as bf16: the gathered x tile [32][128] and the hidden tile [32][256]
//     (conflict-light ds_read_b128, 26 KB in all), two barriers in the whole kernel;
//   * 8 waves per 32-row tile: wave w owns hidden columns [32w, 32w + 32) and output columns [16w, 16w + 16).
// The reverse form (weights_kmajor in the fp32 kernel) is the same kernel fed with the transposed packs.
#include <stdlib.h>
#include "gemm_bf16.h"
#include "fold_gather.h"

namespace mmft {

constexpr int L2_BM = 32, L2_K1 = 128, L2_HD = 256, L2_D2 = 128;
constexpr int L2_XS = L2_K1 + 8, L2_HS = L2_HD + 8;        // LDS row strides in bf16 elements (multiples of 8)

// Four consecutive columns of a row of the hidden tensors (fc_cell_neigh's hidden activations HN, their gradients DHN): fp32, or
// - hid16 - bf16 (round to nearest even; `ld` counts elements of the stored type).  Every consumer rounds these values to
// bf16 anyway (MFMA operands of the weight gradients) or only looks at their sign (the ReLU mask), so the bf16 form changes
// no result and halves 2 KB of traffic per row and direction.
__device__ __forceinline__ void hid_store4(float* base, long long off, f32x4 v, int hid16) {
  if (hid16) {
    const unsigned lo = pack_bf16(v.x, v.y), hi = pack_bf16(v.z, v.w);
    *reinterpret_cast<unsigned long long*>(reinterpret_cast<unsigned short*>(base) + off) = ((unsigned long long)hi << 32) | lo;
  } else {
    *reinterpret_cast<f32x4*>(base + off) = v;
  }
}
__device__ __forceinline__ f32x4 hid_load4(const float* base, long long off, int hid16) {
  if (!hid16) return *reinterpret_cast<const f32x4*>(base + off);
  const unsigned long long u = *reinterpret_cast<const unsigned long long*>(reinterpret_cast<const unsigned short*>(base) + off);
  const unsigned lo = (unsigned)u, hi = (unsigned)(u >> 32);
  return f32x4{__uint_as_float(lo << 16), __uint_as_float(lo & 0xffff0000u), __uint_as_float(hi << 16), __uint_as_float(hi & 0xffff0000u)};
}

struct Mlp2Bf16Args {
  const float* x1;
  long long ldx1;
  const int* rows;
  int n;
  const unsigned short* w1;   // bf16 [HD][K1]
  const float* b1;
  const unsigned short* w2;   // bf16 [D2][HD]
  const float* b2;
  const float* mask;
  long long ldmask;
  float* hid_out;
  long long ldhid;
  int hid16;                   // mask / hid_out hold bf16
  float* out;
  long long ldout;
  int add_act, relu_out;
  const unsigned char* active;
};

__global__ void __launch_bounds__(512) mlp2_rows_bf16_kernel(Mlp2Bf16Args a) {
  __shared__ __attribute__((aligned(16))) unsigned short xs[L2_BM * L2_XS];
  __shared__ __attribute__((aligned(16))) unsigned short hs[L2_BM * L2_HS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = blockIdx.x * L2_BM;
  auto live_row = [&](int r) -> bool { return m0 + r < a.n && (!a.active || a.active[a.rows[m0 + r]]); };
  if (a.active) {
    int any = 0;
    if (tid < L2_BM) any = live_row(tid) ? 1 : 0;
    if (!__syncthreads_or(any)) return;
  }
  const int r16 = lane & 15, q = lane >> 4;

  // ---- every global load of the kernel: the wave's weight fragments of both layers, then the gathered x rows
  bf16x8 w1f[2][4], w2f[8];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
      w1f[j][ks] = *reinterpret_cast<const bf16x8*>(a.w1 + (long long)(wave * 32 + j * 16 + r16) * L2_K1 + ks * 32 + q * 8);
#pragma unroll
  for (int ks = 0; ks < 8; ++ks)
    w2f[ks] = *reinterpret_cast<const bf16x8*>(a.w2 + (long long)(wave * 16 + r16) * L2_HD + ks * 32 + q * 8);
  {
    // 32 rows x 128 floats = 1024 groups of 4 floats, 2 per thread: thread -> (row, 8 consecutive k)
    const int r = tid >> 4, k8 = (tid & 15) * 8;
    f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0;
    if (live_row(r)) {
      const float* p = a.x1 + (long long)a.rows[m0 + r] * a.ldx1 + k8;
      v0 = *reinterpret_cast<const f32x4*>(p);
      v1 = *reinterpret_cast<const f32x4*>(p + 4);
    }
    u32x4 pk = {pack_bf16(v0.x, v0.y), pack_bf16(v0.z, v0.w), pack_bf16(v1.x, v1.y), pack_bf16(v1.z, v1.w)};
    *reinterpret_cast<u32x4*>(xs + r * L2_XS + k8) = pk;
  }
  __syncthreads();

  // ---- phase 1: hidden columns [32 wave, 32 wave + 32) of the 32 rows
  f32x4 acc1[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc1[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    bf16x8 xf[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) xf[i] = *reinterpret_cast<const bf16x8*>(xs + (i * 16 + r16) * L2_XS + ks * 32 + q * 8);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc1[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1f[j][ks], xf[i], acc1[i][j], 0, 0, 0);
  }
  // lane holds hidden columns nn .. nn + 3 of row m: bias + ReLU, or the ReLU mask of the saved forward activations
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int m = i * 16 + r16, nn = wave * 32 + j * 16 + q * 4;
      f32x4 v = acc1[i][j];
      const bool live = live_row(m);
      const long long row = live ? (long long)a.rows[m0 + m] : 0;
      if (a.mask) {
        f32x4 mk = {0.f, 0.f, 0.f, 0.f};
        if (live) mk = hid_load4(a.mask, row * a.ldmask + nn, a.hid16);
        v.x = mk.x > 0.f ? v.x : 0.f; v.y = mk.y > 0.f ? v.y : 0.f;
        v.z = mk.z > 0.f ? v.z : 0.f; v.w = mk.w > 0.f ? v.w : 0.f;
      } else {
        if (a.b1) v += *reinterpret_cast<const f32x4*>(a.b1 + nn);
        v.x = v.x > 0.f ? v.x : 0.f; v.y = v.y > 0.f ? v.y : 0.f;
        v.z = v.z > 0.f ? v.z : 0.f; v.w = v.w > 0.f ? v.w : 0.f;
      }
      unsigned lo = pack_bf16(v.x, v.y), hi = pack_bf16(v.z, v.w);
      *reinterpret_cast<unsigned long long*>(hs + m * L2_HS + nn) = ((unsigned long long)hi << 32) | lo;
      if (a.hid_out && live) hid_store4(a.hid_out, row * a.ldhid + nn, v, a.hid16);
    }
  __syncthreads();

  // ---- phase 2: output columns [16 wave, 16 wave + 16)
  f32x4 acc2[2];
  acc2[0] = acc2[1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      bf16x8 hf = *reinterpret_cast<const bf16x8*>(hs + (i * 16 + r16) * L2_HS + ks * 32 + q * 8);
      acc2[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2f[ks], hf, acc2[i], 0, 0, 0);
    }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = i * 16 + r16;
    if (!live_row(m)) continue;
    const int nn = wave * 16 + q * 4;
    float* o = a.out + (long long)a.rows[m0 + m] * a.ldout + nn;
    f32x4 v = acc2[i];
    if (a.b2) v += *reinterpret_cast<const f32x4*>(a.b2 + nn);
    if (a.add_act) v += *reinterpret_cast<const f32x4*>(o);
    if (a.relu_out) {
      v.x = v.x > 0.f ? v.x : 0.f; v.y = v.y > 0.f ? v.y : 0.f;
      v.z = v.z > 0.f ? v.z : 0.f; v.w = v.w > 0.f ? v.w : 0.f;
    }
    *reinterpret_cast<f32x4*>(o) = v;
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Fused forward LEVEL kernel (bf16 mode): the folded gather of one (net level l - 1, cell level l) pair AND the cell
// level's Linear-ReLU-Linear in one launch.
//   blocks [0, cell_tiles):  32 cell rows each - the 512 threads first compute the softmax-weighted fan-in sums A[v]
//       of their rows (two (row, 4-channel) items per thread, in-neighbours of level l - 1 recomputed from PRE and
//       their driver's row exactly as pair_fwd_gather does), store A / LSE for the reverse sweep and put the bf16 tile
//       straight into LDS; then the two MFMA phases of mlp2_rows_bf16_kernel: h[v] = act(h[v] + fc_cell_neigh(A[v]));
//   blocks [cell_tiles, ...): the net rows of level l - 1: h[u] = act(PRE[u] + mean h[driver]).
// On the bf16 pipe the MLP of a tile is ~0.3 us of arithmetic, so the chain pays one launch per level pair instead of two.
// Rows with very many in-edges are walked serially by one thread group here: the caller uses the two-kernel form
// (pair_fwd_gather with its whole-workgroup path + mlp2_rows_bf16) for levels that have such rows.
// ------------------------------------------------------------------------------------------------------------------
struct LevelFwdArgs {
  float* h;
  const float* pre;
  long long ld;
  const int *in_ptr, *in_idx, *ic_ptr, *ic_idx, *ic_drv;
  int net_row0, n_net;
  const int* rows;
  int cell_row0, n_cell;
  float *A, *LSE;
  const unsigned short *w1, *w2;
  const float *b1, *b2;
  float* hid_out;
  long long ldhid;
  int hid16;                   // mask / hid_out hold bf16
  int relu;
  const unsigned char* active;
  int cell_tiles;
};

// BM rows per tile (16: one gather item per thread, twice as many workgroups - the gather, not the MLP, sets the
// duration of a level, and it wants the memory-level parallelism; 32: the tile of mlp2_rows_bf16_kernel)
template <int BM>
__global__ void __launch_bounds__(512) level_fwd_bf16_kernel(LevelFwdArgs a) {
  constexpr int RT = BM / 16;
  __shared__ __attribute__((aligned(16))) unsigned short xs[BM * L2_XS];
  __shared__ __attribute__((aligned(16))) unsigned short hs[BM * L2_HS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const FoldSrc fs{a.h, a.pre, a.ld, a.in_ptr, a.in_idx, a.ic_idx, a.ic_drv, a.net_row0, a.n_net, a.relu};
  if ((int)blockIdx.x >= a.cell_tiles) {
    // ---- net rows of level l - 1 (32 float4 groups per row at D = 128)
    const long long total = (long long)a.n_net * 32;
    for (long long t = (long long)(blockIdx.x - a.cell_tiles) * 512 + tid; t < total;
         t += (long long)(gridDim.x - a.cell_tiles) * 512) {
      const int u = a.net_row0 + (int)(t >> 5), c = ((int)t & 31) * 4;
      if (a.active && !a.active[u]) continue;
      *reinterpret_cast<f32x4*>(a.h + (long long)u * a.ld + c) = fold_net_value(fs, u, c);
    }
    return;
  }
  const int m0 = blockIdx.x * BM;
  auto row_of = [&](int r) -> int { return a.rows ? a.rows[m0 + r] : a.cell_row0 + m0 + r; };
  auto live_row = [&](int r) -> bool { return m0 + r < a.n_cell && (!a.active || a.active[row_of(r)]); };
  if (a.active) {
    int any = 0;
    if (tid < BM) any = live_row(tid) ? 1 : 0;
    if (!__syncthreads_or(any)) return;
  }
  const int r16 = lane & 15, q = lane >> 4;
  bf16x8 w1f[2][4], w2f[8];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
      w1f[j][ks] = *reinterpret_cast<const bf16x8*>(a.w1 + (long long)(wave * 32 + j * 16 + r16) * L2_K1 + ks * 32 + q * 8);

  // ---- gather: item = (row r, channel group cg), two items per thread
#pragma unroll
  for (int it = 0; it < BM / 16; ++it) {
    const int item = tid + it * 512, r = item >> 5, c = (item & 31) * 4;
    f32x4 av = {0.f, 0.f, 0.f, 0.f};
    if (live_row(r)) {
      const int v = row_of(r);
      const int e0 = a.ic_ptr[v], e1 = a.ic_ptr[v + 1];
      SoftAccT<true> sa;
      sa.init();
      fold_gather_edges(fs, e0, e1, 1, c, sa);
      f32x4 lv = {0.f, 0.f, 0.f, 0.f};
      if (e1 > e0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          av[j] = sa.acc[j] / sa.s[j];
          lv[j] = sa.mx[j] + fg_log<true>(sa.s[j]);
        }
      }
      *reinterpret_cast<f32x4*>(a.A + (long long)v * a.ld + c) = av;
      *reinterpret_cast<f32x4*>(a.LSE + (long long)v * a.ld + c) = lv;
    }
    const unsigned lo = pack_bf16(av.x, av.y), hi = pack_bf16(av.z, av.w);
    *reinterpret_cast<unsigned long long*>(xs + r * L2_XS + c) = ((unsigned long long)hi << 32) | lo;
  }
  // the second layer's fragments are requested only now: held across the gather they cost 32 registers and a wave of occupancy
#pragma unroll
  for (int ks = 0; ks < 8; ++ks)
    w2f[ks] = *reinterpret_cast<const bf16x8*>(a.w2 + (long long)(wave * 16 + r16) * L2_HD + ks * 32 + q * 8);
  __syncthreads();

  // ---- phase 1 / epilogue 1 / phase 2 / epilogue 2: as mlp2_rows_bf16_kernel (forward form)
  f32x4 acc1[RT][2];
#pragma unroll
  for (int i = 0; i < RT; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc1[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    bf16x8 xf[RT];
#pragma unroll
    for (int i = 0; i < RT; ++i) xf[i] = *reinterpret_cast<const bf16x8*>(xs + (i * 16 + r16) * L2_XS + ks * 32 + q * 8);
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc1[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1f[j][ks], xf[i], acc1[i][j], 0, 0, 0);
  }
#pragma unroll
  for (int i = 0; i < RT; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int m = i * 16 + r16, nn = wave * 32 + j * 16 + q * 4;
      f32x4 v = acc1[i][j];
      if (a.b1) v += *reinterpret_cast<const f32x4*>(a.b1 + nn);
      v.x = v.x > 0.f ? v.x : 0.f; v.y = v.y > 0.f ? v.y : 0.f;
      v.z = v.z > 0.f ? v.z : 0.f; v.w = v.w > 0.f ? v.w : 0.f;
      unsigned lo = pack_bf16(v.x, v.y), hi = pack_bf16(v.z, v.w);
      *reinterpret_cast<unsigned long long*>(hs + m * L2_HS + nn) = ((unsigned long long)hi << 32) | lo;
      if (a.hid_out && live_row(m)) hid_store4(a.hid_out, (long long)row_of(m) * a.ldhid + nn, v, a.hid16);
    }
  __syncthreads();
  f32x4 acc2[RT];
#pragma unroll
  for (int i = 0; i < RT; ++i) acc2[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) {
#pragma unroll
    for (int i = 0; i < RT; ++i) {
      bf16x8 hf = *reinterpret_cast<const bf16x8*>(hs + (i * 16 + r16) * L2_HS + ks * 32 + q * 8);
      acc2[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2f[ks], hf, acc2[i], 0, 0, 0);
    }
  }
#pragma unroll
  for (int i = 0; i < RT; ++i) {
    const int m = i * 16 + r16;
    if (!live_row(m)) continue;
    const int nn = wave * 16 + q * 4;
    float* o = a.h + (long long)row_of(m) * a.ld + nn;
    f32x4 v = acc2[i];
    if (a.b2) v += *reinterpret_cast<const f32x4*>(a.b2 + nn);
    v += *reinterpret_cast<const f32x4*>(o);
    if (a.relu) {
      v.x = v.x > 0.f ? v.x : 0.f; v.y = v.y > 0.f ? v.y : 0.f;
      v.z = v.z > 0.f ? v.z : 0.f; v.w = v.w > 0.f ? v.w : 0.f;
    }
    *reinterpret_cast<f32x4*>(o) = v;
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Fused forward level kernel, SLOT-TABLE form.  Same arithmetic as level_fwd_bf16_kernel<16> - every row
// bitwise equal - with the serial latencies of a tile's dependent chain cut:
//   * `slots` (static, built on the host next to the CSR): per cell row of fan-in <= 4 the four (row of h to read, row of
//     PRE to add or -1) pairs of its in-edges in edge order - ONE 32-byte load whose address is known at kernel entry
//     replaces ic_ptr -> (ic_idx, ic_drv); an edge from an older net level reads PRE of a fixed hot row instead of a
//     512-byte row it then ignores;
//   * the read-modify-write operand h[v] of the second epilogue is requested at kernel entry, next to the first layer's
//     weight fragments, instead of behind the second MFMA phase;
//   * the net rows of level l - 1 are no longer separate workgroups (504 cell tiles + 504 net blocks on 512 workgroup
//     slots = two rounds, the second one waiting for the first): workgroup b also writes net rows 16 b .. 16 b + 15,
//     their loads issued ahead of the gather through `net_drv` (the single driver of every net, static).
struct LevelSlotsArgs {
  float* h;
  const float* pre;
  long long ld;
  const int* slots;      // [N][8]: hrow[4], prow[4]; hrow < 0: no edge; prow < 0: the edge's value is h[hrow] itself
  const int* net_drv;    // [N]: driver row of a net, < 0: none
  int net_row0, n_net, cell_row0, n_cell;
  float *A, *LSE;
  const unsigned short *w1, *w2;
  const float *b1, *b2;
  float* hid_out;
  long long ldhid;
  int hid16;                   // mask / hid_out hold bf16
  int relu;
  const unsigned char* active;
  int cell_tiles;
};

template <int RB>
__global__ void __launch_bounds__(512) level_fwd_slots_kernel(LevelSlotsArgs a) {
  // RB row blocks of 16 cell rows per workgroup: thread group g gathers rows g, g + 16, ..; with RB = 2 half as many workgroups
  // fetch the 128 KB of packed weights (64 MB of L2 traffic per launch at RB = 1)
  constexpr int BM = 16 * RB;
  __shared__ __attribute__((aligned(16))) unsigned short xs[BM * L2_XS];
  __shared__ __attribute__((aligned(16))) unsigned short hs[BM * L2_HS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, q = lane >> 4;
  const int b = blockIdx.x, gr = tid >> 5, gc = (tid & 31) * 4;              // gather / net item of this thread: (row, channel group)
  const bool has_cell = b < a.cell_tiles;
  const int m0 = b * BM;
  // ---- requests whose addresses are known now
  int nd[RB];
  f32x4 npre[RB];
  bool net_ok[RB], live[RB];
  int hrow[RB][4], prow[RB][4];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    const int nrow = b * BM + gr + 16 * rb, u = a.net_row0 + nrow;
    net_ok[rb] = nrow < a.n_net && (!a.active || a.active[u]);
    nd[rb] = -1;
    npre[rb] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (net_ok[rb]) {
      nd[rb] = a.net_drv[u];
      npre[rb] = *reinterpret_cast<const f32x4*>(a.pre + (long long)u * a.ld + gc);
    }
    const int v = a.cell_row0 + m0 + gr + 16 * rb;
    live[rb] = has_cell && m0 + gr + 16 * rb < a.n_cell && (!a.active || a.active[v]);
#pragma unroll
    for (int k = 0; k < 4; ++k) hrow[rb][k] = prow[rb][k] = -1;
    if (live[rb]) {
      const int4 s0 = *reinterpret_cast<const int4*>(a.slots + (long long)v * 8);
      const int4 s1 = *reinterpret_cast<const int4*>(a.slots + (long long)v * 8 + 4);
      hrow[rb][0] = s0.x; hrow[rb][1] = s0.y; hrow[rb][2] = s0.z; hrow[rb][3] = s0.w;
      prow[rb][0] = s1.x; prow[rb][1] = s1.y; prow[rb][2] = s1.z; prow[rb][3] = s1.w;
    }
  }
  bf16x8 w1f[2][4], w2f[8];
  // epilogue-2 operand of this lane: rows m0 + r16 (+ 16), output features wave * 16 + 4 q ..
  bool elive[RB];
  f32x4 hold[RB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    const int ev = a.cell_row0 + m0 + r16 + 16 * rb;
    elive[rb] = has_cell && m0 + r16 + 16 * rb < a.n_cell && (!a.active || a.active[ev]);
    hold[rb] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  if (has_cell) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
        w1f[j][ks] = *reinterpret_cast<const bf16x8*>(a.w1 + (long long)(wave * 32 + j * 16 + r16) * L2_K1 + ks * 32 + q * 8);
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
      if (elive[rb]) hold[rb] = *reinterpret_cast<const f32x4*>(a.h + (long long)(a.cell_row0 + m0 + r16 + 16 * rb) * a.ld + wave * 16 + q * 4);
  }
  // ---- net rows of level l - 1: h[u] = act(PRE[u] + h[driver])   (mean over ONE in-edge = the driver's row itself)
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
    if (net_ok[rb]) {
      const int u = a.net_row0 + b * BM + gr + 16 * rb;
      f32x4 hd = {0.f, 0.f, 0.f, 0.f};
      if (nd[rb] >= 0) hd = *reinterpret_cast<const f32x4*>(a.h + (long long)nd[rb] * a.ld + gc);
      *reinterpret_cast<f32x4*>(a.h + (long long)u * a.ld + gc) = fg_finish_net(hd, npre[rb], a.relu);
    }
  if (!has_cell) return;
  if (a.active) {
    int any = 0;
    if (tid < BM) any = (m0 + tid < a.n_cell && a.active[a.cell_row0 + m0 + tid]) ? 1 : 0;
    if (!__syncthreads_or(any)) return;
  }
  // ---- gather of the thread's (row, channel group)s: the four slots requested together, consumed in edge order
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    const int v = a.cell_row0 + m0 + gr + 16 * rb;
    f32x4 av = {0.f, 0.f, 0.f, 0.f};
    if (live[rb]) {
      const int h0 = hrow[rb][0] >= 0 ? hrow[rb][0] : v;             // an empty row still issues (and drops) loads of valid rows
      f32x4 xa[4], xp[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        xa[k] = *reinterpret_cast<const f32x4*>(a.h + (long long)(hrow[rb][k] >= 0 ? hrow[rb][k] : h0) * a.ld + gc);
        xp[k] = *reinterpret_cast<const f32x4*>(a.pre + (long long)(prow[rb][k] >= 0 ? prow[rb][k] : a.net_row0) * a.ld + gc);
      }
      SoftAccT<true> sa;
      sa.init();
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const f32x4 x = prow[rb][k] >= 0 ? fg_finish_net(xa[k], xp[k], a.relu) : xa[k];
        SoftAccT<true> nx = sa;
        nx.add(x);
        const bool ok = hrow[rb][k] >= 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          sa.mx[j] = ok ? nx.mx[j] : sa.mx[j];
          sa.s[j] = ok ? nx.s[j] : sa.s[j];
          sa.acc[j] = ok ? nx.acc[j] : sa.acc[j];
        }
      }
      f32x4 lv = {0.f, 0.f, 0.f, 0.f};
      if (hrow[rb][0] >= 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          av[j] = sa.acc[j] / sa.s[j];
          lv[j] = sa.mx[j] + fg_log<true>(sa.s[j]);
        }
      }
      *reinterpret_cast<f32x4*>(a.A + (long long)v * a.ld + gc) = av;
      *reinterpret_cast<f32x4*>(a.LSE + (long long)v * a.ld + gc) = lv;
    }
    const unsigned lo = pack_bf16(av.x, av.y), hi = pack_bf16(av.z, av.w);
    *reinterpret_cast<unsigned long long*>(xs + (gr + 16 * rb) * L2_XS + gc) = ((unsigned long long)hi << 32) | lo;
  }
#pragma unroll
  for (int ks = 0; ks < 8; ++ks)
    w2f[ks] = *reinterpret_cast<const bf16x8*>(a.w2 + (long long)(wave * 16 + r16) * L2_HD + ks * 32 + q * 8);
  __syncthreads();
  // ---- phase 1 / epilogue 1 / phase 2 / epilogue 2: as level_fwd_bf16_kernel<16>, per row block
  f32x4 acc1[RB][2];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc1[rb][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      const bf16x8 xf = *reinterpret_cast<const bf16x8*>(xs + (r16 + 16 * rb) * L2_XS + ks * 32 + q * 8);
#pragma unroll
      for (int j = 0; j < 2; ++j) acc1[rb][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1f[j][ks], xf, acc1[rb][j], 0, 0, 0);
    }
  }
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int nn = wave * 32 + j * 16 + q * 4;
      f32x4 hv = acc1[rb][j];
      if (a.b1) hv += *reinterpret_cast<const f32x4*>(a.b1 + nn);
      hv.x = hv.x > 0.f ? hv.x : 0.f; hv.y = hv.y > 0.f ? hv.y : 0.f;
      hv.z = hv.z > 0.f ? hv.z : 0.f; hv.w = hv.w > 0.f ? hv.w : 0.f;
      const unsigned lo = pack_bf16(hv.x, hv.y), hi = pack_bf16(hv.z, hv.w);
      *reinterpret_cast<unsigned long long*>(hs + (r16 + 16 * rb) * L2_HS + nn) = ((unsigned long long)hi << 32) | lo;
      if (a.hid_out && elive[rb])
        hid_store4(a.hid_out, (long long)(a.cell_row0 + m0 + r16 + 16 * rb) * a.ldhid + nn, hv, a.hid16);
    }
  __syncthreads();
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    f32x4 acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const bf16x8 hf = *reinterpret_cast<const bf16x8*>(hs + (r16 + 16 * rb) * L2_HS + ks * 32 + q * 8);
      acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2f[ks], hf, acc2, 0, 0, 0);
    }
    if (elive[rb]) {
      const int nn = wave * 16 + q * 4;
      f32x4 o = acc2;
      if (a.b2) o += *reinterpret_cast<const f32x4*>(a.b2 + nn);
      o += hold[rb];
      if (a.relu) {
        o.x = o.x > 0.f ? o.x : 0.f; o.y = o.y > 0.f ? o.y : 0.f;
        o.z = o.z > 0.f ? o.z : 0.f; o.w = o.w > 0.f ? o.w : 0.f;
      }
      *reinterpret_cast<f32x4*>(a.h + (long long)(a.cell_row0 + m0 + r16 + 16 * rb) * a.ld + nn) = o;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Reverse sweep of one (cell level l, net level l + 1) PAIR in ONE launch (three before: pull of the net level, pull of
// the cell level, fc_cell_neigh's backward):
//   net rows w of level l + 1 (the SINKS of the level-l drivers):
//       G[w] = relu'(h[w]) (own[w] ? G[w] : 0  +  sum over the cell consumers c of w, in edge order, of
//                           DA[c] exp(h[w] - LSE[c]) (1 + h[w] - A[c]))                         (src/model.py:113-116 reversed)
//   cell rows v of level l (the DRIVERS):
//       G[v] = relu'(h[v]) (own[v] ? G[v] : 0  +  sum over v's sinks w, in edge order, of G[w])    (src/model.py:186-187: mean over
//                                                                                                 ONE in-edge, weight 1)
//       DA[v] = ((G[v] W2g) * relu'(HN[v])) W1g,  DHN[v] kept for the batched weight gradients   (src/model.py:138-146)
// Precondition (checked on the host, PinGraph.level_bwd_pairs): the level-l rows are a contiguous id range whose out-net
// edge list, read in CSR order, IS the id range of level l + 1 - the sinks of one driver are consecutive ids and the
// drivers' sink runs follow each other.  A workgroup then owns <= 32 consecutive drivers (a static tile table) and with them
// one contiguous run of sinks: thread group g (32 lanes x 4 channels) takes the sinks g, g + 16, .. of the run (the consumers
// come from a static slot table int32[N][4], so the chain is slots -> rows; the next sink's slots / h / own row are
// requested before the current one's rows are consumed), parks the finished sink rows in LDS and the driver's thread group
// adds its own run in edge order - no driver walks its sinks' dependent loads in series (the form that made the earlier
// driver-side fold 5x slower than two pulls), no float atomics, and for such tiles the same additions in the same order as
// level_bwd_pull: bitwise equal (tested).  The finished driver rows go to LDS as bf16 and through the MFMA phases of the level
// MLP (one or two 16-row blocks).
// HEAVY drivers (more sinks than a tile's target: drivers of high-fanout nets) would leave their workgroup walking hundreds
// of sinks while the rest of the card has finished; their sink run is cut into PARTS (16 sinks), one workgroup each.  A part
// finishes its sinks, writes the partial sum of their rows (in edge order) to scratch and bumps a per-driver counter; the
// workgroup that arrives last adds the partial sums IN PART ORDER (whichever workgroup that is: the result does not depend on
// the arrival order), finishes the driver's row and runs the MLP phases for it.  Nobody waits for anybody.
struct LevelBwdPairArgs {
  float* G;
  const float* h;
  const float* A;
  const float* LSE;
  float* DA;
  long long ld;
  const unsigned char* own;
  const int* tiles;            // [gridDim.x][8]: first driver id, driver count (<= 32), part, parts (0 = not a heavy driver),
                               // first scratch row of the driver's parts, counter index, first / end out-net CSR position of
                               // the tile's sinks
  const int* on_ptr;           // out-net CSR row pointers
  int sink_shift;              // sink id = CSR position + sink_shift
  const int* cslots;           // [N][4] cell consumers of a net row in out-edge order, -1 = none; slot 3 <= -2: more than four,
                               // the fourth and later ones are read from the out-cell CSR at position -2 - slot
  const int* oc_ptr;
  const int* oc_idx;
  float* scratch;              // [rows][128] partial sums of heavy drivers' parts
  int* counters;               // one per heavy driver of the level, zero between launches
  int relu, has_mlp;
  const unsigned short* w1;    // bf16 [256][128] = W2g^T   (hidden gradient = G . W2g)
  const unsigned short* w2;    // bf16 [128][256] = W1g^T
  const float* mask;           // HN
  long long ldmask;
  float* hid_out;              // DHN (optional)
  long long ldhid;
  int hid16;                   // mask / hid_out hold bf16
};

__global__ void __launch_bounds__(512) level_bwd_pair_kernel(LevelBwdPairArgs a) {
  constexpr int BM = 32, SC = 32, SGP = L2_K1 + 4;
  __shared__ __attribute__((aligned(16))) float sg[SC * SGP];
  __shared__ __attribute__((aligned(16))) unsigned short xs[BM * L2_XS];
  __shared__ __attribute__((aligned(16))) unsigned short hs[BM * L2_HS];
  __shared__ int s_last;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, q = lane >> 4;
  const int gr = tid >> 5, gc = (tid & 31) * 4;
  const int4 t0 = *reinterpret_cast<const int4*>(a.tiles + 8 * (long long)blockIdx.x);
  const int4 t1 = *reinterpret_cast<const int4*>(a.tiles + 8 * (long long)blockIdx.x + 4);
  const int v0 = t0.x, nd = t0.y, part = t0.z, parts = t0.w, sbase = t1.x, cidx = t1.y;
  const int e0 = t1.z, e1 = t1.w;                          // the tile's sinks: known without a look at the CSR
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  // ---- this thread group's drivers gr and gr + 16: their sink runs, own-row gradients, forward values
  int ps[2] = {0, 0}, pe[2] = {0, 0};
  f32x4 acc[2] = {zero, zero}, hv[2] = {zero, zero};
#pragma unroll
  for (int d = 0; d < 2; ++d) {
    const int v = v0 + gr + 16 * d;
    if (gr + 16 * d < nd) {
      ps[d] = a.on_ptr[v];
      pe[d] = a.on_ptr[v + 1];
      hv[d] = *reinterpret_cast<const f32x4*>(a.h + (long long)v * a.ld + gc);
      const bool mine_v = (!a.own || a.own[v]) && !parts;
      acc[d] = *reinterpret_cast<const f32x4*>(a.G + (long long)v * a.ld + gc);     // dropped by a select when there is none
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[d][j] = mine_v ? acc[d][j] : 0.f;
    }
  }
  // ---- the tile's sinks, 32 at a time (a tile of whole drivers holds at most 32, a part of a heavy driver 16)
  for (int c0 = e0; c0 < e1; c0 += SC) {
    const int cend = c0 + SC < e1 ? c0 + SC : e1;
    int e = c0 + gr;
    // stage 1 of a sink: its slots, forward value, own-row gradient (loaded in any case and dropped by a select: a
    // conditional load would split the request phase into basic blocks, each waiting for its own loads)
    int4 cs = {-1, -1, -1, -1};
    f32x4 hw = zero, gw = zero;
    unsigned char mine = 0;
    if (e < cend) {
      const int w = e + a.sink_shift;
      cs = *reinterpret_cast<const int4*>(a.cslots + (long long)w * 4);
      hw = *reinterpret_cast<const f32x4*>(a.h + (long long)w * a.ld + gc);
      mine = a.own ? a.own[w] : (unsigned char)1;
      gw = *reinterpret_cast<const f32x4*>(a.G + (long long)w * a.ld + gc);
    }
#pragma unroll 1
    for (; e < cend; e += 16) {
      const int w = e + a.sink_shift;
      // stage 2: the four slots' rows requested together (a missing one repeats a row already asked for and is dropped)
      const int c[4] = {cs.x, cs.y, cs.z, cs.w};
      const int c0row = c[0] >= 0 ? c[0] : w;
      f32x4 da[4], aa[4], ll[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const long long o = (long long)(c[k] >= 0 ? c[k] : c0row) * a.ld + gc;
        da[k] = *reinterpret_cast<const f32x4*>(a.DA + o);
        aa[k] = *reinterpret_cast<const f32x4*>(a.A + o);
        ll[k] = *reinterpret_cast<const f32x4*>(a.LSE + o);
      }
      f32x4 g = gw;
      const f32x4 hcur = hw;
#pragma unroll
      for (int j = 0; j < 4; ++j) g[j] = mine ? g[j] : 0.f;
      // stage 1 of the group's next sink, in flight while this one's rows arrive and are consumed
      const int en = e + 16;
      if (en < cend) {
        const int wn = en + a.sink_shift;
        cs = *reinterpret_cast<const int4*>(a.cslots + (long long)wn * 4);
        hw = *reinterpret_cast<const f32x4*>(a.h + (long long)wn * a.ld + gc);
        mine = a.own ? a.own[wn] : (unsigned char)1;
        gw = *reinterpret_cast<const f32x4*>(a.G + (long long)wn * a.ld + gc);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        // the term is ~15 VALU instructions per channel: a slot neither of the wave's two sinks uses is skipped altogether
        if (!__builtin_amdgcn_ballot_w64(c[k] >= 0)) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) g[j] = c[k] >= 0 ? cell_consumer_term<true>(g[j], da[k][j], hcur[j], ll[k][j], aa[k][j]) : g[j];
      }
      if (c[3] <= -2) {
        // more than four consumers (a few percent of the pins): slot 3 holds -2 - (CSR position of the fourth one)
        const int x1 = a.oc_ptr[w + 1];
        for (int x = -2 - c[3]; x < x1; ++x) {
          const long long o = (long long)a.oc_idx[x] * a.ld + gc;
          const f32x4 d2 = *reinterpret_cast<const f32x4*>(a.DA + o), a2 = *reinterpret_cast<const f32x4*>(a.A + o),
                      l2 = *reinterpret_cast<const f32x4*>(a.LSE + o);
#pragma unroll
          for (int j = 0; j < 4; ++j) g[j] = cell_consumer_term<true>(g[j], d2[j], hcur[j], l2[j], a2[j]);
        }
      }
      if (a.relu) {
#pragma unroll
        for (int j = 0; j < 4; ++j) g[j] = hcur[j] > 0.f ? g[j] : 0.f;
      }
      *reinterpret_cast<f32x4*>(a.G + (long long)w * a.ld + gc) = g;
      *reinterpret_cast<f32x4*>(sg + (e - c0) * SGP + gc) = g;
    }
    __syncthreads();
#pragma unroll
    for (int d = 0; d < 2; ++d) {
      const int lo = ps[d] > c0 ? ps[d] : c0, hi = pe[d] < cend ? pe[d] : cend;
      for (int x = lo; x < hi; ++x) acc[d] += *reinterpret_cast<const f32x4*>(sg + (x - c0) * SGP + gc);
    }
    __syncthreads();
  }
  if (parts) {
    // ---- heavy driver: publish this part's partial sum; the last part to arrive adds all of them in part order
    if (gr == 0) {
      float* p = a.scratch + (long long)(sbase + part) * L2_K1 + gc;
#pragma unroll
      for (int j = 0; j < 4; ++j) __hip_atomic_store(p + j, acc[0][j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");          // the partial sum is visible before the counter moves
    }
    __syncthreads();
    if (tid == 0) {
      const int old = __hip_atomic_fetch_add(a.counters + cidx, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_last = old == parts - 1;
      if (old == parts - 1) __hip_atomic_store(a.counters + cidx, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next step
    }
    __syncthreads();
    if (!s_last) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");            // ... and the other parts' sums are read after it was seen full
    if (gr == 0) {
      acc[0] = zero;
      if (!a.own || a.own[v0]) acc[0] = *reinterpret_cast<const f32x4*>(a.G + (long long)v0 * a.ld + gc);
      for (int i = 0; i < parts; ++i) {
        const float* p = a.scratch + (long long)(sbase + i) * L2_K1 + gc;
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[0][j] += __hip_atomic_load(p + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
  // ---- the drivers' rows
#pragma unroll
  for (int d = 0; d < 2; ++d) {
    if (a.relu) {
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[d][j] = hv[d][j] > 0.f ? acc[d][j] : 0.f;
    }
    if (gr + 16 * d < nd) *reinterpret_cast<f32x4*>(a.G + (long long)(v0 + gr + 16 * d) * a.ld + gc) = acc[d];
  }
  if (!a.has_mlp) return;
#pragma unroll
  for (int d = 0; d < 2; ++d) {
    const unsigned lo = pack_bf16(acc[d].x, acc[d].y), hi = pack_bf16(acc[d].z, acc[d].w);       // rows >= nd hold zeros
    *reinterpret_cast<unsigned long long*>(xs + (gr + 16 * d) * L2_XS + gc) = ((unsigned long long)hi << 32) | lo;
  }
  const bool two = nd > 16;                                  // block-uniform: the second 16-row block holds drivers
  bf16x8 w1f[2][4], w2f[8];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
      w1f[j][ks] = *reinterpret_cast<const bf16x8*>(a.w1 + (long long)(wave * 32 + j * 16 + r16) * L2_K1 + ks * 32 + q * 8);
  // the ReLU mask of the hidden rows
  f32x4 mk[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      mk[i][j] = zero;
      if (r16 + 16 * i < nd)
        mk[i][j] = hid_load4(a.mask, (long long)(v0 + r16 + 16 * i) * a.ldmask + wave * 32 + j * 16 + q * 4, a.hid16);
    }
#pragma unroll
  for (int ks = 0; ks < 8; ++ks)
    w2f[ks] = *reinterpret_cast<const bf16x8*>(a.w2 + (long long)(wave * 16 + r16) * L2_HD + ks * 32 + q * 8);
  __syncthreads();
  // ---- hidden gradient = (G . W2g) * relu'(HN), DA = hidden gradient . W1g: the MFMA phases of mlp2_rows_bf16_kernel
  f32x4 acc1[2][2] = {{zero, zero}, {zero, zero}};
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (i == 1 && !two) continue;
      const bf16x8 xf = *reinterpret_cast<const bf16x8*>(xs + (i * 16 + r16) * L2_XS + ks * 32 + q * 8);
#pragma unroll
      for (int j = 0; j < 2; ++j) acc1[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1f[j][ks], xf, acc1[i][j], 0, 0, 0);
    }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    if (i == 1 && !two) continue;
    const int row = r16 + 16 * i;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int nn = wave * 32 + j * 16 + q * 4;
      f32x4 hvv = acc1[i][j];
      hvv.x = mk[i][j].x > 0.f ? hvv.x : 0.f; hvv.y = mk[i][j].y > 0.f ? hvv.y : 0.f;
      hvv.z = mk[i][j].z > 0.f ? hvv.z : 0.f; hvv.w = mk[i][j].w > 0.f ? hvv.w : 0.f;
      const unsigned lo = pack_bf16(hvv.x, hvv.y), hi = pack_bf16(hvv.z, hvv.w);
      *reinterpret_cast<unsigned long long*>(hs + row * L2_HS + nn) = ((unsigned long long)hi << 32) | lo;
      if (a.hid_out && row < nd) hid_store4(a.hid_out, (long long)(v0 + row) * a.ldhid + nn, hvv, a.hid16);
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    if (i == 1 && !two) continue;
    f32x4 acc2 = zero;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const bf16x8 hf = *reinterpret_cast<const bf16x8*>(hs + (i * 16 + r16) * L2_HS + ks * 32 + q * 8);
      acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2f[ks], hf, acc2, 0, 0, 0);
    }
    if (r16 + 16 * i < nd) *reinterpret_cast<f32x4*>(a.DA + (long long)(v0 + r16 + 16 * i) * a.ld + wave * 16 + q * 4) = acc2;
  }
}

// dst[r][c] (bf16) = src[r][c], or with transpose dst[c][r] = src[r][c]   (R x C fp32, row stride ld)
__global__ void __launch_bounds__(256) pack_bf16_kernel(const float* __restrict__ src, long long ld, int R, int C,
                                                        unsigned short* __restrict__ dst, int transpose) {
  const long long total = (long long)R * C;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    // i indexes the DESTINATION (coalesced stores); weights are tiny, the strided reads of the transposed form hit L2
    int dr = (int)(i / (transpose ? R : C)), dc = (int)(i - (long long)dr * (transpose ? R : C));
    float v = transpose ? src[(long long)dc * ld + dr] : src[(long long)dr * ld + dc];
    dst[i] = (unsigned short)(pack_bf16(v, 0.f) & 0xffff);
  }
}

}  // namespace mmft

using namespace mmft;

extern "C" int mmft_pack_bf16(const float* src, long long ld, int R, int C, void* dst, int transpose, int device,
                              void* stream) {
  MMFT_REQUIRE(src && dst && R > 0 && C > 0 && ld >= C, "pack_bf16: bad args");
  DeviceGuard dg(device);
  hipLaunchKernelGGL(pack_bf16_kernel, dim3(ew_grid((long long)R * C)), dim3(256), 0, (hipStream_t)stream, src, ld, R, C,
                     (unsigned short*)dst, transpose);
  return check_launch("pack_bf16");
}

extern "C" int mmft_mlp2_rows_bf16(const float* x1, long long ldx1, const int* rows, int n, const void* w1_bf16,
                                   const float* b1, const void* w2_bf16, const float* b2, const float* mask,
                                   long long ldmask, float* hid_out, long long ldhid, float* out, long long ldout,
                                   int add_act, int relu_out, int K1, int HD, int D2, const unsigned char* active, int hid_bf16,
                                   int device, void* stream) {
  MMFT_REQUIRE(x1 && rows && w1_bf16 && w2_bf16 && out, "mlp2_rows_bf16: null pointer");
  if (K1 != L2_K1 || HD != L2_HD || D2 != L2_D2) {
    set_error("mlp2_rows_bf16: only %d -> %d -> %d is fused (got %d -> %d -> %d)", L2_K1, L2_HD, L2_D2, K1, HD, D2);
    return MMFT_ERR_UNSUPPORTED;
  }
  MMFT_REQUIRE(n >= 0, "mlp2_rows_bf16: negative row count");
  MMFT_REQUIRE(ldx1 % 4 == 0 && ldout % 4 == 0 && aligned16(x1) && aligned16(out) && aligned16(w1_bf16) && aligned16(w2_bf16) &&
                   (!b1 || aligned16(b1)) && (!b2 || aligned16(b2)),
               "mlp2_rows_bf16: operands must be 16-byte aligned");
  MMFT_REQUIRE(!mask || (ldmask % 4 == 0 && aligned16(mask)), "mlp2_rows_bf16: mask alignment");
  MMFT_REQUIRE(!hid_out || (ldhid % 4 == 0 && aligned16(hid_out)), "mlp2_rows_bf16: hid_out alignment");
  if (n == 0) return MMFT_OK;
  DeviceGuard dg(device);
  Mlp2Bf16Args a{x1, ldx1, rows, n, (const unsigned short*)w1_bf16, b1, (const unsigned short*)w2_bf16, b2, mask, ldmask,
                 hid_out, ldhid, hid_bf16 ? 1 : 0, out, ldout, add_act, relu_out, active};
  const double fl = 2.0 * n * ((double)K1 * HD + (double)HD * D2), by = 4.0 * n * ((double)K1 + 2.0 * HD + 2.0 * D2);
  MMFT_LAUNCH(mask ? "mlp2_rows_bf16_kernel<bwd>" : "mlp2_rows_bf16_kernel<fwd>", fl, by, mlp2_rows_bf16_kernel,
              dim3(cdiv(n, L2_BM)), dim3(512), (hipStream_t)stream, a);
  return check_launch("mlp2_rows_bf16");
}

extern "C" int mmft_level_fwd_bf16(float* h, const float* pre, long long ld, int D, const int* in_net_indptr,
                                   const int* in_net_indices, const int* in_cell_indptr, const int* in_cell_indices,
                                   int net_row0, int n_net, const int* cell_rows, int cell_row0, int n_cell, float* A,
                                   float* LSE, const void* w1_bf16, const float* b1, const void* w2_bf16, const float* b2,
                                   float* hid_out, long long ldhid, int relu, const unsigned char* active,
                                   const int* in_cell_driver, long long alg_bytes, int hid_bf16, int device, void* stream) {
  MMFT_REQUIRE(D == L2_K1, "level_fwd_bf16: D must be %d", L2_K1);
  MMFT_REQUIRE(n_net >= 0 && n_cell >= 0 && net_row0 >= 0 && cell_row0 >= 0, "level_fwd_bf16: negative row count / offset");
  if (n_net + n_cell == 0) return MMFT_OK;
  MMFT_REQUIRE(h && pre && in_net_indptr && in_cell_indptr && (n_cell == 0 || (A && LSE && w1_bf16 && w2_bf16)),
               "level_fwd_bf16: null pointer");
  MMFT_REQUIRE(ld >= D && ld % 4 == 0 && aligned16(h) && aligned16(pre) && (!A || aligned16(A)) && (!LSE || aligned16(LSE)) &&
                   (!w1_bf16 || aligned16(w1_bf16)) && (!w2_bf16 || aligned16(w2_bf16)) && (!b1 || aligned16(b1)) &&
                   (!b2 || aligned16(b2)) && (!hid_out || (aligned16(hid_out) && ldhid % 4 == 0)),
               "level_fwd_bf16: operands must be 16-byte aligned");
  DeviceGuard dg(device);
  constexpr int LV_BM = 16;
  const int tiles = cdiv(n_cell, LV_BM);
  int net_blocks = cdiv((long long)n_net * 32, 512);
  if (net_blocks > 1024) net_blocks = 1024;
  LevelFwdArgs a{h, pre, ld, in_net_indptr, in_net_indices, in_cell_indptr, in_cell_indices, in_cell_driver, net_row0, n_net, cell_rows,
                 cell_row0, n_cell, A, LSE, (const unsigned short*)w1_bf16, (const unsigned short*)w2_bf16, b1, b2, hid_out,
                 ldhid, hid_bf16 ? 1 : 0, relu, active, tiles};
  const double fl = 2.0 * n_cell * ((double)L2_K1 * L2_HD + (double)L2_HD * L2_D2);
  MMFT_LAUNCH("level_fwd_bf16_kernel", fl, alg_bytes > 0 ? (double)alg_bytes : 0.0, level_fwd_bf16_kernel<LV_BM>,
              dim3(tiles + net_blocks), dim3(512), (hipStream_t)stream, a);
  return check_launch("level_fwd_bf16");
}

extern "C" int mmft_level_fwd_slots(float* h, const float* pre, long long ld, int D, const int* slots, const int* net_driver,
                                    int net_row0, int n_net, int cell_row0, int n_cell, float* A, float* LSE, const void* w1_bf16,
                                    const float* b1, const void* w2_bf16, const float* b2, float* hid_out, long long ldhid, int relu,
                                    const unsigned char* active, long long alg_bytes, int hid_bf16, int device, void* stream) {
  MMFT_REQUIRE(D == L2_K1, "level_fwd_slots: D must be %d", L2_K1);
  MMFT_REQUIRE(n_net >= 0 && n_cell >= 0 && net_row0 >= 0 && cell_row0 >= 0, "level_fwd_slots: negative row count / offset");
  if (n_net + n_cell == 0) return MMFT_OK;
  MMFT_REQUIRE(h && pre && slots && net_driver && (n_cell == 0 || (A && LSE && w1_bf16 && w2_bf16)), "level_fwd_slots: null pointer");
  MMFT_REQUIRE(ld >= D && ld % 4 == 0 && aligned16(h) && aligned16(pre) && aligned16(slots) && (!A || aligned16(A)) &&
                   (!LSE || aligned16(LSE)) && (!w1_bf16 || aligned16(w1_bf16)) && (!w2_bf16 || aligned16(w2_bf16)) &&
                   (!b1 || aligned16(b1)) && (!b2 || aligned16(b2)) && (!hid_out || (aligned16(hid_out) && ldhid % 4 == 0)),
               "level_fwd_slots: operands must be 16-byte aligned");
  DeviceGuard dg(device);
  // two 16-row blocks per workgroup when that still gives every CU a workgroup: the kernel takes the same 16.7 us at config B and
  // the replayed step is 0.06 ms faster (half the workgroups fetch the packed weights while the U-Net runs beside them); a
  // small level (config C: 1 250 cell rows = 78 blocks) is bound by the latency of one workgroup and takes 13.8 instead of
  // 9.7 us in that form.  MMFT_FWD_RB=1 / 2 forces one form.
  static const int rb_env = getenv("MMFT_FWD_RB") ? atoi(getenv("MMFT_FWD_RB")) : 0;
  const int rows_max = n_cell > n_net ? n_cell : n_net;
  const int rb = rb_env ? rb_env : (rows_max >= 32 * 192 ? 2 : 1);
  const int bm = rb == 2 ? 32 : 16;
  const int tiles = cdiv(n_cell, bm), net_tiles = cdiv(n_net, bm);
  LevelSlotsArgs a{h, pre, ld, slots, net_driver, net_row0, n_net, cell_row0, n_cell, A, LSE, (const unsigned short*)w1_bf16,
                   (const unsigned short*)w2_bf16, b1, b2, hid_out, ldhid, hid_bf16 ? 1 : 0, relu, active, tiles};
  const double fl = 2.0 * n_cell * ((double)L2_K1 * L2_HD + (double)L2_HD * L2_D2);
  if (rb == 2)
    MMFT_LAUNCH("level_fwd_slots_kernel", fl, alg_bytes > 0 ? (double)alg_bytes : 0.0, level_fwd_slots_kernel<2>,
                dim3(tiles > net_tiles ? tiles : net_tiles), dim3(512), (hipStream_t)stream, a);
  else
    MMFT_LAUNCH("level_fwd_slots_kernel", fl, alg_bytes > 0 ? (double)alg_bytes : 0.0, level_fwd_slots_kernel<1>,
                dim3(tiles > net_tiles ? tiles : net_tiles), dim3(512), (hipStream_t)stream, a);
  return check_launch("level_fwd_slots");
}

/* Reverse sweep of one (cell level l, net level l + 1) pair, see level_bwd_pair_kernel: tiles int32[ntiles][8] (first driver id,
 * count <= 16, part, parts, first scratch row, counter index, first / end CSR position of the tile's sinks) cover the level-l id range; the sink of out-net CSR position
 * e is row e + sink_shift; cslots int32[N][4]; scratch (fp32 [scratch_rows][128]) and counters (int32, zero before the first
 * call, left zero) serve the heavy drivers' parts.  has_mlp = 0 (level 0): only the two pulls. */
extern "C" int mmft_level_bwd_pair(float* G, const float* h, const float* A, const float* LSE, float* DA, long long ld, int D, int N,
                                   const unsigned char* own_mask, const int* tiles, int ntiles, const int* out_net_indptr,
                                   int sink_shift, const int* cslots, const int* out_cell_indptr,
                                   const int* out_cell_indices, float* scratch, int* counters, int relu, int has_mlp,
                                   const void* w1_bf16, const void* w2_bf16, const float* mask, long long ldmask, float* hid_out,
                                   long long ldhid, long long alg_bytes, int hid_bf16, int device, void* stream) {
  MMFT_REQUIRE(D == L2_K1, "level_bwd_pair: D must be %d", L2_K1);
  MMFT_REQUIRE(ntiles >= 0 && N > 0, "level_bwd_pair: negative tile count");
  if (ntiles == 0) return MMFT_OK;
  MMFT_REQUIRE(G && h && A && LSE && DA && tiles && out_net_indptr && cslots && out_cell_indptr && out_cell_indices && scratch && counters,
               "level_bwd_pair: null pointer");
  MMFT_REQUIRE(!has_mlp || (w1_bf16 && w2_bf16 && mask), "level_bwd_pair: the MLP part needs both weight packs and the saved hidden rows");
  MMFT_REQUIRE(ld >= D && ld % 4 == 0 && aligned16(G) && aligned16(h) && aligned16(A) && aligned16(LSE) && aligned16(DA) &&
                   aligned16(cslots) && aligned16(tiles) && aligned16(scratch) && (!w1_bf16 || aligned16(w1_bf16)) &&
                   (!w2_bf16 || aligned16(w2_bf16)) && (!mask || (aligned16(mask) && ldmask % 4 == 0)) &&
                   (!hid_out || (aligned16(hid_out) && ldhid % 4 == 0)),
               "level_bwd_pair: operands must be 16-byte aligned");
  DeviceGuard dg(device);
  LevelBwdPairArgs a{G, h, A, LSE, DA, ld, own_mask, tiles, out_net_indptr, sink_shift, cslots, out_cell_indptr, out_cell_indices,
                     scratch, counters, relu, has_mlp, (const unsigned short*)w1_bf16, (const unsigned short*)w2_bf16, mask, ldmask,
                     hid_out, ldhid, hid_bf16 ? 1 : 0};
  MMFT_LAUNCH("level_bwd_pair_kernel", 0.0, alg_bytes > 0 ? (double)alg_bytes : 0.0, level_bwd_pair_kernel, dim3(ntiles), dim3(512),
              (hipStream_t)stream, a);
  return check_launch("level_bwd_pair");
}

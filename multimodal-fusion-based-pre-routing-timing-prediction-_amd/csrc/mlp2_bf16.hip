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
#include "gemm_bf16.h"
#include "fold_gather.h"

namespace mmft {

constexpr int L2_BM = 32, L2_K1 = 128, L2_HD = 256, L2_D2 = 128;
constexpr int L2_XS = L2_K1 + 8, L2_HS = L2_HD + 8;        // LDS row strides in bf16 elements (multiples of 8)

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
        if (live) mk = *reinterpret_cast<const f32x4*>(a.mask + row * a.ldmask + nn);
        v.x = mk.x > 0.f ? v.x : 0.f; v.y = mk.y > 0.f ? v.y : 0.f;
        v.z = mk.z > 0.f ? v.z : 0.f; v.w = mk.w > 0.f ? v.w : 0.f;
      } else {
        if (a.b1) v += *reinterpret_cast<const f32x4*>(a.b1 + nn);
        v.x = v.x > 0.f ? v.x : 0.f; v.y = v.y > 0.f ? v.y : 0.f;
        v.z = v.z > 0.f ? v.z : 0.f; v.w = v.w > 0.f ? v.w : 0.f;
      }
      unsigned lo = pack_bf16(v.x, v.y), hi = pack_bf16(v.z, v.w);
      *reinterpret_cast<unsigned long long*>(hs + m * L2_HS + nn) = ((unsigned long long)hi << 32) | lo;
      if (a.hid_out && live) *reinterpret_cast<f32x4*>(a.hid_out + row * a.ldhid + nn) = v;
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
      SoftAcc sa;
      sa.init();
      fold_gather_edges(fs, e0, e1, 1, c, sa);
      f32x4 lv = {0.f, 0.f, 0.f, 0.f};
      if (e1 > e0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          av[j] = sa.acc[j] / sa.s[j];
          lv[j] = sa.mx[j] + logf(sa.s[j]);
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
      if (a.hid_out && live_row(m)) *reinterpret_cast<f32x4*>(a.hid_out + (long long)row_of(m) * a.ldhid + nn) = v;
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
  int relu;
  const unsigned char* active;
  int cell_tiles;
};

__global__ void __launch_bounds__(512) level_fwd_slots_kernel(LevelSlotsArgs a) {
  constexpr int BM = 16;
  __shared__ __attribute__((aligned(16))) unsigned short xs[BM * L2_XS];
  __shared__ __attribute__((aligned(16))) unsigned short hs[BM * L2_HS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, q = lane >> 4;
  const int b = blockIdx.x, gr = tid >> 5, gc = (tid & 31) * 4;              // gather / net item of this thread: (row, channel group)
  const bool has_cell = b < a.cell_tiles;
  const int m0 = b * BM;
  // ---- requests whose addresses are known now
  const int nrow = b * BM + gr, u = a.net_row0 + nrow;
  const bool net_ok = nrow < a.n_net && (!a.active || a.active[u]);
  int nd = -1;
  f32x4 npre = {0.f, 0.f, 0.f, 0.f};
  if (net_ok) {
    nd = a.net_drv[u];
    npre = *reinterpret_cast<const f32x4*>(a.pre + (long long)u * a.ld + gc);
  }
  const int v = a.cell_row0 + m0 + gr;
  const bool live = has_cell && m0 + gr < a.n_cell && (!a.active || a.active[v]);
  int hrow[4] = {-1, -1, -1, -1}, prow[4] = {-1, -1, -1, -1};
  if (live) {
    const int4 s0 = *reinterpret_cast<const int4*>(a.slots + (long long)v * 8);
    const int4 s1 = *reinterpret_cast<const int4*>(a.slots + (long long)v * 8 + 4);
    hrow[0] = s0.x; hrow[1] = s0.y; hrow[2] = s0.z; hrow[3] = s0.w;
    prow[0] = s1.x; prow[1] = s1.y; prow[2] = s1.z; prow[3] = s1.w;
  }
  bf16x8 w1f[2][4], w2f[8];
  // epilogue-2 operand of this lane: row m0 + r16, output features wave * 16 + 4 q ..
  const int ev = a.cell_row0 + m0 + r16;
  const bool elive = has_cell && m0 + r16 < a.n_cell && (!a.active || a.active[ev]);
  f32x4 hold = {0.f, 0.f, 0.f, 0.f};
  if (has_cell) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
        w1f[j][ks] = *reinterpret_cast<const bf16x8*>(a.w1 + (long long)(wave * 32 + j * 16 + r16) * L2_K1 + ks * 32 + q * 8);
    if (elive) hold = *reinterpret_cast<const f32x4*>(a.h + (long long)ev * a.ld + wave * 16 + q * 4);
  }
  // ---- net row of level l - 1: h[u] = act(PRE[u] + h[driver])   (mean over ONE in-edge = the driver's row itself)
  if (net_ok) {
    f32x4 hd = {0.f, 0.f, 0.f, 0.f};
    if (nd >= 0) hd = *reinterpret_cast<const f32x4*>(a.h + (long long)nd * a.ld + gc);
    *reinterpret_cast<f32x4*>(a.h + (long long)u * a.ld + gc) = fg_finish_net(hd, npre, a.relu);
  }
  if (!has_cell) return;
  if (a.active) {
    int any = 0;
    if (tid < BM) any = (m0 + tid < a.n_cell && a.active[a.cell_row0 + m0 + tid]) ? 1 : 0;
    if (!__syncthreads_or(any)) return;
  }
  // ---- gather of the thread's (row, channel group): the four slots requested together, consumed in edge order
  {
    f32x4 av = {0.f, 0.f, 0.f, 0.f};
    if (live) {
      const int h0 = hrow[0] >= 0 ? hrow[0] : v;             // an empty row still issues (and drops) loads of valid rows
      f32x4 xa[4], xp[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        xa[k] = *reinterpret_cast<const f32x4*>(a.h + (long long)(hrow[k] >= 0 ? hrow[k] : h0) * a.ld + gc);
        xp[k] = *reinterpret_cast<const f32x4*>(a.pre + (long long)(prow[k] >= 0 ? prow[k] : a.net_row0) * a.ld + gc);
      }
      SoftAcc sa;
      sa.init();
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const f32x4 x = prow[k] >= 0 ? fg_finish_net(xa[k], xp[k], a.relu) : xa[k];
        SoftAcc nx = sa;
        nx.add(x);
        const bool ok = hrow[k] >= 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          sa.mx[j] = ok ? nx.mx[j] : sa.mx[j];
          sa.s[j] = ok ? nx.s[j] : sa.s[j];
          sa.acc[j] = ok ? nx.acc[j] : sa.acc[j];
        }
      }
      f32x4 lv = {0.f, 0.f, 0.f, 0.f};
      if (hrow[0] >= 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          av[j] = sa.acc[j] / sa.s[j];
          lv[j] = sa.mx[j] + logf(sa.s[j]);
        }
      }
      *reinterpret_cast<f32x4*>(a.A + (long long)v * a.ld + gc) = av;
      *reinterpret_cast<f32x4*>(a.LSE + (long long)v * a.ld + gc) = lv;
    }
    const unsigned lo = pack_bf16(av.x, av.y), hi = pack_bf16(av.z, av.w);
    *reinterpret_cast<unsigned long long*>(xs + gr * L2_XS + gc) = ((unsigned long long)hi << 32) | lo;
  }
#pragma unroll
  for (int ks = 0; ks < 8; ++ks)
    w2f[ks] = *reinterpret_cast<const bf16x8*>(a.w2 + (long long)(wave * 16 + r16) * L2_HD + ks * 32 + q * 8);
  __syncthreads();
  // ---- phase 1 / epilogue 1 / phase 2 / epilogue 2: as level_fwd_bf16_kernel<16>
  f32x4 acc1[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) acc1[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const bf16x8 xf = *reinterpret_cast<const bf16x8*>(xs + r16 * L2_XS + ks * 32 + q * 8);
#pragma unroll
    for (int j = 0; j < 2; ++j) acc1[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1f[j][ks], xf, acc1[j], 0, 0, 0);
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int nn = wave * 32 + j * 16 + q * 4;
    f32x4 hv = acc1[j];
    if (a.b1) hv += *reinterpret_cast<const f32x4*>(a.b1 + nn);
    hv.x = hv.x > 0.f ? hv.x : 0.f; hv.y = hv.y > 0.f ? hv.y : 0.f;
    hv.z = hv.z > 0.f ? hv.z : 0.f; hv.w = hv.w > 0.f ? hv.w : 0.f;
    const unsigned lo = pack_bf16(hv.x, hv.y), hi = pack_bf16(hv.z, hv.w);
    *reinterpret_cast<unsigned long long*>(hs + r16 * L2_HS + nn) = ((unsigned long long)hi << 32) | lo;
    if (a.hid_out && elive) *reinterpret_cast<f32x4*>(a.hid_out + (long long)ev * a.ldhid + nn) = hv;
  }
  __syncthreads();
  f32x4 acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) {
    const bf16x8 hf = *reinterpret_cast<const bf16x8*>(hs + r16 * L2_HS + ks * 32 + q * 8);
    acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2f[ks], hf, acc2, 0, 0, 0);
  }
  if (elive) {
    const int nn = wave * 16 + q * 4;
    f32x4 o = acc2;
    if (a.b2) o += *reinterpret_cast<const f32x4*>(a.b2 + nn);
    o += hold;
    if (a.relu) {
      o.x = o.x > 0.f ? o.x : 0.f; o.y = o.y > 0.f ? o.y : 0.f;
      o.z = o.z > 0.f ? o.z : 0.f; o.w = o.w > 0.f ? o.w : 0.f;
    }
    *reinterpret_cast<f32x4*>(a.h + (long long)ev * a.ld + nn) = o;
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
                                   int add_act, int relu_out, int K1, int HD, int D2, const unsigned char* active, int device,
                                   void* stream) {
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
                 hid_out, ldhid, out, ldout, add_act, relu_out, active};
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
                                   const int* in_cell_driver, long long alg_bytes, int device, void* stream) {
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
                 ldhid, relu, active, tiles};
  const double fl = 2.0 * n_cell * ((double)L2_K1 * L2_HD + (double)L2_HD * L2_D2);
  MMFT_LAUNCH("level_fwd_bf16_kernel", fl, alg_bytes > 0 ? (double)alg_bytes : 0.0, level_fwd_bf16_kernel<LV_BM>,
              dim3(tiles + net_blocks), dim3(512), (hipStream_t)stream, a);
  return check_launch("level_fwd_bf16");
}

extern "C" int mmft_level_fwd_slots(float* h, const float* pre, long long ld, int D, const int* slots, const int* net_driver,
                                    int net_row0, int n_net, int cell_row0, int n_cell, float* A, float* LSE, const void* w1_bf16,
                                    const float* b1, const void* w2_bf16, const float* b2, float* hid_out, long long ldhid, int relu,
                                    const unsigned char* active, long long alg_bytes, int device, void* stream) {
  MMFT_REQUIRE(D == L2_K1, "level_fwd_slots: D must be %d", L2_K1);
  MMFT_REQUIRE(n_net >= 0 && n_cell >= 0 && net_row0 >= 0 && cell_row0 >= 0, "level_fwd_slots: negative row count / offset");
  if (n_net + n_cell == 0) return MMFT_OK;
  MMFT_REQUIRE(h && pre && slots && net_driver && (n_cell == 0 || (A && LSE && w1_bf16 && w2_bf16)), "level_fwd_slots: null pointer");
  MMFT_REQUIRE(ld >= D && ld % 4 == 0 && aligned16(h) && aligned16(pre) && aligned16(slots) && (!A || aligned16(A)) &&
                   (!LSE || aligned16(LSE)) && (!w1_bf16 || aligned16(w1_bf16)) && (!w2_bf16 || aligned16(w2_bf16)) &&
                   (!b1 || aligned16(b1)) && (!b2 || aligned16(b2)) && (!hid_out || (aligned16(hid_out) && ldhid % 4 == 0)),
               "level_fwd_slots: operands must be 16-byte aligned");
  DeviceGuard dg(device);
  const int tiles = cdiv(n_cell, 16), net_tiles = cdiv(n_net, 16);
  LevelSlotsArgs a{h, pre, ld, slots, net_driver, net_row0, n_net, cell_row0, n_cell, A, LSE, (const unsigned short*)w1_bf16,
                   (const unsigned short*)w2_bf16, b1, b2, hid_out, ldhid, relu, active, tiles};
  const double fl = 2.0 * n_cell * ((double)L2_K1 * L2_HD + (double)L2_HD * L2_D2);
  MMFT_LAUNCH("level_fwd_slots_kernel", fl, alg_bytes > 0 ? (double)alg_bytes : 0.0, level_fwd_slots_kernel,
              dim3(tiles > net_tiles ? tiles : net_tiles), dim3(512), (hipStream_t)stream, a);
  return check_launch("level_fwd_slots");
}

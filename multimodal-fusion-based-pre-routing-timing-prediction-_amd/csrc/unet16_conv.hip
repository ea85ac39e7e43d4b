// bf16-storage 3x3 / pad 1 convolutions of the layout U-Net (src/Unet.py:16,19: Conv2d(k=3, padding=1, bias=False)),
// every channel count of the network (3 -> 16 ... 128 -> 128, the 64 / 128-channel stages included), one tile-resident
// kernel family for forward, input gradient and weight gradient.  Generalises conv_tile.h / conv_wgrad_narrow.h:
//   * activations are bf16 in HBM: a pixel's channels arrive as 16-byte loads of 8 and go to LDS untouched (the LDS tile
//     keeps the NHWC layout, pixel pitch CI + 8 elements - conflict-free ds_read_b64 fragments at any tap shift);
//   * weights are PRE-PACKED once per step in MFMA A-fragment order (mmft_u16_pack_weights: bf16, flipped / transposed for
//     the input-gradient use), so a lane's fragment is one 8-byte load; a workgroup owns a (pixel tile, block of COB output
//     channels) pair and walks the input channels in chunks of 32, reloading the chunk's 9 x 2 x (COB / 16) fragments
//     (L2-resident) - registers stay at 72 for the weights whatever CI is;
//   * the forward epilogue rounds to bf16, stores, and reduces the tile's per-channel sum / sum of squares of the ROUNDED
//     values (the BatchNorm statistics of src/Unet.py:17,20 are taken of exactly the tensor that is stored): one
//     [2][CO] row per tile, combined in fp64 in a fixed order by mmft_u16_bn_finalize - no extra pass over the tensor;
//   * tiles are 4 x 64 pixels (W >= 64) or 8 x 32; partial tiles are masked, so any H, W works.
#include <type_traits>
#include "unet16.h"

namespace mmft {

// ------------------------------------------------------------------------------------------------ weight packing
struct U16PackDesc {
  const float* w;    // source parameter
  u16* out;          // packed destination
  int rows, K;       // logical A matrix per tap: rows x K (rows = output channels, K = reduction channels), both padded to 16
  int taps;          // 9 (3x3 convolution) or 1 (plain matrix)
  int mode;          // 0: w[(row * taps + t) * Ksrc + k]            (forward conv weight [Co][3][3][Ci]; matrix [rows][K])
                     // 1: w[(k * taps + (taps - 1 - t)) * Rsrc + row] (input gradient: flipped taps, transposed channels)
                     // 2: w[k * Rsrc + row]                          (transposed matrix, taps = 1)
                     // + 4: fragments of v_mfma_f32_16x16x32_bf16 (8 consecutive k per lane, K a multiple of 32)
  int Rsrc, Ksrc;    // real extents of the source (Ksrc = 3 for the RGB layer: padded with zeros up to K = 16)
};

// out[((mb * taps + t) * KB + kb) * 64 * KW + lane * KW + j] = A_t[16 mb + r][4 KW kb + KW q + j],  lane = 16 q + r;
// KW = 4 (v_mfma_f32_16x16x16_bf16 fragments) or 8 (16x16x32).  One thread per (fragment, lane, group of 4 k).
__global__ void __launch_bounds__(256) u16_pack_kernel(const U16PackDesc* __restrict__ descs, long long* __restrict__ counters,
                                                       int ncounters, long long inc) {
  // rides along: num_batches_tracked += inc for the network's BatchNorm layers (one launch less per forward)
  if (counters && blockIdx.x == 0 && blockIdx.y == 0 && (int)threadIdx.x < ncounters) counters[threadIdx.x] += inc;
  const U16PackDesc d = descs[blockIdx.y];
  const int mode = d.mode & 3, KW = (d.mode & 4) ? 8 : 4, G = KW / 4;
  const int KB = d.K / (4 * KW), MB = d.rows / 16;
  const long long items = (long long)MB * d.taps * KB * 64 * G;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < items; e += (long long)gridDim.x * 256) {
    const int g = (int)(e % G);
    const int lane = (int)((e / G) & 63);
    long long f = e / (64 * G);
    const int kb = (int)(f % KB);
    f /= KB;
    const int t = (int)(f % d.taps), mb = (int)(f / d.taps);
    const int r = lane & 15, q = lane >> 4;
    const int row = mb * 16 + r;
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = kb * 4 * KW + KW * q + 4 * g + j;
      float x = 0.f;
      if (row < d.Rsrc && k < d.Ksrc) {
        if (mode == 0) x = d.w[((long long)row * d.taps + t) * d.Ksrc + k];
        else if (mode == 1) x = d.w[((long long)k * d.taps + (d.taps - 1 - t)) * d.Rsrc + row];
        else x = d.w[(long long)k * d.Rsrc + row];
      }
      v[j] = x;
    }
    *reinterpret_cast<s16x4*>(d.out + e * 4) = pack_bf16x4(v[0], v[1], v[2], v[3]);
  }
}

typedef __bf16 u16_bf16x8 __attribute__((ext_vector_type(8)));
typedef short u16_s16x8 __attribute__((ext_vector_type(8)));

// ------------------------------------------------------------------------------------------------ forward / input gradient
struct U16ConvArgs {
  const void* x;       // bf16 [N][H][W][CI]  (RGB: fp32 [N][H][W][3])
  const u16* wpk;      // packed weights of the layer (all CO): [(CO / 16)][9][(CI / KC)][64][KC / 4], KC = 32 (CI >= 32) or 16
  u16* y;              // bf16 [N][H][W][CO]
  float* stats;        // [tiles][2][CO] or null
  int N, H, W, CO;
  int tiles_x, tiles_y, tiles;
};

// Input channels >= 32: the 32-channel chunk is ONE v_mfma_f32_16x16x32_bf16 per tap (8 consecutive channels per lane = one
// ds_read_b128, weights packed in the 16x16x32 fragment order); the pixel pitch CI + 16 keeps those reads conflict-free.
template <int CI, int TH, int TW>
struct U16ConvCfg {
  static constexpr bool K32 = CI >= 32;
  static constexpr int XR = TH + 2, XC = TW + 2, PIX = CI + (K32 ? 16 : 8);
  static constexpr int XS_BYTES = XR * XC * PIX * 2;
  static constexpr int RED_BYTES = 4 * 2 * 32 * 4;
  static constexpr int LDS_BYTES = XS_BYTES + RED_BYTES;
};

template <int CI, int COB, int TH, int TW, bool RGB>
__global__ void __launch_bounds__(256) u16_conv3x3_kernel(U16ConvArgs a) {
  using C = U16ConvCfg<CI, TH, TW>;
  constexpr int XR = C::XR, XC = C::XC, PIX = C::PIX, MB = COB / 16;
  constexpr bool K32 = C::K32;
  constexpr int KC = K32 ? 32 : 16, NCH = CI / KC;
  constexpr int STEPS = TH * TW / 16, SPR = TW / 16, SPW = STEPS / 4;
  static_assert(TW % 16 == 0 && CI % 16 == 0 && COB % 16 == 0 && STEPS % 4 == 0 && (!RGB || CI == 16), "granularity");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u16* xs = reinterpret_cast<u16*>(smem);                                   // [XR][XC][PIX]
  float* red = reinterpret_cast<float*>(smem + C::XS_BYTES);               // [4 waves][2][32]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int mb0 = blockIdx.y * MB, co0 = mb0 * 16;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};

  // fragments of chunk ch: [(mb0 + m)][t][ch][lane][KC / 4 elements]
  typedef typename std::conditional<K32, u16_bf16x8, s16x4>::type wfrag_t;
  wfrag_t wf[9][MB];
  auto load_weights = [&](int ch) {
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int m = 0; m < MB; ++m)
        wf[t][m] = *reinterpret_cast<const wfrag_t*>(a.wpk + ((((long long)(mb0 + m) * 9 + t) * NCH + ch) * 64 + lane) * (KC / 4));
  };
  if constexpr (NCH == 1) load_weights(0);

  // staging items: bf16 input = (halo pixel, group of 8 channels) -> one 16-byte load; RGB = one pixel (3 floats)
  constexpr int XG = RGB ? 1 : CI / 8, XI = XR * XC * XG, NX = (XI + 255) / 256;
  if constexpr (RGB) {
    for (int e = tid; e < XR * XC * PIX / 4; e += 256) reinterpret_cast<s16x4*>(xs)[e] = s16x4{0, 0, 0, 0};
    __syncthreads();
  }
  u32x4 xr[NX];
  auto request = [&](int tile) {
    const int tx = tile % a.tiles_x, ty = (tile / a.tiles_x) % a.tiles_y, img = tile / (a.tiles_x * a.tiles_y);
    const int x0 = tx * TW, y0 = ty * TH;
    const long long img0 = (long long)img * a.H;
#pragma unroll
    for (int k = 0; k < NX; ++k) {
      const int it = tid + k * 256;
      const int cg = it % XG, col = (it / XG) % XC, row = it / (XG * XC);
      const int yy = y0 - 1 + row, xx = x0 - 1 + col;
      const bool ok = it < XI && yy >= 0 && yy < a.H && xx >= 0 && xx < a.W;
      const long long pix = (img0 + (ok ? yy : y0)) * a.W + (ok ? xx : x0);       // masked items read a valid pixel
      u32x4 v = {0u, 0u, 0u, 0u};
      if constexpr (RGB) {
        const float* src = reinterpret_cast<const float*>(a.x) + pix * 3;
        const s16x4 p = pack_bf16x4(src[0], src[1], src[2], 0.f);
        const u32x2 pp = as_u32x2(p);
        v[0] = pp[0];
        v[1] = pp[1];
      } else {
        v = *reinterpret_cast<const u32x4*>(reinterpret_cast<const u16*>(a.x) + pix * CI + cg * 8);
      }
      const u32x4 z4 = {0u, 0u, 0u, 0u};
      xr[k] = ok ? v : z4;
    }
  };
  auto deposit = [&]() {
#pragma unroll
    for (int k = 0; k < NX; ++k) {
      const int it = tid + k * 256;
      if (it < XI) {
        if constexpr (RGB) {
          u32x2 p = {xr[k][0], xr[k][1]};
          *reinterpret_cast<u32x2*>(xs + it * PIX) = p;
        } else {
          *reinterpret_cast<u32x4*>(xs + (it / XG) * PIX + (it % XG) * 8) = xr[k];
        }
      }
    }
  };

  if ((int)blockIdx.x < a.tiles) request(blockIdx.x);
  for (int tile = blockIdx.x; tile < a.tiles; tile += gridDim.x) {
    deposit();
    __syncthreads();
    if (tile + (int)gridDim.x < a.tiles) request(tile + gridDim.x);
    const int tx = tile % a.tiles_x, ty = (tile / a.tiles_x) % a.tiles_y, img = tile / (a.tiles_x * a.tiles_y);
    const int x0 = tx * TW, y0 = ty * TH;
    f32x4 acc[SPW][MB];
#pragma unroll
    for (int s = 0; s < SPW; ++s)
#pragma unroll
      for (int m = 0; m < MB; ++m) acc[s][m] = zero;
#pragma unroll 1
    for (int ch = 0; ch < NCH; ++ch) {
      if constexpr (NCH > 1) load_weights(ch);
#pragma unroll
      for (int s = 0; s < SPW; ++s) {
        const int st = wave + 4 * s, row = st / SPR, c0 = (st % SPR) * 16;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx)
          {
            const u16* xp = xs + ((row + ky) * XC + c0 + r + kx) * PIX + ch * KC + (KC / 4) * q;
            if constexpr (K32) {
              const u16_bf16x8 xf = *reinterpret_cast<const u16_bf16x8*>(xp);
#pragma unroll
              for (int m = 0; m < MB; ++m) acc[s][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ky * 3 + kx][m], xf, acc[s][m], 0, 0, 0);
            } else {
              const s16x4 xf = *reinterpret_cast<const s16x4*>(xp);
#pragma unroll
              for (int m = 0; m < MB; ++m) acc[s][m] = mfma_bf16_k16(wf[ky * 3 + kx][m], xf, acc[s][m]);
            }
          }
      }
    }
    // acc[s][m] at lane (pixel r, q) = output channels co0 + 16 m + 4 q .. + 3 of pixel (y0 + row, x0 + c0 + r)
    float sm[MB][4], sq[MB][4];
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
      for (int j = 0; j < 4; ++j) sm[m][j] = sq[m][j] = 0.f;
#pragma unroll
    for (int s = 0; s < SPW; ++s) {
      const int st = wave + 4 * s, row = st / SPR, c0 = (st % SPR) * 16;
      const int yy = y0 + row, xx = x0 + c0 + r;
      const bool valid = yy < a.H && xx < a.W;
      u16* out = a.y + (((long long)img * a.H + yy) * a.W + xx) * a.CO + co0;
#pragma unroll
      for (int m = 0; m < MB; ++m) {
        const s16x4 pk = pack_bf16x4(acc[s][m]);
        if (valid) *reinterpret_cast<s16x4*>(out + m * 16 + 4 * q) = pk;
        if (a.stats) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float v = valid ? bf2f((unsigned)(unsigned short)pk[j]) : 0.f;
            sm[m][j] += v;
            sq[m][j] = __fmaf_rn(v, v, sq[m][j]);
          }
        }
      }
    }
    if (a.stats) {
#pragma unroll
      for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float s1 = sm[m][j], s2 = sq[m][j];
#pragma unroll
          for (int o = 1; o < 16; o <<= 1) {
            s1 += __shfl_xor(s1, o, 64);
            s2 += __shfl_xor(s2, o, 64);
          }
          if (r == 0) {
            red[(wave * 2 + 0) * 32 + m * 16 + 4 * q + j] = s1;
            red[(wave * 2 + 1) * 32 + m * 16 + 4 * q + j] = s2;
          }
        }
    }
    __syncthreads();
    if (a.stats && tid < 2 * COB) {
      const int which = tid / COB, c = tid % COB;
      const float t = ((red[(0 * 2 + which) * 32 + c] + red[(1 * 2 + which) * 32 + c]) + red[(2 * 2 + which) * 32 + c]) +
                      red[(3 * 2 + which) * 32 + c];
      a.stats[((long long)tile * 2 + which) * a.CO + co0 + c] = t;
    }
  }
}

// ------------------------------------------------------------------------------------------------ weight gradient
//   dw[co][ky][kx][ci] = sum over pixels p of dy[p][co] * x[p + (ky - 1, kx - 1)][ci]
// The contraction runs over PIXELS: both MFMA operands want "lane = channel, registers = consecutive pixels" while NHWC
// gives "lane = pixel, registers = channels".  gfx950 transposes on the way out of LDS: x (with its halo) and dy are
// staged in LDS in their natural [pixel][channel] layout (16-byte copies) and every operand is read with
// ds_read_b64_tr_b16 - per 16-lane group a 4 pixel x 16 channel block delivered channel-major.  Two such reads (pixels
// 4g .. 4g+3 of the first and of the second 16 pixels) form the 8-element operand of v_mfma_f32_16x16x32_bf16: a step is
// 32 consecutive pixels of an image row.  (conv_wgrad_narrow.h, the fp32-storage kernel, transposes with extra MFMAs on
// the half-rate 16x16x16 instruction and keeps 36 accumulator fragments per wave: 328 registers, ONE wave per SIMD, every
// latency exposed - 60 us for a 5 GFLOP layer.)
// Work split: blockIdx.y = (block of COB output channels, block of CIB input channels), blockIdx.x walks pixel tiles;
// inside a workgroup the 9 * CIB / 16 (tap, channel block) UNITS are dealt round-robin to the four waves, every wave
// walks all steps of the tile: <= 10 accumulator fragments per wave, and no cross-wave reduction - a wave stores its units
// straight into the workgroup's slab [CO][9][CI]; the slabs are added in a fixed order afterwards.
struct U16WgradArgs {
  const void* x;     // bf16 [N][H][W][CI]  (RGB: fp32 [N][H][W][3])
  const u16* dy;     // bf16 [N][H][W][CO]
  float* slabs;      // [gridDim.x][CO][9][CIs]   (CIs = CI, or 3 for the RGB layer)
  int N, H, W, CI, CO;
  int tiles_x, tiles_y, tiles;
};

// LDS pixel pitch (elements) for which the transposed reads of a 32-lane half - 8 consecutive pixels x 16 bytes - cover
// all 64 banks once: 16 channels -> 8 dwords, 32 channels -> 24 dwords
constexpr int u16_tr_pitch(int ch) { return ch == 16 ? 16 : 48; }

template <int CIB, int COB, int TH, int TW>
struct U16WgradCfg {
  static constexpr int XR = TH + 2, XC = TW + 2, PIXX = u16_tr_pitch(CIB), PIXD = u16_tr_pitch(COB);
  static constexpr int XS_BYTES = XR * XC * PIXX * 2, DS_BYTES = TH * TW * PIXD * 2;
  static constexpr int LDS_BYTES = XS_BYTES + DS_BYTES;
};

typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
__device__ __forceinline__ s16x4 lds_tr_read(const u16* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p);
}

__device__ __forceinline__ u16_bf16x8 join8(s16x4 lo, s16x4 hi) {
  u16_s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(u16_bf16x8, v);
}

template <int CIB, int COB, int TH, int TW, bool RGB>
__global__ void __launch_bounds__(256) u16_conv3x3_wgrad_kernel(U16WgradArgs a) {
  using C = U16WgradCfg<CIB, COB, TH, TW>;
  constexpr int XR = C::XR, XC = C::XC, PIXX = C::PIXX, PIXD = C::PIXD, CB = CIB / 16, MB = COB / 16;
  constexpr int UNITS = 9 * CB, UM = (UNITS + 3) / 4;               // (tap, channel block) units, at most UM per wave
  constexpr int SPT = TW / 32, STEPS = TH * SPT;                    // 32-pixel steps per tile row / per tile
  static_assert(TW % 32 == 0 && CIB % 16 == 0 && COB % 16 == 0 && (!RGB || CIB == 16), "granularity");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u16* xs = reinterpret_cast<u16*>(smem);                           // [XR][XC][PIXX]
  u16* ds = reinterpret_cast<u16*>(smem + C::XS_BYTES);             // [TH * TW][PIXD]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: uniform branches
  const int nci = RGB ? 1 : a.CI / CIB;
  const int co0 = ((int)blockIdx.y / nci) * COB, ci0 = ((int)blockIdx.y % nci) * CIB;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  // transposed-read addressing: lane = 16 g + 4 qq + pp supplies row (pixel) 4 g + qq, columns (channels) 4 pp .. 4 pp + 3
  const int tr_pix = 4 * (lane >> 4) + ((lane >> 2) & 3), tr_ch = 4 * (lane & 3);

  f32x4 acc[MB][UM];
#pragma unroll
  for (int m = 0; m < MB; ++m)
#pragma unroll
    for (int i = 0; i < UM; ++i) acc[m][i] = zero;

  constexpr int XG = RGB ? 1 : CIB / 8, XI = XR * XC * XG, NX = (XI + 255) / 256;
  constexpr int DG = COB / 8, DI = TH * TW * DG, ND = (DI + 255) / 256;
  if constexpr (RGB) {
    for (int e = tid; e < XR * XC * PIXX / 4; e += 256) reinterpret_cast<s16x4*>(xs)[e] = s16x4{0, 0, 0, 0};
    __syncthreads();
  }
  u32x4 xr[NX], dr[ND];
  auto request = [&](int tile) {
    const int tx = tile % a.tiles_x, ty = (tile / a.tiles_x) % a.tiles_y, img = tile / (a.tiles_x * a.tiles_y);
    const int x0 = tx * TW, y0 = ty * TH;
    const long long img0 = (long long)img * a.H;
    const u32x4 z4 = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int k = 0; k < NX; ++k) {
      const int it = tid + k * 256;
      const int cg = it % XG, col = (it / XG) % XC, row = it / (XG * XC);
      const int yy = y0 - 1 + row, xx = x0 - 1 + col;
      const bool ok = it < XI && yy >= 0 && yy < a.H && xx >= 0 && xx < a.W;
      const long long pix = (img0 + (ok ? yy : y0)) * a.W + (ok ? xx : x0);
      u32x4 v = z4;
      if constexpr (RGB) {
        const float* src = reinterpret_cast<const float*>(a.x) + pix * 3;
        const u32x2 pp = as_u32x2(pack_bf16x4(src[0], src[1], src[2], 0.f));
        v[0] = pp[0];
        v[1] = pp[1];
      } else {
        v = *reinterpret_cast<const u32x4*>(reinterpret_cast<const u16*>(a.x) + pix * a.CI + ci0 + cg * 8);
      }
      xr[k] = ok ? v : z4;
    }
#pragma unroll
    for (int k = 0; k < ND; ++k) {
      const int it = tid + k * 256;
      const int cg = it % DG, col = (it / DG) % TW, row = it / (DG * TW);
      const int yy = y0 + row, xx = x0 + col;
      const bool ok = it < DI && yy < a.H && xx < a.W;
      const long long pix = (img0 + (ok ? yy : y0)) * a.W + (ok ? xx : x0);
      const u32x4 v = *reinterpret_cast<const u32x4*>(a.dy + pix * a.CO + co0 + cg * 8);
      dr[k] = ok ? v : z4;
    }
  };
  auto deposit = [&]() {
#pragma unroll
    for (int k = 0; k < NX; ++k) {
      const int it = tid + k * 256;
      if (it < XI) {
        if constexpr (RGB) {
          u32x2 p = {xr[k][0], xr[k][1]};
          *reinterpret_cast<u32x2*>(xs + it * PIXX) = p;
        } else {
          *reinterpret_cast<u32x4*>(xs + (it / XG) * PIXX + (it % XG) * 8) = xr[k];
        }
      }
    }
#pragma unroll
    for (int k = 0; k < ND; ++k) {
      const int it = tid + k * 256;
      if (it < DI) *reinterpret_cast<u32x4*>(ds + (it / DG) * PIXD + (it % DG) * 8) = dr[k];
    }
  };

  if ((int)blockIdx.x < a.tiles) request(blockIdx.x);
  for (int tile = blockIdx.x; tile < a.tiles; tile += gridDim.x) {
    deposit();
    __syncthreads();
    if (tile + (int)gridDim.x < a.tiles) request(tile + gridDim.x);
#pragma unroll 2
    for (int st = 0; st < STEPS; ++st) {
      const int row = st / SPT, c0 = (st % SPT) * 32;
      // A operands: dy^T of the step's 32 pixels, lane = output channel, k slots (g, 0..3) = pixels 4g.., (g, 4..7) = 16 + 4g..
      u16_bf16x8 af[MB];
      const u16* dbase = ds + (row * TW + c0 + tr_pix) * PIXD + tr_ch;
#pragma unroll
      for (int m = 0; m < MB; ++m) af[m] = join8(lds_tr_read(dbase + m * 16), lds_tr_read(dbase + 16 * PIXD + m * 16));
#pragma unroll
      for (int i = 0; i < UM; ++i) {
        const int u = wave + 4 * i;                        // wave-uniform
        if (u < UNITS) {
          const int t = u / CB, c = u % CB, ky = t / 3, kx = t % 3;
          const u16* xb = xs + ((row + ky) * XC + c0 + kx + tr_pix) * PIXX + c * 16 + tr_ch;
          const u16_bf16x8 bfr = join8(lds_tr_read(xb), lds_tr_read(xb + 16 * PIXX));
#pragma unroll
          for (int m = 0; m < MB; ++m) acc[m][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[m], bfr, acc[m][i], 0, 0, 0);
        }
      }
    }
    __syncthreads();
  }

  // acc[m][i][j] at lane (r, q) = dw[co0 + 16 m + 4 q + j][t][ci0 + 16 c + r] of unit u = wave + 4 i = (t, c)
  const int r = lane & 15, q = lane >> 4;
  const int CIs = RGB ? 3 : a.CI;
  float* out = a.slabs + (long long)blockIdx.x * ((long long)a.CO * 9 * CIs);
#pragma unroll
  for (int i = 0; i < UM; ++i) {
    const int u = wave + 4 * i;
    if (u < UNITS) {
      const int t = u / CB, c = u % CB;
#pragma unroll
      for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int co = co0 + m * 16 + 4 * q + j, ci = c * 16 + r;
          if (!RGB || ci < 3) out[((long long)co * 9 + t) * CIs + ci0 + ci] = acc[m][i][j];
        }
    }
  }
}

// ------------------------------------------------------------------------------------------------ host side
static inline void u16_tile_shape(int W, int* th, int* tw) {
  *tw = W >= 64 ? 64 : 32;
  *th = 256 / *tw;
}

template <int CI, int COB, int TH, int TW, bool RGB>
static int u16_conv_launch_t(const U16ConvArgs& a, int grid_x, hipStream_t st, double flops, double bytes) {
  using C = U16ConvCfg<CI, TH, TW>;
  static DynLdsOnce once;
  int rc = ensure_dyn_lds(once, reinterpret_cast<const void*>(&u16_conv3x3_kernel<CI, COB, TH, TW, RGB>), C::LDS_BYTES, "u16_conv3x3");
  if (rc) return rc;
  static char name[64];                                  // per instantiation; written with the same bytes by every caller
  if (!name[0]) snprintf(name, sizeof(name), "u16_conv3x3_kernel<%d,%d,%dx%d%s>", RGB ? 3 : CI, COB, TH, TW, a.stats ? "" : "");
  MMFT_LAUNCH_LDS(name, flops, bytes, (u16_conv3x3_kernel<CI, COB, TH, TW, RGB>), dim3(grid_x, a.CO / COB), dim3(256),
                  C::LDS_BYTES, st, a);
  return check_launch("u16_conv3x3");
}

template <int CIB, int COB, int TH, int TW, bool RGB>
static int u16_wgrad_launch_t(const U16WgradArgs& a, int grid_x, int grid_y, hipStream_t st, double flops, double bytes) {
  using C = U16WgradCfg<CIB, COB, TH, TW>;
  static DynLdsOnce once;
  int rc = ensure_dyn_lds(once, reinterpret_cast<const void*>(&u16_conv3x3_wgrad_kernel<CIB, COB, TH, TW, RGB>), C::LDS_BYTES,
                          "u16_conv3x3_wgrad");
  if (rc) return rc;
  static char name[64];
  if (!name[0]) snprintf(name, sizeof(name), "u16_conv3x3_wgrad_kernel<%d,%d,%dx%d>", RGB ? 3 : CIB, COB, TH, TW);
  MMFT_LAUNCH_LDS(name, flops, bytes, (u16_conv3x3_wgrad_kernel<CIB, COB, TH, TW, RGB>), dim3(grid_x, grid_y),
                  dim3(256), C::LDS_BYTES, st, a);
  return check_launch("u16_conv3x3_wgrad");
}

static inline bool u16_channels_ok(int c) { return c == 16 || c == 32 || c == 64 || c == 128; }

static inline int u16_wgrad_grid_x(int tiles, int grid_y) {
  int cap = 512 / grid_y;                               // ~2 workgroups per CU chip-wide; every workgroup writes (and the
  if (cap < 8) cap = 8;                                 // reduction re-reads) one slab, so no more of them than needed
  int g = (tiles + 1) / 2;                              // at least two tiles per workgroup
  if (g > cap) g = cap;
  return g < 1 ? 1 : g;
}

}  // namespace mmft

using namespace mmft;

extern "C" {

int mmft_u16_pack_weights(const void* descs, int n, long long max_frag_lanes, long long* counters, int ncounters, long long inc,
                          int device, void* stream) {
  MMFT_REQUIRE(descs && n > 0 && max_frag_lanes > 0 && ncounters >= 0 && ncounters <= 256, "u16_pack_weights: bad arguments");
  DeviceGuard dg(device);
  long long gx = (max_frag_lanes + 255) / 256;
  if (gx > 64) gx = 64;
  MMFT_LAUNCH("u16_pack_kernel", 0.0, 6.0 * 4.0 * max_frag_lanes * n / 2, u16_pack_kernel, dim3((int)gx, n), dim3(256), (hipStream_t)stream,
              reinterpret_cast<const U16PackDesc*>(descs), counters, ncounters, inc);
  return check_launch("u16_pack_weights");
}

int mmft_u16_pack_desc_bytes(void) { return (int)sizeof(U16PackDesc); }

/* tiles a launch of mmft_u16_conv3x3 / _wgrad cuts N x H x W into (rows of the statistics buffer); *per_image = tiles per image */
int mmft_u16_conv_tiles(int N, int H, int W, int* per_image) {
  int th, tw;
  u16_tile_shape(W, &th, &tw);
  const int t = ((H + th - 1) / th) * ((W + tw - 1) / tw);
  if (per_image) *per_image = t;
  return N * t;
}

int mmft_u16_conv3x3(const void* x, int rgb_f32, const void* wpk, void* y, float* stats, int N, int H, int W, int Ci, int Co,
                     int device, void* stream) {
  MMFT_REQUIRE(x && wpk && y && N > 0 && H > 0 && W > 0, "u16_conv3x3: bad arguments");
  MMFT_REQUIRE(u16_channels_ok(Co) && (rgb_f32 ? (Ci == 3 && Co == 16) : u16_channels_ok(Ci)),
               "u16_conv3x3: channels must be 16 / 32 / 64 / 128 (or the 3 -> 16 RGB layer)");
  MMFT_REQUIRE(aligned16(x) && aligned16(wpk) && aligned16(y), "u16_conv3x3: 16-byte alignment");
  DeviceGuard dg(device);
  hipStream_t st = (hipStream_t)stream;
  int th, tw;
  u16_tile_shape(W, &th, &tw);
  U16ConvArgs a{x, reinterpret_cast<const u16*>(wpk), reinterpret_cast<u16*>(y), stats, N, H, W, Co, (W + tw - 1) / tw, (H + th - 1) / th, 0};
  a.tiles = N * a.tiles_x * a.tiles_y;
  const int cob = Co % 32 == 0 ? 32 : 16, gy = Co / cob;
  int cap = 2048 / gy;
  const int gx = a.tiles < cap ? a.tiles : cap;
  const double flops = 2.0 * N * H * W * Co * 9.0 * Ci;
  const double bytes = (rgb_f32 ? 4.0 : 2.0) * N * H * W * Ci + 2.0 * N * H * W * Co + 2.0 * Co * 9 * Ci;
#define U16_CONV(CIV, COBV, RGBV)                                                                           \
  (tw == 64 ? u16_conv_launch_t<CIV, COBV, 4, 64, RGBV>(a, gx, st, flops, bytes)                            \
            : u16_conv_launch_t<CIV, COBV, 8, 32, RGBV>(a, gx, st, flops, bytes))
  if (rgb_f32) return U16_CONV(16, 16, true);
  if (cob == 16) {
    if (Ci == 16) return U16_CONV(16, 16, false);
    if (Ci == 32) return U16_CONV(32, 16, false);
    if (Ci == 64) return U16_CONV(64, 16, false);
    return U16_CONV(128, 16, false);
  }
  if (Ci == 16) return U16_CONV(16, 32, false);
  if (Ci == 32) return U16_CONV(32, 32, false);
  if (Ci == 64) return U16_CONV(64, 32, false);
  return U16_CONV(128, 32, false);
#undef U16_CONV
}

long long mmft_u16_conv3x3_wgrad_workspace_bytes(int N, int H, int W, int Ci, int Co) {
  int th, tw;
  u16_tile_shape(W, &th, &tw);
  const int tiles = N * ((H + th - 1) / th) * ((W + tw - 1) / tw);
  const int cib = Ci == 3 ? 16 : (Ci % 32 == 0 ? 32 : 16), cob = Co % 32 == 0 ? 32 : 16;
  const int gy = (Co / cob) * (Ci == 3 ? 1 : Ci / cib);
  return (long long)u16_wgrad_grid_x(tiles, gy) * Co * 9 * Ci * 4;
}

/* number of slabs (Co * 9 * Ci floats each, back to back) the weight-gradient kernel leaves in its workspace */
int mmft_u16_conv3x3_wgrad_slabs(int N, int H, int W, int Ci, int Co) {
  return (int)(mmft_u16_conv3x3_wgrad_workspace_bytes(N, H, W, Ci, Co) / ((long long)Co * 9 * Ci * 4));
}

int mmft_u16_conv3x3_wgrad(const void* x, int rgb_f32, const void* dy, float* dw, int accumulate, int N, int H, int W, int Ci,
                           int Co, float* workspace, long long workspace_bytes, int device, void* stream) {
  MMFT_REQUIRE(x && dy && N > 0 && H > 0 && W > 0, "u16_conv3x3_wgrad: bad arguments");
  MMFT_REQUIRE(u16_channels_ok(Co) && (rgb_f32 ? (Ci == 3 && Co == 16) : u16_channels_ok(Ci)),
               "u16_conv3x3_wgrad: channels must be 16 / 32 / 64 / 128 (or the 3 -> 16 RGB layer)");
  MMFT_REQUIRE(workspace && workspace_bytes >= mmft_u16_conv3x3_wgrad_workspace_bytes(N, H, W, Ci, Co) && aligned16(workspace) &&
                   aligned16(x) && aligned16(dy) && (!dw || aligned16(dw)),
               "u16_conv3x3_wgrad: workspace too small or operands not 16-byte aligned");
  DeviceGuard dg(device);
  hipStream_t st = (hipStream_t)stream;
  int th, tw;
  u16_tile_shape(W, &th, &tw);
  U16WgradArgs a{x, reinterpret_cast<const u16*>(dy), workspace, N, H, W, Ci, Co, (W + tw - 1) / tw, (H + th - 1) / th, 0};
  a.tiles = N * a.tiles_x * a.tiles_y;
  const int cib = rgb_f32 ? 16 : (Ci % 32 == 0 ? 32 : 16), cob = Co % 32 == 0 ? 32 : 16;
  const int gy = (Co / cob) * (rgb_f32 ? 1 : Ci / cib);
  const int gx = u16_wgrad_grid_x(a.tiles, gy);
  const double flops = 2.0 * N * H * W * Co * 9.0 * Ci;
  const double bytes = (rgb_f32 ? 4.0 : 2.0) * N * H * W * Ci + 2.0 * N * H * W * Co;
  int rc;
#define U16_WG(CIBV, COBV, RGBV)                                                                            \
  (tw == 64 ? u16_wgrad_launch_t<CIBV, COBV, 4, 64, RGBV>(a, gx, gy, st, flops, bytes)                      \
            : u16_wgrad_launch_t<CIBV, COBV, 8, 32, RGBV>(a, gx, gy, st, flops, bytes))
  if (rgb_f32) rc = U16_WG(16, 16, true);
  else if (cib == 16 && cob == 16) rc = U16_WG(16, 16, false);
  else if (cib == 16) rc = U16_WG(16, 32, false);
  else if (cob == 16) rc = U16_WG(32, 16, false);
  else rc = U16_WG(32, 32, false);
#undef U16_WG
  if (rc || !dw) return rc;          // dw == NULL: the slabs stay in `workspace` for mmft_slab_reduce_batch
  return launch_slab_reduce(workspace, gx, (long long)Co * 9 * Ci, dw, accumulate, st);
}

}  // extern "C"

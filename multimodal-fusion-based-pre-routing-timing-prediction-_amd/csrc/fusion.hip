// Fusion-head and step-glue kernels: masked projection of the CNN feature map (CSR masks, never the
// dense T x P path map), MSE loss + gradient, flat fused Adam.
// Replaces reference src/train.py:500-501 + src/model.py:271-272, src/train.py:32,522 and :431-435,555.
#include "common.h"

namespace mmft {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// defined in linear.hip: deterministic fixed-order sum of slabs
int launch_slab_reduce(const float* slabs, int splits, long long elems, float* out, int accumulate, hipStream_t st);

__global__ void __launch_bounds__(256) transpose_kernel(const float* __restrict__ src, float* __restrict__ dst, int R,
                                                        int C) {
  __shared__ float tile[32][33];
  int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
  for (int j = ty; j < 32; j += 8) {
    int r = r0 + j, c = c0 + tx;
    tile[j][tx] = (r < R && c < C) ? src[(long long)r * C + c] : 0.f;
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    int c = c0 + j, r = r0 + tx;
    if (c < C && r < R) dst[(long long)c * R + r] = tile[tx][j];
  }
}

// one workgroup per batch row t: 256 threads = J nnz lanes x (Dout/4) channel groups; every thread gathers
// 16-B pieces of wT rows for its share of the mask's non-zeros, partial sums are combined through LDS
__global__ void __launch_bounds__(256) masked_fc_fwd_kernel(const int* __restrict__ indptr, const int* __restrict__ cols,
                                                            const int* __restrict__ paths,
                                                            const int* __restrict__ foff, int T,
                                                            const float* __restrict__ f, const float* __restrict__ wT,
                                                            const float* __restrict__ bias, float* __restrict__ out,
                                                            int Dout) {
  __shared__ f32x4 part[256];
  const int groups = Dout >> 2;
  const int J = 256 / groups;
  const int t = blockIdx.x;
  const int j = threadIdx.x / groups, c4 = threadIdx.x - j * groups;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (j < J) {
    int q = paths[t];
    const float* fb = f + (foff ? foff[t] : 0);
    int e1 = indptr[q + 1];
    int e = indptr[q] + j;
    for (; e + 3 * J < e1; e += 4 * J) {            // four gathers in flight per thread
      int p0 = cols[e], p1 = cols[e + J], p2 = cols[e + 2 * J], p3 = cols[e + 3 * J];
      f32x4 r0 = *reinterpret_cast<const f32x4*>(wT + (long long)p0 * Dout + c4 * 4);
      f32x4 r1 = *reinterpret_cast<const f32x4*>(wT + (long long)p1 * Dout + c4 * 4);
      f32x4 r2 = *reinterpret_cast<const f32x4*>(wT + (long long)p2 * Dout + c4 * 4);
      f32x4 r3 = *reinterpret_cast<const f32x4*>(wT + (long long)p3 * Dout + c4 * 4);
      acc += r0 * fb[p0];
      acc += r1 * fb[p1];
      acc += r2 * fb[p2];
      acc += r3 * fb[p3];
    }
    for (; e < e1; e += J) {
      int p = cols[e];
      acc += *reinterpret_cast<const f32x4*>(wT + (long long)p * Dout + c4 * 4) * fb[p];
    }
  }
  part[threadIdx.x] = acc;
  __syncthreads();
  if (threadIdx.x < groups) {
    f32x4 s = bias ? *reinterpret_cast<const f32x4*>(bias + threadIdx.x * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    for (int jj = 0; jj < J; ++jj) s += part[jj * groups + threadIdx.x];      // fixed order: reproducible
    *reinterpret_cast<f32x4*>(out + (long long)t * Dout + threadIdx.x * 4) = s;
  }
}

// ---- the same projection through block-prefix sums (path masks are unions of boxes = runs of consecutive cells) ----
// GP[b*P + c][:] = sum over the cells c' <= c of c's block of S consecutive cells of f[b*P + c'] * wT[c'][:].
// One thread per (design, block, 4-channel group) scans its S cells; the loads do not depend on the running sum.
__global__ void __launch_bounds__(256) MMFT_NO_PACKED_F32 fc_prefix_kernel(const float* __restrict__ f, const float* __restrict__ wT,
                                                        float* __restrict__ GP, int B, int P, int Dout, int S) {
  const int groups = Dout >> 2, nblk = P / S;
  const long long total = (long long)B * nblk * groups;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
    int c4 = (int)(t % groups);
    long long rest = t / groups;
    int blk = (int)(rest % nblk), b = (int)(rest / nblk);
    const float* w = wT + (long long)blk * S * Dout + c4 * 4;
    const float* fb = f + (long long)b * P + (long long)blk * S;
    float* g = GP + ((long long)b * P + (long long)blk * S) * Dout + c4 * 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
    for (int y = 0; y < S; ++y) {
      acc += *reinterpret_cast<const f32x4*>(w + (long long)y * Dout) * fb[y];
      *reinterpret_cast<f32x4*>(g + (long long)y * Dout) = acc;
    }
  }
}

// one workgroup per batch row: a run [s, e] of the mask contributes GP[e] - GP[s-1] (GP[e] alone at a block start):
// two 16-byte reads per thread and run instead of one per cell; partials combined through LDS in fixed order
__global__ void __launch_bounds__(256) masked_fc_fwd_runs_kernel(const int* __restrict__ run_ptr,
                                                                 const int* __restrict__ run_start,
                                                                 const int* __restrict__ run_len,
                                                                 const int* __restrict__ paths,
                                                                 const int* __restrict__ foff, int T,
                                                                 const float* __restrict__ GP,
                                                                 const float* __restrict__ bias, float* __restrict__ out,
                                                                 int Dout, int S, long long ldo) {
  __shared__ f32x4 part[256];
  const int groups = Dout >> 2;
  const int J = 256 / groups;
  const int t = blockIdx.x;
  const int j = threadIdx.x / groups, c4 = threadIdx.x - j * groups;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (j < J) {
    int q = paths[t];
    const float* gb = GP + (long long)(foff ? foff[t] : 0) * Dout + c4 * 4;
    int r1 = run_ptr[q + 1];
    int r = run_ptr[q] + j;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    for (; r + J < r1; r += 2 * J) {                 // two runs per trip: their index -> row chains overlap
      int s0 = run_start[r], e0 = s0 + run_len[r] - 1;
      int s1 = run_start[r + J], e1 = s1 + run_len[r + J] - 1;
      f32x4 h0 = *reinterpret_cast<const f32x4*>(gb + (long long)e0 * Dout);
      f32x4 h1 = *reinterpret_cast<const f32x4*>(gb + (long long)e1 * Dout);
      f32x4 l0 = (s0 % S) ? *reinterpret_cast<const f32x4*>(gb + (long long)(s0 - 1) * Dout) : z;
      f32x4 l1 = (s1 % S) ? *reinterpret_cast<const f32x4*>(gb + (long long)(s1 - 1) * Dout) : z;
      acc += h0 - l0;
      acc += h1 - l1;
    }
    for (; r < r1; r += J) {
      int s = run_start[r], e = s + run_len[r] - 1;
      f32x4 hi = *reinterpret_cast<const f32x4*>(gb + (long long)e * Dout);
      f32x4 lo = (s % S) ? *reinterpret_cast<const f32x4*>(gb + (long long)(s - 1) * Dout) : z;
      acc += hi - lo;
    }
  }
  part[threadIdx.x] = acc;
  __syncthreads();
  if (threadIdx.x < groups) {
    f32x4 sum = bias ? *reinterpret_cast<const f32x4*>(bias + threadIdx.x * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    for (int jj = 0; jj < J; ++jj) sum += part[jj * groups + threadIdx.x];
    *reinterpret_cast<f32x4*>(out + (long long)t * ldo + threadIdx.x * 4) = sum;
  }
}

// out[t][0 .. D) = row[0 .. D) for every t (the level embedding of mlp_alpha, the same for all endpoints of a level)
__global__ void __launch_bounds__(256) bcast_row_kernel(const float* __restrict__ row, int D, float* __restrict__ out, long long ldo, int T) {
  const long long total = (long long)T * D;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256)
    out[(i / D) * ldo + i % D] = row[i % D];
}

// ---- backward of the run form --------------------------------------------------------------------------------
// One workgroup per (design, block of S cells), 256 threads = J entry lanes x GROUPS channel groups.
// Stage 1: D[c] = sum of gout over the batch rows whose path has a run ENDING at cell c, minus those with a run STARTING
// at c + 1 inside the block (`code` >= 0: path id, run end; < 0: path id = -code - 1, run start).  Entry lane j owns the
// cells [j S/J, (j+1) S/J) and walks their boundary lists with four index -> batch-row -> gradient-row chains in flight;
// D stays in LDS.  Stage 2 (every lane: local suffix sums of its cells + the later lanes' totals): suffix sums of D give
// dg[c] = d loss / d(f[c] wT[c]); emit the design's dwT slab (f dg) and df[c] = <dg, wT[c]> (wave-shuffle reduction over the
// GROUPS lanes).  Fixed orders.
constexpr int MFB_THREADS = 512;      // 16 entry lanes x 32 channel groups at Dout = 128: 4 cells per lane (256: 246 us, 512: 192 us, 1024: 231 us)
template <int GROUPS>
__global__ void __launch_bounds__(MFB_THREADS) masked_fc_bwd_runs_kernel(const int* __restrict__ bptr, const int* __restrict__ bcode,
                                                                 const int* __restrict__ first, const int* __restrict__ next,
                                                                 const float* __restrict__ gout, long long ldg, const float* __restrict__ f,
                                                                 const float* __restrict__ wT, float* __restrict__ dwT,
                                                                 float* __restrict__ df, int P, int S) {
  constexpr int Dout = GROUPS * 4, J = MFB_THREADS / GROUPS;
  extern __shared__ __attribute__((aligned(16))) float dl[];            // [S][Dout]
  const int b = blockIdx.y, blk = blockIdx.x;
  const int j = threadIdx.x / GROUPS, c4 = threadIdx.x % GROUPS;
  const long long cell0 = (long long)b * P + (long long)blk * S;
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  auto rowsum = [&](int t) {                        // duplicates of a path in the batch: rare, summed in batch order
    f32x4 sum = z;
    for (; t >= 0; t = next[t]) sum += *reinterpret_cast<const f32x4*>(gout + (long long)t * ldg + c4 * 4);
    return sum;
  };
  const int cpl = S / J;
  // the boundary-list pointers of the lane's cells in one go (they are consecutive): the cells' chains start together
  int eb[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) eb[i] = (cpl <= 8 && i <= cpl) ? bptr[cell0 + j * cpl + i] : 0;
  for (int y = j * cpl; y < (j + 1) * cpl; ++y) {
    f32x4 acc = z;
    int e, e1;
    if (cpl <= 8) {
      e = e1 = 0;
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (i == y - j * cpl) { e = eb[i]; e1 = eb[i + 1]; }
    } else {
      e = bptr[cell0 + y];
      e1 = bptr[cell0 + y + 1];
    }
    for (; e + 4 <= e1; e += 4) {
      int k0 = bcode[e], k1 = bcode[e + 1], k2 = bcode[e + 2], k3 = bcode[e + 3];
      int t0 = first[k0 >= 0 ? k0 : -k0 - 1], t1 = first[k1 >= 0 ? k1 : -k1 - 1];
      int t2 = first[k2 >= 0 ? k2 : -k2 - 1], t3 = first[k3 >= 0 ? k3 : -k3 - 1];
      f32x4 r0 = t0 >= 0 ? *reinterpret_cast<const f32x4*>(gout + (long long)t0 * ldg + c4 * 4) : z;
      f32x4 r1 = t1 >= 0 ? *reinterpret_cast<const f32x4*>(gout + (long long)t1 * ldg + c4 * 4) : z;
      f32x4 r2 = t2 >= 0 ? *reinterpret_cast<const f32x4*>(gout + (long long)t2 * ldg + c4 * 4) : z;
      f32x4 r3 = t3 >= 0 ? *reinterpret_cast<const f32x4*>(gout + (long long)t3 * ldg + c4 * 4) : z;
      if (t0 >= 0) r0 += rowsum(next[t0]);
      if (t1 >= 0) r1 += rowsum(next[t1]);
      if (t2 >= 0) r2 += rowsum(next[t2]);
      if (t3 >= 0) r3 += rowsum(next[t3]);
      acc = k0 >= 0 ? acc + r0 : acc - r0;
      acc = k1 >= 0 ? acc + r1 : acc - r1;
      acc = k2 >= 0 ? acc + r2 : acc - r2;
      acc = k3 >= 0 ? acc + r3 : acc - r3;
    }
    for (; e < e1; ++e) {
      int code = bcode[e];
      f32x4 sum = rowsum(first[code >= 0 ? code : -code - 1]);
      acc = code >= 0 ? acc + sum : acc - sum;
    }
    *reinterpret_cast<f32x4*>(dl + y * Dout + c4 * 4) = acc;
  }
  // Stage 2: suffix sums over the block's cells, all entry lanes at work: lane j turns its own cells into local suffix sums,
  // the lanes' totals are added from the block's end (fixed order), and every lane finishes its cells.
  __shared__ f32x4 tot[MFB_THREADS];
  f32x4 run = z;
  for (int y = (j + 1) * cpl - 1; y >= j * cpl; --y) {
    run += *reinterpret_cast<const f32x4*>(dl + y * Dout + c4 * 4);
    *reinterpret_cast<f32x4*>(dl + y * Dout + c4 * 4) = run;
  }
  tot[threadIdx.x] = run;
  __syncthreads();
  f32x4 off = z;
  for (int jj = J - 1; jj > j; --jj) off += tot[jj * GROUPS + c4];
  for (int y = (j + 1) * cpl - 1; y >= j * cpl; --y) {
    const long long cell = cell0 + y;
    const f32x4 acc = *reinterpret_cast<const f32x4*>(dl + y * Dout + c4 * 4) + off;
    const f32x4 w = *reinterpret_cast<const f32x4*>(wT + ((long long)blk * S + y) * Dout + c4 * 4);
    *reinterpret_cast<f32x4*>(dwT + cell * Dout + c4 * 4) = acc * f[cell];
    float d = acc.x * w.x + acc.y * w.y + acc.z * w.z + acc.w * w.w;
#pragma unroll
    for (int o = GROUPS / 2; o > 0; o >>= 1) d += __shfl_xor(d, o, 64);
    if (c4 == 0) df[cell] = d;
  }
}

// one thread per (map cell p, 4-channel group), looping over the designs b: gathers the batch rows covering
// cell (b,p) through the transposed masks (fixed order -> bitwise reproducible); df by an LDS reduction
__global__ void __launch_bounds__(256) masked_fc_bwd_kernel(const int* __restrict__ cptr, const int* __restrict__ cpaths,
                                                            const int* __restrict__ first, const int* __restrict__ next,
                                                            const float* __restrict__ gout, long long ldg, const float* __restrict__ f,
                                                            const float* __restrict__ wT, float* __restrict__ dwT,
                                                            float* __restrict__ df, int B, int P, int Dout) {
  __shared__ float red[256];
  const int groups = Dout >> 2;
  const int cells_per_block = 256 / groups;
  const int lc = threadIdx.x / groups;
  const int p = blockIdx.x * cells_per_block + lc;
  const int c4 = threadIdx.x - lc * groups;
  const bool valid = lc < cells_per_block && p < P;
  f32x4 w = {0.f, 0.f, 0.f, 0.f}, dw = {0.f, 0.f, 0.f, 0.f};
  if (valid) w = *reinterpret_cast<const f32x4*>(wT + (long long)p * Dout + c4 * 4);
  for (int b = blockIdx.y; b < B; b += gridDim.y) {
    long long cell = (long long)b * P + p;
    f32x4 S = {0.f, 0.f, 0.f, 0.f};
    if (valid) {
      // four covering paths per trip: their index -> batch-row -> gradient-row load chains overlap.
      // Summation order stays fixed (e ascending, duplicates of a path in batch order): reproducible.
      int e = cptr[cell], e1 = cptr[cell + 1];
      for (; e + 4 <= e1; e += 4) {
        int t0 = first[cpaths[e]], t1 = first[cpaths[e + 1]], t2 = first[cpaths[e + 2]], t3 = first[cpaths[e + 3]];
        f32x4 z = {0.f, 0.f, 0.f, 0.f};
        f32x4 r0 = t0 >= 0 ? *reinterpret_cast<const f32x4*>(gout + (long long)t0 * ldg + c4 * 4) : z;
        f32x4 r1 = t1 >= 0 ? *reinterpret_cast<const f32x4*>(gout + (long long)t1 * ldg + c4 * 4) : z;
        f32x4 r2 = t2 >= 0 ? *reinterpret_cast<const f32x4*>(gout + (long long)t2 * ldg + c4 * 4) : z;
        f32x4 r3 = t3 >= 0 ? *reinterpret_cast<const f32x4*>(gout + (long long)t3 * ldg + c4 * 4) : z;
        S += r0;
        if (t0 >= 0) for (int t = next[t0]; t >= 0; t = next[t]) S += *reinterpret_cast<const f32x4*>(gout + (long long)t * ldg + c4 * 4);
        S += r1;
        if (t1 >= 0) for (int t = next[t1]; t >= 0; t = next[t]) S += *reinterpret_cast<const f32x4*>(gout + (long long)t * ldg + c4 * 4);
        S += r2;
        if (t2 >= 0) for (int t = next[t2]; t >= 0; t = next[t]) S += *reinterpret_cast<const f32x4*>(gout + (long long)t * ldg + c4 * 4);
        S += r3;
        if (t3 >= 0) for (int t = next[t3]; t >= 0; t = next[t]) S += *reinterpret_cast<const f32x4*>(gout + (long long)t * ldg + c4 * 4);
      }
      for (; e < e1; ++e)
        for (int t = first[cpaths[e]]; t >= 0; t = next[t])
          S += *reinterpret_cast<const f32x4*>(gout + (long long)t * ldg + c4 * 4);
      dw += S * f[cell];
    }
    __syncthreads();
    red[threadIdx.x] = w[0] * S[0] + w[1] * S[1] + w[2] * S[2] + w[3] * S[3];
    __syncthreads();
    if (valid && c4 == 0) {
      float d = 0.f;
      for (int j = 0; j < groups; ++j) d += red[lc * groups + j];
      df[cell] = d;
    }
  }
  // one slab of dwT per blockIdx.y (summed over the designs this block column handled)
  if (valid) *reinterpret_cast<f32x4*>(dwT + ((long long)blockIdx.y * P + p) * Dout + c4 * 4) = dw;
}

// idx != NULL: target[i] = table[idx[i] * ld] (labels live in (N, 1) node tensors, src/train.py:519-521)
__global__ void __launch_bounds__(1024) mse_kernel(const float* __restrict__ pred, const float* __restrict__ target, int n,
                                                   float* __restrict__ loss, float* __restrict__ grad,
                                                   const int* __restrict__ idx, long long ld) {
  __shared__ double red[1024];
  double s = 0.0;
  float inv = 2.0f / (float)n;
  // one workgroup between the forward and the backward pass of BOTH streams: four elements per thread are requested together
  // (index, then gathered target and prediction) and consumed in the old order - the sum is bit for bit the serial loop's
  for (int i0 = threadIdx.x; i0 < n; i0 += 4 * (int)blockDim.x) {
    int ii[4];
    float p[4], t[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int i = i0 + k * (int)blockDim.x;
      ii[k] = (i < n && idx) ? idx[i] : 0;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int i = i0 + k * (int)blockDim.x;
      const bool ok = i < n;
      t[k] = ok ? (idx ? target[(long long)ii[k] * ld] : target[i]) : 0.f;
      p[k] = ok ? pred[i] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int i = i0 + k * (int)blockDim.x;
      if (i < n) {
        const float d = p[k] - t[k];
        s += (double)d * (double)d;
        if (grad) grad[i] = d * inv;
      }
    }
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss[0] = (float)(red[0] / (double)n);
}

__global__ void __launch_bounds__(1024) eval_sums_kernel(const float* __restrict__ pred, const float* __restrict__ y,
                                                         const float* __restrict__ req, const float* __restrict__ label,
                                                         int n, double* __restrict__ out) {
  __shared__ double red[10][1024 / 64];
  double a[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    double p = pred[i], t = y[i];
    double d = p - t;
    a[0] += 1.0;
    a[1] += t;
    a[2] += t * t;
    a[3] += d * d;
    a[4] += fabs(d);
    if (t != 0.0) a[5] += fabs(d) / fabs(t);
    bool pc = ((double)req[i] - p) < 0.0, ac = label[i] != 0.f;
    a[6] += (pc && ac) ? 1.0 : 0.0;
    a[7] += (pc && !ac) ? 1.0 : 0.0;
    a[8] += (!pc && !ac) ? 1.0 : 0.0;
    a[9] += (!pc && ac) ? 1.0 : 0.0;
  }
#pragma unroll
  for (int k = 0; k < 10; ++k) {
    double v = a[k];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((threadIdx.x & 63) == 0) red[k][threadIdx.x >> 6] = v;
  }
  __syncthreads();
  if (threadIdx.x < 10) {
    double s = 0.0;
    for (int w = 0; w < 1024 / 64; ++w) s += red[threadIdx.x][w];
    out[threadIdx.x] = s;
  }
}

// nn.CrossEntropyLoss() (mean) over logits [n][C] and int64 class labels (task == 'cls', src/train.py:32,516-518), with the
// gradient (softmax - onehot) / n produced by the same pass; out (optional, fp64[6]) = n, sum of losses, tp, fp, tn, fn with
// predicted class = argmax (first maximum, as th.argmax) and "positive" = class != 0 (src/train.py:536-541).
__global__ void __launch_bounds__(1024) cross_entropy_kernel(const float* __restrict__ z, const long long* __restrict__ y, int n,
                                                             int C, float* __restrict__ loss, float* __restrict__ grad,
                                                             double* __restrict__ out) {
  __shared__ double red[5][1024 / 64];
  double a[5] = {0, 0, 0, 0, 0};
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const float* r = z + (long long)i * C;
    float mx = r[0];
    int arg = 0;
    for (int c = 1; c < C; ++c)
      if (r[c] > mx) {
        mx = r[c];
        arg = c;
      }
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += expf(r[c] - mx);
    const int yi = (int)y[i];
    a[0] += (double)(mx + logf(s) - r[yi]);
    if (grad) {
      const float inv = 1.0f / s, sc = 1.0f / (float)n;
      for (int c = 0; c < C; ++c) grad[(long long)i * C + c] = (expf(r[c] - mx) * inv - (c == yi ? 1.0f : 0.0f)) * sc;
    }
    const bool pp = arg != 0, ap = yi != 0;
    a[1] += (pp && ap) ? 1.0 : 0.0;
    a[2] += (pp && !ap) ? 1.0 : 0.0;
    a[3] += (!pp && !ap) ? 1.0 : 0.0;
    a[4] += (!pp && ap) ? 1.0 : 0.0;
  }
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    double v = a[k];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((threadIdx.x & 63) == 0) red[k][threadIdx.x >> 6] = v;
  }
  __syncthreads();
  if (threadIdx.x < 5) {
    double s = 0.0;
    for (int w = 0; w < 1024 / 64; ++w) s += red[threadIdx.x][w];
    if (threadIdx.x == 0 && loss) loss[0] = (float)(s / (double)n);
    if (out) {
      if (threadIdx.x == 0) out[0] = (double)n;
      out[threadIdx.x + 1] = s;
    }
  }
}

// the same ten sums per topological level (src/test.py:211-216 prints R2 and MAPE per level): block l filters the
// batch rows whose level id is l.  T is ~10^4 and L ~10^2, so every block simply scans the whole batch.
__global__ void __launch_bounds__(1024) eval_sums_by_level_kernel(const float* __restrict__ pred, const float* __restrict__ y,
                                                                  const float* __restrict__ req,
                                                                  const float* __restrict__ label,
                                                                  const int* __restrict__ level, int n,
                                                                  double* __restrict__ out) {
  __shared__ double red[10][1024 / 64];
  const int l = blockIdx.x;
  double a[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    if (level[i] != l) continue;
    double p = pred[i], t = y[i];
    double d = p - t;
    a[0] += 1.0;
    a[1] += t;
    a[2] += t * t;
    a[3] += d * d;
    a[4] += fabs(d);
    if (t != 0.0) a[5] += fabs(d) / fabs(t);
    bool pc = ((double)req[i] - p) < 0.0, ac = label[i] != 0.f;
    a[6] += (pc && ac) ? 1.0 : 0.0;
    a[7] += (pc && !ac) ? 1.0 : 0.0;
    a[8] += (!pc && !ac) ? 1.0 : 0.0;
    a[9] += (!pc && ac) ? 1.0 : 0.0;
  }
#pragma unroll
  for (int k = 0; k < 10; ++k) {
    double v = a[k];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((threadIdx.x & 63) == 0) red[k][threadIdx.x >> 6] = v;
  }
  __syncthreads();
  if (threadIdx.x < 10) {
    double s = 0.0;
    for (int w = 0; w < 1024 / 64; ++w) s += red[threadIdx.x][w];
    out[(long long)l * 10 + threadIdx.x] = s;
  }
}

__global__ void __launch_bounds__(256) adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, long long n, float step_size, float beta1,
                                                   float beta2, float eps, float wd, float bc2_sqrt, float gscale,
                                                   const float* __restrict__ dev_scalars) {
  if (dev_scalars) {
    step_size = dev_scalars[0];
    bc2_sqrt = dev_scalars[1];
  }
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    float gi = g[i] * gscale, pi = p[i];
    if (wd != 0.f) gi += wd * pi;
    float mi = m[i], vi = v[i];
    mi = mi + (gi - mi) * (1.0f - beta1);            // exp_avg.lerp_(grad, 1-beta1)
    vi = vi * beta2 + (1.0f - beta2) * gi * gi;      // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1-beta2)
    float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = pi - step_size * (mi / denom);
    m[i] = mi;
    v[i] = vi;
  }
}

// Adam with the step counter in DEVICE memory: state[0] = optimizer steps taken so far, state[1] = arrival ticket.
// Every workgroup reads the counter when it starts, derives both bias corrections itself (fp64, as torch does on the
// host) and the last workgroup to finish advances the counter - nothing step-dependent is uploaded by the host, so
// a HIP-graph replay (or a host running several steps ahead of the device) cannot pair a step with another step's
// scalars.
__global__ void __launch_bounds__(256) adam_counted_kernel(float* __restrict__ p, float* __restrict__ g,
                                                           float* __restrict__ m, float* __restrict__ v, long long n,
                                                           int* __restrict__ state, float lr, float beta1, float beta2,
                                                           float eps, float wd, float gscale, int zero_grad) {
  const int t = state[0] + 1;
  const double bc1 = 1.0 - pow((double)beta1, (double)t);
  const double bc2 = 1.0 - pow((double)beta2, (double)t);
  const float step_size = (float)((double)lr / bc1);
  const float bc2_sqrt = (float)sqrt(bc2);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    float gi = g[i] * gscale, pi = p[i];
    if (wd != 0.f) gi += wd * pi;
    float mi = m[i], vi = v[i];
    mi = mi + (gi - mi) * (1.0f - beta1);
    vi = vi * beta2 + (1.0f - beta2) * gi * gi;
    float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = pi - step_size * (mi / denom);
    m[i] = mi;
    v[i] = vi;
    if (zero_grad) g[i] = 0.f;                       // the next step's zero_grad() then costs no fill launch
  }
  __syncthreads();                                   // every thread of this workgroup has read state[0]
  if (threadIdx.x == 0) {
    int ticket = atomicAdd(state + 1, 1);
    if (ticket == (int)gridDim.x - 1) {              // last workgroup to finish: all others started (and read) before
      state[1] = 0;
      state[0] = t;
    }
  }
}

// out[t][0 .. Da + Db + Dc) = a[t] | b[t] | c[t]; one float4 per thread
__global__ void __launch_bounds__(256) concat_cols_kernel(const float* __restrict__ a, long long lda, int Da, const float* __restrict__ b,
                                                          long long ldb, int Db, const float* __restrict__ c, long long ldc, int Dc,
                                                          float* __restrict__ out, long long ldo, long long total) {
  const int G = (Da + Db + Dc) / 4;
  for (long long it = (long long)blockIdx.x * 256 + threadIdx.x; it < total; it += (long long)gridDim.x * 256) {
    const long long t = it / G;
    const int col = (int)(it % G) * 4;
    const float* src = col < Da ? a + t * lda + col : (col < Da + Db ? b + t * ldb + (col - Da) : c + t * ldc + (col - Da - Db));
    *reinterpret_cast<f32x4*>(out + t * ldo + col) = *reinterpret_cast<const f32x4*>(src);
  }
}

}  // namespace mmft

using namespace mmft;


extern "C" {

int mmft_concat_cols(const float* a, long long lda, int Da, const float* b, long long ldb, int Db, const float* c, long long ldc,
                     int Dc, float* out, long long ldo, int T, int device, void* stream) {
  MMFT_REQUIRE(a && b && out && T >= 0 && Da > 0 && Db > 0 && Dc >= 0 && (Dc == 0 || c), "concat_cols: bad arguments");
  MMFT_REQUIRE(Da % 4 == 0 && Db % 4 == 0 && Dc % 4 == 0 && lda % 4 == 0 && ldb % 4 == 0 && (Dc == 0 || ldc % 4 == 0) && ldo % 4 == 0 &&
                   aligned16(a) && aligned16(b) && (!c || aligned16(c)) && aligned16(out),
               "concat_cols: widths / strides must be multiples of 4, rows 16-byte aligned");
  if (T == 0) return MMFT_OK;
  DeviceGuard dg(device);
  const long long total = (long long)T * ((Da + Db + Dc) / 4);
  MMFT_LAUNCH("concat_cols_kernel", 0.0, 8.0 * T * (Da + Db + Dc), concat_cols_kernel, dim3(ew_grid(total)), dim3(256), (hipStream_t)stream, a,
              lda, Da, b, ldb, Db, c, ldc, Dc, out, ldo, total);
  return check_launch("concat_cols");
}

int mmft_transpose(const float* src, float* dst, int R, int C, int device, void* stream) {
  MMFT_REQUIRE(src && dst && R > 0 && C > 0, "transpose: bad args");
  DeviceGuard dg(device);
  hipLaunchKernelGGL(transpose_kernel, dim3(cdiv(C, 32), cdiv(R, 32)), dim3(256), 0, (hipStream_t)stream, src, dst, R, C);
  return check_launch("transpose");
}

int mmft_masked_fc_fwd(const int* mask_indptr, const int* mask_cols, const int* paths, const int* f_off, int T,
                       const float* f, const float* wT, const float* bias, float* out, int P, int Dout, int device, void* stream) {
  MMFT_REQUIRE(mask_indptr && paths && f && wT && out, "masked_fc_fwd: null pointer");
  MMFT_REQUIRE(T >= 0 && P > 0 && Dout > 0 && Dout % 4 == 0 && Dout <= 1024, "masked_fc_fwd: bad sizes (Dout %% 4 == 0, <= 1024)");
  MMFT_REQUIRE(aligned16(wT) && aligned16(out) && (!bias || aligned16(bias)), "masked_fc_fwd: 16-byte alignment");
  if (T == 0) return MMFT_OK;
  DeviceGuard dg(device);
  MMFT_LAUNCH("masked_fc_fwd_kernel", 0.0, 0.0, masked_fc_fwd_kernel, dim3(T), dim3(256), (hipStream_t)stream, mask_indptr, mask_cols, paths, f_off, T, f, wT, bias, out, Dout);
  return check_launch("masked_fc_fwd");
}

int mmft_masked_fc_prefix(const float* f, const float* wT, float* GP, int B, int P, int Dout, int S, int device,
                          void* stream) {
  MMFT_REQUIRE(f && wT && GP, "masked_fc_prefix: null pointer");
  MMFT_REQUIRE(B > 0 && P > 0 && S > 0 && P % S == 0 && Dout >= 4 && Dout % 4 == 0 && Dout <= 1024,
               "masked_fc_prefix: P must be a multiple of the block size, Dout a multiple of 4 in [4, 1024]");
  MMFT_REQUIRE(aligned16(wT) && aligned16(GP), "masked_fc_prefix: 16-byte alignment");
  DeviceGuard dg(device);
  long long total = (long long)B * (P / S) * (Dout / 4);
  MMFT_LAUNCH("fc_prefix_kernel", 0.0, 4.0 * ((double)B * P * Dout + (double)P * Dout + (double)B * P), fc_prefix_kernel,
              dim3(ew_grid(total)), dim3(256), (hipStream_t)stream, f, wT, GP, B, P, Dout, S);
  return check_launch("masked_fc_prefix");
}

int mmft_masked_fc_fwd_runs(const int* run_ptr, const int* run_start, const int* run_len, const int* paths,
                            const int* f_off, int T, const float* GP, const float* bias, float* out, int Dout, int S,
                            int device, void* stream) {
  MMFT_REQUIRE(T >= 0 && S > 0 && Dout >= 4 && Dout % 4 == 0 && Dout <= 1024, "masked_fc_fwd_runs: bad sizes");
  if (T == 0) return MMFT_OK;
  MMFT_REQUIRE(run_ptr && run_start && run_len && paths && GP && out, "masked_fc_fwd_runs: null pointer");
  MMFT_REQUIRE(aligned16(GP) && aligned16(out) && (!bias || aligned16(bias)), "masked_fc_fwd_runs: 16-byte alignment");
  DeviceGuard dg(device);
  MMFT_LAUNCH("masked_fc_fwd_runs_kernel", 0.0, 0.0, masked_fc_fwd_runs_kernel, dim3(T), dim3(256), (hipStream_t)stream, run_ptr,
              run_start, run_len, paths, f_off, T, GP, bias, out, Dout, S, (long long)Dout);
  return check_launch("masked_fc_fwd_runs");
}

long long mmft_head_level_workspace_bytes(int T, int Dh, int Dc, int Da, int H1) {
  return (long long)T * ((long long)Dh + Dc + Da + H1) * 4;
}

/* Predictions of ONE level call of PathModel.forward (src/model.py:269-292), without autograd state, as one entry point:
 * z = [h[targets] | fcn(path_map) | mlp_alpha(level)] -> mlp_fuse (Linear - ReLU - Linear).  Five launches, no host work in
 * between; the gradient of these predictions is produced later from the batched recomputation (model.LAZY_HEAD). */
int mmft_head_level_fwd(const float* h, long long ldh, const int* targets, int T, int Dh, const int* run_ptr, const int* run_start,
                        const int* run_len, const int* paths, const int* f_off, const float* GP, const float* fcn_bias, int Dc,
                        int S, const float* alpha_row, int Da, const float* w1, const float* b1, int H1, const float* w2,
                        const float* b2, int nout, float* workspace, long long workspace_bytes, float* out, int device, void* stream) {
  MMFT_REQUIRE(T >= 0 && Dh > 0 && Dc > 0 && Da > 0 && H1 > 0 && nout > 0, "head_level_fwd: bad sizes");
  if (T == 0) return MMFT_OK;
  MMFT_REQUIRE(h && targets && run_ptr && run_start && run_len && paths && GP && alpha_row && w1 && w2 && out && workspace,
               "head_level_fwd: null pointer");
  MMFT_REQUIRE(workspace_bytes >= mmft_head_level_workspace_bytes(T, Dh, Dc, Da, H1) && aligned16(workspace) && Dh % 4 == 0 &&
                   Dc % 4 == 0 && Da % 4 == 0,
               "head_level_fwd: workspace too small / widths not multiples of 4");
  const int Dz = Dh + Dc + Da;
  float* z = workspace;
  float* hid = workspace + (long long)T * Dz;
  int rc = mmft_gather_rows(h, ldh, targets, T, Dh, z, Dz, device, stream);
  if (rc) return rc;
  {
    MMFT_REQUIRE(aligned16(GP) && (!fcn_bias || aligned16(fcn_bias)), "head_level_fwd: 16-byte alignment");
    DeviceGuard dg(device);
    MMFT_LAUNCH("masked_fc_fwd_runs_kernel", 0.0, 0.0, masked_fc_fwd_runs_kernel, dim3(T), dim3(256), (hipStream_t)stream, run_ptr, run_start,
                run_len, paths, f_off, T, GP, fcn_bias, z + Dh, Dc, S, (long long)Dz);
    rc = check_launch("head_level_fwd: masked projection");
    if (rc) return rc;
    hipLaunchKernelGGL(bcast_row_kernel, dim3(ew_grid((long long)T * Da)), dim3(256), 0, (hipStream_t)stream, alpha_row, Da, z + Dh + Dc,
                       (long long)Dz, T);
    rc = check_launch("head_level_fwd: level embedding");
    if (rc) return rc;
  }
  rc = mmft_linear_fwd(z, nullptr, Dz, w1, Dz, b1, hid, nullptr, H1, T, H1, Dz, MMFT_EPI_STORE, MMFT_ACT_RELU, 0.f, device, stream);
  if (rc) return rc;
  return mmft_linear_fwd(hid, nullptr, H1, w2, H1, b2, out, nullptr, nout, T, nout, H1, MMFT_EPI_STORE, MMFT_ACT_NONE, 0.f, device, stream);
}

int mmft_masked_fc_bwd(const int* csc_indptr, const int* csc_paths, const int* first, const int* next, const float* gout, long long ldg,
                       const float* f, const float* wT, float* dwT, float* df, int B, int P, int Dout, float* workspace,
                       long long workspace_bytes, int device, void* stream) {
  MMFT_REQUIRE(csc_indptr && first && next && gout && f && wT && dwT && df, "masked_fc_bwd: null pointer");
  MMFT_REQUIRE(B > 0 && P > 0 && Dout >= 4 && Dout <= 1024 && Dout % 4 == 0,
               "masked_fc_bwd: Dout must be a multiple of 4 in [4, 1024]");
  MMFT_REQUIRE(aligned16(gout) && aligned16(wT) && aligned16(dwT) && ldg >= Dout && ldg % 4 == 0, "masked_fc_bwd: 16-byte alignment");
  DeviceGuard dg(device);
  int groups = Dout / 4, cpb = 256 / groups;
  hipStream_t st = (hipStream_t)stream;
  if (B == 1) {
    MMFT_LAUNCH("masked_fc_bwd_kernel", 0.0, 0.0, masked_fc_bwd_kernel, dim3(cdiv(P, cpb), 1), dim3(256), st, csc_indptr, csc_paths, first, next, gout, ldg, f, wT, dwT, df, B, P, Dout);
    return check_launch("masked_fc_bwd");
  }
  // one block column per design: B-fold parallelism; the per-design dwT slabs are summed in fixed order
  long long need = (long long)B * P * Dout * 4;
  MMFT_REQUIRE(workspace && workspace_bytes >= need && aligned16(workspace), "masked_fc_bwd: workspace too small");
  {
    MMFT_LAUNCH("masked_fc_bwd_kernel", 0.0, 0.0, masked_fc_bwd_kernel, dim3(cdiv(P, cpb), B), dim3(256), st, csc_indptr, csc_paths, first, next, gout, ldg, f, wT, workspace, df, B, P, Dout);
  }
  int rc = check_launch("masked_fc_bwd");
  if (rc) return rc;
  return launch_slab_reduce(workspace, B, (long long)P * Dout, dwT, 0, st);
}

long long mmft_masked_fc_bwd_runs_workspace_bytes(int B, int P, int Dout) {
  return B > 1 ? (long long)B * P * Dout * 4 : 0;                // per-design dwT slabs
}

int mmft_masked_fc_bwd_runs(const int* bnd_ptr, const int* bnd_code, const int* first, const int* next, const float* gout, long long ldg,
                            const float* f, const float* wT, float* dwT, float* df, int B, int P, int Dout, int S,
                            float* workspace, long long workspace_bytes, int device, void* stream) {
  MMFT_REQUIRE(bnd_ptr && bnd_code && first && next && gout && f && wT && dwT && df, "masked_fc_bwd_runs: null pointer");
  const int groups = Dout / 4;
  MMFT_REQUIRE(B > 0 && P > 0 && S > 0 && P % S == 0 && Dout % 4 == 0 && groups >= 1 && groups <= 64 &&
                   (groups & (groups - 1)) == 0 && S % (MFB_THREADS / groups) == 0 && (long long)S * Dout * 4 <= 65536,
               "masked_fc_bwd_runs: Dout / 4 must be a power of two <= 64, the block of S cells x Dout must fit 64 KB of "
               "LDS and S be a multiple of MFB_THREADS / (Dout / 4)");
  MMFT_REQUIRE(aligned16(gout) && aligned16(wT) && aligned16(dwT) && ldg >= Dout && ldg % 4 == 0, "masked_fc_bwd_runs: 16-byte alignment");
  MMFT_REQUIRE(B == 1 || (workspace && workspace_bytes >= mmft_masked_fc_bwd_runs_workspace_bytes(B, P, Dout) &&
                          aligned16(workspace)),
               "masked_fc_bwd_runs: workspace too small");
  DeviceGuard dg(device);
  hipStream_t st = (hipStream_t)stream;
  float* slabs = B > 1 ? workspace : dwT;
  const dim3 grid(P / S, B);
  const size_t lds = (size_t)S * Dout * 4;
#define MMFT_RUNS(G)                                                                                                   \
  MMFT_LAUNCH_LDS("masked_fc_bwd_runs_kernel", 0.0, 0.0, masked_fc_bwd_runs_kernel<G>, grid, dim3(MFB_THREADS), lds, st, bnd_ptr, \
                  bnd_code, first, next, gout, ldg, f, wT, slabs, df, P, S)
  switch (groups) {
    case 1: MMFT_RUNS(1); break;
    case 2: MMFT_RUNS(2); break;
    case 4: MMFT_RUNS(4); break;
    case 8: MMFT_RUNS(8); break;
    case 16: MMFT_RUNS(16); break;
    case 32: MMFT_RUNS(32); break;
    default: MMFT_RUNS(64); break;
  }
#undef MMFT_RUNS
  int rc = check_launch("masked_fc_bwd_runs");
  if (rc || B == 1) return rc;
  return launch_slab_reduce(slabs, B, (long long)P * Dout, dwT, 0, st);
}

int mmft_mse_fwd_bwd(const float* pred, const float* target, int n, float* loss, float* grad, int device, void* stream) {
  MMFT_REQUIRE(pred && target && loss && n > 0 && n <= (1 << 24), "mse_fwd_bwd: bad args");
  DeviceGuard dg(device);
  hipLaunchKernelGGL(mse_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, pred, target, n, loss, grad, (const int*)nullptr, 0LL);
  return check_launch("mse_fwd_bwd");
}

int mmft_mse_gather_fwd_bwd(const float* pred, const float* table, long long ld, const int* idx, int n, float* loss, float* grad,
                            int device, void* stream) {
  MMFT_REQUIRE(pred && table && idx && loss && n > 0 && n <= (1 << 24) && ld >= 1, "mse_gather_fwd_bwd: bad args");
  DeviceGuard dg(device);
  hipLaunchKernelGGL(mse_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, pred, table, n, loss, grad, idx, ld);
  return check_launch("mse_gather_fwd_bwd");
}

int mmft_eval_sums(const float* pred, const float* arrival, const float* required, const float* label, int n, double* out,
                   int device, void* stream) {
  MMFT_REQUIRE(pred && arrival && required && label && out && n > 0 && n <= (1 << 24), "eval_sums: bad args");
  DeviceGuard dg(device);
  hipLaunchKernelGGL(eval_sums_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, pred, arrival, required, label, n, out);
  return check_launch("eval_sums");
}

int mmft_cross_entropy_fwd_bwd(const float* logits, const long long* labels, int n, int C, float* loss, float* grad,
                               double* eval_out, int device, void* stream) {
  MMFT_REQUIRE(logits && labels && (loss || eval_out) && n > 0 && n <= (1 << 24) && C >= 2 && C <= 1024,
               "cross_entropy_fwd_bwd: bad args");
  DeviceGuard dg(device);
  hipLaunchKernelGGL(cross_entropy_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, logits, labels, n, C, loss, grad,
                     eval_out);
  return check_launch("cross_entropy_fwd_bwd");
}

int mmft_eval_sums_by_level(const float* pred, const float* arrival, const float* required, const float* label,
                            const int* level, int n, int num_levels, double* out, int device, void* stream) {
  MMFT_REQUIRE(pred && arrival && required && label && level && out && n > 0 && n <= (1 << 24) && num_levels > 0 &&
                   num_levels <= 65535,
               "eval_sums_by_level: bad args");
  DeviceGuard dg(device);
  hipLaunchKernelGGL(eval_sums_by_level_kernel, dim3(num_levels), dim3(1024), 0, (hipStream_t)stream, pred, arrival, required,
                     label, level, n, out);
  return check_launch("eval_sums_by_level");
}

int mmft_adam_step(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2, float eps,
                   float weight_decay, float bias_correction1, float bias_correction2, float gscale, int device,
                   void* stream) {
  MMFT_REQUIRE(p && g && m && v && n >= 0, "adam_step: bad args");
  MMFT_REQUIRE(bias_correction1 > 0.f && bias_correction2 > 0.f, "adam_step: bias corrections must be positive");
  if (n == 0) return MMFT_OK;
  DeviceGuard dg(device);
  MMFT_LAUNCH("adam_kernel", 0.0, 28.0 * n, adam_kernel, dim3(ew_grid(n)), dim3(256), (hipStream_t)stream, p, g, m, v, n, lr / bias_correction1, beta1, beta2, eps, weight_decay, sqrtf(bias_correction2), gscale, (const float*)nullptr);
  return check_launch("adam_step");
}

int mmft_adam_step_counted(float* p, float* g, float* m, float* v, long long n, int* state, float lr, float beta1,
                           float beta2, float eps, float weight_decay, float gscale, int zero_grad, int device, void* stream) {
  MMFT_REQUIRE(p && g && m && v && state && n >= 0, "adam_step_counted: bad args");
  MMFT_REQUIRE(beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f, "adam_step_counted: betas must be in [0, 1)");
  if (n == 0) return MMFT_OK;
  DeviceGuard dg(device);
  MMFT_LAUNCH("adam_kernel", 0.0, (zero_grad ? 32.0 : 28.0) * n, adam_counted_kernel, dim3(ew_grid(n)), dim3(256), (hipStream_t)stream, p, g, m, v, n, state, lr, beta1, beta2, eps, weight_decay, gscale, zero_grad);
  return check_launch("adam_step_counted");
}

int mmft_adam_step_dev(float* p, const float* g, float* m, float* v, long long n, const float* step_scalars, float beta1,
                       float beta2, float eps, float weight_decay, float gscale, int device, void* stream) {
  MMFT_REQUIRE(p && g && m && v && step_scalars && n >= 0, "adam_step_dev: bad args");
  if (n == 0) return MMFT_OK;
  DeviceGuard dg(device);
  MMFT_LAUNCH("adam_kernel", 0.0, 28.0 * n, adam_kernel, dim3(ew_grid(n)), dim3(256), (hipStream_t)stream, p, g, m, v, n, 0.f, beta1, beta2, eps, weight_decay, 1.f, gscale, step_scalars);
  return check_launch("adam_step_dev");
}

}  // extern "C"

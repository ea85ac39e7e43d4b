/* libmmft_hip.so  --  C ABI of the MI355X (gfx950) hot path.
 *
 * Every entry point takes raw DEVICE pointers, sizes, the HIP device ordinal and a hipStream_t (as void*).
 * The library allocates nothing persistent, keeps no pointer after return, never synchronises the
 * device and is asynchronous on the given stream.  Return value: 0 = MMFT_OK, negative = error; the
 * message is in the thread-local mmft_last_error().  No C++ exception crosses this boundary.
 * All tensors are fp32, row-major; "ld*" are row strides in ELEMENTS; index arrays are int32.
 *
 * Each group names the reference code it replaces (paths relative to the reference repository).
 */
#ifndef MMFT_H_
#define MMFT_H_

#ifdef __cplusplus
extern "C" {
#endif

#define MMFT_OK 0
#define MMFT_ERR_BAD_ARG (-1)
#define MMFT_ERR_UNSUPPORTED (-2)
#define MMFT_ERR_LAUNCH (-3)

/* epilogue modes of the dense kernels */
#define MMFT_EPI_STORE 0    /* y = act(acc + bias)               */
#define MMFT_EPI_ACCUM 1    /* y += acc + bias                   */
#define MMFT_EPI_ADD_ACT 2  /* y = act(y + acc + bias)           */
#define MMFT_ACT_NONE 0
#define MMFT_ACT_RELU 1
#define MMFT_ACT_LEAKY 2
#define MMFT_POOL_MAX 0
#define MMFT_POOL_AVG 1

int mmft_version(void);
const char* mmft_last_error(void);

/* ---------------------------------------------------------------------------------------------
 * Dense layers  --  replaces th.nn.Linear / LeakyReLU inside MLP (src/model.py:10-24), used by
 * PathConv.apply_net_func / apply_cell_func / apply_cell_func_level0 (src/model.py:88-111,138-153),
 * PathModel's fcn / mlp_alpha / mlp_fuse (src/model.py:271-292).
 * xidx / yidx (optional, may be NULL) gather input rows / scatter output rows by node id, which is
 * how the per-level row write-back `h[cur_nodes] = ...` (src/model.py:206-208) is done in place.
 * ------------------------------------------------------------------------------------------- */
/* y[yidx[m]][n] = epi( sum_k x[xidx[m]][k] * w[n][k] + bias[n] ),  w is [N][K] (torch Linear layout) */
int mmft_linear_fwd(const float* x, const int* xidx, long long ldx, const float* w, long long ldw,
                    const float* bias, float* y, const int* yidx, long long ldy, int M, int N, int K,
                    int epi_mode, int act, float slope, int device, void* stream);
/* dx[dxidx[m]][n] = sum_k g[gidx[m]][k] * w[k][n]   (w is [K=out][N=in]);
 * if mask != NULL the result is kept only where mask[maskidx[m]][n] > 0 (ReLU/LeakyReLU(0) backward
 * through the *following* activation, src/model.py:16); epi_mode STORE or ACCUM */
int mmft_linear_dgrad(const float* g, const int* gidx, long long ldg, const float* w, long long ldw,
                      float* dx, const int* dxidx, long long lddx, int M, int N, int K,
                      const float* mask, const int* maskidx, long long ldmask, int epi_mode,
                      int device, void* stream);
/* dw[o][i] (+)= sum_r g[gidx[r]][o] * x[xidx[r]][i];  split over rows into deterministic slabs in
 * `workspace` (>= mmft_linear_wgrad_workspace_bytes) that are summed in fixed order */
long long mmft_linear_wgrad_workspace_bytes(int rows, int out, int in);
int mmft_linear_wgrad(const float* g, const int* gidx, long long ldg, const float* x, const int* xidx,
                      long long ldx, float* dw, long long lddw, int rows, int out, int in, int accumulate,
                      float* workspace, long long workspace_bytes, int device, void* stream);
/* out[c] (+)= sum_r g[idx[r]][c]   (bias gradients); workspace >= mmft_colsum_workspace_bytes */
long long mmft_colsum_workspace_bytes(int rows, int cols);
int mmft_colsum(const float* g, const int* idx, long long ld, int rows, int cols, float* out, int accumulate,
                float* workspace, long long workspace_bytes, int device, void* stream);
/* dpre = dy * act'(y) given the activation OUTPUT y (valid for ReLU and LeakyReLU with slope >= 0) */
int mmft_act_bwd(const float* dy, const float* y, float* dpre, long long n, int act, float slope,
                 int device, void* stream);
/* y = act(x) elementwise */
int mmft_act_fwd(const float* x, float* y, long long n, int act, float slope, int device, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Netlist-graph aggregation  --  replaces DGL graph.pull(...) with fn.copy_src + fn.mean
 * (src/model.py:186-187) and the degree-bucketed UDF PathConv.cell_msg_reduce (src/model.py:113-116,
 * 202-204), plus the activation write-back and target gather (src/model.py:206-213).
 * CSR: indptr[N+1], indices[E]; "in" = in-edges by destination (col = source), "out" = out-edges by
 * source (col = destination).  `rows` lists the node ids of the current topological level.
 * ------------------------------------------------------------------------------------------- */
/* A[v][c] = sum_i softmax_i(h[u_i][c]) h[u_i][c];  LSE[v][c] = log sum_i exp(h[u_i][c])  (deg 0: A=0) */
int mmft_seg_softmax_sum_fwd(const float* h, long long ldh, const int* in_indptr, const int* in_indices,
                             const int* rows, int n, int D, float* A, float* LSE, long long lda,
                             int device, void* stream);
/* h[v] = act(h[v] + mean_{u->v} h[u])   (h[v] holds fc_net_self(x_net[v]) on entry; 0 for degree 0) */
int mmft_seg_mean_add_act_fwd(float* h, long long ldh, const int* in_indptr, const int* in_indices,
                              const int* rows, int n, int D, int relu, int device, void* stream);
/* out[v] = mean_{u->v} src[u]  (standalone fn.mean) */
int mmft_seg_mean_fwd(const float* src, long long lds, const int* in_indptr, const int* in_indices,
                      const int* rows, int n, int D, float* out, long long ldo, int device, void* stream);
/* Reverse sweep, one level (deterministic pull over out-edges, no atomics):
 *   gh = G[v] + sum_{w in out_net(v)} G[w]/indeg_net(w)
 *             + sum_{w in out_cell(v)} DA[w] * exp(h[v]-LSE[w]) * (1 + h[v] - A[w])
 *   G[v] = relu ? (h[v] > 0 ? gh : 0) : gh
 * G rows of later levels hold d(loss)/d(pre-activation), DA rows d(loss)/d(A); both must be zero for
 * nodes whose backward has not run. */
int mmft_level_bwd_pull(float* G, const float* h, long long ld, const int* rows, int n, int D,
                        const int* out_net_indptr, const int* out_net_indices, const int* in_net_indptr,
                        const int* out_cell_indptr, const int* out_cell_indices,
                        const float* A, const float* LSE, const float* DA, int relu,
                        int device, void* stream);
/* dst[i] = src[idx[i]]  /  dst[idx[i]] += src[i] (atomic, duplicates allowed: src/train.py:377-380) */
int mmft_gather_rows(const float* src, long long lds, const int* idx, int n, int D, float* dst, long long ldd,
                     int device, void* stream);
int mmft_scatter_add_rows(float* dst, long long ldd, const int* idx, int n, int D, const float* src,
                          long long lds, int device, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MMFT_H_ */

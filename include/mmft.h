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
/* Arithmetic of the MFMA-bound contractions (dense layers, convolutions, fused level MLPs), process-wide:
 *   MMFT_MATH_F32  (default) exact fp32: v_mfma_f32_16x16x4_f32, a k-ordered fmaf chain - the 1e-4 parity mode;
 *   MMFT_MATH_BF16 operands rounded to bf16 (round to nearest even) on their way into LDS / registers, products
 *                  summed in fp32 on v_mfma_f32_16x16x32_bf16 (16x the fp32 matrix rate) - the throughput mode
 *                  BASELINE.json configs[1] names.  Tensors in HBM stay fp32 in both modes; aggregation, BatchNorm,
 *                  loss and Adam arithmetic is fp32 in both modes. */
#define MMFT_MATH_F32 0
#define MMFT_MATH_BF16 1
int mmft_set_math_mode(int mode);
int mmft_get_math_mode(void);
/* Launch profiling for bench.py's roofline line: while enabled, every instrumented kernel launch is
 * bracketed by HIP events on ITS launch stream; mmft_prof_report writes one line per kernel name
 * ("name\tlaunches\ttotal_ms\talgorithmic_flops\talgorithmic_bytes") and returns the size needed.
 * mmft_prof_report synchronises on the recorded events (the only entry point that waits). */
int mmft_prof_enable(int on);
int mmft_prof_reset(void);
int mmft_prof_report(char* buf, int cap);
/* Algorithmic flops / bytes of the calling thread's NEXT instrumented launch that records none itself (the masked
 * projection and whole-workgroup segment sums: their work depends on per-step run / edge counts only the host has). */
int mmft_prof_hint(double flops, double bytes);
/* One device timestamp (wall_clock64: 100 MHz) written to slots[index] by a one-thread launch on `stream`: marks a point of a
 * stream inside a captured graph, where host timers and the kernel trace (which serialises dispatches) see nothing. */
int mmft_prof_stamp(unsigned long long* slots, int index, int device, void* stream);

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
/* n <= 24 slab reductions in ONE launch: out[i] (+)= sum over slabs z and over `fold` consecutive segments f of
 * slabs[z * stride + f * elems + i].  table (HOST memory) = n rows of seven 64-bit integers: slabs (device pointer), out
 * (device pointer), splits, stride (floats between slabs), elems, fold (>= 1), accumulate.  Fixed summation order. */
int mmft_slab_reduce_batch(const long long* table, int n, int device, void* stream);
/* same GEMM, and db[o] (+)= sum_r g[gidx[r]][o] from the fragments of g the kernel already holds: weight AND bias
 * gradient of one th.nn.Linear (src/model.py:13-14) from a single pass over g */
long long mmft_linear_wgrad_bias_workspace_bytes(int rows, int out, int in);
int mmft_linear_wgrad_bias(const float* g, const int* gidx, long long ldg, const float* x, const int* xidx,
                           long long ldx, float* dw, long long lddw, float* db, int rows, int out, int in,
                           int accumulate, float* workspace, long long workspace_bytes, int device, void* stream);
/* Gradients of the FIRST Linear of a Linear-ReLU-Linear MLP (reference MLP(in, 256, out), src/model.py:10-24; the
 * fc_cell_self / fc_net_self instances of PathConv, src/model.py:66-67) straight from the gradient of the MLP's
 * output, without materialising the hidden gradient:
 *   dH = (g[r] . w2) * (hid[r] > 0);  dw1[HD][fin] (+)= dH^T . x[r];  db1[HD] (+)= column sums of dH
 * over the rows r = rows ? rows[i] : row0 + i, i < n, of the node-indexed buffers g [*, D2], hid [*, HD], x [*, fin];
 * w2 is the second layer's weight [D2][HD].  fin <= 48, HD = 256, D2 = 128 (else MMFT_ERR_UNSUPPORTED). */
long long mmft_mlp2_first_layer_grads_workspace_bytes(int fin, int HD);
int mmft_mlp2_first_layer_grads(const float* g, long long ldg, const float* hid, long long ldh, const float* x,
                                long long ldx, const int* rows, int row0, int n, const float* w2, long long ldw2,
                                float* dw1, float* db1, int fin, int HD, int D2, int accumulate, float* workspace,
                                long long workspace_bytes, int device, void* stream);
/* Fused Linear-ReLU-Linear over gathered rows, hidden tile kept in LDS (one launch per cell level of the sweep;
 * PathConv.apply_cell_func's fc_cell_neigh, src/model.py:138-146, and its backward):
 *   hid = mask ? (x1[rows] . W1) * (mask[rows] > 0) : relu(x1[rows] . W1 + b1)
 *   out[rows] = add_act ? act(out[rows] + hid . W2 + b2) : hid . W2 + b2          (act = ReLU if relu_out)
 * weights_kmajor = 0: W1 is [HD][K1], W2 is [D2][HD] (torch Linear layout, forward);
 * weights_kmajor = 1: W1 is [K1][HD], W2 is [HD][D2] (the same parameters read transposed, backward).
 * hid_out (optional) receives the hidden rows.  Only K1=128, HD=256, D2=128 is fused: MMFT_ERR_UNSUPPORTED
 * otherwise (callers then use two mmft_linear_* launches). */
int mmft_mlp2_rows(const float* x1, long long ldx1, const int* rows, int n, const float* w1, long long ldw1,
                   const float* b1, const float* w2, long long ldw2, const float* b2, int weights_kmajor,
                   const float* mask, long long ldmask, float* hid_out, long long ldhid, float* out,
                   long long ldout, int add_act, int relu_out, int K1, int HD, int D2, const unsigned char* active,
                   int device, void* stream);
/* active (optional, may be NULL; also on mmft_pair_fwd_gather / mmft_level_bwd_pull): one flag per node marking the
 * transitive fan-in cone of the step's sampled endpoints (mmft_fanin_cone_step).  Rows outside it are skipped - a tile of
 * 32 such rows exits at once - and their outputs are left untouched; the reverse pull does not read cell consumers outside
 * the cone.  The caller zero-fills G / DA for such a step (rows outside the cone must read as zero gradients). */
/* MMFT_MATH_BF16 form of mmft_mlp2_rows with the weights PRE-PACKED as bf16 in [out][in] order (mmft_pack_bf16, once
 * per sweep): w1_bf16 [HD][K1], w2_bf16 [D2][HD].  A lane's MFMA weight fragment is then one 16-byte global load and the
 * weights never pass through LDS or a per-tile barrier - on the bf16 matrix pipe the arithmetic of a 32-row tile is 0.25 us
 * and the fp32 kernel's weight-panel staging is what remains.  The reverse form (weights_kmajor = 1 of mmft_mlp2_rows) is
 * this entry point fed with the transposed packs: w1_bf16 = bf16(W2g^T) [HD][D2], w2_bf16 = bf16(W1g^T) [K1][HD]. */
int mmft_pack_bf16(const float* src, long long ld, int R, int C, void* dst_bf16, int transpose, int device, void* stream);
/* hid_bf16 (this and the level kernels below): `mask` / `hid_out` - fc_cell_neigh's hidden activations HN and their gradients
 * DHN - are stored as bf16 (ld in elements) although the pointers are typed float*: their consumers round them to bf16 anyway
 * (weight-gradient MFMA operands, mmft_rows_outer_bf16 with g_bf16 / x_bf16) or look at the sign only (ReLU mask), so no result
 * changes and 2 KB of traffic per row and direction are saved. */
int mmft_mlp2_rows_bf16(const float* x1, long long ldx1, const int* rows, int n, const void* w1_bf16, const float* b1,
                        const void* w2_bf16, const float* b2, const float* mask, long long ldmask, float* hid_out,
                        long long ldhid, float* out, long long ldout, int add_act, int relu_out, int K1, int HD, int D2,
                        const unsigned char* active, int hid_bf16, int device, void* stream);
/* MMFT_MATH_BF16: the folded gather of one (net level l - 1, cell level l) pair (mmft_pair_fwd_gather) AND the cell level's
 * fc_cell_neigh MLP (mmft_mlp2_rows_bf16, forward form, add_act = 1) in ONE launch: h[net rows] = act(pre + mean h[driver]);
 * A / LSE of the cell rows; h[cell rows] = act(h[cell rows] + W2 relu(W1 A + b1) + b2); hid_out receives the hidden rows.
 * D = 128 only.  For levels without rows of very high fan-in (those keep the two-kernel form with its workgroup-per-row path). */
int mmft_level_fwd_bf16(float* h, const float* pre, long long ld, int D, const int* in_net_indptr, const int* in_net_indices,
                        const int* in_cell_indptr, const int* in_cell_indices, int net_row0, int n_net, const int* cell_rows,
                        int cell_row0, int n_cell, float* A, float* LSE, const void* w1_bf16, const float* b1,
                        const void* w2_bf16, const float* b2, float* hid_out, long long ldhid, int relu,
                        const unsigned char* active, const int* in_cell_driver, long long alg_bytes, int hid_bf16,
                        int device, void* stream);
/* The same launch with the per-edge index chain replaced by a static SLOT table (fan-in <= 4 on every cell row of the level,
 * contiguous row ranges): slots[v] = int[8] = the rows of h to read for v's four in-edges (< 0: no edge) and the rows of
 * `pre` to add (< 0: the edge's value is the h row itself; >= 0: relu(h[hrow] + pre[prow]), the net of level l - 1
 * recomputed from its single driver); net_driver[u] = the driver row of net u.  Workgroup b gathers + transforms cell rows
 * 16 b .. and writes net rows 16 b .. of level l - 1 (src/model.py:88-116,138-146). */
int mmft_level_fwd_slots(float* h, const float* pre, long long ld, int D, const int* slots, const int* net_driver, int net_row0,
                         int n_net, int cell_row0, int n_cell, float* A, float* LSE, const void* w1_bf16, const float* b1,
                         const void* w2_bf16, const float* b2, float* hid_out, long long ldhid, int relu,
                         const unsigned char* active, long long alg_bytes, int hid_bf16, int device, void* stream);
/* Reverse sweep of one (cell level l, net level l + 1) pair in ONE launch (three before: mmft_level_bwd_pull on each level and
 * mmft_mlp2_rows_bf16 in its reverse form) - the autograd mirror of graph.pull + the cell MLP, src/model.py:100-117,138-146,186-187:
 *   net rows w (sinks):   G[w] = relu'(h[w]) ((own[w] ? G[w] : 0) + sum_c DA[c] exp(h[w] - LSE[c]) (1 + h[w] - A[c]))
 *   cell rows v (drivers): G[v] = relu'(h[v]) ((own[v] ? G[v] : 0) + sum of G[w] over v's sinks);
 *                          DA[v] = ((G[v] W2g) * relu'(HN[v])) W1g, hidden gradient rows kept in hid_out (may be NULL).
 * Precondition, checked by the caller (PinGraph.level_bwd_pairs): level l is a contiguous id range whose out-net edges in CSR
 * order are exactly the id range of level l + 1 (sink of CSR position e = row e + sink_shift).  cslots int32[N][4] = the first
 * four cell consumers of a net row in out-edge order (-1 = none); a row with more has slot 3 = -2 - (out-cell CSR position of
 * its fourth consumer) and the kernel walks the CSR from there.  tiles int32[ntiles][8] = (first driver id, count <= 16, part,
 * parts, first scratch row, counter index, first and end out-net CSR position of the tile's sinks), one workgroup each: parts = 0
 * is a tile of whole drivers; a HEAVY driver (more sinks than a tile's target) is cut into `parts` tiles that leave their partial sums in scratch
 * (fp32 [rows][128]) and count themselves in counters[counter index] (int32, zero before the first call, left zero) - the
 * part that arrives last adds the partial sums in part order and finishes the driver (no waiting, no float atomics, the result
 * does not depend on the arrival order).  has_mlp = 0: the two pulls only (level 0).  Tiles of whole drivers add in the order
 * of mmft_level_bwd_pull without its heavy-row path: bitwise equal.  D = 128 only. */
int mmft_level_bwd_pair(float* G, const float* h, const float* A, const float* LSE, float* DA, long long ld, int D, int N,
                        const unsigned char* own_mask, const int* tiles, int ntiles, const int* out_net_indptr, int sink_shift,
                        const int* cslots, const int* out_cell_indptr, const int* out_cell_indices, float* scratch, int* counters,
                        int relu, int has_mlp, const void* w1_bf16, const void* w2_bf16, const float* mask, long long ldmask,
                        float* hid_out, long long ldhid, long long alg_bytes, int hid_bf16, int device,
                        void* stream);
/* MMFT_MATH_BF16: the feature MLPs fc_cell_self / fc_net_self (Linear(fin, 256)-ReLU-Linear(256, 128) over the contiguous node
 * rows row0 .. row0 + n - 1; src/model.py:48-51,66-67,148-153,186-189) WITHOUT a stored hidden tensor.
 *   fwd: out[row] = (relu)(W2 relu(W1 x[row] + b1) + b2); x [.][ldx >= fin] and out [.][ldout] are indexed by NODE id.
 *   bwd: from g [.][ldg] (gradient of the output rows) and x: dw1 [256][fin], db1 [256], dw2 [128][256], db2 [128], stored
 *        (accumulate = 0) or added (1); the hidden activations are recomputed with the forward's instruction sequence.
 * fin <= 64; w1 [256][fin], w2 [128][256] plain fp32 row-major (rounded to bf16 in the kernel). */
int mmft_mlp2_feat_fwd_bf16(const float* x, long long ldx, int row0, int n, int fin, const float* w1, const float* b1,
                            const float* w2, const float* b2, float* out, long long ldout, int relu_out, int device,
                            void* stream);
long long mmft_mlp2_feat_bwd_workspace_bytes(int n, int fin);
int mmft_mlp2_feat_bwd_bf16(const float* g, long long ldg, const float* x, long long ldx, int row0, int n, int fin,
                            const float* w1, const float* b1, const float* w2, float* dw1, float* db1, float* dw2, float* db2,
                            int accumulate, float* workspace, long long workspace_bytes, int device, void* stream);
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
 * source (col = destination).  `rows` lists the node ids of the current topological level; rows == NULL means
 * the contiguous id range row0 .. row0+n-1 (levels are contiguous after level-major renumbering, which removes one
 * dependent load from every thread's chain).  alg_bytes (optional, 0 = unknown) is the algorithmic HBM byte count of
 * the launch, recorded with it by the launch profiler for bench.py's roofline.
 * ------------------------------------------------------------------------------------------- */
/* A[v][c] = sum_i softmax_i(h[u_i][c]) h[u_i][c];  LSE[v][c] = log sum_i exp(h[u_i][c])  (deg 0: A=0) */
int mmft_seg_softmax_sum_fwd(const float* h, long long ldh, const int* in_indptr, const int* in_indices,
                             const int* rows, int row0, int n, int D, float* A, float* LSE, long long lda,
                             long long alg_bytes, int device, void* stream);
/* h[v] = act(h[v] + mean_{u->v} h[u])   (h[v] holds fc_net_self(x_net[v]) on entry; 0 for degree 0) */
int mmft_seg_mean_add_act_fwd(float* h, long long ldh, const int* in_indptr, const int* in_indices,
                              const int* rows, int row0, int n, int D, int relu, long long alg_bytes,
                              int device, void* stream);
/* out[i] (+)= sum over the CSR segment of rows[i] of src[u]  (deterministic segmented row sum; with
 * accumulate != 0 the result is added to out[rows[i]] in place) */
/* out[v] (+)= sum of the rows e of src with sorted_keys[e] == v, v < R; sorted_keys ascending (the gradient of
 * mlp_alpha's level table gathered per endpoint in level order, src/model.py:280): segments found by binary search, rows
 * added in a fixed order (bitwise reproducible). */
int mmft_seg_sum_sorted(const float* src, long long lds, const int* sorted_keys, int nsrc, int R, int D, float* out, long long ldo,
                        int accumulate, int device, void* stream);
int mmft_seg_sum_fwd(const float* src, long long lds, const int* indptr, const int* indices,
                     const int* rows, int n, int D, float* out, long long ldo, int accumulate,
                     int device, void* stream);
/* the same sum with one WORKGROUP per row (thread groups stride over the segment, partials combined in a fixed order):
 * for few, long segments - the gradient of PathModel.mlp_alpha's level table (src/model.py:267,280: one row per level,
 * ~T/L endpoints each).  out[rows[i]] (rows == NULL: out[i]) */
int mmft_seg_sum_rows_wg(const float* src, long long lds, const int* indptr, const int* indices, const int* rows, int n,
                         int D, float* out, long long ldo, int accumulate, int device, void* stream);
/* out[v] = mean_{u->v} src[u]  (standalone fn.mean) */
int mmft_seg_mean_fwd(const float* src, long long lds, const int* in_indptr, const int* in_indices,
                      const int* rows, int n, int D, float* out, long long ldo, int device, void* stream);
/* Attention branch of PathConv (flag_attn = True): message_func_attn + cell_msg_reduce_attn (src/model.py:119-136,
 * 190-196).  fc_key = Linear(1, dk, no bias), fc_attn = Linear(2 dk, 1, no bias), so the edge score is
 *   e(u -> v) = leaky_relu(c12[0] * key[u] + c12[1] * key[v], slope),  c12[0] = <fc_attn.w[:dk], fc_key.w>, c12[1] = <fc_attn.w[dk:], fc_key.w>
 * (c12: two floats in DEVICE memory, computed by the caller with mmft_linear_fwd so that autograd reaches both weights).
 *   alpha = softmax of e over the in-edges of v;  A[v] = sum alpha_i h[u_i]  (0 for degree 0);  alpha is stored per
 *   in-edge (CSR position) for the reverse sweep. */
int mmft_seg_attn_fwd(const float* h, long long ldh, const float* key, const float* c12, float slope, const int* in_indptr,
                      const int* in_indices, const int* rows, int row0, int n, int D, float* A, long long lda, float* alpha,
                      int device, void* stream);
/* reverse pull of the attention branch: as mmft_level_bwd_pull with the cell term sum_{e: v->w} alpha[out2in_cell[e]] * DA[w]
 * (out2in_cell maps an out-CSR edge position to the in-CSR position of the same edge) */
int mmft_level_bwd_pull_attn(float* G, const float* h, long long ld, const int* rows, int row0, int n, int D,
                             const int* out_net_indptr, const int* out_net_indices, const float* out_net_weight,
                             const int* out_cell_indptr, const int* out_cell_indices, const int* out2in_cell,
                             const float* alpha, const float* DA, int relu, const unsigned char* own_mask, int device,
                             void* stream);
/* gradient of the edge scores of the cell rows v: dcp[v][0..1] = sum_i da_i * (key[u_i], key[v]) with
 * de_i = alpha_i (<DA[v], h[u_i]> - <DA[v], A[v]>), da_i = de_i * leaky_relu'(pre_i); the caller sums dcp over all rows
 * (mmft_colsum) to obtain d loss / d c12.  D / 4 must be a power of two <= 64. */
int mmft_seg_attn_bwd_scores(const float* DA, const float* h, const float* A, long long ld, const float* alpha,
                             const float* key, const float* c12, float slope, const int* in_indptr, const int* in_indices,
                             const int* rows, int row0, int n, int D, float* dcp, int device, void* stream);
/* out[rows[i] (scatter) or i][0..D) = mean over the CSR segment of row rows[i] of src, any D: ndata['h_drive'] of the
 * attention branch (fn.copy_src('net_feat') + fn.mean over the net in-edges, src/model.py:197-198,66-86) */
int mmft_seg_mean_rows_any(const float* src, long long lds, const int* indptr, const int* indices, const int* rows, int n,
                           int D, float* out, long long ldo, int scatter, int device, void* stream);
/* Reverse sweep, one level (deterministic pull over out-edges, no atomics); out_net_weight[e] = 1/indeg_net of
 * the destination of out-edge e (static per graph, aligned with out_net_indices):
 *   gh = G[v] + sum_{e: v->w in out_net(v)} G[w] * out_net_weight[e]
 *             + sum_{w in out_cell(v)} DA[w] * exp(h[v]-LSE[w]) * (1 + h[v] - A[w])
 *   G[v] = relu ? (h[v] > 0 ? gh : 0) : gh
 * G rows of later levels hold d(loss)/d(pre-activation), DA rows d(loss)/d(A). */
int mmft_level_bwd_pull(float* G, const float* h, long long ld, const int* rows, int row0, int n, int D,
                        const int* out_net_indptr, const int* out_net_indices, const float* out_net_weight,
                        const int* out_cell_indptr, const int* out_cell_indices,
                        const float* A, const float* LSE, const float* DA, int relu, const unsigned char* own_mask,
                        const int* heavy_rows, int nheavy, int heavy_thresh, const unsigned char* active,
                        long long alg_bytes, int device, void* stream);
/* heavy_rows (optional, nheavy of them): exactly the rows of this level whose out-degree (net + cell) exceeds
 * heavy_thresh.  Each is reduced by a whole workgroup - eight thread groups stride over its out-edges, partial sums
 * combined through LDS in a fixed order (deterministic) - instead of one 32-lane group walking hundreds of dependent
 * row loads in series (drivers of clock / reset-like nets; config E's Zipf skew). */
/* Folded forward sweep: one gather launch per (net level l - 1, cell level l) pair (src/model.py:185-204).
 *   net rows u = net_row0 .. net_row0 + n_net - 1:  h[u] = act(pre[u] + mean_{d->u} h[d])     (pre = fc_net_self(x_net))
 *   cell rows v (cell_rows[i] or cell_row0 + i):     A[v], LSE[v] as mmft_seg_softmax_sum_fwd, where in-neighbours inside
 *   the net range are recomputed from pre and their driver rows (bitwise the value stored by the first part) instead of
 *   being read from h - so the two levels need no launch boundary between them.  n_cell = 0 gives a plain net level.
 * heavy_rows (optional): exactly the cell rows with more than heavy_thresh in-edges; each is reduced by a whole workgroup
 * (partial online softmaxes merged in a fixed order), the others by one 32-lane thread group walking the edges in series.
 * in_cell_driver (optional, one int per cell in-edge, aligned with in_cell_indices): the single driver of the net behind
 *   the edge (in_net_indices[in_net_indptr[u]] where that net has exactly one in-edge), or a negative value to resolve the
 *   net through its CSR.  With it the dependent-load chain of an edge is three deep instead of five and four edges are
 *   requested together; results are bitwise those of the serial form. */
int mmft_pair_fwd_gather(float* h, const float* pre, long long ld, int D, const int* in_net_indptr,
                         const int* in_net_indices, const int* in_cell_indptr, const int* in_cell_indices, int net_row0,
                         int n_net, const int* cell_rows, int cell_row0, int n_cell, float* A, float* LSE, long long lda,
                         int relu, const int* heavy_rows, int nheavy, int heavy_thresh, const unsigned char* active,
                         const int* in_cell_driver, long long alg_bytes, int device, void* stream);
/* own_mask (may be NULL): per-node flag telling whether G[v] already holds a gradient of its own (a sampled endpoint,
 * src/model.py:213); rows without the flag start from zero, so G needs no 4*N*D-byte fill per step.
 * mmft_target_rows_begin zeroes the G rows of the endpoints idx[0..n) and sets their flags (the scatter-add of the
 * endpoint gradients follows), mmft_target_rows_end clears the flags after the reverse sweep. */
int mmft_target_rows_begin(float* G, long long ld, const int* idx, int n, int D, unsigned char* flags, int device,
                           void* stream);
int mmft_target_rows_end(const int* idx, int n, unsigned char* flags, int device, void* stream);
/* flags[idx[i]] = value for i < n (seeds the fan-in-cone mask with the step's endpoints) */
int mmft_mark_rows(const int* idx, int n, unsigned char* flags, int value, int device, void* stream);
/* dst[i] = src[idx[i]]  /  dst[idx[i]] += src[i] (atomic, duplicates allowed: src/train.py:377-380) */
int mmft_gather_rows(const float* src, long long lds, const int* idx, int n, int D, float* dst, long long ldd,
                     int device, void* stream);
int mmft_scatter_add_rows(float* dst, long long ldd, const int* idx, int n, int D, const float* src,
                          long long lds, int device, void* stream);
/* the same sum without atomics: `order` is a STABLE argsort of idx (order[p] = batch row at sorted position p); each
 * destination adds its rows in batch order, so endpoints duplicated by oversampling (src/train.py:377-380) give
 * bitwise reproducible gradients */
int mmft_scatter_add_rows_sorted(float* dst, long long ldd, const int* idx, const int* order, int n, int D,
                                 const float* src, long long lds, int device, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Layout-image CNN  --  replaces nn.Conv2d / BatchNorm2d(train) / ReLU / MaxPool2d / AvgPool2d /
 * ConvTranspose2d / F.pad / torch.cat inside DoubleConv, Down, Up, OutConv, UNet
 * (src/Unet.py:8-119) and LayoutNet (src/model.py:216-247).
 * Activations are NHWC fp32 ([Nimg][H][W][C], i.e. torch channels_last); conv weights are
 * [Co][KH][KW][Ci] (the channels_last memory of an OIHW parameter).  Stride 1, 2*pad == K-1.
 * ------------------------------------------------------------------------------------------- */
/* y = act(conv(x, w) + bias)  (bias may be NULL: src/Unet.py:16,19 use bias=False) */
int mmft_conv2d_fwd(const float* x, const float* w, const float* bias, float* y, int Nimg, int H, int W,
                    int Ci, int Co, int KH, int KW, int pad, int act, float slope, int device, void* stream);
/* dx = conv(dy, flip(w)^T); workspace holds the re-laid-out weights (>= Co*KH*KW*Ci*4 bytes) */
int mmft_conv2d_dgrad(const float* dy, const float* w, float* dx, int Nimg, int H, int W, int Ci, int Co,
                      int KH, int KW, int pad, float* workspace, long long workspace_bytes,
                      int device, void* stream);
/* dw[co][kh][kw][ci] = sum_pixels dy[p][co] * x[p+tap][ci]; deterministic split over pixels */
long long mmft_conv2d_wgrad_workspace_bytes(int Nimg, int H, int W, int Ci, int Co, int KH, int KW);
int mmft_conv2d_wgrad(const float* x, const float* dy, float* dw, int Nimg, int H, int W, int Ci, int Co,
                      int KH, int KW, int pad, float* workspace, long long workspace_bytes,
                      int device, void* stream);
/* BatchNorm2d in TRAIN mode (the reference never calls eval()) + optional ReLU.
 * x is [groups][rows][C]; statistics are taken per group: groups=1, rows=N*H*W is nn.BatchNorm2d;
 * groups=N, rows=H*W gives per-sample statistics (N=1 semantics when several designs are batched),
 * running stats then receive `groups` sequential momentum updates.  save_mean/save_invstd: [groups][C]. */
long long mmft_bn_workspace_bytes(int groups, long long rows, int C);
int mmft_bn_train_fwd(const float* x, float* y, const float* gamma, const float* beta, float* running_mean,
                      float* running_var, float momentum, float eps, int groups, long long rows, int C,
                      float* save_mean, float* save_invstd, int relu, float* workspace,
                      long long workspace_bytes, int device, void* stream);
/* g = gy * (y > 0 if relu); dgamma = sum g*xhat; dbeta = sum g;
 * dx = gamma*invstd*(g - mean(g) - xhat*mean(g*xhat)) with means per group.  With `beta` given the ReLU mask is
 * recomputed from x by the forward's own affine (same rounding sequence) and y may be NULL: 5 tensor passes, not 7 */
int mmft_bn_train_bwd(const float* gy, const float* x, const float* y, const float* gamma, const float* beta,
                      const float* save_mean, const float* save_invstd, float* dx, float* dgamma,
                      float* dbeta, int groups, long long rows, int C, int relu, float* workspace,
                      long long workspace_bytes, int device, void* stream);
/* 2x2 stride-2 pooling, floor mode (MaxPool2d(2) / AvgPool2d(2)); backward recomputes the argmax
 * (first maximum in row-major window order, as torch) */
int mmft_pool2x2_fwd(const float* x, float* y, int Nimg, int H, int W, int C, int mode, int device, void* stream);
int mmft_pool2x2_bwd(const float* x, const float* gy, float* dx, int Nimg, int H, int W, int C, int mode,
                     int device, void* stream);
/* nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True) of Up(bilinear=True) (src/Unet.py:48-51): NHWC,
 * (N,H,W,C) -> (N,2H,2W,C), ATen's source-index arithmetic; the backward is a deterministic gather (no atomics) */
int mmft_upsample_bilinear2x_fwd(const float* x, float* y, int Nimg, int H, int W, int C, int device, void* stream);
int mmft_upsample_bilinear2x_bwd(const float* gy, float* dx, int Nimg, int H, int W, int C, int device, void* stream);
/* ConvTranspose2d(k=2,s=2) = GEMM [pixels x Ci]*[Ci x 4Co] + pixel shuffle (src/Unet.py:53):
 * shuffle:   out[n][2y+a][2x+b][co] = in[n][y][x][(a*2+b)*Co+co] + bias[co]
 * unshuffle: out[n][y][x][(a*2+b)*Co+co] = in[n][2y+a][2x+b][co] */
int mmft_pixel_shuffle2(const float* in, const float* bias, float* out, int Nimg, int H, int W, int Co,
                        int device, void* stream);
int mmft_pixel_unshuffle2(const float* in, float* out, int Nimg, int H, int W, int Co, int device, void* stream);
/* the same two moves with the big tensor being a channel slice [c_off, c_off + Co) of an NHWC tensor with ldc channels:
 * the up-sampled half of torch.cat([x2, x1], dim=1) (src/Unet.py:67) is written / read in place, no copy */
int mmft_pixel_shuffle2_into(const float* in, const float* bias, float* out, int Nimg, int H, int W, int Co, int ldc,
                             int c_off, int device, void* stream);
int mmft_pixel_unshuffle2_from(const float* in, float* out, int Nimg, int H, int W, int Co, int ldc, int c_off, int device,
                               void* stream);
/* dst[n][y+y_off][x+x_off][c_off : c_off+Cs] = src[n][y][x][:]   (torch.cat + F.pad, src/Unet.py:59-67);
 * reverse=1 copies the same region from dst back into src (backward of cat/pad) */
int mmft_copy_region_nhwc(float* src, int Nimg, int Hs, int Ws, int Cs, float* dst, int Hd, int Wd, int Cd,
                          int c_off, int y_off, int x_off, int reverse, int device, void* stream);
/* NCHW <-> NHWC with channel padding: dst[n][h][w][c] = c < C ? src[n][c][h][w] : 0, c < Cpad */
int mmft_nchw_to_nhwc(const float* src, float* dst, int Nimg, int C, int H, int W, int Cpad, int device, void* stream);
int mmft_nhwc_to_nchw(const float* src, float* dst, int Nimg, int C, int H, int W, int Cpad, int device, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Fusion head / step glue  --  replaces `path_mask.to_dense()*feat_map` + fcn (src/train.py:500-501,
 * src/model.py:271-272) without materialising the dense T x P map, nn.MSELoss (src/train.py:32,522)
 * and th.optim.Adam (src/train.py:431-435,555).
 * Masks are CSR over path ids: mask_indptr[num_paths+1], mask_cols[nnz] (column = x*map + y,
 * src/verilog_parser_asap7.py:1332-1368); `paths[T]` selects the batch rows (duplicates allowed).
 * ------------------------------------------------------------------------------------------- */
/* dst[c][r] = src[r][c]   (fcn.weight [Dout][P] <-> [P][Dout]) */
int mmft_transpose(const float* src, float* dst, int R, int C, int device, void* stream);
/* Several designs may share one call: f is [B][P] (one feature map per design) and f_off[t] (NULL = 0)
 * is the offset b*P of the design that batch row t belongs to.
 * out[t][:] = bias + sum_{p in mask(paths[t])} f[f_off[t]+p] * wT[p][:]      wT is [P][Dout], Dout % 4 == 0 */
int mmft_masked_fc_fwd(const int* mask_indptr, const int* mask_cols, const int* paths, const int* f_off, int T,
                       const float* f, const float* wT, const float* bias, float* out, int P, int Dout,
                       int device, void* stream);
/* The same projection when the masks are given as runs of consecutive cells (path masks are unions of boxes,
 * src/verilog_parser_asap7.py:1326-1334): GP[b*P + c] = sum of f * wT over the cells <= c of c's block of S cells
 * (prefix), then out[t] = bias + sum over the runs [s, e] of path t of GP[e] - GP[s - 1] (runs never cross a block). */
int mmft_masked_fc_prefix(const float* f, const float* wT, float* GP, int B, int P, int Dout, int S, int device,
                          void* stream);
int mmft_masked_fc_fwd_runs(const int* run_ptr, const int* run_start, const int* run_len, const int* paths,
                            const int* f_off, int T, const float* GP, const float* bias, float* out, int Dout, int S,
                            int device, void* stream);
/* Backward of the run form: bnd_ptr[B*P + 1] / bnd_code list, per cell (design-major), the paths with a run ending at
 * the cell (code = path id) or starting right after it inside the same block (code = -path id - 1); one workgroup per
 * (design, block) gathers the signed sums of gout into LDS, turns them into dg by a suffix scan and emits dwT
 * (per-design slabs summed in fixed order) and df.  Dout / 4 must be a power of two <= 64, S * Dout * 4 <= 64 KB. */
long long mmft_masked_fc_bwd_runs_workspace_bytes(int B, int P, int Dout);
int mmft_masked_fc_bwd_runs(const int* bnd_ptr, const int* bnd_code, const int* first, const int* next, const float* gout,
                            long long ldg, const float* f, const float* wT, float* dwT, float* df, int B, int P, int Dout, int S,
                            float* workspace, long long workspace_bytes, int device, void* stream);
/* Backward of the masked projection as a deterministic GATHER over the transposed masks (no atomics):
 * csc_indptr[B*P+1] / csc_paths[nnz] list, for every map cell (b, p), the path ids whose mask covers it
 * (ascending); first[q] is the first batch row holding path q (-1: not sampled) and next[t] the next batch
 * row with the same path (-1: none), so duplicates from oversampling are summed in batch order.
 *   S = sum_{t : p in mask(paths[t]), design(t) = b} gout[t][:]
 *   dwT[p][:] = sum_b f[b][p] * S      (transposed weight gradient, [P][Dout])
 *   df[b][p]  = sum_c wT[p][c] * S[c]
 * For B > 1 the designs run in parallel block columns; workspace >= B*P*Dout*4 bytes holds their dwT slabs. */
int mmft_masked_fc_bwd(const int* csc_indptr, const int* csc_paths, const int* first, const int* next,
                       const float* gout, long long ldg, const float* f, const float* wT, float* dwT, float* df,
                       int B, int P, int Dout, float* workspace, long long workspace_bytes,
                       int device, void* stream);
/* loss = mean((pred-target)^2); grad[i] = 2*(pred[i]-target[i])/n   (single workgroup, n <= 2^24) */
int mmft_mse_fwd_bwd(const float* pred, const float* target, int n, float* loss, float* grad,
                     int device, void* stream);
/* the same with target[i] = table[idx[i] * ld]: the labels are (N, 1) node tensors indexed by the endpoints
 * (arrival_time[target_list], src/train.py:519-522) */
int mmft_mse_gather_fwd_bwd(const float* pred, const float* table, long long ld, const int* idx, int n, float* loss, float* grad,
                            int device, void* stream);
/* nn.CrossEntropyLoss() (mean) of the classification task (--task cls, nlabels classes; src/train.py:32,516-518) over
 * logits [n][C] and int64 labels: loss[0] = mean_t (logsumexp z_t - z_t[y_t]), grad[t][c] = (softmax - onehot) / n (grad may
 * be NULL).  eval_out (optional fp64[6]) = n, sum of losses, tp, fp, tn, fn with predicted class = argmax (first maximum)
 * and positive = class != 0 (src/train.py:516,536-541).  Labels must lie in [0, C).  Single workgroup, n <= 2^24. */
int mmft_cross_entropy_fwd_bwd(const float* logits, const long long* labels, int n, int C, float* loss, float* grad,
                               double* eval_out, int device, void* stream);
/* Evaluation sums of validate() / test() (src/train.py:230-278, src/test.py:211-216,231-300) in one pass, fp64:
 * out[0..9] = n, sum y, sum y^2, sum (p-y)^2, sum |p-y|, sum |p-y|/|y| (MAPE numerator, y != 0),
 *             tp, fp, tn, fn   with predicted critical = (required - p) < 0 (judge_critical, src/train.py:391-395)
 *             and actual critical = label != 0.   Single workgroup, n <= 2^24. */
int mmft_eval_sums(const float* pred, const float* arrival, const float* required, const float* label,
                   int n, double* out, int device, void* stream);
/* the same ten sums PER TOPOLOGICAL LEVEL: out[l][0..9] over the batch rows with level[i] == l, l < num_levels
 * (src/test.py:211-216 prints R2 and MAPE of every level) */
int mmft_eval_sums_by_level(const float* pred, const float* arrival, const float* required, const float* label,
                            const int* level, int n, int num_levels, double* out, int device, void* stream);
/* one Adam step over flat buffers, same operation order as torch.optim.Adam (amsgrad=False):
 * g += wd*p; m = lerp(m, g, 1-b1); v = b2*v + (1-b2)*g*g; p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
 * gscale multiplies the gradient first (1/world_size after a sum all-reduce) */
int mmft_adam_step(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1,
                   float beta2, float eps, float weight_decay, float bias_correction1,
                   float bias_correction2, float gscale, int device, void* stream);
/* same, with the step-dependent scalars read from DEVICE memory (step_scalars[0] = lr/bias_correction1,
 * step_scalars[1] = sqrt(bias_correction2)) so that the launch can be captured once in a HIP graph and
 * replayed every step */
int mmft_adam_step_dev(float* p, const float* g, float* m, float* v, long long n, const float* step_scalars,
                       float beta1, float beta2, float eps, float weight_decay, float gscale,
                       int device, void* stream);

/* same, with the optimizer's step counter kept in DEVICE memory: state[0] = steps taken so far (advanced by the
 * launch), state[1] = 0 (scratch ticket).  Both bias corrections are derived in the kernel from state[0] + 1, so the
 * host uploads nothing per step and the launch can be captured in a HIP graph and replayed any number of times, with
 * the host running ahead of the device.  state is two int32 words, zeroed by the caller before the first step.
 * zero_grad != 0: g is zeroed by the same pass (optimizer.zero_grad() of src/train.py:552 without a fill launch). */
int mmft_adam_step_counted(float* p, float* g, float* m, float* v, long long n, int* state, float lr, float beta1,
                           float beta2, float eps, float weight_decay, float gscale, int zero_grad, int device, void* stream);

/* ---- design preprocessing: the graph-side steps the reference runs on networkx in Python ---- */
/* Longest-path levels from the primary inputs `pis` (src/verilog_parser_asap7.py:1452-1517, cal_topo_level: frontier
 * expansion + reverse de-duplication = every node keeps the LAST level it appears in).  Up to two out-edge CSRs
 * (net, cell; second may be NULL).  level[v] = -1 for nodes no PI reaches; *num_levels is a HOST int.  Synchronises
 * the stream between chunks of steps (preprocessing, not the training step). */
long long mmft_levelize_workspace_bytes(int n);
int mmft_levelize(const int* out_indptr0, const int* out_indices0, const int* out_indptr1, const int* out_indices1,
                  int n, const int* pis, int npi, int* level, int* num_levels, void* workspace,
                  long long workspace_bytes, int device, void* stream);
/* One level of the fan-in-cone closure (the reference carries the idea unused): every node of
 * rows[0..n) (or row0 + i) whose mark is set marks all its in-neighbours.  Called for levels L-1 .. 1 after the sampled
 * endpoints were marked, it leaves mark = 1 exactly on the nodes that can influence those endpoints. */
int mmft_fanin_cone_step(const int* rows, int row0, int n, const int* in_indptr0, const int* in_indices0,
                         const int* in_indptr1, const int* in_indices1, unsigned char* mark, int device, void* stream);
/* One critical path per endpoint (src/verilog_parser_asap7.py:1433-1450, find_critical_path): walk back while the
 * level is >= 2, each time to the FIRST in-neighbour (CSR 0 then CSR 1, insertion order) exactly one level below;
 * an in-neighbour flagged in `stop` (the reference's 'clk' name test; may be NULL) met first ends the walk.
 * paths[npaths][maxlen] padded with -1; lens[i] > maxlen means row i was truncated. */
int mmft_trace_critical_paths(const int* in_indptr0, const int* in_indices0, const int* in_indptr1,
                              const int* in_indices1, const int* level, const unsigned char* stop,
                              const int* endpoints, int npaths, int maxlen, int* paths, int* lens, int device,
                              void* stream);
/* Path masks (src/verilog_parser_asap7.py:1302-1369, masking == 'critical'): row i = union over consecutive pins of
 * path i of the bounding box of their map locations, column = x * map_y + y, as CSR with ascending columns.
 * Two passes: count (-> counts[npaths]; the caller scans them into indptr), then fill. */
int mmft_path_mask_count(const int* paths, const int* lens, int npaths, int maxlen, const int* loc_x, const int* loc_y,
                         int map_x, int map_y, int* counts, int device, void* stream);
int mmft_path_mask_fill(const int* paths, const int* lens, int npaths, int maxlen, const int* loc_x, const int* loc_y,
                        int map_x, int map_y, const int* indptr, int* cols, int device, void* stream);
/* In-place per-column min-max scaling of columns [start_col, C) (src/train.py:309-318): (a - min) / (max - min),
 * same two roundings; NaN where max == min or the column holds a NaN, as torch. */
long long mmft_minmax_workspace_bytes(int n, int ncol);
int mmft_minmax_normalize(float* feat, long long ld, int n, int C, int start_col, float* workspace,
                          long long workspace_bytes, int device, void* stream);

/* ---- OutConv of the layout U-Net, fused (src/Unet.py:71-82): 1x1 convolution to one channel + bias, 2x2 pooling
 * (mode: MMFT_POOL_MAX / MMFT_POOL_AVG), ReLU.  x [N][H][W][Ci] (NHWC), w [Ci], bias [1] or NULL, out / gout [N][H/2][W/2].
 * Supported: Ci in {16, 32}, H even, W % 32 == 0 (mmft_outconv_supported); other shapes use mmft_conv2d_* + mmft_pool2x2_*
 * + mmft_act_*.  The backward recomputes the pixel values from x (nothing is saved by the forward), writes every element
 * of dx, and (accumulate = 0) stores or (1) adds dw [Ci] and db [1] (db may be NULL). */
int mmft_outconv_supported(int H, int W, int Ci);
int mmft_outconv_fwd(const float* x, const float* w, const float* bias, float* out, int Nimg, int H, int W, int Ci, int mode,
                     int device, void* stream);
long long mmft_outconv_bwd_workspace_bytes(int Nimg, int H, int W, int Ci);
int mmft_outconv_bwd(const float* x, const float* w, const float* bias, const float* gout, float* dx, float* dw, float* db,
                     int accumulate, int Nimg, int H, int W, int Ci, int mode, float* workspace, long long workspace_bytes,
                     int device, void* stream);

/* Predictions of ONE per-level call of PathModel.forward (src/model.py:269-292; the loop of src/train.py:490-511) in one entry
 * point: out[t] = mlp_fuse([h[targets[t]] | fcn(masked path map of path t) | mlp_alpha(level)]) - gather, masked projection
 * (run form over the prefix table GP of mmft_masked_fc_prefix), level embedding, Linear(Dh + Dc + Da, H1) - ReLU -
 * Linear(H1, nout).  No state is kept for a backward pass. */
long long mmft_head_level_workspace_bytes(int T, int Dh, int Dc, int Da, int H1);
int mmft_head_level_fwd(const float* h, long long ldh, const int* targets, int T, int Dh, const int* run_ptr, const int* run_start,
                        const int* run_len, const int* paths, const int* f_off, const float* GP, const float* fcn_bias, int Dc,
                        int S, const float* alpha_row, int Da, const float* w1, const float* b1, int H1, const float* w2,
                        const float* b2, int nout, float* workspace, long long workspace_bytes, float* out, int device, void* stream);

/* out[t] = [a[t] | b[t] | c[t]] (row-wise concatenation of up to three blocks, c may be NULL): torch.cat((h_gnn, h_cnn,
 * h_global), 1) of PathModel.forward (src/model.py:285-290) */
int mmft_concat_cols(const float* a, long long lda, int Da, const float* b, long long ldb, int Db, const float* c, long long ldc,
                     int Dc, float* out, long long ldo, int T, int device, void* stream);

/* MMFT_MATH_BF16: dw [out][in] (+)= g^T x over `rows` rows and db [out] (+)= column sums of g (db may be NULL) for the two
 * weight gradients of fc_cell_neigh (Linear(128, 256) / Linear(256, 128) over every cell node of the batch,
 * src/model.py:48-51): (out, in) in {(128, 256), (256, 128)}, g [rows][ldg], x [rows][ldx] fp32 (rounded to bf16 at
 * staging, fp32 accumulation); the row-major operands are transposed by the LDS hardware on the way to the MFMA. */
int mmft_rows_outer_supported(int out, int in);
long long mmft_rows_outer_workspace_bytes(long long rows, int out, int in);
int mmft_rows_outer_bf16(const float* g, long long ldg, const float* x, long long ldx, float* dw, float* db, long long rows, int out,
                         int in, int accumulate, float* workspace, long long workspace_bytes, int g_bf16, int x_bf16, int device, void* stream);

/* ---- bf16-STORAGE layout U-Net (bf16 math mode; BASELINE config B "bf16 storage / fp32 accumulate") --------------------
 * The same layers as above - DoubleConv / Down / Up / OutConv of src/Unet.py:8-82 - with every activation, pre-activation
 * and activation gradient kept in HBM as bf16 (NHWC, `const void*` = unsigned short), statistics / parameters / parameter
 * gradients fp32, fp32 accumulation on the MFMA.  Host mirror: mmft/unet16.py (one autograd node for UNet.forward,
 * src/Unet.py:110-119).
 *
 * mmft_u16_pack_weights: all weights of a step re-packed by ONE launch into MFMA A-fragment order (bf16).  descs: device
 * array of n records {const float* w; unsigned short* out; int rows, K, taps, mode, Rsrc, Ksrc} (mmft_u16_pack_desc_bytes
 * each): mode 0 = w[(row * taps + t) * Ksrc + k] (Conv2d weight [Co][3][3][Ci], src/Unet.py:16,19; ConvTranspose2d matrix
 * [(a,b,co)][ci], src/Unet.py:53), 1 = flipped taps / transposed channels (input gradient), 2 = transposed matrix; + 4 = fragments of the 16x16x32
 * MFMA (8 consecutive k per lane) - what mmft_u16_conv3x3 expects when its reduction channel count is >= 32. */
int mmft_u16_pack_desc_bytes(void);
int mmft_u16_pack_weights(const void* descs, int n, long long max_frag_lanes, long long* counters, int ncounters, long long inc,
                          int device, void* stream);
/* Conv2d(k=3, padding=1, bias=False) (src/Unet.py:16,19), Ci, Co in {16,32,64,128} or the 3 -> 16 RGB layer (x fp32
 * [N][H][W][3]); the input gradient is the same call on the gradient with the mode-1 pack.  stats != NULL: per tile
 * [2][Co] = sum, sum of squares of the stored (rounded) outputs - the BatchNorm statistics of src/Unet.py:17,20.
 * mmft_u16_conv_tiles: rows of `stats`. */
int mmft_u16_conv_tiles(int N, int H, int W, int* per_image);
int mmft_u16_conv3x3(const void* x, int rgb_f32, const void* wpk, void* y, float* stats, int N, int H, int W, int Ci, int Co,
                     int device, void* stream);
/* Weight gradients: every workgroup leaves one partial result (a SLAB) in `workspace`; with dw != NULL the call also adds
 * the slabs up (in a fixed order) into dw, with dw == NULL they stay in `workspace` and the caller reduces several layers'
 * slabs with ONE mmft_slab_reduce_batch launch (nothing needs a U-Net weight gradient before the optimizer).
 * *_slabs: the number of slabs; conv3x3: Co * 9 * Ci floats each. */
long long mmft_u16_conv3x3_wgrad_workspace_bytes(int N, int H, int W, int Ci, int Co);
int mmft_u16_conv3x3_wgrad_slabs(int N, int H, int W, int Ci, int Co);
int mmft_u16_conv3x3_wgrad(const void* x, int rgb_f32, const void* dy, float* dw, int accumulate, int N, int H, int W, int Ci,
                           int Co, float* workspace, long long workspace_bytes, int device, void* stream);
/* BatchNorm2d in train mode, per-image statistics (src/Unet.py:17,20; one image per call in src/train.py:465) + ReLU
 * (:18,21).  finalize: tile partials -> bnp[5][N][C] = mean, invstd, biased variance, scale, shift.  apply:
 * a = relu(z * scale + shift) written with pixel pitch lda (a channel slice of the concatenation of src/Unet.py:67),
 * pooled != NULL: the 2x2 pooling of src/Unet.py:33-36 in the same pass; running_mean != NULL: the N sequential momentum
 * updates of the running statistics (one per image).  bwd: g -> dz, dgamma, dbeta. */
int mmft_u16_bn_finalize(const float* stats, int tiles_per_image, int N, int C, long long pixels_per_image, float eps,
                         const float* gamma, const float* beta, float* bnp, int device, void* stream);
int mmft_u16_bn_apply(const void* z, const float* bnp, void* a, int lda, void* pooled, int N, int H, int W, int C, int pool_mode,
                      float momentum, float* running_mean, float* running_var, int device, void* stream);
long long mmft_u16_bn_bwd_workspace_bytes(int N, long long pixels_per_image, int C);
int mmft_u16_bn_bwd(const void* g, const void* z, const float* bnp, void* dz, float* dgamma, float* dbeta, int accumulate, int N,
                    long long pixels_per_image, int C, float* workspace, long long workspace_bytes, int device, void* stream);
/* backward of the 2x2 pooling (src/Unet.py:33-36) fused with the add of the skip connection's gradient (src/Unet.py:67):
 * g = gskip + route(gp) */
int mmft_u16_pool_bwd(const void* a, int lda, const void* gskip, int ldg, const void* gp, void* g, int N, int H, int W, int C,
                      int pool_mode, int device, void* stream);
/* ConvTranspose2d(Ci, Ci / 2, kernel_size=2, stride=2) (src/Unet.py:53) into / out of a channel slice (pitch ldu) of the
 * concatenation buffer; dw in the parameter's (a,b,co,ci) memory order */
int mmft_u16_convt_fwd(const void* x, const void* wpk, const float* bias, void* u, int ldu, int N, int h, int w, int Ci, int device,
                       void* stream);
int mmft_u16_convt_dgrad(const void* g, int ldu, const void* wpk_t, void* dx, int N, int h, int w, int Ci, int device, void* stream);
long long mmft_u16_convt_wgrad_workspace_bytes(int N, int h, int w, int Ci);
/* slabs of 4 Co Ci + 4 Co floats: [weights in the parameter's order | bias column sums per (a, b)]; dw == NULL: see above */
int mmft_u16_convt_wgrad_slabs(int N, int h, int w);
int mmft_u16_convt_wgrad(const void* x, const void* g, int ldu, float* dw, float* db, int accumulate, int N, int h, int w, int Ci,
                         float* workspace, long long workspace_bytes, int device, void* stream);
/* OutConv (src/Unet.py:71-82) on a bf16 input of 16 channels; out / gout fp32 */
int mmft_u16_outconv_fwd(const void* x, const float* w, const float* bias, float* out, int N, int H, int W, int mode, int device,
                         void* stream);
long long mmft_u16_outconv_bwd_workspace_bytes(int N, int H, int W);
/* slabs of 17 floats: [dw[16] | db]; dw == NULL: see above */
int mmft_u16_outconv_bwd_slabs(int N, int H, int W);
int mmft_u16_outconv_bwd(const void* x, const float* w, const float* bias, const float* gout, void* dx, float* dw, float* db,
                         int accumulate, int N, int H, int W, int mode, float* workspace, long long workspace_bytes, int device,
                         void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MMFT_H_ */

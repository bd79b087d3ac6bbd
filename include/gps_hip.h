/* gps_hip.h -- C ABI of libgps_hip.so: the MI355X (gfx950) kernels behind GPSLayer.
 *
 * This is the drop-in boundary of the port.  The reference has no
 * native code of its own; every entry point below replaces a *third-party* native op that the
 * reference reaches from Python.  Citations are relative to /root/reference.
 *
 * Conventions (all entry points):
 *   - plain C, no C++/torch types; device pointers are raw `void*`/typed pointers into memory
 *     owned by the caller (torch's caching allocator).  The library never allocates, frees or
 *     synchronises; it only enqueues kernels on `stream` (a hipStream_t passed as void*;
 *     NULL = the legacy default stream), so every call is hipGraph-capturable.
 *   - activations are contiguous row-major fp32; "ld" arguments are row strides in floats.
 *   - indices are int32 on the device (N, E < 2^31 is validated); `edge_index` is accepted in
 *     the reference's own layout, int64 [2, E] (row 0 = source j, row 1 = target i).
 *   - return 0 on success, a negative GPS_E* code otherwise; gps_last_error() returns a
 *     thread-local message.  No C++ exception crosses the boundary.  Re-entrant/thread-safe
 *     (called from the Python main thread in forward and the autograd thread in backward).
 */
#ifndef GPS_HIP_H
#define GPS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GPS_OK 0
#define GPS_EINVAL (-1)   /* bad argument (null pointer, negative size, unsupported dim) */
#define GPS_ELAUNCH (-2)  /* hipLaunchKernel / hipMemsetAsync reported an error */
#define GPS_EUNSUPPORTED (-3)

typedef void* gps_stream_t; /* hipStream_t */

/* One torch.nn.BatchNorm1d in training mode: its affine parameters and the [d] buffers that hold its batch statistics.
 * Consumed by the task-list norm kernels (gps_norm_*) and by the producers that emit batch statistics in their own
 * launch (gps_gatedgcn_fwd_stats, gps_gemm_panel_stats). */
typedef struct gps_bn {       /* one torch.nn.BatchNorm1d in training mode */
  const float* gamma;         /* weight [d] */
  const float* beta;          /* bias [d] */
  float* mean;                /* batch mean [d]: written by the statistics stages, read by apply/backward */
  float* rstd;                /* 1/sqrt(biased var + eps) [d] */
  float* running_mean;        /* updated by the statistics stages (both NULL: not tracked) */
  float* running_var;
  float eps, momentum;
} gps_bn;

int gps_abi_version(void);
const char* gps_last_error(void);

/* Optional device-resident dropout salt: one uint64 in device memory owned by the caller (NULL
 * turns it off, the default).  Every dropout-drawing kernel (segment attention, fused BN/act
 * epilogues) adds salt[0] * 0x9E3779B97F4A7C15 to its by-value seed.  Purpose: a training step
 * captured in a hipGraph replays the SAME by-value seeds; incrementing the salt inside the captured
 * step gives every replay fresh masks, with forward and backward of one replay still agreeing.
 * Process-global; set once before capture. */
int gps_set_dropout_salt(const uint64_t* device_salt);

/* ---------------------------------------------------------------------------------------
 * Per-batch graph index.  Replaces PyG's per-layer `MessagePassing._collect` index_select
 * bookkeeping (graphgps/layer/gatedgcn_layer.py:67-70) and `to_dense_batch`'s cumsum
 * (graphgps/layer/gps_layer.py:199) with structures built ONCE per batch:
 *   CSR by target : rowptr_dst[N+1], src_by_dst[E], eid_by_dst[E]   (edges of target i, in
 *                   ascending original edge id == the order torch_scatter's CPU scatter adds)
 *   CSC by source : rowptr_src[N+1], dst_by_src[E], eid_by_src[E]   (for the backward)
 * Deterministic (stable counting sort); bit-exact vs numpy stable argsort.
 * `ws` needs gps_graph_index_workspace_bytes(N, E) bytes.
 * ------------------------------------------------------------------------------------- */
size_t gps_graph_index_workspace_bytes(int64_t N, int64_t E);
int gps_graph_index_build(const int64_t* edge_index, int64_t N, int64_t E,
                          int32_t* rowptr_dst, int32_t* src_by_dst, int32_t* eid_by_dst,
                          int32_t* rowptr_src, int32_t* dst_by_src, int32_t* eid_by_src,
                          void* ws, size_t ws_bytes, gps_stream_t stream);

/* ptr[B+1] (cumulative node counts, int32) from the sorted int64 `batch` vector
 * (graphgps/layer/gps_layer.py:199 reads only `batch.batch`; PyG loader batches carry `ptr`). */
int gps_segment_ptr_from_batch(const int64_t* batch, int64_t N, int64_t B, int32_t* ptr,
                               gps_stream_t stream);

/* 16-row tile map for the segment-attention kernels: slot t of `tile_graph`/`tile_row0`
 * (both int32[max_tiles], max_tiles >= N/16 + B) holds (graph id, first global row) of one
 * 16-row tile, or -1.  Lets a fixed-size grid walk ragged graphs with no host sync. */
int gps_attn_tile_map(const int32_t* ptr, int64_t B, int64_t max_tiles, int32_t* tile_graph,
                      int32_t* tile_row0, gps_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * GatedGCN sparse core.  Replaces propagate()'s 3 index_select gathers, the sigmoid gate and
 * the 2 torch_scatter.scatter sums + the node update (graphgps/layer/gatedgcn_layer.py:67-70,
 * 90-136) with ONE gather-gate-segment-reduce launch forward and ONE launch backward.
 *   e_hat[eid] = Dx[i] + Ex[j] + Ce[eid]                      (edge order, pre-BN edge output)
 *   x_tilde[i] = Ax[i] + (sum_j sig*Bx[j]) / (sum_j sig + 1e-6)
 * Ax/Bx/Dx/Ex are [N, d] views with row stride ld_node (a fused [N,4d] / [N,7d] projection passes
 * ld_node = 4d / 7d).  Nothing is saved for the backward besides the outputs themselves: the backward
 * recomputes sum_j sig*Bx[j] and sum_j sig from e_hat in the forward's own summation order.
 * `r_edge` (float[E] by edge id, or NULL): the EquivStableLapPE gate r_ij of
 * graphgps/layer/gatedgcn_layer.py:101-104 -- sig is replaced by sig * r_edge[eid] in both sums
 * (and in the backward; the gradient wrt r_edge itself is the caller's: graphgps_amd/ops.py).
 * A workgroup owns a contiguous block of node rows (its CSR / CSC slices staged in LDS); d / vector
 * width must be <= 768 lanes (d <= 3072 for 16-byte aligned rows).
 * ------------------------------------------------------------------------------------- */
int gps_gatedgcn_fwd(const float* Ax, const float* Bx, const float* Dx, const float* Ex,
                     int64_t ld_node, const float* Ce, const int32_t* rowptr_dst,
                     const int32_t* src_by_dst, const int32_t* eid_by_dst, int64_t N, int64_t E,
                     int d, float* x_tilde, float* e_hat, const float* r_edge, gps_stream_t stream);
/* The same launch, additionally producing the batch statistics of x_tilde (-> bn_x: bn_node_x) and of e_hat
 * (-> bn_e: bn_edge_e), graphgps/layer/gatedgcn_layer.py:72-73: every lane accumulates shifted sums of the rows it
 * writes, a workgroup's row lanes meet through LDS, the node blocks' records -- (mean, M2) of both tensors with their two
 * row counts -- go to `ws`, and a SECOND launch (96 workgroups at d = 384) combines them in a fixed order into mean / rstd
 * / the running statistics (round 6; rounds 3 - 5 combined them in-launch through csrc/col_tree.hpp, whose tail cost this
 * HBM-bound kernel as much as the separate statistics pass it replaced).  d % 8 == 0, N, E >= 2.
 *   ws: gps_gatedgcn_stats_floats(N, d) floats.
 *   n_real: padded batches (loader.BucketPadding) -- device word with the number of REAL nodes: padding nodes (the last
 *   rows) and their incoming edges (padding joins padding only) are computed but never counted; NULL: every row counts. */
size_t gps_gatedgcn_stats_floats(int64_t N, int d);
int gps_gatedgcn_fwd_stats(const float* Ax, const float* Bx, const float* Dx, const float* Ex,
                           int64_t ld_node, const float* Ce, const int32_t* rowptr_dst,
                           const int32_t* src_by_dst, const int32_t* eid_by_dst, int64_t N, int64_t E,
                           int d, float* x_tilde, float* e_hat, const float* r_edge, const gps_bn* bn_x,
                           const gps_bn* bn_e, float* ws, size_t ws_floats, const int32_t* n_real, gps_stream_t stream);

/* Backward.  Inputs: g_x [N,d] with row stride ld_gx (grad wrt x_tilde), g_e [E,d] (grad wrt e_hat), the
 * forward's e_hat and x_tilde and its Ax / Bx inputs (ld_node).  Outputs: g_Ce [E,d]; g_Ax/g_Bx/g_Dx/g_Ex
 * [N,d] views with row stride ld_gnode (g_Ax = g_x; pass g_Ax == g_x with ld_gx == ld_gnode when the incoming
 * gradient already sits in its slot and the copy is skipped).  One launch: a target-keyed phase (num, den
 * recomputed, delta -> g_Ce, g_Dx; per-edge delta and sig*a handed over through LDS), a workgroup barrier, a
 * source-keyed phase (g_Ex, g_Bx) that takes the edges into this workgroup's own nodes from LDS and rebuilds the
 * rest (edges whose target another workgroup owns) from their inputs.
 * Deterministic, no data atomics; e_hat / g_e / g_Ce cross HBM once each.
 * ABI v6: amax_node / amax_ce (both or neither; NULL = off): max|.| records (GPS_AMAX_WORDS, below) raised to the maximum
 * over g_Ax | g_Bx | g_Dx | g_Ex and over g_Ce -- the scales of the fp16-form GEMMs that consume them. */
int gps_gatedgcn_bwd(const float* g_x, int64_t ld_gx, const float* g_e, const float* e_hat, const float* Ax,
                     const float* Bx, int64_t ld_node, const float* x_tilde,
                     const int32_t* rowptr_dst, const int32_t* src_by_dst,
                     const int32_t* eid_by_dst, const int32_t* rowptr_src,
                     const int32_t* dst_by_src, const int32_t* eid_by_src, int64_t N, int64_t E,
                     int d, float* g_Ce, float* g_Ax, float* g_Bx, float* g_Dx, float* g_Ex,
                     int64_t ld_gnode, const float* r_edge, uint32_t* amax_node, uint32_t* amax_ce, gps_stream_t stream);

/* ABI v9: the same backward with the two BatchNorm backward applies in front of it folded into its loads.  In
 * GatedGCNLayer (gatedgcn_layer.py:72-83) x~ and e^ each feed ONE BatchNorm1d -> ReLU -> dropout -> residual:
 *   x_out = x_in + dropout(relu(bn_node_x(x~))),   e_out = e_in + dropout(relu(bn_edge_e(e^)))
 * so the gradients this kernel consumes, g_x~ and g_e^, are the outputs of two BatchNorm backward applies that nothing
 * else reads, over tensors (x~, e^) the kernel reads anyway.  gps_gatedgcn_bwd_bn takes g_x1 / g_e1 -- the gradients of
 * the two residual sums -- and evaluates, per loaded element,
 *   g = relu' / dropout mask (row, col; p, seed) * g_y;  zhat = (z - mean) rstd;  gate * gamma rstd (g - S1/n - zhat S2/n)
 * exactly as gps_norm_bwd_apply does (same association; gate = 0 on padding rows past *rdev, n = the real rows), from six
 * column vectors per fold: the BatchNorm's mean / rstd / gamma / beta and its column sums S1 = sum g, S2 = sum g zhat
 * (gps_norm_bwd_partial, or the chain of a gps_norm_bwd_apply task).  Everything else as gps_gatedgcn_bwd; g_Ax receives
 * the folded g_x~ and may not alias g_x1. */
typedef struct gps_bn_bwd_fold {
  const gps_bn* bn;        /* mean, rstd (batch statistics of the forward), gamma, beta */
  const float* sum_g;      /* [d] S1 */
  const float* sum_gz;     /* [d] S2 */
  float p;                 /* dropout on the BatchNorm -> ReLU output (0 = none) */
  uint64_t seed;
  int32_t relu;
  const int32_t* rdev;     /* padded batches: device word with the number of real rows, or NULL */
} gps_bn_bwd_fold;
int gps_gatedgcn_bwd_bn(const float* g_x1, int64_t ld_gx, const float* g_e1, const float* e_hat, const float* Ax,
                        const float* Bx, int64_t ld_node, const float* x_tilde,
                        const int32_t* rowptr_dst, const int32_t* src_by_dst,
                        const int32_t* eid_by_dst, const int32_t* rowptr_src,
                        const int32_t* dst_by_src, const int32_t* eid_by_src, int64_t N, int64_t E,
                        int d, float* g_Ce, float* g_Ax, float* g_Bx, float* g_Dx, float* g_Ex,
                        int64_t ld_gnode, const float* r_edge, uint32_t* amax_node, uint32_t* amax_ce,
                        const gps_bn_bwd_fold* fold_x, const gps_bn_bwd_fold* fold_e, gps_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * GINE sparse core.  Replaces PyG GINEConv's gather + relu + scatter-add + (1+eps)*x
 * (constructed graphgps/layer/gps_layer.py:62-69, called :183-185):
 *   out[i] = (1+eps) * x[i] + sum_{j->i} relu(x[j] + e[eid])
 * Backward: g_e[eid] = g_out[i] * [x[j]+e[eid] > 0];  g_x[j] = (1+eps)*g_out[j] + sum_{j->.} g_e.
 * `r_edge` (float[E] by edge id, or NULL): GINEConvESLapPE's per-edge scale r_ij on the message
 * (graphgps/layer/gine_conv_layer.py:70-84): relu(.) * r_edge[eid], g_e scaled likewise.
 * ------------------------------------------------------------------------------------- */
int gps_gine_fwd(const float* x, const float* e, const int32_t* rowptr_dst,
                 const int32_t* src_by_dst, const int32_t* eid_by_dst, int64_t N, int64_t E, int d,
                 float eps, float* out, const float* r_edge, gps_stream_t stream);
int gps_gine_bwd(const float* g_out, const float* x, const float* e, const int32_t* rowptr_dst,
                 const int32_t* src_by_dst, const int32_t* eid_by_dst, const int32_t* rowptr_src,
                 const int32_t* eid_by_src, int64_t N, int64_t E, int d, float eps, float* g_x,
                 float* g_e, const float* r_edge, gps_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * GCN sparse core: out_i = dinv_i (dinv_i x_i + sum_{j->i, j != i} dinv_j x_j), dinv = deg^-1/2 with
 * deg_i = 1 + #{edges j->i, j != i}.  Replaces PyG GCNConv's gcn_norm (add_remaining_self_loops + scatter-add
 * degree) + propagate (gather, scale, scatter-add) -- the `GCN` local model of graphgps/layer/gps_layer.py:
 * 53-55,176-183 (configs/GPS/{actor,webkb-*,wn-*}-GPS.yaml).  x may be strided (ld_x >= d).
 * Forward: gps_gcn_dinv on the CSR-by-target, then gps_gcn_spmm(rowptr_dst, src_by_dst).  Backward (the
 * operator's transpose): the same gps_gcn_spmm on the CSC half, (rowptr_src, dst_by_src), applied to g_out.
 * ------------------------------------------------------------------------------------- */
int gps_gcn_dinv(const int32_t* rowptr_dst, const int32_t* src_by_dst, int64_t N, int64_t E, float* dinv,
                 gps_stream_t stream);
int gps_gcn_spmm(const float* x, int64_t ld_x, const int32_t* rowptr, const int32_t* nbr, const float* dinv,
                 int64_t N, int64_t E, int d, float* out, gps_stream_t stream);
/* out_i = self_w * x_i + sum_{k in segment(i)} x_{nbr[k]} (all stored edges, loops and duplicates included):
 * PyG GINConv's aggregation (1 + eps) x_i + sum_{j->i} x_j before its MLP, the phi network of SignNet
 * (graphgps/encoder/signnet_pos_encoder.py:70-110).  CSR-by-target forward, CSC-by-source = transpose. */
int gps_adj_sum(const float* x, int64_t ld_x, const int32_t* rowptr, const int32_t* nbr, float self_w, int64_t N,
                int64_t E, int d, float* out, gps_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * SAN edge attention over the REAL edges (csrc/edge_attn.hip).  Replaces, in MultiHeadAttentionLayer /
 * MultiHeadAttention2Layer.propagate_attention (graphgps/layer/san_layer.py:44-92, san2_layer.py:11-33,65-105), the
 * K[src] * Q[dst] * E gathers and product, the clamp-exp (SAN) or pyg_softmax = scatter_max + scatter_add (SAN2),
 * and the scatter-adds of V[src] * weight (and of the weight):
 *     s_e = sum_c K[j,h,c] Q[i,h,c] E[e,h,c] * scale                       edge e = (j -> i), head h
 *     softmax = 0:  w_e = exp(clamp(s_e, -5, 5));        wv[i] = sum_e w_e V[j];   z[i,h] = sum_e w_e
 *     softmax = 1:  w_e = exp(s_e - max_i) / (sum_i + 1e-16);   wv[i] = sum_e w_e V[j]   (mx / lsum [N,H] saved)
 * Q, K, V are [N, H*D] with row stride ld; E is [E, H*D] contiguous; D in {4, 8, 16, 32, 64}.  The fake-edge half
 * of the full-graph variants and the gamma mixing stay with the caller.  Backward: two launches, deterministic;
 * ws needs 2 * E * H floats.
 * ------------------------------------------------------------------------------------- */
int gps_edge_attn_supported(int H, int D);
int gps_edge_attn_fwd(const float* Q, const float* K, const float* V, int64_t ld, const float* Ee,
                      const int32_t* rowptr_dst, const int32_t* src_by_dst, const int32_t* eid_by_dst, int64_t N,
                      int64_t E, int H, int D, float scale, int softmax, float* wv, float* z, float* mx, float* lsum,
                      gps_stream_t stream);
int gps_edge_attn_bwd(const float* g_wv, const float* g_z, const float* Q, const float* K, const float* V, int64_t ld,
                      const float* Ee, const float* wv, const float* mx, const float* lsum,
                      const int32_t* rowptr_dst, const int32_t* src_by_dst, const int32_t* eid_by_dst,
                      const int32_t* rowptr_src, const int32_t* dst_by_src, const int32_t* eid_by_src, int64_t N,
                      int64_t E, int H, int D, float scale, int softmax, float* g_Q, float* g_K, float* g_V,
                      float* g_E, float* ws, gps_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Row-panel GEMM for the nn.Linear modules of the block and their input gradients (csrc/gemm_panel.hip):
 *     C[M, N] = A[M, K] * B[N, K]^T (+ bias[N]) (+ Cin[M, N])  with an optional ReLU / dropout epilogue.
 * Replaces the rocBLAS / hipBLASLt fp32 GEMMs behind graphgps/layer/gatedgcn_layer.py:57-61 (A..E),
 * graphgps/layer/gps_layer.py:104-106 (MultiheadAttention in/out projection) and :143-144,253-257 (FFN).
 * fp32 in, fp32 out; products formed exactly on the bf16 MFMA pipe (3-way exact split, 6 of 9 piece products,
 * fp32 accumulation: the rounding model of an fp32-input MFMA GEMM, ~7e-7 against fp64).
 * B is passed as a pre-split IMAGE (uint16 bf16 patterns, layout [3 pieces][ceil(K/32)][N'][32], N' = N rounded up to
 * whole column panels; the split kernel writes zeros into the k padding) produced from the fp32
 * weight by gps_gemm_split_weights -- once per optimizer step, for the weight (`image_nt`: B[n][k] = W[n][k],
 * forward) and/or its transpose (`image_tn`: B[n][k] = W[k][n], input gradient); gps_gemm_image_elems(N, K)
 * uint16 elements each (padding included).  Shapes: N % 4 == 0 and K % 4 == 0 (gps_gemm_panel_supported: column
 * panels of 192, 128 or 64, any number of 32-wide k-stages; when no panel width divides N or K % 32 != 0 -- d = 304,
 * 96, 72, 52, 48 -- the EDGE variants compute the surplus columns without storing them and multiply the partly empty
 * last stage by the image's zeros); anything else stays on the library GEMMs.  gps_gemm_panel_stats additionally needs whole
 * panels and stages (gps_gemm_stats_supported).
 * epilogue: 0 none | 1 relu then dropout(p_drop, seed) keyed (row, column) like gps_act_drop_add
 *           | 2 multiply by the relu/dropout mask of `mask_src` (= gps_act_drop_bwd applied to the product).
 * ------------------------------------------------------------------------------------- */
typedef struct gps_gemm_split {
  const float* W;      /* [rows][cols] fp32, row stride ldw (floats) */
  int64_t ldw;
  int rows, cols;
  uint16_t* image_nt;  /* image of B = W   (N = rows, K = cols), or NULL */
  uint16_t* image_tn;  /* image of B = W^T (N = cols, K = rows), or NULL */
} gps_gemm_split;
size_t gps_gemm_image_elems(int64_t N, int64_t K);
int gps_gemm_panel_supported(int64_t N, int64_t K);
int gps_gemm_split_weights(int n, const gps_gemm_split* descs, gps_stream_t stream);   /* n <= 56, one launch */
/* Debugging aid (tools/gemm_trace.py): the ring kernel stamps s_memtime per workgroup into buf (4 x uint64 each);
 * NULL switches it off. */
int gps_gemm_panel_trace(unsigned long long* buf);

int gps_gemm_panel(const float* A, int64_t lda, int64_t M, int K, const uint16_t* image, int N, const float* bias,
                   const float* Cin, int64_t ldcin, float* C, int64_t ldc, int epilogue, const float* mask_src,
                   int64_t ldmask, float p_drop, uint64_t seed, gps_stream_t stream);
/* Residual + dropout + BatchNorm statistics in the GEMM's epilogue (ring kernel):
 *   C = Cin + dropout(A W^T + bias; p_drop, seed)      and the batch statistics of C over its M rows
 *   -> stats->mean / rstd (+ running statistics), complete when the launch retires (csrc/col_tree.hpp: one tree per
 *   192-column panel, level-0 records = the row tiles).  Replaces `h + dropout(ff_linear2(t))` / `x + dropout(attn)`
 *   followed by the statistics pass of norm2 / norm1_attn (graphgps/layer/gps_layer.py:212-217,225-229).
 *   ws: gps_gemm_stats_floats(M, N, K) floats; sync: gps_gemm_stats_sync_words(N) uint32, zero at entry, zero at exit.
 *   Shapes: gps_gemm_stats_supported(M, N, K). */
size_t gps_gemm_stats_floats(int64_t M, int N, int K);
int gps_gemm_stats_sync_words(int N);
int gps_gemm_stats_supported(int64_t M, int N, int K);
int gps_gemm_panel_stats(const float* A, int64_t lda, int64_t M, int K, const uint16_t* image, int N, const float* bias,
                         const float* Cin, int64_t ldcin, float* C, int64_t ldc, float p_drop, uint64_t seed,
                         const gps_bn* stats, float* ws, size_t ws_floats, uint32_t* sync, gps_stream_t stream);

/* fp16 form of the same GEMM (round 4; csrc/gemm_panel.hip k_gemm_ring16): identical shapes, images' geometry and
 * epilogues, replaces the same library GEMMs (graphgps/layer/gatedgcn_layer.py:57-61, gps_layer.py:104-106,143-144,
 * 253-257).  Every operand value is carried as TWO fp16 pieces of its power-of-two-scaled value (round to nearest:
 * 22 significant bits) and a*b is formed from 3 piece products on v_mfma_f32_32x32x16_f16 with fp32 accumulation --
 * half the matrix-pipe work of the 6-product form; measured / emulated error against fp64 at or below the 6-product
 * form's (fewer roundings into the accumulator), both below an fp32 library GEMM.  The scale of an operand TENSOR is
 * derived from max|v|, which travels as a device RECORD: GPS_AMAX_WORDS (8) uint32 words holding fp32 bit patterns,
 * GPS_AMAX_STRIDE (64) words apart -- a footprint of GPS_AMAX_RECORD_WORDS (512) words, the rest unused.  The maximum is
 * the unsigned maximum of the eight words (order-free, deterministic); producers raise one of them per workgroup or
 * wavefront atomically (eight different cache lines: atomics on one line serialise at ~6 ns each).  Every `*_amax` /
 * `slot` / `amax*` pointer of this header addresses such a record: zero at allocation, only ever raised.
 *   gps_absmax          max over up to 56 row-major matrices per launch: record = max(record, max|A|) (the caller
 *                       zeroes the records once; several matrices may share a record)
 *   gps_gemm16_split_weights   image [2 pieces][ceil(K/32)][N'][32] fp16 of each weight under the scale of its `amax` word
 *   gps_gemm16_panel(_stats)   as gps_gemm_panel(_stats) with `a_amax` (>= max|A|) and `w_amax` (the word the image was
 *                       made with).  A word LARGER than the true maximum only costs precision (one bit per factor 2). */
#define GPS_AMAX_WORDS 8
#define GPS_AMAX_STRIDE 64
#define GPS_AMAX_RECORD_WORDS (GPS_AMAX_WORDS * GPS_AMAX_STRIDE)
typedef struct gps_absmax_desc {
  const float* A;      /* [rows][cols] fp32, row stride ld (floats); cols % 4 == 0, 16-byte aligned rows */
  int64_t ld, rows;
  int cols;
  uint32_t* slot;
} gps_absmax_desc;
int gps_absmax(int n, const gps_absmax_desc* descs, gps_stream_t stream);
typedef struct gps_gemm_split16 {
  const float* W;
  int64_t ldw;
  int rows, cols;
  uint16_t* image_nt;
  uint16_t* image_tn;
  const uint32_t* amax;   /* max|W| word (gps_absmax), read by the kernel */
} gps_gemm_split16;
size_t gps_gemm16_image_elems(int64_t N, int64_t K);
int gps_gemm16_split_weights(int n, const gps_gemm_split16* descs, gps_stream_t stream);   /* n <= 48, one launch */
int gps_gemm16_panel(const float* A, int64_t lda, int64_t M, int K, const uint32_t* a_amax, const uint16_t* image,
                     const uint32_t* w_amax, int N, const float* bias, const float* Cin, int64_t ldcin, float* C, int64_t ldc,
                     int epilogue, const float* mask_src, int64_t ldmask, float p_drop, uint64_t seed, uint32_t* c_amax,
                     gps_stream_t stream);      /* c_amax (or NULL): raised to max|C| -- the word of the GEMM that reads C next */
int gps_gemm16_panel_stats(const float* A, int64_t lda, int64_t M, int K, const uint32_t* a_amax, const uint16_t* image,
                           const uint32_t* w_amax, int N, const float* bias, const float* Cin, int64_t ldcin, float* C,
                           int64_t ldc, float p_drop, uint64_t seed, const gps_bn* stats, float* ws, size_t ws_floats,
                           uint32_t* sync, const int32_t* m_dev, gps_stream_t stream);
/* m_dev (or NULL): padded batches -- device word with the number of REAL rows; rows past it are computed and stored but
 * stay out of the column statistics (see gps_norm_fwd_task.rdev). */
/* Two INDEPENDENT products C = A W^T + bias (+ Cin) in one dispatch (epilogue 0 each): the edge projection C(e) beside the
 * merged node projection of a GPS block (graphgps/layer/gatedgcn_layer.py:57-61 + gps_layer.py:238), and the two input
 * gradients g_pq Wcat / g_ce W_C of its backward.  Workgroups of `first` are dispatched first: pass the problem with the
 * longer contraction there.  Same arithmetic per problem as gps_gemm16_panel; problems whose shapes take different kernel
 * families (edge shapes, different column panel widths, one with an addend and one without) run as two launches. */
typedef struct gps_gemm16_problem {
  const float* A;
  int64_t lda, M;
  int32_t K, N;
  const uint32_t* a_amax;
  const uint16_t* image;
  const uint32_t* w_amax;
  const float* bias;      /* [N] or NULL */
  const float* Cin;       /* addend [M][N] (row stride ldcin) or NULL; may alias C */
  int64_t ldcin;
  float* C;
  int64_t ldc;
  uint32_t* c_amax;       /* or NULL */
} gps_gemm16_problem;
int gps_gemm16_panel_pair(const gps_gemm16_problem* first, const gps_gemm16_problem* second, gps_stream_t stream);

/* ABI v10: an input-gradient GEMM whose output IS the output gradient of two training-mode BatchNorm1d's leaves with their
 * backward column sums -- autograd's sum(g) and sum(g * zhat) over the batch (= g_beta and g_gamma), which
 * gps_norm_bwd_partial otherwise takes in a pass of its own over g and both inputs.  In a GPS block
 * (graphgps/layer/gps_layer.py:219-222,225-229) g_h = g_z2 + g_f1 W1 -- the residual of z2 = h + ff(h) plus the FFN's input
 * gradient -- is the output gradient of h = norm1_local(x1) + norm1_attn(za).  The sums are complete when the launch retires
 * (one in-launch SUMS tree per column panel, csrc/col_tree.hpp: `sync` = gps_gemm_stats_sync_words(N) counters, zero at
 * entry and exit); rows are summed in tile order: deterministic, not bit-identical to the row pass.  Padding rows of a
 * padded batch must be zero in C (they are: every gradient of the block is).  (The same epilogue on the PAIRED input
 * gradients of a block, for norm2 / bn_edge_e of the block below, was built and measured in round 6: +26 us on the pair
 * for the 23 us launch it removed -- not kept.) */
typedef struct gps_gemm_colsums {
  const float* z;          /* [M][N] input of the first BatchNorm (row stride ldz) */
  int64_t ldz;
  const gps_bn* bn;        /* mean, rstd */
  float* sum_g;            /* [N] out: S1 = sum_rows C */
  float* sum_gz;           /* [N] out: S2 = sum_rows C * zhat */
  const float* z2;         /* [M][N] input of the second BatchNorm */
  int64_t ldz2;
  const gps_bn* bn2;
  float* sum_g2;           /* [N] out: S1 again (the second BatchNorm's g_beta) */
  float* sum_gz2;          /* [N] out: S2 of the second BatchNorm */
  float* ws;               /* gps_gemm_colsums_floats(M, N, K) floats, 16-byte aligned */
  size_t ws_floats;
  uint32_t* sync;
} gps_gemm_colsums;
int gps_gemm_colsums_supported(int64_t M, int N, int K);     /* whole 128- / 192-column panels, whole k-stages, M >= 2 */
size_t gps_gemm_colsums_floats(int64_t M, int N, int K);
/* C = Cin + A W^T (+ bias) and the sums (prob->Cin required; may alias C) */
int gps_gemm16_panel_sums(const gps_gemm16_problem* prob, const gps_gemm_colsums* sums, gps_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Segment (per-graph, varlen) multi-head attention core on fp32 MFMA (v_mfma_f32_16x16x4_f32).
 * Replaces to_dense_batch + the softmax(QK^T/sqrt(dh) + key-padding mask) -> dropout -> .V core
 * of torch.nn.MultiheadAttention + the [mask] un-pad (graphgps/layer/gps_layer.py:199-201,
 * 234-241).  No padding, no mask tensor: graphs are walked through `ptr`.
 *   qkv  [N, 3d] packed in-proj output (q | k | v, head h at columns h*dh..), row stride ld_qkv
 *   out  [N, d]   heads merged;   lse [H, N] log-sum-exp per (head, query), saved for backward
 * Attention dropout (p_drop in [0,1)) uses a counter-based hash of (seed, query, head, key pair) so
 * that forward and backward regenerate the same mask (one 32-bit hash decides two adjacent keys, 16 bits
 * each: the realised drop probability is round(p * 2^16) / 2^16 and survivors are scaled by its exact
 * complement); p_drop == 0 disables it.
 * Two kernel families behind the same entry points.  `num_graphs` = B (entries of `ptr` minus one) and
 * `max_graph_nodes` = an upper bound on the longest graph of the batch known to the HOST (0 = unknown): when it
 * is <= 64 and dh is one of {8, 16, 24, 32}
 * the block form runs -- one wavefront per (graph, head) keeps that graph's K-side operands in registers and
 * serves all of its query tiles; the backward is ONE launch (csrc/sattn.hip).  Everything else takes the
 * one-wavefront-per-(16-row tile, head) form with the online softmax (csrc/seg_attention.hip).
 * Supported head dims: the compiled set {4, 6, 8, 10, 12, 13, 16, 18, 20, 24, 32, 48, 64, 76, 96, 128}
 * (every dim_hidden / n_heads of configs/GPS and configs/Graphormer), see gps_attn_supported_head_dim().
 * ------------------------------------------------------------------------------------- */
int gps_attn_supported_head_dim(int dh);
int gps_seg_attn_fwd(const float* qkv, int64_t ld_qkv, const int32_t* ptr,
                     const int32_t* tile_graph, const int32_t* tile_row0, int64_t max_tiles,
                     int64_t N, int H, int dh, float scale, float p_drop, uint64_t seed, float* out,
                     float* lse, int64_t num_graphs, int64_t max_graph_nodes, uint32_t* amax, const int32_t* graph_order,
                     gps_stream_t stream);
/* ABI v11: `graph_order` (or NULL) of gps_seg_attn_fwd / _bwd: int32 [num_graphs] made by gps_attn_graph_order -- the order
 * in which the block form dispatches the graphs (slot t -> graph id).  Every wavefront of a block-form launch is resident at
 * once, so a SIMD's time is the sum of the work of the wavefronts it holds; dealing long and short graphs to the CUs in a
 * snake levels that sum (P30 x 256: forward 22.5 -> 18 us, backward 47.8 -> 38 us).  Scheduling only: results are
 * bit-identical with or without it; the tile-map kernels ignore it. */
int gps_attn_graph_order(const int32_t* ptr, int64_t B, int H, int32_t* order, gps_stream_t stream);
/* ABI v6: `amax` (or NULL) of gps_seg_attn_fwd / _bwd: a max|.| record (GPS_AMAX_WORDS) raised to max|out| / max|d_qkv| --
 * by the block-form kernels themselves (one atomic per wavefront), by a gps_absmax pass behind the tile kernels. */
/* d_qkv [N,3d] (row stride ld_dqkv) receives dq | dk | dv.  `delta` is an [H,N] scratch buffer
 * (rowsum(dO*O)).  Block form (max_graph_nodes in 1..64, see above): ONE launch -- S, P, dS computed once per
 * tile pair, dQ / dK / dV from it, delta formed on the fly (the scratch buffer stays untouched); otherwise
 * two launches: dQ + delta (query-tile keyed), dK + dV (key-tile keyed). */
int gps_seg_attn_bwd(const float* d_out, const float* qkv, int64_t ld_qkv, const float* out,
                     const float* lse, const int32_t* ptr, const int32_t* tile_graph,
                     const int32_t* tile_row0, int64_t max_tiles, int64_t N, int H, int dh,
                     float scale, float p_drop, uint64_t seed, float* delta, float* d_qkv,
                     int64_t ld_dqkv, int64_t num_graphs, int64_t max_graph_nodes, uint32_t* amax,
                     const int32_t* graph_order, gps_stream_t stream);
/* The same core with an additive attention bias: the reference's `attn_mask=batch.attn_bias` operand of
 * torch.nn.MultiheadAttention in the BiasedTransformer branch (graphgps/layer/gps_layer.py:201-203,
 * 234-241) and in GraphormerLayer (graphgps/layer/graphormer_layer.py:43-44).
 *   bias   fp32 [B*H, nmax, nmax], plane (g*H + h), element [query, key] -- the layout the reference's
 *          BiasEncoder emits (graphgps/encoder/graphormer_encoder.py:148-183); nmax >= every graph size.
 *          softmax(q k^T / sqrt(dh) + bias): only the n_g x n_g corner of each plane is read.
 *   d_bias same shape, caller ZERO-FILLED: the backward writes dS into the n_g x n_g corners only
 *          (the padded region has zero gradient in the reference as well: -inf-masked keys).
 * nmax * nmax must stay below 2^31 (32-bit element offsets inside a plane). */
int gps_seg_attn_bias_fwd(const float* qkv, int64_t ld_qkv, const float* bias, int64_t nmax,
                          const int32_t* ptr, const int32_t* tile_graph, const int32_t* tile_row0,
                          int64_t max_tiles, int64_t N, int H, int dh, float scale, float p_drop,
                          uint64_t seed, float* out, float* lse, gps_stream_t stream);
int gps_seg_attn_bias_bwd(const float* d_out, const float* qkv, int64_t ld_qkv, const float* bias,
                          int64_t nmax, const float* out, const float* lse, const int32_t* ptr,
                          const int32_t* tile_graph, const int32_t* tile_row0, int64_t max_tiles, int64_t N,
                          int H, int dh, float scale, float p_drop, uint64_t seed, float* delta,
                          float* d_qkv, int64_t ld_dqkv, float* d_bias, gps_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * FAVOR+ (Performer softmax-kernel linear attention) over `ptr` segments on fp32 MFMA.
 * Replaces to_dense_batch + performer_pytorch.SelfAttention's core (graphgps/layer/gps_layer.py:
 * 111-114,206; arithmetic graphgps/layer/performer_layer.py:119-144,200-205,485-492): feature maps
 * phi_q/phi_k through the [m, 64] projection `proj`, ksum, ctx = phi_k^T v, out = phi_q ctx / (phi_q ksum).
 * The reference runs on the PADDED batch and masks only V, so every graph's normaliser also sees
 * (Nmax - n_g) zero key rows; `nmax` (device int32[1], from gps_segment_max_len) supplies Nmax and
 * the kernels add that term in closed form.  dim_head must be 64 and m <= 272.
 *   qkv [N, 3*64H] (q | k | v, no bias), out [N, 64H]
 *   saved for backward: ctx [B*H,272,64], ksum [B*H,272], kmax uint64[B*H] (key max + arg-max),
 *   mq [H,N] (query row max), D [H,N] (normaliser)
 * Backward additionally needs scratch gD [H,N], g_ctx [B*H,272,64], g_ksum [B*H,272],
 * gM_part [max_tiles*H]; d_qkv [N, 3*64H] receives dq | dk | dv.
 * ------------------------------------------------------------------------------------- */
int gps_segment_max_len(const int32_t* ptr, int64_t B, int32_t* nmax, gps_stream_t stream);
/* The same over the first b_real[0] graphs only (b_real: device int32 word, or NULL = all B): the Nmax of a PADDED batch
 * (loader.BucketPadding appends dead graphs behind the real ones; the reference's to_dense_batch Nmax -- and with it the
 * padded-key term of performer_layer.py:485-487 -- is the longest graph the loader emitted). */
int gps_segment_max_len_real(const int32_t* ptr, int64_t B, const int32_t* b_real, int32_t* nmax, gps_stream_t stream);
/* ABI v6: `ws` (gps_favor_workspace_floats(N, B, H) floats, or NULL): partial context records -- the rows of a graph are
 * dealt to several wavefronts per (graph, head, feature tile) and summed in slice order (deterministic); without it one
 * wavefront walks all rows of its graph (the round-1 form).
 * Round 5 (no ABI change): the per-tile kernels behind both calls run in one of three forms chosen from the launch shape
 * -- one wavefront per (16-row tile, head); one workgroup per CU with the projection staged in LDS; workgroups over chunks
 * of 4 / 8 tiles of one graph with the (graph, head) context record staged as well (long graphs, B <= 1024).  All three do
 * the same arithmetic in the same order: results are bit-identical whichever is taken.  gM_part is indexed by the tile
 * map of gps_graph_index_build (slot of tile t of graph g: (ptr[g] >> 4) + g + t). */
size_t gps_favor_workspace_floats(int64_t N, int64_t B, int H);
int gps_favor_fwd(const float* qkv, int64_t ld_qkv, const float* proj, int m, const int32_t* ptr,
                  const int32_t* nmax, const int32_t* tile_graph, const int32_t* tile_row0,
                  int64_t max_tiles, int64_t N, int64_t B, int H, int dh, float* out, float* ctx,
                  float* ksum, uint64_t* kmax, float* mq, float* D, float* ws, size_t ws_floats, gps_stream_t stream);
int gps_favor_bwd(const float* g_out, const float* qkv, int64_t ld_qkv, const float* proj, int m,
                  const float* out, const int32_t* ptr, const int32_t* nmax, const int32_t* tile_graph,
                  const int32_t* tile_row0, int64_t max_tiles, int64_t N, int64_t B, int H, int dh,
                  const float* ctx, const float* ksum, const uint64_t* kmax, const float* mq,
                  const float* D, float* gD, float* g_ctx, float* g_ksum, float* gM_part,
                  float* d_qkv, int64_t ld_dqkv, float* ws, size_t ws_floats, gps_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Fused BatchNorm1d epilogues for [R, d] activation streams (training mode, batch statistics).
 * Replace the module chains around every BatchNorm1d of the path:
 *   x_in + dropout(relu(bn(x)))                    graphgps/layer/gatedgcn_layer.py:72-83
 *   norm(h_in + dropout(branch)), norm2(h + ffn)   graphgps/layer/gps_layer.py:191-194,212-217,225-229
 *   dropout(relu(ff_linear1(h)))                   graphgps/layer/gps_layer.py:253-257
 * i.e. ATen batch_norm (collect_statistics/transform/backward_reduce/backward_elemt), relu, dropout
 * (Philox) and add kernels.  torch.nn.BatchNorm1d semantics: biased variance for normalisation,
 * unbiased for running_var, eps inside the sqrt, running = (1-momentum)*running + momentum*batch.
 * Dropout masks come from the same counter hash as the attention kernel, keyed (seed, row, column),
 * so the backward recomputes them (and the ReLU mask) instead of storing them.
 *   gps_bn_stats : mean[d], rstd[d] (+ running stats update when non-NULL); ws >= gps_bn_workspace_floats
 *   gps_bn_apply : y = res + drop(relu((z-mean)*rstd*gamma+beta)); relu / p_drop==0 / res==NULL switch stages off
 *   gps_bn_bwd   : g_z, g_gamma, g_beta from g_y (grad wrt y; the residual's grad is g_y itself)
 *   gps_act_drop_add / _bwd : out = a + drop(relu?(b)) (a may be NULL) and g_b
 * ------------------------------------------------------------------------------------- */
size_t gps_bn_workspace_floats(int64_t R, int d);
int gps_bn_stats(const float* z, int64_t R, int d, float eps, float momentum, float* mean, float* rstd,
                 float* running_mean, float* running_var, float* ws, gps_stream_t stream);
int gps_bn_apply(const float* z, const float* mean, const float* rstd, const float* gamma,
                 const float* beta, const float* res, int64_t R, int d, int relu, float p_drop,
                 uint64_t seed, float* y, gps_stream_t stream);
int gps_bn_bwd(const float* z, const float* g_y, const float* mean, const float* rstd,
               const float* gamma, const float* beta, int64_t R, int d, int relu, float p_drop,
               uint64_t seed, float* g_z, float* g_gamma, float* g_beta, float* ws,
               gps_stream_t stream);
int gps_act_drop_add(const float* a, const float* b, int64_t R, int d, int relu, float p_drop,
                     uint64_t seed, float* out, gps_stream_t stream);
int gps_act_drop_bwd(const float* g, const float* pre, int64_t R, int d, int relu, float p_drop,
                     uint64_t seed, float* g_b, gps_stream_t stream);
/* out[d] = column sums of x[R,d] (deterministic two-stage reduction; ws >= gps_bn_workspace_floats).
 * Replaces ATen's reduce_kernel for the bias gradients of the dense projections (autograd of
 * nn.Linear at graphgps/layer/gatedgcn_layer.py:57-61, graphgps/layer/gps_layer.py:143-144). */
int gps_colsum(const float* x, int64_t R, int d, float* out, float* ws, gps_stream_t stream);

/* Weight and bias gradient of y = x W^T + b in one pass on fp32 MFMA (split-K, deterministic):
 *   gw[M, Nn] = g^T x   with g [R, M] (row stride ldg), x [R, Nn] (row stride ldx);   gb[M] = colsum(g)
 * (gb may be NULL).  Replaces the long-K/small-output GEMMs autograd issues through rocBLAS for the
 * nn.Linear modules of the block (graphgps/layer/gatedgcn_layer.py:57-61, gps_layer.py:143-144) and
 * the separate bias-gradient reductions.  M, Nn, ldg, ldx multiples of 4; ws >=
 * gps_wgrad_workspace_floats(R, M, Nn) floats. */
size_t gps_wgrad_workspace_floats(int64_t R, int M, int Nn);
int gps_wgrad(const float* g, int64_t ldg, const float* x, int64_t ldx, int64_t R, int M, int Nn,
              float* gw, float* gb, float* ws, gps_stream_t stream);
/* Dense projection, fp32 in / fp32 out:  C[R,M] = A[R,K] * B[M,K]^T (+ bias[M]) (+ Cin[R,M]).
 * With B = an nn.Linear weight [out, in] this is y = x W^T + b (graphgps/layer/gatedgcn_layer.py:57-61,
 * graphgps/layer/gps_layer.py:143-144,234-241,253-257); with B = W^T it is the input gradient g W, and Cin
 * folds the residual accumulation of the backward into the epilogue (C may alias Cin).
 * The contraction runs on the bf16 MFMA pipe through an EXACT 3-way bf16 split of both operands (6 of the
 * 9 piece products, fp32 accumulate; every partial product is exact in fp32): same rounding model as an
 * fp32-input MFMA GEMM, error vs fp64 at or below rocBLAS' fp32 GEMM (tests/test_hip_ops.py).
 * K, lda, ldb multiples of 4; A, B 16-byte aligned. */
int gps_gemm_nt(const float* A, int64_t lda, const float* B, int64_t ldb, int64_t R, int M, int K,
                const float* bias, const float* Cin, int64_t ldcin, float* C, int64_t ldc,
                gps_stream_t stream);

/* Grouped form: up to 8 independent weight(+bias)-gradient problems -- the five of one GPS block:
 * merged A|B|D|E|in_proj, C, out_proj, ff_linear1, ff_linear2 -- in ONE launch + ONE reduce launch,
 * the row ranges split so that every workgroup gets an equal share of the whole list.
 * Same arithmetic per problem as gps_wgrad; ws >= gps_wgrad_grouped_workspace_floats(n, probs). */
typedef struct gps_wgrad_problem {
  const float* g;   /* [R, M] upstream gradient, row stride ldg */
  const float* x;   /* [R, Nn] layer input, row stride ldx */
  float* gw;        /* [M, Nn] out */
  float* gb;        /* [M] out, or NULL */
  int64_t ldg, ldx, R;
  int32_t M, Nn;
  /* ABI v6: max|g| / max|x| words (gps_absmax) -- when EVERY problem of a launch carries both (and the shapes take the
   * streaming kernel: M, Nn multiples of 128), the contraction runs in the fp16 form of gps_gemm16_panel (two fp16 pieces
   * per value, 3 piece products); NULL keeps the 3 x bf16 / 6-product form. */
  const uint32_t* g_amax;
  const uint32_t* x_amax;
} gps_wgrad_problem;
size_t gps_wgrad_grouped_workspace_floats(int n, const gps_wgrad_problem* probs);
int gps_wgrad_grouped(int n, const gps_wgrad_problem* probs, float* ws, gps_stream_t stream);
/* gps_wgrad with the operands' max|.| words: the fp16 form where the streaming kernel applies (else as gps_wgrad) */
int gps_wgrad16(const float* g, int64_t ldg, const float* x, int64_t ldx, int64_t R, int M, int Nn, const uint32_t* g_amax,
                const uint32_t* x_amax, float* gw, float* gb, float* ws, gps_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Graph pooling over `ptr` segments (sum: mean = 0, mean: mean = 1) and its backward.
 * Replaces GraphGym pooling_dict['add'|'mean'] (torch_scatter atomics), called from
 * graphgps/head/san_graph.py:35 and graphgps/head/ogb_code_graph.py:37.
 * `node_graph` int32 [N] (graph id per node) comes from gps_node_graph_from_ptr.
 * ------------------------------------------------------------------------------------- */
int gps_node_graph_from_ptr(const int32_t* ptr, int64_t B, int32_t* node_graph, gps_stream_t stream);
/* Weight gradient of an nn.Embedding lookup with a large vocabulary: g_w[t] = sum of the rows g[perm[k]] over the
 * lookups k of token t.  Replaces ATen's embedding_dense_backward (radix sort + sum_and_scatter, summation order not
 * fixed) behind graphgps/encoder/ast_encoder.py:35-83 (the 10,030-entry attribute table of ogbg-code2).
 *   tok_sorted  int64 [n] token ids in ascending order (a STABLE sort of the lookup indices), perm its permutation
 *   g_w         [V, d], ZERO on entry (tokens without lookups keep a zero row); d in {64, 128, 256}
 * Segmented reduction over fixed units of 64 entries, partial rows of runs that span units merged in unit order:
 * deterministic, no atomics.  ws: gps_embedding_grad_workspace_bytes(n, d). */
size_t gps_embedding_grad_workspace_bytes(int64_t n, int d);
int gps_embedding_grad_supported(int d);
int gps_embedding_grad(const float* g, const int64_t* tok_sorted, const int64_t* perm, int64_t n, int64_t V, int d,
                       float* g_w, void* ws, size_t ws_bytes, gps_stream_t stream);
/* Sum-of-embeddings encoders over small vocabularies (csrc/embed.hip, round 5): out[r, :] = sum_i table_i[feats[r, i], :]
 * -- OGB's AtomEncoder / BondEncoder as the reference registers them (graphgps/encoder/ via ogb.graphproppred.mol_encoder:
 * `x_embedding += emb[i](x[:, i])`), the AST type / depth tables (graphgps/encoder/ast_encoder.py:35-83) and the TypeDict
 * encoders (graphgps/encoder/type_dict_encoder.py) -- in ONE launch, columns added in index order (the reference's own
 * summation order).  feats: int64 [R, k] with row stride ld, 1 <= k <= 16; tables: HOST array of k device pointers to
 * contiguous [vocab[i], emb] fp32 tables (16-byte aligned, emb % 4 == 0); indices outside a table are clamped, never faulted
 * on (nn.Embedding raises on them on the host side of the reference).
 * gps_multihot_fill writes the [R, vpad] multi-hot matrix of the same features (1.0 at column offset_i + feats[r, i],
 * vpad = gps_multihot_columns(k, vocab) = sum of the vocabularies rounded up to a multiple of 4) whose transpose times the
 * output gradient is the stacked table gradient: the deterministic form of k embedding_dense_backward passes. */
int gps_embed_sum(const int64_t* feats, int64_t ld, int64_t R, int k, const float* const* tables, const int* vocab, int emb,
                  float* out, gps_stream_t stream);
int gps_multihot_columns(int k, const int* vocab);
int gps_multihot_fill(const int64_t* feats, int64_t ld, int64_t R, int k, const int* vocab, float* out, gps_stream_t stream);
/* Small dense products of the graph-level heads (csrc/small_gemm.hip, round 6): the Linear (+ ReLU) stages of
 * graphgps/head/san_graph.py:19-42 on the pooled [B, dim_in] embedding and what autograd derives for them -- a few hundred
 * rows, widths 1 .. 384 -- as exact fp32 products on v_mfma_f32_16x16x4_f32, one 16 x 16 output tile per workgroup.
 *   gps_small_linear_fwd   y[M, N] = act(x[M, K] w[N, K]^T + bias)   (bias may be NULL; relu != 0: ReLU)
 *   gps_small_linear_bwd   g' = g (.) [y > 0] when y != NULL (the layer applied ReLU), else g;
 *                          g_x[M, K] = g' w  (NULL: skipped);  g_w[N, K] = g'^T x and g_b[N] = column sums of g'
 *                          (g_w NULL: both skipped; g_b may be NULL)
 * Row strides in elements; results are deterministic (contraction slices added in a fixed order, no atomics). */
int gps_small_linear_fwd(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* bias, int M, int N, int K,
                         int relu, float* y, int64_t ldy, gps_stream_t stream);
int gps_small_linear_bwd(const float* g, int64_t ldg, const float* y, int64_t ldy, const float* x, int64_t ldx,
                         const float* w, int64_t ldw, int M, int N, int K, float* g_x, int64_t ldgx, float* g_w,
                         int64_t ldgw, float* g_b, gps_stream_t stream);
int gps_segment_pool_fwd(const float* x, const int32_t* ptr, int64_t B, int d, int mean, float* out,
                         gps_stream_t stream);
/* The same pooling with the rows of a graph cut into slices of 32 (round 5): one workgroup per (graph, slice) -- slot
 * floor(ptr[g] / 32) + g is graph g's first slice, floor(N / 32) + B slots in all, found on the device from `ptr` alone --
 * partial rows of multi-slice graphs summed in slice order by a second launch.  Same result contract as
 * gps_segment_pool_fwd (global_add_pool / global_mean_pool: graphgps/head/san_graph.py:35, ogb_code_graph.py:37; an empty
 * graph pools to 0), deterministic; what it changes is who does the work: 32 graphs of ~800 rows were 8 workgroups.
 * ws: gps_segment_pool_workspace_bytes(N, B, d) bytes, 16-byte aligned, contents irrelevant on entry. */
size_t gps_segment_pool_workspace_bytes(int64_t N, int64_t B, int d);
int gps_segment_pool_fwd_sliced(const float* x, const int32_t* ptr, int64_t N, int64_t B, int d, int mean, float* out,
                                void* ws, size_t ws_bytes, gps_stream_t stream);
int gps_segment_pool_bwd(const float* g_out, const int32_t* ptr, const int32_t* node_graph,
                         int64_t N, int d, int mean, float* g_x, gps_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Task-list BatchNorm / residual / dropout stages of the fused GPS blocks (csrc/block_norm.hip).
 * Same arithmetic as the gps_bn_* / gps_act_drop_* entry points above (identical formulas, statistics and
 * counter-hash dropout), issued as LISTS of up to 4 independent row streams per launch; the column reductions
 * every BatchNorm needs (batch statistics forward; sum g, sum g*zhat backward) complete INSIDE the producing
 * launch (csrc/col_tree.hpp: write-through partial records, arrival tickets, the last-arriving workgroup combines
 * in index order -- deterministic, no finalize launch, nobody waits).
 * Reference lines: graphgps/layer/gatedgcn_layer.py:72-83 (bn_node_x / bn_edge_e + ReLU + dropout +
 * residual), graphgps/layer/gps_layer.py:191-194 (norm1_local), :212-217 (dropout_attn + residual +
 * norm1_attn), :222 (branch sum), :225-229 (ff_dropout2 + residual + norm2), and their autograd backward.
 * All row buffers [R, d] fp32, d % 4 == 0, d <= 1024, 16-byte aligned, R >= 2.
 *   ws / ws_floats  scratch for the partial records: the sum of gps_norm_tree_floats(R, d) over the tasks that
 *                   reduce (forward: tasks with `stats`; partial: every task; apply: tasks with a chain)
 *   sync            gps_norm_sync_words() uint32 arrival counters, ZERO at entry and zero again at exit (so a
 *                   buffer zeroed once serves every later call and every hipGraph replay).  Launches that may run
 *                   concurrently (forked streams) must be given different counters.
 * ------------------------------------------------------------------------------------- */
enum { GPS_NORM_LOAD = 0,     /* rows = a                                  (statistics of an existing tensor) */
       GPS_NORM_ADD_DROP = 1, /* rows = a + dropout(b; p, seed)                                               */
       GPS_NORM_BN_ACT = 2,   /* rows = [res +] dropout(relu?(BN1(a)); p, seed)                                */
       GPS_NORM_BN_DUAL = 3   /* rows = BN1(a) + BN2(b)                    (gps_layer.py:194,217,222)          */ };
typedef struct gps_norm_fwd_task {
  int32_t kind, relu;
  const float *a, *b, *res;
  const gps_bn *bn1, *bn2;    /* statistics already final (a previous launch) */
  float p;
  uint64_t seed;
  float* out;                 /* produced rows, or NULL (statistics only) */
  int64_t R;
  const gps_bn* stats;        /* batch statistics of the produced rows -> stats->mean / rstd (+ running stats), or NULL */
  uint32_t* amax;             /* ABI v6: raised to max|out| (fp32 bit pattern; the word gps_gemm16_panel takes), or NULL */
  const int32_t* rdev;        /* ABI v6, padded batches: device word holding the number of REAL rows r (rows r .. R-1 are
                                 padding: produced like any row, excluded from the statistics), or NULL = all R rows */
} gps_norm_fwd_task;
/* One BatchNorm backward, y = dropout(relu?(BN(z)); p, seed) with g_y = dL/dy (masks recomputed from z and the hash):
 *   partial: g_beta = sum g, g_gamma = sum g * zhat          (g = g_y under the masks)
 *   apply:   g_z = gamma * rstd * (g - g_beta / R - zhat * g_gamma / R), reading g_beta / g_gamma
 * dual (z2 != NULL): out = BN(z) + BN2(z2) (no masks); g_gamma2 / g_beta2 likewise; the apply writes g_z (stored as
 *   dropmask(seed1x, p1x)(g_z) / (1 - p1x) when p1x > 0), g_sum = g_z + g_z2 and
 *   g_drop = dropmask(seed2, p2)(g_z2) / (1 - p2); non-dual: g_drop = dropmask(seed2, p2)(g_z) / (1 - p2).
 * chain (apply only, cz != NULL): the produced g_z (before the p1x mask) is the output gradient of ANOTHER
 *   BatchNorm backward, y' = dropout(relu?(BN_c(cz)); cp, cseed): its sums go to cg_beta / cg_gamma in the same pass. */
typedef struct gps_norm_bwd_task {
  const float *z, *g_y;
  const gps_bn* bn;
  int32_t relu;
  float p;
  uint64_t seed;
  const float* z2;
  const gps_bn* bn2;
  float *g_gamma, *g_beta, *g_gamma2, *g_beta2;
  float *g_z, *g_sum, *g_drop;
  float p2;
  uint64_t seed2;
  float p1x;
  uint64_t seed1x;
  int64_t R;
  const float* cz;
  const gps_bn* cbn;
  int32_t crelu;
  float cp;
  uint64_t cseed;
  float *cg_gamma, *cg_beta;
  uint32_t* amax_drop;        /* ABI v6 (apply only): raised to max|g_drop|, or NULL */
  const int32_t* rdev;        /* ABI v6, padded batches (apply only): number of REAL rows r = rdev[0]: 1/R becomes 1/r and the
                                 padding rows' gradients are forced to zero; or NULL */
} gps_norm_bwd_task;
size_t gps_norm_tree_floats(int64_t R, int d);
int gps_norm_sync_words(void);
/* The `sync` words of every in-launch reduction (gps_norm_*, gps_gemm_panel_stats) must be zero
 * at entry and are zero again at exit.  A launch that finds a non-zero counter TRAPS (csrc/col_tree.hpp) instead of
 * publishing statistics of incomplete records.  gps_sync_reset: re-zero a buffer after a failed launch;
 * gps_sync_nonzero: *count += number of non-zero words (device uint32; a between-launches self-check). */
int gps_sync_reset(uint32_t* sync, size_t words, gps_stream_t stream);
int gps_sync_nonzero(const uint32_t* sync, size_t words, uint32_t* count, gps_stream_t stream);
int gps_norm_fwd(int n, const gps_norm_fwd_task* tasks, int d, float* ws, size_t ws_floats, uint32_t* sync,
                 gps_stream_t stream);
int gps_norm_bwd_partial(int n, const gps_norm_bwd_task* tasks, int d, float* ws, size_t ws_floats, uint32_t* sync,
                         gps_stream_t stream);
int gps_norm_bwd_apply(int n, const gps_norm_bwd_task* tasks, int d, float* ws, size_t ws_floats, uint32_t* sync,
                       gps_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Random-walk structural encoding: out[v][k - kmin] = (P^k)[v][v] * k^(space_dim/2), P = D^-1 A per graph,
 * for every graph of a batch in one launch (one workgroup per graph, P in LDS up to gps_rwse_lds_nodes()
 * nodes, else in `scratch` at float offset scratch_off[g], 3 n^2 floats per such graph).
 * Replaces graphgps/transform/posenc_stats.py:184-230 (get_rw_landing_probs; CPU, per graph).
 * `rowptr_src`/`dst_by_src` = the by-source CSR of gps_graph_index_build, `ptr` the graph offsets.
 * ------------------------------------------------------------------------------------- */
int gps_rwse_lds_nodes(void);
int gps_rwse(const int32_t* rowptr_src, const int32_t* dst_by_src, const int32_t* ptr, int64_t B, int64_t N,
             int kmin, int kmax, float space_dim, float* scratch, const int64_t* scratch_off, float* out,
             gps_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Optimizer side of the step: gradient-norm clip + AdamW over a flat fp32 parameter arena.
 * Replaces torch.nn.utils.clip_grad_norm_(model.parameters(), cfg.optim.clip_grad_norm_value)
 * followed by optimizer.step() (graphgps/train/custom_train.py:33-37) for the optimizer
 * register_optimizer('adamW') builds (graphgps/optimizer/extra_optimizers.py:21-24).
 *   p, g, m, v     float arenas of identical layout (parameters, gradients, exp_avg, exp_avg_sq)
 *   chunk_off[c]   int64 element offset of chunk c; chunk_len[c] <= gps_optim_chunk();
 *   chunk_param[c] index of the parameter the chunk belongs to (chunks never straddle one)
 *   active[param]  0 = the parameter got no gradient this step: skipped (no decay, no moments),
 *                  and excluded from the norm, like torch skips p.grad is None
 *   hyper          DEVICE double[8]: lr, beta1, beta2, eps, weight_decay, max_norm (<= 0: no clip),
 *                  step (optimizer steps taken; incremented by the call), total_norm (written)
 *   pstep          DEVICE float[n_params]: torch's per-parameter `step` (advances only for active
 *                  parameters; drives the bias corrections)
 *   ws             >= n_chunks floats
 * Deterministic (fixed-order reductions), no host sync: replayable from a hipGraph. */
int gps_optim_chunk(void);
int gps_adamw_step(float* p, const float* g, float* m, float* v, const int64_t* chunk_off,
                   const int32_t* chunk_len, const int32_t* chunk_param, const uint8_t* active,
                   int64_t n_chunks, double* hyper, float* pstep, float* ws, gps_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* GPS_HIP_H */

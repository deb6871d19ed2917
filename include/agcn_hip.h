/* agcn_hip.h -- C ABI of libagcn_hip.so: the 2s-AGCN hot path (unit_gcn / unit_tcn / TCN_GCN_unit forward and
 * backward, and the clip+SGD tail of the training step) as hand-written gfx950 (MI355X, CDNA4) HIP kernels.
 *
 * The reference (cheneeheng/2s-AGCN) is pure Python/PyTorch and has no FFI for this path; every entry point below
 * replaces the stock ATen operators that one line range of the reference launches (cited per function as
 * agcn.py:<lines> = model/architecture/aagcn/agcn.py, processor.py = utils/processor.py).  INTEGRATION.md shows the
 * ctypes binding a maintainer of the reference would add.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer to contiguous fp32 owned by the caller (PyTorch's caching allocator in our
 *    host code); the library never allocates, frees or keeps device memory; scratch/workspace is passed in and its
 *    size is given by the matching *_workspace / *_scratch_bytes / *_num_* query;
 *  - activations are (N, C, T, V) row-major ("NCHW"), N = batch*persons, V <= 32 joints, P = T*V;
 *  - `stream` is a hipStream_t passed as void*; all work is enqueued on it, nothing synchronises the device, the
 *    functions are re-entrant and hold no mutable global state (forward and autograd-backward threads may call
 *    concurrently).  What the library does keep is set-once and idempotent: the AGCN_* environment switches are read
 *    on first use and never change afterwards, and the per-(kernel, device) "dynamic LDS limit raised" flags are
 *    atomics whose lost race only repeats an idempotent hipFuncSetAttribute;
 *  - return value: 0 = ok, AGCN_ERR_* (negative) = argument/shape problem detected on the host before any launch,
 *    positive = hipError_t of a failed launch.  Nothing throws or aborts.
 *  - results are bitwise reproducible run to run (no float atomics; all cross-workgroup sums go through slabs that
 *    are added in a fixed order).
 */
#ifndef AGCN_HIP_H
#define AGCN_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AGCN_OK 0
#define AGCN_ERR_ARG (-1)         /* null pointer / non-positive size / V > 32 */
#define AGCN_ERR_WORKSPACE (-2)   /* workspace smaller than the query says */
#define AGCN_ERR_UNSUPPORTED (-3) /* taps not in {1,9}, stride not in {1,2}, element count not a multiple of 4 ... */

/* load-time checks */
int agcn_version(void);          /* 100 = 0.1.0 */
const char* agcn_arch(void);     /* "gfx950" */

/* diagnostics (never used by a compute path): the kernel instantiation the calling thread's last contraction launch
 * enqueued, and the arithmetic mode of the channel contractions ("bf16x6" = fp32-equivalent 6-product bf16 split,
 * "f32" = exact-f32 MFMA, "bf16x3"), read once per process from the environment variable AGCN_GEMM */
const char* agcn_last_kernel(void);
const char* agcn_gemm_mode(void);
/* arithmetic of the aggregate+project chain: "f16x3" (two fp16 pieces per operand, three products, operands range-scaled
 * by the tensor maximum) in the default mode, "bf16x6" with AGCN_CHAIN_F16X3=0, else agcn_gemm_mode() */
const char* agcn_chain_mode(void);

/* ---- tile geometry queries (sizes of the partial slabs below) ---------------------------------------------------- */
int agcn_conv_tile_frames(int V, int T_out);     /* frames per position tile of the contraction kernels (256/V) */
int agcn_conv_num_tiles(int V, int T_out);       /* default tiles per sample */
int agcn_conv_stats_tiles(int Cin, int Cout, int T_out, int V, int taps, int stride);   /* agcn_conv_fwd: stats_part has N * this many slots */
int agcn_scores_num_tiles(int V, int T);         /* tiles per sample of the adjacency-score kernels */
int agcn_dadj_num_slots(int C, int V, int T);    /* slots per (sample, subset) of the adjacency-gradient slab */

/* ---- channel contractions: unit_tcn's Conv2d((k,1), stride (s,1), pad ((k-1)/2,0)) and every 1x1 Conv2d ------------
 * replaces: nn.Conv2d forward/backward at agcn.py:40-41,49 (unit_tcn.conv), :66-68,99-100 (conv_a/conv_b),
 *           :73 (down conv), and the kernel_size=1 residual unit_tcn at :125.   taps in {1,9}, stride in {1,2}.
 * w: (Cout, Cin, taps, 1) as in the reference state_dict.  stats_part (optional, may be NULL): per-channel partial
 * (sum, sum of squares) of y, layout [N*agcn_conv_num_tiles][2][Cout], consumed by agcn_bn_stats_finalize. */
size_t agcn_conv_workspace(int Cin, int Cout, int T, int V, int taps, int stride); /* fwd and bwd_data (packed weights) */
int agcn_conv_fwd(const float* x, const float* w, const float* bias, float* y, float* stats_part, void* workspace,
                  size_t workspace_bytes, int N, int Cin, int Cout, int T, int V, int taps, int stride, void* stream);
/* dx (+)= conv^T(dy) [+ add1*(mask1>0)] [+ add2*(mask2>0)]; add/mask are dx-shaped or NULL (mask NULL = no masking);
 * they fold the ReLU-masked identity-residual gradients (agcn.py:108-109,128-129) into the epilogue. */
int agcn_conv_bwd_data(const float* dy, const float* w, float* dx, int accumulate, const float* add1,
                       const float* mask1, const float* add2, const float* mask2, void* workspace,
                       size_t workspace_bytes, int N, int Cin, int Cout, int T, int V, int taps, int stride,
                       void* stream);
size_t agcn_conv_bwd_weight_workspace(int N, int Cin, int Cout, int T, int V, int taps, int stride);
int agcn_conv_bwd_weight(const float* dy, const float* x, float* dw, void* workspace, size_t workspace_bytes, int N,
                         int Cin, int Cout, int T, int V, int taps, int stride, void* stream);

/* ---- unit_gcn: graph aggregation fused with the conv_d projection --------------------------------------------------
 * replaces agcn.py:103-105:  y = sum_i conv_d[i]( matmul(x.view(N,C*T,V), A^_i) )   (three bmm + three 1x1 convs + adds)
 * adj: (N,3,V,V) = A^ from agcn_adjacency_fwd ; wcat: (Cout, 3*C) = [Wd_0 | Wd_1 | Wd_2] ; bias: sum of the three
 * conv_d biases (or NULL). */
size_t agcn_gcn_workspace(int C, int Cout, int T, int V);   /* aggregate fwd / bwd_data / dadj (packed weights) */
int agcn_gcn_stats_tiles(int C, int Cout, int T, int V);    /* stats_part slots per sample of the tile-per-workgroup kernels */
int agcn_gcn_stats_slots(int N, int C, int Cout, int T, int V);   /* TOTAL stats_part slots agcn_gcn_aggregate_project_fwd writes (what the caller allocates: the persistent kernel's count depends on N) */
int agcn_gcn_aggregate_project_fwd(const float* x, const float* adj, const float* wcat, const float* bias, float* y,
                                   float* stats_part, void* workspace, size_t workspace_bytes, int N, int C, int Cout,
                                   int T, int V, void* stream);
/* ---- operand maxima for the split-fp16 ("f16x3") temporal convolutions ------------------------------------------------
 * The f16x3 kernels scale their streamed operand by a power of two derived from the tensor's max |x|.  The
 * _ex variants let the kernel that PRODUCES a tensor leave that maximum behind (4-byte device scalar) and the kernel that
 * consumes it skip its own pass over the tensor; with NULL they behave exactly like the plain entry points. */
int agcn_bn_act_fwd_ex(const float* y1, const float* scale1, const float* shift1, const float* r, const float* scale2,
                       const float* shift2, float* out, unsigned* sign_bits, float* absmax_out, int N, int C, int P,
                       int res_mode, int relu, void* stream);
int agcn_bn_bwd_apply_ex(const float* part, int nrows, double count, float param_grad_scale, const float* dout,
                         const void* mask, int mask_bits, const float* y1, const float* gamma1, const float* mean1,
                         const float* invstd1, const float* y2, const float* gamma2, const float* mean2,
                         const float* invstd2, float* coef, float* dy1, float* dgamma1, float* dbeta1, float* dy2,
                         float* dgamma2, float* dbeta2, float* absmax1_out, int N, int C, int P, void* stream);
/* the aggregate+project chain runs on f16x3 too (agcn_chain_mode): x_absmax = max |x| (agcn_bn_act_fwd_ex of the previous
 * unit), dy_absmax = max |dy| (agcn_bn_bwd_apply_ex), dtp_absmax = max |dtp|; any of them NULL: a streaming pass inside.
 * agcn_gcn_aggregate_project_bwd_data_ex covers both backward-data forms: dtp NULL = the plain one, else the fused one.
 * agcn_gcn_dadj_ex: dy_absmax for the projection H = Wd^T dy, x_absmax (the forward's max |x|) for the reduction of H
 * against x, both on f16x3; NULL: a streaming pass inside (same bits either way). */
int agcn_gcn_aggregate_project_fwd_ex(const float* x, const float* adj, const float* wcat, const float* bias, float* y,
                                      float* stats_part, void* workspace, size_t workspace_bytes, int N, int C, int Cout,
                                      int T, int V, const float* x_absmax, void* stream);
int agcn_gcn_aggregate_project_bwd_data_ex(const float* dy, const float* adj, const float* wcat, const float* dtp,
                                           const float* w2, int K2, float* dx, int accumulate, const float* add1,
                                           const float* mask1, const float* add2, const float* mask2, int mask_bits,
                                           void* workspace, size_t workspace_bytes, int N, int C, int Cout, int T, int V,
                                           const float* dy_absmax, const float* dtp_absmax, void* stream);
int agcn_gcn_dadj_ex(const float* dy, const float* wcat, const float* x, float* dadj_part, void* workspace,
                     size_t workspace_bytes, int N, int C, int Cout, int T, int V, const float* dy_absmax,
                     const float* x_absmax, void* stream);
int agcn_adjacency_bwd_scores_ex(const float* tp, const float* dS, float* dtp, float* dbpart, void* scratch, float* db,
                                 float* dtp_absmax_out, int N, int Ci, int T, int V, void* stream);
/* weight gradients: with BOTH operand maxima given the tap-free gradients (1x1, stride 1, Cin a multiple of 64; the
 * projection gradient with C a multiple of 64, its aggregation included) run on f16x3; NULL: bf16x6 as the plain entry
 * points (no pass inside).  The 9-tap gradient (rows and channels multiples of 64) runs on f16x3 either way: its
 * transposing pre-pass writes the range-scaled fp16 planes and takes a missing maximum with a reduction pass of its own. */
int agcn_conv_bwd_weight_ex(const float* dy, const float* x, float* dw, void* workspace, size_t workspace_bytes, int N,
                            int Cin, int Cout, int T, int V, int taps, int stride, const float* dy_absmax,
                            const float* x_absmax, void* stream);
int agcn_gcn_project_bwd_weight_ex(const float* dy, const float* x, const float* adj, float* dwcat, void* workspace,
                                   size_t workspace_bytes, int N, int C, int Cout, int T, int V, const float* dy_absmax,
                                   const float* x_absmax, void* stream);
int agcn_absmax(const float* x, long n, float* out, void* stream);   /* *out = max |x|: the pass the f16x3 kernels run when given no maximum */
int agcn_conv_fwd_ex(const float* x, const float* w, const float* bias, float* y, float* stats_part, void* workspace,
                     size_t workspace_bytes, int N, int Cin, int Cout, int T, int V, int taps, int stride,
                     const float* x_absmax, void* stream);
int agcn_conv_bwd_data_ex(const float* dy, const float* w, float* dx, int accumulate, const float* add1,
                          const float* mask1, const float* add2, const float* mask2, void* workspace,
                          size_t workspace_bytes, int N, int Cin, int Cout, int T, int V, int taps, int stride,
                          const float* dy_absmax, void* stream);

/* ---- unit_gcn forward of the first layer (1..4 input channels) ----------------------------------------------------------
 * replaces agcn.py:103-105 AND the `down` convolution (agcn.py:74-77, 108) for in_channels = 3 in one pass over x:
 * ypre = bias + sum_i Wd_i (x . adj_i), dpre = bdown + Wdown x (wdown (Cout, C) row-major or NULL), and the per-tile
 * (sum, sumsq) partials of both ([N * agcn_gcn_first_tiles][2][Cout], NULL to skip) for agcn_bn_stats_finalize. */
int agcn_gcn_first_supported(int C, int Cout, int V);
int agcn_gcn_first_tiles(int T, int V);
int agcn_gcn_first_fwd(const float* x, const float* adj, const float* wcat, const float* bias, const float* wdown,
                       const float* bdown, float* ypre, float* ystats, float* dpre, float* dstats, int N, int C, int Cout,
                       int T, int V, void* stream);

/* ---- BN-folded inference (eval mode) ---------------------------------------------------------------------------------
 * replaces, for model.eval() under no_grad, the whole of unit_gcn.forward after the adjacency (agcn.py:103-109) and of
 * unit_tcn.forward + the TCN_GCN_unit tail (agcn.py:48-50, 127-129): the caller folds every BatchNorm into the weights
 * and bias of the contraction in front of it (w' = w * gamma/sqrt(var+eps), b' = (b - mean) * gamma/sqrt(var+eps) + beta),
 * the residual add and the ReLU ride in the store epilogue.
 *   agcn_gcn_unit_infer: y = act( bias + sum_i W_i (x . adj_i) [+ res] [+ W2 . x2] ); res (N,Cout,T,V) or NULL (the
 *     identity `down`); x2 (N,K2,T,V) with w2 (Cout,K2) row-major or both NULL (the folded conv `down`, K2 % 32 == 0)
 *   agcn_conv9_infer:    y = act( bias + conv9x1(x; w (Cout,Cin,9,1), stride) [+ res] ); res (N,Cout,T_out,V) or NULL
 * Both return AGCN_ERR_UNSUPPORTED where only the exact-f32 kernels apply (C < 32, AGCN_GEMM=f32): run the unfused
 * passes then.  Workspace of agcn_conv9_infer: agcn_conv_workspace. */
size_t agcn_gcn_unit_infer_workspace(int C, int Cout, int K2, int T, int V);
int agcn_gcn_unit_infer(const float* x, const float* adj, const float* wcat, const float* bias, const float* res,
                        const float* x2, const float* w2, int K2, int relu, float* y, void* workspace,
                        size_t workspace_bytes, int N, int C, int Cout, int T, int V, void* stream);
int agcn_conv9_infer(const float* x, const float* w, const float* bias, const float* res, int relu, float* y,
                     void* workspace, size_t workspace_bytes, int N, int Cin, int Cout, int T, int V, int stride,
                     void* stream);

/* mask_bits = 1: mask1/mask2 are sign bit masks of agcn_bn_act_fwd (cast to const float*), 0: fp32 tensors (> 0 passes) */
int agcn_gcn_aggregate_project_bwd_data(const float* dy, const float* adj, const float* wcat, float* dx,
                                        int accumulate, const float* add1, const float* mask1, const float* add2,
                                        const float* mask2, int mask_bits, void* workspace, size_t workspace_bytes, int N, int C,
                                        int Cout, int T, int V, void* stream);
/* the same plus the 1x1 term of the adaptive branch in one pass: dx (+)= ... + w2^T dtp, dtp (N, K2, T, V), w2 (K2, C)
 * row-major = the stacked conv_a/conv_b weights (reference agcn.py:99-100 differentiated).  Chained (bf16x6) path only:
 * agcn_gcn_bwd_data_fused_supported() tells; workspace as agcn_gcn_workspace(C, Cout, T, V). */
int agcn_gcn_bwd_data_fused_supported(int C, int Cout, int V);
int agcn_gcn_aggregate_project_bwd_data_fused(const float* dy, const float* adj, const float* wcat, const float* dtp,
                                              const float* w2, int K2, float* dx, int accumulate, const float* add1,
                                              const float* mask1, const float* add2, const float* mask2, int mask_bits,
                                              void* workspace, size_t workspace_bytes, int N, int C, int Cout, int T,
                                              int V, void* stream);
size_t agcn_gcn_project_bwd_weight_workspace(int N, int C, int Cout, int T, int V);
int agcn_gcn_project_bwd_weight(const float* dy, const float* x, const float* adj, float* dwcat, void* workspace,
                                size_t workspace_bytes, int N, int C, int Cout, int T, int V, void* stream);
/* dadj_part: (N, 3, agcn_dadj_num_slots, V, V) partial adjacency gradients, summed by agcn_adjacency_bwd_softmax */
int agcn_gcn_dadj(const float* dy, const float* wcat, const float* x, float* dadj_part, void* workspace,
                  size_t workspace_bytes, int N, int C, int Cout, int T, int V, void* stream);

/* ---- adaptive adjacency ------------------------------------------------------------------------------------------------
 * replaces agcn.py:95,99-102:  A^_i = softmax_{dim -2}( theta_i^T phi_i / (Ci*T) ) + A_i + PA_i
 * tp: (N, 6*Ci, T, V), rows [theta_0|phi_0|theta_1|phi_1|theta_2|phi_2] (one agcn_conv_fwd with the six 1x1 weights
 * stacked).  A: (3,V,V) constant graph (NULL for the AAGCN form), PA: (3,V,V) parameter, alpha: 1 float or NULL
 * (AAGCN: A^ = PA + alpha*P, aagcn.py:172-173).  spart: scratch (N,3,agcn_scores_num_tiles,V,V).
 * Outputs P (softmax, kept for the backward) and adj, both (N,3,V,V). */
int agcn_adjacency_fwd(const float* tp, const float* A, const float* PA, const float* alpha, float* spart, float* P,
                       float* adj, int N, int Ci, int T, int V, void* stream);
/* The same WITHOUT the theta/phi tensor in HBM (SURVEY Appendix A "kernel A"): x (N,C,T,V) is read once per
 * workgroup tile, [theta;phi] = wab . x + bab (wab: (6*Ci, C) stacked conv_a/conv_b weights, rows as tp above) lives in
 * accumulators/LDS only and is reduced into spart on the spot; then the finalize above.  _bwd_scores recomputes the
 * tile and writes dtp / db exactly as agcn_adjacency_bwd_scores does from a stored tp.  tp_out (may be NULL): if
 * given, the forward also leaves theta/phi (N, 6*Ci, T, V) in HBM as a by-product (coalesced rows from the LDS tile)
 * for a backward that prefers re-reading to recomputing (measured faster on MI355X).  workspace: packed split
 * weights, agcn_adjacency_fused_workspace(C, Ci) bytes.  agcn_adjacency_fused_supported: Ci in {16,32,64}, bf16x6
 * arithmetic (AGCN_GEMM default); otherwise use agcn_conv_fwd + agcn_adjacency_fwd. */
int agcn_adjacency_fused_supported(int C, int Ci, int T, int V);
size_t agcn_adjacency_fused_workspace(int C, int Ci);
int agcn_adjacency_fused_fwd(const float* x, const float* wab, const float* bab, const float* A, const float* PA,
                             const float* alpha, float* tp_out, float* spart, float* P, float* adj, void* workspace,
                             size_t workspace_bytes, int N, int C, int Ci, int T, int V, void* stream);
/* same; x_absmax_out (optional, 4 bytes) receives max |x| as a by-product of the pass that reads all of x, for the f16x3
 * aggregate+project chain that reads x next (agcn_gcn_aggregate_project_fwd_ex); x_absmax_in (optional): max |x| where the
 * producer of x left it behind (the persistent kernel of the Ci <= 32 layers scales x by it; without it a reduction pass
 * takes the maximum first) */
int agcn_adjacency_fused_fwd_ex(const float* x, const float* wab, const float* bab, const float* A, const float* PA,
                                const float* alpha, float* tp_out, float* spart, float* P, float* adj, float* x_absmax_out,
                                const float* x_absmax_in, void* workspace, size_t workspace_bytes, int N, int C, int Ci, int T,
                                int V, void* stream);
int agcn_adjacency_fused_bwd_scores(const float* x, const float* wab, const float* bab, const float* dS, float* dtp,
                                    float* dbpart, void* scratch, float* db, void* workspace, size_t workspace_bytes,
                                    int N, int C, int Ci, int T, int V, void* stream);
/* dadj = sum of slots; dS = alpha*P*(dadj - sum_u P*dadj)/(Ci*T); dPA = sum_n dadj; dalpha_part (N*3) or NULL */
int agcn_adjacency_bwd_softmax(const float* dadj_part, const float* P, const float* alpha, float* dadj, float* dS,
                               float* dPA, float* dalpha_part, int N, int Ci, int T, int V, int nslots, void* stream);
/* dtp from dS; db (6*Ci) = bias gradients of the stacked conv_a/conv_b; dbpart: scratch (N*tiles, 6*Ci);
 * scratch: agcn_colsum_scratch_bytes(6*Ci) bytes */
int agcn_adjacency_bwd_scores(const float* tp, const float* dS, float* dtp, float* dbpart, void* scratch, float* db,
                              int N, int Ci, int T, int V, void* stream);

/* ---- BatchNorm2d (train/eval) + residual + ReLU ---------------------------------------------------------------------
 * replaces agcn.py:43,49 (unit_tcn.bn), :74 (down BN), :79,107-109 (unit_gcn.bn, += down(x), relu), :128-129
 * (TCN_GCN_unit: + residual, relu).  eps 1e-5, momentum 0.1, biased variance for normalisation, unbiased for
 * running_var -- the nn.BatchNorm2d defaults the reference uses. */
size_t agcn_colsum_scratch_bytes(int W);
int agcn_colsum(const float* X, int nslots, int W, void* scratch, float* out, void* stream);
int agcn_bn_stats_finalize(const float* stats_part, int nslots, int C, double count, const float* gamma,
                           const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                           void* scratch /* agcn_colsum_scratch_bytes(2*C) */, float* mean, float* invstd,
                           float* scale, float* shift, void* stream);
int agcn_bn_eval_coeff(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                       float eps, int C, float* scale, float* shift, void* stream);
/* out = act(scale1[c]*y1 + shift1[c] + res); res_mode 0: none, 1: r, 2: scale2[c]*r + shift2[c]; N*C*P % 4 == 0.
 * sign_bits (optional, may be NULL): ceil(N*C*P/32) words, bit e of word w = (out[32*w + e] > 0): the ReLU mask the
 * backward needs, 32x smaller than `out`. */
int agcn_bn_act_fwd(const float* y1, const float* scale1, const float* shift1, const float* r, const float* scale2,
                    const float* shift2, float* out, unsigned* sign_bits, int N, int C, int P, int res_mode, int relu,
                    void* stream);
/* train-mode backward of out = relu(bn1(y1) [+ bn2(y2)] [+ identity]); mask (NULL: no ReLU) is either the fp32 `out`
 * tensor (mask_bits = 0: positive elements pass) or the sign_bits of agcn_bn_act_fwd (mask_bits = 1);
 * part: scratch N*C*3 floats, coef: scratch 6*C floats */
/* the two stages of agcn_bn_bwd, exposed so that a synchronised BatchNorm (reference utils/processor.py:295) can
 * all-reduce the per-channel sums between them: part is (N*C*3); apply sums `nrows` rows of it per channel, uses
 * `count` elements per channel and scales dgamma/dbeta by param_grad_scale. */
int agcn_bn_bwd_reduce(const float* dout, const void* mask, int mask_bits, const float* y1, const float* y2, float* part,
                       int N, int C, int P, void* stream);
int agcn_bn_bwd_apply(const float* part, int nrows, double count, float param_grad_scale, const float* dout,
                      const void* mask, int mask_bits, const float* y1, const float* gamma1, const float* mean1,
                      const float* invstd1, const float* y2, const float* gamma2, const float* mean2, const float* invstd2, float* coef,
                      float* dy1, float* dgamma1, float* dbeta1, float* dy2, float* dgamma2, float* dbeta2, int N, int C,
                      int P, void* stream);
int agcn_bn_bwd(const float* dout, const void* mask, int mask_bits, const float* y1, const float* gamma1,
                const float* mean1, const float* invstd1, const float* y2, const float* gamma2, const float* mean2, const float* invstd2,
                float* part, float* coef, float* dy1, float* dgamma1, float* dbeta1, float* dy2, float* dgamma2,
                float* dbeta2, int N, int C, int P, void* stream);

/* ---- AAGCN attention gates (config 4) ---------------------------------------------------------------------------------
 * replaces the full-tensor passes of aagcn.py:59-116, 268-270 (three "mean -> tiny net -> sigmoid -> y*s + y" gates):
 * agcn_stc_row_reduce: one pass over y (and optionally an elementwise factor g) giving, per (n, c) row,
 *     out_t[t] = scale_t * sum_v wv[n][v] * y*g   and/or   out_v[v] = scale_v * sum_t wt[n or row][t] * y*g
 *   (forward: mean_t y, mean_v y*(1+se_s); backward: the two weighted sums of dout*y in one pass, and sum_t dmv*y);
 * agcn_stc_apply: out = y * a_s[n,v] * a_t[n,t] * a_c[n,c] with a_* = 1 + sigmoid gate, (N,V) / (N,T) / (N,C);
 * agcn_stc_bwd_apply: dy = dout * a_s a_t a_c + dmv[n,c,t] * a_s[n,v] + dms[n,c,v] (the gradients that reach y through
 *   the two means folded into the same pass).  The gate networks themselves: agcn_gate_conv_* / agcn_linear_* below. */
int agcn_stc_row_reduce(const float* y, const float* g, const float* wv, const float* wt, int wt_per_row, float* out_t,
                        float* out_v, float scale_t, float scale_v, int N, int C, int T, int V, void* stream);
int agcn_stc_apply(const float* y, const float* a_s, const float* a_t, const float* a_c, float* out, int N, int C, int T,
                   int V, void* stream);
/* same; absmax_out (optional, 4 bytes) receives max |out| for the f16x3 temporal convolution that reads the gated tensor */
int agcn_stc_apply_ex(const float* y, const float* a_s, const float* a_t, const float* a_c, float* out, float* absmax_out,
                      int N, int C, int T, int V, void* stream);
int agcn_stc_bwd_apply(const float* dout, const float* a_s, const float* a_t, const float* a_c, const float* dmv,
                       const float* dms, float* dy, int N, int C, int T, int V, void* stream);

/* ---- the small operators around the unit stack, deterministic (csrc/small_ops.hip) -----------------------------------
 * Every sum in a fixed order, no vendor library: the model path is bitwise reproducible from process to process
 * (MIOpen's Conv1d for the temporal gate was not: DESIGN.md section 3).
 *
 * data_bn  replaces agcn.py:143,163-165 (x.permute(0,4,3,1,2).view(N, M*V*C, T) -> BatchNorm1d -> view/permute ->
 *   (N*M, C, T, V)) and aagcn.py forward_preprocess.  x is the model input (N, C, T, V, M); channel ch = (m*V+v)*C + c.
 *   agcn_data_bn_stats: part [N][2][C*V*M] partial (sum, sumsq), finalised by agcn_bn_stats_finalize(part, N, C*V*M,
 *   count = N*T, ...); agcn_data_bn_apply: out[(n*M+m), c, t, v] = x[n,c,t,v,m]*scale[ch] + shift[ch];
 *   agcn_data_bn_bwd_reduce: part [N][2][C*V*M] partial (sum dy, sum dy*xhat) (agcn_colsum over N slots);
 *   agcn_data_bn_bwd_apply: dx from the GLOBAL sums [2][C*V*M] and the element count per channel behind them. */
int agcn_data_bn_stats(const float* x, float* part, int N, int C, int T, int V, int M, void* stream);
int agcn_data_bn_apply(const float* x, const float* scale, const float* shift, float* out, int N, int C, int T, int V,
                       int M, void* stream);
int agcn_data_bn_bwd_reduce(const float* dy, const float* x, const float* mean, const float* invstd, float* part, int N,
                            int C, int T, int V, int M, void* stream);
int agcn_data_bn_bwd_apply(const float* dy, const float* x, const float* gamma, const float* mean, const float* invstd,
                           const float* sums, double count, float* dx, int N, int C, int T, int V, int M, void* stream);
/* global average pool  replaces agcn.py:179-181 (x.view(N, M, C, -1).mean(3).mean(1)): x (N*M, C, P) -> pooled (N, C);
 * rowmean: scratch of N*M*C floats.  agcn_pool_bwd: dx[(n*M+m), c, p] = dpooled[n, c] / (M*P). */
int agcn_pool_fwd(const float* x, float* rowmean, float* pooled, int N, int M, int C, int P, void* stream);
int agcn_pool_bwd(const float* dpooled, float* dx, int N, int M, int C, int P, void* stream);
/* small Linear  replaces agcn.py:183 (self.fc) and aagcn.py:111-116 (fc1c -> ReLU -> fc2c -> sigmoid):
 * out[n, o] = act(b[o] + sum_k in[n, k] w[o, k]); act 0 identity, 1 ReLU, 2 "1 + sigmoid" (the gate's y*s + y factor).
 * agcn_linear_bwd: dpre (N, O) scratch = dout * act'(out); din (N, K, may be NULL), dw (O, K), db (O, may be NULL). */
int agcn_linear_fwd(const float* in, const float* w, const float* b, float* out, int N, int K, int O, int act,
                    void* stream);
int agcn_linear_bwd(const float* dout, const float* out, const float* in, const float* w, float* dpre, float* din,
                    float* dw, float* db, int N, int K, int O, int act, void* stream);
/* gate convolution  replaces aagcn.py:72-76 / 92-96 (Conv1d(C -> 1, Ks, padding (Ks-1)/2) + sigmoid on (N, C, L)):
 * a[n, l] = 1 + sigmoid(b + sum_c sum_k w[c, k] in[n, c, l + k - pad]); Ks odd.
 * agcn_gate_conv_bwd: da = gradient w.r.t. a; dpre (N, L) scratch; din (N, C, L), dw (C, Ks), db (1). */
int agcn_gate_conv_fwd(const float* in, const float* w, const float* b, float* a, int N, int C, int L, int Ks,
                       void* stream);
int agcn_gate_conv_bwd(const float* da, const float* a, const float* in, const float* w, float* dpre, float* din,
                       float* dw, float* db, int N, int C, int L, int Ks, void* stream);

/* ---- training-step tail on one flat parameter buffer ----------------------------------------------------------------
 * replaces processor.py:698 (clip_grad_norm_(params, 1.0)) + :703 (optimizer.step() of optim.SGD(momentum, nesterov,
 * weight_decay), :395-401).  grad_scale multiplies the gradient first (1/world_size after a SUM all-reduce).
 * norm_out: 2 device floats {global grad norm, clip coefficient}.  max_norm <= 0 disables clipping. */
size_t agcn_sgd_step_workspace(long n);
int agcn_sgd_step(float* param, const float* grad, float* momentum_buf, long n, float lr, float momentum,
                  float weight_decay, int nesterov, float max_norm, float grad_scale, int first_step, void* workspace,
                  size_t workspace_bytes, float* norm_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* AGCN_HIP_H */

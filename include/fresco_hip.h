/*
 * fresco_hip.h -- C ABI of libfresco_hip.so: the MI355X (gfx950 / CDNA4) kernels behind FRESCO's
 * flow-guided attention, feature warp and feature-optimisation hot path.
 *
 * Conventions (every entry point):
 *   - plain device pointers + explicit sizes; no torch / C++ types cross this boundary;
 *   - returns FRESCO_OK (0) or a negative FRESCO_E* code; nothing is thrown, nothing printed;
 *   - never allocates: scratch memory is passed in by the caller, sized by the matching
 *     *_workspace_bytes() query (host-only, no GPU needed);
 *   - asynchronous on `stream` (a hipStream_t passed as void*; NULL = the null stream);
 *   - no global state (except the opt-in fresco_prof_* timing log, and FRESCO_OPT_SV read once from the
 *     environment): concurrent calls on different streams / devices with disjoint buffers are safe.
 *
 * Reference interface each entry point replaces (paths relative to the FRESCO tree):
 *   src/diffusion_hacked.py  = DH,  src/flow_utils.py = FU,  src/utils.py = UT,
 *   src/ebsynth/deps/gmflow/gmflow/geometry.py = GEO.
 *
 * "half" below is IEEE binary16 (the dtype the SD-1.5 pipeline runs in, run_fresco.py:63-80).
 */
#ifndef FRESCO_HIP_H
#define FRESCO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FRESCO_OK 0
#define FRESCO_EINVAL (-1)      /* null pointer / non-positive size / inconsistent arguments   */
#define FRESCO_EUNSUPPORTED (-2) /* shape outside what the kernels are instantiated for        */
#define FRESCO_EWORKSPACE (-3)  /* workspace too small                                          */
#define FRESCO_ELAUNCH (-4)     /* hipGetLastError() != hipSuccess after a launch              */

/* dtype codes for entry points that accept more than one element type */
#define FRESCO_F16 0
#define FRESCO_F32 1

/* library / build identification: "fresco_hip <version> gfx950" */
const char* fresco_version(void);
/* last HIP error string seen by a FRESCO_ELAUNCH on this thread ("" if none) */
const char* fresco_last_error(void);

/* ------------------------------------------------------------------------------------------
 * Opt-in kernel timing for bench.py's roofline figures (the ONLY process-global state of the
 * library; off by default).  While enabled, the launches tagged below are bracketed by HIP events
 * recorded on the launch stream.  fresco_prof_read synchronises on the recorded events, returns the
 * number of records copied (launch order) and clears the log.
 *   tags[i], dims[4*i..4*i+3], ms[i]:
 *     FRESCO_PROF_ATTN_FLASH : dims = {B*H, Lq, M, D}     FRESCO_PROF_KV_PACK : {groups, H, M, D}
 *     FRESCO_PROF_TEMPORAL   : dims = {chunk*N, HW, H, D}
 * ------------------------------------------------------------------------------------------ */
#define FRESCO_PROF_ATTN_FLASH 1
#define FRESCO_PROF_KV_PACK 2
#define FRESCO_PROF_TEMPORAL 3
/* feature optimisation, dims = {B, C, hw, 0}: */
#define FRESCO_PROF_OPT_TSIGN 4
#define FRESCO_PROF_OPT_TGRAD 5
#define FRESCO_PROF_OPT_COLNORM 6
#define FRESCO_PROF_OPT_GRAM 7
#define FRESCO_PROF_OPT_SV 8
#define FRESCO_PROF_OPT_ADAM 9
#define FRESCO_PROF_LINEAR 10 /* dims = {M, N, K, nw} */
#define FRESCO_PROF_ATTN_F32 11 /* dims = {B, Lq, Lk, D} */
#define FRESCO_PROF_FN_GEMM 12  /* dims = {M, N, K, kernel height (0: linear layer)} */
int fresco_prof_enable(int capacity);
int fresco_prof_disable(void);
int fresco_prof_read(int max_records, int* tags, int* dims, float* ms);

/* ------------------------------------------------------------------------------------------
 * (a2 + a3)  Dense attention with shared / per-batch keys  -- replaces the two
 * F.scaled_dot_product_attention calls at DH:281-285 (spatial-guided) and DH:303-305
 * (efficient cross-frame), the K/V row selection + repeat at DH:225-247, and the head
 * split / merge views at DH:250-254, 371.
 *
 *   out[b, l, h*D + d] = sum_m softmax_m( scale * <q[b,l,h,:], K[g,m,h,:]> + diag_bias*[l==m] ) * V[g,m,h,d]
 *
 *   q   : (B, Lq, H*D) half, row-major (what attn.to_q returns)
 *   k,v : row-major matrices of H*D-wide half rows.  Key group g (0 <= g < n_groups) uses the M
 *         rows   g*group_rows + (kv_rows ? kv_rows[m] : m),  m = 0..M-1.
 *         Query batch b attends to group  g = b / (B / n_groups).
 *           cross-frame : n_groups = unet_chunk_size, group_rows = N*HW, kv_rows = flat indices of
 *                         the True entries of controller.attn_mask (N,HW), or NULL with M = HW for
 *                         "every frame uses frame 0" (former_frame_index, DH:227).
 *           spatial     : n_groups = B, group_rows = HW, kv_rows = NULL, M = HW,
 *                         q = to_q(ref), k = to_k(ref), v = current query, scale = 0.2/sqrt(D).
 *   out : (B, Lq, H*D) half.
 *   D must be a multiple of 8, 8 <= D <= 128.  Softmax in fp32, P and V in half, accumulation fp32.
 *   Numerics: the exponent scale  scale*log2(e)  is folded into the fp16 query (one rounding) only for
 *   queries whose logits are provably small (scale*log2e*|q|*max|k| <= 16, a per-wavefront decision from
 *   the key norms the pack pass records); otherwise scores are scaled in fp32.  Logits are assumed to stay
 *   below 6e4 in log2 units.
 *   Workspace: packed K / V^T images of every key group and head plus one float per 64-key tile.
 * ------------------------------------------------------------------------------------------ */
size_t fresco_attn_workspace_bytes(int n_groups, int H, int M, int D);

/* ------------------------------------------------------------------------------------------
 * (a1)  Fused linear projections  attn.to_q / to_k / to_v (DH:201, 214-215, 260-261) and to_out[0] (DH:375):
 *           out_j = x W_j^T (+ b_j),   j = 0 .. nw-1,  nw <= 3
 *   x    : (M, K) half, row stride x_ld (elements);  read once for all nw outputs
 *   W_j  : (N, K) half row-major = the weight of nn.Linear j, read where it lives (nothing is stacked or cached:
 *          in-place updates of a module's weight are seen by the next call);  W1 / W2 unused beyond nw
 *   b_j  : (N) half or NULL
 *   out_j: (M, N) half, row stride ld_j (elements); unused outputs NULL
 *   fp32 accumulation, one rounding to half at the end (what the library GEMM behind nn.Linear does).
 *   Supported: K in {320, 640} (SD-1.5 up_blocks.3 / up_blocks.2), N % 64 == 0; anything else returns
 *   FRESCO_EUNSUPPORTED and the caller keeps its own GEMM.
 * ------------------------------------------------------------------------------------------ */
int fresco_linear(const void* x, int64_t x_ld, const void* W0, const void* W1, const void* W2, const void* b0,
                  const void* b1, const void* b2, void* out0, void* out1, void* out2, int64_t ld0, int64_t ld1,
                  int64_t ld2, int nw, int M, int N, int K, void* stream);

/* The same with gathered input rows: problem row m reads x row x_rows[m] (int32, M entries, each inside the x buffer --
 * the caller's contract).  Used for the K / V projection of the tokens the efficient cross-frame pass selects
 * (DH:225-247) when nothing else reads K and V: project only what is gathered. */
int fresco_linear_rows(const void* x, int64_t x_ld, const int32_t* x_rows, const void* W0, const void* W1,
                       const void* W2, const void* b0, const void* b1, const void* b2, void* out0, void* out1,
                       void* out2, int64_t ld0, int64_t ld1, int64_t ld2, int nw, int M, int N, int K, void* stream);


int fresco_attn_fwd(const void* q, const void* k, const void* v, const int32_t* kv_rows,
                    void* out, void* workspace, size_t workspace_bytes,
                    int B, int H, int Lq, int D,
                    int n_groups, int M, int64_t group_rows,
                    float scale, float diag_bias, void* stream);

/* Same with explicit row strides (in halfs, multiples of 8, >= H*D): q row r starts at q + r*q_ld,
 * k / v row r at k + r*kv_ld -- for q, k, v that are column slices of one fused projection output
 * (B, HW, 3*H*D).  `out` is always dense (B, Lq, H*D). */
int fresco_attn_fwd_ld(const void* q, const void* k, const void* v, const int32_t* kv_rows,
                       void* out, void* workspace, size_t workspace_bytes,
                       int B, int H, int Lq, int D,
                       int n_groups, int M, int64_t group_rows,
                       float scale, float diag_bias, int64_t q_ld, int64_t kv_ld, void* stream);

/* Cross-frame pass with the K | V projection of the selected rows FUSED into the key pack (round 6) -- for layer calls
 * whose only reader of K and V is the cross-frame pass (DH:201-247 on the cross-frame-only steps): replaces
 * attn.to_k / attn.to_v on the gathered rows (DH:214-215 restricted to the tokens DH:239-247 keep) + the pack.
 *   x      : hidden states, fp16 rows of K_in features, row r at x + r*x_ld
 *   x_rows : int32 (n_groups * M): key m of group g is row x_rows[g*M + m] of x (rows may repeat; must be in range --
 *            the table is not checked on the device)
 *   Wk, Wv : (H*D, K_in) row-major fp16 weights of the bias-free projections (read where they live, every call)
 *   q, out, workspace (fresco_attn_workspace_bytes(n_groups, H, M, D)), B, H, Lq, D, n_groups, M, scale, q_ld: as
 *   fresco_attn_fwd_ld; batch element b uses key group b / (B / n_groups).
 * K = x[rows] Wk^T and V = x[rows] Wv^T are rounded to fp16 exactly once (as the two-launch path rounds them) and go
 * straight into the packed key image: they never exist in HBM.  Supported: (D, K_in) = (40, 320), (80, 640) with
 * H*D == K_in (SD-1.5's decoder self-attentions); fresco_attn_kvproj_supported says so, anything else returns
 * FRESCO_EUNSUPPORTED and the caller uses fresco_linear_rows + fresco_attn_fwd_ld. */
int fresco_attn_kvproj_supported(int H, int D, int K_in);
int fresco_attn_fwd_kvproj(const void* q, const void* x, int64_t x_ld, const int32_t* x_rows, const void* Wk,
                           const void* Wv, void* out, void* workspace, size_t workspace_bytes, int B, int H, int Lq,
                           int D, int n_groups, int M, int K_in, float scale, int64_t q_ld, void* stream);

/* ------------------------------------------------------------------------------------------
 * (a4)  Temporal-guided (FLATTEN) attention -- replaces DH:309-367: 3 rearrange+gather round
 * trips, the per-pixel N x N masked SDPA and the inverse gather.
 *
 *   for every aligned pixel p, CFG half c, head h, frames f,g in [0,N):
 *     row(f) = fwd_map[f*HW + p]
 *     out[c*N+f, row(f), h, :] = sum_g softmax_g( scale*<q[c*N+f,row(f),h,:], k[c*N+g,row(g),h,:]>
 *                                                 | mask[p,f,g] ) * v[c*N+g,row(g),h,:]
 *   q,k,v,out : (chunk*N, HW, H*D) half;  fwd_map : (N, HW) int64 (a permutation per frame);
 *   mask : (HW, N, N) uint8/bool, non-zero = may attend (diagonal always set, FU:120-131).
 *   D in {8, 16, 32, 40, 64, 80}; chunk*N*HW < 2^31.  N <= 32: MFMA kernel (needs N * (6*(H*D + 8) + 4 + N) bytes
 *   <= 160 KiB of LDS for 16 < N <= 32: every SD-1.5 shape fits); N > 32: vector-ALU kernel while 6*N*H*D + 4*N + N*N
 *   bytes <= 160 KiB; otherwise FRESCO_EUNSUPPORTED.  Row-table entries outside [0, HW) are skipped (no row is read or
 *   written for them; a table that is not a permutation is the caller's bug).  A query whose mask row is all zero
 *   yields NaN, like the reference's softmax over an all -inf row.
 * ------------------------------------------------------------------------------------------ */
int fresco_temporal_attn(const void* q, const void* k, const void* v, const int64_t* fwd_map,
                         const uint8_t* mask, void* out,
                         int chunk, int N, int HW, int H, int D, float scale, void* stream);

/* Row-strided form (q, k, v rows start every q_ld / k_ld / v_ld halfs; out dense). */
int fresco_temporal_attn_ld(const void* q, const void* k, const void* v, const int64_t* fwd_map,
                            const uint8_t* mask, void* out,
                            int chunk, int N, int HW, int H, int D, float scale,
                            int64_t q_ld, int64_t k_ld, int64_t v_ld, void* stream);

/* Multi-GPU: frames are sharded over ranks, the temporal pass is sharded by TRAJECTORY, so that
 * every byte crosses the fabric once and the kernel's HBM traffic shrinks with the world size.  Rank r owns the
 * n_loc frames [f0, f0+n_loc) of both CFG halves and the trajectory range [r*P, (r+1)*P), P = HW / world.
 *   fresco_temporal_pack  : buf[(d*n_loc + fl)*chunk + c][pl][0:3C] = (q | k | v)[c*n_loc + fl][fwd_map[f0+fl][d*P + pl]]
 *                           (q, k, v: local (chunk*n_loc, HW, C) with row strides q_ld / k_ld / v_ld); an all-to-all over
 *                           the leading `world` dimension then leaves on rank r the rows of ALL N frames of its range as
 *                           (N, chunk, P, 3C);
 *   fresco_temporal_attn_packed : the attention on such rows, no row table: qkv (N, chunk, P, 3C), mask (P, N, N) = the
 *                           mask rows of the range, out (N, chunk, P, C);
 *   fresco_temporal_unpack: after the all-to-all back, buf (world, n_loc, chunk, P, C) holds this rank's frames' result rows
 *                           per range:  out[c*n_loc + fl][fwd_map[f0+fl][d*P + pl]] = buf[(d*n_loc + fl)*chunk + c][pl]. */
int fresco_temporal_pack(const void* q, const void* k, const void* v, const int64_t* fwd_map, void* buf, int chunk,
                         int n_loc, int f0, int HW, int C, int world, int64_t q_ld, int64_t k_ld, int64_t v_ld,
                         void* stream);
int fresco_temporal_attn_packed(const void* qkv, const uint8_t* mask, void* out, int chunk, int N, int P, int H, int D,
                                float scale, void* stream);
int fresco_temporal_unpack(const void* buf, const int64_t* fwd_map, void* out, int chunk, int n_loc, int f0, int HW,
                           int C, int world, void* stream);

/* ------------------------------------------------------------------------------------------
 * (a8)  flow_warp / bilinear_sample  (GEO:41-72): bilinear, zeros padding, align_corners=True.
 *   x, out : (B, C, h, w) fp32;  flow : (Bf, 2, h, w) fp32, channel 0 = x, 1 = y.
 *   Batch b samples with flow[b % Bf]  (Bf = B, or the un-repeated N when x holds `chunk` copies).
 * ------------------------------------------------------------------------------------------ */
int fresco_flow_warp(const float* x, const float* flow, float* out,
                     int B, int C, int h, int w, int Bf, void* stream);

/* F.interpolate(x * mul, scale_factor=s, mode='bilinear') (align_corners=False, no antialias);
 * (B,C,H,W) -> (B,C,ho,wo) fp32 with the source-coordinate scale rscale = 1/s as torch computes
 * it when scale_factor is given (FU:26,30,35; DH:439,441,937). */
int fresco_resize_bilinear(const float* x, float* out, int BC, int H, int W, int ho, int wo,
                           float rscale_h, float rscale_w, float mul, void* stream);

/* F.max_pool2d(x, kernel_size=k) (stride k, floor) on (BC,H,W) fp32 -> (BC,H/k,W/k)  (FU:27,31; DH:440,442) */
int fresco_max_pool(const float* x, float* out, int BC, int H, int W, int k, void* stream);

/* fp32 attention for the flow network (f3): out = softmax(scale * q k^T) v, one head.
 *   GMFlow's full / window attention (gmflow/transformer.py:8-17, 92-95; windows = batch entries), global
 *   correlation softmax (gmflow/matching.py:7-36: v = pixel grid, Dv = 2) and flow propagation
 *   (gmflow/transformer.py:356-372: v = flow, Dv = 2).
 *   q (B,Lq,D), k (B,Lk,D), v (B,Lk,Dv), out (B,Lq,Dv): fp32 row-major, dense.  D in {32, 64, 128}, Dv <= 128.
 *   fp32-class accuracy on the fp16 matrix pipe: operands split into fp16 pieces (33-bit logits, 22-bit P V products),
 *   fp32 softmax (the flows feed integer decisions downstream).  |q scale|, |k|, |v| < 1000. */
int fresco_attn_f32(const float* q, const float* k, const float* v, float* out, int B, int Lq, int Lk, int D, int Dv,
                    float scale, void* stream);

/* The same with a caller-provided workspace (fresco_attn_f32_workspace_bytes(B, Lk, D, Dv) bytes, 16-byte aligned): K and
 * V are converted to the kernel's split-fp16 LDS images ONCE per launch instead of once per 128-query workgroup (what
 * pays as soon as two workgroups share a key set: Lq >= 256).  Same results as fresco_attn_f32, bit for bit. */
size_t fresco_attn_f32_workspace_bytes(int B, int Lk, int D, int Dv);
int fresco_attn_f32_ws(const float* q, const float* k, const float* v, float* out, void* workspace,
                       size_t workspace_bytes, int B, int Lq, int Lk, int D, int Dv, float scale, void* stream);

/* The same WITHOUT the range limit: a range pass over q, k, v raises *flag (one int32 of device memory the caller lends
 * for the call) when any operand is beyond what the fp16 pieces can hold (|q scale log2 e|, |k|, |v| >= 1000) or not
 * finite, and an exact-fp32 MFMA kernel behind the split-fp16 one then recomputes the launch (it returns at once when the
 * flag is clear: in range the results are fresco_attn_f32's bit for bit).  workspace may be NULL (per-workgroup staging
 * form).  This is the entry fresco_amd.ops.attention_f32 -- and with it the flow network -- uses. */
int fresco_attn_f32_guarded(const float* q, const float* k, const float* v, float* out, void* workspace,
                            size_t workspace_bytes, int* flag, int B, int Lq, int Lk, int D, int Dv, float scale,
                            void* stream);

/* The guarded workspace form without the separate range pass and flag memset (round 6): the k / v range test runs inside the
 * split pass, the q test inside the attention kernel's prologue.  zero_flag: one int32 of device memory that the CALLER
 * guarantees to be ZERO in stream order when the call is issued, and untouched by anyone else until the call's kernels are
 * done (fresco_amd.ops hands out the words of a zero-filled pool, each once).  workspace as fresco_attn_f32_ws (not NULL).
 * Results: fresco_attn_f32_guarded's with a workspace, bit for bit, in range and out of range.  fresco_amd.ops.attention_f32
 * uses it whenever it takes the workspace form (Lq >= 256: every attention call of the flow network). */
int fresco_attn_f32_guarded_ws(const float* q, const float* k, const float* v, float* out, void* workspace,
                               size_t workspace_bytes, int* zero_flag, int B, int Lq, int Lk, int D, int Dv, float scale,
                               void* stream);

/* ---- the flow network's dense layers (f3): GMFlow's CNN encoder, transformer projections / FFN / LayerNorms, upsampler head
 * (gmflow/backbone.py:7-117, transformer.py:111-237, gmflow.py:44-90).  fp32 in, fp32 out, fp32-class accuracy on the fp16
 * matrix pipe: a tensor that feeds a product exists as a pair of fp16 planes (hi, lo), x = hi + lo to 2^-22, written by its
 * producer (fresco_fn_prep / fresco_fn_layernorm / fresco_fn_gemm's epilogue); products are hi hi + hi lo + lo hi.
 * The planes hold x * split_scale (a power of two: the matrix pipe flushes fp16 subnormals, so lo pieces must stay normal
 * numbers; fresco_amd uses 2^6 for activations, |x| < 1000, and 2^10 for weights, |w| < 60; beyond that the scaled value
 * saturates -- finite, wrong -- and the producer ORs 1 into the caller's `range_flag` word (int32 in device memory, cleared
 * by the caller, may be NULL): fresco_amd's flow network checks it once per forward and recomputes with library ops);
 * fresco_fn_gemm multiplies its fp32 accumulators by acc_scale = 1 / (scale_A * scale_W) before the bias.
 * Activations are NHWC: rows m = (image, y, x), channels contiguous -- the transformer's (B, L, C) token layout. */

/* out[m][n] = act( sum_k A(m, k) W[n][k] + bias[n] ),  m < M, n < N, K % 32 == 0.
 *   kh == 0: A = a_hi + a_lo, (M, K) row-major with row stride lda (halfs)               -- nn.Linear / 1 x 1 conv
 *   kh  > 0: implicit im2col of an NHWC tensor (n_img, H, W, cin = K / (kh kw)), pixel stride lda >= cin, cin % 32 == 0,
 *            k = (ky, kx, ci), zero padding `pad`, stride `stride`; M must be n_img * OH * OW    -- nn.Conv2d
 *   w_hi / w_lo: (N, K) row-major fp16 planes; bias (N) fp32 or NULL; act 0 none, 1 ReLU, 2 GELU (erf form).
 *   out (M, ldc) fp32 and / or out_hi / out_lo (M, ldo) fp16 planes of the result (either may be NULL, not both; planes need
 *   N % 8 == 0).  zeros: 16 bytes of zeros in device memory (the source of every row outside the problem and of a
 *   convolution's zero padding: the operands arrive by LDS-DMA).  stats (convolutions whose OH * OW is a multiple of 256;
 *   a_rows / out_rows (linear layers only, int32 (M) device tables or NULL): problem row m reads input row a_rows[m], its
 *   results go to output row out_rows[m] (a token gather / scatter folded into the product).  stats: (M / 64) * N * 2 doubles that
 *   receive per-column partial sums / sums of squares of the results for fresco_fn_colstats_finish (InstanceNorm2d without a
 *   second pass over the convolution's output). */
int fresco_fn_gemm(const void* a_hi, const void* a_lo, int64_t lda, const void* w_hi, const void* w_lo, const float* bias,
                   float* out, void* out_hi, void* out_lo, int64_t ldc, int64_t ldo, int M, int N, int K, int act,
                   float acc_scale, float split_scale, int n_img, int H, int W, int kh, int kw, int stride, int pad,
                   void* stats, const void* zeros, const int32_t* a_rows, const int32_t* out_rows, int32_t* range_flag,
                   int out_col_block, int64_t out_block_stride, void* stream);
/* out_col_block > 0 (fp32 output only, N % out_col_block == 0): column n of the product goes to matrix n / out_col_block of
 * out_col_block columns (row stride ldc >= out_col_block), the matrices out_block_stride floats apart -- several projections of
 * one input as ONE product (W = their weight rows stacked), every projection's rows contiguous (round 6: q | k | v and k | v of the
 * flow network's attention layers).  0: one (M, ldc) matrix. */

/* nn.InstanceNorm2d statistics (affine=False, biased variance): x (n_img * rows, C) fp32 NHWC -> mean, rstd = 1 / sqrt(var +
 * eps), (n_img, C) each.  fp64 partial sums in a fixed order.  C <= 256. */
size_t fresco_fn_colstats_workspace_bytes(int n_img, int rows, int C);
int fresco_fn_colstats(const float* x, float* mean, float* rstd, void* workspace, size_t workspace_bytes, int n_img, int rows,
                       int C, float eps, void* stream);
int fresco_fn_colstats_finish(const void* stats, float* mean, float* rstd, int n_img, int rows, int C, float eps, void* stream);

/* y = relu_b?( relu_a?( (x - mean[img]) * rstd[img] ) + residual ) on (M, C) fp32 rows (mean / rstd / residual may be NULL;
 * img = m / rows_per_img).  Writes y (M, C) fp32 and / or the fp16 planes out_hi / out_lo with row stride ldo >= C, channels
 * C .. ldo-1 zeroed (K padding of the product that reads them).  C % 4 == 0, ldo % 4 == 0. */
int fresco_fn_prep(const float* x, const float* mean, const float* rstd, const float* residual, float* y, void* out_hi,
                   void* out_lo, int64_t M, int C, int ldo, int rows_per_img, int relu_a, int relu_b, float split_scale,
                   int32_t* range_flag, void* stream);

/* nn.LayerNorm(128) (+ residual): y = residual + ((x - mean) / sqrt(var + eps)) gamma + beta on (M, 128) fp32 rows; fp32 y
 * (row stride ldy) and / or fp16 planes (row stride ldo). */
int fresco_fn_layernorm(const float* x, const float* gamma, const float* beta, const float* residual, float* y, void* out_hi,
                        void* out_lo, int64_t ldy, int64_t ldo, int64_t M, int C, float eps, float split_scale,
                        int32_t* range_flag, void* stream);

/* The encoder's stem: Conv2d(3, 64, 7, stride 2, padding 3, bias=False) (gmflow/backbone.py:69) on the matrix pipe (split-fp16
 * products, fp32-class accuracy; round 6 -- rounds 5: direct fp32 FMAs).  x (n_img, H, W, 3) NHWC fp32; w_hi / w_lo: fp16
 * planes (scale 2^10, as fresco_fn_prep writes them) of the (64, 224) matrix W'[cout][32 ky + 3 kx + ci] = weight[cout][ci][ky][kx],
 * the 11 surplus positions of every kernel row ZERO; out (n_img, OH, OW, 64) NHWC fp32, OH = (H - 1) / 2 + 1.
 * stats (may be NULL): fp64 InstanceNorm partial sums of `out`, (n_img OH OW / 64) slabs x 64 channels x 2, combined by
 * fresco_fn_colstats_finish; needs OW % 64 == 0 and OH OW % 256 == 0 (FRESCO_EUNSUPPORTED otherwise).
 * range_flag (may be NULL): OR 1 when an input value leaves the operand planes' range (|x| >= 1015). */
int fresco_fn_conv7_rgb(const float* x, const void* w_hi, const void* w_lo, float* out, void* stats, int n_img, int H, int W,
                        int32_t* range_flag, void* stream);

/* Convex upsampling by 8 (gmflow.py:75-90): out (B, 2, 8h, 8w) = softmax-over-9-weighted mix of the 3 x 3 coarse neighbourhood
 * of 8 * flow.  logits (B, h, w, 576) = the mask head's NHWC rows (channel = n * 64 + ky * 8 + kx), flow (B, h * w, 2). */
int fresco_fn_convex_upsample(const float* logits, const float* flow, float* out, int B, int h, int w, void* stream);

/* forward_backward_consistency_check (gmflow/geometry.py:75-96) fused with the colour-difference
 * occlusion refinement of get_flow_and_interframe_paras (DH:919-926).  Pair n couples frame n with frame
 * (n+1) mod N: fwd_flow[n] maps frame n onto n+1, bwd_flow[n] the reverse; all fp32.
 *   fwd_occ[n] = |fwd + warp(bwd, fwd)| > alpha*(|fwd|+|bwd|) + beta  OR  mean_c |img[n]   - warp(img[n+1], fwd)| > color_thr
 *   bwd_occ[n] = |bwd + warp(fwd, bwd)| > alpha*(|fwd|+|bwd|) + beta  OR  mean_c |img[n+1] - warp(img[n],   bwd)| > color_thr
 *   images (N,C,H,W) in 0..255 or NULL (consistency check only); flows (N,2,H,W); occs (N,H,W) in {0,1}.
 *   Reference constants: alpha 0.01, beta 0.5, color_thr 255*0.25. */
int fresco_flow_occlusion(const float* images, const float* fwd_flow, const float* bwd_flow, float* fwd_occ,
                          float* bwd_occ, int N, int C, int H, int W, float alpha, float beta, float color_thr,
                          void* stream);

/* Dilate (UT:81-93): replicate pad (k-1)/2, k x k box sum, clamp [0,1]; (BC,H,W) fp32, k odd */
int fresco_dilate(const float* x, float* out, int BC, int H, int W, int k, void* stream);

/* ------------------------------------------------------------------------------------------
 * (a7)  warp_tensor frame chain (FU:41-51).  `lat` (chunk*N, C, h, w) fp32 is updated IN PLACE:
 *   for c in chunk: for i in 0..N-2:  lat[c*N+i+1] = lat[c*N+i+1]*(1-m) + warp(lat[c*N+i], bwd_flow[i])*m,
 *                                      m = (1-bwd_occ[i]) * sal[i+1] * warp_sal[i]
 *                   last:             lat[c*N+N-1] blended with warp(lat[c*N], fwd_flow[N-1]),
 *                                      m = (1-fwd_occ[N-1]) * sal[N-1] * warp_sal_last
 *   bwd_flow, fwd_flow : (N,2,h,w); bwd_occ, fwd_occ, sal, warp_sal : (N,h,w); warp_sal_last : (h,w).
 *   The chain is sequential in the frame index (N launches), parallel over chunk, C, h, w.
 * ------------------------------------------------------------------------------------------ */
int fresco_warp_fuse_chain(float* lat, const float* bwd_flow, const float* fwd_flow,
                           const float* bwd_occ, const float* fwd_occ, const float* sal,
                           const float* warp_sal, const float* warp_sal_last,
                           int chunk, int N, int C, int h, int w, void* stream);

/* ------------------------------------------------------------------------------------------
 * (a9)  adaptive_instance_normalization (UT:58-78) over rows of L = h*w elements:
 *   out = (content - mean_c) / sqrt(var_c + eps_content) * sqrt(var_s + eps_style) + mean_s,
 *   unbiased variance.  The reference's style eps is 1.0 (UT:73 passes chunk into eps).
 *   content, style, out : (rows, L), dtype FRESCO_F16 or FRESCO_F32 (all three the same).
 * ------------------------------------------------------------------------------------------ */
int fresco_adain(const void* content, const void* style, void* out, int rows, int L,
                 float eps_content, float eps_style, int dtype, void* stream);

/* calc_mean_std (src/utils.py:58-67): per row (= one (sample, channel) plane of L values, dtype F16 / F32):
 *   mean[row] = mean(x), stdv[row] = sqrt(unbiased variance + eps), both fp32.  L > 1. */
int fresco_chan_mean_std(const void* x, float* mean, float* stdv, int rows, int L, float eps, int dtype,
                         void* stream);

/* ------------------------------------------------------------------------------------------
 * (a6)  optimize_feature (DH:416-488): Adam on an fp32 copy of the features against
 *   L = 2*mean(|(c2 - W_b c1)(1-occ_b)| + |(c1 - W_f c2)(1-occ_f)|) + intra_weight*mean|V V^T - T|.
 *
 *   cs        : (chunk*N, C, h, w) fp32, updated in place (the optimised parameter)
 *   fwd_flow, bwd_flow : (N, 2, h, w) fp32 at feature resolution (already scaled), or NULL
 *   fwd_occ, bwd_occ   : (N, h, w) fp32, or NULL           (temporal term off when NULL)
 *   target    : (chunk*N, hw, hw) fp32 Gram target, or NULL (spatial term off when NULL)
 *   iters Adam steps (lr, beta1, beta2, eps as torch.optim.Adam); no autograd: analytic
 *   gradients.  The adjoint of the bilinear warp is evaluated as a deterministic gather over a
 *   per-call CSR of the tap matrix (no atomics), so results are run-to-run reproducible.
 *   fresco_opt_run keeps everything on `stream` and touches no state outside its arguments.
 *   fresco_opt_run_ctx(ctx, ...) -- the same call with a context (fresco_ctx_create / _destroy: a side stream + events,
 *   created on first use on the device current then, owned by the CALLER; round 6: the library keeps no process-wide
 *   stream table any more) -- runs, with chunk == 2 and N * h * w >= 2048, the two CFG halves (independent problems) as two
 *   pipelines: one on `stream`, one on the context's side stream, forked from and joined back into `stream` by events
 *   inside the call -- the caller sees ordinary stream order (FRESCO_OPT_SPLIT=0 keeps everything on `stream`; the
 *   results are bit-identical either way).  One call at a time per context (a second concurrent call on the same context,
 *   or a call on another device than the context's, runs on one stream); use one context per host thread / stream.
 *
 *   fresco_opt_loss_grad evaluates the closure once: grad (same shape as cs) and, if loss != NULL,
 *   loss[0] = temporal term, loss[1] = spatial term (device floats).  Test / debugging entry.
 * ------------------------------------------------------------------------------------------ */
size_t fresco_opt_workspace_bytes(int chunk, int N, int C, int h, int w, int has_temporal,
                                  int has_target);

int fresco_opt_run(float* cs, const float* fwd_flow, const float* bwd_flow,
                   const float* fwd_occ, const float* bwd_occ, const float* target,
                   void* workspace, size_t workspace_bytes,
                   int chunk, int N, int C, int h, int w,
                   float intra_weight, int iters, float lr, float beta1, float beta2, float eps,
                   void* stream);

int fresco_ctx_create(void** ctx);
int fresco_ctx_destroy(void* ctx); /* waits for the context's stream; FRESCO_EINVAL while a call is using it */
int fresco_opt_run_ctx(void* ctx, float* cs, const float* fwd_flow, const float* bwd_flow,
                       const float* fwd_occ, const float* bwd_occ, const float* target,
                       void* workspace, size_t workspace_bytes,
                       int chunk, int N, int C, int h, int w,
                       float intra_weight, int iters, float lr, float beta1, float beta2, float eps,
                       void* stream);

int fresco_opt_loss_grad(const float* cs, const float* fwd_flow, const float* bwd_flow,
                         const float* fwd_occ, const float* bwd_occ, const float* target,
                         float* grad, float* loss, void* workspace, size_t workspace_bytes,
                         int chunk, int N, int C, int h, int w, float intra_weight, void* stream);

/* Frame-sharded optimize_feature (multi-GPU): this rank owns n_loc consecutive frames of
 * both CFG halves out of N_total.  cs, target : local (chunk*n_loc, ...).  The temporal term couples
 * neighbouring frames, so before EVERY step the host hands over the current values of the frame before
 * (halo_l) and after (halo_r) the owned range, each (chunk, C, h, w) fp32 (ring order, wrap-around).
 * fwd_flow, bwd_flow : (n_loc+1, 2, h, w), fwd_occ, bwd_occ : (n_loc+1, h, w) -- entry j belongs to the
 * frame pair (f0-1+j, f0+j) mod N_total; both loss terms are normalised by the GLOBAL batch 2*N_total.
 * begin: zero the Adam state, build the warp-adjoint CSRs; step `it` = 1..iters: one Adam iteration.
 * With n_loc = N_total and halos = own last / first frame this reproduces fresco_opt_run. */
size_t fresco_opt_sharded_workspace_bytes(int chunk, int n_loc, int C, int h, int w, int has_temporal,
                                          int has_target);
int fresco_opt_sharded_begin(const float* fwd_flow, const float* bwd_flow, const float* fwd_occ,
                             const float* bwd_occ, void* workspace, size_t workspace_bytes,
                             int chunk, int n_loc, int N_total, int C, int h, int w, int has_target,
                             void* stream);
int fresco_opt_sharded_step(float* cs, const float* halo_l, const float* halo_r,
                            const float* fwd_flow, const float* bwd_flow, const float* fwd_occ,
                            const float* bwd_occ, const float* target, void* workspace,
                            size_t workspace_bytes, int chunk, int n_loc, int N_total, int C, int h, int w,
                            float intra_weight, int it, float lr, float beta1, float beta2, float eps,
                            void* stream);
/* The same step in two host calls, so that the neighbour exchange of the halo frames can run UNDER the launches that do
 * not read them.  part = 1: normalisation, the residual signs of the interior pairs, Gram and S V products (halo_l /
 * halo_r are not read and may be NULL); part = 2: the residual signs of the two pairs that touch a halo frame, then
 * Adam (halos required); part = 3: both = fresco_opt_sharded_step.  Part 1 followed by part 2 performs, per element,
 * exactly the operations of the undivided step: results are identical bit for bit.  Typical loop of a rank:
 * start the (asynchronous) exchange of the frames Adam(it-1) produced -> part 1 of step it -> wait for the halos ->
 * part 2 of step it. */
int fresco_opt_sharded_step_part(float* cs, const float* halo_l, const float* halo_r,
                                 const float* fwd_flow, const float* bwd_flow, const float* fwd_occ,
                                 const float* bwd_occ, const float* target, void* workspace,
                                 size_t workspace_bytes, int chunk, int n_loc, int N_total, int C, int h, int w,
                                 float intra_weight, int it, float lr, float beta1, float beta2, float eps,
                                 int part, void* stream);

/* Gram target of get_intraframe_paras (DH:889-895): T[b] = V V^T, V = rows of x (B,C,h,w)
 * viewed as (B, hw, C) and L2-normalised; fp32 (B,hw,hw).  workspace: B*C*hw + 33*B*hw floats
 * (each block rounded up to 256 bytes). */
int fresco_gram_target(const float* x, float* target, void* workspace, size_t workspace_bytes,
                       int B, int C, int hw, void* stream);

/* ------------------------------------------------------------------------------------------
 * (f1)  FLATTEN pixel correspondences -- get_mapping_ind / get_single_mapping_ind (FU:56-138) without
 * the per-pixel Python loop.  Inputs are the reference's intermediates at the reduced resolution
 * (produced with fresco_resize_bilinear, scale_factor = 1/scale):
 *   flow   : (N-1, 2, H, W) resized bwd flow, channel 0 = x, 1 = y, NOT yet divided by scale
 *   occ    : (N-1, H, W)    resized bwd occlusion (> 0.5 = occluded)
 *   frames : (N, 3, H*W)    resized images
 * Outputs: fwd_map, bwd_map (N, H*W) int64, mask (H*W, N, N) uint8 {0,1}.  Integers are bit-exact with
 * the reference's CPU run (ties: the earliest source wins, unlinked targets get the unused sources in
 * ascending order).
 * ------------------------------------------------------------------------------------------ */
size_t fresco_mapping_workspace_bytes(int N, int H, int W);
int fresco_mapping_ind(const float* flow, const float* occ, const float* frames, int64_t* fwd_map,
                       int64_t* bwd_map, uint8_t* mask, void* workspace, size_t workspace_bytes,
                       int N, int H, int W, float scale, void* stream);

/* ------------------------------------------------------------------------------------------
 * (f2)  DDPM step of src/pipe_FRESCO.py:14-77 (+ classifier-free guidance, 212-214), elementwise over n
 * values, dtype FRESCO_F16 / FRESCO_F32 (fp32 arithmetic inside):
 *   fresco_ddpm_x0  : eps = eps_text ? eps_uncond + guidance*(eps_text - eps_uncond) : eps_uncond
 *                     (written to eps_out if non-NULL);  x0 = (xt - sqrt_beta_prod*eps) / sqrt_alpha_prod
 *   fresco_ddpm_prev: out = c_x0*x0 + c_xt*xt + sigma*noise[i % noise_period]
 *                     (noise_period = n, or one frame's element count for repeat_noise)
 * The background-smoothing hack between the two (VAE decode -> warp_tensor -> VAE encode of x0) stays
 * with the caller.
 * ------------------------------------------------------------------------------------------ */
int fresco_ddpm_x0(const void* xt, const void* eps_uncond, const void* eps_text, void* x0, void* eps_out,
                   int64_t n, float guidance, float sqrt_beta_prod, float sqrt_alpha_prod, int dtype,
                   void* stream);
int fresco_ddpm_prev(const void* x0, const void* xt, const void* noise, void* out, int64_t n,
                     int64_t noise_period, float c_x0, float c_xt, float sigma, int dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FRESCO_HIP_H */

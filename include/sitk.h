/* sitk.h -- C ABI of libsitk.so: the MI355X (gfx950) native SiT training hot path.
 *
 * The reference (SD3004/surface-vision-transformers) has no FFI layer: its boundary for this path
 * is the nn.Module API of models/sit.py:26-82 and models/mpp.py:46-134 plus the third-party
 * vit_pytorch.vit.Transformer it constructs at models/sit.py:57.  Every entry point below cites
 * the reference lines whose arithmetic it replaces.  The Python mirror of those modules
 * (surface-vision-transformers_amd/models/{sit,mpp}.py) binds these symbols with ctypes.
 *
 * Conventions:
 *   - plain C: raw DEVICE pointers, explicit sizes / leading dimensions (in ELEMENTS), a
 *     hipStream_t passed as void*; no torch types, no allocation, no ownership transfer,
 *     no host synchronisation, no global mutable state.  Every call only ENQUEUES work on
 *     `stream` and is re-entrant (a communication stream may run concurrently).
 *   - return 0 on success, <0 on error; sitk_last_error() returns a thread-local message.
 *   - `dtype` selects the COMPUTE/STORAGE type of activations and weight copies:
 *       SITK_BF16  bf16 operands, v_mfma_f32_16x16x32_bf16, fp32 accumulate   (benchmark mode)
 *       SITK_F16   IEEE half operands, v_mfma_f32_16x16x32_f16, fp32 accumulate: the same rate and bytes as bf16 with 3
 *                  more mantissa bits (meets the 1e-3 parity bar); 5 exponent bits: callers run backward on a
 *                  loss-scaled gradient stream (the *_scale arguments below)
 *       SITK_F32   f32 operands,  v_mfma_f32_16x16x4_f32 (exact fp32)          (verification mode)
 *     LayerNorm statistics, softmax, GELU, the residual stream, every gradient of a parameter
 *     and the optimizer state are fp32 in both modes.
 *   - all pointers 16-byte aligned; feature counts (dim, mlp_dim, heads*64, patch_dim) and leading
 *     dimensions multiples of 8.
 */
#ifndef SITK_H
#define SITK_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SITK_F32 0
#define SITK_BF16 1
#define SITK_F16 2

#define SITK_OK 0
#define SITK_ERR_INVALID (-1)
#define SITK_ERR_LAUNCH (-2)

#define SITK_ABI_VERSION 12

typedef void* sitk_stream_t; /* hipStream_t */

int sitk_abi_version(void);
const char* sitk_last_error(void);
/* size in bytes of one element of `dtype` (4 or 2) */
int sitk_dtype_size(int dtype);

/* ---------------------------------------------------------------------------------------------
 * a1-a3  Patch gather.  tools/preprocessing.py:74-84 (out[s,c,j,v] = X[s,c,table[v,j]]) fused with
 * Rearrange('b c n v -> b n (v c)') of models/sit.py:49 / models/mpp.py:82-83.
 *   x_bvc     (B, n_vertices, C) fp32, channels last, C in 1..4 (C == 4: one 16-byte record per vertex, the fast path)
 *   table_pv  (P, V) uint16, PATCH-major vertex ids (ids < n_vertices)
 *   tokens    (B*P, ld) `dtype`; columns [0, V*C) are written as f = v*C + c, columns
 *             [V*C, ld) are zero-filled (ld >= V*C, multiple of 4; of 8 when it feeds a bf16 GEMM).
 * Integer-indexed copy: bit exact in SITK_F32; one RNE rounding per element in SITK_BF16.      */
int sitk_gather_tokens(const float* x_bvc, const uint16_t* table_pv, void* tokens, int B, int n_vertices,
                       int C, int P, int V, int ld, int dtype, sitk_stream_t stream);

/* Same with the reference's per-channel normalisation fused in front of the gather
 * (tools/preprocessing.py:72: (data - means) / stds; mean, stdv: (C) fp32 device arrays, a true fp32
 * division) -- the device-resident input pipeline of SURVEY 8(f).2: raw surfaces stay in HBM and only
 * the table indexes them every step.                                                             */
int sitk_gather_tokens_norm(const float* x_bvc, const uint16_t* table_pv, const float* mean, const float* stdv,
                            void* tokens, int B, int n_vertices, int C, int P, int V, int ld, int dtype,
                            sitk_stream_t stream);

/* Batch assembly from a data set that stays resident in HBM (replaces the DataLoader of tools/train.py:97-113 and
 * the per-step H2D copy of tools/train.py:282-283): batch row b is sample sample_idx[b] of x_all (S, n_vertices, C)
 * fp32; mean / stdv as above (both null: no normalisation).  targets_all (S, n_targets) fp32 (or null): the labels of
 * the selected samples are written to target_out (B, n_targets).  Only the B int32 indices move per step.        */
int sitk_gather_tokens_idx(const float* x_all, const int32_t* sample_idx, const uint16_t* table_pv, const float* mean,
                           const float* stdv, void* tokens, const float* targets_all, float* target_out, int n_targets,
                           int B, int n_vertices, int C, int P, int V, int ld, int dtype, sitk_stream_t stream);

/* Drop-in layout of the reference: x_bcpv (B, C, P, V) fp32 (models/sit.py:47-49) -> tokens as above. */
int sitk_patchify(const float* x_bcpv, void* tokens, int B, int C, int P, int V, int ld, int dtype,
                  sitk_stream_t stream);

/* rows x cols fp32 (leading dim lds) -> `dtype` (leading dim ldd >= cols, pad columns zeroed). */
int sitk_cast_rows(const float* src, int lds, void* dst, int ldd, int64_t rows, int cols, int dtype,
                   sitk_stream_t stream);

/* Weight staging for one Linear: w (rows, cols) fp32 ->
 *   w_c (rows, ldc) `dtype`  [pad zeroed]   used as  Y = X W^T      (nn.Linear forward)
 *   w_t (cols, ldt) `dtype`  = W^T          used as dX = dY W       (input gradient)
 * either destination may be NULL.                                                               */
int sitk_stage_weight(const float* w, int rows, int cols, void* w_c, int ldc, void* w_t, int ldt, int dtype,
                      sitk_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * GEMM  C[m][n] = sum_k A[m][k] W[n][k]  (nn.Linear: models/sit.py:50,63; the encoder's to_qkv,
 * to_out.0, net.0, net.3; models/mpp.py:66) with fused epilogues.                               */
typedef struct {
  int group;  /* 0: identity.  else physical_row(m) = (m / group) * stride + offset + m % group */
  int stride;
  int offset;
} sitk_rowmap;

#define SITK_EPI_STORE 0     /* out = acc (+bias)                                  out_dtype any  */
#define SITK_EPI_BIAS_RES 1  /* out(f32) = acc + bias + aux(f32)   residual / pos-embedding add   */
#define SITK_EPI_BIAS_GELU 2 /* u = acc + bias; out = gelu_erf'(u) - 1/2, out2 = gelu_erf(u)  both `dtype` (ABI 9: the centred
                                derivative is saved, not u -- backward needs u only through it)                   */
#define SITK_EPI_DGELU 3     /* out = acc * (aux + 1/2)   aux = the value saved by BIAS_GELU, both `dtype`         */

typedef struct {
  int M, N, K;
  const void* A; /* (M, K): `dtype`, or fp32 when a_is_f32 (converted while staging)        */
  int lda;
  int a_is_f32;
  sitk_rowmap amap;
  const void* W; /* (N, K) `dtype`                                                           */
  int ldw;
  int epilogue;
  void* out;
  int ldo;
  int out_is_f32;
  sitk_rowmap omap;
  void* out2;
  const float* bias; /* (N) or NULL */
  const void* aux;
  int ldaux;
  sitk_rowmap auxmap;
} sitk_gemm_desc;

int sitk_gemm_nt(const sitk_gemm_desc* d, int dtype, sitk_stream_t stream);

/* Weight gradient: dW[n][k] += sum_m dY[m][n] X[m][k], optionally db[n] += sum_m dY[m][n]; fp32 accumulation.
 * sitk_gemm_wgrad / _group (64 x 64 tiles, any shape): token chunks are summed into dW with float atomics (the result
 * depends on arrival order in the last bits).  sitk_gemm_wgrad_group_ws on eligible shapes (the encoder's): 256 x 192 /
 * 128 x 384 tiles, no float atomics -- a tile that covers all tokens is added straight into dW, token-split tiles go through a
 * slab in the workspace and a fixed-order reduction.
 *   dY (M, N): `dtype` or fp32 (dy_is_f32); X (M, K) `dtype`; row maps as above.  The large-tile path reads whole
 * 16-byte vectors: when N (or K) is not a multiple of 8, columns [N, round_up(N, 8)) of dY (X) must exist (lddy / ldx
 * cover them) and hold zeros.  dW: 16-byte aligned with lddw % 4 == 0 takes 16-byte read-modify-writes; a dW that is
 * only 4-byte aligned (or any other lddw) is still correct, element by element.                                       */
typedef struct {
  int M, N, K;
  const void* dY;
  int lddy;
  int dy_is_f32;
  sitk_rowmap dymap;
  const void* X;
  int ldx;
  sitk_rowmap xmap;
  float* dW; /* (N, lddw) fp32, accumulated */
  int lddw;
  float* db; /* (N) fp32 accumulated, or NULL */
} sitk_wgrad_desc;

int sitk_gemm_wgrad(const sitk_wgrad_desc* d, int dtype, sitk_stream_t stream);
/* Up to 4 independent weight gradients (the four Linears of one encoder layer) in ONE launch. */
int sitk_gemm_wgrad_group(const sitk_wgrad_desc* d, int count, int dtype, sitk_stream_t stream);
/* Same, with a caller-provided workspace: when every problem is bf16 with a dimension that is a multiple
 * of 192 the large-tile kernel (256 x 192 or 128 x 384 tiles, token-split tiles reduced through the workspace instead of
 * float atomics) runs; otherwise, or if `ws` is NULL / too small, this is sitk_gemm_wgrad_group.
 * sitk_gemm_wgrad_group_ws_bytes returns the workspace size that selects the large-tile path (0 = n/a). */
size_t sitk_gemm_wgrad_group_ws_bytes(const sitk_wgrad_desc* d, int count, int dtype);
int sitk_gemm_wgrad_group_ws(const sitk_wgrad_desc* d, int count, int dtype, void* ws, size_t ws_bytes,
                             sitk_stream_t stream);
/* The same sized for `cus` compute units instead of the chip (1..256): the token splits are chosen so that the launch
 * has at most ~cus workgroups (one per CU: a workgroup owns its CU's LDS) -- for launches that run on a side stream
 * BESIDE another kernel chain, on the CUs that chain leaves idle.  The workspace of the 256-CU form is large enough.  */
int sitk_gemm_wgrad_group_ws_cus(const sitk_wgrad_desc* d, int count, int dtype, void* ws, size_t ws_bytes, int cus,
                                 sitk_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * LayerNorm over the last dim (eps 1e-5, biased variance, affine): the PreNorm norms of the
 * encoder (state-dict keys transformer.layers.i.{0,1}.norm, utils/utils.py:18-22).
 *   x (rows, D) fp32 -> y (rows, D) `dtype`; mean/rstd (rows) fp32 saved for backward.          */
int sitk_layernorm_fwd(const float* x, const float* gamma, const float* beta, void* y, float* mean,
                       float* rstd, int64_t rows, int D, int dtype, sitk_stream_t stream);
/* dx_out (fp32) = dres (fp32, may be NULL, may alias dx_out) + LN'(dy); dgamma/dbeta accumulated.
 *   dx_out_c : optional second copy of dx_out in `dtype` (feeds the next GEMMs' LDS-DMA path)
 *   partials : optional scratch of sitk_layernorm_bwd_partial_floats(rows, D) floats; when given the
 *              per-workgroup dgamma/dbeta sums are stored there and reduced by a second small kernel
 *              instead of contended float atomics on 2*D addresses.                                */
size_t sitk_layernorm_bwd_partial_floats(int64_t rows, int D);
int sitk_layernorm_bwd(const void* dy, const float* x, const float* mean, const float* rstd,
                       const float* gamma, const float* dres, float* dx_out, void* dx_out_c, float* dgamma,
                       float* dbeta, float* partials, int64_t rows, int D, int dtype, sitk_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Fused MLP half of an encoder block, PreNorm(LayerNorm, FeedForward) + residual
 * (state-dict keys transformer.layers.i.1.{norm, fn.net.0, fn.net.3}, utils/utils.py:21-33;
 * FeedForward = Linear, GELU(erf), Dropout(0), Linear, Dropout(0)), in ONE launch per direction.
 * Specialised: dtype bf16, D == 192, M % 64 == 0, M <= 1024 (sitk_mlp_fused_supported); other
 * shapes use sitk_layernorm_* + sitk_gemm_nt.
 *   forward : out = x + gelu(LN(x) W1^T + b1) W2^T + b2
 *     x (rows, D) fp32; w1_c (M, D), w2_c (D, M) `dtype` copies; out (rows, D) fp32 (may not alias x)
 *     saved for backward (each may be NULL): h = LN(x) (rows, D) `dtype`, mean/rstd (rows) fp32,
 *     gd = gelu'(u) - 1/2 (rows, M) `dtype`, u = the pre-activation LN(x) W1^T + b1 (ABI 9: the DERIVATIVE is saved,
 *     not u itself -- forward has Phi(u) in hand, backward's elementwise phase becomes one multiply-add; centred so that
 *     the 16-bit rounding is finest near u = 0, where most pre-activations are);
 *     g = gelu(u) (rows, M) `dtype`, the operand of net.3's weight gradient.
 *   backward: dx = dy + LN'(dh), dh = du W1, du = (dy W2) * (gd + 1/2)
 *     dy fp32 + dy_c its `dtype` copy; x/mean/rstd/gd as saved; w2t_c = W2^T (M, D), w1t_c = W1^T (D, M);
 *     writes du (rows, M) `dtype` (operand of net.0's weight gradient), dx fp32 and dx_c its `dtype` copy, and
 *     per-workgroup LayerNorm dgamma/dbeta sums to `partials`
 *     (sitk_mlp_bwd_partial_floats(rows) floats, layout [workgroup][2][D], workgroup = 96 or 128 rows).   */
int sitk_mlp_fused_supported(int D, int M, int dtype);
int sitk_mlp_fwd(const float* x, const float* ln_w, const float* ln_b, const void* w1_c, const float* b1,
                 const void* w2_c, const float* b2, void* h, float* mean, float* rstd, void* gd, void* g,
                 float* out, int64_t rows, int D, int M, int dtype, sitk_stream_t stream);
/* forward with the attention output projection folded in (state-dict keys layers.i.0.fn.to_out.0 + the first
 * residual add of the block): x_mid = x + o Wo^T + bo is computed in the kernel's prologue, written to `xmid`
 * (saved for backward) and fed to the LayerNorm; out = x_mid + MLP(LN(x_mid)).  Needs heads * 64 == 192 and at
 * most 24 576 rows (sitk_attn_out_mlp_fused_supported); o_c (rows, 192) and wo_c (192, 192) are `dtype`.       */
int sitk_attn_out_mlp_fused_supported(int64_t rows, int D, int I, int M, int dtype);
int sitk_attn_out_mlp_fwd(const void* o_c, const void* wo_c, const float* bo, const float* x, float* xmid,
                          const float* ln_w, const float* ln_b, const void* w1_c, const float* b1, const void* w2_c,
                          const float* b2, void* h, float* mean, float* rstd, void* gd, void* g, float* out,
                          int64_t rows, int D, int I, int M, int dtype, sitk_stream_t stream);
/* ... and with the NEXT block's LayerNorm + to_qkv appended (layers.{i+1}.0.norm, layers.{i+1}.0.fn.to_qkv):
 * n_h = LN(out) (may be NULL), n_mean/n_rstd, n_qkv = n_h Wqkv^T (rows, N3) -- one launch from the attention
 * output of block i to the attention input of block i + 1.                                                    */
int sitk_attn_out_mlp_next_fwd(const void* o_c, const void* wo_c, const float* bo, const float* x, float* xmid,
                               const float* ln_w, const float* ln_b, const void* w1_c, const float* b1, const void* w2_c,
                               const float* b2, void* h, float* mean, float* rstd, void* gd, void* g, float* out,
                               const float* n_ln_w, const float* n_ln_b, const void* n_wqkv_c, void* n_h, float* n_mean,
                               float* n_rstd, void* n_qkv, int N3, int64_t rows, int D, int I, int M, int dtype,
                               sitk_stream_t stream);
size_t sitk_mlp_bwd_partial_floats(int64_t rows);
/* d to_qkv + LayerNorm backward of layer l (sitk_ln_gemm_bwd: dqkv .. partials1, N = 3 heads 64) and the fused MLP backward of
 * layer l - 1 (sitk_mlp_bwd on the dx / dx_c the first half has just written: xmid .. partials2) in ONE launch (ABI 9).  Both
 * kernels give a workgroup the same 96 rows, so the second half reads its own workgroup's rows back from L2.  Same results,
 * bit for bit, as the two calls.  Needs sitk_ln_gemm_mlp_bwd_supported (h16, dim 192, at most 24 576 rows).              */
int sitk_ln_gemm_mlp_bwd_supported(int64_t rows, int D, int N, int M, int dtype);
int sitk_ln_gemm_mlp_bwd(const void* dqkv, const void* wqkv_t_c, const float* x, const float* mean1, const float* rstd1,
                         const float* ln1_w, const float* dres, float* dx, void* dx_c, float* partials1, int N,
                         const float* xmid, const float* mean2, const float* rstd2, const float* ln2_w, const void* w2t_c,
                         const void* w1t_c, const void* gd, void* du, float* dx_mid, void* dx_mid_c, float* partials2,
                         int64_t rows, int D, int M, int dtype, sitk_stream_t stream);
int sitk_mlp_bwd(const float* dy, const void* dy_c, const float* x, const float* mean, const float* rstd,
                 const float* ln_w, const void* w2t_c, const void* w1t_c, const void* gd, void* du, float* dx,
                 void* dx_c, float* partials, int64_t rows, int D, int M, int dtype, sitk_stream_t stream);
/* sitk_mlp_bwd for the first backward kernel of a chain, where dy exists in fp32 only: dy_c is an OUTPUT here -- the kernel
 * rounds its operand from dy and writes the compute-dtype copy (the weight gradient of net.3 reads it).  sitk_cast_rows +
 * sitk_mlp_bwd in one launch, same bits (ABI 9). */
int sitk_mlp_bwd_cast(const float* dy, void* dy_c, const float* x, const float* mean, const float* rstd,
                 const float* ln_w, const void* w2t_c, const void* w1t_c, const void* gd, void* du, float* dx,
                 void* dx_c, float* partials, int64_t rows, int D, int M, int dtype, sitk_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Fused LayerNorm + bias-free projection of the attention half of a block, PreNorm(LayerNorm, Attention)
 * up to to_qkv (state-dict keys transformer.layers.i.0.{norm, fn.to_qkv}, utils/utils.py:18-24), one
 * launch per direction.  Specialised: dtype bf16, D == 192, N % 64 == 0 (sitk_ln_gemm_fused_supported).
 *   forward : y = LN(x) W^T          x (rows, D) fp32, w_c (N, D) `dtype`, y (rows, N) `dtype`;
 *             saved (each may be NULL): h = LN(x) (rows, D) `dtype`, mean/rstd (rows) fp32
 *   backward: dx = dres + LN'(dy W)  dy (rows, N) `dtype`, wt_c = W^T (D, N) `dtype`, dres fp32 (may be NULL),
 *             dx fp32 and dx_c its `dtype` copy (may be NULL); per-workgroup dgamma/dbeta sums go to
 *             `partials` (sitk_ln_gemm_bwd_partial_floats(rows) floats, [workgroup][2][D], 128 rows each) */
int sitk_ln_gemm_fused_supported(int D, int N, int dtype);
int sitk_ln_gemm_fwd(const float* x, const float* ln_w, const float* ln_b, const void* w_c, void* h, float* mean,
                     float* rstd, void* y, int64_t rows, int D, int N, int dtype, sitk_stream_t stream);
size_t sitk_ln_gemm_bwd_partial_floats(int64_t rows);
int sitk_ln_gemm_bwd(const void* dy, const void* wt_c, const float* x, const float* mean, const float* rstd,
                     const float* ln_w, const float* dres, float* dx, void* dx_c, float* partials, int64_t rows,
                     int D, int N, int dtype, sitk_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Multi-head self-attention core of vit_pytorch.vit.Attention (dim_head = 64):
 * softmax((q k^T) * scale) v, flash-style (the (B,H,N,N) matrix is never materialised).
 *   qkv (B*N, 3*H*64) `dtype`: columns [q | k | v], each (h d) h-major   (to_qkv + chunk(3))
 *   o   (B*N, H*64) `dtype`, 'b h n d -> b n (h d)'
 *   lse (B, H, N) fp32: log-sum-exp of the scaled scores (natural log), saved for backward.    */
int sitk_attention_fwd(const void* qkv, void* o, float* lse, int B, int N, int H, float scale, int dtype,
                       sitk_stream_t stream);
/* dqkv (B*N, 3*H*64) `dtype` <- gradients of q, k, v.  delta (B,H,N) fp32 is scratch.          */
int sitk_attention_bwd(const void* qkv, const void* o, const void* d_o, const float* lse, float* delta,
                       void* dqkv, int B, int N, int H, float scale, int dtype, sitk_stream_t stream);
/* Attention backward with the to_out backward folded in (vit_pytorch.vit.Attention.to_out[0], key
 * `layers.i.0.fn.to_out.0.weight`, utils/utils.py:26): instead of reading d_o, the query-side kernel forms
 *   d_o = dxmid @ Wo        dxmid (B*N, D) `dtype` = gradient of the block's attention-branch output,
 *                           wo_t (H*64, D) `dtype` = Wo^T (sitk_stage_weight's transposed copy)
 * per 16-query tile from a 24 KB LDS copy of the head's slice of Wo^T, uses it in registers and WRITES it to d_o
 * (B*N, H*64) for the key-side kernel.  Replaces one GEMM launch per block.  bf16, N <= 384, D == 192
 * (`_supported` tells; anything else: sitk_gemm_nt + sitk_attention_bwd).                                        */
int sitk_attention_bwd_proj_supported(int N, int D, int dtype);
int sitk_attention_bwd_proj(const void* qkv, const void* o, const void* dxmid, const void* wo_t, void* d_o,
                            const float* lse, float* delta, void* dqkv, int B, int N, int H, int D, float scale,
                            int dtype, sitk_stream_t stream);

/* One kernel of the backward pair by itself (profiling / bench.py's per-kernel roofline rows): phases bit 0 = the
 * query-side kernel (dQ, delta, and d_o_out when wo_t != NULL), bit 1 = the key-side kernel (dK, dV; needs the delta
 * -- and d_o_out -- a query-side launch left behind).  wo_t == NULL: d_o_in is the attention output's gradient
 * (sitk_attention_bwd); else dxmid / wo_t / d_o_out as in sitk_attention_bwd_proj.  phases = 3 equals those calls.   */
int sitk_attention_bwd_phases(const void* qkv, const void* o, const void* d_o_in, const void* dxmid, const void* wo_t,
                              void* d_o_out, const float* lse, float* delta, void* dqkv, int B, int N, int H, int D,
                              float scale, int dtype, int phases, sitk_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Whole encoder = vit_pytorch.vit.Transformer(dim, depth, heads, dim_head=64, mlp_dim, dropout=0)
 * as constructed at models/sit.py:57 and called at models/sit.py:76 / models/mpp.py:128:
 * depth x [ x += to_out(attn(to_qkv(LN(x)))) ; x += W2 gelu(W1 LN(x) + b1) + b2 ].             */
typedef struct sitk_timeline sitk_timeline;
typedef struct {
  int B, N, dim, depth, heads, mlp_dim; /* dim_head fixed at 64 */
  int dtype;
  sitk_timeline* timeline; /* NULL, or a timeline that gets one mark (HIP event + label) behind every launch of
                              sitk_encoder_fwd / _bwd: per-kernel times of the REAL chain (bench.py's roofline rows)   */
} sitk_encoder_cfg;

/* Profiling aid (never part of a training step: recording an event between two kernels costs about a microsecond and
 * events cannot be timed inside a captured graph).  The library owns the events.  read: waits for the last mark, then
 * us[i] = time from mark i to mark i + 1 and labels[i] = label of mark i + 1 (static strings), i < count - 1; returns
 * the number of intervals written (<= max), < 0 on error.                                                            */
sitk_timeline* sitk_timeline_create(int capacity);
void sitk_timeline_destroy(sitk_timeline* t);
void sitk_timeline_reset(sitk_timeline* t);
int sitk_timeline_mark(sitk_timeline* t, const char* label, sitk_stream_t stream);
int sitk_timeline_read(sitk_timeline* t, float* us, const char** labels, int max);

/* Per-layer fp32 parameter (or gradient) pointers, in state-dict order. */
typedef struct {
  float* ln1_w; /* layers.i.0.norm.weight        (dim)            */
  float* ln1_b; /* layers.i.0.norm.bias          (dim)            */
  float* wqkv;  /* layers.i.0.fn.to_qkv.weight   (3*H*64, dim)    */
  float* wo;    /* layers.i.0.fn.to_out.0.weight (dim, H*64)      */
  float* bo;    /* layers.i.0.fn.to_out.0.bias   (dim)            */
  float* ln2_w; /* layers.i.1.norm.weight        (dim)            */
  float* ln2_b; /* layers.i.1.norm.bias          (dim)            */
  float* w1;    /* layers.i.1.fn.net.0.weight    (mlp_dim, dim)   */
  float* b1;    /* layers.i.1.fn.net.0.bias      (mlp_dim)        */
  float* w2;    /* layers.i.1.fn.net.3.weight    (dim, mlp_dim)   */
  float* b2;    /* layers.i.1.fn.net.3.bias      (dim)            */
} sitk_layer_params;

/* Bytes of the caller-provided workspaces. `acts` holds what backward needs (saved activations
 * + staged weights) and must stay untouched between fwd and bwd; `scratch` is transient.      */
size_t sitk_encoder_acts_bytes(const sitk_encoder_cfg* cfg);
size_t sitk_encoder_scratch_bytes(const sitk_encoder_cfg* cfg);
/* The part of `scratch` reserved as the slab of a backward slice's ONE weight-gradient launch (sitk_gemm_wgrad_group_ws): sized
 * for the worst slice LENGTH -- how many token splits a launch takes depends on how its tiles fill the chip's rounds, so a
 * 4-layer slice of SiT-base needs more slab than all 12 layers -- plus room for the patch embedding's and the caller's extra
 * problems.  0 when the shapes do not take the large-tile path.  (A launch that does not fit falls back to generic tiles:
 * sitk_encoder_bwd* refuse to do that silently.)                                                                          */
size_t sitk_encoder_wgrad_slab_bytes(const sitk_encoder_cfg* cfg);

/* x_in (B*N, dim) fp32 -> x_out (B*N, dim) fp32.  save_for_backward = 0 runs the forward-only
 * (inference) schedule that keeps no activations (acts then only needs the staged weights).
 * save_for_backward bit 1 (value 2 or 3): the compute-dtype weight copies in `acts` are already current -- the caller ran
 * sitk_encoder_stage_weights (e.g. on a side stream, beside the gather and the patch embedding) since the last update. */
int sitk_encoder_stage_weights(const sitk_encoder_cfg* cfg, const sitk_layer_params* params, void* acts, size_t acts_bytes,
                               sitk_stream_t stream);
int sitk_encoder_fwd(const sitk_encoder_cfg* cfg, const sitk_layer_params* params, const float* x_in,
                     float* x_out, void* acts, size_t acts_bytes, void* scratch, size_t scratch_bytes,
                     int save_for_backward, sitk_stream_t stream);

/* dx (B*N, dim) fp32: on entry d(loss)/d(x_out), on exit d(loss)/d(x_in) (in place).
 * grads: fp32, ACCUMULATED into (zero them or keep earlier micro-batches' sums).
 * Layers are processed last to first; [layer_begin, layer_end) selects a slice so that the caller
 * can interleave gradient all-reduce buckets with the rest of backward.                        */
int sitk_encoder_bwd(const sitk_encoder_cfg* cfg, const sitk_layer_params* params,
                     const sitk_layer_params* grads, const float* x_in, float* dx, void* acts,
                     size_t acts_bytes, void* scratch, size_t scratch_bytes, int layer_begin, int layer_end,
                     sitk_stream_t stream);
/* The same with the patch embedding's weight gradient riding along (models/sit.py:50, `to_patch_embedding.1`): when
 * the slice ends at layer 0, `embed` (dW[n][k] += sum_m dY[m][n] X[m][k] with dY = the encoder's input gradient in the
 * compute dtype, row-mapped onto the patch rows) joins the slice's ONE weight-gradient launch instead of being a launch
 * of its own.  embed->dY is ignored on entry: the call writes the compute-dtype copy of d(x_in) to `dx_c` (B*N, dim)
 * and points the problem at it.  Returns SITK_OK with *embed_done = 1 when the problem was taken (bf16, large-tile
 * path), 0 when the caller still has to run sitk_gemm_wgrad itself (then dx_c is not written).                      */
int sitk_encoder_bwd_embed(const sitk_encoder_cfg* cfg, const sitk_layer_params* params,
                           const sitk_layer_params* grads, const float* x_in, float* dx, void* acts,
                           size_t acts_bytes, void* scratch, size_t scratch_bytes, int layer_begin, int layer_end,
                           const sitk_wgrad_desc* embed, void* dx_c, int* embed_done, sitk_stream_t stream);
/* The same with n_extra more weight-gradient problems (fully specified; e.g. to_original of models/mpp.py:66,129, whose
 * operands exist before the encoder's backward starts) taken into that one launch: *extra_done = 1 when they were, 0
 * when the caller has to run them itself (sitk_gemm_wgrad).                                                          */
int sitk_encoder_bwd_extra(const sitk_encoder_cfg* cfg, const sitk_layer_params* params,
                           const sitk_layer_params* grads, const float* x_in, float* dx, void* acts,
                           size_t acts_bytes, void* scratch, size_t scratch_bytes, int layer_begin, int layer_end,
                           const sitk_wgrad_desc* embed, void* dx_c, int* embed_done, const sitk_wgrad_desc* extra,
                           int n_extra, int* extra_done, sitk_stream_t stream);

/* The weight gradients of FINISHED layers beside the rest of backward.  Every kernel of the backward chain is one wave of
 * 192 - 214 workgroups that each own a CU's LDS: 42 - 64 of the 256 CUs idle through the whole chain, while the weight
 * gradients -- which nothing downstream reads before the optimizer -- wait for its end.  With an overlap object the call
 * hands the weight-gradient launch of each of the first `layers` layers it finishes to a SIDE stream (forked and joined
 * through events, so it can be captured into the caller's graph), sized for `cus` CUs (42: measured on MI355X, the chain
 * beside it runs ~4 % longer and hides ~75 % of the side work); the remaining layers' gradients run in one launch behind
 * the chain as before.  The object owns one non-blocking HIP stream and layers + 1 events; one object per engine (not
 * re-entrant: two concurrent calls must not share it).  overlap == NULL: sitk_encoder_bwd_extra.                     */
typedef struct sitk_overlap sitk_overlap;
sitk_overlap* sitk_overlap_create(int max_layers, int cus, int caller_joins);
void sitk_overlap_destroy(sitk_overlap* o);
/* The side stream for the caller's own use: sitk_overlap_fork makes it wait for everything enqueued on `stream` so far,
 * sitk_overlap_join makes `stream` wait for everything enqueued on the side stream so far; between the two the caller
 * passes sitk_overlap_stream(o) as the stream argument of any entry point whose work may run beside `stream`'s (the
 * engine: weight staging beside the gather + patch embedding; d pos_embedding / d cls_token beside the last
 * weight-gradient launch).  caller_joins != 0 at creation: sitk_encoder_bwd_overlap does NOT join at its end -- it
 * leaves the side stream behind the chain's last kernel and its own side launches (the LayerNorm parameter-gradient
 * reduction runs there), and the caller joins (sitk_overlap_join) before anything reads a gradient.                 */
sitk_stream_t sitk_overlap_stream(sitk_overlap* o);
/* how many layers the NEXT sitk_encoder_bwd_overlap call hands to the side stream (0 .. max_layers of the creation; a
 * data-parallel caller sets the whole slice for every slice but the last) */
int sitk_overlap_set_layers(sitk_overlap* o, int layers);
int sitk_overlap_fork(sitk_overlap* o, sitk_stream_t stream);
int sitk_overlap_join(sitk_overlap* o, sitk_stream_t stream);
/* Data parallelism (ABI 10; the reference has none: tools/train.py:72 picks one device).  A call of sitk_encoder_bwd_overlap
 * over layers [layer_begin, layer_end) with s = min(`layers`, layer_end - layer_begin) side layers makes ceil(s / g) side
 * launches (g = 2 layers per launch, or what sitk_overlap_set_group set: 1 .. 3): launch i carries the weight + bias gradients
 * (to_qkv, to_out, net.0, net.3 -- NOT the LayerNorm parameters, which one reduction at the end of the call finishes) of the
 * g layers below layer_end - g i (the last launch holds the s mod g that are left).  Behind each launch and its slab reduction the call records an event on the side stream:
 * sitk_overlap_side_launches = how many the last call made; sitk_overlap_wait_side_launch makes `stream` wait for launch i,
 * behind which those layers' gradients are FINAL -- the caller all-reduces that bucket from `stream` while the chain goes on.
 * sitk_overlap_set_tail_cus: the one weight-gradient launch behind the chain (the layers that did not go to the side stream)
 * is sized for `cus` CUs instead of the chip's 256, so that the all-reduce channels still running beside it keep theirs. */
int sitk_overlap_set_group(sitk_overlap* o, int layers_per_launch);
int sitk_overlap_side_launches(const sitk_overlap* o);
int sitk_overlap_wait_side_launch(sitk_overlap* o, int i, sitk_stream_t stream);
int sitk_overlap_set_tail_cus(sitk_overlap* o, int cus);
/* Stream placement probe (ABI 11; the reference has one device and one stream: tools/train.py:72).  Measured on MI355X / ROCm 7.2
 * (tools/micro/blocked_queue.hip, profiles/r06_dp_streams.txt): a stream that sits BLOCKED behind an event is a barrier packet at
 * the head of its hardware queue, and when that queue shares a dispatch pipe with the queue of the stream that runs a chain of
 * dependent kernels (hardware queues created four apart do), every dispatch of the chain is delayed by ~35 us -- 2.7 -> 6.5 ms
 * for 110 kernels -- whatever the blocked stream does afterwards; a stream that shares the chain's hardware QUEUE runs in line
 * behind it instead.  Which stream lands where follows the creation order of the process's streams, so a data-parallel caller
 * picks the stream it reduces its early buckets from by measurement: this call runs a chain of 128 dependent ~11-us launches
 * (214 workgroups) on `main_stream` beside a helper stream's one-workgroup kernel -- with `candidate` idle (chain_free_us; after
 * one discarded warm-up pass), and with `candidate` blocked behind an event that the helper releases after release_us, its wait
 * issued behind the host's enqueue of the chain as the engine issues its collectives (chain_blocked_us; candidate_done_us = when
 * a small kernel behind that wait finished, from the chain's start).  A good
 * candidate: chain_blocked_us ~ chain_free_us and candidate_done_us ~ release_us (the release falls inside the chain: a candidate
 * that shares the chain's hardware queue finishes behind the chain's END instead).
 * Synchronises the device (construction time only); both streams are idle again on return.                                  */
int sitk_stream_probe(sitk_stream_t main_stream, sitk_stream_t candidate, float* chain_free_us, float* chain_blocked_us,
                      float* candidate_done_us, float* release_us);
int sitk_encoder_bwd_overlap(const sitk_encoder_cfg* cfg, const sitk_layer_params* params,
                             const sitk_layer_params* grads, const float* x_in, float* dx, void* acts,
                             size_t acts_bytes, void* scratch, size_t scratch_bytes, int layer_begin, int layer_end,
                             const sitk_wgrad_desc* embed, void* dx_c, int* embed_done, const sitk_wgrad_desc* extra,
                             int n_extra, int* extra_done, sitk_overlap* overlap, sitk_stream_t stream);

/* Row 0 of every sample of the residual stream: x[b, 0, :] = cls_token + pos_embedding[0, :]
 * (models/sit.py:70-73; rows 1..P come from the patch-embedding GEMM's BIAS_RES epilogue).      */
int sitk_embed_cls_rows(float* x, const float* cls_token, const float* pos, int B, int N, int D,
                        sitk_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Pool + head: models/sit.py:78-82 (x[:,0] or mean over tokens; LayerNorm(dim); Linear(dim, classes)). */
int sitk_head_fwd(const float* x, const float* ln_w, const float* ln_b, const float* w, const float* b,
                  float* logits, int B, int N, int D, int n_classes, int pool_mean, sitk_stream_t stream);
/* dx (B*N, D) fp32 is fully written (zeros outside the pooled rows); parameter grads accumulated.
 * ws: sitk_head_ws_floats(B, D, n_classes) floats of scratch, or NULL.  With it every sample's terms of the parameter
 * gradients (and of the loss) are stored per sample and added in sample order by a second small launch: bitwise
 * reproducible.  NULL: float atomics, whose sum depends on arrival order in the last bits.                            */
size_t sitk_head_ws_floats(int B, int D, int n_classes);
int sitk_head_bwd(const float* x, const float* ln_w, const float* ln_b, const float* w, const float* dlogits,
                  float* dx, float* d_ln_w, float* d_ln_b, float* d_w, float* d_b, int B, int N, int D,
                  int n_classes, int pool_mean, float* ws, sitk_stream_t stream);

/* Regression losses of tools/train.py:245-248 on (n) predictions: loss[0] (+)= mean (p-t)^2 or
 * mean |p-t|; dpred = d loss / d pred.  loss must be zeroed by the caller.                      */
int sitk_loss_fwd_bwd(const float* pred, const float* target, float* loss, float* dpred, int n, int l1,
                      sitk_stream_t stream);
/* head_fwd + loss_fwd_bwd + head_bwd in ONE launch (the regression step between the encoder's forward and backward,
 * tools/train.py:245-248,288-290): logits (B, n_classes) out, loss += mean loss (MSE, or L1 when l1), dx (B*N, D) out
 * = d(loss)/d(x_out) for every row, parameter gradients accumulated.  One workgroup per sample; ws as above.
 * grad_scale (2 floats in device memory, or NULL; needs ws): LOSS SCALING for the f16 compute mode.  The call then runs
 * as two launches: forward + loss + every sample's raw d loss / d logits, then backward with every gradient (dx and the
 * head's parameter gradients) multiplied by S = 2^k, k chosen from THIS batch so that max |d loss / d logits| * S lies in
 * [64, 128); grad_scale[0] = S, grad_scale[1] = 1 / S are written for sitk_*_step_dev.  loss and logits are never scaled. */
int sitk_head_loss_fwd_bwd(const float* x, const float* ln_w, const float* ln_b, const float* w, const float* b,
                           const float* target, float* logits, float* loss, float* dx, float* d_ln_w, float* d_ln_b,
                           float* d_w, float* d_b, int B, int N, int D, int n_classes, int pool_mean, int l1, float* ws,
                           float* grad_scale, sitk_stream_t stream);
/* sitk_head_loss_fwd_bwd without the reduction of the workspace rows (ws is required): dx -- all that the backward chain
 * waits for -- is complete when it returns; the head's parameter gradients and the loss are complete only after
 * sitk_head_finalize(ws, ...) has run, on any stream ordered behind this call (the engine: the side stream, beside the chain).
 * Same bits as the one-call form (ABI 9). */
int sitk_head_loss_fwd_bwd_deferred(const float* x, const float* ln_w, const float* ln_b, const float* w, const float* b,
                                    const float* target, float* logits, float* dx, int B, int N, int D, int n_classes,
                                    int pool_mean, int l1, float* ws, float* grad_scale, sitk_stream_t stream);
int sitk_head_finalize(const float* ws, int B, int D, int n_classes, float* d_ln_w, float* d_ln_b, float* d_w, float* d_b,
                       float* loss, sitk_stream_t stream);

/* column sums: out[c] += sum_r in[r][c]  (d_pos_embedding / d_cls_token over the batch).  Up to 512 rows (4 096 when
 * cols >= 4 096) one workgroup per column group sums all rows in a fixed order (bitwise reproducible; out must not be
 * written concurrently); taller inputs are split over workgroups that add their partial sums with float atomics. */
int sitk_colsum_f32(const float* in, int64_t rows, int cols, int ld, float* out, sitk_stream_t stream);
/* the same, with the first cols2 sums also added to out2 (models/sit.py:70-73: d cls_token = the token-0 part of
 * d pos_embedding) */
int sitk_colsum_f32_dup(const float* in, int64_t rows, int cols, int ld, float* out, float* out2, int cols2,
                        sitk_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Masked patch pre-training, models/mpp.py:85-112 and :132.
 *   tokens (B*P, K) fp32 clean tokens; corrupted (B*P, ld) `dtype` (pad zeroed)
 *   masked / swap_draw / replace_draw (B*P) uint8; random_patches (B*P) int32 in [0, P)
 *   swap = masked & swap_draw (source row: clean tokens of random_patches, same sample);
 *   replace = masked & replace_draw (applied second, wins): row = mask_token (K) fp32.          */
int sitk_mpp_corrupt(const float* tokens, const uint8_t* masked, const uint8_t* swap_draw,
                     const int32_t* random_patches, const uint8_t* replace_draw, const float* mask_token,
                     void* corrupted, int B, int P, int K, int ld, int dtype, sitk_stream_t stream);

/* The engine's form of the same step (no host-drawn tensors, nothing but the surfaces is read twice):
 * sitk_mpp_draw fills masked / swap_draw / random_patches / replace_draw (shapes as above) and replaced_full (B, P + 1) uint8
 * (masked & replace_draw at token p + 1, 0 at the cls token) from a Philox4x32-10 stream; state = 2 x uint64 in device memory
 * {seed, draws so far}.  masked has EXACTLY n_mask ones per sample (the n_mask largest of P uniform scores: models/mpp.py:25-33);
 * swap_draw = U < p_swap, random_patches uniform in [0, P), replace_draw = U < p_replace (models/mpp.py:36-43,95-110).  Same
 * distribution as the reference's draws, not the same stream: parity tests replay the reference's captured tensors through
 * sitk_mpp_corrupt instead.  P <= 2048.
 * sitk_mpp_gather_corrupt (C == 4 only: every shipped configuration; other channel counts take sitk_gather_tokens* +
 * sitk_mpp_corrupt) = sitk_gather_tokens_idx (sample_idx / mean / stdv may be NULL) + sitk_mpp_corrupt in one pass:
 * clean (B*P, V*C) fp32 and corrupted (B*P, ld) `dtype` are both written; when state != NULL it also advances state[1],
 * so the next sitk_mpp_draw (e.g. the next replay of a captured graph) draws fresh masks.                                  */
int sitk_mpp_draw(const uint64_t* state, uint8_t* masked, uint8_t* swap_draw, int32_t* random_patches, uint8_t* replace_draw,
                  uint8_t* replaced_full, int B, int P, int n_mask, float p_swap, float p_replace, sitk_stream_t stream);
int sitk_mpp_gather_corrupt(const float* x, const uint16_t* table_pv, const int32_t* sample_idx, const float* mean,
                            const float* stdv, const uint8_t* masked, const uint8_t* swap_draw, const int32_t* random_patches,
                            const uint8_t* replace_draw, const float* mask_token, float* clean, void* corrupted,
                            uint64_t* state, int B, int n_vertices, int C, int P, int V, int ld, int dtype,
                            sitk_stream_t stream);
/* loss[0] += sum_{masked rows} (out - tokens)^2 / (n_masked_total * K); dout likewise (0 elsewhere). */
int sitk_mpp_loss_fwd_bwd(const float* out, const float* tokens, const uint8_t* masked, float* loss,
                          float* dout, int64_t rows, int K, int64_t n_masked_total, sitk_stream_t stream);
/* the same over row-padded buffers (leading dimensions ldo / ldt / lddo) with the gradient in `dout_dtype` (engine path),
 * multiplied by grad_scale (loss scaling of the f16 mode: the masked mean divides by rows * K ~ 1e7; 1 otherwise)      */
int sitk_mpp_loss_fwd_bwd_ld(const float* out, int ldo, const float* tokens, int ldt, const uint8_t* masked, float* loss,
                             void* dout, int lddo, int dout_dtype, int64_t rows, int K, int64_t n_masked_total,
                             float grad_scale, sitk_stream_t stream);
/* out[c] += sum over rows with flag[r] != 0 of in[r][c]   (mask_token gradient, stage 1) */
int sitk_masked_colsum(const void* in, int ld, int in_is_f32, int dtype, const uint8_t* flag_a,
                       const uint8_t* flag_b, int64_t rows, int cols, float* out, sitk_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Dropout and GELU as stand-alone fp32 elementwise kernels: the UNFUSED encoder path the Python mirror takes when the
 * reference's `dropout` constructor argument (models/sit.py:36,57 -> vit_pytorch Attention.to_out.1, FeedForward.net.2 /
 * net.4) is > 0 in training mode.  Every reference configuration uses 0.0 (config/SiT/training/hparams.yml:46).
 *   dropout_fwd: y = res + x * keep / (1 - p) (res may be NULL), keep ~ Bernoulli(1 - p) from a Philox4x32-10 stream
 *                (state = {seed, draws so far} in device memory, advanced by the call); mask (n) uint8 kept for backward.
 *   dropout_bwd: dx = dy * mask / (1 - p).       gelu: exact erf (nn.GELU()).                               */
int sitk_dropout_fwd(const float* x, const float* res, float* y, uint8_t* mask, int64_t n, float p, uint64_t* state,
                     sitk_stream_t stream);
int sitk_dropout_bwd(const float* dy, const uint8_t* mask, float* dx, int64_t n, float p, sitk_stream_t stream);
int sitk_gelu_fwd(const float* u, float* g, int64_t n, sitk_stream_t stream);
int sitk_gelu_bwd(const float* dg, const float* u, float* du, int64_t n, sitk_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Optimizer of tools/train.py:228-243,291: SGD(momentum, weight_decay, nesterov) / Adam / AdamW
 * over one flat fp32 parameter buffer.  grad_scale multiplies the gradient first (1/world).    */
int sitk_sgd_step(float* param, const float* grad, float* momentum_buf, int64_t n, float lr, float momentum,
                  float weight_decay, int nesterov, float grad_scale, sitk_stream_t stream);
int sitk_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                   float beta1, float beta2, float eps, float weight_decay, int decoupled_wd, int step,
                   float grad_scale, sitk_stream_t stream);

/* The same optimizers with their step-dependent scalars in DEVICE memory, so that a captured hipGraph follows the
 * learning-rate schedulers of tools/pretrain.py:42-50 (StepLR, ReduceLROnPlateau, warm-up) and Adam's step count
 * (tools/train.py:228-241): state = 4 doubles {lr, beta1^t, beta2^t, t}; the caller initialises {lr, 1, 1, 0} and rewrites
 * state[0] to change the learning rate.  sitk_adam_step_dev advances t and the two powers before it uses them.
 * zero_grad = 1 folds the next step's optimizer.zero_grad() (tools/train.py:288) into this pass: every consumed gradient
 * and the n_extra accumulator floats stored behind them (grad + n) are overwritten with zeros; the accumulator with
 * index keep_idx (the step's loss; < 0: none) is copied to keep_dst first.  With n_extra > 0 the accumulators start at
 * grad + n and are cleared in 16-byte pieces: n % 4 == 0 is required (the engine pads every parameter to 64 floats).
 * inv_loss_scale (device pointer or NULL): the gradients are additionally multiplied by *inv_loss_scale, the 1 / S that
 * sitk_head_loss_fwd_bwd left behind (f16 mode), read on the device so that a captured graph follows it.
 * nonfinite (device int or NULL; ABI 9, per element since ABI 10): NULL = the reference's behaviour (tools/train.py:291 has no
 * guard: a non-finite gradient reaches the parameters).  Not NULL (the engine: loss-scaled f16 mode only): a gradient ELEMENT that
 * is not finite -- an f16 intermediate that overflowed behind the loss scale -- never reaches the parameters or the optimizer
 * state: it is skipped, zeroed like every consumed gradient, and counted in *nonfinite (atomic add; the caller polls it when it
 * likes).  Adam's step count advances whether or not elements were skipped.
 * grad2 (ABI 12; NULL = none): a SECOND gradient buffer of the same layout (n gradients + n_extra accumulators) -- the other half of
 * a batch that ran as two concurrent half-batch steps (engine.SplitTrainEngine): the pass consumes grad * grad_scale *
 * *inv_loss_scale + grad2 * grad_scale * *inv_loss_scale2 (each half has its own loss scale in f16 mode), clears BOTH buffers
 * (zero_grad), and copies (grad[n + keep_idx] + grad2[n + keep_idx]) * keep_scale to keep_dst (the batch loss = the mean of the
 * halves' losses: keep_scale 1 / 2; without grad2 pass 1).                                                               */
int sitk_sgd_step_dev(float* param, float* grad, float* momentum_buf, int64_t n, const double* state, float momentum,
                      float weight_decay, int nesterov, float grad_scale, int zero_grad, int64_t n_extra,
                      int64_t keep_idx, float* keep_dst, const float* inv_loss_scale, int* nonfinite, float* grad2,
                      const float* inv_loss_scale2, float keep_scale, sitk_stream_t stream);
int sitk_adam_step_dev(float* param, float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, double* state,
                       float beta1, float beta2, float eps, float weight_decay, int decoupled_wd, float grad_scale,
                       int zero_grad, int64_t n_extra, int64_t keep_idx, float* keep_dst, const float* inv_loss_scale,
                       int* nonfinite, float* grad2, const float* inv_loss_scale2, float keep_scale, sitk_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SITK_H */

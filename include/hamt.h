/*
 * hamt.h -- C ABI of libhamt_hip.so: hand-written HIP (gfx950 / MI355X) kernels for the HAMT
 * history-aware multimodal transformer forward/backward.
 *
 * The reference (cshizhe/VLN-HAMT) is pure Python: its "interface" for this path is the torch.nn
 * class surface of pretrain_src/model/{vilmodel,pretrain_cmt}.py.  That surface is re-declared in
 * the vln-hamt_amd/model package; every forward there lowers to the entry points below through ctypes
 * (vln-hamt_amd/_lib.py).  Each entry point names the reference op group it replaces
 * (file:line relative to /root/reference; IDs A1..A24 are SURVEY.md section 8a).
 *
 * Conventions
 *  - plain C: raw device pointers, sizes, a hipStream_t passed as void*; no torch types.
 *  - return 0 on success, negative hamt_status otherwise; hamt_last_error() gives the text
 *    (thread-local).  No C++ exception crosses the boundary.
 *  - asynchronous on the given stream; no allocation, no synchronisation, no global mutable state:
 *    every scratch buffer is passed in by the caller (hipGraph-capture safe).
 *  - row-major tensors; "ld*" are leading dimensions in ELEMENTS; every row must start 16-byte
 *    aligned (ld * sizeof(elem) % 16 == 0).
 *  - dropout is counter based: mask(idx) = hash(rng[0] (seed), rng[1] (epoch), call_id, idx) with
 *    `rng` a DEVICE pointer to two uint64 (so a captured graph draws fresh masks every replay when
 *    the epoch word is bumped) and call_id a host value unique per call site; backward replays the
 *    mask from the same triple, nothing is stored.
 */
#ifndef HAMT_H
#define HAMT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HAMT_ABI_VERSION 2   /* 2 (round 6): hamt_gemm_ln_fwd / hamt_graph_split_* removed (round 5); hamt_embed_sum_bwd takes V and ws_bytes,
                              * hamt_scatter_add_rows_ordered takes T; hamt_wgrad_grouped_ex and the dtype HAMT_F16 added */

typedef enum { HAMT_OK = 0, HAMT_ERR_ARG = -1, HAMT_ERR_UNSUPPORTED = -2, HAMT_ERR_LAUNCH = -3 } hamt_status;
typedef enum { HAMT_F32 = 0, HAMT_BF16 = 1,
               /* one byte per element, `aux` of HAMT_EPI_GELU_GRAD (without HAMT_EPI_DROPOUT) / HAMT_EPI_MUL_AUX only: the saved gelu'(x),
                * whose range is [-0.129, 1.129], as code q in [0, 255] with value = 0.005 q - 0.13 (0 and 1 are exact: q = 26 / 226;
                * |error| <= 0.0025, the size of bf16's rounding of values near 1).  Halves the bytes of the image the FFN-1 epilogue
                * writes next to gelu(x) and the FFN-2 dgrad epilogue reads back (vilmodel.py:168-175 BertIntermediate backward). */
               HAMT_U8G = 2,
               /* IEEE half, for `C` of hamt_gemm (epilogue: none / HAMT_EPI_BIAS / HAMT_EPI_ACCUM) and for `x` / `z` of the LayerNorm family only: the output of a dense layer in front of a
                * LayerNorm (vilmodel.py:139-143, 181-185) kept in two bytes with 10 mantissa bits instead of bf16's 7 -- the rounding of
                * that interface is 8 x finer at the same HBM bytes (round 6: the bf16 head outputs at B = 64 drop from 1.03e-2 to
                * ~6.5e-3 of the fp32 reference; dense outputs are O(1 .. 100), far inside half's range) */
               HAMT_F16 = 3 } hamt_dtype;
/* arithmetic of the contraction: bf16 MFMA operands with fp32 accumulate, or exact fp32 MFMA */
typedef enum { HAMT_PREC_BF16 = 0, HAMT_PREC_F32 = 1 } hamt_prec;

int hamt_version(void);
/* copies the calling thread's last error text into buf (NUL terminated); returns its length */
int hamt_last_error(char* buf, size_t n);
/* name (as rocprofv3 prints it, template arguments included) of the kernel the calling thread's most recent hamt_gemm /
 * hamt_gemm_ws call launched for its contraction -- lets a profiler attribute recorded calls to kernel-trace rows */
int hamt_last_kernel(char* buf, size_t n);
/* Caller-provided scratch sizes (the library never allocates).  shape[] by op:
 *   HAMT_WS_GEMM_SPLITK  {M, N, K}  bytes hamt_gemm_ws can use for its deterministic split-K (0: it will not split)
 *   HAMT_WS_COLSUM       {M, N}     hamt_colsum `ws`
 *   HAMT_WS_SUMSQ        {}         hamt_sumsq `ws`
 *   HAMT_WS_LN_BWD       {M, H}     hamt_ln_bwd / hamt_ln_bwd_add `ws`
 *   HAMT_WS_WGRAD_TABLE  {M_0, M_1, ...}  (output rows of every problem) hamt_wgrad_grouped `table`
 *   HAMT_WS_LNRED_TABLE  {n}        hamt_ln_bwd_reduce_grouped `table`
 *   HAMT_WS_VIS_EMBED_BWD {M, H}    hamt_vis_embed_bwd `ws`
 *   HAMT_WS_EMBED_BWD     {R, H}    hamt_embed_sum_bwd `ws`
 * returns the size in bytes, or 0 for an unknown op / malformed shape */
enum { HAMT_WS_GEMM_SPLITK = 0, HAMT_WS_COLSUM = 1, HAMT_WS_SUMSQ = 2, HAMT_WS_LN_BWD = 3, HAMT_WS_WGRAD_TABLE = 4, HAMT_WS_LNRED_TABLE = 5, HAMT_WS_VIS_EMBED_BWD = 6,
       HAMT_WS_EMBED_BWD = 7 /* {R, H}: hamt_embed_sum_bwd `ws` = max(HAMT_WS_COLSUM {R, H}, 136 R bytes) */ };
size_t hamt_workspace_bytes(int op, const int* shape, int nshape);

/* ------------------------------------------------------------------------------------------------
 * GEMM  C[M,N] = epilogue( A[M,K] * B[K,N] )            (all nn.Linear calls of A2-A5, A7, A10, A11,
 * A15-A20 forward; their dgrad dX = dY*W and wgrad dW = dY^T*X backward)
 *   a_kmajor = 0: A stored [M][K] (K contiguous)      1: A stored [K][M] (M contiguous)
 *   b_kmajor = 0: B stored [N][K] (nn.Linear weight)  1: B stored [K][N]
 *   forward  y = x W^T      : a_kmajor 0, b_kmajor 0      (vilmodel.py:97-99, 140, 169, 182 ...)
 *   dgrad    dx = dy W      : a_kmajor 0, b_kmajor 1
 *   wgrad    dW = dy^T x    : a_kmajor 1, b_kmajor 1
 * epilogue flags (applied in this order): v = alpha*acc; +bias[n]; store pre-activation to aux
 * (HAMT_EPI_SAVE_PRE); GELU(erf) or ReLU; multiply by gelu'(aux[m,n]) / relu'(aux[m,n])
 * (HAMT_EPI_MUL_DGELU / _DRELU: fused activation backward for dgrad); C += v (HAMT_EPI_ACCUM).
 * ---------------------------------------------------------------------------------------------- */
enum {
  HAMT_EPI_BIAS = 1, HAMT_EPI_GELU = 2, HAMT_EPI_RELU = 4, HAMT_EPI_ACCUM = 8,
  HAMT_EPI_MUL_DGELU = 16, HAMT_EPI_MUL_DRELU = 32, HAMT_EPI_SAVE_PRE = 64,
  /* bf16 training path: forward stores gelu(v) to C and gelu'(v) to aux in ONE erf/exp evaluation (A&S 7.1.26 erf,
   * |err| <= 1.5e-7, far below bf16 resolution); backward just multiplies by aux. */
  HAMT_EPI_GELU_GRAD = 128, HAMT_EPI_MUL_AUX = 256,
  /* C = epi(...) + aux  (residual add of the pre-LN ViT blocks, vision_transformer.py:196-197; aux fp32 or bf16 [M][ldaux]) */
  HAMT_EPI_ADD_AUX = 512,
  /* v = dropout(v) after the activation and before ADD_AUX (the proj_drop / Mlp.drop of the ViT blocks, vision_transformer.py:
   * 148-150, 176-177): keep factor of element (m, n) = drop_mask(rng, call_id, m, n >> 2)[n & 3], the same mask
   * hamt_cast_pad_bf16_dropout applies in backward.  With GELU_GRAD the stored gelu' is masked too (so backward's
   * MUL_AUX needs no second mask).  bf16 fast path only (bf16 operands, K % 64 == 0); anything else is an error. */
  HAMT_EPI_DROPOUT = 1024
};
typedef struct {
  int M, N, K;
  int lda, ldb, ldc, ldaux;
  int a_kmajor, b_kmajor;
  int dtype_a, dtype_b, dtype_c, dtype_aux; /* hamt_dtype */
  int prec;                                 /* hamt_prec  */
  int epilogue;                             /* HAMT_EPI_* */
  float alpha;
  int ka_rows, kb_rows; /* K-strided operands only: number of valid reduction rows actually stored in A / B when K was
                           rounded up for the other operand (0 = K).  Rows beyond are never dereferenced. */
  float p_drop;         /* HAMT_EPI_DROPOUT only */
  uint32_t call_id;
  const uint64_t* rng;  /* DEVICE pointer: {seed, epoch} */
} hamt_gemm_desc;
int hamt_gemm(const hamt_gemm_desc* d, const void* A, const void* B, void* C, const float* bias,
              void* aux, void* stream);
/* Same, with a caller-provided fp32 workspace that enables deterministic split-K for small outputs with a long
 * reduction (weight gradients: dW[768,768] = dY^T X over K = B*L rows).  hamt_gemm_ksplit() tells how many K
 * slices S the kernel would use for `d`; pass ws_bytes >= S*M*N*4 (less => fewer slices; NULL => no split). */
int hamt_gemm_ksplit(const hamt_gemm_desc* d);
int hamt_gemm_ws(const hamt_gemm_desc* d, const void* A, const void* B, void* C, const float* bias,
                 void* aux, void* ws, size_t ws_bytes, void* stream);

/* operand preparation for the bf16 fast path (both GEMM operands bf16, K-contiguous, K % 64 == 0):
 *   cast_pad_bf16:   y[Rpad][Cpad] (bf16) = x[R][C] (fp32), rows >= R and columns >= C zero filled
 *   cast_transpose:  y[C][Rpad] (bf16) = x[R][C]^T (fp32 or bf16 source), rows >= R zero filled
 * (dgrad uses the transposed weight, wgrad the transposed activations / gradients.) */
int hamt_cast_pad_bf16(int R, int C, int Rpad, int Cpad, const float* x, int ldx, void* y, int ldy, void* stream);
int hamt_cast_transpose(int R, int C, const void* x, int ldx, int dtype_x, void* y, int ldy, int Rpad, void* stream);
/* y[Rpad][C] (bf16) = x[R][C] (fp32) * keep(m, n) with the mask of a HAMT_EPI_DROPOUT GEMM of the same (rng, call_id):
 * the gradient of the dropped branch, ready as dgrad / wgrad operand.  C % 8 == 0; rows >= R zero filled. */
int hamt_cast_pad_bf16_dropout(int R, int C, int Rpad, const float* x, int ldx, void* y, int ldy, float p, uint32_t call_id,
                               const uint64_t* rng, void* stream);

/* dW[n][k] (+)= sum_m dy[m][n] * x[m][k] for K <= 8 (weight gradient of the 4-wide angle-feature linears, vilmodel.py:498,
 * 550, 558): exact fp32, K weighted column sums.  ws: >= 64*N*K floats */
int hamt_smallk_wgrad(int M, int N, int K, const float* dy, int lddy, const float* x, int ldx, float* dW,
                      int accumulate, float* ws, void* stream);

/* Grouped weight gradients: n independent problems in as few launches as the kernarg table allows,
 *     dW_p[M_p][N_p] (+)= dY_p^T X_p      and, when db != NULL,      db_p[M_p] (+)= column sums of dY_p,
 * with dY_p bf16 [K_p][ldy] (its M_p columns = the layer's output features) and X_p bf16 [K_p][ldx] (the layer's input
 * image), both exactly as the forward / dgrad GEMMs left them (K_p = rows padded to a multiple of 64; the padding rows are
 * zero, or of any content when K_valid names the valid rows).
 * Problems are packed into one launch per tile class (256-square tiles when the operand rows allow, else 128 / 64 rows).
 * This is what torch.autograd does one nn.Linear at a time in the reference (vilmodel.py: every nn.Linear backward);
 * weight gradients are not on the backward critical path, so the host queues them and hands the whole list over once per
 * backward pass: 768x768 outputs that alone fill 36 CUs become one chip-filling grid without split-K.
 * `probs` is a HOST array.  Requirements per problem: K % 64 == 0, ldy % 8 == 0 && ldy >= 64, ldx % 8 == 0 && ldx >= 128,
 * dy / x 16-byte aligned.  Deterministic (fixed summation order per output element). */
typedef struct {
  const void* dy;
  const void* x;
  float* dw;
  float* db; /* may be NULL */
  int M, N, K, ldy, ldx, ldw;
  int accum_dw, accum_db; /* 0: store, 1: += */
  float* ss; /* may be NULL.  ceil(M / 64) * ceil(N / 128) floats, ZERO on entry: every output tile stores the sum of squares of the
              * values it wrote (after accumulation) into the slot of its origin [row / 64][column / 128] -- a tile of any size
              * owns exactly one of these slots, the others stay zero -- so that sum(ss) = ||dW||^2 without reading dW back
              * (the global-norm clip of the step: hamt_sumsq_partials + hamt_sumsq_table over the remaining parameters).
              * Only meaningful for a dW written ONCE in the call sequence (a second, accumulating problem on the same dW
              * must use the same tiling to overwrite the same slots: pass NULL for both and reduce that parameter from memory) */
  int K_valid; /* 0 or K: every reduction row counts.  Else rows [K_valid, K) of dy and x are PADDING of any content
                * (uninitialised memory included): they are neither read (the loads re-read row K_valid - 1) nor multiplied */
  float wire_scale; /* 0: dW is stored as fp32 (the default).  != 0: `dw` points to a BF16 array of the same shape / ldw and the
                     * tile stores bf16(wire_scale * dW) -- the value a data-parallel gradient exchange puts on the wire (the stock DDP
                     * bf16_compress_hook's: divide by the world size, round to bf16), written straight from the accumulators instead
                     * of an fp32 store followed by a pack pass over the arena.  Store only (accum_dw must be 0), `ss` is ignored. */
  /* Optional SECOND pair of operands reduced into the same dW / db in the same launch: dW (+)= dy^T x + dy2^T x2 (a parameter used
   * twice in a pass -- the cross-attention weights LXRTXLayer shares between its two directions, vilmodel.py:401-412 -- as ONE problem
   * instead of a second, accumulating launch behind the first).  dy2 == NULL: none.  Same requirements as dy / x (K2 % 64 == 0, ...);
   * K_valid must be 0 or K (no ragged tail between the two reductions), K2_valid as K_valid; wire_scale must be 0. */
  const void* dy2;
  const void* x2;
  int K2, ldy2, ldx2, K2_valid;
} hamt_wgrad_desc;
/* `table`: caller-provided DEVICE scratch (16-byte aligned) that holds the launch table: HAMT_WGRAD_TABLE_ENTRY bytes per
 * entry, at most sum over the problems of ceil(M_p / 64) entries (large problems are cut into bands of tile rows); it is filled by small kernels from kernarg data, so `probs` need not outlive the call and the whole sequence can
 * be captured in a hipGraph.  The table must stay untouched until the launches have run. */
#define HAMT_WGRAD_TABLE_ENTRY 112
int hamt_wgrad_grouped(int n, const hamt_wgrad_desc* probs, void* table, size_t table_bytes, void* stream);
/* The same in two phases, for callers that replay a FIXED set of problems (a captured training step): phase 1 writes the launch table
 * (once: its content is a function of `probs` only), phase 2 launches the grouped kernels from a table a phase-1 call with the same
 * `probs` has written, phase 3 = both = hamt_wgrad_grouped.  A replayed phase-2 launch has no one-workgroup table-write kernel in front
 * of it: on a second stream such a kernel waited ~275 us behind the first stream's chip-filling tiles before it was dispatched. */
int hamt_wgrad_grouped_ex(int n, const hamt_wgrad_desc* probs, void* table, size_t table_bytes, int phase, void* stream);

/* column sums  out[n] (+)= sum_m x[m,n]   (bias gradients of every nn.Linear).  ws: >= 64*N floats */
int hamt_colsum(int M, int N, const void* x, int ldx, int dtype_x, float* out, int accumulate,
                float* ws, void* stream);

/* ------------------------------------------------------------------------------------------------
 * attn_small: multi-head softmax attention for short sequences (A2 core vilmodel.py:101-126,
 * A7 core vilmodel.py:327-348): S = Q K^T * scale + mask[b, key]; P = softmax(S); P = dropout(P);
 * O = P V, heads split/merged by pointer arithmetic (head h = columns [h*64, h*64+64)).
 * Q rows (b*Sq + i) at q + row*ldq, K/V rows (b*Sk + j); mask is the reference's additive
 * (1-m)*-10000 row, fp32 [B, Sk] (may be NULL).  d_head must be 64.  lse[B,heads,Sq] (fp32) is
 * saved for backward.  Backward recomputes P (flash style) and needs delta = rowsum(dO*O)
 * computed by hamt_attn_small_bwd itself into `delta` ([B,heads,Sq] fp32 scratch).
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  int B, heads, Sq, Sk, d_head;
  int ldq, ldk, ldv, ldo;
  int dtype_qkv, dtype_o; /* hamt_dtype */
  float scale, p_drop;
  uint32_t call_id;
  int prec; /* hamt_prec: HAMT_PREC_BF16 = bf16 MFMA products with fp32 softmax; HAMT_PREC_F32 = exact fp32 MFMA */
} hamt_attn_desc;
int hamt_attn_small_fwd(const hamt_attn_desc* d, const void* q, const void* k, const void* v,
                        const float* add_mask, void* o, float* lse, const uint64_t* rng, void* stream);
/* dq/dk/dv have the layout (ld, dtype) of q/k/v; do has the layout of o */
int hamt_attn_small_bwd(const hamt_attn_desc* d, const void* q, const void* k, const void* v,
                        const float* add_mask, const void* o, const void* d_o, const float* lse,
                        float* delta, void* dq, void* dk, void* dv, const uint64_t* rng, void* stream);

/* Packed ("varlen") self-attention, bf16 path: the B sequences lie back to back, sample b owns rows [cu_seqlens[b], cu_seqlens[b + 1])
 * of q / k / v / o / d_o / dq / dk / dv (row strides as in the descriptor), at most d->Sq = d->Sk <= 128 tokens each, every key real (no
 * mask); lse is [B, heads, d->Sq].  = hamt_attn_small_* on the real tokens of a padded batch (vilmodel.py:96-129) without computing
 * the padded rows; a sequence of length 0 is skipped. */
int hamt_attn_varlen_fwd(const hamt_attn_desc* d, const void* q, const void* k, const void* v, const int* cu_seqlens,
                         void* o, float* lse, const uint64_t* rng, void* stream);
int hamt_attn_varlen_bwd(const hamt_attn_desc* d, const void* q, const void* k, const void* v, const int* cu_seqlens,
                         const void* o, const void* d_o, const float* lse, void* dq, void* dk, void* dv,
                         const uint64_t* rng, void* stream);
/* Cross attention with ONE side packed (BertXAttention, vilmodel.py:351-360, in the x-layers of a ragged batch: the instruction tokens
 * lie back to back, the visual stream keeps its fixed stride).  Exactly one of cu_q / cu_k is given.
 *   cu_q: query sequence b = rows [cu_q[b], cu_q[b + 1]) of q / o / d_o / dq (<= d->Sq rows), its keys = rows [b * Sk, (b + 1) * Sk)
 *         of k / v under add_mask [B, Sk] (may be NULL).  Query sequences b >= n_pairs -- the filler sequences that round a packed
 *         row count up to its bucket -- have no keys: o = 0, dq = 0, nothing added to dk / dv.
 *   cu_k: queries at the fixed stride d->Sq, sample b's keys = rows [cu_k[b], cu_k[b + 1]) of k / v / dk / dv (<= d->Sk, all real:
 *         add_mask is ignored); n_pairs = d->B.  Key rows no sample owns are not written: zero dk / dv behind the last real row.
 * `pair` (optional device array [B]; NULL = "b pairs with b, fillers behind n_pairs"): the key side of query sequence b -- the key sample
 * (cu_q form: rows [pair[b] * Sk, ...) and row pair[b] of add_mask) or the entry of cu_k (cu_k form); negative = a filler.  Every key
 * side must be named by at most one query sequence (dk / dv are stored, not accumulated).  For packed copies of a batch whose fillers lie
 * between the copies (forward_itm's replicated text, vilmodel.py:672-676).
 * lse is [B, heads, d->Sq] in both forms. */
int hamt_attn_varlen_cross_fwd(const hamt_attn_desc* d, const void* q, const void* k, const void* v, const int* cu_q,
                               const int* cu_k, int n_pairs, const int* pair, const float* add_mask, void* o, float* lse,
                               const uint64_t* rng, void* stream);
int hamt_attn_varlen_cross_bwd(const hamt_attn_desc* d, const void* q, const void* k, const void* v, const int* cu_q,
                               const int* cu_k, int n_pairs, const int* pair, const float* add_mask, const void* o, const void* d_o,
                               const float* lse, void* dq, void* dk, void* dv, const uint64_t* rng, void* stream);

/* ------------------------------------------------------------------------------------------------
 * vis_embed: the two-stream visual embedding   e = LN_img(x1) + LN_ang(ang W_ang^T + b_ang)
 *   ImageEmbeddings / HistoryEmbeddings (A10, A11): img_layer_norm(img_linear(img)) + ang_layer_norm(ang_linear(ang))
 *   (vilmodel.py:498-500, 549-551, 557-558; finetune vilmodel_cmt.py:575-578, 585-586); x1 [M, H] = img_linear's output
 *   (bf16 or fp32), ang [M, 4] fp32 with row stride ld_ang, w_ang [H, 4] / b_ang [H] = ang_linear's parameters.
 * fwd: y [M, H] fp32, y16 (optional) its bf16 image [Mpad16, H] (rows [M, Mpad16) zero), stats [4][M] = mean / rstd of the image
 *   stream, mean / rstd of the angle stream (for backward).
 * bwd: dy [M, H] -> dx = d(x1) as fp32 [M, H] and / or dx16 = its bf16 image [Mpad16, H] (the operand of img_linear's weight
 *   gradient; rows [M, Mpad16) zero); the gradients of both LayerNorms' gamma / beta, of b_ang and of w_ang are ADDED to the
 *   six output vectors (dbeta_img or dbeta_ang may be NULL).  ws: HAMT_WS_VIS_EMBED_BWD {M, H} bytes of scratch.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  int M, H;
  int A;      /* angle features: 4 (the only size the reference uses and the only one built) */
  int ld_ang; /* row stride of ang, in floats (a multiple of 4) */
  float eps1, eps2;
  int x_bf16; /* x1 is bf16 (else fp32) */
  int Mpad16; /* rows of the optional bf16 images y16 / dx16 (0: none beyond M) */
} hamt_vis_embed_desc;
int hamt_vis_embed_fwd(const hamt_vis_embed_desc* d, const void* x1, const float* ang, const float* w_ang, const float* b_ang,
                       const float* gamma_img, const float* beta_img, const float* gamma_ang, const float* beta_ang,
                       float* y, void* y16, float* stats, void* stream);
int hamt_vis_embed_bwd(const hamt_vis_embed_desc* d, const float* dy, const void* x1, const float* ang, const float* w_ang,
                       const float* b_ang, const float* gamma_img, const float* gamma_ang, const float* stats, float* dx, void* dx16,
                       float* dgamma_img, float* dbeta_img, float* dgamma_ang, float* dbeta_ang, float* db_ang, float* dw_ang,
                       float* ws, void* stream);

/* ------------------------------------------------------------------------------------------------
 * ln: y = dropout_post( LayerNorm( dropout_pre(x) + residual ) )      fp32 statistics
 *   BertSelfOutput / BertOutput (A3/A5, vilmodel.py:139-143, 181-185): p_pre = p, residual, p_post = 0
 *   BertEmbeddings / Image / History embeddings (A1, A10, A11):        p_pre = 0, p_post = p
 *   prediction heads (pretrain_cmt.py:13-71): LN followed by Dropout:  p_post = p
 * z (the pre-LN sum) is written for backward (may alias x).  y16 (bf16 copy of y) optional.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  int M, H;
  float eps, p_pre, p_post;
  uint32_t call_id;
  int Mpad16; /* rows of the optional bf16 images (y16 / dx16); rows [M, Mpad16) are written as zeros (0: no padding) */
  int io16;   /* HAMT_LN_X_BF16: `x` is bf16 (the dense layer in front wrote bf16, as a linear does under autocast);
               * HAMT_LN_Z_BF16: the saved pre-LN sum `z` is stored / read as bf16 (backward re-normalises it with the exact
               * fp32 mean / rstd).  HAMT_LN_X_F16 / HAMT_LN_Z_F16: the same two as IEEE half (HAMT_F16: same bytes, a rounding 8 x finer).
               * 0: both fp32.  At most one flag per tensor. */
} hamt_ln_desc;
#define HAMT_LN_X_BF16 1
#define HAMT_LN_Z_BF16 2
#define HAMT_LN_X_F16 4
#define HAMT_LN_Z_F16 8
int hamt_ln_fwd(const hamt_ln_desc* d, const void* x, const float* residual, const float* gamma,
                const float* beta, void* z, float* y, void* y16, float* mean, float* rstd,
                const uint64_t* rng, void* stream);
/* dz = d(pre-LN sum) (also the residual gradient); dx = dropout_pre-masked dz (may be NULL); dx16 (optional) = the same
 * as bf16 [Mpad16, H] -- the operand of the dgrad/wgrad GEMMs of the dense layer that produced x; dxsum (optional) +=
 * column sums of dx = that layer's bias gradient; dgamma/dbeta/dxsum are STORED (overwritten).  ws: HAMT_WS_LN_BWD {M, H} bytes
 * (3 H floats per block of the kernel, at most 256 blocks). */
int hamt_ln_bwd(const hamt_ln_desc* d, const float* dy, const void* z, const float* mean,
                const float* rstd, const float* gamma, float* dz, float* dx, void* dx16, float* dgamma,
                float* dbeta, float* dxsum, float* ws, const uint64_t* rng, void* stream);
/* pre-LN blocks (x + f(LN(x)), vision_transformer.py:196-197): dz = LayerNorm-backward(dy) + add, where `add` is the
 * gradient that reaches x through the residual path -- one pass instead of a LayerNorm backward and an add. */
int hamt_ln_bwd_add(const hamt_ln_desc* d, const float* dy, const void* z, const float* mean, const float* rstd,
                    const float* gamma, const float* add, float* dz, float* dgamma, float* dbeta, float* ws,
                    void* stream);
/* The second half of hamt_ln_bwd / hamt_ln_bwd_add on its own: sums the per-block partials a call with
 * dgamma == dbeta == dxsum == NULL left in `ws` (same M, H) into dgamma / dbeta / dxsum (stored; any may be NULL).  The
 * parameter gradients are not on the backward critical path, so the host runs this on an auxiliary stream. */
int hamt_ln_bwd_reduce(int M, int H, const float* ws, float* dgamma, float* dbeta, float* dxsum, void* stream);
/* The same for n calls at once (the end-of-backward form: one launch instead of one per LayerNorm; H % 64 == 0, pointers
 * 16-byte aligned).  `descs` is a HOST array; `table`: DEVICE scratch of >= n * HAMT_LNRED_TABLE_ENTRY bytes, filled by
 * small kernels from kernarg data (capturable; `descs` need not outlive the call). */
typedef struct {
  const float* ws;
  float* dgamma; /* each may be NULL */
  float* dbeta;
  float* dxsum;
  int M, H;
  int atomic; /* bit 0 / 1 / 2: dgamma / dbeta / dxsum is atomically ADDED to instead of stored (several LayerNorm calls sharing
               * one parameter's zero-initialised gradient slot: no separate accumulation pass) */
} hamt_ln_reduce_desc;
#define HAMT_LNRED_TABLE_ENTRY 48
int hamt_ln_bwd_reduce_grouped(int n, const hamt_ln_reduce_desc* descs, void* table, size_t table_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * gather/scatter family (embedding lookups A1/A10/A11, boolean-mask compaction A15/A19
 * pretrain_cmt.py:161-165, SPREL anchor gather A18 pretrain_cmt.py:211-214, cat/slice A9/A12)
 *   gather_rows:  out[r, col0:col0+W] = (base ? base[r, col0:col0+W] : 0) + src[idx[r], :W]   idx int64, r < R
 *   scatter_add:  dst[idx[r], :W] += src[r, col0:col0+W]   (atomic; duplicate indices allowed)
 * idx == NULL means identity (row copy between strided buffers); out may alias base.
 * ---------------------------------------------------------------------------------------------- */
int hamt_gather_rows(int R, int W, const float* src, int ld_src, const int64_t* idx, const float* base,
                     int ld_base, float* out, int ld_out, int col0, void* stream);
int hamt_scatter_add_rows(int R, int W, const float* src, int ld_src, int col0, const int64_t* idx,
                          float* dst, int ld_dst, void* stream);
/* scatter_add in a FIXED summation order (colliding rows are added in row order by one writer per table row: bit-reproducible, unlike the
 * atomic form); idx must be given; T = rows of the table (< 2^31): source rows whose index is outside [0, T) are skipped.  ws: 34 R ints of device
 * scratch.  Beyond 262 144 source rows (the ranking is quadratic) it falls back to the atomic kernel and says so on stderr, once. */
int hamt_scatter_add_rows_ordered(int R, int W, const float* src, int ld_src, int col0, const int64_t* idx,
                                  float* dst, int ld_dst, int T, int* ws, void* stream);
/* the same for a CONTIGUOUS table dst[T][W] of T <= 8 rows (idx must be given): fixed-order sums instead of atomics -- the table's
 * gradient is then bit-reproducible.  ws: 64 * T * W floats. */
int hamt_scatter_add_rows_small(int R, int W, const float* src, int ld_src, int col0, const int64_t* idx, int T, float* dst,
                                float* ws, void* stream);

/* BertEmbeddings sum (A1, vilmodel.py:62-66): z[b*L+l] = word[ids[b,l]] + pos[l] + type0 */
int hamt_embed_sum_fwd(int B, int L, int H, const int64_t* ids, const float* word, const float* pos,
                       const float* type_row, float* z, void* stream);
/* backward: dword[ids] += dz, dpos[l] += sum_b dz, dtype_row += sum dz (any of the three may be NULL);
 * V = rows of the word table (ids outside [0, V) are skipped).  ws / ws_bytes: device scratch and ITS SIZE; needed when dtype_row is given
 * (>= HAMT_WS_COLSUM {B * L, H}, checked).  With ws_bytes >= HAMT_WS_EMBED_BWD {B * L, H} the word rows are summed in a fixed order
 * (hamt_scatter_add_rows_ordered); with less (or beyond 262 144 rows) by atomic adds -- never written past ws_bytes */
int hamt_embed_sum_bwd(int B, int L, int H, int V, const int64_t* ids, const float* dz, float* dword,
                       float* dpos, float* dtype_row, float* ws, size_t ws_bytes, void* stream);

/* broadcast row ops over x[B, S, H]:
 *   mean over the middle axis  (A11 vilmodel.py:563-564):  y[b,:] = mean_s x[b,s,:]
 *   mul_bcast (A16 pretrain_cmt.py:176, A13 :722):  y[b,s,:] = a[b,s,:] * c[b,:]
 */
int hamt_mean_mid_fwd(int B, int S, int H, const float* x, float* y, void* stream);
int hamt_mean_mid_bwd(int B, int S, int H, const float* dy, float* dx, void* stream);
int hamt_mul_bcast_fwd(int B, int S, int H, const float* a, const float* c, int ldc_rows, float* y, void* stream);
int hamt_mul_bcast_bwd(int B, int S, int H, const float* a, const float* c, int ldc_rows, const float* dy,
                       float* da, float* dc, void* stream);
/* sums x[B,S,H] over b and/or s into out (accumulate): mode 0: out[H] += sum_{b,s}; 1: out[s,H] += sum_b */
int hamt_sum_rows(int B, int S, int H, const float* x, int mode, float* out, float* ws, void* stream);

/* elementwise: out = a + b (+ c) ; dropout forward/backward (feature dropout, model_HAMT.py:32-52) */
/* image [N][C][H][W] fp32 -> patch rows y[N*(H/P)*(W/P) .. Rpad)[C*P*P] (fp32 or bf16; rows beyond the patches zero): column
 * order = the flattened conv weight [D][C][P][P], so PatchEmbed's conv (vision_transformer.py:216-221) becomes one GEMM */
int hamt_patchify(int N, int C, int H, int W, int P, const float* x, void* y, int ldy, int dtype_y, int Rpad, void* stream);
int hamt_add3(size_t n, const float* a, const float* b, const float* c, float* out, void* stream);
int hamt_dropout(size_t n, const float* x, float* y, float p, uint32_t call_id, const uint64_t* rng, void* stream);
int hamt_cast_f32_bf16(size_t n, const float* x, void* y, void* stream);
/* Device-side batch collation (replaces the host loops of pretrain_src/data/common.py:5-29 `pad_tensors` /
 * `gen_seq_masks` and torch's pad_sequence as the six *_collate functions of data/r2r_tasks.py use them).  The host packs
 * the ragged per-sample rows back to back (no padding) into one pinned buffer and copies it once; on the device
 *   dst[b][t][:] = t < len_b ? src[prefix[b] + t][:] : pad_byte repeated      (rows of row_bytes bytes, any dtype)
 *   mask[b][t]   = t < len_b + add,   lens_out[b] = len_b + add               (bool bytes / int64; lens_out may be NULL)
 * with prefix[0..B] (int32, DEVICE) the exclusive prefix sum of the sample lengths in rows.  Bit exact by construction. */
int hamt_unpack_padded(const void* src, const int32_t* prefix, int B, int maxlen, int row_bytes, int pad_byte, void* dst, void* stream);
int hamt_seq_masks(const int32_t* prefix, int add, int B, int maxlen, uint8_t* mask, int64_t* lens_out, void* stream);
/* Gradient wire format of the data-parallel exchange (the stock DDP bf16_compress_hook's arithmetic: divide by the world
 * size, round to bf16, all-reduce(SUM) in bf16, widen): y[i] = bf16(x[i] * scale) and x[i] = fp32(y[i]).  x (fp32) and y
 * (bf16) must sit at the same element offset modulo 4 of 16-byte aligned bases (a mirrored staging arena). */
int hamt_wire_pack_bf16(size_t n, const float* x, void* y, float scale, void* stream);
int hamt_wire_unpack_bf16(size_t n, const void* y, float* x, void* stream);
/* x[i] = value where flag[i] == 0  (A16 in-place masked_fill_(nav_types == 0, -inf), pretrain_cmt.py:177;
 * its backward zeroes the gradient at the same positions) */
int hamt_fill_where_zero(size_t n, const int64_t* flag, float* x, float value, void* stream);
/* out[i] = (1 - mask[i]) * -10000: the additive attention mask of a bool (1 byte / element) keep-mask (vilmodel.py:597-599) */
int hamt_extend_mask(size_t n, const void* mask_u8, float* out, void* stream);
/* test aid: fill the LDS of every CU with `pattern` (a kernel that reads LDS it has not written then shows it) */
int hamt_debug_fill_lds(uint32_t pattern, void* stream);
/* measurement aid (bench.py `roofline`): with on != 0 every grouped weight-gradient KERNEL launched eagerly by
 * hamt_wgrad_grouped is bracketed by HIP events on its launch stream; hamt_debug_wgrad_times waits for them and returns
 * their number, writing up to `cap` durations (us), tile heights (256 = wgrad_grouped_p8_kernel) and flops (2 M N K over the
 * launch's table, K = the k-tiles actually multiplied).  Not for captures. */
int hamt_debug_wgrad_timing(int on);
int hamt_debug_wgrad_times(float* us, int* tile_rows, double* flops, int cap);
/* dx = dy * act'(h): mode 1 erf-GELU (vilmodel.py:23-29), mode 2 ReLU (h may be the ReLU output) */
int hamt_act_bwd(size_t n, const float* dy, const float* h, int mode, float* dx, void* stream);

/* ------------------------------------------------------------------------------------------------
 * losses, reduction='none' as in the reference
 *   ce:  loss[r] = logsumexp(x[r,:C]) - x[r,label[r]]     (A15/A16/A20; -inf logits allowed)
 *        backward  dx[r,c] = g[r] * (softmax(x[r])[c] - [c == label[r]])
 *        a label outside [0, C) marks an ignored row (F.cross_entropy's ignore_index): loss 0, dx = 0
 *   mse: loss = (x - t)^2 elementwise (A17/A18); backward dx = 2 g (x - t)
 *   kl:  loss[r] = sum_c t*(log t - log_softmax(x)[c]) with 0*log0 = 0 (A19, pretrain_cmt.py:239-240)
 *        backward dx[r,c] = g[r] * (softmax(x)[c] * sum_c t - t[c])
 * ---------------------------------------------------------------------------------------------- */
int hamt_ce_fwd(int R, int C, const float* x, int ldx, const int64_t* label, float* loss, float* lse, void* stream);
int hamt_ce_bwd(int R, int C, const float* x, int ldx, const int64_t* label, const float* lse,
                const float* g, float* dx, int lddx, void* stream);
int hamt_mse_fwd(size_t n, const float* x, const float* t, float* loss, void* stream);
int hamt_mse_bwd(size_t n, const float* x, const float* t, const float* g, float* dx, void* stream);
int hamt_kl_fwd(int R, int C, const float* x, int ldx, const float* t, int ldt, float* loss, float* lse, void* stream);
int hamt_kl_bwd(int R, int C, const float* x, int ldx, const float* t, int ldt, const float* lse,
                const float* g, float* dx, int lddx, void* stream);
/* A2C rollout loss of the finetune agent (finetune_src/r2r/agent_cmt.py:476-518), all [T, B] steps x episodes at once
 * (row-major [T][B] fp32 arrays): ret[t,b] = discounted return R_t = gamma R_{t+1} + reward_t seeded with last_value[b]
 * (NULL = 0: the caller passes the critic's value of the last state for episodes that have not ended, 0 for the others);
 * out[b] = {sum_t -logp (R - V) mask, sum_t 1/2 (R - V)^2 mask, sum_t -ent_w ent mask} (ent may be NULL).  bwd: gradients
 * of g[0] * sum(out) w.r.t. logp, value (critic term only: the advantage in the policy term is detached, :493) and ent. */
int hamt_a2c_fwd(int T, int B, const float* reward, const float* mask, const float* value, const float* logp, const float* ent,
                 const float* last_value, float gamma, float ent_w, float* ret, float* out, void* stream);
int hamt_a2c_bwd(int T, int B, const float* ret, const float* mask, const float* value, float ent_w, const float* g,
                 float* dlogp, float* dvalue, float* dent, void* stream);

/* ------------------------------------------------------------------------------------------------
 * optimiser side (A24): global L2 norm over a flat gradient arena, then the reference's HF AdamW
 * (optim/adamw.py:53-112): m,v update; p -= step_size * m / (sqrt(v) + eps); THEN p -= lr*wd*p.
 * hyper is a DEVICE array: [0]=lr, [1]=step_size (bias corrected lr), [2]=max_grad_norm (<=0: no clip).
 * gnorm_sq is a device scalar holding sum(g^2) (from hamt_sumsq); the clip coefficient
 * min(1, max_norm/(sqrt(gnorm_sq)+1e-6)) (torch clip_grad_norm_, main_r2r.py:271) is applied to g on
 * the fly.  p16 (optional) receives the bf16 shadow of the updated parameters; g is zeroed when
 * zero_grad != 0 (optimizer.zero_grad(), main_r2r.py:280).
 * ---------------------------------------------------------------------------------------------- */
int hamt_sumsq(size_t n, const float* g, float* out, int accumulate, float* ws, void* stream);
int hamt_adamw_flat(size_t n, float* p, float* g, float* m, float* v, void* p16, const float* hyper,
                    const float* gnorm_sq, float beta1, float beta2, float eps, float weight_decay,
                    int zero_grad, void* stream);
/* Whole-arena form: parameter i owns elements [ends[i-1], ends[i]) (offsets multiples of 8, n = ends[nparams-1]);
 * hyp[4*i..4*i+3] = {lr, step_size, weight_decay, active}; parameters with active == 0 are skipped entirely (the
 * reference's `if p.grad is None: continue`).  ends / hyp are DEVICE arrays refreshed by the host each step, so
 * the launch itself is static (hipGraph-capturable). */
int hamt_adamw_table(size_t n, float* p, float* g, float* m, float* v, void* p16, const int* ends,
                     const float* hyp, int nparams, const float* gnorm_sq, float max_norm, float beta1,
                     float beta2, float eps, int zero_grad, void* stream);
/* the same over the arena elements [first, first + n) only (p, g, m, v, p16 point at element `first`; `ends` / `hyp` still
 * describe the whole arena): a rank of a sharded optimizer updates just the segments it owns (parallel.ShardedGradSync) */
int hamt_adamw_table_range(size_t first, size_t n, float* p, float* g, float* m, float* v, void* p16,
                           const int* ends, const float* hyp, int nparams, const float* gnorm_sq,
                           float max_norm, float beta1, float beta2, float eps, int zero_grad, void* stream);
/* active == 2 in hyp: update like 1 but leave the gradient slot as it is (zero_grad then only applies to the active == 1
 * parameters): for slots whose producer overwrites them (the grouped weight-gradient launch stores, it does not accumulate).
 * hamt_sumsq_table: sum(g^2) over the ACTIVE parameters of the arena elements [first, first + n) (g points at element `first`;
 * same `ends` / `hyp` tables): the global-norm reduction that goes with it -- slots of inactive parameters are not read,
 * whatever they hold (torch clip_grad_norm_ over the parameters that have a gradient, main_r2r.py:271-273).  ws: 1024 floats.
 * accumulate: bit 0 = add to *out; bit 1 (HAMT_SUMSQ_SPARSE) = most of the range is inactive / skipped: the active elements are
 * spread evenly over the blocks (for a mostly active range the plain element ranges are faster). */
#define HAMT_SUMSQ_SPARSE 2
int hamt_sumsq_table(size_t first, size_t n, const float* g, const int* ends, const float* hyp, int nparams,
                     float* out, int accumulate, float* ws, void* stream);
/* active == 3: like 2, and hamt_sumsq_table skips the parameter as well -- its sum of squares comes from the weight-gradient
 * tiles (hamt_wgrad_desc.ss).  hamt_sumsq_partials: out (+)= sum of the n floats of `partials` (fixed order). */
int hamt_sumsq_partials(size_t n, const float* partials, float* out, int accumulate, void* stream);
/* g[0 .. n) = float(y16[0 .. n)) and out (+)= sum of g^2 over the ACTIVE parameters (table as hamt_sumsq_table; `first` = arena offset
 * of element 0): the widening of a reduce-scattered bf16 gradient chunk and its share of the global norm in one pass. */
int hamt_wire_unpack_sumsq(size_t first, size_t n, const void* y16, float* g, const int* ends, const float* hyp, int nparams,
                           float* out, int accumulate, float* ws, void* stream);
/* g *= min(1, max_norm / (sqrt(*gnorm_sq) + 1e-6))  -- standalone clip for torch-optimiser users */
int hamt_clip_scale(size_t n, float* g, const float* gnorm_sq, float max_norm, void* stream);

/* rng[1] += 1 on the stream (new dropout epoch; call once per optimisation step) */
int hamt_rng_advance(uint64_t* rng, void* stream);


#ifdef __cplusplus
}
#endif
#endif /* HAMT_H */

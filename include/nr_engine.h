/*
 * nr_engine.h -- C-ABI of the MI355X-native NRMS scoring engine (libnr_engine.so).
 *
 * The reference (yusanshi/news-recommendation) is pure Python/PyTorch and has no FFI of its own
 * (SURVEY.md section 2.2); this ABI is the drop-in boundary designed in SURVEY.md section 8 row b10: every
 * entry point takes raw DEVICE pointers + sizes + a hipStream_t (passed as void*), transfers no
 * ownership (the caller -- PyTorch in the Python host -- allocates every buffer incl. workspaces),
 * returns 0 on success and a negative code on error (nr_last_error() gives the text; never aborts).
 * Each function cites the reference code it replaces (paths relative to the reference root).
 *
 * Dimensions supported by the hand-tuned kernels: word_embedding_dim D=300, num_attention_heads=15
 * (d_k=20), query_vector_dim<=208; sequence lengths 20 (titles) and 50 (click history) are
 * instantiated, see nr_supported_seq_len().  Other values return NR_ERR_UNSUPPORTED.
 */
#ifndef NR_ENGINE_H
#define NR_ENGINE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define NR_OK 0
#define NR_ERR_UNSUPPORTED (-1)
#define NR_ERR_BADARG (-2)
#define NR_ERR_LAUNCH (-3)

/* Padded layout constants shared with the host (bf16 operand layouts). */
#define NR_D 300        /* word_embedding_dim, src/config.py:34 */
#define NR_KP 320       /* D padded to a multiple of the MFMA K step (32) */
#define NR_HEADS 15     /* num_attention_heads, src/config.py:45 */
#define NR_DK 20
#define NR_NP 320       /* rows of each packed W_Q/W_K/W_V block (= NR_KP, so the packed matrix doubles as the [960][320] dgrad operand) */
#define NR_QP 208       /* query_vector_dim (200, src/config.py:39) padded to a multiple of 16 */

int nr_version(void);
const char* nr_last_error(void);
/* 1 if the MHSA kernels are instantiated for sequence length S (20, 50). */
int nr_supported_seq_len(int S);

/* K1 -- nn.Embedding forward (src/model/NRMS/news_encoder.py:38): out[i,:] = table[ids[i],:].
 * Stand-alone gather used for parity and for the HBM-roofline measurement of the embedding gather.
 * ids: int64[n_tokens]; table: f32[num_rows, d]; out: f32[n_tokens, d]. d % 4 == 0. */
int nr_gather_rows_f32(const int64_t* ids, const float* table, float* out, int64_t n_tokens, int d,
                       int64_t num_rows, void* stream);

/* Pack the three nn.Linear(D,D) of MultiHeadSelfAttention (src/model/general/attention/multihead_self.py:36-38)
 * into the bf16 operand layout of the kernels: Wp bf16[3*NR_NP][NR_KP] (Q rows, then K, then V; zero padded),
 * bp f32[3*NR_NP].  Must be re-run whenever the fp32 parameters change (optimizer step).
 * TILE ORDER: every packed bf16 weight operand of this library (Wp, Wap, WaT, Wc, Wd, tiled W_hh / W_hh^T) is stored as 16 x 32
 * blocks (row tile, k-step) of the 64 lanes' 16-byte MFMA fragments back to back: element (r, k) of M[R][K] (R % 16 == 0,
 * K % 32 == 0) sits at ((r/16)*(K/32) + k/32)*512 + ((k%32)/8)*128 + (r%16)*8 + k%8.  A wave fetches one fragment as one
 * contiguous 1 KB request; from row-major rows the same fragment is 64 scattered 16-byte pieces, which the texture-address unit
 * serialises (measured 2-4x on the kernels bound by these loads).  "[R][K]" below names the logical matrix. */
int nr_pack_qkv(const float* Wq, const float* bq, const float* Wk, const float* bk, const float* Wv,
                const float* bv, uint16_t* Wp, float* bp, void* stream);
/* Same for AdditiveAttention.linear (src/model/general/attention/additive.py:17): Wa f32[qdim][D] ->
 * Wap bf16[NR_QP][NR_KP], bap f32[NR_QP], qvp f32[NR_QP] (zero padded). */
int nr_pack_additive(const float* Wa, const float* ba, const float* qv, int qdim, uint16_t* Wap,
                     float* bap, float* qvp, void* stream);

/* Multi-head self-attention forward, src/model/general/attention/multihead_self.py:46-75 with
 * ScaledDotProductAttention :15-23 (exp / (sum + 1e-8), no W_O), fused with its input stage:
 *   ids != NULL : news-encoder form, x[t,s,:] = table[ids[t,s],:] (src/model/NRMS/news_encoder.py:38)
 *                 followed by F.dropout (:38-40) when p_drop > 0;
 *   ids == NULL : user-encoder form, x = x_dense f32[n_seq, S, D] (src/model/NRMS/user_encoder.py:23).
 * Output ctx bf16[n_seq*S][NR_KP]; column D holds 1.0 and columns D+1.. are zero (so that GEMMs against ctx
 * also yield the bias gradient).  Training: q_save/k_save bf16[n_seq*S][NR_KP] and vt_save
 * bf16[n_seq][15][20][S rounded up to 4] receive Q, K (row-major) and V (dv-major blocks) for nr_attn_bwd;
 * pass NULL for inference.  When p_drop > 0 the second F.dropout of the
 * news encoder (:43-45) is applied to ctx.  Dropout uses a counter-based RNG keyed by (seed, site, element),
 * reproducible by nr_dropout_mask().  */
int nr_mhsa_fwd(const int64_t* ids, const float* table, int64_t num_rows, const float* x_dense,
                const uint16_t* Wp, const float* bp, uint16_t* ctx, uint16_t* q_save, uint16_t* k_save,
                uint16_t* vt_save, int64_t n_seq, int S, float p_drop, uint64_t seed, void* stream);

/* nr_mhsa_fwd_ex with per-sequence KEY LENGTHS: key_len int32[n_seq] (NULL: S).  Keys at positions >= key_len[seq] get zero attention
 * weight for every query of that sequence: the `length` argument of MultiHeadSelfAttention.forward (multihead_self.py:60-70: the mask
 * multiplies exp(scores) before the row sum).  Also how sequences shorter than an instantiated S (config knobs num_words_title,
 * num_clicked_news_a_user) are run: zero-padded to S by the host, key_len = their length, pooled with nr_additive_fwd_v(valid). */
int nr_mhsa_fwd_len(const int64_t* ids, const float* table, int64_t num_rows, const float* x_dense, const uint16_t* Wp,
                    const float* bp, uint16_t* ctx, uint16_t* q_save, uint16_t* k_save, uint16_t* vt_save, uint16_t* x_save,
                    const int32_t* key_len, int64_t n_seq, int S, float p_drop, uint64_t seed, void* stream);

/* Same, and additionally (x_save != NULL, training) emits the dropout-masked bf16 token matrix x_save[n_seq*S][NR_KP] (column D = 1.0,
 * rest of the K padding 0) that the weight-gradient GEMM dW = dqkv^T @ X needs -- what nr_gather_bf16 would recompute. */
int nr_mhsa_fwd_ex(const int64_t* ids, const float* table, int64_t num_rows, const float* x_dense,
                   const uint16_t* Wp, const float* bp, uint16_t* ctx, uint16_t* q_save, uint16_t* k_save,
                   uint16_t* vt_save, uint16_t* x_save, int64_t n_seq, int S, float p_drop, uint64_t seed, void* stream);

/* AdditiveAttention forward, src/model/general/attention/additive.py:27-53:
 * out[t,:] = sum_s softmax_s(tanh(ctx[t,s,:] Wa^T + ba) . qv) * ctx[t,s,:].
 * ctx bf16[n_seq*S][NR_KP]; out f32[n_seq][D]; attn_w f32[n_seq][S] (saved for backward, may be NULL). */
int nr_additive_fwd(const uint16_t* ctx, const uint16_t* Wap, const float* bap, const float* qvp,
                    float* out, float* attn_w, int64_t n_seq, int S, void* stream);

/* DotProductClickPredictor.forward, src/model/general/click_predictor/dot_product.py:8-19:
 * out[b,c] = cand[b,c,:] . user[b,:].  cand f32[B,C,D], user f32[B,D], out f32[B,C]. */
int nr_score_dot(const float* cand, const float* user, float* out, int64_t B, int C, int d, void* stream);

/* Evaluation-time scorer for ragged impressions (replaces the per-impression loop of
 * src/evaluate.py:245-260 around NRMS.get_prediction, src/model/NRMS/__init__.py:73-84):
 * nnz = cand_ptr[n_impr] (host value).  For impression i, for j in [cand_ptr[i], cand_ptr[i+1]): out[j] = news[cand_idx[j],:] . users[user_idx[i],:].
 * A negative cand_idx selects the all-zero PADDED_NEWS vector (src/evaluate.py:203-204). */
int nr_score_csr(const float* news, const float* users, const int32_t* cand_idx, const int64_t* cand_ptr,
                 const int32_t* user_idx, float* out, int64_t n_impr, int64_t nnz, int d, void* stream);

/* ---- backward (autograd of the forward entry points; the reference relies on torch.autograd,
 * triggered at src/train.py:231) -------------------------------------------------------------------------- */
#define NR_LDG (3 * NR_KP)   /* row length of the dQKV gradient matrix: [dQ | dK | dV], each NR_KP wide */

/* Backward of ScaledDotProductAttention (multihead_self.py:15-23) per (sequence, head).  The upstream gradient of
 * ctx is assembled on the fly as dC = (dctx_gemm + attn_w (x) g_out) * dropout2, i.e. the additive layer's
 * dpre @ Wa (bf16 [n_seq*S][ldc], a plain GEMM done by the caller) plus its direct term.  Writes
 * dqkv bf16[n_seq*S][NR_LDG] (padding columns untouched: zero-fill once). */
int nr_attn_bwd(const uint16_t* q_save, const uint16_t* k_save, const uint16_t* vt_save, const uint16_t* dctx_gemm,
                int ldc, const float* attn_w, const float* g_out, uint16_t* dqkv, int64_t n_seq, int S,
                float p_drop, uint64_t seed, void* stream);
/* nr_attn_bwd for a forward that ran with key lengths (nr_mhsa_fwd_len): the recomputed attention probabilities use the same mask. */
int nr_attn_bwd_len(const uint16_t* q_save, const uint16_t* k_save, const uint16_t* vt_save, const uint16_t* dctx_gemm, int ldc,
                    const float* attn_w, const float* g_out, uint16_t* dqkv, const int32_t* key_len, int64_t n_seq, int S, float p_drop,
                    uint64_t seed, void* stream);

/* ---- training form of the news-encoder front end as a gather-fused projection GEMM + a stand-alone attention kernel -----------------
 * (csrc/k_proj.h).  nr_mhsa_fwd's register-resident kernel remains the inference form; in training everything it keeps in registers has
 * to be written for the backward anyway.
 *
 * Saved-activation layout "head-major": qkv bf16[n_seq][15][3][20][20] -- per (sequence, head) the blocks Q, K, V, each [token][d]
 * row-major, back to back (2,400 contiguous bytes), NR_QKV_HM_SEQ elements per sequence. */
#define NR_QKV_HM_SEQ (NR_HEADS * 3 * 20 * NR_DK)
#define NR_K16 19       /* k-steps of 16 over D (304 columns) in the tile32-ordered projection operand */
/* Pack the three nn.Linear(D,D) of MultiHeadSelfAttention (multihead_self.py:36-38) for nr_qkv_proj_fwd: Wp32 bf16[3*NR_NP][304] in
 * "tile32 order" -- 32 x 16 blocks (row tile, k-step) of the 64 lanes' 16-byte v_mfma_f32_32x32x16_bf16 fragments back to back: element
 * (r, k) at ((r/32)*19 + k/16)*512 + (((k%16)/8)*32 + r%32)*8 + k%8 -- with the rows of each projection in PACKED order: row c holds
 * output feature 60*(c/64) + c%64 when c%64 < 60 (groups of 3 heads), zeros otherwise; bp f32[3*NR_NP] in the same row order. */
int nr_pack_qkv32(const float* Wq, const float* bq, const float* Wk, const float* bk, const float* Wv, const float* bv,
                  uint16_t* Wp32, float* bp, void* stream);
/* Every operand packing of ONE encoder in one launch (pack_encoder_kernel, csrc/k_proj.h): any subset of nr_pack_qkv (Wp, bp),
 * nr_pack_qkv32 (Wp32, bp32), nr_pack_qkv_dx (WdX), nr_pack_additive (Wap, bap, qvp), nr_pack_additive_t (WaT) -- null outputs are skipped;
 * same layouts, same bits as the single entry points.  The parameters are the nn.Linear weights / biases of MultiHeadSelfAttention
 * (multihead_self.py:38-40) and AdditiveAttention (additive.py:18-25) of one encoder. */
int nr_pack_encoder(const float* Wq, const float* bq, const float* Wk, const float* bk, const float* Wv, const float* bv, const float* Wa,
                    const float* ba, const float* qv, int qdim, uint16_t* Wp, float* bp, uint16_t* Wp32, float* bp32, uint16_t* WdX,
                    uint16_t* Wap, float* bap, float* qvp, uint16_t* WaT, void* stream);

/* x = F.dropout(table[ids]) (src/model/NRMS/news_encoder.py:38-40), then Q, K, V = x W^T + b (multihead_self.py:53-55), S = 20.
 * ids int64[n_seq*S]; qkv: head-major (above); x_save (optional) bf16[n_seq*S][NR_KP]: the dropout-masked token matrix, column D = 1.0,
 * other padding 0 -- the operand of the weight-gradient GEMM dW = dqkv^T @ [X | 1]. */
int nr_qkv_proj_fwd(const int64_t* ids, const float* table, int64_t num_rows, const uint16_t* Wp32, const float* bp, uint16_t* qkv,
                    uint16_t* x_save, int64_t n_seq, int S, float p_drop, uint64_t seed, void* stream);
/* Input gradient of the three projections as a hand-written GEMM (autograd of multihead_self.py:53-55 w.r.t. its input, triggered at
 * src/train.py:231): dX bf16[n_tok][NR_KP] = dqkv bf16[n_tok][NR_LDG] @ [Wq; Wk; Wv] (rows = the NR_LDG columns of dqkv, padding rows zero;
 * columns >= D of dX come out as exact zeros).  WdX: the weights packed by nr_pack_qkv_dx / nr_pack_encoder OF THE SAME PROCESS, 307,200 bf16,
 * opaque to the caller.  Default form: bf16[60][10][64][8], block (k-step ks, column tile nt) = the 64 lanes' v_mfma_f32_32x32x16_bf16
 * fragments, lane l: Wall[16 ks + 8 (l >> 5) + j][32 nt + (l & 31)], j = 0..7, read by dx_gemm_ring_kernel (csrc/k_proj.h: one workgroup per
 * 256-token tile).  NR_DX_STREAM=1 (read once per process; A/B, a tie on MI355X): row-major [NR_KP n][NR_LDG k], WdX[n][which NR_KP + f] =
 * W_which[f][n], read by the persistent stream kernel (conv_gemm_kernel<., PLAIN>, csrc/k_convgemm.h: one workgroup per CU walks the tiles,
 * the copy ring never drains). */
int nr_pack_qkv_dx(const float* Wq, const float* Wk, const float* Wv, uint16_t* WdX, void* stream);
int nr_dx_gemm(const uint16_t* dqkv, const uint16_t* WdX, uint16_t* dX, int64_t n_tok, void* stream);
/* Weight (and bias) gradients of a linear layer as a hand-written split-K "TN" GEMM (autograd of multihead_self.py:53-55 / additive.py:35
 * w.r.t. weight and bias; src/train.py:231): out f32[P][M][NR_KP], out[p][m][n] = sum over the tokens of partition p of G[tok][m] * X[tok][n];
 * the sum over p is the gradient (nr_wgrad_unpack reduces the partitions in a fixed order).  G bf16[n_tok][ldg] (dqkv with M = NR_LDG, dpre
 * with M = NR_QP), X bf16[n_tok][NR_KP] (column D = 1.0: row D of the result is the bias gradient), zeros: 16 zero bytes in device memory.
 * P: a multiple of 8; nr_tn_gemm_parts(M, n_tok) gives the library's choice (enough workgroups for two per CU). */
int nr_tn_gemm_parts(int M, int64_t n_tok);
int nr_tn_gemm(const uint16_t* G, int ldg, int M, const uint16_t* X, const uint16_t* zeros, float* out, int64_t n_tok, int P, void* stream);
/* General bf16 GEMMs with fp32 accumulation and fp32 results (csrc/k_gemm.h) -- the products the reference leaves to nn.GRU / torch.autograd:
 *   nr_gemm_nt   C f32[M][ldc] = A bf16[M][lda] . B bf16[N][ldb]^T, both operands K-contiguous, K a multiple of 32 (padding columns of both
 *                operands must be finite; zero where the product must not see them), row strides multiples of 8 elements.  Replaces:
 *                the hoisted input projection x W_ih^T of nn.GRU (src/model/LSTUR/user_encoder.py:11-14,43-45), autograd's
 *                dX = dGi W_ih (B = W_ih^T, re-packed by nr_transpose_bf16), and the recurrent product h W_hh^T of the evaluation sweep.
 *   nr_gemm_tn   out f32[P][M][ldo], out[p][m][n] = sum over the tokens of partition p of G[tok][m] * X[tok + n / tapw][n % tapw]: weight
 *                gradients with split K (sum over p = the gradient; nr_sum_parts reduces in a fixed order).  taps = 1: X bf16[n_tok][ldx],
 *                N = tapw <= ldx columns (dW_ih = dGi^T X, dW_hh = dGh^T H of nn.GRU; the nn.Linear weights of multihead_self.py:53-55 /
 *                additive.py:35).  taps = 3: X is a seqpad buffer (nr_conv3_fwd's x_save) with n_tok + 2 readable rows, and the N = 3 tapw
 *                (taps <= 9: any odd window_size on the general-geometry path; the tuned path uses 3) result columns are the three tap gradients of Conv2d(1, F, (3, D)) side by side (src/model/NAML/news_encoder.py:27-28,
 *                src/model/LSTUR/news_encoder.py:26-30): one pass over G for all taps.  P: multiple of 8 (nr_gemm_tn_parts = one workgroup
 *                per CU); zeros: 16 zero bytes in device memory.
 *   nr_transpose_bf16  dst[c][r] = src[r][c] (weight re-packing, once per optimiser step); nr_sum_parts  out[i] (+)= sum_p parts[p][i]. */
int nr_gemm_nt(const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, float* C, int64_t ldc, int64_t M, int N, int K, void* stream);
int nr_gemm_tn_parts(int M, int N, int64_t n_tok);
int nr_gemm_tn(const uint16_t* G, int64_t ldg, int M, const uint16_t* X, int64_t ldx, int tapw, int taps, const uint16_t* zeros, float* out,
               int64_t ldo, int64_t n_tok, int P, void* stream);
int nr_transpose_bf16(const uint16_t* src, int R, int C, int64_t lds, uint16_t* dst, int64_t ldd, void* stream);
int nr_sum_parts(const float* parts, int P, int64_t n, float* out, int accumulate, void* stream);
/* ScaledDotProductAttention (multihead_self.py:15-23: exp / (sum + 1e-8), optional key lengths :60-70) from a head-major qkv buffer;
 * ctx as nr_mhsa_fwd writes it (second dropout of news_encoder.py:43-45 applied when p_drop > 0, column D = 1.0). */
int nr_attn_fwd(const uint16_t* qkv, uint16_t* ctx, const int32_t* key_len, int64_t n_seq, int S, float p_drop, uint64_t seed, void* stream);
/* nr_attn_fwd followed by the additive pooling of each title (additive.py:27-53; news_encoder.py:43-47) on the ctx rows while they are
 * still in LDS: out f32[n_seq][out_stride] (first NR_D columns), attn_w f32[n_seq][S] or null; ctx is written as by nr_attn_fwd (the
 * backward reads it).  ctx equals nr_attn_fwd bit for bit, out / attn_w equal nr_additive_fwd_v on it up to the order of fp32 additions (valid: as there).  Wap / bap / qvp: nr_pack_additive. */
int nr_attn_pool_fwd(const uint16_t* qkv, uint16_t* ctx, const int32_t* key_len, const uint16_t* Wap, const float* bap, const float* qvp,
                     float* out, int64_t out_stride, float* attn_w, int64_t n_seq, int S, int valid, float p_drop, uint64_t seed, void* stream);
/* nr_attn_bwd_len reading the head-major saves of nr_qkv_proj_fwd (S = 20). */
int nr_attn_bwd_hm(const uint16_t* qkv, const uint16_t* dctx_gemm, int ldc, const float* attn_w, const float* g_out, uint16_t* dqkv,
                   const int32_t* key_len, int64_t n_seq, int S, float p_drop, uint64_t seed, void* stream);

/* Backward of AdditiveAttention (additive.py:35-52) up to the pre-activation: dpre bf16[n_seq*S][NR_QP] and
 * per-workgroup partial sums of the query-vector gradient dq_part f32[nr_additive_bwd_grid()][NR_QP]
 * (sum over rows = d attention_query_vector).  The caller finishes with two plain GEMMs:
 * d linear.weight|bias = dpre^T @ ctx (bias = column D, where ctx holds 1.0) and dctx_gemm = dpre @ Wa. */
int64_t nr_additive_bwd_grid(int64_t n_seq, int S);
int nr_additive_bwd(const uint16_t* ctx, const uint16_t* Wap, const float* bap, const float* qvp,
                    const float* attn_w, const float* g_out, uint16_t* dpre, float* dq_part, int64_t n_seq, int S,
                    void* stream);

/* nr_additive_bwd that also emits the GEMM part of the input gradient, dctx bf16[n_seq*S][NR_KP] (columns < D) = dpre @ Wa, from
 * WaT bf16[NR_KP][NR_QKP] = Wa^T (nr_pack_additive_t); saves the caller one [tokens x 208] x [208 x 300] library GEMM and a pass over dpre. */
#define NR_QKP 224     /* NR_QP rounded up to the MFMA k-step */
int nr_pack_additive_t(const float* Wa, int qdim, uint16_t* WaT, void* stream);
int nr_additive_bwd_ex(const uint16_t* ctx, const uint16_t* Wap, const float* bap, const float* qvp, const float* attn_w,
                       const float* g_out, uint16_t* dpre, float* dq_part, const uint16_t* WaT, uint16_t* dctx, int64_t n_seq, int S,
                       void* stream);

/* Weight gradients of one encoder, from the form the two library GEMMs and nr_additive_bwd leave them in to the parameters' gradient
 * buffers, in ONE launch: dW_parts f32[nc_w][3*NR_KP][NR_KP] = per-token-chunk partials of dqkv^T @ [X | 1] (row i*NR_KP + r = d W_i[r][:],
 * column NR_D = d bias_i[r]; i = Q, K, V), dWa_parts f32[nc_a][NR_QP][NR_KP] likewise for the pooling linear, dq_part f32[nwg][NR_QP]
 * (nr_additive_bwd).  Sums the partials in a fixed order and ACCUMULATES (+=) into gW*[NR_D][NR_D], gb*[NR_D], gWa[qdim][NR_D], gba[qdim],
 * gq[qdim] -- what autograd's AccumulateGrad does for W_Q / W_K / W_V (multihead_self.py:35-37), additive.linear and
 * attention_query_vector (additive.py:18-20) when loss.backward() runs (train.py:231), there as 2 + 9 + 1 separate kernels. */
int nr_wgrad_unpack(const float* dW_parts, int nc_w, const float* dWa_parts, int nc_a, const float* dq_part, int64_t nwg, int qdim,
                    float* gWq, float* gbq, float* gWk, float* gbk, float* gWv, float* gbv, float* gWa, float* gba, float* gq, void* stream);

/* Token matrix for the weight-gradient GEMM dW = dqkv^T @ Xb: Xb bf16[n_tokens][NR_KP] = dropout1(table[ids]) (or
 * dense f32 rows), column D = 1.0 (bias gradient), rest 0. */
int nr_gather_bf16(const int64_t* ids, const float* table, int64_t num_rows, const float* x_dense, uint16_t* Xb,
                   int64_t n_tokens, float p_drop, uint64_t seed, void* stream);

/* Backward of nn.Embedding(padding_idx=0) (news_encoder.py:15-20): grad_table[ids[t]] += dropout1(dx[t]) for
 * ids[t] != 0.  dx bf16[n_tokens][ldx]; grad_table f32[num_rows][D] (accumulated with fp32 atomics). */
int nr_embed_scatter_add(const int64_t* ids, const uint16_t* dx, int ldx, float* grad_table, int64_t num_rows,
                         int64_t n_tokens, float p_drop, uint64_t seed, void* stream);

/* Same result without contended atomics: the caller passes the token ids sorted ascending together with the
 * permutation (ids_sorted[i] = ids[perm[i]]); each table row is reduced in registers and ADDED to grad_table once (read, add,
 * write; atomics only where a run spans two waves), so grad_table may be a zeroed scratch or the parameter's live gradient. */
int nr_embed_scatter_sorted(const int64_t* ids_sorted, const int64_t* perm, const uint16_t* dx, int ldx,
                            float* grad_table, int64_t num_rows, int64_t n_tokens, float p_drop, uint64_t seed,
                            void* stream);

/* Backward of DotProductClickPredictor: d_cand[b,c,:] = dl[b,c]*user[b,:], d_user[b,:] = sum_c dl[b,c]*cand[b,c,:]. */
int nr_score_dot_bwd(const float* dl, const float* cand, const float* user, float* d_cand, float* d_user,
                     int64_t B, int C, int d, void* stream);

/* ---- the training loop's scorer + loss in one pass (round 6) ------------------------------------------------------
 * DotProductClickPredictor.forward (src/model/general/click_predictor/dot_product.py:8-19) followed by the training loop's
 * `loss = criterion(y_pred, y)` with criterion = nn.CrossEntropyLoss() (mean reduction; src/train.py:130,205-206, LSTUR :186-187):
 *   logits[b,c] = cand[b,c,:] . user[b,:]           (bit-identical to nr_score_dot; logits may be NULL)
 *   loss_rows[b] = logsumexp_c(logits[b,:]) - logits[b,target[b]]      (target NULL = class 0, what train.py builds with torch.zeros)
 *   loss[0]     = (1 / B) sum_b loss_rows[b]        (one workgroup, fixed order)
 *   dl[b,c]     = (softmax_c(logits[b,:])[c] - [c == target[b]]) / B   = d loss / d logits, saved for nr_score_ce_bwd.
 * cand f32[B,C,d], user f32[B,d] (16-byte aligned), 1 <= C <= 64, d % 4 == 0.  The caller validates target[b] in [0, C). */
int nr_score_ce_fwd(const float* cand, const float* user, const int64_t* target, float* logits, float* dl, float* loss_rows,
                    float* loss, int64_t B, int C, int d, void* stream);

/* Backward of the above for a gradient g = gscale[0] arriving at the scalar loss (gscale NULL = 1): d_cand[b,c,:] = g dl[b,c] user[b,:],
 * d_user[b,:] = g sum_c dl[b,c] cand[b,c,:].  Row (b,c) of d_cand starts at d_cand + (b*C + c)*ldc, row b of d_user at d_user + b*ldu
 * (floats; multiples of 4, >= d): the candidates' gradient can be written straight into the leading rows of the news encoder's output
 * gradient, replacing autograd's concatenation. */
int nr_score_ce_bwd(const float* dl, const float* gscale, const float* cand, const float* user, float* d_cand, int64_t ldc,
                    float* d_user, int64_t ldu, int64_t B, int C, int d, void* stream);

/* bf16 rows -> f32 rows: dst[r*ldd + c] = src[r*ld + c], r < n, c < d (d, ld, ldd multiples of 4).  The dense input gradient of an encoder
 * stage (bf16 [n][NR_KP] out of the NT GEMM) written in the f32 layout of the tensor it is the gradient of -- e.g. straight into the history rows
 * of the news encoder's output gradient (src/model/NRMS/__init__.py:43-48: clicked_news_vector is a slice of the encoder output). */
int nr_rows_to_f32(const uint16_t* src, int64_t ld, int d, float* dst, int64_t ldd, int64_t n, void* stream);

/* Batched accumulate: for every item, dst[r*dst_ld + c] += sum_{p < parts} src[p*part_stride + r*src_ld + c], r < rows, c < cols -- in ONE launch
 * per 48 items.  Replaces torch.autograd's AccumulateGrad (one element-wise add per parameter after loss.backward(), src/train.py:207,228) for
 * a trainer whose gradient buffers are persistent; parts > 1 folds the fixed-order sum over a persistent kernel's per-workgroup partial rows
 * (the query-vector gradient of additive.py:20) into the same launch.  `items` is a HOST array read during the call.  Items of one call must not
 * overlap in dst.  parts >= 1; reserved = 0. */
typedef struct nr_accum_item {
  const float* src;
  float* dst;
  int64_t src_ld, dst_ld;
  int32_t rows, cols;
  int32_t parts, reserved;
  int64_t part_stride;
} nr_accum_item;
int nr_accum_many(const nr_accum_item* items, int n_items, void* stream);

/* ---- NAML / LSTUR: convolutional text encoder, pooling variants, element encoders ------------------------------
 * "seqpad" layout used below: bf16 [n_seq*(S+1)+1][NR_KP]; token s of sequence q sits in row q*(S+1)+1+s and the rows
 * q*(S+1) are all-zero separators (never written by the kernels: zero-fill once), so a tap shift is a row offset. */

/* 1 if the additive-attention kernels are instantiated for S (4 = NAML view stack, 20, 50); conv kernels: 20, 50. */
int nr_supported_pool_len(int S);
int nr_supported_conv_len(int S);

/* Pack nn.Conv2d(1, F, (3, D), padding=(1,0)) parameters (src/model/NAML/news_encoder.py:15-17, src/model/LSTUR/news_encoder.py:24-28;
 * weight f32 [F][1][3][D], bias f32 [F]) into the forward operand Wc bf16 [3][NR_KP][NR_KP] (tap, filter, d), the
 * data-gradient operand Wd bf16 [3][NR_KP][NR_KP] (Wd[t][d][f] = W[f][2-t][d]; may be NULL) and bc f32 [NR_KP]. */
int nr_pack_conv(const float* W, const float* b, int F, int D, uint16_t* Wc, uint16_t* Wd, float* bc, void* stream);

/* TextEncoder front (NAML news_encoder.py:21-32; LSTUR news_encoder.py:58-67):
 * act = dropout(relu(conv3(dropout(table[ids])) + bias)) as bf16 [n_seq*S][NR_KP] (col D = 1.0, cols > D zero) -- the
 * input layout of nr_additive_fwd.  Training: x_save (seqpad, col D = 1.0 on token rows) receives the masked bf16 tokens
 * for the weight-gradient GEMMs; NULL for inference.  tok_offset shifts the dropout counters (site 1 = tokens, site 2 =
 * activations) so several texts can share one seed. */
int nr_conv3_fwd(const int64_t* ids, const float* table, int64_t num_rows, const uint16_t* Wc, const float* bc, uint16_t* act,
                 uint16_t* x_save, int64_t n_seq, int S, float p_drop, uint64_t seed, int64_t tok_offset, void* stream);
/* nr_conv3_fwd for texts of `valid` (1..S) tokens zero-padded to an instantiated S: positions >= valid enter the convolution as zero
 * vectors (what Conv2d's own padding puts after the last token), and are excluded downstream by nr_additive_fwd_v(valid). */
int nr_conv3_fwd_v(const int64_t* ids, const float* table, int64_t num_rows, const uint16_t* Wc, const float* bc, uint16_t* act,
                   uint16_t* x_save, int64_t n_seq, int S, int valid, float p_drop, uint64_t seed, int64_t tok_offset, void* stream);
/* Data gradient of the convolution: dx bf16 [n_seq*S][NR_KP] (cols < D) from dy_pad (seqpad) and Wd. */
int nr_conv3_dgrad(const uint16_t* dy_pad, const uint16_t* Wd, uint16_t* dx, int64_t n_seq, int S, void* stream);
/* The same data gradient as ONE GEMM (csrc/k_gemm.h, NT3 form: the virtual operand row [dy[i], dy[i + 1], dy[i + 2]] makes the three taps a
 * contraction of length 3 * NR_KP): Wd2 bf16 [NR_KP][3 * NR_KP] row-major from nr_pack_conv_dgrad (Wd2[d][t * NR_KP + f] = W[f][2 - t][d]).
 * Same result as nr_conv3_dgrad up to the order of the fp32 additions; columns >= D of dx come out as exact zeros. */
int nr_pack_conv_dgrad(const float* W, int F, int D, uint16_t* Wd2, void* stream);
/* The TRAINING forward of the text encoders (same reference lines and outputs as nr_conv3_fwd_v with x_save: NAML news_encoder.py:23-32, LSTUR
 * news_encoder.py:58-67) as a gather pass + the persistent ring GEMM of csrc/k_convgemm.h (EPI): x_save (required: bf16 seqpad
 * [n_seq*(S+1)+1][NR_KP], the operand of the weight-gradient GEMMs) is written first and is the GEMM's token-row operand; act = dropout2(relu(y + b)),
 * column D = 1.0.  Wf2 bf16 [NR_KP][3 * NR_KP] row-major from nr_pack_conv_fwd2 (Wf2[f][t * NR_KP + d] = W[f][t][d]); bc: nr_pack_conv's f32[NR_KP].
 * Same dropout counters as nr_conv3_fwd_v: same masks, x_save bit-identical, act equal up to the summation order of the taps. */
int nr_pack_conv_fwd2(const float* W, int F, int D, uint16_t* Wf2, void* stream);
int nr_conv3_fwd_gemm(const int64_t* ids, const float* table, int64_t num_rows, const uint16_t* Wf2, const float* bc, uint16_t* act,
                      uint16_t* x_save, int64_t n_seq, int S, int valid, float p_drop, uint64_t seed, int64_t tok_offset, void* stream);
int nr_conv3_dgrad_gemm(const uint16_t* dy_pad, const uint16_t* Wd2, uint16_t* dx, int64_t n_seq, int S, void* stream);
/* Gradient through dropout+relu: dy_pad[row(q,s)] = (dact_gemm[t] + attn_w[t] * g_out[q]) * [act[t] != 0] / (1 - p_drop);
 * dact_gemm bf16 [n_seq*S][ldc] = dpre @ Wa (plain GEMM by the caller), g_out f32 rows of stride g_stride. */
int nr_conv_act_bwd(const uint16_t* act, const uint16_t* dact_gemm, int ldc, const float* attn_w, const float* g_out, int64_t g_stride,
                    uint16_t* dy_pad, int64_t n_seq, int S, float p_drop, void* stream);

/* Backward of the additive pooling over a conv text encoder's activations, through the activation stage, in one call: dpre / dq_part as
 * nr_additive_bwd_ex, and dy_pad as nr_conv_act_bwd would produce it from dctx = dpre @ Wa (act is both the pooling's input ctx and the
 * activation whose zeros are the relu / dropout mask; g_out f32 [n_seq][NR_D]).  With the register-resident pooling kernels (S = 20;
 * S = 50 from 2048 sequences up) the product never reaches memory: the kernel's epilogue adds the direct term, masks, scales and stores
 * the seqpad rows (one pass over the tokens less; the sum is rounded to bf16 once instead of twice).  Other shapes run the two kernels
 * through dctx_scratch bf16 [n_seq*S][NR_KP].  Autograd of additive.py:35-52 composed with F.dropout(F.relu(conv)) (NAML / LSTUR
 * news_encoder.py). */
int nr_additive_bwd_act(const uint16_t* act, const uint16_t* Wap, const float* bap, const float* qvp, const float* attn_w, const float* g_out,
                        uint16_t* dpre, float* dq_part, const uint16_t* WaT, uint16_t* dctx_scratch, uint16_t* dy_pad, float p_drop,
                        int64_t n_seq, int S, void* stream);

/* The pooling backward over a FLAT token stream (csrc/k_pool3.h; autograd of additive.py:27-53): every output of nr_additive_bwd_ex /
 * nr_additive_bwd_act for any sequence length S >= 4 (S >= 16 with dy_pad: 48 consecutive tokens must belong to at most 16 / 4 sequences;
 * shorter ones return NR_ERR_UNSUPPORTED and stay with the sequence-shaped entries) from one persistent kernel.  The sum inside the softmax backward,
 * sum_s w[s] (g_out . x[s]), equals g_out[seq] . y[seq] with y the pooled vector of the FORWARD (nr_additive_fwd*'s `out`: f32 rows of stride
 * y_stride), so token rows are independent and are dealt to waves 48 at a time regardless of sequence boundaries.  tot: f32 [n_seq]
 * scratch (receives g_out . y).  dq_part: f32 [nr_additive_bwd_flat_grid(n_seq * S)][NR_QP] partial rows.  Exactly one of, or neither of,
 * dctx (bf16 [n_seq*S][NR_KP] = dpre @ Wa, columns < D written) and dy_pad (the fused activation gradient of nr_additive_bwd_act, scaled by
 * 1 / (1 - p_drop)) may be given; with neither the call stops at dpre / dq.  Needs no Wa^T operand: the kernel reads Wa through LDS
 * transposing reads.  qdim = query_vector_dim of the packed operands: the kernel keeps 200 rows of Wa in LDS -- 201 .. NR_QP return
 * NR_ERR_UNSUPPORTED (the sequence-shaped entries above handle all NR_QP packed rows). */
int64_t nr_additive_bwd_flat_grid(int64_t n_tok);
int nr_additive_bwd_flat(const uint16_t* ctx, const uint16_t* Wap, const float* bap, const float* qvp, const float* attn_w, const float* g_out,
                         const float* y, int64_t y_stride, float* tot, uint16_t* dpre, float* dq_part, uint16_t* dctx, uint16_t* dy_pad,
                         float p_drop, int64_t n_seq, int S, int qdim, void* stream);
/* Same with the sequence gradients as a column block of wider rows: row r of g_out starts at g_out + r*g_stride (floats; a multiple of 4, >= D,
 * g_out 16-byte aligned).  LSTUR's news vector is [category row | subcategory row | title vector] (src/model/LSTUR/news_encoder.py:73-76): the
 * title encoder's backward reads its third of the [n, 3F] gradient in place instead of from a contiguous copy. */
int nr_additive_bwd_flat_gs(const uint16_t* ctx, const uint16_t* Wap, const float* bap, const float* qvp, const float* attn_w, const float* g_out,
                            int64_t g_stride, const float* y, int64_t y_stride, float* tot, uint16_t* dpre, float* dq_part, uint16_t* dctx,
                            uint16_t* dy_pad, float p_drop, int64_t n_seq, int S, int qdim, void* stream);
/* Profiling aid (tools/pool3_phases.py): with NR_POOL_DEBUG set, the debug instantiation of the flat kernel writes cycle-counter stamps of the
 * first 8 iterations of waves 0-1 of workgroups 0-3 to buf (device memory, 4 * 2 * 8 * 8 uint64, owned by the caller until it passes NULL
 * again, which switches the stamps off). */
int nr_debug_pool3_stamps(uint64_t* buf);
/* Probe of the XCD-local phase barrier (csrc/k_xcd.h; tools/xcd_probe.py): 256 workgroups, `phases` write / barrier / read-back rounds.
 * sync_words: 32 uint32 (zeroed by the call); rec: 8 * 32 * 512 uint32; out: 768 uint32 (stale words per workgroup, XCC ids, slots). */
int nr_debug_xcd_probe(uint32_t* sync_words, uint32_t* rec, uint32_t* out, int phases, void* stream);
/* Fault words of the persistent GRU sweeps (csrc/k_gru_persist.h, csrc/k_xcd.h; nr_gru_fwd_seq / nr_gru_bwd_seq take that form on a 256-CU
 * device for Hd = 900 / 450 and B <= 512 unless NR_GRU_PERSIST=0).  A sweep whose XCD-local wait gives up (or whose team comes out wrong)
 * produces garbage; the reference's nn.GRU (src/model/LSTUR/user_encoder.py:30-37) has no such failure mode, so the engine guarantees that
 * such a sweep is never TRAINED on: four uint32 words in device memory --
 *   [0] / [1]  STICKY error bits of the forward / backward sweeps since the last clear (bit 0: a workgroup found its XCD's team full, bit 1: a
 *              bounded wait gave up); no launch clears them (the per-launch error word of round 5 was erased by the next sweep);
 *   [2]        index of the first optimiser step that was SKIPPED because of them (0: none);
 *   [3]        spare --
 * and while [0] | [1] != 0 every optimiser kernel (nr_adam_flat, nr_row_adam_step, nr_row_adam_catchup) applies NO update (nr_adam_flat still
 * clears the gradient when asked to): parameters and moments stay as of the last good step until the host has looked.  The host then repeats
 * the steps from [2] on with NR_GRU_PERSIST=0 (news_recommendation_amd/train_fast.py) after nr_fault_clear().
 *   nr_set_fault_words   attach four caller-owned, zero-initialised device words (NULL: back to the library's own block) -- a data-parallel
 *                        trainer all-reduces (max) them before the optimiser kernels so that every rank skips the same steps;
 *   nr_fault_state       copies the four words to out4 (HOST memory); SYNCHRONISES the device (never inside a stream capture);
 *   nr_fault_clear       zeroes them; SYNCHRONISES;
 *   nr_gru_persist_status  words [0] and [1] (kept from round 5: now sticky);
 *   nr_debug_gru_fault   test knob: the nth next persistent sweep of kind `which` (0 forward, 1 backward) is launched with workgroup 0 never
 *                        arriving at a barrier and a short spin limit, so that the give-up path runs for real (0: off). */
int nr_set_fault_words(uint32_t* words);
int nr_fault_state(uint32_t* out4);
int nr_fault_clear(void);
int nr_debug_gru_fault(int which, int nth);
/* Which sweeps of a [B] x T recurrence with hidden size Hd take the persistent form on the current device under the current NR_GRU_PERSIST: bit 0
 * forward, bit 1 backward; 0 = the step-per-launch kernels (any other device, shape or NR_GRU_PERSIST=0). */
int nr_gru_persist_enabled(int B, int Hd, int T);
int nr_gru_persist_status(int32_t* fwd, int32_t* bwd);
/* debug: constant-clock (100 MHz) stamps of the persistent forward sweep, [256 workgroups][T][8 waves][8] int64 (null = off) */
int nr_debug_gru_stamps(int64_t* buf);
/* likewise for the backward sweep: [256 workgroups][T + 1 calls][8 waves][8] */
int nr_debug_gru_stamps_bwd(int64_t* buf);
/* The same for nr_attn_bwd_hm (tools/attnb_timeline.py; NR_ATTNB_DEBUG=8 selects the debug instantiation with nothing switched off):
 * [2 workgroups][4 waves][4 sequences][4 rounds][12] stamps. */
int nr_debug_attnb_stamps(uint64_t* buf);

/* nr_additive_fwd with strided outputs: out f32 rows of stride out_stride (may be NULL) and/or out_b, a bf16 copy in the
 * ctx layout (row i at out_b + i*out_b_stride: cols 0..D-1, col D = 1.0, rest 0) that can feed another pooling level
 * directly (NAML final_attention over the 4 views, news_encoder.py:108-114; NAML user encoder, user_encoder.py:18). */
int nr_additive_fwd_ex(const uint16_t* ctx, const uint16_t* Wap, const float* bap, const float* qvp, float* out, int64_t out_stride,
                       uint16_t* out_b, int64_t out_b_stride, float* attn_w, int64_t n_seq, int S, void* stream);
/* nr_additive_fwd_ex pooling only the first `valid` (1..S) tokens of every sequence: tokens >= valid get softmax weight exactly 0
 * (attn_w rows are written for all S positions), so they drop out of the pooled vector and -- through the zero weights -- out of
 * every gradient of nr_additive_bwd_ex.  Sequences shorter than an instantiated S are zero-padded by the host. */
int nr_additive_fwd_v(const uint16_t* ctx, const uint16_t* Wap, const float* bap, const float* qvp, float* out, int64_t out_stride,
                      uint16_t* out_b, int64_t out_b_stride, float* attn_w, int64_t n_seq, int S, int valid, void* stream);
/* nr_additive_fwd_v over WHOLE SEQUENCES PER WAVE, persistent (csrc/k_pool4.h; additive.py:27-53): the projection matrix stays in LDS for the
 * kernel's life, a wave pools floor(64 / S) sequences at a time with no workgroup barrier, the weighted sum runs on the matrix core.  Any S in
 * [16, 64]; qdim = query_vector_dim of the packed operands must be <= 200 (else NR_ERR_UNSUPPORTED: the LDS-tile entries handle all NR_QP rows).
 * Same outputs as nr_additive_fwd_v up to the order of fp32 additions; the forward of every large pooling level of the conv text encoders. */
int nr_additive_fwd_flat(const uint16_t* ctx, const uint16_t* Wap, const float* bap, const float* qvp, float* out, int64_t out_stride, uint16_t* out_b,
                         int64_t out_b_stride, float* attn_w, int64_t n_seq, int S, int valid, int qdim, void* stream);
/* Input gradient of a pooling level: dx[t][:] = dgemm[t][:] + attn_w[t] * g_out[seq(t)][:], f32 [n_seq*S][D]; view_major != 0
 * stores row t at (t % S) * n_seq + t / S (S contiguous [n_seq][D] blocks, one per view). */
int nr_additive_dx(const uint16_t* dgemm, int ldc, const float* attn_w, const float* g_out, float* dx, int64_t n_seq, int S,
                   int view_major, void* stream);

/* NAML ElementEncoder (news_encoder.py:40-47) for every category row at once: E f32 [2][ncat][D],
 * E[w][c] = relu(W_w emb[c] + b_w), w = 0 category, 1 subcategory (shared embedding f32 [ncat][dcat], W_w f32 [D][dcat]). */
int nr_element_table_fwd(const float* emb, int ncat, int dcat, const float* W0, const float* b0, const float* W1, const float* b1,
                         float* E, void* stream);
/* Its backward from dE f32 [2][ncat][D]: dW f32 [2][D][dcat], db f32 [2][D], demb f32 [ncat][dcat] (row 0 zero: padding_idx). */
int nr_element_table_bwd(const float* emb, int ncat, int dcat, const float* W0, const float* W1, const float* E, const float* dE,
                         float* dW, float* db, float* demb, void* stream);
/* Rows 4t+2 / 4t+3 of the view stack bf16 [4T][NR_KP] from the element tables (rows 4t / 4t+1 come from nr_additive_fwd_ex). */
int nr_views_fill(const int64_t* cat, const int64_t* sub, const float* E, int ncat, uint16_t* views, int64_t T, void* stream);

/* Segmented row reduction dst[ids[i]] += src[perm-order rows] for D-wide f32 rows (ids sorted ascending with their permutation,
 * as nr_embed_scatter_sorted: accumulated into dst); rows with id <= pad_row are skipped (-1: none). */
int nr_scatter_sorted_f32(const int64_t* ids_sorted, const int64_t* perm, const float* src, int64_t ld, float* dst, int64_t num_rows,
                          int64_t n, int pad_row, void* stream);
/* Generic atomic row scatter-add, any width d: dst[ids[i]][0:d] += row_scale[i] * src[i][0:d] (row_scale may be NULL). */
int nr_rows_scatter_add(const int64_t* ids, const float* src, int64_t ld, const float* row_scale, float* dst, int64_t num_rows, int d,
                        int64_t n, int pad_row, void* stream);
/* out[i][0:d] = row_scale[i] * table[ids[i]][0:d] into rows of stride ldo (LSTUR: category / subcategory columns of the 900-d
 * news vector, src/model/LSTUR/news_encoder.py:52-55,69-75; user_embedding row with its dropout2d factor, LSTUR/__init__.py:74-77). */
int nr_gather_rows_strided(const int64_t* ids, const float* table, int64_t num_rows, int d, const float* row_scale, float* out,
                           int64_t ldo, int64_t n, void* stream);

/* ---- LSTUR user encoder: nn.GRU(3F, H) over the packed click history (src/model/LSTUR/user_encoder.py:11-14,27-45) ----
 * Padded sizes for hidden size Hd: Hg = Hd up to 16 (gate stride), Hp = Hd+1 up to 32 (h row length, col Hd of bf16 h rows = 1.0),
 * Kp = 3*Hg up to 32 (dGh row length).  One launch per time step; the input projection gi = x W_ih^T (f32 [B*N][3*Hg], row
 * b*N + t, gate q of unit j at column q*Hg + j, no bias) is one plain GEMM done by the caller.
 * The step-to-step operands (h_t, dGh_t) are exchanged in tile order too (see nr_pack_qkv), rows padded up to 16. */
int nr_gru_dims(int Hd, int* Hg, int* Hp, int* Kp);
/* W f32 [3*Hd][K] (weight_ih_l0 / weight_hh_l0, gate order r,z,n) -> dst bf16 [3*Hg][Kpad] (row q*Hg+j) and, if not NULL,
 * dstT bf16 [Kpad][Kp] with dstT[k][q*Hg+j] = W[q*Hd+j][k] (operand of the hidden-state gradient).  tiled != 0: both in tile
 * order (W_hh for the step kernels); 0: row-major (W_ih for the hoisted GEMMs). */
int nr_pack_gru(const float* W, int Hd, int K, int Kpad, uint16_t* dst, uint16_t* dstT, int tiled, void* stream);
/* f32 rows [n][d] (stride ld) -> bf16 rows [n][dp]: col d = 1.0 when d < dp, rest 0 (MFMA / GEMM operand form of dense vectors). */
int nr_rows_to_bf16(const float* src, int64_t ld, int d, uint16_t* dst, int dp, int64_t n, void* stream);
/* bf16 rows [n][K] row-major -> tile order [n up to 16][K], padding rows zero (h_0 for the first forward step). */
int nr_tile_rows_bf16(const uint16_t* src, int n, int K, uint16_t* dst, void* stream);
/* Step t of the GRU forward for all B samples: reads h_{t-1} (h_in_t: bf16 operand in tile order [B up to 16][Hp], h_in_f: f32
 * state [B][Hp]), writes h_t as h_out_t (tile order, zero-initialised by the caller once: padding is never written), h_out_f and,
 * if not NULL, h_out_b (row-major bf16 [B][Hp] with col Hd = 1.0: what the backward sweep and the W_hh gradient GEMM read);
 * samples with t >= len[b] keep their state (pack_padded_sequence: the first len[b] slots are consumed, len >= 1).  Whh in tile
 * order (nr_pack_gru tiled).  gates (training): bf16 [B][4][Hg] = r, z, n, q = Gh_n + b_hn of this step; NULL for inference. */
int nr_gru_fwd_step(const float* gi, const uint16_t* Whh, const float* b_ih, const float* b_hh, const int32_t* len, const uint16_t* h_in_t,
                    uint16_t* h_out_b, uint16_t* h_out_t, const float* h_in_f, float* h_out_f, uint16_t* gates, int B, int N, int Hd, int t,
                    void* stream);
/* Step t of the backward sweep (t = T-1 .. 0, then t = -1): dh_t = carry_next + dgh_next @ W_hh (g_last when first != 0);
 * for t >= 0 it then writes the gate gradients of step t: dgi row b*N+t of bf16 [B*N][Kp] = [dr|dz|dn] pre-activations,
 * dgh bf16 [B][Kp] = [dr|dz|dn*r] and carry f32 [B][Hp] = dh_t * z_t (dh_t for finished samples); for t = -1 carry
 * receives dh_0 (gradient of the initial state = the user_embedding row in the 'ini' method).  dgh_next and WhhT are in tile
 * order; dgh_t receives dgh of this step in tile order ([B up to 16][Kp], zero-initialised once by the caller) for the next launch. */
int nr_gru_bwd_step(const float* g_last, const uint16_t* dgh_next, const float* carry_next, const uint16_t* WhhT, const uint16_t* gates,
                    const uint16_t* h_prev_b, const int32_t* len, uint16_t* dgi, uint16_t* dgh, uint16_t* dgh_t, float* carry, int B, int N,
                    int Hd, int t, int first, void* stream);

/* The whole recurrence in one call (the host pays one FFI crossing instead of 2T+1: at ~100 launches per LSTUR step the Python
 * launch loop, not the GPU, set the step time on slower hosts).  Forward: steps t = 0..T-1 as nr_gru_fwd_step; h_t2 = two tile-order
 * state buffers [2][B up to 16][Hp] (h_0 in the first, the rest zero), h_f2 = two f32 state buffers [2][B][Hp] (h_0 in the first);
 * step t reads buffer t%2 and writes (t+1)%2, so h_T is in buffer T%2.  H_all bf16 [T+1][B][Hp] (row block t+1 receives h_t+1;
 * block 0 = h_0 is the caller's) and gates bf16 [T][B][4][Hg] are for training, both NULL for inference.
 * Backward: steps t = T-1..0 and the final t = -1 as nr_gru_bwd_step; dgh bf16 [T][B][Kp], dgh_t2 [2][B up to 16][Kp] (zeroed once),
 * carry2 f32 [2][B][Hp]; launch i uses buffers i%2, so dh_0 ends in carry2 buffer T%2. */
int nr_gru_fwd_seq(const float* gi, const uint16_t* Whh, const float* b_ih, const float* b_hh, const int32_t* len, uint16_t* h_t2,
                   uint16_t* H_all, float* h_f2, uint16_t* gates, int B, int N, int Hd, int T, void* stream);
int nr_gru_bwd_seq(const float* g_last, const uint16_t* WhhT, const uint16_t* gates, const uint16_t* H_all, const int32_t* len, uint16_t* dgi,
                   uint16_t* dgh, uint16_t* dgh_t2, float* carry2, int B, int N, int Hd, int T, void* stream);

/* The same entry points with an explicit buffer count (n_buf >= 2 tile-order step buffers; the sweep uses buffers t % 2).  Rounds 2-3 ran the whole
 * sweep as ONE persistent launch when n_buf >= T + 1 (grid-wide barrier between the steps); that form measured slower on MI355X in every variant
 * and was removed in round 4 (csrc/k_gru.h): nr_gru_seq_buffers now always answers 2. */
int nr_gru_seq_buffers(int B, int Hd, int T);
int nr_gru_fwd_seq_n(const float* gi, const uint16_t* Whh, const float* b_ih, const float* b_hh, const int32_t* len, uint16_t* h_t2, int n_buf,
                     uint16_t* H_all, float* h_f2, uint16_t* gates, int B, int N, int Hd, int T, void* stream);
/* Inference sweep over histories that INDEX a table of per-item input projections (LSTUR evaluation, src/evaluate.py:218-233 +
 * model/LSTUR/user_encoder.py:27-45: every clicked-news vector is a row of the news matrix, so x_t W_ih^T is computed once per news, not once per
 * (history, position)): gi f32 [n_rows][3*Hg], gi_row int32 [B][N] (row of gi for sample b at step t); h_f f32 [B][Hp] holds h0 on entry and the
 * state after len[b] steps on return (updated in place); h_t2: two tile-order bf16 buffers [ceil16(B)][Hp], the first holding h0
 * (nr_tile_rows_bf16).  active (HOST pointer, optional): active[t] = number of leading samples with len > t when the samples are sorted by
 * length, longest first -- step t is then launched for those rows only (pack_padded_sequence's batch_sizes). */
int nr_gru_fwd_seq_rows(const float* gi, const int32_t* gi_row, const uint16_t* Whh, const float* b_ih, const float* b_hh, const int32_t* len,
                        uint16_t* h_t2, float* h_f, const int32_t* active, int B, int N, int Hd, int T, void* stream);
/* Gate stage of ONE step of the same sweep for large batches, where the recurrent product runs as a plain GEMM (gh f32 [B][3*Hg] =
 * bf16(h) @ bf16(W_hh)^T, rows in nr_pack_gru's order): r, z, n and the state update of user_encoder.py:27-45 / torch.nn.GRU for the first B
 * samples; h_f f32 [B][Hp] updated in place, h_b bf16 [B][Hp] = the new state (column Hd = 1.0), the next step's GEMM operand. */
int nr_gru_gate_rows(const float* gi, const int32_t* gi_row, const float* gh, const float* b_ih, const float* b_hh, const int32_t* len, float* h_f,
                     uint16_t* h_b, int B, int N, int Hd, int t, void* stream);
int nr_gru_bwd_seq_n(const float* g_last, const uint16_t* WhhT, const uint16_t* gates, const uint16_t* H_all, const int32_t* len, uint16_t* dgi,
                     uint16_t* dgh, uint16_t* dgh_t2, int n_buf, float* carry2, int B, int N, int Hd, int T, void* stream);

/* ---- GENERAL GEOMETRY (csrc/k_generic.h): src/config.py's model-geometry knobs away from the tuned instantiation ------------------------------
 * word_embedding_dim (config.py:34), num_attention_heads (:45, any divisor with d_k <= 32), num_filters (:54), window_size (:55, odd),
 * query_vector_dim (:39): the reference builds any of them (multihead_self.py:27-38, NAML/news_encoder.py:10-19, LSTUR/news_encoder.py:23).
 * On this path every dense contraction runs in nr_gemm_nt / nr_gemm_tn on bf16 operands (rows padded to a multiple of 32 columns, one
 * column of 1.0 that carries the bias -- nr_rows_to_bf16 / nr_g_rows_to_seqpad); these entry points are everything else, on f32 rows.
 * Limits: sequence length <= 64, d_k <= 32; counts of elements multiples of 4 where stated.  Not tuned (the tuned kernels keep 300 / 15 /
 * 300 / 3).
 *   nr_g_dropout         y = F.dropout(x) (news_encoder.py:38-40,43-45): element i uses counter elem0 + i of dropout `site` -- nr_dropout_mask's
 *                        numbering; the same call on a gradient is the backward.  n_elem, elem0 multiples of 4; y may alias x.
 *   nr_g_attn_fwd        ScaledDotProductAttention (multihead_self.py:15-23: exp / (sum + 1e-8), optional key lengths :60-70) per (sequence,
 *                        head) from qkv f32 [n_seq * S][ld] (Q at column 0, K at H * dk, V at 2 H * dk) -> ctx f32 [n_seq * S][H * dk].
 *   nr_g_attn_bwd        its backward: dctx f32 [n_seq * S][H * dk] -> dqkv f32 [n_seq * S][ld] (columns 0 .. 3 H dk - 1 written).
 *   nr_g_additive_fwd    AdditiveAttention (additive.py:27-53) given proj = x Wa^T + ba (f32 [n_seq * S][ldp], from the GEMM): softmax over
 *                        the first `valid` tokens of q . tanh(proj), out f32 [n_seq][ldo], attn_w f32 [n_seq][S] (may be NULL).
 *   nr_g_additive_bwd    g_out f32 [n_seq][ldg] -> dpre f32 [n_seq * S][ldq] (gradient of proj) and dq_part f32 [n_seq][Q] (sum over rows =
 *                        gradient of attention_query_vector); the input gradient is dpre @ Wa (a GEMM) + the direct term:
 *   nr_g_rows_axpy       y[r][0:d] (+)= a[r] * g[r / S][0:d]  (a = the attention weights, g = g_out).
 *   nr_g_rows_to_seqpad  f32 token rows -> bf16 rows of dp columns in the seqpad layout of a window-w convolution (pad = (w - 1) / 2: token s
 *                        of sequence q at row pad + q (S + pad) + s; the other rows must be zero: zero-fill once), column d = 1.0 when `one`.
 *                        Conv2d(1, F, (w, D), padding = (pad, 0)) (NAML/news_encoder.py:15-17, LSTUR/news_encoder.py:24-28) is then ONE
 *                        nr_gemm_nt with A = that buffer, lda = dp, K = w * dp (the w rows of a window are contiguous), M = n_seq (S + pad)
 *                        virtual rows; its weight gradient ONE nr_gemm_tn with taps = w.
 *   nr_g_relu_drop       act f32 [n_tok][F] = dropout(relu(y[virtual row])) (news_encoder.py: F.dropout(F.relu(conv))), site 2 counters from elem0.
 *   nr_g_relu_drop_bwd   dy bf16 seqpad rows (fp columns) = dact * [act != 0] / (1 - p).
 *   nr_g_unpad_rows      virtual GEMM rows -> token rows;  nr_g_relu  y = scale * x where gate > 0 (gate = x when NULL), else 0: relu forward, and the
 *                        backward of dropout(relu(.)) read off the output's zeros (ElementEncoder, NAML news_encoder.py:40-47; TextEncoder :29-32). */
int nr_g_dropout(const float* x, float* y, int64_t n_elem, int64_t elem0, float p, uint64_t seed, int site, void* stream);
int nr_g_attn_fwd(const float* qkv, int64_t ld, float* ctx, const int32_t* key_len, int64_t n_seq, int S, int H, int dk, void* stream);
int nr_g_attn_bwd(const float* qkv, int64_t ld, const float* dctx, float* dqkv, const int32_t* key_len, int64_t n_seq, int S, int H, int dk, void* stream);
int nr_g_additive_fwd(const float* x, int64_t ldx, int D, const float* proj, int64_t ldp, int Q, const float* qv, float* out, int64_t ldo, float* attn_w,
                      int64_t n_seq, int S, int valid, void* stream);
int nr_g_additive_bwd(const float* x, int64_t ldx, int D, const float* proj, int64_t ldp, int Q, const float* qv, const float* attn_w, const float* g_out,
                      int64_t ldg, float* dpre, int64_t ldq, float* dq_part, int64_t n_seq, int S, void* stream);
int nr_g_rows_axpy(float* y, int64_t ldy, const float* a, const float* g, int64_t ldg, int S, int d, int64_t n_rows, int accumulate, void* stream);
int nr_g_rows_to_seqpad(const float* src, int64_t lds, int d, uint16_t* dst, int dp, int S, int pad, int64_t n_tok, int one, void* stream);
int nr_g_relu_drop(const float* y, int64_t ldy, float* act, int F, int S, int pad, int64_t n_tok, float p, uint64_t seed, int64_t elem0, void* stream);
int nr_g_relu_drop_bwd(const float* dact, const float* act, uint16_t* dy, int F, int fp, int S, int pad, int64_t n_tok, float p, void* stream);
int nr_g_unpad_rows(const float* src, int64_t lds, float* dst, int d, int S, int pad, int64_t n_tok, void* stream);
int nr_g_relu(const float* x, const float* gate, float* y, int64_t n, float scale, void* stream);
/* f32 rows -> bf16 rows [hi | hi | lo] of 3 dp columns (hi = bf16(x), lo = bf16(x - hi), column d of both hi blocks = 1.0): with weights packed
 * [Wh | Wl | Wh] ONE nr_gemm_nt (K = 3 dp) gives x W^T + b to ~2^-16 relative -- the ElementEncoder's linear layer (NAML news_encoder.py:46),
 * whose relu must not flip against the fp32 reference. */
int nr_g_rows_split_bf16(const float* src, int64_t ld, int d, uint16_t* dst, int dp, int64_t n, void* stream);
/* nr_gemm_nt with OVERLAPPING A rows (lda < K allowed): row m of the product reads the K contiguous elements from A + m * lda -- the w rows of a
 * convolution window in a seqpad buffer (nr_g_rows_to_seqpad).  The caller guarantees (M - 1) * lda + K readable elements. */
int nr_gemm_nt_rows(const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, float* C, int64_t ldc, int64_t M, int N, int K, void* stream);

/* Per-impression ranking metrics of src/evaluate.py:24-42,160-168 for a CSR batch of impressions: scores f32[nnz], labels
 * int32[nnz] (0/1), ptr int64[n_impr+1]; out f32[n_impr][4] = AUC, MRR, nDCG@5, nDCG@10 (four NaNs when an impression has a
 * single label class: the reference's ValueError path).  Replaces the multiprocessing pool of src/evaluate.py:267-268. */
int nr_impression_metrics(const float* scores, const int32_t* labels, const int64_t* ptr, float* out, int64_t n_impr, void* stream);

/* ---- optimiser: torch.optim.Adam(model.parameters(), lr) of src/train.py:127-128, stepped at :227-233 (defaults: betas (0.9, 0.999),
 * eps 1e-8, no weight decay, no amsgrad), on flat fp32 buffers.  sched f32[2 * (step + 1)] holds the per-step scalars
 * sched[2 s] = lr / (1 - beta1^s), sched[2 s + 1] = sqrt(1 - beta2^s), computed by the host in double as torch does; `step` is 1-based.
 * Element update: m += (g - m)(1 - beta1); v = v beta2 + (1 - beta2) g g; p -= sched[2 s] * m / (sqrt(v) / sched[2 s + 1] + eps).
 * nr_adam_flat: one pass over n elements; g is multiplied by grad_scale on the way in (1 / world size after a summing all-reduce) and
 * cleared when zero_grad != 0 (replaces optimizer.zero_grad(), src/train.py:229).  All four buffers 16-byte aligned. */
int nr_adam_flat(float* p, float* g, float* m, float* v, int64_t n, const float* sched, int64_t step, double beta1, double beta2, double eps,
                 float grad_scale, int zero_grad, void* stream);

/* Row-sparse form of the same update for a table of which a step touches few rows (LSTUR user_embedding,
 * src/model/LSTUR/__init__.py:38-42: one row per sample).  The dense recurrence is evaluated lazily and exactly: last int32[num_rows] is
 * the step up to which (p, m, v) of a row are current (0 = the row never received a gradient: m = v = 0, dense Adam does not move it);
 * the idle steps a row missed are replayed one by one when the row is needed.  d <= 1024.
 *   nr_row_adam_catchup  rows ids[0..n) (repeats allowed) -> current as of step `upto`; call before the forward gathers them.
 *   nr_row_adam_flush    every row with state -> current as of `upto` (before state_dict() / checkpoints / full-table reads).
 *   nr_row_adam_step     optimiser step `step` for the rows that received gradients: (id, gradient row) pairs of all ranks, sorted by id
 *                        (ids_sorted int64[n], perm = index of the pair's row in rows f32[.][ld]).  Rows of one id are summed in
 *                        sorted-position order (deterministic), scaled by grad_scale; ids <= pad_row are skipped (padding_idx).
 * With a device step counter attached (nr_set_step_counter) nr_row_adam_step takes `step` from *ctr and nr_row_adam_catchup takes `upto`
 * = *ctr - 1 (the counter is bumped at the START of a step, so inside a step one step fewer has been taken): a row-sparse training step
 * captured into a HIP graph then advances at every replay.  While a counter is attached, call nr_row_adam_catchup only from inside a step;
 * nr_row_adam_flush always uses its by-value argument. */
int nr_row_adam_catchup(const int64_t* ids, int64_t n, float* p, float* m, float* v, int32_t* last, int64_t num_rows, int d, const float* sched,
                        int64_t upto, double beta1, double beta2, double eps, void* stream);
/* The same with an explicit choice: by_value != 0 takes `upto` from the argument even while a device step counter is attached -- for a forward
 * issued OUTSIDE a step (validation between graph replays): the counter is bumped at the START of a step, so between steps it equals the number
 * of completed steps and counter - 1 would leave the rows one step stale. */
int nr_row_adam_catchup_ex(const int64_t* ids, int64_t n, float* p, float* m, float* v, int32_t* last, int64_t num_rows, int d, const float* sched,
                           int64_t upto, int by_value, double beta1, double beta2, double eps, void* stream);
int nr_row_adam_flush(float* p, float* m, float* v, int32_t* last, int64_t num_rows, int d, const float* sched, int64_t upto, double beta1,
                      double beta2, double eps, void* stream);
int nr_row_adam_step(const int64_t* ids_sorted, const int64_t* perm, int64_t n, const float* rows, int64_t ld, float* p, float* m, float* v,
                     int32_t* last, int64_t num_rows, int d, const float* sched, int64_t step, double beta1, double beta2, double eps,
                     float grad_scale, int pad_row, void* stream);

/* Stable sort of token ids with their positions for the embedding backward (autograd of nn.Embedding, src/model/NRMS/news_encoder.py:38;
 * consumer: nr_embed_scatter_sorted).  ids int64[n] (values outside [0, num_rows) are clamped like the forward gather) ->
 * ids_sorted int64[n] ascending, perm int64[n] (original position; equal ids keep their order).  LSD radix on ceil(log2(num_rows))
 * bits in passes of <= 9 bits.  workspace: nr_sort_ids_workspace(n, num_rows) bytes (16-byte aligned), -1 if the sizes are unsupported
 * (n < 2^31, num_rows <= 2^27). */
int64_t nr_sort_ids_workspace(int64_t n, int64_t num_rows);
int nr_sort_ids(const int64_t* ids, int64_t n, int64_t num_rows, int64_t* ids_sorted, int64_t* perm, void* workspace, int64_t workspace_bytes,
                void* stream);

/* Debug/verification helper: the keep-mask (1.0/0.0) the fused kernels use for dropout `site`
 * (1 = embedding output, 2 = MHSA output) over n_elem consecutive elements. */
int nr_dropout_mask(float* mask, int64_t n_elem, float p_drop, uint64_t seed, int site, void* stream);

/* Hardware probe: runs one v_mfma_f32_16x16x32_bf16 on A[16][32], B[32][16] (bf16 bits, row-major)
 * and writes D f32[16][16]; used by the GPU tests to pin the fragment-layout assumptions. */
/* ---- device-resident step counter: what lets ONE captured HIP graph of a training step (forward + backward + Adam; the reference's
 * train.py:183-233 loop body) be replayed for every step.  Seeds and the optimiser's step index are by-value arguments, frozen into the
 * graph's kernel nodes at capture; with a counter attached, every dropout kernel folds *ctr into its key (a new mask per replay) and
 * nr_adam_flat takes its step index from *ctr.  The graph's first node is nr_step_counter_add(ctr, 1).  nr_dropout_mask follows the same
 * rule, so masks stay reproducible by the tests.  ctr: uint32 in device memory, owned by the caller; NULL detaches (the default). */
int nr_set_step_counter(const uint32_t* ctr);
int nr_step_counter_add(uint32_t* ctr, uint32_t inc, void* stream);

int nr_probe_mfma(const uint16_t* A, const uint16_t* B, float* D, void* stream);
/* Test hook: LDS transpose read (ds_read_b64_tr_b16) of a 4096-entry LDS ramp lds[i] = i; lane l reads at byte offset offs[l] (int32[64],
 * 8-byte aligned, < 8184) and stores its four elements to out u16[64][4]. */
int nr_probe_tr16(const int32_t* offs, uint16_t* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif

/*
 * vita_hip.h — C ABI of libvita_hip.so, the MI355X (gfx950) kernels behind the ViTa
 * pretraining hot path.
 *
 * The reference (Yundi-Zhang/ViTa) is pure Python/PyTorch: it has NO FFI layer.  Each entry
 * point below therefore cites the reference Python call it replaces (file:line relative to the
 * reference checkout); the Python host mirror (vita_amd/ *.py) binds these with ctypes — see
 * INTEGRATION.md.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (PyTorch caching allocator);
 *     the library never allocates, frees or synchronises, and never changes the device.
 *   - `stream` is a hipStream_t passed as void*.
 *   - return value: 0 = OK; VITA_ERR_* (negative) otherwise.  No C++ exception crosses.
 *   - bf16 buffers are raw uint16 storage; "ld" = leading dimension in ELEMENTS.
 *   - bf16 buffers and lds must be 16-byte aligned / multiples of 8 elements.
 *   - re-entrant (calls arrive from the autograd worker thread too): the only process-global state is one-time
 *     per-device set-up behind std::call_once, and the device array of vita_layernorm_bwd_noalloc (documented there).
 */
#ifndef VITA_HIP_H
#define VITA_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VITA_OK 0
#define VITA_ERR_INVALID_ARG (-1)
#define VITA_ERR_UNSUPPORTED (-2)
#define VITA_ERR_LAUNCH (-3)

int vita_version(void);
/* One wave that holds a CU for `us` microseconds (<= 100 ms) on `stream`.  No counterpart in the reference: the host side
 * launches two of them to test whether two HIP streams really run concurrently (streams mapped to one hardware queue
 * serialise) before it puts the weight-gradient GEMMs of the Block backward on a second stream. */
int vita_spin_us(int64_t us, void* stream);
/* last HIP error string captured by a failed launch on this thread ("" if none) */
const char* vita_last_error(void);

/* ------------------------------------------------------------------ GEMM (bf16 MFMA, fp32 acc)
 * C[m,n] = epilogue( alpha * sum_k A(m,k) * B(k,n) )          (batched: z = outer*inner)
 *   a_layout 0: A stored [M][K] (k contiguous)    1: stored [K][M] (m contiguous)
 *   b_layout 0: B stored [N][K] (k contiguous)    1: stored [K][N] (n contiguous)
 * Replaces torch nn.Linear fwd / dgrad / wgrad inside timm Block (call sites
 * networks/imaging_encoders.py:36-40,113-114), Conv3d-as-GEMM (networks/tokenizers.py:46),
 * q@k^T / attn@v of the materialised-attention path, and the CLIP logits matmul
 * (utils/clip_loss.py:33).
 * Reduction-dim pads of k-contiguous operands must be zero up to a multiple of 8.
 */
enum { VITA_EPI_NONE = 0, VITA_EPI_GELU = 1, VITA_EPI_RESIDUAL = 2, VITA_EPI_GELU_BWD = 3 };
enum { VITA_DT_BF16 = 0, VITA_DT_F32 = 1 };

typedef struct vita_gemm_desc {
  const void* a;
  const void* b;
  void* c;
  const float* bias;   /* [n] fp32 or NULL (added before the activation) */
  const void* aux_in;  /* EPI_RESIDUAL: fp32 [m][ld_aux]; EPI_GELU_BWD: bf16 GELU derivative [m][ld_aux] (C *= aux) */
  void* aux_out;       /* EPI_GELU: bf16 GELU'(pre-activation) [m][ldc], saved for the backward pass */
  int64_t m, n, k;
  int64_t lda, ldb, ldc, ld_aux;
  int64_t batch_outer, batch_inner;
  int64_t a_stride_outer, a_stride_inner;
  int64_t b_stride_outer, b_stride_inner;
  int64_t c_stride_outer, c_stride_inner;
  int32_t a_layout, b_layout;
  int32_t c_dtype;     /* VITA_DT_* */
  int32_t epilogue;    /* VITA_EPI_* */
  int32_t split_k;     /* >1: fp32 C, EPI_NONE, atomically accumulated into a ZEROED c */
  /* optional physical->logical index maps for fp32 C (weight-grad un-padding):
   * index i maps to (i / seg_phys) * seg_log + (i % seg_phys), dropped when (i % seg_phys) >= seg_log.
   * 0 = identity. */
  int32_t row_seg_phys, row_seg_log, col_seg_phys, col_seg_log;
  float alpha;
} vita_gemm_desc;

int vita_gemm_bf16(const vita_gemm_desc* d, void* stream);
/* Same, with the kernel chosen by the caller (tests, tuning): variant 0 = by shape, 1 = 128x128x64
 * tile kernel, 13 / 14 / 16 = 256 x (16*variant) x 64 LDS-DMA kernel (un-batched problems only). */
int vita_gemm_bf16_ex(const vita_gemm_desc* d, int32_t variant, void* stream);

/* ------------------------------------------------------------------ weight shadows
 * fp32 logical [rows_log][cols_log] -> bf16 physical [rows_phys][ld_phys] with the same
 * segment maps as above (pads written as zero).  Refreshes the padded bf16 copies of
 * nn.Linear / Conv3d weights after an optimizer step. */
int vita_pack_weight_bf16(const float* w, int64_t rows_log, int64_t cols_log, void* out,
                          int64_t rows_phys, int64_t ld_phys, int32_t row_seg_phys,
                          int32_t row_seg_log, int32_t col_seg_phys, int32_t col_seg_log,
                          void* stream);
/* fp32 vector [n_log] -> fp32 padded [n_phys] with a segment map (bias shadows) */
int vita_pack_vec_f32(const float* v, int64_t n_log, float* out, int64_t n_phys,
                      int32_t seg_phys, int32_t seg_log, void* stream);
/* inverse: padded fp32 [n_phys] -> logical [n_log] (bias grads) */
int vita_unpack_vec_f32(const float* v, int64_t n_phys, float* out, int64_t n_log,
                        int32_t seg_phys, int32_t seg_log, void* stream);

/* ------------------------------------------------------------------ LayerNorm
 * nn.LayerNorm(D, eps) over the last dim (timm Block norm1/norm2;
 * networks/imaging_encoders.py:41,115).  x fp32 [rows][ldx]; y bf16 or fp32 [rows][ldy]
 * (columns [d, ldy) are written as zero); mean/rstd fp32 [rows] saved for backward. */
int vita_layernorm_fwd(const float* x, int64_t ldx, const float* gamma, const float* beta,
                       void* y, int64_t ldy, int32_t y_dtype, float* mean, float* rstd,
                       int64_t rows, int64_t d, float eps, void* stream);
/* dx_out[r] = (dx_in ? dx_in[r] : 0) + LN-backward(dy[r]);  dgamma/dbeta are [d] fp32 sums over
 * rows written via partial buffer `partial` fp32 [2][nblk][d] (nblk from
 * vita_layernorm_bwd_nblk).  dy is bf16 or fp32.  dx_bf16 (optional) receives a bf16 copy of
 * dx_out with zeroed pads [rows][ld_dxb]. */
int64_t vita_layernorm_bwd_nblk(int64_t rows);
int vita_layernorm_bwd(const void* dy, int64_t lddy, int32_t dy_dtype, const float* x, int64_t ldx,
                       const float* gamma, const float* mean, const float* rstd,
                       const float* dx_in, float* dx_out, int64_t lddx, void* dx_bf16,
                       int64_t ld_dxb, float* dgamma, float* dbeta, float* partial, int64_t rows,
                       int64_t d, void* stream);
/* vita_layernorm_bwd with the row sums ADDED to dgamma / dbeta (one writer per element, no atomics): what autograd's
 * `weight.grad += dgamma` does for norm1 / norm2 (networks/imaging_encoders.py:41,115) without the extra launches. */
int vita_layernorm_bwd_acc(const void* dy, int64_t lddy, int32_t dy_dtype, const float* x, int64_t ldx,
                           const float* gamma, const float* mean, const float* rstd,
                           const float* dx_in, float* dx_out, int64_t lddx, void* dx_bf16,
                           int64_t ld_dxb, float* dgamma, float* dbeta, float* partial, int64_t rows,
                           int64_t d, void* stream);

/* vita_layernorm_bwd for callers that cannot allocate device memory (code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc).  No counterpart in the reference.  Same arguments and results (bit-identical: the
 * same kernels run) minus `partial`: the dgamma / dbeta partials live in a file-scope device array of the library,
 * 12 MiB per GPU, held until the process ends.  Limit: d <= 2048 (VITA_ERR_UNSUPPORTED beyond, nothing launched).
 * Calls no hipMalloc / hipFree or any allocating routine.  Every call on one GPU shares that GPU's array: calls that
 * overlap on one GPU (two streams) race - serialise them. */
int vita_layernorm_bwd_noalloc(const void* dy, int64_t lddy, int32_t dy_dtype, const float* x, int64_t ldx,
                               const float* gamma, const float* mean, const float* rstd,
                               const float* dx_in, float* dx_out, int64_t lddx, void* dx_bf16,
                               int64_t ld_dxb, float* dgamma, float* dbeta, int64_t rows, int64_t d,
                               void* stream);

/* ------------------------------------------------------------------ masking (integer-exact)
 * Masker.random_masking (utils/imaging_model_related.py:255-286) with the noise passed in:
 * per row argsort(noise) [ties: ascending index], inverse permutation, binary mask.
 * noise fp32 [n][l] (finite, non-negative) -> ids_shuffle int64 [n][l], ids_restore int64 [n][l],
 * mask fp32 [n][l] (0 keep / 1 drop).  l <= 32768. */
int vita_argsort_mask(const float* noise, int64_t n, int64_t l, int64_t len_keep,
                      int64_t* ids_shuffle, int64_t* ids_restore, float* mask, void* stream);

/* ------------------------------------------------------------------ row gather / scatter
 * torch.gather(x, 1, ids[...,None].repeat(1,1,D)) (utils/imaging_model_related.py:246) on fp32
 * rows: out[b][j] = x[b][ids[b][j]];  rows with ids >= l_in take `fill` (decoder mask-token
 * un-shuffle, networks/imaging_decoders.py:288-291) or zero when fill == NULL.
 * x [nb][l_in][ldx], out [nb][l_out][ldo], ids int64 [nb][l_out] with row stride ld_ids.
 * pos (optional fp32 [l_out][d]) is added to output row j (the decoder's `+ dec_pos_embed`,
 * networks/imaging_decoders.py:292-295). */
int vita_row_gather_f32(const float* x, int64_t ldx, int64_t l_in, const int64_t* ids,
                        int64_t ld_ids, const float* fill, const float* pos, float* out, int64_t ldo,
                        int64_t l_out, int64_t nb, int64_t d, void* stream);
/* backward of the gather for unique ids: dx[b][ids[b][j]] = dout[b][j]; dx must be pre-zeroed.
 * Rows with ids >= l_in are summed into dfill (fp32 [d], pre-zeroed, may be NULL). */
int vita_row_scatter_f32(const float* dout, int64_t ldo, int64_t l_out, const int64_t* ids,
                         int64_t ld_ids, float* dx, int64_t ldx, int64_t l_in, float* dfill,
                         int64_t nb, int64_t d, void* stream);

/* ------------------------------------------------------------------ patch tokenizer pieces
 * PatchEmbed.forward (networks/tokenizers.py:38-56) = im2col of the selected tokens (bf16) ->
 * vita_gemm_bf16 -> vita_tokens_assemble.
 * img fp32 [nb][S][T][H][W]; token id = ((s*t + ti)*h + hi)*w + wi; patch vector order (pt,ph,pw).
 * ids int64 [nb][n_tok] (NULL = identity, n_tok = L).  Token (b,j) is written to row
 * b*out_batch_rows + out_row0 + j of out (bf16 [..][ldo], pads zero); other rows are untouched. */
int vita_im2col_bf16(const float* img, int64_t nb, int64_t S, int64_t T, int64_t H, int64_t W,
                     int64_t pt, int64_t ph, int64_t pw, const int64_t* ids, int64_t n_tok,
                     void* out, int64_t ldo, int64_t out_batch_rows, int64_t out_row0, void* stream);
/* x[b][0] = cls + pos[0];  x[b][1+j] = y[b*y_batch_rows + j] + pos[1 + ids[b][j]]
 * (networks/imaging_encoders.py:91-106).  y fp32 rows of ldy; pos fp32 [1+L][d] or NULL;
 * out fp32 [nb][1+n_tok][ldo] with zero pads.  out may alias y row-for-row (in place). */
int vita_tokens_assemble(const float* y, int64_t ldy, const float* cls, const float* pos,
                         const int64_t* ids, int64_t nb, int64_t n_tok, int64_t d, float* out,
                         int64_t ldo, int64_t y_batch_rows, void* stream);

/* ------------------------------------------------------------------ misc row ops */
/* fp32 [rows][ldx] -> bf16 [rows][ldy], cols [d,ldy) zero */
int vita_cast_f32_bf16(const float* x, int64_t ldx, void* y, int64_t ldy, int64_t rows, int64_t d,
                       void* stream);
/* column sums of a bf16 or fp32 matrix [rows][ld] -> fp32 out[cols] (bias gradients) */
int vita_colsum(const void* x, int32_t dtype, int64_t ld, int64_t rows, int64_t cols, float* out,
                void* stream);
/* same, ADDED to out (fp32 atomics): bias gradients accumulated straight into an optimizer's gradient buffer */
int vita_colsum_acc(const void* x, int32_t dtype, int64_t ld, int64_t rows, int64_t cols, float* out,
                    void* stream);
/* materialised attention (first correct path; timm Attention.forward):
 * p = softmax(scale * s) row-wise.  s fp32 [nrows_total][lds], p bf16 [..][ldp]; columns
 * [n, ldp) of p are written as zero. */
int vita_softmax_rows(const float* s, int64_t lds, void* p, int64_t ldp, int64_t rows, int64_t n,
                      float scale, void* stream);
/* ds = p * (dp - sum_j p*dp) * scale  (bf16 out, pads zero) */
int vita_softmax_bwd_rows(const void* p, int64_t ldp, const float* dp, int64_t lddp, void* ds,
                          int64_t ldds, int64_t rows, int64_t n, float scale, void* stream);

/* ------------------------------------------------------------------ fused attention
 * softmax(q k^T * scale) v per (batch, head) without materialising the N x N scores; replaces
 * timm 0.6.12 Attention.forward's (q @ k^T) * scale -> softmax -> @ v (call sites
 * networks/imaging_encoders.py:36-40, networks/imaging_decoders.py:43-45).
 * qkv bf16 [B][N][3][H][hdp] (head dim zero-padded to hdp), o bf16 [B][N][H][hdp], lse fp32 [B][H][N].
 * vita_attn_supported(hd, hdp) != 0 tells whether a kernel is instantiated for that head size. */
int vita_attn_supported(int64_t hd, int64_t hdp);
/* ws (optional, 16-byte aligned, ws_bytes >= vita_attn_fwd_ws_bytes(hdp), a fixed ~0.66 MB * hdp): scratch in which
 * the workgroups of a partly filled last round, launched with their key sweep cut into ranges, leave fp32 partial
 * outputs that a combine kernel merges (fixed order); with ws == NULL the grid is launched whole. */
int64_t vita_attn_fwd_ws_bytes(int64_t hdp);
int vita_attn_fwd(const void* qkv, void* o, float* lse, void* ws, int64_t ws_bytes, int64_t B, int64_t N,
                  int64_t H, int64_t hd, int64_t hdp, float scale, void* stream);
/* backward: recomputes P from q, k and lse; deterministic (no atomics).  dout bf16 [B][N][H][hdp];
 * delta fp32 [B][H][2][N] scratch (written here: a copy of lse and rowsum(dout*o)); dqkv bf16 like qkv, every element of
 * the three parts is written (pads zero).
 * ws (optional, 16-byte aligned, ws_bytes >= vita_attn_bwd_ws_bytes(B, N, H, hdp) = B*H*N^2*2 B rounded up to
 * tiles + a fixed 0.66 MB * hdp): scratch through which the dK/dV kernel hands its bf16 dS tiles to a streaming dQ
 * kernel, so dQ costs one product instead of three, and in which the workgroups of a partly filled last round leave
 * fp32 partial dK/dV (summed in a fixed order); with ws == NULL the dQ kernel recomputes S and dP itself and the
 * dK/dV grid is launched whole (no extra memory). */
int64_t vita_attn_bwd_ws_bytes(int64_t B, int64_t N, int64_t H, int64_t hdp);
int vita_attn_bwd(const void* qkv, const void* o, const void* dout, const float* lse, float* delta,
                  void* dqkv, void* ws, int64_t ws_bytes, int64_t B, int64_t N, int64_t H, int64_t hd,
                  int64_t hdp, float scale, void* stream);

/* ------------------------------------------------------------------ losses
 * CLIPLoss.forward (utils/clip_loss.py:27-40), kept in fp32:
 *   F.normalize(x, dim=1) (eps 1e-12)           -> vita_l2_normalize_fwd / _bwd
 *   logits = out0 @ out1.T / tau                -> vita_sgemm_f32 (alpha = 1/tau)
 *   lambda * CrossEntropy(logits, arange)       -> vita_ce_rows (label = label_offset + row)
 * With global negatives each rank evaluates its local rows of logits and of logits.T against the
 * all-gathered other tower, so both CE terms are row-wise. */
int vita_l2_normalize_fwd(const float* x, int64_t rows, int64_t d, float eps, float* y, float* inv_norm,
                          void* stream);
int vita_l2_normalize_bwd(const float* dy, const float* y, const float* inv_norm, int64_t rows, int64_t d,
                          float* dx, void* stream);
/* C = alpha * A(m,k) B(k,n) + beta * C, fp32 on the vector ALUs (layouts as vita_gemm_bf16) */
int vita_sgemm_f32(const float* a, const float* b, float* c, int64_t m, int64_t n, int64_t k, int64_t lda,
                   int64_t ldb, int64_t ldc, int32_t a_layout, int32_t b_layout, float alpha, float beta,
                   void* stream);
/* loss[0] += weight * sum_rows (logsumexp(row) - row[label]);  dlogits = weight * (softmax(row) - onehot)
 * (dlogits / row_lse may be NULL).  loss must be zeroed by the caller. */
int vita_ce_rows(const float* logits, int64_t ld, int64_t rows, int64_t cols, int64_t label_offset,
                 float weight, float* loss, float* row_lse, float* dlogits, int64_t ldd, void* stream);
/* sigmoid + MSE against the patchified image (networks/imaging_decoders.py:303-304,
 * utils/imaging_model_related.py:103-117, utils/losses.py:117-139):
 * pred = sigmoid(z[b][1+l][:P]); loss[0] += inv_count * sum w_l (pred - target)^2;
 * dz[b][1+l][e] = 2 (pred-target) pred (1-pred) w_l inv_count (bf16, cls row and pads zero).
 * z fp32 [nb][1+L][ldz]; mask fp32 [nb][L] or NULL (mask_loss rows weight); pred optional fp32 [nb][L][P]. */
int vita_sigmoid_mse_patch(const float* z, int64_t ldz, const float* img, int64_t nb, int64_t S,
                           int64_t T, int64_t H, int64_t W, int64_t pt, int64_t ph, int64_t pw,
                           const float* mask, float inv_count, float* pred, float* loss, void* dz_bf16,
                           int64_t ld_dz, void* stream);

/* ------------------------------------------------------------------ Stage-II glue
 * TabularViT.forward_embeddings + cls + pos-embed (networks/tabular_encoders.py:80-93, 96-100) in one launch:
 *   out[b][0]   = cls + pos[0]
 *   out[b][1+f] = (n_class[f] == 0 ? x[b][f] * w_f + b_f : table_f[(long) x[b][f]]) + pos[1+f]
 * x fp32 [B][F]; wptr / bptr int64 [F] (device memory): device addresses of the per-feature fp32 parameters
 * (numerical_fe.i.weight [D][1] and .bias [D]; single_c_fe.i / multi_c_fe.i.weight [n_class][D], bptr 0);
 * n_class int32 [F] (0 = numerical); cls fp32 [D]; pos fp32 [1+F][D] or NULL; out fp32 [B][1+F][ldo], pads zero.
 * An index outside [0, n_class) is clamped (nn.Embedding would raise). */
int vita_tabular_embed_fwd(const float* x, int64_t B, int64_t F, const int64_t* wptr, const int64_t* bptr,
                           const int32_t* n_class, const float* cls, const float* pos, float* out, int64_t ldo,
                           int64_t D, void* stream);
/* backward: ADDS (one owner thread per element, fixed order, no atomics) the batch sums into the gradient tensors whose
 * device addresses gw / gb hold (int64 [F], same shapes as the parameters, 0 = skip) and into dcls fp32 [D] (or NULL). */
int vita_tabular_embed_bwd(const float* dout, int64_t ldo, const float* x, int64_t B, int64_t F, const int32_t* n_class,
                           const int64_t* gw, const int64_t* gb, float* dcls, int64_t D, void* stream);
/* y.mean(dim=1) over ALL tokens (models/MultimodalSimCLR.py:121,133): x fp32 [B][N][ld] (ld % 4 == 0, pads zero)
 * -> out fp32 [B][D]; backward dx[b][n][:D] = dy[b] / N, pads zero. */
int vita_token_mean_fwd(const float* x, int64_t ld, int64_t B, int64_t N, int64_t D, float* out, void* stream);
int vita_token_mean_bwd(const float* dy, int64_t B, int64_t N, int64_t D, float* dx, int64_t ld, void* stream);
/* nn.BatchNorm1d(D) of lightly's SimCLRProjectionHead (call sites models/MultimodalSimCLR.py:51-53,62-64), training mode:
 * per-column batch mean / biased variance of x fp32 [B][D], y = (x - mean) * rstd * gamma + beta, then ReLU when relu != 0;
 * mean / rstd fp32 [D] are saved for the backward; running_mean / running_var (optional) are updated in place with
 * `momentum` and the unbiased variance, as torch does.  Local (per-rank) statistics: no SyncBN, like the reference. */
int vita_batchnorm_fwd(const float* x, int64_t B, int64_t D, const float* gamma, const float* beta, float eps,
                       int32_t relu, float* y, float* mean, float* rstd, float* running_mean, float* running_var,
                       float momentum, void* stream);
/* eval mode: running statistics instead of batch statistics */
int vita_batchnorm_eval(const float* x, int64_t B, int64_t D, const float* gamma, const float* beta, float eps,
                        int32_t relu, const float* running_mean, const float* running_var, float* y, void* stream);
/* backward of vita_batchnorm_fwd; y is its output (the ReLU mask is y > 0).  dgamma / dbeta fp32 [D] are overwritten. */
int vita_batchnorm_bwd(const float* dy, const float* x, const float* y, int64_t B, int64_t D, const float* gamma,
                       const float* mean, const float* rstd, int32_t relu, float* dx, float* dgamma, float* dbeta,
                       void* stream);
/* out[r][c] += bias[c] on a dense fp32 [rows][D] (bias of a projection-head Linear after vita_sgemm_f32) */
int vita_add_bias_rows(float* out, int64_t rows, int64_t D, const float* bias, void* stream);
/* out[c] = sum_r x[r][c] of a dense fp32 [rows][D], any D (bias gradient of a projection-head Linear) */
int vita_colsum_small_f32(const float* x, int64_t rows, int64_t D, float* out, void* stream);

/* ------------------------------------------------------------------ optimizer
 * torch.optim.Adam step on a flat fp32 buffer (models/reconstruction_models.py:36-38;
 * coupled L2 weight decay for models/MultimodalSimCLR.py:92-99). */
int vita_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                   float beta2, float eps, float weight_decay, float grad_scale, int64_t step,
                   void* stream);

/* ------------------------------------------------------------------ input pipeline
 * Cardiac3DplusTAllAX.__getitem__ (datasets/datasets.py:403-427) on the device, for nb subjects at once:
 *   volume = concat(lax, sax[..., z0:z0+s_sax, :]) on the slice axis            (datasets.py:414)
 *   x=y mirror of load_im_seg_arr (datasets.py:169-171, mirror != 0) + transpose(2,3,0,1)  (datasets.py:415)
 *   image_normalization: (v - min) / (max - min) over the subject's whole volume (utils/general.py:139-147)
 *   frame pick: `[:, ::t_downsampling_ratio]` then the frame_to_keep indices     (datasets.py:419-424),
 *   composed by the host into frames int32 [n_frames] (device memory, every entry in [0, T)).
 * lax fp32 [nb][a0][a1][s_lax][T], sax fp32 [nb][a0][a1][s_sax_total][T] (the npz layout, T fastest);
 * out fp32 [nb][s_lax+s_sax][n_frames][R][C], (R, C) = (a1, a0) when mirrored, else (a0, a1);
 * out[b][s][k][i][j] = (volume[j][i][s][frames[k]] - min_b) / (max_b - min_b)   (mirrored), bit-exact
 * with numpy's fp32 arithmetic.  partial: fp32 scratch of vita_cine_partial_floats(nb) elements;
 * minmax: optional fp32 [nb][2] receiving (min, max).  Augmentations (torchvision) are out of scope. */
int64_t vita_cine_partial_floats(int64_t nb);
int vita_cine_prepare(const float* lax, const float* sax, int64_t nb, int64_t a0, int64_t a1,
                      int64_t s_lax, int64_t s_sax_total, int64_t z0, int64_t s_sax, int64_t T,
                      const int32_t* frames, int64_t n_frames, int32_t mirror, float* partial,
                      float* minmax, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif

/* vpc.h - C ABI of libvpc_hip.so: the MI355X (gfx950) kernels behind the VAE posterior-consistency
 * training step.
 *
 * The reference (stschia/VAE-posterior-consistency) is pure Python and has no FFI / plugin registry; its
 * boundary for this path is the model-class API in src/models/VAE.py.  The Python
 * host package `vae-posterior-consistency_amd` mirrors that API and calls ONLY the entry points below
 * (ctypes).  Each entry point names the reference lines whose arithmetic it replaces.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer unless the comment says "host"; the caller owns all buffers, the
 *    library allocates no device memory and keeps no state between calls except two per-device caches of constants
 *    (CU count, "dynamic-LDS attribute already raised for this kernel"), both mutex / atomic protected;
 *  - `stream` is a hipStream_t passed as void* (NULL = default stream); every launch goes on it;
 *  - return value: 0 ok, 1 bad argument, 2 unsupported shape (d > 128 or L > 15), 3 HIP runtime error.
 *    No exceptions cross the ABI;
 *  - fp32 tensors are row-major and exactly the reference's shapes: x, xhat [B][d]; mean, logvar, eps, z
 *    [B][L]; masks are one byte per element holding 0 or 1 (the kernels convert them with v_cvt_f32_ubyte), [B][d];
 *  - `npass` = 1 (vanilla_VAE) or 2 (Reg_VAE: pass 0 = q, encoded with `mask`; pass 1 = p, encoded with
 *    `mask_p`); per-pass pointers are passed as HOST arrays of `npass` device pointers;
 *  - h1 [B][112] and h2 [B][64] are padded fp32 workspaces (16-byte aligned) that carry the hidden
 *    activations from vpc_encoder_fwd to vpc_encoder_bwd;
 *  - weights are consumed as packed "images" (see csrc/vpc_layout.h) produced by vpc_pack_weights from the
 *    flat fp32 parameter vector in state_dict order
 *        seq_encoder.{0,2,4}.{weight,bias}, seq_decoder.{0,2,4}.{weight,bias}   (VAE.py:366-376);
 *  - weight gradients leave the kernels as per-workgroup partial blocks; vpc_reduce_partials turns them
 *    into the flat gradient in the same order (deterministic summation order).
 */
#ifndef VPC_H
#define VPC_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- host-side layout queries (no GPU needed) --------------------------------------------------- */

/* Sizes (in elements) for model (d, L).  Any out pointer may be NULL.  mask_augm != 0 selects the
 * mask-augmented encoder of Reg_VAE_mask / vanilla_VAE_mask (src/models/VAE.py:510-667, 995-1116): the encoder
 * input is [x*mask | mask] of width 2d (needs 2d <= 128), seq_encoder.0.weight is [100][2d]. */
int vpc_layout_sizes(int d, int L, int mask_augm, int* enc_img_floats, int* dec_img_floats, int* n_enc_params,
                     int* n_params, int* enc_part_floats, int* dec_part_floats, int* loss_terms, int* tile_rows);

/* HOST arrays: pack_idx[n_params] (offset of flat parameter i in the combined image [enc | dec]),
 * grad_idx[n_params] (offset of its gradient inside an encoder (i < n_enc) / decoder partial block),
 * img_template[enc_img_floats + dec_img_floats] (zeros plus the constant ones of the bias chain). */
int vpc_build_indices(int d, int L, int mask_augm, int* pack_idx, int* grad_idx, float* img_template);

/* Compute units of the current device. */
int vpc_num_cus(void);
/* Upper bound of *nblocks_out of any kernel below (= 2 * CUs: the small-batch shape spreads the two passes over
 * separate workgroups): size `partials` / `loss_partials` buffers for this many blocks. */
int vpc_max_partial_blocks(void);

/* ---- precision of the matrix products (argument `precision` of vpc_encoder_fwd / vpc_encoder_bwd /
 * vpc_decoder_fused) ------------------------------------------------------------------------------------------
 *   0  f32     v_mfma_f32_16x16x4_f32 on the fp32 images (exact fp32 FMA chains; the path every parity claim of the
 *              1e-4 target refers to)
 *   1  bf16x3  "split bf16": operands carried as bf16 hi + bf16 lo, hi*hi + hi*lo + lo*hi on v_mfma_f32_16x16x32_bf16
 *              with fp32 accumulation - fp32-class accuracy at 16/3 of the fp32 MFMA rate
 *   2  bf16    plain bf16 inputs, fp32 accumulation (BASELINE configs 2 / 3 name bf16); loss math stays fp32
 * 1 and 2 take the images of vpc_build_indices_bf16 / vpc_pack_weights_bf16 instead of the fp32 images, exist for
 * d % 4 == 0 in the plain (not mask-augmented) encoder, and in vpc_decoder_fused for d in (64, 128]. */
int vpc_layout_sizes_bf16(int d, int L, int mask_augm, int* enc_img_floats, int* dec_img_floats);
/* HOST arrays: pack_idx_bf[n_params], img_template_bf[enc + dec floats of vpc_layout_sizes_bf16] */
int vpc_build_indices_bf16(int d, int L, int mask_augm, int* pack_idx_bf, float* img_template_bf);
/* img_bf <- flat parameters as bf16 hi / lo pairs (the layer-1 bias stays fp32) */
int vpc_pack_weights_bf16(const float* flat_params, const int* pack_idx_bf, float* img_bf, int n, void* stream);

/* ---- parameters --------------------------------------------------------------------------------- */

/* img[pack_idx[i]] = flat_params[i].  Replaces nothing in the reference (nn.Linear keeps [out][in]). */
int vpc_pack_weights(const float* flat_params, const int* pack_idx, float* img, int n, void* stream);

/* grad_out[i] = scale * sum_b partials[b * block_stride + grad_idx[i]], i < n. */
int vpc_reduce_partials(const float* partials, int nblocks, long block_stride, const int* grad_idx,
                        float* grad_out, int n, float scale, void* stream);

/* torch.optim.Adam(lr, betas, eps), no weight decay / amsgrad - src/experiment_main/train.py:21,116.
 * `step` is the 1-based step count; if step_dev != NULL the count is read from device memory instead (word 0
 * of the `state` vpc_reduce_step maintains) so that a captured HIP graph can be replayed.  If pack_idx/img are
 * non-NULL the updated value is also written into the packed image (saves the vpc_pack_weights launch).
 * accum != NULL: accum[0] += loss_in[0] in the same launch (data parallel: the epoch total of the all-reduced step
 * loss, train.py:117, without a separate tiny kernel). */
int vpc_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int n, float lr,
                  float beta1, float beta2, float eps, long step, const long long* step_dev, const int* pack_idx,
                  float* img, const float* loss_in, float* accum, void* stream);

/* ---- encoder: Reg_VAE.encoder / vanilla_VAE.encoder, src/models/VAE.py:387-395, 1155-1163 -------- */

/* For each pass p: h = MLP(x * mask[p]); mean[p], logvar[p] = chunk(h); if z[p]: z = mean + eps[p] *
 * exp(logvar / 2) (eps[p] NULL -> z = mean, the sample=False branch).  eps / z arrays may be NULL.
 * lat_pitch = row pitch in floats of mean / logvar: L (the reference's dense [B][L] tensors) or 16 (padded
 * 16-byte aligned workspaces of the fused path, pad entries written as 0; z must then be NULL). */
int vpc_encoder_fwd(const float* x, const float* enc_img, int npass, const uint8_t* const* mask,
                    const float* const* eps, float* const* h1, float* const* h2, float* const* mean,
                    float* const* logvar, float* const* z, int lat_pitch, int mask_augm, int precision, long B, int d,
                    int L, void* stream);

/* Autograd of the above (src/experiment_main/train.py:115): given d loss / d mean and d loss / d logvar
 * (with the reparameterisation path already folded in) accumulate the encoder weight gradients of all
 * passes into partial blocks [*nblocks_out][enc_part_floats].  x needs no gradient (layer-0 dgrad skipped). */
int vpc_encoder_bwd(const float* x, const float* enc_img, int npass, const uint8_t* const* mask,
                    const float* const* h1, const float* const* h2, const float* const* dmean,
                    const float* const* dlogvar, int lat_pitch, int mask_augm, int precision, float* partials,
                    int* nblocks_out, long B, int d, int L, void* stream);

/* ---- decoder: Reg_VAE.decoder, src/models/VAE.py:397-401 ----------------------------------------- */

/* xhat = sigmoid(MLP(z)) */
int vpc_decoder_fwd(const float* z, const float* dec_img, float* xhat, long B, int d, int L, void* stream);

/* Autograd of the above: (z, dxhat) -> dz [B][L] and decoder partial blocks (forward is recomputed). */
int vpc_decoder_bwd(const float* z, const float* dxhat, const float* dec_img, float* dz, float* partials,
                    int* nblocks_out, long B, int d, int L, void* stream);

/* ---- loss ---------------------------------------------------------------------------------------- */
/* Both loss entry points evaluate (VAE.py:403-467, 469-494, 1171-1208)
 *   loss * B = sum_p [ cA[p] * NLL(A_p, xhat_p) + cE[p] * NLL(A_p & ~B_p, xhat_p) ]
 *              + bq * KL(q || N(0,1)) + bp * KL(p || N(0,1)) + cr * KL(q || p) - wml * log N(z'; mu_p, var_p)
 * with NLL(m, xhat) = sum_ij [ 0.5 log 2pi + m_ij (0.5 x_logvar + (x_ij - xhat_ij)^2 / (2 exp(x_logvar))) ]
 * and z' = mean_q + eps_ml * exp(logvar_q / 2).  maskA/maskB are per-pass byte masks (maskB[p] may be
 * NULL).  Per-workgroup partial sums are written as doubles, loss_partials[nblocks][8]:
 *   0 S_A(pass 0)  1 S_E(pass 0)  2 S_A(pass 1)  3 KL0(q)  4 KL0(p)  5 KL(q||p)  6 loglik  7 S_notA(pass 0)
 * where S_* are the NLL sums WITHOUT the 0.5 log 2pi constants.  Seeds are d(loss)/d(.) times inv_B. */

/* K4: loss on materialised tensors (the model.loss(...) API).  dxhat/dmean/dlogvar NULL -> no seeds. */
int vpc_loss_fwd_bwd(const float* x, int npass, const float* const* xhat, const uint8_t* const* maskA,
                     const uint8_t* const* maskB, const float* cA, const float* cE, const float* const* mean,
                     const float* const* logvar, const float* eps_ml, float bq, float bp, float cr, float wml,
                     float inv_B, float x_logvar, float* const* dxhat, float* const* dmean,
                     float* const* dlogvar, double* loss_partials, int max_blocks, int* nblocks_out, long B, int d,
                     int L, void* stream);

/* Fused training path: reparameterise + decoder forward + loss + backward seeds + decoder backward in one
 * pass (nothing of size B x d is written).  Outputs the TOTAL seeds on the encoder outputs
 * (dmean[p], dlogvar[p], reparameterisation path included), decoder partial blocks and loss partials
 * (term 7 unused).  mean / logvar / eps / eps_ml / dmean / dlogvar are padded [B][16] workspaces (lat_pitch must
 * be 16; pad entries of mean / logvar must be 0 as vpc_encoder_fwd writes them, pad entries of eps are ignored).  Replaces VAE.py:389-392 (rsample), 397-401, 403-467 and their autograd. */
int vpc_decoder_fused(const float* x, const float* dec_img, int npass, const uint8_t* const* maskA,
                      const uint8_t* const* maskB, const float* cA, const float* cE, const float* const* mean,
                      const float* const* logvar, const float* const* eps, const float* eps_ml, float bq, float bp,
                      float cr, float wml, float inv_B, float x_logvar, float* const* dmean,
                      float* const* dlogvar, int lat_pitch, int precision, float* partials, double* loss_partials,
                      int* nblocks_out, long B, int d, int L, void* stream);

/* out9[0] = loss / B_global with the NLL constants of the B_local rows this rank processed (so that the
 * sum over data-parallel ranks is the loss of the concatenated batch), out9[1..8] = the 8 raw sums;
 * if accum != NULL, accum[0] += out9[0] (device-side epoch total, train.py:117 without a host sync). */
int vpc_loss_finalize(const double* loss_partials, int nblocks, float cA0, float cE0, float cA1, float bq, float bp,
                      float cr, float wml, long B_local, long B_global, int d, float* out9, float* accum,
                      void* stream);

/* vpc_reduce_partials (encoder) + vpc_reduce_partials (decoder) + vpc_loss_finalize in ONE launch: the whole
 * post-backward reduction of the fused step.  grad_out[0, n_enc) from the encoder blocks, [n_enc, n) from the
 * decoder blocks (grad_idx as returned by vpc_build_indices), out9 / accum as vpc_loss_finalize.  If state != NULL
 * (two int64 words on the device) the kernel also does state[0] += 1 (optimiser step count) and
 * state[1] += rng_inc (Philox counter offset): the per-step counters of a replayed HIP graph.
 * inv_maps (optional): the inverse maps (block position -> parameter) vpc_build_inverse_maps wrote for this
 * grad_idx.  With them - and both strides multiples of 4 floats and the block pointers 16-byte aligned, true for the
 * blocks the kernels above write - the blocks are read in layout order, 16 bytes per lane, instead of 4-byte
 * gathers through grad_idx.  The library allocates nothing and keeps no state: the caller owns the map buffer.
 * Summation order is fixed either way (reproducible), but differs between the two forms in the last bits. */
int vpc_build_inverse_maps(const int* grad_idx, int n_enc, int n, long enc_stride, long dec_stride, int* inv_out,
                           void* stream); /* inv_out: enc_stride + dec_stride ints (device) */
int vpc_reduce_step(const float* enc_partials, int enc_blocks, long enc_stride, const float* dec_partials,
                    int dec_blocks, long dec_stride, const int* grad_idx, const int* inv_maps, float* grad_out,
                    int n_enc, int n,
                    const double* loss_partials, int loss_blocks, float cA0, float cE0, float cA1, float bq, float bp,
                    float cr, float wml, long B_local, long B_global, int d, float* out9, float* accum,
                    long long* state, long long rng_inc, void* stream);

/* vpc_reduce_step + vpc_adam_step in ONE launch (single process, eager): Adam is elementwise, so the thread that
 * finishes gradient i updates parameter i (and its packed-image entry) on the spot.  Same result as the two
 * calls in sequence; `step` >= 1 is the optimiser step count of THIS update. */
int vpc_reduce_step_adam(const float* enc_partials, int enc_blocks, long enc_stride, const float* dec_partials,
                         int dec_blocks, long dec_stride, const int* grad_idx, const int* inv_maps, float* grad_out,
                         int n_enc, int n,
                         const double* loss_partials, int loss_blocks, float cA0, float cE0, float cA1, float bq,
                         float bp, float cr, float wml, long B_local, long B_global, int d, float* out9, float* accum,
                         float* params, float* exp_avg, float* exp_avg_sq, float lr, float beta1, float beta2,
                         float eps, long step, const int* pack_idx, float* img, void* stream);
/* the same, re-packing the compact bf16 image of the whole-step kernel instead of the fp32 images: pack_idx_c / img_c of
 * vpc_step_build_indices_bf16 (a plain-bf16 step then needs no vpc_step_pack_weights_bf16 launch) */
int vpc_reduce_step_adam_bf16c(const float* enc_partials, int enc_blocks, long enc_stride, const float* dec_partials,
                         int dec_blocks, long dec_stride, const int* grad_idx, const int* inv_maps, float* grad_out,
                         int n_enc, int n,
                         const double* loss_partials, int loss_blocks, float cA0, float cE0, float cA1, float bq,
                         float bp, float cr, float wml, long B_local, long B_global, int d, float* out9, float* accum,
                         float* params, float* exp_avg, float* exp_avg_sq, float lr, float beta1, float beta2,
                         float eps, long step, const int* pack_idx_c, float* img_c, void* stream);

/* ---- random draws (Philox4x32-10, counter = GLOBAL element-group index + offset) -------------------
 * Data parallel: a rank that holds rows [row_lo, row_lo + B_local) of a global batch passes
 * where its shard starts, and every element gets the counter it would get in the single-process run on the
 * concatenated batch - the drawn mask_p / eps of a row do not depend on the world size. */

/* mask_out = mask_in AND (U < keep_prob): create_missing_uci(shape, rate) * mask with keep_prob = 1 - rate/100
 * (src/utils/utils.py:36-39, src/experiment_main/train.py:53-55).  mask_in NULL = all ones.  U has 16 random
 * bits (one Philox call serves 8 elements; element i uses counter (elem_lo + i) / 8 + offset): keep_prob is resolved
 * to 2^-16.  elem_lo = index of mask_out[0] in the global array (row_lo * d; 0 for a single process). */
int vpc_draw_mask(const uint8_t* mask_in, uint8_t* mask_out, long n, float keep_prob, unsigned long long seed,
                  unsigned long long offset, long elem_lo, void* stream);

/* vpc_draw_mask + vpc_fill_normal in one launch (the two per-step draws of the fused step).  If state != NULL,
 * state[1] (device) is added to both offsets.  mask_elem_lo as vpc_draw_mask's elem_lo; eps_* as vpc_fill_normal's
 * row-shard arguments. */
int vpc_draw_step(const uint8_t* mask_in, uint8_t* mask_out, long n_mask, float keep_prob, float* eps_out, long n_eps,
                  unsigned long long seed, unsigned long long offset_mask, unsigned long long offset_eps,
                  const long long* state, long mask_elem_lo, long eps_rows_local, long eps_rows_global,
                  long eps_row_lo, int eps_pitch, void* stream);

/* out ~ N(0,1): the eps of Normal.rsample() (VAE.py:389-392); one Philox call per 4 floats.  state (optional,
 * device): state[1] is added to the offset (the per-step counter of a replayed HIP graph, as vpc_draw_step).
 * rows_local == 0: flat array, group g uses counter g + offset.  rows_local > 0: out is [planes][rows_local][pitch]
 * (pitch % 4 == 0), the rows [row_lo, row_lo + rows_local) of a global [planes][rows_global][pitch] array, and a
 * group uses the counter of its position in the global array. */
int vpc_fill_normal(float* out, long n, unsigned long long seed, unsigned long long offset, const long long* state,
                    long rows_local, long rows_global, long row_lo, int pitch, void* stream);

/* ---- data parallel: the step's ONE collective on the caller's stream (RCCL over xGMI) -------
 * The reference has no distributed code; these entry points exist so that the flat bucket [gradients | loss terms]
 * (37 548 + 9 floats) is all-reduced as an ordinary node of the compute stream - no hop to a communication stream,
 * capturable in the same HIP graph as vpc_reduce_step and vpc_adam_step.  librccl is opened lazily (dlopen).
 *   vpc_rccl_unique_id   rank 0: 128-byte ncclUniqueId into HOST memory; the caller ships it to the other ranks
 *                        (the Python host uses the torch.distributed process group it already has for that)
 *   vpc_rccl_comm_init   every rank, collectively: ncclCommInitRank on the CURRENT device -> opaque communicator
 *   vpc_allreduce_flat   in-place ncclAllReduce(sum, fp32) of bucket[0, count) on `stream`
 *   vpc_rccl_comm_destroy */
int vpc_rccl_unique_id(void* id128_host);
int vpc_rccl_comm_init(const void* id128_host, int nranks, int rank, void** comm_out);
int vpc_allreduce_flat(void* comm, float* bucket, long count, void* stream);
int vpc_rccl_comm_destroy(void* comm);

/* ---- active variable selection reward (config 5) -------------------------------------------------------
 * Replaces the candidate loop of active_learning_func (src/experiment_main/evaluate.py:424-433) and the
 * functions it calls, R_lindley_chain / chaini_I / chaini_II (evaluate.py:514-634): for every row n and every
 * feature u < d-1 with mask[n][u] == 0
 *     R[n][u] = 1/M sum_m [ KL_I(n,u,m) - KL_II(n,u,m) ]          (-1e4 where mask[n][u] != 0)
 * with the reference's KL expression (first term divided by the std, evaluate.py:582-583) and its carry-over of
 * the imputed target between MC samples (evaluate.py:531-536).  x [n][d], mask [n][d] bytes, im [M][n][d]
 * (MC imputations, evaluate.py:396-414), W1 [100][d] / b1 [100] = seq_encoder.0 (flat parameter views),
 * enc_img = packed encoder image; pre / stat / w1t are scratch buffers of the sizes vpc_reward_scratch returns
 * (16-byte aligned; w1t also holds the rows' candidate lists).  The target is the last column, as in the reference. */
int vpc_reward_scratch(int n, int d, int M, long* pre_floats, long* stat_floats, long* w1t_floats);
int vpc_reward_matrix(const float* x, const uint8_t* mask, const float* im, const float* W1, const float* b1,
                      const float* enc_img, float* pre, float* stat, float* w1t, float* R, int n, int d, int L, int M,
                      void* stream);

/* ---- MNAR path (config 3): REG_notMIWAE_v2 / notMIWAE_myversion --------------------------------------
 * Reference: src/models/VAE.py:2327-2505 and :2691-2847 (encoder d->128->128 ELU + heads, K-fold replicated
 * draw, decoder L->128->128 ELU + Sigmoid / Hardtanh(-10,0) heads, self-masking missingness model).  The
 * layers are generic fp32 MFMA GEMMs (any M, N, K), the loss is one fused forward+backward kernel.
 * Activation codes: 0 none, 1 ELU, 2 Sigmoid for output features < split and Hardtanh(-10,0) from split on,
 * 3 ReLU.  All matrices row-major with an explicit row pitch (ld*, in floats).  `precision` as above (0 f32, 1 bf16x3,
 * 2 bf16): the operands stay fp32 in memory and in LDS and are converted in registers - no separate weight images. */

/* y[M][N] = act(x[M][K] w[N][K]^T + bias[N])            nn.Linear + activation (VAE.py:2343-2363) */
int vpc_linear_fwd(const float* x, long ldx, const float* w, const float* bias, float* y, long ldy, long M, int N,
                   int K, int act, int act_split, int precision, void* stream);

/* dx[M][K] = ((dy * act'(y_gate)) w[N][K]) * act_prev'(x_out)     data gradient of the same layer.
 * y_gate (the layer's outputs, activation code `gate`) may be NULL when dy is already the pre-activation
 * gradient; x_out (the layer's inputs = previous layer's outputs, activation `act_prev`) may be NULL. */
int vpc_linear_dgrad(const float* dy, long lddy, const float* y_gate, long ldyg, int gate, int gate_split,
                     const float* w, const float* x_out, long ldx, int act_prev, float* dx, long lddx, long M, int N,
                     int K, int precision, void* stream);

/* dw[N][K] (+)= (dy * act'(y_gate))^T x,  db[N] (+)= column sums; split over M into per-workgroup partials in
 * `scratch` (vpc_linear_wgrad_scratch floats) that are summed in a fixed order.  db may be NULL.
 * dw == NULL: only the partials are written (one launch); vpc_linear_wgrad_reduce then sums the partials of up to 8 such
 * calls - each with its own scratch buffer, HOST arrays of n_layers entries - in ONE launch, with the same summation order
 * (the fused MNAR step: 6 weight gradients, 7 launches instead of 12). */
long vpc_linear_wgrad_scratch(long M, int N, int K);
int vpc_linear_wgrad(const float* dy, long lddy, const float* y_gate, long ldyg, int gate, int gate_split,
                     const float* x, long ldx, float* dw, float* db, float* scratch, long scratch_floats, long M, int N,
                     int K, int accumulate, int precision, void* stream);
int vpc_linear_wgrad_reduce(int n_layers, const float* const* scratch, const long* M, const int* N, const int* K,
                            float* const* dw, float* const* db, const int* accumulate, void* stream);

/* z[b*K+k][:] = mean[b] + eps[b][k] * exp(logvar[b]/2), heads = [mean L | logvar L]; eps NULL -> z = mean
 * (encoder, VAE.py:2382-2391 / :2753-2765) and its backward (sum over the K replicas, plus g_heads if given). */
int vpc_nm_sample(const float* heads, long ldh, const float* eps, float* z, long ldz, long B, int K, int L,
                  void* stream);
int vpc_nm_sample_bwd(const float* dz, long lddz, const float* eps, const float* heads, long ldh, const float* g_heads,
                      long ldg, float* out, long ldo, long B, int K, int L, void* stream);
/* out = x * mask (float mask; encoder input VAE.py:2379 / :2750) */
int vpc_nm_mul(const float* x, const float* mask, float* out, long n, void* stream);

/* Loss of REG_notMIWAE_v2 (mask_p != NULL; VAE.py:2398-2471) or notMIWAE_myversion (mask_p == NULL, eps_kl =
 * the fresh draw of VAE.py:2791-2798; :2774-2823) and, when g_xm_q != NULL, every gradient: with respect to the
 * decoder heads (x_mean | x_logvar, rows b*K+k, pitch ldg_*), the encoder heads ([B][2L]), and W / b.
 * Masks are float 0/1.  out8 (device doubles) = loss, loss_q, loss_p, KL_reg, NLL_E, mean RE_q, sum lse_q,
 * sum lse_p; means run over B_global rows (data parallel: sum out8[0] over ranks).  xm_imp != NULL also
 * writes the self-normalised imputation sum_k softmax(-l_w)_k x_mean[b][k] (llh_eval branch :2458-2461).
 * loss_f32 / accum (optional, device): loss as a float, and accum[0] += loss (train.py:117 without a host sync).
 * state (optional, two int64 on the device): state[0] += 1 (optimiser step count, read by vpc_adam_step's step_dev)
 * and state[1] += rng_inc (Philox counter offset read by vpc_nm_prep) - the per-step counters of a replayed graph.
 * gated != 0: g_xm / g_xl are gradients w.r.t. the head PRE-activations (already through Sigmoid' / Hardtanh'), so
 * vpc_linear_dgrad / vpc_linear_wgrad take them with y_gate = NULL (the fused step; autograd callers pass 0). */
int vpc_nm_loss_blocks(long B);
long vpc_nm_loss_scratch(long B, int d);
int vpc_nm_loss(const float* x, const float* mask, const float* mask_p, const float* xm_q, const float* xl_q, long ld_q,
                const float* xm_p, const float* xl_p, long ld_p, const float* heads_q, const float* heads_p, long ldh,
                const float* W, const float* b, const float* eps_kl, float* g_xm_q, float* g_xl_q, long ldg_q,
                float* g_xm_p, float* g_xl_p, long ldg_p, float* g_heads_q, float* g_heads_p, long ldgh, float* gW,
                float* gb, int accumulate_wb, float* xm_imp, void* scratch, long scratch_bytes, double* out8,
                float* loss_f32, float* accum, long long* state, long long rng_inc, int gated, long B, long B_global,
                int K, int d, int L, double alpha, void* stream);

/* Per-step input preparation of the fused MNAR step in one launch: mask_p = mask * (U < keep_prob) (float masks,
 * create_missing_uci * mask, train.py:53-55; Philox counter = element index / 4 + offset), xin[0:B] = x * mask and,
 * when mask_p_out != NULL, xin[B:2B] = x * mask_p (the two encoder passes stacked); when n_eps > 0 also
 * eps_out[0:n_eps] ~ N(0,1) (counter offset_eps).  If state != NULL, state[1] (device) is added to both offsets:
 * the per-step counter of a replayed HIP graph (bumped by vpc_nm_loss).  elem_lo (multiple of 4) = index of x[0] in
 * the global [B_global][d] array and eps_* = row-shard description of eps_out as in vpc_fill_normal (rows = data rows,
 * pitch = K * L): the draws of a row do not depend on the data-parallel sharding. */
int vpc_nm_prep(const float* x, const float* mask, float* mask_p_out, float* xin, long B, int d, float keep_prob,
                float* eps_out, long n_eps, unsigned long long seed, unsigned long long offset,
                unsigned long long offset_eps, const long long* state, long elem_lo, long eps_rows_local,
                long eps_rows_global, long eps_row_lo, int eps_pitch, void* stream);

/* ---- layer-fused K-fold decoder of the regularised MNAR step, plain bf16 MFMA inputs (csrc/vpc_nmdec.hip) ----------
 * Replaces, for REG_notMIWAE_v2 and notMIWAE_myversion at obs_dim = 128 (latent_dim <= 15, 8 <= K <= 64; hidden width 128), the launches
 *   vpc_nm_sample -> 3 x vpc_linear_fwd -> vpc_nm_loss -> 3 x (vpc_linear_wgrad, vpc_linear_dgrad) -> vpc_nm_sample_bwd
 * of one training step with precision = 2 (src/models/VAE.py:2382-2396 / :2753-2772 K-fold rsample + decoder, :2398-2471 /
 * :2774-2823 loss, and
 * their autograd, src/experiment_main/train.py:115): the K replicas of a few data rows are one workgroup tile, and no
 * array of B * K rows crosses HBM.  Same mathematics, gradient weights and bf16 rounding points as that chain, except
 * that ELU' is taken from the bf16-rounded activation and the bias gradients are column sums of the bf16-rounded dY
 * (tests/test_nmdec.py; oracle/notmiwae_oracle.py mirrors both).
 *   vpc_nmdec_applicable     1 when vpc_nmdec_step covers the shape (VPC_NMDEC=0 in the environment: never)
 *   vpc_nmdec_layout         floats of the weight image, floats of one partial block, most workgroups of a launch
 *   vpc_nmdec_build_indices  HOST tables over the model's flat parameter buffer (n entries, order
 *                            [W b | We1 be1 We2 be2 Wmu Wls bmu bls | Wd1 bd1 Wd2 bd2 Wxm Wxl bxm bxl]): pack_idx as
 *                            vpc_step_pack_weights_bf16 reads it (decoder + missingness model, then the encoder), grad_idx =
 *                            position of the parameter's gradient inside a partial block (-1: not produced here)
 *   vpc_nmdec_step           mask_p != NULL (REG_notMIWAE_v2): heads [2 B][ldh] = encoder (mean | logvar) of the q rows, then
 *                            of the p rows; eps [2 B K][L] likewise.  mask_p == NULL (notMIWAE_myversion): heads [B][ldh],
 *                            eps [2][B K][L] = the decoder's draws, then the draws of its Monte-Carlo KL; alpha is ignored.
 *                            dht [2 B][2 L] receives d loss / d heads (sum over K of dz + the analytic KL gradients);
 *                            grad (flat, n entries) receives the entries grad_idx names (fixed-order sum of the blocks; inv_idx,
 *                            optional device table [part_floats]: parameter index of a block position or -1 - the blocks are
 *                            then read in layout order, 16 bytes per lane);
 *                            out8 / loss_f32 / accum / state / rng_inc / B_global / alpha as vpc_nm_loss.
 *                            part: max_blocks x part_floats floats, stat_part: max_blocks x 5 doubles (caller-owned). */
int vpc_nmdec_applicable(long B, int K, int d, int L);
int vpc_nmdec_layout(long B, int K, int d, int L, int* img_floats, long* part_floats, int* max_blocks);
int vpc_nmdec_build_indices(int d, int L, int hid, int* pack_idx, int* grad_idx, int n);
/* the encoder forward of the same step (seq_encoder + q_mu | q_logstd, VAE.py:2378-2384 / :2749-2756) as ONE launch instead of three
 * vpc_linear_fwd: xin [R][128] = x * mask of the stacked passes -> h1, h2 [R][128] (fp32, ELU applied: the backward GEMMs read them),
 * heads [R][2 L]; img = the image buffer vpc_nmdec_layout sizes and vpc_nmdec_build_indices' pack_idx fills (the encoder's weights sit
 * behind the decoder's).  Rounding points of vpc_linear_fwd with precision 2. */
int vpc_nmenc_fwd(const float* img, const float* xin, float* h1, float* h2, float* heads, long R, int d, int L, void* stream);
/* the encoder BACKWARD of the same step (autograd of VAE.py:2378-2384 behind d loss / d heads) as one launch + the fixed-order
 * reduction of its partial blocks, instead of three vpc_linear_wgrad, two vpc_linear_dgrad and vpc_linear_wgrad_reduce:
 * dht [R][2 L] (vpc_nmdec_step's), h1 / h2 / xin [R][128] as vpc_nmenc_fwd saw / stored them -> the gradients of We1 be1 We2 be2
 * [Wmu ; Wls] [bmu ; bls] inside grad (flat, layout of vpc_nmdec_build_indices) at the entries inv_idx names.  Rounding points of the
 * precision-2 GEMMs (bf16 dY, X, W operands; ELU' from the fp32 h), except that the bias gradients sum the bf16-rounded dY.
 * part: scratch of part_floats floats, 16-byte aligned - vpc_nmdec_step's partial blocks serve once that call is enqueued.
 * vpc_nmenc_build_indices: inv_idx [*part_floats ints] = flat parameter index of a block position or -1 (NULL: size query only). */
int vpc_nmenc_bwd(const float* img, const float* xin, const float* h1, const float* h2, const float* dht, float* part,
                  long part_floats, const int* inv_idx, float* grad, long R, int d, int L, void* stream);
int vpc_nmenc_build_indices(int d, int L, int hid, int* inv_idx, long* part_floats, int n);
/* everything behind vpc_nmenc_fwd of a SINGLE-DEVICE step (train.py:87-101, 116: loss.backward(); optimizer.step()) in three launches:
 * vpc_nmdec_step's tile kernel, vpc_nmenc_bwd's tile kernel and one tail launch that sums both sets of partial blocks into grad
 * (fixed order), applies torch.optim.Adam (as vpc_adam_step, host step count) to every gradient it finishes, re-packs that
 * parameter's place in the bf16 image (pack_idx of vpc_nmdec_build_indices; img is read by the tile kernels and updated by the tail)
 * and writes the loss terms.  Arguments as those two calls; inv_idx (decoder blocks) is required; part_e: its own scratch of
 * part_e_floats floats (max_blocks x vpc_nmenc_build_indices' part_floats suffices) - both sets are live until the tail. */
int vpc_nm_fused_bwd_step(float* img, const float* x, const float* mask, const float* mask_p, const float* xin, const float* h1,
                          const float* h2, const float* heads, long ldh, const float* eps, float* dht, float* part,
                          double* stat_part, float* part_e, long part_e_floats, const int* inv_idx, const int* inv_idx_e,
                          float* grad, int n, double* out8, float* loss_f32, float* accum, long B, long B_global, int K, int d,
                          int L, double alpha, float* params, float* exp_avg, float* exp_avg_sq, float lr, float beta1,
                          float beta2, float eps_adam, long step, const int* pack_idx, void* stream);
int vpc_nmdec_step(const float* img, const float* x, const float* mask, const float* mask_p, const float* heads, long ldh,
                   const float* eps, float* dht, float* part, double* stat_part, const int* grad_idx, const int* inv_idx,
                   float* grad, int n, double* out8, float* loss_f32, float* accum, long long* state, long long rng_inc, long B, long B_global,
                   int K, int d, int L, double alpha, void* stream);

/* ---- PNP / EDDI encoder front-end (Reg_EDDI / vanilla_EDDI, src/models/VAE.py:719-733, 903-917) ----------
 * agg[b] = sum_j mask[b][j] relu(W [x_bj, x_bj E_j, t_j] + c), W = pnp_encoder1.0.weight [K][2+K], c its bias,
 * E = type_pars1 [d][K], t = type_bias1 [d].  The layer folds per feature into pre = x_bj A_j + C_j
 * (vpc_eddi_fold writes AC = [A | C], each [K][d]); nothing of size B*d*(2+K) is materialised.  d <= 128, K <= 32.
 * mask2 != NULL stacks a second pass over the same x (the mask / mask_p passes of one training step): agg and dagg
 * then have 2B rows ([pass][B][K]) and the gradients are those of both passes.
 * vpc_eddi_front_bwd: gradients of (E, t, W, c) from dagg; scratch of vpc_eddi_front_scratch(rows, d, K) floats with
 * rows = B or 2B. */
int vpc_eddi_fold(const float* E, const float* tb, const float* Wp, const float* cp, float* AC, int d, int K,
                  void* stream);
int vpc_eddi_front_fwd(const float* x, const uint8_t* mask, const uint8_t* mask2, const float* AC, float* agg, long B,
                       int d, int K, void* stream);
long vpc_eddi_front_scratch(long rows, int d, int K);
int vpc_eddi_front_bwd(const float* x, const uint8_t* mask, const uint8_t* mask2, const float* AC, const float* dagg,
                       const float* E, const float* tb, const float* Wp, float* scratch, long scratch_floats, float* gE,
                       float* gtb, float* gWp, float* gcp, int accumulate, long B, int d, int K, void* stream);

/* ---- whole-step kernel, plain bf16 (csrc/vpc_step.hip): vpc_encoder_fwd + vpc_decoder_fused + vpc_encoder_bwd with
 * precision 2 as ONE launch per 128-row tile - the step body of src/experiment_main/train.py:87-115 for Reg_VAE /
 * vanilla_VAE (src/models/VAE.py:496-507 forward, :403-467 loss, autograd backward) with nothing but x, the masks and eps
 * read from HBM and nothing but the gradient partial blocks and loss terms written.  obs_dim in (64, 128], obs_dim % 4 == 0,
 * plain (not mask-augmented) encoder, any batch (the grid is min(tiles, CUs) workgroups looping over their tiles).
 * Takes its own weight image (all six layers, bf16, 98.5 KB): vpc_step_layout_bf16 gives its size in floats and the
 * kernel's dynamic-LDS bytes; vpc_step_build_indices_bf16 fills HOST arrays pack_idx_c[n_params] / img_template_c[floats];
 * vpc_step_pack_weights_bf16 writes img_c from the flat parameters (after every optimiser step): entry i of pack_idx_c >= 0 is
 * the u16 position of parameter i inside the image, < 0 the dword -(idx + 1) of a value that stays fp32, INT_MIN = skip.
 * vpc_step_fused_bf16: arguments as vpc_decoder_fused (mask[p] is the encoder mask AND the first loss mask of pass p;
 * eps[p] [B][16] padded rows, eps_ml likewise); partE / partD / loss_partials are written in the layouts of
 * vpc_encoder_bwd / vpc_decoder_fused (*nblocks_out blocks each), so vpc_reduce_step(_adam) consumes them unchanged.
 * The kernel runs two sweeps over a workgroup's tiles (decoder-side gradients, then encoder-side gradients: csrc/vpc_step.hip);
 * the seeds on (mean | logvar) cross from one to the other through `workspace` as packed bf16, 16 bytes per lane. */
/* 1 when the library runs a (B, d, L, npass) plain-bf16 step through vpc_step_fused_bf16: obs_dim in (64, 128], obs_dim % 4 == 0, any
 * batch (VPC_TILE=64 / VPC_STEP_FUSED=0 in the environment: the three small-shape kernels, for A/B runs), else 0 */
int vpc_step_fused_applicable(long B, int d, int L, int npass);
/* floats of the caller-owned `workspace` of vpc_step_fused_bf16 for B rows (16-byte aligned; contents are scratch) */
long vpc_step_workspace_floats(long B);
int vpc_step_layout_bf16(int d, int L, int* img_floats, int* lds_bytes);
int vpc_step_build_indices_bf16(int d, int L, int* pack_idx_c, float* img_template_c);
int vpc_step_pack_weights_bf16(const float* flat_params, const int* pack_idx_c, float* img_c, int n, void* stream);
int vpc_step_fused_bf16(const float* x, const float* img_c, int npass, const uint8_t* const* mask,
                        const uint8_t* const* maskB, const float* cA, const float* cE, const float* const* eps,
                        const float* eps_ml, float bq, float bp, float cr, float wml, float inv_B, float x_logvar,
                        float* partE, float* partD, double* loss_partials, float* workspace, int* nblocks_out, long B,
                        int d, int L, void* stream);

/* ---- small-batch whole-step kernel, fp32 (csrc/vpc_small.hip): the same step as vpc_encoder_fwd -> vpc_decoder_fused ->
 * vpc_encoder_bwd with precision 0 (src/experiment_main/train.py:87-115; src/models/VAE.py:496-507, 403-467) as ONE launch in
 * which a workgroup owns a 16-row tile and the feature tiles of every layer are split over its 8 waves - for the reference's
 * own batch sizes (64 / 128, Data/imputation_args*.json) and the per-GPU shards of strong scaling, where the row-tiled kernels
 * are three serial single-tile latencies.  Weights are read from the fp32 images of vpc_build_indices / vpc_pack_weights in
 * global memory; partial blocks and loss terms in the layouts of vpc_encoder_bwd / vpc_decoder_fused.  Plain (not
 * mask-augmented) encoder, obs_dim % 4 == 0, B <= 32 x CUs rows (one workgroup per tile); maskB[p] must be NULL or mask[1 - p].
 * vpc_step_small_max_rows: batches up to this many rows take this path inside FusedTrainer (0 = off; env VPC_STEP_SMALL). */
long vpc_step_small_max_rows(void);
int vpc_step_small_f32(const float* x, const float* enc_img, const float* dec_img, int npass, const uint8_t* const* mask,
                       const uint8_t* const* maskB, const float* cA, const float* cE, const float* const* eps,
                       const float* eps_ml, float bq, float bp, float cr, float wml, float inv_B, float x_logvar, float* partE,
                       float* partD, double* loss_partials, int* nblocks_out, long B, int d, int L, void* stream);
/* vpc_draw_step + vpc_step_small_f32 in ONE launch: every workgroup draws the mask_p bytes (mask[1] = mask_in & keep; mask_in NULL:
 * none - vanilla_VAE) and the normals of ITS 16 rows (eps_out = eps[0], planes [B][16]: eps[1] and eps_ml follow it, n_eps floats
 * in all) with the Philox counters vpc_draw_step would use - seed, offsets, state, mask_elem_lo and the eps_* shard description
 * (eps_pitch 16, eps_rows_local = B) as there -, stores them where the step reads them, and runs the step: the same draws
 * (train.py:53-55, VAE.py:389-392), one launch less at the batch sizes where a launch is a fifth of the step. */
int vpc_step_small_draw_f32(const float* x, const float* enc_img, const float* dec_img, int npass, const uint8_t* const* mask,
                            const uint8_t* const* maskB, const float* cA, const float* cE, const float* const* eps,
                            const float* eps_ml, float bq, float bp, float cr, float wml, float inv_B, float x_logvar,
                            float* partE, float* partD, double* loss_partials, int* nblocks_out, long B, int d, int L,
                            const uint8_t* mask_in, float keep_prob, float* eps_out, long n_eps, unsigned long long seed,
                            unsigned long long offset_mask, unsigned long long offset_eps, const long long* state,
                            long mask_elem_lo, long eps_rows_local, long eps_rows_global, long eps_row_lo, int eps_pitch,
                            void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VPC_H */

/*
 * drs.h -- C ABI of libdrs_hip.so: the MI355X (gfx950) kernels of the dilated-CNN multi-size patch
 * training / sliding-window inference path of keillernogueira/dynamic-rs-segmentation.
 *
 * The reference has no FFI of its own: its device boundary is three `sess.run` call shapes
 * (isprs_dilated_random.py:1750-1752 train, :1274-1275 infer, :1588 validate) behind which TensorFlow runs
 * the ops listed below.  Each entry point here replaces one of those TensorFlow ops (or one per-pixel numpy
 * loop next to them) and cites the reference line it stands in for.  The host-side mirror of the reference's
 * net builders and step loops (dynamic-rs-segmentation_amd/net.py, loops.py) binds exactly these symbols.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (any allocator; PyTorch-ROCm tensors in the host
 *     mirror); nothing is allocated, freed or synchronised inside the library;
 *   - `stream` is a hipStream_t (NULL = the default stream); calls only enqueue work;
 *   - every function returns 0 on success, DRS_ERR_ARG (1) for a rejected argument, DRS_ERR_HIP (2) for a
 *     launch failure; nothing throws across the boundary; not re-entrant on one set of buffers;
 *   - activations are channels-last (the reference's NHWC) with an explicit zero halo:
 *     a "view" (base, S, P, ld, coff) addresses a slab [B][S+2P][S+2P][ld] floats whose interior pixel
 *     (b, y, x), channel c of the slice lives at base[((b*(S+2P) + y+P)*(S+2P) + x+P)*ld + coff + c];
 *     P = 0 gives the reference's plain [B, S, S, C] tensor; filters are HWIO = [k][k][Cin][Cout] as in the
 *     reference (isprs:706);
 *   - B*S*S must be < 2^24, and every activation slab a convolution reads must stay below 4 GiB (2^30 floats): the kernels
 *     address with a wave-uniform 64-bit base plus 32-bit byte offsets.
 */
#ifndef DRS_H_
#define DRS_H_
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DRS_OK 0
#define DRS_ERR_ARG 1
#define DRS_ERR_HIP 2

/* ---- tf.nn.atrous_conv2d / tf.nn.conv2d (SAME, stride 1) + tf.nn.bias_add  (isprs:710-713) -------------
 * out[p, coff_out + o] (=|+=) sum_{u,v,c} in[p + (u,v)*rate - pad_before, c] * w[u][v][c][o] + bias[o]
 * `in` is a haloed view with P >= max(pad_before, pad_after); cout a multiple of 32; cin a multiple of 32, or 8 / 16 for
 * conv1 (its 3..5 bands zero-padded by drs_crop_normalize, the filter by drs_filter_pad_cin -- see there for its size).
 * `out` is [B*S*S][ld_out].
 * stats_partial (or NULL; not together with accumulate): [ceil(B*S*S / drs_conv_mtile(cout))][cout][2], per M tile and
 * output channel (s, M2) = (sum of the tile's outputs, sum of their squared deviations from the TILE's mean): the first half
 * of train-mode batch norm (isprs:658-660), two-pass inside the tile as TensorFlow is two-pass over the batch;
 * drs_conv_stats_reduce combines the tiles.
 * The input-gradient pass is this same call on the haloed output gradient with the filter from
 * drs_filter_flip_transpose and pad_before := pad_after. */
int drs_conv_mtile(int cout);
/* drs_conv_forward_ws: the same with a caller-owned workspace (>= drs_conv_workspace_floats(cout) floats, 16-byte aligned, enables
 * every geometry; NULL, unaligned or too small for the cut the shape asks for: the plain launch of drs_conv_forward -- the cut, and with
 * it the association of the sums, follows from the shape and the device's CU count alone, never from the workspace size).  With it, launches of fewer than 4096 tiles run "stream-K": the K-steps of all
 * tiles are cut into equal ranges, one per workgroup and a whole number of workgroups per CU, the tiles that a cut crosses are
 * completed from partial sums in the workspace in a fixed order (bitwise reproducible; sums associate differently from
 * drs_conv_forward's).  This is the path of the per-rank batches of data parallelism (16 patches of 25..85 pixels a side). */
size_t drs_conv_workspace_floats(int cout);
/* 1 if drs_conv_forward_ws (full workspace) leaves out, for this shape, the filter-tap rows that meet only the zero halo for a whole
 * tile of 128 pixels (exact zeros: results do not change): plain launches of >= 4096 tiles, and any plain launch whose tiles are whole
 * image rows of whole patches (S = 32, 64, 128: those start their full tiles first, the skipping ones last); never a stream-K launch */
int drs_conv_halo_skip(int B, int S, int k, int rate, int pad_before, int cin, int cout);
int drs_conv_forward_ws(const float* in, int B, int S, int P, int ld_in, int coff_in, const float* w, const float* bias,
                        int k, int rate, int pad_before, int cin, int cout, float* out, int ld_out, int coff_out,
                        int accumulate, float* stats_partial, float* workspace, size_t workspace_floats, void* stream);
int drs_conv_forward(const float* in, int B, int S, int P, int ld_in, int coff_in, const float* w, const float* bias,
                     int k, int rate, int pad_before, int cin, int cout, float* out, int ld_out, int coff_out,
                     int accumulate, float* stats_partial, void* stream);

/* ---- filter gradient of the same op (what tf.gradients emits for isprs:710-712) --------------------------
 * grad[u][v][c < cin_real][o] = sum_p x[p + (u,v)*rate - pad_before, c] * g[p, o];  x, g haloed views.
 * slab: workspace of drs_conv_wgrad_splits(...) * k*k*cin*cout floats (per-split partial sums, summed in
 * a fixed order -> bitwise reproducible). */
int drs_conv_wgrad_splits(int B, int S, int k, int cin, int cout);
int drs_conv_wgrad(const float* x, int B, int S, int Px, int ld_x, int coff_x, const float* g, int Pg, int ld_g,
                   int coff_g, int k, int rate, int pad_before, int cin, int cin_real, int cout, float* slab,
                   float* grad, void* stream);

/* wt[k-1-u][k-1-v][o][c] = w[u][v][c][o]: the filter of the input-gradient pass */
int drs_filter_flip_transpose(const float* w, float* wt, int k, int cin, int cout, void* stream);
/* wp[u][v][c < cin_pad][o] = c < cin ? w[u][v][c][o] : 0.  For cin_pad = 8 / 16 (conv1's 3..5 bands in an 8-channel slab: the
 * packed-tap form of drs_conv_forward, 32 / cin_pad taps to a K-step of 32 rows) the caller's `wp` holds
 * round_up(k*k*cin_pad, 32) * cout floats and the rows past the last tap must be ZERO (this call writes the first k*k*cin_pad rows). */
int drs_filter_pad_cin(const float* w, float* wp, int k, int cin, int cin_pad, int cout, void* stream);

/* ---- the same convolution in split-bf16 arithmetic (csrc/conv_split.hip) ---------------------------------------
 * An opt-in second arithmetic for isprs:710-713 and its gradients; the exact-fp32 entry points above stay the default.
 * Every fp32 operand x is carried as `nterms` bf16 terms, term s = round-to-nearest-even bf16 of what the earlier terms
 * left (x = t0 + t1 (+ t2) up to 2^-17 |x| / 2^-25 |x|), and a product is evaluated on the bf16 MFMA pipe as the
 * partial products t_i * u_j with i + j < nterms (3 products for 2 terms, 6 for 3), accumulated in fp32.
 * Term layout: interleaved at 32-channel granularity over the fp32 indexing e of the slab the terms mirror:
 * term s of element e sits at (e & ~31) * nterms + 32 * s + (e & 31), 2 bytes each, so views need ld and coff
 * multiples of 32.  Filter terms are K-contiguous rows interleaved the same way (see drs_filter_split).
 *   drs_split_terms    : n fp32 (n % 32 == 0) -> their terms (a whole slab, halo included).
 *   drs_filter_split   : HWIO filter -> wf = forward operand [cout][k*k*cin_pad] and, if wd != NULL, wd = input-gradient
 *                        operand [cin][k*k*cout] (taps reversed: what drs_filter_flip_transpose is to the fp32 path).
 *   drs_conv_forward_split : drs_conv_forward on terms (`in`, `w`); cout % 64 == 0.  The input-gradient pass is the same
 *                        call on the terms of the haloed output gradient with wd and pad_before := pad_after.
 *                        stats_partial rows hold drs_split_conv_mtile(cout) pixels.
 *   drs_conv_wgrad_split : drs_conv_wgrad on terms; slab = drs_conv_wgrad_split_splits(..., Pg, nterms) * k*k*cin*cout floats
 *                        (an upper bound that is monotone in B and S: a slab sized for (b_max, s_max) serves every smaller call).
 *                        With Pg > 0 pixels past the end read the gradient slab's first halo pixel (zeros). */
int drs_split_conv_mtile(int cout);
int drs_split_terms(const float* src, size_t n, int nterms, unsigned short* terms, void* stream);
int drs_filter_split(const float* w, int k, int cin, int cin_pad, int cout, int nterms, unsigned short* wf,
                     unsigned short* wd, void* stream);
int drs_conv_forward_split(const unsigned short* in, int B, int S, int P, int ld_in, int coff_in, const unsigned short* w,
                           const float* bias, int k, int rate, int pad_before, int cin, int cout, float* out, int ld_out,
                           int coff_out, int accumulate, float* stats_partial, int nterms, void* stream);
int drs_conv_wgrad_split_splits(int B, int S, int k, int cin, int cout, int Pg, int nterms);
int drs_conv_wgrad_split(const unsigned short* x, int B, int S, int Px, int ld_x, int coff_x, const unsigned short* g,
                         int Pg, int ld_g, int coff_g, int k, int rate, int pad_before, int cin, int cin_real, int cout,
                         float* slab, float* grad, int nterms, void* stream);

/* ---- tf.contrib.layers.batch_norm(center=False, scale=False, eps=1e-3, decay=0.999)  (isprs:655-663) -----
 * drs_stats_reduce : partial[nrows][C][2] (fp32) plain column sums -> sums[C][2] (fp64), fixed order (the backward
 *                    pass's (sum g, sum g*xhat) slabs); scratch is unused (kept in the signature).
 * drs_conv_stats_reduce : the statistics slab of drs_conv_forward (M = B*S*S pixels, mtile = drs_conv_mtile(cout) or
 *                    drs_split_conv_mtile(cout)) -> sums[C][2] = (sum z, sum z^2) in fp64 by Chan's combination
 *                    sum z^2 = sum_tiles (M2 + s^2 / n_tile): no cancellation in fp32 whatever |mean| / std is.
 *                    Under data parallelism the caller all-reduces `sums` between this call and drs_bn_finish
 *                    (sync batch norm over the global batch).
 * drs_conv_stats_finish : drs_conv_stats_reduce + drs_bn_finish in ONE launch (single rank); sums may be NULL.
 * drs_bn_finish    : sums, count -> mean_rstd[C][2] = (mean, 1/sqrt(biased var + eps)); if moving_* != NULL,
 *                    moving -= (moving - batch) * (1 - decay) with the Bessel-corrected variance when bessel.
 * drs_bn_eval_coeffs: is_training=False branch: mean_rstd from the moving statistics. */
int drs_colsum_scratch_doubles(int ncols);
int drs_stats_reduce(const float* partial, int nrows, int C, double* sums, double* scratch, void* stream);
/* drs_stats_reduce that also leaves means[C][2] = (float)(sums / count): what drs_bn_backward_apply_means takes (single rank: the
 * sums need no all-reduce between the two); sums may be NULL */
int drs_stats_reduce_means(const float* partial, int nrows, int C, double count, double* sums, float* means, void* stream);
int drs_conv_stats_reduce(const float* partial, int M, int mtile, int C, double* sums, double* scratch, void* stream);
int drs_conv_stats_finish(const float* partial, int M, int mtile, int C, double count, float* mean_rstd, float* moving_mean,
                          float* moving_var, double decay, int bessel, double* sums, void* stream);
int drs_bn_finish(const double* sums, double count, int C, float* mean_rstd, float* moving_mean, float* moving_var,
                  double decay, int bessel, void* stream);
int drs_bn_eval_coeffs(const float* moving_mean, const float* moving_var, int C, float* mean_rstd, void* stream);

/* ---- normalise + tf.nn.relu | tf.maximum(0.1x, x) + tf.nn.max_pool(3x3, stride 1, SAME) ------------------
 * (isprs:715-721, 620-621, 745-746, 1001).  z: [B*S*S][C] raw conv output; alpha = 0 (ReLU) or 0.1;
 * pool: bit 0 = apply the 3x3 max-pool; bit 1 = the halo of `out` is known to be zero already (a previous call on this slab with
 * the same B, S, P_out wrote it and nothing else has written the slab since) and need not be rewritten (pooling path only).
 * out: haloed view (the halo zeros are written here); argmax (pool only, may be NULL): [B*S*S][C] window
 * position 0..8 of the first maximum, which the backward pass routes gradients to (TF MaxPoolGrad). */
int drs_bn_act_pool_forward(const float* z, int B, int S, int C, const float* mean_rstd, float alpha, int pool,
                            float* out, int P_out, int ld_out, int coff_out, unsigned char* argmax, void* stream);
/* drs_bn_finish + drs_bn_act_pool_forward in one launch (pooled blocks with C <= 512): under data parallelism the batch-norm sums
 * come back from an all-reduce, and the kernel that normalises works (mean, rstd) out of them itself (drs_bn_finish's arithmetic, the
 * same bits), leaves them in mean_rstd for the backward pass and updates the moving averages (isprs:655-663).  pool must have bit 0 set. */
int drs_bn_finish_act_pool_forward(const double* sums, double count, float* mean_rstd, float* moving_mean, float* moving_var, double decay,
                                   int bessel_moving_var, const float* z, int B, int S, int C, float alpha, int pool, float* out, int P_out,
                                   int ld_out, int coff_out, unsigned char* argmax, void* stream);
/* the same, also (out != NULL) or only (out == NULL) writing the split-bf16 terms of the output slab (layout above) */
int drs_bn_act_pool_forward_terms(const float* z, int B, int S, int C, const float* mean_rstd, float alpha, int pool,
                                  float* out, int P_out, int ld_out, int coff_out, unsigned char* argmax,
                                  unsigned short* terms, int nterms, void* stream);

/* ---- backward of the block above ----------------------------------------------------------------------
 * reduce: ga [B*S*S][ld_ga]+coff_ga = gradient wrt the block output -> gxhat [B*S*S][C] (gradient wrt the
 *         normalised activation) and partial[drs_bn_backward_rows(B,S,C,pool)][C][2] = (sum g, sum g*xhat);
 * apply : gz = rstd * (gxhat - sum_g/count - xhat * sum_gx/count) into a haloed view (halo zeroed). */
int drs_bn_backward_rows(int B, int S, int C, int pool);
int drs_bn_backward_reduce(const float* ga, int ld_ga, int coff_ga, const float* z, const unsigned char* argmax, int B,
                           int S, int C, const float* mean_rstd, float alpha, int pool, float* gxhat, float* partial,
                           void* stream);
int drs_bn_backward_apply(const float* gxhat, const float* z, int B, int S, int C, const float* mean_rstd,
                          const double* sums, double count, float* gz, int P_out, int ld_out, int coff_out,
                          void* stream);
/* drs_bn_backward_apply with the two means as fp32 from drs_stats_reduce_means (the same bits; no fp64 division per workgroup) */
int drs_bn_backward_apply_means(const float* gxhat, const float* z, int B, int S, int C, const float* mean_rstd,
                                const float* means, float* gz, int P_out, int ld_out, int coff_out, void* stream);
/* the same, also (gz != NULL) or only (gz == NULL) writing the split-bf16 terms of the haloed gradient */
int drs_bn_backward_apply_terms(const float* gxhat, const float* z, int B, int S, int C, const float* mean_rstd,
                                const double* sums, double count, float* gz, int P_out, int ld_out, int coff_out,
                                unsigned short* terms, int nterms, void* stream);

/* ---- tf.nn.avg_pool(k x k, stride 1, SAME) of the `_avgpool` variant (isprs:753-758, 818-854) ------------------
 * forward: in [B*S*S][C] -> interior of a haloed view (halo zeroed), divisor = pixels of the window inside the image;
 * backward: gout [B*S*S][ld_g]+coff_g -> gin [B*S*S][C].  k odd. */
int drs_avg_pool_forward(const float* in, int B, int S, int C, int k, float* out, int P_out, int ld_out, int coff_out,
                         void* stream);
int drs_avg_pool_backward(const float* gout, int ld_g, int coff_g, int B, int S, int C, int k, float* gin, void* stream);

/* ---- _squeeze_excitation_layer of the `_SE` variant (isprs:682-697, _fc_layer :666-679) -------------------------
 * forward : act [B*S*S][C] (activated block output) -> s = spatial mean, e1 = relu(s w1 + b1) [B][R], e2 = sigmoid(e1 w2 + b2)
 *           [B][C] (all three kept for the backward pass), out view = act * e2 (halo zeroed);  w1 [C][R], w2 [R][C].
 * backward: gy (gradient wrt the scaled output) -> gact [B*S*S][C] and the four parameter gradients (sums over this rank's
 *           images, fixed order); scratch: B*(3*C + R) floats. */
int drs_se_forward(const float* act, int B, int S, int C, int R, const float* w1, const float* b1, const float* w2, const float* b2,
                   float* s, float* e1, float* e2, float* out, int P_out, int ld_out, int coff_out, void* stream);
int drs_se_backward(const float* gy, int ld_g, int coff_g, const float* act, const float* s, const float* e1, const float* e2,
                    const float* w1, const float* w2, int B, int S, int C, int R, float* gact, float* dw1, float* db1, float* dw2,
                    float* db2, float* scratch, void* stream);

/* ---- 1x1 classifier + sparse softmax cross-entropy + tf.argmax (+ their gradients) ----------------------
 * (isprs:1024-1031, 1089-1099, 1690; masked loss: contest_dilated_random.py:881-901; confusion matrix:
 * calc_accuracy_by_crop isprs:510-531).  feat: haloed view with C channels (multiple of 64, <= 448);
 * w [C][K], K <= 8.  labels == NULL -> inference (logits / pred only).  inv_n = 1 / number of pixels the
 * loss averages over (all ranks).  Per-workgroup slabs (rows = drs_classifier_rows(B,S)):
 * dw_partial [rows][C][K], db_partial [rows][K], loss_partial [rows] (sum of per-pixel CE, fp64).
 * conf: [K][K] counters, ADDED to (integer atomics), rows = label, cols = prediction, gated by acc_mask. */
int drs_classifier_rows(int B, int S);
int drs_classifier_loss(const float* feat, int B, int S, int P, int ld, int coff, int C, int K, const float* w,
                        const float* bias, const unsigned char* labels, const unsigned char* loss_mask,
                        const unsigned char* acc_mask, float inv_n, float* logits, unsigned char* pred, float* gfeat,
                        int ld_g, int coff_g, float* dw_partial, float* db_partial, double* loss_partial,
                        unsigned int* conf, void* stream);

/* fixed-order column sums (scratch: drs_colsum_scratch_doubles(ncols) doubles) / scalar sums used on the slabs above */
int drs_rows_reduce_f32(const float* in, int nrows, int ncols, float* out, double* scratch, void* stream);
int drs_sum_f64(const double* in, int n, double* out, void* stream);
int drs_scale_f64(double* x, int n, double s, void* stream);      /* x[i] *= s (the 1/N of the mean cross-entropy) */
/* tf.nn.l2_loss over the kernels (isprs:646-651): out[0] = 0.5 * sum w^2; scratch = 256 doubles */
int drs_l2_loss(const float* w, size_t n, double* scratch, double* out, void* stream);

/* ---- tf.train.MomentumOptimizer(momentum).minimize  (isprs:1685-1687) -----------------------------------
 * g = grad*grad_scale (+ weight_decay*w for the first n_decay entries = the kernels; biases are not decayed,
 * isprs:640-652); accum = momentum*accum + g; w -= lr*accum. */
int drs_momentum_update(float* w, const float* grad, float* accum, size_t n, size_t n_decay, float lr,
                        float weight_decay, float momentum, float grad_scale, void* stream);

/* ---- calc_accuracy_by_crop / whole-map confusion loops  (isprs:510-531, 1290-1296) ----------------------- */
int drs_confusion(const unsigned char* labels, const unsigned char* pred, const unsigned char* mask, size_t n, int K,
                  int ignore_label, unsigned int* conf, void* stream);

/* ---- dynamically_create_patches + normalize_images  (isprs:245-334, 74-81; tiler isprs:337-400) ----------
 * inst [B][4] = (map, row, col, flip) with the border shift-back already applied (isprs:260-269);
 * rot [B][6] = (m00 m01 m10 m11 off0 off1) of scipy.ndimage.rotate(order=0, reshape=False) when rot_on[b];
 * noise [B][S][S][C] (fp64, reference-exact) or NULL (device Philox N(0, 0.01) keyed by (seed, noise_index0 + b, element):
 * noise_index0 = index of this call's first patch in the global batch, so a sharded batch draws the same noise) when noise_on[b];
 * out: conv1 input slab [B][S+2P][S+2P][ld] (channels C..ld-1 and the halo are zeroed);
 * out_lab / out_mask [B][S][S] (uint8); out_mask is 0 where the rotation pulled in fill or the label equals
 * void_label (contest_dilated_random.py:235-239; -1 = none).  Normalisation touches channels 0,1,2 only.
 * mean3 / std3 are HOST pointers to 3 doubles each (copied into the kernel arguments).
 * quantize_f16: coffee_dilated_random.py:293 casts its training patches to float16 and :67-74 normalises them in that array:
 * value, value - mean and (value - mean) / std are each rounded to float16.  NumPy >= 2 evaluates float16-array (op) numpy-scalar in
 * the scalar's type: 1 = mean / std are float32 scalars (what coffee's compute_image_mean yields from its float32 patches, :78-79):
 * float32 arithmetic; 2 = float64 scalars (e.g. statistics read from a float64 .npy): float64 arithmetic.  (NumPy 1.x cast the
 * scalar to float16 first: not reproduced.) */
int drs_crop_normalize(const void* tiles, int tiles_are_f64, const unsigned char* labels, const long long* tile_off,
                       const long long* lab_off, const int* tile_h, const int* tile_w, int C, const int* inst,
                       const double* rot, const unsigned char* rot_on, const double* noise,
                       const unsigned char* noise_on, unsigned long long seed, int noise_index0, const double* mean3,
                       const double* std3,
                       int B, int S, int P, int ld, float* out, unsigned char* out_lab, unsigned char* out_mask,
                       int void_label, int quantize_f16, void* stream);

/* ---- overlap-add of window logits and arg-max of the average  (isprs:1261-1284, 1925-1949) ---------------
 * windows [first_window, first_window + n_windows) of the row-major window grid at `stride` (last row/col
 * shifted back to the border) are added, in that order, into prob [h][w][K] / occur [h][w]. */
int drs_stitch_accumulate(float* prob, unsigned int* occur, const float* logits, int h, int w, int K, int S, int stride,
                          int first_window, int n_windows, void* stream);
int drs_stitch_finalize(const float* prob, const unsigned int* occur, int h, int w, int K, unsigned char* out,
                        void* stream);
/* multi-scale evaluation (isprs:1347-1474, softmax isprs:38-43): acc[h][w][K] += softmax_k(prob / max(occur, 1));
 * the label map of the summed scales is drs_stitch_finalize(acc, ones, ...). */
int drs_softmax_accumulate(const float* prob, const unsigned int* occur, int h, int w, int K, float* acc, void* stream);

/* ==================================================================================================================
 * Step level: the reference's three `sess.run` call shapes (isprs:1750-1752 train, :1274-1275 infer, :1588 validate) as entry
 * points, for hosts that do not want to sequence the op-level calls above themselves (csrc/engine.hip).  A drs_net_t holds the
 * net table selected by `net_type` (the reference's if-chain isprs:1660-1680, both spellings of Dilated8Pooling), the variable
 * layout under TensorFlow's scope names and the launch order of a forward pass / a training step.  It owns no device memory:
 *
 *   drs_net_create(...)                                      -> handle
 *   for i < drs_net_num_buffers(h): drs_net_buffer_info(h, i, name, cap, &bytes, &dtype); allocate; drs_net_bind(h, name, ptr, bytes)
 *       (zero-fill "momentum"; fill "params" / "bn" through drs_params_set or directly: layout from drs_net_variable_info)
 *   per step:  drs_crop_normalize(... out = buffer "act:x0" with P, ld from drs_net_layout, out_lab = "labels", out_mask = "acc_mask")
 *              drs_train_step(h, B, S, lr0, flags, global_pixels, stream)     or     drs_forward(h, B, S, flags, ignore_label, stream)
 *   results in the bound buffers: "scalars" (double[4]: [0] mean CE over all ranks, [1] 0.5*sum w^2; total loss = [0] + wd*[1]),
 *       "pred" (u8 [B*S*S]), "logits" (f32 [B*S*S][K], with DRS_WANT_LOGITS), "conf" (i32 [K][K])
 *
 * One stream per handle, not re-entrant per handle, one handle per rank.  Status codes as above; nothing throws.
 * Data parallelism: the sums that must run over all ranks (sync-BN statistics forward and backward, the gradient buffer in
 * buckets, the CE sum, the confusion matrix) go through the callback given to drs_net_set_comm -- the driver owns the
 * communicator (RCCL through torch.distributed in the Python mirror).
 *   allreduce(user, dev_ptr, count, dtype (0 f32, 1 f64, 3 i32), async, stream) -> handle >= 0, or < 0 on failure; in-place sum;
 *       async = 1: may return before the sum is done, the library then calls wait(user, handle, stream) before it reads the data;
 *       async = 0: the sum must be ordered before later work on `stream`.
 *   With a callback installed the sums go through it at world = 1 too (identities there): a single-GPU host can drive the whole
 *   collective path.  world = 1 and no callback: the sums are skipped and batch norm uses its fused single-rank finish.
 */
typedef struct drs_net drs_net_t;
typedef int (*drs_allreduce_fn)(void* user, void* dev_ptr, size_t count, int dtype, int async, void* stream);
typedef int (*drs_wait_fn)(void* user, int handle, void* stream);
#define DRS_WANT_LOGITS 1     /* also write the [B*S*S][K] logits */
#define DRS_WITH_LABELS 2     /* drs_forward: add the confusion matrix of (labels, pred) into "conf" */
#define DRS_USE_ACC_MASK 4    /* gate the confusion matrix by "acc_mask" (augmentation validity / void pixels) */
#define DRS_USE_LOSS_MASK 8   /* drs_train_step: only pixels with loss_mask != 0 enter the loss (contest:881-901) */
#define DRS_NO_UPDATE 16      /* drs_train_step: leave the gradients in "grads", do not apply the momentum update */

int drs_net_create(const char* net_type, int channels, int num_classes, float weight_decay, int b_max, int s_max, int bessel_moving_var,
                   float lr_decay_factor, drs_net_t** out);
void drs_net_destroy(drs_net_t* net);
int drs_net_num_buffers(const drs_net_t* net);
int drs_net_buffer_info(const drs_net_t* net, int index, char* name, int name_cap, size_t* bytes, int* dtype);   /* dtype: 0 f32, 1 f64, 2 u8, 3 i32 */
int drs_net_bind(drs_net_t* net, const char* name, void* dev_ptr, size_t bytes);
int drs_net_buffer(drs_net_t* net, const char* name, void** dev_ptr, size_t* bytes);
int drs_net_layout(const drs_net_t* net, size_t* n_params, size_t* n_decay, size_t* n_bn, int* n_layers, int* x0_channels, int* x0_halo);
/* block `index` of the net as the library runs it (the `_conv_layer` calls of the reference's builder, isprs:761-1086, in order):
 * geom8 = (k, rate, cin, cin_k = input channels the kernel multiplies (conv1: the bands padded to 8), cout, pad_before, pad_after,
 * halo of its input slab); the slabs it reads / writes by name ("x0", "x1", ..., "feat", "concat": buffers "act:<name>"), the
 * channel offset of its output slice (dense / squeeze nets write slices of a shared slab, isprs:921-948), and the pooling after
 * its activation: 0 none, 1 max 3x3 / stride 1, 2 + 256 k = k x k average */
int drs_net_layer_info(const drs_net_t* net, int index, char* name, int name_cap, int* geom8, char* src_slab, char* dst_slab, int slab_cap,
                       int* dst_coff, int* pool);
/* the net as a whole: canonical net_type (aliases resolved), alpha of max(alpha x, x) (0 ReLU, 0.1 leaky ReLU: isprs:620-621), the
 * classifier's input width (isprs:1024-1031), the slab it reads, the topology (0 chain, 1 dense concat isprs:921-948, 2 squeeze
 * isprs:726-742) and the number of squeeze-and-excitation blocks (isprs:1036-1061) */
int drs_net_info(const drs_net_t* net, char* net_type, int name_cap, float* alpha, int* c_last, char* feat_slab, int slab_cap, int* topology,
                 int* n_se);
/* squeeze-and-excitation block `index`: scope ("se1": variables <scope>_fc1/weights ...), the block whose activation it scales,
 * channels C and the reduced width C / 4 (isprs:682-697) */
int drs_net_se_info(const drs_net_t* net, int index, char* scope, int scope_cap, int* layer, int* channels, int* reduced);
/* the net_type strings drs_net_create accepts -- the if-chains isprs:1660-1680, coffee:1188-1215, contest:995-1012 -- one per index
 * (DRS_ERR_ARG past the end): the tables, then the aliases; *canonical = index of the table the name resolves to */
int drs_net_type_name(int index, char* name, int name_cap, int* canonical);
/* variables under their TensorFlow scope names (`conv1/weights`, `conv1/biases`, `conv1/moving_mean`, `conv1/moving_variance`,
 * `conv_classifier/weights`, ...: what tf.train.Saver stores, isprs:1693-1695): offset / count in floats inside "params" (and
 * "grads", "momentum") or, with *in_bn = 1, inside "bn"; shape4 = HWIO for kernels */
int drs_net_num_variables(const drs_net_t* net);
int drs_net_variable_info(const drs_net_t* net, int index, char* name, int name_cap, size_t* offset, size_t* count, int* shape4, int* in_bn);
/* copy one variable (slot NULL) or its optimizer accumulator (slot "Momentum") to / from HOST memory; waits for the copy */
int drs_params_get(drs_net_t* net, const char* name, const char* slot, float* host_dst, size_t count, void* stream);
int drs_params_set(drs_net_t* net, const char* name, const char* slot, const float* host_src, size_t count, void* stream);
/* the flat fp32 gradient buffer the RCCL all-reduce runs on (= buffer "grads") */
int drs_grad_buffer(drs_net_t* net, float** dev_ptr, size_t* count);
long long drs_net_global_step(drs_net_t* net, long long set_to);      /* set_to < 0: read only (`main_global_step`, isprs:1685) */
float drs_net_learning_rate(const drs_net_t* net, float lr0);         /* exponential_decay(lr0, global_step, 50000, factor, staircase) */
int drs_net_set_comm(drs_net_t* net, int world, int rank, drs_allreduce_fn allreduce, drs_wait_fn wait, void* user);
/* Library-side collectives (SURVEY 8b `drs_allreduce(handle, comm)`): the sums above issued by the step engine itself through
 * RCCL, bound at run time by dlopen (csrc/rccl_comm.hip) -- no host callback in the step.  comm_small / comm_big: ncclComm_t of
 * `world` ranks on this process's GPU, made by drs_rccl_comm_create below or by the host from the same librccl; comm_big may be
 * NULL.  comm_small carries the latency-bound sums (forward sync-BN statistics on the compute stream itself, backward ones on a
 * side stream under the filter gradient of the block above), comm_big the gradient buckets on comm_stream (hipStream_t; NULL: the
 * library creates one).  The communicators remain the caller's (destroy them after the net).  comm_small = NULL removes the
 * library-side collectives (a net on the drs_net_set_comm callback, or without any communicator, is left as it is); conversely
 * drs_net_set_comm replaces library-side collectives by the callback.  One communicator is never driven from two streams that no
 * event orders: with the two-stream backward pass of small steps (fewer than 2^18 pixels per rank) comm_small stays on the compute
 * stream and every asynchronous sum goes to comm_big; with comm_big = NULL that pass is off.
 * Default form: INLINE -- comm_big is ignored and every sum is issued on the compute stream itself in program order (the sync-BN sums
 * where they are needed, the whole gradient buffer as one all-reduce after the last filter gradient, the loss, the confusion matrix):
 * no side stream, no event, one communicator on one stream.  DRS_RCCL_ASYNC=1 in the environment (read by drs_net_set_rccl) selects
 * the asynchronous form described above (side streams, gradient buckets on comm_big as the layers finish); measured at world 1 its
 * cross-stream hand-overs cost more than the overlap can return on 8 GPUs.
 * A host that binds this ABI directly must have HSA_ENABLE_IPC_MODE_LEGACY=0 in its environment BEFORE the HIP runtime starts
 * (the first HIP call of the process): on this driver RCCL's intra-node transport needs dmabuf IPC and ncclCommInitRank otherwise
 * fails with `hipIpcGetMemHandle: invalid argument` (the Python client and bench.py set it at import).
 *   drs_rccl_bind_library : name the NCCL-API shared library to bind INSTEAD of librccl (another build of RCCL; a test double).  An
 *                         explicit call of the host program, accepted only before anything below has bound a library
 *                         (DRS_ERR_ARG afterwards, or when `path` cannot be loaded): no environment variable substitutes it.
 *   drs_rccl_available : 1 if librccl (or the named library) could be bound.
 *   drs_rccl_unique_id : ncclGetUniqueId into id128 (128 bytes, host memory); rank 0 calls it, the host hands the bytes to every rank.
 *   drs_rccl_comm_create / _destroy : ncclCommInitRank / ncclCommDestroy (collective over the ranks; the device must be current).
 *   drs_rccl_all_reduce : one in-place sum over the ranks on `stream` (dtype as in drs_net_buffer_info: 0 f32, 1 f64, 3 i32) -- the
 *                         call the step engine issues; the host uses it to check a new communicator before handing it over. */
#define DRS_RCCL_FORM_INLINE 1    /* one communicator, every sum on the compute stream in program order (default) */
#define DRS_RCCL_FORM_ASYNC 2     /* DRS_RCCL_ASYNC != 0: two communicators, side streams, an event hand-over per asynchronous sum (r03) */
#define DRS_RCCL_FORM_BUCKETS 3   /* DRS_RCCL_BUCKETS >= 2: inline + the gradient buffer as two all-reduces on comm_big's stream, the first under the rest of the backward pass */
int drs_rccl_form(void);          /* the form drs_net_set_rccl will take, from the environment (the one place that parses it); ASYNC and BUCKETS want comm_big */
int drs_rccl_bind_library(const char* path);
int drs_rccl_available(void);
int drs_rccl_unique_id(unsigned char* id128);
int drs_rccl_comm_create(int world, int rank, const unsigned char* id128, void** comm);
int drs_rccl_comm_destroy(void* comm);
int drs_rccl_all_reduce(void* comm, void* dev_ptr, size_t count, int dtype, void* stream);
int drs_net_set_rccl(drs_net_t* net, int world, int rank, void* comm_small, void* comm_big, void* comm_stream);
int drs_train_step(drs_net_t* net, int B, int S, float lr0, int flags, double global_pixels, void* stream);
int drs_forward(drs_net_t* net, int B, int S, int flags, int ignore_label, void* stream);
int drs_apply_update(drs_net_t* net, float lr0, void* stream);        /* the update alone (after DRS_NO_UPDATE) */
/* the backward pass of drs_train_step on two streams (the filter gradients on a stream of the library's beside the batch-norm-backward /
 * input-gradient chain on `stream`): mode -1 = by the library's rule (steps of fewer than 2^18 pixels; the default), 0 = never (a
 * host that must see every launch of the step on ITS stream), 1 = always.  Bitwise the same step in every mode. */
int drs_net_set_two_streams(drs_net_t* net, int mode);
/* per-kernel-family HIP-event timing of the launches of a step (bench.py's roofline figures); off by default */
int drs_net_timing(drs_net_t* net, int enable);
int drs_net_num_timing_kinds(void);
int drs_net_timing_summary(drs_net_t* net, int kind, char* name, int name_cap, int* launches, double* ms, double* work);

#ifdef __cplusplus
}
#endif
#endif /* DRS_H_ */

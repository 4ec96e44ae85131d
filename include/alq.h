/*
 * alq.h - C ABI of the MI355X-native active-learning query-scoring library (libalq.so).
 *
 * The reference (jsourati/nn-active-learning) has no FFI: its device boundary is
 * `sess.run(...)` on a TensorFlow-1.x graph.  Each entry point below names the reference
 * call site(s) whose device work it replaces (file:line under /root/reference).
 *
 * Conventions
 *   - plain C, no torch / C++ types in any signature;
 *   - every function returns an int status: 0 = ok, negative = error
 *     (alq_last_error() gives the text); nothing throws, nothing exits;
 *   - pointers named `d_*` are DEVICE pointers owned by the caller (e.g. torch-ROCm
 *     tensors); pointers named `h_*` are HOST pointers; the library never frees caller memory;
 *   - all work is enqueued on the stream given to alq_ctx_create (a hipStream_t passed as
 *     void*; NULL = the null stream) and is stream-ordered; the calls do not synchronise
 *     unless documented.  A context also owns one private side stream: alq_fisher runs the
 *     per-layer statistics kernels there, forked from and joined back into the caller's stream
 *     with events inside the call, so every result is ordered on the caller's stream as if the
 *     whole call had run on it (ALQ_NO_SIDE_STREAM=1 at context creation: one stream only);
 *   - tensors are channels-last fp32: patches [N, D, H, W, C] (D = 1 for the 2-D nets,
 *     i.e. the reference's [N, H, W, C] placeholder, NN.py:1338-1344).
 */
#ifndef ALQ_H
#define ALQ_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct alq_ctx alq_ctx;
typedef struct alq_model alq_model;

enum { ALQ_OK = 0, ALQ_EINVAL = -1, ALQ_EHIP = -2, ALQ_ENOMEM = -3, ALQ_EUNSUPPORTED = -4 };

enum { ALQ_CONV = 0, ALQ_CONVT = 1, ALQ_POOL = 2, ALQ_FC = 3 };

/* One layer of the reference's layer dicts (NN.py:96-110; NN_extended.py:103-124). */
typedef struct {
    int32_t type;      /* ALQ_CONV / ALQ_CONVT / ALQ_POOL / ALQ_FC                               */
    int32_t cout;      /* output channels (conv, conv_transpose) or units (fc); unused for pool  */
    int32_t k[3];      /* kernel (conv / conv_transpose) or window (pool), order D,H,W; 1 = n/a  */
    int32_t s[3];      /* strides, order D,H,W                                                   */
    int32_t relu;      /* 1: ReLU after the main op (NN.py:290,326-327; 'A' in NN_extended.py:352-355) */
    int32_t skip_src;  /* index of the earlier layer whose OUTPUT is concatenated IN FRONT of this
                          layer's input ('con' skip, NN_extended.py:1207-1214), or -1            */
} alq_layer_t;

const char *alq_last_error(void);
int alq_version(void);

/* ---- context ---------------------------------------------------------------------------- */
/* Replaces: tf.Session creation (PW_AL.py:363-377).  `stream` is a hipStream_t or NULL.       */
int alq_ctx_create(int device, void *stream, alq_ctx **out);
int alq_ctx_destroy(alq_ctx *ctx);
/* Moves the context to another stream; the new stream first waits (event) for everything the library enqueued on the old
 * one, so back-to-back calls on two streams need no host synchronisation.                                           */
int alq_ctx_set_stream(alq_ctx *ctx, void *stream);
/* on = 0: the per-layer statistics kernels of later alq_fisher calls stay on the caller's stream instead of the context's side
 * stream (on = 1, the default, restores it).  For a caller that overlaps WHOLE passes on two contexts (device.DeviceModel's two
 * scoring pipelines): the other pipeline's launches already fill the gaps, and four streams competing cost 1.2 % (round 6).
 * Results do not depend on it (same kernels, same order per context).                                                      */
int alq_ctx_use_side_stream(alq_ctx *ctx, int on);
int alq_ctx_synchronize(alq_ctx *ctx);

/* ---- model ------------------------------------------------------------------------------ */
/* Replaces: graph construction NN.CNN.__init__ (NN.py:147-188) / NN_extended.CNN.__init__
 * (NN_extended.py:187-295) + get_gradients (NN.py:621-645).  in_dims = {D, H, W, C}.
 * max_batch = largest N any later call will pass (activation workspace is sized for it).     */
int alq_model_create(alq_ctx *ctx, const alq_layer_t *layers, int n_layers,
                     const int32_t in_dims[4], int max_batch, alq_model **out);
int alq_model_destroy(alq_model *m);
/* Patches per device pass the model's workspace holds: min(max_batch asked for, what the engines' unsigned 32-bit tensor
 * offsets allow - every allocation below 2^30 floats; NET-C at 32^3: 2047).  A larger batch is walked in passes of this
 * size by the host (the reference feeds `ntb` patches per sess.run, PW_NN.py:447-451: the split never changes a result). */
int alq_model_max_batch(const alq_model *m);
/* Number of parameterised layers L ( = len(grad_posts['1'])/2, PW_NNAL.py:751 ).             */
int alq_model_num_param_layers(const alq_model *m);
/* W/b element counts of parameterised layer t (creation order), i.e. prod(W.shape), len(b).  */
int alq_model_param_sizes(const alq_model *m, int t, int64_t *w_elems, int64_t *b_elems);
/* Length of the feature vector of layer `layer_idx`'s output (feature_layer, NN.py:172-175). */
int alq_model_layer_out_elems(const alq_model *m, int layer_idx, int64_t *elems);
/* Replaces: CNN.load_weights / perform_assign_ops (NN.py:396-419, :462-519).  HOST pointers,
 * TF layouts: conv [k...,Ci,Co], conv_transpose [k...,Co,Ci], fc [out,in] with `in` in the
 * reference's flatten order (full axis reversal, NN.py:296-301); bias [Co] / [out,1].
 * Synchronises the stream before returning (the host buffers may be reused at once).         */
int alq_model_set_weights(alq_model *m, int t, const float *h_W, const float *h_b);

/* ---- patch gather + normalisation ------------------------------------------------------- */
/* Replaces: patch_utils.get_patches (patch_utils.py:1087-1173) + the normalisation loops of
 * PW_NN.batch_eval (PW_NN.py:503-506) / PW_NNAL.CNN_query (PW_NNAL.py:125-129) [quirk = 1:
 * channel j < m is normalised with stats[j]] or of patch_utils.get_patches_multimg
 * (patch_utils.py:1203-1207) [quirk = 0: depth slab j*d3..(j+1)*d3 with stats[j]].
 * d_vols: m device pointers (host array of device pointers) to zero-padded volumes, C order,
 * element type double (vol_is_f64 = 1) or float; pad_dims = padded shape; d_inds: int64
 * raveled indices in UN-padded coordinates; h_stats: m pairs (mu, sigma), double.
 * quirk = 2: no normalisation (plain get_patches).  d_out: [n, d1, d2, m*d3], float
 * (out_is_f64 = 0) or double (out_is_f64 = 1, bit-identical to the reference's float64
 * patches).  Arithmetic is fp64, rounded once to fp32 for the float output, like the
 * reference's float64 patches fed to a float32 placeholder.                                  */
int alq_gather_normalize(alq_ctx *ctx, const void *const *d_vols, int m, int vol_is_f64,
                         const int64_t pad_dims[3], const int64_t *d_inds, int64_t n,
                         const int32_t patch_shape[3], const double *h_stats, int quirk,
                         int out_is_f64, void *d_out);

/* ---- forward ---------------------------------------------------------------------------- */
/* Replaces: sess.run(model.posteriors / prediction / feature_layer) in PW_NN.batch_eval
 * (PW_NN.py:522-524).  d_x: [N, D,H,W,C]; d_post: [2... c, N] row-major like the reference's
 * [c, N] posteriors (NN.py:184-188); d_pred: int64 [N] or NULL; d_feat: [N, F] of layer
 * feature_layer_idx (or NULL / -1).  N <= max_batch.                                         */
int alq_forward(alq_model *m, const float *d_x, int N, float *d_post, int64_t *d_pred,
                float *d_feat, int feature_layer_idx);

/* Same, on rows of a resident pool: patch i of the batch is d_pool[d_rows[i]] (d_pool: [n_pool, D,H,W,C] fp32,
 * d_rows: int64 [N] device).  Replaces the `inds[batches[j]]` indirection of PW_NN.batch_eval (PW_NN.py:447-451,
 * :498-501) for pools that live in HBM: the caller passes positions instead of gathering a copy of the patches. */
int alq_forward_rows(alq_model *m, const float *d_pool, const int64_t *d_rows, int N, float *d_post,
                     int64_t *d_pred, float *d_feat, int feature_layer_idx);

/* ---- uncertainty scores ----------------------------------------------------------------- */
/* Replaces: np.abs(posts - .5) (PW_NNAL.py:64,109,728) and NNAL_tools.compute_entropy
 * (NNAL_tools.py:71-85).  d_p1: float [n] (row 1 of the posteriors); d_absdev: double [n]
 * = |double(p1) - 0.5| (exact); d_H: float [n] Shannon entropy of (1-p1, p1) with the
 * reference's +10e-8 guard on exact zeros, or NULL.                                          */
int alq_score_entropy(alq_ctx *ctx, const float *d_p1, int64_t n, double *d_absdev, float *d_H);

/* Replaces: np.argsort(np.abs(posts - .5))[:B] (PW_NNAL.py:64,109-110,671-681,730).
 * Ascending keys, ties -> lower index first (the reference's tie order is unspecified).
 * d_out_idx: int64 [B].  d_work: device scratch of alq_topk_work_bytes(n) bytes.             */
size_t alq_topk_work_bytes(int64_t n);
int alq_topk_uncertain(alq_ctx *ctx, const double *d_keys, int64_t n, int64_t B,
                       int64_t *d_out_idx, void *d_work);

/* Multi-GPU top-B merge step (no reference counterpart: the reference is one process).
 * Host function: merges the candidate (key, GLOBAL index) pairs gathered from all ranks
 * (torch.distributed all_gather over RCCL in pool_shard.merge_topB; entries with index < 0 are padding)
 * into the global top-B, ascending key (compared by bit pattern, so a NaN key sorts behind every number),
 * ties -> lower global index: the rule of alq_topk_uncertain, so the result is identical on every rank.
 * out_idx: int64 [B]; returns the count in *n_out.                                               */
int alq_topk_merge(const double *h_keys, const int64_t *h_idx, int64_t n, int64_t B, int64_t *h_out_idx,
                   int64_t *n_out);

/* The other multi-GPU exchange (no reference counterpart): all-reduce(sum) of the L x L fp64
 * Fisher sum  sum_i A_i  over the ranks that share one pool, on RCCL over xGMI, enqueued on the context's
 * stream.  One communicator per context: rank 0 calls alq_comm_unique_id (128 bytes, host), the host side
 * distributes the id (pool_shard.attach_comm broadcasts it through torch.distributed) and every rank calls
 * alq_comm_init (collective: returns when all `world` ranks have joined).  alq_allreduce_sum reduces d_buf
 * [count] doubles IN PLACE; stream-ordered, does not synchronise.  librccl.so.1 is resolved at run time
 * (dlopen): a process without RCCL gets ALQ_EUNSUPPORTED from these three calls and nothing else changes. */
#define ALQ_COMM_ID_BYTES 128
int alq_comm_unique_id(void *h_id);
int alq_comm_init(alq_ctx *ctx, const void *h_id, int rank, int world);
int alq_comm_destroy(alq_ctx *ctx);
int alq_allreduce_sum(alq_ctx *ctx, double *d_buf, int64_t count);

/* ---- Fisher scoring --------------------------------------------------------------------- */
/* Replaces: the per-sample loop of PW_NNAL.gen_A_matrices (PW_NNAL.py:757-814): up to two
 * sess.run(model.grad_posts[j]) at batch 1, NNAL_tools.shrink_gradient(...,'sum')
 * (NNAL_tools.py:784-796) and the outer products.  d_x: [N, D,H,W,C] normalised patches.
 * d_p1_in: float [N] posteriors to branch on (what batch_eval returned, PW_NNAL.py:767) or
 * NULL to use the posteriors of this forward pass.  Outputs (any may be NULL):
 *   d_p1_out float [N]; d_g0, d_g1 double [N, L] (shrunk class-0 / class-1 gradients, zero
 *   where the reference's saturation branch skips them); d_A double [N, L, L];
 *   d_trace double [N]; d_Asum double [L, L] = sum_i A_i (overwritten, deterministic order). */
int alq_fisher(alq_model *m, const float *d_x, int N, const float *d_p1_in, double diag_load,
               float *d_p1_out, double *d_g0, double *d_g1, double *d_A, double *d_trace,
               double *d_Asum);

/* Same, on rows of a resident pool (see alq_forward_rows): the B filtered candidates `sel_inds` of
 * PW_NNAL.py:108-129 / :549-559 are passed as positions, not as a gathered copy.  d_p1_in, outputs: per row. */
int alq_fisher_rows(alq_model *m, const float *d_pool, const int64_t *d_rows, int N, const float *d_p1_in,
                    double diag_load, float *d_p1_out, double *d_g0, double *d_g1, double *d_A,
                    double *d_trace, double *d_Asum);

/* ---- parameter gradients, dropout passes, optimiser steps ------------------------------- */
/* Total number of parameters P = sum_t (w_elems + b_elems); the flat parameter / gradient vector of this API is
 * [W_0, b_0, W_1, b_1, ...] in variable-creation order and TF layouts (see alq_model_set_weights).            */
int64_t alq_model_num_params(const alq_model *m);

/* Replaces: sess.run(model.posteriors, {keep_prob: p < 1}) of the MC strategies (PW_NNAL.py:232-282: MC-entropy,
 * BALD).  Dropout (x * keep / keep_prob, NN.py:169-171) on the OUTPUT of the listed layers with a counter-based mask
 * keyed (seed, layer, first_sample + n, element): reproducible and independent of the batch split; TF's own random
 * stream is not reproducible outside TF, so the mask is this library's (the oracle restates the generator).     */
int alq_forward_dropout(alq_model *m, const float *d_x, int N, float keep_prob, uint64_t seed, int64_t first_sample,
                        const int32_t *h_drop_layers, int n_drop_layers, float *d_post, int64_t *d_pred);

/* Replaces: sess.run(model.grad_posts[str(cls)], ...) = tf.gradients(tf.log(posteriors[cls, 0]), variables)
 * (NN.py:639-645; mode 0: one full gradient per sample, the reference feeds batches of one, PW_NNAL.py:773-807) and
 * the gradient half of sess.run(model.train_step, {x, y_, keep_prob}) (NN.py:583-615; mode 1: gradient of
 * loss_scale * sum_n CE(softmax(z_n), label_n), loss_scale = 1 / batch for the reference's reduce_mean; d_labels
 * int32 [N], a label outside [0, c) contributes nothing).  per_sample = 1: d_grads [N, P]; 0: [P] summed over the
 * batch (fixed order, fp64).  d_post [c, N] and d_loss (mean CE of the batch, mode 1) optional.  Dropout as above
 * (keep_prob = 1: none).  Never on the 'sum'-shrink scoring path, which forms no weight gradient (alq_fisher).  */
int alq_param_grads(alq_model *m, const float *d_x, int N, int mode, int cls, const int32_t *d_labels,
                    float loss_scale, float keep_prob, uint64_t seed, int64_t first_sample,
                    const int32_t *h_drop_layers, int n_drop_layers, int per_sample, float *d_grads,
                    float *d_post, double *d_loss);

/* Replaces: tf.train.GradientDescentOptimizer / AdamOptimizer .minimize (NN.py:591-615) on flat device vectors:
 * theta -= lr g;  Adam (TF-1.x defaults are the caller's: beta1 .9, beta2 .999, eps 1e-8), step count t >= 1:
 * lr_t = lr sqrt(1 - beta2^t) / (1 - beta1^t), m = b1 m + (1-b1) g, v = b2 v + (1-b2) g^2,
 * theta -= lr_t m / (sqrt(v) + eps).  The updated vector goes back through alq_model_set_weights.              */
int alq_sgd_step(alq_ctx *ctx, float *d_theta, const float *d_grad, int64_t n, float lr);
int alq_adam_step(alq_ctx *ctx, float *d_theta, const float *d_grad, float *d_m, float *d_v, int64_t n, float lr,
                  float beta1, float beta2, float eps, int64_t t);
/* Replaces: the accumulation loop of model_utils.diagonal_Fisher (model_utils.py:294-330):
 * d_acc[i] += sum_n d_grads[n, i]^2 (fp64).                                                                    */
int alq_sq_accum(alq_ctx *ctx, const float *d_grads, int64_t per_sample_len, int N, double *d_acc);

/* ---- image-level multi-class Fisher query (NNAL.py:312-464) --------------------------------------------------- */
/* Replaces: NNAL_tools.shrink_gradient(grads[str(j)], 'sum') (NNAL_tools.py:784-796) on the per-sample gradient lists that
 * session.run(nz_classes_grads) returns (NNAL.py:381-397), with the gradients staying on the device: d_grads [N, P] rows as
 * alq_param_grads(mode 0, per_sample 1) writes them, h_layer_elems [L] = |W_t| + |b_t| (host; their sum must be P);
 * d_out [N, L] double: (sum of layer t's entries) / (|W_t| + |b_t|), fp64 sums in a fixed order.                        */
int alq_shrink_sum(alq_ctx *ctx, const float *d_grads, int N, int64_t P, const int64_t *h_layer_elems, int L, double *d_out);
/* Replaces: the accumulation `Ai += np.outer(g_j, g_j) / new_posts[j] + np.eye(A_size) * 1e-5` over the selected classes
 * (NNAL.py:399-409).  d_g [N, c, L] shrunk class gradients, d_w [N, c] = 1 / new_posts for the classes the reference
 * keeps (posterior >= 1e-6; the ten largest when ten or more remain, :381-394) and 0 for the others - host logic on the
 * posteriors, like the reference's -, d_diag [N] = (number of kept classes) * 1e-5; d_A [N, L, L] double.             */
int alq_fisher_classes(alq_ctx *ctx, const double *d_g, const double *d_w, const double *d_diag, int N, int c, int L,
                       double *d_A);

/* ---- feature similarities (representativeness strategies) -------------------------------------------- */
/* Replaces: the NumPy similarity blocks of query_multimg 'rep-entropy' (PW_NNAL.py:318-327: norms, dots = F.T @ F_u,
 * sims = dots / outer(norms)) and 'core-set' (:386-425, :437-441).  Features are fp32 rows [samples, f] as
 * alq_forward returns them; accumulation is fp64 like the reference's float64 arrays.
 *   alq_row_norms      d_norms[i] = ||A[i, :]||
 *   alq_cosine_sims    d_C[i, j] = <A[i], B[j]> / (na[i] nb[j])   (both norm vectors null: plain dot products)
 *   alq_colsum_max     d_out[j] = sum_i max(d_cmax[i], S[i, j]) over the rows with d_skip[i] == 0 (either may be null):
 *                      the representativeness score of every candidate column in ONE pass (PW_NNAL.py:333-338 scores one
 *                      candidate per Python iteration); fixed summation order; d_work: alq_colsum_work_bytes(n, b) bytes
 *   alq_take_colmax    d_v[i] = max(d_v[i], S[i, j])  (first != 0: d_v[i] = S[i, j]): the running row maxima after a pick
 *   alq_fold_rowmax    d_v[j] = max(d_v[j], max_i S[i, j]) for a [t, n] block (core-set's labelled-set maxima, :420-425) */
int alq_row_norms(alq_ctx *ctx, const float *d_A, int64_t n, int f, double *d_norms);
int alq_cosine_sims(alq_ctx *ctx, const float *d_A, int64_t n, const float *d_B, int b, int f, const double *d_na,
                    const double *d_nb, double *d_C);
size_t alq_colsum_work_bytes(int64_t n, int b);
int alq_colsum_max(alq_ctx *ctx, const double *d_S, int64_t n, int b, const double *d_cmax,
                   const unsigned char *d_skip, double *d_out, void *d_work);
int alq_take_colmax(alq_ctx *ctx, const double *d_S, int64_t n, int b, int j, int first, double *d_v);
int alq_fold_rowmax(alq_ctx *ctx, const double *d_S, int t, int64_t n, double *d_v);

/* ---- fp64 accuracy reference on the device (csrc/ref64.hip; bench.py `accuracy`, tests) ------------------------------- */
/* One inverted decision of an fp64 evaluation.  layer = model layer index of a ReLU'd conv / conv_transpose / fc layer: the
 * ReLU decision of element `idx` of that layer's pre-activation of ONE sample ([vox, C] flattened; fc: the unit) is
 * inverted (passes although <= 0, cut although > 0); layer = index of a max-pool layer: element `idx` of the pool's INPUT is
 * lifted by `delta` before the window maximum is taken (a near-tie decided the other way).  layer < 0: unused slot.
 * In candidate lists `pad` is 0 for a ReLU unit, 1 for a pool window.                                                   */
typedef struct {
    int32_t layer;
    int32_t pad;
    int64_t idx;
    double delta;
} alq_flip_t;
/* fp64 evaluation of the scored path for N samples (rows d_rows of the fp32 pool d_x, or its first N rows when d_rows is
 * NULL): the reference's graph (conv / conv_transpose / max-pool / fc, 'con' skips; NN.py:258-340, NN_extended.py:366-601)
 * in double precision, one backward pass with the unit cotangent (+1, -1) on the two logits, and the per-layer sums
 * S[n][t] = sum of all entries of d(z0 - z1)/d(theta_t) - what NNAL_tools.shrink_gradient(., 'sum') (NNAL_tools.py:784-796)
 * divides by the layer's size.  h_dW / h_db: host arrays of DEVICE pointers to the fp64 weights / biases of the parameterised
 * layers, TF layouts as alq_model_set_weights EXCEPT fc: [out][in] with `in` in ACTIVATION-MEMORY order ([D, H, W, C]
 * flattened) of the layer's input.  d_flips: [N][flips_per_sample] decisions to invert per sample, or NULL.
 * cand_cap > 0: d_cand / d_cand_key / d_cand_count [N][cand_cap] / [N] receive every FRAGILE decision of a sample - a ReLU
 * input with |pre-activation| <= eps * (rms pre-activation of that layer and sample), key = that ratio; a pool window whose two
 * largest inputs lie within eps * (rms of the pool's input) of each other with a positive maximum, key = gap / rms, idx = the
 * runner-up, delta = the lift that makes it win - in no particular order; d_cand_count may exceed cand_cap (list truncated).
 * d_logits [N][2], d_S [N][L], d_rms [n_layers][N] (rms pre-activation / pool input per layer and sample; may be NULL).
 * Allocates its workspace per call and synchronises the stream: an accuracy tool, never on the scoring path.             */
int alq_ref64_scores(alq_ctx *ctx, const alq_layer_t *layers, int n_layers, const int32_t in_dims[4],
                     const double *const *h_dW, const double *const *h_db, const float *d_x, const int64_t *d_rows, int N,
                     const alq_flip_t *d_flips, int flips_per_sample, double eps, int cand_cap,
                     double *d_logits, double *d_S, double *d_rms, alq_flip_t *d_cand, double *d_cand_key, int32_t *d_cand_count);

/* ---- measurement hooks (bench.py only) -------------------------------------------------- */
/* Per-kernel-class HIP-event timing on the context's stream.  alq_prof_enable(ctx, 1) makes
 * every launch of an instrumented kernel class record start/stop events; on = k > 1 samples the
 * launches of every k-th alq_fisher pass only (the event pairs themselves cost a few percent);
 * alq_prof_read returns, for class `cls`, the accumulated milliseconds, launch count and
 * algorithmic FLOPs since the last alq_prof_reset (synchronises the stream).                 */
int alq_prof_enable(alq_ctx *ctx, int on);
int alq_prof_reset(alq_ctx *ctx);
int alq_prof_num_classes(void);
const char *alq_prof_class_name(int cls);
int alq_prof_read(alq_ctx *ctx, int cls, double *ms, int64_t *launches, double *flops);

/* Debug / test hook: copies an internal tensor of the last alq_forward / alq_fisher call into a
 * dense device buffer.  what: 0 = activation of layer `layer_idx` [N, vox, C], 1 = its cotangent
 * (after the ReLU mask), 2 = channel-sum field of the layer's INPUT [N, vox_in], 3 = channel-sum
 * field of its masked cotangent [N, vox_out], 4 = the unit-cotangent layer sums S [N, L] (as
 * float; layer_idx ignored), 5 = the fused fc head's logit-difference partials [N, tiles * 4] (one per
 * (tile, wave) of the last conv's launch).  *elems_out receives the element count.  Tests only.  */
int alq_model_debug_copy(alq_model *m, int layer_idx, int what, int N, float *d_out,
                         int64_t *elems_out);

/* Diagnostic builds only (-DALQ_STAMPS): device buffer of 8 uint64 per workgroup that receives the
 * per-phase shader-clock ticks of the pipelined GEMM kernel's next launches; NULL switches it off.
 * A no-op in the product build.                                                               */
int alq_debug_set_stamp_buffer(void *d_buf);
/* Timing-experiment knobs (tests / profiling only; results may be wrong while a knob is set):
 * key 0 = repeat the MFMA phase n extra times, 1 = flag bits (1 no stores, 2 no loads, 4 no sum
 * MFMAs, 8 no sum stores), 2 = no epilogue fusion in backward GEMMs, 3 = none in forward GEMMs,
 * 4 = use the fp32-MFMA GEMM kernel instead of the bf16x3 split kernel.                         */
int alq_debug_set(int key, int value);
/* What the last pass of a model ran on (tests / bench reporting).  what = 0: 1 when the matrix cores of the context's device
 * keep fp16 subnormal operands (probed once; the one-accumulator form of the plane-sweep engine needs it), 1: 1 when the last
 * forward pass ran the conv under the two-class head on the plane-sweep engine (csrc/c3d.hip; replaces the tf.nn.conv3d call
 * site NN_extended.py:416-426 for that layer), 2: the same for the last backward pass, 3: 1 when that engine accumulates the
 * three piece products in one accumulator, 5: the number of marked 4-channel groups that did not fit their list segment (a quarter of a patch,
 * 128 slots) since the model was created - 0 on every input the tests and the bench use; none is dropped: an overflowing
 * segment is drained by the sweep path of the fix-up kernel (synchronises the stream).
 * 6: 1 when the last forward pass ran a launch on the fp16-pair split with derived input bounds (default; ALQ_NO_F16_DERIVED=1 off).
 * 7 / 8: conv_transpose launches of the last forward / backward pass on the row-sweep engine (csrc/t3d.hip), 9: 1 when the last
 * backward pass ran enc2's backward fused with both pool backward steps (csrc/e3d.hip), 10: 1 when the last forward pass ran
 * dec1 on the plane-sweep kernel of csrc/d3d.hip, 11: the same for its backward-data launch, 12: 1 when the last forward pass
 * ran enc2 and the max-pool behind it as one launch (csrc/f3d.hip), 13: which form of the head conv's backward kernel the last backward
 * pass ran (csrc/c3d.hip): 7 = the 27 taps packed into 7 k-steps (default), 8 / 4 = the 9-k-step kernel (ALQ_C3D_BWD_ROWS), 0 = none.
 * Returns the answer or a negative error code.  */
int alq_model_engine_info(alq_model *m, int what);

/* Synthetic patch generator: counter-based RNG keyed (seed, patch_id, element), standard
 * normal, written to d_out [n, elems_per_patch] for patch ids first_id .. first_id+n-1
 * (shards are reproducible whatever the sharding).                    */
int alq_synth_patches(alq_ctx *ctx, uint64_t seed, int64_t first_id, int64_t n,
                      int64_t elems_per_patch, float *d_out);

#ifdef __cplusplus
}
#endif
#endif /* ALQ_H */

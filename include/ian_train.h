/* ian_train.h -- C ABI of the training-step building blocks of libian.so.
 *
 * The reference's training step (train_IAN.py:47-352 make_training_functions) is a Python function that wires
 * Lasagne layers (layers.py: DeconvLayer, MDCL, MinibatchLayer, beta_layer, MADE/IAFLayer; Lasagne: Conv2DDNNLayer,
 * DenseLayer, batch_norm) into one Theano graph and lets Theano differentiate it.  Here the same wiring is done -- by
 * ian_train_step (bottom of this file; csrc/ian_trainer.cpp) for one GPU, and by the Python host
 * (neural_photo_editor_amd/trainer.py) for the data-parallel step -- over the entry points below:
 *   - ian_layer_*: one object per linear Lasagne layer (conv / transposed conv / MDCL / dense) owning its packed
 *     weights, with forward, backward-data and backward-weight -- the three cuDNN/GEMM calls Theano would emit;
 *   - ian_k_*:     the element-wise / reduction ops around them (batch-statistics batch-norm, activations,
 *     MinibatchLayer, losses, Adam ...), one HIP launch sequence each.
 * Conventions: every pointer is DEVICE memory owned by the caller (torch tensors are only the container);
 * activations are NHWC float32 with the channel (pixel) stride rounded up to a multiple of 32; parameters and their
 * gradients are in the REFERENCE layout (Theano shapes) so that checkpoints and optimiser state
 * are layout independent; all work is ordered on `stream`; return 0 / negative, text via ian_k_last_error().
 * All reductions use fixed summation orders: results are bitwise reproducible run to run.
 */
#ifndef IAN_TRAIN_H_
#define IAN_TRAIN_H_

#include <stdint.h>

#include "ian.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ian_layer ian_layer;

/* Build a layer from the geometry fields of an ian_op_desc (kind, cin, cout, in_h, in_w, flat, unflat, scales).
   kind CONV5S2 : Conv2D(DNN)Layer 5x5/s2/p2, W (Cout,Cin,5,5)                      IAN.py:71-110
        DECONV5S2: layers.py:436-483 DeconvLayer, W (Cin,Cout,5,5)                   IAN.py:139-181
        MDC3    : layers.py:207-258 MDCL, params W (Cout,Cin,3,3), coeff_base, then one coefficient per scale
        DENSE   : DenseLayer, W (in,out)                                             IAN.py:114-134
   Parameter order for ian_layer_set_params / ian_layer_backward_weight: W first, then (MDC3) the coefficients in
   the order coeff_base, scales[0], scales[1], ... (scale 0 = the 1x1 mean branch). */
int ian_layer_create(const ian_op_desc* desc, int32_t deconv_flip, ian_layer** out);
/* Number of parameter tensors and elements of each (reference layout). */
int32_t ian_layer_num_params(ian_layer* l);
int64_t ian_layer_param_numel(ian_layer* l, int32_t which);
/* (Re)pack the current parameter values (device pointers, reference layout) into the kernel layouts.  Call after
   every optimiser update.  LIFETIME: an MDCL layer keeps the pointers -- its backward-weight reads W and the branch
   coefficients again (dW = sum_b coeff_b * dS, dcoeff_b = <dS, W>) -- so the parameter buffers must stay valid (and hold
   the values of this call) until the next ian_layer_set_params / ian_layer_destroy. */
int ian_layer_set_params(ian_layer* l, const float* const* params, int32_t nparams, void* stream);
/* y = act(x (*) W + bias + res).  bias / res may be NULL; bias is indexed by output channel (internal order).
   y_stride: pixel stride of y (0 = round_up(cout,32)). */
int ian_layer_forward(ian_layer* l, const float* x, int32_t n, float* y, int32_t y_stride, const float* bias,
                      const float* res, int32_t act, void* stream);
/* dx (+)= dy (*) W^T   (gradient wrt the layer input; dy is the gradient wrt the pre-activation output) */
int ian_layer_backward_data(ian_layer* l, const float* dy, int32_t n, float* dx, int32_t dx_stride, int32_t accumulate,
                            void* stream);
/* dparams[i] (+)= d loss / d param_i in the reference layout (same order as set_params). */
int ian_layer_backward_weight(ian_layer* l, const float* x, const float* dy, int32_t n, float* const* dparams,
                              int32_t nparams, int32_t accumulate, void* stream);
/* Batch statistics in the epilogue of the launch that produces the tensor (round 5).  ian_layer_stats_next arms the layer's NEXT
   ian_layer_forward / ian_layer_backward_data (/ the data-gradient GEMM of ian_layer_head6_backward) -- one shot -- to write, beside
   its output, per row tile and channel the float64 partial sums the two-stage statistics of ian_k_colstats start from:
     mode 1: s1 = sum v, s2 = sum v*v of the stored values v            (ian_k_colstats mode 0: batch-norm forward)
     mode 2: g = v * act'(a), s1 = sum g, s2 = sum g * (yraw - mean) * inv_std   (mode 1: dbeta, dgamma; a / yraw laid out like the
             output; yraw NULL: s2 = 0)
   partial[chunk][2][C] doubles, cap_doubles available.  ian_layer_stats_chunks: number of chunks the last launch wrote -- feed
   them to ian_k_tree_sum / ian_k_bn_finish / ian_k_bn_bwd_finish as `nchunks` -- or 0 when that launch could not carry them
   (split-K schedule, VALU kernel, workspace too small -- and ALWAYS in the product library: the statistics epilogue measured
   4 % slower per training update than the colstats passes it replaces and is compiled into libian_ablation.so only): the caller
   then runs ian_k_colstats as before.  Chunk boundaries are multiples of the GEMM's row-tile height in (image, pixel) order;
   sums differ from ian_k_colstats' in summation order only. */
int ian_layer_stats_next(ian_layer* l, int32_t mode, const float* a, const float* yraw, const float* mean, const float* inv_std,
                         int32_t act, double* partial, int64_t cap_doubles);
int32_t ian_layer_stats_chunks(ian_layer* l);
/* Forward of THREE 2-filter MDCL layers that read the same 128-channel map with the same tap list (the R, G_a, B_a layers
   of the RGB-Beta head, IAN.py:183-199) in one pass over the map: y_k = act_k(x (*) W_k), k = 0..2, each an NHWC map with
   pixel stride y_stride (kernels_head.hip head6_kernel: one dense 128 -> 6*taps contraction per pixel on the matrix
   cores + shifted adds).  Returns -4 when the layers do not have that shape (use ian_layer_forward then). */
int ian_layer_head6_forward(ian_layer* l0, ian_layer* l1, ian_layer* l2, const float* x, int32_t n, float* y0, float* y1,
                            float* y2, int32_t y_stride, int32_t act0, int32_t act1, int32_t act2, void* stream);
/* Backward of the same three layers, contract-first as well.  dy_k = gradient wrt the pre-activation output of layer k (NHWC,
   pixel stride dy_stride).  A shifted gather builds Z[q][(t,k,f)] = dy_k[q - d_t][f] once; then
     dx (optional, NHWC stride dx_stride; += when dx_accumulate)  = Z . Wcat      -- ONE dense GEMM, K = 6*taps, instead of
                                                                                     three tap GEMMs whose K is 94 % padding
     dparams_k[i] (optional, all three or none; order as in ian_layer_set_params; += when accumulate)
                                                                  from  X^T . Z   -- ONE dense backward-weight GEMM, then the
                                                                                     (W, coefficient) gradients as in
                                                                                     ian_layer_backward_weight.
   Needs n*H*W*round_up(6*taps,32) floats of scratch, owned by the first layer.  x may be NULL when no dparams are given.
   Returns -4 when the shape does not qualify (the caller falls back to the per-layer calls). */
int ian_layer_head6_backward(ian_layer* l0, ian_layer* l1, ian_layer* l2, const float* x, const float* dy0, const float* dy1,
                             const float* dy2, int32_t n, int32_t dy_stride, float* dx, int32_t dx_stride, int32_t dx_accumulate,
                             float* const* dparams0, float* const* dparams1, float* const* dparams2, int32_t nparams,
                             int32_t accumulate, void* stream);
/* Time candidate (tile shape x split-K x K-loop schedule) decompositions of this layer's forward and backward-data
   launches for batch n on this device and keep the fastest (same contract as ian_autotune: results are identical for
   every choice up to float32 summation order).  scratch_a / scratch_b: device buffers of cap_floats floats each, filled
   by the caller with random values (zero-filled operands run at a higher clock and would bias the choice); they are
   used as input / output of the timed launches and are overwritten. */
int ian_layer_autotune(ian_layer* l, int32_t n, float* scratch_a, float* scratch_b, int64_t cap_floats, void* stream);
const char* ian_layer_last_error(ian_layer* l);
void ian_layer_destroy(ian_layer* l);

/* ---- element-wise / reduction ops (names follow the reference construct they implement) ------------------- */
const char* ian_k_last_error(void);
/* per-channel sums over NHWC rows, two-stage, in FLOAT64 (every float32 element is widened before it is squared / added; see
   kernels_train.hip "NUMERICS": the variance E[x^2]-E[x]^2 formed from these sums is better conditioned than Lasagne's float32
   two-pass input.var): workspace >= nchunks*2*C doubles, sums = double[2][C].
   mode 0: (sum x, sum x^2) of x.   mode 1: g = x*act'(a): (sum g, sum g*xhat), xhat=(y-mean)*inv_std.
   mode 2: g = x*act'(a): (sum g, -).                                   batch_norm / bias gradients (App. B.3)
   The rows are cut into nchunks equal chunks (rows % nchunks == 0 for the guarantee below), each summed in a fixed
   order, and the chunk partials meet in a pairwise tree over the chunk index (ian_k_tree_sum).  With power-of-two
   chunk counts the result over a whole minibatch equals, bit for bit, the tree over ranks of per-rank results
   computed with the same chunk size: the data-parallel batch statistics are those of the single-process step. */
int ian_k_colstats(int32_t mode, const float* x, const float* a, const float* y, const float* mean, const float* inv_std,
                   int64_t rows, int32_t C, int32_t stride, int32_t act, double* workspace, int32_t nchunks, double* sums,
                   void* stream);
/* out[i] = pairwise tree (T(lo,hi) = T(lo,lo+m) + T(lo+m,hi), m = largest power of two below hi-lo) over k < count
   of partial[k*width + i]: the second stage of ian_k_colstats, and the rank-ordered combine of all-gathered
   per-rank sums (SyncBN) */
int ian_k_tree_sum(const double* partial, int32_t count, int32_t width, double* out, void* stream);
/* batch statistics -> mean, inv_std = 1/sqrt(var+eps), scale = gamma*inv_std, shift = beta - mean*scale */
int ian_k_bn_make_affine(const double* sums, float count, float eps, const float* gamma, const float* beta, int32_t C,
                         float* mean, float* inv_std, float* scale, float* shift, void* stream);
/* running averages of one normalisation, in place: r <- keep*r + alpha*batch for mean and inv_std in one launch (lasagne
   BatchNormLayer alpha = 0.1); the same arithmetic as the fused ian_k_bn_stats_affine below, bit for bit -- what the data-parallel
   `exact` step calls after ian_k_bn_make_affine */
int ian_k_bn_running(float* run_mean, const float* mean, float* run_inv_std, const float* inv_std, int32_t C, float keep, float alpha,
                     void* stream);
/* Single-process forms of the two-stage statistics above (no collective between the stages): the same chunk sums and the same
   tree, finished in ONE launch.  Bit-identical to ian_k_colstats + ian_k_bn_make_affine (+ the running-average updates
   r = keep*r + alpha*batch of lasagne BatchNormLayer, alpha = 0.1; run_mean/run_inv_std NULL: none), resp. to
   ian_k_colstats(mode 1) + the two gradient accumulations dbeta (+)= s1, dgamma (+)= s2 (gbeta/ggamma NULL: none). */
int ian_k_bn_stats_affine(const float* y, int64_t rows, int32_t C, int32_t stride, double* workspace, int32_t nchunks, double* sums,
                          float count, float eps, const float* gamma, const float* beta, float* mean, float* inv_std, float* scale,
                          float* shift, float* run_mean, float* run_inv_std, float keep, float alpha, void* stream);
/* second stages alone, for partial sums that are already in `workspace` ([nchunks][2][C] doubles: ian_layer_stats_next): the tree +
   affine (+ running averages) of ian_k_bn_stats_affine, resp. the tree + dbeta / dgamma accumulation of ian_k_bn_bwd_stats */
int ian_k_bn_finish(const double* workspace, int32_t nchunks, int32_t C, double* sums, float count, float eps, const float* gamma,
                    const float* beta, float* mean, float* inv_std, float* scale, float* shift, float* run_mean, float* run_inv_std,
                    float keep, float alpha, void* stream);
int ian_k_bn_bwd_finish(const double* workspace, int32_t nchunks, int32_t C, double* sums, float* gbeta, int32_t acc_beta, float* ggamma,
                        int32_t acc_gamma, void* stream);
int ian_k_bn_bwd_stats(const float* dA, const float* a, const float* y, const float* mean, const float* inv_std, int64_t rows, int32_t C,
                       int32_t stride, int32_t act, double* workspace, int32_t nchunks, double* sums, float* gbeta, int32_t acc_beta,
                       float* ggamma, int32_t acc_gamma, void* stream);
/* y = act(x*scale + shift) per channel (scale/shift may be NULL) */
int ian_k_affine(const float* x, float* y, const float* scale, const float* shift, int64_t rows, int32_t C, int32_t stride,
                 int32_t act, void* stream);
/* backward of [batch_norm ->] nonlinearity: dy = scale*(g - s1/N - xhat*s2/N), g = dA*act'(a); sums NULL: dy = g */
int ian_k_bn_bwd(const float* dA, const float* a, const float* y, const float* mean, const float* inv_std,
                 const float* scale, const double* sums, float count, float* dy, int64_t rows, int32_t C, int32_t stride,
                 int32_t act, void* stream);
int ian_k_axpy(float alpha, const float* x, float* y, int64_t n, int32_t accumulate, void* stream);
/* y (+)= (float)(alpha * x) for the float64 column sums of ian_k_colstats (bias / dbeta / dgamma gradients: rounded once) */
int ian_k_axpy_f64(double alpha, const double* x, float* y, int64_t n, int32_t accumulate, void* stream);
int ian_k_gather(const float* src, const int32_t* map, float* dst, int64_t count, void* stream);
int ian_k_nchw_to_nhwc(const float* src, float* dst, int32_t n, int32_t hw, int32_t c, int32_t stride, void* stream);
int ian_k_nhwc_to_nchw(const float* src, int32_t stride, float* dst, int32_t n, int32_t hw, int32_t c, void* stream);
/* GlobalPoolLayer (IAN.py:209) */
int ian_k_globalpool(const float* x, float* y, int32_t n, int32_t hw, int32_t C, int32_t xs, int32_t ys, void* stream);
int ian_k_globalpool_bwd(const float* dy, float* dx, int32_t n, int32_t hw, int32_t C, int32_t xs, int32_t ys,
                         int32_t accumulate, void* stream);
/* MinibatchLayer (layers.py:486-524): weight normalisation (:494), pairwise-L1 kernel features (:507-520) */
int ian_k_mb_weight(const float* theta, const float* lws, float* W, float* colscale, int32_t nin, int32_t ncol, void* stream);
int ian_k_mb_weight_bwd(const float* theta, const float* colscale, const float* dW, float* dtheta, float* dlws, int32_t nin,
                        int32_t ncol, int32_t accumulate, void* stream);
int ian_k_mb_forward(const float* act_all, int32_t nall, int32_t as, int32_t row0, int32_t n, int32_t nk, int32_t nd,
                     const float* bias, const float* feat, int32_t fs, int32_t fin, float* mb, int32_t ms, void* stream);
int ian_k_mb_backward(const float* act_all, int32_t nall, int32_t as, int32_t row0, int32_t n, int32_t nk, int32_t nd,
                      const float* df_all, int32_t dfs, float* dact, int32_t das, void* stream);
/* `discrimi` DenseLayer(3, softmax, b=None) + categorical cross-entropy (IAN.py:210-216, train_IAN.py:228-250) */
int ian_k_disc_head(const float* mb, int32_t ms, int32_t nfeat, const float* Wd, int32_t n, int32_t target0, int32_t target1,
                    int32_t acc_target, float* p, float* loss, void* stream);
int ian_k_disc_head_bwd(const float* p, const float* Wd, int32_t nfeat, int32_t n, int32_t t0, float w0, int32_t t1, float w1,
                        float* dlogits, float* dmb, int32_t ms, void* stream);
int ian_k_disc_head_wgrad(const float* mb, int32_t ms, int32_t nfeat, int32_t n, const float* dlogits, float* dWd,
                          int32_t accumulate, void* stream);
/* GaussianSampleLayer (layers.py:419-433) + KL terms (train_IAN.py:172) */
int ian_k_sample(const float* mu, const float* ls, const float* eps, float* z0, float* klterm, int32_t n, int32_t d,
                 int32_t stride, int32_t eps_stride, void* stream);
int ian_k_sample_bwd(const float* mu, const float* ls, const float* eps, const float* dz0, float* dmu, float* dls, int32_t n,
                     int32_t d, int32_t stride, int32_t eps_stride, float klw, void* stream);
/* MADE x2 + IAFLayer (layers.py:641-650,735-853); wts = 6 pre-masked (in,out) matrices, bias = 6 vectors, order
   mu_input, mu_output_W, mu_output_D, ls_input, ls_output_W, ls_output_D */
int ian_k_made_iaf(const float* z0, float* z, const float* wts, const float* bias, int32_t n, int32_t d, int32_t zs,
                   void* stream);
int ian_k_made_iaf_bwd(const float* z0, const float* dz, float* dz0, const float* wts, const float* bias, int32_t n,
                       int32_t d, int32_t zs, void* stream);
/* beta_layer x3 + concat (layers.py:397-408, IAN.py:207) and ConcatLayer (IAN.py:201), forward and backward */
int ian_k_beta(const float* R, const float* G, const float* B, float* y_nchw, int32_t n, int32_t hw, int32_t rs, void* stream);
int ian_k_beta_bwd(const float* gout_nchw, const float* R, const float* G, const float* B, float* gR, float* gG, float* gB,
                   int32_t n, int32_t hw, int32_t rs, int32_t act, void* stream);
int ian_k_concat2(const float* a, int32_t ca, int32_t sa, const float* b, int32_t cb, int32_t sb, float* y, int32_t sy,
                  int64_t npix, void* stream);
/* gd[p,c] (+)= gs[p,coff+c] * act'(y[p,c])   (identity edges: concat / residual backward, sigmoid backward) */
int ian_k_grad_pass(const float* gs, int32_t ss, int32_t coff, float* gd, const float* y, int32_t ds, int64_t npix,
                    int32_t C, int32_t act, int32_t accumulate, void* stream);
/* mode 0: pixel_loss 2|a-b+1e-8| (+ squared error), mode 1: squared error; da (+)= w * d/da; out[2] = scale * sums */
int ian_k_pair_loss(const float* a, const float* b, float* da, int64_t rows, int32_t C, int32_t stride, int32_t mode, float w,
                    int32_t accumulate, float* workspace, int32_t nblocks, float scale, float* out, void* stream);
int ian_k_sum_rows(const float* x, int32_t n, int32_t width, float scale, float* out, void* stream);
/* orthogonal regulariser (train_IAN.py:158-165) on W (A,B,K,K): vals[a] = sum|y_a|, dW += c * d/dW (dW may be NULL) */
int ian_k_ortho(const float* W, float* dW, int32_t A, int32_t B, int32_t K, float c, float* vals, void* stream);
/* lasagne.updates.adam (App. B.7) on a flat group */
int ian_k_adam(float* p, const float* g, float* m, float* v, int64_t n, float a_t, float b1, float b2, float eps, void* stream);

/* ---- the whole step behind one entry (ian_train_step) ----------------------------------------------------
 * ian_trainer owns the full IAN graph of IAN.py:67-228 in training mode: the three Adam groups of train_IAN.py:184-194
 * (encoder_params, Z_params, decoder_params; reference layouts and Theano names), every activation of the three passes of
 * train_IAN.py:116-149, the batch-norm running averages and the frozen MADE parameters.  csrc/ian_trainer.cpp wires it
 * from the entry points above and nothing else, and it is the ONLY sequencer of the step: one GPU, or one rank of a
 * data-parallel job whose collectives arrive through ian_comm_ops (below). */
typedef struct ian_trainer ian_trainer;
typedef struct ian_train_config {
  int32_t batch;         /* images per update: batch statistics, the MinibatchLayer and the loss means are over exactly this many */
  int32_t num_latents;   /* cfg['num_latents'] = 100 (IAN.py:52) */
  int32_t deconv_flip;   /* SURVEY App. B.2 */
  int32_t reserved;      /* keeps the doubles 8-byte aligned in every ABI */
  double learning_rate;  /* cfg['learning_rate'][0] = 2e-4 (IAN.py:38); later epochs: ian_trainer_set_option("learning_rate").  double: Adam's
                            a_t = lr*sqrt(1-b2^t)/(1-b1^t) is formed in double and rounded once, as lasagne.updates.adam's host arithmetic */
  double beta1;          /* 0.5 (IAN.py:42) */
  float reg;             /* cfg['reg'] = 1e-5: L2 on Z_params (train_IAN.py:211-213) */
  float ortho;           /* cfg['ortho'] = 1e-3 (train_IAN.py:214-221); negative: no orthogonal penalty */
  float recon_weight, feature_weight, dg_weight, dd_weight, agr_weight, ags_weight;   /* IAN.py:53-58 */
} ian_train_config;
int ian_trainer_create(const ian_train_config* cfg, ian_trainer** out);
/* Data parallel (train_IAN.py has none; north_star: minibatch sharded over the GPUs of one node, RCCL gradient all-reduce
   overlapped with backward).  One process per GPU; cfg.batch is the PER-RANK batch; every loss is a mean over the GLOBAL batch
   so that summing per-rank gradients gives the 1-GPU gradient.  The host supplies the collectives on DEVICE buffers and HIP
   streams -- neural_photo_editor_amd/trainer.py fills the table from torch.distributed (backend "nccl" = RCCL over xGMI; gloo
   in the tests); a C caller would fill it from librccl (ncclAllReduce / ncclAllGather on the given stream):
     allreduce_sum  buf[0..count) <- sum over ranks, in place.  Ordered after everything enqueued on `stream` at the time of the
                    call; may complete asynchronously (the trainer calls it on its own side stream while backward goes on).
     wait_all       make `stream` wait (device side where the backend can, else the host) for every allreduce_sum issued since
                    the previous wait_all.
     allgather      dst[r*count .. (r+1)*count) <- rank r's src[0..count); ordered on `stream`: work enqueued on `stream`
                    afterwards sees dst.
   Each returns 0 or a non-zero error code (the step then fails with -30).  exact != 0: batch-norm statistics are combined over
   all ranks in rank order (SyncBN, bitwise the 1-GPU statistics for power-of-two shards) and the MinibatchLayer sees the whole
   global minibatch (layers.py:506-524) -- the N-GPU step is then the same function of the global minibatch as the reference's
   step; exact == 0: local statistics (faster, NOT the reference's arithmetic).  Call BEFORE ian_trainer_finalize. */
typedef struct ian_comm_ops {
  int32_t world, rank;
  void* ctx;
  int (*allreduce_sum)(void* ctx, float* buf, int64_t count, void* stream);
  int (*wait_all)(void* ctx, void* stream);
  int (*allgather)(void* ctx, const float* src, float* dst, int64_t count, void* stream);
} ian_comm_ops;
int ian_trainer_set_comm(ian_trainer* t, const ian_comm_ops* ops, int32_t exact);
/* The table filled from librccl directly (csrc/ian_comm_rccl.cpp; librccl is dlopen'ed at the first call, libian.so does not link
   it): the torch-free route for a C / C++ caller, one process per GPU.  Rank 0 obtains the 128-byte communicator id and hands it
   to every rank out of band; every rank (after hipSetDevice) joins with the same bytes -- the call blocks until all `world` ranks
   have arrived -- and passes the filled table to ian_trainer_set_comm.  allreduce_sum = ncclAllReduce (in place, float32, sum) on
   the stream the trainer hands over, wait_all = one event per distinct stream an all-reduce was issued on since the last wait,
   which the compute stream waits for on the device, allgather = ncclAllGather.  ian_rccl_comm_add_gather (optional, a second id
   from ian_rccl_unique_id, every rank, blocks like comm_create) gives the all-gathers their OWN communicator: one communicator
   runs its collectives in issue order whatever streams they are on, so without it a batch-statistics all-gather the compute
   stream blocks on queues behind every gradient bucket already handed over.  0 / negative (-10: librccl could not be loaded),
   text via ian_rccl_last_error.
   CONTRACT for callers that add the second communicator (two communicators on one device are only deadlock-free under it):
   (1) every rank issues the SAME sequence of collectives on each communicator -- the trainer guarantees that when every rank
   passes the same configuration, batch, `which` sequence and `exact` flag (the bucket plan and the all-gather sites depend on
   nothing else; the loss all-reduce is unconditional, see ian_train_step); (2) the all-gather kernels (compute stream) and the
   bucket all-reduce kernels (side stream) must be able to be co-resident on every GPU: RCCL's kernels are small persistent
   grids, and the trainer never launches a cooperative or grid-barrier kernel of its own, so a tap-GEMM occupying the chip only
   delays them.  A job that cannot guarantee (1)/(2) -- or that has hung once -- runs WITHOUT ian_rccl_comm_add_gather: one
   communicator serialises everything in issue order (slower by the queueing described above, never deadlocks on ordering);
   trainer.NativeRcclComm(one_comm=True) / IAN_RCCL_ONE_COMM=1 select that mode and bench.py falls back to it by itself.
   ian_rccl_available: 0 when librccl can be loaded and exports the seven entry points (no communicator is touched, never
   blocks) -- call it on every rank and agree on the outcome BEFORE any rank enters the blocking ian_rccl_comm_create. */
int ian_rccl_available(void);
int ian_rccl_unique_id(void* out128);
int ian_rccl_comm_create(const void* id128, int32_t rank, int32_t world, ian_comm_ops* ops);
int ian_rccl_comm_add_gather(ian_comm_ops* ops, const void* id128);
void ian_rccl_comm_destroy(ian_comm_ops* ops);
const char* ian_rccl_last_error(void);
/* GANcheckpoints.py:33-57: one call per npz entry, Theano parameter names (trainable parameters, "<bn>.mean|inv_std",
   "l_IAF_{mu,ls}_{input,output_W,output_D}.{W,b}"); host pointer. */
int ian_trainer_load_param(ian_trainer* t, const char* name, const float* data, int64_t numel);
/* layers.py:831-853 MADE.shuffle("Once") result (train_IAN.py:404-405): 0/1 masks, (in,out) row-major */
int ian_trainer_set_made_masks(ian_trainer* t, const float* m0, const float* m1, const float* md, int32_t n);
int ian_trainer_finalize(ian_trainer* t);
/* train_IAN.py:309-329.  which 0 = update_gen, 1 = update_discrim (the loop of :497-504 alternates them).  x (n,3,64,64) in
   [-1,1], zrand (n,100) ~ N(0,1) (:478), eps (n,100) = the GaussianSampleLayer draw (layers.py:433): host or device
   pointers.  metrics: NULL, or 9 HOST floats of THIS minibatch before the update = discrim_d_loss, gen_recon_loss,
   gen_sample_loss, discrim_g_loss, discrim_acc, kl_div, pixel_loss, pixel_acc, feature_loss (train_IAN.py:291-304; reading
   them synchronises `stream`).  Data parallel: the 64-float all-reduce of the loss partials is issued on EVERY step whether or
   not `metrics` is given, so ranks may disagree on that argument without their collective sequences diverging.
   0 / negative, text via ian_trainer_last_error. */
int ian_train_step(ian_trainer* t, int32_t which, const float* x, const float* zrand, const float* eps, int32_t n, float* metrics,
                   void* stream);
/* The same step in pieces, for tests and diagnostics (ian_train_step == forward, [metrics], backward, finish_allreduce,
   regularizers, apply_adam on one stream).  xhat_override / xgen_override (device, (n,3,64,64), or NULL): images fed to the
   encoder passes on X_hat / X_gen instead of the decoder outputs (the decoders still run) -- the discriminator's |a_b - a_b'|
   kernels and the leaky-ReLU kinks make the gradients discontinuous in the activations, so a parity test feeds both
   implementations the SAME images. */
int ian_trainer_forward(ian_trainer* t, const float* x, const float* zrand, const float* eps, int32_t n, const float* xhat_override,
                        const float* xgen_override, void* stream);
int ian_trainer_metrics(ian_trainer* t, float* metrics);
int ian_trainer_backward(ian_trainer* t, int32_t which);
int ian_trainer_finish_allreduce(ian_trainer* t, int32_t which);
int ian_trainer_regularizers(ian_trainer* t, int32_t which);
int ian_trainer_apply_adam(ian_trainer* t, int32_t which);
/* one encoder backward sweep: pass 0 = encoder(X), 1 = encoder(X_hat), 2 = encoder(X_gen); cross-entropy seeds
   dlogits = w0 (p - onehot(t0)) + w1 (p - onehot(t1)) (t < 0: none); reset != 0 starts a fresh gradient sweep first */
int ian_trainer_enc_backward(ian_trainer* t, int32_t pass, int32_t t0, float w0, int32_t t1, float w1, int32_t feature_seeded,
                             int32_t want_w, int32_t want_dx, int32_t reset);
/* device address of an internal buffer: "<pass>.<name>", pass in EX EH EG ZS DZ DG (csrc/ian_trainer.cpp *_alloc),
   "<pass>.<bn>.<mean|inv_std|scale|shift>" (float32 [C]) or ".<sums|bsums>" (float64 [2][C]), "scalars", "ws_loss" */
int ian_trainer_buffer(ian_trainer* t, const char* name, void** ptr, int64_t* numel);
/* flat device buffers of group 0 encoder_params, 1 Z_params, 2 decoder_params (p, gradient, Adam m, v) or 3 = batch-norm
   running averages (p only); reference layouts, offsets via ian_trainer_param_info */
int ian_trainer_group(ian_trainer* t, int32_t group, float** p, float** g, float** m, float** v, int64_t* numel);
int ian_trainer_param_info(ian_trainer* t, const char* name, int32_t* group, int64_t* offset, int64_t* numel);
/* parameters of `group` were written behind the trainer's back: repack before the next forward */
int ian_trainer_mark_dirty(ian_trainer* t, int32_t group);
/* option measure_exposed: "exposed_ms_gen|discrim" = mean stall of the compute stream at wait_all per update (the part of the
   gradient all-reduce backward did not hide -- the wait_all stall ONLY), "gather_ms_gen|discrim" = mean time per update the
   compute stream spends inside the exact-mode all-gathers (batch statistics, MinibatchLayer; includes any queueing behind
   gradient buckets when the backend runs both on one communicator), "gathers_gen|discrim" = their number per update;
   "plan_buckets_gen|discrim", "overlap_log", "world", "rank", "exact", "global_batch" */
int ian_trainer_stat(ian_trainer* t, const char* key, double* out);
/* rec[6] = which, group, first element, bytes, gradient write after which the bucket was handed over, writes of that sweep */
int ian_trainer_overlap_log(ian_trainer* t, int32_t index, int64_t* rec);
/* per-layer (tile shape x split-K x K-loop schedule) choice for this batch on this GPU, as ian_layer_autotune */
int ian_trainer_autotune(ian_trainer* t, void* stream);
/* copy a parameter / running average (grad = 0) or its gradient of the last step (grad = 1) to the host (checkpoints:
   train_IAN.py:563-569; tests) */
int ian_trainer_read_param(ian_trainer* t, const char* name, int32_t grad, float* out, int64_t numel);
/* "overlap" (default 1: gradient buckets are handed to allreduce_sum while backward still runs), "bucket_bytes" (16 MB),
   "measure_exposed";
   "learning_rate" (schedule, train_IAN.py:523-527), "head6", "update_running"; "overlap_wgrad" (default 1): weight-gradient
   GEMMs go to a second HIP stream owned by the trainer and are joined before the regularizers and Adam -- same launches, same
   numbers, bitwise (tests/test_gpu_train_step.py); 0 keeps everything on the caller's stream */
int ian_trainer_set_option(ian_trainer* t, const char* key, double value);
/* Adam step counter of group 0 = encoder_params, 1 = Z_params (stepped by BOTH updates, train_IAN.py:274-276), 2 = decoder_params */
int32_t ian_trainer_adam_steps(ian_trainer* t, int32_t group);
const char* ian_trainer_last_error(ian_trainer* t);
void ian_trainer_destroy(ian_trainer* t);

#ifdef __cplusplus
}
#endif
#endif /* IAN_TRAIN_H_ */

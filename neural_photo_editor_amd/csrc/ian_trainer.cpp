// ian_trainer: the train_IAN.py step (make_training_functions, train_IAN.py:47-352: update_gen / update_discrim on the
// full IAN of IAN.py:67-228) behind ONE C entry, ian_train_step.  A C / C++ caller owns nothing but host or
// device buffers for the minibatch; the three Adam groups, every activation, the batch-norm running averages and the frozen
// MADE parameters live inside the object.
//
// This is the ONE sequencer of the training step (round 4: neural_photo_editor_amd/trainer.py is a thin ctypes caller of it,
// no second wiring exists).  It uses only the public C ABI of include/ian_train.h plus hipMalloc / hipMemcpy / streams / events.
//
// Data parallel (train_IAN.py has none): one process per GPU, the minibatch is sharded, losses are means over the
// GLOBAL batch.  The collectives arrive through a small callback table (ian_comm_ops: allreduce_sum / wait_all / allgather on
// device buffers and HIP streams) that the host fills -- from torch.distributed (backend "nccl" = RCCL over xGMI; gloo in the
// tests) in neural_photo_editor_amd/trainer.py -- so the same entry, ian_train_step, runs at world size 1 and N:
//   * gradient all-reduce OVERLAPPED with backward: a group's flat gradient buffer is cut into buckets (16 MB); the first
//     sweep of each update kind records the order of gradient writes; from then on a bucket is handed to allreduce_sum on a
//     side stream the moment its LAST writer kernel has been issued (events on the compute stream and on the weight-gradient
//     stream), while the compute stream goes on; the compute stream waits (wait_all) only before the regularisers and Adam;
//   * exact = 1 (SyncBN + MinibatchLayer all-gather): the float64 batch-statistics sums of every normalisation are
//     all-gathered and combined in RANK ORDER by the same pairwise tree the per-rank reduction uses (bitwise the
//     single-process statistics for power-of-two shards), the MinibatchLayer activations and their gradients are
//     all-gathered (layers.py:506-524 couples the whole minibatch), which makes the N-GPU step the same function of the
//     global minibatch as the reference's 1-GPU step;  exact = 0: local statistics, NOT the reference's arithmetic.
//
// Graph (train_IAN.py:116-149): encoder(X) -> z ~ N(mu, e^ls) -> IAF -> decoder -> X_hat; encoder(X_hat);
// decoder(IAF(Z)) -> X_gen; encoder(X_gen); every pass in batch-statistics batch-norm mode.
// Updates (train_IAN.py:253-276): three Adam groups -- encoder_params (update_discrim), decoder_params (update_gen),
// Z_params (both, ONE Adam instance).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <deque>
#include <map>
#include <set>
#include <string>
#include <vector>

#include "../../include/ian_train.h"
#include "ian_guard.h"   // IAN_SANITIZE builds: guard bands around every device allocation (no-op otherwise)

namespace {

constexpr float BN_EPS = 1e-4f;
constexpr int ENC_WIDTHS[4] = {128, 256, 512, 1024};
struct DecStage {
  const char* dc;
  int ci, co, hw;
  const char* blk;
  std::vector<int> scales;
};
const std::vector<DecStage>& dec_stages() {  // (deconv, cin, cout, in_hw, block, scales) IAN.py:139-171
  static const std::vector<DecStage> s = {{"dec_conv1", 512, 512, 4, "dec_conv2a", {0, 2}},
                                          {"dec_conv2", 512, 256, 8, "dec_conv3a", {0, 2, 3}},
                                          {"dec_conv3", 256, 128, 16, "dec_conv4a", {0, 2, 3}}};
  return s;
}
const std::vector<int> HEAD_SCALES = {2, 3, 4};
inline int cs(int c) { return (c + 31) / 32 * 32; }

std::vector<std::string> mdcl_names(const std::string& name, const std::vector<int>& scales) {
  std::vector<std::string> v = {name + "W", name + "_coeff_base"};
  for (int s : scales) v.push_back(name + (s == 0 ? std::string("_coeff_1x1") : "_coeff_" + std::to_string(s)));
  return v;
}

struct Shape {
  std::vector<int64_t> d;
  int64_t numel() const {
    int64_t n = 1;
    for (auto v : d) n *= v;
    return n;
  }
};

struct Group {  // one Adam instance: flat device buffers in the reference (Theano) layouts
  std::vector<std::string> names;
  std::map<std::string, std::pair<int64_t, Shape>> off;  // name -> (offset, shape)
  int64_t numel = 0;
  float *p = nullptr, *g = nullptr, *m = nullptr, *v = nullptr;
  int t = 0;
};

struct BN {  // Lasagne batch_norm in training mode: state of one normalisation in one pass
  int C = 0;
  double *sums = nullptr, *bsums = nullptr;   // float64 column sums (kernels_train.hip NUMERICS)
  float *mean = nullptr, *inv_std = nullptr, *scale = nullptr, *shift = nullptr;
  float count = 1.f;
};

struct LayerRef {
  ian_layer* l = nullptr;
  std::vector<std::string> pnames;  // empty: weights are set elsewhere (the MinibatchLayer's normalised theta)
};

typedef std::map<std::string, float*> Bufs;

struct Bucket {  // a slice of one group's flat gradient buffer = one all-reduce message
  int group;     // 0 encoder_params, 1 Z_params, 2 decoder_params
  int64_t lo, hi;
  std::set<std::string> names;
  int ready;     // index (1-based) of the last gradient write that lands in [lo, hi); 0: nothing writes it
  bool fired;
};
struct OverlapRec { int which, group; int64_t lo, bytes; int issued_at_write, writes_in_backward; };

}  // namespace

struct ian_trainer {
  ian_train_config cfg;
  int n = 0;
  std::string err;
  bool finalized = false;
  std::map<std::string, Shape> shapes;             // every parameter the graph owns
  std::map<std::string, std::vector<float>> host;  // values loaded before finalize
  Group enc, zp, dec, stats;                       // stats: batch-norm running averages (not trainable)
  std::map<std::string, Group*> where;
  std::map<std::string, LayerRef> layers;
  std::vector<std::string> layer_order;
  std::vector<float*> allocs;
  std::map<const void*, size_t> alloc_floats;      // size of every allocation (ian_trainer_buffer)
  std::vector<float> masks[3];
  int made_n = 0;
  float *made_w = nullptr, *made_b = nullptr;
  int32_t *fc2_perm = nullptr, *fc2_inv = nullptr;
  float *fc2_bias = nullptr, *fc2_db = nullptr, *fc2_tmp = nullptr;
  double *ws_stats = nullptr, *tmp_vals = nullptr, *tmp_big = nullptr, *gbuf = nullptr;   // float64 statistics workspaces
  float *ws_loss = nullptr, *scalars = nullptr, *mb_W = nullptr, *mb_dW = nullptr, *mb_colscale = nullptr, *ortho_vals = nullptr;
  size_t ws_stats_cap = 0, gbuf_cap = 0;
  bool oom = false;                                 // a device allocation failed (checked at the end of finalize)
  std::set<std::string> dirty, touched;
  Bufs EX, EH, EG, ZS, DZ, DG;
  std::map<std::string, BN> bnEX, bnEH, bnEG, bnZ, bnDZ, bnDG;
  float *zgen = nullptr, *zgen0 = nullptr, *xin = nullptr, *zin = nullptr, *epsin = nullptr;
  const float* X = nullptr;
  const float* eps = nullptr;
  int head6 = 1, update_running = 1;
  // batch statistics in the epilogue of the GEMM that produces the normalised tensor (ian_layer_stats_next) instead of a colstats
  // pass that re-reads it; single-process step only: the data-parallel `exact` step keeps the per-image chunks of ian_k_colstats,
  // whose partition does not depend on the tile shape the autotuner picks per batch size (bitwise rank-order invariance).
  // OFF, and effective in libian_ablation.so only (the product's GEMMs answer every request with 0 chunks): measured on MI355X at
  // 128 images, in-process A/B, 103.9 vs 99.7 ms per G+D pair -- the colstats passes are HBM-bound kernels that already run under
  // the weight-gradient GEMMs of the second stream; folded into the epilogue the same bytes sit on the MFMA-bound critical path
  int fused_stats = 0;
  hipStream_t st = nullptr;
  // weight-gradient GEMMs (ian_layer_backward_weight: 19 % of the step) are off the critical path -- nothing reads a gradient
  // before the regularisers -- so they are issued on a second stream behind an event on the compute stream (their dy operand
  // is final at that point and is not written again in the same sweep) and overlap the element-wise kernels of the
  // backward-data chain; the compute stream joins before the regularisers.  Same launches, same per-parameter order.
  int overlap_wgrad = 1;
  int wgrad_priority = 0;                           // weight-gradient stream created with the lowest stream priority (make_wgrad_stream)
  hipStream_t st2 = nullptr;
  // TIMING-ONLY experiment (IAN_DUAL_CHAIN_TIMING=1, libian_ablation.so only; RESULTS ARE WRONG): the (Z_rand -> decoder -> encoder) chain of
  // forward and backward on its own stream next to the (X -> encoder -> decoder -> encoder) chain, ignoring that both chains use the SAME
  // layer objects' split-K / statistics workspaces and accumulate into the same gradient buffers -- what a real two-chain step (per-chain
  // layer clones, private gradient accumulators) could gain at most
  int dual_timing = 0;
  hipStream_t stB = nullptr;
  std::vector<hipEvent_t> events;
  size_t ev_used = 0;
  hipStream_t last_stream = nullptr;               // stream of the previous entry: a different one is synchronised first
  bool have_last_stream = false;
  // ---- data parallel ------------------------------------------------------------------------------------------------
  ian_comm_ops comm;
  int world = 1, rank = 0, exact = 0, N = 0;        // N = global batch (losses are means over it)
  int overlap = 1;                                  // hand gradient buckets to the all-reduce while backward still runs
  int64_t bucket_bytes = 16 << 20;                  // xGMI is point-to-point: a few large messages
  hipStream_t st_comm = nullptr;                    // side stream the buckets are handed over on
  std::map<int, std::vector<Bucket>> plans;         // per update kind (0 gen, 1 discrim)
  std::map<int, int> plan_key;
  std::vector<Bucket>* buckets = nullptr;           // the plan this sweep follows (nullptr: record, reduce at the end)
  std::vector<std::vector<std::string>> evlog;      // gradient writes of this sweep, in issue order
  int which_now = 0, nfired = 0;
  std::deque<OverlapRec> overlap_log;               // last few sweeps (tests/test_gpu_dp.py, bench.py)
  int measure_exposed = 0;
  hipEvent_t ex0 = nullptr, ex1 = nullptr;
  bool ex_pending = false;
  int ex_which = 0;
  double exposed_ms[2] = {0, 0};
  int exposed_n[2] = {0, 0};
  // measure_exposed: time the compute stream spends inside the exact-mode all-gathers (event pairs around every comm.allgather).
  // Two pools used by alternate steps: a pool is folded (its events completed a whole step ago) right before it is reused.
  std::vector<std::pair<hipEvent_t, hipEvent_t>> gev[2];
  size_t gev_used[2] = {0, 0};
  int gev_which[2] = {0, 0}, gpar = 0;
  double gather_ms[2] = {0, 0}, gather_calls[2] = {0, 0};
  int gather_n[2] = {0, 0};
  const float *xhat_override = nullptr, *xgen_override = nullptr;   // test hook of ian_trainer_forward
};

namespace {

int tfail(ian_trainer* t, int code, const char* fmt, ...) {
  char buf[768];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (t) t->err = buf;
  return code;
}
#define TK(expr)                                                                                      \
  do {                                                                                                \
    const int rc_ = (expr);                                                                           \
    if (rc_) return tfail(t, rc_, "%s failed (%d): %s", #expr, rc_, ian_k_last_error() ? ian_k_last_error() : "?"); \
  } while (0)
#define TL(layer, expr)                                                                               \
  do {                                                                                                \
    const int rc_ = (expr);                                                                           \
    if (rc_) return tfail(t, rc_, "%s failed (%d): %s", #expr, rc_, ian_layer_last_error(layer) ? ian_layer_last_error(layer) : "?"); \
  } while (0)
#define THIP(expr)                                                                                    \
  do {                                                                                                \
    const hipError_t e_ = (expr);                                                                     \
    if (e_ != hipSuccess) return tfail(t, -20, "%s: %s", #expr, hipGetErrorString(e_));              \
  } while (0)

float* dalloc(ian_trainer* t, size_t floats) {   // zero-filled device floats; a failure sets t->oom (finalize reports it)
  float* p = nullptr;
  const size_t bytes = (floats ? floats : 1) * sizeof(float);
  if (hipMalloc((void**)&p, bytes) != hipSuccess) {
    (void)hipGetLastError();
    t->oom = true;
    return nullptr;
  }
  if (hipMemset(p, 0, bytes) != hipSuccess) {
    (void)hipGetLastError();
    t->oom = true;
  }
  t->allocs.push_back(p);
  t->alloc_floats[p] = floats;
  return p;
}
double* dalloc64(ian_trainer* t, size_t doubles) { return reinterpret_cast<double*>(dalloc(t, 2 * doubles)); }

void add_param(ian_trainer* t, Group& g, const std::string& name, std::vector<int64_t> shape) {
  Shape s{shape};
  t->shapes[name] = s;
  g.names.push_back(name);
  g.off[name] = {g.numel, s};
  g.numel += (s.numel() + 3) / 4 * 4;  // keep every tensor 16-byte aligned
  t->where[name] = &g;
}
float* P(ian_trainer* t, const std::string& n) {
  Group* g = t->where.at(n);
  return g->p + g->off.at(n).first;
}
float* G(ian_trainer* t, const std::string& n) {
  Group* g = t->where.at(n);
  return g->g + g->off.at(n).first;
}
int64_t numel_of(ian_trainer* t, const std::string& n) { return t->where.at(n)->off.at(n).second.numel(); }

void declare_parameters(ian_trainer* t) {
  const int Z = t->cfg.num_latents;
  add_param(t, t->enc, "enc_conv1.W", {128, 3, 5, 5});
  add_param(t, t->enc, "enc_conv1.b", {128});
  for (int i = 2; i <= 4; ++i) {
    const std::string s = std::to_string(i);
    add_param(t, t->enc, "enc_conv" + s + ".W", {ENC_WIDTHS[i - 1], ENC_WIDTHS[i - 2], 5, 5});
    add_param(t, t->enc, "bnorm" + s + ".beta", {ENC_WIDTHS[i - 1]});
    add_param(t, t->enc, "bnorm" + s + ".gamma", {ENC_WIDTHS[i - 1]});
  }
  add_param(t, t->enc, "minibatch_discrim.theta", {1024, 500, 5});
  add_param(t, t->enc, "minibatch_discrim.log_weight_scale", {500, 5});
  add_param(t, t->enc, "minibatch_discrim.b", {500});
  add_param(t, t->enc, "discrimi.W", {1524, 3});
  add_param(t, t->zp, "enc_fc1.W", {16384, 1000});
  add_param(t, t->zp, "bnorm_enc_fc1.beta", {1000});
  add_param(t, t->zp, "bnorm_enc_fc1.gamma", {1000});
  for (const char* nm : {"enc_mu", "enc_logsigma"}) {
    const std::string bn = std::string(nm) == "enc_mu" ? "mu_bnorm" : "ls_bnorm";
    add_param(t, t->zp, std::string(nm) + ".W", {1000, Z});
    add_param(t, t->zp, bn + ".beta", {Z});
    add_param(t, t->zp, bn + ".gamma", {Z});
  }
  add_param(t, t->dec, "l_dec_fc2.W", {Z, 8192});
  add_param(t, t->dec, "l_dec_fc2.b", {8192});
  auto add_mdcl = [&](const std::string& name, int co, int ci, const std::vector<int>& sc) {
    auto names = mdcl_names(name, sc);
    add_param(t, t->dec, names[0], {co, ci, 3, 3});
    for (size_t i = 1; i < names.size(); ++i) add_param(t, t->dec, names[i], {co});
  };
  for (const DecStage& s : dec_stages()) {
    const std::string blk = s.blk;
    add_param(t, t->dec, std::string(s.dc) + ".W", {s.ci, s.co, 5, 5});
    add_param(t, t->dec, blk + "bnorm0.beta", {s.co});
    add_param(t, t->dec, blk + "bnorm0.gamma", {s.co});
    add_mdcl(blk, s.co, s.co, s.scales);
    add_param(t, t->dec, blk + "bnorm1.beta", {s.co});
    add_param(t, t->dec, blk + "bnorm1.gamma", {s.co});
    add_mdcl(blk + "2", s.co, s.co, s.scales);
    add_param(t, t->dec, blk + "bnorm2.beta", {s.co});
    add_param(t, t->dec, blk + "bnorm2.gamma", {s.co});
  }
  add_param(t, t->dec, "dec_conv4.W", {128, 128, 5, 5});
  add_param(t, t->dec, "bnorm_dc4.beta", {128});
  add_param(t, t->dec, "bnorm_dc4.gamma", {128});
  const std::pair<const char*, int> heads[5] = {{"R", 128}, {"G_a", 128}, {"G_b", 2}, {"B_a", 128}, {"B_b", 4}};
  for (auto& h : heads) add_mdcl(h.first, 2, h.second, HEAD_SCALES);
  // batch-norm running averages (not trainable; Lasagne BatchNormLayer alpha = 0.1): what the deterministic graphs of
  // API.py / sample_IAN.py normalise with after training
  std::vector<std::pair<std::string, int>> bns = {{"bnorm2", 256}, {"bnorm3", 512}, {"bnorm4", 1024}, {"bnorm_enc_fc1", 1000},
                                                  {"mu_bnorm", Z}, {"ls_bnorm", Z}, {"bnorm_dc4", 128}};
  for (const DecStage& s : dec_stages())
    for (int j = 0; j < 3; ++j) bns.push_back({std::string(s.blk) + "bnorm" + std::to_string(j), s.co});
  for (auto& b : bns)
    for (const char* sfx : {".mean", ".inv_std"}) add_param(t, t->stats, b.first + sfx, {b.second});
  // frozen MADE parameters (never trained: train_IAN.py:184-194)
  for (const char* m : {"l_IAF_mu", "l_IAF_ls"})
    for (const char* l : {"_input", "_output_W", "_output_D"}) {
      t->shapes[std::string(m) + l + ".W"] = Shape{{Z, Z}};
      t->shapes[std::string(m) + l + ".b"] = Shape{{Z}};
    }
}

int make_layer(ian_trainer* t, const std::string& key, int kind, int cin, int cout, int in_h, int in_w, const std::vector<int>& scales,
               const int* flat, const int* unflat, std::vector<std::string> pnames) {
  ian_op_desc d;
  memset(&d, 0, sizeof d);
  d.kind = kind; d.cin = cin; d.cout = cout; d.in_h = in_h; d.in_w = in_w;
  d.src = d.dst = 0; d.src2 = d.src3 = -1;
  if (flat) { d.flat_c = flat[0]; d.flat_h = flat[1]; d.flat_w = flat[2]; }
  if (unflat) { d.unflat_c = unflat[0]; d.unflat_h = unflat[1]; d.unflat_w = unflat[2]; }
  d.n_scales = (int)scales.size();
  for (size_t i = 0; i < scales.size(); ++i) d.scales[i] = scales[i];
  LayerRef r;
  const int rc = ian_layer_create(&d, t->cfg.deconv_flip ? 1 : 0, &r.l);
  if (rc) return tfail(t, rc, "ian_layer_create(%s) failed (%d)%s", key.c_str(), rc, rc == -10 ? ": no HIP device, libian has no CPU fallback" : "");
  r.pnames = pnames;
  t->layers[key] = r;
  t->layer_order.push_back(key);
  return 0;
}

int build_layers(ian_trainer* t) {
  const int Z = t->cfg.num_latents;
  int cin = 3, rc;
  for (int i = 0; i < 4; ++i) {
    const std::string nm = "enc_conv" + std::to_string(i + 1);
    if ((rc = make_layer(t, nm, IAN_OP_CONV5S2, cin, ENC_WIDTHS[i], 64 >> i, 64 >> i, {}, nullptr, nullptr, {nm + ".W"}))) return rc;
    cin = ENC_WIDTHS[i];
  }
  const int flat[3] = {1024, 4, 4}, unflat[3] = {512, 4, 4};
  if ((rc = make_layer(t, "enc_fc1", IAN_OP_DENSE, 16384, 1000, 1, 1, {}, flat, nullptr, {"enc_fc1.W"}))) return rc;
  if ((rc = make_layer(t, "enc_mu", IAN_OP_DENSE, 1000, Z, 1, 1, {}, nullptr, nullptr, {"enc_mu.W"}))) return rc;
  if ((rc = make_layer(t, "enc_logsigma", IAN_OP_DENSE, 1000, Z, 1, 1, {}, nullptr, nullptr, {"enc_logsigma.W"}))) return rc;
  if ((rc = make_layer(t, "mb", IAN_OP_DENSE, 1024, 2500, 1, 1, {}, nullptr, nullptr, {}))) return rc;  // weights = normalised theta (layers.py:494)
  if ((rc = make_layer(t, "l_dec_fc2", IAN_OP_DENSE, Z, 8192, 1, 1, {}, nullptr, unflat, {"l_dec_fc2.W"}))) return rc;
  for (const DecStage& s : dec_stages()) {
    const std::string blk = s.blk;
    if ((rc = make_layer(t, s.dc, IAN_OP_DECONV5S2, s.ci, s.co, s.hw, s.hw, {}, nullptr, nullptr, {std::string(s.dc) + ".W"}))) return rc;
    if ((rc = make_layer(t, blk, IAN_OP_MDC3, s.co, s.co, 2 * s.hw, 2 * s.hw, s.scales, nullptr, nullptr, mdcl_names(blk, s.scales)))) return rc;
    if ((rc = make_layer(t, blk + "2", IAN_OP_MDC3, s.co, s.co, 2 * s.hw, 2 * s.hw, s.scales, nullptr, nullptr, mdcl_names(blk + "2", s.scales)))) return rc;
  }
  if ((rc = make_layer(t, "dec_conv4", IAN_OP_DECONV5S2, 128, 128, 32, 32, {}, nullptr, nullptr, {"dec_conv4.W"}))) return rc;
  const std::pair<const char*, int> heads[5] = {{"R", 128}, {"G_a", 128}, {"G_b", 2}, {"B_a", 128}, {"B_b", 4}};
  for (auto& h : heads)
    if ((rc = make_layer(t, h.first, IAN_OP_MDC3, h.second, 2, 64, 64, HEAD_SCALES, nullptr, nullptr, mdcl_names(h.first, HEAD_SCALES)))) return rc;
  return 0;
}
ian_layer* lay(ian_trainer* t, const std::string& n) { return t->layers.at(n).l; }

BN make_bn(ian_trainer* t, int C) {
  BN b;
  b.C = C;
  b.sums = dalloc64(t, 2 * C); b.bsums = dalloc64(t, 2 * C);
  b.mean = dalloc(t, C); b.inv_std = dalloc(t, C); b.scale = dalloc(t, C); b.shift = dalloc(t, C);
  return b;
}

void enc_alloc(ian_trainer* t, Bufs& E, std::map<std::string, BN>& bn) {
  const size_t n = t->n;
  E["x"] = dalloc(t, n * 64 * 64 * 32); E["dx"] = dalloc(t, n * 64 * 64 * 32);
  for (int i = 0; i < 4; ++i) {
    const int hw = 32 >> i, w = ENC_WIDTHS[i];
    const std::string s = std::to_string(i + 1);
    E["a" + s] = dalloc(t, n * hw * hw * w); E["da" + s] = dalloc(t, n * hw * hw * w);
    if (i > 0) {
      E["y" + s] = dalloc(t, n * hw * hw * w);
      bn["bn" + s] = make_bn(t, w);
    }
  }
  E["feat"] = dalloc(t, n * 1024); E["dfeat"] = dalloc(t, n * 1024);
  E["act"] = dalloc(t, n * cs(2500)); E["dact"] = dalloc(t, n * cs(2500));
  E["mb"] = dalloc(t, n * cs(1524)); E["dmb"] = dalloc(t, n * cs(1524));
  if (t->exact) {  // MinibatchLayer over the GLOBAL minibatch (layers.py:506-524): all ranks' activations / feature gradients
    E["act_all"] = dalloc(t, (size_t)t->N * cs(2500));
    E["dmb_all"] = dalloc(t, (size_t)t->N * cs(1524));
  }
  E["p"] = dalloc(t, n * 3); E["loss"] = dalloc(t, n * 4); E["dlogits"] = dalloc(t, n * 4);
}
void z_alloc(ian_trainer* t) {
  const size_t n = t->n;
  Bufs& Z = t->ZS;
  Z["y_fc1"] = dalloc(t, n * 1024); Z["f"] = dalloc(t, n * 1024); Z["df"] = dalloc(t, n * 1024);
  t->bnZ["bn_fc1"] = make_bn(t, 1000);
  for (const char* nm : {"mu", "ls"}) {
    Z[std::string("y_") + nm] = dalloc(t, n * 128); Z[nm] = dalloc(t, n * 128); Z[std::string("d") + nm] = dalloc(t, n * 128);
    t->bnZ[std::string("bn_") + nm] = make_bn(t, t->cfg.num_latents);
  }
  {  // the two latent normalisations read the same input in the same stage: their float64 sums lie side by side so that the
     // data-parallel `exact` step combines both with ONE all-gather per direction (z_forward / z_backward)
    const int Zd = t->cfg.num_latents;
    double *js = dalloc64(t, 4 * Zd), *jb = dalloc64(t, 4 * Zd);
    t->bnZ["bn_mu"].sums = js; t->bnZ["bn_ls"].sums = js ? js + 2 * Zd : nullptr;
    t->bnZ["bn_mu"].bsums = jb; t->bnZ["bn_ls"].bsums = jb ? jb + 2 * Zd : nullptr;
  }
  Z["z0"] = dalloc(t, n * 128); Z["z"] = dalloc(t, n * 128); Z["dz0"] = dalloc(t, n * 128); Z["kl"] = dalloc(t, n * 100);
}
void dec_alloc(ian_trainer* t, Bufs& D, std::map<std::string, BN>& bn) {
  const size_t n = t->n;
  D["h0"] = dalloc(t, n * 16 * 512); D["dh0"] = dalloc(t, n * 16 * 512);
  for (const DecStage& s : dec_stages()) {
    const size_t e = n * (2 * s.hw) * (2 * s.hw) * s.co;
    const std::string blk = s.blk;
    for (const char* nm : {"x", "a", "b", "c", "e", "h", "dx", "da", "dc", "dh"}) D[blk + "_" + nm] = dalloc(t, e);
    for (int j = 0; j < 3; ++j) bn[blk + "_bn" + std::to_string(j)] = make_bn(t, s.co);
  }
  D["y4"] = dalloc(t, n * 4096 * 128); D["h4"] = dalloc(t, n * 4096 * 128); D["dh4"] = dalloc(t, n * 4096 * 128);
  bn["bn4"] = make_bn(t, 128);
  for (const char* nm : {"R", "G", "B", "Ga", "Ba", "RG", "gR", "gG", "gB", "dRG", "dRt"}) D[nm] = dalloc(t, n * 4096 * 32);
  D["xhat"] = dalloc(t, n * 3 * 4096); D["dxhat"] = dalloc(t, n * 3 * 4096); D["tmp_img"] = dalloc(t, n * 3 * 4096);
  D["dz"] = dalloc(t, n * 128);
}

// ---- parameter refresh (after every optimiser update): reference layout -> kernel layouts -------------------------
int refresh_weights(ian_trainer* t) {
  if (t->dirty.empty()) return 0;
  auto gname = [&](const std::string& p) -> std::string {
    Group* g = t->where.at(p);
    return g == &t->enc ? "enc" : (g == &t->zp ? "Z" : "dec");
  };
  for (const std::string& key : t->layer_order) {
    LayerRef& r = t->layers[key];
    if (r.pnames.empty() || !t->dirty.count(gname(r.pnames[0]))) continue;
    std::vector<const float*> ptrs;
    for (auto& p : r.pnames) ptrs.push_back(P(t, p));
    TL(r.l, ian_layer_set_params(r.l, ptrs.data(), (int)ptrs.size(), t->st));
  }
  if (t->dirty.count("enc")) {
    TK(ian_k_mb_weight(P(t, "minibatch_discrim.theta"), P(t, "minibatch_discrim.log_weight_scale"), t->mb_W, t->mb_colscale, 1024, 2500, t->st));
    const float* w = t->mb_W;
    TL(lay(t, "mb"), ian_layer_set_params(lay(t, "mb"), &w, 1, t->st));
  }
  if (t->dirty.count("dec")) TK(ian_k_gather(P(t, "l_dec_fc2.b"), t->fc2_perm, t->fc2_bias, 8192, t->st));
  t->dirty.clear();
  return 0;
}

// ---- building blocks ------------------------------------------------------------------------------------------------
int chunks(const ian_trainer* t, int64_t rows) {
  // chunk SIZE depends on the per-image extent only (one image, or 512 rows of one): include/ian_train.h, ian_k_colstats
  const int64_t rpi = rows / t->n > 0 ? rows / t->n : 1;
  if (rows == (int64_t)t->n * rpi) return (int)(t->n * (rpi / 512 > 0 ? rpi / 512 : 1));
  return (int)(rows < 256 ? rows : 256);
}
int ws_for(ian_trainer* t, int64_t rows, int C, double** out) {
  const size_t need = (size_t)chunks(t, rows) * 2 * C;
  if (t->ws_stats_cap < need) {
    double* p = dalloc64(t, need);
    if (!p) return tfail(t, -20, "out of device memory (statistics workspace)");
    t->ws_stats = p;
    t->ws_stats_cap = need;
  }
  *out = t->ws_stats;
  return 0;
}
Group& group_of(ian_trainer* t, int g) { return g == 0 ? t->enc : (g == 1 ? t->zp : t->dec); }
int group_index(ian_trainer* t, const Group* g) { return g == &t->enc ? 0 : (g == &t->zp ? 1 : 2); }
int next_event(ian_trainer* t, hipEvent_t* e) {
  if (t->ev_used == t->events.size()) {
    hipEvent_t ne;
    THIP(hipEventCreateWithFlags(&ne, hipEventDisableTiming));
    t->events.push_back(ne);
  }
  *e = t->events[t->ev_used++];
  return 0;
}

// ---- gradient all-reduce overlapped with backward ----------------------------------------------------------
// encoder_params receive contributions from three encoder passes, decoder_params from two decoder passes, so the "last" writer
// of a bucket is a property of the whole backward sweep of an update kind: every gradient write is reported through mark();
// the first sweep of a kind records the order, later sweeps fire a bucket right after its last write.
int fire(ian_trainer* t, Bucket& b) {
  Group& g = group_of(t, b.group);
  hipEvent_t e;
  int rc;
  if ((rc = next_event(t, &e))) return rc;
  THIP(hipEventRecord(e, t->st));                  // after the bucket's last writer on the compute stream ...
  THIP(hipStreamWaitEvent(t->st_comm, e, 0));
  if (t->overlap_wgrad && t->st2) {                // ... and on the weight-gradient stream
    if ((rc = next_event(t, &e))) return rc;
    THIP(hipEventRecord(e, t->st2));
    THIP(hipStreamWaitEvent(t->st_comm, e, 0));
  }
  rc = t->comm.allreduce_sum(t->comm.ctx, g.g + b.lo, b.hi - b.lo, t->st_comm);
  if (rc) return tfail(t, -30, "comm.allreduce_sum failed (%d) on a gradient bucket of group %d", rc, b.group);
  b.fired = true;
  ++t->nfired;
  t->overlap_log.push_back({t->which_now, b.group, b.lo, 4 * (b.hi - b.lo), (int)t->evlog.size(), 0});
  while (t->overlap_log.size() > 256) t->overlap_log.pop_front();
  return 0;
}
int mark(ian_trainer* t, const std::vector<std::string>& names) {  // the gradients of `names` have just been written (issued)
  for (auto& n : names) t->touched.insert(n);
  t->evlog.push_back(names);
  if (!t->buckets) return 0;
  const int ev = (int)t->evlog.size();
  for (Bucket& b : *t->buckets) {
    if (b.fired) {
      for (auto& n : names)
        if (b.names.count(n)) {
          t->plans.erase(t->which_now);            // stale plan: the next sweep of this kind re-records it
          t->buckets = nullptr;
          return tfail(t, -31, "gradient of %s written after its bucket was handed to the all-reduce (write order changed since the "
                               "plan was recorded; plan dropped)", n.c_str());
        }
    } else if (b.ready == ev) {
      int rc = fire(t, b);
      if (rc) return rc;
    }
  }
  return 0;
}
std::vector<Bucket> make_plan(ian_trainer* t, int which) {  // buckets of the groups this update kind moves, with their last writes
  std::map<std::string, int> last;
  for (size_t i = 0; i < t->evlog.size(); ++i)
    for (auto& n : t->evlog[i]) last[n] = (int)i + 1;
  const int64_t step = t->bucket_bytes / 4 > 0 ? t->bucket_bytes / 4 : 1;
  std::vector<Bucket> plan;
  for (int gi : {which == 0 ? 2 : 0, 1}) {
    Group& g = group_of(t, gi);
    for (int64_t o = 0; o < g.numel; o += step) {
      Bucket b;
      b.group = gi; b.lo = o; b.hi = o + step < g.numel ? o + step : g.numel; b.ready = 0; b.fired = false;
      for (auto& nm : g.names) {
        const auto& of = g.off.at(nm);
        const int64_t po = of.first, cnt = of.second.numel();
        if (po < b.hi && po + cnt > b.lo) {
          b.names.insert(nm);
          auto it = last.find(nm);
          if (it != last.end() && it->second > b.ready) b.ready = it->second;
        }
      }
      plan.push_back(b);
    }
  }
  return plan;
}
int join_side_stream(ian_trainer* t);
int begin_backward(ian_trainer* t, int which) {
  // every event handed out so far has been consumed by a hipStreamWaitEvent already issued: the pool restarts here whether or not
  // the weight-gradient stream is in use (with overlap_wgrad = 0 nothing else reset it and a data-parallel run grew it by one
  // event per bucket per step)
  int jrc = join_side_stream(t);
  if (jrc) return jrc;
  t->ev_used = 0;
  t->gev_which[t->gpar] = which;
  t->touched.clear();
  t->evlog.clear();
  t->which_now = which;
  t->nfired = 0;
  t->buckets = nullptr;
  if (t->world == 1) return 0;
  // the recorded write order depends on these switches: a plan made under other settings is discarded
  const int key = (t->head6 ? 1 : 0) | (t->exact ? 2 : 0) | (t->update_running ? 4 : 0) | (t->overlap_wgrad ? 8 : 0);
  if (!t->plan_key.count(which) || t->plan_key[which] != key) {
    t->plans.erase(which);
    t->plan_key[which] = key;
  }
  if (!t->overlap || !t->plans.count(which)) return 0;
  t->buckets = &t->plans[which];
  for (Bucket& b : *t->buckets) b.fired = false;
  for (Bucket& b : *t->buckets)
    if (b.ready == 0) {
      int rc = fire(t, b);
      if (rc) return rc;
    }
  return 0;
}
int finish_allreduce(ian_trainer* t, int which) {  // after backward: reduce what has not been handed over yet, then the compute stream waits
  int rc;
  if ((rc = join_side_stream(t))) return rc;
  if (t->world == 1) return 0;
  if (!t->buckets) {                               // first sweep of this kind (or overlap off): plan, then reduce everything
    t->plans[which] = make_plan(t, which);
    t->buckets = &t->plans[which];
    for (Bucket& b : *t->buckets) b.fired = false;
  }
  for (Bucket& b : *t->buckets)
    if (!b.fired && (rc = fire(t, b))) return rc;
  const int writes = (int)t->evlog.size();
  for (size_t i = t->overlap_log.size() >= (size_t)t->nfired ? t->overlap_log.size() - t->nfired : 0; i < t->overlap_log.size(); ++i)
    t->overlap_log[i].writes_in_backward = writes;
  // how long the COMPUTE stream stalls on communication = the part of the all-reduce backward did not hide
  if (t->measure_exposed && t->ex_pending) {       // fold the previous measurement in (its events have completed long ago)
    float ms = 0.f;
    if (hipEventSynchronize(t->ex1) == hipSuccess && hipEventElapsedTime(&ms, t->ex0, t->ex1) == hipSuccess) {
      t->exposed_ms[t->ex_which] += ms;
      t->exposed_n[t->ex_which] += 1;
    }
    (void)hipGetLastError();
    t->ex_pending = false;
  }
  if (t->measure_exposed) {
    if (!t->ex0) { THIP(hipEventCreate(&t->ex0)); THIP(hipEventCreate(&t->ex1)); }
    THIP(hipEventRecord(t->ex0, t->st));
  }
  rc = t->comm.wait_all(t->comm.ctx, t->st);
  if (rc) return tfail(t, -30, "comm.wait_all failed (%d)", rc);
  if (t->measure_exposed) {
    THIP(hipEventRecord(t->ex1, t->st));
    t->ex_pending = true;
    t->ex_which = which;
  }
  t->buckets = nullptr;
  return 0;
}
// every exact-mode all-gather goes through here (batch statistics, MinibatchLayer activations / gradients): issued on the COMPUTE
// stream, whose next kernel reads the result.  measure_exposed: an event pair around it, folded a step later (fold_gathers).
int gather(ian_trainer* t, const float* src, float* dst, int64_t count, const char* what) {
  std::pair<hipEvent_t, hipEvent_t>* ev = nullptr;
  if (t->measure_exposed) {
    auto& pool = t->gev[t->gpar];
    if (t->gev_used[t->gpar] == pool.size()) {
      hipEvent_t a, b;
      THIP(hipEventCreate(&a));
      THIP(hipEventCreate(&b));
      pool.push_back({a, b});
    }
    ev = &pool[t->gev_used[t->gpar]++];
    THIP(hipEventRecord(ev->first, t->st));
  }
  const int rc = t->comm.allgather(t->comm.ctx, src, dst, count, t->st);
  if (rc) return tfail(t, -30, "comm.allgather failed (%d) on %s", rc, what);
  if (ev) THIP(hipEventRecord(ev->second, t->st));
  return 0;
}
void fold_gathers(ian_trainer* t, int par) {   // events of pool `par` completed long ago (a whole step), or the caller synchronised
  double ms_total = 0.0;
  const size_t n = t->gev_used[par];
  for (size_t i = 0; i < n; ++i) {
    float ms = 0.f;
    if (hipEventSynchronize(t->gev[par][i].second) == hipSuccess && hipEventElapsedTime(&ms, t->gev[par][i].first, t->gev[par][i].second) == hipSuccess)
      ms_total += ms;
  }
  (void)hipGetLastError();
  if (n) {
    const int w = t->gev_which[par];
    t->gather_ms[w] += ms_total;
    t->gather_calls[w] += (double)n;
    t->gather_n[w] += 1;
  }
  t->gev_used[par] = 0;
}

// sums (float64 [width]) <- sum over ranks, combined in RANK ORDER by the pairwise tree of ian_k_tree_sum: the result does not depend
// on the collective's internal algorithm and, for power-of-two shards, equals the single-process reduction bit for bit (SyncBN)
int allreduce_ordered(ian_trainer* t, double* sums, int width) {
  if (t->world == 1) return 0;
  const size_t need = (size_t)width * t->world;
  if (t->gbuf_cap < need) {
    double* p = dalloc64(t, need);
    if (!p) return tfail(t, -20, "out of device memory (statistics gather buffer)");
    t->gbuf = p;
    t->gbuf_cap = need;
  }
  const int rc = gather(t, reinterpret_cast<const float*>(sums), reinterpret_cast<float*>(t->gbuf), 2 * (int64_t)width, "batch statistics");
  if (rc) return rc;
  TK(ian_k_tree_sum(t->gbuf, t->world, width, sums, t->st));
  return 0;
}

int acc(ian_trainer* t, const std::string& pname, const float* src, int64_t count, float alpha = 1.f) {  // grad[pname] (+)= alpha * src
  TK(ian_k_axpy(alpha, src, G(t, pname), count, t->touched.count(pname) ? 1 : 0, t->st));
  return mark(t, {pname});
}
int acc64(ian_trainer* t, const std::string& pname, const double* src, int64_t count, double alpha = 1.0) {  // from float64 column sums
  TK(ian_k_axpy_f64(alpha, src, G(t, pname), count, t->touched.count(pname) ? 1 : 0, t->st));
  return mark(t, {pname});
}
// ---- GEMM-epilogue statistics (include/ian_train.h ian_layer_stats_next) ------------------------------------------------
bool fuse_stats(const ian_trainer* t) { return t->fused_stats && !t->exact; }
int arm_fwd_stats(ian_trainer* t, const std::string& lname) {      // the layer's next forward also sums v, v*v per row tile
  if (!fuse_stats(t)) return 0;
  TL(lay(t, lname), ian_layer_stats_next(lay(t, lname), 1, nullptr, nullptr, nullptr, nullptr, 0, t->ws_stats, (int64_t)t->ws_stats_cap));
  return 0;
}
int arm_bwd_stats(ian_trainer* t, const std::string& lname, const BN& bn, const float* a, const float* yraw, int act) {
  if (!fuse_stats(t)) return 0;   // the layer's next backward-data also sums g = dA act'(a) and g xhat of the gradient it stores
  TL(lay(t, lname), ian_layer_stats_next(lay(t, lname), 2, a, yraw, bn.mean, bn.inv_std, act, t->ws_stats, (int64_t)t->ws_stats_cap));
  return 0;
}
int armed_chunks(ian_trainer* t, const std::string& lname) { return fuse_stats(t) ? ian_layer_stats_chunks(lay(t, lname)) : 0; }

// exact mode, after the per-rank sums have been combined over the ranks: affine, apply, running averages (same running_math as the
// fused single-process second stage: data-parallel checkpoints carry bit-identical running averages)
int bn_forward_finish(ian_trainer* t, BN& bn, const float* y, float* a, int64_t rows, int C, int stride, const float* gamma, const float* beta,
                      int act, int64_t count_rows, float* rm, float* ri) {
  bn.count = (float)(count_rows * t->world);
  TK(ian_k_bn_make_affine(bn.sums, bn.count, BN_EPS, gamma, beta, C, bn.mean, bn.inv_std, bn.scale, bn.shift, t->st));
  TK(ian_k_affine(y, a, bn.scale, bn.shift, rows, C, stride, act, t->st));
  if (rm) TK(ian_k_bn_running(rm, bn.mean, ri, bn.inv_std, C, 0.9f, 0.1f, t->st));
  return 0;
}
// pre > 0: the producing GEMM already left `pre` chunk partials in the statistics workspace (arm_fwd_stats)
int bn_forward(ian_trainer* t, BN& bn, const float* y, float* a, int64_t rows, int C, int stride, const float* gamma, const float* beta, int act,
               int64_t count_rows, const char* running, int pre = 0) {
  double* ws = t->ws_stats;
  int rc;
  if (pre <= 0 && (rc = ws_for(t, rows, C, &ws))) return rc;   // (pre > 0: the partials are already in the workspace -- it must not move)
  float *rm = nullptr, *ri = nullptr;
  if (running && t->update_running) {  // r = (1 - alpha) r + alpha * batch   (Lasagne BatchNormLayer alpha = 0.1)
    rm = P(t, std::string(running) + ".mean");
    ri = P(t, std::string(running) + ".inv_std");
  }
  if (!t->exact) {  // no collective between the two stages: one fused second stage
    bn.count = (float)count_rows;
    if (pre > 0)
      TK(ian_k_bn_finish(t->ws_stats, pre, C, bn.sums, bn.count, BN_EPS, gamma, beta, bn.mean, bn.inv_std, bn.scale, bn.shift, rm, ri, 0.9f, 0.1f, t->st));
    else
      TK(ian_k_bn_stats_affine(y, rows, C, stride, ws, chunks(t, rows), bn.sums, bn.count, BN_EPS, gamma, beta, bn.mean, bn.inv_std, bn.scale,
                               bn.shift, rm, ri, 0.9f, 0.1f, t->st));
    TK(ian_k_affine(y, a, bn.scale, bn.shift, rows, C, stride, act, t->st));
    return 0;
  }
  TK(ian_k_colstats(0, y, nullptr, nullptr, nullptr, nullptr, rows, C, stride, 0, ws, chunks(t, rows), bn.sums, t->st));
  if ((rc = allreduce_ordered(t, bn.sums, 2 * C))) return rc;
  return bn_forward_finish(t, bn, y, a, rows, C, stride, gamma, beta, act, count_rows, rm, ri);
}
int bn_backward_finish(ian_trainer* t, BN& bn, const float* dA, const float* a, const float* y, float* dy, int64_t rows, int C, int stride, int act,
                       const std::string& gname, const std::string& bname, bool want_w);
int bn_backward(ian_trainer* t, BN& bn, const float* dA, const float* a, const float* y, float* dy, int64_t rows, int C, int stride, int act,
                const std::string& gname, const std::string& bname, bool want_w, int pre = 0) {
  double* ws = t->ws_stats;
  int rc;
  if (pre <= 0 && (rc = ws_for(t, rows, C, &ws))) return rc;
  if (!t->exact) {
    float *gb = nullptr, *gg = nullptr;
    int ab = 0, ag = 0;
    if (want_w) {
      gb = G(t, bname); gg = G(t, gname);
      ab = t->touched.count(bname) ? 1 : 0; ag = t->touched.count(gname) ? 1 : 0;
    }
    if (pre > 0) TK(ian_k_bn_bwd_finish(t->ws_stats, pre, C, bn.bsums, gb, ab, gg, ag, t->st));   // partials from the GEMM that stored dA
    else TK(ian_k_bn_bwd_stats(dA, a, y, bn.mean, bn.inv_std, rows, C, stride, act, ws, chunks(t, rows), bn.bsums, gb, ab, gg, ag, t->st));
    if (want_w && (rc = mark(t, {bname, gname}))) return rc;
    TK(ian_k_bn_bwd(dA, a, y, bn.mean, bn.inv_std, bn.scale, bn.bsums, bn.count, dy, rows, C, stride, act, t->st));
    return 0;
  }
  TK(ian_k_colstats(1, dA, a, y, bn.mean, bn.inv_std, rows, C, stride, act, ws, chunks(t, rows), bn.bsums, t->st));
  if ((rc = allreduce_ordered(t, bn.bsums, 2 * C))) return rc;
  return bn_backward_finish(t, bn, dA, a, y, dy, rows, C, stride, act, gname, bname, want_w);
}
int bn_backward_finish(ian_trainer* t, BN& bn, const float* dA, const float* a, const float* y, float* dy, int64_t rows, int C, int stride, int act,
                       const std::string& gname, const std::string& bname, bool want_w) {
  int rc;
  if (want_w) {
    // with exact statistics every rank already holds the GLOBAL dbeta / dgamma: pre-divide so that the gradient all-reduce
    // (a sum over ranks) restores them
    const double sc = 1.0 / t->world;
    if ((rc = acc64(t, bname, bn.bsums, C, sc))) return rc;
    if ((rc = acc64(t, gname, bn.bsums + C, C, sc))) return rc;
  }
  TK(ian_k_bn_bwd(dA, a, y, bn.mean, bn.inv_std, bn.scale, bn.bsums, bn.count, dy, rows, C, stride, act, t->st));
  return 0;
}
// The weight-gradient stream.  wgrad_priority = 1: created with the LOWEST stream priority the device offers, so that the dispatcher
// prefers the compute stream's workgroups whenever both streams have some ready: a data-path GEMM then keeps the whole chip, and
// the weight-gradient GEMMs -- which nothing downstream reads before the optimiser -- queue up and fill the chip while the
// compute stream runs its HBM-bound element-wise passes (batch-statistics sums, batch-norm backward, head gathers) instead of
// being consumed earlier at half rate next to a data-path GEMM (profiles/r05_train_ian_b128.md: the second stream was busy 82 of
// the 200 ms four updates span, the matrix pipe idle during most element-wise passes).  Same launches, same operands, same
// stream-order dependencies: bitwise the unprioritised step.
int make_wgrad_stream(ian_trainer* t) {
  if (t->st2) {
    (void)hipStreamSynchronize(t->st2);
    (void)hipStreamDestroy(t->st2);
    t->st2 = nullptr;
  }
  hipError_t e;
  if (t->wgrad_priority) {
    int least = 0, greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&least, &greatest);      // numerically: least >= greatest (lower number = higher priority)
    e = hipStreamCreateWithPriority(&t->st2, hipStreamNonBlocking, least);
  } else {
    e = hipStreamCreateWithFlags(&t->st2, hipStreamNonBlocking);
  }
  if (e != hipSuccess) {
    (void)hipGetLastError();
    t->st2 = nullptr;
    return 1;
  }
  return 0;
}
int side_stream(ian_trainer* t, hipStream_t* out) {  // the stream a weight-gradient launch goes to, ordered behind what is on t->st now
  *out = t->st;
  if (!t->overlap_wgrad || !t->st2) return 0;
  hipEvent_t e;
  int rc;
  if ((rc = next_event(t, &e))) return rc;
  THIP(hipEventRecord(e, t->st));
  THIP(hipStreamWaitEvent(t->st2, e, 0));
  *out = t->st2;
  return 0;
}
int join_side_stream(ian_trainer* t) {  // the compute stream waits for every weight gradient issued so far
  if (!t->overlap_wgrad || !t->st2 || !t->ev_used) return 0;
  hipEvent_t e;
  int rc;
  if ((rc = next_event(t, &e))) return rc;
  THIP(hipEventRecord(e, t->st2));
  THIP(hipStreamWaitEvent(t->st, e, 0));
  t->ev_used = 0;
  return 0;
}
int wgrad(ian_trainer* t, const std::string& lname, const float* x, const float* dy) {
  LayerRef& r = t->layers.at(lname);
  std::vector<float*> g;
  for (auto& p : r.pnames) g.push_back(G(t, p));
  hipStream_t ws;
  int rc_ = side_stream(t, &ws);
  if (rc_) return rc_;
  TL(r.l, ian_layer_backward_weight(r.l, x, dy, t->n, g.data(), (int)g.size(), t->touched.count(r.pnames[0]) ? 1 : 0, ws));
  return mark(t, r.pnames);
}
int head_backward(ian_trainer* t, const float* x, const float* dR, const float* dG, const float* dB, float* dx, bool want_w,
                  const BN* arm = nullptr, const float* arm_a = nullptr, const float* arm_y = nullptr, int* pre = nullptr) {
  if (pre) *pre = 0;
  const char* names[3] = {"R", "G_a", "B_a"};
  const float* dys[3] = {dR, dG, dB};
  bool accs[3];
  for (int i = 0; i < 3; ++i) accs[i] = t->touched.count(t->layers.at(names[i]).pnames[0]) > 0;
  if (t->head6 && accs[0] == accs[1] && accs[1] == accs[2]) {
    std::vector<float*> g[3];
    for (int i = 0; i < 3; ++i)
      for (auto& p : t->layers.at(names[i]).pnames) g[i].push_back(G(t, p));
    if (arm && fuse_stats(t)) {   // bnorm_dc4's backward statistics ride on the GEMM that stores dh4
      const int arc = arm_bwd_stats(t, "R", *arm, arm_a, arm_y, IAN_ACT_LRELU);
      if (arc) return arc;
    }
    const int rc = ian_layer_head6_backward(lay(t, "R"), lay(t, "G_a"), lay(t, "B_a"), x, dR, dG, dB, t->n, 32, dx, 128, 0,
                                            want_w ? g[0].data() : nullptr, want_w ? g[1].data() : nullptr, want_w ? g[2].data() : nullptr,
                                            want_w ? (int)g[0].size() : 0, accs[0] ? 1 : 0, t->st);
    if (rc == 0) {
      if (pre) *pre = armed_chunks(t, "R");
      if (want_w)
        for (int i = 0; i < 3; ++i) {
          const int mrc = mark(t, t->layers.at(names[i]).pnames);
          if (mrc) return mrc;
        }
      return 0;
    }
    (void)ian_layer_stats_next(lay(t, "R"), 0, nullptr, nullptr, nullptr, nullptr, 0, nullptr, 0);   // the per-layer fallback accumulates dx in three launches
    if (rc != -4) return tfail(t, rc, "ian_layer_head6_backward failed (%d): %s", rc, ian_layer_last_error(lay(t, "R")));
  }
  for (int i = 0; i < 3; ++i) {
    int rc;
    if (want_w && (rc = wgrad(t, names[i], x, dys[i]))) return rc;
    TL(lay(t, names[i]), ian_layer_backward_data(lay(t, names[i]), dys[i], t->n, dx, 0, i > 0 ? 1 : 0, t->st));
  }
  return 0;
}

// ---- encoder pass (IAN.py:71-110 + discriminator head :209-216), training mode ------------------------------------------
int enc_forward(ian_trainer* t, Bufs& E, std::map<std::string, BN>& bn, const float* x_nchw, int t0, int t1, int acc_target, bool running) {
  const int n = t->n;
  int rc;
  TK(ian_k_nchw_to_nhwc(x_nchw, E["x"], n, 4096, 3, 32, t->st));
  TL(lay(t, "enc_conv1"), ian_layer_forward(lay(t, "enc_conv1"), E["x"], n, E["a1"], 0, P(t, "enc_conv1.b"), nullptr, IAN_ACT_LRELU, t->st));
  for (int i = 2; i <= 4; ++i) {
    const int w = ENC_WIDTHS[i - 1], hw = 64 >> i;
    const std::string s = std::to_string(i), sp = std::to_string(i - 1), bnn = "bnorm" + s;
    if ((rc = arm_fwd_stats(t, "enc_conv" + s))) return rc;
    TL(lay(t, "enc_conv" + s), ian_layer_forward(lay(t, "enc_conv" + s), E["a" + sp], n, E["y" + s], 0, nullptr, nullptr, 0, t->st));
    if ((rc = bn_forward(t, bn["bn" + s], E["y" + s], E["a" + s], (int64_t)n * hw * hw, w, w, P(t, bnn + ".gamma"), P(t, bnn + ".beta"),
                         IAN_ACT_LRELU, (int64_t)n * hw * hw, running ? bnn.c_str() : nullptr, armed_chunks(t, "enc_conv" + s))))
      return rc;
  }
  TK(ian_k_globalpool(E["a4"], E["feat"], n, 16, 1024, 1024, 1024, t->st));
  TL(lay(t, "mb"), ian_layer_forward(lay(t, "mb"), E["feat"], n, E["act"], cs(2500), nullptr, nullptr, 0, t->st));
  const float* act_all = E["act"];
  int nall = n, row0 = 0;
  if (t->exact) {  // the kernel features couple every sample with every other one of the GLOBAL minibatch (layers.py:506-520)
    if ((rc = gather(t, E["act"], E["act_all"], (int64_t)n * cs(2500), "the MinibatchLayer activations"))) return rc;
    act_all = E["act_all"]; nall = t->N; row0 = t->rank * n;
  }
  TK(ian_k_mb_forward(act_all, nall, cs(2500), row0, n, 500, 5, P(t, "minibatch_discrim.b"), E["feat"], 1024, 1024, E["mb"], cs(1524), t->st));
  TK(ian_k_disc_head(E["mb"], cs(1524), 1524, P(t, "discrimi.W"), n, t0, t1, acc_target, E["p"], E["loss"], t->st));
  return 0;
}
// ce = (target0, w0, target1, w1): dlogits = sum w_t (p - onehot(target_t)).  feature_seeded: da1..da4 already hold the
// feature-loss seeds (train_IAN.py:244).  want_w: accumulate encoder_params gradients.
int enc_backward(ian_trainer* t, Bufs& E, std::map<std::string, BN>& bn, int t0, float w0, int t1, float w1, bool feature_seeded, bool want_w,
                 bool want_dx) {
  const int n = t->n;
  int rc;
  TK(ian_k_disc_head_bwd(E["p"], P(t, "discrimi.W"), 1524, n, t0, w0, t1, w1, E["dlogits"], E["dmb"], cs(1524), t->st));
  if (want_w) {
    TK(ian_k_disc_head_wgrad(E["mb"], cs(1524), 1524, n, E["dlogits"], G(t, "discrimi.W"), t->touched.count("discrimi.W") ? 1 : 0, t->st));
    if ((rc = mark(t, {"discrimi.W"}))) return rc;
    // db[k] = sum_b df[b,k] : column sums of dmb[:, 1024:1524]
    TK(ian_k_colstats(2, E["dmb"] + 1024, nullptr, nullptr, nullptr, nullptr, n, 500, cs(1524), 0, t->ws_stats, n < 256 ? n : 256, t->tmp_vals, t->st));
    if ((rc = acc64(t, "minibatch_discrim.b", t->tmp_vals, 500))) return rc;
  }
  {
    const float *act_all = E["act"], *dmb_all = E["dmb"];
    int nall = n, row0 = 0;
    if (t->exact) {  // a sample's activations feed every other sample's kernel features: their gradients come from all ranks
      if ((rc = gather(t, E["dmb"], E["dmb_all"], (int64_t)n * cs(1524), "the MinibatchLayer gradients"))) return rc;
      act_all = E["act_all"]; dmb_all = E["dmb_all"]; nall = t->N; row0 = t->rank * n;
    }
    TK(ian_k_mb_backward(act_all, nall, cs(2500), row0, n, 500, 5, dmb_all + 1024, cs(1524), E["dact"], cs(2500), t->st));
  }
  TK(ian_k_grad_pass(E["dmb"], cs(1524), 0, E["dfeat"], nullptr, 1024, n, 1024, 0, 0, t->st));  // direct path of the concat (layers.py:524)
  TL(lay(t, "mb"), ian_layer_backward_data(lay(t, "mb"), E["dact"], n, E["dfeat"], 1024, 1, t->st));
  if (want_w) {
    float* dW = t->mb_dW;
    TL(lay(t, "mb"), ian_layer_backward_weight(lay(t, "mb"), E["feat"], E["dact"], n, &dW, 1, 0, t->st));
    TK(ian_k_mb_weight_bwd(P(t, "minibatch_discrim.theta"), t->mb_colscale, t->mb_dW, G(t, "minibatch_discrim.theta"),
                           G(t, "minibatch_discrim.log_weight_scale"), 1024, 2500, t->touched.count("minibatch_discrim.theta") ? 1 : 0, t->st));
    if ((rc = mark(t, {"minibatch_discrim.theta", "minibatch_discrim.log_weight_scale"}))) return rc;
  }
  TK(ian_k_globalpool_bwd(E["dfeat"], E["da4"], n, 16, 1024, 1024, 1024, feature_seeded ? 1 : 0, t->st));
  int pre_next = 0;   // chunk partials the previous backward-data GEMM left for the next batch-norm backward (0: run colstats)
  for (int i = 4; i >= 2; --i) {
    const int w = ENC_WIDTHS[i - 1], hw = 64 >> i;
    const std::string s = std::to_string(i), sp = std::to_string(i - 1);
    float *da = E["da" + s], *a = E["a" + s], *y = E["y" + s];
    if ((rc = bn_backward(t, bn["bn" + s], da, a, y, da, (int64_t)n * hw * hw, w, w, IAN_ACT_LRELU, "bnorm" + s + ".gamma", "bnorm" + s + ".beta", want_w,
                          pre_next)))
      return rc;
    if (want_w && (rc = wgrad(t, "enc_conv" + s, E["a" + sp], da))) return rc;
    // the gradient this launch stores is the input of the next normalisation down (bnorm<i-1>): its statistics ride along
    if (i - 1 >= 2 && (rc = arm_bwd_stats(t, "enc_conv" + s, bn["bn" + sp], E["a" + sp], E["y" + sp], IAN_ACT_LRELU))) return rc;
    TL(lay(t, "enc_conv" + s), ian_layer_backward_data(lay(t, "enc_conv" + s), da, n, E["da" + sp], 0, feature_seeded ? 1 : 0, t->st));
    pre_next = i - 1 >= 2 ? armed_chunks(t, "enc_conv" + s) : 0;
  }
  // enc_conv1: bias + lrelu, no batch-norm (IAN.py:71-80)
  if (want_w) {
    TK(ian_k_colstats(2, E["da1"], E["a1"], nullptr, nullptr, nullptr, (int64_t)n * 1024, 128, 128, IAN_ACT_LRELU, t->ws_stats, 256, t->tmp_vals, t->st));
    if ((rc = acc64(t, "enc_conv1.b", t->tmp_vals, 128))) return rc;
  }
  TK(ian_k_bn_bwd(E["da1"], E["a1"], nullptr, nullptr, nullptr, nullptr, nullptr, 1.f, E["da1"], (int64_t)n * 1024, 128, 128, IAN_ACT_LRELU, t->st));
  if (want_w && (rc = wgrad(t, "enc_conv1", E["x"], E["da1"]))) return rc;
  if (want_dx) TL(lay(t, "enc_conv1"), ian_layer_backward_data(lay(t, "enc_conv1"), E["da1"], n, E["dx"], 0, 0, t->st));
  return 0;
}

// ---- latent path (IAN.py:114-128): enc_fc1 -> (mu, logsigma) -> z0 = mu + e^ls * eps -> IAF ------------------------------
int z_forward(ian_trainer* t, const float* a4, const float* eps) {
  const int n = t->n, Z = t->cfg.num_latents;
  Bufs& S = t->ZS;
  int rc;
  TL(lay(t, "enc_fc1"), ian_layer_forward(lay(t, "enc_fc1"), a4, n, S["y_fc1"], 1024, nullptr, nullptr, 0, t->st));
  if ((rc = bn_forward(t, t->bnZ["bn_fc1"], S["y_fc1"], S["f"], n, 1000, 1024, P(t, "bnorm_enc_fc1.gamma"), P(t, "bnorm_enc_fc1.beta"), IAN_ACT_RELU, n,
                       "bnorm_enc_fc1")))
    return rc;
  const char* trip[2][3] = {{"mu", "enc_mu", "mu_bnorm"}, {"ls", "enc_logsigma", "ls_bnorm"}};
  if (!t->exact) {
    for (auto& tr : trip) {
      const std::string nm = tr[0], bnn = tr[2];
      TL(lay(t, tr[1]), ian_layer_forward(lay(t, tr[1]), S["f"], n, S["y_" + nm], 128, nullptr, nullptr, 0, t->st));
      if ((rc = bn_forward(t, t->bnZ["bn_" + nm], S["y_" + nm], S[nm], n, Z, 128, P(t, bnn + ".gamma"), P(t, bnn + ".beta"), 0, n, tr[2]))) return rc;
    }
  } else {  // both layers, both per-rank statistics, ONE all-gather of the adjacent sums, then both normalisations
    double* ws;
    if ((rc = ws_for(t, n, Z, &ws))) return rc;
    for (auto& tr : trip) {
      const std::string nm = tr[0];
      TL(lay(t, tr[1]), ian_layer_forward(lay(t, tr[1]), S["f"], n, S["y_" + nm], 128, nullptr, nullptr, 0, t->st));
      TK(ian_k_colstats(0, S["y_" + nm], nullptr, nullptr, nullptr, nullptr, n, Z, 128, 0, ws, chunks(t, n), t->bnZ["bn_" + nm].sums, t->st));
    }
    if ((rc = allreduce_ordered(t, t->bnZ["bn_mu"].sums, 4 * Z))) return rc;
    for (auto& tr : trip) {
      const std::string nm = tr[0], bnn = tr[2];
      float *rm = nullptr, *ri = nullptr;
      if (t->update_running) { rm = P(t, bnn + ".mean"); ri = P(t, bnn + ".inv_std"); }
      if ((rc = bn_forward_finish(t, t->bnZ["bn_" + nm], S["y_" + nm], S[nm], n, Z, 128, P(t, bnn + ".gamma"), P(t, bnn + ".beta"), 0, n, rm, ri))) return rc;
    }
  }
  TK(ian_k_sample(S["mu"], S["ls"], eps, S["z0"], S["kl"], n, Z, 128, Z, t->st));
  TK(ian_k_made_iaf(S["z0"], S["z"], t->made_w, t->made_b, n, Z, 128, t->st));
  return 0;
}
// dz = dL/dz (from the decoder) -> gradients of Z_params, including KL and (later) the L2 penalty
int z_backward(ian_trainer* t, const float* dz, const float* a4) {
  const int n = t->n, Z = t->cfg.num_latents;
  Bufs& S = t->ZS;
  int rc;
  TK(ian_k_made_iaf_bwd(S["z0"], dz, S["dz0"], t->made_w, t->made_b, n, Z, 128, t->st));
  const float klw = 1.f / ((float)t->N * 100.f);  // d(-0.5*mean(...)) over the GLOBAL batch: factor folded in the kernel's formula
  TK(ian_k_sample_bwd(S["mu"], S["ls"], t->eps, S["dz0"], S["dmu"], S["dls"], n, Z, 128, Z, klw, t->st));
  const char* trip[2][3] = {{"mu", "enc_mu", "mu_bnorm"}, {"ls", "enc_logsigma", "ls_bnorm"}};
  bool first = true;
  if (t->exact) {  // both gradient statistics, ONE all-gather (the sums are adjacent: z_alloc)
    double* ws;
    if ((rc = ws_for(t, n, Z, &ws))) return rc;
    for (auto& tr : trip) {
      const std::string nm = tr[0];
      BN& b = t->bnZ["bn_" + nm];
      TK(ian_k_colstats(1, S["d" + nm], nullptr, S["y_" + nm], b.mean, b.inv_std, n, Z, 128, 0, ws, chunks(t, n), b.bsums, t->st));
    }
    if ((rc = allreduce_ordered(t, t->bnZ["bn_mu"].bsums, 4 * Z))) return rc;
  }
  for (auto& tr : trip) {
    const std::string nm = tr[0], bnn = tr[2];
    float* d = S["d" + nm];
    if (t->exact) rc = bn_backward_finish(t, t->bnZ["bn_" + nm], d, nullptr, S["y_" + nm], d, n, Z, 128, 0, bnn + ".gamma", bnn + ".beta", true);
    else rc = bn_backward(t, t->bnZ["bn_" + nm], d, nullptr, S["y_" + nm], d, n, Z, 128, 0, bnn + ".gamma", bnn + ".beta", true);
    if (rc) return rc;
    if ((rc = wgrad(t, tr[1], S["f"], d))) return rc;
    TL(lay(t, tr[1]), ian_layer_backward_data(lay(t, tr[1]), d, n, S["df"], 1024, first ? 0 : 1, t->st));
    first = false;
  }
  if ((rc = bn_backward(t, t->bnZ["bn_fc1"], S["df"], S["f"], S["y_fc1"], S["df"], n, 1000, 1024, IAN_ACT_RELU, "bnorm_enc_fc1.gamma", "bnorm_enc_fc1.beta",
                        true)))
    return rc;
  return wgrad(t, "enc_fc1", a4, S["df"]);
}

// ---- decoder pass (IAN.py:129-207), training mode -----------------------------------------------------------------------
int dec_forward(ian_trainer* t, Bufs& D, std::map<std::string, BN>& bn, const float* zbuf, bool running) {
  const int n = t->n;
  int rc;
  TL(lay(t, "l_dec_fc2"), ian_layer_forward(lay(t, "l_dec_fc2"), zbuf, n, D["h0"], 8192, t->fc2_bias, nullptr, IAN_ACT_LRELU, t->st));
  const float* h = D["h0"];
  for (const DecStage& s : dec_stages()) {
    const std::string blk = s.blk;
    const int64_t rows = (int64_t)n * (2 * s.hw) * (2 * s.hw);
    auto g = [&](int j, const char* w) { return P(t, blk + "bnorm" + std::to_string(j) + "." + w); };
    auto rn = [&](int j) -> std::string { return blk + "bnorm" + std::to_string(j); };
    if ((rc = arm_fwd_stats(t, s.dc))) return rc;
    TL(lay(t, s.dc), ian_layer_forward(lay(t, s.dc), h, n, D[blk + "_x"], 0, nullptr, nullptr, 0, t->st));
    if ((rc = bn_forward(t, bn[blk + "_bn0"], D[blk + "_x"], D[blk + "_a"], rows, s.co, s.co, g(0, "gamma"), g(0, "beta"), IAN_ACT_LRELU, rows,
                         running ? rn(0).c_str() : nullptr, armed_chunks(t, s.dc))))
      return rc;
    if ((rc = arm_fwd_stats(t, blk))) return rc;
    TL(lay(t, blk), ian_layer_forward(lay(t, blk), D[blk + "_a"], n, D[blk + "_b"], 0, nullptr, nullptr, 0, t->st));
    if ((rc = bn_forward(t, bn[blk + "_bn1"], D[blk + "_b"], D[blk + "_c"], rows, s.co, s.co, g(1, "gamma"), g(1, "beta"), IAN_ACT_LRELU, rows,
                         running ? rn(1).c_str() : nullptr, armed_chunks(t, blk))))
      return rc;
    if ((rc = arm_fwd_stats(t, blk + "2"))) return rc;
    TL(lay(t, blk + "2"), ian_layer_forward(lay(t, blk + "2"), D[blk + "_c"], n, D[blk + "_e"], 0, nullptr, D[blk + "_x"], 0, t->st));  // ElemwiseSum (layers.py:415)
    if ((rc = bn_forward(t, bn[blk + "_bn2"], D[blk + "_e"], D[blk + "_h"], rows, s.co, s.co, g(2, "gamma"), g(2, "beta"), IAN_ACT_LRELU, rows,
                         running ? rn(2).c_str() : nullptr, armed_chunks(t, blk + "2"))))
      return rc;
    h = D[blk + "_h"];
  }
  const int64_t rows = (int64_t)n * 4096;
  if ((rc = arm_fwd_stats(t, "dec_conv4"))) return rc;
  TL(lay(t, "dec_conv4"), ian_layer_forward(lay(t, "dec_conv4"), h, n, D["y4"], 0, nullptr, nullptr, 0, t->st));
  if ((rc = bn_forward(t, bn["bn4"], D["y4"], D["h4"], rows, 128, 128, P(t, "bnorm_dc4.gamma"), P(t, "bnorm_dc4.beta"), IAN_ACT_LRELU, rows,
                       running ? "bnorm_dc4" : nullptr, armed_chunks(t, "dec_conv4"))))
    return rc;
  const int sg = IAN_ACT_SIGMOID;
  // R = sigmoid(MDCL(h4)), G_a, B_a (IAN.py:183-199): the three layers that read the 128-channel map, one pass over it
  int h6 = -4;
  if (t->head6) h6 = ian_layer_head6_forward(lay(t, "R"), lay(t, "G_a"), lay(t, "B_a"), D["h4"], n, D["R"], D["Ga"], D["Ba"], 32, sg, 0, 0, t->st);
  if (h6 == -4) {
    TL(lay(t, "R"), ian_layer_forward(lay(t, "R"), D["h4"], n, D["R"], 0, nullptr, nullptr, sg, t->st));  // IAN.py:183-186
    TL(lay(t, "G_a"), ian_layer_forward(lay(t, "G_a"), D["h4"], n, D["Ga"], 0, nullptr, nullptr, 0, t->st));
    TL(lay(t, "B_a"), ian_layer_forward(lay(t, "B_a"), D["h4"], n, D["Ba"], 0, nullptr, nullptr, 0, t->st));
  } else if (h6) {
    return tfail(t, h6, "ian_layer_head6_forward failed (%d): %s", h6, ian_layer_last_error(lay(t, "R")));
  }
  TL(lay(t, "G_b"), ian_layer_forward(lay(t, "G_b"), D["R"], n, D["G"], 0, nullptr, D["Ga"], sg, t->st));   // :187-196
  TK(ian_k_concat2(D["R"], 2, 32, D["G"], 2, 32, D["RG"], 32, rows, t->st));                                // :201
  TL(lay(t, "B_b"), ian_layer_forward(lay(t, "B_b"), D["RG"], n, D["B"], 0, nullptr, D["Ba"], sg, t->st));  // :197-206
  TK(ian_k_beta(D["R"], D["G"], D["B"], D["xhat"], n, 4096, 32, t->st));                                    // :207
  return 0;
}
// D['dxhat'] (NCHW) -> gradients of decoder_params (want_w) and D['dz'] (want_dz)
int dec_backward(ian_trainer* t, Bufs& D, std::map<std::string, BN>& bn, const float* zbuf, bool want_w, bool want_dz) {
  const int n = t->n;
  const int64_t rows = (int64_t)n * 4096;
  const int sg = IAN_ACT_SIGMOID;
  int rc;
  TK(ian_k_beta_bwd(D["dxhat"], D["R"], D["G"], D["B"], D["gR"], D["gG"], D["gB"], n, 4096, 32, sg, t->st));
  // B = sigmoid(B_a(h4) + B_b([R,G]))
  if (want_w && (rc = wgrad(t, "B_b", D["RG"], D["gB"]))) return rc;
  TL(lay(t, "B_b"), ian_layer_backward_data(lay(t, "B_b"), D["gB"], n, D["dRG"], 0, 0, t->st));
  TK(ian_k_grad_pass(D["dRG"], 32, 0, D["gR"], D["R"], 32, rows, 2, sg, 1, t->st));
  TK(ian_k_grad_pass(D["dRG"], 32, 2, D["gG"], D["G"], 32, rows, 2, sg, 1, t->st));
  // G = sigmoid(G_a(h4) + G_b(R))
  if (want_w && (rc = wgrad(t, "G_b", D["R"], D["gG"]))) return rc;
  TL(lay(t, "G_b"), ian_layer_backward_data(lay(t, "G_b"), D["gG"], n, D["dRt"], 0, 0, t->st));
  TK(ian_k_grad_pass(D["dRt"], 32, 0, D["gR"], D["R"], 32, rows, 2, sg, 1, t->st));
  // R = sigmoid(R(h4)): all three seeds are final here
  int pre = 0;   // chunk partials the GEMM that stored a gradient left for the batch-norm backward that consumes it (0: colstats)
  if ((rc = head_backward(t, D["h4"], D["gR"], D["gG"], D["gB"], D["dh4"], want_w, &bn["bn4"], D["h4"], D["y4"], &pre))) return rc;
  // dec_conv4 + bnorm_dc4 + lrelu
  if ((rc = bn_backward(t, bn["bn4"], D["dh4"], D["h4"], D["y4"], D["dh4"], rows, 128, 128, IAN_ACT_LRELU, "bnorm_dc4.gamma", "bnorm_dc4.beta", want_w,
                        pre)))
    return rc;
  const auto& ST = dec_stages();
  const std::string last_blk = ST.back().blk;
  if (want_w && (rc = wgrad(t, "dec_conv4", D[last_blk + "_h"], D["dh4"]))) return rc;
  if ((rc = arm_bwd_stats(t, "dec_conv4", bn[last_blk + "_bn2"], D[last_blk + "_h"], D[last_blk + "_e"], IAN_ACT_LRELU))) return rc;
  TL(lay(t, "dec_conv4"), ian_layer_backward_data(lay(t, "dec_conv4"), D["dh4"], n, D[last_blk + "_dh"], 0, 0, t->st));
  pre = armed_chunks(t, "dec_conv4");
  for (int si = (int)ST.size() - 1; si >= 0; --si) {
    const DecStage& s = ST[si];
    const std::string blk = s.blk;
    const int64_t r = (int64_t)n * (2 * s.hw) * (2 * s.hw);
    auto bnn = [&](int j, const char* w) { return blk + "bnorm" + std::to_string(j) + "." + w; };
    float *dh = D[blk + "_dh"], *dx = D[blk + "_dx"], *da = D[blk + "_da"], *dcg = D[blk + "_dc"];
    // h = lrelu(bn2(x + d)),  d = MDCL2(c)
    if ((rc = bn_backward(t, bn[blk + "_bn2"], dh, D[blk + "_h"], D[blk + "_e"], dh, r, s.co, s.co, IAN_ACT_LRELU, bnn(2, "gamma"), bnn(2, "beta"), want_w,
                          pre)))
      return rc;
    if (want_w && (rc = wgrad(t, blk + "2", D[blk + "_c"], dh))) return rc;
    if ((rc = arm_bwd_stats(t, blk + "2", bn[blk + "_bn1"], D[blk + "_c"], D[blk + "_b"], IAN_ACT_LRELU))) return rc;
    TL(lay(t, blk + "2"), ian_layer_backward_data(lay(t, blk + "2"), dh, n, dcg, 0, 0, t->st));
    if ((rc = bn_backward(t, bn[blk + "_bn1"], dcg, D[blk + "_c"], D[blk + "_b"], dcg, r, s.co, s.co, IAN_ACT_LRELU, bnn(1, "gamma"), bnn(1, "beta"), want_w,
                          armed_chunks(t, blk + "2"))))
      return rc;
    if (want_w && (rc = wgrad(t, blk, D[blk + "_a"], dcg))) return rc;
    if ((rc = arm_bwd_stats(t, blk, bn[blk + "_bn0"], D[blk + "_a"], D[blk + "_x"], IAN_ACT_LRELU))) return rc;
    TL(lay(t, blk), ian_layer_backward_data(lay(t, blk), dcg, n, da, 0, 0, t->st));
    if ((rc = bn_backward(t, bn[blk + "_bn0"], da, D[blk + "_a"], D[blk + "_x"], dx, r, s.co, s.co, IAN_ACT_LRELU, bnn(0, "gamma"), bnn(0, "beta"), want_w,
                          armed_chunks(t, blk))))
      return rc;
    TK(ian_k_axpy(1.f, dh, dx, r * s.co, 1, t->st));  // residual edge: dx += d(x+d)
    const float* src = si == 0 ? D["h0"] : D[std::string(ST[si - 1].blk) + "_h"];
    if (want_w && (rc = wgrad(t, s.dc, src, dx))) return rc;
    pre = 0;
    if (si > 0) {   // the gradient this launch stores enters the previous stage's bnorm2 backward
      const std::string pb = ST[si - 1].blk;
      if ((rc = arm_bwd_stats(t, s.dc, bn[pb + "_bn2"], D[pb + "_h"], D[pb + "_e"], IAN_ACT_LRELU))) return rc;
    }
    TL(lay(t, s.dc), ian_layer_backward_data(lay(t, s.dc), dx, n, si == 0 ? D["dh0"] : D[std::string(ST[si - 1].blk) + "_dh"], 0, 0, t->st));
    if (si > 0) pre = armed_chunks(t, s.dc);
  }
  // l_dec_fc2: bias + lrelu
  TK(ian_k_bn_bwd(D["dh0"], D["h0"], nullptr, nullptr, nullptr, nullptr, nullptr, 1.f, D["dh0"], n, 8192, 8192, IAN_ACT_LRELU, t->st));
  if (want_w) {
    if ((rc = wgrad(t, "l_dec_fc2", zbuf, D["dh0"]))) return rc;
    TK(ian_k_colstats(2, D["dh0"], nullptr, nullptr, nullptr, nullptr, n, 8192, 8192, 0, t->ws_stats, n < 256 ? n : 256, t->tmp_big, t->st));
    TK(ian_k_axpy_f64(1.0, t->tmp_big, t->fc2_tmp, 8192, 0, t->st));   // float64 column sums -> float32, (H,W,C) order
    TK(ian_k_gather(t->fc2_tmp, t->fc2_inv, t->fc2_db, 8192, t->st));
    if ((rc = acc(t, "l_dec_fc2.b", t->fc2_db, 8192))) return rc;
  }
  if (want_dz) TL(lay(t, "l_dec_fc2"), ian_layer_backward_data(lay(t, "l_dec_fc2"), D["dh0"], n, D["dz"], 128, 0, t->st));
  return 0;
}


// ---- timing-only two-chain experiment (see ian_trainer::dual_timing) -------------------------------------------------------------
int chain_fork(ian_trainer* t) {   // stream B starts behind everything issued to the compute stream so far
  if (!t->stB) THIP(hipStreamCreateWithFlags(&t->stB, hipStreamNonBlocking));
  hipEvent_t e;
  THIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  THIP(hipEventRecord(e, t->st));
  THIP(hipStreamWaitEvent(t->stB, e, 0));
  THIP(hipEventDestroy(e));
  return 0;
}
int chain_join(ian_trainer* t) {   // the compute stream waits for stream B
  hipEvent_t e;
  THIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  THIP(hipEventRecord(e, t->stB));
  THIP(hipStreamWaitEvent(t->st, e, 0));
  THIP(hipEventDestroy(e));
  return 0;
}
struct OnStreamB {                 // everything issued while this object lives goes to stream B
  ian_trainer* t;
  hipStream_t keep;
  explicit OnStreamB(ian_trainer* tt) : t(tt), keep(tt->st) { t->st = t->stB; }
  ~OnStreamB() { t->st = keep; }
};

// ---- the step -----------------------------------------------------------------------------------------------------------
int forward(ian_trainer* t, const float* X, const float* Zr, const float* eps) {  // the three passes of train_IAN.py:116-149
  const int n = t->n, Z = t->cfg.num_latents;
  int rc;
  if (t->measure_exposed && t->world > 1) {   // this step's all-gather events go to the pool of two steps ago: fold it first
    t->gpar ^= 1;
    fold_gathers(t, t->gpar);
  }
  if ((rc = refresh_weights(t))) return rc;
  t->X = X;
  t->eps = eps;
  if (t->dual_timing) {   // TIMING ONLY: chain B concurrently on its own stream, workspaces shared without protection
    if ((rc = chain_fork(t))) return rc;
    if ((rc = enc_forward(t, t->EX, t->bnEX, X, 0, -1, 0, true))) return rc;
    if ((rc = z_forward(t, t->EX["a4"], eps))) return rc;
    if ((rc = dec_forward(t, t->DZ, t->bnDZ, t->ZS["z"], true))) return rc;
    if ((rc = enc_forward(t, t->EH, t->bnEH, t->DZ["xhat"], 0, 1, 1, false))) return rc;
    {
      OnStreamB b(t);
      TK(ian_k_grad_pass(Zr, Z, 0, t->zgen0, nullptr, 128, n, Z, 0, 0, t->st));
      TK(ian_k_made_iaf(t->zgen0, t->zgen, t->made_w, t->made_b, n, Z, 128, t->st));
      if ((rc = dec_forward(t, t->DG, t->bnDG, t->zgen, false))) return rc;
      if ((rc = enc_forward(t, t->EG, t->bnEG, t->DG["xhat"], 0, 2, 2, false))) return rc;
    }
    return chain_join(t);
  }
  if ((rc = enc_forward(t, t->EX, t->bnEX, X, 0, -1, 0, true))) return rc;  // p_X vs p1
  if ((rc = z_forward(t, t->EX["a4"], eps))) return rc;
  if ((rc = dec_forward(t, t->DZ, t->bnDZ, t->ZS["z"], true))) return rc;   // X_hat
  if ((rc = enc_forward(t, t->EH, t->bnEH, t->xhat_override ? t->xhat_override : t->DZ["xhat"], 0, 1, 1, false))) return rc;  // p_X_hat
  TK(ian_k_grad_pass(Zr, Z, 0, t->zgen0, nullptr, 128, n, Z, 0, 0, t->st));  // (n,100) -> padded rows
  TK(ian_k_made_iaf(t->zgen0, t->zgen, t->made_w, t->made_b, n, Z, 128, t->st));  // {l_Z_IAF: Z} (train_IAN.py:149)
  if ((rc = dec_forward(t, t->DG, t->bnDG, t->zgen, false))) return rc;     // X_gen
  return enc_forward(t, t->EG, t->bnEG, t->xgen_override ? t->xgen_override : t->DG["xhat"], 0, 2, 2, false);  // p_X_gen
}

// all scalar losses of train_IAN.py:169-250,279: partial sums on the device (+ their all-reduce: issued on EVERY data-parallel step,
// with or without a reader, so that ranks that disagree on `metrics` keep identical collective sequences), then one device->host copy
int metrics_device(ian_trainer* t) {
  const int n = t->n;
  const float N = (float)t->N;   // means are over the GLOBAL batch
  float* s = t->scalars;
  THIP(hipMemsetAsync(s, 0, 64 * sizeof(float), t->st));
  TK(ian_k_sum_rows(t->EX["loss"], n, 4, 1.f / N, s + 0, t->st));   // [0] discrim_d_loss, [2] acc(p_X)
  TK(ian_k_sum_rows(t->EH["loss"], n, 4, 1.f / N, s + 4, t->st));   // [4] gen_recon_loss, [5] CE(p_X_hat,p2), [6] acc
  TK(ian_k_sum_rows(t->EG["loss"], n, 4, 1.f / N, s + 8, t->st));   // [8] gen_sample_loss, [9] CE(p_X_gen,p3), [10] acc
  TK(ian_k_sum_rows(t->ZS["kl"], n * 100, 1, -0.5f / (N * 100.f), s + 12, t->st));
  TK(ian_k_pair_loss(t->DZ["xhat"], t->X, nullptr, (int64_t)n * 3 * 4096, 1, 1, 0, 0.f, 0, t->ws_loss, 1024, 1.f / (N * 3.f * 4096.f), s + 16, t->st));
  for (int i = 0; i < 4; ++i) {
    const float cnt = (float)((32 >> i) * (32 >> i) * ENC_WIDTHS[i]);
    const std::string a = "a" + std::to_string(i + 1);
    TK(ian_k_pair_loss(t->EH[a], t->EX[a], nullptr, (int64_t)n * (int64_t)cnt, 1, 1, 1, 0.f, 0, t->ws_loss, 1024, 1.f / (N * cnt * 4.f), s + 20 + 2 * i, t->st));
  }
  if (t->world > 1) {  // per-rank partial means add up to the global ones
    int rc = t->comm.allreduce_sum(t->comm.ctx, s, 64, t->st);
    if (!rc) rc = t->comm.wait_all(t->comm.ctx, t->st);
    if (rc) return tfail(t, -30, "comm all-reduce of the metrics failed (%d)", rc);
  }
  return 0;
}
int metrics_read(ian_trainer* t, float* out9) {
  float* s = t->scalars;
  float v[32];
  THIP(hipMemcpyAsync(v, s, sizeof v, hipMemcpyDeviceToHost, t->st));
  THIP(hipStreamSynchronize(t->st));
  out9[0] = v[0];                             // discrim_d_loss
  out9[1] = v[4];                             // gen_recon_loss
  out9[2] = v[8];                             // gen_sample_loss
  out9[3] = v[5] + v[9];                      // discrim_g_loss
  out9[4] = (v[2] + v[6] + v[10]) / 3.f;      // discrim_acc
  out9[5] = v[12];                            // kl_div
  out9[6] = v[16];                            // pixel_loss
  out9[7] = 1.f - v[17];                      // pixel_acc
  out9[8] = v[20] + v[22] + v[24] + v[26];    // feature_loss
  return 0;
}
int metrics(ian_trainer* t, float* out9) {
  const int rc = metrics_device(t);
  return rc ? rc : metrics_read(t, out9);
}

int backward(ian_trainer* t, bool gen) {  // gradients of the update rules of train_IAN.py:253-273 (Z_params always)
  const int n = t->n;
  const float N = (float)t->N;   // loss means are over the GLOBAL batch: summing per-rank gradients gives the 1-GPU gradient
  const ian_train_config& c = t->cfg;
  int rc;
  if ((rc = begin_backward(t, gen ? 0 : 1))) return rc;
  // ---- shared generator-side loss S = adv_gen + recon_weight*pixel + feature_weight*feature ---------------------------------
  for (int i = 0; i < 4; ++i) {  // feature_loss seeds (train_IAN.py:244)
    const float cnt = (float)((32 >> i) * (32 >> i) * ENC_WIDTHS[i]);
    const std::string a = "a" + std::to_string(i + 1);
    TK(ian_k_pair_loss(t->EH[a], t->EX[a], t->EH["d" + a], (int64_t)n * (int64_t)cnt, 1, 1, 1, c.feature_weight / (4.f * N * cnt), 0, t->ws_loss, 1024, 0.f,
                       t->scalars + 40, t->st));
  }
  if ((rc = enc_backward(t, t->EH, t->bnEH, 0, c.agr_weight / N, -1, 0.f, true, false, true))) return rc;  // gen_recon_loss (:247)
  TK(ian_k_pair_loss(t->DZ["xhat"], t->X, t->DZ["dxhat"], (int64_t)n * 3 * 4096, 1, 1, 0, c.recon_weight / (N * 3.f * 4096.f), 0, t->ws_loss, 1024, 0.f,
                     t->scalars + 40, t->st));  // pixel_loss (:169)
  TK(ian_k_nhwc_to_nchw(t->EH["dx"], 32, t->DZ["tmp_img"], n, 4096, 3, t->st));
  TK(ian_k_axpy(1.f, t->DZ["tmp_img"], t->DZ["dxhat"], (int64_t)n * 3 * 4096, 1, t->st));
  if (t->dual_timing) {   // TIMING ONLY (see forward)
    if ((rc = chain_fork(t))) return rc;
    if ((rc = dec_backward(t, t->DZ, t->bnDZ, t->ZS["z"], gen, true))) return rc;
    if ((rc = z_backward(t, t->DZ["dz"], t->EX["a4"]))) return rc;
    if (gen) {
      OnStreamB b(t);
      if ((rc = enc_backward(t, t->EG, t->bnEG, 0, c.ags_weight / N, -1, 0.f, false, false, true))) return rc;
      TK(ian_k_nhwc_to_nchw(t->EG["dx"], 32, t->DG["dxhat"], n, 4096, 3, t->st));
      if ((rc = dec_backward(t, t->DG, t->bnDG, t->zgen, true, false))) return rc;
    } else {
      if ((rc = enc_backward(t, t->EX, t->bnEX, 0, c.dd_weight / N, -1, 0.f, false, true, false))) return rc;
      if ((rc = enc_backward(t, t->EH, t->bnEH, 1, c.dg_weight / N, -1, 0.f, false, true, false))) return rc;
      OnStreamB b(t);
      if ((rc = enc_backward(t, t->EG, t->bnEG, 2, c.dg_weight / N, -1, 0.f, false, true, false))) return rc;
    }
    return chain_join(t);
  }
  if ((rc = dec_backward(t, t->DZ, t->bnDZ, t->ZS["z"], gen, true))) return rc;
  if ((rc = z_backward(t, t->DZ["dz"], t->EX["a4"]))) return rc;
  if (gen) {
    if ((rc = enc_backward(t, t->EG, t->bnEG, 0, c.ags_weight / N, -1, 0.f, false, false, true))) return rc;  // gen_sample_loss (:248)
    TK(ian_k_nhwc_to_nchw(t->EG["dx"], 32, t->DG["dxhat"], n, 4096, 3, t->st));
    return dec_backward(t, t->DG, t->bnDG, t->zgen, true, false);
  }
  // ---- discriminator loss, X_hat and X_gen constant (consider_constant, train_IAN.py:253) ------------------------------------
  if ((rc = enc_backward(t, t->EX, t->bnEX, 0, c.dd_weight / N, -1, 0.f, false, true, false))) return rc;  // discrim_d_loss (:234)
  if ((rc = enc_backward(t, t->EH, t->bnEH, 1, c.dg_weight / N, -1, 0.f, false, true, false))) return rc;  // p_X_hat vs p2 (:228)
  return enc_backward(t, t->EG, t->bnEG, 2, c.dg_weight / N, -1, 0.f, false, true, false);                 // p_X_gen vs p3
}

int regularizers(ian_trainer* t, bool gen) {  // train_IAN.py:211-221: L2 on the Z parameters, orthogonal penalty on the 4-D weights
  const ian_train_config& c = t->cfg;
  for (auto& nme : t->zp.names) {
    if (nme.size() >= 5 && nme.compare(nme.size() - 5, 5, ".beta") == 0) continue;
    TK(ian_k_axpy(2.f * c.reg, P(t, nme), G(t, nme), numel_of(t, nme), 1, t->st));
  }
  if (c.ortho < 0.f) return 0;
  Group& grp = gen ? t->dec : t->enc;
  for (auto& nme : grp.names) {
    const Shape& s = grp.off.at(nme).second;
    if (nme.back() == 'W' && s.d.size() == 4)
      TK(ian_k_ortho(P(t, nme), G(t, nme), (int)s.d[0], (int)s.d[1], (int)s.d[2], c.ortho, t->ortho_vals, t->st));
  }
  return 0;
}

int adam(ian_trainer* t, Group& g, const char* gname) {  // lasagne.updates.adam (App. B.7): one (t, m, v) per group
  g.t += 1;
  const double b1 = t->cfg.beta1, b2 = 0.999;
  const double a_t = t->cfg.learning_rate * sqrt(1.0 - pow(b2, g.t)) / (1.0 - pow(b1, g.t));
  TK(ian_k_adam(g.p, g.g, g.m, g.v, g.numel, (float)a_t, (float)b1, (float)b2, 1e-8f, t->st));
  t->dirty.insert(gname);
  return 0;
}

bool is_device_ptr(const void* p) {
  hipPointerAttribute_t a;
  if (hipPointerGetAttributes(&a, p) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  return a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged;
}

}  // namespace

extern "C" {

int ian_trainer_create(const ian_train_config* cfg, ian_trainer** out) {
  if (!cfg || !out || cfg->batch <= 0 || cfg->num_latents != 100) return -1;  // the wired graph is IAN.py's: 100 latents
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    (void)hipGetLastError();
    return -10;  // no HIP device: libian has no CPU fallback
  }
  ian_trainer* t = new ian_trainer();
  t->cfg = *cfg;
  t->n = cfg->batch;
  t->N = cfg->batch;
  memset(&t->comm, 0, sizeof t->comm);
  declare_parameters(t);
  *out = t;
  return 0;
}

int ian_trainer_load_param(ian_trainer* t, const char* name, const float* data, int64_t numel) {
  if (!t || !name || !data) return -1;
  if (t->finalized) return tfail(t, -6, "trainer already finalized");
  auto it = t->shapes.find(name);
  if (it == t->shapes.end()) return tfail(t, -2, "unknown parameter '%s'", name);
  if (it->second.numel() != numel) return tfail(t, -3, "parameter '%s' has %lld elements, expected %lld", name, (long long)numel, (long long)it->second.numel());
  t->host[name].assign(data, data + numel);
  return 0;
}

int ian_trainer_set_made_masks(ian_trainer* t, const float* m0, const float* m1, const float* md, int32_t n) {
  if (!t || !m0 || !m1 || !md || n != t->cfg.num_latents) return tfail(t, -1, "bad argument to ian_trainer_set_made_masks");
  const float* src[3] = {m0, m1, md};
  for (int k = 0; k < 3; ++k) {
    t->masks[k].assign(src[k], src[k] + (size_t)n * n);
    for (float v : t->masks[k])
      if (v != 0.f && v != 1.f) return tfail(t, -3, "MADE mask %d is not 0/1 valued", k);
  }
  t->made_n = n;
  return 0;
}

int ian_trainer_finalize(ian_trainer* t) {
  if (!t) return -1;
  if (t->finalized) return tfail(t, -6, "trainer already finalized");
  const int Z = t->cfg.num_latents;
  if (t->made_n != Z) return tfail(t, -3, "MADE masks not set (ian_trainer_set_made_masks)");
  for (auto& kv : t->shapes)
    if (!t->host.count(kv.first)) return tfail(t, -2, "missing parameter '%s'", kv.first.c_str());
  int rc;
  for (Group* g : {&t->enc, &t->zp, &t->dec, &t->stats}) {
    g->p = dalloc(t, g->numel);
    if (g != &t->stats) { g->g = dalloc(t, g->numel); g->m = dalloc(t, g->numel); g->v = dalloc(t, g->numel); }
    if (!g->p) return tfail(t, -20, "out of device memory (parameter groups)");
    std::vector<float> flat((size_t)g->numel, 0.f);
    for (auto& nme : g->names) memcpy(flat.data() + g->off[nme].first, t->host[nme].data(), t->host[nme].size() * sizeof(float));
    THIP(hipMemcpy(g->p, flat.data(), flat.size() * sizeof(float), hipMemcpyHostToDevice));
  }
  {  // MADE x2: never trained -> pre-masked constants (layers.py:671,703)
    std::vector<float> w((size_t)6 * Z * Z), b((size_t)6 * Z);
    int k = 0;
    for (const char* m : {"l_IAF_mu", "l_IAF_ls"}) {
      int j = 0;
      for (const char* l : {"_input", "_output_W", "_output_D"}) {
        const std::vector<float>& W = t->host[std::string(m) + l + ".W"];
        for (int i = 0; i < Z * Z; ++i) w[(size_t)k * Z * Z + i] = W[i] * t->masks[j][i];
        memcpy(b.data() + (size_t)k * Z, t->host[std::string(m) + l + ".b"].data(), Z * sizeof(float));
        ++k; ++j;
      }
    }
    t->made_w = dalloc(t, w.size()); t->made_b = dalloc(t, b.size());
    THIP(hipMemcpy(t->made_w, w.data(), w.size() * sizeof(float), hipMemcpyHostToDevice));
    THIP(hipMemcpy(t->made_b, b.data(), b.size() * sizeof(float), hipMemcpyHostToDevice));
  }
  {  // l_dec_fc2 output is stored (H,W,C) while its bias is indexed (C,H,W) (App. B.6)
    const int Cc = 512, Hh = 4, Ww = 4;
    std::vector<int32_t> perm(Cc * Hh * Ww), inv(Cc * Hh * Ww);
    for (int hh = 0; hh < Hh; ++hh)
      for (int ww = 0; ww < Ww; ++ww)
        for (int c = 0; c < Cc; ++c) perm[(hh * Ww + ww) * Cc + c] = (c * Hh + hh) * Ww + ww;  // hwc position -> chw index
    for (size_t i = 0; i < perm.size(); ++i) inv[perm[i]] = (int32_t)i;                         // chw index -> hwc position
    t->fc2_perm = (int32_t*)dalloc(t, perm.size()); t->fc2_inv = (int32_t*)dalloc(t, inv.size());
    THIP(hipMemcpy(t->fc2_perm, perm.data(), perm.size() * 4, hipMemcpyHostToDevice));
    THIP(hipMemcpy(t->fc2_inv, inv.data(), inv.size() * 4, hipMemcpyHostToDevice));
  }
  t->fc2_bias = dalloc(t, 8192); t->fc2_db = dalloc(t, 8192); t->fc2_tmp = dalloc(t, 8192); t->tmp_big = dalloc64(t, 2 * 8192);
  t->ws_stats_cap = (size_t)256 * 2 * 8192;
  t->ws_stats = dalloc64(t, t->ws_stats_cap);
  t->ws_loss = dalloc(t, 1024 * 2); t->scalars = dalloc(t, 64);
  t->mb_W = dalloc(t, (size_t)1024 * 2500); t->mb_dW = dalloc(t, (size_t)1024 * 2500); t->mb_colscale = dalloc(t, 2500);
  t->tmp_vals = dalloc64(t, 2048); t->ortho_vals = dalloc(t, 2048);
  if ((rc = build_layers(t))) return rc;
  enc_alloc(t, t->EX, t->bnEX); enc_alloc(t, t->EH, t->bnEH); enc_alloc(t, t->EG, t->bnEG);
  z_alloc(t);
  dec_alloc(t, t->DZ, t->bnDZ); dec_alloc(t, t->DG, t->bnDG);
  t->zgen = dalloc(t, (size_t)t->n * 128); t->zgen0 = dalloc(t, (size_t)t->n * 128);
  t->xin = dalloc(t, (size_t)t->n * 3 * 4096); t->zin = dalloc(t, (size_t)t->n * 100); t->epsin = dalloc(t, (size_t)t->n * 100);
  if (t->oom) return tfail(t, -20, "out of device memory while allocating the training step's buffers (batch %d per GPU)", t->n);
  if (const char* e = getenv("IAN_WGRAD_PRIORITY")) t->wgrad_priority = atoi(e);
#ifdef IAN_ABLATION
  if (const char* e = getenv("IAN_DUAL_CHAIN_TIMING")) t->dual_timing = atoi(e);   // timing-only, results wrong: libian_ablation.so only
#endif
  if (make_wgrad_stream(t) != 0) t->st2 = nullptr;   // no second stream: weight gradients stay on the compute stream
  if (t->world > 1) THIP(hipStreamCreateWithFlags(&t->st_comm, hipStreamNonBlocking));
  if (t->exact && ((t->n & (t->n - 1)) || (t->world & (t->world - 1))))
    fprintf(stderr, "libian: per-rank batch %d x world %d is not a power of two: the partial-sum tree of the batch statistics associates "
                    "differently on N ranks and in one process, so the data-parallel step is NOT guaranteed bit-identical to the "
                    "single-process step (float64 sums: the float32 statistics can differ in the last bit); train_IAN.py's "
                    "semantics are unaffected\n", t->n, t->world);
  t->dirty = {"enc", "Z", "dec"};
  t->host.clear();
  t->finalized = true;
  return 0;
}

int ian_trainer_set_comm(ian_trainer* t, const ian_comm_ops* ops, int32_t exact) {
  if (!t || !ops) return -1;
  if (t->finalized) return tfail(t, -6, "ian_trainer_set_comm must be called before ian_trainer_finalize (buffer sizes depend on the world size)");
  if (ops->world < 1 || ops->rank < 0 || ops->rank >= ops->world) return tfail(t, -1, "ian_trainer_set_comm: bad world / rank (%d / %d)", ops->world, ops->rank);
  if (ops->world > 1 && (!ops->allreduce_sum || !ops->wait_all || !ops->allgather)) return tfail(t, -1, "ian_trainer_set_comm: a callback is NULL");
  t->comm = *ops;
  t->world = ops->world;
  t->rank = ops->rank;
  t->exact = (exact && ops->world > 1) ? 1 : 0;
  t->N = t->n * t->world;
  return 0;
}

}  // extern "C"

namespace {
// entry bookkeeping shared by ian_train_step and the piecewise entries: stream hand-over guard + host -> device staging
int enter(ian_trainer* t, void* stream) {
  if (!t) return -1;
  if (!t->finalized) return tfail(t, -6, "ian_trainer_finalize has not been called");
  hipStream_t st = (hipStream_t)stream;
  if (t->have_last_stream && t->last_stream != st) THIP(hipStreamSynchronize(t->last_stream));  // buffers are shared between calls
  t->st = st;
  t->last_stream = st;
  t->have_last_stream = true;
  return 0;
}
int stage_inputs(ian_trainer* t, const float** x, const float** zrand, const float** eps, int n) {
  const std::pair<const float**, std::pair<float*, size_t>> ins[3] = {{x, {t->xin, (size_t)n * 3 * 4096}}, {zrand, {t->zin, (size_t)n * 100}},
                                                                      {eps, {t->epsin, (size_t)n * 100}}};
  for (auto& in : ins)
    if (!is_device_ptr(*in.first)) {
      THIP(hipMemcpyAsync(in.second.first, *in.first, in.second.second * sizeof(float), hipMemcpyHostToDevice, t->st));
      *in.first = in.second.first;
    }
  return 0;
}
int check_step_args(ian_trainer* t, const float* x, const float* zrand, const float* eps, int n, const char* who) {
  if (!x || !zrand || !eps) return tfail(t, -1, "null pointer passed to %s", who);
  if (n != t->n) return tfail(t, -7, "%s: batch %d, the trainer was created for %d per GPU (batch statistics are per minibatch)", who, n, t->n);
  return 0;
}
int step_body(ian_trainer* t, bool gen, const float* x, const float* zrand, const float* eps, float* metrics9) {
  int rc;
  if ((rc = forward(t, x, zrand, eps))) return rc;
  if ((metrics9 || t->world > 1) && (rc = metrics_device(t))) return rc;
  if (metrics9 && (rc = metrics_read(t, metrics9))) return rc;
  if ((rc = backward(t, gen))) return rc;
  if ((rc = finish_allreduce(t, gen ? 0 : 1))) return rc;
  if ((rc = regularizers(t, gen))) return rc;
  if ((rc = adam(t, gen ? t->dec : t->enc, gen ? "dec" : "enc"))) return rc;
  return adam(t, t->zp, "Z");
}
// error path of any entry that may have issued work on the trainer's own streams: nothing of it may still be running (and
// writing the shared buffers) when the caller sees the error code
void quiesce(ian_trainer* t) {
  const std::string keep = t->err;
  if (t->st2) (void)hipStreamSynchronize(t->st2);
  if (t->st_comm) (void)hipStreamSynchronize(t->st_comm);
  (void)hipStreamSynchronize(t->st);
  (void)hipGetLastError();
  t->ev_used = 0;
  t->buckets = nullptr;
  t->err = keep;
}
}  // namespace

extern "C" {

/* One update of train_IAN.py:309-329.  which: 0 = update_gen, 1 = update_discrim.  x (n,3,64,64) in [-1,1], zrand (n,100),
   eps (n,100): host or device pointers; n = the PER-RANK batch.  metrics: NULL or 9 HOST floats = discrim_d_loss, gen_recon_loss,
   gen_sample_loss, discrim_g_loss, discrim_acc, kl_div, pixel_loss, pixel_acc, feature_loss of THIS (global) minibatch before the
   update (reading them synchronises the stream). */
int ian_train_step(ian_trainer* t, int32_t which, const float* x, const float* zrand, const float* eps, int32_t n, float* metrics9, void* stream) {
  int rc;
  if ((rc = enter(t, stream))) return rc;
  if ((rc = check_step_args(t, x, zrand, eps, n, "ian_train_step"))) return rc;
  if (which != 0 && which != 1) return tfail(t, -7, "ian_train_step: which must be 0 (update_gen) or 1 (update_discrim)");
  t->xhat_override = t->xgen_override = nullptr;
  if ((rc = stage_inputs(t, &x, &zrand, &eps, n))) return rc;
  rc = step_body(t, which == 0, x, zrand, eps, metrics9);
  if (rc) quiesce(t);
  return rc;
}

/* ---- the same step in pieces (tests, diagnostics): forward / metrics / backward / all-reduce / regularizers / Adam --------- */
int ian_trainer_forward(ian_trainer* t, const float* x, const float* zrand, const float* eps, int32_t n, const float* xhat_override,
                        const float* xgen_override, void* stream) {
  int rc;
  if ((rc = enter(t, stream))) return rc;
  if ((rc = check_step_args(t, x, zrand, eps, n, "ian_trainer_forward"))) return rc;
  if ((xhat_override && !is_device_ptr(xhat_override)) || (xgen_override && !is_device_ptr(xgen_override)))
    return tfail(t, -1, "ian_trainer_forward: the override images must be device buffers");
  if ((rc = stage_inputs(t, &x, &zrand, &eps, n))) return rc;
  t->xhat_override = xhat_override;
  t->xgen_override = xgen_override;
  rc = forward(t, x, zrand, eps);
  t->xhat_override = t->xgen_override = nullptr;
  if (rc) quiesce(t);
  return rc;
}
int ian_trainer_metrics(ian_trainer* t, float* metrics9) {
  if (!t || !t->finalized || !metrics9) return -1;
  if (!t->X) return tfail(t, -6, "ian_trainer_metrics: no forward pass has run");
  return metrics(t, metrics9);
}
int ian_trainer_backward(ian_trainer* t, int32_t which) {
  if (!t || !t->finalized || (which != 0 && which != 1)) return -1;
  if (!t->X) return tfail(t, -6, "ian_trainer_backward: no forward pass has run");
  const int rc = backward(t, which == 0);
  if (rc) quiesce(t);
  return rc;
}
int ian_trainer_finish_allreduce(ian_trainer* t, int32_t which) {
  if (!t || !t->finalized || (which != 0 && which != 1)) return -1;
  const int rc = finish_allreduce(t, which);
  if (rc) quiesce(t);
  return rc;
}
int ian_trainer_regularizers(ian_trainer* t, int32_t which) {
  if (!t || !t->finalized || (which != 0 && which != 1)) return -1;
  int rc = join_side_stream(t);
  return rc ? rc : regularizers(t, which == 0);
}
int ian_trainer_apply_adam(ian_trainer* t, int32_t which) {
  if (!t || !t->finalized || (which != 0 && which != 1)) return -1;
  int rc = join_side_stream(t);
  if (rc) return rc;
  if ((rc = adam(t, which == 0 ? t->dec : t->enc, which == 0 ? "dec" : "enc"))) return rc;
  return adam(t, t->zp, "Z");
}
/* One encoder backward sweep of pass 0 = encoder(X), 1 = encoder(X_hat), 2 = encoder(X_gen) with cross-entropy seeds
   dlogits = w0 (p - onehot(t0)) + w1 (p - onehot(t1)) (t < 0: none); reset != 0 starts a fresh gradient sweep first. */
int ian_trainer_enc_backward(ian_trainer* t, int32_t pass, int32_t t0, float w0, int32_t t1, float w1, int32_t feature_seeded, int32_t want_w,
                             int32_t want_dx, int32_t reset) {
  if (!t || !t->finalized || pass < 0 || pass > 2) return -1;
  if (reset) {
    int rc = join_side_stream(t);
    if (rc) return rc;
    t->touched.clear();
    t->evlog.clear();
    t->buckets = nullptr;
  }
  Bufs& E = pass == 0 ? t->EX : (pass == 1 ? t->EH : t->EG);
  auto& bn = pass == 0 ? t->bnEX : (pass == 1 ? t->bnEH : t->bnEG);
  int rc = enc_backward(t, E, bn, t0, w0, t1, w1, feature_seeded != 0, want_w != 0, want_dx != 0);
  if (!rc) rc = join_side_stream(t);
  if (rc) quiesce(t);
  return rc;
}
/* Device address of an internal buffer: "<pass>.<name>" with pass in EX EH EG ZS DZ DG (activations and gradients, NHWC with the
   channel stride rounded up to 32; images NCHW), "<pass>.<bn>.<mean|inv_std|scale|shift>" (float32 [C]),
   "<pass>.<bn>.<sums|bsums>" (float64 [2][C]; numel counts doubles), "scalars", "ws_loss".  Tests / diagnostics only. */
int ian_trainer_buffer(ian_trainer* t, const char* name, void** ptr, int64_t* numel) {
  if (!t || !t->finalized || !name || !ptr) return -1;
  const std::string s = name;
  if (s == "scalars") { *ptr = t->scalars; if (numel) *numel = 64; return 0; }
  if (s == "ws_loss") { *ptr = t->ws_loss; if (numel) *numel = 2048; return 0; }
  const size_t d = s.find('.');
  if (d == std::string::npos) return tfail(t, -2, "unknown buffer '%s'", name);
  const std::string ps = s.substr(0, d), rest = s.substr(d + 1);
  Bufs* B = nullptr;
  std::map<std::string, BN>* bn = nullptr;
  if (ps == "EX") { B = &t->EX; bn = &t->bnEX; } else if (ps == "EH") { B = &t->EH; bn = &t->bnEH; } else if (ps == "EG") { B = &t->EG; bn = &t->bnEG; }
  else if (ps == "ZS") { B = &t->ZS; bn = &t->bnZ; } else if (ps == "DZ") { B = &t->DZ; bn = &t->bnDZ; } else if (ps == "DG") { B = &t->DG; bn = &t->bnDG; }
  else return tfail(t, -2, "unknown pass in buffer name '%s'", name);
  auto it = B->find(rest);
  if (it != B->end()) {
    *ptr = it->second;
    if (numel) {
      auto sz = t->alloc_floats.find(it->second);
      *numel = sz != t->alloc_floats.end() ? (int64_t)sz->second : -1;
    }
    return 0;
  }
  const size_t d2 = rest.rfind('.');
  if (d2 != std::string::npos) {
    auto bi = bn->find(rest.substr(0, d2));
    if (bi != bn->end()) {
      const std::string f = rest.substr(d2 + 1);
      BN& b = bi->second;
      void* p = f == "mean" ? (void*)b.mean : f == "inv_std" ? (void*)b.inv_std : f == "scale" ? (void*)b.scale : f == "shift" ? (void*)b.shift
                : f == "sums" ? (void*)b.sums : f == "bsums" ? (void*)b.bsums : nullptr;
      if (p) {
        *ptr = p;
        if (numel) *numel = (f == "sums" || f == "bsums") ? 2 * b.C : b.C;
        return 0;
      }
    }
  }
  return tfail(t, -2, "unknown buffer '%s'", name);
}
/* group 0 encoder_params, 1 Z_params, 2 decoder_params, 3 batch-norm running averages: the flat device buffers (reference layouts). */
int ian_trainer_group(ian_trainer* t, int32_t group, float** p, float** g, float** m, float** v, int64_t* numel) {
  if (!t || !t->finalized || group < 0 || group > 3) return -1;
  Group& gr = group == 3 ? t->stats : group_of(t, group);
  if (p) *p = gr.p;
  if (g) *g = gr.g;
  if (m) *m = gr.m;
  if (v) *v = gr.v;
  if (numel) *numel = gr.numel;
  return 0;
}
int ian_trainer_param_info(ian_trainer* t, const char* name, int32_t* group, int64_t* offset, int64_t* numel) {
  if (!t || !name) return -1;
  auto it = t->where.find(name);
  if (it == t->where.end()) return tfail(t, -2, "unknown parameter '%s'", name);
  Group* g = it->second;
  if (group) *group = g == &t->stats ? 3 : group_index(t, g);
  if (offset) *offset = g->off.at(name).first;
  if (numel) *numel = g->off.at(name).second.numel();
  return 0;
}
/* Parameters of `group` were written behind the trainer's back (tests, checkpoint loading): repack before the next forward. */
int ian_trainer_mark_dirty(ian_trainer* t, int32_t group) {
  if (!t || group < 0 || group > 2) return -1;
  t->dirty.insert(group == 0 ? "enc" : (group == 1 ? "Z" : "dec"));
  return 0;
}
/* "exposed_ms_gen" / "exposed_ms_discrim": mean stall of the compute stream on the gradient all-reduce per update of that kind
   (option measure_exposed = 1; synchronises); "plan_buckets_gen|discrim"; "overlap_log" (entries kept); "world", "rank", "exact",
   "global_batch". */
int ian_trainer_stat(ian_trainer* t, const char* key, double* out) {
  if (!t || !key || !out) return -1;
  const std::string k = key;
  if (k == "exposed_ms_gen" || k == "exposed_ms_discrim") {
    if (t->ex_pending) {
      float ms = 0.f;
      if (hipEventSynchronize(t->ex1) == hipSuccess && hipEventElapsedTime(&ms, t->ex0, t->ex1) == hipSuccess) {
        t->exposed_ms[t->ex_which] += ms;
        t->exposed_n[t->ex_which] += 1;
      }
      (void)hipGetLastError();
      t->ex_pending = false;
    }
    const int w = k == "exposed_ms_gen" ? 0 : 1;
    *out = t->exposed_n[w] ? t->exposed_ms[w] / t->exposed_n[w] : 0.0;
  } else if (k == "gather_ms_gen" || k == "gather_ms_discrim" || k == "gathers_gen" || k == "gathers_discrim") {
    THIP(hipDeviceSynchronize());
    fold_gathers(t, 0);
    fold_gathers(t, 1);
    const int w = (k == "gather_ms_gen" || k == "gathers_gen") ? 0 : 1;
    const double d = t->gather_n[w] ? (double)t->gather_n[w] : 1.0;
    *out = (k.compare(0, 10, "gather_ms_") == 0 ? t->gather_ms[w] : t->gather_calls[w]) / d;
  } else if (k == "plan_buckets_gen") *out = t->plans.count(0) ? (double)t->plans[0].size() : 0.0;
  else if (k == "plan_buckets_discrim") *out = t->plans.count(1) ? (double)t->plans[1].size() : 0.0;
  else if (k == "overlap_log") *out = (double)t->overlap_log.size();
  else if (k == "world") *out = t->world;
  else if (k == "rank") *out = t->rank;
  else if (k == "exact") *out = t->exact;
  else if (k == "global_batch") *out = t->N;
  else return tfail(t, -1, "unknown statistic '%s'", key);
  return 0;
}
/* rec[6] = which, group, first element, bytes, index of the gradient write after which the bucket was handed over, number of
   gradient writes of that backward sweep (0 while the sweep is still open) */
int ian_trainer_overlap_log(ian_trainer* t, int32_t index, int64_t* rec) {
  if (!t || !rec || index < 0 || (size_t)index >= t->overlap_log.size()) return -1;
  const OverlapRec& r = t->overlap_log[index];
  rec[0] = r.which; rec[1] = r.group; rec[2] = r.lo; rec[3] = r.bytes; rec[4] = r.issued_at_write; rec[5] = r.writes_in_backward;
  return 0;
}

int ian_trainer_autotune(ian_trainer* t, void* stream) {
  int rc;
  if ((rc = enter(t, stream))) return rc;
  const size_t need = (size_t)t->n * 64 * 64 * 128;  // the largest activation of IAN.py (dec_conv4 output)
  std::vector<float> rnd(need);
  unsigned s = 12345u;
  for (auto& v : rnd) {
    s = s * 1664525u + 1013904223u;
    v = ((s >> 8) & 0xFFFF) / 32768.f - 1.f;
  }
  float *a = nullptr, *b = nullptr;
  THIP(hipMalloc((void**)&a, need * sizeof(float)));
  if (hipMalloc((void**)&b, need * sizeof(float)) != hipSuccess) {
    (void)hipGetLastError();
    (void)hipFree(a);
    return tfail(t, -20, "out of device memory (autotune operands)");
  }
  rc = 0;
  if (hipMemcpy(a, rnd.data(), need * sizeof(float), hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(b, rnd.data(), need * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
    (void)hipGetLastError();
    rc = tfail(t, -20, "hipMemcpy of the autotune operands failed");
  }
  if (!rc) rc = refresh_weights(t);
  for (auto& key : t->layer_order) {
    if (rc) break;
    rc = ian_layer_autotune(t->layers[key].l, t->n, a, b, (int64_t)need, t->st);
    if (rc) tfail(t, rc, "ian_layer_autotune(%s) failed (%d): %s", key.c_str(), rc, ian_layer_last_error(t->layers[key].l));
  }
  (void)hipStreamSynchronize(t->st);
  (void)hipFree(a);
  (void)hipFree(b);
  return rc;
}

/* Copy a parameter (trainable, batch-norm running average) or, grad != 0, its gradient of the last step to the host. */
int ian_trainer_read_param(ian_trainer* t, const char* name, int32_t grad, float* out, int64_t numel) {
  if (!t || !t->finalized || !name || !out) return -1;
  auto it = t->where.find(name);
  if (it == t->where.end()) return tfail(t, -2, "unknown parameter '%s'", name);
  Group* g = it->second;
  if (grad && !g->g) return tfail(t, -2, "'%s' has no gradient (not trainable)", name);
  const auto& o = g->off.at(name);
  if (o.second.numel() != numel) return tfail(t, -3, "parameter '%s' has %lld elements", name, (long long)o.second.numel());
  THIP(hipDeviceSynchronize());   // the compute stream, the weight-gradient stream and the bucket stream
  THIP(hipMemcpy(out, (grad ? g->g : g->p) + o.first, numel * sizeof(float), hipMemcpyDeviceToHost));
  return 0;
}

int ian_trainer_set_option(ian_trainer* t, const char* key, double value) {
  if (!t || !key) return -1;
  const std::string k = key;
  if (k == "learning_rate") t->cfg.learning_rate = value;  // train_IAN.py:523-527 learning-rate schedule
  else if (k == "head6") t->head6 = value != 0.0;
  else if (k == "overlap_wgrad") t->overlap_wgrad = value != 0.0;
  else if (k == "wgrad_priority") {      // 0: default-priority weight-gradient stream, 1: lowest priority (see make_wgrad_stream)
    (void)hipDeviceSynchronize();
    t->wgrad_priority = value != 0.0;
    if (t->finalized) (void)make_wgrad_stream(t);
  }
  else if (k == "update_running") t->update_running = value != 0.0;
  else if (k == "fused_stats") t->fused_stats = value != 0.0;                 // GEMM-epilogue batch statistics (single-process step)
  else if (k == "overlap") t->overlap = value != 0.0;                       // gradient buckets handed over during backward
  else if (k == "bucket_bytes") { t->bucket_bytes = (int64_t)value > 4 ? (int64_t)value : 4; t->plans.clear(); }
  else if (k == "measure_exposed") {
    (void)hipDeviceSynchronize();
    t->measure_exposed = value != 0.0; t->exposed_ms[0] = t->exposed_ms[1] = 0; t->exposed_n[0] = t->exposed_n[1] = 0; t->ex_pending = false;
    for (int w = 0; w < 2; ++w) { t->gev_used[w] = 0; t->gather_ms[w] = t->gather_calls[w] = 0; t->gather_n[w] = 0; }
  }
  else return tfail(t, -1, "unknown option '%s'", key);
  return 0;
}

int32_t ian_trainer_adam_steps(ian_trainer* t, int32_t group) {  /* 0 encoder_params, 1 Z_params, 2 decoder_params */
  if (!t) return -1;
  return group == 0 ? t->enc.t : (group == 1 ? t->zp.t : t->dec.t);
}

const char* ian_trainer_last_error(ian_trainer* t) { return t ? t->err.c_str() : "null trainer"; }

void ian_trainer_destroy(ian_trainer* t) {
  if (!t) return;
  IAN_GUARD_CHECK("ian_trainer_destroy");
  (void)hipDeviceSynchronize();
  for (hipEvent_t e : t->events) (void)hipEventDestroy(e);
  if (t->st2) (void)hipStreamDestroy(t->st2);
  if (t->st_comm) (void)hipStreamDestroy(t->st_comm);
  if (t->ex0) (void)hipEventDestroy(t->ex0);
  if (t->ex1) (void)hipEventDestroy(t->ex1);
  for (auto& pool : t->gev)
    for (auto& e : pool) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
  for (auto& kv : t->layers) ian_layer_destroy(kv.second.l);
  for (float* p : t->allocs)
    if (p) (void)hipFree(p);
  delete t;
}

}  // extern "C"

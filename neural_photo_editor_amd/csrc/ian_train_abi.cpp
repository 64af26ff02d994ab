// C-ABI wrappers (include/ian_train.h, ian_k_*) over the element-wise / reduction kernels of the training step.
// No state: every call is a launch sequence on the caller's stream over caller-owned device buffers.
#include <string>

#include "../../include/ian_train.h"
#include "ian_internal.h"
#include "ian_guard.h"

using namespace ian;

namespace {
thread_local std::string g_err;
int chk(hipError_t e, const char* what) {
  if (e == hipSuccess) return 0;
  g_err = std::string(what) + ": " + hipGetErrorString(e);
  (void)hipGetLastError();
  return -2;
}
int bad(const char* what) {
  g_err = std::string(what) + ": bad argument";
  return -1;
}
}  // namespace

#define ST ((hipStream_t)stream)

extern "C" {

const char* ian_k_last_error(void) { return g_err.c_str(); }

int ian_k_colstats(int32_t mode, const float* x, const float* a, const float* y, const float* mean, const float* inv_std,
                   int64_t rows, int32_t C, int32_t stride, int32_t act, double* workspace, int32_t nchunks, double* sums,
                   void* stream) {
  if (!x || !workspace || !sums || rows <= 0 || C <= 0 || (C & 3) || nchunks <= 0 || mode < 0 || mode > 2) return bad("ian_k_colstats");
  if (mode >= 1 && act != IAN_ACT_NONE && !a) return bad("ian_k_colstats (activation output missing)");
  if (mode == 1 && (!y || !mean || !inv_std)) return bad("ian_k_colstats (batch-norm inputs missing)");
  ColStatsArgs s;
  s.x = x; s.a = a; s.y = y; s.mean = mean; s.inv_std = inv_std; s.partial = workspace; s.rows = rows; s.C = C;
  s.stride = stride; s.mode = mode; s.act = act;
  if (nchunks > rows) nchunks = (int32_t)rows;
  return chk(launch_colstats(s, nchunks, sums, ST), "ian_k_colstats");
}

int ian_k_tree_sum(const double* partial, int32_t count, int32_t width, double* out, void* stream) {
  if (!partial || !out || count <= 0 || width <= 0 || count >= 65536) return bad("ian_k_tree_sum");
  return chk(launch_tree_sum(partial, count, width, out, ST), "ian_k_tree_sum");
}

int ian_k_bn_make_affine(const double* sums, float count, float eps, const float* gamma, const float* beta, int32_t C,
                         float* mean, float* inv_std, float* scale, float* shift, void* stream) {
  if (!sums || !gamma || !beta || !mean || !inv_std || !scale || !shift || C <= 0 || count <= 0) return bad("ian_k_bn_make_affine");
  return chk(launch_bn_make_affine(sums, count, eps, gamma, beta, C, mean, inv_std, scale, shift, ST), "ian_k_bn_make_affine");
}

int ian_k_bn_running(float* run_mean, const float* mean, float* run_inv_std, const float* inv_std, int32_t C, float keep, float alpha,
                     void* stream) {
  if (!run_mean || !mean || !run_inv_std || !inv_std || C <= 0) return bad("ian_k_bn_running");
  return chk(launch_bn_running(run_mean, mean, run_inv_std, inv_std, C, keep, alpha, ST), "ian_k_bn_running");
}

int ian_k_bn_stats_affine(const float* y, int64_t rows, int32_t C, int32_t stride, double* workspace, int32_t nchunks, double* sums,
                          float count, float eps, const float* gamma, const float* beta, float* mean, float* inv_std, float* scale,
                          float* shift, float* run_mean, float* run_inv_std, float keep, float alpha, void* stream) {
  if (!y || !workspace || !sums || !gamma || !beta || !mean || !inv_std || !scale || !shift || rows <= 0 || C <= 0 || (C & 3) ||
      nchunks <= 0 || count <= 0 || (!run_mean) != (!run_inv_std))
    return bad("ian_k_bn_stats_affine");
  ColStatsArgs s;
  s.x = y; s.a = nullptr; s.y = nullptr; s.mean = nullptr; s.inv_std = nullptr; s.partial = workspace; s.rows = rows; s.C = C;
  s.stride = stride; s.mode = 0; s.act = 0;
  if (nchunks > rows) nchunks = (int32_t)rows;
  return chk(launch_bn_stats_affine(s, nchunks, sums, count, eps, gamma, beta, mean, inv_std, scale, shift, run_mean, run_inv_std, keep,
                                    alpha, ST),
             "ian_k_bn_stats_affine");
}

int ian_k_bn_finish(const double* workspace, int32_t nchunks, int32_t C, double* sums, float count, float eps, const float* gamma,
                    const float* beta, float* mean, float* inv_std, float* scale, float* shift, float* run_mean, float* run_inv_std,
                    float keep, float alpha, void* stream) {
  if (!workspace || !sums || !gamma || !beta || !mean || !inv_std || !scale || !shift || C <= 0 || nchunks <= 0 || count <= 0 ||
      (!run_mean) != (!run_inv_std))
    return bad("ian_k_bn_finish");
  return chk(launch_bn_finish(workspace, nchunks, C, sums, count, eps, gamma, beta, mean, inv_std, scale, shift, run_mean, run_inv_std, keep,
                              alpha, ST), "ian_k_bn_finish");
}
int ian_k_bn_bwd_finish(const double* workspace, int32_t nchunks, int32_t C, double* sums, float* gbeta, int32_t acc_beta, float* ggamma,
                        int32_t acc_gamma, void* stream) {
  if (!workspace || !sums || C <= 0 || nchunks <= 0 || (!gbeta) != (!ggamma)) return bad("ian_k_bn_bwd_finish");
  return chk(launch_bn_bwd_finish(workspace, nchunks, C, sums, gbeta, acc_beta, ggamma, acc_gamma, ST), "ian_k_bn_bwd_finish");
}

int ian_k_bn_bwd_stats(const float* dA, const float* a, const float* y, const float* mean, const float* inv_std, int64_t rows, int32_t C,
                       int32_t stride, int32_t act, double* workspace, int32_t nchunks, double* sums, float* gbeta, int32_t acc_beta,
                       float* ggamma, int32_t acc_gamma, void* stream) {
  if (!dA || !y || !mean || !inv_std || !workspace || !sums || rows <= 0 || C <= 0 || (C & 3) || nchunks <= 0 || (!gbeta) != (!ggamma))
    return bad("ian_k_bn_bwd_stats");
  if (act != IAN_ACT_NONE && !a) return bad("ian_k_bn_bwd_stats (activation output missing)");
  ColStatsArgs s;
  s.x = dA; s.a = a; s.y = y; s.mean = mean; s.inv_std = inv_std; s.partial = workspace; s.rows = rows; s.C = C;
  s.stride = stride; s.mode = 1; s.act = act;
  if (nchunks > rows) nchunks = (int32_t)rows;
  return chk(launch_bn_bwd_stats(s, nchunks, sums, gbeta, acc_beta, ggamma, acc_gamma, ST), "ian_k_bn_bwd_stats");
}

int ian_k_affine(const float* x, float* y, const float* scale, const float* shift, int64_t rows, int32_t C, int32_t stride,
                 int32_t act, void* stream) {
  if (!x || !y || rows <= 0 || C <= 0 || (C & 3)) return bad("ian_k_affine");
  return chk(launch_affine(x, y, scale, shift, rows, C, stride, act, ST), "ian_k_affine");
}

int ian_k_bn_bwd(const float* dA, const float* a, const float* y, const float* mean, const float* inv_std,
                 const float* scale, const double* sums, float count, float* dy, int64_t rows, int32_t C, int32_t stride,
                 int32_t act, void* stream) {
  if (!dA || !dy || rows <= 0 || C <= 0 || (C & 3)) return bad("ian_k_bn_bwd");
  if (act != IAN_ACT_NONE && !a) return bad("ian_k_bn_bwd (activation output missing)");
  if (sums && (!y || !mean || !inv_std || !scale || count <= 0)) return bad("ian_k_bn_bwd (batch-norm inputs missing)");
  BnBwdArgs b;
  b.dA = dA; b.a = a; b.y = y; b.mean = mean; b.inv_std = inv_std; b.scale = scale; b.sums = sums; b.dy = dy; b.rows = rows;
  b.C = C; b.stride = stride; b.act = act; b.count = count > 0 ? count : 1.f;
  return chk(launch_bn_bwd_apply(b, ST), "ian_k_bn_bwd");
}

int ian_k_axpy(float alpha, const float* x, float* y, int64_t n, int32_t accumulate, void* stream) {
  if (!x || !y || n <= 0) return bad("ian_k_axpy");
  return chk(launch_axpy(alpha, x, y, n, accumulate, ST), "ian_k_axpy");
}

int ian_k_axpy_f64(double alpha, const double* x, float* y, int64_t n, int32_t accumulate, void* stream) {
  if (!x || !y || n <= 0) return bad("ian_k_axpy_f64");
  return chk(launch_axpy_f64(alpha, x, y, n, accumulate, ST), "ian_k_axpy_f64");
}

int ian_k_gather(const float* src, const int32_t* map, float* dst, int64_t count, void* stream) {
  if (!src || !map || !dst || count <= 0) return bad("ian_k_gather");
  return chk(launch_gather_pack(src, map, dst, count, ST), "ian_k_gather");
}

int ian_k_nchw_to_nhwc(const float* src, float* dst, int32_t n, int32_t hw, int32_t c, int32_t stride, void* stream) {
  if (!src || !dst || n <= 0) return bad("ian_k_nchw_to_nhwc");
  return chk(launch_nchw_to_nhwc(src, dst, n, hw, c, stride, ST), "ian_k_nchw_to_nhwc");
}

int ian_k_nhwc_to_nchw(const float* src, int32_t stride, float* dst, int32_t n, int32_t hw, int32_t c, void* stream) {
  if (!src || !dst || n <= 0) return bad("ian_k_nhwc_to_nchw");
  return chk(launch_nhwc_to_nchw(src, stride, dst, n, hw, c, ST), "ian_k_nhwc_to_nchw");
}

int ian_k_globalpool(const float* x, float* y, int32_t n, int32_t hw, int32_t C, int32_t xs, int32_t ys, void* stream) {
  if (!x || !y || n <= 0) return bad("ian_k_globalpool");
  return chk(launch_globalpool(x, y, n, hw, C, xs, ys, ST), "ian_k_globalpool");
}

int ian_k_globalpool_bwd(const float* dy, float* dx, int32_t n, int32_t hw, int32_t C, int32_t xs, int32_t ys,
                         int32_t accumulate, void* stream) {
  if (!dy || !dx || n <= 0) return bad("ian_k_globalpool_bwd");
  return chk(launch_globalpool_bwd(dy, dx, n, hw, C, xs, ys, accumulate, ST), "ian_k_globalpool_bwd");
}

int ian_k_mb_weight(const float* theta, const float* lws, float* W, float* colscale, int32_t nin, int32_t ncol, void* stream) {
  if (!theta || !lws || !W || !colscale) return bad("ian_k_mb_weight");
  return chk(launch_mb_weight(theta, lws, W, colscale, nin, ncol, ST), "ian_k_mb_weight");
}

int ian_k_mb_weight_bwd(const float* theta, const float* colscale, const float* dW, float* dtheta, float* dlws, int32_t nin,
                        int32_t ncol, int32_t accumulate, void* stream) {
  if (!theta || !colscale || !dW || !dtheta || !dlws) return bad("ian_k_mb_weight_bwd");
  return chk(launch_mb_weight_bwd(theta, colscale, dW, dtheta, dlws, nin, ncol, accumulate, ST), "ian_k_mb_weight_bwd");
}

int ian_k_mb_forward(const float* act_all, int32_t nall, int32_t as, int32_t row0, int32_t n, int32_t nk, int32_t nd,
                     const float* bias, const float* feat, int32_t fs, int32_t fin, float* mb, int32_t ms, void* stream) {
  if (!act_all || !bias || !feat || !mb || n <= 0 || row0 < 0 || row0 + n > nall) return bad("ian_k_mb_forward");
  return chk(launch_mb_forward(act_all, nall, as, row0, n, nk, nd, bias, feat, fs, fin, mb, ms, ST), "ian_k_mb_forward");
}

int ian_k_mb_backward(const float* act_all, int32_t nall, int32_t as, int32_t row0, int32_t n, int32_t nk, int32_t nd,
                      const float* df_all, int32_t dfs, float* dact, int32_t das, void* stream) {
  if (!act_all || !df_all || !dact || n <= 0 || row0 < 0 || row0 + n > nall) return bad("ian_k_mb_backward");
  return chk(launch_mb_backward(act_all, nall, as, row0, n, nk, nd, df_all, dfs, dact, das, ST), "ian_k_mb_backward");
}

int ian_k_disc_head(const float* mb, int32_t ms, int32_t nfeat, const float* Wd, int32_t n, int32_t target0, int32_t target1,
                    int32_t acc_target, float* p, float* loss, void* stream) {
  if (!mb || !Wd || !p || !loss || n <= 0) return bad("ian_k_disc_head");
  DiscHeadArgs a;
  a.p = p; a.loss = loss; a.target[0] = target0; a.target[1] = target1; a.acc_target = acc_target;
  return chk(launch_disc_head(mb, ms, nfeat, Wd, 3, n, a, ST), "ian_k_disc_head");
}

int ian_k_disc_head_bwd(const float* p, const float* Wd, int32_t nfeat, int32_t n, int32_t t0, float w0, int32_t t1, float w1,
                        float* dlogits, float* dmb, int32_t ms, void* stream) {
  if (!p || !Wd || !dlogits || !dmb || n <= 0) return bad("ian_k_disc_head_bwd");
  return chk(launch_disc_head_bwd(p, Wd, 3, nfeat, n, t0, w0, t1, w1, dlogits, dmb, ms, ST), "ian_k_disc_head_bwd");
}

int ian_k_disc_head_wgrad(const float* mb, int32_t ms, int32_t nfeat, int32_t n, const float* dlogits, float* dWd,
                          int32_t accumulate, void* stream) {
  if (!mb || !dlogits || !dWd || n <= 0) return bad("ian_k_disc_head_wgrad");
  return chk(launch_disc_head_wgrad(mb, ms, nfeat, n, dlogits, 3, dWd, accumulate, ST), "ian_k_disc_head_wgrad");
}

int ian_k_sample(const float* mu, const float* ls, const float* eps, float* z0, float* klterm, int32_t n, int32_t d,
                 int32_t stride, int32_t eps_stride, void* stream) {
  if (!mu || !ls || !eps || !z0 || !klterm || n <= 0) return bad("ian_k_sample");
  return chk(launch_sample(mu, ls, eps, z0, klterm, n, d, stride, eps_stride, ST), "ian_k_sample");
}

int ian_k_sample_bwd(const float* mu, const float* ls, const float* eps, const float* dz0, float* dmu, float* dls, int32_t n,
                     int32_t d, int32_t stride, int32_t eps_stride, float klw, void* stream) {
  if (!mu || !ls || !eps || !dz0 || !dmu || !dls || n <= 0) return bad("ian_k_sample_bwd");
  return chk(launch_sample_bwd(mu, ls, eps, dz0, dmu, dls, n, d, stride, eps_stride, klw, ST), "ian_k_sample_bwd");
}

int ian_k_made_iaf(const float* z0, float* z, const float* wts, const float* bias, int32_t n, int32_t d, int32_t zs,
                   void* stream) {
  if (!z0 || !z || !wts || !bias || n <= 0) return bad("ian_k_made_iaf");
  return chk(launch_made_iaf(z0, z, wts, bias, n, d, zs, ST), "ian_k_made_iaf");
}

int ian_k_made_iaf_bwd(const float* z0, const float* dz, float* dz0, const float* wts, const float* bias, int32_t n,
                       int32_t d, int32_t zs, void* stream) {
  if (!z0 || !dz || !dz0 || !wts || !bias || n <= 0) return bad("ian_k_made_iaf_bwd");
  return chk(launch_made_iaf_bwd(z0, dz, dz0, wts, bias, n, d, zs, ST), "ian_k_made_iaf_bwd");
}

int ian_k_beta(const float* R, const float* G, const float* B, float* y_nchw, int32_t n, int32_t hw, int32_t rs, void* stream) {
  if (!R || !G || !B || !y_nchw || n <= 0) return bad("ian_k_beta");
  return chk(launch_beta(R, G, B, y_nchw, n, hw, rs, ST), "ian_k_beta");
}

int ian_k_beta_bwd(const float* gout_nchw, const float* R, const float* G, const float* B, float* gR, float* gG, float* gB,
                   int32_t n, int32_t hw, int32_t rs, int32_t act, void* stream) {
  if (!gout_nchw || !R || !G || !B || !gR || !gG || !gB || n <= 0) return bad("ian_k_beta_bwd");
  BetaBwdArgs a;
  const float* v[3] = {R, G, B};
  float* g[3] = {gR, gG, gB};
  for (int c = 0; c < 3; ++c) {
    a.v[c] = v[c]; a.g[c] = g[c]; a.scale[c] = nullptr; a.act[c] = act; a.accumulate[c] = 0;
  }
  return chk(launch_beta_bwd(gout_nchw, a, n, hw, rs, ST), "ian_k_beta_bwd");
}

int ian_k_concat2(const float* a, int32_t ca, int32_t sa, const float* b, int32_t cb, int32_t sb, float* y, int32_t sy,
                  int64_t npix, void* stream) {
  if (!a || !b || !y || npix <= 0) return bad("ian_k_concat2");
  return chk(launch_concat2(a, ca, sa, b, cb, sb, y, sy, npix, ST), "ian_k_concat2");
}

int ian_k_grad_pass(const float* gs, int32_t ss, int32_t coff, float* gd, const float* y, int32_t ds, int64_t npix,
                    int32_t C, int32_t act, int32_t accumulate, void* stream) {
  if (!gs || !gd || npix <= 0 || C <= 0) return bad("ian_k_grad_pass");
  return chk(launch_grad_pass(gs, ss, coff, gd, y, ds, nullptr, npix, C, act, accumulate, ST), "ian_k_grad_pass");
}

int ian_k_pair_loss(const float* a, const float* b, float* da, int64_t rows, int32_t C, int32_t stride, int32_t mode, float w,
                    int32_t accumulate, float* workspace, int32_t nblocks, float scale, float* out, void* stream) {
  if (!a || !b || !workspace || !out || rows <= 0 || C <= 0 || nblocks <= 0) return bad("ian_k_pair_loss");
  return chk(launch_pair_loss(a, b, da, rows, C, stride, mode, w, accumulate, workspace, nblocks, scale, out, ST), "ian_k_pair_loss");
}

int ian_k_sum_rows(const float* x, int32_t n, int32_t width, float scale, float* out, void* stream) {
  if (!x || !out || n <= 0 || width <= 0 || width > 64) return bad("ian_k_sum_rows");
  return chk(launch_sum_rows(x, n, width, scale, out, ST), "ian_k_sum_rows");
}

int ian_k_ortho(const float* W, float* dW, int32_t A, int32_t B, int32_t K, float c, float* vals, void* stream) {
  if (!W || !vals || A <= 0 || B <= 0) return bad("ian_k_ortho");
  return chk(launch_ortho(W, dW, A, B, K, c, vals, ST), "ian_k_ortho");
}

int ian_k_adam(float* p, const float* g, float* m, float* v, int64_t n, float a_t, float b1, float b2, float eps, void* stream) {
  if (!p || !g || !m || !v || n <= 0) return bad("ian_k_adam");
  return chk(launch_adam(p, g, m, v, n, a_t, b1, b2, eps, ST), "ian_k_adam");
}

}  // extern "C"

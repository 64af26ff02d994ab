// Elementwise / reduction kernels of the train_IAN.py step (train_IAN.py:116-276): batch-statistics batch-norm
// forward and backward (Lasagne batch_norm), activation backward, GlobalPool + MinibatchLayer
// (layers.py:486-524) + the 3-way softmax discriminator head (IAN.py:209-216), Gaussian sampling / KL, MADE+IAF
// backward-data, the pixel / feature losses, the orthogonal regulariser (train_IAN.py:158-165) and Adam
// (lasagne.updates.adam, App. B.7).  All HBM-bound: float4 accesses on NHWC rows, two-stage fixed-order
// reductions (no float atomics -> bitwise reproducible).
#include <algorithm>

#include "ian_internal.h"

namespace ian {

__device__ __forceinline__ float t_act(float v, int act) {
  switch (act) {
    case 1: return v > 0.f ? v : 0.f;
    case 2: return v > 0.f ? v : 0.2f * v;
    case 3: return v > 0.f ? v : expm1f(v);
    case 4: return tanhf(v);
    case 5: return 1.f / (1.f + __expf(-v));
    default: return v;
  }
}
__device__ __forceinline__ float t_dact(float y, int act) {  // derivative through the OUTPUT y
  switch (act) {
    case 1: return y > 0.f ? 1.f : 0.f;
    case 2: return y > 0.f ? 1.f : 0.2f;
    case 3: return y > 0.f ? 1.f : y + 1.f;
    case 4: return 1.f - y * y;
    case 5: return y * (1.f - y);
    default: return 1.f;
  }
}

static inline int grid_for(long long total, int cap = 8192) {
  long long b = (total + 255) / 256;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (int)b;
}

// ------------------------------------------------------------------------------------------------
// per-channel column statistics over NHWC rows.  partial[chunk][2][C] DOUBLES; then tree_sum -> sums[2][C] doubles
//   mode 0: s1 = sum x,  s2 = sum x^2                          (batch-norm forward statistics)
//   mode 1: g = dA*act'(a): s1 = sum g, s2 = sum g*xhat, xhat = (y-mean)*inv_std   (batch-norm backward: dbeta, dgamma)
//   mode 2: g = dA*act'(a): s1 = sum g                        (bias gradient)
// grid = (row chunks, column groups); block shape below.
//
// NUMERICS (round 4).  Forward statistics (mode 0): every float32 element is widened to float64 BEFORE it is squared / added, and
// all partial sums, the tree and the mean / variance arithmetic stay in float64; only mean, inv_std, scale, shift are rounded
// to float32, once.  Gradient sums (modes 1, 2): the per-element values g and g*xhat are float32 (they are float32 data), four
// rows are added in float32, everything longer -- chunk, tree, rank combine -- accumulates in float64; dbeta / dgamma are rounded
// to float32 once.  x*x is exact in float64, the sums carry a relative error of ~1e-16 * n, so
// var = E[x^2] - E[x]^2 is accurate to ~1e-13 * mean^2 -- BETTER conditioned than the float32 two-pass input.var(axes)
// of Lasagne's BatchNormLayer (oracle/refexec/minilasagne.py:607) that rounds 1e-7 per operation.  Rounds 1-3 formed the same
// one-pass expression in float32, whose cancellation error scales with mean^2 / var.  The kernels are HBM-bound; the
// float64 adds (full rate on the gfx950 VALU) are hidden behind the loads.
//
// Summation order is a function of the ROW INDEX only, not of how many rows there are: the caller cuts the rows
// into equal chunks whose size depends on the per-image extent alone (one image or a fixed fraction
// of one), a chunk is summed by the block's row lanes in a fixed order, and the chunk partials are combined by tree_sum,
// a pairwise (binary-counter) tree over the chunk index.  For power-of-two chunk counts
//     tree(chunks of the whole minibatch) == tree(rank 0's chunks) + tree(rank 1's chunks)   bit for bit,
// so the data-parallel step (per-rank tree, all-gather, tree over ranks in rank order) computes the SAME
// batch statistics as the single-process step on the whole minibatch (SyncBN).  For other counts the two trees
// associate differently: the float64 sums then differ by ~1e-16 relative, i.e. the float32 statistics derived from them are
// still identical except when a float64 difference straddles a float32 rounding boundary (ian_trainer warns at set-up).
// ------------------------------------------------------------------------------------------------
// Block = CL float4 column lanes x (256 / CL) row lanes.  CL = 8 (32 channels = one 128-byte line per row, 32 row lanes) for
// chunks of >= 64 rows: a 512-row chunk of a 128-channel map is then 4 workgroups instead of one half-idle one, and every
// thread keeps 4 rows (x up to 3 tensors) of loads in flight; CL = 64 (the whole row across the block) for short chunks.
// The order inside a chunk -- 4 interleaved accumulators per thread, (0+1)+(2+3), then a halving tree over the row lanes --
// is fixed by the chunk's row count alone, so the chunk-count independence stated above is unchanged.
struct d4 { double x, y, z, w; };
__device__ __forceinline__ void d4_add(d4& a, const d4& b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
template <int MODE, int CL>
__global__ __launch_bounds__(256) void colstats_kernel(ColStatsArgs a) {
  constexpr int RL = 256 / CL;
  __shared__ d4 red[2][256];
  const int q = threadIdx.x % CL, rl = threadIdx.x / CL;
  const int c = blockIdx.y * (CL * 4) + q * 4;
  const long long rows_per = (a.rows + gridDim.x - 1) / gridDim.x;
  const long long r0 = (long long)blockIdx.x * rows_per, r1 = min(a.rows, r0 + rows_per);
  const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
  const d4 dzero = {0.0, 0.0, 0.0, 0.0};
  d4 s1[4] = {dzero, dzero, dzero, dzero}, s2[4] = {dzero, dzero, dzero, dzero};
  if (c < a.C && r0 < r1) {
    float4 fmean = zero, fistd = zero;
    if (MODE == 1) {
      fmean = *reinterpret_cast<const float4*>(a.mean + c);
      fistd = *reinterpret_cast<const float4*>(a.inv_std + c);
    }
    for (long long r = r0 + rl; r < r1; r += 4 * RL) {
      float4 x[4], av[4], y[4];
      bool ok[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {   // rows past the chunk are clamped (loads stay branch-free) and masked below
        const long long rr = r + (long long)j * RL;
        ok[j] = rr < r1;
        const size_t off = (size_t)(ok[j] ? rr : r1 - 1) * a.stride + c;
        x[j] = *reinterpret_cast<const float4*>(a.x + off);
        if (MODE != 0 && a.act) av[j] = *reinterpret_cast<const float4*>(a.a + off);
        if (MODE == 1) y[j] = *reinterpret_cast<const float4*>(a.y + off);
      }
      if (MODE == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float4 v = ok[j] ? x[j] : zero;
          const d4 w = {v.x, v.y, v.z, v.w};
          d4_add(s1[j], w);
          s2[j].x += w.x * w.x; s2[j].y += w.y * w.y; s2[j].z += w.z * w.z; s2[j].w += w.w * w.w;   // exact products
        }
      } else {
        // gradient sums: g = dA * act'(a) and g * xhat are formed in float32 (the values bn_bwd_apply forms as well) and the four
        // rows of this pass are added in float32, (0+1)+(2+3); the LONG accumulation -- over the chunk, the tree -- is float64.
        // (All-float64 products made this pass compute-bound: 112 -> 155 us per call at 128 images.)
        float4 g[4], t[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float4 v = x[j];
          if (a.act) {
            v.x *= t_dact(av[j].x, a.act); v.y *= t_dact(av[j].y, a.act); v.z *= t_dact(av[j].z, a.act); v.w *= t_dact(av[j].w, a.act);
          }
          if (!ok[j]) v = zero;
          g[j] = v;
          if (MODE == 1) {
            t[j].x = v.x * ((y[j].x - fmean.x) * fistd.x); t[j].y = v.y * ((y[j].y - fmean.y) * fistd.y);
            t[j].z = v.z * ((y[j].z - fmean.z) * fistd.z); t[j].w = v.w * ((y[j].w - fmean.w) * fistd.w);
          }
        }
        const d4 gs = {(double)((g[0].x + g[1].x) + (g[2].x + g[3].x)), (double)((g[0].y + g[1].y) + (g[2].y + g[3].y)),
                       (double)((g[0].z + g[1].z) + (g[2].z + g[3].z)), (double)((g[0].w + g[1].w) + (g[2].w + g[3].w))};
        d4_add(s1[0], gs);
        if (MODE == 1) {
          const d4 ts = {(double)((t[0].x + t[1].x) + (t[2].x + t[3].x)), (double)((t[0].y + t[1].y) + (t[2].y + t[3].y)),
                         (double)((t[0].z + t[1].z) + (t[2].z + t[3].z)), (double)((t[0].w + t[1].w) + (t[2].w + t[3].w))};
          d4_add(s2[0], ts);
        }
      }
    }
  }
  d4 t1, t2;
  t1.x = (s1[0].x + s1[1].x) + (s1[2].x + s1[3].x); t1.y = (s1[0].y + s1[1].y) + (s1[2].y + s1[3].y);
  t1.z = (s1[0].z + s1[1].z) + (s1[2].z + s1[3].z); t1.w = (s1[0].w + s1[1].w) + (s1[2].w + s1[3].w);
  t2.x = (s2[0].x + s2[1].x) + (s2[2].x + s2[3].x); t2.y = (s2[0].y + s2[1].y) + (s2[2].y + s2[3].y);
  t2.z = (s2[0].z + s2[1].z) + (s2[2].z + s2[3].z); t2.w = (s2[0].w + s2[1].w) + (s2[2].w + s2[3].w);
  red[0][threadIdx.x] = t1;
  red[1][threadIdx.x] = t2;
  __syncthreads();
#pragma unroll
  for (int w = RL / 2; w >= 1; w >>= 1) {   // halving tree over the row lanes: lane l takes l + w
    if (rl < w) {
      d4 m = red[0][threadIdx.x], n = red[1][threadIdx.x];
      d4_add(m, red[0][threadIdx.x + w * CL]);
      d4_add(n, red[1][threadIdx.x + w * CL]);
      red[0][threadIdx.x] = m;
      red[1][threadIdx.x] = n;
    }
    __syncthreads();
  }
  if (rl == 0 && c < a.C) {
    double* p = a.partial + (size_t)blockIdx.x * 2 * a.C;
    const d4 u = red[0][threadIdx.x], v = red[1][threadIdx.x];
    p[c] = u.x; p[c + 1] = u.y; p[c + 2] = u.z; p[c + 3] = u.w;
    p[a.C + c] = v.x; p[a.C + c + 1] = v.y; p[a.C + c + 2] = v.z; p[a.C + c + 3] = v.w;
  }
}

// Pairwise tree over `count` values p[0], p[stride], ...: T(lo,hi) = T(lo,lo+m) + T(lo+m,hi) with m the largest power
// of two below hi-lo -- evaluated left to right with a binary counter of partial sums (no recursion, registers only).
constexpr int TS_LEVELS = 16;
__device__ __forceinline__ double tree_sum_seq(const double* __restrict__ p, size_t stride, int count) {
  double acc[TS_LEVELS];
#pragma unroll
  for (int l = 0; l < TS_LEVELS; ++l) acc[l] = 0.0;
  for (int k = 0; k < count; ++k) {
    double v = p[(size_t)k * stride];
    bool placed = false;
#pragma unroll
    for (int l = 0; l < TS_LEVELS; ++l) {
      if (!placed) {
        if ((k >> l) & 1) v = acc[l] + v;   // level l holds the left sibling: combine and carry
        else { acc[l] = v; placed = true; }
      }
    }
  }
  double r = 0.0;
  bool have = false;
#pragma unroll
  for (int l = 0; l < TS_LEVELS; ++l)
    if ((count >> l) & 1) {
      r = have ? acc[l] + r : acc[l];
      have = true;
    }
  return r;
}
// out[i] = tree over k of partial[k*width + i].  Block = TC columns x (256 / TC) lanes; for a power-of-two count each lane
// takes a contiguous 1/L-th (its own subtree) and the lanes meet pairwise in LDS, which is the same tree -- whatever L is, so
// round 5 could go from 16 x 16 to 8 columns (one 64-byte line of doubles) x 32 lanes without moving a bit: the GEMM-epilogue
// statistics (TgStats) hand over one partial per 128-row tile, up to 4096 of them where colstats' per-image chunks were 1024.
// tree_column: every thread of the block calls it; the column's total is in red[col] afterwards.
constexpr int TC = 8, TL = 256 / TC;
__device__ __forceinline__ void tree_column(const double* __restrict__ partial, int count, int width, int i, bool valid,
                                            double* red) {
  const int lane = threadIdx.x / TC;
  const bool pow2 = (count & (count - 1)) == 0;
  const int L = pow2 ? min(TL, count) : 1;   // lanes in use
  double s = 0.0;
  if (valid && lane < L) {
    const int per = count / L;
    s = tree_sum_seq(partial + (size_t)lane * per * width + i, (size_t)width, per);
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int w = 1; w < L; w <<= 1) {
    if (lane % (2 * w) == 0 && lane + w < L) red[threadIdx.x] = red[threadIdx.x] + red[threadIdx.x + TC * w];
    __syncthreads();
  }
}
__global__ __launch_bounds__(256) void tree_sum_kernel(const double* __restrict__ partial, int count, int width,
                                                       double* __restrict__ out) {
  __shared__ double red[256];
  const int col = threadIdx.x % TC, lane = threadIdx.x / TC;
  const int i = blockIdx.x * TC + col;
  tree_column(partial, count, width, i, i < width, red);
  if (lane == 0 && i < width) out[i] = red[col];
}
hipError_t launch_tree_sum(const double* partial, int count, int width, double* out, hipStream_t s) {
  if (count <= 0 || width <= 0 || count >= (1 << TS_LEVELS)) return hipErrorInvalidValue;
  hipLaunchKernelGGL(tree_sum_kernel, dim3((width + TC - 1) / TC), dim3(256), 0, s, partial, count, width, out);
  return hipGetLastError();
}

// batch statistics -> folded affine (App. B.3): mean, biased variance, inv_std = 1/sqrt(var+eps).  One statement of the
// arithmetic, contraction off, for bn_make_affine_kernel and bn_finish_kernel: the fused single-process statistics and the
// two-stage data-parallel ones (all-gather between the tree and this) must agree bit for bit.  Mean and variance are formed in
// float64 from the float64 sums and rounded to float32 once; scale / shift are float32 products of the rounded values (what
// the affine apply kernel and the backward read).
__device__ __forceinline__ void bn_affine_math(double s1, double s2, double count, float eps, float gamma, float beta,
                                               float& m, float& is, float& sc, float& sh) {
#pragma clang fp contract(off)
  const double md = s1 / count;
  double var = s2 / count - md * md;
  var = var > 0.0 ? var : 0.0;
  m = (float)md;
  is = (float)(1.0 / sqrt(var + (double)eps));
  sc = gamma * is;
  sh = beta - m * sc;
}
// r = keep * r + alpha * batch, rounded as the two ian_k_axpy launches it replaces round it (product, then fused add)
__device__ __forceinline__ float running_math(float r, float batch, float keep, float alpha) {
#pragma clang fp contract(off)
  const float kept = keep * r;
  return __builtin_fmaf(alpha, batch, kept);
}

// second stage of the batch-norm FORWARD statistics in one launch (single-process step): the tree of tree_sum_kernel over
// columns c and C + c of partial[chunk][2][C], bn_make_affine's arithmetic, and the running averages of the pass that owns
// them.  Block = TC/2 channels x {s1, s2} x TL lanes.
struct BnFinishArgs {
  const double* partial;
  const float* gamma;
  const float* beta;
  double* sums;
  float* mean;
  float* inv_std;
  float* scale;
  float* shift;
  float* run_mean;     // or nullptr
  float* run_inv_std;
  float n, eps, keep, alpha;
  int count, C;
};
__global__ __launch_bounds__(256) void bn_finish_kernel(BnFinishArgs a) {
  __shared__ double red[256];
  constexpr int CH = TC / 2;
  const int col = threadIdx.x % TC, lane = threadIdx.x / TC;
  const int ch = blockIdx.x * CH + (col % CH);
  tree_column(a.partial, a.count, 2 * a.C, (col / CH) * a.C + ch, ch < a.C, red);
  if (lane == 0 && col < CH && ch < a.C) {
    const double s1 = red[col], s2 = red[col + CH];
    a.sums[ch] = s1;
    a.sums[a.C + ch] = s2;
    float m, is, sc, sh;
    bn_affine_math(s1, s2, (double)a.n, a.eps, a.gamma[ch], a.beta[ch], m, is, sc, sh);
    a.mean[ch] = m;
    a.inv_std[ch] = is;
    a.scale[ch] = sc;
    a.shift[ch] = sh;
    if (a.run_mean) {
      a.run_mean[ch] = running_math(a.run_mean[ch], m, a.keep, a.alpha);
      a.run_inv_std[ch] = running_math(a.run_inv_std[ch], is, a.keep, a.alpha);
    }
  }
}
// second stage of the batch-norm BACKWARD statistics: the tree, then dbeta (+)= s1 and dgamma (+)= s2 (the two gradient
// accumulations that used to be ian_k_axpy launches); the float64 sums are rounded to float32 once, here
__global__ __launch_bounds__(256) void bn_bwd_finish_kernel(const double* __restrict__ partial, int count, int C,
                                                            double* __restrict__ sums, float* __restrict__ gbeta, int acc_beta,
                                                            float* __restrict__ ggamma, int acc_gamma) {
  __shared__ double red[256];
  const int col = threadIdx.x % TC, lane = threadIdx.x / TC;
  const int i = blockIdx.x * TC + col;
  tree_column(partial, count, 2 * C, i, i < 2 * C, red);
  if (lane == 0 && i < 2 * C) {
    const double v = red[col];
    sums[i] = v;
    if (gbeta) {
      if (i < C) gbeta[i] = (acc_beta ? gbeta[i] : 0.f) + (float)v;
      else ggamma[i - C] = (acc_gamma ? ggamma[i - C] : 0.f) + (float)v;
    }
  }
}
hipError_t launch_bn_finish(const double* partial, int nchunks, int C, double* sums, float count, float eps, const float* gamma,
                            const float* beta, float* mean, float* inv_std, float* scale, float* shift, float* run_mean,
                            float* run_inv_std, float keep, float alpha, hipStream_t s) {
  if (nchunks <= 0 || nchunks >= (1 << TS_LEVELS)) return hipErrorInvalidValue;
  BnFinishArgs f;
  f.partial = partial; f.gamma = gamma; f.beta = beta; f.sums = sums; f.mean = mean; f.inv_std = inv_std; f.scale = scale;
  f.shift = shift; f.run_mean = run_mean; f.run_inv_std = run_inv_std; f.n = count; f.eps = eps; f.keep = keep; f.alpha = alpha;
  f.count = nchunks; f.C = C;
  hipLaunchKernelGGL(bn_finish_kernel, dim3((C + TC / 2 - 1) / (TC / 2)), dim3(256), 0, s, f);
  return hipGetLastError();
}
hipError_t launch_bn_bwd_finish(const double* partial, int nchunks, int C, double* sums, float* gbeta, int acc_beta, float* ggamma,
                                int acc_gamma, hipStream_t s) {
  if (nchunks <= 0 || nchunks >= (1 << TS_LEVELS)) return hipErrorInvalidValue;
  hipLaunchKernelGGL(bn_bwd_finish_kernel, dim3((2 * C + TC - 1) / TC), dim3(256), 0, s, partial, nchunks, C, sums, gbeta, acc_beta,
                     ggamma, acc_gamma);
  return hipGetLastError();
}
static void launch_colstats_stage1(const ColStatsArgs& a, int nchunks, hipStream_t s);
hipError_t launch_bn_stats_affine(const ColStatsArgs& a, int nchunks, double* sums, float count, float eps, const float* gamma,
                                  const float* beta, float* mean, float* inv_std, float* scale, float* shift, float* run_mean,
                                  float* run_inv_std, float keep, float alpha, hipStream_t s) {
  if ((a.C & 3) || a.mode != 0 || nchunks <= 0 || nchunks >= (1 << TS_LEVELS)) return hipErrorInvalidValue;
  launch_colstats_stage1(a, nchunks, s);
  return launch_bn_finish(a.partial, nchunks, a.C, sums, count, eps, gamma, beta, mean, inv_std, scale, shift, run_mean, run_inv_std, keep,
                          alpha, s);
}
hipError_t launch_bn_bwd_stats(const ColStatsArgs& a, int nchunks, double* sums, float* gbeta, int acc_beta, float* ggamma,
                               int acc_gamma, hipStream_t s) {
  if ((a.C & 3) || a.mode != 1 || nchunks <= 0 || nchunks >= (1 << TS_LEVELS)) return hipErrorInvalidValue;
  launch_colstats_stage1(a, nchunks, s);
  return launch_bn_bwd_finish(a.partial, nchunks, a.C, sums, gbeta, acc_beta, ggamma, acc_gamma, s);
}
static void launch_colstats_stage1(const ColStatsArgs& a, int nchunks, hipStream_t s) {
  const long long rows_per = (a.rows + nchunks - 1) / nchunks;
  const bool narrow = rows_per >= 64;
  const dim3 grid(nchunks, narrow ? (a.C + 31) / 32 : (a.C + 255) / 256);
#define IAN_COLSTATS(M)                                                                           \
  do {                                                                                            \
    if (narrow) hipLaunchKernelGGL((colstats_kernel<M, 8>), grid, dim3(256), 0, s, a);            \
    else hipLaunchKernelGGL((colstats_kernel<M, 64>), grid, dim3(256), 0, s, a);                  \
  } while (0)
  if (a.mode == 0) IAN_COLSTATS(0);
  else if (a.mode == 1) IAN_COLSTATS(1);
  else IAN_COLSTATS(2);
#undef IAN_COLSTATS
}
hipError_t launch_colstats(const ColStatsArgs& a, int nchunks, double* sums, hipStream_t s) {
  if (a.C & 3) return hipErrorInvalidValue;
  launch_colstats_stage1(a, nchunks, s);
  return launch_tree_sum(a.partial, nchunks, 2 * a.C, sums, s);
}

// batch statistics -> folded affine (App. B.3): mean, biased variance, inv_std = 1/sqrt(var+eps)
__global__ __launch_bounds__(256) void bn_make_affine_kernel(const double* __restrict__ sums, float count, float eps,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, int C,
                                                             float* __restrict__ mean, float* __restrict__ inv_std,
                                                             float* __restrict__ scale, float* __restrict__ shift) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  float m, is, sc, sh;
  bn_affine_math(sums[c], sums[C + c], (double)count, eps, gamma[c], beta[c], m, is, sc, sh);
  mean[c] = m;
  inv_std[c] = is;
  scale[c] = sc;
  shift[c] = sh;
}
hipError_t launch_bn_make_affine(const double* sums, float count, float eps, const float* gamma, const float* beta,
                                 int C, float* mean, float* inv_std, float* scale, float* shift, hipStream_t s) {
  hipLaunchKernelGGL(bn_make_affine_kernel, dim3((C + 255) / 256), dim3(256), 0, s, sums, count, eps, gamma, beta, C, mean,
                     inv_std, scale, shift);
  return hipGetLastError();
}

// running averages of one normalisation, in place: r <- keep * r + alpha * batch for mean and inv_std in ONE launch, with the
// same running_math the fused single-process second stage (bn_finish_kernel) uses -- the data-parallel `exact` step (whose
// statistics pass through an all-gather between the tree and bn_make_affine) updates them bit for bit like the 1-GPU step.
// (Round 4 did this with four axpy launches whose x and y aliased under __restrict__.)
__global__ __launch_bounds__(256) void bn_running_kernel(float* run_mean, const float* __restrict__ mean, float* run_inv_std,
                                                         const float* __restrict__ inv_std, int C, float keep, float alpha) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  run_mean[c] = running_math(run_mean[c], mean[c], keep, alpha);
  run_inv_std[c] = running_math(run_inv_std[c], inv_std[c], keep, alpha);
}
hipError_t launch_bn_running(float* run_mean, const float* mean, float* run_inv_std, const float* inv_std, int C, float keep,
                             float alpha, hipStream_t s) {
  hipLaunchKernelGGL(bn_running_kernel, dim3((C + 255) / 256), dim3(256), 0, s, run_mean, mean, run_inv_std, inv_std, C, keep, alpha);
  return hipGetLastError();
}

// backward through [batch-norm ->] activation:   g = dA * act'(a)
//   bn:   dy = scale * (g - s1/N - xhat * s2/N),  xhat = (y - mean) * inv_std,  (s1, s2) = sums from colstats mode 1
//   else: dy = g
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(BnBwdArgs a) {
  const int c4n = a.C >> 2;
  const long long total = a.rows * c4n;
  const double invn = 1.0 / (double)a.count;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long r = i / c4n;
    const int c = (int)(i % c4n) * 4;
    const size_t off = (size_t)r * a.stride + c;
    float4 g = *reinterpret_cast<const float4*>(a.dA + off);
    if (a.act) {
      const float4 av = *reinterpret_cast<const float4*>(a.a + off);
      g.x *= t_dact(av.x, a.act); g.y *= t_dact(av.y, a.act); g.z *= t_dact(av.z, a.act); g.w *= t_dact(av.w, a.act);
    }
    if (a.sums) {
      const float4 y = *reinterpret_cast<const float4*>(a.y + off);
      const float4 mean = *reinterpret_cast<const float4*>(a.mean + c), istd = *reinterpret_cast<const float4*>(a.inv_std + c);
      const float4 sc = *reinterpret_cast<const float4*>(a.scale + c);
      // mean(g), mean(g*xhat): float64 sums / N rounded to float32 once (8 doubles per thread per row group: L1-resident)
      const double* s1 = a.sums + c;
      const double* s2 = a.sums + a.C + c;
      const float4 m1 = make_float4((float)(s1[0] * invn), (float)(s1[1] * invn), (float)(s1[2] * invn), (float)(s1[3] * invn));
      const float4 m2 = make_float4((float)(s2[0] * invn), (float)(s2[1] * invn), (float)(s2[2] * invn), (float)(s2[3] * invn));
      g.x = sc.x * (g.x - m1.x - (y.x - mean.x) * istd.x * m2.x);
      g.y = sc.y * (g.y - m1.y - (y.y - mean.y) * istd.y * m2.y);
      g.z = sc.z * (g.z - m1.z - (y.z - mean.z) * istd.z * m2.z);
      g.w = sc.w * (g.w - m1.w - (y.w - mean.w) * istd.w * m2.w);
    }
    *reinterpret_cast<float4*>(a.dy + off) = g;
  }
}
hipError_t launch_bn_bwd_apply(const BnBwdArgs& a, hipStream_t s) {
  if (a.C & 3) return hipErrorInvalidValue;
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(grid_for(a.rows * (a.C >> 2))), dim3(256), 0, s, a);
  return hipGetLastError();
}

// y (+)= alpha * x on flat buffers (gradient accumulation, L2 regulariser gradient 2*reg*p)
// (no __restrict__: callers may pass x == y, e.g. an in-place scale)
__global__ __launch_bounds__(256) void axpy_kernel(float alpha, const float* x, float* y, long long n, int accumulate) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
    y[i] = (accumulate ? y[i] : 0.f) + alpha * x[i];
}
hipError_t launch_axpy(float alpha, const float* x, float* y, long long n, int accumulate, hipStream_t s) {
  hipLaunchKernelGGL(axpy_kernel, dim3(grid_for(n)), dim3(256), 0, s, alpha, x, y, n, accumulate);
  return hipGetLastError();
}
// y (+)= (float)(alpha * x), x float64: the float64 column sums of colstats -> a float32 gradient (rounded once)
__global__ __launch_bounds__(256) void axpy_f64_kernel(double alpha, const double* __restrict__ x, float* __restrict__ y,
                                                       long long n, int accumulate) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
    y[i] = (accumulate ? y[i] : 0.f) + (float)(alpha * x[i]);
}
hipError_t launch_axpy_f64(double alpha, const double* x, float* y, long long n, int accumulate, hipStream_t s) {
  hipLaunchKernelGGL(axpy_f64_kernel, dim3(grid_for(n)), dim3(256), 0, s, alpha, x, y, n, accumulate);
  return hipGetLastError();
}

// NCHW [n,c,hw] -> NHWC with pixel stride `stride` (channels >= c left untouched = zero padding)
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                           int n, int hw, int c, int stride) {
  const long long total = (long long)n * hw * c;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int ch = (int)(i % c);
    const long long pix = i / c;
    const int p = (int)(pix % hw), b = (int)(pix / hw);
    dst[pix * stride + ch] = src[((size_t)b * c + ch) * hw + p];
  }
}
hipError_t launch_nchw_to_nhwc(const float* src, float* dst, int n, int hw, int c, int stride, hipStream_t s) {
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(grid_for((long long)n * hw * c)), dim3(256), 0, s, src, dst, n, hw, c, stride);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// GlobalPoolLayer (mean over H*W) forward / backward on NHWC maps
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void globalpool_kernel(const float* __restrict__ x, float* __restrict__ y, int n,
                                                         int hw, int C, int xs, int ys) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)n * C) return;
  const int c = (int)(i % C), b = (int)(i / C);
  float s = 0.f;
  for (int p = 0; p < hw; ++p) s += x[((size_t)b * hw + p) * xs + c];
  y[(size_t)b * ys + c] = s / hw;
}
__global__ __launch_bounds__(256) void globalpool_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx,
                                                             int n, int hw, int C, int xs, int ys, int accumulate) {
  const long long total = (long long)n * hw * C;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % C);
    const long long pix = i / C;
    const int b = (int)(pix / hw);
    const float g = dy[(size_t)b * ys + c] / hw;
    float* o = dx + pix * xs + c;
    *o = accumulate ? *o + g : g;
  }
}
hipError_t launch_globalpool(const float* x, float* y, int n, int hw, int C, int xs, int ys, hipStream_t s) {
  hipLaunchKernelGGL(globalpool_kernel, dim3(grid_for((long long)n * C, 1 << 20)), dim3(256), 0, s, x, y, n, hw, C, xs, ys);
  return hipGetLastError();
}
hipError_t launch_globalpool_bwd(const float* dy, float* dx, int n, int hw, int C, int xs, int ys, int accumulate,
                                 hipStream_t s) {
  hipLaunchKernelGGL(globalpool_bwd_kernel, dim3(grid_for((long long)n * hw * C)), dim3(256), 0, s, dy, dx, n, hw, C, xs, ys,
                     accumulate);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// MinibatchLayer (layers.py:486-524)
// ------------------------------------------------------------------------------------------------
// W = theta * (exp(log_weight_scale) / sqrt(sum_i theta^2))  (layers.py:494).  Block = 32 columns (k,d) x 8 row lanes.
__global__ __launch_bounds__(256) void mb_weight_kernel(const float* __restrict__ theta, const float* __restrict__ lws,
                                                        float* __restrict__ W, float* __restrict__ colscale, int nin,
                                                        int ncol) {
  __shared__ float red[256];
  const int c = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int j = blockIdx.x * 32 + c;
  float ss = 0.f;
  if (j < ncol)
    for (int i = rl; i < nin; i += 8) {
      const float t = theta[(size_t)i * ncol + j];
      ss += t * t;
    }
  red[threadIdx.x] = ss;
  __syncthreads();
  if (j >= ncol) return;
  float tot = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) tot += red[c + 32 * k];
  const float sc = expf(lws[j]) / sqrtf(tot);
  if (rl == 0) colscale[j] = sc;
  for (int i = rl; i < nin; i += 8) W[(size_t)i * ncol + j] = theta[(size_t)i * ncol + j] * sc;
}
// dtheta[i,j] = dW[i,j]*s_j - theta[i,j]*s_j*(sum_i' dW[i',j]*theta[i',j]) / sum_i' theta[i',j]^2 ; dlws[j] = sum_i dW[i,j]*W[i,j]
__global__ __launch_bounds__(256) void mb_weight_bwd_kernel(const float* __restrict__ theta,
                                                            const float* __restrict__ colscale,
                                                            const float* __restrict__ dW, float* __restrict__ dtheta,
                                                            float* __restrict__ dlws, int nin, int ncol, int accumulate) {
  __shared__ float red[2][256];
  const int c = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int j = blockIdx.x * 32 + c;
  float ss = 0.f, dot = 0.f;
  if (j < ncol)
    for (int i = rl; i < nin; i += 8) {
      const float t = theta[(size_t)i * ncol + j];
      ss += t * t;
      dot += dW[(size_t)i * ncol + j] * t;
    }
  red[0][threadIdx.x] = ss;
  red[1][threadIdx.x] = dot;
  __syncthreads();
  if (j >= ncol) return;
  ss = 0.f;
  dot = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    ss += red[0][c + 32 * k];
    dot += red[1][c + 32 * k];
  }
  const float sc = colscale[j];
  for (int i = rl; i < nin; i += 8) {
    const size_t o = (size_t)i * ncol + j;
    const float g = dW[o] * sc - theta[o] * sc * dot / ss;
    dtheta[o] = accumulate ? dtheta[o] + g : g;
  }
  if (rl == 0) {
    const float gl = dot * sc;
    dlws[j] = accumulate ? dlws[j] + gl : gl;
  }
}
hipError_t launch_mb_weight(const float* theta, const float* lws, float* W, float* colscale, int nin, int ncol, hipStream_t s) {
  hipLaunchKernelGGL(mb_weight_kernel, dim3((ncol + 31) / 32), dim3(256), 0, s, theta, lws, W, colscale, nin, ncol);
  return hipGetLastError();
}
hipError_t launch_mb_weight_bwd(const float* theta, const float* colscale, const float* dW, float* dtheta, float* dlws,
                                int nin, int ncol, int accumulate, hipStream_t s) {
  hipLaunchKernelGGL(mb_weight_bwd_kernel, dim3((ncol + 31) / 32), dim3(256), 0, s, theta, colscale, dW, dtheta, dlws, nin,
                     ncol, accumulate);
  return hipGetLastError();
}

// f[b,k] = sum_b' exp(-(sum_d |act[b,k,d]-act[b',k,d]| + 1e6*[b==b'])) + bias[k]   (layers.py:507-520)
// act_all holds the activations of the WHOLE minibatch (all ranks, nall rows of stride as); this rank owns rows
// [row0, row0+n).  out: mb[b, fin + k] (the first fin columns are the input features, copied here: layers.py:524).
// Block = one sample b x 32 kernels k x 8 lanes over the other samples b' (lane s takes b' = s mod 8; the eight partial sums
// meet in LDS as ((0+1)+(2+3))+((4+5)+(6+7))): 2048 workgroups and 16-iteration loops at 128 samples instead of 250 workgroups
// walking all 128 samples serially (99 us forward / 177 us backward for 41 M absolute differences).
__global__ __launch_bounds__(256) void mb_forward_kernel(const float* __restrict__ act_all, int nall, int as, int row0,
                                                         int n, int nk, int nd, const float* __restrict__ bias,
                                                         const float* __restrict__ feat, int fs, int fin,
                                                         float* __restrict__ mb, int ms) {
  __shared__ float red[8][32];
  const int b = blockIdx.x, kl = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int k = blockIdx.y * 32 + kl;
  if (blockIdx.y == 0)
    for (int c = threadIdx.x; c < fin; c += 256) mb[(size_t)b * ms + c] = feat[(size_t)b * fs + c];
  float f = 0.f;
  if (k < nk) {
    float me[8];
#pragma unroll
    for (int d = 0; d < 8; ++d) me[d] = d < nd ? act_all[(size_t)(row0 + b) * as + k * nd + d] : 0.f;
    for (int o = sl; o < nall; o += 8) {
      const float* ot = act_all + (size_t)o * as + k * nd;
      float a = (o == row0 + b) ? 1e6f : 0.f;
#pragma unroll
      for (int d = 0; d < 8; ++d)
        if (d < nd) a += fabsf(me[d] - ot[d]);
      f += expf(-a);
    }
  }
  red[sl][kl] = f;
  __syncthreads();
  if (sl == 0 && k < nk)
    mb[(size_t)b * ms + fin + k] =
        (((red[0][kl] + red[1][kl]) + (red[2][kl] + red[3][kl])) + ((red[4][kl] + red[5][kl]) + (red[6][kl] + red[7][kl]))) + bias[k];
}
// dact[b,k,d] = - sum_b' exp(-A[b,k,b']) * (df[b,k] + df[b',k]) * sign(act[b,k,d]-act[b',k,d]);  df_all: all ranks' df (stride dfs)
__global__ __launch_bounds__(256) void mb_backward_kernel(const float* __restrict__ act_all, int nall, int as, int row0,
                                                          int n, int nk, int nd, const float* __restrict__ df_all,
                                                          int dfs, float* __restrict__ dact, int das) {
  __shared__ float red[8][8][32];   // [d][lane][k]
  const int b = blockIdx.x, kl = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int k = blockIdx.y * 32 + kl;
  float g[8];
#pragma unroll
  for (int d = 0; d < 8; ++d) g[d] = 0.f;
  if (k < nk) {
    float me[8];
#pragma unroll
    for (int d = 0; d < 8; ++d) me[d] = d < nd ? act_all[(size_t)(row0 + b) * as + k * nd + d] : 0.f;
    const float dfb = df_all[(size_t)(row0 + b) * dfs + k];
    for (int o = sl; o < nall; o += 8) {
      if (o == row0 + b) continue;
      const float* ot = act_all + (size_t)o * as + k * nd;
      float df[8];
      float a = 0.f;
#pragma unroll
      for (int d = 0; d < 8; ++d) {
        df[d] = d < nd ? me[d] - ot[d] : 0.f;
        a += fabsf(df[d]);
      }
      const float w = expf(-a) * (dfb + df_all[(size_t)o * dfs + k]);
#pragma unroll
      for (int d = 0; d < 8; ++d) g[d] -= w * (df[d] > 0.f ? 1.f : (df[d] < 0.f ? -1.f : 0.f));
    }
  }
#pragma unroll
  for (int d = 0; d < 8; ++d) red[d][sl][kl] = g[d];
  __syncthreads();
  if (sl == 0 && k < nk)
    for (int d = 0; d < nd; ++d)
      dact[(size_t)b * das + k * nd + d] = ((red[d][0][kl] + red[d][1][kl]) + (red[d][2][kl] + red[d][3][kl])) +
                                           ((red[d][4][kl] + red[d][5][kl]) + (red[d][6][kl] + red[d][7][kl]));
}
hipError_t launch_mb_forward(const float* act_all, int nall, int as, int row0, int n, int nk, int nd, const float* bias,
                             const float* feat, int fs, int fin, float* mb, int ms, hipStream_t s) {
  if (nd > 8 || n <= 0 || nk <= 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL(mb_forward_kernel, dim3(n, (nk + 31) / 32), dim3(256), 0, s, act_all, nall, as, row0, n, nk, nd, bias, feat, fs,
                     fin, mb, ms);
  return hipGetLastError();
}
hipError_t launch_mb_backward(const float* act_all, int nall, int as, int row0, int n, int nk, int nd, const float* df_all,
                              int dfs, float* dact, int das, hipStream_t s) {
  if (nd > 8 || n <= 0 || nk <= 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL(mb_backward_kernel, dim3(n, (nk + 31) / 32), dim3(256), 0, s, act_all, nall, as, row0, n, nk, nd, df_all, dfs,
                     dact, das);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// discriminator head: logits = mb @ Wd (1524 x 3, no bias), softmax, categorical cross-entropy (IAN.py:210-216,
// train_IAN.py:228-250).  One block (256 threads) per sample.
//   out per sample: p[3]; loss terms -log p[target_t] for up to 2 targets; correct = (argmax p == acc_target)
//   dlogits = sum_t w_t * (p - onehot(target_t))          (w_t already holds loss weight / global batch)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void disc_head_kernel(const float* __restrict__ mb, int ms, int nfeat,
                                                        const float* __restrict__ Wd, int ncls, DiscHeadArgs a) {
  __shared__ float red[3][256];
  const int b = blockIdx.x;
  float l[3] = {0.f, 0.f, 0.f};
  for (int j = threadIdx.x; j < nfeat; j += 256) {
    const float v = mb[(size_t)b * ms + j];
    for (int k = 0; k < 3; ++k) l[k] += v * Wd[(size_t)j * ncls + k];
  }
  for (int k = 0; k < 3; ++k) red[k][threadIdx.x] = l[k];
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s)
      for (int k = 0; k < 3; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float z0 = red[0][0], z1 = red[1][0], z2 = red[2][0];
    const float m = fmaxf(z0, fmaxf(z1, z2));
    const float e0 = expf(z0 - m), e1 = expf(z1 - m), e2 = expf(z2 - m);
    const float inv = 1.f / (e0 + e1 + e2);
    const float p[3] = {e0 * inv, e1 * inv, e2 * inv};
    for (int k = 0; k < 3; ++k) a.p[(size_t)b * 3 + k] = p[k];
    int am = 0;
    if (p[1] > p[am]) am = 1;
    if (p[2] > p[am]) am = 2;
    for (int t = 0; t < 2; ++t) a.loss[(size_t)b * 4 + t] = (a.target[t] >= 0) ? -logf(p[a.target[t]]) : 0.f;
    a.loss[(size_t)b * 4 + 2] = (am == a.acc_target) ? 1.f : 0.f;
    a.loss[(size_t)b * 4 + 3] = 0.f;
  }
}
// dlogits[b,k] = sum_t w[t]*(p[b,k] - [k==target[t]]);  dmb[b,j] = sum_k dlogits[b,k]*Wd[j,k]
__global__ __launch_bounds__(256) void disc_head_bwd_kernel(const float* __restrict__ p, const float* __restrict__ Wd,
                                                            int ncls, int nfeat, int t0, float w0, int t1, float w1,
                                                            float* __restrict__ dlogits, float* __restrict__ dmb, int ms) {
  const int b = blockIdx.x;
  float dl[3];
  for (int k = 0; k < 3; ++k) {
    const float pk = p[(size_t)b * 3 + k];
    dl[k] = (t0 >= 0 ? w0 * (pk - (k == t0 ? 1.f : 0.f)) : 0.f) + (t1 >= 0 ? w1 * (pk - (k == t1 ? 1.f : 0.f)) : 0.f);
  }
  if (threadIdx.x < 3) dlogits[(size_t)b * 4 + threadIdx.x] = dl[threadIdx.x];
  for (int j = threadIdx.x; j < nfeat; j += 256) {
    float g = 0.f;
    for (int k = 0; k < 3; ++k) g += dl[k] * Wd[(size_t)j * ncls + k];
    dmb[(size_t)b * ms + j] = g;
  }
}
// dWd[j,k] (+)= sum_b mb[b,j]*dlogits[b,k]   (one thread per (j,k), fixed order over b)
__global__ __launch_bounds__(256) void disc_head_wgrad_kernel(const float* __restrict__ mb, int ms, int nfeat, int n,
                                                              const float* __restrict__ dlogits, int ncls,
                                                              float* __restrict__ dWd, int accumulate) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= nfeat * ncls) return;
  const int j = i / ncls, k = i % ncls;
  float s = 0.f;
  for (int b = 0; b < n; ++b) s += mb[(size_t)b * ms + j] * dlogits[(size_t)b * 4 + k];
  dWd[i] = accumulate ? dWd[i] + s : s;
}
hipError_t launch_disc_head(const float* mb, int ms, int nfeat, const float* Wd, int ncls, int n, const DiscHeadArgs& a,
                            hipStream_t s) {
  if (ncls != 3) return hipErrorInvalidValue;
  hipLaunchKernelGGL(disc_head_kernel, dim3(n), dim3(256), 0, s, mb, ms, nfeat, Wd, ncls, a);
  return hipGetLastError();
}
hipError_t launch_disc_head_bwd(const float* p, const float* Wd, int ncls, int nfeat, int n, int t0, float w0, int t1,
                                float w1, float* dlogits, float* dmb, int ms, hipStream_t s) {
  hipLaunchKernelGGL(disc_head_bwd_kernel, dim3(n), dim3(256), 0, s, p, Wd, ncls, nfeat, t0, w0, t1, w1, dlogits, dmb, ms);
  return hipGetLastError();
}
hipError_t launch_disc_head_wgrad(const float* mb, int ms, int nfeat, int n, const float* dlogits, int ncls, float* dWd,
                                  int accumulate, hipStream_t s) {
  hipLaunchKernelGGL(disc_head_wgrad_kernel, dim3((nfeat * ncls + 255) / 256), dim3(256), 0, s, mb, ms, nfeat, n, dlogits, ncls,
                     dWd, accumulate);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// latent: GaussianSampleLayer (layers.py:419-433) + KL (train_IAN.py:172), forward and backward
// ------------------------------------------------------------------------------------------------
// z0 = mu + exp(ls)*eps ; klterm[b,j] = 1 + 2 ls - mu^2 - exp(2 ls)   (summed later)
__global__ __launch_bounds__(256) void sample_kernel(const float* __restrict__ mu, const float* __restrict__ ls,
                                                     const float* __restrict__ eps, float* __restrict__ z0,
                                                     float* __restrict__ klterm, int n, int d, int stride, int es) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n * d) return;
  const int b = i / d, j = i % d;
  const size_t o = (size_t)b * stride + j;
  const float m = mu[o], l = ls[o];
  z0[o] = m + expf(l) * eps[(size_t)b * es + j];
  klterm[i] = 1.f + 2.f * l - m * m - expf(2.f * l);
}
// dmu = dz0 + klw*mu ; dls = dz0*exp(ls)*eps + klw*(exp(2 ls) - 1)      klw = kl_weight / (B_global * d)
__global__ __launch_bounds__(256) void sample_bwd_kernel(const float* __restrict__ mu, const float* __restrict__ ls,
                                                         const float* __restrict__ eps, const float* __restrict__ dz0,
                                                         float* __restrict__ dmu, float* __restrict__ dls, int n, int d,
                                                         int stride, int es, float klw) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n * d) return;
  const int b = i / d, j = i % d;
  const size_t o = (size_t)b * stride + j;
  const float m = mu[o], l = ls[o], g = dz0[o];
  dmu[o] = g + klw * m;
  dls[o] = g * expf(l) * eps[(size_t)b * es + j] + klw * (expf(2.f * l) - 1.f);
}
hipError_t launch_sample(const float* mu, const float* ls, const float* eps, float* z0, float* klterm, int n, int d,
                         int stride, int es, hipStream_t s) {
  hipLaunchKernelGGL(sample_kernel, dim3((n * d + 255) / 256), dim3(256), 0, s, mu, ls, eps, z0, klterm, n, d, stride, es);
  return hipGetLastError();
}
hipError_t launch_sample_bwd(const float* mu, const float* ls, const float* eps, const float* dz0, float* dmu, float* dls,
                             int n, int d, int stride, int es, float klw, hipStream_t s) {
  hipLaunchKernelGGL(sample_bwd_kernel, dim3((n * d + 255) / 256), dim3(256), 0, s, mu, ls, eps, dz0, dmu, dls, n, d, stride, es,
                     klw);
  return hipGetLastError();
}

// MADE x2 + IAF backward-data (layers.py:641-650, 735-853): z = (z0 - m(z0)) / exp(s(z0)); the MADE parameters are
// never trained (train_IAN.py:184-194), only dL/dz0 is needed.  wts/bias as in made_iaf_kernel (pre-masked), and the
// same reference wiring: h1 = relu(z0.W0+b0), h2 = relu(h1.W0+b0), out = h2.W1+b1 + h1.WD+bD  (see made_iaf_kernel).
//   u = dz / exp(s);  v_m = -u;  v_s = -dz*z;   per MADE: dh2 = (W1 v) [h2>0];  dh1 = (W0 dh2 + WD v) [h1>0];
//   dz0 = u + W0_m dh1_m + W0_s dh1_s
__global__ __launch_bounds__(128) void made_iaf_bwd_kernel(const float* __restrict__ z0, const float* __restrict__ dz,
                                                           float* __restrict__ dz0, const float* __restrict__ wts,
                                                           const float* __restrict__ bias, int d, int zs) {
  __shared__ float zin[128], hm[128], hl[128], gm[128], gl[128], vm[128], vl[128], d2m[128], d2l[128], d1m[128], d1l[128];
  const int t = threadIdx.x, row = blockIdx.x;
  const int dd = d * d;
  zin[t] = (t < d) ? z0[(size_t)row * zs + t] : 0.f;
  __syncthreads();
  float am = 0.f, al = 0.f;
  if (t < d) {
    am = bias[0 * d + t];
    al = bias[3 * d + t];
    for (int i = 0; i < d; ++i) {
      am = fmaf(zin[i], wts[0 * dd + i * d + t], am);
      al = fmaf(zin[i], wts[3 * dd + i * d + t], al);
    }
  }
  hm[t] = am > 0.f ? am : 0.f;
  hl[t] = al > 0.f ? al : 0.f;
  __syncthreads();
  am = 0.f, al = 0.f;
  if (t < d) {
    am = bias[0 * d + t];
    al = bias[3 * d + t];
    for (int i = 0; i < d; ++i) {
      am = fmaf(hm[i], wts[0 * dd + i * d + t], am);
      al = fmaf(hl[i], wts[3 * dd + i * d + t], al);
    }
  }
  gm[t] = am > 0.f ? am : 0.f;
  gl[t] = al > 0.f ? al : 0.f;
  __syncthreads();
  float u = 0.f;
  if (t < d) {
    float om = bias[1 * d + t], dm = bias[2 * d + t], ol = bias[4 * d + t], dl = bias[5 * d + t];
    for (int i = 0; i < d; ++i) {
      om = fmaf(gm[i], wts[1 * dd + i * d + t], om);
      dm = fmaf(hm[i], wts[2 * dd + i * d + t], dm);
      ol = fmaf(gl[i], wts[4 * dd + i * d + t], ol);
      dl = fmaf(hl[i], wts[5 * dd + i * d + t], dl);
    }
    const float mu = om + dm, ls = ol + dl;
    const float z = (zin[t] - mu) / expf(ls);
    const float g = dz[(size_t)row * zs + t];
    u = g / expf(ls);
    vm[t] = -u;       // dL/d(mu output)
    vl[t] = -g * z;   // dL/d(ls output)
  } else {
    vm[t] = 0.f;
    vl[t] = 0.f;
  }
  __syncthreads();
  // dh2[i] = (sum_j v[j]*W1[i][j]) * (h2[i] > 0)
  float a = 0.f, b = 0.f;
  if (t < d) {
    for (int j = 0; j < d; ++j) {
      a = fmaf(vm[j], wts[1 * dd + t * d + j], a);
      b = fmaf(vl[j], wts[4 * dd + t * d + j], b);
    }
  }
  d2m[t] = (gm[t] > 0.f) ? a : 0.f;
  d2l[t] = (gl[t] > 0.f) ? b : 0.f;
  __syncthreads();
  // dh1[i] = (sum_j dh2[j]*W0[i][j] + v[j]*WD[i][j]) * (h1[i] > 0)
  a = 0.f, b = 0.f;
  if (t < d) {
    for (int j = 0; j < d; ++j) {
      a = fmaf(d2m[j], wts[0 * dd + t * d + j], a);
      a = fmaf(vm[j], wts[2 * dd + t * d + j], a);
      b = fmaf(d2l[j], wts[3 * dd + t * d + j], b);
      b = fmaf(vl[j], wts[5 * dd + t * d + j], b);
    }
  }
  d1m[t] = (hm[t] > 0.f) ? a : 0.f;
  d1l[t] = (hl[t] > 0.f) ? b : 0.f;
  __syncthreads();
  if (t < d) {
    float g = u;
    for (int j = 0; j < d; ++j) {
      g = fmaf(d1m[j], wts[0 * dd + t * d + j], g);
      g = fmaf(d1l[j], wts[3 * dd + t * d + j], g);
    }
    dz0[(size_t)row * zs + t] = g;
  }
}
hipError_t launch_made_iaf_bwd(const float* z0, const float* dz, float* dz0, const float* wts, const float* bias, int n,
                               int d, int zs, hipStream_t s) {
  if (d > 128) return hipErrorInvalidValue;
  hipLaunchKernelGGL(made_iaf_bwd_kernel, dim3(n), dim3(128), 0, s, z0, dz, dz0, wts, bias, d, zs);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// losses between two tensors (train_IAN.py:169,244,279).  Strided views: element (r, c) at r*stride + c, c < C.
//   mode 0 (pixel_loss):  v1 = 2*|a - b + 1e-8|,  v2 = (a-b)^2 ;  da (+)= w * 2*sign(a - b + 1e-8)
//   mode 1 (feature MSE): v1 = (a-b)^2          ;  da (+)= w * 2*(a-b)   (b is constant)
// partial[block][2] then sum_finalize.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pair_loss_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                        float* __restrict__ da, long long rows, int C, int stride,
                                                        int mode, float w, int accumulate, float* __restrict__ partial) {
  __shared__ float red[2][256];
  const long long total = rows * C;
  float s1 = 0.f, s2 = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long r = i / C;
    const int c = (int)(i % C);
    const size_t o = (size_t)r * stride + c;
    const float d = a[o] - b[o];
    float g;
    if (mode == 0) {
      const float e = d + 1e-8f;
      s1 += 2.f * fabsf(e);
      s2 += d * d;
      g = 2.f * (e > 0.f ? 1.f : (e < 0.f ? -1.f : 0.f));
    } else {
      s1 += d * d;
      g = 2.f * d;
    }
    if (da) da[o] = (accumulate ? da[o] : 0.f) + w * g;
  }
  red[0][threadIdx.x] = s1;
  red[1][threadIdx.x] = s2;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
      red[0][threadIdx.x] += red[0][threadIdx.x + s];
      red[1][threadIdx.x] += red[1][threadIdx.x + s];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    partial[(size_t)blockIdx.x * 2] = red[0][0];
    partial[(size_t)blockIdx.x * 2 + 1] = red[1][0];
  }
}
// out[q] = scale * sum_k partial[k*width + q]   for q < width <= 64   (single block, fixed order: wave g takes k = g mod 4 in
// four interleaved accumulators -- 16 loads in flight per column instead of one dependent chain -- then (0+1)+(2+3) twice)
__global__ __launch_bounds__(256) void sum_finalize_kernel(const float* __restrict__ partial, int n, int width, float scale,
                                                           float* __restrict__ out) {
  __shared__ float red[256];
  const int q = threadIdx.x & 63, g = threadIdx.x >> 6;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  if (q < width) {
    for (int k = g; k < n; k += 16) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int kk = k + 4 * j;
        const float v = partial[(size_t)min(kk, n - 1) * width + q];
        acc[j] += kk < n ? v : 0.f;
      }
    }
  }
  red[threadIdx.x] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
  __syncthreads();
  if (g == 0 && q < width) out[q] = ((red[q] + red[q + 64]) + (red[q + 128] + red[q + 192])) * scale;
}
hipError_t launch_pair_loss(const float* a, const float* b, float* da, long long rows, int C, int stride, int mode,
                            float w, int accumulate, float* partial, int nblocks, float scale, float* out, hipStream_t s) {
  hipLaunchKernelGGL(pair_loss_kernel, dim3(nblocks), dim3(256), 0, s, a, b, da, rows, C, stride, mode, w, accumulate, partial);
  hipLaunchKernelGGL(sum_finalize_kernel, dim3(1), dim3(256), 0, s, partial, nblocks, 2, scale, out);
  return hipGetLastError();
}
// column sums of a small [n][width] array (per-sample loss terms, KL terms): out[q] = scale * sum_r x[r*width+q]
hipError_t launch_sum_rows(const float* x, int n, int width, float scale, float* out, hipStream_t s) {
  if (width > 64) return hipErrorInvalidValue;
  hipLaunchKernelGGL(sum_finalize_kernel, dim3(1), dim3(256), 0, s, x, n, width, scale, out);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// orthogonal regulariser (train_IAN.py:158-165) on a 4-D W (A,B,K,K) in reference layout:
//   y[a,i,j] = sum_{b,k} W[a,b,i,k]*W[a,b,j,k] - [i==j];   value = sum |y|;   dW[a,b,i,k] += c * 2*sum_j sign(y[a,i,j])*W[a,b,j,k]
// one block per a.  value partial per block -> vals[a].
// ------------------------------------------------------------------------------------------------
template <int K>
__global__ __launch_bounds__(256) void ortho_kernel(const float* __restrict__ W, float* __restrict__ dW, int B, float c,
                                                    float* __restrict__ vals) {
  __shared__ float red[256];
  __shared__ float y[K * K];
  const int a = blockIdx.x;
  const float* Wa = W + (size_t)a * B * K * K;
  float part[K * K];
#pragma unroll
  for (int e = 0; e < K * K; ++e) part[e] = 0.f;
  for (int b = threadIdx.x; b < B; b += 256) {
    float w[K * K];
#pragma unroll
    for (int e = 0; e < K * K; ++e) w[e] = Wa[(size_t)b * K * K + e];
#pragma unroll
    for (int i = 0; i < K; ++i)
#pragma unroll
      for (int j = 0; j < K; ++j) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < K; ++k) s += w[i * K + k] * w[j * K + k];
        part[i * K + j] += s;
      }
  }
  for (int e = 0; e < K * K; ++e) {
    red[threadIdx.x] = part[e];
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
      __syncthreads();
    }
    if (threadIdx.x == 0) y[e] = red[0] - ((e / K == e % K) ? 1.f : 0.f);
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    float v = 0.f;
    for (int e = 0; e < K * K; ++e) v += fabsf(y[e]);
    vals[a] = v;
  }
  if (!dW) return;
  float* dWa = dW + (size_t)a * B * K * K;
  for (int b = threadIdx.x; b < B; b += 256) {
    float w[K * K];
#pragma unroll
    for (int e = 0; e < K * K; ++e) w[e] = Wa[(size_t)b * K * K + e];
#pragma unroll
    for (int i = 0; i < K; ++i)
#pragma unroll
      for (int k = 0; k < K; ++k) {
        float g = 0.f;
#pragma unroll
        for (int j = 0; j < K; ++j) {
          const float yy = y[i * K + j];
          g += (yy > 0.f ? 1.f : (yy < 0.f ? -1.f : 0.f)) * w[j * K + k];
        }
        dWa[(size_t)b * K * K + i * K + k] += 2.f * c * g;
      }
  }
}
hipError_t launch_ortho(const float* W, float* dW, int A, int B, int K, float c, float* vals, hipStream_t s) {
  if (K == 5) hipLaunchKernelGGL(ortho_kernel<5>, dim3(A), dim3(256), 0, s, W, dW, B, c, vals);
  else if (K == 3) hipLaunchKernelGGL(ortho_kernel<3>, dim3(A), dim3(256), 0, s, W, dW, B, c, vals);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// lasagne.updates.adam (App. B.7) on a flat parameter group:
//   m = b1*m + (1-b1)*g ; v = b2*v + (1-b2)*g^2 ; p -= a_t * m / (sqrt(v) + eps),   a_t = lr*sqrt(1-b2^t)/(1-b1^t)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, long long n, float a_t,
                                                   float b1, float b2, float eps) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float gi = g[i];
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    p[i] -= a_t * mi / (sqrtf(vi) + eps);
  }
}
hipError_t launch_adam(float* p, const float* g, float* m, float* v, long long n, float a_t, float b1, float b2, float eps,
                       hipStream_t s) {
  hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n)), dim3(256), 0, s, p, g, m, v, n, a_t, b1, b2, eps);
  return hipGetLastError();
}

}  // namespace ian

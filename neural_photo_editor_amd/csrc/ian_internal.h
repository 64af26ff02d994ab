// Internal declarations shared by the runtime (ian_runtime.cpp) and the device code (kernels_*.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ian {

// ---------------------------------------------------------------------------------------------
// "tap GEMM": every dense-channel layer of the IAN (5x5/s2 conv, 5x5/s2 transposed conv split into
// its 4 output-parity classes, composite multiscale-dilated 3x3, dense) is one implicit GEMM
//     C[m][co] = sum_t sum_ci  X[pixel(m) + d_t][ci] * Wt[t][co][ci]
// over a list of taps t = (dy,dx, weight slab).  m enumerates (image, qy, qx) of a per-class output
// grid; input pixel = q*si + b + d_t, output pixel = q*so + p_class.
// ---------------------------------------------------------------------------------------------
struct TgTap {
  int dy, dx;
};
struct TgClass {
  int ntaps, tap0;  // taps[tap0 .. tap0+ntaps)
  int py, px;       // output parity offset
  long long w_off;  // float offset of this class's first weight slab; slab t at w_off + t*CoutPad*Cin
};
struct TgItem {  // one workgroup's job (host-built table, 64 B = one s_load_dwordx16)
  int cls, m0, n0;
  int ks0, ks1;  // K-step range [ks0,ks1) of 32-channel steps over (tap, ci-chunk)
  int slab;      // split-K: slab tile index; -1 = direct epilogue
  int tile;      // split-K: index of the output tile in the TgTile table (fused combine: counter + slab range)
  // the item's TgClass and the tap its K range starts in, COPIED here (fill_item_class, ian_rt_schedule.inc) so that the kernel's
  // prologue is one table fetch instead of three dependent ones (item -> class -> tap: ~0.5-1 us each from a cold L2, before the
  // first operand load can be addressed); the tables stay as they are for the taps after the first and for the split-bf16 kernel
  int ntaps, tap0, py, px;
  int dy0, dx0;  // taps[tap0 + ks0 / (Cin / 32)]
  int pad1;
  long long w_off;
};
static_assert(sizeof(TgItem) == 64, "TgItem is fetched as one 64-byte scalar load");
struct TgTile {  // reduce pass: one output tile
  int cls, m0, n0, slab0, nsplit;
  int py, px;  // the class's output parity offset, copied here so that the reduce pass needs ONE table load, not two dependent ones
  int pad2;
};

enum { TG_EPI_FWD = 0, TG_EPI_BWD = 1 };

// Batch statistics computed in the epilogue of the launch that PRODUCES the tensor (round 5: the training step's colstats passes
// re-read every normalised tensor -- 5 GB per update).  Per output row tile and channel, float64 partial sums over the tile's rows:
//   mode 1 (forward, Lasagne batch_norm statistics): s1 = sum v, s2 = sum v*v of the stored value v, every element widened first;
//   mode 2 (backward: dbeta, dgamma): g = v * act'(a), s1 = sum g, s2 = sum g * xhat, xhat = (yraw - mean) * inv_std -- float32
//           products, four rows added in float32, the rest in float64 (the arithmetic of colstats_kernel, kernels_train.hip).
// partial[chunk][2][C] doubles, chunk = (row tile index) * ncls + class: the second stage (tree over chunks) is unchanged, and
// the chunk boundaries (multiples of the tile height within the (image, pixel) row order) do not depend on the batch.
// Only launches whose tiles are not split over K carry it (the host falls back to colstats otherwise).
struct TgStats {
  double* partial = nullptr;
  const float* a = nullptr;       // mode 2: post-activation forward value (for act')
  const float* yraw = nullptr;    // mode 2: pre-normalisation value
  const float* mean = nullptr;
  const float* inv_std = nullptr;
  int mode = 0, act = 0, C = 0, ncls = 1;
};

struct TgEpilogue {
  const float* scale;  // per output channel (folded BN gamma*inv_std) or nullptr (=1)
  const float* shift;  // per output channel (folded beta/bias) or nullptr (=0)
  const float* res;    // residual tensor, same layout as y, added before the affine; or nullptr
  const float* yfwd;   // BWD: forward output of the layer whose pre-activation gradient is produced
  int act;             // enum ian_act
  int scale_period;    // 0: scale/shift indexed by channel; >0: by (element offset % period) -- per-feature
                       // batch-norm of a dense layer whose output is viewed as an NHWC map (IAN_simple.py:129-139)
  int mode;            // TG_EPI_FWD: y = act((acc+res)*scale+shift)
                       // TG_EPI_BWD: y = (acc [+res]) * act'(yfwd) * scale   (gradient wrt pre-affine value)
  TgStats st;          // statistics of the stored values (training step), or mode 0
};

struct TgParams {
  const float* x;
  const float* w;
  float* y;
  float* slab;
  const TgItem* items;
  const TgClass* classes;
  const TgTap* taps;
  TgEpilogue epi;
  int M;                     // images * QH * QW
  int IH, IW, Cin;           // Cin = padded input channels = pixel stride of x (multiple of 32)
  int qw_shift, qhw_shift;   // QW = 1<<qw_shift, QH*QW = 1<<qhw_shift
  int si, by, bx;            // input coordinate = q*si + b + d
  int so;                    // output coordinate = q*so + p
  int OH, OW, Cout, y_stride;
  int CoutPad;
  unsigned x_bytes, w_bytes;  // buffer-descriptor extents (out-of-range offsets read as zero)
  int variant;                // K-loop schedule (kernels_tapgemm.hip)
  const TgTile* tiles;        // split-K with the combine fused into this launch (fused != 0): the tile table ...
  int* counters;              // ... and one arrival counter per tile (zero between launches)
  float* raw;                 // fused == 2: one zero-at-rest accumulation tile per output tile (float atomics)
  int fused;                  // 0: slabs + a tapgemm_reduce launch; 1: write-through slabs combined by the tile's last arriver;
                              // 2: atomic accumulation into `raw`, epilogue by the last arriver (kernels_tapgemm.hip)
  const unsigned short* wsplit;  // tapgemm_bf16x3_kernel: the weights pre-split into a bf16 hi plane followed by a lo plane (or nullptr)
  int bf_sched;                  // ... and its K-loop schedule (0..2)
  int noepi;                     // libian_ablation.so: skip everything behind the K loop (timing-only)
  unsigned y_bytes;              // extent of y (and of res / yfwd, same layout) for the buffer-descriptor epilogue; 0: above the descriptor range, general epilogue
};

struct TgReduceParams {
  const float* slab;
  float* y;
  const TgTile* tiles;
  const TgClass* classes;
  TgEpilogue epi;
  int M, qw_shift, qhw_shift, so, OH, OW, Cout, y_stride;
};

constexpr int TG_VARIANT_BF16X3 = 5;   // TgParams::variant of tapgemm_bf16x3_kernel (opt-in, ian_set_option("tg_bf16x3", 1))
enum TgConfig { TG_128x128 = 0, TG_128x64 = 1, TG_64x64 = 2, TG_32x128 = 3, TG_256x128 = 4, TG_128x32 = 5,
                TG_128x128W8 = 6 /* 128x128 tile, 8 waves of 64x32 */, TG_128x64W8 = 7 /* 128x64 tile, 8 waves of 32x32 */,
                TG_NCONFIG = 8 };
struct TgShape {
  int bm, bn;
};
static inline TgShape tg_shape(int cfg) {
  switch (cfg) {
    case TG_128x128:
    case TG_128x128W8: return {128, 128};
    case TG_128x64:
    case TG_128x64W8: return {128, 64};
    case TG_64x64: return {64, 64};
    case TG_32x128: return {32, 128};
    case TG_128x32: return {128, 32};
    default: return {256, 128};
  }
}

// tile configurations whose kernel carries the in-launch split-K combine (TgParams::fused != 0): the small 4-wave tiles the
// batch-1 chains use.  The large tiles are compiled WITHOUT that epilogue (it cost the 256x128 tile 95 spilled VGPRs).
static inline bool tg_fuse_supported(int cfg) {
#ifdef IAN_NO_TG_FUSE   // libian_nofuse.so: the epilogue is compiled out of every tile (A/B build, build.py)
  (void)cfg;
  return false;
#endif
  return cfg == TG_32x128 || cfg == TG_64x64 || cfg == TG_128x64 || cfg == TG_128x32;
}

// tile configurations tapgemm_bf16x3_kernel is instantiated for (its staging pass covers 16 x waves rows: 4 threads per row)
static inline bool tg_bf16x3_supported(int cfg) {
  return cfg == TG_128x128 || cfg == TG_128x64 || cfg == TG_64x64 || cfg == TG_128x128W8;   // 256x128 compiles too but spills (1152 B of scratch)
}

// launches that can carry the GEMM-epilogue batch statistics (TgStats): a separate kernel instantiation exists for the
// register-staged K-loop schedules 1 / 2 of every tile up to 128 x 128
// -- in libian_ablation.so.  The product library carries none of them (measured: the training update got 4 % SLOWER; the colstats
// passes this replaces are HBM-bound and already run under the weight-gradient GEMMs of the second stream, the epilogue version
// puts the same bytes on the MFMA-bound kernel's critical path): there every request is answered with 0 chunks.
static inline bool tg_stats_supported(int cfg, int variant) {
#ifdef IAN_ABLATION
  return cfg != TG_256x128 && (variant == 1 || variant == 2);
#else
  (void)cfg; (void)variant;
  return false;
#endif
}

// one-image form of the 5x5/s2 transposed conv and of its backward-data (kernels_b1.hip): whole contraction per
// workgroup (4x4 output positions x 16 channels), weights repacked as one contiguous stream per workgroup
struct B1Params {
  const float* x;      // NHWC input of this kernel (activation, or dL/d(pre-epilogue output) for the backward form)
  const float* w;      // packed [class][channel slice][tap][Cr/32][64 lanes][8]
  float* y;
  const float* scale;  // as TgEpilogue
  const float* shift;
  const float* yfwd;
  const float* res;
  long long cls_off[4];  // float offset of each output-parity class's weights (backward form: one class)
  int act, bwd, scale_period;
  int IH, IW, Cr, xs;    // input grid, reduction channels (multiple of 32), input pixel stride
  int OH, OW, ys;        // output grid and pixel stride
  int ntiles, tiles_x, nslices;  // 4x4 position blocks (per class), blocks per row, 16-channel output slices
  unsigned x_bytes;              // extent of x for the buffer descriptor (out-of-range offsets read as zero)
  int kshift;                    // log2(Cr / 32)
  int dbg;                       // scripts/ubench/b1conv_bench.cpp only (timing ablations); 0 in the product
};
hipError_t launch_b1conv(const B1Params& p, int mode, hipStream_t s);

hipError_t launch_tapgemm(int cfg, const TgParams& p, int nitems, hipStream_t s);
// fp32 weights -> bf16 hi plane | lo plane (2 n values) for tapgemm_bf16x3_kernel
hipError_t launch_wsplit(const float* w, unsigned short* out, long long n, hipStream_t s);
// ian_box_probe: register-only fp32-MFMA loop, `blocks` workgroups of 256 threads, 4 x iters MFMAs per wave (kernels_misc.hip)
hipError_t launch_box_probe(const float* in, float* out, int blocks, int iters, hipStream_t s);
// kp > 1: four lanes share the slabs of one output element (few tiles, many slabs: batch 1)
hipError_t launch_tapgemm_reduce(int cfg, const TgReduceParams& p, int ntiles, int kp, hipStream_t s);

// ---------------------------------------------------------------------------------------------
// edge layers (3 image channels) and small ops
// ---------------------------------------------------------------------------------------------
// enc_conv1: x NCHW [n,3,H,W] -> y NHWC [n,H/2,W/2,Cout], 5x5 s2 p2 + bias + act. w packed [75][Cout].
hipError_t launch_conv1_nchw(const float* x, const float* w, const float* scale, const float* shift, float* y, int n,
                             int H, int W, int Cout, int act, hipStream_t s);
// dec_out: x NHWC [n,H,W,Cin] -> y NCHW [n,Cout(<=4),2H,2W], 5x5 s2 transposed conv + affine + act.
// w packed [25 (ky,kx)][Cout4=4][Cin].
hipError_t launch_deconv_out_nchw(const float* x, const float* w, const float* scale, const float* shift, float* y,
                                  int n, int H, int W, int Cin, int Cout, int act, hipStream_t s);
hipError_t launch_dense_fwd_gemv(const float* x, const float* w, int K, int nout, const float* scale, const float* shift, int act,
                                 float* y, hipStream_t s);
// upd_z != nullptr: row j's workgroup also applies the brush update z[j] += cg[0] * (dz[j] * cg[1]) (and mirrors z, dz)
hipError_t launch_dense_bwd_gemv(const float* g, const float* wb, int rows, int K, const float* res, float* dz, float* upd_z,
                                 const float* upd_cg, float* z_mirror, float* g_mirror, hipStream_t s);
hipError_t launch_deconv_out_px(const float* x, const float* w, const float* scale, const float* shift, float* y, float* mirror, int n,
                                int H, int W, int Cin, int Cout, int act, hipStream_t s);

// y = act(x*scale[c]+shift[c]) on NHWC tensors (pixel stride = stride)
hipError_t launch_affine(const float* x, float* y, const float* scale, const float* shift, long long npix, int C,
                         int stride, int act, hipStream_t s);
// MADE x2 + IAF on latent rows (stride zs): z' = (z - made_mu(z)) / exp(made_ls(z)); weights pre-masked [6][100][100]+[6][100]
hipError_t launch_made_iaf(const float* z, float* zo, const float* wts, const float* bias, int n, int d, int zs,
                           hipStream_t s);
// beta_layer x3 + concat -> NCHW output [n,3,H,W]; R,G,B are NHWC with 2 channels (stride rs)
hipError_t launch_beta(const float* R, const float* G, const float* B, float* y, int n, int hw, int rs, hipStream_t s);
// channel concat of two NHWC tensors
hipError_t launch_concat2(const float* a, int ca, int sa, const float* b, int cb, int sb, float* y, int sy,
                          long long npix, hipStream_t s);
// layout / copy helpers
hipError_t launch_rows_copy(const float* src, int src_stride, float* dst, int dst_stride, int n, int c, hipStream_t s);
hipError_t launch_nhwc_to_nchw(const float* src, int stride, float* dst, int n, int hw, int c, hipStream_t s);
// loss seeds for the latent-brush gradients (API.py:59,64): writes d loss / d x_hat (NCHW [1,3,H,W], zero outside patch)
hipError_t launch_patch_seed(const float* xhat, const float* rgb, float* g, int H, int W, int c1, int r1, int c2,
                             int r2, int mode, hipStream_t s);
hipError_t launch_patch_seed_dev(const float* xhat, const float* rgb, float* g, int H, int W, const int* patch, int mode,
                                 hipStream_t s);
// backward of dec_out-like layer: g NCHW [n,Cout,2H,2W] (already multiplied by act') -> dx NHWC [n,H,W,Cin]
hipError_t launch_deconv_out_bwd(const float* g, const float* w, float* dx, const float* yfwd, const float* scale,
                                 int n, int H, int W, int Cin, int Cout, int act, hipStream_t s);
// batch-1 latent brush: seed (rectangle in device memory) + output activation derivative + the same backward-data, one launch
hipError_t launch_deconv_out_bwd_seed(const float* xhat, const float* rgb, const int* patch, int mode, int out_act,
                                      const float* oscale, const float* w, float* dx, const float* yfwd, const float* scale, int H,
                                      int W, int Cin, int Cout, int act, hipStream_t s);
// elementwise: g = g * act'(y) * scale  (NCHW small tensors, c channels of hw pixels)
hipError_t launch_dact_nchw(float* g, const float* y, const float* scale, int n, int c, int hw, int act,
                            hipStream_t s);

// few-filter MDCL forward on the VALU (RGB-Beta head).  One launch computes up to MH_MAXCO output filters that read
// the SAME input map with the SAME tap list -- the filters may belong to different layers (IAN.py:183-199: R, G_a
// and B_a all read the 128-channel feature map), each with its own weights, destination, residual and activation.
constexpr int MH_MAXCO = 8;
struct MdcHeadArgs {
  const float* x;              // NHWC, pixel stride xs
  const float* w[MH_MAXCO];    // per output filter: its row in a forward slab; tap t at + t*w_tap_stride
  const float* res[MH_MAXCO];  // per output filter: residual tensor (same addressing as y) or nullptr
  float* y[MH_MAXCO];          // per output filter: destination tensor base
  float scale[MH_MAXCO], shift[MH_MAXCO];
  int ys[MH_MAXCO], yc[MH_MAXCO], act[MH_MAXCO];  // pixel stride, channel index, activation
  int H, W, xs, ntaps;
  long long w_tap_stride;
  signed char dy[48], dx[48];
};
hipError_t launch_mdc_head(const MdcHeadArgs& a, int n, int Cin, int Cout, hipStream_t s);
// MDCL with <= 4 real input channels on the VALU (forward of G_b / B_b, backward-data of every 2-filter head layer)
struct MdcThinArgs {
  const float* x;  // NHWC, pixel stride xs (>= 4 floats readable per pixel)
  const float* w;  // slab [tap][CoutPad][CinPad]
  const float* res;
  const float* scale;
  const float* shift;
  float* y;        // NHWC, pixel stride ys
  int n, H, W, xs, ys, Cout, CoutPad, CinPad, ntaps, act;
  int no_tile;     // 1: never take the LDS-staged form (A/B tests)
  signed char dy[48], dx[48];
};
hipError_t launch_mdc_thin(const MdcThinArgs& a, hipStream_t s);
// few-filter MDCL backward-weight on the VALU; partial: [nblocks][ntaps][2 or 4][Cin] floats
struct MdcHeadWgradArgs {
  const float* x;   // layer input, NHWC, pixel stride xs
  const float* dy;  // gradient wrt the pre-activation output, NHWC, pixel stride dys
  float* partial;
  int H, W, xs, dys, ntaps, total_tiles;
  signed char dy_[48], dx_[48];
};
hipError_t launch_mdc_head_wgrad(const MdcHeadWgradArgs& a, int nblocks, int Cin, int Cout, float* dS, int f_rows,
                                 int f_cols, hipStream_t s);
// RGB-Beta head in two launches (kernels_head.hip).  head6: the six filters that read the 128-channel map (three
// 2-filter MDCL layers with one tap list) -> compact [n][H][W][8] map (filters 0..5, 2 floats padding).
struct HeadFusedArgs {
  const float* x;          // NHWC, pixel stride xs, 128 channels
  const float *w0, *w1, *w2;  // forward slabs [tap][CoutPad][128] of the three layers (rows 0,1 used)
  float* out;              // compact map
  const int* itab;         // 44 slots: [0,12) taps with dy = 0, then 4 per dy = -4..-1, 1..4; tap id | (dx+64) << 8, empty = -1
  const float* ftab;       // [0..5] scale, [8..13] shift, [16..21] activation code of the six filters
  int H, W, xs, ntaps, bands, halo;
  long long w_tap_stride;
};
hipError_t launch_head6(const HeadFusedArgs& a, int n, hipStream_t s);
hipError_t launch_head6_zbuild(const float* dy0, const float* dy1, const float* dy2, int dys, float* Z, int zs, int H, int W,
                               int ntaps, const int* taps, long long npix, hipStream_t s, int max_shift /* largest |dy|, |dx| of the taps; < 0: the per-(pixel, tap) form */);
hipError_t launch_head6_wcat(const float* slab, int f_rows, int f_cols, int kk, int ntaps, float* wcat, int nout, hipStream_t s);
hipError_t launch_head6_dS_scatter(const float* dWref, int nout, int kk, int ntaps, float* dS, int f_rows, int f_cols, hipStream_t s);
hipError_t launch_head6_scatter(const float* comp, float* y0, float* y1, float* y2, int ys, long long npix, hipStream_t s);
struct HeadTailArgs {
  const float* comp;       // compact map: (R0,R1,Ga0,Ga1,Ba0,Ba1,-,-) per pixel
  const float *w_gb, *w_bb;  // forward slabs of G_b (2 in) and B_b (4 in): [tap][CoutPad][CinPad]
  float* out;              // NCHW [n][3][H][W]
  long long gb_tap_stride, bb_tap_stride;
  int gb_cin, bb_cin;      // CinPad of the two slabs
  float scale_g[2], shift_g[2], scale_b[2], shift_b[2];
  int act_g, act_b, H, W, ntaps;
  signed char dy[48], dx[48];
};
hipError_t launch_head_tail(const HeadTailArgs& a, int n, hipStream_t s);

// dec_out of IAN_simple on the matrix cores (kernels_head.hip): x NHWC [n][H][32][128] -> y NCHW [n][Cout][2H][64]
struct DeconvSmallArgs {
  const float* x;
  const float* w;      // edge packing [25 (ky,kx)][4][128]
  const float* scale;  // per output channel or nullptr
  const float* shift;
  float* y;
  int H, W, xs, bands, act;
  int balanced;        // Cout = 3: the 16 x 16 MFMA tiling, five blocks per SIMD (kernels_head.hip; option dec_out_bal)
};
hipError_t launch_deconv_small(const DeconvSmallArgs& a, int n, int Cout, hipStream_t s);

// NPE.paint's photo blend and the uint8 image conversion on the device (kernels_npe.hip; NPE.py:218-231, 110, 261)
struct PhotoBlendArgs {
  const float* xhat;           // decoder output, NCHW [3][64][64]
  const unsigned char* recon;  // RECON uint8 [3][64][64]
  const float* error;          // ERROR float32 [3][64][64]
  unsigned char* im;           // IM uint8 [3][64][64]
  double* mask;                // MASK float64 [64][64] or nullptr
  double w[8];                 // Gaussian weights, centre outwards (scipy _gaussian_kernel1d)
  int radius;
};
hipError_t launch_photo_blend(const PhotoBlendArgs& a, hipStream_t s);
hipError_t launch_latent_update(float* z, const float* g, const float* cg, int n, float* z_mirror, float* g_mirror, hipStream_t s);
hipError_t launch_keep_warm(const int* flag, long long max_ticks, hipStream_t s);   // experiment: see kernels_npe.hip
hipError_t launch_to_uint8(const float* x, unsigned char* y, long long n, hipStream_t s);

// identity-edge gradient hand-over: gd[p,c] (+)= gs[p,coff+c] * act'(y[p,c]) * scale[c]   (NHWC, strides ss / ds)
hipError_t launch_grad_pass(const float* gs, int ss, int coff, float* gd, const float* y, int ds, const float* scale,
                            long long npix, int C, int act, int accumulate, hipStream_t s);
// backward of beta_layer x3 + concat
struct BetaBwdArgs {
  const float* v[3];      // forward values of the three 2-channel maps (after their sigmoid)
  float* g[3];            // gradient buffers of the same maps (pre-epilogue of their producers)
  const float* scale[3];  // producer scale vectors (or nullptr)
  int act[3];             // producer activations
  int accumulate[3];
};
hipError_t launch_beta_bwd(const float* gout, const BetaBwdArgs& a, int n, int hw, int rs, hipStream_t s);

// ---------------------------------------------------------------------------------------------
// training step (train_IAN.py:47-352): backward-weight tap GEMM, parameter (re)packing, elementwise / reductions
// ---------------------------------------------------------------------------------------------
struct WgItem {  // one workgroup of tapwgrad: a (tap, Cout tile, Cin tile, pixel range) -> one partial-slab tile
  int cls, tap;  // class index, GLOBAL tap index (== slab index of the forward packing)
  int co0, ci0;
  int m0, m1;    // contraction range over (image, qy, qx), multiples of 32 except the tail
  int split;     // which partial slab
  int pad;
};
struct WgParams {
  const float* x;   // layer input, NHWC, pixel stride Cin
  const float* dy;  // gradient wrt the layer's (pre-epilogue) output, NHWC, pixel stride dy_stride
  float* partial;   // [nsplit][ntaps][CoutPad][CinPad]
  const WgItem* items;
  const TgClass* classes;
  const TgTap* taps;
  int M, IH, IW, Cin, qw_shift, qhw_shift, si, by, bx, so, OH, OW;
  int dy_stride, CoutPad, CinPad;
  long long slab_total;
  unsigned x_bytes, dy_bytes;
};
enum WgConfig { WG_128x128 = 0, WG_32x128 = 1, WG_128x32 = 2, WG_128x128W8 = 3 /* 8 waves of 64x32 */, WG_128x128P = 4 /* the same tile, software-pipelined K loop */, WG_128x128P2 = 5 /* ... rotated, three fragment buffers */,
                WG_128x128P3 = 6 /* ... P with the loads of two K-steps in flight */ };
hipError_t launch_tapwgrad(int cfg, const WgParams& p, int nitems, hipStream_t s);
hipError_t launch_wgrad_reduce(const float* partial, long long slab_total, int nsplit, const int* inv, float* out,
                               long long count, int accumulate, hipStream_t s);
// the slab -> reference map of a layer when it is affine (kernels_wgrad.hip wgrad_reduce_tiled_kernel); valid == false keeps the gather
struct WgReduceTiledDesc {
  bool valid = false;
  int ntaps = 0, CoutPad = 0, CinPad = 0, Cout = 0, Cin = 0, tco = 16, tci = 16, s_co = 0, s_ci = 0;
  bool ci_inner = true;                       // which axis walks the reference contiguously (stride = ntaps)
  const int* co_off = nullptr;                // device tables replacing co * s_co / ci * s_ci where a permutation makes that axis non-linear
  const int* ci_off = nullptr;
  unsigned char tap_off[48] = {0}, tap_inv[48] = {0};
};
hipError_t launch_wgrad_reduce_tiled(const WgReduceTiledDesc& d, const float* partial, long long slab_total, int nsplit, float* out,
                                     int accumulate, hipStream_t s);
hipError_t launch_pack_tiled(const WgReduceTiledDesc& d, const float* ref, float* slab, hipStream_t s);
hipError_t launch_gather_pack(const float* src, const int* map, float* dst, long long count, hipStream_t s);

constexpr int MDC_MAX_BRANCH = 5;  // base 3x3 + up to IAN_MAX_SCALES dilated branches
struct MdcPackArgs {
  const float* W;                      // (Cout,Cin,3,3) reference layout
  const float* coeff[MDC_MAX_BRANCH];  // per-filter coefficients of the 3x3 branches: [0] = base, then dilations in order
  const float* coeff_1x1;              // or nullptr
  float* slab_f;                       // [ntaps][f_rows][f_cols]
  float* slab_b;                       // [ntaps][b_rows][b_cols] (transposed)
  int ntaps, cout, cin, nbranch, f_rows, f_cols, b_rows, b_cols;
  int tap_start[48];                   // entries of tap t: [tap_start[t], tap_start[t+1])
  unsigned char ent_branch[48], ent_pq[48];
};
struct MdcCoeffGrads {
  float* d[MDC_MAX_BRANCH];
  float* d1x1;
};
hipError_t launch_mdc_pack(const MdcPackArgs& a, hipStream_t s);
hipError_t launch_mdc_unpack_grad(const MdcPackArgs& a, const float* dS, float* dW, const MdcCoeffGrads& g,
                                  int accumulate, hipStream_t s);

struct ColStatsArgs {
  const float* x;  // mode 0: the tensor; modes 1,2: dA
  const float* a;  // post-activation output (for act'), modes 1,2
  const float* y;  // raw (pre-batch-norm) value, mode 1
  const float* mean;
  const float* inv_std;
  double* partial; // [nchunks][2][C] float64 partial sums (kernels_train.hip: NUMERICS)
  long long rows;
  int C, stride, mode, act;
};
hipError_t launch_colstats(const ColStatsArgs& a, int nchunks, double* sums, hipStream_t s);
// single-process batch-norm statistics, second stage fused (kernels_train.hip: bn_finish_kernel / bn_bwd_finish_kernel)
hipError_t launch_bn_stats_affine(const ColStatsArgs& a, int nchunks, double* sums, float count, float eps, const float* gamma,
                                  const float* beta, float* mean, float* inv_std, float* scale, float* shift, float* run_mean,
                                  float* run_inv_std, float keep, float alpha, hipStream_t s);
hipError_t launch_bn_bwd_stats(const ColStatsArgs& a, int nchunks, double* sums, float* gbeta, int acc_beta, float* ggamma,
                               int acc_gamma, hipStream_t s);
// the second stages alone (partials already in place: GEMM-epilogue statistics, TgStats)
hipError_t launch_bn_finish(const double* partial, int nchunks, int C, double* sums, float count, float eps, const float* gamma,
                            const float* beta, float* mean, float* inv_std, float* scale, float* shift, float* run_mean,
                            float* run_inv_std, float keep, float alpha, hipStream_t s);
hipError_t launch_bn_bwd_finish(const double* partial, int nchunks, int C, double* sums, float* gbeta, int acc_beta, float* ggamma,
                                int acc_gamma, hipStream_t s);
// out[i] = pairwise tree over k < count of partial[k*width + i] (fixed order: see kernels_train.hip)
hipError_t launch_tree_sum(const double* partial, int count, int width, double* out, hipStream_t s);
hipError_t launch_bn_make_affine(const double* sums, float count, float eps, const float* gamma, const float* beta,
                                 int C, float* mean, float* inv_std, float* scale, float* shift, hipStream_t s);
hipError_t launch_bn_running(float* run_mean, const float* mean, float* run_inv_std, const float* inv_std, int C, float keep,
                             float alpha, hipStream_t s);
struct BnBwdArgs {
  const float* dA;
  const float* a;
  const float* y;
  const float* mean;
  const float* inv_std;
  const float* scale;
  const double* sums; // [2][C] float64 (sum g, sum g*xhat) or nullptr: activation backward only
  float* dy;
  long long rows;
  int C, stride, act;
  float count;
};
hipError_t launch_bn_bwd_apply(const BnBwdArgs& a, hipStream_t s);
hipError_t launch_axpy(float alpha, const float* x, float* y, long long n, int accumulate, hipStream_t s);
hipError_t launch_axpy_f64(double alpha, const double* x, float* y, long long n, int accumulate, hipStream_t s);
hipError_t launch_nchw_to_nhwc(const float* src, float* dst, int n, int hw, int c, int stride, hipStream_t s);
hipError_t launch_globalpool(const float* x, float* y, int n, int hw, int C, int xs, int ys, hipStream_t s);
hipError_t launch_globalpool_bwd(const float* dy, float* dx, int n, int hw, int C, int xs, int ys, int accumulate,
                                 hipStream_t s);
hipError_t launch_mb_weight(const float* theta, const float* lws, float* W, float* colscale, int nin, int ncol, hipStream_t s);
hipError_t launch_mb_weight_bwd(const float* theta, const float* colscale, const float* dW, float* dtheta, float* dlws,
                                int nin, int ncol, int accumulate, hipStream_t s);
hipError_t launch_mb_forward(const float* act_all, int nall, int as, int row0, int n, int nk, int nd, const float* bias,
                             const float* feat, int fs, int fin, float* mb, int ms, hipStream_t s);
hipError_t launch_mb_backward(const float* act_all, int nall, int as, int row0, int n, int nk, int nd, const float* df_all,
                              int dfs, float* dact, int das, hipStream_t s);
struct DiscHeadArgs {
  float* p;       // [n][3]
  float* loss;    // [n][4]: -log p[target0], -log p[target1], argmax==acc_target, 0
  int target[2];  // -1 = unused
  int acc_target;
};
hipError_t launch_disc_head(const float* mb, int ms, int nfeat, const float* Wd, int ncls, int n, const DiscHeadArgs& a,
                            hipStream_t s);
hipError_t launch_disc_head_bwd(const float* p, const float* Wd, int ncls, int nfeat, int n, int t0, float w0, int t1,
                                float w1, float* dlogits, float* dmb, int ms, hipStream_t s);
hipError_t launch_disc_head_wgrad(const float* mb, int ms, int nfeat, int n, const float* dlogits, int ncls, float* dWd,
                                  int accumulate, hipStream_t s);
hipError_t launch_sample(const float* mu, const float* ls, const float* eps, float* z0, float* klterm, int n, int d,
                         int stride, int es, hipStream_t s);
hipError_t launch_sample_bwd(const float* mu, const float* ls, const float* eps, const float* dz0, float* dmu, float* dls,
                             int n, int d, int stride, int es, float klw, hipStream_t s);
hipError_t launch_made_iaf_bwd(const float* z0, const float* dz, float* dz0, const float* wts, const float* bias, int n,
                               int d, int zs, hipStream_t s);
hipError_t launch_pair_loss(const float* a, const float* b, float* da, long long rows, int C, int stride, int mode,
                            float w, int accumulate, float* partial, int nblocks, float scale, float* out, hipStream_t s);
hipError_t launch_sum_rows(const float* x, int n, int width, float scale, float* out, hipStream_t s);
hipError_t launch_ortho(const float* W, float* dW, int A, int B, int K, float c, float* vals, hipStream_t s);
hipError_t launch_adam(float* p, const float* g, float* m, float* v, long long n, float a_t, float b1, float b2, float eps,
                       hipStream_t s);

}  // namespace ian

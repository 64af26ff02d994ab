/* ian.h -- C ABI of libian.so: the MI355X-native IAN compute path.
 *
 * The reference (ajbrock/Neural-Photo-Editor) has no FFI: its hot path is the
 * plat-style Python class API.py:11-110, whose methods call Theano functions
 * compiled from Lasagne graphs (IAN_simple.py:56-241, IAN.py:67-228) built out
 * of layers.py's custom ops.  This header is what a binding for that path binds
 * (ctypes stub: INTEGRATION.md; the shipped host class is
 * neural_photo_editor_amd/api.py).  Every entry point names the reference
 * interface it replaces.
 *
 * Conventions
 *   - plain C types only; no torch / numpy types cross this boundary;
 *   - every buffer is caller-owned; a pointer may be host or device memory
 *     (detected with hipPointerGetAttributes); the library owns weights,
 *     activations and workspaces;
 *   - external tensor layout is the reference's: float32, C-contiguous, NCHW
 *     images in [-1,1] (API.py:80-88); NHWC is internal only;
 *   - return 0 on success, negative on error; ian_last_error() gives the text;
 *   - one handle per device, not thread-safe, all work ordered on `stream`
 *     (a hipStream_t passed as void*; NULL = the default stream).  Calls with
 *     host output pointers synchronise the stream before returning.
 */
#ifndef IAN_H_
#define IAN_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ian_handle ian_handle;

/* Fused op kinds the host lowers a Lasagne-style graph to. */
enum ian_op_kind {
  IAN_OP_CONV5S2 = 1,   /* Conv2D(DNN)Layer 5x5 stride 2 pad 2, flip_filters=False: IAN_simple.py:73-116, IAN.py:71-110 */
  IAN_OP_DECONV5S2 = 2, /* layers.py:436-483 DeconvLayer (== TransposedConv2DLayer branch IAN_simple.py:182-223) */
  IAN_OP_MDC3 = 3,      /* layers.py:207-258 MDCL: shared-W multiscale dilated 3x3 */
  IAN_OP_DENSE = 4,     /* lasagne DenseLayer: IAN_simple.py:117-135, IAN.py:114-134 */
  IAN_OP_AFFINE = 5,    /* stand-alone BatchNorm(+nonlinearity) on a tensor: layers.py:412 (bnorm0) */
  IAN_OP_MADE_IAF = 6,  /* MADE x2 + IAFLayer: IAN.py:127-128, layers.py:641-650,735-853 */
  IAN_OP_BETA = 7,      /* layers.py:397-408 beta_layer x3 + concat: IAN.py:207 */
  IAN_OP_CONCAT = 8     /* ConcatLayer on channels: IAN.py:201 */
};

/* lasagne.nonlinearities used by the configs */
enum ian_act {
  IAN_ACT_NONE = 0,
  IAN_ACT_RELU = 1,
  IAN_ACT_LRELU = 2, /* LeakyRectify(0.2) */
  IAN_ACT_ELU = 3,
  IAN_ACT_TANH = 4,
  IAN_ACT_SIGMOID = 5
};

/* Which compiled function of API.py / sample_IAN.py an op belongs to. */
enum ian_segment {
  IAN_SEG_ENC = 0, /* l_in -> l_Z_IAF (== l_Z for IAN_simple): Zfn, sample_IAN.py:90 */
  IAN_SEG_IAF = 1, /* l_Z_IAF -> l_Z: Z_IAF_fn, sample_IAN.py:93 */
  IAN_SEG_DEC = 2  /* l_Z -> l_out: X_hat_fn, API.py:46-47 */
};

#define IAN_MAX_SCALES 4

typedef struct ian_op_desc {
  int32_t kind;    /* enum ian_op_kind */
  int32_t segment; /* enum ian_segment */
  int32_t src;     /* input tensor slot */
  int32_t src2;    /* second input: residual added BEFORE the affine (ElemwiseSumLayer, layers.py:412),
                      second/third tensor for BETA / CONCAT; -1 if none */
  int32_t src3;    /* third input (BETA); -1 if none */
  int32_t dst;     /* output tensor slot */
  int32_t cin, cout;  /* channels (DENSE: in / out features) */
  int32_t in_h, in_w; /* input spatial extent (DENSE: 1,1) */
  int32_t act;        /* enum ian_act applied after bias / batch-norm */
  int32_t has_bias;   /* parameter "<name>.b" exists */
  /* DENSE only: the reference flattens (C,H,W) row-major (App. B.6) while the
     internal layout is (H,W,C); non-zero triplets make finalize permute the
     weight rows / columns so no data movement happens at run time. */
  int32_t flat_c, flat_h, flat_w;       /* input was a (C,H,W) map   */
  int32_t unflat_c, unflat_h, unflat_w; /* output is reshaped to (C,H,W): ReshapeLayer IAN_simple.py:136 */
  int32_t n_scales;                     /* MDC3: len(scales) */
  int32_t scales[IAN_MAX_SCALES];       /* MDC3: scales (0 = the 1x1 mean branch) */
  const char* name;    /* Lasagne layer name. Parameters are looked up as "<name>.W"/"<name>.b"
                          (MDC3: "<name>W", "<name>_coeff_base", "<name>_coeff_1x1", "<name>_coeff_<s>";
                          MADE_IAF: "<name>_{mu,ls}_{input,output_W,output_D}.{W,b}") -- SURVEY App. B.5 */
  const char* bn_name; /* BatchNormLayer name ("<bn>.gamma|beta|mean|inv_std") or NULL */
} ian_op_desc;

typedef struct ian_slot_desc {
  int32_t h, w, c; /* logical NHWC extent per image */
} ian_slot_desc;

typedef struct ian_model_desc {
  int32_t n_ops;
  const ian_op_desc* ops; /* topologically ordered */
  int32_t n_slots;
  const ian_slot_desc* slots;
  int32_t x_slot;      /* l_in  (n,3,64,64)                       */
  int32_t zpre_slot;   /* l_Z_IAF: encoder mean, before the IAF   */
  int32_t z_slot;      /* l_Z: what the decoder consumes (== zpre_slot when there is no IAF) */
  int32_t out_slot;    /* l_out (n,3,64,64)                       */
  int32_t num_latents; /* cfg['num_latents'] (API.py:92-96)       */
  int32_t deconv_flip; /* 1: DeconvLayer is the gradient of a true convolution */
} ian_model_desc;

/* API.py:12-21: build the model for a config (the graph arrives already lowered). */
int ian_create(const ian_model_desc* desc, ian_handle** out);
/* GANcheckpoints.py:33-57 load_weights: one call per npz entry, Theano parameter names. Host pointer. */
int ian_load_param(ian_handle* h, const char* name, const float* data, const int64_t* shape, int32_t ndim);
/* layers.py:831-853 MADE.reset("Once") result (API.py:33-36): 0/1 masks, (in,out) row-major, shared by both MADEs. */
int ian_set_made_masks(ian_handle* h, const float* m0, const float* m1, const float* md, int32_t n);
/* Ends API.py:23-36: fold batch-norm statistics, repack weights for the kernels, upload. */
int ian_finalize(ian_handle* h);

/* API.py:78-90 encode_images -> Z_hat_fn (API.py:50-51).  x f32[n,3,64,64] -> z f32[n,num_latents] */
int ian_encode(ian_handle* h, const float* x, int32_t n, float* z, void* stream);
/* API.py:98-110 sample_at -> X_hat_fn (API.py:46-47).     z f32[n,num_latents] -> x f32[n,3,64,64] */
int ian_decode(ian_handle* h, const float* z, int32_t n, float* x, void* stream);
/* sample_IAN.py:90-91 Zfn: x -> l_Z_IAF (deterministic mean) */
int ian_encode_pre_iaf(ian_handle* h, const float* x, int32_t n, float* z, void* stream);
/* sample_IAN.py:93-94 Z_IAF_fn: l_Z_IAF -> l_Z */
int ian_iaf(ian_handle* h, const float* zpre, int32_t n, float* z, void* stream);
/* encode followed by decode with the latent kept on the device (bench config 2: reconstruction). */
int ian_reconstruct(ian_handle* h, const float* x, int32_t n, float* xhat, void* stream);

/* API.py:72-76,64 imgradRGB: d mean((X_hat[0,:,r1:r2,c1:c2]-RGB[0,:,r1:r2,c1:c2])^2) / dZ.
   rgb f32[1,3,64,64], z f32[1,num_latents] -> dz f32[1,num_latents] */
int ian_grad_rgb(ian_handle* h, int32_t c1, int32_t r1, int32_t c2, int32_t r2, const float* rgb, const float* z,
                 float* dz, void* stream);
/* API.py:66-70,59 imgrad: d mean(X_hat[0,:,r1:r2,c1:c2]) / dZ */
int ian_grad_light(ian_handle* h, int32_t c1, int32_t r1, int32_t c2, int32_t r2, const float* z, float* dz,
                   void* stream);

/* ---- the steps NPE.py runs right after the hot path on every edit, chained after the decoder ---- */
/* NPE.py:110,261 (update_photo, RECON): np.uint8(from_tanh(sample_at(z))) -> out u8[n,3,64,64]; the float image stays
   on the device, 12 KB per image cross the bus.  Bare cast as in the reference: truncation toward zero, modulo 256. */
int ian_decode_u8(ian_handle* h, const float* z, int32_t n, uint8_t* out, void* stream);
/* NPE.py:218-231 (NPE.paint, photo mode), for the latent z (f32[1,num_latents]):
     DELTA = sample_at(z)[0] - to_tanh(float32(RECON))
     MASK  = scipy.ndimage.gaussian_filter(min(mean_c |DELTA|, 1), sigma)          float64, 'reflect', truncate 4
     IM    = uint8(from_tanh(to_tanh(RECON) + MASK*DELTA + (1-MASK)*ERROR))
   recon u8[3,64,64], error f32[3,64,64] (host or device; host copies are re-uploaded only when their bytes change),
   gauss_half f64[radius+1] = scipy's normalised 1-D Gaussian from the centre outwards (the host computes it exactly as
   scipy does: npe_ops.gaussian_half_kernel), radius <= 7.  im u8[3,64,64]; mask f64[64,64] or NULL.  Bit-exact with the
   numpy/scipy expression (tests/test_gpu_npe.py).  Reuses the decoder activations of the last ian_decode /
   ian_grad_* call on the same host latent. */
int ian_photo_blend(ian_handle* h, const float* z, const uint8_t* recon, const float* error, const double* gauss_half,
                    int32_t radius, uint8_t* im, double* mask, void* stream);

/* One whole brush event of NPE.paint / NPE.scroll (NPE.py:199-218, 305-316) in ONE call, one graph replay, one
   synchronisation -- instead of imgradRGB + host update + sample_at (two round trips):
     dZ    = rgb ? imgradRGB(c1,r1,c2,r2, rgb, z) : imgrad(c1,r1,c2,r2, z)        (API.py:66-76)
     z_new = z + coef * (dZ * gscale)        float32, each product rounded on its own, in this order -- NPE.py:205-209
             "grad = temp*(1+(x2-x1)); Z -= weight*grad" is coef = -weight, gscale = 1+(x2-x1); NPE.py:313-314 is
             coef = sign*weight -- bit-identical to the numpy expression on a float32 Z
     x     = sample_at(z_new)                                                       (API.py:98-110)
   z, z_new f32[num_latents] (host; may alias), dz f32[num_latents] or NULL, x f32[3,64,64] or NULL (the decoder runs either
   way: its activations stay resident for the next event), photo = NULL or the arguments of ian_photo_blend, applied to
   sample_at(z_new) in the same submission (photo mode, NPE.py:218-231).  rgb f32[1,3,64,64] host or device.
   Falls back to the composition of the public calls when the captured-graph path is not available. */
typedef struct ian_photo_args {
  const uint8_t* recon;       /* u8[3,64,64] */
  const float* error;         /* f32[3,64,64] */
  const double* gauss_half;   /* f64[radius+1] */
  int32_t radius;
  uint8_t* im;                /* out u8[3,64,64] */
  double* mask;               /* out f64[64,64] or NULL */
} ian_photo_args;
int ian_brush_step(ian_handle* h, int32_t c1, int32_t r1, int32_t c2, int32_t r2, const float* rgb, const float* z, float coef,
                   float gscale, float* z_new, float* dz, float* x, const ian_photo_args* photo, void* stream);

/* Introspection used by tests, bench.py and profiling (not part of the reference surface). */
/* Copy the activation of tensor slot `slot` from the last call, converted to NCHW, into out (host or device). */
int ian_read_slot(ian_handle* h, int32_t slot, int32_t n, float* out, void* stream);
/* Same for the gradient buffer the last ian_grad_* call left in `slot`: d loss / d (pre-epilogue value of the
   slot's producer), i.e. before batch-norm scale and activation (NCHW). */
int ian_read_slot_grad(ian_handle* h, int32_t slot, int32_t n, float* out, void* stream);
/* Name and accumulated device time (ms, HIP events on `stream`) of the dominant kernel family since the last reset. */
int ian_profile_enable(ian_handle* h, int32_t on);
int ian_profile_read(ian_handle* h, double* tapgemm_ms, int64_t* tapgemm_launches, double* tapgemm_flops,
                     double* total_ms);
/* Time candidate (tile shape, split-K) decompositions of every tapgemm layer for batch n on this device and keep
   the fastest.  what: bit 0 = forward ops (needs one prior forward call with batch >= n), bit 1 = latent-brush
   backward chain (n must be 1, needs one prior ian_grad_* call).  Results are identical for every choice up to
   float32 summation order; without this call a static heuristic is used. */
int ian_autotune(ian_handle* h, int32_t n, int32_t what, void* stream);
/* Tuning knobs (tile shape / split-K policy); key=value, returns <0 on unknown key. */
int ian_set_option(ian_handle* h, const char* key, int32_t value);

/* Box fingerprint for the bench line (no reference counterpart: the reference has no measurement code; it makes driver numbers
   from different boxes comparable).  Runs `launches` back-to-back launches of a register-only v_mfma_f32_32x32x2_f32 loop of
   `iters` x 4 MFMAs per wave (2 workgroups of 4 waves on each of the 256 CUs, non-zero operands, no memory traffic) on the
   current device's `stream` and returns the sustained fp32 matrix rate in TFLOP/s and the mean launch duration in microseconds.
   iters = 900 gives launches of ~200 us (one IAN_simple batch-64 layer), 3200 ~700 us.  Needs no handle.  0 = ok, <0 = HIP error. */
int ian_box_probe(int32_t iters, int32_t launches, double* tflops, double* us_per_launch, void* stream);

const char* ian_last_error(ian_handle* h);
const char* ian_version(void);
void ian_destroy(ian_handle* h);

#ifdef __cplusplus
}
#endif
#endif /* IAN_H_ */

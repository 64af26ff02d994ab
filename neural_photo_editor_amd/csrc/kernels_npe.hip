// The 64x64 steps right after the decoder in every NPE edit, on the device:
//   photo_blend_kernel  NPE.paint's photo-mode blend, NPE.py:218-231
//       DELTA = X_hat - to_tanh(float32(RECON));  MASK = gaussian_filter(min(mean_c|DELTA|, 1), 0.7)
//       IM    = uint8(from_tanh(to_tanh(RECON) + MASK*DELTA + (1-MASK)*ERROR))
//   to_uint8_kernel     uint8(from_tanh(X_hat)), NPE.py:110,261 (update_photo / RECON)
// so that an edit needs ONE 12 KB device->host copy instead of a 48 KB float image plus host numpy/scipy passes.
//
// Byte work is held to bit-exactness against the reference expression (oracle: npe_ops.photo_blend_host): every operation
// is performed in the precision numpy uses for it (float32 for DELTA and its channel mean, float64 from the np.min(...)
// on), in numpy's order, with explicitly rounded intrinsics so that hipcc cannot contract a*b+c into an fma, and the
// Gaussian follows scipy.ndimage's separable 'reflect' correlate1d summation order for a symmetric kernel
// (t = x[l]*w0; t += (x[l-j] + x[l+j])*w[j] for j = R..1; axis 0 first), with the weights computed by the host exactly
// as scipy computes them.  The bare np.uint8 cast (no clipping, NPE.py:231) is truncation toward zero modulo 256.
#include "ian_internal.h"

// numpy and scipy round every operation, while hipcc's default -ffp-contract=fast fuses a*b+c into ONE rounding
// (measured on MI355X: 1-ulp differences from scipy in 31 % of the mask values).  HIP's __dadd_rn / __dmul_rn / __fadd_rn
// do not help: on AMD targets they are inline plain operators defined under the default mode, and fuse after inlining.
// So: plain operators, with contraction switched off for everything below this line (checked in the ISA: separate
// v_mul_f64 / v_add_f64; the only fmas left are inside the correctly rounded division expansions).
#pragma clang fp contract(off)

namespace ian {

__device__ __forceinline__ unsigned char np_uint8(double v) {  // numpy float64 -> uint8 on x86-64: cvttsd2si, low byte
  const int i = (int)v;   // truncation toward zero; |v| stays far below 2^31 here
  return (unsigned char)(i & 0xFF);
}
__device__ __forceinline__ unsigned char np_uint8f(float v) {
  const int i = (int)v;
  return (unsigned char)(i & 0xFF);
}
__device__ __forceinline__ int reflect_idx(int i, int n) {  // d c b a | a b c d | d c b a
  if (i < 0) i = -i - 1;
  if (i >= n) i = 2 * n - 1 - i;
  return i;
}

constexpr int PB_T = 1024;
__global__ __launch_bounds__(PB_T) void photo_blend_kernel(PhotoBlendArgs a) {
  __shared__ double m0[64 * 64];
  __shared__ double m1[64 * 64];
  constexpr int H = 64, W = 64, HW = H * W;
  const int tid = threadIdx.x;
  // ---- min(mean_c |DELTA|, 1): float32 until np.min promotes to float64
  for (int p = tid; p < HW; p += PB_T) {
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float r = (float)a.recon[c * HW + p];
      const float tt = (2.0f * (r / 255.0f)) - 1.0f;      // to_tanh(np.float32(RECON))
      const float d = fabsf(a.xhat[c * HW + p] - tt);
      s = (c == 0) ? d : s + d;                                           // add.reduce over axis 0
    }
    const float mean = s / 3.0f;
    const double v = (double)mean;
    m0[p] = v < 1.0 ? v : 1.0;
  }
  __syncthreads();
  // ---- separable Gaussian, axis 0 (rows) then axis 1 (columns)
  const int R = a.radius;
  for (int p = tid; p < HW; p += PB_T) {
    const int y = p / W, x = p % W;
    double t = m0[p] * a.w[0];
    for (int j = R; j >= 1; --j)
      t = t + (m0[reflect_idx(y - j, H) * W + x] + m0[reflect_idx(y + j, H) * W + x]) * a.w[j];
    m1[p] = t;
  }
  __syncthreads();
  for (int p = tid; p < HW; p += PB_T) {
    const int y = p / W, x = p % W;
    double t = m1[p] * a.w[0];
    for (int j = R; j >= 1; --j)
      t = t + (m1[y * W + reflect_idx(x - j, W)] + m1[y * W + reflect_idx(x + j, W)]) * a.w[j];
    m0[p] = t;   // every thread rewrites only the pixels it read in the FIRST pass and nobody reads m0 in this pass
  }
  __syncthreads();
  // ---- IM = uint8(from_tanh(to_tanh(RECON) + MASK*DELTA + (1-MASK)*ERROR)), float64
  for (int p = tid; p < HW; p += PB_T) {
    const double mask = m0[p];
    if (a.mask) a.mask[p] = mask;
    const double om = 1.0 - mask;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const unsigned char rb = a.recon[c * HW + p];
      const float r = (float)rb;
      const float tt = (2.0f * (r / 255.0f)) - 1.0f;
      const float delta = a.xhat[c * HW + p] - tt;
      const double D = mask * (double)delta + om * (double)a.error[c * HW + p];
      const double t64 = 2.0 * ((double)rb / 255.0) - 1.0;   // to_tanh(RECON): uint8 -> float64
      const double v = 255.0 * ((t64 + D) + 1.0) / 2.0;
      a.im[c * HW + p] = np_uint8(v);
    }
  }
}
hipError_t launch_photo_blend(const PhotoBlendArgs& a, hipStream_t s) {
  if (a.radius < 0 || a.radius > 7) return hipErrorInvalidValue;
  hipLaunchKernelGGL(photo_blend_kernel, dim3(1), dim3(PB_T), 0, s, a);
  return hipGetLastError();
}

// uint8(from_tanh(x)) on float32: 255.0*(x+1)/2.0 (NPE.py:40-41), truncation modulo 256
__global__ __launch_bounds__(256) void to_uint8_kernel(const float* __restrict__ x, unsigned char* __restrict__ y, long long n) {
  const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i + 3 < n) {
    const float4 v = *reinterpret_cast<const float4*>(x + i);
    uchar4 o;
    o.x = np_uint8f(255.0f * (v.x + 1.0f) / 2.0f);
    o.y = np_uint8f(255.0f * (v.y + 1.0f) / 2.0f);
    o.z = np_uint8f(255.0f * (v.z + 1.0f) / 2.0f);
    o.w = np_uint8f(255.0f * (v.w + 1.0f) / 2.0f);
    *reinterpret_cast<uchar4*>(y + i) = o;
  } else {
    for (long long k = i; k < n; ++k) y[k] = np_uint8f(255.0f * (x[k] + 1.0f) / 2.0f);
  }
}
// Z += coef * (dZ * gscale) in float32, every product rounded on its own (NPE.py:205-209: grad = temp*(1+(x2-x1));
// Z -= weight*grad  ->  coef = -weight;  NPE.py:313-314 likewise with coef = sign*weight).  cg = {coef, gscale} in device memory
// so that a captured graph can be replayed with other values.
// z_mirror / g_mirror (or nullptr): the new latent and the gradient are also written there -- the pinned host block of the
// interactive loop (zero-copy), which spares the captured graph two device -> host copy nodes; cg may live there as well.
__global__ __launch_bounds__(128) void latent_update_kernel(float* __restrict__ z, const float* __restrict__ g,
                                                            const float* __restrict__ cg, int n, float* __restrict__ z_mirror,
                                                            float* __restrict__ g_mirror) {
  const int i = threadIdx.x;
  if (i < n) {
    const float gi = g[i];
    float t = gi * cg[1];
    t = cg[0] * t;
    const float zn = z[i] + t;
    z[i] = zn;
    if (z_mirror) z_mirror[i] = zn;
    if (g_mirror) g_mirror[i] = gi;
  }
}
hipError_t launch_latent_update(float* z, const float* g, const float* cg, int n, float* z_mirror, float* g_mirror, hipStream_t s) {
  if (n > 128) return hipErrorInvalidValue;
  hipLaunchKernelGGL(latent_update_kernel, dim3(1), dim3(128), 0, s, z, g, cg, n, z_mirror, g_mirror);
  return hipGetLastError();
}
hipError_t launch_to_uint8(const float* x, unsigned char* y, long long n, hipStream_t s) {
  hipLaunchKernelGGL(to_uint8_kernel, dim3((unsigned)((n / 4 + 255) / 256 + 1)), dim3(256), 0, s, x, y, n);
  return hipGetLastError();
}


// ---- keep-warm (option edit_keep_warm_us, an EXPERIMENT, off by default) ---------------------------------------------------------
// One wave that keeps the interactive stream's queue busy between two brush events: it polls a flag in the mapped pinned block
// (the next event sets it just before its graph is launched) and gives up after max_ticks of the 100 MHz wall clock (s_memrealtime).  What it is for:
// the first kernel of every graph replay costs ~20 us whatever it does (profiles/r05_batch1_chains.md) -- is that the wake-up of
// an idle queue?  The bench line's edit_step.keep_warm compares the event latency with and without it.
__global__ __launch_bounds__(64) void keep_warm_kernel(const volatile int* flag, long long max_ticks) {
  if (threadIdx.x != 0) return;
  const long long t0 = (long long)wall_clock64();
  while (*flag == 0 && (long long)wall_clock64() - t0 < max_ticks) __builtin_amdgcn_s_sleep(8);
}
hipError_t launch_keep_warm(const int* flag, long long max_ticks, hipStream_t s) {
  hipLaunchKernelGGL(keep_warm_kernel, dim3(1), dim3(64), 0, s, flag, max_ticks);
  return hipGetLastError();
}

}  // namespace ian

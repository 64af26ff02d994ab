// Edge layers (3 image channels) and small elementwise ops of the IAN hot path.
// These are the HBM/latency-bound pieces (SURVEY "hard parts": Cout=3 / Cin=3 layers waste MFMA tiles),
// written as VALU kernels with coalesced NHWC accesses; the NCHW<->NHWC boundary conversion of the
// reference's external layout (API.py:80-88) is folded into them so no separate transpose pass exists.
#include "ian_internal.h"

namespace ian {

__device__ __forceinline__ float m_act(float v, int act) {
  switch (act) {
    case 1: return v > 0.f ? v : 0.f;
    case 2: return v > 0.f ? v : 0.2f * v;
    case 3: return v > 0.f ? v : expm1f(v);
    case 4: return tanhf(v);
    case 5: return 1.f / (1.f + __expf(-v));
    default: return v;
  }
}
__device__ __forceinline__ float m_dact(float y, int act) {
  switch (act) {
    case 1: return y > 0.f ? 1.f : 0.f;
    case 2: return y > 0.f ? 1.f : 0.2f;
    case 3: return y > 0.f ? 1.f : y + 1.f;
    case 4: return 1.f - y * y;
    case 5: return y * (1.f - y);
    default: return 1.f;
  }
}

// ------------------------------------------------------------------------------------------------
// enc_conv1 (IAN_simple.py:73-83): x NCHW [n,3,H,W] -> y NHWC [n,H/2,W/2,Cout]; 5x5 s2 p2 correlation.
// Block = 2 output rows x OW pixels of one image; lane = output channel (coalesced NHWC stores), the
// 75 filter taps of that channel live in registers, the input patch is broadcast from LDS.
// ------------------------------------------------------------------------------------------------
constexpr int C1_ROWS = 2;
template <int COUT>
__global__ __launch_bounds__(256) void conv1_nchw_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ scale,
                                                         const float* __restrict__ shift, float* __restrict__ y,
                                                         int H, int W, int act) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int OW = W >> 1, OH = H >> 1;
  const int PW = W + 4;                    // padded row width
  constexpr int PR = 2 * C1_ROWS + 3;      // patch rows
  const int n = blockIdx.y, oy0 = blockIdx.x * C1_ROWS;
  // stage the 3 x PR x PW patch (zero padded)
  for (int i = threadIdx.x; i < 3 * PR * PW; i += 256) {
    const int c = i / (PR * PW), r = (i / PW) % PR, col = i % PW;
    const int iy = 2 * oy0 - 2 + r, ix = col - 2;
    float v = 0.f;
    if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) v = x[((size_t)(n * 3 + c) * H + iy) * W + ix];
    sm[i] = v;
  }
  __syncthreads();
  constexpr int GROUPS = 256 / COUT;  // pixel groups per block
  const int co = threadIdx.x % COUT, grp = threadIdx.x / COUT;
  float wr[75];
#pragma unroll
  for (int k = 0; k < 75; ++k) wr[k] = w[k * COUT + co];
  const float sc = scale ? scale[co] : 1.f, sh = shift ? shift[co] : 0.f;
  const int npix = C1_ROWS * OW;
  for (int pidx = grp; pidx < npix; pidx += GROUPS) {
    const int r = pidx / OW, ox = pidx % OW;
    if (oy0 + r >= OH) break;
    float acc = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int ky = 0; ky < 5; ++ky) {
        const float* row = sm + (c * PR + 2 * r + ky) * PW + 2 * ox;
#pragma unroll
        for (int kx = 0; kx < 5; ++kx) acc = fmaf(row[kx], wr[(c * 5 + ky) * 5 + kx], acc);
      }
    y[((size_t)(n * OH + oy0 + r) * OW + ox) * COUT + co] = m_act(acc * sc + sh, act);
  }
}

hipError_t launch_conv1_nchw(const float* x, const float* w, const float* scale, const float* shift, float* y, int n,
                             int H, int W, int Cout, int act, hipStream_t s) {
  const int OH = H / 2;
  const size_t lds = (size_t)3 * (2 * C1_ROWS + 3) * (W + 4) * sizeof(float);
  dim3 grid((OH + C1_ROWS - 1) / C1_ROWS, n);
  if (Cout == 128)
    hipLaunchKernelGGL(conv1_nchw_kernel<128>, grid, dim3(256), lds, s, x, w, scale, shift, y, H, W, act);
  else if (Cout == 64)
    hipLaunchKernelGGL(conv1_nchw_kernel<64>, grid, dim3(256), lds, s, x, w, scale, shift, y, H, W, act);
  else if (Cout == 256)
    hipLaunchKernelGGL(conv1_nchw_kernel<256>, grid, dim3(256), lds, s, x, w, scale, shift, y, H, W, act);
  else
    return hipErrorInvalidValue;
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// dec_out (IAN_simple.py:171-181): x NHWC [n,H,W,Cin] -> y NCHW [n,Cout<=4,2H,2W], 5x5 s2 transposed conv.
// Block = 16x16 output tile = 8x8 input pixels (+1 halo) staged in LDS with a Cin+4 pixel stride
// (conflict-free ds_read_b128 across pixels); wave = output parity class (uniform tap list 9/6/6/4,
// weights arrive through scalar loads), lane = position in the 8x8 grid.
// w packed [ky*5+kx][4][Cin] already holding the (possibly flipped) reference filter.
// ------------------------------------------------------------------------------------------------
template <int CIN>
__global__ __launch_bounds__(256) void deconv_out_nchw_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                              const float* __restrict__ scale,
                                                              const float* __restrict__ shift, float* __restrict__ y,
                                                              int H, int W, int Cout, int act) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  constexpr int PS = CIN + 4;  // padded pixel stride
  const int tiles_x = W >> 3;
  const int n = blockIdx.y, ty = blockIdx.x / tiles_x, tx = blockIdx.x % tiles_x;
  const int iy0 = ty * 8 - 1, ix0 = tx * 8 - 1;
  // stage 10x10 input pixels x CIN
  for (int i = threadIdx.x; i < 100 * (CIN / 4); i += 256) {
    const int pix = i / (CIN / 4), c4 = (i % (CIN / 4)) * 4;
    const int iy = iy0 + pix / 10, ix = ix0 + pix % 10;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W)
      v = *reinterpret_cast<const float4*>(x + ((size_t)(n * H + iy) * W + ix) * CIN + c4);
    *reinterpret_cast<float4*>(sm + pix * PS + c4) = v;
  }
  __syncthreads();
  const int cls = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int py = cls >> 1, px = cls & 1;
  const int lane = threadIdx.x & 63;
  const int qy = lane >> 3, qx = lane & 7;
  float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
  // oy = 2*iy - 2 + ky  =>  ky = py + 2 - 2*dy with iy = qy + dy ; even: dy in {1,0,-1}, odd: dy in {1,0}
  for (int ky = py; ky < 5; ky += 2) {
    const int dy = (py + 2 - ky) / 2;
    for (int kx = px; kx < 5; kx += 2) {
      const int dx = (px + 2 - kx) / 2;
      const float* xs = sm + ((qy + dy + 1) * 10 + (qx + dx + 1)) * PS;
      const float* wt = w + (size_t)(ky * 5 + kx) * 4 * CIN;
#pragma unroll 8
      for (int c = 0; c < CIN; c += 4) {
        const float4 xv = *reinterpret_cast<const float4*>(xs + c);
        const float4 w0 = *reinterpret_cast<const float4*>(wt + c);
        const float4 w1 = *reinterpret_cast<const float4*>(wt + CIN + c);
        const float4 w2 = *reinterpret_cast<const float4*>(wt + 2 * CIN + c);
        const float4 w3 = *reinterpret_cast<const float4*>(wt + 3 * CIN + c);
        acc0 = fmaf(xv.x, w0.x, acc0); acc0 = fmaf(xv.y, w0.y, acc0); acc0 = fmaf(xv.z, w0.z, acc0); acc0 = fmaf(xv.w, w0.w, acc0);
        acc1 = fmaf(xv.x, w1.x, acc1); acc1 = fmaf(xv.y, w1.y, acc1); acc1 = fmaf(xv.z, w1.z, acc1); acc1 = fmaf(xv.w, w1.w, acc1);
        acc2 = fmaf(xv.x, w2.x, acc2); acc2 = fmaf(xv.y, w2.y, acc2); acc2 = fmaf(xv.z, w2.z, acc2); acc2 = fmaf(xv.w, w2.w, acc2);
        acc3 = fmaf(xv.x, w3.x, acc3); acc3 = fmaf(xv.y, w3.y, acc3); acc3 = fmaf(xv.z, w3.z, acc3); acc3 = fmaf(xv.w, w3.w, acc3);
      }
    }
  }
  const int OH = 2 * H, OW = 2 * W;
  const int oy = 2 * (ty * 8 + qy) + py, ox = 2 * (tx * 8 + qx) + px;
  const float a[4] = {acc0, acc1, acc2, acc3};
  for (int co = 0; co < Cout; ++co) {
    const float sc = scale ? scale[co] : 1.f, sh = shift ? shift[co] : 0.f;
    y[((size_t)(n * Cout + co) * OH + oy) * OW + ox] = m_act(a[co] * sc + sh, act);
  }
}

hipError_t launch_deconv_out_nchw(const float* x, const float* w, const float* scale, const float* shift, float* y,
                                  int n, int H, int W, int Cin, int Cout, int act, hipStream_t s) {
  if (Cout > 4 || (H & 7) || (W & 7)) return hipErrorInvalidValue;
  dim3 grid((H / 8) * (W / 8), n);
  const size_t lds = (size_t)100 * (Cin + 4) * sizeof(float);
  if (Cin == 128)
    hipLaunchKernelGGL(deconv_out_nchw_kernel<128>, grid, dim3(256), lds, s, x, w, scale, shift, y, H, W, Cout, act);
  else if (Cin == 64)
    hipLaunchKernelGGL(deconv_out_nchw_kernel<64>, grid, dim3(256), lds, s, x, w, scale, shift, y, H, W, Cout, act);
  else if (Cin == 256)
    hipLaunchKernelGGL(deconv_out_nchw_kernel<256>, grid, dim3(256), lds, s, x, w, scale, shift, y, H, W, Cout, act);
  else
    return hipErrorInvalidValue;
  return hipGetLastError();
}

// backward-data of dec_out for the latent brush (API.py:59,64):
//   dacc[n,iy,ix,ci] = ( sum_{ky,kx,co} g[n,co,2iy-2+ky,2ix-2+kx] * w[ky*5+kx][co][ci] ) * act'(yfwd) * scale[ci]
// g is the (patch-sparse) NCHW gradient wrt the pre-activation of dec_out; one thread per (pixel, ci).
__global__ __launch_bounds__(256) void deconv_out_bwd_kernel(const float* __restrict__ g, const float* __restrict__ w,
                                                             float* __restrict__ dx, const float* __restrict__ yfwd,
                                                             const float* __restrict__ scale, int n, int H, int W,
                                                             int Cin, int Cout, int act) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t total = (size_t)n * H * W * Cin;
  if (idx >= total) return;
  const int ci = idx % Cin;
  const size_t pix = idx / Cin;
  const int ix = pix % W, iy = (pix / W) % H, b = pix / ((size_t)W * H);
  const int OH = 2 * H, OW = 2 * W;
  float acc = 0.f;
  for (int ky = 0; ky < 5; ++ky) {
    const int oy = 2 * iy - 2 + ky;
    if ((unsigned)oy >= (unsigned)OH) continue;
    for (int kx = 0; kx < 5; ++kx) {
      const int ox = 2 * ix - 2 + kx;
      if ((unsigned)ox >= (unsigned)OW) continue;
      for (int co = 0; co < Cout; ++co) {
        const float gv = g[((size_t)(b * Cout + co) * OH + oy) * OW + ox];
        acc = fmaf(gv, w[((size_t)(ky * 5 + kx) * 4 + co) * Cin + ci], acc);
      }
    }
  }
  const float d = yfwd ? m_dact(yfwd[idx], act) : 1.f;
  dx[idx] = acc * d * (scale ? scale[ci] : 1.f);
}

hipError_t launch_deconv_out_bwd(const float* g, const float* w, float* dx, const float* yfwd, const float* scale,
                                 int n, int H, int W, int Cin, int Cout, int act, hipStream_t s) {
  const size_t total = (size_t)n * H * W * Cin;
  hipLaunchKernelGGL(deconv_out_bwd_kernel, dim3((total + 255) / 256), dim3(256), 0, s, g, w, dx, yfwd, scale, n, H,
                     W, Cin, Cout, act);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// elementwise helpers
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void affine_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                     const float* __restrict__ scale, const float* __restrict__ shift,
                                                     long long npix, int C, int stride, int act) {
  const int c4n = C >> 2;
  const long long total = npix * c4n;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long pix = i / c4n;
    const int c = (int)(i % c4n) * 4;
    const float4 v = *reinterpret_cast<const float4*>(x + pix * stride + c);
    const float4 sc = scale ? *reinterpret_cast<const float4*>(scale + c) : make_float4(1.f, 1.f, 1.f, 1.f);
    const float4 sh = shift ? *reinterpret_cast<const float4*>(shift + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 o;
    o.x = m_act(v.x * sc.x + sh.x, act);
    o.y = m_act(v.y * sc.y + sh.y, act);
    o.z = m_act(v.z * sc.z + sh.z, act);
    o.w = m_act(v.w * sc.w + sh.w, act);
    *reinterpret_cast<float4*>(y + pix * stride + c) = o;
  }
}
hipError_t launch_affine(const float* x, float* y, const float* scale, const float* shift, long long npix, int C,
                         int stride, int act, hipStream_t s) {
  if (C & 3) return hipErrorInvalidValue;
  long long total = npix * (C >> 2);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(affine_kernel, dim3(blocks), dim3(256), 0, s, x, y, scale, shift, npix, C, stride, act);
  return hipGetLastError();
}

// MADE x2 + IAF (IAN.py:127-128; layers.py:641-650, 735-853).  One block (128 threads) per latent row.
// wts: 6 pre-masked [d][d] matrices (row = input) in the order mu_input, mu_output_W, mu_output_D,
// ls_input, ls_output_W, ls_output_D; bias: 6 x [d].
__global__ __launch_bounds__(128) void made_iaf_kernel(const float* __restrict__ z, float* __restrict__ zo,
                                                       const float* __restrict__ wts, const float* __restrict__ bias,
                                                       int d, int zs) {
  __shared__ float zin[128], hm[128], hl[128];
  const int t = threadIdx.x, row = blockIdx.x;
  zin[t] = (t < d) ? z[(size_t)row * zs + t] : 0.f;
  __syncthreads();
  const int dd = d * d;
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
  if (t < d) {
    float om = bias[1 * d + t], dm = bias[2 * d + t];
    float ol = bias[4 * d + t], dl = bias[5 * d + t];
    for (int i = 0; i < d; ++i) {
      om = fmaf(hm[i], wts[1 * dd + i * d + t], om);
      dm = fmaf(zin[i], wts[2 * dd + i * d + t], dm);
      ol = fmaf(hl[i], wts[4 * dd + i * d + t], ol);
      dl = fmaf(zin[i], wts[5 * dd + i * d + t], dl);
    }
    const float mu = om + dm, ls = ol + dl;
    zo[(size_t)row * zs + t] = (zin[t] - mu) / expf(ls);
  }
}
hipError_t launch_made_iaf(const float* z, float* zo, const float* wts, const float* bias, int n, int d, int zs,
                           hipStream_t s) {
  if (d > 128) return hipErrorInvalidValue;
  hipLaunchKernelGGL(made_iaf_kernel, dim3(n), dim3(128), 0, s, z, zo, wts, bias, d, zs);
  return hipGetLastError();
}

// beta_layer (layers.py:397-408) x3 + ConcatLayer (IAN.py:207): NHWC 2-channel maps -> NCHW [n,3,hw]
__global__ __launch_bounds__(256) void beta_kernel(const float* __restrict__ R, const float* __restrict__ G,
                                                   const float* __restrict__ B, float* __restrict__ y, int n, int hw,
                                                   int rs) {
  const long long total = (long long)n * hw;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int b = (int)(i / hw), p = (int)(i % hw);
  const float* src[3] = {R, G, B};
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float a = src[c][i * rs + 0], bb = src[c][i * rs + 1];
    y[((size_t)b * 3 + c) * hw + p] = 2.f * (a / (a + bb + 1e-8f)) - 1.f;
  }
}
hipError_t launch_beta(const float* R, const float* G, const float* B, float* y, int n, int hw, int rs,
                       hipStream_t s) {
  const long long total = (long long)n * hw;
  hipLaunchKernelGGL(beta_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, R, G, B, y, n, hw, rs);
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void concat2_kernel(const float* __restrict__ a, int ca, int sa,
                                                      const float* __restrict__ b, int cb, int sb,
                                                      float* __restrict__ y, int sy, long long npix) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= npix) return;
  for (int c = 0; c < ca; ++c) y[i * sy + c] = a[i * sa + c];
  for (int c = 0; c < cb; ++c) y[i * sy + ca + c] = b[i * sb + c];
}
hipError_t launch_concat2(const float* a, int ca, int sa, const float* b, int cb, int sb, float* y, int sy,
                          long long npix, hipStream_t s) {
  hipLaunchKernelGGL(concat2_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, s, a, ca, sa, b, cb, sb, y,
                     sy, npix);
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void rows_copy_kernel(const float* __restrict__ src, int ss, float* __restrict__ dst,
                                                        int ds, int n, int c) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n * c) return;
  const int r = i / c, k = i % c;
  dst[(size_t)r * ds + k] = src[(size_t)r * ss + k];
}
hipError_t launch_rows_copy(const float* src, int src_stride, float* dst, int dst_stride, int n, int c,
                            hipStream_t s) {
  hipLaunchKernelGGL(rows_copy_kernel, dim3((n * c + 255) / 256), dim3(256), 0, s, src, src_stride, dst, dst_stride, n,
                     c);
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const float* __restrict__ src, int stride,
                                                           float* __restrict__ dst, int n, int hw, int c) {
  const long long total = (long long)n * hw * c;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int p = (int)(i % hw);
    const int ch = (int)((i / hw) % c);
    const int b = (int)(i / ((long long)hw * c));
    dst[i] = src[((size_t)b * hw + p) * stride + ch];
  }
}
hipError_t launch_nhwc_to_nchw(const float* src, int stride, float* dst, int n, int hw, int c, hipStream_t s) {
  long long total = (long long)n * hw * c;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(blocks), dim3(256), 0, s, src, stride, dst, n, hw, c);
  return hipGetLastError();
}

// d loss / d x_hat for the two brush losses, NCHW [1,3,H,W], zero outside the patch.
//  mode 0 (API.py:59): loss = mean(x_hat[0,:,r1:r2,c1:c2])            -> g = 1/N
//  mode 1 (API.py:64): loss = mean((rgb - x_hat)^2 over the patch)     -> g = 2*(x_hat-rgb)/N
__global__ __launch_bounds__(256) void patch_seed_kernel(const float* __restrict__ xhat, const float* __restrict__ rgb,
                                                         float* __restrict__ g, int H, int W, int c1, int r1, int c2,
                                                         int r2, int mode) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= 3 * H * W) return;
  const int xx = i % W, yy = (i / W) % H;
  const int cnt = 3 * (r2 - r1) * (c2 - c1);
  float v = 0.f;
  if (yy >= r1 && yy < r2 && xx >= c1 && xx < c2 && cnt > 0) {
    const float inv = 1.f / (float)cnt;
    v = (mode == 0) ? inv : 2.f * (xhat[i] - rgb[i]) * inv;
  }
  g[i] = v;
}
hipError_t launch_patch_seed(const float* xhat, const float* rgb, float* g, int H, int W, int c1, int r1, int c2,
                             int r2, int mode, hipStream_t s) {
  hipLaunchKernelGGL(patch_seed_kernel, dim3((3 * H * W + 255) / 256), dim3(256), 0, s, xhat, rgb, g, H, W, c1, r1, c2,
                     r2, mode);
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void dact_nchw_kernel(float* __restrict__ g, const float* __restrict__ y,
                                                        const float* __restrict__ scale, int n, int c, int hw,
                                                        int act) {
  const long long total = (long long)n * c * hw;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int ch = (int)((i / hw) % c);
  g[i] = g[i] * m_dact(y[i], act) * (scale ? scale[ch] : 1.f);
}
hipError_t launch_dact_nchw(float* g, const float* y, const float* scale, int n, int c, int hw, int act,
                            hipStream_t s) {
  const long long total = (long long)n * c * hw;
  hipLaunchKernelGGL(dact_nchw_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, g, y, scale, n, c, hw,
                     act);
  return hipGetLastError();
}

}  // namespace ian

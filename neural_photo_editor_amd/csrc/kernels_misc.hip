// Edge layers (3 image channels) and small elementwise ops of the IAN hot path.
// These are the HBM/latency-bound pieces (SURVEY "hard parts": Cout=3 / Cin=3 layers waste MFMA tiles),
// written as VALU kernels with coalesced NHWC accesses; the NCHW<->NHWC boundary conversion of the
// reference's external layout (API.py:80-88) is folded into them so no separate transpose pass exists.
#include <algorithm>

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
// Block = 2 output rows x OW pixels of one image.  A lane owns CPL output channels (lane, lane+64, ...: coalesced
// NHWC stores) whose 75 filter taps live in registers; the input patch value of a tap is the same for the whole
// wave and is broadcast from LDS, so every LDS read feeds CPL FMAs (the kernel is LDS-issue bound at CPL = 1).
// ------------------------------------------------------------------------------------------------
constexpr int C1_ROWS = 2;
template <int COUT, int CPL>
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
  constexpr int LPP = COUT / CPL;     // lanes per pixel
  constexpr int GROUPS = 256 / LPP;   // pixels in flight per block
  const int co0 = threadIdx.x % LPP, grp = threadIdx.x / LPP;
  float wr[CPL][75];
#pragma unroll
  for (int j = 0; j < CPL; ++j)
#pragma unroll
    for (int k = 0; k < 75; ++k) wr[j][k] = w[k * COUT + co0 + j * LPP];
  float sc[CPL], sh[CPL];
#pragma unroll
  for (int j = 0; j < CPL; ++j) {
    sc[j] = scale ? scale[co0 + j * LPP] : 1.f;
    sh[j] = shift ? shift[co0 + j * LPP] : 0.f;
  }
  const int npix = C1_ROWS * OW;
  for (int pidx = grp; pidx < npix; pidx += GROUPS) {
    const int r = pidx / OW, ox = pidx % OW;
    if (oy0 + r >= OH) break;
    float acc[CPL];
#pragma unroll
    for (int j = 0; j < CPL; ++j) acc[j] = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int ky = 0; ky < 5; ++ky) {
        const float* row = sm + (c * PR + 2 * r + ky) * PW + 2 * ox;
#pragma unroll
        for (int kx = 0; kx < 5; ++kx) {
          const float xv = row[kx];
#pragma unroll
          for (int j = 0; j < CPL; ++j) acc[j] = fmaf(xv, wr[j][(c * 5 + ky) * 5 + kx], acc[j]);
        }
      }
    float* dst = y + ((size_t)(n * OH + oy0 + r) * OW + ox) * COUT + co0;
#pragma unroll
    for (int j = 0; j < CPL; ++j) dst[j * LPP] = m_act(acc[j] * sc[j] + sh[j], act);
  }
}

// MFMA form of enc_conv1 for 64-pixel-wide images: the 75-deep contraction (3 channels x 5x5 taps, padded to 76) of
// 32 output pixels x 32 channels per tile runs on v_mfma_f32_32x32x2_f32 (exact fp32).  The im2col operand never
// exists: a lane reads its patch element straight from the zero-padded LDS image with a compile-time offset per k;
// the wave's 76x32 filter block lives in 38 registers and serves C1M_ROWS output rows.  The staged rows are stored
// de-interleaved by column parity (even columns, then odd columns): the stride-2 gather "column 2*ox + kx" of the 32
// lanes becomes a contiguous read (round 1 measured 47.6 % bank-conflict cycles on the interleaved image).
// Round 5 measured a pipelined variant of the kernel below (filter block requested before the image rows, two row pairs per
// workgroup with the next pair's rows in flight under the MFMAs, patch operands read from LDS a group of 8 k-pairs ahead):
// 24.7 us against 21.8-22.2 us for this form at batch 64 (gpurun_out/r05b traces) -- 199 VGPRs halve the resident waves and the
// lock-step prologue is not what bounds it.  The layer's floor is 8.0 us of exact-fp32 matrix work (1.26 GFLOP at 157 TF/s).
typedef float c1_f32x16 __attribute__((ext_vector_type(16)));
constexpr int C1M_ROWS = 2;   // 4 rows per block measured slower (34 vs 23 us at batch 64): fewer, longer latency chains
template <int COUT>
__global__ __launch_bounds__(256) void conv1_mfma_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ scale,
                                                         const float* __restrict__ shift, float* __restrict__ y,
                                                         int H, int act) {
  constexpr int W = 64, OW = 32, PW = W + 4, PH = PW / 2, PR = 2 * C1M_ROWS + 3;
  __shared__ float sm[3 * PR * PW];
  const int OH = H >> 1;
  const int n = blockIdx.y, oy0 = blockIdx.x * C1M_ROWS;
  for (int i = threadIdx.x; i < 3 * PR * PW; i += 256) {
    const int c = i / (PR * PW), r = (i / PW) % PR, col = i % PW;
    const int iy = 2 * oy0 - 2 + r, ix = col - 2;
    float v = 0.f;
    if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) v = x[((size_t)(n * 3 + c) * H + iy) * W + ix];
    sm[(c * PR + r) * PW + (col & 1) * PH + (col >> 1)] = v;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  for (int cg = wave; cg < COUT / 32; cg += 4) {
    float wr[38];
#pragma unroll
    for (int kk = 0; kk < 38; ++kk) {
      const int k = 2 * kk + half;
      wr[kk] = (k < 75) ? w[k * COUT + cg * 32 + l31] : 0.f;
    }
    __syncthreads();  // (first pass) the staged image is visible
    c1_f32x16 acc[C1M_ROWS];
#pragma unroll
    for (int i = 0; i < C1M_ROWS; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
#pragma unroll
    for (int kk = 0; kk < 38; ++kk) {
      // k -> (c, ky, kx) -> offset in the padded image; k = 75 is padding (zero weight, any valid address).
      // patch column 2*ox + kx sits at (kx & 1) * PH + (kx >> 1) + ox of its de-interleaved row
      const int k0 = 2 * kk, k1 = (2 * kk + 1 < 75) ? 2 * kk + 1 : 74;
      const int o0 = ((k0 / 25) * PR + (k0 % 25) / 5) * PW + ((k0 % 5) & 1) * PH + ((k0 % 5) >> 1);
      const int o1 = ((k1 / 25) * PR + (k1 % 25) / 5) * PW + ((k1 % 5) & 1) * PH + ((k1 % 5) >> 1);
      const int o = half ? o1 : o0;
#pragma unroll
      for (int i = 0; i < C1M_ROWS; ++i) {
        const float a = sm[o + (2 * i) * PW + l31];
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, wr[kk], acc[i], 0, 0, 0);
      }
    }
    // C layout: col = lane&31 (channel), row = (r&3) + 8*(r>>2) + 4*half (pixel ox)
    const int c = cg * 32 + l31;
    const float sc = scale ? scale[c] : 1.f, sh = shift ? shift[c] : 0.f;
#pragma unroll
    for (int i = 0; i < C1M_ROWS; ++i) {
      if (oy0 + i >= OH) break;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ox = (r & 3) + 8 * (r >> 2) + 4 * half;
        y[((size_t)(n * OH + oy0 + i) * OW + ox) * COUT + c] = m_act(acc[i][r] * sc + sh, act);
      }
    }
  }
}

hipError_t launch_conv1_nchw(const float* x, const float* w, const float* scale, const float* shift, float* y, int n,
                             int H, int W, int Cout, int act, hipStream_t s) {
  const int OH = H / 2;
  const size_t lds = (size_t)3 * (2 * C1_ROWS + 3) * (W + 4) * sizeof(float);
  dim3 grid((OH + C1_ROWS - 1) / C1_ROWS, n);
  if (W == 64 && Cout == 128) {
    hipLaunchKernelGGL((conv1_mfma_kernel<128>), dim3((OH + C1M_ROWS - 1) / C1M_ROWS, n), dim3(256), 0, s, x, w, scale, shift, y, H, act);
    return hipGetLastError();
  }
  if (Cout == 128)
    hipLaunchKernelGGL((conv1_nchw_kernel<128, 2>), grid, dim3(256), lds, s, x, w, scale, shift, y, H, W, act);
  else if (Cout == 64)
    hipLaunchKernelGGL((conv1_nchw_kernel<64, 1>), grid, dim3(256), lds, s, x, w, scale, shift, y, H, W, act);
  else if (Cout == 256)
    hipLaunchKernelGGL((conv1_nchw_kernel<256, 2>), grid, dim3(256), lds, s, x, w, scale, shift, y, H, W, act);
  else
    return hipErrorInvalidValue;
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// dec_out (IAN_simple.py:171-181): x NHWC [n,H,W,Cin] -> y NCHW [n,Cout<=4,2H,2W], 5x5 s2 transposed conv.
// Block = 16x16 output tile = 8x8 input pixels (+1 halo) staged in LDS with a Cin+4 pixel stride.
// Each of the 4 waves owns a quarter of the input channels and computes ALL four output-parity classes
// (25 taps: balanced, unlike one 9/6/6/4-tap class per wave) for the 64 input positions, one per lane; the
// filter taps of its channel slice are wave-uniform and arrive through scalar loads.  The lane -> position map
// follows the hardware's ds_read_b128 service groups ({0-3,12-15,20-27},{4-11,16-19,28-31},+32): a group covers
// rows (g, g+4) of the 8x8 grid, whose 16 pixels fall into 16 distinct 16-byte bank slots for the 10-pixel row
// pitch (10*4 = 8 mod 16) -> conflict-free fragment reads.  The four channel-slice partials meet in LDS and
// are written as NCHW rows of 16 consecutive pixels.
// w packed [ky*5+kx][4][Cin] already holding the (possibly flipped) reference filter.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void dq_lane_to_pos(int lane, int& qy, int& qx) {
  const int l5 = lane & 31;
  const bool g1 = (l5 >= 4 && l5 < 12) || (l5 >= 16 && l5 < 20) || l5 >= 28;
  const int pos = g1 ? (l5 < 12 ? l5 - 4 : (l5 < 20 ? l5 - 8 : l5 - 16)) : (l5 < 4 ? l5 : (l5 < 16 ? l5 - 8 : l5 - 12));
  const int gi = (lane >> 5) * 2 + (g1 ? 1 : 0);
  qy = gi + 4 * (pos >> 3);
  qx = pos & 7;
}
__device__ __forceinline__ int dq_pos_to_lane(int qy, int qx) {
  const int gi = qy & 3, pos = (qy >> 2) * 8 + qx;
  const int l5 = (gi & 1) ? (pos < 8 ? pos + 4 : (pos < 12 ? pos + 8 : pos + 16)) : (pos < 4 ? pos : (pos < 8 ? pos + 8 : pos + 12));
  return (gi >> 1) * 32 + l5;
}

constexpr int DQ_NW = 8;  // waves per block: the scalar filter loads of one wave hide behind the other waves' FMAs
template <int CIN, int COUT>
__global__ __launch_bounds__(DQ_NW * 64) void deconv_out_nchw_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                              const float* __restrict__ scale,
                                                              const float* __restrict__ shift, float* __restrict__ y,
                                                              int H, int W, int act) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  constexpr int PS = CIN + 4;   // padded pixel stride
  constexpr int CW = CIN / DQ_NW;  // channels per wave
  const int tiles_x = W >> 3;
  const int n = blockIdx.y, ty = blockIdx.x / tiles_x, tx = blockIdx.x % tiles_x;
  const int iy0 = ty * 8 - 1, ix0 = tx * 8 - 1;
  // stage 10x10 input pixels x CIN
  for (int i = threadIdx.x; i < 100 * (CIN / 4); i += DQ_NW * 64) {
    const int pix = i / (CIN / 4), c4 = (i % (CIN / 4)) * 4;
    const int iy = iy0 + pix / 10, ix = ix0 + pix % 10;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W)
      v = *reinterpret_cast<const float4*>(x + ((size_t)(n * H + iy) * W + ix) * CIN + c4);
    *reinterpret_cast<float4*>(sm + pix * PS + c4) = v;
  }
  __syncthreads();
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  int qy, qx;
  dq_lane_to_pos(lane, qy, qx);
  const int cbase = wave * CW;
  float acc[4][COUT];
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int co = 0; co < COUT; ++co) acc[k][co] = 0.f;
  // oy = 2*iy - 2 + ky  =>  class py = ky&1, input row iy = qy + dy with dy = (py + 2 - ky)/2
#pragma unroll
  for (int ky = 0; ky < 5; ++ky) {
    const int py = ky & 1, dy = (py + 2 - ky) / 2;
#pragma unroll
    for (int kx = 0; kx < 5; ++kx) {
      const int px = kx & 1, dx = (px + 2 - kx) / 2;
      const int cls = py * 2 + px;
      const float* xs = sm + ((qy + dy + 1) * 10 + (qx + dx + 1)) * PS + cbase;
      const float* wt = w + (size_t)(ky * 5 + kx) * 4 * CIN + cbase;  // wave-uniform -> scalar loads
#pragma unroll
      for (int c = 0; c < CW; c += 4) {
        const float4 xv = *reinterpret_cast<const float4*>(xs + c);
#pragma unroll
        for (int co = 0; co < COUT; ++co) {
          const float4 wv = *reinterpret_cast<const float4*>(wt + co * CIN + c);
          float a = acc[cls][co];
          a = fmaf(xv.x, wv.x, a);
          a = fmaf(xv.y, wv.y, a);
          a = fmaf(xv.z, wv.z, a);
          a = fmaf(xv.w, wv.w, a);
          acc[cls][co] = a;
        }
      }
    }
  }
  __syncthreads();  // the input tile is dead: reuse LDS for the partials [wave][cls][co][lane]
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int co = 0; co < COUT; ++co) sm[((wave * 4 + k) * COUT + co) * 64 + lane] = acc[k][co];
  __syncthreads();
  const int OH = 2 * H, OW = 2 * W;
  if (threadIdx.x >= 256) return;
  const int oyl = threadIdx.x >> 4, oxl = threadIdx.x & 15;  // 16x16 output tile, rows of 16 consecutive pixels
  const int cls = (oyl & 1) * 2 + (oxl & 1);
  const int src = dq_pos_to_lane(oyl >> 1, oxl >> 1);
  const int oy = ty * 16 + oyl, ox = tx * 16 + oxl;
#pragma unroll
  for (int co = 0; co < COUT; ++co) {
    float v = 0.f;
#pragma unroll
    for (int wv = 0; wv < DQ_NW; ++wv) v += sm[((wv * 4 + cls) * COUT + co) * 64 + src];
    const float sc = scale ? scale[co] : 1.f, sh = shift ? shift[co] : 0.f;
    y[((size_t)(n * COUT + co) * OH + oy) * OW + ox] = m_act(v * sc + sh, act);
  }
}

template <int CIN>
static hipError_t launch_deconv_out_cin(const float* x, const float* w, const float* scale, const float* shift, float* y,
                                        int n, int H, int W, int Cout, int act, hipStream_t s) {
  dim3 grid((H / 8) * (W / 8), n);
  const size_t lds = (size_t)100 * (CIN + 4) * sizeof(float);
  switch (Cout) {
    case 1: hipLaunchKernelGGL((deconv_out_nchw_kernel<CIN, 1>), grid, dim3(DQ_NW * 64), lds, s, x, w, scale, shift, y, H, W, act); break;
    case 2: hipLaunchKernelGGL((deconv_out_nchw_kernel<CIN, 2>), grid, dim3(DQ_NW * 64), lds, s, x, w, scale, shift, y, H, W, act); break;
    case 3: hipLaunchKernelGGL((deconv_out_nchw_kernel<CIN, 3>), grid, dim3(DQ_NW * 64), lds, s, x, w, scale, shift, y, H, W, act); break;
    case 4: hipLaunchKernelGGL((deconv_out_nchw_kernel<CIN, 4>), grid, dim3(DQ_NW * 64), lds, s, x, w, scale, shift, y, H, W, act); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

// The same layer for one or a few images (the interactive decoder, batch 1): 16 tile workgroups cannot fill 256 CUs and
// their 1200-FMA lanes are a long serial chain, so here the work is cut the other way: 8 lanes per OUTPUT pixel, each lane
// a strided eighth of the input channels (lane p: channels 4p + 32j -> the 8 lanes of a pixel read 128 contiguous bytes
// per j), at most 9 taps of the pixel's parity class, operands straight from L2 (the whole input is 512 KB), an 8-lane
// butterfly at the end.  No LDS, no barrier; 128 workgroups at batch 1.
template <int CIN, int COUT>
__global__ __launch_bounds__(256) void deconv_out_px_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ scale, const float* __restrict__ shift,
                                                            float* __restrict__ y, float* __restrict__ mirror, int n, int H, int W,
                                                            int act) {
  // mirror (or nullptr): a second copy of the image, in practice the pinned host block of the interactive loop (zero-copy:
  // the store goes over the fabric to host memory, which spares the graph a device -> host copy node)
  const int gid = blockIdx.x * 256 + threadIdx.x;
  const int part = gid & 7;
  const int opix = gid >> 3;
  const int OH = 2 * H, OW = 2 * W;
  if (opix >= n * OH * OW) return;            // whole 8-lane groups leave together
  const int ox = opix % OW, oy = (opix / OW) % OH, b = opix / (OW * OH);
  float acc[COUT];
#pragma unroll
  for (int co = 0; co < COUT; ++co) acc[co] = 0.f;
  // oy = 2*iy - 2 + ky  =>  ky = (oy & 1) + 2j, iy = (oy + 2 - ky) / 2
#pragma unroll
  for (int jy = 0; jy < 3; ++jy) {
    const int ky = (oy & 1) + 2 * jy;
    const int iy = (oy + 2 - ky) >> 1;
    if (ky > 4 || (unsigned)iy >= (unsigned)H) continue;
#pragma unroll
    for (int jx = 0; jx < 3; ++jx) {
      const int kx = (ox & 1) + 2 * jx;
      const int ix = (ox + 2 - kx) >> 1;
      if (kx > 4 || (unsigned)ix >= (unsigned)W) continue;
      const float* xp = x + ((size_t)(b * H + iy) * W + ix) * CIN + part * 4;
      const float* wp = w + (size_t)(ky * 5 + kx) * 4 * CIN + part * 4;
#pragma unroll
      for (int j = 0; j < CIN / 32; ++j) {
        const float4 xv = *reinterpret_cast<const float4*>(xp + 32 * j);
#pragma unroll
        for (int co = 0; co < COUT; ++co) {
          const float4 wv = *reinterpret_cast<const float4*>(wp + co * CIN + 32 * j);
          float a = acc[co];
          a = fmaf(xv.x, wv.x, a);
          a = fmaf(xv.y, wv.y, a);
          a = fmaf(xv.z, wv.z, a);
          a = fmaf(xv.w, wv.w, a);
          acc[co] = a;
        }
      }
    }
  }
#pragma unroll
  for (int co = 0; co < COUT; ++co) {
    float v = acc[co];
    v += __shfl_xor(v, 1);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 4);
    if (part == 0) {
      const float sc = scale ? scale[co] : 1.f, sh = shift ? shift[co] : 0.f;
      const float o = m_act(v * sc + sh, act);
      const size_t yo = ((size_t)(b * COUT + co) * OH + oy) * OW + ox;
      y[yo] = o;
      if (mirror) mirror[yo] = o;
    }
  }
}

template <int CIN>
static hipError_t launch_deconv_out_px_cin(const float* x, const float* w, const float* scale, const float* shift, float* y,
                                           float* mirror, int n, int H, int W, int Cout, int act, hipStream_t s) {
  const long long threads = (long long)n * 4 * H * W * 8;
  const dim3 grid((unsigned)((threads + 255) / 256));
  switch (Cout) {
    case 1: hipLaunchKernelGGL((deconv_out_px_kernel<CIN, 1>), grid, dim3(256), 0, s, x, w, scale, shift, y, mirror, n, H, W, act); break;
    case 2: hipLaunchKernelGGL((deconv_out_px_kernel<CIN, 2>), grid, dim3(256), 0, s, x, w, scale, shift, y, mirror, n, H, W, act); break;
    case 3: hipLaunchKernelGGL((deconv_out_px_kernel<CIN, 3>), grid, dim3(256), 0, s, x, w, scale, shift, y, mirror, n, H, W, act); break;
    case 4: hipLaunchKernelGGL((deconv_out_px_kernel<CIN, 4>), grid, dim3(256), 0, s, x, w, scale, shift, y, mirror, n, H, W, act); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}
hipError_t launch_deconv_out_px(const float* x, const float* w, const float* scale, const float* shift, float* y, float* mirror, int n,
                                int H, int W, int Cin, int Cout, int act, hipStream_t s) {
  if (Cout > 4 || Cout < 1) return hipErrorInvalidValue;
  if (Cin == 128) return launch_deconv_out_px_cin<128>(x, w, scale, shift, y, mirror, n, H, W, Cout, act, s);
  if (Cin == 64) return launch_deconv_out_px_cin<64>(x, w, scale, shift, y, mirror, n, H, W, Cout, act, s);
  if (Cin == 256) return launch_deconv_out_px_cin<256>(x, w, scale, shift, y, mirror, n, H, W, Cout, act, s);
  return hipErrorInvalidValue;
}

hipError_t launch_deconv_out_nchw(const float* x, const float* w, const float* scale, const float* shift, float* y,
                                  int n, int H, int W, int Cin, int Cout, int act, hipStream_t s) {
  if (Cout > 4 || Cout < 1 || (H & 7) || (W & 7)) return hipErrorInvalidValue;
  if (Cin == 128) return launch_deconv_out_cin<128>(x, w, scale, shift, y, n, H, W, Cout, act, s);
  if (Cin == 64) return launch_deconv_out_cin<64>(x, w, scale, shift, y, n, H, W, Cout, act, s);
  if (Cin == 256) return launch_deconv_out_cin<256>(x, w, scale, shift, y, n, H, W, Cout, act, s);
  return hipErrorInvalidValue;
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

// The latent brush's first backward step in ONE launch (batch 1): loss seed (API.py:59,64) on the brush rectangle, the
// output activation's derivative and the backward-data of the image-producing transposed conv.  The seed is non-zero
// only inside the rectangle, so a thread visits only the taps whose output pixel lies in it (most threads: none).
//   gout[co,oy,ox] = seed(co,oy,ox) * act'(xhat) * oscale[co];   dx[iy,ix,ci] = (sum_{taps in patch} gout * w) * act_in'(yfwd) * scale[ci]
// Round 5: in the captured brush event `patch` lives in the mapped PINNED HOST block (zero-copy graphs), and round 4's form --
// every thread reading patch[0..3], i.e. one scalar load over PCIe per wave, 2048 waves -- made this the longest kernel of the
// event: 19.0 us (profiles/r05_batch1_chains.md, first pass).  Now ONE lane per workgroup fetches the rectangle and hands it on
// through LDS, and a thread owns four consecutive channels of a pixel (128 workgroups instead of 512, float4 filter loads and
// stores).  Same products in the same order per element: bitwise the round-4 gradient.
__global__ __launch_bounds__(256) void deconv_out_bwd_seed_kernel(const float* __restrict__ xhat, const float* __restrict__ rgb,
                                                                  const int* __restrict__ patch, int mode, int out_act,
                                                                  const float* __restrict__ oscale, const float* __restrict__ w,
                                                                  float* __restrict__ dx, const float* __restrict__ yfwd,
                                                                  const float* __restrict__ scale, int H, int W, int Cin,
                                                                  int Cout, int act) {
  __shared__ int sp[4];
  if (threadIdx.x < 4) sp[threadIdx.x] = patch[threadIdx.x];
  __syncthreads();
  const int c4n = Cin >> 2;
  const int idx4 = blockIdx.x * 256 + threadIdx.x;
  if (idx4 >= H * W * c4n) return;
  const int ci = (idx4 % c4n) * 4, pix = idx4 / c4n;
  const int ix = pix % W, iy = pix / W;
  const int OH = 2 * H, OW = 2 * W;
  const int c1 = sp[0], r1 = sp[1], c2 = sp[2], r2 = sp[3];
  const int cnt = 3 * (r2 - r1) * (c2 - c1);
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  if (cnt > 0) {
    const float inv = 1.f / (float)cnt;
    const int ky0 = max(0, r1 - (2 * iy - 2)), ky1 = min(5, r2 - (2 * iy - 2));
    const int kx0 = max(0, c1 - (2 * ix - 2)), kx1 = min(5, c2 - (2 * ix - 2));
    for (int ky = ky0; ky < ky1; ++ky) {
      const int oy = 2 * iy - 2 + ky;
      if ((unsigned)oy >= (unsigned)OH) continue;
      for (int kx = kx0; kx < kx1; ++kx) {
        const int ox = 2 * ix - 2 + kx;
        if ((unsigned)ox >= (unsigned)OW) continue;
        for (int co = 0; co < Cout; ++co) {
          const int o = (co * OH + oy) * OW + ox;
          const float xh = xhat[o];
          float gv = (mode == 0) ? inv : 2.f * (xh - rgb[o]) * inv;
          gv = gv * m_dact(xh, out_act) * (oscale ? oscale[co] : 1.f);
          const float4 wv = *reinterpret_cast<const float4*>(w + ((size_t)(ky * 5 + kx) * 4 + co) * Cin + ci);
          acc[0] = fmaf(gv, wv.x, acc[0]);
          acc[1] = fmaf(gv, wv.y, acc[1]);
          acc[2] = fmaf(gv, wv.z, acc[2]);
          acc[3] = fmaf(gv, wv.w, acc[3]);
        }
      }
    }
  }
  const size_t o4 = (size_t)pix * Cin + ci;
  float4 out;
  float* op = &out.x;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float d = yfwd ? m_dact(yfwd[o4 + e], act) : 1.f;
    op[e] = acc[e] * d * (scale ? scale[ci + e] : 1.f);
  }
  *reinterpret_cast<float4*>(dx + o4) = out;
}
hipError_t launch_deconv_out_bwd_seed(const float* xhat, const float* rgb, const int* patch, int mode, int out_act,
                                      const float* oscale, const float* w, float* dx, const float* yfwd, const float* scale, int H,
                                      int W, int Cin, int Cout, int act, hipStream_t s) {
  if (Cin & 3) return hipErrorInvalidValue;
  const int total = H * W * (Cin >> 2);
  hipLaunchKernelGGL(deconv_out_bwd_seed_kernel, dim3((total + 255) / 256), dim3(256), 0, s, xhat, rgb, patch, mode, out_act, oscale,
                     w, dx, yfwd, scale, H, W, Cin, Cout, act);
  return hipGetLastError();
}

hipError_t launch_deconv_out_bwd(const float* g, const float* w, float* dx, const float* yfwd, const float* scale,
                                 int n, int H, int W, int Cin, int Cout, int act, hipStream_t s) {
  const size_t total = (size_t)n * H * W * Cin;
  hipLaunchKernelGGL(deconv_out_bwd_kernel, dim3((total + 255) / 256), dim3(256), 0, s, g, w, dx, yfwd, scale, n, H,
                     W, Cin, Cout, act);
  return hipGetLastError();
}

// Batch-1 backward-data of the dense layer that consumes the latent (l_dec_fc2, IAN_simple.py:112-121; the last step of
// API.py:59,64's T.grad): dz[j] = sum_k g[k] * Wb[j][k].  One workgroup per latent row j streams its 32 KB weight row with
// every load in flight at once -- one launch instead of a split-K tap GEMM, its reduce pass and a row copy.
// upd (ian_brush_step, NPE.py:205-209 / 313-314): the workgroup that owns latent row j also applies the brush update
// Z[j] += coef * (dZ[j] * gscale) -- float32, every product rounded on its own, in the reference's order: bit-identical to the numpy
// expression and to latent_update_kernel (kernels_npe.hip), the separate launch this replaces in the captured brush event.  No
// other workgroup of this launch reads Z[j], and the next kernel of the chain (the decoder's first layer) reads the new row.
struct LatentUpd {
  float* z;          // the latent row, updated in place (nullptr: no update)
  const float* cg;   // (coef, gscale), device-readable
  float* z_mirror;   // or nullptr: the new latent ...
  float* g_mirror;   // ... and the gradient also go there (the pinned host block, zero-copy)
};
__device__ __forceinline__ float latent_step(float z, float g, float coef, float gscale) {
#pragma clang fp contract(off)
  float t = g * gscale;
  t = coef * t;
  return z + t;
}
__global__ __launch_bounds__(256) void dense_bwd_gemv_kernel(const float* __restrict__ g, const float* __restrict__ wb, int K,
                                                             const float* __restrict__ res, float* __restrict__ dz, LatentUpd upd) {
  __shared__ float part[4];
  const int row = blockIdx.x;
  const float* wr = wb + (size_t)row * K;
  // (coef, gscale) may live in mapped host memory: requested first, so that the PCIe round trip runs under the weight stream
  float cg0 = 0.f, cg1 = 0.f, zrow = 0.f;
  if (threadIdx.x == 0 && upd.z) { cg0 = upd.cg[0]; cg1 = upd.cg[1]; zrow = upd.z[row]; }
  float acc = 0.f;
#pragma unroll 8
  for (int k = threadIdx.x * 4; k < K; k += 1024) {
    const float4 a = *reinterpret_cast<const float4*>(g + k);
    const float4 w = *reinterpret_cast<const float4*>(wr + k);
    acc = fmaf(a.x, w.x, acc);
    acc = fmaf(a.y, w.y, acc);
    acc = fmaf(a.z, w.z, acc);
    acc = fmaf(a.w, w.w, acc);
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) acc += __shfl_down(acc, off);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float v = (part[0] + part[1]) + (part[2] + part[3]);
    if (res) v += res[row];
    dz[row] = v;
    if (upd.z) {
      const float zn = latent_step(zrow, v, cg0, cg1);
      upd.z[row] = zn;
      if (upd.z_mirror) upd.z_mirror[row] = zn;
      if (upd.g_mirror) upd.g_mirror[row] = v;
    }
  }
}
// Batch-1 forward of a dense layer fed by a short vector (l_dec_fc2: 100 -> 8192): y[o] = act(scale[o] * sum_k x[k] W[o][k]
// + shift[o]), slab [out][K] (K = padded input width, zero beyond the real inputs).  8 lanes per output, lane p takes the
// input columns 4p + 32j so that the 8 lanes of an output read whole 128-byte lines; an 8-lane butterfly at the end.
__global__ __launch_bounds__(256) void dense_fwd_gemv_kernel(const float* __restrict__ x, const float* __restrict__ w, int K,
                                                             int nout, const float* __restrict__ scale,
                                                             const float* __restrict__ shift, int act, float* __restrict__ y) {
  const int gid = blockIdx.x * 256 + threadIdx.x;
  const int part = gid & 7, o = gid >> 3;
  if (o >= nout) return;                      // whole 8-lane groups leave together
  const float* wr = w + (size_t)o * K + part * 4;
  const float* xr = x + part * 4;
  float acc = 0.f;
  for (int k = 0; k < K; k += 32) {
    const float4 a = *reinterpret_cast<const float4*>(xr + k);
    const float4 b = *reinterpret_cast<const float4*>(wr + k);
    acc = fmaf(a.x, b.x, acc);
    acc = fmaf(a.y, b.y, acc);
    acc = fmaf(a.z, b.z, acc);
    acc = fmaf(a.w, b.w, acc);
  }
  acc += __shfl_xor(acc, 1);
  acc += __shfl_xor(acc, 2);
  acc += __shfl_xor(acc, 4);
  if (part == 0) y[o] = m_act(acc * (scale ? scale[o] : 1.f) + (shift ? shift[o] : 0.f), act);
}
hipError_t launch_dense_fwd_gemv(const float* x, const float* w, int K, int nout, const float* scale, const float* shift, int act,
                                 float* y, hipStream_t s) {
  if (K & 31) return hipErrorInvalidValue;
  hipLaunchKernelGGL(dense_fwd_gemv_kernel, dim3((nout * 8 + 255) / 256), dim3(256), 0, s, x, w, K, nout, scale, shift, act, y);
  return hipGetLastError();
}

hipError_t launch_dense_bwd_gemv(const float* g, const float* wb, int rows, int K, const float* res, float* dz, float* upd_z,
                                 const float* upd_cg, float* z_mirror, float* g_mirror, hipStream_t s) {
  if (K & 3) return hipErrorInvalidValue;
  const LatentUpd u{upd_z, upd_cg, z_mirror, g_mirror};
  hipLaunchKernelGGL(dense_bwd_gemv_kernel, dim3(rows), dim3(256), 0, s, g, wb, K, res, dz, u);
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
// The graph is the one lasagne.layers.get_output builds from the reference's objects, not the textbook MADE:
// layers.py:775 overwrites MADE.input_layer with the first MaskedLayer, so get_output (helper.py: all_outputs[
// layer.input_layer]) hands MADE.get_output_for that layer's OUTPUT and the masked MLP runs on it:
//   h1 = relu(z.W0+b0);  h2 = relu(h1.W0+b0);  out = (h2.W1+b1) + (h1.WD+bD);  z' = (z - out_mu) / exp(out_ls)
// (pinned by executing layers.py: tests/golden/ref_layers.npz 'iaf/*', ref_IAN.npz 'z'; needs hidden == d).
__global__ __launch_bounds__(128) void made_iaf_kernel(const float* __restrict__ z, float* __restrict__ zo,
                                                       const float* __restrict__ wts, const float* __restrict__ bias,
                                                       int d, int zs) {
  __shared__ float zin[128], hm[128], hl[128], gm[128], gl[128];
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
  hm[t] = am > 0.f ? am : 0.f;   // h1 of the mu MADE
  hl[t] = al > 0.f ? al : 0.f;   // h1 of the ls MADE
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
  gm[t] = am > 0.f ? am : 0.f;   // h2
  gl[t] = al > 0.f ? al : 0.f;
  __syncthreads();
  if (t < d) {
    float om = bias[1 * d + t], dm = bias[2 * d + t];
    float ol = bias[4 * d + t], dl = bias[5 * d + t];
    for (int i = 0; i < d; ++i) {
      om = fmaf(gm[i], wts[1 * dd + i * d + t], om);
      dm = fmaf(hm[i], wts[2 * dd + i * d + t], dm);
      ol = fmaf(gl[i], wts[4 * dd + i * d + t], ol);
      dl = fmaf(hl[i], wts[5 * dd + i * d + t], dl);
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
// the same with the rectangle read from device memory (patch = {c1, r1, c2, r2}): nothing in the launch depends on the
// brush position, so the whole latent-brush step can be replayed from a captured graph while the brush moves
__global__ __launch_bounds__(256) void patch_seed_dev_kernel(const float* __restrict__ xhat, const float* __restrict__ rgb,
                                                             float* __restrict__ g, int H, int W, const int* __restrict__ patch,
                                                             int mode) {
  __shared__ int sp[4];   // `patch` may live in mapped host memory (zero-copy graphs): one fetch per workgroup, not one per wave
  if (threadIdx.x < 4) sp[threadIdx.x] = patch[threadIdx.x];
  __syncthreads();
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= 3 * H * W) return;
  const int c1 = sp[0], r1 = sp[1], c2 = sp[2], r2 = sp[3];
  const int xx = i % W, yy = (i / W) % H;
  const int cnt = 3 * (r2 - r1) * (c2 - c1);
  float v = 0.f;
  if (yy >= r1 && yy < r2 && xx >= c1 && xx < c2 && cnt > 0) {
    const float inv = 1.f / (float)cnt;
    v = (mode == 0) ? inv : 2.f * (xhat[i] - rgb[i]) * inv;
  }
  g[i] = v;
}
hipError_t launch_patch_seed_dev(const float* xhat, const float* rgb, float* g, int H, int W, const int* patch, int mode,
                                 hipStream_t s) {
  hipLaunchKernelGGL(patch_seed_dev_kernel, dim3((3 * H * W + 255) / 256), dim3(256), 0, s, xhat, rgb, g, H, W, patch, mode);
  return hipGetLastError();
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

// ------------------------------------------------------------------------------------------------
// MDCL with a handful of output filters (RGB-Beta head, IAN.py:183-206: 128 -> 2 channels at 64x64, scales [2,3,4]).
// On the MFMA path these layers pad Cout 2 -> 32 (16x wasted matrix work, 35 % of the full-IAN step); they are
// really FMA / LDS-bandwidth work, so: block = 4x16 output pixels, input tile with a 4-pixel halo (12x24 pixels x Cin)
// staged in LDS with a Cin+4 pixel stride, lane = pixel, wave = quarter of the input channels (all taps), filter taps
// wave-uniform -> scalar loads straight from the forward slab [tap][CoutPad][CinPad]; the four channel-slice partials
// meet in LDS.  A 16-lane ds_read_b128 service group is one 16-pixel row: 16 distinct bank slots for any tap shift.
// ------------------------------------------------------------------------------------------------
constexpr int MH_HALO = 4, MH_TH = 4, MH_TW = 16, MH_PH = MH_TH + 2 * MH_HALO, MH_PW = MH_TW + 2 * MH_HALO;

constexpr int MH_NW = 8;  // waves per block (one block per CU: 152 KB of LDS)
template <int CIN, int COUT>
__global__ __launch_bounds__(MH_NW * 64) void mdc_head_kernel(MdcHeadArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  constexpr int PS = CIN + 4, CW = CIN / MH_NW;
  const int tiles_x = a.W / MH_TW;
  const int n = blockIdx.y, ty = blockIdx.x / tiles_x, tx = blockIdx.x % tiles_x;
  const int iy0 = ty * MH_TH - MH_HALO, ix0 = tx * MH_TW - MH_HALO;
  for (int i = threadIdx.x; i < MH_PH * MH_PW * (CIN / 4); i += MH_NW * 64) {
    const int pix = i / (CIN / 4), c4 = (i % (CIN / 4)) * 4;
    const int iy = iy0 + pix / MH_PW, ix = ix0 + pix % MH_PW;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if ((unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W)
      v = *reinterpret_cast<const float4*>(a.x + ((size_t)(n * a.H + iy) * a.W + ix) * a.xs + c4);
    *reinterpret_cast<float4*>(sm + pix * PS + c4) = v;
  }
  __syncthreads();
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  // ds_read_b128 service groups: {0-3,12-15,20-27}, {4-11,16-19,28-31}, +32 -> group = tile row, position = column
  const int l5 = lane & 31;
  const bool g1 = (l5 >= 4 && l5 < 12) || (l5 >= 16 && l5 < 20) || l5 >= 28;
  const int qx = g1 ? (l5 < 12 ? l5 - 4 : (l5 < 20 ? l5 - 8 : l5 - 16)) : (l5 < 4 ? l5 : (l5 < 16 ? l5 - 8 : l5 - 12));
  const int qy = (lane >> 5) * 2 + (g1 ? 1 : 0);
  const int cbase = wave * CW;
  float acc[COUT];
#pragma unroll
  for (int co = 0; co < COUT; ++co) acc[co] = 0.f;
  for (int t = 0; t < a.ntaps; ++t) {
    const int dy = a.dy[t], dx = a.dx[t];  // wave-uniform
    const float* xs = sm + ((qy + dy + MH_HALO) * MH_PW + (qx + dx + MH_HALO)) * PS + cbase;
    const long long woff = (long long)t * a.w_tap_stride + cbase;
#pragma unroll
    for (int c = 0; c < CW; c += 4) {
      const float4 xv = *reinterpret_cast<const float4*>(xs + c);
#pragma unroll
      for (int co = 0; co < COUT; ++co) {
        const float4 wv = *reinterpret_cast<const float4*>(a.w[co] + woff + c);  // scalar loads
        float v = acc[co];
        v = fmaf(xv.x, wv.x, v);
        v = fmaf(xv.y, wv.y, v);
        v = fmaf(xv.z, wv.z, v);
        v = fmaf(xv.w, wv.w, v);
        acc[co] = v;
      }
    }
  }
  __syncthreads();  // input tile dead: partials [wave][co][lane]
#pragma unroll
  for (int co = 0; co < COUT; ++co) sm[(wave * COUT + co) * 64 + lane] = acc[co];
  __syncthreads();
  // thread -> (pixel, co): consecutive threads write consecutive channels of a pixel
  for (int i = threadIdx.x; i < 64 * COUT; i += MH_NW * 64) {
    const int co = i % COUT, pl = i / COUT;  // pl: pixel in row-major tile order
    const int py = pl / MH_TW, px = pl % MH_TW;
    // inverse of the lane map: row py = group (half = py>>1, g1 = py&1), column px = position
    const int gl = (py & 1) ? (px < 8 ? px + 4 : (px < 12 ? px + 8 : px + 16)) : (px < 4 ? px : (px < 8 ? px + 8 : px + 12));
    const int src = (py >> 1) * 32 + gl;
    float v = 0.f;
#pragma unroll
    for (int wv = 0; wv < MH_NW; ++wv) v += sm[(wv * COUT + co) * 64 + src];
    if (!a.y[co]) continue;  // padding slot of the filter table
    const size_t pix = (size_t)(n * a.H + ty * MH_TH + py) * a.W + tx * MH_TW + px;
    const size_t off = pix * a.ys[co] + a.yc[co];
    if (a.res[co]) v += a.res[co][off];
    v = v * a.scale[co] + a.shift[co];
    a.y[co][off] = m_act(v, a.act[co]);
  }
}

hipError_t launch_mdc_head(const MdcHeadArgs& args, int n, int Cin, int Cout, hipStream_t s) {
  if (Cout < 1 || Cout > MH_MAXCO || (args.H % MH_TH) || (args.W % MH_TW) || args.ntaps > 48) return hipErrorInvalidValue;
  MdcHeadArgs a = args;
  for (int co = Cout; co < MH_MAXCO; ++co) {  // the kernel is instantiated for 2/4/6/8 filters: pad with no-op slots
    a.w[co] = a.w[0];
    a.y[co] = nullptr;
    a.res[co] = nullptr;
  }
  dim3 grid((a.H / MH_TH) * (a.W / MH_TW), n);
  const size_t lds = (size_t)MH_PH * MH_PW * (Cin + 4) * sizeof(float);
#define MH_LAUNCH(CI, CO)                                                                                         \
  {                                                                                                               \
    static bool attr = false;                                                                                     \
    auto k = mdc_head_kernel<CI, CO>;                                                                             \
    if (!attr) {                                                                                                  \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
      if (e != hipSuccess) return e;                                                                              \
      attr = true;                                                                                                \
    }                                                                                                             \
    hipLaunchKernelGGL(k, grid, dim3(MH_NW * 64), lds, s, a);                                                     \
    return hipGetLastError();                                                                                     \
  }
  if (Cin == 128) {
    if (Cout <= 2) MH_LAUNCH(128, 2)
    if (Cout <= 4) MH_LAUNCH(128, 4)
    if (Cout <= 6) MH_LAUNCH(128, 6)
    MH_LAUNCH(128, 8)
  }
  if (Cin == 64) {
    if (Cout <= 2) MH_LAUNCH(64, 2)
    if (Cout <= 4) MH_LAUNCH(64, 4)
    if (Cout <= 6) MH_LAUNCH(64, 6)
    MH_LAUNCH(64, 8)
  }
#undef MH_LAUNCH
  return hipErrorInvalidValue;
}

// ------------------------------------------------------------------------------------------------
// backward-weight of a few-filter MDCL (RGB-Beta head, train_IAN.py:253-259 T.grad wrt R/G_a/B_a weights):
//   dS[t][co][ci] = sum_pixels dY[p][co] * X[p + d_t][ci],   co < COUT <= 4.
// On the MFMA path the 2 filters pad to a 32-wide tile (16x wasted work, 1.7 ms per layer at 128 images).  Here:
// persistent blocks walk 4x16 pixel tiles (input tile + 4-pixel halo and the dY tile in LDS), a thread owns one
// input channel and every NG-th tap, accumulates in registers over all its tiles and writes one partial
// [tap][co][ci]; head_wgrad_reduce sums the partials in block order (reproducible) into the slab layout.
// ------------------------------------------------------------------------------------------------
template <int CIN, int COUT>
__global__ __launch_bounds__(512) void mdc_head_wgrad_kernel(MdcHeadWgradArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  constexpr int NG = 512 / CIN, MAXT = (48 + NG - 1) / NG;
  float* xt = sm;                                 // [MH_PH*MH_PW][CIN]
  float* dyt = sm + MH_PH * MH_PW * CIN;          // [64][COUT]
  const int ci = threadIdx.x % CIN, g = threadIdx.x / CIN;
  float acc[MAXT][COUT];
#pragma unroll
  for (int k = 0; k < MAXT; ++k)
#pragma unroll
    for (int co = 0; co < COUT; ++co) acc[k][co] = 0.f;
  const int tiles_x = a.W / MH_TW, tiles_per_img = (a.H / MH_TH) * tiles_x;
  // LDS offset of every tap this thread owns (slots past the tap list point at the centre pixel: they accumulate
  // a harmless value that is never stored), hoisted out of the pixel loop: no branches, no scalar loads inside
  int off[MAXT];
#pragma unroll
  for (int k = 0; k < MAXT; ++k) {
    const int ti = g + k * NG;
    off[k] = (ti < a.ntaps) ? ((int)a.dy_[ti] * MH_PW + (int)a.dx_[ti]) * CIN : 0;
  }
  const int kmax = (a.ntaps + NG - 1) / NG;  // uniform: k slots in use
  for (int tile = blockIdx.x; tile < a.total_tiles; tile += gridDim.x) {
    const int n = tile / tiles_per_img, tr = tile % tiles_per_img;
    const int ty = tr / tiles_x, tx = tr % tiles_x;
    const int iy0 = ty * MH_TH - MH_HALO, ix0 = tx * MH_TW - MH_HALO;
    __syncthreads();  // the previous tile is fully consumed
    for (int i = threadIdx.x; i < MH_PH * MH_PW * (CIN / 4); i += 512) {
      const int pix = i / (CIN / 4), c4 = (i % (CIN / 4)) * 4;
      const int iy = iy0 + pix / MH_PW, ix = ix0 + pix % MH_PW;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if ((unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W)
        v = *reinterpret_cast<const float4*>(a.x + ((size_t)(n * a.H + iy) * a.W + ix) * a.xs + c4);
      *reinterpret_cast<float4*>(xt + pix * CIN + c4) = v;
    }
    for (int i = threadIdx.x; i < 64 * COUT; i += 512) {
      const int p = i / COUT, co = i % COUT;
      dyt[i] = a.dy[((size_t)(n * a.H + ty * MH_TH + p / MH_TW) * a.W + tx * MH_TW + p % MH_TW) * a.dys + co];
    }
    __syncthreads();
#pragma unroll 2
    for (int p = 0; p < 64; ++p) {
      float dv[COUT];
#pragma unroll
      for (int co = 0; co < COUT; ++co) dv[co] = dyt[p * COUT + co];  // broadcast
      const float* xp = xt + ((p / MH_TW + MH_HALO) * MH_PW + p % MH_TW + MH_HALO) * CIN + ci;
      float xv[MAXT];
#pragma unroll
      for (int k = 0; k < MAXT; ++k)
        if (k < kmax) xv[k] = xp[off[k]];
#pragma unroll
      for (int k = 0; k < MAXT; ++k)
        if (k < kmax)
#pragma unroll
          for (int co = 0; co < COUT; ++co) acc[k][co] = fmaf(dv[co], xv[k], acc[k][co]);
    }
  }
  float* out = a.partial + (size_t)blockIdx.x * a.ntaps * COUT * CIN;
#pragma unroll
  for (int k = 0; k < MAXT; ++k) {
    const int ti = g + k * NG;
    if (ti < a.ntaps)
#pragma unroll
      for (int co = 0; co < COUT; ++co) out[((size_t)ti * COUT + co) * CIN + ci] = acc[k][co];
  }
}

// dS[(t*f_rows + co)*f_cols + ci] = sum_b partial[b][t][co][ci]
// Round 6: the thin head layers have 136-528 outputs and 512 block partials each; one thread per output walking its 512 partials
// was a 2-workgroup launch of 512 dependent loads per thread -- 413 us per call, 3.3 ms of every four updates
// (profiles/r06_train_timeline_before.json).  Now a workgroup owns 8 outputs x 32 partial lanes: lane j sums blocks j, j + 32, ...
// (16 independent loads), then one thread per output adds the 32 lane sums in lane order -- a fixed order, bitwise reproducible
// from run to run (it is NOT the old block order: sums differ from round 5's in the last bits).
__global__ __launch_bounds__(256) void head_wgrad_reduce_kernel(const float* __restrict__ partial, int nblocks, int ntaps,
                                                                int COUT, int CIN, float* __restrict__ dS, int f_rows,
                                                                int f_cols) {
  __shared__ float red[32][8];
  const int per = ntaps * COUT * CIN;
  const int ol = threadIdx.x & 7, bl = threadIdx.x >> 3;
  const int i = blockIdx.x * 8 + ol;
  float s = 0.f;
  if (i < per)
    for (int b = bl; b < nblocks; b += 32) s += partial[(size_t)b * per + i];
  red[bl][ol] = s;
  __syncthreads();
  if (threadIdx.x < 8 && i < per) {
    float t = 0.f;
#pragma unroll
    for (int j = 0; j < 32; ++j) t += red[j][ol];
    const int ci = i % CIN, co = (i / CIN) % COUT, tp = i / (CIN * COUT);
    dS[((size_t)tp * f_rows + co) * f_cols + ci] = t;
  }
}

hipError_t launch_mdc_head_wgrad(const MdcHeadWgradArgs& a, int nblocks, int Cin, int Cout, float* dS, int f_rows,
                                 int f_cols, hipStream_t s) {
  if (Cout < 1 || Cout > 4 || (a.H % MH_TH) || (a.W % MH_TW) || a.ntaps > 48) return hipErrorInvalidValue;
  const int cpad = Cout <= 2 ? 2 : 4;
  const size_t lds = ((size_t)MH_PH * MH_PW * Cin + 64 * cpad) * sizeof(float);
#define MW_LAUNCH(CI, CO)                                                                                         \
  {                                                                                                               \
    static bool attr = false;                                                                                     \
    auto k = mdc_head_wgrad_kernel<CI, CO>;                                                                       \
    if (!attr) {                                                                                                  \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
      if (e != hipSuccess) return e;                                                                              \
      attr = true;                                                                                                \
    }                                                                                                             \
    hipLaunchKernelGGL(k, dim3(nblocks), dim3(512), lds, s, a);                                                   \
  }
  if (Cin == 4 && Cout <= 2) MW_LAUNCH(4, 2)
  else if (Cin == 4) MW_LAUNCH(4, 4)
  else if (Cin == 128 && Cout <= 2) MW_LAUNCH(128, 2)
  else if (Cin == 128) MW_LAUNCH(128, 4)
  else if (Cin == 64 && Cout <= 2) MW_LAUNCH(64, 2)
  else if (Cin == 64) MW_LAUNCH(64, 4)
  else return hipErrorInvalidValue;
#undef MW_LAUNCH
  const int per = a.ntaps * cpad * Cin;
  hipLaunchKernelGGL(head_wgrad_reduce_kernel, dim3((per + 7) / 8), dim3(256), 0, s, a.partial, nblocks, a.ntaps, cpad, Cin,
                     dS, f_rows, f_cols);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// "thin" MDCL: at most 4 real input channels AND at most 4 filters (IAN.py:187-206: G_b reads the 2-channel R map,
// B_b the 4-channel [R,G] map; forward and backward-data).  (The kernel also instantiates for 64-256 output
// channels, but that form lost 10x to the padded MFMA tile and is not dispatched.)
// On the MFMA path the contraction pads 2-4 channels to 32.  Here a thread owns one output channel: its
// taps x 4 weights live in registers for the whole pixel loop, the 16-byte input pixel is read straight from
// global/L1 (all output-channel lanes of a pixel share the address), 4 FMAs per tap, coalesced NHWC stores.
//   y[p][co] = act((sum_t sum_ci x[p+d_t][ci] * w[t][co][ci] + res[p][co]) * scale[co] + shift[co])
// ------------------------------------------------------------------------------------------------
constexpr int MT_MAXT = 36;
template <int CO_T>  // output-channel lanes per pixel: 4 (Cout <= 4) or Cout (64 / 128 / 256)
__global__ __launch_bounds__(512) void mdc_thin_kernel(MdcThinArgs a) {
  constexpr int PG = 512 / CO_T;  // pixels in flight per block
  const int co = threadIdx.x % CO_T, grp = threadIdx.x / CO_T;
  const bool co_ok = co < a.Cout;
  float4 wr[MT_MAXT];
#pragma unroll
  for (int t = 0; t < MT_MAXT; ++t)
    wr[t] = (t < a.ntaps && co_ok) ? *reinterpret_cast<const float4*>(a.w + ((size_t)t * a.CoutPad + co) * a.CinPad)
                                   : make_float4(0.f, 0.f, 0.f, 0.f);
  const float sc = (a.scale && co_ok) ? a.scale[co] : 1.f, sh = (a.shift && co_ok) ? a.shift[co] : 0.f;
  const long long npix = (long long)a.n * a.H * a.W;
  for (long long p = (long long)blockIdx.x * PG + grp; p < npix; p += (long long)gridDim.x * PG) {
    const int px = (int)(p % a.W), py = (int)((p / a.W) % a.H);
    float acc = 0.f;
#pragma unroll
    for (int t = 0; t < MT_MAXT; ++t) {
      if (t < a.ntaps) {
        const int iy = py + a.dy[t], ix = px + a.dx[t];
        if ((unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W) {
          const float4 xv = *reinterpret_cast<const float4*>(a.x + (p + (long long)a.dy[t] * a.W + a.dx[t]) * a.xs);
          acc = fmaf(xv.x, wr[t].x, acc);
          acc = fmaf(xv.y, wr[t].y, acc);
          acc = fmaf(xv.z, wr[t].z, acc);
          acc = fmaf(xv.w, wr[t].w, acc);
        }
      }
    }
    if (co_ok) {
      const size_t off = (size_t)p * a.ys + co;
      if (a.res) acc += a.res[off];
      a.y[off] = m_act(acc * sc + sh, a.act);
    }
  }
}

// The same layer with the input staged through LDS.  Above, every tap of every pixel is a 16-byte read out of its own
// 128-byte line of the 32-float-stride NHWC map (33 taps x 16 lines per wave-load: the texture-address path is the bound,
// 205 us per call at 128 images for 0.3 GFLOP).  Here a workgroup owns MTT_ROWS rows of one image: it copies those rows plus
// a 4-pixel frame (zeros outside the image) into LDS as 16-byte pixels ONCE -- one line touch per pixel -- and the 33 taps
// read LDS (4 lanes per pixel share an address, 16 consecutive pixels per wave-load: conflict free).  Same taps in the same
// order with the same FMAs; an out-of-image tap now adds 0*w instead of being skipped -> identical values.
constexpr int MTT_ROWS = 8, MTT_R = 4;
__global__ __launch_bounds__(512) void mdc_thin_tile_kernel(MdcThinArgs a) {
  extern __shared__ __attribute__((aligned(16))) float4 mtt_sm[];   // [(MTT_ROWS + 2R)][W + 2R]
  const int WP = a.W + 2 * MTT_R;
  const int bands = a.H / MTT_ROWS;
  const int img = blockIdx.x / bands, y0 = (blockIdx.x % bands) * MTT_ROWS;
  const int co = threadIdx.x & 3, grp = threadIdx.x >> 2;
  const bool co_ok = co < a.Cout;
  float4 wr[MT_MAXT];
#pragma unroll
  for (int t = 0; t < MT_MAXT; ++t)
    wr[t] = (t < a.ntaps && co_ok) ? *reinterpret_cast<const float4*>(a.w + ((size_t)t * a.CoutPad + co) * a.CinPad)
                                   : make_float4(0.f, 0.f, 0.f, 0.f);
  const float sc = (a.scale && co_ok) ? a.scale[co] : 1.f, sh = (a.shift && co_ok) ? a.shift[co] : 0.f;
  const float* ximg = a.x + (size_t)img * a.H * a.W * a.xs;
  for (int i = threadIdx.x; i < (MTT_ROWS + 2 * MTT_R) * WP; i += 512) {
    const int ry = i / WP, rx = i - ry * WP;
    const int iy = y0 + ry - MTT_R, ix = rx - MTT_R;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if ((unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W)
      v = *reinterpret_cast<const float4*>(ximg + ((size_t)iy * a.W + ix) * a.xs);
    mtt_sm[i] = v;
  }
  __syncthreads();
  for (int q = grp; q < MTT_ROWS * a.W; q += 128) {
    const int ly = q / a.W, lx = q - ly * a.W;
    const float4* c = mtt_sm + (ly + MTT_R) * WP + lx + MTT_R;
    float acc = 0.f;
#pragma unroll
    for (int t = 0; t < MT_MAXT; ++t) {
      if (t < a.ntaps) {
        const float4 xv = c[a.dy[t] * WP + a.dx[t]];
        acc = fmaf(xv.x, wr[t].x, acc);
        acc = fmaf(xv.y, wr[t].y, acc);
        acc = fmaf(xv.z, wr[t].z, acc);
        acc = fmaf(xv.w, wr[t].w, acc);
      }
    }
    if (co_ok) {
      const size_t off = ((size_t)(img * a.H + y0 + ly) * a.W + lx) * a.ys + co;
      if (a.res) acc += a.res[off];
      a.y[off] = m_act(acc * sc + sh, a.act);
    }
  }
}

hipError_t launch_mdc_thin(const MdcThinArgs& a, hipStream_t s) {
  if (a.ntaps > MT_MAXT || a.Cout < 1) return hipErrorInvalidValue;
  const long long npix = (long long)a.n * a.H * a.W;
  auto blocks = [&](int pg) { return (int)std::min<long long>((npix + pg - 1) / pg, 256 * 8); };
  bool frame_ok = a.Cout <= 4 && (a.H % MTT_ROWS) == 0 && a.W <= 128 && (long long)a.n * (a.H / MTT_ROWS) >= 256;
  for (int t = 0; t < a.ntaps && frame_ok; ++t)
    frame_ok = a.dy[t] >= -MTT_R && a.dy[t] <= MTT_R && a.dx[t] >= -MTT_R && a.dx[t] <= MTT_R;
  if (frame_ok && !a.no_tile) {   // enough row bands to fill the chip: stage the rows through LDS
    const size_t lds = (size_t)(MTT_ROWS + 2 * MTT_R) * (a.W + 2 * MTT_R) * sizeof(float4);
    hipLaunchKernelGGL(mdc_thin_tile_kernel, dim3(a.n * (a.H / MTT_ROWS)), dim3(512), lds, s, a);
    return hipGetLastError();
  }
  if (a.Cout <= 4) hipLaunchKernelGGL(mdc_thin_kernel<4>, dim3(blocks(128)), dim3(512), 0, s, a);
  else if (a.Cout == 64) hipLaunchKernelGGL(mdc_thin_kernel<64>, dim3(blocks(8)), dim3(512), 0, s, a);
  else if (a.Cout == 128) hipLaunchKernelGGL(mdc_thin_kernel<128>, dim3(blocks(4)), dim3(512), 0, s, a);
  else if (a.Cout == 256) hipLaunchKernelGGL(mdc_thin_kernel<256>, dim3(blocks(2)), dim3(512), 0, s, a);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// latent-brush backward through the non-GEMM nodes of the full IAN decoder (IAN.py:139-207)
// ------------------------------------------------------------------------------------------------
// Gradient hand-over along an identity edge (ElemwiseSum residual layers.py:412-416, stand-alone BatchNorm
// layers.py:412, ConcatLayer IAN.py:201): the value-gradient arriving in gs (channels coff..coff+C of an NHWC
// buffer with pixel stride ss) becomes the gradient wrt the pre-epilogue value of the target slot's producer:
//   gd[p,c] (+)= gs[p,coff+c] * act'(y[p,c]) * scale[c]
__global__ __launch_bounds__(256) void grad_pass_kernel(const float* __restrict__ gs, int ss, int coff,
                                                        float* __restrict__ gd, const float* __restrict__ y, int ds,
                                                        const float* __restrict__ scale, long long npix, int C,
                                                        int act, int accumulate) {
  const long long total = npix * C;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long p = i / C;
    const int c = (int)(i % C);
    float v = gs[p * ss + coff + c];
    if (y) v *= m_dact(y[p * ds + c], act);
    if (scale) v *= scale[c];
    float* o = gd + p * ds + c;
    *o = accumulate ? *o + v : v;
  }
}
hipError_t launch_grad_pass(const float* gs, int ss, int coff, float* gd, const float* y, int ds, const float* scale,
                            long long npix, int C, int act, int accumulate, hipStream_t s) {
  long long total = npix * C;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(grad_pass_kernel, dim3(blocks), dim3(256), 0, s, gs, ss, coff, gd, y, ds, scale, npix, C, act,
                     accumulate);
  return hipGetLastError();
}

// backward of beta_layer x3 + concat (layers.py:397-408, IAN.py:207): out_c = 2a/(a+b+1e-8) - 1 with (a,b) the two
// channels of map c.  gout NCHW [n,3,hw] -> gradient wrt the pre-activation of each map's producer
// (times act'(v) * scale of that producer); maps are NHWC with pixel stride rs.
__global__ __launch_bounds__(256) void beta_bwd_kernel(const float* __restrict__ gout, BetaBwdArgs a, int n, int hw,
                                                       int rs) {
  const long long total = (long long)n * hw;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int b = (int)(i / hw), p = (int)(i % hw);
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float va = a.v[c][i * rs + 0], vb = a.v[c][i * rs + 1];
    const float g = gout[((size_t)b * 3 + c) * hw + p];
    const float den = va + vb + 1e-8f;
    const float inv2 = 2.f / (den * den);
    float ga = g * inv2 * (vb + 1e-8f);
    float gb = -g * inv2 * va;
    ga *= m_dact(va, a.act[c]) * (a.scale[c] ? a.scale[c][0] : 1.f);
    gb *= m_dact(vb, a.act[c]) * (a.scale[c] ? a.scale[c][1] : 1.f);
    float* o = a.g[c] + i * rs;
    if (a.accumulate[c]) {
      o[0] += ga;
      o[1] += gb;
    } else {
      o[0] = ga;
      o[1] = gb;
    }
  }
}
hipError_t launch_beta_bwd(const float* gout, const BetaBwdArgs& a, int n, int hw, int rs, hipStream_t s) {
  const long long total = (long long)n * hw;
  hipLaunchKernelGGL(beta_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, gout, a, n, hw, rs);
  return hipGetLastError();
}


// ---- ian_box_probe (include/ian.h): what the fp32 matrix pipe of THIS box sustains ------------------------------------------------
// v_mfma_f32_32x32x2_f32 issued back to back from registers (four independent accumulators per wave, 2 workgroups of 4 waves per CU,
// non-zero operands, no memory traffic inside the loop).  Launched back to back with a chosen loop length: the part clocks down under
// fp32-MFMA load and every launch carries a clock / fill ramp, so the figure depends on the launch length -- the bench line quotes it
// at the launch length of one IAN_simple batch-64 layer (~200 us) next to the spec peak.
typedef float f32x16_probe __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void box_probe_kernel(const float* __restrict__ in, float* __restrict__ out, int iters) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  const float a0 = in[t * 4 + 0], a1 = in[t * 4 + 1], b0 = in[t * 4 + 2], b1 = in[t * 4 + 3];
  f32x16_probe c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
  for (int i = 0; i < iters; ++i) {
    c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, c3, 0, 0, 0);
  }
  float s = 0;
#pragma unroll
  for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
  out[t] = s;
}
hipError_t launch_box_probe(const float* in, float* out, int blocks, int iters, hipStream_t s) {
  hipLaunchKernelGGL(box_probe_kernel, dim3(blocks), dim3(256), 0, s, in, out, iters);
  return hipGetLastError();
}

}  // namespace ian

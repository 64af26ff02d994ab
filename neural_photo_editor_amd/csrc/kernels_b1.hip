// b1conv: the 5x5 / stride-2 transposed convolution (layers.py:436-483, DeconvLayer) and its backward-data form for
// ONE image -- the interactive decoder of the NPE latent brush (NPE.py:192-235: API.py:46-47 sample_at, :59,:64 the two
// T.grad functions).  At batch 1 these layers are weight streams (dec_conv1: 52 MB of filters for 16 input pixels,
// 8 FLOP per byte), not GEMMs: the batch-N tapgemm needs split-K over workgroups to fill the chip, i.e. partial-sum
// slabs plus a second launch per layer, and a 32-deep K-step of 16 MFMAs cannot hide a weight fetch.  Here
//   * a workgroup owns a 4x4 block of output positions (of one output-parity class for the transposed form) x 16 output
//     channels and runs the WHOLE contraction for it: no slabs, no reduce launch, the epilogue (folded batch-norm +
//     activation, or act'(y)*scale for the backward chain) is applied in place;
//   * the contraction is split over the 8 waves of the workgroup (wave w takes K-steps w, w+8, ...), combined once
//     through LDS at the end;
//   * weights are repacked on the host so that the 8 waves together read ONE contiguous stream per workgroup:
//     [class][channel slice][tap][32-channel step][lane][8 floats] -- lane l = (n = l & 15, kg = l >> 4) holds the 8
//     consecutive reduction channels kg*8 .. kg*8+7 of output channel n, i.e. exactly its B operands of eight
//     v_mfma_f32_16x16x4_f32 (the k index is permuted identically for A and B); four K-steps (8 KB of weights per wave, 64 KB
//     per workgroup, two workgroups per CU) are in flight;
//   * the few input pixels a block can touch (<= 6x6 for the transposed form, <= 11x11 for the backward form) stay in
//     L1 / L2 and are read per K-step through a buffer descriptor (out-of-image pixels: out-of-range offset -> zeros), in the
//     same register queue as the weights -- no LDS staging pass and no barrier ahead of the first MFMA;
//   * exact fp32: v_mfma_f32_16x16x4_f32 (32-cycle issue, two accumulators per wave cover the 40-cycle dependent latency).
#include "ian_internal.h"

namespace ian {

typedef float b1_f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float b1_act(float v, int act) {
  switch (act) {
    case 1: return v > 0.f ? v : 0.f;
    case 2: return v > 0.f ? v : 0.2f * v;
    case 3: return v > 0.f ? v : expm1f(v);
    case 4: return tanhf(v);
    case 5: return 1.f / (1.f + __expf(-v));
    default: return v;
  }
}
__device__ __forceinline__ float b1_act_grad(float y, int act) {  // through the OUTPUT y, as tapgemm's epilogue does
  switch (act) {
    case 1: return y > 0.f ? 1.f : 0.f;
    case 2: return y > 0.f ? 1.f : 0.2f;
    case 3: return y > 0.f ? 1.f : y + 1.f;
    case 4: return 1.f - y * y;
    case 5: return y * (1.f - y);
    default: return 1.f;
  }
}

constexpr int B1_WAVES = 8;
constexpr int B1_PF = 4;  // K-steps in flight per wave: 4 x (2 KB of weights + 2 KB of input rows); 2 workgroups per CU

typedef unsigned int b1_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 b1_buf_load4(__amdgpu_buffer_rsrc_t rsrc, unsigned voff) {
  const b1_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 0);
  return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

// MODE 0: transposed conv forward (out = 2*in, 4 parity classes);  MODE 1: its backward-data (out = in/2, 25 taps)
template <int MODE>
__global__ __launch_bounds__(64 * B1_WAVES) void b1conv_kernel(const B1Params p) {
  __shared__ float part[B1_WAVES * 256];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // ---- which block ---------------------------------------------------------------------------------------------
  int b = blockIdx.x;
  const int nslice = b % p.nslices;
  b /= p.nslices;
  const int tile = b % p.ntiles;
  const int cls = b / p.ntiles;  // MODE 1: always 0
  const int ty = tile / p.tiles_x, tx = tile % p.tiles_x;
  const int qy0 = ty * 4, qx0 = tx * 4;
  const int py = MODE == 0 ? (cls >> 1) : 0, px = MODE == 0 ? (cls & 1) : 0;
  const int nky = MODE == 0 ? (py ? 2 : 3) : 5, nkx = MODE == 0 ? (px ? 2 : 3) : 5;
  const int ntaps = nky * nkx;
  const int kpt = p.Cr >> 5;  // K-steps per tap
  const int nks = ntaps * kpt;

  // The input rows (<= 121 pixels per block, all of them L1/L2-resident: the whole activation is <= 512 KB) are read
  // straight from global memory through a buffer descriptor -- a pixel outside the image gets an out-of-range offset and
  // the hardware returns zeros -- in the SAME register queue as the weights: no LDS staging pass, no barrier before the
  // first MFMA, one latency chain per launch (launch -> first loads -> MFMAs as they land -> combine).
  const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.x_bytes, 0x00020000);
  const int m = lane & 15, kg = lane >> 4;
  const int qy = qy0 + (m >> 2), qx = qx0 + (m & 3);

  // ---- weight stream: this block's contiguous region, wave w reads K-steps w, w+8, ... ----------------------------
  // The loop is branch-free (a step past the end re-reads the last weight block against an all-zero input row), so that
  // hipcc counts the loads in flight (s_waitcnt vmcnt(N)) instead of draining them at every step.
  const float* wlane = p.w + p.cls_off[cls] + (size_t)nslice * nks * 512 + lane * 8;
  const int kshift = p.kshift;                                   // kpt = Cr / 32 is a power of two (host-checked)
  const int ymul = nkx == 2 ? 32 : (nkx == 3 ? 22 : 13);         // (tap * ymul) >> 6 == tap / nkx for tap < 25
  const int my_steps = (nks - wave + B1_WAVES - 1) / B1_WAVES;
  const int last_ks = wave + (my_steps - 1) * B1_WAVES;
  float4 bq[B1_PF][2], aq[B1_PF][2];
#pragma unroll
  for (int u = 0; u < B1_PF; ++u) bq[u][0] = bq[u][1] = aq[u][0] = aq[u][1] = make_float4(1.f, 2.f, 3.f, 4.f);

#define B1_ISSUE(U, KS)                                                                                      \
  {                                                                                                          \
    const int ks_ = (KS);                                                                                    \
    const int kc_ = ks_ > last_ks ? last_ks : ks_;                                                           \
    const float4* src_ = reinterpret_cast<const float4*>(wlane + kc_ * 512);                                 \
    if (p.dbg != 2) {                                                                                        \
      bq[U][0] = src_[0];                                                                                    \
      bq[U][1] = src_[1];                                                                                    \
    }                                                                                                        \
    const int tap_ = kc_ >> kshift, cstep_ = kc_ & (kpt - 1);                                                \
    const int ty_ = (tap_ * ymul) >> 6, tx_ = tap_ - ty_ * nkx;   /* no integer division in the K loop */    \
    int iy_, ix_;                                                                                            \
    if (MODE == 0) { /* tap (ky = py + 2 ty, kx = px + 2 tx): input pixel q + (p + 2 - k) / 2 = q + 1 - t */ \
      iy_ = qy + 1 - ty_; ix_ = qx + 1 - tx_;                                                                \
    } else {         /* tap (ky, kx) = (ty, tx): dY pixel 2 q - 2 + k */                                     \
      iy_ = 2 * qy - 2 + ty_; ix_ = 2 * qx - 2 + tx_;                                                        \
    }                                                                                                        \
    const bool ok_ = (ks_ <= last_ks) & ((unsigned)iy_ < (unsigned)p.IH) & ((unsigned)ix_ < (unsigned)p.IW); \
    const unsigned off_ = ok_ ? (unsigned)(((iy_ * p.IW + ix_) * p.xs + (cstep_ << 5) + kg * 8) * 4) : 0xFFFFFFE0u; \
    if (p.dbg != 1) {                                                                                        \
      aq[U][0] = b1_buf_load4(xrsrc, off_);                                                                  \
      aq[U][1] = b1_buf_load4(xrsrc, off_ + 16u);                                                            \
    }                                                                                                        \
  }

#pragma unroll
  for (int u = 0; u < B1_PF; ++u) B1_ISSUE(u, wave + u * B1_WAVES)

  // ---- K loop --------------------------------------------------------------------------------------------------------
  b1_f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  for (int i = 0; i < my_steps; i += B1_PF) {
#pragma unroll
    for (int u = 0; u < B1_PF; ++u) {
      const float4 b0 = bq[u][0], b1 = bq[u][1], a0 = aq[u][0], a1 = aq[u][1];
      B1_ISSUE(u, wave + (i + u + B1_PF) * B1_WAVES)
      if (p.dbg == 3) { acc0[0] += a0.x * b0.x + a1.w * b1.w; continue; }
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, b0.x, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, b0.y, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, b0.z, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, b0.w, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, b1.x, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, b1.y, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, b1.z, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, b1.w, acc1, 0, 0, 0);
    }
  }
#undef B1_ISSUE

  // ---- combine the 8 waves' partial 16x16 tiles (fixed order -> reproducible), epilogue, store -------------------------
  {
    // C layout: col n = lane & 15, rows 4*(lane >> 4) + r
    float* mine = part + wave * 256;
#pragma unroll
    for (int r = 0; r < 4; ++r) mine[(4 * kg + r) * 16 + m] = acc0[r] + acc1[r];
  }
  __syncthreads();
  if (tid < 256) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < B1_WAVES; ++w) s += part[w * 256 + tid];
    const int mm = tid >> 4, n = tid & 15;
    const int oqy = qy0 + (mm >> 2), oqx = qx0 + (mm & 3);
    const int oy = MODE == 0 ? 2 * oqy + py : oqy, ox = MODE == 0 ? 2 * oqx + px : oqx;
    const int c = nslice * 16 + n;
    const size_t off = ((size_t)oy * p.OW + ox) * p.ys + c;
    const int si = p.scale_period ? (int)(off % (size_t)p.scale_period) : c;
    const float sc = p.scale ? p.scale[si] : 1.f;
    if (p.bwd) {
      if (p.res) s += p.res[off];
      const float yf = p.yfwd ? p.yfwd[off] : 0.f;
      p.y[off] = s * b1_act_grad(yf, p.act) * sc;
    } else {
      const float sh = p.shift ? p.shift[si] : 0.f;
      p.y[off] = b1_act(s * sc + sh, p.act);
    }
  }
}

hipError_t launch_b1conv(const B1Params& p, int mode, hipStream_t s) {
  const int nclass = mode == 0 ? 4 : 1;
  const int blocks = nclass * p.ntiles * p.nslices;
  if (mode == 0)
    hipLaunchKernelGGL(b1conv_kernel<0>, dim3(blocks), dim3(64 * B1_WAVES), 0, s, p);
  else
    hipLaunchKernelGGL(b1conv_kernel<1>, dim3(blocks), dim3(64 * B1_WAVES), 0, s, p);
  return hipGetLastError();
}

}  // namespace ian

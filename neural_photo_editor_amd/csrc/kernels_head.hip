// RGB-Beta head of the full IAN in two launches (IAN.py:183-207; layers.py:207-258 MDCL, 397-408 beta_layer).
//
//   R = sigmoid(MDCL_R(h))            G = sigmoid(MDCL_Ga(h) + MDCL_Gb(R))          B = sigmoid(MDCL_Ba(h) + MDCL_Bb([R,G]))
//   out_c = 2a/(a+b+1e-8) - 1  for (a,b) = the two channels of R, G, B           h: (n,128,64,64), every MDCL 2 filters, scales [2,3,4]
//
// Round 1 ran this as 5 launches over 32-channel-padded 2-channel maps (3.27 ms of the 18.7 ms full-IAN step at batch 256,
// h read 2.8x).  The three filters pairs that read h (R, G_a, B_a: 6 filters x 33 taps x 128 channels = 98 % of the head's
// arithmetic) are the problem: 6 output channels waste an MFMA tile, and as FMA work they ran at 16 % of the VALU peak.
//
// head6_kernel: "contract first, shift later".  out[p][f] = sum_t sum_c h[p+d_t][c] * W[t][f][c] is evaluated as
//       Y[q][(t,f)] = sum_c h[q][c] * W[t][f][c]              -- ONE dense 128 -> 198 contraction per input pixel: fp32 MFMA,
//                                                                N = 33 taps x 6 filters = 198 of 224 tile columns used
//       out[p][f]   = sum_t Y[p + d_t][(t,f)]                 -- 33 shifted adds per output value, on the VALU, through LDS
//   so h is read from HBM exactly once (no halo when a workgroup owns whole image rows), the weights sit in registers
//   (128 VGPRs per lane for the whole kernel) and the matrix cores see a well-shaped 64 x 224 x 128 tile per image row.
//   A workgroup walks the input rows of its band in order; Y of one row lives in LDS; an output row is open while the
//   input rows y-4..y+4 pass by (ring of 16 rows x 64 x 6 accumulators in LDS) -- contributions are added by fixed owner
//   threads in a fixed order, so results are bitwise reproducible (no atomics).  Result: a compact [n][H][W][8] map
//   (R after its sigmoid, G_a and B_a raw), 32 B per pixel instead of 3 x 128 B.
// head_tail_kernel: one workgroup per image keeps the compact map in LDS (96 KB) and runs G_b (2->2), B_b (4->2), the
//   sigmoids and the three beta_layers, writing the NCHW image.  ~1.6 M MAC per image: LDS-resident VALU work.
#include <algorithm>

#include "ian_internal.h"

namespace ian {

typedef float hd_f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float hd_act(float v, int act) {
  switch (act) {
    case 1: return v > 0.f ? v : 0.f;
    case 2: return v > 0.f ? v : 0.2f * v;
    case 3: return v > 0.f ? v : expm1f(v);
    case 4: return tanhf(v);
    case 5: return 1.f / (1.f + __expf(-v));
    default: return v;
  }
}

constexpr int HD_W = 64;          // image width (pixels per row = M of the per-row GEMM)
constexpr int HD_CIN = 128;       // input channels (K)
constexpr int HD_XS = HD_CIN + 4; // LDS row stride of the staged input row: ds_read_b128 conflict-free (132*4 B = 16 mod 256)
constexpr int HD_NF = 6;          // fused filters
constexpr int HD_YS = 230;        // LDS row stride of Y: >= 224 and = 6 mod 32 -> the shift-add reads are conflict-free
constexpr int HD_RING = 16;       // open output rows (power of two >= 9)
constexpr int HD_MAXT = 37;       // taps: 37 * 6 = 222 <= 224 columns
constexpr int HD_SLOTS = 44;      // shift-add slots: 12 for dy = 0, 4 for each dy in -4..-1, 1..4

template <int NTILES>  // N tiles of 32 columns actually holding (tap,filter) pairs: ceil(ntaps*6/32) <= 7
__global__ __launch_bounds__(512, 2) void head6_kernel(HeadFusedArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* Xs = sm;                                   // [2][64][132]
  float* Ys = Xs + 2 * HD_W * HD_XS;                // [64][230]
  float* ring = Ys + HD_W * HD_YS;                  // [16][64][6]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 1, wg = wave >> 1;          // M tile (32 pixels), N group
  const int band = blockIdx.x % a.bands, img = blockIdx.x / a.bands;
  const int rows_per_band = a.H / a.bands;
  const int y0 = band * rows_per_band, y1 = y0 + rows_per_band;
  const int r_begin = max(0, y0 - a.halo), r_end = min(a.H, y1 + a.halo);
  const int N = a.ntaps * HD_NF;

  // ---- weights -> registers.  MFMA B operand of step (kk, c): lane l holds W[j = 32*nt + (l&31)][k = 8*kk + 4*(l>>5) + c]
  float wreg[2][64];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int nt = wg + 4 * i;
    const int j = nt * 32 + (lane & 31);
    const bool ok = (nt < NTILES) && (j < N);
    const int t = ok ? j / HD_NF : 0, f = ok ? j % HD_NF : 0;
    const float* wbase = (f >> 1) == 0 ? a.w0 : ((f >> 1) == 1 ? a.w1 : a.w2);
    const float* wp = wbase + (long long)t * a.w_tap_stride + (f & 1) * HD_CIN + (lane >> 5) * 4;
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ok) v = *reinterpret_cast<const float4*>(wp + kk * 8);
      wreg[i][kk * 4 + 0] = v.x; wreg[i][kk * 4 + 1] = v.y; wreg[i][kk * 4 + 2] = v.z; wreg[i][kk * 4 + 3] = v.w;
    }
  }
  for (int i = tid; i < HD_RING * HD_W * HD_NF; i += 512) ring[i] = 0.f;
  int slot_e[HD_SLOTS];   // tap id | (dx + 64) << 8, or -1: wave-uniform, kept in scalar registers for the whole kernel
#pragma unroll
  for (int t = 0; t < HD_SLOTS; ++t) slot_e[t] = a.itab[t];
  const int f_own = tid % HD_NF;
  const float sc_f = a.ftab[f_own], sh_f = a.ftab[8 + f_own];
  const int act_f = (int)a.ftab[16 + f_own];

  // ---- staging of one input row: 64 pixels x 128 channels = 2048 float4, 4 per thread, fully coalesced
  const float* xin = a.x + (size_t)img * a.H * HD_W * a.xs;
  float4 st0, st1, st2, st3;  // named registers (an indexed array captured by a lambda ended up in scratch)
  const int s_px = tid >> 5, s_c4 = (tid & 31) * 4;   // element e = tid + 512*q -> pixel s_px + 16*q, channel quad s_c4
#define HD_LOAD_ROW(r)                                                                                      \
  {                                                                                                        \
    const float* rp_ = xin + ((size_t)(r) * HD_W + s_px) * a.xs + s_c4;                                     \
    st0 = *reinterpret_cast<const float4*>(rp_);                                                           \
    st1 = *reinterpret_cast<const float4*>(rp_ + (size_t)16 * a.xs);                                       \
    st2 = *reinterpret_cast<const float4*>(rp_ + (size_t)32 * a.xs);                                       \
    st3 = *reinterpret_cast<const float4*>(rp_ + (size_t)48 * a.xs);                                       \
  }
#define HD_STORE_ROW(buf)                                                                                   \
  {                                                                                                        \
    float* sp_ = Xs + ((buf) * HD_W + s_px) * HD_XS + s_c4;                                                 \
    *reinterpret_cast<float4*>(sp_) = st0;                                                                 \
    *reinterpret_cast<float4*>(sp_ + 16 * HD_XS) = st1;                                                    \
    *reinterpret_cast<float4*>(sp_ + 32 * HD_XS) = st2;                                                    \
    *reinterpret_cast<float4*>(sp_ + 48 * HD_XS) = st3;                                                    \
  }
  HD_LOAD_ROW(r_begin);
  HD_STORE_ROW(0);
  __syncthreads();

  const float* a_base = Xs + (wm * 32 + (lane & 31)) * HD_XS + (lane >> 5) * 4;
  const int col_l = lane & 31, rhalf = 4 * (lane >> 5);
  float* out = a.out + (size_t)img * a.H * HD_W * 8;

  for (int r = r_begin; r < r_end; ++r) {
    const int buf = (r - r_begin) & 1;
    if (r + 1 < r_end) HD_LOAD_ROW(r + 1);   // in flight during the MFMAs
    __builtin_amdgcn_sched_barrier(0);
    // ---- Y tile(s) of this wave: 32 pixels x 32 columns each, K = 128
    hd_f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    const float* ap = a_base + buf * HD_W * HD_XS;
    const bool two = (wg + 4) < NTILES;  // wave-uniform
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      const float4 av = *reinterpret_cast<const float4*>(ap + kk * 8);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, wreg[0][kk * 4 + 0], acc[0], 0, 0, 0);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, wreg[0][kk * 4 + 1], acc[0], 0, 0, 0);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, wreg[0][kk * 4 + 2], acc[0], 0, 0, 0);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, wreg[0][kk * 4 + 3], acc[0], 0, 0, 0);
      if (two) {
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, wreg[1][kk * 4 + 0], acc[1], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, wreg[1][kk * 4 + 1], acc[1], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, wreg[1][kk * 4 + 2], acc[1], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, wreg[1][kk * 4 + 3], acc[1], 0, 0, 0);
      }
    }
    // ---- Y -> LDS.  C layout of the 32x32 MFMA: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (i == 1 && !two) break;
      const int jcol = (wg + 4 * i) * 32 + col_l;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int px = wm * 32 + (e & 3) + 8 * (e >> 2) + rhalf;
        Ys[px * HD_YS + jcol] = acc[i][e];
      }
    }
    if (r + 1 < r_end) HD_STORE_ROW(buf ^ 1);  // the other buffer: nobody reads it during this row
    __syncthreads();
    // ---- shift-add: out[y = r - dy][x][f] += sum over the taps with that dy of Y[x + dx][(t,f)].  Thread (x,f) owns its
    // ring column for the whole kernel.  The taps arrive grouped by dy in a STATIC slot layout (12 slots for dy = 0, 4 for
    // every other dy, empty = -1; wave-uniform, held in scalar registers), so the whole walk is straight-line code: all
    // reads of Y are issued back to back (no table-lookup -> load latency chains: the first version of this loop spent as
    // long here as in the MFMAs), the nine per-dy sums live in registers, and the order of additions is fixed.
    if (tid < HD_W * HD_NF) {
      const int f = tid % HD_NF, x = tid / HD_NF;
      const float* my_y = Ys + f;
      float sd[9];
#pragma unroll
      for (int g = 0; g < 9; ++g) {
        const int s0 = (g == 0) ? 0 : 12 + (g - 1) * 4, cnt = (g == 0) ? 12 : 4;
        float acc_g = 0.f;
#pragma unroll
        for (int q = 0; q < cnt; ++q) {
          const int e = slot_e[s0 + q];
          const int xx = x + (((e >> 8) & 0xFF) - 64);
          const bool ok = (e >= 0) && ((unsigned)xx < (unsigned)HD_W);
          const float v = my_y[min(max(xx, 0), HD_W - 1) * HD_YS + (e & 0xFF) * HD_NF];   // always a valid address
          acc_g += ok ? v : 0.f;
        }
        sd[g] = acc_g;
      }
#pragma unroll
      for (int g = 0; g < 9; ++g) {
        const int dy = (g == 0) ? 0 : (g <= 4 ? g - 5 : g - 4);
        const int y = r - dy;
        if (y >= y0 && y < y1) ring[((y & (HD_RING - 1)) * HD_W + x) * HD_NF + f] += sd[g];   // uniform branch
      }
      // output rows no later input row can reach are complete: y = r - halo, and everything still open after the
      // image's last row
      const int lo = max(y0, r - a.halo), hi = min(y1 - 1, (r == a.H - 1) ? y1 - 1 : r - a.halo);
      for (int yf = lo; yf <= hi; ++yf) {
        float* rp = ring + ((yf & (HD_RING - 1)) * HD_W + x) * HD_NF + f;
        out[((size_t)yf * HD_W + x) * 8 + f] = hd_act(*rp * sc_f + sh_f, act_f);
        *rp = 0.f;
      }
    }
    __syncthreads();  // Ys is rewritten by the next row
  }
}

#undef HD_LOAD_ROW
#undef HD_STORE_ROW

hipError_t launch_head6(const HeadFusedArgs& a, int n, hipStream_t s) {
  if (a.W != HD_W || a.ntaps > HD_MAXT || a.bands < 1 || (a.H % a.bands) || a.halo > 4 || a.xs < HD_CIN) return hipErrorInvalidValue;
  const size_t lds = (size_t)(2 * HD_W * HD_XS + HD_W * HD_YS + HD_RING * HD_W * HD_NF) * sizeof(float);
  const int ntiles = (a.ntaps * HD_NF + 31) / 32;
#define HD_LAUNCH(NT)                                                                                             \
  {                                                                                                               \
    static bool attr = false;                                                                                     \
    auto k = head6_kernel<NT>;                                                                                    \
    if (!attr) {                                                                                                  \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
      if (e != hipSuccess) return e;                                                                              \
      attr = true;                                                                                                \
    }                                                                                                             \
    hipLaunchKernelGGL(k, dim3(n * a.bands), dim3(512), lds, s, a);                                               \
    return hipGetLastError();                                                                                     \
  }
  if (ntiles <= 4) HD_LAUNCH(4)
  if (ntiles <= 6) HD_LAUNCH(6)
  HD_LAUNCH(7)
#undef HD_LAUNCH
}

// ------------------------------------------------------------------------------------------------
// head_tail_kernel: G = act_g((Ga + MDCL_Gb(R))*s+b), B = act_b((Ba + MDCL_Bb([R,G]))*s+b), beta x3 -> NCHW.  One workgroup
// (1024 threads) per image; the compact map of head6_kernel sits in LDS as [pixel][6] = (R0,R1,Ga0|G0,Ga1|G1,Ba0,Ba1).
// ------------------------------------------------------------------------------------------------
constexpr int HT_T = 1024;
__global__ __launch_bounds__(HT_T) void head_tail_kernel(HeadTailArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int HW = a.H * a.W;
  float* m = sm;                       // [HW][6]
  float* wg = m + HW * 6;              // [ntaps][2 out][2 in]
  float* wb = wg + a.ntaps * 4;        // [ntaps][2 out][4 in]
  const int img = blockIdx.x, tid = threadIdx.x;
  const float* src = a.comp + (size_t)img * HW * 8;
  for (int i = tid; i < HW * 2; i += HT_T) {   // 2 float4 per pixel -> 6 floats kept
    const int p = i >> 1, hlf = i & 1;
    const float4 v = *reinterpret_cast<const float4*>(src + (size_t)p * 8 + hlf * 4);
    if (hlf == 0) { m[p * 6 + 0] = v.x; m[p * 6 + 1] = v.y; m[p * 6 + 2] = v.z; m[p * 6 + 3] = v.w; }
    else { m[p * 6 + 4] = v.x; m[p * 6 + 5] = v.y; }
  }
  for (int i = tid; i < a.ntaps * 4; i += HT_T) {
    const int t = i >> 2, co = (i >> 1) & 1, ci = i & 1;
    wg[i] = a.w_gb[(long long)t * a.gb_tap_stride + co * a.gb_cin + ci];
  }
  for (int i = tid; i < a.ntaps * 8; i += HT_T) {
    const int t = i >> 3, co = (i >> 2) & 1, ci = i & 3;
    wb[i] = a.w_bb[(long long)t * a.bb_tap_stride + co * a.bb_cin + ci];
  }
  __syncthreads();
  // ---- G: reads R of the neighbourhood, its own Ga; written in place of Ga after everyone has read... Ga is only read by
  // its own pixel, so in-place is safe without a second buffer
  for (int p = tid; p < HW; p += HT_T) {
    const int py = p / a.W, px = p % a.W;
    float s0 = 0.f, s1 = 0.f;
    for (int t = 0; t < a.ntaps; ++t) {
      const int yy = py + a.dy[t], xx = px + a.dx[t];
      if ((unsigned)yy < (unsigned)a.H && (unsigned)xx < (unsigned)a.W) {
        const float2 r = *reinterpret_cast<const float2*>(m + (yy * a.W + xx) * 6);
        const float4 w = *reinterpret_cast<const float4*>(wg + t * 4);
        s0 = fmaf(r.x, w.x, s0); s0 = fmaf(r.y, w.y, s0);
        s1 = fmaf(r.x, w.z, s1); s1 = fmaf(r.y, w.w, s1);
      }
    }
    const float g0 = hd_act((s0 + m[p * 6 + 2]) * a.scale_g[0] + a.shift_g[0], a.act_g);
    const float g1 = hd_act((s1 + m[p * 6 + 3]) * a.scale_g[1] + a.shift_g[1], a.act_g);
    m[p * 6 + 2] = g0;
    m[p * 6 + 3] = g1;
  }
  __syncthreads();
  float* outp = a.out + (size_t)img * 3 * HW;
  for (int p = tid; p < HW; p += HT_T) {
    const int py = p / a.W, px = p % a.W;
    float s0 = 0.f, s1 = 0.f;
    for (int t = 0; t < a.ntaps; ++t) {
      const int yy = py + a.dy[t], xx = px + a.dx[t];
      if ((unsigned)yy < (unsigned)a.H && (unsigned)xx < (unsigned)a.W) {
        const float* q = m + (yy * a.W + xx) * 6;   // (R0,R1,G0,G1) = the ConcatLayer([R,G]) channel order (IAN.py:201)
        const float2 r = *reinterpret_cast<const float2*>(q);
        const float2 g = *reinterpret_cast<const float2*>(q + 2);
        const float4 w0 = *reinterpret_cast<const float4*>(wb + t * 8), w1 = *reinterpret_cast<const float4*>(wb + t * 8 + 4);
        s0 = fmaf(r.x, w0.x, s0); s0 = fmaf(r.y, w0.y, s0); s0 = fmaf(g.x, w0.z, s0); s0 = fmaf(g.y, w0.w, s0);
        s1 = fmaf(r.x, w1.x, s1); s1 = fmaf(r.y, w1.y, s1); s1 = fmaf(g.x, w1.z, s1); s1 = fmaf(g.y, w1.w, s1);
      }
    }
    const float b0 = hd_act((s0 + m[p * 6 + 4]) * a.scale_b[0] + a.shift_b[0], a.act_b);
    const float b1 = hd_act((s1 + m[p * 6 + 5]) * a.scale_b[1] + a.shift_b[1], a.act_b);
    const float r0 = m[p * 6 + 0], r1 = m[p * 6 + 1], g0 = m[p * 6 + 2], g1 = m[p * 6 + 3];
    // beta_layer (layers.py:397-408): 2*(alpha/(alpha+beta+1e-8)) - 1
    outp[p] = 2.f * (r0 / (r0 + r1 + 1e-8f)) - 1.f;
    outp[HW + p] = 2.f * (g0 / (g0 + g1 + 1e-8f)) - 1.f;
    outp[2 * HW + p] = 2.f * (b0 / (b0 + b1 + 1e-8f)) - 1.f;
  }
}

// compact map of head6_kernel -> three NHWC 2-channel maps (training: the backward pass and G_b / B_b want them apart)
__global__ __launch_bounds__(256) void head6_scatter_kernel(const float* __restrict__ comp, float* __restrict__ y0,
                                                            float* __restrict__ y1, float* __restrict__ y2, int ys, long long npix) {
  const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
  if (p >= npix) return;
  const float4 a = *reinterpret_cast<const float4*>(comp + p * 8);
  const float2 b = *reinterpret_cast<const float2*>(comp + p * 8 + 4);
  *reinterpret_cast<float2*>(y0 + p * ys) = make_float2(a.x, a.y);
  *reinterpret_cast<float2*>(y1 + p * ys) = make_float2(a.z, a.w);
  *reinterpret_cast<float2*>(y2 + p * ys) = b;
}
hipError_t launch_head6_scatter(const float* comp, float* y0, float* y1, float* y2, int ys, long long npix, hipStream_t s) {
  if (ys & 1) return hipErrorInvalidValue;
  hipLaunchKernelGGL(head6_scatter_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, s, comp, y0, y1, y2, ys, npix);
  return hipGetLastError();
}

// ---- backward-weight of the same three layers, contract-first as well -------------------------------------------------
//   dS_k[t][f][c] = sum_p dY_k[p][f] * X[p + d_t][c] = sum_q X[q][c] * Z[q][(t,k,f)],   Z[q][(t,k,f)] = dY_k[q - d_t][f]
// i.e. ONE dense backward-weight GEMM (198 x 128, contraction over all n*H*W pixels: the existing tapwgrad kernel on a
// helper dense layer) after a cheap shifted gather builds Z -- instead of three VALU passes of 1.04 ms each over X.
__global__ __launch_bounds__(256) void head6_zbuild_kernel(const float* __restrict__ dy0, const float* __restrict__ dy1,
                                                           const float* __restrict__ dy2, int dys, float* __restrict__ Z, int zs,
                                                           int H, int W, int ntaps, const int* __restrict__ taps, long long npix) {
  // one thread per (pixel, tap slot): three float2 gathers from the shifted pixel, six consecutive floats out; the slots
  // past the last tap write the zero padding columns (zs - 6*ntaps <= 6 * extra slots)
  const int slots = (zs + 5) / 6;
  const long long total = npix * slots;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long q = i / slots;
    const int t = (int)(i - q * slots);
    float v[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (t < ntaps) {
      const int qx = (int)(q % W), qy = (int)((q / W) % H);
      const int tp = taps[t];                       // (dy + 64) | (dx + 64) << 8
      const int sy = (tp & 0xFF) - 64, sx = (tp >> 8) - 64;
      const int py = qy - sy, px = qx - sx;
      if ((unsigned)py < (unsigned)H && (unsigned)px < (unsigned)W) {
        const long long src = (q - (long long)sy * W - sx) * dys;
        const float2 a = *reinterpret_cast<const float2*>(dy0 + src);
        const float2 b = *reinterpret_cast<const float2*>(dy1 + src);
        const float2 c = *reinterpret_cast<const float2*>(dy2 + src);
        v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y; v[4] = c.x; v[5] = c.y;
      }
    }
    float* dst = Z + q * zs + t * 6;
    const int room = zs - t * 6;                    // >= 1
    if (room >= 6) {
      *reinterpret_cast<float2*>(dst) = make_float2(v[0], v[1]);
      *reinterpret_cast<float2*>(dst + 2) = make_float2(v[2], v[3]);
      *reinterpret_cast<float2*>(dst + 4) = make_float2(v[4], v[5]);
    } else {
      for (int e = 0; e < room; ++e) dst[e] = v[e];
    }
  }
}
// Round 6: the same Z, built a pixel row at a time.  The per-(pixel, tap) form above issues three 8-byte gathers and three 8-byte
// stores 24 bytes apart per thread: 370 us for the 470 MB of Z at 128 images (1.3 TB/s), with no GEMM to hide under
// (profiles/r06_train_timeline_before.json).  Here a workgroup owns 64 consecutive pixels of one image row: it stages the
// (2 S + 1) source rows x (64 + 2 S) pixels x 6 values its taps can reach in LDS (S = the largest |shift|; out-of-image pixels
// as zeros), then writes its 64 rows of Z as fully coalesced float4 -- each value still one copy of one dY element (bitwise Z).
constexpr int ZB_MAXS = 8;
__global__ __launch_bounds__(256) void head6_zbuild_rows_kernel(const float* __restrict__ dy0, const float* __restrict__ dy1,
                                                                const float* __restrict__ dy2, int dys, float* __restrict__ Z, int zs,
                                                                int H, int W, int ntaps, const int* __restrict__ taps, int S) {
  extern __shared__ float zl[];                       // [(2S+1)][(64+2S)][6], then ntaps packed shifts
  const int PW = 64 + 2 * S, NR = 2 * S + 1;
  int* tsh = reinterpret_cast<int*>(zl + NR * PW * 6);
  const int segs = W / 64;
  const int seg = blockIdx.x % segs, qy = (blockIdx.x / segs) % H;
  const long long n = blockIdx.x / ((long long)segs * H);
  const int qx0 = seg * 64;
  for (int t = threadIdx.x; t < ntaps; t += 256) tsh[t] = taps[t];
  const float* maps[3] = {dy0, dy1, dy2};
  for (int i = threadIdx.x; i < NR * PW * 3; i += 256) {
    const int k = i % 3, px = (i / 3) % PW, r = i / (3 * PW);
    const int py = qy - (r - S), sx = qx0 + px - S;   // LDS row r holds source row qy - sy for sy = r - S
    float2 v = make_float2(0.f, 0.f);
    if ((unsigned)py < (unsigned)H && (unsigned)sx < (unsigned)W)
      v = *reinterpret_cast<const float2*>(maps[k] + ((n * H + py) * W + sx) * dys);
    float* q = zl + (r * PW + px) * 6 + 2 * k;
    q[0] = v.x; q[1] = v.y;
  }
  __syncthreads();
  const int z4 = zs >> 2;
  float* zrow = Z + ((n * H + qy) * W + qx0) * zs;
  for (int i = threadIdx.x; i < 64 * z4; i += 256) {
    const int p = i / z4, j0 = (i - p * z4) * 4;
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int j = j0 + e, t = j / 6;
      float val = 0.f;
      if (t < ntaps) {
        const int tp = tsh[t];
        const int sy = (tp & 0xFF) - 64, sx = (tp >> 8) - 64;
        val = zl[((sy + S) * PW + (p - sx + S)) * 6 + (j - 6 * t)];
      }
      v[e] = val;
    }
    *reinterpret_cast<float4*>(zrow + (size_t)p * zs + j0) = make_float4(v[0], v[1], v[2], v[3]);
  }
}
hipError_t launch_head6_zbuild(const float* dy0, const float* dy1, const float* dy2, int dys, float* Z, int zs, int H, int W,
                               int ntaps, const int* taps, long long npix, hipStream_t s, int max_shift) {
  if ((zs & 1) || (dys & 1) || zs < 6 * ntaps) return hipErrorInvalidValue;
  if (max_shift >= 0 && max_shift <= ZB_MAXS && (W % 64) == 0 && (zs % 4) == 0 && npix % ((long long)H * W) == 0) {
    const int S = max_shift;
    const size_t lds = ((size_t)(2 * S + 1) * (64 + 2 * S) * 6 + ntaps) * sizeof(float);
    hipLaunchKernelGGL(head6_zbuild_rows_kernel, dim3((unsigned)(npix / 64)), dim3(256), lds, s, dy0, dy1, dy2, dys, Z, zs, H, W, ntaps, taps, S);
    return hipGetLastError();
  }
  const long long total = npix * ((zs + 5) / 6);
  int blocks = (int)std::min<long long>((total + 255) / 256, 256 * 64);
  hipLaunchKernelGGL(head6_zbuild_kernel, dim3(blocks), dim3(256), 0, s, dy0, dy1, dy2, dys, Z, zs, H, W, ntaps, taps, npix);
  return hipGetLastError();
}
// dWref[c][j] (the helper dense layer's reference layout W(in = 128, out = 6*ntaps)) -> layer k's slab gradient
// dS[(t * f_rows + f) * f_cols + c], j = t*6 + 2k + f
__global__ __launch_bounds__(256) void head6_dS_scatter_kernel(const float* __restrict__ dWref, int nout, int kk, int ntaps,
                                                               float* __restrict__ dS, int f_rows, int f_cols) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= ntaps * 2 * 128) return;
  const int c = i % 128, f = (i / 128) & 1, t = i / 256;
  dS[((size_t)t * f_rows + f) * f_cols + c] = dWref[(size_t)c * nout + t * 6 + 2 * kk + f];
}
hipError_t launch_head6_dS_scatter(const float* dWref, int nout, int kk, int ntaps, float* dS, int f_rows, int f_cols, hipStream_t s) {
  hipLaunchKernelGGL(head6_dS_scatter_kernel, dim3((ntaps * 256 + 255) / 256), dim3(256), 0, s, dWref, nout, kk, ntaps, dS, f_rows, f_cols);
  return hipGetLastError();
}

// layer k's forward slab [t][f_rows][f_cols] -> the helper dense layer's reference weight Wcat[c][j], j = t*6 + 2k + f
__global__ __launch_bounds__(256) void head6_wcat_kernel(const float* __restrict__ slab, int f_rows, int f_cols, int kk, int ntaps,
                                                         float* __restrict__ wcat, int nout) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= ntaps * 2 * 128) return;
  const int c = i % 128, f = (i / 128) & 1, t = i / 256;
  wcat[(size_t)c * nout + t * 6 + 2 * kk + f] = slab[((size_t)t * f_rows + f) * f_cols + c];
}
hipError_t launch_head6_wcat(const float* slab, int f_rows, int f_cols, int kk, int ntaps, float* wcat, int nout, hipStream_t s) {
  hipLaunchKernelGGL(head6_wcat_kernel, dim3((ntaps * 256 + 255) / 256), dim3(256), 0, s, slab, f_rows, f_cols, kk, ntaps, wcat, nout);
  return hipGetLastError();
}

hipError_t launch_head_tail(const HeadTailArgs& a, int n, hipStream_t s) {
  if (a.ntaps > 48 || a.H * a.W > 4096) return hipErrorInvalidValue;
  const size_t lds = (size_t)(a.H * a.W * 6 + a.ntaps * 12) * sizeof(float);
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(head_tail_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)((4096 * 6 + 48 * 12) * sizeof(float)));
    if (e != hipSuccess) return e;
    attr = true;
  }
  hipLaunchKernelGGL(head_tail_kernel, dim3(n), dim3(HT_T), lds, s, a);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// dec_out of IAN_simple (IAN_simple.py:171-181): 5x5 stride-2 transposed conv 128 -> Cout <= 4 channels + tanh, NHWC in,
// NCHW image out -- the same "contract first, shift later" scheme as head6_kernel:
//     Y[q][(k,co)] = sum_c x[q][c] * W[k][co][c]            one dense 128 -> 25*Cout contraction per INPUT pixel (fp32 MFMA)
//     out[2q - 2 + k][co] += Y[q][(k,co)]                    the 25 taps scatter into the four output parities
// Round 1's VALU kernel ran at 11 % of the HBM roof (41 us at batch 64: LDS-issue bound, 16 workgroups per image tile).
// A workgroup walks pairs of input rows (M = 2 x 32 pixels); thread (co, ox) owns an output column of every open output row
// and gathers its taps from Y in straight-line code: for the row pair (r, r+1) the touched output rows are 2r-2 .. 2r+4, all
// indices static.  Output rows 2r, 2r+1 are complete after the pair and leave through the affine + activation epilogue.
// ------------------------------------------------------------------------------------------------
constexpr int DS_W = 32;            // input width; M per step = 2 rows x 32 pixels
constexpr int DS_RING = 8;

template <int COUT, bool BAL = false>
__global__ __launch_bounds__(512, 2) void deconv_small_kernel(DeconvSmallArgs a) {
  constexpr int N = 25 * COUT;
  constexpr int NTILES = (N + 31) / 32;          // 3 for Cout = 3, 4 for Cout = 4
  constexpr int YS = NTILES * 32 + 2;            // = 2 mod 32: even / odd output columns read disjoint bank parities
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* Xs = sm;                                // [64][132]   (single buffer: refilled between the barriers)
  float* Ys = Xs + 64 * HD_XS;                   // [64][YS]
  float* ring = Ys + 64 * YS;                    // [8][COUT][64]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 1, wg = wave >> 1;
  const int band = blockIdx.x % a.bands, img = blockIdx.x / a.bands;
  const int rows_per_band = a.H / a.bands;       // input rows per band (even)
  const int q0 = band * rows_per_band, q1 = q0 + rows_per_band;
  const int oy_lo = 2 * q0, oy_hi = 2 * q1;      // owned output rows
  // Input rows that reach the owned output rows: q0-1 .. q1 (output row oy collects input rows (oy-2)/2 .. (oy+2)/2), i.e.
  // ONE halo row per side.  The loop walks row PAIRS from r_begin (any parity): when the count is odd one more row is taken on
  // the side where the image has one; its contributions all fall outside [oy_lo, oy_hi) and are filtered below.  (Round 2
  // walked even-aligned pairs from q0-2 to q1+2: 6 pairs per 8-row band instead of 5 -- 17 % of the MFMA and load work.)
  int r_begin = max(0, q0 - 1), r_end = min(a.H, q1 + 1);
  if ((r_end - r_begin) & 1) {
    if (r_begin > 0) --r_begin;
    else ++r_end;                                                   // r_end <= a.H: a.H is even and r_begin == 0
  }

  // BAL (round 6, COUT = 3): the same contraction as 16 x 16 blocks of v_mfma_f32_16x16x4_f32.  The 32 x 32 form has 2 x 3 = 6 blocks
  // for 8 waves: six waves work, and since a workgroup's waves land on the SIMDs in the order 0, 2, 1, 3, 0, 2, 1, 3 two SIMDs carry
  // two of them and two carry one -- the matrix phase of a row pair lasts 2 x 64 MFMAs (8192 cycles) where the work is 6144 per
  // SIMD.  75 columns are five 16-wide blocks (the 32-wide tiling pads to 96), 64 pixels four: 20 blocks, three for each of waves
  // 0-3 and two for each of waves 4-7 = five per SIMD, 5120 cycles.  Wave w < 4 owns column block w for pixel blocks 0-2; wave
  // 4 + i owns (column block i, pixel block 3) and (column block 4, pixel block i).
  float wreg[64];
  const int l16 = lane & 15, kg = lane >> 4;
  if constexpr (BAL) {
    static_assert(COUT == 3, "the balanced 16x16 tiling is laid out for 75 columns");
#pragma unroll
    for (int sidx = 0; sidx < 2; ++sidx) {
      const int cb = sidx == 0 ? (wave & 3) : 4;
      const int j = cb * 16 + l16;
      const bool ok = (j < N) && (sidx == 0 || wave >= 4);
      const int k = ok ? j / COUT : 0, co = ok ? j % COUT : 0;
      const float* wp = a.w + ((size_t)k * 4 + co) * HD_CIN + kg * 4;
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok) v = *reinterpret_cast<const float4*>(wp + kk * 16);
        wreg[sidx * 32 + kk * 4 + 0] = v.x; wreg[sidx * 32 + kk * 4 + 1] = v.y; wreg[sidx * 32 + kk * 4 + 2] = v.z; wreg[sidx * 32 + kk * 4 + 3] = v.w;
      }
    }
  } else
  {
    const int j = wg * 32 + (lane & 31);
    const bool ok = (wg < NTILES) && (j < N);
    const int k = ok ? j / COUT : 0, co = ok ? j % COUT : 0;
    const float* wp = a.w + ((size_t)k * 4 + co) * HD_CIN + (lane >> 5) * 4;
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ok) v = *reinterpret_cast<const float4*>(wp + kk * 8);
      wreg[kk * 4 + 0] = v.x; wreg[kk * 4 + 1] = v.y; wreg[kk * 4 + 2] = v.z; wreg[kk * 4 + 3] = v.w;
    }
  }
  for (int i = tid; i < DS_RING * COUT * 64; i += 512) ring[i] = 0.f;

  const float* xin = a.x + (size_t)img * a.H * DS_W * a.xs;
  float4 st0, st1, st2, st3;
  const int s_px = tid >> 5, s_c4 = (tid & 31) * 4;   // pixel s_px + 16*q of the 64-pixel row pair (contiguous in memory)
#define DS_LOAD(r)                                                                   \
  {                                                                                  \
    const float* rp_ = xin + ((size_t)(r) * DS_W + s_px) * a.xs + s_c4;              \
    st0 = *reinterpret_cast<const float4*>(rp_);                                     \
    st1 = *reinterpret_cast<const float4*>(rp_ + (size_t)16 * a.xs);                 \
    st2 = *reinterpret_cast<const float4*>(rp_ + (size_t)32 * a.xs);                 \
    st3 = *reinterpret_cast<const float4*>(rp_ + (size_t)48 * a.xs);                 \
  }
#define DS_STORE()                                                                   \
  {                                                                                  \
    float* sp_ = Xs + s_px * HD_XS + s_c4;                                           \
    *reinterpret_cast<float4*>(sp_) = st0;                                           \
    *reinterpret_cast<float4*>(sp_ + 16 * HD_XS) = st1;                              \
    *reinterpret_cast<float4*>(sp_ + 32 * HD_XS) = st2;                              \
    *reinterpret_cast<float4*>(sp_ + 48 * HD_XS) = st3;                              \
  }
  DS_LOAD(r_begin);
  DS_STORE();
  __syncthreads();

  const float* a_base = Xs + (wm * 32 + (lane & 31)) * HD_XS + (lane >> 5) * 4;
  const int col_l = lane & 31, rhalf = 4 * (lane >> 5);
  const int OH = 2 * a.H, OW = 2 * DS_W;
  // owner thread: (co, ox); consecutive threads = consecutive ox (coalesced NCHW stores)
  const bool owner = tid < COUT * 64;
  const int o_co = owner ? tid >> 6 : 0, o_ox = tid & 63;
  const float sc = (owner && a.scale) ? a.scale[o_co] : 1.f, sh = (owner && a.shift) ? a.shift[o_co] : 0.f;
  const int par = o_ox & 1;                      // taps with kx = par, par + 2 (, 4)
  float* my_ring = ring + o_co * 64 + o_ox;      // + slot * COUT * 64
  float* outp = a.y + ((size_t)img * COUT + o_co) * OH * OW + o_ox;

  for (int r = r_begin; r < r_end; r += 2) {
    if (r + 2 < r_end) DS_LOAD(r + 2);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (BAL) {
      typedef float hd_f32x4 __attribute__((ext_vector_type(4)));
      // C layout of the 16 x 16 MFMA: col = lane & 15, row = 4 (lane >> 4) + i
      const int kq = kg * 4;
#define DS_M16(acc, av, wbase, kk)                                                                        \
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, wreg[(wbase) + (kk) * 4 + 0], acc, 0, 0, 0);         \
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, wreg[(wbase) + (kk) * 4 + 1], acc, 0, 0, 0);         \
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, wreg[(wbase) + (kk) * 4 + 2], acc, 0, 0, 0);         \
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, wreg[(wbase) + (kk) * 4 + 3], acc, 0, 0, 0);
      if (wave < 4) {                              // column block `wave`, pixel blocks 0, 1, 2 (weight set 0)
        hd_f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0, c2 = c0;
        const float* xb = Xs + l16 * HD_XS + kq;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
          const float4 a0 = *reinterpret_cast<const float4*>(xb + kk * 16);
          const float4 a1 = *reinterpret_cast<const float4*>(xb + 16 * HD_XS + kk * 16);
          const float4 a2 = *reinterpret_cast<const float4*>(xb + 32 * HD_XS + kk * 16);
          DS_M16(c0, a0, 0, kk)
          DS_M16(c1, a1, 0, kk)
          DS_M16(c2, a2, 0, kk)
        }
        float* yb = Ys + (kq) * YS + wave * 16 + l16;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          yb[i * YS] = c0[i];
          yb[(16 + i) * YS] = c1[i];
          yb[(32 + i) * YS] = c2[i];
        }
      } else {                                     // (column block i, pixel block 3) with set 0, (column block 4, pixel block i) with set 1
        const int i4 = wave & 3;
        hd_f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0;
        const float* xa = Xs + (48 + l16) * HD_XS + kq;
        const float* xb = Xs + (i4 * 16 + l16) * HD_XS + kq;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
          const float4 a0 = *reinterpret_cast<const float4*>(xa + kk * 16);
          const float4 a1 = *reinterpret_cast<const float4*>(xb + kk * 16);
          DS_M16(c0, a0, 0, kk)
          DS_M16(c1, a1, 32, kk)
        }
        float* ya = Ys + (48 + kq) * YS + i4 * 16 + l16;
        float* yb = Ys + (i4 * 16 + kq) * YS + 64 + l16;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          ya[i * YS] = c0[i];
          yb[i * YS] = c1[i];
        }
      }
#undef DS_M16
    } else
    if (wg < NTILES) {
      hd_f32x16 acc;
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) {
        const float4 av = *reinterpret_cast<const float4*>(a_base + kk * 8);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, wreg[kk * 4 + 0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, wreg[kk * 4 + 1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, wreg[kk * 4 + 2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, wreg[kk * 4 + 3], acc, 0, 0, 0);
      }
      const int jcol = wg * 32 + col_l;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int px = wm * 32 + (e & 3) + 8 * (e >> 2) + rhalf;
        Ys[px * YS + jcol] = acc[e];
      }
    }
    __syncthreads();                             // Y complete; every wave is done reading Xs
    if (r + 2 < r_end) DS_STORE();
    if (owner) {
      // taps of this thread's column parity: kx = par + 2*i, source column qx = (ox + 2 - kx) / 2 = (ox - par)/2 + 1 - i
      const int qxb = ((o_ox - par) >> 1) + 1;
      float s7[7];
#pragma unroll
      for (int d = 0; d < 7; ++d) s7[d] = 0.f;
#pragma unroll
      for (int ql = 0; ql < 2; ++ql)             // input row r + ql
#pragma unroll
        for (int ky = 0; ky < 5; ++ky) {         // output row 2(r+ql) - 2 + ky = (2r - 2) + (2ql + ky)
          float s = 0.f;
#pragma unroll
          for (int i = 0; i < 3; ++i) {
            const int kx = par + 2 * i;          // i = 2 exists for even columns only (kx = 4)
            const int qx = qxb - i;
            const bool ok = (kx < 5) && ((unsigned)qx < (unsigned)DS_W);
            const float v = Ys[(ql * DS_W + min(max(qx, 0), DS_W - 1)) * YS + (ky * 5 + min(kx, 4)) * COUT + o_co];
            s += ok ? v : 0.f;
          }
          s7[2 * ql + ky] += s;
        }
#pragma unroll
      for (int d = 0; d < 7; ++d) {
        const int oy = 2 * r - 2 + d;            // uniform
        if (oy >= oy_lo && oy < oy_hi) my_ring[(oy & (DS_RING - 1)) * COUT * 64] += s7[d];
      }
      // with input rows <= r+1 done, output rows <= 2r+1 are complete (even rows take their last tap from row oy/2 + 1):
      // 2r-2 .. 2r+1 are new; after the image's last pair also 2r+2, 2r+3
      const int f_lo = max(oy_lo, 2 * r - 2), f_hi = min(oy_hi - 1, (r + 2 >= a.H) ? 2 * r + 3 : 2 * r + 1);
      for (int oy = f_lo; oy <= f_hi; ++oy) {
        float* rp = my_ring + (oy & (DS_RING - 1)) * COUT * 64;
        outp[(size_t)oy * OW] = hd_act(*rp * sc + sh, a.act);
        *rp = 0.f;
      }
    }
    __syncthreads();                             // Xs refilled, Ys free
  }
#undef DS_LOAD
#undef DS_STORE
}

hipError_t launch_deconv_small(const DeconvSmallArgs& a, int n, int Cout, hipStream_t s) {
  if (a.W != DS_W || (a.H & 1) || a.bands < 1 || (a.H % (2 * a.bands)) || a.xs < HD_CIN || Cout < 3 || Cout > 4) return hipErrorInvalidValue;
  const int ntiles = (25 * Cout + 31) / 32, ys = ntiles * 32 + 2;
  const size_t lds = (size_t)(64 * HD_XS + 64 * ys + DS_RING * Cout * 64) * sizeof(float);
#define DS_LAUNCH(CO)                                                                                             \
  {                                                                                                               \
    static bool attr = false;                                                                                     \
    auto k = deconv_small_kernel<CO>;                                                                             \
    if (!attr) {                                                                                                  \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
      if (e != hipSuccess) return e;                                                                              \
      attr = true;                                                                                                \
    }                                                                                                             \
    hipLaunchKernelGGL(k, dim3(n * a.bands), dim3(512), lds, s, a);                                               \
    return hipGetLastError();                                                                                     \
  }
  if (Cout == 3 && a.balanced) {
    static bool attr = false;
    auto k = deconv_small_kernel<3, true>;
    if (!attr) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return e;
      attr = true;
    }
    hipLaunchKernelGGL(k, dim3(n * a.bands), dim3(512), lds, s, a);
    return hipGetLastError();
  }
  if (Cout == 3) DS_LAUNCH(3)
  DS_LAUNCH(4)
#undef DS_LAUNCH
}

}  // namespace ian

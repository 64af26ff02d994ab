// tapgemm: fp32 MFMA implicit-GEMM over a tap list -- the dominant kernel of the IAN hot path.
//
// Replaces the cuDNN calls behind Conv2DDNNLayer (IAN_simple.py:73-116), GpuDnnConvGradI
// (layers.py:476-481), the three conv branches of MDCL (layers.py:223-257) and DenseLayer.
// gfx950 design (not a CUDA tiling):
//   * exact-fp32 matrix cores: v_mfma_f32_32x32x2_f32 (64 cycles, 4096 FLOP) -- 157.3 TF chip peak;
//     bf16 MFMA would break the 1e-4 parity bar and gfx950 has no xf32;
//   * im2col-free: the A operand is gathered straight from the NHWC activation, one tap and one
//     32-channel chunk per K-step, 128 B contiguous per pixel row -> fully coalesced dwordx4 loads;
//   * 256 threads = 4 wave64; LDS tiles are [row][k] with a 36-float row stride so that both the
//     ds_write_b128 staging stores and the ds_read_b128 fragment loads are bank-conflict free
//     (ds_read_b128 banks are (addr/4)%64, serviced in 16-lane groups: 36*i mod 64 covers all banks);
//   * each lane reads 4 consecutive k per ds_read_b128 and feeds them to 4 MFMA k-steps: the k index
//     is permuted identically for A and B, so the contraction is unchanged;
//   * register-staged double buffering: global loads for K-step s+1 are issued before the MFMAs of
//     step s and written to the other LDS buffer afterwards -> one barrier per K-step;
//   * work comes from a host-built item table (class, tile, K-range): split-K and the 9/6/6/4-tap
//     imbalance of the transposed conv's parity classes are scheduled on the host, heavy items first,
//     groups that share a weight slab dealt to one XCD (block b runs on XCD b%8) for L2 reuse;
//   * epilogue fuses residual add, folded batch-norm scale/shift and the activation (forward), or
//     the activation derivative (backward-data chain of the latent brush, API.py:59,64).
#include "ian_internal.h"

namespace ian {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float apply_act(float v, int act) {
  switch (act) {
    case 1: return v > 0.f ? v : 0.f;
    case 2: return v > 0.f ? v : 0.2f * v;
    case 3: return v > 0.f ? v : expm1f(v);
    case 4: return tanhf(v);
    case 5: return 1.f / (1.f + __expf(-v));
    default: return v;
  }
}
// derivative of the activation expressed through its OUTPUT y (all six are invertible that way)
__device__ __forceinline__ float act_grad_from_out(float y, int act) {
  switch (act) {
    case 1: return y > 0.f ? 1.f : 0.f;
    case 2: return y > 0.f ? 1.f : 0.2f;
    case 3: return y > 0.f ? 1.f : y + 1.f;
    case 4: return 1.f - y * y;
    case 5: return y * (1.f - y);
    default: return 1.f;
  }
}

__device__ __forceinline__ float epilogue_value(const TgEpilogue& e, float acc, size_t yoff, int c) {
  if (e.res) acc += e.res[yoff];
  const int si = e.scale_period ? (int)(yoff % (size_t)e.scale_period) : c;
  const float sc = e.scale ? e.scale[si] : 1.f;
  if (e.mode == TG_EPI_FWD) {
    const float sh = e.shift ? e.shift[si] : 0.f;
    return apply_act(acc * sc + sh, e.act);
  }
  const float yf = e.yfwd ? e.yfwd[yoff] : 0.f;
  return acc * act_grad_from_out(yf, e.act) * sc;
}

constexpr int TG_BK = 32;
constexpr int TG_LDS = 36;  // row stride in floats (32 + 4 pad)

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void tapgemm_kernel(const TgParams p) {
  constexpr int FM = BM / WM / 32, FN = BN / WN / 32;
  constexpr int A_CH = BM / 32, B_CH = BN / 32;
  static_assert(WM * WN == 4, "4 waves");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                     // [2][BM][36]
  float* Bs = smem + 2 * BM * TG_LDS;   // [2][BN][36]

  const TgItem it = p.items[blockIdx.x];
  if (it.ks0 >= it.ks1) return;  // padding item
  const TgClass cl = p.classes[it.cls];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;

  // ---- staging assignment: thread -> (16-byte chunk c4 of the 128-byte K row, rows r0+32j) ----
  const int c4 = (tid & 7) * 4;
  const int r0 = tid >> 3;
  int a_iy0[A_CH], a_ix0[A_CH], a_off[A_CH];
  const int qhw_mask = (1 << p.qhw_shift) - 1, qw_mask = (1 << p.qw_shift) - 1;
#pragma unroll
  for (int j = 0; j < A_CH; ++j) {
    const int m = it.m0 + r0 + 32 * j;
    const int n = m >> p.qhw_shift;
    const int rem = m & qhw_mask;
    const int qy = rem >> p.qw_shift, qx = rem & qw_mask;
    const int iy0 = qy * p.si + p.by, ix0 = qx * p.si + p.bx;
    a_iy0[j] = (m < p.M) ? iy0 : -100000;  // invalid rows fail every bounds test
    a_ix0[j] = ix0;
    a_off[j] = ((n * p.IH + iy0) * p.IW + ix0) * p.Cin + c4;
  }
  const int kpt = p.Cin >> 5;  // K-steps per tap
  const size_t slab_stride = (size_t)p.CoutPad * p.Cin;
  const float* wrow = p.w + cl.w_off + (size_t)(it.n0 + r0) * p.Cin + c4;

  int tap = it.ks0 / kpt;
  int cstep = it.ks0 - tap * kpt;

  float4 ra[A_CH], rb[B_CH];
  auto load_tile = [&]() {
    const TgTap tp = p.taps[cl.tap0 + tap];
    const int ci0 = cstep << 5;
    const int doff = (tp.dy * p.IW + tp.dx) * p.Cin + ci0;
#pragma unroll
    for (int j = 0; j < A_CH; ++j) {
      const int iy = a_iy0[j] + tp.dy, ix = a_ix0[j] + tp.dx;
      const bool ok = ((unsigned)iy < (unsigned)p.IH) & ((unsigned)ix < (unsigned)p.IW);
      ra[j] = ok ? *reinterpret_cast<const float4*>(p.x + (a_off[j] + doff)) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const float* wb = wrow + (size_t)tap * slab_stride + ci0;
#pragma unroll
    for (int j = 0; j < B_CH; ++j) rb[j] = *reinterpret_cast<const float4*>(wb + (size_t)(32 * j) * p.Cin);
    if (++cstep == kpt) {
      cstep = 0;
      ++tap;
    }
  };
  auto store_tile = [&](int buf) {
    float* a = As + buf * BM * TG_LDS + r0 * TG_LDS + c4;
    float* b = Bs + buf * BN * TG_LDS + r0 * TG_LDS + c4;
#pragma unroll
    for (int j = 0; j < A_CH; ++j) *reinterpret_cast<float4*>(a + 32 * j * TG_LDS) = ra[j];
#pragma unroll
    for (int j = 0; j < B_CH; ++j) *reinterpret_cast<float4*>(b + 32 * j * TG_LDS) = rb[j];
  };

  f32x16 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int arow = wm * (BM / WM) + (lane & 31);
  const int brow = wn * (BN / WN) + (lane & 31);
  const int koff = (lane >> 5) * 4;

  load_tile();
  store_tile(0);
  __syncthreads();

  const int nks = it.ks1 - it.ks0;
  for (int s = 0; s < nks; ++s) {
    const int cur = s & 1;
    const bool more = (s + 1 < nks);
    if (more) load_tile();
    const float* a_s = As + cur * BM * TG_LDS + arow * TG_LDS + koff;
    const float* b_s = Bs + cur * BN * TG_LDS + brow * TG_LDS + koff;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      float4 av[FM], bv[FN];
#pragma unroll
      for (int i = 0; i < FM; ++i) av[i] = *reinterpret_cast<const float4*>(a_s + i * 32 * TG_LDS + kk * 8);
#pragma unroll
      for (int j = 0; j < FN; ++j) bv[j] = *reinterpret_cast<const float4*>(b_s + j * 32 * TG_LDS + kk * 8);
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].x, bv[j].x, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].y, bv[j].y, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].z, bv[j].z, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].w, bv[j].w, acc[i][j], 0, 0, 0);
        }
    }
    if (more) store_tile(cur ^ 1);
    __syncthreads();
  }

  // ---- epilogue. C layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5) ----
  const int col_l = lane & 31;
  const int rhalf = 4 * (lane >> 5);
  if (it.slab >= 0) {
    float* sl = p.slab + (size_t)it.slab * (BM * BN);
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = wm * (BM / WM) + i * 32 + (r & 3) + 8 * (r >> 2) + rhalf;
          const int col = wn * (BN / WN) + j * 32 + col_l;
          sl[row * BN + col] = acc[i][j][r];
        }
    return;
  }
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = wm * (BM / WM) + i * 32 + (r & 3) + 8 * (r >> 2) + rhalf;
      const int m = it.m0 + row;
      if (m >= p.M) continue;
      const int n = m >> p.qhw_shift;
      const int rem = m & qhw_mask;
      const int oy = (rem >> p.qw_shift) * p.so + cl.py, ox = (rem & qw_mask) * p.so + cl.px;
      const size_t pix = ((size_t)n * p.OH + oy) * p.OW + ox;
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int c = it.n0 + wn * (BN / WN) + j * 32 + col_l;
        if (c < p.Cout) {
          const size_t yoff = pix * p.y_stride + c;
          p.y[yoff] = epilogue_value(p.epi, acc[i][j][r], yoff, c);
        }
      }
    }
}

// split-K second pass: y = epilogue(sum of slabs); one block per output tile, float4 along channels
template <int BM, int BN>
__global__ __launch_bounds__(256) void tapgemm_reduce_kernel(const TgReduceParams p) {
  const TgTile t = p.tiles[blockIdx.x];
  const TgClass cl = p.classes[t.cls];
  constexpr int CG = BN / 4;          // float4 groups per row
  constexpr int RPI = 256 / CG;       // rows per iteration
  const int cg = threadIdx.x % CG, rr = threadIdx.x / CG;
  const int qhw_mask = (1 << p.qhw_shift) - 1, qw_mask = (1 << p.qw_shift) - 1;
  const float* base = p.slab + (size_t)t.slab0 * (BM * BN);
  for (int row = rr; row < BM; row += RPI) {
    const int m = t.m0 + row;
    if (m >= p.M) break;
    float4 s = *reinterpret_cast<const float4*>(base + row * BN + cg * 4);
    for (int k = 1; k < t.nsplit; ++k) {
      const float4 v = *reinterpret_cast<const float4*>(base + (size_t)k * (BM * BN) + row * BN + cg * 4);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    const int n = m >> p.qhw_shift;
    const int rem = m & qhw_mask;
    const int oy = (rem >> p.qw_shift) * p.so + cl.py, ox = (rem & qw_mask) * p.so + cl.px;
    const size_t pix = ((size_t)n * p.OH + oy) * p.OW + ox;
    const int c = t.n0 + cg * 4;
    const size_t yoff = pix * p.y_stride + c;
    const float v[4] = {s.x, s.y, s.z, s.w};
    if (c + 3 < p.Cout) {
      float4 o;
      o.x = epilogue_value(p.epi, v[0], yoff + 0, c + 0);
      o.y = epilogue_value(p.epi, v[1], yoff + 1, c + 1);
      o.z = epilogue_value(p.epi, v[2], yoff + 2, c + 2);
      o.w = epilogue_value(p.epi, v[3], yoff + 3, c + 3);
      *reinterpret_cast<float4*>(p.y + yoff) = o;
    } else {
      for (int e = 0; e < 4; ++e)
        if (c + e < p.Cout) p.y[yoff + e] = epilogue_value(p.epi, v[e], yoff + e, c + e);
    }
  }
}

template <int BM, int BN, int WM, int WN>
static hipError_t launch_cfg(const TgParams& p, int nitems, hipStream_t s) {
  static bool attr_set = false;
  const size_t lds = (size_t)2 * (BM + BN) * TG_LDS * sizeof(float);
  auto k = tapgemm_kernel<BM, BN, WM, WN>;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  hipLaunchKernelGGL(k, dim3(nitems), dim3(256), lds, s, p);
  return hipGetLastError();
}

hipError_t launch_tapgemm(int cfg, const TgParams& p, int nitems, hipStream_t s) {
  if (nitems <= 0) return hipSuccess;
  switch (cfg) {
    case TG_128x128: return launch_cfg<128, 128, 2, 2>(p, nitems, s);
    case TG_128x64: return launch_cfg<128, 64, 2, 2>(p, nitems, s);
    case TG_64x64: return launch_cfg<64, 64, 2, 2>(p, nitems, s);
    case TG_32x128: return launch_cfg<32, 128, 1, 4>(p, nitems, s);
    case TG_256x128: return launch_cfg<256, 128, 2, 2>(p, nitems, s);
    case TG_128x32: return launch_cfg<128, 32, 4, 1>(p, nitems, s);
  }
  return hipErrorInvalidValue;
}

hipError_t launch_tapgemm_reduce(int cfg, const TgReduceParams& p, int ntiles, hipStream_t s) {
  if (ntiles <= 0) return hipSuccess;
  switch (cfg) {
    case TG_128x128: hipLaunchKernelGGL((tapgemm_reduce_kernel<128, 128>), dim3(ntiles), dim3(256), 0, s, p); break;
    case TG_128x64: hipLaunchKernelGGL((tapgemm_reduce_kernel<128, 64>), dim3(ntiles), dim3(256), 0, s, p); break;
    case TG_64x64: hipLaunchKernelGGL((tapgemm_reduce_kernel<64, 64>), dim3(ntiles), dim3(256), 0, s, p); break;
    case TG_32x128: hipLaunchKernelGGL((tapgemm_reduce_kernel<32, 128>), dim3(ntiles), dim3(256), 0, s, p); break;
    case TG_256x128: hipLaunchKernelGGL((tapgemm_reduce_kernel<256, 128>), dim3(ntiles), dim3(256), 0, s, p); break;
    case TG_128x32: hipLaunchKernelGGL((tapgemm_reduce_kernel<128, 32>), dim3(ntiles), dim3(256), 0, s, p); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace ian

// tapwgrad: backward-weight of every tap-GEMM layer (5x5/s2 conv, 5x5/s2 transposed conv, composite MDC stencil,
// dense) for the train_IAN.py step (train_IAN.py:253-273: T.grad wrt the conv / deconv / dense / MDCL weights,
// which Theano lowers to cuDNN GpuDnnConvGradW / GEMM calls).
//
//   dS[t][co][ci] = sum_m dY[out_pixel(m)][co] * X[in_pixel(m) + d_t][ci]        (fwd slab layout [tap][CoutPad][CinPad])
//
// Same geometry tables (classes, taps) as the forward tapgemm; here the pixel index m is the CONTRACTION index:
//   * both operands are read exactly as they lie in HBM (NHWC rows, channels contiguous): 16 B/lane coalesced
//     buffer loads, halo / ragged rows / channel-tile overhang masked by an out-of-range offset (hardware zero fill);
//   * LDS tiles are [m][channel]; an MFMA operand fragment is 32 consecutive channels of one m row -> 32 consecutive
//     floats per half-wave: conflict-free ds_read_b32, no transposition anywhere;
//   * v_mfma_f32_32x32x2_f32 (exact fp32), two m rows per instruction, accumulators in registers over the item's
//     whole m range; the host splits m so that taps x channel tiles x splits fills the chip; each split writes a
//     partial slab, summed (and scattered to the reference parameter layout) by wgrad_reduce_kernel in a fixed order
//     -> bitwise reproducible gradients.
#include "ian_internal.h"

namespace ian {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float4 wg_load4(__amdgpu_buffer_rsrc_t rsrc, unsigned voff) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 0);
  float4 r;
  r.x = __uint_as_float(v.x);
  r.y = __uint_as_float(v.y);
  r.z = __uint_as_float(v.z);
  r.w = __uint_as_float(v.w);
  return r;
}

constexpr int WG_BK = 32;  // m rows per K-step

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN) void tapwgrad_kernel(const WgParams p) {
  constexpr int FM = BM / WM / 32, FN = BN / WN / 32;
  // 4 waves, or 8 waves for the 128x128 tile (2 x 4 waves of 64x32: half the accumulator registers per wave, two
  // workgroups = 16 waves per CU -- the same occupancy argument as tapgemm's 8-wave tile)
  constexpr int NT = 64 * WM * WN;
  constexpr int A_TPR = BM / 4, B_TPR = BN / 4;          // threads per tile row (float4 each)
  constexpr int A_RPP = NT / A_TPR, B_RPP = NT / B_TPR;  // rows per pass
  constexpr int A_CH = (WG_BK + A_RPP - 1) / A_RPP, B_CH = (WG_BK + B_RPP - 1) / B_RPP;
  static_assert(WM * WN == 4 || WM * WN == 8, "4 or 8 waves");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                      // [2][32][BM]
  float* Bs = smem + 2 * WG_BK * BM;     // [2][32][BN]

  const WgItem it = p.items[blockIdx.x];
  if (it.m0 >= it.m1) return;
  const TgClass cl = p.classes[it.cls];
  const TgTap tp = p.taps[it.tap];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dy), 0, p.dy_bytes, 0x00020000);

  const int a_c = (tid % A_TPR) * 4, a_r = tid / A_TPR;
  const int b_c = (tid % B_TPR) * 4, b_r = tid / B_TPR;
  const bool a_col_ok = (it.co0 + a_c) < p.dy_stride;  // channel-tile overhang reads as zero
  const bool b_col_ok = (it.ci0 + b_c) < p.Cin;
  const int qhw_mask = (1 << p.qhw_shift) - 1, qw_mask = (1 << p.qw_shift) - 1;

  float4 ra[A_CH], rb[B_CH];
  f32x16 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  auto load_tiles = [&](int mbase) {
#pragma unroll
    for (int j = 0; j < A_CH; ++j) {
      const int row = a_r + j * A_RPP;
      const int m = mbase + row;
      const int n = m >> p.qhw_shift, rem = m & qhw_mask;
      const int oy = (rem >> p.qw_shift) * p.so + cl.py, ox = (rem & qw_mask) * p.so + cl.px;
      const bool ok = a_col_ok && (row < WG_BK) && (m < it.m1);
      const unsigned off = (unsigned)((((n * p.OH + oy) * p.OW + ox) * p.dy_stride + it.co0 + a_c) * 4);
      ra[j] = wg_load4(yrsrc, ok ? off : 0xFFFFFFF0u);
    }
#pragma unroll
    for (int j = 0; j < B_CH; ++j) {
      const int row = b_r + j * B_RPP;
      const int m = mbase + row;
      const int n = m >> p.qhw_shift, rem = m & qhw_mask;
      const int iy = (rem >> p.qw_shift) * p.si + p.by + tp.dy, ix = (rem & qw_mask) * p.si + p.bx + tp.dx;
      const bool ok = b_col_ok && (row < WG_BK) && (m < it.m1) && ((unsigned)iy < (unsigned)p.IH) && ((unsigned)ix < (unsigned)p.IW);
      const unsigned off = (unsigned)((((n * p.IH + iy) * p.IW + ix) * p.Cin + it.ci0 + b_c) * 4);
      rb[j] = wg_load4(xrsrc, ok ? off : 0xFFFFFFF0u);
    }
  };
  auto store_tiles = [&](int buf) {
#pragma unroll
    for (int j = 0; j < A_CH; ++j) {
      const int row = a_r + j * A_RPP;
      if (row < WG_BK) *reinterpret_cast<float4*>(As + (buf * WG_BK + row) * BM + a_c) = ra[j];
    }
#pragma unroll
    for (int j = 0; j < B_CH; ++j) {
      const int row = b_r + j * B_RPP;
      if (row < WG_BK) *reinterpret_cast<float4*>(Bs + (buf * WG_BK + row) * BN + b_c) = rb[j];
    }
  };

  const int half = lane >> 5, l31 = lane & 31;
  const float* a_base = As + half * BM + wm * (BM / WM) + l31;
  const float* b_base = Bs + half * BN + wn * (BN / WN) + l31;
  auto compute = [&](int buf) {
    const float* a_s = a_base + buf * WG_BK * BM;
    const float* b_s = b_base + buf * WG_BK * BN;
#pragma unroll
    for (int kk = 0; kk < WG_BK / 2; ++kk) {
      float av[FM], bv[FN];
#pragma unroll
      for (int i = 0; i < FM; ++i) av[i] = a_s[kk * 2 * BM + i * 32];
#pragma unroll
      for (int j = 0; j < FN; ++j) bv[j] = b_s[kk * 2 * BN + j * 32];
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
    }
  };

  load_tiles(it.m0);
  store_tiles(0);
  __syncthreads();
  int cur = 0;
  for (int mb = it.m0 + WG_BK; mb < it.m1; mb += WG_BK) {
    load_tiles(mb);
    __builtin_amdgcn_sched_barrier(0);
    compute(cur);
    store_tiles(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }
  compute(cur);

  // partial[split][tap][co][ci]; C layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  float* out = p.partial + (size_t)it.split * p.slab_total + (size_t)it.tap * p.CoutPad * p.CinPad;
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = it.co0 + wm * (BM / WM) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (co >= p.CoutPad) continue;
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int ci = it.ci0 + wn * (BN / WN) + j * 32 + l31;
        if (ci < p.CinPad) out[(size_t)co * p.CinPad + ci] = acc[i][j][r];
      }
    }
}


// ---- tapwgrad_p_kernel: the 8-wave 128x128 tile with a software-pipelined K loop (round 6) -----------------------------------
// Same tile, same operands, same summation order per output element as tapwgrad_kernel<128,128,2,4> (bitwise the same partial
// slabs); what changes is the schedule.  The compiler-scheduled loop above issues each LDS fragment read right before the two
// MFMAs of the PREVIOUS k pair (128 cycles of cover for a 64-128+ cycle LDS round trip) and waits for the global loads of the next
// tile at the bottom of the step: profiles/r06_wgrad.md shows 31 % of wave cycles parked (tapgemm's production schedule: 15 %) and
// the matrix pipe 65-75 % busy.  Here, as in tapgemm's VAR 2:
//   * fragments travel in GROUPS of four k pairs (8 MFMAs = 512 cycles per group), double-buffered in registers: the reads of
//     group g+1 are in flight during the MFMAs of group g;
//   * the A operand (64 output channels of the wave) is ONE ds_read_b64 per k pair: a lane takes channels 2l, 2l+1 of its k row,
//     .x feeds the MFMA block of the EVEN channels, .y the block of the ODD ones (a permutation of which accumulator holds which
//     row; 256 B per 32 lanes, conflict free) -- half the LDS instructions of two ds_read_b32;
//   * the global loads of K-step s+1 are issued at the top of step s, their ds_write_b128 after the second group (>= 1000 cycles
//     later), one barrier per step, and the last group of a step is computed AFTER the barrier, covering it and the first reads
//     of the new buffer;
//   * the bounds tests of the loads are branch-free (the short-circuit && of the old loop compiled to exec-mask branches).
template <int SCHED>
__global__ __launch_bounds__(512, 2) void tapwgrad_p_kernel(const WgParams p) {
  constexpr int BM = 128, BN = 128, WN = 4;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                      // [2][32][BM]
  float* Bs = smem + 2 * WG_BK * BM;     // [2][32][BN]

  const WgItem it = p.items[blockIdx.x];
  if (it.m0 >= it.m1) return;
  const TgClass cl = p.classes[it.cls];
  const TgTap tp = p.taps[it.tap];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dy), 0, p.dy_bytes, 0x00020000);

  // staging: thread -> (row s_r + 16 j of the 32-row K-step, 16-byte chunk s_c of the 128-channel row), the same for both operands
  const int s_c = (tid & 31) * 4, s_r = tid >> 5;
  const bool a_col_ok = (it.co0 + s_c) < p.dy_stride;  // channel-tile overhang reads as zero
  const bool b_col_ok = (it.ci0 + s_c) < p.Cin;
  const int qhw_mask = (1 << p.qhw_shift) - 1, qw_mask = (1 << p.qw_shift) - 1;
  const int a_cbase = it.co0 + s_c, b_cbase = it.ci0 + s_c;
  const int y0 = cl.py, x0 = cl.px, by = p.by + tp.dy, bx = p.bx + tp.dx;

  float4 ra[2], rb[2];
  f32x16 acc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  int mbase = it.m0;
#define WGP_LOAD_INTO(RA, RB)                                                                                   \
  {                                                                                                             \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                             \
      const int m = mbase + s_r + 16 * j;                                                                       \
      const int n = m >> p.qhw_shift, rem = m & qhw_mask;                                                       \
      const int qy = rem >> p.qw_shift, qx = rem & qw_mask;                                                     \
      const bool in = m < it.m1;                                                                                \
      const unsigned aoff = (unsigned)((((n * p.OH + qy * p.so + y0) * p.OW + qx * p.so + x0) * p.dy_stride + a_cbase) * 4); \
      RA[j] = wg_load4(yrsrc, (a_col_ok & in) ? aoff : 0xFFFFFFF0u);                                            \
      const int iy = qy * p.si + by, ix = qx * p.si + bx;                                                       \
      const bool okb = b_col_ok & in & ((unsigned)iy < (unsigned)p.IH) & ((unsigned)ix < (unsigned)p.IW);       \
      const unsigned boff = (unsigned)((((n * p.IH + iy) * p.IW + ix) * p.Cin + b_cbase) * 4);                  \
      RB[j] = wg_load4(xrsrc, okb ? boff : 0xFFFFFFF0u);                                                        \
    }                                                                                                           \
    mbase += WG_BK;                                                                                             \
  }
#define WGP_LOAD() WGP_LOAD_INTO(ra, rb)
#define WGP_STORE_FROM(RA, RB, buf)                                                                             \
  {                                                                                                             \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                             \
      *reinterpret_cast<float4*>(As + ((buf) * WG_BK + s_r + 16 * j) * BM + s_c) = RA[j];                       \
      *reinterpret_cast<float4*>(Bs + ((buf) * WG_BK + s_r + 16 * j) * BN + s_c) = RB[j];                       \
    }                                                                                                           \
  }
#define WGP_STORE(buf) WGP_STORE_FROM(ra, rb, buf)
  // SCHED 3's loads (round 6): the pixel index of a tile row is m = mbase + r with mbase a multiple of 32 and r < 32, and the map extents are
  // powers of two -- so image, row and column of m are the SUMS of their parts in mbase (wave-uniform: scalar arithmetic) and in r (a
  // thread constant), without carries, and every offset is linear in them.  The old form rebuilt both offsets from m with six integer
  // multiplies (quarter rate) and ~20 further vector instructions per row and K-step; a vector instruction of any wave takes its issue
  // cycles from the SIMD's matrix pipe (tapgemm schedule 7 measured 5.5 % of a whole step for a third of this arithmetic).  Now per row and
  // step: the bounds tests against the uniform part (adds + compares, no multiply) and one add per offset.
  unsigned a_vt[2], b_vt[2];
  int t_y[2], t_x[2], t_r[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int r = s_r + 16 * j;
    const int n_t = r >> p.qhw_shift, rem_t = r & qhw_mask;
    const int qy_t = rem_t >> p.qw_shift, qx_t = rem_t & qw_mask;
    t_r[j] = r;
    t_y[j] = qy_t * p.si + by;
    t_x[j] = qx_t * p.si + bx;
    a_vt[j] = (unsigned)((((n_t * p.OH + qy_t * p.so + y0) * p.OW + qx_t * p.so + x0) * p.dy_stride + a_cbase) * 4);
    b_vt[j] = (unsigned)((((n_t * p.IH + t_y[j]) * p.IW + t_x[j]) * p.Cin + b_cbase) * 4);
  }
#define WGP_LOADU_INTO(RA, RB)                                                                                  \
  {                                                                                                             \
    const int n_u = mbase >> p.qhw_shift, rem_u = mbase & qhw_mask;                                             \
    const int qy_u = rem_u >> p.qw_shift, qx_u = rem_u & qw_mask;                                               \
    const unsigned a_su = (unsigned)((((n_u * p.OH + qy_u * p.so) * p.OW + qx_u * p.so) * p.dy_stride) * 4);    \
    const unsigned b_su = (unsigned)((((n_u * p.IH + qy_u * p.si) * p.IW + qx_u * p.si) * p.Cin) * 4);          \
    const int uy = qy_u * p.si, ux = qx_u * p.si, lim = it.m1 - mbase;                                          \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                             \
      const bool in = t_r[j] < lim;                                                                             \
      RA[j] = wg_load4(yrsrc, (a_col_ok & in) ? a_vt[j] + a_su : 0xFFFFFFF0u);                                  \
      const bool okb = b_col_ok & in & ((unsigned)(t_y[j] + uy) < (unsigned)p.IH) & ((unsigned)(t_x[j] + ux) < (unsigned)p.IW); \
      RB[j] = wg_load4(xrsrc, okb ? b_vt[j] + b_su : 0xFFFFFFF0u);                                              \
    }                                                                                                           \
    mbase += WG_BK;                                                                                             \
  }

  const int half = lane >> 5, l31 = lane & 31;
  const float* a_base = As + half * BM + wm * 64 + 2 * l31;   // float2: channels 2 l31, 2 l31 + 1 of the wave's 64
  const float* b_base = Bs + half * BN + wn * 32 + l31;
  // group g of a K-step = k pairs 4g .. 4g+3 = LDS rows 8g + 2q + half
  auto frag_load = [&](int buf, int g, float2 (&av)[4], float (&bv)[4]) {
    const float* a_s = a_base + (buf * WG_BK + 8 * g) * BM;
    const float* b_s = b_base + (buf * WG_BK + 8 * g) * BN;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      av[q] = *reinterpret_cast<const float2*>(a_s + 2 * q * BM);
      bv[q] = b_s[2 * q * BN];
    }
  };
  auto frag_mfma = [&](const float2 (&av)[4], const float (&bv)[4]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q].x, bv[q], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q].y, bv[q], acc[1], 0, 0, 0);
    }
  };

  const int nks = (it.m1 - it.m0 + WG_BK - 1) / WG_BK;
#define WGP_SB() __builtin_amdgcn_sched_barrier(0)
  // Every fragment read is PINNED one whole group (8 MFMAs, 512 cycles) ahead of its use: left alone, hipcc sinks the reads next to
  // their MFMAs (ds_read, s_waitcnt lgkmcnt(0), v_mfma) to shorten live ranges.
  if constexpr (SCHED == 1) {
    // two fragment buffers; the reads of the new buffer's group 0 cross the loop's back edge -- and hipcc's waitcnt pass then
    // waits with lgkmcnt(0) at the top of the body, i.e. also for the group-1 reads issued just before (kept for the A/B)
    float2 av[4], aw[4];
    float bv[4], bw[4];
    WGP_LOAD();
    WGP_STORE(0);
    __syncthreads();
    int cur = 0;
    frag_load(0, 0, av, bv);
    for (int s = 0; s < nks - 1; ++s) {
      WGP_LOAD();                                 // K-step s+1: in flight during the MFMAs below
      WGP_SB();
      frag_load(cur, 1, aw, bw);
      WGP_SB();
      frag_mfma(av, bv);                          // group 0
      WGP_SB();
      frag_load(cur, 2, av, bv);
      WGP_SB();
      frag_mfma(aw, bw);                          // group 1
      WGP_SB();
      WGP_STORE(cur ^ 1);                         // the other buffer: nobody reads it during this step
      WGP_SB();
      frag_load(cur, 3, aw, bw);
      WGP_SB();
      frag_mfma(av, bv);                          // group 2
      WGP_SB();
      __syncthreads();                            // all reads of `cur` issued and returned, all writes of cur ^ 1 done
      cur ^= 1;
      frag_load(cur, 0, av, bv);
      WGP_SB();
      frag_mfma(aw, bw);                          // group 3 of the previous buffer covers the new buffer's first reads
      WGP_SB();
    }
    frag_load(cur, 1, aw, bw);
    WGP_SB();
    frag_mfma(av, bv);
    WGP_SB();
    frag_load(cur, 2, av, bv);
    WGP_SB();
    frag_mfma(aw, bw);
    WGP_SB();
    frag_load(cur, 3, aw, bw);
    WGP_SB();
    frag_mfma(av, bv);
    frag_mfma(aw, bw);
  } else if constexpr (SCHED == 3) {
    // SCHED 1 with the loads of TWO K-steps in flight (tapgemm's schedule 7): the tile written to LDS in step s was requested in
    // step s-1 -- a whole K-step (32 MFMAs of this wave, 128 of its SIMD) more cover for an L2 miss, and the barrier makes every
    // wave wait for the slowest load.  Two staging register sets used alternately, two steps per trip (buffer index and register
    // set are compile-time constants); a K-step past the item's end loads nothing (every row fails `m < it.m1`: zero fill).
    float2 av[4], aw[4];
    float bv[4], bw[4];
    float4 ra1[2], rb1[2];
    WGP_LOADU_INTO(ra, rb);                       // step 0
    WGP_LOADU_INTO(ra1, rb1);                     // step 1
    WGP_STORE(0);
    __syncthreads();
    frag_load(0, 0, av, bv);
#define WGP_DSTEP(CUR, LOADS, SA, SB_)                                                                           \
    {                                                                                                           \
      LOADS                                                                                                     \
      WGP_SB();                                                                                                 \
      frag_load(CUR, 1, aw, bw);                                                                                \
      WGP_SB();                                                                                                 \
      frag_mfma(av, bv);                                                                                        \
      WGP_SB();                                                                                                 \
      frag_load(CUR, 2, av, bv);                                                                                \
      WGP_SB();                                                                                                 \
      frag_mfma(aw, bw);                                                                                        \
      WGP_SB();                                                                                                 \
      WGP_STORE_FROM(SA, SB_, (CUR) ^ 1);                                                                       \
      WGP_SB();                                                                                                 \
      frag_load(CUR, 3, aw, bw);                                                                                \
      WGP_SB();                                                                                                 \
      frag_mfma(av, bv);                                                                                        \
      WGP_SB();                                                                                                 \
      __syncthreads();                                                                                          \
      frag_load((CUR) ^ 1, 0, av, bv);                                                                          \
      WGP_SB();                                                                                                 \
      frag_mfma(aw, bw);                                                                                        \
      WGP_SB();                                                                                                 \
    }
    int s = 0, cur = 0;
    for (; s + 2 < nks; s += 2) {
      WGP_DSTEP(0, WGP_LOADU_INTO(ra, rb);, ra1, rb1)
      WGP_DSTEP(1, WGP_LOADU_INTO(ra1, rb1);, ra, rb)
    }
    if (s + 1 < nks) {                            // two tiles left: s in buffer 0, s+1 in (ra1, rb1)
      WGP_DSTEP(0, , ra1, rb1)
      cur = 1;
    }
#undef WGP_DSTEP
    frag_load(cur, 1, aw, bw);
    WGP_SB();
    frag_mfma(av, bv);
    WGP_SB();
    frag_load(cur, 2, av, bv);
    WGP_SB();
    frag_mfma(aw, bw);
    WGP_SB();
    frag_load(cur, 3, aw, bw);
    WGP_SB();
    frag_mfma(av, bv);
    frag_mfma(aw, bw);
  } else {
    // three fragment buffers, the loop rotated so that NO LDS read is pending at the back edge (the barrier's lgkmcnt(0) has
    // drained them): a step opens with the reads of its groups 0 and 1, then issues the next tile's global loads and computes
    // the PREVIOUS step's group 3 (held in its own buffer across the barrier) while they return -- the waits inside the body
    // are then exact counts (lgkmcnt(4): group 0 landed, group 1 still in flight)
    float2 f0a[4], f1a[4], f3a[4];
    float f0b[4], f1b[4], f3b[4];
    WGP_LOAD();
    WGP_STORE(0);
    __syncthreads();
    int cur = 0;
    for (int s = 0; s < nks; ++s) {
      const bool more = s + 1 < nks;
      frag_load(cur, 0, f0a, f0b);
      frag_load(cur, 1, f1a, f1b);
      WGP_SB();
      if (more) WGP_LOAD();                       // K-step s+1: in flight during the MFMAs below
      WGP_SB();
      if (s > 0) frag_mfma(f3a, f3b);             // group 3 of step s-1: covers the reads above and the load issue
      WGP_SB();
      frag_mfma(f0a, f0b);                        // group 0
      WGP_SB();
      frag_load(cur, 2, f0a, f0b);
      WGP_SB();
      frag_mfma(f1a, f1b);                        // group 1
      WGP_SB();
      if (more) WGP_STORE(cur ^ 1);               // the other buffer: nobody reads it during this step
      WGP_SB();
      frag_load(cur, 3, f3a, f3b);
      WGP_SB();
      frag_mfma(f0a, f0b);                        // group 2
      WGP_SB();
      if (more) __syncthreads();                  // all reads of `cur` returned, all writes of cur ^ 1 done
      cur ^= 1;
    }
    frag_mfma(f3a, f3b);
  }
#undef WGP_SB
#undef WGP_LOAD
#undef WGP_STORE
#undef WGP_LOAD_INTO
#undef WGP_STORE_FROM
#undef WGP_LOADU_INTO

  // partial[split][tap][co][ci]; MFMA C layout: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5); block e holds the wave's
  // channels 2 row + e
  float* out = p.partial + (size_t)it.split * p.slab_total + (size_t)it.tap * p.CoutPad * p.CinPad;
  const int ci = it.ci0 + wn * 32 + l31;
  if (ci < p.CinPad) {
#pragma unroll
    for (int r = 0; r < 16; ++r)
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int co = it.co0 + wm * 64 + 2 * ((r & 3) + 8 * (r >> 2) + 4 * half) + e;
        if (co < p.CoutPad) out[(size_t)co * p.CinPad + ci] = acc[e][r];
      }
  }
}

template <int BM, int BN, int WM, int WN>
static hipError_t launch_wg(const WgParams& p, int nitems, hipStream_t s) {
  static bool attr_set = false;
  const size_t lds = (size_t)2 * WG_BK * (BM + BN) * sizeof(float);
  auto k = tapwgrad_kernel<BM, BN, WM, WN>;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  hipLaunchKernelGGL(k, dim3(nitems), dim3(64 * WM * WN), lds, s, p);
  return hipGetLastError();
}

hipError_t launch_tapwgrad(int cfg, const WgParams& p, int nitems, hipStream_t s) {
  if (nitems <= 0) return hipSuccess;
  switch (cfg) {
#ifdef IAN_ABLATION   // the 4-wave 128x128 tile (superseded by the 8-wave one, bitwise the same result): libian_ablation.so only
    case WG_128x128: return launch_wg<128, 128, 2, 2>(p, nitems, s);
#endif
    case WG_32x128: return launch_wg<32, 128, 1, 4>(p, nitems, s);
    case WG_128x32: return launch_wg<128, 32, 4, 1>(p, nitems, s);
    case WG_128x128W8: return launch_wg<128, 128, 2, 4>(p, nitems, s);
    case WG_128x128P:
    case WG_128x128P2:
    case WG_128x128P3: {
      static bool attr_set = false;
      const size_t lds = (size_t)2 * WG_BK * (128 + 128) * sizeof(float);
      if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(tapwgrad_p_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(tapwgrad_p_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(tapwgrad_p_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_set = true;
      }
      if (cfg == WG_128x128P) hipLaunchKernelGGL(tapwgrad_p_kernel<1>, dim3(nitems), dim3(512), lds, s, p);
      else if (cfg == WG_128x128P3) hipLaunchKernelGGL(tapwgrad_p_kernel<3>, dim3(nitems), dim3(512), lds, s, p);
      else hipLaunchKernelGGL(tapwgrad_p_kernel<2>, dim3(nitems), dim3(512), lds, s, p);
      return hipGetLastError();
    }
  }
  return hipErrorInvalidValue;
}

// out[j] (+)= sum_s partial[s*slab_total + inv[j]]   (inv == nullptr: identity).  Fixed summation order.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partial, long long slab_total,
                                                           int nsplit, const int* __restrict__ inv,
                                                           float* __restrict__ out, long long count, int accumulate) {
  for (long long j = (long long)blockIdx.x * 256 + threadIdx.x; j < count; j += (long long)gridDim.x * 256) {
    const long long src = inv ? (long long)inv[j] : j;
    float s = 0.f;
    if (src >= 0)
      for (int k = 0; k < nsplit; ++k) s += partial[(size_t)k * slab_total + src];
    out[j] = accumulate ? out[j] + s : s;
  }
}
hipError_t launch_wgrad_reduce(const float* partial, long long slab_total, int nsplit, const int* inv, float* out,
                               long long count, int accumulate, hipStream_t s) {
  int blocks = (int)std::min<long long>((count + 255) / 256, 8192);
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, s, partial, slab_total, nsplit, inv, out, count,
                     accumulate);
  return hipGetLastError();
}

// ---- wgrad_reduce_tiled_kernel (round 6): the same sums, the scatter to the reference layout through LDS --------------------------
// wgrad_reduce_kernel walks the REFERENCE parameter in order and gathers from the slabs: consecutive reference elements are the
// taps of one (filter, channel) pair, i.e. one tap plane (CoutPad x Cin x 4 bytes) apart in every partial slab, and for the
// transposed conv's (Cin, Cout, 5, 5) parameter also the filters are a whole slab row apart -- every 4-byte read pulls its own
// 64-byte line: profiles/r06_wgrad.md measured 1.4-1.7 GB fetched per call where the partial slabs hold 0.13 GB, 195-244 us per
// call on three decoder layers (as long as half their GEMM).  Here one workgroup owns a TCO x TCI block of (filter, channel) pairs
// for ALL taps: it reads the slabs as they lie (16 bytes per lane, runs of TCI channels), sums the splits in the same fixed order
// (0 + p_0 + p_1 + ...: bitwise the old result), parks the block in LDS and writes it out in REFERENCE order -- runs of
// TCI x ntaps (conv) or TCO x ntaps (transposed conv) consecutive floats.  The host proves the layer's slab -> reference map affine
// (reference index = filter * s_co + channel * s_ci + tap_off[tap]) before choosing this kernel; anything else keeps the gather.
struct WgReduceTiled {
  const float* partial;
  float* out;
  long long slab_total;
  int nsplit, ntaps, plane /* CoutPad * CinPad */, CinPad, Cout, Cin, tco, tci, s_co, s_ci, accumulate, ci_inner;
  const int* co_off;           // or nullptr: co * s_co
  const int* ci_off;           // or nullptr: ci * s_ci
  unsigned char tap_off[48];   // slab tap -> offset inside the reference's tap block
  unsigned char tap_inv[48];   // reference tap offset -> slab tap
};
__global__ __launch_bounds__(256) void wgrad_reduce_tiled_kernel(const WgReduceTiled a) {
  extern __shared__ float red[];   // [ntaps][tco * (tci + 4) + 1]: an ODD plane stride -- the write-out walks the taps of one pair with consecutive
                                   // lanes, and a plane stride of 320 floats put all 25 of them on one bank (SQ_LDS_BANK_CONFLICT 92 %)
  const int tci_p = a.tci + 4, lplane = a.tco * tci_p + 1;
  const int tiles_ci = (a.Cin + a.tci - 1) / a.tci;
  const int co0 = (blockIdx.x / tiles_ci) * a.tco, ci0 = (blockIdx.x % tiles_ci) * a.tci;
  const int c4n = a.tci >> 2;                       // float4 per tile row
  const int per_tap = a.tco * c4n, total4 = a.ntaps * per_tap;
  for (int e = threadIdx.x; e < total4; e += 256) {
    const int t = e / per_tap, rem = e - t * per_tap;
    const int r = rem / c4n, c = (rem - r * c4n) * 4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (co0 + r < a.Cout && ci0 + c < a.CinPad) {   // CinPad is a multiple of 32 and tci of 4: a float4 never straddles the row end
      const float* src = a.partial + (size_t)t * a.plane + (size_t)(co0 + r) * a.CinPad + ci0 + c;
#pragma unroll 8
      for (int k = 0; k < a.nsplit; ++k) {
        const float4 v = *reinterpret_cast<const float4*>(src + (size_t)k * a.slab_total);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      }
    }
    float* q = red + t * lplane + r * tci_p + c;
    q[0] = s.x; q[1] = s.y; q[2] = s.z; q[3] = s.w;
  }
  __syncthreads();
  const int total = a.ntaps * a.tco * a.tci;
  const bool ci_inner = a.ci_inner != 0;            // conv: (Cout, Cin, taps); transposed conv / dense: (Cin, Cout, taps)
  for (int e = threadIdx.x; e < total; e += 256) {
    int r, c, tp;
    if (ci_inner) {
      r = e / (a.tci * a.ntaps);
      const int rem = e - r * (a.tci * a.ntaps);
      c = rem / a.ntaps; tp = rem - c * a.ntaps;
    } else {
      c = e / (a.tco * a.ntaps);
      const int rem = e - c * (a.tco * a.ntaps);
      r = rem / a.ntaps; tp = rem - r * a.ntaps;
    }
    if (co0 + r >= a.Cout || ci0 + c >= a.Cin) continue;
    const int t = a.tap_inv[tp];
    const float v = red[t * lplane + r * tci_p + c];
    float* o = a.out + (size_t)(a.co_off ? a.co_off[co0 + r] : (co0 + r) * a.s_co) + (size_t)(a.ci_off ? a.ci_off[ci0 + c] : (ci0 + c) * a.s_ci) + tp;
    *o = a.accumulate ? *o + v : v;
  }
}
hipError_t launch_wgrad_reduce_tiled(const WgReduceTiledDesc& d, const float* partial, long long slab_total, int nsplit, float* out,
                                     int accumulate, hipStream_t s) {
  WgReduceTiled a;
  a.partial = partial; a.out = out; a.slab_total = slab_total; a.nsplit = nsplit; a.ntaps = d.ntaps; a.plane = d.CoutPad * d.CinPad;
  a.CinPad = d.CinPad; a.Cout = d.Cout; a.Cin = d.Cin; a.tco = d.tco; a.tci = d.tci; a.s_co = d.s_co; a.s_ci = d.s_ci;
  a.accumulate = accumulate; a.ci_inner = d.ci_inner ? 1 : 0; a.co_off = d.co_off; a.ci_off = d.ci_off;
  for (int t = 0; t < 48; ++t) { a.tap_off[t] = d.tap_off[t]; a.tap_inv[t] = d.tap_inv[t]; }
  const int tiles = ((d.Cout + d.tco - 1) / d.tco) * ((d.Cin + d.tci - 1) / d.tci);
  const size_t lds = (size_t)d.ntaps * (d.tco * (d.tci + 4) + 1) * sizeof(float);
  hipLaunchKernelGGL(wgrad_reduce_tiled_kernel, dim3(tiles), dim3(256), lds, s, a);
  return hipGetLastError();
}

// ---- pack_tiled_kernel (round 6): reference-layout parameter -> packed slab, the inverse walk of wgrad_reduce_tiled_kernel -------
// gather_pack_kernel walks the SLAB and gathers from the reference: consecutive slab elements are 25 floats (conv) or Cout x 25
// floats (transposed conv) apart in the parameter -- 187 + 234 us for the two slabs of dec_conv1 (52 MB each), 0.78 ms at the head
// of every generator update.  Here a workgroup reads a TCO x TCI block of (filter, channel) pairs with all their taps as the
// contiguous runs they are in the reference, parks them in LDS and writes the slab rows (runs of TCI floats, 16 bytes per lane);
// slab padding inside the block is written as zeros, padding outside every block was zero-filled when the layer was created and
// is never touched.  Same values as the gather (a copy): bitwise the old slabs.
struct PackTiled {
  const float* ref;
  float* slab;
  int ntaps, plane, CinPad, CoutPad, Cout, Cin, tco, tci, s_co, s_ci, ci_inner;
  const int* co_off;
  const int* ci_off;
  unsigned char tap_off[48];
};
__global__ __launch_bounds__(256) void pack_tiled_kernel(const PackTiled a) {
  extern __shared__ float red[];   // [ntaps][tco * (tci + 4) + 1]
  const int tci_p = a.tci + 4, lplane = a.tco * tci_p + 1;
  const int tiles_ci = (a.Cin + a.tci - 1) / a.tci;
  const int co0 = (blockIdx.x / tiles_ci) * a.tco, ci0 = (blockIdx.x % tiles_ci) * a.tci;
  const int total = a.ntaps * a.tco * a.tci;
  const bool ci_inner = a.ci_inner != 0;
  for (int e = threadIdx.x; e < total; e += 256) {
    int r, c, tp;
    if (ci_inner) {
      r = e / (a.tci * a.ntaps);
      const int rem = e - r * (a.tci * a.ntaps);
      c = rem / a.ntaps; tp = rem - c * a.ntaps;
    } else {
      c = e / (a.tco * a.ntaps);
      const int rem = e - c * (a.tco * a.ntaps);
      r = rem / a.ntaps; tp = rem - r * a.ntaps;
    }
    float v = 0.f;
    if (co0 + r < a.Cout && ci0 + c < a.Cin)
      v = a.ref[(size_t)(a.co_off ? a.co_off[co0 + r] : (co0 + r) * a.s_co) + (size_t)(a.ci_off ? a.ci_off[ci0 + c] : (ci0 + c) * a.s_ci) + tp];
    red[tp * lplane + r * tci_p + c] = v;          // indexed by the REFERENCE tap position; the write-out maps slab taps to it
  }
  __syncthreads();
  const int c4n = a.tci >> 2, per_tap = a.tco * c4n, total4 = a.ntaps * per_tap;
  for (int e = threadIdx.x; e < total4; e += 256) {
    const int t = e / per_tap, rem = e - t * per_tap;
    const int r = rem / c4n, c = (rem - r * c4n) * 4;
    if (co0 + r >= a.CoutPad || ci0 + c >= a.CinPad) continue;
    const float* q = red + (int)a.tap_off[t] * lplane + r * tci_p + c;
    *reinterpret_cast<float4*>(a.slab + (size_t)t * a.plane + (size_t)(co0 + r) * a.CinPad + ci0 + c) = make_float4(q[0], q[1], q[2], q[3]);
  }
}
hipError_t launch_pack_tiled(const WgReduceTiledDesc& d, const float* ref, float* slab, hipStream_t s) {
  PackTiled a;
  a.ref = ref; a.slab = slab; a.ntaps = d.ntaps; a.plane = d.CoutPad * d.CinPad; a.CinPad = d.CinPad; a.CoutPad = d.CoutPad;
  a.Cout = d.Cout; a.Cin = d.Cin; a.tco = d.tco; a.tci = d.tci; a.s_co = d.s_co; a.s_ci = d.s_ci;
  a.ci_inner = d.ci_inner ? 1 : 0; a.co_off = d.co_off; a.ci_off = d.ci_off;
  for (int t = 0; t < 48; ++t) a.tap_off[t] = d.tap_off[t];
  const int tiles = ((d.Cout + d.tco - 1) / d.tco) * ((d.Cin + d.tci - 1) / d.tci);
  const size_t lds = (size_t)d.ntaps * (d.tco * (d.tci + 4) + 1) * sizeof(float);
  hipLaunchKernelGGL(pack_tiled_kernel, dim3(tiles), dim3(256), lds, s, a);
  return hipGetLastError();
}

// dst[i] = map[i] >= 0 ? src[map[i]] : 0   (reference-layout parameter -> packed slab, built on the device every step)
__global__ __launch_bounds__(256) void gather_pack_kernel(const float* __restrict__ src, const int* __restrict__ map,
                                                          float* __restrict__ dst, long long count) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < count; i += (long long)gridDim.x * 256) {
    const int m = map[i];
    dst[i] = m >= 0 ? src[m] : 0.f;
  }
}
hipError_t launch_gather_pack(const float* src, const int* map, float* dst, long long count, hipStream_t s) {
  int blocks = (int)std::min<long long>((count + 255) / 256, 8192);
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(gather_pack_kernel, dim3(blocks), dim3(256), 0, s, src, map, dst, count);
  return hipGetLastError();
}

// ---- MDCL (layers.py:207-258): composite stencil <-> (W, coefficients) on the device -------------------------
// slab_f[t][co][ci] = sum over the (branch b, pq) pairs that land on tap t of coeff_b[co] * W[co,ci,pq]
//                     (+ tap 0: coeff_1x1[co] * mean_pq W[co,ci,:])          forward  [tap][CoutPad][CinPad]
// slab_b[t][ci][co] = same value, transposed                                  backward [tap][CinPadB][CoutPadB]
__global__ __launch_bounds__(256) void mdc_pack_kernel(MdcPackArgs a) {
  const long long total = (long long)a.ntaps * a.cout * a.cin;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int ci = (int)(i % a.cin);
    const int co = (int)((i / a.cin) % a.cout);
    const int t = (int)(i / ((long long)a.cin * a.cout));
    const float* w = a.W + ((size_t)co * a.cin + ci) * 9;
    float v = 0.f;
    for (int e = a.tap_start[t]; e < a.tap_start[t + 1]; ++e) v += a.coeff[a.ent_branch[e]][co] * w[a.ent_pq[e]];
    if (t == 0 && a.coeff_1x1) {
      float m = 0.f;
      for (int k = 0; k < 9; ++k) m += w[k];
      v += a.coeff_1x1[co] * (m / 9.f);
    }
    a.slab_f[((size_t)t * a.f_rows + co) * a.f_cols + ci] = v;
    a.slab_b[((size_t)t * a.b_rows + ci) * a.b_cols + co] = v;
  }
}
hipError_t launch_mdc_pack(const MdcPackArgs& a, hipStream_t s) {
  const long long total = (long long)a.ntaps * a.cout * a.cin;
  int blocks = (int)std::min<long long>((total + 255) / 256, 8192);
  hipLaunchKernelGGL(mdc_pack_kernel, dim3(blocks), dim3(256), 0, s, a);
  return hipGetLastError();
}

// gradient of the composite stencil dS [tap][f_rows][f_cols] -> dW (Cout,Cin,3,3) and the per-filter coefficients
//   dW[co,ci,pq]   = sum_b coeff_b[co] * dS[t(b,pq)][co][ci]  (+ coeff_1x1[co]/9 * dS[0][co][ci])
//   dcoeff_b[co]   = sum_{ci,pq} W[co,ci,pq] * dS[t(b,pq)][co][ci];   dcoeff_1x1[co] = sum_ci mean_pq(W) * dS[0][co][ci]
// one block per output filter co (coefficient sums reduced in LDS, fixed order).
__global__ __launch_bounds__(256) void mdc_unpack_grad_kernel(MdcPackArgs a, const float* __restrict__ dS,
                                                              float* __restrict__ dW, MdcCoeffGrads g, int accumulate) {
  __shared__ float red[256];
  const int co = blockIdx.x;
  float part[MDC_MAX_BRANCH + 1];
#pragma unroll
  for (int b = 0; b <= MDC_MAX_BRANCH; ++b) part[b] = 0.f;
  for (int ci = threadIdx.x; ci < a.cin; ci += 256) {
    const float* w = a.W + ((size_t)co * a.cin + ci) * 9;
    float dw[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) dw[k] = 0.f;
    for (int t = 0; t < a.ntaps; ++t) {
      const float d = dS[((size_t)t * a.f_rows + co) * a.f_cols + ci];
      for (int e = a.tap_start[t]; e < a.tap_start[t + 1]; ++e) {
        const int b = a.ent_branch[e], pq = a.ent_pq[e];
        dw[pq] += a.coeff[b][co] * d;
        part[b] += w[pq] * d;
      }
      if (t == 0 && a.coeff_1x1) {
        float m = 0.f;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
          dw[k] += a.coeff_1x1[co] * d * (1.f / 9.f);
          m += w[k];
        }
        part[MDC_MAX_BRANCH] += (m / 9.f) * d;
      }
    }
    float* o = dW + ((size_t)co * a.cin + ci) * 9;
#pragma unroll
    for (int k = 0; k < 9; ++k) o[k] = accumulate ? o[k] + dw[k] : dw[k];
  }
  for (int b = 0; b <= MDC_MAX_BRANCH; ++b) {
    float* dst = (b == MDC_MAX_BRANCH) ? g.d1x1 : (b < a.nbranch ? g.d[b] : nullptr);
    if (!dst) continue;
    red[threadIdx.x] = part[b];
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
      __syncthreads();
    }
    if (threadIdx.x == 0) dst[co] = accumulate ? dst[co] + red[0] : red[0];
    __syncthreads();
  }
}
hipError_t launch_mdc_unpack_grad(const MdcPackArgs& a, const float* dS, float* dW, const MdcCoeffGrads& g,
                                  int accumulate, hipStream_t s) {
  hipLaunchKernelGGL(mdc_unpack_grad_kernel, dim3(a.cout), dim3(256), 0, s, a, dS, dW, g, accumulate);
  return hipGetLastError();
}

}  // namespace ian

// Practical fp32-MFMA ceiling on this box: v_mfma_f32_32x32x2_f32 issued back to back from registers
// (no memory traffic), zero vs random operands, for run lengths comparable to one IAN layer (~200 us).
// Used to separate "pipeline stalls" from "DVFS clock" in the tapgemm roofline fraction.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void mfma_loop(const float* __restrict__ in, float* __restrict__ out, int iters) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  float a0 = in[t * 4 + 0], a1 = in[t * 4 + 1], b0 = in[t * 4 + 2], b1 = in[t * 4 + 3];
  f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
  for (int i = 0; i < iters; ++i) {
    c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, c3, 0, 0, 0);
  }
  float s = 0;
  for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
  out[t] = s;
}
int main() {
  const int blocks_per_cu[2] = {1, 2};
  for (int fill = 0; fill < 2; ++fill)
    for (int bi = 0; bi < 2; ++bi) {
      const int blocks = 256 * blocks_per_cu[bi];
      const size_t n = (size_t)blocks * 256;
      std::vector<float> h(n * 4);
      for (auto& v : h) v = fill ? (float)rand() / RAND_MAX * 0.02f - 0.01f : 0.f;
      float *din, *dout;
      hipMalloc(&din, n * 4 * sizeof(float));
      hipMalloc(&dout, n * sizeof(float));
      hipMemcpy(din, h.data(), n * 4 * sizeof(float), hipMemcpyHostToDevice);
      for (int iters : {400, 1600, 6400}) {  // 4 MFMA x 64 cyc x iters: ~43 us, 170 us, 680 us at 2.4 GHz (1 block/CU)
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipLaunchKernelGGL(mfma_loop, dim3(blocks), dim3(256), 0, 0, din, dout, iters);
        hipDeviceSynchronize();
        float best = 1e30f, sum = 0;
        const int reps = 20;
        hipEventRecord(e0);
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(mfma_loop, dim3(blocks), dim3(256), 0, 0, din, dout, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&sum, e0, e1);
        best = sum / reps;
        const double flops = (double)blocks * 4 /*waves*/ * 4.0 * iters * 4096.0;
        printf("fill=%s blocks/CU=%d iters=%d : %.1f us/launch  %.1f TF/s  (implied clock %.2f GHz at 64 cyc/MFMA/SIMD)\n",
               fill ? "random" : "zero", blocks_per_cu[bi], iters, best * 1e3, flops / (best * 1e-3) / 1e12,
               (double)blocks_per_cu[bi] * 4.0 * iters * 64.0 / (best * 1e-3) / 1e9 * 1.0);
      }
      hipFree(din);
      hipFree(dout);
    }
  return 0;
}

// What does a PHASE BOUNDARY cost on this box, as a kernel boundary and as an in-kernel grid barrier?  (DESIGN.md section 4:
// the persistent batch-1 decoder was priced from MI355X_MICROARCH.md's list -- kernel boundary 1.45 us, barrier-xcd 4.1-7.2 us --
// and not built; this measures the two on the shape of the brush event: 256 workgroups, ~10 dependent phases, each phase reads
// what EVERY workgroup of the previous phase wrote.)
//   build: hipcc --offload-arch=gfx950 -O3 -o /tmp/bvb scripts/ubench/boundary_vs_barrier.hip ; run: /tmp/bvb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
constexpr int NWG = 256, NT = 256;

// one phase of "work": workgroup b sums one value from each workgroup of the previous phase and writes its own
__device__ __forceinline__ void phase_body(const float* __restrict__ prev, float* __restrict__ next, int b) {
  __shared__ float red[NT];
  float v = __builtin_nontemporal_load(prev + threadIdx.x);          // NWG == NT: one value per producer
  red[threadIdx.x] = v;
  __syncthreads();
  for (int s = NT / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) next[b] = red[0] * (1.f / NWG) + 1.f;
}
__global__ __launch_bounds__(NT) void phase_kernel(const float* prev, float* next) { phase_body(prev, next, blockIdx.x); }

// (a) flat grid barrier: one counter, agent-scope release / acquire
__device__ __forceinline__ void grid_barrier_flat(unsigned* counter, unsigned target) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();                                                 // release: this workgroup's writes reach the device
    atomicAdd(counter, 1u);
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    __threadfence();                                                 // acquire
  }
  __syncthreads();
}
// (b) hierarchical: workgroups of one XCD (blockIdx % 8) meet on their own counter, the eight leaders on a global one
__device__ __forceinline__ void grid_barrier_xcd(unsigned* xcd_counters, unsigned* top, unsigned* release_flag, unsigned phase) {
  __syncthreads();
  if (threadIdx.x == 0) {
    const int x = blockIdx.x & 7;
    __threadfence();
    const unsigned ticket = atomicAdd(xcd_counters + x * 32, 1u);   // 128 B apart
    if (ticket == phase * (NWG / 8) - 1) {                           // last of this XCD for this phase
      const unsigned t2 = atomicAdd(top, 1u);
      if (t2 == phase * 8 - 1) __hip_atomic_store(release_flag, phase, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
    while (__hip_atomic_load(release_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < phase) __builtin_amdgcn_s_sleep(1);
    __threadfence();
  }
  __syncthreads();
}
template <int KIND>
__global__ __launch_bounds__(NT) void persistent_kernel(float* a, float* b, int phases, unsigned* counters) {
  for (int p = 0; p < phases; ++p) {
    phase_body((p & 1) ? b : a, (p & 1) ? a : b, blockIdx.x);
    if (KIND == 0) grid_barrier_flat(counters, (unsigned)(p + 1) * NWG);
    else grid_barrier_xcd(counters + 64, counters + 32, counters + 48, (unsigned)(p + 1));
  }
}

int main() {
  float *a, *b;
  unsigned* counters;
  CK(hipMalloc(&a, NWG * sizeof(float)));
  CK(hipMalloc(&b, NWG * sizeof(float)));
  CK(hipMalloc(&counters, 4096));
  std::vector<float> ones(NWG, 1.f);
  hipStream_t st;
  CK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int phases : {10, 100}) {
    float ms[3] = {0, 0, 0};
    for (int kind = 0; kind < 3; ++kind) {
      const int reps = 50;
      for (int r = -5; r < reps; ++r) {
        CK(hipMemcpyAsync(a, ones.data(), NWG * sizeof(float), hipMemcpyHostToDevice, st));
        CK(hipMemsetAsync(counters, 0, 4096, st));
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        if (kind == 0) {
          for (int p = 0; p < phases; ++p) hipLaunchKernelGGL(phase_kernel, dim3(NWG), dim3(NT), 0, st, (p & 1) ? b : a, (p & 1) ? a : b);
        } else if (kind == 1) {
          hipLaunchKernelGGL(persistent_kernel<0>, dim3(NWG), dim3(NT), 0, st, a, b, phases, counters);
        } else {
          hipLaunchKernelGGL(persistent_kernel<1>, dim3(NWG), dim3(NT), 0, st, a, b, phases, counters);
        }
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float t;
        CK(hipEventElapsedTime(&t, e0, e1));
        if (r >= 0) ms[kind] += t / reps;
      }
    }
    float out[NWG];
    CK(hipMemcpy(out, (phases & 1) ? b : a, sizeof out, hipMemcpyDeviceToHost));
    printf("%3d phases: launches %.2f us/phase | flat grid barrier %.2f us/phase | XCD-hierarchical barrier %.2f us/phase   (check %.4f)\n",
           phases, ms[0] * 1e3 / phases, ms[1] * 1e3 / phases, ms[2] * 1e3 / phases, out[0]);
  }
  return 0;
}

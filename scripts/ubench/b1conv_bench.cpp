// Times ian::launch_b1conv for the six batch-1 decoder geometries of IAN_simple (random data; links libian.so).
// build: hipcc --offload-arch=gfx950 -O2 -std=c++17 -I neural_photo_editor_amd/csrc -I include scripts/ubench/b1conv_bench.cpp \
//        -L neural_photo_editor_amd -lian -Wl,-rpath,$PWD/neural_photo_editor_amd -o scripts/ubench/b1conv_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "ian_internal.h"
using namespace ian;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
int main(int argc, char** argv) {
  const int dbg = argc > 1 ? atoi(argv[1]) : 0;
  struct G { const char* name; int mode, H, W, cin, cout; } gs[] = {
      {"dc1 fwd", 0, 4, 4, 1024, 512}, {"dc2 fwd", 0, 8, 8, 512, 256}, {"dc3 fwd", 0, 16, 16, 256, 128},
      {"dc3 bwd", 1, 16, 16, 256, 128}, {"dc2 bwd", 1, 8, 8, 512, 256}, {"dc1 bwd", 1, 4, 4, 1024, 512}};
  for (auto& g : gs) {
    const size_t wn = (size_t)25 * g.cin * g.cout;
    float *w, *x, *y;
    std::vector<float> hw(wn);
    for (size_t i = 0; i < wn; ++i) hw[i] = (float)((i * 2654435761u) % 1000) * 1e-3f - 0.5f;
    CK(hipMalloc(&w, wn * 4)); CK(hipMemcpy(w, hw.data(), wn * 4, hipMemcpyHostToDevice));
    const int xin_c = g.mode == 0 ? g.cin : g.cout, xin_h = g.mode == 0 ? g.H : 2 * g.H;
    const int yo_c = g.mode == 0 ? g.cout : g.cin, yo_h = g.mode == 0 ? 2 * g.H : g.H;
    const size_t xn = (size_t)xin_h * xin_h * xin_c, yn = (size_t)yo_h * yo_h * yo_c;
    CK(hipMalloc(&x, xn * 4)); CK(hipMalloc(&y, yn * 4));
    CK(hipMemset(x, 0, xn * 4));
    B1Params p = B1Params();
    p.x = x; p.w = w; p.y = y; p.act = 1; p.bwd = g.mode; p.dbg = dbg;
    p.IH = xin_h; p.IW = xin_h; p.Cr = xin_c; p.xs = xin_c; p.OH = yo_h; p.OW = yo_h; p.ys = yo_c;
    p.nslices = yo_c / 16; p.tiles_x = g.W / 4; p.ntiles = (g.H / 4) * (g.W / 4); p.x_bytes = (unsigned)(xn * 4);
    p.kshift = 0; while ((32 << p.kshift) < p.Cr) ++p.kshift;
    if (g.mode == 0) {
      long long off = 0;
      for (int c = 0; c < 4; ++c) { p.cls_off[c] = off; off += (long long)((c >> 1) ? 2 : 3) * ((c & 1) ? 2 : 3) * g.cin * g.cout; }
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 20; ++i) CK(launch_b1conv(p, g.mode, 0));
    CK(hipDeviceSynchronize());
    const int reps = 200;
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) CK(launch_b1conv(p, g.mode, 0));
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const int blocks = (g.mode == 0 ? 4 : 1) * p.ntiles * p.nslices;
    printf("dbg %d  %s: %7.2f us per launch  (%d blocks, %.1f MB weights -> %.2f TB/s)\n", dbg, g.name, ms / reps * 1e3, blocks, wn * 4 / 1e6,
           wn * 4 / (ms / reps * 1e-3) / 1e12);
    CK(hipFree(w)); CK(hipFree(x)); CK(hipFree(y));
  }
  return 0;
}

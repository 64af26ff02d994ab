/* CPU ORACLE (plain C restatement of the hot primitives) -- test infrastructure only, never linked into or called
 * by the product path.  Third, independently written statement of the index conventions the HIP kernels must
 * reproduce (beside oracle/ian_oracle.py in numpy and oracle/torch_twin.py in torch): scalar loops, float32 data,
 * float64 accumulation.  Index conventions of the third-party primitives are [recalled] (see ian_oracle.py; the
 * reference-owned compositions above them are pinned by execution, tests/golden/ref_*.npz); every function cites the reference lines it restates.  NCHW layout as in the reference.
 *
 * Build: make -C oracle   (gcc -O2 -shared -fPIC -> oracle/_ref/libian_primitives.so, git-ignored)
 */
#include <stddef.h>

/* IAN_simple.py:73-116 / IAN.py:71-110: Conv2D(DNN)Layer 5x5, stride 2, pad 2, flip_filters=False (correlation).
 * x (N,Cin,H,W), W (Cout,Cin,5,5), b (Cout) or NULL -> y (N,Cout,H/2,W/2):
 * y[n,co,oy,ox] = b[co] + sum W[co,ci,ky,kx] * x[n,ci,2oy-2+ky,2ox-2+kx] */
void ref_conv5s2(const float* x, const float* W, const float* b, float* y, int N, int Cin, int H, int Wd, int Cout) {
  const int OH = H / 2, OW = Wd / 2;
  for (int n = 0; n < N; ++n)
    for (int co = 0; co < Cout; ++co)
      for (int oy = 0; oy < OH; ++oy)
        for (int ox = 0; ox < OW; ++ox) {
          double acc = b ? b[co] : 0.0;
          for (int ci = 0; ci < Cin; ++ci)
            for (int ky = 0; ky < 5; ++ky) {
              const int iy = 2 * oy - 2 + ky;
              if (iy < 0 || iy >= H) continue;
              for (int kx = 0; kx < 5; ++kx) {
                const int ix = 2 * ox - 2 + kx;
                if (ix < 0 || ix >= Wd) continue;
                acc += (double)W[((co * Cin + ci) * 5 + ky) * 5 + kx] * x[((size_t)(n * Cin + ci) * H + iy) * Wd + ix];
              }
            }
          y[((size_t)(n * Cout + co) * OH + oy) * OW + ox] = (float)acc;
        }
}

/* layers.py:436-483 DeconvLayer (== IAN_simple.py:182-223 dnn=False branch): 5x5 stride-2 transposed convolution,
 * output forced to 2x the input (:460).  W (Cin,Cout,5,5) (:449-452).  Scatter rule:
 *   y[n,co,oy,ox] += x[n,ci,iy,ix] * W[ci,co,kt_y,kt_x],   oy = 2iy - 2 + ky,  kt = flip ? 4-k : k  (App. B.2) */
void ref_deconv5s2(const float* x, const float* W, float* y, int N, int Cin, int H, int Wd, int Cout, int flip) {
  const int OH = 2 * H, OW = 2 * Wd;
  for (size_t i = 0; i < (size_t)N * Cout * OH * OW; ++i) y[i] = 0.f;
  for (int n = 0; n < N; ++n)
    for (int ci = 0; ci < Cin; ++ci)
      for (int iy = 0; iy < H; ++iy)
        for (int ix = 0; ix < Wd; ++ix) {
          const float xv = x[((size_t)(n * Cin + ci) * H + iy) * Wd + ix];
          for (int ky = 0; ky < 5; ++ky) {
            const int oy = 2 * iy - 2 + ky;
            if (oy < 0 || oy >= OH) continue;
            for (int kx = 0; kx < 5; ++kx) {
              const int ox = 2 * ix - 2 + kx;
              if (ox < 0 || ox >= OW) continue;
              const int ty = flip ? 4 - ky : ky, tx = flip ? 4 - kx : kx;
              for (int co = 0; co < Cout; ++co)
                y[((size_t)(n * Cout + co) * OH + oy) * OW + ox] += xv * W[((ci * Cout + co) * 5 + ty) * 5 + tx];
            }
          }
        }
}

/* layers.py:207-258 MDCL: sum of branches sharing one W (Cout,Cin,3,3) (:220):
 *   base 3x3 pad 1 * coeff_base (:223-232); scale 0: 1x1 with mean(W,[2,3]) * coeff (:238-247);
 *   scale s>0: dilation-s 3x3 on input padded by s * coeff (:250-257).  coeffs: [1+nscales][Cout], base first. */
void ref_mdcl(const float* x, const float* W, const float* coeffs, const int* scales, int nscales, float* y, int N, int Cin,
              int H, int Wd, int Cout) {
  for (int n = 0; n < N; ++n)
    for (int co = 0; co < Cout; ++co)
      for (int oy = 0; oy < H; ++oy)
        for (int ox = 0; ox < Wd; ++ox) {
          double total = 0.0;
          for (int br = 0; br <= nscales; ++br) {
            const int s = br == 0 ? 1 : scales[br - 1];
            double acc = 0.0;
            for (int ci = 0; ci < Cin; ++ci) {
              const float* w = W + (size_t)(co * Cin + ci) * 9;
              if (s == 0) {
                double m = 0.0;
                for (int k = 0; k < 9; ++k) m += w[k];
                acc += (m / 9.0) * x[((size_t)(n * Cin + ci) * H + oy) * Wd + ox];
                continue;
              }
              for (int p = 0; p < 3; ++p) {
                const int iy = oy + s * (p - 1);
                if (iy < 0 || iy >= H) continue;
                for (int q = 0; q < 3; ++q) {
                  const int ix = ox + s * (q - 1);
                  if (ix < 0 || ix >= Wd) continue;
                  acc += (double)w[p * 3 + q] * x[((size_t)(n * Cin + ci) * H + iy) * Wd + ix];
                }
              }
            }
            total += acc * coeffs[br * Cout + co];
          }
          y[((size_t)(n * Cout + co) * H + oy) * Wd + ox] = (float)total;
        }
}

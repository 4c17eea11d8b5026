// probe_rsq.hip -- how accurate is the v_rsq_f64 seed on this chip?  max |1 - x y^2| over 2^26 x in [1, 4) and the
// error left by a second-order (4 instructions) and a third-order (5 instructions, what rsqrt_f64 uses) correction.
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o /tmp/probe_rsq tools/probe_rsq.hip && /tmp/probe_rsq
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>

__global__ void probe(unsigned long long n, double* out) {
  double m_seed = 0, m2 = 0, m3 = 0;
  for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x) {
    const double x = 1.0 + 3.0 * ((double)i + 0.37) / (double)n;
    const double y = __builtin_amdgcn_rsq(x);
    const double e = __builtin_fma(-(x * y), y, 1.0);
    m_seed = fmax(m_seed, fabs(e));
    const double y2 = __builtin_fma(y * e, 0.5, y);                                    // y (1 + e/2)
    const double y3 = __builtin_fma(y, __builtin_fma(0.375, e, 0.5) * e, y);            // y (1 + e/2 + 3 e^2/8)
    const double ex = 1.0 / sqrt(x);
    m2 = fmax(m2, fabs(y2 - ex) / ex);
    m3 = fmax(m3, fabs(y3 - ex) / ex);
  }
  __shared__ double s[3][256];
  s[0][threadIdx.x] = m_seed; s[1][threadIdx.x] = m2; s[2][threadIdx.x] = m3;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 1; k < 256; ++k) { s[0][0] = fmax(s[0][0], s[0][k]); s[1][0] = fmax(s[1][0], s[1][k]); s[2][0] = fmax(s[2][0], s[2][k]); }
    out[3 * blockIdx.x] = s[0][0]; out[3 * blockIdx.x + 1] = s[1][0]; out[3 * blockIdx.x + 2] = s[2][0];
  }
}

int main() {
  const int blocks = 1024;
  double* d;
  if (hipMalloc(&d, sizeof(double) * 3 * blocks) != hipSuccess) return 1;
  hipLaunchKernelGGL(probe, dim3(blocks), dim3(256), 0, 0, 1ull << 26, d);
  static double h[3 * 1024];
  if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return 1;
  double a = 0, b = 0, c = 0;
  for (int k = 0; k < blocks; ++k) { a = fmax(a, h[3 * k]); b = fmax(b, h[3 * k + 1]); c = fmax(c, h[3 * k + 2]); }
  printf("v_rsq_f64 seed: max |1 - x y^2| = %.3e = 2^%.2f\n", a, log2(a));
  printf("after y (1 + e/2)            : max rel. error %.3e   (4 instructions)\n", b);
  printf("after y (1 + e/2 + 3 e^2/8)  : max rel. error %.3e   (5 instructions)\n", c);
  return 0;
}

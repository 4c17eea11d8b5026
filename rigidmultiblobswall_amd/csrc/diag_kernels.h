// diag_kernels.h -- measurement aid: the chip's fp64 VALU issue ceiling, measured in the same process and clock
// state as the kernels it prices (bench.py `roofline.issue.peak`).  Every wave runs ITERS x 16 independent
// v_fma_f64 (8 accumulators, no memory traffic); 4 waves per SIMD on every CU.  tools/ubench.hip is the long form
// (all instruction classes; profiles/r1_ubench_fp64_issue_rates.txt).
#pragma once
#include <hip/hip_runtime.h>

namespace rmb {

constexpr int kUbenchIters = 2048;
constexpr int kUbenchFmaPerIter = 16;

__global__ __launch_bounds__(256) void ubench_fma64_kernel(double* out, double x, double y) {
  double a0 = threadIdx.x * 1e-3 + 1.0, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  for (int i = 0; i < kUbenchIters; ++i) {
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a0) : "v"(x), "v"(y));
      asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a1) : "v"(x), "v"(y));
      asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a2) : "v"(x), "v"(y));
      asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a3) : "v"(x), "v"(y));
      asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a4) : "v"(x), "v"(y));
      asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a5) : "v"(x), "v"(y));
      asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a6) : "v"(x), "v"(y));
      asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a7) : "v"(x), "v"(y));
    }
  }
  out[(long)blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

}  // namespace rmb

// ubench.hip -- fp64 VALU issue-cost microbenchmark for gfx950 (numbers the guides do not list).
// For each instruction: 8 independent chains per wave, W waves per SIMD on every CU; reports
// shader cycles (s_memtime) per wave-instruction as seen by one wave, and x/W = issue cost per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 -o ubench tools/ubench.hip ; run: ./ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITERS = 2048;

#define OP8(ASM)                                                         \
  asm volatile(ASM : "+v"(a0) : "v"(x), "v"(y));                           \
  asm volatile(ASM : "+v"(a1) : "v"(x), "v"(y));                           \
  asm volatile(ASM : "+v"(a2) : "v"(x), "v"(y));                           \
  asm volatile(ASM : "+v"(a3) : "v"(x), "v"(y));                           \
  asm volatile(ASM : "+v"(a4) : "v"(x), "v"(y));                           \
  asm volatile(ASM : "+v"(a5) : "v"(x), "v"(y));                           \
  asm volatile(ASM : "+v"(a6) : "v"(x), "v"(y));                           \
  asm volatile(ASM : "+v"(a7) : "v"(x), "v"(y));

#define KERNEL_T(NAME, ASM, ACC_T, OP_T)                                                        \
  __global__ void NAME(double* out, unsigned long long* cyc, double xd, double yd) {            \
    ACC_T a0 = (ACC_T)(threadIdx.x * 1e-3 + 1.0), a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4,   \
           a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;                                               \
    OP_T x = (OP_T)xd, y = (OP_T)yd;                                                            \
    unsigned long long t0 = __builtin_readcyclecounter();                                       \
    for (int i = 0; i < ITERS; ++i) { OP8(ASM) OP8(ASM) }                                       \
    unsigned long long t1 = __builtin_readcyclecounter();                                       \
    out[blockIdx.x * blockDim.x + threadIdx.x] = (double)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7); \
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;   \
  }

#define KERNEL(NAME, ASM) KERNEL_T(NAME, ASM, double, double)
KERNEL(k_fma64, "v_fma_f64 %0, %1, %2, %0")
KERNEL(k_mul64, "v_mul_f64 %0, %0, %1")
KERNEL(k_add64, "v_add_f64 %0, %0, %1")
KERNEL(k_rsq64, "v_rsq_f64 %0, %0")
KERNEL(k_rcp64, "v_rcp_f64 %0, %0")
KERNEL(k_sqrt64, "v_sqrt_f64 %0, %0")
KERNEL(k_mov64, "v_mov_b64 %0, %1")
KERNEL(k_trunc64, "v_trunc_f64 %0, %0")
KERNEL_T(k_fma32, "v_fma_f32 %0, %1, %2, %0", float, float)
KERNEL_T(k_rsq32, "v_rsq_f32 %0, %0", float, float)
KERNEL(k_pkfma32, "v_pk_fma_f32 %0, %1, %2, %0")
KERNEL_T(k_cvt_f32_f64, "v_cvt_f32_f64 %0, %1", float, double)
KERNEL_T(k_cvt_f64_f32, "v_cvt_f64_f32 %0, %1", double, float)
KERNEL_T(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc", float, float)
KERNEL(k_cmp64, "v_cmp_gt_f64 vcc, %0, %1")
KERNEL(k_fmac64, "v_fmac_f64 %0, %1, %2")
KERNEL(k_ldexp64, "v_ldexp_f64 %0, %0, 1")

// dependent chain latency of v_fma_f64
__global__ void k_fma64_dep(double* out, unsigned long long* cyc, double x, double y) {
  double a0 = threadIdx.x * 1e-3 + 1.0;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int k = 0; k < 16; ++k) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a0) : "v"(x), "v"(y));
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0;
  if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

// broadcast LDS read (all lanes same address), 8 reads per group
__global__ void k_lds_b128_bcast(double* out, unsigned long long* cyc, double x, double y) {
  __shared__ double2 buf[1024];
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) buf[i] = make_double2(x + i, y);
  __syncthreads();
  double acc = 0;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int k = 0; k < 16; ++k) { double2 v = buf[(i * 16 + k) & 1023]; acc += v.x; }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
  if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

typedef void (*kern_t)(double*, unsigned long long*, double, double);

int run(const char* name, kern_t k, int waves_per_simd, double* out, unsigned long long* cyc, double instr_per_iter_group) {
  const int block = 64 * 4 * waves_per_simd > 1024 ? 1024 : 64 * 4 * waves_per_simd;
  const int blocks_per_cu = (64 * 4 * waves_per_simd) / block;
  const int grid = 256 * blocks_per_cu;
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k, dim3(grid), dim3(block), 0, 0, out, cyc, 1.0000001, 1e-9);
  CHK(hipDeviceSynchronize());
  CHK(hipEventRecord(e0));
  hipLaunchKernelGGL(k, dim3(grid), dim3(block), 0, 0, out, cyc, 1.0000001, 1e-9);
  CHK(hipEventRecord(e1));
  CHK(hipDeviceSynchronize());
  float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
  const int nw = grid * block / 64;
  std::vector<unsigned long long> h(nw);
  CHK(hipMemcpy(h.data(), cyc, nw * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  std::sort(h.begin(), h.end());
  const double n_instr = (double)ITERS * instr_per_iter_group;
  const double med = (double)h[nw / 2];
  // s_memtime ticks at a fixed 100 MHz on gfx9 "realtime"? no: s_memtime = shader clock counter.
  printf("%-18s W=%d  cyc/instr(wave)=%7.2f  issue cyc/instr/SIMD=%6.2f  kernel=%.3f ms  -> %.2f Ginstr/s/SIMD-chip (%.1f G wave-instr/s)\n",
         name, waves_per_simd, med / n_instr, med / n_instr / waves_per_simd, ms,
         0.0, (double)nw * n_instr / (ms * 1e-3) / 1e9);
  return 0;
}

int main() {
  double* out; unsigned long long* cyc;
  CHK(hipMalloc(&out, sizeof(double) * 256 * 8 * 1024));
  CHK(hipMalloc(&cyc, sizeof(unsigned long long) * 256 * 8 * 16));
  struct { const char* n; kern_t k; double per; } ks[] = {
      {"v_fma_f64", k_fma64, 16}, {"v_fmac_f64", k_fmac64, 16}, {"v_mul_f64", k_mul64, 16}, {"v_add_f64", k_add64, 16},
      {"v_rsq_f64", k_rsq64, 16}, {"v_rcp_f64", k_rcp64, 16}, {"v_sqrt_f64", k_sqrt64, 16}, {"v_mov_b64", k_mov64, 16},
      {"v_trunc_f64", k_trunc64, 16}, {"v_ldexp_f64", k_ldexp64, 16}, {"v_cmp_gt_f64", k_cmp64, 16},
      {"v_cndmask_b32", k_cndmask, 16}, {"v_cvt_f32_f64", k_cvt_f32_f64, 16}, {"v_cvt_f64_f32", k_cvt_f64_f32, 16},
      {"v_fma_f32", k_fma32, 16}, {"v_pk_fma_f32", k_pkfma32, 16}, {"v_rsq_f32", k_rsq32, 16},
      {"v_fma_f64 dep", k_fma64_dep, 16}, {"ds_read_b128 bc", k_lds_b128_bcast, 16},
  };
  for (auto& k : ks)
    for (int w : {1, 2, 4})
      if (run(k.n, k.k, w, out, cyc, k.per)) return 1;
  return 0;
}

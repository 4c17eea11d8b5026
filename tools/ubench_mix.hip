// ubench_mix.hip -- what does an instruction cost when it sits BETWEEN fp64 FMAs?  (gfx950)
// The pair loops are ~85 fp64 VALU instructions plus a handful of 32-bit integer / compare / move instructions and
// six LDS operations per step.  ubench.hip prices every instruction class alone; this one prices them in the mix:
// a loop of 16 independent v_fma_f64 per wave, plus N extra instructions of one class per 16 FMAs, 4 waves per SIMD
// on every CU.  "extra cycles per extra instruction" = (t_mix - t_fma) / N in units of one FMA issue slot.
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o /tmp/ubench_mix tools/ubench_mix.hip && /tmp/ubench_mix
#include <hip/hip_runtime.h>
#include <cstdio>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITERS = 4096;

#define FMA8                                                              \
  asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a0) : "v"(x), "v"(y));   \
  asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a1) : "v"(x), "v"(y));   \
  asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a2) : "v"(x), "v"(y));   \
  asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a3) : "v"(x), "v"(y));   \
  asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a4) : "v"(x), "v"(y));   \
  asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a5) : "v"(x), "v"(y));   \
  asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a6) : "v"(x), "v"(y));   \
  asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a7) : "v"(x), "v"(y));

// EXTRA is executed twice per iteration (once per group of 8 FMAs)
#define KERNEL(NAME, EXTRA)                                                                        \
  __global__ __launch_bounds__(256) void NAME(double* out, double x, double y, int seed) {         \
    __shared__ double2 lds[256 * 4];                                                               \
    __shared__ double acc[256 * 4];                                                                \
    for (int i = threadIdx.x; i < 1024; i += 256) { lds[i] = make_double2(x, y); acc[i] = 0.0; }   \
    __syncthreads();                                                                               \
    double a0 = threadIdx.x * 1e-3 + 1.0, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4,      \
           a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;                                                  \
    int i0 = seed + threadIdx.x, i1 = seed * 3 + 1, i2 = 5, i3 = 7;                                \
    double d0 = x, d1 = y;                                                                         \
    double2 q0 = make_double2(0, 0);                                                               \
    unsigned la = (threadIdx.x * 16) & 16383, lb = (threadIdx.x * 8) & 8191;                       \
    for (int it = 0; it < ITERS; ++it) {                                                           \
      FMA8 EXTRA FMA8 EXTRA                                                                        \
    }                                                                                              \
    out[blockIdx.x * blockDim.x + threadIdx.x] =                                                   \
        a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + i0 + i1 + i2 + i3 + d0 + d1 + q0.x + q0.y + acc[threadIdx.x]; \
  }

#define NOTHING
KERNEL(k_base, NOTHING)
KERNEL(k_add_u32_x2, asm volatile("v_add_u32 %0, %0, %1" : "+v"(i0) : "v"(i1)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(i2) : "v"(i3));)
KERNEL(k_add_u32_x4, asm volatile("v_add_u32 %0, %0, %1" : "+v"(i0) : "v"(i1)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(i2) : "v"(i3));
                     asm volatile("v_and_b32 %0, 63, %0" : "+v"(i1)); asm volatile("v_mad_u32_u24 %0, %1, 48, %0" : "+v"(i3) : "v"(i0));)
KERNEL(k_mov_b32_x2, asm volatile("v_mov_b32 %0, %1" : "=v"(i0) : "v"(i1)); asm volatile("v_mov_b32 %0, %1" : "=v"(i2) : "v"(i3));)
KERNEL(k_mov_b64_x2, asm volatile("v_mov_b64 %0, %1" : "=v"(d0) : "v"(x)); asm volatile("v_mov_b64 %0, %1" : "=v"(d1) : "v"(y));)
KERNEL(k_cmp_f64_x1, asm volatile("v_cmp_ge_f64 vcc, %0, %1" : : "v"(x), "v"(a0) : "vcc");)
KERNEL(k_cmp_f64_x2, asm volatile("v_cmp_ge_f64 vcc, %0, %1" : : "v"(x), "v"(a0) : "vcc"); asm volatile("v_cmp_ge_f64 vcc, %0, %1" : : "v"(y), "v"(a1) : "vcc");)
KERNEL(k_cmp_u32_x2, asm volatile("v_cmp_ge_u32 vcc, %0, %1" : : "v"(i0), "v"(i1) : "vcc"); asm volatile("v_cmp_ge_u32 vcc, %0, %1" : : "v"(i2), "v"(i3) : "vcc");)
KERNEL(k_rsq_f64_x1, asm volatile("v_rsq_f64 %0, %1" : "=v"(d0) : "v"(x));)
KERNEL(k_rsq_f64_x2, asm volatile("v_rsq_f64 %0, %1" : "=v"(d0) : "v"(x)); asm volatile("v_rsq_f64 %0, %1" : "=v"(d1) : "v"(y));)
KERNEL(k_rsq_f32_x2, asm volatile("v_rsq_f32 %0, %1" : "=v"(i0) : "v"(i1)); asm volatile("v_rsq_f32 %0, %1" : "=v"(i2) : "v"(i3));)
KERNEL(k_add_f64_x2, asm volatile("v_add_f64 %0, %0, %1" : "+v"(d0) : "v"(x)); asm volatile("v_add_f64 %0, %0, %1" : "+v"(d1) : "v"(y));)
KERNEL(k_mul_f64_x2, asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d0) : "v"(x)); asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d1) : "v"(y));)
KERNEL(k_fma_f32_x2, asm volatile("v_fma_f32 %0, %1, %1, %0" : "+v"(i0) : "v"(i1)); asm volatile("v_fma_f32 %0, %1, %1, %0" : "+v"(i2) : "v"(i3));)
KERNEL(k_ds_read_x3, asm volatile("ds_read_b128 %0, %1\n ds_read_b128 %0, %1 offset:16\n ds_read_b128 %0, %1 offset:32\n s_waitcnt lgkmcnt(0)" : "=&v"(q0) : "v"(la) : "memory");)
KERNEL(k_ds_add_x3, asm volatile("ds_add_f64 %0, %1\n ds_add_f64 %0, %1 offset:2048\n ds_add_f64 %0, %1 offset:4096" : : "v"(lb), "v"(x) : "memory");)
KERNEL(k_ds_both, asm volatile("ds_read_b128 %0, %1\n ds_read_b128 %0, %1 offset:16\n ds_read_b128 %0, %1 offset:32\n s_waitcnt lgkmcnt(0)" : "=&v"(q0) : "v"(la) : "memory");
                  asm volatile("ds_add_f64 %0, %1\n ds_add_f64 %0, %1 offset:2048\n ds_add_f64 %0, %1 offset:4096" : : "v"(lb), "v"(x) : "memory");)

typedef void (*kern_t)(double*, double, double, int);

static float time_kernel(kern_t k, double* out, int waves_per_simd) {
  const int grid = 256 * waves_per_simd;   // 256 threads = 4 waves = one per SIMD; waves_per_simd blocks per CU
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 6; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, out, 1.0000001, 1e-9, 3);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    if (rep >= 2 && ms < best) best = ms;
  }
  return best;
}

int main() {
  double* out;
  CHK(hipMalloc(&out, sizeof(double) * 256 * 256 * 8));
  struct { const char* n; kern_t k; int extra; } ks[] = {
      {"16 v_fma_f64 (base)", k_base, 0},
      {"+ 2 x 1 v_add_u32 (2 per 16)", k_add_u32_x2, 4}, {"+ add,add,and,mad_u24 (4 per 8)", k_add_u32_x4, 8},
      {"+ 2 v_mov_b32 per 8", k_mov_b32_x2, 4}, {"+ 2 v_mov_b64 per 8", k_mov_b64_x2, 4},
      {"+ 1 v_cmp_ge_f64 per 8", k_cmp_f64_x1, 2}, {"+ 2 v_cmp_ge_f64 per 8", k_cmp_f64_x2, 4}, {"+ 2 v_cmp_ge_u32 per 8", k_cmp_u32_x2, 4},
      {"+ 1 v_rsq_f64 per 8", k_rsq_f64_x1, 2}, {"+ 2 v_rsq_f64 per 8", k_rsq_f64_x2, 4}, {"+ 2 v_rsq_f32 per 8", k_rsq_f32_x2, 4},
      {"+ 2 v_add_f64 per 8", k_add_f64_x2, 4}, {"+ 2 v_mul_f64 per 8", k_mul_f64_x2, 4}, {"+ 2 v_fma_f32 per 8", k_fma_f32_x2, 4},
      {"+ 3 ds_read_b128 per 8", k_ds_read_x3, 6}, {"+ 3 ds_add_f64 per 8", k_ds_add_x3, 6}, {"+ 3 read + 3 add per 8", k_ds_both, 12},
  };
  for (int W : {2, 4}) {
    // warm the clocks
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k_base, dim3(256 * W), dim3(256), 0, 0, out, 1.0000001, 1e-9, 3);
    (void)hipDeviceSynchronize();
    const float base = time_kernel(k_base, out, W);
    const double slot = base / (16.0 * ITERS);          // time of one FMA issue slot per wave-row
    printf("W = %d waves per SIMD: base %.4f ms for %d x 16 FMAs per wave\n", W, base, ITERS);
    for (auto& e : ks) {
      const float ms = time_kernel(e.k, out, W);
      if (e.extra == 0) continue;
      printf("  %-34s %.4f ms  -> %+.2f FMA slots per extra instruction\n", e.n, ms, (ms - base) / (e.extra * ITERS) / slot);
    }
  }
  return 0;
}

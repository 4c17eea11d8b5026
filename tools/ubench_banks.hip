// ubench_banks.hip -- does the VGPR bank of the three 64-bit sources of v_fma_f64 matter?  (gfx950)
// 16 independent v_fma_f64 per iteration with explicit registers, 4 waves per SIMD on every CU.
//   A  sources v[4:5], v[8:9], v[12:13]      (all three start in bank 0)
//   B  sources v[4:5], v[6:7], v[8:9]        (banks 0, 2, 0)
//   C  sources v[4:5], v[6:7], acc           (dst = src2: the v_fmac form the compiler prefers)
//   D  two sources the same register: v[4:5], v[4:5], v[8:9]
//   E  sources v[4:5], v[6:7], v[10:11]      (banks 0, 2, 2)   (64-bit VGPR tuples are even-aligned on gfx90a+)
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o /tmp/ubench_banks tools/ubench_banks.hip && /tmp/ubench_banks
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int ITERS = 4096;

#define DST16(OP)                                                                                         \
  OP("v[20:21]") OP("v[22:23]") OP("v[24:25]") OP("v[26:27]") OP("v[28:29]") OP("v[30:31]") OP("v[32:33]") OP("v[34:35]") \
  OP("v[36:37]") OP("v[38:39]") OP("v[40:41]") OP("v[42:43]") OP("v[44:45]") OP("v[46:47]") OP("v[48:49]") OP("v[50:51]")

#define A_(D) "v_fma_f64 " D ", v[4:5], v[8:9], v[12:13]\n"
#define B_(D) "v_fma_f64 " D ", v[4:5], v[6:7], v[8:9]\n"
#define C_(D) "v_fma_f64 " D ", v[4:5], v[6:7], " D "\n"
#define D_(D) "v_fma_f64 " D ", v[4:5], v[4:5], v[8:9]\n"
#define E_(D) "v_fma_f64 " D ", v[4:5], v[6:7], v[10:11]\n"

#define CLOB "v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31", \
  "v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51"

#define KERNEL(NAME, BODY)                                                                     \
  __global__ __launch_bounds__(256) void NAME(double* out, double x) {                         \
    asm volatile("v_mov_b32 v4, %0\n v_mov_b32 v5, %1\n v_mov_b32 v6, %0\n v_mov_b32 v7, %1\n"  \
                 "v_mov_b32 v8, %0\n v_mov_b32 v9, %1\n v_mov_b32 v10, %0\n v_mov_b32 v11, %1\n" \
                 "v_mov_b32 v12, %0\n v_mov_b32 v13, %1\n v_mov_b32 v14, %1\n"                  \
                 : : "v"(__double2loint(x)), "v"(__double2hiint(x)) : CLOB);                    \
    for (int it = 0; it < ITERS; ++it) asm volatile(DST16(BODY) : : : CLOB);                   \
    double r;                                                                                  \
    asm volatile("v_mov_b64 %0, v[20:21]" : "=v"(r) : : CLOB);                                 \
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;                                            \
  }

KERNEL(kA, A_) KERNEL(kB, B_) KERNEL(kC, C_) KERNEL(kD, D_) KERNEL(kE, E_)

typedef void (*kern_t)(double*, double);
int main() {
  double* out;
  if (hipMalloc(&out, sizeof(double) * 256 * 1024 * 4) != hipSuccess) return 1;
  struct { const char* n; kern_t k; } ks[] = {{"A  v[4:5], v[8:9], v[12:13]  (same bank)", kA}, {"B  v[4:5], v[6:7], v[8:9]", kB},
                                             {"C  v[4:5], v[6:7], dst      (fmac form)", kC}, {"D  v[4:5], v[4:5], v[8:9]", kD},
                                             {"E  v[4:5], v[6:7], v[10:11]", kE}};
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 30; ++i) hipLaunchKernelGGL(kC, dim3(1024), dim3(256), 0, 0, out, 1.0000001);
  (void)hipDeviceSynchronize();
  for (auto& e : ks) {
    float best = 1e30f;
    for (int rep = 0; rep < 6; ++rep) {
      (void)hipEventRecord(e0);
      hipLaunchKernelGGL(e.k, dim3(1024), dim3(256), 0, 0, out, 1.0000001);
      (void)hipEventRecord(e1);
      (void)hipDeviceSynchronize();
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      if (rep >= 2 && ms < best) best = ms;
    }
    const double instr = 1024.0 * 4 * ITERS * 16;
    printf("%-44s %.4f ms  %.1f G wave-instr/s\n", e.n, best, instr / (best * 1e-3) / 1e9);
  }
  return 0;
}

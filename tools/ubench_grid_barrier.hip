// ubench_grid_barrier.hip -- what does a grid-wide barrier between the O(N) stages of an Arnoldi step cost, against the
// ~4.5 us floor of a dependent tiny launch?  G workgroups of 256 threads, all resident (G <= 256 CUs), B barriers per launch,
// a little work (one cache line per workgroup written and its neighbour's read) between barriers so that the barrier also
// has to publish data.  Every wait is BOUNDED: a workgroup that spins more than `kMaxSpin` times sets a flag and goes on, so
// the grid always drains.     hipcc --offload-arch=gfx950 -O3 -o ubench_grid_barrier ubench_grid_barrier.hip
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr unsigned kMaxSpin = 4u << 20;

// sense-free counting barrier: the counter only grows; barrier k is passed when it reaches (k + 1) * G
__device__ __forceinline__ void grid_barrier(unsigned* counter, unsigned target, unsigned* failed) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
      if (++spins > kMaxSpin) { *failed = 1u; break; }
      __builtin_amdgcn_s_sleep(1);
    }
  }
  __syncthreads();
}

__global__ __launch_bounds__(256) void barrier_kernel(unsigned* counter, unsigned base, int n_barriers, double* data, unsigned* failed) {
  const unsigned G = gridDim.x;
  double v = (double)blockIdx.x;
  for (int k = 0; k < n_barriers; ++k) {
    if (threadIdx.x < 16) data[(size_t)blockIdx.x * 16 + threadIdx.x] = v + k;
    grid_barrier(counter, base + (unsigned)(k + 1) * G, failed);
    if (threadIdx.x < 16) v = data[(size_t)((blockIdx.x + 1) % G) * 16 + threadIdx.x];
  }
  if (threadIdx.x == 0) data[(size_t)G * 16 + blockIdx.x] = v;
}

__global__ __launch_bounds__(256) void tiny_kernel(double* data, int k) {
  if (threadIdx.x < 16) data[(size_t)blockIdx.x * 16 + threadIdx.x] += (double)k;
}

int main() {
  unsigned *counter, *failed;
  double* data;
  CK(hipMalloc(&counter, 64));
  CK(hipMalloc(&failed, 64));
  CK(hipMalloc(&data, 1 << 20));
  CK(hipMemset(counter, 0, 64));
  CK(hipMemset(failed, 0, 64));
  CK(hipMemset(data, 0, 1 << 20));
  hipStream_t st;
  CK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  // prime the clocks
  for (int i = 0; i < 20000; ++i) hipLaunchKernelGGL(tiny_kernel, dim3(64), dim3(256), 0, st, data, i);
  CK(hipStreamSynchronize(st));
  unsigned base = 0;
  for (int G : {8, 32, 64, 128, 256}) {
    for (int B : {0, 1, 4, 16}) {
      const int reps = 400;
      // warm
      for (int i = 0; i < 20; ++i) { hipLaunchKernelGGL(barrier_kernel, dim3(G), dim3(256), 0, st, counter, base, B, data, failed); base += (unsigned)(B * G); }
      CK(hipEventRecord(e0, st));
      for (int i = 0; i < reps; ++i) { hipLaunchKernelGGL(barrier_kernel, dim3(G), dim3(256), 0, st, counter, base, B, data, failed); base += (unsigned)(B * G); }
      CK(hipEventRecord(e1, st));
      CK(hipEventSynchronize(e1));
      float ms = 0;
      CK(hipEventElapsedTime(&ms, e0, e1));
      unsigned f = 0;
      CK(hipMemcpy(&f, failed, 4, hipMemcpyDeviceToHost));
      printf("G %3d workgroups, %2d barriers per launch: %7.2f us per launch%s\n", G, B, ms * 1e3 / reps, f ? "   (A WAIT GAVE UP)" : "");
      fflush(stdout);
      if (f) return 1;
      if (base > 0xE0000000u) { CK(hipMemset(counter, 0, 64)); base = 0; }
    }
  }
  // the alternative: B + 1 dependent tiny launches
  for (int G : {64, 256}) {
    const int reps = 400;
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < reps * 5; ++i) hipLaunchKernelGGL(tiny_kernel, dim3(G), dim3(256), 0, st, data, i);
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("G %3d workgroups, 5 dependent tiny launches: %7.2f us (%.2f us each)\n", G, ms * 1e3 / reps, ms * 1e3 / reps / 5);
  }
  return 0;
}

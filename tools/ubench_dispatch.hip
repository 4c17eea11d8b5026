// ubench_dispatch.hip -- how fast does gfx950 START workgroups?  (A number the guides do not list; the pair shards of
// a small suspension are a one-round launch whose waves start over ~20 us.)
// An almost empty kernel (one store per wave) is launched with B workgroups of T threads and D bytes of dynamic LDS;
// per-wave start stamps (s_memrealtime, 100 MHz) give the time from the first to the last wave start, HIP events the
// launch-to-launch time of back-to-back launches.
// build: hipcc --offload-arch=gfx950 -O3 -o ubench_dispatch tools/ubench_dispatch.hip ; run: ./ubench_dispatch
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

extern __shared__ double dyn_lds[];

template <int VGPRS>
__global__ void start_kernel(long long* stamps, int use_lds) {
  const long long t = __builtin_amdgcn_s_memrealtime();
  // keep VGPRS registers live so that the allocation per wave is what the pair kernels need
  double acc[VGPRS / 2];
#pragma unroll
  for (int k = 0; k < VGPRS / 2; ++k) acc[k] = (double)(threadIdx.x + k);
  if (use_lds) { dyn_lds[threadIdx.x] = acc[0]; __syncthreads(); acc[1] += dyn_lds[(threadIdx.x + 1) % blockDim.x]; }
  double s = 0.0;
#pragma unroll
  for (int k = 0; k < VGPRS / 2; ++k) s += acc[k];
  if ((threadIdx.x & 63) == 0) stamps[((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6] = s == -1.0 ? 0 : t;
}

int main() {
  long long* stamps;
  CHK(hipMalloc(&stamps, sizeof(long long) * 1 << 20));
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  const int threads_list[] = {256, 512, 1024};
  const int lds_list[] = {0, 32 * 1024};
  CHK(hipFuncSetAttribute((const void*)start_kernel<96>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
  CHK(hipFuncSetAttribute((const void*)start_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
  printf("vgprs threads lds_bytes workgroups waves | first->last wave start (us, median of 20) | back-to-back launch period (us) | workgroups/us  waves/us\n");
  for (int vg = 0; vg < 2; ++vg)
    for (int threads : threads_list)
      for (int lds : lds_list)
        for (int total_waves : {1024, 2048, 4096, 5120, 10240}) {
          const int blocks = total_waves / (threads / 64);
          std::vector<double> spans;
          std::vector<long long> h(total_waves);
          for (int rep = 0; rep < 23; ++rep) {
            if (vg) hipLaunchKernelGGL(start_kernel<96>, dim3(blocks), dim3(threads), lds, 0, stamps, lds ? 1 : 0);
            else    hipLaunchKernelGGL(start_kernel<16>, dim3(blocks), dim3(threads), lds, 0, stamps, lds ? 1 : 0);
            CHK(hipDeviceSynchronize());
            CHK(hipMemcpy(h.data(), stamps, sizeof(long long) * total_waves, hipMemcpyDeviceToHost));
            if (rep >= 3) spans.push_back((*std::max_element(h.begin(), h.end()) - *std::min_element(h.begin(), h.end())) * 0.01);
          }
          std::sort(spans.begin(), spans.end());
          const int L = 200;
          CHK(hipEventRecord(e0, 0));
          for (int i = 0; i < L; ++i) {
            if (vg) hipLaunchKernelGGL(start_kernel<96>, dim3(blocks), dim3(threads), lds, 0, stamps, lds ? 1 : 0);
            else    hipLaunchKernelGGL(start_kernel<16>, dim3(blocks), dim3(threads), lds, 0, stamps, lds ? 1 : 0);
          }
          CHK(hipEventRecord(e1, 0));
          CHK(hipEventSynchronize(e1));
          float ms = 0.f;
          CHK(hipEventElapsedTime(&ms, e0, e1));
          const double span = spans[spans.size() / 2], period = ms * 1e3 / L;
          printf("%5d %7d %9d %10d %5d | %8.2f | %8.2f | %7.1f %7.1f\n", vg ? 96 : 16, threads, lds, blocks, total_waves, span, period,
                 span > 0 ? blocks / span : 0.0, span > 0 ? total_waves / span : 0.0);
        }
  return 0;
}

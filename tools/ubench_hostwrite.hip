// ubench_hostwrite.hip -- how should the synchronous host entry point (rmb_matvec) hand 24 N bytes back to the caller?
//   A  hipMemcpyAsync(device -> pageable host) + hipStreamSynchronize            (what rmb_matvec does today)
//   B  hipMemcpyAsync(device -> pinned host)   + hipStreamSynchronize + memcpy
//   C  the producing kernel stores straight into pinned, device-mapped host memory (16 B per lane, coalesced), then every
//      workgroup releases at system scope and bumps a counter in the same host allocation; the host SPINS on the counter
//      (no runtime call on the critical path) and memcpy()s the result out
// plus a check for after-effects of C on an ordinary device-memory kernel (round 4 saw the ordinary path slow down after a
// mapped-memory pass and dropped the idea unexplained).   hipcc --offload-arch=gfx950 -O3 -o ubench_hostwrite ubench_hostwrite.hip
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// "finalize": out[i] = acc[i] * s for 3n doubles; two doubles (16 B) per lane, consecutive lanes consecutive addresses
__global__ __launch_bounds__(256) void fin_kernel(const double* acc, double* out, long n3, double s, unsigned* done) {
  const long i = 2 * ((long)blockIdx.x * blockDim.x + threadIdx.x);
  if (i + 1 < n3) {
    const double2 v = *reinterpret_cast<const double2*>(acc + i);
    *reinterpret_cast<double2*>(out + i) = make_double2(v.x * s, v.y * s);
  } else if (i < n3) out[i] = acc[i] * s;
  if (done) {
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(done, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// stand-in for the sweep: ~`iters` dependent FMAs per thread on device memory
__global__ __launch_bounds__(256) void work_kernel(double* acc, long n3, int iters) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n3) return;
  double x = acc[i], y = 1.0000001;
  for (int k = 0; k < iters; ++k) x = __builtin_fma(x, y, 1e-9);
  acc[i] = x;
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
  const long n = argc > 1 ? atol(argv[1]) : 10000;
  const long n3 = 3 * n;
  const size_t bytes = n3 * sizeof(double);
  const int reps = 300;
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  double *acc, *dout, *pinned, *mapped;
  CK(hipMalloc(&acc, bytes)); CK(hipMalloc(&dout, bytes));
  CK(hipMemset(acc, 0, bytes));
  CK(hipHostMalloc(&pinned, bytes, hipHostMallocDefault));
  CK(hipHostMalloc(&mapped, bytes + 4096, hipHostMallocMapped));
  unsigned* done = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(mapped) + bytes + 64);
  double* dmapped; CK(hipHostGetDevicePointer((void**)&dmapped, mapped, 0));
  unsigned* ddone = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(dmapped) + bytes + 64);
  std::vector<double> pageable(n3);
  const unsigned blocks = (unsigned)((n3 / 2 + 256) / 256);
  const int iters = 20000;      // ~100+ us of "sweep"
  auto work = [&]() { hipLaunchKernelGGL(work_kernel, dim3((unsigned)((n3 + 255) / 256)), dim3(256), 0, s, acc, n3, iters); };
  auto time_work = [&]() {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int k = 0; k < 5; ++k) work();
    CK(hipEventRecord(e0, s));
    for (int k = 0; k < 20; ++k) work();
    CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3 / 20;
  };
  printf("n = %ld (%zu bytes), work kernel alone: %.1f us\n", n, bytes, time_work());
  for (int pass = 0; pass < 2; ++pass) {
    // A
    double t0 = now_us();
    for (int r = 0; r < reps; ++r) {
      work();
      hipLaunchKernelGGL(fin_kernel, dim3(blocks), dim3(256), 0, s, acc, dout, n3, 1.0, (unsigned*)nullptr);
      CK(hipMemcpyAsync(pageable.data(), dout, bytes, hipMemcpyDeviceToHost, s));
      CK(hipStreamSynchronize(s));
    }
    const double tA = (now_us() - t0) / reps;
    // B
    t0 = now_us();
    for (int r = 0; r < reps; ++r) {
      work();
      hipLaunchKernelGGL(fin_kernel, dim3(blocks), dim3(256), 0, s, acc, dout, n3, 1.0, (unsigned*)nullptr);
      CK(hipMemcpyAsync(pinned, dout, bytes, hipMemcpyDeviceToHost, s));
      CK(hipStreamSynchronize(s));
      memcpy(pageable.data(), pinned, bytes);
    }
    const double tB = (now_us() - t0) / reps;
    // C
    t0 = now_us();
    for (int r = 0; r < reps; ++r) {
      *reinterpret_cast<volatile unsigned*>(done) = 0;
      work();
      hipLaunchKernelGGL(fin_kernel, dim3(blocks), dim3(256), 0, s, acc, dmapped, n3, 1.0, ddone);
      while (__atomic_load_n(done, __ATOMIC_ACQUIRE) != blocks) { }
      memcpy(pageable.data(), mapped, bytes);
    }
    const double tC = (now_us() - t0) / reps;
    CK(hipStreamSynchronize(s));
    // C2: as C but wait with hipStreamSynchronize
    t0 = now_us();
    for (int r = 0; r < reps; ++r) {
      work();
      hipLaunchKernelGGL(fin_kernel, dim3(blocks), dim3(256), 0, s, acc, dmapped, n3, 1.0, (unsigned*)nullptr);
      CK(hipStreamSynchronize(s));
      memcpy(pageable.data(), mapped, bytes);
    }
    const double tC2 = (now_us() - t0) / reps;
    printf("pass %d: per call  A pageable D2H + sync %.1f us | B pinned D2H + sync + memcpy %.1f | C mapped stores + spin + memcpy %.1f | C2 mapped stores + stream sync + memcpy %.1f\n",
           pass, tA, tB, tC, tC2);
    printf("        work kernel alone afterwards: %.1f us\n", time_work());
  }
  return 0;
}

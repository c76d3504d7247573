// Host link probe (tuning aid): how fast pinned host memory moves to and from
// the GPU when both directions run at once, by the copy engines
// (hipMemcpyAsync) and by kernels that read / write the pinned memory directly.
// Build: hipcc -O3 --offload-arch=gfx950 tools/link_probe.hip -o tools/bin/link_probe
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void copy_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
  const size_t in_bytes = 768ull << 20, out_bytes = 1792ull << 20;  // the shape of a stripe: 0.8 GB in, 1.8 GB out
  void *h_in, *h_out, *d_in, *d_out;
  CHECK(hipHostMalloc(&h_in, in_bytes, hipHostMallocDefault));
  CHECK(hipHostMalloc(&h_out, out_bytes, hipHostMallocDefault));
  CHECK(hipMalloc(&d_in, in_bytes));
  CHECK(hipMalloc(&d_out, out_bytes));
  memset(h_in, 1, in_bytes);
  memset(h_out, 0, out_bytes);
  CHECK(hipMemset(d_out, 2, out_bytes));
  hipStream_t s1, s2;
  CHECK(hipStreamCreate(&s1));
  CHECK(hipStreamCreate(&s2));
  hipEvent_t a1, b1, a2, b2;
  CHECK(hipEventCreate(&a1)); CHECK(hipEventCreate(&b1)); CHECK(hipEventCreate(&a2)); CHECK(hipEventCreate(&b2));
  const int blocks = 256;  // few workgroups: the link, not the chip, is the limit
  auto run = [&](const char* name, bool in_kernel, bool out_kernel, bool do_in, bool do_out) {
    for (int rep = 0; rep < 3; ++rep) {
      CHECK(hipDeviceSynchronize());
      const double t0 = now();
      if (do_in) {
        CHECK(hipEventRecord(a1, s1));
        if (in_kernel) hipLaunchKernelGGL(copy_kernel, dim3(blocks), dim3(256), 0, s1, (const uint4*)h_in, (uint4*)d_in, in_bytes / 16);
        else CHECK(hipMemcpyAsync(d_in, h_in, in_bytes, hipMemcpyHostToDevice, s1));
        CHECK(hipEventRecord(b1, s1));
      }
      if (do_out) {
        CHECK(hipEventRecord(a2, s2));
        if (out_kernel) hipLaunchKernelGGL(copy_kernel, dim3(blocks), dim3(256), 0, s2, (const uint4*)d_out, (uint4*)h_out, out_bytes / 16);
        else CHECK(hipMemcpyAsync(h_out, d_out, out_bytes, hipMemcpyDeviceToHost, s2));
        CHECK(hipEventRecord(b2, s2));
      }
      CHECK(hipDeviceSynchronize());
      const double dt = now() - t0;
      float m1 = 0, m2 = 0;
      if (do_in) CHECK(hipEventElapsedTime(&m1, a1, b1));
      if (do_out) CHECK(hipEventElapsedTime(&m2, a2, b2));
      if (rep == 2)
        printf("%-46s in %6.1f GB/s (%6.1f ms)  out %6.1f GB/s (%6.1f ms)  wall %6.1f ms  both %6.1f GB/s\n", name,
               do_in ? in_bytes / (m1 * 1e6) : 0.0, m1, do_out ? out_bytes / (m2 * 1e6) : 0.0, m2, dt * 1e3,
               ((do_in ? in_bytes : 0) + (do_out ? out_bytes : 0)) / (dt * 1e9));
    }
  };
  run("copy engine in, alone", false, false, true, false);
  run("copy engine out, alone", false, false, false, true);
  run("kernel in, alone", true, false, true, false);
  run("kernel out, alone", false, true, false, true);
  run("copy engine in + copy engine out", false, false, true, true);
  run("kernel in + copy engine out", true, false, true, true);
  run("copy engine in + kernel out", false, true, true, true);
  run("kernel in + kernel out", true, true, true, true);
  if (((unsigned char*)h_out)[12345] != 2) { fprintf(stderr, "copy check failed\n"); return 1; }
  return 0;
}

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
  const hipStream_t s1_plain = s1, s2_plain = s2;
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
  // The pattern of a loop over stripes: the copy back of one stripe is already queued in slices on one stream when
  // the copy in of the next stripe is issued, also in slices, on another.  When does the copy in run?
  hipStream_t s3, s4;  // two streams that each carry both directions, first used in the order of a decoder's two slots
  CHECK(hipStreamCreateWithFlags(&s3, hipStreamNonBlocking));
  CHECK(hipStreamCreateWithFlags(&s4, hipStreamNonBlocking));
  auto sliced = [&](const char* name, bool in_kernel, bool out_kernel, bool mixed = false) {
    hipStream_t s1 = mixed ? s4 : ::hipStream_t(nullptr), s2 = s3;
    if (!mixed) { s1 = s1_plain; s2 = s2_plain; }
    const size_t out_slice = 32ull << 20, in_slice = 22ull << 20;
    for (int rep = 0; rep < 2; ++rep) {
      CHECK(hipDeviceSynchronize());
      const double t0 = now();
      CHECK(hipEventRecord(a2, s2));
      for (size_t off = 0; off < out_bytes; off += out_slice) {
        const size_t len = out_bytes - off < out_slice ? out_bytes - off : out_slice;
        if (out_kernel) hipLaunchKernelGGL(copy_kernel, dim3(blocks), dim3(256), 0, s2, (const uint4*)((char*)d_out + off), (uint4*)((char*)h_out + off), len / 16);
        else CHECK(hipMemcpyAsync((char*)h_out + off, (char*)d_out + off, len, hipMemcpyDeviceToHost, s2));
      }
      CHECK(hipEventRecord(b2, s2));
      while (now() - t0 < 2e-3) {}
      CHECK(hipEventRecord(a1, s1));
      for (size_t off = 0; off < in_bytes; off += in_slice) {
        const size_t len = in_bytes - off < in_slice ? in_bytes - off : in_slice;
        if (in_kernel) hipLaunchKernelGGL(copy_kernel, dim3(blocks), dim3(256), 0, s1, (const uint4*)((char*)h_in + off), (uint4*)((char*)d_in + off), len / 16);
        else CHECK(hipMemcpyAsync((char*)d_in + off, (char*)h_in + off, len, hipMemcpyHostToDevice, s1));
      }
      CHECK(hipEventRecord(b1, s1));
      CHECK(hipDeviceSynchronize());
      const double dt = now() - t0;
      float out_ms, in_ms, in_start, in_end;
      CHECK(hipEventElapsedTime(&out_ms, a2, b2));
      CHECK(hipEventElapsedTime(&in_ms, a1, b1));
      CHECK(hipEventElapsedTime(&in_start, a2, a1));
      CHECK(hipEventElapsedTime(&in_end, a2, b1));
      if (rep == 1)
        printf("%-46s out 0 .. %5.1f ms, in %5.1f .. %5.1f ms (%5.1f ms), wall %5.1f ms\n", name, out_ms, in_start, in_end, in_ms, dt * 1e3);
    }
  };
  sliced("sliced: copy engine out, then copy engine in", false, false);
  sliced("sliced: copy engine out, then kernel in", true, false);
  sliced("sliced: kernel out, then copy engine in", false, true);
  for (hipStream_t st : {s3, s4}) {  // slot by slot: a copy in, then a copy out
    CHECK(hipMemcpyAsync(d_in, h_in, 1 << 20, hipMemcpyHostToDevice, st));
    CHECK(hipMemcpyAsync(h_out, d_out, 1 << 20, hipMemcpyDeviceToHost, st));
    CHECK(hipStreamSynchronize(st));
  }
  sliced("sliced, streams that carry both directions", false, false, true);
  {  // streams that copy one way only, made after others have taken their engines
    hipStream_t s5, s6;
    CHECK(hipStreamCreateWithFlags(&s5, hipStreamNonBlocking));
    CHECK(hipStreamCreateWithFlags(&s6, hipStreamNonBlocking));
    s1 = s5;
    s2 = s6;
    run("late one-way streams: copy engine in + out", false, false, true, true);
  }
  if (((unsigned char*)h_out)[12345] != 2) { fprintf(stderr, "copy check failed\n"); return 1; }
  return 0;
}

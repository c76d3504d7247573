// Do two launches on ONE stream overlap when the second is launched with
// hipExtAnyOrderLaunch (its dispatch packet does not wait for the one in front)?
// Tuning aid; build: hipcc --offload-arch=gfx950 -O3 tools/anyorder_probe.hip -o tools/bin/anyorder_probe
//
// k_busy keeps `blocks` workgroups of 64 threads occupied for about `spin`
// clock ticks each and then writes one word per thread, so two launches of
// 256 workgroups fit the chip side by side.  Printed: wall time of
//   A ; B            both ordinary          (back to back)
//   A ; B(any order)                        (side by side if the flag is honoured)
//   P ; A ; B(any order), B reading what P wrote: the order P -> B must still hold,
//   because A, in front of B, waits for P.
// and a check that what the any-order launch wrote is seen by a later ordinary
// launch and by a copy to the host after the stream has been synchronised.
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <chrono>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(64) void k_busy(uint32_t* dst, const uint32_t* src, uint32_t spin, uint32_t tag) {
  const uint64_t t0 = __builtin_readcyclecounter();
  uint32_t v = src ? src[blockIdx.x * 64 + threadIdx.x] : 0u;
  while (__builtin_readcyclecounter() - t0 < spin) __builtin_amdgcn_s_sleep(8);
  dst[blockIdx.x * 64 + threadIdx.x] = v + tag;
}

static double now_us() {
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static void launch(uint32_t* dst, const uint32_t* src, uint32_t blocks, uint32_t spin, uint32_t tag, hipStream_t s, bool any) {
  hipExtLaunchKernelGGL(k_busy, dim3(blocks), dim3(64), 0, s, nullptr, nullptr, any ? hipExtAnyOrderLaunch : 0u, dst, src, spin, tag);
  CK(hipGetLastError());
}

int main(int argc, char** argv) {
  const uint32_t blocks = argc > 1 ? atoi(argv[1]) : 256;
  const uint32_t spin = argc > 2 ? atoi(argv[2]) : 200000;  // ticks of the 100 MHz counter: 2 ms
  const size_t n = (size_t)blocks * 64;
  uint32_t *a, *b, *p, *c;
  CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4)); CK(hipMalloc(&p, n * 4)); CK(hipMalloc(&c, n * 4));
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  std::vector<uint32_t> h(n);
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipMemsetAsync(a, 0, n * 4, s)); CK(hipMemsetAsync(b, 0, n * 4, s)); CK(hipMemsetAsync(p, 0, n * 4, s)); CK(hipMemsetAsync(c, 0, n * 4, s));
    CK(hipStreamSynchronize(s));
    double t = now_us();
    launch(a, nullptr, blocks, spin, 1, s, false);
    launch(b, nullptr, blocks, spin, 2, s, false);
    CK(hipStreamSynchronize(s));
    const double serial = now_us() - t;
    t = now_us();
    launch(a, nullptr, blocks, spin, 1, s, false);
    launch(b, nullptr, blocks, spin, 2, s, true);
    CK(hipStreamSynchronize(s));
    const double any = now_us() - t;
    // P writes 100 everywhere; A ordinary (waits for P); B any order reads P's words; C ordinary reads B's
    t = now_us();
    launch(p, nullptr, blocks, spin / 4, 100, s, false);
    launch(a, nullptr, blocks, spin, 1, s, false);
    launch(b, p, blocks, spin, 2, s, true);
    launch(c, b, blocks, 0, 1000, s, false);
    CK(hipStreamSynchronize(s));
    const double chain = now_us() - t;
    CK(hipMemcpy(h.data(), b, n * 4, hipMemcpyDeviceToHost));
    size_t bad_b = 0, bad_c = 0;
    for (size_t i = 0; i < n; ++i) bad_b += h[i] != 102;
    CK(hipMemcpy(h.data(), c, n * 4, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n; ++i) bad_c += h[i] != 1102;
    printf("blocks %u: A;B %.0f us   A;B(any order) %.0f us   P;A;B(any, reads P);C(reads B) %.0f us   wrong words: B %zu, C %zu\n", blocks, serial, any, chain,
           bad_b, bad_c);
  }
  return 0;
}

// Why does a launch of one-wave workgroups fill memory more slowly than one of
// four-wave workgroups?  Tuning aid; build:
//   hipcc --offload-arch=gfx950 -O3 tools/fill_probe.hip -o tools/bin/fill_probe
//
// 2 GiB written in 16 KiB tiles with non-temporal 16-byte stores, nothing read:
//   A  256 threads per workgroup, one tile per workgroup (four stores per thread)
//   B   64 threads per workgroup, one tile per workgroup (sixteen stores per thread)
//   C   64 threads, K consecutive tiles per workgroup
//   D   64 threads, K tiles per workgroup a grid apart
//   F  256 threads per workgroup, every wave fills a tile of its own: the tiles in
//      flight of B with the workgroups of A
//   G  256 threads, every wave its own tile, K tiles per wave a grid apart
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr uint32_t TILE = 16384;

__device__ __forceinline__ void fill_tile_wave(u32x4* tile, uint32_t lane, uint32_t v) {
  const u32x4 q = {v, v + 1, v + 2, v + 3};
#pragma unroll
  for (uint32_t i = 0; i < 16; ++i) __builtin_nontemporal_store(q, tile + i * 64 + lane);
}

__global__ __launch_bounds__(256, 8) void k_a(u32x4* dst, uint32_t n_tiles) {
  __shared__ uint8_t pad[19000];
  if (n_tiles == 0xFFFFFFFFu) pad[threadIdx.x] = 1;
  u32x4* tile = dst + (uint64_t)blockIdx.x * (TILE / 16);
  const u32x4 q = {blockIdx.x, 1, 2, 3};
#pragma unroll
  for (uint32_t i = 0; i < 4; ++i) __builtin_nontemporal_store(q, tile + i * 256 + threadIdx.x);
}

template <int K, bool STRIDED>
__global__ __launch_bounds__(64, 8) void k_wave(u32x4* dst, uint32_t n_tiles) {
  __shared__ uint8_t pad[4800];
  if (n_tiles == 0xFFFFFFFFu) pad[threadIdx.x] = 1;
#pragma unroll
  for (uint32_t k = 0; k < K; ++k) {
    const uint32_t t = STRIDED ? blockIdx.x + k * gridDim.x : blockIdx.x * K + k;
    if (t < n_tiles) fill_tile_wave(dst + (uint64_t)t * (TILE / 16), threadIdx.x, t);
  }
}

template <int K>
__global__ __launch_bounds__(256, 8) void k_four(u32x4* dst, uint32_t n_tiles) {
  __shared__ uint8_t pad[19000];
  if (n_tiles == 0xFFFFFFFFu) pad[threadIdx.x] = 1;
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (uint32_t k = 0; k < K; ++k) {
    const uint32_t t = (blockIdx.x + k * gridDim.x) * 4 + wave;
    if (t < n_tiles) fill_tile_wave(dst + (uint64_t)t * (TILE / 16), lane, t);
  }
}

template <typename F>
static void timed(const char* what, uint64_t bytes, F&& launch) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9f;
  for (int rep = 0; rep < 6; ++rep) {
    CK(hipEventRecord(e0));
    launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  printf("  %-64s %7.1f us  %7.1f GB/s  %5.2f ns per tile\n", what, best * 1e3, bytes / (best * 1e-3) / 1e9, best * 1e6 / (bytes / TILE));
}

int main() {
  const uint64_t bytes = 2ull << 30;
  const uint32_t n = (uint32_t)(bytes / TILE);
  u32x4* d;
  CK(hipMalloc(&d, bytes + TILE));
  timed("A 256 threads, one tile per workgroup", bytes, [&] { k_a<<<n, 256>>>(d, n); });
  timed("B  64 threads, one tile per workgroup", bytes, [&] { k_wave<1, false><<<n, 64>>>(d, n); });
  timed("C  64 threads, 2 consecutive tiles per workgroup", bytes, [&] { k_wave<2, false><<<n / 2, 64>>>(d, n); });
  timed("C  64 threads, 4 consecutive tiles per workgroup", bytes, [&] { k_wave<4, false><<<n / 4, 64>>>(d, n); });
  timed("D  64 threads, 2 tiles per workgroup a grid apart", bytes, [&] { k_wave<2, true><<<n / 2, 64>>>(d, n); });
  timed("D  64 threads, 4 tiles per workgroup a grid apart", bytes, [&] { k_wave<4, true><<<n / 4, 64>>>(d, n); });
  timed("F 256 threads, a tile per wave", bytes, [&] { k_four<1><<<n / 4, 256>>>(d, n); });
  timed("G 256 threads, a tile per wave, 2 per wave a grid apart", bytes, [&] { k_four<2><<<n / 8, 256>>>(d, n); });
  timed("A 256 threads, one tile per workgroup (again)", bytes, [&] { k_a<<<n, 256>>>(d, n); });
  return 0;
}

// What a dependent-load chain in front of a tile's stores costs, and what larger
// tiles buy back (tuning aid; build: hipcc --offload-arch=gfx950 -O3
// tools/chain_probe.hip -o tools/bin/chain_probe).
//
// Every workgroup (256 threads, 19.5 KB of LDS, 8 per CU -- the decoder's shape)
// walks HOPS dependent loads the way a rank-scatter tile does: tile entry
// (scalar) -> partial (scalar) -> bitmap words (vector, 512 bytes) -> values
// (vector, indexed by what the words held), then fills TILE bytes of output
// with non-temporal 16-byte stores.  Output is 2 GiB per run, far past the
// memory-side cache.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t sload32(const uint32_t* p) {
  typedef const uint32_t __attribute__((address_space(4))) * const_ptr;
  return *reinterpret_cast<const_ptr>(reinterpret_cast<uintptr_t>(p));
}

struct Args {
  const uint32_t* entries;  // per tile: index of its partial
  const uint32_t* parts;    // per tile: index of its bitmap block (64 words)
  const uint64_t* words;    // 64 words per tile; low bits = index of its value block
  const uint32_t* values;   // 256 dwords per tile
  u32x4* dst;
  uint32_t n_tiles;
};

template <int TILE, int HOPS>
__global__ __launch_bounds__(256, 8) void k_chain(Args a) {
  __shared__ __align__(16) uint8_t smem[19584];
  uint32_t* s = reinterpret_cast<uint32_t*>(smem);
  uint32_t t = blockIdx.x;
  uint32_t v = t;
  if constexpr (HOPS >= 1) t = sload32(a.entries + blockIdx.x);               // tile entry
  if constexpr (HOPS >= 2) v = sload32(a.parts + t);                          // its partial
  if constexpr (HOPS >= 3) {                                                  // bitmap words, one wave
    if (threadIdx.x < 64) {
      const uint64_t w = a.words[(uint64_t)v * 64 + threadIdx.x];
      s[threadIdx.x] = (uint32_t)w;
    }
    __syncthreads();
    v = s[threadIdx.x & 63];
  }
  if constexpr (HOPS >= 4) {                                                  // values at an index the words gave
    const uint32_t x = a.values[(uint64_t)(v % a.n_tiles) * 256 + threadIdx.x];
    s[64 + threadIdx.x] = x;
    __syncthreads();
    v = s[64 + ((threadIdx.x + 1) & 255)];
  }
  u32x4 o;
  o.x = v; o.y = t; o.z = v ^ t; o.w = 1;
  u32x4* d = a.dst + (uint64_t)blockIdx.x * (TILE / 16);
#pragma unroll
  for (int r = 0; r < TILE / 4096; ++r) __builtin_nontemporal_store(o, d + r * 256 + threadIdx.x);
}

template <int TILE, int HOPS>
static void run(const Args& base, uint64_t bytes) {
  Args a = base;
  a.n_tiles = (uint32_t)(bytes / TILE);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((k_chain<TILE, HOPS>), dim3(a.n_tiles), dim3(256), 0, 0, a);
  CK(hipEventRecord(e0));
  const int reps = 5;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_chain<TILE, HOPS>), dim3(a.n_tiles), dim3(256), 0, 0, a);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= reps;
  printf("tile %3d KiB  hops %d   %8.3f ms  %7.1f GB/s written  %6.2f ns per 16 KiB\n", TILE / 1024, HOPS, ms, bytes / ms / 1e6,
         ms * 1e6 / (bytes / 16384.0));
  CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
}

int main() {
  const uint64_t bytes = 2ull << 30;
  const uint32_t max_tiles = (uint32_t)(bytes / 16384);
  std::vector<uint32_t> entries(max_tiles), parts(max_tiles);
  std::vector<uint64_t> words((size_t)max_tiles * 64);
  // neighbouring tiles read neighbouring entries, as the tiles of one column do
  for (uint32_t i = 0; i < max_tiles; ++i) { entries[i] = i; parts[i] = i; }
  for (size_t i = 0; i < words.size(); ++i) words[i] = (i / 64);
  Args a{};
  uint32_t *d_entries, *d_parts, *d_values;
  uint64_t* d_words;
  u32x4* d_dst;
  CK(hipMalloc(&d_entries, entries.size() * 4)); CK(hipMalloc(&d_parts, parts.size() * 4));
  CK(hipMalloc(&d_words, words.size() * 8)); CK(hipMalloc(&d_values, (size_t)max_tiles * 1024));
  CK(hipMalloc(&d_dst, bytes));
  CK(hipMemcpy(d_entries, entries.data(), entries.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_parts, parts.data(), parts.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_words, words.data(), words.size() * 8, hipMemcpyHostToDevice));
  CK(hipMemset(d_values, 1, (size_t)max_tiles * 1024));
  a.entries = d_entries; a.parts = d_parts; a.words = d_words; a.values = d_values; a.dst = d_dst;
  run<16384, 0>(a, bytes); run<16384, 1>(a, bytes); run<16384, 2>(a, bytes); run<16384, 3>(a, bytes); run<16384, 4>(a, bytes);
  run<32768, 0>(a, bytes); run<32768, 2>(a, bytes); run<32768, 3>(a, bytes); run<32768, 4>(a, bytes);
  run<65536, 0>(a, bytes); run<65536, 2>(a, bytes); run<65536, 3>(a, bytes); run<65536, 4>(a, bytes);
  return 0;
}

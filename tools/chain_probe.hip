// What a dependent-load chain in front of a tile's stores costs, and what buys it
// back (tuning aid; build: hipcc --offload-arch=gfx950 -O3 tools/chain_probe.hip
// -o tools/bin/chain_probe).
//
// Every workgroup writes one 16 KiB tile (non-temporal 16-byte stores) after
// HOPS dependent loads shaped like a rank-scatter tile's: tile entry (scalar)
// -> partial (scalar) -> bitmap words (vector, 512 bytes) -> values (vector,
// at an index the words gave), with WORK vector operations per 16-byte store
// in between.  SPREAD = 1 places every tile's words and values at a hashed
// position in 1 GiB each, so those two hops miss every cache; SPREAD = 0 lays
// them out in tile order.  Two shapes of workgroup:
//   BS = 256: four waves share the tile, 19.5 KiB of LDS, 8 workgroups per CU
//   BS =  64: one wave owns the tile, 5 KiB of LDS, 32 workgroups per CU --
//             four times the chains in flight for the same bytes and waves.
// Output is 2 GiB per run, far past the memory-side cache.
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
  const uint64_t* words;    // 64 words per block; low bits = index of a value block
  const uint32_t* values;   // 256 dwords per block
  u32x4* dst;
  uint32_t n_tiles;
  uint32_t n_blocks;        // blocks of words / values
};

template <int BS, int HOPS, int WORK>
__global__ __launch_bounds__(BS, 8) void k_chain(Args a) {
  constexpr int LDS = BS == 256 ? 19584 : 5120;
  __shared__ __align__(16) uint8_t smem[LDS];
  uint32_t* s = reinterpret_cast<uint32_t*>(smem);
  uint32_t t = blockIdx.x;
  uint32_t v = t;
  if constexpr (HOPS >= 1) t = sload32(a.entries + blockIdx.x);               // tile entry
  if constexpr (HOPS >= 2) v = sload32(a.parts + t);                          // its partial
  if constexpr (HOPS >= 3) {                                                  // bitmap words, one wave
    if (threadIdx.x < 64) {
      const uint64_t w = a.words[(uint64_t)(v % a.n_blocks) * 64 + threadIdx.x];
      s[threadIdx.x] = (uint32_t)w;
    }
    __syncthreads();
    v = s[threadIdx.x & 63];
  }
  if constexpr (HOPS >= 4) {                                                  // values at an index the words gave
    const uint32_t x = a.values[(uint64_t)(v % a.n_blocks) * 256 + (threadIdx.x & 255)];
    s[64 + threadIdx.x] = x;
    __syncthreads();
    v = s[64 + ((threadIdx.x + 1) & (BS - 1))];
  }
  u32x4* d = a.dst + (uint64_t)blockIdx.x * 1024;
#pragma unroll 4
  for (int r = 0; r < 16384 / (BS * 16); ++r) {
    uint32_t x = v + r;
#pragma unroll
    for (int k = 0; k < WORK; ++k) x = x * 0x9E3779B1u + (x >> 7);           // dependent vector operations
    u32x4 o;
    o.x = x; o.y = t; o.z = x ^ t; o.w = r;
    __builtin_nontemporal_store(o, d + r * BS + threadIdx.x);
  }
}

// Four tiles per 256-thread workgroup, one per wave, no workgroup barrier: the
// chains in flight of the one-wave shape at a quarter of its workgroup launches.
template <int HOPS, int WORK>
__global__ __launch_bounds__(256, 8) void k_chain_wave4(Args a) {
  __shared__ __align__(16) uint8_t smem[4 * 4864];
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  uint32_t* s = reinterpret_cast<uint32_t*>(smem + wave * 4864);
  const uint32_t tile = blockIdx.x * 4 + wave;
  uint32_t t = tile;
  uint32_t v = t;
  if constexpr (HOPS >= 1) t = sload32(a.entries + __builtin_amdgcn_readfirstlane(tile));
  if constexpr (HOPS >= 2) v = sload32(a.parts + t);
  if constexpr (HOPS >= 3) {
    const uint64_t w = a.words[(uint64_t)(v % a.n_blocks) * 64 + lane];
    s[lane] = (uint32_t)w;
    __builtin_amdgcn_wave_barrier();
    v = s[(lane + 1) & 63];
  }
  if constexpr (HOPS >= 4) {
    const uint32_t x = a.values[(uint64_t)(v % a.n_blocks) * 256 + lane];
    s[64 + lane] = x;
    __builtin_amdgcn_wave_barrier();
    v = s[64 + ((lane + 1) & 63)];
  }
  u32x4* d = a.dst + (uint64_t)tile * 1024;
#pragma unroll 4
  for (int r = 0; r < 16; ++r) {
    uint32_t x = v + r;
#pragma unroll
    for (int k = 0; k < WORK; ++k) x = x * 0x9E3779B1u + (x >> 7);
    u32x4 o;
    o.x = x; o.y = t; o.z = x ^ t; o.w = r;
    __builtin_nontemporal_store(o, d + r * 64 + lane);
  }
}

template <int HOPS, int WORK>
static void run_wave4(const Args& base, uint64_t bytes, int spread) {
  Args a = base;
  a.n_tiles = (uint32_t)(bytes / 16384);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((k_chain_wave4<HOPS, WORK>), dim3(a.n_tiles / 4), dim3(256), 0, 0, a);
  CK(hipEventRecord(e0));
  const int reps = 5;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_chain_wave4<HOPS, WORK>), dim3(a.n_tiles / 4), dim3(256), 0, 0, a);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= reps;
  printf("4 x 1 wave   hops %d  work %3d  %s   %8.3f ms  %7.1f GB/s written  %6.2f ns per tile\n", HOPS, WORK,
         spread ? "hashed " : "in order", ms, bytes / ms / 1e6, ms * 1e6 / a.n_tiles);
  CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
}

template <int BS, int HOPS, int WORK>
static void run(const Args& base, uint64_t bytes, int spread) {
  Args a = base;
  a.n_tiles = (uint32_t)(bytes / 16384);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((k_chain<BS, HOPS, WORK>), dim3(a.n_tiles), dim3(BS), 0, 0, a);
  CK(hipEventRecord(e0));
  const int reps = 5;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_chain<BS, HOPS, WORK>), dim3(a.n_tiles), dim3(BS), 0, 0, a);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= reps;
  printf("threads %3d  hops %d  work %3d  %s   %8.3f ms  %7.1f GB/s written  %6.2f ns per tile\n", BS, HOPS, WORK,
         spread ? "hashed " : "in order", ms, bytes / ms / 1e6, ms * 1e6 / a.n_tiles);
  CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
}

int main() {
  const uint64_t bytes = 2ull << 30;
  const uint32_t tiles = (uint32_t)(bytes / 16384);
  const uint32_t n_blocks = 1u << 20;  // 1 Mi blocks: 512 MiB of words, 1 GiB of values
  std::vector<uint32_t> entries(tiles), parts(tiles);
  std::vector<uint64_t> words((size_t)n_blocks * 64);
  Args a{};
  uint32_t *d_entries, *d_parts, *d_values;
  uint64_t* d_words;
  u32x4* d_dst;
  CK(hipMalloc(&d_entries, entries.size() * 4)); CK(hipMalloc(&d_parts, parts.size() * 4));
  CK(hipMalloc(&d_words, words.size() * 8)); CK(hipMalloc(&d_values, (size_t)n_blocks * 1024));
  CK(hipMalloc(&d_dst, bytes));
  CK(hipMemset(d_values, 1, (size_t)n_blocks * 1024));
  a.entries = d_entries; a.parts = d_parts; a.words = d_words; a.values = d_values; a.dst = d_dst; a.n_blocks = n_blocks;
  for (int spread = 0; spread < 2; ++spread) {
    for (uint32_t i = 0; i < tiles; ++i) {
      entries[i] = i;
      parts[i] = spread ? (uint32_t)((i * 2654435761u) >> 12) : i;  // the block whose words the tile reads
    }
    for (size_t i = 0; i < words.size(); ++i) {
      const uint32_t b = (uint32_t)(i / 64);
      words[i] = spread ? (uint32_t)((b * 40503u + 977u) * 2246822519u >> 12) : b;  // the block whose values it reads
    }
    CK(hipMemcpy(d_entries, entries.data(), entries.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_parts, parts.data(), parts.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_words, words.data(), words.size() * 8, hipMemcpyHostToDevice));
    run<256, 0, 0>(a, bytes, spread);
    run<256, 4, 0>(a, bytes, spread);
    run<256, 4, 24>(a, bytes, spread);
    run<256, 4, 48>(a, bytes, spread);
    run<64, 0, 0>(a, bytes, spread);
    run<64, 4, 0>(a, bytes, spread);
    run<64, 4, 24>(a, bytes, spread);
    run<64, 4, 48>(a, bytes, spread);
    run_wave4<0, 0>(a, bytes, spread);
    run_wave4<4, 0>(a, bytes, spread);
    run_wave4<4, 24>(a, bytes, spread);
    run_wave4<4, 48>(a, bytes, spread);
  }
  return 0;
}

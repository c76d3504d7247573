// What a gather of {offset, length} string views costs when the alphabet is kept
// in LDS as 16-bit offsets (relative to a base per 256 entries) instead of being
// gathered from memory.  Tuning aid; build:
//   hipcc --offload-arch=gfx950 -O3 tools/dictstr_probe.hip -o tools/bin/dictstr_probe
//
// CHUNKS columns of ROWS rows, each with its own alphabet of A views (8 bytes each,
// contiguous strings shorter than 256 bytes) and 16-bit indices.  A workgroup of
// 256 threads builds the table from the views (A x 8 bytes read), then gathers
// T tiles of 2,048 rows: indices in (the next tile's are requested before this
// one is gathered), views out (non-temporal 16-byte stores).  Compared with the
// same rows gathered from the views in memory (eight workgroups per CU).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr uint32_t TE = 2048;

template <uint32_t AMAX>
__global__ __launch_bounds__(256, 1) void k_lds(const uint64_t* views, const uint16_t* idx, uint64_t* out, uint32_t A, uint32_t rows, uint32_t T, uint32_t groups_per_chunk) {
  constexpr uint32_t NB = AMAX / 256;
  __shared__ __align__(16) uint16_t e16[AMAX + NB + 8];
  __shared__ uint32_t coarse[NB];
  const uint32_t chunk = blockIdx.x / groups_per_chunk, group = blockIdx.x % groups_per_chunk;
  const uint64_t* v = views + (uint64_t)chunk * A;
  for (uint32_t j0 = 0; j0 < A; j0 += 256 * 8) {
    uint64_t e[8], b[8];
#pragma unroll
    for (uint32_t u = 0; u < 8; ++u) {
      const uint32_t j = j0 + u * 256 + threadIdx.x;
      e[u] = v[j < A ? j : A - 1];
      b[u] = v[(j < A ? j : A - 1) & ~255u];
    }
#pragma unroll
    for (uint32_t u = 0; u < 8; ++u) {
      const uint32_t j = j0 + u * 256 + threadIdx.x;
      if (j < A) {
        const uint32_t off = (uint32_t)e[u], len = (uint32_t)(e[u] >> 32), base = (uint32_t)b[u];
        const uint32_t k = j + (j >> 8);
        e16[k] = (uint16_t)(off - base);
        if ((j & 255u) == 255u || j == A - 1) e16[k + 1] = (uint16_t)(off + len - base);
        if ((j & 255u) == 0) coarse[j >> 8] = base;
      }
    }
  }
  __syncthreads();
  const uint32_t n_tiles = (rows + TE - 1) / TE;
  const uint32_t t0 = group * T, t1 = t0 + T < n_tiles ? t0 + T : n_tiles;
  const uint16_t* ix = idx + (uint64_t)chunk * rows;
  uint64_t* dst = out + (uint64_t)chunk * rows;
  const uint32_t* e32 = reinterpret_cast<const uint32_t*>(e16);
  // a thread takes rows (it * 256 + thread) * 2 + {0, 1} of a tile, it = 0..3: the lanes of a store are 16 bytes apart
  uint32_t cur[4] = {0, 0, 0, 0};
  if (t0 < t1) {
#pragma unroll
    for (uint32_t it = 0; it < 4; ++it) cur[it] = reinterpret_cast<const uint32_t*>(ix + (uint64_t)t0 * TE)[it * 256 + threadIdx.x];
  }
  for (uint32_t t = t0; t < t1; ++t) {
    uint32_t nxt[4];
#pragma unroll
    for (uint32_t it = 0; it < 4; ++it) nxt[it] = cur[it];
    if (t + 1 < t1) {
#pragma unroll
      for (uint32_t it = 0; it < 4; ++it) nxt[it] = reinterpret_cast<const uint32_t*>(ix + (uint64_t)(t + 1) * TE)[it * 256 + threadIdx.x];
    }
    u32x4* d = reinterpret_cast<u32x4*>(dst + (uint64_t)t * TE);
#pragma unroll
    for (uint32_t it = 0; it < 4; ++it) {
      uint64_t o[2];
#pragma unroll
      for (uint32_t u = 0; u < 2; ++u) {
        uint32_t i = (cur[it] >> (u * 16)) & 0xFFFFu;
        i = i < A ? i : A - 1;
        const uint32_t k = i + (i >> 8);
        const uint32_t pair = __builtin_amdgcn_alignbit(e32[(k >> 1) + 1], e32[k >> 1], (k & 1u) * 16u);
        const uint32_t lo = pair & 0xFFFFu;
        o[u] = (uint64_t)(coarse[i >> 8] + lo) | ((uint64_t)((pair >> 16) - lo) << 32);
      }
      const u32x4 q = {(uint32_t)o[0], (uint32_t)(o[0] >> 32), (uint32_t)o[1], (uint32_t)(o[1] >> 32)};
      __builtin_nontemporal_store(q, d + it * 256 + threadIdx.x);
    }
#pragma unroll
    for (uint32_t it = 0; it < 4; ++it) cur[it] = nxt[it];
  }
}

__global__ __launch_bounds__(256, 8) void k_mem(const uint64_t* views, const uint16_t* idx, uint64_t* out, uint32_t A, uint32_t rows) {
  __shared__ uint8_t pad[19000];  // eight workgroups per CU, as in the fused launch
  if (A == 0xFFFFFFFFu) pad[threadIdx.x] = 1;
  const uint32_t n_tiles = (rows + TE - 1) / TE;
  const uint32_t chunk = blockIdx.x / n_tiles, t = blockIdx.x % n_tiles;
  const uint64_t* v = views + (uint64_t)chunk * A;
  const uint32_t* ix = reinterpret_cast<const uint32_t*>(idx + (uint64_t)chunk * rows + (uint64_t)t * TE);
  uint32_t cur[4];
#pragma unroll
  for (uint32_t it = 0; it < 4; ++it) cur[it] = ix[it * 256 + threadIdx.x];
  uint64_t o[8];
#pragma unroll
  for (uint32_t u = 0; u < 8; ++u) {
    const uint32_t i = (cur[u >> 1] >> ((u & 1) * 16)) & 0xFFFFu;
    o[u] = v[i < A ? i : A - 1];
  }
  u32x4* d = reinterpret_cast<u32x4*>(out + (uint64_t)chunk * rows + (uint64_t)t * TE);
#pragma unroll
  for (uint32_t it = 0; it < 4; ++it) {
    const u32x4 q = {(uint32_t)o[2 * it], (uint32_t)(o[2 * it] >> 32), (uint32_t)o[2 * it + 1], (uint32_t)(o[2 * it + 1] >> 32)};
    __builtin_nontemporal_store(q, d + it * 256 + threadIdx.x);
  }
}

int main(int argc, char** argv) {
  const uint32_t A = argc > 1 ? atoi(argv[1]) : 65536;
  const uint32_t chunks = argc > 2 ? atoi(argv[2]) : 45;
  const uint32_t rows = 326 * 2048;  // a multiple of the tile
  std::vector<uint64_t> views((size_t)chunks * A);
  std::vector<uint16_t> idx((size_t)chunks * rows);
  uint32_t seed = 7;
  auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return seed >> 8; };
  for (uint32_t c = 0; c < chunks; ++c) {
    uint32_t off = 1000 * c;
    for (uint32_t j = 0; j < A; ++j) {
      const uint32_t len = rnd() % 50;
      views[(size_t)c * A + j] = off | ((uint64_t)len << 32);
      off += len;
    }
  }
  for (auto& x : idx) x = (uint16_t)(rnd() % A);
  uint64_t *d_views, *d_out, *d_ref;
  uint16_t* d_idx;
  CK(hipMalloc(&d_views, views.size() * 8)); CK(hipMalloc(&d_idx, idx.size() * 2));
  CK(hipMalloc(&d_out, (size_t)chunks * rows * 8)); CK(hipMalloc(&d_ref, (size_t)chunks * rows * 8));
  CK(hipMemcpy(d_views, views.data(), views.size() * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_idx, idx.data(), idx.size() * 2, hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const uint32_t n_tiles = rows / TE;
  auto timed = [&](auto&& launch) {
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
      CK(hipEventRecord(e0));
      launch();
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
    }
    return best * 1e3f;
  };
  const double total_rows = (double)chunks * rows;
  const float mem = timed([&] { k_mem<<<chunks * n_tiles, 256>>>(d_views, d_idx, d_ref, A, rows); });
  printf("A = %u, %u chunks of %u rows (%u tiles each): %.0f MB out\n", A, chunks, rows, n_tiles, total_rows * 8 / 1e6);
  printf("  gather from memory                      %8.1f us  %5.2f ns per tile\n", mem, mem * 1e3 / (chunks * n_tiles));
  for (uint32_t T : {12u, 16u, 24u, 32u, 48u}) {
    const uint32_t g = (n_tiles + T - 1) / T;
    const float us = timed([&] {
      if (A <= 16384) k_lds<16384><<<chunks * g, 256>>>(d_views, d_idx, d_out, A, rows, T, g);
      else k_lds<65536><<<chunks * g, 256>>>(d_views, d_idx, d_out, A, rows, T, g);
    });
    printf("  LDS table, %2u tiles per workgroup (%4u) %8.1f us  %5.2f ns per tile\n", T, chunks * g, us, us * 1e3 / (chunks * n_tiles));
  }
  std::vector<uint64_t> a((size_t)chunks * rows), b((size_t)chunks * rows);
  CK(hipMemcpy(a.data(), d_out, a.size() * 8, hipMemcpyDeviceToHost));
  CK(hipMemcpy(b.data(), d_ref, b.size() * 8, hipMemcpyDeviceToHost));
  size_t bad = 0;
  for (size_t i = 0; i < a.size(); ++i) bad += a[i] != b[i];
  printf("  rows that differ between the two: %zu\n", bad);
  return 0;
}

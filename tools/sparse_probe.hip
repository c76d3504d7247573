// What bounds the positions -> bits loop of the SparseBool step: vector memory
// instructions, LDS atomics, or LDS reads at byte addresses?  Tuning aid; build:
//   hipcc --offload-arch=gfx950 -O3 tools/sparse_probe.hip -o tools/bin/sparse_probe
//
// 5,000 workgroups of 256 threads and 20 KB of LDS (eight per CU, as in the
// fused launch) each take a list of N sorted positions packed at 17 bits, in
// one of these ways:
//   0  eight dword loads per thread and round (lane stride 17 bytes), values summed
//   1  the same, every value also sets its bit in LDS (atomic OR)
//   2  atomic ORs only, positions made up from the lane number
//   3  per wave: one 16-byte load per lane -> LDS, eight dword reads at byte addresses, summed
//   4  the same with the atomic ORs
//   5  the 16-byte loads alone
//   6  as 3, reads forced onto dword addresses (wrong values, timing only)
//   8  as 3 with two aligned dword reads and a funnel shift per position; 9: the same with the atomic ORs
//   7  as 1 with plain read-modify-write of LDS instead of atomics (wrong bits, timing only)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr uint32_t W = 17, ROWS = 100000, NW = (ROWS + 63) / 64;

__device__ __forceinline__ uint32_t ldu32(const uint8_t* p) {
  uint32_t v;
  __builtin_memcpy(&v, p, 4);
  return v;
}

template <int V>
__global__ __launch_bounds__(256, 8) void k_probe(const uint8_t* lists, uint32_t list_bytes, uint32_t n, uint32_t* out) {
  __shared__ __align__(16) uint8_t smem[20096];
  uint32_t* bits = reinterpret_cast<uint32_t*>(smem);
  for (uint32_t i = threadIdx.x; i < NW * 2 + 2; i += 256) bits[i] = 0;
  __syncthreads();
  const uint8_t* p = lists + (uint64_t)blockIdx.x * list_bytes;
  const uint32_t msk = (1u << W) - 1;
  uint32_t acc = 0;
  auto set_row = [&](uint32_t rel) {
    const uint32_t q = rel < NW * 64 ? rel : NW * 64;
    if constexpr (V == 7) bits[q >> 5] |= 1u << (q & 31);
    else atomicOr(&bits[q >> 5], 1u << (q & 31));
  };
  if constexpr (V == 0 || V == 1 || V == 7) {
    const uint32_t rounds = n / 2048;
    const uint8_t* q0 = p + threadIdx.x * W;
    uint32_t cur[8], nxt[8];
#pragma unroll
    for (uint32_t u = 0; u < 8; ++u) cur[u] = ldu32(q0 + ((u * W) >> 3));
    for (uint32_t r = 0; r < rounds; ++r) {
      const uint8_t* q = q0 + (uint64_t)(r + 1 < rounds ? r + 1 : r) * 256 * W;
#pragma unroll
      for (uint32_t u = 0; u < 8; ++u) nxt[u] = ldu32(q + ((u * W) >> 3));
#pragma unroll
      for (uint32_t u = 0; u < 8; ++u) {
        const uint32_t v = (cur[u] >> ((u * W) & 7)) & msk;
        if constexpr (V == 0) acc += v;
        else set_row(v);
      }
#pragma unroll
      for (uint32_t u = 0; u < 8; ++u) cur[u] = nxt[u];
    }
  } else if constexpr (V == 2) {
    const uint32_t rounds = n / 2048;
    for (uint32_t r = 0; r < rounds; ++r) {
#pragma unroll
      for (uint32_t u = 0; u < 8; ++u) set_row((r * 2048 + threadIdx.x * 8 + u) * 5 + ((threadIdx.x + u) & 3));
    }
  } else {
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t g = 59, P = g * 8, Bw = g * W, nl = (Bw + 16) >> 4;
    const uint32_t n_wr = n / P;
    uint8_t* st = smem + ((NW + 1) * 8 + 15) / 16 * 16 + wv * 1056;
    const uint8_t* g0 = p + lane * 16;
    auto ld = [&](uint32_t j) {
      u32x4 v = {0u, 0u, 0u, 0u};
      if (lane < nl) __builtin_memcpy(&v, g0 + (uint64_t)(j < n_wr - 1 ? j : n_wr - 1) * Bw, 16);
      return v;
    };
    u32x4 c0 = ld(wv), c1 = ld(wv + 4);
    for (uint32_t j = wv; j < n_wr; j += 4) {
      const u32x4 c2 = ld(j + 8);
      if constexpr (V == 5) {
        acc += c0.x ^ c0.y ^ c0.z ^ c0.w;
      } else {
        if (lane < nl) *reinterpret_cast<u32x4*>(st + lane * 16) = c0;
        if (lane < g) {
          const uint8_t* q = st + lane * W;
#pragma unroll
          for (uint32_t u = 0; u < 8; ++u) {
            uint32_t x;
            uint32_t v;
            if constexpr (V == 8 || V == 9) {
              const uint32_t b = lane * 8 * W + u * W;  // bit offset in the stage
              const uint32_t* d = reinterpret_cast<const uint32_t*>(st + ((b >> 3) & ~3u));
              v = __builtin_amdgcn_alignbit(d[1], d[0], b) & msk;
            } else {
              if constexpr (V == 6) x = *reinterpret_cast<const uint32_t*>(st + ((lane * W + ((u * W) >> 3)) & ~3u));
              else __builtin_memcpy(&x, q + ((u * W) >> 3), 4);
              v = (x >> ((u * W) & 7)) & msk;
            }
            if constexpr (V == 4 || V == 9) set_row(v);
            else acc += v;
          }
        }
      }
      c0 = c1;
      c1 = c2;
    }
  }
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < NW * 2; i += 256) acc += bits[i];
  out[(uint64_t)blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int V>
static void run(const uint8_t* lists, uint32_t list_bytes, uint32_t n, uint32_t* out, uint32_t blocks, const char* what) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9f;
  for (int rep = 0; rep < 5; ++rep) {
    CK(hipEventRecord(e0));
    k_probe<V><<<blocks, 256>>>(lists, list_bytes, n, out);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  uint32_t h[4];
  CK(hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost));
  printf("  %d %-58s %8.1f us  %6.2f positions/clk/CU (2.4 GHz)  [%08x]\n", V, what, best * 1e3, (double)blocks * n / (best * 1e-3) / 256 / 2.4e9, h[0] ^ h[3]);
}

int main(int argc, char** argv) {
  const uint32_t n = argc > 1 ? atoi(argv[1]) : 20000;
  const uint32_t blocks = argc > 2 ? atoi(argv[2]) : 5000;
  const uint32_t list_bytes = ((uint64_t)n * W / 8 + 64 + 255) & ~255u;
  std::vector<uint8_t> one(list_bytes + 8, 0);
  uint32_t pos = 0;
  uint32_t seed = 12345;
  for (uint32_t i = 0; i < n; ++i) {
    seed = seed * 1664525u + 1013904223u;
    pos += 1 + (seed >> 24) % (2 * ROWS / n - 1);  // mean gap rows / n
    if (pos >= ROWS) pos = ROWS - 1;
    const uint64_t bit = (uint64_t)i * W;
    uint64_t cur;
    memcpy(&cur, &one[bit >> 3], 8);
    cur |= (uint64_t)pos << (bit & 7);
    memcpy(&one[bit >> 3], &cur, 8);
  }
  uint8_t* lists;
  uint32_t* out;
  CK(hipMalloc(&lists, (uint64_t)blocks * list_bytes + 4096));
  CK(hipMalloc(&out, (uint64_t)blocks * 256 * 4));
  for (uint32_t b = 0; b < blocks; ++b) CK(hipMemcpy(lists + (uint64_t)b * list_bytes, one.data(), list_bytes, hipMemcpyHostToDevice));
  printf("%u lists of %u positions (%u bytes each)\n", blocks, n, list_bytes);
  run<0>(lists, list_bytes, n, out, blocks, "dword loads (8 per thread and round), summed");
  run<1>(lists, list_bytes, n, out, blocks, "dword loads + LDS atomic OR");
  run<7>(lists, list_bytes, n, out, blocks, "dword loads + plain LDS read-modify-write");
  run<2>(lists, list_bytes, n, out, blocks, "LDS atomic OR only");
  run<5>(lists, list_bytes, n, out, blocks, "16-byte loads only");
  run<3>(lists, list_bytes, n, out, blocks, "16-byte loads -> LDS -> dword reads at byte addresses");
  run<6>(lists, list_bytes, n, out, blocks, "the same, reads on dword addresses");
  run<4>(lists, list_bytes, n, out, blocks, "16-byte loads -> LDS -> reads + atomic OR");
  run<8>(lists, list_bytes, n, out, blocks, "16-byte loads -> LDS -> two aligned dwords + funnel shift");
  run<9>(lists, list_bytes, n, out, blocks, "the same + atomic OR");
  return 0;
}

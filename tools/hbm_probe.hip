// Achievable HBM rates on this GPU for the access shapes the decoder uses
// (tuning aid; build: hipcc --offload-arch=gfx950 -O3 tools/hbm_probe.hip -o /tmp/hbm_probe).
// Every variant moves `bytes` of output; reads are a fraction of that, as in a
// bit-unpack (w/64 of the output for 8-byte values).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// one tile of TILE bytes per workgroup, 256 threads, 16 bytes per lane per round
template <int TILE, bool NT, int READ_NUM>  // reads READ_NUM/8 of the bytes written
__global__ __launch_bounds__(256) void k_tile(const u32x4* __restrict__ src, u32x4* __restrict__ dst, uint64_t n_vec) {
  constexpr int ROUNDS = TILE / (256 * 16);
  const uint64_t base = (uint64_t)blockIdx.x * (TILE / 16);
  u32x4 v[ROUNDS];
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) {
    const uint64_t i = base + r * 256 + threadIdx.x;
    if (READ_NUM > 0) {
      // READ_NUM/8 of the lanes' data: lane reads vector (i * READ_NUM / 8), neighbours share lines
      v[r] = src[(i * READ_NUM) >> 3];
    } else {
      v[r] = (u32x4){(uint32_t)i, 1, 2, 3};
    }
  }
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) {
    const uint64_t i = base + r * 256 + threadIdx.x;
    if (i < n_vec) {
      if (NT) __builtin_nontemporal_store(v[r], dst + i);
      else dst[i] = v[r];
    }
  }
}

// persistent: grid = k * CUs, each workgroup strides over tiles
template <bool NT, int READ_NUM>
__global__ __launch_bounds__(256) void k_persist(const u32x4* __restrict__ src, u32x4* __restrict__ dst, uint64_t n_vec) {
  for (uint64_t base = (uint64_t)blockIdx.x * 1024; base < n_vec; base += (uint64_t)gridDim.x * 1024) {
    u32x4 v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const uint64_t i = base + r * 256 + threadIdx.x;
      v[r] = READ_NUM > 0 ? src[(i * READ_NUM) >> 3] : (u32x4){(uint32_t)i, 1, 2, 3};
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const uint64_t i = base + r * 256 + threadIdx.x;
      if (i < n_vec) {
        if (NT) __builtin_nontemporal_store(v[r], dst + i);
        else dst[i] = v[r];
      }
    }
  }
}

// the decoder's addressing: tile table -> node descriptor -> tile; optional
// byte-misaligned 16-byte source windows (stride `w` bits per 8-byte value)
struct Desc { uint64_t src_off, dst_off, rows, base; uint32_t w, pad; uint64_t fill[13]; };  // 144 bytes
struct Tile { uint32_t node, tile; };
template <bool UNALIGNED>
__global__ __launch_bounds__(256) void k_indirect(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                  const Desc* __restrict__ descs, const Tile* __restrict__ tiles) {
  const Tile t = tiles[blockIdx.x];
  const Desc& d = descs[t.node];
  const uint64_t start = (uint64_t)t.tile * 2048;  // 8-byte values per 16 KiB tile
  const uint32_t w = d.w;
  const uint8_t* tsrc = src + d.src_off + ((start * w) >> 3);
  u32x4* tdst = reinterpret_cast<u32x4*>(dst + d.dst_off + start * 8);
  u32x4 v[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const uint32_t q = (r * 256 + threadIdx.x) * 2;
    const uint32_t lb = q * w;
    if (UNALIGNED) __builtin_memcpy(&v[r], tsrc + (lb >> 3), 16);
    else v[r] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const uint8_t*>((uintptr_t)(tsrc + (lb >> 3)) & ~(uintptr_t)15));
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    v[r].x += (uint32_t)d.base;
    __builtin_nontemporal_store(v[r], tdst + r * 256 + threadIdx.x);
  }
}

// T consecutive tile entries per workgroup: the T descriptor chains are issued
// together, then the tiles are processed one after the other
template <int T>
__global__ __launch_bounds__(256) void k_indirect_multi(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                        const Desc* __restrict__ descs, const Tile* __restrict__ tiles, uint32_t n_tiles) {
  Tile t[T];
  const uint8_t* tsrc[T]; u32x4* tdst[T]; uint32_t w[T]; uint32_t base[T];
#pragma unroll
  for (int k = 0; k < T; ++k) {
    const uint32_t ti = blockIdx.x * T + k;
    t[k] = tiles[ti < n_tiles ? ti : n_tiles - 1];
  }
#pragma unroll
  for (int k = 0; k < T; ++k) {
    const Desc& d = descs[t[k].node];
    const uint64_t start = (uint64_t)t[k].tile * 2048;
    w[k] = d.w; base[k] = (uint32_t)d.base;
    tsrc[k] = src + d.src_off + ((start * w[k]) >> 3);
    tdst[k] = reinterpret_cast<u32x4*>(dst + d.dst_off + start * 8);
  }
#pragma unroll
  for (int k = 0; k < T; ++k) {
    if (blockIdx.x * T + k >= n_tiles) break;
    u32x4 v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const uint32_t lb = (r * 256 + threadIdx.x) * 2 * w[k];
      __builtin_memcpy(&v[r], tsrc[k] + (lb >> 3), 16);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      v[r].x += base[k];
      __builtin_nontemporal_store(v[r], tdst[k] + r * 256 + threadIdx.x);
    }
  }
}

// everything a tile needs in ONE 32-byte entry (a single dependent load)
struct FatTile { uint64_t src_byte; uint64_t dst_byte; uint32_t w, sh, base, rows; };
__global__ __launch_bounds__(256) void k_fat(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, const FatTile* __restrict__ tiles) {
  const FatTile t = tiles[blockIdx.x];
  const uint8_t* tsrc = src + t.src_byte;
  u32x4* tdst = reinterpret_cast<u32x4*>(dst + t.dst_byte);
  u32x4 v[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const uint32_t lb = t.sh + (r * 256 + threadIdx.x) * 2 * t.w;
    __builtin_memcpy(&v[r], tsrc + (lb >> 3), 16);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    v[r].x += t.base;
    __builtin_nontemporal_store(v[r], tdst + r * 256 + threadIdx.x);
  }
}

// LDS-staged windows: the workgroup copies the tile's packed bytes with aligned,
// fully coalesced 16-byte loads (every byte once), then each lane assembles its
// 16-byte window from five dword-aligned LDS reads.
__global__ __launch_bounds__(256) void k_fat_lds(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, const FatTile* __restrict__ tiles) {
  __shared__ u32x4 stage[1028];  // 16 KiB + 64 bytes
  const FatTile t = tiles[blockIdx.x];
  const uint8_t* tsrc = src + t.src_byte;
  const uint32_t mis = (uint32_t)((uintptr_t)tsrc & 15u);
  const u32x4* asrc = reinterpret_cast<const u32x4*>(tsrc - mis);
  const uint32_t tile_bits = t.sh + 2048u * t.w;
  const uint32_t nvec = (mis + ((tile_bits + 7) >> 3) + 4 + 15) >> 4;  // + 4 bytes: the fifth dword of the last window
  u32x4 g[5];
#pragma unroll
  for (int r = 0; r < 5; ++r) {
    const uint32_t i = r * 256 + threadIdx.x;
    g[r] = asrc[i < nvec ? i : nvec - 1];
  }
#pragma unroll
  for (int r = 0; r < 5; ++r) {
    const uint32_t i = r * 256 + threadIdx.x;
    if (i < nvec) stage[i] = g[r];
  }
  __syncthreads();
  const uint32_t* s32 = reinterpret_cast<const uint32_t*>(stage);
  u32x4* tdst = reinterpret_cast<u32x4*>(dst + t.dst_byte);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const uint32_t lb = t.sh + (r * 256 + threadIdx.x) * 2 * t.w + mis * 8;  // bit offset from the aligned base
    const uint32_t d0 = lb >> 5, S = lb & 31u;
    const uint32_t a = s32[d0], b = s32[d0 + 1], c = s32[d0 + 2], d = s32[d0 + 3], e = s32[d0 + 4];
    u32x4 v;
    v.x = __builtin_amdgcn_alignbit(b, a, S) + t.base;
    v.y = __builtin_amdgcn_alignbit(c, b, S);
    v.z = __builtin_amdgcn_alignbit(d, c, S);
    v.w = __builtin_amdgcn_alignbit(e, d, S);
    __builtin_nontemporal_store(v, tdst + r * 256 + threadIdx.x);
  }
}

__global__ __launch_bounds__(256) void k_read(const u32x4* __restrict__ src, uint32_t* sink, uint64_t n_vec) {
  const uint64_t base = (uint64_t)blockIdx.x * 1024;
  u32x4 acc = {0, 0, 0, 0};
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const uint64_t i = base + r * 256 + threadIdx.x;
    if (i < n_vec) acc ^= src[i];
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) *sink = 1;
}

template <typename F>
static float time_ms(F launch, int iters = 20) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 3; ++i) launch();
  CK(hipEventRecord(a));
  for (int i = 0; i < iters; ++i) launch();
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  return ms / iters;
}

int main(int argc, char** argv) {
  const uint64_t bytes = (argc > 1 ? strtoull(argv[1], nullptr, 10) : 2048ull) << 20;
  const uint64_t n_vec = bytes / 16;
  u32x4 *src, *dst; uint32_t* sink;
  CK(hipMalloc(&src, bytes)); CK(hipMalloc(&dst, bytes)); CK(hipMalloc(&sink, 4));
  CK(hipMemset(src, 1, bytes)); CK(hipMemset(dst, 0, bytes));
  auto report = [&](const char* name, float ms, double rd_frac) {
    const double wr = (double)bytes, rd = wr * rd_frac;
    printf("%-44s %8.3f ms  write %7.1f GB/s  read %7.1f GB/s  total %7.1f GB/s\n", name, ms, wr / ms / 1e6, rd / ms / 1e6, (wr + rd) / ms / 1e6);
  };
#define TILE_CASE(TILE, NT, RN) { const uint32_t grid = (uint32_t)((bytes + TILE - 1) / TILE); \
    float ms = time_ms([&] { hipLaunchKernelGGL((k_tile<TILE, NT, RN>), dim3(grid), dim3(256), 0, 0, src, dst, n_vec); }); \
    char nm[96]; snprintf(nm, sizeof nm, "tile %3dK %s read %d/8", TILE / 1024, NT ? "nt   " : "plain", RN); report(nm, ms, RN / 8.0); }
  TILE_CASE(16384, true, 0) TILE_CASE(16384, false, 0)
  TILE_CASE(65536, true, 0) TILE_CASE(65536, false, 0)
  TILE_CASE(16384, true, 1) TILE_CASE(16384, true, 2) TILE_CASE(16384, true, 4) TILE_CASE(16384, true, 8)
  TILE_CASE(16384, false, 2) TILE_CASE(16384, false, 8)
  TILE_CASE(65536, true, 2) TILE_CASE(65536, true, 8)
  for (int k : {1, 2, 4, 8, 16}) {
    const uint32_t grid = 256 * k;
    float ms = time_ms([&] { hipLaunchKernelGGL((k_persist<true, 0>), dim3(grid), dim3(256), 0, 0, src, dst, n_vec); });
    char nm[96]; snprintf(nm, sizeof nm, "persistent x%-2d nt    read 0/8", k); report(nm, ms, 0);
    ms = time_ms([&] { hipLaunchKernelGGL((k_persist<true, 2>), dim3(grid), dim3(256), 0, 0, src, dst, n_vec); });
    snprintf(nm, sizeof nm, "persistent x%-2d nt    read 2/8", k); report(nm, ms, 0.25);
    ms = time_ms([&] { hipLaunchKernelGGL((k_persist<true, 8>), dim3(grid), dim3(256), 0, 0, src, dst, n_vec); });
    snprintf(nm, sizeof nm, "persistent x%-2d nt    read 8/8", k); report(nm, ms, 1.0);
  }
  for (uint32_t w : {1u, 4u, 7u, 9u, 10u, 11u, 12u, 14u, 20u, 40u, 57u}) {
    // columns of 2.5M 8-byte rows (20 MB out each), as the benchmark's chunks
    const uint64_t col_rows = 2500000, col_out = col_rows * 8, col_in = (col_rows * w + 7) / 8 + 64;
    const uint32_t n_cols = (uint32_t)(bytes / col_out);
    std::vector<Desc> hd(n_cols); std::vector<Tile> ht;
    for (uint32_t c = 0; c < n_cols; ++c) {
      hd[c] = Desc{}; hd[c].src_off = c * ((col_in + 15) / 16 * 16) + 3; hd[c].dst_off = c * col_out; hd[c].rows = col_rows; hd[c].w = w; hd[c].base = 5;
      for (uint32_t t = 0; t < (col_rows + 2047) / 2048 - 1; ++t) ht.push_back(Tile{c, t});  // full tiles only
    }
    Desc* dd; Tile* dt;
    CK(hipMalloc(&dd, hd.size() * sizeof(Desc))); CK(hipMalloc(&dt, ht.size() * sizeof(Tile)));
    CK(hipMemcpy(dd, hd.data(), hd.size() * sizeof(Desc), hipMemcpyHostToDevice));
    CK(hipMemcpy(dt, ht.data(), ht.size() * sizeof(Tile), hipMemcpyHostToDevice));
    const double wr = (double)ht.size() * 16384, rd = wr * w / 64.0;
    float ms = time_ms([&] { hipLaunchKernelGGL((k_indirect<true>), dim3((uint32_t)ht.size()), dim3(256), 0, 0, (const uint8_t*)src, (uint8_t*)dst, dd, dt); });
    printf("indirect unaligned w=%2u                      %8.3f ms  write %7.1f GB/s  read %7.1f GB/s  total %7.1f GB/s\n", w, ms, wr / ms / 1e6, rd / ms / 1e6, (wr + rd) / ms / 1e6);
    ms = time_ms([&] { hipLaunchKernelGGL((k_indirect<false>), dim3((uint32_t)ht.size()), dim3(256), 0, 0, (const uint8_t*)src, (uint8_t*)dst, dd, dt); });
    printf("indirect aligned   w=%2u                      %8.3f ms  write %7.1f GB/s  read %7.1f GB/s  total %7.1f GB/s\n", w, ms, wr / ms / 1e6, rd / ms / 1e6, (wr + rd) / ms / 1e6);
#define MULTI(T) { const uint32_t nt = (uint32_t)ht.size(); \
      ms = time_ms([&] { hipLaunchKernelGGL((k_indirect_multi<T>), dim3((nt + T - 1) / T), dim3(256), 0, 0, (const uint8_t*)src, (uint8_t*)dst, dd, dt, nt); }); \
      printf("indirect x%d tiles/wg w=%2u                    %8.3f ms  write %7.1f GB/s  read %7.1f GB/s  total %7.1f GB/s\n", T, w, ms, wr / ms / 1e6, rd / ms / 1e6, (wr + rd) / ms / 1e6); }
    MULTI(2) MULTI(4) MULTI(8)
    {
      std::vector<FatTile> hf;
      for (const Tile& t : ht) {
        const uint64_t start = (uint64_t)t.tile * 2048, bit = start * w;
        hf.push_back(FatTile{hd[t.node].src_off + (bit >> 3), hd[t.node].dst_off + start * 8, w, (uint32_t)(bit & 7), 5, 2048});
      }
      FatTile* df; CK(hipMalloc(&df, hf.size() * sizeof(FatTile)));
      CK(hipMemcpy(df, hf.data(), hf.size() * sizeof(FatTile), hipMemcpyHostToDevice));
      ms = time_ms([&] { hipLaunchKernelGGL(k_fat, dim3((uint32_t)hf.size()), dim3(256), 0, 0, (const uint8_t*)src, (uint8_t*)dst, df); });
      printf("fat tile entries   w=%2u                      %8.3f ms  write %7.1f GB/s  read %7.1f GB/s  total %7.1f GB/s\n", w, ms, wr / ms / 1e6, rd / ms / 1e6, (wr + rd) / ms / 1e6);
      ms = time_ms([&] { hipLaunchKernelGGL(k_fat_lds, dim3((uint32_t)hf.size()), dim3(256), 0, 0, (const uint8_t*)src, (uint8_t*)dst, df); });
      printf("fat tiles, LDS staged w=%2u                   %8.3f ms  write %7.1f GB/s  read %7.1f GB/s  total %7.1f GB/s\n", w, ms, wr / ms / 1e6, rd / ms / 1e6, (wr + rd) / ms / 1e6);
      CK(hipFree(df));
    }
    CK(hipFree(dd)); CK(hipFree(dt));
  }
  {
    const uint32_t grid = (uint32_t)((n_vec + 1023) / 1024);
    float ms = time_ms([&] { hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, src, sink, n_vec); });
    printf("%-44s %8.3f ms  read %7.1f GB/s\n", "read only", ms, (double)bytes / ms / 1e6);
    ms = time_ms([&] { CK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, 0)); });
    printf("%-44s %8.3f ms  copy total %7.1f GB/s\n", "hipMemcpy D2D", ms, 2.0 * bytes / ms / 1e6);
    ms = time_ms([&] { CK(hipMemsetAsync(dst, 0, bytes, 0)); });
    printf("%-44s %8.3f ms  write %7.1f GB/s\n", "hipMemset", ms, (double)bytes / ms / 1e6);
  }
  return 0;
}

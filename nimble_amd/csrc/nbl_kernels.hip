// HIP kernels (gfx950 / CDNA4, wave64) for the batched cascade decode.
//
// One launch processes one cascade step of *every* chunk of the batch: the grid
// is the concatenation of the tiles of all nodes that need that step, and each
// workgroup finds its node through the launch's tile table.  All kernels are
// HBM-bound integer work: loads and stores are 16 bytes per lane and lane
// contiguous, data-dependent positions are resolved with wave ballots /
// popcount prefix scans staged in LDS, and leaf encodings (FixedBitWidth,
// Trivial, Constant) are read through col_get() so that a parent and its leaf
// children run as one pass over memory.
//
// Memory safety: every data-dependent index (dictionary index, run index, rank
// of a selected value, sparse position, varint ordinal) is clamped or range
// checked before it is used as an address; inconsistent payloads raise a bit in
// the status word instead of faulting.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "nbl_desc.h"

namespace nbl {

// ------------------------------------------------------------------ helpers
__device__ __forceinline__ uint64_t ldu64(const uint8_t* p) {
  uint64_t v;
  __builtin_memcpy(&v, p, 8);  // byte-unaligned global load (dwordx2)
  return v;
}
__device__ __forceinline__ uint32_t ldu32(const uint8_t* p) {
  uint32_t v;
  __builtin_memcpy(&v, p, 4);
  return v;
}
__device__ __forceinline__ uint16_t ldu16(const uint8_t* p) {
  uint16_t v;
  __builtin_memcpy(&v, p, 2);
  return v;
}
__device__ __forceinline__ uint64_t wmask(uint32_t w) { return w >= 64 ? ~0ull : ((1ull << w) - 1ull); }

template <int ES>
struct ElemT;
template <> struct ElemT<1> { using type = uint8_t; };
template <> struct ElemT<2> { using type = uint16_t; };
template <> struct ElemT<4> { using type = uint32_t; };
template <> struct ElemT<8> { using type = uint64_t; };

// Bit-packed value i: LSB-first, value i occupies bits [i*w, (i+1)*w) of the
// payload byte stream (FixedBitArray::get, common/FixedBitArray.cpp:41-57).
// An 8-byte load at the containing byte covers w <= 57; wider values whose
// bits run past it take the ninth byte.
__device__ __forceinline__ uint64_t fbw_get(const uint8_t* payload, uint32_t w, uint64_t i) {
  const uint64_t bit = i * (uint64_t)w;
  const uint8_t* p = payload + (bit >> 3);
  const uint32_t sh = (uint32_t)bit & 7u;
  uint64_t v = ldu64(p) >> sh;
  if (w + sh > 64) v |= (uint64_t)p[8] << (64 - sh);
  return v & wmask(w);
}

// Element i of a numeric column (result truncated to the element width).
__device__ __forceinline__ uint64_t col_get(const ColRef& c, const KArgs& a, uint64_t i) {
  uint64_t v;
  switch (c.kind) {
    case COL_FBW:
      v = fbw_get(a.arena + c.off, c.width, i) + c.base;
      break;
    case COL_RAW: {
      const uint8_t* p = a.arena + c.off + i * c.es;
      v = c.es == 8 ? ldu64(p) : c.es == 4 ? ldu32(p) : c.es == 2 ? ldu16(p) : *p;
      return v;
    }
    case COL_CONST:
      return c.base;
    case COL_MAT_WORK:
    case COL_MAT_OUT: {
      const uint8_t* b = (c.kind == COL_MAT_WORK ? a.work : a.out) + c.off;
      return c.es == 8   ? ((const uint64_t*)b)[i]
             : c.es == 4 ? ((const uint32_t*)b)[i]
             : c.es == 2 ? ((const uint16_t*)b)[i]
                         : b[i];
    }
    default:
      return 0;
  }
  return c.es >= 8 ? v : (v & ((1ull << (8 * c.es)) - 1ull));
}

// 64 bits [64*w, 64*w+64) of a bool column; bits at or past c.rows are undefined.
__device__ __forceinline__ uint64_t bits_word(const ColRef& c, const KArgs& a, uint64_t w) {
  switch (c.kind) {
    case COL_BITS_WORK:
      return ((const uint64_t*)(a.work + c.off))[w];
    case COL_BITS_RAW:
      return ldu64(a.arena + c.off + 8 * w);
    case COL_CONST:
      return c.base ? ~0ull : 0ull;
    default:
      return 0;
  }
}
// Same with bits at or past `rows` cleared (w must address a word that starts before rows).
__device__ __forceinline__ uint64_t bits_word_masked(const ColRef& c, const KArgs& a, uint64_t w, bool invert) {
  uint64_t v = bits_word(c, a, w);
  if (invert) v = ~v;
  const uint64_t first = w * 64;
  const uint64_t rows = c.rows;
  if (first + 64 > rows) v &= wmask((uint32_t)(rows - first));
  return v;
}
__device__ __forceinline__ uint32_t col_bit(const ColRef& c, const KArgs& a, uint64_t i) {
  return (uint32_t)(bits_word(c, a, i >> 6) >> (i & 63)) & 1u;
}

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }
__device__ __forceinline__ uint32_t wave_id() { return threadIdx.x >> 6; }

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint32_t t = __shfl_up(v, d, 64);
    if ((int)lane_id() >= d) v += t;
  }
  return v;
}
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
  return v;
}
// Exclusive scan over the 256 threads of the block; `tmp` holds >= 4 words.
// Returns the exclusive prefix of the caller, block total in `total`.
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t* tmp, uint32_t& total) {
  const uint32_t incl = wave_incl_scan(v);
  __syncthreads();  // protect tmp from a previous use
  if (lane_id() == 63) tmp[wave_id()] = incl;
  __syncthreads();
  const uint32_t w0 = tmp[0], w1 = tmp[1], w2 = tmp[2], w3 = tmp[3];
  const uint32_t wid = wave_id();
  const uint32_t base = (wid > 0 ? w0 : 0) + (wid > 1 ? w1 : 0) + (wid > 2 ? w2 : 0);
  total = w0 + w1 + w2 + w3;
  return base + incl - v;
}
__device__ __forceinline__ uint32_t block_sum(uint32_t v, uint32_t* tmp) {
  v = wave_sum(v);
  __syncthreads();
  if (lane_id() == 0) tmp[wave_id()] = v;
  __syncthreads();
  return tmp[0] + tmp[1] + tmp[2] + tmp[3];
}

// Wave-cooperative 64-ary search over a monotone predicate on positions
// [0, n): returns the first position where pred is true, n when none.
// All 64 lanes of the calling wave must take part.
template <typename Pred>
__device__ __forceinline__ uint32_t wave_first_true(uint32_t n, Pred pred) {
  uint32_t lo = 0, hi = n;
  while (true) {
    const uint32_t span = hi - lo;
    if (span == 0) return lo;
    const uint32_t step = (span + 63u) >> 6;
    const uint64_t idx64 = (uint64_t)lo + (uint64_t)(lane_id() + 1) * step - 1;
    const bool t = idx64 < hi ? pred((uint32_t)idx64) : true;
    const uint64_t m = __ballot(t);
    if (m == 0) return hi;  // the last probe sits on hi-1: nothing in [lo, hi) is true
    const uint32_t f = (uint32_t)__ffsll((unsigned long long)m) - 1u;
    const uint64_t nlo = (uint64_t)lo + (uint64_t)f * step;
    if (step == 1) return nlo < hi ? (uint32_t)nlo : hi;
    const uint64_t nhi = nlo + step - 1;  // this position satisfies pred (or is past hi)
    lo = (uint32_t)nlo;
    hi = nhi < hi ? (uint32_t)nhi : hi;
  }
}

// Spread the low 8 bits of b into 8 bytes of 0/1, bit 0 in the lowest byte.
__device__ __forceinline__ uint64_t spread8(uint32_t b) {
  uint64_t x = b & 0xFFu;
  x = (x | (x << 28)) & 0x0000000F0000000Full;
  x = (x | (x << 14)) & 0x0003000300030003ull;
  x = (x | (x << 7)) & 0x0101010101010101ull;
  return x;
}

// 16-byte register vector (native clang vector: usable inside a union).
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x4 make_u32x4(uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
  u32x4 v;
  v.x = a; v.y = b; v.z = c; v.w = d;
  return v;
}

template <int ES>
union Vec16 {
  u32x4 q;
  typename ElemT<ES>::type v[16 / ES];
};

// Store up to 16/ES elements starting at element e0 of `dst`; full vectors as
// one 16-byte store.
template <int ES>
__device__ __forceinline__ void store_vec(typename ElemT<ES>::type* dst, uint64_t e0, uint64_t rows, const Vec16<ES>& x) {
  constexpr uint32_t V = 16 / ES;
  if (e0 + V <= rows) {
    // decoded columns are written once and not re-read by this pass: a
    // non-temporal store keeps them out of the caches' write-allocate path
    __builtin_nontemporal_store(x.q, reinterpret_cast<u32x4*>(dst + e0));
  } else {
#pragma unroll
    for (uint32_t j = 0; j < V; ++j)
      if (e0 + j < rows) dst[e0 + j] = x.v[j];
  }
}

__device__ __forceinline__ uint8_t* self_ptr(const NodeDesc& nd, const KArgs& a) {
  return (nd.self.kind == COL_MAT_OUT ? a.out : a.work) + nd.self.off;
}

// ------------------------------------------------------------- K_MATERIALIZE
// Root leaf: FixedBitWidth unpack (+baseline), Trivial copy or Constant fill
// straight into the output.  FixedBitWidthEncoding::materialize
// (encodings/FixedBitWidthEncoding.h:187-192), TrivialEncoding::materialize
// (encodings/TrivialEncoding.h:295-300), ConstantEncoding (ConstantEncoding.h:72-77).
//
// Three launch variants keep each inner loop branch free:
//   k_materialize_fbw : bit widths whose 16-byte output vector (V = 16/ES values)
//                       is covered by one 16-byte window of packed bits
//   k_materialize_raw : byte-aligned elements (Trivial, or bit width == 8*ES),
//                       copy + baseline
//   k_materialize     : everything else (Constant fill, odd width/type pairs)

// Widths the window kernel handles (the planner uses the same predicate).
__host__ __device__ constexpr bool fbw_window_ok(uint32_t es, uint32_t w) {
  // V*w + (first-bit offset within its byte) must fit 128 bits; the offset is
  // (e0*w)&7 with e0 a multiple of V, so it is 0 for ES<=2, {0,4} for ES=4 and
  // {0,2,4,6} for ES=8.
  // Width 63 (ES=8) overruns the window by up to 4 bits: the kernel then reads a fifth dword.
  return w >= 1 && (es == 8 ? w <= 63 : es == 4 ? w <= 31 : es == 2 ? w <= 15 : w <= 7);
}

template <int ES>
__global__ __launch_bounds__(256) void k_materialize_fbw(KArgs a) {
  using T = typename ElemT<ES>::type;
  constexpr uint32_t V = 16 / ES;
  constexpr uint32_t TE = kTileBytes / ES;
  const TileRef tr = a.tiles[blockIdx.x];
  const NodeDesc& nd = a.nodes[tr.node];
  const uint64_t rows = nd.rows;
  const uint64_t start = (uint64_t)tr.tile * TE;
  const uint32_t w = nd.c0.width;           // wave uniform
  const T base = (T)nd.c0.base;
  T* dst = reinterpret_cast<T*>(self_ptr(nd, a)) + start;
  const uint64_t tile_bit = start * w;      // scalar 64-bit
  const uint8_t* tile_src = a.arena + nd.c0.off + (tile_bit >> 3);
  const uint32_t tile_sh = (uint32_t)tile_bit & 7u;
  const uint32_t tile_rows = rows - start < TE ? (uint32_t)(rows - start) : TE;
  const uint32_t mask32 = w >= 32 ? 0xFFFFFFFFu : ((1u << w) - 1u);
#pragma unroll
  for (uint32_t it = 0; it < 4; ++it) {
    const uint32_t q = (it * kBlock + threadIdx.x) * V;  // first row of this lane's vector, tile relative
    if (q >= tile_rows) break;
    const uint32_t lb = tile_sh + q * w;                 // < 16384*62 + 8: 20 bits
    u32x4 d;
    __builtin_memcpy(&d, tile_src + (lb >> 3), 16);      // packed bytes + 7 bytes slop keep this in bounds
    const uint32_t S = lb & 7u;
    // normalise: n = window >> S
    uint32_t n0 = __builtin_amdgcn_alignbit(d.y, d.x, S);
    uint32_t n1 = __builtin_amdgcn_alignbit(d.z, d.y, S);
    uint32_t n2 = __builtin_amdgcn_alignbit(d.w, d.z, S);
    uint32_t n3 = d.w >> S;
    if constexpr (ES == 8) {
      if (w == 63) n3 = __builtin_amdgcn_alignbit(ldu32(tile_src + (lb >> 3) + 16), d.w, S);  // bits 128..131
    }
    Vec16<ES> x;
    if constexpr (ES == 8) {
      if (w < 32) {
        const uint32_t v0 = n0 & mask32;
        n0 = __builtin_amdgcn_alignbit(n1, n0, w);
        x.v[0] = (uint64_t)v0 + base;
        x.v[1] = (uint64_t)(n0 & mask32) + base;
      } else {
        const uint32_t wh = w - 32;                       // 0..30
        const uint32_t maskh = (1u << wh) - 1u;           // wh == 0 -> 0
        x.v[0] = (((uint64_t)(n1 & maskh) << 32) | n0) + base;
        // drop w bits: one dword, then wh more
        const uint32_t m0 = __builtin_amdgcn_alignbit(n2, n1, wh);
        const uint32_t m1 = __builtin_amdgcn_alignbit(n3, n2, wh);
        x.v[1] = (((uint64_t)(m1 & maskh) << 32) | m0) + base;
      }
    } else {
#pragma unroll
      for (uint32_t j = 0; j < V; ++j) {
        x.v[j] = (T)((T)(n0 & mask32) + base);
        n0 = __builtin_amdgcn_alignbit(n1, n0, w);
        n1 = __builtin_amdgcn_alignbit(n2, n1, w);
        n2 = __builtin_amdgcn_alignbit(n3, n2, w);
        n3 >>= w;
      }
    }
    store_vec<ES>(dst, q, tile_rows, x);
  }
}

template <int ES>
__global__ __launch_bounds__(256) void k_materialize_raw(KArgs a) {
  using T = typename ElemT<ES>::type;
  constexpr uint32_t V = 16 / ES;
  constexpr uint32_t TE = kTileBytes / ES;
  const TileRef tr = a.tiles[blockIdx.x];
  const NodeDesc& nd = a.nodes[tr.node];
  const uint64_t rows = nd.rows;
  const uint64_t start = (uint64_t)tr.tile * TE;
  const T base = nd.c0.kind == COL_FBW ? (T)nd.c0.base : (T)0;
  const uint8_t* src = a.arena + nd.c0.off + start * ES;
  T* dst = reinterpret_cast<T*>(self_ptr(nd, a)) + start;
  const uint32_t tile_rows = rows - start < TE ? (uint32_t)(rows - start) : TE;
#pragma unroll
  for (uint32_t it = 0; it < 4; ++it) {
    const uint32_t q = (it * kBlock + threadIdx.x) * V;
    if (q >= tile_rows) break;
    Vec16<ES> x;
    if (q + V <= tile_rows) {
      __builtin_memcpy(&x.q, src + (size_t)q * ES, 16);
    } else {
      x.q = make_u32x4(0, 0, 0, 0);
#pragma unroll
      for (uint32_t j = 0; j < V; ++j)
        if (q + j < tile_rows) __builtin_memcpy(&x.v[j], src + (size_t)(q + j) * ES, ES);
    }
#pragma unroll
    for (uint32_t j = 0; j < V; ++j) x.v[j] = (T)(x.v[j] + base);
    store_vec<ES>(dst, q, tile_rows, x);
  }
}

template <int ES>
__global__ __launch_bounds__(256) void k_materialize(KArgs a) {
  using T = typename ElemT<ES>::type;
  constexpr uint32_t V = 16 / ES;
  constexpr uint32_t TE = kTileBytes / ES;
  const TileRef tr = a.tiles[blockIdx.x];
  const NodeDesc& nd = a.nodes[tr.node];
  const ColRef src = nd.c0;
  const uint64_t rows = nd.rows;
  const uint64_t start = (uint64_t)tr.tile * TE;
  T* dst = reinterpret_cast<T*>(self_ptr(nd, a));
#pragma unroll
  for (uint32_t it = 0; it < 4; ++it) {
    const uint64_t e0 = start + (uint64_t)(it * kBlock + threadIdx.x) * V;
    if (e0 >= rows) break;
    Vec16<ES> x;
#pragma unroll
    for (uint32_t j = 0; j < V; ++j) x.v[j] = (T)col_get(src, a, e0 + j < rows ? e0 + j : rows - 1);
    store_vec<ES>(dst, e0, rows, x);
  }
}

// ----------------------------------------------------------- K_BITS_TO_BYTES
// Bool bitmap -> one byte per row, what Encoding::materialize yields for bool
// (TrivialEncoding<bool>::materialize, encodings/TrivialEncoding.cpp:329-368).
__global__ __launch_bounds__(256) void k_bits_to_bytes(KArgs a) {
  const TileRef tr = a.tiles[blockIdx.x];
  const NodeDesc& nd = a.nodes[tr.node];
  const ColRef src = nd.c0;
  const uint64_t rows = nd.rows;
  const uint64_t start = (uint64_t)tr.tile * kTileBytes;
  uint8_t* dst = self_ptr(nd, a);
#pragma unroll
  for (uint32_t it = 0; it < 4; ++it) {
    const uint64_t e0 = start + (uint64_t)(it * kBlock + threadIdx.x) * 16;
    if (e0 >= rows) break;
    const uint64_t word = bits_word(src, a, e0 >> 6);
    const uint32_t b16 = (uint32_t)(word >> (e0 & 63)) & 0xFFFFu;
    Vec16<1> x;
    const uint64_t lo = spread8(b16), hi = spread8(b16 >> 8);
    x.q.x = (uint32_t)lo; x.q.y = (uint32_t)(lo >> 32);
    x.q.z = (uint32_t)hi; x.q.w = (uint32_t)(hi >> 32);
    store_vec<1>(dst, e0, rows, x);
  }
}

// -------------------------------------------------------------- tile reduces
__global__ __launch_bounds__(256) void k_reduce_sum(KArgs a) {
  __shared__ uint32_t tmp[4];
  const TileRef tr = a.tiles[blockIdx.x];
  const NodeDesc& nd = a.nodes[tr.node];
  const ColRef src = nd.c0;
  const uint64_t n = src.rows;
  const uint64_t e0 = (uint64_t)tr.tile * kScanTile + (uint64_t)threadIdx.x * 8;
  uint32_t s = 0;
#pragma unroll
  for (uint32_t j = 0; j < 8; ++j)
    if (e0 + j < n) s += (uint32_t)col_get(src, a, e0 + j);
  const uint32_t tot = block_sum(s, tmp);
  if (threadIdx.x == 0) a.parts[nd.part_base + tr.tile] = tot;
}

// Tile = nd.tile_rows rows of the selection bitmap (a multiple of 64, <= 16384).
__global__ __launch_bounds__(256) void k_reduce_popc(KArgs a) {
  __shared__ uint32_t tmp[4];
  const TileRef tr = a.tiles[blockIdx.x];
  const NodeDesc& nd = a.nodes[tr.node];
  const uint32_t words = (uint32_t)(nd.aux2 >> 6);  // tile rows / 64
  const uint64_t w = (uint64_t)tr.tile * words + threadIdx.x;
  uint32_t s = 0;
  if (threadIdx.x < words && w * 64 < nd.sel.rows)
    s = (uint32_t)__popcll(bits_word_masked(nd.sel, a, w, (nd.flags & NF_INVERT_SEL) != 0));
  const uint32_t tot = block_sum(s, tmp);
  if (threadIdx.x == 0) a.parts[nd.part_base + tr.tile] = tot;
}

// A varint starts at payload byte b when b == 0 or byte b-1 has its high bit
// clear (common/Varint.h:78-81).  Tile = kVarintTile payload bytes.
__device__ __forceinline__ uint32_t varint_start_mask16(const uint8_t* payload, uint64_t len, uint64_t b0) {
  // bit j set <=> byte b0+j starts a varint; b0 is a multiple of 16, b0 < len
  uint8_t prev[17];
  if (b0 == 0) {
    prev[0] = 0;
    __builtin_memcpy(prev + 1, payload, 16);
  } else {
    __builtin_memcpy(prev, payload + b0 - 1, 17);  // arena slop keeps this readable
  }
  uint32_t m = 0;
#pragma unroll
  for (uint32_t j = 0; j < 16; ++j)
    if (b0 + j < len && !(prev[j] & 0x80)) m |= 1u << j;
  return m;
}
__global__ __launch_bounds__(256) void k_reduce_varint(KArgs a) {
  __shared__ uint32_t tmp[4];
  const TileRef tr = a.tiles[blockIdx.x];
  const NodeDesc& nd = a.nodes[tr.node];
  const uint8_t* payload = a.arena + nd.c0.off;
  const uint64_t len = nd.aux2;
  const uint64_t b0 = (uint64_t)tr.tile * kVarintTile + (uint64_t)threadIdx.x * 16;
  uint32_t s = 0;
  if (b0 < len) s = (uint32_t)__popc(varint_start_mask16(payload, len, b0));
  const uint32_t tot = block_sum(s, tmp);
  if (threadIdx.x == 0) a.parts[nd.part_base + tr.tile] = tot;
}

// -------------------------------------------------------------- K_SCAN_TILES
// One workgroup per node: exclusive scan of the node's tile partials in place,
// grand total to totals[total_idx].
__global__ __launch_bounds__(256) void k_scan_tiles(KArgs a) {
  __shared__ uint32_t tmp[4];
  const TileRef tr = a.tiles[blockIdx.x];
  const NodeDesc& nd = a.nodes[tr.node];
  uint32_t* p = a.parts + nd.part_base;
  const uint32_t n = nd.n_parts;
  uint32_t carry = 0;
  for (uint32_t base = 0; base < n; base += kBlock) {
    const uint32_t i = base + threadIdx.x;
    const uint32_t v = i < n ? p[i] : 0;
    uint32_t total;
    const uint32_t ex = block_excl_scan(v, tmp, total);
    if (i < n) p[i] = carry + ex;
    carry += total;
  }
  if (threadIdx.x == 0) a.totals[nd.total_idx] = carry;
}

// ---------------------------------------------------------------- K_RLE_ENDS
// ends[k] = len[0] + ... + len[k]; flags zero-length runs in totals[total_idx+1].
__global__ __launch_bounds__(256) void k_rle_ends(KArgs a) {
  __shared__ uint32_t tmp[4];
  const TileRef tr = a.tiles[blockIdx.x];
  const NodeDesc& nd = a.nodes[tr.node];
  const ColRef src = nd.c0;
  const uint64_t n = src.rows;
  const uint64_t e0 = (uint64_t)tr.tile * kScanTile + (uint64_t)threadIdx.x * 8;
  uint32_t v[8];
  uint32_t s = 0;
  bool zero = false;
#pragma unroll
  for (uint32_t j = 0; j < 8; ++j) {
    const uint32_t len = e0 + j < n ? (uint32_t)col_get(src, a, e0 + j) : 0;
    zero |= (e0 + j < n) && len == 0;
    s += len;
    v[j] = s;
  }
  uint32_t total;
  const uint32_t ex = block_excl_scan(s, tmp, total) + a.parts[nd.part_base + tr.tile];
  uint32_t* ends = reinterpret_cast<uint32_t*>(a.work + nd.aux_off);
  if (e0 + 8 <= n) {
    *reinterpret_cast<u32x4*>(ends + e0) = make_u32x4(v[0] + ex, v[1] + ex, v[2] + ex, v[3] + ex);
    *reinterpret_cast<u32x4*>(ends + e0 + 4) = make_u32x4(v[4] + ex, v[5] + ex, v[6] + ex, v[7] + ex);
  } else {
#pragma unroll
    for (uint32_t j = 0; j < 8; ++j)
      if (e0 + j < n) ends[e0 + j] = v[j] + ex;
  }
  if (__ballot(zero) != 0 && lane_id() == 0) atomicOr(&a.totals[nd.total_idx + 1], 1u);
}

// -------------------------------------------------------------- K_RLE_EXPAND
// RLEEncoding::materialize (encodings/RLEEncoding.h:879-897): row p takes the
// value of run k = #{ends <= p}.  Per tile the run boundaries become a bitmap
// in LDS and k is a popcount prefix, so stores stay lane contiguous.
struct RunRange {
  uint32_t k0, k1;  // runs overlapping the tile: [k0, k1], both < R
};
__device__ __forceinline__ void find_run_range(const uint32_t* ends, uint32_t R, uint64_t p0, uint64_t p1, uint32_t* s_k) {
  // wave 0: first run ending after p0; wave 1: first run ending at/after p1
  if (wave_id() == 0) {
    const uint32_t key = (uint32_t)p0;
    uint32_t k = wave_first_true(R, [&](uint32_t i) { return ends[i] > key; });
    if (lane_id() == 0) s_k[0] = k >= R ? R - 1 : k;
  } else if (wave_id() == 1) {
    const uint32_t key = (uint32_t)p1;
    uint32_t k = wave_first_true(R, [&](uint32_t i) { return ends[i] >= key; });
    if (lane_id() == 0) s_k[1] = k >= R ? R - 1 : k;
  }
}
// run index of absolute row p by binary search in ends[lo..hi]
__device__ __forceinline__ uint32_t run_of_row(const uint32_t* ends, uint32_t lo, uint32_t hi, uint32_t p) {
  while (lo < hi) {
    const uint32_t mid = lo + ((hi - lo) >> 1);
    if (ends[mid] > p) hi = mid; else lo = mid + 1;
  }
  return lo;
}

template <int ES>
__global__ __launch_bounds__(256) void k_rle_expand(KArgs a) {
  using T = typename ElemT<ES>::type;
  constexpr uint32_t V = 16 / ES;
  constexpr uint32_t TE = kTileBytes / ES;
  constexpr uint32_t W = TE / 64;
  __shared__ uint64_t s_bits[W];
  __shared__ uint32_t s_pref[W];
  __shared__ T s_vals[TE];
  __shared__ uint32_t s_k[2];
  __shared__ uint32_t tmp[4];
  const TileRef tr = a.tiles[blockIdx.x];
  const NodeDesc& nd = a.nodes[tr.node];
  const uint64_t rows = nd.rows;
  const uint32_t R = nd.c0.rows;
  const uint64_t p0 = (uint64_t)tr.tile * TE;
  const uint64_t p1 = p0 + TE < rows ? p0 + TE : rows;
  T* dst = reinterpret_cast<T*>(self_ptr(nd, a));
  const uint32_t* ends = reinterpret_cast<const uint32_t*>(a.work + nd.aux_off);
  if (tr.tile == 0 && threadIdx.x == 0 && a.totals[nd.total_idx] != nd.rows) atomicOr(a.status, ST_RLE_SUM_MISMATCH);
  if (R == 0) {  // malformed: rows without runs
    for (uint32_t it = 0; it < 4; ++it) {
      const uint64_t e0 = p0 + (uint64_t)(it * kBlock + threadIdx.x) * V;
      if (e0 >= rows) break;
      Vec16<ES> x; x.q = make_u32x4(0, 0, 0, 0);
      store_vec<ES>(dst, e0, rows, x);
    }
    return;
  }
  find_run_range(ends, R, p0, p1, s_k);
  if (threadIdx.x < W) s_bits[threadIdx.x] = 0;
  __syncthreads();
  const uint32_t k0 = s_k[0], k1 = s_k[1] < s_k[0] ? s_k[0] : s_k[1];
  const uint32_t nruns = k1 - k0 + 1;
  const bool slow = a.totals[nd.total_idx + 1] != 0 || nruns > TE;
  if (!slow) {
    uint32_t* bits32 = reinterpret_cast<uint32_t*>(s_bits);
    for (uint32_t j = k0 + threadIdx.x; j < k1; j += kBlock) {
      const uint64_t e = ends[j];
      if (e > p0 && e < p1) {
        const uint32_t q = (uint32_t)(e - p0);
        atomicOr(&bits32[q >> 5], 1u << (q & 31));
      }
    }
    for (uint32_t j = threadIdx.x; j < nruns; j += kBlock) s_vals[j] = (T)col_get(nd.c1, a, k0 + j);
    __syncthreads();
    uint32_t pc = threadIdx.x < W ? (uint32_t)__popcll(s_bits[threadIdx.x]) : 0;
    uint32_t total;
    const uint32_t ex = block_excl_scan(pc, tmp, total);
    if (threadIdx.x < W) s_pref[threadIdx.x] = ex;
    __syncthreads();
#pragma unroll
    for (uint32_t it = 0; it < 4; ++it) {
      const uint32_t q0 = (it * kBlock + threadIdx.x) * V;
      const uint64_t e0 = p0 + q0;
      if (e0 >= rows) break;
      const uint64_t word = s_bits[q0 >> 6];
      const uint32_t b = q0 & 63;
      uint32_t r = s_pref[q0 >> 6] + (uint32_t)__popcll(word & ((2ull << b) - 1ull));
      Vec16<ES> x;
#pragma unroll
      for (uint32_t j = 0; j < V; ++j) {
        if (j > 0) r += (uint32_t)(word >> (b + j)) & 1u;
        x.v[j] = s_vals[r < nruns ? r : nruns - 1];
      }
      store_vec<ES>(dst, e0, rows, x);
    }
  } else {
    for (uint32_t it = 0; it < 4; ++it) {
      const uint64_t e0 = p0 + (uint64_t)(it * kBlock + threadIdx.x) * V;
      if (e0 >= rows) break;
      Vec16<ES> x;
      for (uint32_t j = 0; j < V; ++j) {
        const uint64_t p = e0 + j < rows ? e0 + j : rows - 1;
        x.v[j] = (T)col_get(nd.c1, a, run_of_row(ends, k0, k1, (uint32_t)p));
      }
      store_vec<ES>(dst, e0, rows, x);
    }
  }
}

// RLE<bool> -> bitmap words (RLEEncoding<bool>, encodings/RLEEncoding.cpp:39-68):
// run k is `first` when k is even.  Tile = 16384 rows = 256 words.
__device__ __forceinline__ uint64_t prefix_xor64(uint64_t x) {
  x ^= x << 1; x ^= x << 2; x ^= x << 4; x ^= x << 8; x ^= x << 16; x ^= x << 32;
  return x;
}
__global__ __launch_bounds__(256) void k_rle_expand_bool(KArgs a) {
  constexpr uint32_t W = kBitTileWords;
  __shared__ uint64_t s_bits[W];
  __shared__ uint32_t s_k[2];
  __shared__ uint32_t tmp[4];
  const TileRef tr = a.tiles[blockIdx.x];
  const NodeDesc& nd = a.nodes[tr.node];
  const uint64_t rows = nd.rows;
  const uint32_t R = nd.c0.rows;
  const uint64_t p0 = (uint64_t)tr.tile * kBitTileRows;
  const uint64_t p1 = p0 + kBitTileRows < rows ? p0 + kBitTileRows : rows;
  uint64_t* dst = reinterpret_cast<uint64_t*>(a.work + nd.self.off);
  const uint32_t* ends = reinterpret_cast<const uint32_t*>(a.work + nd.aux_off);
  const uint64_t firstmask = (nd.flags & NF_FIRST_TRUE) ? ~0ull : 0ull;
  const uint64_t w = (uint64_t)tr.tile * W + threadIdx.x;  // output word of this thread
  if (tr.tile == 0 && threadIdx.x == 0 && a.totals[nd.total_idx] != nd.rows) atomicOr(a.status, ST_RLE_SUM_MISMATCH);
  if (R == 0) {
    if (w * 64 < rows) dst[w] = 0;
    return;
  }
  find_run_range(ends, R, p0, p1, s_k);
  s_bits[threadIdx.x] = 0;
  __syncthreads();
  const uint32_t k0 = s_k[0], k1 = s_k[1] < s_k[0] ? s_k[0] : s_k[1];
  const bool slow = a.totals[nd.total_idx + 1] != 0;
  if (!slow) {
    uint32_t* bits32 = reinterpret_cast<uint32_t*>(s_bits);
    for (uint32_t j = k0 + threadIdx.x; j < k1; j += kBlock) {
      const uint64_t e = ends[j];
      if (e > p0 && e < p1) {
        const uint32_t q = (uint32_t)(e - p0);
        atomicOr(&bits32[q >> 5], 1u << (q & 31));
      }
    }
    __syncthreads();
    const uint64_t word = s_bits[threadIdx.x];
    uint32_t total;
    const uint32_t ex = block_excl_scan((uint32_t)__popcll(word), tmp, total);
    // parity of the run index at each bit: odd runs are !first
    uint64_t odd = prefix_xor64(word);
    if ((k0 + ex) & 1u) odd = ~odd;
    uint64_t v = firstmask ^ odd;
    if (w * 64 < rows) {
      if (w * 64 + 64 > rows) v &= wmask((uint32_t)(rows - w * 64));
      dst[w] = v;
    }
  } else {
    // zero-length runs present: resolve every row by binary search, one word per wave pass
    for (uint32_t i = wave_id(); i < W; i += 4) {
      const uint64_t ww = (uint64_t)tr.tile * W + i;
      if (ww * 64 >= rows) break;
      const uint64_t p = ww * 64 + lane_id();
      bool bit = false;
      if (p < rows) {
        const uint32_t k = run_of_row(ends, k0, k1, (uint32_t)p);
        bit = (((k & 1u) ? ~firstmask : firstmask) & 1ull) != 0;
      }
      const uint64_t v = __ballot(bit);
      if (lane_id() == 0) dst[ww] = v;
    }
  }
}

// ------------------------------------------------------------- K_SPARSE_BITS
// SparseBoolEncoding::materializeBoolsAsBits (encodings/SparseBoolEncoding.cpp:121-142):
// every row is !sparseValue except the listed (ascending) positions.
__global__ __launch_bounds__(256) void k_sparse_bits(KArgs a) {
  constexpr uint32_t W = kBitTileWords;
  __shared__ uint64_t s_bits[W];
  __shared__ uint32_t s_k[2];
  const TileRef tr = a.tiles[blockIdx.x];
  const NodeDesc& nd = a.nodes[tr.node];
  const uint64_t rows = nd.rows;
  const ColRef idx = nd.c0;
  const uint32_t K = idx.rows;
  const uint64_t p0 = (uint64_t)tr.tile * kBitTileRows;
  const uint64_t p1 = p0 + kBitTileRows < rows ? p0 + kBitTileRows : rows;
  uint64_t* dst = reinterpret_cast<uint64_t*>(a.work + nd.self.off);
  if (wave_id() == 0) {
    const uint32_t key = (uint32_t)p0;
    uint32_t k = wave_first_true(K, [&](uint32_t i) { return (uint32_t)col_get(idx, a, i) >= key; });
    if (lane_id() == 0) s_k[0] = k;
  } else if (wave_id() == 1) {
    const uint32_t key = (uint32_t)p1;
    uint32_t k = wave_first_true(K, [&](uint32_t i) { return (uint32_t)col_get(idx, a, i) >= key; });
    if (lane_id() == 0) s_k[1] = k;
  }
  s_bits[threadIdx.x] = 0;
  __syncthreads();
  const uint32_t lo = s_k[0], hi = s_k[1];
  uint32_t* bits32 = reinterpret_cast<uint32_t*>(s_bits);
  for (uint32_t j = lo + threadIdx.x; j < hi; j += kBlock) {
    const uint64_t pos = (uint32_t)col_get(idx, a, j);
    if (pos >= p0 && pos < p1) {
      const uint32_t q = (uint32_t)(pos - p0);
      atomicOr(&bits32[q >> 5], 1u << (q & 31));
    }
  }
  __syncthreads();
  const uint64_t w = (uint64_t)tr.tile * W + threadIdx.x;
  if (w * 64 < rows) {
    uint64_t v = s_bits[threadIdx.x];
    if (!(nd.flags & NF_SPARSE_VALUE)) v = ~v;
    if (w * 64 + 64 > rows) v &= wmask((uint32_t)(rows - w * 64));
    dst[w] = v;
  }
}

// ---------------------------------------------------------- K_SELECT_SCATTER
// NullableEncoding::materialize (encodings/NullableEncoding.h:207-236): row i
// takes values[rank(i)] when its bit is set, physicalType() otherwise; and
// MainlyConstantEncoding::materialize (encodings/MainlyConstantEncoding.h:208-253):
// the common value where isCommon is set, otherValues[rank0(i)] elsewhere.
template <int ES>
__global__ __launch_bounds__(256) void k_select_scatter(KArgs a) {
  using T = typename ElemT<ES>::type;
  constexpr uint32_t V = 16 / ES;
  constexpr uint32_t TE = kTileBytes / ES;
  constexpr uint32_t W = TE / 64;
  __shared__ uint64_t s_bits[W];
  __shared__ uint32_t s_pref[W];
  __shared__ uint32_t tmp[4];
  const TileRef tr = a.tiles[blockIdx.x];
  const NodeDesc& nd = a.nodes[tr.node];
  const uint64_t rows = nd.rows;
  const uint64_t p0 = (uint64_t)tr.tile * TE;
  const bool invert = (nd.flags & NF_INVERT_SEL) != 0;
  T* dst = reinterpret_cast<T*>(self_ptr(nd, a));
  const uint32_t nvals = nd.c1.rows;
  if (tr.tile == 0 && threadIdx.x == 0 && a.totals[nd.total_idx] != nvals) atomicOr(a.status, ST_SELECT_COUNT);
  const uint64_t w = (uint64_t)tr.tile * W + threadIdx.x;
  uint64_t word = 0;
  if (threadIdx.x < W && w * 64 < rows) {
    word = bits_word_masked(nd.sel, a, w, invert);
    if (nd.flags & NF_WRITE_NULLS) reinterpret_cast<uint64_t*>(a.out + nd.aux_off)[w] = invert ? bits_word_masked(nd.sel, a, w, false) : word;
  }
  uint32_t total;
  const uint32_t ex = block_excl_scan((uint32_t)__popcll(word), tmp, total);
  if (threadIdx.x < W) { s_bits[threadIdx.x] = word; s_pref[threadIdx.x] = ex; }
  __syncthreads();
  const uint32_t base = a.parts[nd.part_base + tr.tile];
  const T fill = (T)nd.aux;
#pragma unroll
  for (uint32_t it = 0; it < 4; ++it) {
    const uint32_t q0 = (it * kBlock + threadIdx.x) * V;
    const uint64_t e0 = p0 + q0;
    if (e0 >= rows) break;
    const uint64_t wd = s_bits[q0 >> 6];
    const uint32_t b = q0 & 63;
    uint32_t r = base + s_pref[q0 >> 6] + (uint32_t)__popcll(wd & ((1ull << b) - 1ull));
    Vec16<ES> x;
#pragma unroll
    for (uint32_t j = 0; j < V; ++j) {
      const uint32_t bit = (uint32_t)(wd >> (b + j)) & 1u;
      T v = fill;
      if (bit && nvals) v = (T)col_get(nd.c1, a, r < nvals ? r : nvals - 1);
      x.v[j] = v;
      r += bit;
    }
    store_vec<ES>(dst, e0, rows, x);
  }
}

// Nullable<bool>: scatter a bitmap of values by the selection bitmap.
// Tile = 16384 rows; each wave resolves one output word per pass (lane = bit).
__global__ __launch_bounds__(256) void k_select_bits(KArgs a) {
  constexpr uint32_t W = kBitTileWords;
  __shared__ uint32_t s_pref[W];
  __shared__ uint64_t s_bits[W];
  __shared__ uint32_t tmp[4];
  const TileRef tr = a.tiles[blockIdx.x];
  const NodeDesc& nd = a.nodes[tr.node];
  const uint64_t rows = nd.rows;
  const uint32_t nvals = nd.c1.rows;
  uint64_t* dst = reinterpret_cast<uint64_t*>(a.work + nd.self.off);
  if (tr.tile == 0 && threadIdx.x == 0 && a.totals[nd.total_idx] != nvals) atomicOr(a.status, ST_SELECT_COUNT);
  const uint64_t w = (uint64_t)tr.tile * W + threadIdx.x;
  uint64_t word = 0;
  if (w * 64 < rows) {
    word = bits_word_masked(nd.sel, a, w, false);
    if (nd.flags & NF_WRITE_NULLS) reinterpret_cast<uint64_t*>(a.out + nd.aux_off)[w] = word;
  }
  uint32_t total;
  const uint32_t ex = block_excl_scan((uint32_t)__popcll(word), tmp, total);
  s_bits[threadIdx.x] = word; s_pref[threadIdx.x] = ex;
  __syncthreads();
  const uint32_t base = a.parts[nd.part_base + tr.tile];
  for (uint32_t i = wave_id(); i < W; i += 4) {
    const uint64_t ww = (uint64_t)tr.tile * W + i;
    if (ww * 64 >= rows) break;
    const uint64_t wd = s_bits[i];
    const uint32_t l = lane_id();
    const uint32_t bit = (uint32_t)(wd >> l) & 1u;
    const uint32_t r = base + s_pref[i] + (uint32_t)__popcll(wd & ((1ull << l) - 1ull));
    bool v = false;
    if (bit && nvals) v = col_bit(nd.c1, a, r < nvals ? r : nvals - 1) != 0;
    const uint64_t o = __ballot(v);
    if (l == 0) dst[ww] = o;
  }
}

// ------------------------------------------------------------- K_DICT_GATHER
// DictionaryEncoding::materialize (encodings/DictionaryEncoding.h:227-236).
template <int ES>
__global__ __launch_bounds__(256) void k_dict_gather(KArgs a) {
  using T = typename ElemT<ES>::type;
  constexpr uint32_t V = 16 / ES;
  constexpr uint32_t TE = kTileBytes / ES;
  const TileRef tr = a.tiles[blockIdx.x];
  const NodeDesc& nd = a.nodes[tr.node];
  const uint64_t rows = nd.rows;
  const uint64_t start = (uint64_t)tr.tile * TE;
  const uint32_t A = nd.c1.rows;
  T* dst = reinterpret_cast<T*>(self_ptr(nd, a));
  bool bad = false;
#pragma unroll
  for (uint32_t it = 0; it < 4; ++it) {
    const uint64_t e0 = start + (uint64_t)(it * kBlock + threadIdx.x) * V;
    if (e0 >= rows) break;
    Vec16<ES> x;
#pragma unroll
    for (uint32_t j = 0; j < V; ++j) {
      uint32_t ix = (uint32_t)col_get(nd.c0, a, e0 + j < rows ? e0 + j : rows - 1);
      if (ix >= A) { bad = true; ix = A ? A - 1 : 0; }
      x.v[j] = A ? (T)col_get(nd.c1, a, ix) : (T)0;
    }
    store_vec<ES>(dst, e0, rows, x);
  }
  if (bad) atomicOr(a.status, ST_DICT_INDEX_RANGE);
}

// ----------------------------------------------------------- K_VARINT_DECODE
// VarintEncoding::materialize (encodings/VarintEncoding.h:119-140): LEB128,
// low 7 bits first, + baseline (common/Varint.h:83-143 for the byte limits).
template <int ES>
__global__ __launch_bounds__(256) void k_varint_decode(KArgs a) {
  using T = typename ElemT<ES>::type;
  constexpr uint32_t MAXB = ES == 4 ? 5 : 10;
  __shared__ uint32_t tmp[4];
  const TileRef tr = a.tiles[blockIdx.x];
  const NodeDesc& nd = a.nodes[tr.node];
  const uint8_t* payload = a.arena + nd.c0.off;
  const uint64_t len = nd.aux2;
  const uint64_t rows = nd.rows;
  const T base = (T)nd.c0.base;
  T* dst = reinterpret_cast<T*>(self_ptr(nd, a));
  const uint64_t b0 = (uint64_t)tr.tile * kVarintTile + (uint64_t)threadIdx.x * 16;
  uint32_t m = 0;
  if (b0 < len) m = varint_start_mask16(payload, len, b0);
  uint32_t total;
  uint64_t rank = (uint64_t)a.parts[nd.part_base + tr.tile] + block_excl_scan((uint32_t)__popc(m), tmp, total);
  if (tr.tile == 0 && threadIdx.x == 0 && a.totals[nd.total_idx] < nd.rows) atomicOr(a.status, ST_VARINT_COUNT);
  while (m) {
    const uint32_t j = (uint32_t)__ffs((int)m) - 1u;
    m &= m - 1;
    if (rank < rows) {
      const uint8_t* p = payload + b0 + j;
      uint64_t v = 0;
#pragma unroll
      for (uint32_t k = 0; k < MAXB; ++k) {
        // bytes past the payload end (truncated input) read the arena slop / next bytes: never out of the arena
        const uint8_t byte = (b0 + j + k < len) ? p[k] : 0;
        v |= (uint64_t)(byte & 0x7F) << (7 * k);
        if (!(byte & 0x80)) break;
      }
      dst[rank] = (T)((T)v + base);
    }
    ++rank;
  }
}

// ------------------------------------------------------------ K_STRING_VIEWS
// TrivialEncoding<string_view>::materialize (encodings/TrivialEncoding.cpp:151-164):
// view i = {running sum of lengths, length i}; emitted as uint64 = len << 32 | offset,
// offset relative to the chunk's string region.
__global__ __launch_bounds__(256) void k_string_views(KArgs a) {
  __shared__ uint32_t tmp[4];
  const TileRef tr = a.tiles[blockIdx.x];
  const NodeDesc& nd = a.nodes[tr.node];
  const ColRef src = nd.c0;
  const uint64_t n = nd.rows;
  const uint64_t e0 = (uint64_t)tr.tile * kScanTile + (uint64_t)threadIdx.x * 8;
  uint32_t len[8];
  uint32_t s = 0;
#pragma unroll
  for (uint32_t j = 0; j < 8; ++j) {
    len[j] = e0 + j < n ? (uint32_t)col_get(src, a, e0 + j) : 0;
    s += len[j];
  }
  uint32_t total;
  uint32_t off = block_excl_scan(s, tmp, total) + a.parts[nd.part_base + tr.tile] + (uint32_t)nd.aux;
  uint64_t* dst = reinterpret_cast<uint64_t*>(self_ptr(nd, a));
#pragma unroll
  for (uint32_t j = 0; j < 8; ++j) {
    if (e0 + j < n) dst[e0 + j] = ((uint64_t)len[j] << 32) | off;
    off += len[j];
  }
  if (tr.tile == 0 && threadIdx.x == 0 && a.totals[nd.total_idx] > nd.aux2) atomicOr(a.status, ST_STRING_BYTES);
}

// -------------------------------------------------------------- K_COPY_BYTES
// The blob copy TrivialEncoding<string_view> does into stringBufferFactory
// memory (encodings/TrivialEncoding.cpp:129-135).
__global__ __launch_bounds__(256) void k_copy_bytes(KArgs a) {
  const TileRef tr = a.tiles[blockIdx.x];
  const NodeDesc& nd = a.nodes[tr.node];
  const uint8_t* src = a.arena + nd.aux_off;
  uint8_t* dst = a.out + nd.self.off;
  const uint64_t n = nd.aux;
  const uint64_t start = (uint64_t)tr.tile * kTileBytes;
#pragma unroll
  for (uint32_t it = 0; it < 4; ++it) {
    const uint64_t b = start + (uint64_t)(it * kBlock + threadIdx.x) * 16;
    if (b >= n) break;
    if (b + 16 <= n) {
      u32x4 q;
      __builtin_memcpy(&q, src + b, 16);
      *reinterpret_cast<u32x4*>(dst + b) = q;
    } else {
      for (uint64_t i = b; i < n; ++i) dst[i] = src[i];
    }
  }
}

}  // namespace nbl

// ------------------------------------------------------------------ launcher
extern "C" int nbl_launch_kernel(uint32_t kernel, uint32_t es, const nbl::KArgs* args, void* stream) {
  using namespace nbl;
  const KArgs a = *args;
  if (a.n_tiles == 0) return 0;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const dim3 grid(a.n_tiles), block(kBlock);
#define NBL_BY_ES(K)                                                                  \
  switch (es) {                                                                       \
    case 1: hipLaunchKernelGGL(K<1>, grid, block, 0, s, a); break;                    \
    case 2: hipLaunchKernelGGL(K<2>, grid, block, 0, s, a); break;                    \
    case 4: hipLaunchKernelGGL(K<4>, grid, block, 0, s, a); break;                    \
    case 8: hipLaunchKernelGGL(K<8>, grid, block, 0, s, a); break;                    \
    default: return -1;                                                               \
  }
  switch (kernel) {
    case K_MATERIALIZE: NBL_BY_ES(k_materialize); break;
    case K_MATERIALIZE_FBW: NBL_BY_ES(k_materialize_fbw); break;
    case K_MATERIALIZE_RAW: NBL_BY_ES(k_materialize_raw); break;
    case K_BITS_TO_BYTES: hipLaunchKernelGGL(k_bits_to_bytes, grid, block, 0, s, a); break;
    case K_REDUCE_SUM: hipLaunchKernelGGL(k_reduce_sum, grid, block, 0, s, a); break;
    case K_REDUCE_POPC: hipLaunchKernelGGL(k_reduce_popc, grid, block, 0, s, a); break;
    case K_REDUCE_VARINT: hipLaunchKernelGGL(k_reduce_varint, grid, block, 0, s, a); break;
    case K_SCAN_TILES: hipLaunchKernelGGL(k_scan_tiles, grid, block, 0, s, a); break;
    case K_RLE_ENDS: hipLaunchKernelGGL(k_rle_ends, grid, block, 0, s, a); break;
    case K_RLE_EXPAND: NBL_BY_ES(k_rle_expand); break;
    case K_RLE_EXPAND_BOOL: hipLaunchKernelGGL(k_rle_expand_bool, grid, block, 0, s, a); break;
    case K_SPARSE_BITS: hipLaunchKernelGGL(k_sparse_bits, grid, block, 0, s, a); break;
    case K_SELECT_SCATTER: NBL_BY_ES(k_select_scatter); break;
    case K_SELECT_BITS: hipLaunchKernelGGL(k_select_bits, grid, block, 0, s, a); break;
    case K_DICT_GATHER: NBL_BY_ES(k_dict_gather); break;
    case K_VARINT_DECODE:
      if (es == 4) hipLaunchKernelGGL(k_varint_decode<4>, grid, block, 0, s, a);
      else if (es == 8) hipLaunchKernelGGL(k_varint_decode<8>, grid, block, 0, s, a);
      else return -1;
      break;
    case K_STRING_VIEWS: hipLaunchKernelGGL(k_string_views, grid, block, 0, s, a); break;
    case K_COPY_BYTES: hipLaunchKernelGGL(k_copy_bytes, grid, block, 0, s, a); break;
    default: return -1;
  }
#undef NBL_BY_ES
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

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

// Development aid (-DNBL_PHASE_TIMING): thread 0 of every rank-scatter tile adds
// the cycles between marks to g_phase[]; a mark first waits for everything the
// wave has outstanding, so the slots hold arrival times of a tile's dependent
// loads.  Never part of a product build.
#ifdef NBL_PHASE_TIMING
constexpr uint32_t kPhaseTiles = 1u << 18;     // one record of 8 cycle counts per tile, no atomics (they would serialise the launch)
__device__ uint32_t g_phase[kPhaseTiles * 8];
constexpr uint32_t kPhaseBytes = 16;  // thread 0 keeps the time of its last mark behind the workgroup's LDS
#define NBL_PHASE_BEGIN(smem, size)                                                                        \
  do {                                                                                                     \
    if (threadIdx.x == 0) *reinterpret_cast<unsigned long long*>((smem) + (size)) = __builtin_readcyclecounter(); \
  } while (0)
#define NBL_PHASE(smem, size, i)                                                           \
  do {                                                                                     \
    __builtin_amdgcn_s_waitcnt(0);                                                         \
    if (threadIdx.x == 0) {                                                                \
      unsigned long long* tp_ = reinterpret_cast<unsigned long long*>((smem) + (size));    \
      const unsigned long long t_ = __builtin_readcyclecounter();                          \
      g_phase[(blockIdx.x & (kPhaseTiles - 1)) * 8 + (i)] = (uint32_t)(t_ - *tp_);         \
      *tp_ = t_;                                                                           \
    }                                                                                      \
  } while (0)
#else
constexpr uint32_t kPhaseBytes = 0;
#define NBL_PHASE_BEGIN(smem, size) do {} while (0)
#define NBL_PHASE(smem, size, i) do {} while (0)
#endif

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

// Scalar load of a dword that no wave of the running launch writes (descriptor
// tables, tile tables, scan results of an earlier launch), at a wave-uniform
// address.  Read through the constant address space it is always a scalar load
// (s_load, merged into wide ones), whatever stores precede it in the kernel;
// as an ordinary global load the compiler turns it into a vector load as soon
// as any store may precede it, and a vector load on a CU that is busy storing
// waits behind that CU's queue.
__device__ __forceinline__ uint32_t sload32(const uint32_t* p) {
  typedef const uint32_t __attribute__((address_space(4))) * const_ptr;
  return *reinterpret_cast<const_ptr>(reinterpret_cast<uintptr_t>(p));
}

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
__device__ __forceinline__ uint32_t col_bit(const ColRef& c, const KArgs& a, uint64_t i) {
  return (uint32_t)(bits_word(c, a, i >> 6) >> (i & 63)) & 1u;
}
// How a bool column is read, fixed once per tile (the column descriptor is wave
// uniform): word() is then one load with no branch around it, so several words
// are requested back to back.  Through bits_word() every load sits inside a
// switch over the column kind and the compiler waits for each before it issues
// the next -- one memory round trip per word on the tile's critical path.
struct BitsReader {
  const uint8_t* p;  // bitmap bytes, any alignment; a readable dummy for a constant column
  uint64_t konst;    // constant column: every word
  uint64_t flip;     // all ones when the column is read inverted
  uint64_t rows;
  bool is_const;
  __device__ __forceinline__ BitsReader(const ColRef& c, const KArgs& a, bool invert) {
    is_const = !(c.kind == COL_BITS_WORK || c.kind == COL_BITS_RAW);
    p = is_const ? a.work : (c.kind == COL_BITS_WORK ? a.work : a.arena) + c.off;
    konst = (c.kind == COL_CONST && c.base) ? ~0ull : 0ull;
    flip = invert ? ~0ull : 0ull;
    rows = c.rows;
  }
  // bits [64 w, 64 w + 64), bits at or past `rows` undefined
  __device__ __forceinline__ uint64_t raw(uint64_t w) const {
    const uint64_t v = ldu64(p + (is_const ? 0ull : 8ull * w));
    return (is_const ? konst : v) ^ flip;
  }
  // the same with bits at or past `rows` cleared (w must address a word that starts before rows)
  __device__ __forceinline__ uint64_t word(uint64_t w) const {
    uint64_t v = raw(w);
    const uint64_t first = w * 64;
    if (first + 64 > rows) v &= wmask((uint32_t)(rows - first));
    return v;
  }
};

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }
__device__ __forceinline__ uint32_t wave_id() { return threadIdx.x >> 6; }
// What a workgroup of ONE wave puts where its other bodies have a barrier: the
// LDS operations of a wave complete in order, so lanes see each other's LDS
// writes without one; this only keeps the compiler from moving LDS accesses
// across the point.  (A workgroup barrier would also wait until the wave's
// global stores have been acknowledged.)
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Inclusive scan over the 64 lanes of a wave with data-parallel-primitive moves:
// four shifts inside the 16-lane rows, then the last lane of row 0 / 2 is added
// to row 1 / 3 and lane 31 to the upper half.  Six dependent vector adds; the
// same scan through cross-lane shuffles (LDS permutes) is six LDS round trips,
// an order of magnitude more latency on what is the serial part of every tile
// that has data-dependent positions.  Lanes without a source add 0.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_add(uint32_t v) {
  return v + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xF, false);
}
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
  v = dpp_add<0x111, 0xF>(v);  // row_shr:1
  v = dpp_add<0x112, 0xF>(v);  // row_shr:2
  v = dpp_add<0x114, 0xF>(v);  // row_shr:4
  v = dpp_add<0x118, 0xF>(v);  // row_shr:8
  v = dpp_add<0x142, 0xA>(v);  // row_bcast:15 into rows 1 and 3
  v = dpp_add<0x143, 0xC>(v);  // row_bcast:31 into rows 2 and 3
  return v;
}
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_scan(v), 63);
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
// Aligned 16-byte load of data that is read exactly once (encoded payloads):
// non-temporal, which leaves the caches to the descriptor tables and gathered
// columns and raised the sustained rate of the unpack launch by ~13%.  Only
// for ALIGNED vectors: on misaligned addresses the same hint costs throughput.
__device__ __forceinline__ u32x4 ld16_stream(const u32x4* p) { return __builtin_nontemporal_load(p); }
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
// x.v[j] += base for every element (modular in the element width); the
// narrow types add all the elements of a dword at once.
template <int ES>
__device__ __forceinline__ void vec_add_base(Vec16<ES>& x, typename ElemT<ES>::type base) {
  if constexpr (ES >= 4) {
#pragma unroll
    for (uint32_t j = 0; j < 16 / ES; ++j) x.v[j] += base;
  } else {
    constexpr uint32_t H = ES == 2 ? 0x80008000u : 0x80808080u;  // top bit of each element
    const uint32_t b = ES == 2 ? (uint32_t)base * 0x00010001u : (uint32_t)base * 0x01010101u;
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) {
      const uint32_t v = x.q[k];
      x.q[k] = ((v & ~H) + (b & ~H)) ^ ((v ^ b) & H);
    }
  }
}
// x.v[j] = val for a compile-time j on a zero-initialised vector; narrow
// elements are shifted into their dword so the vector stays in registers.
template <int ES>
__device__ __forceinline__ void vec_set(Vec16<ES>& x, uint32_t j, typename ElemT<ES>::type val) {
  if constexpr (ES >= 4) {
    x.v[j] = val;
  } else {
    constexpr uint32_t per = 4 / ES;
    x.q[j / per] |= (uint32_t)val << (8 * ES * (j % per));
  }
}

// Store up to 16/ES elements starting at element e0 of `dst`; full vectors as
// one 16-byte store.  Tile-relative callers pass 32-bit indices and get 32-bit
// compares (e0 + V cannot wrap: a tile has at most kTileBytes elements).
template <int ES, typename I>
__device__ __forceinline__ void store_vec(typename ElemT<ES>::type* dst, I e0, I rows, const Vec16<ES>& x) {
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

// Byte-misaligned 16-byte global loads are served at about half the rate of
// aligned ones once neighbouring lanes are more than two bytes apart, so wide
// rows are not read with one misaligned window per lane.  Instead the
// workgroup copies the tile's packed bytes into LDS with aligned, fully
// coalesced 16-byte loads (every byte fetched once) and each lane assembles
// its window from five dword-aligned LDS reads.
//
// stage_tile: copies [p - mis, p + nbytes + 4) to `stage` (mis = p & 15) and
// returns mis; the caller synchronises.  ROUNDS * 4096 bytes must cover it.
// stage_load / stage_store: the same copy in two halves, so that independent
// work (a scan with its barriers) runs while the loads are in flight.  These
// loads are ordinary (cached) ones: the rank scatter that uses them waits on
// their latency, neighbouring tiles of a sparse selection share cache lines,
// and measured against the streaming hint they were 10-40% faster there.
template <int ROUNDS>
struct StageRegs {
  u32x4 g[ROUNDS];
  uint32_t nvec;
};
template <int ROUNDS, uint32_t BS = kBlock>  // BS: threads of the workgroup
__device__ __forceinline__ uint32_t stage_load(const uint8_t* p, uint32_t nbytes, StageRegs<ROUNDS>& regs) {
  const uint32_t mis = (uint32_t)(reinterpret_cast<uintptr_t>(p) & 15u);
  const u32x4* asrc = reinterpret_cast<const u32x4*>(p - mis);
  const uint32_t nvec = (mis + nbytes + 4u + 15u) >> 4;
  regs.nvec = nvec;
  if (nvec <= BS) {
    regs.g[0] = asrc[threadIdx.x < nvec ? threadIdx.x : nvec - 1];
    return mis;
  }
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) {
    const uint32_t i = r * BS + threadIdx.x;
    regs.g[r] = asrc[i < nvec ? i : nvec - 1];
  }
  return mis;
}
template <int ROUNDS, uint32_t BS = kBlock>
__device__ __forceinline__ void stage_store(const StageRegs<ROUNDS>& regs, u32x4* stage) {
  if (regs.nvec <= BS) {
    if (threadIdx.x < regs.nvec) stage[threadIdx.x] = regs.g[0];
    return;
  }
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) {
    const uint32_t i = r * BS + threadIdx.x;
    if (i < regs.nvec) stage[i] = regs.g[r];
  }
}
template <int ROUNDS>
__device__ __forceinline__ uint32_t stage_tile(const uint8_t* p, uint32_t nbytes, u32x4* stage) {
  const uint32_t mis = (uint32_t)(reinterpret_cast<uintptr_t>(p) & 15u);
  const u32x4* asrc = reinterpret_cast<const u32x4*>(p - mis);
  const uint32_t nvec = (mis + nbytes + 4u + 15u) >> 4;  // + 4: the fifth dword of the last window
  if (nvec <= kBlock) {  // a short range (sparse selections, few runs): one load per thread instead of ROUNDS
    if (threadIdx.x < nvec) stage[threadIdx.x] = ld16_stream(asrc + threadIdx.x);
    return mis;
  }
  u32x4 g[ROUNDS];
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) {
    const uint32_t i = r * kBlock + threadIdx.x;
    g[r] = ld16_stream(asrc + (i < nvec ? i : nvec - 1));
  }
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) {
    const uint32_t i = r * kBlock + threadIdx.x;
    if (i < nvec) stage[i] = g[r];
  }
  return mis;
}
// 128 bits starting `lb` bits into the staged bytes
__device__ __forceinline__ void lds_window(const uint32_t* s32, uint32_t lb, uint32_t& n0, uint32_t& n1, uint32_t& n2, uint32_t& n3) {
  const uint32_t d0 = lb >> 5, S = lb & 31u;
  const uint32_t a = s32[d0], b = s32[d0 + 1], c = s32[d0 + 2], d = s32[d0 + 3], e = s32[d0 + 4];
  n0 = __builtin_amdgcn_alignbit(b, a, S);
  n1 = __builtin_amdgcn_alignbit(c, b, S);
  n2 = __builtin_amdgcn_alignbit(d, c, S);
  n3 = __builtin_amdgcn_alignbit(e, d, S);
}

// ------------------------------------------------------------ column readers
// A Reader fixes, at compile time, HOW a numeric column is read, so that the
// loops using it are straight-line code: all loads of a tile are issued back
// to back and their latencies overlap.  with_reader() picks the reader once
// per tile from the (wave uniform) column descriptor.
template <int CES>
using elem_t = typename ElemT<CES>::type;

template <int CES>
__device__ __forceinline__ elem_t<CES> ld_elem(const uint8_t* p) {
  elem_t<CES> v;
  __builtin_memcpy(&v, p, CES);  // byte-unaligned global load
  return v;
}

enum ReaderKind : int {
  RK_PTR,    // CES-byte elements at any alignment (arena / work / out), + base
  RK_FBW32,  // bit width <= 25: one unaligned dword covers a value
  RK_FBW64,  // bit width <= 57: one unaligned qword covers a value
  RK_FBW72,  // bit width 58..63 (8-byte elements only): qword + ninth byte
  RK_CONST,  // every element == base
};

template <int RK, int CES>
struct Reader {
  using T = elem_t<CES>;
  static constexpr int kKind = RK;
  const uint8_t* p;
  T base;
  uint32_t w;
  uint64_t mask;
  __device__ __forceinline__ Reader(const ColRef& c, const KArgs& a) {
    p = (c.kind == COL_MAT_WORK ? a.work : c.kind == COL_MAT_OUT ? (const uint8_t*)a.out : a.arena) + c.off;
    base = (c.kind == COL_FBW || c.kind == COL_CONST) ? (T)c.base : (T)0;
    w = c.width;
    mask = wmask(w);
  }
  // value whose first bit is `sh` (0..7) bits into the byte at q
  __device__ __forceinline__ T unpack(const uint8_t* q, uint32_t sh) const {
    if constexpr (RK == RK_FBW32) {
      return (T)((T)((ldu32(q) >> sh) & (uint32_t)mask) + base);
    } else if constexpr (RK == RK_FBW64) {
      return (T)((T)((ldu64(q) >> sh) & mask) + base);
    } else {
      uint64_t v = ldu64(q) >> sh;
      v |= ((uint64_t)q[8] << 1) << (63u - sh);  // sh == 0 contributes nothing
      return (T)((v & mask) + base);
    }
  }
  // element i, any i < rows
  __device__ __forceinline__ T get(uint64_t i) const {
    if constexpr (RK == RK_PTR) {
      return (T)(ld_elem<CES>(p + i * CES) + base);
    } else if constexpr (RK == RK_CONST) {
      return base;
    } else {
      const uint64_t bit = i * w;
      return unpack(p + (bit >> 3), (uint32_t)bit & 7u);
    }
  }
  // A view positioned at element `first` (wave uniform): at(rel) = element
  // first + rel with 32-bit per-lane arithmetic on top of a scalar base.
  // rel must stay below 2^24 (rel * 64 bits fits 32 bits).
  struct View {
    const Reader& r;
    const uint8_t* q0;
    uint32_t sh0;
    __device__ __forceinline__ T at(uint32_t rel) const {
      if constexpr (RK == RK_PTR) {
        return (T)(ld_elem<CES>(q0 + rel * (uint32_t)CES) + r.base);
      } else if constexpr (RK == RK_CONST) {
        return r.base;
      } else {
        const uint32_t lb = sh0 + rel * r.w;
        return r.unpack(q0 + (lb >> 3), lb & 7u);
      }
    }
  };
  // Elements [first, first + n) copied as raw bytes into LDS with aligned
  // 16-byte loads (stage_begin / stage_end); at(rel) then unpacks element
  // first + rel from LDS.  The range must fit kTileBytes (see fits()); the
  // caller synchronises between stage_end() and at().
  struct Staged {
    const Reader& r;
    const uint32_t* s32;
    uint32_t bit0;
    __device__ __forceinline__ T at(uint32_t rel) const {
      if constexpr (RK == RK_CONST) {
        return r.base;
      } else if constexpr (RK == RK_PTR && CES >= 4) {
        // whole-dword elements: every element sits at the same (wave uniform) bit
        // shift inside its dwords, so the address is one shift-add and the two
        // or three dwords come from one LDS instruction
        const uint32_t S = bit0 & 31u;
        const uint32_t* q = s32 + (bit0 >> 5) + rel * (uint32_t)(CES / 4);
        const uint32_t lo = __builtin_amdgcn_alignbit(q[1], q[0], S);
        if constexpr (CES == 8) return (T)((((uint64_t)__builtin_amdgcn_alignbit(q[2], q[1], S) << 32) | lo) + r.base);
        else return (T)((T)lo + r.base);
      } else {
        const uint32_t lb = bit0 + rel * (RK == RK_PTR ? 8u * CES : r.w);
        const uint32_t d0 = lb >> 5, S = lb & 31u;
        const uint32_t lo = __builtin_amdgcn_alignbit(s32[d0 + 1], s32[d0], S);
        if constexpr (RK == RK_PTR) {
          if constexpr (CES == 8) return (T)((((uint64_t)__builtin_amdgcn_alignbit(s32[d0 + 2], s32[d0 + 1], S) << 32) | lo) + r.base);
          else return (T)((T)lo + r.base);
        } else if constexpr (RK == RK_FBW32) {
          return (T)((T)(lo & (uint32_t)r.mask) + r.base);
        } else {
          const uint64_t v = ((uint64_t)__builtin_amdgcn_alignbit(s32[d0 + 2], s32[d0 + 1], S) << 32) | lo;
          return (T)((T)(v & r.mask) + r.base);
        }
      }
    }
  };
  __device__ __forceinline__ uint32_t bits_per_elem() const { return RK == RK_PTR ? 8u * CES : RK == RK_CONST ? 0u : w; }
  __device__ __forceinline__ bool fits(uint32_t n, uint32_t bytes = kTileBytes) const { return (uint64_t)n * bits_per_elem() <= 8ull * bytes; }
  // Staging in two halves: stage_begin issues the loads into registers,
  // stage_end writes them to LDS (the caller synchronises after it).
  // ROUNDS * BS * 16 bytes must cover the range and its alignment slack.
  template <int ROUNDS, uint32_t BS = kBlock>
  __device__ __forceinline__ uint32_t stage_begin(uint64_t first, uint32_t n, StageRegs<ROUNDS>& regs) const {
    if constexpr (RK == RK_CONST) {
      regs.nvec = 0;
      return 0u;
    } else {
      const uint64_t bit = first * bits_per_elem();
      const uint32_t sh = (uint32_t)bit & 7u;
      const uint32_t mis = stage_load<ROUNDS, BS>(p + (bit >> 3), (sh + n * bits_per_elem() + 7u) >> 3, regs);
      return mis * 8u + sh;
    }
  }
  template <int ROUNDS, uint32_t BS = kBlock>
  __device__ __forceinline__ Staged stage_end(const StageRegs<ROUNDS>& regs, uint32_t bit0, uint8_t* lds) const {
    if constexpr (RK != RK_CONST) stage_store<ROUNDS, BS>(regs, reinterpret_cast<u32x4*>(lds));
    return Staged{*this, reinterpret_cast<const uint32_t*>(lds), bit0};
  }
  // the view of values that an earlier stage_end() left in LDS
  __device__ __forceinline__ Staged staged_at(uint32_t bit0, const uint8_t* lds) const {
    return Staged{*this, reinterpret_cast<const uint32_t*>(lds), bit0};
  }
  __device__ __forceinline__ View view(uint64_t first) const {
    if constexpr (RK == RK_PTR || RK == RK_CONST) {
      return View{*this, p + first * CES, 0u};
    } else {
      const uint64_t bit = first * w;
      return View{*this, p + (bit >> 3), (uint32_t)bit & 7u};
    }
  }
};

template <int CES, typename F>
__device__ __forceinline__ void with_reader(const ColRef& c, const KArgs& a, F&& f) {
  if (c.kind == COL_FBW && c.width != 8 * CES) {
    if (c.width <= 25) {
      f(Reader<RK_FBW32, CES>(c, a));
    } else if (CES < 8 || c.width <= 57) {
      f(Reader<RK_FBW64, CES>(c, a));
    } else {
      if constexpr (CES == 8) f(Reader<RK_FBW72, CES>(c, a));
    }
  } else if (c.kind == COL_CONST) {
    f(Reader<RK_CONST, CES>(c, a));
  } else {
    f(Reader<RK_PTR, CES>(c, a));  // RAW, MAT_WORK, MAT_OUT, FBW at full width
  }
}

// LDS of one workgroup, shared by every body (the fused launch runs them all):
// [0, kStageBytes) staged bytes or values of a tile (16 KiB + alignment slack),
// then 2 KiB of bitmap words, 1 KiB of prefix counts, 64 bytes of scratch, 256
// bytes of bit-deposit table.
constexpr uint32_t kStageBytes = kTileBytes + 64;
constexpr uint32_t kSmemBytes = kStageBytes + 2048 + 1024 + 64 + 256;
constexpr uint32_t kDepositOff = kStageBytes + 2048 + 1024 + 64;  // 256-byte table of the nested rank scatter
static_assert(kSmemBytes * 8 <= 160 * 1024, "eight workgroups per CU share the LDS");

// =========================================================== tile bodies
// Every body processes ONE tile of one node.  They are instantiated twice: in
// a named __global__ wrapper (one kernel per operation) and in the fused
// per-level kernel at the end of this file.

// ------------------------------------------------------------- K_MATERIALIZE
// Root leaf: FixedBitWidth unpack (+baseline), Trivial copy or Constant fill
// straight into its destination.  FixedBitWidthEncoding::materialize
// (encodings/FixedBitWidthEncoding.h:187-192), TrivialEncoding::materialize
// (encodings/TrivialEncoding.h:295-300), ConstantEncoding (ConstantEncoding.h:72-77).
//
// Three variants keep each inner loop branch free:
//   materialize_fbw : bit widths whose 16-byte output vector (V = 16/ES values)
//                     is covered by one 16-byte window of packed bits
//   materialize_raw : byte-aligned elements (Trivial, or bit width == 8*ES),
//                     copy + baseline
//   materialize     : everything else (Constant fill, odd width/type pairs)

// V values of w bits from the 128-bit window n0..n3 (w < 8*ES), + base
template <int ES>
__device__ __forceinline__ void unpack_window(uint32_t n0, uint32_t n1, uint32_t n2, uint32_t n3, uint32_t w,
                                              typename ElemT<ES>::type base, Vec16<ES>& x) {
  using T = typename ElemT<ES>::type;
  constexpr uint32_t V = 16 / ES;
  const uint32_t mask32 = w >= 32 ? 0xFFFFFFFFu : ((1u << w) - 1u);
  if constexpr (ES == 8) {
    if (w < 32) {
      const uint32_t v0 = n0 & mask32;
      n0 = __builtin_amdgcn_alignbit(n1, n0, w);
      x.v[0] = (uint64_t)v0 + base;
      x.v[1] = (uint64_t)(n0 & mask32) + base;
    } else {
      const uint32_t wh = w - 32;                       // 0..31
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
}

template <int ES>
__device__ __forceinline__ void body_materialize_fbw(const KArgs& a, const NodeDesc& nd, uint32_t tile, uint8_t* smem) {
  using T = typename ElemT<ES>::type;
  constexpr uint32_t V = 16 / ES;
  constexpr uint32_t TE = kTileBytes / ES;
  const uint64_t rows = nd.rows;
  const uint64_t start = (uint64_t)tile * TE;
  const uint32_t w = nd.c0.width;           // wave uniform
  const T base = (T)nd.c0.base;
  T* dst = reinterpret_cast<T*>(self_ptr(nd, a)) + start;
  const uint64_t tile_bit = start * w;      // scalar 64-bit
  const uint8_t* tile_src = a.arena + nd.c0.off + (tile_bit >> 3);
  const uint32_t tile_sh = (uint32_t)tile_bit & 7u;
  const uint32_t tile_rows = rows - start < TE ? (uint32_t)(rows - start) : TE;
  if (V * w <= 16) {
    // Narrow rows: neighbouring lanes' windows start at most two bytes apart and
    // one (misaligned) 16-byte load per lane runs at full rate.  All four loads
    // are issued before the first store; vectors past the end of the tile
    // re-read the tile's last vector (in bounds) and are not stored.
    const uint32_t last_q = ((tile_rows - 1) / V) * V;
    u32x4 win[4];
#pragma unroll
    for (uint32_t it = 0; it < 4; ++it) {
      uint32_t q = (it * kBlock + threadIdx.x) * V;  // first row of this lane's vector, tile relative
      q = q < last_q ? q : last_q;
      const uint32_t lb = tile_sh + q * w;
      __builtin_memcpy(&win[it], tile_src + (lb >> 3), 16);  // packed bytes + slop keep this in bounds
    }
#pragma unroll
    for (uint32_t it = 0; it < 4; ++it) {
      const uint32_t q = (it * kBlock + threadIdx.x) * V;
      const uint32_t qc = q < last_q ? q : last_q;
      const uint32_t S = (tile_sh + qc * w) & 7u;
      const u32x4 d = win[it];
      Vec16<ES> x;
      unpack_window<ES>(__builtin_amdgcn_alignbit(d.y, d.x, S), __builtin_amdgcn_alignbit(d.z, d.y, S),
                        __builtin_amdgcn_alignbit(d.w, d.z, S), d.w >> S, w, base, x);
      if (q < tile_rows) store_vec<ES>(dst, q, tile_rows, x);
    }
    return;
  }
  // Wide rows: stage the tile's packed bytes (at most 16144) in LDS.
  u32x4* stage = reinterpret_cast<u32x4*>(smem);
  const uint32_t mis = stage_tile<4>(tile_src, (tile_sh + tile_rows * w + 7u) >> 3, stage);
  __syncthreads();
  const uint32_t* s32 = reinterpret_cast<const uint32_t*>(stage);
  const uint32_t bit0 = mis * 8u + tile_sh;
#pragma unroll
  for (uint32_t it = 0; it < 4; ++it) {
    const uint32_t q = (it * kBlock + threadIdx.x) * V;
    if (q < tile_rows) {
      uint32_t n0, n1, n2, n3;
      lds_window(s32, bit0 + q * w, n0, n1, n2, n3);
      Vec16<ES> x;
      unpack_window<ES>(n0, n1, n2, n3, w, base, x);
      store_vec<ES>(dst, q, tile_rows, x);
    }
  }
}

// Byte-aligned elements (Trivial, or FixedBitWidth at the full width): a copy
// (+ baseline) from an arbitrarily aligned source.
template <int ES>
__device__ __forceinline__ void body_materialize_raw(const KArgs& a, const NodeDesc& nd, uint32_t tile, uint8_t* smem) {
  using T = typename ElemT<ES>::type;
  constexpr uint32_t V = 16 / ES;
  constexpr uint32_t TE = kTileBytes / ES;
  const uint64_t rows = nd.rows;
  const uint64_t start = (uint64_t)tile * TE;
  const T base = nd.c0.kind == COL_FBW ? (T)nd.c0.base : (T)0;
  // the source is a payload in the arena, or an already materialised column
  // (those start 16-byte aligned and take the aligned branch below)
  const uint8_t* src = (nd.c0.kind == COL_MAT_WORK ? a.work : nd.c0.kind == COL_MAT_OUT ? (const uint8_t*)a.out : a.arena) +
                       nd.c0.off + start * ES;
  T* dst = reinterpret_cast<T*>(self_ptr(nd, a)) + start;
  const uint32_t tile_rows = rows - start < TE ? (uint32_t)(rows - start) : TE;
  Vec16<ES> x[4];
  if ((reinterpret_cast<uintptr_t>(src) & 15u) == 0) {
    // aligned source: four 16-byte loads in flight before the first store; lanes
    // without a full vector re-read the tile's first 16 bytes (in bounds: arena slop)
#pragma unroll
    for (uint32_t it = 0; it < 4; ++it) {
      const uint32_t q = (it * kBlock + threadIdx.x) * V;
      x[it].q = ld16_stream(reinterpret_cast<const u32x4*>(q + V <= tile_rows ? src + q * (uint32_t)ES : src));
    }
  } else {
    u32x4* stage = reinterpret_cast<u32x4*>(smem);
    const uint32_t mis = stage_tile<5>(src, tile_rows * (uint32_t)ES, stage);  // up to 16 KiB + 19 bytes
    __syncthreads();
    const uint32_t* s32 = reinterpret_cast<const uint32_t*>(stage);
#pragma unroll
    for (uint32_t it = 0; it < 4; ++it) {
      const uint32_t q = (it * kBlock + threadIdx.x) * V;
      const uint32_t qc = q + V <= tile_rows ? q : 0;
      uint32_t n0, n1, n2, n3;
      lds_window(s32, mis * 8u + qc * (uint32_t)(8 * ES), n0, n1, n2, n3);
      x[it].q = make_u32x4(n0, n1, n2, n3);
    }
  }
#pragma unroll
  for (uint32_t it = 0; it < 4; ++it) {
    const uint32_t q = (it * kBlock + threadIdx.x) * V;
    if (q + V <= tile_rows) {
      vec_add_base<ES>(x[it], base);
      store_vec<ES>(dst, q, tile_rows, x[it]);
    } else if (q < tile_rows) {  // the column's partial last vector
#pragma unroll 1
      for (uint32_t r = q; r < tile_rows; ++r) dst[r] = (T)(ld_elem<ES>(src + r * (uint32_t)ES) + base);
    }
  }
}

template <int ES>
__device__ __forceinline__ void body_materialize(const KArgs& a, const NodeDesc& nd, uint32_t tile, uint8_t*) {
  using T = typename ElemT<ES>::type;
  constexpr uint32_t V = 16 / ES;
  constexpr uint32_t TE = kTileBytes / ES;
  const uint64_t rows = nd.rows;
  const uint64_t start = (uint64_t)tile * TE;
  T* dst = reinterpret_cast<T*>(self_ptr(nd, a));
  const uint32_t tile_rows = rows - start < TE ? (uint32_t)(rows - start) : TE;
  if (nd.c0.kind == COL_CONST) {  // ConstantEncoding: a fill
    Vec16<ES> x;
#pragma unroll
    for (uint32_t j = 0; j < V; ++j) x.v[j] = (T)nd.c0.base;
#pragma unroll
    for (uint32_t it = 0; it < 4; ++it) {
      const uint32_t q0 = (it * kBlock + threadIdx.x) * V;
      if (q0 < tile_rows) store_vec<ES>(dst + start, q0, tile_rows, x);
    }
    return;
  }
  // anything the specialised variants do not take: element by element
#pragma unroll 1
  for (uint32_t q = threadIdx.x; q < tile_rows; q += kBlock) dst[start + q] = (T)col_get(nd.c0, a, start + q);
}

// ----------------------------------------------------------- K_BITS_TO_BYTES
// Bool bitmap -> one byte per row, what Encoding::materialize yields for bool
// (TrivialEncoding<bool>::materialize, encodings/TrivialEncoding.cpp:329-368).
__device__ __forceinline__ void body_bits_to_bytes(const KArgs& a, const NodeDesc& nd, uint32_t tile, uint8_t*) {
  const BitsReader bits(nd.c0, a, false);
  const uint64_t rows = nd.rows;
  const uint64_t start = (uint64_t)tile * kTileBytes;
  uint8_t* dst = self_ptr(nd, a);
  uint64_t word[4];
#pragma unroll
  for (uint32_t it = 0; it < 4; ++it) {
    const uint64_t e0 = start + (uint64_t)(it * kBlock + threadIdx.x) * 16;
    word[it] = bits.raw((e0 < rows ? e0 : rows - 1) >> 6);  // clamped: every lane loads
  }
#pragma unroll
  for (uint32_t it = 0; it < 4; ++it) {
    const uint64_t e0 = start + (uint64_t)(it * kBlock + threadIdx.x) * 16;
    if (e0 >= rows) break;
    const uint32_t b16 = (uint32_t)(word[it] >> (e0 & 63)) & 0xFFFFu;
    Vec16<1> x;
    const uint64_t lo = spread8(b16), hi = spread8(b16 >> 8);
    x.q.x = (uint32_t)lo; x.q.y = (uint32_t)(lo >> 32);
    x.q.z = (uint32_t)hi; x.q.w = (uint32_t)(hi >> 32);
    store_vec<1>(dst, e0, rows, x);
  }
}

// The kBitTileWords bitmap words of one tile, held in LDS, written as one byte
// per row at rows [start, start + kBitTileRows) of a bool column (start is a
// multiple of the tile): what body_bits_to_bytes does from a bitmap in memory,
// for producers that emit a root bool column themselves (NF_BOOL_BYTES).
__device__ __forceinline__ void tile_words_to_bytes(const uint64_t* s_words, uint8_t* dst, uint64_t start, uint64_t rows) {
#pragma unroll
  for (uint32_t it = 0; it < 4; ++it) {
    const uint32_t r0 = (it * kBlock + threadIdx.x) * 16;
    const uint64_t e0 = start + r0;
    if (e0 >= rows) break;
    const uint32_t b16 = (uint32_t)(s_words[r0 >> 6] >> (r0 & 63)) & 0xFFFFu;
    Vec16<1> x;
    const uint64_t lo = spread8(b16), hi = spread8(b16 >> 8);
    x.q.x = (uint32_t)lo; x.q.y = (uint32_t)(lo >> 32);
    x.q.z = (uint32_t)hi; x.q.w = (uint32_t)(hi >> 32);
    store_vec<1>(dst, e0, rows, x);
  }
}

// ------------------------------------------------------------ chained scans
// Data-dependent positions (run ends, string offsets, ranks of selected rows,
// varint ordinals) need, per tile, the sum of everything in front of it.  A
// node's tiles form a chain inside ONE launch: every tile publishes the sum of
// its own elements as soon as it has it, then adds up the published sums of the
// tiles in front of it, 64 at a time, back to the nearest tile that has already
// published its inclusive prefix (a decoupled look-back), and publishes its own
// inclusive prefix.  No tile waits for a tile that waits: aggregates do not
// depend on anything, so the chain costs a few trips to memory whatever its
// length, and no separate reduce / scan launch level exists.
//
// Tile order is the order in which a node's workgroups arrive (a ticket from an
// atomic counter per node), not blockIdx: every tile in front of a waiting one
// is then held by a workgroup that is already running, whatever order the
// hardware dispatches workgroups in.  The tile holding the last ticket resets
// the counter.  A look-back word is {value, number of the decode, flag bit,
// prefix bit}: words left by an earlier decode do not match the number and read
// as "not there yet", so nothing is cleared between decodes.  The wait is
// bounded; giving up raises ST_SCAN_STALL.
constexpr uint32_t kChainSpinLimit = 1u << 20;
constexpr uint32_t kScratchOff = kStageBytes + 2048 + 1024;  // the 64 bytes of scratch every body may use
__device__ __forceinline__ uint32_t* chain_lds(uint8_t* smem) { return reinterpret_cast<uint32_t*>(smem + kScratchOff) + 12; }
__device__ __forceinline__ uint64_t look_pack(uint32_t value, uint32_t gen, uint32_t flag, uint32_t prefix) {
  return ((uint64_t)((gen << 2) | (flag << 1) | prefix) << 32) | value;
}
__device__ __forceinline__ uint64_t look_ld(const uint64_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void look_st(uint64_t* p, uint64_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// The tile of the node this workgroup takes (0 for a node of one tile: no
// counter is touched).  Every thread calls; the result is workgroup uniform.
__device__ __forceinline__ uint32_t chain_ticket(const KArgs& a, const NodeDesc& nd, uint32_t n_tiles, uint8_t* smem) {
  if (n_tiles <= 1) return 0u;
  uint32_t* lds = chain_lds(smem);
  uint32_t* counter = a.totals + nd.total_idx + 2;
  if (threadIdx.x == 0) {
    const uint32_t t = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t + 1 >= n_tiles) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // the last ticket: nobody asks again in this launch
    lds[2] = t;
  }
  __syncthreads();
  const uint32_t t = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds[2]);
  return t < n_tiles ? t : n_tiles - 1;  // a counter that was not at zero (an aborted decode) stays in range
}

// Sum of the aggregates of tiles [0, t) of the node, flag_before = OR of their
// flags; this tile's own (agg, flag) are published.  Every thread of the
// 256-thread workgroup calls; the result is workgroup uniform.
__device__ __forceinline__ uint32_t chain_exclusive(const KArgs& a, const NodeDesc& nd, uint32_t t, uint32_t n_tiles, uint32_t agg,
                                                    uint32_t flag, uint32_t& flag_before, uint8_t* smem) {
  flag_before = 0;
  if (n_tiles <= 1) return 0u;
  uint32_t* lds = chain_lds(smem);
  if (threadIdx.x < 64) {
    uint64_t* L = a.look + nd.look_base;
    const uint32_t lane = threadIdx.x, gen = a.gen;
    uint32_t excl = 0, fb = 0;
    if (t == 0) {
      if (lane == 0) look_st(L, look_pack(agg, gen, flag, 1u));
    } else {
      if (lane == 0) look_st(L + t, look_pack(agg, gen, flag, 0u));
      int32_t hi = (int32_t)t - 1;  // the nearest tile in front whose word has not been added yet
      uint32_t spins = 0;
      while (hi >= 0) {
        const int32_t i = hi - (int32_t)lane;
        uint64_t w = 0;
        if (i >= 0) w = look_ld(L + i);
        const uint32_t h = (uint32_t)(w >> 32);
        const bool there = i < 0 || (h >> 2) == gen;
        const uint64_t missing = __ballot(!there);
        const uint64_t prefixes = __ballot(i >= 0 && there && (h & 1u));
        const uint32_t nr = missing ? (uint32_t)__ffsll((unsigned long long)missing) - 1u : 64u;    // words usable: lanes [0, nr)
        const uint32_t np = prefixes ? (uint32_t)__ffsll((unsigned long long)prefixes) - 1u : 64u;  // nearest inclusive prefix
        const uint32_t take = np < nr ? np + 1u : nr;
        const bool mine = lane < take && i >= 0;
        excl += wave_sum(mine ? (uint32_t)w : 0u);
        fb |= __ballot(mine && (h & 2u)) != 0 ? 1u : 0u;
        if (np < nr) break;
        hi -= (int32_t)take;
        if (nr < 64u) {
          __builtin_amdgcn_s_sleep(2);
          if (++spins > kChainSpinLimit) {
            if (lane == 0) atomicOr(a.status, ST_SCAN_STALL);
            break;
          }
        }
      }
      if (lane == 0) look_st(L + t, look_pack(excl + agg, gen, fb | flag, 1u));
    }
    if (lane == 0) { lds[0] = excl; lds[1] = fb; }
  }
  __syncthreads();
  flag_before = lds[1];
  return lds[0];
}

// Sum of `s` over the aligned group of `sub` consecutive threads (sub = 32, 64,
// 128 or 256, workgroup uniform; tmp holds 4 words; every thread calls).
__device__ __forceinline__ uint32_t group_sum(uint32_t s, uint32_t sub, uint32_t* tmp) {
  // one scan with data-parallel-primitive moves gives both halves of the wave (no LDS permutes)
  const uint32_t incl = wave_incl_scan(s);
  const uint32_t whole = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63), low = (uint32_t)__builtin_amdgcn_readlane((int)incl, 31);
  s = sub >= 64 ? whole : (lane_id() < 32 ? low : whole - low);
  if (sub > 64) {
    __syncthreads();
    if (lane_id() == 0) tmp[wave_id()] = s;
    __syncthreads();
    s = sub == 128 ? tmp[wave_id() & 2u] + tmp[(wave_id() & 2u) + 1] : tmp[0] + tmp[1] + tmp[2] + tmp[3];
  }
  return s;
}

// Rank partials of a tile of WPT * kBlock consecutive bitmap words (thread i
// counted word u * kBlock + i of the tile in cnt[u]): one partial per consumer
// tile of `sub` words (32, 64, 128 or 256), at most 64 of them.  Leaves the
// exclusive prefix of every partial within the tile in s_part[0 .. 64) and
// returns the tile's total.  s_part: 65 words; tmp: 4 words; every thread calls.
template <uint32_t WPT>
__device__ __forceinline__ uint32_t tile_partials_scan(uint32_t sub, const uint32_t (&cnt)[WPT], uint32_t* s_part, uint32_t* tmp) {
  static_assert(WPT * kBlock <= kWholeTileWords, "at most 64 partials of at least 32 words");
  __syncthreads();  // the caller's threads may still be reading what s_part held
#pragma unroll
  for (uint32_t u = 0; u < WPT; ++u) {
    const uint32_t s = group_sum(cnt[u], sub, tmp);
    const uint32_t slot = (u * kBlock + threadIdx.x) / sub;
    if ((threadIdx.x & (sub - 1)) == 0 && slot < 64) s_part[slot] = s;
  }
  __syncthreads();
  if (threadIdx.x < 64) {
    const uint32_t np = WPT * kBlock / sub;
    const uint32_t v = threadIdx.x < np ? s_part[threadIdx.x] : 0u;
    const uint32_t incl = wave_incl_scan(v);
    s_part[threadIdx.x] = incl - v;
    if (threadIdx.x == 63) s_part[64] = incl;
  }
  __syncthreads();
  return s_part[64];
}
// Writes the partials tile_partials_scan() left, `before` added, as the scanned
// rank partials of consumer tiles first_slot .. of a node of n_parts consumer
// tiles; the node's last tile also writes the total behind the last partial and
// into totals[total_idx], and clears the node's flag word.
template <uint32_t WPT>
__device__ __forceinline__ void write_scanned_partials(const KArgs& a, uint32_t part_base, uint32_t n_parts, uint32_t total_idx,
                                                       uint32_t sub, uint32_t first_slot, uint32_t before, bool last_tile,
                                                       const uint32_t* s_part) {
  if (threadIdx.x < WPT * kBlock / sub && first_slot + threadIdx.x < n_parts) a.parts[part_base + first_slot + threadIdx.x] = before + s_part[threadIdx.x];
  if (last_tile && threadIdx.x == 0) {
    const uint32_t total = before + s_part[64];
    a.parts[part_base + n_parts] = total;  // so that tile t's share is parts[t + 1] - parts[t] for every t
    a.totals[total_idx] = total;
    a.totals[total_idx + 1] = 0;
  }
}

// K_REDUCE_POPC: scanned rank partials of a selection bitmap that no step of
// this batch builds (a Trivial<bool> payload, or a bitmap another encoding
// left in work).  One workgroup counts kPopcTileRows rows (kPopcWordsPerThread
// 64-bit words per thread, all loaded before the first is counted), one partial
// per consumer tile of nd.aux2 rows; the tiles of a node are a chained scan.  A
// node of up to kWholeTileRows rows is one tile (NF_WHOLE_NODE).
template <uint32_t WPT>
__device__ __forceinline__ void reduce_popc_words(const KArgs& a, const NodeDesc& nd, uint32_t tile, uint32_t (&cnt)[WPT]) {
  const BitsReader sel(nd.sel, a, (nd.flags & NF_INVERT_SEL) != 0);
  uint64_t wd[WPT];
#pragma unroll
  for (uint32_t u = 0; u < WPT; ++u) {  // every word requested before the first is counted
    const uint64_t w = ((uint64_t)tile * WPT + u) * kBitTileWords + threadIdx.x;
    wd[u] = w * 64 < nd.sel.rows ? sel.word(w) : 0ull;
  }
#pragma unroll
  for (uint32_t u = 0; u < WPT; ++u) cnt[u] = (uint32_t)__popcll(wd[u]);
}
__device__ __forceinline__ void body_reduce_popc(const KArgs& a, const NodeDesc& nd, uint32_t, uint8_t* smem) {
  uint32_t* tmp = reinterpret_cast<uint32_t*>(smem + kScratchOff);
  uint32_t* s_part = reinterpret_cast<uint32_t*>(smem);
  const uint32_t sub = (uint32_t)(nd.aux2 >> 6);
  if (nd.flags & NF_WHOLE_NODE) {
    constexpr uint32_t WPT = kWholeTileWords / kBlock;
    uint32_t cnt[WPT];
    reduce_popc_words<WPT>(a, nd, 0, cnt);
    tile_partials_scan<WPT>(sub, cnt, s_part, tmp);
    write_scanned_partials<WPT>(a, nd.part_base, nd.n_parts, nd.total_idx, sub, 0u, 0u, true, s_part);
    return;
  }
  constexpr uint32_t WPT = kPopcWordsPerThread;
  const uint32_t n_tiles = (uint32_t)(((uint64_t)nd.sel.rows + kPopcTileRows - 1) / kPopcTileRows);
  const uint32_t tile = chain_ticket(a, nd, n_tiles, smem);
  uint32_t cnt[WPT];
  reduce_popc_words<WPT>(a, nd, tile, cnt);
  const uint32_t total = tile_partials_scan<WPT>(sub, cnt, s_part, tmp);
  uint32_t unused;
  const uint32_t before = chain_exclusive(a, nd, tile, n_tiles, total, 0u, unused, smem);
  write_scanned_partials<WPT>(a, nd.part_base, nd.n_parts, nd.total_idx, sub, tile * (WPT * kBlock / sub), before, tile + 1 == n_tiles, s_part);
}

// first_run[] of an RLE node sits behind its run-end array (R entries, 16-byte rounded)
__device__ __forceinline__ uint32_t* rle_first_run(const KArgs& a, const NodeDesc& nd) {
  return reinterpret_cast<uint32_t*>(a.work + nd.aux_off) + (((uint64_t)nd.c0.rows + 3) & ~3ull);
}

// ---------------------------------------------------------------- K_RLE_ENDS
// ends[k] = len[0] + ... + len[k] (a chained scan over the node's tiles of
// kScanTile runs); the node's last tile leaves the sum of all lengths in
// totals[total_idx] and "some run has length zero" in totals[total_idx + 1].
__device__ __forceinline__ void body_rle_ends(const KArgs& a, const NodeDesc& nd, uint32_t, uint8_t* smem) {
  uint32_t* tmp = reinterpret_cast<uint32_t*>(smem + kScratchOff);
  const uint64_t n = nd.c0.rows;
  const uint32_t n_tiles = nd.n_parts;
  const uint32_t tile = chain_ticket(a, nd, n_tiles, smem);
  const uint64_t e0 = (uint64_t)tile * kScanTile + (uint64_t)threadIdx.x * 8;
  uint32_t v[8];
  uint32_t s = 0;
  bool zero = false;
  {
    const uint64_t t0 = (uint64_t)tile * kScanTile;
    const uint32_t tn = n - t0 < kScanTile ? (uint32_t)(n - t0) : kScanTile;
    const uint32_t r0 = threadIdx.x * 8;
    with_reader<4>(nd.c0, a, [&](auto rd) {
      const auto src = rd.view(t0);
#pragma unroll
      for (uint32_t j = 0; j < 8; ++j) v[j] = src.at(r0 + j < tn ? r0 + j : tn - 1);
    });
  }
#pragma unroll
  for (uint32_t j = 0; j < 8; ++j) {
    const uint32_t len = e0 + j < n ? v[j] : 0;
    zero |= (e0 + j < n) && len == 0;
    s += len;
    v[j] = s;
  }
  uint32_t total;
  uint32_t ex = block_excl_scan(s, tmp, total);
  const uint32_t any_zero = __syncthreads_or(zero) ? 1u : 0u;
  uint32_t zero_before;
  const uint32_t before = chain_exclusive(a, nd, tile, n_tiles, total, any_zero, zero_before, smem);
  ex += before;
  if (tile + 1 == n_tiles && threadIdx.x == 0) {
    a.totals[nd.total_idx] = before + total;
    a.totals[nd.total_idx + 1] = zero_before | any_zero;
  }
  uint32_t* ends = reinterpret_cast<uint32_t*>(a.work + nd.aux_off);
  if (e0 + 8 <= n) {
    *reinterpret_cast<u32x4*>(ends + e0) = make_u32x4(v[0] + ex, v[1] + ex, v[2] + ex, v[3] + ex);
    *reinterpret_cast<u32x4*>(ends + e0 + 4) = make_u32x4(v[4] + ex, v[5] + ex, v[6] + ex, v[7] + ex);
  } else {
#pragma unroll
    for (uint32_t j = 0; j < 8; ++j)
      if (e0 + j < n) ends[e0 + j] = v[j] + ex;
  }
  // first_run[t] = the run that holds row t * tile_rows, for every output tile t:
  // run i covers rows [start, end), so it is the first run of every tile whose
  // first row falls inside it.  Spares the expand step a search per tile.
  uint32_t* first_run = rle_first_run(a, nd);
  const uint32_t tile_rows = (uint32_t)nd.aux2;
  const uint64_t rows = nd.rows;
  if (rows == 0) return;  // no output tile, so no first_run entry exists (the planner schedules nothing for such a node)
  uint32_t start = ex;
#pragma unroll
  for (uint32_t j = 0; j < 8; ++j) {
    const uint32_t end = v[j] + ex;
    if (e0 + j < n && end > start) {  // wrapped sums (corrupt lengths) have end <= start and write nothing
      const uint64_t last_row = (uint64_t)end - 1 < rows ? (uint64_t)end - 1 : rows - 1;  // < rows: t_hi stays inside first_run[]
      const uint32_t t_hi = (uint32_t)(last_row / tile_rows);
#pragma unroll 1
      for (uint32_t t = (uint32_t)(((uint64_t)start + tile_rows - 1) / tile_rows); t <= t_hi; ++t) first_run[t] = (uint32_t)(e0 + j);
    }
    start = end;
  }
}

// -------------------------------------------------------------- K_RLE_EXPAND
// RLEEncoding::materialize (encodings/RLEEncoding.h:879-897): row p takes the
// value of run k = #{ends <= p}.  Per tile the run boundaries become a bitmap
// in LDS and k is a popcount prefix, so stores stay lane contiguous.
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
__device__ __forceinline__ void body_rle_expand(const KArgs& a, const NodeDesc& nd, uint32_t tile, uint8_t* smem) {
  using T = typename ElemT<ES>::type;
  constexpr uint32_t V = 16 / ES;
  constexpr uint32_t TE = kTileBytes / ES;
  constexpr uint32_t W = TE / 64;
  T* s_vals = reinterpret_cast<T*>(smem);                                  // TE elements = kTileBytes
  uint64_t* s_bits = reinterpret_cast<uint64_t*>(smem + kStageBytes);        // W <= 256 words
  uint32_t* s_pref = reinterpret_cast<uint32_t*>(smem + kStageBytes + 2048);  // W words
  uint32_t* tmp = reinterpret_cast<uint32_t*>(smem + kStageBytes + 3072);     // 4 words
  uint32_t* s_k = tmp + 4;                                                   // 2 words
  const uint64_t rows = nd.rows;
  const uint32_t R = nd.c0.rows;
  const uint64_t p0 = (uint64_t)tile * TE;
  const uint64_t p1 = p0 + TE < rows ? p0 + TE : rows;
  T* dst = reinterpret_cast<T*>(self_ptr(nd, a));
  const uint32_t* ends = reinterpret_cast<const uint32_t*>(a.work + nd.aux_off);
  if (tile == 0 && threadIdx.x == 0 && sload32(a.totals + nd.total_idx) != nd.rows) atomicOr(a.status, ST_RLE_SUM_MISMATCH);
  if (R == 0) {  // malformed: rows without runs
    for (uint32_t it = 0; it < 4; ++it) {
      const uint64_t e0 = p0 + (uint64_t)(it * kBlock + threadIdx.x) * V;
      if (e0 >= rows) break;
      Vec16<ES> x; x.q = make_u32x4(0, 0, 0, 0);
      store_vec<ES>(dst, e0, rows, x);
    }
    return;
  }
  // totals, first_run and the tile partials were written by an EARLIER launch: scalar loads
  const bool zero_runs = sload32(a.totals + nd.total_idx + 1) != 0;
  uint32_t k0, k1;
  uint32_t e_first = 0;  // run end this thread marks first (loaded before the barrier below)
  if (zero_runs) {
    find_run_range(ends, R, p0, p1, s_k);
    if (threadIdx.x < W) s_bits[threadIdx.x] = 0;
    __syncthreads();
    k0 = s_k[0];
    k1 = s_k[1] < s_k[0] ? s_k[0] : s_k[1];
  } else {
    // runs of the tile: from the run holding its first row to the run holding the
    // next tile's first row (clamped: these come from payload data).  Every
    // thread reads the two entries itself (one address: a scalar load), so the
    // first run ends are in flight before the barrier.
    const uint32_t* first_run = rle_first_run(a, nd);
    const uint32_t f0 = sload32(first_run + tile);
    const uint32_t f1 = p0 + TE < rows ? sload32(first_run + tile + 1) : R - 1;
    const uint32_t c0 = f0 < R ? f0 : R - 1;
    uint32_t c1 = f1 < R ? f1 : R - 1;
    if (c1 < c0) c1 = c0;
    if (c1 - c0 >= TE) c1 = c0 + TE - 1;  // a tile of TE rows meets at most TE non-empty runs
    k0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)c0);
    k1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)c1);
    if (k0 + threadIdx.x < k1) e_first = ends[k0 + threadIdx.x];
    if (threadIdx.x < W) s_bits[threadIdx.x] = 0;
    __syncthreads();
  }
  const uint32_t nruns = k1 - k0 + 1;
  const bool slow = zero_runs || nruns > TE;
  if (!slow) {
    uint32_t* bits32 = reinterpret_cast<uint32_t*>(s_bits);
    for (uint32_t j = k0 + threadIdx.x; j < k1; j += kBlock) {
      const uint64_t e = j < k0 + kBlock ? e_first : ends[j];
      if (e > p0 && e < p1) {
        const uint32_t q = (uint32_t)(e - p0);
        atomicOr(&bits32[q >> 5], 1u << (q & 31));
      }
    }
    with_reader<ES>(nd.c1, a, [&](auto rd) {
      const auto src = rd.view(k0);
#pragma unroll 2
      for (uint32_t j = threadIdx.x; j < nruns; j += kBlock) s_vals[j] = src.at(j);
    });
    __syncthreads();
    if constexpr (W <= 64) {  // one wave holds every boundary word of the tile: scan without further barriers
      if (threadIdx.x < 64) {
        const uint32_t pc = threadIdx.x < W ? (uint32_t)__popcll(s_bits[threadIdx.x]) : 0u;
        const uint32_t incl = wave_incl_scan(pc);
        if (threadIdx.x < W) s_pref[threadIdx.x] = incl - pc;
      }
    } else {
      uint32_t pc = threadIdx.x < W ? (uint32_t)__popcll(s_bits[threadIdx.x]) : 0;
      uint32_t total;
      const uint32_t ex = block_excl_scan(pc, tmp, total);
      if (threadIdx.x < W) s_pref[threadIdx.x] = ex;
    }
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
      x.q = make_u32x4(0, 0, 0, 0);
#pragma unroll
      for (uint32_t j = 0; j < V; ++j) {
        if (j > 0) r += (uint32_t)(word >> (b + j)) & 1u;
        vec_set<ES>(x, j, s_vals[r < nruns ? r : nruns - 1]);
      }
      store_vec<ES>(dst, e0, rows, x);
    }
  } else {
    // zero-length runs, or more runs than the LDS stage holds: every row finds its run by binary search
#pragma unroll 1
    for (uint64_t p = p0 + threadIdx.x; p < p1; p += kBlock) dst[p] = (T)col_get(nd.c1, a, run_of_row(ends, k0, k1, (uint32_t)p));
  }
}

// RLE<bool> -> bitmap words (RLEEncoding<bool>, encodings/RLEEncoding.cpp:39-68):
// run k is `first` when k is even.  Tile = 16384 rows = 256 words.
__device__ __forceinline__ uint64_t prefix_xor64(uint64_t x) {
  x ^= x << 1; x ^= x << 2; x ^= x << 4; x ^= x << 8; x ^= x << 16; x ^= x << 32;
  return x;
}
__device__ __forceinline__ void body_rle_expand_bool(const KArgs& a, const NodeDesc& nd, uint32_t tile, uint8_t* smem) {
  constexpr uint32_t W = kBitTileWords;
  uint64_t* s_bits = reinterpret_cast<uint64_t*>(smem);            // 256 words
  uint32_t* tmp = reinterpret_cast<uint32_t*>(smem + 2048);
  uint32_t* s_k = tmp + 4;
  const uint64_t rows = nd.rows;
  const uint32_t R = nd.c0.rows;
  const uint64_t p0 = (uint64_t)tile * kBitTileRows;
  const uint64_t p1 = p0 + kBitTileRows < rows ? p0 + kBitTileRows : rows;
  // a root column is written as bytes straight away (NF_BOOL_BYTES: self = the
  // chunk's values in out), anything else as bitmap words in work
  const bool bytes = (nd.flags & NF_BOOL_BYTES) != 0;
  uint64_t* dst = reinterpret_cast<uint64_t*>(a.work + nd.self.off);
  uint8_t* dst_bytes = a.out + nd.self.off;
  const uint32_t* ends = reinterpret_cast<const uint32_t*>(a.work + nd.aux_off);
  const uint64_t firstmask = (nd.flags & NF_FIRST_TRUE) ? ~0ull : 0ull;
  const uint64_t w = (uint64_t)tile * W + threadIdx.x;  // output word of this thread
  if (tile == 0 && threadIdx.x == 0 && sload32(a.totals + nd.total_idx) != nd.rows) atomicOr(a.status, ST_RLE_SUM_MISMATCH);
  if (R == 0) {
    if (bytes) {
      s_bits[threadIdx.x] = 0;
      __syncthreads();
      tile_words_to_bytes(s_bits, dst_bytes, p0, rows);
    } else if (w * 64 < rows) {
      dst[w] = 0;
    }
    return;
  }
  const bool slow = sload32(a.totals + nd.total_idx + 1) != 0;
  if (slow) {
    find_run_range(ends, R, p0, p1, s_k);
  } else if (threadIdx.x == 0) {
    const uint32_t* first_run = rle_first_run(a, nd);
    const uint32_t f0 = sload32(first_run + tile);
    const uint32_t f1 = p0 + kBitTileRows < rows ? sload32(first_run + tile + 1) : R - 1;
    const uint32_t c0 = f0 < R ? f0 : R - 1;
    const uint32_t c1 = f1 < R ? f1 : R - 1;
    s_k[0] = c0; s_k[1] = c1 < c0 ? c0 : c1;
  }
  s_bits[threadIdx.x] = 0;
  __syncthreads();
  const uint32_t k0 = s_k[0], k1 = s_k[1] < s_k[0] ? s_k[0] : s_k[1];
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
    if (w * 64 >= rows) v = 0;
    else if (w * 64 + 64 > rows) v &= wmask((uint32_t)(rows - w * 64));
    if (bytes) {
      __syncthreads();  // every thread has read its boundary word
      s_bits[threadIdx.x] = v;
    } else if (w * 64 < rows) {
      dst[w] = v;
    }
  } else {
    // zero-length runs present: resolve every row by binary search, one word per wave pass
    for (uint32_t i = wave_id(); i < W; i += 4) {
      const uint64_t ww = (uint64_t)tile * W + i;
      if (ww * 64 >= rows) break;
      const uint64_t p = ww * 64 + lane_id();
      bool bit = false;
      if (p < rows) {
        const uint32_t k = run_of_row(ends, k0, k1, (uint32_t)p);
        bit = (((k & 1u) ? ~firstmask : firstmask) & 1ull) != 0;
      }
      const uint64_t v = __ballot(bit);
      if (lane_id() == 0) {
        if (bytes) s_bits[i] = v;
        else dst[ww] = v;
      }
    }
  }
  if (bytes) {
    __syncthreads();
    tile_words_to_bytes(s_bits, dst_bytes, p0, rows);
  }
}

// ------------------------------------------------------------- K_SPARSE_BITS
// SparseBoolEncoding::materializeBoolsAsBits (encodings/SparseBoolEncoding.cpp:121-142):
// every row is !sparseValue except the listed (ascending) positions.
template <uint32_t W>
__device__ __forceinline__ void sparse_bits_tile(const KArgs& a, const NodeDesc& nd, uint32_t tile, uint8_t* smem) {
  constexpr uint32_t WPT = W / kBlock;
  constexpr uint32_t TR = W * 64;  // rows per tile
  static_assert(W % kBlock == 0 && (W + 1) * 8 + 16 + 4 * 64 + 16 <= kScratchOff && kBlock == kBitTileWords, "sparse tile geometry");
  // LDS: the tile's nw <= W words, one spare word (its lowest bit takes what
  // falls outside the tile), the waves' stages for packed positions; the
  // tile's range of the list sits in front of the scratch bytes.
  uint64_t* s_bits = reinterpret_cast<uint64_t*>(smem);
  uint32_t* s_k = reinterpret_cast<uint32_t*>(smem + kScratchOff - 16);
  const uint64_t rows = nd.rows;
  const uint32_t K = nd.c0.rows;
  const uint64_t p0 = (uint64_t)tile * TR;
  const uint64_t p1 = p0 + TR < rows ? p0 + TR : rows;
  uint64_t* dst = reinterpret_cast<uint64_t*>(a.work + nd.self.off);
  const uint32_t nw = (uint32_t)((p1 - p0 + 63) >> 6);  // words of this tile
  const uint32_t stage0 = ((nw + 1) * 8 + 15) & ~15u;
  uint32_t lo_ = 0, hi_ = 0;
  with_reader<4>(nd.c0, a, [&](auto rd) {
    // the node's first tile starts at position 0 and its last one ends with the
    // list: no search (a node of one tile does none at all)
    if (wave_id() == 0) {
      const uint32_t key = (uint32_t)p0;
      uint32_t k = 0;
      if (tile != 0) k = wave_first_true(K, [&](uint32_t i) { return rd.get(i) >= key; });
      if (lane_id() == 0) s_k[0] = k;
    } else if (wave_id() == 1) {
      const uint32_t key = (uint32_t)p1;
      uint32_t k = K;
      if (p1 < rows) k = wave_first_true(K, [&](uint32_t i) { return rd.get(i) >= key; });
      if (lane_id() == 0) s_k[1] = k;
    }
#pragma unroll
    for (uint32_t u = 0; u < WPT; ++u) s_bits[u * kBlock + threadIdx.x] = 0;
    if (threadIdx.x == 0) s_bits[W] = 0;
    __syncthreads();
    NBL_PHASE(smem, kSmemBytes, 1);  // descriptor, searches, zeroed tile
    const uint32_t lo = s_k[0], hi = s_k[1];
    lo_ = lo;
    hi_ = hi;
    uint32_t* bits32 = reinterpret_cast<uint32_t*>(s_bits);
    // A listed row sets bit (row - first row of the tile); anything outside the
    // tile -- the sentinel behind the last row, a position of damaged input --
    // is clamped onto the spare bit behind the tile's words instead of being
    // tested for (a wrapped difference is large, so one unsigned minimum does
    // both sides).
    constexpr uint32_t PPT = 8, ROUND = PPT * kBlock;
    const uint32_t tile_row0 = (uint32_t)p0;
    const uint32_t cnt = hi > lo ? hi - lo : 0u;       // positions of this tile (<= its rows)
    const uint32_t spare = nw * 64;
    auto set_row = [&](uint32_t rel) {
      const uint32_t q = rel < spare ? rel : spare;
      atomicOr(&bits32[q >> 5], 1u << (q & 31));
    };
    using RD = decltype(rd);
    if constexpr (RD::kKind == RK_FBW32) {
      // Packed positions of up to 25 bits.  What bounds a dense list is the rate
      // at which a CU takes vector memory instructions (a dword load per
      // position, lanes a few bytes apart, runs at two positions per clock and
      // CU), so the list comes in as 16-byte loads and is taken apart from LDS:
      // a wave copies the bytes of its next `g` x 8 positions (at most 1 KiB)
      // into a stage of its own behind the tile's words -- one load and one LDS
      // store per lane and round, no workgroup barrier: the LDS operations of a
      // wave complete in order -- and lane i < g takes the eight CONSECUTIVE
      // positions that start 8 i w bits into it, each as two aligned dwords and
      // a funnel shift (LDS reads at byte addresses run at a fifth of the rate).
      // Consecutive positions per lane keep the lanes of one LDS atomic eight
      // positions apart, i.e. on different words, where neighbouring positions
      // of a dense list would queue up on one.  Two rounds of loads are in
      // flight per wave.  What does not fill a wave's round goes through the
      // general accessor, two positions per thread.
      typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
      const uint32_t w = rd.w;
      const uint32_t msk = (uint32_t)rd.mask;
      const uint32_t k = (uint32_t)rd.base - tile_row0;  // value field + k = row relative to the tile
      const uint64_t bit0 = (uint64_t)lo * w;            // wave uniform
      const uint32_t sub = (uint32_t)bit0 & 7u;
      const uint32_t room = ((kScratchOff - 16 - stage0) >> 2) & ~15u;  // bytes of LDS per wave
      const uint32_t sb = room < 1056u ? room : 1056u;
      const uint32_t lim = sb - 32 < 1023u ? sb - 32 : 1023u;  // a lane's 16 bytes times 64 lanes hold the round and one byte more
      const uint32_t g = lim / w < 64u ? lim / w : 64u;         // lanes at work
      const uint32_t P = g * 8, Bw = g * w;                      // positions and bytes of a wave's round
      const uint32_t n_wr = cnt / P, tail0 = n_wr * P;
      const uint32_t nl = (Bw + 16) >> 4;                        // lanes that load: Bw + 1 bytes
      const uint32_t lane = lane_id(), wv = wave_id();
      uint8_t* st = smem + stage0 + wv * sb;
      uint32_t tv[2];
      if (tail0 < cnt) {
        const auto list = rd.view(lo);
#pragma unroll
        for (uint32_t u = 0; u < 2; ++u) {
          const uint32_t i = tail0 + threadIdx.x * 2 + u;
          tv[u] = list.at(i < cnt ? i : cnt - 1);
        }
      }
      if (wv < n_wr) {
        const uint8_t* g0 = rd.p + (bit0 >> 3) + lane * 16;
        const uint32_t last = n_wr - 1;
        auto ld = [&](uint32_t j) {
          u32x4 v = {0u, 0u, 0u, 0u};
          if (lane < nl) __builtin_memcpy(&v, g0 + (uint64_t)(j < last ? j : last) * Bw, 16);
          return v;
        };
        u32x4 c0 = ld(wv), c1 = ld(wv + 4);
        for (uint32_t j = wv; j < n_wr; j += 4) {  // wave uniform
          const u32x4 c2 = ld(j + 8);
          if (lane < nl) *reinterpret_cast<u32x4*>(st + lane * 16) = c0;
          if (lane < g) {
            const uint32_t b0 = sub + lane * 8 * w;  // bit offset of the lane's first position in the stage
#pragma unroll
            for (uint32_t u = 0; u < 8; ++u) {
              const uint32_t b = b0 + u * w;
              const uint32_t* d = reinterpret_cast<const uint32_t*>(st + ((b >> 3) & ~3u));
              set_row((__builtin_amdgcn_alignbit(d[1], d[0], b) & msk) + k);  // the shift takes the low five bits of b
            }
          }
          c0 = c1;
          c1 = c2;
        }
      }
      if (tail0 < cnt) {  // threads past the list sit out
#pragma unroll
        for (uint32_t u = 0; u < 2; ++u)
          if (tail0 + threadIdx.x * 2 + u < cnt) set_row(tv[u] - tile_row0);
      }
    } else {
      // wide or constant position fields (never written by the reference's encoder for row numbers): the general accessor
      const auto list = rd.view(lo);
      uint32_t pos[PPT], nxt[PPT];
      auto fetch = [&](uint32_t (&dst)[PPT], uint32_t i0) {  // positions lo + i0 + u * kBlock
#pragma unroll
        for (uint32_t u = 0; u < PPT; ++u) {
          const uint32_t i = i0 + u * kBlock;
          dst[u] = list.at(i < cnt ? i : cnt - 1);
        }
      };
      if (cnt) fetch(pos, threadIdx.x);
      for (uint32_t r0 = 0; r0 < cnt; r0 += ROUND) {  // r0: workgroup uniform
        if (r0 + ROUND < cnt) fetch(nxt, r0 + threadIdx.x + ROUND);
#pragma unroll
        for (uint32_t u = 0; u < PPT; ++u)
          if (r0 + threadIdx.x + u * kBlock < cnt) set_row(pos[u] - tile_row0);
#pragma unroll
        for (uint32_t u = 0; u < PPT; ++u) pos[u] = nxt[u];
      }
    }
  });
  NBL_PHASE(smem, kSmemBytes, 2);  // positions loaded and set
  __syncthreads();
  NBL_PHASE(smem, kSmemBytes, 3);  // barrier
  // The parent that selects by this bitmap takes its rank partials from here
  // (NF_EMIT_PARTS: part_base / aux2 are the parent's, NF_INVERT_SEL = it
  // selects the zero bits), so no counting pass reads the bitmap back.
  const bool emit = (nd.flags & NF_EMIT_PARTS) != 0, zeros = (nd.flags & NF_INVERT_SEL) != 0;
  // The words are stored after the partials are out: a workgroup barrier waits
  // for every store issued before it to be acknowledged, and the partials'
  // reduction has barriers -- stored first, the 12.5 KB of a column would hold
  // the workgroup (and its slot on the CU) for a trip to memory in the middle.
  uint32_t cnt[WPT];
  uint64_t words[WPT];
#pragma unroll
  for (uint32_t u = 0; u < WPT; ++u) {
    const uint64_t w = (uint64_t)tile * W + u * kBlock + threadIdx.x;
    cnt[u] = 0;
    words[u] = 0;
    if (w * 64 < rows) {
      uint64_t v = s_bits[u * kBlock + threadIdx.x];
      if (!(nd.flags & NF_SPARSE_VALUE)) v = ~v;
      const uint64_t m = w * 64 + 64 > rows ? wmask((uint32_t)(rows - w * 64)) : ~0ull;
      v &= m;
      words[u] = v;
      cnt[u] = (uint32_t)__popcll(zeros ? ~v & m : v);
    }
  }
  if (emit) {
    // The partials go out scanned, without a chain: the positions are sorted, so
    // those in front of the tile's first one, `lo` of them, are exactly the
    // listed rows in front of the tile.  The same property is checked for the
    // tile itself -- its range of the list must have put that many distinct bits
    // into it (the sentinel behind the last row aside) -- so a list that is not
    // sorted raises the status bit instead of yielding ranks that do not add up.
    const uint32_t lo = lo_, hi = hi_;
    uint32_t* tmp = reinterpret_cast<uint32_t*>(smem);
    uint32_t* s_part = tmp + 4;
    const uint32_t sub = (uint32_t)(nd.aux2 >> 6);
    const uint32_t total = tile_partials_scan<WPT>(sub, cnt, s_part, tmp);
    const bool listed_selected = ((nd.flags & NF_SPARSE_VALUE) != 0) != zeros;  // the parent selects the listed rows
    const uint32_t before = listed_selected ? lo : (uint32_t)p0 - lo;
    const bool last = p1 == rows;
    write_scanned_partials<WPT>(a, nd.part_base, nd.n_parts, nd.total_idx, sub, tile * (W / sub), before, last, s_part);
    const uint32_t listed_here = listed_selected ? total : (uint32_t)(p1 - p0) - total;
    const uint32_t in_range = hi > lo ? hi - lo : 0u;
    const bool bad = lo > (uint32_t)p0 || (last ? (in_range == 0 || listed_here != in_range - 1) : listed_here != in_range);
    if (bad && threadIdx.x == 0) atomicOr(a.status, ST_SELECT_COUNT);
  }
#pragma unroll
  for (uint32_t u = 0; u < WPT; ++u) {
    const uint64_t w = (uint64_t)tile * W + u * kBlock + threadIdx.x;
    if (w * 64 < rows) dst[w] = words[u];
  }
}

__device__ __forceinline__ void body_sparse_bits(const KArgs& a, const NodeDesc& nd, uint32_t tile, uint8_t* smem) {
  // A tile is 65536 rows (8 KiB of bitmap, four words per thread): the two
  // searches for the tile's range of positions are a chain of dependent loads
  // and the output is tiny, so few large tiles beat many small ones -- unless
  // the column is dense (NF_SPARSE_DENSE, the planner's choice): then a tile's
  // positions take several rounds of loads, and 16384-row tiles put four times
  // the workgroups on them.
  // A node of up to 131072 rows and few positions is one tile (NF_WHOLE_NODE):
  // no search, and the rank partials it emits are final.
  if (nd.flags & NF_WHOLE_NODE) sparse_bits_tile<kWholeTileWords>(a, nd, tile, smem);
  else if (nd.flags & NF_SPARSE_DENSE) sparse_bits_tile<kBitTileWords>(a, nd, tile, smem);
  else sparse_bits_tile<kSparseTileWords>(a, nd, tile, smem);
}

// ---------------------------------------------------------- K_SELECT_SCATTER
// In-tile exclusive scan of the popcounts of the W bitmap words of a tile
// (thread i < W holds word i).  Up to 64 words fit one wave, which then needs
// no barrier of its own.  The prefixes are kept per 32-bit half of every word, as 16-bit
// counts (a tile has at most 16384 rows): a row then finds its rank with one
// dword of bits, one prefix and a 32-bit popcount, whose bit offset inside the
// dword is the same in every pass of a thread over the tile.  Thread i < W
// holds word i; s_bits receives the words (read back as dwords), s_pref the
// 2 W prefixes.  The caller synchronises before reading either.
template <uint32_t W>
__device__ __forceinline__ void tile_rank_scan32(uint64_t word, uint64_t* s_bits, uint16_t* s_pref, uint32_t* tmp) {
  const uint32_t c0 = (uint32_t)__popc((uint32_t)word), c1 = (uint32_t)__popc((uint32_t)(word >> 32));
  if constexpr (W <= 64) {
    if (threadIdx.x < 64) {
      const uint32_t ex = wave_incl_scan(c0 + c1) - c0 - c1;
      if (threadIdx.x < W) {
        s_bits[threadIdx.x] = word;
        reinterpret_cast<uint32_t*>(s_pref)[threadIdx.x] = ex | ((ex + c0) << 16);
      }
    }
  } else {
    uint32_t total;
    const uint32_t ex = block_excl_scan(c0 + c1, tmp, total);
    if (threadIdx.x < W) {
      s_bits[threadIdx.x] = word;
      reinterpret_cast<uint32_t*>(s_pref)[threadIdx.x] = ex | ((ex + c0) << 16);
    }
  }
}
// Rows q0 .. q0 + V - 1 of the tile (q0 a multiple of V, V <= 16): `bits` = their
// selection bits, `r` = the rank of row q0 (set bits of the tile in front of it).
template <uint32_t V>
__device__ __forceinline__ void vector_ranks(const uint32_t* bits32, const uint16_t* pref16, uint32_t q0, uint32_t& r, uint32_t& bits) {
  const uint32_t d = q0 >> 5, b = q0 & 31u;
  const uint32_t w = bits32[d];
  r = (uint32_t)pref16[d] + (uint32_t)__popc(w & ((1u << b) - 1u));
  bits = (w >> b) & ((1u << V) - 1u);
}
// v where bit j of `bits` is set, otherwise f -- without a compare: the bit is
// stretched into a mask (one bit-field extract) and the mask picks (one bit-field insert)
template <typename T>
__device__ __forceinline__ T pick(uint32_t bits, uint32_t j, T v, T f) {
  const T m = (T)(int64_t)(int32_t)__builtin_amdgcn_sbfe(bits, j, 1u);  // 0 or all ones (the builtin's result is unsigned)
  return (T)((v & m) | (f & ~m));
}
// 16-byte store of the vector at element e0 of a tile of `rows` elements whose
// destination is a column of its own in `work` or `out`: those start on 256-byte
// boundaries and own the bytes up to the next one, so the vector holding the
// column's last row is stored whole (its rows past the end carry the fill)
// instead of element by element under a second layer of branches.
template <int ES>
__device__ __forceinline__ void store_vec_padded(typename ElemT<ES>::type* dst, uint32_t e0, uint32_t rows, const Vec16<ES>& x) {
  if (e0 < rows) __builtin_nontemporal_store(x.q, reinterpret_cast<u32x4*>(dst + e0));
}

// Rank scatter whose values child is an RLE (Nullable{RLE{..}}): the runs are
// expanded here, so the expanded column is never written out and read back.
// The tile's ranks [base, base + n) are rows of the RLE node; the runs that
// hold them lie between first_run[] of the RLE tile holding `base` and
// first_run[] of the tile behind the one holding the last rank -- both known
// from scalar loads, so run ends and run values are fetched together.  Run
// boundaries inside the tile's ranks become a bitmap over ranks (as in
// body_rle_expand over rows) and a row's run is a popcount prefix.
template <int ES, typename RD>
__device__ __forceinline__ void select_rle(const KArgs& a, const NodeDesc& nd, const RD& rd, uint8_t* smem, uint64_t word,
                                           uint32_t base, uint32_t n, uint32_t tile_rows,
                                           typename ElemT<ES>::type* dst, typename ElemT<ES>::type fill) {
  using T = typename ElemT<ES>::type;
  constexpr uint32_t V = 16 / ES;
  constexpr uint32_t TE = select_tile_rows(ES);
  constexpr uint32_t W = TE / 64;
  static_assert(W <= 64 && 2 * W <= 256, "outer words, then the run-boundary words, share the bitmap area");
  uint64_t* s_bits = reinterpret_cast<uint64_t*>(smem + kStageBytes);
  uint32_t* s_pref = reinterpret_cast<uint32_t*>(smem + kStageBytes + 2048);
  uint32_t* s_cnt = reinterpret_cast<uint32_t*>(smem + kStageBytes + 3072) + 8;  // runs of the fetched range ending at or before `base`
  const uint32_t R = nd.c1.rows;                 // runs (> 0: the planner fuses non-empty RLE nodes only)
  const uint32_t rows_r = nd.c0.rows;            // rows of the RLE node
  const uint32_t* ends = reinterpret_cast<const uint32_t*>(a.work + nd.c0.off);
  const uint32_t* first_run = ends + (((uint64_t)R + 3) & ~3ull);
  const uint32_t n_rtiles = (rows_r + TE - 1) / TE;
  const uint32_t t0 = base / TE, t1 = (base + n - 1) / TE;   // RLE tiles holding the first and the last rank
  uint32_t kf = sload32(first_run + t0);
  uint32_t kl = t1 + 1 < n_rtiles ? sload32(first_run + t1 + 1) : R - 1;
  kf = kf < R ? kf : R - 1;                                  // payload data: clamped, never followed
  kl = kl < R ? kl : R - 1;
  kl = kl < kf ? kf : kl;
  const uint32_t nsup = kl - kf + 1;                         // runs fetched
  const bool zero_runs = sload32(a.totals + nd.total_idx2 + 1) != 0;
  const bool fast = !zero_runs && rd.fits(nsup) && nsup <= 4 * kBlock;
  if (threadIdx.x < 64) {                                    // the outer ranks: wave 0, as in the plain scatter
    const uint32_t c = (uint32_t)__popcll(word);
    const uint32_t incl = wave_incl_scan(c);
    if (threadIdx.x < W) { s_bits[threadIdx.x] = word; s_pref[threadIdx.x] = incl - c; }
  }
  if (!fast) {
    // zero-length runs, or more runs than the stage holds: every selected row finds its run by binary search
    __syncthreads();
#pragma unroll 1
    for (uint32_t q = threadIdx.x; q < tile_rows; q += kBlock) {
      const uint64_t wd = s_bits[q >> 6];
      T v = fill;
      if ((wd >> (q & 63)) & 1ull) {
        uint32_t r = base + s_pref[q >> 6] + (uint32_t)__popcll(wd & ((1ull << (q & 63)) - 1ull));
        r = r < base + n - 1 ? r : base + n - 1;
        v = rd.get(run_of_row(ends, kf, kl, r));
      }
      dst[q] = v;
    }
    return;
  }
  // run ends of the fetched range (up to four per thread) and the run values, all in flight together
  uint32_t e[4];
#pragma unroll
  for (uint32_t u = 0; u < 4; ++u) {
    const uint32_t j = kf + u * kBlock + threadIdx.x;
    e[u] = j < kl ? ends[j] : 0xFFFFFFFFu;                   // run kl's end is not needed: nothing of the tile lies behind it
  }
  StageRegs<5> regs;
  const uint32_t bit0 = rd.stage_begin(kf, nsup, regs);
  if (threadIdx.x < W) s_bits[W + threadIdx.x] = 0;
  if (threadIdx.x == 0) *s_cnt = 0;
  __syncthreads();
  uint32_t* bits32 = reinterpret_cast<uint32_t*>(s_bits + W);
  uint32_t before = 0;
#pragma unroll
  for (uint32_t u = 0; u < 4; ++u) {
    if (e[u] <= base) ++before;
    else if (e[u] < base + n) atomicOr(&bits32[(e[u] - base) >> 5], 1u << ((e[u] - base) & 31));
  }
  before = wave_sum(before);
  if (lane_id() == 0 && before) atomicAdd(s_cnt, before);
  const auto vals = rd.stage_end(regs, bit0, smem);
  __syncthreads();
  if (threadIdx.x < 64) {                                    // prefix over the boundary words: one wave
    const uint32_t c = threadIdx.x < W ? (uint32_t)__popcll(s_bits[W + threadIdx.x]) : 0u;
    const uint32_t incl = wave_incl_scan(c);
    if (threadIdx.x < W) s_pref[W + threadIdx.x] = incl - c;
  }
  __syncthreads();
  const uint32_t k0 = *s_cnt;                                // run of rank `base`, relative to kf
  const uint32_t lastq = n - 1, lastrun = nsup - 1;
#pragma unroll
  for (uint32_t it = 0; it < 4; ++it) {
    const uint32_t q0 = (it * kBlock + threadIdx.x) * V;
    if (q0 >= tile_rows) break;
    const uint64_t wd = s_bits[q0 >> 6];
    const uint32_t b = q0 & 63;
    uint32_t q = s_pref[q0 >> 6] + (uint32_t)__popcll(wd & ((1ull << b) - 1ull));  // rank relative to base
    const uint32_t bits = (uint32_t)(wd >> b);
    Vec16<ES> o;
    if ((bits & ((1u << V) - 1u)) == 0) {
#pragma unroll
      for (uint32_t j = 0; j < V; ++j) o.v[j] = fill;
    } else {
      o.q = make_u32x4(0, 0, 0, 0);
#pragma unroll
      for (uint32_t j = 0; j < V; ++j) {
        const uint32_t qc = q < lastq ? q : lastq;
        const uint64_t bw = s_bits[W + (qc >> 6)];
        uint32_t run = k0 + s_pref[W + (qc >> 6)] + (uint32_t)__popcll(bw & ((2ull << (qc & 63)) - 1ull));
        run = run < lastrun ? run : lastrun;
        const T v = vals.at(run);
        vec_set<ES>(o, j, ((bits >> j) & 1u) ? v : fill);
        q += (bits >> j) & 1u;
      }
    }
    store_vec<ES>(dst, q0, tile_rows, o);
  }
}

// Rank scatter whose values child is a MainlyConstant (Nullable{MainlyConstant{..}},
// the usual shape of a sparse feature column): both selections are resolved in
// one pass, so the inner column is never written out and read back.
//   row p is selected by the outer bitmap -> outer rank r1 = a row of the inner node
//   inner bit of r1 set (after inversion: "not common") -> inner rank r2 -> others[r2]
//   otherwise the inner fill (the common value); rows the outer leaves out take `fill`.
// The tile's outer ranks are [base, base + n).  Inner ranks come from the inner
// node's own scanned partials (one per TE inner rows): the partial of the inner
// tile holding `base` plus a popcount prefix over the inner bitmap words from
// that tile's start, kept in LDS behind the outer words.  The other values the
// tile can ask for lie between two of those partials, known before any bitmap
// word arrives, so they are staged in LDS like the values of a plain scatter.
template <int ES, typename RD>
__device__ __forceinline__ void select_nested(const KArgs& a, const NodeDesc& nd, const RD& rd, uint8_t* smem, uint64_t word,
                                              uint32_t base, uint32_t n, uint32_t tile_rows,
                                              typename ElemT<ES>::type* dst, typename ElemT<ES>::type fill) {
  using T = typename ElemT<ES>::type;
  constexpr uint32_t V = 16 / ES;
  constexpr uint32_t TE = select_tile_rows(ES);
  constexpr uint32_t W = TE / 64;
  static_assert(W <= 64 && 3 * W <= 256 && V <= 4, "outer words, then up to two inner tiles of words, share the bitmap area");
  uint64_t* s_bits = reinterpret_cast<uint64_t*>(smem + kStageBytes);
  uint16_t* s_pref = reinterpret_cast<uint16_t*>(smem + kStageBytes + 2048);
  uint8_t* s_dep = smem + kDepositOff;
  const T fill2 = (T)nd.aux3;
  const uint32_t nothers = nd.c1.rows;
  // inner tiles j0..j1 hold the outer ranks of this tile (n <= TE: at most two of them)
  const uint32_t j0 = base / TE, j1 = (base + n - 1) / TE;
  uint32_t ob = sload32(a.parts + nd.part_base2 + j0);      // inner rank at inner row j0 * TE
  uint32_t oe = sload32(a.parts + nd.part_base2 + j1 + 1);  // inner rank behind inner tile j1
  ob = ob < nothers ? ob : nothers;                         // scan results of damaged input are clamped, never followed
  oe = oe < nothers ? oe : nothers;
  oe = oe < ob ? ob : oe;
  NBL_PHASE(smem, kSmemBytes, 2);  // the outer word (requested by the caller) and the inner partials
  const uint32_t cnt2 = oe - ob;
  const bool staged = rd.fits(cnt2);
  const uint32_t w0 = j0 * W;                                // first inner word needed
  const uint32_t nw = (base + n - 1) / 64 - w0 + 1;          // <= 2 W
  const bool inv2 = true;                                    // MainlyConstant: other values where isCommon is 0
  // wave 1 fetches and scans the inner words (lane l: words l and l + 64) while wave 0 scans the outer ones
  uint64_t iw0 = 0, iw1 = 0;
  const uint32_t l = threadIdx.x - 64;
  if (threadIdx.x >= 64 && threadIdx.x < 128) {
    const BitsReader inner(nd.c0, a, inv2);
    const uint64_t r0 = inner.word(w0 + (l < nw ? l : 0));               // both requested before either is used
    const uint64_t r1 = inner.word(w0 + (l + 64 < nw ? l + 64 : 0));
    iw0 = l < nw ? r0 : 0;
    iw1 = l + 64 < nw ? r1 : 0;
  }
  StageRegs<5> regs;
  regs.nvec = 0;
  uint32_t bit0 = 0;
  if (staged && cnt2) bit0 = rd.stage_begin(ob, cnt2, regs);
  NBL_PHASE(smem, kSmemBytes, 3);  // inner words and staged values (this wave's share)
  // Deposit table: entry (inner bits << V | outer bits) spreads the low inner
  // bits over the set positions of the outer bits -- which of a vector's V rows
  // take an other value.  Its 4^V entries cost each thread a handful of
  // operations once, against a serial walk over the rows of every vector.
  if (threadIdx.x < (1u << (2 * V))) {
    const uint32_t sel = threadIdx.x & ((1u << V) - 1u);
    uint32_t ib = threadIdx.x >> V, e = 0;
#pragma unroll
    for (uint32_t j = 0; j < V; ++j) {
      const uint32_t sj = (sel >> j) & 1u;
      e |= (sj & ib) << j;
      ib >>= sj;
    }
    s_dep[threadIdx.x] = (uint8_t)e;
  }
  if (threadIdx.x < 64) {
    tile_rank_scan32<W>(word, s_bits, s_pref, nullptr);
  } else if (threadIdx.x < 128) {
    // inner ranks relative to inner row j0 * TE, i.e. to other value `ob`
    const uint32_t a0 = (uint32_t)__popc((uint32_t)iw0), a1 = (uint32_t)__popc((uint32_t)(iw0 >> 32));
    const uint32_t b0 = (uint32_t)__popc((uint32_t)iw1), b1 = (uint32_t)__popc((uint32_t)(iw1 >> 32));
    const uint32_t i0 = wave_incl_scan(a0 + a1);
    const uint32_t t0 = (uint32_t)__builtin_amdgcn_readlane((int)i0, 63);
    const uint32_t i1 = wave_incl_scan(b0 + b1);
    uint32_t* pref2 = reinterpret_cast<uint32_t*>(s_pref + 2 * W);
    const uint32_t e0 = i0 - a0 - a1, e1 = t0 + i1 - b0 - b1;
    // every lane writes: words past nw are zero, and the window of the last row reads one dword past its own
    s_bits[W + l] = iw0; pref2[l] = e0 | ((e0 + a0) << 16);
    s_bits[W + 64 + l] = iw1; pref2[64 + l] = e1 | ((e1 + b0) << 16);
  }
  const auto vals = rd.stage_end(regs, bit0, smem);  // regs.nvec == 0 when nothing was staged
  NBL_PHASE(smem, kSmemBytes, 4);  // scans and LDS writes
  __syncthreads();
  NBL_PHASE(smem, kSmemBytes, 5);  // barrier
  const uint32_t* ob32 = reinterpret_cast<const uint32_t*>(s_bits);       // the outer words as dwords
  const uint32_t* ib32 = reinterpret_cast<const uint32_t*>(s_bits + W);   // the inner words as dwords, from word w0
  const uint16_t* ipref = s_pref + 2 * W;
  const uint32_t xoff = base - w0 * 64;               // inner row of outer rank 0, relative to the first inner word held
  const uint32_t xlast = xoff + n - 1;                // last inner row this tile may ask for
  const uint32_t last2 = nothers ? nothers - 1 : 0;
  if (!staged) {
    // more other values than the stage holds (a dense inner selection): row by row from global memory
#pragma unroll 1
    for (uint32_t q = threadIdx.x; q < tile_rows; q += kBlock) {
      const uint32_t wd = ob32[q >> 5];
      T v = fill;
      if ((wd >> (q & 31)) & 1u) {
        uint32_t x = xoff + s_pref[q >> 5] + (uint32_t)__popc(wd & ((1u << (q & 31)) - 1u));
        x = x < xlast ? x : xlast;
        const uint32_t wd2 = ib32[x >> 5];
        v = fill2;
        if (((wd2 >> (x & 31)) & 1u) && nothers) {
          const uint32_t r2 = ob + ipref[x >> 5] + (uint32_t)__popc(wd2 & ((1u << (x & 31)) - 1u));
          v = rd.get(r2 < last2 ? r2 : last2);
        }
      }
      dst[q] = v;
    }
    return;
  }
  // One 16-byte vector of rows at a time, without a branch inside the wave:
  // its selected rows are consecutive inner rows x, x + 1, ..., so their inner
  // bits are one short window of the inner bitmap; the deposit table turns
  // window and outer bits into the rows that take an other value, and row j's
  // value is the one at inner rank r2 + (such rows in front of j).
#pragma unroll
  for (uint32_t it = 0; it < 4; ++it) {
    const uint32_t q0 = (it * kBlock + threadIdx.x) * V;
    uint32_t r, sel;
    vector_ranks<V>(ob32, s_pref, q0, r, sel);
    uint32_t x = xoff + r;
    x = x < xlast ? x : xlast;
    const uint32_t d = x >> 5, sh = x & 31u;
    const uint32_t lo = ib32[d], hi = ib32[d + 1];
    const uint32_t ib = __builtin_amdgcn_alignbit(hi, lo, sh) & ((1u << V) - 1u);  // inner bits of rows x, x + 1, ...
    uint32_t e = s_dep[(ib << V) | sel];
    e = cnt2 ? e : 0u;
    Vec16<ES> o;
    o.q = make_u32x4(0, 0, 0, 0);
    if (__ballot(e != 0) == 0) {  // none of the wave's rows in this pass takes an other value (sparse inner selections)
#pragma unroll
      for (uint32_t j = 0; j < V; ++j) vec_set<ES>(o, j, pick<T>(sel, j, fill2, fill));
    } else {
      const uint32_t r2 = (uint32_t)ipref[d] + (uint32_t)__popc(lo & ((1u << sh) - 1u));  // inner rank at row x, relative to ob
      const uint32_t klast = cnt2 - 1;  // some e != 0 here, so cnt2 != 0
#pragma unroll
      for (uint32_t j = 0; j < V; ++j) {
        const uint32_t k = r2 + (uint32_t)__popc(e & ((1u << j) - 1u));
        const T v = vals.at(k < klast ? k : klast);
        vec_set<ES>(o, j, pick<T>(e, j, v, pick<T>(sel, j, fill2, fill)));
      }
    }
    store_vec_padded<ES>(dst, q0, tile_rows, o);
  }
#ifdef NBL_PHASE_TIMING
  __builtin_amdgcn_sched_barrier(0);
  if (threadIdx.x == 0) {  // the loop's instructions, stores issued but not waited for
    unsigned long long* tp_ = reinterpret_cast<unsigned long long*>(smem + kSmemBytes);
    const unsigned long long t_ = __builtin_readcyclecounter();
    g_phase[(blockIdx.x & (kPhaseTiles - 1)) * 8 + 6] = (uint32_t)(t_ - *tp_);
    *tp_ = t_;
  }
#endif
}

// NullableEncoding::materialize (encodings/NullableEncoding.h:207-236): row i
// takes values[rank(i)] when its bit is set, physicalType() otherwise; and
// MainlyConstantEncoding::materialize (encodings/MainlyConstantEncoding.h:208-253):
// the common value where isCommon is set, otherValues[rank0(i)] elsewhere.
//
// The selected rows of a tile take the consecutive ranks [parts[t], parts[t+1]),
// so the tile's values are one contiguous range, staged in LDS while the
// bitmap words are scanned.  `pslot` = the slot of the tile's partial in
// `parts`, handed over by the tile table so that this load does not wait for
// the node descriptor.
template <int ES>
__device__ __forceinline__ void body_select_scatter(const KArgs& a, const NodeDesc& nd, uint32_t tile, uint8_t* smem, uint32_t pslot) {
  using T = typename ElemT<ES>::type;
  constexpr uint32_t V = 16 / ES;
  constexpr uint32_t TE = select_tile_rows(ES);
  constexpr uint32_t W = TE / 64;
  static_assert(TE * ES == kTileBytes && W <= kBlock, "tile geometry");
  const uint32_t base = sload32(a.parts + pslot);
  const uint32_t next = sload32(a.parts + pslot + 1);  // the node's total sits behind its last partial
  uint64_t* s_bits = reinterpret_cast<uint64_t*>(smem + kStageBytes);         // W <= 256 words, read back as 2 W dwords
  uint16_t* s_pref = reinterpret_cast<uint16_t*>(smem + kStageBytes + 2048);  // 2 W prefixes, one per dword
  uint32_t* tmp = reinterpret_cast<uint32_t*>(smem + kStageBytes + 3072);
  const uint32_t* bits32 = reinterpret_cast<const uint32_t*>(s_bits);
  const uint64_t rows = nd.rows;
  const uint64_t p0 = (uint64_t)tile * TE;
  const uint32_t tile_rows = rows - p0 < TE ? (uint32_t)(rows - p0) : TE;
  const bool invert = (nd.flags & NF_INVERT_SEL) != 0;
  T* dst = reinterpret_cast<T*>(self_ptr(nd, a)) + p0;
  bool nested = false, runs = false;
  if constexpr (ES >= 4) {  // the planner fuses 4- and 8-byte rows only
    nested = (nd.flags & NF_NESTED_SELECT) != 0;
    runs = (nd.flags & NF_RLE_VALUES) != 0;
  }
  const uint32_t nvals = (nested || runs) ? nd.c0.rows : nd.c1.rows;  // rows of the values child
  if (tile == 0 && threadIdx.x == 0) {
    bool bad = sload32(a.totals + nd.total_idx) != nvals;
    if (nested) bad |= sload32(a.totals + nd.total_idx2) != nd.c1.rows;
    if (bad) atomicOr(a.status, ST_SELECT_COUNT);
    if (runs && sload32(a.totals + nd.total_idx2) != nvals) atomicOr(a.status, ST_RLE_SUM_MISMATCH);
  }
  // The tile's selected rows take the consecutive ranks [base, next): both come
  // from the scanned partials, so the values can be fetched without waiting for
  // the bitmap.  Counts that disagree with the values child (corrupt input) are
  // clamped, never followed.
  NBL_PHASE(smem, kSmemBytes, 1);  // descriptor and the tile's partials
  const uint32_t count = next - base < TE ? next - base : TE;
  const uint32_t avail = base < nvals ? nvals - base : 0u;
  const uint32_t n = count < avail ? count : avail;
  const uint64_t w = (uint64_t)tile * W + threadIdx.x;
  uint64_t word = 0;
  if (threadIdx.x < W && w * 64 < rows) word = BitsReader(nd.sel, a, invert).word(w);
  const T fill = (T)nd.aux;
  auto write_nulls = [&]() {
    if ((nd.flags & NF_WRITE_NULLS) && threadIdx.x < W && w * 64 < rows)
      reinterpret_cast<uint64_t*>(a.out + nd.aux_off)[w] = invert ? ~word & (w * 64 + 64 > rows ? wmask((uint32_t)(rows - w * 64)) : ~0ull) : word;
  };
  if (n == 0) {  // nothing selected (or nothing to select from): a fill, no ranks needed
    write_nulls();
    Vec16<ES> x;
#pragma unroll
    for (uint32_t j = 0; j < V; ++j) x.v[j] = fill;
#pragma unroll
    for (uint32_t it = 0; it < 4; ++it) {
      const uint32_t q0 = (it * kBlock + threadIdx.x) * V;
      if (q0 < tile_rows) store_vec<ES>(dst, q0, tile_rows, x);
    }
    return;
  }
  const uint32_t last = n - 1;
  // (the copy of the bitmap goes out behind everything the tile loads: a wave's
  // loads return in order with its stores, so a load issued behind a store
  // waits for that store to be acknowledged)
  if constexpr (ES >= 4) if (runs) {
    with_reader<ES>(nd.c1, a, [&](auto rd) { select_rle<ES>(a, nd, rd, smem, word, base, n, tile_rows, dst, fill); });
    write_nulls();
    return;
  }
  if constexpr (ES >= 4) if (nested) {
    with_reader<ES>(nd.c1, a, [&](auto rd) { select_nested<ES>(a, nd, rd, smem, word, base, n, tile_rows, dst, fill); });
    write_nulls();
    return;
  }
  if constexpr (ES >= 4) if (nd.flags & NF_DICT_VALUES) {  // the planner fuses 4- and 8-byte rows only
    // The values child is a Dictionary whose gather is done here: c1 = its uint32
    // indices (staged like any values), c0 = its alphabet as a plain element
    // array.  The gathered rows never exist as an intermediate column.
    const Reader<RK_PTR, ES> alpha(nd.c0, a);
    bool bad = false;
    const uint32_t amax = nd.c0.rows - 1;  // the planner fuses non-empty alphabets only
    with_reader<4>(nd.c1, a, [&](auto rd) {
      StageRegs<5> regs;
      const uint32_t bit0 = rd.stage_begin(base, n, regs);
      write_nulls();
      tile_rank_scan32<W>(word, s_bits, s_pref, tmp);
      const auto idxs = rd.stage_end(regs, bit0, smem);
      __syncthreads();
      constexpr uint32_t IB = ES == 8 ? 4 : 2;  // rounds whose gathers are in flight together (8 rows)
#pragma unroll
      for (uint32_t ib = 0; ib < 4; ib += IB) {
        uint32_t ix[IB][V], sel[IB];
#pragma unroll
        for (uint32_t u = 0; u < IB; ++u) {
          const uint32_t q0 = ((ib + u) * kBlock + threadIdx.x) * V;  // rows past the tile have no bits and store nothing
          uint32_t r;
          vector_ranks<V>(bits32, s_pref, q0, r, sel[u]);
#pragma unroll
          for (uint32_t j = 0; j < V; ++j) {
            const uint32_t rj = r + (uint32_t)__popc(sel[u] & ((1u << j) - 1u));
            uint32_t i = idxs.at(rj < last ? rj : last);
            if (i > amax) { bad = true; i = amax; }
            ix[u][j] = i;
          }
        }
        T x[IB][V];
#pragma unroll
        for (uint32_t u = 0; u < IB; ++u)
#pragma unroll
          for (uint32_t j = 0; j < V; ++j) x[u][j] = alpha.get(ix[u][j]);
#pragma unroll
        for (uint32_t u = 0; u < IB; ++u) {
          const uint32_t q0 = ((ib + u) * kBlock + threadIdx.x) * V;
          Vec16<ES> o;
          o.q = make_u32x4(0, 0, 0, 0);
#pragma unroll
          for (uint32_t j = 0; j < V; ++j) vec_set<ES>(o, j, pick<T>(sel[u], j, x[u][j], fill));
          store_vec_padded<ES>(dst, q0, tile_rows, o);
        }
      }
    });
    if (bad) atomicOr(a.status, ST_DICT_INDEX_RANGE);
    return;
  }
  // The n values are copied as raw bytes into LDS with aligned streaming loads
  // (each byte fetched once) while the bitmap words are still in flight; every
  // row then unpacks its value -- or takes the fill -- from LDS at its rank.
  with_reader<ES>(nd.c1, a, [&](auto rd) {
    StageRegs<5> regs;
    const uint32_t bit0 = rd.stage_begin(base, n, regs);
    write_nulls();
    tile_rank_scan32<W>(word, s_bits, s_pref, tmp);
    const auto vals = rd.stage_end(regs, bit0, smem);
    __syncthreads();
    // no branch inside the wave: row j of a vector reads the value at rank
    // r + (selected rows in front of j) whether it is selected or not, and the
    // selection bit picks between it and the fill.  Rows past the tile have no
    // bits and are not stored.
#pragma unroll
    for (uint32_t it = 0; it < 4; ++it) {
      const uint32_t q0 = (it * kBlock + threadIdx.x) * V;
      uint32_t r, bits;
      vector_ranks<V>(bits32, s_pref, q0, r, bits);
      Vec16<ES> o;
      o.q = make_u32x4(0, 0, 0, 0);
      if (__ballot(bits != 0) == 0) {  // nothing selected in the wave's rows of this pass (sparse selections)
#pragma unroll
        for (uint32_t j = 0; j < V; ++j) vec_set<ES>(o, j, fill);
      } else {
#pragma unroll
        for (uint32_t j = 0; j < V; ++j) {
          const uint32_t rj = r + (uint32_t)__popc(bits & ((1u << j) - 1u));
          const T v = vals.at(rj < last ? rj : last);
          vec_set<ES>(o, j, pick<T>(bits, j, v, fill));
        }
      }
      store_vec_padded<ES>(dst, q0, tile_rows, o);
    }
  });
}

// ------------------------------------------------------ K_SELECT_SCATTER_WAVE
// The same rank scatter with ONE wave per 16 KiB tile, for columns whose tiles
// stage few values (sparse selections; nested ones, whose staged values are
// the few "other" ones).  A tile spends most of its life in a chain of
// dependent loads -- tile entry, descriptor and partials, inner partials,
// values -- each a trip to memory that the tile's own stores keep busy, and a
// CU holds 8 four-wave workgroups but 32 one-wave ones: four times the chains
// in flight for the same bytes, waves and instructions.  No barrier but the
// wave's own; the four-wave body stays for dense selections, whose staged
// values need its 16 KiB of LDS, and for the Dictionary / RLE values children.
// The stage holds what stage_load() copies for kWaveStageBytes of values: up to
// one more byte when the first value starts inside a byte, up to 15 bytes in
// front of it (the copy starts 16-byte aligned), the window's fifth dword, and
// the round-up to whole 16-byte vectors.
constexpr uint32_t kWaveStageVecs = (15 + (kWaveStageBytes + 1) + 4 + 15) >> 4;
constexpr uint32_t kWaveBitsOff = kWaveStageVecs * 16;              // 64 outer words, up to 128 inner ones, a window's slack
static_assert(kWaveStageVecs <= 3 * kWaveBlock, "StageRegs<3> of a 64-thread workgroup cover the stage");
constexpr uint32_t kWaveBitsBytes = 192 * 8 + 16;
constexpr uint32_t kWavePrefOff = kWaveBitsOff + kWaveBitsBytes;    // 384 16-bit prefixes (one per dword of bits)
constexpr uint32_t kWaveDepOff = kWavePrefOff + 384 * 2;            // deposit table of the nested selection
constexpr uint32_t kWaveSmemBytes = kWaveDepOff + 256;
static_assert(kWaveSmemBytes * 32 <= 160 * 1024 && kWaveBitsOff % 16 == 0 && kWavePrefOff % 4 == 0, "32 one-wave workgroups per CU");

template <int ES, typename RD>
__device__ __forceinline__ void wave_select_nested(const KArgs& a, const NodeDesc& nd, const RD& rd, uint8_t* smem, uint64_t word,
                                                   uint32_t base, uint32_t n, uint32_t tile_rows,
                                                   typename ElemT<ES>::type* dst, typename ElemT<ES>::type fill) {
  using T = typename ElemT<ES>::type;
  constexpr uint32_t V = 16 / ES;
  constexpr uint32_t TE = select_tile_rows(ES);
  constexpr uint32_t W = TE / 64;
  constexpr uint32_t ITS = TE / (kWaveBlock * V);
  static_assert(W <= 64 && V <= 4, "one word per lane; deposit table of 4^V entries");
  uint64_t* s_bits = reinterpret_cast<uint64_t*>(smem + kWaveBitsOff);
  uint16_t* s_pref = reinterpret_cast<uint16_t*>(smem + kWavePrefOff);
  uint8_t* s_dep = smem + kWaveDepOff;
  const uint32_t lane = threadIdx.x;
  const T fill2 = (T)nd.aux3;
  const uint32_t nothers = nd.c1.rows;
  const uint32_t j0 = base / TE, j1 = (base + n - 1) / TE;    // inner tiles holding this tile's outer ranks (at most two)
  uint32_t ob = sload32(a.parts + nd.part_base2 + j0);
  uint32_t oe = sload32(a.parts + nd.part_base2 + j1 + 1);
  ob = ob < nothers ? ob : nothers;                           // scan results of damaged input are clamped, never followed
  oe = oe < nothers ? oe : nothers;
  oe = oe < ob ? ob : oe;
  const uint32_t cnt2 = oe - ob;
  const bool staged = rd.fits(cnt2, kWaveStageBytes);
  const uint32_t w0 = j0 * W;
  const uint32_t nw = (base + n - 1) / 64 - w0 + 1;           // <= 2 W <= 128: two inner words per lane
  const BitsReader inner(nd.c0, a, true);                     // MainlyConstant: other values where isCommon is 0
  const uint64_t r0 = inner.word(w0 + (lane < nw ? lane : 0));
  const uint64_t r1 = inner.word(w0 + (lane + 64 < nw ? lane + 64 : 0));
  StageRegs<3> regs;
  regs.nvec = 0;
  uint32_t bit0 = 0;
  if (staged && cnt2) bit0 = rd.template stage_begin<3, kWaveBlock>(ob, cnt2, regs);
  for (uint32_t i = lane; i < (1u << (2 * V)); i += kWaveBlock) {  // deposit table, as in the four-wave body
    const uint32_t sel = i & ((1u << V) - 1u);
    uint32_t ib = i >> V, e = 0;
#pragma unroll
    for (uint32_t j = 0; j < V; ++j) {
      const uint32_t sj = (sel >> j) & 1u;
      e |= (sj & ib) << j;
      ib >>= sj;
    }
    s_dep[i] = (uint8_t)e;
  }
  tile_rank_scan32<W>(word, s_bits, s_pref, nullptr);
  {
    const uint64_t iw0 = lane < nw ? r0 : 0, iw1 = lane + 64 < nw ? r1 : 0;
    const uint32_t a0 = (uint32_t)__popc((uint32_t)iw0), a1 = (uint32_t)__popc((uint32_t)(iw0 >> 32));
    const uint32_t b0 = (uint32_t)__popc((uint32_t)iw1), b1 = (uint32_t)__popc((uint32_t)(iw1 >> 32));
    const uint32_t i0 = wave_incl_scan(a0 + a1);
    const uint32_t t0 = (uint32_t)__builtin_amdgcn_readlane((int)i0, 63);
    const uint32_t i1 = wave_incl_scan(b0 + b1);
    uint32_t* pref2 = reinterpret_cast<uint32_t*>(s_pref + 2 * W);
    const uint32_t e0 = i0 - a0 - a1, e1 = t0 + i1 - b0 - b1;
    s_bits[W + lane] = iw0; pref2[lane] = e0 | ((e0 + a0) << 16);
    s_bits[W + 64 + lane] = iw1; pref2[64 + lane] = e1 | ((e1 + b0) << 16);
  }
  rd.template stage_end<3, kWaveBlock>(regs, bit0, smem);
  wave_lds_sync();
  const uint32_t* ob32 = reinterpret_cast<const uint32_t*>(s_bits);
  const uint32_t* ib32 = reinterpret_cast<const uint32_t*>(s_bits + W);
  const uint16_t* ipref = s_pref + 2 * W;
  const uint32_t xoff = base - w0 * 64;
  const uint32_t xlast = xoff + n - 1;
  const uint32_t last2 = nothers ? nothers - 1 : 0;
  // [ob, oe) are the other values of the whole inner tiles this tile touches;
  // where those are more than the stage holds, the inner ranks of the tile's
  // first and last row (known now) bound what it reads itself: at most n.
  uint32_t kfirst = 0, kcount = cnt2;
  bool have = staged;
  if (!staged) {
    auto rank_at = [&](uint32_t x) { return (uint32_t)ipref[x >> 5] + (uint32_t)__popc(ib32[x >> 5] & ((1u << (x & 31)) - 1u)); };
    kfirst = rank_at(xoff);
    const uint32_t kend = rank_at(xlast) + ((ib32[xlast >> 5] >> (xlast & 31)) & 1u);
    kfirst = kfirst < cnt2 ? kfirst : cnt2;
    kcount = kend > kfirst ? kend - kfirst : 0u;
    kcount = kcount < cnt2 - kfirst ? kcount : cnt2 - kfirst;
    have = rd.fits(kcount, kWaveStageBytes);
    if (have) {
      regs.nvec = 0;
      bit0 = 0;
      if (kcount) bit0 = rd.template stage_begin<3, kWaveBlock>((uint64_t)ob + kfirst, kcount, regs);
      rd.template stage_end<3, kWaveBlock>(regs, bit0, smem);
      wave_lds_sync();
    }
  }
  const auto vals = rd.staged_at(bit0, smem);
  if (!have) {  // a tile that reads more other values than the stage holds: row by row from global memory
#pragma unroll 1
    for (uint32_t q = lane; q < tile_rows; q += kWaveBlock) {
      const uint32_t wd = ob32[q >> 5];
      T v = fill;
      if ((wd >> (q & 31)) & 1u) {
        uint32_t x = xoff + s_pref[q >> 5] + (uint32_t)__popc(wd & ((1u << (q & 31)) - 1u));
        x = x < xlast ? x : xlast;
        const uint32_t wd2 = ib32[x >> 5];
        v = fill2;
        if (((wd2 >> (x & 31)) & 1u) && nothers) {
          const uint32_t r2 = ob + ipref[x >> 5] + (uint32_t)__popc(wd2 & ((1u << (x & 31)) - 1u));
          v = rd.get(r2 < last2 ? r2 : last2);
        }
      }
      dst[q] = v;
    }
    return;
  }
#pragma unroll 4
  for (uint32_t it = 0; it < ITS; ++it) {
    const uint32_t q0 = (it * kWaveBlock + lane) * V;
    uint32_t r, sel;
    vector_ranks<V>(ob32, s_pref, q0, r, sel);
    uint32_t x = xoff + r;
    x = x < xlast ? x : xlast;
    const uint32_t d = x >> 5, sh = x & 31u;
    const uint32_t lo = ib32[d], hi = ib32[d + 1];
    const uint32_t ib = __builtin_amdgcn_alignbit(hi, lo, sh) & ((1u << V) - 1u);
    uint32_t e = s_dep[(ib << V) | sel];
    e = kcount ? e : 0u;
    Vec16<ES> o;
    o.q = make_u32x4(0, 0, 0, 0);
    if (__ballot(e != 0) == 0) {
#pragma unroll
      for (uint32_t j = 0; j < V; ++j) vec_set<ES>(o, j, pick<T>(sel, j, fill2, fill));
    } else {
      // inner rank at row x, relative to the first staged value (an unsigned
      // wrap on damaged input ends in the clamp)
      const uint32_t r2 = (uint32_t)ipref[d] + (uint32_t)__popc(lo & ((1u << sh) - 1u)) - kfirst;
      const uint32_t klast = kcount - 1;
#pragma unroll
      for (uint32_t j = 0; j < V; ++j) {
        const uint32_t k = r2 + (uint32_t)__popc(e & ((1u << j) - 1u));
        const T v = vals.at(k < klast ? k : klast);
        vec_set<ES>(o, j, pick<T>(e, j, v, pick<T>(sel, j, fill2, fill)));
      }
    }
    store_vec_padded<ES>(dst, q0, tile_rows, o);
  }
}

template <int ES>
__device__ __forceinline__ void body_select_scatter_wave(const KArgs& a, const NodeDesc& nd, uint32_t tile, uint8_t* smem, uint32_t pslot) {
  using T = typename ElemT<ES>::type;
  constexpr uint32_t V = 16 / ES;
  constexpr uint32_t TE = select_tile_rows(ES);
  constexpr uint32_t W = TE / 64;
  constexpr uint32_t ITS = TE / (kWaveBlock * V);
  static_assert(ES >= 4 && W <= 64, "4- and 8-byte rows: at most one bitmap word per lane");
  const uint32_t lane = threadIdx.x;
  const uint32_t base = sload32(a.parts + pslot);
  const uint32_t next = sload32(a.parts + pslot + 1);
  uint64_t* s_bits = reinterpret_cast<uint64_t*>(smem + kWaveBitsOff);
  uint16_t* s_pref = reinterpret_cast<uint16_t*>(smem + kWavePrefOff);
  const uint32_t* bits32 = reinterpret_cast<const uint32_t*>(s_bits);
  const uint64_t rows = nd.rows;
  const uint64_t p0 = (uint64_t)tile * TE;
  const uint32_t tile_rows = rows - p0 < TE ? (uint32_t)(rows - p0) : TE;
  const bool invert = (nd.flags & NF_INVERT_SEL) != 0;
  const bool nested = (nd.flags & NF_NESTED_SELECT) != 0;
  T* dst = reinterpret_cast<T*>(self_ptr(nd, a)) + p0;
  const uint32_t nvals = nested ? nd.c0.rows : nd.c1.rows;
  if (tile == 0 && lane == 0) {
    bool bad = sload32(a.totals + nd.total_idx) != nvals;
    if (nested) bad |= sload32(a.totals + nd.total_idx2) != nd.c1.rows;
    if (bad) atomicOr(a.status, ST_SELECT_COUNT);
  }
  const uint32_t count = next - base < TE ? next - base : TE;
  const uint32_t avail = base < nvals ? nvals - base : 0u;
  const uint32_t n = count < avail ? count : avail;
  const uint64_t w = (uint64_t)tile * W + lane;
  uint64_t word = 0;
  if (lane < W && w * 64 < rows) word = BitsReader(nd.sel, a, invert).word(w);
  const T fill = (T)nd.aux;
  // the copy of the bitmap goes out behind everything the tile loads: a wave's
  // loads return in order with its stores, so a load issued behind a store
  // waits for that store to be acknowledged
  auto write_nulls = [&]() {
    if ((nd.flags & NF_WRITE_NULLS) && lane < W && w * 64 < rows)
      reinterpret_cast<uint64_t*>(a.out + nd.aux_off)[w] = invert ? ~word & (w * 64 + 64 > rows ? wmask((uint32_t)(rows - w * 64)) : ~0ull) : word;
  };
  if (n == 0) {  // nothing selected (or nothing to select from): a fill
    write_nulls();
    Vec16<ES> x;
#pragma unroll
    for (uint32_t j = 0; j < V; ++j) x.v[j] = fill;
#pragma unroll 4
    for (uint32_t it = 0; it < ITS; ++it) store_vec_padded<ES>(dst, (it * kWaveBlock + lane) * V, tile_rows, x);
    return;
  }
  if (nested) {
    with_reader<ES>(nd.c1, a, [&](auto rd) { wave_select_nested<ES>(a, nd, rd, smem, word, base, n, tile_rows, dst, fill); });
    write_nulls();
    return;
  }
  with_reader<ES>(nd.c1, a, [&](auto rd) {
    const bool staged = rd.fits(n, kWaveStageBytes);
    StageRegs<3> regs;
    regs.nvec = 0;
    uint32_t bit0 = 0;
    if (staged) bit0 = rd.template stage_begin<3, kWaveBlock>(base, n, regs);
    write_nulls();  // behind the loads of the staged values
    tile_rank_scan32<W>(word, s_bits, s_pref, nullptr);
    const auto vals = rd.template stage_end<3, kWaveBlock>(regs, bit0, smem);
    wave_lds_sync();
    const uint32_t last = n - 1;
    if (!staged) {  // a tile denser than the column's average: its values row by row from global memory
#pragma unroll 1
      for (uint32_t q = lane; q < tile_rows; q += kWaveBlock) {
        const uint32_t wd = bits32[q >> 5];
        T v = fill;
        if ((wd >> (q & 31)) & 1u) {
          const uint32_t r = s_pref[q >> 5] + (uint32_t)__popc(wd & ((1u << (q & 31)) - 1u));
          v = rd.get((uint64_t)base + (r < last ? r : last));
        }
        dst[q] = v;
      }
      return;
    }
#pragma unroll 4
    for (uint32_t it = 0; it < ITS; ++it) {
      const uint32_t q0 = (it * kWaveBlock + lane) * V;
      uint32_t r, bits;
      vector_ranks<V>(bits32, s_pref, q0, r, bits);
      Vec16<ES> o;
      o.q = make_u32x4(0, 0, 0, 0);
      if (__ballot(bits != 0) == 0) {
#pragma unroll
        for (uint32_t j = 0; j < V; ++j) vec_set<ES>(o, j, fill);
      } else {
#pragma unroll
        for (uint32_t j = 0; j < V; ++j) {
          const uint32_t rj = r + (uint32_t)__popc(bits & ((1u << j) - 1u));
          const T v = vals.at(rj < last ? rj : last);
          vec_set<ES>(o, j, pick<T>(bits, j, v, fill));
        }
      }
      store_vec_padded<ES>(dst, q0, tile_rows, o);
    }
  });
}

// Nullable<bool>: scatter a bitmap of values by the selection bitmap.
// Tile = 16384 rows; each wave resolves one output word per pass (lane = bit).
__device__ __forceinline__ void body_select_bits(const KArgs& a, const NodeDesc& nd, uint32_t tile, uint8_t* smem) {
  constexpr uint32_t W = kBitTileWords;
  uint64_t* s_bits = reinterpret_cast<uint64_t*>(smem);
  uint32_t* s_pref = reinterpret_cast<uint32_t*>(smem + 2048);
  uint32_t* tmp = reinterpret_cast<uint32_t*>(smem + 3072);
  const uint64_t rows = nd.rows;
  const uint32_t nvals = nd.c1.rows;
  uint64_t* dst = reinterpret_cast<uint64_t*>(a.work + nd.self.off);
  if (tile == 0 && threadIdx.x == 0 && sload32(a.totals + nd.total_idx) != nvals) atomicOr(a.status, ST_SELECT_COUNT);
  const uint64_t w = (uint64_t)tile * W + threadIdx.x;
  uint64_t word = 0;
  if (w * 64 < rows) {
    word = BitsReader(nd.sel, a, false).word(w);
    if (nd.flags & NF_WRITE_NULLS) reinterpret_cast<uint64_t*>(a.out + nd.aux_off)[w] = word;
  }
  uint32_t total;
  const uint32_t ex = block_excl_scan((uint32_t)__popcll(word), tmp, total);
  s_bits[threadIdx.x] = word; s_pref[threadIdx.x] = ex;
  __syncthreads();
  const uint32_t base = sload32(a.parts + nd.part_base + tile);
  for (uint32_t i = wave_id(); i < W; i += 4) {
    const uint64_t ww = (uint64_t)tile * W + i;
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
// The planner hands the alphabet over as a plain element array (Trivial payload
// in the arena, or materialised into work), so the gather is one load per row;
// all index loads of a batch are issued first, then all gathers, then the stores.
template <int ES>
__device__ __forceinline__ void body_dict_gather(const KArgs& a, const NodeDesc& nd, uint32_t tile, uint8_t* smem) {
  using T = typename ElemT<ES>::type;
  constexpr uint32_t V = 16 / ES;
  constexpr uint32_t TE = kTileBytes / ES;
  const uint64_t rows = nd.rows;
  const uint64_t start = (uint64_t)tile * TE;
  const uint32_t tile_rows = rows - start < TE ? (uint32_t)(rows - start) : TE;
  const uint32_t A = nd.c1.rows;
  T* dst = reinterpret_cast<T*>(self_ptr(nd, a)) + start;
  if (A == 0) {  // malformed: rows without an alphabet
    for (uint32_t it = 0; it < 4; ++it) {
      const uint32_t q0 = (it * kBlock + threadIdx.x) * V;
      if (q0 >= tile_rows) break;
      Vec16<ES> x; x.q = make_u32x4(0, 0, 0, 0);
      store_vec<ES>(dst, q0, tile_rows, x);
    }
    if (threadIdx.x == 0) atomicOr(a.status, ST_DICT_INDEX_RANGE);
    return;
  }
  const Reader<RK_PTR, ES> alpha(nd.c1, a);
  // A small alphabet (<= 16 KiB) is copied to LDS once per tile with coalesced
  // loads; the per-row gathers then stay on chip.  Larger ones are gathered from
  // global memory (L2 resident up to a few MiB).
  const bool staged = (uint64_t)A * ES <= kTileBytes;
  T* s_alpha = reinterpret_cast<T*>(smem);
  if (staged) {
    constexpr uint32_t SB = ES >= 4 ? 8 : 2;  // loads in flight per thread (register budget of the narrow types)
    for (uint32_t j0 = threadIdx.x; j0 < A; j0 += SB * kBlock) {
      T t[SB];
#pragma unroll
      for (uint32_t u = 0; u < SB; ++u) {
        const uint32_t j = j0 + u * kBlock;
        t[u] = alpha.get(j < A ? j : A - 1);
      }
#pragma unroll
      for (uint32_t u = 0; u < SB; ++u) {
        const uint32_t j = j0 + u * kBlock;
        if (j < A) s_alpha[j] = t[u];
      }
    }
  }
  bool bad = false;
  with_reader<4>(nd.c0, a, [&](auto rd) {
    const auto idx = rd.view(start);
    if constexpr (ES == 8) {
      // every index of the tile (8 per thread) is requested before the alphabet copy is waited for
      uint32_t ix[4][V];
#pragma unroll
      for (uint32_t u = 0; u < 4; ++u) {
        const uint32_t q0 = (u * kBlock + threadIdx.x) * V;
#pragma unroll
        for (uint32_t j = 0; j < V; ++j) ix[u][j] = idx.at(q0 + j < tile_rows ? q0 + j : tile_rows - 1);
      }
      if (staged) __syncthreads();
      T x[4][V];
#pragma unroll
      for (uint32_t u = 0; u < 4; ++u)
#pragma unroll
        for (uint32_t j = 0; j < V; ++j)
          if (ix[u][j] >= A) { bad = true; ix[u][j] = A - 1; }
      if (staged) {  // one uniform branch around straight-line gathers
#pragma unroll
        for (uint32_t u = 0; u < 4; ++u)
#pragma unroll
          for (uint32_t j = 0; j < V; ++j) x[u][j] = s_alpha[ix[u][j]];
      } else {
#pragma unroll
        for (uint32_t u = 0; u < 4; ++u)
#pragma unroll
          for (uint32_t j = 0; j < V; ++j) x[u][j] = alpha.get(ix[u][j]);
      }
#pragma unroll
      for (uint32_t u = 0; u < 4; ++u) {
        const uint32_t q0 = (u * kBlock + threadIdx.x) * V;
        if (q0 < tile_rows) {
          Vec16<ES> o;
          o.v[0] = x[u][0]; o.v[1] = x[u][1];
          store_vec<ES>(dst, q0, tile_rows, o);
        }
      }
    } else {
      // narrower rows: groups of at most eight rows -- eight index loads, then eight gathers, in flight
      constexpr uint32_t G = V < 8 ? V : 8;     // rows per group
      constexpr uint32_t GU = 8 / G;            // rounds per group (2 for 4-byte rows, else 1)
      if (staged) __syncthreads();
#pragma unroll
      for (uint32_t ib = 0; ib < 4; ib += GU) {
        Vec16<ES> o[GU];
#pragma unroll
        for (uint32_t u = 0; u < GU; ++u) o[u].q = make_u32x4(0, 0, 0, 0);
#pragma unroll
        for (uint32_t g = 0; g < V; g += G) {
          uint32_t ix[GU][G];
#pragma unroll
          for (uint32_t u = 0; u < GU; ++u) {
            const uint32_t q0 = ((ib + u) * kBlock + threadIdx.x) * V + g;
#pragma unroll
            for (uint32_t j = 0; j < G; ++j) ix[u][j] = idx.at(q0 + j < tile_rows ? q0 + j : tile_rows - 1);
          }
#pragma unroll
          for (uint32_t u = 0; u < GU; ++u)
#pragma unroll
            for (uint32_t j = 0; j < G; ++j)
              if (ix[u][j] >= A) { bad = true; ix[u][j] = A - 1; }
          T x[GU][G];
          if (staged) {  // one uniform branch around straight-line gathers
#pragma unroll
            for (uint32_t u = 0; u < GU; ++u)
#pragma unroll
              for (uint32_t j = 0; j < G; ++j) x[u][j] = s_alpha[ix[u][j]];
          } else {
#pragma unroll
            for (uint32_t u = 0; u < GU; ++u)
#pragma unroll
              for (uint32_t j = 0; j < G; ++j) x[u][j] = alpha.get(ix[u][j]);
          }
#pragma unroll
          for (uint32_t u = 0; u < GU; ++u)
#pragma unroll
            for (uint32_t j = 0; j < G; ++j) vec_set<ES>(o[u], g + j, x[u][j]);
        }
#pragma unroll
        for (uint32_t u = 0; u < GU; ++u) {
          const uint32_t q0 = ((ib + u) * kBlock + threadIdx.x) * V;
          if (q0 < tile_rows) store_vec<ES>(dst, q0, tile_rows, o[u]);
        }
      }
    }
  });
  if (bad) atomicOr(a.status, ST_DICT_INDEX_RANGE);
}

// ---------------------------------------------------------------- K_DICT_LDS
// The same gather -- alone (c0 = indices, c1 = alphabet) or inside a rank
// scatter (NF_DICT_VALUES: c0 = alphabet, c1 = indices of the selected rows) --
// for 8-byte rows whose alphabet has more entries than the fused launch can
// stage (2,048) and at most kDictLdsEntries.  From memory every row of such a
// column costs the vector L1 a 128-byte line fill for its 8 bytes; here the
// alphabet is copied to LDS once per workgroup, which then runs kDictLdsTiles
// consecutive tiles of the node against it.  Three workgroups fit a CU, so the
// loads of the next tile (bitmap word, staged indices) are issued before the
// current tile is gathered: two tiles per workgroup are in flight.
constexpr uint32_t kDictLdsAlphaBytes = kDictLdsEntries * 8;
constexpr uint32_t kDictLdsStageBytes = ((15 + (kTileBytes / 8 * 4 + 1) + 4 + 15) >> 4) * 16;  // 2,048 staged indices of up to 32 bits
constexpr uint32_t kDictLdsSmemBytes = kDictLdsAlphaBytes + kDictLdsStageBytes + 32 * 8 + 64 * 2 + 64;
static_assert(kDictLdsSmemBytes * 3 <= 160 * 1024 && kDictLdsStageBytes <= 3 * kBlock * 16, "three workgroups per CU; StageRegs<3> cover the stage");

__device__ __forceinline__ void body_dict_lds(const KArgs& a, const NodeDesc& nd, uint32_t group, uint8_t* smem, uint32_t pslot0) {
  using T = uint64_t;
  constexpr int ES = 8;
  constexpr uint32_t V = 2, TE = kTileBytes / ES, W = TE / 64;
  const bool scatter = (nd.flags & NF_DICT_VALUES) != 0;
  const ColRef& aref = scatter ? nd.c0 : nd.c1;
  const ColRef& iref = scatter ? nd.c1 : nd.c0;
  const uint32_t A = aref.rows;  // 1 .. kDictLdsEntries: the planner's choice
  const uint64_t rows = nd.rows;
  const uint32_t n_tiles = (uint32_t)((rows + TE - 1) / TE);
  const uint32_t t_begin = group * kDictLdsTiles;
  const uint32_t t_end = t_begin + kDictLdsTiles < n_tiles ? t_begin + kDictLdsTiles : n_tiles;
  T* s_alpha = reinterpret_cast<T*>(smem);
  uint8_t* s_stage = smem + kDictLdsAlphaBytes;
  uint64_t* s_bits = reinterpret_cast<uint64_t*>(s_stage + kDictLdsStageBytes);
  uint16_t* s_pref = reinterpret_cast<uint16_t*>(s_bits + W);
  const uint32_t* bits32 = reinterpret_cast<const uint32_t*>(s_bits);
  const bool invert = (nd.flags & NF_INVERT_SEL) != 0;
  const uint32_t nvals = iref.rows;
  const T fill = (T)nd.aux;
  if (scatter && group == 0 && threadIdx.x == 0 && sload32(a.totals + nd.total_idx) != nvals) atomicOr(a.status, ST_SELECT_COUNT);
  {  // the alphabet: every load of a thread in flight together
    const Reader<RK_PTR, ES> alpha(aref, a);
    constexpr uint32_t PER = kDictLdsEntries / kBlock;
    T e[PER];
#pragma unroll
    for (uint32_t u = 0; u < PER; ++u) {
      const uint32_t j = u * kBlock + threadIdx.x;
      e[u] = alpha.get(j < A ? j : A - 1);
    }
#pragma unroll
    for (uint32_t u = 0; u < PER; ++u) {
      const uint32_t j = u * kBlock + threadIdx.x;
      if (j < A) s_alpha[j] = e[u];
    }
  }
  const uint32_t amax = A - 1;
  bool bad = false;
  with_reader<4>(iref, a, [&](auto rd) {
    StageRegs<3> regs;
    uint32_t bit0 = 0, n = 0;
    uint64_t word = 0;
    // requests everything tile t reads from memory: its bitmap word and its indices
    auto fetch = [&](uint32_t t) {
      const uint64_t p0 = (uint64_t)t * TE;
      uint64_t first;
      if (scatter) {
        const uint32_t base = sload32(a.parts + pslot0 + (t - t_begin));
        const uint32_t next = sload32(a.parts + pslot0 + (t - t_begin) + 1);  // the node's total sits behind its last partial
        const uint32_t count = next - base < TE ? next - base : TE;
        const uint32_t avail = base < nvals ? nvals - base : 0u;
        n = count < avail ? count : avail;  // counts that disagree with the values child are clamped, never followed
        first = base;
        const uint64_t w = (uint64_t)t * W + threadIdx.x;
        word = 0;
        if (threadIdx.x < W && w * 64 < rows) word = BitsReader(nd.sel, a, invert).word(w);
      } else {
        n = rows - p0 < TE ? (uint32_t)(rows - p0) : TE;
        first = p0;
      }
      regs.nvec = 0;
      bit0 = 0;
      if (n) bit0 = rd.template stage_begin<3>(first, n, regs);
    };
    fetch(t_begin);
    for (uint32_t t = t_begin; t < t_end; ++t) {  // workgroup uniform
      const uint64_t p0 = (uint64_t)t * TE;
      const uint32_t tile_rows = rows - p0 < TE ? (uint32_t)(rows - p0) : TE;
      T* dst = reinterpret_cast<T*>(self_ptr(nd, a)) + p0;
      const uint32_t cur_n = n;
      const auto idxs = rd.template stage_end<3>(regs, bit0, s_stage);
      if (scatter) {
        const uint64_t w = (uint64_t)t * W + threadIdx.x;
        if ((nd.flags & NF_WRITE_NULLS) && threadIdx.x < W && w * 64 < rows)
          reinterpret_cast<uint64_t*>(a.out + nd.aux_off)[w] = invert ? ~word & (w * 64 + 64 > rows ? wmask((uint32_t)(rows - w * 64)) : ~0ull) : word;
        tile_rank_scan32<W>(word, s_bits, s_pref, nullptr);
      }
      __syncthreads();
      if (t + 1 < t_end) fetch(t + 1);  // in flight while this tile is gathered
      const uint32_t last = cur_n ? cur_n - 1 : 0u;
#pragma unroll
      for (uint32_t it = 0; it < 4; ++it) {
        const uint32_t q0 = (it * kBlock + threadIdx.x) * V;
        uint32_t r = q0, sel = (1u << V) - 1u;
        if (scatter) vector_ranks<V>(bits32, s_pref, q0, r, sel);
        if (cur_n == 0) sel = 0;  // nothing staged: every row takes the fill
        Vec16<ES> o;
#pragma unroll
        for (uint32_t j = 0; j < V; ++j) {
          const uint32_t rj = scatter ? r + (uint32_t)__popc(sel & ((1u << j) - 1u)) : q0 + j;
          uint32_t i = idxs.at(rj < last ? rj : last);
          if (i > amax) {
            bad |= ((sel >> j) & 1u) != 0 && q0 + j < tile_rows;
            i = amax;
          }
          o.v[j] = pick<T>(sel, j, s_alpha[i], fill);
        }
        store_vec_padded<ES>(dst, q0, tile_rows, o);
      }
      __syncthreads();  // the stage and the bitmap words are the next tile's
    }
  });
  if (bad) atomicOr(a.status, ST_DICT_INDEX_RANGE);
}

// ---------------------------------------------------------------- K_DICT_STR
// The gather of K_DICT_LDS for string alphabets of up to kDictStrEntries entries:
// what sits in LDS is not the 8-byte views but the compact table their
// K_STRING_VIEWS step wrote beside them (two bytes per entry, layout in
// nbl_desc.h), so 65,536 entries take 130 KiB instead of 512.  From memory every
// row of such a column costs the vector L1 a 128-byte line fill (the alphabet
// is far larger than the cache), which bounds the column at a third of the rate
// its bytes could move at.  A view is put together from one pair of aligned
// dwords (the entry's slot and the next one) and the base of its block.
//
// The table leaves room for one workgroup per CU, so that workgroup is sixteen
// waves which, once the table is in place, never meet again: a wave owns two of
// the group's kDictStrTiles tiles and works through them in slices of 512 rows
// with LDS of its own (the tile's bitmap words and their rank prefixes, the
// slice's packed indices), nothing but its own memory accesses to wait for, and
// the loads of two slices ahead in flight.  Indices are bit-packed, 1 .. 16 bits
// (the planner's condition): a slice's are at most 1 KiB, 16 bytes per lane.
constexpr uint32_t kDictStrBlock = 1024, kDictStrWaves = kDictStrBlock / 64;
constexpr uint32_t kDictStrSliceRows = 512;
constexpr uint32_t kDictStrTableBytes = dict_str_table_bytes(kDictStrEntries);
constexpr uint32_t kDictStrPartsBytes = ((kDictStrTiles + 1) * 4 + 15) & ~15u;
constexpr uint32_t kDictStrWaveStage = kDictStrSliceRows * 2 + 32;                    // packed indices of a slice
constexpr uint32_t kDictStrWaveBytes = kDictStrWaveStage + 64 * 8 + 68 * 4;           // + bitmap words and rank prefixes of the wave's two tiles
constexpr uint32_t kDictStrSmemBytes = kDictStrTableBytes + kDictStrPartsBytes + kDictStrWaves * kDictStrWaveBytes;
static_assert(kDictStrSmemBytes <= 160 * 1024 && kDictStrTableBytes % 16 == 0 && kDictStrWaveBytes % 16 == 0, "one workgroup per CU");
static_assert(kDictStrTiles == 2 * kDictStrWaves && kTileBytes / 8 == 4 * kDictStrSliceRows, "two tiles per wave, four slices per tile");

__device__ __forceinline__ void body_dict_str(const KArgs& a, const NodeDesc& nd, uint32_t group, uint8_t* smem, uint32_t pslot0) {
  using T = uint64_t;
  constexpr int ES = 8;
  constexpr uint32_t TE = kTileBytes / ES, W = TE / 64;  // 2,048 rows, 32 bitmap words per tile
  const bool scatter = (nd.flags & NF_DICT_VALUES) != 0;
  const ColRef& aref = scatter ? nd.c0 : nd.c1;
  const ColRef& iref = scatter ? nd.c1 : nd.c0;
  const uint32_t A = aref.rows;  // 1 .. kDictStrEntries: the planner's choice
  const uint64_t rows = nd.rows;
  const uint32_t n_tiles = (uint32_t)((rows + TE - 1) / TE);
  const uint32_t t_begin = group * kDictStrTiles;
  const uint32_t n_t = (t_begin + kDictStrTiles < n_tiles ? t_begin + kDictStrTiles : n_tiles) - t_begin;  // 1 .. 32
  const bool invert = (nd.flags & NF_INVERT_SEL) != 0;
  const uint32_t nvals = iref.rows;
  const T fill = (T)nd.aux;
  const uint32_t lane = lane_id(), wv = wave_id();
  uint32_t* s_parts = reinterpret_cast<uint32_t*>(smem + kDictStrTableBytes);
  if (scatter && group == 0 && threadIdx.x == 0 && sload32(a.totals + nd.total_idx) != nvals) atomicOr(a.status, ST_SELECT_COUNT);
  // The bitmap words of the wave's two tiles (lanes 0 .. 31: tile wv, 32 .. 63:
  // tile wv + 16) are requested before the table, so they arrive under its copy.
  const uint32_t half = lane >> 5, tl = wv + kDictStrWaves * half;
  const uint32_t my_t = t_begin + tl;
  const uint64_t my_w = (uint64_t)my_t * W + (lane & 31u);
  uint64_t word = 0;
  if (tl < n_t && my_w * 64 < rows) {
    if (scatter) word = BitsReader(nd.sel, a, invert).word(my_w);
    else word = my_w * 64 + 64 > rows ? wmask((uint32_t)(rows - my_w * 64)) : ~0ull;  // every row takes the index of its own number
  }
  // so are the group's rank partials, one per tile (the node's total sits behind its last one)
  uint32_t part = 0;
  if (scatter && threadIdx.x < n_t) part = a.parts[pslot0 + threadIdx.x];
  {  // the table: a straight copy, eight loads of a thread in flight together
    const u32x4* src = reinterpret_cast<const u32x4*>(a.work + nd.aux3);
    u32x4* dst = reinterpret_cast<u32x4*>(smem);
    const uint32_t nvec = dict_str_table_bytes(A) >> 4;
    constexpr uint32_t U = 8;
    for (uint32_t v0 = 0; v0 < nvec; v0 += U * kDictStrBlock) {  // workgroup uniform
      u32x4 q[U];
#pragma unroll
      for (uint32_t u = 0; u < U; ++u) {
        const uint32_t i = v0 + u * kDictStrBlock + threadIdx.x;
        q[u] = src[i < nvec ? i : nvec - 1];
      }
#pragma unroll
      for (uint32_t u = 0; u < U; ++u) {
        const uint32_t i = v0 + u * kDictStrBlock + threadIdx.x;
        if (i < nvec) dst[i] = q[u];
      }
    }
    if (scatter && threadIdx.x < n_t) s_parts[threadIdx.x] = part;
  }
  __syncthreads();  // the only one
  const uint32_t* s_coarse = reinterpret_cast<const uint32_t*>(smem);
  const uint32_t* s_slots = reinterpret_cast<const uint32_t*>(smem + dict_str_slots_off(A));  // two 16-bit slots per dword
  auto view_of = [&](uint32_t i) -> T {
    const uint32_t k = i + (i >> 8);
    const uint32_t pair = __builtin_amdgcn_alignbit(s_slots[(k >> 1) + 1], s_slots[k >> 1], (k & 1u) * 16u);  // slots k, k + 1
    const uint32_t lo = pair & 0xFFFFu;
    return (T)(s_coarse[i >> 8] + lo) | ((T)((pair >> 16) - lo) << 32);
  };
  // ---- from here on every wave is on its own
  uint8_t* mine = smem + kDictStrTableBytes + kDictStrPartsBytes + wv * kDictStrWaveBytes;
  const uint32_t* s_stage = reinterpret_cast<const uint32_t*>(mine);
  uint64_t* s_words = reinterpret_cast<uint64_t*>(mine + kDictStrWaveStage);  // words of tile wv (0 .. 31) and of tile wv + 16 (32 .. 63)
  uint32_t* s_pref = reinterpret_cast<uint32_t*>(mine + kDictStrWaveStage + 64 * 8);  // rank of each word's first row; [64], [65]: ranks behind the tiles
  {
    const uint32_t pc = (uint32_t)__popcll(word);
    const uint32_t incl = wave_incl_scan(pc);
    const uint32_t low = (uint32_t)__builtin_amdgcn_readlane((int)incl, 31);  // rows selected in the first of the two tiles
    uint32_t first = 0;  // rank of the tile's first row
    if (tl < n_t) first = scatter ? s_parts[tl] : my_t * TE;
    const uint32_t excl = first + incl - pc - (half ? low : 0u);
    s_words[lane] = word;
    s_pref[lane] = excl;
    if ((lane & 31u) == 31u) s_pref[64 + half] = excl + pc;
  }
  __builtin_amdgcn_wave_barrier();  // (the LDS operations of a wave complete in order; this keeps the compiler from moving them)
  const uint8_t* ibase = a.arena + iref.off;
  const uint32_t iw = iref.width, imask = (1u << iw) - 1u, ibias = (uint32_t)iref.base;
  const uint32_t amax = A - 1;
  bool bad = false;
  struct Slice {
    uint32_t rb, cnt, sub;  // first rank, ranks (clamped to what the index column holds), bit offset of the first index in its byte
  };
  auto slice_of = [&](uint32_t sl) {  // sl = 0 .. 7: four slices of the wave's first tile, then of its second
    Slice x{0, 0, 0};
    const uint32_t half = sl >> 2, m = sl & 3u;
    if (sl < 8 && wv + kDictStrWaves * half < n_t) {
      const uint32_t rb = s_pref[half * 32 + m * 8];
      const uint32_t re = m == 3 ? s_pref[64 + half] : s_pref[half * 32 + m * 8 + 8];
      const uint32_t avail = rb < nvals ? nvals - rb : 0u;  // ranks that disagree with the index column are clamped, never followed
      x.rb = rb;
      x.cnt = re - rb < avail ? re - rb : avail;
      x.sub = (uint32_t)(((uint64_t)rb * iw) & 7u);
    }
    return x;
  };
  auto issue = [&](const Slice& x) {  // the slice's packed indices, 16 bytes per lane
    u32x4 q = {0u, 0u, 0u, 0u};
    const uint32_t bytes = x.cnt ? (x.cnt * iw + x.sub + 7u) >> 3 : 0u;  // <= 1,024; the arena's slop covers the last vector's overhang
    if (lane * 16 < bytes) __builtin_memcpy(&q, ibase + (((uint64_t)x.rb * iw) >> 3) + lane * 16, 16);
    return q;
  };
  Slice x0 = slice_of(0), x1 = slice_of(1);
  u32x4 q0 = issue(x0), q1 = issue(x1);
  // the copy of the bitmap goes out behind the first loads: a wave's loads return
  // in order with its stores
  if (scatter && (nd.flags & NF_WRITE_NULLS) && tl < n_t && my_w * 64 < rows)
    reinterpret_cast<uint64_t*>(a.out + nd.aux_off)[my_w] =
        invert ? ~word & (my_w * 64 + 64 > rows ? wmask((uint32_t)(rows - my_w * 64)) : ~0ull) : word;
  for (uint32_t sl = 0; sl < 8; ++sl) {  // wave uniform
    const uint32_t half = sl >> 2, m = sl & 3u, tl = wv + kDictStrWaves * half;
    if (tl >= n_t) break;
    const Slice x2 = slice_of(sl + 2);
    const u32x4 q2 = issue(x2);
    __builtin_amdgcn_wave_barrier();
    *reinterpret_cast<u32x4*>(mine + lane * 16) = q0;
    __builtin_amdgcn_wave_barrier();
    const uint32_t t = t_begin + tl;
    const uint64_t p0 = (uint64_t)t * TE;
    const uint32_t tile_rows = rows - p0 < TE ? (uint32_t)(rows - p0) : TE;
    T* dst = reinterpret_cast<T*>(self_ptr(nd, a)) + p0;
    const uint32_t last = x0.cnt ? x0.cnt - 1 : 0u;
#pragma unroll
    for (uint32_t it = 0; it < 4; ++it) {
      const uint32_t r = m * kDictStrSliceRows + it * 128 + lane * 2;  // row of the tile; r + 1 shares its bitmap word
      const uint32_t wi = half * 32 + (r >> 6), bit = r & 63u;
      const uint64_t wd = s_words[wi];
      uint32_t sel = (uint32_t)(wd >> bit) & 3u;
      const uint32_t k0 = s_pref[wi] + (uint32_t)__popcll(wd & ((1ull << bit) - 1ull)) - x0.rb;  // index of row r among the slice's, if selected
      if (x0.cnt == 0) sel = 0;  // nothing staged: every row takes the fill
      Vec16<ES> o;
#pragma unroll
      for (uint32_t j = 0; j < 2; ++j) {
        const uint32_t k = k0 + (j ? sel & 1u : 0u);
        const uint32_t b = x0.sub + (k < last ? k : last) * iw;
        const uint32_t* d = s_stage + (b >> 5);
        uint32_t i = (__builtin_amdgcn_alignbit(d[1], d[0], b) & imask) + ibias;
        if (i > amax) {
          bad |= ((sel >> j) & 1u) != 0 && r + j < tile_rows;
          i = amax;
        }
        o.v[j] = pick<T>(sel, j, view_of(i), fill);
      }
      store_vec_padded<ES>(dst, r, tile_rows, o);
    }
    x0 = x1; x1 = x2;
    q0 = q1; q1 = q2;
  }
  if (bad) atomicOr(a.status, ST_DICT_INDEX_RANGE);
}

// ----------------------------------------------------------- K_VARINT_DECODE
// VarintEncoding::materialize (encodings/VarintEncoding.h:119-140): LEB128,
// low 7 bits first, + baseline (common/Varint.h:83-143 for the byte limits).
// The tile's payload bytes (4 KiB + 9 bytes of look-ahead) are staged in LDS.
// Pass 1: every thread finds the varint starts among its 16 bytes and appends
// their offsets to a list in LDS at their rank.  Pass 2: lane i decodes varint
// i of the tile from LDS, so the work is balanced whatever the lengths are and
// the stores of a wave are consecutive.
template <int ES>
__device__ __forceinline__ void body_varint_decode(const KArgs& a, const NodeDesc& nd, uint32_t, uint8_t* smem) {
  using T = typename ElemT<ES>::type;
  const uint32_t n_tiles = nd.n_parts;
  const uint32_t tile = chain_ticket(a, nd, n_tiles, smem);
  u32x4* stage = reinterpret_cast<u32x4*>(smem);                        // <= 16 + 4096 + 10 + 15 + 4 bytes
  uint16_t* s_start = reinterpret_cast<uint16_t*>(smem + 8192);          // <= 4096 offsets
  uint32_t* tmp = reinterpret_cast<uint32_t*>(smem + kStageBytes + 3072);
  const uint8_t* payload = a.arena + nd.c0.off;
  const uint64_t len = nd.aux2;
  const uint64_t rows = nd.rows;
  const T base = (T)nd.c0.base;
  T* dst = reinterpret_cast<T*>(self_ptr(nd, a));
  const uint64_t t0 = (uint64_t)tile * kVarintTile;                      // < len: the tile exists
  const uint32_t tile_len = len - t0 < kVarintTile ? (uint32_t)(len - t0) : kVarintTile;
  const uint32_t lead = tile ? 16u : 0u;                                 // bytes staged ahead of the tile: the previous byte decides whether byte 0 starts a value
  const uint32_t mis = stage_tile<2>(payload + t0 - lead, lead + tile_len + 10u, stage);
  __syncthreads();
  const uint8_t* s8 = reinterpret_cast<const uint8_t*>(stage);
  const uint32_t* s32 = reinterpret_cast<const uint32_t*>(stage);
  const uint32_t org = mis + lead;                                       // LDS byte of the tile's first payload byte
  // pass 1: start mask of this thread's 16 bytes
  const uint32_t o0 = threadIdx.x * 16;
  uint32_t m = 0;
  if (o0 < tile_len) {
    uint32_t n0, n1, n2, n3;
    lds_window(s32, (org + o0) * 8u, n0, n1, n2, n3);
    auto high_bits = [](uint32_t d) { const uint32_t t = d & 0x80808080u; return ((t >> 7) | (t >> 14) | (t >> 21) | (t >> 28)) & 0xFu; };
    const uint32_t hb = high_bits(n0) | (high_bits(n1) << 4) | (high_bits(n2) << 8) | (high_bits(n3) << 12);
    const uint32_t prev = (tile | threadIdx.x) ? (uint32_t)(s8[org + o0 - 1] >> 7) : 0u;  // the very first byte always starts a value
    const uint32_t valid = tile_len - o0 >= 16 ? 0xFFFFu : ((1u << (tile_len - o0)) - 1u);
    m = ~((hb << 1) | prev) & valid;
  }
  uint32_t count;
  uint32_t k = block_excl_scan((uint32_t)__popc(m), tmp, count);
  while (m) {
    const uint32_t j = (uint32_t)__ffs((int)m) - 1u;
    m &= m - 1;
    s_start[k++] = (uint16_t)(o0 + j);
  }
  __syncthreads();
  // pass 2
  // the ordinal of the tile's first value: a chained scan of the tiles' counts
  uint32_t unused;
  const uint64_t rank0 = chain_exclusive(a, nd, tile, n_tiles, count, 0u, unused, smem);
  // the last tile knows how many values the payload holds
  if (tile + 1 == n_tiles && threadIdx.x == 0 && rank0 + count < rows) atomicOr(a.status, ST_VARINT_COUNT);
  for (uint32_t i = threadIdx.x; i < count; i += kBlock) {
    if (rank0 + i >= rows) break;
    const uint32_t off = s_start[i];
    const uint64_t avail = len - (t0 + off);                             // payload bytes from this start on (>= 1)
    const uint32_t L = org + off;
    const uint32_t d0 = L >> 2, S = (L & 3u) * 8u;
    const uint32_t w0 = s32[d0], w1 = s32[d0 + 1], w2 = s32[d0 + 2], w3 = s32[d0 + 3];
    uint64_t lo = (uint64_t)__builtin_amdgcn_alignbit(w1, w0, S) | ((uint64_t)__builtin_amdgcn_alignbit(w2, w1, S) << 32);  // bytes 0..7
    uint32_t hi = __builtin_amdgcn_alignbit(w3, w2, S);                                                                          // bytes 8..11
    if (avail < 8) lo &= (1ull << (8 * (uint32_t)avail)) - 1ull;         // a truncated payload reads as zero bytes
    if (avail < 10) hi &= avail <= 8 ? 0u : 0xFFu;
    // the value ends at the first byte whose high bit is clear
    const uint64_t stop = ~lo & 0x8080808080808080ull;
    uint64_t x = lo & 0x7F7F7F7F7F7F7F7Full;                             // 8 groups of 7 bits ...
    x = ((x & 0x7F007F007F007F00ull) >> 1) | (x & 0x007F007F007F007Full);
    x = ((x & 0x3FFF00003FFF0000ull) >> 2) | (x & 0x00003FFF00003FFFull);
    x = ((x & 0x0FFFFFFF00000000ull) >> 4) | (x & 0x000000000FFFFFFFull);  // ... packed into 56 bits
    uint64_t v;
    if (stop) {
      const uint32_t nbytes = ((uint32_t)__ffsll((unsigned long long)stop) >> 3);  // 1..8
      v = nbytes == 8 ? x : (x & ((1ull << (7 * nbytes)) - 1ull));
    } else {
      v = x | ((uint64_t)(hi & 0x7Fu) << 56);
      if (hi & 0x80u) v |= (uint64_t)((hi >> 8) & 0x7Fu) << 63;
    }
    dst[rank0 + i] = (T)((T)v + base);
  }
}

// ------------------------------------------------------------ K_STRING_VIEWS
// TrivialEncoding<string_view>::materialize (encodings/TrivialEncoding.cpp:151-164):
// view i = {running sum of lengths, length i}; emitted as uint64 = len << 32 | offset,
// offset relative to the chunk's string region.
__device__ __forceinline__ void body_string_views(const KArgs& a, const NodeDesc& nd, uint32_t, uint8_t* smem) {
  uint32_t* tmp = reinterpret_cast<uint32_t*>(smem + kScratchOff);
  const uint64_t n = nd.rows;
  const uint32_t n_tiles = nd.n_parts;
  const uint32_t tile = chain_ticket(a, nd, n_tiles, smem);  // the offsets are a chained scan over the node's tiles
  const uint64_t e0 = (uint64_t)tile * kScanTile + (uint64_t)threadIdx.x * 8;
  uint32_t len[8];
  uint32_t s = 0;
  {
    const uint64_t t0 = (uint64_t)tile * kScanTile;
    const uint32_t tn = n - t0 < kScanTile ? (uint32_t)(n - t0) : kScanTile;
    const uint32_t r0 = threadIdx.x * 8;
    with_reader<4>(nd.c0, a, [&](auto rd) {
      const auto src = rd.view(t0);
#pragma unroll
      for (uint32_t j = 0; j < 8; ++j) len[j] = src.at(r0 + j < tn ? r0 + j : tn - 1);
    });
  }
#pragma unroll
  for (uint32_t j = 0; j < 8; ++j) {
    len[j] = e0 + j < n ? len[j] : 0;
    s += len[j];
  }
  uint32_t total;
  uint32_t off = block_excl_scan(s, tmp, total);
  uint32_t unused;
  const uint32_t before = chain_exclusive(a, nd, tile, n_tiles, total, 0u, unused, smem);
  off += before + (uint32_t)nd.aux;
  uint64_t* dst = reinterpret_cast<uint64_t*>(self_ptr(nd, a));
  uint64_t v[8];
#pragma unroll
  for (uint32_t j = 0; j < 8; ++j) {
    v[j] = ((uint64_t)len[j] << 32) | off;
    off += len[j];
  }
  if (e0 + 8 <= n) {  // 64 contiguous bytes per thread
#pragma unroll
    for (uint32_t j = 0; j < 8; j += 2)
      *reinterpret_cast<u32x4*>(dst + e0 + j) = make_u32x4((uint32_t)v[j], (uint32_t)(v[j] >> 32), (uint32_t)v[j + 1], (uint32_t)(v[j + 1] >> 32));
  } else {
#pragma unroll
    for (uint32_t j = 0; j < 8; ++j)
      if (e0 + j < n) dst[e0 + j] = v[j];
  }
  // the last tile knows the sum of all lengths
  if (tile + 1 == n_tiles && threadIdx.x == 0 && (uint64_t)before + total > nd.aux2) atomicOr(a.status, ST_STRING_BYTES);
  if (nd.flags & NF_COMPACT_VIEWS) {
    // The same views in the compact form a K_DICT_STR gather keeps in LDS (layout:
    // nbl_desc.h).  A thread's eight entries lie in one block of 256, which is the
    // entries of 32 consecutive threads: the block's base is the offset of the
    // first of them.
    const uint32_t base = (uint32_t)__shfl((int)(uint32_t)v[0], (int)(lane_id() & 32u));
    uint8_t* table = a.work + nd.aux_off;
    uint32_t* coarse = reinterpret_cast<uint32_t*>(table);
    uint16_t* slots = reinterpret_cast<uint16_t*>(table + dict_str_slots_off((uint32_t)n));
    if (e0 < n) {
      const uint32_t k0 = (uint32_t)e0 + (uint32_t)(e0 >> 8);
#pragma unroll
      for (uint32_t j = 0; j < 8; ++j)
        if (e0 + j < n) slots[k0 + j] = (uint16_t)((uint32_t)v[j] - base);
      // the slot behind a block's last entry holds where that entry ends (`off` by now)
      if ((threadIdx.x & 31u) == 31u || e0 + 8 >= n) slots[k0 + (e0 + 8 <= n ? 8u : (uint32_t)(n - e0))] = (uint16_t)(off - base);
      if ((threadIdx.x & 31u) == 0) coarse[e0 >> 8] = base;
    }
  }
}

// -------------------------------------------------------------- K_COPY_BYTES
// The blob copy TrivialEncoding<string_view> does into stringBufferFactory
// memory (encodings/TrivialEncoding.cpp:129-135).
__device__ __forceinline__ void body_copy_bytes(const KArgs& a, const NodeDesc& nd, uint32_t tile, uint8_t* smem) {
  const uint64_t n = nd.aux;
  const uint64_t start = (uint64_t)tile * kTileBytes;
  const uint8_t* src = a.arena + nd.aux_off + start;
  uint8_t* dst = a.out + nd.self.off + start;       // string regions start 16-byte aligned
  const uint32_t tile_bytes = n - start < kTileBytes ? (uint32_t)(n - start) : kTileBytes;
  u32x4 q[4];
  if ((reinterpret_cast<uintptr_t>(src) & 15u) == 0) {
#pragma unroll
    for (uint32_t it = 0; it < 4; ++it) {
      const uint32_t b = (it * kBlock + threadIdx.x) * 16;
      q[it] = ld16_stream(reinterpret_cast<const u32x4*>(b + 16 <= tile_bytes ? src + b : src));  // arena slop keeps a short tile readable
    }
  } else {
    u32x4* stage = reinterpret_cast<u32x4*>(smem);
    const uint32_t mis = stage_tile<5>(src, tile_bytes, stage);
    __syncthreads();
    const uint32_t* s32 = reinterpret_cast<const uint32_t*>(stage);
#pragma unroll
    for (uint32_t it = 0; it < 4; ++it) {
      const uint32_t b = (it * kBlock + threadIdx.x) * 16;
      uint32_t n0, n1, n2, n3;
      lds_window(s32, (mis + (b + 16 <= tile_bytes ? b : 0)) * 8u, n0, n1, n2, n3);
      q[it] = make_u32x4(n0, n1, n2, n3);
    }
  }
#pragma unroll
  for (uint32_t it = 0; it < 4; ++it) {
    const uint32_t b = (it * kBlock + threadIdx.x) * 16;
    if (b + 16 <= tile_bytes) {
      __builtin_nontemporal_store(q[it], reinterpret_cast<u32x4*>(dst + b));
    } else if (b < tile_bytes) {
#pragma unroll 1
      for (uint32_t i = b; i < tile_bytes; ++i) dst[i] = src[i];
    }
  }
}

// ------------------------------------------------------------ K_FRAME_UNPACK
// BlockBitPacking (encodings/BlockBitPackingEncoding.h:746-784) and FOR
// (encodings/ForEncoding.h:398-529): the rows are cut into segments (blocks /
// frames) of aux2 rows, the first one possibly shorter; segment s has its own
// baseline c0[s], bit width c1[s] and start sel[s] in the packed payload (byte
// offset for blocks, bit offset for frames).  Inside a segment the values are a
// plain LSB-first bit stream.  The three per-segment arrays are element arrays
// in work.
//
// A tile first reads the parameters of the segments it touches (one thread per
// segment) into LDS and validates their bit ranges, then stages the payload
// bytes between the lowest and the highest bit its rows need with aligned
// 16-byte loads; every lane then unpacks 16 bytes of consecutive rows per round
// from one LDS window, as the FixedBitWidth kernel does.  Vectors that straddle
// two segments, segments whose range or width is not acceptable, and tiles
// whose segments are too many or too scattered for LDS are decoded row by row
// from global memory (frame_row), which is also where inconsistencies are
// reported.
__device__ __forceinline__ uint64_t bits_get(const uint8_t* payload, uint64_t bit, uint32_t w) {
  const uint8_t* p = payload + (bit >> 3);
  const uint32_t sh = (uint32_t)bit & 7u;
  uint64_t v = ldu64(p) >> sh;
  if (w + sh > 64) v |= (uint64_t)p[8] << (64 - sh);
  return v & wmask(w);
}
struct FrameGeom {
  uint32_t seg_size, seg_first, shift;
  bool pow2;
  __device__ __forceinline__ uint32_t seg_of(uint32_t row) const {
    if (row < seg_first) return 0u;
    const uint32_t d = row - seg_first;
    return 1u + (pow2 ? d >> shift : d / seg_size);
  }
  __device__ __forceinline__ uint32_t row0(uint32_t s) const { return s ? seg_first + (s - 1u) * seg_size : 0u; }
  __device__ __forceinline__ uint32_t len(uint32_t s) const { return s ? seg_size : seg_first; }
};
template <int ES>
__device__ __forceinline__ typename ElemT<ES>::type frame_row(const KArgs& a, const NodeDesc& nd, const FrameGeom& g, uint32_t row,
                                                              uint32_t& bad) {
  using T = typename ElemT<ES>::type;
  const bool frames = (nd.flags & NF_FRAME_BITS) != 0;
  const uint8_t* payload = a.arena + nd.aux_off;
  const uint64_t payload_bits = nd.aux * 8;
  const uint32_t s = g.seg_of(row);
  const uint32_t r = row - g.row0(s);
  const uint32_t w = (a.work + nd.c1.off)[s];
  const T base = reinterpret_cast<const T*>(a.work + nd.c0.off)[s];
  const uint8_t* offs = a.work + nd.sel.off;
  const uint64_t bit0 = frames ? reinterpret_cast<const uint64_t*>(offs)[s] : (uint64_t)reinterpret_cast<const uint32_t*>(offs)[s] * 8u;
  if (!frames && w == 255u) {  // block stored as raw values
    const uint64_t byte = (bit0 >> 3) + (uint64_t)r * ES;
    if ((byte + ES) * 8u <= payload_bits) return ld_elem<ES>(payload + byte);
    bad = 1;
    return 0;
  }
  if (w == 0) return base;
  if (w > 64u || (!frames && w >= 8u * ES)) {
    bad = 1;
    return 0;
  }
  const uint64_t rel = (uint64_t)r * w;
  if (bit0 <= payload_bits && rel <= payload_bits - bit0 && w <= payload_bits - bit0 - rel) return (T)(bits_get(payload, bit0 + rel, w) + base);
  bad = 1;
  return 0;
}
constexpr uint32_t kFrameMaxSegs = 160;  // segments of one tile whose parameters fit behind the staged bytes
constexpr uint32_t kFrameSlowSeg = 0xFFu;
static_assert(kStageBytes % 16 == 0 && kStageBytes + kFrameMaxSegs * 17 + 8 <= kSmemBytes, "segment tables fit behind the stage");
template <int ES>
__device__ __forceinline__ void body_frame_unpack(const KArgs& a, const NodeDesc& nd, uint32_t tile, uint8_t* smem) {
  using T = typename ElemT<ES>::type;
  constexpr uint32_t V = 16 / ES;
  constexpr uint32_t TE = kTileBytes / ES;
  constexpr uint32_t WFULL = 8 * ES;
  const uint32_t rows = nd.rows;
  const uint32_t start = tile * TE;
  const uint32_t tile_rows = rows - start < TE ? rows - start : TE;
  T* dst = reinterpret_cast<T*>(self_ptr(nd, a)) + start;
  const bool frames = (nd.flags & NF_FRAME_BITS) != 0;
  FrameGeom g;
  g.seg_size = (uint32_t)nd.aux2;
  g.seg_first = (uint32_t)(nd.aux2 >> 32);
  g.pow2 = (g.seg_size & (g.seg_size - 1u)) == 0;
  g.shift = (uint32_t)__builtin_ctz(g.seg_size);
  const uint8_t* payload = a.arena + nd.aux_off;
  const uint64_t payload_bits = nd.aux * 8;

  u32x4* stage = reinterpret_cast<u32x4*>(smem);
  uint64_t* s_base = reinterpret_cast<uint64_t*>(smem + kStageBytes);
  uint64_t* s_bit = s_base + kFrameMaxSegs;
  uint8_t* s_w = reinterpret_cast<uint8_t*>(s_bit + kFrameMaxSegs);
  uint32_t* s_range = reinterpret_cast<uint32_t*>(s_w + kFrameMaxSegs);  // lowest / highest payload byte needed

  const uint32_t s_lo = g.seg_of(start);
  const uint32_t nseg = g.seg_of(start + tile_rows - 1u) - s_lo + 1u;
  bool tile_fast = nseg <= kFrameMaxSegs;
  if (threadIdx.x == 0) {
    s_range[0] = ~0u;
    s_range[1] = 0u;
  }
  __syncthreads();
  if (tile_fast && threadIdx.x < nseg) {
    const uint32_t s = s_lo + threadIdx.x;
    const uint32_t w = (a.work + nd.c1.off)[s];
    T base = reinterpret_cast<const T*>(a.work + nd.c0.off)[s];
    const uint8_t* offs = a.work + nd.sel.off;
    uint64_t off = frames ? reinterpret_cast<const uint64_t*>(offs)[s] : (uint64_t)reinterpret_cast<const uint32_t*>(offs)[s] * 8u;
    const uint32_t seg_row0 = g.row0(s);
    const uint32_t r_lo = start > seg_row0 ? start - seg_row0 : 0u;          // rows of the segment inside the tile
    const uint32_t r_end = start + tile_rows - seg_row0;
    const uint32_t r_hi = r_end < g.len(s) ? r_end : g.len(s);
    const bool raw = !frames && w == 255u;
    const uint32_t weff = raw ? WFULL : w;
    uint32_t code = kFrameSlowSeg;
    if (raw || (frames ? w <= WFULL : w < WFULL)) {
      if (weff == 0) {
        code = 0;
      } else if (off <= payload_bits && (uint64_t)r_hi * weff <= payload_bits - off) {
        code = weff;
        atomicMin(&s_range[0], (uint32_t)((off + (uint64_t)r_lo * weff) >> 3));
        atomicMax(&s_range[1], (uint32_t)((off + (uint64_t)r_hi * weff + 7u) >> 3));
      }
      if (raw) base = 0;
    }
    s_w[threadIdx.x] = (uint8_t)code;
    s_base[threadIdx.x] = base;
    s_bit[threadIdx.x] = off;
  }
  __syncthreads();
  const uint32_t lo = s_range[0], hi = s_range[1];
  uint32_t mis = 0;
  if (tile_fast && lo < hi) {
    if (hi - lo > kStageBytes - 19u) tile_fast = false;  // the needed bytes are spread too far for LDS
    else mis = stage_tile<5>(payload + lo, hi - lo, stage);
  }
  __syncthreads();
  const uint32_t* s32 = reinterpret_cast<const uint32_t*>(stage);

  uint32_t slow_rounds = 0;
#pragma unroll
  for (uint32_t it = 0; it < 4; ++it) {
    const uint32_t q = (it * kBlock + threadIdx.x) * V;
    if (q >= tile_rows) continue;
    bool done = false;
    if (tile_fast) {
      const uint32_t row = start + q;
      const uint32_t s = g.seg_of(row);
      const uint32_t r = row - g.row0(s);
      const uint32_t here = tile_rows - q < V ? tile_rows - q : V;
      const uint32_t code = s_w[s - s_lo];
      if (g.len(s) - r >= here && code != kFrameSlowSeg) {
        const T base = (T)s_base[s - s_lo];
        Vec16<ES> x;
        if (code == 0) {
          unpack_window<ES>(0u, 0u, 0u, 0u, 0u, base, x);
        } else {
          const uint32_t lb = (uint32_t)(s_bit[s - s_lo] - (uint64_t)lo * 8u) + mis * 8u + r * code;
          uint32_t n0, n1, n2, n3;
          lds_window(s32, lb, n0, n1, n2, n3);
          if (code == WFULL) {
            x.q = make_u32x4(n0, n1, n2, n3);
            vec_add_base<ES>(x, base);
          } else {
            unpack_window<ES>(n0, n1, n2, n3, code, base, x);
          }
        }
        store_vec<ES>(dst, q, tile_rows, x);
        done = true;
      }
    }
    if (!done) slow_rounds |= 1u << it;
  }
  uint32_t bad = 0;
#pragma unroll 1
  for (uint32_t it = 0; it < 4; ++it) {
    if (!((slow_rounds >> it) & 1u)) continue;
    const uint32_t q = (it * kBlock + threadIdx.x) * V;
    Vec16<ES> x;
    x.q = make_u32x4(0, 0, 0, 0);
#pragma unroll
    for (uint32_t j = 0; j < V; ++j) {
      T val = 0;
      if (q + j < tile_rows) val = frame_row<ES>(a, nd, g, start + q + j, bad);
      vec_set<ES>(x, j, val);
    }
    store_vec<ES>(dst, q, tile_rows, x);
  }
  if (bad) atomicOr(a.status, ST_FRAME_RANGE);
}

// ================================================================ dispatch
// run_op<OP>: the body of operation OP = make_op(kernel, element size).
template <uint32_t OP>
__device__ __forceinline__ void run_op(const KArgs& a, const NodeDesc& nd, uint32_t tile, uint8_t* smem, uint32_t aux) {
  constexpr uint32_t K = OP / 4;
  constexpr int ES = 1 << (OP % 4);
  if constexpr (K == K_MATERIALIZE) body_materialize<ES>(a, nd, tile, smem);
  else if constexpr (K == K_MATERIALIZE_FBW) body_materialize_fbw<ES>(a, nd, tile, smem);
  else if constexpr (K == K_MATERIALIZE_RAW) body_materialize_raw<ES>(a, nd, tile, smem);
  else if constexpr (K == K_BITS_TO_BYTES) body_bits_to_bytes(a, nd, tile, smem);
  else if constexpr (K == K_REDUCE_POPC) body_reduce_popc(a, nd, tile, smem);
  else if constexpr (K == K_RLE_ENDS) body_rle_ends(a, nd, tile, smem);
  else if constexpr (K == K_RLE_EXPAND) body_rle_expand<ES>(a, nd, tile, smem);
  else if constexpr (K == K_RLE_EXPAND_BOOL) body_rle_expand_bool(a, nd, tile, smem);
  else if constexpr (K == K_SPARSE_BITS) body_sparse_bits(a, nd, tile, smem);
  else if constexpr (K == K_SELECT_SCATTER) body_select_scatter<ES>(a, nd, tile, smem, aux);
  else if constexpr (K == K_SELECT_BITS) body_select_bits(a, nd, tile, smem);
  else if constexpr (K == K_DICT_GATHER) body_dict_gather<ES>(a, nd, tile, smem);
  else if constexpr (K == K_VARINT_DECODE) body_varint_decode<(ES == 4 ? 4 : 8)>(a, nd, tile, smem);
  else if constexpr (K == K_STRING_VIEWS) body_string_views(a, nd, tile, smem);
  else if constexpr (K == K_COPY_BYTES) body_copy_bytes(a, nd, tile, smem);
  else if constexpr (K == K_FRAME_UNPACK) body_frame_unpack<ES>(a, nd, tile, smem);
}

// The descriptor of a tile's node, copied by value ahead of every store of the
// body: the copy is a handful of scalar loads issued together.  Read through a
// reference instead, each field is fetched where it is first used, and once the
// body has stored anything (`work` holds the descriptors too, so the compiler
// must assume they changed) those fetches are vector loads the wave waits for
// one after the other, behind the CU's queued stores.  Copied dword by dword so
// that the byte-sized fields (kind, width, element size) come out of scalar
// dword loads as well -- there is no scalar byte load.
__device__ __forceinline__ NodeDesc load_node(const NodeDesc* p) {
  static_assert(sizeof(NodeDesc) % 4 == 0, "whole dwords");
  typedef uint32_t __attribute__((may_alias)) word_t;
  NodeDesc nd;
  word_t* dst = reinterpret_cast<word_t*>(&nd);
#pragma unroll
  for (uint32_t i = 0; i < sizeof(NodeDesc) / 4; ++i) dst[i] = sload32(reinterpret_cast<const uint32_t*>(p) + i);
  return nd;
}

// One kernel per operation (used when the plan is not fused: profiling, tuning).
template <uint32_t OP>
__device__ __forceinline__ void single_op_kernel(const KArgs& a) {
  __shared__ __align__(16) uint8_t smem[kSmemBytes + kPhaseBytes];
  NBL_PHASE_BEGIN(smem, kSmemBytes);
  TileRef tr;
  tr.node = sload32(&a.tiles[blockIdx.x].node);
  tr.tile = sload32(&a.tiles[blockIdx.x].tile);
  const uint32_t aux = sload32(a.tile_aux + blockIdx.x);
  NBL_PHASE(smem, kSmemBytes, 0);  // tile entry
  const NodeDesc nd = load_node(a.nodes + tr.node);
  run_op<OP>(a, nd, tr.tile & kTileMask, smem, aux);
  NBL_PHASE(smem, kSmemBytes, 7);  // stores acknowledged
}
#define NBL_KERNEL_ES(name, KID) \
  template <int ES> __global__ __launch_bounds__(256, 8) void name(KArgs a) { single_op_kernel<make_op(KID, ES)>(a); }
#define NBL_KERNEL(name, KID) \
  __global__ __launch_bounds__(256, 8) void name(KArgs a) { single_op_kernel<make_op(KID, 1)>(a); }
NBL_KERNEL_ES(k_materialize, K_MATERIALIZE)
NBL_KERNEL_ES(k_materialize_fbw, K_MATERIALIZE_FBW)
NBL_KERNEL_ES(k_materialize_raw, K_MATERIALIZE_RAW)
NBL_KERNEL(k_bits_to_bytes, K_BITS_TO_BYTES)
NBL_KERNEL(k_reduce_popc, K_REDUCE_POPC)
NBL_KERNEL(k_rle_ends, K_RLE_ENDS)
NBL_KERNEL_ES(k_rle_expand, K_RLE_EXPAND)
NBL_KERNEL(k_rle_expand_bool, K_RLE_EXPAND_BOOL)
NBL_KERNEL(k_sparse_bits, K_SPARSE_BITS)
NBL_KERNEL_ES(k_select_scatter, K_SELECT_SCATTER)
NBL_KERNEL(k_select_bits, K_SELECT_BITS)
NBL_KERNEL_ES(k_dict_gather, K_DICT_GATHER)
NBL_KERNEL_ES(k_varint_decode, K_VARINT_DECODE)
NBL_KERNEL(k_string_views, K_STRING_VIEWS)
NBL_KERNEL(k_copy_bytes, K_COPY_BYTES)
NBL_KERNEL_ES(k_frame_unpack, K_FRAME_UNPACK)
#undef NBL_KERNEL_ES
#undef NBL_KERNEL
template <int ES>
__global__ __launch_bounds__(kWaveBlock, 8) void k_select_scatter_wave(KArgs a) {
  // One tile per workgroup, on purpose: a wave's vector loads return in order
  // with its stores, so a second tile's first loads would queue behind the
  // first tile's 16 KiB of stores (two or four tiles per workgroup, consecutive
  // or a grid apart: 363 -> 396-412 us on the 5,000-column stripe), while a new
  // workgroup starts with an empty queue.
  __shared__ __align__(16) uint8_t smem[kWaveSmemBytes];
  const uint32_t node = sload32(&a.tiles[blockIdx.x].node);
  const uint32_t tile = sload32(&a.tiles[blockIdx.x].tile) & kTileMask;
  const uint32_t aux = sload32(a.tile_aux + blockIdx.x);
  const NodeDesc nd = load_node(a.nodes + node);
  body_select_scatter_wave<ES>(a, nd, tile, smem, aux);
}

// The fused launch: one grid runs every step of one cascade level of the whole
// batch; each workgroup reads its operation from its tile entry (workgroup
// uniform, so the switch costs one scalar branch).  LAST only changes the
// kernel's name, so that profiles tell the output-writing level from the
// preparatory ones.
// The kernel's arguments, read from the kernel-argument segment where they are
// needed instead of through the by-value parameter: the compiler loads a
// parameter's fields at the kernel's entry and keeps them in scalar registers
// across the whole switch, and together with a node descriptor they do not fit
// the 80 scalar registers of a kernel that keeps 8 workgroups per CU -- it
// spilled hundreds of them through vector-register lanes.  KArgs is the only
// parameter, so it starts the segment.
typedef const KArgs __attribute__((address_space(4))) * KArgsSegment;
__device__ __forceinline__ KArgs load_args(KArgsSegment k) {
  KArgs a;
  a.arena = k->arena; a.out = k->out; a.work = k->work; a.nodes = k->nodes; a.tiles = k->tiles;
  a.tile_aux = k->tile_aux; a.parts = k->parts; a.totals = k->totals; a.status = k->status; a.look = k->look;
  a.n_tiles = k->n_tiles; a.gen = k->gen;
  return a;
}

__global__ __launch_bounds__(kBlock, 3) void k_dict_lds(KArgs a) {
  __shared__ __align__(16) uint8_t smem[kDictLdsSmemBytes];
  const uint32_t node = sload32(&a.tiles[blockIdx.x].node);
  const uint32_t group = sload32(&a.tiles[blockIdx.x].tile) & kTileMask;
  const uint32_t aux = sload32(a.tile_aux + blockIdx.x);
  const NodeDesc nd = load_node(a.nodes + node);
  body_dict_lds(a, nd, group, smem, aux);
}

__global__ __launch_bounds__(kDictStrBlock, 1) void k_dict_str(KArgs a) {
  __shared__ __align__(16) uint8_t smem[kDictStrSmemBytes];
  const uint32_t node = sload32(&a.tiles[blockIdx.x].node);
  const uint32_t group = sload32(&a.tiles[blockIdx.x].tile) & kTileMask;
  const uint32_t aux = sload32(a.tile_aux + blockIdx.x);
  const NodeDesc nd = load_node(a.nodes + node);
  body_dict_str(a, nd, group, smem, aux);
}

template <bool LAST>
__global__ __launch_bounds__(256, 8) void k_level(KArgs) {
  __shared__ __align__(16) uint8_t smem[kSmemBytes + kPhaseBytes];
  const KArgsSegment ka = (KArgsSegment)__builtin_amdgcn_kernarg_segment_ptr();
  TileRef tr;
  tr.node = sload32(&ka->tiles[blockIdx.x].node);
  tr.tile = sload32(&ka->tiles[blockIdx.x].tile);
  const uint32_t aux = sload32(ka->tile_aux + blockIdx.x);
  const uint32_t tile = tr.tile & kTileMask;
  // every operation copies the arguments and its node's descriptor itself, so
  // only what that operation reads is live in it
#define NBL_OP(KID, ES)                                      \
  case make_op(KID, ES): {                                   \
    const KArgs a = load_args(ka);                           \
    const NodeDesc nd = load_node(a.nodes + tr.node);        \
    run_op<make_op(KID, ES)>(a, nd, tile, smem, aux);        \
  } break;
#define NBL_OP_ALL_ES(KID) NBL_OP(KID, 1) NBL_OP(KID, 2) NBL_OP(KID, 4) NBL_OP(KID, 8)
  switch (tr.tile >> kOpShift) {
    NBL_OP_ALL_ES(K_MATERIALIZE)
    NBL_OP_ALL_ES(K_MATERIALIZE_FBW)
    NBL_OP_ALL_ES(K_MATERIALIZE_RAW)
    NBL_OP(K_BITS_TO_BYTES, 1)
    NBL_OP(K_REDUCE_POPC, 1)
    NBL_OP(K_RLE_ENDS, 1)
    NBL_OP_ALL_ES(K_RLE_EXPAND)
    NBL_OP(K_RLE_EXPAND_BOOL, 1)
    NBL_OP(K_SPARSE_BITS, 1)
    NBL_OP_ALL_ES(K_SELECT_SCATTER)
    NBL_OP(K_SELECT_BITS, 1)
    NBL_OP_ALL_ES(K_DICT_GATHER)
    NBL_OP(K_VARINT_DECODE, 4)
    NBL_OP(K_VARINT_DECODE, 8)
    NBL_OP(K_STRING_VIEWS, 1)
    NBL_OP(K_COPY_BYTES, 1)
    default: break;  // K_FRAME_UNPACK is always launched on its own (kernel_is_fusable)
  }
#undef NBL_OP_ALL_ES
#undef NBL_OP
}

// ------------------------------------------------------------------ copy back by kernel
// Decoded bytes -> pinned host memory, written by the CUs over the host link
// instead of by a copy engine.  A few resident workgroups are plenty: the link
// (about 55 GB/s), not the chip, sets the pace, and every wave writes whole
// 1 KiB runs.  64 workgroups keep the link as busy as 256 do (32 no longer),
// and a decode that runs beside the copy is held up less the fewer writes to
// the host are pending (2-4 ms beside 256 workgroups, 1-1.5 beside 64, 0.65
// alone, for the 2.6 GB headline stripe).  The bytes behind the last 16-byte
// vector go one by one.
constexpr uint32_t kCopyOutBlocks = 64;
__global__ __launch_bounds__(kBlock) void k_copy_to_host(const u32x4* __restrict__ src, u32x4* __restrict__ dst, uint64_t bytes) {
  const uint64_t n = bytes >> 4;
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kBlock)
    dst[i] = __builtin_nontemporal_load(src + i);
  if (blockIdx.x == 0 && threadIdx.x < (bytes & 15))
    reinterpret_cast<uint8_t*>(dst)[(n << 4) + threadIdx.x] = reinterpret_cast<const uint8_t*>(src)[(n << 4) + threadIdx.x];
}

}  // namespace nbl

// ------------------------------------------------------------------ launcher
// `kernel` is a KernelId, or K_LEVEL / K_LEVEL_LAST for a fused launch (then
// `es` is ignored).  Kernels whose body does not depend on the element size
// take any `es`.
#ifdef NBL_PHASE_TIMING
extern "C" int nbl_phase_read(uint32_t* out, uint32_t n_tiles) {
  if (n_tiles > nbl::kPhaseTiles) n_tiles = nbl::kPhaseTiles;
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(nbl::g_phase), sizeof(uint32_t) * 8 * n_tiles) == hipSuccess ? 0 : -1;
}
#endif

extern "C" int nbl_launch_copy_to_host(void* host_dst, const void* dev_src, uint64_t bytes, void* stream) {
  using namespace nbl;
  if (bytes == 0) return 0;
  const uint64_t vectors = bytes >> 4;
  const uint32_t blocks = (uint32_t)std::min<uint64_t>(kCopyOutBlocks, std::max<uint64_t>(1, (vectors + kBlock - 1) / kBlock));
  hipLaunchKernelGGL(k_copy_to_host, dim3(blocks), dim3(kBlock), 0, static_cast<hipStream_t>(stream), static_cast<const u32x4*>(dev_src),
                     static_cast<u32x4*>(host_dst), bytes);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

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
    case K_LEVEL: hipLaunchKernelGGL(k_level<false>, grid, block, 0, s, a); break;
    case K_LEVEL_LAST: hipLaunchKernelGGL(k_level<true>, grid, block, 0, s, a); break;
    case K_MATERIALIZE: NBL_BY_ES(k_materialize); break;
    case K_MATERIALIZE_FBW: NBL_BY_ES(k_materialize_fbw); break;
    case K_MATERIALIZE_RAW: NBL_BY_ES(k_materialize_raw); break;
    case K_BITS_TO_BYTES: hipLaunchKernelGGL(k_bits_to_bytes, grid, block, 0, s, a); break;
    case K_REDUCE_POPC: hipLaunchKernelGGL(k_reduce_popc, grid, block, 0, s, a); break;
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
    case K_FRAME_UNPACK: NBL_BY_ES(k_frame_unpack); break;
    case K_SELECT_SCATTER_WAVE:
      if (es == 4) hipLaunchKernelGGL(k_select_scatter_wave<4>, grid, dim3(kWaveBlock), 0, s, a);
      else if (es == 8) hipLaunchKernelGGL(k_select_scatter_wave<8>, grid, dim3(kWaveBlock), 0, s, a);
      else return -1;
      break;
    case K_DICT_LDS:
      if (es != 8) return -1;
      hipLaunchKernelGGL(k_dict_lds, grid, block, 0, s, a);
      break;
    case K_DICT_STR:
      if (es != 8) return -1;
      hipLaunchKernelGGL(k_dict_str, grid, dim3(kDictStrBlock), 0, s, a);
      break;
    default: return -1;
  }
#undef NBL_BY_ES
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

/*
 * nimble_oracle.c -- CPU restatement of the reference's cascaded-stream decode.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path may link, load or
 * call this file: only tests/, __graft_entry__.smoke() and the cpu_baseline
 * leg of bench.py use it, as the checker.
 *
 * It follows the reference's decode algorithm function by function (citations
 * are relative to the reference checkout, dwio/nimble/...):
 *
 *   prefix parse ............ encodings/common/EncodingPrefix.h:65-92,
 *                             encodings/common/Encoding.cpp:22-33
 *   factory dispatch ........ encodings/common/EncodingFactory.cpp:62-164
 *   FixedBitArray get ....... common/FixedBitArray.cpp:41-57 (every value through get();
 *                             the checker's path)
 *   FixedBitArray bulk ...... common/FixedBitArray.cpp:114-156,221-326,345-358,426-508,
 *                             common/FixedBitArray.h:104-138 (64-value groups per bit
 *                             width etc.; same results, selected by orc_set_bulk /
 *                             orc_decode_all_fast for the CPU baseline's timing only)
 *   FixedBitWidth ........... encodings/FixedBitWidthEncoding.h:149-192
 *   Trivial numeric ......... encodings/TrivialEncoding.h:252-300
 *   Trivial bool / string ... encodings/TrivialEncoding.cpp:101-164,291-381
 *   RLE ..................... encodings/RLEEncoding.h:113-161,816-897,
 *                             encodings/RLEEncoding.cpp:20-68
 *   BufferedEncoding ........ encodings/common/BufferedEncoding.h:24-64
 *   Dictionary .............. encodings/DictionaryEncoding.h:193-236
 *   Nullable ................ encodings/NullableEncoding.h:151-306
 *   SparseBool .............. encodings/SparseBoolEncoding.cpp:70-142
 *   Constant ................ encodings/ConstantEncoding.h:72-77,292-326,
 *                             encodings/ConstantEncoding.cpp:21-33
 *   MainlyConstant .......... encodings/MainlyConstantEncoding.h:208-253,758-787,
 *                             :842-860, encodings/MainlyConstantEncoding.cpp:21-43
 *   Varint .................. encodings/VarintEncoding.h:94-140,
 *                             common/Varint.h:83-143
 *   BlockBitPacking ......... encodings/BlockBitPackingEncoding.h:311-348,660-874
 *   FOR ..................... encodings/ForEncoding.h:174-196,316-383,398-546
 *   EncodingView read ....... encodings/views/EncodingView.h:32-104
 *   scatter for non-nullable  encodings/common/Encoding.h:412-460
 *   chunk framing ........... common/ChunkHeader.h:36-40,
 *                             velox/ChunkedStream.cpp:36-72
 *   multi-chunk stitching ... velox/ChunkedStreamDecoder.cpp:75-205
 *   selective bulk reads .... velox/selective/ChunkedDecoder.h:206-268,
 *                             velox/selective/ChunkedDecoder.cpp:155-206,270-286
 *
 * Third-party arithmetic the reference calls but does not vendor (velox @
 * 351b0a72f446c75e0bd0ad179063a7a02f8894fc: bits::fillBits / copyBits /
 * setBit / clearBit / isBitSet / findSetBit, Bitmap / BitmapBuilder,
 * simd::simdFill) is restated from its published, unambiguous semantics; the
 * reference pins those in common/tests/BitsTests.cpp:33-112.  BlockBitPacking
 * unpacks full groups with velox's copy of Lemire's fastunpack (same pin,
 * velox/dwio/common/Lemire/BitPacking/bitpackinghelpers.h): the published
 * layout of that scalar packer is the plain LSB-first bit stream, which is also
 * what the reference's FixedBitArray tail and its 64-bit writer
 * (bulkSetWithBaseline, BlockBitPackingEncoding.h:224-227) use for the same
 * blocks, so one block is one FixedBitArray image here.
 *
 * Parity pin status: the reference holds no encoded-byte golden vectors for
 * these encodings and cannot be compiled offline (velox / folly / glog are
 * absent), so byte-level parity of the encodings is pinned only by the
 * reference's layout documentation, its tests' header facts and deterministic
 * inputs (tests/golden/make_golden.py, tests/test_oracle_golden.py,
 * tests/test_oracle_semantics.py): "parity unpinned" beyond those.
 *
 * Plain C99, no dependencies.  All multi-byte fields little-endian, unaligned.
 */
#include <setjmp.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

enum {
  ET_TRIVIAL = 0, ET_RLE = 1, ET_DICTIONARY = 2, ET_FIXEDBITWIDTH = 3, ET_SENTINEL = 4,
  ET_NULLABLE = 5, ET_SPARSEBOOL = 6, ET_VARINT = 7, ET_DELTA = 8, ET_CONSTANT = 9,
  ET_MAINLYCONSTANT = 10, ET_BLOCKBITPACKING = 15, ET_FOR = 18
};
enum {
  DT_UNDEFINED = 0, DT_INT8 = 1, DT_UINT8 = 2, DT_INT16 = 3, DT_UINT16 = 4, DT_INT32 = 5,
  DT_UINT32 = 6, DT_INT64 = 7, DT_UINT64 = 8, DT_FLOAT = 9, DT_DOUBLE = 10, DT_BOOL = 11,
  DT_STRING = 12
};

/* std::string_view stand-in used for string materialisation. */
typedef struct { const char* ptr; uint64_t len; } orc_sv;

typedef struct orc_ctx {
  int varint_rows;           /* Encoding::Options::useVarintRowCount */
  int bulk;                  /* FixedBitWidth reads through the bulk path below (same results; CPU baseline timing) */
  char err[256];
  jmp_buf jb;
  void** allocs;             /* string buffers handed out by the factory */
  size_t nallocs, capallocs;
  struct orc_enc** encs;     /* every node of the tree, for teardown */
  size_t nencs, capencs;
} orc_ctx;

typedef struct orc_enc orc_enc;
struct orc_enc {
  orc_ctx* ctx;
  const uint8_t* data; uint64_t size;
  uint8_t etype, dtype; uint32_t rows; uint32_t prefix;
  int es;                     /* bytes per materialised element (string: 16) */
  void (*reset)(orc_enc*);
  void (*skip)(orc_enc*, uint32_t);
  void (*materialize)(orc_enc*, uint32_t, void*);
  void (*bools_as_bits)(orc_enc*, uint32_t, uint64_t*, int);
  int nullable;
  orc_enc* child[2];
  /* per-encoding state */
  uint32_t row;
  const uint8_t* payload;     /* FBW packed / trivial values / bitmap / varint start */
  const uint8_t* pos;         /* varint cursor, string blob cursor */
  int bit_width; uint64_t baseline; uint64_t mask;
  uint64_t value;             /* constant / common / current run value (numeric) */
  orc_sv svalue;              /* same for strings */
  uint32_t copies_remaining;
  int initial_bool, cur_bool;
  int sparse_value; uint32_t next_index;
  /* BufferedEncoding<uint32_t,32> for child[0], BufferedEncoding<T,128> for child[1] */
  uint32_t buf32[32]; uint16_t buf32_pos; uint32_t buf32_consumed;
  uint8_t* bufv; uint16_t bufv_pos; uint32_t bufv_consumed;
  /* dictionary alphabet */
  uint8_t* alphabet; uint32_t alphabet_count;
  const char* blob; uint64_t blob_bytes;
  /* BlockBitPacking blocks / FOR frames */
  uint32_t seg_size, seg_count, seg_first, seg_last;
  uint8_t* seg_width; uint8_t* seg_base; uint64_t* seg_off;   /* per segment: bit width, baseline (es bytes), offset */
  uint64_t payload_bytes;
};

static void orc_fail(orc_ctx* c, const char* fmt, ...) {
  va_list ap; va_start(ap, fmt); vsnprintf(c->err, sizeof c->err, fmt, ap); va_end(ap);
  longjmp(c->jb, 1);
}
#define ORC_CHECK(c, cond, ...) do { if (!(cond)) orc_fail((c), __VA_ARGS__); } while (0)

static void* orc_xmalloc(orc_ctx* c, size_t n) {
  void* p = malloc(n ? n : 1);
  if (!p) orc_fail(c, "out of memory");
  return p;
}
/* stringBufferFactory stand-in: memory lives until the context is freed. */
static void* orc_string_buffer(orc_ctx* c, uint32_t bytes) {
  if (c->nallocs == c->capallocs) {
    c->capallocs = c->capallocs ? c->capallocs * 2 : 8;
    c->allocs = (void**)realloc(c->allocs, c->capallocs * sizeof(void*));
  }
  void* p = orc_xmalloc(c, bytes);
  c->allocs[c->nallocs++] = p;
  return p;
}

static inline uint64_t ld_le(const uint8_t* p, int n) { uint64_t v = 0; memcpy(&v, p, (size_t)n); return v; }
static inline void st_le(void* p, uint64_t v, int n) { memcpy(p, &v, (size_t)n); }
static inline uint32_t ld32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }

/* ---- velox::bits restatements ------------------------------------------ */
static inline int bit_is_set(const void* bits, uint64_t i) { return (((const uint8_t*)bits)[i >> 3] >> (i & 7)) & 1; }
static inline void bit_set(void* bits, uint64_t i) { ((uint8_t*)bits)[i >> 3] |= (uint8_t)(1u << (i & 7)); }
static inline void bit_clear(void* bits, uint64_t i) { ((uint8_t*)bits)[i >> 3] &= (uint8_t)~(1u << (i & 7)); }
static void fill_bits(void* bits, uint64_t begin, uint64_t end, int value) {
  for (uint64_t i = begin; i < end; ++i) { if (value) bit_set(bits, i); else bit_clear(bits, i); }
}
static void copy_bits(const void* src, uint64_t so, void* dst, uint64_t d0, uint64_t n) {
  for (uint64_t i = 0; i < n; ++i) { if (bit_is_set(src, so + i)) bit_set(dst, d0 + i); else bit_clear(dst, d0 + i); }
}
/* n-th (1-based) set bit in [begin,end); `end` when there are fewer. */
uint32_t orc_find_set_bit(const void* bits, uint32_t begin, uint32_t end, uint32_t n) {
  uint32_t seen = 0;
  for (uint32_t i = begin; i < end; ++i) if (bit_is_set(bits, i) && ++seen == n) return i;
  return end;
}

/* ---- varint (common/Varint.h:83-143) ------------------------------------ */
static uint32_t read_varint32(const uint8_t** pos) {
  uint32_t value = (**pos) & 127;
  if (!(*((*pos)++) & 128)) return value;
  value |= (uint32_t)(**pos & 127) << 7;
  if (!(*((*pos)++) & 128)) return value;
  value |= (uint32_t)(**pos & 127) << 14;
  if (!(*((*pos)++) & 128)) return value;
  value |= (uint32_t)(**pos & 127) << 21;
  if (!(*((*pos)++) & 128)) return value;
  value |= (uint32_t)(*((*pos)++) & 127) << 28;
  return value;
}
static uint64_t read_varint64(const uint8_t** pos) {
  uint64_t value = (**pos) & 127;
  for (int shift = 7; shift <= 56; shift += 7) {
    if (!(*((*pos)++) & 128)) return value;
    value |= (uint64_t)(**pos & 127) << shift;
  }
  if (!(*((*pos)++) & 128)) return value;
  value |= (uint64_t)(*((*pos)++) & 127) << 63;
  return value;
}

/* ---- prefix -------------------------------------------------------------- */
static orc_enc* orc_create_in(orc_ctx* c, const uint8_t* data, uint64_t size);

static orc_enc* enc_alloc(orc_ctx* c, const uint8_t* data, uint64_t size) {
  ORC_CHECK(c, size >= 3, "encoding shorter than its prefix");
  orc_enc* e = (orc_enc*)calloc(1, sizeof(orc_enc));
  if (!e) orc_fail(c, "out of memory");
  if (c->nencs == c->capencs) {
    c->capencs = c->capencs ? c->capencs * 2 : 8;
    c->encs = (orc_enc**)realloc(c->encs, c->capencs * sizeof(orc_enc*));
  }
  c->encs[c->nencs++] = e;
  e->ctx = c; e->data = data; e->size = size;
  e->etype = data[0]; e->dtype = data[1];
  if (c->varint_rows) {                 /* EncodingPrefix.h:77-92 */
    const uint8_t* p = data + 2;
    e->rows = read_varint32(&p);
    e->prefix = (uint32_t)(p - data);
  } else {
    ORC_CHECK(c, size >= 6, "encoding shorter than its prefix");
    e->rows = ld32(data + 2);
    e->prefix = 6;
  }
  ORC_CHECK(c, e->prefix <= size, "prefix beyond end of data");
  switch (e->dtype) {
    case DT_INT8: case DT_UINT8: case DT_BOOL: e->es = 1; break;
    case DT_INT16: case DT_UINT16: e->es = 2; break;
    case DT_INT32: case DT_UINT32: case DT_FLOAT: e->es = 4; break;
    case DT_INT64: case DT_UINT64: case DT_DOUBLE: e->es = 8; break;
    case DT_STRING: e->es = (int)sizeof(orc_sv); break;
    default: orc_fail(c, "invalid DataType %d", (int)data[1]);
  }
  return e;
}

/* ---- FixedBitArray ------------------------------------------------------- */
static inline uint64_t fba_get(const orc_enc* e, uint64_t index) {  /* FixedBitArray.cpp:41-57 */
  const uint64_t bits = index * (uint64_t)e->bit_width;
  const uint64_t offset = bits >> 3, rem = bits & 7;
  uint64_t word; memcpy(&word, e->payload + offset, 8);
  if (e->bit_width > 58) {
    const int overflow = e->bit_width + (int)rem - 64;
    if (overflow > 0) {
      uint64_t next; memcpy(&next, e->payload + offset + 8, 8);
      return ((word >> rem) | (next << (e->bit_width - overflow))) & e->mask;
    }
  }
  return (word >> rem) & e->mask;
}

/* ---- FixedBitArray bulk reads ----------------------------------------------
 * The shape of the reference's fast path, for the CPU baseline's timing: every
 * 64 values end on a word boundary, so after aligning with single gets the
 * values are unpacked in groups of 64 by code specialised per bit width
 * (bulkGet32Internal, FixedBitArray.cpp:114-156,221-326); 8-byte outputs of
 * width >= 32 take the byte-aligned loads (:345-358,438-463), one unaligned
 * word per value up to 58 bits (:478-488) and the two-word stitch above that
 * (:490-507).  Results are identical to fba_get + baseline (checked by
 * tests/test_oracle_semantics.py); only the instruction mix differs. */
#define FBA_GROUP64(NAME, T)                                                                                     \
  static inline __attribute__((always_inline)) void NAME(const uint8_t* src, T* out, T base, const int w) {      \
    uint64_t words[33];                                                                                          \
    memcpy(words, src, (size_t)8 * (size_t)w);                                                                   \
    words[w] = 0;                                                                                                \
    const uint64_t mask = (1ull << w) - 1ull;                                                                    \
    _Pragma("GCC unroll 64")                                                                                     \
    for (int i = 0; i < 64; ++i) {                                                                               \
      const int bit = i * w, wi = bit >> 6, sh = bit & 63;                                                       \
      uint64_t v = words[wi] >> sh;                                                                              \
      if (sh + w > 64) v |= words[wi + 1] << (64 - sh);                                                          \
      out[i] = (T)((T)(v & mask) + base);                                                                        \
    }                                                                                                            \
  }
FBA_GROUP64(fba_group64_u32, uint32_t)
FBA_GROUP64(fba_group64_u64, uint64_t)
#undef FBA_GROUP64
#define FBA_CASE(W, FN) case W: for (uint64_t g = 0; g < groups; ++g) FN(src + g * 8 * W, out + g * 64, base, W); break;
#define FBA_CASES(FN)                                                                                              \
  FBA_CASE(1, FN) FBA_CASE(2, FN) FBA_CASE(3, FN) FBA_CASE(4, FN) FBA_CASE(5, FN) FBA_CASE(6, FN) FBA_CASE(7, FN)    \
  FBA_CASE(8, FN) FBA_CASE(9, FN) FBA_CASE(10, FN) FBA_CASE(11, FN) FBA_CASE(12, FN) FBA_CASE(13, FN)               \
  FBA_CASE(14, FN) FBA_CASE(15, FN) FBA_CASE(16, FN) FBA_CASE(17, FN) FBA_CASE(18, FN) FBA_CASE(19, FN)             \
  FBA_CASE(20, FN) FBA_CASE(21, FN) FBA_CASE(22, FN) FBA_CASE(23, FN) FBA_CASE(24, FN) FBA_CASE(25, FN)             \
  FBA_CASE(26, FN) FBA_CASE(27, FN) FBA_CASE(28, FN) FBA_CASE(29, FN) FBA_CASE(30, FN) FBA_CASE(31, FN)             \
  FBA_CASE(32, FN)
static void fba_groups_u32(const uint8_t* src, uint32_t* out, uint32_t base, int w, uint64_t groups) {
  switch (w) { FBA_CASES(fba_group64_u32) default: break; }
}
static void fba_groups_u64(const uint8_t* src, uint64_t* out, uint64_t base, int w, uint64_t groups) {
  switch (w) { FBA_CASES(fba_group64_u64) default: break; }
}
#undef FBA_CASES
#undef FBA_CASE
/* bulkGet32Internal<uint32_t, true> (w in [1, 32]) */
static void fba_bulk32(const orc_enc* e, uint64_t start, uint64_t length, uint32_t* values, uint32_t base) {
  const uint64_t aligned = (start + 63) / 64 * 64;
  if (start + length < aligned) {
    for (uint64_t i = start; i < start + length; ++i) *values++ = (uint32_t)fba_get(e, i) + base;
    return;
  }
  for (uint64_t i = start; i < aligned; ++i) *values++ = (uint32_t)fba_get(e, i) + base;
  const uint64_t groups = (length - (aligned - start)) >> 6;
  fba_groups_u32(e->payload + ((aligned * (uint64_t)e->bit_width) >> 3), values, base, e->bit_width, groups);
  values += groups << 6;
  for (uint64_t i = aligned + (groups << 6); i < start + length; ++i) *values++ = (uint32_t)fba_get(e, i) + base;
}
/* bulkGet64WithBaseline */
static void fba_bulk64(const orc_enc* e, uint64_t start, uint64_t length, uint64_t* values, uint64_t base) {
  const int w = e->bit_width;
  if (w < 32) {  /* bulkGetWithBaseline32Into64 */
    const uint64_t aligned = (start + 63) / 64 * 64;
    if (start + length < aligned) {
      for (uint64_t i = start; i < start + length; ++i) *values++ = fba_get(e, i) + base;
      return;
    }
    for (uint64_t i = start; i < aligned; ++i) *values++ = fba_get(e, i) + base;
    const uint64_t groups = (length - (aligned - start)) >> 6;
    fba_groups_u64(e->payload + ((aligned * (uint64_t)w) >> 3), values, base, w, groups);
    values += groups << 6;
    for (uint64_t i = aligned + (groups << 6); i < start + length; ++i) *values++ = fba_get(e, i) + base;
    return;
  }
  if ((w & 7) == 0) {  /* 32, 40, 48, 56, 64: byte-aligned residuals */
    const int bw = w >> 3;
    const uint8_t* next = e->payload + start * (uint64_t)bw;
    if (bw == 4) { for (uint64_t i = 0; i < length; ++i, next += 4) { uint32_t v; memcpy(&v, next, 4); values[i] = (uint64_t)v + base; } }
    else if (bw == 8) { for (uint64_t i = 0; i < length; ++i, next += 8) { uint64_t v; memcpy(&v, next, 8); values[i] = v + base; } }
    else { const uint64_t m = e->mask; for (uint64_t i = 0; i < length; ++i, next += bw) { uint64_t v; memcpy(&v, next, 8); values[i] = (v & m) + base; } }
    return;
  }
  const uint64_t mask = e->mask;
  uint64_t bits = start * (uint64_t)w;
  if (w <= 58) {
    for (uint64_t i = 0; i < length; ++i, bits += (uint64_t)w) {
      uint64_t word; memcpy(&word, e->payload + (bits >> 3), 8);
      values[i] = ((word >> (bits & 7)) & mask) + base;
    }
    return;
  }
  for (uint64_t i = 0; i < length; ++i, bits += (uint64_t)w) {
    const uint64_t off = bits >> 3, rem = bits & 7;
    uint64_t word; memcpy(&word, e->payload + off, 8);
    const int overflow = w + (int)rem - 64;
    if (overflow > 0) {
      uint64_t next; memcpy(&next, e->payload + off + 8, 8);
      values[i] = (((word >> rem) | (next << (w - overflow))) & mask) + base;
    } else {
      values[i] = ((word >> rem) & mask) + base;
    }
  }
}
/* bulkGetWithBaseline<T> (FixedBitArray.h:104-138): 8- and 4-byte outputs directly,
 * narrow outputs through 1024-element uint32 chunks */
static void fbw_materialize_bulk(orc_enc* e, uint32_t n, void* buffer) {
  if (e->bit_width == 0) {  /* bufferSize is 0 bits wide: every value is the baseline */
    switch (e->es) {
      case 8: { uint64_t* o = (uint64_t*)buffer; for (uint32_t i = 0; i < n; ++i) o[i] = e->baseline; break; }
      case 4: { uint32_t* o = (uint32_t*)buffer; for (uint32_t i = 0; i < n; ++i) o[i] = (uint32_t)e->baseline; break; }
      case 2: { uint16_t* o = (uint16_t*)buffer; for (uint32_t i = 0; i < n; ++i) o[i] = (uint16_t)e->baseline; break; }
      default: memset(buffer, (int)(uint8_t)e->baseline, n); break;
    }
  } else if (e->es == 8) {
    fba_bulk64(e, e->row, n, (uint64_t*)buffer, e->baseline);
  } else if (e->es == 4) {
    fba_bulk32(e, e->row, n, (uint32_t*)buffer, (uint32_t)e->baseline);
  } else {
    uint32_t tmp[1024];
    for (uint32_t done = 0; done < n;) {
      const uint32_t k = n - done < 1024 ? n - done : 1024;
      fba_bulk32(e, (uint64_t)e->row + done, k, tmp, (uint32_t)e->baseline);
      if (e->es == 2) { uint16_t* o = (uint16_t*)buffer + done; for (uint32_t i = 0; i < k; ++i) o[i] = (uint16_t)tmp[i]; }
      else { uint8_t* o = (uint8_t*)buffer + done; for (uint32_t i = 0; i < k; ++i) o[i] = (uint8_t)tmp[i]; }
      done += k;
    }
  }
  e->row += n;
}

static void fbw_reset(orc_enc* e) { e->row = 0; }
static void fbw_skip(orc_enc* e, uint32_t n) { e->row += n; }
static void fbw_materialize(orc_enc* e, uint32_t n, void* buffer) {
  if (e->ctx->bulk) { fbw_materialize_bulk(e, n, buffer); return; }
  /* bulkGetWithBaseline<T>: value = packed bits + baseline, truncated to T
   * (FixedBitArray.h:104-138; narrow types go through a uint32 widening). */
  uint8_t* out = (uint8_t*)buffer;
  const uint64_t row = e->row;
  switch (e->es) {
    case 8: {
      uint64_t* o = (uint64_t*)out;
      for (uint32_t i = 0; i < n; ++i) o[i] = fba_get(e, row + i) + e->baseline;
      break;
    }
    case 4: {
      uint32_t* o = (uint32_t*)out;
      const uint32_t b = (uint32_t)e->baseline;
      for (uint32_t i = 0; i < n; ++i) o[i] = (uint32_t)fba_get(e, row + i) + b;
      break;
    }
    case 2: {
      uint16_t* o = (uint16_t*)out;
      const uint32_t b = (uint32_t)e->baseline;
      for (uint32_t i = 0; i < n; ++i) o[i] = (uint16_t)((uint32_t)fba_get(e, row + i) + b);
      break;
    }
    default: {
      const uint32_t b = (uint32_t)e->baseline;
      for (uint32_t i = 0; i < n; ++i) out[i] = (uint8_t)((uint32_t)fba_get(e, row + i) + b);
      break;
    }
  }
  e->row += n;
}
static void fbw_init(orc_enc* e) {
  orc_ctx* c = e->ctx;
  ORC_CHECK(c, e->dtype != DT_BOOL && e->dtype != DT_STRING, "FixedBitWidth needs a numeric type");
  ORC_CHECK(c, (uint64_t)e->prefix + 2 + (uint64_t)e->es <= e->size, "FixedBitWidth header beyond end");
  const uint8_t* p = e->data + e->prefix;
  ORC_CHECK(c, p[0] == 0, "compressed FixedBitWidth payload not supported by the oracle");
  e->baseline = ld_le(p + 1, e->es);
  e->bit_width = p[1 + e->es];
  e->payload = p + 2 + e->es;
  ORC_CHECK(c, e->bit_width <= 64, "bit width must lie in [0, 64], got %d", e->bit_width);
  ORC_CHECK(c, e->es == 8 || e->bit_width <= 32, "bit width %d too wide for a %d-byte type", e->bit_width, e->es);
  e->mask = e->bit_width == 64 ? ~0ull : ((1ull << e->bit_width) - 1);
  /* The reference does not validate the payload length; the oracle does so it
   * never reads outside `data` (needs ceil(rows*w/8) bytes + slop for the word loads). */
  const uint64_t need = ((uint64_t)e->rows * (uint64_t)e->bit_width + 7) / 8;
  ORC_CHECK(c, (uint64_t)(e->payload - e->data) + need + (e->rows ? 7 : 0) <= e->size + 0,
            "FixedBitWidth payload truncated");
  e->reset = fbw_reset; e->skip = fbw_skip; e->materialize = fbw_materialize;
}

/* ---- Trivial ------------------------------------------------------------- */
static void triv_reset(orc_enc* e) { e->row = 0; }
static void triv_skip(orc_enc* e, uint32_t n) { e->row += n; }
static void triv_materialize(orc_enc* e, uint32_t n, void* buffer) {
  memcpy(buffer, e->payload + (uint64_t)e->row * e->es, (uint64_t)n * e->es);
  e->row += n;
}
static void trivb_materialize(orc_enc* e, uint32_t n, void* buffer) {  /* TrivialEncoding.cpp:329-368 */
  uint8_t* out = (uint8_t*)buffer;
  for (uint32_t i = 0; i < n; ++i) out[i] = (uint8_t)bit_is_set(e->payload, (uint64_t)e->row + i);
  e->row += n;
}
static void trivb_bits(orc_enc* e, uint32_t n, uint64_t* buffer, int begin) {  /* :370-381 */
  copy_bits(e->payload, e->row, buffer, (uint64_t)begin, n);
  e->row += n;
}
static void trivs_reset(orc_enc* e) { e->row = 0; e->pos = (const uint8_t*)e->blob; e->child[0]->reset(e->child[0]); }
static void trivs_skip(orc_enc* e, uint32_t n) {  /* TrivialEncoding.cpp:144-149 */
  uint32_t* lens = (uint32_t*)orc_xmalloc(e->ctx, (size_t)n * 4);
  e->child[0]->materialize(e->child[0], n, lens);
  uint32_t sum = 0; for (uint32_t i = 0; i < n; ++i) sum += lens[i];
  free(lens);
  e->row += n; e->pos += sum;
}
static void trivs_materialize(orc_enc* e, uint32_t n, void* buffer) {  /* :151-164 */
  uint32_t* lens = (uint32_t*)orc_xmalloc(e->ctx, (size_t)n * 4);
  e->child[0]->materialize(e->child[0], n, lens);
  orc_sv* out = (orc_sv*)buffer;
  const char* pos = (const char*)e->pos;
  for (uint32_t i = 0; i < n; ++i) { out[i].ptr = pos; out[i].len = lens[i]; pos += lens[i]; }
  free(lens);
  e->pos = (const uint8_t*)pos; e->row += n;
}
static void trivial_init(orc_enc* e) {
  orc_ctx* c = e->ctx;
  ORC_CHECK(c, (uint64_t)e->prefix + 1 <= e->size, "Trivial header beyond end");
  const uint8_t* p = e->data + e->prefix;
  ORC_CHECK(c, p[0] == 0, "compressed Trivial payload not supported by the oracle");
  if (e->dtype == DT_STRING) {  /* TrivialEncoding.cpp:33-44,101-136 */
    ORC_CHECK(c, (uint64_t)e->prefix + 5 <= e->size, "Trivial<string> header beyond end");
    const uint32_t lengths_size = ld32(p + 1);
    ORC_CHECK(c, (uint64_t)e->prefix + 5 + lengths_size <= e->size, "Trivial<string> lengths beyond end");
    e->child[0] = orc_create_in(c, p + 5, lengths_size);
    ORC_CHECK(c, e->child[0]->es == 4, "Trivial<string> lengths must be a 4-byte type");
    const uint8_t* blob = p + 5 + lengths_size;
    e->blob_bytes = (uint64_t)(e->data + e->size - blob);
    char* copy = (char*)orc_string_buffer(c, (uint32_t)e->blob_bytes);
    memcpy(copy, blob, e->blob_bytes);
    e->blob = copy; e->pos = (const uint8_t*)copy;
    e->reset = trivs_reset; e->skip = trivs_skip; e->materialize = trivs_materialize;
  } else if (e->dtype == DT_BOOL) {  /* :291-319 */
    e->payload = p + 1;
    ORC_CHECK(c, e->payload + ((uint64_t)e->rows + 7) / 8 + 7 == e->data + e->size, "Unexpected trivial encoding end");
    e->reset = triv_reset; e->skip = triv_skip; e->materialize = trivb_materialize; e->bools_as_bits = trivb_bits;
  } else {  /* TrivialEncoding.h:252-283 */
    e->payload = p + 1;
    ORC_CHECK(c, e->payload + (uint64_t)e->rows * e->es == e->data + e->size, "Unexpected trivial encoding end");
    e->reset = triv_reset; e->skip = triv_skip; e->materialize = triv_materialize;
  }
}

/* ---- BufferedEncoding (BufferedEncoding.h:24-64) ------------------------- */
static uint32_t buf32_next(orc_enc* e) {
  if (e->buf32_pos == 32) {
    uint32_t left = e->child[0]->rows - e->buf32_consumed;
    uint32_t n = left < 32 ? left : 32;
    e->child[0]->materialize(e->child[0], n, e->buf32);
    e->buf32_pos = 0;
  }
  ++e->buf32_consumed;
  return e->buf32[e->buf32_pos++];
}
static void buf32_reset(orc_enc* e) { e->buf32_pos = 32; e->buf32_consumed = 0; e->child[0]->reset(e->child[0]); }
static void bufv_next(orc_enc* e, void* out) {
  orc_enc* ch = e->child[1];
  if (e->bufv_pos == 128) {
    uint32_t left = ch->rows - e->bufv_consumed;
    uint32_t n = left < 128 ? left : 128;
    ch->materialize(ch, n, e->bufv);
    e->bufv_pos = 0;
  }
  ++e->bufv_consumed;
  memcpy(out, e->bufv + (size_t)e->bufv_pos * ch->es, (size_t)ch->es);
  ++e->bufv_pos;
}
static void bufv_reset(orc_enc* e) { e->bufv_pos = 128; e->bufv_consumed = 0; e->child[1]->reset(e->child[1]); }

/* ---- RLE ------------------------------------------------------------------ */
static void rle_advance(orc_enc* e) {  /* RLEEncoding.h:289-301 */
  e->copies_remaining = buf32_next(e);
  if (e->dtype == DT_BOOL) {            /* RLEEncoding.cpp:39-42 */
    e->cur_bool = !e->cur_bool;
    e->value = (uint64_t)!e->cur_bool;
  } else if (e->dtype == DT_STRING) {
    bufv_next(e, &e->svalue);
  } else {
    e->value = 0; bufv_next(e, &e->value);
  }
}
static void rle_reset(orc_enc* e) {  /* RLEEncoding.h:126-130 */
  buf32_reset(e);
  if (e->dtype == DT_BOOL) e->cur_bool = e->initial_bool; else bufv_reset(e);
  e->copies_remaining = 0;
}
static void rle_skip(orc_enc* e, uint32_t n) {  /* RLEEncoding.h:132-142 */
  uint32_t left = n;
  while (left > 0) {
    if (left <= e->copies_remaining) { e->copies_remaining -= left; return; }
    left -= e->copies_remaining;
    rle_advance(e);
  }
}
static void rle_fill(orc_enc* e, uint8_t* out, uint32_t n) {
  switch (e->es) {
    case 1: memset(out, (int)(uint8_t)e->value, n); break;
    case 2: { uint16_t v = (uint16_t)e->value; uint16_t* o = (uint16_t*)out; for (uint32_t i = 0; i < n; ++i) o[i] = v; break; }
    case 4: { uint32_t v = (uint32_t)e->value; uint32_t* o = (uint32_t*)out; for (uint32_t i = 0; i < n; ++i) o[i] = v; break; }
    case 8: { uint64_t v = e->value; uint64_t* o = (uint64_t*)out; for (uint32_t i = 0; i < n; ++i) o[i] = v; break; }
    default: { orc_sv* o = (orc_sv*)out; for (uint32_t i = 0; i < n; ++i) o[i] = e->svalue; break; }
  }
}
static void rle_materialize(orc_enc* e, uint32_t n, void* buffer) {  /* RLEEncoding.h:144-161,879-897 */
  uint32_t left = n; uint8_t* out = (uint8_t*)buffer;
  while (left > 0) {
    if (e->copies_remaining == 0) rle_advance(e);
    if (left < e->copies_remaining) { rle_fill(e, out, left); e->copies_remaining -= left; return; }
    rle_fill(e, out, e->copies_remaining);
    out += (uint64_t)e->copies_remaining * e->es;
    left -= e->copies_remaining;
    e->copies_remaining = 0;
  }
}
static void rle_bits(orc_enc* e, uint32_t n, uint64_t* buffer, int begin) {  /* RLEEncoding.cpp:48-68 */
  uint32_t left = n;
  while (left > 0) {
    if (e->copies_remaining == 0) rle_advance(e);
    if (left < e->copies_remaining) {
      fill_bits(buffer, (uint64_t)begin, (uint64_t)begin + left, (int)e->value);
      e->copies_remaining -= left; return;
    }
    fill_bits(buffer, (uint64_t)begin, (uint64_t)begin + e->copies_remaining, (int)e->value);
    begin += (int)e->copies_remaining; left -= e->copies_remaining; e->copies_remaining = 0;
  }
}
static void rle_init(orc_enc* e) {
  orc_ctx* c = e->ctx;
  ORC_CHECK(c, (uint64_t)e->prefix + 4 <= e->size, "RLE header beyond end");
  const uint8_t* p = e->data + e->prefix;
  const uint32_t rl_size = ld32(p);
  ORC_CHECK(c, (uint64_t)e->prefix + 4 + rl_size <= e->size, "RLE run lengths beyond end");
  e->child[0] = orc_create_in(c, p + 4, rl_size);
  ORC_CHECK(c, e->child[0]->es == 4 && e->child[0]->dtype != DT_FLOAT, "RLE run lengths must be uint32");
  const uint8_t* values = p + 4 + rl_size;
  if (e->dtype == DT_BOOL) {  /* RLEEncoding.cpp:20-37 */
    ORC_CHECK(c, values + 1 == e->data + e->size, "Unexpected run length encoding end");
    e->initial_bool = values[0] != 0;
    e->bools_as_bits = rle_bits;
  } else {  /* RLEEncoding.h:816-837 */
    ORC_CHECK(c, values < e->data + e->size, "RLE run values missing");
    e->child[1] = orc_create_in(c, values, (uint64_t)(e->data + e->size - values));
    ORC_CHECK(c, e->child[1]->dtype == e->dtype, "RLE run values type mismatch");
    e->bufv = (uint8_t*)orc_xmalloc(c, (size_t)128 * e->child[1]->es);
  }
  e->reset = rle_reset; e->skip = rle_skip; e->materialize = rle_materialize;
  rle_reset(e);
}

/* ---- Dictionary ----------------------------------------------------------- */
static void dict_reset(orc_enc* e) { e->child[1]->reset(e->child[1]); }
static void dict_skip(orc_enc* e, uint32_t n) { e->child[1]->skip(e->child[1], n); }
static void dict_materialize(orc_enc* e, uint32_t n, void* buffer) {  /* DictionaryEncoding.h:227-236 */
  uint32_t* idx = (uint32_t*)orc_xmalloc(e->ctx, (size_t)n * 4);
  e->child[1]->materialize(e->child[1], n, idx);
  uint8_t* out = (uint8_t*)buffer;
  for (uint32_t i = 0; i < n; ++i) {
    if (idx[i] >= e->alphabet_count) { free(idx); orc_fail(e->ctx, "dictionary index %u out of range", idx[i]); }
    memcpy(out + (uint64_t)i * e->es, e->alphabet + (uint64_t)idx[i] * e->es, (size_t)e->es);
  }
  free(idx);
}
static void dict_init(orc_enc* e) {  /* DictionaryEncoding.h:193-214 */
  orc_ctx* c = e->ctx;
  ORC_CHECK(c, e->dtype != DT_BOOL, "Dictionary does not support bool");
  ORC_CHECK(c, (uint64_t)e->prefix + 4 <= e->size, "Dictionary header beyond end");
  const uint8_t* p = e->data + e->prefix;
  const uint32_t alpha_size = ld32(p);
  ORC_CHECK(c, (uint64_t)e->prefix + 4 + alpha_size < e->size, "Dictionary alphabet beyond end");
  e->child[0] = orc_create_in(c, p + 4, alpha_size);
  ORC_CHECK(c, e->child[0]->dtype == e->dtype, "Dictionary alphabet type mismatch");
  e->alphabet_count = e->child[0]->rows;
  e->alphabet = (uint8_t*)orc_xmalloc(c, (size_t)e->alphabet_count * e->es);
  e->child[0]->materialize(e->child[0], e->alphabet_count, e->alphabet);
  const uint8_t* ip = p + 4 + alpha_size;
  e->child[1] = orc_create_in(c, ip, (uint64_t)(e->data + e->size - ip));
  ORC_CHECK(c, e->child[1]->es == 4 && e->child[1]->dtype != DT_FLOAT, "Dictionary indices must be uint32");
  e->reset = dict_reset; e->skip = dict_skip; e->materialize = dict_materialize;
}

/* ---- Nullable -------------------------------------------------------------- */
static uint32_t count_true(const uint8_t* b, uint32_t n) { uint32_t s = 0; for (uint32_t i = 0; i < n; ++i) s += b[i]; return s; }
static void nullable_reset(orc_enc* e) { e->row = 0; e->child[0]->reset(e->child[0]); e->child[1]->reset(e->child[1]); }
static void nullable_skip(orc_enc* e, uint32_t n) {  /* NullableEncoding.h:196-204 */
  uint8_t* nb = (uint8_t*)orc_xmalloc(e->ctx, n);
  e->child[1]->materialize(e->child[1], n, nb);
  uint32_t nn = count_true(nb, n);
  free(nb);
  e->child[0]->skip(e->child[0], nn);
}
static void nullable_materialize(orc_enc* e, uint32_t n, void* buffer) {  /* :207-236 */
  uint8_t* nb = (uint8_t*)orc_xmalloc(e->ctx, n);
  e->child[1]->materialize(e->child[1], n, nb);
  const uint32_t nn = count_true(nb, n);
  e->child[0]->materialize(e->child[0], nn, buffer);
  if (nn != n) {
    const int es = e->es;
    uint8_t* base = (uint8_t*)buffer;
    int64_t out = (int64_t)n - 1, last = (int64_t)nn - 1;
    uint32_t pos = n - 1;
    while (out != last) {
      if (nb[pos]) { memmove(base + out * es, base + last * es, (size_t)es); --last; }
      else memset(base + out * es, 0, (size_t)es);      /* physicalType() */
      --out; --pos;
    }
  }
  free(nb);
  e->row += n;
}
static void nullable_init(orc_enc* e) {  /* :151-171 */
  orc_ctx* c = e->ctx;
  ORC_CHECK(c, (uint64_t)e->prefix + 4 <= e->size, "Nullable header beyond end");
  const uint8_t* p = e->data + e->prefix;
  const uint32_t nn_size = ld32(p);
  ORC_CHECK(c, (uint64_t)e->prefix + 4 + nn_size < e->size, "Nullable data beyond end");
  e->child[0] = orc_create_in(c, p + 4, nn_size);
  const uint8_t* np = p + 4 + nn_size;
  e->child[1] = orc_create_in(c, np, (uint64_t)(e->data + e->size - np));
  ORC_CHECK(c, e->child[0]->dtype == e->dtype, "Nullable data type mismatch");
  ORC_CHECK(c, e->child[1]->dtype == DT_BOOL, "Nullable nulls must be bool");
  ORC_CHECK(c, e->child[1]->rows == e->rows, "Nulls count mismatch.");
  e->nullable = 1;
  e->reset = nullable_reset; e->skip = nullable_skip; e->materialize = nullable_materialize;
}

/* ---- SparseBool ------------------------------------------------------------ */
static void sparse_reset(orc_enc* e) { e->row = 0; buf32_reset(e); e->next_index = buf32_next(e); }
static void sparse_skip(orc_enc* e, uint32_t n) {
  const uint32_t end = e->row + n;
  while (e->next_index < end) e->next_index = buf32_next(e);
  e->row = end;
}
static void sparse_materialize(orc_enc* e, uint32_t n, void* buffer) {  /* SparseBoolEncoding.cpp:100-119 */
  if (n == 0) return;
  const uint32_t end = e->row + n;
  uint8_t* out = (uint8_t*)buffer;
  memset(out, e->sparse_value ? 0 : 1, n);
  while (e->next_index < end) { out[e->next_index - e->row] = (uint8_t)(e->sparse_value ? 1 : 0); e->next_index = buf32_next(e); }
  e->row = end;
}
static void sparse_bits(orc_enc* e, uint32_t n, uint64_t* buffer, int begin) {  /* :121-142 */
  if (n == 0) return;
  fill_bits(buffer, (uint64_t)begin, (uint64_t)begin + n, !e->sparse_value);
  const uint32_t end = e->row + n;
  while (e->next_index < end) {
    if (e->sparse_value) bit_set(buffer, (uint64_t)begin + e->next_index - e->row);
    else bit_clear(buffer, (uint64_t)begin + e->next_index - e->row);
    e->next_index = buf32_next(e);
  }
  e->row = end;
}
static void sparse_init(orc_enc* e) {  /* :70-84 */
  orc_ctx* c = e->ctx;
  ORC_CHECK(c, e->dtype == DT_BOOL, "Trying to deserialize a SparseBoolEncoding with a non-bool data type.");
  ORC_CHECK(c, (uint64_t)e->prefix + 1 < e->size, "SparseBool header beyond end");
  e->sparse_value = e->data[e->prefix] != 0;
  e->child[0] = orc_create_in(c, e->data + e->prefix + 1, e->size - e->prefix - 1);
  ORC_CHECK(c, e->child[0]->es == 4 && e->child[0]->dtype != DT_FLOAT, "SparseBool indices must be uint32");
  ORC_CHECK(c, e->child[0]->rows >= 1, "SparseBool indices need the trailing sentinel");
  e->reset = sparse_reset; e->skip = sparse_skip; e->materialize = sparse_materialize; e->bools_as_bits = sparse_bits;
  sparse_reset(e);
}

/* ---- Constant -------------------------------------------------------------- */
static void const_reset(orc_enc* e) { (void)e; }
static void const_skip(orc_enc* e, uint32_t n) { (void)e; (void)n; }
static void const_materialize(orc_enc* e, uint32_t n, void* buffer) { rle_fill(e, (uint8_t*)buffer, n); }
static void const_bits(orc_enc* e, uint32_t n, uint64_t* buffer, int begin) {  /* ConstantEncoding.h:322-325 */
  fill_bits(buffer, (uint64_t)begin, (uint64_t)begin + n, (int)e->value);
}
static const uint8_t* read_value(orc_enc* e, const uint8_t* p) {
  orc_ctx* c = e->ctx;
  if (e->dtype == DT_STRING) {  /* encoding::readString + copy into factory memory */
    ORC_CHECK(c, p + 4 <= e->data + e->size, "value beyond end");
    uint32_t len = ld32(p);
    ORC_CHECK(c, p + 4 + len <= e->data + e->size, "value beyond end");
    char* copy = (char*)orc_string_buffer(c, len);
    memcpy(copy, p + 4, len);
    e->svalue.ptr = copy; e->svalue.len = len;
    return p + 4 + len;
  }
  ORC_CHECK(c, p + e->es <= e->data + e->size, "value beyond end");
  e->value = ld_le(p, e->es);
  if (e->dtype == DT_BOOL) e->value = e->value != 0;
  return p + e->es;
}
static void const_init(orc_enc* e) {  /* ConstantEncoding.h:292-302 */
  const uint8_t* end = read_value(e, e->data + e->prefix);
  ORC_CHECK(e->ctx, end == e->data + e->size, "Unexpected constant encoding end");
  e->reset = const_reset; e->skip = const_skip; e->materialize = const_materialize;
  if (e->dtype == DT_BOOL) e->bools_as_bits = const_bits;
}

/* ---- MainlyConstant -------------------------------------------------------- */
static void mc_reset(orc_enc* e) { e->child[0]->reset(e->child[0]); e->child[1]->reset(e->child[1]); }
static uint32_t mc_uncommon(orc_enc* e, uint32_t n, uint8_t** is_common_out) {
  uint8_t* ic = (uint8_t*)orc_xmalloc(e->ctx, n ? n : 1);
  e->child[0]->materialize(e->child[0], n, ic);
  uint32_t others = n - count_true(ic, n);
  *is_common_out = ic;
  return others;
}
static void mc_skip(orc_enc* e, uint32_t n) {  /* MainlyConstantEncoding.h:178-206 */
  uint8_t* ic; uint32_t others = mc_uncommon(e, n, &ic); free(ic);
  if (others) e->child[1]->skip(e->child[1], others);
}
static void mc_materialize(orc_enc* e, uint32_t n, void* buffer) {  /* :208-253, :768-787 (same result) */
  uint8_t* ic; const uint32_t others = mc_uncommon(e, n, &ic);
  rle_fill(e, (uint8_t*)buffer, n);
  if (others) {
    uint8_t* ov = (uint8_t*)orc_xmalloc(e->ctx, (size_t)others * e->es);
    e->child[1]->materialize(e->child[1], others, ov);
    uint32_t k = 0;
    for (uint32_t i = 0; i < n; ++i) if (!ic[i]) { memcpy((uint8_t*)buffer + (uint64_t)i * e->es, ov + (uint64_t)k * e->es, (size_t)e->es); ++k; }
    free(ov);
  }
  free(ic);
}
static void mc_init(orc_enc* e) {  /* :842-860 */
  orc_ctx* c = e->ctx;
  ORC_CHECK(c, e->dtype != DT_BOOL, "MainlyConstant does not support bool");
  const uint8_t* p = e->data + e->prefix; const uint8_t* end = e->data + e->size;
  ORC_CHECK(c, p + 4 <= end, "MainlyConstant header beyond end");
  const uint32_t ic_size = ld32(p);
  ORC_CHECK(c, p + 4 + (uint64_t)ic_size + 4 <= end, "MainlyConstant isCommon beyond end");
  e->child[0] = orc_create_in(c, p + 4, ic_size);
  p += 4 + (uint64_t)ic_size;
  const uint32_t ov_size = ld32(p);
  ORC_CHECK(c, p + 4 + (uint64_t)ov_size <= end, "MainlyConstant otherValues beyond end");
  e->child[1] = orc_create_in(c, p + 4, ov_size);
  p += 4 + (uint64_t)ov_size;
  ORC_CHECK(c, e->child[0]->dtype == DT_BOOL && e->child[0]->rows == e->rows, "MainlyConstant isCommon mismatch");
  ORC_CHECK(c, e->child[1]->dtype == e->dtype, "MainlyConstant otherValues type mismatch");
  const uint8_t* vend = read_value(e, p);
  ORC_CHECK(c, vend == end, "Unexpected mainly constant encoding end");
  e->reset = mc_reset; e->skip = mc_skip; e->materialize = mc_materialize;
}

/* ---- Varint ----------------------------------------------------------------- */
static void varint_reset(orc_enc* e) { e->row = 0; e->pos = e->payload; }
static void varint_skip(orc_enc* e, uint32_t n) {
  for (uint32_t i = 0; i < n; ++i) { while (*(e->pos++) & 128) {} }
  e->row += n;
}
static void varint_materialize(orc_enc* e, uint32_t n, void* buffer) {  /* VarintEncoding.h:119-140 */
  if (e->es == 4) { uint32_t* o = (uint32_t*)buffer; for (uint32_t i = 0; i < n; ++i) o[i] = read_varint32(&e->pos) + (uint32_t)e->baseline; }
  else { uint64_t* o = (uint64_t*)buffer; for (uint32_t i = 0; i < n; ++i) o[i] = read_varint64(&e->pos) + e->baseline; }
  e->row += n;
}
static void varint_init(orc_enc* e) {
  orc_ctx* c = e->ctx;
  ORC_CHECK(c, (e->es == 4 || e->es == 8) && e->dtype != DT_STRING, "Varint needs a 4- or 8-byte type");
  ORC_CHECK(c, (uint64_t)e->prefix + e->es <= e->size, "Varint header beyond end");
  e->baseline = ld_le(e->data + e->prefix, e->es);
  e->payload = e->data + e->prefix + e->es;
  /* bound check the oracle adds: `rows` terminator bytes must exist */
  uint64_t terms = 0; for (const uint8_t* p = e->payload; p < e->data + e->size; ++p) terms += !(*p & 128);
  ORC_CHECK(c, terms >= e->rows, "Varint payload truncated");
  e->reset = varint_reset; e->skip = varint_skip; e->materialize = varint_materialize;
  varint_reset(e);
}

/* ---- BlockBitPacking / FOR ---------------------------------------------------- */
/* `w` bits at bit position `bit` of an LSB-first stream, read byte-wise inside [p, p+bytes). */
static uint64_t bits_at(orc_ctx* c, const uint8_t* p, uint64_t bytes, uint64_t bit, int w) {
  if (w == 0) return 0;
  ORC_CHECK(c, (bit + (uint64_t)w + 7) / 8 <= bytes, "packed segment beyond end of the payload");
  uint64_t v = 0; int got = 0; uint64_t byte = bit >> 3; int sh = (int)(bit & 7);
  while (got < w) {
    const uint64_t chunk = (uint64_t)(p[byte] >> sh);
    v |= got < 64 ? chunk << got : 0;
    got += 8 - sh; sh = 0; ++byte;
  }
  return w == 64 ? v : v & ((1ull << w) - 1);
}
static uint32_t seg_index(const orc_enc* e, uint32_t row) {   /* blockIndex / frameIndex */
  if (row < e->seg_first) return 0;
  return 1 + (row - e->seg_first) / e->seg_size;
}
static uint32_t seg_row_offset(const orc_enc* e, uint32_t s) { return s == 0 ? 0 : e->seg_first + (s - 1) * e->seg_size; }
/* varint header field; the oracle adds the bound the reference leaves out */
static uint32_t header_varint(orc_enc* e, const uint8_t** p) {
  const uint8_t* end = e->data + e->size;
  const uint8_t* q = *p;
  int n = 0;
  while (q < end && n < 5 && (*q & 128)) { ++q; ++n; }
  ORC_CHECK(e->ctx, q < end, "header field beyond end of encoding");
  return read_varint32(p);
}
static void seg_reset(orc_enc* e) { e->row = 0; }
static void seg_skip(orc_enc* e, uint32_t n) { e->row += n; }
static void seg_materialize(orc_enc* e, uint32_t n, void* buffer) {
  /* BlockBitPackingEncoding.h:746-784,856-874; ForEncoding.h:398-529 */
  orc_ctx* c = e->ctx;
  uint8_t* out = (uint8_t*)buffer;
  ORC_CHECK(c, (uint64_t)e->row + n <= e->rows, "Decoding past end of encoding");
  for (uint32_t i = 0; i < n; ++i) {
    const uint32_t row = e->row + i;
    const uint32_t s = seg_index(e, row);
    ORC_CHECK(c, s < e->seg_count, "row beyond the last block");
    const uint32_t in_seg = row - seg_row_offset(e, s);
    const int w = e->seg_width[s];
    uint64_t base = 0; memcpy(&base, e->seg_base + (size_t)s * e->es, (size_t)e->es);
    uint64_t v;
    if (e->etype == ET_BLOCKBITPACKING) {
      if (w == 255) {   /* raw block */
        ORC_CHECK(c, e->seg_off[s] + ((uint64_t)in_seg + 1) * e->es <= e->payload_bytes, "raw block beyond end of the payload");
        v = ld_le(e->payload + e->seg_off[s] + (uint64_t)in_seg * e->es, e->es);
      } else {
        ORC_CHECK(c, w < 8 * e->es, "block bit width %d too wide for a %d-byte type", w, e->es);
        ORC_CHECK(c, e->seg_off[s] <= e->payload_bytes, "block offset beyond end of the payload");
        v = bits_at(c, e->payload + e->seg_off[s], e->payload_bytes - e->seg_off[s], (uint64_t)in_seg * (uint64_t)w, w) + base;
      }
    } else {
      ORC_CHECK(c, w <= 64, "frame bit width %d", w);
      v = bits_at(c, e->payload, e->payload_bytes, e->seg_off[s] + (uint64_t)in_seg * (uint64_t)w, w) + base;
    }
    st_le(out + (uint64_t)i * e->es, v, e->es);
  }
  e->row += n;
}
static const uint8_t* seg_child(orc_enc* e, const uint8_t* p, uint32_t rows, int es, void* dst) {
  /* varint size + nested encoding, materialised whole (readMetadataStream, BlockBitPackingEncoding.h:698-710) */
  orc_ctx* c = e->ctx;
  const uint8_t* end = e->data + e->size;
  const uint32_t size = header_varint(e, &p);
  ORC_CHECK(c, p <= end && size <= (uint64_t)(end - p), "metadata stream beyond end");
  orc_enc* ch = orc_create_in(c, p, size);
  ORC_CHECK(c, ch->es == es && ch->dtype != DT_BOOL && ch->dtype != DT_STRING, "metadata stream has the wrong element size");
  ORC_CHECK(c, ch->rows >= rows, "metadata stream shorter than the block count");
  ch->materialize(ch, rows, dst);
  return p + size;
}
static void seg_alloc(orc_enc* e) {
  e->seg_width = (uint8_t*)orc_xmalloc(e->ctx, e->seg_count);
  e->seg_base = (uint8_t*)orc_xmalloc(e->ctx, (size_t)e->seg_count * e->es);
  e->seg_off = (uint64_t*)orc_xmalloc(e->ctx, (size_t)e->seg_count * 8);
}
static void seg_geometry(orc_enc* e, const char* what) {
  orc_ctx* c = e->ctx;
  ORC_CHECK(c, e->seg_size > 0 && e->seg_count > 0 && e->rows > 0, "%s needs rows, a block size and blocks", what);
  ORC_CHECK(c, e->seg_first > 0 && e->seg_first <= e->seg_size && e->seg_first <= e->rows, "%s first block rows out of range", what);
  const uint64_t expect = e->rows <= e->seg_first ? 1 : 1 + ((uint64_t)(e->rows - e->seg_first) + e->seg_size - 1) / e->seg_size;
  ORC_CHECK(c, e->seg_count == expect, "%s block count mismatch.", what);
  e->seg_last = e->seg_count == 1 ? e->seg_first : e->rows - e->seg_first - (e->seg_count - 2) * e->seg_size;
}
static void bbp_init(orc_enc* e) {  /* BlockBitPackingEncoding.h:311-348,660-696 */
  orc_ctx* c = e->ctx;
  ORC_CHECK(c, e->dtype != DT_BOOL && e->dtype != DT_STRING, "BlockBitPacking needs a numeric type");
  const uint8_t* p = e->data + e->prefix; const uint8_t* end = e->data + e->size;
  ORC_CHECK(c, p + 3 <= end, "BlockBitPacking header beyond end");
  ORC_CHECK(c, p[0] == 0, "compressed BlockBitPacking payload not supported by the oracle");
  ++p;
  e->seg_size = (uint16_t)header_varint(e, &p);
  e->seg_count = (uint16_t)header_varint(e, &p);
  ORC_CHECK(c, e->seg_size > 0 && e->seg_size <= 1024 && e->seg_count > 0 && e->rows > 0, "BlockBitPacking block size / count out of range");
  seg_alloc(e);
  uint32_t* off32 = (uint32_t*)orc_xmalloc(c, (size_t)e->seg_count * 4);
  p = seg_child(e, p, e->seg_count, e->es, e->seg_base);
  p = seg_child(e, p, e->seg_count, 1, e->seg_width);
  p = seg_child(e, p, e->seg_count, 4, off32);
  for (uint32_t i = 0; i < e->seg_count; ++i) e->seg_off[i] = off32[i];
  free(off32);
  e->seg_first = header_varint(e, &p);
  seg_geometry(e, "BlockBitPacking");
  e->payload = p; e->payload_bytes = (uint64_t)(end - p);
  e->reset = seg_reset; e->skip = seg_skip; e->materialize = seg_materialize;
}
static void for_init(orc_enc* e) {  /* ForEncoding.h:174-196,316-383 */
  orc_ctx* c = e->ctx;
  ORC_CHECK(c, e->dtype >= DT_INT8 && e->dtype <= DT_UINT64, "FOR needs an integer type");
  const uint8_t* p = e->data + e->prefix; const uint8_t* end = e->data + e->size;
  ORC_CHECK(c, p + 4 <= end, "FOR header beyond end");
  ORC_CHECK(c, p[0] == 0, "compressed FOR payload not supported by the oracle");
  ++p;
  e->seg_size = header_varint(e, &p);
  e->seg_count = header_varint(e, &p);
  e->seg_first = header_varint(e, &p);
  seg_geometry(e, "FOR");
  seg_alloc(e);
  p = seg_child(e, p, e->seg_count, 1, e->seg_width);
  p = seg_child(e, p, e->seg_count, e->es, e->seg_base);
  p = seg_child(e, p, e->seg_count, 8, e->seg_off);
  const uint32_t packed = header_varint(e, &p);
  ORC_CHECK(c, p <= end && packed <= (uint64_t)(end - p), "FOR packed data beyond end");
  e->payload = p; e->payload_bytes = packed;
  e->reset = seg_reset; e->skip = seg_skip; e->materialize = seg_materialize;
}

/* ---- factory ----------------------------------------------------------------- */
static void ctx_release(orc_ctx* c) {
  for (size_t i = 0; i < c->nencs; ++i) {
    free(c->encs[i]->bufv); free(c->encs[i]->alphabet);
    free(c->encs[i]->seg_width); free(c->encs[i]->seg_base); free(c->encs[i]->seg_off);
    free(c->encs[i]);
  }
  for (size_t i = 0; i < c->nallocs; ++i) free(c->allocs[i]);
  free(c->encs); free(c->allocs);
  c->encs = NULL; c->allocs = NULL; c->nencs = c->nallocs = 0;
}
static orc_enc* orc_create_in(orc_ctx* c, const uint8_t* data, uint64_t size) {
  orc_enc* e = enc_alloc(c, data, size);
  switch (e->etype) {  /* EncodingFactory.cpp:71-163 */
    case ET_TRIVIAL: trivial_init(e); break;
    case ET_RLE: rle_init(e); break;
    case ET_DICTIONARY: dict_init(e); break;
    case ET_FIXEDBITWIDTH: fbw_init(e); break;
    case ET_NULLABLE: nullable_init(e); break;
    case ET_SPARSEBOOL: sparse_init(e); break;
    case ET_VARINT: varint_init(e); break;
    case ET_CONSTANT: const_init(e); break;
    case ET_MAINLYCONSTANT: mc_init(e); break;
    case ET_BLOCKBITPACKING: bbp_init(e); break;
    case ET_FOR: for_init(e); break;
    default: {
      orc_fail(c, "Trying to deserialize invalid EncodingType:%d -- garbage input?", (int)e->etype);
    }
  }
  return e;
}

/* ======================== public C API (ctypes) =============================== */
typedef struct orc_handle { orc_ctx ctx; orc_enc* root; } orc_handle;

orc_handle* orc_create(const uint8_t* data, uint64_t size, int varint_rows, char* err, int errlen) {
  orc_handle* volatile h = (orc_handle*)calloc(1, sizeof(orc_handle));
  h->ctx.varint_rows = varint_rows;
  if (setjmp(h->ctx.jb)) {
    if (err && errlen > 0) { snprintf(err, (size_t)errlen, "%s", h->ctx.err); }
    ctx_release(&h->ctx); free(h);
    return NULL;
  }
  h->root = orc_create_in(&h->ctx, data, size);
  return h;
}
void orc_destroy(orc_handle* h) {
  if (!h) return;
  ctx_release(&h->ctx); free(h);
}
void orc_set_bulk(orc_handle* h, int bulk) { h->ctx.bulk = bulk; }  /* FixedBitWidth through the bulk path from now on */
uint32_t orc_row_count(const orc_handle* h) { return h->root->rows; }
int orc_encoding_type(const orc_handle* h) { return h->root->etype; }
int orc_data_type(const orc_handle* h) { return h->root->dtype; }
int orc_is_nullable(const orc_handle* h) { return h->root->nullable; }
uint32_t orc_null_count(const orc_handle* h) {  /* NullableEncoding.h:174-176 */
  return h->root->nullable ? h->root->child[1]->rows - h->root->child[0]->rows : 0;
}
const char* orc_last_error(const orc_handle* h) { return h->ctx.err; }

#define ORC_GUARD(h, failret) if (setjmp((h)->ctx.jb)) return failret
int orc_reset(orc_handle* h) { ORC_GUARD(h, -1); h->root->reset(h->root); return 0; }
int orc_skip(orc_handle* h, uint32_t n) { ORC_GUARD(h, -1); h->root->skip(h->root, n); return 0; }
/* numeric: physicalType[n]; bool: one byte per row; string: orc_sv[n] */
int orc_materialize(orc_handle* h, uint32_t n, void* buffer) { ORC_GUARD(h, -1); h->root->materialize(h->root, n, buffer); return 0; }
int orc_materialize_bools_as_bits(orc_handle* h, uint32_t n, uint64_t* buffer, int begin) {
  ORC_GUARD(h, -1);
  ORC_CHECK(&h->ctx, h->root->bools_as_bits != NULL, "materializeBoolsAsBits not implemented");
  h->root->bools_as_bits(h->root, n, buffer, begin);
  return 0;
}

/* EncodingView::read (encodings/views/EncodingView.h:38-41,92-95): rows
 * [offset, offset+length) of the stream, independent of any cursor. */
int orc_view_read(orc_handle* h, uint32_t offset, uint32_t length, void* buffer) {
  ORC_GUARD(h, -1);
  orc_enc* e = h->root;
  ORC_CHECK(&h->ctx, offset <= e->rows && length <= e->rows - offset, "read range beyond the row count");
  e->reset(e);
  e->skip(e, offset);
  e->materialize(e, length, buffer);
  return 0;
}

/* Encoding::materializeNullable.  `nulls` is the caller's bitmap (1 = non-null);
 * *nulls_requested is set when the reference would have called getOutputNulls().
 * `scatter` (may be NULL) is a bitmap of `scatter_size` bits.  Returns the
 * non-null count, or UINT32_MAX on error. */
static uint32_t typed_materialize_nullable(orc_enc* e, uint32_t n, void* buffer, void* nulls, int* nulls_requested,
                                           const void* scatter, uint32_t scatter_size, uint32_t offset) {
  /* encodings/common/Encoding.h:412-460 */
  const int es = e->es;
  uint8_t* base = (uint8_t*)buffer + (uint64_t)offset * es;
  e->materialize(e, n, base);
  const uint32_t scatter_count = scatter ? scatter_size - offset : n;
  if (scatter_count == n) return n;
  ORC_CHECK(e->ctx, n < scatter_count, "Unexpected count");
  *nulls_requested = 1;
  copy_bits(scatter, offset, nulls, offset, scatter_count);   /* BitmapBuilder::copy(scatter, offset, offset+count) */
  uint32_t pos = offset + scatter_count - 1;
  int64_t out = (int64_t)scatter_count - 1, src = (int64_t)n - 1;
  while (out != src) {
    if (bit_is_set(scatter, pos)) { memmove(base + out * es, base + src * es, (size_t)es); --src; }
    --out; --pos;
  }
  return n;
}
static uint32_t nullable_materialize_nullable(orc_enc* e, uint32_t n, void* buffer, void* nulls, int* nulls_requested,
                                              const void* scatter, uint32_t scatter_size, uint32_t offset) {
  /* encodings/NullableEncoding.h:239-306 */
  const int es = e->es;
  uint8_t* nb = (uint8_t*)orc_xmalloc(e->ctx, n ? n : 1);
  e->child[1]->materialize(e->child[1], n, nb);
  const uint32_t nn = count_true(nb, n);
  uint8_t* base = (uint8_t*)buffer + (uint64_t)offset * es;
  e->child[0]->materialize(e->child[0], nn, base);
  const uint32_t scatter_count = scatter ? scatter_size - offset : n;
  if (scatter_count < n) { free(nb); orc_fail(e->ctx, "Scattered output must have at least rowCount positions"); }
  if (nn != scatter_count) {
    *nulls_requested = 1;
    fill_bits(nulls, offset, (uint64_t)offset + scatter_count, 0);
    int64_t pos = (int64_t)offset + scatter_count - 1;
    int64_t out = (int64_t)scatter_count - 1, last = (int64_t)nn - 1;
    int64_t it = (int64_t)n - 1;
    if (scatter_count != n) {
      while (out != last) {
        if (bit_is_set(scatter, (uint64_t)pos)) {
          if (nb[it--]) { bit_set(nulls, (uint64_t)pos); memmove(base + out * es, base + last * es, (size_t)es); --last; }
        }
        --out; --pos;
      }
    } else {
      while (out != last) {
        if (nb[it--]) { bit_set(nulls, (uint64_t)pos); memmove(base + out * es, base + last * es, (size_t)es); --last; }
        --out; --pos;
      }
    }
    if (out >= 0) fill_bits(nulls, offset, (uint64_t)pos + 1, 1);
  }
  free(nb);
  e->row += n;
  return nn;
}
uint32_t orc_materialize_nullable(orc_handle* h, uint32_t n, void* buffer, void* nulls, int* nulls_requested,
                                  const void* scatter, uint32_t scatter_size, uint32_t offset) {
  int dummy = 0;
  int* volatile requested_out = nulls_requested ? nulls_requested : &dummy;
  ORC_GUARD(h, UINT32_MAX);
  orc_enc* e = h->root;
  if (e->etype == ET_NULLABLE)
    return nullable_materialize_nullable(e, n, buffer, nulls, requested_out, scatter, scatter_size, offset);
  return typed_materialize_nullable(e, n, buffer, nulls, requested_out, scatter, scatter_size, offset);
}

/* Flatten orc_sv[n] into lengths + bytes (helper for ctypes callers). */
uint64_t orc_sv_total_bytes(const orc_sv* sv, uint32_t n) { uint64_t t = 0; for (uint32_t i = 0; i < n; ++i) t += sv[i].len; return t; }
void orc_sv_flatten(const orc_sv* sv, uint32_t n, uint32_t* lengths, uint8_t* bytes) {
  uint64_t off = 0;
  for (uint32_t i = 0; i < n; ++i) { lengths[i] = (uint32_t)sv[i].len; if (sv[i].len) memcpy(bytes + off, sv[i].ptr, sv[i].len); off += sv[i].len; }
}

/* ---- chunk framing + ChunkedStreamDecoder ---------------------------------- */
/* Splits a stream into chunks (ChunkedStream.cpp:36-72).  Returns the number of
 * chunks, or -1 on framing error / compressed chunk.  offsets/sizes may be NULL
 * to only count. */
int64_t orc_split_stream(const uint8_t* stream, uint64_t size, uint64_t* offsets, uint64_t* sizes, int64_t cap) {
  uint64_t pos = 0; int64_t n = 0;
  while (pos < size) {
    if (size - pos < 5) return -1;                      /* "Read beyond end of stream" */
    uint32_t len = ld32(stream + pos); uint8_t comp = stream[pos + 4];
    pos += 5;
    if (len > size - pos) return -1;
    if (comp != 0) return -2;                           /* oracle handles uncompressed chunks only */
    if (offsets && n < cap) { offsets[n] = pos; sizes[n] = len; }
    ++n; pos += len;
  }
  return n;
}

typedef struct orc_stream_decoder {
  const uint8_t* stream; uint64_t size; uint64_t pos; int varint_rows;
  orc_handle* enc; uint32_t remaining;
  char err[256];
} orc_stream_decoder;

orc_stream_decoder* orc_sd_create(const uint8_t* stream, uint64_t size, int varint_rows) {
  orc_stream_decoder* d = (orc_stream_decoder*)calloc(1, sizeof *d);
  d->stream = stream; d->size = size; d->varint_rows = varint_rows;
  return d;
}
void orc_sd_destroy(orc_stream_decoder* d) { if (d) { orc_destroy(d->enc); free(d); } }
const char* orc_sd_last_error(const orc_stream_decoder* d) { return d->err; }
void orc_sd_reset(orc_stream_decoder* d) { d->pos = 0; d->remaining = 0; }  /* ChunkedStreamDecoder.cpp:188-191 */
static int sd_ensure_loaded(orc_stream_decoder* d) {                        /* :193-205 */
  if (d->remaining != 0) return 0;
  orc_destroy(d->enc); d->enc = NULL;
  if (d->size - d->pos < 5) { snprintf(d->err, sizeof d->err, "Read beyond end of stream"); return -1; }
  uint32_t len = ld32(d->stream + d->pos); uint8_t comp = d->stream[d->pos + 4];
  d->pos += 5;
  if (len > d->size - d->pos) { snprintf(d->err, sizeof d->err, "Read beyond end of stream"); return -1; }
  if (comp != 0) { snprintf(d->err, sizeof d->err, "compressed chunk not supported by the oracle"); return -1; }
  d->enc = orc_create(d->stream + d->pos, len, d->varint_rows, d->err, (int)sizeof d->err);
  d->pos += len;
  if (!d->enc) return -1;
  d->remaining = d->enc->root->rows;
  if (d->remaining == 0) { snprintf(d->err, sizeof d->err, "Empty chunk"); return -1; }
  return 0;
}
int orc_sd_skip(orc_stream_decoder* d, uint32_t count) {                     /* :178-186 */
  while (count > 0) {
    if (sd_ensure_loaded(d)) return -1;
    uint32_t n = count < d->remaining ? count : d->remaining;
    if (orc_skip(d->enc, n)) { snprintf(d->err, sizeof d->err, "%s", d->enc->ctx.err); return -1; }
    count -= n; d->remaining -= n;
  }
  return 0;
}
/* ChunkedStreamDecoder::next (numeric / bool streams; strings are stitched by the
 * caller from per-chunk views).  `nulls` may be NULL when the caller passes no
 * getOutputNulls.  Returns non-null count or UINT32_MAX. */
uint32_t orc_sd_next(orc_stream_decoder* d, uint32_t count, void* output, void* nulls, int* nulls_requested,
                     const void* scatter, uint32_t scatter_size) {
  int requested = 0;
  if (count == 0) {                                                          /* :83-90 */
    if (nulls && scatter) { memset(nulls, 0, ((uint64_t)scatter_size + 7) / 8); requested = 1; }
    if (nulls_requested) *nulls_requested = requested;
    return 0;
  }
  uint32_t non_null = 0, offset = 0; int has_nulls = 0;
  while (count > 0) {
    if (sd_ensure_loaded(d)) return UINT32_MAX;
    uint32_t rows = count < d->remaining ? count : d->remaining;
    uint32_t chunk_nn, end_offset; int req = 0;
    if (!nulls || !scatter) {
      if (scatter) { snprintf(d->err, sizeof d->err, "Unexpected scatter output bitmap"); return UINT32_MAX; }
      chunk_nn = orc_materialize_nullable(d->enc, rows, output, nulls, &req, NULL, 0, offset);
      end_offset = offset + rows;
    } else {
      end_offset = orc_find_set_bit(scatter, offset, scatter_size, rows + 1);
      chunk_nn = orc_materialize_nullable(d->enc, rows, output, nulls, &req, scatter, end_offset, offset);
    }
    if (chunk_nn == UINT32_MAX) { snprintf(d->err, sizeof d->err, "%s", d->enc->ctx.err); return UINT32_MAX; }
    requested |= req;
    int chunk_has_nulls = chunk_nn != (end_offset - offset);
    if (chunk_has_nulls && !has_nulls) fill_bits(nulls, 0, offset, 1);      /* back-fill :129-134 */
    has_nulls = has_nulls || chunk_has_nulls;
    if (has_nulls && !chunk_has_nulls) fill_bits(nulls, offset, end_offset, 1);   /* :136-140 */
    offset = end_offset; d->remaining -= rows; count -= rows; non_null += chunk_nn;
  }
  if (nulls_requested) *nulls_requested = requested;
  return non_null;
}

/* ---- selective reader's ChunkedDecoder: dense bulk entry points ----------------- */
/* The same chunk cursor as above (loadNextChunk, selective/ChunkedDecoder.cpp:33-85,
 * keeps one encoding and its remaining row count). */
static uint32_t count_set_bits(const void* bits, uint64_t begin, uint64_t end) {
  uint32_t n = 0; for (uint64_t i = begin; i < end; ++i) n += (uint32_t)bit_is_set(bits, i); return n;
}
/* velox::bits::scatterBits(numSource, numTarget, source, targetMask, target): source
 * bit k goes to the k-th set position of the mask, other positions become 0; works in place. */
static void scatter_bits(int64_t n_source, int64_t n_target, const void* source, const void* mask, void* target) {
  int64_t src = n_source - 1;
  for (int64_t pos = n_target - 1; pos >= 0; --pos) {
    if (bit_is_set(mask, (uint64_t)pos)) {
      const int b = src >= 0 ? bit_is_set(source, (uint64_t)src) : 0;
      --src;
      if (b) bit_set(target, (uint64_t)pos); else bit_clear(target, (uint64_t)pos);
    } else {
      bit_clear(target, (uint64_t)pos);
    }
  }
}
/* ChunkedDecoder::decodeNullable<T> (selective/ChunkedDecoder.h:206-268).  `nulls` may
 * be NULL when the caller knows the stream holds no nulls.  0 on success. */
int orc_cd_decode_nullable(orc_stream_decoder* d, void* nulls, void* data, int64_t count, const void* incoming) {
  const int64_t total = incoming ? (int64_t)count_set_bits(incoming, 0, (uint64_t)count) : count;
  if (incoming) {
    if (total == 0) { if (nulls) fill_bits(nulls, 0, (uint64_t)count, 0); return 0; }
    uint32_t offset = 0, end_offset = 0;
    for (int64_t i = 0; i < total;) {
      if (d->remaining == 0 && sd_ensure_loaded(d)) return -1;
      const int64_t n = total - i < (int64_t)d->remaining ? total - i : (int64_t)d->remaining;
      end_offset = orc_find_set_bit(incoming, offset, (uint32_t)count, (uint32_t)n + 1);
      int req = 0;
      const uint32_t nn = orc_materialize_nullable(d->enc, (uint32_t)n, data, nulls, &req, incoming, end_offset, offset);
      if (nn == UINT32_MAX) { snprintf(d->err, sizeof d->err, "%s", d->enc->ctx.err); return -1; }
      if (nulls && nn == end_offset - offset) fill_bits(nulls, offset, end_offset, 1);
      d->remaining -= (uint32_t)n; i += n; offset = end_offset;
    }
  } else {
    for (int64_t i = 0; i < total;) {
      if (d->remaining == 0 && sd_ensure_loaded(d)) return -1;
      const int64_t n = total - i < (int64_t)d->remaining ? total - i : (int64_t)d->remaining;
      int req = 0;
      const uint32_t nn = orc_materialize_nullable(d->enc, (uint32_t)n, data, nulls, &req, NULL, 0, (uint32_t)i);
      if (nn == UINT32_MAX) { snprintf(d->err, sizeof d->err, "%s", d->enc->ctx.err); return -1; }
      if (nulls && nn == (uint32_t)n) fill_bits(nulls, (uint64_t)i, (uint64_t)(i + n), 1);
      d->remaining -= (uint32_t)n; i += n;
    }
  }
  return 0;
}
/* ChunkedDecoder::nextBools (selective/ChunkedDecoder.cpp:155-191).  With
 * decode_values_with_nulls the stream is a nullable uint32 stream whose NULLS are
 * the result and whose values go to nullable_values[count] (getPreloadedValues). */
int orc_cd_next_bools(orc_stream_decoder* d, uint64_t* data, int64_t count, const void* incoming,
                      int decode_values_with_nulls, uint32_t* nullable_values) {
  if (decode_values_with_nulls) return orc_cd_decode_nullable(d, data, nullable_values, count, incoming);
  const int64_t total = incoming ? (int64_t)count_set_bits(incoming, 0, (uint64_t)count) : count;
  for (int64_t i = 0; i < total;) {
    if (d->remaining == 0) {
      if (sd_ensure_loaded(d)) return -1;
      if (d->enc->root->dtype != DT_BOOL) { snprintf(d->err, sizeof d->err, "nextBools on a non-bool stream"); return -1; }
    }
    const int64_t n = total - i < (int64_t)d->remaining ? total - i : (int64_t)d->remaining;
    if (orc_materialize_bools_as_bits(d->enc, (uint32_t)n, data, (int)i)) { snprintf(d->err, sizeof d->err, "%s", d->enc->ctx.err); return -1; }
    d->remaining -= (uint32_t)n; i += n;
  }
  if (incoming) scatter_bits(total, count, data, incoming, data);
  return 0;
}
/* ChunkedDecoder::nextIndices (:193-206): uint32 values, nulls discarded. */
int orc_cd_next_indices(orc_stream_decoder* d, int32_t* data, int64_t count, const void* incoming) {
  void* nulls = incoming ? calloc(((size_t)count + 63) / 64 + 1, 8) : NULL;
  const int rc = orc_cd_decode_nullable(d, nulls, data, count, incoming);
  free(nulls);
  return rc;
}
/* ChunkedDecoder::skipWithoutIndex (:270-286): whole chunks are dropped without touching their encoding. */
int orc_cd_skip(orc_stream_decoder* d, int64_t n) {
  while (n > 0) {
    if (d->remaining == 0 && sd_ensure_loaded(d)) return -1;
    if (n < (int64_t)d->remaining) {
      if (orc_skip(d->enc, (uint32_t)n)) { snprintf(d->err, sizeof d->err, "%s", d->enc->ctx.err); return -1; }
      d->remaining -= (uint32_t)n;
      break;
    }
    n -= d->remaining; d->remaining = 0;
  }
  return 0;
}

/* ---- whole-chunk decode used by the CPU baseline ---------------------------- */
/* Decodes every row of one encoding: values (numeric/bool bytes), optional
 * bitmap (ceil(rows/8) bytes, 1 = non-null, written only for Nullable roots,
 * null slots zero-filled as Encoding::materialize does).  Returns non-null
 * count or UINT32_MAX. */
static uint32_t decode_all(const uint8_t* data, uint64_t size, int varint_rows, void* values, void* nulls, int bulk);
uint32_t orc_decode_all(const uint8_t* data, uint64_t size, int varint_rows, void* values, void* nulls) {
  return decode_all(data, size, varint_rows, values, nulls, 0);
}
/* The same with FixedBitWidth reads going through the bulk path (CPU baseline timing). */
uint32_t orc_decode_all_fast(const uint8_t* data, uint64_t size, int varint_rows, void* values, void* nulls) {
  return decode_all(data, size, varint_rows, values, nulls, 1);
}
static uint32_t decode_all(const uint8_t* data, uint64_t size, int varint_rows, void* values, void* nulls, int bulk) {
  char err[256];
  orc_handle* h = orc_create(data, size, varint_rows, err, sizeof err);
  if (!h) return UINT32_MAX;
  h->ctx.bulk = bulk;
  uint32_t n = h->root->rows, nn = n;
  if (setjmp(h->ctx.jb)) { orc_destroy(h); return UINT32_MAX; }
  if (h->root->nullable && nulls) {
    orc_enc* nulls_enc = h->root->child[1];
    if (nulls_enc->bools_as_bits) { memset(nulls, 0, ((uint64_t)n + 7) / 8 + 8); nulls_enc->bools_as_bits(nulls_enc, n, (uint64_t*)nulls, 0); }
    nulls_enc->reset(nulls_enc);
    nn = h->root->child[0]->rows;
  }
  h->root->materialize(h->root, n, values);
  orc_destroy(h);
  return nn;
}

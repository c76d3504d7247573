// Host pre-pass for compressed payloads.
//
// The reference's default writer compresses leaf payloads (Trivial values /
// bitmaps / string blobs, FixedBitWidth packed bits) and, optionally, whole
// chunks with Zstd or LZ4 (dwio/nimble/compression/Compression.h:72-94;
// framing: u32 uncompressed size, then the codec's frame --
// ZstdCompressor.cpp:79-99, Lz4Compressor.cpp:60-75).  The GPU kernels read
// uncompressed payloads only, so a chunk with compressed leaves is rewritten on
// the host into the byte-identical encoding the writer would have produced with
// compression off: payloads are inflated and the child-size fields of the
// enclosing encodings are patched.  The result goes straight into the staging
// buffer the H2D copy reads, so the pass replaces the plain memcpy.
//
// The codecs are resolved from the system's shared libraries at first use
// (libzstd.so.1, liblz4.so.1); when one is missing the chunk stays
// NBL_ERR_UNSUPPORTED and the caller's CPU factory takes it.
#include <dlfcn.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <unordered_map>

#include "../../include/nimble_amd.h"

extern "C" void nbl_set_last_error(const char* msg);  // nbl_plan.cpp

namespace {

enum : uint8_t { CT_UNCOMPRESSED = 0, CT_ZSTD = 1, CT_LZ4 = 3 };  // common/Types.h:190-197
enum : uint8_t { ET_TRIVIAL = 0, ET_RLE = 1, ET_DICTIONARY = 2, ET_FIXEDBITWIDTH = 3, ET_NULLABLE = 5, ET_SPARSEBOOL = 6,
                 ET_VARINT = 7, ET_CONSTANT = 9, ET_MAINLYCONSTANT = 10, ET_BLOCKBITPACKING = 15, ET_FOR = 18 };
enum : uint8_t { DT_BOOL = 11, DT_STRING = 12 };

struct Fail {
  nbl_status status;
  std::string msg;
};
[[noreturn]] void fail(nbl_status st, const char* fmt, ...) {
  char buf[256];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  throw Fail{st, buf};
}
#define INF_CHECK(cond, ...) \
  do { if (!(cond)) fail(NBL_ERR_CORRUPT, __VA_ARGS__); } while (0)

using ZstdDecompress = size_t (*)(void*, size_t, const void*, size_t);
using ZstdIsError = unsigned (*)(size_t);
using Lz4DecompressSafe = int (*)(const char*, char*, int, int);
struct Codecs {
  ZstdDecompress zstd = nullptr;
  ZstdIsError zstd_is_error = nullptr;
  Lz4DecompressSafe lz4 = nullptr;
};
const Codecs& codecs() {
  static Codecs c;
  static std::once_flag once;
  std::call_once(once, [] {
    for (const char* name : {"libzstd.so.1", "libzstd.so"}) {
      if (void* h = dlopen(name, RTLD_NOW | RTLD_LOCAL)) {
        c.zstd = reinterpret_cast<ZstdDecompress>(dlsym(h, "ZSTD_decompress"));
        c.zstd_is_error = reinterpret_cast<ZstdIsError>(dlsym(h, "ZSTD_isError"));
        if (c.zstd && c.zstd_is_error) break;
      }
    }
    for (const char* name : {"liblz4.so.1", "liblz4.so"}) {
      if (void* h = dlopen(name, RTLD_NOW | RTLD_LOCAL)) {
        c.lz4 = reinterpret_cast<Lz4DecompressSafe>(dlsym(h, "LZ4_decompress_safe"));
        if (c.lz4) break;
      }
    }
  });
  return c;
}

uint32_t rd32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
void wr32(uint8_t* p, uint32_t v) { memcpy(p, &v, 4); }

// size of the payload `data` once uncompressed
uint64_t payload_size(uint8_t ct, const uint8_t* data, uint64_t size) {
  if (ct == CT_UNCOMPRESSED) return size;
  if (ct != CT_ZSTD && ct != CT_LZ4) fail(NBL_ERR_UNSUPPORTED, "payload compression type %d is not handled", (int)ct);
  INF_CHECK(size >= 4, "compressed payload shorter than its size field");
  return rd32(data);
}
// writes the uncompressed payload to out (payload_size bytes)
void payload_inflate(uint8_t ct, const uint8_t* data, uint64_t size, uint8_t* out) {
  if (ct == CT_UNCOMPRESSED) { memcpy(out, data, size); return; }
  const uint32_t usize = rd32(data);
  const Codecs& c = codecs();
  if (ct == CT_ZSTD) {
    if (!c.zstd) fail(NBL_ERR_UNSUPPORTED, "Zstd payload but libzstd.so.1 is not available");
    const size_t r = c.zstd(out, usize, data + 4, size - 4);
    INF_CHECK(!c.zstd_is_error(r) && r == usize, "Error uncompressing data (zstd)");
  } else {
    if (!c.lz4) fail(NBL_ERR_UNSUPPORTED, "LZ4 payload but liblz4.so.1 is not available");
    INF_CHECK(size - 4 <= 0x7FFFFFFFull && usize <= 0x7FFFFFFFu, "LZ4 payload too large");
    const int r = c.lz4(reinterpret_cast<const char*>(data + 4), reinterpret_cast<char*>(out), (int)(size - 4), (int)usize);
    INF_CHECK(r >= 0 && (uint32_t)r == usize, "Error decompressing LZ4 data.");
  }
}

struct Walker {
  bool varint_rows;
  bool any_compressed = false;
  // Rewritten sizes of the children that are preceded by a varint size field
  // (the block encodings' per-segment arrays), keyed by the child's address:
  // the size has to be known before the child is written, and sizing a child by
  // walking it again at every level would cost 2^depth walks on a nested chain.
  std::unordered_map<const uint8_t*, uint64_t> sized;

  // Size of the rewritten encoding [p, p+size); when out != nullptr also writes it.
  uint64_t rewrite(const uint8_t* p, uint64_t size, uint8_t* out, int depth) {
    INF_CHECK(size >= 2 && depth < 32, "encoding shorter than its prefix");
    const uint8_t enc = p[0], dtype = p[1];
    uint64_t pos = 2;
    if (varint_rows) {
      for (int k = 0; k < 5; ++k) {
        INF_CHECK(pos < size, "row count varint beyond end of encoding");
        if (!(p[pos++] & 128)) break;
      }
    } else {
      INF_CHECK(size >= 6, "encoding shorter than its prefix");
      pos = 6;
    }
    const uint64_t prefix = pos;
    uint64_t o = 0;
    auto copy = [&](const uint8_t* src, uint64_t n) { if (out) memcpy(out + o, src, n); o += n; };
    auto child = [&](uint64_t coff, uint64_t csize) {
      INF_CHECK(coff <= size && csize <= size - coff, "child encoding beyond end of its parent");
      const uint64_t n = rewrite(p + coff, csize, out ? out + o : nullptr, depth + 1);
      o += n;
      return n;
    };
    // child preceded by its u32 size, which is patched to the rewritten size
    auto sized_child = [&]() {
      INF_CHECK(pos + 4 <= size, "child size beyond end of encoding");
      const uint32_t csize = rd32(p + pos);
      const uint64_t at = o;
      o += 4;
      const uint64_t n = child(pos + 4, csize);
      INF_CHECK(n <= 0xFFFFFFFFull, "rewritten child exceeds 4 GiB");
      if (out) wr32(out + at, (uint32_t)n);
      pos += 4 + (uint64_t)csize;
    };
    auto payload_of = [&](uint8_t ct, uint64_t bytes) {  // `bytes` at pos are one (possibly compressed) payload
      if (ct != CT_UNCOMPRESSED) any_compressed = true;
      const uint64_t n = payload_size(ct, p + pos, bytes);
      if (out) payload_inflate(ct, p + pos, bytes, out + o);
      o += n;
      pos += bytes;
      return n;
    };
    auto payload = [&](uint8_t ct) { payload_of(ct, size - pos); };  // the rest of the encoding
    auto read_varint = [&]() {  // varint::readVarint32, common/Varint.h:104-125
      uint32_t v = 0;
      for (int k = 0; k < 5; ++k) {
        INF_CHECK(pos < size, "varint beyond end of encoding");
        const uint8_t b = p[pos++];
        v |= (uint32_t)(b & 127) << (7 * k);
        if (!(b & 128)) break;
      }
      return v;
    };
    auto write_varint = [&](uint32_t v) {
      uint8_t buf[5];
      int n = 0;
      while (v >= 128) { buf[n++] = (uint8_t)(128 | (v & 127)); v >>= 7; }
      buf[n++] = (uint8_t)v;
      copy(buf, (uint64_t)n);
    };
    auto copy_varints = [&](int count) {
      const uint64_t from = pos;
      for (int k = 0; k < count; ++k) (void)read_varint();
      copy(p + from, pos - from);
    };
    // child preceded by its varint size, which is rewritten for the child's new size
    auto varint_child = [&]() {
      const uint32_t csize = read_varint();
      INF_CHECK(csize <= size - pos, "child encoding beyond end of its parent");
      uint64_t n;
      if (!out) {  // sizing pass: one walk of the child, remembered for the writing pass
        n = rewrite(p + pos, csize, nullptr, depth + 1);
        sized[p + pos] = n;
        INF_CHECK(n <= 0xFFFFFFFFull, "rewritten child exceeds 4 GiB");
        write_varint((uint32_t)n);
        o += n;
      } else {
        const auto it = sized.find(p + pos);
        n = it != sized.end() ? it->second : rewrite(p + pos, csize, nullptr, depth + 1);
        INF_CHECK(n <= 0xFFFFFFFFull, "rewritten child exceeds 4 GiB");
        write_varint((uint32_t)n);
        child(pos, csize);
      }
      pos += csize;
    };
    copy(p, prefix);
    switch (enc) {
      case ET_TRIVIAL: {
        INF_CHECK(pos + 1 <= size, "Trivial header beyond end");
        const uint8_t ct = p[pos++];
        const uint8_t zero = CT_UNCOMPRESSED;
        copy(&zero, 1);
        if (dtype == DT_STRING) sized_child();
        payload(ct);
        break;
      }
      case ET_FIXEDBITWIDTH: {
        const uint64_t es = (dtype == 7 || dtype == 8 || dtype == 10) ? 8 : (dtype == 5 || dtype == 6 || dtype == 9) ? 4 : (dtype == 3 || dtype == 4) ? 2 : 1;
        INF_CHECK(pos + 2 + es <= size, "FixedBitWidth header beyond end");
        const uint8_t ct = p[pos];
        const uint8_t zero = CT_UNCOMPRESSED;
        copy(&zero, 1);
        copy(p + pos + 1, es + 1);  // baseline, bit width
        pos += 2 + es;
        payload(ct);
        break;
      }
      case ET_RLE:
        sized_child();
        if (dtype == DT_BOOL) copy(p + pos, size - pos);
        else child(pos, size - pos);
        break;
      case ET_DICTIONARY:
      case ET_NULLABLE:
        sized_child();
        child(pos, size - pos);
        break;
      case ET_SPARSEBOOL:
        INF_CHECK(pos + 1 <= size, "SparseBool header beyond end");
        copy(p + pos, 1);
        pos += 1;
        child(pos, size - pos);
        break;
      case ET_MAINLYCONSTANT:
        sized_child();
        sized_child();
        copy(p + pos, size - pos);  // the common value
        break;
      case ET_BLOCKBITPACKING: {  // BlockBitPackingEncoding.h:311-348
        INF_CHECK(pos + 1 <= size, "BlockBitPacking header beyond end");
        const uint8_t ct = p[pos++];
        const uint8_t zero = CT_UNCOMPRESSED;
        copy(&zero, 1);
        copy_varints(2);  // block size, block count
        varint_child(); varint_child(); varint_child();  // baselines, bit widths, block offsets
        copy_varints(1);  // rows of the first block
        payload(ct);
        break;
      }
      case ET_FOR: {  // ForEncoding.h:174-196
        INF_CHECK(pos + 1 <= size, "FOR header beyond end");
        const uint8_t ct = p[pos++];
        const uint8_t zero = CT_UNCOMPRESSED;
        copy(&zero, 1);
        copy_varints(3);  // frame size, frame count, rows of the first frame
        varint_child(); varint_child(); varint_child();  // bit widths, references, bit offsets
        const uint32_t packed = read_varint();
        INF_CHECK(packed <= size - pos, "FOR packed data beyond end");
        const uint64_t usize = payload_size(ct, p + pos, packed);
        INF_CHECK(usize <= 0xFFFFFFFFull, "FOR packed data exceeds 4 GiB");
        write_varint((uint32_t)usize);
        payload_of(ct, packed);
        break;
      }
      case ET_VARINT:
      case ET_CONSTANT:
        copy(p + pos, size - pos);
        break;
      default:
        // unknown or unsupported here: nothing to inflate, the planner passes the verdict
        copy(p + pos, size - pos);
        break;
    }
    return o;
  }
};

template <typename F>
nbl_status guarded(F&& f) {
  try {
    f();
    return NBL_OK;
  } catch (const Fail& e) {
    nbl_set_last_error(e.msg.c_str());
    return e.status;
  }
}

}  // namespace

extern "C" {

nbl_status nbl_uncompressed_size(uint8_t compression, const uint8_t* data, uint64_t size, uint64_t* out_size) {
  if (!out_size || (!data && size)) { nbl_set_last_error("null argument"); return NBL_ERR_ARGUMENT; }
  return guarded([&] { *out_size = payload_size(compression, data, size); });
}

nbl_status nbl_uncompress(uint8_t compression, const uint8_t* data, uint64_t size, uint8_t* out, uint64_t capacity) {
  if ((!data && size) || !out) { nbl_set_last_error("null argument"); return NBL_ERR_ARGUMENT; }
  return guarded([&] {
    if (payload_size(compression, data, size) > capacity) fail(NBL_ERR_ARGUMENT, "output buffer too small");
    payload_inflate(compression, data, size, out);
  });
}

nbl_status nbl_chunk_inflated_size(const uint8_t* chunk, uint64_t size, const nbl_options* options, uint64_t* out_size) {
  if (!out_size) { nbl_set_last_error("null argument"); return NBL_ERR_ARGUMENT; }
  // an empty chunk is malformed Nimble, whatever its pointer (EncodingFactory::create reads the prefix first)
  if (size == 0) { nbl_set_last_error("encoding shorter than its prefix"); return NBL_ERR_CORRUPT; }
  if (!chunk) { nbl_set_last_error("null argument"); return NBL_ERR_ARGUMENT; }
  return guarded([&] {
    Walker w{options && options->struct_size >= 8 && options->use_varint_row_count != 0};
    const uint64_t n = w.rewrite(chunk, size, nullptr, 0);
    *out_size = w.any_compressed ? n : 0;
  });
}

nbl_status nbl_chunk_inflate(const uint8_t* chunk, uint64_t size, const nbl_options* options, uint8_t* out, uint64_t capacity,
                             uint64_t* out_size) {
  if (!chunk || !out || !out_size) { nbl_set_last_error("null argument"); return NBL_ERR_ARGUMENT; }
  return guarded([&] {
    Walker w{options && options->struct_size >= 8 && options->use_varint_row_count != 0};
    const uint64_t need = w.rewrite(chunk, size, nullptr, 0);
    if (need > capacity) fail(NBL_ERR_ARGUMENT, "output buffer too small");
    *out_size = w.rewrite(chunk, size, out, 0);
  });
}

}  // extern "C"

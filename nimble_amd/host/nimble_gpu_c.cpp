// Flat C wrappers over the C++ host layer (nimble_gpu.h) so that the Python
// tests can drive GpuEncodingFactory / Encoding / GpuStripeDecoder through
// ctypes.  Every function returns 0 on success, non-zero on a thrown error
// whose text nblh_last_error() then returns.
#include <cstring>
#include <string>
#include <vector>

#include <hip/hip_runtime_api.h>

#include "nimble_gpu.h"
#include "tablet.h"

using namespace nimble_amd;

namespace {
thread_local std::string g_err;
thread_local int g_status = 0;

struct EncHandle {
  std::unique_ptr<Encoding> enc;
  std::vector<std::unique_ptr<char[]>> strings;  // stringBufferFactory memory
};
struct FactoryHandle {
  DecodingStats stats;  // named by the factory's options when the flags word asks for statistics
  std::unique_ptr<GpuEncodingFactory> f;
};
struct StripeHandle {
  std::vector<std::string> streams;  // owned copies of the stream bytes
  std::unique_ptr<GpuStripeDecoder> dec;
};

template <typename F>
int guard(F&& f) {
  try {
    f();
    g_status = 0;
    return 0;
  } catch (const NimbleError& e) {
    g_err = e.what();
    g_status = e.status();
    return e.status() ? e.status() : -1;
  } catch (const std::exception& e) {
    g_err = e.what();
    g_status = -1;
    return -1;
  }
}
}  // namespace

extern "C" {

const char* nblh_last_error() { return g_err.c_str(); }

void* nblh_factory_create(int use_varint_row_count, int device) {
  FactoryHandle* h = nullptr;
  guard([&] {
    h = new FactoryHandle();
    Options o;
    o.useVarintRowCount = (use_varint_row_count & 1) != 0;
    o.keepDictionary = (use_varint_row_count & 2) != 0;  // bit 1 of the flags word
    if (use_varint_row_count & 4) o.decodingStats = &h->stats;  // bit 2
    h->f = std::make_unique<GpuEncodingFactory>(o, device);
  });
  return h;
}
void nblh_factory_free(void* f) { delete static_cast<FactoryHandle*>(f); }

// {count, sum, min, max} of decodeCPUTimeNanos, decompressCPUTimeNanos, encodedBytes, decodedBytes, planNanos, deviceNanos: 24 values
void nblh_factory_stats(void* f, uint64_t* out) {
  const DecodingStats& s = static_cast<FactoryHandle*>(f)->stats;
  const IoCounter* all[6] = {&s.decodeCPUTimeNanos, &s.decompressCPUTimeNanos, &s.encodedBytes, &s.decodedBytes, &s.planNanos, &s.deviceNanos};
  for (int i = 0; i < 6; ++i) {
    out[4 * i] = all[i]->count();
    out[4 * i + 1] = all[i]->sum();
    out[4 * i + 2] = all[i]->min();
    out[4 * i + 3] = all[i]->max();
  }
}

void* nblh_encoding_create(void* factory, const char* data, uint64_t size) {
  EncHandle* h = new EncHandle();
  const int rc = guard([&] {
    auto sbf = [h](uint32_t bytes) -> void* {
      h->strings.push_back(std::make_unique<char[]>(bytes ? bytes : 1));
      return h->strings.back().get();
    };
    h->enc = static_cast<FactoryHandle*>(factory)->f->create(std::string_view(data, size), sbf);
  });
  if (rc != 0) {
    delete h;
    return nullptr;
  }
  return h;
}
void nblh_encoding_free(void* e) { delete static_cast<EncHandle*>(e); }

static Encoding& E(void* e) { return *static_cast<EncHandle*>(e)->enc; }

uint32_t nblh_row_count(void* e) { return E(e).rowCount(); }
uint32_t nblh_null_count(void* e) { return E(e).nullCount(); }
int nblh_is_nullable(void* e) { return E(e).isNullable(); }
int nblh_data_type(void* e) { return (int)E(e).dataType(); }
int nblh_encoding_type(void* e) { return (int)E(e).encodingType(); }
// debugString into buf (truncated to cap - 1 characters)
void nblh_debug_string(void* e, int offset, char* buf, uint32_t cap) {
  const std::string s = E(e).debugString(offset);
  if (cap) { strncpy(buf, s.c_str(), cap - 1); buf[cap - 1] = 0; }
}
int nblh_reset(void* e) { return guard([&] { E(e).reset(); }); }
int nblh_skip(void* e, uint32_t n) { return guard([&] { E(e).skip(n); }); }
int nblh_materialize(void* e, uint32_t n, void* buffer) { return guard([&] { E(e).materialize(n, buffer); }); }
int nblh_materialize_bools_as_bits(void* e, uint32_t n, uint64_t* buffer, int begin) {
  return guard([&] { E(e).materializeBoolsAsBits(n, buffer, begin); });
}
// *non_null receives the return value; *requested is set when getOutputNulls was called.
int nblh_materialize_nullable(void* e, uint32_t n, void* buffer, void* nulls, int* requested, const void* scatter,
                              uint32_t scatter_size, uint32_t offset, uint32_t* non_null) {
  return guard([&] {
    int req = 0;
    Bitmap bm{scatter, scatter_size};
    *non_null = E(e).materializeNullable(
        n, buffer, [&]() { req = 1; return nulls; }, scatter ? &bm : nullptr, offset);
    if (requested) *requested = req;
  });
}
// std::string_view[n] -> lengths + concatenated bytes
uint64_t nblh_sv_total(const void* sv, uint32_t n) {
  uint64_t t = 0;
  for (uint32_t i = 0; i < n; ++i) t += static_cast<const std::string_view*>(sv)[i].size();
  return t;
}
void nblh_sv_flatten(const void* sv, uint32_t n, uint32_t* lengths, char* bytes) {
  uint64_t off = 0;
  for (uint32_t i = 0; i < n; ++i) {
    const auto& s = static_cast<const std::string_view*>(sv)[i];
    lengths[i] = (uint32_t)s.size();
    if (!s.empty()) memcpy(bytes + off, s.data(), s.size());
    off += s.size();
  }
}
int nblh_sizeof_string_view() { return (int)sizeof(std::string_view); }

// dictionary API
int nblh_dictionary_enabled(void* e) { return E(e).dictionaryEnabled(); }
int nblh_dictionary_size(void* e, uint32_t* size) { return guard([&] { *size = E(e).dictionarySize(); }); }
// *entries receives dictionaryEntries(): physicalType[] or std::string_view[]
int nblh_dictionary_entries(void* e, const void** entries) { return guard([&] { *entries = E(e).dictionaryEntries(); }); }
int nblh_materialize_indices(void* e, uint32_t n, uint32_t* buffer) { return guard([&] { E(e).materializeIndices(n, buffer); }); }

// ---- encoding view ------------------------------------------------------------
int nblh_supports_encoding_view(int encoding_type) { return supportsEncodingView(static_cast<EncodingType>(encoding_type)); }
void* nblh_view_create(void* factory, const char* data, uint64_t size) {
  EncodingView* v = nullptr;
  guard([&] { v = static_cast<FactoryHandle*>(factory)->f->createEncodingView(std::string_view(data, size)).release(); });
  return v;
}
void nblh_view_free(void* v) { delete static_cast<EncodingView*>(v); }
uint32_t nblh_view_row_count(void* v) { return static_cast<EncodingView*>(v)->rowCount(); }
int nblh_view_encoding_type(void* v) { return (int)static_cast<EncodingView*>(v)->encodingType(); }
int nblh_view_data_type(void* v) { return (int)static_cast<EncodingView*>(v)->dataType(); }
int nblh_view_read(void* v, uint32_t offset, uint32_t length, void* output) {
  return guard([&] { static_cast<EncodingView*>(v)->read(offset, length, output); });
}
int nblh_view_read_at(void* v, uint32_t index, void* output) {
  return guard([&] { static_cast<EncodingView*>(v)->readAt(index, output); });
}

// ---- stripe decoder --------------------------------------------------------
void* nblh_stripe_create(void* factory, const char* const* streams, const uint64_t* sizes, uint32_t n) {
  StripeHandle* h = new StripeHandle();
  const int rc = guard([&] {
    std::vector<std::string_view> views;
    for (uint32_t i = 0; i < n; ++i) h->streams.emplace_back(streams[i], sizes[i]);
    for (auto& s : h->streams) views.emplace_back(s);
    h->dec = std::make_unique<GpuStripeDecoder>(*static_cast<FactoryHandle*>(factory)->f, views);
  });
  if (rc != 0) {
    delete h;
    return nullptr;
  }
  return h;
}
void nblh_stripe_free(void* s) { delete static_cast<StripeHandle*>(s); }
int nblh_stripe_next(void* s, uint32_t stream, uint32_t count, void* output, void* nulls, int* requested, const void* scatter,
                     uint32_t scatter_size, uint32_t* non_null, uint32_t* n_string_buffers) {
  return guard([&] {
    int req = 0;
    Bitmap bm{scatter, scatter_size};
    std::vector<std::shared_ptr<void>> bufs;
    std::function<void*()> getNulls;
    if (nulls) getNulls = [&]() { req = 1; return nulls; };
    *non_null = static_cast<StripeHandle*>(s)->dec->decoder(stream).next(count, output, bufs, getNulls, scatter ? &bm : nullptr);
    if (requested) *requested = req;
    if (n_string_buffers) *n_string_buffers = (uint32_t)bufs.size();
  });
}
int nblh_stripe_skip(void* s, uint32_t stream, uint32_t count) {
  return guard([&] { static_cast<StripeHandle*>(s)->dec->decoder(stream).skip(count); });
}
int nblh_stripe_reset(void* s, uint32_t stream) {
  return guard([&] { static_cast<StripeHandle*>(s)->dec->decoder(stream).reset(); });
}

// selective reader's ChunkedDecoder over stream `stream` (created on first use with `with_nulls`)
static ChunkedDecoder& CD(void* s, uint32_t stream, int with_nulls = 0) {
  return static_cast<StripeHandle*>(s)->dec->chunkedDecoder(stream, with_nulls != 0);
}
int nblh_chunked_open(void* s, uint32_t stream, int decode_values_with_nulls) {
  return guard([&] { CD(s, stream, decode_values_with_nulls); });
}
int nblh_chunked_skip(void* s, uint32_t stream, int64_t n) { return guard([&] { CD(s, stream).skip(n); }); }
int nblh_chunked_next_bools(void* s, uint32_t stream, uint64_t* data, int64_t count, const uint64_t* incoming) {
  return guard([&] { CD(s, stream).nextBools(data, count, incoming); });
}
int nblh_chunked_next_indices(void* s, uint32_t stream, int32_t* data, int64_t count, const uint64_t* incoming) {
  return guard([&] { CD(s, stream).nextIndices(data, count, incoming); });
}
int nblh_chunked_decode_nullable(void* s, uint32_t stream, uint64_t* nulls, void* data, int64_t count, const uint64_t* incoming) {
  return guard([&] { CD(s, stream).decodeNullable(nulls, data, count, incoming); });
}
int nblh_chunked_preloaded_values(void* s, uint32_t stream, const uint32_t** values) {
  return guard([&] { *values = CD(s, stream).preloadedValues(); });
}
// *encoding_type = -1 when no chunk is loaded
int nblh_chunked_state(void* s, uint32_t stream, int ensure_loaded, int64_t* remaining, int* encoding_type, int* dictionary_convertible) {
  return guard([&] {
    auto& d = CD(s, stream);
    if (ensure_loaded) d.ensureLoaded();
    *remaining = d.remainingValues();
    const Encoding* e = d.currentEncoding();
    *encoding_type = e ? (int)e->encodingType() : -1;
    *dictionary_convertible = e ? (int)d.dictionaryConvertible() : 0;
  });
}

// ---- device-resident batch ----------------------------------------------------
void* nblh_decode_to_device(void* factory, const char* const* chunks, const uint64_t* sizes, uint32_t n) {
  DeviceBatch* b = nullptr;
  guard([&] {
    std::vector<std::string_view> views;
    for (uint32_t i = 0; i < n; ++i) views.emplace_back(chunks[i], sizes[i]);
    auto& f = *static_cast<FactoryHandle*>(factory)->f;
    b = f.decodeToDevice(views, f.options()).release();
  });
  return b;
}
void nblh_device_batch_free(void* b) { delete static_cast<DeviceBatch*>(b); }
uint64_t nblh_device_batch_data(void* b) { return reinterpret_cast<uint64_t>(static_cast<DeviceBatch*>(b)->data()); }
uint64_t nblh_device_batch_bytes(void* b) { return static_cast<DeviceBatch*>(b)->bytes(); }
int nblh_device_batch_layout(void* b, uint32_t chunk, nbl_chunk_layout* out) {
  return guard([&] { *out = static_cast<DeviceBatch*>(b)->layout(chunk); });
}
// test helper: the whole device buffer copied to `host` (bytes() long)
int nblh_device_batch_copy_out(void* b, void* host) {
  return guard([&] {
    auto* B = static_cast<DeviceBatch*>(b);
    if (B->bytes() && hipMemcpy(host, B->data(), B->bytes(), hipMemcpyDeviceToHost) != hipSuccess)
      throw NimbleError(NBL_ERR_HIP, "hipMemcpy(device batch)");
  });
}

// ---- tablet (file front-end) ------------------------------------------------
// `file` must stay alive while the handle is used
void* nblh_tablet_open(const char* file, uint64_t size) {
  Tablet* t = nullptr;
  guard([&] { t = new Tablet(std::string_view(file, size)); });
  return t;
}
void nblh_tablet_free(void* t) { delete static_cast<Tablet*>(t); }
uint32_t nblh_tablet_stripe_count(void* t) { return static_cast<Tablet*>(t)->stripeCount(); }
uint64_t nblh_tablet_row_count(void* t) { return static_cast<Tablet*>(t)->rowCount(); }
int nblh_tablet_stripe_info(void* t, uint32_t stripe, uint32_t* rows, uint64_t* offset, uint32_t* size, uint32_t* streams) {
  return guard([&] {
    auto* T = static_cast<Tablet*>(t);
    *rows = T->stripeRowCount(stripe); *offset = T->stripeOffset(stripe); *size = T->stripeSize(stripe); *streams = T->streamCount(stripe);
  });
}
// (*data, *size) = the stream's bytes inside the file buffer; size 0 when absent
int nblh_tablet_stream(void* t, uint32_t stripe, uint32_t stream, const char** data, uint64_t* size) {
  return guard([&] {
    const std::string_view v = static_cast<Tablet*>(t)->stream(stripe, stream);
    *data = v.data(); *size = v.size();
  });
}
int nblh_tablet_optional_section(void* t, const char* name, const char** data, uint64_t* size) {
  return guard([&] {
    const std::string_view v = static_cast<Tablet*>(t)->optionalSection(name);
    *data = v.data(); *size = v.size();
  });
}
// all non-empty streams of a stripe decoded in one GPU batch; stream ids (ascending) to ids[0..*n)
void* nblh_tablet_decode_stripe(void* factory, void* t, uint32_t stripe, uint32_t* ids, uint32_t capacity, uint32_t* n) {
  StripeHandle* h = new StripeHandle();
  const int rc = guard([&] {
    std::vector<std::string_view> views;
    const auto streams = static_cast<Tablet*>(t)->streams(stripe);
    *n = (uint32_t)streams.size();
    for (size_t i = 0; i < streams.size(); ++i) {
      if (i < capacity) ids[i] = streams[i].first;
      views.push_back(streams[i].second);
    }
    h->dec = std::make_unique<GpuStripeDecoder>(*static_cast<FactoryHandle*>(factory)->f, views);
  });
  if (rc != 0) { delete h; return nullptr; }
  return h;
}

// ---- all stripes of a file over several devices -----------------------------
// owner[i] = index of the device (0 .. devices - 1) stripe i is assigned to
int nblh_assign_stripes(const uint64_t* sizes, uint32_t n, uint32_t devices, uint32_t* owner) {
  return guard([&] {
    const auto owned = GpuTabletDecoder::assignStripes(std::vector<uint64_t>(sizes, sizes + n), devices);
    for (size_t d = 0; d < owned.size(); ++d)
      for (uint32_t s : owned[d]) owner[s] = (uint32_t)d;
  });
}
void* nblh_tablet_decoder_create(void* t, const int32_t* devices, uint32_t n_devices, int flags, uint32_t lookahead) {
  GpuTabletDecoder* d = nullptr;
  guard([&] {
    Options o;
    o.useVarintRowCount = (flags & 1) != 0;
    o.keepDictionary = (flags & 2) != 0;
    d = new GpuTabletDecoder(*static_cast<Tablet*>(t), std::vector<int>(devices, devices + n_devices), o, lookahead);
  });
  return d;
}
void nblh_tablet_decoder_free(void* d) { delete static_cast<GpuTabletDecoder*>(d); }
int nblh_tablet_decoder_device_of(void* d, uint32_t stripe, int32_t* device) {
  return guard([&] { *device = static_cast<GpuTabletDecoder*>(d)->deviceOf(stripe); });
}
// blocks until the stripe is decoded; the returned handle is a stripe handle (nblh_stripe_*)
void* nblh_tablet_decoder_take(void* d, uint32_t stripe, uint32_t* ids, uint32_t capacity, uint32_t* n) {
  StripeHandle* h = new StripeHandle();
  const int rc = guard([&] {
    auto* D = static_cast<GpuTabletDecoder*>(d);
    const auto sid = D->streamIds(stripe);
    *n = (uint32_t)sid.size();
    for (size_t i = 0; i < sid.size() && i < capacity; ++i) ids[i] = sid[i];
    h->dec = D->take(stripe);
  });
  if (rc != 0) { delete h; return nullptr; }
  return h;
}

}  // extern "C"

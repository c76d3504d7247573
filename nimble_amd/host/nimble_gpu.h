// C++ host layer above the C ABI (include/nimble_amd.h): GPU-backed mirrors of
// the reference's extension and reader interfaces for the cascaded-stream
// decode path, with the same names, argument meaning and error behaviour:
//
//   nimble::Encoding ............ dwio/nimble/encodings/common/Encoding.h:110-389
//   nimble::EncodingFactory ..... dwio/nimble/encodings/common/EncodingFactory.h:32-55
//   nimble::Decoder ............. dwio/nimble/velox/Decoder.h:27-56
//   ChunkedStreamDecoder ........ dwio/nimble/velox/ChunkedStreamDecoder.h:25-74
//   nimble::ChunkedDecoder ...... dwio/nimble/velox/selective/ChunkedDecoder.h:54-311 (dense bulk entry points)
//   nimble::EncodingView ........ dwio/nimble/encodings/views/EncodingView.h:32-104,
//                                 encodings/views/EncodingViewFactory.h:36-46
//
// Velox is not available to this build, so the two Velox types in those
// signatures are replaced by the smallest equivalents: the MemoryPool argument
// is dropped (host memory comes from the C++ allocator) and bits::Bitmap is the
// two-field view below.  A Nimble maintainer plugs these classes into the real
// interfaces with a thin adapter that forwards each virtual one-to-one.
#pragma once

#include <atomic>
#include <cstdint>
#include <functional>
#include <memory>
#include <stdexcept>
#include <string>
#include <string_view>
#include <vector>

#include "nimble_amd.h"

namespace nimble_amd {

// Error convention of the reference (dwio/nimble/common/Exceptions.h:111-155):
// malformed input and broken invariants throw.  `status` is the nbl_status.
class NimbleError : public std::runtime_error {
 public:
  NimbleError(int status, const std::string& msg) : std::runtime_error(msg), status_(status) {}
  int status() const { return status_; }

 private:
  int status_;
};
// Well-formed Nimble the GPU path does not handle (compressed payloads,
// experimental encodings): the integration catches this and calls the CPU factory.
class NimbleUnsupported : public NimbleError {
 public:
  using NimbleError::NimbleError;
};

// velox::bits::Bitmap stand-in: `size` bits at `bits`, LSB first.
struct Bitmap {
  const void* bits;
  uint32_t size;
  bool test(uint32_t pos) const { return (static_cast<const uint8_t*>(bits)[pos >> 3] >> (pos & 7)) & 1; }
};

enum class EncodingType : uint8_t {
  Trivial = 0, RLE = 1, Dictionary = 2, FixedBitWidth = 3, Sentinel = 4, Nullable = 5, SparseBool = 6,
  Varint = 7, Delta = 8, Constant = 9, MainlyConstant = 10, BlockBitPacking = 15, FOR = 18
};
enum class DataType : uint8_t {
  Undefined = 0, Int8 = 1, Uint8 = 2, Int16 = 3, Uint16 = 4, Int32 = 5, Uint32 = 6, Int64 = 7, Uint64 = 8,
  Float = 9, Double = 10, Bool = 11, String = 12
};

// toString(EncodingType) / toString(DataType), dwio/nimble/common/Types.cpp:25-120
std::string toString(EncodingType encodingType);
std::string toString(DataType dataType);

// velox::io::IoCounter as DecodingStats uses it: count, sum, minimum and maximum
// of the samples it was given; safe to increment from several threads.
class IoCounter {
 public:
  void increment(uint64_t value) {
    count_.fetch_add(1, std::memory_order_relaxed);
    sum_.fetch_add(value, std::memory_order_relaxed);
    for (uint64_t m = min_.load(std::memory_order_relaxed); value < m && !min_.compare_exchange_weak(m, value, std::memory_order_relaxed);) {}
    for (uint64_t m = max_.load(std::memory_order_relaxed); value > m && !max_.compare_exchange_weak(m, value, std::memory_order_relaxed);) {}
  }
  uint64_t count() const { return count_.load(std::memory_order_relaxed); }
  uint64_t sum() const { return sum_.load(std::memory_order_relaxed); }
  uint64_t min() const { return count() ? min_.load(std::memory_order_relaxed) : 0; }
  uint64_t max() const { return max_.load(std::memory_order_relaxed); }

 private:
  std::atomic<uint64_t> count_{0}, sum_{0}, min_{~0ull}, max_{0};
};

// velox::dwio::common::DecodingStats as Encoding::Options::decodingStats hands it
// to the encodings (dwio/nimble/encodings/common/Encoding.h:160-170): the
// reference times decompression into it per column, and its readers expect
// decompression to be a part of the decode time
// (velox/selective/tests/SelectiveNimbleReaderTest.cpp:2805-2822).  Here a
// sample is a batch -- a chunk handed to create(), or all chunks of a stripe --
// since that is what the device decodes at once.
struct DecodingStats {
  IoCounter decodeCPUTimeNanos;      // host time of a batch: staging, inflating (all threads), planning, launching, the wait for the status
  IoCounter decompressCPUTimeNanos;  // of that, inflating Zstd / LZ4 payloads: one sample per inflated chunk
  // Not in the reference: what a batch moved and what it cost beside the host's time.
  IoCounter encodedBytes;            // bytes staged for the device (after inflating)
  IoCounter decodedBytes;            // bytes of decoded output
  IoCounter planNanos;               // the header walk (nbl_plan_batch), part of decodeCPUTimeNanos
  IoCounter deviceNanos;             // device time from the first to the last kernel of the decode
};

// Encoding::Options fields that matter on the decode side (Encoding.h:115-175).
struct Options {
  bool useVarintRowCount = false;
  // Not a reference option: also fetch the alphabet and raw indices of
  // dictionary-encoded chunks, which turns on the dictionary API below
  // (costs 4 more bytes per row over the link).
  bool keepDictionary = false;
  // Encoding.h:160-161; may be shared by factories and threads, must outlive the decodes.
  DecodingStats* decodingStats = nullptr;
};

// Same contract as nimble::Encoding: a row cursor over one decoded chunk.
// materialize() writes physicalType[] (bool: one byte per row, string:
// std::string_view[] into stringBufferFactory memory); null rows of a Nullable
// chunk read as physicalType().  materializeNullable() leaves null slots
// untouched, asks for the null bitmap only when the range has nulls (or a
// scatter with gaps), returns the number of non-null rows.
class Encoding {
 public:
  virtual ~Encoding() = default;
  virtual EncodingType encodingType() const = 0;
  virtual DataType dataType() const = 0;
  virtual uint32_t rowCount() const = 0;
  virtual uint32_t nullCount() const = 0;
  virtual bool isNullable() const = 0;
  virtual void reset() = 0;
  virtual void skip(uint32_t rowCount) = 0;
  virtual void materialize(uint32_t rowCount, void* buffer) = 0;
  virtual uint32_t materializeNullable(uint32_t rowCount, void* buffer, std::function<void*()> getOutputNulls,
                                       const Bitmap* scatterOutputBitmap = nullptr, uint32_t offset = 0) = 0;
  virtual void materializeBoolsAsBits(uint32_t rowCount, uint64_t* buffer, int begin) = 0;
  // "<indent>Encoding<DataType> rowCount=N" (Encoding::debugString, Encoding.cpp:45-52)
  virtual std::string debugString(int offset = 0) const {
    return std::string((size_t)(offset > 0 ? offset : 0), ' ') + toString(encodingType()) + "<" + toString(dataType()) + "> rowCount=" +
           std::to_string(rowCount());
  }
  // Dictionary API (Encoding.h:277-311).  True for a Dictionary chunk and for a
  // Nullable chunk over one (NullableEncoding.h:107-122) when the factory was
  // asked to keep dictionaries; everything else answers false, which sends a
  // dictionary-aware reader down its ordinary materialize path.
  virtual bool dictionaryEnabled() const { return false; }
  virtual uint32_t dictionarySize() const { throw NimbleError(NBL_ERR_ARGUMENT, "dictionarySize on non-dictionary encoding"); }
  // contiguous alphabet: physicalType[dictionarySize()], std::string_view[] for strings
  virtual const void* dictionaryEntries() const { throw NimbleError(NBL_ERR_ARGUMENT, "dictionaryEntries on non-dictionary encoding"); }
  virtual const void* dictionaryEntry(uint32_t /*index*/) const { throw NimbleError(NBL_ERR_ARGUMENT, "dictionaryEntry on non-dictionary encoding"); }
  // raw indices of the next rowCount rows; shares the row cursor with materialize (DictionaryEncoding.h:81-83)
  virtual void materializeIndices(uint32_t /*rowCount*/, uint32_t* /*buffer*/) {
    throw NimbleError(NBL_ERR_ARGUMENT, "materializeIndices on non-dictionary encoding");
  }
};

// Same contract as nimble::EncodingView: cursor-free random access to the
// physical values of one stream.  readAt / read write physicalType (bool: one
// byte, string: std::string_view into memory the view owns); a range outside
// [0, rowCount()] throws.
class EncodingView {
 public:
  virtual ~EncodingView() = default;
  virtual void readAt(uint32_t index, void* output) const = 0;
  virtual void read(uint32_t offset, uint32_t length, void* output) const = 0;
  virtual uint32_t rowCount() const = 0;
  virtual EncodingType encodingType() const = 0;
  virtual DataType dataType() const = 0;
};
// supportsEncodingView (EncodingViewFactory.cpp:161-182) restricted to the
// encodings this library decodes.
bool supportsEncodingView(EncodingType encodingType);

class GpuContext;  // device, stream and reusable device buffers

// Mirror of nimble::EncodingFactory with the decode running on the GPU.  A
// factory is const, shareable and re-entrant (one mutex around the device
// buffers); each Encoding it returns is single-threaded and stateful.
class GpuEncodingFactory {
 public:
  using StringBufferFactory = std::function<void*(uint32_t)>;
  using Fallback = std::function<std::unique_ptr<Encoding>(std::string_view, StringBufferFactory)>;

  explicit GpuEncodingFactory(Options options = {}, int device = 0);
  virtual ~GpuEncodingFactory();

  // EncodingFactory::create(pool, data, stringBufferFactory): plans, decodes and
  // fetches the one chunk `data`.  The returned Encoding does not own `data`.
  virtual std::unique_ptr<Encoding> create(std::string_view data, StringBufferFactory stringBufferFactory) const;
  // The overload that takes per-call options (EncodingFactory.h:47-51).
  virtual std::unique_ptr<Encoding> create(std::string_view data, StringBufferFactory stringBufferFactory,
                                           const Options& options) const;
  const Options& options() const { return options_; }
  // Where unsupported-but-valid encodings go (the reference's CPU factory in a
  // real integration).  Without one, NimbleUnsupported propagates.
  void setFallback(Fallback fallback) { fallback_ = std::move(fallback); }

  // Batch form: decodes many chunks with ONE plan / one set of launches and
  // returns one Encoding per chunk.  This is what a stripe-level reader uses.
  std::vector<std::unique_ptr<Encoding>> createBatch(const std::vector<std::string_view>& chunks,
                                                     StringBufferFactory stringBufferFactory,
                                                     const Options& options) const;

  // createEncodingView(data, pool, options) (EncodingViewFactory.h:42-46): the
  // stream is decoded once on the GPU and the view serves reads from the
  // decoded rows, so readAt is an array lookup whatever the encoding.  Throws
  // NimbleUnsupported for encodings the reference has no view for (Nullable,
  // Varint) as well as for those this library does not decode.
  std::unique_ptr<EncodingView> createEncodingView(std::string_view data, const Options& options) const;
  std::unique_ptr<EncodingView> createEncodingView(std::string_view data) const { return createEncodingView(data, options_); }

  // Device-resident form, for consumers that run on the GPU themselves (no
  // counterpart in the reference, whose vectors live in host memory): the
  // chunks are decoded into a device buffer the returned object owns, and
  // nothing is copied back.  This is the shape in which the decoder delivers
  // its throughput; the Encoding objects above pay two PCIe crossings and two
  // host copies for the same bytes.
  std::unique_ptr<class DeviceBatch> decodeToDevice(const std::vector<std::string_view>& chunks, const Options& options) const;

 private:
  Options options_;
  Fallback fallback_;
  std::shared_ptr<GpuContext> ctx_;
};

// Decoded chunks left in device memory (GpuEncodingFactory::decodeToDevice).
// layout(i) says where chunk i's values / null bitmap / string bytes sit
// relative to data(), in the formats of nbl_chunk_layout.
class DeviceBatch {
 public:
  virtual ~DeviceBatch() = default;
  virtual size_t chunkCount() const = 0;
  virtual const nbl_chunk_layout& layout(size_t chunk) const = 0;
  virtual const void* data() const = 0;   // device pointer, bytes() long
  virtual uint64_t bytes() const = 0;
  virtual int device() const = 0;
  virtual void* stream() const = 0;       // hipStream_t the decode ran on (already synchronised)
};

// nimble::Decoder (velox/Decoder.h:27-56) with std::vector<std::shared_ptr<void>>
// standing in for the BufferPtr list that keeps string bytes alive.
class Decoder {
 public:
  virtual ~Decoder() = default;
  virtual uint32_t next(uint32_t count, void* output, std::vector<std::shared_ptr<void>>& stringBuffers,
                        std::function<void*()> getOutputNulls = nullptr,
                        const Bitmap* scatterOutputBitmap = nullptr) = 0;
  virtual void skip(uint32_t count) = 0;
  virtual void reset() = 0;
  virtual const Encoding* encoding() const = 0;
};

// The selective reader's per-stream decoder, restricted to its dense bulk entry
// points (no visitor, no filter): skip, nextBools, nextIndices,
// decodeNullable<T> and the chunk inspection calls the dictionary-aware string
// reader makes.  Bitmaps are uint64 words, LSB first, 1 = not null
// (velox::bits::kNotNull).  `incomingNulls` marks the rows of the read range
// that have a value in this stream; rows without one become nulls.
class ChunkedDecoder {
 public:
  virtual ~ChunkedDecoder() = default;
  // skipWithoutIndex (ChunkedDecoder.cpp:270-286)
  virtual void skip(int64_t numValues) = 0;
  // ChunkedDecoder.cpp:155-191.  With decodeValuesWithNulls the stream is a
  // nullable uint32 stream: its NULL bitmap is the result, its values are kept
  // for preloadedValues().
  virtual void nextBools(uint64_t* data, int64_t count, const uint64_t* incomingNulls) = 0;
  // ChunkedDecoder.cpp:193-206
  virtual void nextIndices(int32_t* data, int64_t count, const uint64_t* incomingNulls) = 0;
  // decodeNullable<T> (ChunkedDecoder.h:206-268); T is the stream's physical type
  virtual void decodeNullable(uint64_t* nulls, void* data, int64_t count, const uint64_t* incomingNulls) = 0;
  virtual const uint32_t* preloadedValues() const = 0;  // getPreloadedValues (:91-96)
  virtual void ensureLoaded() = 0;                      // :289-298
  virtual const Encoding* currentEncoding() const = 0;  // :301-303
  virtual bool dictionaryConvertible() const = 0;       // :305-309
  virtual int64_t remainingValues() const = 0;          // :311-313
};

// All streams of a stripe decoded in one batch; hands out one Decoder per
// stream with ChunkedStreamDecoder semantics (velox/ChunkedStreamDecoder.cpp:75-205).
// Mirrors VeloxReader::loadNextStripe creating every stream's decoder at once
// (velox/VeloxReader.cpp:330-351).
class GpuStripeDecoder {
 public:
  // `streams[i]` = the raw bytes of stream i (chunk headers included); the
  // bytes must stay alive while the decoders are used.
  GpuStripeDecoder(const GpuEncodingFactory& factory, const std::vector<std::string_view>& streams);
  ~GpuStripeDecoder();
  size_t streamCount() const;
  Decoder& decoder(size_t stream);
  // The same decoded stream through the selective reader's interface.  A stream
  // is read through ONE of the two objects: they share its chunks' row cursors.
  ChunkedDecoder& chunkedDecoder(size_t stream, bool decodeValuesWithNulls = false);

 private:
  struct Impl;
  std::unique_ptr<Impl> impl_;
};

}  // namespace nimble_amd

// GPU-backed Encoding / EncodingFactory / Decoder (see nimble_gpu.h).
#include "nimble_gpu.h"
#include "worker_pool.h"

#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <exception>
#include <functional>
#include <mutex>
#include <thread>
#include <memory>
#include <utility>
#include <vector>

namespace nimble_amd {

namespace {

[[noreturn]] void throwStatus(int st) {
  const std::string msg = nbl_last_error();
  if (st == NBL_ERR_UNSUPPORTED) throw NimbleUnsupported(st, msg);
  throw NimbleError(st, msg);
}
void check(int st) {
  if (st != NBL_OK) throwStatus(st);
}
void hipCheck(hipError_t e, const char* what) {
  if (e != hipSuccess) throw NimbleError(NBL_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}

inline bool bitIsSet(const void* bits, uint64_t i) { return (static_cast<const uint8_t*>(bits)[i >> 3] >> (i & 7)) & 1; }
inline void bitSet(void* bits, uint64_t i) { static_cast<uint8_t*>(bits)[i >> 3] |= uint8_t(1u << (i & 7)); }
inline void bitClear(void* bits, uint64_t i) { static_cast<uint8_t*>(bits)[i >> 3] &= uint8_t(~(1u << (i & 7))); }

// bits [begin, end) := v; whole bytes with memset
void fillBits(void* bits, uint64_t begin, uint64_t end, bool v) {
  auto* p = static_cast<uint8_t*>(bits);
  while (begin < end && (begin & 7)) { v ? bitSet(p, begin) : bitClear(p, begin); ++begin; }
  const uint64_t whole = (end - begin) >> 3;
  if (whole) memset(p + (begin >> 3), v ? 0xFF : 0x00, whole);
  begin += whole << 3;
  while (begin < end) { v ? bitSet(p, begin) : bitClear(p, begin); ++begin; }
}
// f(i) for every set bit i in [begin, end), ascending; never reads a byte outside the range
template <typename F>
void forEachSetBit(const void* bits, uint64_t begin, uint64_t end, F&& f) {
  const auto* p = static_cast<const uint8_t*>(bits);
  uint64_t i = begin;
  while (i < end) {
    if ((i & 7) == 0 && i + 64 <= end) {
      uint64_t w;
      memcpy(&w, p + (i >> 3), 8);
      while (w) { f(i + (uint64_t)__builtin_ctzll(w)); w &= w - 1; }
      i += 64;
      continue;
    }
    const uint32_t avail = (uint32_t)std::min<uint64_t>(8 - (i & 7), end - i);
    uint32_t b = (uint32_t)(p[i >> 3] >> (i & 7)) & ((1u << avail) - 1u);
    while (b) { f(i + (uint64_t)__builtin_ctz(b)); b &= b - 1; }
    i += avail;
  }
}
uint32_t countBits(const void* bits, uint64_t begin, uint64_t end) {
  const auto* p = static_cast<const uint8_t*>(bits);
  uint32_t n = 0;
  uint64_t i = begin;
  while (i < end) {
    if ((i & 7) == 0 && i + 64 <= end) {
      uint64_t w;
      memcpy(&w, p + (i >> 3), 8);
      n += (uint32_t)__builtin_popcountll(w);
      i += 64;
      continue;
    }
    const uint32_t avail = (uint32_t)std::min<uint64_t>(8 - (i & 7), end - i);
    n += (uint32_t)__builtin_popcount((uint32_t)(p[i >> 3] >> (i & 7)) & ((1u << avail) - 1u));
    i += avail;
  }
  return n;
}
// velox::bits::findSetBit: position of the n-th (1-based) set bit in [begin,end), `end` if fewer.
uint32_t findSetBit(const void* bits, uint32_t begin, uint32_t end, uint32_t n) {
  if (n == 0) return end;
  const auto* p = static_cast<const uint8_t*>(bits);
  uint64_t i = begin;
  uint32_t seen = 0;
  while (i < end) {
    const uint32_t avail = (uint32_t)std::min<uint64_t>(8 - (i & 7), end - i);
    uint32_t b = (uint32_t)(p[i >> 3] >> (i & 7)) & ((1u << avail) - 1u);
    const uint32_t c = (uint32_t)__builtin_popcount(b);
    if (seen + c >= n) {
      for (;; b &= b - 1)
        if (++seen == n) return (uint32_t)(i + (uint64_t)__builtin_ctz(b));
    }
    seen += c;
    i += avail;
  }
  return end;
}
// dst bits [d, d + n) := src bits [s, s + n)
void copyBits(void* dst, uint64_t d, const void* src, uint64_t s, uint64_t n) {
  if (((d | s) & 7) == 0) {
    memcpy(static_cast<uint8_t*>(dst) + (d >> 3), static_cast<const uint8_t*>(src) + (s >> 3), n >> 3);
    for (uint64_t i = n & ~7ull; i < n; ++i) bitIsSet(src, s + i) ? bitSet(dst, d + i) : bitClear(dst, d + i);
    return;
  }
  for (uint64_t i = 0; i < n; ++i) bitIsSet(src, s + i) ? bitSet(dst, d + i) : bitClear(dst, d + i);
}

struct DeviceBuffer {
  void* ptr = nullptr;
  size_t cap = 0;
  // Grows with a quarter of headroom, so a loop over stripes of one shape never
  // reallocates.  `streams` = the streams whose queued work may still use the old
  // block (a copy-back of the batch before last): waited for before it is freed.
  void reserve(size_t n, std::initializer_list<hipStream_t> streams = {}) {
    if (n <= cap) return;
    if (ptr)
      for (hipStream_t stream : streams) hipCheck(hipStreamSynchronize(stream), "hipStreamSynchronize");
    if (ptr) hipCheck(hipFree(ptr), "hipFree");
    ptr = nullptr;
    cap = 0;
    const size_t want = std::max<size_t>(n + n / 4, 1 << 20);
    hipCheck(hipMalloc(&ptr, want), "hipMalloc");
    cap = want;
  }
  ~DeviceBuffer() {
    if (ptr) (void)hipFree(ptr);
  }
};

// Pinned host memory, recycled: every batch needs one block for the encoded
// bytes and one for the decoded output, and the decoded block stays alive as
// long as any Encoding of the batch does.
class PinnedPool : public std::enable_shared_from_this<PinnedPool> {
 public:
  ~PinnedPool() {
    for (auto& b : free_) (void)hipHostFree(b.first);
  }
  std::shared_ptr<uint8_t> acquire(size_t bytes) {
    const size_t want = std::max<size_t>(bytes, 4096);
    void* ptr = nullptr;
    size_t cap = 0;
    {
      std::lock_guard<std::mutex> lock(mu_);
      size_t best = free_.size();
      for (size_t i = 0; i < free_.size(); ++i)
        if (free_[i].second >= want && (best == free_.size() || free_[i].second < free_[best].second)) best = i;
      if (best != free_.size() && free_[best].second <= 2 * want + (1 << 20)) {
        ptr = free_[best].first;
        cap = free_[best].second;
        free_.erase(free_.begin() + (long)best);
      }
    }
    if (!ptr) {
      cap = want + want / 8;
      hipCheck(hipHostMalloc(&ptr, cap, hipHostMallocDefault), "hipHostMalloc");
      // A new block's pages become resident when they are first touched.  Left to the device's first copy
      // into the block that costs 160-200 ms for a 1.8 GB stripe (a tenth of the link's rate); touched here,
      // by the staging threads, about 100 ms, and it is the thread making the batch that waits, not the
      // readers of the stripe before.  Blocks are recycled, so a loop of stripes pays this for its first
      // three stripes only.
      if (cap >= (8u << 20)) {
        const size_t pages = (cap + 4095) / 4096;
        std::atomic<size_t> next{0};
        const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
        detail::parallelRun(std::min(hw, 16u), [&] {
          for (size_t k = next.fetch_add(256); k < pages; k = next.fetch_add(256))
            for (size_t q = k; q < std::min(pages, k + 256); ++q) static_cast<volatile uint8_t*>(ptr)[q * 4096] = 0;
        });
      }
    }
    auto self = shared_from_this();
    return std::shared_ptr<uint8_t>(static_cast<uint8_t*>(ptr), [self, cap](uint8_t* p) { self->release(p, cap); });
  }

 private:
  void release(void* p, size_t cap) {
    std::lock_guard<std::mutex> lock(mu_);
    if (free_.size() >= 8) {  // keep a few blocks, give the rest back
      (void)hipHostFree(p);
      return;
    }
    free_.emplace_back(p, cap);
  }
  std::mutex mu_;
  std::vector<std::pair<void*, size_t>> free_;
};

}  // namespace

// A context owns two slots, each a stream with device buffers of its own, and
// one more stream for the copies back.  Consecutive batches alternate slots, so
// the copy in and the decode of one stripe run while the decoded output of the
// stripe before is still on its way back (the link is full duplex): the stripe
// loop of a reader (VeloxReader::loadNextStripe, velox/VeloxReader.cpp:330-351)
// then pays the longer of the two transfers per stripe instead of their sum.
// That holds because the copy back is a kernel (nbl_copy_to_host_async) and
// not a copy engine's: the runtime gives a stream a copy engine per direction
// the first time the stream copies that way, and two streams that each copy
// both ways end up with the copy in of one on the engine of the other's copy
// back -- the transfers then take turns (tools/link_probe.hip shows both
// cases).  Here the engines only ever copy in.  Three streams also stay within
// the four hardware queues a process has, so none waits behind another's
// kernel.  A slot is reused by the batch after next, whose decode waits for
// the copy back that still reads the slot's output.
class GpuContext {
 public:
  struct Slot {
    std::mutex mu;
    hipStream_t stream = nullptr;
    DeviceBuffer arena, out, work;
    hipEvent_t decodeBegin = nullptr, decodeEnd = nullptr;  // created on the first batch that asks for statistics
    hipEvent_t outFree = nullptr;  // on copyOut: the last copy back out of the slot's output has landed
    bool outBusy = false;          // outFree has been recorded
  };
  explicit GpuContext(int device) : device_(device) {}
  ~GpuContext() {
    if (copyOut) (void)hipStreamSynchronize(copyOut);
    for (Slot& s : slots_)
      if (s.stream) {
        (void)hipStreamSynchronize(s.stream);
        for (hipEvent_t e : {s.decodeBegin, s.decodeEnd, s.outFree})
          if (e) (void)hipEventDestroy(e);
        (void)hipStreamDestroy(s.stream);
      }
    if (copyOut) (void)hipStreamDestroy(copyOut);
  }
  int device_;
  hipStream_t copyOut = nullptr;  // created with the first slot
  std::shared_ptr<PinnedPool> pinned = std::make_shared<PinnedPool>();
  // The slot of the next batch, locked: the one whose turn it is, the other one
  // when that is busy (two callers never queue on one slot while the other idles),
  // and only then a wait for the first.  The device is current and the slot's
  // stream exists on return.
  std::unique_lock<std::mutex> acquire(Slot*& slot) {
    const unsigned first = next_.fetch_add(1, std::memory_order_relaxed) % kSlots;
    std::unique_lock<std::mutex> lock;
    for (unsigned k = 0; k < kSlots && !lock.owns_lock(); ++k) {
      slot = &slots_[(first + k) % kSlots];
      lock = std::unique_lock<std::mutex>(slot->mu, std::try_to_lock);
    }
    if (!lock.owns_lock()) {
      slot = &slots_[first];
      lock = std::unique_lock<std::mutex>(slot->mu);
    }
    hipCheck(hipSetDevice(device_), "hipSetDevice");
    if (!slot->stream) {
      hipCheck(hipStreamCreateWithFlags(&slot->stream, hipStreamNonBlocking), "hipStreamCreate");
      hipCheck(hipEventCreateWithFlags(&slot->outFree, hipEventDisableTiming), "hipEventCreate");
      std::lock_guard<std::mutex> once(copyMu_);
      if (!copyOut) hipCheck(hipStreamCreateWithFlags(&copyOut, hipStreamNonBlocking), "hipStreamCreate");
    }
    return lock;
  }

 private:
  static constexpr unsigned kSlots = 2;
  Slot slots_[kSlots];
  std::atomic<unsigned> next_{0};
  std::mutex copyMu_;
};

namespace {

// The device-to-host copy of a batch's decoded output, issued in slices with an
// event behind each: the batch is handed out while the copy is still running and
// an Encoding waits only for the slices its own bytes lie in, so the caller's
// reads of the first chunks overlap the transfer of the later ones.  Owns the
// pinned block: it goes back to the pool only after the last slice has landed.
class OutputCopy {
 public:
  explicit OutputCopy(std::shared_ptr<uint8_t> host) : host_(std::move(host)) {}
  OutputCopy(const OutputCopy&) = delete;
  OutputCopy& operator=(const OutputCopy&) = delete;
  ~OutputCopy() {
    if (!events_.empty()) (void)hipEventSynchronize(events_.back());
    for (hipEvent_t e : events_) (void)hipEventDestroy(e);
  }
  const uint8_t* host() const { return host_.get(); }
  // enqueues the copy of [0, bytes) on `stream` (the device of the stream is current)
  void enqueue(const void* dev, uint64_t bytes, hipStream_t stream) {
    constexpr uint64_t kSlice = 32ull << 20;
    for (uint64_t off = 0; off < bytes; off += kSlice) {
      const uint64_t len = std::min(kSlice, bytes - off);
      // by kernel: the copy engines stay with the copies in, whatever engine the runtime gave this stream
      check(nbl_copy_to_host_async(host_.get() + off, static_cast<const uint8_t*>(dev) + off, len, stream));
      hipEvent_t ev = nullptr;
      hipCheck(hipEventCreateWithFlags(&ev, hipEventDisableTiming), "hipEventCreate");
      events_.push_back(ev);
      ends_.push_back(off + len);
      hipCheck(hipEventRecord(ev, stream), "hipEventRecord");
    }
  }
  // on return the bytes [0, end) are in host memory
  void waitFor(uint64_t end) const {
    if (done_.load(std::memory_order_acquire) >= end || events_.empty()) return;
    size_t i = 0;
    while (i + 1 < ends_.size() && ends_[i] < end) ++i;
    hipCheck(hipEventSynchronize(events_[i]), "D2H");  // slices complete in stream order
    uint64_t seen = done_.load(std::memory_order_relaxed);
    while (seen < ends_[i] && !done_.compare_exchange_weak(seen, ends_[i], std::memory_order_release)) {}
  }

 private:
  std::shared_ptr<uint8_t> host_;
  std::vector<hipEvent_t> events_;
  std::vector<uint64_t> ends_;
  mutable std::atomic<uint64_t> done_{0};
};

// One decoded chunk resident in host memory: a window into the batch's pinned
// output block, which every Encoding of the batch keeps alive (no per-chunk copy).
class GpuEncoding final : public Encoding {
 public:
  GpuEncoding(const nbl_chunk_layout& lay, std::shared_ptr<const OutputCopy> block, const GpuEncodingFactory::StringBufferFactory& sbf)
      : lay_(lay), block_(std::move(block)) {
    es_ = lay.elem_bytes;
    const uint8_t* hostOut = block_->host();
    values_ = hostOut + lay.values_offset;
    nulls_ = lay.nullable ? hostOut + lay.nulls_offset : nullptr;
    // the last byte of the output block this chunk reads
    end_ = lay.values_offset + lay.values_bytes;
    if (lay.nullable) end_ = std::max(end_, lay.nulls_offset + lay.nulls_bytes);
    end_ = std::max(end_, lay.string_offset + lay.string_bytes);
    if (lay.has_dictionary) {
      end_ = std::max(end_, lay.dict_alphabet_offset + (uint64_t)lay.dict_size * lay.elem_bytes);
      if (lay.dict_indices_rows) end_ = std::max(end_, lay.dict_indices_offset + lay.dict_indices_rows * 4);
    }
    // Nothing of the output block is read here: every read path waits for the chunk's part of it first
    // (arrived()), and the work a string chunk needs once -- its bytes copied into the caller's buffer, every
    // view checked against it, the dictionary's entries as string_views -- is done behind that wait, by the
    // thread that reads the chunk first.  The batch is handed out without waiting for the copy back, and the
    // readers of a stripe share that work instead of the thread that made the batch doing all of it.
    if (lay.data_type == (uint8_t)DataType::String) {
      // string bytes live in memory the caller's factory hands out, as the
      // reference's string encodings do (TrivialEncoding.cpp:129-135); the factory is called here, on the
      // thread that creates the encoding
      strings_ = lay.string_bytes ? static_cast<char*>(sbf(static_cast<uint32_t>(lay.string_bytes))) : nullptr;
    }
    if (lay.has_dictionary) {
      dictAlphabet_ = hostOut + lay.dict_alphabet_offset;
      dictIndices_ = lay.dict_indices_rows ? reinterpret_cast<const uint32_t*>(hostOut + lay.dict_indices_offset) : nullptr;
    }
  }
  bool dictionaryEnabled() const override { return lay_.has_dictionary != 0; }
  uint32_t dictionarySize() const override {
    if (!lay_.has_dictionary) return Encoding::dictionarySize();
    return lay_.dict_size;
  }
  const void* dictionaryEntries() const override {
    if (!lay_.has_dictionary) return Encoding::dictionaryEntries();
    arrived();
    return lay_.data_type == (uint8_t)DataType::String ? static_cast<const void*>(dictStrings_.data()) : dictAlphabet_;
  }
  const void* dictionaryEntry(uint32_t index) const override {
    if (!lay_.has_dictionary || index >= lay_.dict_size) return Encoding::dictionaryEntry(index);
    arrived();
    if (lay_.data_type == (uint8_t)DataType::String) return &dictStrings_[index];
    return dictAlphabet_ + (size_t)index * es_;
  }
  void materializeIndices(uint32_t n, uint32_t* buffer) override {
    if (!dictIndices_) return Encoding::materializeIndices(n, buffer);
    const uint32_t r = advance(n);
    arrived();
    memcpy(buffer, dictIndices_ + r, (size_t)n * sizeof(uint32_t));
  }
  EncodingType encodingType() const override { return static_cast<EncodingType>(lay_.encoding_type); }
  DataType dataType() const override { return static_cast<DataType>(lay_.data_type); }
  uint32_t rowCount() const override { return lay_.rows; }
  uint32_t nullCount() const override { return lay_.rows - lay_.non_null_rows; }
  bool isNullable() const override { return lay_.nullable != 0; }
  void reset() override { row_ = 0; }
  void skip(uint32_t n) override { advance(n); }

  void materialize(uint32_t n, void* buffer) override {
    const uint32_t r = advance(n);
    arrived();
    copyRows(r, n, buffer, 0);
  }

  uint32_t materializeNullable(uint32_t n, void* buffer, std::function<void*()> getOutputNulls, const Bitmap* scatter,
                               uint32_t offset) override {
    const uint32_t r = advance(n);
    const uint32_t scatterSize = scatter ? scatter->size - offset : n;
    if (scatterSize < n) throw NimbleError(NBL_ERR_ARGUMENT, "Scattered output must have at least rowCount positions");
    arrived();
    if (!lay_.nullable) {  // TypedEncoding::materializeNullable, Encoding.h:412-460
      if (scatterSize == n) {
        copyRows(r, n, buffer, offset);
        return n;
      }
      void* nulls = getOutputNulls();
      copyBits(nulls, offset, scatter->bits, offset, scatterSize);
      uint32_t src = r;
      const uint32_t stop = r + n;
      forEachSetBit(scatter->bits, offset, (uint64_t)offset + scatterSize, [&](uint64_t pos) {
        if (src < stop) copyRows(src++, 1, buffer, (uint32_t)pos);
      });
      return n;
    }
    // NullableEncoding::materializeNullable, NullableEncoding.h:239-306
    const uint32_t nonNull = countBits(nulls_, r, (uint64_t)r + n);
    if (nonNull == scatterSize) {  // no nulls in range and no gaps: bitmap untouched
      copyRows(r, n, buffer, offset);
      return nonNull;
    }
    void* nulls = getOutputNulls();
    if (!scatter) {
      // row r + i lands in slot offset + i: the null bitmap is a bit copy, values go where the bit is set
      copyBits(nulls, offset, nulls_, r, n);
      forEachSetBit(nulls_, r, (uint64_t)r + n, [&](uint64_t row) { copyRows((uint32_t)row, 1, buffer, offset + (uint32_t)(row - r)); });
      return nonNull;
    }
    fillBits(nulls, offset, (uint64_t)offset + scatterSize, false);
    uint32_t src = r;
    const uint32_t stop = r + n;
    forEachSetBit(scatter->bits, offset, (uint64_t)offset + scatterSize, [&](uint64_t pos) {
      if (src >= stop) return;
      if (bitIsSet(nulls_, src)) {
        bitSet(nulls, pos);
        copyRows(src, 1, buffer, (uint32_t)pos);
      }
      ++src;
    });
    return nonNull;
  }

  void materializeBoolsAsBits(uint32_t n, uint64_t* buffer, int begin) override {
    if (lay_.data_type != (uint8_t)DataType::Bool) throw NimbleError(NBL_ERR_ARGUMENT, "materializeBoolsAsBits on a non-bool encoding");
    const uint32_t r = advance(n);
    arrived();
    for (uint32_t i = 0; i < n; ++i) values_[r + i] ? bitSet(buffer, (uint64_t)begin + i) : bitClear(buffer, (uint64_t)begin + i);
  }

  // cursor-free read of rows [offset, offset + length), for EncodingView
  void readRows(uint32_t offset, uint32_t length, void* buffer) const {
    if (offset > lay_.rows || length > lay_.rows - offset) throw NimbleError(NBL_ERR_ARGUMENT, "read range beyond the row count");
    arrived();
    copyRows(offset, length, buffer, 0);
  }

 private:
  uint32_t advance(uint32_t n) {
    if ((uint64_t)row_ + n > lay_.rows) throw NimbleError(NBL_ERR_ARGUMENT, "read past the end of the encoding");
    const uint32_t r = row_;
    row_ += n;
    return r;
  }
  // rows [r, r+n) of the decoded chunk -> slots [slot, slot+n) of `buffer`
  void copyRows(uint32_t r, uint32_t n, void* buffer, uint32_t slot) const {
    if (lay_.data_type == (uint8_t)DataType::String) {
      auto* out = static_cast<std::string_view*>(buffer) + slot;
      const uint64_t* views = reinterpret_cast<const uint64_t*>(values_);
      for (uint32_t i = 0; i < n; ++i) {
        const uint64_t v = views[r + i];
        out[i] = std::string_view(strings_ + (v & 0xFFFFFFFFull), static_cast<size_t>(v >> 32));
      }
      return;
    }
    if (n == 1) {  // the scatter loops copy row by row: keep that a load and a store
      switch (es_) {
        case 8: static_cast<uint64_t*>(buffer)[slot] = reinterpret_cast<const uint64_t*>(values_)[r]; return;
        case 4: static_cast<uint32_t*>(buffer)[slot] = reinterpret_cast<const uint32_t*>(values_)[r]; return;
        case 2: static_cast<uint16_t*>(buffer)[slot] = reinterpret_cast<const uint16_t*>(values_)[r]; return;
        default: static_cast<uint8_t*>(buffer)[slot] = values_[r]; return;
      }
    }
    memcpy(static_cast<uint8_t*>(buffer) + (size_t)slot * es_, values_ + (size_t)r * es_, (size_t)n * es_);
  }

  // blocks until this chunk's part of the output block has arrived in host memory
  void arrived() const {
    block_->waitFor(end_);
    if (lay_.data_type == (uint8_t)DataType::String) std::call_once(stringsOnce_, [this] { takeStrings(); });
  }
  // once per chunk, with the chunk's bytes in host memory
  void takeStrings() const {
    const uint8_t* hostOut = block_->host();
    if (lay_.string_bytes) memcpy(strings_, hostOut + lay_.string_offset, lay_.string_bytes);
    auto check_views = [&](const uint64_t* views, uint64_t n) {
      for (uint64_t i = 0; i < n; ++i) {
        const uint64_t off = views[i] & 0xFFFFFFFFull, len = views[i] >> 32;
        if (off + len > lay_.string_bytes) throw NimbleError(NBL_ERR_DEVICE_DATA, "string view outside the string buffer");
      }
    };
    check_views(reinterpret_cast<const uint64_t*>(values_), lay_.rows);
    if (lay_.has_dictionary) {  // dictionaryEntries() hands out std::string_view[]
      const uint64_t* views = reinterpret_cast<const uint64_t*>(dictAlphabet_);
      check_views(views, lay_.dict_size);
      dictStrings_.clear();
      dictStrings_.reserve(lay_.dict_size);
      for (uint32_t i = 0; i < lay_.dict_size; ++i)
        dictStrings_.emplace_back(strings_ + (views[i] & 0xFFFFFFFFull), static_cast<size_t>(views[i] >> 32));
    }
  }

  nbl_chunk_layout lay_;
  std::shared_ptr<const OutputCopy> block_;
  uint64_t end_ = 0;
  uint32_t es_ = 0;
  uint32_t row_ = 0;
  const uint8_t* values_ = nullptr;
  const uint8_t* nulls_ = nullptr;
  char* strings_ = nullptr;
  const uint8_t* dictAlphabet_ = nullptr;
  const uint32_t* dictIndices_ = nullptr;
  mutable std::vector<std::string_view> dictStrings_;
  mutable std::once_flag stringsOnce_;
};

// EncodingView over a decoded chunk; owns the memory its string views point into.
class GpuEncodingView final : public EncodingView {
 public:
  GpuEncodingView(std::unique_ptr<Encoding> enc, std::shared_ptr<std::vector<std::unique_ptr<char[]>>> strings)
      : enc_(std::move(enc)), strings_(std::move(strings)) {}
  void readAt(uint32_t index, void* output) const override {
    if (index >= enc_->rowCount()) throw NimbleError(NBL_ERR_ARGUMENT, "row index beyond the row count");
    rows().readRows(index, 1, output);
  }
  void read(uint32_t offset, uint32_t length, void* output) const override { rows().readRows(offset, length, output); }
  uint32_t rowCount() const override { return enc_->rowCount(); }
  EncodingType encodingType() const override { return enc_->encodingType(); }
  DataType dataType() const override { return enc_->dataType(); }

 private:
  const GpuEncoding& rows() const { return static_cast<const GpuEncoding&>(*enc_); }
  std::unique_ptr<Encoding> enc_;
  std::shared_ptr<std::vector<std::unique_ptr<char[]>>> strings_;
};

using detail::parallelRun;

struct PlanHolder {
  nbl_plan* p = nullptr;
  PlanHolder() = default;
  PlanHolder(const PlanHolder&) = delete;
  PlanHolder& operator=(const PlanHolder&) = delete;
  PlanHolder(PlanHolder&& o) noexcept : p(o.p) { o.p = nullptr; }
  PlanHolder& operator=(PlanHolder&& o) noexcept {
    if (this != &o) {
      nbl_plan_free(p);
      p = o.p;
      o.p = nullptr;
    }
    return *this;
  }
  ~PlanHolder() { nbl_plan_free(p); }
};

nbl_options planOptions(const Options& opts) {
  nbl_options o{};
  o.struct_size = sizeof(nbl_options);
  o.use_varint_row_count = opts.useVarintRowCount ? 1u : 0u;
  o.flags = opts.keepDictionary ? NBL_PLAN_KEEP_DICTIONARY : 0u;
  return o;
}

// The encoded bytes of a batch on their way to a slot's arena, and the plan over them.
struct StagedBatch {
  PlanHolder plan;
  nbl_plan_info info{};
  std::shared_ptr<uint8_t> staging;  // pinned image of the arena; alive until the copy has been waited for
};

// Host and device time of one batch for Options::decodingStats.  Host time is
// the caller's wall time with the parallel copy / inflate phases counted by
// what their tasks took on every thread, so the inflating time reported
// beside it is a part of it whatever the thread count.
using StatClock = std::chrono::steady_clock;
inline uint64_t nanosSince(StatClock::time_point t) {
  return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(StatClock::now() - t).count();
}
struct BatchTimer {
  DecodingStats* stats;
  StatClock::time_point begin = StatClock::now();
  uint64_t phaseNanos = 0;                // wall time of the parallel phases
  std::atomic<uint64_t> taskNanos{0};     // time of their tasks, all threads
  explicit BatchTimer(DecodingStats* s) : stats(s) {}
  void deviceBegin(GpuContext::Slot& slot) {
    if (!stats) return;
    if (!slot.decodeBegin) {
      hipCheck(hipEventCreate(&slot.decodeBegin), "hipEventCreate");
      hipCheck(hipEventCreate(&slot.decodeEnd), "hipEventCreate");
    }
    hipCheck(hipEventRecord(slot.decodeBegin, slot.stream), "hipEventRecord");
  }
  void deviceEnd(GpuContext::Slot& slot) {
    if (stats) hipCheck(hipEventRecord(slot.decodeEnd, slot.stream), "hipEventRecord");
  }
  // after the slot's stream has been synchronised
  void finish(GpuContext::Slot& slot, const nbl_plan_info& info, uint64_t stagedBytes) {
    if (!stats) return;
    float ms = 0;
    hipCheck(hipEventElapsedTime(&ms, slot.decodeBegin, slot.decodeEnd), "hipEventElapsedTime");
    stats->deviceNanos.increment((uint64_t)((double)ms * 1e6));
    stats->encodedBytes.increment(stagedBytes);
    stats->decodedBytes.increment(info.out_bytes);
    stats->decodeCPUTimeNanos.increment(nanosSince(begin) - phaseNanos + taskNanos.load());
  }
};

// Packs the chunks (16-byte aligned starts) into one pinned staging block -- the
// image the device arena receives; chunks with Zstd / LZ4 leaf payloads, the
// reference writer's default, are inflated straight into it -- and copies it to
// slot.arena on slot.stream.  The block is filled and sent in slices of about
// 32 MiB: while the copy engine moves one slice, the host threads fill the
// next, so a stripe pays the longer of the two instead of their sum.  The plan
// is made once the last slice is on its way (the header walk reads the staged
// bytes), i.e. under the tail of the transfer.  The slot is locked by the caller.
StagedBatch stageBatch(GpuContext& ctx, GpuContext::Slot& slot, const std::vector<std::string_view>& chunks, const Options& opts,
                       BatchTimer& timer, uint64_t& stagedBytes) {
  StagedBatch b;
  const nbl_options o = planOptions(opts);
  std::vector<uint64_t> offs(chunks.size()), sizes(chunks.size());
  std::vector<uint8_t> inflate(chunks.size(), 0);
  uint64_t total = 0;
  for (size_t i = 0; i < chunks.size(); ++i) {
    uint64_t inflated = 0;
    check(nbl_chunk_inflated_size(reinterpret_cast<const uint8_t*>(chunks[i].data()), chunks[i].size(), &o, &inflated));
    inflate[i] = inflated != 0;
    offs[i] = total;
    sizes[i] = inflated ? inflated : chunks[i].size();
    total += (sizes[i] + 15) & ~uint64_t(15);
  }
  b.staging = ctx.pinned->acquire(total);
  slot.arena.reserve(total + 64, {slot.stream});  // + the slop nbl_plan_info::arena_bytes asks for behind the last chunk
  // Slices of whole chunks, about 32 MiB each; a chunk to inflate is one task, a chunk to copy is cut into
  // pieces, so that a slice of three or four large chunks still keeps every thread busy.  The tasks of all
  // slices are one list that the worker threads take from in order, and the calling thread sends a slice off as
  // soon as its last task is done: the copy engine moves slice k while the threads fill the slices behind it,
  // and a stripe of compressed chunks inflates on all threads instead of on as many as a slice has chunks.
  constexpr uint64_t kSlice = 32ull << 20, kPiece = 2ull << 20;
  struct Task { size_t chunk, slice; uint64_t off, len; };
  struct Slice { size_t first; uint64_t bytes; };
  std::vector<Task> tasks;
  std::vector<Slice> slices;
  for (size_t first = 0; first < chunks.size();) {
    size_t last = first;
    uint64_t bytes = 0;
    while (last < chunks.size() && (bytes == 0 || bytes + sizes[last] <= kSlice)) bytes += (sizes[last++] + 15) & ~uint64_t(15);
    for (size_t i = first; i < last; ++i) {
      if (inflate[i]) { tasks.push_back(Task{i, slices.size(), 0, 0}); continue; }
      for (uint64_t at = 0; at < chunks[i].size(); at += kPiece)
        tasks.push_back(Task{i, slices.size(), at, std::min<uint64_t>(kPiece, chunks[i].size() - at)});
    }
    slices.push_back(Slice{first, bytes});
    first = last;
  }
  std::unique_ptr<std::atomic<uint32_t>[]> left(new std::atomic<uint32_t>[slices.size()]);  // tasks of the slice not yet done
  for (size_t k = 0; k < slices.size(); ++k) left[k].store(0, std::memory_order_relaxed);
  for (const Task& t : tasks) left[t.slice].fetch_add(1, std::memory_order_relaxed);
  auto run_task = [&](const Task& task) {
    const size_t i = task.chunk;
    const auto t = StatClock::now();
    if (inflate[i]) {
      uint64_t written = 0;
      check(nbl_chunk_inflate(reinterpret_cast<const uint8_t*>(chunks[i].data()), chunks[i].size(), &o, b.staging.get() + offs[i],
                              sizes[i], &written));
    } else {
      memcpy(b.staging.get() + offs[i] + task.off, chunks[i].data() + task.off, task.len);
    }
    if (timer.stats) {
      const uint64_t ns = nanosSince(t);
      timer.taskNanos.fetch_add(ns, std::memory_order_relaxed);
      if (inflate[i]) timer.stats->decompressCPUTimeNanos.increment(ns);
    }
  };
  auto send = [&](const Slice& sl) {
    hipCheck(hipMemcpyAsync(static_cast<uint8_t*>(slot.arena.ptr) + offs[sl.first], b.staging.get() + offs[sl.first], sl.bytes,
                            hipMemcpyHostToDevice, slot.stream), "H2D");
  };
  const auto phase = StatClock::now();
  const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
  const size_t helpers = std::min<size_t>({tasks.size(), (size_t)std::min(hw, 16u), (size_t)(total >> 21)});  // >= 2 MiB per thread
  if (helpers <= 1) {
    size_t k = 0;
    for (size_t s = 0; s < slices.size(); ++s) {
      for (; k < tasks.size() && tasks[k].slice == s; ++k) run_task(tasks[k]);
      send(slices[s]);
    }
  } else {
    std::atomic<size_t> next{0};
    std::atomic<bool> failed{false};
    std::exception_ptr error;
    std::mutex errorMu;
    const std::thread::id caller = std::this_thread::get_id();
    auto fail = [&] {
      std::lock_guard<std::mutex> lock(errorMu);
      if (!error) error = std::current_exception();
      failed.store(true, std::memory_order_release);
    };
    parallelRun(helpers + 1, [&] {
      try {
        if (std::this_thread::get_id() == caller) {  // sends the slices off, in order, as they fill
          for (size_t s = 0; s < slices.size(); ++s) {
            while (left[s].load(std::memory_order_acquire) != 0) {
              if (failed.load(std::memory_order_acquire)) return;
              std::this_thread::yield();
            }
            send(slices[s]);
          }
        } else {
          for (size_t k = next++; k < tasks.size() && !failed.load(std::memory_order_relaxed); k = next++) {
            run_task(tasks[k]);
            left[tasks[k].slice].fetch_sub(1, std::memory_order_release);
          }
        }
      } catch (...) {
        fail();
      }
    });
    if (error) std::rethrow_exception(error);
  }
  timer.phaseNanos += nanosSince(phase);
  const auto planBegin = StatClock::now();
  check(nbl_plan_batch(b.staging.get(), offs.data(), sizes.data(), (uint32_t)chunks.size(), &o, &b.plan.p));
  if (timer.stats) timer.stats->planNanos.increment(nanosSince(planBegin));
  stagedBytes = total;
  check(nbl_plan_get_info(b.plan.p, &b.info));
  if (b.info.arena_bytes > slot.arena.cap) throw NimbleError(NBL_ERR_ARGUMENT, "arena smaller than the plan asks for");
  return b;
}

// A batch that fails half way leaves copies and kernels queued that still use its pinned staging block and the
// slot's buffers: they are waited for before the error leaves the call (and the block goes back to the pool).
struct DrainOnError {
  GpuContext::Slot& slot;
  bool armed = true;
  ~DrainOnError() {
    if (armed) (void)hipStreamSynchronize(slot.stream);
  }
};

// plan + H2D + decode + D2H for chunks given as (pointer, size) pairs.
std::vector<std::unique_ptr<Encoding>> decodeChunks(GpuContext& ctx, const std::vector<std::string_view>& chunks,
                                                    const GpuEncodingFactory::StringBufferFactory& sbf, const Options& opts) {
  std::vector<std::unique_ptr<Encoding>> result;
  if (chunks.empty()) return result;
  StagedBatch b;
  std::shared_ptr<OutputCopy> hostOut;
  {
    GpuContext::Slot* slotp = nullptr;
    std::unique_lock<std::mutex> lock = ctx.acquire(slotp);
    GpuContext::Slot& slot = *slotp;
    BatchTimer timer(opts.decodingStats);
    uint64_t stagedBytes = 0;
    DrainOnError drain{slot};
    b = stageBatch(ctx, slot, chunks, opts, timer, stagedBytes);
    hostOut = std::make_shared<OutputCopy>(ctx.pinned->acquire(b.info.out_bytes));
    slot.out.reserve(b.info.out_bytes, {ctx.copyOut, slot.stream});
    slot.work.reserve(b.info.work_bytes, {slot.stream});
    check(nbl_plan_upload(b.plan.p, slot.work.ptr, slot.stream));
    if (slot.outBusy) hipCheck(hipStreamWaitEvent(slot.stream, slot.outFree, 0), "hipStreamWaitEvent");
    timer.deviceBegin(slot);
    check(nbl_decode_batch_device(b.plan.p, slot.arena.ptr, slot.out.ptr, slot.work.ptr, slot.stream));
    timer.deviceEnd(slot);
    // synchronises the slot's stream: inconsistent payloads are reported before anything is handed out
    check(nbl_decode_status(b.plan.p, slot.work.ptr, slot.stream));
    drain.armed = false;
    timer.finish(slot, b.info, stagedBytes);
    // the copy back runs on while the caller already reads the first chunks and
    // while the next batch is copied in and decoded on the other slot; the decode
    // of the batch after that, on this slot again, waits for outFree (the slot's
    // stream is idle here: the status call has synchronised it)
    hostOut->enqueue(slot.out.ptr, b.info.out_bytes, ctx.copyOut);
    hipCheck(hipEventRecord(slot.outFree, ctx.copyOut), "hipEventRecord");
    slot.outBusy = true;
  }
  result.reserve(chunks.size());
  for (uint32_t c = 0; c < chunks.size(); ++c) {
    nbl_chunk_layout lay;
    check(nbl_plan_get_chunk_layout(b.plan.p, c, &lay));
    result.push_back(std::make_unique<GpuEncoding>(lay, hostOut, sbf));
  }
  return result;
}

class DeviceBatchImpl final : public DeviceBatch {
 public:
  size_t chunkCount() const override { return layouts.size(); }
  const nbl_chunk_layout& layout(size_t chunk) const override { return layouts.at(chunk); }
  const void* data() const override { return out.ptr; }
  uint64_t bytes() const override { return outBytes; }
  int device() const override { return ctx->device_; }
  void* stream() const override { return stream_; }
  std::shared_ptr<GpuContext> ctx;
  hipStream_t stream_ = nullptr;
  DeviceBuffer out;
  uint64_t outBytes = 0;
  std::vector<nbl_chunk_layout> layouts;
};

std::unique_ptr<DeviceBatch> decodeChunksToDevice(const std::shared_ptr<GpuContext>& ctxp, const std::vector<std::string_view>& chunks,
                                                  const Options& opts) {
  GpuContext& ctx = *ctxp;
  auto batch = std::make_unique<DeviceBatchImpl>();
  batch->ctx = ctxp;
  if (chunks.empty()) return batch;
  StagedBatch b;
  {
    GpuContext::Slot* slotp = nullptr;
    std::unique_lock<std::mutex> lock = ctx.acquire(slotp);
    GpuContext::Slot& slot = *slotp;
    BatchTimer timer(opts.decodingStats);
    uint64_t stagedBytes = 0;
    DrainOnError drain{slot};
    b = stageBatch(ctx, slot, chunks, opts, timer, stagedBytes);
    slot.work.reserve(b.info.work_bytes, {slot.stream});
    batch->out.reserve(b.info.out_bytes);   // owned by the batch: it outlives this call
    batch->outBytes = b.info.out_bytes;
    batch->stream_ = slot.stream;
    check(nbl_plan_upload(b.plan.p, slot.work.ptr, slot.stream));
    timer.deviceBegin(slot);
    check(nbl_decode_batch_device(b.plan.p, slot.arena.ptr, batch->out.ptr, slot.work.ptr, slot.stream));
    timer.deviceEnd(slot);
    check(nbl_decode_status(b.plan.p, slot.work.ptr, slot.stream));  // synchronises the slot's stream
    drain.armed = false;
    timer.finish(slot, b.info, stagedBytes);
  }
  batch->layouts.resize(chunks.size());
  for (uint32_t c = 0; c < chunks.size(); ++c) check(nbl_plan_get_chunk_layout(b.plan.p, c, &batch->layouts[c]));
  return batch;
}

}  // namespace

GpuEncodingFactory::GpuEncodingFactory(Options options, int device)
    : options_(options), ctx_(std::make_shared<GpuContext>(device)) {}
GpuEncodingFactory::~GpuEncodingFactory() = default;

std::unique_ptr<Encoding> GpuEncodingFactory::create(std::string_view data, StringBufferFactory sbf) const {
  return create(data, std::move(sbf), options_);
}

std::unique_ptr<Encoding> GpuEncodingFactory::create(std::string_view data, StringBufferFactory sbf, const Options& options) const {
  try {
    auto v = decodeChunks(*ctx_, {data}, sbf, options);
    return std::move(v[0]);
  } catch (const NimbleUnsupported&) {
    if (fallback_) return fallback_(data, sbf);
    throw;
  }
}

std::string toString(EncodingType t) {
  static const char* const names[] = {"Trivial", "RLE", "Dictionary", "FixedBitWidth", "Sentinel", "Nullable", "SparseBool", "Varint",
                                      "Delta", "Constant", "MainlyConstant", "Prefix", "ALP", "PFOR", "SimdForBitpack", "BlockBitPacking",
                                      "SubIntSplit", "FrequencyPartition", "FOR", "Fsst", "Huffman", "DeltaBlock", "SharedDictionary"};
  const unsigned v = static_cast<uint8_t>(t);
  return v < sizeof names / sizeof names[0] ? names[v] : "Unknown encoding type: " + std::to_string(v);
}
std::string toString(DataType t) {
  static const char* const names[] = {nullptr, "Int8", "Uint8", "Int16", "Uint16", "Int32", "Uint32", "Int64", "Uint64", "Float", "Double",
                                      "Bool", "String"};
  const unsigned v = static_cast<uint8_t>(t);
  return v >= 1 && v < sizeof names / sizeof names[0] ? names[v] : "Unknown data type: " + std::to_string(v);
}

bool supportsEncodingView(EncodingType t) {
  switch (t) {
    case EncodingType::Constant: case EncodingType::Trivial: case EncodingType::MainlyConstant:
    case EncodingType::FixedBitWidth: case EncodingType::Dictionary: case EncodingType::SparseBool:
    case EncodingType::RLE: case EncodingType::FOR: case EncodingType::BlockBitPacking:
      return true;
    default:
      return false;
  }
}

std::unique_ptr<EncodingView> GpuEncodingFactory::createEncodingView(std::string_view data, const Options& options) const {
  if (data.empty()) throw NimbleError(NBL_ERR_CORRUPT, "encoding shorter than its prefix");
  const auto type = static_cast<EncodingType>(static_cast<uint8_t>(data[0]));
  if (!supportsEncodingView(type))
    throw NimbleUnsupported(NBL_ERR_UNSUPPORTED, "EncodingType " + std::to_string((int)type) + " does not support EncodingView.");
  auto strings = std::make_shared<std::vector<std::unique_ptr<char[]>>>();
  auto sbf = [strings](uint32_t bytes) -> void* {
    strings->push_back(std::make_unique<char[]>(bytes ? bytes : 1));
    return strings->back().get();
  };
  Options o = options;
  o.keepDictionary = false;
  auto v = decodeChunks(*ctx_, {data}, sbf, o);  // no CPU fallback: a view is a GPU-decoded chunk or an error
  return std::make_unique<GpuEncodingView>(std::move(v[0]), std::move(strings));
}

std::vector<std::unique_ptr<Encoding>> GpuEncodingFactory::createBatch(const std::vector<std::string_view>& chunks,
                                                                       StringBufferFactory sbf, const Options& options) const {
  try {
    return decodeChunks(*ctx_, chunks, sbf, options);
  } catch (const NimbleUnsupported&) {
    if (!fallback_) throw;
  }
  // some chunk needs the CPU factory: decode one by one so only that chunk falls back
  std::vector<std::unique_ptr<Encoding>> out;
  for (auto c : chunks) out.push_back(create(c, sbf, options));
  return out;
}

std::unique_ptr<DeviceBatch> GpuEncodingFactory::decodeToDevice(const std::vector<std::string_view>& chunks, const Options& options) const {
  return decodeChunksToDevice(ctx_, chunks, options);
}

// ------------------------------------------------------------ stripe decoder
namespace {

class GpuStreamDecoder final : public Decoder {
 public:
  std::vector<std::unique_ptr<Encoding>> chunks;           // decoded up front by the stripe batch
  std::vector<std::shared_ptr<void>> strings;               // keeps the string bytes of chunk i alive

  uint32_t next(uint32_t count, void* output, std::vector<std::shared_ptr<void>>& stringBuffers,
                std::function<void*()> getOutputNulls, const Bitmap* scatter) override {
    // ChunkedStreamDecoder::next, velox/ChunkedStreamDecoder.cpp:75-176
    if (count == 0) {
      if (getOutputNulls && scatter) memset(getOutputNulls(), 0, (scatter->size + 7) / 8);
      return 0;
    }
    uint32_t nonNullCount = 0, offset = 0;
    bool hasNulls = false;
    void* nullsPtr = nullptr;
    std::function<void*()> initNulls = [&]() {
      if (!nullsPtr) nullsPtr = getOutputNulls();
      return nullsPtr;
    };
    while (count > 0) {
      ensureLoaded();
      Encoding& enc = *chunks[cur_];
      const uint32_t rows = std::min(count, remaining_);
      uint32_t chunkNonNull = 0, endOffset = 0;
      if (!getOutputNulls || !scatter) {
        if (scatter) throw NimbleError(NBL_ERR_ARGUMENT, "Unexpected scatter output bitmap");
        chunkNonNull = enc.materializeNullable(rows, output, initNulls, nullptr, offset);
        endOffset = offset + rows;
      } else {
        endOffset = findSetBit(scatter->bits, offset, scatter->size, rows + 1);
        Bitmap local{scatter->bits, endOffset};
        chunkNonNull = enc.materializeNullable(rows, output, initNulls, &local, offset);
      }
      const bool chunkHasNulls = chunkNonNull != endOffset - offset;
      if (chunkHasNulls && !hasNulls) fillBits(nullsPtr, 0, offset, true);           // back-fill, :129-134
      hasNulls = hasNulls || chunkHasNulls;
      if (hasNulls && !chunkHasNulls) fillBits(nullsPtr, offset, endOffset, true);   // :136-140
      offset = endOffset;
      remaining_ -= rows;
      count -= rows;
      nonNullCount += chunkNonNull;
      // every chunk's string bytes stay alive through the buffers handed back (:165-172)
      if (cur_ < strings.size() && strings[cur_]) stringBuffers.push_back(strings[cur_]);
    }
    return nonNullCount;
  }
  void skip(uint32_t count) override {  // :178-186
    while (count > 0) {
      ensureLoaded();
      const uint32_t n = std::min(count, remaining_);
      chunks[cur_]->skip(n);
      count -= n;
      remaining_ -= n;
    }
  }
  void reset() override {  // :188-191
    next_ = 0;
    remaining_ = 0;
  }
  const Encoding* encoding() const override { return cur_ < chunks.size() ? chunks[cur_].get() : nullptr; }

 private:
  void ensureLoaded() {  // :193-205
    if (remaining_ != 0) return;
    if (next_ >= chunks.size()) throw NimbleError(NBL_ERR_CORRUPT, "Read beyond end of stream");
    cur_ = next_++;
    chunks[cur_]->reset();
    remaining_ = chunks[cur_]->rowCount();
    if (remaining_ == 0) throw NimbleError(NBL_ERR_CORRUPT, "Empty chunk");
  }
  size_t cur_ = 0, next_ = 0;
  uint32_t remaining_ = 0;
};

// velox::bits::scatterBits in place: source bit k moves to the k-th set position
// of `mask`, the other positions of [0, numTarget) become 0
void scatterBitsInPlace(int64_t numSource, int64_t numTarget, void* bits, const void* mask) {
  int64_t src = numSource - 1;
  for (int64_t pos = numTarget - 1; pos >= 0; --pos) {
    if (bitIsSet(mask, (uint64_t)pos)) {
      const bool b = src >= 0 && bitIsSet(bits, (uint64_t)src);
      --src;
      b ? bitSet(bits, (uint64_t)pos) : bitClear(bits, (uint64_t)pos);
    } else {
      bitClear(bits, (uint64_t)pos);
    }
  }
}

class GpuChunkedDecoder final : public ChunkedDecoder {
 public:
  GpuChunkedDecoder(std::vector<std::unique_ptr<Encoding>>& chunks, bool decodeValuesWithNulls)
      : chunks_(chunks), decodeValuesWithNulls_(decodeValuesWithNulls) {}

  void skip(int64_t numValues) override {
    if (numValues < 0) throw NimbleError(NBL_ERR_ARGUMENT, "negative skip");
    while (numValues > 0) {
      if (remaining_ == 0) loadNextChunk();
      if (numValues < remaining_) {
        cur().skip((uint32_t)numValues);
        remaining_ -= numValues;
        break;
      }
      numValues -= remaining_;
      remaining_ = 0;
    }
  }

  void nextBools(uint64_t* data, int64_t count, const uint64_t* incomingNulls) override {
    if (decodeValuesWithNulls_) {
      if (nullableValues_.size() < (size_t)count) nullableValues_.assign((size_t)count, 0);
      decodeNullable(data, nullableValues_.data(), count, incomingNulls);
      return;
    }
    const int64_t total = incomingNulls ? (int64_t)countBits(incomingNulls, 0, (uint64_t)count) : count;
    for (int64_t i = 0; i < total;) {
      if (remaining_ == 0) {
        loadNextChunk();
        if (cur().dataType() != DataType::Bool) throw NimbleError(NBL_ERR_ARGUMENT, "nextBools on a non-bool stream");
      }
      const int64_t n = std::min(total - i, remaining_);
      cur().materializeBoolsAsBits((uint32_t)n, data, (int)i);
      remaining_ -= n;
      i += n;
    }
    if (incomingNulls) scatterBitsInPlace(total, count, data, incomingNulls);
  }

  void nextIndices(int32_t* data, int64_t count, const uint64_t* incomingNulls) override {
    std::vector<uint64_t> nulls;
    if (incomingNulls) nulls.assign(((size_t)count + 63) / 64 + 1, 0);
    decodeNullable(incomingNulls ? nulls.data() : nullptr, data, count, incomingNulls);
  }

  void decodeNullable(uint64_t* nulls, void* data, int64_t count, const uint64_t* incomingNulls) override {
    const int64_t total = incomingNulls ? (int64_t)countBits(incomingNulls, 0, (uint64_t)count) : count;
    auto getNulls = [&]() -> void* {
      if (!nulls) throw NimbleError(NBL_ERR_ARGUMENT, "the stream has nulls but no null buffer was passed");
      return nulls;
    };
    if (incomingNulls) {
      if (total == 0) {
        if (nulls) fillBits(nulls, 0, (uint64_t)count, false);
        return;
      }
      uint32_t offset = 0;
      for (int64_t i = 0; i < total;) {
        if (remaining_ == 0) loadNextChunk();
        const int64_t n = std::min(total - i, remaining_);
        const uint32_t endOffset = findSetBit(incomingNulls, offset, (uint32_t)count, (uint32_t)n + 1);
        Bitmap local{incomingNulls, endOffset};
        const uint32_t nonNull = cur().materializeNullable((uint32_t)n, data, getNulls, &local, offset);
        if (nulls && nonNull == endOffset - offset) fillBits(nulls, offset, endOffset, true);
        remaining_ -= n;
        i += n;
        offset = endOffset;
      }
    } else {
      for (int64_t i = 0; i < total;) {
        if (remaining_ == 0) loadNextChunk();
        const int64_t n = std::min(total - i, remaining_);
        const uint32_t nonNull = cur().materializeNullable((uint32_t)n, data, getNulls, nullptr, (uint32_t)i);
        if (nulls && nonNull == (uint32_t)n) fillBits(nulls, (uint64_t)i, (uint64_t)(i + n), true);
        remaining_ -= n;
        i += n;
      }
    }
  }

  const uint32_t* preloadedValues() const override {
    if (!decodeValuesWithNulls_) throw NimbleError(NBL_ERR_ARGUMENT, "Only Array and Map types preload offset values with nulls");
    return nullableValues_.data();
  }
  void ensureLoaded() override {
    if (loaded_ && remaining_ != 0) return;
    if (next_ < chunks_.size()) loadNextChunk();
  }
  const Encoding* currentEncoding() const override { return loaded_ ? chunks_[cur_].get() : nullptr; }
  bool dictionaryConvertible() const override {
    if (!loaded_) throw NimbleError(NBL_ERR_ARGUMENT, "dictionaryConvertible before ensureLoaded");
    return chunks_[cur_]->dictionaryEnabled();
  }
  int64_t remainingValues() const override { return remaining_; }

 private:
  Encoding& cur() { return *chunks_[cur_]; }
  void loadNextChunk() {  // ChunkedDecoder.cpp:33-85 (the chunk itself was decoded with the stripe)
    if (next_ >= chunks_.size()) throw NimbleError(NBL_ERR_CORRUPT, "Failed to read chunk header");
    cur_ = next_++;
    loaded_ = true;
    chunks_[cur_]->reset();
    remaining_ = chunks_[cur_]->rowCount();
    if (remaining_ == 0) throw NimbleError(NBL_ERR_CORRUPT, "Empty chunk");
  }
  std::vector<std::unique_ptr<Encoding>>& chunks_;
  const bool decodeValuesWithNulls_;
  size_t cur_ = 0, next_ = 0;
  bool loaded_ = false;
  int64_t remaining_ = 0;
  std::vector<uint32_t> nullableValues_;
};

}  // namespace

struct GpuStripeDecoder::Impl {
  std::vector<std::unique_ptr<GpuStreamDecoder>> streams;
  std::vector<std::unique_ptr<GpuChunkedDecoder>> chunked;  // created on demand, one per stream
};

GpuStripeDecoder::GpuStripeDecoder(const GpuEncodingFactory& factory, const std::vector<std::string_view>& streams)
    : impl_(std::make_unique<Impl>()) {
  std::vector<std::string_view> chunks;
  std::vector<size_t> owner;
  std::vector<std::unique_ptr<char[]>> uncompressed;  // chunks whose header names Zstd / LZ4 (ChunkedStream.cpp:49-61)
  for (size_t s = 0; s < streams.size(); ++s) {
    uint32_t n = 0;
    const auto* p = reinterpret_cast<const uint8_t*>(streams[s].data());
    check(nbl_split_stream_ex(p, streams[s].size(), nullptr, nullptr, nullptr, 0, &n));
    std::vector<uint64_t> offs(n ? n : 1), sizes(n ? n : 1);
    std::vector<uint8_t> comp(n ? n : 1);
    check(nbl_split_stream_ex(p, streams[s].size(), offs.data(), sizes.data(), comp.data(), n, &n));
    for (uint32_t i = 0; i < n; ++i) {
      if (comp[i] == 0) {
        chunks.emplace_back(streams[s].data() + offs[i], sizes[i]);
      } else {
        uint64_t usize = 0;
        check(nbl_uncompressed_size(comp[i], p + offs[i], sizes[i], &usize));
        uncompressed.push_back(std::make_unique<char[]>(usize ? usize : 1));
        check(nbl_uncompress(comp[i], p + offs[i], sizes[i], reinterpret_cast<uint8_t*>(uncompressed.back().get()), usize));
        chunks.emplace_back(uncompressed.back().get(), usize);
      }
      owner.push_back(s);
    }
    impl_->streams.push_back(std::make_unique<GpuStreamDecoder>());
  }
  // String bytes of the whole batch live in buffers owned by one keep-alive
  // object that every string chunk of every stream shares (coarser than the
  // reference's per-chunk lifetime, never shorter).
  auto keep = std::make_shared<std::vector<std::unique_ptr<char[]>>>();
  auto sbf = [keep](uint32_t bytes) -> void* {
    keep->push_back(std::make_unique<char[]>(bytes ? bytes : 1));
    return keep->back().get();
  };
  auto encs = factory.createBatch(chunks, sbf, factory.options());
  for (size_t c = 0; c < encs.size(); ++c) {
    auto& sd = *impl_->streams[owner[c]];
    const bool isString = encs[c]->dataType() == DataType::String;
    sd.chunks.push_back(std::move(encs[c]));
    sd.strings.push_back(isString ? std::shared_ptr<void>(keep, keep.get()) : std::shared_ptr<void>());
  }
}
GpuStripeDecoder::~GpuStripeDecoder() = default;
size_t GpuStripeDecoder::streamCount() const { return impl_->streams.size(); }
Decoder& GpuStripeDecoder::decoder(size_t stream) { return *impl_->streams.at(stream); }
ChunkedDecoder& GpuStripeDecoder::chunkedDecoder(size_t stream, bool decodeValuesWithNulls) {
  if (impl_->chunked.size() < impl_->streams.size()) impl_->chunked.resize(impl_->streams.size());
  auto& slot = impl_->chunked.at(stream);
  if (!slot) slot = std::make_unique<GpuChunkedDecoder>(impl_->streams.at(stream)->chunks, decodeValuesWithNulls);
  return *slot;
}

}  // namespace nimble_amd

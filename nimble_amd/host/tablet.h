// Read-only view of a Nimble file ("tablet") held in memory: postscript, footer,
// stripes and stripe groups, enough to hand every stream of a stripe to the
// GPU stripe decoder.  Mirrors the read side of
//   dwio/nimble/tablet/Postscript.cpp:38-65        (20-byte postscript)
//   dwio/nimble/tablet/Footer.fbs                   (Footer, Stripes, StripeGroup, MetadataSection)
//   dwio/nimble/tablet/TabletReader.cpp:951-1018    (stream ranges of a stripe)
//   dwio/nimble/tablet/StripeGroup.cpp:186-220      (raw stripe-group indexing)
// FlatBuffers is not available to this build: the tables are read with a small
// bounds-checked vtable walker.  Not covered: the stream-major stripe-group
// layout, compressed metadata sections, checksums, the schema section.
#pragma once

#include <cstdint>
#include <memory>
#include <string>
#include <string_view>
#include <vector>

#include "nimble_gpu.h"

namespace nimble_amd {

class Tablet {
 public:
  // `file` must stay alive while the Tablet is used.  Throws NimbleError
  // (NBL_ERR_CORRUPT / NBL_ERR_UNSUPPORTED) on malformed or unsupported input.
  explicit Tablet(std::string_view file);

  uint16_t majorVersion() const { return major_; }
  uint16_t minorVersion() const { return minor_; }
  uint64_t rowCount() const { return rowCount_; }
  uint32_t stripeCount() const { return (uint32_t)stripeOffsets_.size(); }
  uint32_t stripeRowCount(uint32_t stripe) const { return stripeRows_.at(stripe); }
  uint64_t stripeOffset(uint32_t stripe) const { return stripeOffsets_.at(stripe); }
  uint32_t stripeSize(uint32_t stripe) const { return stripeSizes_.at(stripe); }
  // number of stream slots of the stripe's group (TabletReader::streamCount)
  uint32_t streamCount(uint32_t stripe) const;
  // bytes of one stream (chunk headers included); empty when the stream is absent in that stripe
  std::string_view stream(uint32_t stripe, uint32_t streamId) const;
  // every non-empty stream of the stripe with its id, in id order
  std::vector<std::pair<uint32_t, std::string_view>> streams(uint32_t stripe) const;
  // an optional metadata section by name (e.g. "columnar.schema"), empty if absent; uncompressed sections only
  std::string_view optionalSection(std::string_view name) const;

 private:
  struct Group {
    uint32_t firstStripe = 0, stripeCount = 0, streamCount = 0;
    std::vector<uint32_t> offsets, sizes;  // [stripe in group][stream]
  };
  std::string_view file_;
  uint16_t major_ = 0, minor_ = 0;
  uint64_t rowCount_ = 0;
  std::vector<uint32_t> stripeRows_, stripeSizes_, stripeGroup_;
  std::vector<uint64_t> stripeOffsets_;
  std::vector<Group> groups_;
  std::vector<std::pair<std::string, std::string_view>> optional_;
};

// Every stripe of a file decoded across several GPUs.  A stripe is a
// self-contained set of streams (dwio/nimble/tablet/TabletReader.h:210-219) and
// an Encoding reads nothing outside its own bytes, so nothing moves between
// devices: stripes are dealt to the devices by longest-processing-time-first
// over their byte sizes (the footer's Stripes.sizes, tablet/Footer.fbs:26-31),
// each device has a GpuEncodingFactory and a host thread of its own that walks
// its stripes in ascending order (plan, stage, decode, copy back), and the
// stripes are handed out by index, i.e. in row order, as the reference reader
// walks them (VeloxReader::loadNextStripe, velox/VeloxReader.cpp:299-353; files
// are split across readers by stripe, :229-273).  Completion is a join; there is
// no collective.  A device may be named more than once (two contexts on one GPU).
class GpuTabletDecoder {
 public:
  // `tablet` (and the file bytes it views) must outlive this object.  The device
  // threads start at once and run at most `lookahead` stripes per device ahead
  // of the lowest stripe not yet taken (0 = no bound: the whole file is decoded
  // and held in host memory until taken).
  GpuTabletDecoder(const Tablet& tablet, std::vector<int> devices, Options options = {}, uint32_t lookahead = 2);
  ~GpuTabletDecoder();
  GpuTabletDecoder(const GpuTabletDecoder&) = delete;
  GpuTabletDecoder& operator=(const GpuTabletDecoder&) = delete;

  // Per device, its stripes in ascending order: stripes in decreasing size, each
  // to the device with the fewest bytes so far (ties: fewest stripes, then the
  // lowest device).  Deterministic: any process computes the same assignment
  // from the footer without communicating.
  static std::vector<std::vector<uint32_t>> assignStripes(const std::vector<uint64_t>& stripeSizes, size_t devices);
  const std::vector<std::vector<uint32_t>>& assignment() const { return assignment_; }
  int deviceOf(uint32_t stripe) const { return devices_.at(owner_.at(stripe)); }
  uint32_t stripeCount() const { return (uint32_t)owner_.size(); }

  // Stream ids of the stripe's non-empty streams, in the order of the decoders
  // of the GpuStripeDecoder that take() returns.
  std::vector<uint32_t> streamIds(uint32_t stripe) const;
  // Blocks until the stripe is decoded and hands it over (once per stripe; take
  // them in ascending order when a look-ahead bound is set).  Rethrows what the
  // stripe's decode threw.
  std::unique_ptr<GpuStripeDecoder> take(uint32_t stripe);

 private:
  struct State;
  void startThreads();
  const Tablet& tablet_;
  std::vector<int> devices_;
  std::vector<std::vector<uint32_t>> assignment_;
  std::vector<uint32_t> owner_;  // per stripe: index into devices_
  std::unique_ptr<State> state_;
};

}  // namespace nimble_amd

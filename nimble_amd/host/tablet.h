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

}  // namespace nimble_amd

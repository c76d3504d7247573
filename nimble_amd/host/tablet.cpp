// Tablet reader (see tablet.h).
#include "tablet.h"

#include <cstring>

namespace nimble_amd {

namespace {

constexpr uint16_t kMagicNumber = 0xA1FA;   // tablet/Constants.h:22
constexpr uint32_t kPostscriptSize = 20;    // tablet/Constants.h:29

[[noreturn]] void corrupt(const std::string& what) { throw NimbleError(NBL_ERR_CORRUPT, what); }

// Minimal read-only FlatBuffers access: every read is bounds checked against the buffer.
struct Fb {
  const uint8_t* p;
  uint64_t n;
  template <typename T>
  T rd(uint64_t at) const {
    if (at + sizeof(T) > n) corrupt("flatbuffer read beyond end of metadata");
    T v;
    memcpy(&v, p + at, sizeof(T));
    return v;
  }
  struct Table {
    const Fb* fb;
    uint64_t pos = 0, vt = 0;
    uint16_t vtSize = 0;
    // absolute position of field `i`, 0 when absent
    uint64_t field(uint32_t i) const {
      const uint64_t slot = 4 + 2ull * i;
      if (slot + 2 > vtSize) return 0;
      const uint16_t off = fb->rd<uint16_t>(vt + slot);
      return off ? pos + off : 0;
    }
    template <typename T>
    T scalar(uint32_t i, T dflt) const {
      const uint64_t at = field(i);
      return at ? fb->rd<T>(at) : dflt;
    }
    bool has(uint32_t i) const { return field(i) != 0; }
    uint64_t indirect(uint32_t i) const {  // target of an offset field (table / vector / string)
      const uint64_t at = field(i);
      return at ? at + fb->rd<uint32_t>(at) : 0;
    }
  };
  Table table(uint64_t pos) const {
    Table t{this, pos, 0, 0};
    const int32_t so = rd<int32_t>(pos);
    const int64_t vt = (int64_t)pos - so;
    if (vt < 0 || (uint64_t)vt + 4 > n) corrupt("flatbuffer vtable out of range");
    t.vt = (uint64_t)vt;
    t.vtSize = rd<uint16_t>(t.vt);
    if (t.vtSize < 4 || t.vt + t.vtSize > n) corrupt("flatbuffer vtable out of range");
    return t;
  }
  Table root() const { return table(rd<uint32_t>(0)); }
  // vector at `pos` (0 = absent): element count and first element
  uint32_t vecLen(uint64_t pos) const { return pos ? rd<uint32_t>(pos) : 0; }
  // the same, refusing a count whose elements (of `elemBytes` each) would not fit the buffer
  uint32_t vecLen(uint64_t pos, uint32_t elemBytes) const {
    const uint32_t len = vecLen(pos);
    if (len && pos + 4 + (uint64_t)len * elemBytes > n) corrupt("flatbuffer vector beyond end of metadata");
    return len;
  }
  template <typename T>
  std::vector<T> scalars(uint64_t pos) const {
    const uint32_t len = vecLen(pos, sizeof(T));
    std::vector<T> v(len);
    if (len) memcpy(v.data(), p + pos + 4, (size_t)len * sizeof(T));
    return v;
  }
  uint64_t vecOffsetElem(uint64_t pos, uint32_t i) const {  // element i of a vector of offsets
    const uint64_t at = pos + 4 + 4ull * i;
    return at + rd<uint32_t>(at);
  }
  std::string_view str(uint64_t pos) const {
    const uint32_t len = rd<uint32_t>(pos);
    if (pos + 4 + (uint64_t)len > n) corrupt("flatbuffer string beyond end of metadata");
    return {reinterpret_cast<const char*>(p + pos + 4), len};
  }
};

struct Section {
  uint64_t offset = 0;
  uint32_t size = 0;
  uint8_t compression = 0;
};
Section readSection(const Fb::Table& t) {  // table MetadataSection
  return Section{t.scalar<uint64_t>(0, 0), t.scalar<uint32_t>(1, 0), t.scalar<uint8_t>(2, 0)};
}

}  // namespace

Tablet::Tablet(std::string_view file) : file_(file) {
  if (file.size() < kPostscriptSize) corrupt("Invalid postscript length");
  const auto* end = reinterpret_cast<const uint8_t*>(file.data()) + file.size();
  auto rd16 = [](const uint8_t* p) { uint16_t v; memcpy(&v, p, 2); return v; };
  // Postscript::parse: ... u32 footerSize | u8 footerCompression | u8 checksumType | u64 checksum | u16 major | u16 minor | u16 magic
  if (rd16(end - 2) != kMagicNumber) corrupt("Magic number mismatch. Not a nimble file!");
  major_ = rd16(end - 6);
  minor_ = rd16(end - 4);
  if (major_ > 0) throw NimbleUnsupported(NBL_ERR_UNSUPPORTED, "Unsupported file version");
  uint32_t footerSize;
  memcpy(&footerSize, end - 20, 4);
  const uint8_t footerCompression = *(end - 16);
  if (footerCompression != 0) throw NimbleUnsupported(NBL_ERR_UNSUPPORTED, "compressed footer");
  if ((uint64_t)footerSize + kPostscriptSize > file.size()) corrupt("footer larger than the file");
  const Fb footer{end - kPostscriptSize - footerSize, footerSize};
  const Fb::Table ft = footer.root();
  rowCount_ = ft.scalar<uint64_t>(0, 0);

  auto sectionBytes = [&](const Section& s, const char* what) {
    if (s.compression != 0) throw NimbleUnsupported(NBL_ERR_UNSUPPORTED, std::string("compressed metadata section: ") + what);
    if (s.offset > file_.size() || s.size > file_.size() - s.offset) corrupt(std::string(what) + " section outside the file");
    return Fb{reinterpret_cast<const uint8_t*>(file_.data()) + s.offset, s.size};
  };

  if (ft.has(1)) {  // Footer.stripes
    const Fb st = sectionBytes(readSection(footer.table(ft.indirect(1))), "stripes");
    const Fb::Table t = st.root();
    stripeRows_ = st.scalars<uint32_t>(t.indirect(0));
    stripeOffsets_ = st.scalars<uint64_t>(t.indirect(1));
    stripeSizes_ = st.scalars<uint32_t>(t.indirect(2));
    stripeGroup_ = st.scalars<uint32_t>(t.indirect(3));
    if (stripeRows_.size() != stripeOffsets_.size() || stripeSizes_.size() != stripeOffsets_.size() ||
        stripeGroup_.size() != stripeOffsets_.size())
      corrupt("stripes metadata vectors differ in length");
    for (size_t i = 0; i < stripeOffsets_.size(); ++i)
      if (stripeOffsets_[i] > file_.size() || stripeSizes_[i] > file_.size() - stripeOffsets_[i]) corrupt("stripe outside the file");
  }
  const uint64_t groupsVec = ft.indirect(2);  // Footer.stripe_groups
  const uint32_t nGroups = footer.vecLen(groupsVec, 4);
  groups_.resize(nGroups);
  std::vector<uint32_t> firstStripeOf(nGroups, UINT32_MAX);
  uint64_t tableEntries = 0;
  for (uint32_t s = (uint32_t)stripeGroup_.size(); s-- > 0;)
    if (stripeGroup_[s] < nGroups) firstStripeOf[stripeGroup_[s]] = s;
  for (uint32_t g = 0; g < nGroups; ++g) {
    const Fb sg = sectionBytes(readSection(footer.table(footer.vecOffsetElem(groupsVec, g))), "stripe group");
    if (sg.n >= 8 && memcmp(sg.p + 4, "SGL1", 4) == 0)  // FlatBuffers file identifier of the stream-major layout (StripeGroup.h:66-76)
      throw NimbleUnsupported(NBL_ERR_UNSUPPORTED, "stream-major stripe group layout");
    const Fb::Table t = sg.root();
    Group& G = groups_[g];
    G.stripeCount = t.scalar<uint32_t>(0, 0);
    G.offsets = sg.scalars<uint32_t>(t.indirect(1));
    G.sizes = sg.scalars<uint32_t>(t.indirect(2));
    if (G.offsets.size() != G.sizes.size()) corrupt("stripe group offsets and sizes differ in length");
    // every entry of a genuine file is stored once: sections shared between groups cannot add up to more than the file
    tableEntries += G.offsets.size();
    if (tableEntries * 8 > file_.size()) corrupt("stripe group tables larger than the file");
    if (G.stripeCount == 0 || G.offsets.size() % G.stripeCount) corrupt("stripe group stream table is not stripe_count x stream_count");
    G.streamCount = (uint32_t)(G.offsets.size() / G.stripeCount);
    if (firstStripeOf[g] == UINT32_MAX) corrupt("No stripes found for stripe group.");
    G.firstStripe = firstStripeOf[g];
  }
  for (uint32_t s = 0; s < stripeGroup_.size(); ++s) {
    if (stripeGroup_[s] >= nGroups) corrupt("stripe refers to a missing stripe group");
    const Group& G = groups_[stripeGroup_[s]];
    if (s < G.firstStripe || s - G.firstStripe >= G.stripeCount) corrupt("stripe outside its stripe group");
  }
  if (ft.has(3)) {  // Footer.optional_sections
    const Fb::Table ot = footer.table(ft.indirect(3));
    const uint64_t names = ot.indirect(0);
    const auto offsets = footer.scalars<uint64_t>(ot.indirect(1));
    const auto sizes = footer.scalars<uint32_t>(ot.indirect(2));
    const auto comps = footer.scalars<uint8_t>(ot.indirect(3));
    const uint32_t n = footer.vecLen(names, 4);
    if (offsets.size() != n || sizes.size() != n) corrupt("optional sections vectors differ in length");
    for (uint32_t i = 0; i < n; ++i) {
      const std::string_view name = footer.str(footer.vecOffsetElem(names, i));
      if (i < comps.size() && comps[i] != 0) continue;  // compressed: not exposed
      if (offsets[i] > file_.size() || sizes[i] > file_.size() - offsets[i]) corrupt("optional section outside the file");
      optional_.emplace_back(std::string(name), file_.substr(offsets[i], sizes[i]));
    }
  }
}

uint32_t Tablet::streamCount(uint32_t stripe) const { return groups_.at(stripeGroup_.at(stripe)).streamCount; }

std::string_view Tablet::stream(uint32_t stripe, uint32_t streamId) const {
  const Group& G = groups_.at(stripeGroup_.at(stripe));
  if (streamId >= G.streamCount) return {};
  const size_t at = (size_t)(stripe - G.firstStripe) * G.streamCount + streamId;
  const uint32_t size = G.sizes[at];
  if (size == 0) return {};
  const uint64_t start = stripeOffsets_[stripe] + G.offsets[at];  // TabletReader.cpp:987-988
  if (start > file_.size() || size > file_.size() - start) corrupt("stream outside the file");
  return file_.substr(start, size);
}

std::vector<std::pair<uint32_t, std::string_view>> Tablet::streams(uint32_t stripe) const {
  std::vector<std::pair<uint32_t, std::string_view>> out;
  const uint32_t n = streamCount(stripe);
  for (uint32_t s = 0; s < n; ++s) {
    const std::string_view v = stream(stripe, s);
    if (!v.empty()) out.emplace_back(s, v);
  }
  return out;
}

std::string_view Tablet::optionalSection(std::string_view name) const {
  for (const auto& kv : optional_)
    if (kv.first == name) return kv.second;
  return {};
}

}  // namespace nimble_amd

// Host side of the C ABI: header-only walk of the cascade trees of a batch of
// chunks, output / scratch layout, level-ordered launch list.
//
// The walk mirrors what the reference's constructors do when
// EncodingFactory::create builds an Encoding tree
// (dwio/nimble/encodings/common/EncodingFactory.cpp:62-164 and the per-encoding
// constructors cited below) and what EncodingLayoutCapture::capture does without
// decoding payloads (dwio/nimble/encodings/common/EncodingLayout.cpp:171-405),
// but over a whole batch and without touching payload bytes.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <new>
#include <string>
#include <system_error>
#include <thread>
#include <tuple>
#include <vector>

#include "../../include/nimble_amd.h"
#include "nbl_desc.h"

extern "C" int nbl_launch_kernel(uint32_t kernel, uint32_t es, const nbl::KArgs* args, void* stream);
extern "C" int nbl_launch_copy_to_host(void* host_dst, const void* dev_src, uint64_t bytes, void* stream);

namespace {

using namespace nbl;

thread_local std::string g_last_error;

struct PlanError {
  nbl_status status;
  std::string msg;
};

[[noreturn]] void fail(nbl_status st, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  throw PlanError{st, buf};
}
#define NBL_CHECK(cond, ...)                         \
  do {                                               \
    if (!(cond)) fail(NBL_ERR_CORRUPT, __VA_ARGS__); \
  } while (0)

// EncodingType / DataType (dwio/nimble/common/Types.h:69-183)
enum : uint8_t {
  ET_TRIVIAL = 0, ET_RLE = 1, ET_DICTIONARY = 2, ET_FIXEDBITWIDTH = 3, ET_SENTINEL = 4,
  ET_NULLABLE = 5, ET_SPARSEBOOL = 6, ET_VARINT = 7, ET_DELTA = 8, ET_CONSTANT = 9,
  ET_MAINLYCONSTANT = 10, ET_BLOCKBITPACKING = 15, ET_FOR = 18, ET_MAX_KNOWN = 22
};
enum : uint8_t {
  DT_UNDEFINED = 0, DT_INT8 = 1, DT_UINT8 = 2, DT_INT16 = 3, DT_UINT16 = 4, DT_INT32 = 5,
  DT_UINT32 = 6, DT_INT64 = 7, DT_UINT64 = 8, DT_FLOAT = 9, DT_DOUBLE = 10, DT_BOOL = 11,
  DT_STRING = 12
};

constexpr uint64_t kAlign = 256;
constexpr uint64_t kArenaSlop = 64;
inline uint64_t align_up(uint64_t v, uint64_t a) { return (v + a - 1) / a * a; }
inline uint32_t div_up32(uint64_t a, uint64_t b) { return (uint32_t)((a + b - 1) / b); }

// One encoding of a cascade tree.
struct PNode {
  uint8_t enc = 0, dtype = 0;
  uint8_t es = 0;          // element bytes of the materialised form: numeric size, string 8, bool 0
  bool is_bool = false, is_string = false;
  uint32_t rows = 0;
  uint64_t off = 0, size = 0;  // span of the encoding in the arena
  int child[3] = {-1, -1, -1};
  int chunk = -1;
  int depth = 0;
  // leaf parameters
  uint64_t payload_off = 0;  // packed bits / raw values / bitmap / varint bytes / string blob
  uint64_t payload_bytes = 0;
  uint64_t value = 0;        // baseline, constant, common value (strings: view)
  uint8_t width = 0;
  uint8_t flag = 0;          // sparse value / first bool
  uint32_t seg_size = 0, seg_first = 0, seg_count = 0;  // BlockBitPacking blocks / FOR frames
  // string constants (Constant<string>, MainlyConstant<string> common value)
  uint64_t str_off = 0;      // arena offset of the string bytes
  uint32_t str_len = 0;
  // Dictionary whose gather its Nullable / MainlyConstant parent performs: `ref`
  // then names the uint32 indices and `alpha_ref` the alphabet element array
  bool fused_gather = false;
  ColRef alpha_ref{};
  // Trivial<string>: the longest string its lengths child can hold (from the header: baseline + all bits set) and the
  // descriptor of its lengths -> views step; Dictionary over such an alphabet, short strings, 4,097 .. 65,536 entries:
  // `alpha_table` names the compact table that step also writes (K_DICT_STR)
  uint64_t max_len = ~0ull;
  int views_dn = -1;
  bool alpha_compact = false;
  uint64_t alpha_table = 0;
  // MainlyConstant whose scatter its Nullable / MainlyConstant parent performs
  // together with its own (NF_NESTED_SELECT): `ref` then names the other values,
  // `sel_ref` the isCommon bitmap, and the rank partials stay this node's
  bool fused_select = false;
  ColRef sel_ref{};
  uint32_t sel_part_base = 0, sel_total_idx = 0;
  // RLE whose expansion its Nullable / MainlyConstant parent performs inside its
  // rank scatter (NF_RLE_VALUES): `ref` then names the run values, `ends_ref` the
  // run-end array in work (first_run[] behind it), sel_total_idx its totals slot
  bool fused_runs = false;
  ColRef ends_ref{};
  // SparseBool: descriptor of its positions -> bitmap step (-1: none), so that a
  // parent selecting by the bitmap can have that step count its rank partials
  int sparse_dn = -1;
  // result
  ColRef ref{};              // how a consumer reads this node
  int ready = 0;             // level after which `ref` is valid
};

// What of a descriptor / a chunk layout is an offset into a region whose base
// is only known once the batch is put together (per-thread plans are merged by
// adding the bases): everything else in them is position independent.
enum : uint8_t {
  DM_AUX_WORK = 1,   // aux_off is an offset into work (run ends)
  DM_AUX_OUT = 2,    // aux_off is an offset into out (null bitmap)
  DM_PARTS = 4,      // part_base
  DM_TOTALS = 8,     // total_idx
  DM_PARTS2 = 16,    // part_base2
  DM_TOTALS2 = 32,   // total_idx2
  DM_LOOK = 64,      // look_base
  DM_AUX3_WORK = 128,  // aux3 is an offset into work (compact string views)
};
enum : uint8_t { LM_NULLS = 1, LM_STRING = 2, LM_DICT_ALPHABET = 4, LM_DICT_INDICES = 8 };

constexpr size_t kDenseSparseMaxChunks = 512;
// Positions a SparseBool column of up to kWholeTileRows rows may list and still be
// built by one workgroup: two rounds of loads in a small batch, sixteen where
// thousands of other tiles cover the wait.
constexpr uint32_t kWholeMaxPositions = 4096, kWholeMaxPositionsBatch = 32768;
// Rank-scatter tiles of one level (16 KiB of output each) from which the one-wave
// kernel gets its own launch; a launch of one element size needs an eighth of that.
constexpr uint64_t kWaveLaunchMinTiles = 32768;
// Host threads that walk the chunk headers of one batch, and the chunks a thread
// must have for it to be started (a thread costs tens of microseconds, a chunk one to two).
constexpr uint32_t kMaxPlanThreads = 16, kChunksPerPlanThread = 128;
// Tiles (16 KiB of output each) of 2,049 .. 4,096-entry dictionaries of one level
// from which the LDS-alphabet kernel gets its launch.
constexpr uint64_t kLdsDictMinTiles = 4096;

// std::vector that leaves the elements resize() adds uninitialised (plain data
// only): the big tables of a plan are sized once and then filled, by several
// threads when the batch is large, and zeroing megabytes first is time on one.
template <class T>
struct NoInit : std::allocator<T> {
  template <class U> struct rebind { using other = NoInit<U>; };
  template <class U> void construct(U* p) noexcept { ::new (static_cast<void*>(p)) U; }
  template <class U, class... A> void construct(U* p, A&&... a) { ::new (static_cast<void*>(p)) U(std::forward<A>(a)...); }
};
template <class T>
using RawVec = std::vector<T, NoInit<T>>;

constexpr uint32_t kNoAux = 0xFFFFFFFFu;
struct Step {
  int level;
  uint32_t kernel;
  uint32_t es;
  uint32_t dnode;
  uint32_t n_tiles;
  int chunk;      // >= 0 when this step writes the final values of that chunk
  int owner;      // chunk the step belongs to
  uint32_t tile_begin;  // first tile of the node this step covers (steps can be split)
  uint64_t alg;         // algorithmic bytes credited to this step (set by finish())
  bool filler;          // a free step placed in a level to fill it: its tiles go after the level's own
  uint32_t aux_base;    // rank scatters (a Dictionary gathered inside one included): first slot of the node's scanned rank
                        // partials, from which the tile table's tile_aux entries count; kNoAux otherwise
  uint32_t order_key;   // place among the steps of its (level, kernel, element size) where finish() orders those: taken
                        // from the descriptor while that is in the cache (the descriptors of a wide batch are megabytes)
};

struct Launch {
  uint32_t kernel, es;  // KernelId, or K_LEVEL / K_LEVEL_LAST for a fused launch
  uint32_t tile_begin, n_tiles;
  int level;
  uint64_t alg_bytes;
  std::string name;     // as a profiler shows it, e.g. "k_materialize_fbw<8>", "k_level<true>"
};

// toString(DataType), dwio/nimble/common/Types.cpp
const char* data_type_name(uint32_t dt) {
  static const char* const names[] = {"Undefined", "Int8", "Uint8", "Int16", "Uint16", "Int32", "Uint32", "Int64", "Uint64", "Float", "Double", "Bool", "String"};
  return dt < sizeof names / sizeof names[0] ? names[dt] : "Unknown";
}

const char* kKernelNames[K_COUNT] = {
    "k_materialize", "k_bits_to_bytes", "k_reduce_popc", "k_rle_ends", "k_rle_expand", "k_rle_expand_bool", "k_sparse_bits",
    "k_select_scatter", "k_dict_gather", "k_varint_decode", "k_string_views", "k_copy_bytes",
    "k_select_bits", "k_materialize_fbw", "k_materialize_raw", "k_frame_unpack", "k_select_scatter_wave", "k_dict_lds", "k_dict_str"};

}  // namespace

struct nbl_plan {
  const uint8_t* base = nullptr;
  bool varint_rows = false;
  uint64_t span = 0;
  std::vector<PNode> nodes;               // the cascade tree of the chunk being planned
  uint32_t n_nodes = 0;                   // encodings of all chunks planned so far
  std::vector<nbl_chunk_layout> layouts;
  RawVec<NodeDesc> dnodes;
  RawVec<uint8_t> dn_mask;           // per dnode: DM_* (which fields are offsets to rebase)
  std::vector<uint8_t> layout_mask;       // per chunk: LM_* (which optional layout offsets are set)
  std::vector<uint32_t> chunk_copies;     // dnodes of the string copies of the chunk being planned
  size_t batch_chunks = 0;                // chunks of the whole batch (a per-thread plan holds a range of them)
  std::vector<uint64_t> chunk_alg;        // per chunk: encoded + decoded bytes
  RawVec<Step> steps;
  std::vector<Launch> launches;
  RawVec<TileRef> tiles;
  RawVec<uint32_t> tile_aux;         // parallel to tiles (KArgs::tile_aux)
  bool keep_dictionary = false;  // also emit alphabet + indices of dictionary chunks
  bool fused = true;          // one launch per cascade level (default) or one per (level, kernel, element size)
  bool wave_scatter = true;   // sparse rank scatters on the one-wave kernel (default) or every one on the four-wave tile
  bool lds_dictionary = true; // 8-byte dictionaries on the kernels that keep the alphabet in LDS (default: K_DICT_LDS, K_DICT_STR) or gathered from memory
  uint64_t lds_dict_min_tiles = kLdsDictMinTiles;  // tiles of such dictionaries in one level that make a launch of that kernel
  uint64_t wave_min_tiles = kWaveLaunchMinTiles;  // tiles of one level and element size that make a one-wave launch
  int cur_chunk = -1;         // chunk being planned
  uint64_t tiles_off = 0, aux_off = 0;
  uint64_t ctrl_off = 0, ctrl_bytes = 0;   // status word + per-node totals (zeroed by the upload)
  uint64_t parts_off = 0;
  uint32_t n_parts = 0, n_totals = 0, n_look = 0;
  uint64_t look_off = 0;
  mutable std::atomic<uint32_t> decodes{0};  // number of the last decode (look-back words carry it)
  uint64_t out_bytes = 0, work_bytes = 0, desc_bytes = 0;
  uint64_t cur_out = 0, cur_work = 0;      // allocation cursors
  uint64_t cur_str = 0;                    // string bytes of the chunk being planned
  uint64_t str_base = 0;                   // out offset of that chunk's string region
  uint64_t encoded_bytes = 0, decoded_bytes = 0, total_rows = 0;
  uint32_t max_depth = 0;

  // ---- parsing ----------------------------------------------------------
  uint8_t u8(uint64_t o) const { return base[o]; }
  uint32_t u32(uint64_t o) const { uint32_t v; memcpy(&v, base + o, 4); return v; }
  uint64_t le(uint64_t o, int n) const { uint64_t v = 0; memcpy(&v, base + o, (size_t)n); return v; }

  int parse(uint64_t off, uint64_t size, int chunk, int depth);
  void parse_value(PNode& n, uint64_t& pos, uint64_t end);
  // ---- layout / scheduling ----------------------------------------------
  uint64_t alloc_work(uint64_t bytes) { uint64_t o = cur_work; cur_work = align_up(cur_work + bytes, kAlign); return o; }
  uint64_t alloc_out(uint64_t bytes) { uint64_t o = cur_out; cur_out = align_up(cur_out + bytes, kAlign); return o; }
  uint32_t add_dnode(const NodeDesc& d, uint8_t mask = 0) {
    dnodes.push_back(d);
    dn_mask.push_back(mask);
    return (uint32_t)dnodes.size() - 1;
  }
  void add_step(int level, uint32_t kernel, uint32_t es, uint32_t dnode, uint32_t n_tiles) {
    if (n_tiles == 0) return;
    if (n_tiles > kTileMask) fail(NBL_ERR_UNSUPPORTED, "a node of %u tiles exceeds the tile table's range", n_tiles);
    const NodeDesc& d = dnodes[dnode];
    const bool ranked = kernel == K_SELECT_SCATTER || kernel == K_SELECT_SCATTER_WAVE ||
                        ((kernel == K_DICT_LDS || kernel == K_DICT_STR) && (d.flags & NF_DICT_VALUES));
    steps.push_back(Step{level, kernel, es, dnode, n_tiles, -1, cur_chunk, 0, 0, false, ranked ? d.part_base : kNoAux,
                         order_key_of(kernel, d, n_tiles)});
  }
  // Rank scatters are grouped by what their tiles execute (which values child is
  // resolved inside, how values and bitmap are read, how dense the selection
  // is); SparseBool bitmap steps by the positions a tile reads, longest first.
  static uint32_t order_key_of(uint32_t kernel, const NodeDesc& d, uint32_t n_tiles) {
    if (kernel == K_SPARSE_BITS) return ~(d.c0.rows / n_tiles);
    if (kernel == K_SELECT_SCATTER || kernel == K_SELECT_SCATTER_WAVE || kernel == K_DICT_LDS || kernel == K_DICT_STR) {
      const uint32_t density = d.rows ? (uint32_t)std::min<uint64_t>(7, (uint64_t)d.c1.rows * 8 / d.rows) : 0u;
      return (d.flags & (NF_NESTED_SELECT | NF_RLE_VALUES | NF_DICT_VALUES | NF_INVERT_SEL)) << 16 | (uint32_t)d.c1.kind << 12 |
             (uint32_t)(d.c1.width > 25) << 11 | (uint32_t)d.sel.kind << 4 | density;
    }
    return 0;
  }
  // A node whose positions depend on its data: the totals slot (total, flag
  // word, ticket counter, spare), `chain_tiles` look-back words for a chained
  // scan over that many tiles, and -- for the rank scatters, whose tiles read
  // their share from a table -- `rank_parts` scanned partials (+ the total).
  // Returns the DM_* bits of the fields it set (for add_dnode).
  uint8_t alloc_scan(NodeDesc& d, uint32_t n_tiles, uint32_t chain_tiles, bool rank_parts) {
    d.n_parts = n_tiles;
    if (rank_parts) { d.part_base = n_parts; n_parts += n_tiles + 1; }
    d.total_idx = n_totals; n_totals += 4;
    d.look_base = n_look; n_look += chain_tiles > 1 ? chain_tiles : 0;
    return (uint8_t)(DM_TOTALS | DM_LOOK | (rank_parts ? DM_PARTS : 0));
  }
  struct MergeBase { size_t dnode, step, chunk; uint64_t out, work; uint32_t parts, totals, look; };
  MergeBase reserve_behind(const nbl_plan& sub, size_t chunk_base);
  void copy_from(const nbl_plan& sub, const MergeBase& b);
  uint64_t string_copy(int level, uint64_t arena_off, uint64_t bytes);
  uint64_t column_to_out(const ColRef& src, uint32_t rows, uint32_t es, int level);
  void keep_dictionary_of(int dict_node, bool with_indices);
  void place(int idx, bool root, const ColRef* root_dst);
  void plan_chunk(uint32_t c, uint64_t off, uint64_t size);
  void spread_free_steps(int last_level);
  void finish();
};

namespace {

int elem_size(uint8_t dt) {
  switch (dt) {
    case DT_INT8: case DT_UINT8: return 1;
    case DT_INT16: case DT_UINT16: return 2;
    case DT_INT32: case DT_UINT32: case DT_FLOAT: return 4;
    case DT_INT64: case DT_UINT64: case DT_DOUBLE: return 8;
    default: return 0;
  }
}
bool is_u32_index_type(uint8_t dt) { return dt == DT_UINT32 || dt == DT_INT32; }

// Which materialise variant unpacks the leaf column `ref` of `es`-byte elements.
uint32_t leaf_kernel(const ColRef& ref, uint32_t es) {
  const uint32_t w = ref.width;
  const bool window = w >= 1 && (es == 8 ? w <= 63 : es == 4 ? w <= 31 : es == 2 ? w <= 15 : w <= 7);
  if (ref.kind == COL_RAW || ref.kind == COL_MAT_WORK || ref.kind == COL_MAT_OUT || (ref.kind == COL_FBW && w == 8 * es))
    return K_MATERIALIZE_RAW;
  if (ref.kind == COL_FBW && window) return K_MATERIALIZE_FBW;
  return K_MATERIALIZE;
}

}  // namespace

// Reads the constant / common value that trails an encoding (encoding::read<physicalType>,
// dwio/nimble/encodings/common/EncodingPrimitives.h:123-160).
void nbl_plan::parse_value(PNode& n, uint64_t& pos, uint64_t end) {
  if (n.is_string) {
    NBL_CHECK(pos + 4 <= end, "string value beyond end of encoding");
    const uint32_t len = u32(pos);
    NBL_CHECK(pos + 4 + (uint64_t)len <= end, "string value beyond end of encoding");
    n.str_off = pos + 4;
    n.str_len = len;
    pos += 4 + (uint64_t)len;
  } else if (n.is_bool) {
    NBL_CHECK(pos + 1 <= end, "value beyond end of encoding");
    n.value = u8(pos) != 0;
    pos += 1;
  } else {
    NBL_CHECK(pos + n.es <= end, "value beyond end of encoding");
    n.value = le(pos, n.es);
    pos += n.es;
  }
}

int nbl_plan::parse(uint64_t off, uint64_t size, int chunk, int depth) {
  NBL_CHECK(size >= 2, "encoding shorter than its prefix");
  NBL_CHECK(depth < 32, "cascade deeper than 32 levels");
  PNode n;
  n.off = off; n.size = size; n.chunk = chunk; n.depth = depth;
  n.enc = u8(off);
  n.dtype = u8(off + 1);
  uint64_t pos;  // first byte after the prefix
  if (varint_rows) {  // EncodingPrefix.h:77-92, Varint.h:83-102
    uint32_t v = 0; pos = off + 2; bool done = false;
    for (int k = 0; k < 5 && !done; ++k) {
      NBL_CHECK(pos < off + size, "row count varint beyond end of encoding");
      const uint8_t b = u8(pos++);
      v |= (uint32_t)(b & 127) << (7 * k);
      done = !(b & 128) || k == 4;
    }
    n.rows = v;
  } else {
    NBL_CHECK(size >= 6, "encoding shorter than its prefix");
    n.rows = u32(off + 2);
    pos = off + 6;
  }
  const uint64_t end = off + size;
  if (n.enc > ET_MAX_KNOWN || n.enc == ET_SENTINEL)
    fail(NBL_ERR_CORRUPT, "Trying to deserialize invalid EncodingType:%d -- garbage input?", (int)n.enc);
  NBL_CHECK(n.dtype >= DT_INT8 && n.dtype <= DT_STRING, "invalid DataType %d", (int)n.dtype);
  n.is_bool = n.dtype == DT_BOOL;
  n.is_string = n.dtype == DT_STRING;
  n.es = n.is_string ? 8 : (uint8_t)elem_size(n.dtype);
  max_depth = std::max<uint32_t>(max_depth, (uint32_t)depth + 1);

  const int self = (int)nodes.size();
  nodes.push_back(n);
  auto N = [&]() -> PNode& { return nodes[self]; };
  // what follows a first child (the next child's prefix) is far from this
  // header: start fetching it before the first child's subtree is walked
  auto ahead = [&](uint64_t first_child, uint64_t first_size) {
    if (first_size < end - first_child) __builtin_prefetch(base + first_child + first_size);
  };
  auto child = [&](int slot, uint64_t coff, uint64_t csize) {
    NBL_CHECK(coff <= end && csize <= end - coff, "child encoding beyond end of its parent");
    const int c = parse(coff, csize, chunk, depth + 1);
    N().child[slot] = c;
    return c;
  };

  switch (n.enc) {
    case ET_TRIVIAL: {  // TrivialEncoding.h:252-283, TrivialEncoding.cpp:33-44,101-136,291-319
      NBL_CHECK(pos + 1 <= end, "Trivial header beyond end");
      if (u8(pos) != 0) fail(NBL_ERR_UNSUPPORTED, "compressed Trivial payload (CompressionType %d)", (int)u8(pos));
      pos += 1;
      if (n.is_string) {
        NBL_CHECK(pos + 4 <= end, "Trivial<string> header beyond end");
        const uint32_t lsize = u32(pos);
        pos += 4;
        const int c = child(0, pos, lsize);
        NBL_CHECK(is_u32_index_type(nodes[c].dtype), "Trivial<string> lengths must be uint32");
        NBL_CHECK(nodes[c].rows >= N().rows, "Trivial<string> has fewer lengths than rows");
        pos += lsize;
        N().payload_off = pos;
        N().payload_bytes = end - pos;
      } else if (n.is_bool) {
        N().payload_off = pos;
        N().payload_bytes = ((uint64_t)n.rows + 7) / 8;
        NBL_CHECK(pos + N().payload_bytes + 7 == end, "Unexpected trivial encoding end");
      } else {
        N().payload_off = pos;
        N().payload_bytes = (uint64_t)n.rows * n.es;
        NBL_CHECK(pos + N().payload_bytes == end, "Unexpected trivial encoding end");
      }
      break;
    }
    case ET_FIXEDBITWIDTH: {  // FixedBitWidthEncoding.h:149-175
      NBL_CHECK(!n.is_bool && !n.is_string, "Trying to deserialize a non-numeric stream for a numeric data type %s.", data_type_name(n.dtype));
      NBL_CHECK(pos + 2 + n.es <= end, "FixedBitWidth header beyond end");
      if (u8(pos) != 0) fail(NBL_ERR_UNSUPPORTED, "compressed FixedBitWidth payload (CompressionType %d)", (int)u8(pos));
      N().value = le(pos + 1, n.es);
      N().width = u8(pos + 1 + n.es);
      pos += 2 + n.es;
      NBL_CHECK(N().width <= 64, "bit width must lie in [0, 64], got: %d", (int)N().width);
      NBL_CHECK(n.es == 8 || N().width <= 32, "bit width must lie in [1, 32], got: %d", (int)N().width);
      N().payload_off = pos;
      N().payload_bytes = ((uint64_t)n.rows * N().width + 7) / 8;
      // FixedBitArray::bufferSize: packed bytes + 7 bytes the word loads may touch
      NBL_CHECK(n.rows == 0 || pos + N().payload_bytes + 7 <= end, "FixedBitWidth payload truncated");
      break;
    }
    case ET_VARINT: {  // VarintEncoding.h:94-111
      NBL_CHECK(!n.is_bool && !n.is_string && (n.es == 4 || n.es == 8), "Trying to deserialize a varint stream for an incompatible data type %s.", data_type_name(n.dtype));
      NBL_CHECK(pos + n.es <= end, "Varint header beyond end");
      N().value = le(pos, n.es);
      pos += n.es;
      N().payload_off = pos;
      N().payload_bytes = end - pos;
      NBL_CHECK(N().payload_bytes >= n.rows, "Varint payload truncated");
      break;
    }
    case ET_CONSTANT: {  // ConstantEncoding.h:292-302, ConstantEncoding.cpp:21-33
      parse_value(N(), pos, end);
      NBL_CHECK(pos == end, "Unexpected constant encoding end");
      break;
    }
    case ET_RLE: {  // RLEEncoding.h:113-124,816-837, RLEEncoding.cpp:20-37
      NBL_CHECK(pos + 4 <= end, "RLE header beyond end");
      const uint32_t lsize = u32(pos);
      pos += 4;
      ahead(pos, lsize);
      const int c0 = child(0, pos, lsize);
      NBL_CHECK(is_u32_index_type(nodes[c0].dtype), "RLE run lengths must be uint32");
      pos += lsize;
      if (n.is_bool) {
        NBL_CHECK(pos + 1 == end, "Unexpected run length encoding end");
        N().flag = u8(pos) != 0;
      } else {
        NBL_CHECK(pos < end, "RLE run values missing");
        const int c1 = child(1, pos, end - pos);
        NBL_CHECK(nodes[c1].dtype == n.dtype, "RLE run values have a different data type");
        NBL_CHECK(nodes[c1].rows >= nodes[c0].rows, "RLE has fewer run values than run lengths");
      }
      break;
    }
    case ET_DICTIONARY: {  // DictionaryEncoding.h:193-214
      NBL_CHECK(!n.is_bool, "Trying to deserialize a non-bool stream for the bool data type %s.", data_type_name(n.dtype));
      NBL_CHECK(pos + 4 <= end, "Dictionary header beyond end");
      const uint32_t asize = u32(pos);
      pos += 4;
      ahead(pos, asize);
      const int c0 = child(0, pos, asize);
      NBL_CHECK(nodes[c0].dtype == n.dtype, "Dictionary alphabet has a different data type");
      pos += asize;
      NBL_CHECK(pos < end, "Dictionary indices missing");
      const int c1 = child(1, pos, end - pos);
      NBL_CHECK(is_u32_index_type(nodes[c1].dtype), "Dictionary indices must be uint32");
      NBL_CHECK(nodes[c1].rows >= N().rows, "Dictionary has fewer indices than rows");
      break;
    }
    case ET_NULLABLE: {  // NullableEncoding.h:151-171
      NBL_CHECK(pos + 4 <= end, "Nullable header beyond end");
      const uint32_t dsize = u32(pos);
      pos += 4;
      ahead(pos, dsize);
      const int c0 = child(0, pos, dsize);
      NBL_CHECK(nodes[c0].dtype == n.dtype, "Nullable data has a different data type");
      pos += dsize;
      NBL_CHECK(pos < end, "Nullable nulls missing");
      const int c1 = child(1, pos, end - pos);
      NBL_CHECK(nodes[c1].dtype == DT_BOOL, "Nullable nulls must be bool");
      NBL_CHECK(nodes[c1].rows == N().rows, "Nulls count mismatch.");
      NBL_CHECK(nodes[c0].rows <= N().rows, "Nullable has more non-null values than rows");
      break;
    }
    case ET_SPARSEBOOL: {  // SparseBoolEncoding.cpp:70-84, EncodingFactory.cpp:90-97
      NBL_CHECK(n.is_bool, "Trying to deserialize a SparseBoolEncoding with a non-bool data type.");
      NBL_CHECK(pos + 1 < end, "SparseBool header beyond end");
      N().flag = u8(pos) != 0;
      pos += 1;
      const int c0 = child(0, pos, end - pos);
      NBL_CHECK(is_u32_index_type(nodes[c0].dtype), "SparseBool indices must be uint32");
      NBL_CHECK(nodes[c0].rows >= 1, "SparseBool indices lack the row-count sentinel");
      break;
    }
    case ET_MAINLYCONSTANT: {  // MainlyConstantEncoding.h:842-860, MainlyConstantEncoding.cpp:21-43
      NBL_CHECK(!n.is_bool, "Trying to deserialize a non-bool stream for the bool data type %s.", data_type_name(n.dtype));
      NBL_CHECK(pos + 4 <= end, "MainlyConstant header beyond end");
      const uint32_t isize = u32(pos);
      pos += 4;
      ahead(pos, isize);
      const int c0 = child(0, pos, isize);
      NBL_CHECK(nodes[c0].dtype == DT_BOOL && nodes[c0].rows == N().rows, "MainlyConstant isCommon mismatch");
      pos += isize;
      NBL_CHECK(pos + 4 <= end, "MainlyConstant header beyond end");
      const uint32_t osize = u32(pos);
      pos += 4;
      const int c1 = child(1, pos, osize);
      NBL_CHECK(nodes[c1].dtype == n.dtype, "MainlyConstant other values have a different data type");
      NBL_CHECK(nodes[c1].rows <= N().rows, "MainlyConstant has more other values than rows");
      pos += osize;
      parse_value(N(), pos, end);
      NBL_CHECK(pos == end, "Unexpected mainly constant encoding end");
      break;
    }
    case ET_BLOCKBITPACKING:  // BlockBitPackingEncoding.h:311-348,660-696
    case ET_FOR: {            // ForEncoding.h:174-196,316-383
      const bool frames = n.enc == ET_FOR;
      const char* name = frames ? "FOR" : "BlockBitPacking";
      if (frames) NBL_CHECK(n.dtype >= DT_INT8 && n.dtype <= DT_UINT64, "Trying to deserialize an integer stream for an incompatible data type %s.", data_type_name(n.dtype));
      else NBL_CHECK(!n.is_bool && !n.is_string, "Trying to deserialize a non-numeric stream for a numeric data type %s.", data_type_name(n.dtype));
      auto varint = [&]() {  // varint::readVarint32, common/Varint.h:104-125
        uint32_t v = 0;
        for (int k = 0; k < 5; ++k) {
          NBL_CHECK(pos < end, "%s header beyond end", name);
          const uint8_t b = u8(pos++);
          v |= (uint32_t)(b & 127) << (7 * k);
          if (!(b & 128)) break;
        }
        return v;
      };
      auto meta = [&](int slot, uint32_t es, const char* what) {  // varint size + nested encoding of one value per segment
        const uint32_t csize = varint();
        const int c = child(slot, pos, csize);
        NBL_CHECK(!nodes[c].is_bool && !nodes[c].is_string && nodes[c].es == es, "%s %s have the wrong element size", name, what);
        NBL_CHECK(nodes[c].rows >= N().seg_count, "%s has fewer %s than blocks", name, what);
        pos += csize;
      };
      NBL_CHECK(pos + 1 <= end, "%s header beyond end", name);
      if (u8(pos) != 0) fail(NBL_ERR_UNSUPPORTED, "compressed %s payload (CompressionType %d)", name, (int)u8(pos));
      pos += 1;
      N().seg_size = varint();
      N().seg_count = varint();
      if (frames) {
        N().seg_first = varint();
      } else {  // both fields are read into uint16_t; block size is capped by kBlockBitPackingBlockSize
        N().seg_size &= 0xFFFFu;
        N().seg_count &= 0xFFFFu;
        NBL_CHECK(N().seg_size <= 1024, "BlockBitPacking block size exceeds 1024");
      }
      NBL_CHECK(N().seg_size > 0 && N().seg_count > 0 && n.rows > 0, "%s needs rows, a block size and blocks", name);
      if (frames) { meta(1, 1, "bit widths"); meta(0, n.es, "references"); meta(2, 8, "bit offsets"); }
      else { meta(0, n.es, "baselines"); meta(1, 1, "bit widths"); meta(2, 4, "block offsets"); N().seg_first = varint(); }
      const uint32_t first = N().seg_first, size = N().seg_size;
      NBL_CHECK(first > 0 && first <= size && first <= n.rows, "%s first block rows out of range", name);
      const uint64_t expect = n.rows <= first ? 1 : 1 + ((uint64_t)(n.rows - first) + size - 1) / size;
      NBL_CHECK(N().seg_count == expect, "%s frame count mismatch.", name);
      if (frames) {
        const uint32_t packed = varint();
        NBL_CHECK(packed <= end - pos, "FOR packed data beyond end");
        N().payload_off = pos;
        N().payload_bytes = packed;
      } else {
        N().payload_off = pos;
        N().payload_bytes = end - pos;
      }
      break;
    }
    default:
      // Delta, Prefix, ALP, PFOR, ... : valid Nimble, decoded by the CPU factory.
      fail(NBL_ERR_UNSUPPORTED, "EncodingType %d is not decoded on the GPU", (int)n.enc);
  }
  return self;
}

// Registers a copy of `bytes` string bytes from the arena into the chunk's
// string region; returns the offset of the copy relative to that region.
uint64_t nbl_plan::string_copy(int level, uint64_t arena_off, uint64_t bytes) {
  const uint64_t rel = cur_str;
  cur_str = align_up(cur_str + bytes, 16);
  if (bytes) {
    NodeDesc d{};
    d.self.kind = COL_MAT_OUT;
    d.self.off = rel;        // plan_chunk adds the region's offset once it is known
    d.aux = bytes;
    d.aux_off = arena_off;
    const uint32_t dn = add_dnode(d);
    chunk_copies.push_back(dn);
    add_step(level, K_COPY_BYTES, 1, dn, div_up32(bytes, kTileBytes));
    decoded_bytes += bytes;
  }
  return rel;
}

// Schedules a copy of the column `src` (any kind) into a fresh region of `out`;
// returns the region's offset.
uint64_t nbl_plan::column_to_out(const ColRef& src, uint32_t rows, uint32_t es, int level) {
  const uint64_t off = alloc_out((uint64_t)rows * es);
  if (rows == 0) return off;
  NodeDesc d{};
  d.self.kind = COL_MAT_OUT; d.self.off = off; d.self.rows = rows; d.self.es = (uint8_t)es;
  d.c0 = src; d.c0.rows = rows; d.rows = rows;
  const uint32_t dn = add_dnode(d);
  add_step(level, leaf_kernel(src, es), es, dn, div_up32(rows, kTileBytes / es));
  decoded_bytes += (uint64_t)rows * es;
  return off;
}

// NBL_PLAN_KEEP_DICTIONARY: alphabet (and raw indices) of the placed Dictionary
// node `dict_node` of the current chunk go to `out` as well.
void nbl_plan::keep_dictionary_of(int dict_node, bool with_indices) {
  const PNode& dn = nodes[(size_t)dict_node];
  const PNode& alpha = nodes[(size_t)dn.child[0]];
  const PNode& ind = nodes[(size_t)dn.child[1]];
  nbl_chunk_layout& L = layouts[(size_t)dn.chunk];
  L.has_dictionary = 1;
  L.dict_size = alpha.rows;
  L.dict_alphabet_offset = column_to_out(alpha.ref, alpha.rows, dn.es, alpha.ready);
  layout_mask[(size_t)dn.chunk] |= LM_DICT_ALPHABET | (with_indices ? LM_DICT_INDICES : 0);
  if (with_indices) {
    L.dict_indices_rows = dn.rows;
    L.dict_indices_offset = column_to_out(ind.ref, dn.rows, 4, ind.ready);
  }
}

// Decides how node `idx` is produced and schedules its kernels.  `root_dst`
// (roots only) is where the values go in `out`.
void nbl_plan::place(int idx, bool root, const ColRef* root_dst) {
  PNode& nn = nodes[idx];  // `nodes` no longer grows once parsing is done
  const uint32_t rows = nn.rows;
  // A Dictionary of 4- or 8-byte rows (numbers, string views) directly under a
  // Nullable / MainlyConstant is not expanded into a column of its own: the
  // parent's rank scatter gathers from the alphabet itself, which saves writing
  // and re-reading every non-null row.
  if ((nn.enc == ET_NULLABLE || nn.enc == ET_MAINLYCONSTANT) && !nn.is_bool && nn.es >= 4 && !nn.fused_select) {
    PNode& vals = nodes[(size_t)nn.child[nn.enc == ET_NULLABLE ? 0 : 1]];
    if (vals.enc == ET_DICTIONARY && vals.rows > 0 && nodes[(size_t)vals.child[0]].rows > 0) vals.fused_gather = true;
    // A MainlyConstant directly underneath (Nullable{MainlyConstant{..}}: sparse
    // feature columns) is resolved by this node's rank scatter as well, so the
    // inner column is never written and read back.  Its other values must be a
    // column the scatter can read by index (anything but a Dictionary that would
    // want its own fused gather).
    if (vals.enc == ET_MAINLYCONSTANT && vals.rows > 0 && nodes[(size_t)vals.child[1]].enc != ET_DICTIONARY &&
        !(root && keep_dictionary))
      vals.fused_select = true;
    // So is an RLE directly underneath (Nullable{RLE{..}}): the parent's scatter
    // expands the runs; run ends and first_run[] are still prepared by the RLE's
    // own steps.
    if (vals.enc == ET_RLE && vals.rows > 0 && nodes[(size_t)vals.child[0]].rows > 0) vals.fused_runs = true;
  }
  // children first
  for (int s = 0; s < 3; ++s)
    if (nn.child[s] >= 0) place(nn.child[s], false, nullptr);
  auto C = [&](int s) -> PNode& { return nodes[nn.child[s]]; };
  int lvl = 0;
  for (int s = 0; s < 3; ++s)
    if (nn.child[s] >= 0) lvl = std::max(lvl, C(s).ready);

  auto mat_ref = [&](uint32_t es) {  // destination of a materialised numeric / view column
    ColRef r{};
    if (root && !nn.is_bool) { r = *root_dst; }
    else { r.kind = COL_MAT_WORK; r.off = alloc_work((uint64_t)rows * es); }
    r.rows = rows; r.es = (uint8_t)es;
    return r;
  };
  auto bits_ref = [&]() {
    ColRef r{};
    r.kind = COL_BITS_WORK; r.off = alloc_work(((uint64_t)rows + 63) / 64 * 8); r.rows = rows; r.es = 0;
    return r;
  };
  const uint32_t es = nn.es;
  const uint32_t tile_rows = es ? kTileBytes / es : kBitTileRows;

  switch (nn.enc) {
    case ET_FIXEDBITWIDTH: {
      ColRef r{};
      if (nn.width == 0) { r.kind = COL_CONST; r.base = nn.value; }
      else { r.kind = COL_FBW; r.off = nn.payload_off; r.base = nn.value; r.width = nn.width; }
      r.rows = rows; r.es = (uint8_t)es;
      nn.ref = r; nn.ready = 0;
      break;
    }
    case ET_TRIVIAL: {
      if (nn.is_string) {
        // views = running sum of the lengths; blob copied to the string region
        const uint64_t rel = string_copy(0, nn.payload_off, nn.payload_bytes);
        NodeDesc d{};
        d.self = mat_ref(8);
        d.c0 = C(0).ref; d.c0.rows = std::min(d.c0.rows, C(0).rows);
        d.rows = rows; d.aux = rel; d.aux2 = nn.payload_bytes;
        const uint32_t nt = div_up32(rows, kScanTile);
        const uint8_t sm = alloc_scan(d, nt, nt, false);  // the offsets: a chained scan over the view tiles, one step
        const uint32_t dn = add_dnode(d, sm);
        add_step(lvl, K_STRING_VIEWS, 8, dn, nt);
        nn.ref = d.self; nn.ready = lvl + 1;
        nn.views_dn = (int)dn;
        if (d.c0.kind == COL_CONST) nn.max_len = (uint32_t)d.c0.base;
        else if (d.c0.kind == COL_FBW && d.c0.width < 32) nn.max_len = (uint64_t)(uint32_t)d.c0.base + ((1ull << d.c0.width) - 1);
      } else if (nn.is_bool) {
        ColRef r{}; r.kind = COL_BITS_RAW; r.off = nn.payload_off; r.rows = rows;
        nn.ref = r; nn.ready = 0;
      } else {
        ColRef r{}; r.kind = COL_RAW; r.off = nn.payload_off; r.rows = rows; r.es = (uint8_t)es;
        nn.ref = r; nn.ready = 0;
      }
      break;
    }
    case ET_CONSTANT: {
      ColRef r{}; r.kind = COL_CONST; r.rows = rows; r.es = nn.is_bool ? 0 : (uint8_t)es;
      if (nn.is_string) {
        const uint64_t rel = string_copy(0, nn.str_off, nn.str_len);
        r.base = ((uint64_t)nn.str_len << 32) | (uint32_t)rel;
      } else {
        r.base = nn.value;
      }
      nn.ref = r; nn.ready = 0;
      break;
    }
    case ET_VARINT: {
      NodeDesc d{};
      d.self = mat_ref(es);
      d.c0.kind = COL_RAW; d.c0.off = nn.payload_off; d.c0.base = nn.value; d.c0.rows = rows; d.c0.es = (uint8_t)es;
      d.rows = rows; d.aux2 = nn.payload_bytes;
      const uint32_t nt = div_up32(nn.payload_bytes, kVarintTile);
      const uint8_t sm = alloc_scan(d, nt, nt, false);  // the ordinals: a chained scan over the decode tiles, one step
      const uint32_t dn = add_dnode(d, sm);
      add_step(lvl, K_VARINT_DECODE, es, dn, nt);
      nn.ref = d.self; nn.ready = lvl + 1;
      break;
    }
    case ET_RLE: {
      if (rows == 0) {
        // Nothing to expand whatever the run children hold (the reference reads a
        // run only when a row asks for one, RLEEncoding.h:289-301): an empty column,
        // no kernels.
        ColRef r{};
        if (nn.is_bool) { r.kind = COL_BITS_WORK; r.off = alloc_work(0); }
        else if (root) { r = *root_dst; }
        else { r.kind = COL_MAT_WORK; r.off = alloc_work(0); }
        r.rows = 0; r.es = nn.is_bool ? 0 : (uint8_t)es;
        nn.ref = r; nn.ready = lvl;
        break;
      }
      NodeDesc d{};
      d.c0 = C(0).ref;
      d.rows = rows;
      const uint32_t R = C(0).rows;
      d.c0.rows = R;
      const uint32_t nt = div_up32(R, kScanTile);
      const uint8_t sm = alloc_scan(d, nt, nt, false);  // the run ends: a chained scan over the tiles of runs, one step
      // run ends (R entries, 16-byte rounded) followed by first_run[] (one entry per output tile)
      const uint32_t out_tile_rows = nn.is_bool ? kBitTileRows : tile_rows;
      d.aux2 = out_tile_rows;
      d.aux_off = alloc_work((((uint64_t)R + 3) & ~3ull) * 4 + (uint64_t)div_up32(rows, out_tile_rows) * 4);
      if (nn.is_bool) {
        if (root) {  // the expansion writes the chunk's bytes itself: no bitmap, no separate bit -> byte step
          d.self = *root_dst; d.self.rows = rows; d.self.es = 1;
          d.flags |= NF_BOOL_BYTES;
        } else {
          d.self = bits_ref();
        }
        if (nn.flag) d.flags |= NF_FIRST_TRUE;
      } else {
        if (!nn.fused_runs) d.self = mat_ref(es);
        d.c1 = C(1).ref;
      }
      const uint32_t dn = add_dnode(d, sm | DM_AUX_WORK);
      int at = lvl;
      add_step(at++, K_RLE_ENDS, 4, dn, nt);
      if (nn.fused_runs) {
        // the parent expands: hand it the run ends (a work array, rebased with the
        // other intermediates), the run values and the totals slot
        nn.ends_ref.kind = COL_MAT_WORK; nn.ends_ref.off = d.aux_off; nn.ends_ref.rows = rows; nn.ends_ref.es = 4;
        nn.sel_total_idx = d.total_idx;
        nn.ref = d.c1; nn.ref.rows = R;
        nn.ready = at;
        break;
      }
      if (nn.is_bool) add_step(at, K_RLE_EXPAND_BOOL, 0, dn, div_up32(rows, kBitTileRows));
      else add_step(at, K_RLE_EXPAND, es, dn, div_up32(rows, tile_rows));
      nn.ref = d.self; nn.ready = at + 1;
      break;
    }
    case ET_SPARSEBOOL: {
      NodeDesc d{};
      d.self = bits_ref();
      d.c0 = C(0).ref;
      d.rows = rows;
      if (nn.flag) d.flags |= NF_SPARSE_VALUE;
      // more than 2048 positions per 65536 rows: a large tile would need several
      // rounds of position loads, so the column is cut into four times the tiles --
      // in batches of few chunks, whose bitmap tiles do not fill the chip's
      // workgroup slots and wait for the slowest tile.  A batch of thousands of
      // columns is bound by the fixed cost per tile instead (measured: +16%
      // on the bitmap level of a 5,000-column stripe with the small tiles).
      const bool dense = (uint64_t)C(0).rows * 32 > rows && batch_chunks <= kDenseSparseMaxChunks;
      // A column of up to 131072 rows is one tile when its positions take few
      // rounds of loads (16 rounds of 2048 in a batch that fills the chip anyway,
      // 2 otherwise): no search for the tile's range of positions, and a parent
      // selecting by the bitmap gets its rank partials already scanned.
      const bool whole = rows > 0 && rows <= kWholeTileRows &&
                         C(0).rows <= (batch_chunks > kDenseSparseMaxChunks ? kWholeMaxPositionsBatch : kWholeMaxPositions);
      if (whole) d.flags |= NF_WHOLE_NODE;
      else if (dense) d.flags |= NF_SPARSE_DENSE;
      const uint32_t dn = add_dnode(d);
      add_step(lvl, K_SPARSE_BITS, 0, dn, whole ? 1u : div_up32(rows, dense ? kBitTileRows : kSparseTileRows));
      if (rows > 0) nn.sparse_dn = (int)dn;
      nn.ref = d.self; nn.ready = lvl + 1;
      break;
    }
    case ET_DICTIONARY: {
      NodeDesc d{};
      d.c0 = C(1).ref;  // indices
      d.c1 = C(0).ref;  // alphabet
      {
        // the gather reads the alphabet as a plain element array: anything else
        // (bit-packed, constant) is unpacked into work first -- alphabets are small
        const ColRef& ar = d.c1;
        const bool plain = ar.kind == COL_RAW || ar.kind == COL_MAT_WORK || ar.kind == COL_MAT_OUT ||
                           (ar.kind == COL_FBW && ar.width == 8 * es);
        if (!plain && C(0).rows > 0) {
          NodeDesc m{};
          m.self.kind = COL_MAT_WORK; m.self.off = alloc_work((uint64_t)C(0).rows * es);
          m.self.rows = C(0).rows; m.self.es = (uint8_t)es;
          m.c0 = ar; m.rows = C(0).rows;
          const uint32_t mn = add_dnode(m);
          add_step(C(0).ready, leaf_kernel(ar, es), es, mn, div_up32(C(0).rows, tile_rows));
          d.c1 = m.self;
          lvl = std::max(lvl, C(0).ready + 1);
        }
      }
      // A string alphabet too large for the 8-byte views to sit in LDS, made of
      // short strings: its lengths -> views step also writes the views in compact
      // form, which a K_DICT_STR gather keeps in LDS instead.
      if (lds_dictionary && es == 8 && C(0).is_string && C(0).enc == ET_TRIVIAL && C(0).views_dn >= 0 &&
          C(0).max_len <= kDictStrMaxLen && C(0).rows > kTileBytes / 8 && C(0).rows <= kDictStrEntries &&
          d.c0.kind == COL_FBW && d.c0.width >= 1 && d.c0.width <= 16) {  // the kernel takes bit-packed indices of up to 16 bits
        const size_t vn = (size_t)C(0).views_dn;
        if (!(dnodes[vn].flags & NF_COMPACT_VIEWS)) {
          const uint64_t table = alloc_work(dict_str_table_bytes(C(0).rows));
          dnodes[vn].flags |= NF_COMPACT_VIEWS;
          dnodes[vn].aux_off = table;
          dn_mask[vn] |= DM_AUX_WORK;
        }
        nn.alpha_compact = true;
        nn.alpha_table = dnodes[vn].aux_off;
      }
      if (nn.fused_gather) {
        nn.ref = d.c0; nn.ref.rows = rows;
        nn.alpha_ref = d.c1; nn.alpha_ref.rows = C(0).rows;
        nn.ready = lvl;
        break;
      }
      d.self = mat_ref(es);
      d.rows = rows;
      if (nn.alpha_compact) d.aux3 = nn.alpha_table;
      const uint32_t dn = add_dnode(d, nn.alpha_compact ? DM_AUX3_WORK : 0);
      // 8-byte rows over an alphabet that the fused launch cannot stage (more than
      // kTileBytes of entries) but kDictLdsEntries hold: the LDS-alphabet kernel,
      // one workgroup per kDictLdsTiles tiles
      if (nn.alpha_compact)
        add_step(lvl, K_DICT_STR, es, dn, div_up32(div_up32(rows, tile_rows), kDictStrTiles));
      else if (lds_dictionary && es == 8 && C(0).rows > kTileBytes / 8 && C(0).rows <= kDictLdsEntries)
        add_step(lvl, K_DICT_LDS, es, dn, div_up32(div_up32(rows, tile_rows), kDictLdsTiles));
      else
        add_step(lvl, K_DICT_GATHER, es, dn, div_up32(rows, tile_rows));
      nn.ref = d.self; nn.ready = lvl + 1;
      break;
    }
    case ET_NULLABLE:
    case ET_MAINLYCONSTANT: {
      const bool nullable = nn.enc == ET_NULLABLE;
      NodeDesc d{};
      d.rows = rows;
      d.sel = nullable ? C(1).ref : C(0).ref;
      d.sel.rows = rows;
      d.c1 = nullable ? C(0).ref : C(1).ref;
      if (const PNode& vals = nullable ? C(0) : C(1); vals.fused_gather) {
        d.c0 = vals.alpha_ref;
        d.c1.rows = vals.rows;
        d.flags |= NF_DICT_VALUES;
        if (vals.alpha_compact) d.aux3 = vals.alpha_table;
      }
      if (!nullable) {
        d.flags |= NF_INVERT_SEL;
        if (nn.is_string) {
          const uint64_t rel = string_copy(0, nn.str_off, nn.str_len);
          d.aux = ((uint64_t)nn.str_len << 32) | (uint32_t)rel;
        } else {
          d.aux = nn.value;
        }
      }
      if (const PNode& vals = nullable ? C(0) : C(1); vals.fused_runs) {
        d.c0 = vals.ends_ref;         // run ends (+ first_run[]); its rows = this node's ranks
        d.c1 = vals.ref;              // run values, one per run
        d.total_idx2 = vals.sel_total_idx;
        d.flags |= NF_RLE_VALUES;
      }
      if (const PNode& vals = nullable ? C(0) : C(1); vals.fused_select) {
        d.c0 = vals.sel_ref;          // the inner isCommon bitmap; its rows = this node's ranks
        d.c1 = vals.ref;              // the inner other values
        d.aux3 = vals.value;          // the inner common value (strings: its view)
        d.part_base2 = vals.sel_part_base;
        d.total_idx2 = vals.sel_total_idx;
        d.flags |= NF_NESTED_SELECT;
      }
      if (nullable && root) {
        d.flags |= NF_WRITE_NULLS;
        d.aux_off = layouts[(size_t)nn.chunk].nulls_offset;
      }
      const uint32_t trows = nn.is_bool ? kBitTileRows : select_tile_rows(es);
      d.aux2 = trows;
      const uint32_t nt = div_up32(rows, trows);
      // The ranks need the selection bitmap only, and they come scanned out of one
      // step: a SparseBool selection writes them while it builds the bitmap (its
      // sorted positions give every tile the count in front of it), any other
      // bitmap is counted by a chained scan as soon as that child is ready,
      // underneath whatever the values child still has to do.
      const PNode& selc = nullable ? C(1) : C(0);
      const bool counted = selc.enc == ET_SPARSEBOOL && selc.sparse_dn >= 0 && selc.rows == rows;
      const bool whole = !counted && rows > 0 && rows <= kWholeTileRows;  // one workgroup counts the node alone
      const uint32_t popc_tiles = whole ? 1u : div_up32(rows, kPopcTileRows);
      uint8_t sm = alloc_scan(d, nt, counted ? 0u : popc_tiles, true);
      if (d.flags & NF_WRITE_NULLS) sm |= DM_AUX_OUT;
      if (d.flags & NF_NESTED_SELECT) sm |= DM_PARTS2 | DM_TOTALS2;
      if (d.flags & NF_RLE_VALUES) sm |= DM_TOTALS2;
      const bool str_dict = (d.flags & NF_DICT_VALUES) && es == 8 && (nullable ? C(0) : C(1)).alpha_compact;
      if (str_dict) sm |= DM_AUX3_WORK;
      if (counted) {
        dn_mask[(size_t)selc.sparse_dn] |= DM_PARTS | DM_TOTALS;
        NodeDesc& sd = dnodes[(size_t)selc.sparse_dn];
        sd.flags |= NF_EMIT_PARTS | (nullable ? 0u : (uint32_t)NF_INVERT_SEL);
        sd.part_base = d.part_base;
        sd.n_parts = d.n_parts;
        sd.total_idx = d.total_idx;
        sd.aux2 = trows;
      } else if (whole) {
        d.flags |= NF_WHOLE_NODE;
      }
      const int sel_lvl = selc.ready;
      const int fin = std::max(lvl, counted ? sel_lvl : sel_lvl + 1);
      if (nn.fused_select) {
        // the parent scatters for this node: it only counts its ranks
        const uint32_t dn = add_dnode(d, sm);
        if (!counted) add_step(sel_lvl, K_REDUCE_POPC, 0, dn, popc_tiles);
        nn.sel_ref = d.sel;
        nn.sel_part_base = d.part_base;
        nn.sel_total_idx = d.total_idx;
        nn.value = d.aux;
        nn.ref = d.c1;
        nn.ready = fin;
        break;
      }
      d.self = nn.is_bool ? bits_ref() : mat_ref(es);
      const uint32_t dn = add_dnode(d, sm);
      if (!counted) add_step(sel_lvl, K_REDUCE_POPC, 0, dn, popc_tiles);  // each tile emits the partials of kPopcTileRows / trows consumer tiles
      // Tiles that stage few values (a sparse selection, or a nested one whose
      // staged values are the few "other" ones) go to the one-wave kernel: on the
      // column's average a tile may fill two thirds of that kernel's stage.
      bool one_wave = wave_scatter && (es == 4 || es == 8) && !(d.flags & (NF_DICT_VALUES | NF_RLE_VALUES));
      if (one_wave) {
        const ColRef& v = d.c1;
        const uint64_t bits = v.kind == COL_CONST ? 0u : (v.kind == COL_FBW ? v.width : 8u * es);
        one_wave = (uint64_t)v.rows * bits / 8 * trows * 3 <= (uint64_t)kWaveStageBytes * 2 * rows;
      }
      // the rank-scatter bodies store the vector holding a column's last row whole:
      // the destination must be a column of its own that starts on a 16-byte
      // boundary and owns the bytes up to the next one (alloc_out / alloc_work
      // hand out kAlign-sized blocks)
      if (!nn.is_bool && (d.self.off % 16 != 0 || kAlign % 16 != 0)) fail(NBL_ERR_ARGUMENT, "rank-scatter destination is not a column of its own");
      const bool lds_dict = lds_dictionary && (d.flags & NF_DICT_VALUES) && es == 8 && d.c0.rows > kTileBytes / 8 && d.c0.rows <= kDictLdsEntries;
      if (nn.is_bool) add_step(fin, K_SELECT_BITS, 0, dn, nt);
      else if (str_dict) add_step(fin, K_DICT_STR, es, dn, div_up32(nt, kDictStrTiles));
      else if (lds_dict) add_step(fin, K_DICT_LDS, es, dn, div_up32(nt, kDictLdsTiles));
      else add_step(fin, one_wave ? K_SELECT_SCATTER_WAVE : K_SELECT_SCATTER, es, dn, nt);
      nn.ref = d.self; nn.ready = fin + 1;
      break;
    }
    case ET_BLOCKBITPACKING:
    case ET_FOR: {
      // the per-segment arrays (baselines, widths, offsets) are small: whatever
      // their encoding, the unpack reads them as element arrays in work
      auto as_array = [&](int slot, uint32_t ces) {
        PNode& ch = C(slot);
        if (ch.ref.kind == COL_MAT_WORK) return ch.ref;
        NodeDesc m{};
        m.self.kind = COL_MAT_WORK; m.self.off = alloc_work((uint64_t)nn.seg_count * ces);
        m.self.rows = nn.seg_count; m.self.es = (uint8_t)ces;
        m.c0 = ch.ref; m.c0.rows = nn.seg_count; m.rows = nn.seg_count;
        const uint32_t mn = add_dnode(m);
        add_step(ch.ready, leaf_kernel(ch.ref, ces), ces, mn, div_up32(nn.seg_count, kTileBytes / ces));
        lvl = std::max(lvl, ch.ready + 1);
        return m.self;
      };
      NodeDesc d{};
      d.c0 = as_array(0, es);
      d.c1 = as_array(1, 1);
      d.sel = as_array(2, nn.enc == ET_FOR ? 8 : 4);
      d.self = mat_ref(es);
      d.rows = rows;
      d.aux = nn.payload_bytes; d.aux_off = nn.payload_off;
      d.aux2 = (uint64_t)nn.seg_size | ((uint64_t)nn.seg_first << 32);
      if (nn.enc == ET_FOR) d.flags |= NF_FRAME_BITS;
      const uint32_t dn = add_dnode(d);
      add_step(lvl, K_FRAME_UNPACK, es, dn, div_up32(rows, tile_rows));
      nn.ref = d.self; nn.ready = lvl + 1;
      break;
    }
    default:
      fail(NBL_ERR_UNSUPPORTED, "EncodingType %d is not decoded on the GPU", (int)nn.enc);
  }

  if (root && keep_dictionary) {
    if (nn.enc == ET_DICTIONARY) keep_dictionary_of(idx, /*with_indices=*/true);
    else if (nn.enc == ET_NULLABLE && nodes[(size_t)nn.child[0]].enc == ET_DICTIONARY) keep_dictionary_of(nn.child[0], false);
  }
  if (root) {
    PNode& r = nodes[idx];
    if (r.is_bool && r.ref.kind == COL_MAT_OUT) {
      // an RLE<bool> root has written its bytes already
    } else if (r.is_bool) {  // bitmap (or constant) -> one byte per row
      NodeDesc d{};
      d.self = *root_dst; d.self.rows = rows; d.self.es = 1;
      d.c0 = r.ref; d.rows = rows;
      const uint32_t dn = add_dnode(d);
      add_step(r.ready, K_BITS_TO_BYTES, 1, dn, div_up32(rows, kTileBytes));
      r.ready += 1;
    } else if (r.ref.kind != COL_MAT_OUT) {  // leaf root: unpack / copy / fill into out
      NodeDesc d{};
      d.self = *root_dst; d.self.rows = rows; d.self.es = (uint8_t)es;
      d.c0 = r.ref; d.rows = rows;
      const uint32_t dn = add_dnode(d);
      add_step(r.ready, leaf_kernel(r.ref, es), es, dn, div_up32(rows, tile_rows));
      r.ready += 1;
    }
  }
}

void nbl_plan::plan_chunk(uint32_t c, uint64_t off, uint64_t size) {
  chunk_copies.clear();
  nodes.clear();  // a tree is needed only while its chunk is placed
  cur_chunk = (int)c;
  const int root = parse(off, size, (int)c, 0);
  const PNode r = nodes[root];
  nbl_chunk_layout& L = layouts[c];
  memset(&L, 0, sizeof L);
  L.rows = r.rows; L.encoding_type = r.enc; L.data_type = r.dtype;
  L.nullable = r.enc == ET_NULLABLE;
  L.elem_bytes = r.is_bool ? 1 : r.es;
  L.non_null_rows = L.nullable ? nodes[r.child[0]].rows : r.rows;
  L.values_bytes = (uint64_t)r.rows * L.elem_bytes;
  L.values_offset = alloc_out(L.values_bytes);
  layout_mask[c] = 0;
  if (L.nullable) {
    L.nulls_bytes = ((uint64_t)r.rows + 63) / 64 * 8;
    L.nulls_offset = alloc_out(L.nulls_bytes);
    layout_mask[c] |= LM_NULLS;
  }
  ColRef dst{};
  dst.kind = COL_MAT_OUT; dst.off = L.values_offset; dst.rows = r.rows; dst.es = L.elem_bytes;
  cur_str = 0;
  const uint64_t decoded_before = decoded_bytes;  // string_copy adds the string bytes while placing
  const size_t first_step = steps.size();
  place(root, true, &dst);
  {  // the step that writes the chunk's values: the latest one of this chunk
    int best = -1;
    for (size_t i = first_step; i < steps.size(); ++i)
      if (steps[i].kernel != K_COPY_BYTES && (best < 0 || steps[i].level >= steps[(size_t)best].level)) best = (int)i;
    if (best >= 0) steps[(size_t)best].chunk = (int)c;
  }
  if (r.is_string) {
    L.string_bytes = cur_str;
    L.string_offset = alloc_out(cur_str);
    layout_mask[c] |= LM_STRING;
    for (uint32_t dn : chunk_copies) dnodes[dn].self.off += L.string_offset;
  }
  encoded_bytes += size;
  decoded_bytes += L.values_bytes + L.nulls_bytes;
  chunk_alg.push_back(size + (decoded_bytes - decoded_before));
  total_rows += r.rows;
  n_nodes += (uint32_t)nodes.size();
}

// Putting per-thread plans together: each was made over a range of chunks with
// cursors that started at zero, so its indices and offsets move by what the
// plans in front of it hold -- exactly where a single walk over all chunks
// would have put them (every allocation is a multiple of kAlign long).
// reserve_behind() gives a plan its place (called in chunk order, by one
// thread), copy_from() moves its tables there (ranges do not overlap: one
// thread per plan).
nbl_plan::MergeBase nbl_plan::reserve_behind(const nbl_plan& sub, size_t chunk_base) {
  MergeBase b{dnodes.size(), steps.size(), chunk_base, cur_out, cur_work, n_parts, n_totals, n_look};
  dnodes.resize(dnodes.size() + sub.dnodes.size());
  dn_mask.resize(dnodes.size());
  steps.resize(steps.size() + sub.steps.size());
  chunk_alg.resize(chunk_base + sub.chunk_alg.size());
  cur_out += sub.cur_out; cur_work += sub.cur_work;
  n_parts += sub.n_parts; n_totals += sub.n_totals; n_look += sub.n_look;
  n_nodes += sub.n_nodes;
  encoded_bytes += sub.encoded_bytes; decoded_bytes += sub.decoded_bytes; total_rows += sub.total_rows;
  max_depth = std::max(max_depth, sub.max_depth);
  return b;
}

void nbl_plan::copy_from(const nbl_plan& sub, const MergeBase& b) {
  for (size_t i = 0; i < sub.dnodes.size(); ++i) {
    NodeDesc d = sub.dnodes[i];
    const uint8_t m = sub.dn_mask[i];
    ColRef* refs[4] = {&d.self, &d.c0, &d.c1, &d.sel};
    for (ColRef* r : refs) {
      if (r->kind == COL_MAT_OUT) r->off += b.out;
      else if (r->kind == COL_MAT_WORK || r->kind == COL_BITS_WORK) r->off += b.work;
    }
    if (m & DM_AUX_WORK) d.aux_off += b.work;
    if (m & DM_AUX_OUT) d.aux_off += b.out;
    if (m & DM_PARTS) d.part_base += b.parts;
    if (m & DM_TOTALS) d.total_idx += b.totals;
    if (m & DM_LOOK) d.look_base += b.look;
    if (m & DM_PARTS2) d.part_base2 += b.parts;
    if (m & DM_TOTALS2) d.total_idx2 += b.totals;
    if (m & DM_AUX3_WORK) d.aux3 += b.work;
    dnodes[b.dnode + i] = d;
    dn_mask[b.dnode + i] = m;
  }
  for (size_t i = 0; i < sub.steps.size(); ++i) {
    Step st = sub.steps[i];
    st.dnode += (uint32_t)b.dnode;
    if (st.aux_base != kNoAux) st.aux_base += (uint32_t)b.parts;
    if (st.chunk >= 0) st.chunk += (int)b.chunk;
    if (st.owner >= 0) st.owner += (int)b.chunk;
    steps[b.step + i] = st;
  }
  for (size_t i = 0; i < sub.layouts.size(); ++i) {
    nbl_chunk_layout L = sub.layouts[i];
    const uint8_t m = sub.layout_mask[i];
    L.values_offset += b.out;
    if (m & LM_NULLS) L.nulls_offset += b.out;
    if (m & LM_STRING) L.string_offset += b.out;
    if (m & LM_DICT_ALPHABET) L.dict_alphabet_offset += b.out;
    if (m & LM_DICT_INDICES) L.dict_indices_offset += b.out;
    layouts[b.chunk + i] = L;
    layout_mask[b.chunk + i] = m;
    chunk_alg[b.chunk + i] = sub.chunk_alg[i];
  }
}

// A step is free when nothing on the device depends on it and it depends on
// nothing: the single unpack / copy / fill of a leaf-only chunk, and the string
// blob copies.  Such steps need not wait for the last level.  They are cut into
// tile ranges and dealt out over ALL levels so that every level carries about
// the same number of tiles: the preparatory steps of deep cascades (tile sums,
// scans, run ends -- a few hundred small, latency-bound tiles per level) then
// run underneath bandwidth-bound work instead of holding the GPU on their own.
void nbl_plan::spread_free_steps(int last_level) {
  const size_t n_levels = (size_t)last_level + 1;
  std::vector<uint32_t> steps_of_chunk(layouts.size(), 0);
  for (const Step& st : steps)
    if (st.owner >= 0) ++steps_of_chunk[(size_t)st.owner];
  std::vector<uint64_t> fixed(n_levels, 0);
  std::vector<size_t> free_steps;
  uint64_t free_tiles = 0, all_tiles = 0;
  for (size_t i = 0; i < steps.size(); ++i) {
    const Step& st = steps[i];
    const bool is_free = st.kernel == K_COPY_BYTES || (st.owner >= 0 && steps_of_chunk[(size_t)st.owner] == 1);
    all_tiles += st.n_tiles;
    if (is_free) { free_steps.push_back(i); free_tiles += st.n_tiles; }
    else fixed[(size_t)st.level] += st.n_tiles;
  }
  if (free_tiles == 0) return;
  // Water filling: the free tiles raise the emptiest levels to a common height.
  // With plenty of them every level ends up with the same number of tiles; with
  // few, they are shared out among the preparatory levels -- each of which costs
  // its chain of dependent loads whatever else it carries -- instead of all
  // going into the first one.
  std::vector<uint64_t> room(n_levels, 0);
  {
    std::vector<size_t> order(n_levels);
    for (size_t l = 0; l < n_levels; ++l) order[l] = l;
    std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return fixed[a] < fixed[b]; });
    uint64_t pool = free_tiles, height = fixed[order[0]];
    size_t k = 1;  // levels order[0..k) are being raised together
    while (pool > 0) {
      const uint64_t next = k < n_levels ? fixed[order[k]] : UINT64_MAX;
      const uint64_t need = next == UINT64_MAX ? UINT64_MAX : (next - height) * k;  // tiles to reach the next level's height
      if (need <= pool) { pool -= need; height = next; ++k; }
      else { height += (pool + k - 1) / k; pool = 0; }
    }
    for (size_t i = 0; i < k && i < n_levels; ++i) room[order[i]] = height > fixed[order[i]] ? height - fixed[order[i]] : 0;
  }
  (void)all_tiles;
  size_t level = 0;
  std::vector<Step> pieces;
  for (size_t i : free_steps) {
    Step st = steps[i];
    const uint64_t whole_alg = st.alg;
    const uint32_t whole_tiles = st.n_tiles;
    uint32_t done = 0;
    uint64_t alg_done = 0;
    bool first = true;
    while (done < whole_tiles) {
      while (level + 1 < n_levels && room[level] == 0) ++level;
      uint32_t take = whole_tiles - done;
      if (level + 1 < n_levels && take > room[level]) take = (uint32_t)room[level];
      Step piece = st;
      piece.filler = true;
      piece.level = (int)level;
      piece.tile_begin = st.tile_begin + done;
      piece.n_tiles = take;
      done += take;
      // bytes in proportion to tiles; the last piece takes the remainder so the sum stays exact
      piece.alg = done == whole_tiles ? whole_alg - alg_done : whole_alg * take / whole_tiles;
      alg_done += piece.alg;
      if (level + 1 < n_levels) room[level] -= take;
      if (first) { steps[i] = piece; first = false; }
      else pieces.push_back(piece);
    }
  }
  steps.insert(steps.end(), pieces.begin(), pieces.end());
}

// Schedules the steps and lays out the work buffer.
//
// Levels: a step's level is its dependency depth inside its chunk.  Every
// chunk is then shifted as late as possible, so that the steps writing final
// outputs of ALL chunks share the last level: the preparatory levels hold only
// the small scan / run-end / bitmap steps of the deeper cascades, and the last
// one is a single large launch that mixes bandwidth-bound tiles (unpack, copy)
// with latency-bound ones (run expansion, rank scatter).
// Launches: fused (default) = one launch per level, tiles sorted by operation;
// otherwise one launch per (level, kernel, element size).
// NBL_PLAN_TIMING=1 in the environment: nbl_plan_batch prints the milliseconds of its phases (tuning aid).
static const bool g_plan_timing = getenv("NBL_PLAN_TIMING") != nullptr;
struct PhaseClock {
  std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
  void mark(const char* what) {
    if (!g_plan_timing) return;
    const auto n = std::chrono::steady_clock::now();
    fprintf(stderr, "[nbl_plan] %-12s %7.3f ms\n", what, std::chrono::duration<double, std::milli>(n - t).count());
    t = n;
  }
};

void nbl_plan::finish() {
  PhaseClock clock;
  int last_level = 0;
  std::vector<int> chunk_last(layouts.size(), 0);
  for (const Step& st : steps) {
    last_level = std::max(last_level, st.level);
    if (st.owner >= 0) chunk_last[(size_t)st.owner] = std::max(chunk_last[(size_t)st.owner], st.level);
  }
  for (Step& st : steps)
    if (st.owner >= 0) st.level += last_level - chunk_last[(size_t)st.owner];
  for (Step& st : steps)
    if (st.chunk >= 0) st.alg = chunk_alg[(size_t)st.chunk];
  {
    // The one-wave rank scatter is a launch of its own, and what it buys is
    // tolerance to memory latency: it pays where the dependent loads of a tile
    // go to memory under the level's own write stream.  Measured on stripes of
    // 100,000-row sparse float columns: decoded once with cold caches (a reader's
    // case) the one-wave launch is level at 7,500 rank-scatter tiles and wins by
    // 8 / 11% at 31,000 / 62,500; decoded over and over, so that a small stripe's
    // inputs stay in the memory-side cache, it loses 10% at 62,500, breaks even
    // at 78,000 (1.3 GB of output per step) and wins by 13-20% from 94,000 up.
    // A launch of a few thousand tiles loses either way (a 100-column stripe:
    // 3-7%), hence the second bound per element size.  Fewer tiles run as
    // four-wave tiles inside the level's fused launch.
    std::vector<uint64_t> scatter_tiles((size_t)last_level + 1, 0), wave_tiles((size_t)(last_level + 1) * 4, 0);
    for (const Step& st : steps) {
      if (st.kernel == K_SELECT_SCATTER || st.kernel == K_SELECT_SCATTER_WAVE) scatter_tiles[(size_t)st.level] += st.n_tiles;
      if (st.kernel == K_SELECT_SCATTER_WAVE) wave_tiles[(size_t)st.level * 4 + es_slot(st.es)] += st.n_tiles;
    }
    for (Step& st : steps)
      if (st.kernel == K_SELECT_SCATTER_WAVE &&
          (scatter_tiles[(size_t)st.level] < wave_min_tiles ||
           wave_tiles[(size_t)st.level * 4 + es_slot(st.es)] * 8 < wave_min_tiles))
        st.kernel = K_SELECT_SCATTER;
  }
  {
    // The LDS-alphabet dictionary kernel is a launch of its own as well (40 KiB of
    // LDS per workgroup).  Alone it runs at the rate its bytes move (5.5 TB/s
    // against 4.1-4.4 for the same columns gathered from memory), but a launch
    // costs its ramp and tail and what its tiles would have hidden inside the
    // fused launch: below lds_dict_min_tiles tiles of a level the nodes keep the
    // fused launch's gather.
    // The same holds for the compact-views kernel, counted by itself.
    std::vector<uint64_t> lds_tiles((size_t)(last_level + 1) * 2, 0);
    auto tiles_of = [&](const Step& st) { return (uint64_t)div_up32(dnodes[st.dnode].rows, kTileBytes / 8); };
    auto slot_of = [&](const Step& st) { return (size_t)st.level * 2 + (st.kernel == K_DICT_STR ? 1u : 0u); };
    for (const Step& st : steps)
      if (st.kernel == K_DICT_LDS || st.kernel == K_DICT_STR) lds_tiles[slot_of(st)] += tiles_of(st);
    for (Step& st : steps)
      if ((st.kernel == K_DICT_LDS || st.kernel == K_DICT_STR) && lds_tiles[slot_of(st)] < lds_dict_min_tiles) {
        st.kernel = (dnodes[st.dnode].flags & NF_DICT_VALUES) ? K_SELECT_SCATTER : K_DICT_GATHER;
        st.n_tiles = (uint32_t)tiles_of(st);
      }
  }
  if (fused && last_level > 0) spread_free_steps(last_level);
  clock.mark("levels");
  {  // stable counting sort by (level, filler, kernel, element size): wide tables have ~10 steps per column
    // within a level: the level's own fusable steps, the fillers, then the steps that launch on their own
    auto key = [](const Step& st) {
      const size_t cls = !kernel_is_fusable(st.kernel) ? 2 : st.filler ? 1 : 0;
      return (((size_t)st.level * 3 + cls) * K_COUNT + st.kernel) * 4 + es_slot(st.es);
    };
    // The order is worked out on step numbers; the steps themselves move once.
    std::vector<uint32_t> first((size_t)(last_level + 1) * 3 * K_COUNT * 4 + 1, 0);
    for (const Step& st : steps) ++first[key(st) + 1];
    for (size_t i = 1; i < first.size(); ++i) first[i] += first[i - 1];
    RawVec<uint32_t> perm(steps.size());
    {
      std::vector<uint32_t> next(first.begin(), first.end() - 1);
      for (size_t k = 0; k < steps.size(); ++k) perm[next[key(steps[k])]++] = (uint32_t)k;
    }
    // Rank scatters of one (level, element size) are grouped by what their tiles
    // execute -- which values child is resolved inside, how values and bitmap
    // are read, how dense the selection is -- so that the workgroups resident
    // on a CU together run the same code path at the same pace: a stripe of
    // thousands of columns mixes all of them otherwise (-2.7% on the output level
    // of a 5,000-column stripe; a 100-column stripe is indifferent to slightly
    // worse, so only levels with hundreds of scatters are grouped).  Chunk order
    // is kept inside a group.
    constexpr size_t kGroupVariantsFrom = 256;
    {
      // a group is ordered by a key taken once per step: a stable radix sort of
      // (key, step number) pairs, three passes of eleven bits
      RawVec<uint32_t> ka, kb, ia, ib;
      auto reorder = [&](size_t i, size_t j, auto&& key_of) {
        const size_t n = j - i;
        ka.resize(n); kb.resize(n); ia.resize(n); ib.resize(n);
        for (size_t k = 0; k < n; ++k) {
          ia[k] = perm[i + k];
          ka[k] = key_of(steps[ia[k]]);
        }
        for (uint32_t shift = 0; shift < 33; shift += 11) {
          uint32_t count[2049] = {0};
          for (size_t k = 0; k < n; ++k) ++count[((ka[k] >> shift) & 2047u) + 1];
          for (uint32_t d = 1; d <= 2048; ++d) count[d] += count[d - 1];
          for (size_t k = 0; k < n; ++k) {
            const uint32_t at = count[(ka[k] >> shift) & 2047u]++;
            kb[at] = ka[k];
            ib[at] = ia[k];
          }
          ka.swap(kb);
          ia.swap(ib);
        }
        for (size_t k = 0; k < n; ++k) perm[i + k] = ia[k];
      };
      auto cached = [](const Step& st) -> uint32_t { return st.order_key; };
      for (size_t g = 0; g + 1 < first.size(); ++g) {
        const size_t i = first[g], j = first[g + 1];
        if (j - i < 2) continue;
        const uint32_t kernel = steps[perm[i]].kernel;
        // Bitmap tiles of SparseBool columns cost in proportion to the positions
        // they read (a column of 20,000 takes ten rounds of loads, one of 1,000
        // a single one): the long ones go first, so that the level ends with
        // short tiles filling the gaps instead of a few long ones running alone.
        if ((kernel == K_SELECT_SCATTER || kernel == K_SELECT_SCATTER_WAVE || kernel == K_SPARSE_BITS) && j - i >= kGroupVariantsFrom)
          reorder(i, j, cached);
        // A compact-views workgroup has its CU to itself and first copies its
        // alphabet's table: the ones with the largest tables go first, so the
        // launch ends on the short ones.
        if (kernel == K_DICT_STR)
          reorder(i, j, [&](const Step& x) {
            const NodeDesc& d = dnodes[x.dnode];
            return ~((d.flags & NF_DICT_VALUES) ? d.c0.rows : d.c1.rows);
          });
      }
    }
    RawVec<Step> sorted(steps.size());
    for (size_t k = 0; k < steps.size(); ++k) sorted[k] = steps[perm[k]];
    steps.swap(sorted);
  }
  clock.mark("sort");
  {
    size_t n_tiles_total = 0;
    for (const Step& st : steps) n_tiles_total += st.n_tiles;
    tiles.resize(n_tiles_total);
    tile_aux.resize(n_tiles_total);
  }
  size_t n_filled = 0;  // entries of the two tables written so far
  for (size_t i = 0; i < steps.size();) {
    size_t j = i;
    Launch L{steps[i].kernel, steps[i].es, (uint32_t)n_filled, 0, steps[i].level, 0, std::string()};
    const bool fuse = fused && kernel_is_fusable(steps[i].kernel);
    size_t first_filler = n_filled;  // where the level's filler tiles start in the table
    bool fillers = false;
    while (j < steps.size() && steps[j].level == steps[i].level &&
           (fuse ? kernel_is_fusable(steps[j].kernel) : (steps[j].kernel == steps[i].kernel && steps[j].es == steps[i].es))) {
      const uint32_t op = op_of(steps[j].kernel, steps[j].es) << kOpShift;
      const bool scatter = steps[j].aux_base != kNoAux;
      const uint32_t aux_base = scatter ? steps[j].aux_base : 0u;
      if (steps[j].filler && !fillers) { fillers = true; first_filler = n_filled; }
      {
        TileRef* tr = tiles.data() + n_filled;
        uint32_t* ax = tile_aux.data() + n_filled;
        const uint32_t dn = steps[j].dnode, t0 = steps[j].tile_begin, nt = steps[j].n_tiles;
        const uint32_t per = steps[j].kernel == K_DICT_LDS ? kDictLdsTiles : steps[j].kernel == K_DICT_STR ? kDictStrTiles : 1u;
        for (uint32_t t = 0; t < nt; ++t) tr[t] = TileRef{dn, op | (t0 + t)};
        for (uint32_t t = 0; t < nt; ++t) ax[t] = scatter ? aux_base + (t0 + t) * per : 0u;
        n_filled += nt;
      }
      L.alg_bytes += steps[j].alg;
      ++j;
    }
    L.n_tiles = (uint32_t)n_filled - L.tile_begin;
    if (fuse && fillers && first_filler > L.tile_begin) {
      // The level's own tiles (chained scans, bitmap builds: each a chain of trips
      // to memory) hold a workgroup slot for several times as long as a filler
      // tile moves its 16 KiB.  Launched first they fill every slot of the chip
      // and the fillers' bandwidth waits for them; dealt out among the fillers
      // over the first three quarters of the launch, both run side by side.
      const size_t n_own = first_filler - L.tile_begin, n_fill = n_filled - first_filler;
      const size_t span = std::max(n_own, (n_own + n_fill) * 3 / 4);  // table slots the own tiles are spread over
      RawVec<TileRef> merged;
      RawVec<uint32_t> merged_aux;
      merged.reserve(n_own + n_fill);
      merged_aux.reserve(n_own + n_fill);
      size_t o = 0, f = 0;
      for (size_t pos = 0; pos < n_own + n_fill; ++pos) {
        const bool own = o < n_own && (f >= n_fill || o * span <= pos * n_own);
        const size_t src = own ? L.tile_begin + o++ : first_filler + f++;
        merged.push_back(tiles[src]);
        merged_aux.push_back(tile_aux[src]);
      }
      std::copy(merged.begin(), merged.end(), tiles.begin() + (ptrdiff_t)L.tile_begin);
      std::copy(merged_aux.begin(), merged_aux.end(), tile_aux.begin() + (ptrdiff_t)L.tile_begin);
    }
    if (fuse) {
      L.kernel = L.level == last_level ? K_LEVEL_LAST : K_LEVEL;
      L.es = 0;
      L.name = L.kernel == K_LEVEL_LAST ? "k_level<true>" : "k_level<false>";
    } else {
      L.name = kKernelNames[L.kernel];
      if (kernel_has_es(L.kernel)) L.name += "<" + std::to_string(L.es) + ">";
    }
    launches.push_back(L);
    i = j;
  }
  clock.mark("tile table");
  // work = [NodeDesc[] | TileRef[] | tile_aux[]] [ctrl] [parts] [intermediates...]
  const uint64_t nd_bytes = align_up(dnodes.size() * sizeof(NodeDesc), kAlign);
  const uint64_t tl_bytes = align_up(tiles.size() * sizeof(TileRef), kAlign);
  const uint64_t ax_bytes = align_up(tile_aux.size() * sizeof(uint32_t), kAlign);
  tiles_off = nd_bytes;
  aux_off = nd_bytes + tl_bytes;
  desc_bytes = nd_bytes + tl_bytes + ax_bytes;
  ctrl_off = desc_bytes;
  look_off = align_up(4 + (uint64_t)n_totals * 4, 8);  // relative to ctrl_off
  ctrl_bytes = align_up(look_off + (uint64_t)n_look * 8, kAlign);
  parts_off = ctrl_off + ctrl_bytes;
  const uint64_t scratch_base = align_up(parts_off + (uint64_t)n_parts * 4, kAlign);
  // intermediates were allocated from a cursor starting at 0: rebase them
  for (NodeDesc& d : dnodes) {
    ColRef* refs[4] = {&d.self, &d.c0, &d.c1, &d.sel};
    for (ColRef* r : refs)
      if (r->kind == COL_MAT_WORK || r->kind == COL_BITS_WORK) r->off += scratch_base;
  }
  for (size_t i = 0; i < dnodes.size(); ++i)
    if (dn_mask[i] & DM_AUX_WORK) dnodes[i].aux_off += scratch_base;  // run-end arrays, compact string views
  for (size_t i = 0; i < dnodes.size(); ++i)
    if (dn_mask[i] & DM_AUX3_WORK) dnodes[i].aux3 += scratch_base;
  work_bytes = scratch_base + cur_work + kAlign;
  out_bytes = cur_out + kAlign;  // slack: staged 16-byte loads may run a few bytes past a column
  clock.mark("layout");
}

// =============================== C ABI =====================================
extern "C" {

const char* nbl_last_error(void) { return g_last_error.c_str(); }
void nbl_set_last_error(const char* msg) { g_last_error = msg ? msg : ""; }  // for the library's other translation units
const char* nbl_build_arch(void) { return "gfx950"; }
#ifndef NBL_SOURCE_ID
#define NBL_SOURCE_ID "unknown"
#endif
const char* nbl_build_source_id(void) { return NBL_SOURCE_ID; }

nbl_status nbl_split_stream_ex(const uint8_t* stream, uint64_t size, uint64_t* offsets, uint64_t* sizes, uint8_t* compression,
                               uint32_t capacity, uint32_t* n_chunks) {
  if ((!stream && size) || !n_chunks) { g_last_error = "null argument"; return NBL_ERR_ARGUMENT; }
  uint64_t pos = 0; uint32_t n = 0;
  while (pos < size) {  // InMemoryChunkedStream::nextChunk, velox/ChunkedStream.cpp:36-72
    if (size - pos < 5) { g_last_error = "Read beyond end of stream"; return NBL_ERR_CORRUPT; }
    uint32_t len; memcpy(&len, stream + pos, 4);
    const uint8_t comp = stream[pos + 4];
    pos += 5;
    if (len > size - pos) { g_last_error = "Read beyond end of stream"; return NBL_ERR_CORRUPT; }
    if (comp != 0 && comp != 1 && comp != 3) { g_last_error = "Unexpected stream compression type: " + std::to_string((int)comp); return NBL_ERR_CORRUPT; }
    if (n < capacity && offsets && sizes) { offsets[n] = pos; sizes[n] = len; if (compression) compression[n] = comp; }
    ++n; pos += len;
  }
  *n_chunks = n;
  return NBL_OK;
}

nbl_status nbl_split_stream(const uint8_t* stream, uint64_t size, uint64_t* offsets, uint64_t* sizes,
                            uint32_t capacity, uint32_t* n_chunks) {
  if ((!stream && size) || !n_chunks) { g_last_error = "null argument"; return NBL_ERR_ARGUMENT; }
  uint64_t pos = 0; uint32_t n = 0;
  while (pos < size) {  // InMemoryChunkedStream::nextChunk, velox/ChunkedStream.cpp:36-72
    if (size - pos < 5) { g_last_error = "Read beyond end of stream"; return NBL_ERR_CORRUPT; }
    uint32_t len; memcpy(&len, stream + pos, 4);
    const uint8_t comp = stream[pos + 4];
    pos += 5;
    if (len > size - pos) { g_last_error = "Read beyond end of stream"; return NBL_ERR_CORRUPT; }
    if (comp == 1 || comp == 3) { g_last_error = "compressed chunk (Zstd/Lz4): decompress on the host first"; return NBL_ERR_UNSUPPORTED; }
    if (comp != 0) { g_last_error = "Unexpected stream compression type: " + std::to_string((int)comp); return NBL_ERR_CORRUPT; }
    if (n < capacity && offsets && sizes) { offsets[n] = pos; sizes[n] = len; }
    ++n; pos += len;
  }
  *n_chunks = n;
  return NBL_OK;
}

nbl_status nbl_plan_batch(const uint8_t* base, const uint64_t* offsets, const uint64_t* sizes, uint32_t n_chunks,
                          const nbl_options* options, nbl_plan** plan_out) {
  if (!plan_out || (n_chunks && (!base || !offsets || !sizes))) { g_last_error = "null argument"; return NBL_ERR_ARGUMENT; }
  *plan_out = nullptr;
  nbl_plan* p = new nbl_plan();
  try {
    p->base = base;
    p->varint_rows = options && options->struct_size >= 8 && options->use_varint_row_count;
    p->fused = !(options && options->struct_size >= 12 && (options->flags & NBL_PLAN_PER_KERNEL_LAUNCHES));
    p->keep_dictionary = options && options->struct_size >= 12 && (options->flags & NBL_PLAN_KEEP_DICTIONARY);
    p->wave_scatter = !(options && options->struct_size >= 12 && (options->flags & NBL_PLAN_FOUR_WAVE_SCATTER));
    p->lds_dictionary = !(options && options->struct_size >= 12 && (options->flags & NBL_PLAN_NO_LDS_DICTIONARY));
    if (options && options->struct_size >= 16 && options->wave_scatter_min_tiles) p->wave_min_tiles = options->wave_scatter_min_tiles;
    if (options && options->struct_size >= 20 && options->lds_dictionary_min_tiles) p->lds_dict_min_tiles = options->lds_dictionary_min_tiles;
    p->layouts.resize(n_chunks);
    p->layout_mask.resize(n_chunks);
    p->batch_chunks = n_chunks;
    for (uint32_t c = 0; c < n_chunks; ++c) {
      if (offsets[c] + sizes[c] < offsets[c]) fail(NBL_ERR_ARGUMENT, "chunk %u: offset + size overflows", c);
      p->span = std::max(p->span, offsets[c] + sizes[c]);
    }
    auto reserve_for = [](nbl_plan& q, size_t chunks) {
      q.nodes.reserve(64);
      q.dnodes.reserve(chunks * 4);
      q.dn_mask.reserve(chunks * 4);
      q.steps.reserve(chunks * 8);
      q.chunk_alg.reserve(chunks);
    };
    // The chunks are independent (so are their headers, spread over the whole
    // stripe: the walk waits on memory), so a large batch is walked by several
    // threads, each over a range of chunks with cursors of its own, and the
    // per-thread plans are put together in chunk order.
    uint32_t threads = options && options->struct_size >= 24 ? options->plan_threads : 0;
    if (threads == 0) {
      if (const char* e = getenv("NBL_PLAN_THREADS")) threads = (uint32_t)atoi(e);
      if (threads == 0) threads = std::min<uint32_t>(std::max(1u, std::thread::hardware_concurrency()), kMaxPlanThreads);
      threads = std::min<uint32_t>(threads, n_chunks / kChunksPerPlanThread);  // small batches: not worth starting threads for
    }
    threads = std::max(1u, std::min<uint32_t>(std::min<uint32_t>(threads, 64), std::max(1u, n_chunks)));
    PhaseClock clock;
    if (threads <= 1) {
      reserve_for(*p, n_chunks);
      for (uint32_t c = 0; c < n_chunks; ++c) {
        // the headers of a wide table are spread over hundreds of megabytes: have
        // the prefix of a later chunk on its way while this one is walked
        if (c + 8 < n_chunks && sizes[c + 8]) __builtin_prefetch(base + offsets[c + 8]);
        p->plan_chunk(c, offsets[c], sizes[c]);
      }
    } else {
      std::vector<std::unique_ptr<nbl_plan>> subs(threads);
      std::vector<PlanError> errors(threads, PlanError{NBL_OK, std::string()});
      std::vector<uint32_t> first(threads + 1);
      for (uint32_t t = 0; t <= threads; ++t) first[t] = (uint32_t)((uint64_t)n_chunks * t / threads);
      auto walk = [&](uint32_t t) {
        try {
          subs[t].reset(new nbl_plan());
          nbl_plan& q = *subs[t];
          q.base = base; q.varint_rows = p->varint_rows; q.fused = p->fused; q.keep_dictionary = p->keep_dictionary;
          q.wave_scatter = p->wave_scatter; q.lds_dictionary = p->lds_dictionary; q.batch_chunks = n_chunks;
          const uint32_t c0 = first[t], c1 = first[t + 1];
          q.layouts.resize(c1 - c0);
          q.layout_mask.resize(c1 - c0);
          reserve_for(q, c1 - c0);
          for (uint32_t c = c0; c < c1; ++c) {
            if (c + 8 < c1 && sizes[c + 8]) __builtin_prefetch(base + offsets[c + 8]);
            q.plan_chunk(c - c0, offsets[c], sizes[c]);
          }
        } catch (const PlanError& e) {
          errors[t] = e;
        } catch (const std::exception& e) {
          errors[t] = PlanError{NBL_ERR_ARGUMENT, e.what()};
        }
      };
      // runs job(t) for every range: range 0 on this thread, the others on threads of their own
      auto on_all = [&](auto&& job) {
        std::vector<std::thread> pool;
        pool.reserve(threads - 1);
        uint32_t started = 1;
        try {
          for (; started < threads; ++started) pool.emplace_back(job, started);
        } catch (const std::system_error&) {
          // no more threads to be had: this one takes the ranges that got none
        }
        job(0u);
        for (uint32_t t = started; t < threads; ++t) job(t);
        for (std::thread& th : pool) th.join();
      };
      on_all(walk);
      clock.mark("walk");
      // the error a single walk would have met first: that of the lowest range
      for (uint32_t t = 0; t < threads; ++t)
        if (errors[t].status != NBL_OK) throw errors[t];
      size_t nd = 0, ns = 0;
      for (const auto& q : subs) { nd += q->dnodes.size(); ns += q->steps.size(); }
      p->dnodes.reserve(nd); p->dn_mask.reserve(nd); p->steps.reserve(ns);
      std::vector<nbl_plan::MergeBase> bases(threads);
      for (uint32_t t = 0; t < threads; ++t) bases[t] = p->reserve_behind(*subs[t], first[t]);
      on_all([&](uint32_t t) {
        p->copy_from(*subs[t], bases[t]);
        subs[t].reset();
      });
    }
    clock.mark(threads <= 1 ? "walk" : "merge");
    p->finish();
    p->base = nullptr;  // the host bytes are not needed after planning
  } catch (const PlanError& e) {
    g_last_error = e.msg;
    delete p;
    return e.status;
  } catch (const std::exception& e) {
    g_last_error = e.what();
    delete p;
    return NBL_ERR_ARGUMENT;
  }
  *plan_out = p;
  return NBL_OK;
}

void nbl_plan_free(nbl_plan* plan) { delete plan; }

nbl_status nbl_plan_get_info(const nbl_plan* p, nbl_plan_info* info) {
  if (!p || !info) { g_last_error = "null argument"; return NBL_ERR_ARGUMENT; }
  memset(info, 0, sizeof *info);
  info->n_chunks = (uint32_t)p->layouts.size();
  info->n_nodes = p->n_nodes;
  info->n_launches = (uint32_t)p->launches.size();
  info->max_depth = p->max_depth;
  info->arena_bytes = p->span + kArenaSlop;
  info->out_bytes = p->out_bytes;
  info->work_bytes = p->work_bytes;
  info->descriptor_bytes = p->desc_bytes;
  info->total_rows = p->total_rows;
  info->encoded_bytes = p->encoded_bytes;
  info->decoded_bytes = p->decoded_bytes;
  return NBL_OK;
}

nbl_status nbl_plan_get_chunk_layout(const nbl_plan* p, uint32_t chunk, nbl_chunk_layout* layout) {
  if (!p || !layout || chunk >= p->layouts.size()) { g_last_error = "bad chunk index"; return NBL_ERR_ARGUMENT; }
  *layout = p->layouts[chunk];
  return NBL_OK;
}

nbl_status nbl_plan_get_launch(const nbl_plan* p, uint32_t launch, nbl_launch_info* info) {
  if (!p || !info || launch >= p->launches.size()) { g_last_error = "bad launch index"; return NBL_ERR_ARGUMENT; }
  const Launch& L = p->launches[launch];
  info->kernel = L.name.c_str();
  info->elem_bytes = L.es;
  info->n_tiles = L.n_tiles;
  info->level = (uint32_t)L.level;
  info->reserved = 0;
  info->alg_bytes = L.alg_bytes;
  return NBL_OK;
}

nbl_status nbl_plan_fingerprint(const nbl_plan* p, uint64_t* fingerprint) {
  if (!p || !fingerprint) { g_last_error = "null argument"; return NBL_ERR_ARGUMENT; }
  uint64_t h = 0xcbf29ce484222325ull;  // FNV-1a over 8-byte words (tails byte by byte)
  auto mix = [&](const void* data, size_t bytes) {
    const uint8_t* b = static_cast<const uint8_t*>(data);
    size_t i = 0;
    for (; i + 8 <= bytes; i += 8) { uint64_t w; memcpy(&w, b + i, 8); h = (h ^ w) * 0x100000001b3ull; }
    for (; i < bytes; ++i) h = (h ^ b[i]) * 0x100000001b3ull;
  };
  mix(p->dnodes.data(), p->dnodes.size() * sizeof(NodeDesc));
  mix(p->tiles.data(), p->tiles.size() * sizeof(TileRef));
  mix(p->tile_aux.data(), p->tile_aux.size() * sizeof(uint32_t));
  for (const nbl_chunk_layout& L : p->layouts) mix(&L, sizeof L);
  for (const Launch& L : p->launches) {
    const uint64_t f[6] = {L.kernel, L.es, L.tile_begin, L.n_tiles, (uint64_t)L.level, L.alg_bytes};
    mix(f, sizeof f);
  }
  const uint64_t sizes[12] = {p->span, p->out_bytes, p->work_bytes, p->desc_bytes, p->tiles_off, p->aux_off, p->ctrl_off,
                              p->ctrl_bytes, p->parts_off, p->look_off, p->n_nodes, p->total_rows};
  mix(sizes, sizeof sizes);
  *fingerprint = h;
  return NBL_OK;
}

static nbl_status hip_fail(const char* what, hipError_t e) {
  g_last_error = std::string(what) + ": " + hipGetErrorString(e);
  return NBL_ERR_HIP;
}

nbl_status nbl_plan_upload(nbl_plan* p, void* d_work, void* stream) {
  if (!p || !d_work) { g_last_error = "null argument"; return NBL_ERR_ARGUMENT; }
  // the three descriptor tables go up from where the planner built them (no
  // image of `work` is assembled on the host); the padding between them is never read
  hipStream_t s = static_cast<hipStream_t>(stream);
  uint8_t* work = static_cast<uint8_t*>(d_work);
  hipError_t e = hipSuccess;
  if (!p->dnodes.empty()) e = hipMemcpyAsync(work, p->dnodes.data(), p->dnodes.size() * sizeof(NodeDesc), hipMemcpyHostToDevice, s);
  if (e == hipSuccess && !p->tiles.empty())
    e = hipMemcpyAsync(work + p->tiles_off, p->tiles.data(), p->tiles.size() * sizeof(TileRef), hipMemcpyHostToDevice, s);
  if (e == hipSuccess && !p->tile_aux.empty())
    e = hipMemcpyAsync(work + p->aux_off, p->tile_aux.data(), p->tile_aux.size() * sizeof(uint32_t), hipMemcpyHostToDevice, s);
  if (e != hipSuccess) return hip_fail("hipMemcpyAsync(descriptors)", e);
  // The control words start at zero here and are not touched by the host again
  // until nbl_decode_status: the flag word only gathers bits, and the kernels
  // reset the per-node words they accumulate into, so a decode is launches only.
  e = hipMemsetAsync(work + p->ctrl_off, 0, p->ctrl_bytes, s);
  if (e != hipSuccess) return hip_fail("hipMemsetAsync(control words)", e);
  return NBL_OK;
}

}  // extern "C"

// Consecutive launches of the same kernel form a group and share their boundary
// events: a step records n_groups + 1 events instead of 2 per launch (an event
// costs the stream a few microseconds), and a launch's time is its group's span
// divided by the group's launches (the gaps inside a group are counted in).
struct nbl_profile {
  uint32_t n_launches = 0, max_steps = 0, steps_recorded = 0, n_groups = 0;
  std::vector<uint32_t> group_of;    // per launch
  std::vector<uint32_t> group_size;  // per group
  std::vector<hipEvent_t> events;    // [step][boundary 0 .. n_groups]
};

static nbl_status decode_impl(const nbl_plan* p, nbl_profile* prof, uint32_t step, const void* d_arena, void* d_out,
                              void* d_work, void* stream);

extern "C" {

nbl_status nbl_decode_batch_device(const nbl_plan* p, const void* d_arena, void* d_out, void* d_work, void* stream) {
  return decode_impl(p, nullptr, 0, d_arena, d_out, d_work, stream);
}

nbl_status nbl_profile_create(const nbl_plan* p, uint32_t max_steps, nbl_profile** out) {
  if (!p || !out || max_steps == 0) { g_last_error = "bad argument"; return NBL_ERR_ARGUMENT; }
  nbl_profile* pr = new nbl_profile();
  pr->n_launches = (uint32_t)p->launches.size();
  pr->max_steps = max_steps;
  // NBL_PROFILE_UNGROUPED=1 (tuning aid): every launch is a group of its own
  const char* ungrouped = getenv("NBL_PROFILE_UNGROUPED");
  const bool split = ungrouped && ungrouped[0] == '1';
  for (size_t l = 0; l < p->launches.size(); ++l) {
    if (l == 0 || split || p->launches[l].name != p->launches[l - 1].name) { pr->group_size.push_back(0); }
    pr->group_of.push_back((uint32_t)pr->group_size.size() - 1);
    ++pr->group_size.back();
  }
  pr->n_groups = (uint32_t)pr->group_size.size();
  pr->events.resize((size_t)(pr->n_groups + 1) * max_steps, nullptr);
  for (hipEvent_t& e : pr->events) {
    hipError_t rc = hipEventCreate(&e);
    if (rc != hipSuccess) { nbl_profile_free(pr); g_last_error = std::string("hipEventCreate: ") + hipGetErrorString(rc); return NBL_ERR_HIP; }
  }
  *out = pr;
  return NBL_OK;
}

void nbl_profile_free(nbl_profile* pr) {
  if (!pr) return;
  for (hipEvent_t e : pr->events) if (e) (void)hipEventDestroy(e);
  delete pr;
}

nbl_status nbl_decode_batch_profiled(const nbl_plan* p, nbl_profile* pr, uint32_t step, const void* d_arena, void* d_out,
                                     void* d_work, void* stream) {
  if (!pr || step >= pr->max_steps || !p || pr->n_launches != p->launches.size()) { g_last_error = "bad profile argument"; return NBL_ERR_ARGUMENT; }
  const nbl_status st = decode_impl(p, pr, step, d_arena, d_out, d_work, stream);
  if (st == NBL_OK && step + 1 > pr->steps_recorded) pr->steps_recorded = step + 1;
  return st;
}

nbl_status nbl_profile_collect(nbl_profile* pr, float* mean_ms, uint32_t n_launches) {
  if (!pr || !mean_ms || n_launches != pr->n_launches) { g_last_error = "bad profile argument"; return NBL_ERR_ARGUMENT; }
  for (uint32_t l = 0; l < n_launches; ++l) mean_ms[l] = 0.f;
  if (pr->steps_recorded == 0 || n_launches == 0) return NBL_OK;
  const size_t per_step = (size_t)pr->n_groups + 1;
  hipError_t rc = hipEventSynchronize(pr->events[(size_t)(pr->steps_recorded - 1) * per_step + pr->n_groups]);
  if (rc != hipSuccess) { g_last_error = std::string("hipEventSynchronize: ") + hipGetErrorString(rc); return NBL_ERR_HIP; }
  for (uint32_t s = 0; s < pr->steps_recorded; ++s)
    for (uint32_t l = 0; l < n_launches; ++l) {
      const uint32_t g = pr->group_of[l];
      float ms = 0.f;
      rc = hipEventElapsedTime(&ms, pr->events[(size_t)s * per_step + g], pr->events[(size_t)s * per_step + g + 1]);
      if (rc != hipSuccess) { g_last_error = std::string("hipEventElapsedTime: ") + hipGetErrorString(rc); return NBL_ERR_HIP; }
      mean_ms[l] += ms / (float)pr->group_size[g] / (float)pr->steps_recorded;
    }
  return NBL_OK;
}

}  // extern "C"

static nbl_status decode_impl(const nbl_plan* p, nbl_profile* prof, uint32_t step, const void* d_arena, void* d_out,
                              void* d_work, void* stream) {
  if (!p || !d_work || (!d_arena && p->span) || !d_out) { g_last_error = "null argument"; return NBL_ERR_ARGUMENT; }
  if ((reinterpret_cast<uintptr_t>(d_arena) | reinterpret_cast<uintptr_t>(d_out) | reinterpret_cast<uintptr_t>(d_work)) & 15u) {
    g_last_error = "d_arena, d_out and d_work must be 16-byte aligned";
    return NBL_ERR_ARGUMENT;
  }
  uint8_t* work = static_cast<uint8_t*>(d_work);
  hipError_t e = hipSuccess;
  KArgs a{};
  a.arena = static_cast<const uint8_t*>(d_arena);
  a.out = static_cast<uint8_t*>(d_out);
  a.work = work;
  a.nodes = reinterpret_cast<const NodeDesc*>(work);
  a.status = reinterpret_cast<uint32_t*>(work + p->ctrl_off);
  a.totals = a.status + 1;
  a.parts = reinterpret_cast<uint32_t*>(work + p->parts_off);
  a.look = reinterpret_cast<uint64_t*>(work + p->ctrl_off + p->look_off);
  // every decode has a number of its own (1 .. 2^30 - 1): the chained scans tell
  // this decode's look-back words from those an earlier one left in `work`
  a.gen = (p->decodes.fetch_add(1, std::memory_order_relaxed) % 0x3FFFFFFFu) + 1u;
  const TileRef* tiles = reinterpret_cast<const TileRef*>(work + p->tiles_off);
  const uint32_t* tile_aux = reinterpret_cast<const uint32_t*>(work + p->aux_off);
  const size_t nl = p->launches.size();
  for (size_t li = 0; li < nl; ++li) {
    const Launch& L = p->launches[li];
    a.tiles = tiles + L.tile_begin;
    a.tile_aux = tile_aux + L.tile_begin;
    a.n_tiles = L.n_tiles;
    if (prof && (li == 0 || prof->group_of[li] != prof->group_of[li - 1])) {
      e = hipEventRecord(prof->events[(size_t)step * (prof->n_groups + 1) + prof->group_of[li]], static_cast<hipStream_t>(stream));
      if (e != hipSuccess) return hip_fail("hipEventRecord", e);
    }
    const int rc = nbl_launch_kernel(L.kernel, L.es, &a, stream);
    if (prof && li + 1 == nl) {
      e = hipEventRecord(prof->events[(size_t)step * (prof->n_groups + 1) + prof->n_groups], static_cast<hipStream_t>(stream));
      if (e != hipSuccess) return hip_fail("hipEventRecord", e);
    }
    if (rc != 0) {
      g_last_error = "launch of " + L.name + " failed";
      return NBL_ERR_HIP;
    }
  }
  return NBL_OK;
}

extern "C" {

nbl_status nbl_copy_to_host_async(void* host_dst, const void* dev_src, uint64_t bytes, void* stream) {
  if (bytes == 0) return NBL_OK;
  if (!host_dst || !dev_src) { g_last_error = "null argument"; return NBL_ERR_ARGUMENT; }
  if ((reinterpret_cast<uintptr_t>(host_dst) | reinterpret_cast<uintptr_t>(dev_src)) & 15) {
    g_last_error = "nbl_copy_to_host_async: pointers must be 16-byte aligned";
    return NBL_ERR_ARGUMENT;
  }
  if (nbl_launch_copy_to_host(host_dst, dev_src, bytes, stream) != 0) return hip_fail("copy to host", hipGetLastError());
  return NBL_OK;
}

nbl_status nbl_decode_status(const nbl_plan* p, const void* d_work, void* stream) {
  if (!p || !d_work) { g_last_error = "null argument"; return NBL_ERR_ARGUMENT; }
  uint32_t st = 0;
  hipError_t e = hipMemcpyAsync(&st, static_cast<const uint8_t*>(d_work) + p->ctrl_off, 4, hipMemcpyDeviceToHost,
                                static_cast<hipStream_t>(stream));
  if (e != hipSuccess) return hip_fail("hipMemcpyAsync(status)", e);
  e = hipStreamSynchronize(static_cast<hipStream_t>(stream));
  if (e != hipSuccess) return hip_fail("hipStreamSynchronize", e);
  if (st == 0) return NBL_OK;
  // reported once: the word gathers bits again from the next decode on
  e = hipMemsetAsync(const_cast<uint8_t*>(static_cast<const uint8_t*>(d_work)) + p->ctrl_off, 0, 4, static_cast<hipStream_t>(stream));
  if (e != hipSuccess) return hip_fail("hipMemsetAsync(status)", e);
  std::string m = "inconsistent payload:";
  if (st & ST_RLE_SUM_MISMATCH) m += " run lengths do not add up to the row count;";
  if (st & ST_DICT_INDEX_RANGE) m += " dictionary index out of range;";
  if (st & ST_SELECT_COUNT) m += " selected-row count differs from the values child;";
  if (st & ST_VARINT_COUNT) m += " fewer varints than rows;";
  if (st & ST_STRING_BYTES) m += " string lengths exceed the blob;";
  if (st & ST_FRAME_RANGE) m += " block offset or bit width outside the packed payload;";
  if (st & ST_SCAN_STALL) m += " a chained scan gave up waiting;";
  g_last_error = m;
  return NBL_ERR_DEVICE_DATA;
}

}  // extern "C"

// Descriptor tables shared by the host planner and the HIP kernels.
//
// A planned batch is a flat table of nodes (one per encoding in the cascade
// trees) plus, per kernel launch, a table of tiles.  Everything a kernel needs
// is addressed as a byte offset into one of three device buffers:
//   arena : byte image of the encoded chunks (read only)
//   out   : decoded outputs of the top-level encodings
//   work  : descriptor tables, scan partials and materialised intermediates
#pragma once
#include <stdint.h>

namespace nbl {

// How a consumer reads element i of a column produced by a child encoding.
// Leaves of the cascade (FixedBitWidth, numeric Trivial, Constant, bool
// Trivial) are never materialised for their parent: the parent's kernel reads
// the encoded payload through this accessor, which fuses the two levels.
enum ColKind : uint8_t {
  COL_NONE = 0,
  COL_MAT_WORK = 1,   // element array in `work` (aligned to 16 bytes)
  COL_MAT_OUT = 2,    // element array in `out`
  COL_FBW = 3,        // bit-packed payload in `arena`: (bits >> .. & mask) + base
  COL_RAW = 4,        // little-endian elements in `arena`, any alignment
  COL_CONST = 5,      // every element == base
  COL_BITS_WORK = 6,  // bool column as bitmap words (uint64, LSB first) in `work`
  COL_BITS_RAW = 7,   // bool column as bitmap bytes in `arena`, any alignment
};

struct ColRef {
  uint64_t off;    // byte offset into the buffer selected by `kind`
  uint64_t base;   // FBW baseline / CONST value (bool: 0 or 1)
  uint32_t rows;   // number of elements
  uint8_t kind;    // ColKind
  uint8_t width;   // FBW bit width (0..64)
  uint8_t es;      // bytes per element (1, 2, 4, 8); bool bitmaps: 0
  uint8_t pad;
};

// Kernel ids (also the index into the launch-name table).
enum KernelId : uint32_t {
  K_MATERIALIZE = 0,   // root leaf -> out               (src = c0)
  K_BITS_TO_BYTES,     // bool bitmap -> one byte per row (src = c0)
  K_REDUCE_POPC,       // scanned rank partials of a bitmap (src = sel, invert flag): a chained scan
  K_RLE_ENDS,          // inclusive run ends: a chained scan
  K_RLE_EXPAND,        // runs -> rows
  K_RLE_EXPAND_BOOL,   // bool runs -> bitmap
  K_SPARSE_BITS,       // sorted positions -> bitmap
  K_SELECT_SCATTER,    // Nullable / MainlyConstant
  K_DICT_GATHER,       // alphabet[indices[i]]
  K_VARINT_DECODE,
  K_STRING_VIEWS,      // lengths -> {offset,len}
  K_COPY_BYTES,        // string blobs arena -> out
  K_SELECT_BITS,       // Nullable<bool>: scatter a bitmap by a bitmap
  K_MATERIALIZE_FBW,   // root FixedBitWidth, window-per-vector fast path
  K_MATERIALIZE_RAW,   // root Trivial / byte-aligned FixedBitWidth: copy (+ baseline)
  K_FRAME_UNPACK,      // BlockBitPacking blocks / FOR frames: per-segment baseline + width
  K_SELECT_SCATTER_WAVE,  // Nullable / MainlyConstant of 4- / 8-byte rows with few values per tile: one wave per tile
  K_DICT_LDS,             // Dictionary of 8-byte rows (numbers, string views) whose alphabet fits kDictLdsEntries but not
                          // the fused launch's stage, alone or gathered inside a rank scatter (NF_DICT_VALUES): the
                          // alphabet sits in LDS and one workgroup runs kDictLdsTiles tiles against it
  K_DICT_STR,             // Dictionary of strings no longer than kDictStrMaxLen bytes with up to kDictStrEntries entries,
                          // alone or inside a rank scatter: the alphabet's views sit in LDS in compact form (a 16-bit
                          // offset per entry, relative to a base per 256 entries: NodeDesc::aux3 names the table that
                          // the alphabet's K_STRING_VIEWS step wrote) and one workgroup runs kDictStrTiles tiles
  K_COUNT,
  // Launch ids of the fused launches: one launch runs every step of a cascade
  // level, each tile naming its own operation (TileRef::tile, high byte).
  K_LEVEL = K_COUNT,   // any level but the last
  K_LEVEL_LAST,        // the last level (the one that writes most of the output)
  K_LAUNCH_COUNT
};

// Operation of a tile = kernel family + element size, packed into the high
// byte of TileRef::tile; the low 24 bits are the tile number within the node.
constexpr uint32_t kOpShift = 24;
constexpr uint32_t kTileMask = (1u << kOpShift) - 1u;
constexpr uint32_t es_slot(uint32_t es) { return es == 8 ? 3u : es == 4 ? 2u : es == 2 ? 1u : 0u; }
constexpr uint32_t make_op(uint32_t kernel, uint32_t es) { return kernel * 4u + es_slot(es); }
// Kernel families whose code depends on the element size; the others always use slot 0.
constexpr bool kernel_has_es(uint32_t k) {
  return k == K_MATERIALIZE || k == K_MATERIALIZE_FBW || k == K_MATERIALIZE_RAW || k == K_RLE_EXPAND ||
         k == K_SELECT_SCATTER || k == K_DICT_GATHER || k == K_VARINT_DECODE || k == K_FRAME_UNPACK ||
         k == K_SELECT_SCATTER_WAVE || k == K_DICT_LDS || k == K_DICT_STR;
}
// Operations the fused level launch carries.  The block encodings are rare
// (experimental in the reference) and their body would cost the fused kernel
// registers it does not have to spare, so they keep a launch of their own.
// The one-wave rank scatter is a launch of its own by construction: 64-thread
// workgroups with a quarter of the LDS, so that a CU holds 32 of its tiles.
// So are the LDS-alphabet dictionary kernels: 40 KiB of LDS per workgroup, or most of a CU's for compact string views.
constexpr bool kernel_is_fusable(uint32_t k) {
  return k != K_FRAME_UNPACK && k != K_SELECT_SCATTER_WAVE && k != K_DICT_LDS && k != K_DICT_STR;
}
constexpr uint32_t op_of(uint32_t kernel, uint32_t es) { return make_op(kernel, kernel_has_es(kernel) ? es : 1u); }

enum NodeFlags : uint32_t {
  NF_INVERT_SEL = 1u << 0,   // select where the bit is 0 (MainlyConstant isCommon)
  NF_WRITE_NULLS = 1u << 1,  // also emit the selection bitmap to out (Nullable root)
  NF_SPARSE_VALUE = 1u << 2, // SparseBool: listed positions are 1
  NF_FIRST_TRUE = 1u << 3,   // RLE<bool>: first run is true
  NF_DICT_VALUES = 1u << 5,  // K_SELECT_SCATTER: c1 holds dictionary indices (uint32), c0 the alphabet; gather inline
  NF_RLE_VALUES = 1u << 7,     // K_SELECT_SCATTER: the values child is an RLE expanded here as well: c0 = its run ends
                               // (uint32 array in work, c0.rows = its row count; first_run[] behind it), c1 = its run
                               // values, total_idx2 = its slot in the node-totals array
  NF_NESTED_SELECT = 1u << 6,  // K_SELECT_SCATTER: the values child is a MainlyConstant resolved here as well: c0 = its
                               // isCommon bitmap (other values where the bit is 0), c1 = its other values, aux3 = its
                               // common value, part_base2 / total_idx2 = its scanned rank partials and total
  NF_EMIT_PARTS = 1u << 8,   // K_SPARSE_BITS: the bitmap is the selection of its parent; also write the parent's rank
                             // partials, already scanned, and its total (part_base / n_parts / total_idx are the
                             // parent's, aux2 = rows per parent tile; NF_INVERT_SEL: count the zero bits)
  NF_SPARSE_DENSE = 1u << 9, // K_SPARSE_BITS: tiles of kBitTileRows rows instead of kSparseTileRows
  NF_BOOL_BYTES = 1u << 10,  // K_RLE_EXPAND_BOOL: the node is a chunk's root: self = its values in out, one byte per row
  NF_WHOLE_NODE = 1u << 11,  // K_SPARSE_BITS / K_REDUCE_POPC: one workgroup holds the whole node (<= kWholeTileRows rows):
                             // no search for a tile's range of positions, no ticket and no look-back
  NF_COMPACT_VIEWS = 1u << 12,  // K_STRING_VIEWS: also write the views in compact form (dict_str_table_bytes(rows) bytes at
                                // work + aux_off) for a K_DICT_STR gather; every length is at most kDictStrMaxLen
  NF_FRAME_BITS = 1u << 4,   // K_FRAME_UNPACK: FOR (u64 bit offsets, no raw marker) instead of
                             // BlockBitPacking (u32 byte offsets, width 255 = raw block)
};

// Status bits OR-ed into work[status_off] by kernels that meet inconsistent data.
enum StatusBits : uint32_t {
  ST_RLE_SUM_MISMATCH = 1u << 0,   // run lengths do not add up to rowCount
  ST_DICT_INDEX_RANGE = 1u << 1,   // dictionary index >= alphabet size
  ST_SELECT_COUNT = 1u << 2,       // set bits != rows of the values child
  ST_VARINT_COUNT = 1u << 3,       // fewer varints than rowCount
  ST_STRING_BYTES = 1u << 4,       // string lengths exceed the blob
  ST_FRAME_RANGE = 1u << 5,        // block / frame offset or bit width outside the packed payload
  ST_SCAN_STALL = 1u << 6,         // a chained scan gave up waiting for a tile in front of it (never expected)
};

struct NodeDesc {
  ColRef self;        // where the node's own output goes
  ColRef c0;          // first input (lengths / indices / source of a root leaf ...)
  ColRef c1;          // second input (run values / alphabet / selected values ...)
  ColRef sel;         // bitmap input (nulls / isCommon)
  uint64_t aux;       // fill value (Nullable: 0, MainlyConstant: common value);
                      // K_COPY_BYTES: byte count; K_STRING_VIEWS: blob base offset;
                      // K_FRAME_UNPACK: bytes of the packed payload
  uint64_t aux_off;   // second array: run ends (work) / nulls bitmap (out) / copy src (arena) /
                      // K_FRAME_UNPACK: packed payload (arena)
  uint64_t aux2;      // K_STRING_VIEWS: blob bytes; varint: payload byte count;
                      // select kernels: rows per tile of the popcount partials;
                      // K_FRAME_UNPACK: rows per segment | rows of the first segment << 32
  uint32_t rows;      // rows of this node
  uint32_t flags;     // NodeFlags
  uint32_t part_base; // first slot of this node in the tile-partials array
  uint32_t n_parts;   // number of scan tiles
  uint32_t total_idx; // slot in the node-totals array: [total, flag word, ticket counter of the chained scan, spare]
  uint32_t look_base; // first look-back word of the node's chained scan (KArgs::look), one per scan tile
  uint64_t aux3;       // NF_NESTED_SELECT: fill value of the inner selection (its common value);
                       // K_DICT_STR: the alphabet's compact table (work)
  uint32_t part_base2; // NF_NESTED_SELECT: first slot of the inner node's rank partials
  uint32_t total_idx2; // NF_NESTED_SELECT: the inner node's slot in the node-totals array
};

struct TileRef {
  uint32_t node;   // index into the node table
  uint32_t tile;   // make_op(kernel, es) << kOpShift | tile number within that node
};

// Addresses handed to every kernel.
struct KArgs {
  const uint8_t* arena;
  uint8_t* out;
  uint8_t* work;
  const NodeDesc* nodes;   // in work
  const TileRef* tiles;    // in work, already offset to this launch's first tile
  const uint32_t* tile_aux;  // per tile, parallel to `tiles`: for the rank-scatter tiles the slot of
                             // their partial in `parts`, known without reading the node (0 otherwise)
  uint32_t* parts;         // tile partials (work)
  uint32_t* totals;        // per-node totals (work)
  uint32_t* status;        // status word (work)
  uint64_t* look;          // look-back words of the chained scans (work)
  uint32_t n_tiles;
  uint32_t gen;            // number of this decode (1 .. 2^30 - 1): look-back words of any other number are stale
};

// Tile geometry: every row-producing kernel writes 16 KiB of output per
// workgroup of 256 threads (four 16-byte stores per thread).
constexpr uint32_t kBlock = 256;
constexpr uint32_t kTileBytes = 16384;
constexpr uint32_t kScanTile = 2048;        // elements per tile of the u32 scans
constexpr uint32_t kBitTileWords = 256;     // 64-bit words per bitmap tile (16384 rows)
constexpr uint32_t kBitTileRows = kBitTileWords * 64;
constexpr uint32_t kPopcWordsPerThread = 4;  // K_REDUCE_POPC: bitmap words per thread
constexpr uint32_t kPopcTileRows = kBitTileRows * kPopcWordsPerThread;
constexpr uint32_t kSparseTileWords = 1024;  // words per tile of the positions -> bitmap step (65536 rows)
constexpr uint32_t kSparseTileRows = kSparseTileWords * 64;
constexpr uint32_t kWholeTileWords = 2048;   // words of a bitmap that one workgroup builds or counts alone (NF_WHOLE_NODE)
constexpr uint32_t kWholeTileRows = kWholeTileWords * 64;
constexpr uint32_t kVarintTile = 4096;      // encoded bytes per varint tile
// Rows per tile of the rank-scatter (Nullable / MainlyConstant) of `es`-byte elements.
constexpr uint32_t select_tile_rows(uint32_t es) { return kTileBytes / es; }
// The one-wave rank scatter (K_SELECT_SCATTER_WAVE): 64 threads per tile and the
// bytes of staged values a tile may need (a tile that needs more reads them
// from memory row by row; the planner picks the kernel by the column's average).
constexpr uint32_t kWaveBlock = 64;
// K_DICT_LDS: alphabet entries (8 bytes each) the kernel keeps in LDS, and the
// 16 KiB tiles one workgroup decodes against them (the 32 KiB copy is paid once).
constexpr uint32_t kDictLdsEntries = 4096;
constexpr uint32_t kDictLdsTiles = 8;
constexpr uint32_t kWaveStageBytes = 2432;
// K_DICT_STR: string views {offset, length} whose strings lie back to back, as
// K_STRING_VIEWS writes them, in compact form: per block of 256 entries the
// offset of its first entry (uint32), then per entry its offset relative to
// that (uint16) with one more slot per block for the end of its last entry:
// entry i sits in slot i + i / 256, its length is the next slot minus its own.
// Lengths of at most 255 bytes keep a block's span below 65,536.
constexpr uint32_t kDictStrEntries = 65536;
constexpr uint32_t kDictStrMaxLen = 255;
constexpr uint32_t kDictStrTiles = 32;   // 16 KiB tiles one workgroup decodes against its copy of the table
constexpr uint32_t dict_str_blocks(uint32_t entries) { return (entries + 255u) >> 8; }
constexpr uint32_t dict_str_slots_off(uint32_t entries) { return (dict_str_blocks(entries) * 4u + 15u) & ~15u; }
constexpr uint32_t dict_str_table_bytes(uint32_t entries) {
  return dict_str_slots_off(entries) + (((entries + dict_str_blocks(entries)) * 2u + 4u + 15u) & ~15u);
}

}  // namespace nbl

/*
 * nimble_amd.h -- C ABI of the MI355X-native decoder for Nimble's cascading
 * stream encodings (Trivial, FixedBitWidth, RLE, Dictionary, Nullable,
 * SparseBool, Constant, MainlyConstant, Varint, and the block encodings
 * BlockBitPacking and FOR).
 *
 * The library replaces, for a *batch* of encoded chunks, what the reference
 * does one chunk at a time on the CPU:
 *
 *   EncodingFactory::create(pool, chunkBytes, stringBufferFactory)
 *       dwio/nimble/encodings/common/EncodingFactory.h:40-51,
 *       dwio/nimble/encodings/common/EncodingFactory.cpp:62-173
 *   followed by Encoding::materialize / materializeNullable over all rows
 *       dwio/nimble/encodings/common/Encoding.h:231-260
 *
 * The reference has no FFI; a maintainer binds these entry points from a
 * subclass of nimble::EncodingFactory / an implementation of nimble::Decoder
 * (dwio/nimble/velox/Decoder.h:27-56); nimble_amd/host/ holds both classes.
 *
 * Conventions: plain pointers and sizes only; every function returns an
 * nbl_status (0 = ok); nbl_last_error() gives the message of the calling
 * thread's last failure.  Device pointers are ordinary HIP device addresses,
 * `stream` is a hipStream_t passed as void*.  Nothing here allocates device
 * memory: the caller owns the arena (encoded bytes), output and work buffers,
 * sized from nbl_plan_info.
 */
#ifndef NIMBLE_AMD_H_
#define NIMBLE_AMD_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int nbl_status;
enum {
  NBL_OK = 0,
  /* Malformed bytes: what the reference reports through NIMBLE_CHECK /
   * NIMBLE_UNREACHABLE (dwio/nimble/common/Exceptions.h:111-155). */
  NBL_ERR_CORRUPT = 1,
  /* Well-formed Nimble that this library does not decode on the GPU
   * (compressed payloads until inflated, the other experimental encodings): the caller delegates the
   * chunk to the reference's CPU EncodingFactory. */
  NBL_ERR_UNSUPPORTED = 2,
  NBL_ERR_HIP = 3,      /* a HIP runtime call failed */
  NBL_ERR_ARGUMENT = 4, /* bad argument from the caller */
  NBL_ERR_DEVICE_DATA = 5 /* a kernel found inconsistent payload data */
};

/* Mirrors the fields of Encoding::Options that affect decode
 * (dwio/nimble/encodings/common/Encoding.h:115-175). */
typedef struct nbl_options {
  uint32_t struct_size;            /* sizeof(nbl_options) */
  uint32_t use_varint_row_count;   /* Encoding::Options::useVarintRowCount */
  uint32_t flags;                  /* NBL_PLAN_* (no counterpart in the reference) */
  uint32_t wave_scatter_min_tiles; /* rank-scatter tiles (16 KiB of output each) of one cascade level from
                                    * which the one-wave kernel gets a launch of its own (fewer run inside
                                    * the level's fused launch); 0 = the library's default (32768, about
                                    * 0.5 GB of scattered output).  Read when struct_size >= 16.
                                    * No counterpart in the reference. */
  uint32_t lds_dictionary_min_tiles; /* tiles of 8-byte dictionaries with 2,049 .. 4,096 entries (short strings
                                      * under bit-packed indices: up to 65,536) in one cascade level from which
                                      * the kernel that keeps the alphabet in LDS gets a launch of its own (fewer
                                      * are gathered from memory inside the fused launch); 0 = the library's
                                      * default (4096).  Read when struct_size >= 20.  No counterpart in the
                                      * reference. */
  uint32_t plan_threads;             /* host threads that walk the chunk headers of one nbl_plan_batch call; 0 = the
                                      * library's choice (up to 16, one per 128 chunks; NBL_PLAN_THREADS in the
                                      * environment overrides it), 1 = the calling thread only.  The plan does not
                                      * depend on it.  Read when struct_size >= 24.  The reference builds its
                                      * Encoding objects one stream at a time (velox/VeloxReader.cpp:330-351). */
} nbl_options;

/* By default every cascade level of the batch is ONE kernel launch whose tiles
 * name their own operation.  With this flag each (level, operation) pair is a
 * launch of its own, so that a profiler attributes time to operations. */
#define NBL_PLAN_PER_KERNEL_LAUNCHES 1u
/* Also emit the alphabet (and the raw indices) of dictionary-encoded chunks,
 * so that a reader can hand out dictionary vectors without expanding them. */
#define NBL_PLAN_KEEP_DICTIONARY 2u
/* Keep every Nullable / MainlyConstant rank scatter on the four-wave tile
 * kernel; by default columns whose tiles stage few values take the one-wave
 * kernel (more tiles in flight per CU).  Same results; for tuning and tests. */
#define NBL_PLAN_FOUR_WAVE_SCATTER 4u
/* Gather every dictionary from memory; by default dictionaries of 8-byte rows
 * (64-bit numbers, string views) with 2,049 .. 4,096 entries run on a kernel
 * that keeps the alphabet in LDS, and dictionaries of strings no longer than
 * 255 bytes whose indices are bit-packed at up to 16 bits, with 2,049 ..
 * 65,536 entries, on one that keeps the views in LDS in compact form.  Same
 * results; for tuning and tests. */
#define NBL_PLAN_NO_LDS_DICTIONARY 8u

typedef struct nbl_plan nbl_plan;

/* Sizes the caller needs to allocate buffers for one planned batch. */
typedef struct nbl_plan_info {
  uint32_t n_chunks;
  uint32_t n_nodes;        /* encodings in all cascade trees */
  uint32_t n_launches;     /* kernel launches nbl_decode_batch_device issues */
  uint32_t max_depth;      /* deepest cascade */
  uint64_t arena_bytes;    /* device bytes for the encoded image (input span + slop) */
  uint64_t out_bytes;      /* device bytes for all decoded outputs */
  uint64_t work_bytes;     /* device bytes for descriptors + intermediates */
  uint64_t descriptor_bytes; /* leading part of work that nbl_plan_upload fills */
  uint64_t total_rows;     /* sum of top-level row counts */
  uint64_t encoded_bytes;  /* sum of chunk sizes (B_enc, slop included) */
  uint64_t decoded_bytes;  /* bytes of decoded output written (B_out) */
} nbl_plan_info;

/* Where one chunk's decoded output lives inside the output buffer. */
typedef struct nbl_chunk_layout {
  uint32_t rows;
  uint8_t encoding_type;   /* EncodingType of the top-level encoding */
  uint8_t data_type;       /* DataType */
  uint8_t nullable;        /* 1 when the top-level encoding is Nullable */
  uint8_t elem_bytes;      /* numeric: sizeof(physicalType); bool: 1; string: 8 */
  uint32_t non_null_rows;  /* rows of the non-null child (== rows when !nullable) */
  uint32_t reserved;
  /* numeric: physicalType[rows]; bool: one byte (0/1) per row, as
   * Encoding::materialize yields; string: {uint32 offset, uint32 length}[rows]
   * relative to string_offset.  Null rows hold physicalType() / an empty view. */
  uint64_t values_offset;
  uint64_t values_bytes;
  /* Nullable only: bitmap, bit i = 1 when row i is non-null (LSB first),
   * ceil(rows/64)*8 bytes, bits past `rows` zero. */
  uint64_t nulls_offset;
  uint64_t nulls_bytes;
  /* string only: the bytes the reference would have placed in memory obtained
   * from stringBufferFactory. */
  uint64_t string_offset;
  uint64_t string_bytes;
  /* NBL_PLAN_KEEP_DICTIONARY only, when the chunk is Dictionary or
   * Nullable{Dictionary} (Encoding::dictionaryEnabled(), Encoding.h:277-311):
   * the alphabet, physicalType[dict_size] (string: {offset,length} views like
   * the values), and -- for a top-level Dictionary -- the raw indices
   * uint32[rows] that Encoding::materializeIndices yields
   * (DictionaryEncoding.h:81-83).  has_dictionary is 0 otherwise. */
  uint32_t has_dictionary;
  uint32_t dict_size;
  uint64_t dict_alphabet_offset;
  uint64_t dict_indices_offset;   /* valid when dict_indices_rows != 0 */
  uint64_t dict_indices_rows;
} nbl_chunk_layout;

/*
 * Walks the encoding headers of `n_chunks` top-level encodings (no payload is
 * decoded) and builds the decode tree, the output layout and the launch list.
 * Chunk i occupies [base + offsets[i], base + offsets[i] + sizes[i]) in host
 * memory; the device arena must be a byte image of the span
 * [base, base + max(offsets[i] + sizes[i])), so chunk offsets carry over
 * unchanged (no alignment requirement).  `chunk` here is the payload of one
 * stream chunk, i.e. the bytes after the 5-byte chunk header
 * (dwio/nimble/common/ChunkHeader.h:28-40); use nbl_split_stream first.
 * Replaces the recursive constructor walk of EncodingFactory::create.
 */
/* The (offset, size) pairs are trusted to lie inside the caller's buffer: the
 * library is not told its length.  A pair whose sum overflows is refused
 * (NBL_ERR_ARGUMENT); every bound inside a chunk is checked against the
 * chunk's own size. */
nbl_status nbl_plan_batch(const uint8_t* base, const uint64_t* offsets,
                          const uint64_t* sizes, uint32_t n_chunks,
                          const nbl_options* options, nbl_plan** plan_out);

void nbl_plan_free(nbl_plan* plan);

nbl_status nbl_plan_get_info(const nbl_plan* plan, nbl_plan_info* info);

nbl_status nbl_plan_get_chunk_layout(const nbl_plan* plan, uint32_t chunk,
                                     nbl_chunk_layout* layout);

/*
 * Splits one stream into its chunks: u32 length | u8 CompressionType | bytes
 * (InMemoryChunkedStream::nextChunk, dwio/nimble/velox/ChunkedStream.cpp:36-72).
 * Writes up to `capacity` (offset, size) pairs of the chunk payloads relative
 * to `stream` and the total count to *n_chunks.  A compressed chunk yields
 * NBL_ERR_UNSUPPORTED, bad framing NBL_ERR_CORRUPT.
 */
nbl_status nbl_split_stream(const uint8_t* stream, uint64_t size,
                            uint64_t* offsets, uint64_t* sizes,
                            uint32_t capacity, uint32_t* n_chunks);

/* Same, but compressed chunks (CompressionType Zstd = 1, Lz4 = 3) are reported
 * in compression[i] instead of refused; nbl_uncompress inflates them. */
nbl_status nbl_split_stream_ex(const uint8_t* stream, uint64_t size,
                               uint64_t* offsets, uint64_t* sizes, uint8_t* compression,
                               uint32_t capacity, uint32_t* n_chunks);

/*
 * Host pre-pass for compressed payloads (Compression::uncompress,
 * dwio/nimble/compression/Compression.h:72-94; framing u32 uncompressed size +
 * codec frame, ZstdCompressor.cpp:79-99, Lz4Compressor.cpp:60-75).  The codecs
 * come from the system's libzstd.so.1 / liblz4.so.1, resolved at first use;
 * without them these return NBL_ERR_UNSUPPORTED.
 *
 * nbl_uncompressed_size / nbl_uncompress: one compressed blob (a whole chunk
 * as nbl_split_stream_ex reports it, or a leaf payload).
 * nbl_chunk_inflated_size: size of the chunk once every compressed leaf
 * payload (Trivial values / bitmap / string blob, FixedBitWidth bits) is
 * inflated, or 0 when nothing in it is compressed (plan it as it is).
 * nbl_chunk_inflate: writes that encoding -- byte for byte what the writer
 * emits with compression off -- into `out`; nbl_plan_batch then takes it.
 */
nbl_status nbl_uncompressed_size(uint8_t compression, const uint8_t* data, uint64_t size, uint64_t* out_size);
nbl_status nbl_uncompress(uint8_t compression, const uint8_t* data, uint64_t size, uint8_t* out, uint64_t capacity);
nbl_status nbl_chunk_inflated_size(const uint8_t* chunk, uint64_t size, const nbl_options* options, uint64_t* out_size);
nbl_status nbl_chunk_inflate(const uint8_t* chunk, uint64_t size, const nbl_options* options,
                             uint8_t* out, uint64_t capacity, uint64_t* out_size);

/* Copies the plan's descriptor tables to the front of `d_work`
 * (info.descriptor_bytes, asynchronous on `stream`).  Must precede
 * nbl_decode_batch_device on the same stream for a given d_work.
 * Lifetime: the copy reads host memory the plan owns, and it is asynchronous,
 * so the plan must stay alive until `stream` has passed the upload (and the
 * decodes that use it): synchronise the stream -- nbl_decode_status does --
 * before nbl_plan_free. */
nbl_status nbl_plan_upload(nbl_plan* plan, void* d_work, void* stream);

/*
 * Decodes the whole batch: every cascade level is one batched kernel launch
 * over all chunks.  Asynchronous on `stream`; no allocation, no host
 * synchronisation.  Replaces Encoding::materialize / materializeNullable of
 * every chunk.  `d_arena` holds the encoded image (info.arena_bytes readable),
 * `d_out` receives the outputs (info.out_bytes), `d_work` is scratch
 * (info.work_bytes) already initialised by nbl_plan_upload.  The three device
 * pointers must be 16-byte aligned (hipMalloc gives 256).
 */
nbl_status nbl_decode_batch_device(const nbl_plan* plan, const void* d_arena,
                                   void* d_out, void* d_work, void* stream);

/* Synchronises `stream` and reports whether any kernel flagged inconsistent
 * data (e.g. run lengths that do not sum to the row count, dictionary index
 * out of range): NBL_OK or NBL_ERR_DEVICE_DATA.  The flags live in `d_work`
 * and gather over every decode since nbl_plan_upload or the last call that
 * reported them (reporting clears them), so one call after a series of
 * decodes with the same plan covers the series. */
nbl_status nbl_decode_status(const nbl_plan* plan, const void* d_work, void* stream);

/*
 * Copies `bytes` of device memory to pinned host memory by a kernel, asynchronous
 * on `stream`, for the copy back of a decoded batch into the buffers the
 * caller's Encoding::materialize (Encoding.h:231) is served from.  A
 * hipMemcpyAsync does the same on a copy engine; which engine a stream gets is
 * the runtime's choice, and two streams can end up with the copy back of one
 * and the copy in of the other on the same engine, taking turns.  A kernel
 * leaves the engines to the copy in, so both directions of the link run at
 * once whatever else the process has copied.  `host_dst` must be memory the
 * device can address (hipHostMalloc / hipHostRegister); both pointers 16-byte
 * aligned.  No allocation, no synchronisation.
 */
nbl_status nbl_copy_to_host_async(void* host_dst, const void* dev_src, uint64_t bytes, void* stream);

/* One kernel launch of the plan (for profiling and roofline accounting). */
typedef struct nbl_launch_info {
  const char* kernel;      /* kernel as a profiler names it, e.g. "k_materialize_fbw<8>",
                            * "k_level<true>" (the fused launch of the last level) */
  uint32_t elem_bytes;     /* element-size specialisation (0 when not specialised) */
  uint32_t n_tiles;        /* workgroups */
  uint32_t level;          /* cascade step the launch belongs to */
  uint32_t reserved;
  /* Algorithmic bytes of the chunks whose final output this launch writes:
   * their encoded bytes plus their decoded bytes (B_enc + B_out).  Launches
   * that only produce intermediates carry 0, so the sum over all launches is
   * info.encoded_bytes + info.decoded_bytes. */
  uint64_t alg_bytes;
} nbl_launch_info;

nbl_status nbl_plan_get_launch(const nbl_plan* plan, uint32_t launch, nbl_launch_info* info);

/*
 * A 64-bit hash of everything a decode of the plan reads or reports: the
 * descriptor tables as they are uploaded, the launch list, every chunk layout
 * and the buffer sizes.  Two plans of the same bytes with equal fingerprints
 * decode alike; how a plan was made (nbl_options.plan_threads) must not show
 * in it.  No counterpart in the reference; for tests.
 */
nbl_status nbl_plan_fingerprint(const nbl_plan* plan, uint64_t* fingerprint);

/*
 * Per-launch timing with HIP events recorded on the decode stream (no
 * counterpart in the reference; measurement only).  Consecutive launches of
 * the same kernel form a group that shares its boundary events, so a step
 * records (groups + 1) events; a launch's duration is its group's span divided
 * by the group's launches.  nbl_decode_batch_profiled is
 * nbl_decode_batch_device plus those events for step `step` (< max_steps);
 * nbl_profile_collect synchronises the last event and writes, for every
 * launch, the mean duration in milliseconds over the steps recorded so far.
 */
typedef struct nbl_profile nbl_profile;
nbl_status nbl_profile_create(const nbl_plan* plan, uint32_t max_steps, nbl_profile** profile_out);
void nbl_profile_free(nbl_profile* profile);
nbl_status nbl_decode_batch_profiled(const nbl_plan* plan, nbl_profile* profile, uint32_t step,
                                     const void* d_arena, void* d_out, void* d_work, void* stream);
nbl_status nbl_profile_collect(nbl_profile* profile, float* mean_ms_per_launch, uint32_t n_launches);

/* Message of the calling thread's most recent failure (never NULL). */
const char* nbl_last_error(void);

/* Library / device introspection: returns the gfx arch string the kernels
 * were built for ("gfx950"). */
const char* nbl_build_arch(void);
/* Hash of the library's source files as the build saw them (nimble_amd/build.py
 * passes it in; "unknown" for a build that did not): lets a caller -- and the
 * GPU test suite -- tell a library built from other sources than the ones
 * beside it.  No counterpart in the reference. */
const char* nbl_build_source_id(void);

#ifdef __cplusplus
}
#endif
#endif /* NIMBLE_AMD_H_ */

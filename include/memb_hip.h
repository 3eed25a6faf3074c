/*
 * memb_hip.h -- C ABI of the MI355X (gfx950) batch-lookup path of memb.
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++ or torch
 * types, no exceptions. It replaces, for a whole batch at once, what the
 * reference does per word behind its plugin interface
 *     CompressedStorage::extract(word, float*)      reference src/compression_strategy.h:7-12
 * namely
 *     TrainedCompressedStorage::extract              reference src/trained_compression.cpp:113-140
 *       HuffmanTableDecoder::next / BitStreamReader::pull
 *                                                    reference src/huffman_table_decoder.h:102-118,
 *                                                    reference src/bit_stream_reader.h:16-31
 *     UniformCompressedStorage::extract              reference src/uniform_compression.cpp:54-77
 *     FullCompressedStorage::extract                 reference src/full_compression.cpp:37-47
 *     the zero fill of a missing word                reference src/reader.cpp:41-47
 * and the batch driver Reader::batchEmbeddingToBuffer (reference src/reader.cpp:59-86).
 *
 * The caller (memb::Reader in memb_amd/csrc/reader.cpp, or any other host
 * language through its FFI -- see INTEGRATION.md) hands over row ids; row id
 * MEMB_HIP_MISSING_ROW produces a zero row. Word -> row (the search in front of
 * every extract: reference src/trained_compression.cpp:115-125, flatbuffers LookupByKey at
 * src/uniform_compression.cpp:56 and src/full_compression.cpp:39) is either the caller's
 * business on the host, or this library's on the device: memb_hip_ctx_stage_words +
 * memb_hip_words_* + memb_hip_resolve_rows_device below (round 5).
 *
 * Every function returns 0 on success and a non-zero code on failure;
 * memb_hip_last_error() returns the message for the calling thread.
 * There is no CPU fallback in this library: without a usable HIP device every
 * decode entry point fails. (The C++ classes above it can decode on the host when
 * a caller asks them to -- see INTEGRATION.md section 4 -- and then never call in here.)
 */
#ifndef MEMB_HIP_H
#define MEMB_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MEMB_HIP_MISSING_ROW 0xFFFFFFFFu

#define MEMB_HIP_OK 0
#define MEMB_HIP_ERR_INVALID 1   /* bad argument or inconsistent storage description */
#define MEMB_HIP_ERR_DEVICE 2    /* HIP runtime error (no device, out of memory, launch failure) */
#define MEMB_HIP_UNSUPPORTED 3   /* not an error: this combination has no fused kernel, use the plain calls */

typedef struct memb_hip_ctx memb_hip_ctx;

/*
 * `trained` storage (reference src/flatbuffers/trained_compression.fbs:6-13), as
 * host pointers into the mapped file. Row r is the r-th word in sorted order;
 * its bitstream starts, byte aligned, at packed_values + value_offsets[r]
 * (reference src/trained_compression.cpp:126-131).
 */
typedef struct memb_hip_trained_desc {
    uint32_t dim;
    uint64_t n_rows;
    const uint8_t* packed_values;
    uint64_t packed_values_bytes;
    const uint32_t* value_offsets;   /* [n_rows] */
    const uint8_t* keys;             /* symbols by increasing code length (huffman_decoder.fbs:4) */
    uint32_t n_keys;
    const uint32_t* size_offsets;    /* size_offsets[k] = #symbols with length <= k (huffman_decoder.fbs:5) */
    uint32_t n_size_offsets;
    const float* centroids;          /* k-means codebook (kmeans.fbs:4) */
    uint32_t n_centroids;
    /*
     * Upper bound on the bits resolved by the first-level lookup table, the
     * counterpart of maxDirectDecodeBitLength (reference
     * src/trained_compression.h:11-16). 0 selects the library default. Results
     * do not depend on it; small values force the two-level path, which is how
     * the reference tests that branch (reference src/tests.cpp:76-88). A hint:
     * it is capped at 12 and raised as far as needed for the tables to fit into
     * on-chip memory (codes of up to 16 bits make narrow first levels expensive);
     * memb_hip_ctx_info.root_bits reports the width in use.
     */
    uint32_t max_direct_bits;
} memb_hip_trained_desc;

/* One row of `uniform` storage (uniform_compression.fbs:3-12). */
typedef struct memb_hip_uniform_row {
    const uint8_t* values;   /* n_values quantised weights, one byte each */
    uint32_t n_values;
    float min_value;
    float max_value;
} memb_hip_uniform_row;

typedef struct memb_hip_uniform_desc {
    uint32_t dim;
    uint64_t n_rows;
    const memb_hip_uniform_row* rows;   /* [n_rows], sorted-word order */
    uint8_t quantization_levels;        /* uniform_compression.fbs:16 */
} memb_hip_uniform_desc;

/* One row of `full` storage (full_compression.fbs:3-6). */
typedef struct memb_hip_full_row {
    const float* values;
    uint32_t n_values;
} memb_hip_full_row;

typedef struct memb_hip_full_desc {
    uint32_t dim;
    uint64_t n_rows;
    const memb_hip_full_row* rows;
} memb_hip_full_desc;

/*
 * Version of this interface: bumped whenever a struct of this header changes size or layout,
 * or an entry point changes meaning. 3 = round 3 (memb_hip_ctx_info gained struct_size;
 * memb_hip_ctx_set_option and the builder entry points were added). 4 = round 4 (memb_hip_ctx_info: the
 * three large_batch_* fields of the per-context kernel timing are gone with it, tiles_per_wavefront
 * takes their place; options nt_loads, blocks_per_cu, autotune, pipeline, grid_policy no longer exist;
 * memb_hip_encoder_rows was added). 5 = round 5: word -> row on the device (memb_hip_ctx_stage_words, memb_hip_words_*,
 * memb_hip_resolve_rows_device, memb_hip_resolve_packed_device), several batches in one launch
 * (memb_hip_decode_batches_device); memb_hip_ctx_info is back on its ABI-3 offsets -- two reserved dwords follow
 * tiles_per_wavefront, where ABI 3 had the rest of its large_batch_* fields -- and grew at the END only (word_index_*);
 * memb_hip_ctx_get_info refuses the struct_size of the ABI-4 declaration, whose union_kernel sat 8 bytes lower.
 */
#define MEMB_HIP_ABI_VERSION 5
int memb_hip_abi_version(void);

/*
 * Facts about a context, for reporting (bench.py) and tests. The caller sets struct_size to
 * sizeof(memb_hip_ctx_info) as ITS header declares it; the library fills at most that many
 * bytes, so a client built against an older (shorter) declaration is never written past its
 * struct, and writes back the number of bytes it filled.
 */
typedef struct memb_hip_ctx_info {
    uint32_t struct_size;
    int32_t device;
    uint32_t storage;            /* 1 full, 2 uniform, 3 trained (wire::Storage tags) */
    uint32_t dim;
    uint64_t n_rows;
    uint64_t device_bytes;       /* HBM held by the context */
    uint32_t root_bits;          /* trained: first-level table bits */
    uint32_t max_code_bits;      /* trained: longest Huffman code */
    uint32_t table_entries;      /* trained: entries of the device lookup table */
    uint32_t max_stream_bytes;   /* trained: longest per-word bitstream */
    uint32_t waves_per_block;    /* launch geometry chosen for the decode kernel */
    uint32_t lanes_per_word;     /* trained: lanes that decode one word side by side */
    uint32_t segment_symbols;    /* trained: symbols decoded by each of those lanes */
    uint32_t lds_bytes_per_block;
    char kernel[96];             /* the kernel a dense device-resident batch runs, spelled as rocprofv3 prints it */
    uint32_t row_layout;         /* trained: 2 = row records (fixed-size row regions, record in front of the stream),
                                    1 = compact streams + one 16-byte index record per row, 0 = compact streams + index arrays */
    uint32_t row_bytes;          /* trained, row records: bytes every row owns */
    uint32_t kernel_registers;   /* trained, persistent kernel: vector registers per lane as the runtime reports them */
    uint32_t register_waves_per_cu;   /* ... and the wavefronts per CU those registers allow (32 = no limit from registers) */
    uint64_t batch_words;        /* IN: the batch size `kernel` and the geometry fields are reported for (the kernel is chosen
                                    by batch size); 0 = a large batch */
    uint32_t tiles_per_wavefront;   /* trained, decode_trained: tiles a wavefront decodes one after the other behind one copy
                                       of table and codebook into LDS (1, or 2 for tables of 16 KiB and more) */
    uint32_t reserved_abi3[2];   /* zero (ABI 3 had two more dwords here; keeps union_kernel where ABI 3 clients read it) */
    char union_kernel[96];       /* the kernel the last memb_hip_decode_rows_union_device call with this context as its FIRST
                                    model launched ("" = none yet, or the call returned MEMB_HIP_UNSUPPORTED) */
    /* ---- appended in ABI 5 ---- */
    uint64_t word_index_bytes;   /* HBM held by the word -> row index (keys + hash table); 0 = not staged (part of device_bytes) */
    uint32_t word_index_slots;   /* slots of the hash table (a power of two >= 2 x n_rows) */
    uint32_t word_index_keys;    /* keys in the table (n_rows minus repeated keys, which resolve to their first row) */
} memb_hip_ctx_info;

int memb_hip_device_count(int* count);

/*
 * Stage a storage to HBM on `device` (copied once; the host memory may be
 * unmapped afterwards). One context per (Reader, device).
 */
int memb_hip_ctx_create_trained(memb_hip_ctx** ctx, int device, const memb_hip_trained_desc* desc);
int memb_hip_ctx_create_uniform(memb_hip_ctx** ctx, int device, const memb_hip_uniform_desc* desc);
int memb_hip_ctx_create_full(memb_hip_ctx** ctx, int device, const memb_hip_full_desc* desc);
void memb_hip_ctx_destroy(memb_hip_ctx* ctx);

int memb_hip_ctx_get_info(const memb_hip_ctx* ctx, memb_hip_ctx_info* info);

/*
 * Knobs of a live context for tests and measurements (results never depend on them; every default is what the library
 * ships with). Unknown names and values out of range return MEMB_HIP_ERR_INVALID and change nothing.
 *   "waves_per_block" 0 (default) = by rule: four wavefronts; for batches of more than 524 000 words on 256 CUs eight when
 *                     the rows come in key order and seven when they do not (see MEMB_HIP_ROWS_IN_RANDOM_ORDER); or 1 .. 16
 *   "persistent"      1 (default) = the kernel by batch size: decode_trained (one tile per wavefront at a time), except
 *                     decode_records_persistent from the batch that no longer fits the CUs at once up to four tiles per 16
 *                     wavefronts per CU (57 000 - 131 000 words on 256 CUs);
 *                     0 = decode_trained always, 2 = decode_records_persistent wherever the row layout allows
 *   "tiles_per_wave"  0 (default) = by rule (two for models whose tables take 16 KiB of LDS and more, and for the split
 *                     union; else one); K = a wavefront of decode_trained / decode_union_split decodes K tiles one after
 *                     the other behind one copy of the tables into LDS
 *   "fine_lanes"      0 (default) = a row-record model's finer segment index (about sixteen lanes per word instead of eight)
 *                     decodes the batches whose tiles under it are all resident at once (28 600 words on 256 CUs),
 *                     1 = never, 2 = always
 *   "union_split"     1 (default) = a union of two models staged as row records runs decode_union_split
 *                     (the wavefront's word slots divided between the models), 0 = never (option of the FIRST model's context)
 *   "union_fused"     1 (default) = memb_hip_decode_rows_union_device launches ONE kernel for its two to four models where they
 *                     can share one; 0 = it returns MEMB_HIP_UNSUPPORTED and the caller launches per model, as for models that
 *                     cannot (option of the FIRST model's context; tests compare the two paths)
 *   "host_expand"     1 (default) = centroid indices instead of fp32 rows over PCIe (host-buffer entry point), 0 = fp32 rows
 * Builds with -DMEMB_HIP_MEASURE (tools/perf/build_measure.py; never shipped) also accept "debug" and "lds_pad",
 * the measurement switches of hip_trained_kernels.h; the shipped library refuses them.
 * Not thread-safe against lookups running on the same context.
 */
int memb_hip_ctx_set_option(memb_hip_ctx* ctx, const char* name, uint64_t value);

/*
 * Batch lookup, host buffers (the reference's calling convention: caller-owned
 * output, fully overwritten, reference src/reader.cpp:41-57). Writes row i of
 * the batch to out[i * ld + col_off .. + dim); ld >= col_off + dim. Synchronous;
 * calls on one context are serialised.
 *
 * How the rows reach host memory (results do not depend on it):
 *   n <= 512        the kernel reads the ids from and writes the rows to one pinned,
 *                   device-mapped buffer; the rows are then copied to `out`;
 *   larger batches  results stream through a ring of pinned buffers that the copy
 *                   engine fills while a few pooled host threads (kept by the
 *                   context) move landed chunks to `out`. For trained storages
 *                   the rows cross PCIe as centroid indices (1 or 1/2 byte per
 *                   weight, written by the same decode kernel) and those threads
 *                   expand them with the file's centroids -- bit-identical values,
 *                   1/4 to 1/8 of the transfer.
 */
int memb_hip_decode_rows(
    memb_hip_ctx* ctx, const uint32_t* rows, size_t n, float* out, size_t ld, size_t col_off);

/*
 * Batch lookup, device buffers: `rows` and `out` are device pointers on the
 * context's device; the kernel is enqueued on `stream` (a hipStream_t; NULL is
 * HIP's default stream) and the call returns without waiting.
 */
int memb_hip_decode_rows_device(
    memb_hip_ctx* ctx, const uint32_t* rows, size_t n, float* out, size_t ld, size_t col_off, void* stream);

/*
 * Same with an epilogue, for merging several models into one matrix on the
 * device (reference python/memb/readers_union.py:18, numpy.mean over the readers):
 * with MEMB_HIP_ACCUMULATE the decoded rows are added (fp32) to what `out` holds,
 * and a non-zero `divisor` divides the result. R readers average as: first call
 * plain, calls 2..R with MEMB_HIP_ACCUMULATE, the last one also with divisor = R --
 * the same additions in the same order and the same single division as numpy.mean.
 */
#define MEMB_HIP_ACCUMULATE 1u
/*
 * An override, never a requirement (results do not depend on it): the rows of this batch come in no particular order
 * (token ids, shuffled keys). Batches of more than 524 000 words run blocks of eight wavefronts for rows in key order
 * (dumps: 1-5 % over the alternatives) and of seven for rows in no particular order (3-5 %). Which of the two a batch is,
 * the library finds out by itself: the kernel of every such batch looks at sixty-four pairs of neighbouring row ids and
 * leaves word for the context's NEXT such batch (the first one is taken for a key-order dump). A caller whose batches
 * alternate between the two can say so here for the ones that are not in key order.
 */
#define MEMB_HIP_ROWS_IN_RANDOM_ORDER 2u
int memb_hip_decode_rows_device_ex(
    memb_hip_ctx* ctx, const uint32_t* rows, size_t n, float* out, size_t ld, size_t col_off, void* stream,
    uint32_t flags, float divisor);

/*
 * ReadersUnion 'concatenate' (reference python/memb/readers_union.py:21-32) as ONE launch:
 * row i of the result is [model 0's vector of rows[0][i] | model 1's vector of rows[1][i] | ...],
 * model m's block starting at column col_offs[m]. Same result as one
 * memb_hip_decode_rows_device call per model, but the merged rows are written whole
 * instead of one column block per launch. Device pointers, enqueued on `stream`.
 * With MEMB_HIP_UNION_AVERAGE in `flags` the 'average' mode instead (readers_union.py:5-18):
 * row i = (vector 0 + vector 1 + ...) / count, added in model order in fp32 and divided once,
 * as numpy.mean does, written at column col_offs[0].
 * Returns MEMB_HIP_UNSUPPORTED (and does nothing) when the models cannot share a
 * kernel: other than 2 to 4 trained storages of equal dim and lane geometry on one
 * device, or an output that is not 16-byte aligned in every block
 * (memb_hip_last_error() names the condition).
 */
#define MEMB_HIP_UNION_AVERAGE 1u
int memb_hip_decode_rows_union_device(
    memb_hip_ctx* const* ctxs, const uint32_t* const* rows, const size_t* col_offs, size_t count, size_t n,
    float* out, size_t ld, void* stream, uint32_t flags);

/*
 * Several batches of one context in ONE launch (a serving loop's way to amortise launch gap, prologue and tail of
 * batches too small to fill the device: reference src/reader.cpp:59-86 is a per-call driver): batch k looks up
 * batches[k].n rows batches[k].rows into batches[k].out with its own ld / col_off; results are those of `count`
 * memb_hip_decode_rows_device calls. Device pointers, enqueued on `stream`, returns without waiting. Trained
 * storages run the tiles of all batches numbered through in one decode_trained grid (up to MEMB_HIP_MAX_BATCHES per
 * launch, more are split); uniform and full storages launch once per batch.
 */
#define MEMB_HIP_MAX_BATCHES 16
typedef struct memb_hip_batch {
    const uint32_t* rows;   /* device */
    size_t n;
    float* out;             /* device */
    size_t ld;
    size_t col_off;
} memb_hip_batch;
int memb_hip_decode_batches_device(memb_hip_ctx* ctx, const memb_hip_batch* batches, size_t count, void* stream);

/*
 * Word -> row on the device (replaces, for whole batches, the search in front of every extract:
 *     TrainedCompressedStorage::extract   lower_bound + strcmp    reference src/trained_compression.cpp:115-125
 *     Uniform / Full ...::extract         LookupByKey             reference src/uniform_compression.cpp:56, src/full_compression.cpp:39
 * and the list -> vector<string> copy of the binding, reference python/memb_bindings.cpp:54-63).
 *
 * memb_hip_ctx_stage_words copies the model's keys to HBM once and builds an open-addressing hash table over them
 * on the device: 16-byte slots {hash tag, row, key offset, key length}, FNV-1a 64 over the key's bytes, linear
 * probing, at most half full. n_words keys, NUL terminated, key r (= row r, sorted order) at
 * packed_words + word_offsets[r]; packed_words[packed_bytes - 1] must be NUL (the trained storage's own
 * packed_words / word_offsets arrays qualify as they are: trained_compression.fbs:7,9). n_words must equal the
 * context's row count. A key equal to its predecessor is left out, so a repeated key resolves to its FIRST row, as
 * lower_bound does. Idempotent: a second call returns MEMB_HIP_OK and changes nothing.
 *
 * A lookup hashes the query, probes, and CONFIRMS a tag match by comparing length and bytes with the key: the
 * answer is the one the reference's binary search gives for every file whose keys are sorted (what its writers
 * produce: src/trained_compression.cpp:73-79, CreateVectorOfSortedTables), misses included -- also the word that
 * sorts after every key, where the reference dereferences end() (src/trained_compression.cpp:125).
 */
int memb_hip_ctx_stage_words(
    memb_hip_ctx* ctx, const char* packed_words, uint64_t packed_bytes, const uint32_t* word_offsets, uint64_t n_words);

/*
 * A batch of query words for one device, in pinned host memory that the lookup kernel reads over PCIe as it goes (the
 * copy to the device and the lookup are one kernel: nothing is staged in HBM, the row ids come out there). One object
 * may serve several contexts of that device (a ReadersUnion resolves one batch against every reader).
 *
 * Layout (memb_hip_words_plan): the batch is cut into JOBS of job_words words (a power of two, a multiple of 64; the
 * last job may be shorter). Job j owns the bytes [j * job_bytes, (j + 1) * job_bytes) of `bytes` and the entries
 * [j * (job_words + 1), ...) of `offsets`: its words back to back from the start of its region, and per word the
 * position of its first byte RELATIVE TO `bytes`, followed by one more entry, the end of its last word. Jobs are
 * independent, so any number of caller threads can fill them side by side, each word's bytes touched once.
 *
 *   begin     sizes the pinned buffers for n words of about bytes_per_word bytes (0 = default) and fills in the plan;
 *             waits first for whatever lookup still reads the previous batch of this object.
 *   (the caller writes jobs; a job that would overflow job_bytes: begin again with a larger bytes_per_word)
 *   commit    all jobs are written: memb_hip_words_count() == n from here on.
 *   pack      begin + fill + commit for C strings: words[i] is exactly lengths[i] bytes long (no terminator needed; a
 *             NUL among them is a byte like any other and matches no key), or NUL terminated when lengths is NULL.
 *             (The reference compares with strcmp, so for IT a std::string ends at its first NUL: memb::WordBatch and
 *             the Python binding cut words there.) Pooled host threads of the object do the filling; the host strings
 *             are free again when the call returns.
 *   resolve   rows_dev[i] = row of word i, or MEMB_HIP_MISSING_ROW, for all words of a committed batch
 *             (memb_hip_resolve_rows_device) or for words [first_word, first_word + n_words) of a batch whose jobs
 *             covering them are written, committed or not (memb_hip_resolve_range_device; first_word a multiple of
 *             job_words: lookups of finished jobs overlap the filling of later ones). Enqueued on `stream`, returns
 *             without waiting. The rows never visit the host: hand rows_dev to memb_hip_decode_rows_device.
 * The pinned buffers must stay untouched until the lookups that read them have run (begin waits for them; destroy too).
 * Not thread-safe: one batch in the making per object.
 */
typedef struct memb_hip_words memb_hip_words;
typedef struct memb_hip_words_plan {
    uint8_t* bytes;       /* pinned host memory, jobs * job_bytes bytes */
    uint32_t* offsets;    /* pinned host memory, jobs * (job_words + 1) entries */
    size_t n;             /* words of the batch */
    size_t job_words;
    size_t jobs;          /* ceil(n / job_words), at least 1 */
    size_t job_bytes;     /* capacity of one job's region (a multiple of 16) */
} memb_hip_words_plan;
int memb_hip_words_create(memb_hip_words** words, int device);
void memb_hip_words_destroy(memb_hip_words* words);
int memb_hip_words_begin(memb_hip_words* batch, size_t n, size_t bytes_per_word, memb_hip_words_plan* plan);
int memb_hip_words_commit(memb_hip_words* batch);
int memb_hip_words_pack(memb_hip_words* batch, const char* const* words, const uint32_t* lengths, size_t n);
int memb_hip_words_count(const memb_hip_words* batch, size_t* n);
int memb_hip_resolve_rows_device(memb_hip_ctx* ctx, const memb_hip_words* batch, uint32_t* rows_dev, void* stream);
int memb_hip_resolve_range_device(
    memb_hip_ctx* ctx, const memb_hip_words* batch, size_t first_word, size_t n_words, uint32_t* rows_dev, void* stream);
/*
 * The same for `count` (1 .. 4) contexts of one device in ONE launch -- a ReadersUnion's readers: every word is fetched
 * and hashed once and probed in each model's table; rows_dev[m][i] = row of word i in model m. rows_dev[m] is the whole
 * batch's array of model m, as above.
 */
int memb_hip_resolve_range_union_device(
    memb_hip_ctx* const* ctxs, size_t count, const memb_hip_words* batch, size_t first_word, size_t n_words,
    uint32_t* const* rows_dev, void* stream);
/*
 * Words in, host rows out -- the reference's own calling convention (Reader::batchEmbeddingToBuffer, src/reader.cpp:59-86)
 * with both of its halves on the device: the words of a committed batch are looked up by resolve_words and decoded as
 * memb_hip_decode_rows decodes row ids (same staging, same host threads; the row ids come back over PCIe once, 4 bytes per
 * word, for the threads that zero the rows of unknown words). The context's keys must be staged. Synchronous. Returns
 * MEMB_HIP_UNSUPPORTED for a batch so large that the result is staged in slices (4 GiB of fp32 rows and more).
 */
int memb_hip_decode_words(memb_hip_ctx* ctx, const memb_hip_words* batch, float* out, size_t ld, size_t col_off);

/*
 * The same lookup for callers whose words are on the device already: word i = bytes_dev[offsets_dev[i] ..
 * offsets_dev[i + 1]) (n + 1 offsets, ascending; no NUL inside a word).
 */
int memb_hip_resolve_packed_device(
    memb_hip_ctx* ctx, const uint8_t* bytes_dev, const uint32_t* offsets_dev, size_t n, uint32_t* rows_dev, void* stream);

/* Wait for the context's own stream (used by memb_hip_decode_rows). */
int memb_hip_sync(memb_hip_ctx* ctx);

/*
 * Algorithmic bytes moved for a batch: per row the row id,
 * the row's index entry, its compressed payload and the fp32 row written;
 * a missing row counts the id and the zero row.
 */
int memb_hip_algorithmic_bytes(const memb_hip_ctx* ctx, const uint32_t* rows, size_t n, uint64_t* bytes);

const char* memb_hip_last_error(void);

/*
 * Write side (optional; a file written with or without it has the same bytes). It replaces, for whole
 * blocks of vectors, the per-scalar and per-word work of the reference's trained compressor
 *     TrainedCompressor::finalize                    reference src/trained_compression.cpp:40-71
 *       KMeansClusterizer::predict                   reference src/kmeans.cpp:66-80
 *       HuffmanEncoderBuilder::updateFrequencies     reference src/huffman_encoder.cpp:22-29
 *       HuffmanEncoder::encode / BitStream::push     reference src/huffman_encoder.cpp:88-97, src/bit_stream.h:18-34
 * The caller keeps what depends on the order of the data or is tiny: the k-means fit on the first
 * 10 000 words (src/kmeans.cpp:26-64) and the Huffman tree. Protocol:
 *   create(split points = mid-points of neighbouring centroids, src/kmeans.cpp:121-128)
 *   add_rows(...) any number of times, in insertion order  -> symbols kept in HBM, histogram updated
 *   counts()                                                -> build the canonical code on the host
 *   pack(codes, lengths)                                    -> per-word stream lengths; streams laid out back to
 *                                                              back in insertion order (trained_compression.cpp:65-71)
 *   fetch(buffer)                                           -> packed_values
 * A symbol is the number of split points that compare less than the scalar (std::lower_bound; 0 for a NaN).
 */
typedef struct memb_hip_encoder memb_hip_encoder;

int memb_hip_encoder_create(memb_hip_encoder** encoder, int device, uint32_t dim, const float* split_points, uint32_t n_split_points);
void memb_hip_encoder_destroy(memb_hip_encoder* encoder);
/* rows: host memory, n_rows x dim floats, row-major; consumed before the call returns */
int memb_hip_encoder_add_rows(memb_hip_encoder* encoder, const float* rows, size_t n_rows);
/* counts[256]: how often each symbol occurred so far */
int memb_hip_encoder_counts(memb_hip_encoder* encoder, uint64_t* counts);
/* rows taken so far (whole add_rows calls that succeeded): what pack will lay out; a caller that keeps
 * the words checks it against its own count before it writes a file */
int memb_hip_encoder_rows(memb_hip_encoder* encoder, uint64_t* n_rows);
/*
 * codes[256] / lengths[256]: the prefix code of every symbol (length 0 = symbol never occurs), at most 16 bits;
 * of a code value the low `length` bits are written (BitStream::push, reference src/bit_stream.h:29).
 * stream_bytes[rows added]: length of every word's byte-aligned stream; total_bytes: their sum.
 */
int memb_hip_encoder_pack(
    memb_hip_encoder* encoder, const uint16_t* codes, const uint8_t* lengths, uint32_t* stream_bytes, uint64_t* total_bytes);
/* the packed streams into host memory of at least total_bytes */
int memb_hip_encoder_fetch(memb_hip_encoder* encoder, uint8_t* packed, uint64_t capacity);

#ifdef __cplusplus
}
#endif

#endif /* MEMB_HIP_H */

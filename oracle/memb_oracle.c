/*
 * memb_oracle.c -- CPU restatement of the reference's batch-lookup path.
 *
 * TEST INFRASTRUCTURE ONLY. Nothing in the memb_amd package imports, links or
 * executes this file; it is the checker the HIP path is compared against in
 * tests/, in __graft_entry__.smoke() and in bench.py's cpu_baseline leg.
 *
 * Each function restates one piece of the reference (thousandvoices/memb) in
 * plain C and cites the lines it follows. Parity pinning:
 *   - HuffmanTableDecoder / BitStreamReader / canonical codes / BitStream are
 *     checked against the reference's own headers compiled in place
 *     (oracle/_ref, built by oracle/Makefile from /root/reference/src) and
 *     against golden vectors generated from them (tests/golden/).
 *   - bit packing is pinned by the reference's known-answer test
 *     (src/bit_stream_tests.cpp:31-59), files by its round-trip tests
 *     (src/tests.cpp:20-113).
 *   - The uniform expression and the word search cannot be compiled from the
 *     reference here (they need flatc-generated headers); they are restated
 *     from the source text and pinned only by the reference's own test
 *     vectors (src/tests.cpp:20-57).
 */
#include <pthread.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <fcntl.h>
#include <unistd.h>

#define EXPORT __attribute__((visibility("default")))

/* ------------------------------------------------------------------------
 * BitStreamReader -- reference src/bit_stream_reader.h:7-38
 * ---------------------------------------------------------------------- */

typedef struct {
    uint64_t accumulator;
    const uint8_t* data;
    const uint8_t* data_end;
    int extra_bits;
} bit_reader;

static void bit_reader_init(bit_reader* r, const uint8_t* data, const uint8_t* data_end)
{
    r->accumulator = 0;
    r->data = data;
    r->data_end = data_end;
    r->extra_bits = 0;
}

/* reference src/bit_stream_reader.h:16-31: big-endian 32-bit refills, zero fill past the end, result unmasked */
static uint64_t bit_reader_pull(bit_reader* r, size_t bits_count)
{
    r->extra_bits -= (int)bits_count;
    while (r->extra_bits < 0) {
        for (size_t i = 0; i < 4; ++i) {
            r->accumulator <<= 8;
            if (r->data < r->data_end) {
                r->accumulator += *r->data;
                ++r->data;
            }
        }
        r->extra_bits += 32;
    }
    return r->accumulator >> r->extra_bits;
}

/* ------------------------------------------------------------------------
 * Canonical prefix codes -- reference src/prefix_code.cpp:5-22
 * codes are indexed by key, as in the reference's unordered_map
 * ---------------------------------------------------------------------- */

typedef struct {
    uint16_t code;
    size_t bits_count;
} prefix_code;

static void create_canonical_prefix_codes(
    const uint8_t* keys, const size_t* lengths, size_t count, prefix_code* codebook /* [256] */)
{
    prefix_code current = {0, 0};
    for (size_t i = 0; i < count; ++i) {
        while (current.bits_count < lengths[i]) {
            ++current.bits_count;
            current.code = (uint16_t)(current.code << 1);
        }
        codebook[keys[i]] = current;
        ++current.code;
    }
}

EXPORT void memb_oracle_canonical_codes(
    const uint8_t* keys, const uint32_t* lengths, size_t count, uint16_t* code_by_key, uint32_t* bits_by_key)
{
    prefix_code codebook[256];
    size_t* wide = (size_t*)malloc(sizeof(size_t) * (count ? count : 1));
    memset(codebook, 0, sizeof(codebook));
    for (size_t i = 0; i < count; ++i) {
        wide[i] = lengths[i];
    }
    create_canonical_prefix_codes(keys, wide, count, codebook);
    for (size_t k = 0; k < 256; ++k) {
        code_by_key[k] = codebook[k].code;
        bits_by_key[k] = (uint32_t)codebook[k].bits_count;
    }
    free(wide);
}

/* ------------------------------------------------------------------------
 * BitStream (writer) -- reference src/bit_stream.h:11-46, used here only for
 * the reference's known-answer test (src/bit_stream_tests.cpp:31-59)
 * ---------------------------------------------------------------------- */

EXPORT size_t memb_oracle_bitstream_pack(
    const uint16_t* codes, const uint32_t* bits, size_t count, uint8_t* out, size_t capacity)
{
    size_t size = 0;
    size_t free_bits = 0;
    for (size_t i = 0; i < count; ++i) {
        size_t bits_count = bits[i];
        while (bits_count > 0) {
            if (free_bits == 0) {
                if (size >= capacity) {
                    return (size_t)-1;
                }
                out[size++] = 0;
                free_bits = 8;
            }
            size_t bits_to_take = bits_count < free_bits ? bits_count : free_bits;
            uint16_t sliced = (uint16_t)((codes[i] & ((1U << bits_count) - 1)) >> (bits_count - bits_to_take));
            uint16_t shifted = (uint16_t)(sliced << (free_bits - bits_to_take));
            out[size - 1] = (uint8_t)(out[size - 1] + shifted);
            bits_count -= bits_to_take;
            free_bits -= bits_to_take;
        }
    }
    return size;
}

/* ------------------------------------------------------------------------
 * HuffmanTableDecoder -- reference src/huffman_table_decoder.h:11-142
 * ---------------------------------------------------------------------- */

typedef struct {
    uint8_t key;
    uint8_t bits_count;
} direct_decode_data;

typedef struct {
    size_t offset;
    uint8_t max_bits_count;
} indirect_decode_data;

typedef struct oracle_decoder {
    size_t max_direct_bits;
    uint64_t table_bit_mask;
    direct_decode_data* decode_table;
    size_t decode_table_size;
    indirect_decode_data* indirect_offsets;
    size_t indirect_offsets_size;
    direct_decode_data* indirect_table;
    size_t indirect_table_size;
    uint8_t code_bits[256]; /* code length by key (reporting only) */
} oracle_decoder;

/* reference src/huffman_table_decoder.h:121-125 */
static uint16_t base_offset(const prefix_code* codes, uint8_t key, size_t max_direct_bits)
{
    prefix_code code = codes[key];
    return (uint16_t)(code.code >> (code.bits_count - max_direct_bits));
}

/* reference src/huffman_table_decoder.h:18-94 */
EXPORT oracle_decoder* memb_oracle_decoder_create(
    const uint8_t* keys, size_t key_count, const uint32_t* size_offsets, size_t size_offset_count, uint32_t max_direct_bits)
{
    oracle_decoder* d = (oracle_decoder*)calloc(1, sizeof(oracle_decoder));
    d->max_direct_bits = max_direct_bits;
    d->table_bit_mask = (1U << max_direct_bits) - 1;
    size_t decode_table_size = (size_t)1 << max_direct_bits;

    /* :28-37 code length of keys[i] = smallest k with i < size_offsets[k] */
    size_t* lengths = (size_t*)malloc(sizeof(size_t) * (key_count ? key_count : 1));
    size_t current_size = 0;
    for (size_t key_index = 0; key_index < key_count; ++key_index) {
        while (key_index >= size_offsets[current_size]) {
            ++current_size;
        }
        lengths[key_index] = current_size;
    }

    prefix_code codes[256];
    memset(codes, 0, sizeof(codes));
    create_canonical_prefix_codes(keys, lengths, key_count, codes);
    for (size_t key_index = 0; key_index < key_count; ++key_index) {
        d->code_bits[keys[key_index]] = (uint8_t)lengths[key_index];
    }

    /* :40-56 direct table: 2^(L - len) copies per symbol, canonical order */
    size_t direct_symbols = (size_offset_count > max_direct_bits) ? size_offsets[max_direct_bits] : key_count;
    d->decode_table = (direct_decode_data*)malloc(sizeof(direct_decode_data) * decode_table_size);
    for (size_t key_index = 0; key_index < direct_symbols; ++key_index) {
        uint8_t key = keys[key_index];
        size_t size = codes[key].bits_count;
        direct_decode_data entry = {key, (uint8_t)size};
        size_t repeats = (size_t)1 << (max_direct_bits - size);
        for (size_t repeat = 0; repeat < repeats && d->decode_table_size < decode_table_size; ++repeat) {
            d->decode_table[d->decode_table_size++] = entry;
        }
    }

    /* :58-93 indirect tables for codes longer than L, grouped by their first L bits */
    if (direct_symbols < key_count) {
        uint8_t min_indirect_key = keys[direct_symbols];
        uint16_t min_indirect_offset = base_offset(codes, min_indirect_key, max_direct_bits);

        size_t groups = decode_table_size - min_indirect_offset;
        size_t* max_bits = (size_t*)calloc(groups ? groups : 1, sizeof(size_t));
        size_t indirect_capacity = 0;
        for (size_t key_index = direct_symbols; key_index < key_count; ++key_index) {
            uint8_t key = keys[key_index];
            size_t group = (size_t)base_offset(codes, key, max_direct_bits) - min_indirect_offset;
            if (codes[key].bits_count > max_bits[group]) {
                max_bits[group] = codes[key].bits_count;
            }
        }
        for (size_t key_index = direct_symbols; key_index < key_count; ++key_index) {
            uint8_t key = keys[key_index];
            size_t group = (size_t)base_offset(codes, key, max_direct_bits) - min_indirect_offset;
            indirect_capacity += (size_t)1 << (max_bits[group] - codes[key].bits_count);
        }
        d->indirect_offsets = (indirect_decode_data*)malloc(sizeof(indirect_decode_data) * (groups ? groups : 1));
        d->indirect_table = (direct_decode_data*)malloc(sizeof(direct_decode_data) * (indirect_capacity ? indirect_capacity : 1));

        uint16_t previous_base_offset = UINT16_MAX;
        for (size_t key_index = direct_symbols; key_index < key_count; ++key_index) {
            uint8_t key = keys[key_index];
            prefix_code current_code = codes[key];
            uint16_t current_base_offset = base_offset(codes, key, max_direct_bits);
            size_t current_max_bits = max_bits[current_base_offset - min_indirect_offset];
            if (current_base_offset != previous_base_offset) {
                previous_base_offset = current_base_offset;
                indirect_decode_data group = {d->indirect_table_size, (uint8_t)(current_max_bits - max_direct_bits)};
                d->indirect_offsets[d->indirect_offsets_size++] = group;
            }
            direct_decode_data entry = {key, (uint8_t)(current_code.bits_count - max_direct_bits)};
            for (size_t repeat = 0; repeat < ((size_t)1 << (current_max_bits - current_code.bits_count)); ++repeat) {
                d->indirect_table[d->indirect_table_size++] = entry;
            }
        }
        free(max_bits);
    }
    free(lengths);
    return d;
}

EXPORT void memb_oracle_decoder_destroy(oracle_decoder* d)
{
    if (d) {
        free(d->decode_table);
        free(d->indirect_offsets);
        free(d->indirect_table);
        free(d);
    }
}

EXPORT void memb_oracle_decoder_sizes(const oracle_decoder* d, uint64_t* sizes /* [3] */)
{
    sizes[0] = d->decode_table_size;
    sizes[1] = d->indirect_offsets_size;
    sizes[2] = d->indirect_table_size;
}

typedef struct {
    bit_reader reader;
    size_t bits_to_pull;
} decode_state;

/* reference src/huffman_table_decoder.h:96-100 */
static decode_state decoder_decode(const oracle_decoder* d, const uint8_t* source, size_t source_size)
{
    decode_state state;
    bit_reader_init(&state.reader, source, source + source_size);
    state.bits_to_pull = d->max_direct_bits;
    return state;
}

/* reference src/huffman_table_decoder.h:102-118 */
static uint8_t decoder_next(const oracle_decoder* d, decode_state* state)
{
    size_t offset = (size_t)(bit_reader_pull(&state->reader, state->bits_to_pull) & d->table_bit_mask);
    if (offset < d->decode_table_size) {
        direct_decode_data entry = d->decode_table[offset];
        state->bits_to_pull = entry.bits_count;
        return entry.key;
    } else {
        indirect_decode_data indirect = d->indirect_offsets[offset - d->decode_table_size];
        size_t bit_mask = (1U << indirect.max_bits_count) - 1;
        size_t indirect_key = (size_t)(bit_reader_pull(&state->reader, indirect.max_bits_count) & bit_mask);
        direct_decode_data entry = d->indirect_table[indirect.offset + indirect_key];
        state->bits_to_pull = d->max_direct_bits - indirect.max_bits_count + entry.bits_count;
        return entry.key;
    }
}

EXPORT void memb_oracle_decode_symbols(
    const oracle_decoder* d, const uint8_t* source, size_t source_size, size_t count, uint8_t* out_keys)
{
    decode_state state = decoder_decode(d, source, source_size);
    for (size_t i = 0; i < count; ++i) {
        out_keys[i] = decoder_next(d, &state);
    }
}

/* ------------------------------------------------------------------------
 * Wire format -- FlatBuffers accessors for the six tables
 * (reference src/flatbuffers/, all six .fbs files; field slot = 4 + 2 * id)
 * ---------------------------------------------------------------------- */

typedef struct {
    const uint8_t* data;
    size_t size;
} blob;

static uint32_t rd_u32(const blob* b, size_t pos)
{
    uint32_t v = 0;
    if (pos + 4 <= b->size) {
        memcpy(&v, b->data + pos, 4);
    }
    return v;
}

static uint16_t rd_u16(const blob* b, size_t pos)
{
    uint16_t v = 0;
    if (pos + 2 <= b->size) {
        memcpy(&v, b->data + pos, 2);
    }
    return v;
}

/* position of field `id` of the table at `table`, 0 if absent */
static size_t fb_field(const blob* b, size_t table, size_t id)
{
    int32_t soffset = (int32_t)rd_u32(b, table);
    size_t vtable = (size_t)((int64_t)table - soffset);
    uint16_t vtable_bytes = rd_u16(b, vtable);
    size_t slot = 4 + 2 * id;
    if (slot + 2 > vtable_bytes) {
        return 0;
    }
    uint16_t offset = rd_u16(b, vtable + slot);
    return offset ? table + offset : 0;
}

static size_t fb_indirect(const blob* b, size_t table, size_t id)
{
    size_t pos = fb_field(b, table, id);
    return pos ? pos + rd_u32(b, pos) : 0;
}

/* vector: returns element 0 position, writes the element count */
static size_t fb_vector(const blob* b, size_t table, size_t id, size_t* count)
{
    size_t pos = fb_indirect(b, table, id);
    if (!pos) {
        *count = 0;
        return 0;
    }
    *count = rd_u32(b, pos);
    return pos + 4;
}

enum { STORAGE_NONE = 0, STORAGE_FULL = 1, STORAGE_UNIFORM = 2, STORAGE_TRAINED = 3 };

#define DEFAULT_DECODE_TABLE_BIT_LENGTH 10 /* reference src/trained_compression.h:11 */
#define THREADED_DECODER_THRESHOLD 1024    /* reference src/reader.cpp:9 */

typedef struct oracle_reader {
    blob file;
    size_t num_threads;
    uint32_t dim;
    uint8_t storage_type;
    size_t storage; /* table position */

    /* trained */
    const uint32_t* word_offsets;
    size_t word_count;
    const uint32_t* value_offsets;
    const char* packed_words;
    const uint8_t* packed_values;
    size_t packed_values_size;
    oracle_decoder* decoder;
    const uint8_t* decoder_keys;
    size_t decoder_key_count;
    const uint32_t* decoder_size_offsets;
    size_t decoder_size_offset_count;
    float* centroids;
    size_t centroid_count;

    /* uniform / full */
    size_t nodes; /* position of element 0 of the vector of table offsets */
    size_t node_count;
    uint8_t quantization_levels;
} oracle_reader;

static void set_error(char* err, size_t err_size, const char* message)
{
    if (err && err_size) {
        snprintf(err, err_size, "%s", message);
    }
}

/* reference src/reader.cpp:113-120 */
static size_t adjusted_num_threads(size_t num_threads)
{
    if (num_threads > 0) {
        return num_threads;
    }
    long cores = sysconf(_SC_NPROCESSORS_ONLN);
    return cores > 2 ? (size_t)cores : 2;
}

/*
 * Reader constructors -- reference src/reader.cpp:13-29, :104-111;
 * TrainedCompressedStorage ctor -- reference src/trained_compression.cpp:103-111.
 * max_direct_bits = 0 selects DEFAULT_DECODE_TABLE_BIT_LENGTH; other values are
 * what the reference's test injects through a strategy subclass (src/tests.cpp:76-88).
 */
EXPORT oracle_reader* memb_oracle_open(
    const char* filename, size_t num_threads, uint32_t max_direct_bits, char* err, size_t err_size)
{
    int fd = open(filename, O_RDONLY);
    if (fd < 0) {
        set_error(err, err_size, "failed opening file");
        return NULL;
    }
    struct stat info;
    if (fstat(fd, &info) != 0) {
        close(fd);
        set_error(err, err_size, "failed opening file");
        return NULL;
    }
    oracle_reader* r = (oracle_reader*)calloc(1, sizeof(oracle_reader));
    r->num_threads = adjusted_num_threads(num_threads);
    r->file.size = (size_t)info.st_size;
    if (r->file.size) {
        void* mapping = mmap(NULL, r->file.size, PROT_READ, MAP_PRIVATE, fd, 0);
        if (mapping == MAP_FAILED) {
            close(fd);
            free(r);
            set_error(err, err_size, "failed mapping file");
            return NULL;
        }
        r->file.data = (const uint8_t*)mapping;
    }
    close(fd);

    /* getIndexChecked: size >= 8 and identifier "memb" at bytes 4..8 */
    if (r->file.size < 8 || memcmp(r->file.data + 4, "memb", 4) != 0) {
        if (r->file.data) {
            munmap((void*)r->file.data, r->file.size);
        }
        free(r);
        set_error(err, err_size, "File format verification failed");
        return NULL;
    }
    const blob* b = &r->file;
    size_t index = rd_u32(b, 0);
    size_t pos = fb_field(b, index, 0);
    r->storage_type = pos ? b->data[pos] : STORAGE_NONE;
    r->storage = fb_indirect(b, index, 1);
    pos = fb_field(b, index, 2);
    r->dim = pos ? rd_u32(b, pos) : 0;

    if (r->storage_type == STORAGE_TRAINED) {
        size_t count = 0;
        r->word_offsets = (const uint32_t*)(b->data + fb_vector(b, r->storage, 0, &r->word_count));
        r->value_offsets = (const uint32_t*)(b->data + fb_vector(b, r->storage, 1, &count));
        r->packed_words = (const char*)(b->data + fb_vector(b, r->storage, 2, &count));
        r->packed_values = b->data + fb_vector(b, r->storage, 3, &r->packed_values_size);
        /* HuffmanDecoder::load -- reference src/huffman_decoder.cpp:25-30 */
        size_t decoder = fb_indirect(b, r->storage, 4);
        size_t key_count = 0;
        size_t size_offset_count = 0;
        const uint8_t* keys = b->data + fb_vector(b, decoder, 0, &key_count);
        const uint32_t* size_offsets = (const uint32_t*)(b->data + fb_vector(b, decoder, 1, &size_offset_count));
        r->decoder_keys = keys;
        r->decoder_key_count = key_count;
        r->decoder_size_offsets = size_offsets;
        r->decoder_size_offset_count = size_offset_count;
        r->decoder = memb_oracle_decoder_create(
            keys, key_count, size_offsets, size_offset_count,
            max_direct_bits ? max_direct_bits : DEFAULT_DECODE_TABLE_BIT_LENGTH);
        /* KMeansClusterizer::load(...).centroids() -- reference src/kmeans.cpp:121-128 */
        size_t clusterizer = fb_indirect(b, r->storage, 5);
        const uint8_t* centroids = b->data + fb_vector(b, clusterizer, 0, &r->centroid_count);
        r->centroids = (float*)malloc(sizeof(float) * (r->centroid_count ? r->centroid_count : 1));
        memcpy(r->centroids, centroids, sizeof(float) * r->centroid_count);
    } else if (r->storage_type == STORAGE_UNIFORM) {
        r->nodes = fb_vector(b, r->storage, 0, &r->node_count);
        pos = fb_field(b, r->storage, 1);
        r->quantization_levels = pos ? b->data[pos] : 0;
    } else if (r->storage_type == STORAGE_FULL) {
        r->nodes = fb_vector(b, r->storage, 0, &r->node_count);
    } else {
        /* reference src/compression_strategy.cpp:36-39 */
        char message[96];
        snprintf(message, sizeof(message), "Storage strategy %u is not supported", (unsigned)r->storage_type);
        munmap((void*)r->file.data, r->file.size);
        free(r);
        set_error(err, err_size, message);
        return NULL;
    }
    return r;
}

EXPORT void memb_oracle_close(oracle_reader* r)
{
    if (r) {
        memb_oracle_decoder_destroy(r->decoder);
        free(r->centroids);
        if (r->file.data) {
            munmap((void*)r->file.data, r->file.size);
        }
        free(r);
    }
}

EXPORT uint32_t memb_oracle_dim(const oracle_reader* r) { return r->dim; }
EXPORT uint32_t memb_oracle_storage_type(const oracle_reader* r) { return r->storage_type; }

EXPORT size_t memb_oracle_size(const oracle_reader* r)
{
    return r->storage_type == STORAGE_TRAINED ? r->word_count : r->node_count;
}

static size_t node_table(const oracle_reader* r, size_t index)
{
    size_t pos = r->nodes + 4 * index;
    return pos + rd_u32(&r->file, pos);
}

static const char* node_word(const oracle_reader* r, size_t index)
{
    size_t count = 0;
    return (const char*)(r->file.data + fb_vector(&r->file, node_table(r, index), 0, &count));
}

/* keys() -- reference src/trained_compression.cpp:142-153, src/uniform_compression.cpp:79-89 */
EXPORT const char* memb_oracle_key(const oracle_reader* r, size_t index)
{
    if (r->storage_type == STORAGE_TRAINED) {
        return r->packed_words + r->word_offsets[index];
    }
    return node_word(r, index);
}

/*
 * Word -> row. trained: std::lower_bound with strcmp over word_offsets
 * (reference src/trained_compression.cpp:115-125); the reference dereferences
 * the result even at end(), which is an out-of-bounds read -- a miss here.
 * uniform/full: flatbuffers LookupByKey = binary search with strcmp over the
 * sorted vector of tables (call sites src/uniform_compression.cpp:56,
 * src/full_compression.cpp:39).
 */
EXPORT int memb_oracle_resolve(const oracle_reader* r, const char* word, uint32_t* row)
{
    size_t count = memb_oracle_size(r);
    size_t first = 0;
    size_t length = count;
    while (length > 0) {
        size_t half = length / 2;
        size_t middle = first + half;
        if (strcmp(memb_oracle_key(r, middle), word) < 0) {
            first = middle + 1;
            length -= half + 1;
        } else {
            length = half;
        }
    }
    if (first < count && strcmp(memb_oracle_key(r, first), word) == 0) {
        *row = (uint32_t)first;
        return 1;
    }
    return 0;
}

/*
 * The same search for a whole batch (tests of the device word search compare 2.2 M answers): words[i] NUL
 * terminated; rows[i] = the row, or 0xFFFFFFFF for a miss.
 */
EXPORT void memb_oracle_resolve_many(const oracle_reader* r, const char* const* words, size_t count, uint32_t* rows)
{
    for (size_t i = 0; i < count; ++i) {
        uint32_t row = 0;
        rows[i] = memb_oracle_resolve(r, words[i], &row) ? row : 0xFFFFFFFFu;
    }
}

/* decode of one present row */
static void extract_row(const oracle_reader* r, size_t row, float* destination)
{
    if (r->storage_type == STORAGE_TRAINED) {
        /* reference src/trained_compression.cpp:126-135 */
        size_t offset = r->value_offsets[row];
        decode_state state = decoder_decode(r->decoder, r->packed_values + offset, r->packed_values_size - offset);
        for (size_t i = 0; i < r->dim; ++i) {
            destination[i] = r->centroids[decoder_next(r->decoder, &state)];
        }
    } else if (r->storage_type == STORAGE_UNIFORM) {
        /* reference src/uniform_compression.cpp:58-72 */
        const blob* b = &r->file;
        size_t vector = fb_indirect(b, node_table(r, row), 1);
        size_t pos = fb_field(b, vector, 0);
        float min_value = 0.0f;
        float max_value = 0.0f;
        if (pos) {
            memcpy(&min_value, b->data + pos, 4);
        }
        pos = fb_field(b, vector, 1);
        if (pos) {
            memcpy(&max_value, b->data + pos, 4);
        }
        size_t count = 0;
        const uint8_t* values = b->data + fb_vector(b, vector, 2, &count);
        uint8_t quantization_levels = r->quantization_levels;
        for (size_t i = 0; i < count; ++i) {
            float float_value = (float)values[i];
            destination[i] = min_value + (max_value - min_value) * float_value / quantization_levels;
        }
    } else {
        /* reference src/full_compression.cpp:40-43 */
        size_t count = 0;
        const uint8_t* values = r->file.data + fb_vector(&r->file, node_table(r, row), 1, &count);
        memcpy(destination, values, count * sizeof(float));
    }
}

/* the single uniform expression, for known-answer checks */
EXPORT float memb_oracle_uniform_value(float min_value, float max_value, uint8_t value, uint8_t quantization_levels)
{
    float float_value = (float)value;
    return min_value + (max_value - min_value) * float_value / quantization_levels;
}

/* CompressedStorage::extract -- returns 0 when the word is absent */
EXPORT int memb_oracle_extract(const oracle_reader* r, const char* word, float* destination)
{
    uint32_t row = 0;
    if (!memb_oracle_resolve(r, word, &row)) {
        return 0;
    }
    extract_row(r, row, destination);
    return 1;
}

/* Reader::wordEmbeddingToBuffer -- reference src/reader.cpp:41-47 */
EXPORT void memb_oracle_word_embedding(const oracle_reader* r, const char* word, float* buffer)
{
    if (!memb_oracle_extract(r, word, buffer)) {
        for (size_t i = 0; i < r->dim; ++i) {
            buffer[i] = 0;
        }
    }
}

typedef struct {
    const oracle_reader* reader;
    const char* const* words;
    const uint32_t* rows;
    size_t count;
    float* buffer;
    size_t ld;
    size_t col_off;
} batch_job;

/* Reader::batchEmbeddingToBufferImpl -- reference src/reader.cpp:49-57 */
static void* batch_job_run(void* argument)
{
    batch_job* job = (batch_job*)argument;
    const oracle_reader* r = job->reader;
    for (size_t idx = 0; idx < job->count; ++idx) {
        float* destination = job->buffer + job->ld * idx + job->col_off;
        if (job->words) {
            memb_oracle_word_embedding(r, job->words[idx], destination);
        } else if (job->rows[idx] < memb_oracle_size(r)) {
            extract_row(r, job->rows[idx], destination);
        } else {
            for (size_t i = 0; i < r->dim; ++i) {
                destination[i] = 0;
            }
        }
    }
    return NULL;
}

/* Reader::batchEmbeddingToBuffer -- reference src/reader.cpp:59-86: serial below
 * 1024 words or with one thread, else ceil(n / T) words per job */
static void run_batch(
    const oracle_reader* r, const char* const* words, const uint32_t* rows, size_t count, float* buffer, size_t ld,
    size_t col_off, size_t num_threads)
{
    if (count < THREADED_DECODER_THRESHOLD || num_threads == 1) {
        batch_job job = {r, words, rows, count, buffer, ld, col_off};
        batch_job_run(&job);
        return;
    }
    size_t job_size = (count + num_threads - 1) / num_threads;
    size_t job_count = (count + job_size - 1) / job_size;
    batch_job* jobs = (batch_job*)calloc(job_count, sizeof(batch_job));
    pthread_t* threads = (pthread_t*)calloc(job_count, sizeof(pthread_t));
    size_t started = 0;
    for (size_t start = 0; start < count; start += job_size, ++started) {
        size_t end = start + job_size < count ? start + job_size : count;
        batch_job job = {r, words ? words + start : NULL, rows ? rows + start : NULL, end - start,
                         buffer + start * ld, ld, col_off};
        jobs[started] = job;
        pthread_create(&threads[started], NULL, batch_job_run, &jobs[started]);
    }
    for (size_t i = 0; i < started; ++i) {
        pthread_join(threads[i], NULL);
    }
    free(threads);
    free(jobs);
}

EXPORT void memb_oracle_batch_embedding(
    const oracle_reader* r, const char* const* words, size_t count, float* buffer)
{
    run_batch(r, words, NULL, count, buffer, r->dim, 0, r->num_threads);
}

/* decode only: rows already resolved (row >= size -> zero row); ld / col_off as in the C ABI */
EXPORT void memb_oracle_rows_embedding(
    const oracle_reader* r, const uint32_t* rows, size_t count, float* buffer, size_t ld, size_t col_off,
    size_t num_threads)
{
    run_batch(r, NULL, rows, count, buffer, ld, col_off, num_threads ? num_threads : r->num_threads);
}

/* bytes of the trained bitstream one row's decode consumes (for the roofline byte count) */
/* The arrays of a trained storage as the file holds them (pointers into the
 * mapping, valid while the reader is open), for checkers that run another
 * decoder over the same file (oracle/ref_driver.cpp). Returns 0 for other storages. */
typedef struct {
    const uint8_t* packed_values;
    uint64_t packed_values_size;
    const uint32_t* value_offsets;
    uint64_t word_count;
    const uint8_t* keys;
    uint64_t key_count;
    const uint32_t* size_offsets;
    uint64_t size_offset_count;
    const float* centroids;
    uint64_t centroid_count;
} memb_oracle_trained_arrays;

EXPORT int memb_oracle_trained_view(const oracle_reader* r, memb_oracle_trained_arrays* view)
{
    if (r->storage_type != STORAGE_TRAINED) {
        return 0;
    }
    view->packed_values = r->packed_values;
    view->packed_values_size = r->packed_values_size;
    view->value_offsets = r->value_offsets;
    view->word_count = r->word_count;
    view->keys = r->decoder_keys;
    view->key_count = r->decoder_key_count;
    view->size_offsets = r->decoder_size_offsets;
    view->size_offset_count = r->decoder_size_offset_count;
    view->centroids = r->centroids;
    view->centroid_count = r->centroid_count;
    return 1;
}

EXPORT uint32_t memb_oracle_stream_bytes(const oracle_reader* r, uint32_t row)
{
    if (r->storage_type != STORAGE_TRAINED || row >= r->word_count) {
        return 0;
    }
    size_t offset = r->value_offsets[row];
    decode_state state = decoder_decode(r->decoder, r->packed_values + offset, r->packed_values_size - offset);
    size_t consumed = 0;
    for (size_t i = 0; i < r->dim; ++i) {
        consumed += r->decoder->code_bits[decoder_next(r->decoder, &state)];
    }
    return (uint32_t)((consumed + 7) / 8);
}

// HIP (gfx950 / CDNA4) implementation of the memb batch-lookup path behind the
// C ABI in include/memb_hip.h.
//
// Kernels
//   decode_trained : canonical-Huffman bitstream decode + k-means codebook gather
//                    (reference src/trained_compression.cpp:129-135,
//                     src/huffman_table_decoder.h:102-118, src/bit_stream_reader.h:16-31)
//   dequant_uniform: min + (max - min) * v / levels, four IEEE fp32 operations
//                    (reference src/uniform_compression.cpp:64-72)
//   gather_full    : raw fp32 row copy (reference src/full_compression.cpp:37-47)
// A row id of MEMB_HIP_MISSING_ROW yields a zero row (reference src/reader.cpp:43-46).
//
// Work decomposition of decode_trained. A Huffman bitstream is serial, so the
// parallelism is across words and across SEGMENTS of a word: when a model is
// staged, one pass over all rows records the bit position at which every
// S-th symbol of each row starts (a side index, derived data: the file is
// unchanged). A wavefront then owns a tile of 64 / G consecutive batch entries
// and G = ceil(dim / S) lanes decode one word, each its own segment of S
// symbols. Lookup table and codebook sit in LDS; the tile's bitstreams are
// first copied into LDS with wide loads (one 16-byte piece per lane), decoded
// from there, the symbols are staged in LDS one byte each, and the tile is
// written out row contiguous, 16 bytes per lane, so every store instruction
// covers whole 16-byte-aligned runs of output rows. G lanes per word divide
// the LDS needed per lane in flight by G, which is what bounds occupancy.
#include <hip/hip_runtime.h>

#include "../../include/memb_hip.h"
#include "codec.h"
#include "wire.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

constexpr int WAVE = 64;
constexpr uint32_t MISSING = MEMB_HIP_MISSING_ROW;
constexpr uint32_t ZERO_KEY = 255;  // codebook slot that always holds 0.0f: at most 255 centroids exist
                                    // (reference src/trained_compression.cpp:29)

thread_local std::string g_lastError;

int fail(int code, const std::string& message)
{
    g_lastError = message;
    return code;
}

#define HIP_TRY(expr)                                                                            \
    do {                                                                                         \
        hipError_t status_ = (expr);                                                             \
        if (status_ != hipSuccess) {                                                             \
            return fail(                                                                         \
                MEMB_HIP_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(status_));   \
        }                                                                                        \
    } while (0)

// ---------------------------------------------------------------------------
// decode_trained
// ---------------------------------------------------------------------------

struct TrainedParams {
    const uint32_t* rows;        // batch -> row id; null = identity (row = batch position)
    float* out;
    unsigned long long n;
    unsigned long long ld;
    unsigned long long colOff;
    const uint4* streams;           // re-packed bitstreams: big-endian dwords, one 16-byte aligned run per row
    const uint32_t* streamStarts;   // [nRows + 1] first 16-byte piece of each row's run
    const uint16_t* segmentIndex;   // [nRows][lanesPerWord - 1] bit offsets of segments 1.. from the stream start
    uint16_t* segmentIndexOut;      // OUT_INDEX: index being built, [nRows][indexLanes - 1]
    const uint32_t* table;          // 8-byte entries, see TableEntry
    const float* codebook;          // 256 centroids, or 256 centroid pairs (FAST)
    unsigned long long nRows;
    uint32_t tableDwords;     // multiple of 4
    uint32_t codebookDwords;  // 256 or 512
    uint32_t rootBits;
    uint32_t dim;
    uint32_t slotDwords;      // LDS dwords reserved per bitstream, multiple of 4
    uint32_t slotMagic;       // fastDivide magic for slotDwords / 4
    uint32_t lanesPerWord;    // G
    uint32_t laneMagic;       // fastDivide magic for G
    uint32_t wordsPerWave;    // 64 / G
    uint32_t segmentSymbols;  // S, multiple of the decode group (4, or 8 when FAST)
    uint32_t keyRowBytes;     // bytes per word in the symbol tile
    uint32_t keyTileDwords;   // dwords of the symbol tile of one wave
    uint32_t pieceMagic;      // fastDivide magic for dim / 4 (vector output)
    uint32_t indexLanes;      // OUT_INDEX: lanes per word of the index being built
    uint32_t indexSegmentSymbols;
    uint32_t debugFlags;      // measurement only (MEMB_HIP_DEBUG): 1 = skip decode, 2 = skip output
    uint32_t accumulate;      // epilogue: add to what the output already holds ...
    float divisor;            // ... and / or divide by this (0 = no division)
};

enum OutputMode { OUT_SCALAR = 0, OUT_VEC4 = 1, OUT_FLAT = 2, OUT_INDEX = 3 };

// Device form of one lookup-table entry (logical layout: memb::DecodeTable).
//   x: leaf    -> code length                       (bits 8..31 zero)
//      pointer -> TABLE_POINTER_FLAG | extra bits | first sub-table entry << 8
//   y: leaf    -> the symbol replicated into every byte (or every nibble, FAST),
//                 so that packing symbol s of a group is one AND-OR with a
//                 constant mask
struct TableEntry {
    uint32_t x;
    uint32_t y;
};

// q / d with a host-computed magic = ceil(2^32 / d) (exact while q * d < 2^32);
// magic == 0 means "no magic" (d == 1, or the range is too large): plain division.
__device__ __forceinline__ uint32_t fastDivide(uint32_t q, uint32_t magic, uint32_t d)
{
    return magic ? __umulhi(q, magic) : q / d;
}

// Orders this wave's LDS writes before its later LDS reads (and vice versa).
// LDS operations of one wave execute in order; the fence makes the compiler
// wait for them and keeps it from moving accesses across.
__device__ __forceinline__ void waveLdsFence()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
}

// Single-lane-op IEEE fp32 add / sub / mul. Written as instructions because the
// optimiser otherwise pairs neighbouring operations into v_pk_add_f32 /
// v_pk_mul_f32, and the packed forms flush subnormal values on gfx950 (measured:
// min = 1e-40 came back as 0), which would break bit parity with the CPU.
__device__ __forceinline__ float addRn(float a, float b)
{
    float r;
    asm("v_add_f32_e32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

__device__ __forceinline__ float subRn(float a, float b)
{
    float r;
    asm("v_sub_f32_e32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

__device__ __forceinline__ float mulRn(float a, float b)
{
    float r;
    asm("v_mul_f32_e32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// Epilogue of ReadersUnion 'average' (reference python/memb/readers_union.py:18:
// numpy.mean over the readers = fp32 sums in reader order, then one division
// by the reader count): later readers add to what earlier ones stored, the last
// one divides. Same operations and order as numpy, so the result is bit-identical.
__device__ __forceinline__ float epilogue(float value, const float* destination, uint32_t accumulate, float divisor)
{
    if (accumulate) {
        value = addRn(*destination, value);
    }
    if (divisor != 0.f) {
        value = __fdiv_rn(value, divisor);
    }
    return value;
}

__device__ __forceinline__ float4 epilogue4(float4 value, const float* destination, uint32_t accumulate, float divisor)
{
    if (accumulate) {
        const float4 old = *reinterpret_cast<const float4*>(destination);
        value.x = addRn(old.x, value.x);
        value.y = addRn(old.y, value.y);
        value.z = addRn(old.z, value.z);
        value.w = addRn(old.w, value.w);
    }
    if (divisor != 0.f) {
        value.x = __fdiv_rn(value.x, divisor);
        value.y = __fdiv_rn(value.y, divisor);
        value.z = __fdiv_rn(value.z, divisor);
        value.w = __fdiv_rn(value.w, divisor);
    }
    return value;
}

// ---- building blocks shared by the one-shot and the persistent kernel ----

struct LaneRole {
    uint32_t word;      // word of the tile this lane works on
    uint32_t segment;   // segment of that word
    bool spare;         // 64 % G lanes at the top: decode word 0's slot, store nothing
};

__device__ __forceinline__ LaneRole laneRole(const TrainedParams& p, uint32_t lane)
{
    LaneRole role;
    const uint32_t laneWord = fastDivide(lane, p.laneMagic, p.lanesPerWord);
    role.segment = lane - laneWord * p.lanesPerWord;
    role.spare = laneWord >= p.wordsPerWave;
    role.word = role.spare ? 0 : laneWord;
    return role;
}

// Row id of the lane's word in tile `tile`; MISSING for padding lanes and past the batch end.
__device__ __forceinline__ uint32_t loadTileRow(const TrainedParams& p, unsigned long long tile, const LaneRole& role)
{
    const unsigned long long index = tile * p.wordsPerWave + role.word;
    if (role.spare || index >= p.n) {
        return MISSING;
    }
    return p.rows ? p.rows[index] : static_cast<uint32_t>(index);
}

struct WordMeta {
    uint32_t row;
    uint32_t start;         // first 16-byte piece of the word's bitstream
    uint32_t segmentBits;   // bit offset of the lane's segment inside that stream
};

__device__ __forceinline__ WordMeta loadWordMeta(const TrainedParams& p, uint32_t row, const LaneRole& role)
{
    WordMeta meta;
    meta.row = row;
    meta.start = 0;
    meta.segmentBits = 0;
    if (row < p.nRows) {
        meta.start = p.streamStarts[row];
        if (role.segment > 0) {
            meta.segmentBits =
                p.segmentIndex[static_cast<unsigned long long>(row) * (p.lanesPerWord - 1) + role.segment - 1];
        }
    }
    return meta;
}

constexpr int STREAM_REGISTERS = 4;   // 16-byte pieces one lane can hold for a prefetched tile

// Named members, not an array: indexed storage ends up in scratch memory, and a
// load whose result goes to scratch is waited for at once, which would undo the prefetch.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));   // plain SSA value (HIP's uint4 is a class)

struct StreamRegisters {
    u32x4 r0, r1, r2, r3;
};

// One round of the tile's bitstream copy: piece q = (word, 16-byte piece) -> one lane.
// A slot's worth of pieces is read from each row's start (running into the next
// rows' streams, which is harmless; the array ends with a guard of one slot).
// Absent words read the start of the array and never emit what they decode;
// lanes past the tile's last piece re-read its last piece.
__device__ __forceinline__ void issueStreamLoad(
    const TrainedParams& p, uint32_t sourceStart, uint32_t lane, uint32_t round, u32x4& destination)
{
    const uint32_t piecesPerWord = p.slotDwords / 4;
    const uint32_t totalPieces = p.wordsPerWave * piecesPerWord;
    if (round * WAVE < totalPieces) {   // wave-uniform
        const uint32_t q = min(round * WAVE + lane, totalPieces - 1);
        const uint32_t w = fastDivide(q, p.slotMagic, piecesPerWord);
        const uint32_t piece = q - w * piecesPerWord;
        const uint32_t wordStart = __shfl(sourceStart, w * p.lanesPerWord);
        destination = reinterpret_cast<const u32x4*>(p.streams)[static_cast<unsigned long long>(wordStart) + piece];
    }
}

__device__ __forceinline__ void issueStreamLoads(
    const TrainedParams& p, const WordMeta& meta, uint32_t lane, uint32_t firstRound, StreamRegisters& v)
{
    const uint32_t sourceStart = meta.row < p.nRows ? meta.start : 0u;
    issueStreamLoad(p, sourceStart, lane, firstRound + 0, v.r0);
    issueStreamLoad(p, sourceStart, lane, firstRound + 1, v.r1);
    issueStreamLoad(p, sourceStart, lane, firstRound + 2, v.r2);
    issueStreamLoad(p, sourceStart, lane, firstRound + 3, v.r3);
}

// Into the LDS slots (already big-endian dwords, so the decoder extracts bits with plain shifts).
__device__ __forceinline__ void writeStream(
    const TrainedParams& p, uint32_t* slots, uint32_t lane, uint32_t round, const u32x4& value)
{
    const uint32_t piecesPerWord = p.slotDwords / 4;
    const uint32_t totalPieces = p.wordsPerWave * piecesPerWord;
    const uint32_t q = round * WAVE + lane;
    if (q < totalPieces) {
        const uint32_t w = fastDivide(q, p.slotMagic, piecesPerWord);
        const uint32_t piece = q - w * piecesPerWord;
        *reinterpret_cast<u32x4*>(slots + w * p.slotDwords + 4 * piece) = value;
    }
}

__device__ __forceinline__ void writeStreams(
    const TrainedParams& p, uint32_t* slots, uint32_t lane, uint32_t firstRound, const StreamRegisters& v)
{
    writeStream(p, slots, lane, firstRound + 0, v.r0);
    writeStream(p, slots, lane, firstRound + 1, v.r1);
    writeStream(p, slots, lane, firstRound + 2, v.r2);
    writeStream(p, slots, lane, firstRound + 3, v.r3);
}

// FAST: codebook of at most 16 centroids and no code longer than 8 bits (2- and
// 4-bit models). Eight symbols are decoded per 64-bit window (7 * 8 consumed bits
// + 8 looked-ahead bits fit), symbols are staged as nibbles, and the output phase
// fetches two centroids per LDS read from a 256-entry table of pairs.
//
// Decode the lane's segment from its word's LDS slot into the symbol tile
// (or, OUT_INDEX, record segment start positions).
template <bool HAS_SUB, int MODE, bool FAST>
__device__ __forceinline__ void decodeSegment(
    const TrainedParams& p, const TableEntry* tableLds, const uint32_t* slots, uint32_t* keyTile,
    const LaneRole& role, const WordMeta& meta)
{
    constexpr int GROUP = FAST ? 8 : 4;
    constexpr uint32_t KEY_BITS = FAST ? 4 : 8;
    constexpr uint32_t KEY_MASK = FAST ? 0xFu : 0xFFu;

    const bool present = meta.row < p.nRows;
    const uint32_t* slot = slots + role.word * p.slotDwords;
    uint8_t* keyBytes = reinterpret_cast<uint8_t*>(keyTile);
    const uint32_t lastWindow = p.slotDwords - 3;
    const uint32_t rootShift = 32 - p.rootBits;
    uint32_t bitPos = meta.segmentBits;   // streams start on a slot boundary
    // byte position of this lane's first group inside the symbol tile
    uint32_t keyOffset = role.word * p.keyRowBytes + role.segment * (p.segmentSymbols * KEY_BITS / 8);
    const uint32_t keyRowEnd = role.spare ? 0 : (role.word + 1) * p.keyRowBytes;
    const uint32_t absentFill = present ? 0u : 0xFFFFFFFFu;   // byte keys: ZERO_KEY everywhere
    uint32_t nextIndexSymbol = p.indexSegmentSymbols;
    uint32_t indexSlot = 0;

    for (uint32_t j = 0; j < p.segmentSymbols; j += GROUP) {
        if (MODE == OUT_INDEX) {
            // one lane per word here; record where every indexSegmentSymbols-th symbol starts
            if (j == nextIndexSymbol) {
                if (present && indexSlot + 1 < p.indexLanes) {
                    p.segmentIndexOut[static_cast<unsigned long long>(meta.row) * (p.indexLanes - 1) + indexSlot] =
                        static_cast<uint16_t>(bitPos);
                }
                ++indexSlot;
                nextIndexSymbol += p.indexSegmentSymbols;
            }
        }
        const uint32_t d = min(bitPos >> 5, lastWindow);
        const uint32_t shift = bitPos & 31;
        const uint32_t w0 = slot[d];
        const uint32_t w1 = slot[d + 1];
        const uint32_t w2 = slot[d + 2];
        // 64 valid bits starting at the current bit position, MSB first.
        unsigned long long window = ((static_cast<unsigned long long>(w0) << 32) | w1) << shift;
        window |= static_cast<uint32_t>(static_cast<unsigned long long>(w2) >> (32 - shift));
        uint32_t keys = 0;
        uint32_t lengths = 0;
#pragma unroll
        for (int s = 0; s < GROUP; ++s) {
            TableEntry entry = tableLds[static_cast<uint32_t>(window >> 32) >> rootShift];
            if (HAS_SUB) {
                if (entry.x & memb::TABLE_POINTER_FLAG) {
                    const uint32_t subBits = entry.x & 0xff;
                    const uint32_t base = (entry.x & ~memb::TABLE_POINTER_FLAG) >> 8;
                    const uint32_t subIndex =
                        static_cast<uint32_t>((window << p.rootBits) >> 32) >> (32 - subBits);
                    entry = tableLds[base + subIndex];
                }
            }
            window <<= (entry.x & 63);
            lengths += entry.x;
            keys |= entry.y & (KEY_MASK << (KEY_BITS * s));
        }
        bitPos += lengths;
        if (MODE != OUT_INDEX) {
            keys |= absentFill;
            if (FAST) {
                // rows are dim / 2 bytes: 2-byte aligned only
                if (keyOffset + 2 <= keyRowEnd) {
                    *reinterpret_cast<uint16_t*>(keyBytes + keyOffset) = static_cast<uint16_t>(keys);
                }
                if (keyOffset + 4 <= keyRowEnd) {
                    *reinterpret_cast<uint16_t*>(keyBytes + keyOffset + 2) = static_cast<uint16_t>(keys >> 16);
                }
            } else {
                if (keyOffset + 4 <= keyRowEnd) {
                    *reinterpret_cast<uint32_t*>(keyBytes + keyOffset) = keys;
                }
            }
            keyOffset += 4;
        }
    }
}

// Symbol tile -> fp32 rows: codebook gather and row-contiguous stores.
template <int MODE, bool FAST>
__device__ __forceinline__ void outputTile(
    const TrainedParams& p, const uint32_t* codebookLds, const uint32_t* keyTile, unsigned long long tileBase,
    uint32_t tileWords, uint32_t lane, const LaneRole& role, bool present)
{
    const float* centroidLds = reinterpret_cast<const float*>(codebookLds);
    const float2* pairLds = reinterpret_cast<const float2*>(codebookLds);
    const uint8_t* keyBytes = reinterpret_cast<const uint8_t*>(keyTile);

    // Nibble keys have no spare code for "absent" (byte keys use ZERO_KEY): rows of
    // absent words are zeroed after the tile is written -- or, when an epilogue
    // reads the destination, their pieces go through it as zeros.
    const bool hasEpilogue = p.accumulate || p.divisor != 0.f;   // wave-uniform
    unsigned long long absent = 0;
    if (FAST) {
        absent = __ballot(!present && !role.spare && role.segment == 0 && role.word < tileWords);
    }
    const bool checkWords = FAST && hasEpilogue && absent != 0;

    if (MODE == OUT_FLAT || MODE == OUT_VEC4) {
        // Piece q = 4 consecutive floats; the symbol tile is linear in q for both layouts
        // (byte keys: rows of dim bytes; nibble keys: rows of dim / 2 bytes).
        // BURST pieces per lane are gathered first and then stored back to back, so a
        // tile reaches memory as one burst of consecutive KiBs rather than one KiB per
        // LDS round trip.
        constexpr int BURST = 5;
        const uint32_t piecesPerWord = p.dim / 4;
        const uint32_t pieces = tileWords * piecesPerWord;
        float* tileOut = p.out + tileBase * p.ld + p.colOff;
        for (uint32_t q0 = lane; q0 < pieces; q0 += WAVE * BURST) {
            uint32_t k[BURST];
            float4 f[BURST];
#pragma unroll
            for (int u = 0; u < BURST; ++u) {
                const uint32_t q = min(q0 + WAVE * u, pieces - 1);
                k[u] = FAST ? reinterpret_cast<const uint16_t*>(keyTile)[q] : keyTile[q];
            }
#pragma unroll
            for (int u = 0; u < BURST; ++u) {
                if (FAST) {
                    const float2 a = pairLds[k[u] & 0xff];
                    const float2 b = pairLds[k[u] >> 8];
                    f[u] = make_float4(a.x, a.y, b.x, b.y);
                } else {
                    f[u].x = centroidLds[k[u] & 0xff];
                    f[u].y = centroidLds[(k[u] >> 8) & 0xff];
                    f[u].z = centroidLds[(k[u] >> 16) & 0xff];
                    f[u].w = centroidLds[k[u] >> 24];
                }
            }
#pragma unroll
            for (int u = 0; u < BURST; ++u) {
                const uint32_t q = q0 + WAVE * u;
                if (q < pieces) {
                    float* destination;
                    if (MODE == OUT_FLAT && !checkWords) {
                        destination = tileOut + 4 * static_cast<size_t>(q);
                    } else {
                        const uint32_t w = fastDivide(q, p.pieceMagic, piecesPerWord);
                        const uint32_t c = q - w * piecesPerWord;
                        destination = tileOut + w * p.ld + 4 * c;
                        if (checkWords && ((absent >> (w * p.lanesPerWord)) & 1)) {
                            f[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                        }
                    }
                    if (hasEpilogue) {   // off the common path
                        f[u] = epilogue4(f[u], destination, p.accumulate, p.divisor);
                    }
                    *reinterpret_cast<float4*>(destination) = f[u];
                }
            }
        }
    } else {
        const uint32_t total = tileWords * p.dim;
        for (uint32_t q = lane; q < total; q += WAVE) {
            const uint32_t w = q / p.dim;
            const uint32_t c = q - w * p.dim;
            float value;
            if (FAST) {
                const uint32_t k = keyBytes[w * p.keyRowBytes + (c >> 1)];
                value = pairLds[(k >> (4 * (c & 1))) & 15].x;
                if (checkWords && ((absent >> (w * p.lanesPerWord)) & 1)) {
                    value = 0.f;
                }
            } else {
                value = centroidLds[keyBytes[w * p.keyRowBytes + c]];
            }
            float* destination = p.out + (tileBase + w) * p.ld + p.colOff + c;
            if (hasEpilogue) {
                value = epilogue(value, destination, p.accumulate, p.divisor);
            }
            *destination = value;
        }
    }

    if (FAST && !hasEpilogue) {
        // zero the rows of absent words (same wave, same addresses: program order holds)
        while (absent) {
            const uint32_t w = fastDivide(__ffsll(static_cast<long long>(absent)) - 1, p.laneMagic, p.lanesPerWord);
            absent &= absent - 1;
            float* rowOut = p.out + (tileBase + w) * p.ld + p.colOff;
            if (MODE == OUT_SCALAR) {
                for (uint32_t c = lane; c < p.dim; c += WAVE) {
                    rowOut[c] = 0.f;
                }
            } else {
                for (uint32_t c = lane; c < p.dim / 4; c += WAVE) {
                    reinterpret_cast<float4*>(rowOut)[c] = make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
        }
    }
}

struct WaveLds {
    const TableEntry* table;
    const uint32_t* codebook;
    uint32_t* slots;
    uint32_t* keyTile;
};

// LDS layout: lookup table | codebook | per wave { bitstream slots | symbol tile }.
// Loads table and codebook; ends with a block barrier.
template <int MODE>
__device__ __forceinline__ WaveLds setUpLds(const TrainedParams& p, uint32_t* lds)
{
    const uint32_t wave = threadIdx.x / WAVE;
    uint32_t* codebookLds = lds + p.tableDwords;
    const uint32_t perWave = p.wordsPerWave * p.slotDwords + p.keyTileDwords;
    WaveLds result;
    result.table = reinterpret_cast<const TableEntry*>(lds);
    result.codebook = codebookLds;
    result.slots = lds + p.tableDwords + p.codebookDwords + wave * perWave;
    result.keyTile = result.slots + p.wordsPerWave * p.slotDwords;

    for (uint32_t i = threadIdx.x; i < p.tableDwords / 4; i += blockDim.x) {
        reinterpret_cast<uint4*>(lds)[i] = reinterpret_cast<const uint4*>(p.table)[i];
    }
    if (MODE != OUT_INDEX) {
        for (uint32_t i = threadIdx.x; i < p.codebookDwords; i += blockDim.x) {
            codebookLds[i] = reinterpret_cast<const uint32_t*>(p.codebook)[i];
        }
    }
    __syncthreads();
    return result;
}

// One-shot kernel: one tile per wavefront. Used to build the segment index
// (OUT_INDEX) and for tiles too wide for the persistent kernel's registers.
template <bool HAS_SUB, int MODE, bool FAST>
__global__ void decode_trained(TrainedParams p)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const uint32_t lane = threadIdx.x & (WAVE - 1);
    const WaveLds mem = setUpLds<MODE>(p, lds);

    const unsigned long long tile =
        static_cast<unsigned long long>(blockIdx.x) * (blockDim.x / WAVE) + threadIdx.x / WAVE;
    const unsigned long long tileBase = tile * p.wordsPerWave;
    if (tileBase >= p.n) {
        return;
    }
    const uint32_t tileWords =
        static_cast<uint32_t>(min(static_cast<unsigned long long>(p.wordsPerWave), p.n - tileBase));

    const LaneRole role = laneRole(p, lane);
    const WordMeta meta = loadWordMeta(p, loadTileRow(p, tile, role), role);

    const uint32_t rounds = (p.wordsPerWave * (p.slotDwords / 4) + WAVE - 1) / WAVE;
    for (uint32_t round = 0; round < rounds; round += STREAM_REGISTERS) {
        StreamRegisters v;
        issueStreamLoads(p, meta, lane, round, v);
        writeStreams(p, mem.slots, lane, round, v);
    }
    waveLdsFence();

    decodeSegment<HAS_SUB, MODE, FAST>(p, mem.table, mem.slots, mem.keyTile, role, meta);
    if (MODE == OUT_INDEX) {
        return;
    }
    waveLdsFence();
    outputTile<MODE, FAST>(p, mem.codebook, mem.keyTile, tileBase, tileWords, lane, role, meta.row < p.nRows);
}

// Persistent kernel: every wavefront walks tiles wave, wave + W, wave + 2W, ...
// and keeps three tiles' worth of loads in flight, so that no decode waits for
// global memory: while tile t is decoded, the bitstream bytes of tile t + 1 sit
// in registers, the offsets / segment positions of tile t + 2 and the row ids
// of tile t + 3 are on their way. Each of those hops depends on the previous
// one (row id -> offset -> stream bytes); issued back to back they are what a
// one-tile wavefront spends most of its life waiting for.
template <bool HAS_SUB, int MODE, bool FAST>
__global__ void decode_trained_persistent(TrainedParams p)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const uint32_t lane = threadIdx.x & (WAVE - 1);
    const WaveLds mem = setUpLds<MODE>(p, lds);

    const unsigned long long stride = static_cast<unsigned long long>(gridDim.x) * (blockDim.x / WAVE);
    unsigned long long tile = static_cast<unsigned long long>(blockIdx.x) * (blockDim.x / WAVE) + threadIdx.x / WAVE;
    const unsigned long long tiles = (p.n + p.wordsPerWave - 1) / p.wordsPerWave;
    if (tile >= tiles) {
        return;
    }
    const LaneRole role = laneRole(p, lane);

    // prologue: fill the pipeline (these hops are dependent and exposed, once per wavefront)
    uint32_t rowLoading = loadTileRow(p, tile + 3 * stride, role);
    WordMeta meta0 = loadWordMeta(p, loadTileRow(p, tile, role), role);
    WordMeta meta1 = loadWordMeta(p, loadTileRow(p, tile + stride, role), role);
    WordMeta metaLoading = loadWordMeta(p, loadTileRow(p, tile + 2 * stride, role), role);
    StreamRegisters streams;
    issueStreamLoads(p, meta0, lane, 0, streams);
    writeStreams(p, mem.slots, lane, 0, streams);
    issueStreamLoads(p, meta1, lane, 0, streams);
    waveLdsFence();

    // Invariant at the top, for the current tile t:
    //   LDS slots hold the bitstreams of t;
    //   in flight since the end of the previous round: `streams` = stream bytes of t + 1,
    //   `metaLoading` = offsets of t + 2, `rowLoading` = row ids of t + 3.
    // In-flight registers are touched at ONE point per round, right after the
    // decode (which gave them a whole decode to land) and before this round's
    // stores are issued, so that the wait there is only for loads; the new
    // loads are the last memory instructions of the round.
    for (; tile < tiles; tile += stride) {
        const unsigned long long tileBase = tile * p.wordsPerWave;
        const uint32_t tileWords =
            static_cast<uint32_t>(min(static_cast<unsigned long long>(p.wordsPerWave), p.n - tileBase));

        if (!(p.debugFlags & 1)) {
            decodeSegment<HAS_SUB, MODE, FAST>(p, mem.table, mem.slots, mem.keyTile, role, meta0);
        }
        waveLdsFence();

        // consume point
        writeStreams(p, mem.slots, lane, 0, streams);   // bitstreams of t + 1 replace those of t
        // (copies pinned here: left to the register allocator they move to the loop
        // header, and the wait for the loads moves with them)
        WordMeta meta2;
        uint32_t row3;
        asm volatile("v_mov_b32 %0, %1" : "=v"(meta2.row) : "v"(metaLoading.row));
        asm volatile("v_mov_b32 %0, %1" : "=v"(meta2.start) : "v"(metaLoading.start));
        asm volatile("v_mov_b32 %0, %1" : "=v"(meta2.segmentBits) : "v"(metaLoading.segmentBits));
        asm volatile("v_mov_b32 %0, %1" : "=v"(row3) : "v"(rowLoading));
        __builtin_amdgcn_sched_barrier(0);

        if (!(p.debugFlags & 2)) {
            outputTile<MODE, FAST>(p, mem.codebook, mem.keyTile, tileBase, tileWords, lane, role, meta0.row < p.nRows);
        }
        __builtin_amdgcn_sched_barrier(0);

        // next round of loads; each uses what the previous round fetched
        issueStreamLoads(p, meta2, lane, 0, streams);          // stream bytes of t + 2
        metaLoading = loadWordMeta(p, row3, role);              // offsets of t + 3
        rowLoading = loadTileRow(p, tile + 4 * stride, role);   // row ids of t + 4
        meta0 = meta1;
        meta1 = meta2;
        waveLdsFence();
    }
}

// Staging-time re-pack of the file's bitstreams (byte aligned, insertion order,
// reference src/trained_compression.cpp:65-71) into the layout the decoder
// reads: row r's stream starts at piece streamStarts[r], 16-byte aligned, in
// row (= sorted key) order, stored as big-endian dwords. One wavefront per row.
__global__ void repack_streams(
    const uint8_t* packed, unsigned long long packedBytes, const uint32_t* valueOffsets, const uint32_t* streamStarts,
    unsigned long long nRows, uint4* streams)
{
    const unsigned long long row = (static_cast<unsigned long long>(blockIdx.x) * blockDim.x + threadIdx.x) / WAVE;
    if (row >= nRows) {
        return;
    }
    const uint32_t lane = threadIdx.x & (WAVE - 1);
    const uint32_t first = streamStarts[row];
    const uint32_t pieces = streamStarts[row + 1] - first;
    const unsigned long long source = valueOffsets[row];
    for (uint32_t piece = lane; piece < pieces; piece += WAVE) {
        uint32_t dwords[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            uint32_t value = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const unsigned long long at = source + 16ull * piece + 4 * k + b;
                value = (value << 8) | (at < packedBytes ? packed[at] : 0u);
            }
            dwords[k] = value;
        }
        streams[static_cast<unsigned long long>(first) + piece] = make_uint4(dwords[0], dwords[1], dwords[2], dwords[3]);
    }
}

// ---------------------------------------------------------------------------
// dequant_uniform / gather_full
// ---------------------------------------------------------------------------

struct UniformParams {
    uint32_t accumulate;
    float divisor;
    const uint32_t* rows;
    float* out;
    unsigned long long n;
    unsigned long long ld;
    unsigned long long colOff;
    const uint8_t* values;   // dense [nRows][dim]
    const float2* minMax;    // [nRows]
    unsigned long long nRows;
    uint32_t dim;
    uint32_t wordsPerBlock;
    uint32_t pieceMagic;     // ceil(2^32 / (dim / 4)), vector path
    float levels;
};

// reference src/uniform_compression.cpp:70-71, evaluated left to right in fp32:
// sub, mul, div, add -- each correctly rounded, nothing fused, subnormals kept.
__device__ __forceinline__ float dequant(float minValue, float range, uint32_t v, float levels)
{
    const float scaled = mulRn(range, static_cast<float>(v));
    return addRn(minValue, __fdiv_rn(scaled, levels));
}

constexpr uint32_t ROWWISE_MAX_WORDS = 64;   // words per block of the row-wise kernels
constexpr int ROWWISE_BATCH = 4;             // 16-byte pieces a thread keeps in flight

// Row-wise kernels (uniform, full): a block first stages the row ids (and the
// per-row constants) of its words in LDS -- one dependent pair of global loads
// per block instead of per piece -- then every thread keeps ROWWISE_BATCH value
// loads in flight before it converts and stores.
template <bool VEC4>
__global__ void dequant_uniform(UniformParams p)
{
    __shared__ uint32_t rowLds[ROWWISE_MAX_WORDS];
    __shared__ float2 minMaxLds[ROWWISE_MAX_WORDS];
    const unsigned long long blockBase = static_cast<unsigned long long>(blockIdx.x) * p.wordsPerBlock;
    const uint32_t blockWords =
        static_cast<uint32_t>(min(static_cast<unsigned long long>(p.wordsPerBlock), p.n - blockBase));
    if (threadIdx.x < blockWords) {
        const uint32_t row = p.rows[blockBase + threadIdx.x];
        rowLds[threadIdx.x] = row;
        minMaxLds[threadIdx.x] = row < p.nRows ? p.minMax[row] : make_float2(0.f, 0.f);
    }
    __syncthreads();

    if (VEC4) {
        const uint32_t piecesPerWord = p.dim / 4;
        const uint32_t pieces = blockWords * piecesPerWord;
        for (uint32_t q0 = threadIdx.x; q0 < pieces; q0 += blockDim.x * ROWWISE_BATCH) {
            uint32_t word[ROWWISE_BATCH];
            uint32_t column[ROWWISE_BATCH];
            uint32_t packed[ROWWISE_BATCH];
#pragma unroll
            for (int u = 0; u < ROWWISE_BATCH; ++u) {
                const uint32_t q = min(q0 + u * blockDim.x, pieces - 1);
                word[u] = fastDivide(q, p.pieceMagic, piecesPerWord);
                column[u] = q - word[u] * piecesPerWord;
                const uint32_t row = rowLds[word[u]];
                packed[u] = 0;
                if (row < p.nRows) {
                    packed[u] = *reinterpret_cast<const uint32_t*>(
                        p.values + static_cast<unsigned long long>(row) * p.dim + 4 * column[u]);
                }
            }
#pragma unroll
            for (int u = 0; u < ROWWISE_BATCH; ++u) {
                if (q0 + u * blockDim.x < pieces) {
                    float4 f = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (rowLds[word[u]] < p.nRows) {
                        const float2 mm = minMaxLds[word[u]];
                        const float range = subRn(mm.y, mm.x);
                        f.x = dequant(mm.x, range, packed[u] & 0xff, p.levels);
                        f.y = dequant(mm.x, range, (packed[u] >> 8) & 0xff, p.levels);
                        f.z = dequant(mm.x, range, (packed[u] >> 16) & 0xff, p.levels);
                        f.w = dequant(mm.x, range, packed[u] >> 24, p.levels);
                    }
                    float* dst = p.out + (blockBase + word[u]) * p.ld + p.colOff + 4 * column[u];
                    if (p.accumulate || p.divisor != 0.f) {
                        f = epilogue4(f, dst, p.accumulate, p.divisor);
                    }
                    *reinterpret_cast<float4*>(dst) = f;
                }
            }
        }
    } else {
        const uint32_t total = blockWords * p.dim;
        for (uint32_t q = threadIdx.x; q < total; q += blockDim.x) {
            const uint32_t w = q / p.dim;
            const uint32_t c = q - w * p.dim;
            const uint32_t row = rowLds[w];
            float f = 0.f;
            if (row < p.nRows) {
                const float2 mm = minMaxLds[w];
                const float range = subRn(mm.y, mm.x);
                f = dequant(mm.x, range, p.values[static_cast<unsigned long long>(row) * p.dim + c], p.levels);
            }
            float* dst = p.out + (blockBase + w) * p.ld + p.colOff + c;
            if (p.accumulate || p.divisor != 0.f) {
                f = epilogue(f, dst, p.accumulate, p.divisor);
            }
            *dst = f;
        }
    }
}

struct FullParams {
    uint32_t accumulate;
    float divisor;
    const uint32_t* rows;
    float* out;
    unsigned long long n;
    unsigned long long ld;
    unsigned long long colOff;
    const float* values;     // dense [nRows][dim]
    unsigned long long nRows;
    uint32_t dim;
    uint32_t wordsPerBlock;
    uint32_t pieceMagic;
};

template <bool VEC4>
__global__ void gather_full(FullParams p)
{
    __shared__ uint32_t rowLds[ROWWISE_MAX_WORDS];
    const unsigned long long blockBase = static_cast<unsigned long long>(blockIdx.x) * p.wordsPerBlock;
    const uint32_t blockWords =
        static_cast<uint32_t>(min(static_cast<unsigned long long>(p.wordsPerBlock), p.n - blockBase));
    if (threadIdx.x < blockWords) {
        rowLds[threadIdx.x] = p.rows[blockBase + threadIdx.x];
    }
    __syncthreads();
    if (VEC4) {
        const uint32_t piecesPerWord = p.dim / 4;
        const uint32_t pieces = blockWords * piecesPerWord;
        for (uint32_t q0 = threadIdx.x; q0 < pieces; q0 += blockDim.x * ROWWISE_BATCH) {
            uint32_t word[ROWWISE_BATCH];
            uint32_t column[ROWWISE_BATCH];
            float4 f[ROWWISE_BATCH];
#pragma unroll
            for (int u = 0; u < ROWWISE_BATCH; ++u) {
                const uint32_t q = min(q0 + u * blockDim.x, pieces - 1);
                word[u] = fastDivide(q, p.pieceMagic, piecesPerWord);
                column[u] = q - word[u] * piecesPerWord;
                const uint32_t row = rowLds[word[u]];
                f[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (row < p.nRows) {
                    f[u] = *reinterpret_cast<const float4*>(
                        p.values + static_cast<unsigned long long>(row) * p.dim + 4 * column[u]);
                }
            }
#pragma unroll
            for (int u = 0; u < ROWWISE_BATCH; ++u) {
                if (q0 + u * blockDim.x < pieces) {
                    float* dst = p.out + (blockBase + word[u]) * p.ld + p.colOff + 4 * column[u];
                    if (p.accumulate || p.divisor != 0.f) {
                        f[u] = epilogue4(f[u], dst, p.accumulate, p.divisor);
                    }
                    *reinterpret_cast<float4*>(dst) = f[u];
                }
            }
        }
    } else {
        const uint32_t total = blockWords * p.dim;
        for (uint32_t q = threadIdx.x; q < total; q += blockDim.x) {
            const uint32_t w = q / p.dim;
            const uint32_t c = q - w * p.dim;
            const uint32_t row = rowLds[w];
            float f = row < p.nRows ? p.values[static_cast<unsigned long long>(row) * p.dim + c] : 0.f;
            float* dst = p.out + (blockBase + w) * p.ld + p.colOff + c;
            if (p.accumulate || p.divisor != 0.f) {
                f = epilogue(f, dst, p.accumulate, p.divisor);
            }
            *dst = f;
        }
    }
}

// ceil(2^32 / d) for fastDivide: exact for every q <= maxQ when maxQ * d < 2^32.
// Returns 0 (plain division) for d == 1, where the magic does not fit 32 bits,
// and when the range is too large.
uint32_t magicFor(uint32_t d, uint64_t maxQ)
{
    if (d <= 1 || maxQ * d >= (1ull << 32)) {
        return 0;
    }
    return static_cast<uint32_t>(((1ull << 32) + d - 1) / d);
}

uint32_t envUint(const char* name, uint32_t fallback)
{
    const char* text = std::getenv(name);
    if (!text || !*text) {
        return fallback;
    }
    return static_cast<uint32_t>(std::strtoul(text, nullptr, 10));
}

}  // namespace

// ---------------------------------------------------------------------------
// Context
// ---------------------------------------------------------------------------

struct memb_hip_ctx {
    int device = 0;
    uint32_t storage = 0;
    uint32_t dim = 0;
    uint64_t nRows = 0;
    uint64_t deviceBytes = 0;
    hipStream_t stream = nullptr;
    std::vector<void*> allocations;

    // trained
    uint4* streams = nullptr;            // re-packed bitstreams
    uint32_t* streamStarts = nullptr;    // [nRows + 1]
    uint32_t* table = nullptr;
    float* codebook = nullptr;
    bool fast = false;                   // <= 16 centroids, codes <= 8 bits, one-level table
    memb::DecodeTable hostTable;
    uint32_t tableDwords = 0;
    uint32_t maxStreamBytes = 0;
    uint32_t slotDwords = 0;
    uint16_t* segmentIndex = nullptr;    // [nRows][lanesPerWord - 1]
    uint32_t lanesPerWord = 1;           // G: lanes that share one word
    uint32_t segmentSymbols = 0;         // S: symbols per lane, multiple of 4
    std::vector<uint32_t> streamBytes;   // per row, host side (reporting only)
    uint32_t ldsLimit = 0;
    uint32_t cuCount = 0;

    // uniform / full
    uint8_t* uniformValues = nullptr;
    float2* minMax = nullptr;
    float levels = 0.f;
    float* fullValues = nullptr;

    // staging for the host-buffer entry point
    uint32_t* stagedRows = nullptr;
    float* stagedOut = nullptr;
    size_t stagedCapacity = 0;   // words
    size_t stagedLd = 0;
    std::mutex mutex;
};

namespace {

struct TrainedGeometry {
    uint32_t waves;      // wavefronts per block
    uint32_t ldsBytes;   // dynamic LDS per block
    int mode;
};

uint32_t roundUp4(uint32_t v)
{
    return (v + 3) / 4 * 4;
}

// symbol tile of one wave: rows of dim bytes (dword aligned) or dim / 2 bytes (FAST)
uint32_t keyRowBytes(const memb_hip_ctx* ctx)
{
    return ctx->fast ? ((ctx->dim + 1) / 2 + 1) / 2 * 2 : roundUp4(ctx->dim);
}

uint32_t keyTileDwords(const memb_hip_ctx* ctx, uint32_t wordsPerWave)
{
    return wordsPerWave ? (wordsPerWave * keyRowBytes(ctx) + 3) / 4 + 1 : 0;
}

uint32_t codebookDwords(const memb_hip_ctx* ctx)
{
    return ctx->fast ? 512u : 256u;
}

uint32_t trainedLdsBytes(const memb_hip_ctx* ctx, uint32_t waves, uint32_t wordsPerWave, bool withKeys)
{
    uint32_t perWave = wordsPerWave * ctx->slotDwords + (withKeys ? keyTileDwords(ctx, wordsPerWave) : 0);
    return 4u * (ctx->tableDwords + codebookDwords(ctx) + waves * perWave);
}

// Waves per block: as many resident wavefronts per CU as LDS allows (the
// decode is a chain of dependent LDS lookups, so occupancy is what hides it),
// larger blocks on ties (fewer copies of the lookup table).
TrainedGeometry chooseGeometry(
    const memb_hip_ctx* ctx, uint32_t wordsPerWave, size_t ld, size_t colOff, const float* out)
{
    TrainedGeometry best{};
    double bestWaves = -1;
    const uint32_t forcedWaves = envUint("MEMB_HIP_WAVES", 0);
    for (uint32_t waves : {8u, 4u, 2u, 1u}) {
        if (forcedWaves && waves != forcedWaves) {
            continue;
        }
        uint32_t ldsBytes = trainedLdsBytes(ctx, waves, wordsPerWave, true);
        if (ldsBytes > ctx->ldsLimit) {
            continue;
        }
        // LDS is handed out in 1 KiB steps of a 160 KiB pool; at most 32 waves per CU.
        uint32_t blocksPerCu = std::min<uint32_t>(ctx->ldsLimit / ((ldsBytes + 1023) / 1024 * 1024), 32 / waves);
        double residentWaves = blocksPerCu * waves;
        if (residentWaves > bestWaves) {
            bestWaves = residentWaves;
            best.waves = waves;
            best.ldsBytes = ldsBytes;
        }
    }
    const bool vec = (ctx->dim % 4 == 0) && (ld % 4 == 0) && (colOff % 4 == 0) &&
        (reinterpret_cast<uintptr_t>(out) % 16 == 0);
    if (!vec) {
        best.mode = OUT_SCALAR;
    } else if (ld == ctx->dim && colOff == 0) {
        best.mode = OUT_FLAT;
    } else {
        best.mode = OUT_VEC4;
    }
    return best;
}

template <bool HAS_SUB, int MODE, bool FAST>
hipError_t launchTrainedVariant(const TrainedParams& params, uint32_t blocks, uint32_t threads, uint32_t ldsBytes, hipStream_t stream)
{
    static thread_local int configuredDevice = -1;
    int device = 0;
    (void)hipGetDevice(&device);
    if (configuredDevice != device) {
        hipError_t status = hipFuncSetAttribute(
            reinterpret_cast<const void*>(&decode_trained<HAS_SUB, MODE, FAST>),
            hipFuncAttributeMaxDynamicSharedMemorySize,
            160 * 1024);
        if (status != hipSuccess) {
            return status;
        }
        configuredDevice = device;
    }
    hipLaunchKernelGGL(
        (decode_trained<HAS_SUB, MODE, FAST>), dim3(blocks), dim3(threads), ldsBytes, stream, params);
    return hipGetLastError();
}

template <bool HAS_SUB, int MODE, bool FAST>
hipError_t launchPersistentVariant(
    const memb_hip_ctx* ctx, const TrainedParams& params, uint32_t tileBlocks, uint32_t threads, uint32_t ldsBytes, hipStream_t stream)
{
    static thread_local int configuredDevice = -1;
    static thread_local int blocksPerCu = 0;
    static thread_local uint32_t configuredThreads = 0;
    static thread_local uint32_t configuredLds = 0;
    int device = 0;
    (void)hipGetDevice(&device);
    const void* kernel = reinterpret_cast<const void*>(&decode_trained_persistent<HAS_SUB, MODE, FAST>);
    if (configuredDevice != device || configuredThreads != threads || configuredLds != ldsBytes) {
        hipError_t status = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (status != hipSuccess) {
            return status;
        }
        status = hipOccupancyMaxActiveBlocksPerMultiprocessor(
            &blocksPerCu, decode_trained_persistent<HAS_SUB, MODE, FAST>, static_cast<int>(threads), ldsBytes);
        if (status != hipSuccess) {
            return status;
        }
        blocksPerCu = std::max(blocksPerCu, 1);
        configuredDevice = device;
        configuredThreads = threads;
        configuredLds = ldsBytes;
    }
    // as many blocks as are resident at once; each wavefront strides over the tiles
    const uint32_t resident = static_cast<uint32_t>(blocksPerCu) * ctx->cuCount;
    const uint32_t blocks = std::min(tileBlocks, resident);
    hipLaunchKernelGGL(
        (decode_trained_persistent<HAS_SUB, MODE, FAST>), dim3(blocks), dim3(threads), ldsBytes, stream, params);
    return hipGetLastError();
}

template <int MODE>
hipError_t launchPersistentMode(
    const memb_hip_ctx* ctx, const TrainedParams& params, uint32_t blocks, uint32_t threads, uint32_t ldsBytes, hipStream_t stream)
{
    if (ctx->fast) {
        return launchPersistentVariant<false, MODE, true>(ctx, params, blocks, threads, ldsBytes, stream);
    }
    return ctx->hostTable.hasSubTables
        ? launchPersistentVariant<true, MODE, false>(ctx, params, blocks, threads, ldsBytes, stream)
        : launchPersistentVariant<false, MODE, false>(ctx, params, blocks, threads, ldsBytes, stream);
}

template <int MODE>
hipError_t launchTrainedMode(
    const memb_hip_ctx* ctx, const TrainedParams& params, uint32_t blocks, uint32_t threads, uint32_t ldsBytes, hipStream_t stream)
{
    if (ctx->fast) {
        return launchTrainedVariant<false, MODE, true>(params, blocks, threads, ldsBytes, stream);
    }
    return ctx->hostTable.hasSubTables ? launchTrainedVariant<true, MODE, false>(params, blocks, threads, ldsBytes, stream)
                                       : launchTrainedVariant<false, MODE, false>(params, blocks, threads, ldsBytes, stream);
}

TrainedParams baseTrainedParams(const memb_hip_ctx* ctx)
{
    TrainedParams params{};
    params.streams = ctx->streams;
    params.streamStarts = ctx->streamStarts;
    params.segmentIndex = ctx->segmentIndex;
    params.table = ctx->table;
    params.codebook = ctx->codebook;
    params.nRows = ctx->nRows;
    params.tableDwords = ctx->tableDwords;
    params.codebookDwords = codebookDwords(ctx);
    params.rootBits = ctx->hostTable.rootBits;
    params.dim = ctx->dim;
    params.slotDwords = ctx->slotDwords;
    params.slotMagic = magicFor(ctx->slotDwords / 4, 64ull * (ctx->slotDwords / 4) * 5);
    params.debugFlags = envUint("MEMB_HIP_DEBUG", 0);
    return params;
}

struct Epilogue {
    uint32_t accumulate = 0;
    float divisor = 0.f;
};

int launchTrained(
    memb_hip_ctx* ctx, const uint32_t* rows, size_t n, float* out, size_t ld, size_t colOff, hipStream_t stream,
    const Epilogue& epilogue)
{
    const uint32_t wordsPerWave = WAVE / ctx->lanesPerWord;
    TrainedGeometry geometry = chooseGeometry(ctx, wordsPerWave, ld, colOff, out);
    if (!geometry.waves) {
        return fail(MEMB_HIP_ERR_INVALID, "decode tables and bitstream slots do not fit into LDS");
    }
    TrainedParams params = baseTrainedParams(ctx);
    params.rows = rows;
    params.out = out;
    params.n = n;
    params.ld = ld;
    params.colOff = colOff;
    params.accumulate = epilogue.accumulate;
    params.divisor = epilogue.divisor;
    params.lanesPerWord = ctx->lanesPerWord;
    params.laneMagic = magicFor(ctx->lanesPerWord, WAVE);
    params.wordsPerWave = wordsPerWave;
    params.segmentSymbols = ctx->segmentSymbols;
    params.keyRowBytes = keyRowBytes(ctx);
    params.keyTileDwords = keyTileDwords(ctx, wordsPerWave);
    params.pieceMagic = magicFor(ctx->dim / 4, uint64_t(wordsPerWave) * (ctx->dim / 4));

    // The kernels index LDS and the symbol tile from these numbers without further checks.
    {
        const uint32_t group = ctx->fast ? 8 : 4;
        const bool consistent = wordsPerWave >= 1 && wordsPerWave * params.lanesPerWord <= WAVE &&
            params.slotDwords >= 4 && params.slotDwords % 4 == 0 && params.segmentSymbols % group == 0 &&
            uint64_t(params.lanesPerWord) * params.segmentSymbols >= params.dim &&
            uint64_t(params.lanesPerWord - 1) * params.segmentSymbols < params.dim &&
            params.keyRowBytes * (ctx->fast ? 2u : 1u) >= params.dim &&
            uint64_t(params.keyTileDwords) * 4 >= uint64_t(wordsPerWave) * params.keyRowBytes &&
            (params.lanesPerWord == 1 || params.segmentIndex != nullptr) &&
            geometry.ldsBytes == trainedLdsBytes(ctx, geometry.waves, wordsPerWave, true) &&
            geometry.ldsBytes <= ctx->ldsLimit && ld >= colOff + params.dim;
        if (!consistent) {
            return fail(MEMB_HIP_ERR_INVALID, "internal error: inconsistent decode geometry");
        }
    }
    const size_t tiles = (n + wordsPerWave - 1) / wordsPerWave;
    const uint32_t blocks = static_cast<uint32_t>((tiles + geometry.waves - 1) / geometry.waves);
    const uint32_t threads = geometry.waves * WAVE;
    // The persistent kernel keeps one tile's bitstreams in registers; tiles wider than that
    // (long streams with few lanes per word) take the one-shot kernel.
    const uint32_t streamRounds = (wordsPerWave * (ctx->slotDwords / 4) + WAVE - 1) / WAVE;
    const bool persistent = streamRounds <= STREAM_REGISTERS && envUint("MEMB_HIP_PERSISTENT", 1) != 0;
    hipError_t status;
    switch (geometry.mode) {
        case OUT_FLAT:
            status = persistent ? launchPersistentMode<OUT_FLAT>(ctx, params, blocks, threads, geometry.ldsBytes, stream)
                                : launchTrainedMode<OUT_FLAT>(ctx, params, blocks, threads, geometry.ldsBytes, stream);
            break;
        case OUT_VEC4:
            status = persistent ? launchPersistentMode<OUT_VEC4>(ctx, params, blocks, threads, geometry.ldsBytes, stream)
                                : launchTrainedMode<OUT_VEC4>(ctx, params, blocks, threads, geometry.ldsBytes, stream);
            break;
        default:
            status = persistent ? launchPersistentMode<OUT_SCALAR>(ctx, params, blocks, threads, geometry.ldsBytes, stream)
                                : launchTrainedMode<OUT_SCALAR>(ctx, params, blocks, threads, geometry.ldsBytes, stream);
            break;
    }
    if (status != hipSuccess) {
        return fail(MEMB_HIP_ERR_DEVICE, std::string("decode_trained launch: ") + hipGetErrorString(status));
    }
    return MEMB_HIP_OK;
}

// One pass over every row with one lane per word: records the bit position at
// which each segment of each row starts (segmentIndex), so that lanesPerWord
// lanes can later decode a row side by side.
int buildSegmentIndex(memb_hip_ctx* ctx)
{
    if (ctx->lanesPerWord <= 1 || ctx->nRows == 0) {
        return MEMB_HIP_OK;
    }
    TrainedParams params = baseTrainedParams(ctx);
    params.segmentIndex = nullptr;
    params.segmentIndexOut = ctx->segmentIndex;
    params.rows = nullptr;
    params.n = ctx->nRows;
    params.lanesPerWord = 1;
    params.laneMagic = 0;
    params.wordsPerWave = WAVE;
    params.segmentSymbols = (ctx->dim + 7) / 8 * 8;
    params.keyRowBytes = 0;
    params.keyTileDwords = 0;
    params.indexLanes = ctx->lanesPerWord;
    params.indexSegmentSymbols = ctx->segmentSymbols;

    uint32_t waves = 4;
    while (waves > 1 && trainedLdsBytes(ctx, waves, WAVE, false) > ctx->ldsLimit) {
        waves /= 2;
    }
    const uint32_t ldsBytes = trainedLdsBytes(ctx, waves, WAVE, false);
    if (ldsBytes > ctx->ldsLimit) {
        return fail(MEMB_HIP_ERR_INVALID, "bitstream slots do not fit into LDS");
    }
    const size_t tiles = (ctx->nRows + WAVE - 1) / WAVE;
    const uint32_t blocks = static_cast<uint32_t>((tiles + waves - 1) / waves);
    hipError_t status = launchTrainedMode<OUT_INDEX>(ctx, params, blocks, waves * WAVE, ldsBytes, ctx->stream);
    if (status == hipSuccess) {
        status = hipStreamSynchronize(ctx->stream);
    }
    if (status != hipSuccess) {
        return fail(MEMB_HIP_ERR_DEVICE, std::string("segment index build: ") + hipGetErrorString(status));
    }
    return MEMB_HIP_OK;
}

constexpr uint32_t ROWWISE_THREADS = 256;

uint32_t rowwiseWordsPerBlock(uint32_t dim)
{
    // about 16 KiB of output per block (one batch of pieces per thread), at least one word
    return std::max<uint32_t>(1, std::min<uint32_t>(ROWWISE_MAX_WORDS, 4096 / std::max<uint32_t>(dim, 1)));
}

int launchUniform(
    memb_hip_ctx* ctx, const uint32_t* rows, size_t n, float* out, size_t ld, size_t colOff, hipStream_t stream,
    const Epilogue& epilogue)
{
    UniformParams params{};
    params.accumulate = epilogue.accumulate;
    params.divisor = epilogue.divisor;
    params.rows = rows;
    params.out = out;
    params.n = n;
    params.ld = ld;
    params.colOff = colOff;
    params.values = ctx->uniformValues;
    params.minMax = ctx->minMax;
    params.nRows = ctx->nRows;
    params.dim = ctx->dim;
    params.wordsPerBlock = rowwiseWordsPerBlock(ctx->dim);
    params.levels = ctx->levels;
    const bool vec = (ctx->dim % 4 == 0) && (ld % 4 == 0) && (colOff % 4 == 0) &&
        (reinterpret_cast<uintptr_t>(out) % 16 == 0);
    const uint32_t blocks = static_cast<uint32_t>((n + params.wordsPerBlock - 1) / params.wordsPerBlock);
    if (vec) {
        params.pieceMagic = magicFor(ctx->dim / 4, uint64_t(params.wordsPerBlock) * (ctx->dim / 4));
        hipLaunchKernelGGL(dequant_uniform<true>, dim3(blocks), dim3(ROWWISE_THREADS), 0, stream, params);
    } else {
        hipLaunchKernelGGL(dequant_uniform<false>, dim3(blocks), dim3(ROWWISE_THREADS), 0, stream, params);
    }
    hipError_t status = hipGetLastError();
    if (status != hipSuccess) {
        return fail(MEMB_HIP_ERR_DEVICE, std::string("dequant_uniform launch: ") + hipGetErrorString(status));
    }
    return MEMB_HIP_OK;
}

int launchFull(
    memb_hip_ctx* ctx, const uint32_t* rows, size_t n, float* out, size_t ld, size_t colOff, hipStream_t stream,
    const Epilogue& epilogue)
{
    FullParams params{};
    params.accumulate = epilogue.accumulate;
    params.divisor = epilogue.divisor;
    params.rows = rows;
    params.out = out;
    params.n = n;
    params.ld = ld;
    params.colOff = colOff;
    params.values = ctx->fullValues;
    params.nRows = ctx->nRows;
    params.dim = ctx->dim;
    params.wordsPerBlock = rowwiseWordsPerBlock(ctx->dim);
    const bool vec = (ctx->dim % 4 == 0) && (ld % 4 == 0) && (colOff % 4 == 0) &&
        (reinterpret_cast<uintptr_t>(out) % 16 == 0);
    const uint32_t blocks = static_cast<uint32_t>((n + params.wordsPerBlock - 1) / params.wordsPerBlock);
    if (vec) {
        params.pieceMagic = magicFor(ctx->dim / 4, uint64_t(params.wordsPerBlock) * (ctx->dim / 4));
        hipLaunchKernelGGL(gather_full<true>, dim3(blocks), dim3(ROWWISE_THREADS), 0, stream, params);
    } else {
        hipLaunchKernelGGL(gather_full<false>, dim3(blocks), dim3(ROWWISE_THREADS), 0, stream, params);
    }
    hipError_t status = hipGetLastError();
    if (status != hipSuccess) {
        return fail(MEMB_HIP_ERR_DEVICE, std::string("gather_full launch: ") + hipGetErrorString(status));
    }
    return MEMB_HIP_OK;
}

int launch(
    memb_hip_ctx* ctx, const uint32_t* rows, size_t n, float* out, size_t ld, size_t colOff, hipStream_t stream,
    const Epilogue& epilogue = Epilogue())
{
    if (n == 0) {
        return MEMB_HIP_OK;
    }
    if (n > (size_t(1) << 37)) {
        return fail(MEMB_HIP_ERR_INVALID, "batch too large");
    }
    switch (ctx->storage) {
        case memb::wire::Storage_Trained:
            return launchTrained(ctx, rows, n, out, ld, colOff, stream, epilogue);
        case memb::wire::Storage_Uniform:
            return launchUniform(ctx, rows, n, out, ld, colOff, stream, epilogue);
        case memb::wire::Storage_Full:
            return launchFull(ctx, rows, n, out, ld, colOff, stream, epilogue);
        default:
            return fail(MEMB_HIP_ERR_INVALID, "context has no storage");
    }
}

template <typename T>
int deviceAlloc(memb_hip_ctx* ctx, T** pointer, size_t bytes)
{
    void* raw = nullptr;
    HIP_TRY(hipMalloc(&raw, std::max<size_t>(bytes, 16)));
    ctx->allocations.push_back(raw);
    ctx->deviceBytes += std::max<size_t>(bytes, 16);
    *pointer = static_cast<T*>(raw);
    return MEMB_HIP_OK;
}

// Host -> device copy of a (possibly file-mapped) range. Pinning the mapped
// pages first lets the copy run at the PCIe rate; registration of a read-only
// mapping can be refused, in which case the plain copy is used.
int copyToDevice(void* dst, const void* src, size_t bytes)
{
    if (!bytes) {
        return MEMB_HIP_OK;
    }
    bool registered = false;
    if (bytes >= (8u << 20)) {
        hipError_t status = hipHostRegister(const_cast<void*>(src), bytes, hipHostRegisterReadOnly);
        if (status == hipSuccess) {
            registered = true;
        } else {
            (void)hipGetLastError();
        }
    }
    hipError_t status = hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice);
    if (registered) {
        (void)hipHostUnregister(const_cast<void*>(src));
    }
    if (status != hipSuccess) {
        return fail(MEMB_HIP_ERR_DEVICE, std::string("hipMemcpy to device: ") + hipGetErrorString(status));
    }
    return MEMB_HIP_OK;
}

int openDevice(memb_hip_ctx* ctx, int device)
{
    int count = 0;
    hipError_t status = hipGetDeviceCount(&count);
    if (status != hipSuccess || count == 0) {
        (void)hipGetLastError();
        return fail(MEMB_HIP_ERR_DEVICE, "no HIP device available");
    }
    if (device < 0 || device >= count) {
        return fail(MEMB_HIP_ERR_INVALID, "device index out of range");
    }
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t properties;
    HIP_TRY(hipGetDeviceProperties(&properties, device));
    ctx->device = device;
    ctx->cuCount = properties.multiProcessorCount;
    ctx->ldsLimit = static_cast<uint32_t>(std::min<size_t>(properties.sharedMemPerBlock, 160 * 1024));
    if (properties.maxSharedMemoryPerMultiProcessor >= 160 * 1024) {
        ctx->ldsLimit = 160 * 1024;
    }
    HIP_TRY(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    return MEMB_HIP_OK;
}

void destroy(memb_hip_ctx* ctx)
{
    if (!ctx) {
        return;
    }
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) {
        (void)hipStreamSynchronize(ctx->stream);
    }
    for (void* allocation : ctx->allocations) {
        (void)hipFree(allocation);
    }
    if (ctx->stagedRows) {
        (void)hipFree(ctx->stagedRows);
    }
    if (ctx->stagedOut) {
        (void)hipFree(ctx->stagedOut);
    }
    if (ctx->stream) {
        (void)hipStreamDestroy(ctx->stream);
    }
    delete ctx;
}

}  // namespace

extern "C" {

const char* memb_hip_last_error(void)
{
    return g_lastError.c_str();
}

int memb_hip_device_count(int* count)
{
    if (!count) {
        return fail(MEMB_HIP_ERR_INVALID, "count is null");
    }
    *count = 0;
    hipError_t status = hipGetDeviceCount(count);
    if (status != hipSuccess) {
        (void)hipGetLastError();
        *count = 0;
        return fail(MEMB_HIP_ERR_DEVICE, std::string("hipGetDeviceCount: ") + hipGetErrorString(status));
    }
    return MEMB_HIP_OK;
}

int memb_hip_ctx_create_trained(memb_hip_ctx** out, int device, const memb_hip_trained_desc* desc)
{
    if (!out || !desc) {
        return fail(MEMB_HIP_ERR_INVALID, "null argument");
    }
    *out = nullptr;
    if (desc->dim == 0 || desc->n_keys == 0 || desc->n_keys > 256 || desc->n_centroids > 255 ||
        (desc->n_rows && !desc->value_offsets) || (desc->packed_values_bytes && !desc->packed_values)) {
        return fail(MEMB_HIP_ERR_INVALID, "inconsistent trained storage description");
    }

    const bool verbose = envUint("MEMB_HIP_VERBOSE", 0) != 0;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double tStart = now();
    memb_hip_ctx* ctx = new memb_hip_ctx();
    ctx->storage = memb::wire::Storage_Trained;
    ctx->dim = desc->dim;
    ctx->nRows = desc->n_rows;
    int code = MEMB_HIP_OK;
    try {
        auto lengths = memb::codeLengthsFromSizeOffsets(
            desc->keys, desc->n_keys, desc->size_offsets, desc->n_size_offsets);
        for (const auto& info : lengths) {
            if (info.key >= desc->n_centroids) {
                throw std::runtime_error("Huffman symbol without a centroid");
            }
        }
        uint32_t limit = desc->max_direct_bits ? desc->max_direct_bits : envUint("MEMB_HIP_ROOT_BITS", 11);
        ctx->hostTable = memb::buildDecodeTable(lengths, std::min<uint32_t>(limit, 12));
    } catch (const std::exception& error) {
        delete ctx;
        return fail(MEMB_HIP_ERR_INVALID, error.what());
    }

    // Per-row stream length: streams are laid out back to back, so a stream ends
    // where the next one (in storage order) begins. Every start is marked in a
    // bitmap over the byte positions; each row then scans forward to the next
    // mark. Both passes run on a few host threads.
    {
        for (uint64_t r = 0; r < desc->n_rows; ++r) {
            if (desc->value_offsets[r] > desc->packed_values_bytes) {
                delete ctx;
                return fail(MEMB_HIP_ERR_INVALID, "value offset beyond packed values");
            }
        }
        const uint64_t totalBytes = desc->packed_values_bytes;
        const size_t bitmapWords = static_cast<size_t>(totalBytes / 64 + 2);
        std::vector<std::atomic<uint64_t>> marks(bitmapWords);
        for (auto& word : marks) {
            word.store(0, std::memory_order_relaxed);
        }
        marks[totalBytes / 64].fetch_or(uint64_t(1) << (totalBytes % 64), std::memory_order_relaxed);   // end sentinel
        const uint64_t bound =
            (static_cast<uint64_t>(desc->dim) * std::max<uint32_t>(ctx->hostTable.maxCodeBits, 1) + 7) / 8;
        ctx->streamBytes.assign(desc->n_rows, 0);
        const size_t threads = std::max<size_t>(1, std::min<size_t>({std::thread::hardware_concurrency(), size_t(16), size_t(desc->n_rows / 65536 + 1)}));
        const uint64_t perThread = (desc->n_rows + threads - 1) / threads;
        auto parallel = [&](auto body) {
            std::vector<std::thread> pool;
            for (size_t t = 1; t < threads; ++t) {
                pool.emplace_back(body, t * perThread, std::min<uint64_t>(desc->n_rows, (t + 1) * perThread));
            }
            body(uint64_t(0), std::min<uint64_t>(desc->n_rows, perThread));
            for (auto& thread : pool) {
                thread.join();
            }
        };
        parallel([&](uint64_t first, uint64_t last) {
            for (uint64_t r = first; r < last; ++r) {
                const uint64_t offset = desc->value_offsets[r];
                marks[offset / 64].fetch_or(uint64_t(1) << (offset % 64), std::memory_order_relaxed);
            }
        });
        std::vector<uint32_t> threadMax(threads, 0);
        parallel([&](uint64_t first, uint64_t last) {
            uint32_t longest = 0;
            for (uint64_t r = first; r < last; ++r) {
                const uint64_t offset = desc->value_offsets[r];
                uint64_t bytes = 0;
                if (offset < totalBytes) {
                    // next mark strictly after `offset` (a stream never holds more than dim codes of the longest length)
                    uint64_t position = offset + 1;
                    size_t word = position / 64;
                    uint64_t bits = marks[word].load(std::memory_order_relaxed) & (~uint64_t(0) << (position % 64));
                    while (!bits && (word + 1) * 64 <= offset + bound + 64) {
                        bits = marks[++word].load(std::memory_order_relaxed);
                    }
                    const uint64_t next = bits ? word * 64 + static_cast<uint64_t>(__builtin_ctzll(bits)) : offset + bound;
                    bytes = std::min(next - offset, bound);
                }
                ctx->streamBytes[r] = static_cast<uint32_t>(bytes);
                longest = std::max(longest, ctx->streamBytes[r]);
            }
            threadMax[first / std::max<uint64_t>(perThread, 1)] = longest;
        });
        ctx->maxStreamBytes = *std::max_element(threadMax.begin(), threadMax.end());
    }
    // Slot: stream plus the 12-byte window the decoder reads at its last
    // position; whole 16-byte pieces, an odd number of them so that equal
    // positions in consecutive slots fall into different LDS banks.
    ctx->slotDwords = (((ctx->maxStreamBytes + 12 + 15) / 16) | 1u) * 4;
    ctx->fast = desc->n_centroids <= 16 && ctx->hostTable.maxCodeBits <= 8 && !ctx->hostTable.hasSubTables &&
        !envUint("MEMB_HIP_NO_FAST", 0);
    ctx->tableDwords = static_cast<uint32_t>((2 * ctx->hostTable.entries.size() + 3) / 4 * 4);

    // Lanes per word (G) and symbols per lane (S). The side index stores 16-bit
    // bit offsets, so rows longer than 65535 bits keep one lane per word.
    {
        uint32_t lanes = envUint("MEMB_HIP_LANES", 8);
        lanes = std::max<uint32_t>(1, std::min<uint32_t>(lanes, WAVE));
        if (uint64_t(desc->dim) * std::max<uint32_t>(ctx->hostTable.maxCodeBits, 1) >= 65536 || desc->dim < 8) {
            lanes = 1;
        }
        const uint32_t group = ctx->fast ? 8 : 4;
        ctx->segmentSymbols = std::max<uint32_t>(group, ((desc->dim + lanes - 1) / lanes + group - 1) / group * group);
        ctx->lanesPerWord = (desc->dim + ctx->segmentSymbols - 1) / ctx->segmentSymbols;
    }

    const double tSorted = now();
    // Re-packed layout: row r's stream occupies ceil(bytes / 16) pieces from streamStarts[r].
    std::vector<uint32_t> streamStarts(desc->n_rows + 1, 0);
    {
        uint64_t next = 0;
        for (uint64_t r = 0; r < desc->n_rows; ++r) {
            streamStarts[r] = static_cast<uint32_t>(next);
            next += (ctx->streamBytes[r] + 15) / 16;
        }
        if (next + ctx->slotDwords / 4 + 1 >= (1ull << 32)) {
            delete ctx;
            return fail(MEMB_HIP_ERR_INVALID, "bitstreams too large");
        }
        streamStarts[desc->n_rows] = static_cast<uint32_t>(next);
    }
    const size_t streamPieces = size_t(streamStarts[desc->n_rows]) + ctx->slotDwords / 4 + 1;   // + guard of one slot

    code = openDevice(ctx, device);
    uint8_t* filePacked = nullptr;      // temporary device copies of the file's arrays
    uint32_t* fileOffsets = nullptr;
    if (code == MEMB_HIP_OK) {
        code = deviceAlloc(ctx, &ctx->streams, streamPieces * 16);
    }
    if (code == MEMB_HIP_OK) {
        hipError_t status = hipMemset(ctx->streams, 0, streamPieces * 16);
        if (status != hipSuccess) {
            code = fail(MEMB_HIP_ERR_DEVICE, std::string("hipMemset: ") + hipGetErrorString(status));
        }
    }
    if (code == MEMB_HIP_OK) {
        code = deviceAlloc(ctx, &ctx->streamStarts, streamStarts.size() * 4);
    }
    if (code == MEMB_HIP_OK) {
        code = copyToDevice(ctx->streamStarts, streamStarts.data(), streamStarts.size() * 4);
    }
    if (code == MEMB_HIP_OK && desc->n_rows) {
        hipError_t status = hipMalloc(reinterpret_cast<void**>(&filePacked), std::max<size_t>(desc->packed_values_bytes, 16));
        if (status == hipSuccess) {
            status = hipMalloc(reinterpret_cast<void**>(&fileOffsets), desc->n_rows * 4);
        }
        if (status != hipSuccess) {
            code = fail(MEMB_HIP_ERR_DEVICE, std::string("hipMalloc: ") + hipGetErrorString(status));
        }
        if (code == MEMB_HIP_OK) {
            code = copyToDevice(filePacked, desc->packed_values, desc->packed_values_bytes);
        }
        if (code == MEMB_HIP_OK) {
            code = copyToDevice(fileOffsets, desc->value_offsets, desc->n_rows * 4);
        }
        if (code == MEMB_HIP_OK) {
            const uint32_t threads = 256;
            const uint64_t blocks = (desc->n_rows * WAVE + threads - 1) / threads;
            hipLaunchKernelGGL(
                repack_streams, dim3(static_cast<uint32_t>(blocks)), dim3(threads), 0, ctx->stream, filePacked,
                desc->packed_values_bytes, fileOffsets, ctx->streamStarts, desc->n_rows, ctx->streams);
            hipError_t launched = hipGetLastError();
            if (launched == hipSuccess) {
                launched = hipStreamSynchronize(ctx->stream);
            }
            if (launched != hipSuccess) {
                code = fail(MEMB_HIP_ERR_DEVICE, std::string("repack_streams: ") + hipGetErrorString(launched));
            }
        }
        if (filePacked) {
            (void)hipFree(filePacked);
        }
        if (fileOffsets) {
            (void)hipFree(fileOffsets);
        }
    }
    const double tRepacked = now();
    if (code == MEMB_HIP_OK) {
        code = deviceAlloc(ctx, &ctx->table, size_t(ctx->tableDwords) * 4);
    }
    if (code == MEMB_HIP_OK) {
        // device form of the table: {length or pointer, symbol replicated per byte / nibble}
        std::vector<uint32_t> expanded(ctx->tableDwords, 0);
        for (size_t i = 0; i < ctx->hostTable.entries.size(); ++i) {
            const uint32_t entry = ctx->hostTable.entries[i];
            if (entry & memb::TABLE_POINTER_FLAG) {
                expanded[2 * i] = entry;
            } else {
                const uint32_t key = (entry >> 8) & 0xff;
                expanded[2 * i] = entry & 0xff;
                expanded[2 * i + 1] = ctx->fast ? key * 0x11111111u : key * 0x01010101u;
            }
        }
        code = copyToDevice(ctx->table, expanded.data(), expanded.size() * 4);
    }
    if (code == MEMB_HIP_OK) {
        code = deviceAlloc(ctx, &ctx->codebook, 512 * 4);
    }
    if (code == MEMB_HIP_OK) {
        std::vector<float> centroids(256, 0.f);
        std::copy(desc->centroids, desc->centroids + desc->n_centroids, centroids.begin());
        centroids[ZERO_KEY] = 0.f;
        std::vector<float> codebook(512, 0.f);
        if (ctx->fast) {
            // pair b = {centroid[low nibble], centroid[high nibble]}: two symbols per LDS read
            for (uint32_t b = 0; b < 256; ++b) {
                codebook[2 * b] = centroids[b & 15];
                codebook[2 * b + 1] = centroids[b >> 4];
            }
        } else {
            std::copy(centroids.begin(), centroids.end(), codebook.begin());
        }
        code = copyToDevice(ctx->codebook, codebook.data(), 512 * 4);
    }
    if (code == MEMB_HIP_OK) {
        TrainedGeometry geometry = chooseGeometry(ctx, WAVE / ctx->lanesPerWord, ctx->dim, 0, nullptr);
        if (!geometry.waves) {
            code = fail(MEMB_HIP_ERR_INVALID, "decode tables and bitstream slots do not fit into LDS");
        }
    }
    if (code == MEMB_HIP_OK && ctx->lanesPerWord > 1) {
        code = deviceAlloc(ctx, &ctx->segmentIndex, size_t(desc->n_rows) * (ctx->lanesPerWord - 1) * sizeof(uint16_t));
    }
    if (code == MEMB_HIP_OK) {
        code = buildSegmentIndex(ctx);
    }
    if (verbose) {
        std::fprintf(stderr, "memb_hip: stage trained rows=%llu: host lengths %.3fs, device open + copy + repack %.3fs, tables + index %.3fs\n",
                     static_cast<unsigned long long>(desc->n_rows), tSorted - tStart, tRepacked - tSorted, now() - tRepacked);
    }
    if (code != MEMB_HIP_OK) {
        std::string message = g_lastError;
        destroy(ctx);
        g_lastError = message;
        return code;
    }
    *out = ctx;
    return MEMB_HIP_OK;
}

int memb_hip_ctx_create_uniform(memb_hip_ctx** out, int device, const memb_hip_uniform_desc* desc)
{
    if (!out || !desc) {
        return fail(MEMB_HIP_ERR_INVALID, "null argument");
    }
    *out = nullptr;
    if (desc->dim == 0 || (desc->n_rows && !desc->rows)) {
        return fail(MEMB_HIP_ERR_INVALID, "inconsistent uniform storage description");
    }
    memb_hip_ctx* ctx = new memb_hip_ctx();
    ctx->storage = memb::wire::Storage_Uniform;
    ctx->dim = desc->dim;
    ctx->nRows = desc->n_rows;
    ctx->levels = static_cast<float>(desc->quantization_levels);

    // HBM layout: dense [n_rows][dim] bytes plus one {min, max} pair per row.
    // The file scatters each row in its own table; rows shorter than dim are
    // zero padded (the reference writes only values->size() outputs there).
    std::vector<uint8_t> values(size_t(desc->n_rows) * desc->dim, 0);
    std::vector<float2> minMax(desc->n_rows);
    for (uint64_t r = 0; r < desc->n_rows; ++r) {
        const memb_hip_uniform_row& row = desc->rows[r];
        size_t count = std::min<size_t>(row.n_values, desc->dim);
        if (count) {
            std::memcpy(values.data() + r * desc->dim, row.values, count);
        }
        minMax[r] = make_float2(row.min_value, row.max_value);
    }

    int code = openDevice(ctx, device);
    if (code == MEMB_HIP_OK) {
        code = deviceAlloc(ctx, &ctx->uniformValues, values.size() + 16);
    }
    if (code == MEMB_HIP_OK) {
        code = copyToDevice(ctx->uniformValues, values.data(), values.size());
    }
    if (code == MEMB_HIP_OK) {
        code = deviceAlloc(ctx, &ctx->minMax, minMax.size() * sizeof(float2));
    }
    if (code == MEMB_HIP_OK) {
        code = copyToDevice(ctx->minMax, minMax.data(), minMax.size() * sizeof(float2));
    }
    if (code != MEMB_HIP_OK) {
        std::string message = g_lastError;
        destroy(ctx);
        g_lastError = message;
        return code;
    }
    *out = ctx;
    return MEMB_HIP_OK;
}

int memb_hip_ctx_create_full(memb_hip_ctx** out, int device, const memb_hip_full_desc* desc)
{
    if (!out || !desc) {
        return fail(MEMB_HIP_ERR_INVALID, "null argument");
    }
    *out = nullptr;
    if (desc->dim == 0 || (desc->n_rows && !desc->rows)) {
        return fail(MEMB_HIP_ERR_INVALID, "inconsistent full storage description");
    }
    memb_hip_ctx* ctx = new memb_hip_ctx();
    ctx->storage = memb::wire::Storage_Full;
    ctx->dim = desc->dim;
    ctx->nRows = desc->n_rows;

    std::vector<float> values(size_t(desc->n_rows) * desc->dim, 0.f);
    for (uint64_t r = 0; r < desc->n_rows; ++r) {
        const memb_hip_full_row& row = desc->rows[r];
        size_t count = std::min<size_t>(row.n_values, desc->dim);
        if (count) {
            std::memcpy(values.data() + r * desc->dim, row.values, count * sizeof(float));
        }
    }
    int code = openDevice(ctx, device);
    if (code == MEMB_HIP_OK) {
        code = deviceAlloc(ctx, &ctx->fullValues, values.size() * sizeof(float) + 16);
    }
    if (code == MEMB_HIP_OK) {
        code = copyToDevice(ctx->fullValues, values.data(), values.size() * sizeof(float));
    }
    if (code != MEMB_HIP_OK) {
        std::string message = g_lastError;
        destroy(ctx);
        g_lastError = message;
        return code;
    }
    *out = ctx;
    return MEMB_HIP_OK;
}

void memb_hip_ctx_destroy(memb_hip_ctx* ctx)
{
    destroy(ctx);
}

int memb_hip_ctx_get_info(const memb_hip_ctx* ctx, memb_hip_ctx_info* info)
{
    if (!ctx || !info) {
        return fail(MEMB_HIP_ERR_INVALID, "null argument");
    }
    std::memset(info, 0, sizeof(*info));
    info->device = ctx->device;
    info->storage = ctx->storage;
    info->dim = ctx->dim;
    info->n_rows = ctx->nRows;
    info->device_bytes = ctx->deviceBytes;
    if (ctx->storage == memb::wire::Storage_Trained) {
        info->root_bits = ctx->hostTable.rootBits;
        info->max_code_bits = ctx->hostTable.maxCodeBits;
        info->table_entries = static_cast<uint32_t>(ctx->hostTable.entries.size());
        info->max_stream_bytes = ctx->maxStreamBytes;
        TrainedGeometry geometry = chooseGeometry(ctx, WAVE / ctx->lanesPerWord, ctx->dim, 0, nullptr);
        info->waves_per_block = geometry.waves;
        info->lanes_per_word = ctx->lanesPerWord;
        info->segment_symbols = ctx->segmentSymbols;
        info->lds_bytes_per_block = geometry.ldsBytes;
    } else {
        info->waves_per_block = ROWWISE_THREADS / WAVE;
    }
    return MEMB_HIP_OK;
}

int memb_hip_decode_rows_device(
    memb_hip_ctx* ctx, const uint32_t* rows, size_t n, float* out, size_t ld, size_t col_off, void* stream)
{
    if (!ctx || (n && (!rows || !out))) {
        return fail(MEMB_HIP_ERR_INVALID, "null argument");
    }
    if (ld < col_off + ctx->dim) {
        return fail(MEMB_HIP_ERR_INVALID, "ld must be at least col_off + dim");
    }
    HIP_TRY(hipSetDevice(ctx->device));
    return launch(ctx, rows, n, out, ld, col_off, static_cast<hipStream_t>(stream));
}

int memb_hip_decode_rows_device_ex(
    memb_hip_ctx* ctx, const uint32_t* rows, size_t n, float* out, size_t ld, size_t col_off, void* stream,
    uint32_t flags, float divisor)
{
    if (!ctx || (n && (!rows || !out))) {
        return fail(MEMB_HIP_ERR_INVALID, "null argument");
    }
    if (ld < col_off + ctx->dim) {
        return fail(MEMB_HIP_ERR_INVALID, "ld must be at least col_off + dim");
    }
    if ((flags & ~uint32_t(MEMB_HIP_ACCUMULATE)) || !(divisor == divisor)) {
        return fail(MEMB_HIP_ERR_INVALID, "unknown flags or NaN divisor");
    }
    HIP_TRY(hipSetDevice(ctx->device));
    Epilogue epilogue;
    epilogue.accumulate = (flags & MEMB_HIP_ACCUMULATE) ? 1u : 0u;
    epilogue.divisor = divisor;
    return launch(ctx, rows, n, out, ld, col_off, static_cast<hipStream_t>(stream), epilogue);
}

int memb_hip_decode_rows(
    memb_hip_ctx* ctx, const uint32_t* rows, size_t n, float* out, size_t ld, size_t col_off)
{
    if (!ctx || (n && (!rows || !out))) {
        return fail(MEMB_HIP_ERR_INVALID, "null argument");
    }
    if (ld < col_off + ctx->dim) {
        return fail(MEMB_HIP_ERR_INVALID, "ld must be at least col_off + dim");
    }
    if (n == 0) {
        return MEMB_HIP_OK;
    }
    std::lock_guard<std::mutex> lock(ctx->mutex);
    HIP_TRY(hipSetDevice(ctx->device));

    // Device staging holds dense [words][dim] rows; batches larger than the
    // staging area are processed in slices.
    const size_t dim = ctx->dim;
    const size_t sliceWords = std::max<size_t>(1, std::min<size_t>(n, (size_t(512) << 20) / (dim * sizeof(float))));
    if (ctx->stagedCapacity < sliceWords) {
        if (ctx->stagedRows) {
            (void)hipFree(ctx->stagedRows);
            ctx->stagedRows = nullptr;
        }
        if (ctx->stagedOut) {
            (void)hipFree(ctx->stagedOut);
            ctx->stagedOut = nullptr;
        }
        ctx->stagedCapacity = 0;
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&ctx->stagedRows), sliceWords * sizeof(uint32_t)));
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&ctx->stagedOut), sliceWords * dim * sizeof(float)));
        ctx->stagedCapacity = sliceWords;
    }
    // Pin the caller's output range for the duration of the call: a fresh host
    // buffer is otherwise faulted in page by page behind the DMA engine (measured
    // 10-17 GB/s against 53-57 GB/s pinned; registering 2.6 GB of untouched pages
    // takes 0.11 s). Only for buffers of 32 MiB and more: those are mappings of
    // their own (glibc's mmap threshold never exceeds 32 MiB), whereas smaller
    // ones share heap pages with unrelated live data that must not be pinned and
    // unpinned under it. Opt-in (MEMB_HIP_PIN_OUTPUT=1): one full GPU test run
    // aborted while registration was on by default for >= 1 MiB buffers and the
    // cause could not be established.
    const bool verbose = envUint("MEMB_HIP_VERBOSE", 0) != 0;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = now();
    char* pinBase = reinterpret_cast<char*>(out + col_off);
    const size_t pinBytes = ((n - 1) * ld + dim) * sizeof(float);
    bool pinned = false;
    if (pinBytes >= (size_t(32) << 20) && envUint("MEMB_HIP_PIN_OUTPUT", 0)) {
        hipError_t registered = hipHostRegister(pinBase, pinBytes, hipHostRegisterDefault);
        if (registered == hipSuccess) {
            pinned = true;
        } else {
            (void)hipGetLastError();
            if (envUint("MEMB_HIP_VERBOSE", 0)) {
                std::fprintf(stderr, "memb_hip: hipHostRegister(%zu bytes): %s\n", pinBytes, hipGetErrorString(registered));
            }
        }
    }
    const double t1 = now();
    int result = MEMB_HIP_OK;
    for (size_t start = 0; start < n && result == MEMB_HIP_OK; start += sliceWords) {
        const size_t words = std::min(sliceWords, n - start);
        hipError_t status = hipMemcpyAsync(
            ctx->stagedRows, rows + start, words * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream);
        if (status == hipSuccess) {
            result = launch(ctx, ctx->stagedRows, words, ctx->stagedOut, dim, 0, ctx->stream);
            if (result != MEMB_HIP_OK) {
                break;
            }
            if (ld == dim) {
                status = hipMemcpyAsync(
                    out + start * ld, ctx->stagedOut, words * dim * sizeof(float), hipMemcpyDeviceToHost, ctx->stream);
            } else {
                status = hipMemcpy2DAsync(
                    out + start * ld + col_off,
                    ld * sizeof(float),
                    ctx->stagedOut,
                    dim * sizeof(float),
                    dim * sizeof(float),
                    words,
                    hipMemcpyDeviceToHost,
                    ctx->stream);
            }
        }
        if (status == hipSuccess) {
            status = hipStreamSynchronize(ctx->stream);
        }
        if (status != hipSuccess) {
            result = fail(MEMB_HIP_ERR_DEVICE, std::string("batch copy: ") + hipGetErrorString(status));
        }
    }
    const double t2 = now();
    if (pinned) {
        (void)hipHostUnregister(pinBase);
    }
    if (verbose) {
        std::fprintf(stderr, "memb_hip: decode_rows n=%zu pinned=%d register %.4fs copy+kernel %.4fs unregister %.4fs\n",
                     n, int(pinned), t1 - t0, t2 - t1, now() - t2);
    }
    return result;
}

int memb_hip_sync(memb_hip_ctx* ctx)
{
    if (!ctx) {
        return fail(MEMB_HIP_ERR_INVALID, "null argument");
    }
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return MEMB_HIP_OK;
}

int memb_hip_algorithmic_bytes(const memb_hip_ctx* ctx, const uint32_t* rows, size_t n, uint64_t* bytes)
{
    if (!ctx || !bytes || (n && !rows)) {
        return fail(MEMB_HIP_ERR_INVALID, "null argument");
    }
    uint64_t total = 0;
    const uint64_t rowBytes = 4ull * ctx->dim;
    for (size_t i = 0; i < n; ++i) {
        total += 4 + rowBytes;
        if (rows[i] >= ctx->nRows) {
            continue;
        }
        switch (ctx->storage) {
            case memb::wire::Storage_Trained:
                total += 4 + ctx->streamBytes[rows[i]];
                break;
            case memb::wire::Storage_Uniform:
                total += 12 + ctx->dim;
                break;
            case memb::wire::Storage_Full:
                total += 4 + rowBytes;
                break;
            default:
                break;
        }
    }
    *bytes = total;
    return MEMB_HIP_OK;
}

}  // extern "C"

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
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
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
    const uint8_t* packed;
    const uint32_t* valueOffsets;
    const uint16_t* segmentIndex;   // [nRows][lanesPerWord - 1] bit offsets of segments 1.. from the stream start
    uint16_t* segmentIndexOut;      // OUT_INDEX: index being built, [nRows][indexLanes - 1]
    const uint32_t* table;
    const float* centroids;
    unsigned long long nRows;
    uint32_t tableDwords;     // multiple of 4
    uint32_t rootBits;
    uint32_t dim;
    uint32_t slotDwords;      // LDS dwords reserved per bitstream, multiple of 4
    uint32_t slotMagic;       // fastDivide magic for slotDwords / 4
    uint32_t lanesPerWord;    // G
    uint32_t laneMagic;       // fastDivide magic for G
    uint32_t wordsPerWave;    // 64 / G
    uint32_t segmentSymbols;  // S, multiple of 4
    uint32_t keyStride;       // dwords per word in the symbol tile = ceil(dim / 4)
    uint32_t keyMagic;        // fastDivide magic for keyStride (vector output)
    uint32_t indexLanes;      // OUT_INDEX: lanes per word of the index being built
    uint32_t indexSegmentSymbols;
};

enum OutputMode { OUT_SCALAR = 0, OUT_VEC4 = 1, OUT_FLAT = 2, OUT_INDEX = 3 };

// 16-byte load from an address that is only 4-byte aligned (bitstreams start
// on arbitrary bytes; the staging copy starts at the enclosing dword).
typedef uint4 __attribute__((aligned(4))) uint4_align4;

// q / d with a host-computed magic = ceil(2^32 / d) (exact while q * d < 2^32);
// magic == 0 means "no magic" (d == 1, or the range is too large): plain division.
__device__ __forceinline__ uint32_t fastDivide(uint32_t q, uint32_t magic, uint32_t d)
{
    return magic ? __umulhi(q, magic) : q / d;
}

__device__ __forceinline__ uint32_t byteSwap(uint32_t v)
{
    return __builtin_bswap32(v);
}

// Orders this wave's LDS writes before its later LDS reads (and vice versa).
// LDS operations of one wave execute in order; the fence makes the compiler
// wait for them and keeps it from moving accesses across.
__device__ __forceinline__ void waveLdsFence()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
}

template <bool HAS_SUB, int MODE>
__global__ void decode_trained(TrainedParams p)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];

    const uint32_t lane = threadIdx.x & (WAVE - 1);
    const uint32_t wave = threadIdx.x / WAVE;
    const uint32_t wavesPerBlock = blockDim.x / WAVE;

    uint32_t* tableLds = lds;
    float* centroidLds = reinterpret_cast<float*>(lds + p.tableDwords);
    const uint32_t perWave = p.wordsPerWave * (p.slotDwords + p.keyStride);
    uint32_t* slots = lds + p.tableDwords + 256 + wave * perWave;
    uint32_t* keyTile = slots + p.wordsPerWave * p.slotDwords;

    for (uint32_t i = threadIdx.x; i < p.tableDwords / 4; i += blockDim.x) {
        reinterpret_cast<uint4*>(tableLds)[i] = reinterpret_cast<const uint4*>(p.table)[i];
    }
    if (MODE != OUT_INDEX) {
        for (uint32_t i = threadIdx.x; i < 256; i += blockDim.x) {
            centroidLds[i] = p.centroids[i];
        }
    }
    __syncthreads();

    const unsigned long long tileBase =
        (static_cast<unsigned long long>(blockIdx.x) * wavesPerBlock + wave) * p.wordsPerWave;
    if (tileBase >= p.n) {
        return;
    }
    const uint32_t tileWords =
        static_cast<uint32_t>(min(static_cast<unsigned long long>(p.wordsPerWave), p.n - tileBase));

    // lane -> (word of the tile, segment of the word)
    // (64 % G spare lanes at the top get word == wordsPerWave: they decode word 0's slot and store nothing)
    const uint32_t laneWord = fastDivide(lane, p.laneMagic, p.lanesPerWord);
    const uint32_t segment = lane - laneWord * p.lanesPerWord;
    const bool spare = laneWord >= p.wordsPerWave;
    const uint32_t word = spare ? 0 : laneWord;

    uint32_t row = MISSING;
    if (!spare && word < tileWords) {
        row = p.rows ? p.rows[tileBase + word] : static_cast<uint32_t>(tileBase + word);
    }
    const bool present = row < p.nRows;
    const uint32_t offset = present ? p.valueOffsets[row] : 0;
    uint32_t segmentBits = 0;
    if (present && segment > 0) {
        segmentBits = p.segmentIndex[static_cast<unsigned long long>(row) * (p.lanesPerWord - 1) + segment - 1];
    }

    // Stage the tile's bitstreams: piece q = (word, 16-byte piece) -> one lane.
    // All loads of a batch are issued before the first one is waited for.
    // Absent words read the start of the array (always mapped: the guard is a
    // slot long) and never emit what they decode.
    {
        const uint32_t piecesPerWord = p.slotDwords / 4;
        const uint32_t totalPieces = p.wordsPerWave * piecesPerWord;
        const uint32_t sourceOffset = present ? (offset & ~3u) : 0u;
        constexpr int BATCH = 4;
        for (uint32_t q0 = 0; q0 < totalPieces; q0 += WAVE * BATCH) {
            uint4 v[BATCH];
            uint32_t destination[BATCH];
#pragma unroll
            for (int b = 0; b < BATCH; ++b) {
                const uint32_t q = q0 + WAVE * b + lane;
                const uint32_t w = min(fastDivide(q, p.slotMagic, piecesPerWord), p.wordsPerWave - 1);
                const uint32_t piece = q - w * piecesPerWord;
                const uint32_t wordOffset = __shfl(sourceOffset, w * p.lanesPerWord);
                destination[b] = w * p.slotDwords + 4 * piece;
                if (q < totalPieces) {
                    v[b] = *reinterpret_cast<const uint4_align4*>(p.packed + wordOffset + 16u * piece);
                }
            }
#pragma unroll
            for (int b = 0; b < BATCH; ++b) {
                const uint32_t q = q0 + WAVE * b + lane;
                if (q < totalPieces) {
                    uint4 t = v[b];
                    t.x = byteSwap(t.x);
                    t.y = byteSwap(t.y);
                    t.z = byteSwap(t.z);
                    t.w = byteSwap(t.w);
                    *reinterpret_cast<uint4*>(slots + destination[b]) = t;
                }
            }
        }
    }
    waveLdsFence();

    const uint32_t* slot = slots + word * p.slotDwords;
    uint32_t* keyRow = keyTile + word * p.keyStride;
    const uint32_t lastWindow = p.slotDwords - 3;
    const uint32_t rootShift = 32 - p.rootBits;
    const uint32_t startBit = (offset & 3u) * 8;
    uint32_t bitPos = startBit + segmentBits;
    const uint32_t firstColumn = segment * (p.segmentSymbols / 4);
    uint32_t nextIndexSymbol = p.indexSegmentSymbols;
    uint32_t indexSlot = 0;

    for (uint32_t j = 0; j < p.segmentSymbols; j += 4) {
        if (MODE == OUT_INDEX) {
            // one lane per word here; record where every indexSegmentSymbols-th symbol starts
            if (j == nextIndexSymbol) {
                if (present && indexSlot + 1 < p.indexLanes) {
                    p.segmentIndexOut[static_cast<unsigned long long>(row) * (p.indexLanes - 1) + indexSlot] =
                        static_cast<uint16_t>(bitPos - startBit);
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
        unsigned long long window =
            (((static_cast<unsigned long long>(w0) << 32) | w1) << shift) |
            (static_cast<unsigned long long>(w2) >> (32 - shift));
        uint32_t keys = 0;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            uint32_t entry = tableLds[static_cast<uint32_t>(window >> 32) >> rootShift];
            if (HAS_SUB) {
                if (entry & memb::TABLE_POINTER_FLAG) {
                    const uint32_t subBits = entry & 0xff;
                    const uint32_t base = (entry & ~memb::TABLE_POINTER_FLAG) >> 8;
                    const uint32_t subIndex =
                        static_cast<uint32_t>((window << p.rootBits) >> 32) >> (32 - subBits);
                    entry = tableLds[base + subIndex];
                }
            }
            const uint32_t length = entry & 0xff;
            window <<= length;
            bitPos += length;
            keys |= ((entry >> 8) & 0xff) << (8 * s);
        }
        if (MODE != OUT_INDEX) {
            const uint32_t column = firstColumn + (j >> 2);
            if (column < p.keyStride && !spare) {
                keyRow[column] = present ? keys : 0xFFFFFFFFu;
            }
        }
    }
    if (MODE == OUT_INDEX) {
        return;
    }
    waveLdsFence();

    if (MODE == OUT_FLAT) {
        // ld == dim: the symbol tile [tileWords][dim / 4] and the output tile are both linear.
        const uint32_t pieces = tileWords * p.keyStride;
        float4* dst = reinterpret_cast<float4*>(p.out + tileBase * p.ld);
        for (uint32_t q = lane; q < pieces; q += WAVE) {
            const uint32_t k = keyTile[q];
            float4 f;
            f.x = centroidLds[k & 0xff];
            f.y = centroidLds[(k >> 8) & 0xff];
            f.z = centroidLds[(k >> 16) & 0xff];
            f.w = centroidLds[k >> 24];
            dst[q] = f;
        }
    } else if (MODE == OUT_VEC4) {
        const uint32_t pieces = tileWords * p.keyStride;
        for (uint32_t q = lane; q < pieces; q += WAVE) {
            const uint32_t w = fastDivide(q, p.keyMagic, p.keyStride);
            const uint32_t c = q - w * p.keyStride;
            const uint32_t k = keyTile[q];
            float4 f;
            f.x = centroidLds[k & 0xff];
            f.y = centroidLds[(k >> 8) & 0xff];
            f.z = centroidLds[(k >> 16) & 0xff];
            f.w = centroidLds[k >> 24];
            float* dst = p.out + (tileBase + w) * p.ld + p.colOff + 4 * c;
            *reinterpret_cast<float4*>(dst) = f;
        }
    } else {
        const uint32_t total = tileWords * p.dim;
        const uint8_t* keyBytes = reinterpret_cast<const uint8_t*>(keyTile);
        for (uint32_t q = lane; q < total; q += WAVE) {
            const uint32_t w = q / p.dim;
            const uint32_t c = q - w * p.dim;
            const uint32_t k = keyBytes[w * p.keyStride * 4 + c];
            p.out[(tileBase + w) * p.ld + p.colOff + c] = centroidLds[k];
        }
    }
}

// ---------------------------------------------------------------------------
// dequant_uniform / gather_full
// ---------------------------------------------------------------------------

struct UniformParams {
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

// reference src/uniform_compression.cpp:70-71, evaluated left to right in fp32:
// sub, mul, div, add -- each correctly rounded, nothing fused, subnormals kept.
__device__ __forceinline__ float dequant(float minValue, float range, uint32_t v, float levels)
{
    const float scaled = mulRn(range, static_cast<float>(v));
    return addRn(minValue, __fdiv_rn(scaled, levels));
}

template <bool VEC4>
__global__ void dequant_uniform(UniformParams p)
{
    const unsigned long long blockBase = static_cast<unsigned long long>(blockIdx.x) * p.wordsPerBlock;
    const uint32_t blockWords =
        static_cast<uint32_t>(min(static_cast<unsigned long long>(p.wordsPerBlock), p.n - blockBase));

    if (VEC4) {
        const uint32_t piecesPerWord = p.dim / 4;
        const uint32_t pieces = blockWords * piecesPerWord;
        for (uint32_t q = threadIdx.x; q < pieces; q += blockDim.x) {
            const uint32_t w = fastDivide(q, p.pieceMagic, piecesPerWord);
            const uint32_t c = q - w * piecesPerWord;
            const uint32_t row = p.rows[blockBase + w];
            float4 f = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < p.nRows) {
                const float2 mm = p.minMax[row];
                const float range = subRn(mm.y, mm.x);
                const uint32_t v =
                    *reinterpret_cast<const uint32_t*>(p.values + static_cast<unsigned long long>(row) * p.dim + 4 * c);
                f.x = dequant(mm.x, range, v & 0xff, p.levels);
                f.y = dequant(mm.x, range, (v >> 8) & 0xff, p.levels);
                f.z = dequant(mm.x, range, (v >> 16) & 0xff, p.levels);
                f.w = dequant(mm.x, range, v >> 24, p.levels);
            }
            float* dst = p.out + (blockBase + w) * p.ld + p.colOff + 4 * c;
            *reinterpret_cast<float4*>(dst) = f;
        }
    } else {
        const uint32_t total = blockWords * p.dim;
        for (uint32_t q = threadIdx.x; q < total; q += blockDim.x) {
            const uint32_t w = q / p.dim;
            const uint32_t c = q - w * p.dim;
            const uint32_t row = p.rows[blockBase + w];
            float f = 0.f;
            if (row < p.nRows) {
                const float2 mm = p.minMax[row];
                const float range = subRn(mm.y, mm.x);
                f = dequant(mm.x, range, p.values[static_cast<unsigned long long>(row) * p.dim + c], p.levels);
            }
            p.out[(blockBase + w) * p.ld + p.colOff + c] = f;
        }
    }
}

struct FullParams {
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
    const unsigned long long blockBase = static_cast<unsigned long long>(blockIdx.x) * p.wordsPerBlock;
    const uint32_t blockWords =
        static_cast<uint32_t>(min(static_cast<unsigned long long>(p.wordsPerBlock), p.n - blockBase));
    if (VEC4) {
        const uint32_t piecesPerWord = p.dim / 4;
        const uint32_t pieces = blockWords * piecesPerWord;
        for (uint32_t q = threadIdx.x; q < pieces; q += blockDim.x) {
            const uint32_t w = fastDivide(q, p.pieceMagic, piecesPerWord);
            const uint32_t c = q - w * piecesPerWord;
            const uint32_t row = p.rows[blockBase + w];
            float4 f = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < p.nRows) {
                f = *reinterpret_cast<const float4*>(p.values + static_cast<unsigned long long>(row) * p.dim + 4 * c);
            }
            *reinterpret_cast<float4*>(p.out + (blockBase + w) * p.ld + p.colOff + 4 * c) = f;
        }
    } else {
        const uint32_t total = blockWords * p.dim;
        for (uint32_t q = threadIdx.x; q < total; q += blockDim.x) {
            const uint32_t w = q / p.dim;
            const uint32_t c = q - w * p.dim;
            const uint32_t row = p.rows[blockBase + w];
            p.out[(blockBase + w) * p.ld + p.colOff + c] =
                row < p.nRows ? p.values[static_cast<unsigned long long>(row) * p.dim + c] : 0.f;
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
    uint8_t* packed = nullptr;
    uint32_t* valueOffsets = nullptr;
    uint32_t* table = nullptr;
    float* centroids = nullptr;
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

uint32_t trainedLdsBytes(const memb_hip_ctx* ctx, uint32_t waves, uint32_t wordsPerWave, uint32_t keyStride)
{
    return 4u * (ctx->tableDwords + 256u + waves * wordsPerWave * (ctx->slotDwords + keyStride));
}

// Waves per block: as many resident wavefronts per CU as LDS allows (the
// decode is a chain of dependent LDS lookups, so occupancy is what hides it),
// larger blocks on ties (fewer copies of the lookup table).
TrainedGeometry chooseGeometry(
    const memb_hip_ctx* ctx, uint32_t wordsPerWave, uint32_t keyStride, size_t ld, size_t colOff, const float* out)
{
    TrainedGeometry best{};
    double bestWaves = -1;
    const uint32_t forcedWaves = envUint("MEMB_HIP_WAVES", 0);
    for (uint32_t waves : {8u, 4u, 2u, 1u}) {
        if (forcedWaves && waves != forcedWaves) {
            continue;
        }
        uint32_t ldsBytes = trainedLdsBytes(ctx, waves, wordsPerWave, keyStride);
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

template <bool HAS_SUB, int MODE>
hipError_t launchTrainedVariant(const TrainedParams& params, uint32_t blocks, uint32_t threads, uint32_t ldsBytes, hipStream_t stream)
{
    static thread_local int configuredDevice = -1;
    int device = 0;
    (void)hipGetDevice(&device);
    if (configuredDevice != device) {
        hipError_t status = hipFuncSetAttribute(
            reinterpret_cast<const void*>(&decode_trained<HAS_SUB, MODE>),
            hipFuncAttributeMaxDynamicSharedMemorySize,
            160 * 1024);
        if (status != hipSuccess) {
            return status;
        }
        configuredDevice = device;
    }
    hipLaunchKernelGGL(
        (decode_trained<HAS_SUB, MODE>), dim3(blocks), dim3(threads), ldsBytes, stream, params);
    return hipGetLastError();
}

template <int MODE>
hipError_t launchTrainedMode(
    bool hasSubTables, const TrainedParams& params, uint32_t blocks, uint32_t threads, uint32_t ldsBytes, hipStream_t stream)
{
    return hasSubTables ? launchTrainedVariant<true, MODE>(params, blocks, threads, ldsBytes, stream)
                        : launchTrainedVariant<false, MODE>(params, blocks, threads, ldsBytes, stream);
}

TrainedParams baseTrainedParams(const memb_hip_ctx* ctx)
{
    TrainedParams params{};
    params.packed = ctx->packed;
    params.valueOffsets = ctx->valueOffsets;
    params.segmentIndex = ctx->segmentIndex;
    params.table = ctx->table;
    params.centroids = ctx->centroids;
    params.nRows = ctx->nRows;
    params.tableDwords = ctx->tableDwords;
    params.rootBits = ctx->hostTable.rootBits;
    params.dim = ctx->dim;
    params.slotDwords = ctx->slotDwords;
    params.slotMagic = magicFor(ctx->slotDwords / 4, 64ull * (ctx->slotDwords / 4) * 5);
    return params;
}

int launchTrained(
    memb_hip_ctx* ctx, const uint32_t* rows, size_t n, float* out, size_t ld, size_t colOff, hipStream_t stream)
{
    const uint32_t wordsPerWave = WAVE / ctx->lanesPerWord;
    const uint32_t keyStride = (ctx->dim + 3) / 4;
    TrainedGeometry geometry = chooseGeometry(ctx, wordsPerWave, keyStride, ld, colOff, out);
    if (!geometry.waves) {
        return fail(MEMB_HIP_ERR_INVALID, "decode tables and bitstream slots do not fit into LDS");
    }
    TrainedParams params = baseTrainedParams(ctx);
    params.rows = rows;
    params.out = out;
    params.n = n;
    params.ld = ld;
    params.colOff = colOff;
    params.lanesPerWord = ctx->lanesPerWord;
    params.laneMagic = magicFor(ctx->lanesPerWord, WAVE);
    params.wordsPerWave = wordsPerWave;
    params.segmentSymbols = ctx->segmentSymbols;
    params.keyStride = keyStride;
    params.keyMagic = magicFor(keyStride, uint64_t(wordsPerWave) * keyStride);

    const size_t tiles = (n + wordsPerWave - 1) / wordsPerWave;
    const uint32_t blocks = static_cast<uint32_t>((tiles + geometry.waves - 1) / geometry.waves);
    const uint32_t threads = geometry.waves * WAVE;
    const bool sub = ctx->hostTable.hasSubTables;
    hipError_t status;
    switch (geometry.mode) {
        case OUT_FLAT:
            status = launchTrainedMode<OUT_FLAT>(sub, params, blocks, threads, geometry.ldsBytes, stream);
            break;
        case OUT_VEC4:
            status = launchTrainedMode<OUT_VEC4>(sub, params, blocks, threads, geometry.ldsBytes, stream);
            break;
        default:
            status = launchTrainedMode<OUT_SCALAR>(sub, params, blocks, threads, geometry.ldsBytes, stream);
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
    params.segmentSymbols = roundUp4(ctx->dim);
    params.keyStride = 0;
    params.indexLanes = ctx->lanesPerWord;
    params.indexSegmentSymbols = ctx->segmentSymbols;

    uint32_t waves = 4;
    while (waves > 1 && trainedLdsBytes(ctx, waves, WAVE, 0) > ctx->ldsLimit) {
        waves /= 2;
    }
    const uint32_t ldsBytes = trainedLdsBytes(ctx, waves, WAVE, 0);
    if (ldsBytes > ctx->ldsLimit) {
        return fail(MEMB_HIP_ERR_INVALID, "bitstream slots do not fit into LDS");
    }
    const size_t tiles = (ctx->nRows + WAVE - 1) / WAVE;
    const uint32_t blocks = static_cast<uint32_t>((tiles + waves - 1) / waves);
    hipError_t status = launchTrainedMode<OUT_INDEX>(
        ctx->hostTable.hasSubTables, params, blocks, waves * WAVE, ldsBytes, ctx->stream);
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
    // about 16 KiB of output per block, at least one word
    return std::max<uint32_t>(1, std::min<uint32_t>(64, 4096 / std::max<uint32_t>(dim, 1)));
}

int launchUniform(
    memb_hip_ctx* ctx, const uint32_t* rows, size_t n, float* out, size_t ld, size_t colOff, hipStream_t stream)
{
    UniformParams params{};
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
    memb_hip_ctx* ctx, const uint32_t* rows, size_t n, float* out, size_t ld, size_t colOff, hipStream_t stream)
{
    FullParams params{};
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

int launch(memb_hip_ctx* ctx, const uint32_t* rows, size_t n, float* out, size_t ld, size_t colOff, hipStream_t stream)
{
    if (n == 0) {
        return MEMB_HIP_OK;
    }
    if (n > (size_t(1) << 37)) {
        return fail(MEMB_HIP_ERR_INVALID, "batch too large");
    }
    switch (ctx->storage) {
        case memb::wire::Storage_Trained:
            return launchTrained(ctx, rows, n, out, ld, colOff, stream);
        case memb::wire::Storage_Uniform:
            return launchUniform(ctx, rows, n, out, ld, colOff, stream);
        case memb::wire::Storage_Full:
            return launchFull(ctx, rows, n, out, ld, colOff, stream);
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
        hipHostUnregister(const_cast<void*>(src));
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

    // Per-row stream length: streams are laid out back to back, so a stream
    // ends where the next one (in storage order) begins.
    {
        std::vector<uint64_t> order(desc->n_rows);
        for (uint64_t r = 0; r < desc->n_rows; ++r) {
            if (desc->value_offsets[r] > desc->packed_values_bytes) {
                delete ctx;
                return fail(MEMB_HIP_ERR_INVALID, "value offset beyond packed values");
            }
            order[r] = (static_cast<uint64_t>(desc->value_offsets[r]) << 32) | r;
        }
        std::sort(order.begin(), order.end());
        ctx->streamBytes.assign(desc->n_rows, 0);
        uint64_t nextStart = desc->packed_values_bytes;
        uint64_t previousOffset = desc->packed_values_bytes;
        for (size_t i = order.size(); i > 0; --i) {
            uint64_t offset = order[i - 1] >> 32;
            uint32_t rowIndex = static_cast<uint32_t>(order[i - 1]);
            if (offset != previousOffset) {
                nextStart = previousOffset;
                previousOffset = offset;
            }
            uint64_t bytes = nextStart - offset;
            // a stream never holds more than dim codes of the longest length
            uint64_t bound = (static_cast<uint64_t>(desc->dim) * std::max<uint32_t>(ctx->hostTable.maxCodeBits, 1) + 7) / 8;
            ctx->streamBytes[rowIndex] = static_cast<uint32_t>(std::min(bytes, bound));
            ctx->maxStreamBytes = std::max(ctx->maxStreamBytes, ctx->streamBytes[rowIndex]);
        }
    }
    // Slot: stream, up to 3 bytes of alignment slack in front, and the 12-byte
    // window the decoder reads at its last position; whole 16-byte pieces, an
    // odd number of them so that equal positions in consecutive slots fall
    // into different LDS banks.
    ctx->slotDwords = (((ctx->maxStreamBytes + 3 + 12 + 15) / 16) | 1u) * 4;
    ctx->tableDwords = static_cast<uint32_t>((ctx->hostTable.entries.size() + 3) / 4 * 4);

    // Lanes per word (G) and symbols per lane (S). The side index stores 16-bit
    // bit offsets, so rows longer than 65535 bits keep one lane per word.
    {
        uint32_t lanes = envUint("MEMB_HIP_LANES", 8);
        lanes = std::max<uint32_t>(1, std::min<uint32_t>(lanes, WAVE));
        if (uint64_t(desc->dim) * std::max<uint32_t>(ctx->hostTable.maxCodeBits, 1) >= 65536 || desc->dim < 8) {
            lanes = 1;
        }
        ctx->segmentSymbols = std::max<uint32_t>(4, roundUp4((desc->dim + lanes - 1) / lanes));
        ctx->lanesPerWord = (desc->dim + ctx->segmentSymbols - 1) / ctx->segmentSymbols;
    }

    code = openDevice(ctx, device);
    if (code == MEMB_HIP_OK) {
        // guard: a slot-sized read may start at the last byte
        size_t guard = size_t(ctx->slotDwords) * 4 + 16;
        code = deviceAlloc(ctx, &ctx->packed, desc->packed_values_bytes + guard);
        if (code == MEMB_HIP_OK) {
            hipError_t status = hipMemset(ctx->packed + desc->packed_values_bytes, 0, guard);
            if (status != hipSuccess) {
                code = fail(MEMB_HIP_ERR_DEVICE, std::string("hipMemset: ") + hipGetErrorString(status));
            }
        }
    }
    if (code == MEMB_HIP_OK) {
        code = copyToDevice(ctx->packed, desc->packed_values, desc->packed_values_bytes);
    }
    if (code == MEMB_HIP_OK) {
        code = deviceAlloc(ctx, &ctx->valueOffsets, desc->n_rows * 4);
    }
    if (code == MEMB_HIP_OK) {
        code = copyToDevice(ctx->valueOffsets, desc->value_offsets, desc->n_rows * 4);
    }
    if (code == MEMB_HIP_OK) {
        code = deviceAlloc(ctx, &ctx->table, size_t(ctx->tableDwords) * 4);
    }
    if (code == MEMB_HIP_OK) {
        std::vector<uint32_t> padded(ctx->tableDwords, 0);
        std::copy(ctx->hostTable.entries.begin(), ctx->hostTable.entries.end(), padded.begin());
        code = copyToDevice(ctx->table, padded.data(), padded.size() * 4);
    }
    if (code == MEMB_HIP_OK) {
        code = deviceAlloc(ctx, &ctx->centroids, 256 * 4);
    }
    if (code == MEMB_HIP_OK) {
        std::vector<float> codebook(256, 0.f);
        std::copy(desc->centroids, desc->centroids + desc->n_centroids, codebook.begin());
        codebook[ZERO_KEY] = 0.f;
        code = copyToDevice(ctx->centroids, codebook.data(), 256 * 4);
    }
    if (code == MEMB_HIP_OK) {
        TrainedGeometry geometry = chooseGeometry(ctx, WAVE / ctx->lanesPerWord, (ctx->dim + 3) / 4, ctx->dim, 0, nullptr);
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
        TrainedGeometry geometry = chooseGeometry(ctx, WAVE / ctx->lanesPerWord, (ctx->dim + 3) / 4, ctx->dim, 0, nullptr);
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
    for (size_t start = 0; start < n; start += sliceWords) {
        const size_t words = std::min(sliceWords, n - start);
        HIP_TRY(hipMemcpyAsync(
            ctx->stagedRows, rows + start, words * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
        int code = launch(ctx, ctx->stagedRows, words, ctx->stagedOut, dim, 0, ctx->stream);
        if (code != MEMB_HIP_OK) {
            return code;
        }
        HIP_TRY(hipMemcpy2DAsync(
            out + start * ld + col_off,
            ld * sizeof(float),
            ctx->stagedOut,
            dim * sizeof(float),
            dim * sizeof(float),
            words,
            hipMemcpyDeviceToHost,
            ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
    }
    return MEMB_HIP_OK;
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

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
// One tile per wavefront at a time and short-lived blocks: the hardware's dispatch keeps the
// memory system busier than any persistent pipeline did (rounds 1-3 built three; round 4
// measured them out: planTrained below). The one pipeline left, decode_records_persistent,
// serves batches of two to four tiles per 16 wavefronts per CU. DESIGN.md section 5 has the
// measurements behind every choice.
//
// Files: hip_device_common.h, hip_trained_kernels.h, hip_rowwise_kernels.h -- device
// code; this file -- context, staging of a model to HBM, launch geometry, the C
// ABI; hip_host_path.h -- how rows reach host buffers (pinned ring, copy threads,
// centroid indices over PCIe).
#include <hip/hip_runtime.h>

#include "../../include/memb_hip.h"
#include "codec.h"
#include "wire.h"
#include "worker_pool.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <memory>
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

// Device code (all inside this anonymous namespace).
#include "hip_device_common.h"
#include "hip_trained_kernels.h"
#include "hip_rowwise_kernels.h"
#include "hip_encoder_kernels.h"
#include "hip_words_kernels.h"

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

// Measurement / test switches: options of a live context (memb_hip_ctx_set_option) and a few environment variables read once
// when a context is created (tools/perf/README.md has the list).
struct Switches {
    uint32_t waves = 0;            // MEMB_HIP_WAVES / option waves_per_block: force the wavefronts per block (0 = choose)
    uint32_t debugFlags = 0;       // MEMB_HIP_DEBUG, builds with -DMEMB_HIP_MEASURE only (hip_trained_kernels.h: measureFlags)
    uint32_t persistent = 1;       // option persistent: 1 = the kernel by batch size (planTrained); 0 = one tile per wavefront
                                   // always, 2 = decode_records_persistent whenever the layout allows (tests, measurements)
    uint32_t unionSplit = 1;       // option union_split: decode_union_split for pairs of models with row records:
                                   // 0 = never, 1 = whenever the pair qualifies
    uint32_t unionFused = 1;       // option union_fused (of the FIRST model's context): 0 = no union kernel, the caller launches per model (tests)
    uint32_t tilesPerWave = 0;     // option tiles_per_wave: tiles a wavefront of the one-tile kernels decodes one after the
                                   // other: 0 = by rule (oneTileSteps), K = K (measurements)
    uint32_t fineLanes = 0;        // option fine_lanes: the finer index of small batches: 0 = by rule (planTrained), 1 = never,
                                   // 2 = every batch of a model that has one (tests, measurements)
    uint32_t ldsPad = 0;           // option lds_pad, builds with -DMEMB_HIP_MEASURE only: unused LDS bytes added to every block of
                                   // decode_trained (fewer resident wavefronts per CU from the same code: round 5's residency table, profiles/r05_experiments.txt)
    bool hostExpand = true;        // option host_expand: centroid indices over PCIe for host buffers
    uint32_t sliceWords = ~0u;     // MEMB_HIP_SLICE_WORDS: staging slice (tests)
    uint32_t copyChunkRows = 0;    // MEMB_HIP_COPY_CHUNK_ROWS: rows per ring chunk (tests; 0 = by size)
    uint32_t hostStreaming = 1;    // MEMB_HIP_HOST_STREAMING: non-temporal stores when host threads expand results: 0 = never,
                                   // 1 = results of 64 MB and more (default), 2 = always (tests)
    uint32_t copyThreads = 16;     // MEMB_HIP_COPY_THREADS
    bool verbose = false;          // MEMB_HIP_VERBOSE
};

struct memb_hip_ctx {
    Switches switches;
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
    memb::DecodeTable hostTable;         // 8-byte device entries: nibble keys, the index pass, the union kernel
    uint32_t tableDwords = 0;
    // byte keys (!fast): the table of the PACKED kernels -- 4-byte entries, a first level that covers
    // the longest code where 32 KiB hold it (hip_trained_kernels.h: decodeSegment)
    memb::DecodeTable byteTable;
    std::vector<memb::CodeInfo> codeLengths;   // kept to rebuild byteTable narrower when LDS is short
    uint32_t* table32 = nullptr;
    // nibble-key models only: the codebook in its BYTE-key form as well (256 plain centroids) -- with table32, which every
    // model has, a union with a byte-key model can then run as one kernel
    float* codebookBytes = nullptr;
    uint32_t maxStreamBytes = 0;
    uint32_t slotDwords = 0;
    uint16_t* segmentIndex = nullptr;    // [nRows][lanesPerWord - 1]; uint32_t entries when indexWide
    bool indexWide = false;              // some row is longer than 65535 bits
    uint32_t* rowMeta = nullptr;         // 16-byte records {start, 13-bit segment offsets}: what lookups read (or null)
    // The finer index of small batches (row-record models): [nRows][fineLanes - 1] uint16_t offsets at which symbols
    // fineSymbols, 2 fineSymbols, ... of each row start -- about sixteen lanes decode a word side by side where the records
    // have offsets for eight, which shortens the one chain of dependent LDS lookups a small batch has nothing to hide behind
    uint16_t* fineIndex = nullptr;
    uint32_t fineLanes = 0;
    uint32_t fineSymbols = 0;
    uint32_t recordPieces = 0;           // non-zero: row records (TrainedParams::recordPieces); `streams` is that array
    char unionKernel[96] = {0};            // what the last union launch with this context as its first model ran
    uint32_t lanesPerWord = 1;           // G: lanes that share one word
    uint32_t segmentSymbols = 0;         // S: symbols per lane, multiple of 4
    std::vector<uint32_t> streamBytes;   // per row, host side (reporting only)
    uint32_t ldsLimit = 0;
    uint32_t cuCount = 0;

    // uniform / full
    uint4* uniformRecords = nullptr;     // row records: {min, max, 0, 0} + the weights, regionPieces 16-byte pieces per row
    uint32_t regionPieces = 0;
    float levels = 0.f;
    float* fullValues = nullptr;

    // word -> row on the device (memb_hip_ctx_stage_words; hip_words.h): the keys and a hash table over them
    uint8_t* wordKeyBytes = nullptr;
    void* wordSlots = nullptr;           // WordSlot [wordSlotMask + 1]; non-null = staged
    uint32_t wordSlotMask = 0;
    uint32_t wordIndexKeys = 0;
    uint64_t wordIndexBytes = 0;

    // staging for the host-buffer entry point
    uint32_t* stagedRows = nullptr;
    float* stagedOut = nullptr;   // fp32 rows, or rows of centroid indices (trained: see decodeRowsAsKeys)
    size_t stagedCapacity = 0;    // words
    size_t stagedRowBytes = 0;
    std::vector<float> hostCodebook;   // the device codebook's host copy (256 centroids or 256 pairs)
    uint32_t* hostRowsPinned = nullptr;   // memb_hip_decode_words: the row ids the device looked up, for the host threads
    size_t hostRowsCapacity = 0;
    // What the last very large batch looked like to the kernel that decoded it (hip_trained_kernels.h: noteBatchOrder):
    // one word of pinned host memory, 1 = rows mostly consecutive, 0 = no particular order, ORDER_UNKNOWN before the first
    uint32_t* orderSeen = nullptr;
    uint32_t* orderSeenDevice = nullptr;
    // small batches: pinned host memory the kernel reads row ids from and writes rows to directly
    void* smallHost = nullptr;
    void* smallDevice = nullptr;
    bool smallUnavailable = false;
    // pinned ring the DMA engine fills while host threads copy earlier chunks to the caller's rows
    static constexpr int RING = 4;
    void* ring[RING] = {};
    hipEvent_t ringEvents[RING] = {};
    bool ringUnavailable = false;
    std::unique_ptr<memb::WorkerPool> copyPool;   // the host threads that empty the ring
    std::mutex mutex;
};

namespace {

struct TrainedGeometry {
    uint32_t waves;      // wavefronts per block
    uint32_t ldsBytes;   // dynamic LDS per block
    int mode;
    uint32_t resident;   // wavefronts a CU holds at this block size (LDS and registers)
};

// Wavefronts per CU the registers of the one-tile kernels (decode_trained, decode_trained_batches, decode_union_split)
// admit: 57-61 vector registers would allow 8 per SIMD, but the hardware hands out scalar registers too -- 800 per SIMD
// in steps of 16 plus 16 (MI355X_MICROARCH.md, "Residency and cooperative launch") -- and the compiler, which does not
// count that, took 106: SIX per SIMD, 24 per CU, in rounds 1-4 (seen in round 5 as a step in the time of small batches at
// exactly 24 x CUs tiles). hip_trained_kernels.h now holds these kernels to a budget (MEMB_HIP_SGPRS = 96: .sgpr_count 94,
// no vector register more, 12 lane spills in the headline kernel's 2 000 instructions; 88 gives the same seven with 20): SEVEN per SIMD. A budget of 80 (8 per SIMD) costs two vector registers of spills, 3-5 % on the
// chain of a small batch, and in blocks of eight +5 % on a key-order dump (round 5, batches 16-18, profiles/r05_experiments.txt);
// tests/test_isa.py pins the seven.
constexpr uint32_t ONE_TILE_WAVES_PER_CU = 28;
constexpr uint32_t ORDER_UNKNOWN = 2;   // memb_hip_ctx::orderSeen before any very large batch

uint32_t roundUp4(uint32_t v)
{
    return (v + 3) / 4 * 4;
}

// LDS dwords of one word's bitstream slot: the stream plus the 12-byte window the decoder reads
// at its last position, whole 16-byte pieces, an odd number of them so that equal positions in
// consecutive slots fall into different LDS banks.
uint32_t compactSlotDwords(uint32_t maxStreamBytes)
{
    return (((maxStreamBytes + 12 + 15) / 16) | 1u) * 4;
}

// Same with a row record in front (16 bytes of record + the longest stream, rounded up to 32:
// stageStreams), which must fit in front of that window as well.
uint32_t recordSlotDwords(uint32_t maxStreamBytes)
{
    return (((((16 + maxStreamBytes + 31) / 32) * 32 + 12 + 15) / 16) | 1u) * 4;
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

// LDS dwords of the PACKED kernels' table (4-byte entries, padded to 16 bytes)
uint32_t packedTableDwords(const memb_hip_ctx* ctx)
{
    return roundUp4(static_cast<uint32_t>(ctx->byteTable.entries.size()));
}

// withKeys: a lookup kernel (symbol tiles; byte-key models then use the PACKED layout); without: the index pass
uint32_t trainedLdsBytes(const memb_hip_ctx* ctx, uint32_t waves, uint32_t wordsPerWave, bool withKeys)
{
    uint32_t perWave = wordsPerWave * ctx->slotDwords + (withKeys ? keyTileDwords(ctx, wordsPerWave) : 0);
    if (withKeys && !ctx->fast) {
        return 4u * (packedTableDwords(ctx) + codebookDwords(ctx) + waves * perWave);
    }
    return 4u * (ctx->tableDwords + codebookDwords(ctx) + waves * perWave);
}

// Wavefronts per block: the PREFERRED size (planTrained: eight for batches of more than 16 R tiles, else four), unless
// another size lets a CU hold at least a quarter more resident wavefronts (LDS is handed out per block: the 8-bit model's
// 33 KiB of tables leave 12 wavefronts per CU in blocks of four and 16 in blocks of eight, which is worth 10-17 %; the
// 6-bit model's 28 against 32 is not worth leaving the preference: round 5, batch 7). Where the preferred size does not
// fit: the size with the most resident wavefronts, in the order 4, 8, 2, 1. MEMB_HIP_WAVES / option waves_per_block forces a
// size. registerWavesPerCu: how many wavefronts of the kernel about to be launched its registers let a CU hold (32 when
// unknown): a block size whose LDS would allow more resident wavefronts than the registers do gains nothing by it.
TrainedGeometry chooseGeometry(
    const memb_hip_ctx* ctx, uint32_t wordsPerWave, size_t ld, size_t colOff, const float* out,
    uint32_t registerWavesPerCu = ONE_TILE_WAVES_PER_CU, uint32_t preferred = 4)
{
    TrainedGeometry best{};
    const uint32_t forcedWaves = ctx->switches.waves;   // (1 .. 16; anything but 1, 2, 4, 8: measurements)
    auto residentWith = [&](uint32_t waves, uint32_t* ldsBytes) -> uint32_t {
        *ldsBytes = trainedLdsBytes(ctx, waves, wordsPerWave, true);
        if (*ldsBytes > ctx->ldsLimit) {
            return 0;
        }
        // LDS is handed out in 1 KiB steps of a 160 KiB pool; at most 32 waves per CU, fewer when the kernel's
        // registers say so (512 per lane and SIMD, MI355X_MICROARCH.md "Register files")
        const uint32_t blocksPerCu = std::min<uint32_t>(
            ctx->ldsLimit / ((*ldsBytes + 1023) / 1024 * 1024), std::max<uint32_t>(1, std::min<uint32_t>(32, registerWavesPerCu) / waves));
        return blocksPerCu * waves;
    };
    uint32_t preferredLds = 0;
    const uint32_t wanted = forcedWaves ? forcedWaves : preferred;
    const uint32_t preferredResident = residentWith(wanted, &preferredLds);
    uint32_t bestResident = 0;
    for (uint32_t waves : {4u, 8u, 2u, 1u}) {
        if (forcedWaves) {
            break;
        }
        uint32_t ldsBytes = 0;
        const uint32_t resident = residentWith(waves, &ldsBytes);
        if (resident > bestResident) {
            bestResident = resident;
            best.waves = waves;
            best.ldsBytes = ldsBytes;
        }
    }
    best.resident = bestResident;
    if (preferredResident && (forcedWaves || 4 * bestResident < 5 * preferredResident)) {
        best.waves = wanted;
        best.ldsBytes = preferredLds;
        best.resident = preferredResident;
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
hipError_t launchBatchesVariant(
    const TrainedParams& params, const BatchList& list, uint32_t blocks, uint32_t threads, uint32_t ldsBytes, hipStream_t stream)
{
    static thread_local int configuredDevice = -1;
    int device = 0;
    (void)hipGetDevice(&device);
    if (configuredDevice != device) {
        hipError_t status = hipFuncSetAttribute(
            reinterpret_cast<const void*>(&decode_trained_batches<HAS_SUB, MODE, FAST>),
            hipFuncAttributeMaxDynamicSharedMemorySize,
            160 * 1024);
        if (status != hipSuccess) {
            return status;
        }
        configuredDevice = device;
    }
    hipLaunchKernelGGL(
        (decode_trained_batches<HAS_SUB, MODE, FAST>), dim3(blocks), dim3(threads), ldsBytes, stream, params, list);
    return hipGetLastError();
}

template <int MODE>
hipError_t launchBatchesMode(
    const memb_hip_ctx* ctx, const TrainedParams& params, const BatchList& list, uint32_t blocks, uint32_t threads,
    uint32_t ldsBytes, hipStream_t stream)
{
    if (ctx->fast) {
        return launchBatchesVariant<false, MODE, true>(params, list, blocks, threads, ldsBytes, stream);
    }
    return ctx->byteTable.hasSubTables ? launchBatchesVariant<true, MODE, false>(params, list, blocks, threads, ldsBytes, stream)
                                       : launchBatchesVariant<false, MODE, false>(params, list, blocks, threads, ldsBytes, stream);
}

typedef void (*TrainedKernel)(TrainedParams);

// decode_records_persistent as instantiated for a context and an output mode.
template <int MODE>
TrainedKernel recordsKernelOfMode(const memb_hip_ctx* ctx)
{
    if (ctx->fast) {
        return &decode_records_persistent<false, MODE, true>;
    }
    return ctx->byteTable.hasSubTables ? &decode_records_persistent<true, MODE, false>
                                       : &decode_records_persistent<false, MODE, false>;
}

TrainedKernel recordsKernel(const memb_hip_ctx* ctx, int mode)
{
    switch (mode) {
        case OUT_FLAT:
            return recordsKernelOfMode<OUT_FLAT>(ctx);
        case OUT_VEC4:
            return recordsKernelOfMode<OUT_VEC4>(ctx);
        case OUT_KEYS:
            return recordsKernelOfMode<OUT_KEYS>(ctx);
        default:
            return recordsKernelOfMode<OUT_SCALAR>(ctx);
    }
}

// What the runtime knows about a kernel on a device: registers (-> wavefronts a CU can hold) and, per
// (block size, LDS), the resident blocks per CU. Looked up once.
struct KernelFacts {
    int numRegs = 0;
    uint32_t registerWavesPerCu = 32;
    bool ldsRaised = false;
    std::vector<std::pair<std::pair<uint32_t, uint32_t>, int>> blocksPerCu;   // (threads, ldsBytes) -> blocks
};

std::mutex g_kernelFactsMutex;
std::vector<std::pair<std::pair<const void*, int>, KernelFacts>> g_kernelFacts;   // (kernel, device) ->

// (caller holds g_kernelFactsMutex)
hipError_t kernelFactsLocked(const void* kernel, KernelFacts** out)
{
    int device = 0;
    (void)hipGetDevice(&device);
    for (auto& entry : g_kernelFacts) {
        if (entry.first.first == kernel && entry.first.second == device) {
            *out = &entry.second;
            return hipSuccess;
        }
    }
    KernelFacts facts;
    hipFuncAttributes attributes;
    hipError_t status = hipFuncGetAttributes(&attributes, kernel);
    if (status != hipSuccess) {
        return status;
    }
    facts.numRegs = attributes.numRegs;
    // allocation granule 8 registers, 512 per lane per SIMD, 4 SIMDs, at most 8 wavefronts each
    const int allocated = std::max(8, (attributes.numRegs + 7) / 8 * 8);
    facts.registerWavesPerCu = 4u * static_cast<uint32_t>(std::min(8, 512 / allocated));
    status = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (status != hipSuccess) {
        return status;
    }
    facts.ldsRaised = true;
    g_kernelFacts.push_back({{kernel, device}, facts});
    *out = &g_kernelFacts.back().second;
    return hipSuccess;
}

hipError_t registerWavesPerCu(TrainedKernel kernel, uint32_t* waves, int* numRegs)
{
    std::lock_guard<std::mutex> lock(g_kernelFactsMutex);
    KernelFacts* facts = nullptr;
    hipError_t status = kernelFactsLocked(reinterpret_cast<const void*>(kernel), &facts);
    if (status == hipSuccess) {
        *waves = facts->registerWavesPerCu;
        *numRegs = facts->numRegs;
    }
    return status;
}

// launch(blocks) enqueues `kernel` (threads per block, ldsBytes of dynamic LDS) with that many blocks.
template <typename Launch>
hipError_t launchPersistentGeneric(
    const memb_hip_ctx* ctx, const void* kernel, Launch launch, uint32_t tileBlocks, uint32_t threads, uint32_t ldsBytes,
    uint32_t tilesPerWave = 0)
{
    int blocksPerCu = 0;
    {
        std::lock_guard<std::mutex> lock(g_kernelFactsMutex);
        KernelFacts* facts = nullptr;
        hipError_t status = kernelFactsLocked(kernel, &facts);
        if (status != hipSuccess) {
            return status;
        }
        for (const auto& known : facts->blocksPerCu) {
            if (known.first.first == threads && known.first.second == ldsBytes) {
                blocksPerCu = known.second;
            }
        }
        if (!blocksPerCu) {
            status = hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocksPerCu, kernel, static_cast<int>(threads), ldsBytes);
            if (status != hipSuccess) {
                return status;
            }
            blocksPerCu = std::max(blocksPerCu, 1);
            facts->blocksPerCu.push_back({{threads, ldsBytes}, blocksPerCu});
        }
    }
    // as many blocks as are resident at once; each wavefront strides over the tiles. (Round 5, batch 11: a grid cut down so
    // that every wavefront gets the SAME number of tiles -- 12 500 tiles as 4 167 wavefronts x 3 instead of 5 120 wavefronts
    // of which 44 % run a third round -- is slower: 100 000 uncached rows +10 % (4-bit), +15 % (6-bit, 2-bit). More wavefronts
    // in flight beat an even last round.)
    const uint32_t resident = static_cast<uint32_t>(blocksPerCu) * ctx->cuCount;
    launch(std::min(tileBlocks, resident));
    return hipGetLastError();
}

hipError_t launchPersistent(
    const memb_hip_ctx* ctx, TrainedKernel kernel, const TrainedParams& params, uint32_t tileBlocks, uint32_t threads,
    uint32_t ldsBytes, hipStream_t stream)
{
    return launchPersistentGeneric(
        ctx, reinterpret_cast<const void*>(kernel), [&](uint32_t blocks) {
            hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads), ldsBytes, stream, params);
        }, tileBlocks, threads, ldsBytes);
}

template <int MODE>
hipError_t launchTrainedMode(
    const memb_hip_ctx* ctx, const TrainedParams& params, uint32_t blocks, uint32_t threads, uint32_t ldsBytes, hipStream_t stream)
{
    if (ctx->fast) {
        return launchTrainedVariant<false, MODE, true>(params, blocks, threads, ldsBytes, stream);
    }
    // (the index pass reads the 8-byte table, every other byte-key kernel the packed one)
    const bool hasSub = MODE == OUT_INDEX ? ctx->hostTable.hasSubTables : ctx->byteTable.hasSubTables;
    return hasSub ? launchTrainedVariant<true, MODE, false>(params, blocks, threads, ldsBytes, stream)
                  : launchTrainedVariant<false, MODE, false>(params, blocks, threads, ldsBytes, stream);
}

TrainedParams baseTrainedParams(const memb_hip_ctx* ctx)
{
    TrainedParams params{};
    params.streams = ctx->streams;
    params.streamStarts = ctx->streamStarts;
    params.segmentIndex = ctx->segmentIndex;
    params.indexWide = ctx->indexWide ? 1u : 0u;
    params.rowMeta = ctx->rowMeta;
    params.recordPieces = ctx->recordPieces;
    params.loadPieces = ctx->recordPieces ? ctx->recordPieces : ctx->slotDwords / 4;
    params.table = ctx->table;
    params.codebook = ctx->codebook;
    params.nRows = ctx->nRows;
    params.tableDwords = ctx->tableDwords;
    params.codebookDwords = codebookDwords(ctx);
    params.rootBits = ctx->hostTable.rootBits;
    params.dim = ctx->dim;
    params.slotDwords = ctx->slotDwords;
    params.slotMagic = magicFor(params.loadPieces, 64ull * params.loadPieces * 5);
    params.debugFlags = ctx->switches.debugFlags;
    params.tilesPerWave = 1;
    return params;
}

struct Epilogue {
    uint32_t accumulate = 0;
    float divisor = 0.f;
    bool randomOrder = false;   // the caller's hint MEMB_HIP_ROWS_IN_RANDOM_ORDER (launch geometry only)
};

// The parameters of a lookup kernel that do not depend on the batch. fine: decode with the finer index (more lanes per
// word, fewer words per wavefront; memb_hip_ctx::fineIndex).
TrainedParams lookupParams(const memb_hip_ctx* ctx, bool fine = false)
{
    const uint32_t lanesPerWord = fine ? ctx->fineLanes : ctx->lanesPerWord;
    const uint32_t wordsPerWave = WAVE / lanesPerWord;
    TrainedParams params = baseTrainedParams(ctx);
    if (fine) {
        params.segmentIndex = ctx->fineIndex;
        params.fineIndex = 1;
        params.slotMagic = magicFor(params.loadPieces, 64ull * params.loadPieces * 5);
    }
    params.lanesPerWord = lanesPerWord;
    params.laneMagic = magicFor(lanesPerWord, WAVE);
    params.wordsPerWave = wordsPerWave;
    params.segmentSymbols = fine ? ctx->fineSymbols : ctx->segmentSymbols;
    params.keyRowBytes = keyRowBytes(ctx);
    params.keyTileDwords = keyTileDwords(ctx, wordsPerWave);
    params.pieceMagic = magicFor(ctx->dim / 4, uint64_t(wordsPerWave) * (ctx->dim / 4));
    if (!ctx->fast) {
        // byte keys: the PACKED kernels' table (4-byte entries, tableDwords of them incl. padding)
        params.table = ctx->table32;
        params.rootBits = ctx->byteTable.rootBits;
        params.tableDwords = packedTableDwords(ctx);
    }
    return params;
}

// The kernels index LDS and the symbol tile from these numbers without further checks.
bool lookupParamsConsistent(const memb_hip_ctx* ctx, const TrainedParams& params, const TrainedGeometry& geometry)
{
    const uint32_t wordsPerWave = params.wordsPerWave;
    const uint32_t group = ctx->fast ? 8 : 4;
    return wordsPerWave >= 1 && wordsPerWave * params.lanesPerWord <= WAVE &&
        params.slotDwords >= 4 && params.slotDwords % 4 == 0 && params.segmentSymbols % group == 0 &&
        params.loadPieces >= 1 && params.loadPieces * 4 <= params.slotDwords &&
        (!params.recordPieces || ((params.fineIndex || params.lanesPerWord <= ROW_META_MAX_LANES) && params.slotDwords >= 4 * params.recordPieces + 3)) &&
        uint64_t(params.lanesPerWord) * params.segmentSymbols >= params.dim &&
        uint64_t(params.lanesPerWord - 1) * params.segmentSymbols < params.dim &&
        params.keyRowBytes * (ctx->fast ? 2u : 1u) >= params.dim &&
        uint64_t(params.keyTileDwords) * 4 >= uint64_t(wordsPerWave) * params.keyRowBytes &&
        (params.lanesPerWord == 1 || params.segmentIndex != nullptr) &&
        geometry.ldsBytes == trainedLdsBytes(ctx, geometry.waves, wordsPerWave, true) &&
        (ctx->fast || (params.table != nullptr && params.tableDwords >= (1u << params.rootBits))) &&
        geometry.ldsBytes <= ctx->ldsLimit;
}

// Tiles a wavefront of the one-tile kernels (decode_trained, decode_union_split) decodes one after the other, behind
// ONE copy of table(s) and codebook(s) into LDS per block. Measured (round 4, batches 2-4 and 12; `copyBytes` = what a
// block copies): the copy is 2.8 % of a 4-bit dump (4 KiB) and more than it gains back there (T = 2: -0.9 % / +1.9 % key
// order / shuffled, T = 4: +3.6 %); it is what made the 8-bit model (33 KiB) the last customer of the general persistent
// kernel (T = 2: dumps +0.1..0.3 % against it, 500 k rows -1.5 %). The split union -- 8 KiB for four tiles of four words,
// and with T > 1 a pipeline of depth one (the next tile's row regions in flight during a tile's decode and stores) --
// wants T = 2 at every size measured once its copy is overlapped with the first tile's loads (batch 12: 30 k words -6 %,
// 100 k -2 %, 160 k -3 %, 500 k -4 %, 1 M -3 % against T = 1; T = 3 and 4 are behind T = 2 everywhere).
uint32_t oneTileSteps(const memb_hip_ctx* ctx, uint64_t tiles, uint32_t copyBytes, bool unionSplit)
{
    if (ctx->switches.tilesPerWave) {
        return std::min<uint32_t>(ctx->switches.tilesPerWave, 64);
    }
    const uint64_t slots = uint64_t(ctx->cuCount) * 32;   // (the unit the thresholds below were measured in)
    if (unionSplit) {
        return 2 * tiles >= slots ? 2u : 1u;   // (from 16 k words on 256 CUs; below, halving the grid leaves CUs idle)
    }
    return copyBytes >= 16 * 1024 && tiles >= 2 * slots ? 2u : 1u;
}

// Which kernel a batch of this output shape runs, and with what launch geometry.
struct TrainedPlan {
    bool fine = false;                   // decode_trained with the finer index (small batches)
    bool persistent = false;             // decode_records_persistent (else decode_trained)
    TrainedGeometry geometry{};
    TrainedKernel kernel = nullptr;      // persistent only
    uint32_t registerWavesPerCu = 32;    // persistent only: what the kernel's registers allow
    int numRegs = 0;
};

constexpr uint64_t PIPELINE_WAVES_PER_CU = 16;   // the unit R of the rule below (times the CUs)
// Do the rows of very large batches come in no particular order? The caller's word for it (MEMB_HIP_ROWS_IN_RANDOM_ORDER),
// else what the kernel of the last such batch saw (memb_hip_ctx::orderSeen); key order until something was seen.
bool rowsUnordered(const memb_hip_ctx* ctx, bool callerSaysRandom)
{
    if (callerSaysRandom) {
        return true;
    }
    return ctx->orderSeen && __atomic_load_n(ctx->orderSeen, __ATOMIC_RELAXED) == 0;
}

// (the context's device is current)
// Which kernel by batch size (n words): a STATIC rule. t = tiles of the batch, R = 16 x CUs. Round 4 measured every
// kernel of rounds 1-3 on every model kind (2-, 4-, 6-bit, byte-key 4-bit, Student-t 4-bit; key order, shuffled, 100 k,
// 500 k; round 4, batch 1, profiles/r04_experiments.txt: two boxes, A/A floor 0.5 %) and kept what wins a BASELINE configuration by 3 %:
//   decode_trained (one tile per wavefront at a time, 52-61 VGPRs; 28 wavefronts per CU by its scalar registers, 24 until
//   round 5: ONE_TILE_WAVES_PER_CU) -- everything, except
//   one round of the one-tile kernel (28 x CUs = 1.75 R) < t <= 4 R on row-record models: decode_records_persistent (24
//   wavefronts per CU, software pipeline):
//       100 000 rows -4.3..-6 % (4-bit), -8..-9 % (6-bit), -1.5 % (2-bit); at 500 k rows it is 3-11 % BEHIND.
// The general persistent pipeline lost every dump (+2.3 % 4-bit, +4.9 % 2-bit; 6-bit -1.8 % on one box) and every
// shuffled batch (+1.7..+9.4 %); with it went the per-context timing that chose between the two ("autotune").
// Block size: four wavefronts; EIGHT for batches of more than 16 R tiles (524 000 words on 256 CUs) -- dumps: the
// reference's own large batch is keys() in key order (python/memb/reader.py:27-28). Round 5, batch 7 (one box, every model
// kind, both orders; eight against four): key-order dumps 4-bit -2.4 %, 2-bit -5.0 %, 6-bit -1.0 %, 9-bit-code 4-bit
// -1.1 %, Student-t -2.5 %; the same batches SHUFFLED +3.4 %, -1.3 %, +3.6 %, +3.5 %, +2.9 %; 1 M random rows +2.0 %, -1.5 %,
// +2.5 %, +2.7 %, +2.1 %: the rule takes the dump's side for every key format and says what it costs the other order
// (HISTORY.md, "(r5) 5.0", has the table; DESIGN.md section 5.0 the rule as it stands). The 8-bit model runs blocks of eight at every size: chooseGeometry.
// force: -1 = by the rule, 0 = one tile per wavefront, 1 = decode_records_persistent where the layout allows
int planTrained(
    const memb_hip_ctx* ctx, size_t n, size_t ld, size_t colOff, const float* out, bool keysOut, TrainedPlan* plan, int force = -1,
    bool mayBeFine = true, bool randomOrder = false)
{
    uint32_t wordsPerWave = WAVE / ctx->lanesPerWord;
    const uint64_t tiles = (n + wordsPerWave - 1) / wordsPerWave;
    const uint64_t R = uint64_t(ctx->cuCount) * PIPELINE_WAVES_PER_CU;
    // One round = the tiles the CUs hold at once (`resident` = wavefronts per CU at the block size of a small batch, by LDS
    // and registers; x CUs). Two edges of the rule are rounds:
    //   * the FINER INDEX while the batch's fine tiles all fit one round (28 600 words on 256 CUs): every wavefront has one
    //     tile, the batch is one chain of dependent steps per wavefront, and the finer index shortens its longest link, the
    //     decode. Round 5 (batches 3, 17 and 26, profiles/r05_experiments.txt; the usual index = 100 %), 1 000 /
    //     10 000 / 20 000 / 28 000 rows: 4-bit -12 / -13 / -8 / -4 % with the same batch repeated, -15 / -8 / -5 / -5 % with
    //     nothing cached between launches; 6-bit -15 / -17 / -9 / -8 % and -21 / -13 / -9 / -8 %. One row more than a round
    //     and its second round costs what the index saved (30 000 rows +9 / +5 %; nothing cached: +16 / +12 %, 50 000 rows +33 %).
    //   * decode_records_persistent from ONE ROUND OF THE USUAL INDEX on (57 344 words; until the end of round 5: from 2 R =
    //     65 536): past one round the one-tile kernel runs a nearly empty second one (56 000 -> 60 000 rows: 17.3 -> 21.0 us),
    //     the pipeline's wavefronts take a second tile instead. 58 000 / 62 000 / 65 000 rows against the one-tile kernel,
    //     nothing cached: 4-bit -9 / -10 / -11 %, 6-bit -7 / -10 / -12 %; the same batch repeated: -6 / 0 % at 60 000 rows.
    //     (For a few hours the rule sent this band to the finer index, on measurements of a repeated batch alone: -6..-12 %
    //     there, +9..+18 % with nothing cached -- 20-32 % behind the pipeline. Batch 26.)
    // (a forced kernel -- option persistent = 2, force = 1 -- wins over the rule, a forced finer index over both)
    bool fineByRule = false;
    if (mayBeFine && ctx->fineIndex && ctx->switches.fineLanes == 0 && ctx->switches.persistent != 2 && force != 1) {
        const uint32_t fineWords = WAVE / ctx->fineLanes;
        const uint64_t fineTiles = (n + fineWords - 1) / fineWords;
        const uint64_t fineRound = uint64_t(ctx->cuCount) * chooseGeometry(ctx, fineWords, ld, colOff, out).resident;
        fineByRule = fineTiles <= fineRound;
    }
    plan->fine = mayBeFine && ctx->fineIndex && force != 1 && (ctx->switches.fineLanes == 2 || fineByRule);
    // (models whose tables leave a CU fewer than 1.5 R wavefronts -- the 8-bit one -- keep the edge at 2 R: not measured there)
    const uint64_t usualRound = uint64_t(ctx->cuCount) * chooseGeometry(ctx, wordsPerWave, ld, colOff, out).resident;
    const uint64_t pipelineFrom = 2 * usualRound >= 3 * R ? std::min<uint64_t>(2 * R, usualRound) : 2 * R;
    if (plan->fine) {
        wordsPerWave = WAVE / ctx->fineLanes;
    }
    // the pipeline keeps a tile's row regions in two registers per lane: the slot image must fit two 64-lane rounds
    const bool recordsFit = ctx->recordPieces && wordsPerWave * (ctx->slotDwords / 4) <= RECORD_ROUNDS * WAVE;
    bool wantPersistent = ctx->switches.persistent == 2 || (ctx->switches.persistent == 1 && tiles > pipelineFrom && tiles <= 4 * R);
    if (force >= 0) {
        wantPersistent = force != 0;
    }
    plan->persistent = recordsFit && wantPersistent && !plan->fine;
    // Block size of very large batches (more than 16 R tiles): EIGHT wavefronts for rows in key order -- the dumps; three
    // blocks = 24 resident wavefronts per CU and a table copy per eight tiles -- and, for rows in no particular order
    // (randomOrder: the caller's hint or what the last such batch looked like, rowsUnordered), SEVEN: four whole blocks = the
    // 28 wavefronts the registers admit, which random row regions (two lines each) want in flight. Round 6, batches 2-3, two
    // boxes, seven against eight: shuffled 2.2 M rows 4-bit -4.7 / -3.7 %, 6-bit -3.6 / -4.7 %, byte-key 4-bit -4.6 %,
    // Student-t -3.8 %, 1 M random rows -5.6 / -3.8 % (four: -3.0 / -2.3, -1.2 / -3.6, -3.4, -2.8, -4.6 / -2.3 %); key-order
    // dumps -0.3 / +4.9 % (4-bit), -5.5 / -1.0 % (6-bit), +2.5 %, +5.0 %: the order decides, which is why it is looked at.
    // Models with row regions below 160 bytes (the 2-bit one) keep four: seven costs them 5-10 % in either order. And seven
    // only where it DOES hold more resident wavefronts than eight: models with large tables (8-bit: 33-45 KiB of LDS per
    // block) get their block size for residency (chooseGeometry; eight against four: -10..-15 %, round 5) and keep eight.
    // (Batch 7, the 8-bit Student-t model, 21 resident in blocks of seven against 24: seven -2.0 % shuffled, +1.2 % key order.)
    uint32_t unorderedWaves = ctx->recordPieces >= 10 ? 7u : 4u;
    if (unorderedWaves == 7 && !ctx->switches.waves &&
        chooseGeometry(ctx, wordsPerWave, ld, colOff, out, ONE_TILE_WAVES_PER_CU, 7).resident <=
            chooseGeometry(ctx, wordsPerWave, ld, colOff, out, ONE_TILE_WAVES_PER_CU, 8).resident) {
        unorderedWaves = 8;
    }
    const uint32_t preferred = !plan->persistent && tiles > 16 * R ? (randomOrder ? unorderedWaves : 8u) : 4u;
    plan->geometry = chooseGeometry(ctx, wordsPerWave, ld, colOff, out, ONE_TILE_WAVES_PER_CU, preferred);
    if (keysOut) {
        plan->geometry.mode = OUT_KEYS;
    }
    if (plan->persistent && plan->geometry.waves) {
        // again with what the kernel's registers allow (a block size whose LDS would hold more wavefronts than
        // the registers admit is no better than a smaller one)
        plan->kernel = recordsKernel(ctx, plan->geometry.mode);
        hipError_t status = registerWavesPerCu(plan->kernel, &plan->registerWavesPerCu, &plan->numRegs);
        if (status != hipSuccess) {
            return fail(MEMB_HIP_ERR_DEVICE, std::string("hipFuncGetAttributes: ") + hipGetErrorString(status));
        }
        const int mode = plan->geometry.mode;
        plan->geometry = chooseGeometry(ctx, wordsPerWave, ld, colOff, out, plan->registerWavesPerCu);
        plan->geometry.mode = mode;
    }
    if (plan->fine && !plan->geometry.waves) {   // (cannot happen: fewer words per wavefront need less LDS)
        plan->fine = false;
        return planTrained(ctx, n, ld, colOff, out, keysOut, plan, force, false, randomOrder);
    }
    return MEMB_HIP_OK;
}

// keysOut: `out` receives rows of centroid indices (OUT_KEYS) instead of fp32 rows; ld = dim, colOff = 0.
// Enqueues one kernel on `stream` and returns.
int launchTrained(
    memb_hip_ctx* ctx, const uint32_t* rows, size_t n, float* out, size_t ld, size_t colOff, hipStream_t stream,
    const Epilogue& epilogue, bool keysOut = false, int force = -1)
{
    TrainedPlan plan;
    int planned = planTrained(ctx, n, ld, colOff, out, keysOut, &plan, force, true, rowsUnordered(ctx, epilogue.randomOrder));
    if (planned != MEMB_HIP_OK) {
        return planned;
    }
    const uint32_t wordsPerWave = WAVE / (plan.fine ? ctx->fineLanes : ctx->lanesPerWord);
    const bool persistent = plan.persistent;
    TrainedGeometry geometry = plan.geometry;
    TrainedKernel kernel = plan.kernel;
    if (!geometry.waves) {
        return fail(MEMB_HIP_ERR_INVALID, "decode tables and bitstream slots do not fit into LDS");
    }
    TrainedParams params = lookupParams(ctx, plan.fine);
    params.rows = rows;
    params.out = out;
    params.n = n;
    params.ld = ld;
    params.colOff = colOff;
    params.accumulate = epilogue.accumulate;
    params.divisor = epilogue.divisor;
    if (!lookupParamsConsistent(ctx, params, geometry) || ld < colOff + params.dim) {
        return fail(MEMB_HIP_ERR_INVALID, "internal error: inconsistent decode geometry");
    }
    const size_t tiles = (n + wordsPerWave - 1) / wordsPerWave;
    const uint32_t threads = geometry.waves * WAVE;
    hipError_t status;
    if (persistent) {
        const uint32_t blocks = static_cast<uint32_t>((tiles + geometry.waves - 1) / geometry.waves);
        status = launchPersistent(ctx, kernel, params, blocks, threads, geometry.ldsBytes, stream);
    } else {
        params.tilesPerWave = oneTileSteps(ctx, tiles, 4u * (params.tableDwords + params.codebookDwords), false);
        // very large batches leave word of their order for the next one (device-resident row ids only: the host-buffer
        // entry points stage slices of the caller's batch)
        const uint64_t R = uint64_t(ctx->cuCount) * PIPELINE_WAVES_PER_CU;
        if (tiles > 16 * R && !keysOut && rows && ctx->orderSeenDevice) {
            params.segmentIndexOut = reinterpret_cast<uint16_t*>(ctx->orderSeenDevice);
        }
        const size_t perBlock = size_t(geometry.waves) * params.tilesPerWave;
        const uint32_t blocks = static_cast<uint32_t>((tiles + perBlock - 1) / perBlock);
        const uint32_t ldsBytes = std::min<uint32_t>(geometry.ldsBytes + ctx->switches.ldsPad, 160 * 1024);   // (ldsPad: measurement builds)
        switch (geometry.mode) {
            case OUT_FLAT:
                status = launchTrainedMode<OUT_FLAT>(ctx, params, blocks, threads, ldsBytes, stream);
                break;
            case OUT_VEC4:
                status = launchTrainedMode<OUT_VEC4>(ctx, params, blocks, threads, ldsBytes, stream);
                break;
            case OUT_KEYS:
                status = launchTrainedMode<OUT_KEYS>(ctx, params, blocks, threads, ldsBytes, stream);
                break;
            default:
                status = launchTrainedMode<OUT_SCALAR>(ctx, params, blocks, threads, ldsBytes, stream);
                break;
        }
    }
    if (status != hipSuccess) {
        return fail(MEMB_HIP_ERR_DEVICE, std::string("decode_trained launch: ") + hipGetErrorString(status));
    }
    return MEMB_HIP_OK;
}

// memb_hip_decode_batches_device for a trained storage: the tiles of up to MAX_BATCHES batches numbered through in one
// decode_trained_batches grid (the same body as decode_trained: one tile per wavefront at a time), launch geometry by
// the rule of a single batch with as many words as all of them together.
int launchTrainedBatches(memb_hip_ctx* ctx, const memb_hip_batch* batches, size_t count, hipStream_t stream)
{
    size_t words = 0;
    int mode = OUT_FLAT;
    for (size_t k = 0; k < count; ++k) {
        const memb_hip_batch& batch = batches[k];
        words += batch.n;
        const bool vec = (ctx->dim % 4 == 0) && (batch.ld % 4 == 0) && (batch.col_off % 4 == 0) &&
            (reinterpret_cast<uintptr_t>(batch.out) % 16 == 0);
        const int batchMode = !vec ? OUT_SCALAR : (batch.ld == ctx->dim && batch.col_off == 0) ? OUT_FLAT : OUT_VEC4;
        // one output mode for the launch: the most general one any batch needs (OUT_SCALAR < OUT_VEC4 < OUT_FLAT)
        mode = std::min(mode, batchMode);
    }
    // (the finer index as for one batch of as many words: a serving loop's handful of small lookups is a small batch)
    TrainedPlan plan;
    const int planned = planTrained(ctx, words, ctx->dim, 0, nullptr, false, &plan, 0, true);
    if (planned != MEMB_HIP_OK) {
        return planned;
    }
    const uint32_t wordsPerWave = WAVE / (plan.fine ? ctx->fineLanes : ctx->lanesPerWord);
    BatchList list{};
    uint64_t tiles = 0;
    for (size_t k = 0; k < count; ++k) {
        const memb_hip_batch& batch = batches[k];
        list.firstTile[list.count] = tiles;
        list.rows[list.count] = batch.rows;
        list.out[list.count] = batch.out;
        list.n[list.count] = batch.n;
        list.ld[list.count] = batch.ld;
        list.colOff[list.count] = batch.col_off;
        ++list.count;
        tiles += (batch.n + wordsPerWave - 1) / wordsPerWave;
    }
    list.firstTile[list.count] = tiles;
    TrainedGeometry geometry = plan.geometry;
    geometry.mode = mode;
    if (!geometry.waves) {
        return fail(MEMB_HIP_ERR_INVALID, "decode tables and bitstream slots do not fit into LDS");
    }
    TrainedParams params = lookupParams(ctx, plan.fine);
    if (!lookupParamsConsistent(ctx, params, geometry)) {
        return fail(MEMB_HIP_ERR_INVALID, "internal error: inconsistent decode geometry");
    }
    params.tilesPerWave = oneTileSteps(ctx, tiles, 4u * (params.tableDwords + params.codebookDwords), false);
    const size_t perBlock = size_t(geometry.waves) * params.tilesPerWave;
    const uint64_t blocks = (tiles + perBlock - 1) / perBlock;
    if (blocks >= 0x7FFFFFFFull) {
        return fail(MEMB_HIP_ERR_INVALID, "batches too large for one launch");
    }
    const uint32_t threads = geometry.waves * WAVE;
    hipError_t status;
    switch (mode) {
        case OUT_FLAT:
            status = launchBatchesMode<OUT_FLAT>(ctx, params, list, static_cast<uint32_t>(blocks), threads, geometry.ldsBytes, stream);
            break;
        case OUT_VEC4:
            status = launchBatchesMode<OUT_VEC4>(ctx, params, list, static_cast<uint32_t>(blocks), threads, geometry.ldsBytes, stream);
            break;
        default:
            status = launchBatchesMode<OUT_SCALAR>(ctx, params, list, static_cast<uint32_t>(blocks), threads, geometry.ldsBytes, stream);
            break;
    }
    if (status != hipSuccess) {
        return fail(MEMB_HIP_ERR_DEVICE, std::string("decode_trained_batches launch: ") + hipGetErrorString(status));
    }
    return MEMB_HIP_OK;
}

typedef void (*UnionKernel)(UnionParams);

template <bool HAS_SUB, bool FAST, bool AVERAGE>
UnionKernel unionKernelOf(size_t count)
{
    switch (count) {
        case 2:
            return &decode_trained_union<HAS_SUB, FAST, 2, AVERAGE>;
        case 3:
            return &decode_trained_union<HAS_SUB, FAST, 3, AVERAGE>;
        default:
            return &decode_trained_union<HAS_SUB, FAST, 4, AVERAGE>;
    }
}

UnionKernel unionKernel(bool hasSub, bool fast, bool average, size_t count)
{
    if (fast) {
        return average ? unionKernelOf<false, true, true>(count) : unionKernelOf<false, true, false>(count);
    }
    if (hasSub) {
        return average ? unionKernelOf<true, false, true>(count) : unionKernelOf<true, false, false>(count);
    }
    return average ? unionKernelOf<false, false, true>(count) : unionKernelOf<false, false, false>(count);
}

// See memb_hip_decode_rows_union_device. MEMB_HIP_UNSUPPORTED when the models cannot share the kernel.
int launchTrainedUnion(
    memb_hip_ctx* const* ctxs, const uint32_t* const* rows, const size_t* colOffs, size_t count, size_t n, float* out,
    size_t ld, hipStream_t stream, bool average)
{
    // (MEMB_HIP_UNSUPPORTED is not an error, but memb_hip_last_error() says which condition it was)
    if (count < 2 || count > UNION_MAX_MODELS || (ctxs[0] && ctxs[0]->switches.unionFused == 0)) {
        return fail(MEMB_HIP_UNSUPPORTED, "union kernel: two to four models (or switched off by the first context's option union_fused = 0)");
    }
    const memb_hip_ctx* first = ctxs[0];
    bool hasSub = false;
    bool allFast = true;
    for (size_t m = 0; m < count; ++m) {
        const memb_hip_ctx* ctx = ctxs[m];
        if (ctx->storage != memb::wire::Storage_Trained || ctx->device != first->device || ctx->dim != first->dim ||
            ctx->lanesPerWord != first->lanesPerWord ||
            ctx->segmentSymbols != first->segmentSymbols || ctx->dim % 4 != 0 || colOffs[m] % 4 != 0 ||
            ld < colOffs[m] + ctx->dim) {
            return fail(MEMB_HIP_UNSUPPORTED, "union kernel: the models differ in storage, device, dim or lane geometry");
        }
        allFast = allFast && ctx->fast;
    }
    for (size_t m = 0; m < count; ++m) {
        // (byte keys decode through the 4-byte PACKED tables, whose first level is wider: hip_trained_kernels.h)
        hasSub = hasSub || (allFast ? ctxs[m]->hostTable.hasSubTables : ctxs[m]->byteTable.hasSubTables);
    }
    if (!allFast) {
        for (size_t m = 0; m < count; ++m) {
            if (!ctxs[m]->table32) {
                return fail(MEMB_HIP_UNSUPPORTED, "union kernel: a model has no byte-key table on the device");
            }
        }
    }
    // Nibble keys (<= 16 centroids, codes <= 8 bits) only when every model has them; a mixed union decodes the
    // nibble-key models through their byte-key tables (memb_hip_ctx::table32): the same symbols, one per byte.
    if (ld % 4 != 0 || reinterpret_cast<uintptr_t>(out) % 16 != 0 || n > (size_t(1) << 37)) {
        return fail(MEMB_HIP_UNSUPPORTED, "union kernel: the output is not 16-byte aligned in every row");
    }
    const uint32_t wordsPerWave = WAVE / first->lanesPerWord;
    const size_t tiles = (n + wordsPerWave - 1) / wordsPerWave;

    UnionParams params{};
    uint32_t sharedDwords = 0;
    for (size_t m = 0; m < count; ++m) {
        memb_hip_ctx* ctx = ctxs[m];
        TrainedParams& p = params.model[m];
        p = baseTrainedParams(ctx);
        p.rows = rows[m];
        p.out = out;
        p.n = n;
        p.ld = ld;
        p.colOff = colOffs[m];
        p.lanesPerWord = ctx->lanesPerWord;
        p.laneMagic = magicFor(ctx->lanesPerWord, WAVE);
        p.wordsPerWave = wordsPerWave;
        p.segmentSymbols = ctx->segmentSymbols;
        p.keyRowBytes = allFast ? keyRowBytes(ctx) : roundUp4(ctx->dim);
        p.keyTileDwords = (wordsPerWave * p.keyRowBytes + 3) / 4 + 1;
        if (!allFast) {
            // byte keys: the single-model kernels' 4-byte table entries (decodeSegment<..., PACKED>); a nibble-key model
            // beside a byte-key one goes through its own byte-key table and plain codebook
            p.table = ctx->table32;
            p.tableDwords = packedTableDwords(ctx);
            p.rootBits = ctx->byteTable.rootBits;
            if (ctx->fast) {
                p.codebook = ctx->codebookBytes;
                p.codebookDwords = 256;
            }
        }
        p.pieceMagic = magicFor(ctx->dim / 4, uint64_t(wordsPerWave) * count * (ctx->dim / 4));
        p.debugFlags = first->switches.debugFlags;   // (measurement builds: the first reader's switches for all)
        params.tableOffsetDwords[m] = sharedDwords;
        sharedDwords += p.tableDwords;
    }
    params.codebookOffsetDwords = sharedDwords;
    sharedDwords += static_cast<uint32_t>(count) * 512;
    params.sharedDwords = sharedDwords;
    params.rowPieces = (average ? 1u : static_cast<uint32_t>(count)) * (first->dim / 4);
    params.rowMagic = magicFor(params.rowPieces, uint64_t(wordsPerWave) * params.rowPieces);

    // one wavefront's LDS area in decode_trained_union: bitstream slots and a symbol tile per model
    auto layOut = [&] {
        uint32_t at = 0;
        for (size_t m = 0; m < count; ++m) {
            params.slotOffsetDwords[m] = at;
            at += roundUp4(wordsPerWave * ctxs[m]->slotDwords);
            params.keyTileOffsetDwords[m] = at;
            at += roundUp4(params.model[m].keyTileDwords);
        }
        params.perWaveDwords = at;
    };

    // waves per block: most resident wavefronts per CU -- by LDS and by the kernel's registers --, blocks of
    // four on ties (as chooseGeometry)
    auto chooseWaves = [&](uint32_t sharedDwords, uint32_t perWaveDwords, uint32_t registerWaves, uint32_t* waves, uint32_t* ldsBytes) {
        double bestResident = -1;
        *waves = 0;
        // (a forced block size -- option waves_per_block, 1 .. 16 -- is the only candidate, as in chooseGeometry)
        const uint32_t forced = first->switches.waves;
        for (uint32_t candidate : {forced ? forced : 4u, 8u, 2u, 1u}) {
            if (forced && candidate != forced) {
                continue;
            }
            const uint32_t bytes = 4u * (sharedDwords + candidate * perWaveDwords);
            if (bytes > first->ldsLimit) {
                continue;
            }
            const uint32_t blocksPerCu = std::min<uint32_t>(
                first->ldsLimit / ((bytes + 1023) / 1024 * 1024), std::max<uint32_t>(1, std::min<uint32_t>(32, registerWaves) / candidate));
            if (double(blocksPerCu) * candidate > bestResident) {
                bestResident = double(blocksPerCu) * candidate;
                *waves = candidate;
                *ldsBytes = bytes;
            }
        }
    };
    uint32_t waves = 0;
    uint32_t ldsBytes = 0;
    UnionKernel kernel = nullptr;
    uint32_t registerWaves = 32;
    int numRegs = 0;

    // decode_union_split: two models staged as row records -- the wavefront's word slots are divided
    // between the models, a tile is half as many words, LDS per wavefront as in the single-model kernel.
    // Against decode_trained_union (nibble keys, round 3 batch 32): 10 k words -20 %, 30 k -14 %, 100 k -10 %, 500 k -9 %.
    bool split = count == 2 && first->switches.unionSplit != 0 && wordsPerWave % 2 == 0 &&
        ((wordsPerWave / 2) * params.model[0].keyRowBytes) % 4 == 0;
    // the slots take the geometry of the model with the larger row regions; the other model's loads then run up to as
    // many pieces into the rows behind its own (its array ends with a guard of more than one region)
    const size_t larger = count == 2 && ctxs[1]->slotDwords > ctxs[0]->slotDwords ? 1 : 0;
    for (size_t m = 0; m < count && split; ++m) {
        split = ctxs[m]->recordPieces && ctxs[m]->recordPieces <= ctxs[larger]->recordPieces &&
            ctxs[larger]->recordPieces <= 2 * ctxs[m]->recordPieces &&
            wordsPerWave * (ctxs[larger]->slotDwords / 4) <= RECORD_ROUNDS * WAVE;
    }
    if (split) {
        UnionParams sp = params;
        uint32_t shared = sharedDwords;
        const bool packedSub = !allFast && hasSub;
        // nibble keys through the 4-byte tables (round 5: -2.5 % at 500 000 and 1 M words): half the table bytes in the block's LDS image
        const bool compact = allFast && ctxs[0]->table32 && ctxs[1]->table32 &&
            !ctxs[0]->byteTable.hasSubTables && !ctxs[1]->byteTable.hasSubTables;
        if (compact) {
            shared = 0;
            for (size_t m = 0; m < 2; ++m) {
                sp.model[m].table = ctxs[m]->table32;
                sp.model[m].tableDwords = packedTableDwords(ctxs[m]);
                sp.model[m].rootBits = ctxs[m]->byteTable.rootBits;
                sp.tableOffsetDwords[m] = shared;
                shared += sp.model[m].tableDwords;
            }
            sp.codebookOffsetDwords = shared;
            shared += 2 * 512;
            sp.sharedDwords = shared;
        }
        const uint32_t half = wordsPerWave / 2;
        sp.model[2] = sp.model[larger];
        sp.model[2].nRows = 0xFFFFFFFFu;   // (rows reach the decoder checked against their own model, or MISSING)
        sp.slotOffsetDwords[0] = 0;
        sp.keyTileOffsetDwords[0] = roundUp4(wordsPerWave * ctxs[larger]->slotDwords);
        sp.keyTileOffsetDwords[1] = sp.keyTileOffsetDwords[0] + half * sp.model[0].keyRowBytes / 4;
        sp.perWaveDwords = sp.keyTileOffsetDwords[0] + roundUp4(sp.model[0].keyTileDwords);
        kernel = compact ? (average ? &decode_union_split<false, true, true, true> : &decode_union_split<false, true, false, true>)
            : allFast ? (average ? &decode_union_split<false, true, true> : &decode_union_split<false, true, false>)
            : packedSub ? (average ? &decode_union_split<true, false, true> : &decode_union_split<true, false, false>)
                        : (average ? &decode_union_split<false, false, true> : &decode_union_split<false, false, false>);
        hipError_t status = registerWavesPerCu(reinterpret_cast<TrainedKernel>(kernel), &registerWaves, &numRegs);
        if (status != hipSuccess) {
            return fail(MEMB_HIP_ERR_DEVICE, std::string("hipFuncGetAttributes: ") + hipGetErrorString(status));
        }
        chooseWaves(shared, sp.perWaveDwords, registerWaves, &waves, &ldsBytes);
        if (waves && (n + half - 1) / half / waves >= 0x7FFFFFFFull) {
            waves = 0;   // (more blocks than a grid holds: the forms below have tiles twice as large)
        }
        if (waves) {
            const size_t splitTiles = (n + half - 1) / half;
            sp.model[2].tilesPerWave = oneTileSteps(first, splitTiles, 4u * shared, true);
            const size_t perBlock = size_t(waves) * sp.model[2].tilesPerWave;
            {
                std::lock_guard<std::mutex> lock(g_kernelFactsMutex);   // (raises the kernel's LDS limit on first use)
                KernelFacts* facts = nullptr;
                status = kernelFactsLocked(reinterpret_cast<const void*>(kernel), &facts);
            }
            if (status == hipSuccess) {
                hipLaunchKernelGGL(kernel, dim3(static_cast<uint32_t>((splitTiles + perBlock - 1) / perBlock)), dim3(waves * WAVE), ldsBytes, stream, sp);
                status = hipGetLastError();
            }
            if (status != hipSuccess) {
                return fail(MEMB_HIP_ERR_DEVICE, std::string("decode_union_split launch: ") + hipGetErrorString(status));
            }
            std::snprintf(ctxs[0]->unionKernel, sizeof(ctxs[0]->unionKernel), "decode_union_split<%s, %s, %s, %s>",
                          packedSub ? "true" : "false", allFast ? "true" : "false", average ? "true" : "false", compact ? "true" : "false");
            return MEMB_HIP_OK;
        }
        // (does not fit: the forms below lay their areas out afresh)
        registerWaves = 32;
    }
    // decode_trained_union, one tile per wavefront: three or four models, pairs without row records
    layOut();
    kernel = unionKernel(hasSub, allFast, average, count);
    hipError_t status = registerWavesPerCu(reinterpret_cast<TrainedKernel>(kernel), &registerWaves, &numRegs);
    if (status != hipSuccess) {
        return fail(MEMB_HIP_ERR_DEVICE, std::string("hipFuncGetAttributes: ") + hipGetErrorString(status));
    }
    chooseWaves(sharedDwords, params.perWaveDwords, registerWaves, &waves, &ldsBytes);
    if (!waves) {
        return fail(MEMB_HIP_UNSUPPORTED, "union kernel: tables and bitstream slots of the models do not fit into LDS together");
    }
    const uint32_t threads = waves * WAVE;
    const uint32_t tileBlocks = static_cast<uint32_t>((tiles + waves - 1) / waves);
    {
        std::lock_guard<std::mutex> lock(g_kernelFactsMutex);   // (raises the kernel's LDS limit on first use)
        KernelFacts* facts = nullptr;
        status = kernelFactsLocked(reinterpret_cast<const void*>(kernel), &facts);
    }
    if (status == hipSuccess) {
        hipLaunchKernelGGL(kernel, dim3(tileBlocks), dim3(threads), ldsBytes, stream, params);
        status = hipGetLastError();
    }
    if (status != hipSuccess) {
        return fail(MEMB_HIP_ERR_DEVICE, std::string("decode_trained_union launch: ") + hipGetErrorString(status));
    }
    std::snprintf(ctxs[0]->unionKernel, sizeof(ctxs[0]->unionKernel), "decode_trained_union<%s, %s, %zu, %s>",
                  hasSub ? "true" : "false", allFast ? "true" : "false", count, average ? "true" : "false");
    return MEMB_HIP_OK;
}

// One pass over every row with one lane per word: records the bit position at
// which each segment of each row starts (segmentIndex), so that lanesPerWord
// lanes can later decode a row side by side.
// (lanes, symbols, target: the index to build -- the context's own, or the finer one of small batches)
int buildSegmentIndex(memb_hip_ctx* ctx, uint32_t lanes, uint32_t symbols, uint16_t* target)
{
    if (lanes <= 1 || ctx->nRows == 0) {
        return MEMB_HIP_OK;
    }
    TrainedParams params = baseTrainedParams(ctx);
    params.segmentIndex = nullptr;
    params.segmentIndexOut = target;
    params.rows = nullptr;
    params.n = ctx->nRows;
    params.lanesPerWord = 1;
    params.laneMagic = 0;
    params.segmentSymbols = (ctx->dim + 7) / 8 * 8;
    params.keyRowBytes = 0;
    params.keyTileDwords = 0;
    params.indexLanes = lanes;
    params.indexSegmentSymbols = symbols;

    // one lane per word, normally 64 words per wavefront; fewer (the other lanes idle) when
    // 64 bitstream slots are more than LDS holds
    uint32_t wordsPerWave = WAVE;
    while (wordsPerWave > 1 && trainedLdsBytes(ctx, 1, wordsPerWave, false) > ctx->ldsLimit) {
        wordsPerWave /= 2;
    }
    uint32_t waves = 4;
    while (waves > 1 && trainedLdsBytes(ctx, waves, wordsPerWave, false) > ctx->ldsLimit) {
        waves /= 2;
    }
    const uint32_t ldsBytes = trainedLdsBytes(ctx, waves, wordsPerWave, false);
    if (ldsBytes > ctx->ldsLimit) {
        return fail(MEMB_HIP_ERR_INVALID, "a row's bitstream does not fit into LDS");
    }
    params.wordsPerWave = wordsPerWave;
    const size_t tiles = (ctx->nRows + wordsPerWave - 1) / wordsPerWave;
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

// Whether a batch of this output shape runs dequant_uniform_tile (one tile per wavefront, row regions through LDS), and
// with what geometry: output pieces of 16 bytes, a tile's regions fitting LDS (blocks of four wavefronts, fewer for very
// wide rows), 32-bit piece numbers; option `persistent` = 0 keeps the block kernel (tests). launchUniform and
// memb_hip_ctx_get_info both ask here.
struct UniformTilePlan {
    bool tiled = false;
    uint32_t waves = 0;
    uint32_t tileWords = 0;
};

UniformTilePlan planUniform(const memb_hip_ctx* ctx, size_t ld, size_t colOff, const float* out)
{
    UniformTilePlan plan;
    const bool vec = (ctx->dim % 4 == 0) && (ld % 4 == 0) && (colOff % 4 == 0) &&
        (reinterpret_cast<uintptr_t>(out) % 16 == 0);
    plan.tileWords = std::max<uint32_t>(1, std::min<uint32_t>(WAVE, (9600 + ctx->dim * 4 - 1) / (ctx->dim * 4)));
    plan.waves = 4;
    while (plan.waves > 1 && uint64_t(plan.waves) * plan.tileWords * ctx->regionPieces * 16 > ctx->ldsLimit) {
        plan.waves /= 2;
    }
    plan.tiled = vec && ctx->switches.persistent != 0 && uint64_t(ctx->nRows + 1) * ctx->regionPieces < (1ull << 32) &&
        uint64_t(plan.waves) * plan.tileWords * ctx->regionPieces * 16 <= ctx->ldsLimit;
    return plan;
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
    params.records = ctx->uniformRecords;
    params.regionPieces = ctx->regionPieces;
    params.nRows = ctx->nRows;
    params.dim = ctx->dim;
    params.wordsPerBlock = rowwiseWordsPerBlock(ctx->dim);
    params.levels = ctx->levels;
    const bool vec = (ctx->dim % 4 == 0) && (ld % 4 == 0) && (colOff % 4 == 0) &&
        (reinterpret_cast<uintptr_t>(out) % 16 == 0);
    const UniformTilePlan tilePlan = planUniform(ctx, ld, colOff, out);
    const uint32_t tileWords = tilePlan.tileWords;
    const uint32_t waves = tilePlan.waves;
    if (tilePlan.tiled) {
        params.wordsPerWave = tileWords;
        params.regionMagic = magicFor(ctx->regionPieces, uint64_t(tileWords) * ctx->regionPieces + WAVE);
        params.pieceMagic = magicFor(ctx->dim / 4, uint64_t(tileWords) * (ctx->dim / 4));
        const uint32_t ldsBytes = waves * tileWords * ctx->regionPieces * 16;
        const uint64_t tiles = (n + tileWords - 1) / tileWords;
        const bool flat = ld == ctx->dim && colOff == 0;
        void (*kernel)(UniformParams) = flat ? &dequant_uniform_tile<true> : &dequant_uniform_tile<false>;
        hipError_t status;
        {
            std::lock_guard<std::mutex> lock(g_kernelFactsMutex);   // (raises the kernel's LDS limit on first use)
            KernelFacts* facts = nullptr;
            status = kernelFactsLocked(reinterpret_cast<const void*>(kernel), &facts);
        }
        if (status == hipSuccess) {
            hipLaunchKernelGGL(kernel, dim3(static_cast<uint32_t>((tiles + waves - 1) / waves)), dim3(waves * WAVE), ldsBytes, stream, params);
            status = hipGetLastError();
        }
        if (status != hipSuccess) {
            return fail(MEMB_HIP_ERR_DEVICE, std::string("dequant_uniform_tile launch: ") + hipGetErrorString(status));
        }
        return MEMB_HIP_OK;
    }
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

// Gives an allocation of deviceAlloc back before the context goes (optional structures whose build failed).
template <typename T>
void deviceRelease(memb_hip_ctx* ctx, T** pointer, size_t bytes)
{
    if (!*pointer) {
        return;
    }
    auto found = std::find(ctx->allocations.begin(), ctx->allocations.end(), static_cast<void*>(*pointer));
    if (found != ctx->allocations.end()) {
        ctx->allocations.erase(found);
        ctx->deviceBytes -= std::max<size_t>(bytes, 16);
    }
    (void)hipFree(*pointer);
    *pointer = nullptr;
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

// Makes the context's device current for the duration of an entry point and puts the caller's
// device back afterwards (the calling thread may belong to PyTorch, whose current device must not
// change under it).
class DeviceScope {
public:
    explicit DeviceScope(int device)
    {
        if (hipGetDevice(&previous_) != hipSuccess) {
            (void)hipGetLastError();
            previous_ = -1;
        }
        status_ = previous_ == device ? hipSuccess : hipSetDevice(device);
        changed_ = status_ == hipSuccess && previous_ != device;
    }
    ~DeviceScope()
    {
        if (changed_ && previous_ >= 0) {
            (void)hipSetDevice(previous_);
        }
    }
    DeviceScope(const DeviceScope&) = delete;
    DeviceScope& operator=(const DeviceScope&) = delete;
    hipError_t status() const { return status_; }

private:
    int previous_ = -1;
    hipError_t status_ = hipSuccess;
    bool changed_ = false;
};

// Puts the calling thread's current device back when it goes out of scope (context creation and
// destruction switch devices as they go).
class DeviceRestore {
public:
    DeviceRestore()
    {
        if (hipGetDevice(&previous_) != hipSuccess) {
            (void)hipGetLastError();
            previous_ = -1;
        }
    }
    ~DeviceRestore()
    {
        if (previous_ >= 0) {
            (void)hipSetDevice(previous_);
        }
    }
    DeviceRestore(const DeviceRestore&) = delete;
    DeviceRestore& operator=(const DeviceRestore&) = delete;

private:
    int previous_ = -1;
};

// (callers hold a DeviceRestore: the context's device stays current for the staging that follows)
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
    // the word the kernels of very large batches leave about their rows' order (noteBatchOrder); a context that cannot have
    // it runs the block size of key order whatever comes
    void* seen = nullptr;
    if (hipHostMalloc(&seen, 64, hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess) {
        void* device = nullptr;
        if (hipHostGetDevicePointer(&device, seen, 0) == hipSuccess) {
            ctx->orderSeen = static_cast<uint32_t*>(seen);
            ctx->orderSeenDevice = static_cast<uint32_t*>(device);
            *ctx->orderSeen = ORDER_UNKNOWN;
        } else {
            (void)hipHostFree(seen);
        }
    }
    (void)hipGetLastError();
    return MEMB_HIP_OK;
}

Switches readSwitches()
{
    Switches switches;
    switches.waves = envUint("MEMB_HIP_WAVES", 0);   // (also an option; as a variable it is known when the tables are sized: chooseLanes)
#ifdef MEMB_HIP_MEASURE
    switches.debugFlags = envUint("MEMB_HIP_DEBUG", 0);
#endif
    switches.sliceWords = envUint("MEMB_HIP_SLICE_WORDS", ~0u);
    switches.copyChunkRows = envUint("MEMB_HIP_COPY_CHUNK_ROWS", 0);
    switches.copyThreads = std::min<uint32_t>(envUint("MEMB_HIP_COPY_THREADS", 16), 64);
    switches.hostStreaming = std::min<uint32_t>(envUint("MEMB_HIP_HOST_STREAMING", 1), 2);
    switches.verbose = envUint("MEMB_HIP_VERBOSE", 0) != 0;
    return switches;
}

void destroy(memb_hip_ctx* ctx)
{
    if (!ctx) {
        return;
    }
    DeviceRestore restore;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) {
        (void)hipStreamSynchronize(ctx->stream);
    }
    for (void* allocation : ctx->allocations) {
        (void)hipFree(allocation);
    }
    if (ctx->orderSeen) {
        (void)hipHostFree(ctx->orderSeen);
    }
    if (ctx->smallHost) {
        (void)hipHostFree(ctx->smallHost);
    }
    if (ctx->hostRowsPinned) {
        (void)hipHostFree(ctx->hostRowsPinned);
    }
    for (int i = 0; i < memb_hip_ctx::RING; ++i) {
        if (ctx->ring[i]) {
            (void)hipHostFree(ctx->ring[i]);
        }
        if (ctx->ringEvents[i]) {
            (void)hipEventDestroy(ctx->ringEvents[i]);
        }
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

struct ContextGuard {
    explicit ContextGuard(memb_hip_ctx* ctx): ctx_(ctx) {}
    ~ContextGuard() { destroy(ctx_); }
    ContextGuard(const ContextGuard&) = delete;
    ContextGuard& operator=(const ContextGuard&) = delete;
    memb_hip_ctx* release()
    {
        memb_hip_ctx* ctx = ctx_;
        ctx_ = nullptr;
        return ctx;
    }

private:
    memb_hip_ctx* ctx_;
};

#include "hip_host_path.h"

}  // namespace

// Implementations of the entry points; the extern "C" functions at the end of the
// file call them behind an exception barrier.
namespace {



int device_count_checked(int* count)
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

// ---- staging of a trained storage, step by step (memb_hip_ctx_create_trained) ----

// Code lengths -> two-level lookup table (host form).
int buildHostTable(memb_hip_ctx* ctx, const memb_hip_trained_desc* desc)
{
    try {
        auto lengths = memb::codeLengthsFromSizeOffsets(
            desc->keys, desc->n_keys, desc->size_offsets, desc->n_size_offsets);
        for (const auto& info : lengths) {
            if (info.key >= desc->n_centroids) {
                throw std::runtime_error("Huffman symbol without a centroid");
            }
        }
        // First-level width: a hint (results never depend on it). With long codes a narrow first
        // level makes the second-level tables large -- 2^(longest code - width) entries each --
        // so the width is raised until the whole table takes at most 64 KiB of the 160 KiB of LDS,
        // which leaves room for the bitstream slots of at least a few wavefronts.
        uint32_t limit = desc->max_direct_bits ? desc->max_direct_bits : envUint("MEMB_HIP_ROOT_BITS", 11);
        limit = std::max<uint32_t>(1, std::min<uint32_t>(limit, 12));
        for (;;) {
            ctx->hostTable = memb::buildDecodeTable(lengths, limit);
            if (ctx->hostTable.entries.size() * 8 <= 64 * 1024 || limit >= 12) {
                break;
            }
            ++limit;
        }
        // The byte-key kernels' table. What costs there is the second-level lookup, not the size of
        // the first level: a branch and a second dependent LDS round trip per symbol as soon as one
        // lane of the wavefront needs it (6-bit GloVe-shaped model, codes up to 10 bits: 0.71 ms with
        // an 8-bit first level, 0.61 ms with a 10-bit one; 8-bit model, codes up to 13 bits: 0.74 /
        // 0.72 / 0.68 ms with 11 / 12 / 13 bits). So the first level covers the longest code
        // whenever 32 KiB of 4-byte entries hold it (13 bits); a given max_direct_bits is still
        // honoured (tests force the two-level path with it), raised only while the tables would not fit.
        ctx->codeLengths = lengths;
        uint32_t byteLimit = desc->max_direct_bits ? desc->max_direct_bits : 13u;
        byteLimit = std::max<uint32_t>(1, std::min<uint32_t>(byteLimit, 13));
        for (;;) {
            ctx->byteTable = memb::buildDecodeTable(lengths, byteLimit);
            if (ctx->byteTable.entries.size() * 4 <= 48 * 1024 || byteLimit >= 13) {
                break;
            }
            ++byteLimit;
        }
    } catch (const std::exception& error) {
        return fail(MEMB_HIP_ERR_INVALID, error.what());
    }

    return MEMB_HIP_OK;
}

// Per-row stream length: streams are laid out back to back, so a stream ends where the
// next one (in storage order) begins. Sets ctx->streamBytes and ctx->maxStreamBytes.
int measureStreams(memb_hip_ctx* ctx, const memb_hip_trained_desc* desc)
{
    // Every start is marked in a bitmap over the byte positions; each row then scans forward
    // to the next mark. Both passes run on a few host threads.
    {
        for (uint64_t r = 0; r < desc->n_rows; ++r) {
            if (desc->value_offsets[r] > desc->packed_values_bytes) {
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
    return MEMB_HIP_OK;
}

// Re-packed layout on the device: streamStarts, then the bitstreams themselves (repack_streams).
int stageStreams(memb_hip_ctx* ctx, const memb_hip_trained_desc* desc)
{
    // Row records (TrainedParams::recordPieces) where the model allows: at most 8 lanes per word and
    // streams below 1 KiB (the record's 13-bit offsets), and rows of similar length (the regions may take
    // 35 % more than the compact layout; the 2-bit model, rows of 1..3 bits per weight, needs 33 % and
    // still gains 1 % on the key-order dump and 3 % on shuffled rows) -- every row then
    // owns 16 + the longest stream bytes, rounded up to 32 so that a row never touches a third line.
    // Otherwise (or with MEMB_HIP_ROW_RECORDS=0) the compact layout: row r's stream occupies
    // ceil(bytes / 16) pieces from streamStarts[r], offsets in rowMeta or the two arrays.
    std::vector<uint32_t> streamStarts(desc->n_rows + 1, 0);
    uint64_t compactPieces = 0;
    for (uint64_t r = 0; r < desc->n_rows; ++r) {
        compactPieces += (ctx->streamBytes[r] + 15) / 16;
    }
    const uint32_t recordPieces = ((16 + ctx->maxStreamBytes + 31) / 32) * 2;
    ctx->recordPieces = 0;
    if (desc->n_rows && ctx->lanesPerWord > 1 && ctx->lanesPerWord <= ROW_META_MAX_LANES &&
        uint64_t(ctx->maxStreamBytes) * 8 + 64 < (1u << ROW_META_BITS) && envUint("MEMB_HIP_ROW_RECORDS", 1) &&
        envUint("MEMB_HIP_ROW_META", 1) && ctx->slotDwords >= 4 * recordPieces + 3 &&
        uint64_t(recordPieces) * desc->n_rows * 100 <=
            (compactPieces + desc->n_rows) * (100 + 35) &&   // at most 35 % padding
        uint64_t(recordPieces) * (desc->n_rows + 2) < (1ull << 32)) {
        ctx->recordPieces = recordPieces;
    } else {
        // compact layout: the slot holds the stream and the decoder's last window only (no record, no
        // rounding to 32), so lookups copy two pieces per word less and the wavefronts need less LDS
        ctx->slotDwords = compactSlotDwords(ctx->maxStreamBytes);
    }
    {
        uint64_t next = 0;
        for (uint64_t r = 0; r < desc->n_rows; ++r) {
            if (ctx->recordPieces) {
                streamStarts[r] = static_cast<uint32_t>(r * ctx->recordPieces + 1);   // the stream follows the record
                continue;
            }
            streamStarts[r] = static_cast<uint32_t>(next);
            next += (ctx->streamBytes[r] + 15) / 16;
        }
        if (ctx->recordPieces) {
            next = uint64_t(ctx->recordPieces) * desc->n_rows;
        }
        if (next + ctx->slotDwords / 4 + 1 >= (1ull << 32)) {
                return fail(MEMB_HIP_ERR_INVALID, "bitstreams too large");
        }
        streamStarts[desc->n_rows] = static_cast<uint32_t>(next);
    }
    const size_t streamPieces = size_t(streamStarts[desc->n_rows]) + ctx->slotDwords / 4 + 1;   // + guard of one slot

    // (Where the array lies does not matter: round 3 placed it at 1 GiB and 2 MiB alignments with 64 MiB / 4 KiB / 256 B
    // offsets, all within 0.25 % -- profiles/r03_aa_control.txt; the switches for that experiment are gone.)
    int code = deviceAlloc(ctx, &ctx->streams, streamPieces * 16);
    if (ctx->switches.verbose) {
        std::fprintf(stderr, "memb_hip: stream array at %p, %zu bytes\n", static_cast<void*>(ctx->streams), streamPieces * 16);
    }
    uint8_t* filePacked = nullptr;      // temporary device copies of the file's arrays
    uint32_t* fileOffsets = nullptr;
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
                desc->packed_values_bytes, fileOffsets, ctx->streamStarts, desc->n_rows, ctx->streams, ctx->recordPieces);
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
    return code;
}

// Lanes per word (G), symbols per lane (S) and the width of the segment index.
void chooseLanes(memb_hip_ctx* ctx, const memb_hip_trained_desc* desc)
{
    // MEMB_HIP_LANES (default 8, the measured
    // optimum for 300-dimensional rows) or, for rows so long that a wavefront's 64 / G
    // bitstream slots and symbol rows would not fit into LDS, the next power of two that does;
    // rows of fewer than 8 weights are not split.
    uint32_t lanes = std::max<uint32_t>(1, std::min<uint32_t>(envUint("MEMB_HIP_LANES", 8), WAVE));
    if (desc->dim < 8) {
        lanes = 1;
    }
    const uint32_t group = ctx->fast ? 8 : 4;
    // (byte keys: a first level wider than 11 bits is a luxury -- it goes first when LDS is short)
    auto narrowTable = [ctx] {
        if (ctx->fast || ctx->byteTable.rootBits <= 11) {
            return false;
        }
        ctx->byteTable = memb::buildDecodeTable(ctx->codeLengths, ctx->byteTable.rootBits - 1);
        return true;
    };
    for (;;) {
        ctx->segmentSymbols = std::max<uint32_t>(group, ((desc->dim + lanes - 1) / lanes + group - 1) / group * group);
        ctx->lanesPerWord = (desc->dim + ctx->segmentSymbols - 1) / ctx->segmentSymbols;
        if (lanes >= WAVE || trainedLdsBytes(ctx, 1, WAVE / ctx->lanesPerWord, true) <= ctx->ldsLimit) {
            break;
        }
        if (narrowTable()) {
            continue;
        }
        lanes = std::min<uint32_t>(WAVE, lanes < 8 ? 8 : 2 * lanes);
    }
    // a forced block size (MEMB_HIP_WAVES, tests) may still need the room
    while (!chooseGeometry(ctx, WAVE / ctx->lanesPerWord, ctx->dim, 0, nullptr).waves && narrowTable()) {
    }
    ctx->indexWide = uint64_t(ctx->maxStreamBytes) * 8 + 64 >= 65536;
}

// Lookup table and codebook in their device forms.
int stageTables(memb_hip_ctx* ctx, const memb_hip_trained_desc* desc)
{
    int code = MEMB_HIP_OK;
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
        // (setUpLds copies whole 16-byte pieces: the device copy is padded to packedTableDwords like the LDS image;
        // nibble-key models have it too: decode_union_split decodes them through it beside a byte-key partner)
        std::vector<uint32_t> padded(ctx->byteTable.entries);
        padded.resize(packedTableDwords(ctx), 0);
        code = deviceAlloc(ctx, &ctx->table32, padded.size() * 4);
        if (code == MEMB_HIP_OK) {
            code = copyToDevice(ctx->table32, padded.data(), padded.size() * 4);
        }
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
        ctx->hostCodebook = codebook;
        if (code == MEMB_HIP_OK && ctx->fast) {
            // the byte-key form of the codebook (see memb_hip_ctx::codebookBytes): 1 KiB
            code = deviceAlloc(ctx, &ctx->codebookBytes, 256 * 4);
            if (code == MEMB_HIP_OK) {
                code = copyToDevice(ctx->codebookBytes, centroids.data(), 256 * 4);
            }
        }
    }
    return code;
}

// Segment index (one decode pass over every row) and the rowMeta records packed from it.
int stageIndex(memb_hip_ctx* ctx, const memb_hip_trained_desc* desc)
{
    int code = MEMB_HIP_OK;
    if (code == MEMB_HIP_OK) {
        TrainedGeometry geometry = chooseGeometry(ctx, WAVE / ctx->lanesPerWord, ctx->dim, 0, nullptr);
        if (!geometry.waves) {
            code = fail(MEMB_HIP_ERR_INVALID, "decode tables and bitstream slots do not fit into LDS");
        }
    }
    if (code == MEMB_HIP_OK && ctx->lanesPerWord > 1) {
        code = deviceAlloc(
            ctx, &ctx->segmentIndex,
            size_t(desc->n_rows) * (ctx->lanesPerWord - 1) * (ctx->indexWide ? sizeof(uint32_t) : sizeof(uint16_t)));
    }
    if (code == MEMB_HIP_OK) {
        code = buildSegmentIndex(ctx, ctx->lanesPerWord, ctx->segmentSymbols, ctx->segmentIndex);
    }
    // Lookups read a row's stream start and segment offsets as one 16-byte record with one load
    // per lane: one request per word of a random batch (the two arrays cost two or three), one
    // per tile of a key-order dump. With row records the same record sits in front of the row's
    // stream instead and costs no request of its own.
    const bool records = ctx->recordPieces != 0;
    if (code == MEMB_HIP_OK && desc->n_rows && ctx->lanesPerWord > 1 && ctx->lanesPerWord <= ROW_META_MAX_LANES &&
        uint64_t(ctx->maxStreamBytes) * 8 + 64 < (1u << ROW_META_BITS) && (records || envUint("MEMB_HIP_ROW_META", 1))) {
        uint32_t* lengths = nullptr;   // per-row stream bytes, staging only
        if (records) {
            hipError_t status = hipMalloc(reinterpret_cast<void**>(&lengths), desc->n_rows * 4);
            if (status != hipSuccess) {
                code = fail(MEMB_HIP_ERR_DEVICE, std::string("hipMalloc: ") + hipGetErrorString(status));
            } else {
                code = copyToDevice(lengths, ctx->streamBytes.data(), desc->n_rows * 4);
            }
        } else {
            code = deviceAlloc(ctx, &ctx->rowMeta, size_t(desc->n_rows) * 16 + 16);
        }
        if (code == MEMB_HIP_OK) {
            const uint32_t threads = 256;
            hipLaunchKernelGGL(
                pack_row_meta, dim3(static_cast<uint32_t>((desc->n_rows + threads - 1) / threads)), dim3(threads), 0,
                ctx->stream, ctx->streamStarts, ctx->segmentIndex, ctx->lanesPerWord, desc->n_rows,
                records ? reinterpret_cast<uint32_t*>(ctx->streams) : ctx->rowMeta, ctx->recordPieces, lengths);
            hipError_t status = hipGetLastError();
            if (status == hipSuccess) {
                status = hipStreamSynchronize(ctx->stream);
            }
            if (status != hipSuccess) {
                code = fail(MEMB_HIP_ERR_DEVICE, std::string("pack_row_meta: ") + hipGetErrorString(status));
            }
        }
        if (lengths) {
            (void)hipFree(lengths);
        }
    }
    // The finer index of small batches (memb_hip_ctx::fineIndex): about sixteen lanes per word for models staged as row
    // records -- a second pass of the index kernel, 2 bytes per offset (53 MB for the 2.2 M-word model of dim 300: a
    // seventh of its footprint on a 288 GB part).
    if (code == MEMB_HIP_OK && records && desc->n_rows && !ctx->indexWide) {
        const uint32_t group = ctx->fast ? 8 : 4;
        const uint32_t wanted = 16;
        const uint32_t symbols = std::max<uint32_t>(group, ((desc->dim + wanted - 1) / wanted + group - 1) / group * group);
        const uint32_t lanes = (desc->dim + symbols - 1) / symbols;
        if (lanes > ctx->lanesPerWord && lanes <= WAVE) {
            // (an optimisation of small batches: a model that cannot have it -- no memory left, a failed pass -- is staged without)
            const size_t fineBytes = size_t(desc->n_rows) * (lanes - 1) * sizeof(uint16_t) + 16;
            int fine = deviceAlloc(ctx, &ctx->fineIndex, fineBytes);
            if (fine == MEMB_HIP_OK) {
                fine = buildSegmentIndex(ctx, lanes, symbols, ctx->fineIndex);
            }
            if (fine == MEMB_HIP_OK) {
                ctx->fineLanes = lanes;
                ctx->fineSymbols = symbols;
            } else {
                (void)hipGetLastError();
                deviceRelease(ctx, &ctx->fineIndex, fineBytes);
            }
        }
    }
    return code;
}

int ctx_create_trained_checked(memb_hip_ctx** out, int device, const memb_hip_trained_desc* desc)
{
    if (!out || !desc) {
        return fail(MEMB_HIP_ERR_INVALID, "null argument");
    }
    *out = nullptr;
    if (desc->dim == 0 || desc->n_keys == 0 || desc->n_keys > 256 || desc->n_centroids > 255 ||
        (desc->n_rows && !desc->value_offsets) || (desc->packed_values_bytes && !desc->packed_values)) {
        return fail(MEMB_HIP_ERR_INVALID, "inconsistent trained storage description");
    }

    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double tStart = now();
    DeviceRestore restore;
    memb_hip_ctx* ctx = new memb_hip_ctx();
    ContextGuard guard(ctx);   // destroys the context unless it is handed to the caller
    ctx->switches = readSwitches();
    ctx->storage = memb::wire::Storage_Trained;
    ctx->dim = desc->dim;
    ctx->nRows = desc->n_rows;

    // host side: tables and stream lengths (everything that can refuse a description does so
    // before a device is opened)
    int code = buildHostTable(ctx, desc);
    if (code == MEMB_HIP_OK) {
        code = measureStreams(ctx, desc);
    }
    if (code != MEMB_HIP_OK) {
        return code;
    }
    // (sized for row records first: stageStreams shrinks it when the model stays on the compact layout)
    ctx->slotDwords = recordSlotDwords(ctx->maxStreamBytes);
    ctx->fast = desc->n_centroids <= 16 && ctx->hostTable.maxCodeBits <= 8 && !ctx->hostTable.hasSubTables &&
        !envUint("MEMB_HIP_NO_FAST", 0);
    ctx->tableDwords = static_cast<uint32_t>((2 * ctx->hostTable.entries.size() + 3) / 4 * 4);
    const double tSorted = now();

    code = openDevice(ctx, device);
    if (code == MEMB_HIP_OK) {
        chooseLanes(ctx, desc);   // before the streams: the layout depends on the lanes per word
        code = stageStreams(ctx, desc);
    }
    const double tRepacked = now();
    if (code == MEMB_HIP_OK) {
        code = stageTables(ctx, desc);
    }
    if (code == MEMB_HIP_OK) {
        code = stageIndex(ctx, desc);
    }
    if (ctx->switches.verbose) {
        std::fprintf(stderr, "memb_hip: stage trained rows=%llu: host lengths %.3fs, device open + copy + repack %.3fs, tables + index %.3fs\n",
                     static_cast<unsigned long long>(desc->n_rows), tSorted - tStart, tRepacked - tSorted, now() - tRepacked);
    }
    if (code != MEMB_HIP_OK) {
        return code;
    }
    *out = guard.release();
    return MEMB_HIP_OK;
}

int ctx_create_uniform_checked(memb_hip_ctx** out, int device, const memb_hip_uniform_desc* desc)
{
    if (!out || !desc) {
        return fail(MEMB_HIP_ERR_INVALID, "null argument");
    }
    *out = nullptr;
    if (desc->dim == 0 || (desc->n_rows && !desc->rows)) {
        return fail(MEMB_HIP_ERR_INVALID, "inconsistent uniform storage description");
    }
    DeviceRestore restore;
    memb_hip_ctx* ctx = new memb_hip_ctx();
    ContextGuard guard(ctx);   // destroys the context unless it is handed to the caller
    ctx->switches = readSwitches();
    ctx->storage = memb::wire::Storage_Uniform;
    ctx->dim = desc->dim;
    ctx->nRows = desc->n_rows;
    ctx->levels = static_cast<float>(desc->quantization_levels);

    // HBM layout: ROW RECORDS (UniformParams::records) -- every row owns 16 bytes of {min, max} and its
    // weights, one byte each, zero padded to whole 16-byte pieces. The file scatters each row in its own
    // table; rows shorter than dim are zero padded (the reference writes only values->size() outputs there).
    ctx->regionPieces = 1 + (desc->dim + 15) / 16;
    const size_t regionBytes = size_t(ctx->regionPieces) * 16;
    std::vector<uint8_t> records((size_t(desc->n_rows) + 1) * regionBytes, 0);   // + one region of guard
    for (uint64_t r = 0; r < desc->n_rows; ++r) {
        const memb_hip_uniform_row& row = desc->rows[r];
        uint8_t* region = records.data() + r * regionBytes;
        const float header[2] = {row.min_value, row.max_value};
        std::memcpy(region, header, sizeof(header));
        size_t count = std::min<size_t>(row.n_values, desc->dim);
        if (count) {
            std::memcpy(region + 16, row.values, count);
        }
    }

    int code = openDevice(ctx, device);
    if (code == MEMB_HIP_OK) {
        code = deviceAlloc(ctx, &ctx->uniformRecords, records.size());
    }
    if (code == MEMB_HIP_OK) {
        code = copyToDevice(ctx->uniformRecords, records.data(), records.size());
    }
    if (code != MEMB_HIP_OK) {
        return code;
    }
    *out = guard.release();
    return MEMB_HIP_OK;
}

int ctx_create_full_checked(memb_hip_ctx** out, int device, const memb_hip_full_desc* desc)
{
    if (!out || !desc) {
        return fail(MEMB_HIP_ERR_INVALID, "null argument");
    }
    *out = nullptr;
    if (desc->dim == 0 || (desc->n_rows && !desc->rows)) {
        return fail(MEMB_HIP_ERR_INVALID, "inconsistent full storage description");
    }
    DeviceRestore restore;
    memb_hip_ctx* ctx = new memb_hip_ctx();
    ContextGuard guard(ctx);   // destroys the context unless it is handed to the caller
    ctx->switches = readSwitches();
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
        return code;
    }
    *out = guard.release();
    return MEMB_HIP_OK;
}


int fillInfo(const memb_hip_ctx* ctx, memb_hip_ctx_info* info, uint64_t batchWords);

int ctx_get_info_checked(const memb_hip_ctx* ctx, memb_hip_ctx_info* info)
{
    if (!ctx || !info) {
        return fail(MEMB_HIP_ERR_INVALID, "null argument");
    }
    // the caller's declaration of the struct may be shorter (or longer) than this library's
    const uint32_t callerSize = info->struct_size;
    if (callerSize < sizeof(uint32_t) || callerSize > 4096) {
        return fail(MEMB_HIP_ERR_INVALID, "memb_hip_ctx_info.struct_size must be set to sizeof(memb_hip_ctx_info)");
    }
    // ABI 4 declared union_kernel 8 bytes lower than ABI 3 and ABI 5 do (two dwords fewer in front of it): a client
    // built against that header would read the kernel name out of step. Its size gives it away.
    if (callerSize == offsetof(memb_hip_ctx_info, word_index_bytes) - 8) {
        return fail(MEMB_HIP_ERR_INVALID, "memb_hip_ctx_info: this is the ABI-4 layout; rebuild against include/memb_hip.h of ABI 5");
    }
    memb_hip_ctx_info filled;
    uint64_t batchWords = 0;
    if (callerSize >= offsetof(memb_hip_ctx_info, batch_words) + sizeof(uint64_t)) {
        batchWords = info->batch_words;
    }
    const int code = fillInfo(ctx, &filled, batchWords);
    filled.batch_words = batchWords;
    if (code != MEMB_HIP_OK) {
        return code;
    }
    const uint32_t bytes = std::min<uint32_t>(callerSize, sizeof(filled));
    filled.struct_size = bytes;
    std::memcpy(info, &filled, bytes);
    return MEMB_HIP_OK;
}

int option_set_checked(memb_hip_ctx* ctx, const char* name, uint64_t value)
{
    if (!ctx || !name) {
        return fail(MEMB_HIP_ERR_INVALID, "null argument");
    }
    const std::string key(name);
    std::lock_guard<std::mutex> lock(ctx->mutex);
    if (key == "waves_per_block" && value <= 16) {
        ctx->switches.waves = static_cast<uint32_t>(value);
    } else if (key == "tiles_per_wave" && value <= 64) {
        ctx->switches.tilesPerWave = static_cast<uint32_t>(value);
    } else if (key == "fine_lanes" && value <= 2) {
        ctx->switches.fineLanes = static_cast<uint32_t>(value);
    } else if (key == "union_fused" && value <= 1) {
        ctx->switches.unionFused = static_cast<uint32_t>(value);
    } else if (key == "union_split" && value <= 1) {
        ctx->switches.unionSplit = static_cast<uint32_t>(value);
    } else if (key == "persistent" && value <= 2) {
        ctx->switches.persistent = static_cast<uint32_t>(value);
    } else if (key == "host_expand" && value <= 1) {
        ctx->switches.hostExpand = value != 0;
#ifdef MEMB_HIP_MEASURE
    } else if (key == "debug" && value <= 0xFFFFFFFFull) {
        ctx->switches.debugFlags = static_cast<uint32_t>(value);
    } else if (key == "lds_pad" && value <= 128 * 1024) {
        ctx->switches.ldsPad = static_cast<uint32_t>(value);
#endif
    } else {
        return fail(MEMB_HIP_ERR_INVALID, "unknown option or value out of range: " + key);
    }
    return MEMB_HIP_OK;
}

int fillInfo(const memb_hip_ctx* ctx, memb_hip_ctx_info* info, uint64_t batchWords)
{
    std::memset(info, 0, sizeof(*info));
    info->device = ctx->device;
    info->storage = ctx->storage;
    info->dim = ctx->dim;
    info->n_rows = ctx->nRows;
    info->device_bytes = ctx->deviceBytes;
    info->word_index_bytes = ctx->wordIndexBytes;
    info->word_index_slots = ctx->wordSlots ? static_cast<uint32_t>(std::min<uint64_t>(uint64_t(ctx->wordSlotMask) + 1, 0xFFFFFFFFull)) : 0;   // (saturates: a table of 2^32 slots)
    info->word_index_keys = ctx->wordIndexKeys;
    if (ctx->storage == memb::wire::Storage_Trained) {
        const memb::DecodeTable& table = ctx->fast ? ctx->hostTable : ctx->byteTable;   // the lookup kernels' table
        info->root_bits = table.rootBits;
        info->max_code_bits = table.maxCodeBits;
        info->table_entries = static_cast<uint32_t>(table.entries.size());
        info->max_stream_bytes = ctx->maxStreamBytes;
        // the plan of a dense device-resident batch
        TrainedPlan plan;
        {
            DeviceScope deviceScope(ctx->device);
            HIP_TRY(deviceScope.status());
            const int planned = planTrained(
                ctx, batchWords ? size_t(batchWords) : size_t(1) << 30, ctx->dim, 0, nullptr, false, &plan, -1, true, rowsUnordered(ctx, false));
            if (planned != MEMB_HIP_OK) {
                return planned;
            }
        }
        const TrainedGeometry geometry = plan.geometry;
        info->waves_per_block = geometry.waves;
        std::snprintf(info->union_kernel, sizeof(info->union_kernel), "%s", ctx->unionKernel);
        info->kernel_registers = static_cast<uint32_t>(plan.numRegs);
        info->register_waves_per_cu = plan.persistent ? plan.registerWavesPerCu : 0;
        info->lanes_per_word = plan.fine ? ctx->fineLanes : ctx->lanesPerWord;
        info->segment_symbols = plan.fine ? ctx->fineSymbols : ctx->segmentSymbols;
        info->lds_bytes_per_block = geometry.ldsBytes;
        info->row_layout = ctx->recordPieces ? 2u : (ctx->rowMeta ? 1u : 0u);
        info->row_bytes = ctx->recordPieces * 16;
        // template arguments as in the symbol: <two-level table, output mode (2 = dense rows), nibble keys, ...>
        std::snprintf(
            info->kernel, sizeof(info->kernel), "%s<%s, %d, %s>",
            plan.persistent ? "decode_records_persistent" : "decode_trained",
            (ctx->fast ? ctx->hostTable : ctx->byteTable).hasSubTables ? "true" : "false", static_cast<int>(OUT_FLAT),
            ctx->fast ? "true" : "false");
        if (!plan.persistent) {
            const uint32_t wordsPerWave = WAVE / info->lanes_per_word;
            const uint64_t words = batchWords ? batchWords : uint64_t(1) << 30;
            info->tiles_per_wavefront = oneTileSteps(
                ctx, (words + wordsPerWave - 1) / wordsPerWave,
                4u * ((ctx->fast ? ctx->tableDwords : packedTableDwords(ctx)) + codebookDwords(ctx)), false);
        }
    } else {
        // (a dense, aligned device-resident batch: what bench.py and the tests report for)
        const UniformTilePlan tilePlan = ctx->storage == memb::wire::Storage_Uniform ? planUniform(ctx, ctx->dim, 0, nullptr) : UniformTilePlan();
        const bool tiled = tilePlan.tiled;
        info->waves_per_block = tiled ? tilePlan.waves : ROWWISE_THREADS / WAVE;
        info->lds_bytes_per_block = tiled ? tilePlan.waves * tilePlan.tileWords * ctx->regionPieces * 16 : 0;
        std::snprintf(
            info->kernel, sizeof(info->kernel), "%s<true>",
            tiled ? "dequant_uniform_tile" : ctx->storage == memb::wire::Storage_Uniform ? "dequant_uniform" : "gather_full");
    }
    return MEMB_HIP_OK;
}

int decode_rows_device_checked(
    memb_hip_ctx* ctx, const uint32_t* rows, size_t n, float* out, size_t ld, size_t col_off, void* stream)
{
    if (!ctx || (n && (!rows || !out))) {
        return fail(MEMB_HIP_ERR_INVALID, "null argument");
    }
    if (ld < col_off + ctx->dim) {
        return fail(MEMB_HIP_ERR_INVALID, "ld must be at least col_off + dim");
    }
    DeviceScope deviceScope(ctx->device);
    HIP_TRY(deviceScope.status());
    return launch(ctx, rows, n, out, ld, col_off, static_cast<hipStream_t>(stream));
}

int decode_rows_device_ex_checked(
    memb_hip_ctx* ctx, const uint32_t* rows, size_t n, float* out, size_t ld, size_t col_off, void* stream,
    uint32_t flags, float divisor)
{
    if (!ctx || (n && (!rows || !out))) {
        return fail(MEMB_HIP_ERR_INVALID, "null argument");
    }
    if (ld < col_off + ctx->dim) {
        return fail(MEMB_HIP_ERR_INVALID, "ld must be at least col_off + dim");
    }
    if ((flags & ~uint32_t(MEMB_HIP_ACCUMULATE | MEMB_HIP_ROWS_IN_RANDOM_ORDER)) || !(divisor == divisor)) {
        return fail(MEMB_HIP_ERR_INVALID, "unknown flags or NaN divisor");
    }
    DeviceScope deviceScope(ctx->device);
    HIP_TRY(deviceScope.status());
    Epilogue epilogue;
    epilogue.accumulate = (flags & MEMB_HIP_ACCUMULATE) ? 1u : 0u;
    epilogue.divisor = divisor;
    epilogue.randomOrder = (flags & MEMB_HIP_ROWS_IN_RANDOM_ORDER) != 0;
    return launch(ctx, rows, n, out, ld, col_off, static_cast<hipStream_t>(stream), epilogue);
}

int decode_batches_device_checked(memb_hip_ctx* ctx, const memb_hip_batch* batches, size_t count, void* stream)
{
    if (!ctx || (count && !batches)) {
        return fail(MEMB_HIP_ERR_INVALID, "null argument");
    }
    std::vector<memb_hip_batch> live;   // (empty batches take no part)
    for (size_t k = 0; k < count; ++k) {
        const memb_hip_batch& batch = batches[k];
        if (batch.n && (!batch.rows || !batch.out)) {
            return fail(MEMB_HIP_ERR_INVALID, "null argument");
        }
        if (batch.ld < batch.col_off + ctx->dim) {
            return fail(MEMB_HIP_ERR_INVALID, "ld must be at least col_off + dim");
        }
        if (batch.n > (size_t(1) << 37)) {
            return fail(MEMB_HIP_ERR_INVALID, "batch too large");
        }
        if (batch.n) {
            live.push_back(batch);
        }
    }
    DeviceScope deviceScope(ctx->device);
    HIP_TRY(deviceScope.status());
    if (ctx->storage != memb::wire::Storage_Trained || live.size() == 1) {
        // uniform / full: their kernels have no shared prologue worth one launch; a single batch: the plain call
        for (const memb_hip_batch& batch : live) {
            const int code = launch(ctx, batch.rows, batch.n, batch.out, batch.ld, batch.col_off, static_cast<hipStream_t>(stream));
            if (code != MEMB_HIP_OK) {
                return code;
            }
        }
        return MEMB_HIP_OK;
    }
    for (size_t first = 0; first < live.size(); first += MAX_BATCHES) {
        const int code = launchTrainedBatches(
            ctx, live.data() + first, std::min<size_t>(MAX_BATCHES, live.size() - first), static_cast<hipStream_t>(stream));
        if (code != MEMB_HIP_OK) {
            return code;
        }
    }
    return MEMB_HIP_OK;
}

// (hip_words.h, included below: the lookup of a committed word batch into device memory, enqueued on the context's stream)
int resolveBatchOnContextStream(memb_hip_ctx* ctx, const memb_hip_words* batch, uint32_t* rowsDevice);
size_t wordBatchCount(const memb_hip_words* batch);

// memb_hip_decode_rows (host row ids in `rows`) and memb_hip_decode_words (`batch`: the row ids are looked up on the
// device and fetched into pinned memory for the host threads that zero the rows of unknown words).
int decode_rows_checked(
    memb_hip_ctx* ctx, const uint32_t* rows, size_t n, float* out, size_t ld, size_t col_off, const memb_hip_words* batch = nullptr)
{
    if (batch) {
        n = wordBatchCount(batch);
    }
    if (!ctx || (n && ((!rows && !batch) || !out))) {
        return fail(MEMB_HIP_ERR_INVALID, "null argument");
    }
    if (ld < col_off + ctx->dim) {
        return fail(MEMB_HIP_ERR_INVALID, "ld must be at least col_off + dim");
    }
    if (n == 0) {
        return MEMB_HIP_OK;
    }
    std::lock_guard<std::mutex> lock(ctx->mutex);
    DeviceScope deviceScope(ctx->device);
    HIP_TRY(deviceScope.status());

    // Small batches (single words above all): two tiny copies cost more than the
    // decode. Row ids and results go through one pinned, device-mapped host
    // buffer instead: the kernel reads the ids from it and writes the rows into it
    // over PCIe, the host then copies them to the caller's (possibly strided) rows.
    constexpr size_t SMALL_WORDS = 512;
    const size_t rowBytes = size_t(ctx->dim) * sizeof(float);
    // (at most 2 MiB of rows that way: very wide rows leave the small path after fewer words)
    const size_t smallWords = std::min<size_t>(SMALL_WORDS, (size_t(2) << 20) / rowBytes);
    if (n <= smallWords && !ctx->smallUnavailable && !batch) {
        // row ids first, rows from the next 256-byte boundary
        constexpr size_t outOffset = (SMALL_WORDS * sizeof(uint32_t) + 255) / 256 * 256;
        if (!ctx->smallHost) {
            void* host = nullptr;
            void* device = nullptr;
            if (hipHostMalloc(&host, outOffset + smallWords * rowBytes, hipHostMallocMapped) == hipSuccess &&
                hipHostGetDevicePointer(&device, host, 0) == hipSuccess) {
                ctx->smallHost = host;
                ctx->smallDevice = device;
            } else {
                (void)hipGetLastError();
                if (host) {
                    (void)hipHostFree(host);
                }
                ctx->smallUnavailable = true;
            }
        }
        if (ctx->smallHost) {
            std::memcpy(ctx->smallHost, rows, n * sizeof(uint32_t));
            float* deviceOut = reinterpret_cast<float*>(static_cast<char*>(ctx->smallDevice) + outOffset);
            int code = launch(ctx, static_cast<const uint32_t*>(ctx->smallDevice), n, deviceOut, ctx->dim, 0, ctx->stream);
            if (code != MEMB_HIP_OK) {
                return code;
            }
            HIP_TRY(hipStreamSynchronize(ctx->stream));
            const char* hostOut = static_cast<const char*>(ctx->smallHost) + outOffset;
            for (size_t i = 0; i < n; ++i) {
                std::memcpy(out + i * ld + col_off, hostOut + i * rowBytes, rowBytes);
            }
            return MEMB_HIP_OK;
        }
    }

    // Device staging holds one row per word -- fp32, or centroid indices for trained
    // storages (decodeRowsAsKeys) -- ; batches larger than the staging area (4 GiB of
    // fp32 rows) are processed in slices.
    const size_t dim = ctx->dim;
    const bool asKeys = ctx->storage == memb::wire::Storage_Trained && !ctx->hostCodebook.empty() &&
        keyRowBytes(ctx) <= RING_CHUNK_BYTES && ctx->switches.hostExpand && ensureRing(ctx);
    const size_t stagedRowBytes = asKeys ? keyRowBytes(ctx) : dim * sizeof(float);
    const size_t sliceLimit = std::min<size_t>((size_t(4) << 30) / (dim * sizeof(float)), ctx->switches.sliceWords);
    const size_t sliceWords = std::max<size_t>(1, std::min<size_t>(n, sliceLimit));
    if (batch && sliceWords < n) {
        return fail(MEMB_HIP_UNSUPPORTED, "a word batch of this size is decoded in slices: look the words up first and pass row ids");
    }
    if (batch && ctx->hostRowsCapacity < n) {
        if (ctx->hostRowsPinned) {
            (void)hipHostFree(ctx->hostRowsPinned);
            ctx->hostRowsPinned = nullptr;
            ctx->hostRowsCapacity = 0;
        }
        void* pinned = nullptr;
        HIP_TRY(hipHostMalloc(&pinned, (n + n / 4) * sizeof(uint32_t), hipHostMallocDefault));
        ctx->hostRowsPinned = static_cast<uint32_t*>(pinned);
        ctx->hostRowsCapacity = n + n / 4;
    }
    if (ctx->stagedCapacity < sliceWords || ctx->stagedRowBytes < stagedRowBytes) {
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
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&ctx->stagedOut), sliceWords * stagedRowBytes + 16));
        ctx->stagedCapacity = sliceWords;
        ctx->stagedRowBytes = stagedRowBytes;
    }
    const bool verbose = ctx->switches.verbose;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = now();
    int result = MEMB_HIP_OK;
    for (size_t start = 0; start < n && result == MEMB_HIP_OK; start += sliceWords) {
        const size_t words = std::min(sliceWords, n - start);
        hipError_t status;
        if (batch) {
            // word -> row on the device, straight into the staging array; the ids also go to pinned host memory, for the
            // threads that expand the result (an unknown word's row is zeroed there). Every copy of result chunks is
            // enqueued behind this one, so the ids have landed when the first chunk's event fires.
            result = resolveBatchOnContextStream(ctx, batch, ctx->stagedRows);
            if (result != MEMB_HIP_OK) {
                break;
            }
            status = hipMemcpyAsync(ctx->hostRowsPinned, ctx->stagedRows, words * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream);
            rows = ctx->hostRowsPinned;
        } else {
            status = hipMemcpyAsync(ctx->stagedRows, rows + start, words * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream);
        }
        if (status != hipSuccess) {
            result = fail(MEMB_HIP_ERR_DEVICE, std::string("row id copy: ") + hipGetErrorString(status));
            break;
        }
        float* destination = out + start * ld + col_off;
        if (asKeys) {
            result = decodeRowsAsKeys(
                ctx, ctx->stagedRows, rows + start, words, reinterpret_cast<uint8_t*>(ctx->stagedOut), destination, ld);
        } else {
            result = launch(ctx, ctx->stagedRows, words, ctx->stagedOut, dim, 0, ctx->stream);
            if (result == MEMB_HIP_OK) {
                result = copyRowsToHost(ctx, ctx->stagedOut, words, destination, ld);
            }
        }
    }
    if (result != MEMB_HIP_OK) {
        (void)hipStreamSynchronize(ctx->stream);   // nothing of this call may still be running when the caller's buffers go away
    }
    // a full-vocabulary dump should not keep gigabytes of HBM for the next small batch
    if (ctx->stagedCapacity * ctx->stagedRowBytes > (size_t(1) << 30)) {
        (void)hipFree(ctx->stagedRows);
        (void)hipFree(ctx->stagedOut);
        ctx->stagedRows = nullptr;
        ctx->stagedOut = nullptr;
        ctx->stagedCapacity = 0;
        ctx->stagedRowBytes = 0;
    }
    if (verbose) {
        std::fprintf(stderr, "memb_hip: decode_rows n=%zu (%s over PCIe) rows + kernel + copy %.4fs\n",
                     n, asKeys ? "centroid indices" : "fp32 rows", now() - t0);
    }
    return result;
}

int decode_rows_union_device_checked(
    memb_hip_ctx* const* ctxs, const uint32_t* const* rows, const size_t* col_offs, size_t count, size_t n, float* out,
    size_t ld, void* stream, uint32_t flags)
{
    if (flags & ~uint32_t(MEMB_HIP_UNION_AVERAGE)) {
        return fail(MEMB_HIP_ERR_INVALID, "unknown flags");
    }
    if (!ctxs || !rows || !col_offs || count == 0 || (n && !out)) {
        return fail(MEMB_HIP_ERR_INVALID, "null argument");
    }
    for (size_t m = 0; m < count; ++m) {
        if (!ctxs[m] || (n && !rows[m])) {
            return fail(MEMB_HIP_ERR_INVALID, "null argument");
        }
    }
    if (n == 0) {
        return MEMB_HIP_OK;
    }
    DeviceScope deviceScope(ctxs[0]->device);
    HIP_TRY(deviceScope.status());
    return launchTrainedUnion(
        ctxs, rows, col_offs, count, n, out, ld, static_cast<hipStream_t>(stream), (flags & MEMB_HIP_UNION_AVERAGE) != 0);
}

int sync_checked(memb_hip_ctx* ctx)
{
    if (!ctx) {
        return fail(MEMB_HIP_ERR_INVALID, "null argument");
    }
    DeviceScope deviceScope(ctx->device);
    HIP_TRY(deviceScope.status());
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return MEMB_HIP_OK;
}

int algorithmic_bytes_checked(const memb_hip_ctx* ctx, const uint32_t* rows, size_t n, uint64_t* bytes)
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

}  // namespace

#include "hip_encoder.h"
#include "hip_words.h"

namespace {

// No C++ exception leaves the library: allocation failures and the like become error codes.
template <typename Call>
int guarded(Call call)
{
    try {
        return call();
    } catch (const std::bad_alloc&) {
        return fail(MEMB_HIP_ERR_DEVICE, "out of host memory");
    } catch (const std::exception& error) {
        return fail(MEMB_HIP_ERR_DEVICE, std::string("internal error: ") + error.what());
    } catch (...) {
        return fail(MEMB_HIP_ERR_DEVICE, "internal error");
    }
}

}  // namespace

extern "C" {

const char* memb_hip_last_error(void)
{
    return g_lastError.c_str();
}

void memb_hip_ctx_destroy(memb_hip_ctx* ctx)
{
    destroy(ctx);
}

int memb_hip_device_count(int* count)
{
    return guarded([&] { return device_count_checked(count); });
}

int memb_hip_ctx_create_trained(memb_hip_ctx** out, int device, const memb_hip_trained_desc* desc)
{
    return guarded([&] { return ctx_create_trained_checked(out, device, desc); });
}

int memb_hip_ctx_create_uniform(memb_hip_ctx** out, int device, const memb_hip_uniform_desc* desc)
{
    return guarded([&] { return ctx_create_uniform_checked(out, device, desc); });
}

int memb_hip_ctx_create_full(memb_hip_ctx** out, int device, const memb_hip_full_desc* desc)
{
    return guarded([&] { return ctx_create_full_checked(out, device, desc); });
}

int memb_hip_ctx_get_info(const memb_hip_ctx* ctx, memb_hip_ctx_info* info)
{
    return guarded([&] { return ctx_get_info_checked(ctx, info); });
}

int memb_hip_encoder_create(memb_hip_encoder** encoder, int device, uint32_t dim, const float* split_points, uint32_t n_split_points)
{
    return guarded([&] { return encoder_create_checked(encoder, device, dim, split_points, n_split_points); });
}

void memb_hip_encoder_destroy(memb_hip_encoder* encoder)
{
    destroyEncoder(encoder);
}

int memb_hip_encoder_add_rows(memb_hip_encoder* encoder, const float* rows, size_t n_rows)
{
    return guarded([&] { return encoder_add_rows_checked(encoder, rows, n_rows); });
}

int memb_hip_encoder_counts(memb_hip_encoder* encoder, uint64_t* counts)
{
    return guarded([&] { return encoder_counts_checked(encoder, counts); });
}

int memb_hip_encoder_rows(memb_hip_encoder* encoder, uint64_t* n_rows)
{
    return guarded([&] { return encoder_rows_checked(encoder, n_rows); });
}

int memb_hip_encoder_pack(
    memb_hip_encoder* encoder, const uint16_t* codes, const uint8_t* lengths, uint32_t* stream_bytes, uint64_t* total_bytes)
{
    return guarded([&] { return encoder_pack_checked(encoder, codes, lengths, stream_bytes, total_bytes); });
}

int memb_hip_encoder_fetch(memb_hip_encoder* encoder, uint8_t* packed, uint64_t capacity)
{
    return guarded([&] { return encoder_fetch_checked(encoder, packed, capacity); });
}

int memb_hip_abi_version(void)
{
    return MEMB_HIP_ABI_VERSION;
}

int memb_hip_ctx_set_option(memb_hip_ctx* ctx, const char* name, uint64_t value)
{
    return guarded([&] { return option_set_checked(ctx, name, value); });
}

int memb_hip_decode_rows_device(memb_hip_ctx* ctx, const uint32_t* rows, size_t n, float* out, size_t ld, size_t col_off, void* stream)
{
    return guarded([&] { return decode_rows_device_checked(ctx, rows, n, out, ld, col_off, stream); });
}

int memb_hip_decode_rows_device_ex(memb_hip_ctx* ctx, const uint32_t* rows, size_t n, float* out, size_t ld, size_t col_off, void* stream, uint32_t flags, float divisor)
{
    return guarded([&] { return decode_rows_device_ex_checked(ctx, rows, n, out, ld, col_off, stream, flags, divisor); });
}

int memb_hip_decode_batches_device(memb_hip_ctx* ctx, const memb_hip_batch* batches, size_t count, void* stream)
{
    return guarded([&] { return decode_batches_device_checked(ctx, batches, count, stream); });
}

int memb_hip_decode_rows(memb_hip_ctx* ctx, const uint32_t* rows, size_t n, float* out, size_t ld, size_t col_off)
{
    return guarded([&] { return decode_rows_checked(ctx, rows, n, out, ld, col_off); });
}

int memb_hip_decode_words(memb_hip_ctx* ctx, const memb_hip_words* batch, float* out, size_t ld, size_t col_off)
{
    return guarded([&] {
        if (!ctx || !batch) {
            return fail(MEMB_HIP_ERR_INVALID, "null argument");
        }
        return decode_rows_checked(ctx, nullptr, 0, out, ld, col_off, batch);
    });
}

int memb_hip_decode_rows_union_device(
    memb_hip_ctx* const* ctxs, const uint32_t* const* rows, const size_t* col_offs, size_t count, size_t n, float* out,
    size_t ld, void* stream, uint32_t flags)
{
    return guarded([&] { return decode_rows_union_device_checked(ctxs, rows, col_offs, count, n, out, ld, stream, flags); });
}

int memb_hip_sync(memb_hip_ctx* ctx)
{
    return guarded([&] { return sync_checked(ctx); });
}

int memb_hip_ctx_stage_words(
    memb_hip_ctx* ctx, const char* packed_words, uint64_t packed_bytes, const uint32_t* word_offsets, uint64_t n_words)
{
    return guarded([&] { return stage_words_checked(ctx, packed_words, packed_bytes, word_offsets, n_words); });
}

int memb_hip_words_create(memb_hip_words** words, int device)
{
    return guarded([&] { return words_create_checked(words, device); });
}

void memb_hip_words_destroy(memb_hip_words* words)
{
    destroyWords(words);
}

int memb_hip_words_begin(memb_hip_words* batch, size_t n, size_t bytes_per_word, memb_hip_words_plan* plan)
{
    return guarded([&] { return words_begin_checked(batch, n, bytes_per_word, plan); });
}

int memb_hip_words_commit(memb_hip_words* batch)
{
    return guarded([&] { return words_commit_checked(batch); });
}

int memb_hip_words_pack(memb_hip_words* batch, const char* const* words, const uint32_t* lengths, size_t n)
{
    return guarded([&] { return words_pack_checked(batch, words, lengths, n); });
}

int memb_hip_resolve_range_device(
    memb_hip_ctx* ctx, const memb_hip_words* batch, size_t first_word, size_t n_words, uint32_t* rows_dev, void* stream)
{
    return guarded([&] { return resolve_range_device_checked(ctx, batch, first_word, n_words, rows_dev, static_cast<hipStream_t>(stream)); });
}

int memb_hip_words_count(const memb_hip_words* batch, size_t* n)
{
    return guarded([&] { return words_count_checked(batch, n); });
}

int memb_hip_resolve_rows_device(memb_hip_ctx* ctx, const memb_hip_words* batch, uint32_t* rows_dev, void* stream)
{
    return guarded([&] { return resolve_rows_device_checked(ctx, batch, rows_dev, static_cast<hipStream_t>(stream)); });
}

int memb_hip_resolve_range_union_device(
    memb_hip_ctx* const* ctxs, size_t count, const memb_hip_words* batch, size_t first_word, size_t n_words,
    uint32_t* const* rows_dev, void* stream)
{
    return guarded([&] {
        return resolve_range_union_device_checked(ctxs, count, batch, first_word, n_words, rows_dev, static_cast<hipStream_t>(stream));
    });
}

int memb_hip_resolve_packed_device(
    memb_hip_ctx* ctx, const uint8_t* bytes_dev, const uint32_t* offsets_dev, size_t n, uint32_t* rows_dev, void* stream)
{
    return guarded([&] { return resolve_packed_device_checked(ctx, bytes_dev, offsets_dev, n, rows_dev, static_cast<hipStream_t>(stream)); });
}

int memb_hip_algorithmic_bytes(const memb_hip_ctx* ctx, const uint32_t* rows, size_t n, uint64_t* bytes)
{
    return guarded([&] { return algorithmic_bytes_checked(ctx, rows, n, bytes); });
}

}  // extern "C"

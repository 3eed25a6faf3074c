// libmemb_ceilings.so -- what this GPU does with the decoder's MEMORY pattern and nothing else.
//
// Measurement code, not product: bench.py loads it (ctypes) to put the box's own ceilings next to the
// kernel's time in the JSON line (`roofline.box_ceilings`); tools/perf/r4/small_ceilings.py and store_patterns.py
// run the same entry points at other sizes and shapes.
// No decode, no tables; every pattern writes `words` rows of 300 floats (the 2.2 M-word dump: 2.635 GB)
// and, from pattern 2 on, reads what a decoder of row records reads -- the stored values DEPEND on the
// loaded bytes (through LDS, as in the decoder), so no load can be dropped or overtaken by its tile's stores.
//
//   0  linear fill: one 16-byte store per thread, the wavefront exits (the best write pattern this part has)
//   1  one tile per wavefront: 8 rows = 9600 B per wavefront (10 stores per lane), then exit   -- decode_trained's stores
//   2  1 + the tile's 8 records read first, SEQUENTIAL rows (8 x 160 B = 1280 consecutive bytes) -- a key-order dump
//   3  1 + the tile's 8 records at RANDOM rows (row ids from an array; 160-byte records at 32-byte
//      alignment: two 128-byte lines each)                                                      -- random / shuffled rows
//   4  persistent tiles: 16 wavefronts per CU walk the tiles, write only                        -- a persistent pipeline's stores
//   5  4 + sequential records, the next tile's loads in flight while this tile is stored
//   6  4 + random records, same prefetch
//   7  union shape: a tile is 4 merged rows of 600 floats (9600 B), 8 records at random rows of TWO arrays
//   8, 9  uniform-storage shapes (320-byte records): one row per wavefront / eight rows per wavefront
//   10, 11  5 / 6 with a grid of tiles / 2 wavefronts instead of a resident one: two tiles per wavefront, half a batch apart, the
//      second tile's records in flight during the first tile's stores, then exit (round 5, batch 28: the fastest tile pattern found)
//
// Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC (build_native.py). gfx950 only.
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {

constexpr uint32_t WAVE = 64;
constexpr uint32_t ROW_FLOATS = 300;
constexpr uint32_t TILE_ROWS = 8;
constexpr uint32_t TILE_PIECES = TILE_ROWS * ROW_FLOATS / 4;   // 600 16-byte pieces = 9600 B
constexpr uint32_t RECORD_PIECES = 10;                         // 160-byte row regions
constexpr uint32_t TILE_RECORD_PIECES = TILE_ROWS * RECORD_PIECES;   // 80

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

struct Params {
    float* out;
    unsigned long long words;
    const u32x4* records;        // row r at piece r * RECORD_PIECES
    const u32x4* records2;       // pattern 7: the second model's
    unsigned long long rows;     // rows in `records`
    const uint32_t* ids;         // random patterns: row of every word (>= rows: absent, nothing is loaded)
    const uint32_t* ids2;
};

__global__ void fill_linear(float4* out, unsigned long long pieces)
{
    const unsigned long long i = static_cast<unsigned long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < pieces) {
        out[i] = make_float4(1.f, 2.f, 3.f, 4.f);
    }
}

// the tile's record pieces: piece q of 80 -> (word q / 10, piece q % 10); two rounds of 64 lanes
template <bool RANDOM>
__device__ __forceinline__ void loadTile(
    const Params& p, unsigned long long tile, uint32_t lane, u32x4& a, u32x4& b, const u32x4* records, const uint32_t* ids)
{
    a = u32x4{0, 0, 0, 0};
    b = u32x4{0, 0, 0, 0};
#pragma unroll
    for (int round = 0; round < 2; ++round) {
        const uint32_t q = round * WAVE + lane;
        if (q < TILE_RECORD_PIECES) {
            const uint32_t w = q / RECORD_PIECES;
            const uint32_t piece = q - w * RECORD_PIECES;
            const unsigned long long word = tile * TILE_ROWS + w;
            unsigned long long row = word;
            if (RANDOM) {
                row = word < p.words ? ids[word] : 0xFFFFFFFFull;
            }
            if (row < p.rows) {
                const u32x4 v = records[row * RECORD_PIECES + piece];
                if (round == 0) {
                    a = v;
                } else {
                    b = v;
                }
            }
        }
    }
}

__device__ __forceinline__ void stageTile(uint32_t* slots, uint32_t lane, const u32x4& a, const u32x4& b)
{
    *reinterpret_cast<u32x4*>(slots + 4 * lane) = a;
    if (lane + WAVE < TILE_RECORD_PIECES) {
        *reinterpret_cast<u32x4*>(slots + 4 * (lane + WAVE)) = b;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// 600 pieces, 64 lanes: the decoder's output phase (values out of LDS, 16 B per lane, 1 KiB per instruction)
template <bool READS>
__device__ __forceinline__ void storeTile(float* out, unsigned long long words, unsigned long long tile, uint32_t lane, const uint32_t* slots)
{
    const unsigned long long firstPiece = tile * TILE_PIECES;
    const unsigned long long endPiece = words * (ROW_FLOATS / 4);
    float4* tileOut = reinterpret_cast<float4*>(out) + firstPiece;
    for (uint32_t q = lane; q < TILE_PIECES; q += WAVE) {
        float4 value = make_float4(1.f, 2.f, 3.f, 4.f);
        if (READS) {
            const uint32_t key = slots[q % (4 * TILE_RECORD_PIECES)];
            value.x = __uint_as_float(key & 0x3f800000u);
        }
        if (firstPiece + q < endPiece) {
            tileOut[q] = value;
        }
    }
}

// MODE 0 = write only, 1 = sequential records, 2 = random records
template <int MODE>
__global__ void tile_per_wave(Params p)
{
    __shared__ __attribute__((aligned(16))) uint32_t lds[4 * 4 * TILE_RECORD_PIECES];   // four wavefronts per block
    const uint32_t lane = threadIdx.x & (WAVE - 1);
    const uint32_t wave = threadIdx.x / WAVE;
    const unsigned long long tile = static_cast<unsigned long long>(blockIdx.x) * (blockDim.x / WAVE) + wave;
    if (tile * TILE_ROWS >= p.words) {
        return;
    }
    uint32_t* slots = lds + wave * 4 * TILE_RECORD_PIECES;
    if (MODE) {
        u32x4 a, b;
        loadTile<MODE == 2>(p, tile, lane, a, b, p.records, p.ids);
        stageTile(slots, lane, a, b);
    }
    storeTile<MODE != 0>(p.out, p.words, tile, lane, slots);
}

template <int MODE>
__global__ void tiles_persistent(Params p)
{
    __shared__ __attribute__((aligned(16))) uint32_t lds[4 * 4 * TILE_RECORD_PIECES];
    const uint32_t lane = threadIdx.x & (WAVE - 1);
    const uint32_t wave = threadIdx.x / WAVE;
    const unsigned long long stride = static_cast<unsigned long long>(gridDim.x) * (blockDim.x / WAVE);
    const unsigned long long tiles = (p.words + TILE_ROWS - 1) / TILE_ROWS;
    unsigned long long tile = static_cast<unsigned long long>(blockIdx.x) * (blockDim.x / WAVE) + wave;
    uint32_t* slots = lds + wave * 4 * TILE_RECORD_PIECES;
    u32x4 a = {0, 0, 0, 0};
    u32x4 b = {0, 0, 0, 0};
    if (MODE && tile < tiles) {
        loadTile<MODE == 2>(p, tile, lane, a, b, p.records, p.ids);
    }
    for (; tile < tiles; tile += stride) {
        if (MODE) {
            stageTile(slots, lane, a, b);
            if (tile + stride < tiles) {
                loadTile<MODE == 2>(p, tile + stride, lane, a, b, p.records, p.ids);   // in flight during the stores
            }
        }
        storeTile<MODE != 0>(p.out, p.words, tile, lane, slots);
        __builtin_amdgcn_wave_barrier();
    }
}

// Experiments (tools/perf/r4/store_patterns.py; not part of bench.py's table). What makes the linear fill fast?
//   EXPERIMENT 0: every wavefront stores ONE KiB and exits -- after `delay` x ~0.43 us of sleeping (a wavefront that lives
//                 as long as a decoding one, but still stores once)
//   EXPERIMENT 1: every wavefront stores `stores` consecutive KiB (its own run) and exits, `delay` sleeps before the first
//   EXPERIMENT 2: as 1, `delay` sleeps between consecutive stores
__global__ void store_experiment(float4* out, unsigned long long pieces, int experiment, int stores, int delay)
{
    const uint32_t lane = threadIdx.x & (WAVE - 1);
    const unsigned long long wave = (static_cast<unsigned long long>(blockIdx.x) * blockDim.x + threadIdx.x) / WAVE;
    const int perWave = experiment == 0 ? 1 : stores;
    if (experiment != 2) {
        for (int i = 0; i < delay; ++i) {
            __builtin_amdgcn_s_sleep(16);   // 16 x 64 cycles
        }
    }
    for (int k = 0; k < perWave; ++k) {
        const unsigned long long i = (wave * perWave + k) * WAVE + lane;
        if (i < pieces) {
            out[i] = make_float4(1.f, 2.f, 3.f, 4.f);
        }
        if (experiment == 2) {
            for (int d = 0; d < delay; ++d) {
                __builtin_amdgcn_s_sleep(16);
            }
        }
    }
}

// Uniform-storage shapes (tools/perf/r4/store_patterns.py): a row's record is 20 pieces (320 B: {min, max} + 300 weights).
//   ROWS = 1: one row per wavefront -- 20 lanes load the record, the row leaves as TWO stores (1 KiB + 176 B), exit
//   ROWS = 8: eight rows per wavefront -- 160 pieces in three rounds, 9600 B as ten stores, exit
template <int ROWS>
__global__ void uniform_rows_per_wave(Params p)
{
    __shared__ __attribute__((aligned(16))) uint32_t lds[4 * 4 * 20 * 8];
    const uint32_t lane = threadIdx.x & (WAVE - 1);
    const uint32_t wave = threadIdx.x / WAVE;
    const unsigned long long first = (static_cast<unsigned long long>(blockIdx.x) * (blockDim.x / WAVE) + wave) * ROWS;
    if (first >= p.words) {
        return;
    }
    uint32_t* slots = lds + wave * 4 * 20 * 8;
    constexpr uint32_t RECORD = 20;
    for (uint32_t q = lane; q < ROWS * RECORD; q += WAVE) {
        const uint32_t w = q / RECORD;
        const unsigned long long word = first + w;
        unsigned long long row = word < p.words ? (p.ids ? p.ids[word] : word) : 0xFFFFFFFFull;
        u32x4 v = {0, 0, 0, 0};
        if (row < p.rows) {
            v = p.records[row * RECORD + (q - w * RECORD)];
        }
        *reinterpret_cast<u32x4*>(slots + 4 * q) = v;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const unsigned long long endPiece = p.words * (ROW_FLOATS / 4);
    float4* out = reinterpret_cast<float4*>(p.out) + first * (ROW_FLOATS / 4);
    for (uint32_t q = lane; q < ROWS * (ROW_FLOATS / 4); q += WAVE) {
        const uint32_t w = q / (ROW_FLOATS / 4);
        const uint32_t key = slots[4 * RECORD * w + 4 + (q - w * (ROW_FLOATS / 4))];   // the piece's four weights
        float4 value = make_float4(1.f, 2.f, 3.f, 4.f);
        value.x = __uint_as_float(key & 0x3f800000u);
        if (first * (ROW_FLOATS / 4) + q < endPiece) {
            out[q] = value;
        }
    }
}

// A parametrised tile pattern (tools/perf/r4/tile_patterns.py): `rowsPerTile` rows per wavefront (records of consecutive or
// random rows read first, through LDS), `delay` x ~0.43 us of sleeping between the loads and the stores (a decode's
// latency), and the order of the stores: 0 = the wavefront's own tile, ascending; 1 = descending; 2 = after a block
// barrier the block's wavefronts sweep the block's region together (wavefront w stores KiB w, w + W, ...).
__global__ void tile_experiment(Params p, int rowsPerTile, int delay, int order)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t dynamicLds[];
    const uint32_t lane = threadIdx.x & (WAVE - 1);
    const uint32_t wave = threadIdx.x / WAVE;
    const uint32_t wavesPerBlock = blockDim.x / WAVE;
    const unsigned long long tile = static_cast<unsigned long long>(blockIdx.x) * wavesPerBlock + wave;
    const uint32_t recordPieces = static_cast<uint32_t>(rowsPerTile) * RECORD_PIECES;
    uint32_t* slots = dynamicLds + wave * recordPieces * 4;
    const bool active = tile * rowsPerTile < p.words;
    if (active) {
        for (uint32_t q = lane; q < recordPieces; q += WAVE) {
            const uint32_t w = q / RECORD_PIECES;
            const unsigned long long word = tile * rowsPerTile + w;
            const unsigned long long row = word < p.words ? (p.ids ? p.ids[word] : word) : 0xFFFFFFFFull;
            u32x4 v = {0, 0, 0, 0};
            if (row < p.rows) {
                v = p.records[row * RECORD_PIECES + (q - w * RECORD_PIECES)];
            }
            *reinterpret_cast<u32x4*>(slots + 4 * q) = v;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int i = 0; i < delay; ++i) {
        __builtin_amdgcn_s_sleep(16);
    }
    const unsigned long long endPiece = p.words * (ROW_FLOATS / 4);
    const uint32_t tilePieces = static_cast<uint32_t>(rowsPerTile) * (ROW_FLOATS / 4);
    if (order == 2) {
        __syncthreads();
        const unsigned long long blockFirst = static_cast<unsigned long long>(blockIdx.x) * wavesPerBlock * tilePieces;
        const uint32_t blockPieces = wavesPerBlock * tilePieces;
        float4* out = reinterpret_cast<float4*>(p.out) + blockFirst;
        for (uint32_t q = wave * WAVE + lane; q < blockPieces; q += wavesPerBlock * WAVE) {
            float4 value = make_float4(1.f, 2.f, 3.f, 4.f);
            value.x = __uint_as_float(dynamicLds[q % (wavesPerBlock * recordPieces * 4)] & 0x3f800000u);
            if (blockFirst + q < endPiece) {
                out[q] = value;
            }
        }
        return;
    }
    if (!active) {
        return;
    }
    const unsigned long long firstPiece = tile * tilePieces;
    float4* out = reinterpret_cast<float4*>(p.out) + firstPiece;
    const uint32_t rounds = (tilePieces + WAVE - 1) / WAVE;
    for (uint32_t k = 0; k < rounds; ++k) {
        const uint32_t q = (order == 1 ? rounds - 1 - k : k) * WAVE + lane;
        float4 value = make_float4(1.f, 2.f, 3.f, 4.f);
        value.x = __uint_as_float(slots[q % (recordPieces * 4)] & 0x3f800000u);
        if (q < tilePieces && firstPiece + q < endPiece) {
            out[q] = value;
        }
    }
}

// Round 5, batch 28: T tiles per wavefront WITHOUT a persistent grid -- a block of W wavefronts owns W x T consecutive tiles, in
// step t wavefront w takes tile t x W + w of them (the block's wavefronts stay next to each other, as in decode_trained with
// tiles_per_wave = T); prefetch = 1: the records of step t + 1 are loaded (two registers per lane) before step t's tile is stored.
// The question: would a depth-one pipeline inside short-lived blocks -- what decode_union_split has with T = 2 -- beat one tile
// per wavefront for a single model?
template <int MODE>   // 1 = sequential records, 2 = random records
__global__ void tiles_chunked(Params p, int steps, int prefetch, int fronts)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t dynamicLds[];
    const uint32_t lane = threadIdx.x & (WAVE - 1);
    const uint32_t wave = threadIdx.x / WAVE;
    const uint32_t wavesPerBlock = blockDim.x / WAVE;
    const unsigned long long tiles = (p.words + TILE_ROWS - 1) / TILE_ROWS;
    // fronts F > 1: the grid writes F regions of the output at once -- groups of eight consecutive blocks (one per XCD) take
    // turns between the F equal parts of the batch (is it the SECOND write front that the strided pattern gains from?)
    unsigned long long chunk = blockIdx.x;
    if (fronts > 1) {
        const unsigned long long group = blockIdx.x / 8;
        const unsigned long long perFront = (static_cast<unsigned long long>(gridDim.x) / 8 + fronts - 1) / fronts;   // groups per front
        chunk = ((group % fronts) * perFront + group / fronts) * 8 + blockIdx.x % 8;
    }
    const unsigned long long first = chunk * wavesPerBlock * steps + wave;
    uint32_t* slots = dynamicLds + wave * 4 * TILE_RECORD_PIECES;
    u32x4 a = {0, 0, 0, 0};
    u32x4 b = {0, 0, 0, 0};
    if (first < tiles) {
        loadTile<MODE == 2>(p, first, lane, a, b, p.records, p.ids);
    }
    for (int t = 0; t < steps; ++t) {
        const unsigned long long tile = first + static_cast<unsigned long long>(t) * wavesPerBlock;
        if (tile >= tiles) {
            break;
        }
        stageTile(slots, lane, a, b);
        const unsigned long long next = tile + wavesPerBlock;
        if (prefetch && t + 1 < steps && next < tiles) {
            loadTile<MODE == 2>(p, next, lane, a, b, p.records, p.ids);   // in flight during the stores
        }
        storeTile<true>(p.out, p.words, tile, lane, slots);
        __builtin_amdgcn_wave_barrier();
        if (!prefetch && t + 1 < steps && next < tiles) {
            loadTile<MODE == 2>(p, next, lane, a, b, p.records, p.ids);
        }
    }
}

// Round 5, batch 29: what separates the records pipeline's memory skeleton from the two-tile pattern? tiles_persistent<1 / 2> again,
// T tiles per wavefront a grid apart, plus -- one at a time -- what the kernel has and the pattern has not:
//   copyBytes  a block-wide copy of that many bytes from global memory into LDS and a block barrier in front (tables + codebook: 6 KiB)
//   padBytes   unused LDS per block (fewer resident wavefronts: 15.5 KiB per block of four in the kernel; registers cap it at 24 per CU)
//   viaIds     row numbers come from an array even for consecutive rows (a dependent load in front of every tile's records)
template <int MODE>
__global__ void tiles_skeleton(Params p, const u32x4* tables, uint32_t copyPieces, uint32_t slotOffsetDwords, int viaIds)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t dynamicLds[];
    const uint32_t lane = threadIdx.x & (WAVE - 1);
    const uint32_t wave = threadIdx.x / WAVE;
    if (copyPieces) {
        for (uint32_t q = threadIdx.x; q < copyPieces; q += blockDim.x) {
            *reinterpret_cast<u32x4*>(dynamicLds + 4 * q) = tables[q];
        }
        __syncthreads();
    }
    const unsigned long long stride = static_cast<unsigned long long>(gridDim.x) * (blockDim.x / WAVE);
    const unsigned long long tiles = (p.words + TILE_ROWS - 1) / TILE_ROWS;
    unsigned long long tile = static_cast<unsigned long long>(blockIdx.x) * (blockDim.x / WAVE) + wave;
    uint32_t* slots = dynamicLds + slotOffsetDwords + wave * 4 * TILE_RECORD_PIECES;
    u32x4 a = {0, 0, 0, 0};
    u32x4 b = {0, 0, 0, 0};
    if (tile < tiles) {
        if (MODE == 2 || viaIds) {
            loadTile<true>(p, tile, lane, a, b, p.records, p.ids);
        } else {
            loadTile<false>(p, tile, lane, a, b, p.records, p.ids);
        }
    }
    for (; tile < tiles; tile += stride) {
        stageTile(slots, lane, a, b);
        if (tile + stride < tiles) {
            if (MODE == 2 || viaIds) {
                loadTile<true>(p, tile + stride, lane, a, b, p.records, p.ids);
            } else {
                loadTile<false>(p, tile + stride, lane, a, b, p.records, p.ids);
            }
        }
        storeTile<true>(p.out, p.words, tile, lane, slots);
        __builtin_amdgcn_wave_barrier();
    }
}

// Round 5: wavefronts that STORE are not the ones that LOAD. A persistent block of W wavefronts, the
// first `loaders` of which do nothing but fetch row records into an LDS double buffer while the others do nothing but
// drain finished tiles with stores (the write-only persistent pattern, which is the fastest tile pattern some boxes
// have), one block barrier per round: in round g the storers write the S = W - loaders tiles of round g out of buffer
// g % 2 and the loaders fill buffer (g + 1) % 2 with the records of round g + 1's tiles, S / loaders tiles each, all of a
// loader's loads in flight before the first is written to LDS. The stored values depend on the loaded bytes, as everywhere here.
template <int MODE>   // 1 = sequential records, 2 = random records
__global__ void tiles_specialised(Params p, uint32_t loaders)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t dynamicLds[];
    constexpr uint32_t MAX_TILES_PER_LOADER = 7;
    const uint32_t lane = threadIdx.x & (WAVE - 1);
    const uint32_t wave = threadIdx.x / WAVE;
    const uint32_t storers = blockDim.x / WAVE - loaders;
    const unsigned long long tiles = (p.words + TILE_ROWS - 1) / TILE_ROWS;
    constexpr uint32_t SLOT = 4 * TILE_RECORD_PIECES;   // dwords of one tile's records
    auto buffer = [&](unsigned long long round) { return dynamicLds + (round & 1) * storers * SLOT; };
    auto firstTile = [&](unsigned long long round) { return (static_cast<unsigned long long>(blockIdx.x) + round * gridDim.x) * storers; };
    auto loadRound = [&](unsigned long long round) {
        u32x4 a[MAX_TILES_PER_LOADER], b[MAX_TILES_PER_LOADER];
#pragma unroll
        for (uint32_t i = 0; i < MAX_TILES_PER_LOADER; ++i) {
            const uint32_t s = wave + i * loaders;
            a[i] = u32x4{0, 0, 0, 0};
            b[i] = u32x4{0, 0, 0, 0};
            if (s < storers && firstTile(round) + s < tiles) {
                loadTile<MODE == 2>(p, firstTile(round) + s, lane, a[i], b[i], p.records, p.ids);
            }
        }
#pragma unroll
        for (uint32_t i = 0; i < MAX_TILES_PER_LOADER; ++i) {
            const uint32_t s = wave + i * loaders;
            if (s < storers) {
                stageTile(buffer(round) + s * SLOT, lane, a[i], b[i]);
            }
        }
    };
    if (wave < loaders) {
        loadRound(0);
    }
    __syncthreads();
    for (unsigned long long round = 0; firstTile(round) < tiles; ++round) {
        if (wave < loaders) {
            if (firstTile(round + 1) < tiles) {
                loadRound(round + 1);
            }
        } else {
            const unsigned long long tile = firstTile(round) + (wave - loaders);
            if (tile < tiles) {
                storeTile<true>(p.out, p.words, tile, lane, buffer(round) + (wave - loaders) * SLOT);
            }
        }
        __syncthreads();
    }
}

// union: 4 words per tile, each 600 floats wide; records of the 4 words from both arrays (slots 0-3 / 4-7)
__global__ void union_tile_per_wave(Params p)
{
    __shared__ __attribute__((aligned(16))) uint32_t lds[4 * 4 * TILE_RECORD_PIECES];
    const uint32_t lane = threadIdx.x & (WAVE - 1);
    const uint32_t wave = threadIdx.x / WAVE;
    const unsigned long long tile = static_cast<unsigned long long>(blockIdx.x) * (blockDim.x / WAVE) + wave;
    constexpr uint32_t HALF = TILE_ROWS / 2;
    if (tile * HALF >= p.words) {
        return;
    }
    uint32_t* slots = lds + wave * 4 * TILE_RECORD_PIECES;
    u32x4 a = {0, 0, 0, 0};
    u32x4 b = {0, 0, 0, 0};
#pragma unroll
    for (int round = 0; round < 2; ++round) {
        const uint32_t q = round * WAVE + lane;
        if (q < TILE_RECORD_PIECES) {
            const uint32_t slot = q / RECORD_PIECES;
            const uint32_t piece = q - slot * RECORD_PIECES;
            const bool upper = slot >= HALF;
            const unsigned long long word = tile * HALF + (slot - (upper ? HALF : 0));
            const uint32_t* ids = upper ? p.ids2 : p.ids;
            const unsigned long long row = word < p.words ? ids[word] : 0xFFFFFFFFull;
            if (row < p.rows) {
                const u32x4 v = (upper ? p.records2 : p.records)[row * RECORD_PIECES + piece];
                if (round == 0) {
                    a = v;
                } else {
                    b = v;
                }
            }
        }
    }
    stageTile(slots, lane, a, b);
    // 4 merged rows of 600 floats = the same 600 pieces, consecutive
    const unsigned long long firstPiece = tile * TILE_PIECES;
    const unsigned long long endPiece = p.words * (2 * ROW_FLOATS / 4);
    float4* tileOut = reinterpret_cast<float4*>(p.out) + firstPiece;
    for (uint32_t q = lane; q < TILE_PIECES; q += WAVE) {
        float4 value = make_float4(1.f, 2.f, 3.f, 4.f);
        value.x = __uint_as_float(slots[q % (4 * TILE_RECORD_PIECES)] & 0x3f800000u);
        if (firstPiece + q < endPiece) {
            tileOut[q] = value;
        }
    }
}

}  // namespace

extern "C" {

// out: words x 300 floats (pattern 7: words x 600), 16-byte aligned. records: rows x 160 bytes. ids: words row numbers
// (device memory). Returns a hipError_t (0 = launched). Enqueues on `stream` and returns.
int memb_ceiling_launch(
    int pattern, float* out, unsigned long long words, const void* records, const void* records2, unsigned long long rows,
    const uint32_t* ids, const uint32_t* ids2, void* stream, int computeUnits)
{
    Params p{};
    p.out = out;
    p.words = words;
    p.records = static_cast<const u32x4*>(records);
    p.records2 = static_cast<const u32x4*>(records2);
    p.rows = rows;
    p.ids = ids;
    p.ids2 = ids2;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const unsigned long long tiles = (words + TILE_ROWS - 1) / TILE_ROWS;
    const uint32_t tileBlocks = static_cast<uint32_t>((tiles + 3) / 4);
    const uint32_t resident = static_cast<uint32_t>(computeUnits) * 4;   // 4 blocks of 4 wavefronts per CU = 16 wavefronts
    if ((pattern >= 2 && pattern != 4 && (!records || !rows)) || ((pattern == 3 || pattern == 6 || pattern == 7 || pattern == 11) && !ids) ||
        (pattern == 7 && (!records2 || !ids2)) || !out || words == 0) {
        return static_cast<int>(hipErrorInvalidValue);
    }
    switch (pattern) {
        case 0: {
            const unsigned long long pieces = words * (ROW_FLOATS / 4);
            hipLaunchKernelGGL(fill_linear, dim3(static_cast<uint32_t>((pieces + 255) / 256)), dim3(256), 0, s,
                               reinterpret_cast<float4*>(out), pieces);
            break;
        }
        case 1: hipLaunchKernelGGL(tile_per_wave<0>, dim3(tileBlocks), dim3(256), 0, s, p); break;
        case 2: hipLaunchKernelGGL(tile_per_wave<1>, dim3(tileBlocks), dim3(256), 0, s, p); break;
        case 3: hipLaunchKernelGGL(tile_per_wave<2>, dim3(tileBlocks), dim3(256), 0, s, p); break;
        case 4: hipLaunchKernelGGL(tiles_persistent<0>, dim3(tileBlocks < resident ? tileBlocks : resident), dim3(256), 0, s, p); break;
        case 5: hipLaunchKernelGGL(tiles_persistent<1>, dim3(tileBlocks < resident ? tileBlocks : resident), dim3(256), 0, s, p); break;
        case 6: hipLaunchKernelGGL(tiles_persistent<2>, dim3(tileBlocks < resident ? tileBlocks : resident), dim3(256), 0, s, p); break;
        case 8:   // (records: rows x 320 bytes here; ids may be null = consecutive rows)
            hipLaunchKernelGGL(uniform_rows_per_wave<1>, dim3(static_cast<uint32_t>((words + 3) / 4)), dim3(256), 0, s, p);
            break;
        case 10:   // two tiles per wavefront, a grid apart, the second one's records in flight during the first one's stores; then exit
            hipLaunchKernelGGL(tiles_persistent<1>, dim3((tileBlocks + 1) / 2), dim3(256), 0, s, p);
            break;
        case 11:
            hipLaunchKernelGGL(tiles_persistent<2>, dim3((tileBlocks + 1) / 2), dim3(256), 0, s, p);
            break;
        case 9:
            hipLaunchKernelGGL(uniform_rows_per_wave<8>, dim3(static_cast<uint32_t>(((words + 7) / 8 + 3) / 4)), dim3(256), 0, s, p);
            break;
        case 7: {
            const unsigned long long unionTiles = (words + TILE_ROWS / 2 - 1) / (TILE_ROWS / 2);
            hipLaunchKernelGGL(union_tile_per_wave, dim3(static_cast<uint32_t>((unionTiles + 3) / 4)), dim3(256), 0, s, p);
            break;
        }
        default:
            return static_cast<int>(hipErrorInvalidValue);
    }
    return static_cast<int>(hipGetLastError());
}

// tools/perf/r4/tile_patterns.py: see tile_experiment. ids may be null (consecutive rows); wavesPerBlock 1 .. 16.
int memb_ceiling_tile_experiment(
    float* out, unsigned long long words, const void* records, unsigned long long rows, const uint32_t* ids, int rowsPerTile,
    int wavesPerBlock, int delay, int order, void* stream)
{
    if (!out || !records || words == 0 || rowsPerTile < 1 || rowsPerTile > 64 || wavesPerBlock < 1 || wavesPerBlock > 16) {
        return static_cast<int>(hipErrorInvalidValue);
    }
    Params p{};
    p.out = out;
    p.words = words;
    p.records = static_cast<const u32x4*>(records);
    p.rows = rows;
    p.ids = ids;
    const unsigned long long tiles = (words + rowsPerTile - 1) / rowsPerTile;
    const uint32_t blocks = static_cast<uint32_t>((tiles + wavesPerBlock - 1) / wavesPerBlock);
    const uint32_t ldsBytes = static_cast<uint32_t>(wavesPerBlock) * rowsPerTile * RECORD_PIECES * 16;
    static bool raised = false;
    if (!raised) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tile_experiment), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        raised = true;
    }
    hipLaunchKernelGGL(tile_experiment, dim3(blocks), dim3(wavesPerBlock * WAVE), ldsBytes, static_cast<hipStream_t>(stream), p, rowsPerTile, delay, order);
    return static_cast<int>(hipGetLastError());
}

// tools/perf/r5/specialised.py: see tiles_specialised. wavesPerBlock 2 .. 16, 1 <= loaders < wavesPerBlock with at most seven
// tiles per loader, wavesPerCu resident wavefronts per CU (the grid); ids null = consecutive rows.
int memb_ceiling_specialised(
    float* out, unsigned long long words, const void* records, unsigned long long rows, const uint32_t* ids, int wavesPerBlock,
    int loaders, int wavesPerCu, void* stream, int computeUnits)
{
    if (!out || !records || words == 0 || wavesPerBlock < 2 || wavesPerBlock > 16 || loaders < 1 || loaders >= wavesPerBlock ||
        (wavesPerBlock - loaders + loaders - 1) / loaders > 7 || wavesPerCu < wavesPerBlock) {
        return static_cast<int>(hipErrorInvalidValue);
    }
    Params p{};
    p.out = out;
    p.words = words;
    p.records = static_cast<const u32x4*>(records);
    p.rows = rows;
    p.ids = ids;
    const uint32_t storers = static_cast<uint32_t>(wavesPerBlock - loaders);
    const unsigned long long tiles = (words + TILE_ROWS - 1) / TILE_ROWS;
    const unsigned long long rounds = (tiles + storers - 1) / storers;
    const unsigned long long resident = static_cast<unsigned long long>(computeUnits) * (wavesPerCu / wavesPerBlock);
    const uint32_t blocks = static_cast<uint32_t>(rounds < resident ? rounds : resident);
    const uint32_t ldsBytes = 2 * storers * 4 * TILE_RECORD_PIECES * 4;
    if (ids) {
        hipLaunchKernelGGL(tiles_specialised<2>, dim3(blocks), dim3(wavesPerBlock * WAVE), ldsBytes, static_cast<hipStream_t>(stream), p, static_cast<uint32_t>(loaders));
    } else {
        hipLaunchKernelGGL(tiles_specialised<1>, dim3(blocks), dim3(wavesPerBlock * WAVE), ldsBytes, static_cast<hipStream_t>(stream), p, static_cast<uint32_t>(loaders));
    }
    return static_cast<int>(hipGetLastError());
}

// tools/perf/r4/store_patterns.py: see store_experiment. threads = 64 .. 1024 per block.
int memb_ceiling_store_experiment(float* out, unsigned long long words, int experiment, int stores, int delay, int threads, void* stream)
{
    if (!out || words == 0 || stores < 1 || threads < 64 || threads > 1024 || threads % 64) {
        return static_cast<int>(hipErrorInvalidValue);
    }
    const unsigned long long pieces = words * (ROW_FLOATS / 4);
    const unsigned long long perWave = static_cast<unsigned long long>(experiment == 0 ? 1 : stores) * WAVE;
    const unsigned long long waves = (pieces + perWave - 1) / perWave;
    const unsigned long long blocks = (waves * WAVE + threads - 1) / threads;
    hipLaunchKernelGGL(store_experiment, dim3(static_cast<uint32_t>(blocks)), dim3(threads), 0, static_cast<hipStream_t>(stream),
                       reinterpret_cast<float4*>(out), pieces, experiment, stores, delay);
    return static_cast<int>(hipGetLastError());
}

// tiles_chunked: W wavefronts per block, `steps` tiles per wavefront, prefetch 0 / 1; ids = null: consecutive rows
int memb_ceiling_chunked(
    float* out, unsigned long long words, const void* records, unsigned long long rows, const uint32_t* ids, int wavesPerBlock,
    int steps, int prefetch, int fronts, void* stream)
{
    if (!out || !records || words == 0 || wavesPerBlock < 1 || wavesPerBlock > 16 || steps < 1 || steps > 64 || fronts < 1 || fronts > 64) {
        return static_cast<int>(hipErrorInvalidValue);
    }
    Params p{};
    p.out = out;
    p.words = words;
    p.records = static_cast<const u32x4*>(records);
    p.rows = rows;
    p.ids = ids;
    const unsigned long long tiles = (words + TILE_ROWS - 1) / TILE_ROWS;
    const unsigned long long perBlock = static_cast<unsigned long long>(wavesPerBlock) * steps;
    uint32_t blocks = static_cast<uint32_t>((tiles + perBlock - 1) / perBlock);
    if (fronts > 1) {
        blocks = (blocks + 8 * fronts - 1) / (8 * fronts) * (8 * fronts);   // (whole groups per front; surplus blocks find no tile)
    }
    const uint32_t ldsBytes = static_cast<uint32_t>(wavesPerBlock) * TILE_RECORD_PIECES * 16;
    if (ids) {
        hipLaunchKernelGGL(tiles_chunked<2>, dim3(blocks), dim3(wavesPerBlock * WAVE), ldsBytes, static_cast<hipStream_t>(stream), p, steps, prefetch, fronts);
    } else {
        hipLaunchKernelGGL(tiles_chunked<1>, dim3(blocks), dim3(wavesPerBlock * WAVE), ldsBytes, static_cast<hipStream_t>(stream), p, steps, prefetch, fronts);
    }
    return static_cast<int>(hipGetLastError());
}

// tiles_skeleton: T tiles per wavefront (grid = tile blocks / T), blocks of four; ids: row numbers (random rows, or consecutive ones
// with viaIds = 1); tables: copyBytes of device memory to copy into LDS per block (may be null when copyBytes = 0)
int memb_ceiling_skeleton(
    float* out, unsigned long long words, const void* records, unsigned long long rows, const uint32_t* ids, int randomRows, int viaIds,
    int tilesPerWave, const void* tables, unsigned int copyBytes, unsigned int padBytes, void* stream)
{
    if (!out || !records || words == 0 || tilesPerWave < 1 || copyBytes % 16 || padBytes % 16 || copyBytes + padBytes > 120 * 1024 ||
        ((randomRows || viaIds) && !ids) || (copyBytes && !tables)) {
        return static_cast<int>(hipErrorInvalidValue);
    }
    Params p{};
    p.out = out;
    p.words = words;
    p.records = static_cast<const u32x4*>(records);
    p.rows = rows;
    p.ids = ids;
    const unsigned long long tiles = (words + TILE_ROWS - 1) / TILE_ROWS;
    const uint32_t tileBlocks = static_cast<uint32_t>((tiles + 3) / 4);
    const uint32_t blocks = (tileBlocks + tilesPerWave - 1) / tilesPerWave;
    const uint32_t slotOffsetDwords = (copyBytes + padBytes) / 4;
    const uint32_t ldsBytes = copyBytes + padBytes + 4 * TILE_RECORD_PIECES * 16;
    static bool raised = false;
    if (!raised) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tiles_skeleton<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tiles_skeleton<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        raised = true;
    }
    if (randomRows) {
        hipLaunchKernelGGL(tiles_skeleton<2>, dim3(blocks), dim3(256), ldsBytes, static_cast<hipStream_t>(stream), p,
                           static_cast<const u32x4*>(tables), copyBytes / 16, slotOffsetDwords, viaIds);
    } else {
        hipLaunchKernelGGL(tiles_skeleton<1>, dim3(blocks), dim3(256), ldsBytes, static_cast<hipStream_t>(stream), p,
                           static_cast<const u32x4*>(tables), copyBytes / 16, slotOffsetDwords, viaIds);
    }
    return static_cast<int>(hipGetLastError());
}

}  // extern "C"

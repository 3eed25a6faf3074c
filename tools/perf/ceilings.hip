// libmemb_ceilings.so -- what this GPU does with the decoder's MEMORY pattern and nothing else.
//
// Measurement code, not product: bench.py loads it (ctypes) to put the box's own ceilings next to the
// kernel's time (`bench.py --extras`: tools/perf/bench_extras.py; tools/perf/r6/rot_ceilings.py at small sizes). The
// experiments of rounds 4-5 that lived here -- stores per wavefront, 54 tile shapes, chunked and specialised wavefronts, the
// skeleton of the records pipeline -- are in profiles/r04_experiments.txt / r05_experiments.txt with their numbers.
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
//   12, 13, 14  expand_keys: the output half of a two-kernel decoder (one / two / four 16-byte pieces per thread behind a load of nibble keys)
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

// Uniform-storage shapes (round 4; patterns 8 and 9): a row's record is 20 pieces (320 B: {min, max} + 300 weights).
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

// Round 6: the OUTPUT HALF of a two-kernel decoder as a pattern: symbols already decoded lie in memory as nibble keys (one uint16 =
// four weights = one 16-byte piece of output), a thread reads its uint16 (128 consecutive bytes per wavefront), picks four of sixteen
// centroids held in the lanes of one register (ds_bpermute: no LDS memory, no block barrier), stores ONE float4 and the wavefront
// exits -- the linear fill's store pattern behind a dependent load. PIECES > 1: that many pieces per thread, 64 pieces apart.
template <int PIECES>
__global__ void expand_keys(float4* out, unsigned long long pieces, const uint16_t* keys)
{
    const uint32_t lane = threadIdx.x & (WAVE - 1);
    const float centroid = lane < 16 ? 0.125f * static_cast<float>(lane) - 1.f : 0.f;
    const unsigned long long first = (static_cast<unsigned long long>(blockIdx.x) * blockDim.x + threadIdx.x - lane) * PIECES + lane;
    uint32_t key[PIECES];
#pragma unroll
    for (int u = 0; u < PIECES; ++u) {
        const unsigned long long i = first + static_cast<unsigned long long>(u) * WAVE;
        key[u] = i < pieces ? keys[i] : 0u;
    }
#pragma unroll
    for (int u = 0; u < PIECES; ++u) {
        const unsigned long long i = first + static_cast<unsigned long long>(u) * WAVE;
        float4 value;
        value.x = __shfl(centroid, key[u] & 15);
        value.y = __shfl(centroid, (key[u] >> 4) & 15);
        value.z = __shfl(centroid, (key[u] >> 8) & 15);
        value.w = __shfl(centroid, (key[u] >> 12) & 15);
        if (i < pieces) {
            out[i] = value;
        }
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
        case 12:   // (records: at least words x 150 bytes of anything: the keys)
        case 13:
        case 14: {
            const unsigned long long pieces = words * (ROW_FLOATS / 4);
            const int per = pattern == 12 ? 1 : pattern == 13 ? 2 : 4;
            const uint32_t blocks = static_cast<uint32_t>((pieces + 256ull * per - 1) / (256ull * per));
            if (pattern == 12) {
                hipLaunchKernelGGL(expand_keys<1>, dim3(blocks), dim3(256), 0, s, reinterpret_cast<float4*>(out), pieces, reinterpret_cast<const uint16_t*>(records));
            } else if (pattern == 13) {
                hipLaunchKernelGGL(expand_keys<2>, dim3(blocks), dim3(256), 0, s, reinterpret_cast<float4*>(out), pieces, reinterpret_cast<const uint16_t*>(records));
            } else {
                hipLaunchKernelGGL(expand_keys<4>, dim3(blocks), dim3(256), 0, s, reinterpret_cast<float4*>(out), pieces, reinterpret_cast<const uint16_t*>(records));
            }
            break;
        }
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

}  // extern "C"

// Write-pattern microbenchmark: how fast can 2.635 GB be written under the tile patterns the decoder uses?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// classic: thread i writes float4 i, grid-stride
__global__ void fill_linear(float4* out, size_t n4) {
    size_t stride = size_t(gridDim.x) * blockDim.x;
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n4; i += stride) out[i] = make_float4(1, 2, 3, 4);
}
// persistent waves, tile of `tilePieces` float4 per wave-iteration, tiles strided by total waves
__global__ void fill_tiles(float4* out, size_t n4, unsigned tilePieces, int gapSleeps) {
    unsigned lane = threadIdx.x & 63;
    size_t wave = (size_t(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
    size_t waves = (size_t(gridDim.x) * blockDim.x) >> 6;
    size_t tiles = (n4 + tilePieces - 1) / tilePieces;
    for (size_t t = wave; t < tiles; t += waves) {
        size_t base = t * tilePieces;
        for (unsigned q = lane; q < tilePieces; q += 64) {
            if (base + q < n4) out[base + q] = make_float4(1, 2, 3, 4);
            for (int s = 0; s < gapSleeps; ++s) __builtin_amdgcn_s_sleep(4);   // ~64 cycles each
        }
    }
}
// persistent blocks: a block's waves write one contiguous region of waves*tilePieces together
__global__ void fill_block_tiles(float4* out, size_t n4, unsigned tilePieces) {
    unsigned wavesPerBlock = blockDim.x >> 6;
    size_t blockPieces = size_t(tilePieces) * wavesPerBlock;
    size_t regions = (n4 + blockPieces - 1) / blockPieces;
    for (size_t r = blockIdx.x; r < regions; r += gridDim.x) {
        size_t base = r * blockPieces;
        for (size_t q = threadIdx.x; q < blockPieces; q += blockDim.x)
            if (base + q < n4) out[base + q] = make_float4(1, 2, 3, 4);
    }
}
// tiles + a gather: per word `piecesPerWord` 16-byte loads starting at a (pseudo)random or sequential offset of `src`
__global__ void fill_tiles_gather(float4* out, size_t n4, unsigned tileWords, const uint4* src, size_t srcPieces,
                                  unsigned piecesPerWord, unsigned strideBytes, int randomOrder, unsigned* sink) {
    unsigned lane = threadIdx.x & 63;
    size_t wave = (size_t(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
    size_t waves = (size_t(gridDim.x) * blockDim.x) >> 6;
    unsigned tilePieces = tileWords * 75;
    size_t tiles = (n4 + tilePieces - 1) / tilePieces;
    unsigned acc = 0;
    for (size_t t = wave; t < tiles; t += waves) {
        // gather for this tile: tileWords * piecesPerWord pieces, lane -> (word, piece)
        unsigned total = tileWords * piecesPerWord;
        for (unsigned q = lane; q < total; q += 64) {
            unsigned w = q / piecesPerWord, piece = q - w * piecesPerWord;
            size_t word = t * tileWords + w;
            size_t start = randomOrder ? (word * 2654435761ull) % (srcPieces - piecesPerWord - 16) : (word * strideBytes) / 16;
            uint4 v = src[start + piece];
            acc += v.x ^ v.y ^ v.z ^ v.w;
        }
        size_t base = t * tilePieces;
        for (unsigned q = lane; q < tilePieces; q += 64)
            if (base + q < n4) out[base + q] = make_float4(1, 2, 3, 4);
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
// tiles + ONE contiguous read per tile: the wave reads `tileReadPieces` consecutive 16-byte pieces (coalesced)
__global__ void fill_tiles_block_read(float4* out, size_t n4, unsigned tileWords, const uint4* src, size_t srcPieces,
                                      unsigned tileReadPieces, unsigned* sink) {
    unsigned lane = threadIdx.x & 63;
    size_t wave = (size_t(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
    size_t waves = (size_t(gridDim.x) * blockDim.x) >> 6;
    unsigned tilePieces = tileWords * 75;
    size_t tiles = (n4 + tilePieces - 1) / tilePieces;
    unsigned acc = 0;
    for (size_t t = wave; t < tiles; t += waves) {
        size_t start = (t * tileReadPieces) % (srcPieces - tileReadPieces - 64);
        for (unsigned q = lane; q < tileReadPieces; q += 64) {
            uint4 v = src[start + q];
            acc += v.x ^ v.y ^ v.z ^ v.w;
        }
        size_t base = t * tilePieces;
        for (unsigned q = lane; q < tilePieces; q += 64)
            if (base + q < n4) out[base + q] = make_float4(1, 2, 3, 4);
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
// read-only touch of a range (pull it into L2 / Infinity Cache)
__global__ void touch_range(const uint4* src, size_t first, size_t count, unsigned* sink) {
    unsigned acc = 0;
    size_t stride = size_t(gridDim.x) * blockDim.x;
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < count; i += stride) {
        uint4 v = src[first + i];
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
// tiles [tileFirst, tileFirst + tileCount) with one contiguous read per tile at src[(tile * tileReadPieces)]
__global__ void fill_tiles_range(float4* out, size_t n4, unsigned tileWords, const uint4* src, unsigned tileReadPieces,
                                 size_t tileFirst, size_t tileCount, unsigned* sink) {
    unsigned lane = threadIdx.x & 63;
    size_t wave = (size_t(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
    size_t waves = (size_t(gridDim.x) * blockDim.x) >> 6;
    unsigned tilePieces = tileWords * 75;
    unsigned acc = 0;
    for (size_t k = wave; k < tileCount; k += waves) {
        size_t t = tileFirst + k;
        size_t start = t * tileReadPieces;
        for (unsigned q = lane; q < tileReadPieces; q += 64) {
            uint4 v = src[start + q];
            acc += v.x ^ v.y ^ v.z ^ v.w;
        }
        size_t base = t * tilePieces;
        for (unsigned q = lane; q < tilePieces; q += 64)
            if (base + q < n4) out[base + q] = make_float4(1, 2, 3, 4);
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
// persistent tiles + contiguous read per tile, plus a "touch ahead": every `ahead` tiles a wave touches, with ONE load
// instruction per 64 lines, every 128-B line its next `ahead` tiles will read (bringing them into L2 / Infinity Cache)
__global__ void fill_tiles_touch_ahead(float4* out, size_t n4, unsigned tileWords, const uint4* src, unsigned tileReadPieces,
                                       unsigned ahead, unsigned* sink) {
    unsigned lane = threadIdx.x & 63;
    size_t wave = (size_t(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
    size_t waves = (size_t(gridDim.x) * blockDim.x) >> 6;
    unsigned tilePieces = tileWords * 75;
    size_t tiles = (n4 + tilePieces - 1) / tilePieces;
    unsigned linesPerTile = (tileReadPieces * 16 + 127) / 128 + 1;
    unsigned acc = 0;
    size_t iteration = 0;
    for (size_t t = wave; t < tiles; t += waves, ++iteration) {
        if (ahead && iteration % ahead == 0) {
            // lines of tiles t + ahead*waves .. t + (2*ahead-1)*waves
            for (unsigned l = lane; l < ahead * linesPerTile; l += 64) {
                size_t tt = t + size_t(ahead + l / linesPerTile) * waves;
                if (tt < tiles) {
                    const unsigned* line = reinterpret_cast<const unsigned*>(src + tt * tileReadPieces) + 32 * (l % linesPerTile);
                    acc += *line;
                }
            }
        }
        size_t start = t * tileReadPieces;
        for (unsigned q = lane; q < tileReadPieces; q += 64) {
            uint4 v = src[start + q];
            acc += v.x ^ v.y ^ v.z ^ v.w;
        }
        size_t base = t * tilePieces;
        for (unsigned q = lane; q < tilePieces; q += 64)
            if (base + q < n4) out[base + q] = make_float4(1, 2, 3, 4);
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
// non-persistent: every wave writes `storesPerWave` consecutive KiB and exits
__global__ void fill_short_waves(float4* out, size_t n4, unsigned storesPerWave) {
    unsigned lane = threadIdx.x & 63;
    size_t wave = (size_t(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
    size_t base = wave * storesPerWave * 64;
    for (unsigned s = 0; s < storesPerWave; ++s) {
        size_t i = base + s * 64 + lane;
        if (i < n4) out[i] = make_float4(1, 2, 3, 4);
    }
}

// persistent tiles with at most N stores outstanding per wave (s_waitcnt vmcnt(N) after every store)
template <int N>
__global__ void fill_tiles_throttled(float4* out, size_t n4, unsigned tilePieces) {
    unsigned lane = threadIdx.x & 63;
    size_t wave = (size_t(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
    size_t waves = (size_t(gridDim.x) * blockDim.x) >> 6;
    size_t tiles = (n4 + tilePieces - 1) / tilePieces;
    for (size_t t = wave; t < tiles; t += waves) {
        size_t base = t * tilePieces;
        for (unsigned q = lane; q < tilePieces; q += 64) {
            if (base + q < n4) out[base + q] = make_float4(1, 2, 3, 4);
            if (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
            if (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            if (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            if (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        }
    }
}
template <typename F> float timeIt(F f) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) f();
    std::vector<float> ms;
    for (int i = 0; i < 10; ++i) { hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b); float t; hipEventElapsedTime(&t, a, b); ms.push_back(t); }
    std::sort(ms.begin(), ms.end()); return ms[ms.size() / 2];
}
int main() {
    const size_t words = 2196017, n4 = words * 75;
    float4* out; CHECK(hipMalloc(&out, n4 * 16));
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount;
    double gb = n4 * 16 / 1e9;
    auto show = [&](const char* name, float ms) { printf("%-64s %.3f ms  %.2f TB/s\n", name, ms, gb / ms); fflush(stdout); };
    show("linear grid-stride, 256 thr, 8 blocks/CU", timeIt([&] { hipLaunchKernelGGL(fill_linear, dim3(cus * 8), dim3(256), 0, 0, out, n4); }));
    show("linear, one float4 per thread (n4/256 blocks)", timeIt([&] { hipLaunchKernelGGL(fill_linear, dim3((n4 + 255) / 256), dim3(256), 0, 0, out, n4); }));
    for (int wavesPerCu : {8, 16, 32, 64}) {
        char name[128];
        #define THR(N) snprintf(name, sizeof name, "THROTTLED tiles 9600 B, <= %d stores in flight per wave, %2d waves/CU", N, wavesPerCu); \
            show(name, timeIt([&] { hipLaunchKernelGGL(fill_tiles_throttled<N>, dim3(cus * wavesPerCu / 8), dim3(512), 0, 0, out, n4, 600u); }));
        THR(0) THR(1) THR(2) THR(4) THR(8) THR(99)
    }
    if (getenv("WP_ONLY_THROTTLE")) return 0;
    for (unsigned words_per_tile : {8u, 16u, 64u}) for (int wavesPerCu : {8, 16, 32}) for (int gap : {0, 4}) {
        char name[128]; snprintf(name, sizeof name, "persistent waves: tile %2u words (%u B), %2d waves/CU, gap %d", words_per_tile, words_per_tile * 1200, wavesPerCu, gap);
        show(name, timeIt([&] { hipLaunchKernelGGL(fill_tiles, dim3(cus * wavesPerCu / 8), dim3(512), 0, 0, out, n4, words_per_tile * 75, gap); }));
    }
    for (unsigned words_per_tile : {8u, 16u}) {
        char name[128]; snprintf(name, sizeof name, "persistent blocks (8 waves write %u B together), 2 blocks/CU", words_per_tile * 1200 * 8);
        show(name, timeIt([&] { hipLaunchKernelGGL(fill_block_tiles, dim3(cus * 2), dim3(512), 0, 0, out, n4, words_per_tile * 75); }));
    }
    {
        size_t srcBytes = 286u << 20; uint4* src; CHECK(hipMalloc(&src, srcBytes)); CHECK(hipMemset(src, 1, srcBytes));
        unsigned* sink; CHECK(hipMalloc(&sink, 64));
        for (int randomOrder : {1, 0}) for (unsigned ppw : {11u, 9u, 6u}) for (int wavesPerCu : {16, 32}) {
            char name[160]; snprintf(name, sizeof name, "tiles 8 words + %s gather %u x16B/word from 286 MB, %d waves/CU", randomOrder ? "RANDOM" : "sequential", ppw, wavesPerCu);
            show(name, timeIt([&] { hipLaunchKernelGGL(fill_tiles_gather, dim3(cus * wavesPerCu / 8), dim3(512), 0, 0, out, n4, 8u, src, srcBytes / 16, ppw, 130u, randomOrder, sink); }));
        }
        for (size_t mb : {8u, 32u, 128u, 286u}) for (unsigned tileRead : {72u, 128u}) {
            char name[160]; snprintf(name, sizeof name, "tiles 8 words + one contiguous %u-B read per tile, source %zu MB, 32 waves/CU", tileRead * 16, mb);
            show(name, timeIt([&] { hipLaunchKernelGGL(fill_tiles_block_read, dim3(cus * 4), dim3(512), 0, 0, out, n4, 8u, src, (mb << 20) / 16, tileRead, sink); }));
        }
        for (size_t mb : {8u, 32u, 128u}) {
            char name[160]; snprintf(name, sizeof name, "tiles 8 words + RANDOM gather 9 x16B/word from %zu MB, 32 waves/CU", mb);
            show(name, timeIt([&] { hipLaunchKernelGGL(fill_tiles_gather, dim3(cus * 4), dim3(512), 0, 0, out, n4, 8u, src, (mb << 20) / 16, 9u, 130u, 1, sink); }));
        }
        {
            // chunked: per chunk a read-only touch kernel, then the tile kernel; source = 69 pieces (1104 B) per tile, 303 MB in all
            const unsigned tileRead = 69; const size_t tiles = (n4 + 599) / 600;
            uint4* big; CHECK(hipMalloc(&big, (tiles + 8) * tileRead * 16)); CHECK(hipMemset(big, 1, (tiles + 8) * tileRead * 16));
            for (unsigned ahead : {0u, 1u, 2u, 4u, 6u, 8u, 12u}) for (int wavesPerCu : {16, 32}) {
                char name[160]; snprintf(name, sizeof name, "TOUCH-AHEAD %2u tiles, %d waves/CU: tiles + contiguous 1104-B read per tile (303 MB source)", ahead, wavesPerCu);
                show(name, timeIt([&] { hipLaunchKernelGGL(fill_tiles_touch_ahead, dim3(cus * wavesPerCu / 8), dim3(512), 0, 0, out, n4, 8u, big, tileRead, ahead, sink); }));
            }
            for (int chunks : {1, 12}) for (int touch : {0, 1}) {
                char name[160]; snprintf(name, sizeof name, "CHUNKED x%2d %s: tiles + contiguous 1104-B read per tile (303 MB source)", chunks, touch ? "touch-then-decode" : "decode only      ");
                show(name, timeIt([&] {
                    size_t per = (tiles + chunks - 1) / chunks;
                    for (int c = 0; c < chunks; ++c) {
                        size_t first = size_t(c) * per, count = std::min(per, tiles - first);
                        if (touch) hipLaunchKernelGGL(touch_range, dim3(cus * 8), dim3(256), 0, 0, big, first * tileRead, count * tileRead, sink);
                        hipLaunchKernelGGL(fill_tiles_range, dim3(cus * 4), dim3(512), 0, 0, out, n4, 8u, big, tileRead, first, count, sink);
                    }
                }));
            }
        }
        size_t small = 165u << 20;
        show("tiles 8 words + RANDOM gather 7 x16B/word from 165 MB, 32 waves/CU", timeIt([&] { hipLaunchKernelGGL(fill_tiles_gather, dim3(cus * 4), dim3(512), 0, 0, out, n4, 8u, src, small / 16, 7u, 75u, 1, sink); }));
    }
    for (unsigned storesPerWave : {1u, 2u, 3u, 5u, 10u, 20u, 40u}) for (unsigned threads : {256u, 512u}) {
        char name[128]; snprintf(name, sizeof name, "short-lived waves: %2u x 1 KiB per wave, %u-thread blocks", storesPerWave, threads);
        size_t waves = (n4 + size_t(storesPerWave) * 64 - 1) / (size_t(storesPerWave) * 64);
        size_t blocks = (waves * 64 + threads - 1) / threads;
        show(name, timeIt([&] { hipLaunchKernelGGL(fill_short_waves, dim3((unsigned)blocks), dim3(threads), 0, 0, out, n4, storesPerWave); }));
    }
    // one tile per wave, non-persistent
    {
        unsigned tilePieces = 8 * 75; size_t tiles = (n4 + tilePieces - 1) / tilePieces;
        show("one 9600-B tile per wave, non-persistent (tiles/8 blocks)", timeIt([&] { hipLaunchKernelGGL(fill_tiles, dim3((tiles + 7) / 8), dim3(512), 0, 0, out, n4, tilePieces, 0); }));
    }
    return 0;
}

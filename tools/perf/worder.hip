// Write-order microbenchmark (round 2): what makes "one 1-KiB store per wave" (6.9 TB/s) faster than any
// pattern in which a wave stores three or more times (5.5 TB/s)? Every kernel writes the same 2.635 GB.
//
//   one store per wave, blocks of 256 threads = 4 KiB chunks, chunk order permuted:
//     identity / shuffled inside groups of 64 chunks / bit-reversed / multiplied by a large odd number
//   N stores per wave, the N chunks of a wave far apart (one per "pass" over the buffer)
//   N stores per wave issued by N *different* lanes groups at once (one instruction, 64 B... not possible) -- skipped
//   tiles of 9600 B, but the stores of a tile spread over the 8 waves of a block (block writes its 8 tiles
//     as one sequential sweep, wave w storing KiB 8k + w)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ unsigned reverseBits(unsigned v, unsigned bits) { return __brev(v) >> (32 - bits); }

// mode 0 identity, 1 shuffle inside groups of 64 chunks, 2 bit reversal over `bits`, 3 multiply by odd constant mod 2^bits
__global__ void fill_one_store(float4* out, size_t n4, unsigned chunks, unsigned bits, int mode) {
    unsigned b = blockIdx.x;
    unsigned chunk = b;
    if (mode == 1) chunk = (b & ~63u) | reverseBits(b & 63u, 6);
    if (mode == 2) chunk = reverseBits(b, bits);
    if (mode == 3) chunk = (b * 2654435761u) & ((1u << bits) - 1);
    if (chunk >= chunks) return;
    size_t i = size_t(chunk) * blockDim.x + threadIdx.x;
    if (i < n4) out[i] = make_float4(1, 2, 3, 4);
}

// one-shot waves, `steps` stores each: step s of block b goes to chunk s * gridDim.x + b (a pass over the buffer per step)
__global__ void fill_passes(float4* out, size_t n4, unsigned steps) {
    for (unsigned s = 0; s < steps; ++s) {
        size_t i = (size_t(s) * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x;
        if (i < n4) out[i] = make_float4(1, 2, 3, 4);
    }
}

// persistent blocks of 8 waves: per round a block owns 8 tiles of `tilePieces` float4 (contiguous); either every wave
// writes its own tile (sweep = 0) or the block writes the whole region as one sweep, thread t storing piece t, t + 512, ...
__global__ void fill_block_rounds(float4* out, size_t n4, unsigned tilePieces, int sweep, int barrier) {
    const unsigned wavesPerBlock = blockDim.x >> 6;
    const size_t regionPieces = size_t(tilePieces) * wavesPerBlock;
    const size_t regions = (n4 + regionPieces - 1) / regionPieces;
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (size_t r = blockIdx.x; r < regions; r += gridDim.x) {
        const size_t base = r * regionPieces;
        if (barrier) __syncthreads();
        if (sweep) {
            for (size_t q = threadIdx.x; q < regionPieces; q += blockDim.x)
                if (base + q < n4) out[base + q] = make_float4(1, 2, 3, 4);
        } else {
            for (unsigned q = lane; q < tilePieces; q += 64)
                if (base + size_t(wave) * tilePieces + q < n4) out[base + size_t(wave) * tilePieces + q] = make_float4(1, 2, 3, 4);
        }
    }
}

template <typename F>
float timeIt(F launch, int reps = 12) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 40; ++i) launch();   // run-in: the power state settles (tools/perf/ramp.py)
    std::vector<float> ms;
    for (int i = 0; i < reps; ++i) {
        hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b);
        float t; hipEventElapsedTime(&t, a, b); ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    return ms[ms.size() / 2];
}

int main() {
    const size_t words = 2196017, n4 = words * 75;   // 16-byte pieces of the output
    const double bytes = double(n4) * 16;
    float4* out;
    CHECK(hipMalloc(&out, n4 * 16 + 4096));
    const unsigned chunks = unsigned((n4 + 255) / 256);
    unsigned bits = 1;
    while ((1u << bits) < chunks) ++bits;
    const char* names[] = {"identity", "shuffled inside groups of 64 chunks (256 KiB)", "bit-reversed", "multiplied by an odd constant"};
    for (int mode = 0; mode < 4; ++mode) {
        float ms = timeIt([&] { hipLaunchKernelGGL(fill_one_store, dim3(1u << bits), dim3(256), 0, 0, out, n4, chunks, bits, mode); });
        printf("one 1-KiB store per wave, 4-KiB chunks %-50s %.3f ms  %.2f TB/s\n", names[mode], ms, bytes / ms / 1e9);
    }
    for (unsigned steps : {1u, 2u, 3u, 5u, 10u}) {
        unsigned blocks = unsigned((chunks + steps - 1) / steps);
        float ms = timeIt([&] { hipLaunchKernelGGL(fill_passes, dim3(blocks), dim3(256), 0, 0, out, n4, steps); });
        printf("%2u stores per wave, one per pass over the buffer                                      %.3f ms  %.2f TB/s\n", steps, ms, bytes / ms / 1e9);
    }
    for (int sweep = 0; sweep < 2; ++sweep) {
        for (int barrier = 0; barrier < 2; ++barrier) {
            for (unsigned blocksPerCu : {2u, 4u}) {
                float ms = timeIt([&] { hipLaunchKernelGGL(fill_block_rounds, dim3(256 * blocksPerCu), dim3(512), 0, 0, out, n4, 600u, sweep, barrier); });
                printf("persistent 8-wave blocks x %u per CU, 8 tiles of 9600 B per round, %-28s%s %.3f ms  %.2f TB/s\n", blocksPerCu,
                       sweep ? "block sweeps the region" : "each wave its own tile", barrier ? ", barrier per round" : "                  ", ms, bytes / ms / 1e9);
            }
        }
    }
    hipFree(out);
    return 0;
}

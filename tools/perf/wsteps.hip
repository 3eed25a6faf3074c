// Which property of a write stream costs bandwidth: stores per wave, block interleave, blocked vs strided assignment?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// E1 one-shot block writes steps * blockDim * 16 B, step-major (whole block writes consecutive 16*blockDim bytes per step)
__global__ void oneshot_block_interleaved(float4* out, size_t n4, unsigned steps) {
    size_t base = size_t(blockIdx.x) * steps * blockDim.x;
    for (unsigned s = 0; s < steps; ++s) {
        size_t i = base + size_t(s) * blockDim.x + threadIdx.x;
        if (i < n4) out[i] = make_float4(1, 2, 3, 4);
    }
}
// E1b one-shot, wave-major (each wave writes `steps` consecutive KiB)
__global__ void oneshot_wave_major(float4* out, size_t n4, unsigned steps) {
    unsigned lane = threadIdx.x & 63;
    size_t wave = (size_t(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
    size_t base = wave * steps * 64;
    for (unsigned s = 0; s < steps; ++s) {
        size_t i = base + s * 64 + lane;
        if (i < n4) out[i] = make_float4(1, 2, 3, 4);
    }
}
// E2 persistent, chunks handed out by one atomic counter per block-iteration (dispatch-like order)
__global__ void persistent_atomic(float4* out, size_t n4, unsigned chunkPieces, unsigned* counter) {
    __shared__ unsigned next;
    size_t chunks = (n4 + chunkPieces - 1) / chunkPieces;
    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) next = atomicAdd(counter, 1u);
        __syncthreads();
        size_t chunk = next;
        if (chunk >= chunks) break;
        size_t base = chunk * chunkPieces;
        for (unsigned q = threadIdx.x; q < chunkPieces; q += blockDim.x)
            if (base + q < n4) out[base + q] = make_float4(1, 2, 3, 4);
    }
}
// E4 persistent waves, BLOCKED assignment: wave w writes tiles [w*per, (w+1)*per) one after the other
__global__ void persistent_wave_blocked(float4* out, size_t n4, unsigned tilePieces) {
    unsigned lane = threadIdx.x & 63;
    size_t wave = (size_t(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
    size_t waves = (size_t(gridDim.x) * blockDim.x) >> 6;
    size_t tiles = (n4 + tilePieces - 1) / tilePieces;
    size_t per = (tiles + waves - 1) / waves;
    size_t first = wave * per, last = first + per < tiles ? first + per : tiles;
    for (size_t t = first; t < last; ++t) {
        size_t base = t * tilePieces;
        for (unsigned q = lane; q < tilePieces; q += 64)
            if (base + q < n4) out[base + q] = make_float4(1, 2, 3, 4);
    }
}
// E4b persistent blocks, BLOCKED assignment: block b writes one contiguous run, all threads together
__global__ void persistent_block_blocked(float4* out, size_t n4) {
    size_t per = (n4 + gridDim.x - 1) / gridDim.x;
    per = (per + blockDim.x - 1) / blockDim.x * blockDim.x;
    size_t first = size_t(blockIdx.x) * per, last = first + per < n4 ? first + per : n4;
    for (size_t i = first + threadIdx.x; i < last; i += blockDim.x) out[i] = make_float4(1, 2, 3, 4);
}
// E5 one store per thread, occupancy limited through dynamic LDS
__global__ void oneshot_lds(float4* out, size_t n4) {
    extern __shared__ float pad[];
    size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (threadIdx.x == 9999) pad[0] = 1;
    if (i < n4) out[i] = make_float4(1, 2, 3, 4);
}
// E6 persistent strided waves (the decoder's pattern) for reference
__global__ void persistent_wave_strided(float4* out, size_t n4, unsigned tilePieces) {
    unsigned lane = threadIdx.x & 63;
    size_t wave = (size_t(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
    size_t waves = (size_t(gridDim.x) * blockDim.x) >> 6;
    size_t tiles = (n4 + tilePieces - 1) / tilePieces;
    for (size_t t = wave; t < tiles; t += waves) {
        size_t base = t * tilePieces;
        for (unsigned q = lane; q < tilePieces; q += 64)
            if (base + q < n4) out[base + q] = make_float4(1, 2, 3, 4);
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
    float4* out; CHECK(hipMalloc(&out, n4 * 16 + (1 << 20)));
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount;
    double gb = n4 * 16 / 1e9;
    auto show = [&](const char* name, float ms) { printf("%-76s %.3f ms  %.2f TB/s\n", name, ms, gb / ms); fflush(stdout); };
    unsigned* counter; CHECK(hipMalloc(&counter, 64));
    char name[200];
    for (unsigned threads : {256u, 512u, 1024u}) for (unsigned steps : {1u, 2u, 3u, 5u, 10u}) {
        size_t per = size_t(threads) * steps; size_t blocks = (n4 + per - 1) / per;
        snprintf(name, sizeof name, "one-shot %4u thr, %2u steps, BLOCK-interleaved (%6zu B per block)", threads, steps, per * 16);
        show(name, timeIt([&] { hipLaunchKernelGGL(oneshot_block_interleaved, dim3((unsigned)blocks), dim3(threads), 0, 0, out, n4, steps); }));
        snprintf(name, sizeof name, "one-shot %4u thr, %2u steps, WAVE-major      (%6zu B per block)", threads, steps, per * 16);
        show(name, timeIt([&] { hipLaunchKernelGGL(oneshot_wave_major, dim3((unsigned)blocks), dim3(threads), 0, 0, out, n4, steps); }));
    }
    for (unsigned chunkBytes : {4096u, 16384u, 65536u}) for (unsigned threads : {256u, 1024u}) for (int perCu : {2, 8}) {
        if (threads == 1024 && perCu == 8) continue;
        snprintf(name, sizeof name, "persistent, chunks by atomic counter: %u thr x %d blocks/CU, %5u-B chunks", threads, perCu, chunkBytes);
        show(name, timeIt([&] { hipMemsetAsync(counter, 0, 4, 0); hipLaunchKernelGGL(persistent_atomic, dim3(cus * perCu), dim3(threads), 0, 0, out, n4, chunkBytes / 16, counter); }));
    }
    for (int wavesPerCu : {8, 16, 32}) {
        snprintf(name, sizeof name, "persistent waves STRIDED 9600-B tiles, %2d waves/CU", wavesPerCu);
        show(name, timeIt([&] { hipLaunchKernelGGL(persistent_wave_strided, dim3(cus * wavesPerCu / 8), dim3(512), 0, 0, out, n4, 600u); }));
        snprintf(name, sizeof name, "persistent waves BLOCKED 9600-B tiles (each wave one contiguous run), %2d waves/CU", wavesPerCu);
        show(name, timeIt([&] { hipLaunchKernelGGL(persistent_wave_blocked, dim3(cus * wavesPerCu / 8), dim3(512), 0, 0, out, n4, 600u); }));
    }
    for (unsigned threads : {256u, 512u, 1024u}) for (int perCu : {1, 2, 4, 8}) {
        if (threads * perCu > 2048) continue;
        snprintf(name, sizeof name, "persistent blocks BLOCKED (each block one contiguous run), %u thr x %d blocks/CU", threads, perCu);
        show(name, timeIt([&] { hipLaunchKernelGGL(persistent_block_blocked, dim3(cus * perCu), dim3(threads), 0, 0, out, n4); }));
    }
    for (unsigned lds : {0u, 16384u, 32768u, 65536u}) {
        snprintf(name, sizeof name, "one-shot 256 thr, one store per thread, %u B LDS per block (occupancy limit)", lds);
        show(name, timeIt([&] { hipLaunchKernelGGL(oneshot_lds, dim3((unsigned)((n4 + 255) / 256)), dim3(256), lds, 0, out, n4); }));
    }
    return 0;
}

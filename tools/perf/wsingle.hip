// One write pattern per process (for rocprofv3 --pmc):
//   wsingle steps <stores per thread> <threads per block>     one-shot fill
//   wsingle tiles <waves per CU>                               persistent 9600-B tiles
//   wsingle gather <waves per CU> <random 0|1>                 persistent tiles + 144-B gather per word from 290 MB
// Build: hipcc --offload-arch=gfx950 -O3 wsingle.hip -o wsingle
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
__global__ void fill_steps(float4* out, size_t n4, unsigned steps) {
    size_t base = size_t(blockIdx.x) * steps * blockDim.x;
    for (unsigned s = 0; s < steps; ++s) {
        size_t i = base + size_t(s) * blockDim.x + threadIdx.x;
        if (i < n4) out[i] = make_float4(1, 2, 3, 4);
    }
}
__global__ void fill_tiles(float4* out, size_t n4, unsigned tilePieces) {
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
__global__ void fill_tiles_gather(float4* out, size_t n4, const uint4* src, size_t srcPieces, int randomOrder, unsigned* sink) {
    unsigned lane = threadIdx.x & 63;
    size_t wave = (size_t(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
    size_t waves = (size_t(gridDim.x) * blockDim.x) >> 6;
    const unsigned tileWords = 8, piecesPerWord = 9, tilePieces = tileWords * 75;
    size_t tiles = (n4 + tilePieces - 1) / tilePieces;
    unsigned acc = 0;
    for (size_t t = wave; t < tiles; t += waves) {
        for (unsigned q = lane; q < tileWords * piecesPerWord; q += 64) {
            unsigned w = q / piecesPerWord, piece = q - w * piecesPerWord;
            size_t word = t * tileWords + w;
            size_t start = randomOrder ? (word * 2654435761ull) % (srcPieces - 32) : (word * 130) / 16;
            uint4 v = src[start + piece];
            acc += v.x ^ v.y ^ v.z ^ v.w;
        }
        size_t base = t * tilePieces;
        for (unsigned q = lane; q < tilePieces; q += 64)
            if (base + q < n4) out[base + q] = make_float4(1, 2, 3, 4);
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
int main(int argc, char** argv) {
    const char* mode = argc > 1 ? argv[1] : "steps";
    unsigned a = argc > 2 ? atoi(argv[2]) : 1, b = argc > 3 ? atoi(argv[3]) : 256;
    const size_t words = 2196017, n4 = words * 75;
    float4* out; if (hipMalloc(&out, n4 * 16 + (1 << 20)) != hipSuccess) return 1;
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    int cus = prop.multiProcessorCount;
    size_t srcBytes = size_t(290) << 20; uint4* src; unsigned* sink;
    if (hipMalloc(&src, srcBytes) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) return 1;
    hipMemset(src, 1, srcBytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int i = 0; i < 5; ++i) {
        hipEventRecord(e0);
        if (!strcmp(mode, "steps")) {
            size_t per = size_t(b) * a, blocks = (n4 + per - 1) / per;
            hipLaunchKernelGGL(fill_steps, dim3((unsigned)blocks), dim3(b), 0, 0, out, n4, a);
        } else if (!strcmp(mode, "tiles")) {
            hipLaunchKernelGGL(fill_tiles, dim3(cus * a / 8), dim3(512), 0, 0, out, n4, 600u);
        } else {
            hipLaunchKernelGGL(fill_tiles_gather, dim3(cus * a / 8), dim3(512), 0, 0, out, n4, src, srcBytes / 16, (int)b, sink);
        }
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    printf("%s %u %u: best %.3f ms\n", mode, a, b, best);
    return 0;
}

// Does the allocation type of the OUTPUT change what a gather from a 128 / 290 MB source costs next to the write stream?
// (i.e. can the source stay in the 256 MB Infinity Cache if the stores do not allocate there?)
// Build: hipcc --offload-arch=gfx950 -O3 wmall.hip -o wmall
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
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
            size_t start = randomOrder ? (word * 2654435761ull) % (srcPieces - 32) : ((word * 130) / 16) % (srcPieces - 32);
            uint4 v = src[start + piece];
            acc += v.x ^ v.y ^ v.z ^ v.w;
        }
        size_t base = t * tilePieces;
        for (unsigned q = lane; q < tilePieces; q += 64)
            if (base + q < n4) out[base + q] = make_float4(1, 2, 3, 4);
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
template <typename F> float timeIt(F f) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) f();
    std::vector<float> ms;
    for (int i = 0; i < 10; ++i) { hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b); float t; hipEventElapsedTime(&t, a, b); ms.push_back(t); }
    std::sort(ms.begin(), ms.end()); return ms[ms.size() / 2];
}
int main() {
    const size_t words = 2196017, n4 = words * 75, bytes = n4 * 16 + (1 << 20);
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    int cus = prop.multiProcessorCount;
    unsigned* sink; hipMalloc(&sink, 64);
    const char* outNames[] = {"out: hipMalloc", "out: Uncached", "out: Finegrained"};
    const char* srcNames[] = {"src: hipMalloc", "src: Uncached"};
    for (int outKind = 0; outKind < 3; ++outKind) for (int srcKind = 0; srcKind < 2; ++srcKind) for (size_t mb : {32u, 128u, 290u}) {
        float4* out = nullptr; uint4* src = nullptr; size_t srcBytes = mb << 20;
        hipError_t e = outKind == 0 ? hipMalloc(&out, bytes) : hipExtMallocWithFlags((void**)&out, bytes, outKind == 1 ? hipDeviceMallocUncached : hipDeviceMallocFinegrained);
        if (e == hipSuccess) e = srcKind == 0 ? hipMalloc(&src, srcBytes) : hipExtMallocWithFlags((void**)&src, srcBytes, hipDeviceMallocUncached);
        if (e != hipSuccess) { printf("allocation failed\n"); (void)hipGetLastError(); continue; }
        hipMemset(src, 1, srcBytes); hipMemset(out, 0, bytes); hipDeviceSynchronize();
        for (int randomOrder : {0, 1}) {
            float ms = timeIt([&] { hipLaunchKernelGGL(fill_tiles_gather, dim3(cus * 4), dim3(512), 0, 0, out, n4, src, srcBytes / 16, randomOrder, sink); });
            printf("%-18s %-16s source %3zu MB %-10s %.3f ms\n", outNames[outKind], srcNames[srcKind], mb, randomOrder ? "random" : "sequential", ms); fflush(stdout);
        }
        hipFree(out); hipFree(src);
    }
    return 0;
}

// Build: hipcc --offload-arch=gfx950 -O3 wwindow.hip -o wwindow
// Window-size hypothesis: multi-step one-shot blocks with occupancy limited through LDS
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void oneshot_block_interleaved(float4* out, size_t n4, unsigned steps) {
    extern __shared__ float pad[];
    if (threadIdx.x == 9999) pad[0] = 1;
    size_t base = size_t(blockIdx.x) * steps * blockDim.x;
    for (unsigned s = 0; s < steps; ++s) {
        size_t i = base + size_t(s) * blockDim.x + threadIdx.x;
        if (i < n4) out[i] = make_float4(1, 2, 3, 4);
    }
}
__global__ void oneshot_wave_major(float4* out, size_t n4, unsigned steps) {
    extern __shared__ float pad[];
    if (threadIdx.x == 9999) pad[0] = 1;
    unsigned lane = threadIdx.x & 63;
    size_t wave = (size_t(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
    size_t base = wave * steps * 64;
    for (unsigned s = 0; s < steps; ++s) {
        size_t i = base + s * 64 + lane;
        if (i < n4) out[i] = make_float4(1, 2, 3, 4);
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
    double gb = n4 * 16 / 1e9;
    auto show = [&](const char* name, float ms) { printf("%-84s %.3f ms  %.2f TB/s\n", name, ms, gb / ms); fflush(stdout); };
    CHECK(hipFuncSetAttribute((const void*)oneshot_block_interleaved, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CHECK(hipFuncSetAttribute((const void*)oneshot_wave_major, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    char name[200];
    for (unsigned threads : {64u, 256u, 512u}) for (unsigned steps : {1u, 3u, 10u}) for (unsigned ldsKb : {0u, 20u, 40u, 80u, 160u}) {
        size_t per = size_t(threads) * steps; size_t blocks = (n4 + per - 1) / per;
        unsigned blocksPerCu = ldsKb ? 160 / ldsKb : 99;
        double windowMb = 256.0 * std::min<unsigned>(blocksPerCu, 2048 / threads) * per * 16 / 1e6;
        snprintf(name, sizeof name, "one-shot %4u thr, %2u steps, block-interleaved, LDS %3u KB (window ~%5.1f MB)", threads, steps, ldsKb, windowMb);
        show(name, timeIt([&] { hipLaunchKernelGGL(oneshot_block_interleaved, dim3((unsigned)blocks), dim3(threads), ldsKb * 1024, 0, out, n4, steps); }));
        if (steps > 1 && threads > 64) {
            snprintf(name, sizeof name, "one-shot %4u thr, %2u steps, wave-major,        LDS %3u KB (window ~%5.1f MB)", threads, steps, ldsKb, windowMb);
            show(name, timeIt([&] { hipLaunchKernelGGL(oneshot_wave_major, dim3((unsigned)blocks), dim3(threads), ldsKb * 1024, 0, out, n4, steps); }));
        }
    }
    return 0;
}

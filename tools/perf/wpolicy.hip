// Store cache-policy bits (gfx950: sc0, sc1, nt) on the 3-stores-per-thread fill and on persistent 9600-B tiles
// Build: hipcc --offload-arch=gfx950 -O3 wpolicy.hip -o wpolicy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int POLICY> __device__ __forceinline__ void store(float4* address, f32x4 v) {
    if (POLICY == 0) asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(address), "v"(v) : "memory");
    if (POLICY == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" :: "v"(address), "v"(v) : "memory");
    if (POLICY == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0" :: "v"(address), "v"(v) : "memory");
    if (POLICY == 3) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(address), "v"(v) : "memory");
    if (POLICY == 4) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(address), "v"(v) : "memory");
    if (POLICY == 5) asm volatile("global_store_dwordx4 %0, %1, off sc0 nt" :: "v"(address), "v"(v) : "memory");
    if (POLICY == 6) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" :: "v"(address), "v"(v) : "memory");
    if (POLICY == 7) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" :: "v"(address), "v"(v) : "memory");
}
template <int POLICY> __global__ void fill_steps(float4* out, size_t n4, unsigned steps) {
    size_t base = size_t(blockIdx.x) * steps * blockDim.x;
    f32x4 v = {1, 2, 3, 4};
    for (unsigned s = 0; s < steps; ++s) {
        size_t i = base + size_t(s) * blockDim.x + threadIdx.x;
        if (i < n4) store<POLICY>(out + i, v);
    }
}
template <int POLICY> __global__ void fill_tiles(float4* out, size_t n4, unsigned tilePieces) {
    unsigned lane = threadIdx.x & 63;
    size_t wave = (size_t(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
    size_t waves = (size_t(gridDim.x) * blockDim.x) >> 6;
    size_t tiles = (n4 + tilePieces - 1) / tilePieces;
    f32x4 v = {1, 2, 3, 4};
    for (size_t t = wave; t < tiles; t += waves) {
        size_t base = t * tilePieces;
        for (unsigned q = lane; q < tilePieces; q += 64)
            if (base + q < n4) store<POLICY>(out + base + q, v);
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
    const char* names[8] = {"(none)", "nt", "sc0", "sc1", "sc0 sc1", "sc0 nt", "sc1 nt", "sc0 sc1 nt"};
    auto show = [&](const char* what, int policy, float ms) { printf("%-44s %-12s %.3f ms  %.2f TB/s\n", what, names[policy], ms, gb / ms); fflush(stdout); };
#define RUN(P) \
    for (unsigned steps : {1u, 3u}) { size_t per = 256 * steps, blocks = (n4 + per - 1) / per; char w[64]; snprintf(w, sizeof w, "one-shot 256 thr, %u steps", steps); \
        show(w, P, timeIt([&] { hipLaunchKernelGGL(fill_steps<P>, dim3((unsigned)blocks), dim3(256), 0, 0, out, n4, steps); })); } \
    show("persistent 9600-B tiles, 32 waves/CU", P, timeIt([&] { hipLaunchKernelGGL(fill_tiles<P>, dim3(cus * 4), dim3(512), 0, 0, out, n4, 600u); }));
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7)
    return 0;
}

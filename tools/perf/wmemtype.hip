// Does the allocation type of the output buffer change the write rate? (default / fine-grained / uncached / managed)
// Build: hipcc --offload-arch=gfx950 -O3 wmemtype.hip -o wmemtype
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
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
    double gb = n4 * 16 / 1e9;
    struct Kind { const char* name; int which; } kinds[] = {{"hipMalloc", 0}, {"hipExtMallocWithFlags(Finegrained)", 1}, {"hipExtMallocWithFlags(Uncached)", 2}, {"hipMallocManaged", 3}, {"hipExtMallocWithFlags(Contiguous)", 4}};
    for (auto kind : kinds) {
        float4* out = nullptr; hipError_t e = hipSuccess;
        if (kind.which == 0) e = hipMalloc(&out, bytes);
        if (kind.which == 1) e = hipExtMallocWithFlags((void**)&out, bytes, hipDeviceMallocFinegrained);
        if (kind.which == 2) e = hipExtMallocWithFlags((void**)&out, bytes, hipDeviceMallocUncached);
        if (kind.which == 3) { e = hipMallocManaged(&out, bytes); if (e == hipSuccess) hipMemPrefetchAsync(out, bytes, 0, 0); }
        if (kind.which == 4) e = hipExtMallocWithFlags((void**)&out, bytes, hipDeviceMallocContiguous);
        if (e != hipSuccess) { printf("%-40s allocation failed: %s\n", kind.name, hipGetErrorString(e)); (void)hipGetLastError(); continue; }
        hipMemset(out, 0, bytes); hipDeviceSynchronize();
        for (unsigned steps : {1u, 3u}) {
            size_t per = 256 * steps, blocks = (n4 + per - 1) / per;
            float ms = timeIt([&] { hipLaunchKernelGGL(fill_steps, dim3((unsigned)blocks), dim3(256), 0, 0, out, n4, steps); });
            printf("%-40s one-shot 256 thr, %u steps          %.3f ms  %.2f TB/s\n", kind.name, steps, ms, gb / ms); fflush(stdout);
        }
        float ms = timeIt([&] { hipLaunchKernelGGL(fill_tiles, dim3(cus * 4), dim3(512), 0, 0, out, n4, 600u); });
        printf("%-40s persistent 9600-B tiles, 32 waves/CU %.3f ms  %.2f TB/s\n", kind.name, ms, gb / ms); fflush(stdout);
        hipFree(out);
    }
    return 0;
}

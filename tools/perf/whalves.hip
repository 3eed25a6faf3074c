// configs[4] write pattern: 500 k rows of 2400 B, written as two half-rows of 1200 B.
//   separate : kernel A writes the first halves of all rows, kernel B the second halves (today's two launches)
//   fused    : every wavefront writes the first halves of its 8-row tile, then the second halves (a fused launch)
//   dense    : every wavefront writes its tile's 8 whole rows contiguously (the bound)
// Build: hipcc --offload-arch=gfx950 -O3 whalves.hip -o whalves
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__device__ inline void writeHalf(float4* out, size_t row, unsigned half, unsigned lane, unsigned tileRows) {
    // pieces of the tile's half rows: tileRows x 75 pieces of 16 B, row stride 150 pieces
    for (unsigned q = lane; q < tileRows * 75; q += 64) {
        unsigned w = q / 75, c = q - w * 75;
        out[(row + w) * 150 + half * 75 + c] = make_float4(1, 2, 3, 4);
    }
}
__global__ void halves(float4* out, size_t rows, int mode, unsigned gapSleeps) {
    unsigned lane = threadIdx.x & 63;
    size_t wave = (size_t(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
    size_t waves = (size_t(gridDim.x) * blockDim.x) >> 6;
    size_t tiles = (rows + 7) / 8;
    for (size_t t = wave; t < tiles; t += waves) {
        size_t row = t * 8; unsigned tileRows = (unsigned)min((size_t)8, rows - row);
        if (mode == 0 || mode == 2) writeHalf(out, row, 0, lane, tileRows);
        if (mode == 2) for (unsigned s = 0; s < gapSleeps; ++s) __builtin_amdgcn_s_sleep(64);   // ~1 us each: the other model's decode
        if (mode == 1 || mode == 2) writeHalf(out, row, 1, lane, tileRows);
        if (mode == 3) for (unsigned q = lane; q < tileRows * 150; q += 64) out[row * 150 + q] = make_float4(1, 2, 3, 4);
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
    const size_t rows = 500000;
    float4* out; if (hipMalloc(&out, rows * 2400 + (1 << 20)) != hipSuccess) return 1;
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    dim3 grid(prop.multiProcessorCount * 4), block(512);
    printf("separate launches (A then B):          %.3f ms\n", timeIt([&] { hipLaunchKernelGGL(halves, grid, block, 0, 0, out, rows, 0, 0u); hipLaunchKernelGGL(halves, grid, block, 0, 0, out, rows, 1, 0u); }));
    for (unsigned gap : {0u, 2u, 5u, 10u, 20u})
        printf("fused, %2u us between the two halves:    %.3f ms\n", gap, timeIt([&] { hipLaunchKernelGGL(halves, grid, block, 0, 0, out, rows, 2, gap); }));
    printf("dense whole rows:                      %.3f ms\n", timeIt([&] { hipLaunchKernelGGL(halves, grid, block, 0, 0, out, rows, 3, 0u); }));
    return 0;
}

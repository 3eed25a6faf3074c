// One write pattern per run (for rocprofv3 --pmc): wsingle <steps> <threads> [persistent]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void fill_steps(float4* out, size_t n4, unsigned steps) {
    size_t base = size_t(blockIdx.x) * steps * blockDim.x;
    for (unsigned s = 0; s < steps; ++s) {
        size_t i = base + size_t(s) * blockDim.x + threadIdx.x;
        if (i < n4) out[i] = make_float4(1, 2, 3, 4);
    }
}
int main(int argc, char** argv) {
    unsigned steps = argc > 1 ? atoi(argv[1]) : 1, threads = argc > 2 ? atoi(argv[2]) : 256;
    const size_t words = 2196017, n4 = words * 75;
    float4* out; if (hipMalloc(&out, n4 * 16 + (1 << 20)) != hipSuccess) return 1;
    size_t per = size_t(threads) * steps; size_t blocks = (n4 + per - 1) / per;
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(fill_steps, dim3((unsigned)blocks), dim3(threads), 0, 0, out, n4, steps);
    hipDeviceSynchronize();
    printf("done steps=%u threads=%u\n", steps, threads);
    return 0;
}

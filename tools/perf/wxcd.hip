// Does the 4-KiB chunk -> XCD assignment of a write stream matter? (MI355X: 8 XCDs, blocks dispatched round robin)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ inline unsigned xccId() { return __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15; }   // HW_REG_XCC_ID[3:0]

// one chunk of `blockDim.x * 16` bytes per block; the chunk's class (index mod 8) is (xcd + shift) mod 8
__global__ void fill_by_xcd(float4* out, size_t n4, unsigned shift, unsigned* histogram) {
    unsigned xcd = xccId();
    size_t group = blockIdx.x / 8;
    size_t chunk = group * 8 + ((xcd + shift) & 7);
    size_t i = chunk * blockDim.x + threadIdx.x;
    if (i < n4) out[i] = make_float4(1, 2, 3, 4);
    if (histogram && threadIdx.x == 0) atomicAdd(&histogram[(blockIdx.x & 7) * 8 + xcd], 1u);
}
// same without reading the XCD: class = (blockIdx + shift) mod 8
__global__ void fill_by_block(float4* out, size_t n4, unsigned shift) {
    size_t group = blockIdx.x / 8;
    size_t chunk = group * 8 + ((blockIdx.x + shift) & 7);
    size_t i = chunk * blockDim.x + threadIdx.x;
    if (i < n4) out[i] = make_float4(1, 2, 3, 4);
}
// chunk permutation inside groups of `span` chunks (span = 8: within the round-robin group, 64: beyond it)
__global__ void fill_permuted(float4* out, size_t n4, unsigned span, unsigned mult) {
    size_t group = blockIdx.x / span;
    size_t chunk = group * span + ((blockIdx.x % span) * mult) % span;
    size_t i = chunk * blockDim.x + threadIdx.x;
    if (i < n4) out[i] = make_float4(1, 2, 3, 4);
}
// persistent: every block finds its XCD and its rank inside the XCD, then walks chunks of its class
__global__ void fill_persistent_xcd(float4* out, size_t n4, unsigned chunkPieces, unsigned shift, unsigned* ranks, int useXcd) {
    __shared__ unsigned rankShared, xcdShared;
    if (threadIdx.x == 0) {
        unsigned xcd = useXcd ? xccId() : (blockIdx.x & 7);
        xcdShared = xcd;
        rankShared = atomicAdd(&ranks[xcd], 1u);
    }
    __syncthreads();
    unsigned cls = (xcdShared + shift) & 7, rank = rankShared;
    size_t chunks = (n4 + chunkPieces - 1) / chunkPieces;
    unsigned perXcd = gridDim.x / 8;
    for (size_t k = rank; ; k += perXcd) {
        size_t chunk = k * 8 + cls;
        if (chunk >= chunks) break;
        size_t base = chunk * chunkPieces;
        for (unsigned q = threadIdx.x; q < chunkPieces; q += blockDim.x)
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
    printf("out = %p\n", (void*)out);
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount;
    double gb = n4 * 16 / 1e9;
    auto show = [&](const char* name, float ms) { printf("%-72s %.3f ms  %.2f TB/s\n", name, ms, gb / ms); fflush(stdout); };
    unsigned* histogram; CHECK(hipMalloc(&histogram, 64 * 4)); CHECK(hipMemset(histogram, 0, 64 * 4));
    unsigned* ranks; CHECK(hipMalloc(&ranks, 8 * 4));
    for (unsigned threads : {64u, 128u, 256u, 512u, 1024u}) {
        size_t blocks = (n4 + threads - 1) / threads; blocks = (blocks + 7) / 8 * 8;
        char name[128];
        for (unsigned shift : {0u, 1u, 3u, 4u}) {
            snprintf(name, sizeof name, "one %5u-B chunk per block, class = (blockIdx + %u) %% 8", threads * 16, shift);
            show(name, timeIt([&] { hipLaunchKernelGGL(fill_by_block, dim3((unsigned)blocks), dim3(threads), 0, 0, out, n4, shift); }));
        }
        for (unsigned shift : {0u, 1u, 3u, 4u}) {
            snprintf(name, sizeof name, "one %5u-B chunk per block, class = (XCC_ID + %u) %% 8", threads * 16, shift);
            show(name, timeIt([&] { hipLaunchKernelGGL(fill_by_xcd, dim3((unsigned)blocks), dim3(threads), 0, 0, out, n4, shift, (unsigned*)nullptr); }));
        }
        for (unsigned span : {8u, 64u, 1024u}) {
            snprintf(name, sizeof name, "one %5u-B chunk per block, chunks permuted (x5 mod span) in spans of %u", threads * 16, span);
            show(name, timeIt([&] { hipLaunchKernelGGL(fill_permuted, dim3((unsigned)(blocks / span * span)), dim3(threads), 0, 0, out, n4, span, span == 8 ? 5u : (span == 64 ? 37u : 613u)); }));
        }
    }
    {
        size_t blocks = ((n4 + 255) / 256 + 7) / 8 * 8;
        hipLaunchKernelGGL(fill_by_xcd, dim3((unsigned)blocks), dim3(256), 0, 0, out, n4, 0u, histogram);
        unsigned h[64]; CHECK(hipMemcpy(h, histogram, sizeof h, hipMemcpyDeviceToHost));
        printf("blockIdx%%8 (rows) x XCC_ID (columns):\n");
        for (int r = 0; r < 8; ++r) { for (int c = 0; c < 8; ++c) printf("%8u", h[r * 8 + c]); printf("\n"); }
    }
    for (unsigned chunkBytes : {2048u, 4096u, 8192u, 16384u}) for (unsigned threads : {256u, 512u}) for (int perCu : {2, 4, 8}) for (int useXcd : {1, 0}) for (unsigned shift : {0u, 3u}) {
        if (!useXcd && shift) continue;
        char name[160];
        snprintf(name, sizeof name, "persistent %u thr x %d blocks/CU, %5u-B chunks, class = (%s + %u) %% 8", threads, perCu, chunkBytes, useXcd ? "XCC_ID" : "blockIdx", shift);
        show(name, timeIt([&] { hipMemsetAsync(ranks, 0, 32, 0); hipLaunchKernelGGL(fill_persistent_xcd, dim3(cus * perCu), dim3(threads), 0, 0, out, n4, chunkBytes / 16, shift, ranks, useXcd); }));
    }
    return 0;
}

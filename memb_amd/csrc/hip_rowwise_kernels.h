// dequant_uniform / gather_full: one 16-byte piece of an output row per lane.
//
// Device code of libmemb_hip.so (gfx950 / CDNA4). Included by memb_hip.hip only,
// inside its anonymous namespace; see that file for the overview.
#pragma once

// ---------------------------------------------------------------------------
// dequant_uniform / gather_full
// ---------------------------------------------------------------------------

struct UniformParams {
    uint32_t accumulate;
    float divisor;
    const uint32_t* rows;
    float* out;
    unsigned long long n;
    unsigned long long ld;
    unsigned long long colOff;
    const uint8_t* values;   // dense [nRows][dim]
    const float2* minMax;    // [nRows]
    unsigned long long nRows;
    uint32_t dim;
    uint32_t wordsPerBlock;
    uint32_t pieceMagic;     // ceil(2^32 / (dim / 4)), vector path
    float levels;
};

// reference src/uniform_compression.cpp:70-71, evaluated left to right in fp32:
// sub, mul, div, add -- each correctly rounded, nothing fused, subnormals kept.
__device__ __forceinline__ float dequant(float minValue, float range, uint32_t v, float levels)
{
    const float scaled = mulRn(range, static_cast<float>(v));
    return addRn(minValue, __fdiv_rn(scaled, levels));
}

constexpr uint32_t ROWWISE_MAX_WORDS = 64;   // words per block of the row-wise kernels
constexpr int ROWWISE_BATCH = 4;             // 16-byte pieces a thread keeps in flight

// Row-wise kernels (uniform, full): a block first stages the row ids (and the
// per-row constants) of its words in LDS -- one dependent pair of global loads
// per block instead of per piece -- then every thread keeps ROWWISE_BATCH value
// loads in flight before it converts and stores.
template <bool VEC4>
__global__ void dequant_uniform(UniformParams p)
{
    __shared__ uint32_t rowLds[ROWWISE_MAX_WORDS];
    __shared__ float2 minMaxLds[ROWWISE_MAX_WORDS];
    const unsigned long long blockBase = static_cast<unsigned long long>(blockIdx.x) * p.wordsPerBlock;
    const uint32_t blockWords =
        static_cast<uint32_t>(min(static_cast<unsigned long long>(p.wordsPerBlock), p.n - blockBase));
    if (threadIdx.x < blockWords) {
        const uint32_t row = p.rows[blockBase + threadIdx.x];
        rowLds[threadIdx.x] = row;
        minMaxLds[threadIdx.x] = row < p.nRows ? p.minMax[row] : make_float2(0.f, 0.f);
    }
    __syncthreads();

    if (VEC4) {
        const uint32_t piecesPerWord = p.dim / 4;
        const uint32_t pieces = blockWords * piecesPerWord;
        for (uint32_t q0 = threadIdx.x; q0 < pieces; q0 += blockDim.x * ROWWISE_BATCH) {
            uint32_t word[ROWWISE_BATCH];
            uint32_t column[ROWWISE_BATCH];
            uint32_t packed[ROWWISE_BATCH];
#pragma unroll
            for (int u = 0; u < ROWWISE_BATCH; ++u) {
                const uint32_t q = min(q0 + u * blockDim.x, pieces - 1);
                word[u] = fastDivide(q, p.pieceMagic, piecesPerWord);
                column[u] = q - word[u] * piecesPerWord;
                const uint32_t row = rowLds[word[u]];
                packed[u] = 0;
                if (row < p.nRows) {
                    const uint32_t* source = reinterpret_cast<const uint32_t*>(
                        p.values + static_cast<unsigned long long>(row) * p.dim + 4 * column[u]);
                    // non-temporal: a row's bytes are read once per lookup (500 k random rows 0.167 vs 0.178 ms,
                    // the key-order dump 0.164 vs 0.162)
                    packed[u] = __builtin_nontemporal_load(source);
                }
            }
#pragma unroll
            for (int u = 0; u < ROWWISE_BATCH; ++u) {
                if (q0 + u * blockDim.x < pieces) {
                    float4 f = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (rowLds[word[u]] < p.nRows) {
                        const float2 mm = minMaxLds[word[u]];
                        const float range = subRn(mm.y, mm.x);
                        f.x = dequant(mm.x, range, packed[u] & 0xff, p.levels);
                        f.y = dequant(mm.x, range, (packed[u] >> 8) & 0xff, p.levels);
                        f.z = dequant(mm.x, range, (packed[u] >> 16) & 0xff, p.levels);
                        f.w = dequant(mm.x, range, packed[u] >> 24, p.levels);
                    }
                    float* dst = p.out + (blockBase + word[u]) * p.ld + p.colOff + 4 * column[u];
                    if (p.accumulate || p.divisor != 0.f) {
                        f = epilogue4(f, dst, p.accumulate, p.divisor);
                    }
                    *reinterpret_cast<float4*>(dst) = f;
                }
            }
        }
    } else {
        const uint32_t total = blockWords * p.dim;
        for (uint32_t q = threadIdx.x; q < total; q += blockDim.x) {
            const uint32_t w = q / p.dim;
            const uint32_t c = q - w * p.dim;
            const uint32_t row = rowLds[w];
            float f = 0.f;
            if (row < p.nRows) {
                const float2 mm = minMaxLds[w];
                const float range = subRn(mm.y, mm.x);
                f = dequant(mm.x, range, p.values[static_cast<unsigned long long>(row) * p.dim + c], p.levels);
            }
            float* dst = p.out + (blockBase + w) * p.ld + p.colOff + c;
            if (p.accumulate || p.divisor != 0.f) {
                f = epilogue(f, dst, p.accumulate, p.divisor);
            }
            *dst = f;
        }
    }
}

struct FullParams {
    uint32_t accumulate;
    float divisor;
    const uint32_t* rows;
    float* out;
    unsigned long long n;
    unsigned long long ld;
    unsigned long long colOff;
    const float* values;     // dense [nRows][dim]
    unsigned long long nRows;
    uint32_t dim;
    uint32_t wordsPerBlock;
    uint32_t pieceMagic;
};

template <bool VEC4>
__global__ void gather_full(FullParams p)
{
    __shared__ uint32_t rowLds[ROWWISE_MAX_WORDS];
    const unsigned long long blockBase = static_cast<unsigned long long>(blockIdx.x) * p.wordsPerBlock;
    const uint32_t blockWords =
        static_cast<uint32_t>(min(static_cast<unsigned long long>(p.wordsPerBlock), p.n - blockBase));
    if (threadIdx.x < blockWords) {
        rowLds[threadIdx.x] = p.rows[blockBase + threadIdx.x];
    }
    __syncthreads();
    if (VEC4) {
        const uint32_t piecesPerWord = p.dim / 4;
        const uint32_t pieces = blockWords * piecesPerWord;
        for (uint32_t q0 = threadIdx.x; q0 < pieces; q0 += blockDim.x * ROWWISE_BATCH) {
            uint32_t word[ROWWISE_BATCH];
            uint32_t column[ROWWISE_BATCH];
            float4 f[ROWWISE_BATCH];
#pragma unroll
            for (int u = 0; u < ROWWISE_BATCH; ++u) {
                const uint32_t q = min(q0 + u * blockDim.x, pieces - 1);
                word[u] = fastDivide(q, p.pieceMagic, piecesPerWord);
                column[u] = q - word[u] * piecesPerWord;
                const uint32_t row = rowLds[word[u]];
                f[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (row < p.nRows) {
                    // (non-temporal loads bought nothing here: 0.246 ms either way on 500 k random rows)
                    f[u] = *reinterpret_cast<const float4*>(
                        p.values + static_cast<unsigned long long>(row) * p.dim + 4 * column[u]);
                }
            }
#pragma unroll
            for (int u = 0; u < ROWWISE_BATCH; ++u) {
                if (q0 + u * blockDim.x < pieces) {
                    float* dst = p.out + (blockBase + word[u]) * p.ld + p.colOff + 4 * column[u];
                    if (p.accumulate || p.divisor != 0.f) {
                        f[u] = epilogue4(f[u], dst, p.accumulate, p.divisor);
                    }
                    *reinterpret_cast<float4*>(dst) = f[u];
                }
            }
        }
    } else {
        const uint32_t total = blockWords * p.dim;
        for (uint32_t q = threadIdx.x; q < total; q += blockDim.x) {
            const uint32_t w = q / p.dim;
            const uint32_t c = q - w * p.dim;
            const uint32_t row = rowLds[w];
            float f = row < p.nRows ? p.values[static_cast<unsigned long long>(row) * p.dim + c] : 0.f;
            float* dst = p.out + (blockBase + w) * p.ld + p.colOff + c;
            if (p.accumulate || p.divisor != 0.f) {
                f = epilogue(f, dst, p.accumulate, p.divisor);
            }
            *dst = f;
        }
    }
}

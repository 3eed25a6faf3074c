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
    // ROW RECORDS: row r owns regionPieces 16-byte pieces at piece r * regionPieces: piece 0 = {min, max, 0, 0},
    // the quantised weights (one byte each, zero padded) from piece 1 on. A lookup is one address computation
    // and one contiguous fetch; the tile kernel moves whole regions into LDS.
    const uint4* records;
    uint32_t regionPieces;   // 1 + ceil(dim / 16)
    unsigned long long nRows;
    uint32_t dim;
    uint32_t wordsPerBlock;
    uint32_t pieceMagic;     // ceil(2^32 / (dim / 4)), vector path
    float levels;
    // dequant_uniform_tile
    uint32_t wordsPerWave;   // words of a wavefront's tile
    uint32_t regionMagic;    // fastDivide magic for regionPieces
};

__device__ __forceinline__ const uint8_t* uniformRegion(const UniformParams& p, uint32_t row)
{
    return reinterpret_cast<const uint8_t*>(p.records + static_cast<unsigned long long>(row) * p.regionPieces);
}

// reference src/uniform_compression.cpp:70-71, evaluated left to right in fp32:
// sub, mul, div, add -- each correctly rounded, nothing fused, subnormals kept.
// (The division is hipcc's correctly rounded expansion, about ten of the 21.7 vector instructions a weight costs. Round 4
// built the cheaper exact form for divisors 1 .. 255 -- q = RN(a * RN(1 / levels)), r = a - q * levels by one FMA,
// RN(q + r * RN(1 / levels)) by another; correctly rounded inside a safe exponent range, the full division outside it;
// bit-identical on 115 million operands on the CPU and in tests/test_gpu_parity.py::
// test_uniform_division_for_every_level_count on the device -- and it bought nothing: 0.162-0.168 ms against
// 0.158-0.165 for the 500 000-word dump, +3 % on 100 000 rows (round 4, batch 8, profiles/r04_experiments.txt; twice the registers for the two
// paths). The vector ALU is not what these kernels wait for. Removed; the test stays.)
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
        minMaxLds[threadIdx.x] = row < p.nRows ? *reinterpret_cast<const float2*>(uniformRegion(p, row)) : make_float2(0.f, 0.f);
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
                    const uint32_t* source = reinterpret_cast<const uint32_t*>(uniformRegion(p, row) + 16 + 4 * column[u]);
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
                f = dequant(mm.x, range, uniformRegion(p, row)[16 + c], p.levels);
            }
            float* dst = p.out + (blockBase + w) * p.ld + p.colOff + c;
            if (p.accumulate || p.divisor != 0.f) {
                f = epilogue(f, dst, p.accumulate, p.divisor);
            }
            *dst = f;
        }
    }
}

// Row of the tile's word `lane` (lanes past the tile's words and past the batch: MISSING).
__device__ __forceinline__ uint32_t loadUniformRow(const UniformParams& p, unsigned long long tile, uint32_t lane)
{
    const unsigned long long index = tile * p.wordsPerWave + lane;
    return lane < p.wordsPerWave && index < p.n ? p.rows[index] : 0xFFFFFFFFu;
}

// One tile per wavefront at a time, short-lived blocks -- the shape that won every large batch of the trained storage in
// round 4 (decode_trained), for the uniform one: a wavefront owns wordsPerWave words (8 at dim 300: 9600 bytes of output, a
// whole number of 128-byte lines), fetches their row regions with 16-byte loads into its own LDS slots, converts and
// stores, and exits. No tables to copy, so no block barrier at all. Round 3's form was a persistent pipeline fed by
// LDS-DMA (dequant_uniform_persistent, 0.58-0.60 of the HBM peak on the 500 000-word dump); the memory pattern alone
// (tools/perf/ceilings.hip, uniform_rows_per_wave<8>) runs that dump in 0.137 ms where the pipeline took 0.157-0.164, and
// this kernel takes 0.136: -17 % on the dump, -12 % shuffled, -16 % on 2 M rows, ties at 1 k and 10 k rows; the pipeline
// was 8 % ahead at 100 000 rows (two to four tiles per resident wavefront, as for the trained storage), which no BASELINE
// configuration is: removed (round 4, batch 16).
template <bool FLAT>
__global__ void dequant_uniform_tile(UniformParams p)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t uniformLds[];
    const uint32_t lane = threadIdx.x & (WAVE - 1);
    const uint32_t wave = threadIdx.x / WAVE;
    const unsigned long long tile = static_cast<unsigned long long>(blockIdx.x) * (blockDim.x / WAVE) + wave;
    const unsigned long long tileBase = tile * p.wordsPerWave;
    if (tileBase >= p.n) {
        return;
    }
    const uint32_t tileWords = static_cast<uint32_t>(min(static_cast<unsigned long long>(p.wordsPerWave), p.n - tileBase));
    uint32_t* slots = uniformLds + wave * p.wordsPerWave * p.regionPieces * 4;
    const uint32_t row = loadUniformRow(p, tile, lane);              // lane w: the row of the tile's word w
    const unsigned long long absent = __ballot(row >= p.nRows);      // bit w: word w is missing (or past the batch)
    const uint32_t start = row < p.nRows ? row * p.regionPieces : 0u;   // absent words fetch row 0 and store zeros
    const uint32_t totalPieces = tileWords * p.regionPieces;
    for (uint32_t q0 = 0; q0 < totalPieces; q0 += WAVE) {
        const uint32_t q = min(q0 + lane, totalPieces - 1);
        const uint32_t w = fastDivide(q, p.regionMagic, p.regionPieces);
        const uint32_t wordStart = __shfl(start, w);   // (every lane active here)
        if (q0 + lane < totalPieces) {
            const uint4 piece = p.records[static_cast<unsigned long long>(wordStart) + (q - w * p.regionPieces)];
            *reinterpret_cast<uint4*>(slots + 4 * q) = piece;
        }
    }
    waveLdsFence();

    const uint32_t piecesPerWord = p.dim / 4;
    const bool hasEpilogue = p.accumulate || p.divisor != 0.f;
    const uint32_t pieces = tileWords * piecesPerWord;
    float* tileOut = p.out + tileBase * p.ld + p.colOff;
    constexpr int BURST = 4;
    for (uint32_t q0 = lane; q0 < pieces; q0 += WAVE * BURST) {
        uint32_t packed[BURST];
        float2 minMax[BURST];
        uint32_t word[BURST];
        uint32_t column[BURST];
#pragma unroll
        for (int b = 0; b < BURST; ++b) {
            const uint32_t q = min(q0 + WAVE * b, pieces - 1);
            word[b] = fastDivide(q, p.pieceMagic, piecesPerWord);
            column[b] = q - word[b] * piecesPerWord;
            const uint32_t* region = slots + word[b] * p.regionPieces * 4;
            minMax[b] = *reinterpret_cast<const float2*>(region);
            packed[b] = region[4 + column[b]];
        }
#pragma unroll
        for (int b = 0; b < BURST; ++b) {
            if (q0 + WAVE * b < pieces) {
                float4 f = make_float4(0.f, 0.f, 0.f, 0.f);
                if (!((absent >> word[b]) & 1)) {
                    const float range = subRn(minMax[b].y, minMax[b].x);
                    f.x = dequant(minMax[b].x, range, packed[b] & 0xff, p.levels);
                    f.y = dequant(minMax[b].x, range, (packed[b] >> 8) & 0xff, p.levels);
                    f.z = dequant(minMax[b].x, range, (packed[b] >> 16) & 0xff, p.levels);
                    f.w = dequant(minMax[b].x, range, packed[b] >> 24, p.levels);
                }
                float* destination = FLAT ? tileOut + 4 * static_cast<size_t>(q0 + WAVE * b)
                                          : tileOut + word[b] * p.ld + 4 * column[b];
                if (hasEpilogue) {
                    f = epilogue4(f, destination, p.accumulate, p.divisor);
                }
                *reinterpret_cast<float4*>(destination) = f;
            }
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

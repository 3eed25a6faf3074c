// Write side on the device: quantise_rows / stream_lengths / pack_streams.
//
// Device code of libmemb_hip.so (gfx950 / CDNA4). Included by memb_hip.hip only, inside its anonymous
// namespace. What the reference's TrainedCompressor::finalize does per scalar and per word on the host
// (src/trained_compression.cpp:54-71):
//   KMeansClusterizer::predict            src/kmeans.cpp:66-80    lower_bound of every scalar over the split points
//   HuffmanEncoderBuilder::updateFrequencies  src/huffman_encoder.cpp:22-29  symbol histogram
//   HuffmanEncoder::encode + BitStream::push  src/huffman_encoder.cpp:88-97, src/bit_stream.h:18-34
//                                         one MSB-first bitstream per word, zero padded to a whole byte
// The order-dependent part -- the k-means fit on the first 10 000 words (src/kmeans.cpp:26-64, running
// means in data order :92-98) -- and the Huffman tree (a few hundred symbols) stay on the host.
// All three kernels are HBM bound byte shufflers: no MFMA, LDS for the split points / the histogram copies /
// the word being packed.
#pragma once

constexpr uint32_t ENCODER_THREADS = 256;
constexpr uint32_t HISTOGRAM_COPIES = 32;   // one copy of the 256 counters per LDS bank: lanes l and l + 32 share one

struct QuantiseParams {
    const float* values;        // [count] scalars, row-major rows of dim
    uint8_t* symbols;           // [count]
    unsigned long long count;   // multiple of 4 handled by float4 loads, the rest one by one
    const float* splits;        // [splitCount] sorted mid-points between neighbouring centroids
    uint32_t splitCount;        // <= 254
    uint32_t firstStep;         // largest power of two <= max(splitCount, 1)
    unsigned long long* counts; // [256] global histogram, added to
};

// std::lower_bound(splits, splits + n, x) - splits: the number of split points that compare less than x
// (for a NaN none does: symbol 0, as on the host). Branch-free binary search over the LDS copy, which is
// padded with +inf up to 256 entries.
__device__ __forceinline__ uint32_t lowerBound(const float* splitsLds, uint32_t firstStep, float x)
{
    uint32_t index = 0;
    for (uint32_t step = firstStep; step >= 1; step >>= 1) {
        index += splitsLds[index + step - 1] < x ? step : 0u;
    }
    return index;
}

// VEC: one float4 per thread and iteration -- 16 bytes in, 4 symbols (one dword) out, both coalesced; needs
// `values` 16-byte and `symbols` 4-byte aligned (always so when dim is a multiple of 4). Otherwise one
// scalar per thread.
template <bool VEC>
__global__ void quantise_rows(QuantiseParams p)
{
    __shared__ float splitsLds[512];
    __shared__ uint32_t histogram[256 * HISTOGRAM_COPIES];
    for (uint32_t i = threadIdx.x; i < 512; i += blockDim.x) {
        splitsLds[i] = i < p.splitCount ? p.splits[i] : __builtin_inff();
    }
    for (uint32_t i = threadIdx.x; i < 256 * HISTOGRAM_COPIES; i += blockDim.x) {
        histogram[i] = 0;
    }
    __syncthreads();
    const uint32_t copy = threadIdx.x & (HISTOGRAM_COPIES - 1);
    const unsigned long long quads = VEC ? p.count / 4 : 0;
    const unsigned long long stride = static_cast<unsigned long long>(gridDim.x) * blockDim.x;
    for (unsigned long long q = static_cast<unsigned long long>(blockIdx.x) * blockDim.x + threadIdx.x; q < quads; q += stride) {
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p.values) + q);   // read once
        const uint32_t s0 = lowerBound(splitsLds, p.firstStep, v.x);
        const uint32_t s1 = lowerBound(splitsLds, p.firstStep, v.y);
        const uint32_t s2 = lowerBound(splitsLds, p.firstStep, v.z);
        const uint32_t s3 = lowerBound(splitsLds, p.firstStep, v.w);
        reinterpret_cast<uint32_t*>(p.symbols)[q] = s0 | (s1 << 8) | (s2 << 16) | (s3 << 24);
        atomicAdd(&histogram[s0 * HISTOGRAM_COPIES + copy], 1u);
        atomicAdd(&histogram[s1 * HISTOGRAM_COPIES + copy], 1u);
        atomicAdd(&histogram[s2 * HISTOGRAM_COPIES + copy], 1u);
        atomicAdd(&histogram[s3 * HISTOGRAM_COPIES + copy], 1u);
    }
    // VEC: the last one to three scalars; otherwise all of them
    for (unsigned long long i = quads * 4 + static_cast<unsigned long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < p.count;
         i += stride) {
        const uint32_t s = lowerBound(splitsLds, p.firstStep, p.values[i]);
        p.symbols[i] = static_cast<uint8_t>(s);
        atomicAdd(&histogram[s * HISTOGRAM_COPIES + copy], 1u);
    }
    __syncthreads();
    for (uint32_t symbol = threadIdx.x; symbol < 256; symbol += blockDim.x) {
        unsigned long long total = 0;
        for (uint32_t c = 0; c < HISTOGRAM_COPIES; ++c) {
            total += histogram[symbol * HISTOGRAM_COPIES + c];
        }
        if (total) {
            atomicAdd(&p.counts[symbol], total);
        }
    }
}

struct PackParams {
    const uint8_t* symbols;      // [nRows][dim]
    unsigned long long nRows;
    uint32_t dim;
    uint32_t symbolsPerLane;     // ceil(dim / 64)
    const uint32_t* codes;       // [256] code | length << 16
    uint32_t* streamBytes;       // stream_lengths: [nRows] out
    const unsigned long long* streamOffsets;   // pack_streams: [nRows] byte offset of each word's stream
    uint8_t* packed;             // pack_streams: out
    uint32_t slotDwords;         // LDS dwords per wavefront: the longest possible stream, rounded up
};

__device__ __forceinline__ uint32_t waveSum(uint32_t value)
{
#pragma unroll
    for (int offset = 32; offset >= 1; offset >>= 1) {
        value += __shfl_xor(value, offset);
    }
    return value;
}

// Bytes of every word's bitstream: ceil(sum of its symbols' code lengths / 8). One wavefront per word.
__global__ void stream_lengths(PackParams p)
{
    __shared__ uint32_t codesLds[256];
    for (uint32_t i = threadIdx.x; i < 256; i += blockDim.x) {
        codesLds[i] = p.codes[i];
    }
    __syncthreads();
    const uint32_t lane = threadIdx.x & (WAVE - 1);
    const unsigned long long waves = static_cast<unsigned long long>(gridDim.x) * (blockDim.x / WAVE);
    for (unsigned long long row = static_cast<unsigned long long>(blockIdx.x) * (blockDim.x / WAVE) + threadIdx.x / WAVE;
         row < p.nRows; row += waves) {
        const uint8_t* symbols = p.symbols + row * p.dim;
        uint32_t bits = 0;
        for (uint32_t i = lane; i < p.dim; i += WAVE) {
            bits += codesLds[symbols[i]] >> 16;
        }
        bits = waveSum(bits);
        if (lane == 0) {
            p.streamBytes[row] = (bits + 7) / 8;
        }
    }
}

// One wavefront per word: every lane takes symbolsPerLane consecutive symbols, a wave-wide exclusive scan
// of their bit counts gives each lane its bit position, the codes are OR-ed into the word's LDS image
// (big-endian dwords, zeroed first; a code of at most 16 bits touches at most two dwords), and the image
// leaves as consecutive bytes -- the stream starts on a byte of its own in the file, not on a dword.
__global__ void pack_streams(PackParams p)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    uint32_t* codesLds = lds;
    uint32_t* image = lds + 256 + (threadIdx.x / WAVE) * p.slotDwords;
    for (uint32_t i = threadIdx.x; i < 256; i += blockDim.x) {
        codesLds[i] = p.codes[i];
    }
    __syncthreads();
    const uint32_t lane = threadIdx.x & (WAVE - 1);
    const unsigned long long waves = static_cast<unsigned long long>(gridDim.x) * (blockDim.x / WAVE);
    for (unsigned long long row = static_cast<unsigned long long>(blockIdx.x) * (blockDim.x / WAVE) + threadIdx.x / WAVE;
         row < p.nRows; row += waves) {
        const uint8_t* symbols = p.symbols + row * p.dim;
        const uint32_t first = lane * p.symbolsPerLane;
        const uint32_t last = min(p.dim, first + p.symbolsPerLane);
        uint32_t bits = 0;
        for (uint32_t i = first; i < last; ++i) {
            bits += codesLds[symbols[i]] >> 16;
        }
        // exclusive prefix sum over the lanes
        uint32_t position = bits;
#pragma unroll
        for (int offset = 1; offset < WAVE; offset <<= 1) {
            const uint32_t lower = __shfl_up(position, offset);
            position += lane >= static_cast<uint32_t>(offset) ? lower : 0u;
        }
        const uint32_t totalBits = __shfl(position, WAVE - 1);
        position -= bits;
        const uint32_t bytes = (totalBits + 7) / 8;
        for (uint32_t d = lane; d < (bytes + 3) / 4; d += WAVE) {
            image[d] = 0;
        }
        waveLdsFence();
        for (uint32_t i = first; i < last; ++i) {
            const uint32_t entry = codesLds[symbols[i]];
            const uint32_t length = entry >> 16;
            if (length) {
                // the code's `length` bits, MSB first, at bit `position` of the big-endian image
                const unsigned long long window = static_cast<unsigned long long>(entry & 0xffffu) << (64 - length - (position & 31));
                const uint32_t d = position >> 5;
                atomicOr(&image[d], static_cast<uint32_t>(window >> 32));
                if (static_cast<uint32_t>(window)) {
                    atomicOr(&image[d + 1], static_cast<uint32_t>(window));
                }
                position += length;
            }
        }
        waveLdsFence();
        uint8_t* out = p.packed + p.streamOffsets[row];
        for (uint32_t b = lane; b < bytes; b += WAVE) {
            out[b] = static_cast<uint8_t>(image[b >> 2] >> (24 - 8 * (b & 3)));
        }
        waveLdsFence();   // the image is zeroed again for the wavefront's next word
    }
}

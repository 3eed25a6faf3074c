// Thin extern "C" driver around the REFERENCE's own decode-side headers,
// compiled where they lie under /root/reference/src (nothing is copied):
//   huffman_table_decoder.h, bit_stream_reader.h, bit_stream.h, prefix_code.{h,cpp}
// These are std-only. The three standard headers below must come first because
// the reference headers use size_t / numeric_limits / std::max without
// including them. Output goes to oracle/_ref/ (git-ignored); see oracle/Makefile.
//
// TEST INFRASTRUCTURE ONLY: used to validate oracle/memb_oracle.c, to generate
// tests/golden/*.json (tests/golden/make_golden.py) and as the timed CPU
// baseline of bench.py (cpu_baseline.kind = "reference").
#include <cstddef>
#include <limits>
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

#include "huffman_table_decoder.h"
#include "bit_stream.h"

extern "C" {

void* memb_ref_decoder_create(
    const uint8_t* keys, size_t keyCount, const uint32_t* sizeOffsets, size_t sizeOffsetCount, uint32_t maxDirectBits)
{
    return new memb::HuffmanTableDecoder(
        std::vector<uint8_t>(keys, keys + keyCount),
        std::vector<uint32_t>(sizeOffsets, sizeOffsets + sizeOffsetCount),
        maxDirectBits);
}

void memb_ref_decoder_destroy(void* decoder)
{
    delete static_cast<memb::HuffmanTableDecoder*>(decoder);
}

void memb_ref_decode_symbols(
    const void* decoder, const uint8_t* source, size_t sourceSize, size_t count, uint8_t* outKeys)
{
    const auto* tableDecoder = static_cast<const memb::HuffmanTableDecoder*>(decoder);
    auto state = tableDecoder->decode(source, sourceSize);
    for (size_t i = 0; i < count; ++i) {
        outKeys[i] = tableDecoder->next(state);
    }
}

// Rows -> fp32 with the reference's decoder: the loop of
// TrainedCompressedStorage::extract (src/trained_compression.cpp:129-137) --
// state = decode(packed_values + offset, bytes to the end of the array), then
// dim x centroids[next(state)] -- for pre-resolved rows (the word search is not
// part of it; 0xFFFFFFFF and rows past the end give the zero row of
// Reader::wordEmbeddingToBuffer, src/reader.cpp:41-47). The batch is cut into
// `threads` contiguous jobs as Reader::batchEmbeddingToBuffer does (src/reader.cpp:65-84).
void memb_ref_rows_embedding(
    const void* decoder, const uint8_t* packedValues, uint64_t packedValuesSize, const uint32_t* valueOffsets,
    uint64_t wordCount, const float* centroids, const uint32_t* rows, uint64_t count, uint32_t dim, float* out,
    uint64_t ld, uint32_t threads)
{
    const auto* tableDecoder = static_cast<const memb::HuffmanTableDecoder*>(decoder);
    auto job = [=](uint64_t first, uint64_t last) {
        for (uint64_t i = first; i < last; ++i) {
            float* destination = out + i * ld;
            const uint32_t row = rows[i];
            if (row >= wordCount) {
                std::fill(destination, destination + dim, 0.f);
                continue;
            }
            const uint64_t offset = valueOffsets[row];
            auto state = tableDecoder->decode(packedValues + offset, packedValuesSize - offset);
            for (uint32_t k = 0; k < dim; ++k) {
                destination[k] = centroids[tableDecoder->next(state)];
            }
        }
    };
    if (threads <= 1 || count < 1024) {
        job(0, count);
        return;
    }
    const uint64_t jobSize = (count + threads - 1) / threads;
    std::vector<std::thread> pool;
    for (uint64_t first = jobSize; first < count; first += jobSize) {
        pool.emplace_back(job, first, std::min(count, first + jobSize));
    }
    job(0, std::min(count, jobSize));
    for (auto& thread : pool) {
        thread.join();
    }
}

void memb_ref_canonical_codes(
    const uint8_t* keys, const uint32_t* lengths, size_t count, uint16_t* codeByKey, uint32_t* bitsByKey)
{
    std::vector<memb::CodeInfo> codeLengths;
    for (size_t i = 0; i < count; ++i) {
        codeLengths.push_back({keys[i], lengths[i]});
    }
    auto codebook = memb::createCanonicalPrefixCodes(codeLengths);
    for (size_t k = 0; k < 256; ++k) {
        codeByKey[k] = 0;
        bitsByKey[k] = 0;
    }
    for (const auto& item : codebook) {
        codeByKey[item.first] = item.second.code;
        bitsByKey[item.first] = static_cast<uint32_t>(item.second.bitsCount);
    }
}

size_t memb_ref_bitstream_pack(
    const uint16_t* codes, const uint32_t* bits, size_t count, uint8_t* out, size_t capacity)
{
    memb::BitStream stream;
    for (size_t i = 0; i < count; ++i) {
        stream.push(memb::PrefixCode{codes[i], bits[i]});
    }
    auto data = stream.data();
    if (data.size() > capacity) {
        return static_cast<size_t>(-1);
    }
    std::copy(data.begin(), data.end(), out);
    return data.size();
}

}  // extern "C"

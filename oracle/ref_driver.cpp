// Thin extern "C" driver around the REFERENCE's own decode-side headers,
// compiled where they lie under /root/reference/src (nothing is copied):
//   huffman_table_decoder.h, bit_stream_reader.h, bit_stream.h, prefix_code.{h,cpp}
// These are std-only. The three standard headers below must come first because
// the reference headers use size_t / numeric_limits / std::max without
// including them. Output goes to oracle/_ref/ (git-ignored); see oracle/Makefile.
//
// TEST INFRASTRUCTURE ONLY: used to validate oracle/memb_oracle.c and to
// generate tests/golden/*.json (tests/golden/make_golden.py).
#include <cstddef>
#include <limits>
#include <algorithm>
#include <cstdint>
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

// Host-side codec pieces of the product: canonical prefix codes, the lookup
// table the HIP decoder walks, and the encoder half used by the Builder.
// Nothing in this file decodes a vector: decoding happens only on the GPU
// (memb_hip.hip). The CPU restatement used for checking lives in oracle/.
#pragma once

#include <cstddef>
#include <cstdint>
#include <algorithm>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

namespace memb {

struct CodeInfo {
    uint8_t key;
    uint32_t length;
};

struct PrefixCode {
    uint32_t code;
    uint32_t bitsCount;
};

static const uint32_t MAX_CODE_BITS = 16;  // PrefixCode::code is uint16_t in the reference (src/prefix_code.h:10-13)

// Lengths from the serialised decoder: size_offsets[k] = number of symbols
// whose code length is <= k, keys are listed by increasing length
// (producer: reference src/huffman_encoder.cpp:100-117,
//  consumer: reference src/huffman_table_decoder.h:31-37).
inline std::vector<CodeInfo> codeLengthsFromSizeOffsets(
    const uint8_t* keys, size_t keyCount, const uint32_t* sizeOffsets, size_t sizeOffsetCount)
{
    std::vector<CodeInfo> result;
    result.reserve(keyCount);
    size_t currentSize = 0;
    for (size_t keyIndex = 0; keyIndex < keyCount; ++keyIndex) {
        while (currentSize < sizeOffsetCount && keyIndex >= sizeOffsets[currentSize]) {
            ++currentSize;
        }
        if (currentSize >= sizeOffsetCount) {
            throw std::runtime_error("Huffman decoder description is inconsistent");
        }
        result.push_back({keys[keyIndex], static_cast<uint32_t>(currentSize)});
    }
    return result;
}

// Canonical codes for symbols listed by increasing length
// (reference src/prefix_code.cpp:5-22). Indexed like the input.
inline std::vector<PrefixCode> canonicalCodes(const std::vector<CodeInfo>& codeLengths)
{
    std::vector<PrefixCode> codes;
    codes.reserve(codeLengths.size());
    uint32_t code = 0;
    uint32_t bits = 0;
    for (const auto& info : codeLengths) {
        if (info.length < bits) {
            throw std::runtime_error("Huffman code lengths are not sorted");
        }
        code <<= (info.length - bits);
        bits = info.length;
        // an over-subscribed description (lengths that break the Kraft inequality) runs out of
        // codes of this length: the next one would need bits + 1 bits
        if (bits < 32 && (code >> bits) != 0) {
            throw std::runtime_error("Huffman code lengths describe no prefix code");
        }
        codes.push_back({code, bits});
        ++code;
    }
    return codes;
}

// ---------------------------------------------------------------------------
// Lookup table walked by the HIP decoder. Same two-level idea as the
// reference's direct + indirect tables (src/huffman_table_decoder.h:44-91),
// in a layout of its own: one array of 32-bit entries, the first 2^rootBits
// of them indexed by the next rootBits bits of the stream.
//   leaf    : bits 0..7 total code length, bits 8..15 symbol
//   pointer : bit 31 set, bits 0..7 = extra bits to read after the root bits,
//             bits 8..30 = index of the sub-table's first entry
// Sub-table leaves carry the TOTAL code length as well.
// ---------------------------------------------------------------------------

static const uint32_t TABLE_POINTER_FLAG = 0x80000000u;

struct DecodeTable {
    uint32_t rootBits = 0;
    uint32_t maxCodeBits = 0;
    bool hasSubTables = false;
    std::vector<uint32_t> entries;
};

inline uint32_t leafEntry(uint8_t key, uint32_t length)
{
    return length | (static_cast<uint32_t>(key) << 8);
}

inline DecodeTable buildDecodeTable(const std::vector<CodeInfo>& codeLengths, uint32_t rootBitsLimit)
{
    DecodeTable table;
    if (codeLengths.empty()) {
        throw std::runtime_error("Huffman decoder has no symbols");
    }
    for (const auto& info : codeLengths) {
        table.maxCodeBits = std::max(table.maxCodeBits, info.length);
    }
    if (table.maxCodeBits > MAX_CODE_BITS) {
        throw std::runtime_error("Huffman codes longer than 16 bits are not supported");
    }
    auto codes = canonicalCodes(codeLengths);

    uint32_t rootBits = std::max<uint32_t>(1, std::min(table.maxCodeBits, std::max<uint32_t>(1, rootBitsLimit)));
    table.rootBits = rootBits;
    size_t rootSize = size_t(1) << rootBits;
    table.entries.assign(rootSize, leafEntry(codeLengths[0].key, 0));

    // Longest code below every root prefix that needs a sub-table.
    std::vector<uint32_t> prefixMaxBits(rootSize, 0);
    for (size_t i = 0; i < codes.size(); ++i) {
        if (codes[i].bitsCount > rootBits) {
            uint32_t prefix = codes[i].code >> (codes[i].bitsCount - rootBits);
            if (prefix >= rootSize) {
                throw std::runtime_error("Huffman code does not fit its table");
            }
            prefixMaxBits[prefix] = std::max(prefixMaxBits[prefix], codes[i].bitsCount);
        }
    }
    for (size_t prefix = 0; prefix < rootSize; ++prefix) {
        if (prefixMaxBits[prefix]) {
            uint32_t subBits = prefixMaxBits[prefix] - rootBits;
            uint32_t base = static_cast<uint32_t>(table.entries.size());
            table.entries[prefix] = TABLE_POINTER_FLAG | subBits | (base << 8);
            table.entries.resize(table.entries.size() + (size_t(1) << subBits), leafEntry(codeLengths[0].key, 0));
            table.hasSubTables = true;
        }
    }

    for (size_t i = 0; i < codes.size(); ++i) {
        uint32_t bits = codes[i].bitsCount;
        uint32_t entry = leafEntry(codeLengths[i].key, bits);
        if (bits <= rootBits) {
            size_t first = size_t(codes[i].code) << (rootBits - bits);
            size_t count = size_t(1) << (rootBits - bits);
            if (first + count > rootSize) {
                throw std::runtime_error("Huffman code does not fit its table");
            }
            std::fill(table.entries.begin() + first, table.entries.begin() + first + count, entry);
        } else {
            uint32_t prefix = codes[i].code >> (bits - rootBits);
            uint32_t pointer = table.entries[prefix];
            uint32_t subBits = pointer & 0xff;
            size_t base = (pointer & ~TABLE_POINTER_FLAG) >> 8;
            uint32_t suffixBits = bits - rootBits;
            size_t suffix = codes[i].code & ((1u << suffixBits) - 1);
            size_t first = base + (suffix << (subBits - suffixBits));
            size_t count = size_t(1) << (subBits - suffixBits);
            std::fill(table.entries.begin() + first, table.entries.begin() + first + count, entry);
        }
    }
    return table;
}

// ---------------------------------------------------------------------------
// Encoder half (Builder only).
// ---------------------------------------------------------------------------

// Huffman code lengths from symbol counts: repeatedly merge the two lightest
// subtrees (reference src/huffman_encoder.cpp:43-72). Ties are broken by
// (count, creation order), which the reference leaves to its heap; any choice
// yields a valid optimal code and only decoded values are ever compared.
// Returned sorted by (length, key) (reference :79-85 sorts by length only).
inline std::vector<CodeInfo> huffmanCodeLengths(const std::vector<uint64_t>& counts)
{
    struct Node {
        uint64_t count;
        int left;
        int right;
        uint8_t key;
    };
    std::vector<Node> nodes;
    std::vector<int> alive;
    for (size_t key = 0; key < counts.size(); ++key) {
        if (counts[key]) {
            alive.push_back(static_cast<int>(nodes.size()));
            nodes.push_back({counts[key], -1, -1, static_cast<uint8_t>(key)});
        }
    }
    if (alive.empty()) {
        throw std::runtime_error("Nothing to encode");
    }
    auto lighter = [&nodes](int a, int b) {
        if (nodes[a].count != nodes[b].count) {
            return nodes[a].count < nodes[b].count;
        }
        return a < b;
    };
    while (alive.size() > 1) {
        std::sort(alive.begin(), alive.end(), lighter);
        int left = alive[0];
        int right = alive[1];
        alive.erase(alive.begin(), alive.begin() + 2);
        alive.push_back(static_cast<int>(nodes.size()));
        nodes.push_back({nodes[left].count + nodes[right].count, left, right, 0});
    }

    std::vector<CodeInfo> lengths;
    std::vector<std::pair<int, uint32_t>> stack;
    stack.push_back({alive[0], 0});
    while (!stack.empty()) {
        auto item = stack.back();
        stack.pop_back();
        const Node& node = nodes[item.first];
        if (node.left >= 0) {
            stack.push_back({node.left, item.second + 1});
            stack.push_back({node.right, item.second + 1});
        } else {
            lengths.push_back({node.key, item.second});
        }
    }
    std::sort(lengths.begin(), lengths.end(), [](const CodeInfo& a, const CodeInfo& b) {
        return a.length != b.length ? a.length < b.length : a.key < b.key;
    });
    return lengths;
}

// keys + size_offsets as stored in wire::HuffmanDecoder
// (reference src/huffman_encoder.cpp:100-117).
inline void decoderDescription(
    const std::vector<CodeInfo>& codeLengths, std::vector<uint8_t>* keys, std::vector<uint32_t>* sizeOffsets)
{
    size_t currentSize = 0;
    for (size_t i = 0; i < codeLengths.size(); ++i) {
        while (currentSize < codeLengths[i].length) {
            ++currentSize;
            sizeOffsets->push_back(static_cast<uint32_t>(i));
        }
        keys->push_back(codeLengths[i].key);
    }
    sizeOffsets->push_back(static_cast<uint32_t>(codeLengths.size()));
}

// MSB-first bit packer; a word's stream is padded with zero bits to the next
// byte and no extra byte is added (reference src/bit_stream.h:18-34,
// known-answer test src/bit_stream_tests.cpp:31-59).
class BitWriter {
public:
    void push(uint32_t code, uint32_t bitsCount)
    {
        accumulator_ = (accumulator_ << bitsCount) | (code & ((uint64_t(1) << bitsCount) - 1));
        pending_ += bitsCount;
        while (pending_ >= 8) {
            pending_ -= 8;
            bytes_.push_back(static_cast<uint8_t>(accumulator_ >> pending_));
        }
    }

    void flushToByte()
    {
        if (pending_) {
            bytes_.push_back(static_cast<uint8_t>(accumulator_ << (8 - pending_)));
            pending_ = 0;
        }
        accumulator_ = 0;
    }

    std::vector<uint8_t>& bytes() { return bytes_; }

private:
    std::vector<uint8_t> bytes_;
    uint64_t accumulator_ = 0;
    uint32_t pending_ = 0;
};

// 1-D k-means used to train the codebook (reference src/kmeans.cpp:26-112):
// linspace initialisation, 30 Lloyd iterations with a running-mean update in
// data order, clusters smaller than max/128 pruned, one more update; centroids
// kept sorted, assignment by lower_bound over the mid-points.
class KMeansClusterizer {
public:
    explicit KMeansClusterizer(size_t levels): levels_(levels) {}

    void fit(const std::vector<float>& data)
    {
        if (data.empty()) {
            throw std::runtime_error("Nothing to cluster");
        }
        auto minMax = std::minmax_element(data.begin(), data.end());
        float minValue = *minMax.first;
        float maxValue = *minMax.second;

        std::vector<float> centroids;
        for (size_t i = 0; i < levels_; ++i) {
            centroids.push_back(minValue + i / static_cast<float>(levels_ - 1) * (maxValue - minValue));
        }
        setCentroids(centroids);

        std::vector<uint8_t> assignments;
        for (size_t epoch = 0; epoch < MAX_ITERATIONS; ++epoch) {
            predict(data.data(), data.size(), &assignments);
            updateCentroids(data, assignments);
        }

        predict(data.data(), data.size(), &assignments);
        std::vector<size_t> counts(256, 0);
        for (auto a : assignments) {
            counts[a] += 1;
        }
        size_t maxCount = *std::max_element(counts.begin(), counts.end());
        double smallClusterSizeLimit = maxCount / SMALL_CLUSTER_FACTOR;
        std::vector<float> pruned;
        for (size_t i = 0; i < centroids_.size(); ++i) {
            if (counts[i] > smallClusterSizeLimit) {
                pruned.push_back(centroids_[i]);
            }
        }
        setCentroids(pruned);

        predict(data.data(), data.size(), &assignments);
        updateCentroids(data, assignments);
    }

    void predict(const float* data, size_t count, std::vector<uint8_t>* result) const
    {
        result->resize(count);
        predict(data, count, result->data());
    }

    // every value on its own: large inputs are split over threads (the result does not depend on it)
    void predict(const float* data, size_t count, uint8_t* result) const
    {
        if (centroids_.empty()) {
            throw std::runtime_error("Attempt to use KMeansClusterizer before fitting");
        }
        const size_t threads = threadsFor(count);
        const size_t chunk = (count + threads - 1) / threads;
        runThreads(threads, [&](size_t t) {
            const float* first = splits_.data();
            const float* last = first + splits_.size();
            const size_t end = std::min(count, (t + 1) * chunk);
            for (size_t i = std::min(count, t * chunk); i < end; ++i) {
                result[i] = static_cast<uint8_t>(std::lower_bound(first, last, data[i]) - first);
            }
        });
    }

    const std::vector<float>& centroids() const { return centroids_; }
    // the mid-points predict() searches: symbol = number of split points below the value
    const std::vector<float>& splits() const { return splits_; }

private:
    static constexpr size_t MAX_ITERATIONS = 30;
    static constexpr double SMALL_CLUSTER_FACTOR = 128;

    // Running means in data order (reference src/kmeans.cpp:92-98): a cluster's mean depends on the
    // order of its own members only, so threads take whole clusters (cluster a -> thread a mod T) and
    // every mean is the very sequence of fp32 operations the single loop performs.
    void updateCentroids(const std::vector<float>& data, const std::vector<uint8_t>& assignments)
    {
        std::vector<size_t> counts(centroids_.size(), 0);
        std::vector<float> centroids(centroids_.size(), 0);
        const size_t threads = std::min(threadsFor(data.size()), std::max<size_t>(centroids_.size(), 1));
        runThreads(threads, [&](size_t t) {
            // (accumulators of its own: neighbouring clusters share cache lines in the common arrays)
            std::vector<size_t> ownCounts(centroids_.size() + 16, 0);
            std::vector<float> ownCentroids(centroids_.size() + 32, 0);
            for (size_t i = 0; i < data.size(); ++i) {
                const size_t a = assignments[i];
                if (a % threads != t) {
                    continue;
                }
                float n = static_cast<float>(ownCounts[a]);
                ownCentroids[a] = n / (n + 1) * ownCentroids[a] + 1 / (n + 1) * data[i];
                ownCounts[a] += 1;
            }
            for (size_t a = t; a < centroids_.size(); a += threads) {
                centroids[a] = ownCentroids[a];
                counts[a] = ownCounts[a];
            }
        });
        std::sort(centroids.begin(), centroids.end());
        setCentroids(centroids);
    }

    static size_t threadsFor(size_t count)
    {
        return std::max<size_t>(1, std::min<size_t>({std::thread::hardware_concurrency(), size_t(64), count / 65536 + 1}));
    }

    template <typename F>
    static void runThreads(size_t threads, F body)
    {
        if (threads <= 1) {
            body(0);
            return;
        }
        std::vector<std::thread> pool;
        for (size_t t = 1; t < threads; ++t) {
            pool.emplace_back(body, t);
        }
        body(0);
        for (auto& thread : pool) {
            thread.join();
        }
    }

    void setCentroids(const std::vector<float>& centroids)
    {
        centroids_ = centroids;
        splits_.clear();
        for (size_t i = 0; i + 1 < centroids_.size(); ++i) {
            splits_.push_back(static_cast<float>(0.5 * (centroids_[i] + centroids_[i + 1])));
        }
    }

    size_t levels_;
    std::vector<float> centroids_;
    std::vector<float> splits_;
};

}  // namespace memb

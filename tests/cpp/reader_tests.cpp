// The reference's C++ test cases (src/tests.cpp) against this project's memb::Builder / memb::Reader /
// CompressionStrategy, case by case: the same six words, the same three storages at 8 bits within 1 %,
// sorted keys, a missing word -> zeros, the forced two-level decoder (first-level width 1), a 1025-word
// batch serial vs threaded, and the four error cases. No test framework (Boost is not here): CHECK
// counts failures, main() returns how many.
//
//   reader_tests            everything (the lookups need a HIP device)
//   reader_tests --host     only the cases that never touch the device (refusals, and the two known answers
//                           of the writer side: src/bit_stream_tests.cpp, src/kmeans_tests.cpp)
#include "../../memb_amd/csrc/builder.h"
#include "../../memb_amd/csrc/reader.h"
#include "../../memb_amd/csrc/codec.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace {

int failures = 0;

#define CHECK(condition)                                                                  \
    do {                                                                                  \
        if (!(condition)) {                                                               \
            ++failures;                                                                   \
            std::printf("  FAILED %s:%d: %s\n", __FILE__, __LINE__, #condition);          \
        }                                                                                 \
    } while (0)

template <typename Exception, typename Call>
bool throws(Call call)
{
    try {
        call();
    } catch (const Exception&) {
        return true;
    } catch (...) {
        return false;
    }
    return false;
}

struct Sample {
    const char* word;
    std::vector<float> vector;
};

const std::vector<Sample> SAMPLES = {
    {"the", {0.f, 1.f, 2.f}}, {"of", {0.f, -1.f, 2.f}}, {"th", {2.f, 0.f, 1.f}},
    {"a", {1.f, 0.f, -2.f}},  {"tho", {2.f, 0.f, -1.f}}, {"abc", {-2.f, 0.f, 1.f}},
};

const std::string MODEL_FILE = "reader_tests_model.bin";

void writeModel(memb::wire::Storage storage)
{
    memb::Builder builder(3, storage, 8);
    for (const Sample& sample : SAMPLES) {
        builder.addWord(sample.word, sample.vector);
    }
    builder.save(MODEL_FILE);
}

// reference src/tests.cpp:29-59
void roundTrip(const char* name, memb::wire::Storage storage, std::shared_ptr<memb::CompressionStrategy> strategy)
{
    std::printf("%s\n", name);
    writeModel(storage);
    memb::Reader reader(MODEL_FILE, strategy);

    std::vector<std::string> sortedWords;
    for (const Sample& sample : SAMPLES) {
        sortedWords.push_back(sample.word);
    }
    std::sort(sortedWords.begin(), sortedWords.end());
    CHECK(reader.keys() == sortedWords);
    CHECK(reader.dim() == 3);

    for (const Sample& sample : SAMPLES) {
        const std::vector<float> decoded = reader.wordEmbedding(sample.word);
        CHECK(decoded.size() == sample.vector.size());
        for (size_t i = 0; i < decoded.size() && i < sample.vector.size(); ++i) {
            // within 1 % (of the larger magnitude, as BOOST_CHECK_CLOSE_FRACTION measures it)
            const float scale = std::max(std::fabs(decoded[i]), std::fabs(sample.vector[i]));
            CHECK(std::fabs(decoded[i] - sample.vector[i]) <= 0.01f * scale);
        }
    }
    for (float value : reader.wordEmbedding("o")) {
        CHECK(value == 0.0f);
    }
}

// reference src/tests.cpp:90-113
void threadedEqualsSerial()
{
    std::printf("threaded batch equals serial batch\n");
    writeModel(memb::wire::Storage_Trained);
    memb::Reader serial(MODEL_FILE, 1);
    memb::Reader threaded(MODEL_FILE, 4);
    std::vector<std::string> batch;
    for (size_t i = 0; i < 1025; ++i) {
        batch.push_back(SAMPLES[i % SAMPLES.size()].word);
    }
    const std::vector<float> first = serial.batchEmbedding(batch);
    const std::vector<float> second = threaded.batchEmbedding(batch);
    CHECK(first.size() == 1025 * 3);
    CHECK(first.size() == second.size());
    CHECK(first.size() == second.size() && std::memcmp(first.data(), second.data(), first.size() * sizeof(float)) == 0);
    // and row i is the vector of word i % 6, decoded alone
    for (size_t i = 0; i < 1025 && first.size() == 1025 * 3; i += 97) {
        const std::vector<float> alone = serial.wordEmbedding(batch[i]);
        CHECK(std::memcmp(alone.data(), first.data() + 3 * i, 3 * sizeof(float)) == 0);
    }
}

// Beyond the reference's cases: a batch large enough for the word search to run on the device (memb::Reader searches
// host batches of 4096 words and more with resolve_words, include/memb_hip.h: memb_hip_decode_words; on a host reader
// the batch takes the reference's threaded path) equals the same words looked up one at a time, misses included, for
// every storage.
void largeBatchEqualsSingleWords(memb::wire::Storage storage, const char* name)
{
    std::printf("a 5000-word batch equals single words (%s)\n", name);
    writeModel(storage);
    memb::Reader reader(MODEL_FILE, 4);
    std::vector<std::string> batch;
    for (size_t i = 0; i < 5000; ++i) {
        batch.push_back(i % 7 == 3 ? std::string("o") : SAMPLES[i % SAMPLES.size()].word);   // "o": the reference's missing word
    }
    batch[11] = "";
    batch[12] = std::string("the\0suffix", 10);   // a std::string ends at its first NUL for strcmp: "the"
    const std::vector<float> rows = reader.batchEmbedding(batch);
    CHECK(rows.size() == 5000 * 3);
    for (size_t i = 0; i < 5000 && rows.size() == 5000 * 3; i += (i < 20 ? 1 : 53)) {
        const std::vector<float> alone = reader.wordEmbedding(batch[i].c_str());
        CHECK(std::memcmp(alone.data(), rows.data() + 3 * i, 3 * sizeof(float)) == 0);
    }
    CHECK(rows.size() == 5000 * 3 && rows[3 * 3] == 0.f && rows[3 * 3 + 1] == 0.f && rows[3 * 3 + 2] == 0.f);   // "o" -> zeros
}

// reference src/tests.cpp:115-153; none of these needs a device
void refusals()
{
    std::printf("refusals\n");
    {
        memb::Builder builder(15, memb::wire::Storage_Full, 8);
        CHECK(throws<std::runtime_error>([&] { builder.addWord("the", std::vector<float>{0.f, 1.f, 2.f}); }));
    }
    {
        memb::Builder builder(3, memb::wire::Storage_Full, 8);
        builder.addWord("the", std::vector<float>{0.f, 1.f, 2.f});
        CHECK(throws<std::runtime_error>([&] { builder.addWord("the", std::vector<float>{2.f, 1.f, 2.f}); }));
    }
    CHECK(throws<std::exception>([] { memb::Reader reader("missing.bin"); }));
    {
        const std::string invalid = "reader_tests_invalid.bin";
        {
            std::ofstream file(invalid);
            file << "0123456789";
        }
        CHECK(throws<std::runtime_error>([&] { memb::Reader reader(invalid); }));
        std::remove(invalid.c_str());
    }
    // strategies by name and by tag (reference src/compression_strategy.cpp:13-78)
    CHECK((memb::availableCompressionStrategies() == std::vector<std::string>{"full", "uniform", "trained"}));
    CHECK(throws<std::runtime_error>([] { memb::createCompressionStrategy("zip"); }));
    CHECK(memb::createCompressionStrategy("trained")->storageType() == memb::wire::Storage_Trained);
}

// reference src/bit_stream_tests.cpp:29-58 and src/kmeans_tests.cpp:9-37: the writer's two known answers
void codecKnownAnswers()
{
    std::printf("bit packer and k-means known answers\n");
    const std::vector<std::pair<uint32_t, uint32_t>> codes = {{1023, 14}, {33, 6}, {0, 4}, {1234, 11}, {7, 2}};
    memb::BitWriter writer;
    std::string expected;
    for (const auto& code : codes) {
        writer.push(code.first, code.second);
        for (uint32_t bit = code.second; bit-- > 0;) {
            expected.push_back(((code.first >> bit) & 1) ? '1' : '0');
        }
    }
    writer.flushToByte();
    expected.append((8 - expected.size() % 8) % 8, '0');   // 37 bits: three bits of padding
    std::string packed;
    for (uint8_t byte : writer.bytes()) {
        for (int bit = 7; bit >= 0; --bit) {
            packed.push_back(((byte >> bit) & 1) ? '1' : '0');
        }
    }
    CHECK(packed == expected);

    std::vector<float> data;
    std::vector<uint8_t> expectedClusters;
    for (int i = 0; i < 8; ++i) {
        data.push_back(-0.5f + i * 0.125f);
        expectedClusters.push_back(1);
    }
    for (int i = 0; i < 16; ++i) {
        data.push_back(-9.f + i * 0.125f);
        expectedClusters.push_back(0);
    }
    for (int i = 0; i < 4; ++i) {
        data.push_back(11.75f + i * 0.125f);
        expectedClusters.push_back(2);
    }
    memb::KMeansClusterizer clusterizer(3);
    clusterizer.fit(data);
    std::vector<uint8_t> clusters;
    clusterizer.predict(data.data(), data.size(), &clusters);
    CHECK(clusters == expectedClusters);
}

// Descriptions of prefix codes that cannot exist must be refused by the table builder, never indexed
// with (the CPU suite runs this file under AddressSanitizer: an over-subscribed set of lengths used to
// write past the first-level table).
void malformedCodes()
{
    std::printf("malformed code descriptions\n");
    auto lengths = [](std::initializer_list<uint32_t> values) {
        std::vector<memb::CodeInfo> result;
        uint8_t key = 0;
        for (uint32_t length : values) {
            result.push_back({key++, length});
        }
        return result;
    };
    for (uint32_t limit : {1u, 4u, 10u, 12u}) {
        CHECK(throws<std::runtime_error>([&] { memb::buildDecodeTable(lengths({1, 1, 1}), limit); }));
        CHECK(throws<std::runtime_error>([&] { memb::buildDecodeTable(lengths({1, 1, 1, 13}), limit); }));
        CHECK(throws<std::runtime_error>([&] { memb::buildDecodeTable(lengths({1, 2, 3, 3, 3}), limit); }));
        CHECK(throws<std::runtime_error>([&] { memb::buildDecodeTable(lengths({2, 1}), limit); }));        // not sorted
        CHECK(throws<std::runtime_error>([&] { memb::buildDecodeTable(lengths({1, 17}), limit); }));       // too long
        CHECK(throws<std::runtime_error>([&] { memb::buildDecodeTable({}, limit); }));
        // complete and incomplete codes are fine: 1 + 2 + 3 + 3 bits, and a lone 16-bit code beside a 1-bit one
        const memb::DecodeTable complete = memb::buildDecodeTable(lengths({1, 2, 3, 3}), limit);
        CHECK(complete.maxCodeBits == 3 && complete.entries.size() >= (size_t(1) << complete.rootBits));
        const memb::DecodeTable sparse = memb::buildDecodeTable(lengths({1, 16}), limit);
        CHECK(sparse.maxCodeBits == 16 && sparse.hasSubTables);
    }
}

}  // namespace

int main(int argc, char** argv)
{
    const bool hostOnly = argc > 1 && std::strcmp(argv[1], "--host") == 0;
    try {
        refusals();
        codecKnownAnswers();
        malformedCodes();
        if (!hostOnly) {
            roundTrip("full storage round trip", memb::wire::Storage_Full,
                      memb::createCompressionStrategy(memb::wire::Storage_Full));
            roundTrip("uniform storage round trip", memb::wire::Storage_Uniform,
                      memb::createCompressionStrategy(memb::wire::Storage_Uniform));
            roundTrip("trained storage round trip", memb::wire::Storage_Trained,
                      memb::createCompressionStrategy(memb::wire::Storage_Trained));
            // the reference forces its indirect tables with maxDirectDecodeBitLength = 1 (src/tests.cpp:76-88)
            roundTrip("trained storage, first-level table of 1 bit", memb::wire::Storage_Trained,
                      std::make_shared<memb::TrainedCompressionStrategy>(1));
            threadedEqualsSerial();
            largeBatchEqualsSingleWords(memb::wire::Storage_Trained, "trained");
            largeBatchEqualsSingleWords(memb::wire::Storage_Uniform, "uniform");
            largeBatchEqualsSingleWords(memb::wire::Storage_Full, "full");
        }
    } catch (const std::exception& error) {
        std::printf("  FAILED with exception: %s\n", error.what());
        ++failures;
    }
    std::remove(MODEL_FILE.c_str());
    std::printf("%s (%d failed checks)\n", failures ? "FAILED" : "ok", failures);
    return failures ? 1 : 0;
}

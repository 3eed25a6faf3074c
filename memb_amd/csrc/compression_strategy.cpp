#include "compression_strategy.h"
#include "codec.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <future>
#include <map>
#include <stdexcept>
#include <thread>

namespace memb {

namespace {

const std::string INVALID_STRATEGY_PREFIX = "Storage strategy ";  // reference src/compression_strategy.cpp:11
const std::string INVALID_STRATEGY_SUFFIX = " is not supported";

const std::vector<std::shared_ptr<CompressionStrategy>>& compressionStrategies()
{
    // same order as the reference registry (src/compression_strategy.cpp:13-18)
    static const std::vector<std::shared_ptr<CompressionStrategy>> strategies = {
        std::make_shared<FullCompressionStrategy>(),
        std::make_shared<UniformCompressionStrategy>(),
        std::make_shared<TrainedCompressionStrategy>(),
    };
    return strategies;
}

[[noreturn]] void throwDeviceError(const char* what)
{
    throw std::runtime_error(std::string(what) + ": " + memb_hip_last_error());
}

uint8_t quantizationLevelsFor(size_t bitsPerWeight)
{
    // reference src/trained_compression.cpp:29, src/uniform_compression.cpp:7
    size_t levels = bitsPerWeight >= 8 ? 256 : (size_t(1) << bitsPerWeight);
    return static_cast<uint8_t>(std::min<size_t>(levels, 255));
}

}  // namespace

// ---------------------------------------------------------------------------
// CompressedStorage: device context handling shared by all storages
// ---------------------------------------------------------------------------

CompressedStorage::~CompressedStorage()
{
    if (context_) {
        memb_hip_ctx_destroy(context_);
    }
}

void CompressedStorage::setDevice(int device)
{
    std::lock_guard<std::mutex> lock(contextMutex_);
    if (context_ && device != device_) {
        memb_hip_ctx_destroy(context_);
        context_ = nullptr;
        wordsStaged_.store(false, std::memory_order_release);   // (the keys went with the context)
    }
    device_ = device;
}

memb_hip_ctx* CompressedStorage::deviceContext() const
{
    if (onHost()) {
        throw std::runtime_error("this reader decodes on the host (device 'cpu'): it has no HIP device context");
    }
    std::lock_guard<std::mutex> lock(contextMutex_);
    if (!context_) {
        context_ = createDeviceContext(device_);
    }
    return context_;
}

// The reference's batch driver over host threads (src/reader.cpp:59-86): serial below
// THREADED_DECODER_THRESHOLD words or with one thread, else jobs of ceil(n / threads) rows, each
// writing its own slice of the output; a missing row is a zero row (src/reader.cpp:43-46).
void CompressedStorage::decodeRowsHost(const uint32_t* rows, size_t n, float* out, size_t ld, size_t colOff) const
{
    const size_t width = dim();
    const size_t known = rowCount();
    auto work = [=](size_t first, size_t last) {
        for (size_t i = first; i < last; ++i) {
            float* destination = out + i * ld + colOff;
            if (rows[i] < known) {
                extractRowHost(rows[i], destination);
            } else {
                std::fill(destination, destination + width, 0.f);
            }
        }
    };
    hostRows_.fetch_add(n, std::memory_order_relaxed);
    if (n < THREADED_DECODER_THRESHOLD || hostThreads_ <= 1) {
        work(0, n);
        return;
    }
    const size_t jobSize = (n + hostThreads_ - 1) / hostThreads_;
    std::vector<std::future<void>> jobs;
    for (size_t first = jobSize; first < n; first += jobSize) {
        jobs.push_back(std::async(std::launch::async, work, first, std::min(n, first + jobSize)));
    }
    work(0, std::min(n, jobSize));
    for (auto& job : jobs) {
        job.get();
    }
}

void CompressedStorage::decodeRows(const uint32_t* rows, size_t n, float* out, size_t ld, size_t colOff) const
{
    if (onHost() || n <= hostBelow_) {
        decodeRowsHost(rows, n, out, ld, colOff);
        return;
    }
    if (memb_hip_decode_rows(deviceContext(), rows, n, out, ld, colOff) != MEMB_HIP_OK) {
        throwDeviceError("HIP batch lookup failed");
    }
}

void CompressedStorage::decodeRowsDevice(
    const uint32_t* rows, size_t n, float* out, size_t ld, size_t colOff, void* stream, bool accumulate,
    float divisor, bool randomOrder) const
{
    if (onHost()) {
        throw std::runtime_error("this reader decodes on the host (device 'cpu'): device buffers need a reader on a HIP device");
    }
    if (memb_hip_decode_rows_device_ex(
            deviceContext(), rows, n, out, ld, colOff, stream,
            (accumulate ? MEMB_HIP_ACCUMULATE : 0u) | (randomOrder ? MEMB_HIP_ROWS_IN_RANDOM_ORDER : 0u), divisor) !=
        MEMB_HIP_OK) {
        throwDeviceError("HIP batch lookup failed");
    }
}

// ---------------------------------------------------------------------------
// Word -> row on the device
// ---------------------------------------------------------------------------

void CompressedStorage::packedKeys(PackedKeys* keys) const
{
    const size_t count = rowCount();
    keys->ownedOffsets.resize(count);
    size_t total = 0;
    for (size_t row = 0; row < count; ++row) {
        total += std::strlen(key(row)) + 1;
    }
    if (total >= 0xFFFFFFF0ull) {
        throw std::runtime_error("the keys of this model take 4 GiB and more: word search on the device is not available");
    }
    keys->ownedBytes.reserve(total + 1);
    for (size_t row = 0; row < count; ++row) {
        keys->ownedOffsets[row] = static_cast<uint32_t>(keys->ownedBytes.size());
        const char* word = key(row);
        keys->ownedBytes.append(word, std::strlen(word) + 1);
    }
    if (keys->ownedBytes.empty()) {
        keys->ownedBytes.push_back('\0');
    }
    keys->bytes = keys->ownedBytes.data();
    keys->size = keys->ownedBytes.size();
    keys->offsets = keys->ownedOffsets.data();
}

void CompressedStorage::stageWords() const
{
    if (wordsStaged_.load(std::memory_order_acquire)) {
        return;
    }
    memb_hip_ctx* context = deviceContext();
    std::lock_guard<std::mutex> lock(stageWordsMutex_);
    if (wordsStaged_.load(std::memory_order_acquire)) {
        return;
    }
    PackedKeys keys;
    packedKeys(&keys);
    if (memb_hip_ctx_stage_words(context, keys.bytes, keys.size, keys.offsets, rowCount()) != MEMB_HIP_OK) {
        throwDeviceError("Cannot stage the keys on the HIP device");
    }
    wordsStaged_.store(true, std::memory_order_release);
}

void CompressedStorage::resolveRowsDevice(const memb_hip_words* batch, uint32_t* rowsDevice, void* stream) const
{
    stageWords();
    if (memb_hip_resolve_rows_device(deviceContext(), batch, rowsDevice, stream) != MEMB_HIP_OK) {
        throwDeviceError("HIP word search failed");
    }
}

void CompressedStorage::resolveRangeDevice(
    const memb_hip_words* batch, size_t firstWord, size_t count, uint32_t* rowsDevice, void* stream) const
{
    stageWords();
    if (memb_hip_resolve_range_device(deviceContext(), batch, firstWord, count, rowsDevice, stream) != MEMB_HIP_OK) {
        throwDeviceError("HIP word search failed");
    }
}

bool CompressedStorage::decodeWords(const memb_hip_words* batch, float* out, size_t ld, size_t colOff) const
{
    stageWords();
    const int code = memb_hip_decode_words(deviceContext(), batch, out, ld, colOff);
    if (code == MEMB_HIP_UNSUPPORTED) {
        return false;
    }
    if (code != MEMB_HIP_OK) {
        throwDeviceError("HIP batch lookup failed");
    }
    return true;
}

// ---------------------------------------------------------------------------
// Hash index over the keys: open addressing, 64-bit slots {hash tag : 32, row : 32},
// FNV-1a over the word's bytes; a hit is confirmed with strcmp, so the answer is
// always the one the binary search would give.
// ---------------------------------------------------------------------------

struct CompressedStorage::WordIndex {
    static constexpr uint64_t EMPTY = ~uint64_t(0);
    std::vector<std::atomic<uint64_t>> slots;
    uint64_t mask = 0;

    static uint64_t hash(const char* word)
    {
        uint64_t h = 1469598103934665603ull;
        for (const unsigned char* c = reinterpret_cast<const unsigned char*>(word); *c; ++c) {
            h = (h ^ *c) * 1099511628211ull;
        }
        return h ^ (h >> 29);
    }

    explicit WordIndex(const CompressedStorage& storage)
    {
        const size_t count = storage.rowCount();
        size_t capacity = 16;
        while (capacity < 2 * count) {
            capacity *= 2;
        }
        mask = capacity - 1;
        slots = std::vector<std::atomic<uint64_t>>(capacity);
        for (auto& slot : slots) {
            slot.store(EMPTY, std::memory_order_relaxed);
        }
        const size_t threads = std::max<size_t>(1, std::min<size_t>(std::thread::hardware_concurrency(), count / 65536 + 1));
        const size_t perThread = (count + threads - 1) / threads;
        auto insertRange = [this, &storage, count](size_t first, size_t last) {
            for (size_t row = first; row < std::min(last, count); ++row) {
                // keys are sorted: a repeated key is its predecessor's neighbour and stays out, so that the index
                // answers with the FIRST of equal keys, as the binary search does
                if (row > 0 && std::strcmp(storage.key(row), storage.key(row - 1)) == 0) {
                    continue;
                }
                const uint64_t h = hash(storage.key(row));
                const uint64_t value = (h & 0xFFFFFFFF00000000ull) | row;
                for (uint64_t at = h & mask;; at = (at + 1) & mask) {
                    uint64_t expected = EMPTY;
                    if (slots[at].compare_exchange_strong(expected, value, std::memory_order_relaxed)) {
                        break;
                    }
                }
            }
        };
        std::vector<std::thread> pool;
        for (size_t t = 1; t < threads; ++t) {
            pool.emplace_back(insertRange, t * perThread, (t + 1) * perThread);
        }
        insertRange(0, perThread);
        for (auto& thread : pool) {
            thread.join();
        }
    }

    bool find(const CompressedStorage& storage, const char* word, uint32_t* row) const
    {
        const uint64_t h = hash(word);
        const uint64_t tag = h & 0xFFFFFFFF00000000ull;
        for (uint64_t at = h & mask;; at = (at + 1) & mask) {
            const uint64_t value = slots[at].load(std::memory_order_relaxed);
            if (value == EMPTY) {
                return false;
            }
            if ((value & 0xFFFFFFFF00000000ull) == tag) {
                const uint32_t candidate = static_cast<uint32_t>(value);
                if (std::strcmp(storage.key(candidate), word) == 0) {
                    *row = candidate;
                    return true;
                }
            }
        }
    }
};

const CompressedStorage::WordIndex* CompressedStorage::wordIndex() const
{
    std::call_once(wordIndexOnce_, [this] {
        wordIndex_ = std::make_shared<WordIndex>(*this);
        wordIndexBuilt_.store(true, std::memory_order_release);
    });
    return wordIndex_.get();
}

void CompressedStorage::resolveMany(const char* const* words, size_t count, uint32_t* rows, bool useIndex) const
{
    const WordIndex* index = useIndex ? wordIndex() : nullptr;
    for (size_t i = 0; i < count; ++i) {
        uint32_t row = MEMB_HIP_MISSING_ROW;
        bool found = index ? index->find(*this, words[i], &row) : resolve(words[i], &row);
        rows[i] = found ? row : MEMB_HIP_MISSING_ROW;
    }
}

bool CompressedStorage::extract(const std::string& word, float* destination) const
{
    uint32_t row = MEMB_HIP_MISSING_ROW;
    if (!resolve(word.c_str(), &row)) {
        return false;
    }
    decodeRows(&row, 1, destination, dim(), 0);
    return true;
}

// ---------------------------------------------------------------------------
// Registry (reference src/compression_strategy.cpp:24-78)
// ---------------------------------------------------------------------------

std::shared_ptr<CompressionStrategy> createCompressionStrategy(wire::Storage storage)
{
    for (const auto& strategy : compressionStrategies()) {
        if (strategy->storageType() == storage) {
            return strategy;
        }
    }
    throw std::runtime_error(INVALID_STRATEGY_PREFIX + std::to_string(storage) + INVALID_STRATEGY_SUFFIX);
}

std::shared_ptr<CompressionStrategy> createCompressionStrategy(const std::string& name)
{
    for (const auto& strategy : compressionStrategies()) {
        if (strategy->storageName() == name) {
            return strategy;
        }
    }
    throw std::runtime_error(INVALID_STRATEGY_PREFIX + name + INVALID_STRATEGY_SUFFIX);
}

std::vector<std::string> availableCompressionStrategies()
{
    std::vector<std::string> result;
    for (const auto& strategy : compressionStrategies()) {
        result.push_back(strategy->storageName());
    }
    return result;
}

// ---------------------------------------------------------------------------
// trained
// ---------------------------------------------------------------------------

namespace {

class TrainedCompressedStorage : public CompressedStorage {
public:
    TrainedCompressedStorage(const wire::TableView& flatStorage, size_t dim, size_t maxDirectDecodeBitLength):
        dim_(dim),
        maxDirectDecodeBitLength_(maxDirectDecodeBitLength)
    {
        wordOffsets_ = flatStorage.vector<uint32_t>(wire::field::Trained_word_offsets);
        valueOffsets_ = flatStorage.vector<uint32_t>(wire::field::Trained_value_offsets);
        packedWords_ = flatStorage.string(wire::field::Trained_packed_words);
        packedValues_ = flatStorage.vector<uint8_t>(wire::field::Trained_packed_values);
        wire::TableView decoder = flatStorage.table(wire::field::Trained_decoder);
        keys_ = decoder.vector<uint8_t>(wire::field::HuffmanDecoder_keys);
        sizeOffsets_ = decoder.vector<uint32_t>(wire::field::HuffmanDecoder_size_offsets);
        wire::TableView clusterizer = flatStorage.table(wire::field::Trained_clusterizer);
        centroids_ = clusterizer.vector<float>(wire::field::KMeansClusterizer_centroids);

        if (valueOffsets_.size != wordOffsets_.size) {
            throw std::runtime_error(wire::VERIFICATION_FAILED);
        }
        // every word must start inside packed_words, which ends with a NUL
        for (uint32_t offset : wordOffsets_) {
            if (offset > packedWords_.size) {
                throw std::runtime_error(wire::VERIFICATION_FAILED);
            }
        }
    }

    size_t dim() const override { return dim_; }
    size_t rowCount() const override { return wordOffsets_.size; }
    const char* key(size_t index) const override { return packedWords_.data + wordOffsets_[index]; }

    // reference src/trained_compression.cpp:115-125: lower_bound with strcmp
    // over the NUL separated sorted words. The reference dereferences the
    // result even when it is end(); here that is a miss.
    bool resolve(const char* word, uint32_t* row) const override
    {
        const char* words = packedWords_.data;
        const uint32_t* first = wordOffsets_.begin();
        const uint32_t* last = wordOffsets_.end();
        const uint32_t* it = std::lower_bound(
            first, last, word, [words](uint32_t offset, const char* w) { return std::strcmp(words + offset, w) < 0; });
        if (it == last || std::strcmp(words + *it, word) != 0) {
            return false;
        }
        *row = static_cast<uint32_t>(it - first);
        return true;
    }

    std::vector<std::string> keys() const override
    {
        std::vector<std::string> result;
        result.reserve(wordOffsets_.size);
        for (uint32_t offset : wordOffsets_) {
            result.emplace_back(packedWords_.data + offset);
        }
        return result;
    }

protected:
    // the file's packed_words (the string's own terminator included) and word_offsets, as they are
    void packedKeys(PackedKeys* keys) const override
    {
        keys->bytes = packedWords_.data;
        keys->size = packedWords_.size + 1;
        keys->offsets = wordOffsets_.data;
    }

    // Host decode of one row: the canonical-Huffman walk of the reference's extract
    // (src/trained_compression.cpp:126-135: dim x next() -> centroid) over this project's two-level
    // table (codec.h; results do not depend on the table width, like the reference's L). A 64-bit
    // window is refilled bytewise, MSB first, with zeros past the end of packed_values
    // (src/bit_stream_reader.h:16-31).
    void extractRowHost(uint32_t row, float* destination) const override
    {
        const HostDecoder& decoder = hostDecoder();
        const uint8_t* cursor = packedValues_.data + std::min<size_t>(valueOffsets_[row], packedValues_.size);
        const uint8_t* const end = packedValues_.data + packedValues_.size;
        const uint32_t* entries = decoder.table.entries.data();
        const uint32_t rootBits = decoder.table.rootBits;
        const float* centroids = decoder.centroids.data();
        uint64_t window = 0;
        uint32_t filled = 0;
        for (size_t i = 0; i < dim_; ++i) {
            while (filled <= 56) {
                window |= static_cast<uint64_t>(cursor < end ? *cursor++ : 0) << (56 - filled);
                filled += 8;
            }
            uint32_t entry = entries[window >> (64 - rootBits)];
            if (entry & TABLE_POINTER_FLAG) {
                const uint32_t subBits = entry & 0xff;
                const size_t base = (entry & ~TABLE_POINTER_FLAG) >> 8;
                entry = entries[base + ((window << rootBits) >> (64 - subBits))];
            }
            const uint32_t length = entry & 0xff;
            window <<= length;
            filled -= length;
            destination[i] = centroids[(entry >> 8) & 0xff];
        }
    }

    memb_hip_ctx* createDeviceContext(int device) const override
    {
        memb_hip_trained_desc desc{};
        desc.dim = static_cast<uint32_t>(dim_);
        desc.n_rows = valueOffsets_.size;
        desc.packed_values = packedValues_.data;
        desc.packed_values_bytes = packedValues_.size;
        desc.value_offsets = valueOffsets_.data;
        desc.keys = keys_.data;
        desc.n_keys = static_cast<uint32_t>(keys_.size);
        desc.size_offsets = sizeOffsets_.data;
        desc.n_size_offsets = static_cast<uint32_t>(sizeOffsets_.size);
        desc.centroids = centroids_.data;
        desc.n_centroids = static_cast<uint32_t>(centroids_.size);
        desc.max_direct_bits = static_cast<uint32_t>(maxDirectDecodeBitLength_);
        memb_hip_ctx* context = nullptr;
        if (memb_hip_ctx_create_trained(&context, device, &desc) != MEMB_HIP_OK) {
            throwDeviceError("Cannot stage trained storage on the HIP device");
        }
        return context;
    }

private:
    struct HostDecoder {
        DecodeTable table;
        std::vector<float> centroids;   // 256 entries, so that no symbol indexes past the codebook
    };

    // built on the first host decode (a reader on a device never needs it)
    const HostDecoder& hostDecoder() const
    {
        std::call_once(hostDecoderOnce_, [this] {
            auto lengths = codeLengthsFromSizeOffsets(keys_.data, keys_.size, sizeOffsets_.data, sizeOffsets_.size);
            for (const auto& info : lengths) {
                if (info.key >= centroids_.size) {
                    throw std::runtime_error("Huffman symbol without a centroid");
                }
            }
            // DEFAULT_DECODE_TABLE_BIT_LENGTH of the reference (src/trained_compression.h:11) unless told otherwise
            const uint32_t limit = maxDirectDecodeBitLength_ ? static_cast<uint32_t>(maxDirectDecodeBitLength_) : 10;
            hostDecoder_.table = buildDecodeTable(lengths, std::min<uint32_t>(limit, 12));
            hostDecoder_.centroids.assign(256, 0.f);
            std::copy(centroids_.begin(), centroids_.end(), hostDecoder_.centroids.begin());
        });
        return hostDecoder_;
    }

    mutable std::once_flag hostDecoderOnce_;
    mutable HostDecoder hostDecoder_;
    size_t dim_;
    size_t maxDirectDecodeBitLength_;
    wire::VectorView<uint32_t> wordOffsets_;
    wire::VectorView<uint32_t> valueOffsets_;
    wire::VectorView<char> packedWords_;
    wire::VectorView<uint8_t> packedValues_;
    wire::VectorView<uint8_t> keys_;
    wire::VectorView<uint32_t> sizeOffsets_;
    wire::VectorView<float> centroids_;
};

// reference src/trained_compression.cpp:25-101
class TrainedCompressor : public Compressor {
public:
    TrainedCompressor(wire::BufferBuilder& builder, size_t bitsPerWeight):
        builder_(builder),
        quantizationLevels_(quantizationLevelsFor(bitsPerWeight)),
        clusterizer_(quantizationLevels_)
    {}

    ~TrainedCompressor() override
    {
        if (encoder_) {
            memb_hip_encoder_destroy(encoder_);
        }
    }

    void setDevice(int device) override { device_ = device; }

    void add(const std::string& word, const float* source, size_t dim) override
    {
        refuseAfterFailure();
        words_.push_back(word);
        dim_ = dim;
        values_.insert(values_.end(), source, source + dim);
        if (device_ >= 0) {
            pendingDeviceRows(false);
        }
    }

    void addMany(const std::string* words, const float* matrix, size_t count, size_t dim) override
    {
        refuseAfterFailure();
        if (count == 0) {
            return;
        }
        words_.insert(words_.end(), words, words + count);
        dim_ = dim;
        if (device_ >= 0 && encoder_) {
            // straight from the caller's matrix to the device, behind whatever single words are still waiting
            pendingDeviceRows(true);
            deviceCall(memb_hip_encoder_add_rows(encoder_, matrix, count));
            return;
        }
        values_.insert(values_.end(), matrix, matrix + count * dim);
        if (device_ >= 0) {
            pendingDeviceRows(false);
        }
    }

    wire::BufferBuilder::Ref finalize() override
    {
        refuseAfterFailure();
        const size_t wordCount = words_.size();
        if (wordCount == 0) {
            throw std::runtime_error("Nothing to encode");
        }
        // MEMB_BUILDER_VERBOSE=1: where the time of a save goes, step by step, on stderr
        const bool verbose = std::getenv("MEMB_BUILDER_VERBOSE") && std::getenv("MEMB_BUILDER_VERBOSE")[0] == '1';
        auto clock = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        double stamp = clock();
        auto lap = [&](const char* what) {
            const double now = clock();
            if (verbose) {
                std::fprintf(stderr, "memb builder: %-34s %.3f s\n", what, now - stamp);
            }
            stamp = now;
        };

        std::vector<uint32_t> streamLengths(wordCount);
        std::vector<uint8_t> packedValues;     // host path: the streams; device path: fetched straight into the file buffer
        uint64_t totalBytes = 0;
        std::vector<CodeInfo> codeLengths;
        if (device_ >= 0) {
            if (!encoder_) {
                startEncoder();   // fewer words than the k-means sample: nothing has gone to the device yet
            }
            lap("k-means fit (if not done while adding)");
        }
        if (device_ >= 0) {   // (startEncoder hands models it cannot take to the host path)
            pendingDeviceRows(true);
            // every word must have its row on the device: pack lays streams out by row number, and a word
            // without one would be given its neighbour's stream
            uint64_t deviceRows = 0;
            deviceCall(memb_hip_encoder_rows(encoder_, &deviceRows));
            if (deviceRows != wordCount) {
                failed_ = "memb builder on the HIP device: " + std::to_string(deviceRows) + " rows on the device for " +
                    std::to_string(wordCount) + " words";
                throw std::runtime_error(failed_);
            }
            std::vector<uint64_t> counts(256, 0);
            deviceCall(memb_hip_encoder_counts(encoder_, counts.data()));
            lap("device: last rows + histogram");
            codeLengths = huffmanCodeLengths(counts);
            const std::vector<PrefixCode> codebook = codebookFor(codeLengths);
            std::vector<uint16_t> codes(256, 0);
            std::vector<uint8_t> lengths(256, 0);
            for (size_t symbol = 0; symbol < 256; ++symbol) {
                codes[symbol] = static_cast<uint16_t>(codebook[symbol].code);
                lengths[symbol] = static_cast<uint8_t>(codebook[symbol].bitsCount);
            }
            deviceCall(memb_hip_encoder_pack(encoder_, codes.data(), lengths.data(), streamLengths.data(), &totalBytes));
            lap("device: Huffman code + bit packing");
        } else {
            fitClusterizer();
            lap("k-means fit (10 000-word sample)");

            // quantise everything, count symbol frequencies
            std::vector<uint8_t> quantized(values_.size());
            const size_t threads = std::max<size_t>(1, std::min<size_t>(std::thread::hardware_concurrency(), wordCount / 4096 + 1));
            const size_t wordsPerThread = (wordCount + threads - 1) / threads;
            std::vector<std::vector<uint64_t>> partialCounts(threads, std::vector<uint64_t>(256, 0));
            clusterizer_.predict(values_.data(), values_.size(), quantized.data());   // threaded inside
            lap("predict (every scalar)");
            runParallel(threads, [&](size_t t) {
                size_t first = std::min(wordCount, t * wordsPerThread) * dim_;
                size_t last = std::min(wordCount, (t + 1) * wordsPerThread) * dim_;
                for (size_t i = first; i < last; ++i) {
                    partialCounts[t][quantized[i]] += 1;
                }
            });
            std::vector<float>().swap(values_);
            std::vector<uint64_t> counts(256, 0);
            for (const auto& partial : partialCounts) {
                for (size_t k = 0; k < 256; ++k) {
                    counts[k] += partial[k];
                }
            }
            lap("symbol histogram + free fp32");

            codeLengths = huffmanCodeLengths(counts);
            const std::vector<PrefixCode> codebook = codebookFor(codeLengths);

            // one byte aligned stream per word, concatenated in insertion order
            std::vector<std::vector<uint8_t>> partialStreams(threads);
            runParallel(threads, [&](size_t t) {
                size_t first = std::min(wordCount, t * wordsPerThread);
                size_t last = std::min(wordCount, (t + 1) * wordsPerThread);
                BitWriter writer;
                size_t written = 0;
                for (size_t w = first; w < last; ++w) {
                    const uint8_t* symbols = quantized.data() + w * dim_;
                    for (size_t i = 0; i < dim_; ++i) {
                        const PrefixCode& code = codebook[symbols[i]];
                        writer.push(code.code, code.bitsCount);
                    }
                    writer.flushToByte();
                    streamLengths[w] = static_cast<uint32_t>(writer.bytes().size() - written);
                    written = writer.bytes().size();
                }
                partialStreams[t].swap(writer.bytes());
            });
            std::vector<uint8_t>().swap(quantized);
            lap("Huffman code + bit packing");

            for (const auto& part : partialStreams) {
                totalBytes += part.size();
            }
            if (totalBytes <= 0xFFFFFFFFull) {
                packedValues.reserve(totalBytes);
                for (auto& part : partialStreams) {
                    packedValues.insert(packedValues.end(), part.begin(), part.end());
                    std::vector<uint8_t>().swap(part);
                }
            }
            lap("concatenate streams");
        }
        if (totalBytes > 0xFFFFFFFFull) {
            throw std::runtime_error("Packed values exceed 4 GiB");
        }

        std::vector<uint32_t> insertionOffsets(wordCount);
        {
            uint64_t offset = 0;
            for (size_t w = 0; w < wordCount; ++w) {
                insertionOffsets[w] = static_cast<uint32_t>(offset);
                offset += streamLengths[w];
            }
        }

        // sort words with std::string::operator< (reference :73-79). The words are distinct, so the order is
        // unique: chunks sorted on threads and merged pairwise give what one std::sort gives.
        std::vector<uint32_t> order(wordCount);
        for (size_t w = 0; w < wordCount; ++w) {
            order[w] = static_cast<uint32_t>(w);
        }
        {
            auto before = [this](uint32_t a, uint32_t b) { return words_[a] < words_[b]; };
            size_t chunks = 1;
            while (chunks < 32 && chunks * 2 <= std::max<size_t>(1, std::thread::hardware_concurrency()) &&
                   wordCount / (chunks * 2) >= 16384) {
                chunks *= 2;
            }
            const size_t chunkWords = (wordCount + chunks - 1) / chunks;
            auto bound = [&](size_t chunk) { return order.begin() + std::min(wordCount, chunk * chunkWords); };
            runParallel(chunks, [&](size_t chunk) { std::sort(bound(chunk), bound(chunk + 1), before); });
            for (size_t width = 1; width < chunks; width *= 2) {
                runParallel(chunks / (2 * width), [&](size_t pair) {
                    const size_t first = pair * 2 * width;
                    std::inplace_merge(bound(first), bound(first + width), bound(first + 2 * width), before);
                });
            }
        }

        std::string packedWords;
        std::vector<uint32_t> wordOffsets;
        std::vector<uint32_t> valueOffsets;
        wordOffsets.reserve(wordCount);
        valueOffsets.reserve(wordCount);
        for (uint32_t w : order) {
            wordOffsets.push_back(static_cast<uint32_t>(packedWords.size()));
            valueOffsets.push_back(insertionOffsets[w]);
            packedWords.append(words_[w].c_str(), std::strlen(words_[w].c_str()) + 1);
        }

        lap("sort words + offsets");
        std::vector<uint8_t> decoderKeys;
        std::vector<uint32_t> sizeOffsets;
        decoderDescription(codeLengths, &decoderKeys, &sizeOffsets);

        builder_.reserve(totalBytes + packedWords.size() + 8 * wordCount + 4096);
        auto wordOffsetsRef = builder_.createVector(wordOffsets);
        auto valueOffsetsRef = builder_.createVector(valueOffsets);
        auto packedWordsRef = builder_.createString(packedWords);
        wire::BufferBuilder::Ref packedValuesRef;
        if (device_ >= 0) {
            uint8_t* destination = nullptr;
            packedValuesRef = builder_.createVectorUninitialized<uint8_t>(totalBytes, &destination);
            deviceCall(memb_hip_encoder_fetch(encoder_, destination, totalBytes));
            memb_hip_encoder_destroy(encoder_);
            encoder_ = nullptr;
            lap("device: streams -> file buffer");
        } else {
            packedValuesRef = builder_.createVector(packedValues);
        }

        auto keysRef = builder_.createVector(decoderKeys);
        auto sizeOffsetsRef = builder_.createVector(sizeOffsets);
        builder_.startTable();
        builder_.addOffset(wire::field::HuffmanDecoder_keys, keysRef);
        builder_.addOffset(wire::field::HuffmanDecoder_size_offsets, sizeOffsetsRef);
        auto decoderRef = builder_.endTable();

        auto centroidsRef = builder_.createVector(clusterizer_.centroids());
        builder_.startTable();
        builder_.addOffset(wire::field::KMeansClusterizer_centroids, centroidsRef);
        auto clusterizerRef = builder_.endTable();

        builder_.startTable();
        builder_.addOffset(wire::field::Trained_word_offsets, wordOffsetsRef);
        builder_.addOffset(wire::field::Trained_value_offsets, valueOffsetsRef);
        builder_.addOffset(wire::field::Trained_packed_words, packedWordsRef);
        builder_.addOffset(wire::field::Trained_packed_values, packedValuesRef);
        builder_.addOffset(wire::field::Trained_decoder, decoderRef);
        builder_.addOffset(wire::field::Trained_clusterizer, clusterizerRef);
        auto result = builder_.endTable();
        lap("flatbuffer assembly");
        return result;
    }

private:
    static constexpr size_t CLUSTER_SAMPLE_SIZE = 10000;  // reference src/trained_compression.cpp:21

    // The codebook is trained on the first min(10 000, all) words (reference :44-47). It depends on nothing
    // that comes later, so the device path fits as soon as those words are there; values_ then still holds
    // every row added so far.
    void fitClusterizer()
    {
        const size_t sampleWords = std::min(CLUSTER_SAMPLE_SIZE, words_.size());
        std::vector<float> sample(values_.begin(), values_.begin() + sampleWords * dim_);
        clusterizer_.fit(sample);
    }

    static std::vector<PrefixCode> codebookFor(const std::vector<CodeInfo>& codeLengths)
    {
        auto codes = canonicalCodes(codeLengths);
        std::vector<PrefixCode> codebook(256, PrefixCode{0, 0});
        for (size_t i = 0; i < codeLengths.size(); ++i) {
            if (codes[i].bitsCount > MAX_CODE_BITS) {
                throw std::runtime_error("Huffman codes longer than 16 bits are not supported");
            }
            codebook[codeLengths[i].key] = codes[i];
        }
        return codebook;
    }

    // A device call that fails leaves words registered whose rows never reached the device (the caller may
    // catch the exception and go on: tools/converter does, block by block). From then on the compressor
    // refuses everything, so no file with words and streams out of step can be written.
    void deviceCall(int code)
    {
        if (code != MEMB_HIP_OK) {
            failed_ = std::string("memb builder on the HIP device: ") + memb_hip_last_error();
            throw std::runtime_error(failed_);
        }
    }

    void refuseAfterFailure() const
    {
        if (!failed_.empty()) {
            throw std::runtime_error("memb builder: an earlier device call failed, the builder cannot be used further (" + failed_ + ")");
        }
    }

    void startEncoder()
    {
        fitClusterizer();
        const std::vector<float>& splits = clusterizer_.splits();
        // The device quantiser searches sorted, finite split points. A sample with NaN or infinite weights can
        // leave the fit with others; the host path takes whatever the fit produced, so such a model is written
        // there -- same bytes as a host builder's, which is what `device` promises. Nothing has left for the
        // device at this point: values_ still holds every row.
        for (size_t i = 0; i < splits.size(); ++i) {
            if (!(splits[i] == splits[i]) || (i && splits[i] < splits[i - 1])) {
                device_ = -1;
                return;
            }
        }
        deviceCall(memb_hip_encoder_create(
            &encoder_, device_, static_cast<uint32_t>(dim_), splits.data(), static_cast<uint32_t>(splits.size())));
    }

    // Device path: rows wait in values_ until the k-means sample is complete, then go to the device in blocks
    // (single words are collected into blocks of 16 384; `all` sends whatever is waiting).
    void pendingDeviceRows(bool all)
    {
        if (!encoder_) {
            if (words_.size() < CLUSTER_SAMPLE_SIZE) {
                return;
            }
            startEncoder();
            if (device_ < 0) {
                return;   // (the host path from here on; the rows stay in values_)
            }
            all = true;
        }
        const size_t waiting = dim_ ? values_.size() / dim_ : 0;
        if (waiting == 0 || (!all && waiting < 16384)) {
            return;
        }
        deviceCall(memb_hip_encoder_add_rows(encoder_, values_.data(), waiting));
        values_.clear();
        if (all) {
            std::vector<float>().swap(values_);
        }
    }

private:
    template <typename F>
    static void runParallel(size_t threads, F body)
    {
        if (threads <= 1) {
            body(0);
            return;
        }
        std::vector<std::thread> pool;
        for (size_t t = 0; t < threads; ++t) {
            pool.emplace_back(body, t);
        }
        for (auto& thread : pool) {
            thread.join();
        }
    }

    wire::BufferBuilder& builder_;
    uint8_t quantizationLevels_;
    KMeansClusterizer clusterizer_;
    int device_ = -1;
    memb_hip_encoder* encoder_ = nullptr;
    std::string failed_;          // non-empty: a device call failed (deviceCall)
    size_t dim_ = 0;
    std::vector<std::string> words_;
    std::vector<float> values_;   // host path: every row; device path: rows that have not gone to the device yet
};

}  // namespace

std::shared_ptr<Compressor> TrainedCompressionStrategy::createCompressor(
    wire::BufferBuilder& builder, size_t bitsPerWeight) const
{
    return std::make_shared<TrainedCompressor>(builder, bitsPerWeight);
}

std::shared_ptr<CompressedStorage> TrainedCompressionStrategy::createCompressedStorage(
    const wire::TableView& flatStorage, size_t dim) const
{
    return std::make_shared<TrainedCompressedStorage>(flatStorage, dim, maxDirectDecodeBitLength_);
}

// ---------------------------------------------------------------------------
// uniform and full: vectors of tables sorted by their `word` key
// ---------------------------------------------------------------------------

namespace {

// flatbuffers' LookupByKey is a binary search with strcmp over the sorted
// vector of tables (reference call sites src/uniform_compression.cpp:56,
// src/full_compression.cpp:39).
bool lookupByKey(const std::vector<const char*>& words, const char* word, uint32_t* row)
{
    auto it = std::lower_bound(
        words.begin(), words.end(), word, [](const char* a, const char* b) { return std::strcmp(a, b) < 0; });
    if (it == words.end() || std::strcmp(*it, word) != 0) {
        return false;
    }
    *row = static_cast<uint32_t>(it - words.begin());
    return true;
}

class UniformCompressedStorage : public CompressedStorage {
public:
    UniformCompressedStorage(const wire::TableView& flatStorage, size_t dim):
        dim_(dim)
    {
        auto nodes = flatStorage.vector<uint32_t>(wire::field::Uniform_nodes);
        quantizationLevels_ = flatStorage.scalar<uint8_t>(wire::field::Uniform_quantization_levels, 0);
        words_.reserve(nodes.size);
        rows_.reserve(nodes.size);
        for (size_t i = 0; i < nodes.size; ++i) {
            wire::TableView node = flatStorage.tableAt(nodes, i);
            words_.push_back(node.string(wire::field::UniformQuantizedNode_word).data);
            wire::TableView vector = node.table(wire::field::UniformQuantizedNode_compressed_values);
            auto values = vector.vector<uint8_t>(wire::field::UniformQuantizedVector_values);
            memb_hip_uniform_row row{};
            row.values = values.data;
            row.n_values = static_cast<uint32_t>(values.size);
            row.min_value = vector.scalar<float>(wire::field::UniformQuantizedVector_min_value, 0.f);
            row.max_value = vector.scalar<float>(wire::field::UniformQuantizedVector_max_value, 0.f);
            rows_.push_back(row);
        }
    }

    size_t dim() const override { return dim_; }
    size_t rowCount() const override { return rows_.size(); }
    const char* key(size_t index) const override { return words_[index]; }
    bool resolve(const char* word, uint32_t* row) const override { return lookupByKey(words_, word, row); }

    std::vector<std::string> keys() const override
    {
        return std::vector<std::string>(words_.begin(), words_.end());
    }

protected:
    // reference src/uniform_compression.cpp:64-72: four separately rounded fp32 operations per weight
    // (this file is built with -ffp-contract=off for baseline x86-64, like the reference's -O3 build);
    // rows shorter than dim are zero padded, as the staged device copy is.
    void extractRowHost(uint32_t row, float* destination) const override
    {
        const memb_hip_uniform_row& source = rows_[row];
        const size_t count = std::min<size_t>(source.n_values, dim_);
        const float minValue = source.min_value;
        const float maxValue = source.max_value;
        const uint8_t quantizationLevels = quantizationLevels_;
        for (size_t i = 0; i < count; ++i) {
            const float floatValue = static_cast<float>(source.values[i]);
            destination[i] = minValue + (maxValue - minValue) * floatValue / quantizationLevels;
        }
        std::fill(destination + count, destination + dim_, 0.f);
    }

    memb_hip_ctx* createDeviceContext(int device) const override
    {
        memb_hip_uniform_desc desc{};
        desc.dim = static_cast<uint32_t>(dim_);
        desc.n_rows = rows_.size();
        desc.rows = rows_.data();
        desc.quantization_levels = quantizationLevels_;
        memb_hip_ctx* context = nullptr;
        if (memb_hip_ctx_create_uniform(&context, device, &desc) != MEMB_HIP_OK) {
            throwDeviceError("Cannot stage uniform storage on the HIP device");
        }
        return context;
    }

private:
    size_t dim_;
    uint8_t quantizationLevels_ = 0;
    std::vector<const char*> words_;
    std::vector<memb_hip_uniform_row> rows_;
};

// reference src/uniform_compression.cpp:5-48
class UniformCompressor : public Compressor {
public:
    UniformCompressor(wire::BufferBuilder& builder, size_t bitsPerWeight):
        builder_(builder),
        quantizationLevels_(quantizationLevelsFor(bitsPerWeight))
    {}

    void add(const std::string& word, const float* source, size_t dim) override
    {
        auto minMax = std::minmax_element(source, source + dim);
        float minValue = *minMax.first;
        float maxValue = *minMax.second;
        std::vector<uint8_t> quantized(dim);
        for (size_t i = 0; i < dim; ++i) {
            float scaled = quantizationLevels_ * (source[i] - minValue) / (maxValue - minValue);
            quantized[i] = static_cast<uint8_t>(scaled);
        }
        auto valuesRef = builder_.createVector(quantized);
        builder_.startTable();
        builder_.addScalar<float>(wire::field::UniformQuantizedVector_min_value, minValue);
        builder_.addScalar<float>(wire::field::UniformQuantizedVector_max_value, maxValue);
        builder_.addOffset(wire::field::UniformQuantizedVector_values, valuesRef);
        embeddings_.emplace(word, builder_.endTable());
    }

    wire::BufferBuilder::Ref finalize() override
    {
        std::vector<wire::BufferBuilder::Ref> nodes;  // std::map iterates in key order
        for (const auto& item : embeddings_) {
            auto wordRef = builder_.createString(item.first.c_str(), std::strlen(item.first.c_str()));
            builder_.startTable();
            builder_.addOffset(wire::field::UniformQuantizedNode_word, wordRef);
            builder_.addOffset(wire::field::UniformQuantizedNode_compressed_values, item.second);
            nodes.push_back(builder_.endTable());
        }
        auto nodesRef = builder_.createVectorOfTables(nodes);
        builder_.startTable();
        builder_.addOffset(wire::field::Uniform_nodes, nodesRef);
        builder_.addScalar<uint8_t>(wire::field::Uniform_quantization_levels, quantizationLevels_);
        return builder_.endTable();
    }

private:
    wire::BufferBuilder& builder_;
    uint8_t quantizationLevels_;
    std::map<std::string, wire::BufferBuilder::Ref> embeddings_;
};

class FullCompressedStorage : public CompressedStorage {
public:
    FullCompressedStorage(const wire::TableView& flatStorage, size_t dim):
        dim_(dim)
    {
        auto nodes = flatStorage.vector<uint32_t>(wire::field::Full_nodes);
        words_.reserve(nodes.size);
        rows_.reserve(nodes.size);
        for (size_t i = 0; i < nodes.size; ++i) {
            wire::TableView node = flatStorage.tableAt(nodes, i);
            words_.push_back(node.string(wire::field::FullNode_word).data);
            auto values = node.vector<float>(wire::field::FullNode_values);
            memb_hip_full_row row{};
            row.values = values.data;
            row.n_values = static_cast<uint32_t>(values.size);
            rows_.push_back(row);
        }
    }

    size_t dim() const override { return dim_; }
    size_t rowCount() const override { return rows_.size(); }
    const char* key(size_t index) const override { return words_[index]; }
    bool resolve(const char* word, uint32_t* row) const override { return lookupByKey(words_, word, row); }

    std::vector<std::string> keys() const override
    {
        return std::vector<std::string>(words_.begin(), words_.end());
    }

protected:
    // reference src/full_compression.cpp:40-43
    void extractRowHost(uint32_t row, float* destination) const override
    {
        const memb_hip_full_row& source = rows_[row];
        const size_t count = std::min<size_t>(source.n_values, dim_);
        if (count) {
            std::memcpy(destination, source.values, count * sizeof(float));
        }
        std::fill(destination + count, destination + dim_, 0.f);
    }

    memb_hip_ctx* createDeviceContext(int device) const override
    {
        memb_hip_full_desc desc{};
        desc.dim = static_cast<uint32_t>(dim_);
        desc.n_rows = rows_.size();
        desc.rows = rows_.data();
        memb_hip_ctx* context = nullptr;
        if (memb_hip_ctx_create_full(&context, device, &desc) != MEMB_HIP_OK) {
            throwDeviceError("Cannot stage full storage on the HIP device");
        }
        return context;
    }

private:
    size_t dim_;
    std::vector<const char*> words_;
    std::vector<memb_hip_full_row> rows_;
};

// reference src/full_compression.cpp:5-31
class FullCompressor : public Compressor {
public:
    explicit FullCompressor(wire::BufferBuilder& builder): builder_(builder) {}

    void add(const std::string& word, const float* source, size_t dim) override
    {
        embeddings_.emplace(word, builder_.createVector(source, dim));
    }

    wire::BufferBuilder::Ref finalize() override
    {
        std::vector<wire::BufferBuilder::Ref> nodes;
        for (const auto& item : embeddings_) {
            auto wordRef = builder_.createString(item.first.c_str(), std::strlen(item.first.c_str()));
            builder_.startTable();
            builder_.addOffset(wire::field::FullNode_word, wordRef);
            builder_.addOffset(wire::field::FullNode_values, item.second);
            nodes.push_back(builder_.endTable());
        }
        auto nodesRef = builder_.createVectorOfTables(nodes);
        builder_.startTable();
        builder_.addOffset(wire::field::Full_nodes, nodesRef);
        return builder_.endTable();
    }

private:
    wire::BufferBuilder& builder_;
    std::map<std::string, wire::BufferBuilder::Ref> embeddings_;
};

}  // namespace

std::shared_ptr<Compressor> UniformCompressionStrategy::createCompressor(
    wire::BufferBuilder& builder, size_t bitsPerWeight) const
{
    return std::make_shared<UniformCompressor>(builder, bitsPerWeight);
}

std::shared_ptr<CompressedStorage> UniformCompressionStrategy::createCompressedStorage(
    const wire::TableView& flatStorage, size_t dim) const
{
    return std::make_shared<UniformCompressedStorage>(flatStorage, dim);
}

std::shared_ptr<Compressor> FullCompressionStrategy::createCompressor(
    wire::BufferBuilder& builder, size_t /*bitsPerWeight*/) const
{
    return std::make_shared<FullCompressor>(builder);
}

std::shared_ptr<CompressedStorage> FullCompressionStrategy::createCompressedStorage(
    const wire::TableView& flatStorage, size_t dim) const
{
    return std::make_shared<FullCompressedStorage>(flatStorage, dim);
}

}  // namespace memb

// Strategy plugin interface of the read and write side, mirroring the
// reference's (src/compression_strategy.h:7-43): three abstract classes and a
// registry looked up by wire::Storage tag or by name.
//
// Difference that the GPU path forces: the reference's plugin decodes one word
// at a time on the CPU (CompressedStorage::extract). Here a storage resolves
// words to row ids on the host (`resolve`) and decodes whole batches of rows on
// the device (`decodeRows`, through the C ABI in include/memb_hip.h);
// `extract` is kept with the reference's signature and is a batch of one.
//
// Host decode (the reference's serial path, src/reader.cpp:61-63 and SURVEY 8b "else CPU path") exists
// only as something a caller ASKS for: a storage placed on HOST_DEVICE (Reader(..., device = 'cpu'),
// for hosts without a GPU: BASELINE.json configs[0]) or a `hostBelow` word count under which host
// batches stay on the host (latency of single words). It is never taken on its own account: a storage
// on a HIP device that cannot reach that device throws, it does not fall back.
#pragma once

#include "wire.h"
#include "../../include/memb_hip.h"

#include <atomic>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

namespace memb {

const size_t THREADED_DECODER_THRESHOLD = 1024;  // reference src/reader.cpp:9

class CompressedStorage {
public:
    virtual ~CompressedStorage();

    // reference src/compression_strategy.h:9-10
    bool extract(const std::string& word, float* destination) const;
    virtual std::vector<std::string> keys() const = 0;

    virtual size_t dim() const = 0;
    virtual size_t rowCount() const = 0;

    // Row id (position in sorted key order) of `word`, false if absent.
    virtual bool resolve(const char* word, uint32_t* row) const = 0;

    // The i-th key (sorted order), NUL terminated, valid while the storage lives.
    virtual const char* key(size_t index) const = 0;

    // rows[i] = row id of words[i] or MEMB_HIP_MISSING_ROW. With useIndex the
    // lookups go through a hash index over the keys that is built on first use
    // (thread safe; the per-word binary search of `resolve` costs ~21 dependent
    // cache misses on a 2.2 M-word vocabulary); results are identical.
    void resolveMany(const char* const* words, size_t count, uint32_t* rows, bool useIndex) const;
    bool hasWordIndex() const { return wordIndexBuilt_.load(std::memory_order_acquire); }

    // Decode rows[0..n) into out[i * ld + colOff ..], host buffers.
    void decodeRows(const uint32_t* rows, size_t n, float* out, size_t ld, size_t colOff) const;
    // Same with device buffers, enqueued on `stream` (hipStream_t, may be null).
    // `accumulate` / `divisor`: see memb_hip_decode_rows_device_ex.
    // randomOrder: the hint MEMB_HIP_ROWS_IN_RANDOM_ORDER (launch geometry of very large batches; never the result).
    void decodeRowsDevice(
        const uint32_t* rows, size_t n, float* out, size_t ld, size_t colOff, void* stream,
        bool accumulate = false, float divisor = 0.f, bool randomOrder = false) const;

    // device == HOST_DEVICE: rows are decoded by extractRowHost on host threads
    static constexpr int HOST_DEVICE = -2;
    void setDevice(int device);
    int device() const { return device_; }
    bool onHost() const { return device_ == HOST_DEVICE; }
    // Host batches (decodeRows) of at most `words` words are decoded on the host even though
    // the storage lives on a device; 0 (the default) = never.
    void setHostBelow(size_t words) { hostBelow_ = words; }
    size_t hostBelow() const { return hostBelow_; }
    void setHostThreads(size_t threads) { hostThreads_ = threads ? threads : 1; }
    // rows decoded by the host path so far (tests: a device reader with default settings stays at 0)
    uint64_t hostRowsDecoded() const { return hostRows_.load(std::memory_order_relaxed); }
    // The device context, created (payload staged to HBM) on first use.
    memb_hip_ctx* deviceContext() const;

    // Word -> row on the device (include/memb_hip.h: memb_hip_ctx_stage_words, memb_hip_resolve_rows_device): the keys
    // and a hash table over them go to HBM with the first call (thread safe), a batch of packed query words is then
    // resolved by one kernel and the row ids stay on the device. Same answers as `resolve`.
    void stageWords() const;
    void resolveRowsDevice(const memb_hip_words* batch, uint32_t* rowsDevice, void* stream) const;
    void resolveRangeDevice(const memb_hip_words* batch, size_t firstWord, size_t count, uint32_t* rowsDevice, void* stream) const;
    // Words in, host rows out, both halves on the device (memb_hip_decode_words): false when the batch is too large for
    // one staging slice (the caller then looks the words up on the host and passes row ids).
    bool decodeWords(const memb_hip_words* batch, float* out, size_t ld, size_t colOff) const;

protected:
    virtual memb_hip_ctx* createDeviceContext(int device) const = 0;
    // One row decoded on the host into destination[0 .. dim): what the reference's extract does
    // after its word search (src/trained_compression.cpp:126-135, src/uniform_compression.cpp:58-72,
    // src/full_compression.cpp:40-43).
    virtual void extractRowHost(uint32_t row, float* destination) const = 0;
    // The keys as the device wants them: NUL terminated, key r at bytes + offsets[r], bytes[size - 1] == 0. The default
    // collects them from key(); a trained storage hands out the file's own arrays (packed_words / word_offsets).
    struct PackedKeys {
        const char* bytes = nullptr;
        uint64_t size = 0;
        const uint32_t* offsets = nullptr;
        std::string ownedBytes;
        std::vector<uint32_t> ownedOffsets;
    };
    virtual void packedKeys(PackedKeys* keys) const;

private:
    void decodeRowsHost(const uint32_t* rows, size_t n, float* out, size_t ld, size_t colOff) const;

    struct WordIndex;
    const WordIndex* wordIndex() const;

    int device_ = 0;
    size_t hostBelow_ = 0;
    size_t hostThreads_ = 1;
    mutable std::atomic<uint64_t> hostRows_{0};
    mutable std::mutex contextMutex_;
    mutable memb_hip_ctx* context_ = nullptr;
    mutable std::once_flag wordIndexOnce_;
    mutable std::shared_ptr<WordIndex> wordIndex_;
    mutable std::atomic<bool> wordIndexBuilt_{false};
    mutable std::mutex stageWordsMutex_;
    mutable std::atomic<bool> wordsStaged_{false};
};

class Compressor {
public:
    virtual void add(const std::string& word, const float* source, size_t dim) = 0;
    // Rows of a row-major matrix (rowLength == dim floats per row) for words[0 .. count); same result as
    // `count` calls of add.
    virtual void addMany(const std::string* words, const float* matrix, size_t count, size_t dim)
    {
        for (size_t row = 0; row < count; ++row) {
            add(words[row], matrix + row * dim, dim);
        }
    }
    // HIP device that does the per-scalar and per-word work of finalize (memb_hip_encoder_*); negative =
    // the host, as in the reference. The file has the same bytes either way. Storages without device
    // work ignore it.
    virtual void setDevice(int /*device*/) {}
    // Writes the storage table, returns its handle (the union value of Index.storage).
    virtual wire::BufferBuilder::Ref finalize() = 0;
    virtual ~Compressor() {}
};

class CompressionStrategy {
public:
    virtual std::shared_ptr<Compressor> createCompressor(
        wire::BufferBuilder& builder, size_t bitsPerWeight) const = 0;

    virtual std::shared_ptr<CompressedStorage> createCompressedStorage(
        const wire::TableView& flatStorage, size_t dim) const = 0;

    virtual std::string storageName() const = 0;
    virtual wire::Storage storageType() const = 0;
    virtual ~CompressionStrategy() {}
};

std::shared_ptr<CompressionStrategy> createCompressionStrategy(wire::Storage storage);
std::shared_ptr<CompressionStrategy> createCompressionStrategy(const std::string& storageName);
std::vector<std::string> availableCompressionStrategies();

// trained: k-means codebook + canonical Huffman (reference src/trained_compression.h)
class TrainedCompressionStrategy : public CompressionStrategy {
public:
    // maxDirectDecodeBitLength: see memb_hip_trained_desc::max_direct_bits; 0 = default.
    explicit TrainedCompressionStrategy(size_t maxDirectDecodeBitLength = 0):
        maxDirectDecodeBitLength_(maxDirectDecodeBitLength)
    {}
    std::shared_ptr<Compressor> createCompressor(wire::BufferBuilder& builder, size_t bitsPerWeight) const override;
    std::shared_ptr<CompressedStorage> createCompressedStorage(
        const wire::TableView& flatStorage, size_t dim) const override;
    std::string storageName() const override { return "trained"; }
    wire::Storage storageType() const override { return wire::Storage_Trained; }

private:
    size_t maxDirectDecodeBitLength_;
};

// uniform: per-word min/max + one byte per weight (reference src/uniform_compression.h)
class UniformCompressionStrategy : public CompressionStrategy {
public:
    std::shared_ptr<Compressor> createCompressor(wire::BufferBuilder& builder, size_t bitsPerWeight) const override;
    std::shared_ptr<CompressedStorage> createCompressedStorage(
        const wire::TableView& flatStorage, size_t dim) const override;
    std::string storageName() const override { return "uniform"; }
    wire::Storage storageType() const override { return wire::Storage_Uniform; }
};

// full: raw fp32 (reference src/full_compression.h)
class FullCompressionStrategy : public CompressionStrategy {
public:
    std::shared_ptr<Compressor> createCompressor(wire::BufferBuilder& builder, size_t bitsPerWeight) const override;
    std::shared_ptr<CompressedStorage> createCompressedStorage(
        const wire::TableView& flatStorage, size_t dim) const override;
    std::string storageName() const override { return "full"; }
    wire::Storage storageType() const override { return wire::Storage_Full; }
};

}  // namespace memb

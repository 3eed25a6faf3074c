// memb::Reader -- the reference's public read API (src/reader.h:11-40) over
// the HIP batch-lookup path: same constructors and methods, same semantics
// (caller-owned row-major output, missing words give zero rows).
#pragma once

#include "compression_strategy.h"
#include "worker_pool.h"

#include <atomic>
#include <memory>
#include <string>
#include <vector>

namespace memb {

// Read-only mapping of a whole file (the reference uses
// boost::iostreams::mapped_file_source, src/reader.h:37).
class MappedFile {
public:
    explicit MappedFile(const std::string& filename);
    ~MappedFile();
    MappedFile(const MappedFile&) = delete;
    MappedFile& operator=(const MappedFile&) = delete;

    const uint8_t* data() const { return data_; }
    size_t size() const { return size_; }

private:
    const uint8_t* data_ = nullptr;
    size_t size_ = 0;
};

// A batch of query words on a device (include/memb_hip.h: memb_hip_words): packed by pooled host threads into pinned
// memory, copied by the copy engine; one batch can be resolved by several Readers of that device (ReadersUnion).
class WordBatch {
public:
    explicit WordBatch(int device);
    ~WordBatch();
    WordBatch(const WordBatch&) = delete;
    WordBatch& operator=(const WordBatch&) = delete;

    // words[i]: exactly lengths[i] bytes (or NUL terminated when lengths is null). The std::string form cuts every
    // word at its first NUL, where the reference's strcmp stops. The strings are free again on return.
    void pack(const char* const* words, const uint32_t* lengths, size_t count);
    void pack(const std::vector<std::string>& words);
    // Caller-filled batches (include/memb_hip.h: memb_hip_words_plan): begin hands out the pinned job regions, the
    // caller's threads write the words into them, commit closes the batch.
    memb_hip_words_plan begin(size_t count, size_t bytesPerWord = 0);
    void commit();
    size_t size() const;
    int device() const { return device_; }
    const memb_hip_words* handle() const { return handle_; }

private:
    int device_;
    memb_hip_words* handle_ = nullptr;
};

class Reader {
public:
    // device: a HIP device index; CompressedStorage::HOST_DEVICE = decode on the host (the reference's
    // CPU path, for hosts without a GPU); any other negative value: taken from the environment
    // variable MEMB_HIP_DEVICE (an index, or "cpu"), default 0.
    Reader(const std::string& filename, size_t numThreads = 0, int device = -1);
    Reader(
        const std::string& filename,
        std::shared_ptr<CompressionStrategy> compressionStrategy,
        size_t numThreads = 0,
        int device = -1);

    size_t dim() const;
    std::vector<std::string> keys() const;

    void wordEmbeddingToBuffer(const std::string& word, float* buffer) const;
    void batchEmbeddingToBuffer(const std::vector<std::string>& words, float* buffer) const;

    std::vector<float> wordEmbedding(const std::string& word) const;
    std::vector<float> batchEmbedding(const std::vector<std::string>& words) const;

    // --- additions for callers that keep data on the device or build wider rows ---

    size_t size() const;  // number of words in the file
    int device() const;
    // Host batches of at most `words` words are decoded on the host although the reader lives on a
    // device (single-word latency: no launch, no PCIe); 0 = never, the default (or MEMB_HOST_BELOW).
    void setHostBelow(size_t words);
    size_t hostBelow() const;
    uint64_t hostRowsDecoded() const;   // rows the host path has decoded so far
    std::string storageName() const;

    // Sorted-order row ids of `words` (MEMB_HIP_MISSING_ROW for unknown words),
    // resolved on up to numThreads host threads.
    void resolveRows(const std::vector<std::string>& words, uint32_t* rows) const;
    void resolveRows(const char* const* words, size_t count, uint32_t* rows) const;
    void batchEmbeddingToStridedBuffer(
        const char* const* words, size_t count, float* buffer, size_t ld, size_t colOff) const;

    // batchEmbeddingToBuffer into a wider row-major matrix: row i goes to
    // buffer[i * ld + colOff .. + dim) (ReadersUnion 'concatenate').
    void batchEmbeddingToStridedBuffer(
        const std::vector<std::string>& words, float* buffer, size_t ld, size_t colOff) const;

    // Lookup by row id with host or device buffers (see include/memb_hip.h).
    void rowsToBuffer(const uint32_t* rows, size_t n, float* buffer, size_t ld, size_t colOff) const;
    void rowsToDeviceBuffer(
        const uint32_t* rows, size_t n, float* buffer, size_t ld, size_t colOff, void* stream,
        bool accumulate = false, float divisor = 0.f, bool randomOrder = false) const;

    // Several device-buffer lookups in one kernel launch (include/memb_hip.h: memb_hip_decode_batches_device).
    void batchesToDeviceBuffers(const memb_hip_batch* batches, size_t count, void* stream) const;

    memb_hip_ctx* deviceContext() const;
    bool hasWordIndex() const;   // whether lookups go through the hash index by now

    // The Reader's own word batch, for host-buffer lookups that search on the device (batchEmbeddingToBuffer from
    // DEVICE_SEARCH_THRESHOLD words on): held by one call at a time. `batch` is null when the reader decodes on the host
    // or another call holds the batch -- that caller then searches on the host, as before.
    struct WordBatchLease {
        std::unique_lock<std::mutex> lock;
        WordBatch* batch = nullptr;
    };
    WordBatchLease leaseWordBatch(size_t count) const;
    // Rows of a filled (committed) lease into buffer[i * ld + colOff ..]; false: too large for the device path, the
    // caller falls back to resolveRows + rowsToBuffer.
    bool leasedWordsToBuffer(const WordBatchLease& lease, float* buffer, size_t ld, size_t colOff) const;

    // Word -> row on the device (reference src/trained_compression.cpp:115-125): the keys and a hash
    // table over them are staged with the first call (or by stageWords), a batch is resolved by one kernel enqueued
    // on `stream`, and rowsDevice[i] (device memory, batch.size() entries) = the row resolveRows would give. The row
    // ids never visit the host: hand them to rowsToDeviceBuffer.
    void stageWords() const;
    void resolveRowsToDevice(const WordBatch& batch, uint32_t* rowsDevice, void* stream) const;
    // ... of words [firstWord, firstWord + count) of a batch whose jobs covering them are written (committed or not):
    // lookups of finished jobs overlap the filling of later ones. rowsDevice is the whole batch's array.
    void resolveRangeToDevice(const WordBatch& batch, size_t firstWord, size_t count, uint32_t* rowsDevice, void* stream) const;
    // ... against several readers of one device at once (a ReadersUnion): every word is fetched and hashed once and probed in
    // each reader's table (memb_hip_resolve_range_union_device, four readers per launch). rowsDevice[r] is reader r's array.
    static void resolveRangeToDevice(
        const std::vector<const Reader*>& readers, const WordBatch& batch, size_t firstWord, size_t count,
        const std::vector<uint32_t*>& rowsDevice, void* stream);

private:
    wire::TableView getIndexChecked() const;
    size_t adjustedNumThreads(size_t numThreads) const;
    void init(std::shared_ptr<CompressionStrategy> compressionStrategy, int device);

    // Word search of rows[begin, end) as one batch on the pool (which the caller
    // has locked): start only, the caller waits.
    void startSearch(const char* const* words, size_t count, uint32_t* rows, bool useIndex) const;

    size_t numThreads_;
    // threads for the word search: created with the first large batch, used by one batch at
    // a time (a second concurrent caller falls back to threads of its own)
    // words looked up so far: the hash index is built once this passes the size of a batch that
    // would have built it, so that a stream of small batches gets it too
    mutable std::atomic<size_t> wordsResolved_{0};
    mutable std::mutex poolMutex_;
    mutable std::unique_ptr<WorkerPool> pool_;
    mutable std::mutex wordBatchMutex_;
    mutable std::unique_ptr<WordBatch> wordBatch_;
    MappedFile mappedFile_;
    wire::TableView flatIndex_;
    size_t dim_ = 0;
    std::string storageName_;
    std::shared_ptr<CompressedStorage> compressedStorage_;
};

}  // namespace memb

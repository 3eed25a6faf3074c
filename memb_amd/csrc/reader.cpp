#include "reader.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cerrno>
#include <cstdlib>
#include <cstring>
#include <future>
#include <stdexcept>
#include <thread>

namespace memb {

namespace {

// MEMB_HIP_DEVICE: a HIP device index, or "cpu" / "host" for host decode (CompressedStorage::HOST_DEVICE)
int deviceFromEnvironment()
{
    const char* text = std::getenv("MEMB_HIP_DEVICE");
    if (!text || !*text) {
        return 0;
    }
    if (std::strcmp(text, "cpu") == 0 || std::strcmp(text, "host") == 0) {
        return CompressedStorage::HOST_DEVICE;
    }
    return std::atoi(text);
}

// MEMB_HOST_BELOW: host batches of at most this many words are decoded on the host (default 0 = never)
size_t hostBelowFromEnvironment()
{
    const char* text = std::getenv("MEMB_HOST_BELOW");
    return (text && *text) ? static_cast<size_t>(std::strtoull(text, nullptr, 10)) : 0;
}

}  // namespace

MappedFile::MappedFile(const std::string& filename)
{
    int descriptor = ::open(filename.c_str(), O_RDONLY);
    if (descriptor < 0) {
        throw std::runtime_error("failed opening file: " + filename + ": " + std::strerror(errno));
    }
    struct stat info;
    if (::fstat(descriptor, &info) != 0) {
        int error = errno;
        ::close(descriptor);
        throw std::runtime_error("failed reading file size: " + filename + ": " + std::strerror(error));
    }
    size_ = static_cast<size_t>(info.st_size);
    if (size_ > 0) {
        void* mapping = ::mmap(nullptr, size_, PROT_READ, MAP_PRIVATE, descriptor, 0);
        if (mapping == MAP_FAILED) {
            int error = errno;
            ::close(descriptor);
            throw std::runtime_error("failed mapping file: " + filename + ": " + std::strerror(error));
        }
        data_ = static_cast<const uint8_t*>(mapping);
    }
    ::close(descriptor);
}

MappedFile::~MappedFile()
{
    if (data_) {
        ::munmap(const_cast<uint8_t*>(data_), size_);
    }
}

WordBatch::WordBatch(int device):
    device_(device)
{
    if (memb_hip_words_create(&handle_, device) != MEMB_HIP_OK) {
        throw std::runtime_error(std::string("Cannot create a word batch on the HIP device: ") + memb_hip_last_error());
    }
}

WordBatch::~WordBatch()
{
    memb_hip_words_destroy(handle_);
}

void WordBatch::pack(const char* const* words, const uint32_t* lengths, size_t count)
{
    if (memb_hip_words_pack(handle_, words, lengths, count) != MEMB_HIP_OK) {
        throw std::runtime_error(std::string("Packing words for the HIP device failed: ") + memb_hip_last_error());
    }
}

void WordBatch::pack(const std::vector<std::string>& words)
{
    std::vector<const char*> pointers(words.size());
    std::vector<uint32_t> lengths(words.size());
    for (size_t i = 0; i < words.size(); ++i) {
        pointers[i] = words[i].c_str();
        // (up to the first NUL: the reference searches words[i].c_str() with strcmp, src/trained_compression.cpp:119)
        lengths[i] = static_cast<uint32_t>(std::min<size_t>(::strnlen(pointers[i], words[i].size()), 0x7FFFFFFF));
    }
    pack(pointers.data(), lengths.data(), words.size());
}

memb_hip_words_plan WordBatch::begin(size_t count, size_t bytesPerWord)
{
    memb_hip_words_plan plan;
    if (memb_hip_words_begin(handle_, count, bytesPerWord, &plan) != MEMB_HIP_OK) {
        throw std::runtime_error(std::string("Cannot start a word batch on the HIP device: ") + memb_hip_last_error());
    }
    return plan;
}

void WordBatch::commit()
{
    if (memb_hip_words_commit(handle_) != MEMB_HIP_OK) {
        throw std::runtime_error(std::string("Cannot commit the word batch: ") + memb_hip_last_error());
    }
}

size_t WordBatch::size() const
{
    size_t count = 0;
    memb_hip_words_count(handle_, &count);
    return count;
}

// reference src/reader.cpp:13-29
Reader::Reader(
        const std::string& filename,
        std::shared_ptr<CompressionStrategy> compressionStrategy,
        size_t numThreads,
        int device):
    numThreads_(adjustedNumThreads(numThreads)),
    mappedFile_(filename),
    flatIndex_(getIndexChecked())
{
    init(compressionStrategy, device);
}

Reader::Reader(const std::string& filename, size_t numThreads, int device):
    numThreads_(adjustedNumThreads(numThreads)),
    mappedFile_(filename),
    flatIndex_(getIndexChecked())
{
    auto storageType = static_cast<wire::Storage>(
        flatIndex_.scalar<uint8_t>(wire::field::Index_storage_type, wire::Storage_NONE));
    init(createCompressionStrategy(storageType), device);
}

void Reader::init(std::shared_ptr<CompressionStrategy> compressionStrategy, int device)
{
    dim_ = flatIndex_.scalar<uint32_t>(wire::field::Index_dim, 0);
    storageName_ = compressionStrategy->storageName();
    compressedStorage_ = compressionStrategy->createCompressedStorage(
        flatIndex_.table(wire::field::Index_storage), dim_);
    compressedStorage_->setDevice(
        device >= 0 || device == CompressedStorage::HOST_DEVICE ? device : deviceFromEnvironment());
    compressedStorage_->setHostThreads(numThreads_);
    compressedStorage_->setHostBelow(hostBelowFromEnvironment());
}

void Reader::setHostBelow(size_t words)
{
    compressedStorage_->setHostBelow(words);
}

size_t Reader::hostBelow() const
{
    return compressedStorage_->hostBelow();
}

uint64_t Reader::hostRowsDecoded() const
{
    return compressedStorage_->hostRowsDecoded();
}

size_t Reader::dim() const
{
    return dim_;
}

size_t Reader::size() const
{
    return compressedStorage_->rowCount();
}

int Reader::device() const
{
    return compressedStorage_->device();
}

std::string Reader::storageName() const
{
    return storageName_;
}

std::vector<std::string> Reader::keys() const
{
    return compressedStorage_->keys();
}

bool Reader::hasWordIndex() const
{
    return compressedStorage_->hasWordIndex();
}

memb_hip_ctx* Reader::deviceContext() const
{
    return compressedStorage_->deviceContext();
}

void Reader::batchesToDeviceBuffers(const memb_hip_batch* batches, size_t count, void* stream) const
{
    if (memb_hip_decode_batches_device(compressedStorage_->deviceContext(), batches, count, stream) != MEMB_HIP_OK) {
        throw std::runtime_error(std::string("HIP batch lookup failed: ") + memb_hip_last_error());
    }
}

void Reader::stageWords() const
{
    compressedStorage_->stageWords();
}

void Reader::resolveRowsToDevice(const WordBatch& batch, uint32_t* rowsDevice, void* stream) const
{
    if (batch.device() != compressedStorage_->device()) {
        throw std::runtime_error("the word batch lives on another device than the reader");
    }
    compressedStorage_->resolveRowsDevice(batch.handle(), rowsDevice, stream);
}

void Reader::resolveRangeToDevice(const WordBatch& batch, size_t firstWord, size_t count, uint32_t* rowsDevice, void* stream) const
{
    if (batch.device() != compressedStorage_->device()) {
        throw std::runtime_error("the word batch lives on another device than the reader");
    }
    compressedStorage_->resolveRangeDevice(batch.handle(), firstWord, count, rowsDevice, stream);
}

// reference src/reader.cpp:41-47 (extract, zeros for a word that is not there): a batch of one.
// It goes through the same search as batches, i.e. through the hash index once that exists.
void Reader::wordEmbeddingToBuffer(const std::string& word, float* buffer) const
{
    const char* pointer = word.c_str();
    uint32_t row = MEMB_HIP_MISSING_ROW;
    resolveRows(&pointer, 1, &row);
    compressedStorage_->decodeRows(&row, 1, buffer, dim(), 0);
}

// The search half of the reference's batch driver (src/reader.cpp:59-86): the
// batch is split over host threads as there, but only to find row ids -- the
// decode of the whole batch is one kernel launch. The threads come from a pool
// that lives as long as the Reader (at most 64: the search is a few cache
// misses per word and stops scaling long before the 256 threads of a GPU
// host); jobs are not made smaller than 1024 words.
namespace {

const size_t MIN_JOB_SIZE = 1024;
// Host-buffer batches from here on are searched on the device (round 5: 10 000 words 0.07 against 0.28 ms, 100 000
// words 0.14 against 0.9 ms, 2.2 M words 1.1 against 11 ms; at 1 000 words the host's hash index is the faster one)
const size_t DEVICE_SEARCH_THRESHOLD = 4096;
const size_t INDEX_THRESHOLD = 4096;      // smaller batches do not pay for building the hash index
const size_t MAX_POOL_THREADS = 64;
const size_t OVERLAP_THRESHOLD = 262144;  // batches from here on search and decode at the same time

}  // namespace

void Reader::startSearch(const char* const* words, size_t count, uint32_t* rows, bool useIndex) const
{
    const CompressedStorage* storage = compressedStorage_.get();
    const size_t jobs = std::max<size_t>(1, std::min((pool_->size() + 1) * 4, count / MIN_JOB_SIZE));
    const size_t jobSize = (count + jobs - 1) / jobs;
    pool_->start(jobs, [=](size_t job) {
        const size_t first = std::min(count, job * jobSize);
        storage->resolveMany(words + first, std::min(jobSize, count - first), rows + first, useIndex);
    });
}

void Reader::resolveRows(const char* const* words, size_t count, uint32_t* rows) const
{
    const CompressedStorage* storage = compressedStorage_.get();
    const bool useIndex = count >= INDEX_THRESHOLD || storage->hasWordIndex() ||
        wordsResolved_.fetch_add(count, std::memory_order_relaxed) + count >= INDEX_THRESHOLD;

    // (the reference goes threaded at 1024 words; here a second thread only pays from two jobs' worth on)
    if (count < std::max(THREADED_DECODER_THRESHOLD, 2 * MIN_JOB_SIZE) || numThreads_ == 1) {
        storage->resolveMany(words, count, rows, useIndex);
        return;
    }
    std::unique_lock<std::mutex> poolLock(poolMutex_, std::try_to_lock);
    if (poolLock.owns_lock()) {
        if (!pool_) {
            pool_.reset(new WorkerPool(std::min(numThreads_, MAX_POOL_THREADS) - 1));
        }
        startSearch(words, count, rows, useIndex);
        pool_->help();
        pool_->wait();
        return;
    }
    // the pool is busy with another caller's batch: threads of this call's own, as the reference does
    const size_t jobs = std::max<size_t>(1, std::min(std::min(numThreads_, MAX_POOL_THREADS), count / (2 * MIN_JOB_SIZE)));
    const size_t jobSize = (count + jobs - 1) / jobs;
    std::vector<std::future<void>> results;
    for (size_t startIndex = jobSize; startIndex < count; startIndex += jobSize) {
        const size_t jobCount = std::min(jobSize, count - startIndex);
        results.push_back(std::async(std::launch::async, [=] {
            storage->resolveMany(words + startIndex, jobCount, rows + startIndex, useIndex);
        }));
    }
    storage->resolveMany(words, std::min(jobSize, count), rows, useIndex);
    for (auto& future : results) {
        future.get();
    }
}

void Reader::resolveRows(const std::vector<std::string>& words, uint32_t* rows) const
{
    std::vector<const char*> pointers(words.size());
    for (size_t i = 0; i < words.size(); ++i) {
        pointers[i] = words[i].c_str();
    }
    resolveRows(pointers.data(), pointers.size(), rows);
}

void Reader::resolveRangeToDevice(
    const std::vector<const Reader*>& readers, const WordBatch& batch, size_t firstWord, size_t count,
    const std::vector<uint32_t*>& rowsDevice, void* stream)
{
    if (readers.size() != rowsDevice.size()) {
        throw std::runtime_error("one row-id array per reader is needed");
    }
    for (size_t first = 0; first < readers.size(); first += 4) {
        const size_t group = std::min<size_t>(4, readers.size() - first);
        memb_hip_ctx* contexts[4] = {};
        uint32_t* rows[4] = {};
        for (size_t r = 0; r < group; ++r) {
            const Reader* reader = readers[first + r];
            if (batch.device() != reader->compressedStorage_->device()) {
                throw std::runtime_error("the word batch lives on another device than the reader");
            }
            reader->compressedStorage_->stageWords();
            contexts[r] = reader->compressedStorage_->deviceContext();
            rows[r] = rowsDevice[first + r];
        }
        if (memb_hip_resolve_range_union_device(contexts, group, batch.handle(), firstWord, count, rows, stream) != MEMB_HIP_OK) {
            throw std::runtime_error(std::string("HIP word search failed: ") + memb_hip_last_error());
        }
    }
}

Reader::WordBatchLease Reader::leaseWordBatch(size_t count) const
{
    WordBatchLease lease;
    if (compressedStorage_->onHost() || count < DEVICE_SEARCH_THRESHOLD || count <= compressedStorage_->hostBelow()) {
        return lease;
    }
    lease.lock = std::unique_lock<std::mutex>(wordBatchMutex_, std::try_to_lock);
    if (!lease.lock.owns_lock()) {
        return lease;
    }
    if (!wordBatch_ || wordBatch_->device() != compressedStorage_->device()) {
        wordBatch_.reset(new WordBatch(compressedStorage_->device()));
    }
    lease.batch = wordBatch_.get();
    return lease;
}

bool Reader::leasedWordsToBuffer(const WordBatchLease& lease, float* buffer, size_t ld, size_t colOff) const
{
    return compressedStorage_->decodeWords(lease.batch->handle(), buffer, ld, colOff);
}

void Reader::batchEmbeddingToStridedBuffer(
    const char* const* words, size_t count, float* buffer, size_t ld, size_t colOff) const
{
    if (count == 0) {
        return;
    }
    {
        // word search AND decode on the device: the words are packed into pinned memory by pooled threads
        // (memb_hip_words_pack), looked up by resolve_words and decoded from the row ids that leaves in HBM
        WordBatchLease lease = leaseWordBatch(count);
        if (lease.batch) {
            lease.batch->pack(words, nullptr, count);
            if (leasedWordsToBuffer(lease, buffer, ld, colOff)) {
                return;
            }
        }
    }
    std::vector<uint32_t> rows(count);
    if (count >= OVERLAP_THRESHOLD && numThreads_ > 2) {
        // Large batch: the first quarter is searched, then decoded and copied
        // back while the pool searches the rest.
        std::unique_lock<std::mutex> poolLock(poolMutex_, std::try_to_lock);
        if (poolLock.owns_lock()) {
            if (!pool_) {
                pool_.reset(new WorkerPool(std::min(numThreads_, MAX_POOL_THREADS) - 1));
            }
            const CompressedStorage* storage = compressedStorage_.get();
            const size_t head = count / 4;
            startSearch(words, head, rows.data(), true);
            pool_->help();
            pool_->wait();
            startSearch(words + head, count - head, rows.data() + head, true);
            try {
                storage->decodeRows(rows.data(), head, buffer, ld, colOff);
            } catch (...) {
                try {
                    pool_->wait();   // the jobs write into `rows`
                } catch (...) {
                }
                throw;
            }
            pool_->wait();
            poolLock.unlock();
            storage->decodeRows(rows.data() + head, count - head, buffer + head * ld, ld, colOff);
            return;
        }
    }
    resolveRows(words, count, rows.data());
    compressedStorage_->decodeRows(rows.data(), rows.size(), buffer, ld, colOff);
}

void Reader::batchEmbeddingToStridedBuffer(
    const std::vector<std::string>& words, float* buffer, size_t ld, size_t colOff) const
{
    std::vector<const char*> pointers(words.size());
    for (size_t i = 0; i < words.size(); ++i) {
        pointers[i] = words[i].c_str();
    }
    batchEmbeddingToStridedBuffer(pointers.data(), pointers.size(), buffer, ld, colOff);
}

void Reader::batchEmbeddingToBuffer(const std::vector<std::string>& words, float* buffer) const
{
    batchEmbeddingToStridedBuffer(words, buffer, dim(), 0);
}

void Reader::rowsToBuffer(const uint32_t* rows, size_t n, float* buffer, size_t ld, size_t colOff) const
{
    compressedStorage_->decodeRows(rows, n, buffer, ld, colOff);
}

void Reader::rowsToDeviceBuffer(
    const uint32_t* rows, size_t n, float* buffer, size_t ld, size_t colOff, void* stream, bool accumulate,
    float divisor, bool randomOrder) const
{
    compressedStorage_->decodeRowsDevice(rows, n, buffer, ld, colOff, stream, accumulate, divisor, randomOrder);
}

std::vector<float> Reader::wordEmbedding(const std::string& word) const
{
    std::vector<float> result(dim());
    wordEmbeddingToBuffer(word, result.data());
    return result;
}

std::vector<float> Reader::batchEmbedding(const std::vector<std::string>& words) const
{
    std::vector<float> result(dim() * words.size());
    batchEmbeddingToBuffer(words, result.data());
    return result;
}

// reference src/reader.cpp:104-111
wire::TableView Reader::getIndexChecked() const
{
    wire::Blob blob;
    blob.data = mappedFile_.data();
    blob.size = mappedFile_.size();
    return wire::getIndexChecked(blob);
}

// reference src/reader.cpp:113-120
size_t Reader::adjustedNumThreads(size_t numThreads) const
{
    if (numThreads > 0) {
        return numThreads;
    }
    return std::max(std::thread::hardware_concurrency(), 2u);
}

}  // namespace memb

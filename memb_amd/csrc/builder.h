// Write side of the file format. Interface of the reference's memb::Builder
// (src/builder.h:11-26): words go in one at a time, dump()/save() compress them
// with the storage's Compressor and emit one buffer: Index{dim, storage}.
#pragma once

#include "compression_strategy.h"

#include <iosfwd>
#include <memory>
#include <mutex>
#include <string>
#include <unordered_set>
#include <vector>

namespace memb {

class Builder {
public:
    // device: HIP device that does the bulk of the compression (trained storage: quantisation, symbol
    // histogram and bit packing, memb_hip_encoder_* in include/memb_hip.h); negative = everything on the
    // host, as in the reference. The file has the same bytes either way.
    Builder(size_t dim, wire::Storage storageType, size_t bitsPerWeight, int device = -1);
    Builder(size_t dim, const std::string& storageName, size_t bitsPerWeight, int device = -1);

    // Throws std::runtime_error for a vector of the wrong length and for a repeated word.
    void addWord(const std::string& word, const float* embedding, size_t size);
    void addWord(const std::string& word, const std::vector<float>& embedding);
    // Row i of the row-major matrix (rowLength floats per row) belongs to words[i]; the
    // same checks, word by word, so a failure leaves the words before it added.
    // (Adding and dumping take a lock: the Python binding releases the GIL around addWords, so calls
    // from several Python threads are serialised here; the order of words is then the order of calls.)
    void addWords(const std::vector<std::string>& words, const float* matrix, size_t rowLength);

    size_t dim() const { return dim_; }
    size_t wordCount() const { return seen_.size(); }

    void dump(std::ostream& sink);            // the finished buffer, as bytes
    void save(const std::string& filename);

private:
    void addWordLocked(const std::string& word, const float* embedding, size_t size);
    void attach(const std::shared_ptr<CompressionStrategy>& strategy, size_t bitsPerWeight, int device);

    const size_t dim_;
    wire::Storage storageType_ = wire::Storage_NONE;
    wire::BufferBuilder buffer_;
    std::shared_ptr<Compressor> compressor_;
    std::unordered_set<std::string> seen_;
    std::mutex mutex_;
};

}  // namespace memb

#include "builder.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <ostream>
#include <sstream>
#include <stdexcept>

namespace memb {

void Builder::attach(const std::shared_ptr<CompressionStrategy>& strategy, size_t bitsPerWeight, int device)
{
    storageType_ = strategy->storageType();
    compressor_ = strategy->createCompressor(buffer_, bitsPerWeight);
    compressor_->setDevice(device);
}

Builder::Builder(size_t dim, wire::Storage storageType, size_t bitsPerWeight, int device): dim_(dim)
{
    attach(createCompressionStrategy(storageType), bitsPerWeight, device);
}

Builder::Builder(size_t dim, const std::string& storageName, size_t bitsPerWeight, int device): dim_(dim)
{
    attach(createCompressionStrategy(storageName), bitsPerWeight, device);
}

// The two refusals and their messages are the reference's (src/builder.cpp:12-16, 34-48);
// its tests match on them (src/tests.cpp:115-136).
void Builder::addWord(const std::string& word, const float* embedding, size_t size)
{
    std::lock_guard<std::mutex> lock(mutex_);
    addWordLocked(word, embedding, size);
}

void Builder::addWordLocked(const std::string& word, const float* embedding, size_t size)
{
    if (size != dim_) {
        std::ostringstream message;
        message << "Vector dimension (" << size << ") for word " << word << " doesn't match builder dimension ("
                << dim_ << ")";
        throw std::runtime_error(message.str());
    }
    if (!seen_.insert(word).second) {
        throw std::runtime_error("Attempt to add duplicate word " + word + " to index");
    }
    compressor_->add(word, embedding, dim_);
}

void Builder::addWord(const std::string& word, const std::vector<float>& embedding)
{
    addWord(word, embedding.data(), embedding.size());
}

void Builder::addWords(const std::vector<std::string>& words, const float* matrix, size_t rowLength)
{
    std::lock_guard<std::mutex> lock(mutex_);
    seen_.reserve(seen_.size() + words.size());
    if (rowLength != dim_ && !words.empty()) {
        addWordLocked(words[0], matrix, rowLength);   // throws the reference's message
    }
    // the checks word by word; the rows that pass go to the compressor as one block
    const bool verbose = std::getenv("MEMB_BUILDER_VERBOSE") && std::getenv("MEMB_BUILDER_VERBOSE")[0] == '1';
    auto clock = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double start = clock();
    size_t accepted = 0;
    for (; accepted < words.size(); ++accepted) {
        if (!seen_.insert(words[accepted]).second) {
            break;
        }
    }
    const double checked = clock();
    compressor_->addMany(words.data(), matrix, accepted, dim_);
    if (verbose) {
        std::fprintf(stderr, "memb builder: addWords %zu words: duplicate check %.3f s, compressor %.3f s\n",
                     words.size(), checked - start, clock() - checked);
    }
    if (accepted < words.size()) {
        throw std::runtime_error("Attempt to add duplicate word " + words[accepted] + " to index");
    }
}

// Root table Index{storage_type, storage, dim} behind the "memb" identifier
// (reference src/flatbuffers/embeddings.fbs:7-19, src/builder.cpp:50-61).
void Builder::dump(std::ostream& sink)
{
    std::lock_guard<std::mutex> lock(mutex_);
    const wire::BufferBuilder::Ref storage = compressor_->finalize();
    buffer_.startTable();
    buffer_.addScalar<uint32_t>(wire::field::Index_dim, static_cast<uint32_t>(dim_));
    buffer_.addScalar<uint8_t>(wire::field::Index_storage_type, static_cast<uint8_t>(storageType_));
    buffer_.addOffset(wire::field::Index_storage, storage);
    buffer_.finish(buffer_.endTable(), wire::FILE_IDENTIFIER);
    sink.write(reinterpret_cast<const char*>(buffer_.data()), static_cast<std::streamsize>(buffer_.size()));
}

void Builder::save(const std::string& filename)
{
    std::ofstream file(filename, std::ios::binary | std::ios::trunc);
    if (!file) {
        throw std::runtime_error("failed opening file for writing: " + filename);
    }
    dump(file);
    file.flush();
    if (!file) {
        throw std::runtime_error("failed writing file: " + filename);
    }
}

}  // namespace memb

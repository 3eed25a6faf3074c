#include "builder.h"

#include <fstream>
#include <stdexcept>

namespace memb {

// reference src/builder.cpp:20-32
Builder::Builder(size_t dim, wire::Storage storageType, size_t bitsPerWeight):
    dim_(dim),
    storageType_(storageType),
    compressor_(createCompressionStrategy(storageType)->createCompressor(builder_, bitsPerWeight))
{}

Builder::Builder(size_t dim, const std::string& storageName, size_t bitsPerWeight):
    dim_(dim)
{
    auto compressionStrategy = createCompressionStrategy(storageName);
    storageType_ = compressionStrategy->storageType();
    compressor_ = compressionStrategy->createCompressor(builder_, bitsPerWeight);
}

void Builder::addWord(const std::string& word, const std::vector<float>& embedding)
{
    addWord(word, embedding.data(), embedding.size());
}

// reference src/builder.cpp:34-48 (messages: :12-16)
void Builder::addWord(const std::string& word, const float* embedding, size_t size)
{
    if (size != dim_) {
        throw std::runtime_error(
            "Vector dimension (" + std::to_string(size) + ") for word " + word +
            " doesn't match builder dimension (" + std::to_string(dim_) + ")");
    }
    auto insertionResult = addedWords_.insert(word);
    if (!insertionResult.second) {
        throw std::runtime_error("Attempt to add duplicate word " + word + " to index");
    }
    compressor_->add(word, embedding, dim_);
}

// reference src/builder.cpp:50-61
void Builder::dump(std::ostream& sink)
{
    auto storage = compressor_->finalize();
    builder_.startTable();
    builder_.addScalar<uint32_t>(wire::field::Index_dim, static_cast<uint32_t>(dim_));
    builder_.addScalar<uint8_t>(wire::field::Index_storage_type, static_cast<uint8_t>(storageType_));
    builder_.addOffset(wire::field::Index_storage, storage);
    auto root = builder_.endTable();
    builder_.finish(root, wire::FILE_IDENTIFIER);

    sink.write(reinterpret_cast<const char*>(builder_.data()), static_cast<std::streamsize>(builder_.size()));
}

void Builder::save(const std::string& filename)
{
    std::ofstream f(filename, std::ios::binary);
    if (!f) {
        throw std::runtime_error("failed opening file for writing: " + filename);
    }
    dump(f);
}

}  // namespace memb

// memb::Builder -- the reference's write API (src/builder.h:11-26): collect
// words, compress with the chosen strategy, write one "memb" file.
#pragma once

#include "compression_strategy.h"

#include <ostream>
#include <set>
#include <string>
#include <vector>

namespace memb {

class Builder {
public:
    Builder(size_t dim, wire::Storage storageType, size_t bitsPerWeight);
    Builder(size_t dim, const std::string& storageName, size_t bitsPerWeight);

    void addWord(const std::string& word, const std::vector<float>& embedding);
    void addWord(const std::string& word, const float* embedding, size_t size);
    void dump(std::ostream& sink);
    void save(const std::string& filename);

private:
    size_t dim_;
    wire::Storage storageType_;
    wire::BufferBuilder builder_;
    std::shared_ptr<Compressor> compressor_;
    std::set<std::string> addedWords_;
};

}  // namespace memb

// Minimal FlatBuffers-compatible reader and writer for the "memb" container.
//
// The reference stores everything in one FlatBuffers file (schemas:
// reference src/flatbuffers/embeddings.fbs:7-19, trained_compression.fbs:6-13,
// huffman_decoder.fbs:3-6, kmeans.fbs:3-5, uniform_compression.fbs:3-17,
// full_compression.fbs:3-10) and reads it through flatc-generated accessors.
// Neither flatc nor the flatbuffers headers exist in this environment, so the
// binary layout is followed directly from the FlatBuffers wire specification:
//   buffer  : u32 root offset @0, 4-byte file identifier @4
//   table   : i32 soffset @T -> vtable @ T - soffset
//   vtable  : u16 vtable_bytes, u16 table_bytes, u16 field_offset[id] (0 = absent)
//   offsets : u32, relative to the location that holds them, always forward
//   vector  : u32 count, elements;  string: u32 length, bytes, NUL
// All multi-byte values are little-endian.
#pragma once

#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>
#include <algorithm>

namespace memb {
namespace wire {

// Union tags of `Storage` (reference src/flatbuffers/embeddings.fbs:7-11).
enum Storage : uint8_t {
    Storage_NONE = 0,
    Storage_Full = 1,
    Storage_Uniform = 2,
    Storage_Trained = 3,
};

static const char* const FILE_IDENTIFIER = "memb";  // embeddings.fbs:19
static const char* const VERIFICATION_FAILED = "File format verification failed";  // reference src/reader.cpp:107

// ---------------------------------------------------------------------------
// Reader side. Every access is bounds-checked against the mapped range; the
// reference runs no Verifier (a corrupt file is UB there), here it is the
// same runtime_error the identifier check raises.
// ---------------------------------------------------------------------------

struct Blob {
    const uint8_t* data = nullptr;
    size_t size = 0;

    void require(size_t pos, size_t len) const
    {
        if (pos > size || len > size - pos) {
            throw std::runtime_error(VERIFICATION_FAILED);
        }
    }

    template <typename T>
    T read(size_t pos) const
    {
        require(pos, sizeof(T));
        T value;
        std::memcpy(&value, data + pos, sizeof(T));
        return value;
    }
};

template <typename T>
struct VectorView {
    const T* data = nullptr;
    size_t size = 0;
    size_t position = 0;  // byte position of element 0 inside the blob

    const T* begin() const { return data; }
    const T* end() const { return data + size; }
    const T& operator[](size_t i) const { return data[i]; }
};

class TableView {
public:
    TableView() = default;
    TableView(const Blob& blob, size_t position):
        blob_(blob),
        position_(position)
    {
        int32_t soffset = blob_.read<int32_t>(position_);
        int64_t vtable = static_cast<int64_t>(position_) - soffset;
        if (vtable < 0) {
            throw std::runtime_error(VERIFICATION_FAILED);
        }
        vtable_ = static_cast<size_t>(vtable);
        vtableBytes_ = blob_.read<uint16_t>(vtable_);
        blob_.require(vtable_, vtableBytes_);
        if (vtableBytes_ < 4) {
            throw std::runtime_error(VERIFICATION_FAILED);
        }
    }

    bool valid() const { return blob_.data != nullptr; }

    // Byte position of field `id`, 0 when the field is absent (default value).
    size_t field(size_t id) const
    {
        size_t slot = 4 + 2 * id;
        if (slot + 2 > vtableBytes_) {
            return 0;
        }
        uint16_t offset = blob_.read<uint16_t>(vtable_ + slot);
        return offset ? position_ + offset : 0;
    }

    template <typename T>
    T scalar(size_t id, T defaultValue) const
    {
        size_t pos = field(id);
        return pos ? blob_.read<T>(pos) : defaultValue;
    }

    // Position of the object an offset field points at; 0 when absent.
    size_t indirect(size_t id) const
    {
        size_t pos = field(id);
        if (!pos) {
            return 0;
        }
        size_t target = pos + blob_.read<uint32_t>(pos);
        blob_.require(target, 4);
        return target;
    }

    TableView table(size_t id) const
    {
        size_t target = indirect(id);
        if (!target) {
            throw std::runtime_error(VERIFICATION_FAILED);
        }
        return TableView(blob_, target);
    }

    template <typename T>
    VectorView<T> vector(size_t id) const
    {
        size_t target = indirect(id);
        if (!target) {
            throw std::runtime_error(VERIFICATION_FAILED);
        }
        VectorView<T> view;
        view.size = blob_.read<uint32_t>(target);
        view.position = target + 4;
        blob_.require(view.position, view.size * sizeof(T));
        view.data = reinterpret_cast<const T*>(blob_.data + view.position);
        // elements are read through typed pointers: a crafted offset must not make them misaligned
        if (reinterpret_cast<uintptr_t>(view.data) % alignof(T) != 0) {
            throw std::runtime_error(VERIFICATION_FAILED);
        }
        return view;
    }

    // Strings are vectors of char followed by a NUL that is not counted.
    VectorView<char> string(size_t id) const
    {
        VectorView<char> view = vector<char>(id);
        blob_.require(view.position, view.size + 1);
        // storages keep raw char pointers into the mapping and run strcmp / strlen on them
        if (blob_.data[view.position + view.size] != 0) {
            throw std::runtime_error(VERIFICATION_FAILED);
        }
        return view;
    }

    // Element `index` of a vector of tables.
    TableView tableAt(const VectorView<uint32_t>& offsets, size_t index) const
    {
        size_t pos = offsets.position + 4 * index;
        size_t target = pos + blob_.read<uint32_t>(pos);
        return TableView(blob_, target);
    }

    const Blob& blob() const { return blob_; }

private:
    Blob blob_;
    size_t position_ = 0;
    size_t vtable_ = 0;
    size_t vtableBytes_ = 0;
};

// Field ids (vtable slot = 4 + 2 * id), in schema declaration order.
namespace field {
// table Index { storage: Storage; dim: uint; }  -- a union takes two ids
enum { Index_storage_type = 0, Index_storage = 1, Index_dim = 2 };
// table Trained
enum {
    Trained_word_offsets = 0,
    Trained_value_offsets = 1,
    Trained_packed_words = 2,
    Trained_packed_values = 3,
    Trained_decoder = 4,
    Trained_clusterizer = 5
};
enum { HuffmanDecoder_keys = 0, HuffmanDecoder_size_offsets = 1 };
enum { KMeansClusterizer_centroids = 0 };
enum { Uniform_nodes = 0, Uniform_quantization_levels = 1 };
enum { UniformQuantizedNode_word = 0, UniformQuantizedNode_compressed_values = 1 };
enum { UniformQuantizedVector_min_value = 0, UniformQuantizedVector_max_value = 1, UniformQuantizedVector_values = 2 };
enum { Full_nodes = 0 };
enum { FullNode_word = 0, FullNode_values = 1 };
}  // namespace field

// reference src/reader.cpp:104-111 (size >= 8 and identifier at bytes 4..8)
inline TableView getIndexChecked(const Blob& blob)
{
    if (blob.size < 8 || std::memcmp(blob.data + 4, FILE_IDENTIFIER, 4) != 0) {
        throw std::runtime_error(VERIFICATION_FAILED);
    }
    size_t root = blob.read<uint32_t>(0);
    blob.require(root, 4);
    return TableView(blob, root);
}

// ---------------------------------------------------------------------------
// Writer side: builds the buffer back to front, like FlatBufferBuilder does,
// so children are written before (and end up at higher addresses than) the
// tables that refer to them. Object handles are distances from the END of the
// buffer.
// ---------------------------------------------------------------------------

class BufferBuilder {
public:
    typedef uint32_t Ref;  // distance from buffer end to the start of an object
    static constexpr size_t MAX_BUFFER_SIZE = (size_t(1) << 31) - 1;

    explicit BufferBuilder(size_t initialCapacity = 1024):
        storage_(std::max<size_t>(initialCapacity, 64)),
        used_(0),
        minAlign_(1)
    {}

    size_t size() const { return used_; }
    const uint8_t* data() const { return storage_.data() + storage_.size() - used_; }

    void reserve(size_t extraBytes) { ensure(extraBytes); }

    template <typename T>
    Ref createVector(const T* elements, size_t count)
    {
        preAlign(count * sizeof(T), 4);
        preAlign(count * sizeof(T), sizeof(T));
        pushBytes(elements, count * sizeof(T));
        return pushScalar<uint32_t>(static_cast<uint32_t>(count));
    }

    template <typename T>
    Ref createVector(const std::vector<T>& elements)
    {
        return createVector(elements.data(), elements.size());
    }

    // A vector whose elements the caller fills in afterwards (same bytes as createVector of the same
    // elements). *elements stays valid until the next call that adds to the buffer.
    template <typename T>
    Ref createVectorUninitialized(size_t count, T** elements)
    {
        preAlign(count * sizeof(T), 4);
        preAlign(count * sizeof(T), sizeof(T));
        ensure(count * sizeof(T));
        used_ += count * sizeof(T);
        const size_t position = used_;
        const Ref result = pushScalar<uint32_t>(static_cast<uint32_t>(count));   // (reserved below: no reallocation)
        *elements = reinterpret_cast<T*>(at(position));
        return result;
    }

    Ref createString(const char* chars, size_t length)
    {
        preAlign(length + 1, 4);
        uint8_t zero = 0;
        pushBytes(&zero, 1);
        pushBytes(chars, length);
        return pushScalar<uint32_t>(static_cast<uint32_t>(length));
    }

    Ref createString(const std::string& value) { return createString(value.data(), value.size()); }

    // Vector of offsets to already written tables, in the given order.
    Ref createVectorOfTables(const std::vector<Ref>& tables)
    {
        preAlign(tables.size() * 4, 4);
        for (size_t i = tables.size(); i > 0; --i) {
            pushScalar<uint32_t>(referTo(tables[i - 1]));
        }
        return pushScalar<uint32_t>(static_cast<uint32_t>(tables.size()));
    }

    void startTable()
    {
        fields_.clear();
        tableStart_ = used_;
    }

    // The official FlatBuffers writers (a) leave out a scalar field whose value equals the schema
    // default (all defaults are zero in memb's schemas; readers supply the default) and (b) share
    // one vtable between tables with identical layouts, so that the vtable of a table can lie at a
    // higher address than the table (negative soffset). This writer does neither unless told to
    // mimic them (used by tests of the three parsers).
    static bool& omitDefaults()
    {
        static bool enabled = false;
        return enabled;
    }

    template <typename T>
    void addScalar(size_t id, T value)
    {
        if (omitDefaults() && value == T()) {
            return;
        }
        align(sizeof(T));
        pushScalar<T>(value);
        fields_.push_back({id, used_});
    }

    void addOffset(size_t id, Ref target)
    {
        align(4);
        pushScalar<uint32_t>(referTo(target));
        fields_.push_back({id, used_});
    }

    Ref endTable()
    {
        align(4);
        pushScalar<int32_t>(0);  // soffset, patched below
        size_t tableRef = used_;

        size_t maxId = 0;
        for (const auto& f : fields_) {
            maxId = std::max(maxId, f.id);
        }
        size_t slots = fields_.empty() ? 0 : maxId + 1;
        std::vector<uint16_t> vtable(2 + slots, 0);
        vtable[0] = static_cast<uint16_t>(2 * vtable.size());
        vtable[1] = static_cast<uint16_t>(tableRef - tableStart_);
        for (const auto& f : fields_) {
            vtable[2 + f.id] = static_cast<uint16_t>(tableRef - f.ref);
        }
        pushBytes(vtable.data(), 2 * vtable.size());
        size_t vtableRef = used_;
        if (omitDefaults()) {
            for (const auto& earlier : sharedVtables_) {
                if (earlier.first == vtable) {
                    used_ -= 2 * vtable.size();   // drop the copy just written, point at the earlier one
                    vtableRef = earlier.second;
                    break;
                }
            }
            if (vtableRef == used_) {
                sharedVtables_.push_back({vtable, vtableRef});
            }
        }

        int32_t soffset = static_cast<int32_t>(vtableRef - tableRef);
        std::memcpy(at(tableRef), &soffset, 4);
        return static_cast<Ref>(tableRef);
    }

    void finish(Ref root, const char* identifier)
    {
        preAlign(8, std::max<size_t>(minAlign_, 4));
        pushBytes(identifier, 4);
        align(4);
        pushScalar<uint32_t>(referTo(root));
    }

private:
    struct FieldLocation {
        size_t id;
        size_t ref;
    };
    std::vector<std::pair<std::vector<uint16_t>, size_t>> sharedVtables_;

    uint8_t* at(size_t ref) { return storage_.data() + storage_.size() - ref; }

    void ensure(size_t extra)
    {
        // offsets are 32 bits; like FlatBufferBuilder (FLATBUFFERS_MAX_BUFFER_SIZE) refuse to grow past
        // 2 GiB - 1 instead of writing wrapped offsets
        if (extra > MAX_BUFFER_SIZE || used_ > MAX_BUFFER_SIZE - extra) {
            throw std::runtime_error("memb file would exceed 2 GiB: the format's offsets are 32 bits");
        }
        if (used_ + extra <= storage_.size()) {
            return;
        }
        size_t newSize = std::max(storage_.size() * 2, used_ + extra + 64);
        std::vector<uint8_t> grown(newSize);
        std::memcpy(grown.data() + newSize - used_, storage_.data() + storage_.size() - used_, used_);
        storage_.swap(grown);
    }

    void pushBytes(const void* bytes, size_t length)
    {
        ensure(length);
        used_ += length;
        if (length) {
            std::memcpy(at(used_), bytes, length);
        }
    }

    void pad(size_t count)
    {
        ensure(count);
        used_ += count;
        std::memset(at(used_), 0, count);
    }

    template <typename T>
    Ref pushScalar(T value)
    {
        align(sizeof(T));
        pushBytes(&value, sizeof(T));
        return static_cast<Ref>(used_);
    }

    void align(size_t alignment)
    {
        minAlign_ = std::max(minAlign_, alignment);
        pad((alignment - used_ % alignment) % alignment);
    }

    // Pad so that after `length` more bytes the write position is aligned.
    void preAlign(size_t length, size_t alignment)
    {
        minAlign_ = std::max(minAlign_, alignment);
        pad((alignment - (used_ + length) % alignment) % alignment);
    }

    uint32_t referTo(Ref target)
    {
        align(4);
        return static_cast<uint32_t>(used_ - target + 4);
    }

    std::vector<uint8_t> storage_;
    size_t used_;
    size_t minAlign_;
    std::vector<FieldLocation> fields_;
    size_t tableStart_ = 0;
};

}  // namespace wire
}  // namespace memb

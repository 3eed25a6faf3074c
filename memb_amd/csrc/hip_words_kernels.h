// build_word_table / resolve_words: word -> row on the device.
//
// Device code of libmemb_hip.so (gfx950 / CDNA4). Included by memb_hip.hip only,
// inside its anonymous namespace; see that file for the overview.
//
// What it replaces: the search in front of every extract of the reference --
// std::lower_bound + strcmp over the sorted keys (src/trained_compression.cpp:115-125), flatbuffers'
// LookupByKey (src/uniform_compression.cpp:56, src/full_compression.cpp:39) -- for a whole batch at once.
// The answer is the binary search's answer, found another way: an open-addressing hash table over the
// keys, built on the device when a model's keys are staged, 16-byte slots
//     {hash tag, row, byte offset of the key, length of the key}
// FNV-1a 64 over the bytes (the function of the host index, memb_amd/csrc/compression_strategy.cpp), the
// slot index from the low bits, the tag from the high 32; a slot whose tag and length match is confirmed
// by comparing the key's bytes with the query's, so a tag collision costs a compare and never an answer.
// A random word costs two dependent line requests -- its slot, the key it names -- where 21 steps of a
// binary search over 2.2 M keys are 21.
#pragma once

constexpr unsigned long long FNV_OFFSET_BASIS = 1469598103934665603ull;
constexpr unsigned long long FNV_PRIME = 1099511628211ull;
constexpr uint32_t WORD_SLOT_EMPTY = 0xFFFFFFFFu;   // in the `row` field (a model has fewer than 2^32 - 1 rows)

// x = tag, y = row, z = byte offset of the key inside the staged keys, w = its length in bytes
typedef u32x4 WordSlot;

__device__ __forceinline__ unsigned long long finishWordHash(unsigned long long h)
{
    return h ^ (h >> 29);   // (the multiply leaves the low bits poorly mixed; index = low bits, tag = high 32)
}

struct WordTableParams {
    const uint8_t* keyBytes;        // n keys, NUL terminated; keyBytesTotal bytes, the last one NUL, + 16 zero bytes
    const uint32_t* keyOffsets;     // [n]
    unsigned long long n;
    unsigned long long keyBytesTotal;
    WordSlot* slots;                // slotMask + 1 of them, all bytes 0xFF before the build
    uint32_t slotMask;
    uint32_t* inserted;             // counts the keys that went in (repeated keys do not)
};

// One lane per key. Keys are sorted, so neighbouring lanes read neighbouring bytes and a repeated key is
// the neighbour of its first occurrence: row r is left out when key r equals key r - 1, which makes the
// table answer with the FIRST of equal keys, as lower_bound does. A slot is claimed by a 64-bit
// compare-and-swap on {tag, row}; offset and length follow as a plain store (nothing reads the table
// before this kernel has finished).
__global__ void build_word_table(WordTableParams p)
{
    const unsigned long long row = static_cast<unsigned long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (row >= p.n) {
        return;
    }
    const uint32_t offset = p.keyOffsets[row];
    const uint8_t* key = p.keyBytes + offset;
    unsigned long long h = FNV_OFFSET_BASIS;
    uint32_t length = 0;
    while (offset + static_cast<unsigned long long>(length) < p.keyBytesTotal) {
        const uint32_t c = key[length];
        if (c == 0) {
            break;
        }
        h = (h ^ c) * FNV_PRIME;
        ++length;
    }
    bool insert = true;
    if (row > 0) {
        const uint32_t previousOffset = p.keyOffsets[row - 1];
        const uint8_t* previous = p.keyBytes + previousOffset;
        bool equal = previousOffset + static_cast<unsigned long long>(length) < p.keyBytesTotal;
        for (uint32_t i = 0; equal && i < length; ++i) {
            equal = previous[i] == key[i];
        }
        insert = !(equal && previous[length] == 0);
    }
    h = finishWordHash(h);
    const uint32_t tag = static_cast<uint32_t>(h >> 32);
    const unsigned long long claim = (static_cast<unsigned long long>(row) << 32) | tag;   // little endian: x = tag, y = row
    // (at most half of the slots are ever taken: the loop ends)
    for (uint32_t at = static_cast<uint32_t>(h) & p.slotMask; insert; at = (at + 1) & p.slotMask) {
        unsigned long long* head = reinterpret_cast<unsigned long long*>(p.slots + at);
        if (atomicCAS(head, ~0ull, claim) == ~0ull) {
            head[1] = (static_cast<unsigned long long>(length) << 32) | offset;            // z = offset, w = length
            break;
        }
    }
    const unsigned long long inserting = __ballot(insert);
    if (inserting && static_cast<uint32_t>(__ffsll(static_cast<long long>(__ballot(1))) - 1) == (threadIdx.x & (WAVE - 1))) {
        atomicAdd(p.inserted, static_cast<uint32_t>(__popcll(inserting)));
    }
}

constexpr uint32_t RESOLVE_MAX_MODELS = 4;

struct ResolveParams {
    const uint8_t* queryBytes;      // the batch's words; device memory, or pinned host memory read over PCIe
    const uint32_t* queryOffsets;   // word i = queryBytes[queryOffsets[at(i)] .. queryOffsets[at(i) + 1]) with
                                    // at(i) = i (jobShift == 0: n + 1 entries) or i + (i >> jobShift): every job of
                                    // 2^jobShift words has an entry for the end of its last word (memb_hip_words_plan)
    unsigned long long first;       // the launch looks up words [first, first + n) (first is a multiple of 64)
    uint32_t jobShift;
    unsigned long long n;
    unsigned long long queryBytesTotal;   // bytes behind queryBytes: a word that claims more is answered MISSING
    // One word, several models (a ReadersUnion resolves one batch against every reader): the word is fetched and hashed
    // once and probed in every model's table.
    uint32_t models;                // 1 .. RESOLVE_MAX_MODELS
    const WordSlot* slots[RESOLVE_MAX_MODELS];
    uint32_t slotMask[RESOLVE_MAX_MODELS];
    const uint8_t* keyBytes[RESOLVE_MAX_MODELS];
    uint32_t* rows[RESOLVE_MAX_MODELS];   // out: [first + n] each
    uint32_t stageQueries;          // queryBytes is 16-byte aligned: a wavefront's words go through LDS
};

constexpr uint32_t RESOLVE_WAVES = 4;            // wavefronts per block
constexpr uint32_t RESOLVE_STAGE_PIECES = 128;   // 16-byte pieces of LDS per wavefront: 64 words of up to 31 bytes on average

// The probe, for query bytes in LDS or in global memory (one instantiation per call site, so that the
// compiler knows the address space of `query`).
template <typename QueryBytes>
__device__ __forceinline__ unsigned long long hashWord(QueryBytes query, uint32_t length)
{
    unsigned long long h = FNV_OFFSET_BASIS;
    for (uint32_t i = 0; i < length; ++i) {
        h = (h ^ query[i]) * FNV_PRIME;
    }
    return finishWordHash(h);
}

template <typename QueryBytes>
__device__ __forceinline__ uint32_t probeWord(
    const WordSlot* slots, uint32_t slotMask, const uint8_t* keyBytes, unsigned long long h, QueryBytes query, uint32_t length)
{
    const uint32_t tag = static_cast<uint32_t>(h >> 32);
    uint32_t at = static_cast<uint32_t>(h) & slotMask;
    // (the table is at most half full: an empty slot ends every probe sequence; the bound is for a table
    // that is not what build_word_table leaves behind)
    for (uint32_t probes = 0; probes <= slotMask; ++probes, at = (at + 1) & slotMask) {
        const WordSlot slot = slots[at];
        if (slot.y == WORD_SLOT_EMPTY) {
            break;
        }
        if (slot.x != tag || slot.w != length) {
            continue;
        }
        const uint8_t* key = keyBytes + slot.z;
        uint32_t difference = 0;
        for (uint32_t i = 0; i < length; i += 4) {
            // four independent loads per round; the last round re-reads the last byte
            const uint32_t i1 = min(i + 1, length - 1), i2 = min(i + 2, length - 1), i3 = min(i + 3, length - 1);
            const uint32_t k0 = key[i], k1 = key[i1], k2 = key[i2], k3 = key[i3];
            difference |= (k0 ^ query[i]) | (k1 ^ query[i1]) | (k2 ^ query[i2]) | (k3 ^ query[i3]);
        }
        if (difference == 0) {
            return slot.y;
        }
    }
    return MISSING;
}

// One lane per word, 64 consecutive words per wavefront. Their bytes are one contiguous run of the packed
// batch (a job's words lie back to back and a job is a multiple of 64 words): the wavefront copies it into
// LDS with 16-byte loads (coalesced; a lane reading its own word byte by byte would issue a request per
// byte -- and the batch usually lies in pinned HOST memory: the loads are PCIe reads, the copy to the device
// and the lookup are this one kernel) and every lane hashes and compares out of LDS. Runs that do not fit
// (very long words) are read in place instead.
__global__ __launch_bounds__(RESOLVE_WAVES * WAVE) void resolve_words(ResolveParams p)
{
    __shared__ u32x4 stage[RESOLVE_WAVES][RESOLVE_STAGE_PIECES];
    const uint32_t lane = threadIdx.x & (WAVE - 1);
    const uint32_t wave = threadIdx.x / WAVE;
    const unsigned long long base = (static_cast<unsigned long long>(blockIdx.x) * RESOLVE_WAVES + wave) * WAVE;
    if (base >= p.n) {
        return;
    }
    const bool valid = base + lane < p.n;
    const unsigned long long word = p.first + (valid ? base + lane : p.n - 1);
    const unsigned long long index = word + (p.jobShift ? word >> p.jobShift : 0ull);
    const uint32_t begin = p.queryOffsets[index];
    const uint32_t end = p.queryOffsets[index + 1];
    const bool sane = begin <= end && end <= p.queryBytesTotal;   // (offsets that are not what they should be: MISSING, never a wild read)
    const uint32_t length = sane ? end - begin : 0u;
    const uint32_t waveBegin = __shfl(begin, 0);
    const uint32_t waveEnd = __shfl(end, WAVE - 1);   // (lanes past the batch end hold the last word)
    const uint32_t alignedBegin = waveBegin & ~15u;
    // (every model's answer, then the stores: p.models is small and uniform)
    auto lookUp = [&](auto query) {
        const unsigned long long h = hashWord(query, length);
        for (uint32_t m = 0; m < p.models; ++m) {
            const uint32_t row = probeWord(p.slots[m], p.slotMask[m], p.keyBytes[m], h, query, length);
            p.rows[m][p.first + base + lane] = row;
        }
    };
    auto allMissing = [&] {
        for (uint32_t m = 0; m < p.models; ++m) {
            p.rows[m][p.first + base + lane] = MISSING;
        }
    };
    if (p.stageQueries && waveEnd >= alignedBegin && waveEnd - alignedBegin <= RESOLVE_STAGE_PIECES * 16 &&
        __all(sane && begin >= waveBegin && end <= waveEnd)) {   // wave-uniform
        const uint32_t pieces = (waveEnd - alignedBegin + 15) / 16;
        const u32x4* source = reinterpret_cast<const u32x4*>(p.queryBytes + alignedBegin);
        // (every piece holds at least one byte of the batch, and an aligned 16-byte load never leaves the page of its
        // first byte: nothing past the allocation is touched)
        for (uint32_t piece = lane; piece < pieces; piece += WAVE) {
            stage[wave][piece] = source[piece];
        }
        waveLdsFence();
        const uint8_t* query = reinterpret_cast<const uint8_t*>(&stage[wave][0]) + (begin - alignedBegin);
        if (valid) {
            lookUp(query);
        }
    } else if (valid) {
        if (sane) {
            lookUp(p.queryBytes + begin);
        } else {
            allMissing();
        }
    }
}

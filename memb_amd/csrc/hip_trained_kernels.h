// decode_trained / decode_records_persistent / decode_union_split / decode_trained_union / repack_streams:
// canonical-Huffman bitstream decode + codebook gather.
//
// Device code of libmemb_hip.so (gfx950 / CDNA4). Included by memb_hip.hip only,
// inside its anonymous namespace; see that file for the overview.
#pragma once

// ---------------------------------------------------------------------------
// decode_trained
// ---------------------------------------------------------------------------

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));   // plain SSA value (HIP's uint4 is a class)

struct TrainedParams {
    const uint32_t* rows;        // batch -> row id; null = identity (row = batch position)
    float* out;
    unsigned long long n;
    unsigned long long ld;
    unsigned long long colOff;
    const uint4* streams;           // re-packed bitstreams: big-endian dwords, one 16-byte aligned run per row
    const uint32_t* streamStarts;   // [nRows + 1] first 16-byte piece of each row's run
    const uint16_t* segmentIndex;   // [nRows][lanesPerWord - 1] bit offsets of segments 1.. from the stream start
    const uint32_t* rowMeta;        // or null: one 16-byte record per row, {start, offsets of segments 1..7 in 13 bits
                                    // each}: what streamStarts and segmentIndex hold, fetched by ONE load per lane
    uint32_t recordPieces;          // non-zero: ROW RECORDS -- `streams` holds one fixed-size region of this many
                                    // 16-byte pieces per row, row r at piece r * recordPieces: piece 0 = the record
                                    // {stream bytes, 7 x 13-bit segment offsets}, the bitstream from piece 1 on. A
                                    // row's address needs no lookup, its offsets arrive with its stream: a random
                                    // word costs two line requests (index entry + straddling stream: 3.2)
    uint32_t loadPieces;            // pieces copied to LDS per word (recordPieces, or the whole slot: slotDwords / 4)
    uint16_t* segmentIndexOut;      // OUT_INDEX: index being built, [nRows][indexLanes - 1]
                                    // (both hold 32-bit entries when indexWide: rows longer than 65535 bits)
                                    // Lookup modes of decode_trained: null, or where the launch leaves what it SAW of the
                                    // batch's order (one uint32_t in pinned host memory, noteBatchOrder below)
    const uint32_t* table;          // 8-byte entries, see TableEntry
    const float* codebook;          // 256 centroids, or 256 centroid pairs (FAST)
    unsigned long long nRows;
    uint32_t tableDwords;     // multiple of 4
    uint32_t codebookDwords;  // 256 or 512
    uint32_t rootBits;
    uint32_t dim;
    uint32_t slotDwords;      // LDS dwords reserved per bitstream, multiple of 4
    uint32_t slotMagic;       // fastDivide magic for loadPieces
    uint32_t lanesPerWord;    // G
    uint32_t laneMagic;       // fastDivide magic for G
    uint32_t wordsPerWave;    // 64 / G
    uint32_t segmentSymbols;  // S, multiple of the decode group (4, or 8 when FAST)
    uint32_t keyRowBytes;     // bytes per word in the symbol tile
    uint32_t keyTileDwords;   // dwords of the symbol tile of one wave
    uint32_t pieceMagic;      // fastDivide magic for dim / 4 (vector output)
    uint32_t indexLanes;      // OUT_INDEX: lanes per word of the index being built
    uint32_t indexSegmentSymbols;
    uint32_t indexWide;       // segment index entries are uint32_t rather than uint16_t
    uint32_t fineIndex;       // row records whose segment offsets come from `segmentIndex` ([nRows][lanesPerWord - 1], the FINER
                              // index of small batches: more lanes per word than the record in front of a row's stream has
                              // offsets for) instead of from that record; both loads hang off the row id, side by side
    uint32_t debugFlags;      // measurement builds only (MEMB_HIP_MEASURE; see measureFlags below)
    uint32_t tilesPerWave;    // decode_trained / decode_union_split: tiles a wavefront decodes one after the other (>= 1):
                              // the block's copy of table and codebook into LDS is paid once for all of them
    uint32_t accumulate;      // epilogue: add to what the output already holds ...
    float divisor;            // ... and / or divide by this (0 = no division)
};

// Measurement switches, TrainedParams::debugFlags (option `debug` / MEMB_HIP_DEBUG): bit 0 skip the decode, bit 1 skip
// the output, bit 2 skip the row id / index / bitstream loads (the decoder then chews on whatever LDS holds: output values
// are garbage, the access pattern is kept), bit 13 (0x2000) store constants instead of gathering centroids from LDS,
// bit 14 (0x4000) no copy of table and codebook into LDS, bit 15 (0x8000) decode_union_split's tiles per wavefront a grid apart,
// bit 16 (0x10000) row ids are not loaded (row = batch position: right for a key-order dump, the dependent load in front of it gone).
// They exist in builds with -DMEMB_HIP_MEASURE only
// (tools/perf/build_measure.py): the shipped library folds every one of these branches away, so no environment
// variable or option can make it write anything but the decoded rows.
__device__ __forceinline__ uint32_t measureFlags(const TrainedParams& p)
{
#ifdef MEMB_HIP_MEASURE
    return p.debugFlags;
#else
    (void)p;
    return 0u;
#endif
}

// One 16-byte piece of a row's bitstream (or an index record): a plain load. (Non-temporal loads were built as a
// template argument in round 3 and measured on one allocation: -0.5 % on the key-order dump, +0.7 % shuffled, +28..48 %
// on 100 000 rows, whose streams then no longer survive between launches in L2 / Infinity Cache -- removed in round 4.)
__device__ __forceinline__ u32x4 loadPiece(const u32x4* source)
{
    return *source;
}

// OUT_KEYS: no codebook gather -- `out` receives the symbol tile itself, dense rows of
// keyRowBytes (one centroid index per byte, or per nibble when FAST), for the host-buffer
// entry point: PCIe then carries 1/4 or 1/8 of the bytes and the host threads that copy rows
// out of pinned memory anyway expand them (memb_hip.hip: expandKeyRows).
enum OutputMode { OUT_SCALAR = 0, OUT_VEC4 = 1, OUT_FLAT = 2, OUT_INDEX = 3, OUT_KEYS = 4 };


// Device form of one lookup-table entry (logical layout: memb::DecodeTable).
//   x: leaf    -> code length                       (bits 8..31 zero)
//      pointer -> TABLE_POINTER_FLAG | extra bits | first sub-table entry << 8
//   y: leaf    -> the symbol replicated into every byte (or every nibble, FAST),
//                 so that packing symbol s of a group is one AND-OR with a
//                 constant mask
struct TableEntry {
    uint32_t x;
    uint32_t y;
};

// ---- building blocks shared by the kernels ----

struct LaneRole {
    uint32_t word;      // word of the tile this lane works on
    uint32_t segment;   // segment of that word
    bool spare;         // 64 % G lanes at the top: decode word 0's slot, store nothing
};

__device__ __forceinline__ LaneRole laneRole(const TrainedParams& p, uint32_t lane)
{
    LaneRole role;
    const uint32_t laneWord = fastDivide(lane, p.laneMagic, p.lanesPerWord);
    role.segment = lane - laneWord * p.lanesPerWord;
    role.spare = laneWord >= p.wordsPerWave;
    role.word = role.spare ? 0 : laneWord;
    return role;
}

// Row id of the lane's word in tile `tile`; MISSING for padding lanes and past the batch end.
__device__ __forceinline__ uint32_t loadTileRow(const TrainedParams& p, unsigned long long tile, const LaneRole& role)
{
    const unsigned long long index = tile * p.wordsPerWave + role.word;
    if (role.spare || index >= p.n) {
        return MISSING;
    }
    return p.rows && !(measureFlags(p) & (4 | 0x10000)) ? p.rows[index] : static_cast<uint32_t>(index);
}

constexpr uint32_t ROW_META_BITS = 13;        // a segment offset inside a rowMeta record: streams below 1 KiB
constexpr uint32_t ROW_META_MAX_LANES = 8;    // 32 + 7 * 13 bits = 123 of 128

struct WordMeta {
    uint32_t row;
    uint32_t start;         // first 16-byte piece of the word's bitstream
    uint32_t segmentBits;   // bit offset of the lane's segment inside that stream; with rowMeta: dword 1 of the record
    uint32_t packed2;       // with rowMeta: dwords 2 and 3 of the record, until unpackMeta has run
    uint32_t packed3;
};

// Issues the loads only; the values may be used after unpackMeta.
__device__ __forceinline__ WordMeta loadWordMeta(const TrainedParams& p, uint32_t row, const LaneRole& role)
{
    WordMeta meta;
    meta.row = row;
    meta.start = 0;
    meta.segmentBits = 0;
    meta.packed2 = 0;
    meta.packed3 = 0;
    if (measureFlags(p) & 4) {
        return meta;
    }
    if (p.recordPieces) {   // row records: the address is arithmetic, the offsets come with the stream
        meta.start = row < p.nRows ? row * p.recordPieces : 0u;
        if (p.fineIndex && row < p.nRows && role.segment > 0) {   // ... or from the finer index, in parallel with it
            meta.segmentBits = p.segmentIndex[static_cast<unsigned long long>(row) * (p.lanesPerWord - 1) + role.segment - 1];
        }
        return meta;
    }
    if (row < p.nRows && p.rowMeta) {
        const u32x4* source = reinterpret_cast<const u32x4*>(p.rowMeta) + row;
        const u32x4 record = loadPiece(source);
        meta.start = record.x;
        meta.segmentBits = record.y;
        meta.packed2 = record.z;
        meta.packed3 = record.w;
    } else if (row < p.nRows) {
        meta.start = p.streamStarts[row];
        if (role.segment > 0) {
            const unsigned long long at = static_cast<unsigned long long>(row) * (p.lanesPerWord - 1) + role.segment - 1;
            meta.segmentBits = p.indexWide ? reinterpret_cast<const uint32_t*>(p.segmentIndex)[at] : p.segmentIndex[at];
        }
    }
    return meta;
}

// The lane's 13-bit field out of a record's 96 offset bits (field s - 1 of segment s at bit 13 (s - 1)).
__device__ __forceinline__ uint32_t segmentField(uint32_t segment, uint32_t bits0, uint32_t bits1, uint32_t bits2)
{
    // (two conditional moves per step, spelled out: a three-way select by index is turned into
    // a table in scratch memory)
    const uint32_t bit = segment ? ROW_META_BITS * (segment - 1) : 0u;
    const bool second = bit >= 32;
    const bool third = bit >= 64;
    uint32_t low = second ? bits1 : bits0;
    uint32_t high = second ? bits2 : bits1;
    low = third ? bits2 : low;
    high = third ? 0u : high;
    const uint32_t field = __builtin_amdgcn_alignbit(high, low, bit & 31) & ((1u << ROW_META_BITS) - 1);
    return segment ? field : 0u;
}

// rowMeta: the lane's segment offset out of the record just loaded; no-op for the other layouts
// (two arrays: already in segmentBits; row records: read from LDS by decodeSegment).
__device__ __forceinline__ void unpackMeta(const TrainedParams& p, const LaneRole& role, WordMeta& meta)
{
    if (!p.rowMeta || p.recordPieces) {
        return;
    }
    meta.segmentBits = segmentField(role.segment, meta.segmentBits, meta.packed2, meta.packed3);
}

// Scalar registers the one-tile kernels may use (0 = no limit). The hardware admits wavefronts by scalar registers as well
// as vector ones -- 800 per SIMD, .sgpr_count (this budget minus 2: VCC and friends are counted in) rounded up to 16, plus
// 16 -- and without a limit the compiler takes 106 for these kernels: six wavefronts per SIMD where 61 vector registers
// allow eight. 96 -> seven (memb_hip.hip: ONE_TILE_WAVES_PER_CU; HISTORY.md, "(r5) 5.0", has the measurements of 80, 88 and none).
#ifndef MEMB_HIP_SGPRS
#define MEMB_HIP_SGPRS 96
#endif
#if MEMB_HIP_SGPRS
#define MEMB_SGPR_BUDGET __attribute__((amdgpu_num_sgpr(MEMB_HIP_SGPRS)))
#else
#define MEMB_SGPR_BUDGET
#endif

// The output burst of decode_records_persistent for nibble keys: its registers decide between five and six wavefronts per
// SIMD (82 / 78 vector registers with bursts of 5 / 4 in the dense kernel), and the sixth is worth more than the fifth
// piece of a burst in this kernel's class (57 000 - 131 000 rows; round 5, batch 24, profiles/r05_experiments.txt:
// 100 000 rows with nothing cached -5.3 % for the 4-bit model, -7 % for the 2-bit one). Byte keys keep the burst of 5
// (78 registers: six already; with 4 they would run seven and gain nothing: -4..+3 %).
#ifndef MEMB_HIP_RECORDS_BURST_NIBBLE
#define MEMB_HIP_RECORDS_BURST_NIBBLE 4
#endif

// decode_records_persistent held to the vector registers of N wavefronts per SIMD (0 = the compiler's own 78-85: five, for
// the nibble-key dense kernel). Measured with 6 (round 5, batch 23, profiles/r05_experiments.txt): 12 bytes of scratch in
// that kernel, BASELINE configs[1] -0.9 % uncached and +5.6 % cached, the rest of its class -6..+1 %: not taken.
#ifndef MEMB_HIP_RECORDS_WAVES
#define MEMB_HIP_RECORDS_WAVES 0
#endif
#ifndef MEMB_HIP_RECORDS_SGPRS
#define MEMB_HIP_RECORDS_SGPRS 0   // (a scalar-register budget for the same kernel: experiments)
#endif
#if MEMB_HIP_RECORDS_WAVES && MEMB_HIP_RECORDS_SGPRS
#define MEMB_RECORDS_WAVES __attribute__((amdgpu_waves_per_eu(MEMB_HIP_RECORDS_WAVES), amdgpu_num_sgpr(MEMB_HIP_RECORDS_SGPRS)))
#elif MEMB_HIP_RECORDS_WAVES
#define MEMB_RECORDS_WAVES __attribute__((amdgpu_waves_per_eu(MEMB_HIP_RECORDS_WAVES)))
#elif MEMB_HIP_RECORDS_SGPRS
#define MEMB_RECORDS_WAVES __attribute__((amdgpu_num_sgpr(MEMB_HIP_RECORDS_SGPRS)))
#else
#define MEMB_RECORDS_WAVES
#endif

#ifndef MEMB_HIP_OUTPUT_BURST
#define MEMB_HIP_OUTPUT_BURST 5   // 16-byte pieces a lane gathers before it stores them back to back (outputTile)
#endif

constexpr int STREAM_REGISTERS = 4;   // 16-byte pieces one lane can hold for a prefetched tile

// Named members, not an array: indexed storage ends up in scratch memory, and a
// load whose result goes to scratch is waited for at once, which would undo the prefetch.

struct StreamRegisters {
    u32x4 r0, r1, r2, r3;
};

// One round of the tile's bitstream copy: piece q = (word, 16-byte piece) -> one lane.
// A slot's worth of pieces is read from each row's start (running into the next
// rows' streams, which is harmless; the array ends with a guard of one slot).
// Absent words read the start of the array and never emit what they decode;
// lanes past the tile's last piece re-read its last piece.
__device__ __forceinline__ void issueStreamLoad(
    const TrainedParams& p, uint32_t sourceStart, uint32_t lane, uint32_t round, u32x4& destination)
{
    const uint32_t piecesPerWord = p.loadPieces;
    const uint32_t totalPieces = p.wordsPerWave * piecesPerWord;
    if (round * WAVE < totalPieces && !(measureFlags(p) & 4)) {   // wave-uniform
        const uint32_t q = min(round * WAVE + lane, totalPieces - 1);
        const uint32_t w = fastDivide(q, p.slotMagic, piecesPerWord);
        const uint32_t piece = q - w * piecesPerWord;
        const uint32_t wordStart = __shfl(sourceStart, w * p.lanesPerWord);
        const u32x4* source = reinterpret_cast<const u32x4*>(p.streams) + (static_cast<unsigned long long>(wordStart) + piece);
        destination = loadPiece(source);
    }
}

__device__ __forceinline__ void issueStreamLoads(
    const TrainedParams& p, const WordMeta& meta, uint32_t lane, uint32_t firstRound, StreamRegisters& v)
{
    const uint32_t sourceStart = meta.row < p.nRows ? meta.start : 0u;
    issueStreamLoad(p, sourceStart, lane, firstRound + 0, v.r0);
    issueStreamLoad(p, sourceStart, lane, firstRound + 1, v.r1);
    issueStreamLoad(p, sourceStart, lane, firstRound + 2, v.r2);
    issueStreamLoad(p, sourceStart, lane, firstRound + 3, v.r3);
}

// Into the LDS slots (already big-endian dwords, so the decoder extracts bits with plain shifts).
__device__ __forceinline__ void writeStream(
    const TrainedParams& p, uint32_t* slots, uint32_t lane, uint32_t round, const u32x4& value)
{
    const uint32_t piecesPerWord = p.loadPieces;
    const uint32_t totalPieces = p.wordsPerWave * piecesPerWord;
    const uint32_t q = round * WAVE + lane;
    if (q < totalPieces) {
        const uint32_t w = fastDivide(q, p.slotMagic, piecesPerWord);
        const uint32_t piece = q - w * piecesPerWord;
        *reinterpret_cast<u32x4*>(slots + w * p.slotDwords + 4 * piece) = value;
    }
}

__device__ __forceinline__ void writeStreams(
    const TrainedParams& p, uint32_t* slots, uint32_t lane, uint32_t firstRound, const StreamRegisters& v)
{
    writeStream(p, slots, lane, firstRound + 0, v.r0);
    writeStream(p, slots, lane, firstRound + 1, v.r1);
    writeStream(p, slots, lane, firstRound + 2, v.r2);
    writeStream(p, slots, lane, firstRound + 3, v.r3);
}

// Row records: the lane's segment offset out of the record at the head of its word's LDS slot, once
// the slot has been written (and fenced). No-op for the other layouts.
__device__ __forceinline__ void recordSegmentBits(
    const TrainedParams& p, const uint32_t* slots, const LaneRole& role, WordMeta& meta)
{
    if (p.recordPieces && !p.fineIndex) {
        const uint32_t* slot = slots + role.word * p.slotDwords;
        meta.segmentBits = segmentField(role.segment, slot[1], slot[2], slot[3]);
    }
}

// FAST: codebook of at most 16 centroids and no code longer than 8 bits (2- and
// 4-bit models). Eight symbols are decoded per 64-bit window (7 * 8 consumed bits
// + 8 looked-ahead bits fit), symbols are staged as nibbles, and the output phase
// fetches two centroids per LDS read from a 256-entry table of pairs.
//
// Decode the lane's segment from its word's LDS slot into the symbol tile
// (or, OUT_INDEX, record segment start positions).
//
// PACKED (byte keys: more than 16 centroids or codes longer than 8 bits): table entries are the host
// table's 4-byte form {length | symbol << 8, or a pointer} -- one ds_read_b32 per symbol where the
// 8-byte form cost the compiler two (length first, symbol later), the symbol moved into its byte of
// the group by one v_perm_b32. The first level covers the longest code whenever 32 KiB hold it
// (13 bits), so the second-level branch disappears from all but pathological codes. Measured on the
// byte-key models: the decode alone went from 0.51 to 0.33 ms (6-bit) and from
// 0.56 to 0.36 ms (8-bit). Bank copies of table and codebook (one copy per LDS bank: conflict-free
// lookups, 63 % -> 28 % conflict cycles) were built and measured too; they changed nothing in time
// and cost small batches their setup, so they are not here.
template <bool HAS_SUB, int MODE, bool FAST, bool PACKED = false>
__device__ __forceinline__ void decodeSegment(
    const TrainedParams& p, const TableEntry* tableLds, const uint32_t* slots, uint32_t* keyTile,
    const LaneRole& role, const WordMeta& meta, uint32_t rootBits = 0)   // rootBits: of the lane's table when lanes differ
{
    // (FAST && PACKED: nibble keys through the 4-byte table -- one ds_read_b32 per symbol and half the table to copy into
    // LDS, the symbol moved into its nibble by a bit-field extract and a shift-or instead of one and-or with a mask)
    constexpr int GROUP = FAST ? 8 : 4;
    constexpr uint32_t KEY_BITS = FAST ? 4 : 8;
    constexpr uint32_t KEY_MASK = FAST ? 0xFu : 0xFFu;

    const bool present = meta.row < p.nRows;
    const uint32_t* slot = slots + role.word * p.slotDwords;
    uint8_t* keyBytes = reinterpret_cast<uint8_t*>(keyTile);
    uint32_t lastWindow = p.slotDwords - 3;
    const uint32_t root = rootBits ? rootBits : p.rootBits;   // ONE width for both levels: the sub-table index
    const uint32_t rootShift = 32 - root;                    // starts where the lane's own first level ended
    uint32_t bitPos = meta.segmentBits;   // streams start on a slot boundary
    if (p.recordPieces) {
        // row records: the slot begins with the row's record (read by recordSegmentBits), the bitstream follows it
        slot += 4;
        lastWindow -= 4;
    }
    // byte position of this lane's first group inside the symbol tile
    uint32_t keyOffset = role.word * p.keyRowBytes + role.segment * (p.segmentSymbols * KEY_BITS / 8);
    const uint32_t keyRowEnd = role.spare ? 0 : (role.word + 1) * p.keyRowBytes;
    const uint32_t absentFill = present ? 0u : 0xFFFFFFFFu;   // byte keys: ZERO_KEY everywhere
    uint32_t nextIndexSymbol = p.indexSegmentSymbols;
    uint32_t indexSlot = 0;
    const uint32_t* table32 = reinterpret_cast<const uint32_t*>(tableLds);   // PACKED

    for (uint32_t j = 0; j < p.segmentSymbols; j += GROUP) {
        if (MODE == OUT_INDEX) {
            // one lane per word here; record where every indexSegmentSymbols-th symbol starts
            if (j == nextIndexSymbol) {
                if (present && indexSlot + 1 < p.indexLanes) {
                    const unsigned long long at = static_cast<unsigned long long>(meta.row) * (p.indexLanes - 1) + indexSlot;
                    if (p.indexWide) {
                        reinterpret_cast<uint32_t*>(p.segmentIndexOut)[at] = bitPos;
                    } else {
                        p.segmentIndexOut[at] = static_cast<uint16_t>(bitPos);
                    }
                }
                ++indexSlot;
                nextIndexSymbol += p.indexSegmentSymbols;
            }
        }
        const uint32_t d = min(bitPos >> 5, lastWindow);
        const uint32_t shift = bitPos & 31;
        const uint32_t w0 = slot[d];
        const uint32_t w1 = slot[d + 1];
        const uint32_t w2 = slot[d + 2];
        // 64 valid bits starting at the current bit position, MSB first.
        unsigned long long window = ((static_cast<unsigned long long>(w0) << 32) | w1) << shift;
        window |= static_cast<uint32_t>(static_cast<unsigned long long>(w2) >> (32 - shift));
        uint32_t keys = 0;
        uint32_t lengths = 0;
        if (PACKED) {
#pragma unroll
            for (int s = 0; s < GROUP; ++s) {
                uint32_t entry = table32[static_cast<uint32_t>(window >> 32) >> rootShift];
                if (HAS_SUB) {
                    if (entry & memb::TABLE_POINTER_FLAG) {
                        const uint32_t subBits = entry & 0xff;
                        const uint32_t base = (entry & ~memb::TABLE_POINTER_FLAG) >> 8;
                        const uint32_t subIndex =
                            static_cast<uint32_t>((window << root) >> 32) >> (32 - subBits);
                        entry = table32[base + subIndex];
                    }
                }
                window <<= (entry & 63);
                lengths += entry;   // the low byte sums the lengths (4 x 16 or 8 x 8 at most), the symbols pile up above it
                if (FAST) {
                    keys |= ((entry >> 8) & 0xFu) << (4 * s);
                } else {
                    // symbol (byte 1 of the entry) into byte s of the group: one v_perm_b32
                    // (selector bytes 0..3 pick bytes of `keys`, 5 picks byte 1 of `entry`)
                    const uint32_t select = s == 0 ? 0x03020105u : s == 1 ? 0x03020500u : s == 2 ? 0x03050100u : 0x05020100u;
                    keys = __builtin_amdgcn_perm(entry, keys, select);
                }
            }
            lengths &= 0xff;
        } else {
#pragma unroll
        for (int s = 0; s < GROUP; ++s) {
            TableEntry entry = tableLds[static_cast<uint32_t>(window >> 32) >> rootShift];
            if (HAS_SUB) {
                if (entry.x & memb::TABLE_POINTER_FLAG) {
                    const uint32_t subBits = entry.x & 0xff;
                    const uint32_t base = (entry.x & ~memb::TABLE_POINTER_FLAG) >> 8;
                    const uint32_t subIndex =
                        static_cast<uint32_t>((window << root) >> 32) >> (32 - subBits);
                    entry = tableLds[base + subIndex];
                }
            }
            window <<= (entry.x & 63);
            lengths += entry.x;
            keys |= entry.y & (KEY_MASK << (KEY_BITS * s));
        }
        }
        bitPos += lengths;
        if (MODE != OUT_INDEX) {
            keys |= absentFill;
            if (FAST) {
                // rows are dim / 2 bytes: 2-byte aligned only
                if (keyOffset + 2 <= keyRowEnd) {
                    *reinterpret_cast<uint16_t*>(keyBytes + keyOffset) = static_cast<uint16_t>(keys);
                }
                if (keyOffset + 4 <= keyRowEnd) {
                    *reinterpret_cast<uint16_t*>(keyBytes + keyOffset + 2) = static_cast<uint16_t>(keys >> 16);
                }
            } else {
                if (keyOffset + 4 <= keyRowEnd) {
                    *reinterpret_cast<uint32_t*>(keyBytes + keyOffset) = keys;
                }
            }
            keyOffset += 4;
        }
    }
}

// Symbol tile -> fp32 rows: codebook gather and row-contiguous stores.
template <int MODE, bool FAST, int BURST = MEMB_HIP_OUTPUT_BURST>
__device__ __forceinline__ void outputTile(
    const TrainedParams& p, const uint32_t* codebookLds, const uint32_t* keyTile, unsigned long long tileBase,
    uint32_t tileWords, uint32_t lane, const LaneRole& role, bool present)
{
    if (MODE == OUT_KEYS) {
        // rows of a tile are dense in LDS and in the output; absent words carry whatever was
        // decoded for them (the host knows them by their row ids)
        const unsigned long long firstByte = tileBase * p.keyRowBytes;
        uint8_t* keysOut = reinterpret_cast<uint8_t*>(p.out) + firstByte;
        const uint32_t bytes = tileWords * p.keyRowBytes;   // even
        if ((firstByte & 3) == 0) {
            for (uint32_t q = lane; q < bytes / 4; q += WAVE) {
                reinterpret_cast<uint32_t*>(keysOut)[q] = keyTile[q];
            }
            if ((bytes & 2) && lane == 0) {   // the next tile owns the bytes after these two
                reinterpret_cast<uint16_t*>(keysOut)[bytes / 2 - 1] = reinterpret_cast<const uint16_t*>(keyTile)[bytes / 2 - 1];
            }
        } else {
            for (uint32_t q = lane; q < bytes / 2; q += WAVE) {
                reinterpret_cast<uint16_t*>(keysOut)[q] = reinterpret_cast<const uint16_t*>(keyTile)[q];
            }
        }
        return;
    }
    const float* centroidLds = reinterpret_cast<const float*>(codebookLds);
    const float2* pairLds = reinterpret_cast<const float2*>(codebookLds);
    const uint8_t* keyBytes = reinterpret_cast<const uint8_t*>(keyTile);

    // Nibble keys have no spare code for "absent" (byte keys use ZERO_KEY): in a
    // tile that contains absent words every piece looks up its word in the ballot
    // and absent ones become zeros. (Writing the tile first and zeroing those rows
    // afterwards relies on two stores of one wave to one address landing in
    // order; that held in HBM but not for rows written over PCIe into pinned
    // host memory.)
    const bool hasEpilogue = p.accumulate || p.divisor != 0.f;   // wave-uniform
    unsigned long long absent = 0;
    if (FAST) {
        absent = __ballot(!present && !role.spare && role.segment == 0 && role.word < tileWords);
    }
    const bool checkWords = FAST && absent != 0;

    if (MODE == OUT_FLAT || MODE == OUT_VEC4) {
        // Piece q = 4 consecutive floats; the symbol tile is linear in q for both layouts
        // (byte keys: rows of dim bytes; nibble keys: rows of dim / 2 bytes).
        // BURST pieces per lane are gathered first and then stored back to back, so a
        // tile reaches memory as one burst of consecutive KiBs rather than one KiB per
        // LDS round trip.
        const uint32_t piecesPerWord = p.dim / 4;
        const uint32_t pieces = tileWords * piecesPerWord;
        float* tileOut = p.out + tileBase * p.ld + p.colOff;
        const bool noGather = (measureFlags(p) & 0x2000) != 0;   // measurement: store constants, no LDS reads
        for (uint32_t q0 = lane; q0 < pieces; q0 += WAVE * BURST) {
            uint32_t k[BURST];
            float4 f[BURST];
#pragma unroll
            for (int u = 0; u < BURST; ++u) {
                const uint32_t q = min(q0 + WAVE * u, pieces - 1);
                k[u] = noGather ? 0u : (FAST ? reinterpret_cast<const uint16_t*>(keyTile)[q] : keyTile[q]);
            }
#pragma unroll
            for (int u = 0; u < BURST; ++u) {
                if (noGather) {
                    f[u] = make_float4(1.f, 2.f, 3.f, 4.f);
                } else if (FAST) {
                    const float2 a = pairLds[k[u] & 0xff];
                    const float2 b = pairLds[k[u] >> 8];
                    f[u] = make_float4(a.x, a.y, b.x, b.y);
                } else {
                    f[u].x = centroidLds[k[u] & 0xff];
                    f[u].y = centroidLds[(k[u] >> 8) & 0xff];
                    f[u].z = centroidLds[(k[u] >> 16) & 0xff];
                    f[u].w = centroidLds[k[u] >> 24];
                }
            }
#pragma unroll
            for (int u = 0; u < BURST; ++u) {
                const uint32_t q = q0 + WAVE * u;
                if (q < pieces) {
                    float* destination;
                    if (MODE == OUT_FLAT && !checkWords) {
                        destination = tileOut + 4 * static_cast<size_t>(q);
                    } else {
                        const uint32_t w = fastDivide(q, p.pieceMagic, piecesPerWord);
                        const uint32_t c = q - w * piecesPerWord;
                        destination = tileOut + w * p.ld + 4 * c;
                        if (checkWords && ((absent >> (w * p.lanesPerWord)) & 1)) {
                            f[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                        }
                    }
                    if (hasEpilogue) {   // off the common path
                        f[u] = epilogue4(f[u], destination, p.accumulate, p.divisor);
                    }
                    *reinterpret_cast<float4*>(destination) = f[u];
                }
            }
        }
    } else {
        const uint32_t total = tileWords * p.dim;
        for (uint32_t q = lane; q < total; q += WAVE) {
            const uint32_t w = q / p.dim;
            const uint32_t c = q - w * p.dim;
            float value;
            if (FAST) {
                const uint32_t k = keyBytes[w * p.keyRowBytes + (c >> 1)];
                value = pairLds[(k >> (4 * (c & 1))) & 15].x;
                if (checkWords && ((absent >> (w * p.lanesPerWord)) & 1)) {
                    value = 0.f;
                }
            } else {
                value = centroidLds[keyBytes[w * p.keyRowBytes + c]];
            }
            float* destination = p.out + (tileBase + w) * p.ld + p.colOff + c;
            if (hasEpilogue) {
                value = epilogue(value, destination, p.accumulate, p.divisor);
            }
            *destination = value;
        }
    }

}

struct WaveLds {
    const TableEntry* table;
    const uint32_t* codebook;
    uint32_t* slots;
    uint32_t* keyTile;
};

// LDS layout: lookup table | codebook | per wave { bitstream slots | symbol tile }.
__device__ __forceinline__ WaveLds waveLds(const TrainedParams& p, uint32_t* lds)
{
    const uint32_t wave = threadIdx.x / WAVE;
    const uint32_t perWave = p.wordsPerWave * p.slotDwords + p.keyTileDwords;
    WaveLds result;
    result.table = reinterpret_cast<const TableEntry*>(lds);
    result.codebook = lds + p.tableDwords;
    result.slots = lds + p.tableDwords + p.codebookDwords + wave * perWave;
    result.keyTile = result.slots + p.wordsPerWave * p.slotDwords;
    return result;
}

// Loads table and codebook; ends with a block barrier.
template <int MODE>
__device__ __forceinline__ WaveLds setUpLds(const TrainedParams& p, uint32_t* lds)
{
    uint32_t* codebookLds = lds + p.tableDwords;
    const bool copy = !(measureFlags(p) & 0x4000);   // (measurement builds, bit 14: no table / codebook copy)
    for (uint32_t i = threadIdx.x; copy && i < p.tableDwords / 4; i += blockDim.x) {
        reinterpret_cast<uint4*>(lds)[i] = reinterpret_cast<const uint4*>(p.table)[i];
    }
    if (MODE != OUT_INDEX && MODE != OUT_KEYS) {
        for (uint32_t i = threadIdx.x; copy && i < p.codebookDwords; i += blockDim.x) {
            codebookLds[i] = reinterpret_cast<const uint32_t*>(p.codebook)[i];
        }
    }
    __syncthreads();
    return waveLds(p, lds);
}

// Several batches in one launch (memb_hip_decode_batches_device): the tiles of all batches are numbered through, batch k
// owning the virtual tiles [firstTile[k], firstTile[k + 1]); a wavefront finds its tile's batch by comparing against the
// (wave-uniform, scalar) boundaries and decodes it with that batch's row ids, output, size, ld and col_off. Everything
// else -- the model, the launch geometry, the block's one copy of table and codebook -- is shared: one prologue and one
// tail for all of them.
constexpr uint32_t MAX_BATCHES = MEMB_HIP_MAX_BATCHES;

struct BatchList {
    uint32_t count;
    unsigned long long firstTile[MAX_BATCHES + 1];
    const uint32_t* rows[MAX_BATCHES];
    float* out[MAX_BATCHES];
    unsigned long long n[MAX_BATCHES];
    unsigned long long ld[MAX_BATCHES];
    unsigned long long colOff[MAX_BATCHES];
};

// `p` with the fields of the batch that owns virtual tile *tile (a wave-uniform number); *tile becomes the tile's number
// inside that batch.
__device__ __forceinline__ TrainedParams batchOfTile(const TrainedParams& p, const BatchList& list, unsigned long long* tile)
{
    uint32_t batch = 0;
    for (uint32_t k = 1; k < list.count; ++k) {
        batch += *tile >= list.firstTile[k] ? 1u : 0u;
    }
    TrainedParams q = p;
    q.rows = list.rows[batch];
    q.out = list.out[batch];
    q.n = list.n[batch];
    q.ld = list.ld[batch];
    q.colOff = list.colOff[batch];
    *tile -= list.firstTile[batch];
    return q;
}

// What order do the rows of this batch come in? Sixty-four pairs of neighbouring row ids, spread over the batch, looked at by
// the first wavefront of the grid: 1 = at least three quarters of the pairs are consecutive rows (a key-order dump, or runs of
// one), 0 = not. The answer goes to pinned host memory, where the NEXT launch of a very large batch reads it when it picks
// its block size (memb_hip.hip: launchTrained) -- key-order dumps and shuffled batches want different ones, the caller of
// the reference's API (src/reader.cpp:49-57) has no way to say which it brings, and the batches of one caller tend to look
// like the batch before. A hint to the launch geometry, never to the result; two launches of one context on two streams
// may both write it (a plain store of 0 or 1).
__device__ __forceinline__ void noteBatchOrder(const TrainedParams& p, uint32_t lane)
{
    uint32_t* seen = reinterpret_cast<uint32_t*>(p.segmentIndexOut);
    uint32_t consecutive = 1;
    if (p.rows && p.n >= 2) {
        const unsigned long long at = (p.n - 2) / (WAVE - 1) * lane;   // <= n - 2
        consecutive = p.rows[at + 1] == p.rows[at] + 1 ? 1u : 0u;
    }
    const uint32_t pairs = __popcll(__ballot(consecutive != 0));
    if (lane == 0) {
        __hip_atomic_store(seen, pairs >= 48 ? 1u : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// One tile per wavefront at a time, no software pipeline: the hardware's dispatch of short-lived blocks is what
// keeps the memory system busy. Also builds the segment index (OUT_INDEX).
// A wavefront decodes p.tilesPerWave tiles one after the other -- tiles wave, wave + W, ... of its block's run of
// W * tilesPerWave tiles (W = wavefronts per block), so at every step the block's wavefronts write W adjacent tiles --
// and the block copies table and codebook into LDS once for all of them; the row ids of the next tile are loaded while
// this one is decoded.
// BATCHES: the tiles are virtual tiles of a BatchList (decode_trained_batches); otherwise `list` is not looked at.
template <bool HAS_SUB, int MODE, bool FAST, bool BATCHES>
__device__ __forceinline__ void decodeTilesOfBlock(const TrainedParams& p, const BatchList& list, uint32_t* lds)
{
    // byte keys: 4-byte table entries; the index pass keeps the 8-byte table. (Nibble keys through the 4-byte entries here
    // as well -- round 5, batch 5, two builds alternating: 4-bit dumps -0.1..-0.2 %, shuffled +0.1 %, 100 000 cached rows
    // -3 %, uncached +-0, 2-bit dump +0.9 %: not taken; decode_union_split, whose decode is NOT hidden, takes them.)
    constexpr bool PACKED = !FAST && MODE != OUT_INDEX;
    uint32_t lane = threadIdx.x & (WAVE - 1);
    const uint32_t wavesPerBlock = blockDim.x / WAVE;
    // (Round 4, batch 4: giving every XCD one contiguous run of the batch instead of every eighth block: +3.2 % on the
    // key-order dump, +-0 shuffled -- blocks stay dealt out as the dispatcher deals them.)
    const uint32_t wave = BATCHES ? __builtin_amdgcn_readfirstlane(threadIdx.x / WAVE) : threadIdx.x / WAVE;
    unsigned long long tile = static_cast<unsigned long long>(blockIdx.x) * (wavesPerBlock * p.tilesPerWave) + wave;
    const uint32_t rounds = (p.wordsPerWave * p.loadPieces + WAVE - 1) / WAVE;
    // The copy of table and codebook stays IN FRONT of the tile's loads. Round 4, batch 10 issued the row ids first, the
    // copy's pieces behind them into a register, the row regions behind those, and wrote the pieces to LDS (with the
    // block's barrier) while the regions were in flight -- every wait for exactly what it needs, loads returning in order.
    // Against this form, three builds alternating on one box: key-order dumps +1.4..+2.3 % (4-bit), +1.6 % (6-bit), 500 k
    // rows +1.2 %; shuffled 2.2 M rows -1..-2.8 % (4-bit), -4 % (6-bit); 10 000 rows -2 %. The same as straight-line code
    // without the T loop: 10 000 rows -5 %, dumps +6..+10 %. The BASELINE configurations that run this kernel are dumps:
    // a block that starts its dependent loads a microsecond later is the better citizen there (round 3 had seen the same
    // sign with a cruder ordering).
    const WaveLds mem = setUpLds<MODE>(p, lds);
#ifndef MEMB_HIP_NO_ORDER_PROBE   // (two builds side by side: tools/perf/r6/probe_cost.sh)
    if (!BATCHES && MODE != OUT_INDEX && blockIdx.x == 0 && threadIdx.x < WAVE && p.segmentIndexOut) {   // (one wavefront of the grid)
        noteBatchOrder(p, lane);
    }
#endif
    const unsigned long long tiles = BATCHES ? list.firstTile[list.count] : (p.n + p.wordsPerWave - 1) / p.wordsPerWave;
    if (tile >= tiles) {
        return;
    }
    // (the tile of a wavefront: `q` = the parameters it is decoded with, `local` = its number inside its batch)
    unsigned long long local = tile;
    TrainedParams q = BATCHES ? batchOfTile(p, list, &local) : p;
    uint32_t tileRow = loadTileRow(q, local, laneRole(q, lane));
#pragma nounroll
    for (uint32_t step = 0; step < p.tilesPerWave; ++step, tile += wavesPerBlock) {
        if (tile >= tiles) {
            break;
        }
        if (BATCHES && step) {
            local = tile;
            q = batchOfTile(p, list, &local);
        } else if (!BATCHES) {
            local = tile;
        }
        const unsigned long long tileBase = local * q.wordsPerWave;
        // Everything a lane derives from its number -- word, segment, the pieces it copies and stores -- is worked out
        // afresh per tile, as a wavefront with one tile did: hoisted out of the loop those values cost 50 vector
        // registers and half the resident wavefronts (103 against 52: tools/perf/isa.py).
        asm volatile("" : "+v"(lane));
        const LaneRole role = laneRole(q, lane);
        WordMeta meta = loadWordMeta(q, tileRow, role);
        unpackMeta(q, role, meta);
        StreamRegisters first = {};   // (defined on every path: otherwise the values are carried around the loop)
        issueStreamLoads(q, meta, lane, 0, first);
        if (step + 1 < p.tilesPerWave && tile + wavesPerBlock < tiles) {
            if (BATCHES) {
                unsigned long long nextLocal = tile + wavesPerBlock;
                const TrainedParams next = batchOfTile(p, list, &nextLocal);
                tileRow = loadTileRow(next, nextLocal, role);
            } else {
                tileRow = loadTileRow(q, tile + wavesPerBlock, role);
            }
        }
        const uint32_t tileWords =
            static_cast<uint32_t>(min(static_cast<unsigned long long>(q.wordsPerWave), q.n - tileBase));
        writeStreams(q, mem.slots, lane, 0, first);
        for (uint32_t round = STREAM_REGISTERS; round < rounds; round += STREAM_REGISTERS) {
            StreamRegisters v = {};
            issueStreamLoads(q, meta, lane, round, v);
            writeStreams(q, mem.slots, lane, round, v);
        }
        waveLdsFence();

        recordSegmentBits(q, mem.slots, role, meta);
        if (MODE == OUT_INDEX || !(measureFlags(q) & 1)) {   // (measurement builds: 1 = no decode, 2 = no output, 4 = no loads)
            decodeSegment<HAS_SUB, MODE, FAST, PACKED>(q, mem.table, mem.slots, mem.keyTile, role, meta);
        }
        if (MODE != OUT_INDEX) {
            waveLdsFence();
            if (!(measureFlags(q) & 2)) {
                outputTile<MODE, FAST>(q, mem.codebook, mem.keyTile, tileBase, tileWords, lane, role, meta.row < q.nRows);
            }
        }
        waveLdsFence();   // (the next tile's streams and symbols go where this one's were)
    }
}

template <bool HAS_SUB, int MODE, bool FAST>
__global__ MEMB_SGPR_BUDGET void decode_trained(TrainedParams p)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    decodeTilesOfBlock<HAS_SUB, MODE, FAST, false>(p, BatchList(), lds);
}

template <bool HAS_SUB, int MODE, bool FAST>
__global__ MEMB_SGPR_BUDGET void decode_trained_batches(TrainedParams p, BatchList list)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    decodeTilesOfBlock<HAS_SUB, MODE, FAST, true>(p, list, lds);
}

// ---------------------------------------------------------------------------
// decode_records_persistent: a software pipeline for batches of more than one and up to four tiles per resident wavefront
// ---------------------------------------------------------------------------
// The one kernel besides decode_trained that a single model runs (memb_hip.hip: planTrained): batches of 57 000 to
// 131 000 words on 256 CUs -- BASELINE.json configs[1] -- where every wavefront has two or three tiles and a
// wavefront with one tile spends its life in the three dependent hops row id -> row region -> decode. Row-record
// layout only (TrainedParams::recordPieces): a row's address is arithmetic and its segment offsets arrive with its
// bitstream, so a tile in flight is its row ids and two stream registers (a tile of at most 128 pieces).
// Timeline of a round: decode tile t out of LDS | wait for the loads issued one decode ago (tile t + 1's regions,
// tile t + 2's row ids) | store tile t | issue the loads of tile t + 2 (row ids of t + 3). 78 VGPRs: 24 wavefronts
// per CU (nibble keys: 82 and 20 with an output burst of 5, rounds 3-5; MEMB_HIP_RECORDS_BURST_NIBBLE). Measured against one tile per wavefront on 100 000 random rows (round 4, batches 1 and 3, two boxes):
// 4-bit -4.3..-6 %, 6-bit -8..-9 %, 2-bit -1.5 %; everywhere else the one-tile kernel wins or ties.
// (Rounds 1-3 also had a general persistent pipeline for every layout, an LDS-DMA form of this one and a persistent
// union: none of them won a BASELINE configuration by 3 % in round 4's table -- DESIGN.md section 5 -- and they are gone.
// Round 6 built the fastest memory pattern known -- two tiles per wavefront half a batch apart, the second tile's regions in
// flight during the first tile's stores, a grid that is NOT resident -- as a kernel of its own, straight-line, 52 vector and
// 70 scalar registers (decode_two_tiles), and this kernel launched that way had been round 5's option pipeline_tiles:
// key-order dumps +4.0 % (blocks of four) / +8.4 % (of eight) against decode_trained, shuffled -2.1 % where plain blocks of
// four make -3.5 %, 100 000 rows with nothing cached +0.5 % against this kernel, 60 000 rows +4 %. Both are gone;
// profiles/r06_experiments.txt, batch 1.)
constexpr int RECORD_ROUNDS = 2;   // 64-lane rounds per tile: 8 words x (160-byte region + padding piece) = 88 pieces

__device__ __forceinline__ void issueRecordLoads(
    const TrainedParams& p, uint32_t row, uint32_t lane, u32x4& first, u32x4& second)
{
    const uint32_t start = row < p.nRows ? row * p.recordPieces : 0u;   // absent words read row 0 and never emit it
    issueStreamLoad(p, start, lane, 0, first);
    issueStreamLoad(p, start, lane, 1, second);
}

template <bool HAS_SUB, int MODE, bool FAST>
__global__ MEMB_RECORDS_WAVES void decode_records_persistent(TrainedParams p)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    constexpr bool PACKED = !FAST;
    const uint32_t lane = threadIdx.x & (WAVE - 1);
    const unsigned long long stride = static_cast<unsigned long long>(gridDim.x) * (blockDim.x / WAVE);
    unsigned long long tile = static_cast<unsigned long long>(blockIdx.x) * (blockDim.x / WAVE) + threadIdx.x / WAVE;
    const unsigned long long tiles = (p.n + p.wordsPerWave - 1) / p.wordsPerWave;
    const LaneRole role = laneRole(p, lane);
    // row ids before the table copy, which hides their latency
    uint32_t rowCurrent = loadTileRow(p, tile, role);
    uint32_t rowNext = loadTileRow(p, tile + stride, role);
    uint32_t rowLoading = loadTileRow(p, tile + 2 * stride, role);
    u32x4 stream0 = {0, 0, 0, 0};                       // the next tile's pieces (a tile of at most 64 pieces
    u32x4 stream1 = {0, 0, 0, 0};                       // never loads the second one)
    // The batches this kernel serves are random lookups of two or three tiles per wavefront: mostly prologue. When two
    // 16-byte pieces per thread cover the copy of table and codebook, the pieces are loaded into registers behind the row
    // ids, the first tile's row regions behind them, and the pieces go to LDS -- with the block's barrier -- while the
    // regions are in flight (as decode_union_split; the single-model dumps of decode_trained are what this does NOT pay for).
    const uint32_t tablePieces = p.tableDwords / 4;
    const uint32_t copyPieces = tablePieces + (MODE != OUT_KEYS ? p.codebookDwords / 4 : 0u);
    WaveLds mem;
    if (copyPieces <= 2 * blockDim.x && !(measureFlags(p) & 0x8000)) {
        mem = waveLds(p, lds);
        u32x4 image0 = {0, 0, 0, 0};
        u32x4 image1 = {0, 0, 0, 0};
        auto imageSource = [&](uint32_t at) {   // (table and codebook are neighbours in LDS: one image)
            return at < tablePieces ? reinterpret_cast<const u32x4*>(p.table) + at
                                    : reinterpret_cast<const u32x4*>(p.codebook) + (at - tablePieces);
        };
        if (!(measureFlags(p) & 0x4000)) {
            if (threadIdx.x < copyPieces) {
                image0 = *imageSource(threadIdx.x);
            }
            if (threadIdx.x + blockDim.x < copyPieces) {
                image1 = *imageSource(threadIdx.x + blockDim.x);
            }
        }
        if (tile < tiles) {
            issueRecordLoads(p, rowCurrent, lane, stream0, stream1);
        }
        if (threadIdx.x < copyPieces) {
            reinterpret_cast<u32x4*>(lds)[threadIdx.x] = image0;
        }
        if (threadIdx.x + blockDim.x < copyPieces) {
            reinterpret_cast<u32x4*>(lds)[threadIdx.x + blockDim.x] = image1;
        }
        __syncthreads();
        if (tile >= tiles) {
            return;
        }
    } else {
        mem = setUpLds<MODE>(p, lds);
        if (tile >= tiles) {
            return;
        }
        issueRecordLoads(p, rowCurrent, lane, stream0, stream1);
    }
    uint32_t* slots = mem.slots;

    // prologue
    writeStream(p, slots, lane, 0, stream0);
    writeStream(p, slots, lane, 1, stream1);
    issueRecordLoads(p, rowNext, lane, stream0, stream1);
    waveLdsFence();

    for (; tile < tiles; tile += stride) {
        const unsigned long long tileBase = tile * p.wordsPerWave;
        const uint32_t tileWords =
            static_cast<uint32_t>(min(static_cast<unsigned long long>(p.wordsPerWave), p.n - tileBase));
        WordMeta meta;
        meta.row = rowCurrent;
        meta.start = 0;
        meta.packed2 = 0;
        meta.packed3 = 0;
        recordSegmentBits(p, slots, role, meta);
        if (!(measureFlags(p) & 1)) {
            decodeSegment<HAS_SUB, MODE, FAST, PACKED>(p, mem.table, slots, mem.keyTile, role, meta);
        }
        waveLdsFence();

        // consume point: everything issued one decode ago. (In-flight registers are touched HERE and nowhere else in a
        // round: a loop-carried copy of a load result is otherwise hoisted to the loop header and drags a vmcnt(0) with it.)
        uint32_t rowAfterNext;
        writeStream(p, slots, lane, 0, stream0);
        writeStream(p, slots, lane, 1, stream1);
        asm volatile("v_mov_b32 %0, %1" : "=v"(rowAfterNext) : "v"(rowLoading));
        __builtin_amdgcn_sched_barrier(0);

        if (!(measureFlags(p) & 2)) {
            outputTile<MODE, FAST, FAST ? MEMB_HIP_RECORDS_BURST_NIBBLE : MEMB_HIP_OUTPUT_BURST>(p, mem.codebook, mem.keyTile, tileBase, tileWords, lane, role, rowCurrent < p.nRows);
        }
        __builtin_amdgcn_sched_barrier(0);

        // the loads of tile t + 2 (its row ids landed a round ago) and the row ids of tile t + 3
        issueRecordLoads(p, rowAfterNext, lane, stream0, stream1);
        rowLoading = loadTileRow(p, tile + 3 * stride, role);
        rowCurrent = rowNext;
        rowNext = rowAfterNext;
        waveLdsFence();
    }
}

// ---------------------------------------------------------------------------
// decode_trained_union: ReadersUnion 'concatenate' in one launch
// ---------------------------------------------------------------------------
// The reference concatenates the readers' results on the host (python/memb/readers_union.py:32).
// Decoding every reader into its column block with a launch of its own makes each launch write
// every other `dim` floats of the merged rows -- half lines, twice, far apart in time (2.7 TB/s
// against 4.4 TB/s for dense rows). Here one wavefront decodes its tile of the batch for ALL
// models (each into a symbol tile of its own) and then writes the merged rows whole.
// One tile per wavefront (union batches are random lookups), COUNT models of the same geometry
// (dim, lanes per word) and key format.
// AVERAGE: the 'average' mode instead (python/memb/readers_union.py:5-18, numpy.mean over the
// readers): the models' vectors are added in reader order and divided by COUNT in registers --
// the operations numpy performs, so the same bits -- and the row is written once, where separate
// launches store, then read, add, divide and store again.
constexpr int UNION_MAX_MODELS = 4;

struct UnionParams {
    TrainedParams model[UNION_MAX_MODELS];   // out / ld shared, colOff per model; n, wordsPerWave etc. equal
    uint32_t tableOffsetDwords[UNION_MAX_MODELS];
    // inside one wavefront's LDS area: where model m's bitstream slots and symbol tile live (decode_trained_union:
    // slots and a tile per model; decode_union_split: one set of slots, one tile, [1] = where model 1's rows begin in it)
    uint32_t slotOffsetDwords[UNION_MAX_MODELS];
    uint32_t keyTileOffsetDwords[UNION_MAX_MODELS];
    uint32_t codebookOffsetDwords;   // model m's codebook at this + m * 512 dwords
    uint32_t sharedDwords;           // tables + codebooks
    uint32_t perWaveDwords;          // one wavefront's area
    uint32_t rowPieces;              // 16-byte pieces of a merged row: COUNT * dim / 4, or dim / 4 when averaging
    uint32_t rowMagic;               // fastDivide magic for rowPieces
};

// Tables and codebooks of all models into LDS; ends with a block barrier.
template <int COUNT>
__device__ __forceinline__ void setUpUnionLds(const UnionParams& u, uint32_t* lds)
{
    const bool copy = !(measureFlags(u.model[0]) & 0x4000);   // (measurement builds, bit 14: no table / codebook copy)
#pragma unroll
    for (int m = 0; m < COUNT; ++m) {
        const TrainedParams& p = u.model[m];
        uint32_t* tableLds = lds + u.tableOffsetDwords[m];
        for (uint32_t i = threadIdx.x; copy && i < p.tableDwords / 4; i += blockDim.x) {
            reinterpret_cast<uint4*>(tableLds)[i] = reinterpret_cast<const uint4*>(p.table)[i];
        }
        uint32_t* codebookLds = lds + u.codebookOffsetDwords + m * 512;
        for (uint32_t i = threadIdx.x; copy && i < p.codebookDwords; i += blockDim.x) {
            codebookLds[i] = reinterpret_cast<const uint32_t*>(p.codebook)[i];
        }
    }
    __syncthreads();
}

// Per model, the ballot of a tile's words the model does not know. Named members and compile-time
// selection: an array indexed through a reference ended up in scratch memory.
struct AbsentMasks {
    unsigned long long m0 = 0, m1 = 0, m2 = 0, m3 = 0;

    // (called from unrolled loops: `model` is a constant by then)
    __device__ __forceinline__ void set(int model, unsigned long long value)
    {
        if (model == 0) m0 = value;
        if (model == 1) m1 = value;
        if (model == 2) m2 = value;
        if (model == 3) m3 = value;
    }

    // Does model `model` lack the word whose first lane is `firstLane`? (Arithmetic, not a chain of
    // selects on `model`: LLVM turns such a chain into a table in scratch memory.)
    template <int COUNT>
    __device__ __forceinline__ bool lacks(uint32_t model, uint32_t firstLane) const
    {
        uint32_t perModel = static_cast<uint32_t>(m0 >> firstLane) & 1u;
        if (COUNT > 1) perModel |= (static_cast<uint32_t>(m1 >> firstLane) & 1u) << 1;
        if (COUNT > 2) perModel |= (static_cast<uint32_t>(m2 >> firstLane) & 1u) << 2;
        if (COUNT > 3) perModel |= (static_cast<uint32_t>(m3 >> firstLane) & 1u) << 3;
        return (perModel >> model) & 1u;
    }
};

// The merged rows of one tile out of the models' symbol tiles: 16 bytes per lane, row contiguous when
// the column blocks are adjacent. absent: per model, the ballot of the tile's words the model does not know
// (nibble keys have no code for "absent").
// The tile is walked as tileWords * COUNT "half rows" of dim / 4 pieces (one model's vector of one word):
// piece q -> half row h = q / (dim / 4), word h / COUNT, model h % COUNT. Everything that depends on the
// model only -- symbol tile, codebook, column -- is one multiply-add away, and whether a half row is absent
// is one bit of a per-tile mask: the round-2 form spent more vector instructions here than in the decode
// (profiles/r03_union_*: 112 VALU instructions per decoded word against 68 in the single-model kernel).
template <bool FAST, int COUNT, bool AVERAGE>
__device__ __forceinline__ void outputUnionTile(
    const UnionParams& u, const uint32_t* lds, const uint32_t* waveLds, unsigned long long tileBase, uint32_t tileWords,
    uint32_t lane, const AbsentMasks& absent)
{
    const TrainedParams& first = u.model[0];
    const uint32_t piecesPerWord = first.dim / 4;
    // bit h of absentHalves: half row h (word h / COUNT of model h % COUNT) is absent. Lane h works its own bit out.
    unsigned long long absentHalves = 0;
    if (FAST) {
        const uint32_t word = lane / COUNT;
        const uint32_t model = lane - word * COUNT;
        absentHalves = __ballot(lane < first.wordsPerWave * COUNT && absent.template lacks<COUNT>(model, word * first.lanesPerWord));
    }
    const uint32_t keyRowUnits = first.keyRowBytes / (FAST ? 2 : 4);   // symbol-tile row in 16-bit (nibble keys) or 32-bit units
    // symbol tiles and column blocks of the models as "model 0 + model * step" where the steps are equal, else by select
    auto keyTileOf = [&](uint32_t model) -> const uint32_t* {
        uint32_t offset = u.keyTileOffsetDwords[0];
#pragma unroll
        for (int i = 1; i < COUNT; ++i) {
            offset = model == static_cast<uint32_t>(i) ? u.keyTileOffsetDwords[i] : offset;
        }
        return waveLds + offset;
    };
    auto readKey = [&](uint32_t model, uint32_t word, uint32_t column) -> uint32_t {
        const uint32_t* keyTile = keyTileOf(model);
        const uint32_t at = word * keyRowUnits + column;
        return FAST ? reinterpret_cast<const uint16_t*>(keyTile)[at] : keyTile[at];
    };
    auto gather = [&](uint32_t model, uint32_t k) -> float4 {
        const uint32_t* codebook = lds + u.codebookOffsetDwords + model * 512;
        if (FAST) {
            const float2 lo = reinterpret_cast<const float2*>(codebook)[k & 0xff];
            const float2 hi = reinterpret_cast<const float2*>(codebook)[k >> 8];
            return make_float4(lo.x, lo.y, hi.x, hi.y);
        }
        const float* centroids = reinterpret_cast<const float*>(codebook);
        return make_float4(centroids[k & 0xff], centroids[(k >> 8) & 0xff], centroids[(k >> 16) & 0xff], centroids[k >> 24]);
    };
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);

    if (AVERAGE) {
        const uint32_t pieces = tileWords * piecesPerWord;
        constexpr int BURST = 2;
        for (uint32_t q0 = lane; q0 < pieces; q0 += WAVE * BURST) {
            uint32_t k[BURST][COUNT];
            uint32_t w[BURST];
            uint32_t c[BURST];
#pragma unroll
            for (int b = 0; b < BURST; ++b) {
                const uint32_t q = min(q0 + WAVE * b, pieces - 1);
                w[b] = fastDivide(q, first.pieceMagic, piecesPerWord);
                c[b] = q - w[b] * piecesPerWord;
#pragma unroll
                for (int i = 0; i < COUNT; ++i) {
                    k[b][i] = readKey(i, w[b], c[b]);
                }
            }
#pragma unroll
            for (int b = 0; b < BURST; ++b) {
                float4 f = gather(0, k[b][0]);
                if (FAST && ((absentHalves >> (w[b] * COUNT)) & 1)) {
                    f = zero;
                }
#pragma unroll
                for (int i = 1; i < COUNT; ++i) {
                    float4 next = gather(i, k[b][i]);
                    if (FAST && ((absentHalves >> (w[b] * COUNT + i)) & 1)) {
                        next = zero;
                    }
                    f.x = addRn(f.x, next.x);
                    f.y = addRn(f.y, next.y);
                    f.z = addRn(f.z, next.z);
                    f.w = addRn(f.w, next.w);
                }
                const float count = static_cast<float>(COUNT);
                f = make_float4(__fdiv_rn(f.x, count), __fdiv_rn(f.y, count), __fdiv_rn(f.z, count), __fdiv_rn(f.w, count));
                if (q0 + WAVE * b < pieces) {
                    *reinterpret_cast<float4*>(first.out + (tileBase + w[b]) * first.ld + first.colOff + 4 * c[b]) = f;
                }
            }
        }
        return;
    }

    const uint32_t pieces = tileWords * COUNT * piecesPerWord;
    float* tileOut = first.out + tileBase * first.ld;
    // (half row, column) of the lane's first piece by one division; each further piece lies 64 pieces on, which is
    // stepHalves half rows and stepColumns columns with at most one carry -- no division per piece
    const uint32_t stepHalves = WAVE / piecesPerWord;
    const uint32_t stepColumns = WAVE - stepHalves * piecesPerWord;
    uint32_t half = fastDivide(lane, first.pieceMagic, piecesPerWord);
    uint32_t column = lane - half * piecesPerWord;
    constexpr int BURST = 4;
    for (uint32_t q0 = lane; q0 < pieces; q0 += WAVE * BURST) {
        uint32_t k[BURST];
        uint32_t h[BURST];
        uint32_t c[BURST];
#pragma unroll
        for (int b = 0; b < BURST; ++b) {
            h[b] = half;
            c[b] = column;
            // (past the tile's end: the last piece once more, never stored)
            const bool inside = q0 + WAVE * b < pieces;
            const uint32_t hh = inside ? half : tileWords * COUNT - 1;
            const uint32_t cc = inside ? column : piecesPerWord - 1;
            k[b] = (measureFlags(first) & 0x2000) ? 0u : readKey(hh % COUNT, hh / COUNT, cc);   // (measurement: no LDS reads)
            column += stepColumns;
            const bool carry = column >= piecesPerWord;
            column -= carry ? piecesPerWord : 0u;
            half += stepHalves + (carry ? 1u : 0u);
        }
#pragma unroll
        for (int b = 0; b < BURST; ++b) {
            const uint32_t word = h[b] / COUNT;
            const uint32_t model = h[b] % COUNT;
            float4 f = (measureFlags(first) & 0x2000) ? make_float4(1.f, 2.f, 3.f, 4.f) : gather(model, k[b]);
            if (FAST && ((absentHalves >> (h[b] & 63)) & 1)) {
                f = zero;
            }
            unsigned long long colOff = u.model[0].colOff;
#pragma unroll
            for (int i = 1; i < COUNT; ++i) {
                colOff = model == static_cast<uint32_t>(i) ? u.model[i].colOff : colOff;
            }
            if (q0 + WAVE * b < pieces) {
                *reinterpret_cast<float4*>(tileOut + word * first.ld + colOff + 4 * c[b]) = f;
            }
        }
    }
}

// One tile per wavefront, the models' decodes one after the other: unions of three or four models and pairs that
// decode_union_split does not take (no row records, row regions too different in size).
template <bool HAS_SUB, bool FAST, int COUNT, bool AVERAGE>
__global__ void decode_trained_union(UnionParams u)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const uint32_t lane = threadIdx.x & (WAVE - 1);
    const uint32_t wave = threadIdx.x / WAVE;
    const TrainedParams& first = u.model[0];
    setUpUnionLds<COUNT>(u, lds);

    const unsigned long long tile = static_cast<unsigned long long>(blockIdx.x) * (blockDim.x / WAVE) + wave;
    const unsigned long long tileBase = tile * first.wordsPerWave;
    if (tileBase >= first.n) {
        return;
    }
    const uint32_t tileWords =
        static_cast<uint32_t>(min(static_cast<unsigned long long>(first.wordsPerWave), first.n - tileBase));
    const LaneRole role = laneRole(first, lane);
    uint32_t* waveLds = lds + u.sharedDwords + wave * u.perWaveDwords;

    // the dependent hops of all models side by side: row ids, then offsets, then bitstreams
    uint32_t rows[COUNT];
    WordMeta meta[COUNT];
#pragma unroll
    for (int m = 0; m < COUNT; ++m) {
        rows[m] = loadTileRow(u.model[m], tile, role);
    }
#pragma unroll
    for (int m = 0; m < COUNT; ++m) {
        meta[m] = loadWordMeta(u.model[m], rows[m], role);
    }
#pragma unroll
    for (int m = 0; m < COUNT; ++m) {
        const TrainedParams& p = u.model[m];
        unpackMeta(p, role, meta[m]);
        uint32_t* slots = waveLds + u.slotOffsetDwords[m];
        const uint32_t rounds = (p.wordsPerWave * p.loadPieces + WAVE - 1) / WAVE;
        for (uint32_t round = 0; round < rounds; round += STREAM_REGISTERS) {
            StreamRegisters v;
            issueStreamLoads(p, meta[m], lane, round, v);
            writeStreams(p, slots, lane, round, v);
        }
    }
    waveLdsFence();

    AbsentMasks absent;
#pragma unroll
    for (int m = 0; m < COUNT; ++m) {
        const TrainedParams& p = u.model[m];
        uint32_t* slots = waveLds + u.slotOffsetDwords[m];
        recordSegmentBits(p, slots, role, meta[m]);
        decodeSegment<HAS_SUB, OUT_VEC4, FAST, !FAST>(
            p, reinterpret_cast<const TableEntry*>(lds + u.tableOffsetDwords[m]), slots, waveLds + u.keyTileOffsetDwords[m], role, meta[m]);
        absent.set(m, __ballot(!(meta[m].row < p.nRows) && !role.spare && role.segment == 0 && role.word < tileWords));
    }
    waveLdsFence();
    outputUnionTile<FAST, COUNT, AVERAGE>(u, lds, waveLds, tileBase, tileWords, lane, absent);
}

// Two models staged as row records: the wavefront's lanes are SPLIT between the models -- the
// lower half of its word slots decodes the tile's words for model 0, the upper half the same words for model 1, in ONE
// pass of the decoder -- so a tile is wordsPerWave / 2 words, its LDS footprint that of the single-model kernel
// (decode_trained_union needs slots and a symbol tile per model, which caps it at 20 wavefronts per CU), and the chain
// of a wavefront is row ids -> regions -> one decode -> merged rows, as short as the single-model one-tile kernel's.
// u.model[2] = the slot geometry of the model with the larger row regions, with nRows = 2^32 - 1 (rows are checked per
// lane against the lane's model here and arrive as MISSING or valid); u.keyTileOffsetDwords[1] = where the upper half's rows begin inside the one symbol tile.
// FAST: nibble keys in both models (8-byte table entries); else byte keys for both (a nibble-key model through its
// byte-key forms), decoded through 4-byte PACKED tables.
// COMPACT (nibble keys only): the models' 4-byte tables (memb_hip_ctx::table32) instead of the 8-byte ones.
template <bool HAS_SUB, bool FAST, bool AVERAGE, bool COMPACT = false>
__global__ MEMB_SGPR_BUDGET void decode_union_split(UnionParams u)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    uint32_t lane = threadIdx.x & (WAVE - 1);
    const uint32_t wave = threadIdx.x / WAVE;
    const uint32_t wavesPerBlock = blockDim.x / WAVE;
    const TrainedParams& both = u.model[2];

    const uint32_t half = both.wordsPerWave / 2;   // words of a tile
    // both.tilesPerWave tiles one after the other (tiles wave, wave + W, ... of the block's run), as decode_trained
    // (measurement builds, flag 0x8000: the wavefront's tiles a GRID apart instead -- round 5, batch 31)
    const bool gridApart = (measureFlags(u.model[0]) & 0x8000) != 0;
    const unsigned long long tileStride = gridApart ? static_cast<unsigned long long>(gridDim.x) * wavesPerBlock : wavesPerBlock;
    unsigned long long tile = gridApart ? static_cast<unsigned long long>(blockIdx.x) * wavesPerBlock + wave
                                        : static_cast<unsigned long long>(blockIdx.x) * (wavesPerBlock * both.tilesPerWave) + wave;
    const bool active = tile * half < both.n;
    // The block's copy of two tables and two codebooks into LDS (8 KiB for two nibble-key models) and the first tile's
    // dependent hops (row ids -> row regions) do not need each other. Union batches are random lookups -- the case in
    // which overlapping the two paid for the single-model kernel (round 4, batch 10: shuffled rows -1..-4 %; its dumps
    // +1.4..2.3 %, which is why decode_trained keeps the copy in front) -- so when two 16-byte pieces per thread cover the
    // copy, the row ids are loaded first, the copy's pieces behind them into registers, the regions behind those, and the
    // pieces go to LDS, with the block's barrier, while the regions are in flight.
    const uint32_t tableDwords = u.model[0].tableDwords + u.model[1].tableDwords;   // tables first, then 512 dwords per codebook
    const uint32_t imagePieces = (tableDwords + 2 * 512) / 4;
    const bool copyInFlight = imagePieces <= 2 * blockDim.x && !(measureFlags(u.model[0]) & 0x8000);
    if (!copyInFlight) {
        setUpUnionLds<2>(u, lds);
        if (!active) {
            return;
        }
    }
    // piece `at` of the LDS image (16 bytes): where it comes from, or null (padding behind a 256-entry codebook)
    auto imageSource = [&](uint32_t at) -> const u32x4* {
        const uint32_t dword = 4 * at;
        if (dword < u.model[0].tableDwords) {
            return reinterpret_cast<const u32x4*>(u.model[0].table + dword);
        }
        if (dword < tableDwords) {
            return reinterpret_cast<const u32x4*>(u.model[1].table + (dword - u.model[0].tableDwords));
        }
        const uint32_t inCodebooks = dword - tableDwords;
        const bool second = inCodebooks >= 512;
        const uint32_t offset = inCodebooks - (second ? 512u : 0u);
        const TrainedParams& model = second ? u.model[1] : u.model[0];
        return at < imagePieces && offset < model.codebookDwords ? reinterpret_cast<const u32x4*>(model.codebook) + offset / 4 : nullptr;
    };
    uint32_t* waveLds = lds + u.sharedDwords + wave * u.perWaveDwords;
    uint32_t* slots = waveLds + u.slotOffsetDwords[0];
    const uint32_t measure = measureFlags(u.model[0]);   // (measurement builds: 1 = no decode, 2 = no output, 4 = no loads)
    const uint32_t totalPieces = both.wordsPerWave * both.loadPieces;

    // the lane's row id of tile `ofTile` as the batch has it (MISSING past the batch end): the load only -- `checked`
    // compares it with the lane's model when the id is used, a tile later, so that nothing waits for it here
    auto loadRow = [&](unsigned long long ofTile, uint32_t ofLane) -> uint32_t {
        const LaneRole role = laneRole(both, ofLane);
        const bool upper = role.word >= half;
        const unsigned long long index = ofTile * half + (role.word - (upper ? half : 0u));
        uint32_t row = MISSING;
        if (!role.spare && index < both.n && !(measure & 4)) {
            const uint32_t* ids = upper ? u.model[1].rows : u.model[0].rows;
            row = ids ? ids[index] : static_cast<uint32_t>(index);
        }
        return row;
    };
    auto checked = [&](uint32_t row, uint32_t ofLane) -> uint32_t {
        const bool upper = laneRole(both, ofLane).word >= half;
        return row < (upper ? u.model[1].nRows : u.model[0].nRows) ? row : MISSING;
    };
    // the row regions of all eight (word, model) pairs of a tile: piece q of the slot image belongs to word slot
    // q / loadPieces (regions of the lane's model; when the models' regions differ in size, `both` has the larger slot
    // geometry and the smaller model's slots take a few pieces of the following row along, as every compact-layout load does)
    auto issueRegions = [&](uint32_t ofRow, uint32_t ofLane, u32x4& first, u32x4& second) {
        const LaneRole role = laneRole(both, ofLane);
        const bool upper = role.word >= half;
        const uint32_t start = ofRow != MISSING ? ofRow * (upper ? u.model[1].recordPieces : u.model[0].recordPieces) : 0u;   // absent words read row 0 and never emit it
#pragma unroll
        for (int round = 0; round < RECORD_ROUNDS; ++round) {
            // (the shuffle with every lane active: lanes past the image's last piece are the source lanes of others)
            const uint32_t q = min(round * WAVE + ofLane, totalPieces - 1);
            const uint32_t slot = fastDivide(q, both.slotMagic, both.loadPieces);
            const uint32_t wordStart = __shfl(start, slot * both.lanesPerWord);
            if (static_cast<uint32_t>(round) * WAVE < totalPieces && !(measure & 4)) {
                const u32x4* base = reinterpret_cast<const u32x4*>(slot >= half ? u.model[1].streams : u.model[0].streams);
                const u32x4 piece = loadPiece(base + (static_cast<unsigned long long>(wordStart) + (q - slot * both.loadPieces)));
                if (round == 0) {
                    first = piece;
                } else {
                    second = piece;
                }
            }
        }
    };
    static_assert(RECORD_ROUNDS == 2, "two stream registers per lane");

    // A pipeline of depth one over the wavefront's T tiles: while tile k is decoded and stored, the row regions of tile
    // k + 1 are in flight (two registers per lane) and the row ids of tile k + 2 behind them.
    uint32_t row = active ? loadRow(tile, lane) : MISSING;
    uint32_t rowNext = active && both.tilesPerWave > 1 ? loadRow(tile + tileStride, lane) : MISSING;
    u32x4 image0 = {0, 0, 0, 0};
    u32x4 image1 = {0, 0, 0, 0};
    if (copyInFlight && !(measure & 0x4000)) {
        const u32x4* source0 = imageSource(threadIdx.x);
        const u32x4* source1 = imageSource(threadIdx.x + blockDim.x);
        if (source0) {
            image0 = *source0;
        }
        if (source1) {
            image1 = *source1;
        }
    }
    u32x4 piece0 = {0, 0, 0, 0};
    u32x4 piece1 = {0, 0, 0, 0};
    if (active) {
        row = checked(row, lane);
        issueRegions(row, lane, piece0, piece1);
    }
    if (copyInFlight) {
        // (the codebook areas are 512 dwords each whatever the codebook's size: padding pieces are written as zeros)
        if (threadIdx.x < imagePieces) {
            reinterpret_cast<u32x4*>(lds)[threadIdx.x] = image0;
        }
        if (threadIdx.x + blockDim.x < imagePieces) {
            reinterpret_cast<u32x4*>(lds)[threadIdx.x + blockDim.x] = image1;
        }
        __syncthreads();
        if (!active) {
            return;
        }
    }
#pragma nounroll
    for (uint32_t step = 0; step < both.tilesPerWave; ++step, tile += tileStride) {
        const unsigned long long tileBase = tile * half;
        if (tileBase >= both.n) {
            break;
        }
        // (what a lane derives from its number is worked out per tile, not kept across the loop: see decode_trained)
        asm volatile("" : "+v"(lane));
        const LaneRole role = laneRole(both, lane);
        const bool upper = role.word >= half;           // the lane's model
        const uint32_t word = role.word - (upper ? half : 0u);
        const uint32_t tileWords = static_cast<uint32_t>(min(static_cast<unsigned long long>(half), both.n - tileBase));
        // consume point of the loads issued a tile ago -- the only place that touches them
        writeStream(both, slots, lane, 0, piece0);
        writeStream(both, slots, lane, 1, piece1);
        const uint32_t rowNow = row;
        waveLdsFence();
        if (step + 1 < both.tilesPerWave) {
            row = checked(rowNext, lane);
            issueRegions(row, lane, piece0, piece1);
            if (step + 2 < both.tilesPerWave) {
                rowNext = loadRow(tile + 2 * tileStride, lane);
            }
        }

        WordMeta meta;
        meta.row = rowNow;
        meta.start = 0;
        meta.packed2 = 0;
        meta.packed3 = 0;
        recordSegmentBits(both, slots, role, meta);
        const uint32_t* table = lds + (upper ? u.tableOffsetDwords[1] : u.tableOffsetDwords[0]);
        if (!(measure & 1)) {
            decodeSegment<HAS_SUB, OUT_VEC4, FAST, !FAST || COMPACT>(
                both, reinterpret_cast<const TableEntry*>(table), slots, waveLds + u.keyTileOffsetDwords[0], role, meta,
                upper ? u.model[1].rootBits : u.model[0].rootBits);
        }
        // per model, bit (word * lanesPerWord): the model lacks the tile's word
        const unsigned long long lacking = __ballot(rowNow == MISSING && !role.spare && role.segment == 0 && word < tileWords);
        AbsentMasks absent;
        const uint32_t upperShift = half * both.lanesPerWord;   // 32 with eight lanes per word
        absent.set(0, upperShift < 64 ? lacking & ((1ull << upperShift) - 1) : lacking);
        absent.set(1, upperShift < 64 ? lacking >> upperShift : 0ull);
        waveLdsFence();
        if (!(measure & 2)) {
            outputUnionTile<FAST, 2, AVERAGE>(u, lds, waveLds, tileBase, tileWords, lane, absent);
        }
        waveLdsFence();   // (the next tile's regions and symbols go where this one's were)
    }
}

// Staging: streamStarts + segmentIndex -> rowMeta records (see TrainedParams::rowMeta). One thread per row.
// With recordPieces != 0 the record goes to piece row * recordPieces of `rowMeta` (= the row-record
// array) and its first dword holds the row's stream length in bytes instead of a start.
__global__ void pack_row_meta(
    const uint32_t* streamStarts, const uint16_t* segmentIndex, uint32_t lanesPerWord, unsigned long long nRows,
    uint32_t* rowMeta, uint32_t recordPieces, const uint32_t* streamBytes)
{
    const unsigned long long row = static_cast<unsigned long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (row >= nRows) {
        return;
    }
    uint32_t offsets[3] = {0, 0, 0};
    for (uint32_t segment = 1; segment < lanesPerWord; ++segment) {
        const unsigned long long value = segmentIndex[row * (lanesPerWord - 1) + segment - 1];
        const uint32_t bit = ROW_META_BITS * (segment - 1);
        const unsigned long long shifted = value << (bit & 31);
        offsets[bit >> 5] |= static_cast<uint32_t>(shifted);
        if ((bit >> 5) + 1 < 3) {
            offsets[(bit >> 5) + 1] |= static_cast<uint32_t>(shifted >> 32);
        }
    }
    if (recordPieces) {
        reinterpret_cast<uint4*>(rowMeta)[row * recordPieces] = make_uint4(streamBytes[row], offsets[0], offsets[1], offsets[2]);
    } else {
        reinterpret_cast<uint4*>(rowMeta)[row] = make_uint4(streamStarts[row], offsets[0], offsets[1], offsets[2]);
    }
}

// Staging-time re-pack of the file's bitstreams (byte aligned, insertion order,
// reference src/trained_compression.cpp:65-71) into the layout the decoder
// reads: row r's stream starts at piece streamStarts[r], 16-byte aligned, in
// row (= sorted key) order, stored as big-endian dwords. One wavefront per row.
__global__ void repack_streams(
    const uint8_t* packed, unsigned long long packedBytes, const uint32_t* valueOffsets, const uint32_t* streamStarts,
    unsigned long long nRows, uint4* streams, uint32_t recordPieces)
{
    const unsigned long long row = (static_cast<unsigned long long>(blockIdx.x) * blockDim.x + threadIdx.x) / WAVE;
    if (row >= nRows) {
        return;
    }
    const uint32_t lane = threadIdx.x & (WAVE - 1);
    const uint32_t first = streamStarts[row];
    // (row records: every row owns recordPieces pieces, the first of them its record)
    const uint32_t pieces = recordPieces ? recordPieces - 1 : streamStarts[row + 1] - first;
    const unsigned long long source = valueOffsets[row];
    for (uint32_t piece = lane; piece < pieces; piece += WAVE) {
        uint32_t dwords[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            uint32_t value = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const unsigned long long at = source + 16ull * piece + 4 * k + b;
                value = (value << 8) | (at < packedBytes ? packed[at] : 0u);
            }
            dwords[k] = value;
        }
        streams[static_cast<unsigned long long>(first) + piece] = make_uint4(dwords[0], dwords[1], dwords[2], dwords[3]);
    }
}

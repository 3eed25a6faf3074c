// Host half of the write side on the device (memb_hip_encoder_*): staging of vector blocks to HBM,
// the symbol array, launches of hip_encoder_kernels.h. Included by memb_hip.hip only.
#pragma once

struct memb_hip_encoder {
    int device = 0;
    uint32_t dim = 0;
    uint32_t cuCount = 0;
    uint32_t ldsLimit = 0;
    hipStream_t stream = nullptr;
    float* splits = nullptr;
    uint32_t splitCount = 0;
    unsigned long long* counts = nullptr;   // [256]
    uint8_t* symbols = nullptr;             // [capacityRows][dim]
    uint64_t rows = 0;
    uint64_t capacityRows = 0;
    // blocks cross PCIe through two pinned buffers, the copy of one overlapping the host's filling of the other
    static constexpr size_t STAGE_BYTES = size_t(32) << 20;
    void* pinned[2] = {};
    float* staged[2] = {};
    hipEvent_t stagedFree[2] = {};
    std::unique_ptr<memb::WorkerPool> copyPool;
    // result of pack
    uint8_t* packed = nullptr;
    uint64_t packedBytes = 0;
    bool broken = false;   // a device call failed half way through a block: symbols and histogram no longer agree
    std::mutex mutex;
};

namespace {

void destroyEncoder(memb_hip_encoder* encoder)
{
    if (!encoder) {
        return;
    }
    DeviceRestore restore;
    (void)hipSetDevice(encoder->device);
    if (encoder->stream) {
        (void)hipStreamSynchronize(encoder->stream);
    }
    for (void* pointer : {static_cast<void*>(encoder->splits), static_cast<void*>(encoder->counts),
                          static_cast<void*>(encoder->symbols), static_cast<void*>(encoder->packed),
                          static_cast<void*>(encoder->staged[0]), static_cast<void*>(encoder->staged[1])}) {
        if (pointer) {
            (void)hipFree(pointer);
        }
    }
    for (int i = 0; i < 2; ++i) {
        if (encoder->pinned[i]) {
            (void)hipHostFree(encoder->pinned[i]);
        }
        if (encoder->stagedFree[i]) {
            (void)hipEventDestroy(encoder->stagedFree[i]);
        }
    }
    if (encoder->stream) {
        (void)hipStreamDestroy(encoder->stream);
    }
    delete encoder;
}

struct EncoderGuard {
    explicit EncoderGuard(memb_hip_encoder* encoder): encoder_(encoder) {}
    ~EncoderGuard() { destroyEncoder(encoder_); }
    EncoderGuard(const EncoderGuard&) = delete;
    EncoderGuard& operator=(const EncoderGuard&) = delete;
    memb_hip_encoder* release()
    {
        memb_hip_encoder* encoder = encoder_;
        encoder_ = nullptr;
        return encoder;
    }

private:
    memb_hip_encoder* encoder_;
};

int encoder_create_checked(memb_hip_encoder** out, int device, uint32_t dim, const float* splits, uint32_t nSplits)
{
    if (!out || (nSplits && !splits)) {
        return fail(MEMB_HIP_ERR_INVALID, "null argument");
    }
    *out = nullptr;
    if (dim == 0 || nSplits > 254) {
        return fail(MEMB_HIP_ERR_INVALID, "encoder: dim must be positive and there are at most 254 split points");
    }
    for (uint32_t i = 0; i < nSplits; ++i) {
        if (!(splits[i] == splits[i]) || (i && splits[i] < splits[i - 1])) {
            return fail(MEMB_HIP_ERR_INVALID, "encoder: split points must be sorted numbers");
        }
    }
    DeviceRestore restore;
    memb_hip_encoder* encoder = new memb_hip_encoder();
    EncoderGuard guard(encoder);
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count == 0) {
        (void)hipGetLastError();
        return fail(MEMB_HIP_ERR_DEVICE, "no HIP device available");
    }
    if (device < 0 || device >= count) {
        return fail(MEMB_HIP_ERR_INVALID, "device index out of range");
    }
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t properties;
    HIP_TRY(hipGetDeviceProperties(&properties, device));
    encoder->device = device;
    encoder->dim = dim;
    encoder->cuCount = properties.multiProcessorCount;
    encoder->ldsLimit = properties.maxSharedMemoryPerMultiProcessor >= 160 * 1024
        ? 160 * 1024 : static_cast<uint32_t>(std::min<size_t>(properties.sharedMemPerBlock, 160 * 1024));
    encoder->splitCount = nSplits;
    HIP_TRY(hipStreamCreateWithFlags(&encoder->stream, hipStreamNonBlocking));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&encoder->splits), 256 * sizeof(float)));
    if (nSplits) {
        HIP_TRY(hipMemcpy(encoder->splits, splits, nSplits * sizeof(float), hipMemcpyHostToDevice));
    }
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&encoder->counts), 256 * sizeof(unsigned long long)));
    HIP_TRY(hipMemset(encoder->counts, 0, 256 * sizeof(unsigned long long)));
    for (int i = 0; i < 2; ++i) {
        HIP_TRY(hipHostMalloc(&encoder->pinned[i], memb_hip_encoder::STAGE_BYTES, hipHostMallocDefault));
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&encoder->staged[i]), memb_hip_encoder::STAGE_BYTES));
        HIP_TRY(hipEventCreateWithFlags(&encoder->stagedFree[i], hipEventDisableTiming));
    }
    encoder->copyPool.reset(new memb::WorkerPool(std::min<size_t>(8, std::max<size_t>(1, std::thread::hardware_concurrency() / 2))));
    *out = guard.release();
    return MEMB_HIP_OK;
}

int growSymbols(memb_hip_encoder* encoder, uint64_t rowsNeeded)
{
    if (rowsNeeded <= encoder->capacityRows) {
        return MEMB_HIP_OK;
    }
    const uint64_t capacity = std::max<uint64_t>({rowsNeeded, encoder->capacityRows * 2, (uint64_t(64) << 20) / encoder->dim + 1});
    uint8_t* grown = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&grown), capacity * encoder->dim + 16));
    if (encoder->rows) {
        hipError_t status = hipMemcpyAsync(grown, encoder->symbols, encoder->rows * encoder->dim, hipMemcpyDeviceToDevice, encoder->stream);
        if (status == hipSuccess) {
            status = hipStreamSynchronize(encoder->stream);
        }
        if (status != hipSuccess) {
            (void)hipFree(grown);
            return fail(MEMB_HIP_ERR_DEVICE, std::string("encoder: growing the symbol array: ") + hipGetErrorString(status));
        }
    }
    if (encoder->symbols) {
        (void)hipFree(encoder->symbols);
    }
    encoder->symbols = grown;
    encoder->capacityRows = capacity;
    return MEMB_HIP_OK;
}

// Rows (host memory, row-major, dim floats each) -> symbols appended to the device array, histogram updated.
// The block crosses PCIe in STAGE_BYTES pieces through the two pinned buffers: pooled host threads fill one
// while the copy engine and the kernel work on the other.
int encoder_add_rows_checked(memb_hip_encoder* encoder, const float* rows, size_t nRows)
{
    if (!encoder || (nRows && !rows)) {
        return fail(MEMB_HIP_ERR_INVALID, "null argument");
    }
    if (nRows == 0) {
        return MEMB_HIP_OK;
    }
    std::lock_guard<std::mutex> lock(encoder->mutex);
    if (encoder->packed) {
        return fail(MEMB_HIP_ERR_INVALID, "encoder: rows cannot be added after pack");
    }
    if (encoder->broken) {
        return fail(MEMB_HIP_ERR_DEVICE, "encoder: an earlier block failed on the device; the encoder cannot be used further");
    }
    DeviceScope deviceScope(encoder->device);
    HIP_TRY(deviceScope.status());
    int code = growSymbols(encoder, encoder->rows + nRows);
    if (code != MEMB_HIP_OK) {
        return code;
    }
    const size_t dim = encoder->dim;
    // whole rows per piece, and a multiple of four scalars whenever possible (the kernel's float4 path)
    size_t rowsPerPiece = std::max<size_t>(1, memb_hip_encoder::STAGE_BYTES / (dim * sizeof(float)));
    if (rowsPerPiece > 4) {
        rowsPerPiece = rowsPerPiece / 4 * 4;
    }
    if (dim * sizeof(float) > memb_hip_encoder::STAGE_BYTES) {
        return fail(MEMB_HIP_ERR_INVALID, "encoder: a row does not fit the staging buffer");
    }
    uint32_t firstStep = 1;
    while (firstStep * 2 <= std::max<uint32_t>(encoder->splitCount, 1)) {
        firstStep *= 2;
    }
    // (from here on a failure leaves part of the block counted: the flag is cleared when the whole block is enqueued)
    encoder->broken = true;
    int buffer = 0;
    for (size_t start = 0; start < nRows; start += rowsPerPiece, buffer ^= 1) {
        const size_t pieceRows = std::min(rowsPerPiece, nRows - start);
        const size_t bytes = pieceRows * dim * sizeof(float);
        HIP_TRY(hipEventSynchronize(encoder->stagedFree[buffer]));   // the copy that last read this pinned buffer is done
        const char* source = reinterpret_cast<const char*>(rows + start * dim);
        char* pinned = static_cast<char*>(encoder->pinned[buffer]);
        const size_t jobs = std::min<size_t>(encoder->copyPool->size() + 1, bytes / (size_t(1) << 20) + 1);
        const size_t perJob = (bytes + jobs - 1) / jobs;
        encoder->copyPool->run(jobs, [&](size_t job) {
            const size_t first = std::min(bytes, job * perJob);
            const size_t last = std::min(bytes, first + perJob);
            std::memcpy(pinned + first, source + first, last - first);
        });
        HIP_TRY(hipMemcpyAsync(encoder->staged[buffer], pinned, bytes, hipMemcpyHostToDevice, encoder->stream));
        HIP_TRY(hipEventRecord(encoder->stagedFree[buffer], encoder->stream));
        QuantiseParams params{};
        params.values = encoder->staged[buffer];
        params.symbols = encoder->symbols + (encoder->rows + start) * dim;
        params.count = pieceRows * dim;
        params.splits = encoder->splits;
        params.splitCount = encoder->splitCount;
        params.firstStep = firstStep;
        params.counts = encoder->counts;
        const uint32_t blocks = static_cast<uint32_t>(std::min<uint64_t>(
            uint64_t(encoder->cuCount) * 8, (params.count / 4 + ENCODER_THREADS - 1) / ENCODER_THREADS + 1));
        const bool vec = reinterpret_cast<uintptr_t>(params.symbols) % 4 == 0;
        if (vec) {
            hipLaunchKernelGGL(quantise_rows<true>, dim3(blocks), dim3(ENCODER_THREADS), 0, encoder->stream, params);
        } else {
            hipLaunchKernelGGL(quantise_rows<false>, dim3(blocks), dim3(ENCODER_THREADS), 0, encoder->stream, params);
        }
        HIP_TRY(hipGetLastError());
        // (the kernel reads staged[buffer]; the next copy into it is enqueued on the same stream, behind the kernel)
    }
    encoder->rows += nRows;
    encoder->broken = false;
    return MEMB_HIP_OK;
}

int encoder_counts_checked(memb_hip_encoder* encoder, uint64_t* counts)
{
    if (!encoder || !counts) {
        return fail(MEMB_HIP_ERR_INVALID, "null argument");
    }
    std::lock_guard<std::mutex> lock(encoder->mutex);
    DeviceScope deviceScope(encoder->device);
    HIP_TRY(deviceScope.status());
    static_assert(sizeof(unsigned long long) == sizeof(uint64_t), "64-bit counters");
    HIP_TRY(hipMemcpyAsync(counts, encoder->counts, 256 * sizeof(uint64_t), hipMemcpyDeviceToHost, encoder->stream));
    HIP_TRY(hipStreamSynchronize(encoder->stream));
    return MEMB_HIP_OK;
}

int encoder_rows_checked(memb_hip_encoder* encoder, uint64_t* rows)
{
    if (!encoder || !rows) {
        return fail(MEMB_HIP_ERR_INVALID, "null argument");
    }
    std::lock_guard<std::mutex> lock(encoder->mutex);
    if (encoder->broken) {
        return fail(MEMB_HIP_ERR_DEVICE, "encoder: an earlier block failed on the device; the encoder cannot be used further");
    }
    *rows = encoder->rows;
    return MEMB_HIP_OK;
}

// Code table in, per-word stream lengths out; the packed streams stay on the device until fetched.
int encoder_pack_checked(
    memb_hip_encoder* encoder, const uint16_t* codes, const uint8_t* lengths, uint32_t* streamBytes, uint64_t* totalBytes)
{
    if (!encoder || !codes || !lengths || !totalBytes || (encoder->rows && !streamBytes)) {
        return fail(MEMB_HIP_ERR_INVALID, "null argument");
    }
    std::lock_guard<std::mutex> lock(encoder->mutex);
    if (encoder->broken) {
        return fail(MEMB_HIP_ERR_DEVICE, "encoder: an earlier block failed on the device; the encoder cannot be used further");
    }
    DeviceScope deviceScope(encoder->device);
    HIP_TRY(deviceScope.status());
    uint32_t longest = 0;
    std::vector<uint32_t> table(256, 0);
    for (int symbol = 0; symbol < 256; ++symbol) {
        if (lengths[symbol] > 16) {
            return fail(MEMB_HIP_ERR_INVALID, "encoder: codes have at most 16 bits");
        }
        // the low `length` bits of the code, as BitStream::push takes them (reference src/bit_stream.h:29; its
        // known answer pushes the value 7 with a length of 2: src/bit_stream_tests.cpp:35-41) and as the host writer does
        const uint32_t code = codes[symbol] & ((1u << lengths[symbol]) - 1u);
        table[symbol] = code | (static_cast<uint32_t>(lengths[symbol]) << 16);
        longest = std::max<uint32_t>(longest, lengths[symbol]);
    }
    *totalBytes = 0;
    if (encoder->packed) {
        (void)hipFree(encoder->packed);
        encoder->packed = nullptr;
        encoder->packedBytes = 0;
    }
    const uint64_t rows = encoder->rows;
    if (rows == 0) {
        return MEMB_HIP_OK;
    }
    // LDS image of one word: the longest stream the code allows plus the dword a straddling code spills into
    const uint64_t imageDwords = (uint64_t(encoder->dim) * longest + 31) / 32 + 2;
    uint32_t waves = ENCODER_THREADS / WAVE;
    while (waves > 1 && 4 * (256 + waves * imageDwords) > encoder->ldsLimit) {
        waves /= 2;
    }
    if (4 * (256 + waves * imageDwords) > encoder->ldsLimit) {
        return fail(MEMB_HIP_ERR_INVALID, "encoder: a word's bitstream does not fit into LDS");
    }
    uint32_t* deviceTable = nullptr;
    uint32_t* deviceLengths = nullptr;
    unsigned long long* deviceOffsets = nullptr;
    auto release = [&] {
        for (void* pointer : {static_cast<void*>(deviceTable), static_cast<void*>(deviceLengths), static_cast<void*>(deviceOffsets)}) {
            if (pointer) {
                (void)hipFree(pointer);
            }
        }
        deviceTable = nullptr;
        deviceLengths = nullptr;
        deviceOffsets = nullptr;
    };
    auto check = [&](hipError_t status, const char* what) {
        if (status != hipSuccess) {
            // (copies out of `table` / `offsets` below may still be reading those host vectors)
            (void)hipStreamSynchronize(encoder->stream);
            release();
            return fail(MEMB_HIP_ERR_DEVICE, std::string("encoder: ") + what + ": " + hipGetErrorString(status));
        }
        return MEMB_HIP_OK;
    };
    int code = check(hipMalloc(reinterpret_cast<void**>(&deviceTable), 256 * 4), "hipMalloc");
    if (code == MEMB_HIP_OK) {
        code = check(hipMalloc(reinterpret_cast<void**>(&deviceLengths), rows * 4), "hipMalloc");
    }
    if (code == MEMB_HIP_OK) {
        code = check(hipMalloc(reinterpret_cast<void**>(&deviceOffsets), rows * 8), "hipMalloc");
    }
    if (code == MEMB_HIP_OK) {
        code = check(hipMemcpyAsync(deviceTable, table.data(), 256 * 4, hipMemcpyHostToDevice, encoder->stream), "code table copy");
    }
    if (code != MEMB_HIP_OK) {
        return code;
    }
    PackParams params{};
    params.symbols = encoder->symbols;
    params.nRows = rows;
    params.dim = encoder->dim;
    params.symbolsPerLane = (encoder->dim + WAVE - 1) / WAVE;
    params.codes = deviceTable;
    params.streamBytes = deviceLengths;
    params.slotDwords = static_cast<uint32_t>(imageDwords);
    const uint32_t threads = waves * WAVE;
    const uint32_t blocks = static_cast<uint32_t>(std::min<uint64_t>(uint64_t(encoder->cuCount) * 16, (rows + waves - 1) / waves));
    hipLaunchKernelGGL(stream_lengths, dim3(blocks), dim3(threads), 0, encoder->stream, params);
    code = check(hipGetLastError(), "stream_lengths launch");
    if (code == MEMB_HIP_OK) {
        code = check(hipMemcpyAsync(streamBytes, deviceLengths, rows * 4, hipMemcpyDeviceToHost, encoder->stream), "stream length copy");
    }
    if (code == MEMB_HIP_OK) {
        code = check(hipStreamSynchronize(encoder->stream), "stream_lengths");
    }
    if (code != MEMB_HIP_OK) {
        return code;
    }
    // streams are laid out back to back in insertion order (reference src/trained_compression.cpp:65-71)
    std::vector<unsigned long long> offsets(rows);
    uint64_t total = 0;
    for (uint64_t row = 0; row < rows; ++row) {
        offsets[row] = total;
        total += streamBytes[row];
    }
    code = check(hipMalloc(reinterpret_cast<void**>(&encoder->packed), std::max<uint64_t>(total, 16)), "hipMalloc");
    if (code == MEMB_HIP_OK) {
        code = check(hipMemcpyAsync(deviceOffsets, offsets.data(), rows * 8, hipMemcpyHostToDevice, encoder->stream), "offset copy");
    }
    if (code == MEMB_HIP_OK) {
        params.streamOffsets = deviceOffsets;
        params.packed = encoder->packed;
        hipError_t status = hipFuncSetAttribute(
            reinterpret_cast<const void*>(&pack_streams), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (status == hipSuccess) {
            hipLaunchKernelGGL(
                pack_streams, dim3(blocks), dim3(threads), 4 * (256 + waves * static_cast<uint32_t>(imageDwords)), encoder->stream, params);
            status = hipGetLastError();
        }
        code = check(status, "pack_streams launch");
    }
    if (code == MEMB_HIP_OK) {
        code = check(hipStreamSynchronize(encoder->stream), "pack_streams");
    }
    if (code != MEMB_HIP_OK) {
        if (encoder->packed) {
            (void)hipFree(encoder->packed);
            encoder->packed = nullptr;
        }
        return code;
    }
    release();
    encoder->packedBytes = total;
    *totalBytes = total;
    return MEMB_HIP_OK;
}

int encoder_fetch_checked(memb_hip_encoder* encoder, uint8_t* packed, uint64_t capacity)
{
    if (!encoder || (capacity && !packed)) {
        return fail(MEMB_HIP_ERR_INVALID, "null argument");
    }
    std::lock_guard<std::mutex> lock(encoder->mutex);
    if (!encoder->packed && encoder->rows) {
        return fail(MEMB_HIP_ERR_INVALID, "encoder: nothing packed yet");
    }
    if (capacity < encoder->packedBytes) {
        return fail(MEMB_HIP_ERR_INVALID, "encoder: the buffer is smaller than the packed streams");
    }
    if (encoder->packedBytes == 0) {
        return MEMB_HIP_OK;
    }
    DeviceScope deviceScope(encoder->device);
    HIP_TRY(deviceScope.status());
    HIP_TRY(hipMemcpy(packed, encoder->packed, encoder->packedBytes, hipMemcpyDeviceToHost));
    return MEMB_HIP_OK;
}

}  // namespace

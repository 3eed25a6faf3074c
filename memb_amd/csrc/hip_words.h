// Word -> row on the device, host half: staging of a model's keys (memb_hip_ctx_stage_words), batches of query
// words (memb_hip_words_*: gather into pinned memory on pooled threads, copy engine, HBM), the lookup launch.
//
// Host code of libmemb_hip.so. Included by memb_hip.hip only, after the context; device code in
// hip_words_kernels.h.
#pragma once

struct memb_hip_words {
    int device = 0;
    size_t count = 0;                 // words of the committed batch
    memb_hip_words_plan plan{};       // the batch in the making / the committed one
    uint32_t jobShift = 0;            // log2(plan.job_words)
    bool committed = false;
    // pinned, device-mapped host memory (the kernel reads it in place) and the same memory as the device sees it
    uint8_t* hostBytes = nullptr;
    size_t hostBytesCapacity = 0;
    const uint8_t* deviceBytes = nullptr;
    uint32_t* hostOffsets = nullptr;
    size_t hostOffsetsCapacity = 0;   // entries
    const uint32_t* deviceOffsets = nullptr;
    hipEvent_t lastUse = nullptr;     // behind EVERY lookup that reads the buffers (noteUse chains streams)
    hipStream_t lastStream = nullptr; // the stream lastUse was recorded on
    bool inUse = false;
    uint32_t threads = 64;            // MEMB_HIP_PACK_THREADS: threads of memb_hip_words_pack (at most; one per 32 768 words, at least sixteen)
    std::unique_ptr<memb::WorkerPool> pool;
};

namespace {

void destroyWords(memb_hip_words* batch)
{
    if (!batch) {
        return;
    }
    DeviceRestore restore;
    (void)hipSetDevice(batch->device);
    if (batch->lastUse) {
        if (batch->inUse) {
            (void)hipEventSynchronize(batch->lastUse);
        }
        (void)hipEventDestroy(batch->lastUse);
    }
    batch->pool.reset();
    if (batch->hostOffsets) {
        (void)hipHostFree(batch->hostOffsets);
    }
    if (batch->hostBytes) {
        (void)hipHostFree(batch->hostBytes);
    }
    delete batch;
}

int words_create_checked(memb_hip_words** out, int device)
{
    if (!out) {
        return fail(MEMB_HIP_ERR_INVALID, "null argument");
    }
    *out = nullptr;
    int count = 0;
    hipError_t status = hipGetDeviceCount(&count);
    if (status != hipSuccess || count == 0) {
        (void)hipGetLastError();
        return fail(MEMB_HIP_ERR_DEVICE, "no HIP device available");
    }
    if (device < 0 || device >= count) {
        return fail(MEMB_HIP_ERR_INVALID, "device index out of range");
    }
    DeviceScope deviceScope(device);
    HIP_TRY(deviceScope.status());
    std::unique_ptr<memb_hip_words> batch(new memb_hip_words());
    batch->device = device;
    batch->threads = std::max<uint32_t>(1, std::min<uint32_t>(envUint("MEMB_HIP_PACK_THREADS", 64), 128));
    batch->threads = std::min<uint32_t>(batch->threads, std::max(1u, std::thread::hardware_concurrency()));
    HIP_TRY(hipEventCreateWithFlags(&batch->lastUse, hipEventDisableTiming));
    *out = batch.release();
    return MEMB_HIP_OK;
}

// A lookup on `stream` reads the batch's pinned buffers: lastUse must come to lie behind ALL lookups in flight, not only
// the latest one. A lookup on another stream than the one before first makes its stream wait for the earlier record, so
// the one event words_begin / destroyWords wait for completes after every reader on every stream.
int noteUse(memb_hip_words* batch, hipStream_t stream)
{
    if (batch->inUse && batch->lastStream != stream) {
        HIP_TRY(hipStreamWaitEvent(stream, batch->lastUse, 0));
    }
    HIP_TRY(hipEventRecord(batch->lastUse, stream));
    batch->lastStream = stream;
    batch->inUse = true;
    return MEMB_HIP_OK;
}

// Grows a pinned, device-mapped buffer (contents are not kept: every batch starts afresh). The old buffer is released
// only once the new one exists: a failed allocation leaves the object as it was.
template <typename T>
int growPinned(T** buffer, const T** deviceView, size_t* capacity, size_t wanted)
{
    if (*capacity >= wanted) {
        return MEMB_HIP_OK;
    }
    const size_t entries = std::max(wanted + wanted / 4, size_t(4096));   // (a quarter of headroom: batches of one loop vary a little)
    void* raw = nullptr;
    // (coherent: host threads write a batch, the kernel of the next launch reads it over PCIe -- nothing of it may sit in a
    // device cache from the batch before)
    HIP_TRY(hipHostMalloc(&raw, entries * sizeof(T), hipHostMallocMapped | hipHostMallocPortable | hipHostMallocCoherent));
    void* device = nullptr;
    hipError_t status = hipHostGetDevicePointer(&device, raw, 0);
    if (status != hipSuccess) {
        (void)hipHostFree(raw);
        return fail(MEMB_HIP_ERR_DEVICE, std::string("hipHostGetDevicePointer: ") + hipGetErrorString(status));
    }
    if (*buffer) {
        (void)hipHostFree(*buffer);
    }
    *buffer = static_cast<T*>(raw);
    *deviceView = static_cast<const T*>(device);
    *capacity = entries;
    return MEMB_HIP_OK;
}

// Words per job: a power of two, at least 64 (a wavefront's 64 words never straddle two jobs), small enough that a
// large batch gives every thread several jobs (filling overlaps the lookups of finished jobs chunk by chunk), large
// enough that a job is worth handing out.
uint32_t jobShiftFor(size_t n)
{
    uint32_t shift = 6;
    while (shift < 13 && (size_t(1) << shift) * 64 < n) {   // 64 jobs and more: 128 .. 8192 words per job
        ++shift;
    }
    return shift;
}

constexpr size_t DEFAULT_BYTES_PER_WORD = 16;

int words_begin_checked(memb_hip_words* batch, size_t n, size_t bytesPerWord, memb_hip_words_plan* plan)
{
    if (!batch || !plan) {
        return fail(MEMB_HIP_ERR_INVALID, "null argument");
    }
    // (whatever happens below, the batch of the call before is gone: its plan is set again on success only, so that a batch
    // whose begin failed resolves nothing -- a range lookup on stale buffers would be a wild read on the device)
    batch->count = 0;
    batch->committed = false;
    batch->plan = memb_hip_words_plan{};
    if (n >= 0x7FFFFFFFull) {
        return fail(MEMB_HIP_ERR_INVALID, "batch too large");
    }
    DeviceScope deviceScope(batch->device);
    HIP_TRY(deviceScope.status());
    if (batch->inUse) {
        HIP_TRY(hipEventSynchronize(batch->lastUse));   // the previous batch's lookups read these buffers
        batch->inUse = false;
    }
    const uint32_t shift = jobShiftFor(n);
    const size_t jobWords = size_t(1) << shift;
    const size_t jobs = std::max<size_t>(1, (n + jobWords - 1) / jobWords);
    if (bytesPerWord == 0) {
        bytesPerWord = DEFAULT_BYTES_PER_WORD;
    }
    if (bytesPerWord > 0x7FFFFFFFull / jobWords) {
        return fail(MEMB_HIP_ERR_INVALID, "words this long do not fit a batch: look them up in smaller batches");
    }
    const size_t jobBytes = (jobWords * bytesPerWord + 15) / 16 * 16;
    if (jobs * jobBytes >= 0xFFFFFFF0ull) {
        return fail(MEMB_HIP_ERR_INVALID, "the words of one batch must stay below 4 GiB: split the batch");
    }
    int code = growPinned(&batch->hostBytes, &batch->deviceBytes, &batch->hostBytesCapacity, jobs * jobBytes + 16);
    if (code == MEMB_HIP_OK) {
        code = growPinned(&batch->hostOffsets, &batch->deviceOffsets, &batch->hostOffsetsCapacity, jobs * (jobWords + 1));
    }
    if (code != MEMB_HIP_OK) {
        return code;
    }
    batch->jobShift = shift;
    batch->plan.bytes = batch->hostBytes;
    batch->plan.offsets = batch->hostOffsets;
    batch->plan.n = n;
    batch->plan.job_words = jobWords;
    batch->plan.jobs = jobs;
    batch->plan.job_bytes = jobBytes;
    *plan = batch->plan;
    return MEMB_HIP_OK;
}

int words_commit_checked(memb_hip_words* batch)
{
    if (!batch) {
        return fail(MEMB_HIP_ERR_INVALID, "null argument");
    }
    if (!batch->plan.bytes) {
        return fail(MEMB_HIP_ERR_INVALID, "commit without begin");
    }
    batch->count = batch->plan.n;
    batch->committed = true;
    return MEMB_HIP_OK;
}

// begin + fill + commit for C strings, on the object's own pool.
int words_pack_checked(memb_hip_words* batch, const char* const* words, const uint32_t* lengths, size_t n)
{
    if (!batch || (n && !words)) {
        return fail(MEMB_HIP_ERR_INVALID, "null argument");
    }
    size_t bytesPerWord = DEFAULT_BYTES_PER_WORD;
    for (int attempt = 0; attempt < 40; ++attempt) {
        memb_hip_words_plan plan;
        const int code = words_begin_checked(batch, n, bytesPerWord, &plan);
        if (code != MEMB_HIP_OK) {
            return code;
        }
        std::atomic<uint64_t> needed{0};   // the largest job that did not fit, in bytes
        auto fill = [&](size_t job) {
            const size_t first = job * plan.job_words, last = std::min(n, first + plan.job_words);
            uint32_t* offsets = plan.offsets + job * (plan.job_words + 1);
            const uint64_t base = uint64_t(job) * plan.job_bytes;
            uint64_t at = 0;
            bool fits = true;
            for (size_t i = first; i < last; ++i) {
                const size_t length = lengths ? lengths[i] : std::strlen(words[i]);
                if (fits && at + length <= plan.job_bytes) {
                    offsets[i - first] = static_cast<uint32_t>(base + at);
                    std::memcpy(plan.bytes + base + at, words[i], length);
                } else {
                    fits = false;
                }
                at += length;
            }
            if (fits) {
                offsets[last - first] = static_cast<uint32_t>(base + at);
                return;
            }
            uint64_t seen = needed.load(std::memory_order_relaxed);
            while (seen < at && !needed.compare_exchange_weak(seen, at, std::memory_order_relaxed)) {
            }
        };
        if (n < 8192 || batch->threads <= 1) {   // (waking a pool costs as much as packing a few thousand words)
            for (size_t job = 0; job < plan.jobs && n; ++job) {
                fill(job);
            }
            if (!n) {
                plan.offsets[0] = 0;
            }
        } else {
            if (!batch->pool) {
                batch->pool.reset(new memb::WorkerPool(batch->threads - 1));
            }
            batch->pool->run(plan.jobs, fill, std::min<size_t>(batch->threads, std::max<size_t>(16, n / 32768)));
        }
        if (needed.load() == 0) {
            return words_commit_checked(batch);
        }
        // a job of long words: again with regions that hold the longest job seen (and a quarter more)
        bytesPerWord = std::max<size_t>(2 * bytesPerWord, (needed.load() + plan.job_words - 1) / plan.job_words * 5 / 4 + 1);
    }
    return fail(MEMB_HIP_ERR_INVALID, "internal error: the word batch does not converge");
}

int words_count_checked(const memb_hip_words* batch, size_t* n)
{
    if (!batch || !n) {
        return fail(MEMB_HIP_ERR_INVALID, "null argument");
    }
    *n = batch->count;
    return MEMB_HIP_OK;
}

// ---- the model's side ----

int stage_words_checked(
    memb_hip_ctx* ctx, const char* packedWords, uint64_t packedBytes, const uint32_t* wordOffsets, uint64_t nWords)
{
    if (!ctx || (nWords && (!packedWords || !wordOffsets))) {
        return fail(MEMB_HIP_ERR_INVALID, "null argument");
    }
    std::lock_guard<std::mutex> lock(ctx->mutex);
    if (ctx->wordSlots) {
        return MEMB_HIP_OK;
    }
    if (nWords != ctx->nRows) {
        return fail(MEMB_HIP_ERR_INVALID, "stage_words: one key per row of the context is needed");
    }
    if (packedBytes >= 0xFFFFFFF0ull || nWords >= 0x7FFFFFFFull) {
        return fail(MEMB_HIP_ERR_INVALID, "stage_words: keys of 4 GiB and more are not supported");
    }
    if (nWords && (packedBytes == 0 || packedWords[packedBytes - 1] != 0)) {
        return fail(MEMB_HIP_ERR_INVALID, "stage_words: the packed keys must end with a NUL");
    }
    for (uint64_t r = 0; r < nWords; ++r) {
        if (wordOffsets[r] >= packedBytes) {
            return fail(MEMB_HIP_ERR_INVALID, "stage_words: key offset beyond the packed keys");
        }
    }
    DeviceScope deviceScope(ctx->device);
    HIP_TRY(deviceScope.status());
    uint64_t capacity = 16;
    while (capacity < 2 * nWords) {
        capacity *= 2;
    }
    const uint64_t before = ctx->deviceBytes;
    uint8_t* keyBytes = nullptr;
    WordSlot* slots = nullptr;
    uint32_t* offsets = nullptr;    // the build's input only
    uint32_t* inserted = nullptr;
    int code = deviceAlloc(ctx, &keyBytes, packedBytes + 16);
    if (code == MEMB_HIP_OK) {
        code = deviceAlloc(ctx, &slots, capacity * sizeof(WordSlot));
    }
    if (code == MEMB_HIP_OK) {
        hipError_t status = hipMalloc(reinterpret_cast<void**>(&offsets), std::max<size_t>(nWords * 4, 16) + 16);
        if (status != hipSuccess) {
            code = fail(MEMB_HIP_ERR_DEVICE, std::string("hipMalloc: ") + hipGetErrorString(status));
        }
    }
    if (code == MEMB_HIP_OK) {
        inserted = offsets + std::max<size_t>(nWords, 4);
        hipError_t status = hipMemsetAsync(keyBytes + packedBytes, 0, 16, ctx->stream);
        if (status == hipSuccess) {
            status = hipMemsetAsync(slots, 0xFF, capacity * sizeof(WordSlot), ctx->stream);
        }
        if (status == hipSuccess) {
            status = hipMemsetAsync(inserted, 0, 4, ctx->stream);
        }
        if (status == hipSuccess) {
            status = hipStreamSynchronize(ctx->stream);
        }
        if (status != hipSuccess) {
            code = fail(MEMB_HIP_ERR_DEVICE, std::string("hipMemset: ") + hipGetErrorString(status));
        }
    }
    if (code == MEMB_HIP_OK) {
        code = copyToDevice(keyBytes, packedWords, packedBytes);
    }
    if (code == MEMB_HIP_OK) {
        code = copyToDevice(offsets, wordOffsets, nWords * 4);
    }
    uint32_t keysInTable = 0;
    if (code == MEMB_HIP_OK && nWords) {
        WordTableParams params{};
        params.keyBytes = keyBytes;
        params.keyOffsets = offsets;
        params.n = nWords;
        params.keyBytesTotal = packedBytes;
        params.slots = slots;
        params.slotMask = static_cast<uint32_t>(capacity - 1);
        params.inserted = inserted;
        const uint32_t threads = 256;
        hipLaunchKernelGGL(
            build_word_table, dim3(static_cast<uint32_t>((nWords + threads - 1) / threads)), dim3(threads), 0, ctx->stream, params);
        hipError_t status = hipGetLastError();
        if (status == hipSuccess) {
            status = hipMemcpyAsync(&keysInTable, inserted, 4, hipMemcpyDeviceToHost, ctx->stream);
        }
        if (status == hipSuccess) {
            status = hipStreamSynchronize(ctx->stream);
        }
        if (status != hipSuccess) {
            code = fail(MEMB_HIP_ERR_DEVICE, std::string("build_word_table: ") + hipGetErrorString(status));
        }
    }
    if (offsets) {
        (void)hipFree(offsets);
    }
    if (code != MEMB_HIP_OK) {
        deviceRelease(ctx, &slots, capacity * sizeof(WordSlot));   // (a retry starts from nothing)
        deviceRelease(ctx, &keyBytes, packedBytes + 16);
        return code;
    }
    ctx->wordKeyBytes = keyBytes;
    ctx->wordSlotMask = static_cast<uint32_t>(capacity - 1);
    ctx->wordIndexBytes = ctx->deviceBytes - before;
    ctx->wordIndexKeys = keysInTable;
    ctx->wordSlots = slots;
    return MEMB_HIP_OK;
}

// One launch for `count` contexts of one device (a ReadersUnion: the words are read and hashed once, probed per model).
int launchResolve(
    memb_hip_ctx* const* ctxs, uint32_t* const* rows, size_t count, const uint8_t* bytes, uint64_t totalBytes,
    const uint32_t* offsets, uint32_t jobShift, size_t first, size_t n, hipStream_t stream)
{
    if (count == 0 || count > RESOLVE_MAX_MODELS) {
        return fail(MEMB_HIP_ERR_INVALID, "one to four contexts per word lookup");
    }
    for (size_t m = 0; m < count; ++m) {
        if (!ctxs[m] || (n && !rows[m])) {
            return fail(MEMB_HIP_ERR_INVALID, "null argument");
        }
        if (ctxs[m]->device != ctxs[0]->device) {
            return fail(MEMB_HIP_ERR_INVALID, "the contexts of one word lookup must live on one device");
        }
        if (!ctxs[m]->wordSlots) {
            return fail(MEMB_HIP_ERR_INVALID, "the context's keys are not on the device: call memb_hip_ctx_stage_words first");
        }
    }
    if (n == 0) {
        return MEMB_HIP_OK;
    }
    if (n >= 0xFFFFFFFFull || first % WAVE != 0) {
        return fail(MEMB_HIP_ERR_INVALID, "batch too large, or a range that does not start on a multiple of 64 words");
    }
    ResolveParams params{};
    params.queryBytes = bytes;
    params.queryOffsets = offsets;
    params.first = first;
    params.jobShift = jobShift;
    params.n = n;
    params.queryBytesTotal = totalBytes;
    params.models = static_cast<uint32_t>(count);
    for (size_t m = 0; m < count; ++m) {
        params.slots[m] = static_cast<const WordSlot*>(ctxs[m]->wordSlots);
        params.slotMask[m] = ctxs[m]->wordSlotMask;
        params.keyBytes[m] = ctxs[m]->wordKeyBytes;
        params.rows[m] = rows[m];
    }
    params.stageQueries = reinterpret_cast<uintptr_t>(bytes) % 16 == 0 ? 1u : 0u;
    const size_t perBlock = size_t(RESOLVE_WAVES) * WAVE;
    hipLaunchKernelGGL(
        resolve_words, dim3(static_cast<uint32_t>((n + perBlock - 1) / perBlock)), dim3(RESOLVE_WAVES * WAVE), 0, stream, params);
    hipError_t status = hipGetLastError();
    if (status != hipSuccess) {
        return fail(MEMB_HIP_ERR_DEVICE, std::string("resolve_words launch: ") + hipGetErrorString(status));
    }
    return MEMB_HIP_OK;
}

int launchResolve(
    memb_hip_ctx* ctx, const uint8_t* bytes, uint64_t totalBytes, const uint32_t* offsets, uint32_t jobShift, size_t first,
    size_t n, uint32_t* rows, hipStream_t stream)
{
    return launchResolve(&ctx, &rows, 1, bytes, totalBytes, offsets, jobShift, first, n, stream);
}

// memb_hip_resolve_range_union_device: words [firstWord, firstWord + nWords) of one batch against `count` contexts.
int resolve_range_union_device_checked(
    memb_hip_ctx* const* ctxs, size_t count, const memb_hip_words* batch, size_t firstWord, size_t nWords, uint32_t* const* rows,
    hipStream_t stream)
{
    if (!ctxs || !batch || !rows || count == 0 || !ctxs[0]) {
        return fail(MEMB_HIP_ERR_INVALID, "null argument");
    }
    if (batch->device != ctxs[0]->device) {
        return fail(MEMB_HIP_ERR_INVALID, "the word batch lives on another device than the context");
    }
    if (!batch->plan.bytes || firstWord > batch->plan.n || nWords > batch->plan.n - firstWord ||
        (nWords && firstWord % batch->plan.job_words != 0)) {
        return fail(MEMB_HIP_ERR_INVALID, "the range is not a run of whole jobs of the batch");
    }
    DeviceScope deviceScope(ctxs[0]->device);
    HIP_TRY(deviceScope.status());
    const int code = launchResolve(
        ctxs, rows, count, batch->deviceBytes, uint64_t(batch->plan.jobs) * batch->plan.job_bytes, batch->deviceOffsets,
        batch->jobShift, firstWord, nWords, stream);
    if (code == MEMB_HIP_OK && nWords) {
        return noteUse(const_cast<memb_hip_words*>(batch), stream);   // (bookkeeping of who still reads the buffers)
    }
    return code;
}

int resolve_range_device_checked(
    memb_hip_ctx* ctx, const memb_hip_words* batch, size_t firstWord, size_t nWords, uint32_t* rows, hipStream_t stream)
{
    if (!ctx || !batch || (nWords && !rows)) {
        return fail(MEMB_HIP_ERR_INVALID, "null argument");
    }
    return resolve_range_union_device_checked(&ctx, 1, batch, firstWord, nWords, &rows, stream);
}

int resolve_rows_device_checked(memb_hip_ctx* ctx, const memb_hip_words* batch, uint32_t* rows, hipStream_t stream)
{
    if (!ctx || !batch) {
        return fail(MEMB_HIP_ERR_INVALID, "null argument");
    }
    if (!batch->committed) {
        return fail(MEMB_HIP_ERR_INVALID, "the word batch is not committed");
    }
    return resolve_range_device_checked(ctx, batch, 0, batch->count, rows, stream);
}

// for decode_rows_checked (memb_hip.hip), which sits above this header: the context's mutex is held, its device current
int resolveBatchOnContextStream(memb_hip_ctx* ctx, const memb_hip_words* batch, uint32_t* rowsDevice)
{
    if (!batch->committed) {
        return fail(MEMB_HIP_ERR_INVALID, "the word batch is not committed");
    }
    if (batch->device != ctx->device) {
        return fail(MEMB_HIP_ERR_INVALID, "the word batch lives on another device than the context");
    }
    const int code = launchResolve(
        ctx, batch->deviceBytes, uint64_t(batch->plan.jobs) * batch->plan.job_bytes, batch->deviceOffsets, batch->jobShift, 0,
        batch->count, rowsDevice, ctx->stream);
    if (code == MEMB_HIP_OK && batch->count) {
        return noteUse(const_cast<memb_hip_words*>(batch), ctx->stream);
    }
    return code;
}

size_t wordBatchCount(const memb_hip_words* batch)
{
    return batch->committed ? batch->count : 0;
}

int resolve_packed_device_checked(
    memb_hip_ctx* ctx, const uint8_t* bytes, const uint32_t* offsets, size_t n, uint32_t* rows, hipStream_t stream)
{
    if (!ctx || (n && (!offsets || !rows))) {
        return fail(MEMB_HIP_ERR_INVALID, "null argument");
    }
    DeviceScope deviceScope(ctx->device);
    HIP_TRY(deviceScope.status());
    // (the extent of the caller's byte buffer is not known here: the offsets are taken at their word)
    return launchResolve(ctx, bytes, 0xFFFFFFFFull, offsets, 0, 0, n, rows, stream);
}

}  // namespace

// Host-buffer half of memb_hip_decode_rows: how decoded rows get from HBM into the
// caller's (pageable, possibly strided) host matrix.
//
// Host code of libmemb_hip.so. Included by memb_hip.hip only, inside its anonymous
// namespace, after the launch functions; see that file for the overview.
#pragma once

constexpr size_t RING_CHUNK_BYTES = size_t(32) << 20;

bool ensureRing(memb_hip_ctx* ctx)
{
    if (ctx->ringUnavailable) {
        return false;
    }
    if (ctx->ring[0]) {
        return true;
    }
    for (int i = 0; i < memb_hip_ctx::RING; ++i) {
        if (hipHostMalloc(&ctx->ring[i], RING_CHUNK_BYTES, hipHostMallocDefault) != hipSuccess ||
            hipEventCreateWithFlags(&ctx->ringEvents[i], hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            for (int k = 0; k < memb_hip_ctx::RING; ++k) {
                if (ctx->ring[k]) {
                    (void)hipHostFree(ctx->ring[k]);
                    ctx->ring[k] = nullptr;
                }
                if (ctx->ringEvents[k]) {
                    (void)hipEventDestroy(ctx->ringEvents[k]);
                    ctx->ringEvents[k] = nullptr;
                }
            }
            ctx->ringUnavailable = true;
            return false;
        }
    }
    return true;
}

void copyRows(float* destination, size_t ld, const float* source, size_t dim, size_t first, size_t last)
{
    if (ld == dim) {
        std::memcpy(destination + first * ld, source + first * dim, (last - first) * dim * sizeof(float));
        return;
    }
    for (size_t i = first; i < last; ++i) {
        std::memcpy(destination + i * ld, source + i * dim, dim * sizeof(float));
    }
}

// Rows of centroid indices (what OUT_KEYS wrote: keyRowBytes per row, one index per byte or,
// `fast`, per nibble, low nibble first) -> fp32 rows. The values are copies of the file's
// centroids, exactly what the device-side gather stores; rows whose id is not in the file
// become zeros (reference src/reader.cpp:41-47).
// streaming: results far larger than the caches are written with non-temporal stores -- a plain store first reads the
// line it is about to overwrite, and this loop is bound by the host's memory traffic (2.6 GB of rows + as much again read
// for ownership on the 2.2 M-word dump).
void expandKeyRows(
    const memb_hip_ctx* ctx, const uint8_t* keys, size_t keyRowBytes, const uint32_t* rows, float* destination, size_t ld,
    size_t first, size_t last, bool streaming = false)
{
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const size_t dim = ctx->dim;
    const float* codebook = ctx->hostCodebook.data();
    streaming = streaming && dim % 4 == 0 && ld % 4 == 0 && reinterpret_cast<uintptr_t>(destination) % 16 == 0;
    for (size_t i = first; i < last; ++i) {
        float* out = destination + i * ld;
        if (rows[i] >= ctx->nRows) {
            std::memset(out, 0, dim * sizeof(float));
            continue;
        }
        const uint8_t* source = keys + i * keyRowBytes;
        if (streaming) {
            for (size_t k = 0; k < dim; k += 4) {
                f32x4 value;
                if (ctx->fast) {   // one byte = two weights = one pair of the 256-pair table
                    const float* lo = codebook + 2 * size_t(source[k / 2]);
                    const float* hi = codebook + 2 * size_t(source[k / 2 + 1]);
                    value = f32x4{lo[0], lo[1], hi[0], hi[1]};
                } else {
                    value = f32x4{codebook[source[k]], codebook[source[k + 1]], codebook[source[k + 2]], codebook[source[k + 3]]};
                }
                __builtin_nontemporal_store(value, reinterpret_cast<f32x4*>(out + k));
            }
            continue;
        }
        if (ctx->fast) {
            const size_t pairs = dim / 2;
            for (size_t k = 0; k < pairs; ++k) {
                std::memcpy(out + 2 * k, codebook + 2 * size_t(source[k]), 2 * sizeof(float));
            }
            if (dim & 1) {
                out[dim - 1] = codebook[2 * size_t(source[pairs] & 15)];
            }
        } else {
            for (size_t k = 0; k < dim; ++k) {
                out[k] = codebook[source[k]];
            }
        }
    }
#if defined(__x86_64__)
    if (streaming) {
        asm volatile("sfence" ::: "memory");   // the non-temporal stores are visible before the thread reports its share done
    }
#endif
}

// Device rows of `rowBytes` each -> the caller's host rows, through the pinned ring.
// The copy engine writes chunks of the device buffer into the ring (enqueued on the
// context's stream, behind the decode); as a chunk lands, `threads` pooled host
// threads turn their shares of its rows into the caller's rows with
// emit(chunk data, first row of the chunk, first, last) -- a copy or an expansion,
// and in either case where fresh pages of the result are first touched, in
// parallel -- while the engine fills the next chunks. (A pageable hipMemcpy of a
// 2.6 GB result ran at 14-37 GB/s depending on the host, pages being faulted in
// one by one behind the engine, and 2-D copies for ld > dim slower still.)
template <typename Emit>
int streamRowsToHost(memb_hip_ctx* ctx, const void* deviceRows, size_t rowBytes, size_t words, size_t resultRowBytes, Emit emit)
{
    constexpr size_t RING = memb_hip_ctx::RING;
    // chunks of at most 32 MiB of device rows and at most 32 MiB worth of result rows
    // (tests shrink the chunk and force the threads to run every branch on small batches)
    size_t chunkRows = std::min(RING_CHUNK_BYTES / rowBytes, std::max<size_t>(1, RING_CHUNK_BYTES / resultRowBytes));
    if (ctx->switches.copyChunkRows) {
        chunkRows = std::min<size_t>(chunkRows, ctx->switches.copyChunkRows);
    }
    chunkRows = std::max<size_t>(1, chunkRows);
    const size_t chunks = (words + chunkRows - 1) / chunkRows;
    const size_t wanted = ctx->switches.copyThreads;
    const bool parallel = words * resultRowBytes >= (size_t(8) << 20) || ctx->switches.copyChunkRows != 0;
    const size_t threads = parallel ? wanted : 0;   // 0: this thread does the rows

    std::mutex mutex;
    std::condition_variable changed;
    size_t ready = 0;      // chunks that have landed in the ring
    size_t finished = 0;   // chunks emitted by every thread
    size_t pending[RING] = {};
    bool failed = false;

    auto chunkWords = [&](size_t chunk) { return std::min(chunkRows, words - chunk * chunkRows); };
    auto worker = [&](size_t index) {
        for (size_t chunk = 0; chunk < chunks; ++chunk) {
            {
                std::unique_lock<std::mutex> lock(mutex);
                changed.wait(lock, [&] { return ready > chunk || failed; });
                if (failed) {
                    return;
                }
            }
            const size_t count = chunkWords(chunk);
            const size_t share = (count + threads - 1) / threads;
            const size_t first = std::min(count, index * share), last = std::min(count, first + share);
            emit(ctx->ring[chunk % RING], chunk * chunkRows, first, last);
            {
                std::lock_guard<std::mutex> lock(mutex);
                if (++pending[chunk % RING] == threads) {
                    pending[chunk % RING] = 0;
                    ++finished;
                    changed.notify_all();
                }
            }
        }
    };
    // every job runs until the last chunk, so each needs a thread of its own
    if (threads && (!ctx->copyPool || ctx->copyPool->size() != threads)) {
        ctx->copyPool.reset(new memb::WorkerPool(threads));
    }
    if (threads) {
        ctx->copyPool->start(threads, worker);
    }

    hipError_t status = hipSuccess;
    size_t issued = 0;
    for (;;) {
        // keep the engine busy: a chunk may be issued once its ring slot has been emptied
        size_t issuable;
        {
            std::lock_guard<std::mutex> lock(mutex);
            if (finished >= chunks) {
                break;
            }
            issuable = std::min(chunks, finished + RING);
        }
        for (; issued < issuable && status == hipSuccess; ++issued) {
            status = hipMemcpyAsync(
                ctx->ring[issued % RING], static_cast<const char*>(deviceRows) + issued * chunkRows * rowBytes,
                chunkWords(issued) * rowBytes, hipMemcpyDeviceToHost, ctx->stream);
            if (status == hipSuccess) {
                status = hipEventRecord(ctx->ringEvents[issued % RING], ctx->stream);
            }
        }
        if (status != hipSuccess) {
            break;
        }
        if (ready < issued) {   // `ready` is written by this thread only
            status = hipEventSynchronize(ctx->ringEvents[ready % RING]);
            if (status != hipSuccess) {
                break;
            }
            if (threads == 0) {
                emit(ctx->ring[ready % RING], ready * chunkRows, 0, chunkWords(ready));
            }
            std::lock_guard<std::mutex> lock(mutex);
            ++ready;
            if (threads == 0) {
                ++finished;
            }
            changed.notify_all();
        } else {
            // every issued chunk has landed: wait until the threads free a slot (or are done)
            std::unique_lock<std::mutex> lock(mutex);
            changed.wait(lock, [&] { return finished >= chunks || (issued < chunks && finished + RING > issued); });
        }
    }
    if (status != hipSuccess) {
        std::lock_guard<std::mutex> lock(mutex);
        failed = true;
        changed.notify_all();
    }
    if (threads) {
        ctx->copyPool->wait();
    }
    if (status != hipSuccess) {
        (void)hipStreamSynchronize(ctx->stream);
        return fail(MEMB_HIP_ERR_DEVICE, std::string("batch copy: ") + hipGetErrorString(status));
    }
    return MEMB_HIP_OK;
}

// Dense device rows [words][dim] -> destination[i * ld .. + dim), i < words, host memory.
int copyRowsToHost(memb_hip_ctx* ctx, const float* deviceRows, size_t words, float* destination, size_t ld)
{
    const size_t dim = ctx->dim;
    const size_t rowBytes = dim * sizeof(float);
    if (rowBytes > RING_CHUNK_BYTES || !ensureRing(ctx)) {
        // no pinned memory to be had: plain (2-D) copy
        hipError_t status = hipMemcpy2DAsync(
            destination, ld * sizeof(float), deviceRows, rowBytes, rowBytes, words, hipMemcpyDeviceToHost, ctx->stream);
        if (status == hipSuccess) {
            status = hipStreamSynchronize(ctx->stream);
        }
        return status == hipSuccess ? MEMB_HIP_OK
                                    : fail(MEMB_HIP_ERR_DEVICE, std::string("batch copy: ") + hipGetErrorString(status));
    }
    return streamRowsToHost(
        ctx, deviceRows, rowBytes, words, rowBytes,
        [=](const void* chunk, size_t chunkFirst, size_t first, size_t last) {
            copyRows(destination + chunkFirst * ld, ld, static_cast<const float*>(chunk), dim, first, last);
        });
}

// Trained storage, host buffers: the kernel decodes the bitstreams into rows of
// centroid indices (1 or 1/2 byte per weight), those cross PCIe, and the ring's
// host threads expand them -- they write every output byte anyway, and read 4-8x
// less than a copy of fp32 rows would. The Huffman decode, the part that costs a
// CPU 3-4 ns per weight, stays on the GPU; the result is bit-identical (the
// values are the file's centroids either way).
int decodeRowsAsKeys(
    memb_hip_ctx* ctx, const uint32_t* deviceRowIds, const uint32_t* hostRowIds, size_t words, uint8_t* deviceKeys,
    float* destination, size_t ld)
{
    const size_t rowBytes = keyRowBytes(ctx);
    const bool streaming = ctx->switches.hostStreaming == 2 ||
        (ctx->switches.hostStreaming == 1 && words * ctx->dim * sizeof(float) >= (size_t(64) << 20));
    int code = launchTrained(
        ctx, deviceRowIds, words, reinterpret_cast<float*>(deviceKeys), ctx->dim, 0, ctx->stream, Epilogue(), true);
    if (code != MEMB_HIP_OK) {
        return code;
    }
    return streamRowsToHost(
        ctx, deviceKeys, rowBytes, words, ctx->dim * sizeof(float),
        [=](const void* chunk, size_t chunkFirst, size_t first, size_t last) {
            expandKeyRows(
                ctx, static_cast<const uint8_t*>(chunk), rowBytes, hostRowIds + chunkFirst, destination + chunkFirst * ld,
                ld, first, last, streaming);
        });
}

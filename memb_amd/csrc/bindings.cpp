// pybind11 module `_memb`: the reference's binding surface
// (python/memb_bindings.cpp:11-72) -- Reader(filename, num_threads) with
// dim / word_embedding / batch_embedding / keys, Builder(dim, storage, bits)
// with add_word / save, available_compression_strategies() -- plus additions
// for device-resident results and strided (concatenated) outputs.
#include "builder.h"
#include "reader.h"
#include "compression_strategy.h"
#include "codec.h"

#include <pybind11/pybind11.h>
#include <pybind11/stl.h>
#include <pybind11/numpy.h>

#include <sys/mman.h>

#include <chrono>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>

namespace py = pybind11;

namespace {

std::shared_ptr<memb::Reader> makeReader(
    const std::string& filename, size_t numThreads, int device, size_t maxDirectDecodeBits)
{
    if (maxDirectDecodeBits == 0) {
        return std::make_shared<memb::Reader>(filename, numThreads, device);
    }
    // the reference's test hook: a trained strategy with a short first-level
    // table, injected through the second Reader constructor (src/tests.cpp:76-88)
    return std::make_shared<memb::Reader>(
        filename, std::make_shared<memb::TrainedCompressionStrategy>(maxDirectDecodeBits), numThreads, device);
}

// UTF-8 pointers of a sequence of str, without copying the words (pybind11's
// vector<string> caster copies every one). The pointers stay valid while the
// sequence is alive; a word with an embedded NUL ends there, as it does for the
// reference's strcmp.
// A large result array is filled once, right after numpy allocated it: ask for
// transparent huge pages on it (hosts run THP in "madvise" mode), so that filling
// 2.6 GB takes ~1300 page faults instead of ~640,000. Advice only; ignored where
// THP is off.
void adviseHugePages(void* data, size_t bytes)
{
    const size_t huge = size_t(2) << 20;
    if (bytes < 16 * huge) {
        return;
    }
    const uintptr_t first = (reinterpret_cast<uintptr_t>(data) + huge - 1) / huge * huge;
    const uintptr_t last = (reinterpret_cast<uintptr_t>(data) + bytes) / huge * huge;
    if (last > first) {
        (void)::madvise(reinterpret_cast<void*>(first), last - first, MADV_HUGEPAGE);
    }
}

struct WordPointers {
    py::object fast;   // keeps the items alive
    std::vector<const char*> pointers;

    explicit WordPointers(const py::handle& words)
    {
        fast = py::reinterpret_steal<py::object>(PySequence_Fast(words.ptr(), "expected a list of str"));
        if (!fast) {
            throw py::error_already_set();
        }
        const Py_ssize_t count = PySequence_Fast_GET_SIZE(fast.ptr());
        PyObject** items = PySequence_Fast_ITEMS(fast.ptr());
        pointers.resize(static_cast<size_t>(count));
        for (Py_ssize_t i = 0; i < count; ++i) {
            if (!PyUnicode_Check(items[i])) {
                throw py::type_error("words must be str");
            }
            const char* text = PyUnicode_AsUTF8(items[i]);
            if (!text) {
                throw py::error_already_set();
            }
            pointers[static_cast<size_t>(i)] = text;
        }
    }

    size_t size() const { return pointers.size(); }
    const char* const* data() const { return pointers.data(); }
};

// A list of str into a word batch on the device (memb::WordBatch, include/memb_hip.h: memb_hip_words_plan) -- the part
// of a device word search that is host work, and at 2.2 M words the longest part of a lookup, so every word is touched
// ONCE: pooled threads read the str objects and write their UTF-8 bytes (up to the first NUL, where the reference's
// strcmp stops) straight into the batch's pinned job regions, which the lookup kernel reads over PCIe. The calling
// thread holds the GIL throughout, so no Python code runs meanwhile; the pool reads the objects' memory only: a compact
// ASCII str (nearly every token) has its bytes behind its header, a str with a cached UTF-8 form has pointer and length
// in its header. A job that meets anything else (non-ASCII without the cache, not a str at all) is redone after the
// calling thread has gone over its words through the API, which may allocate and raise. The walk is a cache miss per
// word and little else: every thread prefetches the objects a few words ahead.
// onChunk(firstWord, words) is called on the calling thread for consecutive runs of finished jobs, in order, while
// later jobs are still being filled: the lookups of a run overlap the filling of the next.
// The pool threads read str objects WITHOUT the GIL, relying on the calling thread's hold of it to keep the list and its
// strings from changing, and on the object layout of the CPython versions this was written against. Where either does
// not hold -- a free-threaded build, another interpreter, a CPython newer than the layouts known here -- every word goes
// through PyUnicode_AsUTF8AndSize on the calling thread instead (fillJobWithApi).
#if defined(Py_GIL_DISABLED) || defined(PYPY_VERSION) || defined(GRAALVM_PYTHON) || PY_VERSION_HEX < 0x03080000 || PY_VERSION_HEX >= 0x030E0000
constexpr bool DIRECT_STR_ACCESS = false;
#else
constexpr bool DIRECT_STR_ACCESS = true;
static_assert(sizeof(PyASCIIObject) <= sizeof(PyCompactUnicodeObject), "str layout: ASCII header inside the compact one");
#endif

struct WordFiller {
    static size_t poolThreads()
    {
        const char* text = std::getenv("MEMB_PACK_THREADS");
        const unsigned wanted = text && *text ? static_cast<unsigned>(std::strtoul(text, nullptr, 10)) : 64u;
        return std::max(2u, std::min(std::min(wanted, 128u), std::thread::hardware_concurrency()));
    }

    // Threads for a batch of `count` words: the walk is a cache miss per word, so large batches want many (2.2 M shuffled
    // words: 2.1 / 1.5 / 0.87 ms with 16 / 32 / 64 threads) and small ones few (100 000 words: 0.068 / 0.079 / 0.109 ms):
    // one per 32 768 words, at least sixteen (eight: 100 000 words 0.16-0.22 ms against 0.14-0.15), at most the pool
    // (MEMB_PACK_THREADS, default 64).
    static size_t threadsFor(size_t count)
    {
        return std::min<size_t>(pool().size(), std::max<size_t>(16, count / 32768));
    }

    static memb::WorkerPool& pool()
    {
        static memb::WorkerPool workers(poolThreads() - 1);
        return workers;
    }

    // The pool serves one batch at a time (start .. wait): whoever fills through it -- this filler or PackedFiller, which
    // runs without the GIL -- holds this mutex meanwhile, and fills on its own thread when it cannot get it.
    static std::mutex& poolMutex()
    {
        static std::mutex mutex;
        return mutex;
    }

    enum JobState : int { FILLED = 0, NEEDS_API = 1, TOO_LONG = 2 };

    // One job of the plan; returns its state (and, for TOO_LONG, the bytes it needs in *needed).
    static JobState fillJob(const memb_hip_words_plan& plan, PyObject** items, size_t job, uint64_t* needed)
    {
        constexpr size_t AHEAD = 12;
        const size_t first = job * plan.job_words, last = std::min(plan.n, first + plan.job_words);
        uint32_t* offsets = plan.offsets + job * (plan.job_words + 1);
        const uint64_t base = uint64_t(job) * plan.job_bytes;
        uint64_t at = 0;
        bool fits = true;
        for (size_t i = first; i < last; ++i) {
            if (i + AHEAD < last) {
                __builtin_prefetch(items[i + AHEAD]);
                __builtin_prefetch(reinterpret_cast<const char*>(items[i + AHEAD]) + 64);
            }
            PyObject* item = items[i];
            const char* text = nullptr;
            Py_ssize_t size = 0;
            bool ready = PyUnicode_Check(item);
#if PY_VERSION_HEX < 0x030C0000
            ready = ready && PyUnicode_IS_READY(item);
#endif
            if (ready && PyUnicode_IS_COMPACT_ASCII(item)) {
                text = reinterpret_cast<const char*>(reinterpret_cast<PyASCIIObject*>(item) + 1);
                size = PyUnicode_GET_LENGTH(item);
            } else if (ready && reinterpret_cast<PyCompactUnicodeObject*>(item)->utf8) {
                // (every str that is not compact ASCII -- compact or not -- begins with a PyCompactUnicodeObject, whose
                // utf8 / utf8_length hold the cached UTF-8 form once PyUnicode_AsUTF8AndSize has been asked for it)
                text = reinterpret_cast<PyCompactUnicodeObject*>(item)->utf8;
                size = reinterpret_cast<PyCompactUnicodeObject*>(item)->utf8_length;
            }
            if (!text || size < 0) {
                return NEEDS_API;
            }
            const size_t length = ::strnlen(text, static_cast<size_t>(size));
            if (fits && at + length <= plan.job_bytes) {
                offsets[i - first] = static_cast<uint32_t>(base + at);
                std::memcpy(plan.bytes + base + at, text, length);
            } else {
                fits = false;
            }
            at += length;
        }
        if (!fits) {
            *needed = at;
            return TOO_LONG;
        }
        offsets[last - first] = static_cast<uint32_t>(base + at);
        return FILLED;
    }

    // The same job through the C API alone, on the calling thread (GIL held): interpreters whose str layout is not known here.
    static JobState fillJobWithApi(const memb_hip_words_plan& plan, PyObject** items, size_t job, uint64_t* needed)
    {
        const size_t first = job * plan.job_words, last = std::min(plan.n, first + plan.job_words);
        uint32_t* offsets = plan.offsets + job * (plan.job_words + 1);
        const uint64_t base = uint64_t(job) * plan.job_bytes;
        uint64_t at = 0;
        bool fits = true;
        for (size_t i = first; i < last; ++i) {
            if (!PyUnicode_Check(items[i])) {
                throw py::type_error("words must be str");
            }
            Py_ssize_t size = 0;
            const char* text = PyUnicode_AsUTF8AndSize(items[i], &size);
            if (!text) {
                throw py::error_already_set();
            }
            const size_t length = ::strnlen(text, static_cast<size_t>(size));
            if (fits && at + length <= plan.job_bytes) {
                offsets[i - first] = static_cast<uint32_t>(base + at);
                std::memcpy(plan.bytes + base + at, text, length);
            } else {
                fits = false;
            }
            at += length;
        }
        if (!fits) {
            *needed = at;
            return TOO_LONG;
        }
        offsets[last - first] = static_cast<uint32_t>(base + at);
        return FILLED;
    }

    // The words a job could not read without the API: the calling thread (GIL held) makes their UTF-8 form
    // available, or raises for what is not a str.
    static void prepareWithApi(PyObject** items, size_t first, size_t last)
    {
        for (size_t i = first; i < last; ++i) {
            if (!PyUnicode_Check(items[i])) {
                throw py::type_error("words must be str");
            }
            Py_ssize_t size = 0;
            if (!PyUnicode_AsUTF8AndSize(items[i], &size)) {   // (caches the UTF-8 form in the object)
                throw py::error_already_set();
            }
        }
    }

    template <typename OnChunk>
    static size_t fill(memb::WordBatch& batch, const py::handle& words, OnChunk onChunk)
    {
        py::object fast = py::reinterpret_steal<py::object>(PySequence_Fast(words.ptr(), "expected a list of str"));
        if (!fast) {
            throw py::error_already_set();
        }
        const size_t count = static_cast<size_t>(PySequence_Fast_GET_SIZE(fast.ptr()));
        PyObject** items = PySequence_Fast_ITEMS(fast.ptr());
        size_t bytesPerWord = 0;
        for (int attempt = 0; attempt < 48; ++attempt) {
            memb_hip_words_plan plan;
            {
                py::gil_scoped_release release;   // (begin waits for the lookups that still read the previous batch)
                plan = batch.begin(count, bytesPerWord);
            }
            std::vector<int> state(plan.jobs, FILLED);
            std::vector<uint64_t> needed(plan.jobs, 0);
            if (count == 0) {
                plan.offsets[0] = 0;
                batch.commit();
                return 0;
            }
            std::unique_lock<std::mutex> lock(WordFiller::poolMutex(), std::try_to_lock);
            const bool pooled = DIRECT_STR_ACCESS && count >= 8192 && lock.owns_lock();
            bool clean = true;
            if (!pooled) {
                for (size_t job = 0; job < plan.jobs; ++job) {
                    state[job] = DIRECT_STR_ACCESS ? fillJob(plan, items, job, &needed[job]) : fillJobWithApi(plan, items, job, &needed[job]);
                    clean = clean && state[job] == FILLED;
                }
                if (clean) {
                    onChunk(size_t(0), count);
                }
            } else {
                // chunks of consecutive jobs; the pool hands jobs out in order, so chunks complete roughly in order
                // (at most eight, of at least 65 536 words: a lookup launch is a few microseconds of this thread's time)
                const size_t chunkJobs = std::max((plan.jobs + 7) / 8, (size_t(65536) + plan.job_words - 1) / plan.job_words);
                const size_t chunks = (plan.jobs + chunkJobs - 1) / chunkJobs;
                std::vector<std::atomic<size_t>> done(chunks);
                for (auto& counter : done) {
                    counter.store(0, std::memory_order_relaxed);
                }
                pool().start(plan.jobs, [&](size_t job) {
                    state[job] = fillJob(plan, items, job, &needed[job]);
                    done[job / chunkJobs].fetch_add(1, std::memory_order_release);
                }, threadsFor(count));
                std::exception_ptr failure;
                for (size_t chunk = 0; chunk < chunks; ++chunk) {
                    const size_t firstJob = chunk * chunkJobs, lastJob = std::min(plan.jobs, firstJob + chunkJobs);
                    // (the GIL stays with this thread while the pool reads the list and its strings: it is what keeps them
                    // from changing under the pool -- as the reference holds it for a whole batch_embedding call,
                    // python/memb_bindings.cpp:54-63. A 2.2 M-word fill is about a millisecond)
                    while (done[chunk].load(std::memory_order_acquire) < lastJob - firstJob) {
                        std::this_thread::yield();
                    }
                    for (size_t job = firstJob; job < lastJob; ++job) {
                        clean = clean && state[job] == FILLED;
                    }
                    if (clean && !failure) {
                        try {
                            const size_t firstWord = firstJob * plan.job_words;
                            onChunk(firstWord, std::min(count, lastJob * plan.job_words) - firstWord);
                        } catch (...) {
                            failure = std::current_exception();   // (the pool still works on this thread's stack)
                        }
                    }
                }
                pool().wait();
                if (failure) {
                    std::rethrow_exception(failure);
                }
            }
            if (clean) {
                batch.commit();
                return count;
            }
            // the rare paths: make the words of the jobs that need it readable, size the regions for the longest job
            uint64_t longest = 0;
            for (size_t job = 0; job < plan.jobs; ++job) {
                if (state[job] == NEEDS_API) {
                    prepareWithApi(items, job * plan.job_words, std::min(count, (job + 1) * plan.job_words));
                }
                longest = std::max(longest, needed[job]);
            }
            if (longest) {
                bytesPerWord = std::max<size_t>(
                    2 * (plan.job_bytes / plan.job_words), (longest + plan.job_words - 1) / plan.job_words * 5 / 4 + 1);
            } else {
                bytesPerWord = plan.job_bytes / plan.job_words;
            }
        }
        throw std::runtime_error("internal error: the word batch does not converge");
    }
};

// Words that are packed already -- a tokenizer's output: one blob of UTF-8 bytes and n + 1 ascending offsets -- into a word
// batch's pinned job regions: per job one memcpy and a run of rebased offsets, on the pool, no str object in sight (the
// walk over 2.2 M str objects is a cache miss per word: 0.8-1.2 ms; this is 0.1-0.2 ms of memcpy). The caller holds no GIL.
// onChunk as in WordFiller::fill. Offsets that are not ascending or leave the blob are refused before anything is written.
struct PackedFiller {
    template <typename OnChunk>
    static size_t fill(memb::WordBatch& batch, const uint8_t* blob, size_t blobBytes, const uint32_t* offsets, size_t count, OnChunk onChunk)
    {
        if (count == 0) {
            const memb_hip_words_plan plan = batch.begin(0, 0);
            plan.offsets[0] = 0;
            batch.commit();
            return 0;
        }
        if (offsets[count] > blobBytes) {
            throw std::invalid_argument("offsets[n] lies beyond the end of the bytes");
        }
        size_t bytesPerWord = std::max<size_t>(1, (size_t(offsets[count]) - offsets[0] + count - 1) / count + 1);
        for (int attempt = 0; attempt < 3; ++attempt) {
            const memb_hip_words_plan plan = batch.begin(count, bytesPerWord);
            // the longest job decides the region size (jobs are runs of job_words consecutive words)
            uint64_t longest = 0;
            for (size_t job = 0; job < plan.jobs; ++job) {
                const size_t first = job * plan.job_words, last = std::min(count, first + plan.job_words);
                if (offsets[last] < offsets[first]) {
                    throw std::invalid_argument("offsets must ascend");
                }
                longest = std::max<uint64_t>(longest, offsets[last] - offsets[first]);
            }
            if (longest > plan.job_bytes) {
                bytesPerWord = (longest + plan.job_words - 1) / plan.job_words + 1;
                continue;
            }
            std::atomic<bool> descending{false};
            auto fillJob = [&](size_t job) {
                const size_t first = job * plan.job_words, last = std::min(count, first + plan.job_words);
                uint32_t* target = plan.offsets + job * (plan.job_words + 1);
                const uint32_t base = static_cast<uint32_t>(job * plan.job_bytes);
                const uint32_t origin = offsets[first];
                bool ordered = true;
                for (size_t i = first; i <= last; ++i) {
                    ordered = ordered && (i == first || offsets[i] >= offsets[i - 1]);
                    target[i - first] = base + (offsets[i] - origin);
                }
                if (!ordered) {
                    descending.store(true, std::memory_order_relaxed);
                    return;
                }
                std::memcpy(plan.bytes + size_t(job) * plan.job_bytes, blob + origin, offsets[last] - origin);
            };
            std::unique_lock<std::mutex> lock(WordFiller::poolMutex(), std::try_to_lock);
            if (count < 8192 || !lock.owns_lock()) {
                for (size_t job = 0; job < plan.jobs; ++job) {
                    fillJob(job);
                }
                if (!descending.load()) {
                    onChunk(size_t(0), count);
                }
            } else {
                // Runs of jobs whose lookups are launched as they finish: a sixteenth of the batch, then up to a quarter, a half,
                // the rest. The lookup reads the words over PCIe (2.2 M words: 0.68 ms at 51 GB/s) and every launch costs it a
                // ramp of ~15 us, so few launches, the first one early (round 6, tools/perf/r6/packed.sh: eight equal runs
                // 0.87-0.98 ms, sixteen 0.92-0.98, thirty-two 1.07-1.12, sixty-four 1.29-1.36).
                std::vector<size_t> ends;
                for (size_t fraction : {16u, 4u, 2u, 1u}) {
                    const size_t end = fraction == 1 ? plan.jobs : std::max<size_t>(1, plan.jobs / fraction);
                    if ((ends.empty() || end > ends.back()) && (fraction == 1 || end * plan.job_words >= 65536)) {
                        ends.push_back(end);
                    }
                }
                const size_t chunks = ends.size();
                auto chunkOf = [&](size_t job) {
                    size_t chunk = 0;
                    while (job >= ends[chunk]) {
                        ++chunk;
                    }
                    return chunk;
                };
                std::vector<std::atomic<size_t>> done(chunks);
                for (auto& counter : done) {
                    counter.store(0, std::memory_order_relaxed);
                }
                WordFiller::pool().start(plan.jobs, [&](size_t job) {
                    fillJob(job);
                    done[chunkOf(job)].fetch_add(1, std::memory_order_release);
                }, WordFiller::threadsFor(count));
                std::exception_ptr failure;
                for (size_t chunk = 0; chunk < chunks; ++chunk) {
                    const size_t firstJob = chunk ? ends[chunk - 1] : 0, lastJob = ends[chunk];
                    while (done[chunk].load(std::memory_order_acquire) < lastJob - firstJob) {
                        std::this_thread::yield();
                    }
                    if (!failure && !descending.load(std::memory_order_relaxed)) {
                        try {
                            const size_t firstWord = firstJob * plan.job_words;
                            onChunk(firstWord, std::min(count, lastJob * plan.job_words) - firstWord);
                        } catch (...) {
                            failure = std::current_exception();
                        }
                    }
                }
                WordFiller::pool().wait();
                if (failure) {
                    std::rethrow_exception(failure);
                }
            }
            if (descending.load()) {
                throw std::invalid_argument("offsets must ascend");
            }
            batch.commit();
            return count;
        }
        throw std::runtime_error("internal error: the word batch does not converge");
    }
};

// batch_embedding / batch_embedding_into: a list of str -> rows in host memory. From a few thousand words on (and when no
// other call of this Reader holds its word batch) the str objects are written straight into the reader's pinned word
// batch by pooled threads and BOTH the word search and the decode run on the device (memb_hip_decode_words); otherwise the
// host search on UTF-8 pointers, as before. The GIL is held while the str objects are read, released for the device work.
void wordsToHostRows(memb::Reader& reader, const py::sequence& wordList, size_t count, float* destination, size_t ld, size_t colOff)
{
    memb::Reader::WordBatchLease lease = reader.leaseWordBatch(count);
    if (lease.batch) {
        WordFiller::fill(*lease.batch, wordList, [](size_t, size_t) {});
        bool done;
        {
            py::gil_scoped_release release;
            done = reader.leasedWordsToBuffer(lease, destination, ld, colOff);
        }
        if (done) {
            return;
        }
    }
    lease = memb::Reader::WordBatchLease();
    WordPointers words(wordList);
    py::gil_scoped_release release;
    reader.batchEmbeddingToStridedBuffer(words.data(), words.size(), destination, ld, colOff);
}

py::dict contextInfo(memb::Reader& reader, uint64_t batchWords)
{
    if (reader.device() == memb::CompressedStorage::HOST_DEVICE) {
        py::dict result;
        result["device"] = "cpu";
        result["dim"] = reader.dim();
        result["n_rows"] = reader.size();
        result["kernel"] = "host decode (" + reader.storageName() + ")";
        return result;
    }
    memb_hip_ctx_info info{};
    info.struct_size = sizeof(info);
    info.batch_words = batchWords;
    if (memb_hip_ctx_get_info(reader.deviceContext(), &info) != MEMB_HIP_OK) {
        throw std::runtime_error(memb_hip_last_error());
    }
    py::dict result;
    result["device"] = info.device;
    result["storage"] = info.storage;
    result["dim"] = info.dim;
    result["n_rows"] = info.n_rows;
    result["device_bytes"] = info.device_bytes;
    result["root_bits"] = info.root_bits;
    result["max_code_bits"] = info.max_code_bits;
    result["table_entries"] = info.table_entries;
    result["max_stream_bytes"] = info.max_stream_bytes;
    result["waves_per_block"] = info.waves_per_block;
    result["lanes_per_word"] = info.lanes_per_word;
    result["segment_symbols"] = info.segment_symbols;
    result["lds_bytes_per_block"] = info.lds_bytes_per_block;
    result["kernel"] = std::string(info.kernel);
    result["row_layout"] = info.row_layout;
    result["row_bytes"] = info.row_bytes;
    result["kernel_registers"] = info.kernel_registers;
    result["register_waves_per_cu"] = info.register_waves_per_cu;
    result["tiles_per_wavefront"] = info.tiles_per_wavefront;
    result["union_kernel"] = std::string(info.union_kernel);
    result["word_index_bytes"] = info.word_index_bytes;
    result["word_index_slots"] = info.word_index_slots;
    result["word_index_keys"] = info.word_index_keys;
    return result;
}

}  // namespace

// The caller's output matrix for the *_into calls, used in place: float32, writeable, two-dimensional,
// unit column stride; rows may be strided (a column or row range of a wider matrix). Anything
// else is refused -- letting pybind11 convert it would fill a temporary copy and leave the
// caller's matrix untouched without a word.
struct OutputMatrix {
    float* data;
    size_t rows;
    size_t columns;
    size_t ld;   // floats between the starts of consecutive rows
};

OutputMatrix outputMatrix(const py::array& out)
{
    if (!py::isinstance<py::array_t<float>>(out) || !py::dtype::of<float>().is(out.dtype())) {
        throw py::type_error("out must be a numpy float32 array (no conversion is made: the result is written in place)");
    }
    if (!out.writeable()) {
        throw py::type_error("out is read-only");
    }
    if (out.ndim() != 2) {
        throw py::type_error("out must be 2-dimensional");
    }
    const py::ssize_t item = static_cast<py::ssize_t>(sizeof(float));
    if ((out.shape(1) > 1 && out.strides(1) != item) || out.strides(0) < 0 || out.strides(0) % item != 0 ||
        (out.shape(0) > 1 && out.strides(0) < out.shape(1) * item)) {
        throw py::type_error("out must have unit column stride and non-overlapping rows (row slices and column ranges of a C-contiguous matrix are fine)");
    }
    OutputMatrix matrix;
    matrix.data = static_cast<float*>(const_cast<void*>(out.data()));
    matrix.rows = static_cast<size_t>(out.shape(0));
    matrix.columns = static_cast<size_t>(out.shape(1));
    matrix.ld = out.shape(0) > 1 ? static_cast<size_t>(out.strides(0) / item) : matrix.columns;
    return matrix;
}

PYBIND11_MODULE(_memb, m) {
    py::class_<memb::Builder>(m, "Builder")
        .def(py::init<size_t, const std::string&, size_t>())
        .def(py::init<size_t, const std::string&, size_t, int>(),
             py::arg("dim"), py::arg("storage_type"), py::arg("bits_per_weight"), py::arg("device"))
        .def(
            "add_word",
            [](memb::Builder& builder, const std::string& word, py::array_t<float, py::array::c_style> values)
            {
                auto valuesBuffer = values.request();
                if (valuesBuffer.ndim != 1) {
                    throw std::runtime_error("Word vector must be 1-dimensional");
                }
                builder.addWord(
                    word, reinterpret_cast<const float*>(valuesBuffer.ptr), static_cast<size_t>(valuesBuffer.shape[0]));
            })
        .def(
            "add_words",
            [](memb::Builder& builder,
               const std::vector<std::string>& words,
               py::array_t<float, py::array::c_style | py::array::forcecast> matrix)
            {
                auto buffer = matrix.request();
                if (buffer.ndim != 2 || static_cast<size_t>(buffer.shape[0]) != words.size()) {
                    throw std::runtime_error("Expected a matrix with one row per word");
                }
                const float* values = reinterpret_cast<const float*>(buffer.ptr);
                const size_t dim = static_cast<size_t>(buffer.shape[1]);
                py::gil_scoped_release release;   // copies rows: other Python threads may run meanwhile
                builder.addWords(words, values, dim);
            })
        .def(
            "save",
            [](memb::Builder& builder, const std::string& filename)
            {
                py::gil_scoped_release release;
                builder.save(filename);
            });

    // a batch of query words on a device (memb::WordBatch): see Reader.words_to_rows_device
    py::class_<memb::WordBatch, std::shared_ptr<memb::WordBatch>>(m, "WordBatch")
        .def(py::init<int>(), py::arg("device"))
        .def("size", [](memb::WordBatch& batch) { return batch.size(); })
        .def("device", [](memb::WordBatch& batch) { return batch.device(); })
        .def(
            "pack",
            [](memb::WordBatch& batch, const py::sequence& wordList) {
                return WordFiller::fill(batch, wordList, [](size_t, size_t) {});
            },
            py::arg("words"),
            "fills the batch from a list of str (no lookup); returns the number of words");

    py::class_<memb::Reader, std::shared_ptr<memb::Reader>>(m, "Reader")
        .def(py::init([](const std::string& filename, size_t numThreads) {
            return makeReader(filename, numThreads, -1, 0);
        }))
        .def(
            py::init([](const std::string& filename, size_t numThreads, int device, size_t maxDirectDecodeBits) {
                return makeReader(filename, numThreads, device, maxDirectDecodeBits);
            }),
            py::arg("filename"),
            py::arg("num_threads"),
            py::arg("device"),
            py::arg("max_direct_decode_bits") = 0)
        .def("dim", [](memb::Reader& reader) { return reader.dim(); })
        .def(
            "word_embedding",
            [](memb::Reader& reader, const std::string& word)
            {
                py::array_t<float> result(reader.dim());
                auto buffer = result.request();
                reader.wordEmbeddingToBuffer(word, reinterpret_cast<float*>(buffer.ptr));
                return result;
            })
        .def(
            "batch_embedding",
            [](memb::Reader& reader, const py::sequence& wordList)
            {
                const size_t count = static_cast<size_t>(py::len(wordList));
                py::array_t<float> result({count, reader.dim()});
                auto buffer = result.request();
                float* destination = reinterpret_cast<float*>(buffer.ptr);
                adviseHugePages(destination, count * reader.dim() * sizeof(float));
                wordsToHostRows(reader, wordList, count, destination, reader.dim(), 0);
                return result;
            })
        .def("keys", [](memb::Reader& reader) { return reader.keys(); })
        // ---- additions ----
        .def("size", [](memb::Reader& reader) { return reader.size(); })
        .def("device", [](memb::Reader& reader) { return reader.device(); })
        .def("set_host_below", [](memb::Reader& reader, size_t words) { reader.setHostBelow(words); })
        .def("host_below", [](memb::Reader& reader) { return reader.hostBelow(); })
        .def("host_rows_decoded", [](memb::Reader& reader) { return reader.hostRowsDecoded(); })
        .def("storage_name", [](memb::Reader& reader) { return reader.storageName(); })
        .def("info", &contextInfo, py::arg("batch_words") = 0)
        .def(
            "set_option",
            [](memb::Reader& reader, const std::string& name, uint64_t value) {
                if (memb_hip_ctx_set_option(reader.deviceContext(), name.c_str(), value) != MEMB_HIP_OK) {
                    throw std::runtime_error(memb_hip_last_error());
                }
            },
            "tuning knob of the device context (include/memb_hip.h: memb_hip_ctx_set_option)")
        .def("has_word_index", [](memb::Reader& reader) { return reader.hasWordIndex(); })
        .def(
            "context_handle",
            [](memb::Reader& reader) { return reinterpret_cast<uintptr_t>(reader.deviceContext()); })
        .def(
            "resolve_rows",
            [](memb::Reader& reader, const py::sequence& wordList)
            {
                WordPointers words(wordList);
                py::array_t<uint32_t> rows(words.size());
                auto buffer = rows.request();
                uint32_t* destination = reinterpret_cast<uint32_t*>(buffer.ptr);
                {
                    py::gil_scoped_release release;
                    reader.resolveRows(words.data(), words.size(), destination);
                }
                return rows;
            })
        .def(
            "batches_to_device",
            [](memb::Reader& reader, const std::vector<std::tuple<uintptr_t, size_t, uintptr_t, size_t, size_t>>& batches,
               uintptr_t stream) {
                std::vector<memb_hip_batch> list(batches.size());
                for (size_t k = 0; k < batches.size(); ++k) {
                    list[k].rows = reinterpret_cast<const uint32_t*>(std::get<0>(batches[k]));
                    list[k].n = std::get<1>(batches[k]);
                    list[k].out = reinterpret_cast<float*>(std::get<2>(batches[k]));
                    list[k].ld = std::get<3>(batches[k]);
                    list[k].col_off = std::get<4>(batches[k]);
                }
                reader.batchesToDeviceBuffers(list.data(), list.size(), reinterpret_cast<void*>(stream));
            },
            py::arg("batches"),
            py::arg("stream") = 0,
            "several (rows_ptr, n, out_ptr, ld, col_off) lookups in one launch (memb_hip_decode_batches_device)")
        .def("stage_words", [](memb::Reader& reader) {
            py::gil_scoped_release release;
            reader.stageWords();
        })
        .def(
            "resolve_batch_to_device",
            [](memb::Reader& reader, memb::WordBatch& batch, uintptr_t rows, uintptr_t stream) {
                py::gil_scoped_release release;   // (may stage the keys on first use)
                reader.resolveRowsToDevice(batch, reinterpret_cast<uint32_t*>(rows), reinterpret_cast<void*>(stream));
            },
            py::arg("batch"),
            py::arg("rows_ptr"),
            py::arg("stream") = 0,
            "row ids of an already packed batch into device memory (batch.size() uint32 entries)")
        .def(
            "words_to_rows_device",
            [](memb::Reader& reader, memb::WordBatch& batch, const py::sequence& wordList, uintptr_t rows, uintptr_t stream) {
                // fill + lookup under one hold of the GIL (the batch object is not shared between two calls in flight);
                // the lookups of finished runs of jobs are enqueued while the pool fills the next. Staging the keys (first
                // call only: tens of MB to HBM, the hash table built, a synchronize) runs with the GIL released.
                {
                    py::gil_scoped_release release;
                    reader.stageWords();
                }
                return WordFiller::fill(batch, wordList, [&](size_t firstWord, size_t words) {
                    reader.resolveRangeToDevice(batch, firstWord, words, reinterpret_cast<uint32_t*>(rows), reinterpret_cast<void*>(stream));
                });
            },
            py::arg("batch"),
            py::arg("words"),
            py::arg("rows_ptr"),
            py::arg("stream") = 0,
            "words -> row ids in device memory (len(words) uint32 entries at rows_ptr), enqueued on `stream`")
        .def(
            "packed_to_rows_device",
            [](memb::Reader& reader, memb::WordBatch& batch, const py::buffer& bytes, const py::array& offsets, uintptr_t rows, uintptr_t stream) {
                const py::buffer_info blob = bytes.request();
                if (blob.ndim != 1 || blob.itemsize != 1 || (blob.size > 1 && blob.strides[0] != 1)) {
                    throw py::type_error("bytes must be a contiguous one-dimensional buffer of bytes");
                }
                if (!py::isinstance<py::array_t<uint32_t>>(offsets) || offsets.ndim() != 1 || offsets.shape(0) < 1 ||
                    !(offsets.flags() & py::array::c_style)) {
                    throw py::type_error("offsets must be a contiguous numpy.uint32 array of n + 1 entries");
                }
                const size_t count = static_cast<size_t>(offsets.shape(0)) - 1;
                const uint32_t* starts = static_cast<const uint32_t*>(offsets.data());
                const uint8_t* data = static_cast<const uint8_t*>(blob.ptr);
                const size_t dataBytes = static_cast<size_t>(blob.size);
                py::gil_scoped_release release;   // (plain memory from here on: the views above keep it alive)
                reader.stageWords();
                return PackedFiller::fill(batch, data, dataBytes, starts, count, [&](size_t firstWord, size_t words) {
                    reader.resolveRangeToDevice(batch, firstWord, words, reinterpret_cast<uint32_t*>(rows), reinterpret_cast<void*>(stream));
                });
            },
            py::arg("batch"),
            py::arg("bytes"),
            py::arg("offsets"),
            py::arg("rows_ptr"),
            py::arg("stream") = 0,
            "words packed as one buffer of UTF-8 bytes + n + 1 ascending uint32 offsets -> row ids in device memory")
        .def(
            "packed_device_to_rows_device",
            [](memb::Reader& reader, uintptr_t bytes, uintptr_t offsets, size_t count, uintptr_t rows, uintptr_t stream) {
                py::gil_scoped_release release;
                reader.stageWords();
                if (memb_hip_resolve_packed_device(
                        reader.deviceContext(), reinterpret_cast<const uint8_t*>(bytes), reinterpret_cast<const uint32_t*>(offsets), count,
                        reinterpret_cast<uint32_t*>(rows), reinterpret_cast<void*>(stream)) != MEMB_HIP_OK) {
                    throw std::runtime_error(std::string("HIP word search failed: ") + memb_hip_last_error());
                }
            },
            py::arg("bytes_ptr"),
            py::arg("offsets_ptr"),
            py::arg("n"),
            py::arg("rows_ptr"),
            py::arg("stream") = 0,
            "the same for bytes and offsets that are in device memory already (memb_hip_resolve_packed_device)")
        .def(
            "batch_embedding_into",
            [](memb::Reader& reader,
               const py::sequence& wordList,
               const py::array& out,
               size_t colOff)
            {
                const size_t count = static_cast<size_t>(py::len(wordList));
                const OutputMatrix matrix = outputMatrix(out);
                if (matrix.rows != count || matrix.columns < colOff + reader.dim()) {
                    throw std::runtime_error("Output must be a (len(words), >= col_off + dim) float32 matrix");
                }
                wordsToHostRows(reader, wordList, count, matrix.data, matrix.ld, colOff);
            })
        .def(
            "rows_embedding",
            [](memb::Reader& reader, py::array_t<uint32_t, py::array::c_style | py::array::forcecast> rows)
            {
                auto rowsBuffer = rows.request();
                if (rowsBuffer.ndim != 1) {
                    throw std::runtime_error("Row ids must be 1-dimensional");
                }
                size_t n = static_cast<size_t>(rowsBuffer.shape[0]);
                py::array_t<float> result({n, reader.dim()});
                auto buffer = result.request();
                float* destination = reinterpret_cast<float*>(buffer.ptr);
                adviseHugePages(destination, n * reader.dim() * sizeof(float));
                const uint32_t* source = reinterpret_cast<const uint32_t*>(rowsBuffer.ptr);
                {
                    py::gil_scoped_release release;
                    reader.rowsToBuffer(source, n, destination, reader.dim(), 0);
                }
                return result;
            })
        .def(
            "rows_embedding_into",
            [](memb::Reader& reader,
               py::array_t<uint32_t, py::array::c_style | py::array::forcecast> rows,
               const py::array& out,
               size_t colOff)
            {
                auto rowsBuffer = rows.request();
                const OutputMatrix matrix = outputMatrix(out);
                if (rowsBuffer.ndim != 1 || matrix.rows != static_cast<size_t>(rowsBuffer.shape[0]) ||
                    matrix.columns < colOff + reader.dim()) {
                    throw std::runtime_error("Output must be a (len(rows), >= col_off + dim) float32 matrix");
                }
                const uint32_t* source = reinterpret_cast<const uint32_t*>(rowsBuffer.ptr);
                const size_t n = matrix.rows;
                py::gil_scoped_release release;
                reader.rowsToBuffer(source, n, matrix.data, matrix.ld, colOff);
            },
            py::arg("rows"),
            py::arg("out"),
            py::arg("col_off") = 0)
        .def(
            "rows_to_device",
            [](memb::Reader& reader,
               uintptr_t rows,
               size_t n,
               uintptr_t out,
               size_t ld,
               size_t colOff,
               uintptr_t stream,
               bool accumulate,
               float divisor,
               bool randomOrder)
            {
                reader.rowsToDeviceBuffer(
                    reinterpret_cast<const uint32_t*>(rows),
                    n,
                    reinterpret_cast<float*>(out),
                    ld,
                    colOff,
                    reinterpret_cast<void*>(stream),
                    accumulate,
                    divisor,
                    randomOrder);
            },
            py::arg("rows_ptr"),
            py::arg("n"),
            py::arg("out_ptr"),
            py::arg("ld"),
            py::arg("col_off") = 0,
            py::arg("stream") = 0,
            py::arg("accumulate") = false,
            py::arg("divisor") = 0.0f,
            py::arg("random_order") = false);

    m.def("available_compression_strategies", &memb::availableCompressionStrategies);

    // Hooks for the reference's unit tests of the Builder's pieces
    // (src/kmeans_tests.cpp:9-38, src/bit_stream_tests.cpp:31-59).
    m.def("_kmeans_fit_predict", [](const std::vector<float>& data, size_t levels) {
        memb::KMeansClusterizer clusterizer(levels);
        clusterizer.fit(data);
        std::vector<uint8_t> assignments;
        clusterizer.predict(data.data(), data.size(), &assignments);
        return py::make_tuple(clusterizer.centroids(), std::vector<int>(assignments.begin(), assignments.end()));
    });
    m.def("_bit_pack", [](const std::vector<std::pair<uint32_t, uint32_t>>& codes) {
        memb::BitWriter writer;
        for (const auto& code : codes) {
            writer.push(code.first, code.second);
        }
        writer.flushToByte();
        return py::bytes(reinterpret_cast<const char*>(writer.bytes().data()), writer.bytes().size());
    });
    m.def("_huffman_description", [](const std::vector<uint64_t>& counts) {
        auto lengths = memb::huffmanCodeLengths(counts);
        std::vector<uint8_t> keys;
        std::vector<uint32_t> sizeOffsets;
        memb::decoderDescription(lengths, &keys, &sizeOffsets);
        return py::make_tuple(std::vector<int>(keys.begin(), keys.end()), sizeOffsets);
    });
    m.def("_decode_table", [](const std::vector<int>& keys, const std::vector<uint32_t>& sizeOffsets, uint32_t rootBitsLimit) {
        std::vector<uint8_t> narrow(keys.begin(), keys.end());
        auto lengths = memb::codeLengthsFromSizeOffsets(narrow.data(), narrow.size(), sizeOffsets.data(), sizeOffsets.size());
        auto table = memb::buildDecodeTable(lengths, rootBitsLimit);
        return py::make_tuple(table.rootBits, table.maxCodeBits, table.hasSubTables, table.entries);
    });

    // ReadersUnion 'concatenate' as one launch; false when the readers cannot share a kernel
    // (the caller then decodes them one by one). Pointers are device pointers.
    m.def(
        "union_rows_to_device",
        [](const std::vector<std::shared_ptr<memb::Reader>>& readers,
           const std::vector<uintptr_t>& rows,
           const std::vector<size_t>& colOffs,
           size_t n,
           uintptr_t out,
           size_t ld,
           uintptr_t stream,
           bool average)
        {
            if (readers.size() != rows.size() || readers.size() != colOffs.size() || readers.empty()) {
                throw std::runtime_error("One row-id array and one column offset per reader are needed");
            }
            std::vector<memb_hip_ctx*> contexts;
            std::vector<const uint32_t*> rowPointers;
            for (size_t i = 0; i < readers.size(); ++i) {
                contexts.push_back(readers[i]->deviceContext());
                rowPointers.push_back(reinterpret_cast<const uint32_t*>(rows[i]));
            }
            int code = memb_hip_decode_rows_union_device(
                contexts.data(), rowPointers.data(), colOffs.data(), readers.size(), n, reinterpret_cast<float*>(out), ld,
                reinterpret_cast<void*>(stream), average ? MEMB_HIP_UNION_AVERAGE : 0u);
            if (code == MEMB_HIP_UNSUPPORTED) {
                return false;
            }
            if (code != MEMB_HIP_OK) {
                throw std::runtime_error(memb_hip_last_error());
            }
            return true;
        },
        py::arg("readers"),
        py::arg("rows_ptrs"),
        py::arg("col_offs"),
        py::arg("n"),
        py::arg("out_ptr"),
        py::arg("ld"),
        py::arg("stream") = 0,
        py::arg("average") = false);
    // one batch of words resolved by several readers of one device (ReadersUnion): packed and copied once
    m.def(
        "union_words_to_rows_device",
        [](memb::WordBatch& batch,
           const py::sequence& wordList,
           const std::vector<std::shared_ptr<memb::Reader>>& readers,
           const std::vector<uintptr_t>& rows,
           uintptr_t stream)
        {
            if (readers.size() != rows.size()) {
                throw std::runtime_error("One row-id array per reader is needed");
            }
            std::vector<const memb::Reader*> models;
            std::vector<uint32_t*> targets;
            for (size_t i = 0; i < readers.size(); ++i) {
                {
                    py::gil_scoped_release release;   // (first call: copies the keys to HBM and builds the table)
                    readers[i]->stageWords();
                }
                models.push_back(readers[i].get());
                targets.push_back(reinterpret_cast<uint32_t*>(rows[i]));
            }
            // one launch per finished run of jobs: every word fetched and hashed once, probed in each reader's table
            return WordFiller::fill(batch, wordList, [&](size_t firstWord, size_t words) {
                memb::Reader::resolveRangeToDevice(models, batch, firstWord, words, targets, reinterpret_cast<void*>(stream));
            });
        },
        py::arg("batch"),
        py::arg("words"),
        py::arg("readers"),
        py::arg("rows_ptrs"),
        py::arg("stream") = 0);
    // measurement hook (bench.py's word_search block, tools/perf): seconds the calling thread and the pool spend
    // writing a list of str into a word batch's pinned memory -- the host work of a device word search
    m.def("_word_fill_seconds", [](memb::WordBatch& batch, const py::sequence& wordList) {
        const auto start = std::chrono::steady_clock::now();
        WordFiller::fill(batch, wordList, [](size_t, size_t) {});
        return std::chrono::duration<double>(std::chrono::steady_clock::now() - start).count();
    });
    m.def("_packed_fill_seconds", [](memb::WordBatch& batch, const py::buffer& bytes, const py::array_t<uint32_t, py::array::c_style>& offsets) {
        const py::buffer_info blob = bytes.request();
        const size_t count = static_cast<size_t>(offsets.shape(0)) - 1;
        const uint32_t* starts = offsets.data();
        const uint8_t* data = static_cast<const uint8_t*>(blob.ptr);
        const size_t dataBytes = static_cast<size_t>(blob.size);
        py::gil_scoped_release release;
        const auto start = std::chrono::steady_clock::now();
        PackedFiller::fill(batch, data, dataBytes, starts, count, [](size_t, size_t) {});
        return std::chrono::duration<double>(std::chrono::steady_clock::now() - start).count();
    });
    m.attr("HOST_DEVICE") = static_cast<int>(memb::CompressedStorage::HOST_DEVICE);
    m.def("_writer_mimics_official_layout", [](bool enabled) {
        memb::wire::BufferBuilder::omitDefaults() = enabled;
    });
    m.def("hip_device_count", []() {
        int count = 0;
        memb_hip_device_count(&count);
        return count;
    });
}

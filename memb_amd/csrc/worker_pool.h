// A small persistent thread pool for the host side of the batch path: word
// search (memb::Reader::resolveRows) and the copy of decoded rows out of pinned
// memory (memb_hip_decode_rows).
//
// The reference starts one std::async thread per job on every call
// (src/reader.cpp:65-84). On the hosts this runs on (hundreds of hardware
// threads) creating those threads costs more than the work of a 100 k-word
// batch, so the threads are kept.
//
// One batch at a time: start() hands out job(0) .. job(jobs - 1), each pool
// thread taking the next unclaimed index; wait() returns when all have
// returned. run() = start + help from the calling thread + wait. Callers
// serialise batches themselves (a mutex around start..wait).
#pragma once

#include <condition_variable>
#include <exception>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace memb {

class WorkerPool {
public:
    explicit WorkerPool(size_t threads)
    {
        for (size_t i = 0; i < threads; ++i) {
            threads_.emplace_back([this] { loop(); });
        }
    }

    ~WorkerPool()
    {
        {
            std::lock_guard<std::mutex> lock(mutex_);
            stop_ = true;
        }
        wake_.notify_all();
        for (auto& thread : threads_) {
            thread.join();
        }
    }

    WorkerPool(const WorkerPool&) = delete;
    WorkerPool& operator=(const WorkerPool&) = delete;

    size_t size() const { return threads_.size(); }

    // maxThreads: wake at most this many pool threads for the batch (0 = as many as there are jobs): work that is a
    // cache miss per item wants many threads for large batches and few for small ones, where rousing a thread costs
    // more than it then does
    void start(size_t jobs, std::function<void(size_t)> job, size_t maxThreads = 0)
    {
        {
            std::lock_guard<std::mutex> lock(mutex_);
            job_ = std::move(job);
            jobs_ = jobs;
            next_ = 0;
            remaining_ = jobs;
            error_ = nullptr;
        }
        // wake no more threads than there are jobs (a small batch must not pay for rousing everyone)
        const size_t wanted = maxThreads ? std::min(jobs, maxThreads) : jobs;
        if (wanted >= threads_.size()) {
            wake_.notify_all();
        } else {
            for (size_t i = 0; i < wanted; ++i) {
                wake_.notify_one();
            }
        }
    }

    // The calling thread takes jobs too, until none is left unclaimed.
    void help()
    {
        std::unique_lock<std::mutex> lock(mutex_);
        while (next_ < jobs_) {
            execute(lock);
        }
    }

    void wait()
    {
        std::unique_lock<std::mutex> lock(mutex_);
        done_.wait(lock, [this] { return remaining_ == 0; });
        job_ = nullptr;
        jobs_ = 0;
        if (error_) {
            std::exception_ptr error = error_;
            error_ = nullptr;
            std::rethrow_exception(error);
        }
    }

    void run(size_t jobs, std::function<void(size_t)> job, size_t maxThreads = 0)
    {
        start(jobs, std::move(job), maxThreads ? maxThreads - 1 : 0);   // (the calling thread is one of them)
        help();
        wait();
    }

private:
    // Takes the next job; called and returns with the lock held.
    void execute(std::unique_lock<std::mutex>& lock)
    {
        const size_t index = next_++;
        lock.unlock();
        std::exception_ptr error;
        try {
            job_(index);
        } catch (...) {
            error = std::current_exception();
        }
        lock.lock();
        if (error && !error_) {
            error_ = error;
        }
        if (--remaining_ == 0) {
            done_.notify_all();
        }
    }

    void loop()
    {
        std::unique_lock<std::mutex> lock(mutex_);
        for (;;) {
            wake_.wait(lock, [this] { return stop_ || next_ < jobs_; });
            if (stop_) {
                return;
            }
            execute(lock);
        }
    }

    std::mutex mutex_;
    std::condition_variable wake_;
    std::condition_variable done_;
    std::function<void(size_t)> job_;
    size_t jobs_ = 0;
    size_t next_ = 0;
    size_t remaining_ = 0;
    std::exception_ptr error_;
    bool stop_ = false;
    std::vector<std::thread> threads_;
};

}  // namespace memb

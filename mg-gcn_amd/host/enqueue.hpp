// enqueue.hpp -- one enqueue thread per GPU for the single-process, P-GPU host layer.
//
// The reference drives its P GPUs from ONE host thread: every distributed operator is a loop
// `for j: ctx[j].set(); launch` (src/cuda_utils.hpp:57-92, src/gcn.hpp:191-296).  An epoch of the
// 3x128 model is ~150 launches / event calls per GPU, ~0.8 ms of host time per GPU: at P = 8 the
// host needs 6.5 ms for 3 ms of device work per GPU.  Here every GPU has a
// command queue drained by its own thread: the operator loops of ops.hpp / gcn.hpp keep their
// shape (and the reference's names) but PUSH the per-GPU body instead of running it; the calling
// thread runs ahead, the P threads issue their GPU's launches, event records / waits, peer copies
// and RCCL calls side by side.  Per GPU the commands run in program order, so whatever the
// one-thread form enqueues on GPU j's streams is enqueued here, in the same order, by thread j:
// results are bit-identical.  Cross-GPU order exists only inside the collectives
// (include/mggcn_comm.h: per-rank entry points).
//
// Header-only and free of HIP so that a CPU test can run it under ThreadSanitizer
// (tests/test_enqueue_cpu.py).  Rules for the commands:
//   * capture by VALUE (matrices and contexts are shared handles, cheap to copy);
//   * never capture the dist_context (or anything else that owns the queues): the last owner must
//     die on the calling thread -- a queue cannot join itself;
//   * an exception thrown by a command is kept and rethrown by the next drain() on the caller.
#pragma once

#include <atomic>
#include <condition_variable>
#include <cstddef>
#include <deque>
#include <exception>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <utility>
#include <vector>

namespace mggcn {

class enqueue_queue {
    using command = std::function<void()>;
    std::mutex mu_;
    std::condition_variable work_cv_, idle_cv_;
    std::deque<command> q_;                 // guarded by mu_
    bool running_ = false;                  // the thread is inside a batch                (mu_)
    bool sleeping_ = false;                 // the thread waits on work_cv_                (mu_)
    bool stop_ = false;                     //                                             (mu_)
    std::exception_ptr error_;              // first exception of a command                (mu_)
    std::atomic<std::size_t> pushed_{0};    // lets the idle thread spin without the lock
    std::size_t taken_ = 0;                 // thread-private mirror of pushed_
    std::thread thread_;

    void loop(const std::function<void()> &init) {
        if (init) init();
        std::deque<command> batch;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(mu_);
                running_ = false;
                if (q_.empty()) {
                    idle_cv_.notify_all();
                    if (stop_) return;
                    // a short spin before sleeping: the producer pushes an epoch's commands in bursts, and a
                    // condition-variable wake-up costs more than most commands
                    lk.unlock();
                    for (int spin = 0; spin < 2000 && pushed_.load(std::memory_order_acquire) == taken_; spin++) std::this_thread::yield();
                    lk.lock();
                    while (q_.empty() && !stop_) {
                        sleeping_ = true;
                        work_cv_.wait(lk);
                        sleeping_ = false;
                    }
                    if (q_.empty()) { idle_cv_.notify_all(); return; }
                }
                batch.swap(q_);
                running_ = true;
            }
            taken_ += batch.size();
            for (auto &c : batch) {
                try {
                    c();
                } catch (...) {
                    std::lock_guard<std::mutex> lk(mu_);
                    if (!error_) error_ = std::current_exception();
                }
            }
            batch.clear();                  // command destructors (captured handles) run on this thread, outside the lock
        }
    }

public:
    // `init` runs first on the new thread (the host layer binds the thread to its GPU there)
    explicit enqueue_queue(std::function<void()> init = {}) : thread_([this, init] { loop(init); }) {}
    enqueue_queue(const enqueue_queue &) = delete;
    enqueue_queue &operator=(const enqueue_queue &) = delete;

    ~enqueue_queue() {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
        }
        work_cv_.notify_all();
        if (thread_.joinable()) thread_.join();      // pending commands still run: nothing enqueued is ever dropped
    }

    template <typename F>
    void push(F &&f) {
        bool wake;
        {
            std::lock_guard<std::mutex> lk(mu_);
            q_.emplace_back(std::forward<F>(f));
            pushed_.fetch_add(1, std::memory_order_release);
            wake = sleeping_;
        }
        if (wake) work_cv_.notify_one();
    }

    // returns once every command pushed before the call has run; rethrows the first exception a command threw
    void drain() {
        std::exception_ptr e;
        {
            std::unique_lock<std::mutex> lk(mu_);
            idle_cv_.wait(lk, [this] { return q_.empty() && !running_; });
            std::swap(e, error_);
        }
        if (e) std::rethrow_exception(e);
    }
};

// the P queues of one dist_context
class enqueue_pool {
    std::vector<std::unique_ptr<enqueue_queue>> queues_;

public:
    // init(j) runs first on thread j
    enqueue_pool(std::size_t P, const std::function<void(std::size_t)> &init) {
        for (std::size_t j = 0; j < P; j++) queues_.push_back(std::make_unique<enqueue_queue>([init, j] { if (init) init(j); }));
    }
    std::size_t size() const { return queues_.size(); }
    template <typename F>
    void push(std::size_t j, F &&f) { queues_[j]->push(std::forward<F>(f)); }
    void drain(std::size_t j) { queues_[j]->drain(); }
    // every queue is drained even when one of them reports an error (the first one is rethrown)
    void drain() {
        std::exception_ptr first;
        for (auto &q : queues_) {
            try {
                q->drain();
            } catch (...) {
                if (!first) first = std::current_exception();
            }
        }
        if (first) std::rethrow_exception(first);
    }
};

}  // namespace mggcn

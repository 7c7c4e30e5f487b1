// ps_pool.hpp — tiny persistent fork-join pool for the host side of a batch (query planner, K1d
// descriptors).  run(fn) executes fn(part, parts) on the workers and on the calling thread.
//
// The jobs are short (tens of microseconds per part) and come in bursts - a serving loop issues
// three or four per batch, a batch every ~0.6 ms - so a worker that has just finished a job keeps
// polling for the next one for ~150 us before it goes to sleep on the condition variable: waking a
// sleeping thread costs about as much as the whole job (measured: 50 us per run() with sleeping
// workers).  An idle pool sleeps.
#pragma once
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <exception>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace ps {

class Pool {
 public:
  explicit Pool(unsigned workers) {
    for (unsigned i = 0; i < workers; ++i) threads_.emplace_back([this, i] { loop(i + 1); });
  }
  ~Pool() {
    {
      std::lock_guard<std::mutex> l(mu_);
      stop_.store(true, std::memory_order_release);
      gen_.fetch_add(1, std::memory_order_release);
    }
    cv_.notify_all();
    for (auto& t : threads_) t.join();
  }
  unsigned size() const { return (unsigned)threads_.size() + 1; }

  // fn(part, parts) for part in [0, parts); parts == size().  Blocks until all parts are done.
  // An exception thrown by any part (e.g. bad_alloc in plan_query) is rethrown here.
  void run(const std::function<void(unsigned, unsigned)>& fn) {
    {
      std::lock_guard<std::mutex> l(mu_);  // orders fn_ / pending_ before the generation bump for sleepers
      fn_ = &fn;
      pending_.store((unsigned)threads_.size(), std::memory_order_relaxed);
      gen_.fetch_add(1, std::memory_order_release);
    }
    if (sleepers_.load(std::memory_order_acquire)) cv_.notify_all();
    std::exception_ptr mine;
    try {
      fn(0, size());
    } catch (...) {
      mine = std::current_exception();
    }
    // the workers are about done: poll briefly, then block
    for (int i = 0; i < 4000 && pending_.load(std::memory_order_acquire); ++i) relax();
    if (pending_.load(std::memory_order_acquire)) {
      std::unique_lock<std::mutex> l(mu_);
      done_.wait(l, [this] { return pending_.load(std::memory_order_acquire) == 0; });
    }
    fn_ = nullptr;
    std::exception_ptr err;
    {
      std::lock_guard<std::mutex> l(mu_);
      err = mine ? mine : err_;
      err_ = nullptr;
    }
    if (err) std::rethrow_exception(err);
  }

 private:
  static void relax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#else
    std::this_thread::yield();
#endif
  }
  void loop(unsigned id) {
    unsigned seen = 0;
    for (;;) {
      // hot phase: a new job usually follows within microseconds
      bool got = false;
      const auto t0 = std::chrono::steady_clock::now();
      for (int i = 0;; ++i) {
        if (gen_.load(std::memory_order_acquire) != seen) { got = true; break; }
        relax();
        if ((i & 255) == 255 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(150)) break;
      }
      if (!got) {
        std::unique_lock<std::mutex> l(mu_);
        sleepers_.fetch_add(1, std::memory_order_acq_rel);
        cv_.wait(l, [&] { return gen_.load(std::memory_order_acquire) != seen; });
        sleepers_.fetch_sub(1, std::memory_order_acq_rel);
      }
      seen = gen_.load(std::memory_order_acquire);
      if (stop_.load(std::memory_order_acquire)) return;
      const std::function<void(unsigned, unsigned)>* fn = fn_;
      std::exception_ptr e;
      try {
        if (fn) (*fn)(id, size());
      } catch (...) {
        e = std::current_exception();
      }
      if (e) {
        std::lock_guard<std::mutex> l(mu_);
        if (!err_) err_ = e;
      }
      if (pending_.fetch_sub(1, std::memory_order_acq_rel) == 1) {
        std::lock_guard<std::mutex> l(mu_);  // the caller may be about to block on done_
        done_.notify_one();
      }
    }
  }
  std::vector<std::thread> threads_;
  std::mutex mu_;
  std::condition_variable cv_, done_;
  const std::function<void(unsigned, unsigned)>* fn_ = nullptr;
  std::atomic<unsigned> gen_{0}, pending_{0}, sleepers_{0};
  std::atomic<bool> stop_{false};
  std::exception_ptr err_;
};

}  // namespace ps

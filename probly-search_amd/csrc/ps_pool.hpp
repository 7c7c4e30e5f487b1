// ps_pool.hpp — tiny persistent fork-join pool for the host query planner.
// parallel_for(n, fn) runs fn(chunk, n_chunks) on the workers and on the calling thread.
#pragma once
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
      stop_ = true;
      ++gen_;
    }
    cv_.notify_all();
    for (auto& t : threads_) t.join();
  }
  unsigned size() const { return (unsigned)threads_.size() + 1; }

  // fn(part, parts) for part in [0, parts); parts == size().  Blocks until all parts are done.
  void run(const std::function<void(unsigned, unsigned)>& fn) {
    std::unique_lock<std::mutex> l(mu_);
    fn_ = &fn;
    pending_ = (unsigned)threads_.size();
    ++gen_;
    l.unlock();
    cv_.notify_all();
    std::exception_ptr mine;
    try {
      fn(0, size());
    } catch (...) {
      mine = std::current_exception();
    }
    l.lock();
    done_.wait(l, [this] { return pending_ == 0; });
    fn_ = nullptr;
    // an exception thrown on a worker (e.g. bad_alloc in plan_query) surfaces on the calling thread
    std::exception_ptr err = mine ? mine : err_;
    err_ = nullptr;
    if (err) std::rethrow_exception(err);
  }

 private:
  void loop(unsigned id) {
    unsigned seen = 0;
    for (;;) {
      std::unique_lock<std::mutex> l(mu_);
      cv_.wait(l, [&] { return gen_ != seen; });
      seen = gen_;
      if (stop_) return;
      const std::function<void(unsigned, unsigned)>* fn = fn_;
      l.unlock();
      std::exception_ptr e;
      try {
        if (fn) (*fn)(id, size());
      } catch (...) {
        e = std::current_exception();
      }
      l.lock();
      if (e && !err_) err_ = e;
      if (--pending_ == 0) done_.notify_one();
    }
  }
  std::vector<std::thread> threads_;
  std::mutex mu_;
  std::condition_variable cv_, done_;
  const std::function<void(unsigned, unsigned)>* fn_ = nullptr;
  unsigned gen_ = 0, pending_ = 0;
  bool stop_ = false;
  std::exception_ptr err_;
};

}  // namespace ps

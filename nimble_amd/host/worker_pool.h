// Worker threads of the host layer's staging step (internal header).
#pragma once

#include <condition_variable>
#include <cstddef>
#include <cstdint>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace nimble_amd {
namespace detail {

// A few threads that stay around between batches: a stripe is staged in slices
// of 32 MiB, each a parallel copy of well under a millisecond, and starting
// sixteen threads takes a good part of that.  One job at a time; a caller that
// finds the pool busy (two batches staged at once) starts threads of its own.
class WorkerPool {
 public:
  static WorkerPool& instance() {
    static WorkerPool pool;
    return pool;
  }
  // job() on `workers` threads, the caller being one of them; false (nothing run) when the pool is taken
  bool run(size_t workers, const std::function<void()>& job) {
    std::unique_lock<std::mutex> turn(jobMu_, std::try_to_lock);
    if (!turn.owns_lock()) return false;
    {
      std::lock_guard<std::mutex> lock(mu_);
      while (threads_.size() + 1 < workers) threads_.emplace_back([this] { work(); });
      job_ = &job;
      wanted_ = workers - 1;
      taken_ = finished_ = 0;
      ++generation_;
    }
    wake_.notify_all();
    job();
    std::unique_lock<std::mutex> lock(mu_);
    done_.wait(lock, [&] { return finished_ == wanted_; });
    job_ = nullptr;
    return true;
  }

 private:
  WorkerPool() = default;
  ~WorkerPool() {
    {
      std::lock_guard<std::mutex> lock(mu_);
      stop_ = true;
    }
    wake_.notify_all();
    for (auto& t : threads_) t.join();
  }
  void work() {
    uint64_t seen = 0;
    std::unique_lock<std::mutex> lock(mu_);
    for (;;) {
      wake_.wait(lock, [&] { return stop_ || generation_ != seen; });
      if (stop_) return;
      seen = generation_;
      if (taken_ >= wanted_) continue;  // the job has the threads it asked for
      ++taken_;
      const std::function<void()>* job = job_;
      lock.unlock();
      (*job)();
      lock.lock();
      if (++finished_ == wanted_) done_.notify_all();
    }
  }
  std::mutex jobMu_, mu_;
  std::condition_variable wake_, done_;
  std::vector<std::thread> threads_;
  const std::function<void()>* job_ = nullptr;
  size_t wanted_ = 0, taken_ = 0, finished_ = 0;
  uint64_t generation_ = 0;
  bool stop_ = false;
};

// job() on `workers` threads, the caller being one of them: the pool's when it is free, else threads of its own
inline void parallelRun(size_t workers, const std::function<void()>& job) {
  if (workers <= 1) {
    job();
    return;
  }
  if (WorkerPool::instance().run(workers, job)) return;
  std::vector<std::thread> own;
  for (size_t w = 1; w < workers; ++w) own.emplace_back(job);
  job();
  for (auto& t : own) t.join();
}

}  // namespace detail
}  // namespace nimble_amd

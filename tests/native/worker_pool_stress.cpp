// Stress of the host layer's staging threads (nimble_amd/host/worker_pool.h), built with ThreadSanitizer by
// tests/test_worker_pool_cpu.py: jobs of changing width from one caller, then from three callers at once (one
// gets the pool, the others start threads of their own), every job a shared counter over a list of tasks as the
// staging step uses it.  Prints "ok <jobs>" when every task of every job ran exactly once.
#include <atomic>
#include <cstdio>
#include <thread>
#include <vector>

#include "worker_pool.h"

using nimble_amd::detail::parallelRun;

static bool one_job(size_t workers, size_t n_tasks) {
  std::vector<int> hits(n_tasks, 0);  // written by whichever thread takes the task: exactly one each
  std::atomic<size_t> next{0};
  parallelRun(workers, [&] {
    for (size_t k = next++; k < n_tasks; k = next++) ++hits[k];
  });
  for (int h : hits)
    if (h != 1) return false;
  return true;
}

int main() {
  size_t jobs = 0;
  for (int round = 0; round < 400; ++round) {
    const size_t workers = 1 + (size_t)(round * 7 % 17), n_tasks = (size_t)(round * 13 % 257);
    if (!one_job(workers, n_tasks)) { printf("failed: single caller, round %d\n", round); return 1; }
    ++jobs;
  }
  std::atomic<bool> bad{false};
  std::atomic<size_t> done{0};
  std::vector<std::thread> callers;
  for (int c = 0; c < 3; ++c)
    callers.emplace_back([&, c] {
      for (int round = 0; round < 150; ++round) {
        if (!one_job(2 + (size_t)((round + c) % 9), 64 + (size_t)(round % 50))) bad = true;
        ++done;
      }
    });
  for (auto& t : callers) t.join();
  if (bad) { printf("failed: concurrent callers\n"); return 1; }
  printf("ok %zu\n", jobs + done.load());
  return 0;
}

// End-to-end rate of the C++ host layer (tuning aid): GpuEncodingFactory::createBatch
// over a packed workload file + Encoding::materialize of every chunk.
// build:  g++ -O2 -std=c++17 -Iinclude -Inimble_amd/host tools/host_bench.cpp -o tools/bin/host_bench \
//             -Lnimble_amd/_lib -lnimble_amd_host -lnimble_amd -Wl,-rpath,'$ORIGIN/../../nimble_amd/_lib'
// input:  tools/dump_workload.py <workload> <prefix>  ->  <prefix>.bin, <prefix>.off
// usage:  host_bench <prefix> [iterations] [reader threads] [pipeline]
//         readers > 1: the chunks are materialised by that many threads, each with a buffer of its own, as the
//         reference's field readers run on an executor; pipeline = 1: the stripes (iterations) are decoded as a
//         reader's stripe loop would drive them -- createBatch of stripe n+1 runs on a loader thread while the
//         readers drain stripe n -- and one line gives the rate over all stripes
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <deque>
#include <chrono>
#include <cstdio>
#include <algorithm>
#include <array>
#include <cstdlib>
#include <memory>
#include <thread>
#include <vector>

#include "nimble_gpu.h"

using namespace nimble_amd;
using Clock = std::chrono::steady_clock;
static double ms(Clock::time_point a, Clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); }

int main(int argc, char** argv) {
  if (argc < 2) { fprintf(stderr, "usage: host_bench <prefix> [iterations]\n"); return 2; }
  const std::string prefix = argv[1];
  const int iters = argc > 2 ? atoi(argv[2]) : 5;
  const int readers = argc > 3 ? std::max(1, atoi(argv[3])) : 1;
  std::vector<char> buf;
  std::vector<uint64_t> off;
  {
    FILE* f = fopen((prefix + ".bin").c_str(), "rb");
    if (!f) { perror("bin"); return 1; }
    fseek(f, 0, SEEK_END); const long n = ftell(f); fseek(f, 0, SEEK_SET);
    buf.resize((size_t)n);
    if (fread(buf.data(), 1, (size_t)n, f) != (size_t)n) return 1;
    fclose(f);
    f = fopen((prefix + ".off").c_str(), "rb");
    if (!f) { perror("off"); return 1; }
    fseek(f, 0, SEEK_END); const long m = ftell(f); fseek(f, 0, SEEK_SET);
    off.resize((size_t)m / 8);
    if (fread(off.data(), 8, off.size(), f) != off.size()) return 1;
    fclose(f);
  }
  std::vector<std::string_view> chunks;
  for (size_t i = 0; i + 1 < off.size(); i += 2) chunks.emplace_back(buf.data() + off[i], off[i + 1]);
  const bool pipeline = argc > 4 && atoi(argv[4]) != 0;
  auto elem_size = [](const Encoding& e) -> size_t {
    const DataType t = e.dataType();
    return t == DataType::String ? sizeof(std::string_view) : t == DataType::Bool ? 1
           : (t == DataType::Int64 || t == DataType::Uint64 || t == DataType::Double) ? 8
           : (t == DataType::Int32 || t == DataType::Uint32 || t == DataType::Float) ? 4
           : (t == DataType::Int16 || t == DataType::Uint16) ? 2 : 1;
  };
  if (pipeline) {
    try {
      DecodingStats stats;
      Options fopts;
      fopts.decodingStats = &stats;
      GpuEncodingFactory factory(fopts);
      struct Stripe {
        std::vector<std::unique_ptr<Encoding>> encs;
        std::vector<std::unique_ptr<char[]>> strings;
      };
      std::mutex mu;
      std::condition_variable cv;
      std::deque<std::unique_ptr<Stripe>> ready;  // at most one decoded stripe waits while another is drained
      bool done = false;
      std::string error;
      { auto warm = factory.createBatch(chunks, [](uint32_t n) -> void* { static std::vector<std::unique_ptr<char[]>> keep; keep.push_back(std::make_unique<char[]>(n ? n : 1)); return keep.back().get(); }, factory.options()); }
      const auto t0 = Clock::now();
      // when each stripe's createBatch began and ended and when its drain began and ended, ms after t0
      std::vector<std::array<double, 4>> when((size_t)iters);
      std::thread loader([&] {
        try {
          for (int it = 0; it < iters; ++it) {
            auto st = std::make_unique<Stripe>();
            auto* strings = &st->strings;
            when[(size_t)it][0] = ms(t0, Clock::now());
            st->encs = factory.createBatch(chunks, [strings](uint32_t n) -> void* { strings->push_back(std::make_unique<char[]>(n ? n : 1)); return strings->back().get(); }, factory.options());
            when[(size_t)it][1] = ms(t0, Clock::now());
            std::unique_lock<std::mutex> lock(mu);
            cv.wait(lock, [&] { return ready.size() < 1; });
            ready.push_back(std::move(st));
            cv.notify_all();
          }
        } catch (const std::exception& e) { std::lock_guard<std::mutex> lock(mu); error = e.what(); }
        std::lock_guard<std::mutex> lock(mu);
        done = true;
        cv.notify_all();
      });
      uint64_t rows = 0, bytes = 0;
      std::vector<std::vector<uint8_t>> sinks((size_t)readers);
      // the first stripes also pay for the pinned blocks of every stripe in flight (hipHostMalloc of a few GB):
      // the steady-state rate is taken over the stripes after them
      constexpr int kWarm = 6;
      int drained = 0;
      Clock::time_point t_steady = t0;
      uint64_t bytes_at_steady = 0;
      for (;;) {
        std::unique_ptr<Stripe> st;
        {
          std::unique_lock<std::mutex> lock(mu);
          cv.wait(lock, [&] { return !ready.empty() || done; });
          if (ready.empty()) break;
          st = std::move(ready.front());
          ready.pop_front();
          cv.notify_all();
        }
        when[(size_t)drained][2] = ms(t0, Clock::now());
        std::atomic<size_t> next{0};
        std::atomic<uint64_t> r{0}, b{0};
        auto drain = [&](std::vector<uint8_t>& sink) {
          for (size_t c = next++; c < st->encs.size(); c = next++) {
            auto& e = st->encs[c];
            const uint32_t n = e->rowCount();
            const size_t es = elem_size(*e);
            if (sink.size() < (size_t)n * es) sink.resize((size_t)n * es);
            e->materialize(n, sink.data());
            r += n;
            b += (uint64_t)n * (e->dataType() == DataType::String ? 8 : es);
          }
        };
        std::vector<std::thread> pool;
        for (int t = 1; t < readers; ++t) pool.emplace_back([&, t] { drain(sinks[(size_t)t]); });
        drain(sinks[0]);
        for (auto& t : pool) t.join();
        rows += r.load();
        bytes += b.load();
        when[(size_t)drained][3] = ms(t0, Clock::now());
        if (++drained == kWarm) { t_steady = Clock::now(); bytes_at_steady = bytes; }
      }
      loader.join();
      if (!error.empty()) { fprintf(stderr, "error: %s\n", error.c_str()); return 1; }
      const auto t1 = Clock::now();
      printf("pipelined: %d stripes, %d readers: %.2f ms per stripe, rows %llu, decoded %.3f GB, end-to-end %.2f GB/s\n", iters, readers,
             ms(t0, t1) / iters, (unsigned long long)rows, bytes / 1e9, bytes / 1e6 / ms(t0, t1));
      if (drained > kWarm)
        printf("steady state (stripes %d..%d): %.2f ms per stripe, end-to-end %.2f GB/s\n", kWarm, drained - 1,
               ms(t_steady, t1) / (drained - kWarm), (bytes - bytes_at_steady) / 1e6 / ms(t_steady, t1));
      if (getenv("HOST_BENCH_TIMELINE"))
        for (int it = 0; it < drained; ++it)
          printf("stripe %2d: createBatch %8.2f .. %8.2f (%6.2f ms)   drained %8.2f .. %8.2f (%6.2f ms)\n", it, when[(size_t)it][0], when[(size_t)it][1],
                 when[(size_t)it][1] - when[(size_t)it][0], when[(size_t)it][2], when[(size_t)it][3], when[(size_t)it][3] - when[(size_t)it][2]);
      // what a batch costs the thread that makes it (stage, plan, launch, wait for the status) and the device
      const double nb = (double)std::max<uint64_t>(1, stats.decodeCPUTimeNanos.count());
      printf("per batch: host %.2f ms (min %.2f, max %.2f; planning %.2f ms, inflating %.2f ms), device %.3f ms, staged %.1f MB, decoded %.1f MB\n",
             stats.decodeCPUTimeNanos.sum() / nb / 1e6, stats.decodeCPUTimeNanos.min() / 1e6, stats.decodeCPUTimeNanos.max() / 1e6,
             stats.planNanos.sum() / nb / 1e6, stats.decompressCPUTimeNanos.sum() / nb / 1e6, stats.deviceNanos.sum() / nb / 1e6,
             stats.encodedBytes.sum() / nb / 1e6, stats.decodedBytes.sum() / nb / 1e6);
    } catch (const std::exception& e) {
      fprintf(stderr, "error: %s\n", e.what());
      return 1;
    }
    return 0;
  }
  try {
    GpuEncodingFactory factory;
    std::vector<std::unique_ptr<char[]>> strings;
    auto sbf = [&](uint32_t n) -> void* { strings.push_back(std::make_unique<char[]>(n ? n : 1)); return strings.back().get(); };
    std::vector<uint8_t> sink;
    for (int it = 0; it < iters; ++it) {
      strings.clear();
      const auto t0 = Clock::now();
      auto encs = factory.createBatch(chunks, sbf, factory.options());
      const auto t1 = Clock::now();
      std::atomic<uint64_t> rows{0}, bytes{0};
      std::atomic<size_t> next{0};
      auto read_chunks = [&](std::vector<uint8_t>& sink) {
        for (size_t c = next++; c < encs.size(); c = next++) {
          auto& e = encs[c];
          const uint32_t n = e->rowCount();
          const bool str = e->dataType() == DataType::String;
          const size_t es = str ? sizeof(std::string_view) : e->dataType() == DataType::Bool ? 1
                            : (e->dataType() == DataType::Int64 || e->dataType() == DataType::Uint64 || e->dataType() == DataType::Double) ? 8
                            : (e->dataType() == DataType::Int32 || e->dataType() == DataType::Uint32 || e->dataType() == DataType::Float) ? 4
                            : (e->dataType() == DataType::Int16 || e->dataType() == DataType::Uint16) ? 2 : 1;
          if (sink.size() < (size_t)n * es) sink.resize((size_t)n * es);
          e->materialize(n, sink.data());
          rows += n;
          bytes += (uint64_t)n * (str ? 8 : es);
        }
      };
      if (readers == 1) {
        read_chunks(sink);
      } else {
        std::vector<std::vector<uint8_t>> sinks((size_t)readers);
        std::vector<std::thread> pool;
        for (int t = 0; t < readers; ++t) pool.emplace_back([&, t] { read_chunks(sinks[(size_t)t]); });
        for (auto& t : pool) t.join();
      }
      const auto t2 = Clock::now();
      printf("iter %d: createBatch %.2f ms  materialize %.2f ms  rows %llu  decoded %.3f GB  end-to-end %.2f GB/s (createBatch only %.2f GB/s)\n", it,
             ms(t0, t1), ms(t1, t2), (unsigned long long)rows.load(), bytes.load() / 1e9, bytes.load() / 1e6 / ms(t0, t2), bytes.load() / 1e6 / ms(t0, t1));
    }
  } catch (const std::exception& e) {
    fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
  return 0;
}

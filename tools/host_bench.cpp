// End-to-end rate of the C++ host layer (tuning aid): GpuEncodingFactory::createBatch
// over a packed workload file + Encoding::materialize of every chunk.
// build:  g++ -O2 -std=c++17 -Iinclude -Inimble_amd/host tools/host_bench.cpp -o tools/bin/host_bench \
//             -Lnimble_amd/_lib -lnimble_amd_host -lnimble_amd -Wl,-rpath,'$ORIGIN/../../nimble_amd/_lib'
// input:  tools/dump_workload.py <workload> <prefix>  ->  <prefix>.bin, <prefix>.off
// usage:  host_bench <prefix> [iterations] [reader threads]   (readers > 1: the chunks are materialised by that
//         many threads, each with a buffer of its own, as the reference's field readers run on an executor)
#include <atomic>
#include <chrono>
#include <cstdio>
#include <algorithm>
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

// Times nbl_plan_batch on a dumped workload (tools/dump_workload.py): host only, no GPU.
//   g++ -O2 -std=c++17 -Iinclude tools/plan_bench.cpp -Lnimble_amd/_lib -lnimble_amd -Wl,-rpath,$PWD/nimble_amd/_lib -o /tmp/plan_bench
//   /tmp/plan_bench /tmp/wide [iterations]
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "nimble_amd.h"

static std::vector<uint8_t> slurp(const std::string& path) {
  FILE* f = std::fopen(path.c_str(), "rb");
  if (!f) { std::perror(path.c_str()); std::exit(2); }
  std::fseek(f, 0, SEEK_END);
  const long n = std::ftell(f);
  std::fseek(f, 0, SEEK_SET);
  std::vector<uint8_t> v((size_t)n);
  if (n && std::fread(v.data(), 1, (size_t)n, f) != (size_t)n) std::exit(2);
  std::fclose(f);
  return v;
}

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  const std::vector<uint8_t> bin = slurp(std::string(argv[1]) + ".bin");
  const std::vector<uint8_t> offb = slurp(std::string(argv[1]) + ".off");
  const size_t n = offb.size() / 16;
  std::vector<uint64_t> offs(n), sizes(n);
  const uint64_t* p = reinterpret_cast<const uint64_t*>(offb.data());
  for (size_t i = 0; i < n; ++i) { offs[i] = p[2 * i]; sizes[i] = p[2 * i + 1]; }
  const int iters = argc > 2 ? std::atoi(argv[2]) : 20;
  nbl_options opt{};
  opt.struct_size = sizeof(opt);
  double best = 1e9, sum = 0;
  nbl_plan_info info{};
  for (int it = 0; it < iters; ++it) {
    const auto t0 = std::chrono::steady_clock::now();
    nbl_plan* plan = nullptr;
    if (nbl_plan_batch(bin.data(), offs.data(), sizes.data(), (uint32_t)n, &opt, &plan) != NBL_OK) {
      std::fprintf(stderr, "plan failed: %s\n", nbl_last_error());
      return 1;
    }
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    nbl_plan_get_info(plan, &info);
    nbl_plan_free(plan);
    best = ms < best ? ms : best;
    sum += ms;
  }
  std::printf("%zu chunks, %u nodes, %u launches: plan %.2f ms best, %.2f ms mean (descriptors %.1f MB)\n", n, info.n_nodes,
              info.n_launches, best, sum / iters, info.descriptor_bytes / 1e6);
  return 0;
}

// GpuTabletDecoder: every stripe of a file over several GPUs (declared in
// tablet.h).  Kept apart from the file parser, which has no GPU dependency.
#include <algorithm>
#include <condition_variable>
#include <exception>
#include <mutex>
#include <thread>

#include "tablet.h"

namespace nimble_amd {

// ------------------------------------------------------------ GpuTabletDecoder
struct GpuTabletDecoder::State {
  std::mutex mu;
  std::condition_variable cv;
  std::vector<std::unique_ptr<GpuEncodingFactory>> factories;  // one per entry of devices_
  std::vector<std::unique_ptr<GpuStripeDecoder>> done;         // per stripe, until taken
  std::vector<std::exception_ptr> errors;                      // per stripe
  std::vector<uint8_t> ready, taken;                           // per stripe
  uint32_t lowestNotTaken = 0;
  uint32_t lookahead = 0;
  bool stop = false;
  std::vector<std::thread> threads;
};

std::vector<std::vector<uint32_t>> GpuTabletDecoder::assignStripes(const std::vector<uint64_t>& sizes, size_t devices) {
  if (devices == 0) throw NimbleError(NBL_ERR_ARGUMENT, "assignStripes needs at least one device");
  std::vector<uint32_t> order(sizes.size());
  for (uint32_t i = 0; i < order.size(); ++i) order[i] = i;
  std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return sizes[a] > sizes[b]; });  // ties keep index order
  std::vector<uint64_t> load(devices, 0);
  std::vector<std::vector<uint32_t>> owned(devices);
  for (uint32_t s : order) {
    size_t best = 0;
    for (size_t d = 1; d < devices; ++d)
      if (load[d] < load[best] || (load[d] == load[best] && owned[d].size() < owned[best].size())) best = d;
    owned[best].push_back(s);
    load[best] += sizes[s];
  }
  for (auto& o : owned) std::sort(o.begin(), o.end());
  return owned;
}

GpuTabletDecoder::GpuTabletDecoder(const Tablet& tablet, std::vector<int> devices, Options options, uint32_t lookahead)
    : tablet_(tablet), devices_(std::move(devices)), state_(std::make_unique<State>()) {
  if (devices_.empty()) throw NimbleError(NBL_ERR_ARGUMENT, "GpuTabletDecoder needs at least one device");
  const uint32_t n = tablet_.stripeCount();
  std::vector<uint64_t> sizes(n);
  for (uint32_t i = 0; i < n; ++i) sizes[i] = tablet_.stripeSize(i);
  assignment_ = assignStripes(sizes, devices_.size());
  owner_.assign(n, 0);
  for (size_t d = 0; d < assignment_.size(); ++d)
    for (uint32_t s : assignment_[d]) owner_[s] = (uint32_t)d;
  State& st = *state_;
  st.done.resize(n);
  st.errors.resize(n);
  st.ready.assign(n, 0);
  st.taken.assign(n, 0);
  st.lookahead = lookahead;
  for (int dev : devices_) st.factories.push_back(std::make_unique<GpuEncodingFactory>(options, dev));
  try {
    startThreads();
  } catch (...) {  // a thread could not be started: end the ones that were
    {
      std::lock_guard<std::mutex> lock(st.mu);
      st.stop = true;
    }
    st.cv.notify_all();
    for (std::thread& t : st.threads) t.join();
    throw;
  }
}

void GpuTabletDecoder::startThreads() {
  State& st = *state_;
  for (size_t d = 0; d < devices_.size(); ++d) {
    st.threads.emplace_back([this, d] {
      State& st = *state_;
      for (uint32_t stripe : assignment_[d]) {
        {
          // not further ahead of the reader than `lookahead` stripes of this device
          std::unique_lock<std::mutex> lock(st.mu);
          st.cv.wait(lock, [&] {
            if (st.stop || st.lookahead == 0) return true;
            uint32_t ahead = 0;
            for (uint32_t s : assignment_[d])
              if (s >= st.lowestNotTaken && s < stripe) ++ahead;
            return ahead < st.lookahead;
          });
          if (st.stop) return;
        }
        std::unique_ptr<GpuStripeDecoder> dec;
        std::exception_ptr err;
        try {
          std::vector<std::string_view> views;
          for (const auto& idAndBytes : tablet_.streams(stripe)) views.push_back(idAndBytes.second);
          dec = std::make_unique<GpuStripeDecoder>(*st.factories[d], views);
        } catch (...) {
          err = std::current_exception();
        }
        {
          std::lock_guard<std::mutex> lock(st.mu);
          st.done[stripe] = std::move(dec);
          st.errors[stripe] = err;
          st.ready[stripe] = 1;
        }
        st.cv.notify_all();
      }
    });
  }
}

GpuTabletDecoder::~GpuTabletDecoder() {
  {
    std::lock_guard<std::mutex> lock(state_->mu);
    state_->stop = true;
  }
  state_->cv.notify_all();
  for (std::thread& t : state_->threads) t.join();
}

std::vector<uint32_t> GpuTabletDecoder::streamIds(uint32_t stripe) const {
  std::vector<uint32_t> ids;
  for (const auto& idAndBytes : tablet_.streams(stripe)) ids.push_back(idAndBytes.first);
  return ids;
}

std::unique_ptr<GpuStripeDecoder> GpuTabletDecoder::take(uint32_t stripe) {
  State& st = *state_;
  if (stripe >= st.ready.size()) throw NimbleError(NBL_ERR_ARGUMENT, "stripe index out of range");
  std::unique_ptr<GpuStripeDecoder> dec;
  std::exception_ptr err;
  {
    std::unique_lock<std::mutex> lock(st.mu);
    if (st.taken[stripe]) throw NimbleError(NBL_ERR_ARGUMENT, "stripe taken twice");
    st.cv.wait(lock, [&] { return st.ready[stripe] != 0; });
    st.taken[stripe] = 1;
    dec = std::move(st.done[stripe]);
    err = st.errors[stripe];
    while (st.lowestNotTaken < st.taken.size() && st.taken[st.lowestNotTaken]) ++st.lowestNotTaken;
  }
  st.cv.notify_all();
  if (err) std::rethrow_exception(err);
  return dec;
}

}  // namespace nimble_amd

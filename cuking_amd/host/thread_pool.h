// Fixed-size worker pool + ParallelFor for the per-file decode/pack loop
// (role of the reference's ThreadPool/ParallelFor, cuking.cu:355-433), on
// std::thread / std::mutex / std::condition_variable only.
#ifndef CUKING_AMD_HOST_THREAD_POOL_H_
#define CUKING_AMD_HOST_THREAD_POOL_H_

#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace cuking_host {

// Runs func(i) for i in [begin, end) on up to `num_threads` threads, handing
// out indices dynamically.  func returns "" or an error message; after the
// first error the remaining indices are skipped.  Returns the first error
// recorded (like the reference, which one is unspecified under races).
inline std::string ParallelFor(size_t num_threads, size_t begin, size_t end,
                               const std::function<std::string(size_t)> &func) {
  if (end <= begin) return "";
  std::atomic<size_t> next(begin);
  std::atomic<bool> failed(false);
  std::mutex mu;
  std::string first_error;
  auto worker = [&]() {
    while (!failed.load(std::memory_order_relaxed)) {
      const size_t i = next.fetch_add(1, std::memory_order_relaxed);
      if (i >= end) return;
      std::string err = func(i);
      if (!err.empty()) {
        std::lock_guard<std::mutex> lock(mu);
        if (first_error.empty()) first_error = std::move(err);
        failed.store(true, std::memory_order_relaxed);
      }
    }
  };
  const size_t n = std::min(num_threads, end - begin);
  std::vector<std::thread> threads;
  threads.reserve(n);
  for (size_t t = 1; t < n; ++t) threads.emplace_back(worker);
  worker();  // the calling thread takes part
  for (auto &t : threads) t.join();
  return first_error;
}

}  // namespace cuking_host

#endif  // CUKING_AMD_HOST_THREAD_POOL_H_

// Minimal fork-join helper for the host-side per-block work (exact block cost,
// block splitting, encoding).  The reference is single-threaded; spreading the
// order-insensitive host work over cores removes the Amdahl wall once the
// device owns the O(bytes x iterations) part.
#pragma once
#include <atomic>
#include <cstdlib>
#include <thread>
#include <vector>

namespace zamd {

inline unsigned HostThreads() {
  static const unsigned n = [] {
    if (const char* e = std::getenv("ZOPFLI_AMD_THREADS")) {
      const int v = std::atoi(e);
      if (v > 0) return static_cast<unsigned>(v);
    }
    const unsigned hc = std::thread::hardware_concurrency();
    return hc ? hc : 1u;
  }();
  return n;
}

// Calls fn(i) for i in [0, n), dynamically load-balanced.  Nested calls run inline.
template <typename Fn>
void ParallelFor(size_t n, Fn&& fn) {
  static thread_local bool inside = false;
  const unsigned want = static_cast<unsigned>(n < HostThreads() ? n : HostThreads());
  if (want <= 1 || inside) {
    for (size_t i = 0; i < n; ++i) fn(i);
    return;
  }
  std::atomic<size_t> next{0};
  auto worker = [&] {
    inside = true;
    for (;;) {
      const size_t i = next.fetch_add(1, std::memory_order_relaxed);
      if (i >= n) break;
      fn(i);
    }
    inside = false;
  };
  std::vector<std::thread> threads;
  threads.reserve(want - 1);
  for (unsigned t = 1; t < want; ++t) threads.emplace_back(worker);
  worker();
  for (auto& t : threads) t.join();
}

}  // namespace zamd

// The library's own cache of large host blocks.
//
// With block splitting the symbols of every block pass through host arrays — a quarter of a million symbols per master
// block as (litlen, dist) pairs, byte positions and sampled histograms, hundreds of MB per 100 MB of input — that a few
// dozen worker threads allocate and free.  Left to glibc, every such block above the mmap threshold is a mmap / munmap
// pair and every heap that shrinks a madvise, each with a TLB shootdown on all the workers' CPUs: on incompressible input
// returning that memory once took as long as the compression.  Round 5 bought that back with mallopt(M_TRIM_THRESHOLD,
// M_TOP_PAD) — a setting of the HOST PROCESS's allocator, which a drop-in libzopfli.so.1 has no business changing (the
// reference has no side effects outside its arguments).  Instead the arrays that matter take their memory
// from here: blocks of a few size classes, handed back to a free list instead of to malloc, up to a budget
// (ZOPFLI_AMD_HOST_CACHE_MB, default 1024; 0 = no caching: plain malloc / free); beyond it a block is freed as usual.
// zmx_host_cache_trim() (include/zopfli_amd.h) gives everything cached back.  Nothing process-wide is touched.
#pragma once
#include <pthread.h>

#include <atomic>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <mutex>
#include <new>
#include <vector>

namespace zamd {

class BlockCache {
 public:
  static constexpr int kMinLog2 = 10;
  static constexpr int kClasses = 80;

  // smaller requests are malloc's own business (ZOPFLI_AMD_HOST_CACHE_MIN, bytes; at least 2^kMinLog2)
  static size_t MinBytes() {
    static const size_t m = [] {
      const char* e = std::getenv("ZOPFLI_AMD_HOST_CACHE_MIN");
      const long v = e ? std::atol(e) : (32l << 10);
      return v > (1l << kMinLog2) ? static_cast<size_t>(v) : static_cast<size_t>(1) << kMinLog2;
    }();
    return m;
  }

  // Size class of a request: capacities 2^k and 3 * 2^(k-2) (at most a third wasted), k >= kMinLog2.
  static int ClassOf(size_t bytes, size_t* cap) {
    int k = kMinLog2;
    while ((static_cast<size_t>(1) << k) < bytes) ++k;
    const size_t pow2 = static_cast<size_t>(1) << k;
    const size_t mid = (pow2 >> 2) * 3;           // 0.75 * 2^k
    if (k > kMinLog2 && mid >= bytes) {
      *cap = mid;
      return 2 * (k - kMinLog2) - 1;
    }
    *cap = pow2;
    return 2 * (k - kMinLog2);
  }

  static void* Take(size_t bytes) {
    if (bytes < MinBytes() || Budget() == 0) return std::malloc(bytes ? bytes : 1);
    size_t cap;
    const int c = ClassOf(bytes, &cap);
    if (c < kClasses) {
      State& s = Get();
      std::lock_guard<std::mutex> lock(s.mu);
      std::vector<void*>& fl = s.free_list[c];
      if (!fl.empty()) {
        void* p = fl.back();
        fl.pop_back();
        s.cached -= cap;
        return p;
      }
    }
    return std::malloc(cap);
  }

  static void Give(void* p, size_t bytes) {
    if (!p) return;
    if (bytes >= MinBytes() && Budget() != 0) {
      size_t cap;
      const int c = ClassOf(bytes, &cap);
      if (c < kClasses) {
        State& s = Get();
        std::lock_guard<std::mutex> lock(s.mu);
        if (s.cached + cap <= Budget()) {
          s.free_list[c].push_back(p);
          s.cached += cap;
          return;
        }
      }
    }
    std::free(p);
  }

  // Frees every cached block; returns the bytes given back.
  static size_t Trim() {
    State& s = Get();
    std::vector<void*> all;
    size_t bytes = 0;
    {
      std::lock_guard<std::mutex> lock(s.mu);
      for (auto& fl : s.free_list) {
        all.insert(all.end(), fl.begin(), fl.end());
        fl.clear();
      }
      bytes = s.cached;
      s.cached = 0;
    }
    for (void* p : all) std::free(p);
    return bytes;
  }

  static size_t CachedBytes() {
    State& s = Get();
    std::lock_guard<std::mutex> lock(s.mu);
    return s.cached;
  }

 private:
  struct State {
    std::mutex mu;
    std::vector<void*> free_list[kClasses];
    size_t cached = 0;
  };
  static size_t Budget() {
    static const size_t b = [] {
      const char* e = std::getenv("ZOPFLI_AMD_HOST_CACHE_MB");
      const long mb = e ? std::atol(e) : 1024;
      return mb > 0 ? static_cast<size_t>(mb) << 20 : static_cast<size_t>(0);
    }();
    return b;
  }
  // (leaked on purpose, like the worker pools; a forked child starts with an empty cache and a fresh mutex — a thread of
  //  the parent may have held the old one at the fork)
  static State& Get() {
    static std::atomic<State*> slot{nullptr};
    State* s = slot.load(std::memory_order_acquire);
    if (s) return *s;
    static std::mutex create;
    std::lock_guard<std::mutex> lock(create);
    s = slot.load(std::memory_order_acquire);
    if (!s) {
      s = new State();
      slot.store(s, std::memory_order_release);
      static std::once_flag atfork;
      std::call_once(atfork, [] {
        pthread_atfork(nullptr, nullptr, [] {
          new (&create) std::mutex();
          slot.store(new State(), std::memory_order_release);
        });
      });
    }
    return *s;
  }
};

// std::allocator's interface over the cache: for the vectors named above.
template <class T>
struct CachedAlloc {
  using value_type = T;
  CachedAlloc() = default;
  template <class U>
  CachedAlloc(const CachedAlloc<U>&) {}
  T* allocate(size_t n) {
    void* p = BlockCache::Take(n * sizeof(T));
    if (!p) throw std::bad_alloc();
    return static_cast<T*>(p);
  }
  void deallocate(T* p, size_t n) { BlockCache::Give(p, n * sizeof(T)); }
  template <class U>
  bool operator==(const CachedAlloc<U>&) const { return true; }
  template <class U>
  bool operator!=(const CachedAlloc<U>&) const { return false; }
};

template <class T>
using CVec = std::vector<T, CachedAlloc<T>>;

}  // namespace zamd

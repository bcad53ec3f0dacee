// Fork-join helper for the host-side per-block work (exact block cost, block
// splitting, cost model, encoding).  The reference is single-threaded; spreading
// the order-insensitive host work over cores removes the Amdahl wall once the
// device owns the O(bytes x iterations) part.
//
// The workers are persistent: the per-run cost-model step is called 2 x
// numiterations times per batch with ~20 us of work per block, and spawning a
// thread per core per call cost more than the work itself.
#pragma once
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <functional>
#include <memory>
#include <mutex>
#include <system_error>
#include <thread>
#include <vector>

#include <pthread.h>
#include <sched.h>

#include <cstdio>
#include <cstring>

namespace zamd {

// The processors this process may run on: the online count cut down by the scheduling affinity.  A
// cgroup CPU quota is deliberately NOT applied: the host phases are bursts of a few milliseconds, and
// a quota of n CPUs' worth of time per 100 ms period still lets such a burst spread over every core
// (measured on the 256-thread box with a 16-CPU quota: the split and encode phases take 2.5x longer
// on 16 threads than on 64).
inline unsigned UsableCpus() {
  static const unsigned n = [] {
    unsigned hc = std::thread::hardware_concurrency();
    if (hc == 0) hc = 1;
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) {
      const unsigned a = static_cast<unsigned>(CPU_COUNT(&set));
      if (a > 0 && a < hc) hc = a;
    }
    return hc;
  }();
  return n;
}

// The cgroup CPU quota (cpu.max) in CPUs, 0 = none.  The pools do not stop at it — see above — but they stop at FOUR
// times it: on a 256-thread host a container limited to 2 CPUs would otherwise wake 128 threads per fork-join and
// be throttled for whole periods (the measurements behind "do not apply the quota" were taken at 16 CPUs: 64 threads).
inline unsigned CpuQuota() {
  static const unsigned q = [] {
    unsigned v = 0;
    if (std::FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
      char a[32] = {0};
      long period = 0;
      if (std::fscanf(f, "%31s %ld", a, &period) == 2 && std::strcmp(a, "max") != 0 && period > 0) {
        const long quota = std::atol(a);
        if (quota > 0) v = static_cast<unsigned>((quota + period - 1) / period);
      }
      std::fclose(f);
    }
    return v;
  }();
  return q;
}
inline unsigned QuotaCap(unsigned n) {
  const unsigned q = CpuQuota();
  return q && 4 * q < n ? 4 * q : n;
}

inline unsigned HostThreads() {
  static const unsigned n = [] {
    if (const char* e = std::getenv("ZOPFLI_AMD_THREADS")) {
      const int v = std::atoi(e);
      if (v > 0) return static_cast<unsigned>(v);
    }
    const unsigned hc = UsableCpus();
    // beyond this the wake-up cost outweighs the per-block work (the cost model of a block's next run takes a few
    // microseconds since the package-merge rewrite; measured per 100 MB request, 15 runs: 2.8 ms of cost-model phase
    // on 32 threads, 3.8 on 16, 3.9 on 64)
    const unsigned cap = 32;
    return QuotaCap(hc < cap ? hc : cap);
  }();
  return n;
}

// Threads of the second, wider pool: for the few long per-master-block phases of a call (block
// split, block type costs, encoding), where one task per thread beats two rounds on fewer threads
// and the wake-up cost is paid three times per call, not twice per squeeze run.
inline unsigned WideThreads() {
  static const unsigned n = [] {
    if (const char* e = std::getenv("ZOPFLI_AMD_WIDE_THREADS")) {    // (for measuring)
      const int v = std::atoi(e);
      if (v > 0) return static_cast<unsigned>(v);
    }
    if (std::getenv("ZOPFLI_AMD_THREADS")) return HostThreads();   // an explicit budget covers both pools
    const unsigned hc = UsableCpus();
    const unsigned cap = 128;
    const unsigned w = QuotaCap(hc < cap ? hc : cap);
    return w > HostThreads() ? w : HostThreads();
  }();
  return n;
}

// (Round 4 gave every shard thread of a call a wide pool of its own — a pool took one job at a time, and the block-split
//  searches of three contexts stood in line — which left a drop-in caller's process with three times WideThreads()
//  sleeping threads.  The wide pool now takes any number of jobs at once: WidePool below.)

class WorkerPool {
 public:
  // The pools are created on first use and leaked on purpose (no join at exit).  A forked child has
  // none of the worker threads: it forgets the parent's pools and makes its own on first use.
  static WorkerPool& Get() { return Instance(0, HostThreads()); }

  // Runs body(i) for i in [0, n) on the workers and the calling thread; returns
  // when all are done.  One job at a time (callers are serialised).
  void Run(size_t n, const std::function<void(size_t)>& body) {
    std::lock_guard<std::mutex> serial(run_mutex_);
    {
      std::lock_guard<std::mutex> lock(mutex_);
      body_ = &body;
      n_ = n;
      next_.store(0, std::memory_order_relaxed);
      pending_.store(workers_.size(), std::memory_order_relaxed);
      generation_.fetch_add(1, std::memory_order_release);
    }
    wake_.notify_all();
    Drain(body);
    // (the workers finish within microseconds of each other: look before going to sleep)
    for (int i = 0; i < spin_ && pending_.load(std::memory_order_acquire) != 0; ++i) CpuRelax();
    if (pending_.load(std::memory_order_acquire) != 0) {
      std::unique_lock<std::mutex> lock(mutex_);
      done_.wait(lock, [&] { return pending_.load(std::memory_order_acquire) == 0; });
    }
    body_ = nullptr;
  }

 private:
  static std::atomic<WorkerPool*>* Slots() {
    static std::atomic<WorkerPool*> slots[1];
    return slots;
  }
  static std::mutex& CreateMutex() {
    static std::mutex* m = new std::mutex();   // (leaked: usable in a forked child whatever the parent held)
    return *m;
  }
  static void ForgetInChild() {
    Slots()[0].store(nullptr);
    new (&CreateMutex()) std::mutex();         // the parent may have forked while holding it
  }
  static WorkerPool& Instance(int which, unsigned threads) {
    WorkerPool* p = Slots()[which].load(std::memory_order_acquire);
    if (p) return *p;
    std::lock_guard<std::mutex> lock(CreateMutex());
    p = Slots()[which].load(std::memory_order_acquire);
    if (!p) {
      static std::once_flag atfork;
      std::call_once(atfork, [] { pthread_atfork(nullptr, nullptr, &WorkerPool::ForgetInChild); });
      p = new WorkerPool(threads, which == 0 ? kSpin : 0);   // (the wide pool's 127 workers go to sleep at once: they would burn a cgroup quota spinning)
      Slots()[which].store(p, std::memory_order_release);
    }
    return *p;
  }

  WorkerPool(unsigned threads, int spin) : spin_(spin) {
    const unsigned extra = threads > 1 ? threads - 1 : 0;
    workers_.reserve(extra);
    // (a pids cgroup or RLIMIT_NPROC makes std::thread throw: the pool then works with the threads it got — the
    //  caller's own included, Run() never depends on a count — instead of aborting somebody else's process)
    for (unsigned t = 0; t < extra; ++t) {
      try {
        workers_.emplace_back([this] { Loop(); });
      } catch (const std::system_error&) {
        break;
      }
    }
    for (auto& w : workers_) w.detach();
  }

  void Drain(const std::function<void(size_t)>& body) {
    for (;;) {
      const size_t i = next_.fetch_add(1, std::memory_order_relaxed);
      if (i >= n_) break;
      body(i);
    }
  }

  // A fork-join of a few dozen microseconds of work follows the last one within microseconds where it matters (the
  // rounds of the block-split search, the two cost-model steps of a squeeze run): a worker that goes to sleep on the
  // condition variable at once pays a futex wake-up — and 31 or 127 of them a thundering herd on one mutex — per
  // round.  Each looks at the generation counter for a few tens of microseconds first.
  static constexpr int kSpin = 1500;
  static void CpuRelax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#else
    std::this_thread::yield();
#endif
  }

  void Loop() {
    uint64_t seen = 0;
    for (;;) {
      bool got = false;
      for (int i = 0; i < spin_; ++i) {
        if (generation_.load(std::memory_order_acquire) != seen) { got = true; break; }
        CpuRelax();
      }
      if (!got) {
        std::unique_lock<std::mutex> lock(mutex_);
        wake_.wait(lock, [&] { return generation_.load(std::memory_order_acquire) != seen; });
      }
      seen = generation_.load(std::memory_order_acquire);
      const std::function<void(size_t)>* body = body_;   // (written before the generation moved)
      if (body) Drain(*body);
      if (pending_.fetch_sub(1, std::memory_order_acq_rel) == 1) {
        std::lock_guard<std::mutex> lock(mutex_);     // (the caller may be between its check and its wait)
        done_.notify_one();
      }
    }
  }

  std::vector<std::thread> workers_;
  std::mutex run_mutex_, mutex_;
  std::condition_variable wake_, done_;
  const std::function<void(size_t)>* body_ = nullptr;
  size_t n_ = 0;
  std::atomic<size_t> next_{0};
  std::atomic<size_t> pending_{0};
  const int spin_;
  std::atomic<uint64_t> generation_{0};
};

// The wide pool: WideThreads() workers shared by every caller, ANY NUMBER OF JOBS AT ONCE.  Its tasks are long (a block-split
// search, a master block's encoding: milliseconds), so a mutex per task taken is nothing, and a call dealt over three
// contexts has its three shard threads' jobs served side by side by the same workers — each job gets the whole pool while
// the others have nothing to do.  Created on first use, leaked on purpose, forgotten in a forked child (as WorkerPool).
class WidePool {
 public:
  static WidePool& Get() {
    WidePool* p = Slot().load(std::memory_order_acquire);
    if (p) return *p;
    std::lock_guard<std::mutex> lock(CreateMutex());
    p = Slot().load(std::memory_order_acquire);
    if (!p) {
      static std::once_flag atfork;
      std::call_once(atfork, [] { pthread_atfork(nullptr, nullptr, &WidePool::ForgetInChild); });
      p = new WidePool(WideThreads());
      Slot().store(p, std::memory_order_release);
    }
    return *p;
  }

  void Run(size_t n, const std::function<void(size_t)>& body) {
    auto job = std::make_shared<Job>();
    job->body = &body;
    job->n = n;
    {
      std::lock_guard<std::mutex> lock(mu_);
      jobs_.push_back(job);
    }
    // (a small job — the nine probes of a split search's round — wakes as many sleepers as it has tasks to give away,
    //  not all hundred-odd: a round is a few hundred microseconds of work)
    if (n > 16) {
      wake_.notify_all();
    } else {
      for (size_t k = 1; k < n; ++k) wake_.notify_one();
    }
    // the caller works on its own job, then waits for the tasks others took
    for (;;) {
      const size_t i = job->next.fetch_add(1, std::memory_order_relaxed);
      if (i >= n) break;
      body(i);
      job->done.fetch_add(1, std::memory_order_acq_rel);
    }
    {
      std::unique_lock<std::mutex> lock(mu_);
      for (auto it = jobs_.begin(); it != jobs_.end(); ++it) {
        if (it->get() == job.get()) { jobs_.erase(it); break; }
      }
      done_.wait(lock, [&] { return job->done.load(std::memory_order_acquire) >= n; });
    }
  }

 private:
  struct Job {
    const std::function<void(size_t)>* body = nullptr;
    size_t n = 0;
    std::atomic<size_t> next{0}, done{0};
  };
  static std::atomic<WidePool*>& Slot() {
    static std::atomic<WidePool*> slot{nullptr};
    return slot;
  }
  static std::mutex& CreateMutex() {
    static std::mutex* m = new std::mutex();
    return *m;
  }
  static void ForgetInChild() {
    Slot().store(nullptr);
    new (&CreateMutex()) std::mutex();
  }
  explicit WidePool(unsigned threads) {
    const unsigned extra = threads > 1 ? threads - 1 : 0;
    for (unsigned t = 0; t < extra; ++t) {
      try {
        std::thread([this] { Loop(); }).detach();
      } catch (const std::system_error&) {
        break;      // (a pids cgroup / RLIMIT_NPROC: the callers do the work themselves)
      }
    }
  }
  void Loop() {
    size_t turn = 0;
    for (;;) {
      std::shared_ptr<Job> job;
      size_t i = 0;
      {
        std::unique_lock<std::mutex> lock(mu_);
        for (;;) {
          // a job that still has tasks to hand out, the jobs in turn
          const size_t nj = jobs_.size();
          for (size_t k = 0; k < nj && !job; ++k) {
            const std::shared_ptr<Job>& c = jobs_[(turn + k) % nj];
            const size_t at = c->next.load(std::memory_order_relaxed);
            if (at < c->n) {
              i = c->next.fetch_add(1, std::memory_order_relaxed);
              if (i < c->n) job = c;
            }
          }
          if (job) break;
          idle_.fetch_add(1, std::memory_order_relaxed);
          wake_.wait(lock);
          idle_.fetch_sub(1, std::memory_order_relaxed);
        }
        ++turn;
      }
      // (tasks of one job in a row without the lock while it has any)
      for (;;) {
        (*job->body)(i);
        const bool last = job->done.fetch_add(1, std::memory_order_acq_rel) + 1 >= job->n;
        if (last) {
          std::lock_guard<std::mutex> lock(mu_);
          done_.notify_all();
        }
        i = job->next.fetch_add(1, std::memory_order_relaxed);
        if (i >= job->n) break;
      }
    }
  }

  std::mutex mu_;
  std::condition_variable wake_, done_;
  std::vector<std::shared_ptr<Job>> jobs_;
  std::atomic<int> idle_{0};     // workers with nothing to do (a nested fork-join is only worth registering when there are any)

 public:
  int Idle() const { return idle_.load(std::memory_order_relaxed); }
};

inline thread_local bool g_inside_parallel_for = false;
// The calling thread's host phases run on the calling thread alone (api.cc: a small call among other calls in flight).
inline thread_local bool g_host_inline = false;
inline thread_local int g_nested_level = 0;       // ParallelForNested: one level only

// Calls fn(i) for i in [0, n), dynamically load-balanced.  Nested calls run inline.
template <typename Fn>
void ParallelFor(size_t n, Fn&& fn) {
  bool& inside = g_inside_parallel_for;
  if (n <= 1 || HostThreads() <= 1 || inside || g_host_inline) {
    for (size_t i = 0; i < n; ++i) fn(i);
    return;
  }
  const std::function<void(size_t)> body = [&](size_t i) {
    const bool was = g_inside_parallel_for;   // (thread-local of the executing thread)
    g_inside_parallel_for = true;
    fn(i);
    g_inside_parallel_for = was;
  };
  WorkerPool::Get().Run(n, body);
}

// The same on the wide pool when there are more tasks than the regular pool has threads.
template <typename Fn>
void ParallelForWide(size_t n, Fn&& fn) {
  if (n <= HostThreads() || g_inside_parallel_for || g_host_inline) {
    ParallelFor(n, fn);
    return;
  }
  const std::function<void(size_t)> body = [&](size_t i) {
    const bool was = g_inside_parallel_for;
    g_inside_parallel_for = true;
    fn(i);
    g_inside_parallel_for = was;
  };
  // (always the wide pool, also where it is no wider than the regular one — a host of <= 32 CPUs, ZOPFLI_AMD_THREADS set:
  //  WorkerPool takes one job at a time, and the shard threads of a dealt call would queue their split searches and
  //  encodes behind each other and behind the others' cost-model fork-joins)
  WidePool::Get().Run(n, body);
}

// A fork-join INSIDE a task of a parallel loop — the nine probes of a round of one master block's split search
// (block_split.cc) while the loop runs a master block per task.  On the wide pool, which serves any number of jobs at
// once: while every worker has a master block of its own the caller does its nine probes itself, as before; when the loop
// comes to its tail — a few master blocks whose searches take ten times the others' (PNG-like data: the first master
// blocks' 27 + 21 ms against 3 + 0.1) — the idle workers take the probes and the tail is up to nine times shorter.  Deeper
// levels run inline.
template <typename Fn>
void ParallelForNested(size_t n, Fn&& fn) {
  if (!g_inside_parallel_for) {        // (not nested after all)
    ParallelFor(n, fn);
    return;
  }
  // (no idle worker: every one has a master block of its own — or the host's CPUs are all taken, as with incompressible data,
  //  where registering a job per round only added switches: 327 -> 272 MB/s)
  if (n <= 1 || WideThreads() <= 1 || g_host_inline || g_nested_level > 0 || WidePool::Get().Idle() < 2) {
    for (size_t i = 0; i < n; ++i) fn(i);
    return;
  }
  const std::function<void(size_t)> body = [&](size_t i) {
    const int was = g_nested_level;
    const bool was_inside = g_inside_parallel_for;     // (a worker that took the task: whatever it does inside runs inline)
    g_nested_level = 1;
    g_inside_parallel_for = true;
    fn(i);
    g_inside_parallel_for = was_inside;
    g_nested_level = was;
  };
  ++g_nested_level;
  WidePool::Get().Run(n, body);
  --g_nested_level;
}

}  // namespace zamd

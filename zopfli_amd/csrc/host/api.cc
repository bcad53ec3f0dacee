// C ABI of libzopfli_amd.so, part 1: the reference's public surface
// (zopfli.h:67,86; deflate.h:58,67; gzip_container.h:42; zlib_container.h:42),
// plus the resident-input stream entry points used by bench.py and by the
// multi-GPU gather.
#include <chrono>
#include <malloc.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "block_cache.h"
#include "block_cost.h"
#include "deflate.h"
#include "lz77_optimal.h"
#include "symbols.h"
#include "deal.h"
#include "thread_pool.h"
#include "zopfli_amd.h"

// implemented by the device layer: kernel-only seconds / launches of the squeeze kernel
extern "C" void zmx_internal_kernel_stats(double* seconds3, double* squeeze_launches, int reset);
extern "C" void zmx_internal_seg_stats(double* out8, int reset);
extern "C" void zmx_internal_match_stats(double* out4, int reset);
extern "C" void zmx_internal_match5_stats(double* out3, int reset);
extern "C" void zmx_internal_stats_take(double* out19);
extern "C" void zmx_internal_stats_add(const double* in19);
// implemented by the device layer: size of the resident input
extern "C" size_t zmx_internal_input_size(zmx_ctx* ctx);
// implemented by the device layer: the caller's host copy of the resident input (borrowed)
extern "C" const unsigned char* zmx_internal_input_host(zmx_ctx* ctx);
extern "C" void zmx_internal_set_error(const char* msg);

namespace {

using zamd::kMasterBlock;

[[noreturn]] void Die(const char* what) {
  std::fprintf(stderr, "zopfli_amd: %s: %s\n", what, zmx_last_error());
  std::exit(EXIT_FAILURE);
}

// OPT-IN ONLY (ZOPFLI_AMD_KEEP_HEAP=1 / 2; unset or 0: malloc is left alone): mallopt so that glibc neither trims its
// arenas nor shrinks the worker threads' heaps.  Round 5 made this the default (+12 % with block splitting) and the review
// was right to object: it reconfigures the allocator of the whole host process, for good — a drop-in libzopfli.so.1 must
// not (the reference has no side effects outside its arguments).  Round 6: the host arrays that caused the
// churn — the blocks' symbol stores, positions, sampled histograms, bit buffers — take their memory from the library's
// own block cache (block_cache.h, ZOPFLI_AMD_HOST_CACHE_MB), which touches nothing process-wide.  The switch stays for
// measuring one against the other.
void MaybeKeepHeap() {
  static const bool once = [] {
    const char* e = std::getenv("ZOPFLI_AMD_KEEP_HEAP");
    if (!e || std::atoi(e) == 0) return true;
    if (std::atoi(e) == 2) {          // (for measuring: large blocks from the heap too — 32 MB is the most glibc takes; no better, nor is 1 MB)
      mallopt(M_MMAP_THRESHOLD, 32 << 20);
      mallopt(M_TRIM_THRESHOLD, 1 << 30);
      mallopt(M_TOP_PAD, 64 << 20);
    } else {
      mallopt(M_TRIM_THRESHOLD, 1 << 30);
      mallopt(M_TOP_PAD, 256 << 20);
    }
    return true;
  }();
  (void)once;
}

// The device contexts of the Zopfli* entry points.
//
// Which devices: ONE by default — ZOPFLI_AMD_DEVICE, else LOCAL_RANK (one process per GPU under torchrun), else
// device 0: a program that links libzopfli.so.1 must not find itself holding every GPU of the node.  Several only
// when asked: ZOPFLI_AMD_DEVICES = "all", a count, or a comma separated list of HIP device indices (an index may
// repeat: two contexts on one device, which is how the multi-device path is exercised on a one-GPU box); master
// blocks are independent (deflate.c:916-923), so a request with several of them is dealt across those devices.
//
// Re-entrancy (the reference has no globals: callers may run concurrent calls on distinct buffers): a
// device has up to ZOPFLI_AMD_LANES contexts (default 3), created when first needed; a request takes one free
// context on each device it uses and gives them back when it is done, so callers overlap — one's host phases
// (cost models, block splitting, merging) with the others' kernels — and a fourth waits.  A device whose context
// cannot be created (not gfx950, out of memory) is dropped from the list; only when none is left does the call die.
class ContextPool {
 public:
  // one free context on each of up to `want` devices (at least one), in device order; with `per_device` > 1 up to
  // that many free contexts of every device it uses (a large request on one device is dealt over two of its
  // contexts: one half's host phases run beside the other half's kernels)
  // `polite`: a call that is not large takes several contexts of a device only while it is the only caller — with other
  // calls in flight (holding contexts or waiting for one) it takes one, as every call below the dealing threshold does.
  // `small`: a call below the 32 master blocks from which calls are dealt whatever else runs (small files — what zopfli is
  // mostly used on — and the medium calls that, alone, politely take all three dealing contexts: a second such caller
  // no longer waits for the first to finish).  When every context of its
  // device is busy such a call gets a context of its own beyond the ZOPFLI_AMD_LANES of the dealing — up to
  // ZOPFLI_AMD_SMALL_LANES (16) per device — instead of waiting: sixteen callers with 64 KiB files keep eight streams of
  // small kernels and eight host threads' split searches going, where three contexts left thirteen of them waiting.
  std::vector<zmx_ctx*> Acquire(size_t want, size_t per_device = 1, std::vector<int>* device_of = nullptr,
                                bool polite = false, bool small = false) {
    std::unique_lock<std::mutex> lock(mu_);
    Init();
    ++in_flight_;
    // ... and it does not CREATE the further contexts before the eighth such call of the process (ZOPFLI_AMD_DEAL_AFTER):
    // a context costs ~ 50 ms to set up and saves such a call 5 - 15 ms, which a program that compresses a few files
    // and exits never earns back (zopflipng on one 1024 x 1024 image: 0.62 -> 0.70 s when its calls set up two more
    // contexts); a long-lived caller pays once.  (The first context a call takes is created whenever none is free.)
    static const size_t deal_after = [] {
      const char* e = std::getenv("ZOPFLI_AMD_DEAL_AFTER");
      return e ? static_cast<size_t>(std::max(0, std::atoi(e))) : static_cast<size_t>(8);
    }();
    const bool may_create_more = !polite || per_device <= 1 || ++polite_wishes_ >= deal_after;
    for (;;) {
      if (polite && in_flight_ > 1) per_device = 1;
      std::vector<Slot*> slots;
      size_t used_devices = 0;
      for (auto& dev : devices_) {
        if (used_devices == want) break;
        if (dev.dead) continue;
        size_t here = 0;
        for (size_t lane = 0; lane < per_device; ++lane) {
          Slot* s = nullptr;
          for (auto& sl : dev.slots) {
            if (!sl->busy && std::find(slots.begin(), slots.end(), sl.get()) == slots.end()) { s = sl.get(); break; }
          }
          if (!s && (dev.slots.size() < lanes_ || (small && per_device == 1 && dev.slots.size() < small_lanes_)) &&
              (may_create_more || lane == 0)) {
            // a new context: the slot is taken now, the context is created below without the pool's lock (HIP
            // start-up, streams, events: up to seconds on first use, and every Release would wait behind it)
            dev.slots.emplace_back(new Slot{nullptr, false, &dev});
            s = dev.slots.back().get();
          }
          if (!s) break;
          slots.push_back(s);
          ++here;
        }
        if (here) ++used_devices;
      }
      if (!slots.empty()) {
        for (Slot* s : slots) s->busy = true;
        bool need_create = false;
        for (Slot* s : slots) need_create |= s->ctx == nullptr;
        if (need_create) {
          lock.unlock();
          std::vector<std::pair<Slot*, zmx_ctx*>> made;
          std::vector<std::pair<Slot*, std::string>> failed;
          for (Slot* s : slots) {
            if (s->ctx) continue;
            zmx_ctx* c = nullptr;
            if (zmx_ctx_create(s->dev->index, &c) != 0) failed.emplace_back(s, zmx_last_error());
            else made.emplace_back(s, c);
          }
          lock.lock();
          for (auto& m : made) {
            m.first->ctx = m.second;
            // budgets per DEVICE, not per context: every context of the device gets its share of what a lone
            // context would keep cached / spend on one batch's DP edges
            size_t same = 0;
            for (auto& d : devices_) same += d.index == m.first->dev->index ? 1 : 0;
            zmx_ctx_set_share(m.second, static_cast<unsigned>(lanes_ * same));
          }
          for (auto& f : failed) {
            Device* dev = f.first->dev;
            for (size_t i = 0; i < dev->slots.size(); ++i) {
              if (dev->slots[i].get() == f.first) { dev->slots.erase(dev->slots.begin() + static_cast<long>(i)); break; }
            }
            slots.erase(std::find(slots.begin(), slots.end(), f.first));
            if (dev->slots.empty()) {
              std::fprintf(stderr, "zopfli_amd: device %d is not usable: %s\n", dev->index, f.second.c_str());
              dev->dead = true;
            }
          }
          if (!failed.empty()) cv_.notify_all();
        }
      }
      bool any_alive = false;
      for (auto& dev : devices_) any_alive |= !dev.dead;
      if (!any_alive) Die("no usable gfx950 device (there is no CPU fallback)");
      if (!slots.empty()) {
        std::vector<zmx_ctx*> got;
        for (Slot* s : slots) got.push_back(s->ctx);
        if (device_of) {
          device_of->clear();
          for (Slot* s : slots) device_of->push_back(s->dev->index);
        }
        return got;
      }
      cv_.wait(lock);   // every context of every device is busy
    }
  }
  // zmx_set_oom_hook: a context of `device` is out of memory even after dropping its own cache — the idle contexts of
  // that device give their cached arrays back
  void TrimIdle(int device) {
    // hipFree synchronises the device: not under the pool's lock (every Acquire / Release would wait behind it).  The
    // idle contexts are taken out of circulation, trimmed, and put back.
    std::vector<Slot*> mine;
    {
      std::lock_guard<std::mutex> lock(mu_);
      for (auto& dev : devices_) {
        if (dev.index != device) continue;
        for (auto& sl : dev.slots) if (!sl->busy && sl->ctx) { sl->busy = true; mine.push_back(sl.get()); }
      }
    }
    if (mine.empty()) return;
    for (Slot* sl : mine) zmx_ctx_trim_cache(sl->ctx);
    {
      std::lock_guard<std::mutex> lock(mu_);
      for (Slot* sl : mine) sl->busy = false;
    }
    cv_.notify_all();
  }
  size_t InFlight() {
    std::lock_guard<std::mutex> lock(mu_);
    return in_flight_;
  }
  void Release(const std::vector<zmx_ctx*>& ctxs) {
    {
      std::lock_guard<std::mutex> lock(mu_);
      if (in_flight_) --in_flight_;
      for (auto& dev : devices_)
        for (auto& sl : dev.slots)
          if (std::find(ctxs.begin(), ctxs.end(), sl->ctx) != ctxs.end()) sl->busy = false;
    }
    cv_.notify_all();
  }

 private:
  struct Device;
  struct Slot { zmx_ctx* ctx; bool busy; Device* dev; };
  struct Device { int index; bool dead = false; std::vector<std::unique_ptr<Slot>> slots; };
  void Init() {
    if (!devices_.empty()) return;
    MaybeKeepHeap();
    const int visible = zmx_device_count();
    std::vector<int> list;
    if (const char* e = std::getenv("ZOPFLI_AMD_DEVICES")) {
      if (std::strcmp(e, "all") == 0) {
        for (int i = 0; i < visible; ++i) list.push_back(i);
      } else if (std::strchr(e, ',')) {
        for (const char* p = e; *p;) {
          list.push_back(std::atoi(p));
          const char* q = std::strchr(p, ',');
          if (!q) break;
          p = q + 1;
        }
      } else {
        const int n = std::atoi(e);
        for (int i = 0; i < n && i < visible; ++i) list.push_back(i);
        if (list.empty()) list.push_back(0);   // ("0": no count — device 0)
      }
    } else if (const char* e = std::getenv("ZOPFLI_AMD_DEVICE")) {
      list.push_back(std::atoi(e));
    } else if (const char* r = std::getenv("LOCAL_RANK")) {
      // (one process per GPU under torchrun; with HIP_VISIBLE_DEVICES set per rank every rank sees ONE device and
      //  LOCAL_RANK = k would name a device that is not there: take it modulo what is visible)
      const int k = std::atoi(r);
      list.push_back(visible > 0 && k >= 0 ? k % visible : k);
    } else {
      list.push_back(0);
    }
    for (int d : list) {
      if (d < 0 || d >= visible) {
        std::fprintf(stderr, "zopfli_amd: no HIP device %d (%d visible): ignored\n", d, visible);
        continue;
      }
      Device dev;
      dev.index = d;
      devices_.push_back(std::move(dev));
    }
    if (devices_.empty()) {
      zmx_internal_set_error("no HIP device to run on");
      Die("no usable gfx950 device (there is no CPU fallback)");
    }
    if (const char* e = std::getenv("ZOPFLI_AMD_LANES")) lanes_ = static_cast<size_t>(std::max(1, std::atoi(e)));
    if (const char* e = std::getenv("ZOPFLI_AMD_SMALL_LANES")) small_lanes_ = static_cast<size_t>(std::max(1, std::atoi(e)));
    small_lanes_ = std::max(small_lanes_, lanes_);
    zmx_set_oom_hook(&ContextPool::OomHook);
  }
  static void OomHook(int device);
  std::mutex mu_;
  std::condition_variable cv_;
  std::vector<Device> devices_;
  size_t lanes_ = 3;
  size_t small_lanes_ = 16;  // contexts per device that calls of one or two master blocks may bring into being (1000 x 64 KiB through 16 callers: 8.9 MB/s with 3, 20.4 with 8, 27.7 with 16; profiles/r06_small_files.txt)
  size_t in_flight_ = 0;     // calls between Acquire and Release
  size_t polite_wishes_ = 0; // polite calls so far that asked for more than one context of a device
};

ContextPool& Pool() {
  static ContextPool* pool = new ContextPool();   // (never destroyed: HIP may be gone by the time statics are)
  return *pool;
}
void ContextPool::OomHook(int device) { Pool().TrimIdle(device); }

// ZOPFLI_AMD_TRACE_CALL=1: where a Zopfli* call's wall time goes, per shard and for the call (stderr)
bool TraceCall() {
  static const bool on = [] { const char* e = std::getenv("ZOPFLI_AMD_TRACE_CALL"); return e && std::atoi(e) != 0; }();
  return on;
}
double WallMs() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

struct Lease {
  std::vector<int> device_of;      // HIP device index of ctxs[i]
  std::vector<zmx_ctx*> ctxs;
  explicit Lease(size_t want, size_t per_device = 1, bool polite = false, bool small = false)
      : ctxs(Pool().Acquire(want, per_device, &device_of, polite, small)) {}
  ~Lease() { Pool().Release(ctxs); }
};

size_t PartsPerBatch() {
  static const size_t n = [] {
    if (const char* e = std::getenv("ZOPFLI_AMD_PARTS_PER_BATCH")) {
      const long v = std::atol(e);
      if (v > 0) return static_cast<size_t>(v);
    }
    return static_cast<size_t>(256);  // ~40 MB of tables per 1 MB master block
  }();
  return n;
}

std::vector<zamd::Part> MasterBlocks(size_t insize, bool final) {
  // deflate.c:916-923: do { ... } while (i < insize), so an empty input still
  // yields one (empty) part.
  std::vector<zamd::Part> parts;
  size_t i = 0;
  do {
    const bool masterfinal = i + kMasterBlock >= insize;
    const size_t size = masterfinal ? insize - i : kMasterBlock;
    parts.push_back({i, i + size, final && masterfinal});
    i += size;
  } while (i < insize);
  return parts;
}

int RunParts(zmx_ctx* ctx, const ZopfliOptions& options, int btype, const std::vector<zamd::Part>& parts,
             std::vector<zamd::Chunk>* chunks) {
  size_t step = PartsPerBatch();
  for (size_t a = 0; a < parts.size();) {
    const size_t b = a + step < parts.size() ? a + step : parts.size();
    std::vector<zamd::Part> group(parts.begin() + static_cast<long>(a), parts.begin() + static_cast<long>(b));
    std::vector<zamd::Chunk> got;
    const int rc = zamd::DeflateParts(ctx, options, btype, group, &got);
    if (rc == -2 && b - a > 1) {   // the DP edges of the batch do not fit the device layer's budget: smaller batches
      step = (b - a + 1) / 2;
      continue;
    }
    if (rc) return rc;
    for (auto& c : got) chunks->push_back(std::move(c));
    a = b;
  }
  return 0;
}

// The parts of one request (positions relative to `in`) dealt over the shared contexts in contiguous
// runs, one host thread per device; each device gets its parts' bytes plus the 32 KiB before them
// (all a part reads: lz77.c:551-552).  Chunks come back in stream order, stored chunks carrying
// positions relative to `in`.
//
// `sum` (optional): the container's checksum over in[0, sum->limit), taken on the devices from the bytes
// they hold anyway — each device its own parts' bytes, put together in stream order.
struct ChecksumRequest {
  int kind;         // ZMX_CRC32 / ZMX_ADLER32
  size_t limit;     // bytes covered (the parts must start at 0 and cover them)
  uint32_t value;
};

int RunPartsShardedOnce(const ZopfliOptions& options, int btype, const unsigned char* in,
                        const std::vector<zamd::Part>& parts, std::vector<zamd::Chunk>* chunks,
                        ChecksumRequest* sum = nullptr) {
  // (ZOPFLI_AMD_SPLIT_MB: from this many master blocks on, a request is dealt over ZOPFLI_AMD_SPLIT_WAYS = 3 contexts of
  //  each device — measured on 100 MB of text: 2 ways 123.2 ms, 3 ways 121.1, 4 ways 140; with block splitting 225 / 197 / 231;
  //  0 = never.  The GPU idles while the host computes a hundred cost models between two squeeze runs — 6 % of a
  //  100 MB call — and through the whole block-split search; two halves fill each other's gaps.)
  // Round 5, from how many master blocks on (profiles/r05_split_from.txt): with block splitting from 4 — the contexts'
  // split searches fall beside each other's kernels: 4 MB of text 33.7 -> 29.2 ms, 8 MB 46.5 -> 37.4, 12 MB 59.8 -> 43.6,
  // 24 MB 97.4 -> 62.8 (it was 32 until then); without block splitting it is worth 3 - 6 % from 12 MB on and nothing
  // below: from 16.
  static const long split_from_env = [] {
    const char* e = std::getenv("ZOPFLI_AMD_SPLIT_MB");
    return e ? static_cast<long>(std::max(0, std::atoi(e))) : -1L;
  }();
  const size_t split_from = split_from_env >= 0 ? static_cast<size_t>(split_from_env)
                                                : (options.blocksplitting && btype == 2 ? 4 : 16);
  static const size_t split_ways = [] {
    const char* e = std::getenv("ZOPFLI_AMD_SPLIT_WAYS");
    return e ? static_cast<size_t>(std::max(1, std::atoi(e))) : static_cast<size_t>(3);
  }();
  // Data with long runs of equal bytes: there the squeeze runs wait for a few very long single-wave tasks (zmx_dp5.h)
  // and most of the device idles — but a second context's tasks on the same SIMDs slow exactly those tasks (round 3,
  // class Z: 61 -> 35 MB/s on two contexts, so such data stayed on one).  Round 5: dealt all the same, with the contexts
  // at three stream PRIORITIES (as calls with block splitting are, below): the first context's long tasks win their
  // SIMDs, the others fill what it leaves — class Z 131 -> 190 MB/s, class M 165 -> 245 (189 / 245 on four, 187 / 250 on
  // six contexts; without the priorities 112 / 170; profiles/r05_runs_ctx.txt).  Sampled: one probe every 4096 bytes,
  // "the next 64 bytes are equal"; 1 % of the probes make a call "data with runs".
  // (ZOPFLI_AMD_SPLIT_RUNS=0 or ZOPFLI_AMD_STREAM_PRIO=0: such data on one context, as before — for measuring)
  bool runs = false, one_context = false;
  if (split_from && parts.size() >= split_from && in != nullptr) {
    const size_t lo = parts.front().instart, hi = parts.back().inend;
    size_t probes = 0, hits = 0;
    for (size_t i = lo; i + 64 <= hi; i += 4096, ++probes) {
      const unsigned char c0 = in[i];
      size_t k = 1;
      while (k < 64 && in[i + k] == c0) ++k;
      hits += k == 64;
    }
    runs = probes > 0 && hits * 100 >= probes;
    static const int prio_on = [] { const char* e = std::getenv("ZOPFLI_AMD_STREAM_PRIO"); return e ? std::atoi(e) : 1; }();
    static const bool split_runs = [] { const char* e = std::getenv("ZOPFLI_AMD_SPLIT_RUNS"); return !e || std::atoi(e) != 0; }();
    one_context = runs && !(prio_on && split_runs);
  }
  const double tr_begin = WallMs();
  const Lease lease(parts.size(), split_from && parts.size() >= split_from && !one_context ? split_ways : 1,
                    /*polite=*/parts.size() < 32, /*small=*/parts.size() < 32);
  const double tr_lease = WallMs();
  // A small call among other calls in flight (many small files, a caller thread each): its host phases — the split
  // searches' rounds of nine probes, the cost models of a block or two — run on the calling thread.  The worker pool takes
  // one fork-join at a time: sixteen callers queueing for it, each job a few microseconds of work per woken thread, were
  // slower than three (profiles/r06_small_files.txt); the callers are the parallelism.
  struct InlineHostWork {
    bool on, was;
    explicit InlineHostWork(bool o) : on(o), was(zamd::g_host_inline) { if (on) zamd::g_host_inline = true; }
    ~InlineHostWork() { if (on) zamd::g_host_inline = was; }
  } inline_host(parts.size() <= 2 && Pool().InFlight() > 1);
  const std::vector<zmx_ctx*>& ctxs = lease.ctxs;
  const size_t ndev = std::min(ctxs.size(), parts.size());
  struct Shard {
    size_t first = 0, last = 0, base = 0;
    std::vector<zamd::Chunk> chunks;
    int rc = 0;
    std::string err;
    int err_class = ZMX_ERR_NONE;   // zmx_last_error_class() of the failure
    zamd::Timing timing;
    uint32_t sum = 0;
    size_t sum_bytes = 0;
    bool redone = false;
    double stats[19] = {0};        // the shard thread's kernel / match / task statistics (zmx_internal_stats_take)
  };
  std::vector<Shard> shards(ndev);
  for (size_t d = 0; d < ndev; ++d) {
    shards[d].first = parts.size() * d / ndev;
    shards[d].last = parts.size() * (d + 1) / ndev;
  }
  // Shards of equal COST, not of equal count (deal.h): on a mixed corpus the master blocks of long runs of equal bytes
  // cost several times the others, and contiguous equal-count shards leave them to one or two contexts.  From the
  // bytes alone — the one-process-per-GPU launchers compute the same ranges (zmx_master_block_costs).
  // (ZOPFLI_AMD_DEAL=count: equal counts, as before — for measuring)
  static const bool deal_by_cost = [] { const char* e = std::getenv("ZOPFLI_AMD_DEAL"); return !e || std::strcmp(e, "count") != 0; }();
  if (deal_by_cost && ndev > 1 && in != nullptr && parts.size() > ndev) {
    std::vector<double> cost(parts.size());
    zamd::ParallelFor(parts.size(), [&](size_t i) { cost[i] = zamd::MasterBlockCost(in, parts[i].instart, parts[i].inend); });
    std::vector<size_t> first;
    zamd::DealByCost(cost, ndev, &first);
    for (size_t d = 0; d < ndev; ++d) { shards[d].first = first[d]; shards[d].last = first[d + 1]; }
  }
  // (ZOPFLI_AMD_SHARD_WEIGHTS="28,36,36": the shares of the shards, for measuring)
  if (const char* e = std::getenv("ZOPFLI_AMD_SHARD_WEIGHTS")) {
    std::vector<double> w;
    for (const char* q = e; *q;) {
      char* end = nullptr;
      const double v = std::strtod(q, &end);
      if (end == q) break;
      w.push_back(v > 0 ? v : 0);
      q = *end == ',' ? end + 1 : end;
    }
    double total = 0;
    for (size_t d = 0; d < ndev && d < w.size(); ++d) total += w[d];
    if (w.size() >= ndev && total > 0) {
      double acc = 0;
      size_t at = 0;
      for (size_t d = 0; d < ndev; ++d) {
        acc += w[d];
        size_t to = d + 1 == ndev ? parts.size() : static_cast<size_t>(parts.size() * acc / total + 0.5);
        to = std::max(to, at + 1);                       // no empty shard
        to = std::min(to, parts.size() - (ndev - 1 - d));
        shards[d].first = at;
        shards[d].last = to;
        at = to;
      }
    }
  }
  // The contexts of ONE device take their bytes over the same link: asked for at once, three uploads end together and
  // the device has nothing to do until then (12 ms of a 122 ms call on 100 MB; the kernel timeline of
  // tools/r04_timeline.sh).  One after the other, in stream order, the first context computes while the second's bytes
  // travel — and the contexts stay out of step from there on: their host phases (block split, cost models) fall
  // beside the others' kernels instead of beside each other.
  struct UploadOrder {
    std::mutex mu;
    std::condition_variable cv;
    std::vector<char> done;
    std::vector<long> after;     // the shard whose upload this one waits for, -1 = none
  } order;
  order.done.assign(ndev, 0);
  order.after.assign(ndev, -1);
  // With block splitting the contexts of one device run at three stream priorities — one after the other instead of
  // side by side: their split searches and joins then fall beside the others' kernels (zmx_ctx_set_priority; 100 MB of
  // text 152 -> 145 ms, without block splitting 123 -> 130: there every context stays on its default streams).
  // Data with runs: with or without block splitting (above).
  // (ZOPFLI_AMD_STREAM_PRIO=0: never; 2: always — for measuring)
  static const int use_priorities = [] { const char* e = std::getenv("ZOPFLI_AMD_STREAM_PRIO"); return e ? std::atoi(e) : 1; }();
  std::vector<int> shard_priority(ndev, 0);
  if (use_priorities && ((options.blocksplitting && btype == 2) || runs || use_priorities == 2)) {
    for (size_t d = 0; d < ndev; ++d) {
      size_t before = 0, same = 0;
      for (size_t e = 0; e < ndev; ++e) {
        if (lease.device_of[e] != lease.device_of[d]) continue;
        ++same;
        if (e < d) ++before;
      }
      if (same > 1) shard_priority[d] = before == 0 ? 1 : before + 1 == same ? -1 : 0;
    }
  }
  // (ZOPFLI_AMD_UPLOAD_ORDER=0: all at once, as before — for measuring)
  static const bool ordered_uploads = [] { const char* e = std::getenv("ZOPFLI_AMD_UPLOAD_ORDER"); return !e || std::atoi(e) != 0; }();
  for (size_t d = 1; d < ndev && ordered_uploads; ++d) {
    for (size_t e = d; e-- > 0;) {
      if (lease.device_of[e] == lease.device_of[d]) { order.after[d] = static_cast<long>(e); break; }
    }
  }
  struct UploadTurn {     // marks a shard's upload as over, however its attempt ends
    UploadOrder& o; size_t d; bool released = false;
    void Wait() {
      if (o.after[d] < 0) return;
      std::unique_lock<std::mutex> lock(o.mu);
      o.cv.wait(lock, [&] { return o.done[static_cast<size_t>(o.after[d])] != 0; });
    }
    void Release() {
      if (released) return;
      released = true;
      { std::lock_guard<std::mutex> lock(o.mu); o.done[d] = 1; }
      o.cv.notify_all();
    }
    ~UploadTurn() { Release(); }
  };
  // (ZOPFLI_AMD_TEST_FAIL_SHARD=k: the k-th shard's first attempt fails before it does anything — the test of the
  //  re-queue below)
  static const long fail_shard = [] { const char* e = std::getenv("ZOPFLI_AMD_TEST_FAIL_SHARD"); return e ? std::atol(e) : -1L; }();
  auto work = [&](size_t d, zmx_ctx* ctx, bool retry) {
    Shard& sh = shards[d];
    UploadTurn turn{order, d};
    int level = retry ? 0 : shard_priority[d];
    // (a call that is one shard — a small file — keeps the level its context was given when first seen: the contexts of
    //  concurrent small callers then lie on streams of all three priorities, i.e. on three sets of hardware queues instead of
    //  one — the runtime gives a priority four queues, and sixteen streams on four queues run their short kernels one after
    //  the other: 1 000 files of 64 KiB through 16 callers 26.8 -> 29.9 MB/s, 160 of 1 MB 220 -> 283, profiles/r06_small_hwq.txt;
    //  ZOPFLI_AMD_SMALL_PRIO=0: every such call on the default priority, as before)
    static const bool small_prio = [] { const char* e = std::getenv("ZOPFLI_AMD_SMALL_PRIO"); return !e || std::atoi(e) != 0; }();
    if (small_prio && ndev == 1 && !retry) {
      static std::mutex mu;
      static std::unordered_map<zmx_ctx*, int> levels;
      std::lock_guard<std::mutex> g(mu);
      auto it = levels.find(ctx);
      if (it == levels.end()) it = levels.emplace(ctx, static_cast<int>(levels.size() % 3) - 1).first;
      level = it->second;
    }
    if (zmx_ctx_set_priority(ctx, level) != 0) {
      // (not fatal: the context stays on the streams it has, the shards then run side by side instead of in turn)
      std::fprintf(stderr, "zopfli_amd: stream priorities unavailable (%s)\n", zmx_last_error());
    }
    sh.rc = 0;
    sh.err.clear();
    sh.err_class = ZMX_ERR_NONE;
    sh.chunks.clear();
    sh.sum = 0;
    sh.sum_bytes = 0;
    if (!retry && fail_shard == static_cast<long>(d)) {
      sh.rc = -1;
      sh.err = "injected failure (ZOPFLI_AMD_TEST_FAIL_SHARD): PoolAlloc(pool) too large — the text must not matter";
      sh.err_class = ZMX_ERR_OUT_OF_MEMORY;
      return;
    }
    const size_t start = parts[sh.first].instart, end = parts[sh.last - 1].inend;
    sh.base = start > zamd::kWindow ? start - zamd::kWindow : 0;
    const double tr0 = WallMs();
    if (!retry) turn.Wait();
    const double tr1 = WallMs();
    const int up = zmx_set_input(ctx, in + sh.base, end - sh.base);
    turn.Release();
    const double tr2 = WallMs();
    if (up != 0) {
      sh.rc = -1;
      sh.err = zmx_last_error();
      sh.err_class = zmx_last_error_class();
      return;
    }
    if (sum && start < sum->limit) {
      sh.sum_bytes = std::min(end, sum->limit) - start;
      if (zmx_checksum(ctx, sum->kind, start - sh.base, start - sh.base + sh.sum_bytes, &sh.sum) != 0) {
        sh.rc = -1;
        sh.err = zmx_last_error();
        sh.err_class = zmx_last_error_class();
        return;
      }
    }
    std::vector<zamd::Part> mine(parts.begin() + static_cast<long>(sh.first), parts.begin() + static_cast<long>(sh.last));
    for (auto& p : mine) { p.instart -= sh.base; p.inend -= sh.base; }
    const double tr3 = WallMs();
    sh.rc = RunParts(ctx, options, btype, mine, &sh.chunks);
    if (sh.rc) { sh.err = zmx_last_error(); sh.err_class = zmx_last_error_class(); }
    if (TraceCall()) {
      std::fprintf(stderr, "  shard %zu (%zu parts): start +%.2f ms, wait for turn %.2f, upload %.2f, checksum %.2f, parts %.2f, end +%.2f\n",
                   d, sh.last - sh.first, tr0 - tr_begin, tr1 - tr0, tr2 - tr1, tr3 - tr2, WallMs() - tr3, WallMs() - tr_begin);
    }
    for (auto& c : sh.chunks) {
      if (c.kind == zamd::Chunk::kStored) { c.start += sh.base; c.end += sh.base; }
    }
    sh.timing = zamd::ThreadTiming();
    if (d != 0 && !retry) zmx_internal_stats_take(sh.stats);   // (a thread of its own: its sums go to the caller's below)
  };
  if (ndev == 1) {
    work(0, ctxs[0], false);
  } else {
    std::vector<std::thread> threads;
    for (size_t d = 1; d < ndev; ++d) threads.emplace_back(work, d, ctxs[d], false);
    work(0, ctxs[0], false);
    for (auto& t : threads) t.join();
    for (size_t d = 1; d < ndev; ++d) zmx_internal_stats_add(shards[d].stats);
    // the slowest device's breakdown stands for the request (zmx_last_timing)
    for (size_t d = 1; d < ndev; ++d) {
      const zamd::Timing& a = shards[d].timing;
      zamd::Timing& t = zamd::ThreadTiming();
      if (a.tables + a.greedy + a.squeeze + a.cost_model + a.split + a.encode >
          t.tables + t.greedy + t.squeeze + t.cost_model + t.split + t.encode) t = a;
    }
  }
  // A shard that failed (its device ran out of memory, its context is broken) is done again on a context that just
  // finished its own shard without error — another device's where there is one — before the request gives up: the
  // parts are independent (deflate.c:916-923), whoever computes them computes the same bits.
  for (size_t d = 0; d < ndev; ++d) {
    if (!shards[d].rc) continue;
    // (not a failure that would repeat itself on any context — a request the device layer refuses, a table set that
    //  overflows its pools after the retries the device layer makes itself: ZMX_ERR_REFUSED.  By the error's CLASS, not its
    //  text: an out-of-memory inside PoolAlloc reads "PoolAlloc(...): out of memory" and is exactly what a retry is for)
    if (shards[d].err_class == ZMX_ERR_REFUSED) continue;
    zmx_ctx* other = nullptr;
    for (size_t e = 0; e < ndev && !other; ++e) if (e != d && !shards[e].rc && !shards[e].redone) other = ctxs[e];
    if (!other) break;
    std::fprintf(stderr, "zopfli_amd: a shard failed (%s): done again on another context\n", shards[d].err.c_str());
    work(d, other, true);
    shards[d].redone = true;
  }
  const double tr_joined = WallMs();
  for (auto& sh : shards) {
    if (sh.rc) {
      std::fprintf(stderr, "zopfli_amd: device error: %s\n", sh.err.c_str());
      return sh.rc;
    }
    for (auto& c : sh.chunks) chunks->push_back(std::move(c));
  }
  if (TraceCall()) {
    std::fprintf(stderr, "RunPartsSharded: lease %.2f ms, shards done +%.2f, chunks moved +%.2f\n", tr_lease - tr_begin,
                 tr_joined - tr_begin, WallMs() - tr_begin);
  }
  if (sum) {
    sum->value = sum->kind == ZMX_ADLER32 ? 1u : 0u;   // of no bytes
    for (auto& sh : shards) {
      if (sh.sum_bytes) sum->value = zmx_checksum_combine(sum->kind, sum->value, sh.sum, sh.sum_bytes);
    }
  }
  return 0;
}

// The device layer indexes the positions of a resident input with 32 bits; the reference takes a size_t.  A request
// of more master blocks than ZOPFLI_AMD_ROUND_PARTS (2000: 2 GB, so that a round's shard plus its window stays below
// 2^32 bytes whatever the dealing) is done in ROUNDS, one after the other, each dealt over the contexts like a call of
// its own; master blocks are independent (deflate.c:916-923), their bit chunks are joined in stream order as always,
// and the container checksum of the rounds is put together like that of the shards (zmx_checksum_combine).
int RunPartsSharded(const ZopfliOptions& options, int btype, const unsigned char* in,
                    const std::vector<zamd::Part>& parts, std::vector<zamd::Chunk>* chunks,
                    ChecksumRequest* sum = nullptr) {
  static const size_t round_parts = [] {
    const char* e = std::getenv("ZOPFLI_AMD_ROUND_PARTS");
    const long v = e ? std::atol(e) : 0;
    return v > 0 ? static_cast<size_t>(v) : static_cast<size_t>(2000);
  }();
  if (parts.size() <= round_parts) return RunPartsShardedOnce(options, btype, in, parts, chunks, sum);
  uint32_t acc = sum ? (sum->kind == ZMX_ADLER32 ? 1u : 0u) : 0u;   // of no bytes
  for (size_t a = 0; a < parts.size(); a += round_parts) {
    const size_t b = std::min(parts.size(), a + round_parts);
    const std::vector<zamd::Part> round(parts.begin() + static_cast<long>(a), parts.begin() + static_cast<long>(b));
    ChecksumRequest rs{sum ? sum->kind : 0, sum ? sum->limit : 0, 0};
    const int rc = RunPartsShardedOnce(options, btype, in, round, chunks, sum ? &rs : nullptr);
    if (rc) return rc;
    if (sum && round.front().instart < sum->limit) {
      const size_t covered = std::min(round.back().inend, sum->limit) - round.front().instart;
      acc = zmx_checksum_combine(sum->kind, acc, rs.value, covered);
    }
  }
  if (sum) sum->value = acc;
  return 0;
}

// Appends merged chunks at (*out, *outsize, *bp), reference conventions.
void EmitChunks(const std::vector<zamd::Chunk>& chunks, const unsigned char* in, unsigned char* bp,
                unsigned char** out, size_t* outsize, bool verbose = false) {
  zamd::MergeChunks(chunks, in, bp, out, outsize, verbose);
}

void ResetTiming() {
  zamd::ThreadTiming() = zamd::Timing();
  double a[8], b;
  zmx_internal_kernel_stats(a, &b, 1);
  zmx_internal_seg_stats(a, 1);
  zmx_internal_match_stats(a, 1);
  zmx_internal_match5_stats(a, 1);
}

void PushByte(unsigned v, unsigned char** out, size_t* outsize) {
  const uint8_t b = static_cast<uint8_t>(v);
  zamd::AppendToOutput(&b, 1, out, outsize);
}

}  // namespace

extern "C" {

size_t zmx_host_cache_trim(void) { return zamd::BlockCache::Trim(); }

void ZopfliInitOptions(ZopfliOptions* options) {
  options->verbose = 0;
  options->verbose_more = 0;
  options->numiterations = 15;
  options->blocksplitting = 1;
  options->blocksplittinglast = 0;
  options->blocksplittingmax = 15;
}

void ZopfliDeflatePart(const ZopfliOptions* options, int btype, int final, const unsigned char* in,
                       size_t instart, size_t inend, unsigned char* bp, unsigned char** out,
                       size_t* outsize) {
  ResetTiming();
  // only in[windowstart, inend) is read (lz77.c:551-552): that becomes the resident input
  std::vector<zamd::Part> parts{{instart, inend, final != 0}};
  std::vector<zamd::Chunk> chunks;
  if (RunPartsSharded(*options, btype, in, parts, &chunks) != 0) Die("device error");
  EmitChunks(chunks, in, bp, out, outsize, options->verbose != 0);
}

namespace {
// ZopfliDeflate (deflate.c:908-931); `sum`: see RunPartsSharded
void DeflateWhole(const ZopfliOptions* options, int btype, int final, const unsigned char* in, size_t insize,
                  unsigned char* bp, unsigned char** out, size_t* outsize, ChecksumRequest* sum) {
  const size_t offset = *outsize;
  {
    const double tr0 = WallMs();
    ResetTiming();
    const std::vector<zamd::Part> parts = MasterBlocks(insize, final != 0);
    std::vector<zamd::Chunk> chunks;
    const double tr1 = WallMs();
    if (RunPartsSharded(*options, btype, in, parts, &chunks, sum) != 0) Die("device error");
    const double tr2 = WallMs();
    EmitChunks(chunks, in, bp, out, outsize, options->verbose != 0);
    const double tr3 = WallMs();
    chunks.clear();
    chunks.shrink_to_fit();
    if (TraceCall()) {
      std::fprintf(stderr, "DeflateWhole: set-up %.2f ms, parts %.2f, merge %.2f, chunks freed %.2f\n", tr1 - tr0, tr2 - tr1,
                   tr3 - tr2, WallMs() - tr3);
    }
  }
  if (options->verbose) {
    std::fprintf(stderr, "Original Size: %lu, Deflate: %lu, Compression: %f%% Removed\n",
                 static_cast<unsigned long>(insize), static_cast<unsigned long>(*outsize - offset),
                 100.0 * static_cast<double>(insize - (*outsize - offset)) / static_cast<double>(insize));
  }
}
}  // namespace

void ZopfliDeflate(const ZopfliOptions* options, int btype, int final, const unsigned char* in,
                   size_t insize, unsigned char* bp, unsigned char** out, size_t* outsize) {
  DeflateWhole(options, btype, final, in, insize, bp, out, outsize, nullptr);
}

void ZopfliGzipCompress(const ZopfliOptions* options, const unsigned char* in, size_t insize,
                        unsigned char** out, size_t* outsize) {
  // the CRC is taken on the device(s) from the resident input (zmx_checksum)
  ChecksumRequest sum{ZMX_CRC32, insize, 0};
  unsigned char bp = 0;
  static const unsigned char header[10] = {31, 139, 8, 0, 0, 0, 0, 0, 2, 3};  // gzip_container.c:90-101
  zamd::AppendToOutput(header, 10, out, outsize);
  DeflateWhole(options, 2, 1, in, insize, &bp, out, outsize, &sum);
  const uint32_t crc = sum.value;
  for (int i = 0; i < 4; ++i) PushByte((crc >> (8 * i)) & 255, out, outsize);
  for (int i = 0; i < 4; ++i) PushByte((insize >> (8 * i)) & 255, out, outsize);
  if (options->verbose) {
    std::fprintf(stderr, "Original Size: %d, Gzip: %d, Compression: %f%% Removed\n", static_cast<int>(insize),
                 static_cast<int>(*outsize), 100.0 * static_cast<double>(insize - *outsize) / static_cast<double>(insize));
  }
}

void ZopfliZlibCompress(const ZopfliOptions* options, const unsigned char* in, size_t insize,
                        unsigned char** out, size_t* outsize) {
  // the reference truncates the size to unsigned here (zlib_container.c:54)
  ChecksumRequest sum{ZMX_ADLER32, static_cast<unsigned>(insize), 0};
  unsigned char bp = 0;
  const unsigned cmf = 120, flevel = 3, fdict = 0;  // CM 8, CINFO 7
  unsigned cmfflg = 256 * cmf + fdict * 32 + flevel * 64;
  cmfflg += 31 - cmfflg % 31;
  PushByte(cmfflg / 256, out, outsize);
  PushByte(cmfflg % 256, out, outsize);
  DeflateWhole(options, 2, 1, in, insize, &bp, out, outsize, &sum);
  const uint32_t checksum = sum.value;
  for (int i = 3; i >= 0; --i) PushByte((checksum >> (8 * i)) & 255, out, outsize);
  if (options->verbose) {
    std::fprintf(stderr, "Original Size: %d, Zlib: %d, Compression: %f%% Removed\n", static_cast<int>(insize),
                 static_cast<int>(*outsize), 100.0 * static_cast<double>(insize - *outsize) / static_cast<double>(insize));
  }
}

void ZopfliCompress(const ZopfliOptions* options, ZopfliFormat output_type, const unsigned char* in,
                    size_t insize, unsigned char** out, size_t* outsize) {
  if (output_type == ZOPFLI_FORMAT_GZIP) {
    ZopfliGzipCompress(options, in, insize, out, outsize);
  } else if (output_type == ZOPFLI_FORMAT_ZLIB) {
    ZopfliZlibCompress(options, in, insize, out, outsize);
  } else if (output_type == ZOPFLI_FORMAT_DEFLATE) {
    unsigned char bp = 0;
    ZopfliDeflate(options, 2, 1, in, insize, &bp, out, outsize);
  } else {
    std::fprintf(stderr, "zopfli_amd: invalid ZopfliFormat %d\n", static_cast<int>(output_type));
    std::abort();  // the reference asserts (zopfli_lib.c:40)
  }
}

int zmx_deflate_range(zmx_ctx* ctx, const ZopfliOptions* options, size_t instart, size_t inend, int final,
                      unsigned char** blob, size_t* blobsize) {
  MaybeKeepHeap();
  ResetTiming();
  if (inend < instart || inend > zmx_internal_input_size(ctx)) return -1;
  // deflate.c:916-923 on [instart, inend)
  std::vector<zamd::Part> parts;
  size_t i = instart;
  do {
    const bool masterfinal = i + kMasterBlock >= inend;
    const size_t size = masterfinal ? inend - i : kMasterBlock;
    parts.push_back({i, i + size, final != 0 && masterfinal});
    i += size;
  } while (i < inend);
  std::vector<zamd::Chunk> chunks;
  const auto tr0 = std::chrono::steady_clock::now();
  const int rc = RunParts(ctx, *options, 2, parts, &chunks);
  if (rc) return rc;
  const auto ts0 = std::chrono::steady_clock::now();
  if (std::getenv("ZOPFLI_AMD_PROF"))
    std::fprintf(stderr, "zmx_deflate_range: RunParts %.1f ms\n", std::chrono::duration<double>(ts0 - tr0).count() * 1e3);
  *blob = zamd::SerializeChunks(chunks, zmx_internal_input_host(ctx), blobsize);
  if (!*blob) return -1;
  zamd::ThreadTiming().serialize += std::chrono::duration<double>(std::chrono::steady_clock::now() - ts0).count();
  return 0;
}

int zmx_chunks_merge(const unsigned char* const* blobs, const size_t* blobsizes, size_t nblobs,
                     unsigned char* bp, unsigned char** out, size_t* outsize) {
  std::vector<zamd::Chunk> chunks;
  for (size_t i = 0; i < nblobs; ++i) {
    if (!zamd::DeserializeChunks(blobs[i], blobsizes[i], &chunks)) return -1;
  }
  EmitChunks(chunks, nullptr, bp, out, outsize);
  return 0;
}

int zmx_last_timing(double* out8) {
  const zamd::Timing& t = zamd::ThreadTiming();
  out8[0] = t.tables;
  out8[1] = t.greedy;
  out8[2] = t.squeeze;
  out8[3] = t.cost_model;
  out8[4] = t.split;
  out8[5] = t.encode;
  double k[3];
  zmx_internal_kernel_stats(k, &out8[7], 0);
  out8[6] = k[1];
  return 0;
}

int zmx_last_kernel_timing(double* out4) {
  zmx_internal_kernel_stats(out4, &out4[3], 0);
  return 0;
}

int zmx_last_match_timing(double* out4) {
  zmx_internal_match_stats(out4, 0);
  return 0;
}

int zmx_last_match_walk(double* out3) {
  zmx_internal_match5_stats(out3, 0);
  return 0;
}

int zmx_last_seg_stats(double* out8) {
  zmx_internal_seg_stats(out8, 0);
  return 0;
}

int zmx_last_host_timing(double* out2) {
  out2[0] = zamd::ThreadTiming().download;
  out2[1] = zamd::ThreadTiming().serialize;
  return 0;
}

// deflate.h:79,85 on the reference's own store type (lz77.h:44-62): the histogram of the range from litlens / dists,
// its byte length from pos (lz77.c:160-166), then the block-cost code of the library (host/block_cost.cc)
namespace {
zamd::Histogram RangeHistogram(const ZopfliLZ77Store* lz77, size_t lstart, size_t lend) {
  zamd::Histogram h;
  h.Clear();
  for (size_t i = lstart; i < lend; ++i) {
    if (lz77->dists[i] == 0) {
      h.ll[lz77->litlens[i]]++;
    } else {
      h.ll[zamd::LengthSymbol(lz77->litlens[i])]++;
      h.d[zamd::DistSymbol(lz77->dists[i])]++;
    }
  }
  return h;
}
double StoredSize(const ZopfliLZ77Store* lz77, size_t lstart, size_t lend) {
  size_t length = 0;
  if (lstart != lend) {
    const size_t l = lend - 1;
    length = lz77->pos[l] + (lz77->dists[l] == 0 ? 1 : lz77->litlens[l]) - lz77->pos[lstart];
  }
  const size_t rem = length % 65535;
  const size_t blocks = length / 65535 + (rem ? 1 : 0);
  return static_cast<double>(blocks * 5 * 8 + length * 8);      // deflate.c:591-597
}
}  // namespace

double ZopfliCalculateBlockSize(const ZopfliLZ77Store* lz77, size_t lstart, size_t lend, int btype) {
  if (btype == 0) return StoredSize(lz77, lstart, lend);
  return zamd::BlockSizeFromHistogram(RangeHistogram(lz77, lstart, lend), btype);
}

double ZopfliCalculateBlockSizeAutoType(const ZopfliLZ77Store* lz77, size_t lstart, size_t lend) {
  const double stored = StoredSize(lz77, lstart, lend);
  const zamd::Histogram h = RangeHistogram(lz77, lstart, lend);
  const double fixed = lz77->size > 1000 ? stored : zamd::BlockSizeFromHistogram(h, 1);   // deflate.c:614-616
  const double dynamic = zamd::BlockSizeFromHistogram(h, 2);
  return (stored < fixed && stored < dynamic) ? stored : (fixed < dynamic ? fixed : dynamic);
}

// zopflipng's per-row filter search on one of the entry points' contexts (include/zopfli_amd.h)
int zmx_png_filter_types_pooled(const unsigned char* image, size_t linebytes, size_t height, size_t bytewidth,
                                unsigned char* minsum_types, unsigned char* entropy_types) {
  Lease lease(1);
  return zmx_png_filter_types(lease.ctxs[0], image, linebytes, height, bytewidth, minsum_types, entropy_types);
}

}  // extern "C"

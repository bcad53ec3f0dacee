// Device layer of libzopfli_amd.so (C ABI part 2 of include/zopfli_amd.h):
// HBM residency, launch orchestration and the parity probes.  The kernels are
// in zmx_kernels.h.  gfx950 only; there is no host fallback — every entry
// point reports an error if the device or a launch fails.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "zmx_kernels.h"
#include "zmx_match2.h"
// The kernels that lost their measurement — k_bucket + k_match3 / k_match4 (sorted candidate slices, round 3) and the
// four-wave run task k_dp6_spec (round 5) — are NOT in the shipped library: they are compiled in with -DZMX_EXPERIMENTS
// only (tools/build_variant.py --experiments), for the comparison and stress scripts under tools/.
#ifdef ZMX_EXPERIMENTS
#include "zmx_match3.h"
#include "zmx_match4.h"
#else
#define BK_CH 32768u      // (the chunk size the chunk_base bookkeeping below shares with the experiments)
#endif
#include "zmx_match5.h"
#include "zmx_dp4.h"
#include "zmx_dp5.h"     // (includes zmx_dp6.h: the cooperative run-task job)
#include "zmx_encode.h"
#include "zmx_checksum.h"
#include "zmx_trace.h"
#include "zmx_greedy.h"
#include "zmx_png.h"
#include "zmx_blockcost.h"
#include "zopfli_amd.h"
#include "../host/thread_pool.h"

namespace {

thread_local std::string g_err;   // per calling thread (zmx_last_error)
// Kernel, match and task statistics: per calling THREAD (a Zopfli* call's shard threads hand theirs to the caller's
// at the join, api.cc), so that concurrent callers read their own numbers (zmx_last_*).
struct ThreadStats {
  double kernel_seconds[3];  // k_wtab + k_badscan, chain kernels (k_dp5_spec + k_dpcheck + k_dp4_fix), k_trace (HIP events)
  double squeeze_launches;
  double match5[3];          // k_match5: entries in flight summed over lanes and iterations, wave iterations, positions it walked
  double match[4];           // match kernel seconds, k_same + k_chain (+ k_levels ...) seconds, table builds, positions matched
  double seg[8];             // tasks, accepted, re-run: state / level / tie, positions re-run, re-run: values, positions
};
static_assert(sizeof(ThreadStats) == 19 * sizeof(double), "zmx_internal_stats_take / _add move 19 doubles");
thread_local ThreadStats g_ts = {};
#define g_kernel_seconds g_ts.kernel_seconds
#define g_squeeze_launches g_ts.squeeze_launches
#define g_match5_stats g_ts.match5
#define g_match_stats g_ts.match
#define g_seg_stats g_ts.seg

thread_local bool g_last_oom = false;   // the last failure of this thread was an allocation the device could not serve
// What KIND of failure the last one of this thread was (zmx_last_error_class): callers decide by this code, never by the
// message's text (the text holds the failing expression and __FILE__: "PoolAlloc(...)", a build path).
thread_local int g_err_class = ZMX_ERR_NONE;

int Fail(const char* what, hipError_t e, const char* file, int line) {
  char buf[512];
  std::snprintf(buf, sizeof(buf), "%s: %s (%s:%d)", what, hipGetErrorString(e), file, line);
  g_err = buf;
  g_last_oom = e == hipErrorOutOfMemory;
  g_err_class = g_last_oom ? ZMX_ERR_OUT_OF_MEMORY : ZMX_ERR_DEVICE;
  if (g_last_oom) (void)hipGetLastError();
  return -1;
}
// a request the device layer refuses whoever runs it: bad arguments, a size limit, a pool that overflows after its retries
int FailMsg(const std::string& m) {
  g_err = m;
  g_err_class = ZMX_ERR_REFUSED;
  return -1;
}
// the device did something it should not have (a kernel's guard fired, consistency flags): another context may fare better
int FailFault(const std::string& m) {
  g_err = m;
  g_err_class = ZMX_ERR_DEVICE;
  return -1;
}

#define HIPCHK(expr)                                                  \
  do {                                                                \
    hipError_t e_ = (expr);                                           \
    if (e_ != hipSuccess) return Fail(#expr, e_, __FILE__, __LINE__); \
  } while (0)

// Every zmx_* entry point runs on its context's device and leaves the calling thread's current HIP
// device as it found it.
struct DeviceGuard {
  int old = -1;
  hipError_t err;
  explicit DeviceGuard(int device) {
    if (hipGetDevice(&old) != hipSuccess) old = -1;
    err = hipSetDevice(device);
  }
  ~DeviceGuard() { if (old >= 0) (void)hipSetDevice(old); }
};

constexpr int kTooLarge = -2;   // zmx_tables_build*: the batch does not fit the code budget, try fewer blocks
constexpr u32 kMatchGrid = 1024;  // persistent workgroups: 256 CUs x 4 (LDS-limited)
constexpr u32 kMatchGrid3 = 768;  // k_match3: 256 CUs x 3
constexpr int kMatchDefault = 0;   // ZOPFLI_AMD_MATCH when unset: per block, k_match5 where k_hits says the chains are long, else k_match2
constexpr u32 kMatchGrid5 = 1536; // k_match5: 256 CUs x 6 workgroups of 4 waves (its scratch is k_match2's: 1536 x 256 <= 1024 x 512 lanes)
constexpr size_t kInputPad = 4096;

// ---------------------------------------------------------------------------------------------
// ZOPFLI_AMD_GUARD=1 — a debugging mode for the device allocations (round-2 verdict: an unexplained
// "Memory access fault by GPU" must be localisable).  Every pooled or direct allocation gets a red zone of
// kGuardBytes before and after it, filled with 0xA5; its body is filled with 0xCD whenever it is handed out
// (fresh or recycled: stale contents of an earlier batch cannot stand in for data a kernel forgot to write);
// after EVERY kernel launch the stream is drained and k_guard_check reads all red zones of the context: the
// first byte that changed is reported with the kernel that just ran, the allocation's tag and size, and the
// offset.  Slow (a synchronisation per launch); off by default.
// ---------------------------------------------------------------------------------------------
constexpr size_t kGuardBytes = 4096;
constexpr u32 kGuardMaxAllocs = 256;
bool GuardOn() {
  static const bool on = [] { const char* e = std::getenv("ZOPFLI_AMD_GUARD"); return e && std::atoi(e) != 0; }();
  return on;
}

}  // namespace

// zones[2 i], zones[2 i + 1] = device addresses of the two red zones of allocation i; res = {flag, alloc, offset, value}
__global__ __launch_bounds__(256) void k_guard_check(const u64* zones, u32 nzones, u32* res) {
  const u32 z = blockIdx.x;
  if (z >= nzones) return;
  const u32* q = reinterpret_cast<const u32*>(zones[z]);
  for (u32 i = threadIdx.x; i < kGuardBytes / 4; i += 256) {
    const u32 v = q[i];
    if (v != 0xa5a5a5a5u && atomicCAS(&res[0], 0u, 1u) == 0u) { res[1] = z; res[2] = i * 4; res[3] = v; }
  }
}

struct zmx_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  u8* d_in = nullptr;
  size_t insize = 0, in_cap = 0;
  const unsigned char* h_in = nullptr;  // caller's buffer (borrowed until the next zmx_set_input)
  u32* d_scratch = nullptr;  // k_match2 per-lane overflow change points
  u32* d_scratch5 = nullptr; // k_match5's (it may run beside k_match2)
  // table arrays are recycled between batches and calls: hipMalloc/hipFree of multi-GB arrays
  // cost more than the kernels that fill them
  std::unordered_map<void*, size_t> pool_live;
  std::vector<std::pair<void*, size_t>> pool_free;
  size_t pool_free_bytes = 0;
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
  hipStream_t stream2 = nullptr;   // the run tasks' k_dp5_spec beside the others' (zmx_squeeze_run)
  bool stream2_outstanding = false;   // a kernel on stream2 reads pooled scratch arrays and `stream` has not been made to wait for it yet
  hipStream_t alt_stream[3][2] = {{nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}};   // zmx_ctx_set_priority: [0] the created pair, [1] high, [2] low
  hipEvent_t ev2[2] = {nullptr, nullptr};
  u32* h_stage = nullptr;    // pinned staging for store downloads (grow-only)
  // pinned buffers of table sets that were freed (h_runin / h_runout: a few KB per block), kept for the next set:
  // hipHostMalloc + hipHostFree were ~ 0.4 ms of every table set, a tenth of a small call's fixed cost
  std::vector<std::pair<unsigned char*, size_t>> pinned_free;
  size_t stage_cap = 0;      // in u32
  // ZOPFLI_AMD_GUARD (below): the bytes the caller asked for and who asked, per live allocation (keyed like pool_live)
  struct GuardInfo { size_t bytes; const char* tag; };
  std::unordered_map<void*, GuardInfo> guard_live;
  size_t pool_keep = 0;      // what the pool may keep cached between batches, and what one batch's DP edges may take:
  size_t code_budget = 0;    // a third of the device's memory each (hipMemGetInfo at creation), at most 96 GiB,
  size_t keep_base = 0;
  unsigned shares = 1;       // contexts on this device (zmx_ctx_set_share)
  u64* d_guard_tab = nullptr;   // [kGuardMaxAllocs][2] zone pairs for k_guard_check, then 4 result words
  u64 guard_checks = 0;
};

struct zmx_tables {
  size_t nb = 0;
  std::vector<BlockDesc> blocks;
  std::vector<u32> bsize;
  std::vector<u32> store_begin[2];  // first valid entry of each block's store slot
  size_t total_b = 0, total_l = 0;
  BlockDesc* d_blocks = nullptr;
  u32* d_tile_off = nullptr;
  u16* d_same16 = nullptr;
  ushort4* d_links = nullptr;
  bool links_partial = false;     // built from a parent: the hash arrays exist only where the match kernel read them
  // k_bucket's order of every 32768-position chunk by (hash value, position), per hash (zmx_match3.h)
  u16* d_sorted_alloc = nullptr;
  u16* d_sorted[2] = {nullptr, nullptr};   // (inside d_sorted_alloc)
  u16* d_rank[2] = {nullptr, nullptr};
  u32* d_bucket[2] = {nullptr, nullptr};
  u8* d_ssame = nullptr;
  u32* d_chunk_base = nullptr;
  std::vector<u32> chunk_base;    // [nb + 1] first chunk of each block
  size_t merged_tasks = 0;        // tasks merged into their predecessors (BuildTables): the set has long tasks
  bool matches_only = false;      // built by zmx_tables_build_matches: no DP rows, codes, windows or tasks
  bool trimmed = false;           // zmx_tables_trim: only the stores are left
  bool buckets = false;           // the hash arrays are k_bucket's (k_match3), not k_chain's links (k_match2)
  u32* d_recs = nullptr;
  u32* d_pool = nullptr;
  u32 pool_cap = 0;
  u16* d_la = nullptr;
  u32* d_store[2] = {nullptr, nullptr};
  u32* d_hist = nullptr;
  u32* d_nsym = nullptr;
  double* d_cost = nullptr;
  double* d_mincost = nullptr;
  int* d_slot = nullptr;
  uint2* d_dph = nullptr;         // per position: DP row offset, kend | shortcut flag (k_rowscan)
  u64* d_block_edges = nullptr;   // per block: DP edges
  u32* d_badpos = nullptr;        // bit per position: it owns a match edge below mincost (k_badscan, per run)
  size_t badpos_words = 0;
  bool badpos_clean = false;      // the bitmap is all zero (no run since the last memset has marked a position)
  u64* d_code_base = nullptr;     // per block: first slot in d_codes
  u16* d_codes = nullptr;         // the DP edges as weight codes (k_codes)
  double* d_wtab = nullptr;       // [nb][ZMX_WTAB] the weights of the current run (k_wtab)
  u32* d_badcodes = nullptr;      // [nb][40] the weights below mincost (k_wtab)
  u32* d_seg_off = nullptr;       // per block: first trace segment (cumulative)
  u32* d_extab = nullptr;         // per trace segment: exit table (k_trace_exits)
  uint2* d_seginfo = nullptr;     // per trace segment: entry, symbol offset (k_trace_link)
  std::vector<u32> seg_off;
  std::vector<u64> block_edges;
  std::vector<u32> tile_off;
  u32* d_counters = nullptr;  // 48 words, see MatchParams (16 .. 21: k_match3's profile counts; 24 .. 31: k_match5's tile cursors, 32 .. 39: its watchdog's dump)
  u32* d_flags = nullptr;     // 4 words
  // what a squeeze run takes and gives, each side ONE array on the device and one pinned mirror on the host, so
  // that a run has one copy down and one up (eight small copies a run were 3 ms of copy kernels per 15 runs):
  // d_runin = cost | mincost | runinfo | slot, d_runout = hist | nsym | segstats | flags (the d_* above point into them)
  unsigned char* d_runin = nullptr;
  unsigned char* d_runout = nullptr;
  unsigned char* h_runin = nullptr;
  unsigned char* h_runout = nullptr;
  size_t h_runin_cap = 0, h_runout_cap = 0;
  size_t runin_bytes = 0, runout_bytes = 0;
  u64* d_prof = nullptr;      // nb * ZMX_PROF_N counters when ZOPFLI_AMD_PROF is set
  // the chain's tasks (zmx_dp4.h)
  std::vector<SegTask> tasks;
  std::vector<u32> task_off;  // [nb + 1]
  SegTask* d_tasks = nullptr;
  u32* d_task_off = nullptr;
  u32* d_wg_tasks = nullptr;       // k_dp5_spec's workgroups: four tasks of one block each
  u32 n_wg = 0;
  u32 n_wg_runs = 0;               // ... of which the last n_wg_runs hold run tasks (k_taskkind): k_dp5_spec<.., true>
  u32* d_task_kind = nullptr;      // [tasks] k_taskkind: 1 = a run task
  u32* d_run_list = nullptr;       // the run tasks, longest first: k_dp6_spec's workgroups (zmx_dp6.h)
  u32 n_run_list = 0;
  u32* d_wmeta = nullptr;          // per 32-position window: 40 words, what k_dp5_spec needs to fetch its rows (k_mkdesc)
  u32* d_winroff = nullptr;        // per 32-position window: offset of its first row in the block's codes (k_mkdesc)
  u32* d_winflag = nullptr;        // per 32-position window: fast path possible (k_mkdesc)
  u32* d_win_off = nullptr;        // [nb]
  std::vector<u32> win_off;
  float* d_lvl = nullptr;
  SegSnap* d_entry = nullptr;
  SegSnap* d_exit = nullptr;
  SegSnap* d_mid = nullptr;      // per task: the state where it first came near the end of its binade (zmx_dp5.h)
  SegCheck* d_chk = nullptr;
  u16* d_over = nullptr;      // [tasks][SEG_OVER]
  u32* d_redo = nullptr;      // [1 + 3 pad + tasks * 4]: k_dpscan's list of tasks to run a second time
  float* d_runinfo = nullptr; // [3][nb]: wmax, tie mask (as bits), estimated block cost
  u32* d_segstats = nullptr;  // 8 words
  std::vector<u32> h_hist;    // the histograms of the last greedy parse / squeeze run (host copy)
  bool have_hist = false;
  u32 squeeze_runs = 0;
  // host cache for the parity probe
  std::vector<std::vector<u32>> probe_recs;
  std::vector<u32> probe_pool;
  bool probe_pool_ready = false;
};

namespace {

constexpr size_t kPoolKeepMax = 96ull << 30;

// The red zones and the poison of an allocation that is being handed out (guard mode): base = what hipMalloc
// returned, the caller gets base + kGuardBytes.
hipError_t GuardDress(zmx_ctx* c, void* base, size_t bytes, size_t cap, const char* tag, void** user) {
  unsigned char* b = static_cast<unsigned char*>(base);
  const size_t body = (bytes + 15) & ~static_cast<size_t>(15);
  hipError_t e = hipMemsetAsync(b, 0xa5, kGuardBytes, c->stream);
  if (e == hipSuccess) e = hipMemsetAsync(b + kGuardBytes, 0xcd, cap - 2 * kGuardBytes, c->stream);
  if (e == hipSuccess) e = hipMemsetAsync(b + kGuardBytes + body, 0xa5, kGuardBytes, c->stream);
  *user = b + kGuardBytes;
  c->guard_live[*user] = zmx_ctx::GuardInfo{body, tag};
  return e;
}

std::atomic<zmx_oom_hook_t> g_oom_hook{nullptr};
// What the pools of ALL contexts of a device keep cached between batches, against ONE budget per device (a third of its
// memory): a lone busy context may cache all of it (100 MB of long runs are 52 GB of DP codes per batch; with a
// per-context third of a third they were hipFree'd and hipMalloc'ed every step: class Z 132 -> 49 MB/s), three busy
// ones share it, and idle ones are trimmed when another runs out (zmx_set_oom_hook).
constexpr int kMaxDevices = 64;
std::atomic<size_t> g_dev_cached[kMaxDevices];
inline std::atomic<size_t>& DevCached(const zmx_ctx* c);

inline std::atomic<size_t>& DevCached(const zmx_ctx* c) { return g_dev_cached[c->device >= 0 && c->device < kMaxDevices ? c->device : 0]; }

hipError_t PoolAllocBytes(zmx_ctx* c, void** p, size_t bytes, const char* tag) {
  if (bytes == 0) bytes = 1;
  const bool guard = GuardOn();
  const size_t want = guard ? ((bytes + 15) & ~static_cast<size_t>(15)) + 2 * kGuardBytes : bytes;
  size_t best = c->pool_free.size();
  for (size_t i = 0; i < c->pool_free.size(); ++i) {
    const size_t cap = c->pool_free[i].second;
    if (cap >= want && cap <= 2 * want + (1u << 20) && (best == c->pool_free.size() || cap < c->pool_free[best].second)) {
      best = i;
    }
  }
  void* base = nullptr;
  size_t cap = want;
  if (best != c->pool_free.size()) {
    base = c->pool_free[best].first;
    cap = c->pool_free[best].second;
    c->pool_free_bytes -= cap;
    DevCached(c).fetch_sub(cap, std::memory_order_relaxed);
    c->pool_free.erase(c->pool_free.begin() + static_cast<long>(best));
  } else {
    hipError_t e = hipMalloc(&base, want);
    if (e != hipSuccess && !c->pool_free.empty()) {  // out of memory: drop the cache and retry
      (void)hipGetLastError();
      for (auto& f : c->pool_free) (void)hipFree(f.first);
      c->pool_free.clear();
      DevCached(c).fetch_sub(c->pool_free_bytes, std::memory_order_relaxed);
      c->pool_free_bytes = 0;
      e = hipMalloc(&base, want);
    }
    if (e != hipSuccess) {
      // still out of memory: the idle contexts of the same device may sit on gigabytes of cached arrays (the owner
      // of the contexts — api.cc's ContextPool — trims them through this hook)
      if (const zmx_oom_hook_t hook = g_oom_hook.load(std::memory_order_acquire)) {
        (void)hipGetLastError();
        hook(c->device);
        e = hipMalloc(&base, want);
      }
    }
    if (e != hipSuccess) return e;
  }
  *p = base;
  if (guard) {
    const hipError_t e = GuardDress(c, base, bytes, cap, tag, p);
    if (e != hipSuccess) return e;
  }
  c->pool_live[*p] = cap;
  return hipSuccess;
}

template <typename T>
hipError_t PoolAllocT(zmx_ctx* c, T** p, size_t n, const char* tag) {
  return PoolAllocBytes(c, reinterpret_cast<void**>(p), (n ? n : 1) * sizeof(T), tag);
}
#define PoolAlloc(c, p, n) PoolAllocT(c, p, n, #p)

// Temporary arrays of one call: back to the pool when the call returns, whichever way (HIPCHK returns early).
// k_match2's four-byte candidate filter (zmx_match2.h, FILT); ZOPFLI_AMD_MATCH_FILTER=0 keeps the one-byte test
bool MatchFilter() {
  static const bool on = [] { const char* e = std::getenv("ZOPFLI_AMD_MATCH_FILTER"); return e ? std::atoi(e) != 0 : true; }();
  return on;
}

// Which match-table kernel (ZOPFLI_AMD_MATCH / zmx_set_match_kernel):
//   0 (default) per block: k_hits estimates the hits per position of the reference's walk; blocks above
//       ZOPFLI_AMD_MATCH_HITS (300) take the exact skip-walk k_match5 (level links + counted hits, zmx_match5.h: 5 - 9 x
//       faster on PNG-like and two-symbol data, profiles/r04_match.txt), the others k_match2, side by side on two streams
//   2 = k_chain + k_match2 everywhere (prev links, a lane per position)
//   3 = k_bucket + k_match3 (sorted candidate slices, a wave per position, 64 candidates per coalesced load)
//   4 = k_bucket + k_match4 (the same slices streamed by a lane per position, four candidates per step)
//   5 = k_match5 everywhere
// All produce the same records (test_match_kernels_agree).  Tables built from a parent recompute their few tiles with
// k_match2 either way.
std::atomic<int> g_match_kernel{-1};
int MatchKernel() {
  int v = g_match_kernel.load(std::memory_order_relaxed);
  if (v < 0) {
    const char* e = std::getenv("ZOPFLI_AMD_MATCH");
    const int k = e ? std::atoi(e) : kMatchDefault;
#ifdef ZMX_EXPERIMENTS
    v = k == 0 || k == 3 || k == 4 || k == 5 ? k : 2;
#else
    v = k == 0 || k == 5 ? k : 2;      // (3 and 4 exist in -DZMX_EXPERIMENTS builds only)
#endif
    g_match_kernel.store(v, std::memory_order_relaxed);
  }
  return v;
}
// kernel 0: blocks whose estimated hits per position (k_hits) exceed this take k_match5 (ZOPFLI_AMD_MATCH_HITS)
u64 MatchAutoHits() {
  static const u64 v = [] { const char* e = std::getenv("ZOPFLI_AMD_MATCH_HITS"); return e ? static_cast<u64>(std::max<long>(0, std::atol(e))) : 300ull; }();
  return v;
}

struct PoolScope {
  zmx_ctx* c;
  std::vector<void*> held;
  explicit PoolScope(zmx_ctx* ctx) : c(ctx) {}
  ~PoolScope();
  template <typename T>
  hipError_t AllocT(T** p, size_t n, const char* tag) {
    const hipError_t e = PoolAllocT(c, p, n, tag);
    if (e == hipSuccess) held.push_back(*p);
    return e;
  }
};

void PoolFree(zmx_ctx* c, void* p) {
  if (!p) return;
  void* base = p;
  if (c) {
    auto it = c->pool_live.find(p);
    if (it != c->pool_live.end()) {
      const size_t cap = it->second;
      c->pool_live.erase(it);
      if (c->guard_live.erase(p)) base = static_cast<unsigned char*>(p) - kGuardBytes;
      if (DevCached(c).load(std::memory_order_relaxed) + cap > c->pool_keep &&
          DevCached(c).load(std::memory_order_relaxed) > c->pool_free_bytes) {
        // the device's cache budget is used up and some of it is ANOTHER context's: the idle contexts' caches go
        // first — the context that is working is the one whose arrays will be asked for again.  (Not when the cache is
        // all this context's own: the hook would take the pool's lock and find nothing, on every free of the hot path.)
        if (const zmx_oom_hook_t hook = g_oom_hook.load(std::memory_order_acquire)) hook(c->device);
      }
      if (DevCached(c).load(std::memory_order_relaxed) + cap <= c->pool_keep) {
        c->pool_free.emplace_back(base, cap);
        c->pool_free_bytes += cap;
        DevCached(c).fetch_add(cap, std::memory_order_relaxed);
        return;
      }
    }
  }
  (void)hipFree(base);
}

// Guard mode: drain the stream and check every red zone of the context.  `where` = the kernel that just ran.
int GuardVerify(zmx_ctx* c, const char* where) {
  if (hipStreamSynchronize(c->stream) != hipSuccess) return FailFault(std::string("ZOPFLI_AMD_GUARD: the stream failed after ") + where);
  if (c->guard_live.empty()) return 0;
  // (ZOPFLI_AMD_GUARD_SELFTEST=N: the N-th check finds a byte that this function itself just broke — the test that the
  //  mode reports what it is there to report)
  static const u64 selftest = [] { const char* e = std::getenv("ZOPFLI_AMD_GUARD_SELFTEST"); return e ? static_cast<u64>(std::atoll(e)) : 0ull; }();
  std::vector<std::pair<void*, zmx_ctx::GuardInfo>> live(c->guard_live.begin(), c->guard_live.end());
  if (!c->d_guard_tab) HIPCHK(hipMalloc(reinterpret_cast<void**>(&c->d_guard_tab), (2 * static_cast<size_t>(kGuardMaxAllocs) + 2) * sizeof(u64)));
  ++c->guard_checks;
  // every live allocation, kGuardMaxAllocs at a time (a context that holds parent, optimal and fixed-tree tables plus
  // temporaries has more than one table's worth)
  for (size_t first = 0; first < live.size(); first += kGuardMaxAllocs) {
    const u32 n = static_cast<u32>(std::min<size_t>(live.size() - first, kGuardMaxAllocs));
    std::vector<u64> tab(2 * static_cast<size_t>(kGuardMaxAllocs) + 2, 0);
    for (u32 i = 0; i < n; ++i) {
      const unsigned char* user = static_cast<const unsigned char*>(live[first + i].first);
      tab[2 * i] = reinterpret_cast<u64>(user - kGuardBytes);
      tab[2 * i + 1] = reinterpret_cast<u64>(user + live[first + i].second.bytes);
    }
    if (first == 0 && selftest && c->guard_checks == selftest) HIPCHK(hipMemset(reinterpret_cast<void*>(tab[1] + 100), 0x5a, 1));
    HIPCHK(hipMemcpy(c->d_guard_tab, tab.data(), tab.size() * sizeof(u64), hipMemcpyHostToDevice));   // (the result words zeroed with it)
    u32* res = reinterpret_cast<u32*>(c->d_guard_tab + 2 * static_cast<size_t>(kGuardMaxAllocs));
    hipLaunchKernelGGL(k_guard_check, dim3(2 * n), dim3(256), 0, c->stream, c->d_guard_tab, 2 * n, res);
    HIPCHK(hipGetLastError());
    u32 h[4] = {0, 0, 0, 0};
    HIPCHK(hipMemcpy(h, res, sizeof(h), hipMemcpyDeviceToHost));
    if (h[0] == 0) continue;
    const auto& g = live[first + (h[1] >> 1)].second;
    char buf[400];
    std::snprintf(buf, sizeof(buf), "ZOPFLI_AMD_GUARD: after %s the red zone %s allocation '%s' (%zu bytes) changed: byte offset %u of the zone holds 0x%08x",
                  where, (h[1] & 1u) ? "behind" : "in front of", g.tag ? g.tag : "?", g.bytes, h[2], h[3]);
    std::fprintf(stderr, "%s\n", buf);
    return FailFault(buf);
  }
  return 0;
}
// after every kernel launch: the launch error, and in guard mode the red zones
#define KCHK(c, name)                                            \
  do {                                                           \
    HIPCHK(hipGetLastError());                                   \
    if (GuardOn()) { const int rc_ = GuardVerify(c, name); if (rc_) return rc_; } \
  } while (0)

PoolScope::~PoolScope() {
  // (an error return between a launch on stream2 and the join: the kernel may still be reading these arrays)
  if (c->stream2_outstanding) {
    (void)hipStreamSynchronize(c->stream2);
    c->stream2_outstanding = false;
  }
  for (void* p : held) PoolFree(c, p);
}

}  // namespace

extern "C" {

int zmx_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

const char* zmx_last_error(void) { return g_err.c_str(); }
int zmx_last_error_class(void) { return g_err_class; }
// The phase times of a squeeze run (k_wtab / the chain / the trace: zmx_last_kernel_timing, zmx_last_timing's dp_kernel)
// are four event records and three readings a run — seven of a run's ~22 runtime calls, which is what sixteen concurrent
// callers of small files queue for.  Off unless somebody asks: ZOPFLI_AMD_KERNEL_TIMING=1, ZOPFLI_AMD_PROF, or this call
// (bench.py, tools/latency.py and the tests' harness do).
static std::atomic<int> g_kernel_timing{-1};
static bool KernelTiming() {
  int v = g_kernel_timing.load(std::memory_order_relaxed);
  if (v < 0) {
    const char* e = std::getenv("ZOPFLI_AMD_KERNEL_TIMING");
    v = (e ? std::atoi(e) != 0 : std::getenv("ZOPFLI_AMD_PROF") != nullptr) ? 1 : 0;
    g_kernel_timing.store(v, std::memory_order_relaxed);
  }
  return v != 0;
}
void zmx_set_kernel_timing(int on) { g_kernel_timing.store(on ? 1 : 0, std::memory_order_relaxed); }

int zmx_has_experiments(void) {
#ifdef ZMX_EXPERIMENTS
  return 1;
#else
  return 0;
#endif
}

void zmx_set_oom_hook(zmx_oom_hook_t hook) { g_oom_hook.store(hook, std::memory_order_release); }

int zmx_ctx_set_share(zmx_ctx* c, unsigned contexts_on_device) {
  if (!c) return FailMsg("zmx_ctx_set_share: no context");
  // (The budgets are per DEVICE now, whoever uses them: the cache of all its contexts together against one third of its
  //  memory — g_dev_cached —, one batch's DP edges against one third.  The number of contexts that share the device is
  //  kept for the record; a build that cannot allocate because the others are busy comes back as "too large" and the
  //  caller halves the batch.)
  c->shares = contexts_on_device ? contexts_on_device : 1;
  return 0;
}

int zmx_ctx_trim_cache(zmx_ctx* c) {
  if (!c) return 0;
  DeviceGuard dev_guard(c->device);
  HIPCHK(dev_guard.err);
  for (auto& f : c->pool_free) (void)hipFree(f.first);
  c->pool_free.clear();
  DevCached(c).fetch_sub(c->pool_free_bytes, std::memory_order_relaxed);
  c->pool_free_bytes = 0;
  return 0;
}

int zmx_set_match_kernel(int kernel) {
#ifdef ZMX_EXPERIMENTS
  if (kernel != 0 && kernel != 2 && kernel != 3 && kernel != 4 && kernel != 5) return FailMsg("zmx_set_match_kernel: 0, 2, 3, 4 or 5");
#else
  if (kernel == 3 || kernel == 4) return FailMsg("zmx_set_match_kernel: kernels 3 and 4 are in -DZMX_EXPERIMENTS builds only");
  if (kernel != 0 && kernel != 2 && kernel != 5) return FailMsg("zmx_set_match_kernel: 0, 2 or 5");
#endif
  g_match_kernel.store(kernel, std::memory_order_relaxed);
  return 0;
}

size_t zmx_internal_input_size(zmx_ctx* ctx) { return ctx->insize; }
int zmx_internal_device(zmx_ctx* ctx) { return ctx->device; }
void zmx_internal_set_error(const char* msg) { g_err = msg; g_err_class = ZMX_ERR_DEVICE; }
const unsigned char* zmx_internal_input_host(zmx_ctx* ctx) { return ctx->h_in; }

void zmx_internal_seg_stats(double* out8, int reset) {
  // (thread-local: no lock)
  for (int i = 0; i < 8; ++i) out8[i] = g_seg_stats[i];
  if (reset) for (int i = 0; i < 8; ++i) g_seg_stats[i] = 0;
}

void zmx_internal_match5_stats(double* out3, int reset) {
  // (thread-local: no lock)
  for (int i = 0; i < 3; ++i) out3[i] = g_match5_stats[i];
  if (reset) for (int i = 0; i < 3; ++i) g_match5_stats[i] = 0;
}

void zmx_internal_match_stats(double* out4, int reset) {
  // (thread-local: no lock)
  for (int i = 0; i < 4; ++i) out4[i] = g_match_stats[i];
  if (reset) for (int i = 0; i < 4; ++i) g_match_stats[i] = 0;
}

void zmx_internal_kernel_stats(double* seconds3, double* squeeze_launches, int reset) {
  // (thread-local: no lock)
  for (int i = 0; i < 3; ++i) seconds3[i] = g_kernel_seconds[i];
  *squeeze_launches = g_squeeze_launches;
  if (reset) {
    g_kernel_seconds[0] = g_kernel_seconds[1] = g_kernel_seconds[2] = 0;
    g_squeeze_launches = 0;
  }
}

// a shard thread's sums since it started, zeroed — and added to the calling thread's (api.cc, at the join of a call's shards)
void zmx_internal_stats_take(double* out19) {
  std::memcpy(out19, &g_ts, sizeof(g_ts));
  g_ts = ThreadStats{};
}
void zmx_internal_stats_add(const double* in19) {
  double* d = reinterpret_cast<double*>(&g_ts);
  for (int i = 0; i < 19; ++i) d[i] += in19[i];
}

// Pinned host buffers of a context, recycled between table sets (see zmx_ctx::pinned_free).
static hipError_t PinnedTake(zmx_ctx* c, unsigned char** p, size_t bytes, size_t* cap) {
  size_t best = c->pinned_free.size();
  for (size_t i = 0; i < c->pinned_free.size(); ++i) {
    const size_t k = c->pinned_free[i].second;
    if (k >= bytes && k <= 4 * bytes + 4096 && (best == c->pinned_free.size() || k < c->pinned_free[best].second)) best = i;
  }
  if (best != c->pinned_free.size()) {
    *p = c->pinned_free[best].first;
    *cap = c->pinned_free[best].second;
    c->pinned_free.erase(c->pinned_free.begin() + static_cast<long>(best));
    return hipSuccess;
  }
  *cap = bytes < 4096 ? 4096 : bytes;
  return hipHostMalloc(reinterpret_cast<void**>(p), *cap, hipHostMallocDefault);
}
static void PinnedGive(zmx_ctx* c, unsigned char** p, size_t cap) {
  if (!*p) return;
  if (c && c->pinned_free.size() < 8 && cap <= (64u << 20)) c->pinned_free.emplace_back(*p, cap);
  else (void)hipHostFree(*p);
  *p = nullptr;
}

int zmx_ctx_create(int device, zmx_ctx** out) {
  int n = 0;
  HIPCHK(hipGetDeviceCount(&n));
  if (device < 0 || device >= n) return FailMsg("zmx_ctx_create: no such HIP device");
  DeviceGuard dev_guard(device);
  HIPCHK(dev_guard.err);
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, device));
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    return FailMsg(std::string("zmx_ctx_create: kernels are built for gfx950 only, device is ") + prop.gcnArchName);
  }
  zmx_ctx* c = new zmx_ctx();
  c->device = device;
  HIPCHK(hipStreamCreate(&c->stream));
  for (int i = 0; i < 4; ++i) HIPCHK(hipEventCreate(&c->ev[i]));
  HIPCHK(hipStreamCreate(&c->stream2));
  for (int i = 0; i < 2; ++i) HIPCHK(hipEventCreateWithFlags(&c->ev2[i], hipEventDisableTiming));
  {
    // Budgets from what the device has, not from what an MI355X has on paper: several contexts may share one
    // device (ZOPFLI_AMD_DEVICES=0,0), other processes may hold memory already.
    size_t mem_free = 0, mem_total = 0;
    HIPCHK(hipMemGetInfo(&mem_free, &mem_total));
    c->keep_base = std::min<size_t>(kPoolKeepMax, mem_free / 3);
    c->pool_keep = c->keep_base;
    c->code_budget = c->keep_base;
  }

  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_chain), hipFuncAttributeMaxDynamicSharedMemorySize,
                             CH_LDS_BYTES));
#ifdef ZMX_EXPERIMENTS
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_bucket), hipFuncAttributeMaxDynamicSharedMemorySize,
                             BK_LDS_BYTES));
#endif
  *out = c;
  return 0;
}

// Which streams the context's next calls run on: level 0 = the pair it was created with, +1 / -1 = a pair of the highest /
// lowest priority the device has (made when first asked for).  A call with block splitting that is dealt over the three
// contexts of a device gives them three priorities (api.cc): the contexts then run one AFTER the other where they would
// share the device evenly, so their host phases — the split searches, the joins — come at different times and fall beside
// the others' kernels instead of beside each other (100 MB of text: 152 -> 145 ms); without block splitting there is
// little host work to hide and running in turn only costs the overlap of the kernels' tails (123 -> 130 ms), so such
// calls stay on level 0.  Only between calls: the context must be idle.
int zmx_ctx_set_priority(zmx_ctx* c, int level) {
  DeviceGuard dev_guard(c->device);
  HIPCHK(dev_guard.err);
  const int k = level > 0 ? 1 : level < 0 ? 2 : 0;
  // (first: from here on zmx_ctx_destroy frees by alt_stream[][], a pair created half-way below included)
  if (!c->alt_stream[0][0]) { c->alt_stream[0][0] = c->stream; c->alt_stream[0][1] = c->stream2; }   // the pair of zmx_ctx_create
  if (k && !c->alt_stream[k][1]) {
    int least = 0, greatest = 0;
    HIPCHK(hipDeviceGetStreamPriorityRange(&least, &greatest));
    for (int i = 0; i < 2; ++i) {
      if (c->alt_stream[k][i]) continue;      // (an earlier call got this far)
      HIPCHK(hipStreamCreateWithPriority(&c->alt_stream[k][i], hipStreamDefault, k == 1 ? greatest : least));
    }
  }
  HIPCHK(hipStreamSynchronize(c->stream));
  HIPCHK(hipStreamSynchronize(c->stream2));
  c->stream = c->alt_stream[k][0];
  c->stream2 = c->alt_stream[k][1];
  return 0;
}

void zmx_ctx_destroy(zmx_ctx* c) {
  if (!c) return;
  DeviceGuard dev_guard(c->device);
  if (c->h_stage) (void)hipHostFree(c->h_stage);
  for (auto& f : c->pinned_free) (void)hipHostFree(f.first);
  for (auto& f : c->pool_free) (void)hipFree(f.first);
  DevCached(c).fetch_sub(c->pool_free_bytes, std::memory_order_relaxed);
  // (the input and k_match2's scratch are pooled allocations too; in guard mode the caller's pointer lies behind a red zone)
  for (auto& f : c->pool_live) (void)hipFree(c->guard_live.count(f.first) ? static_cast<unsigned char*>(f.first) - kGuardBytes : f.first);
  (void)hipFree(c->d_guard_tab);
  for (int i = 0; i < 4; ++i) if (c->ev[i]) (void)hipEventDestroy(c->ev[i]);
  for (int i = 0; i < 2; ++i) if (c->ev2[i]) (void)hipEventDestroy(c->ev2[i]);
  if (c->alt_stream[0][0]) {       // (stream / stream2 are one of these pairs)
    for (int k = 0; k < 3; ++k) for (int i = 0; i < 2; ++i) if (c->alt_stream[k][i]) (void)hipStreamDestroy(c->alt_stream[k][i]);
  } else {
    if (c->stream2) (void)hipStreamDestroy(c->stream2);
    if (c->stream) (void)hipStreamDestroy(c->stream);
  }
  delete c;
}

int zmx_set_input(zmx_ctx* c, const unsigned char* in, size_t insize) {
  DeviceGuard dev_guard(c->device);
  HIPCHK(dev_guard.err);
  if (insize + kInputPad > c->in_cap) {
    PoolFree(c, c->d_in);
    c->d_in = nullptr;
    c->in_cap = insize + kInputPad;
    HIPCHK(PoolAllocT(c, &c->d_in, c->in_cap, "d_in"));
  }
  if (insize) HIPCHK(hipMemcpyAsync(c->d_in, in, insize, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemsetAsync(c->d_in + insize, 0, kInputPad, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  c->insize = insize;
  c->h_in = in;
  return 0;
}

// Gives back the device arrays of a table set — all of them, or all but the two LZ77 stores (zmx_tables_trim).
static void ReleaseTableArrays(zmx_ctx* c, zmx_tables* t, bool keep_stores) {
  auto rel = [&](auto*& p) { PoolFree(c, p); p = nullptr; };
  rel(t->d_blocks);
  rel(t->d_tile_off);
  rel(t->d_same16);
  rel(t->d_links);
  rel(t->d_sorted_alloc);
  rel(t->d_ssame);
  rel(t->d_chunk_base);
  rel(t->d_recs);
  rel(t->d_pool);
  rel(t->d_la);
  rel(t->d_dph);
  rel(t->d_block_edges);
  rel(t->d_badpos);
  rel(t->d_code_base);
  rel(t->d_codes);
  rel(t->d_wtab);
  rel(t->d_badcodes);
  rel(t->d_seg_off);
  rel(t->d_extab);
  rel(t->d_seginfo);
  rel(t->d_counters);
  rel(t->d_prof);
  rel(t->d_tasks);
  rel(t->d_task_off);
  rel(t->d_wg_tasks);
  rel(t->d_task_kind);
  rel(t->d_run_list);
  rel(t->d_wmeta);
  rel(t->d_runin);
  rel(t->d_runout);
  rel(t->d_winroff);
  rel(t->d_winflag);
  rel(t->d_win_off);
  rel(t->d_lvl);
  rel(t->d_entry);
  rel(t->d_exit);
  rel(t->d_mid);
  rel(t->d_chk);
  rel(t->d_over);
  rel(t->d_redo);
  for (int h = 0; h < 2; ++h) { rel(t->d_rank[h]); rel(t->d_bucket[h]); }
  t->d_sorted[0] = t->d_sorted[1] = nullptr;
  PinnedGive(c, &t->h_runin, t->h_runin_cap);
  PinnedGive(c, &t->h_runout, t->h_runout_cap);
  if (!keep_stores) { rel(t->d_store[0]); rel(t->d_store[1]); }
}

void zmx_tables_free(zmx_ctx* c, zmx_tables* t) {
  if (!t) return;
  DeviceGuard dev_guard(c ? c->device : 0);
  // (an error return between a launch on the second stream and its join leaves that kernel in flight: nothing of
  //  this table set may go back to the pool under it)
  if (c && c->stream2) (void)hipStreamSynchronize(c->stream2);
  ReleaseTableArrays(c, t, false);
  delete t;
}

// What zmx_encode_blocks and zmx_store_download read of a table set are its two stores (and the host's notes on where
// each block's symbols lie): everything else — 32 bytes of match record per position, the DP codes, window records, the
// chain's snapshots — can go once the parses are final.  deflate.cc trims the tables of the optimal batch before it
// builds the tables of the fixed-tree re-parses: on incompressible or short input, where most blocks ask for one, the
// peak otherwise doubles.
int zmx_tables_trim(zmx_ctx* c, zmx_tables* t) {
  if (!t) return 0;
  DeviceGuard dev_guard(c->device);
  HIPCHK(dev_guard.err);
  HIPCHK(hipStreamSynchronize(c->stream));      // (nothing of this table set may still be in flight)
  HIPCHK(hipStreamSynchronize(c->stream2));
  ReleaseTableArrays(c, t, true);
  t->trimmed = true;
  return 0;
}

// `parent` (optional): a table set over blocks that contain the new ones.  The match record of a
// position depends on its block only through the block end: limit = min(258, end -
// pos), same[] truncated at the end, zero bytes in the hashes of the last two positions.  So the
// records of a sub-block equal the parent's except where pos + 258 > end or pos lies in the run
// of equal bytes that reaches the end — only the tiles holding such positions are recomputed,
// everything else is copied.  Hash links (k_same, k_chain) are rebuilt: they are cheap.
static unsigned EnvU32(const char* name, unsigned dflt, unsigned lo, unsigned hi) {
  const char* e = std::getenv(name);
  if (!e) return dflt;
  const long v = std::atol(e);
  return v < static_cast<long>(lo) ? lo : v > static_cast<long>(hi) ? hi : static_cast<unsigned>(v);
}
// Task geometry of the chain (zmx_dp4.h): positions per task and warm-up positions before it.
// ZOPFLI_AMD_SEG_L = 0 turns the cut off (one task per block: the serial chain).
// Task length of the chain.  A task is one serial wave, and a run is at least a task, a second-pass task and the
// serial re-runs long: with little to do (small calls: zopflipng's IDATs, files of a MB or less) short tasks cut
// that latency (1 MB: 53 -> 47 ms, 64 KiB: 32 -> 24 ms) although every task pays its 512-position warm-up; with
// a full batch 2048 and 4096 run alike and the longer tasks walk fewer positions.  ZOPFLI_AMD_SEG_L overrides
// (0 = no tasks: the serial chain).
static unsigned SegL(u64 total_positions) {
  static const bool set = std::getenv("ZOPFLI_AMD_SEG_L") != nullptr;
  static const unsigned v = EnvU32("ZOPFLI_AMD_SEG_L", 4096, 0, 1u << 24) & ~63u;
  if (set) return v;
  // (with the tasks started at cut points, below, the warm-up is ~70 positions instead of 512 and 2048 beats 4096
  //  for a full batch too: half the positions re-run where a task crosses a binade, 4.17 -> 3.75 ms per run of 100 MB;
  //  1024 loses to the per-task set-up again: 4.2 ms)
  // (a call of 64 KiB is 64 tasks of 1024 positions on 256 CUs: 512 halves the latency of every pass — 18.6 -> 16.8 ms
  //  for the call —, at 1 MB 512 loses to the per-task set-up again: 39 -> 44 ms)
  // (round 5: with k_dp4_fix judging a chunk's tasks at once the per-task set-up is gone and short tasks win further up —
  //  64 KiB: 256 13.2 ms against 13.8 with 512; 1 MB: 512 23.1 against 24.1 with 1024; 4 MB: 512 33.6 against 38.1;
  //  100 MB: 2048 3.63 ms per run against 3.94 with 1024 — tools/r05_segl.sh)
  return total_positions <= (128u << 10) ? 256u : total_positions <= (16u << 20) ? 512u : total_positions <= (48u << 20) ? 1024u : 2048u;
}
// The first task of a block is exact by construction and runs beside the others: let it cover the
// stretch where the costs double every few thousand positions and no guess would stay in its binade.
// The exact head of a block (positions run from the true initial state; the values double every few hundred
// positions there and speculative tasks would not stay inside a binade).  It is one serial wave: with few blocks in
// the batch the whole run waits for it (short: 4096), with many it hides behind the other tasks and a long head
// saves the serial re-runs of the early tasks (8192; 16384 and 4096 are within 1 %).  ZOPFLI_AMD_SEG_HEAD overrides.
static unsigned SegHead(size_t nb) {
  static const unsigned v = EnvU32("ZOPFLI_AMD_SEG_HEAD", 0, 0, 1u << 24) & ~63u;
  return v ? v : (nb >= 48 ? 8192u : 0u);       // (0: as long as a task)
}
static unsigned SegWarm() { static const unsigned v = (EnvU32("ZOPFLI_AMD_SEG_WARM", 512, 64, 1u << 20) + 63u) & ~63u; return v; }

static int BuildTables(zmx_ctx* c, const zmx_block* blocks, size_t nb, zmx_tables* t, zmx_tables* parent = nullptr, bool with_dp = true) {
  const int mk = MatchKernel();   // (one choice per build: zmx_set_match_kernel may be called meanwhile)
  t->matches_only = !with_dp;
  t->nb = nb;
  t->blocks.resize(nb);
  t->bsize.resize(nb);
  t->store_begin[0].assign(nb, 0);
  t->store_begin[1].assign(nb, 0);
  std::vector<u32> tile_off(nb + 1, 0);
  u64 pos_off = 0, reg_off = 0, la_off = 0, max_l = 0;
  for (size_t b = 0; b < nb; ++b) {
    if (blocks[b].inend < blocks[b].instart || blocks[b].inend > c->insize) {
      return FailMsg("zmx_tables_build: block outside the resident input");
    }
    BlockDesc& d = t->blocks[b];
    d.instart = blocks[b].instart;
    d.inend = blocks[b].inend;
    d.ws = d.instart > ZMX_WINDOW ? d.instart - ZMX_WINDOW : 0;
    d.pos_off = pos_off;
    d.reg_off = reg_off;
    d.la_off = la_off;
    const u64 B = d.inend - d.instart, L = d.inend - d.ws;
    if (B > 0x7fff0000ull) return FailMsg("zmx_tables_build: block too large (positions are 32-bit)");
    t->bsize[b] = static_cast<u32>(B);
    pos_off += B;
    reg_off += (L + 7) & ~7ull;
    la_off += (B + 1 + 7) & ~7ull;
    max_l = std::max(max_l, L);
    tile_off[b + 1] = tile_off[b] + static_cast<u32>((B + MT - 1) / MT);
  }
  t->total_b = pos_off;
  t->total_l = reg_off;
  t->tile_off = tile_off;
  if (nb == 0) return 0;

  // ---- reuse of the parent's records
  std::vector<u64> src_pos;
  std::vector<u64> link_lo;       // per block: first links[] index the recomputed tiles read (its L: none)
  std::vector<u32> tile_list;
  bool reuse = parent != nullptr && parent->nb > 0 && c->h_in != nullptr && parent->d_recs != nullptr &&
               parent->total_b + pos_off < (3ull << 30);
  if (reuse) {
    src_pos.resize(nb);
    link_lo.resize(nb);
    size_t pb = 0;
    for (size_t b = 0; b < nb && reuse; ++b) {
      const BlockDesc& d = t->blocks[b];
      while (pb < parent->nb && parent->blocks[pb].inend < d.inend) ++pb;   // both lists are ascending
      if (pb == parent->nb || d.instart < parent->blocks[pb].instart || d.inend > parent->blocks[pb].inend) {
        reuse = false;
        break;
      }
      const BlockDesc& pd = parent->blocks[pb];
      src_pos[b] = pd.pos_off + (d.instart - pd.instart);
      const u64 B = d.inend - d.instart;
      link_lo[b] = d.inend - d.ws;
      if (B == 0 || d.inend == pd.inend) continue;   // same end: every record is the same
      // first position whose record may differ
      u64 t0 = B > ZMX_MAX_MATCH ? d.inend - ZMX_MAX_MATCH : d.instart;
      const unsigned char lastb = c->h_in[d.inend - 1];
      u64 r = d.inend - 1;
      while (r > d.instart && d.inend - r < 65600 && c->h_in[r - 1] == lastb) --r;
      if (r < t0) t0 = r;
      const u32 tile_first = static_cast<u32>((t0 - d.instart) / MT);
      for (u32 tile = tile_first; tile < tile_off[b + 1] - tile_off[b]; ++tile) {
        tile_list.push_back(tile_off[b] + tile);
      }
      // the walks of those tiles stay inside the 32 KiB before them (lz77.c:464)
      const u64 first_index = d.instart + static_cast<u64>(tile_first) * MT - d.ws;
      link_lo[b] = first_index > ZMX_WINDOW ? first_index - ZMX_WINDOW : 0;
    }
  }

  HIPCHK(PoolAlloc(c, &t->d_blocks, nb));
  HIPCHK(PoolAlloc(c, &t->d_tile_off, nb + 1));
  HIPCHK(PoolAlloc(c, &t->d_same16, reg_off));
#ifdef ZMX_EXPERIMENTS
  t->buckets = mk == 3 || mk == 4;
#else
  t->buckets = false;
#endif
  t->chunk_base.assign(nb + 1, 0);
  for (size_t b = 0; b < nb; ++b) {
    const u64 L = t->blocks[b].inend - t->blocks[b].ws;
    t->chunk_base[b + 1] = t->chunk_base[b] + static_cast<u32>((L + BK_CH - 1) / BK_CH);
  }
#ifdef ZMX_EXPERIMENTS
  if (t->buckets) {
    // (the two orders in one allocation, M4_PAD entries in front: k_match4 reads 16 bytes at a time, from up to 15 entries
    //  below a chunk's first one, and takes the second order as an offset from the first)
    HIPCHK(PoolAlloc(c, &t->d_sorted_alloc, 2 * reg_off + 2 * M4_PAD));
    t->d_sorted[0] = t->d_sorted_alloc + M4_PAD;
    t->d_sorted[1] = t->d_sorted[0] + reg_off + M4_PAD;
    for (int h = 0; h < 2; ++h) {
      HIPCHK(PoolAlloc(c, &t->d_rank[h], reg_off));
      HIPCHK(PoolAlloc(c, &t->d_bucket[h], static_cast<size_t>(t->chunk_base[nb]) * 32768u));
    }
    HIPCHK(PoolAlloc(c, &t->d_ssame, reg_off));
    HIPCHK(PoolAlloc(c, &t->d_chunk_base, nb + 1));
    HIPCHK(hipMemcpyAsync(t->d_chunk_base, t->chunk_base.data(), (nb + 1) * sizeof(u32), hipMemcpyHostToDevice, c->stream));
  } else
#endif
  {
    HIPCHK(PoolAlloc(c, &t->d_links, reg_off));
  }
  HIPCHK(PoolAlloc(c, &t->d_recs, pos_off * 8));
  HIPCHK(PoolAlloc(c, &t->d_la, la_off));
  HIPCHK(PoolAlloc(c, &t->d_store[0], pos_off));
  HIPCHK(PoolAlloc(c, &t->d_store[1], pos_off));
  {
    const size_t in_cost = 0, in_min = in_cost + nb * ZMX_HIST * sizeof(double), in_info = in_min + nb * sizeof(double),
                 in_slot = in_info + 3 * nb * sizeof(float);
    t->runin_bytes = in_slot + nb * sizeof(int);
    const size_t out_hist = 0, out_nsym = out_hist + nb * ZMX_HIST * sizeof(u32), out_stats = out_nsym + nb * sizeof(u32),
                 out_flags = out_stats + 8 * sizeof(u32);
    t->runout_bytes = out_flags + 4 * sizeof(u32);
    HIPCHK(PoolAlloc(c, &t->d_runin, t->runin_bytes));
    HIPCHK(PoolAlloc(c, &t->d_runout, t->runout_bytes));
    HIPCHK(PinnedTake(c, &t->h_runin, t->runin_bytes, &t->h_runin_cap));
    HIPCHK(PinnedTake(c, &t->h_runout, t->runout_bytes, &t->h_runout_cap));
    t->d_cost = reinterpret_cast<double*>(t->d_runin + in_cost);
    t->d_mincost = reinterpret_cast<double*>(t->d_runin + in_min);
    t->d_runinfo = reinterpret_cast<float*>(t->d_runin + in_info);
    t->d_slot = reinterpret_cast<int*>(t->d_runin + in_slot);
    t->d_hist = reinterpret_cast<u32*>(t->d_runout + out_hist);
    t->d_nsym = reinterpret_cast<u32*>(t->d_runout + out_nsym);
    t->d_segstats = reinterpret_cast<u32*>(t->d_runout + out_stats);
    t->d_flags = reinterpret_cast<u32*>(t->d_runout + out_flags);
    HIPCHK(hipMemsetAsync(t->d_segstats, 0, 12 * sizeof(u32), c->stream));
  }
  HIPCHK(PoolAlloc(c, &t->d_dph, pos_off));
  HIPCHK(PoolAlloc(c, &t->d_block_edges, nb));
  t->badpos_words = pos_off / 32 + 4;
  HIPCHK(PoolAlloc(c, &t->d_badpos, t->badpos_words));
  HIPCHK(PoolAlloc(c, &t->d_code_base, nb));
  HIPCHK(PoolAlloc(c, &t->d_wtab, nb * ZMX_WTAB));
  HIPCHK(PoolAlloc(c, &t->d_badcodes, nb * 40));
  HIPCHK(PoolAlloc(c, &t->d_counters, 48));
  HIPCHK(hipMemcpyAsync(t->d_blocks, t->blocks.data(), nb * sizeof(BlockDesc), hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(t->d_tile_off, tile_off.data(), (nb + 1) * sizeof(u32), hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemsetAsync(t->d_flags, 0, 4 * sizeof(u32), c->stream));

  HIPCHK(hipEventRecord(c->ev[0], c->stream));
  PoolScope hash_tmp(c);
  u16* d_lev = nullptr;    // k_levels / k_rank2 (ZOPFLI_AMD_MATCH=5), alive until the match kernel has run
  u16* d_tot2 = nullptr;
  u16* d_rank2 = nullptr;
  uint4* d_xrec = nullptr;
  u32* d_tot12 = nullptr;
  unsigned long long* d_energy = nullptr;   // k_hits (kernel 0 = per block: k_match5 where the chains are long)
  u32* d_cmax = nullptr;                    // k_hits: the chunks' largest classes (k_rank2)
  bool skip_any = false, skip_all = false;  // some / all blocks of this build take k_match5
  double skip_positions = 0;                // positions of those blocks
  unsigned long long* d_m5stats = nullptr;  // k_match5's per-wave sums
  bool join_stream2 = false;
  auto launch_hash = [&](const u64* d_link_lo) -> int {
    if (max_l == 0) return 0;
    const dim3 g1(static_cast<unsigned>((max_l + 256 * SAME_CH - 1) / (256 * SAME_CH)), static_cast<unsigned>(nb));
    hipLaunchKernelGGL(k_same, g1, dim3(256), 0, c->stream, c->d_in, t->d_blocks, t->d_same16, d_link_lo);
    KCHK(c, "k_same");
    HIPCHK(hipGetLastError());
    const dim3 g2(static_cast<unsigned>((max_l + CH_EMIT - 1) / CH_EMIT), static_cast<unsigned>(nb), 2);
#ifdef ZMX_EXPERIMENTS
    if (t->buckets) {
      BucketParams kp;
      kp.in = c->d_in;
      kp.blocks = t->d_blocks;
      kp.same16 = t->d_same16;
      kp.chunk_base = t->d_chunk_base;
      kp.link_lo = d_link_lo;
      for (int h = 0; h < 2; ++h) { kp.sorted[h] = t->d_sorted[h]; kp.rank[h] = t->d_rank[h]; kp.bucket[h] = t->d_bucket[h]; }
      kp.ssame = t->d_ssame;
      hipLaunchKernelGGL(k_bucket, g2, dim3(BK_THREADS), BK_LDS_BYTES, c->stream, kp);
      KCHK(c, "k_bucket");
    } else
#endif
    {
      hipLaunchKernelGGL(k_chain, g2, dim3(64), CH_LDS_BYTES, c->stream, c->d_in, t->d_blocks, t->d_same16, t->d_links, d_link_lo);
      KCHK(c, "k_chain");
      if ((mk == 5 || mk == 0) && d_link_lo == nullptr) {
        // The skip-walk (k_match5) for the blocks whose chains are long: k_hits estimates the hits per position the
        // reference's walk would make, block by block; kernel 5 forces it for every block.  (Whole blocks only: a
        // table built from a parent recomputes a few tiles with k_match2.)
        skip_any = mk == 5;
        if (mk == 5) for (size_t b = 0; b < nb; ++b) skip_positions += static_cast<double>(t->blocks[b].inend - t->blocks[b].instart);
        const unsigned hits_chunks = static_cast<unsigned>((max_l + RK_CH - 1) / RK_CH);
        {
          // k_hits: the blocks' hit estimates (kernel 0: which walk a block gets) and the chunks' largest classes (k_rank2:
          // where the 8192-hit cap can bind; ZOPFLI_AMD_RANK_ALL=1: ranks everywhere, as in round 4)
          static const bool rank_all = [] { const char* e = std::getenv("ZOPFLI_AMD_RANK_ALL"); return e && std::atoi(e) != 0; }();
          if (!d_energy) HIPCHK(hash_tmp.AllocT(&d_energy, nb, "d_energy"));
          if (!d_cmax && !rank_all) HIPCHK(hash_tmp.AllocT(&d_cmax, nb * static_cast<size_t>(hits_chunks) + 1, "d_cmax"));
          HIPCHK(hipMemsetAsync(d_energy, 0, nb * sizeof(unsigned long long), c->stream));
          if (d_cmax) HIPCHK(hipMemsetAsync(d_cmax, 0, (nb * static_cast<size_t>(hits_chunks) + 1) * sizeof(u32), c->stream));
          HitsParams hp;
          hp.in = c->d_in;
          hp.blocks = t->d_blocks;
          hp.same16 = t->d_same16;
          hp.energy = d_energy;
          hp.cmax = d_cmax;
          const dim3 g5(hits_chunks, static_cast<unsigned>(nb));
          hipLaunchKernelGGL(k_hits, g5, dim3(RK_THREADS), 0, c->stream, hp);
          KCHK(c, "k_hits");
        }
        if (mk == 0) {
          std::vector<unsigned long long> energy(nb);
          HIPCHK(hipMemcpyAsync(energy.data(), d_energy, nb * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
          HIPCHK(hipStreamSynchronize(c->stream));
          size_t on = 0;
          for (size_t b = 0; b < nb; ++b) {
            if (energy[b] > MatchAutoHits() * (t->blocks[b].inend - t->blocks[b].ws)) {
              ++on;
              skip_positions += static_cast<double>(t->blocks[b].inend - t->blocks[b].instart);
            }
          }
          skip_any = on != 0;
          skip_all = on == nb;
        }
        if (skip_any) {
          if (!d_lev) HIPCHK(hash_tmp.AllocT(&d_lev, static_cast<size_t>(LV_N) * reg_off, "d_lev"));
          if (!d_tot2) HIPCHK(hash_tmp.AllocT(&d_tot2, 2 * reg_off, "d_tot"));
          if (!d_rank2) HIPCHK(hash_tmp.AllocT(&d_rank2, 2 * reg_off, "d_rank"));
          if (!d_tot12) HIPCHK(hash_tmp.AllocT(&d_tot12, reg_off, "d_tot12"));
          if (!d_xrec) HIPCHK(hash_tmp.AllocT(&d_xrec, 2 * reg_off, "d_xrec"));
          LevelParams lp;
          lp.in = c->d_in;
          lp.blocks = t->d_blocks;
          lp.lev = d_lev;
          lp.total_l = reg_off;
          lp.energy = mk == 0 ? d_energy : nullptr;
          lp.thr = MatchAutoHits();
          const dim3 g4(static_cast<unsigned>((max_l + LV_CH - 1) / LV_CH), static_cast<unsigned>(nb), LV_N);
          hipLaunchKernelGGL(k_levels, g4, dim3(64), 0, c->stream, lp);
          KCHK(c, "k_levels");
          RankParams rp;
          rp.in = c->d_in;
          rp.blocks = t->d_blocks;
          rp.links = t->d_links;
          rp.same16 = t->d_same16;
          rp.lev = d_lev;
          rp.total_l = reg_off;
          rp.tot = d_tot2;
          rp.rank = d_rank2;
          rp.xrec = d_xrec;
          rp.tot12 = d_tot12;
          rp.energy = lp.energy;
          rp.thr = lp.thr;
          rp.cmax = d_cmax;
          rp.cmax_stride = hits_chunks;
          const dim3 g3(static_cast<unsigned>((max_l + RK_CH - 1) / RK_CH), static_cast<unsigned>(nb));
          hipLaunchKernelGGL(k_rank2, g3, dim3(RK_THREADS), 0, c->stream, rp);
          KCHK(c, "k_rank2");
        }
      }
    }
    return 0;
  };
  {
    u64* d_link_lo = nullptr;
    if (reuse) {
      HIPCHK(hash_tmp.AllocT(&d_link_lo, nb, "d_link_lo"));
      HIPCHK(hipMemcpyAsync(d_link_lo, link_lo.data(), nb * sizeof(u64), hipMemcpyHostToDevice, c->stream));
    }
    if (launch_hash(d_link_lo) != 0) return -1;
    t->links_partial = reuse;
  }

  if (mk != 3 && !c->d_scratch) HIPCHK(PoolAllocT(c, &c->d_scratch, static_cast<size_t>(kMatchGrid) * M2_THREADS * SCRATCH_CPS, "d_scratch"));
  HIPCHK(hipEventRecord(c->ev[1], c->stream));
  double match_positions = 0;
  double m5_lane_iters = 0, m5_iters = 0;   // k_match5's own counts
  // the match-table kernel over `total_tiles` tiles (all of them, or those of tile_list)
  auto launch_match = [&](u32* pool, u32 pool_cap, u32 total_tiles, const u32* d_tiles, bool prof) -> int {
    if (total_tiles == 0) return 0;
#ifdef ZMX_EXPERIMENTS
    if (t->buckets) {
      Match3Params mp;
      mp.in = c->d_in;
      mp.blocks = t->d_blocks;
      mp.tile_off = t->d_tile_off;
      mp.nb = static_cast<u32>(nb);
      mp.total_tiles = total_tiles;
      mp.same16 = t->d_same16;
      mp.chunk_base = t->d_chunk_base;
      for (int h = 0; h < 2; ++h) { mp.sorted[h] = t->d_sorted[h]; mp.rank[h] = t->d_rank[h]; mp.bucket[h] = t->d_bucket[h]; }
      mp.ssame = t->d_ssame;
      mp.recs = t->d_recs;
      mp.pool = pool;
      mp.pool_cap = pool_cap;
      mp.counters = t->d_counters;
      mp.tile_list = d_tiles;
      mp.scratch = c->d_scratch;
      if (mk == 3) {
        if (prof) hipLaunchKernelGGL((k_match3<true>), dim3(kMatchGrid3), dim3(M3_THREADS), 0, c->stream, mp);
        else hipLaunchKernelGGL((k_match3<false>), dim3(kMatchGrid3), dim3(M3_THREADS), 0, c->stream, mp);
        KCHK(c, "k_match3");
      } else {
        if (prof) hipLaunchKernelGGL((k_match4<true>), dim3(kMatchGrid3), dim3(M4_THREADS), 0, c->stream, mp);
        else hipLaunchKernelGGL((k_match4<false>), dim3(kMatchGrid3), dim3(M4_THREADS), 0, c->stream, mp);
        KCHK(c, "k_match4");
      }
      return 0;
    }
#endif
    MatchParams mp;
    mp.in = c->d_in;
    mp.blocks = t->d_blocks;
    mp.tile_off = t->d_tile_off;
    mp.nb = static_cast<u32>(nb);
    mp.total_tiles = total_tiles;
    mp.links = t->d_links;
    mp.recs = t->d_recs;
    mp.pool = pool;
    mp.pool_cap = pool_cap;
    mp.counters = t->d_counters;
    mp.scratch = c->d_scratch;
    mp.tile_list = d_tiles;
    mp.skip_energy = nullptr;
    mp.skip_thr = 0;
    if ((mk == 5 || mk == 0) && d_tiles == nullptr && skip_any) {
      if (!c->d_scratch5) HIPCHK(PoolAllocT(c, &c->d_scratch5, static_cast<size_t>(kMatchGrid5) * M5_THREADS * SCRATCH_CPS, "d_scratch5"));
      Match5Params q;
      q.m = mp;
      q.m.scratch = c->d_scratch5;
      q.xrec = d_xrec;
      q.tot12 = d_tot12;
      q.energy = mk == 0 ? d_energy : nullptr;
      q.thr = MatchAutoHits();
      const size_t m5_waves = static_cast<size_t>(kMatchGrid5) * (M5_THREADS / 64);
      if (!d_m5stats) HIPCHK(hash_tmp.AllocT(&d_m5stats, 2 * m5_waves, "d_m5stats"));
      HIPCHK(hipMemsetAsync(d_m5stats, 0, 2 * m5_waves * sizeof(unsigned long long), c->stream));
      q.wave_stats = d_m5stats;
      {
        // positions a wave takes at a time: larger pieces keep the lanes busier (fewer ends of a piece, where lanes
        // wait for the piece's longest walks), smaller ones the waves when there are few tiles.  100 MB, pieces of 512 /
        // 1024 / 2048 positions: T 22.4 / 20.0 / 20.4 ms, P 30.1 / 27.9 / 29.5, B 97.1 / 93.1 / 100.0 (ZOPFLI_AMD_M5_UNIT =
        // 2 / 1 / 0 forces one)
        static const int forced = [] { const char* e = std::getenv("ZOPFLI_AMD_M5_UNIT"); return e ? std::atoi(e) : -1; }();
        const u64 waves = static_cast<u64>(kMatchGrid5) * (M5_THREADS / 64);
        q.sub_shift = forced >= 0 ? static_cast<u32>(forced > 2 ? 2 : forced) : (mp.total_tiles >= 4 * waves ? 1u : 2u);
      }
      if (mk == 5 || skip_all) {
        hipLaunchKernelGGL(k_match5, dim3(kMatchGrid5), dim3(M5_THREADS), 0, c->stream, q);   // (no profile counts: tools/match_skip_model.c has the entries touched)
        KCHK(c, "k_match5");
        return 0;
      }
      // Some blocks each: k_match5 on the second stream beside k_match2 (which passes over k_match5's blocks) — the
      // skip-walk of a few heavy blocks is a handful of long-running waves, the rest of the device is k_match2's.
      HIPCHK(hipEventRecord(c->ev2[0], c->stream));
      HIPCHK(hipStreamWaitEvent(c->stream2, c->ev2[0], 0));
      hipLaunchKernelGGL(k_match5, dim3(kMatchGrid5), dim3(M5_THREADS), 0, c->stream2, q);
      HIPCHK(hipGetLastError());
      c->stream2_outstanding = true;
      HIPCHK(hipEventRecord(c->ev2[1], c->stream2));
      join_stream2 = true;
      mp.skip_energy = d_energy;
      mp.skip_thr = MatchAutoHits();
    }
    const bool filt = MatchFilter();
    if (prof && filt) hipLaunchKernelGGL((k_match2<true, true>), dim3(kMatchGrid), dim3(M2_THREADS), 0, c->stream, mp);
    else if (prof) hipLaunchKernelGGL((k_match2<true, false>), dim3(kMatchGrid), dim3(M2_THREADS), 0, c->stream, mp);
    else if (filt) hipLaunchKernelGGL((k_match2<false, true>), dim3(kMatchGrid), dim3(M2_THREADS), 0, c->stream, mp);
    else hipLaunchKernelGGL((k_match2<false, false>), dim3(kMatchGrid), dim3(M2_THREADS), 0, c->stream, mp);
    if (join_stream2) {
      join_stream2 = false;
      HIPCHK(hipStreamWaitEvent(c->stream, c->ev2[1], 0));
      c->stream2_outstanding = false;      // (whatever reuses the arrays does so in `stream`'s order, behind the kernel)
    }
    KCHK(c, "k_match2");
    return 0;
  };

  if (reuse) {
    // copy every record, adopt the parent's change-point pool (copied records point into it) and
    // recompute the listed tiles; on pool overflow fall through to the full build
    u64* d_src_pos = nullptr;
    u32* d_tile_list = nullptr;
    PoolScope tmp(c);
    HIPCHK(tmp.AllocT(&d_src_pos, nb, "d_src_pos"));
    HIPCHK(tmp.AllocT(&d_tile_list, tile_list.size(), "d_tile_list"));
    HIPCHK(hipMemcpyAsync(d_src_pos, src_pos.data(), nb * sizeof(u64), hipMemcpyHostToDevice, c->stream));
    if (!tile_list.empty()) {
      HIPCHK(hipMemcpyAsync(d_tile_list, tile_list.data(), tile_list.size() * sizeof(u32), hipMemcpyHostToDevice, c->stream));
    }
    CopyRecsParams cp;
    cp.blocks = t->d_blocks;
    cp.src_pos = d_src_pos;
    cp.src = reinterpret_cast<const uint4*>(parent->d_recs);
    cp.dst = reinterpret_cast<uint4*>(t->d_recs);
    u64 max_b = 0;
    for (size_t b = 0; b < nb; ++b) max_b = std::max<u64>(max_b, t->bsize[b]);
    const unsigned gx = static_cast<unsigned>(std::min<u64>(std::max<u64>((max_b * 2 + 256 * 16 - 1) / (256 * 16), 1), 4096));
    hipLaunchKernelGGL(k_copy_recs, dim3(gx, static_cast<unsigned>(nb)), dim3(256), 0, c->stream, cp);
    KCHK(c, "k_copy_recs");
    HIPCHK(hipGetLastError());
    // the pool cursor continues where the parent's stopped
    HIPCHK(hipMemsetAsync(t->d_counters, 0, 48 * sizeof(u32), c->stream));
    HIPCHK(hipMemcpyAsync(t->d_counters, parent->d_counters, sizeof(u32), hipMemcpyDeviceToDevice, c->stream));
    if (launch_match(parent->d_pool, parent->pool_cap, static_cast<u32>(tile_list.size()), d_tile_list, false) != 0) return -1;
    u32 counters[2] = {0, 0};
    HIPCHK(hipMemcpyAsync(counters, t->d_counters, sizeof(counters), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if ((counters[1] & 1u) == 0) {
      t->d_pool = parent->d_pool;          // ownership moves: the parent must not be used for records again
      t->pool_cap = parent->pool_cap;
      parent->d_pool = nullptr;
      parent->pool_cap = 0;
    } else {
      reuse = false;                       // pool overflow: build everything with a pool of our own,
      if (launch_hash(nullptr) != 0) return -1;   // which needs the hash arrays of whole blocks
      t->links_partial = false;
    }
  }

  // Change points beyond the 8 inline ones go to a pool; start with 4 entries per
  // position and grow on overflow (worst case 256 per position).
  // (ZOPFLI_AMD_POOL_ENTRIES: test hook, the first pool has that many entries in all, so that the
  // overflow / retry path and the fall-through from a reused parent pool run on small inputs)
  static const u64 pool_entries = [] { const char* e = std::getenv("ZOPFLI_AMD_POOL_ENTRIES"); return e ? static_cast<u64>(std::atoll(e)) : 0ull; }();
  u64 per_pos = 4;
  for (bool first_try = true; !reuse; first_try = false) {
    u64 cap = std::max<u64>(pos_off * per_pos, 1u << 16);
    if (first_try && pool_entries) cap = pool_entries;
    if (cap > 0xfffffff0ull) cap = 0xfffffff0ull;
    PoolFree(c, t->d_pool);
    t->d_pool = nullptr;
    HIPCHK(PoolAlloc(c, &t->d_pool, cap));
    t->pool_cap = static_cast<u32>(cap);
    HIPCHK(hipMemsetAsync(t->d_counters, 0, 48 * sizeof(u32), c->stream));
    static const bool match_prof = std::getenv("ZOPFLI_AMD_PROF") != nullptr;
    if (launch_match(t->d_pool, t->pool_cap, tile_off[nb], nullptr, match_prof) != 0) return -1;
    u32 counters[2] = {0, 0};
    HIPCHK(hipMemcpyAsync(counters, t->d_counters, sizeof(counters), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    match_positions += static_cast<double>(pos_off);
    if (d_m5stats) {      // (of the last attempt: the pool may grow and the kernel run again)
      const size_t m5_waves = static_cast<size_t>(kMatchGrid5) * (M5_THREADS / 64);
      std::vector<unsigned long long> ws(2 * m5_waves);
      HIPCHK(hipMemcpy(ws.data(), d_m5stats, ws.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
      m5_lane_iters = m5_iters = 0;
      for (size_t w = 0; w < m5_waves; ++w) { m5_lane_iters += static_cast<double>(ws[2 * w]); m5_iters += static_cast<double>(ws[2 * w + 1]); }
    }
    if (counters[1] & 2u) {
      u32 dbg[8] = {0};
      HIPCHK(hipMemcpy(dbg, t->d_counters + 32, sizeof(dbg), hipMemcpyDeviceToHost));
      char buf[320];
      std::snprintf(buf, sizeof(buf), "zmx_tables_build: k_match5's wave loop did not end (state %08x xd %u curd %u bestlen %u limit %u nlink %u eqd %u "
                    "li %u idx %u same %u cur %u bestdist %u)", dbg[0], dbg[1], dbg[2], dbg[3] & 0xffffu, dbg[3] >> 16, dbg[4] & 0xffffu, dbg[4] >> 16,
                    dbg[5], dbg[6] & 0xffffu, dbg[6] >> 16, dbg[7] & 0xffffu, dbg[7] >> 16);
      return FailFault(buf);
    }
    if ((counters[1] & 1u) == 0) break;
    if (per_pos >= 256) return FailMsg("zmx_tables_build: change-point pool overflow");
    per_pos *= 8;
  }
  {
    HIPCHK(hipEventRecord(c->ev[2], c->stream));
    HIPCHK(hipEventSynchronize(c->ev[2]));
    float ms_hash = 0, ms_match = 0;
    HIPCHK(hipEventElapsedTime(&ms_hash, c->ev[0], c->ev[1]));
    HIPCHK(hipEventElapsedTime(&ms_match, c->ev[1], c->ev[2]));
    // (thread-local: no lock)
    g_match_stats[0] += ms_match * 1e-3;
    g_match_stats[1] += ms_hash * 1e-3;
    g_match_stats[2] += 1;
    g_match_stats[3] += reuse ? static_cast<double>(tile_list.size()) * MT : match_positions;
    if (skip_any && !reuse) {
      g_match5_stats[0] += m5_lane_iters;
      g_match5_stats[1] += m5_iters;
      g_match5_stats[2] += skip_positions;
    }
    if (std::getenv("ZOPFLI_AMD_PROF") && !reuse) {
      unsigned long long hc[2] = {0, 0};
      HIPCHK(hipMemcpy(hc, t->d_counters + 4, sizeof(hc), hipMemcpyDeviceToHost));
      const double pos = static_cast<double>(pos_off);
      std::fprintf(stderr, "%s: %.2f ms for %.0f positions: %.1f chain hits per "
                   "position, %.1f of 64 lanes with a hit per wave-loop iteration; %.1f SIMD cycles per hit (2.4 GHz, 1024 SIMDs)\n", mk == 5 || mk == 0 ? "k_match5 / k_match2 (hits = entries touched)" : mk == 4 ? "k_match4" : mk == 3 ? "k_match3" : "k_match2", ms_match, pos,
                   static_cast<double>(hc[0]) / pos, static_cast<double>(hc[0]) / static_cast<double>(hc[1] ? hc[1] : 1),
                   ms_match * 1e-3 * 2.4e9 * 1024 / static_cast<double>(hc[0] ? hc[0] : 1));
      if (mk == 3) {
        unsigned long long h3[3] = {0, 0, 0};
        HIPCHK(hipMemcpy(h3, t->d_counters + 16, sizeof(h3), hipMemcpyDeviceToHost));
        const double nbat = static_cast<double>(hc[1] ? hc[1] : 1);
        std::fprintf(stderr, "k_match3: per batch of <= 64 candidates: %.2f pass the 4-byte filter, %.2f steps of the byte compare, "
                     "%.1f %% of the batches change nothing; %.0f SIMD cycles per batch\n", static_cast<double>(h3[0]) / nbat,
                     static_cast<double>(h3[1]) / nbat, 100.0 * static_cast<double>(h3[2]) / nbat, ms_match * 1e-3 * 2.4e9 * 1024 / nbat);
      }
    }
  }

  // (zmx_tables_build_matches: the greedy pass over master blocks that will be split wants the matches only; the codes
  //  of its DP edges — two bytes for each of up to 258 edges a position, 52 GB for 100 MB of long runs — would be
  //  written, never read, and held while the tables of the split blocks allocate their own)
  if (with_dp) {
  // DP row layout (k_rowscan), then the edges as weight codes (k_codes) and a buffer descriptor per row
  RowScanParams rp;
  rp.blocks = t->d_blocks;
  rp.recs = t->d_recs;
  rp.dph = t->d_dph;
  rp.block_edges = t->d_block_edges;
  // (ZOPFLI_AMD_RUN_CODES=1: codes for every row, as in round 5; the serial chain — ZOPFLI_AMD_SEG_L=0, k_dp4's pipeline
  //  over whole blocks — needs them)
  static const bool all_codes = [] { const char* e = std::getenv("ZOPFLI_AMD_RUN_CODES"); return e && std::atoi(e) != 0; }();
  rp.codeless = !all_codes && SegL(t->total_b) != 0 ? 1u : 0u;
  hipLaunchKernelGGL(k_rowscan, dim3(static_cast<unsigned>(nb)), dim3(1024), 0, c->stream, rp);
  KCHK(c, "k_rowscan");
  HIPCHK(hipGetLastError());
  t->block_edges.resize(nb);
  HIPCHK(hipMemcpyAsync(t->block_edges.data(), t->d_block_edges, nb * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  {
    // ZOPFLI_AMD_CODE_BUDGET_MB: what the codes of one batch may take (two bytes per DP edge; a position
    // has 1..258 edges).  Beyond it the caller is told to come back with fewer blocks (kTooLarge).
    static const u64 env_mb = [] {
      const char* e = std::getenv("ZOPFLI_AMD_CODE_BUDGET_MB");
      return e ? static_cast<u64>(std::max<long>(1, std::atol(e))) : 0ull;
    }();
    const u64 budget = (env_mb ? env_mb * (1ull << 20) : static_cast<u64>(c->code_budget)) / sizeof(u16);
    std::vector<u64> code_base(nb, 0);
    u64 cur = 0;
    for (size_t b = 0; b < nb; ++b) {
      if (t->block_edges[b] > 0xffff0000ull) return FailMsg("zmx_tables_build: block too large (DP row offsets are 32-bit)");
      code_base[b] = cur;
      cur += ((t->block_edges[b] + DP_PIECE - 1) & ~static_cast<u64>(DP_PIECE - 1)) + DP_PIECE;   // (the ring's DMA reads whole pieces)
    }
    if (cur > budget && nb > 1) {
      g_err = "zmx_tables_build: the batch needs more room for its DP edges than ZOPFLI_AMD_CODE_BUDGET_MB allows";
      g_err_class = ZMX_ERR_REFUSED;   // (this batch, anywhere; the caller comes back with fewer blocks)
      return kTooLarge;
    }
    HIPCHK(PoolAlloc(c, &t->d_codes, cur + 2048));   // (k_dp5_spec stages whole KB: it reads a little past a window's rows)
    HIPCHK(hipMemcpyAsync(t->d_code_base, code_base.data(), nb * sizeof(u64), hipMemcpyHostToDevice, c->stream));
    CodeParams kp;
    kp.blocks = t->d_blocks;
    kp.tile_off = t->d_tile_off;
    kp.nb_total = static_cast<u32>(nb);
    kp.recs = t->d_recs;
    kp.pool = t->d_pool;
    kp.dph = t->d_dph;
    kp.codes = t->d_codes;
    kp.code_base = t->d_code_base;
    if (tile_off[nb]) hipLaunchKernelGGL(k_codes, dim3(tile_off[nb]), dim3(256), 0, c->stream, kp);
    KCHK(c, "k_codes");
    HIPCHK(hipGetLastError());
    // row lengths, first rows and flags of the 32-position windows of k_dp5_spec
    t->win_off.assign(nb + 1, 0);
    for (size_t b = 0; b < nb; ++b) t->win_off[b + 1] = t->win_off[b] + (t->bsize[b] + 31) / 32 + 1;   // (+ 1: where the block's rows end)
    HIPCHK(PoolAlloc(c, &t->d_winflag, t->win_off[nb]));
    HIPCHK(PoolAlloc(c, &t->d_winroff, t->win_off[nb]));
    HIPCHK(PoolAlloc(c, &t->d_wmeta, (static_cast<size_t>(t->win_off[nb]) + 2) * D5_WM));
    HIPCHK(PoolAlloc(c, &t->d_win_off, nb + 1));
    HIPCHK(hipMemcpyAsync(t->d_win_off, t->win_off.data(), (nb + 1) * sizeof(u32), hipMemcpyHostToDevice, c->stream));
    MkDescParams mp;
    mp.blocks = t->d_blocks;
    mp.dph = t->d_dph;
    mp.wmeta = t->d_wmeta;
    mp.winroff = t->d_winroff;
    mp.win_off = t->d_win_off;
    mp.winflag = t->d_winflag;
    u32 max_b = 1;
    for (size_t b = 0; b < nb; ++b) max_b = std::max(max_b, t->bsize[b]);
    hipLaunchKernelGGL(k_mkdesc, dim3((max_b + 255) / 256, static_cast<unsigned>(nb)), dim3(256), 0, c->stream, mp);
    KCHK(c, "k_mkdesc");
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));   // (code_base is a local)
  }
  // the chain's tasks (zmx_dp4.h): SEG_L positions each, the last one of a block takes the remainder
  {
    const u32 L = SegL(t->total_b), warm = SegWarm();
    const u32 head = std::max(SegHead(nb), L);
    t->task_off.assign(nb + 1, 0);
    t->tasks.clear();
    for (size_t b = 0; b < nb; ++b) {
      const u32 B = t->bsize[b];
      // the head [0, head), then tasks of L positions; the last one takes the remainder
      const u32 n = (L == 0 || B < head + L) ? 1u : 1u + (B - head) / L;
      for (u32 s = 0; s < n; ++s) {
        SegTask k;
        k.block = static_cast<u32>(b);
        k.pout = s == 0 ? 0u : head + (s - 1) * L;
        k.q = s == 0 ? 0u : k.pout - std::min(warm, k.pout);
        k.pend = s + 1 == n ? B + 1 : head + s * L;
        t->tasks.push_back(k);
      }
      t->task_off[b + 1] = static_cast<u32>(t->tasks.size());
    }
    const size_t nt = t->tasks.size();
    HIPCHK(PoolAlloc(c, &t->d_tasks, nt));
    HIPCHK(PoolAlloc(c, &t->d_task_off, nb + 1));
    HIPCHK(PoolAlloc(c, &t->d_lvl, nt));
    HIPCHK(PoolAlloc(c, &t->d_entry, nt));
    HIPCHK(PoolAlloc(c, &t->d_exit, nt));
    HIPCHK(PoolAlloc(c, &t->d_mid, nt));
    HIPCHK(PoolAlloc(c, &t->d_chk, nt));
    HIPCHK(PoolAlloc(c, &t->d_over, nt * SEG_OVER));
    HIPCHK(PoolAlloc(c, &t->d_redo, 4 + nt * 4));
    HIPCHK(hipMemcpyAsync(t->d_tasks, t->tasks.data(), nt * sizeof(SegTask), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(t->d_task_off, t->task_off.data(), (nb + 1) * sizeof(u32), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemsetAsync(t->d_segstats, 0, 8 * sizeof(u32), c->stream));
    // start the tasks at cut points of the DP where there is one close enough (zmx_dp5.h: k_cutpoints);
    // ZOPFLI_AMD_SEG_CUTS = how far before a task's first owned position to look, 0 = every task warms up
    // (the search costs 1.1 ms per 100 MB at 1024, the configuration the whole GPU suite ran with; 512 would halve
    //  it and finds the same cut point for 99.8 % of the tasks of text)
    static const u32 cut_depth = EnvU32("ZOPFLI_AMD_SEG_CUTS", 1024, 0, 1u << 16);
    // ZOPFLI_AMD_SEG_MERGE=0: never merge tasks (every task that finds no cut point warms up, as in round 2)
    static const bool merge_on = EnvU32("ZOPFLI_AMD_SEG_MERGE", 1, 0, 1) != 0;
    if (cut_depth && nt) {
      PoolScope tmp(c);
      u32* d_found = nullptr;
      u32* d_wide = nullptr;
      HIPCHK(tmp.AllocT(&d_found, 2, "d_found"));
      HIPCHK(tmp.AllocT(&d_wide, nt, "d_wide"));
      HIPCHK(hipMemsetAsync(d_found, 0, 2 * sizeof(u32), c->stream));
      HIPCHK(hipMemsetAsync(d_wide, 0, nt * sizeof(u32), c->stream));
      CutParams cp;
      cp.blocks = t->d_blocks;
      cp.dph = t->d_dph;
      cp.tasks = t->d_tasks;
      cp.depth = cut_depth;
      cp.warm = warm;
      cp.found = d_found;
      cp.wide = d_wide;
      hipLaunchKernelGGL(k_cutpoints, dim3(static_cast<unsigned>(nt)), dim3(64), 0, c->stream, cp);
      KCHK(c, "k_cutpoints");
      static const bool prof = std::getenv("ZOPFLI_AMD_PROF") != nullptr;
      u32 found[2] = {0, 0};
      std::vector<u32> wide(nt);
      HIPCHK(hipMemcpyAsync(found, d_found, sizeof(found), hipMemcpyDeviceToHost, c->stream));
      HIPCHK(hipMemcpyAsync(wide.data(), d_wide, nt * sizeof(u32), hipMemcpyDeviceToHost, c->stream));
      HIPCHK(hipMemcpyAsync(t->tasks.data(), t->d_tasks, nt * sizeof(SegTask), hipMemcpyDeviceToHost, c->stream));
      HIPCHK(hipStreamSynchronize(c->stream));
      // merge the tasks that must not speculate (k_cutpoints) into their predecessors: the predecessor walks on to
      // the merged task's end.  (The head of a block is never merged: pout = 0.)
      size_t merged = 0;
      if (merge_on) {
        std::vector<SegTask> kept;
        kept.reserve(nt);
        std::vector<u32> off(nb + 1, 0);
        for (size_t b = 0; b < nb; ++b) {
          for (u32 k = t->task_off[b]; k < t->task_off[b + 1]; ++k) {
            if (k > t->task_off[b] && wide[k]) {
              kept.back().pend = t->tasks[k].pend;
              ++merged;
            } else {
              kept.push_back(t->tasks[k]);
            }
          }
          off[b + 1] = static_cast<u32>(kept.size());
        }
        t->merged_tasks = merged;
        if (merged) {
          t->tasks.swap(kept);
          t->task_off.swap(off);
          HIPCHK(hipMemcpyAsync(t->d_tasks, t->tasks.data(), t->tasks.size() * sizeof(SegTask), hipMemcpyHostToDevice, c->stream));
          HIPCHK(hipMemcpyAsync(t->d_task_off, t->task_off.data(), (nb + 1) * sizeof(u32), hipMemcpyHostToDevice, c->stream));
          HIPCHK(hipStreamSynchronize(c->stream));
        }
      }
      if (prof) {
        std::fprintf(stderr, "k_cutpoints: %u of %zu tasks start at a cut point, %.1f positions before their first owned one on average; "
                     "%zu tasks merged into their predecessors (long-run material in the warm-up stretch), %zu tasks left\n",
                     found[0], nt, found[0] ? static_cast<double>(found[1]) / found[0] : 0.0, merged, t->tasks.size());
      }
    }
  }
  {
    // k_dp5_spec's workgroups: four tasks of one block each (they share the block's weight table in
    // LDS); the workgroups that hold a head (several times the length of the other tasks) go first.  Tasks that
    // walk runs of equal bytes (k_taskkind) get workgroups of their own, listed after the others: they are run by the
    // kernel's other variant, beside the rest.
    const size_t ntk = t->tasks.size();
    std::vector<u32> kind(ntk, 0);
    if (ntk) {
      PoolScope tmp(c);
      u32* d_kind = nullptr;
      HIPCHK(tmp.AllocT(&d_kind, ntk, "d_kind"));
      TaskKindParams kp;
      kp.tasks = t->d_tasks;
      kp.blocks = t->d_blocks;
      kp.winflag = t->d_winflag;
      kp.win_off = t->d_win_off;
      kp.kind = d_kind;
      hipLaunchKernelGGL(k_taskkind, dim3(static_cast<unsigned>(ntk)), dim3(64), 0, c->stream, kp);
      KCHK(c, "k_taskkind");
      HIPCHK(hipMemcpyAsync(kind.data(), d_kind, ntk * sizeof(u32), hipMemcpyDeviceToHost, c->stream));
      HIPCHK(hipStreamSynchronize(c->stream));
    }
    std::vector<u32> wg, wg_runs;
    for (int pass = 0; pass < 2; ++pass) {
      for (size_t b = 0; b < nb; ++b) {
        const u32 a0 = t->task_off[b], a1 = t->task_off[b + 1];
        // pass 0: the group that holds the block's head; pass 1: the others.  Inside a block the tasks of either kind
        // are grouped four at a time in order.
        std::vector<u32> grp[2];
        for (u32 k = a0; k < a1; ++k) grp[kind[k] ? 1 : 0].push_back(k);
        for (int kd = 0; kd < 2; ++kd) {
          std::vector<u32>& dst = kd ? wg_runs : wg;
          for (size_t i = 0; i < grp[kd].size(); i += D5_WG) {
            const bool has_head = i == 0 && !grp[kd].empty() && grp[kd][0] == a0;
            if ((pass == 0) != has_head) continue;
            for (u32 w = 0; w < D5_WG; ++w) dst.push_back(i + w < grp[kd].size() ? grp[kd][i + w] : SEG_NONE);
          }
        }
      }
    }
    t->n_wg = static_cast<u32>(wg.size() / D5_WG);
    t->n_wg_runs = static_cast<u32>(wg_runs.size() / D5_WG);
    // the cooperative kernel's list (zmx_dp6.h): a workgroup per run task, the longest first — a squeeze run waits for
    // its longest task, and a giant started behind a queue of small ones ends that much later
    {
      std::vector<u32> runs;
      for (size_t k = 0; k < ntk; ++k) if (kind[k]) runs.push_back(static_cast<u32>(k));
      auto walked = [&](u32 k) {
        const SegTask& T = t->tasks[k];
        const u32 Bk = t->bsize[T.block];
        return (T.pend < Bk ? T.pend : Bk) - T.q;
      };
      std::stable_sort(runs.begin(), runs.end(), [&](u32 a, u32 b2) { return walked(a) > walked(b2); });
      t->n_run_list = static_cast<u32>(runs.size());
      HIPCHK(PoolAlloc(c, &t->d_task_kind, ntk + 4));
      HIPCHK(PoolAlloc(c, &t->d_run_list, runs.size() + 4));
      if (ntk) HIPCHK(hipMemcpyAsync(t->d_task_kind, kind.data(), ntk * sizeof(u32), hipMemcpyHostToDevice, c->stream));
      if (!runs.empty()) HIPCHK(hipMemcpyAsync(t->d_run_list, runs.data(), runs.size() * sizeof(u32), hipMemcpyHostToDevice, c->stream));
      HIPCHK(hipStreamSynchronize(c->stream));   // (locals)
    }
    wg.insert(wg.end(), wg_runs.begin(), wg_runs.end());
    HIPCHK(PoolAlloc(c, &t->d_wg_tasks, wg.size() + 4));
    if (!wg.empty()) HIPCHK(hipMemcpyAsync(t->d_wg_tasks, wg.data(), wg.size() * sizeof(u32), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));   // `wg` is a local
  }
  }  // with_dp
  // trace segments (zmx_trace.h)
  t->seg_off.assign(nb + 1, 0);
  for (size_t b = 0; b < nb; ++b) t->seg_off[b + 1] = t->seg_off[b] + (t->bsize[b] + TS_SEG - 1) / TS_SEG;
  HIPCHK(PoolAlloc(c, &t->d_seg_off, nb + 1));
  HIPCHK(PoolAlloc(c, &t->d_extab, static_cast<size_t>(t->seg_off[nb]) * GS_STATES));   // shared by the trace (TS_ENT per segment)
  HIPCHK(PoolAlloc(c, &t->d_seginfo, t->seg_off[nb]));
  HIPCHK(hipMemcpyAsync(t->d_seg_off, t->seg_off.data(), (nb + 1) * sizeof(u32), hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return 0;
}

int zmx_tables_build(zmx_ctx* c, const zmx_block* blocks, size_t nblocks, zmx_tables** out) {
  return zmx_tables_build_from(c, nullptr, blocks, nblocks, out);
}

int zmx_tables_build_matches(zmx_ctx* c, const zmx_block* blocks, size_t nblocks, zmx_tables** out) {
  DeviceGuard dev_guard(c->device);
  HIPCHK(dev_guard.err);
  zmx_tables* t = new zmx_tables();
  g_last_oom = false;
  const int rc = BuildTables(c, blocks, nblocks, t, nullptr, false);
  if (rc) {
    zmx_tables_free(c, t);
    // out of device memory (other contexts of the device hold theirs): the caller may come back with fewer blocks,
    // as for a batch beyond the code budget
    return rc == -1 && g_last_oom && nblocks > 1 ? kTooLarge : rc;
  }
  *out = t;
  return 0;
}

int zmx_tables_build_from(zmx_ctx* c, zmx_tables* parent, const zmx_block* blocks, size_t nblocks, zmx_tables** out) {
  if (parent && parent->trimmed) return FailMsg("zmx_tables_build_from: these tables were trimmed to their stores (zmx_tables_trim)");
  DeviceGuard dev_guard(c->device);
  HIPCHK(dev_guard.err);
  zmx_tables* t = new zmx_tables();
  g_last_oom = false;
  const int rc = BuildTables(c, blocks, nblocks, t, parent);
  if (rc) {
    zmx_tables_free(c, t);
    return rc == -1 && g_last_oom && nblocks > 1 ? kTooLarge : rc;
  }
  *out = t;
  return 0;
}

int zmx_lz77_greedy(zmx_ctx* c, zmx_tables* t, int slot, uint32_t* nsym, uint32_t* hist) {
  if (t && t->trimmed) return FailMsg("zmx_lz77_greedy: these tables were trimmed to their stores (zmx_tables_trim)");
  if (t->nb == 0) return 0;
  if (slot != 0 && slot != 1) return FailMsg("zmx_lz77_greedy: slot must be 0 or 1");
  DeviceGuard dev_guard(c->device);
  HIPCHK(dev_guard.err);
  GreedySegParams gp;
  gp.blocks = t->d_blocks;
  gp.seg_off = t->d_seg_off;
  gp.nb = static_cast<u32>(t->nb);
  gp.recs = t->d_recs;
  gp.store = t->d_store[slot];
  gp.hist_out = t->d_hist;
  gp.nsym_out = t->d_nsym;
  gp.extab = t->d_extab;
  gp.seginfo = t->d_seginfo;
  const unsigned nseg = t->seg_off[t->nb];
  if (nseg) hipLaunchKernelGGL(k_greedy_exits, dim3(nseg), dim3(GS_THREADS), 0, c->stream, gp);
  KCHK(c, "k_greedy_exits");
  hipLaunchKernelGGL(k_greedy_link, dim3(static_cast<unsigned>(t->nb)), dim3(64), 0, c->stream, gp);
  KCHK(c, "k_greedy_link");
  if (nseg) hipLaunchKernelGGL(k_greedy_emit, dim3(nseg), dim3(64), 0, c->stream, gp);
  KCHK(c, "k_greedy_emit");
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(nsym, t->d_nsym, t->nb * sizeof(u32), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipMemcpyAsync(hist, t->d_hist, t->nb * ZMX_HIST * sizeof(u32), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  for (size_t b = 0; b < t->nb; ++b) t->store_begin[slot][b] = 0;
  t->h_hist.assign(hist, hist + t->nb * ZMX_HIST);
  t->have_hist = true;
  return 0;
}

// What the acceptance test of the chain's tasks (zmx_dp4.h) needs to know about a run's cost model:
// an upper bound of every edge weight, the binades in which a weight can tie in the float rounding,
// and a first guess of the block's cost.  The weights are the 256 literal costs and
// ((lbits + dbits) + ll[lsym]) + d[dsym] for the 29 x 30 symbol pairs (squeeze.c:155).
static void RunInfo(const double* cost, const u32* hist, u32 B, float* wmax_out, u32* tiemask_out, float* est_out,
                    double mincost = 0.0, bool* below_mincost = nullptr) {
  const double* ll = cost;
  const double* d = cost + ZMX_NUM_LL;
  auto lbits = [](int s) { return s < 265 || s == 285 ? 0 : (s - 261) / 4; };
  auto dbits = [](int s) { return s < 4 ? 0 : s / 2 - 1; };
  double w[256 + 29 * 30];
  int n = 0;
  for (int i = 0; i < 256; ++i) w[n++] = ll[i];
  for (int ls = 257; ls <= 285; ++ls)
    for (int ds = 0; ds < 30; ++ds) w[n++] = (static_cast<double>(lbits(ls) + dbits(ds)) + ll[ls]) + d[ds];
  double wmax = 0;
  for (int i = 0; i < n; ++i) wmax = std::max(wmax, w[i]);
  // a match weight below mincost (possible only through rounding in the cost model): k_wtab finds the same ones with
  // the same arithmetic, and only then do k_badscan and its bitmap have anything to do (zmx_squeeze_run)
  if (below_mincost) {
    bool any = false;
    for (int i = 256; i < n; ++i) any |= w[i] < mincost;
    *below_mincost = any;
  }
  // dbl(w + c) = c + RNE(w / 2^(e-52)) 2^(e-52) for a float c of binade e; the float rounding of that
  // sum ties iff the remainder modulo the float ulp 2^(e-23) is exactly half of it, i.e. iff
  // r = RNE(w 2^(52-e)) has r mod 2^29 = 2^28.  Integer arithmetic on the mantissa: w = m 2^x.
  u32 mask = 0;
  for (int i = 0; i < n; ++i) {
    if (!(w[i] > 0)) continue;                       // (0 shifts nothing)
    int x;
    const double fr = std::frexp(w[i], &x);          // w = fr 2^x, 0.5 <= fr < 1
    const uint64_t m = static_cast<uint64_t>(std::ldexp(fr, 53));   // 53-bit integer, exact
    x -= 53;                                         // w = m 2^x
    {
      // r mod 2^29 = 2^28 needs 28 equal bits in a row somewhere in m — zeros, or ones that a rounding carry
      // turns into zeros, or nothing but zeros below the leading bit (shifts past the mantissa's end).  Entropy
      // costs are log2 values with random mantissas: this test, not the loop over the binades, is what they cost
      // (the loop was 0.25 ms per block and run, a quarter of what the host spent between two runs).
      auto run27 = [](uint64_t y) {                  // 27 ones in a row in y?
        uint64_t a = y & (y >> 1);
        a &= a >> 2; a &= a >> 4; a &= a >> 8;       // 16 in a row
        return (a & (a >> 11)) != 0;
      };
      const uint64_t mask53 = (1ull << 53) - 1;
      if (!run27(~m & mask53) && !run27(m) && (m & ((1ull << 26) - 1)) != 0) continue;
    }
    for (int e = 4; e < 32; ++e) {
      const int sh = -(x + 52 - e);                  // r = RNE(m / 2^sh)
      uint64_t r;
      if (sh <= 0) {
        if (-sh >= 29) continue;                     // r is a multiple of 2^29
        r = m << -sh;
      } else if (sh >= 64) {
        continue;                                    // r = 0
      } else {
        const uint64_t q = m >> sh, rem = m & ((1ull << sh) - 1), half = 1ull << (sh - 1);
        r = q + ((rem > half || (rem == half && (q & 1))) ? 1 : 0);
      }
      if ((r & 0x1fffffffull) == 0x10000000ull) mask |= 1u << e;
    }
  }
  *wmax_out = static_cast<float>(wmax) + 1.0f;
  *tiemask_out = mask;
  // no parse to go by (the fixed-tree re-parse, deflate.c:770-781, which is tried on blocks that
  // compress badly): close to 8 bits per byte; the run refines the tasks' levels itself
  double est = 8.0 * B;
  if (hist) {
    est = 0;
    for (int i = 0; i < ZMX_NUM_LL; ++i) est += hist[i] * (ll[i] + (i > 256 ? lbits(i) : 0));
    for (int i = 0; i < 30; ++i) est += hist[ZMX_NUM_LL + i] * (d[i] + dbits(i));
  }
  *est_out = static_cast<float>(est);
}

// Test hook (tests/test_cpu_abi.py): the acceptance facts of one cost model, no device involved.
__attribute__((visibility("default"))) void zmx_internal_run_info(const double* cost320, float* wmax, uint32_t* tiemask) {
  float est;
  RunInfo(cost320, nullptr, 0, wmax, tiemask, &est);
}

int zmx_squeeze_run(zmx_ctx* c, zmx_tables* t, const double* cost, const double* mincost, const int32_t* slot,
                    uint32_t* nsym, uint32_t* hist) {
  if (t && t->trimmed) return FailMsg("zmx_squeeze_run: these tables were trimmed to their stores (zmx_tables_trim)");
  if (t->nb == 0) return 0;
  if (t->matches_only) return FailMsg("zmx_squeeze_run: these tables hold matches only (zmx_tables_build_matches)");
  DeviceGuard dev_guard(c->device);
  HIPCHK(dev_guard.err);
  for (size_t b = 0; b < t->nb; ++b) {
    if (slot[b] != 0 && slot[b] != 1) return FailMsg("zmx_squeeze_run: slot must be 0 or 1");
  }
  const size_t nb = t->nb;
  bool any_below_mincost = false;
  // the run's input in the pinned mirror: costs, mincosts, slots, and per block what the chain's acceptance test has
  // to know about this cost model (RunInfo)
  {
    double* h_cost = reinterpret_cast<double*>(t->h_runin);
    double* h_min = h_cost + nb * ZMX_HIST;
    float* wmax = reinterpret_cast<float*>(h_min + nb);
    u32* tiemask = reinterpret_cast<u32*>(wmax + nb);
    float* est = wmax + 2 * nb;
    int* h_slot = reinterpret_cast<int*>(wmax + 3 * nb);
    std::memcpy(h_cost, cost, nb * ZMX_HIST * sizeof(double));
    std::memcpy(h_min, mincost, nb * sizeof(double));
    std::memcpy(h_slot, slot, nb * sizeof(int));
    const u32* hh = t->have_hist ? t->h_hist.data() : nullptr;
    std::vector<char> below(nb, 0);
    zamd::ParallelFor(nb, [&](size_t b) {
      bool bad = false;
      RunInfo(cost + b * ZMX_HIST, hh ? hh + b * ZMX_HIST : nullptr, t->bsize[b], &wmax[b], &tiemask[b], &est[b], mincost[b], &bad);
      below[b] = bad ? 1 : 0;
    });
    for (size_t b = 0; b < nb; ++b) any_below_mincost |= below[b] != 0;
  }
  HIPCHK(hipMemcpyAsync(t->d_runin, t->h_runin, t->runin_bytes, hipMemcpyHostToDevice, c->stream));
  static const bool want_prof = std::getenv("ZOPFLI_AMD_PROF") != nullptr;
  if (want_prof && !t->d_prof) HIPCHK(PoolAlloc(c, &t->d_prof, nb * ZMX_PROF_N));
  if (t->d_prof) HIPCHK(hipMemsetAsync(t->d_prof, 0, nb * ZMX_PROF_N * sizeof(u64), c->stream));
  WtabParams wp;
  wp.cost = t->d_cost;
  wp.mincost = t->d_mincost;
  wp.wtab = t->d_wtab;
  wp.badcodes = t->d_badcodes;
  wp.stats = t->d_segstats;
  BadScanParams bp;
  bp.blocks = t->d_blocks;
  bp.tile_off = t->d_tile_off;
  bp.nb_total = static_cast<u32>(nb);
  bp.dph = t->d_dph;
  bp.codes = t->d_codes;
  bp.code_base = t->d_code_base;
  bp.badcodes = t->d_badcodes;
  bp.badpos = t->d_badpos;
  Dp4Params cp;
  cp.blocks = t->d_blocks;
  cp.block0 = 0;
  cp.task0 = 0;
  cp.dph = t->d_dph;
  cp.cost = t->d_cost;
  cp.mincost = t->d_mincost;
  cp.codes = t->d_codes;
  cp.code_base = t->d_code_base;
  cp.block_edges = t->d_block_edges;
  cp.wtab = t->d_wtab;
  cp.la = t->d_la;
  cp.prof = t->d_prof;
  cp.badpos = t->d_badpos;
  cp.tasks = t->d_tasks;
  cp.task_off = t->d_task_off;
  cp.lvl = t->d_lvl;
  cp.est_bits = t->squeeze_runs == 0 ? t->d_runinfo + 2 * nb : nullptr;
  cp.entry = t->d_entry;
  cp.exit = t->d_exit;
  static const bool mid_on = EnvU32("ZOPFLI_AMD_SEG_MID", 1, 0, 1) != 0;   // 0: no mid snapshots (tasks that leave their binade are re-run whole)
  cp.mid = mid_on ? t->d_mid : nullptr;
  cp.chk = t->d_chk;
  cp.over = t->d_over;
  cp.wmax = t->d_runinfo;
  cp.tiemask = reinterpret_cast<const u32*>(t->d_runinfo + nb);
  cp.stats = t->d_segstats;
  static const float level_scale = [] { const char* e = std::getenv("ZOPFLI_AMD_SEG_SCALE"); return e ? static_cast<float>(std::atof(e)) : 1.0f; }();
  cp.level_scale = level_scale;
  cp.wg_tasks = t->d_wg_tasks;
  static const int seg_debug = [] { const char* e = std::getenv("ZOPFLI_AMD_SEG_DEBUG"); return e ? std::atoi(e) : 0; }();
  cp.debug = seg_debug;
  // (ZOPFLI_AMD_FIX_LEAN: 0 = every serial re-run by the lean one-wave job, large = none, unset = by the task's windows; zmx_dp5.h)
  static const int fix_lean = [] { const char* e = std::getenv("ZOPFLI_AMD_FIX_LEAN"); return e ? std::atoi(e) : -1; }();
  cp.fix_lean_min = fix_lean;
  static const int int_path = [] { const char* e = std::getenv("ZOPFLI_AMD_INT_PATH"); return e ? std::atoi(e) : 1; }();
  cp.int_path = int_path;
  static const int chain_fast = [] { const char* e = std::getenv("ZOPFLI_AMD_SHORTCUT_CHAIN"); return e ? std::atoi(e) : 1; }();
  cp.chain_fast = chain_fast;
  cp.redo_count = t->d_redo;
  cp.redo_wg = t->d_redo + 4;
  cp.redo_pass = 0;
  // (ZOPFLI_AMD_COOP=1: run tasks by k_dp6_spec, four waves a task (zmx_dp6.h) — built and measured in round 5, NOT the
  //  default: 52.6 ms of chain per run on class Z against 45.9 with one wave a task)
#ifdef ZMX_EXPERIMENTS
  static const int coop = [] { const char* e = std::getenv("ZOPFLI_AMD_COOP"); return e ? std::atoi(e) : 0; }();
#else
  constexpr int coop = 0;     // (k_dp6_spec is in -DZMX_EXPERIMENTS builds only)
#endif
  cp.coop = coop != 0 ? 1 : 0;
  cp.kind = t->d_task_kind;
  cp.run_list = t->d_run_list;
  cp.flags = t->d_flags;
  cp.wmeta = t->d_wmeta;
  cp.winroff = t->d_winroff;
  cp.winflag = t->d_winflag;
  cp.win_off = t->d_win_off;
  // k_badscan's bitmap: all zero unless some block of this run has a match weight below mincost (RunInfo) — nearly never,
  // and then neither the memset nor the scan is launched (two of a run's ~20 stream operations; a small call is mostly
  // the gaps between them)
  const bool scan_bad = any_below_mincost;
  if (scan_bad || !t->badpos_clean) HIPCHK(hipMemsetAsync(t->d_badpos, 0, t->badpos_words * sizeof(u32), c->stream));
  t->badpos_clean = !scan_bad;
  TraceSegParams tp;
  tp.blocks = t->d_blocks;
  tp.seg_off = t->d_seg_off;
  tp.nb_total = static_cast<u32>(nb);
  tp.recs = t->d_recs;
  tp.pool = t->d_pool;
  tp.la = t->d_la;
  tp.slot = t->d_slot;
  tp.store0 = t->d_store[0];
  tp.store1 = t->d_store[1];
  tp.hist_out = t->d_hist;
  tp.nsym_out = t->d_nsym;
  tp.flags = t->d_flags;
  tp.extab = t->d_extab;
  tp.seginfo = t->d_seginfo;
  tp.block0 = 0;
  tp.seg0 = 0;
  double ksec[3] = {0, 0, 0};
  const bool timing = KernelTiming();
  const dim3 dpdim(64 * (D3_NB + 2));
  {
    const unsigned nblk = static_cast<unsigned>(nb);
    const unsigned tiles = t->tile_off[nb];
    const unsigned ntask = t->task_off[nb];
    if (timing) HIPCHK(hipEventRecord(c->ev[0], c->stream));
    // the run's weights per block, and (rarely) the positions that own an edge below mincost
    hipLaunchKernelGGL(k_wtab, dim3(nblk), dim3(256), 0, c->stream, wp);
    KCHK(c, "k_wtab");
    if (tiles && scan_bad) hipLaunchKernelGGL(k_badscan, dim3(tiles), dim3(256), 0, c->stream, bp);
    KCHK(c, "k_badscan");
    HIPCHK(hipGetLastError());
    if (timing) HIPCHK(hipEventRecord(c->ev[1], c->stream));
    // the chain: every task speculatively on all CUs (four tasks of a block per workgroup, the workgroups
    // with a head first), then the per-block walk that accepts or re-runs
    {
      const dim3 bdim(64 * D5_WG);
      // the run tasks on a second stream, beside the others: few and long (one wave may walk 100 000 positions)
      if (t->n_wg_runs) {
        HIPCHK(hipEventRecord(c->ev2[0], c->stream));
        HIPCHK(hipStreamWaitEvent(c->stream2, c->ev2[0], 0));
        Dp4Params cr = cp;
        cr.task0 = t->n_wg;
#ifdef ZMX_EXPERIMENTS
        if (cp.coop && cp.prof) hipLaunchKernelGGL((k_dp6_spec<2, true>), dim3(t->n_run_list), dim3(64 * D6_NW), 0, c->stream2, cr);
        else if (cp.coop) hipLaunchKernelGGL((k_dp6_spec<2, false>), dim3(t->n_run_list), dim3(64 * D6_NW), 0, c->stream2, cr);
        else
#endif
        if (cp.prof) hipLaunchKernelGGL((k_dp5_spec<true, 2, true>), dim3(t->n_wg_runs), bdim, 0, c->stream2, cr);
        else hipLaunchKernelGGL((k_dp5_spec<false, 2, true>), dim3(t->n_wg_runs), bdim, 0, c->stream2, cr);
        HIPCHK(hipGetLastError());
        HIPCHK(hipEventRecord(c->ev2[1], c->stream2));
      }
      if (t->n_wg) {
        if (cp.prof) hipLaunchKernelGGL((k_dp5_spec<true, 4, false>), dim3(t->n_wg), bdim, 0, c->stream, cp);
        else hipLaunchKernelGGL((k_dp5_spec<false, 4, false>), dim3(t->n_wg), bdim, 0, c->stream, cp);
        HIPCHK(hipGetLastError());
      }
      if (t->n_wg_runs) HIPCHK(hipStreamWaitEvent(c->stream, c->ev2[1], 0));
      if (GuardOn()) HIPCHK(hipStreamSynchronize(c->stream2));
      KCHK(c, "k_dp5_spec");
    }
    if (ntask > nblk) {
      hipLaunchKernelGGL(k_dpcheck, dim3(ntask), dim3(64), 0, c->stream, cp);
      KCHK(c, "k_dpcheck");
      // a second speculative run, from the level the chain of shifts implies, for the tasks that only
      // missed their level (ZOPFLI_AMD_SEG_REDO=0: leave them to the serial pass)
      // (ZOPFLI_AMD_SEG_REDO = how many such passes, default 1.  More settle more tasks before the serial pass — class Z:
      //  87 / 92 / 93 % accepted with 1 / 2 / 3 — but every pass waits for its longest task: 1 264 / 1 362 / 1 601 ms)
      static const int redo = [] { const char* e = std::getenv("ZOPFLI_AMD_SEG_REDO"); return e ? std::atoi(e) : 1; }();
      for (int pass = 0; pass < redo; ++pass) {
        // (the list's counter was zeroed by the k_dpcheck before: zmx_dp4.h)
        hipLaunchKernelGGL(k_dpscan, dim3(nblk), dim3(64), 0, c->stream, cp);
        KCHK(c, "k_dpscan");
        Dp4Params c2 = cp;
        c2.redo_pass = 1;
        c2.est_bits = nullptr;
        // (one workgroup per listed task; the workgroups beyond the list have nothing to do)
        const unsigned cap = ntask;
        // (the variant for run tasks wherever the set has any: what is run again there is mostly theirs)
#ifdef ZMX_EXPERIMENTS
        if (t->n_wg_runs && cp.coop) {
          // the listed run tasks by k_dp6_spec, the listed text tasks by the text variant (each passes over the other's)
          hipLaunchKernelGGL((k_dp6_spec<2, false>), dim3(cap), dim3(64 * D6_NW), 0, c->stream, c2);
          if (cp.prof) hipLaunchKernelGGL((k_dp5_spec<true, 4, false>), dim3(cap), dim3(64 * D5_WG), 0, c->stream, c2);
          else hipLaunchKernelGGL((k_dp5_spec<false, 4, false>), dim3(cap), dim3(64 * D5_WG), 0, c->stream, c2);
        } else
#endif
        if (t->n_wg_runs) {
          if (cp.prof) hipLaunchKernelGGL((k_dp5_spec<true, 2, true>), dim3(cap), dim3(64 * D5_WG), 0, c->stream, c2);
          else hipLaunchKernelGGL((k_dp5_spec<false, 2, true>), dim3(cap), dim3(64 * D5_WG), 0, c->stream, c2);
        } else {
          if (cp.prof) hipLaunchKernelGGL((k_dp5_spec<true, 4, false>), dim3(cap), dim3(64 * D5_WG), 0, c->stream, c2);
          else hipLaunchKernelGGL((k_dp5_spec<false, 4, false>), dim3(cap), dim3(64 * D5_WG), 0, c->stream, c2);
        }
        KCHK(c, "k_dp5_spec");
        hipLaunchKernelGGL(k_dpcheck, dim3(ntask), dim3(64), 0, c->stream, cp);
        KCHK(c, "k_dpcheck");
      }
      if (cp.prof) hipLaunchKernelGGL(k_dp4_fix<true>, dim3(nblk), dpdim, 0, c->stream, cp);
      else hipLaunchKernelGGL(k_dp4_fix<false>, dim3(nblk), dpdim, 0, c->stream, cp);
      KCHK(c, "k_dp4_fix");
    }
    HIPCHK(hipGetLastError());
    if (timing) HIPCHK(hipEventRecord(c->ev[2], c->stream));
    const unsigned nseg = t->seg_off[nb];
    if (nseg) hipLaunchKernelGGL(k_trace_exits, dim3(nseg), dim3(TS_THREADS), 0, c->stream, tp);
    KCHK(c, "k_trace_exits");
    hipLaunchKernelGGL(k_trace_link, dim3(nblk), dim3(64), 0, c->stream, tp);
    KCHK(c, "k_trace_link");
    if (nseg) hipLaunchKernelGGL(k_trace_emit, dim3(nseg), dim3(64), 0, c->stream, tp);
    KCHK(c, "k_trace_emit");
    HIPCHK(hipGetLastError());
    if (timing) HIPCHK(hipEventRecord(c->ev[3], c->stream));
  }
  // ONE host round trip per run: the results travel behind the last kernel, the host waits for the copy and reads the
  // phase times then (waiting for ev[3] first and only then asking for the copy was two).  The task statistics are
  // zeroed by the next run's k_wtab.
  u32 segstats[8];
  HIPCHK(hipMemcpyAsync(t->h_runout, t->d_runout, t->runout_bytes, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  for (int i = 0; i < 3 && timing; ++i) {
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, c->ev[i], c->ev[i + 1]));
    ksec[i] += ms * 1e-3;
  }
  {
    const u32* o_hist = reinterpret_cast<const u32*>(t->h_runout);
    const u32* o_nsym = o_hist + nb * ZMX_HIST;
    const u32* o_stats = o_nsym + nb;
    const u32* o_flags = o_stats + 8;
    std::memcpy(hist, o_hist, nb * ZMX_HIST * sizeof(u32));
    std::memcpy(nsym, o_nsym, nb * sizeof(u32));
    std::memcpy(segstats, o_stats, sizeof(segstats));
    if (o_flags[1]) {
      char buf[128];
      std::snprintf(buf, sizeof(buf), "zmx_squeeze_run: device consistency flags 0x%x", o_flags[1]);
      return FailFault(buf);
    }
  }
  t->h_hist.assign(hist, hist + nb * ZMX_HIST);
  t->have_hist = true;
  ++t->squeeze_runs;
  {
    // (thread-local: no lock)
    for (int i = 0; i < 3; ++i) g_kernel_seconds[i] += ksec[i];
    g_squeeze_launches += 1;
    for (int i = 0; i < 7; ++i) g_seg_stats[i] += segstats[i];
    g_seg_stats[7] += static_cast<double>(t->total_b);
  }
  for (size_t b = 0; b < nb; ++b) t->store_begin[slot[b]][b] = t->bsize[b] - nsym[b];
  if (t->d_prof) {
    std::vector<u64> pr(nb * ZMX_PROF_N);
    HIPCHK(hipMemcpy(pr.data(), t->d_prof, pr.size() * sizeof(u64), hipMemcpyDeviceToHost));
    double a[ZMX_PROF_N] = {};
    for (size_t b = 0; b < nb; ++b)
      for (unsigned k = 0; k < ZMX_PROF_N; ++k) a[k] += static_cast<double>(pr[b * ZMX_PROF_N + k]);
    if (cp.coop && t->n_run_list) {
      // k_dp6_spec's first pass (zmx_dp6.h): wave 0's cycles by activity, summed over the run tasks (the text tasks' counters
      // of k_dp5_spec share the slots: read this line on long-run data only)
      double mxj = 0;
      for (size_t b = 0; b < nb; ++b) mxj = std::max(mxj, static_cast<double>(pr[b * ZMX_PROF_N + 16]));
      std::fprintf(stderr, "coop prof (k_dp6_spec, first pass, wave 0): job cycles %.3g (longest job %.3g): token+flow %.3g, class 1 %.3g, class 2 %.3g, "
                   "headers %.3g, other stretches %.3g (%.0f positions, %.0f each), run stretches %.3g (%.0f positions, %.0f each), general step %.3g (%.0f, %.0f each), "
                   "shortcuts %.3g (%.0f, %.0f each), waiting for wave 1 at the rotation %.3g (%.0f windows, %.0f each); wave 1: %.3g of %.3g cycles inside get()\n",
                   a[9], mxj, a[0], a[1], a[2], a[3], a[4], a[10], a[4] / (a[10] + 1e-9), a[5], a[11], a[5] / (a[11] + 1e-9), a[6], a[12], a[6] / (a[12] + 1e-9),
                   a[7], a[13], a[7] / (a[13] + 1e-9), a[8], a[14], a[8] / (a[14] + 1e-9), a[15], a[17]);
    }
    std::fprintf(stderr, "squeeze prof: edges %.2f ms chain %.2f ms trace %.2f ms; chain wave busy %.1f cycles/position, "
                 "%.0f steps, fast %.1f%% of %.0f positions walked (%zu in the blocks); tasks %u accepted %u re-run "
                 "state %u values %u level %u tie %u (%u positions, %u by the lean job)\n",
                 ksec[0] * 1e3, ksec[1] * 1e3, ksec[2] * 1e3, a[1] / a[4], a[0],
                 100.0 * a[2] / (a[2] + a[3] + 1e-9), a[4], t->total_b, segstats[0], segstats[1], segstats[2],
                 segstats[6], segstats[3], segstats[4], segstats[5], segstats[7]);
    {
      double mx = 0, hd = 0;
      for (size_t b = 0; b < nb; ++b) { mx = std::max(mx, static_cast<double>(pr[b * ZMX_PROF_N + 7])); hd = std::max(hd, static_cast<double>(pr[b * ZMX_PROF_N + 8])); }
      std::fprintf(stderr, "  k_dp5_spec: longest task %.0f cycles, longest head task %.0f cycles; %.1f%% of the positions walked by the integer step (cycles per position there: fetch issue %.1f, fetch wait %.1f, gather + chain %.1f)\n", mx, hd, 100.0 * a[10] / (a[4] + 1e-9), a[11] / (a[10] + 1e-9), a[12] / (a[10] + 1e-9), a[13] / (a[10] + 1e-9));
    }
    std::fprintf(stderr, "  k_dp5_spec integer windows, cycles per position: class decision %.1f, waiting for the prefetched record and codes %.1f, prefetch issue %.1f\n",
                 a[20] / (a[10] + 1e-9), a[21] / (a[10] + 1e-9), a[22] / (a[10] + 1e-9));
    std::fprintf(stderr, "  k_dp5_spec windows: integer %.1f%% of positions at %.0f cycles each, class 1 in doubles %.1f%% at %.0f, class 2 %.1f%% at %.0f, generic %.1f%% at %.0f\n",
                 100.0 * a[28] / (a[4] + 1e-9), a[24] / (a[28] + 1e-9), 100.0 * a[29] / (a[4] + 1e-9), a[25] / (a[29] + 1e-9),
                 100.0 * a[30] / (a[4] + 1e-9), a[26] / (a[30] + 1e-9), 100.0 * a[31] / (a[4] + 1e-9), a[27] / (a[31] + 1e-9));
    std::fprintf(stderr, "  k_dp5_spec generic windows, cycles each: shortcut %.0f (%.0f of them), run row integer %.0f (%.0f), run row doubles %.0f (%.0f), other row %.0f (%.0f), window header %.0f (%.0f)\n",
                 a[32] / (a[33] + 1e-9), a[33], a[34] / (a[35] + 1e-9), a[35], a[36] / (a[37] + 1e-9), a[37], a[38] / (a[39] + 1e-9), a[39], a[40] / (a[41] + 1e-9), a[41]);
    std::fprintf(stderr, "  k_dp5_spec positions: run interior %.0f in unrolled windows, %.0f in loops with room, %.0f with checks; general step: %.0f run rows, %.0f other rows; class 2: %.0f rows that reach register 2\n",
                 a[42], a[43], a[44], a[45], a[46], a[47]);
    {
      const double gen = a[51] - a[49] - a[50] - a[52];     // (the loop over a window's positions, less its stretches)
      std::fprintf(stderr, "  k_dp5_spec generic windows, cycles: headers %.3g, run stretches: whole windows %.3g (%.0f per position), others %.3g (%.0f); stretches of other rows %.3g (%.0f); the general step %.3g (%.0f per position incl. shortcuts)\n",
                   a[48], a[49], a[49] / (a[42] + 1e-9), a[50], a[50] / (a[43] + a[44] + 1e-9), a[52], a[52] / (a[53] + 1e-9), gen, gen / (a[45] + a[46] + a[33] + 1e-9));
    }
    const char* nm[5] = {"32", "16", "8", "8 (two registers)", "generic"};
    for (int i = 0; i < 5; ++i)
      std::fprintf(stderr, "  path %-18s %5.1f%% of positions, %6.1f cycles/position\n", nm[i], 100.0 * a[6 + 2 * i] / a[4],
                   a[5 + 2 * i] / (a[6 + 2 * i] + 1e-9));
    for (int w = 0; w < 2; ++w)
      std::fprintf(stderr, "  producer wave %d, cycles/step: walk %.0f ring %.0f tiles %.0f barrier %.0f\n", w + 1,
                   a[16 + 8 * w] / a[0], a[17 + 8 * w] / a[0], a[18 + 8 * w] / a[0], a[19 + 8 * w] / a[0]);
  }
  return 0;
}

int zmx_store_download(zmx_ctx* c, zmx_tables* t, size_t block, int slot, uint16_t* litlens, uint16_t* dists,
                       size_t nsym) {
  if (block >= t->nb || (slot != 0 && slot != 1)) return FailMsg("zmx_store_download: bad block or slot");
  if (nsym == 0) return 0;
  const u32 begin = t->store_begin[slot][block];
  if (begin + nsym > t->bsize[block]) return FailMsg("zmx_store_download: nsym exceeds the store");
  DeviceGuard dev_guard(c->device);
  HIPCHK(dev_guard.err);
  std::vector<u32> tmp(nsym);
  HIPCHK(hipMemcpyAsync(tmp.data(), t->d_store[slot] + t->blocks[block].pos_off + begin, nsym * sizeof(u32),
                        hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  for (size_t i = 0; i < nsym; ++i) {
    litlens[i] = static_cast<uint16_t>(tmp[i] & 0xffffu);
    dists[i] = static_cast<uint16_t>(tmp[i] >> 16);
  }
  return 0;
}

int zmx_store_download_batch(zmx_ctx* c, zmx_tables* t, size_t n, const size_t* block, const int32_t* slot,
                             const size_t* nsym, uint16_t* const* litlens, uint16_t* const* dists) {
  std::vector<size_t> off(n + 1, 0);
  for (size_t i = 0; i < n; ++i) {
    if (block[i] >= t->nb || (slot[i] != 0 && slot[i] != 1)) return FailMsg("zmx_store_download_batch: bad block or slot");
    if (t->store_begin[slot[i]][block[i]] + nsym[i] > t->bsize[block[i]])
      return FailMsg("zmx_store_download_batch: nsym exceeds the store");
    off[i + 1] = off[i] + nsym[i];
  }
  if (off[n] == 0) return 0;
  DeviceGuard dev_guard(c->device);
  HIPCHK(dev_guard.err);
  if (off[n] > c->stage_cap) {
    if (c->h_stage) HIPCHK(hipHostFree(c->h_stage));
    c->h_stage = nullptr;
    c->stage_cap = 0;
    const size_t cap = off[n] + off[n] / 4;
    HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&c->h_stage), cap * sizeof(u32), hipHostMallocDefault));
    c->stage_cap = cap;
  }
  // every store in one go into pinned memory, one synchronisation, then the split into the
  // reference's two u16 arrays on the host workers
  for (size_t i = 0; i < n; ++i) {
    if (nsym[i] == 0) continue;
    const u32* src = t->d_store[slot[i]] + t->blocks[block[i]].pos_off + t->store_begin[slot[i]][block[i]];
    HIPCHK(hipMemcpyAsync(c->h_stage + off[i], src, nsym[i] * sizeof(u32), hipMemcpyDeviceToHost, c->stream));
  }
  HIPCHK(hipStreamSynchronize(c->stream));
  const u32* stage = c->h_stage;
  zamd::ParallelFor(n, [&](size_t i) {
    const u32* s = stage + off[i];
    uint16_t* ll = litlens[i];
    uint16_t* dd = dists[i];
    for (size_t k = 0; k < nsym[i]; ++k) {
      ll[k] = static_cast<uint16_t>(s[k] & 0xffffu);
      dd[k] = static_cast<uint16_t>(s[k] >> 16);
    }
  });
  return 0;
}

int zmx_verify_stores(zmx_ctx* c, zmx_tables* t, size_t n, const size_t* block, const int32_t* slot, const size_t* nsym) {
  if (t && t->trimmed) return FailMsg("zmx_verify_stores: these tables were trimmed to their stores (zmx_tables_trim)");
  if (n == 0) return 0;
  std::vector<VerifyJob> vj(n);
  for (size_t i = 0; i < n; ++i) {
    if (block[i] >= t->nb || (slot[i] != 0 && slot[i] != 1)) return FailMsg("zmx_verify_stores: bad block or slot");
    if (t->store_begin[slot[i]][block[i]] + nsym[i] > t->bsize[block[i]]) return FailMsg("zmx_verify_stores: nsym exceeds the store");
    vj[i].sym_off = t->blocks[block[i]].pos_off + t->store_begin[slot[i]][block[i]];
    vj[i].instart = t->blocks[block[i]].instart;
    vj[i].inend = t->blocks[block[i]].inend;
    vj[i].nsym = static_cast<u32>(nsym[i]);
    vj[i].slot = static_cast<u32>(slot[i]);
  }
  DeviceGuard dev_guard(c->device);
  HIPCHK(dev_guard.err);
  PoolScope tmp(c);
  VerifyJob* d_jobs = nullptr;
  u32* d_bad = nullptr;
  HIPCHK(tmp.AllocT(&d_jobs, n, "d_jobs"));
  HIPCHK(tmp.AllocT(&d_bad, 2 * n, "d_bad"));
  HIPCHK(hipMemcpyAsync(d_jobs, vj.data(), n * sizeof(VerifyJob), hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemsetAsync(d_bad, 0, 2 * n * sizeof(u32), c->stream));
  VerifyParams P;
  P.jobs = d_jobs;
  P.in = c->d_in;
  P.store[0] = t->d_store[0];
  P.store[1] = t->d_store[1];
  P.bad = d_bad;
  hipLaunchKernelGGL(k_verify, dim3(static_cast<unsigned>(n)), dim3(256), 0, c->stream, P);
  KCHK(c, "k_verify");
  HIPCHK(hipGetLastError());
  std::vector<u32> bad(2 * n);
  HIPCHK(hipMemcpyAsync(bad.data(), d_bad, 2 * n * sizeof(u32), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  for (size_t i = 0; i < n; ++i) {
    if (bad[2 * i] == 0) continue;
    static const char* why[4] = {"", "length or distance out of range", "the bytes it stands for are not the input's", "the symbols do not add up to the block"};
    char msg[200];
    std::snprintf(msg, sizeof(msg), "zmx_verify_stores: block %zu, symbol %u: %s", block[i], (bad[2 * i] >> 2) - 1, why[bad[2 * i] & 3]);
    g_err = msg;
    return -1;
  }
  return 0;
}

int zmx_checksum(zmx_ctx* c, int kind, size_t begin, size_t end, uint32_t* value) {
  if (kind != ZMX_CRC32 && kind != ZMX_ADLER32) return FailMsg("zmx_checksum: unknown kind");
  if (begin > end || end > c->insize) return FailMsg("zmx_checksum: range outside the resident input");
  const size_t n = end - begin;
  const size_t npieces = (n + zamd::kChecksumPieceBytes - 1) / zamd::kChecksumPieceBytes;
  std::vector<zamd::ChecksumPiece> pieces(npieces);
  if (npieces) {
    DeviceGuard dev_guard(c->device);
    HIPCHK(dev_guard.err);
    PoolScope tmp(c);
    u32* d_out = nullptr;
    HIPCHK(tmp.AllocT(&d_out, 3 * npieces, "d_out"));
    ChecksumParams P;
    P.in = c->d_in;
    P.begin = static_cast<long long>(begin);
    P.end = static_cast<long long>(end);
    P.out = d_out;
    zamd::ChecksumTreePowers(P.xpow);
    hipLaunchKernelGGL(k_checksum, dim3(static_cast<unsigned>(npieces)), dim3(256), 0, c->stream, P);
    KCHK(c, "k_checksum");
    HIPCHK(hipGetLastError());
    static_assert(sizeof(zamd::ChecksumPiece) == 12, "three words per piece");
    HIPCHK(hipMemcpyAsync(pieces.data(), d_out, npieces * sizeof(zamd::ChecksumPiece), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
  }
  *value = kind == ZMX_CRC32 ? zamd::FinishCrc32(pieces.data(), npieces, n) : zamd::FinishAdler32(pieces.data(), npieces, n);
  return 0;
}

int zmx_encode_blocks(zmx_ctx* c, zmx_tables* t, size_t njobs, const zmx_enc_job* jobs, const uint32_t* codes,
                      unsigned char* const* out) {
  if (njobs == 0) return 0;
  std::vector<EncJob> ej(njobs);
  std::vector<u32> tile_job;
  std::vector<size_t> out_off(njobs + 1, 0);     // in the device / staging buffer, 8-byte aligned
  for (size_t j = 0; j < njobs; ++j) {
    const zmx_enc_job& q = jobs[j];
    if (q.block >= t->nb || (q.slot != 0 && q.slot != 1)) return FailMsg("zmx_encode_blocks: bad block or slot");
    if (t->store_begin[q.slot][q.block] + q.nsym > t->bsize[q.block]) return FailMsg("zmx_encode_blocks: nsym exceeds the store");
    out_off[j + 1] = out_off[j] + (((q.bit_start + q.nbits + 7) / 8 + 8 + 7) & ~static_cast<size_t>(7));
    EncJob& e = ej[j];
    e.sym_off = t->blocks[q.block].pos_off + t->store_begin[q.slot][q.block];
    e.out_word = out_off[j] / 4;
    e.nbits = q.nbits;
    e.nsym = q.nsym;
    e.slot = static_cast<u32>(q.slot);
    e.bit_start = q.bit_start;
    e.tile0 = static_cast<u32>(tile_job.size());
    e.code = static_cast<u32>(j);
    e.pad = 0;
    const u32 ntiles = q.nsym / ENC_TILE + 1;
    for (u32 k = 0; k < ntiles; ++k) tile_job.push_back(static_cast<u32>(j));
  }
  DeviceGuard dev_guard(c->device);
  HIPCHK(dev_guard.err);
  const size_t ntile = tile_job.size(), out_words = out_off[njobs] / 4;
  EncJob* d_jobs = nullptr;
  u32 *d_tile_job = nullptr, *d_codes = nullptr, *d_tile_bits = nullptr, *d_out = nullptr, *d_flag = nullptr;
  u64* d_tile_off = nullptr;
  PoolScope tmp(c);
  HIPCHK(tmp.AllocT(&d_jobs, njobs, "d_jobs"));
  HIPCHK(tmp.AllocT(&d_tile_job, ntile, "d_tile_job"));
  HIPCHK(tmp.AllocT(&d_codes, njobs * 320, "d_codes"));
  HIPCHK(tmp.AllocT(&d_tile_bits, ntile, "d_tile_bits"));
  HIPCHK(tmp.AllocT(&d_tile_off, ntile, "d_tile_off"));
  HIPCHK(tmp.AllocT(&d_out, out_words + 2, "d_out"));
  HIPCHK(tmp.AllocT(&d_flag, 4, "d_flag"));
  HIPCHK(hipMemcpyAsync(d_jobs, ej.data(), njobs * sizeof(EncJob), hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(d_tile_job, tile_job.data(), ntile * sizeof(u32), hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(d_codes, codes, njobs * 320 * sizeof(u32), hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemsetAsync(d_out, 0, (out_words + 2) * sizeof(u32), c->stream));
  HIPCHK(hipMemsetAsync(d_flag, 0, 4 * sizeof(u32), c->stream));
  EncParams P;
  P.jobs = d_jobs;
  P.tile_job = d_tile_job;
  P.codes = d_codes;
  P.store[0] = t->d_store[0];
  P.store[1] = t->d_store[1];
  P.tile_bits = d_tile_bits;
  P.tile_off = d_tile_off;
  P.out = d_out;
  P.flags = d_flag;
  P.njobs = static_cast<u32>(njobs);
  hipLaunchKernelGGL(k_enc_len, dim3(static_cast<unsigned>(ntile)), dim3(ENC_THREADS), 0, c->stream, P);
  KCHK(c, "k_enc_len");
  hipLaunchKernelGGL(k_enc_scan, dim3(static_cast<unsigned>(njobs)), dim3(64), 0, c->stream, P);
  KCHK(c, "k_enc_scan");
  hipLaunchKernelGGL(k_enc_emit, dim3(static_cast<unsigned>(ntile)), dim3(ENC_THREADS), 0, c->stream, P);
  KCHK(c, "k_enc_emit");
  HIPCHK(hipGetLastError());
  // down through the pinned staging buffer, then into the caller's memory on the host workers
  if (out_words + 1 > c->stage_cap) {
    if (c->h_stage) HIPCHK(hipHostFree(c->h_stage));
    c->h_stage = nullptr;
    c->stage_cap = 0;
    const size_t cap = out_words + out_words / 4 + 16;
    HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&c->h_stage), cap * sizeof(u32), hipHostMallocDefault));
    c->stage_cap = cap;
  }
  u32 flag = 0;
  HIPCHK(hipMemcpyAsync(c->h_stage, d_out, out_words * sizeof(u32), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipMemcpyAsync(&flag, d_flag, sizeof(u32), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  if (flag) return FailMsg("zmx_encode_blocks: the symbols of a block take a different number of bits than the histogram says");
  const unsigned char* stage = reinterpret_cast<const unsigned char*>(c->h_stage);
  zamd::ParallelFor(njobs, [&](size_t j) {
    const size_t nby = static_cast<size_t>((jobs[j].bit_start + jobs[j].nbits + 7) / 8);
    std::memcpy(out[j], stage + out_off[j], nby);
  });
  return 0;
}

// PNG filter heuristics (zmx_png.h): the filter type LodePNG's MINSUM / ENTROPY strategy picks for every scanline.
int zmx_png_filter_types(zmx_ctx* c, const unsigned char* image, size_t linebytes, size_t height, size_t bytewidth,
                         unsigned char* minsum_types, unsigned char* entropy_types) {
  if (!c) return FailMsg("zmx_png_filter_types: no context");
  if (height == 0 || (!minsum_types && !entropy_types)) return 0;
  if (linebytes == 0 || bytewidth == 0 || linebytes > 0x7fffffffu || height > 0x7fffffffu || bytewidth > 8) {
    return FailMsg("zmx_png_filter_types: bad geometry");
  }
  DeviceGuard dev_guard(c->device);
  HIPCHK(dev_guard.err);
  PoolScope tmp(c);
  u8* d_img = nullptr;
  u8* d_types = nullptr;
  const size_t bytes = linebytes * height;
  HIPCHK(tmp.AllocT(&d_img, bytes, "d_png_image"));
  HIPCHK(tmp.AllocT(&d_types, 2 * height, "d_png_types"));
  HIPCHK(hipMemcpyAsync(d_img, image, bytes, hipMemcpyHostToDevice, c->stream));
  PngFilterParams pp;
  pp.image = d_img;
  pp.linebytes = static_cast<u32>(linebytes);
  pp.height = static_cast<u32>(height);
  pp.bytewidth = static_cast<u32>(bytewidth);
  pp.minsum = minsum_types ? d_types : nullptr;
  pp.entropy = entropy_types ? d_types + height : nullptr;
  hipLaunchKernelGGL(k_png_filter_types, dim3(static_cast<unsigned>(height)), dim3(PNGF_THREADS), 0, c->stream, pp);
  KCHK(c, "k_png_filter_types");
  if (minsum_types) HIPCHK(hipMemcpyAsync(minsum_types, d_types, height, hipMemcpyDeviceToHost, c->stream));
  if (entropy_types) HIPCHK(hipMemcpyAsync(entropy_types, d_types + height, height, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return 0;
}

// Parity probe over WHOLE tables: two 64-bit sums over all positions of a hash of (block, position in the block,
// length, distance, same, literal, every change point of sublen) — the logical content of the records, whatever
// the order the pool handed out its entries in.  Two table sets over the same blocks with equal digests hold the
// same ZopfliFindLongestMatch results (test_match_kernels_agree compares the kernels at sizes no per-position
// loop reaches).
__device__ __forceinline__ u64 dg_mix(u64 h, u64 v) {
  h ^= v + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2);
  h *= 0xBF58476D1CE4E5B9ull;
  h ^= h >> 31;
  return h;
}
__global__ __launch_bounds__(256) void k_rec_digest(const BlockDesc* __restrict__ blocks, const u32* __restrict__ recs,
                                                    const u32* __restrict__ pool, unsigned long long* out) {
  const BlockDesc bd = blocks[blockIdx.y];
  const u64 B = bd.inend - bd.instart;
  u64 s0 = 0, s1 = 0;
  for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < B; i += (u64)gridDim.x * 256) {
    const u32* r = recs + (bd.pos_off + i) * 8;
    u64 h = dg_mix((u64)blockIdx.y * 0x100000001B3ull, i);
    h = dg_mix(h, r[0]);
    const u32 d1 = r[1];
    const u32 ncpf = d1 >> 24;
    h = dg_mix(h, d1 & 0xffffffu);
    if (ncpf != 0xffu) {
      const u8* b = reinterpret_cast<const u8*>(r) + 8;
      h = dg_mix(h, ncpf);
      for (u32 e = 0; e < ncpf; ++e) h = dg_mix(h, ((u32)b[3 * e] + 3u) | (((u32)b[3 * e + 1] | ((u32)b[3 * e + 2] << 8)) << 16));
    } else {
      const u32 off = r[2], n = r[3] & 0xffffu;
      h = dg_mix(h, n);
      for (u32 e = 0; e < n; ++e) h = dg_mix(h, pool[off + e]);
    }
    s0 += h;
    s1 += (h >> 17 | h << 47) * 0x94D049BB133111EBull;
  }
  atomicAdd(out, s0);
  atomicAdd(out + 1, s1);
}

int zmx_match_digest(zmx_ctx* c, zmx_tables* t, uint64_t* out2) {
  if (!t || t->trimmed || !t->d_recs) return FailMsg("zmx_match_digest: no match records in these tables");
  DeviceGuard dev_guard(c->device);
  HIPCHK(dev_guard.err);
  out2[0] = out2[1] = 0;
  if (t->nb == 0) return 0;
  PoolScope tmp(c);
  unsigned long long* d_out = nullptr;
  HIPCHK(tmp.AllocT(&d_out, 2, "d_digest"));
  HIPCHK(hipMemsetAsync(d_out, 0, 2 * sizeof(unsigned long long), c->stream));
  u64 max_b = 0;
  for (size_t b = 0; b < t->nb; ++b) max_b = std::max<u64>(max_b, t->bsize[b]);
  const unsigned gx = static_cast<unsigned>(std::min<u64>(std::max<u64>((max_b + 256 * 8 - 1) / (256 * 8), 1), 2048));
  hipLaunchKernelGGL(k_rec_digest, dim3(gx, static_cast<unsigned>(t->nb)), dim3(256), 0, c->stream, t->d_blocks, t->d_recs, t->d_pool, d_out);
  KCHK(c, "k_rec_digest");
  unsigned long long h[2] = {0, 0};
  HIPCHK(hipMemcpyAsync(h, d_out, sizeof(h), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  out2[0] = h[0];
  out2[1] = h[1];
  return 0;
}

int zmx_find_longest_match(zmx_ctx* c, zmx_tables* t, size_t block, size_t pos, uint16_t* sublen,
                           uint16_t* distance, uint16_t* length) {
  if (t && t->trimmed) return FailMsg("zmx_find_longest_match: these tables were trimmed to their stores (zmx_tables_trim)");
  if (block >= t->nb) return FailMsg("zmx_find_longest_match: bad block");
  const BlockDesc& d = t->blocks[block];
  if (pos < d.instart || pos >= d.inend) return FailMsg("zmx_find_longest_match: pos outside the block");
  DeviceGuard dev_guard(c->device);
  HIPCHK(dev_guard.err);
  if (t->probe_recs.empty()) t->probe_recs.resize(t->nb);
  std::vector<u32>& recs = t->probe_recs[block];
  if (recs.empty()) {
    recs.resize(static_cast<size_t>(t->bsize[block]) * 8);
    HIPCHK(hipMemcpy(recs.data(), t->d_recs + d.pos_off * 8, recs.size() * sizeof(u32), hipMemcpyDeviceToHost));
  }
  const u32* r = &recs[(pos - d.instart) * 8];
  *length = static_cast<uint16_t>(r[0] & 0xffffu);
  *distance = static_cast<uint16_t>(r[0] >> 16);
  if (!sublen) return 0;
  const u32 ncpf = r[1] >> 24;
  unsigned prev = 2;  // sublen[3..] only; the reference also fills sublen[2] for 2-byte hits, which nobody reads
  if (ncpf != 0xffu) {
    const u8* bytes = reinterpret_cast<const u8*>(r) + 8;
    for (u32 e = 0; e < ncpf; ++e) {
      const unsigned len = bytes[3 * e] + 3u, dist = bytes[3 * e + 1] | (bytes[3 * e + 2] << 8);
      for (unsigned l = prev + 1; l <= len; ++l) sublen[l] = static_cast<uint16_t>(dist);
      prev = len;
    }
  } else {
    if (!t->probe_pool_ready) {
      u32 used = 0;
      HIPCHK(hipMemcpy(&used, t->d_counters, sizeof(u32), hipMemcpyDeviceToHost));
      t->probe_pool.resize(used);
      if (used) HIPCHK(hipMemcpy(t->probe_pool.data(), t->d_pool, used * sizeof(u32), hipMemcpyDeviceToHost));
      t->probe_pool_ready = true;
    }
    const u32 off = r[2], n = r[3] & 0xffffu;
    for (u32 e = 0; e < n; ++e) {
      const u32 x = t->probe_pool[off + e];
      const unsigned len = x & 0xffffu, dist = x >> 16;
      for (unsigned l = prev + 1; l <= len; ++l) sublen[l] = static_cast<uint16_t>(dist);
      prev = len;
    }
  }
  return 0;
}

int zmx_hash_links_download(zmx_ctx* c, zmx_tables* t, size_t block, uint16_t* same, uint16_t* prev1, uint16_t* prev2) {
  if (t && t->trimmed) return FailMsg("zmx_hash_links_download: these tables were trimmed to their stores (zmx_tables_trim)");
  if (block >= t->nb) return FailMsg("zmx_hash_links_download: bad block");
  DeviceGuard dev_guard(c->device);
  HIPCHK(dev_guard.err);
  if (t->links_partial) return FailMsg("zmx_hash_links_download: tables built from a parent hold the hash arrays only near the block ends");
  const BlockDesc& d = t->blocks[block];
  const size_t n = static_cast<size_t>(d.inend - d.ws);
  if (t->buckets) {
    // k_bucket's arrays, read back as the reference's links: the previous position of the same hash value is the
    // entry below in the chunk's order, or the last of that value's bucket in the previous chunk — if it is
    // less than 32768 back (hash.c:110-114).  Every array of the structure is read here: sorted, rank, bucket.
    if (n) HIPCHK(hipMemcpy(same, t->d_same16 + d.reg_off, n * sizeof(u16), hipMemcpyDeviceToHost));
    const size_t nch = (n + BK_CH - 1) / BK_CH;
    std::vector<u16> srt(n), rnk(n);
    std::vector<u32> bkt(nch * 32768u);
    for (int h = 0; h < 2; ++h) {
      uint16_t* prev = h == 0 ? prev1 : prev2;
      if (n) {
        HIPCHK(hipMemcpy(srt.data(), t->d_sorted[h] + d.reg_off, n * sizeof(u16), hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(rnk.data(), t->d_rank[h] + d.reg_off, n * sizeof(u16), hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(bkt.data(), t->d_bucket[h] + static_cast<size_t>(t->chunk_base[block]) * 32768u, bkt.size() * sizeof(u32),
                         hipMemcpyDeviceToHost));
      }
      for (size_t k = 0; k < n; ++k) {
        const size_t cch = k / BK_CH, off = k % BK_CH;
        const unsigned char* in = c->h_in + d.ws;
        const u32 b0 = in[k], b1 = k + 1 < n ? in[k + 1] : 0u, b2 = k + 2 < n ? in[k + 2] : 0u;
        u32 key = ((b0 << 10) ^ (b1 << 5) ^ b2) & 32767u;
        if (h) key ^= (static_cast<u32>(same[k]) - 3u) & 255u;
        const u32 e = bkt[cch * 32768u + key];
        const u32 r = rnk[k];
        if (srt[cch * BK_CH + r] != off || r < (e & 0xffffu) || r >= (e >> 16)) {
          return FailMsg("zmx_hash_links_download: sorted / rank / bucket disagree at region position " + std::to_string(k));
        }
        u32 dist = 0;
        if (r > (e & 0xffffu)) {
          dist = static_cast<u32>(off) - srt[cch * BK_CH + r - 1];
        } else if (cch > 0) {
          const u32 ep = bkt[(cch - 1) * 32768u + key];
          if ((ep >> 16) > (ep & 0xffffu)) {
            const u32 offp = srt[(cch - 1) * BK_CH + (ep >> 16) - 1];
            if (offp > off) dist = BK_CH + static_cast<u32>(off) - offp;
          }
        }
        prev[k] = static_cast<uint16_t>(dist);
      }
    }
    return 0;
  }
  std::vector<ushort4> lk(n);
  if (n) HIPCHK(hipMemcpy(lk.data(), t->d_links + d.reg_off, n * sizeof(ushort4), hipMemcpyDeviceToHost));
  for (size_t i = 0; i < n; ++i) {
    prev1[i] = lk[i].x;
    prev2[i] = lk[i].y;
    same[i] = lk[i].z;
  }
  return 0;
}

int zmx_length_array_download(zmx_ctx* c, zmx_tables* t, size_t block, uint16_t* out) {
  if (t && t->trimmed) return FailMsg("zmx_length_array_download: these tables were trimmed to their stores (zmx_tables_trim)");
  if (block >= t->nb) return FailMsg("zmx_length_array_download: bad block");
  DeviceGuard dev_guard(c->device);
  HIPCHK(dev_guard.err);
  HIPCHK(hipMemcpy(out, t->d_la + t->blocks[block].la_off, (static_cast<size_t>(t->bsize[block]) + 1) * sizeof(u16),
                   hipMemcpyDeviceToHost));
  return 0;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// f-1 on the device: block sizes of ranges of symbol sequences (zmx_blockcost.h)
// ---------------------------------------------------------------------------------------------
struct zmx_cost_stores {
  size_t n = 0;
  std::vector<CostStoreDev> desc;
  u32* d_sym = nullptr;            // the sequences, one after the other
  u32* d_samples = nullptr;        // their sampled prefix counts
  CostStoreDev* d_desc = nullptr;
  CostEval* d_evals = nullptr;
  double* d_out = nullptr;
  CostEval* h_evals = nullptr;     // pinned
  double* h_out = nullptr;         // pinned
  size_t eval_cap = 0;
};

namespace {

void CostStoresRelease(zmx_ctx* c, zmx_cost_stores* s) {
  if (!s) return;
  PoolFree(c, s->d_sym);
  PoolFree(c, s->d_samples);
  PoolFree(c, s->d_desc);
  PoolFree(c, s->d_evals);
  PoolFree(c, s->d_out);
  if (s->h_evals) (void)hipHostFree(s->h_evals);
  if (s->h_out) (void)hipHostFree(s->h_out);
  delete s;
}

// the arrays of `n` sequences of sizes nsym[]; the symbols are put in place by the caller (between Layout and Finish)
int CostStoresLayout(zmx_ctx* c, size_t n, const std::vector<size_t>& nsym, zmx_cost_stores* s, std::vector<size_t>* sym_off) {
  size_t total = 0, samples = 0;
  sym_off->assign(n + 1, 0);
  for (size_t i = 0; i < n; ++i) {
    // (the sort key of the package-merge is count << 9 | symbol in 32 bits, and a package weighs at most 15 sequences)
    if (nsym[i] >= (1u << 22)) return FailMsg("zmx_cost_stores_create: a sequence of 2^22 symbols or more");
    (*sym_off)[i + 1] = (*sym_off)[i] + nsym[i];
    samples += nsym[i] / BC_S + 1;
  }
  total = (*sym_off)[n];
  s->n = n;
  HIPCHK(PoolAllocT(c, &s->d_sym, total + 1, "cost_sym"));
  HIPCHK(PoolAllocT(c, &s->d_samples, samples * BC_SW, "cost_samples"));
  HIPCHK(PoolAllocT(c, &s->d_desc, n, "cost_desc"));
  s->desc.resize(n);
  size_t so = 0;
  for (size_t i = 0; i < n; ++i) {
    s->desc[i].sym = s->d_sym + (*sym_off)[i];
    s->desc[i].samples = s->d_samples + so * BC_SW;
    s->desc[i].n = static_cast<u32>(nsym[i]);
    s->desc[i].nsamples = static_cast<u32>(nsym[i] / BC_S + 1);
    so += nsym[i] / BC_S + 1;
  }
  HIPCHK(hipMemcpyAsync(s->d_desc, s->desc.data(), n * sizeof(CostStoreDev), hipMemcpyHostToDevice, c->stream));
  return 0;
}

int CostStoresFinish(zmx_ctx* c, zmx_cost_stores* s) {
  const size_t n = s->n;
  std::vector<u32> chunk_first(n + 1, 0);
  for (size_t i = 0; i < n; ++i) chunk_first[i + 1] = chunk_first[i] + s->desc[i].n / BC_S;
  PoolScope tmp(c);
  u32* d_cf = nullptr;
  HIPCHK(tmp.AllocT(&d_cf, n + 1, "cost_chunk_first"));
  HIPCHK(hipMemcpyAsync(d_cf, chunk_first.data(), (n + 1) * sizeof(u32), hipMemcpyHostToDevice, c->stream));
  if (chunk_first[n]) {
    hipLaunchKernelGGL(k_cost_chunks, dim3((chunk_first[n] + 3) / 4), dim3(256), 0, c->stream, s->d_desc, d_cf, static_cast<u32>(n));
    KCHK(c, "k_cost_chunks");
  }
  hipLaunchKernelGGL(k_cost_prefix, dim3(static_cast<unsigned>(n)), dim3(384), 0, c->stream, s->d_desc);
  KCHK(c, "k_cost_prefix");
  HIPCHK(hipStreamSynchronize(c->stream));    // (chunk_first and the caller's staging arrays go out of scope)
  return 0;
}

}  // namespace

extern "C" {

int zmx_cost_stores_create(zmx_ctx* c, zmx_tables* t, size_t nstores, const size_t* piece_first, const size_t* block,
                           const int32_t* slot, const size_t* nsym, zmx_cost_stores** out) {
  *out = nullptr;
  if (nstores == 0) return FailMsg("zmx_cost_stores_create: no sequence");
  const size_t np = piece_first[nstores];
  std::vector<size_t> total(nstores, 0);
  for (size_t s = 0; s < nstores; ++s) {
    for (size_t p = piece_first[s]; p < piece_first[s + 1]; ++p) {
      if (block[p] >= t->nb || (slot[p] != 0 && slot[p] != 1)) return FailMsg("zmx_cost_stores_create: bad block or slot");
      if (t->store_begin[slot[p]][block[p]] + nsym[p] > t->bsize[block[p]]) return FailMsg("zmx_cost_stores_create: nsym exceeds the store");
      total[s] += nsym[p];
    }
  }
  DeviceGuard dev_guard(c->device);
  HIPCHK(dev_guard.err);
  zmx_cost_stores* s = new zmx_cost_stores();
  std::vector<size_t> sym_off;
  int rc = CostStoresLayout(c, nstores, total, s, &sym_off);
  if (rc) { CostStoresRelease(c, s); return rc; }
  auto fail = [&](hipError_t e, const char* what) {
    CostStoresRelease(c, s);
    return Fail(what, e, __FILE__, __LINE__);
  };
  std::vector<CostPiece> pieces(np);
  size_t longest = 0;
  for (size_t q = 0; q < nstores; ++q) {
    size_t off = sym_off[q];
    for (size_t p = piece_first[q]; p < piece_first[q + 1]; ++p) {
      pieces[p].src = t->d_store[slot[p]] + t->blocks[block[p]].pos_off + t->store_begin[slot[p]][block[p]];
      pieces[p].dst = s->d_sym + off;
      pieces[p].n = static_cast<u32>(nsym[p]);
      pieces[p].pad = 0;
      off += nsym[p];
      longest = std::max(longest, nsym[p]);
    }
  }
  {
    PoolScope tmp(c);
    CostPiece* d_pieces = nullptr;
    hipError_t e = tmp.AllocT(&d_pieces, np, "cost_pieces");
    if (e != hipSuccess) return fail(e, "cost_pieces");
    e = hipMemcpyAsync(d_pieces, pieces.data(), np * sizeof(CostPiece), hipMemcpyHostToDevice, c->stream);
    if (e != hipSuccess) return fail(e, "cost_pieces copy");
    if (longest) {
      hipLaunchKernelGGL(k_cost_gather, dim3(static_cast<unsigned>((longest + 4095) / 4096), static_cast<unsigned>(np)), dim3(256), 0, c->stream, d_pieces);
      e = hipGetLastError();
      if (e != hipSuccess) return fail(e, "k_cost_gather");
    }
    rc = CostStoresFinish(c, s);
  }
  if (rc) { CostStoresRelease(c, s); return rc; }
  *out = s;
  return 0;
}

int zmx_cost_stores_create_host(zmx_ctx* c, size_t nstores, const uint16_t* const* litlens, const uint16_t* const* dists,
                                const size_t* nsym, zmx_cost_stores** out) {
  *out = nullptr;
  if (nstores == 0) return FailMsg("zmx_cost_stores_create_host: no sequence");
  DeviceGuard dev_guard(c->device);
  HIPCHK(dev_guard.err);
  zmx_cost_stores* s = new zmx_cost_stores();
  std::vector<size_t> sym_off;
  int rc = CostStoresLayout(c, nstores, std::vector<size_t>(nsym, nsym + nstores), s, &sym_off);
  if (rc) { CostStoresRelease(c, s); return rc; }
  std::vector<u32> packed(sym_off[nstores] + 1);
  for (size_t q = 0; q < nstores; ++q) {
    for (size_t i = 0; i < nsym[q]; ++i) packed[sym_off[q] + i] = static_cast<u32>(litlens[q][i]) | (static_cast<u32>(dists[q][i]) << 16);
  }
  hipError_t e = hipMemcpyAsync(s->d_sym, packed.data(), sym_off[nstores] * sizeof(u32), hipMemcpyHostToDevice, c->stream);
  if (e != hipSuccess) { CostStoresRelease(c, s); return Fail("zmx_cost_stores_create_host: copy", e, __FILE__, __LINE__); }
  rc = CostStoresFinish(c, s);
  if (rc) { CostStoresRelease(c, s); return rc; }
  *out = s;
  return 0;
}

int zmx_cost_positions(zmx_ctx* c, zmx_cost_stores* s, size_t n, const uint32_t* pairs, uint64_t* bytes) {
  if (n == 0) return 0;
  for (size_t i = 0; i < n; ++i) {
    if (pairs[2 * i] >= s->n || pairs[2 * i + 1] > s->desc[pairs[2 * i]].n) return FailMsg("zmx_cost_positions: an index outside its sequence");
  }
  DeviceGuard dev_guard(c->device);
  HIPCHK(dev_guard.err);
  const hipStream_t cs = c->stream;   // (a stream of its own at the highest priority was measured: no shorter rounds — profiles/r06_device_split.txt)
  PoolScope tmp(c);
  CostPosQuery* d_q = nullptr;
  u64* d_out = nullptr;
  HIPCHK(tmp.AllocT(&d_q, n, "cost_pos_q"));
  HIPCHK(tmp.AllocT(&d_out, n, "cost_pos_out"));
  static_assert(sizeof(CostPosQuery) == 2 * sizeof(uint32_t), "the queries come as pairs");
  HIPCHK(hipMemcpyAsync(d_q, pairs, n * sizeof(CostPosQuery), hipMemcpyHostToDevice, cs));
  hipLaunchKernelGGL(k_cost_positions, dim3(static_cast<unsigned>(n)), dim3(64), 0, cs, s->d_desc, d_q, d_out);
  KCHK(c, "k_cost_positions");
  HIPCHK(hipMemcpyAsync(bytes, d_out, n * sizeof(u64), hipMemcpyDeviceToHost, cs));
  HIPCHK(hipStreamSynchronize(cs));
  return 0;
}

void zmx_cost_stores_free(zmx_ctx* c, zmx_cost_stores* s) {
  if (!s) return;
  DeviceGuard dev_guard(c->device);
  CostStoresRelease(c, s);
}

int zmx_block_costs(zmx_ctx* c, zmx_cost_stores* s, size_t n, const uint32_t* ranges, double* cost) {
  if (n == 0) return 0;
  for (size_t i = 0; i < n; ++i) {
    const uint32_t store = ranges[3 * i], lstart = ranges[3 * i + 1], lend = ranges[3 * i + 2];
    if (store >= s->n || lstart > lend || lend > s->desc[store].n) return FailMsg("zmx_block_costs: a range outside its sequence");
  }
  DeviceGuard dev_guard(c->device);
  HIPCHK(dev_guard.err);
  const hipStream_t cs = c->stream;   // (a stream of its own at the highest priority was measured: no shorter rounds — profiles/r06_device_split.txt)
  if (n > s->eval_cap) {
    const size_t cap = n + n / 2 + 256;
    PoolFree(c, s->d_evals); s->d_evals = nullptr;
    PoolFree(c, s->d_out); s->d_out = nullptr;
    if (s->h_evals) { (void)hipHostFree(s->h_evals); s->h_evals = nullptr; }
    if (s->h_out) { (void)hipHostFree(s->h_out); s->h_out = nullptr; }
    s->eval_cap = 0;
    HIPCHK(PoolAllocT(c, &s->d_evals, cap, "cost_evals"));
    HIPCHK(PoolAllocT(c, &s->d_out, cap, "cost_out"));
    HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&s->h_evals), cap * sizeof(CostEval), hipHostMallocDefault));
    HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&s->h_out), cap * sizeof(double), hipHostMallocDefault));
    s->eval_cap = cap;
  }
  for (size_t i = 0; i < n; ++i) s->h_evals[i] = {ranges[3 * i], ranges[3 * i + 1], ranges[3 * i + 2], 0u};
  HIPCHK(hipMemcpyAsync(s->d_evals, s->h_evals, n * sizeof(CostEval), hipMemcpyHostToDevice, cs));
  BlockCostParams P;
  P.stores = s->d_desc;
  P.evals = s->d_evals;
  P.out = s->d_out;
  P.n = static_cast<u32>(n);
  P.prof = nullptr;
  static const bool bc_prof = std::getenv("ZOPFLI_AMD_BC_PROF") != nullptr;
  PoolScope tmp(c);
  if (bc_prof) {
    HIPCHK(tmp.AllocT(&P.prof, 16, "bc_prof"));
    HIPCHK(hipMemsetAsync(P.prof, 0, 16 * sizeof(u64), cs));
  }
  const auto t0_ = std::chrono::steady_clock::now();
  hipLaunchKernelGGL(k_block_cost, dim3(static_cast<unsigned>((n + BC_WAVES / 2 - 1) / (BC_WAVES / 2))), dim3(64 * BC_WAVES), 0, cs, P);
  KCHK(c, "k_block_cost");
  HIPCHK(hipMemcpyAsync(s->h_out, s->d_out, n * sizeof(double), hipMemcpyDeviceToHost, cs));
  HIPCHK(hipStreamSynchronize(cs));
  std::memcpy(cost, s->h_out, n * sizeof(double));
  if (bc_prof) {
    u64 pr[16];
    HIPCHK(hipMemcpy(pr, P.prof, sizeof(pr), hipMemcpyDeviceToHost));
    const double us = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0_).count() * 1e6;
    std::fprintf(stderr, "k_block_cost: %zu block sizes in %.0f us; kilocycles a wave: histogram %.1f, lengths ll %.1f d %.1f, tree %.1f, data %.1f, smoothing %.1f, lengths ll %.1f d %.1f, tree + data %.1f\n",
                 n, us, pr[0] / 1e3 / n, pr[1] / 1e3 / n, pr[2] / 1e3 / n, pr[3] / 1e3 / n, pr[4] / 1e3 / n, pr[5] / 1e3 / n, pr[6] / 1e3 / n, pr[7] / 1e3 / n, pr[8] / 1e3 / n);
  }
  return 0;
}

}  // extern "C"

#include "block_split.h"

#include <algorithm>
#include <vector>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>

#include "block_cost.h"
#include "thread_pool.h"

namespace zamd {

namespace {

// ZOPFLI_AMD_PROF: evaluations of the split cost and the time they take, summed over all threads, printed per block.
std::atomic<unsigned long long> g_split_evals{0}, g_split_ns{0};
const bool g_split_prof = std::getenv("ZOPFLI_AMD_PROF") != nullptr;

constexpr double kLarge = 1e30;  // ZOPFLI_LARGE_FLOAT, util.h:65

struct SplitCost {
  const Lz77Store& lz77;
  size_t start, end;
  double operator()(size_t i) const {
    if (!g_split_prof) return CalculateBlockSizeAutoType(lz77, start, i) + CalculateBlockSizeAutoType(lz77, i, end);
    const auto t0 = std::chrono::steady_clock::now();
    const double v = CalculateBlockSizeAutoType(lz77, start, i) + CalculateBlockSizeAutoType(lz77, i, end);
    g_split_ns += static_cast<unsigned long long>(std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count());
    ++g_split_evals;
    return v;
  }
};

// Minimum of f over [start, end): exhaustive below 1024 candidates, otherwise
// a 9-point recursive refinement that stops as soon as a round does not improve
// (blocksplitter.c:43-96).  The candidates of a round are independent block-size evaluations: they
// run on the worker pool when this call is not itself inside a parallel loop — a request of one
// master block (every zopflipng IDAT below 1 MB) has no other parallelism to offer.
size_t FindMinimum(const SplitCost& f, size_t start, size_t end, double* smallest) {
  if (end - start < 1024) {
    const size_t n = end - start;
    std::vector<double> v(n);
    ParallelForNested((n + 15) / 16, [&](size_t c) {
      for (size_t i = c * 16; i < n && i < c * 16 + 16; ++i) v[i] = f(start + i);
    });
    double best = kLarge;
    size_t arg = start;
    for (size_t i = 0; i < n; ++i) {
      if (v[i] < best) {      // strict: the first minimum, as the reference's loop
        best = v[i];
        arg = start + i;
      }
    }
    *smallest = best;
    return arg;
  }
  constexpr int kProbes = 9;
  size_t probe[kProbes];
  double value[kProbes];
  double last_best = kLarge;
  size_t pos = start;
  while (end - start > kProbes) {
    const size_t step = (end - start) / (kProbes + 1);
    for (int i = 0; i < kProbes; ++i) probe[i] = start + (i + 1) * step;
    ParallelForNested(kProbes, [&](size_t i) { value[i] = f(probe[i]); });
    int arg = 0;
    for (int i = 1; i < kProbes; ++i) {
      if (value[i] < value[arg]) arg = i;
    }
    if (value[arg] > last_best) break;
    const size_t lo = arg == 0 ? start : probe[arg - 1];
    const size_t hi = arg == kProbes - 1 ? end : probe[arg + 1];
    start = lo;
    end = hi;
    pos = probe[arg];
    last_best = value[arg];
  }
  *smallest = last_best;
  return pos;
}

}  // namespace

void BlockSplitLz77(const Lz77Store& lz77, size_t maxblocks, std::vector<size_t>* points) {
  const size_t n = lz77.size();
  if (n < 10) return;
  std::vector<char> done(n, 0);
  size_t lstart = 0, lend = n;
  size_t numblocks = 1;
  for (;;) {
    if (maxblocks > 0 && numblocks >= maxblocks) break;
    double splitcost;
    const SplitCost f{lz77, lstart, lend};
    const size_t llpos = FindMinimum(f, lstart + 1, lend, &splitcost);
    const double origcost = CalculateBlockSizeAutoType(lz77, lstart, lend);
    if (splitcost > origcost || llpos == lstart + 1 || llpos == lend) {
      done[lstart] = 1;
    } else {
      points->insert(std::upper_bound(points->begin(), points->end(), llpos), llpos);
      numblocks++;
    }
    // next candidate: the longest block not yet marked unsplittable; the last
    // block is measured up to n-1 (blocksplitter.c:203)
    bool found = false;
    size_t longest = 0;
    for (size_t i = 0; i <= points->size(); ++i) {
      const size_t s = i == 0 ? 0 : (*points)[i - 1];
      const size_t e = i == points->size() ? n - 1 : (*points)[i];
      if (!done[s] && e - s > longest) {
        lstart = s;
        lend = e;
        found = true;
        longest = e - s;
      }
    }
    if (!found) break;
    if (lend - lstart < 10) break;
  }
  if (g_split_prof)
    std::fprintf(stderr, "BlockSplitLz77: %zu symbols, %zu split points; so far %llu evaluations of the split cost, %.1f us each\n", n,
                 points->size(), g_split_evals.load(), g_split_evals.load() ? g_split_ns.load() * 1e-3 / g_split_evals.load() : 0.0);
}

// ---------------------------------------------------------------------------------------------
// The same search for a FEW symbol sequences at once, in rounds.
//
// ZopfliBlockSplitLZ77 is sequential twice over: a block is split at a time (the longest not yet found
// unsplittable, blocksplitter.c:195-213), and every FindMinimum is a chain of rounds of 9 evaluations each of which
// needs the winner of the one before (:61-89).  BlockSplitLz77 above runs one sequence per worker thread, which is the
// right thing for a hundred master blocks — and leaves a request of one to eight master blocks (a zopflipng IDAT,
// a file of a few MB) on as many threads for ~70 dependent rounds each: 5 ms of a 19 ms call for 64 KiB, 23 ms of 61
// for 4 MB.  Here every sequence advances round by round and all evaluations of a round, of all sequences, are ONE
// parallel loop; and since the search of a block depends on nothing but the block (its range of the sequence), the
// blocks that EXIST are searched before their turn comes — both halves of a split at once — so the number of rounds
// is the depth of the split tree, not the number of splits.  The decisions are then taken in the reference's order
// from the finished searches; searches whose block never gets its turn (the limit of `maxblocks` was reached first)
// are thrown away.  Same split points as BlockSplitLz77 (tests/test_cpu_oracle_vs_reference.py).
// ---------------------------------------------------------------------------------------------
namespace {

struct BlockSearch {           // FindMinimum over (s, e) plus the cost of the block unsplit
  size_t s = 0, e = 0;
  size_t lo = 0, hi = 0;       // the range still in play
  bool exhaustive = false, finished = false;
  double last_best = kLarge, origcost = 0;
  size_t pos = 0;
  std::vector<size_t> probes;  // this round's evaluations: split positions, and s itself = "the block unsplit"
  std::vector<double> values;
  double splitcost = 0;
  size_t llpos = 0;

  void Start(size_t s_, size_t e_) {
    s = s_; e = e_;
    lo = s + 1; hi = e;
    pos = lo;
    exhaustive = hi - lo < 1024;
    Plan(true);
  }
  void Plan(bool first) {
    probes.clear();
    if (first) probes.push_back(s);   // (no split position equals s: the unsplit cost rides along with the first round)
    if (exhaustive) {
      for (size_t i = lo; i < hi; ++i) probes.push_back(i);
    } else if (hi - lo > 9) {
      const size_t step = (hi - lo) / 10;
      for (size_t i = 0; i < 9; ++i) probes.push_back(lo + (i + 1) * step);
    }
    values.assign(probes.size(), 0.0);
    if (probes.size() == (first ? 1u : 0u) && !exhaustive) {
      // nothing to evaluate: the range has 9 candidates or fewer from the start (blocksplitter.c:61: the loop does not run)
      if (!first) Finish();
    }
  }
  void Finish() {
    finished = true;
    splitcost = last_best;
    llpos = pos;
  }
  // the round's values are in: blocksplitter.c:49-58 (exhaustive) or one turn of :61-89
  void Feed() {
    size_t k = 0;
    if (!probes.empty() && probes[0] == s) { origcost = values[0]; k = 1; }
    if (exhaustive) {
      double best = kLarge;
      size_t arg = lo;
      for (size_t i = lo; i < hi; ++i, ++k) {
        if (values[k] < best) { best = values[k]; arg = i; }
      }
      last_best = best;
      pos = arg;
      Finish();
      return;
    }
    if (probes.size() - k != 9) { Finish(); return; }    // hi - lo <= 9 at the start: FindMinimum returns (start, kLarge)
    size_t arg = 0;
    for (size_t i = 1; i < 9; ++i) if (values[k + i] < values[k + arg]) arg = i;
    if (values[k + arg] > last_best) { Finish(); return; }
    const size_t nlo = arg == 0 ? lo : probes[k + arg - 1];
    const size_t nhi = arg == 8 ? hi : probes[k + arg + 1];
    pos = probes[k + arg];
    last_best = values[k + arg];
    lo = nlo;
    hi = nhi;
    if (hi - lo > 9) Plan(false); else Finish();
  }
};

struct SeqSplit {              // ZopfliBlockSplitLZ77's loop over one sequence, fed by finished searches
  const Lz77Store* lz77 = nullptr;
  size_t n = 0, maxblocks = 0, numblocks = 1;
  std::vector<size_t>* points = nullptr;
  std::vector<char> done;
  size_t lstart = 0, lend = 0;
  bool over = false;
  std::vector<BlockSearch> searches;     // of the blocks that exist now (or existed)

  BlockSearch* Find(size_t s, size_t e) {
    for (auto& b : searches) if (b.s == s && b.e == e) return &b;
    return nullptr;
  }
  void Want(size_t s, size_t e) {
    if (e - s < 10 || Find(s, e)) return;
    searches.emplace_back();
    searches.back().Start(s, e);
  }
  void Init(const Lz77Store* store, size_t size, size_t maxb, std::vector<size_t>* pts) {
    lz77 = store;
    n = size;
    maxblocks = maxb;
    points = pts;
    if (n < 10) { over = true; return; }
    done.assign(n, 0);
    lstart = 0;
    lend = n;
    if (maxblocks > 0 && numblocks >= maxblocks) { over = true; return; }
    Want(lstart, lend);     // (the whole sequence: even below 10 symbols' length it is searched once — n >= 10 here)
    if (!Find(lstart, lend)) { searches.emplace_back(); searches.back().Start(lstart, lend); }
  }
  // every block that exists and may still get its turn
  void Speculate() {
    for (size_t i = 0; i <= points->size(); ++i) {
      const size_t s = i == 0 ? 0 : (*points)[i - 1];
      const size_t e = i == points->size() ? n - 1 : (*points)[i];
      if (!done[s]) Want(s, e);
    }
  }
  // blocksplitter.c:233-262 for as long as the searches it needs are finished
  void Advance(bool speculate) {
    while (!over) {
      BlockSearch* b = Find(lstart, lend);
      if (!b) { searches.emplace_back(); searches.back().Start(lstart, lend); return; }
      if (!b->finished) return;
      if (b->splitcost > b->origcost || b->llpos == lstart + 1 || b->llpos == lend) {
        done[lstart] = 1;
      } else {
        points->insert(std::upper_bound(points->begin(), points->end(), b->llpos), b->llpos);
        numblocks++;
      }
      if (maxblocks > 0 && numblocks >= maxblocks) { over = true; break; }
      bool found = false;
      size_t longest = 0;
      for (size_t i = 0; i <= points->size(); ++i) {
        const size_t s = i == 0 ? 0 : (*points)[i - 1];
        const size_t e = i == points->size() ? n - 1 : (*points)[i];
        if (!done[s] && e - s > longest) {
          lstart = s;
          lend = e;
          found = true;
          longest = e - s;
        }
      }
      if (!found || lend - lstart < 10) { over = true; break; }
      if (speculate) Speculate();
    }
  }
};

}  // namespace

namespace {

// the rounds of BlockSplitLz77Batch; false: a round the device had to serve (no host store) was not served
bool SplitRounds(std::vector<SeqSplit>& seq, const CostBatchFn* device, size_t device_min) {
  const size_t ns = seq.size();
  struct Eval { SeqSplit* q; BlockSearch* b; size_t k; };
  std::vector<Eval> evals;
  std::vector<CostQuery> queries;
  std::vector<double> answers;
  static const bool trace = [] { const char* e = std::getenv("ZOPFLI_AMD_TRACE_CALL"); return e && std::atoi(e) != 0; }();
  size_t rounds_dev = 0, rounds_host = 0, n_dev = 0, n_host = 0;
  double t_dev = 0, t_host = 0;
  auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  bool host_possible = true;
  for (auto& q : seq) host_possible &= q.lz77 != nullptr || q.over;
  for (;;) {
    evals.clear();
    for (auto& q : seq) {
      if (q.over) continue;
      for (auto& b : q.searches) {
        if (b.finished) continue;
        for (size_t k = 0; k < b.probes.size(); ++k) evals.push_back({&q, &b, k});
      }
    }
    if (evals.empty()) {
      bool any = false;
      for (auto& q : seq) any |= !q.over;
      if (!any) break;
      // (searches without evaluations finish in Feed: ranges of 9 candidates or fewer)
    }
    // (a round of a few hundred evaluations of ~10 us: the regular pool's 32 threads, not a wake-up of the wide one's 128)
    auto one = [&](size_t i) {
      const Eval& e = evals[i];
      const size_t p = e.b->probes[e.k];
      e.b->values[e.k] = p == e.b->s ? CalculateBlockSizeAutoType(*e.q->lz77, e.b->s, e.b->e)
                                     : CalculateBlockSizeAutoType(*e.q->lz77, e.b->s, p) + CalculateBlockSizeAutoType(*e.q->lz77, p, e.b->e);
    };
    bool served = false;
    const double tr0 = trace ? now() : 0.0;
    if (device && *device && !evals.empty() && (2 * evals.size() >= device_min || !host_possible)) {
      // the round's block sizes in one launch: the block unsplit is one size, a split position two (blocksplitter.c:117-135)
      queries.clear();
      for (const Eval& e : evals) {
        const uint32_t sq = static_cast<uint32_t>(e.q - seq.data());
        const uint32_t p = static_cast<uint32_t>(e.b->probes[e.k]), s = static_cast<uint32_t>(e.b->s), en = static_cast<uint32_t>(e.b->e);
        if (p == s) {
          queries.push_back({sq, s, en});
        } else {
          queries.push_back({sq, s, p});
          queries.push_back({sq, p, en});
        }
      }
      answers.resize(queries.size());
      served = (*device)(queries.data(), queries.size(), answers.data());
      if (served) {
        size_t a = 0;
        for (const Eval& e : evals) {
          const size_t p = e.b->probes[e.k];
          if (p == e.b->s) { e.b->values[e.k] = answers[a]; a += 1; }
          else { e.b->values[e.k] = answers[a] + answers[a + 1]; a += 2; }
        }
      }
    }
    if (!served) {
      if (!host_possible && !evals.empty()) return false;
      if (evals.size() > 16u * HostThreads()) ParallelForWide(evals.size(), one); else ParallelFor(evals.size(), one);
    }
    if (trace) {
      const double dt = now() - tr0;
      if (served) { ++rounds_dev; n_dev += queries.size(); t_dev += dt; } else { ++rounds_host; n_host += evals.size(); t_host += dt; }
    }
    for (auto& q : seq) {
      if (q.over) continue;
      // (Feed may append to nothing; Advance may append searches: indices, not references, across the call)
      for (size_t i = 0; i < q.searches.size(); ++i) if (!q.searches[i].finished) q.searches[i].Feed();
      q.Advance(true);
    }
  }
  if (trace) {
    std::fprintf(stderr, "      block-split rounds (%zu sequences): %zu on the device (%zu block sizes, %.2f ms), %zu on the host (%zu split costs, %.2f ms)\n",
                 ns, rounds_dev, n_dev, t_dev * 1e3, rounds_host, n_host, t_host * 1e3);
  }
  return true;
}

}  // namespace

void BlockSplitLz77Batch(const std::vector<const Lz77Store*>& stores, size_t maxblocks,
                         std::vector<std::vector<size_t>>* points, const CostBatchFn* device, size_t device_min) {
  const size_t ns = stores.size();
  points->assign(ns, {});
  std::vector<SeqSplit> seq(ns);
  for (size_t i = 0; i < ns; ++i) seq[i].Init(stores[i], stores[i]->size(), maxblocks, &(*points)[i]);
  SplitRounds(seq, device, device_min);
}

bool BlockSplitSizesBatch(const std::vector<size_t>& sizes, size_t maxblocks, std::vector<std::vector<size_t>>* points,
                          const CostBatchFn& device) {
  const size_t ns = sizes.size();
  points->assign(ns, {});
  std::vector<SeqSplit> seq(ns);
  for (size_t i = 0; i < ns; ++i) seq[i].Init(nullptr, sizes[i], maxblocks, &(*points)[i]);
  return SplitRounds(seq, &device, 0);
}

std::vector<size_t> SplitPointsToBytes(const Lz77Store& lz77, const std::vector<size_t>& points,
                                       size_t instart) {
  std::vector<size_t> out;
  out.reserve(points.size());
  (void)instart;  // positions in the store are already absolute
  for (size_t p : points) out.push_back(lz77.pos(p));
  return out;
}

}  // namespace zamd

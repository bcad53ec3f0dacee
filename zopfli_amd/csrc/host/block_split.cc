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
    ParallelFor((n + 15) / 16, [&](size_t c) {
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
    ParallelFor(kProbes, [&](size_t i) { value[i] = f(probe[i]); });
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

std::vector<size_t> SplitPointsToBytes(const Lz77Store& lz77, const std::vector<size_t>& points,
                                       size_t instart) {
  std::vector<size_t> out;
  out.reserve(points.size());
  (void)instart;  // positions in the store are already absolute
  for (size_t p : points) out.push_back(lz77.pos(p));
  return out;
}

}  // namespace zamd

#include "lz77_optimal.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "block_cost.h"
#include "huffman.h"
#include "symbols.h"
#include "thread_pool.h"

namespace zamd {

namespace {

constexpr double kLarge = 1e30;  // ZOPFLI_LARGE_FLOAT

double Now() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// SymbolStats (squeeze.c:32-42): frequencies and the entropy costs derived from them.
struct SymbolStats {
  size_t litlens[kNumLL];
  size_t dists[kNumD];
  double ll_symbols[kNumLL];
  double d_symbols[kNumD];
};

void CalculateStatistics(SymbolStats* s) {  // squeeze.c:392
  CalculateEntropy(s->litlens, kNumLL, s->ll_symbols);
  CalculateEntropy(s->dists, kNumD, s->d_symbols);
}

// Frequencies from a device histogram, plus the end symbol (GetStatistics, squeeze.c:398).
void StatsFromHistogram(const uint32_t* hist, SymbolStats* s) {
  for (int i = 0; i < kNumLL; ++i) s->litlens[i] = hist[i];
  for (int i = 0; i < kNumD; ++i) s->dists[i] = hist[kNumLL + i];
  s->litlens[256] = 1;
  CalculateStatistics(s);
}

// Marsaglia multiply-with-carry, seeded (1, 2) per block (squeeze.c:80-94).
struct Mwc {
  uint32_t w = 1, z = 2;
  uint32_t Next() {
    z = 36969 * (z & 65535) + (z >> 16);
    w = 18000 * (w & 65535) + (w >> 16);
    return (z << 16) + w;
  }
};

void RandomizeFreqs(Mwc* rng, size_t* freqs, int n) {  // squeeze.c:96
  for (int i = 0; i < n; ++i) {
    if ((rng->Next() >> 4) % 3 == 0) freqs[i] = freqs[rng->Next() % n];
  }
}

// GetCostStat for a match (squeeze.c:146-157): int extra bits first, then the
// two symbol costs, added left to right.
inline double MatchCost(const double* ll, const double* d, int length, int dist) {
  return LengthExtraBits(length) + DistExtraBits(dist) + ll[LengthSymbol(length)] + d[DistSymbol(dist)];
}

// GetCostModelMinCost (squeeze.c:163-198).
double ModelMinCost(const double* ll, const double* d) {
  static const int kFirstDist[30] = {1,   2,   3,   4,   5,   7,    9,    13,   17,   25,
                                     33,  49,  65,  97,  129, 193,  257,  385,  513,  769,
                                     1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
  int bestlength = 0, bestdist = 0;
  double m = kLarge;
  for (int i = 3; i < 259; ++i) {
    const double c = MatchCost(ll, d, i, 1);
    if (c < m) {
      bestlength = i;
      m = c;
    }
  }
  m = kLarge;
  for (int i = 0; i < 30; ++i) {
    const double c = MatchCost(ll, d, 3, kFirstDist[i]);
    if (c < m) {
      bestdist = kFirstDist[i];
      m = c;
    }
  }
  return MatchCost(ll, d, bestlength, bestdist);
}

void ToHistogram(const uint32_t* hist, Histogram* h) {
  for (int i = 0; i < kNumLL; ++i) h->ll[i] = hist[i];
  for (int i = 0; i < kNumD; ++i) h->d[i] = hist[kNumLL + i];
}

struct BlockIter {
  SymbolStats stats, beststats, laststats;
  Mwc rng;
  double bestcost = kLarge, lastcost = 0;
  int lastrandomstep = -1;
  int best_slot = -1;   // device store slot holding the best parse so far
  uint32_t best_nsym = 0;
};

// Stores slot[b] of every block b with slot[b] >= 0, nsym[b] symbols each, into (*out)[b].
int DownloadAll(zmx_ctx* ctx, zmx_tables* t, const std::vector<int32_t>& slot, const std::vector<uint32_t>& nsym,
                std::vector<SymbolRun>* out) {
  std::vector<size_t> block, n;
  std::vector<int32_t> sl;
  std::vector<uint16_t*> ll, dd;
  ParallelFor(slot.size(), [&](size_t b) {   // first touch of ~0.6 B per input byte: not on one thread
    if (slot[b] < 0) return;
    (*out)[b].litlens.resize(nsym[b]);
    (*out)[b].dists.resize(nsym[b]);
  });
  for (size_t b = 0; b < slot.size(); ++b) {
    if (slot[b] < 0) continue;
    SymbolRun& run = (*out)[b];
    block.push_back(b);
    sl.push_back(slot[b]);
    n.push_back(nsym[b]);
    ll.push_back(run.litlens.data());
    dd.push_back(run.dists.data());
  }
  return zmx_store_download_batch(ctx, t, block.size(), block.data(), sl.data(), n.data(), ll.data(), dd.data());
}

}  // namespace

// ZOPFLI_AMD_VERIFY: ZopfliVerifyLenDist (lz77.c:270-295) on the device for every parse that is kept
static bool VerifyWanted() {
  static const bool v = [] { const char* e = std::getenv("ZOPFLI_AMD_VERIFY"); return e && std::atoi(e) != 0; }();
  return v;
}
static int VerifyAll(zmx_ctx* ctx, zmx_tables* t, const std::vector<int32_t>& slot, const std::vector<uint32_t>& nsym) {
  std::vector<size_t> block(slot.size()), n(nsym.begin(), nsym.end());
  for (size_t b = 0; b < block.size(); ++b) block[b] = b;
  return zmx_verify_stores(ctx, t, block.size(), block.data(), slot.data(), n.data());
}

Timing& ThreadTiming() {
  static thread_local Timing t;
  return t;
}

int Lz77GreedyDownload(zmx_ctx* ctx, zmx_tables* tables, const std::vector<uint32_t>& nsym, std::vector<SymbolRun>* out) {
  const double t1 = Now();
  out->assign(nsym.size(), SymbolRun());
  const int rc = DownloadAll(ctx, tables, std::vector<int32_t>(nsym.size(), 0), nsym, out);
  ThreadTiming().greedy += Now() - t1;
  return rc;
}

int Lz77GreedyBatch(zmx_ctx* ctx, const std::vector<zmx_block>& blocks, std::vector<SymbolRun>* out,
                    zmx_tables** keep, std::vector<uint32_t>* nsym_out, bool defer_download) {
  const size_t nb = blocks.size();
  out->assign(nb, SymbolRun());
  if (keep) *keep = nullptr;
  if (nb == 0) return 0;
  zmx_tables* t = nullptr;
  double t0 = Now();
  int rc = zmx_tables_build_matches(ctx, blocks.data(), nb, &t);   // (nobody runs the squeeze on these blocks)
  if (rc) return rc;
  double t1 = Now();
  ThreadTiming().tables += t1 - t0;
  std::vector<uint32_t> nsym(nb), hist(nb * ZMX_HIST);
  rc = zmx_lz77_greedy(ctx, t, 0, nsym.data(), hist.data());
  if (!rc && VerifyWanted()) rc = VerifyAll(ctx, t, std::vector<int32_t>(nb, 0), nsym);
  if (!rc && !(defer_download && nsym_out)) rc = DownloadAll(ctx, t, std::vector<int32_t>(nb, 0), nsym, out);
  if (nsym_out) *nsym_out = nsym;
  ThreadTiming().greedy += Now() - t1;
  if (keep && !rc) *keep = t;
  else zmx_tables_free(ctx, t);
  return rc;
}

int Lz77OptimalBatch(zmx_ctx* ctx, const ZopfliOptions& options, const std::vector<zmx_block>& blocks,
                     std::vector<SymbolRun>* out, zmx_tables* parent, OptimalKeep* keep) {
  const size_t nb = blocks.size();
  out->assign(nb, SymbolRun());
  if (nb == 0) {
    if (parent) zmx_tables_free(ctx, parent);
    return 0;
  }
  zmx_tables* t = nullptr;
  double t0 = Now();
  int rc = zmx_tables_build_from(ctx, parent, blocks.data(), nb, &t);
  if (parent) zmx_tables_free(ctx, parent);
  if (rc) return rc;
  double t1 = Now();
  ThreadTiming().tables += t1 - t0;

  std::vector<uint32_t> nsym(nb), hist(nb * ZMX_HIST), best_hist(keep ? nb * ZMX_HIST : 0);
  std::vector<BlockIter> it(nb);
  std::vector<double> cost(nb * ZMX_HIST), mincost(nb);
  std::vector<int32_t> slot(nb, 0);

  // Seed: greedy parse -> statistics (squeeze.c:480-482).  The greedy store is
  // only needed for its histogram, so it goes to slot 0 and is overwritten.
  rc = zmx_lz77_greedy(ctx, t, 0, nsym.data(), hist.data());
  if (rc) { zmx_tables_free(ctx, t); return rc; }
  double t2 = Now();
  ThreadTiming().greedy += t2 - t1;
  ParallelFor(nb, [&](size_t b) {
    std::memset(&it[b].stats, 0, sizeof(SymbolStats));
    StatsFromHistogram(&hist[b * ZMX_HIST], &it[b].stats);
  });
  ThreadTiming().cost_model += Now() - t2;

  // the cost model of a block's next run, from its statistics (done at the end of the loop below for every run but
  // the first: one wake-up of the workers per run instead of two)
  auto next_model = [&](size_t b) {
    std::memcpy(&cost[b * ZMX_HIST], it[b].stats.ll_symbols, sizeof(double) * kNumLL);
    std::memcpy(&cost[b * ZMX_HIST + kNumLL], it[b].stats.d_symbols, sizeof(double) * kNumD);
    mincost[b] = ModelMinCost(it[b].stats.ll_symbols, it[b].stats.d_symbols);
    slot[b] = it[b].best_slot == 0 ? 1 : 0;  // never overwrite the best parse
  };
  {
    const double ta = Now();
    ParallelFor(nb, next_model);
    ThreadTiming().cost_model += Now() - ta;
  }
  for (int i = 0; i < options.numiterations; ++i) {
    double ta = Now();
    double tb = ta;
    rc = zmx_squeeze_run(ctx, t, cost.data(), mincost.data(), slot.data(), nsym.data(), hist.data());
    if (rc) { zmx_tables_free(ctx, t); return rc; }
    double tc = Now();
    ParallelFor(nb, [&](size_t b) {
      BlockIter& s = it[b];
      const uint32_t* h = &hist[b * ZMX_HIST];
      Histogram hh;
      ToHistogram(h, &hh);
      // ZopfliCalculateBlockSize(&currentstore, 0, size, 2): depends on the
      // histogram only (deflate.c:584 -> :569 -> :383)
      const double c = BlockSizeFromHistogram(hh, 2);
      if (options.verbose_more || (options.verbose && c < s.bestcost)) {
        char line[64];
        std::snprintf(line, sizeof(line), "Iteration %d: %d bit\n", i, static_cast<int>(c));
        (*out)[b].log += line;   // (printed in the reference's order when the stream is assembled)
      }
      if (c < s.bestcost) {
        if (keep) std::memcpy(&best_hist[b * ZMX_HIST], h, sizeof(uint32_t) * ZMX_HIST);
        s.best_slot = slot[b];
        s.best_nsym = nsym[b];
        s.beststats = s.stats;
        s.bestcost = c;
      }
      s.laststats = s.stats;
      std::memset(s.stats.litlens, 0, sizeof(s.stats.litlens));
      std::memset(s.stats.dists, 0, sizeof(s.stats.dists));
      StatsFromHistogram(h, &s.stats);
      if (s.lastrandomstep != -1) {
        // stats = stats*1.0 + laststats*0.5, truncated (AddWeighedStatFreqs, squeeze.c:65)
        for (int k = 0; k < kNumLL; ++k) {
          s.stats.litlens[k] = static_cast<size_t>(s.stats.litlens[k] * 1.0 + s.laststats.litlens[k] * 0.5);
        }
        for (int k = 0; k < kNumD; ++k) {
          s.stats.dists[k] = static_cast<size_t>(s.stats.dists[k] * 1.0 + s.laststats.dists[k] * 0.5);
        }
        s.stats.litlens[256] = 1;
        CalculateStatistics(&s.stats);
      }
      if (i > 5 && c == s.lastcost) {
        s.stats = s.beststats;
        RandomizeFreqs(&s.rng, s.stats.litlens, kNumLL);
        RandomizeFreqs(&s.rng, s.stats.dists, kNumD);
        s.stats.litlens[256] = 1;
        CalculateStatistics(&s.stats);
        s.lastrandomstep = i;
      }
      s.lastcost = c;
      next_model(b);
    });
    double td = Now();
    ThreadTiming().squeeze += tc - tb;
    ThreadTiming().cost_model += (tb - ta) + (td - tc);
  }

  const double tdl = Now();
  {
    std::vector<int32_t> best_slot(nb);
    std::vector<uint32_t> best_nsym(nb);
    for (size_t b = 0; b < nb; ++b) { best_slot[b] = it[b].best_slot; best_nsym[b] = it[b].best_nsym; }
    if (!rc && VerifyWanted()) rc = VerifyAll(ctx, t, best_slot, best_nsym);
    if (!rc && !(keep && keep->skip_download)) rc = DownloadAll(ctx, t, best_slot, best_nsym, out);
    if (keep && !rc) {
      keep->tables = t;
      keep->slot = best_slot;
      keep->nsym = best_nsym;
      keep->hist.swap(best_hist);
      t = nullptr;
    }
  }
  ThreadTiming().download += Now() - tdl;
  if (t) zmx_tables_free(ctx, t);
  return rc;
}

int Lz77OptimalFixedBatch(zmx_ctx* ctx, const std::vector<zmx_block>& blocks, std::vector<SymbolRun>* out,
                          OptimalKeep* keep) {
  const size_t nb = blocks.size();
  out->assign(nb, SymbolRun());
  if (nb == 0) return 0;
  zmx_tables* t = nullptr;
  double t0 = Now();
  int rc = zmx_tables_build(ctx, blocks.data(), nb, &t);
  if (rc) return rc;
  double t1 = Now();
  ThreadTiming().tables += t1 - t0;
  // GetCostFixed (squeeze.c:125-140) as a symbol-cost table: the sums are small
  // integers, exact in double in any association.
  std::vector<double> cost(nb * ZMX_HIST), mincost(nb);
  std::vector<int32_t> slot(nb, 0);
  std::vector<uint32_t> nsym(nb), hist(nb * ZMX_HIST);
  double ll[kNumLL], d[kNumD];
  for (int i = 0; i < kNumLL; ++i) ll[i] = i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8;
  for (int i = 0; i < kNumD; ++i) d[i] = 5;
  const double mc = ModelMinCost(ll, d);
  for (size_t b = 0; b < nb; ++b) {
    std::memcpy(&cost[b * ZMX_HIST], ll, sizeof(ll));
    std::memcpy(&cost[b * ZMX_HIST + kNumLL], d, sizeof(d));
    mincost[b] = mc;
  }
  rc = zmx_squeeze_run(ctx, t, cost.data(), mincost.data(), slot.data(), nsym.data(), hist.data());
  if (!rc && VerifyWanted()) rc = VerifyAll(ctx, t, std::vector<int32_t>(nb, 0), nsym);
  if (!rc && !(keep && keep->skip_download)) rc = DownloadAll(ctx, t, std::vector<int32_t>(nb, 0), nsym, out);
  ThreadTiming().squeeze += Now() - t1;
  if (keep && !rc) {
    keep->tables = t;
    keep->slot.assign(nb, 0);
    keep->nsym = nsym;
    keep->hist.swap(hist);
    t = nullptr;
  }
  if (t) zmx_tables_free(ctx, t);
  return rc;
}

}  // namespace zamd

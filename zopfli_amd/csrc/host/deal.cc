#include "deal.h"

#include <algorithm>
#include <cstdint>

#include "zopfli_amd.h"

namespace zamd {

// What a kind of data costs relative to text, per byte (profiles/r05_classes.json: ms per 100 MB through ZopfliCompress,
// numiterations 15 — text 123, long runs of equal bytes ~ 4.6 x, two-symbol data 2.3 x; markup, PNG-like and random data
// within 30 % of text).  Long runs: the chain walks them as a few lone waves; few distinct bytes:
// hash chains at the 8192-hit cap.
static constexpr double kRunWeight = 3.6;      // added per byte that lies in a run of 64+ equal bytes
static constexpr double kFewWeight = 1.3;      // added per byte in a stretch of at most 4 distinct byte values (not a run)

double MasterBlockCost(const unsigned char* in, size_t begin, size_t end) {
  if (end <= begin) return 0.0;
  size_t probes = 0, runs = 0, few = 0;
  for (size_t i = begin; i + 64 <= end; i += 1024, ++probes) {
    const unsigned char c0 = in[i];
    size_t k = 1;
    while (k < 64 && in[i + k] == c0) ++k;
    if (k == 64) { ++runs; continue; }
    uint64_t seen[4] = {0, 0, 0, 0};
    unsigned distinct = 0;
    for (size_t j = 0; j < 64 && distinct <= 4; ++j) {
      const unsigned char c = in[i + j];
      const uint64_t bit = 1ull << (c & 63);
      if (!(seen[c >> 6] & bit)) { seen[c >> 6] |= bit; ++distinct; }
    }
    if (distinct <= 4) ++few;
  }
  const double n = static_cast<double>(end - begin) / 1e6;
  if (probes == 0) return n;
  return n * (1.0 + kRunWeight * static_cast<double>(runs) / static_cast<double>(probes) +
              kFewWeight * static_cast<double>(few) / static_cast<double>(probes));
}

void DealByCost(const std::vector<double>& cost, size_t shards, std::vector<size_t>* first) {
  const size_t n = cost.size();
  if (shards == 0) shards = 1;
  first->assign(shards + 1, n);
  (*first)[0] = 0;
  double total = 0;
  for (double c : cost) total += c;
  // shard s ends where the running cost comes closest to its share of the total, but leaves a block for every shard
  // still to come (and takes at least one itself) while there are blocks to go round
  if (n <= shards) {      // a block each while they last; the shards at the end are empty
    for (size_t s = 0; s <= shards; ++s) (*first)[s] = s < n ? s : n;
    return;
  }
  size_t at = 0;
  double acc = 0;
  for (size_t s = 0; s + 1 < shards; ++s) {
    const double want = total * static_cast<double>(s + 1) / static_cast<double>(shards);
    const size_t left = shards - 1 - s;                                   // shards after this one
    const size_t hi = std::max(at, n > left ? n - left : static_cast<size_t>(0));   // ... each of which gets a block if there is one
    const size_t lo = at < hi ? at + 1 : at;                              // and so does this one
    size_t to = at;
    while (to < hi && (to < lo || acc + cost[to] * 0.5 <= want)) {        // a block goes to the side its midpoint lies on
      acc += cost[to];
      ++to;
    }
    (*first)[s + 1] = to;
    at = to;
  }
  (*first)[shards] = n;
}

}  // namespace zamd

extern "C" {

// include/zopfli_amd.h
int zmx_master_block_costs(const unsigned char* in, size_t insize, double* cost, size_t ncost) {
  const size_t kMb = 1000000;   // ZOPFLI_MASTER_BLOCK_SIZE, util.h:60
  const size_t n = insize == 0 ? 1 : (insize + kMb - 1) / kMb;
  if (ncost < n) return -1;
  for (size_t b = 0; b < n; ++b) cost[b] = zamd::MasterBlockCost(in, b * kMb, std::min(insize, (b + 1) * kMb));
  return static_cast<int>(n > 0x7fffffff ? 0x7fffffff : n);
}

int zmx_deal_master_blocks(const double* cost, size_t nblocks, size_t shards, size_t* first) {
  if (shards == 0) return -1;
  std::vector<size_t> f;
  zamd::DealByCost(std::vector<double>(cost, cost + nblocks), shards, &f);
  for (size_t s = 0; s <= shards; ++s) first[s] = f[s];
  return 0;
}

}  // extern "C"

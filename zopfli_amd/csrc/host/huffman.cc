#include "huffman.h"

#include <algorithm>
#include <cmath>
#include <vector>

namespace zamd {

namespace {

struct Leaf {
  size_t weight;
  int symbol;
};

// One chain of the boundary package-merge lattice.  `count` is the number of
// leaves consumed in this chain's list, `tail` the chain in the previous list
// it packages (pool index, -1 for none).
struct Chain {
  size_t weight;
  int count;
  int tail;
};

class BoundaryPackageMerge {
 public:
  BoundaryPackageMerge(const std::vector<Leaf>& leaves, int maxbits)
      : leaves_(leaves), nleaves_(static_cast<int>(leaves.size())), maxbits_(maxbits) {
    pool_.reserve(static_cast<size_t>(2) * maxbits * nleaves_ + 4);
    pool_.push_back({leaves_[0].weight, 1, -1});
    pool_.push_back({leaves_[1].weight, 2, -1});
    for (int i = 0; i < maxbits_; ++i) {
      look_[i][0] = 0;
      look_[i][1] = 1;
    }
  }

  // Adds one chain to list `top`, replenishing the look-ahead pairs of the
  // lower lists as packages consume them (katajainen.c:69, iteratively).
  void Advance(int top) {
    int pending[40];
    int np = 0;
    pending[np++] = top;
    while (np > 0) {
      const int list = pending[--np];
      const int used = pool_[look_[list][1]].count;
      if (list == 0 && used >= nleaves_) continue;
      const int prev_last = look_[list][1];
      const int fresh = static_cast<int>(pool_.size());
      pool_.push_back({0, 0, -1});
      look_[list][0] = prev_last;
      look_[list][1] = fresh;
      if (list == 0) {
        pool_[fresh] = {leaves_[used].weight, used + 1, -1};
        continue;
      }
      const size_t package = pool_[look_[list - 1][0]].weight + pool_[look_[list - 1][1]].weight;
      if (used < nleaves_ && package > leaves_[used].weight) {
        pool_[fresh] = {leaves_[used].weight, used + 1, pool_[prev_last].tail};
      } else {
        pool_[fresh] = {package, used, look_[list - 1][1]};
        pending[np++] = list - 1;  // both look-ahead chains of the lower list
        pending[np++] = list - 1;  // were consumed
      }
    }
  }

  // The last chain only needs its leaf count / tail (katajainen.c:107).
  void Finish() {
    const int top = maxbits_ - 1;
    const int last = look_[top][1];
    const int used = pool_[last].count;
    const size_t package = pool_[look_[top - 1][0]].weight + pool_[look_[top - 1][1]].weight;
    if (used < nleaves_ && package > leaves_[used].weight) {
      const int fresh = static_cast<int>(pool_.size());
      pool_.push_back({0, used + 1, pool_[last].tail});
      look_[top][1] = fresh;
    } else {
      pool_[last].tail = look_[top - 1][1];
    }
  }

  // Number of active leaves per list, read off the final chain (katajainen.c:140).
  void Extract(unsigned* lengths) const {
    int active[16] = {0};
    int first = 16;
    for (int c = look_[maxbits_ - 1][1]; c != -1; c = pool_[c].tail) active[--first] = pool_[c].count;
    int leaf = active[15];
    unsigned bits = 1;
    for (int slot = 15; slot >= first; --slot, ++bits) {
      for (; leaf > active[slot - 1]; --leaf) lengths[leaves_[leaf - 1].symbol] = bits;
    }
  }

 private:
  const std::vector<Leaf>& leaves_;
  const int nleaves_;
  const int maxbits_;
  std::vector<Chain> pool_;
  int look_[16][2];
};

}  // namespace

bool LengthLimitedCodeLengths(const size_t* freq, int n, int maxbits, unsigned* lengths) {
  std::vector<Leaf> leaves;
  leaves.reserve(n);
  for (int i = 0; i < n; ++i) {
    lengths[i] = 0;
    if (freq[i]) leaves.push_back({freq[i], i});
  }
  const int used = static_cast<int>(leaves.size());
  if ((1 << maxbits) < used) return false;
  if (used == 0) return true;
  if (used <= 2) {
    for (const Leaf& l : leaves) lengths[l.symbol] = 1;
    return true;
  }
  for (const Leaf& l : leaves) {
    if (l.weight >= (static_cast<size_t>(1) << (sizeof(size_t) * 8 - 9))) return false;
  }
  // lightest first, symbol index breaks ties (the reference packs the index
  // into the low 9 bits of the sort key, katajainen.c:221-229)
  std::sort(leaves.begin(), leaves.end(), [](const Leaf& a, const Leaf& b) {
    return a.weight != b.weight ? a.weight < b.weight : a.symbol < b.symbol;
  });
  if (used - 1 < maxbits) maxbits = used - 1;

  BoundaryPackageMerge bpm(leaves, maxbits);
  const int chains_needed = 2 * used - 4;  // two already exist in every list
  for (int i = 0; i + 1 < chains_needed; ++i) bpm.Advance(maxbits - 1);
  bpm.Finish();
  bpm.Extract(lengths);
  return true;
}

void LengthsToSymbols(const unsigned* lengths, size_t n, unsigned maxbits, unsigned* symbols) {
  std::vector<unsigned> per_length(maxbits + 1, 0), next(maxbits + 1, 0);
  for (size_t i = 0; i < n; ++i) per_length[lengths[i]]++;
  per_length[0] = 0;
  unsigned code = 0;
  for (unsigned bits = 1; bits <= maxbits; ++bits) {
    code = (code + per_length[bits - 1]) << 1;
    next[bits] = code;
  }
  for (size_t i = 0; i < n; ++i) symbols[i] = lengths[i] ? next[lengths[i]]++ : 0;
}

void CalculateEntropy(const size_t* count, size_t n, double* bitlengths) {
  // 1/ln(2) with the reference's 14 digits (tree.c:72); the products below
  // must round exactly like the reference's, so no FMA (-ffp-contract=off).
  static const double kInvLog2 = 1.4426950408889;
  unsigned sum = 0;  // 32-bit on purpose (tree.c:73)
  for (size_t i = 0; i < n; ++i) sum += static_cast<unsigned>(count[i]);
  const double log2sum = (sum == 0 ? std::log(static_cast<double>(n)) : std::log(static_cast<double>(sum))) * kInvLog2;
  for (size_t i = 0; i < n; ++i) {
    double b = count[i] == 0 ? log2sum : log2sum - std::log(static_cast<double>(count[i])) * kInvLog2;
    if (b < 0 && b > -1e-5) b = 0;
    bitlengths[i] = b;
  }
}

}  // namespace zamd

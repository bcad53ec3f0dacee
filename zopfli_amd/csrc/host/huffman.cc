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

constexpr int kMaxLeaves = 288;   // litlen alphabet; the callers' other alphabets are smaller
constexpr int kMaxBits = 15;

}  // namespace

// Boundary package-merge (katajainen.c:172-262) on flat arrays: the leaves on the stack, the chains in a
// per-thread array that is allocated once (the block-split search calls this ~20 times per candidate split
// point, deflate.c:569 and :277 — heap traffic was two thirds of its time).
bool LengthLimitedCodeLengths(const size_t* freq, int n, int maxbits, unsigned* lengths) {
  for (int i = 0; i < n; ++i) lengths[i] = 0;
  if (n > kMaxLeaves || maxbits > kMaxBits) return false;
  Leaf leaves[kMaxLeaves];
  int used = 0;
  for (int i = 0; i < n; ++i) {
    if (freq[i]) leaves[used++] = {freq[i], i};
  }
  if ((1 << maxbits) < used) return false;
  if (used == 0) return true;
  if (used <= 2) {
    for (int i = 0; i < used; ++i) lengths[leaves[i].symbol] = 1;
    return true;
  }
  for (int i = 0; i < used; ++i) {
    if (leaves[i].weight >= (static_cast<size_t>(1) << (sizeof(size_t) * 8 - 9))) return false;
  }
  // lightest first, symbol index breaks ties: the index rides in the low 9 bits of the sort key, as in
  // katajainen.c:221-229 (weights are below 2^55, checked above)
  {
    uint64_t key[kMaxLeaves];
    for (int i = 0; i < used; ++i) key[i] = (static_cast<uint64_t>(leaves[i].weight) << 9) | static_cast<uint64_t>(leaves[i].symbol);
    if (used <= 24) {
      for (int i = 1; i < used; ++i) {
        const uint64_t x = key[i];
        int j = i;
        for (; j > 0 && key[j - 1] > x; --j) key[j] = key[j - 1];
        key[j] = x;
      }
    } else {
      std::sort(key, key + used);
    }
    for (int i = 0; i < used; ++i) leaves[i] = {static_cast<size_t>(key[i] >> 9), static_cast<int>(key[i] & 511)};
  }
  if (used - 1 < maxbits) maxbits = used - 1;

  // two chains to start with, then at most one per list for each of the 2 * used - 4 steps
  static thread_local std::vector<Chain> pool_store;
  const size_t need = static_cast<size_t>(2) * maxbits * used + 8;
  if (pool_store.size() < need) pool_store.resize(std::max(need, static_cast<size_t>(2) * kMaxBits * kMaxLeaves + 8));
  Chain* const pool = pool_store.data();
  int np = 0;
  pool[np++] = {leaves[0].weight, 1, -1};
  pool[np++] = {leaves[1].weight, 2, -1};
  int look[kMaxBits][2];   // the two look-ahead chains of every list
  for (int i = 0; i < maxbits; ++i) {
    look[i][0] = 0;
    look[i][1] = 1;
  }
  const int top = maxbits - 1;
  // Adds one chain to the top list, replenishing the look-ahead pairs of the lower lists as packages
  // consume them (katajainen.c:69, iteratively); the last chain only needs its leaf count / tail (:107).
  for (int step = 0; step + 1 < 2 * used - 4; ++step) {
    int pending[2 * kMaxBits + 2];
    int npend = 0;
    pending[npend++] = top;
    while (npend > 0) {
      const int list = pending[--npend];
      const int prev_last = look[list][1];
      const int cnt = pool[prev_last].count;
      if (list == 0 && cnt >= used) continue;
      const int fresh = np++;
      look[list][0] = prev_last;
      look[list][1] = fresh;
      if (list == 0) {
        pool[fresh] = {leaves[cnt].weight, cnt + 1, -1};
        continue;
      }
      const size_t package = pool[look[list - 1][0]].weight + pool[look[list - 1][1]].weight;
      if (cnt < used && package > leaves[cnt].weight) {
        pool[fresh] = {leaves[cnt].weight, cnt + 1, pool[prev_last].tail};
      } else {
        pool[fresh] = {package, cnt, look[list - 1][1]};
        pending[npend++] = list - 1;  // both look-ahead chains of the lower list
        pending[npend++] = list - 1;  // were consumed
      }
    }
  }
  {
    const int last = look[top][1];
    const int cnt = pool[last].count;
    const size_t package = pool[look[top - 1][0]].weight + pool[look[top - 1][1]].weight;
    if (cnt < used && package > leaves[cnt].weight) {
      const int fresh = np++;
      pool[fresh] = {0, cnt + 1, pool[last].tail};
      look[top][1] = fresh;
    } else {
      pool[last].tail = look[top - 1][1];
    }
  }
  // number of active leaves per list, read off the final chain (katajainen.c:140)
  int active[17] = {0};
  int first = 16;
  for (int c = look[top][1]; c != -1; c = pool[c].tail) active[--first] = pool[c].count;
  int leaf = active[15];
  unsigned bits = 1;
  for (int slot = 15; slot >= first; --slot, ++bits) {
    for (; leaf > active[slot - 1]; --leaf) lengths[leaves[leaf - 1].symbol] = bits;
  }
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

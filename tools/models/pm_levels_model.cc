// Model (CPU) of the package-merge formulation the device's block-cost kernel uses (zmx_blockcost.h): the lists of
// katajainen.c's boundary package-merge computed level by level as MERGES of the sorted leaves with the pair sums of the
// list below (a package goes before a leaf of equal weight: katajainen.c:85 takes the leaf only if the sum is GREATER), the
// code lengths read off by walking down from the first 2n - 2 items of the top list.  Checked here against the product's
// host implementation (huffman.cc, itself pinned to the reference) on random and tie-heavy histograms.
//   g++ -O2 -std=c++17 -I zopfli_amd/csrc/host tools/models/pm_levels_model.cc zopfli_amd/csrc/host/huffman.cc -o /tmp/pm_model && /tmp/pm_model
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <vector>
#include "huffman.h"

static bool PmLevels(const size_t* freq, int n, int maxbits, unsigned* lengths) {
  for (int i = 0; i < n; ++i) lengths[i] = 0;
  std::vector<uint32_t> key;
  for (int i = 0; i < n; ++i) if (freq[i]) key.push_back((uint32_t)(freq[i] << 9) | (uint32_t)i);
  const int used = (int)key.size();
  if ((1 << maxbits) < used) return false;
  if (used == 0) return true;
  if (used <= 2) { for (auto k : key) lengths[k & 511] = 1; return true; }
  std::sort(key.begin(), key.end());
  std::vector<uint32_t> W(used);
  for (int i = 0; i < used; ++i) W[i] = key[i] >> 9;
  const int L = std::min(maxbits, used - 1);
  std::vector<std::vector<uint32_t>> list(L);
  std::vector<std::vector<char>> isleaf(L);
  list[0] = W;
  isleaf[0].assign(used, 1);
  for (int j = 1; j < L; ++j) {
    const auto& prev = list[j - 1];
    const int m = (int)prev.size() / 2;
    std::vector<uint32_t> P(m);
    for (int k = 0; k < m; ++k) P[k] = prev[2 * k] + prev[2 * k + 1];
    list[j].assign(used + m, 0);
    isleaf[j].assign(used + m, 0);
    for (int i = 0; i < used; ++i) {      // a leaf goes behind every package that is not heavier
      const int np = (int)(std::upper_bound(P.begin(), P.end(), W[i]) - P.begin());
      list[j][i + np] = W[i];
      isleaf[j][i + np] = 1;
    }
    for (int k = 0; k < m; ++k) {         // a package goes behind every leaf that is lighter
      const int nl = (int)(std::lower_bound(W.begin(), W.end(), P[k]) - W.begin());
      list[j][k + nl] = P[k];
    }
  }
  std::vector<int> a(L, 0);
  int c = 2 * used - 2;
  for (int j = L - 1; j >= 0; --j) {
    if (c > (int)list[j].size()) { fprintf(stderr, "list %d too short: %d of %zu\n", j, c, list[j].size()); return false; }
    int leaves = 0;
    for (int x = 0; x < c; ++x) leaves += isleaf[j][x];
    a[j] = leaves;
    c = 2 * (c - leaves);
  }
  for (int i = 0; i < used; ++i) {
    unsigned len = 0;
    for (int j = 0; j < L; ++j) len += i < a[j] ? 1u : 0u;
    lengths[key[i] & 511] = len;
  }
  return true;
}

static uint64_t rng = 88172645463325252ull;
static uint32_t rnd() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return (uint32_t)(rng >> 11); }

int main() {
  long cases = 0, bad = 0;
  for (int it = 0; it < 300000; ++it) {
    const int kind = rnd() % 8;
    int n, maxbits;
    switch (rnd() % 4) { case 0: n = 288; maxbits = 15; break; case 1: n = 32; maxbits = 15; break; case 2: n = 19; maxbits = 7; break; default: n = 3 + rnd() % 286; maxbits = 7 + rnd() % 9; }
    size_t f[288];
    for (int i = 0; i < n; ++i) {
      uint32_t v = 0;
      switch (kind) {
        case 0: v = rnd() % 1000; break;
        case 1: v = rnd() % 4; break;                         // ties everywhere
        case 2: v = (rnd() % 3 == 0) ? 0 : 1 + rnd() % 3; break;
        case 3: v = 1u << (rnd() % 20); break;                // powers of two: package sums tie with leaves
        case 4: v = (rnd() % 10 == 0) ? rnd() % 1000000 : rnd() % 3; break;
        case 5: v = i < 5 ? 100000 + rnd() % 100 : rnd() % 2; break;
        case 6: { uint32_t a = 1, b = 1; for (int k = 0; k < (int)(rnd() % 24); ++k) { uint32_t t = a + b; a = b; b = t; } v = a; break; }   // Fibonacci: the deepest trees
        default: v = 3900 + rnd() % 16; break;                // random data: 256 counts of about the same size
      }
      f[i] = v;
    }
    if ((1 << maxbits) < n) continue;
    unsigned l0[288], l1[288];
    const bool ok0 = zamd::LengthLimitedCodeLengths(f, n, maxbits, l0);
    const bool ok1 = PmLevels(f, n, maxbits, l1);
    ++cases;
    if (ok0 != ok1 || memcmp(l0, l1, sizeof(unsigned) * n) != 0) {
      if (++bad < 5) {
        fprintf(stderr, "MISMATCH kind %d n %d maxbits %d ok %d %d\n", kind, n, maxbits, ok0, ok1);
        for (int i = 0; i < n; ++i) if (l0[i] != l1[i]) fprintf(stderr, "  sym %d freq %zu: %u vs %u\n", i, f[i], l0[i], l1[i]);
      }
    }
  }
  printf("%ld cases, %ld mismatches\n", cases, bad);
  return bad != 0;
}

// Model (CPU) of the wave-parallel formulation of OptimizeHuffmanForRle (deflate.c:413-491) that zmx_blockcost.h uses
// (bc_smooth_wave): frozen runs from run lengths, next0[b] = the next break after a break at b, the visited breaks = the
// chain from `first`, every visited stretch collapsed from ORIGINAL counts — against the host's serial transcription
// (block_cost.cc: OptimizeCountsForRle, itself pinned to the reference).
//   g++ -O2 -std=c++17 -I zopfli_amd/csrc/host -I include tools/models/smooth_model.cc zopfli_amd/csrc/host/block_cost.cc zopfli_amd/csrc/host/huffman.cc -o /tmp/smooth_model && /tmp/smooth_model
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>
#include "block_cost.h"

static void SmoothModel(int n, const size_t* src, size_t* dst) {
  int length = n;
  while (length > 0 && src[length - 1] == 0) --length;
  for (int i = 0; i < n; ++i) dst[i] = src[i];
  if (length == 0) return;
  std::vector<char> frozen(length + 1, 0);
  for (int i = 0; i < length; ++i) {
    int s = i, e = i + 1;
    while (s > 0 && src[s - 1] == src[i]) --s;
    while (e < length && src[e] == src[i]) ++e;
    const int run = e - s;
    frozen[i] = (src[i] == 0 && run >= 5) || (src[i] != 0 && run >= 7);
  }
  frozen[length] = 1;
  std::vector<size_t> psum(length + 1, 0);
  for (int i = 0; i < length; ++i) psum[i + 1] = psum[i] + src[i];
  auto absd = [](size_t a, size_t b) { return a > b ? a - b : b - a; };
  std::vector<int> next0(length + 1, length);
  for (int b = 0; b < length; ++b) {
    const size_t limit = b + 3 < length ? (src[b] + src[b + 1] + src[b + 2] + src[b + 3] + 2) / 4 : src[b];
    int i = b + 1;
    while (i < length && !frozen[i] && absd(src[i], limit) < 4) ++i;
    next0[b] = i;
  }
  int first = 0;
  if (!frozen[0]) {
    const size_t limit = src[0];
    first = 1;
    while (first < length && !frozen[first] && absd(src[first], limit) < 4) ++first;
  }
  auto collapse = [&](int b, int nb) {
    const size_t stride = nb - b, sum = psum[nb] - psum[b];
    if (stride >= 4 || (stride >= 3 && sum == 0)) {
      size_t mean = (sum + stride / 2) / stride;
      if (mean < 1) mean = 1;
      if (sum == 0) mean = 0;
      for (int k = b; k < nb; ++k) dst[k] = mean;
    }
  };
  if (first > 0) collapse(0, first);
  for (int x = first; x < length; x = next0[x]) collapse(x, next0[x]);
}

static uint64_t rng = 0x9e3779b97f4a7c15ull;
static uint32_t rnd() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return (uint32_t)(rng >> 11); }

int main() {
  long cases = 0, bad = 0;
  for (int it = 0; it < 2000000; ++it) {
    const int n = (rnd() % 3 == 0) ? 32 : (rnd() % 3 == 0) ? 1 + rnd() % 288 : 288;
    size_t a[288], want[288], got[288];
    const int kind = rnd() % 8;
    size_t level = rnd() % 1000;
    for (int i = 0; i < n; ++i) {
      uint32_t v;
      switch (kind) {
        case 0: v = rnd() % 6; break;
        case 1: v = (rnd() % 4 == 0) ? rnd() % 20 : 0; break;
        case 2: if (rnd() % 9 == 0) level = rnd() % 1000; v = (uint32_t)level + rnd() % 5; break;
        case 3: v = rnd() % 3 == 0 ? 0 : 1 + rnd() % 2; break;
        case 4: if (rnd() % 13 == 0) level = rnd() % 50; v = (uint32_t)level; break;       // long equal runs
        case 5: v = 3900 + rnd() % 120; break;
        case 6: v = (i % 17 < 6) ? 0 : rnd() % 9; break;
        default: v = rnd() % 4 == 0 ? rnd() % 100000 : rnd() % 8; break;
      }
      a[i] = v;
    }
    if (rnd() % 4 == 0) for (int i = n - 1 - (int)(rnd() % 20); i < n; ++i) if (i >= 0) a[i] = 0;
    memcpy(want, a, sizeof(size_t) * n);
    zamd::OptimizeCountsForRle(n, want);
    SmoothModel(n, a, got);
    ++cases;
    if (memcmp(want, got, sizeof(size_t) * n) != 0) {
      if (++bad < 4) {
        fprintf(stderr, "MISMATCH kind %d n %d\n", kind, n);
        for (int i = 0; i < n; ++i) fprintf(stderr, "%s%zu:%zu/%zu", i ? " " : "", a[i], want[i], got[i]);
        fprintf(stderr, "\n");
      }
    }
  }
  printf("%ld cases, %ld mismatches\n", cases, bad);
  return bad != 0;
}

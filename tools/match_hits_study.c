/* How even is k_match2's work?  (CPU only; test infrastructure: builds on the oracle.)
 *
 * Counts, for every position of a block of synthetic data, the candidates ZopfliFindLongestMatch visits
 * (lz77.c:464-530: what k_match2 calls a hit), then schedules the positions of each tile of MT positions on
 * 512 lanes the way the kernel hands them out (next position to the next free lane) and reports how much of
 * the lanes' time is work: a tile ends when its last lane does, so one position with thousands of hits
 * leaves 511 lanes waiting.
 *
 *   gcc -O2 -o /tmp/match_hits_study tools/match_hits_study.c zopfli_amd/csrc/tools/datagen.c && /tmp/match_hits_study T 4000000
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../oracle/zopfli_oracle.c"

int zopfli_amd_datagen(char cls, unsigned long long seed, unsigned char* out, size_t n);

static unsigned walk_hits(const zo_table* t, size_t pos) {
  const unsigned char* in = t->in;
  const size_t size = t->inend;
  unsigned limit = ZO_MAX_MATCH, bestlength = 1, dist, hits = 0;
  int chain = 1, hits_left = ZO_MAX_CHAIN_HITS;
  size_t cand;
  if (size - pos < ZO_MIN_MATCH) return 0;
  if (pos + limit > size) limit = (unsigned)(size - pos);
  dist = t->prev1[pos - t->ws];
  if (dist == 0) dist = ZO_WINDOW;
  cand = pos - (dist < ZO_WINDOW ? dist : 0);
  while (dist < ZO_WINDOW) {
    unsigned cur = 0, step;
    ++hits;
    if (pos + bestlength >= size || in[pos + bestlength] == in[cand + bestlength]) {
      while (cur < limit && in[pos + cur] == in[cand + cur]) cur++;
    }
    if (cur > bestlength) {
      bestlength = cur;
      if (cur >= limit) break;
    }
    if (chain == 1 && bestlength >= t->same[pos - t->ws] && zo_val2(t, pos) == zo_val2(t, cand)) chain = 2;
    step = chain == 1 ? t->prev1[cand - t->ws] : t->prev2[cand - t->ws];
    if (step == 0) break;
    cand -= step;
    dist += step;
    if (--hits_left <= 0) break;
  }
  return hits;
}

static int cmp_u32(const void* a, const void* b) { return (*(const unsigned*)a > *(const unsigned*)b) - (*(const unsigned*)a < *(const unsigned*)b); }

int main(int argc, char** argv) {
  const char cls = argc > 1 ? argv[1][0] : 'T';
  const size_t n = argc > 2 ? (size_t)atol(argv[2]) : 2000000;
  const size_t MB = 1000000;
  unsigned char* in = (unsigned char*)malloc(n);
  unsigned* hits = (unsigned*)malloc(sizeof(unsigned) * n);
  size_t b, i;
  double total = 0;
  zopfli_amd_datagen(cls, 1, in, n);
  for (b = 0; b < n; b += MB) {   /* master blocks, as ZopfliDeflate cuts them */
    const size_t e = b + MB < n ? b + MB : n;
    zo_table* t = (zo_table*)calloc(1, sizeof(zo_table));
    t->in = in; t->instart = b; t->inend = e; t->ws = b > ZO_WINDOW ? b - ZO_WINDOW : 0;
    t->same = (unsigned short*)malloc(sizeof(unsigned short) * (e - t->ws + 1));
    t->prev1 = (unsigned short*)malloc(sizeof(unsigned short) * (e - t->ws + 1));
    t->prev2 = (unsigned short*)malloc(sizeof(unsigned short) * (e - t->ws + 1));
    zo_build_static(t);
    for (i = b; i < e; i++) { hits[i] = walk_hits(t, i); total += hits[i]; }
    free(t->same); free(t->prev1); free(t->prev2); free(t);
  }
  {
    unsigned* s = (unsigned*)malloc(sizeof(unsigned) * n);
    memcpy(s, hits, sizeof(unsigned) * n);
    qsort(s, n, sizeof(unsigned), cmp_u32);
    printf("class %c, %zu positions: %.1f hits per position; median %u, p90 %u, p99 %u, p99.9 %u, max %u\n", cls, n, total / n,
           s[n / 2], s[n * 9 / 10], s[n * 99 / 100], s[(size_t)(n * 0.999)], s[n - 1]);
    free(s);
  }
  {
    const unsigned tiles[] = {2048, 4096, 8192, 16384, 65536};
    const unsigned lanes = 512, c0 = 3;   /* iterations a lane spends on a position besides its hits */
    size_t k;
    for (k = 0; k < sizeof(tiles) / sizeof(tiles[0]); k++) {
      const unsigned MT = tiles[k];
      double busy = 0, span = 0;
      size_t t0;
      for (b = 0; b < n; b += MB) {
        const size_t e = b + MB < n ? b + MB : n;
        for (t0 = b; t0 < e; t0 += MT) {
          const size_t t1 = t0 + MT < e ? t0 + MT : e;
          unsigned long long lane_end[512];
          unsigned long long mk = 0;
          memset(lane_end, 0, sizeof(lane_end));
          for (i = t0; i < t1; i++) {   /* next position to the lane that frees first */
            unsigned l, best = 0;
            for (l = 1; l < lanes; l++) if (lane_end[l] < lane_end[best]) best = l;
            lane_end[best] += hits[i] + c0;
            busy += hits[i] + c0;
          }
          for (i = 0; i < lanes; i++) if (lane_end[i] > mk) mk = lane_end[i];
          span += (double)mk * lanes;
        }
      }
      printf("  tile %6u positions: lanes busy %.1f %% of the time (a tile ends with its last lane)\n", MT, 100.0 * busy / span);
    }
  }
  return 0;
}

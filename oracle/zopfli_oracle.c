/*
 * TEST INFRASTRUCTURE — CPU restatement of the zopfli hot path, see zopfli_oracle.h.
 *
 * Citations are to /root/reference/src/zopfli/.  The restatement follows
 * the hash tables are replaced by static per-position
 * arrays (pure functions of the data and the block end), the longest-match
 * cache by a per-position match record, and FollowPath's re-query by a lookup
 * in that record.  It is deliberately scalar and simple.
 */
#include "zopfli_oracle.h"

#include <stdlib.h>
#include <string.h>

#define ZO_WINDOW 32768u
#define ZO_MAX_MATCH 258u
#define ZO_MIN_MATCH 3u
#define ZO_MAX_CHAIN_HITS 8192 /* util.h:89 */
#define ZO_LARGE_FLOAT 1e30    /* util.h:65 */

struct zo_table {
  const unsigned char* in;
  size_t instart, inend, ws; /* ws = windowstart = max(0, instart - 32768) */
  unsigned short* same;      /* [inend - ws] */
  unsigned short* prev1;     /* distance to the previous position of the same hash, 0 = none */
  unsigned short* prev2;
  /* match record per block position i - instart */
  unsigned short* length;
  unsigned short* dist;
  size_t* cp_off;            /* [B + 1] offsets into cp_len / cp_dist */
  unsigned short* cp_len;    /* change points of sublen: sublen[l] = cp_dist[k] for cp_len[k-1] < l <= cp_len[k] */
  unsigned short* cp_dist;
  size_t cp_cap, cp_n;
};

/* hash.c:96-98,107-108,139-143: three rolling updates, zero for bytes at or past `end`. */
static unsigned zo_val(const unsigned char* in, size_t p, size_t end) {
  unsigned b0 = in[p];
  unsigned b1 = p + 1 < end ? in[p + 1] : 0;
  unsigned b2 = p + 2 < end ? in[p + 2] : 0;
  return (((b0 << 10) ^ (b1 << 5) ^ b2) & 32767u);
}

/* hash.c:129 */
static unsigned zo_val2(const zo_table* t, size_t p) {
  int same = t->same[p - t->ws];
  return (unsigned)((same - (int)ZO_MIN_MATCH) & 255) ^ zo_val(t->in, p, t->inend);
}

static void zo_build_static(zo_table* t) {
  const unsigned char* in = t->in;
  const size_t ws = t->ws, end = t->inend, n = end - ws;
  long long* head1 = (long long*)malloc(sizeof(long long) * 32768);
  long long* head2 = (long long*)malloc(sizeof(long long) * 32768);
  size_t k, p;
  /* same[p]: number of following bytes equal to in[p], inside the block, capped
   * at 65535 (hash.c:116-126).  Backward recurrence; the cap is absorbing. */
  for (k = n; k-- > 0;) {
    p = ws + k;
    if (p + 1 < end && in[p + 1] == in[p]) {
      unsigned next = t->same[k + 1];
      t->same[k] = (unsigned short)(next < 65535u ? next + 1 : 65535u);
    } else {
      t->same[k] = 0;
    }
  }
  /* prev links: most recent earlier position with the same hash value, if it
   * is 1..32767 back (hash.c:110-114, 131-135; at exactly 32768 the window
   * slot aliases and the reference stores a self-loop = none). */
  for (k = 0; k < 32768; k++) head1[k] = head2[k] = -1;
  for (k = 0; k < n; k++) {
    unsigned v, v2;
    p = ws + k;
    v = zo_val(in, p, end);
    v2 = zo_val2(t, p);
    t->prev1[k] = (unsigned short)((head1[v] >= 0 && p - (size_t)head1[v] < ZO_WINDOW) ? p - (size_t)head1[v] : 0);
    head1[v] = (long long)p;
    t->prev2[k] = (unsigned short)((head2[v2] >= 0 && p - (size_t)head2[v2] < ZO_WINDOW) ? p - (size_t)head2[v2] : 0);
    head2[v2] = (long long)p;
  }
  free(head1);
  free(head2);
}

static void zo_push_cp(zo_table* t, unsigned short len, unsigned short dist) {
  if (t->cp_n == t->cp_cap) {
    t->cp_cap = t->cp_cap ? t->cp_cap * 2 : 1024;
    t->cp_len = (unsigned short*)realloc(t->cp_len, t->cp_cap * sizeof(unsigned short));
    t->cp_dist = (unsigned short*)realloc(t->cp_dist, t->cp_cap * sizeof(unsigned short));
  }
  t->cp_len[t->cp_n] = len;
  t->cp_dist[t->cp_n] = dist;
  t->cp_n++;
}

/* The chain walk of ZopfliFindLongestMatch (lz77.c:436-541) with limit 258 and
 * sublen requested, on the static arrays.  Appends the change points. */
static void zo_walk(zo_table* t, size_t pos, unsigned short* out_len, unsigned short* out_dist) {
  const unsigned char* in = t->in;
  const size_t size = t->inend;
  unsigned limit = ZO_MAX_MATCH, bestlength = 1, bestdist = 0, dist;
  int chain = 1, hits_left = ZO_MAX_CHAIN_HITS;
  size_t cand;

  if (size - pos < ZO_MIN_MATCH) { /* lz77.c:440-446 */
    *out_len = 0;
    *out_dist = 0;
    return;
  }
  if (pos + limit > size) limit = (unsigned)(size - pos); /* lz77.c:448-450 */

  dist = t->prev1[pos - t->ws];
  if (dist == 0) dist = ZO_WINDOW; /* no candidate: the loop below does not run */
  cand = pos - (dist < ZO_WINDOW ? dist : 0);
  while (dist < ZO_WINDOW) { /* lz77.c:464 */
    unsigned cur = 0, step;
    /* lz77.c:478-493: cheap reject on the byte after the current best, then the
     * common prefix length capped at limit (the same[] skip is an acceleration) */
    if (pos + bestlength >= size || in[pos + bestlength] == in[cand + bestlength]) {
      while (cur < limit && in[pos + cur] == in[cand + cur]) cur++;
    }
    if (cur > bestlength) { /* lz77.c:495-505 */
      zo_push_cp(t, (unsigned short)cur, (unsigned short)dist);
      bestdist = dist;
      bestlength = cur;
      if (cur >= limit) break;
    }
    /* lz77.c:509-519: switch to the second hash once the best length covers the run */
    if (chain == 1 && bestlength >= t->same[pos - t->ws] && zo_val2(t, pos) == zo_val2(t, cand)) chain = 2;
    step = chain == 1 ? t->prev1[cand - t->ws] : t->prev2[cand - t->ws];
    if (step == 0) break; /* lz77.c:521-523 */
    cand -= step;
    dist += step;
    if (--hits_left <= 0) break; /* lz77.c:527-530 */
  }
  *out_len = (unsigned short)bestlength;
  *out_dist = (unsigned short)bestdist;
}

zo_table* zo_table_build(const unsigned char* in, size_t instart, size_t inend) {
  zo_table* t = (zo_table*)calloc(1, sizeof(zo_table));
  const size_t B = inend - instart;
  size_t n, i;
  t->in = in;
  t->instart = instart;
  t->inend = inend;
  t->ws = instart > ZO_WINDOW ? instart - ZO_WINDOW : 0;
  n = inend - t->ws;
  t->same = (unsigned short*)malloc(sizeof(unsigned short) * (n + 1));
  t->prev1 = (unsigned short*)malloc(sizeof(unsigned short) * (n + 1));
  t->prev2 = (unsigned short*)malloc(sizeof(unsigned short) * (n + 1));
  t->length = (unsigned short*)malloc(sizeof(unsigned short) * (B + 1));
  t->dist = (unsigned short*)malloc(sizeof(unsigned short) * (B + 1));
  t->cp_off = (size_t*)malloc(sizeof(size_t) * (B + 2));
  zo_build_static(t);
  for (i = 0; i < B; i++) {
    t->cp_off[i] = t->cp_n;
    zo_walk(t, instart + i, &t->length[i], &t->dist[i]);
  }
  t->cp_off[B] = t->cp_n;
  return t;
}

void zo_table_free(zo_table* t) {
  if (!t) return;
  free(t->same); free(t->prev1); free(t->prev2);
  free(t->length); free(t->dist); free(t->cp_off);
  free(t->cp_len); free(t->cp_dist);
  free(t);
}

unsigned short zo_same(const zo_table* t, size_t p) { return t->same[p - t->ws]; }
unsigned short zo_prev1(const zo_table* t, size_t p) { return t->prev1[p - t->ws]; }
unsigned short zo_prev2(const zo_table* t, size_t p) { return t->prev2[p - t->ws]; }

void zo_find_longest_match(const zo_table* t, size_t pos, unsigned short* sublen,
                           unsigned short* distance, unsigned short* length) {
  const size_t i = pos - t->instart;
  *length = t->length[i];
  *distance = t->dist[i];
  if (sublen) {
    size_t k;
    unsigned l = ZO_MIN_MATCH - 2; /* bestlength starts at 1: first change point fills 2.. */
    for (k = t->cp_off[i]; k < t->cp_off[i + 1]; k++) {
      for (l = l + 1; l <= t->cp_len[k]; l++) sublen[l] = t->cp_dist[k];
      l = t->cp_len[k];
    }
  }
}

/* distance used to reach `len` at block position i = sublen[len] */
static unsigned short zo_dist_for(const zo_table* t, size_t i, unsigned len) {
  size_t k;
  for (k = t->cp_off[i]; k < t->cp_off[i + 1]; k++) {
    if (t->cp_len[k] >= len) return t->cp_dist[k];
  }
  return 0;
}

/* lz77.c:265-271 */
static int zo_length_score(int length, int distance) { return distance > 1024 ? length - 1 : length; }

size_t zo_greedy(const zo_table* t, unsigned short* litlens, unsigned short* dists) {
  const unsigned char* in = t->in;
  size_t i, j, n = 0;
  unsigned prev_length = 0, prev_match = 0;
  int match_available = 0;
  for (i = t->instart; i < t->inend; i++) { /* lz77.c:571 */
    unsigned leng = t->length[i - t->instart], dist = t->dist[i - t->instart];
    int lengthscore = zo_length_score((int)leng, (int)dist);
    int prevlengthscore = zo_length_score((int)prev_length, (int)prev_match);
    if (match_available) { /* lz77.c:581-607 */
      match_available = 0;
      if (lengthscore > prevlengthscore + 1) {
        litlens[n] = in[i - 1]; dists[n] = 0; n++;
        if (lengthscore >= (int)ZO_MIN_MATCH && leng < ZO_MAX_MATCH) {
          match_available = 1;
          prev_length = leng;
          prev_match = dist;
          continue;
        }
      } else {
        leng = prev_length;
        dist = prev_match;
        litlens[n] = (unsigned short)leng; dists[n] = (unsigned short)dist; n++;
        for (j = 2; j < leng; j++) i++;
        continue;
      }
    } else if (lengthscore >= (int)ZO_MIN_MATCH && leng < ZO_MAX_MATCH) { /* lz77.c:608-613 */
      match_available = 1;
      prev_length = leng;
      prev_match = dist;
      continue;
    }
    if (lengthscore >= (int)ZO_MIN_MATCH) { /* lz77.c:618-624 */
      litlens[n] = (unsigned short)leng; dists[n] = (unsigned short)dist; n++;
    } else {
      leng = 1;
      litlens[n] = in[i]; dists[n] = 0; n++;
    }
    for (j = 1; j < leng; j++) i++;
  }
  return n;
}

/* RFC 1951 tables, restated (symbols.h:38-237) */
static int zo_dist_extra_bits(unsigned d) {
  int l = 0;
  if (d < 5) return 0;
  d -= 1;
  while (d >> (l + 1)) l++;
  return l - 1;
}
static int zo_dist_symbol(unsigned d) {
  int l = 0;
  unsigned e;
  if (d < 5) return (int)d - 1;
  e = d - 1;
  while (e >> (l + 1)) l++;
  return 2 * l + (int)((e >> (l - 1)) & 1);
}
static int zo_length_symbol(unsigned l) {
  static const unsigned short base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51,
                                          59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
  int s = 28;
  while (base[s] > l) s--;
  return 257 + s;
}
static int zo_length_extra_bits(unsigned l) {
  int s = zo_length_symbol(l) - 257;
  return (s < 8 || s == 28) ? 0 : (s - 4) / 4;
}

/* GetCostStat for a match: int + int, then two double adds left to right (squeeze.c:155) */
static double zo_match_cost(const double* ll, const double* d, unsigned len, unsigned dist) {
  return zo_length_extra_bits(len) + zo_dist_extra_bits(dist) + ll[zo_length_symbol(len)] + d[zo_dist_symbol(dist)];
}

double zo_get_best_lengths(const zo_table* t, const double* ll, const double* d, double mincost,
                           unsigned short* length_array) {
  const unsigned char* in = t->in;
  const size_t instart = t->instart, inend = t->inend, blocksize = inend - instart;
  float* costs;
  size_t i, k, kend;
  double result;
  if (instart == inend) return 0;
  costs = (float*)malloc(sizeof(float) * (blocksize + 1));
  for (i = 1; i < blocksize + 1; i++) costs[i] = ZO_LARGE_FLOAT; /* squeeze.c:243 */
  costs[0] = 0;
  length_array[0] = 0;
  for (i = instart; i < inend; i++) {
    size_t j = i - instart;
    unsigned leng;
    double mincostaddcostj;
    /* squeeze.c:251-271 long-run shortcut */
    if (t->same[i - t->ws] > ZO_MAX_MATCH * 2 && i > instart + ZO_MAX_MATCH + 1 &&
        i + ZO_MAX_MATCH * 2 + 1 < inend && t->same[i - ZO_MAX_MATCH - t->ws] > ZO_MAX_MATCH) {
      double symbolcost = zo_match_cost(ll, d, ZO_MAX_MATCH, 1);
      for (k = 0; k < ZO_MAX_MATCH; k++) {
        costs[j + ZO_MAX_MATCH] = (float)(costs[j] + symbolcost);
        length_array[j + ZO_MAX_MATCH] = ZO_MAX_MATCH;
        i++;
        j++;
      }
    }
    leng = t->length[j];
    /* squeeze.c:277-284 literal */
    if (i + 1 <= inend) {
      double newCost = ll[in[i]] + costs[j];
      if (newCost < costs[j + 1]) {
        costs[j + 1] = (float)newCost;
        length_array[j + 1] = 1;
      }
    }
    /* squeeze.c:286-302 lengths */
    kend = leng < inend - i ? leng : inend - i;
    mincostaddcostj = mincost + costs[j];
    {
      size_t cp = t->cp_off[j];
      for (k = 3; k <= kend; k++) {
        double newCost;
        while (t->cp_len[cp] < k) cp++; /* sublen[k] */
        if (costs[j + k] <= mincostaddcostj) continue;
        newCost = zo_match_cost(ll, d, (unsigned)k, t->cp_dist[cp]) + costs[j];
        if (newCost < costs[j + k]) {
          costs[j + k] = (float)newCost;
          length_array[j + k] = (unsigned short)k;
        }
      }
    }
  }
  result = costs[blocksize];
  free(costs);
  return result;
}

size_t zo_trace_follow(const zo_table* t, const unsigned short* length_array,
                       unsigned short* litlens, unsigned short* dists) {
  const size_t size = t->inend - t->instart;
  size_t index = size, npath = 0, n = 0, a, pos;
  unsigned short* path;
  if (size == 0) return 0;
  path = (unsigned short*)malloc(sizeof(unsigned short) * size);
  for (;;) { /* squeeze.c:320-328 */
    path[npath++] = length_array[index];
    index -= length_array[index];
    if (index == 0) break;
  }
  pos = t->instart;
  for (a = npath; a-- > 0;) { /* squeeze.c:357-388, forward over the mirrored path */
    unsigned length = path[a];
    if (length >= ZO_MIN_MATCH) {
      litlens[n] = (unsigned short)length;
      dists[n] = zo_dist_for(t, pos - t->instart, length);
      n++;
    } else {
      length = 1;
      litlens[n] = t->in[pos];
      dists[n] = 0;
      n++;
    }
    pos += length;
  }
  free(path);
  return n;
}

void zo_histogram(const unsigned short* litlens, const unsigned short* dists, size_t n, unsigned* hist320) {
  size_t i;
  memset(hist320, 0, sizeof(unsigned) * 320);
  for (i = 0; i < n; i++) {
    if (dists[i] == 0) {
      hist320[litlens[i]]++;
    } else {
      hist320[zo_length_symbol(litlens[i])]++;
      hist320[288 + zo_dist_symbol(dists[i])]++;
    }
  }
}

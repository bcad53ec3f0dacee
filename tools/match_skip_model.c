/* CPU model of the exact skip-walk match finder (k_match5) — test infrastructure, builds on the oracle.
 *
 * ZopfliFindLongestMatch (lz77.c:407-542) visits every entry of the hash chain of a position, newest first, but
 * its result — the change points of sublen — depends only on the visited candidates whose common prefix with the
 * position is LONGER than the best so far (lz77.c:494-505), on where the walk changes to the second hash
 * (:509-519) and on which candidate is the 8192nd (:527-530).  The model keeps, beside the reference's two chains,
 * "level" chains (nearest earlier position whose first k bytes hash alike, k = 4, 8, 16 ...), walks the level
 * k <= bestlength + 1 (every candidate that can beat bestlength shares bestlength + 1 bytes with the position,
 * hence its level-k hash), and counts the reference's hits it jumped over from per-position ranks within the two
 * reference chains.  It prints, per class, whether every record equals the oracle's walk and how many chain
 * entries each walk touches.
 *
 *   gcc -O2 -o /tmp/match_skip_model tools/match_skip_model.c zopfli_amd/csrc/tools/datagen.c
 *   /tmp/match_skip_model T 4000000 [levels, e.g. 4,8,16] [hash bits of the level tables, e.g. 15]
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../oracle/zopfli_oracle.c"

int zopfli_amd_datagen(char cls, unsigned long long seed, unsigned char* out, size_t n);

#define MAXLEV 12
static unsigned NL = 3, LEVK[MAXLEV] = {4, 8, 16}, HBITS = 15;
static int RAW_A = 0;   /* 1: the first chain is walked hit by hit, levels only on the second chain */

typedef struct {
  unsigned short* lv[MAXLEV]; /* level links: distance to the previous position of the same level hash, 0 = none */
  unsigned* rank1;            /* index of the position within its val class, in position order (region) */
  unsigned* rank2;
  unsigned short* val;
  unsigned short* val2;
} Extra;

static uint32_t level_hash(const unsigned char* in, size_t p, size_t n, unsigned k) {
  uint64_t h = 1469598103934665603ull;
  unsigned i;
  for (i = 0; i < k; i++) {
    const unsigned c = p + i < n ? in[p + i] : 0;
    h = (h ^ c) * 1099511628211ull;
  }
  h ^= h >> 29;
  return (uint32_t)(h & ((1u << HBITS) - 1u));
}

static void build_extra(const zo_table* t, size_t n_in, Extra* x) {
  const size_t ws = t->ws, n = t->inend - ws;
  unsigned* cnt = (unsigned*)calloc(32768, sizeof(unsigned));
  long long* head = (long long*)malloc(sizeof(long long) << HBITS);
  size_t k, p;
  unsigned l;
  x->rank1 = (unsigned*)malloc(sizeof(unsigned) * n);
  x->rank2 = (unsigned*)malloc(sizeof(unsigned) * n);
  x->val = (unsigned short*)malloc(2 * n);
  x->val2 = (unsigned short*)malloc(2 * n);
  for (k = 0; k < n; k++) {
    p = ws + k;
    x->val[k] = (unsigned short)zo_val(t->in, p, t->inend);
    x->val2[k] = (unsigned short)zo_val2(t, p);
    x->rank1[k] = cnt[x->val[k]]++;
  }
  memset(cnt, 0, 32768 * sizeof(unsigned));
  for (k = 0; k < n; k++) x->rank2[k] = cnt[x->val2[k]]++;
  for (l = 0; l < NL; l++) {
    x->lv[l] = (unsigned short*)malloc(2 * n);
    for (k = 0; k < ((size_t)1 << HBITS); k++) head[k] = -1;
    for (k = 0; k < n; k++) {
      const uint32_t h = level_hash(t->in, ws + k, n_in, LEVK[l]);
      p = ws + k;
      x->lv[l][k] = (unsigned short)((head[h] >= 0 && p - (size_t)head[h] < ZO_WINDOW) ? p - (size_t)head[h] : 0);
      head[h] = (long long)p;
    }
  }
  free(cnt);
  free(head);
}

static void free_extra(Extra* x) {
  unsigned l;
  for (l = 0; l < NL; l++) free(x->lv[l]);
  free(x->rank1); free(x->rank2); free(x->val); free(x->val2);
}

typedef struct { unsigned short len, dist; } Cp;

static unsigned long long g_touch_raw, g_touch_lev, g_touch_sc, g_ref_hits, g_cmp;

static unsigned lcp(const unsigned char* in, size_t a, size_t b, unsigned limit) {
  unsigned c = 0;
  while (c < limit && in[a + c] == in[b + c]) c++;
  return c;
}

/* the skip-walk: change points (len >= 2 included, as the oracle records them) into cps[], returns their number */
static unsigned skip_walk(const zo_table* t, const Extra* x, size_t pos, Cp* cps, unsigned* out_len, unsigned* out_dist) {
  const unsigned char* in = t->in;
  const size_t size = t->inend, ws = t->ws;
  unsigned limit = ZO_MAX_MATCH, b = 1, bestdist = 0, ncp = 0, idx = 0, chain = 1;
  const unsigned S = t->same[pos - ws];
  const unsigned vpos = x->val[pos - ws], v2pos = x->val2[pos - ws];
  size_t cur = pos;             /* last visited candidate (a member of the current reference chain) */
  int ek = -1;                  /* level of the enumerator, -1 = raw */
  size_t eq = pos;              /* enumerator: the entry whose link is followed next */
  size_t sc = pos;              /* same-class enumerator: on pos's second chain */
  int sc_done = 0;

  if (size - pos < ZO_MIN_MATCH) { *out_len = 0; *out_dist = 0; return 0; }
  if (pos + limit > size) limit = (unsigned)(size - pos);

  for (;;) {
    size_t q = 0;     /* next candidate to visit */
    unsigned hops = 0;
    int have = 0;
    /* level for "needs bestlength + 1 bytes" */
    int k = -1;
    {
      unsigned l;
      for (l = 0; l < NL; l++) if (LEVK[l] <= b + 1) k = (int)l;
    }
    /* On the second chain a candidate has the position's run length (mod 256) besides its hash: as selective as
     * S + 2 shared bytes.  Where that beats the level's k bytes — inside runs, where a level chain links every
     * position of every run of the byte — the reference's own chain is the shorter list. */
    if (chain == 2 && k >= 0 && S + 2 > LEVK[k]) k = -1;
    if (chain == 1 && RAW_A) k = -1;
    if (k < 0) {
      /* raw: the reference's own next hit */
      const unsigned step = chain == 1 ? t->prev1[cur - ws] : t->prev2[cur - ws];
      g_touch_raw++;
      if (step == 0) break;
      q = cur - step;
      hops = 1;
      have = 1;
      ek = -1;
    } else {
      size_t q1 = 0, sw = 0;
      int have1 = 0, havesw = 0;
      if (k != ek) {
        ek = k;
        eq = (LEVK[k] <= b && bestdist != 0) ? pos - bestdist : pos;   /* the candidate that set bestlength shares b >= k bytes */
      }
      /* next level entry below cur: the entries at or above cur were visited or are not on the walk.
       * Not needed on the first chain once bestlength > S: a longer match has exactly the position's run length,
       * so it is of the position's val2 class, and the nearest member of that class ends the first chain anyway. */
      if (!(chain == 1 && b >= S + 1)) {
        size_t e = eq;
        for (;;) {
          const unsigned step = x->lv[ek][e - ws];
          if (step == 0) break;
          g_touch_lev++;
          e -= step;
          if (pos - e >= ZO_WINDOW) break;
          if (e < cur) { q1 = e; have1 = 1; break; }
          eq = e;
        }
      }
      /* the switch point: nearest member of pos's val class AND val2 class below cur */
      if (chain == 1 && b >= S && !sc_done) {
        while (sc >= cur || x->val[sc - ws] != vpos) {
          const unsigned step = t->prev2[sc - ws];
          if (step == 0 || pos - (sc - step) >= ZO_WINDOW) { sc_done = 1; break; }
          g_touch_sc++;
          sc -= step;
        }
        if (!sc_done) { sw = sc; havesw = 1; }
      }
      if (!have1 && !havesw) break;
      if (have1 && (!havesw || q1 >= sw)) {
        q = q1;
        eq = q1;
        /* is q1 visited at all?  a member of the current reference chain */
        if (chain == 1 ? x->val[q - ws] != vpos : x->val2[q - ws] != v2pos) continue;   /* not on the chain: never visited */
      } else {
        q = sw;
      }
      hops = chain == 1 ? x->rank1[cur - ws] - x->rank1[q - ws] : x->rank2[cur - ws] - x->rank2[q - ws];
      have = 1;
    }
    if (!have) break;
    if (pos - q >= ZO_WINDOW) break;           /* lz77.c:464 */
    if (idx + hops > ZO_MAX_CHAIN_HITS) break; /* lz77.c:527-530: only 8192 candidates are looked at */
    idx += hops;
    cur = q;
    {
      unsigned len;
      g_cmp++;
      len = lcp(in, pos, q, limit);
      if (len > b) {
        cps[ncp].len = (unsigned short)len;
        cps[ncp].dist = (unsigned short)(pos - q);
        ncp++;
        b = len;
        bestdist = (unsigned)(pos - q);
        if (len >= limit) break;
      }
    }
    if (chain == 1 && b >= S && x->val2[q - ws] == v2pos) chain = 2;   /* lz77.c:509-519 */
  }
  *out_len = b;
  *out_dist = bestdist;
  return ncp;
}

static unsigned ref_hits(const zo_table* t, size_t pos) {
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
    if (pos + bestlength >= size || in[pos + bestlength] == in[cand + bestlength]) cur = lcp(in, pos, cand, limit);
    if (cur > bestlength) { bestlength = cur; if (cur >= limit) break; }
    if (chain == 1 && bestlength >= t->same[pos - t->ws] && zo_val2(t, pos) == zo_val2(t, cand)) chain = 2;
    step = chain == 1 ? t->prev1[cand - t->ws] : t->prev2[cand - t->ws];
    if (step == 0) break;
    cand -= step;
    dist += step;
    if (--hits_left <= 0) break;
  }
  return hits;
}

int main(int argc, char** argv) {
  const char cls = argc > 1 ? argv[1][0] : 'T';
  const size_t n = argc > 2 ? (size_t)atol(argv[2]) : 2000000;
  const size_t MB = 1000000;
  unsigned char* in;
  size_t b, i, bad = 0, first_bad = (size_t)-1;
  Cp cps[300];
  if (argc > 3) {
    char* s = argv[3];
    NL = 0;
    while (*s && NL < MAXLEV) { LEVK[NL++] = (unsigned)strtoul(s, &s, 10); if (*s == ',') s++; }
  }
  if (argc > 4) HBITS = (unsigned)atoi(argv[4]);
  if (argc > 5) RAW_A = atoi(argv[5]);
  in = (unsigned char*)calloc(n + 4096, 1);
  zopfli_amd_datagen(cls, 1, in, n);
  for (b = 0; b < n; b += MB) {
    const size_t e = b + MB < n ? b + MB : n;
    zo_table* t = zo_table_build(in, b, e);
    Extra x;
    build_extra(t, n, &x);
    for (i = b; i < e; i++) {
      unsigned len, dist, ncp, k, ok = 1;
      const size_t o0 = t->cp_off[i - b], o1 = t->cp_off[i - b + 1];
      g_ref_hits += ref_hits(t, i);
      ncp = skip_walk(t, &x, i, cps, &len, &dist);
      if (len != t->length[i - b] || dist != t->dist[i - b] || ncp != o1 - o0) ok = 0;
      for (k = 0; ok && k < ncp; k++) if (cps[k].len != t->cp_len[o0 + k] || cps[k].dist != t->cp_dist[o0 + k]) ok = 0;
      if (!ok) { bad++; if (first_bad == (size_t)-1) first_bad = i; }
    }
    free_extra(&x);
    zo_table_free(t);
  }
  printf("class %c %zu positions, levels", cls, n);
  for (i = 0; i < NL; i++) printf(" %u", LEVK[i]);
  printf(" (%u hash bits): %zu records differ%s; reference hits %.1f per position; skip-walk touches %.2f (raw %.2f, level %.2f, "
         "class %.2f), compares %.2f\n", HBITS, bad, bad ? " (FIRST BAD below)" : "", (double)g_ref_hits / n,
         (double)(g_touch_raw + g_touch_lev + g_touch_sc) / n, (double)g_touch_raw / n, (double)g_touch_lev / n,
         (double)g_touch_sc / n, (double)g_cmp / n);
  if (bad) printf("  first differing position %zu\n", first_bad);
  return bad != 0;
}

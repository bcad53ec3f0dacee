/* CPU model of k_match3's search structure (test infrastructure: builds on the oracle).
 *
 * The reference walks a hash chain newest to oldest (lz77.c:464-530).  On the device a chain is a CONTIGUOUS SLICE:
 * the positions of every 32768-position chunk of a block's region [windowstart, inend) are sorted by (hash value,
 * position) — once for each of the two hashes (hash.c:110-114, 129-135) — so that the candidates of a position p are
 *
 *     own chunk :  sorted[rank[p] - 1], sorted[rank[p] - 2], ...  down to the start of p's bucket
 *     chunk - 1 :  the bucket of the same hash value from its end downwards, while the candidate is less than
 *                  32768 back (its offset in its chunk is larger than p's offset in its own)
 *
 * and 64 of them are one coalesced load.  The switch to the second hash (lz77.c:509-519) at candidate q continues
 * in the second hash's order just below q (rank2[q] - 1 downwards).  This file restates the walk on those arrays,
 * 64 candidates at a time exactly as the kernel does it (prefix maximum of the common prefix lengths in visit
 * order, first lane that reaches the limit, first lane that satisfies the switch rule, the 8192-hit cap), and
 * compares every position's (length, distance, sublen change points) with the oracle's chain walk.
 *
 *   gcc -O2 -o /tmp/match_bucket_model tools/match_bucket_model.c zopfli_amd/csrc/tools/datagen.c && /tmp/match_bucket_model TXZBPRM 1200000
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../oracle/zopfli_oracle.c"

int zopfli_amd_datagen(char cls, unsigned long long seed, unsigned char* out, size_t n);

#define CH 32768u
#define BATCH 64u

typedef struct {
  unsigned nchunk;
  size_t n;                 /* region length */
  unsigned short* sorted[2];  /* [n]: chunk c's part at c * CH: offsets within the chunk, by (key, offset) */
  unsigned short* rank[2];    /* [n]: index of region position k within its chunk's sorted part */
  unsigned* bucket[2];        /* [nchunk * 32768]: start | count << 15 (count <= 32768 needs 16 bits) */
  unsigned long long batches, cands;
} Buckets;

static unsigned key_of(const zo_table* t, size_t p, int h) { return h == 0 ? zo_val(t->in, p, t->inend) : zo_val2(t, p); }

static void build_buckets(const zo_table* t, Buckets* B) {
  const size_t n = t->inend - t->ws;
  unsigned c, h;
  B->n = n;
  B->nchunk = (unsigned)((n + CH - 1) / CH);
  for (h = 0; h < 2; h++) {
    B->sorted[h] = (unsigned short*)malloc(sizeof(unsigned short) * (n + 1));
    B->rank[h] = (unsigned short*)malloc(sizeof(unsigned short) * (n + 1));
    B->bucket[h] = (unsigned*)calloc((size_t)B->nchunk * 32768u, sizeof(unsigned));
    for (c = 0; c < B->nchunk; c++) {
      const size_t k0 = (size_t)c * CH, k1 = k0 + CH < n ? k0 + CH : n;
      unsigned* cnt = (unsigned*)calloc(32769, sizeof(unsigned));
      unsigned* bk = B->bucket[h] + (size_t)c * 32768u;
      size_t k;
      unsigned key, acc = 0;
      for (k = k0; k < k1; k++) cnt[key_of(t, t->ws + k, (int)h)]++;
      for (key = 0; key < 32768; key++) { bk[key] = acc | (cnt[key] << 15); acc += cnt[key]; cnt[key] = 0; }
      for (k = k0; k < k1; k++) {     /* stable: ascending position within a bucket */
        key = key_of(t, t->ws + k, (int)h);
        const unsigned idx = (bk[key] & 32767u) + cnt[key]++;
        B->sorted[h][k0 + idx] = (unsigned short)(k - k0);
        B->rank[h][k] = (unsigned short)idx;
      }
      free(cnt);
    }
  }
}

/* One position, the kernel's way.  cps: change points (len | dist << 16), returns their number. */
static unsigned walk_buckets(const zo_table* t, Buckets* B, size_t pos, unsigned* cps, unsigned short* out_len,
                             unsigned short* out_dist) {
  const unsigned char* in = t->in;
  const size_t size = t->inend;
  unsigned limit = ZO_MAX_MATCH, bestlength = 1, bestdist = 0, ncp = 0;
  unsigned hits_left = ZO_MAX_CHAIN_HITS;
  const size_t kp = pos - t->ws;                 /* region index of pos */
  const unsigned cp = (unsigned)(kp / CH), op = (unsigned)(kp % CH);
  const unsigned same_p = t->same[kp];
  unsigned h = 0;                                 /* current hash: 0 = first, 1 = second */
  unsigned cc, idx, lo;                           /* cursor: chunk, next index to read (exclusive upper end), bucket start */
  unsigned key[2];
  if (size - pos < ZO_MIN_MATCH) { *out_len = 0; *out_dist = 0; return 0; }
  if (pos + limit > size) limit = (unsigned)(size - pos);
  key[0] = key_of(t, pos, 0);
  key[1] = key_of(t, pos, 1);
  /* the slice of the first hash in pos's own chunk: [bucket start, rank[pos]) */
  cc = cp;
  lo = B->bucket[0][(size_t)cc * 32768u + key[0]] & 32767u;
  idx = B->rank[0][kp];
  for (;;) {
    unsigned nb, i, L[BATCH], cand_off[BATCH], run, stop_lane, switch_lane;
    int done = 0;
    if (idx == lo) {
      /* this chunk's part is used up: on to the previous chunk's bucket, from its end */
      if (cc != cp || cc == 0) break;
      cc = cp - 1;
      {
        const unsigned bk = B->bucket[h][(size_t)cc * 32768u + key[h]];
        lo = bk & 32767u;
        idx = lo + (bk >> 15);
      }
      if (idx == lo) break;
    }
    nb = idx - lo < BATCH ? idx - lo : BATCH;
    if (nb > hits_left) nb = hits_left;
    B->batches++;
    /* lanes 0 .. nb - 1: candidates idx - 1, idx - 2, ... ; in the previous chunk only while offset > op */
    for (i = 0; i < nb; i++) {
      const unsigned off = B->sorted[h][(size_t)cc * CH + idx - 1 - i];
      if (cc != cp && off <= op) { nb = i; done = 1; break; }     /* 32768 or more back: the walk ends (lz77.c:464) */
      cand_off[i] = off;
    }
    if (nb == 0) break;
    B->cands += nb;
    for (i = 0; i < nb; i++) {
      const size_t cand = t->ws + (size_t)cc * CH + cand_off[i];
      unsigned cur = 0;
      while (cur < limit && in[pos + cur] == in[cand + cur]) cur++;
      L[i] = cur;
    }
    /* in visit order: change points = strict prefix maxima above bestlength; the walk stops after the first lane
       whose running maximum reaches the limit (lz77.c:505), or switches after the first lane where the running
       maximum covers same[pos] and the candidate has pos's second hash value (lz77.c:509-519) */
    run = bestlength;
    stop_lane = nb;
    switch_lane = nb;
    for (i = 0; i < nb; i++) {
      const size_t cand = t->ws + (size_t)cc * CH + cand_off[i];
      if (L[i] > run) {
        run = L[i];
        cps[ncp++] = run | ((unsigned)(pos - cand) << 16);
        bestdist = (unsigned)(pos - cand);
        if (run >= limit) { stop_lane = i; break; }
      }
      if (h == 0 && run >= same_p && key_of(t, cand, 1) == key[1]) { switch_lane = i; break; }
    }
    bestlength = run;
    if (stop_lane < nb) break;
    if (switch_lane < nb) {
      /* continue in the second hash's order just below the candidate */
      const unsigned kc = cc * CH + cand_off[switch_lane];
      hits_left -= switch_lane + 1;
      if (hits_left == 0) break;
      h = 1;
      lo = B->bucket[1][(size_t)cc * 32768u + key[1]] & 32767u;
      idx = B->rank[1][kc];
      continue;
    }
    hits_left -= nb;
    if (hits_left == 0 || done) break;
    idx -= nb;
  }
  *out_len = (unsigned short)bestlength;
  *out_dist = (unsigned short)bestdist;
  return ncp;
}

int main(int argc, char** argv) {
  const char* classes = argc > 1 ? argv[1] : "TXZBPRM";
  const size_t n = argc > 2 ? (size_t)atol(argv[2]) : 1200000;
  const size_t MB = 1000000;
  unsigned char* in = (unsigned char*)malloc(n);
  unsigned* cps = (unsigned*)malloc(sizeof(unsigned) * 600);
  int bad = 0;
  for (; *classes; classes++) {
    size_t b, npos = 0;
    unsigned long long batches = 0, cands = 0;
    static const unsigned long long seeds[128] = {['T'] = 1, ['X'] = 2, ['R'] = 3, ['Z'] = 4, ['B'] = 5, ['P'] = 6, ['M'] = 7};
    zopfli_amd_datagen(*classes, seeds[(int)*classes], in, n);
    for (b = 0; b < n; b += MB) {
      const size_t e = b + MB < n ? b + MB : n;
      zo_table* t = zo_table_build(in, b, e);
      Buckets B;
      size_t pos;
      memset(&B, 0, sizeof(B));
      build_buckets(t, &B);
      for (pos = b; pos < e; pos++) {
        unsigned short ml, md;
        const unsigned ncp = walk_buckets(t, &B, pos, cps, &ml, &md);
        const size_t i = pos - b;
        unsigned k, okay = 1, m = 0;
        /* the oracle keeps every change point, also a 2-byte one; so do we */
        if (t->length[i] != ml || (ml >= 3 && t->dist[i] != md)) okay = 0;
        if (okay && t->cp_off[i + 1] - t->cp_off[i] != ncp) okay = 0;
        for (k = 0; okay && k < ncp; k++, m++) {
          const size_t q = t->cp_off[i] + k;
          if ((cps[k] & 0xffffu) != t->cp_len[q] || (cps[k] >> 16) != t->cp_dist[q]) okay = 0;
        }
        if (!okay && bad < 10) {
          printf("class %c pos %zu: model (%u, %u, %u cps) oracle (%u, %u, %zu cps)\n", *classes, pos, ml, md, ncp, t->length[i],
                 t->dist[i], t->cp_off[i + 1] - t->cp_off[i]);
          bad++;
        }
      }
      npos += e - b;
      batches += B.batches;
      cands += B.cands;
      free(B.sorted[0]); free(B.sorted[1]); free(B.rank[0]); free(B.rank[1]); free(B.bucket[0]); free(B.bucket[1]);
      zo_table_free(t);
    }
    printf("class %c: %zu positions, %.2f batches of <= 64 candidates per position, %.1f candidates per position (%.1f per batch)%s\n",
           *classes, npos, (double)batches / npos, (double)cands / npos, batches ? (double)cands / batches : 0.0, bad ? "  MISMATCHES" : "");
  }
  free(in);
  free(cps);
  return bad != 0;
}

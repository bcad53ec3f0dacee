/*
 * TEST INFRASTRUCTURE — not part of the product.
 *
 * Plain-C restatement of the reference's LZ77 optimal-parse hot path
 * (google/zopfli @ /root/reference), used only by tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg as the checker for the HIP kernels.  Nothing in
 * zopfli_amd/ may include, link or call it.
 *
 * Parity pin: tests/test_oracle_vs_reference.py checks every function here
 * against the real reference compiled into oracle/_ref/libzopfli_ref.so
 * (ZopfliFindLongestMatch, ZopfliLZ77Greedy, ZopfliLZ77Optimal[Fixed] called
 * directly on the same inputs) and against the known-answer vectors of
 * SURVEY.md Appendix B.3 committed under tests/golden/.
 */
#ifndef ZOPFLI_ORACLE_H_
#define ZOPFLI_ORACLE_H_

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct zo_table zo_table;

/* Static hash arrays + match records for block [instart, inend) of `in`
 * (hash.c:100-137 as pure functions of data and inend; lz77.c:407-542). */
zo_table* zo_table_build(const unsigned char* in, size_t instart, size_t inend);
void zo_table_free(zo_table* t);

/* ZopfliFindLongestMatch(pos, size=inend, limit=258, sublen) — lz77.c:407.
 * sublen may be NULL; otherwise 259 entries, [3..length] are written. */
void zo_find_longest_match(const zo_table* t, size_t pos, unsigned short* sublen,
                           unsigned short* distance, unsigned short* length);

/* Accessors to the static arrays (absolute position p in [windowstart, inend)). */
unsigned short zo_same(const zo_table* t, size_t p);   /* hash.c:116-126 */
unsigned short zo_prev1(const zo_table* t, size_t p);  /* distance to previous same-hash position, 0 = none */
unsigned short zo_prev2(const zo_table* t, size_t p);  /* same for the second hash (hash.c:129-135) */

/* ZopfliLZ77Greedy — lz77.c:544.  litlens/dists must hold inend-instart entries;
 * returns the number of symbols. */
size_t zo_greedy(const zo_table* t, unsigned short* litlens, unsigned short* dists);

/* GetBestLengths — squeeze.c:217.  ll[288], d[32] are the symbol costs in bits,
 * mincost = GetCostModelMinCost.  length_array has inend-instart+1 entries.
 * Returns costs[blocksize]. */
double zo_get_best_lengths(const zo_table* t, const double* ll, const double* d, double mincost,
                           unsigned short* length_array);

/* TraceBackwards (squeeze.c:317) + FollowPath (:338).  Returns the number of symbols. */
size_t zo_trace_follow(const zo_table* t, const unsigned short* length_array,
                       unsigned short* litlens, unsigned short* dists);

/* Histogram of a symbol run: 288 litlen bins then 32 dist bins, no end symbol. */
void zo_histogram(const unsigned short* litlens, const unsigned short* dists, size_t n, unsigned* hist320);

#ifdef __cplusplus
}
#endif
#endif

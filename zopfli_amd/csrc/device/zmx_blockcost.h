// f-1 of SURVEY 8 on the device: ZopfliCalculateBlockSizeAutoType (deflate.c:610-621) of ranges of LZ77 symbol
// sequences — what every probe of the block-split search costs (blocksplitter.c:103-135: EstimateCost, SplitCost; two
// block sizes a probe, ~800 probes a master block of incompressible data, 30 - 40 us each on a host core).
//
// TWO WAVES per block size (k_block_cost: one prices the counts as they are, one the smoothed counts).  Nothing here is
// floating point: the stored, fixed and dynamic sizes are integers (the reference returns them as doubles), so "the same
// result" is equality of integers.
//
//   histogram        ZopfliLZ77GetHistogram (lz77.c:189-222): prefix counts sampled every BC_S symbols (k_cost_chunks +
//                    k_cost_prefix, once per sequence) + the symbols between the range's ends and their samples
//   code lengths     ZopfliLengthLimitedCodeLengths (katajainen.c:172-262).  The boundary package-merge is a lazy
//                    evaluation of the package-merge lists; here the lists are computed outright, level by level, as
//                    MERGES of the sorted leaves with the pair sums of the list below — two binary searches per item, all
//                    lanes busy — with katajainen.c:85's tie rule (the leaf is taken only if the package is HEAVIER: a
//                    package goes before a leaf of equal weight), and the lengths read off by walking down from the first
//                    2n - 2 items of the top list (a leaf's length = the number of lists it is active in).
//                    tools/models/pm_levels_model.cc checks this formulation against the host's boundary package-merge.
//   tree size        CalculateTreeSize (deflate.c:277-291): EncodeTree's eight ways of using the repeat codes 16 / 17 / 18
//                    (:105-249), each a 19-symbol code of at most 7 bits — the eight package-merges side by side
//   RLE smoothing    OptimizeHuffmanForRle (deflate.c:413-491) + TryOptimizeHuffmanForRle (:525-566): by the whole wave
//                    (frozen-run masks, next-break pointers, jump tables over the breaks the walk visits;
//                    tools/models/smooth_model.cc)
//   data size        CalculateBlockSymbolSizeGivenCounts (deflate.c:383-405)
#pragma once

#define BC_S 1024u           // symbols between two samples of the prefix counts (a sample is 1.3 KB: at 256 a million literals had 5 MB of them)
#define BC_SW 324u           // words of a sample: 288 litlen counts, 32 distance counts, [320] the bytes covered, 3 of padding
#define BC_BYTES 320u
#define BC_WAVES 4u          // waves per workgroup: two block sizes, two waves each

struct CostStoreDev {        // one symbol sequence (ZopfliLZ77Store: litlen | dist << 16 per symbol)
  const u32* sym;
  u32* samples;              // [n / BC_S + 1][BC_SW]: sample k = the counts of symbols [0, k BC_S)
  u32 n;
  u32 nsamples;
};
struct CostPiece {           // k_cost_gather: a store of a table set that becomes part of a sequence
  const u32* src;
  u32* dst;
  u32 n, pad;
};
struct CostEval { u32 store, lstart, lend, pad; };

__device__ __forceinline__ u32 bc_wave_sum_early(u32 v) { return rdlane_u32(wave_scan_add(v), 63); }

// ---- the sequences: pieces copied into place, counts per chunk of BC_S symbols, prefix sums over the chunks
__global__ __launch_bounds__(256) void k_cost_gather(const CostPiece* __restrict__ pieces) {
  const CostPiece P = pieces[blockIdx.y];
  for (u32 i = blockIdx.x * 4096u + threadIdx.x; i < P.n && i < (blockIdx.x + 1u) * 4096u; i += 256u) P.dst[i] = P.src[i];
}

// one wave per chunk: sample k + 1 <- the counts of symbols [k BC_S, (k + 1) BC_S) (only whole chunks have a sample)
__global__ __launch_bounds__(256) void k_cost_chunks(const CostStoreDev* __restrict__ stores, const u32* __restrict__ chunk_first, u32 nstores) {
  __shared__ u32 s_h[4][BC_SW];
  const u32 wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  const u32 g = blockIdx.x * 4u + wave;          // chunk index over all stores
  // which store: chunk_first[s] <= g < chunk_first[s + 1] (a few hundred stores at most: a binary search)
  u32 lo = 0, hi = nstores;
  while (hi - lo > 1u) { const u32 mid = (lo + hi) >> 1; if (chunk_first[mid] <= g) lo = mid; else hi = mid; }
  if (g >= chunk_first[nstores]) return;
  const CostStoreDev S = stores[lo];
  const u32 k = g - chunk_first[lo];
  u32* h = s_h[wave];
  for (u32 w = lane; w < BC_SW; w += 64u) h[w] = 0;
  wave_lds_sync();
  u32 bytes = 0;
  for (u32 i = lane; i < BC_S; i += 64u) {
    const u32 v = S.sym[k * BC_S + i];
    hist_add_symbol(h, v & 0xffffu, v >> 16);
    bytes += (v >> 16) ? (v & 0xffffu) : 1u;
  }
  atomicAdd(&h[BC_BYTES], bytes);
  wave_lds_sync();
  u32* out = S.samples + (u64)(k + 1u) * BC_SW;
  for (u32 w = lane; w < BC_SW; w += 64u) out[w] = h[w];
  if (k == 0) for (u32 w = lane; w < BC_SW; w += 64u) S.samples[w] = 0;
}

// one thread per (store, word): the running sum down the samples
__global__ __launch_bounds__(384) void k_cost_prefix(const CostStoreDev* __restrict__ stores) {
  const CostStoreDev S = stores[blockIdx.x];
  const u32 w = threadIdx.x;
  if (w >= BC_SW) return;
  if (S.nsamples == 1u) { S.samples[w] = 0; return; }       // (fewer than BC_S symbols: k_cost_chunks had no chunk to write sample 0 from)
  u32 acc = 0;
  u32 k = 1;
  for (; k + 8u <= S.nsamples; k += 8u) {
    u32 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = S.samples[(u64)(k + u) * BC_SW + w];
#pragma unroll
    for (int u = 0; u < 8; ++u) { acc += v[u]; S.samples[(u64)(k + u) * BC_SW + w] = acc; }
  }
  for (; k < S.nsamples; ++k) { acc += S.samples[(u64)k * BC_SW + w]; S.samples[(u64)k * BC_SW + w] = acc; }
}

// ZopfliLZ77GetByteRange (lz77.c:160-166) from the start of a sequence: out[q] = the bytes symbols [0, index) stand for —
// where a split point lies in the input (blocksplitter.c:303-314).  One wave per query.
struct CostPosQuery { u32 store, index; };
__global__ __launch_bounds__(64) void k_cost_positions(const CostStoreDev* __restrict__ stores, const CostPosQuery* __restrict__ q, u64* __restrict__ out) {
  const CostPosQuery Q = q[blockIdx.x];
  const CostStoreDev S = stores[Q.store];
  const u32 lane = threadIdx.x, k = Q.index / BC_S;
  u32 bytes = 0;
  for (u32 i = k * BC_S + lane; i < Q.index; i += 64u) {
    const u32 v = S.sym[i];
    bytes += (v >> 16) ? (v & 0xffffu) : 1u;
  }
  bytes = bc_wave_sum_early(bytes);
  if (lane == 0) out[blockIdx.x] = (u64)S.samples[(u64)k * BC_SW + BC_BYTES] + bytes;
}

// ---- one wave's LDS
struct __attribute__((aligned(16))) BcWave {
  u32 h[BC_SW];            // the range's counts (litlen 0 .. 287, distance 288 .. 319, [320] bytes)
  u32 hs[320];             // ... smoothed for the repeat codes
  u8 len[2][320];          // code lengths from h / from hs
  // (from W on: 6368 bytes that the package-merge, the tree sizes and the smoothing use in turn)
  u32 W[288];              // package-merge: the leaves' weights, lightest first
  u16 sym[288];            // ... and their symbols
  u32 M[2][580];           // the lists of two levels
  u32 mask[15][19];        // which items of every list are leaves (the smoothing: its prefix sums)
  u32 act[16];             // active leaves per list
  u32 pad[3];
};
static_assert(sizeof(BcWave) % 16 == 0 && (BC_SW * 4 + 320 * 4 + 640 + 288 * 4 + 288 * 2) % 16 == 0, "the lists are read sixteen bytes at a time");

__device__ __forceinline__ u32 bc_wave_sum(u32 v) { return rdlane_u32(wave_scan_add(v), 63); }
// the largest power of two <= n (n >= 1)
__device__ __forceinline__ u32 bc_top_bit(u32 n) { return 1u << (31 - __clz((int)n)); }

// ZopfliLengthLimitedCodeLengths: counts cnt[0 .. n), n <= 64 NT (NT items a lane), into len8[0 .. n)
template <int NT>
__device__ __forceinline__ void bc_code_lengths(BcWave& S, const u32* cnt, u32 n, u32 maxbits, u8* len8, u32 lane) {
  u32* key = S.M[1];                       // (the first level written is M[1]: its keys are consumed by then)
  // the symbols in use, in symbol order
  u32 used = 0;
  for (u32 i0 = 0; i0 < n; i0 += 64u) {
    const u32 i = i0 + lane;
    const u32 c = i < n ? cnt[i] : 0u;
    if (i < n) len8[i] = 0;
    const u64 nz = __ballot(c != 0);
    if (c != 0) key[used + (u32)__popcll(nz & ((1ull << lane) - 1ull))] = (c << 9) | i;
    used += (u32)__popcll(nz);
  }
  if (lane < 4u) key[used + lane] = 0xffffffffu;      // (the ranks below read the keys four at a time)
  wave_lds_sync();
  if (used == 0) return;
  if (used <= 2u) {                        // katajainen.c:200-207
    if (lane < used) len8[key[lane] & 511u] = 1;
    wave_lds_sync();
    return;
  }
  const u32 nt = (used + 63u) >> 6;        // items per lane
  // lightest first, the symbol breaks ties (katajainen.c:221-229: the symbol rides in the low bits of the sort key): the
  // rank of a key = the number of smaller ones — the keys are distinct
  {
    u32 mine[NT], rank[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) { const u32 e = lane + 64u * (u32)t; mine[t] = e < used ? key[e] : 0xffffffffu; rank[t] = 0; }
    for (u32 j = 0; j < used; j += 4u) {
      const uint4 kj = *reinterpret_cast<const uint4*>(key + j);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        if ((u32)t < nt) rank[t] += (kj.x < mine[t] ? 1u : 0u) + (kj.y < mine[t] ? 1u : 0u) + (kj.z < mine[t] ? 1u : 0u) + (kj.w < mine[t] ? 1u : 0u);
      }
    }
    wave_lds_sync();
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      if (lane + 64u * (u32)t < used) { S.W[rank[t]] = mine[t] >> 9; S.sym[rank[t]] = (u16)(mine[t] & 511u); }
    }
    wave_lds_sync();
  }
  const u32 L = maxbits < used - 1u ? maxbits : used - 1u;      // katajainen.c:234
  // the lists: list 0 = the leaves; list j = leaves merged with the pair sums of list j - 1.  Every item finds its place by
  // a binary search in the other kind; the searches of a lane's items go step by step TOGETHER (a wave alone on its data
  // waits out every LDS round trip: one wait per step, not one per step and item).
  u32 wl[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) { const u32 i = lane + 64u * (u32)t; wl[t] = S.W[i < used ? i : used - 1u]; }   // (every load unconditional, its index clamped: a load inside a branch is waited for inside the branch, one round trip after the other)
  const u32* prev = S.W;
  u32 sp = used;
  for (u32 j = 1; j < L; ++j) {
    u32* next = S.M[j & 1u];
    const u32 m = sp >> 1;
    const u32 ntp = (m + 63u) >> 6;
    if (lane < 19u) S.mask[j][lane] = 0;
    u32 pk[NT], lo_l[NT], lo_p[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      lo_l[t] = 0; lo_p[t] = 0;
      const u32 k = lane + 64u * (u32)t;
      const uint2 pr = *reinterpret_cast<const uint2*>(prev + 2u * (k < m ? k : m - 1u));
      pk[t] = pr.x + pr.y;
    }
    wave_lds_sync();
    // leaf i goes behind every package that is not heavier: lo_l = the number of packages <= its weight
    for (u32 step = bc_top_bit(m); step > 0; step >>= 1) {
      uint2 pr[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) { const u32 cand = lo_l[t] + step; pr[t] = *reinterpret_cast<const uint2*>(prev + 2u * ((cand <= m ? cand : m) - 1u)); }
#pragma unroll
      for (int t = 0; t < NT; ++t) { const u32 cand = lo_l[t] + step; lo_l[t] = (cand <= m && pr[t].x + pr[t].y <= wl[t]) ? cand : lo_l[t]; }
    }
    // package k goes behind every leaf that is lighter: lo_p = the number of leaves < its weight
    for (u32 step = bc_top_bit(used); step > 0; step >>= 1) {
      u32 wv[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) { const u32 cand = lo_p[t] + step; wv[t] = S.W[(cand <= used ? cand : used) - 1u]; }
#pragma unroll
      for (int t = 0; t < NT; ++t) { const u32 cand = lo_p[t] + step; lo_p[t] = (cand <= used && wv[t] < pk[t]) ? cand : lo_p[t]; }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const u32 i = lane + 64u * (u32)t;
      if ((u32)t < nt && i < used) {
        const u32 pos = i + lo_l[t];
        next[pos] = wl[t];
        atomicOr(&S.mask[j][pos >> 5], 1u << (pos & 31u));
      }
      if ((u32)t < ntp && i < m) next[i + lo_p[t]] = pk[t];
    }
    wave_lds_sync();
    prev = next;
    sp = used + m;
  }
  // from the first 2n - 2 items of the top list down: the leaves among them are the list's active leaves, every package
  // among them stands for two items of the list below
  u32 c = 2u * used - 2u;
  for (u32 jj = 0; jj < L; ++jj) {
    const u32 j = L - 1u - jj;
    u32 a;
    if (j == 0) {
      a = c;
    } else {
      u32 bits = 0;
      if (lane < 19u) {
        const u32 wd = S.mask[j][lane];
        const u32 lo_ = 32u * lane;
        bits = c >= lo_ + 32u ? wd : c > lo_ ? wd & ((1u << (c - lo_)) - 1u) : 0u;
      }
      a = bc_wave_sum((u32)__popc(bits));
    }
    if (lane == 0) S.act[j] = a;
    c = 2u * (c - a);
  }
  wave_lds_sync();
  {
    u32 len[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) len[t] = 0;
    for (u32 j = 0; j < L; ++j) {
      const u32 a = S.act[j];
#pragma unroll
      for (int t = 0; t < NT; ++t) len[t] += lane + 64u * (u32)t < a ? 1u : 0u;
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) if (lane + 64u * (u32)t < used) len8[S.sym[lane + 64u * (u32)t]] = (u8)len[t];
  }
  wave_lds_sync();
}

// PatchDistanceCodesForBuggyDecoders (deflate.c:86-103)
__device__ __forceinline__ void bc_two_distance_codes(u8* d8, u32 lane) {
  const u32 m = (u32)__ballot(lane < 30u && d8[lane < 30u ? lane : 0u] != 0);
  const u32 pc = (u32)__popc(m);
  if (pc == 0) { if (lane < 2u) d8[lane] = 1; }
  else if (pc == 1) { if (lane == 0) d8[(m & 1u) ? 1 : 0] = 1; }
  wave_lds_sync();
}

__device__ __forceinline__ u32 bc_length_symbol_extra(u32 s) { return s < 265u || s == 285u ? 0u : (s - 261u) >> 2; }
__device__ __forceinline__ u32 bc_dist_symbol_extra(u32 s) { return s < 4u ? 0u : (s >> 1) - 1u; }

// CalculateBlockSymbolSizeGivenCounts (deflate.c:383-405): the symbols with their extra bits, and the end symbol
__device__ __forceinline__ u32 bc_symbol_bits(const u32* h, const u8* ll8, const u8* d8, u32 lane) {
  u32 s = 0;
  for (u32 i = lane; i < 286u; i += 64u) {
    if (i == 256u) continue;
    s += h[i] * ((u32)ll8[i] + (i > 256u ? bc_length_symbol_extra(i) : 0u));
  }
  if (lane < 30u) s += h[288u + lane] * ((u32)d8[lane] + bc_dist_symbol_extra(lane));
  return bc_wave_sum(s) + (u32)ll8[256];
}

// CalculateTreeSize (deflate.c:277-291): the smallest of EncodeTree's eight headers (:105-249, size only).  The eight
// 19-symbol codes of at most 7 bits are built SIDE BY SIDE by the same package-merge as above: 8 x (19 leaves + 18
// packages) items a level, five a lane.
struct BcTree {                 // the scratch, inside BcWave::M
  u16 rstart[324];              // where every run of equal lengths starts; [nruns] = total
  u8 rsym[324];
  u32 cnt[8][20];               // per way of using the repeat codes: the counts of the 19 code-length symbols
  u16 w[8][20];                 // per way: the sorted leaves' weights
  u16 m[8][2][40];              // ... and its two lists
  u32 lm[8][8][2];              // ... the leaf bits of its lists
  u8 used[8], sp[8];
};
static_assert(sizeof(BcTree) <= sizeof(u32) * 2 * 580, "the tree-size scratch lives in the lists of the package-merge");

__device__ __forceinline__ u32 bc_tree_size(BcWave& S, const u8* ll8, const u8* d8, u32 lane) {
  BcTree& T = *reinterpret_cast<BcTree*>(&S.M[0][0]);
  // hlit / hdist: trailing zero lengths are not sent (deflate.c:122-123)
  const u32 ml = (u32)__ballot(lane < 29u && ll8[257u + (lane < 29u ? lane : 0u)] != 0);
  const u32 md = (u32)__ballot(lane < 29u && d8[1u + (lane < 29u ? lane : 0u)] != 0);
  const u32 hlit = ml ? 32u - (u32)__clz((int)ml) : 0u, hdist = md ? 32u - (u32)__clz((int)md) : 0u;
  const u32 nll = hlit + 257u, total = nll + hdist + 1u;
  // the runs of equal lengths in the sequence litlen lengths ++ distance lengths
  u32 nruns = 0;
  for (u32 i0 = 0; i0 < total; i0 += 64u) {
    const u32 i = i0 + lane;
    u32 cur = 0, prv = 0xffffu;
    if (i < total) {
      cur = i < nll ? ll8[i] : d8[i - nll];
      if (i > 0) prv = i - 1u < nll ? ll8[i - 1u] : d8[i - 1u - nll];
    }
    const u64 st = __ballot(i < total && cur != prv);
    if (i < total && cur != prv) {
      const u32 r = nruns + (u32)__popcll(st & ((1ull << lane) - 1ull));
      T.rstart[r] = (u16)i;
      T.rsym[r] = (u8)cur;
    }
    nruns += (u32)__popcll(st);
  }
  if (lane == 0) T.rstart[nruns] = (u16)total;
  for (u32 x = lane; x < 160u; x += 64u) (&T.cnt[0][0])[x] = 0;
  for (u32 x = lane; x < 128u; x += 64u) (&T.lm[0][0][0])[x] = 0;
  wave_lds_sync();
  // the tokens of every run under each of the eight ways (bit 0: 16 in use, bit 1: 17, bit 2: 18)
  for (u32 r = lane; r < nruns; r += 64u) {
    const u32 symbol = T.rsym[r];
    const u32 run0 = (u32)T.rstart[r + 1u] - (u32)T.rstart[r];
#pragma unroll
    for (u32 v = 0; v < 8u; ++v) {
      const bool use16 = (v & 1u) != 0, use17 = (v & 2u) != 0, use18 = (v & 4u) != 0;
      u32 run = run0;
      if (!(use16 || (symbol == 0 && (use17 || use18)))) {     // no run is looked for: a token per length
        atomicAdd(&T.cnt[v][symbol], run);
        continue;
      }
      if (symbol == 0 && run >= 3u) {
        if (use18) {                                           // while (run >= 11) take min(138, run)
          u32 q = run / 138u, rem = run - 138u * q;
          if (rem >= 11u) { ++q; rem = 0; }
          if (q) atomicAdd(&T.cnt[v][18], q);
          run = rem;
        }
        if (use17) {                                           // while (run >= 3) take min(10, run)
          u32 q = run / 10u, rem = run - 10u * q;
          if (rem >= 3u) { ++q; rem = 0; }
          if (q) atomicAdd(&T.cnt[v][17], q);
          run = rem;
        }
      }
      if (use16 && run >= 4u) {                                // the length once, then repeats of 3 .. 6
        atomicAdd(&T.cnt[v][symbol], 1u);
        --run;
        u32 q = run / 6u, rem = run - 6u * q;
        if (rem >= 3u) { ++q; rem = 0; }
        if (q) atomicAdd(&T.cnt[v][16], q);
        run = rem;
      }
      if (run) atomicAdd(&T.cnt[v][symbol], run);
    }
  }
  wave_lds_sync();
  // ---- the eight codes.  Sorted leaves: item x = (way x / 19, symbol x % 19), its rank among its way's keys
  {
    u32 keyx[3], rk[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const u32 x0 = lane + 64u * (u32)t, x = x0 < 152u ? x0 : 151u, v = x / 19u, i = x - 19u * v;
      const u32 c = T.cnt[v][i];
      keyx[t] = (c && x0 < 152u) ? (c << 5) | i : 0xffffffffu;
      rk[t] = 0;
    }
    for (u32 j = 0; j < 19u; ++j) {
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        const u32 x0 = lane + 64u * (u32)t, v = (x0 < 152u ? x0 : 151u) / 19u;
        const u32 c = T.cnt[v][j];
        const u32 kj = c ? (c << 5) | j : 0xffffffffu;
        rk[t] += (kj < keyx[t] && keyx[t] != 0xffffffffu) ? 1u : 0u;
      }
    }
    if (lane < 8u) {
      u32 u = 0;
      for (u32 j = 0; j < 19u; ++j) u += T.cnt[lane][j] ? 1u : 0u;
      T.used[lane] = (u8)u;
      T.sp[lane] = (u8)u;
    }
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const u32 x = lane + 64u * (u32)t, v = x / 19u;
      if (x < 152u && keyx[t] != 0xffffffffu) T.w[v][rk[t]] = (u16)(keyx[t] >> 5);
    }
    wave_lds_sync();
  }
  // the lists, level by level, for every way at once: item y = (way y / 40, index y % 40): a leaf below `used`, then packages
  for (u32 j = 1; j < 7u; ++j) {
    u32 wgt[5], lo[5];
    u32 kind[5];                        // 0 nothing, 1 leaf, 2 package
#pragma unroll
    for (int t = 0; t < 5; ++t) {
      const u32 y = lane + 64u * (u32)t, v = y / 40u, idx = y - 40u * v;
      const u32 used = T.used[v], sp = T.sp[v], m = sp >> 1;
      const u32 Lv = used < 3u ? 0u : (7u < used - 1u ? 7u : used - 1u);
      const u16* prev = j == 1u ? T.w[v] : T.m[v][(j - 1u) & 1u];
      const u32 wleaf = T.w[v][idx < 19u ? idx : 19u];
      const u32 k = idx >= used ? idx - used : 0u, kk = k < m ? k : (m ? m - 1u : 0u);
      const u32 pa = prev[2u * kk], pb = prev[2u * kk + 1u];
      kind[t] = j >= Lv ? 0u : idx < used ? 1u : idx - used < m ? 2u : 0u;
      wgt[t] = kind[t] == 1u ? wleaf : pa + pb;
      lo[t] = 0;
    }
    // (a leaf looks at the pair sums of the list below, a package at the leaves: two 16-bit loads either way, unconditional)
    const u16* base[5];
    u32 bound[5];
#pragma unroll
    for (int t = 0; t < 5; ++t) {
      const u32 y = lane + 64u * (u32)t, v = y / 40u;
      const u16* prev = j == 1u ? T.w[v] : T.m[v][(j - 1u) & 1u];
      base[t] = kind[t] == 1u ? prev : T.w[v];
      bound[t] = kind[t] == 1u ? (u32)T.sp[v] >> 1 : kind[t] == 2u ? (u32)T.used[v] : 0u;
    }
    for (u32 step = 16u; step > 0; step >>= 1) {      // (at most 19 leaves, 18 packages)
      u32 va[5], vb[5];
#pragma unroll
      for (int t = 0; t < 5; ++t) {
        const u32 cand = lo[t] + step;
        const u32 c1 = (cand <= bound[t] ? cand : (bound[t] ? bound[t] : 1u)) - 1u;
        const u32 at = kind[t] == 1u ? 2u * c1 : c1;
        va[t] = base[t][at];
        vb[t] = base[t][at + 1u];
      }
#pragma unroll
      for (int t = 0; t < 5; ++t) {
        // a leaf counts the packages <= it, a package the leaves < it
        const u32 cand = lo[t] + step;
        const bool ok = cand <= bound[t] && (kind[t] == 1u ? va[t] + vb[t] <= wgt[t] : va[t] < wgt[t]);
        lo[t] = ok ? cand : lo[t];
      }
    }
    wave_lds_sync();
#pragma unroll
    for (int t = 0; t < 5; ++t) {
      const u32 y = lane + 64u * (u32)t, v = y / 40u, idx = y - 40u * v;
      if (kind[t] == 0u) continue;
      const u32 used = T.used[v];
      u16* next = T.m[v][j & 1u];
      if (kind[t] == 1u) {
        const u32 pos = idx + lo[t];
        next[pos] = (u16)wgt[t];
        atomicOr(&T.lm[v][j][pos >> 5], 1u << (pos & 31u));
      } else {
        next[idx - used + lo[t]] = (u16)wgt[t];
      }
    }
    wave_lds_sync();
    if (lane < 8u) {
      const u32 used = T.used[lane];
      const u32 Lv = used < 3u ? 0u : (7u < used - 1u ? 7u : used - 1u);
      if (j < Lv) T.sp[lane] = (u8)(used + ((u32)T.sp[lane] >> 1));
    }
    wave_lds_sync();
  }
  // lane v < 8: its way's lengths from the lists' leaf bits, and its header's size
  u32 size = 0xffffffffu;
  if (lane < 8u) {
    const u32* cnt = T.cnt[lane];
    const u32 used = T.used[lane];
    u32 bits = 0;
    if (used == 1u || used == 2u) {
      for (u32 i = 0; i < used; ++i) bits += T.w[lane][i];       // (every length 1)
      // (T.w holds the sorted weights whatever `used` is)
    } else if (used > 2u) {
      const u32 L = 7u < used - 1u ? 7u : used - 1u;
      u32 act[7];
      u32 c = 2u * used - 2u;
#pragma unroll
      for (u32 jj = 0; jj < 7u; ++jj) {
        const u32 j = 6u - jj;
        act[j] = 0;
        if (j < L) {
          u32 a = c;
          if (j != 0) {
            const u64 lmj = ((u64)T.lm[lane][j][1] << 32) | T.lm[lane][j][0];
            a = (u32)__popcll(lmj & (c >= 64u ? ~0ull : ((1ull << c) - 1ull)));
          }
          act[j] = a;
          c = 2u * (c - a);
        }
      }
      for (u32 i = 0; i < used; ++i) {
        u32 len = 0;
#pragma unroll
        for (u32 j = 0; j < 7u; ++j) len += (j < L && i < act[j]) ? 1u : 0u;
        bits += len * (u32)T.w[lane][i];
      }
    }
    // hclen: trailing zero COUNTS in the order of RFC 1951 3.2.7 are not sent (deflate.c:200-201)
    // order = 16 17 18 0 8 7 9 6 10 5 11 4 12 3 13 2 14 1 15, a nibble-and-a-bit each: packed five bits a symbol
    const u64 ord_lo = 16ull | (17ull << 5) | (18ull << 10) | (0ull << 15) | (8ull << 20) | (7ull << 25) | (9ull << 30) | (6ull << 35) | (10ull << 40) | (5ull << 45) | (11ull << 50) | (4ull << 55);
    const u64 ord_hi = 12ull | (3ull << 5) | (13ull << 10) | (2ull << 15) | (14ull << 20) | (1ull << 25) | (15ull << 30);
    u32 hclen = 15;
    while (hclen > 0) {
      const u32 o = hclen + 3u;
      const u32 symb = o < 12u ? (u32)((ord_lo >> (5u * o)) & 31ull) : (u32)((ord_hi >> (5u * (o - 12u))) & 31ull);
      if (cnt[symb] != 0) break;
      --hclen;
    }
    size = 14u + (hclen + 4u) * 3u + bits + cnt[16] * 2u + cnt[17] * 3u + cnt[18] * 7u;
  }
  // the first of the smallest (deflate.c:283-289)
  u32 best = rdlane_u32(size, 0);
#pragma unroll
  for (u32 v = 1; v < 8u; ++v) { const u32 sv = rdlane_u32(size, v); best = sv < best ? sv : best; }
  wave_lds_sync();
  return best;
}

// OptimizeHuffmanForRle (deflate.c:413-491) of src[0 .. n) into dst (which holds a copy of src), by the whole wave.
// The reference walks the counts once with a running `limit`; every decision it takes reads ORIGINAL counts only (what it
// rewrites lies behind it), so: the frozen runs from the run boundaries (bit masks), for every position b the next break
// after a break at b (next0[b], a short scan), the breaks the walk really visits = the chain first, next0[first], ... —
// enumerated through jump tables next0^(2^k) —, and every visited stretch collapsed to its rounded mean by its own lane.
struct BcSmooth {               // the scratch, from BcWave::W on
  u8 frozen[296];
  u16 nxt[9][292];
};
static_assert(sizeof(BcSmooth) <= 288 * 4 + 288 * 2 + 2 * 580 * 4, "the smoothing's scratch lives in the package-merge's arrays");

__device__ __forceinline__ void bc_smooth_wave(BcWave& S, const u32* src, u32* dst, u32 n, u32 lane) {
  BcSmooth& Q = *reinterpret_cast<BcSmooth*>(&S.W[0]);
  u32* psum = &S.mask[0][0];                      // [length + 1] exclusive prefix sums
  // trailing zeros stay untouched
  u32 length = 0;
  for (u32 i0 = 0; i0 < n; i0 += 64u) {
    const u32 i = i0 + lane;
    const u64 nz = __ballot(i < n && src[i < n ? i : 0u] != 0);
    if (nz) length = i0 + 64u - (u32)__clzll((long long)nz);
  }
  if (length == 0) return;
  // run boundaries (a run = equal counts), as bits; prefix sums
  u64 B[5] = {0, 0, 0, 0, 0};
  u32 carry = 0;
#pragma unroll
  for (int t = 0; t < 5; ++t) {
    const u32 i = lane + 64u * (u32)t;
    if (64u * (u32)t < length) {
      const u32 c = i < length ? src[i] : 0u;
      const u32 p = (i > 0 && i < length) ? src[i - 1u] : 0u;
      B[t] = __ballot(i < length && (i == 0 || c != p));
      const u32 inc = wave_scan_add(c);
      if (i < length) psum[i] = carry + inc - c;
      carry += rdlane_u32(inc, 63);
    }
  }
  if (lane == 0) psum[length] = carry;
  // frozen: runs of >= 5 zeros or >= 7 equal non-zero counts (deflate.c:437-459)
#pragma unroll
  for (int t = 0; t < 5; ++t) {
    const u32 i = lane + 64u * (u32)t;
    if (i < length) {
      // the run's start: the highest boundary bit at or below i; its end: the lowest above i, or length
      int w = (int)(i >> 6);
      u64 x = B[t] & (~0ull >> (63u - (i & 63u)));
#pragma unroll
      for (int g = 4; g > 0; --g) { if (x == 0 && w > 0) { --w; x = w == 3 ? B[3] : w == 2 ? B[2] : w == 1 ? B[1] : B[0]; } }
      const u32 start = (u32)w * 64u + 63u - (u32)__clzll((long long)x);
      u32 end = length;
      {
        int w2 = (int)((i + 1u) >> 6);
        u64 y = 0;
        if (w2 < 5) { const u64 bw = w2 == 4 ? B[4] : w2 == 3 ? B[3] : w2 == 2 ? B[2] : w2 == 1 ? B[1] : B[0]; y = bw & (~0ull << ((i + 1u) & 63u)); }
#pragma unroll
        for (int g = 4; g > 0; --g) { if (y == 0 && w2 < 4) { ++w2; y = w2 == 4 ? B[4] : w2 == 3 ? B[3] : w2 == 2 ? B[2] : B[1]; } }
        if (y != 0 && w2 < 5) end = (u32)w2 * 64u + (u32)__ffsll((long long)y) - 1u;
      }
      const u32 run = end - start, c = src[i];
      Q.frozen[i] = ((c == 0 && run >= 5u) || (c != 0 && run >= 7u)) ? 1 : 0;
    }
  }
  if (lane == 0) Q.frozen[length] = 1;
  wave_lds_sync();
  // next0[b]: the first i > b with i == length, frozen[i] or |count[i] - limit(b)| >= 4, limit(b) as set at a break at b
  // (deflate.c:476-485); lane 63's extra item: the walk's start, where limit = count[0] (:463)
#pragma unroll
  for (int t = 0; t < 5; ++t) {
    const u32 b = lane + 64u * (u32)t;
    if (b < length) {
      const u32 limit = b + 3u < length ? (src[b] + src[b + 1u] + src[b + 2u] + src[b + 3u] + 2u) / 4u : src[b];
      u32 i = b + 1u;
      while (i < length && Q.frozen[i] == 0) {
        const u32 a = src[i];
        if ((a > limit ? a - limit : limit - a) >= 4u) break;
        ++i;
      }
      Q.nxt[0][b] = (u16)i;
    }
  }
  u32 first = 0;
  {
    u32 i = 0;
    if (lane == 63u && Q.frozen[0] == 0) {
      const u32 limit = src[0];
      i = 1u;
      while (i < length && Q.frozen[i] == 0) {
        const u32 a = src[i];
        if ((a > limit ? a - limit : limit - a) >= 4u) break;
        ++i;
      }
    }
    first = rdlane_u32(i, 63);
  }
  if (lane == 0) Q.nxt[0][length] = (u16)length;
  wave_lds_sync();
  for (u32 k = 1; k < 9u; ++k) {
#pragma unroll
    for (int t = 0; t < 5; ++t) {
      const u32 b = lane + 64u * (u32)t;
      if (b <= length) Q.nxt[k][b] = Q.nxt[k - 1u][Q.nxt[k - 1u][b]];
    }
    wave_lds_sync();
  }
  // the stretches the walk collapses: [0, first) if the walk does not break at 0, then [b, next0[b]) for every break b
  // it visits (the l-th = next0^l(first))
  auto collapse = [&](u32 b, u32 nb) {
    const u32 stride = nb - b, sum = psum[nb] - psum[b];
    if (stride >= 4u || (stride >= 3u && sum == 0)) {
      u32 mean = (sum + stride / 2u) / stride;
      if (mean < 1u) mean = 1;
      if (sum == 0) mean = 0;
      for (u32 k = b; k < nb; ++k) dst[k] = mean;
    }
  };
  if (lane == 63u && first > 0) collapse(0, first);
#pragma unroll
  for (int t = 0; t < 5; ++t) {
    const u32 l = lane + 64u * (u32)t;
    if (64u * (u32)t < length) {
      u32 x = first;
#pragma unroll
      for (u32 k = 0; k < 9u; ++k) if ((l >> k) & 1u) x = Q.nxt[k][x];
      if (x < length) collapse(x, Q.nxt[0][x]);
    }
  }
  wave_lds_sync();
}

struct BlockCostParams {
  const CostStoreDev* stores;
  const CostEval* evals;
  double* out;
  u32 n;
  u64* prof;                 // ZOPFLI_AMD_BC_PROF: [10] cycles per phase, summed over the waves (nullptr: no timers)
};
#define BC_T(K) if (P.prof) { const u64 t_ = (u64)__builtin_readcyclecounter(); if (lane == 0) atomicAdd(&P.prof[K], t_ - tp_); tp_ = t_; }

// Two waves a block size: wave A prices the code of the counts as they are, wave B the code of the smoothed counts
// (TryOptimizeHuffmanForRle's second try, deflate.c:541-556) — the two halves of GetDynamicLengths are independent until
// their sums are compared, and a round of the split search waits for its slowest wave.
__global__ __launch_bounds__(64 * BC_WAVES) void k_block_cost(BlockCostParams P) {
  __shared__ BcWave s_w[BC_WAVES];
  __shared__ u32 s_second[BC_WAVES / 2];
  const u32 wave = (u32)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63u;
  const u32 pair = wave >> 1;
  const bool second = (wave & 1u) != 0;
  const u32 e = blockIdx.x * (BC_WAVES / 2u) + pair;
  const bool valid = e < P.n;
  BcWave& S = s_w[wave];
  u32 dynamic = 0xffffffffu;
  u64 stored = 0, fixed = 0;
  if (valid) {
    const CostEval E = P.evals[e];
    const CostStoreDev St = P.stores[E.store];
    u64 tp_ = P.prof ? (u64)__builtin_readcyclecounter() : 0ull;
    // ---- ZopfliLZ77GetHistogram: the difference of two prefix counts (either wave its own copy)
    const u32 shi = E.lend / BC_S, slo = E.lstart / BC_S;
    const bool direct = E.lend - E.lstart < 2u * BC_S;
    if (direct) {
      for (u32 w = lane; w < BC_SW; w += 64u) S.h[w] = 0;
    } else {
      const u32* a = St.samples + (u64)shi * BC_SW;
      const u32* b = St.samples + (u64)slo * BC_SW;
      for (u32 w = lane; w < BC_SW; w += 64u) S.h[w] = a[w] - b[w];
    }
    wave_lds_sync();
    {
      u32 bytes_add = 0, bytes_sub = 0;
      for (u32 i = (direct ? E.lstart : shi * BC_S) + lane; i < E.lend; i += 64u) {
        const u32 v = St.sym[i];
        hist_add_symbol(S.h, v & 0xffffu, v >> 16);
        bytes_add += (v >> 16) ? (v & 0xffffu) : 1u;
      }
      if (!direct) {
        for (u32 i = slo * BC_S + lane; i < E.lstart; i += 64u) {
          const u32 v = St.sym[i];
          const u32 litlen = v & 0xffffu, dist = v >> 16;
          if (dist == 0) {
            atomicSub(&S.h[litlen], 1u);
          } else {
            atomicSub(&S.h[dev_length_symbol(litlen)], 1u);
            atomicSub(&S.h[288 + dev_dist_symbol(dist)], 1u);
          }
          bytes_sub += dist ? litlen : 1u;
        }
      }
      if (bytes_add) atomicAdd(&S.h[BC_BYTES], bytes_add);
      if (bytes_sub) atomicSub(&S.h[BC_BYTES], bytes_sub);
    }
    wave_lds_sync();
    if (!second) {
      // ---- stored and fixed (deflate.c:610-621)
      const u64 length = S.h[BC_BYTES];
      stored = (length / 65535ull + (length % 65535ull ? 1ull : 0ull)) * 40ull + length * 8ull;   // deflate.c:591-597
      fixed = stored;
      if (St.n <= 1000u) {                     // deflate.c:615: the fixed tree is only priced for a small store
        u32 s = 0;
        for (u32 i = lane; i < 286u; i += 64u) {
          if (i == 256u) continue;
          s += S.h[i] * (i < 144u ? 8u : i < 256u ? 9u : i < 280u ? 7u + bc_length_symbol_extra(i) : 8u + bc_length_symbol_extra(i));
        }
        if (lane < 30u) s += S.h[288u + lane] * (5u + bc_dist_symbol_extra(lane));
        fixed = 3ull + bc_wave_sum(s) + 7ull;
      }
    }
    // GetDynamicLengths (deflate.c:569-583)
    if (lane == 0) S.h[256] = 1;             // the end symbol
    wave_lds_sync();
    if (!second) {
      BC_T(0)
      bc_code_lengths<5>(S, S.h, 288u, 15u, S.len[0], lane);
      BC_T(1)
      bc_code_lengths<1>(S, S.h + 288, 32u, 15u, S.len[0] + 288, lane);
      bc_two_distance_codes(S.len[0] + 288, lane);
      BC_T(2)
      const u32 tree = bc_tree_size(S, S.len[0], S.len[0] + 288, lane);
      BC_T(3)
      const u32 data = bc_symbol_bits(S.h, S.len[0], S.len[0] + 288, lane);
      BC_T(4)
      dynamic = tree + data;
    } else {
      // TryOptimizeHuffmanForRle (deflate.c:525-566): the counts smoothed for the repeat codes, priced on the real counts
      for (u32 w = lane; w < 320u; w += 64u) S.hs[w] = S.h[w];
      wave_lds_sync();
      bc_smooth_wave(S, S.h, S.hs, 288u, lane);
      bc_smooth_wave(S, S.h + 288, S.hs + 288, 32u, lane);
      BC_T(5)
      bool changed = false;
      for (u32 w = lane; w < 320u; w += 64u) changed |= S.hs[w] != S.h[w];
      if (__ballot(changed) != 0) {            // (nothing smoothed: the same counts give the same lengths and the same sizes)
        bc_code_lengths<5>(S, S.hs, 288u, 15u, S.len[1], lane);
        BC_T(6)
        bc_code_lengths<1>(S, S.hs + 288, 32u, 15u, S.len[1] + 288, lane);
        bc_two_distance_codes(S.len[1] + 288, lane);
        BC_T(7)
        const u32 tree2 = bc_tree_size(S, S.len[1], S.len[1] + 288, lane);
        const u32 data2 = bc_symbol_bits(S.h, S.len[1], S.len[1] + 288, lane);
        dynamic = tree2 + data2;
        BC_T(8)
      }
      if (lane == 0) s_second[pair] = dynamic;
    }
  }
  __syncthreads();
  if (valid && !second) {
    const u32 d2 = s_second[pair];
    if (d2 < dynamic) dynamic = d2;            // deflate.c:557: the second try only if it is smaller
    const u64 dyn = 3ull + dynamic;
    const u64 best = (stored < fixed && stored < dyn) ? stored : (fixed < dyn ? fixed : dyn);
    if (lane == 0) P.out[e] = (double)best;
  }
}

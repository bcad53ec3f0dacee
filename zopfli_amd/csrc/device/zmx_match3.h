// k_bucket + k_match3: the match table (ZopfliFindLongestMatch(limit 258, sublen) for every position,
// lz77.c:407-542) with the hash chains laid out as CONTIGUOUS CANDIDATE SLICES and the candidates of a
// position evaluated 64 at a time by a whole wave.  Included only by zmx_hip.hip, after zmx_kernels.h.
//
// The reference walks a chain of hash.c's prev links newest to oldest, one dependent load per candidate;
// k_match2 does the same with one lane per position: 10^10 dependent 8-byte loads per 100 MB of text, a full
// memory round trip each, and on data whose 3-byte hashes collide (filtered image rows, two-symbol sources)
// thousands of them per position.  Here the links are replaced by a sort:
//
//   k_bucket   every 32768-position chunk of a block's region [windowstart, inend) sorted by (hash value,
//              position), once per hash (hash.c:110-114 the 3-byte hash, :129-135 the run-length hash):
//                sorted[h][k]  u16  offset in its chunk of the k-th position of the chunk in that order
//                rank[h][p]    u16  where position p stands in its chunk's order
//                bucket[h][chunk][v]  u32  start | end << 16 of hash value v's positions in sorted[h]
//                ssame[k]      u8   (same[] & 255) of the position sorted[0][k] (the switch rule's test)
//              so that the candidates of p — the earlier positions of the same hash value less than 32768
//              back, newest first (hash.c:110-114: older ones are unreachable) — are two slices read
//              DOWNWARDS: own chunk from rank[p] - 1 to the bucket's start, then the previous chunk's bucket
//              from its end while the candidate's offset in its chunk is larger than p's in its own.
//   k_match3   a wave takes 64 positions of a tile: lane = position for the set-up (hash values, ranks,
//              bucket bounds: coalesced or gathered loads, nothing serial), then position by position the
//              wave reads 64 candidates with ONE coalesced load, tests the four bytes that end at the best
//              length so far against the LDS window (lz77.c:478-479 generalised: a candidate that fails
//              cannot be longer), computes the common prefix length of the survivors (GetMatch, lz77.c:297),
//              takes the prefix maximum in visit order (a lane whose length exceeds everything before it is
//              a change point of sublen, lz77.c:495-505), and resolves with ballots where the walk stops
//              (the limit is reached, :505) or switches to the second hash (:509-519: the first candidate,
//              once the best length covers same[pos], that has pos's second hash value — the walk goes on
//              just below that candidate in the second hash's order), the 8192-hit cap (:527-530) and the
//              window (:464).  tools/match_bucket_model.c is this walk in plain C, checked against the oracle's
//              chain walk on every class.
#pragma once

#define BK_THREADS 1024u
#define BK_WAVES 16u
#define BK_CH 32768u
#define BK_LDS_BYTES (65536u + 65536u + BK_WAVES * 256u * 4u + 1024u + 64u)   // keys, order after pass 1, per-wave digit counters, digit totals

struct BucketParams {
  const u8* in;
  const BlockDesc* blocks;
  const u16* same16;       // [region] (k_same)
  const u32* chunk_base;   // [nb] first chunk of each block in bucket[]
  const u64* link_lo;      // optional (tables built from a parent): positions below it are never read
  u16* sorted[2];          // [region]
  u16* rank[2];            // [region]
  u32* bucket[2];          // [chunks][32768]
  u8* ssame;               // [region]
};

// The lanes of the wave whose `d` equals mine (NBITS-bit digits; inactive lanes are in nobody's group).
template <int NBITS>
__device__ __forceinline__ u64 bk_peers(u32 d, bool act) {
  u64 grp = __ballot(act);
#pragma unroll
  for (int bit = 0; bit < NBITS; ++bit) {
    const bool mine = (d >> bit) & 1;
    const u64 bm = __ballot(mine);
    grp &= mine ? bm : ~bm;
  }
  return grp;
}

// One stable counting pass over the chunk: elements in the order j = 0 .. n - 1 (FIRST: the positions
// themselves; otherwise the order pass 1 left in s_ord), digit = (key >> SHIFT) & (2^NBITS - 1).  Wave w owns
// j in [2048 w, 2048 w + 2048): count per (wave, digit), exclusive scan over (digit, wave), then every wave
// places its elements in order.  FIRST writes the new order into s_ord; the second pass writes the final
// arrays in HBM.
template <int NBITS, int SHIFT, bool FIRST>
__device__ __forceinline__ void bk_pass(const u16* s_key, u16* s_ord, u32 (*s_cnt)[256], u32* s_tot, u32 n,
                                        u16* g_sorted, u16* g_rank, u8* g_ssame, const u16* g_same) {
  const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const u64 lt_mask = (1ull << lane) - 1;
  constexpr u32 ND = 1u << NBITS;
  for (u32 i = tid; i < BK_WAVES * 256u; i += BK_THREADS) (&s_cnt[0][0])[i] = 0;
  __syncthreads();
  const u32 j0 = wave * 2048u;
  // ---- count
  for (u32 s = 0; s < 32; ++s) {
    const u32 j = j0 + s * 64u + lane;
    const bool act = j < n;
    const u32 e = act ? (FIRST ? j : (u32)s_ord[j]) : 0u;
    const u32 d = ((u32)s_key[e] >> SHIFT) & (ND - 1u);
    const u64 grp = bk_peers<NBITS>(d, act);
    if (act && (grp & lt_mask) == 0) s_cnt[wave][d] += (u32)__popcll(grp);
    wave_lds_sync();
  }
  __syncthreads();
  // ---- exclusive scan over (digit, wave)
  if (tid < ND) {
    u32 run = 0;
    for (u32 w = 0; w < BK_WAVES; ++w) { const u32 c = s_cnt[w][tid]; s_cnt[w][tid] = run; run += c; }
    s_tot[tid] = run;
  }
  __syncthreads();
  if (tid < 64) {
    u32 carry = 0;
    for (u32 d0 = 0; d0 < ND; d0 += 64) {
      const u32 v = s_tot[d0 + lane];
      const u32 incl = wave_scan_add(v);
      s_tot[d0 + lane] = carry + incl - v;
      carry += rdlane_u32(incl, 63);
    }
  }
  __syncthreads();
  // ---- place
  for (u32 s = 0; s < 32; ++s) {
    const u32 j = j0 + s * 64u + lane;
    const bool act = j < n;
    const u32 e = act ? (FIRST ? j : (u32)s_ord[j]) : 0u;
    const u32 d = ((u32)s_key[e] >> SHIFT) & (ND - 1u);
    const u64 grp = bk_peers<NBITS>(d, act);
    const u32 base = s_cnt[wave][d] + s_tot[d];
    const u32 dest = base + (u32)__popcll(grp & lt_mask);
    wave_lds_sync();                                  // every lane has read its counter
    if (act && (grp >> lane) == 1ull) s_cnt[wave][d] += (u32)__popcll(grp);   // the last lane of the group
    wave_lds_sync();
    if (act) {
      if (FIRST) {
        s_ord[dest] = (u16)e;                         // (read only after the barrier below: the waves write disjoint slots)
      } else {
        g_sorted[dest] = (u16)e;
        g_rank[e] = (u16)dest;
        if (g_ssame) g_ssame[dest] = (u8)g_same[e];
      }
    }
  }
  __syncthreads();
}

// In the FIRST pass s_ord is written while other waves may still read it?  No: pass 1 reads s_key only
// (its elements are the positions themselves), pass 2 reads s_ord and writes HBM.
__global__ __launch_bounds__(BK_THREADS) void k_bucket(BucketParams P) {
  extern __shared__ __align__(16) u8 bk_lds[];
  u16* s_key = reinterpret_cast<u16*>(bk_lds);
  u16* s_ord = s_key + BK_CH;
  u32 (*s_cnt)[256] = reinterpret_cast<u32 (*)[256]>(bk_lds + 131072u);
  u32* s_tot = reinterpret_cast<u32*>(bk_lds + 131072u + BK_WAVES * 256u * 4u);

  const u32 b = blockIdx.y, h = blockIdx.z, tid = threadIdx.x;
  const BlockDesc bd = P.blocks[b];
  const u64 L = bd.inend - bd.ws;
  const u64 e0 = (u64)blockIdx.x * BK_CH;
  if (e0 >= L) return;
  const u32 n = (u32)((L - e0 < BK_CH) ? L - e0 : BK_CH);
  if (P.link_lo && e0 + n <= P.link_lo[b]) return;    // nobody reads this chunk
  const u8* base = P.in + bd.ws + e0;
  const u16* same = P.same16 + bd.reg_off + e0;
  const u64 left = L - e0;                            // bytes from the chunk's start to the block end
  u32* bucket = P.bucket[h] + ((u64)P.chunk_base[b] + blockIdx.x) * 32768u;
  // empty hash values: start = end = 0.  Done (acknowledged) before anybody writes a real entry.
  for (u32 i = tid; i < 32768u / 4u; i += BK_THREADS) reinterpret_cast<uint4*>(bucket)[i] = make_uint4(0, 0, 0, 0);
  // ---- the keys (hash.c:96-98 three rolling updates, zero past the block end: :107-108, :139-143; :129 the second hash)
  for (u32 i = tid; i < n; i += BK_THREADS) {
    const u32 b0 = base[i];
    const u32 b1 = (u64)i + 1 < left ? (u32)base[i + 1] : 0u;
    const u32 b2 = (u64)i + 2 < left ? (u32)base[i + 2] : 0u;
    u32 v = ((b0 << 10) ^ (b1 << 5) ^ b2) & 32767u;
    if (h) v ^= ((u32)same[i] - 3u) & 255u;
    s_key[i] = (u16)v;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  u16* g_sorted = P.sorted[h] + bd.reg_off + e0;
  u16* g_rank = P.rank[h] + bd.reg_off + e0;
  bk_pass<8, 0, true>(s_key, s_ord, s_cnt, s_tot, n, nullptr, nullptr, nullptr, nullptr);
  bk_pass<7, 8, false>(s_key, s_ord, s_cnt, s_tot, n, g_sorted, g_rank, h == 0 ? P.ssame + bd.reg_off + e0 : nullptr, same);
  // ---- the bucket bounds, from the finished order: this workgroup wrote it, the writes are done
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  u16* bk16 = reinterpret_cast<u16*>(bucket);
  {
    const u32 f0 = tid * 32u;
    if (f0 < n) {
      u32 prev = f0 ? (u32)s_key[__builtin_nontemporal_load(g_sorted + f0 - 1)] : 0xffffffffu;
      const u32 f1 = f0 + 32u < n ? f0 + 32u : n;
      for (u32 f = f0; f < f1; ++f) {
        const u32 k = s_key[__builtin_nontemporal_load(g_sorted + f)];
        if (k != prev) {
          bk16[2u * k] = (u16)f;
          if (prev != 0xffffffffu) bk16[2u * prev + 1u] = (u16)f;
        }
        prev = k;
      }
      if (f1 == n) bk16[2u * prev + 1u] = (u16)n;     // (32768 fits in 16 bits)
    }
  }
}

// ---------------------------------------------------------------------------------------------
// k_match3
// ---------------------------------------------------------------------------------------------
#define M4_PAD 32u      // sorted[] / ssame[] have this many entries in front: a 16-byte read of k_match4 may start below index 0
#define M3_THREADS 512u
#define M3_WAVES 8u
#define M3_SCR 264u                      // change points of one position (lengths 2..258 strictly increasing: at most 257)

struct Match3Params {
  const u8* in;
  const BlockDesc* blocks;
  const u32* tile_off;     // [nb + 1] cumulative tile counts
  u32 nb;
  u32 total_tiles;
  const u16* same16;
  const u32* chunk_base;
  const u16* sorted[2];
  const u16* rank[2];
  const u32* bucket[2];
  const u8* ssame;
  u32* recs;
  u32* pool;
  u32 pool_cap;
  u32* counters;           // [0] pool cursor, [1] error flags, [4..7] profile sums, [8..15] per-XCD tile cursors
  const u32* tile_list;    // optional: the tiles to do (total_tiles entries); null = all of them
  u32* scratch;            // k_match4: gridDim.x * M4_THREADS * SCRATCH_CPS change points beyond a record's eight
};

__device__ __forceinline__ u32 m3_lds_u32(const u32* win, u32 byte_off) {
  u32 x;
  __builtin_memcpy(&x, reinterpret_cast<const char*>(win) + byte_off, 4);
  return x;
}
__device__ __forceinline__ u64 m3_lds_u64(const u32* win, u32 byte_off) {
  u64 x;
  __builtin_memcpy(&x, reinterpret_cast<const char*>(win) + byte_off, 8);
  return x;
}
__device__ __forceinline__ u32 m3_sgpr(u32 v) { return (u32)__builtin_amdgcn_readfirstlane((int)v); }

template <bool PROF>
__global__ __launch_bounds__(M3_THREADS, 6) void k_match3(Match3Params P) {
  __shared__ __align__(16) u32 win[MWIN_BYTES / 4 + 12];    // (+ the extra zeroed vector and the 8-byte compares' overshoot)
  __shared__ u32 s_scr[M3_WAVES][M3_SCR];              // the change points of the wave's current position, len | dist << 16
  __shared__ __align__(16) u32 s_rec[M3_WAVES][8];     // its record as it will lie in HBM
  __shared__ u32 s_next, s_tile;

  const u32 tid = threadIdx.x, lane = tid & 63;
  const u32 wave = m3_sgpr(tid >> 6);
  const u32 xcd = blockIdx.x & 7;
  u32* const scr = s_scr[wave];
  u32* const srec = s_rec[wave];
  const u64 lt_mask = (1ull << lane) - 1;
  u64 n_hits = 0, n_batches = 0, n_pass = 0, n_lcp = 0, n_quiet = 0;

  for (;;) {
    __syncthreads();  // previous tile fully consumed before the window is overwritten
    if (tid == 0) {
      // tiles are dealt to the XCDs in groups of M_XCD_GROUP consecutive tiles (one 32 KiB stretch: its window
      // stays in one L2), round robin — not in contiguous eighths of the input: a stretch of expensive data (long
      // chains) would be one XCD's alone while the others sit idle
      const u32 k = atomicAdd(&P.counters[8 + xcd], 1u);
      s_tile = ((k / M_XCD_GROUP) * 8u + xcd) * M_XCD_GROUP + (k % M_XCD_GROUP);
      s_next = 0;
    }
    __syncthreads();
    if (s_tile >= P.total_tiles) break;
    const u32 tile = P.tile_list ? P.tile_list[s_tile] : s_tile;
    u32 lo_b = 0, hi_b = P.nb;
    while (hi_b - lo_b > 1) {
      const u32 mid = (lo_b + hi_b) >> 1;
      if (P.tile_off[mid] <= tile) lo_b = mid; else hi_b = mid;
    }
    const BlockDesc bd = P.blocks[lo_b];
    const u64 p0 = bd.instart + (u64)(tile - P.tile_off[lo_b]) * MT;
    const u64 p1 = (p0 + MT < bd.inend) ? p0 + MT : bd.inend;
    const u32 ntile = (u32)(p1 - p0);
    // stage bytes [p0 - 32768, p1 + 258) (clipped to [0, inend): zeros outside) at LDS offset (abs - wb)
    const long long wb = ((long long)p0 - (long long)ZMX_WINDOW) & ~15ll;
    const u64 hi_abs = (p1 + ZMX_MAX_MATCH < bd.inend) ? p1 + ZMX_MAX_MATCH : bd.inend;
    const u32 nvec = (u32)(((long long)hi_abs - wb + 15) >> 4);
    for (u32 v = tid; v < nvec + 1; v += M3_THREADS) {
      const long long a = wb + (long long)v * 16;
      uint4 x = make_uint4(0, 0, 0, 0);
      if (a >= 0 && v < nvec) {
        x = *reinterpret_cast<const uint4*>(P.in + a);  // input is padded past its end
        // bytes at or past the block end count as zero (hash.c:107-108; no match runs past it, lz77.c:448-450)
        if (a + 16 > (long long)hi_abs) {
          const u32 keep = (u32)((long long)hi_abs - a);
          u32 w[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const u32 kb = keep > 4u * q ? keep - 4u * q : 0u;
            w[q] = kb >= 4 ? w[q] : kb == 0 ? 0u : (w[q] & ((1u << (8u * kb)) - 1u));
          }
          x = make_uint4(w[0], w[1], w[2], w[3]);
        }
      }
      reinterpret_cast<uint4*>(win)[v] = x;
    }
    __syncthreads();

    const u32 li0 = (u32)(p0 - bd.ws);             // region index of the tile's first position
    const u32 lp0 = (u32)((long long)p0 - wb);     // its LDS byte offset
    const u32 rem0 = (u32)((bd.inend - p0 < 70000) ? bd.inend - p0 : 70000);
    u32* const rec0 = P.recs + (bd.pos_off + (p0 - bd.instart)) * 8;
    const u16* const g_same = P.same16 + bd.reg_off;
    const u16* const g_sorted0 = P.sorted[0] + bd.reg_off;
    const u16* const g_sorted1 = P.sorted[1] + bd.reg_off;
    const u16* const g_rank0 = P.rank[0] + bd.reg_off;
    const u16* const g_rank1 = P.rank[1] + bd.reg_off;
    const u8* const g_ssame = P.ssame + bd.reg_off;
    const u32* const g_bucket0 = P.bucket[0] + (u64)P.chunk_base[lo_b] * 32768u;
    const u32* const g_bucket1 = P.bucket[1] + (u64)P.chunk_base[lo_b] * 32768u;

    for (;;) {
      // ---- 64 positions of the tile: lane = position for the set-up
      u32 first = 0;
      if (lane == 0) first = atomicAdd(&s_next, 64u);
      first = m3_sgpr(first);
      if (first >= ntile) break;
      const u32 npos = ntile - first < 64u ? ntile - first : 64u;
      const bool have = lane < npos;
      const u32 my = first + (have ? lane : 0u);
      const u32 kp_v = li0 + my;                   // region index
      const u32 lp_v = lp0 + my;                   // LDS offset
      const u32 rem_v = rem0 - my;                 // bytes to the block end, saturated
      const u32 same_v = g_same[kp_v];
      const u32 b012 = m3_lds_u32(win, lp_v);      // (zeros past the block end: staged that way)
      const u32 key0_v = (((b012 & 255u) << 10) ^ (((b012 >> 8) & 255u) << 5) ^ ((b012 >> 16) & 255u)) & 32767u;
      const u32 key1_v = key0_v ^ ((same_v - 3u) & 255u);
      const u32 c_v = kp_v >> 15, o_v = kp_v & 32767u;
      const u32 r0_v = g_rank0[kp_v];
      const u32 bk0_v = g_bucket0[(u64)c_v * 32768u + key0_v];            // own chunk, first hash: start | end << 16
      const u32 bk0p_v = c_v ? g_bucket0[(u64)(c_v - 1) * 32768u + key0_v] : 0u;   // previous chunk
      const u32 bk1_v = g_bucket1[(u64)c_v * 32768u + key1_v];
      const u32 bk1p_v = c_v ? g_bucket1[(u64)(c_v - 1) * 32768u + key1_v] : 0u;
      // The usual case folded into the set-up (lane-parallel): same[pos] <= 1 and the nearest candidate of the first
      // hash has pos's second hash value too — the switch rule fires on it at once (lz77.c:509-519: bestlength >= 1 >=
      // same), so the walk IS the second hash's order from that candidate (inclusive) downwards.
      u32 h_v = 0, cc_v = c_v, idx_v = r0_v, lo_v = bk0_v & 0xffffu;
      {
        u32 cch = c_v, ci = 0xffffffffu;           // chunk and index in sorted[0] of the nearest candidate
        if (r0_v > (bk0_v & 0xffffu)) ci = r0_v - 1;
        else if (c_v && (bk0p_v >> 16) > (bk0p_v & 0xffffu)) { cch = c_v - 1; ci = (bk0p_v >> 16) - 1; }
        if (have && ci != 0xffffffffu && same_v <= 1u) {
          const u32 si = cch * 32768u + ci;
          const u32 off = g_sorted0[si];
          const u32 s8 = g_ssame[si];
          const bool inwin = cch == c_v || off > o_v;
          if (inwin && s8 == (same_v & 255u)) {
            h_v = 1;
            cc_v = cch;
            idx_v = (u32)g_rank1[cch * 32768u + off] + 1u;
            lo_v = (cch == c_v ? bk1_v : bk1p_v) & 0xffffu;
          }
        }
      }

      for (u32 u = 0; u < npos; ++u) {
        // ---- one position, the whole wave
        const u32 lp = rdlane_u32(lp_v, u);
        const u32 size_rem = rdlane_u32(rem_v, u);
        const u32 same_p = rdlane_u32(same_v, u);
        const u32 byte0 = rdlane_u32(b012, u) & 255u;
        u32* const rec = rec0 + (u64)(first + u) * 8;
        if (size_rem < 3) {                            // lz77.c:440-446
          if (lane < 8) rec[lane] = lane == 1 ? (same_p | (byte0 << 16)) : 0u;
          continue;
        }
        const u32 limit = size_rem < ZMX_MAX_MATCH ? size_rem : ZMX_MAX_MATCH;   // lz77.c:448-450
        const u32 cp_chunk = rdlane_u32(c_v, u), op = rdlane_u32(o_v, u);
        u32 h = rdlane_u32(h_v, u), cc = rdlane_u32(cc_v, u), idx = rdlane_u32(idx_v, u), lo = rdlane_u32(lo_v, u);
        const u32 bkp0 = rdlane_u32(bk0p_v, u), bk1 = rdlane_u32(bk1_v, u), bkp1 = rdlane_u32(bk1p_v, u);
        u32 bestlen = 1, bestdist = 0, ncp = 0, hits_left = ZMX_MAX_CHAIN_HITS;
        u32 foff = 0, fmask = 0xffffu;                 // the filter: bytes [foff, foff + 4) under fmask must equal pos's
        u32 pbytes = rdlane_u32(b012, u);
        const u32 kp = li0 + first + u;                // region index of the position (same[] of a candidate: kp - dist)
        if (lane < 8) srec[lane] = 0;

        // One batch of up to 64 candidates in visit order: lane i holds entry idx - 1 - i of the slice (its offset in
        // chunk cc; s8 = its same & 255, first hash only), the first nb_in of them exist.  Returns 0: all of them dealt
        // with, the walk goes on in this slice (idx, hits_left moved); 1: the walk is over; 2: it switched to the second
        // hash (the cursor h / idx / lo was put there).
        auto batch = [&](u32 off, u32 s8, u32 nb_in) -> u32 {
          u32 nb = nb_in;
          bool ended = false;
          if (cc != cp_chunk) {
            // previous chunk: only candidates less than 32768 back (lz77.c:464); offsets fall along the lanes
            const u64 ok = __ballot(lane < nb && off > op);
            const u32 nv = (u32)__popcll(ok);
            ended = nv < nb;
            nb = nv;
            if (nb == 0) return 1u;
          }
          const bool act = lane < nb;
          const u32 dist = (cp_chunk - cc) * 32768u + op - off;
          const u32 lc = lp - dist;
          if (PROF) { n_hits += nb; ++n_batches; }
          // ---- the candidate's length: nothing for one that cannot beat bestlen (lz77.c:478-479, four bytes wide)
          u32 len = 0;
          {
            const u32 cw = m3_lds_u32(win, act ? lc + foff : 0u);
            bool go = act && ((cw ^ pbytes) & fmask) == 0;
            u64 m_go = __ballot(go);
            if (m_go != 0) {
              u32 cur = 0;
              if (same_p > 2u) {
                // lz77.c:481-490: inside a run both sides repeat their first byte — skip what same[] vouches for
                if (go && lds_byte(win, lc) == byte0) {
                  const u32 lz = g_same[kp - dist];
                  const u32 sk = same_p < lz ? same_p : lz;
                  cur = sk < limit ? sk : limit;
                  go = cur < limit;
                }
              }
              // (the whole wave comparing one passed candidate at a time, 264 bytes in one step, was tried for batches
              //  with few of them: 1.5 - 2x slower on the long-list classes — their common prefixes are short)
              if (PROF) n_pass += (u32)__popcll(m_go);
              while (__any(go)) {                      // GetMatch (lz77.c:297), 8 bytes per step
                if (PROF) ++n_lcp;
                if (go) {
                  const u64 x = m3_lds_u64(win, lp + cur) ^ m3_lds_u64(win, lc + cur);
                  const u32 m = x ? (u32)(__ffsll((unsigned long long)x) - 1) >> 3 : 8u;
                  cur += m;
                  go = m == 8 && cur < limit;
                }
              }
              len = cur < limit ? cur : limit;
              if (!act || ((m_go >> lane) & 1ull) == 0) len = 0;
            }
            // nobody beats bestlen and nobody satisfies the switch rule (lz77.c:509-519: it needs no improvement, only
            // bestlength >= same and the second hash value): the batch changes nothing but the counts — most batches of
            // a long list
            u64 m_ev = __ballot(len > bestlen);
            if (h == 0 && bestlen >= same_p) m_ev |= __ballot(act && s8 == (same_p & 255u));
            if (m_ev == 0) {
              if (PROF) ++n_quiet;
              hits_left -= nb;
              if (hits_left == 0 || ended) return 1u;
              idx -= nb;
              return 0u;
            }
          }
          // ---- in visit order: running maximum, change points, where the walk stops or switches
          const u32 incl = wave_scan_max(len);
          u32 excl = (u32)__builtin_amdgcn_update_dpp(0, (int)incl, 0x138, 0xf, 0xf, false);   // wave_shr 1
          excl = excl > bestlen ? excl : bestlen;
          const u32 run = incl > bestlen ? incl : bestlen;
          const u64 m_stop = __ballot(act && run >= limit);
          u64 m_sw = 0;
          if (h == 0) m_sw = __ballot(act && run >= same_p && s8 == (same_p & 255u));
          const u32 l_stop = m_stop ? (u32)__ffsll((unsigned long long)m_stop) - 1u : 64u;
          const u32 l_sw = m_sw ? (u32)__ffsll((unsigned long long)m_sw) - 1u : 64u;
          u32 ne = nb;                                 // candidates of this batch that the walk visits
          if (l_stop < ne) ne = l_stop + 1;
          const bool sw = l_sw < ne && l_sw < l_stop;  // (at the limit the walk breaks before the switch test, lz77.c:505)
          if (sw) ne = l_sw + 1;
          // (a 2-byte "match" moves bestlength like any other, but sublen[2] is never read and the record has no slot for it)
          const u64 m_up = __ballot(lane < ne && len > excl);
          if (m_up) {
            const u64 m_cp = m_up & __ballot(len >= 3);
            if ((m_cp >> lane) & 1ull) {
              const u32 slot = ncp + (u32)__popcll(m_cp & lt_mask);
              scr[slot] = len | (dist << 16);
              if (slot < 8) {
                u8* r8 = reinterpret_cast<u8*>(srec) + 8u + 3u * slot;
                r8[0] = (u8)(len - 3u); r8[1] = (u8)dist; r8[2] = (u8)(dist >> 8);
              }
            }
            const u32 last = 63u - (u32)__builtin_clzll(m_up);
            bestlen = rdlane_u32(len, last);
            bestdist = rdlane_u32(dist, last);
            ncp += (u32)__popcll(m_cp);
            foff = bestlen >= 3 ? bestlen - 3u : 0u;
            fmask = bestlen >= 3 ? 0xffffffffu : 0xffffffu;
            pbytes = m3_sgpr(m3_lds_u32(win, lp + foff));
          }
          if (l_stop < nb && !sw) return 1u;           // the limit is reached
          if (sw) {
            hits_left -= l_sw + 1;
            if (hits_left == 0) return 1u;
            // on in the second hash's order, just below this candidate (its chunk stays cc)
            const u32 off_sw = rdlane_u32(off, l_sw);
            h = 1;
            idx = m3_sgpr((u32)g_rank1[cc * 32768u + off_sw]);
            lo = (cc == cp_chunk ? bk1 : bkp1) & 0xffffu;
            return 2u;
          }
          hits_left -= nb;
          if (hits_left == 0 || ended) return 1u;
          idx -= nb;
          return 0u;
        };
        for (;;) {
          if (idx == lo) {
            // this chunk's part of the slice is used up: the previous chunk's bucket, from its end
            if (cc != cp_chunk || cc == 0) break;
            cc = cp_chunk - 1;
            const u32 e = h ? bkp1 : bkp0;
            lo = e & 0xffffu;
            idx = e >> 16;
            if (idx == lo) break;
          }
          u32 nb = idx - lo < 64u ? idx - lo : 64u;
          nb = nb < hits_left ? nb : hits_left;
          const u32 si = cc * 32768u + idx - 1u - (lane < nb ? lane : 0u);
          const u32 off = (h ? g_sorted1 : g_sorted0)[si];
          u32 s8 = 0;
          if (h == 0) s8 = g_ssame[si];
          // Tried on top of this and measured (profiles/r03_match_ab.txt), none of them a gain across the classes:
          //  - the entries of the next batches requested ahead (register sets refilled in turn, also across positions):
          //    same time on long lists, 25 % slower on text;
          //  - a test of 256 entries at a time (four loads, the filter, the first 8 bytes) that skips the quarters in which
          //    no candidate can change anything: 87 % of class P's batches skipped for 8 % of the time, B 45 % slower;
          //  - the whole wave comparing one passed candidate at a time, 264 bytes in one step: 1.5 - 2x slower, the
          //    common prefixes are short (one 8-byte step per batch on average).
          // PMC on class P: ~115 instructions per batch, 66 scalar; each wave issues one every ~12 cycles outside its
          // wait for the load — a chain of scalar / vector round trips and taken branches that 6 waves per SIMD (the
          // window's 37 KiB of LDS allow no more) do not cover.
          if (batch(off, s8, nb) == 1u) break;
        }
        // ---- the record (same layout as k_match2's)
        wave_lds_sync();
        {
          const u32 w0 = bestlen | (bestdist << 16);
          if (ncp <= 8) {
            const u32 w1 = same_p | (byte0 << 16) | (ncp << 24);
            if (lane < 8) rec[lane] = lane == 0 ? w0 : lane == 1 ? w1 : srec[lane];
          } else {
            u32 poff = 0;
            if (lane == 0) poff = atomicAdd(&P.counters[0], ncp);
            poff = m3_sgpr(poff);
            const u32 w1 = same_p | (byte0 << 16) | (0xffu << 24);
            if (poff + ncp <= P.pool_cap) {
              for (u32 e = lane; e < ncp; e += 64) P.pool[poff + e] = scr[e];
              if (lane < 8) rec[lane] = lane == 0 ? w0 : lane == 1 ? w1 : lane == 2 ? poff : lane == 3 ? ncp : srec[lane];
            } else {
              if (lane == 0) atomicOr(&P.counters[1], 1u);   // host retries with a larger pool
              if (lane < 8) rec[lane] = lane == 0 ? w0 : lane == 1 ? w1 : lane < 4 ? 0u : srec[lane];
            }
          }
        }
        wave_lds_sync();
      }
    }
  }
  if (PROF) {
    atomicAdd(reinterpret_cast<unsigned long long*>(P.counters + 4), (unsigned long long)(lane == 0 ? n_hits : 0));
    atomicAdd(reinterpret_cast<unsigned long long*>(P.counters + 6), (unsigned long long)(lane == 0 ? n_batches : 0));
    atomicAdd(reinterpret_cast<unsigned long long*>(P.counters + 16), (unsigned long long)(lane == 0 ? n_pass : 0));
    atomicAdd(reinterpret_cast<unsigned long long*>(P.counters + 18), (unsigned long long)(lane == 0 ? n_lcp : 0));
    atomicAdd(reinterpret_cast<unsigned long long*>(P.counters + 20), (unsigned long long)(lane == 0 ? n_quiet : 0));
  }
}

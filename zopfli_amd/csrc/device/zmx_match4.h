// k_match4: the match table (ZopfliFindLongestMatch(limit 258, sublen) for every position, lz77.c:407-542) on
// k_bucket's sorted candidate slices (zmx_match3.h), ONE LANE PER POSITION, the slices streamed.  Included only
// by zmx_hip.hip, after zmx_match3.h (same parameters, same record format, same window staging).
//
// k_match2 walks prev links: one dependent 8-byte load and ~70 instructions of the wave per candidate step.
// k_match3 gives a position a whole wave: right for lists of thousands, but a list of text has 20..500 entries
// and every 64 of them cost a wave a memory round trip and ~100 instructions.  Here a lane keeps its position
// (as in k_match2: 64 walks per wave hide each other's latency) but its candidates are a CONTIGUOUS slice of
// sorted[] read downwards, so nothing is chased: 8 entries per 16-byte load, requested two steps ahead, and the
// usual step tests FOUR candidates at once against the LDS window — the four bytes that end at the best length
// so far (lz77.c:478-479 generalised: a candidate that fails cannot be longer) — and moves on when all four
// fail, which is what almost every candidate of a long list does.  A candidate that passes, the end of a slice,
// the window's edge (lz77.c:464), the hit cap (:527-530) and everything about the first hash (:509-519) go
// through the one-candidate step, the byte compare (GetMatch, :297, with the same[] skip of :481-490) through
// a side block as in k_match2.
//
// The switch to the second hash without a per-candidate lookup.  The rule (:509-519) fires at the first candidate
// of the first hash's chain — once bestlength >= same[pos] — whose second hash value equals pos's.  Both hash
// values equal means: a member of pos's bucket in the SECOND hash's order whose 3-byte hash equals pos's.  Those
// are found by walking that order down from pos (normally the very next entry); the set-up does it once per
// position ("qs": chunk, index, offset of the next such candidate) and the first-hash walk only compares offsets
// with it.  If qs is also the nearest candidate of the first hash and same[pos] <= 1 — text, nearly always — the
// whole walk is the second hash's order from qs on: the first-hash phase disappears.
#pragma once

#define M4_THREADS 512u
#define M4_IDLE 0u      // needs a position
#define M4_WALK 1u      // at a candidate, not compared yet
#define M4_CMP 2u       // comparing bytes with the candidate
#define M4_PEND 3u      // walk ended, record not written yet
#define M4_DONE 4u      // the tile has no more positions
#define M4_QS_NONE 0xffffu

__device__ __forceinline__ u32 m4_hash3(u32 b012) {
  return (((b012 & 255u) << 10) ^ (((b012 >> 8) & 255u) << 5) ^ ((b012 >> 16) & 255u)) & 32767u;
}

// The next member of pos's second-hash bucket below (chunk flag cprev, index i) that is less than 32768 back and
// has pos's 3-byte hash: the next candidate the switch rule can fire at.  Returns idx | off << 16 | cprev << 31,
// or M4_QS_NONE in the low half.  own_lo: the bucket's start in pos's own chunk; pb: its start | end << 16 in
// the previous chunk (0 = none).  Rarely more than one step.
__device__ __forceinline__ u32 m4_qs_next(const u16* srt1, u32 c, u32 cprev, u32 i, u32 own_lo, u32 pb, u32 op,
                                          const u32* win, u32 lp, u32 key0) {
  for (;;) {
    if (!cprev) {
      if (i == own_lo) {
        if (c == 0 || (pb >> 16) == (pb & 0xffffu)) return M4_QS_NONE;
        cprev = 1;
        i = pb >> 16;
      }
    }
    if (cprev && i == (pb & 0xffffu)) return M4_QS_NONE;
    --i;
    const u32 off = srt1[(c - cprev) * 32768u + i];
    if (cprev && off <= op) return M4_QS_NONE;            // 32768 or more back: so is everything below
    const u32 dist = cprev * 32768u + op - off;
    if (m4_hash3(m3_lds_u32(win, lp - dist)) == key0) return i | (off << 16) | (cprev << 31);
  }
}

template <bool PROF>
__global__ __launch_bounds__(M4_THREADS, 6) void k_match4(Match3Params P) {
  __shared__ __align__(16) u32 win[MWIN_BYTES / 4 + 12];
  __shared__ uint2 s_desc[MT];
  __shared__ u32 s_next, s_tile;

  const u32 tid = threadIdx.x;
  const u32 xcd = blockIdx.x & 7;
  const long long srt_delta = P.sorted[1] - P.sorted[0];      // the two orders live in one allocation
  u32* my_scratch = P.scratch + ((u64)blockIdx.x * M4_THREADS + tid) * SCRATCH_CPS;   // change points beyond the eighth

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
    for (u32 v = tid; v < nvec + 1; v += M4_THREADS) {
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
    const u32 rem0 = (u32)((bd.inend - p0 < 70000) ? bd.inend - p0 : 70000);   // bytes to the block end, saturated
    u32* const rec0 = P.recs + (bd.pos_off + (p0 - bd.instart)) * 8;
    const u16* const g_same = P.same16 + bd.reg_off;
    const u16* const srt0 = P.sorted[0] + bd.reg_off;
    const u16* const srt1 = P.sorted[1] + bd.reg_off;
    const u16* const g_rank0 = P.rank[0] + bd.reg_off;
    const u16* const g_rank1 = P.rank[1] + bd.reg_off;
    const u32* const g_bucket0 = P.bucket[0] + (u64)P.chunk_base[lo_b] * 32768u;
    const u32* const g_bucket1 = P.bucket[1] + (u64)P.chunk_base[lo_b] * 32768u;

    // ---- set-up, lane = position, for the whole tile: where each walk starts
    //   x = idx | lo << 16 | h << 31         the slice [lo, idx) of sorted[h] the walk reads downwards first
    //   y = h = 1: cprev | key1 << 1         (the slice lies in the previous chunk; pos's second hash value)
    //       h = 0: qs (m4_qs_next)           the next candidate the switch rule can fire at
    for (u32 i = tid; i < ntile; i += M4_THREADS) {
      const u32 kp = li0 + i, lp = lp0 + i;
      uint2 ds = make_uint2(1u << 31, M4_QS_NONE);       // h = 1 with an empty slice: no candidate at all
      if (rem0 - i >= 3) {
        const u32 same_p = g_same[kp];
        const u32 b012 = m3_lds_u32(win, lp);
        const u32 key0 = m4_hash3(b012);
        const u32 key1 = key0 ^ ((same_p - 3u) & 255u);
        const u32 c = kp >> 15, op = kp & 32767u;
        const u32 r0 = g_rank0[kp], r1 = g_rank1[kp];
        const u32 lo0 = g_bucket0[(u64)c * 32768u + key0] & 0xffffu;
        const u32 lo1 = g_bucket1[(u64)c * 32768u + key1] & 0xffffu;
        const u32 pb0 = c ? g_bucket0[(u64)(c - 1) * 32768u + key0] : 0u;
        const u32 pb1 = c ? g_bucket1[(u64)(c - 1) * 32768u + key1] : 0u;
        // the nearest candidate of the first hash
        u32 c0prev = 0, i0 = r0;
        bool have0 = r0 > lo0;
        if (!have0 && c && (pb0 >> 16) > (pb0 & 0xffffu)) { c0prev = 1; i0 = pb0 >> 16; have0 = true; }
        u32 off0 = 0;
        if (have0) {
          off0 = srt0[(c - c0prev) * 32768u + i0 - 1];
          if (c0prev && off0 <= op) have0 = false;       // 32768 or more back
        }
        if (have0) {
          const u32 qs = m4_qs_next(srt1, c, 0u, r1, lo1, pb1, op, win, lp, key0);
          const bool qs_ok = (qs & 0xffffu) != M4_QS_NONE;
          if (same_p <= 1u && qs_ok && (qs >> 31) == c0prev && ((qs >> 16) & 32767u) == off0) {
            // the walk is the second hash's order from qs (inclusive) on
            const u32 qprev = qs >> 31;
            ds.x = ((qs & 0xffffu) + 1u) | ((qprev ? (pb1 & 0xffffu) : lo1) << 16) | (1u << 31);
            ds.y = qprev | (key1 << 1);
          } else {
            ds.x = r0 | (lo0 << 16);
            ds.y = qs;
          }
        }
      }
      s_desc[i] = ds;
    }
    __syncthreads();

    // ---- per-lane walk state
    u32 st = M4_IDLE;
    u32 lp = 0, kp = 0, op = 0, cch = 0;      // the position: LDS offset, region index, offset in its chunk, its chunk
    u32 idx = 0, lo = 0, h = 0, cprev = 0;    // the slice being read: sorted[h], chunk cch - cprev, entries [lo, idx) downwards
    u32 ebase = 0;                            // E holds entries ebase .. ebase + 7 of that chunk's order, N the eight below
    uint4 E = make_uint4(0, 0, 0, 0), N = make_uint4(0, 0, 0, 0);
    u32 qs = M4_QS_NONE, pbk = 0, key1 = 0;   // h = 0: the switch candidate; the current hash's bucket in the previous chunk
    u32 limit = 0, bestlen = 0, bestdist = 0, ncp = 0, same_p = 0, cur = 0, size_rem = 0, hits_left = 0;
    u32 byte0 = 0, pbyte = 0, foff = 0, fmask = 0;
    u32 off = 0, dist = 0, lc = 0;            // the candidate at hand
    u64 cw0 = 0, cw1 = 0, cw2 = 0;
    u32 n_hits = 0, n_iter = 0;

    // 16 bytes of sorted[hh] of chunk (cch - cp) from entry eb on (eb may be "negative": the arrays are padded in front)
    auto load8 = [&](u32 hh, u32 cp, u32 eb) -> uint4 {
      const u16* base = srt0 + (hh ? srt_delta : 0ll);
      uint4 r;
      __builtin_memcpy(&r, base + (long long)((cch - cp) * 32768u) + (long long)(int)eb, 16);
      return r;
    };
    auto entry_of = [&](u32 k) -> u32 {       // entry k (0..7) of E
      const u32 w = k < 4 ? (k < 2 ? E.x : E.y) : (k < 6 ? E.z : E.w);
      return (k & 1u) ? w >> 16 : w & 0xffffu;
    };

    for (;;) {
      // ---- service: record writes and refills, queued (MATCH_BATCH lanes, or nobody left walking)
      const u64 m_need = __ballot(st == M4_PEND || st == M4_IDLE);
      const u64 m_run = __ballot(st == M4_WALK || st == M4_CMP);
      if (m_need != 0 && ((u32)__popcll(m_need) >= MATCH_BATCH || m_run == 0)) {
        if (st == M4_PEND) {
          st = M4_IDLE;
          uint4 r0, r1;
          r0.x = bestlen | (bestdist << 16);
          r0.z = (u32)cw0; r0.w = (u32)(cw0 >> 32);
          r1.x = (u32)cw1; r1.y = (u32)(cw1 >> 32); r1.z = (u32)cw2; r1.w = (u32)(cw2 >> 32);
          if (ncp <= 8) {
            r0.y = same_p | (byte0 << 16) | (ncp << 24);
          } else {
            r0.y = same_p | (byte0 << 16) | (0xffu << 24);
            const u32 poff = atomicAdd(&P.counters[0], ncp);
            if (poff + ncp <= P.pool_cap) {
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                // entry e = bits 24 e .. 24 e + 23 of cw2 : cw1 : cw0
                const u32 bit = 24u * (u32)e, wi = bit >> 6, sh = bit & 63u;
                const u64 a = wi == 0 ? cw0 : wi == 1 ? cw1 : cw2, b2 = wi == 0 ? cw1 : cw2;
                const u32 v = (u32)((sh > 40 ? (a >> sh) | (b2 << (64u - sh)) : a >> sh) & 0xffffffu);
                P.pool[poff + e] = ((v & 255u) + 3u) | ((v >> 8) << 16);
              }
              for (u32 e = 8; e < ncp; ++e) P.pool[poff + e] = my_scratch[e];
              r0.z = poff;
              r0.w = ncp;
            } else {
              atomicOr(&P.counters[1], 1u);  // host retries with a larger pool
              r0.z = 0;
              r0.w = 0;
            }
          }
          u32* const rec = rec0 + (u64)(lp - lp0) * 8;
          reinterpret_cast<uint4*>(rec)[0] = r0;
          reinterpret_cast<uint4*>(rec)[1] = r1;
        }
        if (st == M4_IDLE) {
          const u32 pi = atomicAdd(&s_next, 1u);
          if (pi >= ntile) {
            st = M4_DONE;
          } else {
            lp = lp0 + pi;
            kp = li0 + pi;
            cch = kp >> 15;
            op = kp & 32767u;
            size_rem = rem0 - pi;                  // saturated: only compared with 3 and 258
            u32* const rec = rec0 + (u64)pi * 8;
            same_p = g_same[kp];
            byte0 = lds_byte(win, lp);
            ncp = 0;
            cw0 = 0; cw1 = 0; cw2 = 0;
            bestlen = 1; bestdist = 0; hits_left = ZMX_MAX_CHAIN_HITS;
            const uint2 ds = s_desc[pi];
            idx = ds.x & 0xffffu;
            lo = (ds.x >> 16) & 32767u;
            h = ds.x >> 31;
            if (size_rem < 3) {                    // lz77.c:440-446
              rec[0] = 0;
              rec[1] = same_p | (byte0 << 16);
            } else if (idx == lo && (h || cch == 0)) {
              // no candidate at all (an empty first slice with h = 0 still has the previous chunk to look at)
              rec[0] = 1;
              rec[1] = same_p | (byte0 << 16);
            } else {
              limit = size_rem < ZMX_MAX_MATCH ? size_rem : ZMX_MAX_MATCH;  // lz77.c:448-450
              const u32 key0 = m4_hash3(m3_lds_u32(win, lp));
              if (h) {
                cprev = ds.y & 1u;
                key1 = (ds.y >> 1) & 32767u;
                qs = M4_QS_NONE;
                pbk = (cch && !cprev) ? g_bucket1[(u64)(cch - 1) * 32768u + key1] : 0u;
              } else {
                cprev = 0;
                qs = ds.y;
                key1 = key0 ^ ((same_p - 3u) & 255u);
                pbk = cch ? g_bucket0[(u64)(cch - 1) * 32768u + key0] : 0u;
                if (idx == lo) {                   // the own chunk has nothing: the previous chunk's bucket, from its end
                  lo = pbk & 0xffffu;
                  idx = pbk >> 16;
                  cprev = 1;
                }
              }
              if (idx == lo) {
                rec[0] = 1;
                rec[1] = same_p | (byte0 << 16);
              } else {
                ebase = idx - 8u;
                E = load8(h, cprev, ebase);
                N = load8(h, cprev, ebase - 8u);
                pbyte = m3_lds_u32(win, lp); foff = 0; fmask = 0xffffu;   // bestlength 1: bytes 0 and 1
                st = M4_WALK;
              }
            }
          }
        }
        continue;   // states changed: take the ballots again
      }
      if (m_run == 0) {
        if (m_need == 0) break;                    // every lane is done
        continue;
      }
      if (PROF) ++n_iter;

      // ---- the usual step: four candidates of the second hash's order at once; all fail the filter -> on
      const u32 kE = idx - ebase;                  // entries of E not yet visited (the next one is kE - 1)
      bool single = st == M4_WALK;
      {
        const bool hi8 = kE == 8u;
        const u32 w_hi = hi8 ? E.w : E.y, w_lo = hi8 ? E.z : E.x;
        const u32 o_a = w_hi >> 16, o_b = w_hi & 0xffffu, o_c = w_lo >> 16, o_d = w_lo & 0xffffu;   // in visit order
        const bool can4 = st == M4_WALK && h == 1u && (kE == 8u || kE == 4u) && idx - lo >= 4u && hits_left > 4u &&
                          (!cprev || o_d > op);
        if (__any(can4)) {
          const u32 a0 = lp - (cprev * 32768u + op) + foff;     // candidate's LDS offset + foff = a0 + its chunk offset
          const u32 x_a = (m3_lds_u32(win, can4 ? a0 + o_a : 0u) ^ pbyte) & fmask;
          const u32 x_b = (m3_lds_u32(win, can4 ? a0 + o_b : 0u) ^ pbyte) & fmask;
          const u32 x_c = (m3_lds_u32(win, can4 ? a0 + o_c : 0u) ^ pbyte) & fmask;
          const u32 x_d = (m3_lds_u32(win, can4 ? a0 + o_d : 0u) ^ pbyte) & fmask;
          const u32 mn = min(min(x_a, x_b), min(x_c, x_d));
          if (can4 && mn != 0u) {                  // none can beat bestlength (lz77.c:478-479)
            idx -= 4u;
            hits_left -= 4u;
            single = false;
            if (PROF) n_hits += 4u;
            if (idx == ebase) {                    // E is used up: the eight below, and ask for the next eight
              E = N;
              ebase -= 8u;
              N = load8(h, cprev, ebase - 8u);
            }
            if (idx == lo) {
              // this chunk's part of the slice is used up: the previous chunk's bucket, from its end
              if (!cprev && cch && (pbk >> 16) > (pbk & 0xffffu)) {
                lo = pbk & 0xffffu;
                idx = pbk >> 16;
                cprev = 1;
                ebase = idx - 8u;
                E = load8(h, 1u, ebase);
                N = load8(h, 1u, ebase - 8u);
              } else {
                st = M4_PEND;
              }
            }
          }
        }
      }

      // ---- the one-candidate step
      bool ev = false;                             // a candidate has been dealt with: move on
      bool fin = false;
      if (single) {
        off = entry_of(kE - 1u);
        if (cprev && off <= op) {
          st = M4_PEND;                            // 32768 or more back: the walk ends (lz77.c:464)
          single = false;
        }
      }
      if (single) {
        dist = cprev * 32768u + op - off;
        lc = lp - dist;
        if (PROF) ++n_hits;
        const u32 cwv = m3_lds_u32(win, lc + foff);
        const bool pass = ((cwv ^ pbyte) & fmask) == 0;
        ev = !pass;
        if (pass) {
          // lz77.c:481-490: skip the common run (pure acceleration)
          cur = 0;
          if (same_p > 2 && lds_byte(win, lc) == byte0) {
            const u32 lz = g_same[kp - dist];
            const u32 s = same_p < lz ? same_p : lz;
            cur = s < limit ? s : limit;
          }
          st = M4_CMP;
        }
      }
      if (st == M4_CMP) {  // GetMatch (lz77.c:297), 8 bytes per step
        const u32 rem = limit - cur;
        bool end = rem == 0;
        if (!end) {
          const u64 x = m3_lds_u64(win, lp + cur) ^ m3_lds_u64(win, lc + cur);
          u32 m = x ? (u32)(__ffsll((unsigned long long)x) - 1) >> 3 : 8u;
          if (m > rem) m = rem;
          cur += m;
          end = m < 8 || cur >= limit;
        }
        if (end) {
          ev = true;
          st = M4_WALK;
          if (cur > bestlen) {  // lz77.c:495-505: new change point of sublen
            // (a 2-byte "match" only moves bestlength; sublen[2] is never read)
            if (cur >= 3) {
              if (ncp < 8) {
                const u64 v = (u64)((cur - 3u) | (dist << 8));          // 24 bits at bit 24 ncp of cw2 : cw1 : cw0
                const u32 bit = 24u * ncp, wi = bit >> 6, sh = bit & 63u;
                const u64 lo64 = v << sh, hi64 = sh > 40 ? v >> (64u - sh) : 0ull;
                cw0 |= wi == 0 ? lo64 : 0ull;
                cw1 |= wi == 1 ? lo64 : wi == 0 ? hi64 : 0ull;
                cw2 |= wi == 2 ? lo64 : wi == 1 ? hi64 : 0ull;
              } else if (ncp < SCRATCH_CPS) {
                my_scratch[ncp] = cur | (dist << 16);
              }
              ++ncp;
            }
            bestlen = cur;
            bestdist = dist;
            fin = cur >= limit;
            foff = cur >= 3 ? cur - 3u : 0u;
            fmask = cur >= 3 ? 0xffffffffu : 0xffffffu;    // (cur = 2: bytes 0..2)
            pbyte = m3_lds_u32(win, lp + foff);
          }
        }
      }
      if (ev) {
        if (fin) {
          st = M4_PEND;
        } else {
          bool moved = false;                      // the cursor was put somewhere else: E and N are to be read anew
          --hits_left;
          // lz77.c:509-519: the switch to the second hash, at the candidate the set-up found (qs)
          if (h == 0u && (qs & 0xffffu) != M4_QS_NONE && (qs >> 31) == cprev && ((qs >> 16) & 32767u) == off) {
            const u32 own1 = g_bucket1[(u64)cch * 32768u + key1];
            const u32 pb1 = cch ? g_bucket1[(u64)(cch - 1) * 32768u + key1] : 0u;
            if (bestlen >= same_p) {
              h = 1u;
              idx = qs & 0xffffu;                  // on just below qs in the second hash's order (its chunk stays)
              lo = (cprev ? pb1 : own1) & 0xffffu;
              pbk = pb1;
              moved = true;
            } else {
              // not yet: the next candidate with pos's second hash value
              qs = m4_qs_next(srt1, cch, cprev, qs & 0xffffu, own1 & 0xffffu, pb1, op, win, lp, m4_hash3(m3_lds_u32(win, lp)));
              --idx;
            }
          } else {
            --idx;
          }
          if (hits_left == 0u) {
            st = M4_PEND;                          // lz77.c:527-530
          } else {
            if (idx == lo) {
              if (!cprev && cch && (pbk >> 16) > (pbk & 0xffffu)) {
                lo = pbk & 0xffffu;
                idx = pbk >> 16;
                cprev = 1;
                moved = true;
              } else {
                st = M4_PEND;                      // lz77.c:521-523: the chain ends
              }
            }
            if (st != M4_PEND) {
              if (moved) {
                ebase = idx - 8u;
                E = load8(h, cprev, ebase);
                N = load8(h, cprev, ebase - 8u);
              } else if (idx == ebase) {
                E = N;
                ebase -= 8u;
                N = load8(h, cprev, ebase - 8u);
              }
            }
          }
        }
      }
    }
    if (PROF) {
      atomicAdd(reinterpret_cast<unsigned long long*>(P.counters + 4), (unsigned long long)n_hits);
      if ((tid & 63) == 0) atomicAdd(reinterpret_cast<unsigned long long*>(P.counters + 6), (unsigned long long)n_iter);
    }
  }
}

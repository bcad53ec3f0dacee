// k_match2: the match table kernel (ZopfliFindLongestMatch(limit 258, sublen) for every position,
// lz77.c:407-542) with the hot loop written for the VALU issue rate.  Included only by zmx_hip.hip,
// after zmx_kernels.h (same MatchParams, window staging and record format as k_match).
//
// k_match is VALU-bound (78 % of the issue rate, profiles/r01_v7) on a per-lane state machine whose
// compiled form spends most of its instructions on copies between divergent branches.  Here a step
// of the wave is straight-line code for the common case — the candidate fails the one-byte test
// (lz77.c:478-479) and the lane moves down its chain: one LDS byte, the hash-switch test
// (:509-519), one 8-byte link load — with the rare cases (byte compare :297, a new change point
// :495-505, writing the record, fetching the next position) as skipped-when-empty side blocks.
#pragma once

#ifndef M2_THREADS
#define M2_THREADS 512   // 8 waves share one staged window: twice k_match's waves per CU for the same LDS
#endif
#define M2_IDLE 0u      // needs a position
#define M2_WALK 1u      // at a candidate, not compared yet
#define M2_CMP 2u       // comparing bytes with the candidate
#define M2_PEND 3u      // walk ended, record not written yet
#define M2_DONE 4u      // the tile has no more positions

// PROF: counts the chain hits (candidates visited, lz77.c:464-530) and the iterations of the wave loop into
// P.counters[4..7] (two 64-bit sums) — ZOPFLI_AMD_PROF prints hits per position and cycles per hit.
//
// FILT: the one-byte test reads the four bytes that END at offset bestlength instead (the candidate cannot beat
// bestlength unless all of them match — a candidate the reference compares and then drops because it is not
// longer, lz77.c:494, is dropped here without the compare: same results, and the byte-compare side block runs in
// fewer of the wave's steps), and the compare reads 8 bytes per side with one unaligned LDS load each.
__device__ __forceinline__ u32 m2_lds_u32(const u32* win, u32 byte_off) {
  u32 x;
  __builtin_memcpy(&x, reinterpret_cast<const char*>(win) + byte_off, 4);
  return x;
}
__device__ __forceinline__ u64 m2_lds_u64(const u32* win, u32 byte_off) {
  u64 x;
  __builtin_memcpy(&x, reinterpret_cast<const char*>(win) + byte_off, 8);
  return x;
}

template <bool PROF, bool FILT>
__global__ __launch_bounds__(M2_THREADS, M2_THREADS == 512 ? 8 : 4) void k_match2(MatchParams P) {
  __shared__ __align__(16) u32 win[MWIN_BYTES / 4 + 4];
  __shared__ u32 s_next, s_tile;

  const u32 tid = threadIdx.x;
  const u32 xcd = blockIdx.x & 7;
  u32* my_scratch = P.scratch + ((u64)blockIdx.x * M2_THREADS + tid) * SCRATCH_CPS;

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

    // block of this tile: largest b with tile_off[b] <= tile
    u32 lo = 0, hi = P.nb;
    while (hi - lo > 1) {
      const u32 mid = (lo + hi) >> 1;
      if (P.tile_off[mid] <= tile) lo = mid; else hi = mid;
    }
    const BlockDesc bd = P.blocks[lo];
    if (P.skip_energy && P.skip_energy[lo] > P.skip_thr * (bd.inend - bd.ws)) continue;   // the skip-walk's block (zmx_match5.h)
    const u64 p0 = bd.instart + (u64)(tile - P.tile_off[lo]) * MT;
    const u64 p1 = (p0 + MT < bd.inend) ? p0 + MT : bd.inend;
    const u32 ntile = (u32)(p1 - p0);

    // stage bytes [p0 - 32768, p1 + 258) (clipped to [0, inend)) at LDS offset (abs - wb)
    const long long wb = ((long long)p0 - (long long)ZMX_WINDOW) & ~15ll;  // 16-byte aligned base, may be < 0
    const u64 hi_abs = (p1 + ZMX_MAX_MATCH < bd.inend) ? p1 + ZMX_MAX_MATCH : bd.inend;
    const u32 nvec = (u32)(((long long)hi_abs - wb + 15) >> 4);
    for (u32 v = tid; v < nvec; v += M2_THREADS) {
      const long long a = wb + (long long)v * 16;
      uint4 x = make_uint4(0, 0, 0, 0);
      if (a >= 0) x = *reinterpret_cast<const uint4*>(P.in + a);  // input is padded past its end
      reinterpret_cast<uint4*>(win)[v] = x;
    }
    __syncthreads();

    // index: abs - ws; {prev1 | prev2 << 16, same | ..}.  The base is the same for the whole workgroup: as a scalar
    // pair the walk's one dependent load takes a 32-bit lane offset instead of 64-bit address arithmetic per step.
    const uint2* lk;
    {
      const u64 a = reinterpret_cast<u64>(reinterpret_cast<const uint2*>(P.links + bd.reg_off));
      const u32 alo = (u32)__builtin_amdgcn_readfirstlane((int)(u32)a), ahi = (u32)__builtin_amdgcn_readfirstlane((int)(u32)(a >> 32));
      lk = reinterpret_cast<const uint2*>(((u64)ahi << 32) | alo);
    }
    const u32 li0 = (u32)(p0 - bd.ws);            // link index of the tile's first position
    const u32 lp0 = (u32)((long long)p0 - wb);    // its LDS byte offset
    const u32 rem0 = (u32)((bd.inend - p0 < 70000) ? bd.inend - p0 : 70000);   // bytes to the block end, saturated
    u32* const rec0 = P.recs + (bd.pos_off + (p0 - bd.instart)) * 8;

    // ---- per-lane walk state (all 32-bit)
    u32 st = M2_IDLE;
    u32 lp = 0, lc = 0;            // LDS byte offsets of the position and of the candidate
    u32 li = 0;                    // link index of the position
    u32 limit = 0, bestlen = 0, bestdist = 0, dist = 0, ncp = 0, same_pos = 0, cur = 0, size_rem = 0;
    u32 byte0 = 0, pbyte = 0;      // in[pos], in[pos + bestlen] (FILT: the four bytes in[pos + foff ..], of which fmask count)
    u32 foff = 0, fmask = 0;
    u32 hits_left = 0, chain = 1;
    uint2 L = make_uint2(0, 0);    // link record of the candidate
    u32 n_hits = 0, n_iter = 0;
    // the first 8 change points of sublen (3 bytes each: length - 3, distance) as they will lie in the record:
    // built in registers, written with the record in two 16-byte stores (byte stores into HBM as they were
    // found cost 9.2 GB of write traffic for 3.2 GB of records)
    u64 cw0 = 0, cw1 = 0, cw2 = 0;

    for (;;) {
      // ---- service: record writes and refills, queued (MATCH_BATCH lanes, or nobody left walking)
      const u64 m_need = __ballot(st == M2_PEND || st == M2_IDLE);
      const u64 m_run = __ballot(st == M2_WALK || st == M2_CMP);
      if (m_need != 0 && ((u32)__popcll(m_need) >= MATCH_BATCH || m_run == 0)) {
        if (st == M2_PEND) {
          st = M2_IDLE;
          uint4 r0, r1;
          r0.x = bestlen | (bestdist << 16);
          r0.z = (u32)cw0; r0.w = (u32)(cw0 >> 32);
          r1.x = (u32)cw1; r1.y = (u32)(cw1 >> 32); r1.z = (u32)cw2; r1.w = (u32)(cw2 >> 32);
          if (ncp <= 8) {
            r0.y = same_pos | (byte0 << 16) | (ncp << 24);
          } else {
            r0.y = same_pos | (byte0 << 16) | (0xffu << 24);
            const u32 off = atomicAdd(&P.counters[0], ncp);
            if (off + ncp <= P.pool_cap) {
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                // entry e = bits 24 e .. 24 e + 23 of cw2 : cw1 : cw0
                const u32 bit = 24u * (u32)e, wi = bit >> 6, sh = bit & 63u;
                const u64 a = wi == 0 ? cw0 : wi == 1 ? cw1 : cw2, b2 = wi == 0 ? cw1 : cw2;
                const u32 v = (u32)((sh > 40 ? (a >> sh) | (b2 << (64u - sh)) : a >> sh) & 0xffffffu);
                P.pool[off + e] = ((v & 255u) + 3u) | ((v >> 8) << 16);
              }
              for (u32 e = 8; e < ncp; ++e) P.pool[off + e] = my_scratch[e];
              r0.z = off;
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
        if (st == M2_IDLE) {
          const u32 idx = atomicAdd(&s_next, 1u);
          if (idx >= ntile) {
            st = M2_DONE;
          } else {
            lp = lp0 + idx;
            li = li0 + idx;
            size_rem = rem0 - idx;               // saturated: only compared with 3 and 258
            u32* const rec = rec0 + (u64)idx * 8;
            const uint2 Lp = lk[li];
            same_pos = Lp.y & 0xffffu;
            byte0 = lds_byte(win, lp);
            ncp = 0;
            cw0 = 0; cw1 = 0; cw2 = 0;
            bestlen = 1; bestdist = 0; chain = 1; hits_left = ZMX_MAX_CHAIN_HITS;
            if (size_rem < 3) {                      // lz77.c:440-446
              rec[0] = 0;
              rec[1] = same_pos | (byte0 << 16);
            } else {
              limit = size_rem < ZMX_MAX_MATCH ? size_rem : ZMX_MAX_MATCH;  // lz77.c:448-450
              dist = Lp.x & 0xffffu;
              if (dist == 0) {                       // empty chain
                rec[0] = 1;
                rec[1] = same_pos | (byte0 << 16);
              } else {
                lc = lp - dist;
                L = lk[li - dist];
                if (FILT) { pbyte = m2_lds_u32(win, lp); foff = 0; fmask = 0xffffu; }   // bestlength 1: bytes 0 and 1
                else pbyte = lds_byte(win, lp + 1);
                st = M2_WALK;
              }
            }
          }
        }
        continue;   // states changed: take the ballots again
      }
      if (m_run == 0) {
        if (m_need == 0) break;                      // every lane is done
        continue;
      }

      // ---- the candidate's one-byte test (lz77.c:478-479)
      const bool walk = st == M2_WALK;
      bool pass;
      if (FILT) {
        const u32 cw = m2_lds_u32(win, walk ? lc + foff : 0u);
        // (bestlength >= size - pos, the other half of lz77.c:478, cannot hold here: bestlength <= limit <= size - pos,
        // and the walk ends when bestlength reaches limit)
        pass = walk && ((cw ^ pbyte) & fmask) == 0;
      } else {
        const u32 cb = lds_byte(win, walk ? lc + bestlen : 0u);
        pass = walk && (bestlen >= size_rem || cb == pbyte);
      }
      bool ev = walk && !pass;                       // rejected: nothing to record, move on
      if (pass) {
        // lz77.c:481-490: skip the common run (pure acceleration)
        cur = 0;
        if (same_pos > 2 && lds_byte(win, lc) == byte0) {
          const u32 lz = L.y & 0xffffu;
          const u32 s = same_pos < lz ? same_pos : lz;
          cur = s < limit ? s : limit;
        }
        st = M2_CMP;
      }
      bool fin = false;
      if (st == M2_CMP) {  // GetMatch (lz77.c:297), 8 bytes per step (markup and source code: matches of 50+ bytes)
        const u32 rem = limit - cur;
        bool end = rem == 0;
        if (!end && FILT) {
          const u64 x = m2_lds_u64(win, lp + cur) ^ m2_lds_u64(win, lc + cur);
          u32 m = x ? (u32)(__ffsll((unsigned long long)x) - 1) >> 3 : 8u;
          if (m > rem) m = rem;
          cur += m;
          end = m < 8 || cur >= limit;
        } else if (!end) {
          const u32 a0 = (lp + cur) >> 2, b0 = (lc + cur) >> 2, as = (lp + cur) & 3u, bs = (lc + cur) & 3u;
          const u32 a_lo = win[a0], a_mi = win[a0 + 1], a_hi = win[a0 + 2];
          const u32 b_lo = win[b0], b_mi = win[b0 + 1], b_hi = win[b0 + 2];
          const u32 x0 = __builtin_amdgcn_alignbyte(a_mi, a_lo, as) ^ __builtin_amdgcn_alignbyte(b_mi, b_lo, bs);
          const u32 x1 = __builtin_amdgcn_alignbyte(a_hi, a_mi, as) ^ __builtin_amdgcn_alignbyte(b_hi, b_mi, bs);
          u32 m = x0 ? (u32)(__ffs((int)x0) - 1) >> 3 : x1 ? 4u + ((u32)(__ffs((int)x1) - 1) >> 3) : 8u;
          if (m > rem) m = rem;
          cur += m;
          end = m < 8 || cur >= limit;
        }
        if (end) {
          ev = true;
          st = M2_WALK;
          if (cur > bestlen) {  // lz77.c:495-505: new change point of sublen
            // (a 2-byte "match" only moves bestlength; sublen[2] is never read)
            if (cur >= 3) {
              if (ncp < 8) {
                const u64 v = (u64)((cur - 3u) | (dist << 8));          // 24 bits at bit 24 ncp of cw2 : cw1 : cw0
                const u32 bit = 24u * ncp, wi = bit >> 6, sh = bit & 63u;
                const u64 lo = v << sh, hi = sh > 40 ? v >> (64u - sh) : 0ull;
                cw0 |= wi == 0 ? lo : 0ull;
                cw1 |= wi == 1 ? lo : wi == 0 ? hi : 0ull;
                cw2 |= wi == 2 ? lo : wi == 1 ? hi : 0ull;
              } else if (ncp < SCRATCH_CPS) {
                my_scratch[ncp] = cur | (dist << 16);
              }
              ++ncp;
            }
            bestlen = cur;
            bestdist = dist;
            fin = cur >= limit;
            if (FILT) {
              foff = cur >= 3 ? cur - 3u : 0u;
              fmask = cur >= 3 ? 0xffffffffu : 0xffffffu;    // (cur = 2: bytes 0..2)
              pbyte = m2_lds_u32(win, lp + foff);
            } else if (cur < size_rem) {
              pbyte = lds_byte(win, lp + cur);
            }
          }
        }
      }
      if (PROF) { n_hits += ev ? 1u : 0u; ++n_iter; }
      if (ev) {
        if (!fin) {
          // lz77.c:509-519: switch to the run-length hash; on chain 1 the 3-byte
          // hashes are equal, so val2 equality is equality of ((same-3)&255)
          const u32 lz = L.y & 0xffffu;
          if (chain == 1 && bestlen >= same_pos && ((lz - 3u) & 255u) == ((same_pos - 3u) & 255u)) chain = 2;
          const u32 step = chain == 1 ? (L.x & 0xffffu) : (L.x >> 16);
          lc -= step;
          dist += step;
          --hits_left;
          // lz77.c:521-523 (end of chain), :464 (window), :527-530 (hit cap)
          fin = step == 0 || dist >= ZMX_WINDOW || hits_left == 0;
          if (!fin) L = lk[li - dist];
        }
        if (fin) st = M2_PEND;
      }
    }
    if (PROF) {
      atomicAdd(reinterpret_cast<unsigned long long*>(P.counters + 4), (unsigned long long)n_hits);
      if ((tid & 63) == 0) atomicAdd(reinterpret_cast<unsigned long long*>(P.counters + 6), (unsigned long long)n_iter);
    }
  }
}

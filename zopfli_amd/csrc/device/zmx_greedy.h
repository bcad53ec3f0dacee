// Segmented ZopfliLZ77Greedy (lz77.c:544-630) on the match table.  Included only by
// zmx_hip.hip, after zmx_kernels.h and zmx_trace.h (same segmentation: TS_SEG positions).
//
// The lazy-matching automaton is a chain of dependent decisions, but its whole state between
// two visited positions is (next position i, "a match is held at i - 1") — the held match itself
// is the record of position i - 1.  A visit advances by at most 258, so the walk enters a
// segment within its first 258 positions: 2 x 258 possible entry states.
//
//   k_greedy_exits  one workgroup per segment: for every entry state run the automaton over
//                   the segment's record headers (in LDS): exit state + symbols emitted.
//   k_greedy_link   one wave per block: one table lookup per segment gives every segment its
//                   real entry state and its offset in the symbol store.
//   k_greedy_emit   one wave per segment: the register-window walk of the automaton from the
//                   real entry state; marked lanes write symbols and histogram.
#pragma once

#define GS_STATES (2u * TS_ENT)   // (entry offset j, held flag): index 2 j + held

struct GreedySegParams {
  const BlockDesc* blocks;
  const u32* seg_off;      // [nb + 1] cumulative segment counts
  u32 nb;
  const u32* recs;
  u32* store;
  u32* hist_out;           // [nb][320]
  u32* nsym_out;           // [nb]
  u32* extab;              // [segments][GS_STATES]: exit state | symbols << 16
  uint2* seginfo;          // [segments]: {entry state, symbol offset}
};

// One visit of the automaton.  Returns the symbols emitted (0..2); i and held are updated.
// h = record header of position i, hp = header of position i - 1 (the held match).
__device__ __forceinline__ int gs_score(u32 h) {
  const u32 leng = h & 0xffffu, dist = h >> 16;
  return dist > 1024 ? (int)leng - 1 : (int)leng;      // lz77.c:265-271
}

// Pointer jumping over the automaton's states (see k_trace_exits): J[2 (i - lo) + held] = (state
// after a run of visits, symbols emitted on the way); a state that left the segment is
// 0x8000 | 2 (i - hi) + held.
#define GS_ROUNDS 6
#define GS_THREADS 512u

__global__ __launch_bounds__(GS_THREADS) void k_greedy_exits(GreedySegParams P) {
  __shared__ u32 s_h[TS_SEG + 1];   // s_h[x] = header of position lo - 1 + x
  __shared__ u32 s_j[2 * TS_SEG];
  const u32 seg = blockIdx.x;
  const u32 b = ts_find_block(P.seg_off, P.nb, seg);
  const BlockDesc bd = P.blocks[b];
  const u32 B = (u32)(bd.inend - bd.instart);
  const u32 lo = (seg - P.seg_off[b]) * TS_SEG;
  const u32 hi = lo + TS_SEG < B ? lo + TS_SEG : B;
  const u32* rbase = P.recs + bd.pos_off * 8;
  for (u32 x = threadIdx.x; x <= hi - lo; x += GS_THREADS) {
    const u32 pos = lo + x;   // header of position pos - 1
    s_h[x] = pos >= 1 ? rbase[(u64)(pos - 1) * 8] : 0u;
  }
  __syncthreads();
  const u32 ns = 2 * (hi - lo);
  // one visit of the automaton from every state (lz77.c:581-629)
  for (u32 st = threadIdx.x; st < ns; st += GS_THREADS) {
    u32 i = lo + (st >> 1);
    bool held = st & 1;
    u32 cnt = 0;
    const u32 h = s_h[i - lo + 1];
    const u32 leng = h & 0xffffu;
    const int score = gs_score(h);
    bool done = false;
    if (held) {                                                               // lz77.c:581-607
      held = false;
      const u32 hp = s_h[i - lo];
      if ((hp & 0xffffu) < 3u) {                 // no match can be held at i - 1: the state is never reached;
        s_j[st] = 0x8000u;                       // park it (every live transition moves forward, runs terminate)
        continue;
      }
      ++cnt;                                     // the literal or the match of position i - 1
      if (score > gs_score(hp) + 1) {
        if (score >= 3 && leng < ZMX_MAX_MATCH) { held = true; ++i; done = true; }
      } else {
        i += (hp & 0xffffu) - 1;
        done = true;
      }
    } else if (score >= 3 && leng < ZMX_MAX_MATCH) {                          // lz77.c:608-613
      held = true; ++i;
      done = true;
    }
    if (!done) {                                                              // lz77.c:618-629
      ++cnt;
      i += score >= 3 ? leng : 1u;
    }
    const u32 code = i < hi ? 2 * (i - lo) + (held ? 1u : 0u) : 0x8000u | (2 * (i - hi) + (held ? 1u : 0u));
    s_j[st] = code | (cnt << 16);
  }
  __syncthreads();
  for (int r = 0; r < GS_ROUNDS; ++r) {
    for (u32 st = threadIdx.x; st < ns; st += GS_THREADS) {
      const u32 v = s_j[st];
      if (!(v & 0x8000u)) {
        const u32 w = s_j[v & 0xffffu];
        s_j[st] = (w & 0xffffu) | ((v & 0xffff0000u) + (w & 0xffff0000u));
      }
    }
    __syncthreads();
  }
  for (u32 st = threadIdx.x; st < GS_STATES; st += GS_THREADS) {
    const u32 i0 = lo + (st >> 1);
    if (i0 >= hi || ((st & 1) && i0 == 0)) continue;     // not an entry state of this segment
    u32 v = s_j[st];
    while (!(v & 0x8000u)) {
      const u32 w = s_j[v & 0xffffu];
      v = (w & 0xffffu) | ((v & 0xffff0000u) + (w & 0xffff0000u));
    }
    P.extab[(u64)seg * GS_STATES + st] = (v & 0x7fffu) | (v & 0xffff0000u);   // exit state | symbols << 16
  }
}

__global__ __launch_bounds__(64) void k_greedy_link(GreedySegParams P) {
  const u32 b = blockIdx.x;
  const u32 lane = threadIdx.x;
  for (u32 i = lane; i < 320; i += 64) P.hist_out[(u64)b * 320 + i] = 0;   // k_greedy_emit adds into it
  if (lane != 0) return;
  const BlockDesc bd = P.blocks[b];
  const u32 B = (u32)(bd.inend - bd.instart);
  const u32 s0 = P.seg_off[b], ns = P.seg_off[b + 1] - s0;
  u32 st = 0, off = 0;
  for (u32 s = 0; s < ns; ++s) {
    const u32 lo = s * TS_SEG, hi = lo + TS_SEG < B ? lo + TS_SEG : B;
    if (lo + (st >> 1) >= hi) {                  // the walk jumped over this (short, last) segment
      P.seginfo[s0 + s] = make_uint2(0xffffffffu, off);
      st = 2 * (lo + (st >> 1) - hi) + (st & 1);
      continue;
    }
    P.seginfo[s0 + s] = make_uint2(st, off);
    const u32 v = P.extab[(u64)(s0 + s) * GS_STATES + st];
    off += v >> 16;
    st = v & 0xffffu;
  }
  P.nsym_out[b] = off;
}

__global__ __launch_bounds__(64) void k_greedy_emit(GreedySegParams P) {
  __shared__ u32 s_hist[320];
  const u32 seg = blockIdx.x;
  const u32 b = ts_find_block(P.seg_off, P.nb, seg);
  const BlockDesc bd = P.blocks[b];
  const u32 B = (u32)(bd.inend - bd.instart);
  const u32 lane = threadIdx.x;
  const u32 lo = (seg - P.seg_off[b]) * TS_SEG;
  const u32 E = lo + TS_SEG < B ? lo + TS_SEG : B;   // end of the segment
  const uint2 info = P.seginfo[seg];
  const u32* rbase = P.recs + bd.pos_off * 8;
  u32* sbase = P.store + bd.pos_off;

  for (u32 k = lane; k < 320; k += 64) s_hist[k] = 0;
  __syncthreads();

  if (info.x != 0xffffffffu) {
    u32 total = info.y;
    u32 i = lo + (info.x >> 1);                  // next position to visit
    bool avail = info.x & 1;                     // a match is held at position i - 1 (lz77.c:558-562)
    u32 prev_h = 0, prev_lit = 0;                // its length | dist << 16, and the byte at i - 1
    int prevscore = 0;
    if (avail) {
      const uint2 hp = *reinterpret_cast<const uint2*>(rbase + (u64)(i - 1) * 8);
      prev_h = hp.x;
      prev_lit = (hp.y >> 16) & 255u;
      prevscore = gs_score(prev_h);
    }
    // 64 record headers sit in VGPRs (lane l = position wb + l, the next window is prefetched); a
    // visit is one v_readlane plus scalar ALU and sets a bit in one of two SGPR masks ("this
    // lane's literal" / "this lane's match"); the marked lanes write once per window.
    u32 wb = i;
    uint2 cur = greedy_window(rbase, wb, lane, B);
    uint2 nxt = greedy_window(rbase, wb + 64, lane, B);
    while (i < E) {
      const u32 nwin = E - wb < 64u ? E - wb : 64u;
      const u32 d0_v = cur.x;                    // length | dist << 16 of the position of this lane
      const u32 lit_v = (cur.y >> 16) & 255u;
      u64 m_lit = 0, m_match = 0;
      u32 carry = 0, carry_sym = 0;              // a symbol for position wb - 1 (held across the window edge)
      int idx = (int)(i - wb);
      int last = idx;
      while (idx < (int)nwin) {
        const u32 h = rdlane_u32(d0_v, (u32)idx);
        const u32 leng = h & 0xffffu;
        const int score = gs_score(h);
        last = idx;
        if (avail) {                                                          // lz77.c:581-607
          avail = false;
          if (score > prevscore + 1) {
            if (idx > 0) m_lit |= 1ull << (idx - 1); else { carry = 1; carry_sym = prev_lit; }
            if (score >= 3 && leng < ZMX_MAX_MATCH) {
              avail = true; prev_h = h; prevscore = score;
              idx += 1;
              continue;
            }
          } else {
            if (idx > 0) m_match |= 1ull << (idx - 1); else { carry = 1; carry_sym = prev_h; }
            idx += (int)(prev_h & 0xffffu) - 1;                               // (i - 1) + prev_length
            continue;
          }
        } else if (score >= 3 && leng < ZMX_MAX_MATCH) {                      // lz77.c:608-613
          avail = true; prev_h = h; prevscore = score;
          idx += 1;
          continue;
        }
        if (score >= 3) {                                                     // lz77.c:618-629
          m_match |= 1ull << idx;
          idx += (int)leng;
        } else {
          m_lit |= 1ull << idx;
          idx += 1;
        }
      }
      if (avail) prev_lit = rdlane_u32(lit_v, (u32)last);   // the held match is at the last visited position
      i = wb + (u32)idx;
      // ---- emit this window's symbols in position order
      const u64 m_any = m_lit | m_match;
      if (carry && lane == 0) {
        sbase[total] = carry_sym;
        hist_add_symbol(s_hist, carry_sym & 0xffffu, carry_sym >> 16);
      }
      if ((m_any >> lane) & 1) {
        const u32 below = (u32)__popcll(m_any & ((1ull << lane) - 1));
        const u32 e = ((m_match >> lane) & 1) ? d0_v : lit_v;
        sbase[total + carry + below] = e;
        hist_add_symbol(s_hist, e & 0xffffu, e >> 16);
      }
      total += carry + (u32)__popcll(m_any);
      if (i >= E) break;
      // ---- next window: the prefetched one if the walk ended inside it
      if (i < wb + 128) {
        wb += 64;
        cur = nxt;
      } else {
        wb = i;
        cur = greedy_window(rbase, wb, lane, B);
      }
      nxt = greedy_window(rbase, wb + 64, lane, B);
    }
  }
  __syncthreads();
  for (u32 k = lane; k < 320; k += 64) {
    const u32 v = s_hist[k];
    if (v) atomicAdd(&P.hist_out[(u64)b * 320 + k], v);
  }
}

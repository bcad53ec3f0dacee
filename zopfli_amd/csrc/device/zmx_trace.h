// Segmented TraceBackwards + FollowPath (squeeze.c:317, :338).  Included only by
// zmx_hip.hip, after zmx_kernels.h.
//
// The backward walk over length_array is a chain of dependent reads, but only ACROSS
// segments: a step is at most 258 cells long, so the walk can enter a segment of TS_SEG cells
// only at one of its top 258 cells.  Three kernels:
//
//   k_trace_exits   one workgroup per segment: for each of the <= 258 possible entry cells,
//                   walk the segment (its length_array slice sits in LDS) and record where the
//                   walk leaves it and how many symbols it emits on the way.
//   k_trace_link    one wave per block: follow the exit table from the top segment down —
//                   one table lookup per segment — giving every segment its real entry cell
//                   and its offset in the symbol store; also the block's symbol count.
//   k_trace_emit    one wave per segment: walk the segment from its real entry (k_trace's
//                   SGPR-mask walk over 64-cell register windows) and resolve every symbol
//                   against the match records (dist = sublen[length]), writing
//                   symbols and histogram.
//
// The serial part shrinks from "symbols per block" dependent steps to "segments per block"
// (245 for a 1 MB block); everything else runs on all CUs.
#pragma once

#define TS_SEG 4096u          // cells per segment
#define TS_ENT 258u           // possible entry cells per segment (ZMX_MAX_MATCH)

struct TraceSegParams {
  const BlockDesc* blocks;
  const u32* seg_off;      // [nb_total + 1] cumulative segment counts
  u32 nb_total;
  u32 block0;              // first block of this launch
  u32 seg0;                // first segment of this launch
  const u32* recs;
  const u32* pool;
  const u16* la;
  const int* slot;         // [nb_total]
  u32* store0;
  u32* store1;
  u32* hist_out;           // [nb_total][320]
  u32* nsym_out;           // [nb_total]
  u32* flags;              // [1] error bits
  u32* extab;              // [total segments][TS_ENT]: (lo - exit) | symbols << 16
  uint2* seginfo;          // [total segments]: {entry index (hi - entry), symbol offset}
};

// segment s of a block with B cells: heads in (lo, hi], hi = B - s * TS_SEG
__device__ __forceinline__ void ts_bounds(u32 B, u32 s, u32& lo, u32& hi) {
  hi = B - s * TS_SEG;
  lo = hi > TS_SEG ? hi - TS_SEG : 0u;
}

__device__ __forceinline__ u32 ts_find_block(const u32* seg_off, u32 nb, u32 seg) {
  u32 lo = 0, hi = nb;
  while (hi - lo > 1) {
    const u32 mid = (lo + hi) >> 1;
    if (seg_off[mid] <= seg) lo = mid; else hi = mid;
  }
  return lo;
}

// Pointer jumping: J[h] = (where a run of steps from cell h ends, how many steps it is).  After
// TS_ROUNDS rounds of J[h] = J[h] o J[end of J[h]] over all cells every entry is 2^TS_ROUNDS steps
// per lookup away from its exit, instead of one dependent LDS read per symbol.  Updates are
// in place: a concurrently updated J[t] is a valid run from t either way (32-bit words).
#define TS_ROUNDS 6
#define TS_THREADS 512u

__global__ __launch_bounds__(TS_THREADS) void k_trace_exits(TraceSegParams P) {
  __shared__ u32 s_j[TS_SEG + 1];   // index h - lo: cell code (inside: h' - lo; left: 0x8000 | lo - h') | steps << 16
  const u32 seg = P.seg0 + blockIdx.x;
  const u32 b = ts_find_block(P.seg_off, P.nb_total, seg);
  const BlockDesc bd = P.blocks[b];
  const u32 B = (u32)(bd.inend - bd.instart);
  u32 lo, hi;
  ts_bounds(B, seg - P.seg_off[b], lo, hi);
  const u16* la = P.la + bd.la_off;
  const u32 n = hi - lo;             // cells lo + 1 .. hi
  for (u32 x = threadIdx.x + 1; x <= n; x += TS_THREADS) {
    const u32 h = lo + x;
    u32 len = la[h];
    if (len == 0) len = 1;                   // never-reached cell: keep moving (flagged by k_trace_emit if on the path)
    if (len > h) len = h;
    const u32 t = h - len;
    s_j[x] = (t > lo ? t - lo : 0x8000u | (lo - t)) | (1u << 16);
  }
  __syncthreads();
  for (int r = 0; r < TS_ROUNDS; ++r) {
    for (u32 x = threadIdx.x + 1; x <= n; x += TS_THREADS) {
      const u32 v = s_j[x];
      if (!(v & 0x8000u)) {
        const u32 w = s_j[v & 0xffffu];
        s_j[x] = (w & 0xffffu) | ((v & 0xffff0000u) + (w & 0xffff0000u));
      }
    }
    __syncthreads();
  }
  const u32 j = threadIdx.x;
  if (j < TS_ENT && n > j) {                // entry head e = hi - j > lo
    u32 v = s_j[n - j];
    while (!(v & 0x8000u)) {
      const u32 w = s_j[v & 0xffffu];
      v = (w & 0xffffu) | ((v & 0xffff0000u) + (w & 0xffff0000u));
    }
    P.extab[(u64)seg * TS_ENT + j] = (v & 0x7fffu) | (v & 0xffff0000u);   // (lo - exit) | symbols << 16
  }
}

__global__ __launch_bounds__(64) void k_trace_link(TraceSegParams P) {
  const u32 b = P.block0 + blockIdx.x;
  const BlockDesc bd = P.blocks[b];
  const u32 B = (u32)(bd.inend - bd.instart);
  const u32 lane = threadIdx.x;
  for (u32 i = lane; i < 320; i += 64) P.hist_out[(u64)b * 320 + i] = 0;   // k_trace_emit adds into it
  if (lane != 0) return;
  const u32 s0 = P.seg_off[b], ns = P.seg_off[b + 1] - s0;
  u32 j = 0, off = 0;
  for (u32 s = 0; s < ns; ++s) {
    P.seginfo[s0 + s] = make_uint2(j, off);
    u32 lo, hi;
    ts_bounds(B, s, lo, hi);
    if (hi - lo <= j) {                      // the walk ended above this (short, last) segment
      P.seginfo[s0 + s] = make_uint2(0xffffffffu, off);
      continue;
    }
    const u32 v = P.extab[(u64)(s0 + s) * TS_ENT + j];
    off += v >> 16;
    j = v & 0xffffu;
  }
  P.nsym_out[b] = off;
}

__global__ __launch_bounds__(64) void k_trace_emit(TraceSegParams P) {
  __shared__ __align__(16) u16 s_la[TR_CHUNK];
  __shared__ u32 s_sym[128];       // (start position, length) of walked symbols awaiting resolution
  __shared__ u32 s_len[128];
  __shared__ u32 s_hist[320];

  const u32 seg = P.seg0 + blockIdx.x;
  const u32 b = ts_find_block(P.seg_off, P.nb_total, seg);
  const BlockDesc bd = P.blocks[b];
  const u32 B = (u32)(bd.inend - bd.instart);
  const u32 lane = threadIdx.x;
  u32 seg_lo, seg_hi;
  ts_bounds(B, seg - P.seg_off[b], seg_lo, seg_hi);
  const uint2 info = P.seginfo[seg];
  const u32* rbase = P.recs + bd.pos_off * 8;
  const u16* la = P.la + bd.la_off;
  u32* sbase = (P.slot[b] ? P.store1 : P.store0) + bd.pos_off;

  for (u32 i = lane; i < 320; i += 64) s_hist[i] = 0;
  __syncthreads();

  // 64 cells of length_array sit in one VGPR (lane i = cell wb + i); a step is v_readlane +
  // s_sub and sets the bit of the visited cell in an SGPR mask — the lane of a visited cell x
  // holds everything about the symbol that ENDS there: length la[x], start x - la[x].  After a
  // window the marked lanes are compacted (highest cell first = stream order from the back) into
  // an LDS queue; every 64 queued symbols the lanes resolve FollowPath in parallel, with the
  // record loads of one batch in flight while the next windows are walked.
  u32 head = seg_hi - info.x;       // the real entry of this segment
  if (seg_hi - seg_lo <= info.x) head = seg_lo;   // (segment not on the path)
  u32 total = info.y;               // symbols before this segment, counted from the back
  u32 queued = 0;
  u32 lo = 0, hi = 0;               // cells [lo, hi] are staged in s_la
  bool bad = false;
  u32 pend_n = 0, pend_total = 0, pend_len = 0;
  uint4 pend_ra = make_uint4(0, 0, 0, 0), pend_rb = make_uint4(0, 0, 0, 0);

  for (;;) {
    while (queued < 64 && head > seg_lo && !bad) {
      const u32 wb = head > 63 ? head - 63 : 0;
      if (hi == 0 || wb < lo) {   // restage [lo, head]: four 16-byte loads per lane, issued together
        lo = head > TR_CHUNK - 8 ? (head - (TR_CHUNK - 8)) & ~7u : 0;
        hi = head;
        __syncthreads();
        uint4 v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const u32 i8 = (lane + 64u * r) * 8;
          v[r] = make_uint4(0, 0, 0, 0);
          if (lo + i8 <= hi) v[r] = *reinterpret_cast<const uint4*>(la + lo + i8);   // la rows are padded to 8 entries
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) reinterpret_cast<uint4*>(s_la)[lane + 64 * r] = v[r];
        __syncthreads();
      }
      const u32 cell = wb + lane;
      const u32 la_raw = cell <= head ? (u32)s_la[cell - lo] : 0u;
      const u32 la_v = la_raw ? la_raw : 1u;        // never-reached cells hold 0: keep the walk moving,
      u64 mask = 0;                                 // validity of the visited cells is checked below
      int idx = (int)(head - wb);
      // heads <= seg_lo belong to the next segment (cell 0 starts the block: not a symbol end)
      const int lim = __builtin_amdgcn_readfirstlane((int)(seg_lo + 1 > wb ? seg_lo + 1 - wb : 0u));
      do {
        const u32 len = rdlane_u32(la_v, (u32)idx);
        mask |= 1ull << idx;
        idx -= (int)len;
      } while (idx >= lim);
      if (__ballot(((mask >> lane) & 1) && (la_raw == 0 || la_raw > cell))) { bad = true; break; }   // corrupt length_array
      head = (u32)((int)wb + idx);   // first head below the window / the segment
      const bool on = (mask >> lane) & 1;
      const u32 above = (u32)__popcll(lane < 63 ? mask >> (lane + 1) : 0ull);
      if (on) {
        s_sym[queued + above] = cell - la_v;
        s_len[queued + above] = la_v;
      }
      queued += (u32)__popcll(mask);
      __syncthreads();
    }
    // ---- resolve the pending batch (its records were requested one round ago)
    if (pend_n) {
      if (lane < pend_n) {
        u32 e;
        const u32 d1 = pend_ra.y;
        if (pend_len >= 3) {
          const u32 ncpf = d1 >> 24;
          u32 dist = 0;
          if (ncpf != 0xffu) {
            const u32 w[6] = {pend_ra.z, pend_ra.w, pend_rb.x, pend_rb.y, pend_rb.z, pend_rb.w};
#pragma unroll
            for (int k = 7; k >= 0; --k) {
              const u32 bit = 24u * k;
              const u32 lo32 = w[bit >> 5] >> (bit & 31);
              const u32 v = (bit & 31) > 8 ? (lo32 | (w[(bit >> 5) + 1 > 5 ? 5 : (bit >> 5) + 1] << (32 - (bit & 31)))) : lo32;
              if ((u32)k < ncpf && (v & 255u) + 3u >= pend_len) dist = (v >> 8) & 0xffffu;
            }
          } else {
            const u32 off = pend_ra.z, cnt = pend_ra.w & 0xffffu;
            u32 plo = 0, phi = cnt;   // first entry with len >= pend_len
            while (plo < phi) {
              const u32 mid = (plo + phi) >> 1;
              if ((P.pool[off + mid] & 0xffffu) < pend_len) plo = mid + 1; else phi = mid;
            }
            if (plo < cnt) dist = P.pool[off + plo] >> 16;
          }
          e = pend_len | (dist << 16);
          if (dist == 0) atomicOr(&P.flags[1], 4u);
        } else {
          e = (d1 >> 16) & 255u;
        }
        sbase[B - 1 - (pend_total + lane)] = e;
        hist_add_symbol(s_hist, e & 0xffffu, e >> 16);
      }
      pend_n = 0;
    }
    if (queued == 0) break;
    // ---- request the records of up to 64 queued symbols, keep the rest queued
    {
      const u32 n = queued < 64 ? queued : 64;
      pend_n = n;
      pend_total = total;
      if (lane < n) {
        const u32 pos = s_sym[lane];
        pend_len = s_len[lane];
        const u32* rec = rbase + (u64)pos * 8;
        pend_ra = *reinterpret_cast<const uint4*>(rec);
        pend_rb = *reinterpret_cast<const uint4*>(rec + 4);
      }
      total += n;
      const u32 rest = queued - n;
      __syncthreads();
      u32 mv_s = 0, mv_l = 0;
      if (lane < rest) { mv_s = s_sym[n + lane]; mv_l = s_len[n + lane]; }
      __syncthreads();
      if (lane < rest) { s_sym[lane] = mv_s; s_len[lane] = mv_l; }
      queued = rest;
      __syncthreads();
    }
  }
  if (bad && lane == 0) atomicOr(&P.flags[1], 2u);
  __syncthreads();
  for (u32 i = lane; i < 320; i += 64) {
    const u32 v = s_hist[i];
    if (v) atomicAdd(&P.hist_out[(u64)b * 320 + i], v);
  }
}

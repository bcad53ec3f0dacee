// gfx950 kernels of the zopfli hot path.  Included only by zmx_hip.hip.
//
// Data layout in HBM (one "batch" = the blocks passed to zmx_tables_build):
//   in[]      the resident input bytes (+ zero padding)
//   links[]   per region position p in [ws, inend) of every block, 8 bytes:
//               .x prev1  distance to the previous position with the same 3-byte hash (0 = none)
//               .y prev2  same for the run-length hash (hash.c:129-135)
//               .z same   number of following equal bytes inside the block (hash.c:116-126)
//   recs[]    per block position, 32 bytes = the ZopfliFindLongestMatch result:
//               d0 = length | dist << 16
//               d1 = same | literal << 16 | ncp << 24        (ncp 0..8, 0xff = overflow)
//               24 bytes: 8 change points of sublen, 3 bytes each (len-3, dist lo, dist hi)
//               overflow: bytes 8..11 = offset into pool[], bytes 12..13 = ncp; pool entries
//               are len | dist << 16
//   la[]      length_array of the last squeeze run, u16, blocksize+1 per block
//   store[]   two slots of LZ77 symbols per block, u32 = litlen | dist << 16
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned char u8;
typedef unsigned short u16;
typedef unsigned int u32;
typedef unsigned long long u64;

struct BlockDesc {
  u64 instart, inend, ws;  // ws = max(0, instart - 32768)
  u64 pos_off;             // first record / store entry of this block
  u64 reg_off;             // first links[] entry (position ws)
  u64 la_off;              // first length_array entry
};

#define ZMX_WINDOW 32768u
#define ZMX_MAX_MATCH 258u
#define ZMX_MAX_CHAIN_HITS 8192
#define ZMX_NONE16 0xffffu

// ----------------------------------------------------------------------------
// RFC 1951 symbol geometry (symbols.h of the reference, closed forms)
// ----------------------------------------------------------------------------
__device__ __forceinline__ int dev_dist_symbol(u32 d) {
  if (d < 5) return (int)d - 1;
  const int l = 31 - __clz((int)(d - 1));
  return 2 * l + (int)(((d - 1) >> (l - 1)) & 1);
}
__device__ __forceinline__ int dev_dist_extra_bits(u32 d) {
  return d < 5 ? 0 : (31 - __clz((int)(d - 1))) - 1;
}
// lengths 3..258 -> 257..285
__device__ __forceinline__ int dev_length_symbol(u32 l) {
  if (l < 11) return 254 + (int)l;
  if (l == 258) return 285;
  const int e = 31 - __clz((int)(l - 3)) - 2;          // extra bits 1..5
  return 261 + 4 * e + (int)(((l - 3) >> e) & 3);
}
__device__ __forceinline__ int dev_length_extra_bits(u32 l) {
  if (l < 11 || l == 258) return 0;
  return 31 - __clz((int)(l - 3)) - 2;
}

// ----------------------------------------------------------------------------
// K1a  same[]: run length ahead, bounded by the block end, capped at 65535
// ----------------------------------------------------------------------------
#define SAME_CH 64

__global__ __launch_bounds__(256) void k_same(const u8* __restrict__ in, const BlockDesc* __restrict__ blocks,
                                              u16* __restrict__ same16) {
  const BlockDesc bd = blocks[blockIdx.y];
  const u64 L = bd.inend - bd.ws;
  const u64 c0 = ((u64)blockIdx.x * blockDim.x + threadIdx.x) * SAME_CH;
  if (c0 >= L) return;
  const u64 c1 = (c0 + SAME_CH < L) ? c0 + SAME_CH : L;
  const u8* base = in + bd.ws;
  // same[c1] by scanning forward (at most 65535 bytes, 8 at a time once aligned)
  u32 next = 0;
  if (c1 < L) {
    const u8 c = base[c1];
    u64 q = c1 + 1;
    u32 r = 0;
    bool stop = false;
    while (q < L && r < 65535u && (((u64)(base + q)) & 7)) {
      if (base[q] != c) { stop = true; break; }
      ++q; ++r;
    }
    if (!stop) {
      const u64 pat = 0x0101010101010101ull * c;
      while (q + 8 <= L && r < 65535u) {
        const u64 x = *reinterpret_cast<const u64*>(base + q) ^ pat;
        if (x) { r += (u32)(__ffsll((long long)x) - 1) >> 3; stop = true; break; }
        q += 8; r += 8;
      }
      if (!stop) {
        while (q < L && r < 65535u && base[q] == c) { ++q; ++r; }
      }
    }
    next = r < 65535u ? r : 65535u;
  }
  u16* out = same16 + bd.reg_off;
  for (u64 k = c1; k-- > c0;) {
    u32 s = 0;
    if (k + 1 < L && base[k + 1] == base[k]) s = next < 65535u ? next + 1 : 65535u;
    out[k] = (u16)s;
    next = s;
  }
}

// ----------------------------------------------------------------------------
// K1b  prev links: sequential head-table replay in LDS, one wave per
//      (block, 32768-position chunk, chain).  Positions older than 32767 are
//      unreachable (hash.c:110-114 + window aliasing), so every chunk warms up
//      from 32768 positions before its first emitted position.
// ----------------------------------------------------------------------------
#define CH_EMIT 32768u
#define CH_TILE 1024u
#define CH_LDS_BYTES (65536 + 2 * CH_TILE * 2)

__global__ __launch_bounds__(64) void k_chain(const u8* __restrict__ in, const BlockDesc* __restrict__ blocks,
                                              const u16* __restrict__ same16, ushort4* __restrict__ links) {
  extern __shared__ __align__(16) u8 dyn_lds[];
  u16* head = reinterpret_cast<u16*>(dyn_lds);
  u16* keys = head + 32768;
  u16* outs = keys + CH_TILE;

  const BlockDesc bd = blocks[blockIdx.y];
  const u32 chain = blockIdx.z;
  const u64 L = bd.inend - bd.ws;
  const u64 e0 = (u64)blockIdx.x * CH_EMIT;
  if (e0 >= L) return;
  const u64 e1 = (e0 + CH_EMIT < L) ? e0 + CH_EMIT : L;
  const u64 w0 = e0 >= ZMX_WINDOW ? e0 - ZMX_WINDOW : 0;
  const u32 lane = threadIdx.x;

  for (u32 i = lane; i < 16384; i += 64) reinterpret_cast<u32*>(head)[i] = 0xffffffffu;
  __syncthreads();

  const u8* base = in + bd.ws;
  const u16* same = same16 + bd.reg_off;
  ushort4* lk = links + bd.reg_off;

  for (u64 t0 = w0; t0 < e1; t0 += CH_TILE) {
    const u32 tn = (u32)((e1 - t0 < CH_TILE) ? e1 - t0 : CH_TILE);
    for (u32 i = lane; i < tn; i += 64) {
      const u64 k = t0 + i;
      const u32 b0 = base[k];
      const u32 b1 = k + 1 < L ? base[k + 1] : 0;
      const u32 b2 = k + 2 < L ? base[k + 2] : 0;
      u32 v = ((b0 << 10) ^ (b1 << 5) ^ b2) & 32767u;  // hash.c:96-98, three rolling updates
      if (chain) v ^= ((u32)same[k] - 3u) & 255u;      // hash.c:129
      keys[i] = (u16)v;
    }
    __syncthreads();
    if (lane == 0) {
      const u32 cur0 = (u32)(t0 - w0);
      u32 i = 0;
      for (; i + 8 <= tn; i += 8) {
        u32 kk[8], old[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) kk[u] = keys[i + u];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          old[u] = head[kk[u]];
          head[kk[u]] = (u16)(cur0 + i + u);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          u32 d = old[u] == ZMX_NONE16 ? 0u : (cur0 + i + u) - old[u];
          if (d > 32767u) d = 0;
          outs[i + u] = (u16)d;
        }
      }
      for (; i < tn; ++i) {
        const u32 key = keys[i];
        const u32 old = head[key];
        head[key] = (u16)(cur0 + i);
        u32 d = old == ZMX_NONE16 ? 0u : (cur0 + i) - old;
        if (d > 32767u) d = 0;
        outs[i] = (u16)d;
      }
    }
    __syncthreads();
    for (u32 i = lane; i < tn; i += 64) {
      const u64 k = t0 + i;
      if (k >= e0) {
        if (chain == 0) {
          lk[k].x = outs[i];
          lk[k].z = same[k];
        } else {
          lk[k].y = outs[i];
        }
      }
    }
    __syncthreads();
  }
}

// ----------------------------------------------------------------------------
// K2  match table: ZopfliFindLongestMatch(limit 258, sublen) for every position
//     (lz77.c:407-542).  Persistent workgroups pull 2048-position tiles; the
//     32 KiB window + tile + 258 bytes are staged in LDS; each lane walks the
//     hash chain of one position at a time and refills from the tile queue as
//     soon as its walk ends, so long walks do not idle the wave.
// ----------------------------------------------------------------------------
#define MT 2048u
#define MWIN_BYTES (32768u + MT + 288u)     // + slack for 16-byte alignment and 4-byte compares
#define MATCH_THREADS 256
#define SCRATCH_CPS 256u                     // per-lane overflow change points

struct MatchParams {
  const u8* in;
  const BlockDesc* blocks;
  const u32* tile_off;   // [nb + 1] cumulative tile counts
  u32 nb;
  u32 total_tiles;
  const ushort4* links;
  u32* recs;
  u32* pool;
  u32 pool_cap;
  u32* counters;         // [0] pool cursor, [1] error flags, [8..15] per-XCD tile cursors
  u32* scratch;          // gridDim.x * MATCH_THREADS * SCRATCH_CPS
};

__device__ __forceinline__ u32 lds_byte(const u32* w, u32 a) { return (w[a >> 2] >> ((a & 3) * 8)) & 255u; }
__device__ __forceinline__ u32 lds_u32_unaligned(const u32* w, u32 a) {
  const u32 lo = w[a >> 2], hi = w[(a >> 2) + 1];
  return __builtin_amdgcn_alignbyte(hi, lo, a & 3);
}

__global__ __launch_bounds__(MATCH_THREADS) void k_match(MatchParams P) {
  __shared__ __align__(16) u32 win[MWIN_BYTES / 4 + 4];
  __shared__ u32 s_next, s_tile;

  const u32 tid = threadIdx.x;
  const u32 xcd = blockIdx.x & 7;
  const u32 t_begin = (u32)(((u64)P.total_tiles * xcd) / 8);
  const u32 t_end = (u32)(((u64)P.total_tiles * (xcd + 1)) / 8);
  u32* my_scratch = P.scratch + ((u64)blockIdx.x * MATCH_THREADS + tid) * SCRATCH_CPS;

  for (;;) {
    __syncthreads();  // previous tile fully consumed before the window is overwritten
    if (tid == 0) {
      s_tile = t_begin + atomicAdd(&P.counters[8 + xcd], 1u);
      s_next = 0;
    }
    __syncthreads();
    const u32 tile = s_tile;
    if (tile >= t_end) break;

    // block of this tile: largest b with tile_off[b] <= tile
    u32 lo = 0, hi = P.nb;
    while (hi - lo > 1) {
      const u32 mid = (lo + hi) >> 1;
      if (P.tile_off[mid] <= tile) lo = mid; else hi = mid;
    }
    const BlockDesc bd = P.blocks[lo];
    const u64 p0 = bd.instart + (u64)(tile - P.tile_off[lo]) * MT;
    const u64 p1 = (p0 + MT < bd.inend) ? p0 + MT : bd.inend;
    const u32 ntile = (u32)(p1 - p0);

    // stage bytes [p0 - 32768, p1 + 258) (clipped to [0, inend)) at LDS offset (abs - wb)
    const long long wb = ((long long)p0 - (long long)ZMX_WINDOW) & ~15ll;  // 16-byte aligned base, may be < 0
    const u64 hi_abs = (p1 + ZMX_MAX_MATCH < bd.inend) ? p1 + ZMX_MAX_MATCH : bd.inend;
    const u32 nvec = (u32)(((long long)hi_abs - wb + 15) >> 4);
    for (u32 v = tid; v < nvec; v += MATCH_THREADS) {
      const long long a = wb + (long long)v * 16;
      uint4 x = make_uint4(0, 0, 0, 0);
      if (a >= 0) x = *reinterpret_cast<const uint4*>(P.in + a);  // input is padded past its end
      reinterpret_cast<uint4*>(win)[v] = x;
    }
    __syncthreads();

    const ushort4* lk = P.links + bd.reg_off;  // index: abs - ws
    const u64 ws = bd.ws;

    // ---- per-lane walk state
    bool active = false, done = false, comparing = false;
    u32 lp = 0, lc = 0;            // LDS byte offsets of pos and candidate
    u32 limit = 0, bestlen = 0, bestdist = 0, dist = 0, ncp = 0, same_pos = 0, cur = 0, size_rem = 0;
    int hits_left = 0, chain = 1;
    u64 pos = 0;
    ushort4 L = make_ushort4(0, 0, 0, 0);  // links of the candidate
    u32* rec = nullptr;

    for (;;) {
      if (!active && !done) {
        const u32 idx = atomicAdd(&s_next, 1u);
        if (idx >= ntile) {
          done = true;
        } else {
          pos = p0 + idx;
          lp = (u32)((long long)pos - wb);
          size_rem = (u32)((bd.inend - pos < 65536) ? bd.inend - pos : 65536);
          const ushort4 Lp = lk[pos - ws];
          same_pos = Lp.z;
          rec = P.recs + (bd.pos_off + (pos - bd.instart)) * 8;
          ncp = 0;
          bestlen = 1; bestdist = 0; chain = 1; hits_left = ZMX_MAX_CHAIN_HITS; comparing = false;
          if (size_rem < 3) {                      // lz77.c:440-446
            rec[0] = 0;
            rec[1] = same_pos | (lds_byte(win, lp) << 16);
          } else {
            limit = size_rem < ZMX_MAX_MATCH ? size_rem : ZMX_MAX_MATCH;  // lz77.c:448-450
            if (Lp.x == 0) {                       // empty chain
              rec[0] = 1;
              rec[1] = same_pos | (lds_byte(win, lp) << 16);
            } else {
              dist = Lp.x;
              lc = lp - dist;
              L = lk[pos - dist - ws];
              active = true;
            }
          }
        }
      }
      if (!__any(active)) {
        if (__all(done)) break;
        continue;
      }
      if (active) {
        bool finish = false;
        if (!comparing) {
          // lz77.c:478-479: test the byte after the current best first
          cur = 0;
          if (bestlen >= size_rem || lds_byte(win, lp + bestlen) == lds_byte(win, lc + bestlen)) {
            comparing = true;
            // lz77.c:481-490: skip the common run (pure acceleration)
            if (same_pos > 2 && lds_byte(win, lp) == lds_byte(win, lc)) {
              u32 s = same_pos < L.z ? same_pos : L.z;
              cur = s < limit ? s : limit;
            }
          }
        }
        if (comparing) {  // GetMatch (lz77.c:297), 4 bytes per step
          const u32 rem = limit - cur;
          if (rem == 0) {
            comparing = false;
          } else {
            const u32 x = lds_u32_unaligned(win, lp + cur) ^ lds_u32_unaligned(win, lc + cur);
            u32 m = x ? (u32)(__ffs((int)x) - 1) >> 3 : 4u;
            if (m > rem) m = rem;
            cur += m;
            if (m < 4 || cur >= limit) comparing = false;
          }
        }
        if (!comparing) {
          if (cur > bestlen) {  // lz77.c:495-505: new change point of sublen
            // (a 2-byte "match" only moves bestlength; sublen[2] is never read)
            if (cur < 3) {
            } else if (ncp < 8) {
              u8* b = reinterpret_cast<u8*>(rec) + 8 + 3 * ncp;
              b[0] = (u8)(cur - 3);
              b[1] = (u8)(dist & 255);
              b[2] = (u8)(dist >> 8);
            } else if (ncp < SCRATCH_CPS) {
              my_scratch[ncp] = cur | (dist << 16);
            }
            if (cur >= 3) ++ncp;
            bestlen = cur;
            bestdist = dist;
            if (cur >= limit) finish = true;
          }
          if (!finish) {
            // lz77.c:509-519: switch to the run-length hash; on chain 1 the 3-byte
            // hashes are equal, so val2 equality is equality of ((same-3)&255)
            if (chain == 1 && bestlen >= same_pos && (((u32)L.z - 3u) & 255u) == ((same_pos - 3u) & 255u)) chain = 2;
            const u32 step = chain == 1 ? L.x : L.y;
            if (step == 0) {
              finish = true;                            // lz77.c:521-523
            } else {
              lc -= step;
              dist += step;
              --hits_left;
              if (dist >= ZMX_WINDOW || hits_left <= 0) finish = true;  // lz77.c:464, 527-530
              else L = lk[pos - dist - ws];
            }
          }
          if (finish) {
            rec[0] = bestlen | (bestdist << 16);
            if (ncp <= 8) {
              rec[1] = same_pos | (lds_byte(win, lp) << 16) | (ncp << 24);
            } else {
              rec[1] = same_pos | (lds_byte(win, lp) << 16) | (0xffu << 24);
              const u32 off = atomicAdd(&P.counters[0], ncp);
              if (off + ncp <= P.pool_cap) {
                const u8* b = reinterpret_cast<const u8*>(rec) + 8;
                u32 first[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) first[e] = ((u32)b[3 * e] + 3u) | (((u32)b[3 * e + 1] | ((u32)b[3 * e + 2] << 8)) << 16);
#pragma unroll
                for (int e = 0; e < 8; ++e) P.pool[off + e] = first[e];
                for (u32 e = 8; e < ncp; ++e) P.pool[off + e] = my_scratch[e];
                rec[2] = off;
                rec[3] = ncp;
              } else {
                atomicOr(&P.counters[1], 1u);  // host retries with a larger pool
                rec[2] = 0;
                rec[3] = 0;
              }
            }
            active = false;
          }
        }
      }
    }
  }
}

// ----------------------------------------------------------------------------
// record access helpers shared by greedy / squeeze
// ----------------------------------------------------------------------------
// distance for `len` at a record: first change point with cp.len >= len  (= sublen[len])
__device__ __forceinline__ u32 rec_dist_for(const u32* __restrict__ rec, const u32* __restrict__ pool, u32 len) {
  const u32 d1 = rec[1];
  const u32 ncpf = d1 >> 24;
  if (ncpf != 0xffu) {
    const u8* b = reinterpret_cast<const u8*>(rec) + 8;
    for (u32 e = 0; e < ncpf; ++e) {
      if ((u32)b[3 * e] + 3u >= len) return (u32)b[3 * e + 1] | ((u32)b[3 * e + 2] << 8);
    }
    return 0;
  }
  const u32 off = rec[2], n = rec[3] & 0xffffu;
  for (u32 e = 0; e < n; ++e) {
    const u32 x = pool[off + e];
    if ((x & 0xffffu) >= len) return x >> 16;
  }
  return 0;
}

__device__ __forceinline__ void hist_add_symbol(u32* hist, u32 litlen, u32 dist) {
  if (dist == 0) {
    atomicAdd(&hist[litlen], 1u);
  } else {
    atomicAdd(&hist[dev_length_symbol(litlen)], 1u);
    atomicAdd(&hist[288 + dev_dist_symbol(dist)], 1u);
  }
}

// ----------------------------------------------------------------------------
// K3  ZopfliLZ77Greedy (lz77.c:544-630) on the match table: one wave per block.
//     The wave stages 1024 record headers at a time in LDS, lane 0 runs the
//     lazy-matching state machine over them, all lanes flush the symbols.
// ----------------------------------------------------------------------------
#define GR_CHUNK 1024u

__global__ __launch_bounds__(64) void k_greedy(const BlockDesc* __restrict__ blocks, const u32* __restrict__ recs,
                                               u32* __restrict__ store, u32* __restrict__ hist_out,
                                               u32* __restrict__ nsym_out) {
  __shared__ u32 s_d0[GR_CHUNK];
  __shared__ u8 s_lit[GR_CHUNK];
  __shared__ u32 s_out[2 * GR_CHUNK + 2];
  __shared__ u32 s_hist[320];
  __shared__ u32 s_i, s_nout, s_prev_len, s_prev_match, s_prev_lit, s_avail;

  const u32 b = blockIdx.x;
  const BlockDesc bd = blocks[b];
  const u32 B = (u32)(bd.inend - bd.instart);
  const u32 lane = threadIdx.x;
  const u32* rbase = recs + bd.pos_off * 8;
  u32* sbase = store + bd.pos_off;

  for (u32 i = lane; i < 320; i += 64) s_hist[i] = 0;
  if (lane == 0) { s_i = 0; s_prev_len = 0; s_prev_match = 0; s_prev_lit = 0; s_avail = 0; }
  __syncthreads();

  u32 total = 0;
  for (;;) {
    const u32 c0 = s_i;
    if (c0 >= B) break;
    const u32 cn = (B - c0 < GR_CHUNK) ? B - c0 : GR_CHUNK;
    for (u32 t = lane; t < cn; t += 64) {
      const uint2 h = *reinterpret_cast<const uint2*>(rbase + (u64)(c0 + t) * 8);
      s_d0[t] = h.x;
      s_lit[t] = (u8)(h.y >> 16);
    }
    __syncthreads();
    if (lane == 0) {
      u32 i = c0, nout = 0;
      u32 prev_len = s_prev_len, prev_match = s_prev_match, prev_lit = s_prev_lit, avail = s_avail;
      while (i < c0 + cn) {
        const u32 h = s_d0[i - c0];
        u32 leng = h & 0xffffu, dist = h >> 16;
        const u32 lit = s_lit[i - c0];
        int lengthscore = dist > 1024 ? (int)leng - 1 : (int)leng;                  // lz77.c:265-271
        const int prevscore = prev_match > 1024 ? (int)prev_len - 1 : (int)prev_len;
        bool emit = true;
        if (avail) {                                                                  // lz77.c:581-607
          avail = 0;
          if (lengthscore > prevscore + 1) {
            s_out[nout++] = prev_lit;                                                 // literal in[i-1]
            if (lengthscore >= 3 && leng < ZMX_MAX_MATCH) {
              avail = 1; prev_len = leng; prev_match = dist; prev_lit = lit;
              emit = false;
            }
          } else {
            s_out[nout++] = prev_len | (prev_match << 16);                           // the held match, at i-1
            i += prev_len - 1;                                                        // (i-1) + prev_len
            continue;
          }
        } else if (lengthscore >= 3 && leng < ZMX_MAX_MATCH) {                       // lz77.c:608-613
          avail = 1; prev_len = leng; prev_match = dist; prev_lit = lit;
          emit = false;
        }
        if (emit) {                                                                   // lz77.c:618-629
          if (lengthscore >= 3) {
            s_out[nout++] = leng | (dist << 16);
          } else {
            leng = 1;
            s_out[nout++] = lit;
          }
          i += leng;
        } else {
          i += 1;
        }
      }
      s_i = i; s_nout = nout;
      s_prev_len = prev_len; s_prev_match = prev_match; s_prev_lit = prev_lit; s_avail = avail;
    }
    __syncthreads();
    const u32 nout = s_nout;
    for (u32 t = lane; t < nout; t += 64) {
      const u32 e = s_out[t];
      sbase[total + t] = e;
      hist_add_symbol(s_hist, e & 0xffffu, e >> 16);
    }
    total += nout;
    __syncthreads();
  }
  __syncthreads();
  for (u32 i = lane; i < 320; i += 64) hist_out[(u64)b * 320 + i] = s_hist[i];
  if (lane == 0) nsym_out[b] = total;
}

// ----------------------------------------------------------------------------
// K4  one LZ77OptimalRun per block (squeeze.c:429): GetBestLengths forward DP
//     (:217) + TraceBackwards (:317) + FollowPath (:338) + histogram.
//     One wave per block; positions are sequential, lanes cover the match
//     lengths k of the current position.  costs[] lives in a 1024-entry LDS
//     ring (only cells j..j+516 are ever live), the symbol cost tables in LDS.
// ----------------------------------------------------------------------------
#define RING 1024u
#define RMASK 1023u
#define TR_CHUNK 2048u

struct SqueezeParams {
  const BlockDesc* blocks;
  const u32* recs;
  const u32* pool;
  const double* cost;      // [nb][320]
  const double* mincost;   // [nb]
  const int* slot;         // [nb]
  u16* la;
  u32* store0;
  u32* store1;
  u32* hist_out;
  u32* nsym_out;
  u32* flags;              // [1] error bits
};

__global__ __launch_bounds__(64) void k_squeeze(SqueezeParams P) {
  __shared__ double s_ll[288];
  __shared__ double s_d[32];
  __shared__ float s_cost[RING];
  __shared__ u16 s_len[RING];
  __shared__ u32 s_hist[320];
  __shared__ u16 s_la[TR_CHUNK + 2];
  __shared__ u32 s_bpos[64];
  __shared__ u32 s_blen[64];
  __shared__ u32 s_n, s_idx;

  const u32 b = blockIdx.x;
  const BlockDesc bd = P.blocks[b];
  const u32 B = (u32)(bd.inend - bd.instart);
  const u32 lane = threadIdx.x;
  const u32* rbase = P.recs + bd.pos_off * 8;
  u16* la = P.la + bd.la_off;
  u32* sbase = (P.slot[b] ? P.store1 : P.store0) + bd.pos_off;

  for (u32 i = lane; i < 288; i += 64) s_ll[i] = P.cost[(u64)b * 320 + i];
  if (lane < 32) s_d[lane] = P.cost[(u64)b * 320 + 288 + lane];
  for (u32 i = lane; i < RING; i += 64) { s_cost[i] = 1e30f; s_len[i] = 0; }
  for (u32 i = lane; i < 320; i += 64) s_hist[i] = 0;
  __syncthreads();
  if (B == 0) {
    for (u32 i = lane; i < 320; i += 64) P.hist_out[(u64)b * 320 + i] = 0;
    if (lane == 0) P.nsym_out[b] = 0;
    return;
  }
  if (lane == 0) { s_cost[0] = 0.0f; la[0] = 0; }
  __syncthreads();

  const double mincost = P.mincost[b];
  // per-lane constants for k = 64 r + lane + 1 (k = 1 is the literal edge)
  int k_lbits[5];
  double k_ll[5];
#pragma unroll
  for (int r = 0; r < 5; ++r) {
    const u32 k = 64u * r + lane + 1;
    if (k >= 3 && k <= ZMX_MAX_MATCH) {
      k_lbits[r] = dev_length_extra_bits(k);
      k_ll[r] = s_ll[dev_length_symbol(k)];
    } else {
      k_lbits[r] = 0;
      k_ll[r] = 0.0;
    }
  }
  // squeeze.c:260: cost of (length 258, dist 1) = (0 + 0) + ll[285] + d[0]
  const double symbolcost258 = (double)(0 + 0) + s_ll[285] + s_d[0];

  u32 j = 0;
  bool allow_shortcut = true;  // the reference tests the shortcut once per loop iteration (squeeze.c:247-251)
  while (j < B) {
    // record of position j (uniform address -> broadcast load)
    const uint4 ra = *reinterpret_cast<const uint4*>(rbase + (u64)j * 8);
    u32 leng = __builtin_amdgcn_readfirstlane(ra.x) & 0xffffu;
    const u32 d1 = __builtin_amdgcn_readfirstlane(ra.y);
    const u32 same_i = d1 & 0xffffu;

    // ---- long-run shortcut (squeeze.c:251-271): i > instart + 259, i + 517 < inend
    if (allow_shortcut && same_i > 2 * ZMX_MAX_MATCH && j > ZMX_MAX_MATCH + 1 && j + 2 * ZMX_MAX_MATCH + 1 < B) {
      const u32 same_back = __builtin_amdgcn_readfirstlane(rbase[(u64)(j - ZMX_MAX_MATCH) * 8 + 1]) & 0xffffu;
      if (same_back > ZMX_MAX_MATCH) {
        // costs[j+t+258] = costs[j+t] + symbolcost for t = 0..257 (reads and writes are disjoint),
        // cells j..j+257 are consumed: their lengths are final.
        __syncthreads();
        for (u32 t = lane; t < ZMX_MAX_MATCH; t += 64) {
          const float c = s_cost[(j + t) & RMASK];
          if (j + t >= 1) la[j + t] = s_len[(j + t) & RMASK];
          s_cost[(j + t + ZMX_MAX_MATCH) & RMASK] = (float)((double)c + symbolcost258);
          s_len[(j + t + ZMX_MAX_MATCH) & RMASK] = (u16)ZMX_MAX_MATCH;
        }
        __syncthreads();
        for (u32 t = lane; t < ZMX_MAX_MATCH; t += 64) s_cost[(j + t) & RMASK] = 1e30f;  // slots recycle
        __syncthreads();
        j += ZMX_MAX_MATCH;
        allow_shortcut = false;  // squeeze.c:273 continues with the match query at the new i
        continue;
      }
    }

    // ---- cell j becomes final
    const float cjf = s_cost[j & RMASK];
    if (lane == 0 && j >= 1) la[j] = s_len[j & RMASK];
    __syncthreads();
    if (lane == 0) s_cost[j & RMASK] = 1e30f;
    const double cj = (double)cjf;
    const double mincostaddcostj = mincost + cj;     // squeeze.c:287
    const u32 lit = (d1 >> 16) & 255u;
    const u32 ncpf = d1 >> 24;
    const uint4 rb = *reinterpret_cast<const uint4*>(rbase + (u64)j * 8 + 4);

    // change points as uniform values
    u32 w[6] = {ra.z, ra.w, rb.x, rb.y, rb.z, rb.w};
    if (leng > B - j) leng = B - j;                  // squeeze.c:286 kend
    const u32 rounds = leng < 3 ? 1u : (leng + 63u) / 64u;
    for (u32 r = 0; r < rounds; ++r) {
      const u32 k = 64u * r + lane + 1;
      const bool is_lit = (k == 1);
      const bool is_match = (k >= 3 && k <= leng);
      if (is_lit || is_match) {
        const u32 cell = (j + k) & RMASK;
        const float oldf = s_cost[cell];
        const double old = (double)oldf;
        double newCost;
        bool consider = true;
        if (is_lit) {
          newCost = s_ll[lit] + cj;                  // squeeze.c:278
        } else {
          if (old <= mincostaddcostj) consider = false;  // squeeze.c:293
          // sublen[k]
          u32 mydist = 0;
          if (ncpf != 0xffu) {
#pragma unroll
            for (int e = 7; e >= 0; --e) {
              if ((u32)e < ncpf) {
                const u32 bit = 24u * e;
                const u32 lo = w[bit >> 5] >> (bit & 31);
                const u32 v = (bit & 31) > 8 ? (lo | (w[(bit >> 5) + ((bit >> 5) < 5 ? 1 : 0)] << (32 - (bit & 31)))) : lo;
                const u32 clen = (v & 255u) + 3u;
                const u32 cdist = (v >> 8) & 0xffffu;
                if (k <= clen) mydist = cdist;
              }
            }
          } else {
            const u32 off = w[0], n = w[1] & 0xffffu;
            for (u32 e = n; e-- > 0;) {
              const u32 x = P.pool[off + e];
              if (k <= (x & 0xffffu)) mydist = x >> 16;
            }
          }
          int lb, rr = (int)r;
          double kl;
          // select per-round constants without dynamic register indexing
          lb = rr == 0 ? k_lbits[0] : rr == 1 ? k_lbits[1] : rr == 2 ? k_lbits[2] : rr == 3 ? k_lbits[3] : k_lbits[4];
          kl = rr == 0 ? k_ll[0] : rr == 1 ? k_ll[1] : rr == 2 ? k_ll[2] : rr == 3 ? k_ll[3] : k_ll[4];
          // squeeze.c:155: (lbits + dbits) as int, then + ll, then + d
          const double c = ((double)(lb + dev_dist_extra_bits(mydist)) + kl) + s_d[dev_dist_symbol(mydist)];
          newCost = c + cj;                          // squeeze.c:297
        }
        if (consider && newCost < old) {
          s_cost[cell] = (float)newCost;
          s_len[cell] = (u16)k;
        }
      }
    }
    __syncthreads();
    ++j;
    allow_shortcut = true;
  }
  // cell B
  if (lane == 0) la[B] = s_len[B & RMASK];
  __threadfence();
  __syncthreads();

  // ---- TraceBackwards + FollowPath: walk length_array from the end in LDS
  //      chunks; every 64 steps the lanes resolve distances in parallel and
  //      write the symbols back to front.
  u32 idx = B, total = 0;
  u32 lo = 0;
  bool have_chunk = false;
  while (idx > 0) {
    if (!have_chunk || (lo > 0 && idx < lo + ZMX_MAX_MATCH)) {
      lo = idx > TR_CHUNK ? idx - TR_CHUNK : 0;
      __syncthreads();
      for (u32 t = lane; t <= idx - lo; t += 64) s_la[t] = la[lo + t];
      have_chunk = true;
      __syncthreads();
    }
    if (lane == 0) {
      u32 n = 0, cur = idx;
      while (n < 64 && cur > 0 && cur >= lo) {
        const u32 len = s_la[cur - lo];
        if (len == 0 || len > cur) { atomicOr(&P.flags[1], 2u); cur = 0; break; }
        s_bpos[n] = cur - len;
        s_blen[n] = len;
        cur -= len;
        ++n;
      }
      s_n = n;
      s_idx = cur;
    }
    __syncthreads();
    const u32 n = s_n;
    if (lane < n) {
      const u32 pos = s_bpos[lane], len = s_blen[lane];
      const u32* rec = rbase + (u64)pos * 8;
      u32 e;
      if (len >= 3) {
        const u32 dist = rec_dist_for(rec, P.pool, len);
        e = len | (dist << 16);
        if (dist == 0) atomicOr(&P.flags[1], 4u);
      } else {
        e = (rec[1] >> 16) & 255u;
      }
      sbase[B - 1 - (total + lane)] = e;
      hist_add_symbol(s_hist, e & 0xffffu, e >> 16);
    }
    total += n;
    idx = s_idx;
    __syncthreads();
    if (n == 0) break;  // error path only
  }
  __syncthreads();
  for (u32 i = lane; i < 320; i += 64) P.hist_out[(u64)b * 320 + i] = s_hist[i];
  if (lane == 0) P.nsym_out[b] = total;
}

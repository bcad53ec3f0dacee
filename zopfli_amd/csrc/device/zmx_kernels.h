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
// wave-level helpers
// ----------------------------------------------------------------------------
// wave64 inclusive scans on the DPP network (row_shr 1/2/4/8, row_bcast 15/31)
#define ZMX_WAVE_SCAN(NAME, OP, IDENT)                                                   \
  __device__ __forceinline__ u32 NAME(u32 v) {                                           \
    u32 t;                                                                               \
    t = (u32)__builtin_amdgcn_update_dpp((int)(IDENT), (int)v, 0x111, 0xf, 0xf, false); v = OP(v, t); \
    t = (u32)__builtin_amdgcn_update_dpp((int)(IDENT), (int)v, 0x112, 0xf, 0xf, false); v = OP(v, t); \
    t = (u32)__builtin_amdgcn_update_dpp((int)(IDENT), (int)v, 0x114, 0xf, 0xf, false); v = OP(v, t); \
    t = (u32)__builtin_amdgcn_update_dpp((int)(IDENT), (int)v, 0x118, 0xf, 0xf, false); v = OP(v, t); \
    t = (u32)__builtin_amdgcn_update_dpp((int)(IDENT), (int)v, 0x142, 0xa, 0xf, false); v = OP(v, t); \
    t = (u32)__builtin_amdgcn_update_dpp((int)(IDENT), (int)v, 0x143, 0xc, 0xf, false); v = OP(v, t); \
    return v;                                                                            \
  }
__device__ __forceinline__ u32 zmx_addu(u32 a, u32 b) { return a + b; }
__device__ __forceinline__ u32 zmx_maxu(u32 a, u32 b) { return a > b ? a : b; }
ZMX_WAVE_SCAN(wave_scan_add, zmx_addu, 0u)
ZMX_WAVE_SCAN(wave_scan_max, zmx_maxu, 0u)

__device__ __forceinline__ u32 rdlane_u32(u32 v, u32 l) { return (u32)__builtin_amdgcn_readlane((int)v, (int)l); }
__device__ __forceinline__ float rdlane_f32(float v, u32 l) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), (int)l));
}
// cross-lane hand-off through LDS inside ONE wave: the LDS queue is in order per
// wave, only the compiler has to be kept from reordering
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
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
// K1b  prev links: the reference's head-table replay (hash.c:110-114, 131-135), one wave per
//      (block, 32768-position chunk, chain), 64 positions per step.  All lanes read
//      head[key] (the state before the step); positions of the same step that share a
//      key are found with 15 ballots — one per key bit, lane mask = AND of "lanes whose
//      bit equals mine" — so a lane's previous occurrence is the nearest lower lane of
//      its key group, else the old head; the last lane of each group writes the head.
//      Positions older than 32767 are unreachable (hash.c:110-114 + window aliasing),
//      so every chunk warms up from 32768 positions before its first emitted position;
//      warm-up steps only settle the heads (write, read back, the overwritten retry).
// ----------------------------------------------------------------------------
#define CH_EMIT 32768u
#define CH_TILE 1024u
#define CH_LDS_BYTES (65536 + CH_TILE * 2)

__global__ __launch_bounds__(64) void k_chain(const u8* __restrict__ in, const BlockDesc* __restrict__ blocks,
                                              const u16* __restrict__ same16, ushort4* __restrict__ links) {
  extern __shared__ __align__(16) u8 dyn_lds[];
  u16* head = reinterpret_cast<u16*>(dyn_lds);   // last position (relative to w0) of every key, 0xffff = none
  u16* keys = head + 32768;

  const BlockDesc bd = blocks[blockIdx.y];
  const u32 chain = blockIdx.z;
  const u64 L = bd.inend - bd.ws;
  const u64 e0 = (u64)blockIdx.x * CH_EMIT;
  if (e0 >= L) return;
  const u64 e1 = (e0 + CH_EMIT < L) ? e0 + CH_EMIT : L;
  const u64 w0 = e0 >= ZMX_WINDOW ? e0 - ZMX_WINDOW : 0;
  const u32 lane = threadIdx.x;
  const u64 lt_mask = (1ull << lane) - 1;          // lanes below me
  const u64 gt_mask = ~((2ull << lane) - 1);       // lanes above me (lane 63: 2 << 63 = 0, mask = 0)

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
    const u32 cur0 = (u32)(t0 - w0);
    for (u32 i0 = 0; i0 < tn; i0 += 64) {
      const u32 i = i0 + lane;
      const bool act = i < tn;
      const u32 key = act ? (u32)keys[i] : 0u;
      if (t0 + i0 + 64 <= e0) {
        // warm-up step (nothing of it is emitted): only the heads matter — the highest position of
        // every key.  All lanes write, one per key gets through; whoever is above the survivor
        // writes again (one round unless two positions of the step share a key).
        const u32 r = cur0 + i;
        bool todo = act;
        while (__any(todo)) {
          if (todo) head[key] = (u16)r;
          wave_lds_sync();
          const u32 w = act ? (u32)head[key] : 0xffffu;
          wave_lds_sync();
          todo = act && w < r;
        }
        continue;
      }
      const u32 old = act ? (u32)head[key] : (u32)ZMX_NONE16;
      u64 grp = __ballot(act);                       // lanes of this step with my key
#pragma unroll
      for (int bit = 0; bit < 15; ++bit) {
        const bool mine = (key >> bit) & 1;
        const u64 bm = __ballot(mine);
        grp &= mine ? bm : ~bm;
      }
      const u64 lower = grp & lt_mask;
      const u32 r = cur0 + i;
      u32 d;
      if (lower) d = lane - (63u - (u32)__clzll((long long)lower));   // nearest lower lane of the group
      else d = old == ZMX_NONE16 ? 0u : r - old;
      if (d > 32767u) d = 0;
      wave_lds_sync();                               // every lane has read the old heads
      if (act && (grp & gt_mask) == 0) head[key] = (u16)r;   // the last of each group
      wave_lds_sync();
      const u64 k = t0 + i;
      if (act && k >= e0) {
        if (chain == 0) {
          lk[k].x = (u16)d;
          lk[k].z = same[k];
        } else {
          lk[k].y = (u16)d;
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
#define SCRATCH_CPS 256u
#define MATCH_BATCH 8u                        // lanes that wait for a record write / a new position before the wave serves them                     // per-lane overflow change points

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
  const u32* tile_list;  // optional: the tiles to do (total_tiles entries); null = all of them
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
    if (s_tile >= t_end) break;
    const u32 tile = P.tile_list ? P.tile_list[s_tile] : s_tile;

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

    bool pending = false;          // the walk has ended, its record is not written yet
    for (;;) {
      // Writing a record and fetching the next position are long, rarely needed code: a wave that
      // ran them whenever one lane asked would execute them on almost every step with one or two
      // lanes active.  Lanes queue up instead (MATCH_BATCH of them, or nobody left walking).
      const u64 m_need = __ballot(pending || (!active && !done));
      const bool service = m_need != 0 && ((u32)__popcll(m_need) >= MATCH_BATCH || !__any(active));
      if (service && pending) {
        pending = false;
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
      }
      if (service && !active && !done) {
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
            pending = true;
            active = false;
          }
        }
      }
    }
  }
}

// Match records of a block that lies inside a block of an earlier table: the records of all
// positions but the last few (see BuildTablesFrom) are the same — copy them.
struct CopyRecsParams {
  const BlockDesc* blocks;   // the new blocks
  const u64* src_pos;        // [nb] record index in `src` of each block's first position
  const uint4* src;
  uint4* dst;
};

__global__ __launch_bounds__(256) void k_copy_recs(CopyRecsParams P) {
  const BlockDesc bd = P.blocks[blockIdx.y];
  const u64 n = (u64)(bd.inend - bd.instart) * 2;   // uint4 per record: 2
  const uint4* src = P.src + P.src_pos[blockIdx.y] * 2;
  uint4* dst = P.dst + bd.pos_off * 2;
  for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += (u64)gridDim.x * 256) dst[i] = src[i];
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
// K3  ZopfliLZ77Greedy (lz77.c:544-630) on the match table: zmx_greedy.h
//     (segmented: k_greedy_exits / k_greedy_link / k_greedy_emit).  Here only the
//     window loader they share.
// ----------------------------------------------------------------------------
__device__ __forceinline__ uint2 greedy_window(const u32* rbase, u32 wb, u32 lane, u32 B) {
  const u32 q = wb + lane < B ? wb + lane : (B ? B - 1 : 0);
  return *reinterpret_cast<const uint2*>(rbase + (u64)q * 8);
}

// ----------------------------------------------------------------------------
// K4  one LZ77OptimalRun per block (squeeze.c:429): GetBestLengths forward DP
//     (:217) + TraceBackwards (:317) + FollowPath (:338) + histogram, split
//     by how each part parallelises:
//
//     k_rowscan (once per table build)  row layout of the DP edges: position j
//         of a block owns the edges k = 1 (literal), 3..kend (matches,
//         kend = min(leng, inend - i), squeeze.c:286); dph[j] = {roff, kend |
//         shortcut flag << 16} with roff the exclusive prefix sum of the row
//         lengths (k = 2 is a dead slot so that row[k-1] addresses edge k).
//     k_edges   (every run, all CUs)    cost(k, sublen[k]) of every edge
//         (squeeze.c:146-157; depends on the run's cost model, not on the DP
//         state), one lane per edge, written as doubles to rows[] in HBM.
//     k_dp   (every run, one wave per block)  the serial part: the chain
//         through the float-rounded absolute costs.  The live cells
//         costs[j .. j+258] stay in registers (lane l owns cells base + 64 s +
//         l of the current 64-position group), the cost of the expanding
//         position is a v_readlane, edge rows stream HBM -> LDS ring by
//         LDS-DMA one ring ahead, and 8 positions of row values are preloaded
//         into registers so that no memory latency sits on the chain.
//     zmx_dp3.h  k_dp3: the same chain with the row fetching on producer waves (default).
//     zmx_trace.h  TraceBackwards + FollowPath + histogram, segmented (k_trace_exits /
//         k_trace_link / k_trace_emit).
// ----------------------------------------------------------------------------
#define TR_CHUNK 2048u

// ---------------------------------------------------------------- k_rowscan
struct RowScanParams {
  const BlockDesc* blocks;
  const u32* recs;
  uint2* dph;          // per block position: {roff, kend | shortcut << 16}
  u64* block_edges;    // per block: total row length
};

__global__ __launch_bounds__(1024) void k_rowscan(RowScanParams P) {
  __shared__ u32 s_wsum[16];
  const BlockDesc bd = P.blocks[blockIdx.x];
  const u32 B = (u32)(bd.inend - bd.instart);
  const u32 tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const u32* rbase = P.recs + bd.pos_off * 8;
  uint2* out = P.dph + bd.pos_off;
  u32 carry = 0;
  for (u32 t0 = 0; t0 < B; t0 += 1024) {
    const u32 jj = t0 + tid;
    const bool act = jj < B;
    u32 kend = 0, sflag = 0;
    if (act) {
      const uint2 h = *reinterpret_cast<const uint2*>(rbase + (u64)jj * 8);
      const u32 leng = h.x & 0xffffu, same_i = h.y & 0xffffu;
      kend = leng < B - jj ? leng : B - jj;        // squeeze.c:286
      if (kend < 3) kend = 1;
      // long-run shortcut test (squeeze.c:251-258): i > instart + 259, i + 517 < inend
      if (same_i > 2 * ZMX_MAX_MATCH && jj > ZMX_MAX_MATCH + 1 && jj + 2 * ZMX_MAX_MATCH + 1 < B) {
        const u32 same_back = rbase[(u64)(jj - ZMX_MAX_MATCH) * 8 + 1] & 0xffffu;
        sflag = same_back > ZMX_MAX_MATCH ? 1u : 0u;
      }
    }
    const u32 incl = wave_scan_add(kend);
    if (lane == 63) s_wsum[wid] = incl;
    __syncthreads();
    u32 woff = 0, tot = 0;
#pragma unroll
    for (u32 w = 0; w < 16; ++w) {
      const u32 v = s_wsum[w];
      if (w < wid) woff += v;
      tot += v;
    }
    if (act) out[jj] = make_uint2(carry + woff + incl - kend, kend | (sflag << 16));
    carry += tot;
    __syncthreads();
  }
  if (tid == 0) P.block_edges[blockIdx.x] = carry;
}

// ------------------------------------------------------------------ k_edges
#define EG_CAP 2048u   // edges expanded per wave at a time

struct EdgeParams {
  const BlockDesc* blocks;
  const u32* tile_off;     // [nb_total + 1] cumulative 2048-position tiles
  u32 nb_total;
  u32 tile0;               // first tile of this launch
  const u32* recs;
  const u32* pool;
  const uint2* dph;
  const double* cost;      // [nb_total][320]
  double* rows;
  const u64* row_base;     // [nb_total] first row slot of each block (in doubles)
  const double* mincost;   // [nb_total]
  u32* badpos;             // bit per position (pos_off + p): a match edge of it costs less than mincost (zeroed per run)
};

__global__ __launch_bounds__(256) void k_edges(EdgeParams P) {
  __shared__ double s_ll[288];
  __shared__ double s_d[32];
  __shared__ double s_kll[260];              // ll[length symbol of k]
  __shared__ u8 s_klb[260];                  // length extra bits of k
  __shared__ u8 s_mark[4][EG_CAP];           // row starts, for the edge -> position map
  __shared__ u32 s_hoff[4][64];              // row offset in the sub-chunk | literal << 16 | overflow << 24
  __shared__ u32 s_hthr[4][64][2];           // 8 change-point thresholds (len - 3), ascending, 0xff padded
  __shared__ u16 s_hdist[4][64][8];          // their distances
  __shared__ u32 s_hpool[4][64][2];          // overflow records: pool offset, count

  const u32 tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const u32 tile = P.tile0 + blockIdx.x;
  u32 lo = 0, hi = P.nb_total;
  while (hi - lo > 1) {
    const u32 mid = (lo + hi) >> 1;
    if (P.tile_off[mid] <= tile) lo = mid; else hi = mid;
  }
  const u32 b = lo;
  const BlockDesc bd = P.blocks[b];
  const u32 B = (u32)(bd.inend - bd.instart);
  const u32 tp0 = (tile - P.tile_off[b]) * MT;
  const u32* rbase = P.recs + bd.pos_off * 8;
  const uint2* dbase = P.dph + bd.pos_off;
  double* rows = P.rows + P.row_base[b];
  // squeeze.c:293's mincost test is a no-op unless an edge costs less than mincost (possible only
  // through rounding in the cost model): such positions are reported, k_dp3 tests them literally
  const double mincost = P.mincost[b];

  for (u32 i = tid; i < 288; i += 256) s_ll[i] = P.cost[(u64)b * 320 + i];
  if (tid < 32) s_d[tid] = P.cost[(u64)b * 320 + 288 + tid];
  __syncthreads();
  for (u32 k = tid; k < 260; k += 256) {
    const bool ok = k >= 3 && k <= ZMX_MAX_MATCH;
    s_kll[k] = ok ? s_ll[dev_length_symbol(k)] : 0.0;
    s_klb[k] = ok ? (u8)dev_length_extra_bits(k) : (u8)0;
  }
  __syncthreads();
  const double kInf = __longlong_as_double(0x7ff0000000000000ll);
  u8* mark = s_mark[wid];

  for (u32 g = wid; g < MT / 64; g += 4) {
    const u32 base = tp0 + g * 64;
    if (base >= B) break;
    const u32 navail = (B - base < 64u) ? B - base : 64u;
    const u32 jj = base + lane;
    const bool act = lane < navail;
    uint4 ra = make_uint4(0, 0, 0, 0), rb = make_uint4(0, 0, 0, 0);
    uint2 dh = make_uint2(0, 0);
    if (act) {
      ra = *reinterpret_cast<const uint4*>(rbase + (u64)jj * 8);
      rb = *reinterpret_cast<const uint4*>(rbase + (u64)jj * 8 + 4);
      dh = dbase[jj];
    }
    const u32 lit = (ra.y >> 16) & 255u;
    const u32 ncpf = ra.y >> 24;
    const u32 kend = dh.y & 0xffffu;
    const u32 off = dh.x;                       // block-relative row offset
    const u32 offend = off + (act ? kend : 0u);
    const u32 off0 = rdlane_u32(off, 0);

    wave_lds_sync();  // the previous group's headers are dead
    {
      const bool ovf = ncpf == 0xffu;
      s_hoff[wid][lane] = (off - off0) | (lit << 16) | ((ovf ? 1u : 0u) << 24);
      if (ovf) {
        s_hpool[wid][lane][0] = ra.z;
        s_hpool[wid][lane][1] = ra.w & 0xffffu;
        s_hthr[wid][lane][0] = 0xffffffffu;
        s_hthr[wid][lane][1] = 0xffffffffu;
      } else {
        const u32 w[6] = {ra.z, ra.w, rb.x, rb.y, rb.z, rb.w};
        u32 t0 = 0, t1 = 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const u32 bit = 24u * e;
          const u32 lo32 = w[bit >> 5] >> (bit & 31);
          const u32 v = (bit & 31) > 8 ? (lo32 | (w[(bit >> 5) + 1 > 5 ? 5 : (bit >> 5) + 1] << (32 - (bit & 31)))) : lo32;
          const u32 thr = (u32)e < ncpf ? (v & 255u) : 255u;
          if (e < 4) t0 |= thr << (8 * e); else t1 |= thr << (8 * (e - 4));
          s_hdist[wid][lane][e] = (u16)((v >> 8) & 0xffffu);
        }
        s_hthr[wid][lane][0] = t0;
        s_hthr[wid][lane][1] = t1;
      }
    }

    u32 q = 0;
    while (q < navail) {
      const u32 off_q = rdlane_u32(off, q);
      const u64 fit = __ballot(act && lane >= q && offend - off_q <= EG_CAP);
      const u32 n = (u32)__popcll(fit);
      const u32 E = rdlane_u32(offend, q + n - 1) - off_q;
      wave_lds_sync();
      for (u32 e = lane; e < (E + 3) / 4; e += 64) reinterpret_cast<u32*>(mark)[e] = 0;
      wave_lds_sync();
      if (lane >= q && lane < q + n) mark[off - off_q] = (u8)(lane + 1);
      wave_lds_sync();
      const u32 rel_q = off_q - off0;
      u32 carry = q + 1;
      u64 ovf_mask = __ballot(lane >= q && lane < q + n && ncpf == 0xffu);
      for (u32 t0 = 0; t0 < E; t0 += 64) {
        const u32 e = t0 + lane;
        const u32 mk = e < E ? (u32)mark[e] : 0u;
        u32 own = wave_scan_max(mk);
        own = own > carry ? own : carry;
        carry = rdlane_u32(own, 63);
        if (e < E) {
          const u32 p = own - 1;
          const u32 h = s_hoff[wid][p];
          const u32 k = e - ((h & 0xffffu) - rel_q) + 1;
          double w;
          if (k == 1) {
            w = s_ll[(h >> 16) & 255u];            // literal edge, squeeze.c:278
          } else if (k == 2 || (h >> 24)) {
            w = kInf;                              // dead slot (overflow rows are filled below)
          } else {
            const u32 x = k - 3;
            // first change point with len >= k: thresholds ascending, binary search over 8 bytes
            const u32 tlo = s_hthr[wid][p][0], thi = s_hthr[wid][p][1];
            u32 idx = ((tlo >> 24) < x) ? 4u : 0u;
            u32 half = idx ? thi : tlo;
            if (((half >> 8) & 255u) < x) { idx += 2; half >>= 16; }
            if ((half & 255u) < x) idx += 1;
            const u32 dist = s_hdist[wid][p][idx];
            // squeeze.c:155: (lbits + dbits) as int, then + ll, then + d
            w = ((double)((int)s_klb[k] + dev_dist_extra_bits(dist)) + s_kll[k]) + s_d[dev_dist_symbol(dist)];
            if (w < mincost) {
              const u64 gp = bd.pos_off + base + p;
              atomicOr(&P.badpos[gp >> 5], 1u << (gp & 31));
            }
          }
          rows[(u64)off_q + e] = w;
        }
      }
      // rows of records with more than 8 change points (pool): whole wave per position
      while (ovf_mask) {
        const u32 p = (u32)__ffsll((long long)ovf_mask) - 1;
        ovf_mask &= ovf_mask - 1;
        const u32 roff_p = rdlane_u32(off, p);
        const u32 ke = rdlane_u32(kend, p);
        const u32 poff = s_hpool[wid][p][0], pn = s_hpool[wid][p][1];
        for (u32 k = 3 + lane; k <= ke; k += 64) {
          u32 plo = 0, phi = pn;   // first entry with len >= k (entries ascending in len)
          while (plo < phi) {
            const u32 mid = (plo + phi) >> 1;
            if ((P.pool[poff + mid] & 0xffffu) < k) plo = mid + 1; else phi = mid;
          }
          const u32 dist = plo < pn ? P.pool[poff + plo] >> 16 : 1u;
          const double w = ((double)((int)s_klb[k] + dev_dist_extra_bits(dist)) + s_kll[k]) + s_d[dev_dist_symbol(dist)];
          rows[(u64)roff_p + k - 1] = w;
          if (w < mincost) {
            const u64 gp = bd.pos_off + base + p;
            atomicOr(&P.badpos[gp >> 5], 1u << (gp & 31));
          }
        }
      }
      q += n;
    }
  }
}

// ------------------------------------------------------------------ k_dp
#define ZMX_PROF_N 32u                // u64 profiling counters per block (ZOPFLI_AMD_PROF)
#define DP_RING 4096u                 // doubles in the LDS ring (32 KB)
#define DP_PIECE 128u                 // doubles per LDS-DMA instruction (64 lanes x 16 B)
#define DP_SPAN (DP_RING / 2 - 256u)  // row span of one sub-chunk: the next one is always resident too
#define DP_XN 704u                    // long-run shortcut staging: 384 cells
#define DP_FRONT 64u                  // slack before the ring: masked-off lanes address up to 64 slots back
#define DP_MIRROR 384u                // the first 3 pieces are mirrored behind the ring: a row never wraps

struct DpParams {
  const BlockDesc* blocks;
  u32 block0;              // first block of this launch
  const uint2* dph;
  const double* cost;      // [nb_total][320]
  const double* mincost;   // [nb_total]
  const double* rows;
  const u64* row_base;
  const u64* block_edges;
  u16* la;
  u64* prof;               // optional [nb_total][8] cycle counters (ZOPFLI_AMD_PROF=1), else null
  int debug_nofetch;       // timing experiment (profiling build only): producers build nothing, results are wrong
  const u32* recs;         // k_sq: the match records and the change-point pool
  const u32* pool;
  const u32* badpos;       // k_edges' bad-edge bitmap (k_dp3)
};

// One position of the chain on cell register `CS` (round S): the edge values `WV`
// were preloaded, invalid lanes hold +inf.  MCL = mincost (or -inf on the literal
// lane: squeeze.c:277-284 has no mincost test) so MCL + cj is mincostaddcostj.
// "costs[j+k] > mincostaddcostj && newCost < costs[j+k]" (squeeze.c:293,298) as
// one compare: max(newCost, mincostaddcostj) < costs[j+k] (no NaNs here).
#define DP_RELAX(CS, LS, WV, MCL)                                           \
  {                                                                         \
    const double old_ = (double)(CS);                                       \
    const double nc_ = (WV) + cj;                    /* squeeze.c:278,297 */ \
    const bool upd_ = fmax(nc_, (MCL) + cj) < old_;                         \
    CS = upd_ ? (float)nc_ : CS;                                            \
    LS = upd_ ? src1 : LS;                                                  \
  }

// One 1 KiB piece HBM -> LDS by LDS-DMA (global_load_lds_dwordx4: lane i moves 16 B to
// M0 + 16 i).  Issued as asm so that the compiler does not order every later LDS read
// behind it with vmcnt(0); the consumer side waits explicitly before a barrier.
__device__ __forceinline__ void dp_dma_piece(const double* lane_src, u32 lds_byte_off) {
  lds_byte_off = (u32)__builtin_amdgcn_readfirstlane((int)lds_byte_off);   // wave-uniform by construction
  u32 keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(lane_src), "s"(lds_byte_off)
      : "memory");
}

// Eight consecutive positions p0..p0+7 of a group, none with an edge beyond cell
// register 1 (TWO) / 0 (!TWO): straight-line code.  The row values are fetched
// first (one readlane + one ds_read per position and register), then the chain
// runs on registers only.
template <bool TWO>
__device__ __forceinline__ void dp_fast_block(const double* ring0, const uint2* tab, u32 p0, u32 lane, u32 base,
                                              double mincost, float& c0, u32& l0, float& c1, u32& l1) {
  const double kInf = __longlong_as_double(0x7ff0000000000000ll);
  double w0[8], w1[8], mcl0[8];
  u32 ke8[8];
  const u32 d0 = lane - p0 - 1;                         // k - 1 of register 0 at u = 0
  // tab[p] = {byte offset of row[0] from ring0, kend} (an LDS broadcast read: a v_readlane would
  // cost an SGPR round trip of ~35 cycles per position on a lone wave)
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const uint2 t = tab[p0 + u];
    ke8[u] = t.y;
    const double* row = reinterpret_cast<const double*>(reinterpret_cast<const char*>(ring0) + (int)t.x);
    w0[u] = row[lane];                                  // row[lane] = edge k = lane - p
    if (TWO) w1[u] = row[lane + 64];
  }
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const u32 km1 = d0 - u;
    w0[u] = km1 < ke8[u] ? w0[u] : kInf;
    mcl0[u] = km1 == 0 ? -kInf : mincost;
    if (TWO) w1[u] = km1 + 64 < ke8[u] ? w1[u] : kInf;
  }
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const u32 p = p0 + u;
    const double cj = (double)rdlane_f32(c0, p);
    const u32 src1 = base + p + 1;
    DP_RELAX(c0, l0, w0[u], mcl0[u])
    if (TWO) {
      const double mcl1 = lane + 63u == p ? -kInf : mincost;   // position 63's literal edge
      DP_RELAX(c1, l1, w1[u], mcl1)
    }
  }
}

template <bool PROF>
__global__ __launch_bounds__(64) void k_dp(DpParams P) {
  __shared__ __align__(16) double s_ring[DP_FRONT + DP_RING + DP_MIRROR];
  __shared__ float s_xc[DP_XN];
  __shared__ u16 s_xl[DP_XN];
  __shared__ uint2 s_tab[64];   // per position of the group: row byte offset from the ring start, kend

  const u32 b = P.block0 + blockIdx.x;
  const BlockDesc bd = P.blocks[b];
  const u32 B = (u32)(bd.inend - bd.instart);
  const u32 lane = threadIdx.x;
  if (B == 0) return;
  const uint2* dbase = P.dph + bd.pos_off;
  u16* la = P.la + bd.la_off;
  const double* rows = P.rows + P.row_base[b];
  const u32 total_pad = (u32)((P.block_edges[b] + DP_PIECE - 1) & ~(u64)(DP_PIECE - 1));

  const double mincost = P.mincost[b];
  // squeeze.c:260: cost of (length 258, dist 1) = (0 + 0) + ll[285] + d[0]
  const double symbolcost258 = (double)(0 + 0) + P.cost[(u64)b * 320 + 285] + P.cost[(u64)b * 320 + 288];
  const double kInf = __longlong_as_double(0x7ff0000000000000ll);
  const u32 ring_lds = (u32)(unsigned long)(__attribute__((address_space(3))) double*)s_ring;

  u64 t_stage = 0, t_chain = 0, t_mark = 0, n_fast = 0, n_slow = 0, t_fast = 0, n_two = 0, t_two = 0;
  const bool prof = PROF && P.prof != nullptr;   // the counters exist only in the profiling instantiation
#define DP_TICK() (PROF ? (u64)__builtin_readcyclecounter() : 0ull)

  // cost cells of the current group: c[s] of lane l = cell base + 64 s + l; l[s] = 1 + the
  // position the cell was reached from (0 = never), so length_array = cell + 1 - l[s]
  float c[6];
  u32 l[6];
#pragma unroll
  for (int s = 0; s < 6; ++s) { c[s] = 1e30f; l[s] = 0; }
  if (lane == 0) c[0] = 0.0f;

  u32 base = 0;
  u32 loaded_end = 0;          // rows [.., loaded_end) have been requested into the ring
  bool noshort = false;        // the reference tests the shortcut once per loop iteration (squeeze.c:247-251)
  u32 pf_base = 0xffffffffu;
  uint2 pf_dh = make_uint2(0, 0);

  while (base <= B) {
    t_mark = DP_TICK();
    const u32 navail = (B - base < 64u) ? B - base : 64u;
    const u32 jj = base + lane;
    const bool act = lane < navail;
    uint2 dh = pf_dh;
    if (pf_base != base) dh = dbase[jj < B ? jj : B - 1];   // first group, or a shortcut moved the group start
    pf_base = base + 64;
    pf_dh = dbase[jj + 64 < B ? jj + 64 : B - 1];            // next group, one group ahead (clamped, unconditional)

    const u32 kend = act ? (dh.y & 0xffffu) : 0u;
    const bool sflag = act && (dh.y >> 16) != 0;
    const u32 roff = dh.x;
    const u32 offend = roff + kend;
    __syncthreads();   // the previous group's table is dead
    s_tab[lane] = make_uint2(((roff & (DP_RING - 1)) - lane - 1) * 8u, kend);
    const u64 m_short = __ballot(sflag);
    const u64 m_r1 = __ballot(kend + lane >= 64u);     // position needs cell register 1
    const u64 m_r2 = __ballot(kend + lane >= 128u);    // ... and 2 or more: generic path

    u32 q = 0;
    bool regroup = false;
    while (q < navail) {
      // ---- sub-chunk: positions q .. q+n-1 whose rows span at most DP_SPAN from a_cur
      const u32 off_q = rdlane_u32(roff, q);
      const u32 a_cur = off_q & ~(DP_PIECE - 1);
      const u64 fit = __ballot(act && lane >= q && offend - a_cur <= DP_SPAN);
      const u32 n = (u32)__popcll(fit);
      const u32 need_end = (rdlane_u32(offend, q + n - 1) + DP_PIECE - 1) & ~(DP_PIECE - 1);
      // everything requested so far has landed; top the ring up to a_cur + DP_RING
      // (the youngest VMEM op of a group's first sub-chunk is the dph prefetch: leave it in flight)
      if (q == 0) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      {
        const u32 lim = a_cur + DP_RING < total_pad ? a_cur + DP_RING : total_pad;
        const bool cold = loaded_end < need_end;
        while (loaded_end < lim) {
          const u32 slot = loaded_end & (DP_RING - 1);
          dp_dma_piece(rows + loaded_end + lane * 2, ring_lds + (DP_FRONT + slot) * 8);
          if (slot < DP_MIRROR) dp_dma_piece(rows + loaded_end + lane * 2, ring_lds + (DP_FRONT + DP_RING + slot) * 8);
          loaded_end += DP_PIECE;
        }
        if (cold) {  // first sub-chunk of the block (or a jump after a shortcut)
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __syncthreads();
        }
      }
      { const u64 t = DP_TICK(); t_stage += t - t_mark; t_mark = t; }

      // ---- the serial chain over positions q .. q+n-1, 8 at a time
      u32 p0 = q;
      while (p0 < q + n) {
        const bool full = p0 + 8 <= q + n;
        const u32 sbits = (u32)(m_short >> p0) & 255u;
        const u32 r2bits = (u32)(m_r2 >> p0) & 255u;
        if (full && sbits == 0 && r2bits == 0) {
          // ---- fast path: straight-line code, edge values preloaded
          const u64 tf0 = DP_TICK();
          if (((u32)(m_r1 >> p0) & 255u) != 0) {
            dp_fast_block<true>(s_ring + DP_FRONT, s_tab, p0, lane, base, mincost, c[0], l[0], c[1], l[1]);
            n_two += 8;
            t_two += DP_TICK() - tf0;
          } else {
            dp_fast_block<false>(s_ring + DP_FRONT, s_tab, p0, lane, base, mincost, c[0], l[0], c[1], l[1]);
          }
          t_fast += DP_TICK() - tf0;
          noshort = false;
          n_fast += 8;
          p0 += 8;
          continue;
        }
        // ---- generic path (ragged tail, long matches, shortcut candidates)
        const u32 pend = full ? p0 + 8 : q + n;
        u32 p = p0;
        for (; p < pend; ++p) {
          if (((m_short >> p) & 1) && !noshort) { regroup = true; break; }
          noshort = false;
          const u32 ke = rdlane_u32(kend, p);
          const u32 ro = rdlane_u32(roff, p);
          const double cj = (double)rdlane_f32(c[0], p);
          const u32 src1 = base + p + 1;
          const u32 km1 = lane - p - 1;
          const u32 smax = (ke + p) >> 6;
#pragma unroll
          for (int s = 0; s < 6; ++s) {
            if ((u32)s <= smax) {
              const u32 k1 = km1 + 64u * s;
              if (k1 < ke) {
                const double w = s_ring[DP_FRONT + ((ro + k1) & (DP_RING - 1))];
                const double mcl = k1 == 0 ? -kInf : mincost;
                DP_RELAX(c[s], l[s], w, mcl)
              }
            }
          }
        }
        n_slow += p - p0;
        p0 = p;
        if (regroup) break;
      }
      { const u64 t = DP_TICK(); t_chain += t - t_mark; t_mark = t; }
      if (regroup) {
        // ---- long-run shortcut at position p0 of the group (squeeze.c:251-271)
        const u32 p = p0;
        const u32 j = base + p;
        if (lane < p && base + lane >= 1) la[base + lane] = (u16)(l[0] ? base + lane + 1 - l[0] : 0u);
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 6; ++s) {
          const u32 x = base + 64u * s + lane;
          s_xc[64 * s + lane] = c[s];
          s_xl[64 * s + lane] = (u16)(l[s] ? x + 1 - l[s] : 0u);
        }
        __syncthreads();
        // costs[j+t+258] = costs[j+t] + symbolcost for t = 0..257, unconditionally; cells
        // j..j+257 are consumed with the lengths they have now
        float nc4[5];
#pragma unroll
        for (int r = 0; r < 5; ++r) {
          const u32 t = 64u * r + lane;
          nc4[r] = 1e30f;
          if (t < ZMX_MAX_MATCH) {
            la[j + t] = s_xl[p + t];
            nc4[r] = (float)((double)s_xc[p + t] + symbolcost258);
          }
        }
        // new group at j + 258: cell j+258+t <- nc4 (reached from j+t), everything beyond is untouched
#pragma unroll
        for (int s = 0; s < 6; ++s) { c[s] = 1e30f; l[s] = 0; }
#pragma unroll
        for (int r = 0; r < 5; ++r) {
          const u32 t = 64u * r + lane;
          if (t < ZMX_MAX_MATCH) { c[r] = nc4[r]; l[r] = j + t + 1; }
        }
        base = j + ZMX_MAX_MATCH;
        noshort = true;   // squeeze.c:273 continues with the match query at the new i
        // rows between the old and the new position are never read: restart the ring there
        {
          const u32 nb_ = base < B ? base : B - 1;
          const u32 ro = dbase[nb_].x & ~(DP_PIECE - 1);
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __syncthreads();
          if (ro > loaded_end) loaded_end = ro;
        }
        break;
      }
      q += n;
    }
    if (regroup) continue;

    // ---- group done: cells base..base+63 are final
    if (jj <= B && jj >= 1) la[jj] = (u16)(l[0] ? jj + 1 - l[0] : 0u);
#pragma unroll
    for (int s = 0; s < 5; ++s) { c[s] = c[s + 1]; l[s] = l[s + 1]; }
    c[5] = 1e30f;
    l[5] = 0;
    base += 64;
  }
  if (lane == 0) la[0] = 0;
  if (prof && lane == 0) {
    u64* o = P.prof + (u64)b * ZMX_PROF_N;
    o[0] = t_stage; o[1] = t_chain; o[2] = n_fast; o[3] = n_slow; o[4] = B; o[5] = t_fast; o[6] = n_two; o[7] = t_two;
  }
}

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

// link_lo (optional, per block): only the links from this index on will be read (tables built from a parent:
// k_match2 recomputes the tiles at the block's end only) — k_chain skips the chunks that end below it, k_same
// what lies more than a chunk and its warm-up window below it.
__global__ __launch_bounds__(256) void k_same(const u8* __restrict__ in, const BlockDesc* __restrict__ blocks,
                                              u16* __restrict__ same16, const u64* __restrict__ link_lo) {
  const BlockDesc bd = blocks[blockIdx.y];
  const u64 L = bd.inend - bd.ws;
  const u64 c0 = ((u64)blockIdx.x * blockDim.x + threadIdx.x) * SAME_CH;
  if (c0 >= L) return;
  const u64 c1 = (c0 + SAME_CH < L) ? c0 + SAME_CH : L;
  if (link_lo) {
    const u64 lo = link_lo[blockIdx.y];
    if (lo >= L) return;
    const u64 chunk0 = lo / 32768u * 32768u;                 // first chunk k_chain keeps (CH_EMIT)
    if (c1 + 32768u <= chunk0) return;                       // below its warm-up window
  }
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
#define CH_LDS_BYTES (65536 + CH_TILE * 4)     // the head table, a tile's keys, a tile's run lengths

__global__ __launch_bounds__(64) void k_chain(const u8* __restrict__ in, const BlockDesc* __restrict__ blocks,
                                              const u16* __restrict__ same16, ushort4* __restrict__ links,
                                              const u64* __restrict__ link_lo) {
  extern __shared__ __align__(16) u8 dyn_lds[];
  u16* head = reinterpret_cast<u16*>(dyn_lds);   // last position (relative to w0) of every key, 0xffff = none
  u16* keys = head + 32768;
  u16* sames = keys + CH_TILE;     // same[] of the tile's positions (chain 0 writes them into the link records)

  const BlockDesc bd = blocks[blockIdx.y];
  const u32 chain = blockIdx.z;
  const u64 L = bd.inend - bd.ws;
  const u64 e0 = (u64)blockIdx.x * CH_EMIT;
  if (e0 >= L) return;
  const u64 e1 = (e0 + CH_EMIT < L) ? e0 + CH_EMIT : L;
  if (link_lo && e1 <= link_lo[blockIdx.y]) return;     // nobody reads this chunk's links (see k_same)
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
    {
      // The tile's keys: lane l takes positions 16 l .. 16 l + 15 with all its loads in flight at once (18 bytes,
      // 16 run lengths).  One position per lane and round was 16 rounds of dependent loads per tile, and with the
      // two waves a CU holds (64 KB head table each) nothing hid them: most of the kernel's time.
      const u32 i0 = 16u * lane;
      u32 by[18];
      u32 sm[16];
#pragma unroll
      for (u32 q = 0; q < 18; ++q) {
        const u64 k = t0 + i0 + q;
        by[q] = (i0 < tn && k < L) ? (u32)base[k] : 0u;          // (zero past the block end: hash.c:100-104's caller)
      }
#pragma unroll
      for (u32 q = 0; q < 16; ++q) sm[q] = (i0 + q < tn) ? (u32)same[t0 + i0 + q] : 0u;
#pragma unroll
      for (u32 q = 0; q < 16; ++q) {
        if (i0 + q < tn) {
          u32 v = ((by[q] << 10) ^ (by[q + 1] << 5) ^ by[q + 2]) & 32767u;   // hash.c:96-98, three rolling updates
          if (chain) v ^= (sm[q] - 3u) & 255u;                                // hash.c:129
          keys[i0 + q] = (u16)v;
          sames[i0 + q] = (u16)sm[q];
        }
      }
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
          lk[k].z = sames[i];      // (from the staged tile: a global load here stalled every step of a wave that
                                   //  shares its CU with one other wave)
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
#define SCRATCH_CPS 256u                      // per-lane overflow change points
#define M_XCD_GROUP 16u                        // consecutive tiles that go to one XCD (k_match*)
#define MATCH_BATCH 8u                        // lanes that wait for a record write / a new position before the wave serves them

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
  u32* scratch;          // gridDim.x * M2_THREADS * SCRATCH_CPS
  const u32* tile_list;  // optional: the tiles to do (total_tiles entries); null = all of them
  const unsigned long long* skip_energy;   // optional (k_hits): blocks whose estimate exceeds skip_thr x positions are k_match5's
  u64 skip_thr;
};

// (ds_read_u8: the byte itself, no shifting around a 32-bit read)
__device__ __forceinline__ u32 lds_byte(const u32* w, u32 a) { return reinterpret_cast<const u8*>(w)[a]; }
__device__ __forceinline__ u32 lds_u32_unaligned(const u32* w, u32 a) {
  const u32 lo = w[a >> 2], hi = w[(a >> 2) + 1];
  return __builtin_amdgcn_alignbyte(hi, lo, a & 3);
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
//     k_codes   (once per table build)  which of the run's 1127 weights every edge takes
//         (squeeze.c:146-157 depends on the cost model only through the symbols), one lane per
//         edge, 16 bits each; k_wtab (every run) the 1127 weights per block.
//     zmx_dp4.h  (every run)  the chain through the float-rounded absolute costs, cut into
//         verified tasks (k_dp4_spec / k_dpcheck / k_dp4_fix).
//     zmx_trace.h  TraceBackwards + FollowPath + histogram, segmented (k_trace_exits /
//         k_trace_link / k_trace_emit).
// ----------------------------------------------------------------------------
#define TR_CHUNK 2048u

// ---------------------------------------------------------------- k_rowscan
struct RowScanParams {
  const BlockDesc* blocks;
  const u32* recs;
  uint2* dph;          // per block position: {roff, kend | shortcut << 16 | run row << 17 | literal << 18 | codeless << 26}
  u64* block_edges;    // per block: total row length
  u32 codeless;        // 1 = wide run rows (64+ edges) get no codes: DPH_CODELESS (0: the serial chain k_dp4 reads every row's codes)
};
// A WIDE RUN ROW — one change point, at distance 1, 64 or more edges: the inside of a run of equal bytes — has no codes in
// codes[]: its weights are w(length symbol of k, distance symbol 0), which every chain kernel takes from tables by length
// (zmx_dp5.h: s_rk / s_ri / s_w1), never from the row.  Round 5 wrote them all the same — 258 two-byte codes for every such
// position, 49 GB per 100 MB of long runs, 16 x the input, ~90 % of it never read (profiles/r05_classZ100MB_pmc.json) —
// and the code budget then split such batches.  The row takes no room in the layout (its offset is its successor's).
#define DPH_CODELESS (1u << 26)
__device__ __forceinline__ u32 dph_layout_len(u32 y) { return (y & DPH_CODELESS) ? 0u : (y & 0xffffu); }

__global__ __launch_bounds__(1024) void k_rowscan(RowScanParams P) {
  __shared__ u32 s_wsum[16];
  const BlockDesc bd = P.blocks[blockIdx.x];
  const u32 B = (u32)(bd.inend - bd.instart);
  const u32 tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const u32* rbase = P.recs + bd.pos_off * 8;
  uint2* out = P.dph + bd.pos_off;
  u64 carry = 0;     // (64 bits: the host refuses a block whose rows do not fit 32-bit offsets, it has to see the real sum)
  for (u32 t0 = 0; t0 < B; t0 += 1024) {
    const u32 jj = t0 + tid;
    const bool act = jj < B;
    u32 kend = 0, sflag = 0;
    if (act) {
      const uint2 h = *reinterpret_cast<const uint2*>(rbase + (u64)jj * 8);
      const u32 leng = h.x & 0xffffu, same_i = h.y & 0xffffu;
      // a "run row": one change point, at distance 1 — every match edge of the position is (k, distance 1), k = 3 .. kend
      // (the inside of a run of equal bytes): the chain kernels take its weights from a 258-entry table instead of
      // the row's codes (zmx_dp5.h).  And the literal, so that they need not look at the codes at all.
      if (leng >= 3 && (h.x >> 16) == 1u && (h.y >> 24) == 1u) sflag |= 2u;
      sflag |= ((h.y >> 16) & 255u) << 2;
      kend = leng < B - jj ? leng : B - jj;        // squeeze.c:286
      if (kend < 3) kend = 1;
      // long-run shortcut test (squeeze.c:251-258): i > instart + 259, i + 517 < inend
      if (same_i > 2 * ZMX_MAX_MATCH && jj > ZMX_MAX_MATCH + 1 && jj + 2 * ZMX_MAX_MATCH + 1 < B) {
        const u32 same_back = rbase[(u64)(jj - ZMX_MAX_MATCH) * 8 + 1] & 0xffffu;
        sflag |= same_back > ZMX_MAX_MATCH ? 1u : 0u;
      }
    }
    const bool nocodes = P.codeless != 0 && (sflag & 2u) != 0 && kend >= 64u;
    const u32 klay = nocodes ? 0u : kend;          // what the row takes in codes[]
    const u32 incl = wave_scan_add(klay);
    if (lane == 63) s_wsum[wid] = incl;
    __syncthreads();
    u32 woff = 0, tot = 0;
#pragma unroll
    for (u32 w = 0; w < 16; ++w) {
      const u32 v = s_wsum[w];
      if (w < wid) woff += v;
      tot += v;
    }
    if (act) out[jj] = make_uint2((u32)carry + woff + incl - klay, kend | (sflag << 16) | (nocodes ? DPH_CODELESS : 0u));
    carry += tot;
    __syncthreads();
  }
  if (tid == 0) P.block_edges[blockIdx.x] = carry;
}

// ------------------------------------------------------------------ k_codes
// The DP edges of a block as 16-bit WEIGHT CODES, written once per table build: the cost of an edge
// (squeeze.c:146-157) depends on the run's cost model only through the symbols it is coded with, so
// what is stored per edge is which of the 1127 weights of the run it takes:
//     0                                      no edge (the dead slot k = 2 of a row)
//     1 + byte                               literal (squeeze.c:278)
//     257 + 30 (lsym - 257) + dsym           match of a length with symbol lsym at a distance with symbol dsym
// stored times 8, i.e. as the byte offset of the weight in the run's table of doubles (k_wtab).  Two
// bytes per edge instead of a double per edge per run: the squeeze runs read a quarter of the bytes,
// write none, and the table build pays once.  A lane outside its row reads code 0 (the chain
// kernels fetch a row as a buffer of its own: the range check returns zero), whose weight is +inf.
#define EG_CAP 2048u   // edges expanded per wave at a time
#define ZMX_NUM_W 1127u                 // weights of a run: 0 = +inf, 256 literals, 29 x 30 matches
#define ZMX_WTAB 1152u                  // doubles per block in wtab[] (padded)

__device__ __forceinline__ u32 weight_code(u32 k, u32 dist) {
  return (257u + 30u * (u32)(dev_length_symbol(k) - 257) + (u32)dev_dist_symbol(dist)) * 8u;
}

struct CodeParams {
  const BlockDesc* blocks;
  const u32* tile_off;     // [nb_total + 1] cumulative 2048-position tiles
  u32 nb_total;
  const u32* recs;
  const u32* pool;
  const uint2* dph;
  u16* codes;
  const u64* code_base;    // [nb_total] first slot of each block in codes[]
};

__global__ __launch_bounds__(256) void k_codes(CodeParams P) {
  __shared__ u8 s_mark[4][EG_CAP];           // row starts, for the edge -> position map
  __shared__ u32 s_hoff[4][64];              // row offset in the sub-chunk | literal << 16 | overflow << 24
  __shared__ u32 s_hthr[4][64][2];           // 8 change-point thresholds (len - 3), ascending, 0xff padded
  __shared__ u16 s_hdist[4][64][8];          // their distances
  __shared__ u32 s_hpool[4][64][2];          // overflow records: pool offset, count

  const u32 tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const u32 tile = blockIdx.x;
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
  u16* codes = P.codes + P.code_base[b];
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
    const u32 klay = dph_layout_len(dh.y);      // (0: a wide run row, no codes)
    const u32 off = dh.x;                       // block-relative row offset
    const u32 offend = off + (act ? klay : 0u);
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
      if (lane >= q && lane < q + n && klay != 0) mark[off - off_q] = (u8)(lane + 1);
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
          u32 code;
          if (k == 1) {
            code = (1u + ((h >> 16) & 255u)) * 8u;   // literal edge, squeeze.c:278
          } else if (k == 2 || (h >> 24)) {
            code = 0;                                // dead slot (overflow rows are filled below)
          } else {
            const u32 x = k - 3;
            // first change point with len >= k: thresholds ascending, binary search over 8 bytes
            const u32 tlo = s_hthr[wid][p][0], thi = s_hthr[wid][p][1];
            u32 idx = ((tlo >> 24) < x) ? 4u : 0u;
            u32 half = idx ? thi : tlo;
            if (((half >> 8) & 255u) < x) { idx += 2; half >>= 16; }
            if ((half & 255u) < x) idx += 1;
            code = weight_code(k, s_hdist[wid][p][idx]);
          }
          codes[(u64)off_q + e] = (u16)code;
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
          codes[(u64)roff_p + k - 1] = (u16)weight_code(k, dist);
        }
      }
      q += n;
    }
  }
}

// ------------------------------------------------------------------ k_wtab
// The run's weights per block (GetCostStat, squeeze.c:146-157: (lbits + dbits) as int, then + ll,
// then + d) and which match weights lie below mincost: squeeze.c:293's mincost test is a no-op for
// every other edge (zmx_dp4.h), the positions that own an edge below it (possible only through
// rounding in the cost model) are marked by k_badscan and take the chain kernels' literal path.
struct WtabParams {
  const double* cost;      // [nb][320]
  const double* mincost;   // [nb]
  double* wtab;            // [nb][ZMX_WTAB]
  u32* badcodes;           // [nb][40]: word 0 = number of weights below mincost, words 4.. = their bitmap
  u32* stats;              // the run's 8 task statistics (k_dp4_fix adds to them): zeroed here, the run's first kernel
};

__global__ __launch_bounds__(256) void k_wtab(WtabParams P) {
  __shared__ u32 s_bad[40];
  const u32 b = blockIdx.x, tid = threadIdx.x;
  if (tid < 40) s_bad[tid] = 0;
  if (b == 0 && tid < 8 && P.stats) P.stats[tid] = 0;
  __syncthreads();
  const double* ll = P.cost + (u64)b * 320;
  const double* d = ll + 288;
  const double mincost = P.mincost[b];
  for (u32 c = tid; c < ZMX_WTAB; c += 256) {
    double w = __longlong_as_double(0x7ff0000000000000ll);    // code 0 and the padding: no edge
    if (c >= 1 && c <= 256) {
      w = ll[c - 1];
    } else if (c >= 257 && c < ZMX_NUM_W) {
      const u32 ls = 257u + (c - 257u) / 30u, ds = (c - 257u) % 30u;
      const int lb = ls < 265 || ls == 285 ? 0 : (int)(ls - 261) / 4;
      const int db = ds < 4 ? 0 : (int)ds / 2 - 1;
      w = ((double)(lb + db) + ll[ls]) + d[ds];
      if (w < mincost) {
        atomicAdd(&s_bad[0], 1u);
        atomicOr(&s_bad[4 + (c >> 5)], 1u << (c & 31));
      }
    }
    P.wtab[(u64)b * ZMX_WTAB + c] = w;
  }
  __syncthreads();
  if (tid < 40) P.badcodes[(u64)b * 40 + tid] = s_bad[tid];
}

// ---------------------------------------------------------------- k_badscan
// Only for blocks whose run has a weight below mincost: mark the positions that own such an edge.
struct BadScanParams {
  const BlockDesc* blocks;
  const u32* tile_off;
  u32 nb_total;
  const uint2* dph;
  const u16* codes;
  const u64* code_base;
  const u32* badcodes;
  u32* badpos;             // bit per position (pos_off + p), zeroed per run
};

__global__ __launch_bounds__(256) void k_badscan(BadScanParams P) {
  const u32 tile = blockIdx.x;
  u32 lo = 0, hi = P.nb_total;
  while (hi - lo > 1) {
    const u32 mid = (lo + hi) >> 1;
    if (P.tile_off[mid] <= tile) lo = mid; else hi = mid;
  }
  const u32 b = lo;
  const u32* bad = P.badcodes + (u64)b * 40;
  if (bad[0] == 0) return;
  const BlockDesc bd = P.blocks[b];
  const u32 B = (u32)(bd.inend - bd.instart);
  const u32 tp0 = (tile - P.tile_off[b]) * MT;
  const u16* codes = P.codes + P.code_base[b];
  for (u32 p = tp0 + threadIdx.x; p < tp0 + MT && p < B; p += 256) {
    const uint2 dh = P.dph[bd.pos_off + p];
    const u32 kend = dh.y & 0xffffu;
    bool any = false;
    if (dh.y & DPH_CODELESS) {
      // a wide run row: (k, distance 1) for k = 3 .. kend, i.e. every length symbol up to kend's at distance symbol 0
      const u32 smax = (u32)(dev_length_symbol(kend) - 257);
      for (u32 sy = 0; sy <= smax; ++sy) {
        const u32 c = 257u + 30u * sy;
        any |= ((bad[4 + (c >> 5)] >> (c & 31)) & 1u) != 0;
      }
    } else {
      for (u32 k = 3; k <= kend; ++k) {
        const u32 c = (u32)codes[(u64)dh.x + k - 1] >> 3;
        any |= ((bad[4 + (c >> 5)] >> (c & 31)) & 1u) != 0;
      }
    }
    if (any) {
      const u64 gp = bd.pos_off + p;
      atomicOr(&P.badpos[gp >> 5], 1u << (gp & 31));
    }
  }
}

// ------------------------------------------- shared by the chain kernels (zmx_dp4.h)
#define ZMX_PROF_N 56u                // u64 profiling counters per block (ZOPFLI_AMD_PROF)
#define DP_RING 16384u                // weight codes in the LDS ring (32 KB)
#define DP_PIECE 512u                 // codes per LDS-DMA instruction (64 lanes x 16 B)
#define DP_XN 704u                    // long-run shortcut staging: 384 cells
#define DP_FRONT 64u                  // slack before the ring: masked-off lanes address up to 64 slots back
#define DP_MIRROR 512u                // the first piece is mirrored behind the ring: a row (<= 258 slots) never wraps

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
__device__ __forceinline__ void dp_dma_piece(const u16* lane_src, u32 lds_byte_off) {
  lds_byte_off = (u32)__builtin_amdgcn_readfirstlane((int)lds_byte_off);   // wave-uniform by construction
  u32 keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(lane_src), "s"(lds_byte_off)
      : "memory");
}

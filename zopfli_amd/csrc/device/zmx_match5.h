// k_match5: the match table (ZopfliFindLongestMatch(limit 258, sublen) for every position, lz77.c:407-542) as an
// EXACT SKIP-WALK.  Included only by zmx_hip.hip, after zmx_match2.h (same MatchParams fields, window staging,
// record format, pool and scratch as k_match2).
//
// The reference visits every entry of a position's hash chain, newest first (99 per position on text, 2 000 on
// PNG-like data, 8 192 where the cap of lz77.c:527-530 binds), but its result depends only on
//   (1) the visited candidates whose common prefix with the position is LONGER than the best so far
//       (lz77.c:494-505: the change points of sublen),
//   (2) the candidate at which the walk changes to the second hash (lz77.c:509-519), and
//   (3) which candidate is the 8192nd (lz77.c:527-530) and which the last inside the window (:464).
// k_match2 emulates the visit sequence hit by hit.  Here only the candidates that can matter are touched:
//   * A candidate that beats bestlength shares bestlength + 1 bytes with the position.  Beside the reference's two
//     chains (k_chain) every position has LEVEL links (k_levels): the nearest earlier position whose first k bytes
//     hash alike, k = 4, 5, 6, 7, 8, 10, 12, 16.  The walk follows the level k <= bestlength + 1: a superset of the
//     candidates that can beat bestlength (hash collisions only add entries; every entry is compared with the
//     position's bytes before it counts), an order of magnitude shorter than the 3-byte chain.
//   * The first chain (before the hash switch) is walked hit by hit as the reference does: on text it is one
//     candidate long (same <= 1: the first candidate of the val2 class switches), on runs it is the reference's walk.
//   * On the second chain a visited candidate must be of the position's val2 class (lz77.c:521: the walk follows
//     hashval2's chain): a level entry that shares >= 3 bytes has the position's val, so the class test is
//     (same - 3) & 255 (hash.c:129).  Where the run length makes the second chain itself the more selective list
//     (same + 2 > k: inside runs a level chain links every position of every run of the byte), it is walked instead.
//   * The hits the walk jumped over are COUNTED, not visited: k_rank2 gives every position its rank within its val2
//     class (per 32768-position chunk of the region, plus the class's size in the chunk before), so the number of
//     chain entries between two members of the class is a subtraction, the 8192-candidate cap fires at the same
//     candidate as in the reference, and a candidate beyond it is never looked at.
// tools/match_skip_model.c is the CPU model of this walk (checked against the oracle's hit-by-hit walk on every
// position of every class, with the cap and the switch rule binding on classes B, Z and P); the kernels are checked
// against k_match2 and the oracle in tests/test_gpu_parity.py.
#pragma once

#define LV_N 10u
#define LV_HBITS 14u
#define LV_CH 65536u            // positions a k_levels job emits (it warms up over the 32768 before them)
#define LV_U 4u                 // steps of 64 positions whose loads and LDS atomics are in flight together
__device__ __constant__ const u32 kLevelK[LV_N] = {4, 5, 6, 7, 8, 10, 12, 16, 24, 32};
// level index for "needs n bytes": n = 4 .. 15 from a table (4 bits each, n - 4): 4→0 5→1 6→2 7→3 8,9→4 10,11→5 12..15→6;
// 16..23→7, 24..31→8, 32 and more→9
#define LV_LUT 0x666655443210ull
__device__ __forceinline__ int lv_level_for(u32 n) {
  return n < 16 ? (int)((LV_LUT >> (4u * (n - 4u))) & 15ull) : n < 24 ? 7 : n < 32 ? 8 : 9;
}

// One 32-byte record per region position for k_match5 (k_rank2 writes it), as 16-bit fields:
//   0 prev1, 1 prev2, 2 same (k_chain's links)   3 wc2, 4 tot2 (below)   5 unused   6 .. 15 the level links
#define XR_LV0 6u

struct LvBytes { uint4 a, b; };   // 32 bytes at a position

// the level's key: the first k bytes (masks m[0..6] for bytes 4.., 8.., ...) through multiply-xor rounds, top LV_HBITS bits
__device__ __forceinline__ u32 lv_key(const LvBytes& w, const u32* m) {
  u32 h = w.a.x * 0x9E3779B1u;
  h = (h ^ (w.a.y & m[0])) * 0x85EBCA6Bu;
  h = (h ^ (w.a.z & m[1])) * 0xC2B2AE35u;
  h = (h ^ (w.a.w & m[2])) * 0x27D4EB2Fu;
  h = (h ^ (w.b.x & m[3])) * 0x165667B1u;
  h = (h ^ (w.b.y & m[4])) * 0xD3A2646Du;
  h = (h ^ (w.b.z & m[5])) * 0xFD7046C5u;
  h = (h ^ (w.b.w & m[6])) * 0xB55A4F09u;
  h ^= h >> 15;
  h *= 0x2C1B3C6Du;
  return h >> (32u - LV_HBITS);
}

__device__ __forceinline__ LvBytes lv_load32(const u8* p) {
  LvBytes x;
  __builtin_memcpy(&x, p, 32);     // (unaligned: two global_load_dwordx4)
  return x;
}

// wave-wide minimum (every lane gets it)
__device__ __forceinline__ u32 wave_min_u32(u32 v) {
  const u32 m = wave_scan_max(~v);
  return ~rdlane_u32(m, 63);
}

// ----------------------------------------------------------------------------
// k_levels: level links.  One wave per (chunk of LV_CH positions, block, level): the head-table replay of
// hash.c:110-114 for another hash — 64 positions per step, the table (last position of every key, 2^LV_HBITS
// 32-bit entries) in LDS.  A step is ONE ds_max_rtn_u32 per lane: positions grow with the lane, so the maximum
// leaves the step's last position of every key in the table whatever the order the hardware applies the lanes in,
// and the value a lane gets back is the nearest earlier position of its key — unless two lanes of the step share
// the key, which shows as a returned position inside the step; those groups are redone with ballots.  Nothing a
// step issues depends on what the step before got back (the table is right after the atomic itself), so the loads
// and atomics of LV_U steps are in flight together: a lone wave per SIMD would otherwise sit out every latency.
//   lev[level * total_l + reg_off + (p - ws)] = distance to the previous position of the key, 0 = none within 32767
// ----------------------------------------------------------------------------
struct LevelParams {
  const u8* in;
  const BlockDesc* blocks;
  u16* lev;
  u64 total_l;
};

__global__ __launch_bounds__(64) void k_levels(LevelParams P) {
  __shared__ u32 head[1u << LV_HBITS];
  const BlockDesc bd = P.blocks[blockIdx.y];
  const u32 lvl = blockIdx.z;
  const u64 L = bd.inend - bd.ws;
  const u64 e0 = (u64)blockIdx.x * LV_CH;
  if (e0 >= L) return;
  const u64 e1 = (e0 + LV_CH < L) ? e0 + LV_CH : L;
  const u64 w0 = e0 >= ZMX_WINDOW ? e0 - ZMX_WINDOW : 0;
  const u32 lane = threadIdx.x;
  const u64 lt_mask = (1ull << lane) - 1;
  const u32 k = kLevelK[lvl];
  u32 m[7];
#pragma unroll
  for (u32 j = 0; j < 7; ++j) {
    const u32 lo = 4u * (j + 1u);     // the word holds bytes lo .. lo + 3
    m[j] = k >= lo + 4u ? 0xffffffffu : k > lo ? (1u << (8u * (k - lo))) - 1u : 0u;
  }
  for (u32 i = lane; i < (1u << LV_HBITS); i += 64) head[i] = 0;
  __syncthreads();
  const u8* base = P.in + bd.ws;
  u16* out = P.lev + (u64)lvl * P.total_l + bd.reg_off;

  LvBytes w[LV_U];
#pragma unroll
  for (u32 u = 0; u < LV_U; ++u) w[u] = lv_load32(base + w0 + 64u * u + lane);   // (the input is padded past its end)
  for (u64 s = w0; s < e1; s += 64u * LV_U) {
    u32 key[LV_U], ret[LV_U];
#pragma unroll
    for (u32 u = 0; u < LV_U; ++u) key[u] = lv_key(w[u], m);
    if (s + 64u * LV_U < e1) {
#pragma unroll
      for (u32 u = 0; u < LV_U; ++u) w[u] = lv_load32(base + s + 64u * (LV_U + u) + lane);   // the next round's bytes
    }
    const bool warm = s + 64u * LV_U <= e0;      // (e0 - w0 is a multiple of 64 LV_U: a round is all warm-up or none)
#pragma unroll
    for (u32 u = 0; u < LV_U; ++u) {
      const u64 p = s + 64u * u + lane;
      const u32 r = (u32)(p - w0) + 1u;          // 0 = no position yet
      ret[u] = 0;
      if (p < e1) {
        if (warm) atomicMax(&head[key[u]], r); else ret[u] = atomicMax(&head[key[u]], r);
      }
    }
    if (warm) continue;
#pragma unroll
    for (u32 u = 0; u < LV_U; ++u) {
      const u64 p = s + 64u * u + lane;
      const bool act = p < e1;
      const u32 r = (u32)(p - w0) + 1u;
      const u32 step0 = (u32)(s + 64u * u - w0) + 1u;
      u32 d = ret[u] ? r - ret[u] : 0u;
      u64 F = __ballot(act && ret[u] >= step0);                  // lanes that saw a position of this very step
      while (F) {
        const u32 l0 = (u32)__ffsll((unsigned long long)F) - 1u;
        const u32 k0 = rdlane_u32(key[u], l0);
        const u64 G = __ballot(act && key[u] == k0);
        const u32 old = wave_min_u32((G >> lane) & 1 ? ret[u] : 0xffffffffu);   // what the table held before the step
        if ((G >> lane) & 1) {
          const u64 lower = G & lt_mask;
          d = lower ? lane - (63u - (u32)__clzll((long long)lower)) : (old ? r - old : 0u);
        }
        F &= ~G;
      }
      if (d > 32767u) d = 0;
      if (act && p >= e0) out[p] = (u16)d;
    }
  }
}

// ----------------------------------------------------------------------------
// k_rank2: the position's record for k_match5 — k_chain's links, the level links, and the position's rank within
// its val2 class (the second hash's chain, hash.c:129-135) per 32768-position chunk of the region:
//   tot2 = number of positions of the chunk BEFORE the position's with its val2
//   wc2  = tot2 + number of positions before it in its own chunk with its val2
// so that the number of chain entries from a member c of pos's class down to a member q (both within 32767 of pos:
// in pos's chunk or the one before) is g(c) - g(q), g(x) = x in pos's chunk ? wc2[x] : wc2[x] - tot2[x].
// One wave per (chunk, block); 32768 16-bit counters, two to a word, in LDS (64 KB): a class has at most 32768
// members in a chunk, so a half never carries into its neighbour.  Three passes: count the chunk before; read
// those counts for the chunk's positions (tot2); count the chunk itself, a step at a time, the count before the
// step being the rank of the step's first member of a class — members of one class inside a step (prev2 says so)
// are ordered with ballots.
// ----------------------------------------------------------------------------
#define RK_CH 32768u
#define RK_U 4u

struct RankParams {
  const u8* in;
  const BlockDesc* blocks;
  const ushort4* links;
  const u16* lev;
  u64 total_l;
  u16* tot2;          // scratch, one per region position
  uint4* xrec;        // out: two per region position
};

__device__ __forceinline__ u32 rk_val2(u32 bytes, u64 p, u64 L, u32 same) {
  const u32 b0 = bytes & 255u, b1 = p + 1 < L ? (bytes >> 8) & 255u : 0u, b2 = p + 2 < L ? (bytes >> 16) & 255u : 0u;   // hash.c:107-108
  return ((((b0 << 10) ^ (b1 << 5) ^ b2) & 32767u) ^ ((same - 3u) & 255u));
}
__device__ __forceinline__ u32 rk_load_u32(const u8* p) {
  u32 x;
  __builtin_memcpy(&x, p, 4);
  return x;
}

__global__ __launch_bounds__(64) void k_rank2(RankParams P) {
  __shared__ u32 cnt[16384];
  const BlockDesc bd = P.blocks[blockIdx.y];
  const u64 L = bd.inend - bd.ws;
  const u64 e0 = (u64)blockIdx.x * RK_CH;
  if (e0 >= L) return;
  const u64 e1 = (e0 + RK_CH < L) ? e0 + RK_CH : L;
  const u32 lane = threadIdx.x;
  const u64 lt_mask = (1ull << lane) - 1;
  const u8* base = P.in + bd.ws;
  const ushort4* lk = P.links + bd.reg_off;
  u16* tt = P.tot2 + bd.reg_off;
  const u16* lev = P.lev + bd.reg_off;
  uint4* xr = P.xrec + bd.reg_off * 2;

  for (u32 i = lane; i < 16384; i += 64) cnt[i] = 0;
  __syncthreads();
  if (e0 > 0) {
    for (u64 s = e0 - RK_CH; s < e0; s += 64u * RK_U) {
      u32 by[RK_U], sm[RK_U];
#pragma unroll
      for (u32 u = 0; u < RK_U; ++u) {
        const u64 p = s + 64u * u + lane;
        by[u] = rk_load_u32(base + p);
        sm[u] = lk[p].z;
      }
#pragma unroll
      for (u32 u = 0; u < RK_U; ++u) {
        const u32 key = rk_val2(by[u], s + 64u * u + lane, L, sm[u]);
        atomicAdd(&cnt[key >> 1], 1u << (16u * (key & 1u)));
      }
    }
  }
  __syncthreads();
  for (u64 s = e0; s < e1; s += 64u * RK_U) {
    u32 by[RK_U], sm[RK_U];
#pragma unroll
    for (u32 u = 0; u < RK_U; ++u) {
      const u64 p = s + 64u * u + lane;
      by[u] = p < e1 ? rk_load_u32(base + p) : 0u;
      sm[u] = p < e1 ? (u32)lk[p].z : 0u;
    }
#pragma unroll
    for (u32 u = 0; u < RK_U; ++u) {
      const u64 p = s + 64u * u + lane;
      if (p < e1) {
        const u32 key = rk_val2(by[u], p, L, sm[u]);
        tt[p] = (u16)(cnt[key >> 1] >> (16u * (key & 1u)));
      }
    }
  }
  __syncthreads();
  for (u32 i = lane; i < 16384; i += 64) cnt[i] = 0;
  __syncthreads();
  for (u64 s = e0; s < e1; s += 128) {
    // two steps' loads in flight
    ushort4 l4[2];
    u32 by[2], tot[2];
    u32 lv[2][LV_N];
#pragma unroll
    for (u32 u = 0; u < 2; ++u) {
      const u64 p = s + 64u * u + lane;
      const bool act = p < e1;
      l4[u] = act ? lk[p] : make_ushort4(0, 0, 0, 0);
      by[u] = act ? rk_load_u32(base + p) : 0u;
      tot[u] = act ? (u32)tt[p] : 0u;
#pragma unroll
      for (u32 j = 0; j < LV_N; ++j) lv[u][j] = act ? (u32)lev[(u64)j * P.total_l + p] : 0u;
    }
#pragma unroll
    for (u32 u = 0; u < 2; ++u) {
      const u64 p = s + 64u * u + lane;
      const bool act = p < e1;
      const u32 key = act ? rk_val2(by[u], p, L, l4[u].z) : 0u;
      u32 rank = act ? (cnt[key >> 1] >> (16u * (key & 1u))) & 0xffffu : 0u;   // the state before this step
      wave_lds_sync();
      if (act) atomicAdd(&cnt[key >> 1], 1u << (16u * (key & 1u)));
      // positions of this step with the same key: the previous member of the class lies inside the step
      const bool dup = act && l4[u].y != 0 && l4[u].y <= lane;
      if (__any(dup)) {
        u64 grp = __ballot(act);
#pragma unroll
        for (int bit = 0; bit < 15; ++bit) {
          const bool mine = (key >> bit) & 1;
          const u64 bm = __ballot(mine);
          grp &= mine ? bm : ~bm;
        }
        rank += (u32)__popcll(grp & lt_mask);
      }
      wave_lds_sync();
      if (act) {
        uint4 a, b;
        a.x = (u32)l4[u].x | ((u32)l4[u].y << 16);
        a.y = (u32)l4[u].z | (((tot[u] + rank) & 0xffffu) << 16);
        a.z = tot[u];
        a.w = lv[u][0] | (lv[u][1] << 16);
        b.x = lv[u][2] | (lv[u][3] << 16);
        b.y = lv[u][4] | (lv[u][5] << 16);
        b.z = lv[u][6] | (lv[u][7] << 16);
        b.w = lv[u][8] | (lv[u][9] << 16);
        xr[2 * p] = a;
        xr[2 * p + 1] = b;
      }
    }
  }
}

// ----------------------------------------------------------------------------
// k_match5
// ----------------------------------------------------------------------------
struct Match5Params {
  MatchParams m;
  const uint4* xrec;     // k_rank2: two per region position
};

#define M5_THREADS 512
#define M5_BATCH 8u     // lanes that wait for a record write / a new position before the wave serves them
#define M5_IDLE 0u
#define M5_WALK 1u      // the loads of the entry at distance xd are in flight
#define M5_CMP 2u
#define M5_PEND 3u
#define M5_DONE 4u

// level link number j (0 .. 3) of the four 16-bit links in v
__device__ __forceinline__ u32 m5_pick(uint2 v, u32 j) {
  const u32 w = j & 2u ? v.y : v.x;
  return j & 1u ? w >> 16 : w & 0xffffu;
}

template <bool PROF>
__global__ __launch_bounds__(M5_THREADS, 6) void k_match5(Match5Params Q) {
  const MatchParams& P = Q.m;
  __shared__ __align__(16) u32 win[MWIN_BYTES / 4 + 4];
  __shared__ u32 s_cp[8 * M5_THREADS];       // the first 8 change points of every lane's position (len | dist << 16), slot-major
  __shared__ u32 s_next, s_tile;

  const u32 tid = threadIdx.x;
  const u32 xcd = blockIdx.x & 7;
  u32* my_scratch = P.scratch + ((u64)blockIdx.x * M5_THREADS + tid) * SCRATCH_CPS;

  for (;;) {
    __syncthreads();
    if (tid == 0) {
      const u32 k = atomicAdd(&P.counters[8 + xcd], 1u);
      s_tile = ((k / M_XCD_GROUP) * 8u + xcd) * M_XCD_GROUP + (k % M_XCD_GROUP);
      s_next = 0;
    }
    __syncthreads();
    if (s_tile >= P.total_tiles) break;
    const u32 tile = P.tile_list ? P.tile_list[s_tile] : s_tile;

    u32 lo = 0, hi = P.nb;
    while (hi - lo > 1) {
      const u32 mid = (lo + hi) >> 1;
      if (P.tile_off[mid] <= tile) lo = mid; else hi = mid;
    }
    const BlockDesc bd = P.blocks[lo];
    const u64 p0 = bd.instart + (u64)(tile - P.tile_off[lo]) * MT;
    const u64 p1 = (p0 + MT < bd.inend) ? p0 + MT : bd.inend;
    const u32 ntile = (u32)(p1 - p0);

    const long long wb = ((long long)p0 - (long long)ZMX_WINDOW) & ~15ll;
    const u64 hi_abs = (p1 + ZMX_MAX_MATCH < bd.inend) ? p1 + ZMX_MAX_MATCH : bd.inend;
    const u32 nvec = (u32)(((long long)hi_abs - wb + 15) >> 4);
    for (u32 v = tid; v < nvec; v += M5_THREADS) {
      const long long a = wb + (long long)v * 16;
      uint4 x = make_uint4(0, 0, 0, 0);
      if (a >= 0) x = *reinterpret_cast<const uint4*>(P.in + a);
      reinterpret_cast<uint4*>(win)[v] = x;
    }
    __syncthreads();

    const u8* xr;      // the records of the block's region, as bytes (scalar base, 32-bit lane offsets)
    {
      const u64 a = reinterpret_cast<u64>(Q.xrec + bd.reg_off * 2);
      const u32 alo = (u32)__builtin_amdgcn_readfirstlane((int)(u32)a), ahi = (u32)__builtin_amdgcn_readfirstlane((int)(u32)(a >> 32));
      xr = reinterpret_cast<const u8*>(((u64)ahi << 32) | alo);
    }
    const u32 li0 = (u32)(p0 - bd.ws);
    const u32 lp0 = (u32)((long long)p0 - wb);
    const u32 rem0 = (u32)((bd.inend - p0 < 70000) ? bd.inend - p0 : 70000);
    u32* const rec0 = P.recs + (bd.pos_off + (p0 - bd.instart)) * 8;

    // ---- per-lane walk state
    u32 st = M5_IDLE;
    u32 lp = 0, li = 0;
    u32 limit = 0, bestlen = 0, bestdist = 0, ncp = 0, same_pos = 0, cur = 0, size_rem = 0;
    u32 byte0 = 0, pbyte = 0, foff = 0, fmask = 0;
    u32 chain = 1, idx = 0;        // idx: candidates the reference has visited so far (lz77.c:527-530 stops at 8192)
    u32 curd = 0;                  // distance of the last visited candidate (0: none yet)
    u32 cprev = 0;                 // its prev1 | prev2 << 16 (raw steps follow them)
    u32 gcur = 0;                  // its g value on the second chain
    int lev_k = -1;                // level the walk follows, -1 = the reference's own chain
    u32 eqd = 0, nlink = 0;        // level walk: distance of the last entry and its link
    bool need_link = false;        // the link of (lev_k, eqd) has to be fetched first
    u32 xd = 0;                    // the entry being fetched
    uint4 A = make_uint4(0, 0, 0, 0);   // its record: prev1 | prev2, same | wc2, tot2, (lv0 | lv1)
    uint2 V = make_uint2(0, 0);    // four of its level links, from level vb on
    u32 vb = 0;
    u32 plv0 = 0, plv1 = 0, plv2 = 0, plv3 = 0, plv4 = 0;   // the position's own level links (lv0|lv1, lv2|lv3, ...)
    u32 n_touch = 0, n_iter = 0;

    // the entry the walk touches next: false = the walk is over (end of the chain, window, lz77.c:464, :521-523)
    auto next_entry = [&]() -> bool {
      if (lev_k < 0) {
        const u32 step = chain == 1 ? (cprev & 0xffffu) : (cprev >> 16);
        if (step == 0) return false;
        xd = curd + step;
        const u32 n = (bestlen < 3 ? 3u : bestlen) + 1u;
        const int g = lv_level_for(n > 32u ? 32u : n);     // the level an improvement here most likely asks for
        vb = g > 6 ? 6u : (u32)g;
      } else {
        if (need_link) {
          xd = eqd;
        } else {
          if (nlink == 0) return false;
          xd = eqd + nlink;
        }
        vb = lev_k > 6 ? 6u : (u32)lev_k;
      }
      if (xd >= ZMX_WINDOW) return false;
      const u8* r = xr + (u64)(li - xd) * 32u;
      A = *reinterpret_cast<const uint4*>(r);
      __builtin_memcpy(&V, r + 2u * (XR_LV0 + vb), 8);
      return true;
    };
    auto pos_link = [&](u32 k) -> u32 {
      const u32 w = k < 2 ? plv0 : k < 4 ? plv1 : k < 6 ? plv2 : k < 8 ? plv3 : plv4;
      return k & 1u ? w >> 16 : w & 0xffffu;
    };
    // level for the walk after bestlength or the chain changed; `at_x`: bestlength was just set by the entry (xd, V)
    auto choose_level = [&](bool at_x) {
      int k = -1;
      if (chain == 2 && bestlen >= 3) {
        k = lv_level_for(bestlen + 1 > 32 ? 32u : bestlen + 1);
        if (same_pos + 2 > kLevelK[k]) k = -1;       // the second chain is the more selective list
      }
      if (k != lev_k) {
        lev_k = k;
        if (k >= 0) {
          need_link = false;
          if (kLevelK[k] > bestlen) {
            // k = bestlength + 1 bytes: the position's own chain of that level; entries at or above the last
            // visited candidate are passed over
            eqd = 0;
            nlink = pos_link((u32)k);
          } else if (at_x && (u32)k >= vb && (u32)k < vb + 4u) {
            // the candidate that set bestlength shares bestlength >= k bytes with the position: its level-k chain
            // is the position's
            eqd = xd;
            nlink = m5_pick(V, (u32)k - vb);
          } else {
            eqd = bestdist;
            need_link = true;
          }
        }
      } else if (k >= 0 && at_x) {
        eqd = xd;
        nlink = m5_pick(V, (u32)k - vb);     // (k = lev_k lies in V's range: vb = min(lev_k, 6))
      }
    };

    for (;;) {
      const u64 m_need = __ballot(st == M5_PEND || st == M5_IDLE);
      const u64 m_run = __ballot(st == M5_WALK || st == M5_CMP);
      if (m_need != 0 && ((u32)__popcll(m_need) >= M5_BATCH || m_run == 0)) {
        if (st == M5_PEND) {
          st = M5_IDLE;
          // the record: the first 8 change points, 3 bytes each (length - 3, distance), from the lane's slots
          u32 v[8];
#pragma unroll
          for (u32 e = 0; e < 8; ++e) {
            const u32 x = s_cp[e * M5_THREADS + tid];
            v[e] = e < ncp ? (((x & 0xffffu) - 3u) | ((x >> 16) << 8)) : 0u;
          }
          uint4 r0, r1;
          r0.x = bestlen | (bestdist << 16);
          r0.z = v[0] | (v[1] << 24);
          r0.w = (v[1] >> 8) | (v[2] << 16);
          r1.x = (v[2] >> 16) | (v[3] << 8);
          r1.y = v[4] | (v[5] << 24);
          r1.z = (v[5] >> 8) | (v[6] << 16);
          r1.w = (v[6] >> 16) | (v[7] << 8);
          if (ncp <= 8) {
            r0.y = same_pos | (byte0 << 16) | (ncp << 24);
          } else {
            r0.y = same_pos | (byte0 << 16) | (0xffu << 24);
            const u32 off = atomicAdd(&P.counters[0], ncp);
            if (off + ncp <= P.pool_cap) {
#pragma unroll
              for (u32 e = 0; e < 8; ++e) P.pool[off + e] = s_cp[e * M5_THREADS + tid];
              for (u32 e = 8; e < ncp; ++e) P.pool[off + e] = my_scratch[e];
              r0.z = off;
              r0.w = ncp;
            } else {
              atomicOr(&P.counters[1], 1u);
              r0.z = 0;
              r0.w = 0;
            }
          }
          u32* const rec = rec0 + (u64)(lp - lp0) * 8;
          reinterpret_cast<uint4*>(rec)[0] = r0;
          reinterpret_cast<uint4*>(rec)[1] = r1;
        }
        if (st == M5_IDLE) {
          const u32 i = atomicAdd(&s_next, 1u);
          if (i >= ntile) {
            st = M5_DONE;
          } else {
            lp = lp0 + i;
            li = li0 + i;
            size_rem = rem0 - i;
            u32* const rec = rec0 + (u64)i * 8;
            const uint4* rp = reinterpret_cast<const uint4*>(xr + (u64)li * 32u);
            const uint4 Ap = rp[0], Bp = rp[1];
            same_pos = Ap.y & 0xffffu;
            byte0 = lds_byte(win, lp);
            ncp = 0;
            bestlen = 1; bestdist = 0; chain = 1; idx = 0; curd = 0; lev_k = -1; need_link = false;
            if (size_rem < 3) {                      // lz77.c:440-446
              rec[0] = 0;
              rec[1] = same_pos | (byte0 << 16);
            } else {
              limit = size_rem < ZMX_MAX_MATCH ? size_rem : ZMX_MAX_MATCH;  // lz77.c:448-450
              cprev = Ap.x;
              if ((Ap.x & 0xffffu) == 0) {           // empty chain
                rec[0] = 1;
                rec[1] = same_pos | (byte0 << 16);
              } else {
                plv0 = Ap.w; plv1 = Bp.x; plv2 = Bp.y; plv3 = Bp.z; plv4 = Bp.w;
                pbyte = m2_lds_u32(win, lp); foff = 0; fmask = 0xffffu;   // bestlength 1: bytes 0 and 1
                next_entry();                        // (prev1 < 32768: always an entry)
                st = M5_WALK;
              }
            }
          }
        }
        continue;
      }
      if (m_run == 0) {
        if (m_need == 0) break;
        continue;
      }

      // ---- the fetched entry
      const bool walk = st == M5_WALK;
      bool fin = false;            // the walk of this lane is over
      bool moved = false;          // this iteration ended with a decision: fetch the next entry
      bool visited = false;        // the entry is a candidate the reference visits (and it is not longer than the best)
      bool pass = false;
      if (PROF) { n_touch += walk ? 1u : 0u; ++n_iter; }
      if (walk) {
        if (lev_k >= 0 && need_link) {               // the link of the entry point itself
          need_link = false;
          nlink = m5_pick(V, (u32)lev_k - vb);
          moved = true;
        } else {
          bool cand = true;
          if (lev_k >= 0) {
            eqd = xd;
            nlink = m5_pick(V, (u32)lev_k - vb);
            if (xd <= curd) { cand = false; moved = true; }   // at or above the last visited candidate (entered at pos)
          }
          if (cand) {
            const u32 cw = m2_lds_u32(win, lp - xd + foff);
            pass = ((cw ^ pbyte) & fmask) == 0;
            if (!pass) {
              if (lev_k < 0) visited = true; else moved = true;
            }
          }
        }
      }
      if (pass) {
        cur = 0;
        if (same_pos > 2 && lds_byte(win, lp - xd) == byte0) {     // lz77.c:481-490
          const u32 lz = A.y & 0xffffu;
          const u32 s = same_pos < lz ? same_pos : lz;
          cur = s < limit ? s : limit;
        }
        st = M5_CMP;
      }
      if (st == M5_CMP) {
        const u32 rem = limit - cur;
        bool end = rem == 0;
        if (!end) {
          const u64 x = m2_lds_u64(win, lp + cur) ^ m2_lds_u64(win, lp - xd + cur);
          u32 m = x ? (u32)(__ffsll((unsigned long long)x) - 1) >> 3 : 8u;
          if (m > rem) m = rem;
          cur += m;
          end = m < 8 || cur >= limit;
        }
        if (end) {
          st = M5_WALK;
          if (cur > bestlen) {
            // a longer match: is it a candidate the reference visits, and which one?
            bool ok = true;
            u32 hops = 1;
            // g on the second chain (k_rank2): positions of pos's chunk count from the chunk before's total
            const u32 gx = ((li - xd) >> 15 == li >> 15 ? A.y >> 16 : (A.y >> 16) - A.z) & 0xffffu;
            if (chain == 2 && lev_k >= 0) {
              ok = (((A.y & 0xffffu) - 3u) & 255u) == ((same_pos - 3u) & 255u);   // of the position's val2 class
              hops = (gcur - gx) & 0xffffu;
            }
            if (!ok) {
              moved = true;                        // not on the second chain: never visited
            } else if (idx + hops > ZMX_MAX_CHAIN_HITS) {
              fin = true;                          // beyond the 8192nd candidate (lz77.c:527-530)
            } else {
              idx += hops;
              if (cur >= 3) {
                if (ncp < 8) s_cp[ncp * M5_THREADS + tid] = cur | (xd << 16);
                else if (ncp < SCRATCH_CPS) my_scratch[ncp] = cur | (xd << 16);
                ++ncp;
              }
              bestlen = cur;
              bestdist = xd;
              foff = cur >= 3 ? cur - 3u : 0u;
              fmask = cur >= 3 ? 0xffffffffu : 0xffffffu;
              pbyte = m2_lds_u32(win, lp + foff);
              curd = xd;
              cprev = A.x;
              gcur = gx;
              if (cur >= limit) {
                fin = true;                        // lz77.c:500-502
              } else {
                // lz77.c:509-519 (on chain 1 the 3-byte hashes are equal: val2 equality is equality of (same - 3) & 255)
                if (chain == 1 && bestlen >= same_pos && (((A.y & 0xffffu) - 3u) & 255u) == ((same_pos - 3u) & 255u)) chain = 2;
                choose_level(true);
                moved = true;
              }
            }
          } else if (lev_k < 0) {
            visited = true;                        // compared, not longer
          } else {
            moved = true;
          }
        }
      }
      if (visited) {
        // a candidate of the reference's own chain that does not beat bestlength: count it, test the switch rule
        if (idx + 1 > ZMX_MAX_CHAIN_HITS) {
          fin = true;
        } else {
          ++idx;
          curd = xd;
          cprev = A.x;
          gcur = ((li - xd) >> 15 == li >> 15 ? A.y >> 16 : (A.y >> 16) - A.z) & 0xffffu;
          if (chain == 1 && bestlen >= same_pos && (((A.y & 0xffffu) - 3u) & 255u) == ((same_pos - 3u) & 255u)) {
            chain = 2;
            choose_level(false);
          }
          moved = true;
        }
      }
      if (moved && !fin) fin = !next_entry();
      if (fin) st = M5_PEND;
    }
    if (PROF) {
      atomicAdd(reinterpret_cast<unsigned long long*>(P.counters + 4), (unsigned long long)n_touch);
      if ((tid & 63) == 0) atomicAdd(reinterpret_cast<unsigned long long*>(P.counters + 6), (unsigned long long)n_iter);
    }
  }
}

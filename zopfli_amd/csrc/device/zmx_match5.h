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
//     hash alike, k = 4, 5, 6, 7, 8, 10, 12, 16, 24, 32.  The walk follows the level k <= bestlength + 1: a superset
//     of the candidates that can beat bestlength (hash collisions only add entries; every entry is compared with the
//     position's bytes before it counts), an order of magnitude shorter than the 3-byte chain.
//   * On the FIRST chain (before the hash switch, lz77.c:509-519) two kinds of candidates matter: level entries
//     that beat bestlength, while bestlength <= same, and the SWITCH POINT — the nearest position below the last
//     visited candidate that is of both of pos's classes (its 3-byte hash and its val2): found by walking pos's
//     second chain for a member with pos's 3-byte hash.  Once bestlength >= same + 1 a longer match has exactly pos's
//     run length, so it is of pos's val2 class and the switch point comes first anyway: the walk goes there
//     directly.
//   * On the second chain a visited candidate must be of the position's val2 class (lz77.c:521: the walk follows
//     hashval2's chain): a level entry that shares >= 3 bytes has the position's val, so the class test is
//     (same - 3) & 255 (hash.c:129).  Where the run length makes the second chain itself the more selective list
//     (same + 2 > k: inside runs a level chain links every position of every run of the byte), it is walked instead.
//   * The hits the walk jumped over are COUNTED, not visited: k_rank2 gives every position its rank within each of
//     its two classes (per 32768-position chunk of the region, plus the class's size in the chunk before), so the
//     number of chain entries between two members of a class is a subtraction, the 8192-candidate cap fires at the
//     same candidate as in the reference, and a candidate beyond it is never looked at.
// Which blocks: k_hits estimates the hits per position the reference's walk would make (sum over 32768-position
// chunks of the squared val2 class sizes / positions); a block above ZOPFLI_AMD_MATCH_HITS (300) takes this kernel,
// the others k_match2, side by side on two streams (zmx_hip.hip: BuildTables).
// tools/match_skip_model.c is the CPU model of this walk (checked against the oracle's hit-by-hit walk on every
// position of every class, with the cap and the switch rule binding on classes B, Z and P); the kernels are checked
// against k_match2 and the oracle in tests/test_gpu_parity.py and tests/test_gpu_match_adversarial.py.
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
//   0 prev1, 1 prev2, 2 same (k_chain's links)   3 rank1, 4 rank2 (k_rank2)   5 unused   6 .. 15 the level links
#define XR_LV0 6u

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
  const unsigned long long* energy;   // k_hits (null: every block)
  u64 thr;
};

// does block b get the skip-walk (k_match5) or the hit-by-hit walk (k_match2)?  k_hits' estimate of the hits per
// position against the threshold, decided on the device by every kernel that asks
__device__ __forceinline__ bool m5_block_on(const unsigned long long* energy, u64 thr, u32 b, u64 L) {
  return energy == nullptr || energy[b] > thr * L;
}

// NW = dwords of a position's bytes the level's key reads (1, 2, 3, 4, 6 or 8: levels 4 / 5-8 / 10, 12 / 16 / 24 / 32)
template <u32 NW>
__device__ __forceinline__ void lv_load(u32 (&w)[NW], const u8* p) {
  if (NW <= 4) {
    uint4 x;
    __builtin_memcpy(&x, p, 16);
    w[0] = x.x;
    if (NW > 1) w[1] = x.y;
    if (NW > 2) w[2] = x.z;
    if (NW > 3) w[3] = x.w;
  } else {
    uint4 x, y;
    __builtin_memcpy(&x, p, 16);
    __builtin_memcpy(&y, p + 16, 16);
    w[0] = x.x; w[1] = x.y; w[2] = x.z; w[3] = x.w;
    w[4] = y.x; w[5] = y.y;
    if (NW > 6) { w[6] = y.z; w[7] = y.w; }
  }
}
template <u32 NW>
__device__ __forceinline__ u32 lv_key(const u32 (&w)[NW], u32 mlast) {
  const u32 mul[8] = {0x9E3779B1u, 0x85EBCA6Bu, 0xC2B2AE35u, 0x27D4EB2Fu, 0x165667B1u, 0xD3A2646Du, 0xFD7046C5u, 0xB55A4F09u};
  u32 h = 0;
#pragma unroll
  for (u32 j = 0; j < NW; ++j) h = (h ^ (j + 1 == NW ? w[j] & mlast : w[j])) * mul[j];
  h ^= h >> 15;
  h *= 0x2C1B3C6Du;
  return h >> (32u - LV_HBITS);
}

// one round of LV_U steps of 64 positions from job position s on; GUARD: the round may reach beyond the job's end n
template <u32 NW, bool GUARD>
__device__ __forceinline__ void lv_round(u32 (&w)[LV_U][NW], const u8* base, u16* out, u32* head, u32 s, u32 n, u32 mlast, u32 lane, u64 lt_mask) {
  u32 key[LV_U], ret[LV_U];
#pragma unroll
  for (u32 u = 0; u < LV_U; ++u) key[u] = lv_key<NW>(w[u], mlast);
  if (s + 64u * LV_U < n) {
#pragma unroll
    for (u32 u = 0; u < LV_U; ++u) lv_load<NW>(w[u], base + s + 64u * (LV_U + u) + lane);   // the next round's bytes
  }
  // what the table holds BEFORE each step (the LDS queue is in order: the read of step u comes after the atomic of
  // step u - 1), then the atomic, whose returned value only says whether two lanes of the step share a key
  u32 old[LV_U];
#pragma unroll
  for (u32 u = 0; u < LV_U; ++u) {
    const u32 q = s + 64u * u + lane;
    ret[u] = 0;
    old[u] = 0;
    if (!GUARD || q < n) {
      old[u] = head[key[u]];                          // (program order: after step u - 1's atomic, before this one's)
      ret[u] = atomicMax(&head[key[u]], q + 1u);     // 0 = no position yet
    }
  }
#pragma unroll
  for (u32 u = 0; u < LV_U; ++u) {
    const u32 q = s + 64u * u + lane;
    const bool act = !GUARD || q < n;
    const u32 r = q + 1u;
    const u32 step0 = s + 64u * u + 1u;
    u32 d = old[u] ? r - old[u] : 0u;
    u64 F = __ballot(act && ret[u] >= step0);                  // lanes that saw a position of this very step
    while (F) {
      const u32 l0 = (u32)__ffsll((unsigned long long)F) - 1u;
      const u32 k0 = rdlane_u32(key[u], l0);
      const u64 G = __ballot(act && key[u] == k0);
      const u64 lower = G & lt_mask;
      if (((G >> lane) & 1) && lower) d = lane - (63u - (u32)__clzll((long long)lower));   // the nearest lower lane of the key
      F &= ~G;
    }
    if (d > 32767u) d = 0;
    if (act) out[q] = (u16)d;
  }
}

template <u32 NW>
__device__ __forceinline__ void lv_job(const LevelParams& P, const BlockDesc& bd, u32 lvl, u32* head) {
  const u64 L = bd.inend - bd.ws;
  const u64 e0 = (u64)blockIdx.x * LV_CH;
  const u32 n1 = (u32)((e0 + LV_CH < L ? e0 + LV_CH : L) - e0);     // positions to emit
  const u32 nw = e0 >= ZMX_WINDOW ? ZMX_WINDOW : 0u;                 // warm-up positions before them
  const u32 lane = threadIdx.x;
  const u64 lt_mask = (1ull << lane) - 1;
  const u32 k = kLevelK[lvl];
  const u32 mlast = (k & 3u) ? (1u << (8u * (k & 3u))) - 1u : 0xffffffffu;   // the level's last dword holds k mod 4 bytes (or 4)
  const u8* base = P.in + bd.ws + e0 - nw;                           // job position 0 (uniform: scalar base, 32-bit offsets)
  u16* out = P.lev + (u64)lvl * P.total_l + bd.reg_off + e0 - nw;
  const u32 n = nw + n1;

  u32 w[LV_U][NW];
#pragma unroll
  for (u32 u = 0; u < LV_U; ++u) lv_load<NW>(w[u], base + 64u * u + lane);   // (the input is padded past its end)
  u32 s = 0;
  // warm-up (a multiple of 64 LV_U positions): only the table matters — atomics whose results nobody waits for
  for (; s < nw; s += 64u * LV_U) {
    u32 key[LV_U];
#pragma unroll
    for (u32 u = 0; u < LV_U; ++u) key[u] = lv_key<NW>(w[u], mlast);
#pragma unroll
    for (u32 u = 0; u < LV_U; ++u) lv_load<NW>(w[u], base + s + 64u * (LV_U + u) + lane);   // (n > nw: there is a next round)
#pragma unroll
    for (u32 u = 0; u < LV_U; ++u) atomicMax(&head[key[u]], s + 64u * u + lane + 1u);
  }
  for (; s + 64u * LV_U <= n; s += 64u * LV_U) lv_round<NW, false>(w, base, out, head, s, n, mlast, lane, lt_mask);
  if (s < n) lv_round<NW, true>(w, base, out, head, s, n, mlast, lane, lt_mask);
}

__global__ __launch_bounds__(64) void k_levels(LevelParams P) {
  __shared__ u32 head[1u << LV_HBITS];
  const BlockDesc bd = P.blocks[blockIdx.y];
  const u32 lvl = blockIdx.z;
  const u64 L = bd.inend - bd.ws;
  if ((u64)blockIdx.x * LV_CH >= L) return;
  if (!m5_block_on(P.energy, P.thr, blockIdx.y, L)) return;
  for (u32 i = threadIdx.x; i < (1u << LV_HBITS); i += 64) head[i] = 0;
  __syncthreads();
  switch (lvl) {
    case 0: lv_job<1>(P, bd, lvl, head); break;
    case 1: case 2: case 3: case 4: lv_job<2>(P, bd, lvl, head); break;
    case 5: case 6: lv_job<3>(P, bd, lvl, head); break;
    case 7: lv_job<4>(P, bd, lvl, head); break;
    case 8: lv_job<6>(P, bd, lvl, head); break;
    default: lv_job<8>(P, bd, lvl, head); break;
  }
}

// ----------------------------------------------------------------------------
// k_hits: what would the reference's walk cost here?  Sum over the 32768-position chunks of a block's region of the
// squared sizes of the val2 classes (hash.c:129) = sum over positions of the number of positions of their chunk on
// their second chain: per position that is the chain length the walk has in front of it — 100 on text, 186 on
// markup, 2 000 on PNG-like and two-symbol data, 45 on long runs, within a few percent of the hits the
// instrumented reference counts.  energy[b] > threshold x positions sends block b to the skip-walk.
// ----------------------------------------------------------------------------
struct HitsParams {
  const u8* in;
  const BlockDesc* blocks;
  const u16* same16;
  unsigned long long* energy;
  u32* cmax;          // [blocks][gridDim.x] the largest class of the chunk: of the 3-byte hash | of val2 << 16 (k_rank2: can the 8192-hit cap bind here?)
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

#define RK_CH 32768u
#define RK_THREADS 256u

__global__ __launch_bounds__(RK_THREADS) void k_hits(HitsParams P) {
  __shared__ u32 cnt[16384];
  const BlockDesc bd = P.blocks[blockIdx.y];
  const u64 L = bd.inend - bd.ws;
  const u64 e0 = (u64)blockIdx.x * RK_CH;
  if (e0 >= L) return;
  const u64 e1 = (e0 + RK_CH < L) ? e0 + RK_CH : L;
  const u32 tid = threadIdx.x;
  const u8* base = P.in + bd.ws;
  const u16* same = P.same16 + bd.reg_off;
  for (u32 i = tid; i < 16384; i += RK_THREADS) cnt[i] = 0;
  __syncthreads();
  for (u64 s = e0; s < e1; s += RK_THREADS * 4u) {
    u32 by[4], sm[4];
#pragma unroll
    for (u32 u = 0; u < 4; ++u) {
      const u64 p = s + RK_THREADS * u + tid;
      by[u] = p < e1 ? rk_load_u32(base + p) : 0u;
      sm[u] = p < e1 ? (u32)same[p] : 0u;
    }
#pragma unroll
    for (u32 u = 0; u < 4; ++u) {
      const u64 p = s + RK_THREADS * u + tid;
      if (p < e1) {
        const u32 key = rk_val2(by[u], p, L, sm[u]);
        atomicAdd(&cnt[key >> 1], 1u << (16u * (key & 1u)));
      }
    }
  }
  __syncthreads();
  unsigned long long sum = 0;
  u32 mx2 = 0;
  for (u32 i = tid; i < 16384; i += RK_THREADS) {
    const u32 c = cnt[i];
    const unsigned long long a = c & 0xffffu, b = c >> 16;
    sum += a * a + b * b;
    mx2 = mx2 > (u32)a ? mx2 : (u32)a;
    mx2 = mx2 > (u32)b ? mx2 : (u32)b;
  }
  for (int off = 32; off > 0; off >>= 1) sum += __shfl_down(sum, off, 64);
  if ((tid & 63u) == 0) atomicAdd(&P.energy[blockIdx.y], sum);
  if (P.cmax == nullptr) return;
  // The same count for the 3-byte hash (the first chain's classes), and the largest class of either kind: a position's
  // walk visits at most the members of its two classes in its own chunk and the one before (a superset of its window),
  // so where the four maxima stay below 8192 the cap of lz77.c:527-530 cannot bind and nobody needs ranks (k_rank2).
  __shared__ u32 s_mx[2];
  __syncthreads();
  if (tid < 2) s_mx[tid] = 0;
  for (u32 i = tid; i < 16384; i += RK_THREADS) cnt[i] = 0;
  __syncthreads();
  for (u64 s = e0; s < e1; s += RK_THREADS * 4u) {
    u32 by[4];
#pragma unroll
    for (u32 u = 0; u < 4; ++u) {
      const u64 p = s + RK_THREADS * u + tid;
      by[u] = p < e1 ? rk_load_u32(base + p) : 0u;
    }
#pragma unroll
    for (u32 u = 0; u < 4; ++u) {
      const u64 p = s + RK_THREADS * u + tid;
      if (p < e1) {
        const u32 key = rk_val2(by[u], p, L, 3u);
        atomicAdd(&cnt[key >> 1], 1u << (16u * (key & 1u)));
      }
    }
  }
  __syncthreads();
  u32 mx1 = 0;
  for (u32 i = tid; i < 16384; i += RK_THREADS) {
    const u32 c = cnt[i];
    mx1 = mx1 > (c & 0xffffu) ? mx1 : (c & 0xffffu);
    mx1 = mx1 > (c >> 16) ? mx1 : (c >> 16);
  }
  mx1 = rdlane_u32(wave_scan_max(mx1), 63);
  mx2 = rdlane_u32(wave_scan_max(mx2), 63);
  if ((tid & 63u) == 0) { atomicMax(&s_mx[0], mx1); atomicMax(&s_mx[1], mx2); }
  __syncthreads();
  if (tid == 0) P.cmax[(u64)blockIdx.y * gridDim.x + blockIdx.x] = (s_mx[0] > 0xffffu ? 0xffffu : s_mx[0]) | ((s_mx[1] > 0xffffu ? 0xffffu : s_mx[1]) << 16);
}

// ----------------------------------------------------------------------------
// k_rank2: the position's record for k_match5 — k_chain's links, the level links, and the position's RANK within each
// of its two classes — positions of its 3-byte hash (the first chain, hash.c:110-114), positions of its val2 (the
// second chain, hash.c:129-135) — per 32768-position chunk of the region:
//   rank = number of positions before it in its own chunk with its hash value        (in the record)
//   tot  = number of positions of the chunk BEFORE its own with that hash value       (tot12[]: asked for pos only)
// so that the number of chain entries from a member c of pos's class down to a member q (both within 32767 of pos:
// in pos's chunk or the one before) is g(c) - g(q), g(x) = rank[x] + (x in pos's chunk ? tot[pos] : 0).
// One workgroup per (chunk, block); 32768 16-bit counters, two to a word, in LDS (64 KB): a class has at most 32768
// members in a chunk, so a half never carries into its neighbour.  Per class three passes: all waves count the chunk
// before; all waves read those counts for the chunk's positions (tot); ONE wave counts the chunk itself in position
// order, a step of 64 at a time — the count before the step is the rank of the step's first member of a class,
// members of one class inside a step (the chain's link says so) are ordered with ballots.  Then all waves put the
// records together.
// ----------------------------------------------------------------------------
#define RK_U 4u

struct RankParams {
  const u8* in;
  const BlockDesc* blocks;
  const ushort4* links;
  const u16* same16;
  const u16* lev;
  u64 total_l;
  u16* tot;           // scratch, two per region position (class-major: [cls * total_l + position])
  u16* rank;          // scratch, likewise
  uint4* xrec;        // out: two per region position
  u32* tot12;         // out: one per region position
  const unsigned long long* energy;
  u64 thr;
  const u32* cmax;    // k_hits: [blocks][cmax_stride] the chunks' largest classes, or null (ranks everywhere)
  u32 cmax_stride;
};

// val2 of 8 consecutive positions p .. p + 7 from the 16 bytes at p and their 8 run lengths
__device__ __forceinline__ void rk_keys8(uint4 by, uint4 sm, u64 p, u64 L, u32 (&key)[8]) {
  const u32 b[4] = {by.x, by.y, by.z, by.w};
  const u32 m[4] = {sm.x, sm.y, sm.z, sm.w};
#pragma unroll
  for (u32 i = 0; i < 8; ++i) {
    const u32 w0 = b[i >> 2], w1 = b[(i >> 2) + 1];           // (i + 2 <= 9: dword (i >> 2) + 1 <= 2)
    const u32 sh = 8u * (i & 3u);
    const u32 three = sh ? (w0 >> sh) | (w1 << (32u - sh)) : w0;
    const u32 same = (i & 1u) ? m[i >> 1] >> 16 : m[i >> 1] & 0xffffu;
    key[i] = rk_val2(three, p + i, L, same);
  }
}
__device__ __forceinline__ uint4 rk_load16u(const u8* p) {
  uint4 x;
  __builtin_memcpy(&x, p, 16);
  return x;
}

__global__ __launch_bounds__(RK_THREADS) void k_rank2(RankParams P) {
  __shared__ u32 cnt[16384];
  const BlockDesc bd = P.blocks[blockIdx.y];
  const u64 L = bd.inend - bd.ws;
  const u64 e0 = (u64)blockIdx.x * RK_CH;
  if (e0 >= L) return;
  if (!m5_block_on(P.energy, P.thr, blockIdx.y, L)) return;
  const u64 e1 = (e0 + RK_CH < L) ? e0 + RK_CH : L;
  const u32 tid = threadIdx.x;
  const u32 lane = tid & 63u;
  const u64 lt_mask = (1ull << lane) - 1;
  const u8* base = P.in + bd.ws;
  const ushort4* lk = P.links + bd.reg_off;
  const u16* same = P.same16 + bd.reg_off;      // (reg_off is a multiple of 8 entries: 16-byte loads of 8 are aligned)
  const u16* lev = P.lev + bd.reg_off;
  uint4* xr = P.xrec + bd.reg_off * 2;
  u32* t12 = P.tot12 + bd.reg_off;

  // Can the 8192-hit cap (lz77.c:527-530) bind for a position of this chunk?  Its walk visits at most the members of its
  // two classes in this chunk and the one before: if the largest classes of the two chunks (k_hits) add up to less, no —
  // k_match5 then neither counts nor tests (the record says so: bit 31 of word 2), and ranks are needed only where a
  // capped chunk FOLLOWS (its positions' candidates reach back into this one).  Text and markup: no chunk is capped, and
  // the three rank passes — one of them a single wave walking the chunk in order — were 8 of the 13 ms of the record build.
  bool cap_own = true, need = true;
  if (P.cmax != nullptr) {
    const u32* cm = P.cmax + (u64)blockIdx.y * P.cmax_stride;
    auto capped = [&](u32 ch) -> bool {
      const u32 a = cm[ch], b = ch ? cm[ch - 1] : 0u;
      return (a & 0xffffu) + (b & 0xffffu) + (a >> 16) + (b >> 16) >= ZMX_MAX_CHAIN_HITS;
    };
    cap_own = capped(blockIdx.x);
    need = cap_own || (e1 < L && capped(blockIdx.x + 1u));
  }
  // the two classes in turn: cls 0 = the 3-byte hash (val: the first chain), cls 1 = val2 (the second chain)
  for (u32 cls = 0; cls < 2 && need; ++cls) {
    u16* tt = P.tot + cls * P.total_l + bd.reg_off;
    u16* rk = P.rank + cls * P.total_l + bd.reg_off;
    __syncthreads();
    for (u32 i = tid; i < 16384; i += RK_THREADS) cnt[i] = 0;
    __syncthreads();
    // 1. the class sizes of the chunk before (a whole chunk: 8 positions a thread and turn)
    if (e0 > 0) {
      for (u64 s = e0 - RK_CH + 8u * tid; s < e0; s += 8u * RK_THREADS) {
        u32 key[8];
        rk_keys8(rk_load16u(base + s), cls ? *reinterpret_cast<const uint4*>(same + s) : make_uint4(0x00030003u, 0x00030003u, 0x00030003u, 0x00030003u), s, L, key);
#pragma unroll
        for (u32 i = 0; i < 8; ++i) atomicAdd(&cnt[key[i] >> 1], 1u << (16u * (key[i] & 1u)));
      }
    }
    __syncthreads();
    // 2. the class sizes for the chunk's positions (the arrays are padded to a multiple of 8 entries per block: whole
    //    groups), and what pass 3 needs of a position in ONE 16-bit word: its key, and (bit 15) whether the previous member
    //    of its class lies inside its step of 64 — the chain's link says so
    for (u64 s = e0 + 8u * tid; s < e1; s += 8u * RK_THREADS) {
      u32 key[8];
      rk_keys8(rk_load16u(base + s), cls ? *reinterpret_cast<const uint4*>(same + s) : make_uint4(0x00030003u, 0x00030003u, 0x00030003u, 0x00030003u), s, L, key);
      uint4 l4[4];
#pragma unroll
      for (u32 i = 0; i < 4; ++i) l4[i] = reinterpret_cast<const uint4*>(lk + s)[i];
      u32 t[8], kd[8];
#pragma unroll
      for (u32 i = 0; i < 8; ++i) {
        t[i] = (cnt[key[i] >> 1] >> (16u * (key[i] & 1u))) & 0xffffu;
        const u32 l01 = (i & 1u) ? l4[i >> 1].z : l4[i >> 1].x;                  // prev1 | prev2 << 16 of position s + i
        const u32 d = cls ? l01 >> 16 : l01 & 0xffffu;
        const u32 in_step = (u32)(s + i - e0) & 63u;
        kd[i] = key[i] | ((d != 0 && d <= in_step) ? 0x8000u : 0u);
      }
      *reinterpret_cast<uint4*>(tt + s) = make_uint4(t[0] | (t[1] << 16), t[2] | (t[3] << 16), t[4] | (t[5] << 16), t[6] | (t[7] << 16));
      *reinterpret_cast<uint4*>(rk + s) = make_uint4(kd[0] | (kd[1] << 16), kd[2] | (kd[3] << 16), kd[4] | (kd[5] << 16), kd[6] | (kd[7] << 16));
    }
    __syncthreads();
    for (u32 i = tid; i < 16384; i += RK_THREADS) cnt[i] = 0;
    __syncthreads();
    // 3. ranks within the chunk: one wave, in position order, eight steps' words in flight
    if (tid < 64) {
      for (u64 s = e0; s < e1; s += 64u * 8u) {
        u32 kd[8];
#pragma unroll
        for (u32 u = 0; u < 8; ++u) {
          const u64 p = s + 64u * u + lane;
          kd[u] = p < e1 ? (u32)rk[p] : 0xffffffffu;
        }
#pragma unroll
        for (u32 u = 0; u < 8; ++u) {
          const u64 p = s + 64u * u + lane;
          const bool act = kd[u] != 0xffffffffu;
          const u32 key = kd[u] & 0x7fffu;
          u32 rank = act ? (cnt[key >> 1] >> (16u * (key & 1u))) & 0xffffu : 0u;   // the state before this step
          wave_lds_sync();
          if (act) atomicAdd(&cnt[key >> 1], 1u << (16u * (key & 1u)));
          if (__any(act && (kd[u] & 0x8000u))) {     // members of one class inside the step: in lane order
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
          if (act) rk[p] = (u16)rank;
        }
      }
    }
  }
  __syncthreads();
  // 4. the records: 8 positions a thread and turn, every array in 16-byte pieces
  const u16* tt1 = P.tot + bd.reg_off;
  const u16* tt2 = P.tot + P.total_l + bd.reg_off;
  const u16* rk1 = P.rank + bd.reg_off;
  const u16* rk2 = P.rank + P.total_l + bd.reg_off;
  for (u64 s = e0 + 8u * tid; s < e1; s += 8u * RK_THREADS) {
    uint4 l4[4], lv[LV_N];
#pragma unroll
    for (u32 i = 0; i < 4; ++i) l4[i] = reinterpret_cast<const uint4*>(lk + s)[i];
    const uint4 z4 = make_uint4(0, 0, 0, 0);
    const uint4 ta = need ? *reinterpret_cast<const uint4*>(tt1 + s) : z4, tb = need ? *reinterpret_cast<const uint4*>(tt2 + s) : z4;
    const uint4 ra = need ? *reinterpret_cast<const uint4*>(rk1 + s) : z4, rb = need ? *reinterpret_cast<const uint4*>(rk2 + s) : z4;
#pragma unroll
    for (u32 j = 0; j < LV_N; ++j) lv[j] = *reinterpret_cast<const uint4*>(lev + (u64)j * P.total_l + s);
    const u32 taw[4] = {ta.x, ta.y, ta.z, ta.w}, tbw[4] = {tb.x, tb.y, tb.z, tb.w};
    const u32 raw[4] = {ra.x, ra.y, ra.z, ra.w}, rbw[4] = {rb.x, rb.y, rb.z, rb.w};
#pragma unroll
    for (u32 i = 0; i < 8; ++i) {
      if (s + i >= e1) break;
      const u32 lx = (i & 1u) ? l4[i >> 1].z : l4[i >> 1].x, ly = (i & 1u) ? l4[i >> 1].w : l4[i >> 1].y;   // links[s + i] as two dwords
      const u32 sh = 16u * (i & 1u);
      const u32 tot1 = (taw[i >> 1] >> sh) & 0xffffu, tot2 = (tbw[i >> 1] >> sh) & 0xffffu;
      const u32 r1 = (raw[i >> 1] >> sh) & 0xffffu, r2 = (rbw[i >> 1] >> sh) & 0xffffu;
      u32 l[LV_N];
#pragma unroll
      for (u32 j = 0; j < LV_N; ++j) {
        const u32 wsel = (i >> 1) == 0 ? lv[j].x : (i >> 1) == 1 ? lv[j].y : (i >> 1) == 2 ? lv[j].z : lv[j].w;
        l[j] = (wsel >> sh) & 0xffffu;
      }
      uint4 a, b;
      a.x = lx;                                        // prev1 | prev2 << 16
      a.y = (ly & 0xffffu) | (r1 << 16);               // same | rank within its 3-byte-hash class (in its chunk)
      a.z = r2 | (cap_own ? 0u : 0x80000000u);         // rank within its val2 class; bit 31: the cap cannot bind for this position
      a.w = l[0] | (l[1] << 16);
      b.x = l[2] | (l[3] << 16);
      b.y = l[4] | (l[5] << 16);
      b.z = l[6] | (l[7] << 16);
      b.w = l[8] | (l[9] << 16);
      xr[2 * (s + i)] = a;
      xr[2 * (s + i) + 1] = b;
      t12[s + i] = tot1 | (tot2 << 16);
    }
  }
}

// ----------------------------------------------------------------------------
// k_match5.  A wave owns a half or a quarter of a 2048-position tile at a time (sub_shift: 1024 positions where there are
// many tiles, 512 where few) and its lanes take the piece's positions one after the other; nothing is staged in LDS: a walk touches ~10 entries, the 32 KiB window of
// bytes k_match2 stages per tile would be read ~40 times per position staged, and with it goes the tile's barrier —
// k_match2's lanes wait for the tile's longest walk, here a lane waits only at the end of its wave's piece.  Per entry the lane has in flight together: the entry's record (16 + 8 bytes), the 4 bytes the filter
// tests and the entry's first 16 bytes; the position's own first 16 bytes stay in registers, so a common prefix
// of up to 15 bytes — most of them — is decided in the iteration the entry arrives in.
// ----------------------------------------------------------------------------
struct Match5Params {
  MatchParams m;
  const uint4* xrec;     // k_rank2: two per region position
  const u32* tot12;      // k_rank2: the sizes of a position's two classes in the chunk before its own
  const unsigned long long* energy;   // k_hits (null: every block)
  u64 thr;
  u32 sub_shift;         // a wave takes 2048 >> sub_shift positions at a time (0, 1 or 2)
  unsigned long long* wave_stats;   // optional, two per wave of the grid: entries in flight summed over lanes and iterations, iterations
};

#define M5_THREADS 256
#define M5_BATCH 8u     // lanes that wait for a record write / a new position before the wave serves them
#define M5_IDLE 0u
#define M5_WALK 1u      // the loads of the entry at distance xd are in flight
#define M5_CMP 2u       // 16 more bytes of both sides are in flight
#define M5_PEND 3u
#define M5_DONE 4u

// level link number j (0 .. 3) of the four 16-bit links in v
__device__ __forceinline__ u32 m5_pick(uint2 v, u32 j) {
  const u32 w = j & 2u ? v.y : v.x;
  return j & 1u ? w >> 16 : w & 0xffffu;
}
__device__ __forceinline__ uint4 m5_load16(const u8* p) {
  uint4 x;
  __builtin_memcpy(&x, p, 16);
  return x;
}
__device__ __forceinline__ u32 m5_load4(const u8* p) {
  u32 x;
  __builtin_memcpy(&x, p, 4);
  return x;
}
// number of equal leading bytes of two 16-byte strings
__device__ __forceinline__ u32 m5_lcp16(uint4 a, uint4 b) {
  const u64 lo = ((u64)(a.y ^ b.y) << 32) | (a.x ^ b.x), hi = ((u64)(a.w ^ b.w) << 32) | (a.z ^ b.z);
  return lo ? (u32)(__ffsll((unsigned long long)lo) - 1) >> 3 : hi ? 8u + ((u32)(__ffsll((unsigned long long)hi) - 1) >> 3) : 16u;
}

__global__ __launch_bounds__(M5_THREADS, 4) void k_match5(Match5Params Q) {
  const MatchParams& P = Q.m;
  __shared__ u32 s_cp[8 * M5_THREADS];       // the first 8 change points of every lane's position (len | dist << 16), slot-major

  const u32 tid = threadIdx.x;
  const u32 lane = tid & 63u;
  const u64 lt_mask = (1ull << lane) - 1;
  const u32 xcd = blockIdx.x & 7;
  u32* my_scratch = P.scratch + ((u64)blockIdx.x * M5_THREADS + tid) * SCRATCH_CPS;
  unsigned long long sum_lane_iter = 0, sum_iter = 0;     // what this wave's walks cost (wave_stats)

  for (;;) {
    // a wave takes a piece of a 2048-position tile at a time (its own cursors: k_match2 may run beside it): a whole
    // tile of two-symbol data is seconds of one wave's dependent loads, and the kernel ends with its last wave
    u32 tk = 0;
    if (lane == 0) tk = atomicAdd(&P.counters[24 + xcd], 1u);
    tk = (u32)__builtin_amdgcn_readfirstlane((int)tk);
    const u32 sub = 1u << Q.sub_shift, grp = sub * M_XCD_GROUP;          // pieces per tile, per group of tiles of one XCD
    const u32 q_idx = ((tk / grp) * 8u + xcd) * grp + (tk % grp);
    const u32 t_idx = q_idx >> Q.sub_shift;
    if (t_idx >= P.total_tiles) break;
    const u32 tile = P.tile_list ? P.tile_list[t_idx] : t_idx;
    const u32 unit = MT >> Q.sub_shift;

    u32 lo = 0, hi = P.nb;
    while (hi - lo > 1) {
      const u32 mid = (lo + hi) >> 1;
      if (P.tile_off[mid] <= tile) lo = mid; else hi = mid;
    }
    const BlockDesc bd = P.blocks[lo];
    if (!m5_block_on(Q.energy, Q.thr, lo, bd.inend - bd.ws)) continue;     // k_match2's block
    const u64 p0 = bd.instart + (u64)(tile - P.tile_off[lo]) * MT + (u64)(q_idx & (sub - 1u)) * unit;
    if (p0 >= bd.inend) continue;
    const u64 p1 = (p0 + unit < bd.inend) ? p0 + unit : bd.inend;
    const u32 ntile = (u32)(p1 - p0);

    // scalar bases, 32-bit lane offsets: region position li -> its record, its bytes
    const u8* xr;
    const u8* inr;
    {
      const u64 a = reinterpret_cast<u64>(Q.xrec + bd.reg_off * 2);
      const u32 alo = (u32)__builtin_amdgcn_readfirstlane((int)(u32)a), ahi = (u32)__builtin_amdgcn_readfirstlane((int)(u32)(a >> 32));
      xr = reinterpret_cast<const u8*>(((u64)ahi << 32) | alo);
      const u64 b = reinterpret_cast<u64>(P.in + bd.ws);
      const u32 blo = (u32)__builtin_amdgcn_readfirstlane((int)(u32)b), bhi = (u32)__builtin_amdgcn_readfirstlane((int)(u32)(b >> 32));
      inr = reinterpret_cast<const u8*>(((u64)bhi << 32) | blo);
    }
    const u32 li0 = (u32)(p0 - bd.ws);
    const u32 rem0 = (u32)((bd.inend - p0 < 70000) ? bd.inend - p0 : 70000);
    u32* const rec0 = P.recs + (bd.pos_off + (p0 - bd.instart)) * 8;
    const u32* const tot12 = Q.tot12 + bd.reg_off;
    u32 qnext = 0;                 // next position of the tile nobody has taken (wave-uniform)

    // ---- per-lane walk state
    u32 st = M5_IDLE;
    u32 li = 0, ti = 0;            // region index of the position, its index in the tile
    u32 limit = 0, bestlen = 0, bestdist = 0, ncp = 0, same_pos = 0, cur = 0, size_rem = 0;
    u32 byte0 = 0, pbyte = 0, foff = 0, fmask = 0;
    uint4 P0 = make_uint4(0, 0, 0, 0);  // the position's first 16 bytes
    u32 vpos = 0;                  // its 3-byte hash (hash.c:96-98)
    u32 totpos = 0;                // the sizes of its two classes in the chunk before its own (k_rank2): tot1 | tot2 << 16
    u32 chain = 1, idx = 0;        // idx: candidates the reference has visited so far (lz77.c:527-530 stops at 8192)
    u32 curd = 0;                  // distance of the last visited candidate (0: none yet)
    u32 cprev = 0;                 // its prev1 | prev2 << 16 (raw steps follow them)
    u32 gcur1 = 0, gcur2 = 0;      // its g value on the first / second chain
    int lev_k = -1;                // level the walk follows, -1 = the reference's own chain
    u32 eqd = 0, nlink = 0;        // level walk: distance of the last entry and its link
    bool need_link = false;        // the link of (lev_k, eqd) has to be fetched first
    u32 scd = 0, scp2 = 0;         // the search for the switch point (first chain): a member of pos's second chain and its prev2
    u32 sc_st = 0;                 // 0 searching, 1 scd is the nearest member of both of pos's classes below the last visited, 2 none
    u32 fk = 0;                    // what is being fetched: 0 the reference's next hit, 1 a level entry, 2 a level link,
                                   // 3 the next member of pos's second chain (switch-point search), 4 the switch point, to visit it
    u32 xd = 0;                    // its distance
    uint4 A = make_uint4(0, 0, 0, 0);   // its record: prev1 | prev2, same | rank1, rank2, (lv0 | lv1)
    uint2 V = make_uint2(0, 0);    // four of its level links, from level vb on
    u32 vb = 0;
    u32 F = 0;                     // its bytes foff .. foff + 3
    uint4 C0 = make_uint4(0, 0, 0, 0);  // its first 16 bytes; in M5_CMP: 16 bytes of the entry and (PA) of the position from `cur` on
    uint4 PA = make_uint4(0, 0, 0, 0);
    u32 plv0 = 0, plv1 = 0, plv2 = 0, plv3 = 0, plv4 = 0;   // the position's own level links (lv0|lv1, lv2|lv3, ...)
    u32 n_iter = 0;
    u32 n_lane_iter = 0;           // lanes with an entry in flight, summed over the piece's iterations

    auto fetch = [&](u32 d, u32 kind) {
      xd = d;
      fk = kind;
      if (kind == 1u || kind == 2u) {
        vb = lev_k > 6 ? 6u : (u32)lev_k;
      } else {
        const u32 n = (bestlen < 3 ? 3u : bestlen) + 1u;
        const int g = lv_level_for(n > 32u ? 32u : n);     // the level an improvement here most likely asks for
        vb = g > 6 ? 6u : (u32)g;
      }
      const u32 e = li - d;
      const u8* r = xr + (u64)e * 32u;
      A = *reinterpret_cast<const uint4*>(r);
      __builtin_memcpy(&V, r + 2u * (XR_LV0 + vb), 8);
      F = m5_load4(inr + e + foff);
      C0 = m5_load16(inr + e);
    };
    // the entry the walk touches next: false = the walk is over (end of the chain, window, lz77.c:464, :521-523)
    auto next_entry = [&]() -> bool {
      if (chain == 1 && bestlen >= 3) {
        // The FIRST chain with level links.  The reference walks every position of pos's 3-byte hash until one of pos's
        // val2 class comes by at a time when bestlength >= same (lz77.c:509-519).  Of those only two kinds matter:
        // candidates that beat bestlength — entries of the level chain, while bestlength <= same; beyond that a longer
        // match has exactly pos's run length, i.e. is of pos's val2 class, and the nearest member of that class ends
        // this chain anyway — and that nearest member of both classes, the switch point: the next position on pos's
        // SECOND chain (below the last visited candidate) whose 3-byte hash is pos's.  Everything in between is
        // counted from the ranks.
        const bool can_switch = bestlen >= same_pos;
        if (can_switch && sc_st == 0) {
          if (scp2 == 0 || scd + scp2 >= ZMX_WINDOW) sc_st = 2;
          else { fetch(scd + scp2, 3u); return true; }
        }
        const u32 swd = can_switch && sc_st == 1 ? scd : 0xffffffffu;
        if (bestlen <= same_pos) {
          if (need_link) { fetch(eqd, 2u); return true; }
          if (nlink != 0 && eqd + nlink < ZMX_WINDOW && eqd + nlink <= swd) { fetch(eqd + nlink, 1u); return true; }
        }
        if (swd != 0xffffffffu) { fetch(swd, 4u); return true; }
        return false;
      }
      if (lev_k < 0 || chain == 1) {
        const u32 step = chain == 1 ? (cprev & 0xffffu) : (cprev >> 16);
        if (step == 0 || curd + step >= ZMX_WINDOW) return false;
        fetch(curd + step, 0u);
        return true;
      }
      if (need_link) { fetch(eqd, 2u); return true; }
      if (nlink == 0 || eqd + nlink >= ZMX_WINDOW) return false;
      fetch(eqd + nlink, 1u);
      return true;
    };
    auto pos_link = [&](u32 k) -> u32 {
      const u32 w = k < 2 ? plv0 : k < 4 ? plv1 : k < 6 ? plv2 : k < 8 ? plv3 : plv4;
      return k & 1u ? w >> 16 : w & 0xffffu;
    };
    // level for the walk after bestlength or the chain changed; `at_x`: bestlength was just set by the entry (xd, V)
    auto choose_level = [&](bool at_x) {
      int k = -1;
      if (bestlen >= 3) {
        k = lv_level_for(bestlen + 1 > 32 ? 32u : bestlen + 1);
        if (chain == 2 && same_pos + 2 > kLevelK[k]) k = -1;       // the second chain is the more selective list
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
      } else if (k >= 0 && at_x && (u32)k >= vb && (u32)k < vb + 4u) {
        eqd = xd;
        nlink = m5_pick(V, (u32)k - vb);
      } else if (k >= 0 && at_x) {
        eqd = xd;                              // (the entry was not fetched as a level entry: its link of level k is not in V)
        need_link = true;
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
          u32* const rec = rec0 + (u64)ti * 8;
          reinterpret_cast<uint4*>(rec)[0] = r0;
          reinterpret_cast<uint4*>(rec)[1] = r1;
        }
        // the lanes without a position take the tile's next ones, in lane order
        const u64 m_idle = __ballot(st == M5_IDLE);
        if (st == M5_IDLE) {
          const u32 i = qnext + (u32)__popcll(m_idle & lt_mask);
          if (i >= ntile) {
            st = M5_DONE;
          } else {
            ti = i;
            li = li0 + i;
            size_rem = rem0 - i;
            u32* const rec = rec0 + (u64)i * 8;
            const uint4* rp = reinterpret_cast<const uint4*>(xr + (u64)li * 32u);
            const uint4 Ap = rp[0], Bp = rp[1];
            P0 = m5_load16(inr + li);
            same_pos = Ap.y & 0xffffu;
            byte0 = P0.x & 255u;
            ncp = 0;
            bestlen = 1; bestdist = 0; chain = 1; curd = 0; lev_k = -1; need_link = false;
            // (where the cap cannot bind — k_rank2's verdict, bit 31 — the count starts so far below zero that no sum of
            //  hops, garbage there, brings it up to 8192: the tests below are signed)
            idx = (Ap.z >> 31) ? 0xC0000000u : 0u;
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
                pbyte = P0.x; foff = 0; fmask = 0xffffu;   // bestlength 1: bytes 0 and 1
                vpos = (((P0.x & 255u) << 10) ^ (((P0.x >> 8) & 255u) << 5) ^ ((P0.x >> 16) & 255u)) & 32767u;   // (size_rem >= 3: real bytes)
                totpos = tot12[li];
                gcur1 = (Ap.y >> 16) + (totpos & 0xffffu);
                scd = 0; scp2 = Ap.x >> 16; sc_st = 0;
                next_entry();                        // (prev1 < 32768: always an entry)
                st = M5_WALK;
              }
            }
          }
        }
        qnext += (u32)__popcll(m_idle);
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
      bool ended = false;          // the common prefix with the entry is known: cur
      // A wave-loop that does not end would hang the device: far beyond what a tile can take (8 positions a lane,
      // at most 8192 hits each), the tile is given up, the state of a running lane goes to counters[32..39] and
      // the host fails the build.
      n_lane_iter += (u32)__popcll(m_run);
      if (++n_iter > (1u << 20)) {
        atomicOr(&P.counters[1], 2u);
        {
          const u64 mr = __ballot(st == M5_WALK || st == M5_CMP);
          const u32 l0 = mr ? (u32)__ffsll((unsigned long long)mr) - 1u : 0u;
          if (lane == l0 && atomicCAS(&P.counters[32], 0u, 0x80000000u | st | (chain << 4) | ((u32)(lev_k + 1) << 8) | ((need_link ? 1u : 0u) << 12) | (fk << 13) | (vb << 16) | (sc_st << 20)) == 0u) {
            P.counters[33] = xd; P.counters[34] = curd; P.counters[35] = bestlen | (limit << 16); P.counters[36] = nlink | (eqd << 16);
            P.counters[37] = li; P.counters[38] = idx | (same_pos << 16); P.counters[39] = scd | (bestdist << 16);
          }
        }
        break;
      }
      // is the entry the switch point of the first chain (visited whatever its bytes)?  and is it a level entry, which
      // only counts when it beats bestlength?
      const bool same_chunk = (li - xd) >> 15 == li >> 15;
      if (walk) {
        bool cand = true;
        if (fk == 2u) {                                  // the link of the level walk's entry point
          need_link = false;
          nlink = m5_pick(V, (u32)lev_k - vb);
          cand = false; moved = true;
        } else if (fk == 1u) {
          eqd = xd;
          nlink = m5_pick(V, (u32)lev_k - vb);
          if (xd <= curd) { cand = false; moved = true; }     // at or above the last visited candidate (entered at pos)
        } else if (fk == 3u) {                           // the next member of pos's second chain: of pos's 3-byte hash too?
          scd = xd;
          scp2 = A.x >> 16;
          const u32 v = (((C0.x & 255u) << 10) ^ (((C0.x >> 8) & 255u) << 5) ^ ((C0.x >> 16) & 255u)) & 32767u;
          if (xd > curd && v == vpos) {
            sc_st = 1;
            if (bestlen <= same_pos) { cand = false; moved = true; }   // level entries may come before it: not its turn yet
          } else {
            cand = false; moved = true;
          }
        }
        if (cand) {
          const bool is_sw = chain == 1 && bestlen >= 3 && sc_st == 1 && xd == scd;
          if (((F ^ pbyte) & fmask) != 0) {          // cannot be longer than bestlength (lz77.c:478-479, 494)
            if (fk == 0u || is_sw) visited = true; else moved = true;
          } else {
            const u32 l16 = m5_lcp16(P0, C0);
            if (l16 < 16u) {
              cur = l16 < limit ? l16 : limit;
              ended = true;
            } else {
              cur = 16;
              if (same_pos > 2 && (C0.x & 255u) == byte0) {     // lz77.c:481-490
                const u32 lz = A.y & 0xffffu;
                const u32 sm = same_pos < lz ? same_pos : lz;
                if (sm > cur) cur = sm;
              }
              if (cur >= limit) {
                cur = limit;
                ended = true;
              } else {
                PA = m5_load16(inr + li + cur);
                C0 = m5_load16(inr + li - xd + cur);
                st = M5_CMP;
              }
            }
          }
        }
      } else if (st == M5_CMP) {
        const u32 rem = limit - cur;
        u32 m = m5_lcp16(PA, C0);
        if (m > rem) m = rem;
        cur += m;
        if (m < 16u || cur >= limit) {
          ended = true;
          st = M5_WALK;
        } else {
          PA = m5_load16(inr + li + cur);
          C0 = m5_load16(inr + li - xd + cur);
        }
      }
      // the candidate's g values (k_rank2): positions of pos's chunk count from the chunk before's totals
      const u32 gx1 = ((A.y >> 16) + (same_chunk ? totpos & 0xffffu : 0u)) & 0xffffu;
      const u32 gx2 = ((A.z & 0xffffu) + (same_chunk ? totpos >> 16 : 0u)) & 0xffffu;
      if (ended) {
        const bool is_sw = chain == 1 && bestlen >= 3 && sc_st == 1 && xd == scd;
        if (cur > bestlen) {
          // a longer match: is it a candidate the reference visits, and which one?
          bool ok = true;
          u32 hops = 1;
          if (fk != 0u) {
            if (chain == 2) {
              ok = (((A.y & 0xffffu) - 3u) & 255u) == ((same_pos - 3u) & 255u);   // of the position's val2 class
              hops = (gcur2 - gx2) & 0xffffu;
            } else {
              hops = (gcur1 - gx1) & 0xffffu;      // (it shares >= 4 bytes: of pos's 3-byte hash)
            }
          }
          if (!ok) {
            moved = true;                        // not on the second chain: never visited
          } else if ((int)(idx + hops) > (int)ZMX_MAX_CHAIN_HITS) {
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
            pbyte = m5_load4(inr + li + foff);   // (arrives with the next entry)
            curd = xd;
            cprev = A.x;
            gcur1 = gx1;
            gcur2 = gx2;
            if (cur >= limit) {
              fin = true;                        // lz77.c:500-502
            } else {
              // lz77.c:509-519 (on chain 1 the 3-byte hashes are equal: val2 equality is equality of (same - 3) & 255)
              if (chain == 1 && bestlen >= same_pos && (((A.y & 0xffffu) - 3u) & 255u) == ((same_pos - 3u) & 255u)) chain = 2;
              choose_level(true);
              moved = true;
            }
          }
        } else if (fk == 0u || is_sw) {
          visited = true;                        // compared, not longer
        } else {
          moved = true;
        }
      }
      if (visited) {
        // a candidate the reference visits that does not beat bestlength — its own chain's next hit, or the switch
        // point: count it (and what lies between), test the switch rule
        const u32 hops = fk == 0u ? 1u : (gcur1 - gx1) & 0xffffu;
        if ((int)(idx + hops) > (int)ZMX_MAX_CHAIN_HITS) {
          fin = true;
        } else {
          idx += hops;
          curd = xd;
          cprev = A.x;
          gcur1 = gx1;
          gcur2 = gx2;
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
    sum_lane_iter += n_lane_iter;
    sum_iter += n_iter;
#ifdef M5_ATOMIC_STATS
    if (Q.wave_stats) {
      atomicAdd(&Q.wave_stats[0], (unsigned long long)n_lane_iter);
      atomicAdd(&Q.wave_stats[1], (unsigned long long)n_iter);
    }
#endif
  }
  // (plain stores, one slot per wave: two atomic adds per piece at this point made the kernel not come back — twice,
  //  never understood)
  if (Q.wave_stats && lane == 0) {
    unsigned long long* w = Q.wave_stats + 2u * ((u64)blockIdx.x * (M5_THREADS / 64) + (tid >> 6));
    w[0] = sum_lane_iter;
    w[1] = sum_iter;
  }
}

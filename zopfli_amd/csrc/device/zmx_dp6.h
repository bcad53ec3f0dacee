// d6_run_job / k_dp6_spec: a RUN TASK of the chain (zmx_dp4.h, zmx_dp5.h) by FOUR waves.  Included only by zmx_hip.hip,
// after zmx_dp5.h (same jobs, snapshots, tables, window kinds, arithmetic: this file restates d5_run_job<.., RUNS = true>).
//
// Why.  On runs of equal bytes a row of the DP is 258 edges wide: five cell registers per position, and a task — the
// stretch between two exact cut points, tens of thousands of positions ("Long runs") — is ONE wave
// alone on its SIMD: ~650 cycles per walked position, of which the chain proper (cell register 0, where the sources of
// a 32-position window all lie) is a fifth.  A squeeze run on such data waits for its longest task.  Here a task is a
// workgroup of four waves, one per SIMD of a CU, and the CELL REGISTERS are dealt over them:
//
//     wave 0: register 0 (cells wbase .. wbase + 63: the chain)     wave 2: register 2
//     wave 1: register 1                                            wave 3: registers 3 and 4 (a row ends at cell 289)
//
// Every wave runs the SAME program — the window loop of d5_run_job, its stretches, shortcuts and snapshots — on the
// registers it owns; what the program's control flow depends on is static (dph, wmeta, the bad-edge bitmap) except for a
// handful of decisions taken from cell values, and those, like the source values themselves, come from wave 0:
//
//   * THE STREAM.  Wave 0 puts every value the others need into a ring in LDS, in program order: the source cell of every
//     walked position (the bit pattern of costs[j], squeeze.c:290), a token per window (its kind; "leave a mid snapshot
//     here"), the decisions of a run stretch (is there room for the whole stretch in the integer table's binade).  A slot
//     is 8 bytes {value, sequence number}, written by one ds_write_b64: a reader spins on the slot until the number is
//     the one it waits for.  The others consume the stream in the same program order, so every cell sees its updates in
//     position order, exactly as in the serial chain: NOTHING about the arithmetic changes, only who holds the cell.
//   * ROTATION.  When the window moves on by 32 cells, the low half of register s + 1 becomes the high half of register
//     s: wave s + 1 leaves its low half (values and sources) in a channel in LDS, wave s picks it up — the one point per
//     window where wave 0 waits for wave 1 (which trails it by about one position).
//   * SHORTCUTS (squeeze.c:251-271) move every live cell 258 cells on: all four waves leave their registers in a shared
//     area, meet at a barrier, and take the shifted cells back.
//   * SNAPSHOTS (entry / mid / exit, zmx_dp4.h) are written by the owners of the cells, the scalars by wave 0.
//
// Flow control: a wave other than 0 can never be ahead of wave 0 (it needs the window's token); wave 0 does not start
// window w before every other wave has finished window w - 3 (it reads their counters); a rotation channel is eight
// windows deep and the stream ring 1024 slots — more than four windows can fill.  No spin is unbounded: a wave that has
// waited 2^22 rounds raises flags[1] bit 3 (the host fails the run), tells the others through LDS and ends.
//
// The text tasks keep k_dp5_spec<.., 4, false> (a wave per task: their rows fit one register); k_dp4_fix's serial re-runs
// keep the one-wave job (with the cooperative job compiled into it, that kernel was a fifth slower on text).
#pragma once

#define D6_NW 4u                   // waves of a cooperative job
#define D6_RING 1024u              // stream slots (a window is at most ~160 words: a token, 32 sources, stretch decisions, sources asked for again)
#define D6_ROTD 8u                 // windows a rotation channel holds
#define D6_SPIN_MAX (1u << 22)
#ifndef D6_OTHER_GROUP
#define D6_OTHER_GROUP 4u           // other rows whose LDS reads are issued together, ahead of their stream operations
#endif
#ifndef D6_RUN_GROUP
#define D6_RUN_GROUP 8u             // run rows likewise
#endif
#define D6_FLAG_STUCK 8u           // flags[1]: a wave of a cooperative job gave up waiting
// what a cooperative job needs in LDS besides the run's tables: one set per TASK (workgroup)
#define D6_LDS_BYTES (D6_RING * 8u + 3u * D6_ROTD * 32u * 8u + 64u)
// (LDS byte addresses, not pointers: the stream, the counters and the channels are read and written through explicit
//  address-space-3 pointers — a volatile access through a generic pointer is compiled as a FLAT instruction, which waits
//  for every outstanding global load and store of the wave, the LDS-DMA prefetches included: measured, the first version
//  of this job was 2.4 x slower than the one-wave job for that reason alone)
struct D6Lds {
  u32 ring;                        // [D6_RING] values, then [D6_RING] sequence numbers (u32 each)
  u32 rot;                         // [3][D6_ROTD][32] 8 bytes each: channel s - 1 carries register s's low half to wave s - 1
  u32 done;                        // [D6_NW] windows wave r has finished (its low half is in the channel)
  u32 abort;                       // [1] a wave gave up: everybody ends
  u32 vmaxw;                       // [1] the largest finite cell value (float bits) the job's waves have seen
  u32 xc;                          // [DP_XN] floats: shortcut exchange, cell values ...
  u32 xl;                          // [DP_XN] u16: ... and lengths
};
typedef __attribute__((address_space(3))) volatile unsigned long long* d6_vu64p;
typedef __attribute__((address_space(3))) volatile u32* d6_vu32p;
typedef __attribute__((address_space(3))) u32* d6_u32p;
typedef __attribute__((address_space(3))) unsigned long long* d6_u64p;
typedef __attribute__((address_space(3))) float* d6_f32p;
typedef __attribute__((address_space(3))) u16* d6_u16p;
__device__ __forceinline__ u32 d6_lds_addr(const void* p) {
  return (u32)(size_t)(__attribute__((address_space(3))) const void*)p;
}
__device__ __forceinline__ D6Lds d6_lds_carve(unsigned char* base, float* xc, u16* xl) {
  D6Lds L;
  L.ring = d6_lds_addr(base);
  L.rot = L.ring + D6_RING * 8u;
  L.done = L.rot + 3u * D6_ROTD * 32u * 8u;
  L.abort = L.done + 32u;
  L.vmaxw = L.done + 36u;
  L.xc = d6_lds_addr(xc);
  L.xl = d6_lds_addr(xl);
  return L;
}
// LDS-only ordering inside the workgroup: the wave's LDS operations are complete (a generic fence would also wait for
// its global loads and stores)
// The stream's own LDS instructions, as asm: a volatile C++ access makes the compiler wait for every LDS operation in
// flight before AND after it; the stream needs neither.  A slot is TWO words in two arrays, the value written first, the
// sequence number second (the LDS queue of a wave is in order), and read in the opposite order — number, then value: a
// reader that sees the number sees the value, whatever the LDS does with two waves' instructions at once (one 8-byte
// slot, written and read whole, was the first version: 5 block runs in 1200 differed).
__device__ __forceinline__ void d6_ds_write32(u32 addr, u32 v) { asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
__device__ __forceinline__ u32 d6_ds_read32(u32 addr) {
  u32 v;
  asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  return v;
}
__device__ __forceinline__ void d6_lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void d6_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// The job.  Called by ALL D6_NW waves of the workgroup with the same arguments (role = the wave's index); s_stage, s_ri,
// s_rk are the calling wave's own areas, everything else is shared.  Every wave executes the same barriers.
// (ROLE is a template parameter: with the wave's index as a run-time value every position went through a dozen taken
//  scalar branches — "am I wave 0", "do I own register 4" — and a lone wave pays ~20 cycles for each: 490 cycles a run row)
template <bool PROF, u32 ROLE>
__device__ __forceinline__ void d6_run_job(const Dp4Params& P, const D4Job& J, u32 b, const BlockDesc& bd,
                                           const double (&s_wtab)[ZMX_WTAB], const D6Lds& L, u16* s_stage,
                                           const uint2 (&s_itab)[ZMX_WTAB], const D5IntTab& IT, const double* s_w1,
                                           const u8* s_sym1, uint2* s_ri, uint2* s_rk) {
  typedef __attribute__((address_space(3))) const u16* lds_u16p;
  const u32 stage_half = (u32)__builtin_amdgcn_readfirstlane((int)((u32)(size_t)(__attribute__((address_space(3))) void*)s_stage >> 1));
  const u32 lane = threadIdx.x & 63;
  const u32 lane2 = lane * 2u;
  constexpr u32 role = ROLE;
  constexpr bool w0 = ROLE == 0;
  constexpr bool has1 = ROLE == D6_NW - 1;         // this wave also owns register 4
  const u32 B = (u32)(bd.inend - bd.instart);
  const uint2* __restrict__ dbase = uniform_ptr(P.dph + bd.pos_off);
  const u32* __restrict__ badpos = uniform_ptr(P.badpos + (bd.pos_off >> 5));
  const u32 bit_off = (u32)(bd.pos_off & 31);
  u16* la = P.la + bd.la_off;
  const u16* __restrict__ rows = uniform_ptr(P.codes + P.code_base[b]);
  auto code_w = [&](u32 code) -> double {
    return *reinterpret_cast<const double*>(reinterpret_cast<const char*>(s_wtab) + code);
  };
  typedef const __attribute__((address_space(4))) u32* cu32p;
  const cu32p winflag = (cu32p)uniform_ptr(P.winflag + P.win_off[b]);
  const cu32p winroff = (cu32p)uniform_ptr(P.winroff + P.win_off[b]);
  const u32* __restrict__ wmeta = uniform_ptr(P.wmeta + (u64)P.win_off[b] * D5_WM);
  const u64 rows_addr = reinterpret_cast<u64>(rows);
  const cu32p badw = (cu32p)badpos;
  const double mincost = P.mincost[b];
  const double symbolcost258 = (double)(0 + 0) + P.cost[(u64)b * 320 + 285] + P.cost[(u64)b * 320 + 288];
  const double kInf = __longlong_as_double(0x7ff0000000000000ll);

  // ---- the job's shared state: zeroed by everybody, then a barrier
  for (u32 i = threadIdx.x; i < D6_RING; i += 64u * D6_NW) ((d6_u64p)L.ring)[i] = 0ull;
  if (threadIdx.x < 16) ((d6_u32p)L.done)[threadIdx.x] = 0u;      // done[0..3], (spare), abort, vmaxw
  __syncthreads();

  // ---- the stream (wave 0 writes, the others read, in program order)
  u32 seq = 1;
  // PROF (ZOPFLI_AMD_PROF): wave 0's cycles by what it was doing — [0] flow control + token, [1] class-1 windows, [2] class-2,
  // [3] generic header, [4] stretches of other rows, [5] run stretches, [6] the general step, [7] shortcuts, [8] waiting for
  // wave 1's low half, [9] the whole job; counts — [10] positions in other stretches, [11] in run stretches, [12] in the
  // general step, [13] shortcuts, [14] windows retired; wave 1: [15] cycles inside get()
  u64 pc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  auto tick = [&]() -> u64 { return PROF ? (u64)__builtin_readcyclecounter() : 0ull; };
  const u64 t_job0 = tick();
  const d6_vu32p v_done = (d6_vu32p)L.done;
  const d6_vu32p v_abort = (d6_vu32p)L.abort;
  auto give_up = [&]() {
    *v_abort = 1u;
    if (lane == 0) atomicOr(&P.flags[1], D6_FLAG_STUCK);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __builtin_amdgcn_endpgm();
  };
  auto put = [&](u32 v) {
    if (lane == 0) {
      const u32 a = L.ring + 4u * (seq & (D6_RING - 1u));
      d6_ds_write32(a, v);
      d6_ds_write32(a + 4u * D6_RING, seq);
    }
    ++seq;
  };
  // A reader takes 64 slots at a time (lane i: slot seq + i) and serves itself from the registers while the sequence
  // numbers fit: one LDS round trip for as many words as wave 0 is ahead, none per word (a round trip per source made
  // the readers slower than the one-wave job's whole row).
  u32 cache_base = 0, cache_lo = 0, cache_hi = 0;     // slots cache_base .. cache_base + 63 as read last (cache_base 0: nothing read)
  auto get = [&]() -> u32 {
    u32 spins = 0;
    u64 t0_ = 0;
    for (;;) {
      const u32 i = seq - cache_base;
      if (i < 64u && rdlane_u32(cache_hi, i) == seq) {
        const u32 v = rdlane_u32(cache_lo, i);
        ++seq;
        if (PROF && spins) pc[15] += (u64)__builtin_readcyclecounter() - t0_;
        return v;
      }
      if (PROF && spins == 0) t0_ = __builtin_readcyclecounter();
      const u32 a = L.ring + 4u * ((seq + lane) & (D6_RING - 1u));
      cache_hi = d6_ds_read32(a + 4u * D6_RING);      // the numbers first, then the values (see d6_ds_write32)
      cache_lo = d6_ds_read32(a);
      cache_base = seq;
      if ((++spins & 63u) == 0 && (*v_abort != 0u || spins > D6_SPIN_MAX)) give_up();
    }
  };
  // a value every wave needs that only wave 0 can compute
  auto share = [&](u32 v) -> u32 {
    if (w0) { put(v); return v; }
    return get();
  };
  // N words at once (the sources of a group of rows): wave 0 collects them in lanes 0 .. N - 1 of `acc` (v_writelane, no
  // exec games, no LDS) and writes them with ONE instruction; a reader waits until all N sequence numbers are there and
  // takes the words by v_readlane with constant lane numbers.  A get / put per source was a dozen scalar branches each
  // (the loop "slot there? else read again" does not compile into anything short): 280 cycles a run row.
  auto put_block = [&](u32 acc, u32 n) {           // lanes 0 .. n - 1 of acc
    if (lane < n) {
      const u32 a = L.ring + 4u * ((seq + lane) & (D6_RING - 1u));
      d6_ds_write32(a, acc);
      d6_ds_write32(a + 4u * D6_RING, seq + lane);
    }
    seq += n;
  };
  auto get_block = [&](u32 n) -> u32 {             // returns the words in lanes 0 .. n - 1
    const u64 want = (1ull << n) - 1ull;
    u32 spins = 0;
    const u32 a = L.ring + 4u * ((seq + lane) & (D6_RING - 1u));
    for (;;) {
      const u32 tag = d6_ds_read32(a + 4u * D6_RING);
      if ((__ballot(tag == seq + lane) & want) == want) break;
      if ((++spins & 63u) == 0 && (*v_abort != 0u || spins > D6_SPIN_MAX)) give_up();
    }
    seq += n;
    return d6_ds_read32(a);
  };

  // ---- the cell registers this wave owns: register `role` in (c0, l0), register 4 in (c1, l1) on the last wave
  float c0, c1 = 1e30f;
  u32 l0, l1 = 0;
  u32 reach;
  if (J.load) {
    {
      const float ec = J.init->c[64u * role + lane];
      c0 = ec < 1e29f ? (float)((double)ec + J.delta) : 1e30f;
      l0 = J.init->l[64u * role + lane];
    }
    if (has1) {
      const float ec = J.init->c[256u + lane];
      c1 = ec < 1e29f ? (float)((double)ec + J.delta) : 1e30f;
      l1 = J.init->l[256u + lane];
    }
    reach = SEG_CELLS - 1;
  } else {
    c0 = 1e30f;
    l0 = 0;
    if (w0 && lane == J.cell) c0 = J.level;
    reach = J.cell;
  }
#define D6_EACH(...)                                                                  \
  {                                                                                   \
    { const u32 s = role; float& cs = c0; u32& ls = l0; (void)s; __VA_ARGS__ }        \
    if (has1) { const u32 s = 4u; float& cs = c1; u32& ls = l1; (void)s; __VA_ARGS__ } \
  }
  u32 la_lo = J.la_lo;
  const u32 over_lo = J.over_lo;
  u16* const over = J.over;
  auto put_la = [&](u32 x, u16 v) {
    if (x < over_lo) la[x] = v;
    else if (x - over_lo < SEG_OVER) over[x - over_lo] = v;
  };
  float vmax = 0.0f;
  u32 mid_hi = 0;                                   // wave 0 only
  const float mid_margin = 2.0f * P.wmax[b];
  if (J.mid != nullptr && threadIdx.x == 0) J.mid->base = SEG_NONE;
  u32 wbase = (u32)__builtin_amdgcn_readfirstlane((int)J.start);
  bool noshort = J.noshort != 0;
  u32 skip = J.load ? (u32)__builtin_amdgcn_readfirstlane((int)J.init->skip) : 0u;
  u32 wcount = 0;                                   // windows finished (retired)
  u32 known_done = 0;                               // wave 0: what it last saw of the others' counters (their minimum)

  // the wave's integer table of the run-row weights for one binade (zmx_dp5.h: r1_build), every wave its own copy, rebuilt
  // at the same points of the program from the same source values
  u32 r1_lo = 0, r1_lit = 0xffffffffu, r1_rmax = 0;
  const u32 tiemask_b = P.tiemask[b];
  auto r1_build = [&](u32 sj, u32 lit) {
    const int e = (int)(sj >> 23) - 127;
    r1_lo = 0;
    if (((tiemask_b >> (e & 31)) & 1u) == 0) {
      double w = kInf;
      if (lane < 29) w = s_wtab[257u + 30u * lane];
      else if (lane == 29) w = s_wtab[1u + lit];
      uint2 v = make_uint2(D5_NOEDGE, D5_NOEDGE);
      u32 rr = 0;
      if (w < 1e300) {
        const u64 r = d5_rne_scaled(w, e);
        const u64 rh = r >> 29;
        if (rh < 0x800000ull) {
          rr = (u32)rh + ((r & 0x1fffffffull) > 0x10000000ull ? 1u : 0u);
          v.x = rr;
          v.y = (u32)rh;
        } else {
          rr = 0x800000u;
        }
      }
      wave_lds_sync();
      if (lane < 32) s_ri[lane] = v;
      wave_lds_sync();
#pragma unroll
      for (u32 i = 0; i < D5_RKN; i += 64) {
        const u32 k1 = i + lane - D5_RK0;
        if (i + lane < D5_RKN) s_rk[i + lane] = s_ri[k1 < ZMX_MAX_MATCH ? s_sym1[k1] : 31u];
      }
      wave_lds_sync();
      r1_rmax = d5_max64(rr);
      r1_lo = sj & 0x7f800000u;
      r1_lit = lit;
    }
  };
  // generic rows: the wave's staging area as a ring of two regions of 1024 codes (zmx_dp5.h)
  u32 st_iss = 0, st_land = 0;
  bool st_ok = false;
  u32 gpf_base = SEG_NONE, gpf_bw = 0;
  uint2 gpf_dh = make_uint2(0, 0);

  // ---- class-1 windows (wave 0 only: every edge stays in register 0): as in d5_run_job
#define D6_PICK16(CD, META, LB, U0)                                                               \
  {                                                                                               \
    _Pragma("unroll") for (int u = 0; u < 16; ++u) {                                              \
      const u32 w_ = rdlane_u32(META, (u32)((U0) + u));                                           \
      const u32 v_ = *(lds_u16p)((LB + (w_ >> 16)) << 1);                                         \
      u64 m_;                                                                                     \
      D5_BFM(m_, w_, (U0) + u + 1);                                                               \
      CD[u] = __builtin_amdgcn_inverse_ballot_w64(m_) ? v_ : 0u;                                  \
    }                                                                                             \
  }
#define D6_CHAIN_I(WV, U0)                                                                        \
  {                                                                                               \
    _Pragma("unroll") for (int u = 0; u < 16; ++u) {                                              \
      const u32 sj_ = rdlane_u32(cb, (u32)((U0) + u));                                            \
      const u32 t_ = sj_ + WV[u].x, th_ = sj_ + WV[u].y;                                          \
      lt_ = th_ < cb ? (u32)((U0) + u + 1) : lt_;                                                 \
      cb = cb < t_ ? cb : t_;                                                                     \
    }                                                                                             \
  }
  auto win_kind = [&](u32 wb) -> u32 {
    if ((wb & 31u) != 0 || wb + 32u > B) return 0u;
    const u32 f = winflag[wb >> 5] & 0xffu;
    if (f == 0) return 0u;
    const u32 g = bit_off + wb;
    const u64 two = ((u64)badw[(g >> 5) + 1] << 32) | badw[g >> 5];
    return (u32)(two >> (g & 31u)) == 0 ? f : 0u;
  };
  u32 pf_w = SEG_NONE, pf_meta = 0;
  const u32* __restrict__ badpos_v = P.badpos + (bd.pos_off >> 5);
  auto meta_load = [&](u32 w) -> u32 {
    const u32* a_ = lane < 40 ? wmeta + (u64)w * D5_WM + (lane < 35 ? lane : 0u) : badpos_v + ((bit_off + 32u * w) >> 5) + (lane & 1u);
    return *a_;
  };
  // every wave's part of a snapshot: the registers it owns (the last wave also the dead register 5)
  auto snap_store = [&](SegSnap* S) {
    D6_EACH({ S->c[64u * s + lane] = cs; S->l[64u * s + lane] = ls; })
    if (has1) { S->c[320u + lane] = 1e30f; S->l[320u + lane] = 0u; }
  };
  // the largest value any wave has seen so far, known to every wave afterwards (two barriers)
  auto vmax_all = [&]() -> float {
    const float m = wave_max_f32(vmax);
    if (lane == 0) __hip_atomic_fetch_max((d6_u32p)L.vmaxw, __float_as_uint(m), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // (non-negative floats order like their bit patterns)
    d6_barrier();
    const float r = __uint_as_float(*(d6_vu32p)L.vmaxw);
    d6_barrier();
    return r;
  };

  while (wbase < J.pend) {          // (J.pend = B + 1 on the last task: the window at B retires cell B)
    wbase = (u32)__builtin_amdgcn_readfirstlane((int)wbase);
    if (J.spec && la_lo == SEG_NONE && wbase >= J.pout) {
      // the first window at or after pout: the entry snapshot, every wave its cells
      snap_store(J.entry);
      if (threadIdx.x == 0) { J.entry->base = wbase; J.entry->noshort = noshort ? 1u : 0u; J.entry->skip = skip; }
      la_lo = wbase;
      vmax = 0.0f;
      if (w0) {
        const u32 e0_ = rdlane_u32(__float_as_uint(c0), skip < 32u ? skip : 0u);
        mid_hi = (J.mid != nullptr && e0_ < 0x70000000u) ? (e0_ & 0x7f800000u) + 0x00800000u : 0u;
      }
    }
    // ---- the window's token: its kind, and whether the mid snapshot is left here (zmx_dp5.h: the first window at a
    //      multiple of 64 whose first cell comes within two of the largest weights of the binade's end)
    u32 tok;
    const u64 t_tok0 = tick();
    if (w0) {
      u32 midnow = 0;
      if (mid_hi != 0 && la_lo != SEG_NONE && skip == 0 && (wbase & 63u) == 0) {
        const u32 c0_ = rdlane_u32(__float_as_uint(c0), 0);
        if (c0_ < 0x70000000u && __uint_as_float(c0_) + mid_margin >= __uint_as_float(mid_hi)) { midnow = 4u; mid_hi = 0; }
      }
      u32 kind;
      if (skip) {
        kind = 0;
      } else if (pf_w == (wbase >> 5) && (wbase & 31u) == 0 && wbase + 32u <= B) {
        const u32 f = rdlane_u32(pf_meta, 34) & 0xffu, g = bit_off + wbase;
        const u64 two = ((u64)rdlane_u32(pf_meta, 41) << 32) | rdlane_u32(pf_meta, 40);
        kind = f != 0 && (u32)(two >> (g & 31u)) == 0 ? f : 0u;
      } else {
        kind = win_kind(wbase);
      }
      if (kind == 3) kind = 0u;
      // not more than three windows ahead of the slowest of the others
      if (wcount > known_done + 2u) {
        u32 spins = 0;
        for (;;) {
          const u32 a = v_done[1], b2 = v_done[2], c2 = v_done[3];
          known_done = a < b2 ? (a < c2 ? a : c2) : (b2 < c2 ? b2 : c2);
          if (wcount <= known_done + 2u) break;
          if ((++spins & 63u) == 0 && (*v_abort != 0u || spins > D6_SPIN_MAX)) give_up();
        }
      }
      tok = kind | midnow;
      put(tok);
    } else {
      tok = get();
    }
    const u32 kind = tok & 3u;
    const u64 t_win0 = tick();
    if (PROF) pc[0] += t_win0 - t_tok0;
    if (tok & 4u) {
      // the mid snapshot (run tasks that grow out of their binade are re-run from here, zmx_dp5.h)
      snap_store(J.mid);
      const float pv_ = vmax_all();
      if (threadIdx.x == 0) { J.mid->base = wbase; J.mid->noshort = noshort ? 1u : 0u; J.mid->skip = 0u; J.mid->vmax = pv_; }
    }
    bool jumped = false;
    if (kind != 0 && st_ok) {           // (class-1 / class-2 windows use wave 0's staging area their own way)
      if (st_land < st_iss) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      st_ok = false;
    }
    if (kind == 1) {
      // ---- 32 positions, one cell register, no flags: wave 0's alone
      if (w0) {
        u32 lt_ = 0;
        bool ipath = false;
        if (IT.on) {
          const u32 cb = __float_as_uint(c0);
          const u32 mx = d5_max64(cb < 0x70000000u ? cb : 0u), mn = d5_min64(cb);
          ipath = mn >= IT.lo && mx + IT.span < IT.lo + 0x800000u;
        }
        const u32 wi_ = wbase >> 5;
        u32 meta_;
        u32 half_ = (wi_ & 1u) * D5_STAGE_HALF;
        if (pf_w == wi_) {
          meta_ = pf_meta;
        } else {
          meta_ = meta_load(wi_);
          const u16* src_ = reinterpret_cast<const u16*>((rows_addr + 2ull * winroff[wi_]) & ~15ull) + 8u * lane;
          dp_dma_piece(src_, (stage_half << 1) + half_);
          dp_dma_piece(src_ + 512, (stage_half << 1) + half_ + 1024u);
        }
        const u32 r0_ = rdlane_u32(meta_, 32), r1_ = rdlane_u32(meta_, 33);
        const u64 ad_ = rows_addr + 2ull * r0_, a0_ = ad_ & ~15ull;
        const u32 off_ = (u32)(ad_ - a0_);
        const u32 nb_ = 2u * (r1_ - r0_) + off_;
        if (nb_ > 2048u) {
          half_ = 0;
          const u16* src_ = reinterpret_cast<const u16*>(a0_) + 8u * lane;
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
          for (u32 k = 0; k < 4; ++k) dp_dma_piece(src_ + 512u * k, (stage_half << 1) + 1024u * k);
          pf_w = SEG_NONE;
        } else {
          const u16* src_ = reinterpret_cast<const u16*>((rows_addr + 2ull * r1_) & ~15ull) + 8u * lane;
          const u32 nh_ = ((wi_ + 1u) & 1u) * D5_STAGE_HALF;
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          pf_meta = meta_load(wi_ + 1u);
          dp_dma_piece(src_, (stage_half << 1) + nh_);
          dp_dma_piece(src_ + 512, (stage_half << 1) + nh_ + 1024u);
          pf_w = wi_ + 1u;
        }
        if (nb_ > 2048u) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const u32 lb_ = lane + (stage_half + ((half_ + off_) >> 1) - 32u);
        if (ipath) {
          u32 cd0[16], cd1[16];
          D6_PICK16(cd0, meta_, lb_, 0)
          D6_PICK16(cd1, meta_, lb_, 16)
          uint2 wi0[16], wi1[16];
#pragma unroll
          for (int u = 0; u < 16; ++u) wi0[u] = *reinterpret_cast<const uint2*>(reinterpret_cast<const char*>(s_itab) + cd0[u]);
#pragma unroll
          for (int u = 0; u < 16; ++u) wi1[u] = *reinterpret_cast<const uint2*>(reinterpret_cast<const char*>(s_itab) + cd1[u]);
          u32 cb = __float_as_uint(c0);
          D6_CHAIN_I(wi0, 0)
          D6_CHAIN_I(wi1, 16)
          c0 = __uint_as_float(cb);
        } else {
#pragma unroll 1
          for (u32 g = 0; g < 32u; g += 8u) {
            double wv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              const u32 w_ = rdlane_u32(meta_, g + (u32)u);
              const u32 v_ = *(lds_u16p)((lb_ + (w_ >> 16)) << 1);
              u64 m_;
              asm("s_bfm_b64 %0, %1, %2" : "=s"(m_) : "s"(w_), "s"(g + (u32)u + 1u));
              wv[u] = code_w(__builtin_amdgcn_inverse_ballot_w64(m_) ? v_ : 0u);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              const double cj = (double)rdlane_f32(c0, g + (u32)u);
              D3_RELAX_K(c0, lt_, wv[u], g + (u32)u + 1u)
            }
          }
        }
        l0 = lt_ ? wbase + lt_ : l0;
      }
      reach = reach > 63 ? reach : 63;
      noshort = false;
      if (PROF) pc[1] += tick() - t_win0;
    } else if (kind == 2) {
      // ---- longer edges, no flags: wave 0 the row's first 64 edges, wave 1 the next 64, the few rows that reach further
      //      their rest on demand (zmx_dp5.h, class 2); no edge of the window lies below mincost
      if (w0 && pf_w != SEG_NONE) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); pf_w = SEG_NONE; }
      u32 lt_ = 0;
      const u32 wi_ = wbase >> 5;
      u64 rb_ = rows_addr + 2ull * winroff[wi_];
#pragma unroll 1
      for (int h = 0; h < 4; ++h) {
        const d5_cdscp km_ = (d5_cdscp)(wmeta + (u64)wi_ * D5_WM + 8u * (u32)h);
        const d5_u32x4 kw_[2] = {km_[0], km_[1]};
        double wa[8];
        u32 ca[8];
        u32 ke8[8];
        u64 ra8[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          ke8[u] = kw_[u >> 2][u & 3] & 0xffffu;
          ra8[u] = rb_;
          ca[u] = 0;
          if (role < 2) {
            const __amdgpu_buffer_rsrc_t rs_ = __builtin_amdgcn_make_buffer_rsrc(
                reinterpret_cast<void*>(rb_), (short)0, (int)(2u * ke8[u]), 0x00020000);
            // (ONE register holding the whole offset, the instruction's offset field 0: the hardware range-checks the register
            //  by itself, and the compiler, left alone, shares one register — negative for the lanes just past the row's start
            //  — between the eight rows and puts the differences into the offset field: those lanes then read 0, "no edge".
            //  Found as a lost literal edge, in one build out of two.)
            int vo = (int)(lane2 - 2u * (u32)(8 * h + u + 1)) + (int)(128u * role);
            asm("" : "+v"(vo));
            ca[u] = (u32)(u16)__builtin_amdgcn_raw_buffer_load_b16(rs_, vo, 0, 0);
          }
          rb_ += 2u * ke8[u];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) wa[u] = code_w(ca[u]);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const u32 p = (u32)(8 * h + u);
          const double cj = (double)__uint_as_float(share(w0 ? rdlane_u32(__float_as_uint(c0), p) : 0u));
          if (role < 2) {
            D3_RELAX_K(c0, lt_, wa[u], p + 1u)
          }
          if (ke8[u] + p >= 128u) {            // the row reaches cell register 2 or beyond
            const u16* row = reinterpret_cast<const u16*>(ra8[u]);
            const u32 src1 = wbase + p + 1;
            reach = reach > ke8[u] + p ? reach : ke8[u] + p;
            if (role >= 2) {
              D6_EACH({
                const u32 k1 = lane + 64u * s - p - 1;
                if (k1 < ke8[u]) {
                  const double nc = code_w(row[k1]) + cj;
                  const bool upd = nc < (double)cs;
                  cs = upd ? (float)nc : cs;
                  ls = upd ? src1 : ls;
                }
              })
            }
          }
        }
      }
      if (role < 2) l0 = lt_ ? wbase + lt_ : l0;
      reach = reach > 127 ? reach : 127;
      noshort = false;
      if (PROF) pc[2] += tick() - t_win0;
    } else {
      // ---- position by position: stretches of run rows and of other rows, shortcuts, the general step
      if (w0 && pf_w != SEG_NONE) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); pf_w = SEG_NONE; }
      D5Cls W;
      W.nav = B - wbase < 32u ? B - wbase : 32u;
      {
        const u32 jj = wbase + lane;
        const bool act = lane < W.nav;
        const u32 cur = jj < B ? jj : B - 1;
        uint2 dh;
        u32 bw;
        if (gpf_base == wbase) { dh = gpf_dh; bw = gpf_bw; }
        else { dh = dbase[cur]; bw = badpos[(bit_off + cur) >> 5]; }
        {
          u64 ms_ = __ballot(act && ((dh.y >> 16) & 1u) != 0 && lane >= skip);
          if (noshort) ms_ &= ~(1ull << skip);
          const u32 nxt = ms_ ? ((wbase + (u32)__ffsll((long long)ms_) - 1u + ZMX_MAX_MATCH) & ~31u) : wbase + 32u;
          gpf_base = nxt;
          const u32 nj = nxt + lane < B ? nxt + lane : B - 1;
          gpf_dh = dbase[nj];
          gpf_bw = badpos[(bit_off + nj) >> 5];
        }
        W.kend = act ? (dh.y & 0xffffu) : 0u;
        W.roff = dh.x;
        W.ms = __ballot(act && ((dh.y >> 16) & 1u) != 0);
        W.fl = dh.y >> 17;
        W.mb = __ballot(act && ((bw >> ((bit_off + cur) & 31u)) & 1u) != 0);
      }
      if (noshort) W.ms &= ~(1ull << skip);   // squeeze.c:273
      const u32 skip0 = skip;
      if (PROF) pc[3] += tick() - t_win0 + (u64)(W.kend & 0u);
      // the source of window position p: the bit pattern of cell wbase + p, from wave 0
      auto src = [&](u32 p) -> u32 { return share(w0 ? rdlane_u32(__float_as_uint(c0), p) : 0u); };
      // the codes of a generic row are in the wave's ring (zmx_dp5.h)
      auto ring_at = [&](u32 ro, u32 ke) {
        const u32 rg0 = ro >> 10, rg1 = (ro + ke - 1u) >> 10;
        if (!st_ok || st_iss < rg0 || st_iss > rg0 + 2u) {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          st_iss = rg0;
          st_land = rg0;
          st_ok = true;
        }
        // (a region comes in over the one two below it: every read of that one this wave has issued must have been served —
        //  the DMA's data can be back from the L2 before a queued ds_read is.  Found by tools/r05_coop_stress.py: with the
        //  code reads of four rows issued back to back, 20 block runs in 2 400 read a row's codes from the region after next.)
        if (st_iss < rg0 + 2u) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        while (st_iss < rg0 + 2u) {
          const u16* src_ = rows + 1024u * st_iss + 8u * lane;
          const u32 dst_ = (stage_half << 1) + 2048u * (st_iss & 1u);
          dp_dma_piece(src_, dst_);
          dp_dma_piece(src_ + 512, dst_ + 1024u);
          ++st_iss;
        }
        if (st_land <= rg1) {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          st_land = st_iss;
        }
      };
      // ---- a stretch of OTHER rows from position p0 on (zmx_dp5.h: other_stretch), every wave its registers
      auto other_stretch = [&](u32 p0) -> u32 {
        const u64 rest = (~0ull << p0) & (W.nav >= 64u ? ~0ull : ((1ull << W.nav) - 1ull));
        const u64 odd = (__ballot((W.fl & 1u) != 0) | W.ms | W.mb) & rest;
        const u32 stop = odd ? (u32)__ffsll((long long)odd) - 1u : W.nav;
        // Four rows at a time: their codes (from the ring), then their weights (from the run's table) — the LDS reads of a
        // group issued together, before the group's stream operations: a put or a get orders every LDS access around it
        // (volatile), and a row whose two dependent LDS round trips sat between two stream operations cost 900 cycles.
        auto group = [&](u32 g, auto NTAG) {
          constexpr u32 N = decltype(NTAG)::value;
          u32 cdx[N], cdy[N];
#pragma unroll
          for (u32 u = 0; u < N; ++u) {
            const u32 p = g + u;
            const u32 ro = rdlane_u32(W.roff, p), ke = rdlane_u32(W.kend, p);
            ring_at(ro, ke);
            {
              const u32 k1 = lane + 64u * role - p - 1u;
              const u32 v = s_stage[(ro + k1) & 2047u];
              cdx[u] = k1 < ke ? (v & 0x3ff8u) : 0u;
            }
            cdy[u] = 0;
            if (has1) {
              const u32 k1 = lane + 256u - p - 1u;
              const u32 v = s_stage[(ro + k1) & 2047u];
              cdy[u] = k1 < ke ? (v & 0x3ff8u) : 0u;
            }
          }
          double wx[N], wy[N];
#pragma unroll
          for (u32 u = 0; u < N; ++u) { wx[u] = code_w(cdx[u]); wy[u] = has1 ? code_w(cdy[u]) : kInf; }
          if (N == 1) {
            const double cj = (double)__uint_as_float(src(g));
            const u32 src1 = wbase + g + 1u;
            D3_RELAX_K(c0, l0, wx[0], src1)
            if (has1) {
              D3_RELAX_K(c1, l1, wy[0], src1)
            }
          } else if (w0) {
            u32 acc = 0;
#pragma unroll
            for (u32 u = 0; u < N; ++u) {
              const u32 sj = rdlane_u32(__float_as_uint(c0), g + u);
              acc = lane == u ? sj : acc;
              const double cj = (double)__uint_as_float(sj);
              const u32 src1 = wbase + g + u + 1u;
              D3_RELAX_K(c0, l0, wx[u], src1)
            }
            put_block(acc, N);
          } else {
            const u32 blk = get_block(N);
#pragma unroll
            for (u32 u = 0; u < N; ++u) {
              const double cj = (double)__uint_as_float(rdlane_u32(blk, u));
              const u32 src1 = wbase + g + u + 1u;
              D3_RELAX_K(c0, l0, wx[u], src1)
              if (has1) {
                D3_RELAX_K(c1, l1, wy[u], src1)
              }
            }
          }
        };
        {
          u32 g = p0;
#pragma unroll 1
          for (; g + D6_OTHER_GROUP <= stop; g += D6_OTHER_GROUP) group(g, std::integral_constant<u32, D6_OTHER_GROUP>());
          if (g + 2u <= stop) { group(g, std::integral_constant<u32, 2u>()); g += 2u; }
#pragma unroll 1
          for (; g < stop; ++g) group(g, std::integral_constant<u32, 1u>());
        }
        const u32 far_ = d5_max64(lane >= p0 && lane < stop ? lane + W.kend : 0u);
        reach = reach > far_ ? reach : far_;
        noshort = false;
        return stop;
      };
      // ---- the interior of a run: a stretch of FULL run rows from p0 on (zmx_dp5.h: run_stretch)
      auto run_stretch = [&](u32 p0) -> u32 {
        const u32 f0 = rdlane_u32(W.fl, p0);
        const u64 rest = (~0ull << p0) & (W.nav >= 64u ? ~0ull : ((1ull << W.nav) - 1ull));
        const u32 endp = lane + W.kend;
        const u64 shortm = __ballot(W.kend != ZMX_MAX_MATCH) & rest;
        const u32 e_rel = shortm ? rdlane_u32(endp, (u32)__ffsll((long long)shortm) - 1u) : 0xffffffffu;
        const u64 odd = (__ballot(W.fl != f0 || endp > e_rel || (W.kend != ZMX_MAX_MATCH && endp != e_rel) || W.kend < 3u) | W.ms | W.mb) & rest;
        const u32 stop = odd ? (u32)__ffsll((long long)odd) - 1u : W.nav;
        if (stop <= p0) return p0;
        const bool cut_e = (shortm & ((stop >= 64u ? 0ull : (1ull << stop)) - 1ull)) != 0;
        const u32 lit = (f0 >> 1) & 255u;
        typedef __attribute__((address_space(3))) const d5_u32x2* lds_u2p;
        // (lane l of register s needs entry l + 64 s - p - 1 of the table by k: the register's 512 s bytes go into the address)
        const u32 rk_lane = (u32)(size_t)(__attribute__((address_space(3))) void*)s_rk + 8u * (lane + D5_RK0 - 1u) + 512u * role;
        auto row = [&](u32 sj, u32 a, u32 src1) {
          {
            const d5_u32x2 e = *(lds_u2p)(a);
            const u32 t_ = sj + e.x, th_ = sj + e.y;
            const u32 cb = __float_as_uint(c0);
            l0 = th_ < cb ? src1 : l0;
            c0 = __uint_as_float(cb < t_ ? cb : t_);
          }
          if (has1) {
            const d5_u32x2 e = *(lds_u2p)(a + 512u);
            const u32 t_ = sj + e.x, th_ = sj + e.y;
            const u32 cb = __float_as_uint(c1);
            l1 = th_ < cb ? src1 : l1;
            c1 = __uint_as_float(cb < t_ ? cb : t_);
          }
        };
        u32 p = p0;
        float c_sv0 = c0, c_sv1 = c1;
        u32 l_sv0 = l0, l_sv1 = l1;
        // Room for the whole stretch at once (zmx_dp5.h)?  Wave 0 looks at its cells and tells the others: the source at
        // p0 (every wave rebuilds its table from it at the same point) and the verdict.
        bool roomy = false;
        {
          const u32 s0 = share(w0 ? rdlane_u32(__float_as_uint(c0), p0) : 0u);
          if (s0 >= 0x41800000u && s0 < 0x4f000000u) {
            if ((s0 & 0x7f800000u) != r1_lo || lit != r1_lit) r1_build(s0, lit);
            if (r1_lo != 0) {
              u32 verdict = 0;
              if (w0) {
                const u32 cb0 = __float_as_uint(c0);
                const bool inw = lane >= p0 && lane < stop;
                const u32 mx = d5_max64(inw ? cb0 : 0u), mn = d5_min64(inw ? cb0 : 0xffffffffu);
                verdict = mn >= r1_lo && mx + r1_rmax < r1_lo + 0x800000u ? 1u : 0u;
              }
              roomy = share(verdict) != 0;
            }
          }
        }
        if (roomy) {
          // eight rows at a time, their table entries read together BEFORE the group's stream operations (see other_stretch)
          auto rgroup = [&](u32 g, auto NTAG) {
            constexpr u32 N = decltype(NTAG)::value;
            d5_u32x2 ex[N], ey[N];
#pragma unroll
            for (u32 u = 0; u < N; ++u) {
              const u32 a = rk_lane - 8u * (g + u);
              ex[u] = *(lds_u2p)(a);
              if (has1) ey[u] = *(lds_u2p)(a + 512u);
            }
            u32 acc = 0, blk = 0;
            if (N > 1 && !w0) blk = get_block(N);
#pragma unroll
            for (u32 u = 0; u < N; ++u) {
              u32 sj;
              if (N == 1) {
                sj = src(g);
              } else if (w0) {
                sj = rdlane_u32(__float_as_uint(c0), g + u);
                acc = lane == u ? sj : acc;
              } else {
                sj = rdlane_u32(blk, u);
              }
              const u32 src1 = wbase + g + u + 1u;
              {
                const u32 t_ = sj + ex[u].x, th_ = sj + ex[u].y;
                const u32 cb = __float_as_uint(c0);
                l0 = th_ < cb ? src1 : l0;
                c0 = __uint_as_float(cb < t_ ? cb : t_);
              }
              if (has1) {
                const u32 t_ = sj + ey[u].x, th_ = sj + ey[u].y;
                const u32 cb = __float_as_uint(c1);
                l1 = th_ < cb ? src1 : l1;
                c1 = __uint_as_float(cb < t_ ? cb : t_);
              }
            }
            if (N > 1 && w0) put_block(acc, N);
          };
#pragma unroll 1
          for (; p + D6_RUN_GROUP <= stop; p += D6_RUN_GROUP) rgroup(p, std::integral_constant<u32, D6_RUN_GROUP>());
          if (p + 4u <= stop) { rgroup(p, std::integral_constant<u32, 4u>()); p += 4u; }
          if (p + 2u <= stop) { rgroup(p, std::integral_constant<u32, 2u>()); p += 2u; }
#pragma unroll 1
          for (; p < stop; ++p) rgroup(p, std::integral_constant<u32, 1u>());
          p = stop;
        } else {
          u32 a0 = rk_lane - 8u * p0;
          for (; p < stop; ++p, a0 -= 8u) {
            const u32 sj = src(p);               // (a position the loop leaves is asked for again by the general step: every wave alike)
            if (sj < 0x41800000u || sj >= 0x4f000000u) break;                       // (2^4 .. 2^31)
            if ((sj & 0x7f800000u) != r1_lo || lit != r1_lit) r1_build(sj, lit);
            if (r1_lo == 0 || sj + r1_rmax >= r1_lo + 0x800000u) break;
            row(sj, a0, wbase + p + 1u);
          }
        }
        if (cut_e) {
          {
            const bool beyond = 64u * role + lane > e_rel;
            c0 = beyond ? c_sv0 : c0;
            l0 = beyond ? l_sv0 : l0;
          }
          if (has1) {
            const bool beyond = 256u + lane > e_rel;
            c1 = beyond ? c_sv1 : c1;
            l1 = beyond ? l_sv1 : l1;
          }
        }
        if (p > p0) {
          const u32 far_ = ZMX_MAX_MATCH + p - 1u < e_rel ? ZMX_MAX_MATCH + p - 1u : e_rel;
          reach = reach > far_ ? reach : far_;
          noshort = false;
        }
        return p;
      };
      for (u32 p = skip; p < W.nav;) {
        if (((W.ms >> p) & 1ull) == 0 && ((W.mb >> p) & 1ull) == 0) {
          if ((rdlane_u32(W.fl, p) & 1u) == 0) {
            const u64 t0_ = tick();
            const u32 q = other_stretch(p);
            if (PROF) { pc[4] += tick() - t0_ + (u64)(__float_as_uint(c0) & 0u); pc[10] += q - p; }
            p = q;
            continue;
          }
          if (P.int_path != 0) {
            const u64 t0_ = tick();
            const u32 q = run_stretch(p);
            if (PROF) { pc[5] += tick() - t0_ + (u64)(__float_as_uint(c0) & 0u); pc[11] += q - p; }
            if (q > p) { p = q; continue; }
          }
        }
        const u64 t_gen0 = tick();
        const u32 j = wbase + p;
        if ((W.ms >> p) & 1) {
          // long-run shortcut at position j (squeeze.c:251-271): every live cell moves 258 cells on
          if (w0 && lane >= skip0 && lane < p && wbase + lane >= la_lo) put_la(wbase + lane, (u16)(l0 ? wbase + lane + 1 - l0 : 0u));
          d6_barrier();                           // (everybody has taken the last shortcut's cells out of the area)
          D6_EACH({
            const u32 x = wbase + 64u * s + lane;
            ((d6_f32p)L.xc)[64u * s + lane] = cs;
            ((d6_u16p)L.xl)[64u * s + lane] = (u16)(ls ? x + 1 - ls : 0u);
            vmax = fmaxf(vmax, cs < 1e29f ? cs : 0.0f);
          })
          d6_barrier();
          // costs[j+t+258] = costs[j+t] + symbolcost for t = 0..257, unconditionally; cells j..j+257 are consumed with the
          // lengths they have now
          D6_EACH({
            const u32 t = 64u * s + lane;
            if (t < ZMX_MAX_MATCH && j + t >= la_lo) put_la(j + t, ((d6_u16p)L.xl)[p + t]);
          })
          // the registers go to the window that holds position j + 258: cell j + 258 + t sits at index t + d
          const u32 d = (j + ZMX_MAX_MATCH) & 31u;
          D6_EACH({
            const u32 t = 64u * s + lane - d;
            const bool in = t < ZMX_MAX_MATCH;
            const float v = ((d6_f32p)L.xc)[in ? p + t : 0u];
            cs = in ? (float)((double)v + symbolcost258) : 1e30f;
            ls = in ? j + t + 1 : 0u;
          })
          wbase = j + ZMX_MAX_MATCH - d;
          skip = d;
          reach = ZMX_MAX_MATCH - 1 + d;
          noshort = true;
          jumped = true;
          if (PROF) { pc[7] += tick() - t_gen0 + (u64)(__float_as_uint(c0) & 0u); ++pc[13]; }
          break;
        }
        const u32 ke = rdlane_u32(W.kend, p);
        const u32 ro = rdlane_u32(W.roff, p);
        const u32 sj = src(p);
        const double cj = (double)__uint_as_float(sj);
        const u32 src1 = j + 1;
        const u32 km1 = lane - p - 1;
        reach = reach > ke + p ? reach : ke + p;
        const u32 fl = rdlane_u32(W.fl, p);
        if (fl & 1u) {
          // a run row: the literal and (k, distance 1) for k = 3 .. ke (zmx_dp5.h)
          const u32 lit = (fl >> 1) & 255u;
          bool ipos = P.int_path != 0 && ((W.mb >> p) & 1ull) == 0 && sj >= 0x41800000u && sj < 0x4f000000u;
          if (ipos && ((sj & 0x7f800000u) != r1_lo || lit != r1_lit)) r1_build(sj, lit);
          ipos = ipos && r1_lo != 0 && sj + r1_rmax < r1_lo + 0x800000u;
          if (ipos) {
            D6_EACH({
              const u32 k1 = km1 + 64u * s;
              const uint2 e = s_ri[s_sym1[k1 < ke ? k1 : 1u]];
              const u32 t_ = sj + e.x, th_ = sj + e.y;
              const u32 cb = __float_as_uint(cs);
              ls = th_ < cb ? src1 : ls;
              cs = __uint_as_float(cb < t_ ? cb : t_);
            })
          } else {
            const double wl = code_w((1u + lit) * 8u);
            D6_EACH({
              const u32 k1 = km1 + 64u * s;
              const double w5 = s_w1[k1 < ke ? k1 : 1u];
              const double w = k1 == 0 ? wl : w5;
              const double mcl = k1 == 0 ? -kInf : mincost;
              DP_RELAX(cs, ls, w, mcl)
            })
          }
        } else {
          // any other row: its codes from the wave's ring
          ring_at(ro, ke);
          D6_EACH({
            const u32 k1 = km1 + 64u * s;
            const u32 cd = s_stage[(ro + (k1 < ke ? k1 : 0u)) & 2047u];
            const double w = k1 < ke ? code_w(cd & 0x3ff8u) : kInf;
            const double mcl = k1 == 0 ? -kInf : mincost;
            DP_RELAX(cs, ls, w, mcl)
          })
        }
        noshort = false;
        ++p;
        if (PROF) { pc[6] += tick() - t_gen0 + (u64)(__float_as_uint(c0) & 0u); ++pc[12]; }
      }
    }
    if (jumped) continue;
    // ---- cells wbase .. wbase + 31 are final: their lengths (wave 0), then the window moves on by 32 cells
    {
      const u32 jj = wbase + lane;
      if (w0) {
        if (lane < 32 && lane >= skip && jj >= la_lo && jj <= B) put_la(jj, (u16)(l0 ? jj + 1 - l0 : 0u));
        vmax = fmaxf(vmax, c0 < 1e29f ? c0 : 0.0f);
      }
      skip = 0;
      const bool lo = lane < 32;
      if (reach < 64) {              // only register 0 holds anything
        if (w0) {
          c0 = __uint_as_float(__builtin_amdgcn_permlane32_swap(__float_as_uint(c0), __float_as_uint(1e30f), false, false)[1]);
          l0 = __builtin_amdgcn_permlane32_swap(l0, 0u, false, false)[1];
        }
      } else {
        // register s's low half goes to wave s - 1 ...
        if (!w0) {
          const d6_u64p ch = (d6_u64p)(L.rot + ((role - 1u) * D6_ROTD + (wcount & (D6_ROTD - 1u))) * 256u);
          if (lo) ch[lane] = ((unsigned long long)l0 << 32) | __float_as_uint(c0);
        }
        // (... published by the counter below; first the own registers' halves move down)
        u32 in_c = __float_as_uint(1e30f), in_l = 0u;          // what comes in from above: fresh cells for the last register
        const u32 own_hi_c = __builtin_amdgcn_permlane32_swap(__float_as_uint(c0), __float_as_uint(c0), false, false)[1];
        const u32 own_hi_l = __builtin_amdgcn_permlane32_swap(l0, l0, false, false)[1];
        if (has1) {
          // registers 3 and 4 are both here: 3 <- (3.hi, 4.lo), 4 <- (4.hi, fresh)
          const auto r34c = __builtin_amdgcn_permlane32_swap(__float_as_uint(c0), __float_as_uint(c1), false, false);
          const auto r34l = __builtin_amdgcn_permlane32_swap(l0, l1, false, false);
          const auto r4c = __builtin_amdgcn_permlane32_swap(__float_as_uint(c1), __float_as_uint(1e30f), false, false);
          const auto r4l = __builtin_amdgcn_permlane32_swap(l1, 0u, false, false);
          c0 = __uint_as_float(lo ? r34c[1] : r34c[0]);
          l0 = lo ? r34l[1] : r34l[0];
          c1 = __uint_as_float(r4c[1]);
          l1 = r4l[1];
        }
        if (!w0) {
          d6_lds_fence();                         // (the channel's data before the counter: the LDS queue of a wave is in order, this is for the compiler)
          if (lane == 0) v_done[role] = wcount + 1u;
        }
        if (!has1) {
          // ... and register s + 1's low half comes from wave s + 1, once it has finished this window
          u32 spins = 0;
          const u64 t0_ = tick();
          while (v_done[role + 1u] <= wcount) {
            if ((++spins & 63u) == 0 && (*v_abort != 0u || spins > D6_SPIN_MAX)) give_up();
          }
          if (PROF) pc[8] += tick() - t0_;
          d6_lds_fence();
          const d6_vu64p ch = (d6_vu64p)(L.rot + (role * D6_ROTD + (wcount & (D6_ROTD - 1u))) * 256u);
          if (!lo) { const unsigned long long v = ch[lane - 32u]; in_c = (u32)v; in_l = (u32)(v >> 32); }
          c0 = __uint_as_float(lo ? own_hi_c : in_c);
          l0 = lo ? own_hi_l : in_l;
        }
      }
      if (reach < 64 && !w0) {
        // (nothing moved, but the counter says "window finished": wave 0's flow control reads it)
        if (lane == 0) v_done[role] = wcount + 1u;
      }
      reach = reach >= 32 ? reach - 32 : 0;
      wbase += 32;
      ++wcount;
      if (PROF) ++pc[14];
    }
  }
#undef D6_PICK16
#undef D6_CHAIN_I
  if (J.la_lo == 1 && threadIdx.x == 0) la[0] = 0;   // the head of the block
  if (J.exit) {
    snap_store(J.exit);
    D6_EACH({ vmax = fmaxf(vmax, cs < 1e29f ? cs : 0.0f); })
    const float vm = vmax_all();
    if (threadIdx.x == 0) { J.exit->vmax = vm; J.exit->base = wbase; J.exit->noshort = noshort ? 1u : 0u; J.exit->skip = skip; }
  }
#undef D6_EACH
  if (PROF && P.prof && lane == 0 && role < 2) {
    u64* o = P.prof + (u64)b * ZMX_PROF_N;
    if (w0) {
      pc[9] = tick() - t_job0;
      for (int i = 0; i < 15; ++i) atomicAdd(&o[i], pc[i]);
      atomicMax(&o[16], pc[9]);          // the longest job of the block
    } else {
      atomicAdd(&o[15], pc[15]);
      atomicAdd(&o[17], tick() - t_job0);
    }
  }
}

// One workgroup = one run task by four waves (d6_run_job).  The tasks come from P.run_list (the first pass: the run
// tasks of the table set, longest first) or from k_dpscan's list (P.redo_pass: the listed tasks of the run kind).
template <int WGS, bool PROF>
__global__ __launch_bounds__(64 * D6_NW, WGS) void k_dp6_spec(Dp4Params P) {
  __shared__ __align__(16) double s_wtab[ZMX_WTAB];
  __shared__ __align__(16) unsigned char s_buf[D6_NW][D5_STAGE_BYTES];
  __shared__ __align__(8) uint2 s_itab[ZMX_WTAB];
  __shared__ __align__(8) double s_w1[D5_W1];
  __shared__ u8 s_sym1[D5_W1];
  __shared__ __align__(8) uint2 s_ri[D6_NW][32];
  __shared__ __align__(8) uint2 s_rk[D6_NW][D5_RKN];
  __shared__ __align__(16) unsigned char s_coop[D6_LDS_BYTES];
  __shared__ float s_xc[DP_XN];
  __shared__ u16 s_xl[DP_XN];
  __shared__ u32 s_rmax;
  const u32 wave = (u32)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  u32 t;
  if (P.redo_pass) {
    if (blockIdx.x >= *P.redo_count) return;
    t = P.redo_wg[(u64)blockIdx.x * D5_WG];
    if (P.kind[t] == 0) return;                  // a text task: k_dp5_spec<.., 4, false>'s
  } else {
    t = P.run_list[blockIdx.x];
  }
  const SegTask T = P.tasks[t];
  const BlockDesc bd = P.blocks[T.block];
  const u32 B = (u32)(bd.inend - bd.instart);
  if (B == 0) return;
  for (u32 i = threadIdx.x; i < ZMX_WTAB; i += 64 * D6_NW) s_wtab[i] = P.wtab[(u64)T.block * ZMX_WTAB + i];
  if (threadIdx.x == 0) s_rmax = 0;
  __syncthreads();
  d5_build_w1(s_wtab, s_w1, s_sym1);
  float level = 0.0f;
  if (T.pout != 0) {
    level = P.est_bits ? P.est_bits[T.block] * ((float)T.q / (float)B) : P.lvl[t];
    if (!P.redo_pass) level *= P.level_scale;
    level = level >= 16.0f ? level : 16.0f;
  }
  // the integer table of the class-1 windows (wave 0's): the binade of the task's level
  D5IntTab IT;
  IT.on = false; IT.lo = 0; IT.span = 0;
  if (T.pout != 0 && P.int_path) {
    const u32 lb = __float_as_uint(level);
    const int e = (int)(lb >> 23) - 127;
    if (e >= 4 && e < 31 && ((P.tiemask[T.block] >> (e & 31)) & 1u) == 0) {
      d5_build_inttab(s_wtab, s_itab, s_rmax, e);
      __syncthreads();
      const u32 rm = s_rmax;
      IT.on = true;
      IT.lo = lb & 0x7f800000u;
      IT.span = rm < 0x100000u ? 33u * rm : 0x40000000u;
    }
  }
  D4Job J;
  J.start = T.q & ~31u;
  J.cell = T.q & 31u;
  J.noshort = 0;
  J.pout = T.pout;
  J.pend = T.pend;
  J.load = false;
  J.delta = 0;
  J.init = nullptr;
  J.entry = &P.entry[t];
  J.exit = &P.exit[t];
  J.mid = T.pout != 0 && P.mid != nullptr ? &P.mid[t] : nullptr;
  J.over_lo = T.pend <= B ? T.pend : SEG_NONE;
  J.over = P.over + (u64)t * SEG_OVER;
  if (T.pout == 0) {       // the head of the block
    J.spec = false;
    J.la_lo = 1;
    J.level = 0.0f;
  } else {
    J.spec = true;
    J.la_lo = SEG_NONE;
    J.level = level;
    if (P.est_bits && threadIdx.x == 0) P.lvl[t] = level;
  }
  const D6Lds L = d6_lds_carve(s_coop, s_xc, s_xl);
  switch (wave) {
    case 0: d6_run_job<PROF, 0>(P, J, T.block, bd, s_wtab, L, reinterpret_cast<u16*>(s_buf[0]), s_itab, IT, s_w1, s_sym1, s_ri[0], s_rk[0]); break;
    case 1: d6_run_job<PROF, 1>(P, J, T.block, bd, s_wtab, L, reinterpret_cast<u16*>(s_buf[1]), s_itab, IT, s_w1, s_sym1, s_ri[1], s_rk[1]); break;
    case 2: d6_run_job<PROF, 2>(P, J, T.block, bd, s_wtab, L, reinterpret_cast<u16*>(s_buf[2]), s_itab, IT, s_w1, s_sym1, s_ri[2], s_rk[2]); break;
    default: d6_run_job<PROF, 3>(P, J, T.block, bd, s_wtab, L, reinterpret_cast<u16*>(s_buf[3]), s_itab, IT, s_w1, s_sym1, s_ri[3], s_rk[3]); break;
  }
}

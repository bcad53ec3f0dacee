// k_dp5_spec: the speculative pass over the chain's tasks (see zmx_dp4.h for what a task is and why
// its result can be trusted) written for THROUGHPUT: one wave per task, no LDS staging, no helper
// waves, so that a CU runs a dozen tasks at once and the waves of a SIMD fill each other's stalls.
// Included only by zmx_hip.hip, after zmx_dp4.h (same jobs, snapshots, cell registers, arithmetic).
//
// k_dp4's four-wave pipeline is built for the latency of ONE chain (one workgroup per CU, 142 KB of
// LDS): right for the stretches k_dp4_fix has to run serially, wasteful when 50 000 independent tasks
// are waiting.  Here the unit is a WINDOW of 32 positions (the cell registers move 32 cells at a
// time, as in k_dp4), and a window is one of three kinds (k_mkdesc):
//
//   class 1  every position's edges stay inside cell register 0, no shortcut flag, no edge below mincost
//            (91 % of the positions of text).  The window's rows are consecutive in codes[]: they come as
//            one piece by LDS-DMA into the wave's staging area, requested a window ahead together with the
//            window's 40-word record (wmeta); lane l of row u then reads code l - u - 1 from LDS and its
//            weight from the run's table in LDS.  Inside one binade the chain step is integer arithmetic on
//            the floats' bit patterns (D5IntTab below).
//   class 2  longer rows, no flags: two buffer_load_ushort per row (lanes 0..127 of the row), the range
//            check of a per-row buffer descriptor dropping the lanes outside it; the few rows that reach
//            further fetch the rest on demand.
//   class 3  class 2 with rows that reach beyond cell register 1 (k_mkdesc): class 2 for the text variant of the
//            job; the run variant takes it as class 0 — stretches of other rows, below.
//   class 0  everything else, position by position with the reference's tests literally: long-run shortcut
//            positions (squeeze.c:251-271), edges below mincost (:293), ragged tails — and runs of equal
//            bytes, for which the RUNS variant of the job cuts a window into STRETCHES of one kind: run rows
//            (one table by length for the wave's binade, whole windows as straight-line code) and other rows
//            (codes from a ring of two regions kept ahead by LDS-DMA); DESIGN.md section 4, "Long runs".
//
// (The first version fetched every row through a buffer descriptor per position: the CU's address unit is
// busy 16 cycles per wave-wide load whatever the lanes do — DESIGN.md section 4 has the history.)
#pragma once

// A pointer every lane holds the same value of, as the compiler can see it (an SGPR pair).
template <typename T>
__device__ __forceinline__ T* uniform_ptr(T* p) {
  const u64 a = reinterpret_cast<u64>(p);
  const u32 lo = (u32)__builtin_amdgcn_readfirstlane((int)(u32)a), hi = (u32)__builtin_amdgcn_readfirstlane((int)(u32)(a >> 32));
  return reinterpret_cast<T*>(((u64)hi << 32) | lo);
}
// dph[] through the scalar cache: it is written by k_rowscan long before this kernel, and the
// constant address space is what makes the compiler use s_load for it although the kernel stores
// to other global arrays in between
typedef u32 d5_u32x4 __attribute__((ext_vector_type(4)));
typedef u32 d5_u32x2 __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(4))) d5_u32x4* d5_cdscp;

// What k_dp5_spec needs to fetch the rows of a 32-position window (aligned to the block start).  The rows of
// consecutive positions are consecutive in codes[] (dph's offsets are a running sum of the row lengths), so the
// wave copies the window's codes into LDS in one piece and lane l finds code l - u - 1 of row u at index
// rel[u] + l of the copy.  Everything a window needs besides the codes is ONE 40-word record (wmeta), fetched by
// ONE vector load a window ahead: words 0..31 = (rel[u] + 32) << 16 | kend[u], 32 / 33 = where the window's rows
// start / end in the block's codes, 34 = the window's class (1 = 32 positions, none flagged for the long-run
// shortcut, no edge beyond cell register 0; 2 = none flagged, edges of any length; 0 = neither).
//
// Why this shape — what round 2 measured on the way (all bit-exact, profiles/): a 16-byte buffer descriptor per
// position and a buffer_load per row: the CU's address unit is busy 16 cycles per wave-wide load whatever the
// lanes do, 256 cycles per position with 16 waves; descriptors built by SALU from packed row lengths: 30
// instructions per position around a chain step of six, and a wave issues one instruction per four cycles at
// best; s_load for the per-window words: SMEM shares its counter with LDS, so nothing scalar can be in flight
// across a window; and, throughout, three dependent memory round trips per window that four waves per SIMD do
// not hide (95 % of a window's 9 000 cycles were waiting).
#define D5_WM 40u          // words per window in wmeta[]
#define D5_WF_CODELESS 0x100u   // winflag / wmeta word 34, beside the window's kind in the low byte: a row without codes (k_rowscan)
#define D5_STAGE_BYTES 4352u    // a wave's staging area: >= 4096 + 16, >= 6 DP_XN
#define D5_STAGE_HALF 2176u     // two halves of >= 2048 + 16 for the LDS-DMA double buffer
struct MkDescParams {
  const BlockDesc* blocks;
  const uint2* dph;
  u32* wmeta;           // [windows + 1 per block][D5_WM]
  u32* winroff;         // [windows + 1 per block] = wmeta word 32 (the scalar path of windows nobody prefetched)
  const u32* win_off;   // [nb] first window of each block in winflag[] / winroff[] / wmeta[]
  u32* winflag;         // [windows + 1 per block] = wmeta word 34
};

__global__ __launch_bounds__(256) void k_mkdesc(MkDescParams P) {
  const BlockDesc bd = P.blocks[blockIdx.y];
  const u32 B = (u32)(bd.inend - bd.instart);
  const u32 p = blockIdx.x * 256u + threadIdx.x;
  const u32 lane = threadIdx.x & 63;
  uint2 dhw = make_uint2(0, 0);
  if (p < B) dhw = P.dph[bd.pos_off + p];
  // (whole waves take part in the ballots and the shuffle; a wave covers two windows)
  const bool flagged = p >= B || ((dhw.y >> 16) & 1u) != 0;
  const u64 not1 = __ballot(flagged || (dhw.y & 0xffffu) + (lane & 31u) >= 64u);    // an edge beyond cell register 0
  // flagged for the shortcut, or short — or a wide run row (k_rowscan): those go position by position, where their
  // weights come from the run-row table instead of six scattered code loads (d5_run_job)
  const u64 not2 = __ballot(flagged || (((dhw.y >> 17) & 1u) != 0 && (dhw.y & 0xffffu) >= 64u));
  // a row that reaches beyond cell register 1: kind 3 = kind 2 with such rows.  The text variant of the job takes it as
  // kind 2 (the rest of such a row on demand: rare there); the run variant position by position with staged codes
  // (the last 257 positions of every run of class Z are such rows: 1 800 cycles a position as kind 2)
  const u64 far3 = __ballot(p < B && (dhw.y & 0xffffu) + (lane & 31u) >= 128u);
  const u32 first = (u32)__shfl((int)dhw.x, (int)(lane & 32u), 64);                  // row offset of the window's first position
  const u32 w = P.win_off[blockIdx.y] + (p >> 5);
  u32* wm = P.wmeta + (u64)w * D5_WM;
  const u32 kend = dhw.y & 0xffffu;
  const u32 klay = dph_layout_len(dhw.y);      // (what the row takes in codes[]: nothing for a wide run row)
  const u64 nocodes = __ballot(p < B && (dhw.y & DPH_CODELESS) != 0);
  if (p < ((B + 31u) & ~31u)) wm[p & 31u] = p < B ? ((dhw.x - first + 32u - ((p & 31u) + 1u)) << 16) | kend : 0u;
  if ((lane & 31u) == 0 && p < B) {
    const u32 sh = lane & 32u;
    // (bit 8, D5_WF_CODELESS: a row of the window has no codes — such a window is of kind 0, and a task that holds one is
    //  re-run by the lean one-wave job, never by k_dp4's pipeline, whose ring expects every row's codes)
    const u32 f = ((u32)(not1 >> sh) == 0 ? 1u : (u32)(not2 >> sh) != 0 ? 0u : (u32)(far3 >> sh) == 0 ? 2u : 3u) |
                  ((u32)(nocodes >> sh) != 0 ? D5_WF_CODELESS : 0u);
    P.winflag[w] = f;
    P.winroff[w] = dhw.x;
    wm[32] = dhw.x;
    wm[34] = f;
  }
  if (p < B && ((p & 31u) == 31u || p + 1 == B)) wm[33] = dhw.x + klay;           // where the window's rows end
  if (p + 1 == B) {      // one record past the last window: where the block's rows end, class 0
    P.winroff[w + 1] = dhw.x + klay;
    P.winflag[w + 1] = 0;
    wm[D5_WM + 32] = dhw.x + klay;
    wm[D5_WM + 33] = dhw.x + klay;
    wm[D5_WM + 34] = 0;
  }
}

// ---------------------------------------------------------------------------------------------
// The chain step in INTEGER arithmetic, exact inside one binade.
//
// Let every value a window touches lie in [2^e, 2^(e+1)): floats there are the multiples of u = 2^(e-23),
// doubles the multiples of g = 2^(e-52) = u / 2^29, and the bit patterns of the floats are consecutive
// integers.  For a source cell cj = Cj u and a weight w the reference computes (squeeze.c:290-299)
//     nc = fl64(cj + w) = cj + r g,   r = RNE(w / g)        (cj is a multiple of 2^29 g: the rounding only sees w)
//     nc < (double)c   <=>   Cj 2^29 + r < C 2^29   <=>   Cj + (r >> 29) < C
//     (float)nc = (Cj + (r >> 29) + [r mod 2^29 > 2^28]) u           (no tie: r mod 2^29 != 2^28)
// so with the two small integers RH = r >> 29 and RR = RH + [r mod 2^29 > 2^28] per weight, on the BIT
// PATTERNS of the floats:   update <=> bits(cj) + RH < bits(c),   bits(c) <- min(bits(c), bits(cj) + RR)
// (if bits(cj) + RH < bits(c) then bits(cj) + RR <= bits(c), so the min is the update; otherwise it leaves
// c alone) — five full-rate 32-bit VALU instructions instead of six double-precision ones.  r is the very
// integer the tie mask is computed from (RunInfo, zmx_hip.hip); a binade in which a weight can tie is never
// taken.  D5IntTab is the table {RR, RH} of one binade, built by the workgroup from the run's weights;
// a window takes the integer path only if its 32 source cells lie in that binade and, with 33 times the
// largest weight on top (a cell can become a source 32 times over), stay below its end.
// ---------------------------------------------------------------------------------------------
struct D5IntTab {
  bool on;             // the workgroup has a table: s_itab[code >> 3] = {RR, RH} (no edge: 2^30 both)
  u32 lo;              // bit pattern of 2^e
  u32 span;            // 33 x the largest finite RR (saturated)
};
#define D5_NOEDGE 0x40000000u
#define D5_RK0 32u       // s_rk: entry of k - 1 = 0 (window positions 0 .. 31: up to 32 entries below it are asked for)
#define D5_RKN 352u      // ... and its size (k - 1 up to 63 + 256 above)
// the lane mask of bits [OFS, OFS + WIDTH) (WIDTH < 64; only the low six bits of either operand count)
#define D5_BFM(M, WIDTH, OFS) asm("s_bfm_b64 %0, %1, %2" : "=s"(M) : "s"(WIDTH), "n"(OFS))

// RNE(w 2^(52 - e)) for a finite w >= 0, by integer arithmetic on the mantissa (as RunInfo does it)
__device__ __forceinline__ u64 d5_rne_scaled(double w, int e) {
  const u64 bits = (u64)__double_as_longlong(w);
  const int ew = (int)((bits >> 52) & 0x7ffu);
  if (ew == 0) return 0;                                // zero (subnormals: far below any g)
  const u64 m = (bits & 0xfffffffffffffull) | (1ull << 52);   // w = m 2^(ew - 1075)
  const int sh = e - (ew - 1023);                       // r = RNE(m / 2^sh)
  if (sh <= 0) return sh < -10 ? ~0ull : m << -sh;
  if (sh >= 64) return 0;
  const u64 q = m >> sh, rem = m & ((1ull << sh) - 1), half = 1ull << (sh - 1);
  return q + ((rem > half || (rem == half && (q & 1))) ? 1 : 0);
}

// max / min over the 64 lanes of a wave (four DPP rows), the result uniform
__device__ __forceinline__ u32 d5_max64(u32 v) {
#define D5_DPP_STEP(CTRL) { const u32 t_ = (u32)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xf, 0xf, false); v = v > t_ ? v : t_; }
  D5_DPP_STEP(0x111) D5_DPP_STEP(0x112) D5_DPP_STEP(0x114) D5_DPP_STEP(0x118)
#undef D5_DPP_STEP
  const u32 a = (u32)__builtin_amdgcn_readlane((int)v, 15), b2 = (u32)__builtin_amdgcn_readlane((int)v, 31);
  const u32 a3 = (u32)__builtin_amdgcn_readlane((int)v, 47), a4 = (u32)__builtin_amdgcn_readlane((int)v, 63);
  const u32 m1 = a > b2 ? a : b2, m2 = a3 > a4 ? a3 : a4;
  return m1 > m2 ? m1 : m2;
}
__device__ __forceinline__ u32 d5_min64(u32 v) {
#define D5_DPP_STEP(CTRL) { const u32 t_ = (u32)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xf, 0xf, false); v = v < t_ ? v : t_; }
  D5_DPP_STEP(0x111) D5_DPP_STEP(0x112) D5_DPP_STEP(0x114) D5_DPP_STEP(0x118)
#undef D5_DPP_STEP
  const u32 a = (u32)__builtin_amdgcn_readlane((int)v, 15), b2 = (u32)__builtin_amdgcn_readlane((int)v, 31);
  const u32 a3 = (u32)__builtin_amdgcn_readlane((int)v, 47), a4 = (u32)__builtin_amdgcn_readlane((int)v, 63);
  const u32 m1 = a < b2 ? a : b2, m2 = a3 < a4 ? a3 : a4;
  return m1 < m2 ? m1 : m2;
}

// The workgroup builds the table of binade `e` from the run's weights (s_wtab is in place); returns false
// (every thread alike) when the binade is out of range or a weight can tie there.
__device__ __forceinline__ void d5_build_inttab(const double (&s_wtab)[ZMX_WTAB], uint2 (&s_itab)[ZMX_WTAB], u32& s_rmax, int e) {
  for (u32 i = threadIdx.x; i < ZMX_WTAB; i += blockDim.x) {
    const double w = s_wtab[i];
    uint2 v = make_uint2(D5_NOEDGE, D5_NOEDGE);
    if (w < 1e300 && i != 0) {
      const u64 r = d5_rne_scaled(w, e);
      const u64 rh = r >> 29;
      u32 rr = 0xfffffu;                // (a weight that large: the span test keeps every window out)
      if (rh < 0xfffffull) {
        rr = (u32)rh + ((r & 0x1fffffffull) > 0x10000000ull ? 1u : 0u);
        v.x = rr;
        v.y = (u32)rh;
      }
      atomicMax(&s_rmax, rr);
    }
    s_itab[i] = v;
  }
}

// The weights of a run row's edges, by slot of the row (slot k1 = edge k - 1): [0] the literal's place (the caller puts
// the position's own literal there), [1] the dead slot k = 2, [k1] = weight of (length k1 + 1, distance 1) — squeeze.c:146-157
// with dsym = 0 — for k1 = 2 .. 257, +inf beyond.  Built by the whole workgroup from its weight table.
#define D5_W1 264u
__device__ __forceinline__ void d5_build_w1(const double (&s_wtab)[ZMX_WTAB], double* s_w1, u8* s_sym1) {
  for (u32 k1 = threadIdx.x; k1 < D5_W1; k1 += blockDim.x) {
    double w = __longlong_as_double(0x7ff0000000000000ll);
    u32 sym = 31u;                           // no edge
    if (k1 >= 2 && k1 <= 257) {
      sym = (u32)(dev_length_symbol(k1 + 1u) - 257);
      w = s_wtab[257u + 30u * sym];
    }
    if (k1 == 0) sym = 29u;                  // the literal's entry of the wave's integer table
    s_w1[k1] = w;
    s_sym1[k1] = (u8)sym;
  }
  __syncthreads();
}

struct D5Cls {          // one window on the generic path: lane l < 32 = position wbase + l
  u32 kend, roff;
  u64 ms;               // flagged for the long-run shortcut
  u32 fl;               // bit 0: a run row (k_rowscan), bits 1..8: the literal
  u64 mb;               // owns an edge below mincost in this run (k_badscan)
  u32 nav;
};

// RUNS: the variant for tasks that walk runs of equal bytes (the run-row vector, staged codes of wide rows, window
// headers asked for ahead): built as a kernel of its own, k_dp5_spec<.., true>, so that what it keeps in registers
// is not the text tasks' problem (with both in one body k_dp5_spec<.., false> spilled and lost a fifth of its speed).
template <bool PROF, bool RUNS>
__device__ __forceinline__ void d5_run_job(const Dp4Params& P, const D4Job& J, u32 b, const BlockDesc& bd,
                                           const double (&s_wtab)[ZMX_WTAB], float* s_xc, u16* s_xl, u16* s_stage,
                                           const uint2 (&s_itab)[ZMX_WTAB], const D5IntTab& IT, const double* s_w1,
                                           const u8* s_sym1, uint2* s_ri, uint2* s_rk) {
  typedef __attribute__((address_space(3))) const u16* lds_u16p;
  typedef __attribute__((address_space(3))) d5_u32x4* lds_u4p;
  // (the LDS byte address of the wave's staging area, halved: it joins the scalar part of the lanes' addresses)
  const u32 stage_half = (u32)__builtin_amdgcn_readfirstlane((int)((u32)(size_t)(__attribute__((address_space(3))) void*)s_stage >> 1));
  const u32 lane = threadIdx.x & 63;
  const u32 lane2 = lane * 2u;
  // (byte offset of lane l into the row of position u of a window: 2 (l - u - 1), wrapping below the row so that
  //  the buffer's range check drops the lane.  It has to be ONE register: a negative register plus a positive
  //  instruction offset is out of range for the hardware even where the sum is not — measured, wrong data.)
  const u32 B = (u32)(bd.inend - bd.instart);
  const uint2* __restrict__ dbase = uniform_ptr(P.dph + bd.pos_off);
  const u32* __restrict__ badpos = uniform_ptr(P.badpos + (bd.pos_off >> 5));
  const u32 bit_off = (u32)(bd.pos_off & 31);
  u16* la = P.la + bd.la_off;
  const u16* __restrict__ rows = uniform_ptr(P.codes + P.code_base[b]);
  // the weight of a code (a byte offset into the run's table)
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

  float c[6];
  u32 l[6];
  u32 reach;
  if (J.load) {
#pragma unroll
    for (int s = 0; s < 6; ++s) {
      const float ec = J.init->c[64u * s + lane];
      c[s] = ec < 1e29f ? (float)((double)ec + J.delta) : 1e30f;
      l[s] = J.init->l[64u * s + lane];
    }
    reach = SEG_CELLS - 1;
  } else {
#pragma unroll
    for (int s = 0; s < 6; ++s) { c[s] = 1e30f; l[s] = 0; }
    if (lane == J.cell) c[0] = J.level;     // (cell > 0: a task that starts at a cut point inside its first window)
    reach = J.cell;
  }
  u32 la_lo = J.la_lo;
  // the length of cell x: into length_array, or — from pend on, where the successor may be writing
  // already (d4_copy_over) — into the task's side buffer
  const u32 over_lo = J.over_lo;
  u16* const over = J.over;
  auto put_la = [&](u32 x, u16 v) {
    if (x < over_lo) la[x] = v;
    else if (x - over_lo < SEG_OVER) over[x - over_lo] = v;
  };
  float vmax = 0.0f;
  u32 mid_hi = 0;                                   // (set with the entry snapshot)
  // (text tasks, round 5: between two looks — every 64 positions — the cells of text grow by ~250 bits, ten largest weights;
  //  the snapshot is left that much earlier.  The margin only decides WHERE the snapshot lies, never whether it counts.)
  const float mid_margin = (RUNS ? 2.0f : 16.0f) * P.wmax[b];
  if (J.mid != nullptr && lane == 0) J.mid->base = SEG_NONE;
  u32 wbase = (u32)__builtin_amdgcn_readfirstlane((int)J.start);
  bool noshort = J.noshort != 0;
  // After a long-run shortcut the cell registers are put back on the 32-position grid (windows at multiples of 32 from
  // the block start: every task then takes its snapshots at the same bases, whatever shortcuts lie behind it); the
  // first `skip` positions of that window lie inside the shortcut's span and are not walked.
  u32 skip = J.load ? (u32)__builtin_amdgcn_readfirstlane((int)J.init->skip) : 0u;
  // the wave's integer table of the run-row weights (29 length symbols at distance symbol 0, and one literal) for one
  // binade: r1_lo = bit pattern of its 2^e (0 = none built), see the run-row step below
  u32 r1_lo = 0, r1_lit = 0xffffffffu, r1_rmax = 0;
  const u32 tiemask_b = P.tiemask[b];
  // (re)builds the table for the binade of source value sj (a float's bit pattern) and literal byte lit; r1_lo stays 0
  // if that binade can tie.  s_ri: by symbol (0 .. 28 the length symbols, 29 the literal, 30 / 31 no edge); s_rk: by
  // k - 1 + D5_RK0 for a FULL row (k = 1 the literal, 2 no edge, 3 .. 258), no edge on either side of it — what lane l
  // of cell register s needs for the row of window position p is entry l + 64 s - p - 1 + D5_RK0, whatever l, s, p.
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
          rr = 0x800000u;                 // (a weight of the binade's own size: the room test keeps every position out)
        }
      }
      wave_lds_sync();
      if (lane < 32) s_ri[lane] = v;
      wave_lds_sync();
      if (RUNS) {
#pragma unroll
        for (u32 i = 0; i < D5_RKN; i += 64) {
          const u32 k1 = i + lane - D5_RK0;
          if (i + lane < D5_RKN) s_rk[i + lane] = s_ri[k1 < ZMX_MAX_MATCH ? s_sym1[k1] : 31u];
        }
        wave_lds_sync();
      }
      r1_rmax = d5_max64(rr);
      r1_lo = sj & 0x7f800000u;
      r1_lit = lit;
    }
  };
  // the header (dph, bad-edge word) of the generic window that will follow the current one, asked for as soon as the
  // current one's shortcut flags say where that is: behind a shortcut a window is one position and a jump, and the
  // header's round trip was most of its time
  // generic rows: the staging area is a ring of two regions of 1024 codes (region r of the block's codes at half r & 1);
  // the regions below st_iss have been asked for, those below st_land are known to have arrived (st_ok: the ring is valid)
  u32 st_iss = 0, st_land = 0;
  bool st_ok = false;
  u32 gpf_base = SEG_NONE, gpf_bw = 0;
  uint2 gpf_dh = make_uint2(0, 0);
  u64 n_fast = 0, n_slow = 0, n_int = 0;
  u64 kc[4] = {0, 0, 0, 0}, kn[4] = {0, 0, 0, 0};   // PROF: cycles / positions per window class: integer, class 1 in doubles, class 2, generic
  u64 pw[4] = {0, 0, 0, 0};   // PROF, integer windows: class decision, wait for the prefetched data, staging + prefetch issue, retire
  u64 pq[3] = {0, 0, 0};   // PROF: cycles of the integer windows: issuing the row fetches, waiting for them, the chain
  u64 go[2] = {0, 0};         // PROF: cycles / positions of whole windows of other rows
  u64 gt[4] = {0, 0, 0, 0};   // PROF, cycles of generic windows: header, run interior in unrolled windows, in loops, the general step's loop
  u64 gr[6] = {0, 0, 0, 0, 0, 0};   // PROF, positions: run interior (unrolled window, loop with room, loop with checks), run rows and other rows of the general step, class-2 rows that reach register 2
  u64 gq[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};   // PROF, generic windows: cycles / count of shortcuts, run rows (integer, doubles), other rows, window headers
  const u64 t_begin = PROF ? (u64)__builtin_readcyclecounter() : 0ull;

  // The codes of the rows of positions U0 .. U0 + 15 of a window whose codes are staged in LDS (below): lane l
  // wants code l - u - 1 of row u, staged at index rel[u] + l (+ where the copy starts: in LB, per lane); the lanes
  // that have one are bits u + 1 .. u + kend, the others get code 0 = no edge.  META: the window's record, word u
  // in lane u.  Per position: v_readlane, s_lshr, v_add_lshl, ds_read_u16, s_bfm, v_cndmask.
#define D5_PICK16(CD, META, LB, U0)                                                               \
  {                                                                                               \
    _Pragma("unroll") for (int u = 0; u < 16; ++u) {                                              \
      const u32 w_ = rdlane_u32(META, (u32)((U0) + u));                                           \
      const u32 v_ = *(lds_u16p)((LB + (w_ >> 16)) << 1);                                         \
      u64 m_;                                                                                     \
      D5_BFM(m_, w_, (U0) + u + 1);                                                               \
      CD[u] = __builtin_amdgcn_inverse_ballot_w64(m_) ? v_ : 0u;                                  \
    }                                                                                             \
  }
  // the chain over positions U0 .. U0 + 15 of the window at wbase
#define D5_CHAIN(WV, U0)                                                                          \
  {                                                                                               \
    _Pragma("unroll") for (int u = 0; u < 16; ++u) {                                              \
      const double cj = (double)rdlane_f32(c[0], (u32)((U0) + u));                                \
      D3_RELAX_K(c[0], lt_, WV[u], (u32)((U0) + u + 1))                                           \
    }                                                                                             \
  }
  // the integer step (D5IntTab) over positions U0 .. U0 + 15
#define D5_CHAIN_I(WV, U0)                                                                        \
  {                                                                                               \
    _Pragma("unroll") for (int u = 0; u < 16; ++u) {                                              \
      const u32 sj_ = rdlane_u32(cb, (u32)((U0) + u));                                            \
      const u32 t_ = sj_ + WV[u].x, th_ = sj_ + WV[u].y;                                          \
      lt_ = th_ < cb ? (u32)((U0) + u + 1) : lt_;                                                 \
      cb = cb < t_ ? cb : t_;                                                                     \
    }                                                                                             \
  }
  // a window that can take the fast path: at a multiple of 32 from the block start, statically
  // clean (k_mkdesc) and without an edge below mincost in this run (k_edges' bitmap)
  auto win_kind = [&](u32 wb) -> u32 {     // 1 / 2: cell registers the window's edges reach; 0: generic
    if ((wb & 31u) != 0 || wb + 32u > B) return 0u;
    const u32 f = winflag[wb >> 5] & 0xffu;
    if (f == 0) return 0u;
    const u32 g = bit_off + wb;
    const u64 two = ((u64)badw[(g >> 5) + 1] << 32) | badw[g >> 5];
    return (u32)(two >> (g & 31u)) == 0 ? f : 0u;
  };
  // What a class-1 window has requested for the window after it (nothing scalar: SMEM shares its counter with
  // LDS): that window's record — with the two words of k_badscan's bitmap that cover it in lanes 40 and 41 —
  // and the first 2 KB of its codes.
  typedef __attribute__((address_space(1))) const d5_u32x4* glb_u4p;
  u32 pf_w = SEG_NONE, pf_meta = 0;
  const u32* __restrict__ badpos_v = P.badpos + (bd.pos_off >> 5);
  auto meta_load = [&](u32 w) -> u32 {
    const u32* a_ = lane < 40 ? wmeta + (u64)w * D5_WM + (lane < 35 ? lane : 0u) : badpos_v + ((bit_off + 32u * w) >> 5) + (lane & 1u);
    return *a_;
  };
  while (wbase < J.pend) {          // (J.pend = B + 1 on the last task: the window at B retires cell B)
    wbase = (u32)__builtin_amdgcn_readfirstlane((int)wbase);
    if (J.spec && la_lo == SEG_NONE && wbase >= J.pout) {
      // the first window at or after pout: from here on the task owns the length_array; what the
      // registers hold now is compared with the predecessor's exit state
#pragma unroll
      for (int s = 0; s < 6; ++s) { J.entry->c[64u * s + lane] = c[s]; J.entry->l[64u * s + lane] = l[s]; }
      if (lane == 0) { J.entry->base = wbase; J.entry->noshort = noshort ? 1u : 0u; J.entry->skip = skip; }
      la_lo = wbase;
      vmax = 0.0f;
      // where the binade of the entry state ends (the bit pattern of its 2^(e+1)); 0: no mid snapshot for this task
      const u32 e0_ = rdlane_u32(__float_as_uint(c[0]), skip < 32u ? skip : 0u);
      mid_hi = (J.mid != nullptr && e0_ < 0x70000000u) ? (e0_ & 0x7f800000u) + 0x00800000u : 0u;
    }
    // A task that grows out of its binade is only accepted with a level that was exactly right (zmx_dp4.h: d4_accept),
    // and long tasks on long-run data mostly do grow out of one.  What they did BEFORE they came near the end of the
    // binade is as good as any task's: so the state at the first window whose first cell lies within two of the largest
    // weights of that end is left in mid[t], with the largest source value up to there — k_dp4_fix accepts that prefix by
    // the ordinary test and re-runs the task from the snapshot instead of from its start.  (The trigger only decides
    // WHERE the snapshot lies; whether the prefix stayed inside the binade is decided by its recorded maximum.)
    // (at a multiple of 64 only: k_dp4's pipeline, which may be the one to continue from here, walks in groups of 64 and
    //  has to stop at the base its successor started from)
    if (mid_hi != 0 && la_lo != SEG_NONE && skip == 0 && (wbase & 63u) == 0) {
      const u32 c0_ = rdlane_u32(__float_as_uint(c[0]), 0);
      if (c0_ < 0x70000000u && __uint_as_float(c0_) + mid_margin >= __uint_as_float(mid_hi)) {
#pragma unroll
        for (int s = 0; s < 6; ++s) { J.mid->c[64u * s + lane] = c[s]; J.mid->l[64u * s + lane] = l[s]; }
        const float pv_ = wave_max_f32(vmax);
        if (lane == 0) { J.mid->base = wbase; J.mid->noshort = noshort ? 1u : 0u; J.mid->skip = 0u; J.mid->vmax = pv_; }
        mid_hi = 0;
      }
    }
    bool jumped = false;
    const u64 tw0 = PROF ? (u64)__builtin_readcyclecounter() : 0ull;
    const u64 np0 = n_fast + n_slow;
    u32 pcls = 3;
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
    if (kind == 3) kind = RUNS ? 0u : 2u;
    if (kind != 0 && st_ok) {           // (the other windows use the staging area their own way)
      if (st_land < st_iss) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      st_ok = false;
    }
    if (kind == 1) {
      // ---- 32 positions, one cell register, no flags
      u32 lt_ = 0;                             // 1 + index of the last position that updated the cell
      bool ipath = false;
      pcls = 1;
      const u64 ta_ = PROF ? (u64)__builtin_readcyclecounter() : 0ull;
      u64 tb_ = 0;
      if (IT.on) {
        // every finite cell of the register inside the binade (cells not reached yet, 1e30f, become sources with a
        // value one of the others gave them: the span covers that)
        const u32 cb = __float_as_uint(c[0]);
        const u32 mx = d5_max64(cb < 0x70000000u ? cb : 0u), mn = d5_min64(cb);
        ipath = mn >= IT.lo && mx + IT.span < IT.lo + 0x800000u;
      }
      // The window's rows are consecutive in codes[] (at most 32 x 63 codes): the wave copies them into its LDS
      // staging area, normally from the registers the previous window asked for them in.
      const u32 wi_ = wbase >> 5;
      // The codes arrive by LDS-DMA (no registers in between: eight more live VGPRs made the allocator spill in the
      // chain), into one of two 2 KB halves of the staging area, the one the previous window asked for them in.
      u32 meta_;
      u32 half_ = (wi_ & 1u) * D5_STAGE_HALF;                  // byte offset of this window's half
      if (pf_w == wi_) {
        meta_ = pf_meta;
      } else {                                   // (the first window of a task, or after a window of another class)
        meta_ = meta_load(wi_);
        const u16* src_ = reinterpret_cast<const u16*>((rows_addr + 2ull * winroff[wi_]) & ~15ull) + 8u * lane;
        dp_dma_piece(src_, (stage_half << 1) + half_);
        dp_dma_piece(src_ + 512, (stage_half << 1) + half_ + 1024u);
      }
      const u32 r0_ = rdlane_u32(meta_, 32), r1_ = rdlane_u32(meta_, 33);
      const u64 ad_ = rows_addr + 2ull * r0_, a0_ = ad_ & ~15ull;
      const u32 off_ = (u32)(ad_ - a0_);                       // bytes the staged copy starts before the first row
      const u32 nb_ = 2u * (r1_ - r0_) + off_;
      if (nb_ > 2048u) {                                       // rare (rows of 33+ codes on average): the whole area, now
        half_ = 0;
        const u16* src_ = reinterpret_cast<const u16*>(a0_) + 8u * lane;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (u32 k = 0; k < 4; ++k) dp_dma_piece(src_ + 512u * k, (stage_half << 1) + 1024u * k);
        pf_w = SEG_NONE;
      } else {
        // the next window's record and codes: in flight while this one is worked on
        const u16* src_ = reinterpret_cast<const u16*>((rows_addr + 2ull * r1_) & ~15ull) + 8u * lane;
        const u32 nh_ = ((wi_ + 1u) & 1u) * D5_STAGE_HALF;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // (this window's codes are in place; they were asked for a window ago)
        if (PROF) tb_ = __builtin_readcyclecounter();
        pf_meta = meta_load(wi_ + 1u);
        dp_dma_piece(src_, (stage_half << 1) + nh_);
        dp_dma_piece(src_ + 512, (stage_half << 1) + nh_ + 1024u);
        pf_w = wi_ + 1u;
      }
      if (nb_ > 2048u) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const u32 lb_ = lane + (stage_half + ((half_ + off_) >> 1) - 32u);   // lane + (staged index of the window's first code) - 32, the LDS base included
      if (ipath) {
        // ---- every source of the window and every sum it can form lie in the table's binade: the integer step
        u32 cd0[16], cd1[16];
        u64 tq0 = 0, tq1 = 0, tq2 = 0;
        if (PROF) tq0 = __builtin_readcyclecounter();
        D5_PICK16(cd0, meta_, lb_, 0)
        D5_PICK16(cd1, meta_, lb_, 16)
        if (PROF) { tq1 = __builtin_readcyclecounter(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); tq2 = __builtin_readcyclecounter(); }
        uint2 wi0[16], wi1[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) wi0[u] = *reinterpret_cast<const uint2*>(reinterpret_cast<const char*>(s_itab) + cd0[u]);
#pragma unroll
        for (int u = 0; u < 16; ++u) wi1[u] = *reinterpret_cast<const uint2*>(reinterpret_cast<const char*>(s_itab) + cd1[u]);
        u32 cb = __float_as_uint(c[0]);
        D5_CHAIN_I(wi0, 0)
        D5_CHAIN_I(wi1, 16)
        c[0] = __uint_as_float(cb);
        if (PROF) {
          pcls = 0;
          n_int += 32;
          const u64 tq3 = __builtin_readcyclecounter();
          pq[0] += tq1 - tq0; pq[1] += tq2 - tq1; pq[2] += tq3 - tq2;
          pw[0] += ta_ - tw0; pw[1] += tb_ > ta_ ? tb_ - ta_ : 0; pw[2] += tb_ > ta_ ? tq0 - tb_ : 0;
        }
      } else {
        // ---- the reference's arithmetic (the head of a block, windows that straddle a binade, a binade with a
        //      possible tie): eight positions at a time, a real loop — few registers, it is the rare case
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
            const double cj = (double)rdlane_f32(c[0], g + (u32)u);
            D3_RELAX_K(c[0], lt_, wv[u], g + (u32)u + 1u)
          }
        }
      }
      l[0] = lt_ ? wbase + lt_ : l[0];
      reach = reach > 63 ? reach : 63;
      noshort = false;
      n_fast += 32;
    } else if (kind == 2) {
      if (pf_w != SEG_NONE) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); pf_w = SEG_NONE; }   // (codes on their way into the staging area)
      // ---- the same with longer edges (matches of more than 32 bytes: markup, source code): a second
      //      request per row for lanes 64 .. 127 of it and a second relaxation off the chain's critical
      //      path (register 1 is never the source of a position of this window); the few rows that
      //      reach further fetch the rest on demand.  No edge of the window lies below mincost
      //      (win_kind), so squeeze.c:293's test is a no-op here as well.
      pcls = 2;
      u32 lt_ = 0, lt1_ = 0;
      const u32 wi_ = wbase >> 5;
      u64 rb_ = rows_addr + 2ull * winroff[wi_];
#pragma unroll 1
      for (int h = 0; h < 4; ++h) {          // (eight positions at a time, a real loop: the registers of four waves per SIMD)
        const d5_cdscp km_ = (d5_cdscp)(wmeta + (u64)wi_ * D5_WM + 8u * (u32)h);
        const d5_u32x4 kw_[2] = {km_[0], km_[1]};
        double wa[8], wb[8];
        u32 ca[8], cb[8];
        u32 ke8[8];
        u64 ra8[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          ke8[u] = kw_[u >> 2][u & 3] & 0xffffu;
          ra8[u] = rb_;
          const __amdgpu_buffer_rsrc_t rs_ = __builtin_amdgcn_make_buffer_rsrc(
              reinterpret_cast<void*>(rb_), (short)0, (int)(2u * ke8[u]), 0x00020000);
          // (each offset ONE register of its own, the instruction's offset field 0 — see the note at lane2 above; the empty asm
          //  keeps the compiler from sharing a register between rows and putting the differences into the offset field, which
          //  it does in some builds: zmx_dp6.h has the case)
          int vo = (int)(lane2 - 2u * (u32)(8 * h + u + 1));
          int vo2 = vo + 128;
          asm("" : "+v"(vo), "+v"(vo2));
          ca[u] = (u32)(u16)__builtin_amdgcn_raw_buffer_load_b16(rs_, vo, 0, 0);
          cb[u] = (u32)(u16)__builtin_amdgcn_raw_buffer_load_b16(rs_, vo2, 0, 0);
          rb_ += 2u * ke8[u];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) { wa[u] = code_w(ca[u]); wb[u] = code_w(cb[u]); }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const u32 p = (u32)(8 * h + u);
          const double cj = (double)rdlane_f32(c[0], p);
          D3_RELAX_K(c[0], lt_, wa[u], p + 1u)
          D3_RELAX_K(c[1], lt1_, wb[u], p + 1u)
          if (ke8[u] + p >= 128u) {            // the row reaches cell register 2 or beyond
            if (PROF) ++gr[5];
            const u16* row = reinterpret_cast<const u16*>(ra8[u]);
            const u32 src1 = wbase + p + 1;
            reach = reach > ke8[u] + p ? reach : ke8[u] + p;
#pragma unroll
            for (int s = 2; s < 6; ++s) {
              const u32 k1 = lane + 64u * s - p - 1;
              if (k1 < ke8[u]) {
                const double nc = code_w(row[k1]) + cj;
                const bool upd = nc < (double)c[s];
                c[s] = upd ? (float)nc : c[s];
                l[s] = upd ? src1 : l[s];
              }
            }
          }
        }
      }
      l[0] = lt_ ? wbase + lt_ : l[0];
      l[1] = lt1_ ? wbase + lt1_ : l[1];
      reach = reach > 127 ? reach : 127;
      noshort = false;
      n_fast += 32;
    } else {
      // ---- position by position
      if (pf_w != SEG_NONE) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); pf_w = SEG_NONE; }   // (the shortcut spills cells where codes may be landing)
      D5Cls W;
      W.nav = B - wbase < 32u ? B - wbase : 32u;
      const bool fine_ = PROF && P.debug == 7;   // (the per-position timers cost a scalar-memory round trip each)
      const u64 th0_ = PROF ? (u64)__builtin_readcyclecounter() : 0ull;
      {
        const u32 jj = wbase + lane;
        const bool act = lane < W.nav;
        const u32 cur = jj < B ? jj : B - 1;
        uint2 dh;
        u32 bw;
        if (RUNS && gpf_base == wbase) { dh = gpf_dh; bw = gpf_bw; }
        else { dh = dbase[cur]; bw = badpos[(bit_off + cur) >> 5]; }
        if (RUNS) {
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
      if (noshort) W.ms &= ~(1ull << skip);   // squeeze.c:273: the position right after a shortcut is not tested again
      if (fine_) { gq[8] += (u64)__builtin_readcyclecounter() - th0_ + (u64)(W.kend & 0u); ++gq[9]; }
      if (PROF) gt[0] += (u64)__builtin_readcyclecounter() - th0_ + (u64)(W.kend & 0u);
      const u32 skip0 = skip;                 // (cells below it were given their lengths by the shortcut)
      // ---- a stretch of OTHER rows from position p0 on (the last 257 positions of a run: 258 edges a row, two or three
      // distances; neither run rows nor shortcuts, no edge below mincost — W.mb: squeeze.c:293 is a no-op for them, as in
      // a kind-2 window): nothing to decide per position, so the rows are straight-line code, the codes of the next row
      // read from the ring while this one is relaxed.  Position by position (the general step below) such a row cost
      // ~2 200 cycles: ~130 instructions with a scalar decision between every few of them, and a lone wave waits out
      // every one.  Returns the position the stretch ends at.
      auto other_stretch = [&](u32 p0) -> u32 {
        const u64 to0_ = PROF ? (u64)__builtin_readcyclecounter() : 0ull;
        const u64 rest = (~0ull << p0) & (W.nav >= 64u ? ~0ull : ((1ull << W.nav) - 1ull));
        const u64 odd = (__ballot((W.fl & 1u) != 0) | W.ms | W.mb) & rest;
        const u32 stop = odd ? (u32)__ffsll((long long)odd) - 1u : W.nav;
        auto ring = [&](u32 ro, u32 ke) {
          const u32 rg0 = ro >> 10, rg1 = (ro + ke - 1u) >> 10;
          if (!st_ok || st_iss < rg0 || st_iss > rg0 + 2u) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            st_iss = rg0;
            st_land = rg0;
            st_ok = true;
          }
          // (the region that comes in replaces the one two below it: this wave's reads of that one must have been served
          //  — zmx_dp6.h has the case that showed it; here the reads were always consumed before, this makes it explicit)
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
        // lane l of register s wants code l + 64 s - p - 1 of row p (none if that is not in [0, ke): code 0 = no edge)
        auto codes = [&](u32 (&cd)[5], u32 p) {
          const u32 ro = rdlane_u32(W.roff, p), ke = rdlane_u32(W.kend, p);
          ring(ro, ke);
#pragma unroll
          for (int s = 0; s < 5; ++s) {
            const u32 k1 = lane + 64u * (u32)s - p - 1u;
            const u32 v = s_stage[(ro + k1) & 2047u];
            cd[s] = k1 < ke ? (v & 0x3ff8u) : 0u;
          }
        };
        // (three stages, a row each: the codes of row p + 2 are asked for and the weights of row p + 1 — a second LDS round
        //  trip, behind the codes' — while row p is relaxed; with the weights read inside the relaxation a lone wave sat
        //  out that round trip at every position: 861 cycles a position, profiles/r06_profZ.txt)
        auto weights = [&](double (&w)[5], const u32 (&cd)[5]) {
#pragma unroll
          for (int s = 0; s < 5; ++s) w[s] = code_w(cd[s]);
        };
        auto relax = [&](const double (&w)[5], u32 p) {
          const double cj = (double)rdlane_f32(c[0], p);
          const u32 src1 = wbase + p + 1u;
#pragma unroll
          for (int s = 0; s < 5; ++s) { D3_RELAX_K(c[s], l[s], w[s], src1) }
        };
        u32 cdA[5], cdB[5];
        double wA[5], wB[5];
        codes(cdA, p0);
        if (p0 == 0 && stop == 32u) {           // (a whole window, the usual case: a loop without decisions)
          codes(cdB, 1u);
          weights(wA, cdA);
#pragma unroll 1
          for (u32 p = 0; p < 32u; p += 2u) {
            weights(wB, cdB);
            if (p + 2u < 32u) codes(cdA, p + 2u);
            relax(wA, p);
            if (p + 2u < 32u) weights(wA, cdA);
            if (p + 3u < 32u) codes(cdB, p + 3u);
            relax(wB, p + 1u);
          }
        } else {
          if (p0 + 1u < stop) codes(cdB, p0 + 1u);
          weights(wA, cdA);
#pragma unroll 1
          for (u32 p = p0; p < stop; p += 2u) {
            if (p + 1u < stop) weights(wB, cdB);
            if (p + 2u < stop) codes(cdA, p + 2u);
            relax(wA, p);
            if (p + 1u >= stop) break;
            if (p + 2u < stop) weights(wA, cdA);
            if (p + 3u < stop) codes(cdB, p + 3u);
            relax(wB, p + 1u);
          }
        }
        const u32 far_ = d5_max64(lane >= p0 && lane < stop ? lane + W.kend : 0u);
        reach = reach > far_ ? reach : far_;
        noshort = false;
        n_slow += stop - p0;
        if (PROF) { go[0] += (u64)__builtin_readcyclecounter() - to0_ + (u64)(__float_as_uint(c[0]) & 0u); go[1] += stop - p0; }
        return stop;
      };
      // ---- the interior of a run: a stretch of run rows from position p0 on — FULL ones (the literal and k = 3 .. 258 at
      // distance 1), same byte, no shortcut, no edge below mincost.  Lane l of register s then needs entry
      // l + 64 s - p - 1 of ONE table (s_rk) — an LDS address that moves down 8 bytes per position, the five registers
      // 512 bytes apart in the instruction's offset: per position one v_readlane, five ds_read_b64 and the 25 integer
      // operations of the five registers, ~35 instructions against ~100 of the general run-row step below (a symbol
      // look-up and a table look-up per register, clamped to the row's length: 1 100 cycles a position; a whole window
      // of this: 205).  Rows shorter than 258 belong too if they all end at the same cell E (the last positions of a run
      // whose matches do not continue elsewhere, or the block's end): the table has no edge beyond k = 258 but knows
      // nothing of E, so the cells beyond E are put back as they were once the stretch is done — no row of the stretch
      // has an edge to them (a full row of the same run ends at or before E), and no position of it lies beyond E.
      // Returns the position the stretch ends at (p0: no stretch — the general step takes the position).
      auto run_stretch = [&](u32 p0) -> u32 {
        const u32 f0 = rdlane_u32(W.fl, p0);
        const u64 rest = (~0ull << p0) & (W.nav >= 64u ? ~0ull : ((1ull << W.nav) - 1ull));
        const u32 endp = lane + W.kend;                                   // the last cell of the row, from the window's base
        const u64 shortm = __ballot(W.kend != ZMX_MAX_MATCH) & rest;
        const u32 e_rel = shortm ? rdlane_u32(endp, (u32)__ffsll((long long)shortm) - 1u) : 0xffffffffu;
        const u64 odd = (__ballot(((W.fl ^ f0) & 0x1ffu) != 0 || endp > e_rel || (W.kend != ZMX_MAX_MATCH && endp != e_rel) || W.kend < 3u) | W.ms | W.mb) & rest;
        const u32 stop = odd ? (u32)__ffsll((long long)odd) - 1u : W.nav;
        if (stop <= p0) return p0;
        const bool cut_e = (shortm & ((stop >= 64u ? 0ull : (1ull << stop)) - 1ull)) != 0;   // a short row in [p0, stop)
        const u32 lit = (f0 >> 1) & 255u;
        const u64 tr0_ = PROF ? (u64)__builtin_readcyclecounter() : 0ull;
        typedef __attribute__((address_space(3))) const d5_u32x2* lds_u2p;
        const u32 rk_lane = (u32)(size_t)(__attribute__((address_space(3))) void*)s_rk + 8u * (lane + D5_RK0 - 1u);
        auto row = [&](u32 sj, u32 a, u32 src1) {
          d5_u32x2 e5[5];
#pragma unroll
          for (int s = 0; s < 5; ++s) e5[s] = *(lds_u2p)(a + 512u * (u32)s);
#pragma unroll
          for (int s = 0; s < 5; ++s) {
            const u32 t_ = sj + e5[s].x, th_ = sj + e5[s].y;
            const u32 cb = __float_as_uint(c[s]);
            l[s] = th_ < cb ? src1 : l[s];
            c[s] = __uint_as_float(cb < t_ ? cb : t_);
          }
        };
        u32 p = p0;
        float c_sv[5];
        u32 l_sv[5];
        if (cut_e) {
#pragma unroll
          for (int s = 0; s < 5; ++s) { c_sv[s] = c[s]; l_sv[s] = l[s]; }
        }
        // Room for the whole stretch at once?  Every cell of it reached and inside the table's binade with the largest
        // weight to spare: cells only go down, and not below the smallest of them plus a weight, so every source of the
        // stretch passes the test the last loop below makes per position — which then is not in the chain any more (a
        // scalar compare and branch between dependent vector instructions: half of a position's time).
        bool roomy = false;
        {
          const u32 cb0 = __float_as_uint(c[0]);
          const u32 s0 = rdlane_u32(cb0, p0);
          if (s0 >= 0x41800000u && s0 < 0x4f000000u) {
            if ((s0 & 0x7f800000u) != r1_lo || lit != r1_lit) r1_build(s0, lit);
            if (r1_lo != 0) {
              const bool inw = lane >= p0 && lane < stop;
              const u32 mx = d5_max64(inw ? cb0 : 0u), mn = d5_min64(inw ? cb0 : 0xffffffffu);
              roomy = mn >= r1_lo && mx + r1_rmax < r1_lo + 0x800000u;
            }
          }
        }
        const bool whole = roomy && p0 == 0 && stop == 32u && !cut_e;
        if (whole) {
#pragma unroll
          for (u32 q = 0; q < 32u; ++q) row(rdlane_u32(__float_as_uint(c[0]), q), rk_lane - 8u * q, wbase + q + 1u);
          p = 32u;
          if (PROF) { gr[0] += 32; gt[1] += (u64)__builtin_readcyclecounter() - tr0_ + (u64)(__float_as_uint(c[0]) & 0u); }
        } else if (roomy) {
          if (PROF) gr[1] += stop - p0;
          u32 a0 = rk_lane - 8u * p0;
          for (; p < stop; ++p, a0 -= 8u) row(rdlane_u32(__float_as_uint(c[0]), p), a0, wbase + p + 1u);
        } else {
          u32 a0 = rk_lane - 8u * p0;
          for (; p < stop; ++p, a0 -= 8u) {
            const u32 sj = rdlane_u32(__float_as_uint(c[0]), p);
            if (sj < 0x41800000u || sj >= 0x4f000000u) break;                       // (2^4 .. 2^31)
            if ((sj & 0x7f800000u) != r1_lo || lit != r1_lit) r1_build(sj, lit);
            if (r1_lo == 0 || sj + r1_rmax >= r1_lo + 0x800000u) break;
            row(sj, a0, wbase + p + 1u);
            if (PROF) ++gr[2];
          }
        }
        if (cut_e) {
#pragma unroll
          for (int s = 0; s < 5; ++s) {
            const bool beyond = 64u * (u32)s + lane > e_rel;
            c[s] = beyond ? c_sv[s] : c[s];
            l[s] = beyond ? l_sv[s] : l[s];
          }
        }
        if (PROF && !whole) gt[2] += (u64)__builtin_readcyclecounter() - tr0_ + (u64)(__float_as_uint(c[0]) & 0u);
        if (p > p0) {
          const u32 far_ = ZMX_MAX_MATCH + p - 1u < e_rel ? ZMX_MAX_MATCH + p - 1u : e_rel;
          reach = reach > far_ ? reach : far_;
          noshort = false;
          n_slow += p - p0;
          if (PROF) n_int += p - p0;
        }
        return p;
      };
      const u64 tg0_ = PROF ? (u64)__builtin_readcyclecounter() : 0ull;
      for (u32 p = skip; p < W.nav;) {
        if (RUNS && ((W.ms >> p) & 1ull) == 0 && ((W.mb >> p) & 1ull) == 0) {
          // (a window is cut into stretches of one kind; what is left for the step below: shortcuts, rows with an edge
          //  below mincost, run rows outside the integer table's reach)
          if ((rdlane_u32(W.fl, p) & 1u) == 0) { p = other_stretch(p); continue; }
          if (P.int_path != 0) {
            const u32 q = run_stretch(p);
            if (q > p) { p = q; continue; }
          }
        }
        const u32 j = wbase + p;
        const u64 tp0_ = fine_ ? (u64)__builtin_readcyclecounter() : 0ull;
        u32 gk_ = 6;
        if ((W.ms >> p) & 1) {
          // long-run shortcut at position j (squeeze.c:251-271)
          // ---- (RUNS) what FOLLOWS the shortcut deep inside a run is known before it is walked: position e = j + 258 is
          // exempt from the test (squeeze.c:273), a full run row, and e + 1 is flagged again — the reference then does 258
          // more copies, and so on every 259 bytes to the end of the run.  Walked the ordinary way each such cycle was a
          // spill of the cells to put them back on the 32-position grid, a window header with its memory round trip, a
          // one-position stretch with its room test: 10 000 - 16 000 cycles for one row and 258 additions (23 % of a run
          // task's time, profiles/r06_profZ.txt).  The CHAIN below asks for the headers of up to 32 cycles ahead in ONE
          // load, keeps the cells in a frame that does not move (cycle m: the frame starts at e_0 + 258 m, the exempt cell is
          // its index m, the copies land on the lanes they came from) and goes back on the grid once, when the pattern or a
          // task boundary ends it.  Every cycle does what the windows it replaces would have done, in their order.
          u64 ch_good = 0;
          uint2 ch_dh = make_uint2(0, 0);
          if (RUNS && P.chain_fast) {
            const u32 pe = j + ZMX_MAX_MATCH + 259u * (lane >> 1) + (lane & 1u);     // lane 2m: e_m, lane 2m + 1: e_m + 1
            const bool pin = pe < B;
            const u32 pq = pin ? pe : B - 1u;
            ch_dh = dbase[pq];
            const u32 pbw = badpos[(bit_off + pq) >> 5];
            const bool bad = ((pbw >> ((bit_off + pq) & 31u)) & 1u) != 0;
            const u64 ok_e = __ballot(pin && ((ch_dh.y >> 17) & 1u) != 0 && (ch_dh.y & 0xffffu) == ZMX_MAX_MATCH && !bad);
            const u64 ok_s = __ballot(pin && ((ch_dh.y >> 16) & 1u) != 0);
            ch_good = ok_e & (ok_s >> 1) & 0x5555555555555555ull;
          }
          if (lane >= skip0 && lane < p && wbase + lane >= la_lo) put_la(wbase + lane, (u16)(l[0] ? wbase + lane + 1 - l[0] : 0u));
          if (st_ok && st_land < st_iss) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          st_ok = false;                          // (the cells are spilled where the staged codes lie)
          wave_lds_sync();
#pragma unroll
          for (int s = 0; s < 6; ++s) {
            const u32 x = wbase + 64u * s + lane;
            s_xc[64 * s + lane] = c[s];
            s_xl[64 * s + lane] = (u16)(l[s] ? x + 1 - l[s] : 0u);
            vmax = fmaxf(vmax, c[s] < 1e29f ? c[s] : 0.0f);
          }
          wave_lds_sync();
          // costs[j+t+258] = costs[j+t] + symbolcost for t = 0..257, unconditionally; cells
          // j..j+257 are consumed with the lengths they have now
#pragma unroll
          for (int r = 0; r < 5; ++r) {
            const u32 t = 64u * r + lane;
            if (t < ZMX_MAX_MATCH && j + t >= la_lo) put_la(j + t, s_xl[p + t]);
          }
          u32 e_next = j + ZMX_MAX_MATCH;         // the exempt position behind the last shortcut taken
          u32 src0 = p;                           // where its cell lies in the spilled copy
          bool ch_ran = false;
          if (RUNS && (ch_good & 1ull) != 0) {
            // the frame: index i = 64 s + lane is cell fb + i; the exempt cell of cycle m is index m
            u32 fb = e_next, n = 0;
            bool loaded = false;
            for (;;) {
              const u32 e = fb + n;               // = j + 258 + 259 m
              const u32 d_ = e & 31u, wb_ = e - d_;
              // what the loop head would do with the window at wb_ (and, for d_ = 31, with the one behind it): end the
              // task, take the entry snapshot, take a mid snapshot — the ordinary walk does those
              if (((ch_good >> (2u * n)) & 1ull) == 0 || n >= 30u || d_ == 31u || wb_ >= J.pend) break;
              if (J.spec && la_lo == SEG_NONE && wb_ >= J.pout) break;
              if (!loaded) {
#pragma unroll
                for (int s = 0; s < 6; ++s) {
                  const u32 i = 64u * s + lane;
                  const bool in = i < ZMX_MAX_MATCH;
                  const float v = s_xc[in ? p + i : 0u];
                  c[s] = in ? (float)((double)v + symbolcost258) : 1e30f;
                  l[s] = in ? j + i + 1u : 0u;
                }
                loaded = true;
              }
              if (mid_hi != 0 && la_lo != SEG_NONE && d_ == 0 && (wb_ & 63u) == 0) {
                const u32 c0_ = rdlane_u32(__float_as_uint(c[0]), n);
                if (c0_ < 0x70000000u && __uint_as_float(c0_) + mid_margin >= __uint_as_float(mid_hi)) break;
              }
              // ---- the exempt position e: a full run row (the literal and (k, distance 1), k = 3 .. 258; no edge of it below
              //      mincost — its bit of k_badscan's map is clear —, so squeeze.c:293 is a no-op) from the cell at index n
              {
                const u32 y_ = rdlane_u32(ch_dh.y, 2u * n);
                const double wl = code_w((1u + ((y_ >> 18) & 255u)) * 8u);
                const double cj = (double)rdlane_f32(c[0], n);
                const u32 src1 = e + 1u;
                double w5[5];
#pragma unroll
                for (int s = 0; s < 5; ++s) {
                  const u32 k1 = 64u * (u32)s + lane - n - 1u;
                  w5[s] = s_w1[k1 < ZMX_MAX_MATCH ? k1 : 1u];         // ([1] = the dead slot k = 2: +inf)
                }
#pragma unroll
                for (int s = 0; s < 5; ++s) {
                  const u32 k1 = 64u * (u32)s + lane - n - 1u;
                  const double w = k1 == 0 ? wl : w5[s];
                  D3_RELAX_K(c[s], l[s], w, src1)
                }
              }
              // cell e is final
              if (lane == n && e >= la_lo) put_la(e, (u16)(l[0] ? e + 1u - l[0] : 0u));
#pragma unroll
              for (int s = 0; s < 5; ++s) vmax = fmaxf(vmax, c[s] < 1e29f ? c[s] : 0.0f);
              // ---- the shortcut at e + 1: cells e + 1 .. e + 258 (indices n + 1 .. n + 258) are consumed with the lengths
              //      they have, cell x + 258 = cell x + symbolcost — on the lane it is on, in the frame 258 further on
#pragma unroll
              for (int s = 0; s < 5; ++s) {
                const u32 i = 64u * (u32)s + lane;
                const u32 x = fb + i;
                const bool in = i - n - 1u < ZMX_MAX_MATCH;
                if (in && x >= la_lo) put_la(x, (u16)(l[s] ? x + 1u - l[s] : 0u));
                c[s] = in ? (float)((double)c[s] + symbolcost258) : 1e30f;
                l[s] = in ? x + 1u : 0u;
              }
              fb += ZMX_MAX_MATCH;
              ++n;
              ++n_slow;
              if (PROF) ++gr[2];
            }
            if (loaded) {
              // (the header of the window the walk goes on in: asked for now — the one requested before the chain was for
              //  the window behind the first shortcut)
              {
                const u32 nb_ = (fb + n) & ~31u;
                const u32 nj = nb_ + lane < B ? nb_ + lane : B - 1u;
                gpf_base = nb_;
                gpf_dh = dbase[nj];
                gpf_bw = badpos[(bit_off + nj) >> 5];
              }
              wave_lds_sync();
#pragma unroll
              for (int s = 0; s < 5; ++s) s_xc[64 * s + lane] = c[s];
              wave_lds_sync();
              ch_ran = true;
              e_next = fb + n;
              src0 = n;
            }
          }
          // the registers go to the window that holds position e_next (j + 258 without the chain): its cell t sits at index t + d
          const u32 d = e_next & 31u;
#pragma unroll
          for (int s = 0; s < 6; ++s) {
            const u32 t = 64u * s + lane - d;
            const bool in = t < ZMX_MAX_MATCH;
            const float v = s_xc[in ? src0 + t : 0u];
            c[s] = in ? (ch_ran ? v : (float)((double)v + symbolcost258)) : 1e30f;
            l[s] = in ? e_next + t - (ZMX_MAX_MATCH - 1u) : 0u;
          }
          wave_lds_sync();
          if (fine_) { gq[0] += (u64)__builtin_readcyclecounter() - tp0_; ++gq[1]; }
          wbase = e_next - d;
          skip = d;
          reach = ZMX_MAX_MATCH - 1 + d;
          noshort = true;
          jumped = true;
          break;
        }
        const u32 ke = rdlane_u32(W.kend, p);
        const u32 ro = rdlane_u32(W.roff, p);
        const double cj = (double)rdlane_f32(c[0], p);
        const u32 src1 = j + 1;
        const u32 km1 = lane - p - 1;
        const u32 smax = (ke + p) >> 6;
        reach = reach > ke + p ? reach : ke + p;
        const u32 fl = rdlane_u32(W.fl, p);
        if (PROF) ++gr[(fl & 1u) ? 3 : 4];
        if ((RUNS || ((fl >> 9) & 1u) != 0) && (fl & 1u)) {
          // (the text variant comes here for the wide run rows, which have no codes: DPH_CODELESS = bit 9 of fl)
          // a run row (k_rowscan): the literal and (k, distance 1) for k = 3 .. ke — weights from tables of those edges,
          // nothing read from codes[]: inside runs of equal bytes, where every row is 258 wide, the six scattered code
          // loads per position were all the time there was (2 600 - 3 500 cycles a position).
          const u32 lit = (fl >> 1) & 255u;
          // The integer step (see D5IntTab above; the same arithmetic on the bit patterns) needs only the SOURCE cell
          // inside the table's binade with room for the largest weight: a target in a higher binade (or unreached,
          // 1e30) compares greater as a bit pattern exactly when it is greater, one in a lower binade smaller.  The
          // table is the wave's own — 29 length symbols at distance symbol 0 plus the run's literal, for one binade —
          // and is rebuilt when the chain leaves the binade or the literal changes (a few times per task).
          const u32 sj = rdlane_u32(__float_as_uint(c[0]), p);
          bool ipos = RUNS && P.int_path != 0 && ((W.mb >> p) & 1ull) == 0 && sj >= 0x41800000u && sj < 0x4f000000u;   // (2^4 .. 2^31)
          if (ipos && ((sj & 0x7f800000u) != r1_lo || lit != r1_lit)) r1_build(sj, lit);
          ipos = ipos && r1_lo != 0 && sj + r1_rmax < r1_lo + 0x800000u;
          if (ipos) {
            // branch-free over the five registers a row can reach (index <= 31 + 258): the five symbol reads, then the five
            // table reads, in flight together — one dependent pair of LDS round trips per register, register after
            // register, was 1 700 cycles a position.  (Tried and slower: the symbols by symbols.h's closed form, 80
            // instructions; the weights as a register vector moved one lane up per position by DPP, 1 550 cycles a
            // position against 1 250 — a lone wave on its SIMD pays for every instruction, not for the LDS.)
            u32 sym5[5];
#pragma unroll
            for (int s = 0; s < 5; ++s) {
              const u32 k1 = km1 + 64u * s;
              sym5[s] = s_sym1[k1 < ke ? k1 : 1u];              // ([0] = 29: the literal, [1] = 31: the dead slot k = 2, no edge)
            }
            uint2 e5[5];
#pragma unroll
            for (int s = 0; s < 5; ++s) e5[s] = s_ri[sym5[s]];
#pragma unroll
            for (int s = 0; s < 5; ++s) {
              const u32 t_ = sj + e5[s].x, th_ = sj + e5[s].y;
              const u32 cb = __float_as_uint(c[s]);
              l[s] = th_ < cb ? src1 : l[s];
              c[s] = __uint_as_float(cb < t_ ? cb : t_);
            }
            if (PROF) ++n_int;
            gk_ = 2;
          } else {
            gk_ = 4;
            const double wl = code_w((1u + lit) * 8u);
            double w5[5];
#pragma unroll
            for (int s = 0; s < 5; ++s) {
              const u32 k1 = km1 + 64u * s;
              w5[s] = s_w1[k1 < ke ? k1 : 1u];                  // ([1] = the dead slot k = 2: +inf)
            }
#pragma unroll
            for (int s = 0; s < 5; ++s) {
              const u32 k1 = km1 + 64u * s;
              const double w = k1 == 0 ? wl : w5[s];
              const double mcl = k1 == 0 ? -kInf : mincost;
              DP_RELAX(c[s], l[s], w, mcl)
            }
          }
        } else if (RUNS) {
          // any other row: its codes from the wave's staging area, where 2048 codes of the block's rows are kept from
          // the row's first code on (the rows of consecutive positions are consecutive in codes[]: the copy serves the
          // next rows too — eight of the widest), brought in by LDS-DMA.  One scattered 2-byte global load per register
          // and position, a full memory round trip each time, was what these rows cost before (class Z: the last 257
          // positions of every long run are such rows).
          // (the row's first region and the one behind it are in the ring or on their way — the second is asked for
          //  when the rows enter the first, three or more 258-wide rows before one of them reaches into it: with one
          //  region of 2048 codes, fetched when a row ran over its end, every eighth position waited a full round trip)
          const u32 rg0 = ro >> 10, rg1 = (ro + ke - 1u) >> 10;
          if (!st_ok || st_iss < rg0 || st_iss > rg0 + 2u) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // (nothing still landing where the new regions go)
            st_iss = rg0;
            st_land = rg0;
            st_ok = true;
          }
          if (st_iss < rg0 + 2u) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (this wave's reads of the region that is replaced: served)
          while (st_iss < rg0 + 2u) {                                // (region st_iss takes the half of region st_iss - 2 < rg0)
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
          u32 cd[5];
#pragma unroll
          for (int s = 0; s < 5; ++s) {
            const u32 k1 = km1 + 64u * s;
            cd[s] = s_stage[(ro + (k1 < ke ? k1 : 0u)) & 2047u];
          }
          double wv[5];
#pragma unroll
          for (int s = 0; s < 5; ++s) wv[s] = code_w(cd[s] & 0x3ff8u);
#pragma unroll
          for (int s = 0; s < 5; ++s) {
            const u32 k1 = km1 + 64u * s;
            const double w = k1 < ke ? wv[s] : kInf;
            const double mcl = k1 == 0 ? -kInf : mincost;
            DP_RELAX(c[s], l[s], w, mcl)
          }
        }
        else if (smax < 2) {
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            if ((u32)s <= smax) {
              const u32 k1 = km1 + 64u * s;
              if (k1 < ke) {
                const double w = code_w(rows[ro + k1]);
                const double mcl = k1 == 0 ? -kInf : mincost;
                DP_RELAX(c[s], l[s], w, mcl)
              }
            }
          }
        } else {
          // a long row: all six registers, branch-free, the six code loads in flight together (a lane
          // outside the row reads the row's first code and takes +inf instead of its weight)
          u32 cd[6];
#pragma unroll
          for (int s = 0; s < 6; ++s) {
            const u32 k1 = km1 + 64u * s;
            cd[s] = rows[ro + (k1 < ke ? k1 : 0u)];
          }
#pragma unroll
          for (int s = 0; s < 6; ++s) {
            const u32 k1 = km1 + 64u * s;
            const double w = k1 < ke ? code_w(cd[s]) : kInf;
            const double mcl = k1 == 0 ? -kInf : mincost;
            DP_RELAX(c[s], l[s], w, mcl)
          }
        }
        noshort = false;
        ++n_slow;
        if (fine_) { gq[gk_] += (u64)__builtin_readcyclecounter() - tp0_ + (u64)(__float_as_uint(c[0]) & 0u); ++gq[gk_ + 1]; }
        ++p;
      }
      if (PROF) gt[3] += (u64)__builtin_readcyclecounter() - tg0_ + (u64)(__float_as_uint(c[0]) & 0u);
    }
    if (PROF) { kc[pcls] += (u64)__builtin_readcyclecounter() - tw0; kn[pcls] += n_fast + n_slow - np0; }
    if (jumped) continue;
    // ---- cells wbase .. wbase + 31 are final
    {
      const u32 jj = wbase + lane;
      if (lane < 32 && lane >= skip && jj >= la_lo && jj <= B) put_la(jj, (u16)(l[0] ? jj + 1 - l[0] : 0u));
      skip = 0;
      D4_TRACK_MAX()
      D3_ROT32()
      wbase += 32;
    }
  }
#undef D5_PICK16
#undef D5_CHAIN
#undef D5_CHAIN_I
  if (J.la_lo == 1 && lane == 0) la[0] = 0;   // the head of the block
  if (J.exit) {
#pragma unroll
    for (int s = 0; s < 6; ++s) {
      J.exit->c[64u * s + lane] = c[s];
      J.exit->l[64u * s + lane] = l[s];
      vmax = fmaxf(vmax, c[s] < 1e29f ? c[s] : 0.0f);
    }
    vmax = wave_max_f32(vmax);
    if (lane == 0) { J.exit->vmax = vmax; J.exit->base = wbase; J.exit->noshort = noshort ? 1u : 0u; J.exit->skip = skip; }
  }
  if (PROF && P.prof && lane == 0) {
    u64* o = P.prof + (u64)b * ZMX_PROF_N;
    atomicAdd(&o[1], (u64)__builtin_readcyclecounter() - t_begin);
    atomicAdd(&o[2], n_fast); atomicAdd(&o[3], n_slow); atomicAdd(&o[4], n_fast + n_slow);
    atomicAdd(&o[5], (u64)__builtin_readcyclecounter() - t_begin); atomicAdd(&o[6], n_fast);
    atomicAdd(&o[14], n_slow);
    atomicAdd(&o[10], n_int);
    for (int i = 0; i < 3; ++i) atomicAdd(&o[20 + i], pw[i]);
    for (int i = 0; i < 4; ++i) { atomicAdd(&o[24 + i], kc[i]); atomicAdd(&o[28 + i], kn[i]); }
    atomicAdd(&o[11], pq[0]); atomicAdd(&o[12], pq[1]); atomicAdd(&o[13], pq[2]);
    for (int i = 0; i < 10; ++i) atomicAdd(&o[32 + i], gq[i]);
    for (int i = 0; i < 6; ++i) atomicAdd(&o[42 + i], gr[i]);
    for (int i = 0; i < 4; ++i) atomicAdd(&o[48 + i], gt[i]);
    atomicAdd(&o[52], go[0]); atomicAdd(&o[53], go[1]);
    const u64 dt = (u64)__builtin_readcyclecounter() - t_begin;
    atomicMax(&o[7], dt);                               // the longest task of the block
    if (J.la_lo == 1) atomicAdd(&o[8], dt);             // the head task
  }
}

// ---------------------------------------------------------------------------------------------
// CUT POINTS.  A position q that no DP edge crosses — p + kend(p) <= q for every p < q (kend: the longest
// match at p, or 1 for the literal: dph[p].y) — splits GetBestLengths exactly: every cell beyond q gets its
// value through cell q alone, so a chain started at q from ONE cell is the true chain shifted by a constant,
// with no warm-up and nothing left to chance (a long-run shortcut cannot span a cut either: the run's own
// matches would cross it).  Text has one every ~25 positions (largest gap seen: 688), data made of long runs
// almost none.  One wave per task looks for the last cut point at most `depth` positions before the task's
// pout and makes it the task's start (SegTask.q, which then need not be a multiple of 32: k_dp5_spec starts
// at the window that holds it, with the level in that window's cell); tasks without one keep the warm-up.
// ---------------------------------------------------------------------------------------------
struct CutParams {
  const BlockDesc* blocks;
  const uint2* dph;
  SegTask* tasks;
  u32 depth;
  u32 warm;        // the warm-up a task without a cut point would get
  u32* found;      // [2]: tasks that got a cut point, sum of (pout - q) over them
  u32* wide;       // [tasks]: 1 = no cut point, and the warm-up stretch holds long-run material (see below)
};

// A task that finds no cut point has to rely on coalescence (zmx_dp4.h) over its warm-up stretch.  Inside and
// behind runs of equal bytes that does not happen: the long-run shortcut (squeeze.c:251-271) copies 258 cells
// unmixed, and where every row is 258 edges of one distance the paths run side by side — measured on class Z,
// 90 % of such tasks were re-run serially.  So a task whose warm-up stretch holds a shortcut position or a run row
// of 64+ edges is not started at all: the host merges it into its predecessor (BuildTables), whose own start IS a
// cut point (or the block's head) — tasks become the stretches between cut points there, every one exact by
// construction, and the only thing left to verify is the level.
__global__ __launch_bounds__(64) void k_cutpoints(CutParams P) {
  const u32 t = blockIdx.x;
  const SegTask K = P.tasks[t];
  if (K.pout == 0) return;      // the head of a block starts from the block's true state
  const BlockDesc bd = P.blocks[K.block];
  const uint2* dph = P.dph + bd.pos_off;
  const u32 lane = threadIdx.x;
  const u32 lo = K.pout > P.depth + ZMX_MAX_MATCH ? K.pout - P.depth - ZMX_MAX_MATCH : 0u;
  const u32 wlo = K.pout > P.warm ? K.pout - P.warm : 0u;
  // the prefix maximum from lo on is the true one for q >= lo + 258 (no edge is longer), and for every q if lo = 0
  const u32 q_min = lo == 0 ? 1u : lo + ZMX_MAX_MATCH;
  u32 carry = 0, best = SEG_NONE;
  bool wide = false;
  for (u32 p0 = lo; p0 < K.pout; p0 += 64) {
    const u32 p = p0 + lane;
    const u32 y = p < K.pout ? dph[p].y : 0u;
    const u32 r = p < K.pout ? p + (y & 0xffffu) : 0u;
    wide |= p >= wlo && (((y >> 16) & 1u) != 0 || (((y >> 17) & 1u) != 0 && (y & 0xffffu) >= 64u));
    u32 incl = wave_scan_max(r);
    incl = incl > carry ? incl : carry;
    // q = p + 1 is a cut point; not pout itself: the start cell has no source, and the cells from pout on are
    // compared with the predecessor's, sources included
    const u64 ok = __ballot(p + 1 < K.pout && p + 1 >= q_min && incl <= p + 1);
    if (ok) best = p0 + (63u - (u32)__builtin_clzll(ok)) + 1u;
    carry = rdlane_u32(incl, 63);
  }
  const bool any_wide = __any(wide);
  if (lane == 0) {
    P.wide[t] = best == SEG_NONE && any_wide ? 1u : 0u;
    if (best != SEG_NONE) {
      P.tasks[t].q = best;
      atomicAdd(&P.found[0], 1u);
      atomicAdd(&P.found[1], K.pout - best);
    }
  }
}

// Which of the two k_dp5_spec variants a task is for: 1 = at least a quarter of its 32-position windows are of the
// generic kind (shortcut positions, wide run rows: k_mkdesc), i.e. it walks runs of equal bytes.  One wave per task.
struct TaskKindParams {
  const SegTask* tasks;
  const BlockDesc* blocks;
  const u32* winflag;
  const u32* win_off;
  u32* kind;
};
__global__ __launch_bounds__(64) void k_taskkind(TaskKindParams P) {
  const u32 t = blockIdx.x;
  const SegTask K = P.tasks[t];
  const BlockDesc bd = P.blocks[K.block];
  const u32 B = (u32)(bd.inend - bd.instart);
  const u32* wf = P.winflag + P.win_off[K.block];
  const u32 w0 = K.q >> 5, w1 = ((K.pend < B ? K.pend : B) + 31u) >> 5;
  u32 g = 0;
  for (u32 w = w0 + threadIdx.x; w < w1; w += 64) g += (wf[w] & 0xffu) == 0 || (wf[w] & 0xffu) == 3 ? 1u : 0u;
  g = wave_scan_add(g);
  if (threadIdx.x == 63) P.kind[t] = 4u * g >= (w1 - w0) && g >= 4u ? 1u : 0u;
}

// One workgroup = four waves = up to four tasks of ONE block (P.wg_tasks), sharing the run's weight
// table in LDS; after the table is in place the waves go their own ways.
#define D5_WG 4u
#define FIX_CH 256u       // tasks whose summaries k_dp4_fix holds in LDS at a time
template <bool PROF, int WAVES, bool RUNS>
__global__ __launch_bounds__(64 * D5_WG, WAVES) void k_dp5_spec(Dp4Params P) {
  __shared__ __align__(16) double s_wtab[ZMX_WTAB];
  // per wave: the staged codes of a window (4 KB + the alignment slack), or the cells of a long-run shortcut
  __shared__ __align__(16) unsigned char s_buf[D5_WG][D5_STAGE_BYTES];
  __shared__ __align__(8) uint2 s_itab[ZMX_WTAB];
  __shared__ __align__(8) double s_w1[D5_W1];
  __shared__ u8 s_sym1[D5_W1];
  __shared__ __align__(8) uint2 s_ri[D5_WG][32];
  __shared__ __align__(8) uint2 s_rk[RUNS ? D5_WG : 1u][RUNS ? D5_RKN : 1u];
  __shared__ u32 s_rmax;
  const u32 wave = (u32)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (P.redo_pass && blockIdx.x >= *P.redo_count) return;
  const u32* wg = (P.redo_pass ? P.redo_wg : P.wg_tasks) + (u64)(P.task0 + blockIdx.x) * D5_WG;
  const u32 t0 = wg[0];
  if (P.redo_pass && P.coop && P.kind[t0] != 0) return;      // a run task: k_dp6_spec's (zmx_dp6.h)
  const u32 b0 = P.tasks[t0].block;
  for (u32 i = threadIdx.x; i < ZMX_WTAB; i += 64 * D5_WG) s_wtab[i] = P.wtab[(u64)b0 * ZMX_WTAB + i];
  if (threadIdx.x == 0) s_rmax = 0;
  __syncthreads();
  d5_build_w1(s_wtab, s_w1, s_sym1);
  // the level a speculative task starts from
  auto task_level = [&](u32 tt) -> float {
    const SegTask K = P.tasks[tt];
    float lv = P.est_bits ? P.est_bits[K.block] * ((float)K.q / (float)(u32)(P.blocks[K.block].inend - P.blocks[K.block].instart)) : P.lvl[tt];
    if (!P.redo_pass) lv *= P.level_scale;
    return lv >= 16.0f ? lv : 16.0f;
  };
  // the binade of the workgroup's integer table (zmx_dp5.h, D5IntTab): that of its first speculative task
  D5IntTab IT;
  IT.on = false; IT.lo = 0; IT.span = 0;
  {
    const u32 tl = P.tasks[t0].pout != 0 ? t0 : wg[1];
    if (tl != SEG_NONE && P.int_path) {
      const u32 lb = __float_as_uint(task_level(tl));
      const int e = (int)(lb >> 23) - 127;
      if (e >= 4 && e < 31 && ((P.tiemask[b0] >> (e & 31)) & 1u) == 0) {
        d5_build_inttab(s_wtab, s_itab, s_rmax, e);
        __syncthreads();
        const u32 rm = s_rmax;
        IT.on = true;
        IT.lo = lb & 0x7f800000u;
        IT.span = rm < 0x100000u ? 33u * rm : 0x40000000u;
      }
    }
  }
  const u32 t = wg[wave];
  if (t == SEG_NONE) return;
  const SegTask T = P.tasks[t];
  const BlockDesc bd = P.blocks[T.block];
  const u32 B = (u32)(bd.inend - bd.instart);
  if (B == 0) return;
  D4Job J;
  J.start = T.q & ~31u;       // windows lie at multiples of 32 from the block start
  J.cell = T.q & 31u;
  J.noshort = 0;
  J.pout = T.pout;
  J.pend = T.pend;
  J.load = false;
  J.delta = 0;
  J.init = nullptr;
  J.entry = &P.entry[t];
  J.exit = &P.exit[t];
  // (round 3: run tasks only — the long ones, tens of thousands of positions between two cut points; round 5: text tasks
  //  too: a block has a binade boundary per doubling of its cost whatever its size, and in a SMALL call the serial re-runs
  //  of the tasks that cross one were most of k_dp4_fix: 0.66 ms per run of a 1 MB call)
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
    J.level = task_level(t);
    if (P.est_bits && (threadIdx.x & 63) == 0) P.lvl[t] = J.level;
  }
  d5_run_job<PROF, RUNS>(P, J, T.block, bd, s_wtab, reinterpret_cast<float*>(s_buf[wave]), reinterpret_cast<u16*>(s_buf[wave] + 4u * DP_XN),
                   reinterpret_cast<u16*>(s_buf[wave]), s_itab, IT, s_w1, s_sym1, s_ri[wave], s_rk[RUNS ? wave : 0u]);
}

// the cooperative run-task job (four waves a task): d6_run_job, k_dp6_spec — measured 10 % slower than the one-wave job
// (round 5), so only in -DZMX_EXPERIMENTS builds
#ifdef ZMX_EXPERIMENTS
#include "zmx_dp6.h"
#endif

// ---------------------------------------------------------------------------------------------
// FIX: one workgroup per block walks the block's tasks in order, accepts every task whose entry
// state is the true state up to a shift that keeps the task inside its binade, and runs the others
// again from the true state — the serial chain, for exactly the stretches that need it.  A re-run uses
// k_dp4's four-wave pipeline (d4_run_job: 57 cycles per position on text, but a ring restart and two
// bubble steps per long-run shortcut and 14 positions per step where rows are 258 wide), or, for a
// task with at least P.fix_lean_min windows of the generic kind (shortcut flags, long rows), the lean
// one-wave job of k_dp5_spec (d5_run_job in load mode: a shortcut is a handful of LDS operations).
// ---------------------------------------------------------------------------------------------
template <bool PROF>
__global__ __launch_bounds__(64 * (D3_NB + 2)) void k_dp4_fix(Dp4Params P) {
  D4_LDS_DECL
  const u32 b = P.block0 + blockIdx.x;
  const BlockDesc bd = P.blocks[b];
  const u32 B = (u32)(bd.inend - bd.instart);
  if (B == 0) return;
  const u32 t0 = P.task_off[b], t1 = P.task_off[b + 1];
  const u32 lane = threadIdx.x & 63;
  const bool lead = threadIdx.x == 0;
  const double wmax = (double)P.wmax[b] + 1.0;
  const u32 tiemask = P.tiemask[b];
  for (u32 i = threadIdx.x; i < ZMX_WTAB; i += blockDim.x) s_wtab[i] = P.wtab[(u64)b * ZMX_WTAB + i];
  __syncthreads();
  __shared__ __align__(8) double s_w1[D5_W1];
  __shared__ u8 s_sym1[D5_W1];
  __shared__ __align__(8) uint2 s_ri[32];
  d5_build_w1(s_wtab, s_w1, s_sym1);
  u16* la_block = P.la + bd.la_off;
  d4_copy_over(P, t0, B, la_block);   // the head is exact
  double delta_prev = 0.0;     // what has to be added to exit[t - 1] to get the true values
  bool rerun_prev = false;     // exit[t - 1] was rewritten by this workgroup: P.chk[t] is stale
  u32 n_ok = 0, n_state = 0, n_level = 0, n_tie = 0, n_pos = 0, n_values = 0, n_lean = 0, n_mid = 0;
  const u32* winflag = P.winflag + P.win_off[b];
  const u64 cyc0 = __builtin_readcyclecounter();
  u64 cyc_run = 0;
  // The walk reads a few words per task and almost always just accepts it: with a dependent global load or two
  // per task, 240 tasks a block, the walk — not the re-runs — was most of this kernel's 0.75 ms.  So: the tasks'
  // summaries come into LDS FIX_CH at a time in one sweep, and what an accepted task leaves to do (its overshoot
  // cells into length_array, its level for the next run) is done for the whole chunk afterwards, all threads at once.
  __shared__ double s_ck_d[FIX_CH], s_dl[FIX_CH];
  __shared__ float s_ck_vmin[FIX_CH], s_vmax[FIX_CH];
  __shared__ u32 s_ck_match[FIX_CH], s_xbase[FIX_CH], s_pend[FIX_CH], s_acc[FIX_CH];
  __shared__ double s_wsum[4], s_dprev;
  __shared__ u32 s_wfail[4];
  static_assert(FIX_CH == 64 * (D3_NB + 2), "the sweep judges a chunk's tasks a thread each");
  for (u32 c0 = t0 + 1; c0 < t1; c0 += FIX_CH) {
  const u32 cn = t1 - c0 < FIX_CH ? t1 - c0 : FIX_CH;
  for (u32 i = threadIdx.x; i < cn; i += blockDim.x) {
    const SegCheck k = P.chk[c0 + i];
    s_ck_d[i] = k.d; s_ck_vmin[i] = k.vmin; s_ck_match[i] = k.match;
    s_vmax[i] = P.exit[c0 + i].vmax;
    s_xbase[i] = P.exit[c0 + i].base + P.exit[c0 + i].skip;   // (where the task's own cells end: d4_copy_over)
    s_pend[i] = P.tasks[c0 + i].pend;
    s_acc[i] = 0;
    s_dl[i] = 0.0;
  }
  __syncthreads();
  for (u32 t = c0; t < c0 + cn; ++t) {
    // The walk, a SWEEP at a time (round 5): nearly every task is simply accepted, and deciding so one task after the
    // other — four waves doing the same dozen double-precision operations — was 1 200 cycles a task, 0.25 ms of this
    // kernel's 0.43 per 1 MB block.  All tasks of the chunk from t on are judged at once, a thread each: their shifts are a
    // prefix sum of the checks' differences (sums of multiples of float ulps: exact in any order, as in k_dpscan), the
    // acceptance test is the one below with that shift; everything up to the first task that fails is accepted, and
    // that task takes the serial step — which re-runs it — as before.  (Not behind a re-run: the next task's check is stale.)
    if (!rerun_prev) {
      const u32 ti0 = t - c0;
      const u32 i = threadIdx.x;                       // (FIX_CH = the workgroup's threads: a thread per task of the chunk)
      const bool act = i >= ti0 && i < cn;
      double incl = act ? s_ck_d[i] : 0.0;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const double up = __shfl_up(incl, o, 64);
        if ((int)lane >= o) incl += up;
      }
      if (lane == 63) s_wsum[threadIdx.x >> 6] = incl;
      __syncthreads();
      double before = 0.0;
      for (u32 w = 0; w < (threadIdx.x >> 6); ++w) before += s_wsum[w];
      const double delta_i = delta_prev + before + incl;
      const bool ok_i = act && s_ck_match[i] == 1 && d4_accept(s_ck_vmin[i], (double)s_vmax[i], delta_i, wmax, tiemask) == 0;
      const u64 failm = __ballot(act && !ok_i);
      if (lane == 0) s_wfail[threadIdx.x >> 6] = failm ? (threadIdx.x & ~63u) + (u32)__ffsll((long long)failm) - 1u : 0xffffffffu;
      __syncthreads();
      u32 first_fail = cn;
      for (u32 w = 0; w < (blockDim.x >> 6); ++w) first_fail = s_wfail[w] < first_fail ? s_wfail[w] : first_fail;
      if (act && i < first_fail) { s_acc[i] = 1; s_dl[i] = delta_i; }
      if (first_fail > ti0 && i == first_fail - 1u) s_dprev = delta_i;
      __syncthreads();
      if (first_fail > ti0) { delta_prev = s_dprev; n_ok += first_fail - ti0; }
      t = c0 + first_fail;
      if (t >= c0 + cn) break;
    }
    const u32 ti = t - c0;
    SegCheck ck;
    if (rerun_prev) ck = d4_check(&P.exit[t - 1], &P.entry[t], lane);   // every wave computes the same
    else { ck.d = s_ck_d[ti]; ck.vmin = s_ck_vmin[ti]; ck.match = s_ck_match[ti]; }
    const double delta = delta_prev + ck.d;
    // the guess to start the next run of this task from
    if (lead) s_dl[ti] = delta;
    bool ok = ck.match == 1;
    u32 why = 0;
    if (ok) {
      why = d4_accept(ck.vmin, (double)s_vmax[ti], delta, wmax, tiemask);
      ok = why == 0;
    }
    if (P.debug == 6 && ck.match == 0 && threadIdx.x < 64) {
      // where the two states differ: the first cell whose reach, source or value shape is not the same
      const SegSnap* E = &P.exit[t - 1];
      const SegSnap* N = &P.entry[t];
      u32 first = 0xffffffffu;
      for (int s6 = 0; s6 < 6; ++s6) {
        const u32 i = 64u * s6 + lane;
        const bool ef = E->c[i] < 1e29f, nf = N->c[i] < 1e29f;
        if ((ef != nf || E->l[i] != N->l[i]) && i < first) first = i;
      }
      for (int o = 32; o >= 1; o >>= 1) { const u32 v = __shfl_xor(first, o, 64); first = v < first ? v : first; }
      if (lane == 0) {
        const u32 i = first < SEG_CELLS ? first : 0;
        printf("diff b %u t %u pout %u base %u/%u noshort %u/%u skip %u/%u first cell %u: exit c %.4f l %u | entry c %.4f l %u | q %u\n", b, t - t0, P.tasks[t].pout,
               E->base, N->base, E->noshort, N->noshort, E->skip, N->skip, first, (double)E->c[i], E->l[i], (double)N->c[i], N->l[i], P.tasks[t].q);
      }
    }
    if (P.debug == 1 && lead) {
      printf("fix b %u t %u pout %u: match %u d %.6f delta %.6f vmin %.4f vmax %.4f why %u ok %d entry base %u exit-1 base %u\n", b,
             t - t0, P.tasks[t].pout, ck.match, ck.d, delta, (double)ck.vmin, (double)P.exit[t].vmax, why, ok ? 1 : 0,
             P.entry[t].base, P.exit[t - 1].base);
    }
    if (ok) {
      if (lead) s_acc[ti] = 1;
      delta_prev = delta;
      rerun_prev = false;
      ++n_ok;
      continue;
    }
    if (ck.match == 0) ++n_state; else if (ck.match == 2) ++n_values; else if (why == 1) ++n_level; else ++n_tie;
    const SegTask T = P.tasks[t];
    D4Job J;
    J.start = P.exit[t - 1].base;
    J.noshort = P.exit[t - 1].noshort;
    J.pout = 0;
    J.pend = T.pend;
    J.over_lo = SEG_NONE;    // (this workgroup is the only writer of the block's length_array now)
    J.spec = false;
    J.load = true;
    J.level = 0.0f;
    J.delta = delta_prev;
    J.init = &P.exit[t - 1];
    J.entry = nullptr;
    J.exit = &P.exit[t];
    J.over = nullptr;
    // The task matched its predecessor and only grew out of its binade: if what it did up to its mid snapshot passes the
    // test (the same test, with the largest source value of that prefix), the prefix stands — its lengths are in
    // length_array already — and the re-run starts at the snapshot, from the task's own cells plus the shift.
    bool from_mid = false;
    if (ck.match == 1 && why == 1 && P.mid != nullptr) {
      const u32 mb_ = P.mid[t].base;
      if (mb_ != SEG_NONE && mb_ > P.entry[t].base && mb_ < (T.pend < B ? T.pend : B) &&
          d4_accept(ck.vmin, (double)P.mid[t].vmax, delta, wmax, tiemask) == 0) {
        from_mid = true;
        J.start = mb_;
        J.noshort = P.mid[t].noshort;
        J.delta = delta;
        J.init = &P.mid[t];
        ++n_mid;
      }
    }
    J.la_lo = J.start;
    n_pos += (T.pend < B ? T.pend : B) - (J.start < B ? J.start : B);
    // windows of the task that k_dp5_spec's fast paths cannot take (k_mkdesc)
    const u32 w0 = J.start >> 5, w1 = ((T.pend < B ? T.pend : B) + 31u) >> 5;
    u32 generic = 0, mine = 0, nocodes = 0;
    for (u32 w = w0 + threadIdx.x; w < w1; w += blockDim.x) {
      generic += (winflag[w] & 0xffu) == 0 ? 1u : 0u;
      nocodes |= winflag[w] & D5_WF_CODELESS;
      ++mine;
    }
    // (the barriers also mean: every wave has read the old exit[t] / exit[t - 1])
    // (automatic, P.fix_lean_min < 0: the lean job where most of the windows are of the generic kind — runs of equal
    //  bytes: its run-row path reads no codes, the pipeline's ring would restart at every shortcut.  Counted per
    //  thread: a merged task has thousands of windows)
    const int n_generic = __syncthreads_count((int)generic);
    const int n_major = __syncthreads_count(mine > 0 && 2 * generic >= mine ? 1 : 0);
    const int n_have = __syncthreads_count(mine > 0 ? 1 : 0);
    const int n_nocodes = __syncthreads_count(nocodes != 0 ? 1 : 0);   // (a wide run row without codes: only the lean job's tables know its weights)
    // (a predecessor that stopped inside a shortcut's window — skip — can only be continued by the job that knows
    //  about it)
    const bool lean = (!from_mid && P.exit[t - 1].skip != 0) || n_nocodes > 0 || (P.fix_lean_min >= 0 ? n_generic >= P.fix_lean_min : 2 * n_major >= n_have);
    const u64 cr0 = __builtin_readcyclecounter();
    if (lean) {
      ++n_lean;
      D5IntTab IT;
      IT.on = false; IT.lo = 0; IT.span = 0;
      if (threadIdx.x < 64) {
        d5_run_job<PROF, true>(P, J, b, bd, s_wtab, s_xc, s_xl, s_ring, *reinterpret_cast<const uint2 (*)[ZMX_WTAB]>(&s_t1[0][0]), IT, s_w1, s_sym1, s_ri,
                               reinterpret_cast<uint2*>(&s_t2[0][0]));   // (s_rk: the pipeline's tiles are idle in this job)
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      __syncthreads();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    } else {
      d4_run_job<PROF>(P, J, b, bd, s_ring, s_wtab, s_t1, s_t2, s_tab, s_desc, s_tabc, s_xc, s_xl, s_lout);
    }
    cyc_run += __builtin_readcyclecounter() - cr0;
    delta_prev = 0.0;
    rerun_prev = true;
  }
  __syncthreads();
  // the chunk's accepted tasks: their cells beyond pend (d4_copy_over), one task per wave at a time; and every
  // task's level for the next run
  // (sixteen lanes a task, sixteen tasks at a time: a task's copy is one or two dependent global round trips of a few dozen
  //  cells, and with a wave per task — four in flight — those round trips were a third of this kernel on text)
  for (u32 i = threadIdx.x >> 4; i < cn; i += blockDim.x >> 4) {
    if (!s_acc[i]) continue;
    const u32 pend = s_pend[i], stop = s_xbase[i];
    if (pend > B) continue;                   // the last task of the block runs to the end
    const u16* over = P.over + (u64)(c0 + i) * SEG_OVER;
    for (u32 k = threadIdx.x & 15u; pend + k < stop && pend + k <= B; k += 16) la_block[pend + k] = over[k];
  }
  for (u32 i = threadIdx.x; i < cn; i += blockDim.x) P.lvl[c0 + i] = (float)((double)P.lvl[c0 + i] + s_dl[i]);
  __syncthreads();
  }
  if (P.debug >= 2 && lead) {
    printf("fix b %u: %u tasks ok %u state %u values %u level %u (%u of them from their mid snapshot) tie %u lean %u, %u positions re-run, %llu cycles in all, %llu in re-runs\n", b,
           t1 - t0, n_ok, n_state, n_values, n_level, n_mid, n_tie, n_lean, n_pos, (unsigned long long)(__builtin_readcyclecounter() - cyc0),
           (unsigned long long)cyc_run);
  }
  if (lead && P.stats) {
    atomicAdd(&P.stats[0], t1 - t0);
    atomicAdd(&P.stats[1], n_ok);
    atomicAdd(&P.stats[2], n_state);
    atomicAdd(&P.stats[3], n_level);
    atomicAdd(&P.stats[4], n_tie);
    atomicAdd(&P.stats[5], n_pos);
    atomicAdd(&P.stats[6], n_values);
    atomicAdd(&P.stats[7], n_lean);
  }
}

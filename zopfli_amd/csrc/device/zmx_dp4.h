// k_dp4: GetBestLengths (squeeze.c:217-309) of one LZ77OptimalRun, cut into TASKS that run on all
// CUs at once, with every task's result verified before it is used.  Included only by zmx_hip.hip,
// after zmx_kernels.h.  DESIGN.md section 4 has the reasoning and the measurements.
//
// The DP is a chain through float-rounded absolute costs (squeeze.c:243,281,299): position j + 1
// needs costs[j + 1], which position j may just have written.  One wave per block walking a million
// dependent positions was the whole launch time of round 1.  Two facts cut the chain:
//
//  * Translation.  While every value a stretch of the chain computes lies in one binade
//    [2^e, 2^(e+1)), adding D (a multiple of the float ulp of that binade) to its entry state adds D
//    to every value and changes no comparison: dbl(w + c + D) = dbl(w + c) + D because c and D are
//    multiples of 2^(e-23), an even multiple of the double ulp 2^(e-52), and fl(x + D) = fl(x) + D
//    unless x lies exactly half-way between two floats — which depends only on the edge weight w
//    and e and is excluded per block and binade up front (zmx_hip.hip, tie mask).
//  * Coalescence.  The state of the chain at position p (the <= 259 live cells, their values up
//    to a common shift and the edges that reached them) is the same whether the chain started at 0
//    or a few hundred positions before p from a single cell: all surviving paths pass through a
//    common ancestor shortly before p.  That is a property of the data, not a theorem.
//
// So a block is cut into tasks of SEG_L positions.  Task s > 0 starts SEG_WARM positions early from
// a single cell holding a guessed level, records its state when it reaches its first group at or
// after pout ("entry"), writes length_array from there on and records its state where it stops
// ("exit").  k_dpcheck compares exit[s - 1] with entry[s] cell by cell: same group base, same
// source edges, values differing by one constant d.  k_dp4<FIX> (one workgroup per block) then
// walks the tasks in order: the true shift of task s is the true shift of s - 1 plus d; if the
// states matched and both the guessed and the shifted values of the task stay inside one binade
// whose weights cannot tie, the task's length_array is exactly what the serial chain would have
// written.  Any task that fails a test is run again from the true exit state of its predecessor
// (no guess, no warm-up), which is the serial chain itself.  Nothing unverified is ever used.
//
// The speculative pass over all tasks is k_dp5_spec (zmx_dp5.h: one wave per task, built for
// throughput).  The serial re-runs of k_dp4_fix below use the round-1 k_dp3 pipeline, built for the
// latency of ONE chain: one workgroup of four waves (one per SIMD), with one
// s_barrier per STEP (a run of positions of one 64-position group whose edge rows span at most
// D3_SPAN ring slots; where a step ends depends only on dph[], never on DP values):
//
//   wave 1 (the walk and the ring)   walks step i: reads dph[] and k_badscan's bad-edge bitmap,
//                       publishes a 6-word descriptor and the group's {row offset, kend}; keeps the
//                       32 KB LDS row ring filled by LDS-DMA for step i - 1; turns the lengths the
//                       chain wave left in LDS for step i - 3 into length_array.
//   waves 2..3 (tiles)  build the tile of step i - 1: ready-made 64-lane rows (+inf outside the
//                       row, lane = cell of the 32-cell window) in a double-buffered LDS tile.
//   wave 0 (the chain)  runs step i - 2: cells in registers (lane l owns cells w + 64 s + l of a
//                       window that moves 32 cells at a time), 8 VALU instructions per position on
//                       the usual path; positions that need more than two cell registers, flagged
//                       or bad-edge positions, ragged tails and shortcuts take the generic path
//                       straight from the ring.
//
// After a shortcut (and at the start) the ring restarts at a new place: the walk inserts two
// bubble steps so that nobody reads the ring while wave 1 primes it.
#pragma once

// On the fast paths the mincost test of squeeze.c:293 is provably a no-op: k_wtab verifies that
// every match edge of the position costs at least mincost (w >= mincost); rounding is monotone,
// so fl(w + cj) >= fl(mincost + cj), hence "newCost < costs[j+k]" already implies
// "costs[j+k] > mincostaddcostj".  A position with an edge below mincost (possible only through
// rounding in the cost model) is reported in k_badscan's bitmap and takes the generic path, which
// tests literally.
//
// The hot form.  The source is recorded as a small constant K (1 + index of the position in its
// block): the select takes an inline constant.  The new cost goes through v_min_f64 instead of
// compare + select: (float)min(nc, (double)c) is (float)nc when nc < c and c itself otherwise
// (exactly: c is a float), and it keeps the cost chain clear of the slow VALU -> VCC -> VALU path.
#define D3_RELAX_K(CS, LT, WV, K)                                            \
  {                                                                          \
    const double old_ = (double)(CS);                                        \
    const double nc_ = (WV) + cj;                                            \
    LT = nc_ < old_ ? (K) : LT;                                              \
    CS = (float)fmin(nc_, old_);                                             \
  }

// Move the window of cell registers 32 cells on.
#define D3_ROT32()                                                           \
  {                                                                          \
    if (reach < 64) {              /* only register 0 holds anything */      \
      c[0] = __uint_as_float(__builtin_amdgcn_permlane32_swap(__float_as_uint(c[0]), __float_as_uint(1e30f), false, false)[1]); \
      l[0] = __builtin_amdgcn_permlane32_swap(l[0], 0u, false, false)[1];    \
    } else {                                                                 \
      d3_rot32(c, l, lane);                                                  \
    }                                                                        \
    reach = reach >= 32 ? reach - 32 : 0;                                    \
  }

// The largest finite value this task has held in cell register 0 (per lane; reduced at the end).
#define D4_TRACK_MAX() { vmax = fmaxf(vmax, c[0] < 1e29f ? c[0] : 0.0f); }

// Cells WB .. WB + 31 (lanes 0..31 of register 0) are final: write their lengths and move the
// window on.  Cells below la_lo belong to the previous task.
#define D3_RETIRE32(WB)                                                      \
  {                                                                          \
    const u32 jj_ = (WB) + lane;                                             \
    if (lane < 32 && jj_ >= la_lo && jj_ <= B) la[jj_] = (u16)(l[0] ? jj_ + 1 - l[0] : 0u); \
    /* wave 1 stores the lengths of a clean-flagged step from s_lout whichever path ran it */ \
    if ((WB) - base <= 32u) s_lout[it & 1][((WB) - base) * 2 + lane] = l[0]; \
    D3_ROT32()                                                               \
  }

#define D3_NB 2u         // tile-building waves (waves 2..); wave 0 = the chain, wave 1 = the walk and the ring
#define D3_SPAN 3584u    // rows of four consecutive steps fit in the ring (4 * (3584 + 511 of alignment) <= 16384);
                         // a row has up to 258 slots: at least 13 positions per step
#define D3_DESC_CLEAN (1u << 25)   // descriptor word 0: a whole group, no flagged / two-register / bad-edge position
#define D3_EV_NONE 0u
#define D3_EV_SHORTCUT 1u
#define D3_EV_GROUP_END 2u
#define D3_EV_BUBBLE 3u  // nothing to do
#define D3_EV_PRIME 4u   // wave 1 primes the ring at the start of the segment that follows

#define SEG_CELLS 384u   // the six cell registers of the chain wave
#define SEG_OVER 384u    // a task stops at the first window base >= pend: less than 64 + 258 cells beyond pend
#define SEG_NONE 0xffffffffu

// The state of the chain between two groups: the window of cell registers sits at `base`.
struct SegSnap {
  u32 base;              // block-relative position of the next group
  u32 noshort;           // squeeze.c:273: the shortcut is not tested at `base` (it was just taken)
  float vmax;            // exit snapshots: the largest finite cell value the task produced from its entry on
  u32 skip;              // the first `skip` positions of the window at `base` lie inside a long-run shortcut's span (zmx_dp5.h)
  float c[SEG_CELLS];    // cell values (1e30 = never reached)
  u32 l[SEG_CELLS];      // 1 + block-relative position the cell was reached from (0 = never)
};

struct SegTask {
  u32 block;             // block index in the table set
  u32 q;                 // first position walked: pout - warm-up (0 for the first task of a block)
  u32 pout;              // length_array is written from the first group base >= pout on
  u32 pend;              // the walk stops at the first group base >= pend (B + 1: runs to the block end)
};

struct SegCheck {        // k_dpcheck: exit[t - 1] against entry[t]
  double d;              // exit[t - 1].c - entry[t].c (the same for every reached cell when match is set)
  float vmin;            // smallest cell value of entry[t]
  u32 match;             // 1: same base, same shortcut state, same reached cells, same sources, one constant difference;
                         // 0: the structure differs; 2: only the values differ (by more than one constant)
};

struct D3Group {          // the current group: lane l = position base + l
  u32 roff, kend, offend; // kend = 0 beyond the block end
  u64 m_short, m_r1, m_bad;   // m_bad: positions that cannot be in a fast block
  u32 navail;
};

struct D3Walk {           // wave-uniform walker state (+ the per-lane dph prefetch)
  u32 base = 0, q = 0;
  bool noshort = false, have_group = false;
  u32 bubbles = 2;        // the walk starts with BUBBLE, PRIME
  // dph of the next group, requested one group ahead.  Two register sets used alternately: with
  // one, the compiler parks the new load in a temporary and copies it over at once — a full
  // global-load stall per group.
  u32 pf_base = 0xffffffffu;
  u32 pf_sel = 0;
  uint2 pf_a = make_uint2(0, 0), pf_b = make_uint2(0, 0);
  u32 pf_wa = 0, pf_wb = 0;   // the word of k_badscan's bad-edge bitmap that holds the lane's position
};

struct D3Step {
  u32 base, q, n, event, a_cur;
};

__device__ __forceinline__ void d3_load_group(D3Walk& W, D3Group& G, const uint2* dbase, const u32* badpos,
                                              u32 pos_off, u32 B, u32 lane) {
  const u32 jj = W.base + lane;
  G.navail = (B - W.base < 64u) ? B - W.base : 64u;   // W.base <= B
  const bool act = lane < G.navail;
  const u32 cur = jj < B ? jj : B - 1, nxt = jj + 64 < B ? jj + 64 : B - 1;
  uint2 dh;
  u32 bw;
  if (W.pf_sel == 0) {
    dh = W.pf_a; bw = W.pf_wa;
    if (W.pf_base != W.base) { dh = dbase[cur]; bw = badpos[(pos_off + cur) >> 5]; }
    W.pf_b = dbase[nxt];
    W.pf_wb = badpos[(pos_off + nxt) >> 5];
  } else {
    dh = W.pf_b; bw = W.pf_wb;
    if (W.pf_base != W.base) { dh = dbase[cur]; bw = badpos[(pos_off + cur) >> 5]; }
    W.pf_a = dbase[nxt];
    W.pf_wa = badpos[(pos_off + nxt) >> 5];
  }
  W.pf_sel ^= 1;
  W.pf_base = W.base + 64;
  G.kend = act ? (dh.y & 0xffffu) : 0u;
  G.roff = dh.x;
  G.offend = G.roff + G.kend;
  G.m_short = __ballot(act && ((dh.y >> 16) & 1u) != 0);
  // the chain wave re-bases its cell registers every 32 positions (see the consumer): a position
  // sits in lane (lane & 31) of the window, its edges reach cell register (kend + (lane & 31)) >> 6
  G.m_r1 = __ballot(G.kend + (lane & 31u) >= 64u);                    // needs cell register 1
  // not for the fast path: flagged, more than two registers, two registers in rows 0..31 (tile 2
  // holds rows 32..63 only), or a match edge below mincost (k_badscan's bitmap; see D3_RELAX_K)
  G.m_bad = G.m_short | __ballot(G.kend + (lane & 31u) >= 128u) | (G.m_r1 & 0xffffffffull) |
            __ballot(act && ((bw >> ((pos_off + cur) & 31u)) & 1u) != 0);
  W.have_group = true;
}

// The next step of the walk (wave 1 only; the other waves read its description from LDS).
__device__ __forceinline__ D3Step d3_next(D3Walk& W, D3Group& G, const uint2* dbase, const u32* badpos, u32 pos_off,
                                          u32 B, u32 lane) {
  D3Step S;
  if (W.bubbles) {
    S.base = W.base; S.q = 0; S.n = 0; S.a_cur = 0;
    S.event = W.bubbles == 2 ? D3_EV_BUBBLE : D3_EV_PRIME;
    --W.bubbles;
    if (S.event == D3_EV_PRIME) {
      // the segment that follows starts at W.base: load its group now; the ring is primed from its first row
      if (!W.have_group && W.base <= B) { d3_load_group(W, G, dbase, badpos, pos_off, B, lane); W.q = 0; }
      S.a_cur = rdlane_u32(G.roff, 0) & ~(DP_PIECE - 1);
    }
    return S;
  }
  if (!W.have_group) { d3_load_group(W, G, dbase, badpos, pos_off, B, lane); W.q = 0; }
  u64 ms = W.q < 64 ? G.m_short & ~((1ull << W.q) - 1) : 0ull;
  if (W.noshort) ms &= ~(1ull << W.q);               // squeeze.c:273: not tested again right after a shortcut
  const u32 stop = ms ? (u32)__ffsll((long long)ms) - 1 : 64u;
  const u32 limit = stop < G.navail ? stop : G.navail;
  S.base = W.base; S.q = W.q; S.n = 0; S.a_cur = 0;
  if (W.q < limit) {
    S.a_cur = rdlane_u32(G.roff, W.q) & ~(DP_PIECE - 1);
    const u64 fit = __ballot(lane >= W.q && lane < limit && G.offend - S.a_cur <= D3_SPAN);
    S.n = (u32)__popcll(fit);
  }
  if (ms && W.q + S.n == stop) {                     // a flagged position follows
    S.event = D3_EV_SHORTCUT;
    W.base = W.base + stop + ZMX_MAX_MATCH;
    W.noshort = true;
    W.have_group = false;
    W.bubbles = 2;
  } else if (W.q + S.n == G.navail) {
    S.event = D3_EV_GROUP_END;
    W.base += 64;
    if (S.n) W.noshort = false;
    W.have_group = false;
  } else {
    S.event = D3_EV_NONE;
    W.q += S.n;
    W.noshort = false;
  }
  return S;
}

// Move six cell registers 32 lanes down: new x[s] = { x[s] lanes 32..63, x[s+1] lanes 0..31 }.
// v_permlane32_swap(a, b) exchanges lanes 32..63 of a with lanes 0..31 of b: one swap and one
// select per register.
__device__ __forceinline__ void d3_rot32_u(u32 (&x)[6], u32 fresh, bool lo) {
  u32 t = x[0];
#pragma unroll
  for (int s = 0; s < 5; ++s) {
    const auto r = __builtin_amdgcn_permlane32_swap(t, x[s + 1], false, false);   // {t.lo | x.lo}, {t.hi | x.hi}
    x[s] = lo ? r[1] : r[0];
    t = r[1];
  }
  const auto r = __builtin_amdgcn_permlane32_swap(t, fresh, false, false);
  x[5] = r[1];
}

__device__ __forceinline__ void d3_rot32(float (&c)[6], u32 (&l)[6], u32 lane) {
  const bool lo = lane < 32;
  u32 cu[6];
#pragma unroll
  for (int s = 0; s < 6; ++s) cu[s] = __float_as_uint(c[s]);
  d3_rot32_u(cu, __float_as_uint(1e30f), lo);
#pragma unroll
  for (int s = 0; s < 6; ++s) c[s] = __uint_as_float(cu[s]);
  d3_rot32_u(l, 0u, lo);
}

__device__ __forceinline__ bool d3_fast(u32 end, u64 m_bad, u32 p0) {
  return p0 + 8 <= end && (p0 & 31u) <= 24u && ((u32)(m_bad >> p0) & 255u) == 0;   // inside one window
}

__device__ __forceinline__ float wave_max_f32(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float wave_min_f32(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
  return v;
}

// exit state E of a task against the entry state N of the next one, by one wave
__device__ __forceinline__ SegCheck d4_check(const SegSnap* E, const SegSnap* N, u32 lane) {
  bool bad = E->base != N->base || E->noshort != N->noshort || E->skip != N->skip;
  bool badv = false;
  double dl[6];
  bool fin[6];
  float vmin = 3e38f;
#pragma unroll
  for (int s = 0; s < 6; ++s) {
    const u32 i = 64u * s + lane;
    const float ec = E->c[i], nc = N->c[i];
    const bool ef = ec < 1e29f, nf = nc < 1e29f;
    bad |= ef != nf || E->l[i] != N->l[i];
    fin[s] = ef && nf;
    dl[s] = (double)ec - (double)nc;          // exact in double
    if (nf) vmin = fminf(vmin, nc);
  }
  // the cell of the next position — `base`, or base + skip behind a long-run shortcut — is always reached
  const u32 ref = (u32)__builtin_amdgcn_readfirstlane((int)(E->skip < 32u ? E->skip : 0u));
  const u32 lo = (u32)__builtin_amdgcn_readlane((int)(u32)__double_as_longlong(dl[0]), ref);
  const u32 hi = (u32)__builtin_amdgcn_readlane((int)(u32)(__double_as_longlong(dl[0]) >> 32), ref);
  const double d0 = __longlong_as_double((long long)(((u64)hi << 32) | lo));
  const bool fin0 = __builtin_amdgcn_readlane(fin[0] ? 1 : 0, ref) != 0;
#pragma unroll
  for (int s = 0; s < 6; ++s) badv |= fin[s] && dl[s] != d0;
  SegCheck r;
  r.d = d0;
  r.vmin = wave_min_f32(vmin);
  r.match = (__any(bad) || !fin0) ? 0u : __any(badv) ? 2u : 1u;
  return r;
}

struct Dp4Params {
  const BlockDesc* blocks;
  u32 block0;              // FIX: first block of this launch
  u32 task0;               // SPEC / k_dpcheck: first task of this launch
  const uint2* dph;
  const double* cost;      // [nb_total][320]
  const double* mincost;   // [nb_total]
  const u16* codes;        // the DP edges as weight codes (k_codes), block b from code_base[b] on
  const u64* code_base;
  const u64* block_edges;
  const double* wtab;      // [nb_total][ZMX_WTAB] the run's weights (k_wtab)
  u16* la;
  u64* prof;               // optional [nb_total][ZMX_PROF_N] counters (ZOPFLI_AMD_PROF), else null
  const u32* badpos;       // k_badscan's bad-edge bitmap
  const SegTask* tasks;
  const u32* task_off;     // [nb_total + 1] first task of each block
  float* lvl;              // [tasks] guessed value of cell q, refined by every run
  const float* est_bits;   // [nb_total] estimated block cost in bits (first run of a table set), or null
  SegSnap* entry;          // [tasks]
  SegSnap* exit;           // [tasks]
  SegSnap* mid;            // [tasks] the state where a task first comes near the end of its binade (zmx_dp5.h), base = SEG_NONE: none
  SegCheck* chk;           // [tasks]
  const float* wmax;       // [nb_total] no edge weight of the run exceeds this
  const u32* tiemask;      // [nb_total] bit e: a weight of the run can tie in the float rounding of binade e
  u32* stats;              // [8] tasks / accepted / re-run: state, level, tie / positions re-run / re-run: values
  float level_scale;       // test hook (ZOPFLI_AMD_SEG_SCALE): the guessed levels are multiplied by this
  const u32* wg_tasks;     // k_dp5_spec: [workgroups][4] the tasks of each workgroup (one block each; SEG_NONE = none),
                           // the workgroups that hold a head first; task0 = first workgroup of the launch
  const u32* wmeta;        // k_dp5_spec: [windows][40] the record of each 32-position window (k_mkdesc)
  const u32* winroff;      // k_dp5_spec: [windows] first row of each window (k_mkdesc)
  const u32* winflag;      // k_dp5_spec: per 32-position window, 1 = can take the fast path (k_mkdesc)
  const u32* win_off;      // [nb_total] first window of each block in winflag[]
  int debug;               // ZOPFLI_AMD_SEG_DEBUG: k_dp4_fix prints its decisions
  u16* over;               // [tasks][SEG_OVER] lengths of the cells a speculative task computed beyond its pend
  u32* redo_count;         // k_dpscan: number of tasks to run a second time ...
  u32* redo_wg;            // ... and their workgroups for k_dp5_spec ([count][4]: the task, then SEG_NONE)
  int int_path;            // k_dp5_spec: 1 = clean windows inside the workgroup's binade take the integer chain step (ZOPFLI_AMD_INT_PATH)
  int fix_lean_min;        // k_dp4_fix: a task with this many generic windows is re-run by the lean one-wave job
  int redo_pass;           // k_dp5_spec: 1 = this launch runs P.redo_wg (workgroups beyond *redo_count have nothing to do)
  int chain_fast;          // run tasks: 1 = chains of long-run shortcuts in a frame that does not move (zmx_dp5.h; ZOPFLI_AMD_SHORTCUT_CHAIN=0: window by window)
  // the cooperative run tasks (zmx_dp6.h)
  int coop;                // 1 = run tasks are k_dp6_spec's (four waves a task); k_dp5_spec's second pass then skips them
  const u32* kind;         // [tasks] k_taskkind: 1 = a run task
  const u32* run_list;     // k_dp6_spec, first pass: the run tasks, longest first
  u32* flags;              // [2] the table set's consistency flags (flags[1] bit 3: a cooperative job's wave gave up waiting)
};

// What one pass over a stretch of the chain does.  k_dp5_spec's jobs: the head of a block (exact: the
// block's true initial state) and the speculative tasks (one cell holding a guessed level, entry
// state recorded at the first window at or after pout).  k_dp4_fix's jobs: a task again, from the
// true exit state of its predecessor.
struct D4Job {
  u32 cell = 0;            // spec, not load: the window cell that holds `level` (the task starts at start + cell)
  u32 start;               // first window / group base
  u32 noshort;             // walk state there
  u32 pout;                // spec: the entry state is recorded at the first window base >= pout
  u32 pend;                // the walk stops at the first window / group base >= pend
  u32 la_lo;               // length_array is written for cells >= la_lo (SEG_NONE: from the entry on)
  u32 over_lo;             // ... into `over` instead for cells >= over_lo (SEG_NONE: never)
  bool spec;
  bool load;               // initial state from `init` (+ delta) instead of a single cell holding `level`
  float level;
  double delta;
  const SegSnap* init;
  SegSnap* entry;
  SegSnap* exit;
  SegSnap* mid = nullptr;  // spec: where to leave the state at the first window that comes near the end of the binade
  u16* over;               // [SEG_OVER] lengths of the cells from over_lo on
};

#define D4_LDS_DECL                                                                                   \
  __shared__ __align__(16) u16 s_ring[DP_FRONT + DP_RING + DP_MIRROR];   /* weight codes of the rows */ \
  __shared__ __align__(16) double s_wtab[ZMX_WTAB];                      /* the run's weights */        \
  __shared__ __align__(16) double s_t1[2][64 * 64];   /* register-0 rows, row = position in the group */ \
  __shared__ __align__(16) double s_t2[2][32 * 64];   /* register-1 rows, row = position & 31 */      \
  __shared__ uint2 s_tab[D3_NB][64];                                                                  \
  /* step descriptors, written by wave 1 a step ahead of the builders, two ahead of the chain wave: */ \
  /* [0] q | n << 8 | event << 16 | last << 24 [1] base [2..3] m_r1 [4..5] m_bad */                  \
  __shared__ __align__(8) u32 s_desc[3][8];                                                           \
  __shared__ uint2 s_tabc[3][64];   /* {roff, kend} of the step's group */                            \
  __shared__ float s_xc[DP_XN];                                                                       \
  __shared__ u16 s_xl[DP_XN];                                                                         \
  /* "1 + source" of the 2 x 32 cells a clean step retires, per tile buffer: wave 1 turns them */     \
  /* into length_array a step later (the chain wave only does one LDS write per window) */            \
  __shared__ u32 s_lout[2][128];

// One job, by all four waves of the workgroup (every wave executes the same number of barriers).
template <bool PROF>
__device__ __forceinline__ void d4_run_job(const Dp4Params& P, const D4Job& J, u32 b, const BlockDesc& bd,
                                           u16 (&s_ring)[DP_FRONT + DP_RING + DP_MIRROR], double (&s_wtab)[ZMX_WTAB],
                                           double (&s_t1)[2][64 * 64],
                                           double (&s_t2)[2][32 * 64], uint2 (&s_tab)[D3_NB][64], u32 (&s_desc)[3][8],
                                           uint2 (&s_tabc)[3][64], float (&s_xc)[DP_XN], u16 (&s_xl)[DP_XN],
                                           u32 (&s_lout)[2][128]) {
  const u32 tid = threadIdx.x;
  const u32 wave = (u32)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
  const u32 lane = tid & 63;
  const u32 B = (u32)(bd.inend - bd.instart);
  const uint2* dbase = P.dph + bd.pos_off;
  u16* la = P.la + bd.la_off;
  const u16* rows = P.codes + P.code_base[b];
  const u32 total_pad = (u32)((P.block_edges[b] + DP_PIECE - 1) & ~(u64)(DP_PIECE - 1));
  const double mincost = P.mincost[b];
  // squeeze.c:260: cost of (length 258, dist 1) = (0 + 0) + ll[285] + d[0]
  const double symbolcost258 = (double)(0 + 0) + P.cost[(u64)b * 320 + 285] + P.cost[(u64)b * 320 + 288];
  const double kInf = __longlong_as_double(0x7ff0000000000000ll);
  const u32 ring_lds = (u32)(unsigned long)(__attribute__((address_space(3))) u16*)s_ring;
  // the weight of the code in ring slot i (codes are byte offsets into the table; a slot that holds
  // no code of this block — masked off by the caller — may hold anything: keep the read inside LDS)
  auto ring_w = [&](const u16* slot) -> double {
    return *reinterpret_cast<const double*>(reinterpret_cast<const char*>(s_wtab) + ((u32)*slot & 0x3ff8u));
  };
#define D3_TICK() (PROF ? (u64)__builtin_readcyclecounter() : 0ull)

  if (wave == 0) {
    // ================================================================= consumer
    float c[6];
    u32 l[6];
    u32 reach;   // no cell beyond window cell `reach` has been written: registers above reach >> 6 are fresh
    {
#pragma unroll
      for (int s = 0; s < 6; ++s) {
        const float ec = J.init->c[64u * s + lane];
        c[s] = ec < 1e29f ? (float)((double)ec + J.delta) : 1e30f;   // exact: both are multiples of the binade's ulp
        l[s] = J.init->l[64u * s + lane];
      }
      reach = SEG_CELLS - 1;
    }
    const u32 la_lo = J.la_lo;
    u32 wo = 0;   // the cell registers cover cells base + wo + 64 s + lane: wo = 32 once the chain is past position 31 of the group
    u64 t_work = 0, n_fast = 0, n_slow = 0, n_steps = 0;
    u64 tp[5] = {0, 0, 0, 0, 0}, np[5] = {0, 0, 0, 0, 0};   // PROF: cycles and positions per path

    __syncthreads();   // iteration 0: wave 1 walks step 0
    __syncthreads();   // iteration 1: the builders' first step, nothing to consume yet
    u32 it = 2;        // this wave works on step it - 2
    // the first two words of the next step's descriptor are requested a step early (they were
    // written two barriers ago); the masks are read only for a step that is not a clean group
    uint2 nd = *reinterpret_cast<const uint2*>(s_desc[0]);
    for (;;) {
      const u64 tw0 = PROF ? (u64)__builtin_readcyclecounter() : 0ull;
      // the step as wave 1 described it (the consumer never looks at dph[] itself)
      u32 dv[6];
      dv[0] = (u32)__builtin_amdgcn_readfirstlane((int)nd.x);
      dv[1] = (u32)__builtin_amdgcn_readfirstlane((int)nd.y);
      nd = *reinterpret_cast<const uint2*>(s_desc[(it - 1) % 3]);
      dv[2] = dv[3] = dv[4] = dv[5] = 0;
      if (!(dv[0] & D3_DESC_CLEAN)) {
#pragma unroll
        for (int i = 2; i < 6; ++i) dv[i] = s_desc[(it - 2) % 3][i];
#pragma unroll
        for (int i = 2; i < 6; ++i) dv[i] = (u32)__builtin_amdgcn_readfirstlane((int)dv[i]);
      }
      D3Step S;
      S.q = dv[0] & 255u; S.n = (dv[0] >> 8) & 255u; S.event = (dv[0] >> 16) & 255u; S.base = dv[1]; S.a_cur = 0;
      const bool last = ((dv[0] >> 24) & 1u) != 0;
      struct { u64 m_r1, m_bad; } G;   // the step's group, as wave 1 described it
      G.m_r1 = ((u64)dv[3] << 32) | dv[2];
      G.m_bad = ((u64)dv[5] << 32) | dv[4];
      const uint2* tabc = s_tabc[(it - 2) % 3];
      const double* t1 = s_t1[it & 1];
      const double* t2 = s_t2[it & 1];
      const u32 send = S.q + S.n;
      const u32 base = S.base;
      u32 p0 = S.q;
      u32 bi = 0;   // block index within the step
      if ((dv[0] & D3_DESC_CLEAN) && reach < 64) {   // (q = 0: a new group, so wo = 0; only register 0 is live)
        // a whole group of single-register positions (the usual step): both windows' rows are
        // requested up front, the second window's arrive while the first one runs
        const u64 tk = D3_TICK();
        double wa[32], wb[32];
#pragma unroll
        for (int u = 0; u < 32; ++u) wa[u] = t1[u * 64 + lane];
#pragma unroll
        for (int u = 0; u < 32; ++u) wb[u] = t1[(32 + u) * 64 + lane];
        u32 lt = 0;                              // 1 + index of the last position that updated the cell
#pragma unroll
        for (int u = 0; u < 32; ++u) {
          const double cj = (double)rdlane_f32(c[0], (u32)u);
          D3_RELAX_K(c[0], lt, wa[u], (u32)(u + 1))
        }
        l[0] = lt ? base + lt : l[0];
        s_lout[it & 1][lane] = l[0];             // cells base .. base + 31 are final (lanes 0..31)
        c[0] = __uint_as_float(__builtin_amdgcn_permlane32_swap(__float_as_uint(c[0]), __float_as_uint(1e30f), false, false)[1]);
        l[0] = __builtin_amdgcn_permlane32_swap(l[0], 0u, false, false)[1];
        lt = 0;
#pragma unroll
        for (int u = 0; u < 32; ++u) {
          const double cj = (double)rdlane_f32(c[0], (u32)u);
          D3_RELAX_K(c[0], lt, wb[u], (u32)(u + 1))
        }
        l[0] = lt ? base + 32 + lt : l[0];
        n_fast += 64;
        if (PROF) { tp[0] += D3_TICK() - tk; np[0] += 64; }
        // its event is D3_EV_GROUP_END (64 positions, none flagged): retire the second window
        s_lout[it & 1][64 + lane] = l[0];        // cells base + 32 .. base + 63
        c[0] = __uint_as_float(__builtin_amdgcn_permlane32_swap(__float_as_uint(c[0]), __float_as_uint(1e30f), false, false)[1]);
        l[0] = __builtin_amdgcn_permlane32_swap(l[0], 0u, false, false)[1];
        reach = 31;                              // cells up to window cell 63 were written, the window moved twice
        wo = 0;
        if (PROF) { t_work += (u64)__builtin_readcyclecounter() - tw0; ++n_steps; }
        __syncthreads();
        ++it;
        if (last) break;
        continue;
      }
      for (; p0 < S.q + S.n; ++bi) {
        const u64 tk = D3_TICK();
        if (p0 >= 32 && wo == 0) { D3_RETIRE32(base) wo = 32; }
        const u32 pw0 = p0 - wo;                 // window lane of position p0
        // a whole window of single-register positions (the usual case)
        if (pw0 == 0 && p0 + 32 <= S.q + S.n && ((u32)(G.m_bad >> p0)) == 0 &&
            ((u32)(G.m_r1 >> p0)) == 0) {
          double w0[32];
#pragma unroll
          for (int u = 0; u < 32; ++u) w0[u] = t1[(p0 + u) * 64 + lane];
          u32 lt = 0;                            // 1 + index of the last position that updated the cell
#pragma unroll
          for (int u = 0; u < 32; ++u) {
            const double cj = (double)rdlane_f32(c[0], (u32)u);
            D3_RELAX_K(c[0], lt, w0[u], (u32)(u + 1))
          }
          l[0] = lt ? base + p0 + lt : l[0];
          reach = reach > 63 ? reach : 63;
          n_fast += 32;
          p0 += 32;
          bi += 3;
          if (PROF) { tp[0] += D3_TICK() - tk; np[0] += 32; }
          continue;
        }
        // two single-register blocks in one go: 16 rows in flight, half the dispatch
        if (pw0 <= 16 && d3_fast(send, G.m_bad, p0) && d3_fast(send, G.m_bad, p0 + 8) &&
            ((u32)(G.m_r1 >> p0) & 0xffffu) == 0) {
          double w0[16];
#pragma unroll
          for (int u = 0; u < 16; ++u) w0[u] = t1[(p0 + u) * 64 + lane];
          u32 lt = 0;
#pragma unroll
          for (int u = 0; u < 16; ++u) {
            const double cj = (double)rdlane_f32(c[0], pw0 + u);
            D3_RELAX_K(c[0], lt, w0[u], (u32)(u + 1))
          }
          l[0] = lt ? base + p0 + lt : l[0];
          reach = reach > 63 ? reach : 63;
          n_fast += 16;
          p0 += 16;
          ++bi;
          if (PROF) { tp[1] += D3_TICK() - tk; np[1] += 16; }
          continue;
        }
        if (d3_fast(send, G.m_bad, p0)) {
          const bool two = ((u32)(G.m_r1 >> p0) & 255u) != 0;
          double w0[8], w1[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) w0[u] = t1[(p0 + u) * 64 + lane];
          if (two) {
#pragma unroll
            for (int u = 0; u < 8; ++u) w1[u] = t2[((p0 + u) & 31) * 64 + lane];
          }
          u32 lt = 0, lt1 = 0;
          if (!two) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              const double cj = (double)rdlane_f32(c[0], pw0 + u);
              D3_RELAX_K(c[0], lt, w0[u], (u32)(u + 1))
            }
          } else {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              const double cj = (double)rdlane_f32(c[0], pw0 + u);
              D3_RELAX_K(c[0], lt, w0[u], (u32)(u + 1))
              D3_RELAX_K(c[1], lt1, w1[u], (u32)(u + 1))
            }
            l[1] = lt1 ? base + p0 + lt1 : l[1];
          }
          l[0] = lt ? base + p0 + lt : l[0];
          { const u32 r_ = two ? 127u : 63u; reach = reach > r_ ? reach : r_; }
          n_fast += 8;
          p0 += 8;
          if (PROF) { tp[two ? 3 : 2] += D3_TICK() - tk; np[two ? 3 : 2] += 8; }
          continue;
        }
        // generic path straight from the ring (ragged tails, long matches, exempt flagged positions,
        // blocks that straddle two windows or have an edge below mincost): the reference's tests, literally
        const u32 pend = p0 + 8 <= S.q + S.n ? p0 + 8 : S.q + S.n;
        for (u32 p = p0; p < pend; ++p) {
          if (p >= 32 && wo == 0) { D3_RETIRE32(base) wo = 32; }
          const u32 pw = p - wo;
          const uint2 tc = tabc[p];
          const u32 ro = (u32)__builtin_amdgcn_readfirstlane((int)tc.x);
          const u32 ke = (u32)__builtin_amdgcn_readfirstlane((int)tc.y);
          const double cj = (double)rdlane_f32(c[0], pw);
          const u32 src1 = base + p + 1;
          const u32 km1 = lane - pw - 1;
          const u32 smax = (ke + pw) >> 6;
          reach = reach > ke + pw ? reach : ke + pw;
          if (smax < 2) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
              if ((u32)s <= smax) {
                const u32 k1 = km1 + 64u * s;
                if (k1 < ke) {
                  const double w = ring_w(&s_ring[DP_FRONT + ((ro + k1) & (DP_RING - 1))]);
                  const double mcl = k1 == 0 ? -kInf : mincost;
                  DP_RELAX(c[s], l[s], w, mcl)
                }
              }
            }
          } else {
            // a long row (runs, long repeats): all six registers, branch-free, so that the six code
            // reads and then the six weight reads are in flight together instead of one dependent
            // pair of LDS round trips per register.  A lane outside the row takes slot "FRONT - 1 -
            // lane", whatever it holds, and +inf instead of its weight.
            u32 cd[6];
            double wv[6];
#pragma unroll
            for (int s = 0; s < 6; ++s) {
              const u32 k1 = km1 + 64u * s;
              cd[s] = s_ring[DP_FRONT + ((ro + (k1 < ke ? k1 : 0u)) & (DP_RING - 1))];
            }
#pragma unroll
            for (int s = 0; s < 6; ++s) {
              const u32 k1 = km1 + 64u * s;
              const double w = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(s_wtab) + (cd[s] & 0x3ff8u));
              wv[s] = k1 < ke ? w : kInf;
            }
#pragma unroll
            for (int s = 0; s < 6; ++s) {
              const u32 k1 = km1 + 64u * s;
              const double mcl = k1 == 0 ? -kInf : mincost;
              DP_RELAX(c[s], l[s], wv[s], mcl)
            }
          }
        }
        n_slow += pend - p0;
        if (PROF) { tp[4] += D3_TICK() - tk; np[4] += pend - p0; }
        p0 = pend;
      }
      if (S.event == D3_EV_GROUP_END) {
        // every cell of the group is final: retire what is left of it, the registers move on to
        // the next group's base
        if (wo == 0) { D3_RETIRE32(base) }
        { D3_RETIRE32(base + 32) }
        wo = 0;
      } else if (S.event == D3_EV_SHORTCUT) {
        // long-run shortcut at position q + n of the group (squeeze.c:251-271)
        const u32 p = S.q + S.n;
        if (p >= 32 && wo == 0) { D3_RETIRE32(base) wo = 32; }
        const u32 pw = p - wo, wbase = base + wo;
        const u32 j = base + p;
        if (lane < pw && wbase + lane >= la_lo) la[wbase + lane] = (u16)(l[0] ? wbase + lane + 1 - l[0] : 0u);
        wave_lds_sync();
#pragma unroll
        for (int s = 0; s < 6; ++s) {
          const u32 x = wbase + 64u * s + lane;
          s_xc[64 * s + lane] = c[s];
          s_xl[64 * s + lane] = (u16)(l[s] ? x + 1 - l[s] : 0u);
        }
        wave_lds_sync();
        // costs[j+t+258] = costs[j+t] + symbolcost for t = 0..257, unconditionally; cells
        // j..j+257 are consumed with the lengths they have now
        float nc4[5];
#pragma unroll
        for (int r = 0; r < 5; ++r) {
          const u32 t = 64u * r + lane;
          nc4[r] = 1e30f;
          if (t < ZMX_MAX_MATCH) {
            if (j + t >= la_lo) la[j + t] = s_xl[pw + t];
            nc4[r] = (float)((double)s_xc[pw + t] + symbolcost258);
          }
        }
#pragma unroll
        for (int s = 0; s < 6; ++s) { c[s] = 1e30f; l[s] = 0; }
#pragma unroll
        for (int r = 0; r < 5; ++r) {
          const u32 t = 64u * r + lane;
          if (t < ZMX_MAX_MATCH) { c[r] = nc4[r]; l[r] = j + t + 1; }
        }
        wo = 0;                                  // the registers now sit at the next group's base, j + 258
        reach = ZMX_MAX_MATCH - 1;
        wave_lds_sync();
      }
      if (PROF) { t_work += (u64)__builtin_readcyclecounter() - tw0; ++n_steps; }
      __syncthreads();
      ++it;
      if (last) break;
    }
    if (J.exit) {
      // the registers sit at the base the walk stopped at (wave 1 writes the header)
#pragma unroll
      for (int s = 0; s < 6; ++s) {
        J.exit->c[64u * s + lane] = c[s];
        J.exit->l[64u * s + lane] = l[s];
      }
    }
    if (PROF && P.prof && lane == 0) {
      u64* o = P.prof + (u64)b * ZMX_PROF_N;
      atomicAdd(&o[0], n_steps); atomicAdd(&o[1], t_work); atomicAdd(&o[2], n_fast); atomicAdd(&o[3], n_slow);
      atomicAdd(&o[4], n_fast + n_slow);
      for (int i = 0; i < 5; ++i) { atomicAdd(&o[5 + 2 * i], tp[i]); atomicAdd(&o[6 + 2 * i], np[i]); }
    }
  } else if (wave == 1) {
    // =================================================================== wave 1: the walk and the ring
    u64 tq[4] = {0, 0, 0, 0};   // PROF: cycles in the walk / ring upkeep / tiles / barrier
    D3Walk W;
    W.base = J.start;
    W.noshort = J.noshort != 0;
    D3Group G;
    G.roff = G.kend = G.offend = 0; G.m_short = G.m_r1 = G.m_bad = 0; G.navail = 0;
    D3Step cur;              // the step the builders work on during this iteration (walked one iteration ago)
    cur.base = cur.q = cur.n = cur.a_cur = 0; cur.event = D3_EV_BUBBLE;
    u32 issued_end = 0;      // rows [.., issued_end) have been requested into the ring
    u32 a_prev = 0;          // a_cur of the step the chain wave works on during this iteration
    u32 it = 0, tail = 0;
    bool more = true;
    const u32 la_lo = J.la_lo;
    // base and "clean" of the steps walked 1, 2, 3 iterations ago: the chain wave finished the
    // oldest during the previous iteration and left its lengths in s_lout
    u32 hb1 = 0, hb2 = 0, hb3 = 0;
    bool hc1 = false, hc2 = false, hc3 = false;
#define D3_STORE_LA()                                                        \
    if (hc3) {                                                               \
      const u32 v_ = s_lout[(it - 1) & 1][(lane >> 5) * 64 + (lane & 31)];   \
      const u32 jj_ = hb3 + lane;                                            \
      if (jj_ >= la_lo && jj_ <= B) la[jj_] = (u16)(v_ ? jj_ + 1 - v_ : 0u); \
    }
    while (tail < 2) {       // two more barriers after the last step has been walked
      const u64 tk0 = D3_TICK();
      D3_STORE_LA()
      // ---- the ring, for the step walked one iteration ago
      if (cur.event == D3_EV_PRIME) {
        const u32 a0 = cur.a_cur;
        issued_end = a0;
        a_prev = a0;
        const u32 lim = a0 + DP_RING < total_pad ? a0 + DP_RING : total_pad;
        while (issued_end < lim) {
          const u32 slot = issued_end & (DP_RING - 1);
          dp_dma_piece(rows + issued_end + lane * 8, ring_lds + (DP_FRONT + slot) * 2);
          if (slot < DP_MIRROR) dp_dma_piece(rows + issued_end + lane * 8, ring_lds + (DP_FRONT + DP_RING + slot) * 2);
          issued_end += DP_PIECE;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } else if (cur.event != D3_EV_BUBBLE) {
        // what was requested during the previous iteration has landed; keep the ring one ring ahead
        // of the step the chain wave is reading
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const u32 lim = a_prev + DP_RING < total_pad ? a_prev + DP_RING : total_pad;
        while (issued_end < lim) {
          const u32 slot = issued_end & (DP_RING - 1);
          dp_dma_piece(rows + issued_end + lane * 8, ring_lds + (DP_FRONT + slot) * 2);
          if (slot < DP_MIRROR) dp_dma_piece(rows + issued_end + lane * 8, ring_lds + (DP_FRONT + DP_RING + slot) * 2);
          issued_end += DP_PIECE;
        }
        if (cur.n) a_prev = cur.a_cur;
      }
      const u64 tk1 = D3_TICK();
      // ---- walk one step ahead and describe it
      cur.n = 0; cur.event = D3_EV_BUBBLE;
      hb3 = hb2; hc3 = hc2; hb2 = hb1; hc2 = hc1; hc1 = false;
      if (more) {
        cur = d3_next(W, G, dbase, P.badpos + (bd.pos_off >> 5), (u32)(bd.pos_off & 31), B, lane);
        more = W.bubbles || W.base < J.pend;
        if (cur.n) s_tabc[it % 3][lane] = make_uint2(G.roff, G.kend);
        const bool clean = cur.q == 0 && cur.n == 64 && (G.m_r1 | G.m_bad) == 0;
        hb1 = cur.base; hc1 = clean;
        if (lane == 0) {
          u32* d = s_desc[it % 3];
          d[0] = cur.q | (cur.n << 8) | (cur.event << 16) | ((more ? 0u : 1u) << 24) | (clean ? D3_DESC_CLEAN : 0u);
          d[1] = cur.base;
          d[2] = (u32)G.m_r1; d[3] = (u32)(G.m_r1 >> 32);
          d[4] = (u32)G.m_bad; d[5] = (u32)(G.m_bad >> 32);
        }
      } else {
        ++tail;
      }
      const u64 tk2 = D3_TICK();
      __syncthreads();
      if (PROF) { tq[1] += tk1 - tk0; tq[0] += tk2 - tk1; tq[3] += D3_TICK() - tk2; }
      ++it;
    }
    D3_STORE_LA()                                       // the last step
#undef D3_STORE_LA
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS-DMA in flight when the ring is reused
    if (J.exit && lane == 0) { J.exit->base = W.base; J.exit->noshort = W.noshort ? 1u : 0u; J.exit->skip = 0u; }
    if (PROF && P.prof && lane == 0) {
      u64* o = P.prof + (u64)b * ZMX_PROF_N + 16;
      for (int i = 0; i < 4; ++i) atomicAdd(&o[i], tq[i]);
    }
  } else {
    // ================================================================= waves 2..: the tiles
    u64 tq[4] = {0, 0, 0, 0};
    const u32 my = wave - 2;
    __syncthreads();         // iteration 0: wave 1 walks step 0
    u32 it = 1;              // this wave works on step it - 1
    for (;;) {
      const u64 tk0 = D3_TICK();
      u32 dv[6];
      dv[0] = (u32)__builtin_amdgcn_readfirstlane((int)s_desc[(it - 1) % 3][0]);
      dv[2] = dv[3] = dv[4] = dv[5] = 0;
      const bool clean = (dv[0] & D3_DESC_CLEAN) != 0;
      if (!clean) {
#pragma unroll
        for (int i = 2; i < 6; ++i) dv[i] = s_desc[(it - 1) % 3][i];
#pragma unroll
        for (int i = 2; i < 6; ++i) dv[i] = (u32)__builtin_amdgcn_readfirstlane((int)dv[i]);
      }
      const u32 sq = dv[0] & 255u, sn = (dv[0] >> 8) & 255u, sev = (dv[0] >> 16) & 255u;
      const bool last = ((dv[0] >> 24) & 1u) != 0;
      const u64 m_r1 = ((u64)dv[3] << 32) | dv[2], m_bad = ((u64)dv[5] << 32) | dv[4];
      if (sn && sev != D3_EV_BUBBLE && sev != D3_EV_PRIME) {
        const uint2 tc = s_tabc[(it - 1) % 3][lane];
        wave_lds_sync();
        s_tab[my][lane] = make_uint2(((tc.x & (DP_RING - 1)) - lane - 1) * 2u, tc.y);   // byte offset of slot row[-1 - lane]
        wave_lds_sync();
        double* t1 = s_t1[(it - 1) & 1];
        double* t2 = s_t2[(it - 1) & 1];
        const char* ring0 = reinterpret_cast<const char*>(s_ring + DP_FRONT);
        if (clean && D3_NB == 2) {
          // a whole group of single-register positions: this wave's 16 positions of each window in
          // one pass (32 rows in flight)
          const u32 pa = 16u * my, pb = pa + 32u;
          uint2 ta[16], tb[16];
#pragma unroll
          for (int u = 0; u < 16; ++u) { ta[u] = s_tab[my][pa + u]; tb[u] = s_tab[my][pb + u]; }
          double va[16], vb[16];
#pragma unroll
          for (int u = 0; u < 16; ++u) {
            va[u] = ring_w(reinterpret_cast<const u16*>(ring0 + (int)ta[u].x) + lane);        // row[x] = edge k = x - p
            vb[u] = ring_w(reinterpret_cast<const u16*>(ring0 + (int)tb[u].x) + lane + 32u);
          }
          const u32 da = lane - pa - 1, db = lane + 32u - pb - 1;
#pragma unroll
          for (int u = 0; u < 16; ++u) {
            t1[(pa + u) * 64 + lane] = da - u < ta[u].y ? va[u] : kInf;
            t1[(pb + u) * 64 + lane] = db - u < tb[u].y ? vb[u] : kInf;
          }
          const u64 tk1c = D3_TICK();
          __syncthreads();
          if (PROF) { tq[2] += tk1c - tk0; tq[3] += D3_TICK() - tk1c; }
          ++it;
          if (last) break;
          continue;
        }
        u32 blk = 0;
        for (u32 p0 = sq; p0 < sq + sn; p0 += 8) {
          if (!d3_fast(sq + sn, m_bad, p0)) continue;   // (a non-fast block is at most 8 positions: the walk realigns after it)
          const u32 wl = lane + (p0 & 32u);   // window lane -> group lane
          const u32 d0 = wl - p0 - 1;
          // two single-register blocks of one window in one pass: twice the LDS reads in flight
          if ((p0 & 31u) <= 16u && d3_fast(sq + sn, m_bad, p0 + 8) && ((u32)(m_r1 >> p0) & 0xffffu) == 0) {
            if ((blk++ % D3_NB) == my) {
              uint2 t[16];
#pragma unroll
              for (int u = 0; u < 16; ++u) t[u] = s_tab[my][p0 + u];
              double v0[16];
#pragma unroll
              for (int u = 0; u < 16; ++u) v0[u] = ring_w(reinterpret_cast<const u16*>(ring0 + (int)t[u].x) + wl);   // row[x] = edge k = x - p
#pragma unroll
              for (int u = 0; u < 16; ++u) t1[(p0 + u) * 64 + lane] = d0 - u < t[u].y ? v0[u] : kInf;
            }
            p0 += 8;
            continue;
          }
          if ((blk++ % D3_NB) != my) continue;
          const bool two = ((u32)(m_r1 >> p0) & 255u) != 0;
          uint2 t[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) t[u] = s_tab[my][p0 + u];
          if (!two) {
            double v0[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v0[u] = ring_w(reinterpret_cast<const u16*>(ring0 + (int)t[u].x) + wl);
#pragma unroll
            for (int u = 0; u < 8; ++u) t1[(p0 + u) * 64 + lane] = d0 - u < t[u].y ? v0[u] : kInf;
          } else {
            double v0[8], v1[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              const u16* row = reinterpret_cast<const u16*>(ring0 + (int)t[u].x);
              v0[u] = ring_w(row + wl);
              v1[u] = ring_w(row + wl + 64);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              t1[(p0 + u) * 64 + lane] = d0 - u < t[u].y ? v0[u] : kInf;
              t2[((p0 + u) & 31) * 64 + lane] = d0 - u + 64 < t[u].y ? v1[u] : kInf;
            }
          }
        }
      }
      const u64 tk1 = D3_TICK();
      __syncthreads();
      if (PROF) { tq[2] += tk1 - tk0; tq[3] += D3_TICK() - tk1; }
      ++it;
      if (last) break;
    }
    __syncthreads();         // the chain wave's last step
    if (PROF && P.prof && wave == 2 && lane == 0) {
      u64* o = P.prof + (u64)b * ZMX_PROF_N + 24;
      for (int i = 0; i < 4; ++i) atomicAdd(&o[i], tq[i]);
    }
  }
#undef D3_TICK
  // the LDS (ring, tiles, descriptors) is reused by the next job, and the snapshots written here are
  // read back by the other waves (k_dp4_fix): make them visible beyond this wave's stores
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

// exit[t - 1] against entry[t] for every task but the heads: one wave per task
__global__ __launch_bounds__(64) void k_dpcheck(Dp4Params P) {
  // (k_dpscan's list of tasks to run again starts empty: every k_dpscan follows a k_dpcheck, and whoever reads the
  //  counter — the redo pass of k_dp5_spec — has run by the time the next k_dpcheck starts)
  if (blockIdx.x == 0 && threadIdx.x == 0) *P.redo_count = 0;
  const u32 t = P.task0 + blockIdx.x;
  if (P.tasks[t].pout == 0) return;
  const SegCheck r = d4_check(&P.exit[t - 1], &P.entry[t], threadIdx.x);
  if (threadIdx.x == 0) P.chk[t] = r;
}

// The acceptance test of a task whose entry state matched its predecessor's exit state up to the
// shift delta: 0 = accept, 1 = the guessed or the shifted values leave the binade, 2 = a weight of
// the run can tie in that binade's float rounding.
__device__ __forceinline__ u32 d4_accept(float vmin, double vmax, double delta, double wmax, u32 tiemask) {
  if (delta == 0.0) return 0;
  const int e = (int)((__float_as_uint(vmin) >> 23) & 255u) - 127;
  const double lo = ldexp(1.0, e), hi = ldexp(1.0, e + 1);
  const bool pure = vmin >= 16.0f && vmax + wmax < hi && (double)vmin + delta >= lo && vmax + wmax + delta < hi;
  if (!pure) return 1;
  return ((tiemask >> (e & 31)) & 1u) != 0 ? 2u : 0u;
}

// ---------------------------------------------------------------------------------------------
// k_dpscan: between the speculative pass and the fix pass.  A task that only missed because its
// guessed level was off (wrong binade, or a warm-up that crossed one) is worth a second speculative
// run: the chain of shifts gives its true level to within an ulp or two, and a run started from THAT
// level usually reproduces the true entry state exactly (shift 0: accepted whatever binades the task
// crosses afterwards).  One wave per block walks the checks, assuming every task will be accepted in
// the end, and lists the tasks to run again with their new levels; k_dp5_spec runs the list, k_dpcheck
// checks everything again, and only what still fails is left to the serial pass.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_dpscan(Dp4Params P) {
  const u32 b = P.block0 + blockIdx.x;
  const u32 t0 = P.task_off[b], t1 = P.task_off[b + 1];
  const u32 lane = threadIdx.x;
  const double wmax = (double)P.wmax[b] + 1.0;
  const u32 tiemask = P.tiemask[b];
  // The shift of task t is the sum of the differences up to t — sums of multiples of float ulps, exact in any
  // order — so the walk is a prefix sum: 64 tasks at a time, a lane each (one lane walking 240 tasks with two
  // dependent loads per task was 0.12 ms a run).
  double carry = 0.0;
  for (u32 c0 = t0 + 1; c0 < t1; c0 += 64) {
    const u32 t = c0 + lane;
    const bool act = t < t1;
    SegCheck ck;
    ck.d = 0.0; ck.vmin = 0.0f; ck.match = 0;
    float vmax = 0.0f;
    if (act) { ck = P.chk[t]; vmax = P.exit[t].vmax; }
    double incl = act ? ck.d : 0.0;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const double up = __shfl_up(incl, o, 64);
      if ((int)lane >= o) incl += up;
    }
    const double delta = carry + incl;
    if (act) {
      const bool ok = ck.match == 1 && d4_accept(ck.vmin, (double)vmax, delta, wmax, tiemask) == 0;
      if (!ok && ck.match != 0) {          // same structure, wrong level: again from the level the chain implies
        const u32 slot = atomicAdd(P.redo_count, 1u);
        u32* wg = P.redo_wg + (u64)slot * 4;
        wg[0] = t; wg[1] = SEG_NONE; wg[2] = SEG_NONE; wg[3] = SEG_NONE;
        P.lvl[t] = (float)((double)P.lvl[t] + delta);
      }
    }
    // (the exit snapshots at hand are those of the first run: whatever becomes of a task, its exit there plus
    //  its shift is the best estimate of the true state the next task starts from)
    carry += __shfl(incl, 63, 64);
  }
}

// A task of the speculative pass stops at its first window base >= pend and has the lengths of the
// cells from pend on in its side buffer, not in length_array: its successor may have started writing
// at a SMALLER base than the one it stopped at (walks out of step after a long-run shortcut), and two
// workgroups must not write the same cells.  Whoever accepts the task puts them in place.
__device__ __forceinline__ void d4_copy_over(const Dp4Params& P, u32 t, u32 B, u16* la) {
  const u32 pend = P.tasks[t].pend;
  if (pend > B) return;                       // the last task of the block runs to the end
  const u32 stop = P.exit[t].base + P.exit[t].skip;   // (cells of the last window that a shortcut had already consumed: zmx_dp5.h)
  const u16* over = P.over + (u64)t * SEG_OVER;
  for (u32 i = threadIdx.x; pend + i < stop && pend + i <= B; i += blockDim.x) la[pend + i] = over[i];
}

// (k_dp4_fix, the serial pass, is at the end of zmx_dp5.h: it uses the jobs of both files)

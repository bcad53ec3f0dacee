// k_dp5_spec: the speculative pass over the chain's tasks (see zmx_dp4.h for what a task is and why
// its result can be trusted) written for THROUGHPUT: one wave per task, no LDS staging, no helper
// waves, so that a CU runs a dozen tasks at once and the waves of a SIMD fill each other's stalls.
// Included only by zmx_hip.hip, after zmx_dp4.h (same jobs, snapshots, cell registers, arithmetic).
//
// k_dp4's four-wave pipeline is built for the latency of ONE chain (one workgroup per CU, 142 KB of
// LDS): right for the stretches k_dp4_fix has to run serially, wasteful when 24 000 independent tasks
// are waiting.  Here the unit is a WINDOW of 32 positions (the cell registers move 32 cells at a
// time, as in k_dp4).  For a window whose positions all reach no further than cell register 0 and
// carry no flag (99 % of text), the wave fetches the 32 edge rows itself, straight from rows[] in
// HBM/L2 into registers, already in the lane layout the chain wants:
//
//     lane l of row u  =  rows[roff_u + l - u - 1]   if 0 <= l - u - 1 < kend_u, else +inf
//
// with everything per position on the scalar side: {roff, kend} arrive by s_load (uniform address),
// the row's lane mask is ((1 << kend) - 1) << (u + 1) in an SGPR pair, the row's base address an
// SGPR pair, and lanes outside the row are pointed at one +inf that k_edges leaves behind the
// block's rows — 2 VALU instructions and one global_load_dwordx2 per position, then the 8-instruction
// chain step of k_dp4.  Other windows (long matches, shortcut flags, edges below mincost, ragged
// tails) take the generic path, position by position, with the reference's tests literally.
#pragma once

struct D5Cls {          // one window: lane l < 32 = position wbase + l
  u32 kend, roff;
  u64 ms, m_r1, m_bad;  // flagged for the long-run shortcut / reach beyond cell register 0 / edge below mincost
  u32 nav;
};

template <bool PROF>
__device__ __forceinline__ void d5_run_job(const Dp4Params& P, const D4Job& J, u32 b, const BlockDesc& bd,
                                           float (&s_xc)[DP_XN], u16 (&s_xl)[DP_XN]) {
  const u32 lane = threadIdx.x & 63;
  const u32 lane8 = lane * 8u;
  const u32 B = (u32)(bd.inend - bd.instart);
  const uint2* __restrict__ dbase = P.dph + bd.pos_off;
  const u32* __restrict__ badpos = P.badpos + (bd.pos_off >> 5);
  const u32 bit_off = (u32)(bd.pos_off & 31);
  u16* la = P.la + bd.la_off;
  const double* __restrict__ rows = P.rows + P.row_base[b];
  // k_edges leaves one +inf at the first padded slot behind the block's rows
  const u32 tail_bytes = (u32)(((P.block_edges[b] + DP_PIECE - 1) & ~(u64)(DP_PIECE - 1)) * 8u);
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
    if (lane == 0) c[0] = J.level;
    reach = 0;
  }
  u32 la_lo = J.la_lo;
  float vmax = 0.0f;
  u32 wbase = (u32)__builtin_amdgcn_readfirstlane((int)J.start);
  bool noshort = J.noshort != 0;
  u64 n_fast = 0, n_slow = 0;
  const u64 t_begin = PROF ? (u64)__builtin_readcyclecounter() : 0ull;

  while (wbase < J.pend) {          // (J.pend = B + 1 on the last task: the window at B retires cell B)
    if (J.spec && la_lo == SEG_NONE && wbase >= J.pout) {
      // the first window at or after pout: from here on the task owns the length_array; what the
      // registers hold now is compared with the predecessor's exit state
#pragma unroll
      for (int s = 0; s < 6; ++s) { J.entry->c[64u * s + lane] = c[s]; J.entry->l[64u * s + lane] = l[s]; }
      if (lane == 0) { J.entry->base = wbase; J.entry->noshort = noshort ? 1u : 0u; }
      la_lo = wbase;
      vmax = 0.0f;
    }
    // ---- the window's positions
    D5Cls W;
    W.nav = B - wbase < 32u ? B - wbase : 32u;
    {
      const u32 jj = wbase + lane;
      const bool act = lane < W.nav;
      const u32 cur = jj < B ? jj : B - 1;
      const uint2 dh = dbase[cur];
      const u32 bw = badpos[(bit_off + cur) >> 5];
      W.kend = act ? (dh.y & 0xffffu) : 0u;
      W.roff = dh.x;
      W.ms = __ballot(act && (dh.y >> 16) != 0);
      W.m_r1 = __ballot(W.kend + lane >= 64u);
      W.m_bad = __ballot(act && ((bw >> ((bit_off + cur) & 31u)) & 1u) != 0);
    }
    if (noshort) W.ms &= ~1ull;      // squeeze.c:273: the position right after a shortcut is not tested again
    bool jumped = false;
    if (W.nav == 32u && (W.ms | W.m_r1 | W.m_bad) == 0) {
      // ---- 32 positions, one cell register, no flags: rows straight into registers, then the chain
      const uint2* __restrict__ dw = dbase + wbase;    // uniform: s_load
      double wv[32];
#pragma unroll
      for (int u = 0; u < 32; ++u) {
        const uint2 dh = dw[u];
        const u32 ke = dh.y & 0xffffu;
        const u64 mask = ((1ull << ke) - 1ull) << (u + 1);           // lanes u + 1 .. u + kend (kend + u < 64 here)
        const bool valid = __builtin_amdgcn_inverse_ballot_w64(mask);
        const int soff = (int)(dh.x - (u32)(u + 1)) * 8;              // byte offset of lane 0's slot, may be < 0
        const char* sa = reinterpret_cast<const char*>(rows) + (long long)soff;
        const u32 vo = valid ? lane8 : tail_bytes - (u32)soff;        // outside the row: the +inf behind the rows
        wv[u] = *reinterpret_cast<const double*>(sa + vo);
      }
      u32 lt = 0;                              // 1 + index of the last position that updated the cell
#pragma unroll
      for (int u = 0; u < 32; ++u) {
        const double cj = (double)rdlane_f32(c[0], (u32)u);
        D3_RELAX_K(c[0], lt, wv[u], (u32)(u + 1))
      }
      l[0] = lt ? wbase + lt : l[0];
      reach = reach > 63 ? reach : 63;
      noshort = false;
      n_fast += 32;
    } else {
      // ---- position by position
      for (u32 p = 0; p < W.nav; ++p) {
        const u32 j = wbase + p;
        if ((W.ms >> p) & 1) {
          // long-run shortcut at position j (squeeze.c:251-271)
          if (lane < p && wbase + lane >= la_lo) la[wbase + lane] = (u16)(l[0] ? wbase + lane + 1 - l[0] : 0u);
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
          float nc4[5];
#pragma unroll
          for (int r = 0; r < 5; ++r) {
            const u32 t = 64u * r + lane;
            nc4[r] = 1e30f;
            if (t < ZMX_MAX_MATCH) {
              if (j + t >= la_lo) la[j + t] = s_xl[p + t];
              nc4[r] = (float)((double)s_xc[p + t] + symbolcost258);
            }
          }
#pragma unroll
          for (int s = 0; s < 6; ++s) { c[s] = 1e30f; l[s] = 0; }
#pragma unroll
          for (int r = 0; r < 5; ++r) {
            const u32 t = 64u * r + lane;
            if (t < ZMX_MAX_MATCH) { c[r] = nc4[r]; l[r] = j + t + 1; }
          }
          wave_lds_sync();
          wbase = j + ZMX_MAX_MATCH;           // the registers now sit there
          reach = ZMX_MAX_MATCH - 1;
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
#pragma unroll
        for (int s = 0; s < 6; ++s) {
          if ((u32)s <= smax) {
            const u32 k1 = km1 + 64u * s;
            if (k1 < ke) {
              const double w = rows[ro + k1];
              const double mcl = k1 == 0 ? -kInf : mincost;
              DP_RELAX(c[s], l[s], w, mcl)
            }
          }
        }
        noshort = false;
        ++n_slow;
      }
    }
    if (jumped) continue;
    // ---- cells wbase .. wbase + 31 are final
    {
      const u32 jj = wbase + lane;
      if (lane < 32 && jj >= la_lo && jj <= B) la[jj] = (u16)(l[0] ? jj + 1 - l[0] : 0u);
      D4_TRACK_MAX()
      D3_ROT32()
      wbase += 32;
    }
  }
  if (J.la_lo == 1 && lane == 0) la[0] = 0;   // the head of the block
  if (J.exit) {
#pragma unroll
    for (int s = 0; s < 6; ++s) {
      J.exit->c[64u * s + lane] = c[s];
      J.exit->l[64u * s + lane] = l[s];
      vmax = fmaxf(vmax, c[s] < 1e29f ? c[s] : 0.0f);
    }
    vmax = wave_max_f32(vmax);
    if (lane == 0) { J.exit->vmax = vmax; J.exit->base = wbase; J.exit->noshort = noshort ? 1u : 0u; }
  }
  if (PROF && P.prof && lane == 0) {
    u64* o = P.prof + (u64)b * ZMX_PROF_N;
    atomicAdd(&o[1], (u64)__builtin_readcyclecounter() - t_begin);
    atomicAdd(&o[2], n_fast); atomicAdd(&o[3], n_slow); atomicAdd(&o[4], n_fast + n_slow);
    atomicAdd(&o[5], (u64)__builtin_readcyclecounter() - t_begin); atomicAdd(&o[6], n_fast);
    atomicAdd(&o[14], n_slow);
  }
}

template <bool PROF>
__global__ __launch_bounds__(64) void k_dp5_spec(Dp4Params P) {
  __shared__ float s_xc[DP_XN];
  __shared__ u16 s_xl[DP_XN];
  const u32 t = P.order[P.task0 + blockIdx.x];
  const SegTask T = P.tasks[t];
  const BlockDesc bd = P.blocks[T.block];
  const u32 B = (u32)(bd.inend - bd.instart);
  if (B == 0) return;
  D4Job J;
  J.start = T.q;
  J.noshort = 0;
  J.pout = T.pout;
  J.pend = T.pend;
  J.load = false;
  J.delta = 0;
  J.init = nullptr;
  J.entry = &P.entry[t];
  J.exit = &P.exit[t];
  if (T.pout == 0) {       // the head of the block
    J.spec = false;
    J.la_lo = 1;
    J.level = 0.0f;
  } else {
    J.spec = true;
    J.la_lo = SEG_NONE;
    J.level = P.est_bits ? P.est_bits[T.block] * ((float)T.q / (float)B) : P.lvl[t];
    J.level *= P.level_scale;
    if (!(J.level >= 16.0f)) J.level = 16.0f;
    if (P.est_bits && threadIdx.x == 0) P.lvl[t] = J.level;
  }
  d5_run_job<PROF>(P, J, T.block, bd, s_xc, s_xl);
}

// k_sq: GetBestLengths (squeeze.c:217-309) of one LZ77OptimalRun, edge costs and chain fused.
// Included only by zmx_hip.hip, after zmx_kernels.h and zmx_dp3.h.
//
// Same workgroup shape as k_dp3 — wave 0 runs the serial chain on cell registers, waves 1..3
// prepare ready-to-use 64-lane rows in double-buffered LDS tiles, all waves walk one
// deterministic sequence of steps, one s_barrier per step — but the producers compute the edge
// costs cost(k, sublen[k]) (squeeze.c:146-157) straight from the match records instead of
// copying them out of a precomputed array: no k_edges launch, no rows[] in HBM (8 B per edge
// written and read back every run), no LDS-DMA ring.  HBM traffic of a run is the match records
// (32 B), dph (8 B) and length_array (2 B) per position.
//
// A step is a run of positions of one 64-position group, cut at a long-run shortcut, at the
// group end, or when the cell registers beyond the two fixed tiles need more than SQ_X3 rows.
//   tile 1   row p            cell register 0 of position p (cells base + lane)
//   tile 2   row p & 31       cell register 1 for positions p >= 32
//   tile 3   SQ_X3 rows       every other register of positions with long matches
#pragma once

#define SQ_NP 3u
#define SQ_X3 16u

// per-lane data of the current group (lane = position base + lane)
struct SqGroup {
  u32 roff;                 // edge index of the position's first edge (k_rowscan)
  u32 kend;                 // 0 beyond the block end
  uint4 ra, rb;             // the match record
  u32 regs, x3;             // cell registers the position touches; rows it needs in tile 3
  u64 m_short, m_r1, m_bad; // flagged; needs register 1; cannot be in a fast block
  u32 navail;
};

struct SqWalk {
  u32 base = 0, q = 0;
  bool noshort = false, have_group = false;
  u32 pf_base = 0xffffffffu, pf_sel = 0;
  uint2 pf_ka = make_uint2(0, 0), pf_kb = make_uint2(0, 0);   // dph of the next group (A/B register sets, see D3Walk)
  uint4 pf_raa = make_uint4(0, 0, 0, 0), pf_rba = make_uint4(0, 0, 0, 0);
  uint4 pf_rab = make_uint4(0, 0, 0, 0), pf_rbb = make_uint4(0, 0, 0, 0);
};

struct SqStep {
  u32 base, q, n, event;    // D3_EV_NONE / D3_EV_SHORTCUT / D3_EV_GROUP_END
};

__device__ __forceinline__ void sq_load_group(SqWalk& W, SqGroup& G, const uint2* dbase, const u32* rbase, u32 B,
                                              u32 lane) {
  const u32 jj = W.base + lane;
  G.navail = (B - W.base < 64u) ? B - W.base : 64u;   // W.base <= B
  const bool act = lane < G.navail;
  const u32 cur = jj < B ? jj : B - 1, nxt = jj + 64 < B ? jj + 64 : B - 1;
  uint2 ky;
  if (W.pf_sel == 0) {
    ky = W.pf_ka; G.ra = W.pf_raa; G.rb = W.pf_rba;
    if (W.pf_base != W.base) {
      ky = dbase[cur];
      G.ra = *reinterpret_cast<const uint4*>(rbase + (u64)cur * 8);
      G.rb = *reinterpret_cast<const uint4*>(rbase + (u64)cur * 8 + 4);
    }
    W.pf_kb = dbase[nxt];
    W.pf_rab = *reinterpret_cast<const uint4*>(rbase + (u64)nxt * 8);
    W.pf_rbb = *reinterpret_cast<const uint4*>(rbase + (u64)nxt * 8 + 4);
  } else {
    ky = W.pf_kb; G.ra = W.pf_rab; G.rb = W.pf_rbb;
    if (W.pf_base != W.base) {
      ky = dbase[cur];
      G.ra = *reinterpret_cast<const uint4*>(rbase + (u64)cur * 8);
      G.rb = *reinterpret_cast<const uint4*>(rbase + (u64)cur * 8 + 4);
    }
    W.pf_ka = dbase[nxt];
    W.pf_raa = *reinterpret_cast<const uint4*>(rbase + (u64)nxt * 8);
    W.pf_rba = *reinterpret_cast<const uint4*>(rbase + (u64)nxt * 8 + 4);
  }
  W.pf_sel ^= 1;
  W.pf_base = W.base + 64;
  G.roff = ky.x;
  G.kend = act ? (ky.y & 0xffffu) : 0u;
  G.regs = act ? ((G.kend + lane) >> 6) + 1 : 0u;
  G.x3 = act ? (lane >= 32 ? (G.regs > 2 ? G.regs - 2 : 0u) : G.regs - 1) : 0u;
  G.m_short = __ballot(act && (ky.y >> 16) != 0);
  G.m_r1 = __ballot(G.regs >= 2);
  G.m_bad = G.m_short | __ballot(G.x3 != 0);
  W.have_group = true;
}

// The next step of the walk (identical in every producer wave); x3_excl = tile-3 row of each lane's position.
__device__ __forceinline__ SqStep sq_next(SqWalk& W, SqGroup& G, const uint2* dbase, const u32* rbase, u32 B, u32 lane,
                                          u32& x3_base) {
  if (!W.have_group) { sq_load_group(W, G, dbase, rbase, B, lane); W.q = 0; }
  u64 ms = W.q < 64 ? G.m_short & ~((1ull << W.q) - 1) : 0ull;
  if (W.noshort) ms &= ~(1ull << W.q);               // squeeze.c:273: not tested again right after a shortcut
  const u32 stop = ms ? (u32)__ffsll((long long)ms) - 1 : 64u;
  const u32 limit = stop < G.navail ? stop : G.navail;
  SqStep S;
  S.base = W.base; S.q = W.q; S.n = 0;
  const u32 X = wave_scan_add(G.x3);
  const u32 Xq = W.q ? rdlane_u32(X, W.q - 1) : 0u;
  x3_base = X - G.x3 - Xq;
  if (W.q < limit) {
    const u64 fit = __ballot(lane >= W.q && lane < limit && X - Xq <= SQ_X3);
    S.n = (u32)__popcll(fit);
  }
  if (ms && W.q + S.n == stop) {                     // a flagged position follows
    S.event = D3_EV_SHORTCUT;
    W.base = W.base + stop + ZMX_MAX_MATCH;
    W.noshort = true;
    W.have_group = false;
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

template <bool PROF>
__global__ __launch_bounds__(64 * (SQ_NP + 1)) void k_sq(DpParams P) {
  __shared__ __align__(16) double s_t1[2][64 * 64];
  __shared__ __align__(16) double s_t2[2][32 * 64];
  __shared__ __align__(16) double s_t3[2][SQ_X3 * 64];
  __shared__ double s_ll[288];
  __shared__ double s_d[32];
  __shared__ double s_kll[260];              // ll[length symbol of k]
  __shared__ u8 s_klb[260];                  // length extra bits of k
  __shared__ __align__(16) u32 s_hdr[SQ_NP][64][8];   // producer scratch: decoded record of each position of the group
  __shared__ u32 s_eoff[SQ_NP][64];          // producer scratch: first edge of each position of the group
  __shared__ u32 s_x3[SQ_NP][64];            // producer scratch: tile-3 row of each position (this step)
  __shared__ u32 s_badblk[2];                // per tile buffer: bit i = block i of the step has an edge below mincost
  __shared__ u32 s_desc[2][8];               // [0] q | n << 8 | event << 16 | last << 24 [1] base [2..3] m_r1 [4..5] m_bad
  __shared__ u32 s_tabc[2][64];              // per tile buffer: kend | tile-3 row << 16 of every position (generic path)
  __shared__ float s_xc[DP_XN];
  __shared__ u16 s_xl[DP_XN];

  const u32 tid = threadIdx.x;
  const u32 wave = (u32)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
  const u32 lane = tid & 63;
  const u32 b = P.block0 + blockIdx.x;
  const BlockDesc bd = P.blocks[b];
  const u32 B = (u32)(bd.inend - bd.instart);
  if (B == 0) return;
  const uint2* dbase = P.dph + bd.pos_off;
  const u32* rbase = P.recs + bd.pos_off * 8;
  u16* la = P.la + bd.la_off;
  const double mincost = P.mincost[b];
  const double kInf = __longlong_as_double(0x7ff0000000000000ll);

  for (u32 i = tid; i < 288; i += blockDim.x) s_ll[i] = P.cost[(u64)b * 320 + i];
  if (tid < 32) s_d[tid] = P.cost[(u64)b * 320 + 288 + tid];
  if (tid < 2) s_badblk[tid] = 0;
  __syncthreads();
  for (u32 k = tid; k < 260; k += blockDim.x) {
    const bool ok = k >= 3 && k <= ZMX_MAX_MATCH;
    s_kll[k] = ok ? s_ll[dev_length_symbol(k)] : 0.0;
    s_klb[k] = ok ? (u8)dev_length_extra_bits(k) : (u8)0;
  }
  // squeeze.c:260: cost of (length 258, dist 1) = (0 + 0) + ll[285] + d[0]
  const double symbolcost258 = (double)(0 + 0) + s_ll[285] + s_d[0];
  __syncthreads();

  if (wave == 0) {
    // ================================================================= the chain
    float c[6];
    u32 l[6];
#pragma unroll
    for (int s = 0; s < 6; ++s) { c[s] = 1e30f; l[s] = 0; }
    if (lane == 0) c[0] = 0.0f;
    u64 t_work = 0, n_fast = 0, n_slow = 0, n_steps = 0;

    __syncthreads();   // iteration 0: the producers' first step, nothing to consume yet
    u32 it = 1;
    for (;;) {
      const u64 tw0 = PROF ? (u64)__builtin_readcyclecounter() : 0ull;
      u32 dv[6];
#pragma unroll
      for (int i = 0; i < 6; ++i) dv[i] = s_desc[(it - 1) & 1][i];
#pragma unroll
      for (int i = 0; i < 6; ++i) dv[i] = (u32)__builtin_amdgcn_readfirstlane((int)dv[i]);
      const u32 sq = dv[0] & 255u, sn = (dv[0] >> 8) & 255u, event = (dv[0] >> 16) & 255u;
      const bool last = (dv[0] >> 24) != 0;
      const u32 base = dv[1];
      const u64 m_r1 = ((u64)dv[3] << 32) | dv[2];
      const u64 m_bad = ((u64)dv[5] << 32) | dv[4];
      const u32* tabc = s_tabc[(it - 1) & 1];
      const double* t1 = s_t1[(it - 1) & 1];
      const double* t2 = s_t2[(it - 1) & 1];
      const double* t3 = s_t3[(it - 1) & 1];
      u32 badblk = 0;
      if (sn) {
        badblk = (u32)__builtin_amdgcn_readfirstlane((int)s_badblk[(it - 1) & 1]);
        if (lane == 0) s_badblk[(it - 1) & 1] = 0;   // the producers OR into it again two steps from now
      }
// a block with a register-1 position must lie in lanes 32..63: tile 2 has one row per such position
#define SQ_FAST(P0) ((P0) + 8 <= sq + sn && ((u32)(m_bad >> (P0)) & 255u) == 0 && \
                     ((P0) >= 32 || ((u32)(m_r1 >> (P0)) & 255u) == 0))
      u32 p0 = sq;
      u32 bi = 0;   // block index within the step
      for (; p0 < sq + sn; ++bi) {
        // four single-register blocks in one go (the usual start of a group)
        if (p0 + 32 <= sq + sn && ((u32)(m_bad >> p0)) == 0 && ((badblk >> bi) & 15) == 0 &&
            ((u32)(m_r1 >> p0)) == 0) {
          double w0[32];
#pragma unroll
          for (int u = 0; u < 32; ++u) w0[u] = t1[(p0 + u) * 64 + lane];
#pragma unroll
          for (int u = 0; u < 32; ++u) {
            const u32 p = p0 + u;
            const double cj = (double)rdlane_f32(c[0], p);
            const u32 src1 = base + p + 1;
            D3_RELAX(c[0], l[0], w0[u])
          }
          n_fast += 32;
          p0 += 32;
          bi += 3;
          continue;
        }
        // two single-register blocks in one go
        if (SQ_FAST(p0) && SQ_FAST(p0 + 8) && ((badblk >> bi) & 3) == 0 && ((u32)(m_r1 >> p0) & 0xffffu) == 0) {
          double w0[16];
#pragma unroll
          for (int u = 0; u < 16; ++u) w0[u] = t1[(p0 + u) * 64 + lane];
#pragma unroll
          for (int u = 0; u < 16; ++u) {
            const u32 p = p0 + u;
            const double cj = (double)rdlane_f32(c[0], p);
            const u32 src1 = base + p + 1;
            D3_RELAX(c[0], l[0], w0[u])
          }
          n_fast += 16;
          p0 += 16;
          ++bi;
          continue;
        }
        if (SQ_FAST(p0) && !((badblk >> bi) & 1)) {
          const bool two = ((u32)(m_r1 >> p0) & 255u) != 0;
          double w0[8], w1[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) w0[u] = t1[(p0 + u) * 64 + lane];
          if (two) {
#pragma unroll
            for (int u = 0; u < 8; ++u) w1[u] = t2[((p0 + u) & 31) * 64 + lane];
          }
          if (!two) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              const u32 p = p0 + u;
              const double cj = (double)rdlane_f32(c[0], p);
              const u32 src1 = base + p + 1;
              D3_RELAX(c[0], l[0], w0[u])
            }
          } else {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              const u32 p = p0 + u;
              const double cj = (double)rdlane_f32(c[0], p);
              const u32 src1 = base + p + 1;
              D3_RELAX(c[0], l[0], w0[u])
              D3_RELAX(c[1], l[1], w1[u])
            }
          }
          n_fast += 8;
          p0 += 8;
          continue;
        }
        // generic path (ragged tails, long matches, exempt flagged positions, blocks with an edge
        // below mincost): the reference's tests, literally, on the same ready-made rows
        const u32 pend = p0 + 8 <= sq + sn ? p0 + 8 : sq + sn;
        for (u32 p = p0; p < pend; ++p) {
          const u32 tc = (u32)__builtin_amdgcn_readfirstlane((int)tabc[p]);
          const u32 ke = tc & 0xffffu, x3b = tc >> 16;
          const double cj = (double)rdlane_f32(c[0], p);
          const u32 src1 = base + p + 1;
          const u32 nregs = ((ke + p) >> 6) + 1;
#pragma unroll
          for (int s = 0; s < 6; ++s) {
            if ((u32)s < nregs) {
              const double* row = s == 0 ? t1 + p * 64
                                : (s == 1 && p >= 32) ? t2 + (p & 31) * 64
                                : t3 + (x3b + (u32)s - (p >= 32 ? 2u : 1u)) * 64;
              const double w = row[lane];
              const double mcl = lane + 64u * s == p + 1 ? -kInf : mincost;   // the literal has no mincost test
              DP_RELAX(c[s], l[s], w, mcl)
            }
          }
        }
        n_slow += pend - p0;
        p0 = pend;
      }
      if (event == D3_EV_GROUP_END) {
        // cells base..base+63 are final
        const u32 jj = base + lane;
        if (jj <= B && jj >= 1) la[jj] = (u16)(l[0] ? jj + 1 - l[0] : 0u);
#pragma unroll
        for (int s = 0; s < 5; ++s) { c[s] = c[s + 1]; l[s] = l[s + 1]; }
        c[5] = 1e30f;
        l[5] = 0;
      } else if (event == D3_EV_SHORTCUT) {
        // long-run shortcut at position q + n of the group (squeeze.c:251-271)
        const u32 p = sq + sn;
        const u32 j = base + p;
        if (lane < p && base + lane >= 1) la[base + lane] = (u16)(l[0] ? base + lane + 1 - l[0] : 0u);
        wave_lds_sync();
#pragma unroll
        for (int s = 0; s < 6; ++s) {
          const u32 x = base + 64u * s + lane;
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
            la[j + t] = s_xl[p + t];
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
      }
      if (PROF) { t_work += (u64)__builtin_readcyclecounter() - tw0; ++n_steps; }
      __syncthreads();
      ++it;
      if (last) break;
    }
    if (lane == 0) la[0] = 0;
    if (PROF && P.prof && lane == 0) {
      u64* o = P.prof + (u64)b * ZMX_PROF_N;
      o[0] = n_steps; o[1] = t_work; o[2] = n_fast; o[3] = n_slow; o[4] = B; o[5] = 0; o[6] = 0; o[7] = 0;
    }
    return;
  }

  // =================================================================== producers
  const u32 my = wave - 1;
  SqWalk W;
  SqGroup G;
  G.kend = 0; G.regs = G.x3 = 0; G.m_short = G.m_r1 = G.m_bad = 0; G.navail = 0;
  G.ra = G.rb = make_uint4(0, 0, 0, 0);
  u32 it = 0;
  bool more = true;
  u64 tp_next = 0, tp_dec = 0, tp_fill = 0, tp_pass = 0, tp_bar = 0, n_pass = 0;
#define SQ_TICK() (PROF ? (u64)__builtin_readcyclecounter() : 0ull)
  while (more) {
    const bool fresh = !W.have_group;
    u32 x3_base = 0;
    const u64 tk0 = SQ_TICK();
    const SqStep S = sq_next(W, G, dbase, rbase, B, lane, x3_base);
    more = W.base <= B;
    const u64 tk1 = SQ_TICK();
    u64 tk2 = tk1, tk3 = tk1, tk4 = tk1;
    if (wave == 1) {   // describe the step for the chain wave
      if (S.n) s_tabc[it & 1][lane] = G.kend | (x3_base << 16);
      if (lane == 0) {
        u32* d = s_desc[it & 1];
        d[0] = S.q | (S.n << 8) | (S.event << 16) | ((more ? 0u : 1u) << 24);
        d[1] = S.base;
        d[2] = (u32)G.m_r1; d[3] = (u32)(G.m_r1 >> 32);
        d[4] = (u32)G.m_bad; d[5] = (u32)(G.m_bad >> 32);
      }
    }
    if (S.n) {
      if (fresh) {
        // decode my copy of the group's records once: thresholds (len - 3, ascending, 0xff padded)
        // and distances of the <= 8 sublen change points (the reference's LMC format, cache.c:54)
        wave_lds_sync();
        u32* h = s_hdr[my][lane];
        const u32 lit = (G.ra.y >> 16) & 255u, ncpf = G.ra.y >> 24;
        s_eoff[my][lane] = G.roff;
        h[0] = G.kend | (lit << 16) | (ncpf << 24);
        if (ncpf == 0xffu) {
          h[1] = G.ra.z;              // pool offset
          h[2] = G.ra.w & 0xffffu;    // pool count
        } else {
          const u32 w[6] = {G.ra.z, G.ra.w, G.rb.x, G.rb.y, G.rb.z, G.rb.w};
          u32 t0 = 0, t1v = 0, dd[4] = {0, 0, 0, 0};
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const u32 bit = 24u * e;
            const u32 lo32 = w[bit >> 5] >> (bit & 31);
            const u32 v = (bit & 31) > 8 ? (lo32 | (w[(bit >> 5) + 1 > 5 ? 5 : (bit >> 5) + 1] << (32 - (bit & 31)))) : lo32;
            const u32 thr = (u32)e < ncpf ? (v & 255u) : 255u;
            if (e < 4) t0 |= thr << (8 * e); else t1v |= thr << (8 * (e - 4));
            dd[e >> 1] |= ((v >> 8) & 0xffffu) << (16 * (e & 1));
          }
          h[1] = t0; h[2] = t1v;
          h[3] = dd[0]; h[4] = dd[1]; h[5] = dd[2]; h[6] = dd[3];
        }
        wave_lds_sync();
      }
      wave_lds_sync();
      tk2 = SQ_TICK();
      s_x3[my][lane] = x3_base;   // tile-3 rows are numbered per step
      wave_lds_sync();
      double* t1 = s_t1[it & 1];
      double* t2 = s_t2[it & 1];
      double* t3 = s_t3[it & 1];
      // my contiguous share of the step's positions
      const u32 pa = S.q + S.n * my / SQ_NP, pb = S.q + S.n * (my + 1) / SQ_NP;
      if (pa < pb) {
        // 1. every row of my positions = +inf (one wave-wide store per row)
        for (u32 p = pa; p < pb; ++p) {
          t1[p * 64 + lane] = kInf;
          if (p >= 32) t2[(p & 31) * 64 + lane] = kInf;
        }
        {
          const u32 xa = rdlane_u32(x3_base, pa), xb = rdlane_u32(x3_base, pb - 1) + rdlane_u32(G.x3, pb - 1);
          for (u32 x = xa; x < xb; ++x) t3[x * 64 + lane] = kInf;
        }
        // 2. one lane per edge (squeeze.c:146-157), scattered to lane (p + k) & 63 of its row
        const u32 e_lo = rdlane_u32(G.roff, pa);
        const u32 e_hi = rdlane_u32(G.roff, pb - 1) + rdlane_u32(G.kend, pb - 1);
        // first edges of my (at most 22) positions, as wave-uniform values: the owner of an edge is
        // found by counting, not by a chain of dependent LDS reads
        u32 eo[22];
#pragma unroll
        for (int i = 0; i < 22; ++i) eo[i] = s_eoff[my][pa + i < 63 ? pa + i : 63];
        const u32 npos = pb - pa;
        wave_lds_sync();
        tk3 = SQ_TICK();
        for (u32 e0 = e_lo; e0 < e_hi; e0 += 64) {
          if (PROF) ++n_pass;
          const u32 e = e0 + lane;
          if (e < e_hi) {
            u32 p = pa;
#pragma unroll
            for (int i = 1; i < 22; ++i) p += ((u32)i < npos && eo[i] <= e) ? 1u : 0u;
            const u32 k = e - s_eoff[my][p] + 1;
            const uint4 ha = *reinterpret_cast<const uint4*>(s_hdr[my][p]);
            const u32 lit = (ha.x >> 16) & 255u, ncpf = ha.x >> 24;
            if (k != 2) {   // k = 2 is the dead slot of the row
              double w;
              if (k == 1) {
                w = s_ll[lit];                                         // literal edge, squeeze.c:278
              } else {
                u32 dist;
                if (ncpf != 0xffu) {
                  const uint4 hb = *reinterpret_cast<const uint4*>(s_hdr[my][p] + 4);
                  const u32 x = k - 3;
                  // first change point with len >= k: thresholds ascending, binary search over 8 bytes
                  u32 idx = ((ha.y >> 24) < x) ? 4u : 0u;
                  u32 half = idx ? ha.z : ha.y;
                  if (((half >> 8) & 255u) < x) { idx += 2; half >>= 16; }
                  if ((half & 255u) < x) idx += 1;
                  const u32 dw = idx < 2 ? ha.w : idx < 4 ? hb.x : idx < 6 ? hb.y : hb.z;
                  dist = (dw >> (16 * (idx & 1))) & 0xffffu;
                } else {
                  u32 qlo = 0, qhi = ha.z;   // first pool entry with len >= k (entries ascending in len)
                  while (qlo < qhi) {
                    const u32 mid = (qlo + qhi) >> 1;
                    if ((P.pool[ha.y + mid] & 0xffffu) < k) qlo = mid + 1; else qhi = mid;
                  }
                  dist = qlo < ha.z ? P.pool[ha.y + qlo] >> 16 : 1u;
                }
                // squeeze.c:155: (lbits + dbits) as int, then + ll, then + d
                w = ((double)((int)s_klb[k] + dev_dist_extra_bits(dist)) + s_kll[k]) + s_d[dev_dist_symbol(dist)];
                if (w < mincost) atomicOr(&s_badblk[it & 1], 1u << ((p - S.q) >> 3));
              }
              const u32 cell = p + k, sreg = cell >> 6;
              double* row = sreg == 0 ? t1 + p * 64
                          : (sreg == 1 && p >= 32) ? t2 + (p & 31) * 64
                          : t3 + (s_x3[my][p] + sreg - (p >= 32 ? 2u : 1u)) * 64;
              row[cell & 63] = w;
            }
          }
        }
      }
    }
    tk4 = SQ_TICK();
    if (tk3 == tk1) tk3 = tk4;
    if (tk2 == tk1) tk2 = tk1;
    __syncthreads();
    if (PROF) {
      const u64 tk5 = SQ_TICK();
      tp_next += tk1 - tk0; tp_dec += tk2 - tk1; tp_fill += tk3 - tk2; tp_pass += tk4 - tk3; tp_bar += tk5 - tk4;
    }
    ++it;
  }
  __syncthreads();   // the chain wave's last step
  if (PROF && P.prof && wave == 1 && lane == 0) {
    u64* o = P.prof + (u64)b * ZMX_PROF_N + 8;
    o[0] = tp_next; o[1] = tp_dec; o[2] = tp_fill; o[3] = tp_pass; o[4] = tp_bar; o[5] = n_pass;
  }
#undef SQ_TICK
}

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
//     lane l of row u  =  wtab[codes[roff_u + l - u - 1]]   if 0 <= l - u - 1 < kend_u, else +inf
//
// with everything per position on the scalar side: a ready-made buffer descriptor whose buffer IS
// the row's 16-bit weight codes (base codes + roff, kend * 2 bytes; k_mkdesc) arrives by s_load
// (uniform address); lane l asks for offset 2 (l - u - 1), which wraps below the row and overshoots
// beyond it, so the hardware's range check drops those lanes before they reach the L1 (a plain
// 64-lane load costs an L1 access per 32 bytes of lanes whatever they point at: measured, the first
// version was L1-bound) and returns zero for them — the code of "no edge", weight +inf.  The weight
// itself comes from the run's table (k_wtab) that the four waves of the workgroup, four tasks of one
// block, share in LDS: one buffer_load_ushort, one ds_read_b64 and one VALU instruction per position,
// then the 8-instruction chain step of k_dp4.  Other windows (long matches, shortcut flags, edges below mincost, ragged
// tails) take the generic path, position by position, with the reference's tests literally.
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
typedef const __attribute__((address_space(4))) d5_u32x4* d5_cdscp;

// One buffer descriptor per block position whose buffer IS the position's row of weight codes: base =
// its first slot in codes[], kend * 2 bytes long (k_mkdesc, once per table build).
struct MkDescParams {
  const BlockDesc* blocks;
  const uint2* dph;
  const u64* code_base;
  const u16* codes;
  d5_u32x4* dsc;
  const u32* win_off;   // [nb] first entry of each block in winflag[]
  u32* winflag;         // per 32-position window of a block (aligned to the block start): 1 = 32 positions, none flagged
                        // for the long-run shortcut, none with an edge beyond cell register 0 of a window at that
                        // place; 2 = 32 positions, none flagged, edges of any length; 0 = neither
};

__global__ __launch_bounds__(256) void k_mkdesc(MkDescParams P) {
  const BlockDesc bd = P.blocks[blockIdx.y];
  const u32 B = (u32)(bd.inend - bd.instart);
  const u32 p = blockIdx.x * 256u + threadIdx.x;
  const u32 lane = threadIdx.x & 63;
  uint2 dhw = make_uint2(0, 0);
  if (p < B) dhw = P.dph[bd.pos_off + p];
  // (whole waves take part in the ballots; a wave covers two windows)
  const bool flagged = p >= B || (dhw.y >> 16) != 0;
  const u64 not1 = __ballot(flagged || (dhw.y & 0xffffu) + (lane & 31u) >= 64u);    // an edge beyond cell register 0
  const u64 not2 = __ballot(flagged);                                                // flagged for the shortcut, or short
  if ((lane & 31u) == 0 && p < B) {
    const u32 sh = lane & 32u;
    P.winflag[P.win_off[blockIdx.y] + (p >> 5)] = (u32)(not1 >> sh) == 0 ? 1u : (u32)(not2 >> sh) == 0 ? 2u : 0u;
  }
  if (p >= B) return;
  const uint2 dh = dhw;
  const u64 a = reinterpret_cast<u64>(P.codes + P.code_base[blockIdx.y] + dh.x);
  d5_u32x4 d;
  d.x = (u32)a;
  d.y = (u32)(a >> 32) & 0xffffu;     // stride 0
  d.z = (dh.y & 0xffffu) * 2u;        // bytes in the row
  d.w = 0x00020000u;                  // raw 32-bit data format
  P.dsc[bd.pos_off + p] = d;
}

struct D5Cls {          // one window on the generic path: lane l < 32 = position wbase + l
  u32 kend, roff;
  u64 ms;               // flagged for the long-run shortcut
  u32 nav;
};

template <bool PROF>
__device__ __forceinline__ void d5_run_job(const Dp4Params& P, const D4Job& J, u32 b, const BlockDesc& bd,
                                           const double (&s_wtab)[ZMX_WTAB], float (&s_xc)[DP_XN], u16 (&s_xl)[DP_XN]) {
  const u32 lane = threadIdx.x & 63;
  const u32 lane2 = lane * 2u;
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
  const d5_u32x4* __restrict__ dsc = uniform_ptr(static_cast<const d5_u32x4*>(P.dsc) + bd.pos_off);
  typedef const __attribute__((address_space(4))) u32* cu32p;
  const cu32p winflag = (cu32p)uniform_ptr(P.winflag + P.win_off[b]);
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
    if (lane == 0) c[0] = J.level;
    reach = 0;
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
  u32 wbase = (u32)__builtin_amdgcn_readfirstlane((int)J.start);
  bool noshort = J.noshort != 0;
  u64 n_fast = 0, n_slow = 0;
  const u64 t_begin = PROF ? (u64)__builtin_readcyclecounter() : 0ull;

  // weights of the rows of positions WB + U0 .. WB + U0 + 15 into WV: the row's codes are the buffer;
  // lanes below u + 1 (the offset wraps) and beyond u + kend are out of range, never reach the cache
  // and come back as code 0 = +inf
#define D5_ISSUE(WV, WB, U0)                                                                      \
  {                                                                                               \
    const d5_cdscp dw_ = (d5_cdscp)(dsc + (WB) + (U0));     /* uniform: s_load */                  \
    u32 cd_[16];                                                                                  \
    _Pragma("unroll") for (int u = 0; u < 16; ++u) {                                              \
      const d5_u32x4 d_ = dw_[u];                                                                 \
      const __amdgpu_buffer_rsrc_t rs_ = __builtin_amdgcn_make_buffer_rsrc(                       \
          reinterpret_cast<void*>(((u64)d_.y << 32) | d_.x), (short)0, (int)d_.z, (int)d_.w);     \
      cd_[u] = (u32)(u16)__builtin_amdgcn_raw_buffer_load_b16(rs_, (int)(lane2 - 2u * (u32)((U0) + u + 1)), 0, 0); \
    }                                                                                             \
    _Pragma("unroll") for (int u = 0; u < 16; ++u) WV[u] = code_w(cd_[u]);                        \
  }
  // the chain over positions U0 .. U0 + 15 of the window at wbase
#define D5_CHAIN(WV, U0)                                                                          \
  {                                                                                               \
    _Pragma("unroll") for (int u = 0; u < 16; ++u) {                                              \
      const double cj = (double)rdlane_f32(c[0], (u32)((U0) + u));                                \
      D3_RELAX_K(c[0], lt_, WV[u], (u32)((U0) + u + 1))                                           \
    }                                                                                             \
  }
  // a window that can take the fast path: at a multiple of 32 from the block start, statically
  // clean (k_mkdesc) and without an edge below mincost in this run (k_edges' bitmap)
  auto win_kind = [&](u32 wb) -> u32 {     // 1 / 2: cell registers the window's edges reach; 0: generic
    if ((wb & 31u) != 0 || wb + 32u > B) return 0u;
    const u32 f = winflag[wb >> 5];
    if (f == 0) return 0u;
    const u32 g = bit_off + wb;
    const u64 two = ((u64)badw[(g >> 5) + 1] << 32) | badw[g >> 5];
    return (u32)(two >> (g & 31u)) == 0 ? f : 0u;
  };
  while (wbase < J.pend) {          // (J.pend = B + 1 on the last task: the window at B retires cell B)
    wbase = (u32)__builtin_amdgcn_readfirstlane((int)wbase);
    if (J.spec && la_lo == SEG_NONE && wbase >= J.pout) {
      // the first window at or after pout: from here on the task owns the length_array; what the
      // registers hold now is compared with the predecessor's exit state
#pragma unroll
      for (int s = 0; s < 6; ++s) { J.entry->c[64u * s + lane] = c[s]; J.entry->l[64u * s + lane] = l[s]; }
      if (lane == 0) { J.entry->base = wbase; J.entry->noshort = noshort ? 1u : 0u; }
      la_lo = wbase;
      vmax = 0.0f;
    }
    bool jumped = false;
    const u32 kind = win_kind(wbase);
    if (kind == 1) {
      // ---- 32 positions, one cell register, no flags: the rows' weights come straight into registers,
      //      then the chain.  The other waves of the SIMD run while this one waits for its rows.
      //      Requesting the next window's codes half a window ahead of the chain was measured SLOWER,
      //      twice: with the registers for three waves per SIMD (dp 97 vs 91 ms per 15 runs) and, spilling,
      //      for four (119 ms); six waves per SIMD spill as well (the D5W = 6 build: 2.6x slower).
      double wv0[16], wv1[16];
      D5_ISSUE(wv0, wbase, 0)
      D5_ISSUE(wv1, wbase, 16)
      u32 lt_ = 0;                             // 1 + index of the last position that updated the cell
      D5_CHAIN(wv0, 0)
      D5_CHAIN(wv1, 16)
      l[0] = lt_ ? wbase + lt_ : l[0];
      reach = reach > 63 ? reach : 63;
      noshort = false;
      n_fast += 32;
    } else if (kind == 2) {
      // ---- the same with longer edges (matches of more than 32 bytes: markup, source code): a second
      //      request per row for lanes 64 .. 127 of it and a second relaxation off the chain's critical
      //      path (register 1 is never the source of a position of this window); the few rows that
      //      reach further fetch the rest on demand.  No edge of the window lies below mincost
      //      (win_kind), so squeeze.c:293's test is a no-op here as well.
      u32 lt_ = 0, lt1_ = 0;
#pragma unroll 1
      for (int h = 0; h < 4; ++h) {          // (eight positions at a time, a real loop: the registers of four waves per SIMD)
        const d5_cdscp dw_ = (d5_cdscp)(dsc + wbase + 8u * h);
        double wa[8], wb[8];
        u32 ca[8], cb[8];
        u32 ke8[8];
        u64 ra8[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const d5_u32x4 d_ = dw_[u];
          ke8[u] = d_.z >> 1;
          ra8[u] = ((u64)d_.y << 32) | d_.x;
          const __amdgpu_buffer_rsrc_t rs_ = __builtin_amdgcn_make_buffer_rsrc(
              reinterpret_cast<void*>(ra8[u]), (short)0, (int)d_.z, (int)d_.w);
          const int vo = (int)(lane2 - 2u * (u32)(8 * h + u + 1));
          ca[u] = (u32)(u16)__builtin_amdgcn_raw_buffer_load_b16(rs_, vo, 0, 0);
          cb[u] = (u32)(u16)__builtin_amdgcn_raw_buffer_load_b16(rs_, vo + 128, 0, 0);
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
      }
      if (noshort) W.ms &= ~1ull;      // squeeze.c:273: the position right after a shortcut is not tested again
      for (u32 p = 0; p < W.nav; ++p) {
        const u32 j = wbase + p;
        if ((W.ms >> p) & 1) {
          // long-run shortcut at position j (squeeze.c:251-271)
          if (lane < p && wbase + lane >= la_lo) put_la(wbase + lane, (u16)(l[0] ? wbase + lane + 1 - l[0] : 0u));
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
              if (j + t >= la_lo) put_la(j + t, s_xl[p + t]);
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
        if (smax < 2) {
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
      }
    }
    if (jumped) continue;
    // ---- cells wbase .. wbase + 31 are final
    {
      const u32 jj = wbase + lane;
      if (lane < 32 && jj >= la_lo && jj <= B) put_la(jj, (u16)(l[0] ? jj + 1 - l[0] : 0u));
      D4_TRACK_MAX()
      D3_ROT32()
      wbase += 32;
    }
  }
#undef D5_ISSUE
#undef D5_CHAIN
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
    const u64 dt = (u64)__builtin_readcyclecounter() - t_begin;
    atomicMax(&o[7], dt);                               // the longest task of the block
    if (J.la_lo == 1) atomicAdd(&o[8], dt);             // the head task
  }
}

// One workgroup = four waves = up to four tasks of ONE block (P.wg_tasks), sharing the run's weight
// table in LDS; after the table is in place the waves go their own ways.
#define D5_WG 4u
template <bool PROF, int WAVES>
__global__ __launch_bounds__(64 * D5_WG, WAVES) void k_dp5_spec(Dp4Params P) {
  __shared__ __align__(16) double s_wtab[ZMX_WTAB];
  __shared__ float s_xc[D5_WG][DP_XN];
  __shared__ u16 s_xl[D5_WG][DP_XN];
  const u32 wave = (u32)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (P.redo_pass && blockIdx.x >= *P.redo_count) return;
  const u32* wg = (P.redo_pass ? P.redo_wg : P.wg_tasks) + (u64)(P.task0 + blockIdx.x) * D5_WG;
  const u32 t0 = wg[0];
  const u32 b0 = P.tasks[t0].block;
  for (u32 i = threadIdx.x; i < ZMX_WTAB; i += 64 * D5_WG) s_wtab[i] = P.wtab[(u64)b0 * ZMX_WTAB + i];
  __syncthreads();
  const u32 t = wg[wave];
  if (t == SEG_NONE) return;
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
  J.over_lo = T.pend <= B ? T.pend : SEG_NONE;
  J.over = P.over + (u64)t * SEG_OVER;
  if (T.pout == 0) {       // the head of the block
    J.spec = false;
    J.la_lo = 1;
    J.level = 0.0f;
  } else {
    J.spec = true;
    J.la_lo = SEG_NONE;
    J.level = P.est_bits ? P.est_bits[T.block] * ((float)T.q / (float)B) : P.lvl[t];
    if (!P.redo_pass) J.level *= P.level_scale;
    if (!(J.level >= 16.0f)) J.level = 16.0f;
    if (P.est_bits && (threadIdx.x & 63) == 0) P.lvl[t] = J.level;
  }
  d5_run_job<PROF>(P, J, T.block, bd, s_wtab, s_xc[wave], s_xl[wave]);
}

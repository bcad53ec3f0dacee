// Micro-benchmark of dependent-instruction latencies for the serial DP chain (one wave, gfx950).
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_chain.hip -o tools/ubench_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../zopfli_amd/csrc/device/zmx_kernels.h"
#define N 4096

__global__ void k_add_f64(double* out, u64* cyc, double x) {
  double a = out[threadIdx.x];
  u64 t0 = __builtin_readcyclecounter();
#pragma unroll 16
  for (int i = 0; i < N; ++i) a = a + x;
  u64 t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = a; if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_add_f32(float* out, u64* cyc, float x) {
  float a = out[threadIdx.x];
  u64 t0 = __builtin_readcyclecounter();
#pragma unroll 16
  for (int i = 0; i < N; ++i) a = a + x;
  u64 t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = a; if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
// readlane -> VALU (uses the SGPR) -> readlane ...
__global__ void k_readlane(float* out, u64* cyc, float x) {
  float a = out[threadIdx.x];
  u64 t0 = __builtin_readcyclecounter();
#pragma unroll 16
  for (int i = 0; i < N; ++i) {
    float s = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a), i & 63));
    a = s + x;
  }
  u64 t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = a; if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
// cmp f64 -> cndmask -> cmp ...
__global__ void k_cmp_sel(double* out, u64* cyc, double x, double y) {
  double a = out[threadIdx.x];
  u64 t0 = __builtin_readcyclecounter();
#pragma unroll 16
  for (int i = 0; i < N; ++i) a = a < x ? a * 1.0 + y : a;   // add + cmp + select
  u64 t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = a; if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
// the current k_dp step: readlane, cvt, 2 add, 2 cmp, s_and, cvt, 2 cndmask
__global__ void k_step_v1(float* out, u64* cyc, const double* w, double mincost) {
  float c = out[threadIdx.x];
  u32 l = 0;
  const double w0 = w[threadIdx.x], mcl = mincost;
  u64 t0 = __builtin_readcyclecounter();
#pragma unroll 8
  for (int i = 0; i < N; ++i) {
    const double cj = (double)__int_as_float(__builtin_amdgcn_readlane(__float_as_int(c), i & 63));
    const double old_ = (double)c, nc_ = w0 + cj;
    const bool upd = old_ > mcl + cj && nc_ < old_;
    c = upd ? (float)nc_ : c;
    l = upd ? (u32)i : l;
  }
  u64 t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = c + (float)l; if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
// same with max instead of two compares + s_and
__global__ void k_step_v2(float* out, u64* cyc, const double* w, double mincost) {
  float c = out[threadIdx.x];
  u32 l = 0;
  const double w0 = w[threadIdx.x], mcl = mincost;
  u64 t0 = __builtin_readcyclecounter();
#pragma unroll 8
  for (int i = 0; i < N; ++i) {
    const double cj = (double)__int_as_float(__builtin_amdgcn_readlane(__float_as_int(c), i & 63));
    const double old_ = (double)c, nc_ = w0 + cj;
    const bool upd = fmax(nc_, mcl + cj) < old_;
    c = upd ? (float)nc_ : c;
    l = upd ? (u32)i : l;
  }
  u64 t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = c + (float)l; if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
// uniform literal chain in doubles: add, cvt, cvt, cmp, cndmask (no readlane on the chain)
__global__ void k_lit_chain(double* out, u64* cyc, const double* pre, double lit) {
  double cj = out[0];
  const double p0 = pre[threadIdx.x];
  u64 t0 = __builtin_readcyclecounter();
#pragma unroll 8
  for (int i = 0; i < N; ++i) {
    const double nc = lit + cj;
    const double r = (double)(float)nc;
    cj = nc < p0 ? r : p0;
  }
  u64 t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = cj; if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
// LDS pointer chase
__global__ void k_lds_chase(u32* out, u64* cyc) {
  __shared__ u32 s[1024];
  for (int i = threadIdx.x; i < 1024; i += 64) s[i] = (i * 7 + 13) & 1023;
  __syncthreads();
  u32 a = threadIdx.x;
  u64 t0 = __builtin_readcyclecounter();
#pragma unroll 8
  for (int i = 0; i < N; ++i) a = s[a];
  u64 t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = a; if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
// independent f64 adds (throughput)
__global__ void k_add_f64_tp(double* out, u64* cyc, double x) {
  double a0 = out[threadIdx.x], a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  u64 t0 = __builtin_readcyclecounter();
#pragma unroll 4
  for (int i = 0; i < N / 8; ++i) { a0 += x; a1 += x; a2 += x; a3 += x; a4 += x; a5 += x; a6 += x; a7 += x; }
  u64 t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7; if (threadIdx.x == 0) cyc[0] = t1 - t0;
}


// the real fast block of k_dp on synthetic rows (ke = 8 everywhere): full / fetch only / chain only
template <int MODE>
__global__ void k_fast_block(float* out, u64* cyc, double mincost) {
  __shared__ double ring[DP_FRONT + DP_RING + DP_MIRROR];
  const u32 lane = threadIdx.x;
  for (u32 i = lane; i < DP_FRONT + DP_RING + DP_MIRROR; i += 64) ring[i] = 3.0 + (i & 7);
  __syncthreads();
  const u32 hdr_v = ((lane * 8) & (DP_RING - 1)) | (8u << 16);
  __shared__ uint2 tab[64];
  tab[lane] = make_uint2((((lane * 8) & (DP_RING - 1)) - lane - 1) * 8u, 8u);
  __syncthreads();
  float c0 = lane == 0 ? 0.0f : 1e30f, c1 = 1e30f;
  u32 l0 = 0, l1 = 0;
  u64 t0 = __builtin_readcyclecounter();
  for (int it = 0; it < N / 8; ++it) {
    const u32 p0 = (it & 7) * 8;
    if (MODE == 0) {
      dp_fast_block<false>(ring + DP_FRONT, tab, p0, lane, 0, mincost, c0, l0, c1, l1);
    } else if (MODE == 1) {   // fetch + masks only
      const double kInf = __longlong_as_double(0x7ff0000000000000ll);
      double acc = 0;
      const u32 d0 = lane - p0 - 1;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const uint2 t = tab[p0 + u];
        const double* row = reinterpret_cast<const double*>(reinterpret_cast<const char*>(ring + DP_FRONT) + (int)t.x);
        const u32 km1 = d0 - u;
        const double w = km1 < t.y ? row[lane] : kInf;
        const double m = km1 == 0 ? -kInf : mincost;
        acc = fmax(acc, fmin(w, m));
      }
      c0 = (float)acc;
    } else {                  // chain only
      const double w = 3.5, m = mincost;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const u32 p = p0 + u;
        const double cj = (double)rdlane_f32(c0, p);
        const u32 src1 = p + 1;
        DP_RELAX(c0, l0, w, m)
      }
    }
  }
  u64 t1 = __builtin_readcyclecounter();
  out[lane] = c0 + c1 + (float)(l0 + l1); if (lane == 0) cyc[0] = t1 - t0;
}
// independent integer VALU ops (issue rate of a lone wave)
__global__ void k_int_tp(u32* out, u64* cyc, u32 x) {
  u32 a0 = out[threadIdx.x], a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  u64 t0 = __builtin_readcyclecounter();
#pragma unroll 4
  for (int i = 0; i < N / 8; ++i) { a0 = a0 * 3 + x; a1 = a1 * 3 + x; a2 = a2 * 3 + x; a3 = a3 * 3 + x; a4 = a4 * 3 + x; a5 = a5 * 3 + x; a6 = a6 * 3 + x; a7 = a7 * 3 + x; }
  u64 t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7; if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
// independent readlanes feeding VALU
__global__ void k_readlane_tp(u32* out, u64* cyc) {
  u32 a = out[threadIdx.x], acc = 0;
  u64 t0 = __builtin_readcyclecounter();
#pragma unroll 8
  for (int i = 0; i < N; ++i) acc += rdlane_u32(a, i & 63) ^ i;
  u64 t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = acc; if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

// cvt round trip: f64 -> f32 -> f64 dependent
__global__ void k_cvt_rt(double* out, u64* cyc, double x) {
  double a = out[threadIdx.x];
  u64 t0 = __builtin_readcyclecounter();
#pragma unroll 16
  for (int i = 0; i < N; ++i) a = (double)(float)a;
  u64 t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = a + x; if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
// k_dp3's fast path: one position per step (prune-free relax), rows in registers
__global__ void k_step_d3(float* out, u64* cyc, const double* w) {
  float c = out[threadIdx.x];
  u32 l = 0;
  double w0[16];
#pragma unroll
  for (int u = 0; u < 16; ++u) w0[u] = w[(threadIdx.x + u) & 63] + u;
  u64 t0 = __builtin_readcyclecounter();
  for (int i = 0; i < N / 16; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const u32 p = (i * 16 + u) & 63;
      const double cj = (double)rdlane_f32(c, p);
      const double old_ = (double)c, nc_ = w0[u] + cj;
      const bool upd = nc_ < old_;
      c = upd ? (float)nc_ : c;
      l = upd ? (u32)(i * 16 + u) : l;
    }
  }
  u64 t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = c + (float)l; if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
// two positions per step: c[p] and the not-yet-final c[p+1] are read together; the literal edge
// p -> p+1 is evaluated on wave-uniform VGPR values (no second SGPR round trip)
__global__ void k_step_pair(float* out, u64* cyc, const double* w) {
  __shared__ double s_lit[64];
  s_lit[threadIdx.x] = w[threadIdx.x];
  __syncthreads();
  float c = out[threadIdx.x];
  u32 l = 0;
  double w0[16], wl[8];
#pragma unroll
  for (int u = 0; u < 16; ++u) w0[u] = w[(threadIdx.x + u) & 63] + u;
#pragma unroll
  for (int u = 0; u < 8; ++u) wl[u] = s_lit[u * 2];   // uniform address: the value in every lane
  u64 t0 = __builtin_readcyclecounter();
  for (int i = 0; i < N / 16; ++i) {
#pragma unroll
    for (int u = 0; u < 16; u += 2) {
      const u32 p = (i * 16 + u) & 63;
      const float sa = rdlane_f32(c, p), sb = rdlane_f32(c, (p + 1) & 63);
      const double cja = (double)sa;
      const double t = wl[u >> 1] + cja;
      const float cb = t < (double)sb ? (float)t : sb;      // c[p+1], final
      const double cjb = (double)cb;
      {
        const double old_ = (double)c, nc_ = w0[u] + cja;
        const bool upd = nc_ < old_;
        c = upd ? (float)nc_ : c;
        l = upd ? (u32)(i * 16 + u) : l;
      }
      {
        const double old_ = (double)c, nc_ = w0[u + 1] + cjb;
        const bool upd = nc_ < old_;
        c = upd ? (float)nc_ : c;
        l = upd ? (u32)(i * 16 + u + 1) : l;
      }
    }
  }
  u64 t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = c + (float)l; if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

// issue rate with only the first NACT lanes active (does the SIMD skip idle 16-lane passes?)
template <int NACT>
__global__ void k_int_tp_lanes(u32* out, u64* cyc, u32 x) {
  u32 a0 = out[threadIdx.x], a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  double d0 = a0, d1 = a1, d2 = a2, d3 = a3;
  u64 t0 = 0, t1 = 0;
  if (threadIdx.x < NACT) {
    t0 = __builtin_readcyclecounter();
#pragma unroll 4
    for (int i = 0; i < N / 8; ++i) { a0 = a0 * 3 + x; a1 = a1 * 3 + x; a2 = a2 * 3 + x; a3 = a3 * 3 + x; d0 += 1.5; d1 += 1.5; d2 += 1.5; d3 += 1.5; }
    t1 = __builtin_readcyclecounter();
  }
  out[threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (u32)(d0 + d1 + d2 + d3); if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

// k_step_d3 with the cost through v_min_f64 (no vcc on the cost chain); l from a side compare
__global__ void k_step_min(float* out, u64* cyc, const double* w) {
  float c = out[threadIdx.x];
  u32 l = 0;
  double w0[16];
#pragma unroll
  for (int u = 0; u < 16; ++u) w0[u] = w[(threadIdx.x + u) & 63] + u;
  double cd = (double)c;
  u64 t0 = __builtin_readcyclecounter();
  for (int i = 0; i < N / 16; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const u32 p = (i * 16 + u) & 63;
      const double cj = (double)rdlane_f32(c, p);
      const double nc_ = w0[u] + cj;
      const bool upd = nc_ < cd;
      c = (float)fmin(nc_, cd);
      cd = (double)c;
      l = upd ? (u32)(u + 1) : l;
    }
  }
  u64 t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = c + (float)l; if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
// the same with the constant-source trick of k_dp3 (reference point)
__global__ void k_step_d3k(float* out, u64* cyc, const double* w) {
  float c = out[threadIdx.x];
  u32 l = 0;
  double w0[16];
#pragma unroll
  for (int u = 0; u < 16; ++u) w0[u] = w[(threadIdx.x + u) & 63] + u;
  u64 t0 = __builtin_readcyclecounter();
  for (int i = 0; i < N / 16; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const u32 p = (i * 16 + u) & 63;
      const double cj = (double)rdlane_f32(c, p);
      const double old_ = (double)c, nc_ = w0[u] + cj;
      const bool upd = nc_ < old_;
      c = upd ? (float)nc_ : c;
      l = upd ? (u32)(u + 1) : l;
    }
  }
  u64 t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = c + (float)l; if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
  double* d; float* f; u32* u; u64* cyc; double* w;
  hipMalloc(&d, 64 * 8); hipMalloc(&f, 64 * 4); hipMalloc(&u, 64 * 4); hipMalloc(&cyc, 8); hipMalloc(&w, 64 * 8);
  std::vector<double> hd(64, 1.0), hw(64, 3.25);
  std::vector<float> hf(64, 1.0f);
  hipMemcpy(d, hd.data(), 512, hipMemcpyHostToDevice);
  hipMemcpy(w, hw.data(), 512, hipMemcpyHostToDevice);
  hipMemcpy(f, hf.data(), 256, hipMemcpyHostToDevice);
  u64 h;
#define RUN(name, ...)                                                                  \
  for (int rep = 0; rep < 2; ++rep) {                                                   \
    hipLaunchKernelGGL(name, dim3(1), dim3(64), 0, 0, __VA_ARGS__);                     \
    hipDeviceSynchronize();                                                             \
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);                                       \
    if (rep) std::printf("%-14s %8.2f cycles/iter\n", #name, (double)h / N);           \
  }
  RUN(k_add_f64, d, cyc, 1.5)
  RUN(k_add_f64_tp, d, cyc, 1.5)
  RUN(k_add_f32, f, cyc, 1.5f)
  RUN(k_readlane, f, cyc, 1.5f)
  RUN(k_cmp_sel, d, cyc, 1e300, 0.5)
  RUN(k_step_v1, f, cyc, w, 2.0)
  RUN(k_step_v2, f, cyc, w, 2.0)
  RUN(k_cvt_rt, d, cyc, 1.5)
  RUN(k_step_d3, f, cyc, w)
  RUN(k_step_pair, f, cyc, w)
  RUN(k_step_d3k, f, cyc, w)
  RUN(k_step_min, f, cyc, w)
  RUN(k_lit_chain, d, cyc, w, 2.0)
  RUN(k_lds_chase, u, cyc)
  RUN(k_int_tp, u, cyc, 5u)
  RUN(k_readlane_tp, u, cyc)
  RUN(k_int_tp_lanes<64>, u, cyc, 5u)
  RUN(k_int_tp_lanes<32>, u, cyc, 5u)
  RUN(k_int_tp_lanes<16>, u, cyc, 5u)
  RUN(k_fast_block<0>, f, cyc, 2.0)
  RUN(k_fast_block<1>, f, cyc, 2.0)
  RUN(k_fast_block<2>, f, cyc, 2.0)
  return 0;
}

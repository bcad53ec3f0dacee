// Micro-benchmark of dependent-instruction latencies for the serial DP chain (one wave, gfx950).
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_chain.hip -o tools/ubench_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32;
typedef unsigned long long u64;
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
  RUN(k_lit_chain, d, cyc, w, 2.0)
  RUN(k_lds_chase, u, cyc)
  return 0;
}

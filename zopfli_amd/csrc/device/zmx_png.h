// PNG scanline filter heuristics on the device: which of the five PNG filter types LodePNG's encoder
// picks for every scanline under its MINSUM and ENTROPY strategies (lodepng.cpp:5444-5570 `filter`, :5379-5421
// `filterScanline`), the per-row search zopflipng's strategy trials spend their filtering time in
// (zopflipng_lib.cc:160-268 TryOptimize, :270-305 AutoChooseFilterStrategy).  Included only by zmx_hip.hip.
//
// A scanline is `linebytes` bytes, a pixel `bytewidth` bytes (1 when the bit depth is below 8: the filters then work on
// the packed bytes).  For every row all five filtered versions are produced on the fly — a lane owns byte x of the
// row and needs x - bytewidth of it and x, x - bytewidth of the row above, all from the raw image — and scored:
//   MINSUM   sum of the bytes (type 0) or of their distance from zero as signed bytes (types 1 - 4); the first smallest
//   ENTROPY  sum over the 256 byte values (the filter type byte counted in) of LodePNG's integer i log2 i; the first largest
// One workgroup per row: integer sums only, any order gives LodePNG's value.  Output: the filter type per row for each
// strategy — what LodePNG takes back as LFS_PREDEFINED; the filtered bytes themselves it then writes in one pass.
#pragma once

#define PNGF_THREADS 256u

struct PngFilterParams {
  const u8* image;       // height rows of linebytes bytes
  u32 linebytes, height, bytewidth;
  u8* minsum;            // [height] or null
  u8* entropy;           // [height] or null
};

// lodepng.cpp:3974-3981 paethPredictor
__device__ __forceinline__ u32 pngf_paeth(int a, int b, int c) {
  int pa = b - c, pb = a - c, pc = pa + pb;
  pa = pa < 0 ? -pa : pa;
  pb = pb < 0 ? -pb : pb;
  pc = pc < 0 ? -pc : pc;
  if (pb < pa) { a = b; pa = pb; }
  return (u32)(pc < pa ? c : a);
}
// lodepng.cpp:5424-5442
__device__ __forceinline__ u32 pngf_ilog2i(u32 i) {
  if (i == 0) return 0;
  const u32 l = 31u - (u32)__clz((int)i);
  return i * l + ((i - (1u << l)) << 1);
}

__global__ __launch_bounds__(PNGF_THREADS) void k_png_filter_types(PngFilterParams P) {
  __shared__ u32 s_cnt[5][256];
  __shared__ u32 s_red[5][PNGF_THREADS / 64];
  const u32 y = blockIdx.x;
  const u32 tid = threadIdx.x;
  const u32 n = P.linebytes, bw = P.bytewidth;
  const u8* row = P.image + (u64)y * n;
  const u8* prev = y ? row - n : nullptr;
  const bool ent = P.entropy != nullptr;
  if (ent) {
    for (u32 i = tid; i < 5 * 256; i += PNGF_THREADS) (&s_cnt[0][0])[i] = 0;
    __syncthreads();
  }
  u32 sum[5] = {0, 0, 0, 0, 0};
  for (u32 x = tid; x < n; x += PNGF_THREADS) {
    const int s = row[x];
    const int a = x >= bw ? row[x - bw] : 0;            // left
    const int b = prev ? prev[x] : 0;                   // up
    const int c = (prev && x >= bw) ? prev[x - bw] : 0; // up left
    // lodepng.cpp:5379-5421: without a row above, Up is the bytes themselves, Average halves the left byte only,
    // Paeth is Sub (the predictor of (a, 0, 0) is a) — which the general forms give with b = c = 0
    u32 f[5];
    f[0] = (u32)s;
    f[1] = (u32)(s - a) & 255u;
    f[2] = (u32)(s - b) & 255u;
    f[3] = (u32)(s - ((a + b) >> 1)) & 255u;
    f[4] = (u32)(s - (int)pngf_paeth(a, b, c)) & 255u;
    sum[0] += f[0];
#pragma unroll
    for (int t = 1; t < 5; ++t) sum[t] += f[t] < 128u ? f[t] : 255u - f[t];
    if (ent) {
#pragma unroll
      for (int t = 0; t < 5; ++t) atomicAdd(&s_cnt[t][f[t]], 1u);
    }
  }
  // MINSUM: the five sums over the row
#pragma unroll
  for (int t = 0; t < 5; ++t) {
    u32 v = sum[t];
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if ((tid & 63u) == 0) s_red[t][tid >> 6] = v;
  }
  __syncthreads();
  if (tid == 0 && P.minsum) {
    u32 best = 0, smallest = 0;
    for (u32 t = 0; t < 5; ++t) {
      u32 v = 0;
      for (u32 w = 0; w < PNGF_THREADS / 64; ++w) v += s_red[t][w];
      if (t == 0 || v < smallest) { best = t; smallest = v; }     // lodepng.cpp:5521
    }
    P.minsum[y] = (u8)best;
  }
  if (!ent) return;
  __syncthreads();
  // ENTROPY: a thread per byte value; the filter type byte is part of the scanline (lodepng.cpp:5555)
  u32 e[5];
#pragma unroll
  for (int t = 0; t < 5; ++t) e[t] = pngf_ilog2i(s_cnt[t][tid] + (tid == (u32)t ? 1u : 0u));
#pragma unroll
  for (int t = 0; t < 5; ++t) {
    u32 v = e[t];
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if ((tid & 63u) == 0) s_red[t][tid >> 6] = v;
  }
  __syncthreads();
  if (tid == 0) {
    u32 best = 0, largest = 0;
    for (u32 t = 0; t < 5; ++t) {
      u32 v = 0;
      for (u32 w = 0; w < PNGF_THREADS / 64; ++w) v += s_red[t][w];
      if (t == 0 || v > largest) { best = t; largest = v; }       // lodepng.cpp:5560
    }
    P.entropy[y] = (u8)best;
  }
}

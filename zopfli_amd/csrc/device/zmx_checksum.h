// k_checksum: CRC-32 and Adler-32 pieces of the resident input (the host loops it
// replaces: gzip_container.c:75-82 and zlib_container.c:29-48).  The arithmetic that puts the pieces
// together is host/checksum.{h,cc}.
//
// One workgroup per piece of 256 KiB, pieces and lanes aligned to the END of the range (a short
// leftmost lane or piece counts as full: leading zeros change neither the CRC register started from 0
// nor the Adler sums).  A lane walks its 1 KiB with the four-table (slicing) CRC step, tables in LDS,
// and keeps the byte sum and the sum weighted by the distance to its end; a tree over the 256 lanes
// multiplies the left half by x^(8 * bytes of the right half) (32 shift-and-xor steps) and adds.
// HBM traffic: the input once, 12 B out per piece.
#ifndef ZMX_CHECKSUM_H_
#define ZMX_CHECKSUM_H_

#include "checksum.h"

struct ChecksumParams {
  const u8* in;
  long long begin, end;
  u32* out;       // [pieces][3]: crc0, sum, wsum (zamd::ChecksumPiece), piece 0 = rightmost
  u32 xpow[8];    // x^(8 * lane bytes * 2^k)
};

__device__ __forceinline__ u32 ck_mulmod(u32 a, u32 b) {
  u32 p = 0;
#pragma unroll 1
  for (u32 m = 0x80000000u; m; m >>= 1) {
    if (a & m) p ^= b;
    b = (b & 1) ? (b >> 1) ^ zamd::kCrcPoly : b >> 1;
  }
  return p;
}

__global__ __launch_bounds__(256) void k_checksum(ChecksumParams P) {
  constexpr long long LANE = zamd::kChecksumLaneBytes;
  constexpr u32 BASE = zamd::kAdlerBase;
  static_assert(zamd::kChecksumLanes == 256, "one lane per table entry");
  __shared__ u32 s_t[4][256];
  __shared__ u32 s_c[256], s_a[256], s_b[256];
  const u32 t = threadIdx.x;
  {
    u32 c = t;
    for (int k = 0; k < 8; ++k) c = (c & 1) ? zamd::kCrcPoly ^ (c >> 1) : c >> 1;
    s_t[0][t] = c;
    __syncthreads();
    for (int i = 1; i < 4; ++i) {
      c = s_t[0][c & 255] ^ (c >> 8);   // entry t followed by i zero bytes
      s_t[i][t] = c;
    }
    __syncthreads();
  }
  const long long wg_end = P.end - static_cast<long long>(blockIdx.x) * (LANE * 256);
  const long long hi = wg_end - static_cast<long long>(255 - t) * LANE;
  long long p = hi - LANE;
  if (p < P.begin) p = P.begin;
  u32 crc = 0, sum = 0, wsum = 0;
  const u8* in = P.in;
  for (; p < hi && (p & 3); ++p) {
    const u32 d = in[p];
    crc = s_t[0][(crc ^ d) & 255] ^ (crc >> 8);
    sum += d;
    wsum += static_cast<u32>(hi - p) * d;
  }
  for (; p + 4 <= hi; p += 4) {
    const u32 w = *reinterpret_cast<const u32*>(in + p);
    crc ^= w;
    crc = s_t[3][crc & 255] ^ s_t[2][(crc >> 8) & 255] ^ s_t[1][(crc >> 16) & 255] ^ s_t[0][crc >> 24];
    const u32 b0 = w & 255, b1 = (w >> 8) & 255, b2 = (w >> 16) & 255, b3 = w >> 24;
    const u32 s4 = b0 + b1 + b2 + b3;
    sum += s4;
    wsum += static_cast<u32>(hi - p) * s4 - (b1 + 2 * b2 + 3 * b3);   // < 2^28 per lane
  }
  for (; p < hi; ++p) {
    const u32 d = in[p];
    crc = s_t[0][(crc ^ d) & 255] ^ (crc >> 8);
    sum += d;
    wsum += static_cast<u32>(hi - p) * d;
  }
  s_c[t] = crc;
  s_a[t] = sum % BASE;
  s_b[t] = wsum % BASE;
  for (u32 s = 1, k = 0; s < 256; s <<= 1, ++k) {
    __syncthreads();
    if ((t & (2 * s - 1)) == 0) {
      const u32 right_bytes = (s * static_cast<u32>(LANE)) % BASE;
      const u32 la = s_a[t];
      s_c[t] = ck_mulmod(s_c[t], P.xpow[k]) ^ s_c[t + s];
      s_b[t] = static_cast<u32>((s_b[t] + static_cast<unsigned long long>(right_bytes) * la + s_b[t + s]) % BASE);
      s_a[t] = (la + s_a[t + s]) % BASE;
    }
  }
  if (t == 0) {
    P.out[3 * blockIdx.x + 0] = s_c[0];
    P.out[3 * blockIdx.x + 1] = s_a[0];
    P.out[3 * blockIdx.x + 2] = s_b[0];
  }
}

#endif  // ZMX_CHECKSUM_H_

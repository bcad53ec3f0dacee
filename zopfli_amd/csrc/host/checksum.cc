#include "checksum.h"

#include "zopfli_amd.h"

namespace zamd {

uint32_t Gf2MulMod(uint32_t a, uint32_t b) {
  uint32_t p = 0;
  for (uint32_t m = 0x80000000u; m; m >>= 1) {
    if (a & m) p ^= b;
    b = (b & 1) ? (b >> 1) ^ kCrcPoly : b >> 1;
  }
  return p;
}

uint32_t Gf2XPow8(uint64_t nbytes) {
  // x^(8n) by square and multiply over the bits of n; x^8 is bit 23
  uint32_t result = 0x80000000u, sq = 0x00800000u;
  for (; nbytes; nbytes >>= 1) {
    if (nbytes & 1) result = Gf2MulMod(result, sq);
    sq = Gf2MulMod(sq, sq);
  }
  return result;
}

void ChecksumTreePowers(uint32_t xpow[8]) {
  xpow[0] = Gf2XPow8(kChecksumLaneBytes);
  for (int k = 1; k < 8; ++k) xpow[k] = Gf2MulMod(xpow[k - 1], xpow[k - 1]);
}

uint32_t FinishCrc32(const ChecksumPiece* pieces, size_t npieces, uint64_t n) {
  const uint32_t xpiece = Gf2XPow8(kChecksumPieceBytes);
  uint32_t acc = 0;
  for (size_t i = npieces; i-- > 0;) acc = Gf2MulMod(acc, xpiece) ^ pieces[i].crc0;
  return acc ^ Gf2MulMod(0xffffffffu, Gf2XPow8(n)) ^ 0xffffffffu;
}

uint32_t FinishAdler32(const ChecksumPiece* pieces, size_t npieces, uint64_t n) {
  uint64_t sum = 0, wsum = 0;
  for (size_t i = npieces; i-- > 0;) {
    // the bytes so far move kChecksumPieceBytes further from the end
    wsum = (wsum + (kChecksumPieceBytes % kAdlerBase) * sum + pieces[i].wsum) % kAdlerBase;
    sum = (sum + pieces[i].sum) % kAdlerBase;
  }
  const uint32_t s1 = static_cast<uint32_t>((1 + sum) % kAdlerBase);
  const uint32_t s2 = static_cast<uint32_t>((n % kAdlerBase + wsum) % kAdlerBase);
  return (s2 << 16) | s1;
}

}  // namespace zamd

extern "C" uint32_t zmx_checksum_combine(int kind, uint32_t a, uint32_t b, uint64_t len_b) {
  using namespace zamd;
  if (kind == ZMX_CRC32) return Gf2MulMod(a, Gf2XPow8(len_b)) ^ b;
  // Adler-32 of A||B: s1 = s1a + s1b - 1, s2 = s2a + s2b + |B| (s1a - 1)
  const uint64_t s1a = a & 0xffff, s2a = a >> 16, s1b = b & 0xffff, s2b = b >> 16;
  const uint64_t s1 = (s1a + s1b + kAdlerBase - 1) % kAdlerBase;
  const uint64_t s2 = (s2a + s2b + (len_b % kAdlerBase) * ((s1a + kAdlerBase - 1) % kAdlerBase)) % kAdlerBase;
  return static_cast<uint32_t>((s2 << 16) | s1);
}

// CRC-32 / Adler-32 of the resident input, computed on the device in pieces (zmx_checksum.h) and put
// together here.  Replaces the byte-serial loops of gzip_container.c:75 (CRC) and zlib_container.c:29
// (adler32) — SURVEY 8 row f-2.
//
// CRC-32 is linear over GF(2) once the 0xffffffff start value and final inversion are taken out:
//   crc0(A||B) = crc0(A) * x^(8|B|) + crc0(B)   (mod the CRC polynomial, bit-reflected),
//   crc0(0..0||M) = crc0(M),
//   crc(M) = crc0(M) ^ 0xffffffff * x^(8|M|) ^ 0xffffffff.
// Adler-32 of M (n bytes): s1 = 1 + sum d_i, s2 = n + sum (n - i) d_i, both mod 65521: a piece needs
// its byte sum and its sum weighted by the distance to the piece's END.
// So the pieces are aligned to the END of the range: all have the same length but the leftmost, which
// is as good as full (leading zeros change neither sum).
#ifndef ZOPFLI_AMD_CHECKSUM_H_
#define ZOPFLI_AMD_CHECKSUM_H_

#include <cstddef>
#include <cstdint>

namespace zamd {

constexpr uint32_t kCrcPoly = 0xedb88320u;
constexpr uint32_t kAdlerBase = 65521u;
// the device kernel's geometry: one lane per kChecksumLaneBytes, 256 lanes per piece
constexpr uint32_t kChecksumLaneBytes = 1024;
constexpr uint32_t kChecksumLanes = 256;
constexpr uint32_t kChecksumPieceBytes = kChecksumLaneBytes * kChecksumLanes;

struct ChecksumPiece {
  uint32_t crc0;  // CRC register after the piece's bytes, started from 0, not inverted
  uint32_t sum;   // sum of the bytes mod 65521
  uint32_t wsum;  // sum of (piece end - position) * byte mod 65521
};

// a(x) * b(x) mod P, bit-reflected (x^0 is bit 31)
uint32_t Gf2MulMod(uint32_t a, uint32_t b);
// x^(8 * nbytes) mod P
uint32_t Gf2XPow8(uint64_t nbytes);
// xpow[k] = x^(8 * kChecksumLaneBytes * 2^k), k = 0..7: what the in-piece tree of the kernel needs
void ChecksumTreePowers(uint32_t xpow[8]);

// pieces[0] is the RIGHTMOST piece (the one that ends at the range's end); n = bytes in the range
uint32_t FinishCrc32(const ChecksumPiece* pieces, size_t npieces, uint64_t n);
uint32_t FinishAdler32(const ChecksumPiece* pieces, size_t npieces, uint64_t n);

}  // namespace zamd

#endif  // ZOPFLI_AMD_CHECKSUM_H_

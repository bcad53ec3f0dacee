// Deflate bit packing (LSB-first within bytes), replacing the bit-at-a-time
// AddBit/AddBits/AddHuffmanBits of the reference (deflate.c:38-72) with a
// 64-bit accumulator.  A BitWriter always starts at bit 0 of its own buffer;
// streams are joined later with AppendBits(), which re-aligns to the running
// bit pointer of the destination.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <vector>

#include "block_cache.h"

namespace zamd {

class BitWriter {
 public:
  void Reserve(size_t bytes) { buf_.reserve(bytes); }

  // `nbits` <= 32 low bits of `value`, least significant bit first.  (Fewer than 32 bits wait in the accumulator:
  // 32 more always fit; four bytes leave at a time.)
  void AddBits(uint32_t value, unsigned nbits) {
    acc_ |= static_cast<uint64_t>(value & ((nbits >= 32) ? 0xffffffffu : ((1u << nbits) - 1))) << fill_;
    fill_ += nbits;
    if (fill_ >= 32) {
      const uint32_t low = static_cast<uint32_t>(acc_);
      const size_t at = buf_.size();
      buf_.resize(at + 4);
      // the stream's byte order, whatever the host's (the compiler turns the four stores into one on x86)
      buf_[at] = static_cast<uint8_t>(low);
      buf_[at + 1] = static_cast<uint8_t>(low >> 8);
      buf_[at + 2] = static_cast<uint8_t>(low >> 16);
      buf_[at + 3] = static_cast<uint8_t>(low >> 24);
      acc_ >>= 32;
      fill_ -= 32;
    }
  }

  // Huffman codes are packed most significant bit first (RFC 1951 §3.1.1).
  void AddHuffmanBits(uint32_t code, unsigned len) { AddBits(Reverse(code, len), len); }

  size_t BitCount() const { return buf_.size() * 8 + fill_; }

  // Returns the bytes (last one zero-padded) and leaves the writer empty.
  CVec<uint8_t> Finish(size_t* nbits) {
    *nbits = BitCount();
    for (unsigned done = 0; done < fill_; done += 8) buf_.push_back(static_cast<uint8_t>(acc_ >> done));
    acc_ = 0;
    fill_ = 0;
    return std::move(buf_);
  }

 private:
  static uint32_t Reverse(uint32_t v, unsigned len) {
    uint32_t r = 0;
    for (unsigned i = 0; i < len; ++i) r |= ((v >> i) & 1u) << (len - 1 - i);
    return r;
  }
  CVec<uint8_t> buf_;
  uint64_t acc_ = 0;
  unsigned fill_ = 0;
};

// Destination stream with the reference's (bytes, bp) convention: `bp` in 0..7
// is the number of bits already used in the last byte (deflate.h:50-53).
struct BitStream {
  std::vector<uint8_t> bytes;
  unsigned bp = 0;

  void AppendBits(const uint8_t* src, size_t nbits) {
    if (nbits == 0) return;
    const size_t nbytes = (nbits + 7) / 8;
    if (bp == 0) {
      bytes.insert(bytes.end(), src, src + nbytes);
    } else {
      const unsigned sh = bp;
      const size_t old = bytes.size();
      // bits still free in the current last byte: 8 - sh
      const size_t total_bits = (old - 1) * 8 + sh + nbits;
      bytes.resize((total_bits + 7) / 8, 0);
      uint8_t* dst = bytes.data() + old - 1;
      uint8_t* const end = bytes.data() + bytes.size();
      uint64_t carry = *dst;   // the sh bits already used in the last byte
      size_t i = 0;
      // eight source bytes per step: out = carry | v << sh, next carry = the sh bits shifted out
      for (; i + 8 <= nbytes && dst + 8 <= end; i += 8) {
        uint64_t v;
        std::memcpy(&v, src + i, 8);
        const uint64_t out = carry | (v << sh);
        std::memcpy(dst, &out, 8);
        dst += 8;
        carry = v >> (64 - sh);
      }
      for (; i < nbytes; ++i) {
        const unsigned v = src[i];
        *dst++ = static_cast<uint8_t>(carry | (v << sh));
        carry = v >> (8 - sh);
      }
      if (dst < end) *dst = static_cast<uint8_t>(carry);
    }
    bp = static_cast<unsigned>((bp + nbits) & 7);
  }
  void AppendBit(unsigned bit) {
    uint8_t b = static_cast<uint8_t>(bit & 1);
    AppendBits(&b, 1);
  }
  void AppendByteAligned(uint8_t v) { bytes.push_back(v); }
};

}  // namespace zamd

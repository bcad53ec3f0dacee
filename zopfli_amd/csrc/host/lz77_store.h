// Host-side LZ77 symbol sequence (role of ZopfliLZ77Store, lz77.h:44-62).
//
// The device returns flat (litlen, dist) arrays; the host only needs range
// histograms (for the block-cost model) and byte positions.  Instead of the
// reference's chunk-wrapped cumulative arrays (lz77.c:98-149) we keep prefix
// histograms sampled every kSample symbols, which gives identical counts for
// any [lstart, lend).
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <vector>

#include "block_cache.h"
#include "symbols.h"

namespace zamd {

struct Histogram {
  size_t ll[kNumLL];
  size_t d[kNumD];
  void Clear() { std::memset(this, 0, sizeof(*this)); }
};

class Lz77Store {
 public:
  // (a sample is 2.5 KB: every 256 symbols that was 10 bytes written per symbol, more than the symbols themselves — a store of
  //  random data, a million literals per master block, took as long to build as to search; at 1024 a range histogram adds
  //  up to 2 x 1023 symbols instead of 2 x 255, ~ 1 us of a 20 - 40 us evaluation)
  static constexpr size_t kSample = 1024;

  explicit Lz77Store(const unsigned char* data = nullptr) : data_(data) {}

  size_t size() const { return litlens_.size(); }
  const unsigned char* data() const { return data_; }
  uint16_t litlen(size_t i) const { return litlens_[i]; }
  uint16_t dist(size_t i) const { return dists_[i]; }
  size_t pos(size_t i) const { return pos_[i]; }
  // bytes covered by symbol i
  size_t span(size_t i) const { return dists_[i] == 0 ? 1 : litlens_[i]; }

  void Reserve(size_t n) {
    litlens_.reserve(n);
    dists_.reserve(n);
    pos_.reserve(n);
  }

  // ZopfliStoreLitLenDist (lz77.c:98)
  void Push(uint16_t litlen, uint16_t dist, size_t pos) {
    const size_t i = litlens_.size();
    if (i % kSample == 0) {
      if (samples_.empty()) {
        samples_.emplace_back();
        samples_.back().Clear();
      } else {
        samples_.push_back(running_);
      }
    }
    litlens_.push_back(litlen);
    dists_.push_back(dist);
    pos_.push_back(static_cast<uint32_t>(pos));
    if (dist == 0) {
      running_.ll[litlen]++;
    } else {
      running_.ll[LengthSymbol(litlen)]++;
      running_.d[DistSymbol(dist)]++;
    }
  }

  // Append `n` symbols starting at byte position `pos` (positions are implied
  // by the symbol lengths).  ZopfliAppendLZ77Store (lz77.c:151).
  void Append(const uint16_t* litlens, const uint16_t* dists, size_t n, size_t pos) {
    // (in one piece: a master block is a quarter of a million symbols, and three push_backs and a look at the sample
    //  grid per symbol were most of what joining the blocks' stores cost)
    const size_t base = size();
    litlens_.resize(base + n);
    dists_.resize(base + n);
    pos_.resize(base + n);
    if (n) {
      std::memcpy(litlens_.data() + base, litlens, n * sizeof(uint16_t));
      std::memcpy(dists_.data() + base, dists, n * sizeof(uint16_t));
    }
    samples_.reserve((base + n) / kSample + 1);
    for (size_t i = 0; i < n; ++i) {
      if ((base + i) % kSample == 0) samples_.push_back(running_);    // counts of the symbols before this one
      pos_[base + i] = static_cast<uint32_t>(pos);
      const unsigned litlen = litlens[i], dist = dists[i];
      if (dist == 0) {
        running_.ll[litlen]++;
        pos += 1;
      } else {
        running_.ll[LengthSymbol(litlen)]++;
        running_.d[DistSymbol(dist)]++;
        pos += litlen;
      }
    }
  }
  void Append(const Lz77Store& other) {
    Reserve(size() + other.size());
    for (size_t i = 0; i < other.size(); ++i) Push(other.litlens_[i], other.dists_[i], other.pos_[i]);
  }

  // ZopfliLZ77GetByteRange (lz77.c:160)
  size_t ByteRange(size_t lstart, size_t lend) const {
    if (lstart == lend) return 0;
    const size_t l = lend - 1;
    return pos_[l] + span(l) - pos_[lstart];
  }

  // Histogram of symbols [lstart, lend) without the end symbol.
  // ZopfliLZ77GetHistogram (lz77.c:189).
  void GetHistogram(size_t lstart, size_t lend, Histogram* h) const {
    if (lend - lstart < 2 * kSample) {
      h->Clear();
      Accumulate(lstart, lend, h, +1);
      return;
    }
    PrefixAt(lend, h);
    if (lstart > 0) {
      Histogram lo;
      PrefixAt(lstart, &lo);
      for (int i = 0; i < kNumLL; ++i) h->ll[i] -= lo.ll[i];
      for (int i = 0; i < kNumD; ++i) h->d[i] -= lo.d[i];
    }
  }

 private:
  void Accumulate(size_t a, size_t b, Histogram* h, int) const {
    for (size_t i = a; i < b; ++i) {
      if (dists_[i] == 0) {
        h->ll[litlens_[i]]++;
      } else {
        h->ll[LengthSymbol(litlens_[i])]++;
        h->d[DistSymbol(dists_[i])]++;
      }
    }
  }
  // counts of symbols [0, l)
  void PrefixAt(size_t l, Histogram* h) const {
    const size_t s = l / kSample;
    if (s < samples_.size()) {
      *h = samples_[s];
      Accumulate(s * kSample, l, h, +1);
    } else {  // l == size() and size() % kSample == 0
      *h = running_;
    }
  }

  const unsigned char* data_;
  // (a master block's store is ~ 5 MB of these: from the library's block cache, not malloc — block_cache.h)
  CVec<uint16_t> litlens_, dists_;
  CVec<uint32_t> pos_;           // (byte positions in the resident input: 32 bits, as the device layer's)
  CVec<Histogram> samples_;  // samples_[k] = counts of symbols [0, k*kSample)
  Histogram running_ = {};          // counts of all symbols pushed so far
};

}  // namespace zamd

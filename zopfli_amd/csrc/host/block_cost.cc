#include "block_cost.h"

#include <cstdlib>
#include <cstring>
#include <vector>

#include "huffman.h"

namespace zamd {

namespace {

// RFC 1951 §3.2.7 order of the code-length code lengths.
const unsigned kClOrder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

struct ClToken {
  uint8_t symbol;  // 0..18
  uint8_t extra;   // repeat count minus base for 16/17/18
};

// Run-length codes the concatenated litlen+dist code lengths with the given
// subset of the repeat codes and returns the size of the dynamic block header
// in bits.  Writes the header when `out` is given.  Behaviour of EncodeTree
// (deflate.c:105-249).
size_t EncodeCodeLengths(const unsigned* ll_lengths, const unsigned* d_lengths, bool use16, bool use17,
                         bool use18, BitWriter* out) {
  unsigned hlit = 29, hdist = 29;  // gzip rejects hdist > 29 (deflate.c:115)
  while (hlit > 0 && ll_lengths[257 + hlit - 1] == 0) hlit--;
  while (hdist > 0 && d_lengths[1 + hdist - 1] == 0) hdist--;
  const unsigned nll = hlit + 257;
  const unsigned total = nll + hdist + 1;

  uint8_t seq[kNumLL + kNumD];
  for (unsigned i = 0; i < total; ++i) seq[i] = static_cast<uint8_t>(i < nll ? ll_lengths[i] : d_lengths[i - nll]);

  size_t clcount[19] = {0};
  ClToken tokens[kNumLL + kNumD];
  unsigned ntok = 0;
  auto emit = [&](unsigned symbol, unsigned extra) {
    clcount[symbol]++;
    tokens[ntok++] = {static_cast<uint8_t>(symbol), static_cast<uint8_t>(extra)};
  };

  for (unsigned i = 0; i < total;) {
    const unsigned symbol = seq[i];
    unsigned run = 1;
    if (use16 || (symbol == 0 && (use17 || use18))) {
      while (i + run < total && seq[i + run] == symbol) run++;
    }
    i += run;

    if (symbol == 0 && run >= 3) {
      if (use18) {
        while (run >= 11) {
          const unsigned take = run > 138 ? 138 : run;
          emit(18, take - 11);
          run -= take;
        }
      }
      if (use17) {
        while (run >= 3) {
          const unsigned take = run > 10 ? 10 : run;
          emit(17, take - 3);
          run -= take;
        }
      }
    }
    if (use16 && run >= 4) {
      emit(symbol, 0);  // code 16 repeats the previous length, so send one first
      run--;
      while (run >= 3) {
        const unsigned take = run > 6 ? 6 : run;
        emit(16, take - 3);
        run -= take;
      }
    }
    for (; run > 0; --run) emit(symbol, 0);
  }

  unsigned clcl[19];
  LengthLimitedCodeLengths(clcount, 19, 7, clcl);

  unsigned hclen = 15;
  while (hclen > 0 && clcount[kClOrder[hclen + 4 - 1]] == 0) hclen--;

  if (out) {
    unsigned clcode[19];
    LengthsToSymbols(clcl, 19, 7, clcode);
    out->AddBits(hlit, 5);
    out->AddBits(hdist, 5);
    out->AddBits(hclen, 4);
    for (unsigned i = 0; i < hclen + 4; ++i) out->AddBits(clcl[kClOrder[i]], 3);
    for (unsigned t = 0; t < ntok; ++t) {
      const unsigned s = tokens[t].symbol;
      out->AddHuffmanBits(clcode[s], clcl[s]);
      if (s == 16) out->AddBits(tokens[t].extra, 2);
      else if (s == 17) out->AddBits(tokens[t].extra, 3);
      else if (s == 18) out->AddBits(tokens[t].extra, 7);
    }
  }

  size_t bits = 14 + (hclen + 4) * 3;
  for (int i = 0; i < 19; ++i) bits += clcl[i] * clcount[i];
  bits += clcount[16] * 2 + clcount[17] * 3 + clcount[18] * 7;
  return bits;
}

// Size in bits of the header EncodeCodeLengths would write, for all eight subsets of the repeat codes at
// once: the runs of the length sequence are found once, every subset only counts its tokens (the same
// rules as above), and subsets that come to the same token counts share one code-length computation.
// Index (bit0 = use16, bit1 = use17, bit2 = use18) of the smallest header; the first minimum wins as in
// AddDynamicTree (deflate.c:251-272).
int BestCodeLengthEncoding(const unsigned* ll_lengths, const unsigned* d_lengths, size_t* best_bits) {
  unsigned hlit = 29, hdist = 29;
  while (hlit > 0 && ll_lengths[257 + hlit - 1] == 0) hlit--;
  while (hdist > 0 && d_lengths[1 + hdist - 1] == 0) hdist--;
  const unsigned nll = hlit + 257;
  const unsigned total = nll + hdist + 1;
  uint8_t run_symbol[kNumLL + kNumD];
  uint16_t run_length[kNumLL + kNumD];
  unsigned nruns = 0;
  for (unsigned i = 0; i < total;) {
    const unsigned symbol = i < nll ? ll_lengths[i] : d_lengths[i - nll];
    unsigned run = 1;
    while (i + run < total && (i + run < nll ? ll_lengths[i + run] : d_lengths[i + run - nll]) == symbol) run++;
    run_symbol[nruns] = static_cast<uint8_t>(symbol);
    run_length[nruns++] = static_cast<uint16_t>(run);
    i += run;
  }
  size_t counts[8][19];
  size_t sizes[8];
  int best = 0;
  size_t best_size = 0;
  for (int v = 0; v < 8; ++v) {
    const bool use16 = v & 1, use17 = v & 2, use18 = v & 4;
    size_t* clcount = counts[v];
    for (int i = 0; i < 19; ++i) clcount[i] = 0;
    for (unsigned r = 0; r < nruns; ++r) {
      const unsigned symbol = run_symbol[r];
      unsigned run = run_length[r];
      if (!(use16 || (symbol == 0 && (use17 || use18)))) {   // no run is looked for: one token per length
        clcount[symbol] += run;
        continue;
      }
      if (symbol == 0 && run >= 3) {
        if (use18) {
          while (run >= 11) {
            run -= run > 138 ? 138 : run;
            clcount[18]++;
          }
        }
        if (use17) {
          while (run >= 3) {
            run -= run > 10 ? 10 : run;
            clcount[17]++;
          }
        }
      }
      if (use16 && run >= 4) {
        clcount[symbol]++;
        run--;
        while (run >= 3) {
          run -= run > 6 ? 6 : run;
          clcount[16]++;
        }
      }
      clcount[symbol] += run;
    }
    int same = -1;
    for (int w = 0; w < v && same < 0; ++w) {
      if (std::memcmp(counts[w], clcount, sizeof(counts[w])) == 0) same = w;
    }
    if (same >= 0) {
      sizes[v] = sizes[same];
    } else {
      unsigned clcl[19];
      LengthLimitedCodeLengths(clcount, 19, 7, clcl);
      unsigned hclen = 15;
      while (hclen > 0 && clcount[kClOrder[hclen + 4 - 1]] == 0) hclen--;
      size_t bits = 14 + (hclen + 4) * 3;
      for (int i = 0; i < 19; ++i) bits += clcl[i] * clcount[i];
      bits += clcount[16] * 2 + clcount[17] * 3 + clcount[18] * 7;
      sizes[v] = bits;
    }
    if (best_size == 0 || sizes[v] < best_size) {
      best_size = sizes[v];
      best = v;
    }
  }
  *best_bits = best_size;
  return best;
}

// At least two distance codes, for decoders that choke otherwise
// (PatchDistanceCodesForBuggyDecoders, deflate.c:86).
void EnsureTwoDistanceCodes(unsigned* d_lengths) {
  int used = 0;
  for (int i = 0; i < 30; ++i) {
    if (d_lengths[i]) used++;
    if (used >= 2) return;
  }
  if (used == 0) {
    d_lengths[0] = d_lengths[1] = 1;
  } else {
    d_lengths[d_lengths[0] ? 1 : 0] = 1;
  }
}

void FixedTree(unsigned* ll_lengths, unsigned* d_lengths) {
  for (int i = 0; i < kNumLL; ++i) ll_lengths[i] = i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8;
  for (int i = 0; i < kNumD; ++i) d_lengths[i] = 5;
}

// Bits of all symbols + extra bits + the end symbol, from a histogram
// (CalculateBlockSymbolSizeGivenCounts, deflate.c:383; the per-symbol variant
// for short ranges, deflate.c:354, sums the same terms).
size_t SymbolBits(const Histogram& h, const unsigned* ll_lengths, const unsigned* d_lengths) {
  size_t bits = 0;
  for (int i = 0; i < 256; ++i) bits += ll_lengths[i] * h.ll[i];
  for (int i = 257; i < 286; ++i) bits += (ll_lengths[i] + LengthSymbolExtraBits(i)) * h.ll[i];
  for (int i = 0; i < 30; ++i) bits += (d_lengths[i] + DistSymbolExtraBits(i)) * h.d[i];
  return bits + ll_lengths[256];
}

size_t AbsDiff(size_t a, size_t b) { return a > b ? a - b : b - a; }

}  // namespace

size_t TreeSize(const unsigned* ll_lengths, const unsigned* d_lengths) {
  size_t bits;
  BestCodeLengthEncoding(ll_lengths, d_lengths, &bits);
  return bits;
}

void OptimizeCountsForRle(int length, size_t* counts) {
  // trailing zeros stay untouched
  while (length > 0 && counts[length - 1] == 0) --length;
  if (length == 0) return;

  // Runs that already code well with the repeat codes are frozen: >=5 zeros or
  // >=7 equal non-zero counts.
  char frozen[kNumLL] = {0};
  for (int start = 0; start < length;) {
    int stop = start + 1;
    while (stop < length && counts[stop] == counts[start]) ++stop;
    const int run = stop - start;
    if ((counts[start] == 0 && run >= 5) || (counts[start] != 0 && run >= 7)) {
      for (int k = start; k < stop; ++k) frozen[k] = 1;
    }
    start = stop;
  }

  // Collapse stretches of similar counts to their rounded mean.
  int stride = 0;
  size_t limit = counts[0];
  size_t sum = 0;
  for (int i = 0; i <= length; ++i) {
    if (i == length || frozen[i] || AbsDiff(counts[i], limit) >= 4) {
      if (stride >= 4 || (stride >= 3 && sum == 0)) {
        int mean = static_cast<int>((sum + stride / 2) / stride);
        if (mean < 1) mean = 1;
        if (sum == 0) mean = 0;  // never promote an all-zero stretch
        for (int k = 0; k < stride; ++k) counts[i - k - 1] = mean;
      }
      stride = 0;
      sum = 0;
      if (i < length - 3) {
        limit = (counts[i] + counts[i + 1] + counts[i + 2] + counts[i + 3] + 2) / 4;
      } else if (i < length) {
        limit = counts[i];
      } else {
        limit = 0;
      }
    }
    ++stride;
    if (i != length) sum += counts[i];
  }
}

double DynamicLengths(const Histogram& hin, unsigned* ll_lengths, unsigned* d_lengths) {
  Histogram h = hin;
  h.ll[256] = 1;  // end symbol
  LengthLimitedCodeLengths(h.ll, kNumLL, 15, ll_lengths);
  LengthLimitedCodeLengths(h.d, kNumD, 15, d_lengths);
  EnsureTwoDistanceCodes(d_lengths);

  // Try RLE-friendlier counts and keep whichever gives the smaller
  // header + data (TryOptimizeHuffmanForRle, deflate.c:525).
  const double tree = static_cast<double>(TreeSize(ll_lengths, d_lengths));
  const double data = static_cast<double>(SymbolBits(h, ll_lengths, d_lengths));

  Histogram smooth = h;
  OptimizeCountsForRle(kNumLL, smooth.ll);
  OptimizeCountsForRle(kNumD, smooth.d);
  // (nothing smoothed — incompressible data: 256 literal counts of a few hundred each, no stretch within 4 of its mean —:
  //  the same counts give the same lengths and the same sizes; the second pair of package-merges was 45 % of a block-size
  //  evaluation there, and the split search of random data is 2.8 s of CPU per 100 MB)
  if (std::memcmp(&smooth, &h, sizeof(h)) == 0) return tree + data;
  unsigned ll2[kNumLL], d2[kNumD];
  LengthLimitedCodeLengths(smooth.ll, kNumLL, 15, ll2);
  LengthLimitedCodeLengths(smooth.d, kNumD, 15, d2);
  EnsureTwoDistanceCodes(d2);
  const bool same_lengths = std::memcmp(ll2, ll_lengths, sizeof(ll2)) == 0 && std::memcmp(d2, d_lengths, sizeof(d2)) == 0;
  const double tree2 = same_lengths ? tree : static_cast<double>(TreeSize(ll2, d2));
  const double data2 = static_cast<double>(SymbolBits(h, ll2, d2));

  if (tree2 + data2 < tree + data) {
    std::memcpy(ll_lengths, ll2, sizeof(ll2));
    std::memcpy(d_lengths, d2, sizeof(d2));
    return tree2 + data2;
  }
  return tree + data;
}

double BlockSizeFromHistogram(const Histogram& h, int btype) {
  unsigned ll_lengths[kNumLL], d_lengths[kNumD];
  if (btype == 1) {
    FixedTree(ll_lengths, d_lengths);
    return 3 + static_cast<double>(SymbolBits(h, ll_lengths, d_lengths));
  }
  return 3 + DynamicLengths(h, ll_lengths, d_lengths);
}

double CalculateBlockSize(const Lz77Store& lz77, size_t lstart, size_t lend, int btype) {
  if (btype == 0) {
    const size_t length = lz77.ByteRange(lstart, lend);
    const size_t rem = length % 65535;
    const size_t blocks = length / 65535 + (rem ? 1 : 0);
    // 5 bytes of header per stored block (deflate.c:591-597)
    return static_cast<double>(blocks * 5 * 8 + length * 8);
  }
  Histogram h;
  lz77.GetHistogram(lstart, lend, &h);
  return BlockSizeFromHistogram(h, btype);
}

double CalculateBlockSizeAutoType(const Lz77Store& lz77, size_t lstart, size_t lend) {
  return CalculateBlockSizeAutoTypeOf(lz77, lstart, lend, lz77.size());
}

double CalculateBlockSizeAutoTypeOf(const Lz77Store& lz77, size_t lstart, size_t lend, size_t store_size) {
  const double stored = CalculateBlockSize(lz77, lstart, lend, 0);
  Histogram h;
  lz77.GetHistogram(lstart, lend, &h);
  // fixed-tree size is only evaluated for small stores (deflate.c:615)
  const double fixed = store_size > 1000 ? stored : BlockSizeFromHistogram(h, 1);
  const double dynamic = BlockSizeFromHistogram(h, 2);
  return (stored < fixed && stored < dynamic) ? stored : (fixed < dynamic ? fixed : dynamic);
}

void EncodeBlock(const Lz77Store& lz77, size_t lstart, size_t lend, int btype, bool final_block,
                 BitWriter* out, size_t* tree_bits) {
  unsigned ll_lengths[kNumLL], d_lengths[kNumD];
  unsigned ll_codes[kNumLL], d_codes[kNumD];
  out->AddBits(final_block ? 1 : 0, 1);
  out->AddBits(static_cast<uint32_t>(btype), 2);
  if (btype == 1) {
    FixedTree(ll_lengths, d_lengths);
  } else {
    Histogram h;
    lz77.GetHistogram(lstart, lend, &h);
    DynamicLengths(h, ll_lengths, d_lengths);
    size_t unused;
    const int enc = BestCodeLengthEncoding(ll_lengths, d_lengths, &unused);
    const size_t before = out->BitCount();
    EncodeCodeLengths(ll_lengths, d_lengths, enc & 1, enc & 2, enc & 4, out);
    if (tree_bits) *tree_bits = out->BitCount() - before;
  }
  LengthsToSymbols(ll_lengths, kNumLL, 15, ll_codes);
  LengthsToSymbols(d_lengths, kNumD, 15, d_codes);

  // The codes bit-reversed once (RFC 1951 3.1.1: Huffman codes go most significant bit first), a symbol's code and
  // its extra bits in one piece (15 + 5 and 15 + 13 bits): this loop is what a block costs whose symbols the device
  // cannot write (deflate.cc), 16 ns a symbol with a reversal and up to four pieces per symbol.
  auto rev = [](unsigned v, unsigned len) {
    unsigned r = 0;
    for (unsigned i = 0; i < len; ++i) r |= ((v >> i) & 1u) << (len - 1 - i);
    return r;
  };
  for (int k = 0; k < kNumLL; ++k) ll_codes[k] = rev(ll_codes[k], ll_lengths[k]);
  for (int k = 0; k < kNumD; ++k) d_codes[k] = rev(d_codes[k], d_lengths[k]);
  out->Reserve((out->BitCount() + 7) / 8 + (lend - lstart) * 2 + 16);
  for (size_t i = lstart; i < lend; ++i) {
    const unsigned litlen = lz77.litlen(i), dist = lz77.dist(i);
    if (dist == 0) {
      out->AddBits(ll_codes[litlen], ll_lengths[litlen]);
    } else {
      const int ls = LengthSymbol(litlen), ds = DistSymbol(dist);
      out->AddBits(ll_codes[ls] | (static_cast<uint32_t>(LengthExtraValue(litlen)) << ll_lengths[ls]),
                   ll_lengths[ls] + static_cast<unsigned>(LengthExtraBits(litlen)));
      out->AddBits(d_codes[ds] | (static_cast<uint32_t>(DistExtraValue(dist)) << d_lengths[ds]),
                   d_lengths[ds] + static_cast<unsigned>(DistExtraBits(dist)));
    }
  }
  out->AddBits(ll_codes[256], ll_lengths[256]);
}

size_t EncodeBlockHeader(const Histogram& h, int btype, bool final_block, BitWriter* out, size_t* tree_bits,
                         uint32_t* codes320) {
  unsigned ll_lengths[kNumLL], d_lengths[kNumD];
  unsigned ll_codes[kNumLL], d_codes[kNumD];
  out->AddBits(final_block ? 1 : 0, 1);
  out->AddBits(static_cast<uint32_t>(btype), 2);
  if (btype == 1) {
    FixedTree(ll_lengths, d_lengths);
  } else {
    DynamicLengths(h, ll_lengths, d_lengths);
    size_t unused;
    const int enc = BestCodeLengthEncoding(ll_lengths, d_lengths, &unused);
    const size_t before = out->BitCount();
    EncodeCodeLengths(ll_lengths, d_lengths, enc & 1, enc & 2, enc & 4, out);
    if (tree_bits) *tree_bits = out->BitCount() - before;
  }
  LengthsToSymbols(ll_lengths, kNumLL, 15, ll_codes);
  LengthsToSymbols(d_lengths, kNumD, 15, d_codes);
  auto rev = [](unsigned v, unsigned len) {
    unsigned r = 0;
    for (unsigned i = 0; i < len; ++i) r |= ((v >> i) & 1u) << (len - 1 - i);
    return r;
  };
  size_t bits = ll_lengths[256];
  for (int s = 0; s < kNumLL; ++s) {
    codes320[s] = rev(ll_codes[s], ll_lengths[s]) | (ll_lengths[s] << 16);
    if (s == 256 || s > 285) continue;
    bits += h.ll[s] * (ll_lengths[s] + (s > 256 ? static_cast<unsigned>(LengthSymbolExtraBits(s)) : 0u));
  }
  for (int s = 0; s < kNumD; ++s) {
    codes320[kNumLL + s] = rev(d_codes[s], d_lengths[s]) | (d_lengths[s] << 16);
    if (s < 30) bits += h.d[s] * (d_lengths[s] + static_cast<unsigned>(DistSymbolExtraBits(s)));
  }
  return bits;
}

}  // namespace zamd

// RFC 1951 §3.2.5 length / distance code geometry.
//
// Replaces the lookup functions of the reference's symbols.h
// (ZopfliGetLengthSymbol symbols.h:183, ZopfliGetLengthExtraBits :138,
// ZopfliGetLengthExtraBitsValue :161, ZopfliGetDistSymbol :88,
// ZopfliGetDistExtraBits :38, ZopfliGetDistExtraBitsValue :61,
// ZopfliGetLengthSymbolExtraBits :222, ZopfliGetDistSymbolExtraBits :231).
// The tables are generated from the RFC's base/extra lists instead of being
// spelled out; values are identical for every valid length (3..258) and
// distance (1..32768).
#pragma once
#include <array>
#include <cstdint>

namespace zamd {

constexpr int kNumLL = 288;        // literal/length alphabet
constexpr int kNumD = 32;          // distance alphabet
constexpr int kMinMatch = 3;
constexpr int kMaxMatch = 258;
constexpr int kWindow = 32768;
constexpr size_t kMasterBlock = 1000000;  // reference util.h:60

struct LengthCodes {
  std::array<uint16_t, 259> symbol{};   // 257..285
  std::array<uint8_t, 259> ebits{};     // number of extra bits
  std::array<uint8_t, 259> evalue{};    // value of the extra bits
  std::array<uint8_t, 29> sym_ebits{};  // extra bits per length symbol (index sym-257)
};

constexpr LengthCodes MakeLengthCodes() {
  LengthCodes t{};
  // 8 codes per "octave" of 4 after the first 8 single-length codes.
  int base = 3;
  for (int s = 0; s < 28; ++s) {
    int eb = s < 8 ? 0 : (s - 4) / 4;
    t.sym_ebits[s] = static_cast<uint8_t>(eb);
    int span = 1 << eb;
    for (int v = 0; v < span && base + v <= 257; ++v) {
      t.symbol[base + v] = static_cast<uint16_t>(257 + s);
      t.ebits[base + v] = static_cast<uint8_t>(eb);
      t.evalue[base + v] = static_cast<uint8_t>(v);
    }
    base += span;
  }
  // 258 has its own zero-extra-bit code; 227..257 stay on code 284.
  t.symbol[258] = 285;
  t.ebits[258] = 0;
  t.evalue[258] = 0;
  t.sym_ebits[28] = 0;
  return t;
}

inline constexpr LengthCodes kLen = MakeLengthCodes();

inline int LengthSymbol(int l) { return kLen.symbol[l]; }
inline int LengthExtraBits(int l) { return kLen.ebits[l]; }
inline int LengthExtraValue(int l) { return kLen.evalue[l]; }
inline int LengthSymbolExtraBits(int sym) { return kLen.sym_ebits[sym - 257]; }

inline int FloorLog2(unsigned v) { return 31 - __builtin_clz(v); }

inline int DistSymbol(int d) {
  if (d < 5) return d - 1;
  int l = FloorLog2(static_cast<unsigned>(d - 1));
  return 2 * l + (((d - 1) >> (l - 1)) & 1);
}
inline int DistExtraBits(int d) {
  return d < 5 ? 0 : FloorLog2(static_cast<unsigned>(d - 1)) - 1;
}
inline int DistExtraValue(int d) {
  if (d < 5) return 0;
  int l = FloorLog2(static_cast<unsigned>(d - 1));
  return (d - (1 + (1 << l))) & ((1 << (l - 1)) - 1);
}
inline int DistSymbolExtraBits(int sym) { return sym < 4 ? 0 : (sym - 2) / 2; }

}  // namespace zamd

// Huffman helpers used by the host-side block-cost model and encoder.
#pragma once
#include <cstddef>
#include <cstdint>

namespace zamd {

// Length-limited prefix code lengths (boundary package-merge).  Same results,
// including tie-breaks, as ZopfliLengthLimitedCodeLengths (katajainen.c:172).
// Returns false when maxbits cannot represent the used symbols.
bool LengthLimitedCodeLengths(const size_t* freq, int n, int maxbits, unsigned* lengths);

// Canonical code assignment, RFC 1951 §3.2.2 (reference: tree.c:30).
void LengthsToSymbols(const unsigned* lengths, size_t n, unsigned maxbits, unsigned* symbols);

// Entropy-based bit costs, reference tree.c:71 (ZopfliCalculateEntropy).
// Stays on the host so that `log` is the same libm the reference links.
void CalculateEntropy(const size_t* count, size_t n, double* bitlengths);

}  // namespace zamd

// Choice of deflate block boundaries on an LZ77 symbol sequence
// (ZopfliBlockSplitLZ77, blocksplitter.c:215; FindMinimum :43;
// FindLargestSplittableBlock :195).  Host side: cheap, order-sensitive search.
#pragma once
#include <cstddef>
#include <vector>

#include "lz77_store.h"

namespace zamd {

// Appends up to maxblocks-1 split points (symbol indices, ascending) to `points`.
void BlockSplitLz77(const Lz77Store& lz77, size_t maxblocks, std::vector<size_t>* points);

// The same for several sequences at once, round by round on the worker pool, the blocks that exist searched before
// their turn: for requests of a few master blocks, where one thread per sequence leaves the search latency-bound
// (block_split.cc).  Same points as BlockSplitLz77 for every sequence.
void BlockSplitLz77Batch(const std::vector<const Lz77Store*>& stores, size_t maxblocks,
                         std::vector<std::vector<size_t>>* points);

// Converts symbol-index split points to byte positions, counting from
// `instart` (tail of ZopfliBlockSplit, blocksplitter.c:303-314).
std::vector<size_t> SplitPointsToBytes(const Lz77Store& lz77, const std::vector<size_t>& points,
                                       size_t instart);

}  // namespace zamd

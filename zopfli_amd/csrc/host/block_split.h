// Choice of deflate block boundaries on an LZ77 symbol sequence
// (ZopfliBlockSplitLZ77, blocksplitter.c:215; FindMinimum :43;
// FindLargestSplittableBlock :195).  Host side: cheap, order-sensitive search.
#pragma once
#include <cstddef>
#include <cstdint>
#include <functional>
#include <vector>

#include "lz77_store.h"

namespace zamd {

// Appends up to maxblocks-1 split points (symbol indices, ascending) to `points`.
void BlockSplitLz77(const Lz77Store& lz77, size_t maxblocks, std::vector<size_t>* points);

// The same for several sequences at once, round by round on the worker pool, the blocks that exist searched before
// their turn: for requests of a few master blocks, where one thread per sequence leaves the search latency-bound
// (block_split.cc).  Same points as BlockSplitLz77 for every sequence.
//
// `device` (optional): the block sizes of a round by the device (zmx_block_costs, zmx_blockcost.h: a wave per block size) —
// cost[i] = ZopfliCalculateBlockSizeAutoType(sequence q[i].store, q[i].lstart, q[i].lend), the same integers as the host's;
// it returns false where it cannot serve (the host evaluates that round).  Rounds of fewer than `device_min` block sizes
// stay on the host's pool: a launch and a round trip cost what a few dozen evaluations cost there.
struct CostQuery { uint32_t store, lstart, lend; };
using CostBatchFn = std::function<bool(const CostQuery* q, size_t n, double* cost)>;
void BlockSplitLz77Batch(const std::vector<const Lz77Store*>& stores, size_t maxblocks,
                         std::vector<std::vector<size_t>>* points, const CostBatchFn* device = nullptr,
                         size_t device_min = 0);

// The same with NO host stores: sequences of sizes[i] symbols that live on the device only, every round's block sizes by
// `device` (the greedy stores of a call's master blocks: building their host stores — byte positions, sampled
// histograms — took as long as the search).  false: the device did not serve a round; `points` is unusable then.
bool BlockSplitSizesBatch(const std::vector<size_t>& sizes, size_t maxblocks, std::vector<std::vector<size_t>>* points,
                          const CostBatchFn& device);

// Converts symbol-index split points to byte positions, counting from
// `instart` (tail of ZopfliBlockSplit, blocksplitter.c:303-314).
std::vector<size_t> SplitPointsToBytes(const Lz77Store& lz77, const std::vector<size_t>& points,
                                       size_t instart);

}  // namespace zamd

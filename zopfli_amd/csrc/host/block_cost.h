// Exact deflate block size model and block encoder (host side).
//
// Mirrors the behaviour of the reference's deflate.c cost functions
// (ZopfliCalculateBlockSize :584, ZopfliCalculateBlockSizeAutoType :610,
// GetDynamicLengths :569, TryOptimizeHuffmanForRle :525, OptimizeHuffmanForRle
// :434, CalculateTreeSize :277, EncodeTree :105, AddLZ77Block :682) on top of
// plain histograms, because the device returns one 320-bin histogram per block
// and iteration rather than a symbol store.
#pragma once
#include <cstddef>

#include "bit_writer.h"
#include "lz77_store.h"

namespace zamd {

// `h` must not contain the end symbol; it is added here (deflate.c:576).
// Returns tree + data bits (no 3-bit header) and the chosen code lengths.
double DynamicLengths(const Histogram& h, unsigned* ll_lengths, unsigned* d_lengths);

// Block size in bits for btype 1 or 2 given the histogram of the block.
double BlockSizeFromHistogram(const Histogram& h, int btype);

// ZopfliCalculateBlockSize (deflate.c:584) for symbols [lstart, lend) of a store.
double CalculateBlockSize(const Lz77Store& lz77, size_t lstart, size_t lend, int btype);

// ZopfliCalculateBlockSizeAutoType (deflate.c:610).
double CalculateBlockSizeAutoType(const Lz77Store& lz77, size_t lstart, size_t lend);
// the same for symbols [lstart, lend) of a store that holds more than the store the reference would pass:
// `store_size` = the size of that store (deflate.c:615 looks at it)
double CalculateBlockSizeAutoTypeOf(const Lz77Store& lz77, size_t lstart, size_t lend, size_t store_size);

// OptimizeHuffmanForRle (deflate.c:434); exposed for unit tests.
void OptimizeCountsForRle(int length, size_t* counts);

// Smallest encoding of the two code-length sequences (deflate.c:277).
size_t TreeSize(const unsigned* ll_lengths, const unsigned* d_lengths);

// Emits one compressed block (btype 1 or 2) including its 3 header bits and
// the end symbol (AddLZ77Block, deflate.c:682).
// *tree_bits (optional): bits of the dynamic tree header (what the reference's -v prints as "treesize").
void EncodeBlock(const Lz77Store& lz77, size_t lstart, size_t lend, int btype, bool final_block,
                 BitWriter* out, size_t* tree_bits = nullptr);

// The part of EncodeBlock that does not touch the symbols: the 3 header bits and (btype 2) the tree of a block with
// histogram `h` (without the end symbol, as above), plus what a symbol writer needs — codes[s] = bit-reversed
// Huffman code | length << 16 for the 288 litlen symbols and, from 288 on, the 32 distance symbols — and the number
// of bits the symbols and the end symbol will take (AddLZ77Data, deflate.c:297-333).  The device writes those
// (zmx_encode_blocks) behind the header.
size_t EncodeBlockHeader(const Histogram& h, int btype, bool final_block, BitWriter* out, size_t* tree_bits,
                         uint32_t* codes320);

}  // namespace zamd

// Host-side iteration control of the optimal parse, batched over blocks.
//
// The device runs the O(bytes x iterations) work (match tables, greedy parse,
// squeeze DP, traceback, follow-path, histograms); everything here is the
// cheap, order-sensitive control flow the reference keeps per block in
// ZopfliLZ77Optimal (squeeze.c:446-526): statistics, libm entropy, exact block
// cost, best-so-far tracking and the RNG perturbation.  All blocks of a batch
// advance in lock-step, one device launch per iteration.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "block_cache.h"
#include "zopfli_amd.h"

namespace zamd {

struct SymbolRun {
  CVec<uint16_t> litlens, dists;         // lz77.h:44-49 convention (memory from the library's block cache)
  std::string log;                       // ZopfliOptions::verbose: the block's "Iteration i: n bit" lines (squeeze.c:493)
};

struct Timing {
  double tables = 0, greedy = 0, squeeze = 0, cost_model = 0, split = 0, encode = 0;
  double download = 0, serialize = 0;   // best stores device -> host; chunks -> blob
};
Timing& ThreadTiming();

// ZopfliLZ77Greedy (lz77.c:544) for each block.  With `keep`, the device tables of the blocks are
// handed to the caller (to be passed to Lz77OptimalBatch or freed with zmx_tables_free).
// With `nsym` and `defer_download`, the symbols stay on the device (only their number comes back): Lz77GreedyDownload
// fetches them later if somebody needs them on the host after all.
int Lz77GreedyBatch(zmx_ctx* ctx, const std::vector<zmx_block>& blocks, std::vector<SymbolRun>* out,
                    zmx_tables** keep = nullptr, std::vector<uint32_t>* nsym = nullptr, bool defer_download = false);
int Lz77GreedyDownload(zmx_ctx* ctx, zmx_tables* tables, const std::vector<uint32_t>& nsym, std::vector<SymbolRun>* out);

// ZopfliLZ77Optimal (squeeze.c:446) for each block: best of `numiterations`
// cost-model iterations seeded by a greedy parse.  `parent` (optional, consumed): tables of blocks
// that contain these ones — their match records are reused (zmx_tables_build_from).
// With `keep`, the table set stays alive (the caller frees it with zmx_tables_free) together with where each block's
// best parse is: what zmx_encode_blocks needs to write the blocks' bits on the device.
struct OptimalKeep {
  bool skip_download = false;      // in: the caller does not need the symbols on the host (`out` only gets the logs)
  zmx_tables* tables = nullptr;
  std::vector<int32_t> slot;
  std::vector<uint32_t> nsym;
  std::vector<uint32_t> hist;      // [blocks][ZMX_HIST] histogram of each block's best parse (no end symbol)
};
int Lz77OptimalBatch(zmx_ctx* ctx, const ZopfliOptions& options, const std::vector<zmx_block>& blocks,
                     std::vector<SymbolRun>* out, zmx_tables* parent = nullptr, OptimalKeep* keep = nullptr);

// ZopfliLZ77OptimalFixed (squeeze.c:528): one DP run with the fixed-tree costs.
// With `keep` as for Lz77OptimalBatch (the parse is in slot 0 of every block).
int Lz77OptimalFixedBatch(zmx_ctx* ctx, const std::vector<zmx_block>& blocks,
                          std::vector<SymbolRun>* out, OptimalKeep* keep = nullptr);

}  // namespace zamd

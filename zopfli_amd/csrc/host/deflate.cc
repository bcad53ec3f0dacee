#include "deflate.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "block_cost.h"
#include "block_split.h"
#include "lz77_optimal.h"
#include "lz77_store.h"
#include "thread_pool.h"

namespace zamd {

namespace {
// Few enough parts that one thread per part would leave the block-split search latency-bound (block_split.cc:
// BlockSplitLz77Batch).  ZOPFLI_AMD_BATCH_SPLIT = 0 / 1 forces the choice.
bool BatchSplit(size_t np) {
  static const int forced = [] { const char* e = std::getenv("ZOPFLI_AMD_BATCH_SPLIT"); return e ? std::atoi(e) : -1; }();
  if (forced >= 0) return forced != 0;
  return np * 2 <= WideThreads();
}
// f-1 on the device (zmx_blockcost.h): the block sizes of the split search's rounds by zmx_block_costs.
// ZOPFLI_AMD_DEVICE_SPLIT: 0 = the host evaluates everything, 2 = the device wherever there are `from` sequences, 1 (default)
// = where that pays.  What the host's search costs grows with the symbols in use (the package-merge) and with the
// sequence's length (rounds, and block sizes a round): on literal-heavy data — a symbol for nearly every byte: random or
// encrypted bytes, 1 M symbols and 257 codes in use a master block — it is 14 000 block sizes of 35 us per shard of a 100 MB
// call, 38 ms on the box's 16 CPUs beside the other shards' work, against 12 ms of device rounds (18 x 0.2 ms plus the
// queueing behind the other contexts' kernels); on text (4 000 block sizes of ~10 us, seven rounds) the host's 4 ms beat
// the device's 5 - 24 (profiles/r06_device_split.txt).  Hence: at least `from` sequences AND at least 0.45 symbols a byte.
bool DeviceSplit(size_t nseq, size_t from_default, size_t symbols, size_t bytes) {
  static const int on = [] { const char* e = std::getenv("ZOPFLI_AMD_DEVICE_SPLIT"); return e ? std::atoi(e) : 1; }();
  static const long from_env = [] { const char* e = std::getenv("ZOPFLI_AMD_DEVICE_SPLIT_FROM"); return e ? std::atol(e) : -1L; }();
  const size_t from = from_env >= 0 ? static_cast<size_t>(from_env) : from_default;
  if (on == 0 || nseq < from || nseq == 0) return false;
  return on >= 2 || 20 * symbols >= 9 * bytes;
}
size_t DeviceSplitMin() {
  static const size_t v = [] { const char* e = std::getenv("ZOPFLI_AMD_DEVICE_SPLIT_MIN"); return e ? static_cast<size_t>(std::atoll(e)) : static_cast<size_t>(128); }();
  return v;
}
struct CostStoresGuard {
  zmx_ctx* ctx;
  zmx_cost_stores* cs = nullptr;
  ~CostStoresGuard() { if (cs) zmx_cost_stores_free(ctx, cs); }
};
// the evaluator BlockSplitLz77Batch calls (false = this round on the host: the two give the same integers)
CostBatchFn DeviceCosts(zmx_ctx* ctx, zmx_cost_stores* cs) {
  return [ctx, cs](const CostQuery* q, size_t n, double* cost) {
    static_assert(sizeof(CostQuery) == 3 * sizeof(uint32_t), "zmx_block_costs takes the queries as triples");
    return zmx_block_costs(ctx, cs, n, reinterpret_cast<const uint32_t*>(q), cost) == 0;
  };
}
}  // namespace

namespace {

double Now() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

struct FinalBlock {
  size_t lstart = 0, lend = 0;       // symbol range in the part's store
  double stored = 0, fixed = 0, dynamic = 0;
  bool expensive_fixed = false;      // re-parse with the fixed-tree cost model
  long fixed_request = -1;           // index into the batched re-parse
};

// A compressed block whose symbols the device writes (zmx_encode_blocks): it is a whole block of the optimal batch.
struct DeviceEncode {
  bool used = false;                 // (a slot per final block while the blocks are encoded side by side; the unused go)
  size_t chunk = 0;                  // index in the part's chunks
  size_t block = 0;                  // index in the batched block list (or in the fixed-tree re-parse batch)
  bool from_fixed = false;           // its symbols are the fixed-tree re-parse's
  CVec<uint8_t> header;              // the 3 header bits and the tree, from bit 0
  size_t header_bits = 0;
  size_t data_bits = 0;              // symbols + end symbol
  uint32_t codes[320];
};

struct PartState {
  Part part{};
  std::vector<zmx_block> blocks;     // first-pass deflate blocks (byte ranges)
  std::vector<size_t> block_sym_end; // symbols of the part's store up to the end of each first-pass block
  std::vector<DeviceEncode> enc;
  size_t first_block = 0;            // offset into the batched block list
  Lz77Store lz77;                    // optimal parse of the whole part
  std::vector<size_t> splitpoints;   // final split points (symbol indices)
  std::vector<FinalBlock> finals;
  std::vector<Chunk> chunks;
  std::string log;                   // verbose: what the reference prints for the part before its blocks are written
};

// "block split points: ..." as PrintBlockSplitPoints does (blocksplitter.c:148-180): uncompressed
// offsets from the start of the store, decimal then hex.
std::string SplitPointsLine(const Lz77Store& lz77, const std::vector<size_t>& points) {
  std::string dec = "block split points: ", hex = "(hex:";
  const size_t origin = lz77.size() ? lz77.pos(0) : 0;
  char buf[32];
  for (size_t p : points) {
    const int v = static_cast<int>(lz77.pos(p) - origin);
    std::snprintf(buf, sizeof(buf), "%d ", v);
    dec += buf;
    std::snprintf(buf, sizeof(buf), " %x", v);
    hex += buf;
  }
  return dec + hex + ")\n";
}

Lz77Store StoreFromRun(const SymbolRun& run, size_t pos) {
  Lz77Store s;
  s.Append(run.litlens.data(), run.dists.data(), run.litlens.size(), pos);
  return s;
}

// ZOPFLI_AMD_TRACE_CALL=1 (api.cc prints the call's shards): the phases of DeflateParts on stderr
bool TraceCallEnv() {
  static const bool on = [] { const char* e = std::getenv("ZOPFLI_AMD_TRACE_CALL"); return e && std::atoi(e) != 0; }();
  return on;
}

Chunk BitsChunk(BitWriter* w) {
  Chunk c;
  c.kind = Chunk::kBits;
  c.bits = w->Finish(&c.nbits);
  return c;
}

}  // namespace

int DeflateParts(zmx_ctx* ctx, const ZopfliOptions& options, int btype, const std::vector<Part>& parts,
                 std::vector<Chunk>* chunks) {
  const size_t np = parts.size();
  std::vector<PartState> st(np);
  for (size_t p = 0; p < np; ++p) st[p].part = parts[p];
  int rc = 0;

  if (btype == 0) {  // deflate.c:827: stored blocks only
    for (size_t p = 0; p < np; ++p) {
      Chunk c;
      c.kind = Chunk::kStored;
      c.start = parts[p].instart;
      c.end = parts[p].inend;
      c.final_block = parts[p].final_part;
      chunks->push_back(std::move(c));
    }
    return 0;
  }

  if (btype == 1) {  // deflate.c:830-842: one fixed-tree block per part
    std::vector<zmx_block> blocks(np);
    for (size_t p = 0; p < np; ++p) blocks[p] = {parts[p].instart, parts[p].inend};
    std::vector<SymbolRun> runs;
    rc = Lz77OptimalFixedBatch(ctx, blocks, &runs);
    if (rc) return rc;
    ParallelFor(np, [&](size_t p) {
      Lz77Store s = StoreFromRun(runs[p], parts[p].instart);
      BitWriter w;
      EncodeBlock(s, 0, s.size(), 1, parts[p].final_part, &w);
      Chunk c = BitsChunk(&w);
      c.log_block = true;
      c.log_btype = 1;
      c.log_unc = parts[p].inend - parts[p].instart;
      st[p].chunks.push_back(std::move(c));
    });
    for (size_t p = 0; p < np; ++p) {
      for (auto& c : st[p].chunks) chunks->push_back(std::move(c));
    }
    return 0;
  }

  // ---- 1. first block split on a greedy parse of each part (deflate.c:845-850,
  //         blocksplitter.c:275)
  std::vector<std::vector<size_t>> split_bytes(np);
  zmx_tables* split_tables = nullptr;   // the master blocks' match tables, reused for their deflate blocks
  if (options.blocksplitting) {
    std::vector<zmx_block> ranges(np);
    for (size_t p = 0; p < np; ++p) ranges[p] = {parts[p].instart, parts[p].inend};
    std::vector<SymbolRun> greedy;
    std::vector<uint32_t> greedy_nsym;
    const double tg0 = Now();
    // (the symbols stay on the device until somebody wants them on the host: with the search's block sizes computed there
    //  nobody does — a shard of incompressible data is 130 MB of them)
    rc = Lz77GreedyBatch(ctx, ranges, &greedy, &split_tables, &greedy_nsym, DeviceSplit(np, 6, 1, 1));
    if (rc) return rc;
    const double t0 = Now();
    std::atomic<uint64_t> ns_store{0}, ns_search{0};
    bool dev_done = false;
    bool have_symbols = !DeviceSplit(np, 6, 1, 1);
    size_t greedy_symbols = 0, part_bytes = 0;
    for (size_t p = 0; p < np; ++p) { greedy_symbols += greedy_nsym[p]; part_bytes += parts[p].inend - parts[p].instart; }
    if (DeviceSplit(np, 6, greedy_symbols, part_bytes)) {
      // the greedy stores are on the device already (slot 0 of the master blocks' tables): every round of the search's
      // block sizes there (zmx_block_costs), no host store at all — and the split points' byte positions from there too
      const double a = Now();
      CostStoresGuard dev{ctx};
      std::vector<size_t> first(np + 1), blk(np), nsym(np);
      std::vector<int32_t> slot(np, 0);
      for (size_t p = 0; p < np; ++p) { first[p] = p; blk[p] = p; nsym[p] = greedy_nsym[p]; }
      first[np] = np;
      if (zmx_cost_stores_create(ctx, split_tables, np, first.data(), blk.data(), slot.data(), nsym.data(), &dev.cs) == 0) {
        const double b = Now();
        std::vector<std::vector<size_t>> pts;
        dev_done = BlockSplitSizesBatch(nsym, static_cast<size_t>(options.blocksplittingmax), &pts, DeviceCosts(ctx, dev.cs));
        const double c = Now();
        if (dev_done) {
          // SplitPointsToBytes without a store: the bytes the symbols before each point stand for (zmx_cost_positions)
          std::vector<uint32_t> pairs;
          for (size_t p = 0; p < np; ++p) for (size_t pt : pts[p]) { pairs.push_back(static_cast<uint32_t>(p)); pairs.push_back(static_cast<uint32_t>(pt)); }
          std::vector<uint64_t> bytes(pairs.size() / 2);
          dev_done = zmx_cost_positions(ctx, dev.cs, bytes.size(), pairs.data(), bytes.data()) == 0;
          size_t k = 0;
          for (size_t p = 0; p < np && dev_done; ++p) {
            split_bytes[p].clear();
            for (size_t i = 0; i < pts[p].size(); ++i) split_bytes[p].push_back(parts[p].instart + bytes[k++]);
            if (options.verbose) {    // blocksplitter.c:266-268, PrintBlockSplitPoints :148-180
              std::string dec = "block split points: ", hex = "(hex:";
              char buf[32];
              for (size_t bp : split_bytes[p]) {
                const int v = static_cast<int>(bp - parts[p].instart);
                std::snprintf(buf, sizeof(buf), "%d ", v);
                dec += buf;
                std::snprintf(buf, sizeof(buf), " %x", v);
                hex += buf;
              }
              st[p].log += dec + hex + ")\n";
            }
          }
        }
        if (TraceCallEnv()) std::fprintf(stderr, "    DeflateParts(%zu parts): first split on the device: sequences %.2f ms, rounds %.2f ms, points %.2f ms\n", np, (b - a) * 1e3, (c - b) * 1e3, (Now() - c) * 1e3);
      }
      if (!dev_done && TraceCallEnv()) std::fprintf(stderr, "    DeflateParts: no device block sizes (%s): the host evaluates\n", zmx_last_error());
    }
    if (!dev_done && !have_symbols) {
      rc = Lz77GreedyDownload(ctx, split_tables, greedy_nsym, &greedy);
      if (rc) return rc;
    }
    if (dev_done) {
    } else if (BatchSplit(np)) {
      // a few parts: all their searches advance together, round by round, on the whole pool (block_split.cc)
      std::vector<Lz77Store> stores(np);
      ParallelForWide(np, [&](size_t p) { stores[p] = StoreFromRun(greedy[p], parts[p].instart); });
      std::vector<const Lz77Store*> ptrs(np);
      for (size_t p = 0; p < np; ++p) ptrs[p] = &stores[p];
      std::vector<std::vector<size_t>> pts;
      BlockSplitLz77Batch(ptrs, static_cast<size_t>(options.blocksplittingmax), &pts);
      for (size_t p = 0; p < np; ++p) {
        if (options.verbose) st[p].log += SplitPointsLine(stores[p], pts[p]);   // blocksplitter.c:266-268
        split_bytes[p] = SplitPointsToBytes(stores[p], pts[p], parts[p].instart);
      }
    } else {
      ParallelForWide(np, [&](size_t p) {
        const double a = Now();
        Lz77Store s = StoreFromRun(greedy[p], parts[p].instart);
        const double b = Now();
        std::vector<size_t> pts;
        BlockSplitLz77(s, static_cast<size_t>(options.blocksplittingmax), &pts);
        if (options.verbose) st[p].log += SplitPointsLine(s, pts);   // blocksplitter.c:266-268
        split_bytes[p] = SplitPointsToBytes(s, pts, parts[p].instart);
        ns_store.fetch_add(static_cast<uint64_t>((b - a) * 1e9), std::memory_order_relaxed);
        ns_search.fetch_add(static_cast<uint64_t>((Now() - b) * 1e9), std::memory_order_relaxed);
      });
    }
    ThreadTiming().split += Now() - t0;
    if (TraceCallEnv()) {
      std::fprintf(stderr, "    DeflateParts(%zu parts): greedy batch %.2f ms, first split %.2f ms (per part: store %.2f, search %.2f)\n", np,
                   (t0 - tg0) * 1e3, (Now() - t0) * 1e3, ns_store.load() / 1e6 / np, ns_search.load() / 1e6 / np);
    }
  }

  // ---- 2. optimal parse of every block of every part, one batch (deflate.c:854-869)
  std::vector<zmx_block> all_blocks;
  for (size_t p = 0; p < np; ++p) {
    st[p].first_block = all_blocks.size();
    const auto& sp = split_bytes[p];
    for (size_t i = 0; i <= sp.size(); ++i) {
      const size_t s = i == 0 ? parts[p].instart : sp[i - 1];
      const size_t e = i == sp.size() ? parts[p].inend : sp[i];
      st[p].blocks.push_back({s, e});
      all_blocks.push_back({s, e});
    }
  }
  static const bool trace_phases = std::getenv("ZOPFLI_AMD_PROF") != nullptr || TraceCallEnv();
  const double tp0 = Now();
  std::vector<SymbolRun> runs;
  // (ZOPFLI_AMD_DEVICE_ENCODE=0: every block's bits on the host, as in round 1)
  static const bool device_encode = [] { const char* e = std::getenv("ZOPFLI_AMD_DEVICE_ENCODE"); return !e || std::atoi(e) != 0; }();
  // Without block splitting a part is one block and nothing below needs its symbols on the host: sizes and trees
  // come from the histogram of the best parse, the bits from the device.
  const bool no_symbols = device_encode && !options.blocksplitting && all_blocks.size() == np;
  OptimalKeep keep;
  keep.skip_download = no_symbols;
  rc = Lz77OptimalBatch(ctx, options, all_blocks, &runs, split_tables, device_encode ? &keep : nullptr);
  if (rc) return rc;
  struct TablesGuard {
    zmx_ctx* ctx; zmx_tables* t;
    ~TablesGuard() { if (t) zmx_tables_free(ctx, t); }
  } tables_guard{ctx, keep.tables};
  const double tp1 = Now();

  // ---- 3. join the blocks, second split attempt, per-block type costs
  std::vector<zmx_block> fixed_requests;
  std::vector<std::pair<size_t, size_t>> fixed_owner;  // (part, final index)
  const double t3 = Now();
  auto block_hist = [&](size_t block) {
    Histogram h;
    const uint32_t* c = &keep.hist[block * ZMX_HIST];
    for (int k = 0; k < kNumLL; ++k) h.ll[k] = c[k];
    for (int k = 0; k < kNumD; ++k) h.d[k] = c[kNumLL + k];
    return h;
  };
  if (no_symbols) {
    ParallelForWide(np, [&](size_t p) {   // one block per part; AddLZ77BlockAutoType (deflate.c:747-762) on its histogram
      PartState& s = st[p];
      s.log += runs[s.first_block].log;
      const size_t nsym = keep.nsym[s.first_block];
      const size_t length = s.blocks[0].inend - s.blocks[0].instart;
      const Histogram h = block_hist(s.first_block);
      FinalBlock f;
      f.lstart = 0;
      f.lend = nsym;
      f.stored = static_cast<double>((length / 65535 + (length % 65535 ? 1 : 0)) * 5 * 8 + length * 8);   // deflate.c:591-597
      f.fixed = BlockSizeFromHistogram(h, 1);
      f.dynamic = BlockSizeFromHistogram(h, 2);
      f.expensive_fixed = nsym < 1000 || f.fixed <= f.dynamic * 1.1;
      s.block_sym_end.push_back(nsym);
      s.finals.push_back(f);
    });
  } else {
  std::vector<double> totalcost(np, 0.0);
  ParallelForWide(np, [&](size_t p) {
    PartState& s = st[p];
    const size_t npoints = s.blocks.size() - 1;
    {
      size_t total = 0;      // (one allocation for the part's store: appending block by block re-allocated it every time)
      for (size_t i = 0; i <= npoints; ++i) total += runs[s.first_block + i].litlens.size();
      s.lz77.Reserve(total);
    }
    for (size_t i = 0; i <= npoints; ++i) {
      const SymbolRun& run = runs[s.first_block + i];
      s.log += run.log;       // "Iteration i: n bit" (squeeze.c:493), block after block
      // (the block's symbols straight into the part's store; the reference prices the block's own store,
      //  deflate.c:866: its size is what the fixed-tree rule of :615 looks at)
      const size_t lstart = s.lz77.size();
      s.lz77.Append(run.litlens.data(), run.dists.data(), run.litlens.size(), s.blocks[i].instart);
      totalcost[p] += CalculateBlockSizeAutoTypeOf(s.lz77, lstart, s.lz77.size(), run.litlens.size());
      s.block_sym_end.push_back(s.lz77.size());
      if (i < npoints) s.splitpoints.push_back(s.lz77.size());
    }
  });
  const double tj1 = Now();
  // deflate.c:872-893: the second split attempt, on the optimal parse
  std::vector<std::vector<size_t>> pts2(np);
  std::vector<char> tried(np, 0);
  for (size_t p = 0; p < np; ++p) tried[p] = options.blocksplitting && st[p].blocks.size() - 1 > 1;
  size_t ntried = 0, tried_symbols = 0, tried_bytes = 0;
  for (size_t p = 0; p < np; ++p) {
    if (!tried[p]) continue;
    ++ntried;
    tried_symbols += st[p].lz77.size();
    tried_bytes += parts[p].inend - parts[p].instart;
  }
  // (the second try has fewer sequences — only the parts that were split — and as many rounds: from 24 sequences on)
  const bool dev_split2 = device_encode && keep.tables != nullptr && DeviceSplit(ntried, 24, tried_symbols, tried_bytes);
  if (BatchSplit(np) || dev_split2) {
    std::vector<const Lz77Store*> ptrs;
    std::vector<size_t> owner;
    for (size_t p = 0; p < np; ++p) if (tried[p]) { ptrs.push_back(&st[p].lz77); owner.push_back(p); }
    std::vector<std::vector<size_t>> got;
    CostStoresGuard dev{ctx};
    CostBatchFn fn;
    if (dev_split2 && !ptrs.empty()) {
      // a part's optimal parse = the best stores of its blocks, one after the other, where they lie on the device
      std::vector<size_t> first(owner.size() + 1, 0), blk, nsym;
      std::vector<int32_t> slot;
      for (size_t i = 0; i < owner.size(); ++i) {
        const PartState& s = st[owner[i]];
        for (size_t k = 0; k < s.blocks.size(); ++k) {
          blk.push_back(s.first_block + k);
          slot.push_back(keep.slot[s.first_block + k]);
          nsym.push_back(keep.nsym[s.first_block + k]);
        }
        first[i + 1] = blk.size();
      }
      if (zmx_cost_stores_create(ctx, keep.tables, owner.size(), first.data(), blk.data(), slot.data(), nsym.data(), &dev.cs) == 0) fn = DeviceCosts(ctx, dev.cs);
      else if (TraceCallEnv()) std::fprintf(stderr, "    DeflateParts: no device block sizes (%s): the host evaluates\n", zmx_last_error());
    }
    BlockSplitLz77Batch(ptrs, static_cast<size_t>(options.blocksplittingmax), &got, fn ? &fn : nullptr, DeviceSplitMin());
    for (size_t i = 0; i < owner.size(); ++i) pts2[owner[i]].swap(got[i]);
  } else {
    ParallelForWide(np, [&](size_t p) {
      if (tried[p]) BlockSplitLz77(st[p].lz77, static_cast<size_t>(options.blocksplittingmax), &pts2[p]);
    });
  }
  const double tj2 = Now();
  ParallelForWide(np, [&](size_t p) {
    PartState& s = st[p];
    if (tried[p]) {
      if (options.verbose) s.log += SplitPointsLine(s.lz77, pts2[p]);
      double totalcost2 = 0;
      for (size_t i = 0; i <= pts2[p].size(); ++i) {
        const size_t a = i == 0 ? 0 : pts2[p][i - 1];
        const size_t b = i == pts2[p].size() ? s.lz77.size() : pts2[p][i];
        totalcost2 += CalculateBlockSizeAutoType(s.lz77, a, b);
      }
      if (totalcost2 < totalcost[p]) s.splitpoints.swap(pts2[p]);
    }
    for (size_t i = 0; i <= s.splitpoints.size(); ++i) {  // AddLZ77BlockAutoType, deflate.c:747-762
      FinalBlock f;
      f.lstart = i == 0 ? 0 : s.splitpoints[i - 1];
      f.lend = i == s.splitpoints.size() ? s.lz77.size() : s.splitpoints[i];
      f.stored = CalculateBlockSize(s.lz77, f.lstart, f.lend, 0);
      f.fixed = CalculateBlockSize(s.lz77, f.lstart, f.lend, 1);
      f.dynamic = CalculateBlockSize(s.lz77, f.lstart, f.lend, 2);
      f.expensive_fixed = (s.lz77.size() < 1000) || f.fixed <= f.dynamic * 1.1;
      s.finals.push_back(f);
    }
  });
  if (TraceCallEnv()) {
    std::fprintf(stderr, "    DeflateParts(%zu parts): optimal batch %.2f ms; stores joined %.2f ms, second split %.2f ms, block costs %.2f ms\n", np,
                 (tp1 - tp0) * 1e3, (tj1 - t3) * 1e3, (tj2 - tj1) * 1e3, (Now() - tj2) * 1e3);
  }
  }
  ThreadTiming().split += Now() - t3;

  for (size_t p = 0; p < np; ++p) {
    for (size_t i = 0; i < st[p].finals.size(); ++i) {
      FinalBlock& f = st[p].finals[i];
      if (f.lstart == f.lend || !f.expensive_fixed) continue;
      const size_t instart = no_symbols ? st[p].blocks[0].instart : st[p].lz77.pos(f.lstart);
      const size_t inend = no_symbols ? st[p].blocks[0].inend : instart + st[p].lz77.ByteRange(f.lstart, f.lend);
      f.fixed_request = static_cast<long>(fixed_requests.size());
      fixed_requests.push_back({instart, inend});
    }
  }

  // ---- 4. fixed-tree re-parse where it may win (deflate.c:770-781)
  const double tp2 = Now();
  // (with the device bit writer the re-parses stay on the device as well: their cost comes from their histograms,
  //  their bits — where the fixed tree wins — from zmx_encode_blocks.  Incompressible input asks for a re-parse of
  //  every block: downloading those stores and indexing them on the host was most of what such input cost.)
  std::vector<SymbolRun> fixed_runs;
  // (the tables of the optimal batch are only wanted for their stores from here on: the bit writer reads them.  Records,
  //  codes, window records and snapshots go back to the pool before the fixed-tree batch builds its own)
  if (keep.tables) {
    rc = zmx_tables_trim(ctx, keep.tables);
    if (rc) return rc;
  }
  OptimalKeep fkeep;
  fkeep.skip_download = device_encode;
  rc = Lz77OptimalFixedBatch(ctx, fixed_requests, &fixed_runs, device_encode ? &fkeep : nullptr);
  if (rc) return rc;
  TablesGuard fixed_guard{ctx, fkeep.tables};
  const double tp3 = Now();

  // ---- 5. pick the block type and encode
  const double t5 = Now();
  // One task per final BLOCK, not per part: a block the device cannot write (the second split attempt moved its
  // ends: its symbols are a stretch of several device blocks) has its bits made here, symbol by symbol — 4 ms for the
  // 250 000 symbols of one master block, a tenth of a 1 MB call when its blocks took turns on one thread.
  struct BlockTask { size_t p, i; };
  std::vector<BlockTask> block_tasks;
  for (size_t p = 0; p < np; ++p) {
    st[p].chunks.assign(st[p].finals.size(), Chunk());
    st[p].enc.assign(st[p].finals.size(), DeviceEncode());
    for (size_t i = 0; i < st[p].finals.size(); ++i) block_tasks.push_back({p, i});
  }
  ParallelForWide(block_tasks.size(), [&](size_t task) {
    const size_t p = block_tasks[task].p, i = block_tasks[task].i;
    PartState& s = st[p];
    const FinalBlock& f = s.finals[i];
    const bool final_block = (i + 1 == s.finals.size()) && s.part.final_part;
    BitWriter w;
    if (f.lstart == f.lend) {  // smallest empty block: fixed, end symbol only
      w.AddBits(final_block ? 1 : 0, 1);
      w.AddBits(1, 2);
      w.AddBits(0, 7);
      s.chunks[i] = BitsChunk(&w);
      return;
    }
    double fixedcost = f.fixed;
    Lz77Store fixedstore;
    Histogram fixedhist;
    if (f.expensive_fixed) {
      if (fkeep.tables) {
        const uint32_t* c = &fkeep.hist[static_cast<size_t>(f.fixed_request) * ZMX_HIST];
        for (int k = 0; k < kNumLL; ++k) fixedhist.ll[k] = c[k];
        for (int k = 0; k < kNumD; ++k) fixedhist.d[k] = c[kNumLL + k];
        fixedcost = BlockSizeFromHistogram(fixedhist, 1);
      } else {
        fixedstore = StoreFromRun(fixed_runs[f.fixed_request], fixed_requests[f.fixed_request].instart);
        fixedcost = CalculateBlockSize(fixedstore, 0, fixedstore.size(), 1);
      }
    }
    if (f.stored < fixedcost && f.stored < f.dynamic) {
      Chunk c;
      c.kind = Chunk::kStored;
      c.start = no_symbols ? s.blocks[0].instart : s.lz77.pos(f.lstart);
      c.end = no_symbols ? s.blocks[0].inend : c.start + s.lz77.ByteRange(f.lstart, f.lend);
      c.final_block = final_block;
      s.chunks[i] = std::move(c);
      return;
    }
    size_t tree_bits = 0;
    const int used_btype = fixedcost < f.dynamic ? 1 : 2;
    // the symbols of a block that is a whole block of the optimal batch are still on the device: it writes them
    long dev_block = -1;
    const bool fixed_on_device = used_btype == 1 && f.expensive_fixed && fkeep.tables != nullptr;
    if (fixed_on_device) dev_block = f.fixed_request;
    if (keep.tables && !(used_btype == 1 && f.expensive_fixed)) {
      for (size_t k = 0; k < s.block_sym_end.size(); ++k) {
        if (s.block_sym_end[k] == f.lend && (k == 0 ? 0 : s.block_sym_end[k - 1]) == f.lstart) dev_block = static_cast<long>(s.first_block + k);
      }
    }
    Chunk c;
    if (dev_block >= 0) {
      DeviceEncode e;
      e.from_fixed = fixed_on_device;
      Histogram h;
      if (fixed_on_device) h = fixedhist;          // (not read for btype 1)
      else if (no_symbols) h = block_hist(s.first_block);
      else s.lz77.GetHistogram(f.lstart, f.lend, &h);
      e.data_bits = EncodeBlockHeader(h, used_btype, final_block, &w, &tree_bits, e.codes);
      e.header = w.Finish(&e.header_bits);
      e.block = static_cast<size_t>(dev_block);
      e.chunk = i;
      e.used = true;
      c.kind = Chunk::kBits;
      c.nbits = e.header_bits + e.data_bits;
      s.enc[i] = std::move(e);
    } else {
      if (used_btype == 1) {
        if (f.expensive_fixed) {
          EncodeBlock(fixedstore, 0, fixedstore.size(), 1, final_block, &w);
        } else {
          EncodeBlock(s.lz77, f.lstart, f.lend, 1, final_block, &w);
        }
      } else {
        EncodeBlock(s.lz77, f.lstart, f.lend, 2, final_block, &w, &tree_bits);
      }
      c = BitsChunk(&w);
    }
    c.log_block = true;
    c.log_btype = used_btype;
    c.log_tree_bits = tree_bits;
    c.log_unc = no_symbols ? s.blocks[0].inend - s.blocks[0].instart : s.lz77.ByteRange(f.lstart, f.lend);
    s.chunks[i] = std::move(c);
  });
  for (size_t p = 0; p < np; ++p) {
    PartState& s = st[p];
    s.enc.erase(std::remove_if(s.enc.begin(), s.enc.end(), [](const DeviceEncode& e) { return !e.used; }), s.enc.end());
    if (options.verbose && !s.chunks.empty()) s.chunks.front().log_pre = std::move(s.log);
  }
  // ---- 5b. the device writes the symbols of its blocks behind the headers (the blocks of the optimal batch, then
  //          those of the fixed-tree re-parses)
  for (int pass = 0; pass < 2; ++pass) {
    const OptimalKeep& kp = pass == 0 ? keep : fkeep;
    if (!kp.tables) continue;
    std::vector<zmx_enc_job> jobs;
    std::vector<uint32_t> codes;
    std::vector<unsigned char*> outs;
    std::vector<std::pair<size_t, size_t>> owner;    // (part, index in its enc)
    for (size_t p = 0; p < np; ++p) {
      for (size_t k = 0; k < st[p].enc.size(); ++k) {
        DeviceEncode& e = st[p].enc[k];
        if (e.from_fixed != (pass == 1)) continue;
        zmx_enc_job j;
        j.block = static_cast<uint32_t>(e.block);
        j.slot = kp.slot[e.block];
        j.nsym = kp.nsym[e.block];
        j.bit_start = static_cast<uint32_t>(e.header_bits);
        j.nbits = e.data_bits;
        jobs.push_back(j);
        codes.insert(codes.end(), e.codes, e.codes + 320);
        owner.push_back({p, k});
      }
    }
    if (jobs.empty()) continue;
    // (the chunks' memory — a third of the input's size in all — is allocated and touched by the workers, not by
    //  this thread: first-touch page faults of 30 MB were most of what this phase took)
    outs.resize(jobs.size());
    ParallelFor(jobs.size(), [&](size_t i) {
      Chunk& c = st[owner[i].first].chunks[st[owner[i].first].enc[owner[i].second].chunk];
      c.bits.assign((c.nbits + 7) / 8, 0);
      outs[i] = c.bits.data();
    });
    rc = zmx_encode_blocks(ctx, kp.tables, jobs.size(), jobs.data(), codes.data(), outs.data());
    if (rc) return rc;
    ParallelFor(jobs.size(), [&](size_t i) {
      const DeviceEncode& e = st[owner[i].first].enc[owner[i].second];
      uint8_t* b = st[owner[i].first].chunks[e.chunk].bits.data();
      for (size_t k = 0; k < e.header.size(); ++k) b[k] |= e.header[k];
    });
  }
  ThreadTiming().encode += Now() - t5;

  const double tp4 = Now();
  for (size_t p = 0; p < np; ++p) {
    for (auto& c : st[p].chunks) chunks->push_back(std::move(c));
  }
  if (trace_phases) {
    std::fprintf(stderr, "DeflateParts: optimal batch %.1f ms, join/split %.1f ms, fixed batch %.1f ms (%zu requests), "
                 "encode %.1f ms, chunk move %.1f ms\n", (tp1 - tp0) * 1e3, (tp2 - tp1) * 1e3, (tp3 - tp2) * 1e3,
                 fixed_requests.size(), (tp4 - tp3) * 1e3, (Now() - tp4) * 1e3);
  }
  return 0;
}

namespace {

// Copies bits [s0, s0 + n) of `src` (LSB-first) to dst[0 .. n / 8): n is a multiple of 8.
void CopyBitRun(const uint8_t* src, size_t src_bytes, size_t s0, size_t n, uint8_t* dst) {
  const size_t nb = n / 8, byte0 = s0 / 8;
  const unsigned r = static_cast<unsigned>(s0 & 7);
  if (r == 0) {
    std::memcpy(dst, src + byte0, nb);
    return;
  }
  size_t i = 0;
  for (; i + 8 <= nb && byte0 + i + 9 <= src_bytes; i += 8) {   // 8 output bytes from 9 source bytes
    uint64_t v;
    std::memcpy(&v, src + byte0 + i, 8);
    const uint64_t out = (v >> r) | (static_cast<uint64_t>(src[byte0 + i + 8]) << (64 - r));
    std::memcpy(dst + i, &out, 8);
  }
  for (; i < nb; ++i) {
    const unsigned lo = src[byte0 + i];
    const unsigned hi = byte0 + i + 1 < src_bytes ? src[byte0 + i + 1] : 0u;
    dst[i] = static_cast<uint8_t>((lo >> r) | (hi << (8 - r)));
  }
}

// bits [s0, s0 + n) of src, n <= 8, in the low bits of the result
unsigned PeekBits(const uint8_t* src, size_t src_bytes, size_t s0, unsigned n) {
  const size_t byte0 = s0 / 8;
  const unsigned r = static_cast<unsigned>(s0 & 7);
  unsigned v = src[byte0];
  if (byte0 + 1 < src_bytes) v |= static_cast<unsigned>(src[byte0 + 1]) << 8;
  return (v >> r) & ((1u << n) - 1u);
}

}  // namespace

// Joins the chunks at the running bit position of `stream`.  Every chunk's place in the output is
// known from a prefix sum of bit lengths, so the chunks are shifted into place in parallel (each
// writes only the bytes it owns entirely); the <= 2 bytes a chunk shares with its neighbours are
// OR-ed in afterwards.  Stored blocks follow AddNonCompressedBlock (deflate.c:625-665): pieces of
// at most 65535 bytes, each byte-aligned after its 3 header bits.
void MergeChunks(const std::vector<Chunk>& chunks, const unsigned char* in, unsigned char* bp,
                 unsigned char** outp, size_t* outsize, bool verbose) {
  struct Place { size_t bit0; };
  std::vector<Place> place(chunks.size());
  // bit position 0 = the first bit of (*outp)[0]; *bp bits of the last byte are used
  unsigned bp0 = *bp & 7u;
  if (*outsize == 0) bp0 = 0;
  const size_t begin = bp0 ? (*outsize - 1) * 8 + bp0 : *outsize * 8;
  size_t cur = begin;
  for (size_t i = 0; i < chunks.size(); ++i) {
    const Chunk& c = chunks[i];
    place[i].bit0 = cur;
    if (c.kind == Chunk::kBits) {
      cur += c.nbits;
    } else {
      size_t pos = c.start;
      for (;;) {
        size_t piece = 65535;
        if (pos + piece > c.end) piece = c.end - pos;
        cur = (cur + 3 + 7) / 8 * 8 + 8 * (4 + piece);
        if (pos + piece >= c.end) break;
        pos += piece;
      }
    }
  }
  if (verbose) {
    // what the reference prints while it writes the blocks (deflate.c:719-744): byte counts of the
    // growing output array, i.e. differences of ceil(bit position / 8)
    for (size_t i = 0; i < chunks.size(); ++i) {
      const Chunk& c = chunks[i];
      if (!c.log_pre.empty()) std::fputs(c.log_pre.c_str(), stderr);
      if (c.kind != Chunk::kBits || !c.log_block) continue;
      const size_t p1 = place[i].bit0 + 3, p2 = p1 + c.log_tree_bits, p3 = place[i].bit0 + c.nbits;
      if (c.log_btype == 2) std::fprintf(stderr, "treesize: %d\n", static_cast<int>((p2 + 7) / 8 - (p1 + 7) / 8));
      const size_t compressed = (p3 + 7) / 8 - (p2 + 7) / 8;
      std::fprintf(stderr, "compressed block size: %d (%dk) (unc: %d)\n", static_cast<int>(compressed),
                   static_cast<int>(compressed / 1024), static_cast<int>(c.log_unc));
    }
  }
  const size_t newsize = (cur + 7) / 8;
  const size_t keep = *outsize;   // bytes [0, keep) hold earlier output
  ReserveOutput(newsize - *outsize, outp, outsize, /*zero=*/false);
  uint8_t* const out = *outp;
  // Every new byte is either written whole by the chunk that owns it (below) or is one of the few
  // bytes that are OR-ed together afterwards: only those are cleared (not the whole range).
  auto clear = [&](size_t idx) { if (idx >= keep) out[idx] = 0; };
  for (size_t i = 0; i < chunks.size(); ++i) {
    const Chunk& c = chunks[i];
    const size_t b0 = place[i].bit0;
    if (c.kind == Chunk::kBits) {
      if (c.nbits == 0) continue;
      const size_t e = b0 + c.nbits;
      if (e / 8 < (b0 + 7) / 8) { clear(b0 / 8); continue; }
      if (b0 & 7) clear(b0 / 8);
      if (e & 7) clear(e / 8);
    } else {
      size_t bit = b0, pos = c.start;
      for (;;) {
        size_t piece = 65535;
        if (pos + piece > c.end) piece = c.end - pos;
        clear(bit / 8);
        clear((bit + 2) / 8);
        bit = ((bit + 3 + 7) / 8 + 4 + piece) * 8;
        if (pos + piece >= c.end) break;
        pos += piece;
      }
    }
  }

  ParallelFor(chunks.size(), [&](size_t i) {
    const Chunk& c = chunks[i];
    const size_t b0 = place[i].bit0;
    if (c.kind == Chunk::kBits) {
      // bytes entirely inside [b0, b0 + nbits)
      const size_t first = (b0 + 7) / 8, last = (b0 + c.nbits) / 8;
      if (last > first) CopyBitRun(c.BitData(), c.BitBytes(), first * 8 - b0, (last - first) * 8, out + first);
      return;
    }
    const unsigned char* src = c.view ? c.view - c.start : c.raw.empty() ? in : c.raw.data() - c.start;
    size_t bit = b0, pos = c.start;
    for (;;) {
      size_t piece = 65535;
      if (pos + piece > c.end) piece = c.end - pos;
      const bool last = pos + piece >= c.end;
      uint8_t* p = out + (bit + 3 + 7) / 8;            // after the header bits and the padding
      const unsigned len = static_cast<unsigned>(piece), nlen = ~len & 0xffffu;
      p[0] = static_cast<uint8_t>(len & 255);
      p[1] = static_cast<uint8_t>(len >> 8);
      p[2] = static_cast<uint8_t>(nlen & 255);
      p[3] = static_cast<uint8_t>(nlen >> 8);
      std::memcpy(p + 4, src + pos, piece);
      bit = static_cast<size_t>(p + 4 + piece - out) * 8;
      if (last) break;
      pos += piece;
    }
  });

  // shared bytes, in stream order
  for (size_t i = 0; i < chunks.size(); ++i) {
    const Chunk& c = chunks[i];
    const size_t b0 = place[i].bit0;
    if (c.kind == Chunk::kBits) {
      if (c.nbits == 0) continue;
      const size_t e = b0 + c.nbits;
      const size_t first = (b0 + 7) / 8, last = e / 8;
      if (last < first) {                              // the whole chunk sits inside one byte
        out[b0 / 8] |= static_cast<uint8_t>(PeekBits(c.BitData(), c.BitBytes(), 0, static_cast<unsigned>(c.nbits)) << (b0 & 7));
        continue;
      }
      if (b0 & 7) {                                    // head: the free bits of the byte b0 falls in
        const unsigned n = 8 - static_cast<unsigned>(b0 & 7);
        out[b0 / 8] |= static_cast<uint8_t>(PeekBits(c.BitData(), c.BitBytes(), 0, n) << (b0 & 7));
      }
      if (e & 7) {                                     // tail: what is left after the last whole byte
        const unsigned n = static_cast<unsigned>(e & 7);
        out[last] |= static_cast<uint8_t>(PeekBits(c.BitData(), c.BitBytes(), c.nbits - n, n));
      }
    } else {
      size_t bit = b0, pos = c.start;                  // BFINAL of every piece; BTYPE 00 is already zero
      for (;;) {
        size_t piece = 65535;
        if (pos + piece > c.end) piece = c.end - pos;
        const bool last = pos + piece >= c.end;
        if (c.final_block && last) out[bit / 8] |= static_cast<uint8_t>(1u << (bit & 7));
        bit = ((bit + 3 + 7) / 8 + 4 + piece) * 8;
        if (last) break;
        pos += piece;
      }
    }
  }
  *outsize = newsize;
  *bp = static_cast<unsigned char>(cur & 7);
}

namespace {
uint64_t GetU64(const unsigned char* p) {
  uint64_t x = 0;
  for (int i = 0; i < 8; ++i) x |= static_cast<uint64_t>(p[i]) << (8 * i);
  return x;
}
}  // namespace

// Blob layout: u64 count, then per chunk: u8 kind, u8 final, u64 a, u64 b and a
// payload: bit chunks a = nbits, b = payload bytes; stored chunks a = 0,
// b = payload bytes (the raw input bytes of the block).
unsigned char* SerializeChunks(const std::vector<Chunk>& chunks, const unsigned char* in, size_t* size) {
  // every chunk's place follows from the payload sizes: headers serially, payloads in parallel
  std::vector<size_t> off(chunks.size());
  size_t total = 8;
  for (size_t i = 0; i < chunks.size(); ++i) {
    const Chunk& c = chunks[i];
    off[i] = total;
    total += 18 + (c.kind == Chunk::kBits ? c.BitBytes() : c.end - c.start);
  }
  unsigned char* blob = static_cast<unsigned char*>(std::malloc(total ? total : 1));
  if (!blob) return nullptr;
  auto put64 = [](unsigned char* p, uint64_t x) { for (int k = 0; k < 8; ++k) p[k] = static_cast<uint8_t>(x >> (8 * k)); };
  put64(blob, chunks.size());
  ParallelFor(chunks.size(), [&](size_t i) {
    const Chunk& c = chunks[i];
    unsigned char* p = blob + off[i];
    p[0] = static_cast<uint8_t>(c.kind);
    p[1] = c.final_block ? 1 : 0;
    if (c.kind == Chunk::kBits) {
      put64(p + 2, c.nbits);
      put64(p + 10, c.BitBytes());
      std::memcpy(p + 18, c.BitData(), c.BitBytes());
    } else {
      put64(p + 2, 0);
      put64(p + 10, c.end - c.start);
      const unsigned char* src = c.view ? c.view : c.raw.empty() ? in + c.start : c.raw.data();
      std::memcpy(p + 18, src, c.end - c.start);
    }
  });
  *size = total;
  return blob;
}

bool DeserializeChunks(const unsigned char* blob, size_t size, std::vector<Chunk>* chunks) {
  if (size < 8) return false;
  const uint64_t n = GetU64(blob);
  size_t off = 8;
  for (uint64_t i = 0; i < n; ++i) {
    if (size - off < 18) return false;
    Chunk c;
    c.kind = static_cast<Chunk::Kind>(blob[off]);
    c.final_block = blob[off + 1] != 0;
    const uint64_t a = GetU64(blob + off + 2), b = GetU64(blob + off + 10);
    off += 18;
    if (c.kind == Chunk::kBits) {
      if (b > size - off || (a + 7) / 8 > b) return false;
      c.nbits = a;
      c.view = blob + off;
      c.view_bytes = b;
      off += b;
    } else if (c.kind == Chunk::kStored) {
      if (b > size - off) return false;
      c.start = 0;
      c.end = b;
      c.view = blob + off;
      c.view_bytes = b;
      off += b;
    } else {
      return false;
    }
    chunks->push_back(std::move(c));
  }
  return off == size;
}

void ReserveOutput(size_t n, unsigned char** out, size_t* outsize, bool zero) {
  if (n == 0) return;
  const size_t newsize = *outsize + n;
  size_t cap = 1;
  while (cap < newsize) cap <<= 1;
  size_t oldcap = 0;
  if (*outsize > 0) {
    oldcap = 1;
    while (oldcap < *outsize) oldcap <<= 1;
  }
  if (*outsize == 0) {
    // ZOPFLI_APPEND_DATA mallocs when the size is 0, whatever *out holds (util.h:147): a caller may
    // leave *out uninitialised
    void* p = std::malloc(cap);
    if (!p) std::exit(-1);  // the reference also exits on allocation failure
    *out = static_cast<unsigned char*>(p);
  } else if (cap > oldcap) {
    void* p = std::realloc(*out, cap);
    if (!p) std::exit(-1);
    *out = static_cast<unsigned char*>(p);
  }
  if (zero) std::memset(*out + *outsize, 0, n);
}

void AppendToOutput(const uint8_t* data, size_t n, unsigned char** out, size_t* outsize) {
  if (n == 0) return;
  const size_t newsize = *outsize + n;
  // ZOPFLI_APPEND_DATA keeps capacity at the smallest power of two >= size and
  // doubles it when size itself is a power of two; leave the array in that state.
  size_t cap = 1;
  while (cap < newsize) cap <<= 1;
  size_t oldcap = 0;
  if (*outsize > 0) {
    oldcap = 1;
    while (oldcap < *outsize) oldcap <<= 1;
  }
  if (*outsize == 0) {
    void* p = std::malloc(cap);   // (as above: *out is not looked at while the size is 0)
    if (!p) std::exit(-1);  // the reference also exits on allocation failure
    *out = static_cast<unsigned char*>(p);
  } else if (cap > oldcap) {
    void* p = std::realloc(*out, cap);
    if (!p) std::exit(-1);
    *out = static_cast<unsigned char*>(p);
  }
  std::memcpy(*out + *outsize, data, n);
  *outsize = newsize;
}

}  // namespace zamd

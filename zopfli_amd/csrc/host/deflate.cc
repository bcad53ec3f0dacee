#include "deflate.h"

#include <chrono>
#include <cstdlib>
#include <cstring>

#include "block_cost.h"
#include "block_split.h"
#include "lz77_optimal.h"
#include "lz77_store.h"
#include "thread_pool.h"

namespace zamd {

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

struct PartState {
  Part part{};
  std::vector<zmx_block> blocks;     // first-pass deflate blocks (byte ranges)
  size_t first_block = 0;            // offset into the batched block list
  Lz77Store lz77;                    // optimal parse of the whole part
  std::vector<size_t> splitpoints;   // final split points (symbol indices)
  std::vector<FinalBlock> finals;
  std::vector<Chunk> chunks;
};

Lz77Store StoreFromRun(const SymbolRun& run, size_t pos) {
  Lz77Store s;
  s.Append(run.litlens.data(), run.dists.data(), run.litlens.size(), pos);
  return s;
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
      st[p].chunks.push_back(BitsChunk(&w));
    });
    for (size_t p = 0; p < np; ++p) {
      for (auto& c : st[p].chunks) chunks->push_back(std::move(c));
    }
    return 0;
  }

  // ---- 1. first block split on a greedy parse of each part (deflate.c:845-850,
  //         blocksplitter.c:275)
  std::vector<std::vector<size_t>> split_bytes(np);
  if (options.blocksplitting) {
    std::vector<zmx_block> ranges(np);
    for (size_t p = 0; p < np; ++p) ranges[p] = {parts[p].instart, parts[p].inend};
    std::vector<SymbolRun> greedy;
    rc = Lz77GreedyBatch(ctx, ranges, &greedy);
    if (rc) return rc;
    const double t0 = Now();
    ParallelFor(np, [&](size_t p) {
      Lz77Store s = StoreFromRun(greedy[p], parts[p].instart);
      std::vector<size_t> pts;
      BlockSplitLz77(s, static_cast<size_t>(options.blocksplittingmax), &pts);
      split_bytes[p] = SplitPointsToBytes(s, pts, parts[p].instart);
    });
    ThreadTiming().split += Now() - t0;
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
  std::vector<SymbolRun> runs;
  rc = Lz77OptimalBatch(ctx, options, all_blocks, &runs);
  if (rc) return rc;

  // ---- 3. join the blocks, second split attempt, per-block type costs
  std::vector<zmx_block> fixed_requests;
  std::vector<std::pair<size_t, size_t>> fixed_owner;  // (part, final index)
  const double t3 = Now();
  ParallelFor(np, [&](size_t p) {
    PartState& s = st[p];
    const size_t npoints = s.blocks.size() - 1;
    double totalcost = 0;
    for (size_t i = 0; i <= npoints; ++i) {
      Lz77Store bs = StoreFromRun(runs[s.first_block + i], s.blocks[i].instart);
      totalcost += CalculateBlockSizeAutoType(bs, 0, bs.size());
      s.lz77.Append(bs);
      if (i < npoints) s.splitpoints.push_back(s.lz77.size());
    }
    if (options.blocksplitting && npoints > 1) {  // deflate.c:872-893
      std::vector<size_t> pts2;
      BlockSplitLz77(s.lz77, static_cast<size_t>(options.blocksplittingmax), &pts2);
      double totalcost2 = 0;
      for (size_t i = 0; i <= pts2.size(); ++i) {
        const size_t a = i == 0 ? 0 : pts2[i - 1];
        const size_t b = i == pts2.size() ? s.lz77.size() : pts2[i];
        totalcost2 += CalculateBlockSizeAutoType(s.lz77, a, b);
      }
      if (totalcost2 < totalcost) s.splitpoints.swap(pts2);
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
  ThreadTiming().split += Now() - t3;

  for (size_t p = 0; p < np; ++p) {
    for (size_t i = 0; i < st[p].finals.size(); ++i) {
      FinalBlock& f = st[p].finals[i];
      if (f.lstart == f.lend || !f.expensive_fixed) continue;
      const size_t instart = st[p].lz77.pos(f.lstart);
      const size_t inend = instart + st[p].lz77.ByteRange(f.lstart, f.lend);
      f.fixed_request = static_cast<long>(fixed_requests.size());
      fixed_requests.push_back({instart, inend});
    }
  }

  // ---- 4. fixed-tree re-parse where it may win (deflate.c:770-781)
  std::vector<SymbolRun> fixed_runs;
  rc = Lz77OptimalFixedBatch(ctx, fixed_requests, &fixed_runs);
  if (rc) return rc;

  // ---- 5. pick the block type and encode
  const double t5 = Now();
  ParallelFor(np, [&](size_t p) {
    PartState& s = st[p];
    for (size_t i = 0; i < s.finals.size(); ++i) {
      const FinalBlock& f = s.finals[i];
      const bool final_block = (i + 1 == s.finals.size()) && s.part.final_part;
      BitWriter w;
      if (f.lstart == f.lend) {  // smallest empty block: fixed, end symbol only
        w.AddBits(final_block ? 1 : 0, 1);
        w.AddBits(1, 2);
        w.AddBits(0, 7);
        s.chunks.push_back(BitsChunk(&w));
        continue;
      }
      double fixedcost = f.fixed;
      Lz77Store fixedstore;
      if (f.expensive_fixed) {
        fixedstore = StoreFromRun(fixed_runs[f.fixed_request], fixed_requests[f.fixed_request].instart);
        fixedcost = CalculateBlockSize(fixedstore, 0, fixedstore.size(), 1);
      }
      if (f.stored < fixedcost && f.stored < f.dynamic) {
        Chunk c;
        c.kind = Chunk::kStored;
        c.start = s.lz77.pos(f.lstart);
        c.end = c.start + s.lz77.ByteRange(f.lstart, f.lend);
        c.final_block = final_block;
        s.chunks.push_back(std::move(c));
        continue;
      }
      if (fixedcost < f.dynamic) {
        if (f.expensive_fixed) {
          EncodeBlock(fixedstore, 0, fixedstore.size(), 1, final_block, &w);
        } else {
          EncodeBlock(s.lz77, f.lstart, f.lend, 1, final_block, &w);
        }
      } else {
        EncodeBlock(s.lz77, f.lstart, f.lend, 2, final_block, &w);
      }
      s.chunks.push_back(BitsChunk(&w));
    }
  });
  ThreadTiming().encode += Now() - t5;

  for (size_t p = 0; p < np; ++p) {
    for (auto& c : st[p].chunks) chunks->push_back(std::move(c));
  }
  return 0;
}

void MergeChunks(const std::vector<Chunk>& chunks, const unsigned char* in, BitStream* stream) {
  size_t total = stream->bytes.size() + 16;
  for (const Chunk& c : chunks) {
    total += c.kind == Chunk::kBits ? c.nbits / 8 + 1 : (c.end - c.start) + 5 * ((c.end - c.start) / 65535 + 1);
  }
  stream->bytes.reserve(total);   // one allocation instead of a doubling chain of copies
  for (const Chunk& c : chunks) {
    if (c.kind == Chunk::kBits) {
      stream->AppendBits(c.bits.data(), c.nbits);
      continue;
    }
    // AddNonCompressedBlock (deflate.c:625-665): pieces of at most 65535 bytes,
    // each byte-aligned after its 3 header bits.
    const unsigned char* src = c.raw.empty() ? in : c.raw.data() - c.start;
    size_t pos = c.start;
    for (;;) {
      size_t piece = 65535;
      if (pos + piece > c.end) piece = c.end - pos;
      const bool last = pos + piece >= c.end;
      stream->AppendBit(c.final_block && last);
      stream->AppendBit(0);
      stream->AppendBit(0);
      stream->bp = 0;  // rest of the byte is padding
      const unsigned len = static_cast<unsigned>(piece), nlen = ~len & 0xffffu;
      stream->AppendByteAligned(static_cast<uint8_t>(len & 255));
      stream->AppendByteAligned(static_cast<uint8_t>(len >> 8));
      stream->AppendByteAligned(static_cast<uint8_t>(nlen & 255));
      stream->AppendByteAligned(static_cast<uint8_t>(nlen >> 8));
      stream->bytes.insert(stream->bytes.end(), src + pos, src + pos + piece);
      if (last) break;
      pos += piece;
    }
  }
}

namespace {
void PutU64(std::vector<uint8_t>* v, uint64_t x) {
  for (int i = 0; i < 8; ++i) v->push_back(static_cast<uint8_t>(x >> (8 * i)));
}
uint64_t GetU64(const unsigned char* p) {
  uint64_t x = 0;
  for (int i = 0; i < 8; ++i) x |= static_cast<uint64_t>(p[i]) << (8 * i);
  return x;
}
}  // namespace

// Blob layout: u64 count, then per chunk: u8 kind, u8 final, u64 a, u64 b and a
// payload: bit chunks a = nbits, b = payload bytes; stored chunks a = 0,
// b = payload bytes (the raw input bytes of the block).
std::vector<uint8_t> SerializeChunks(const std::vector<Chunk>& chunks, const unsigned char* in) {
  std::vector<uint8_t> blob;
  PutU64(&blob, chunks.size());
  for (const Chunk& c : chunks) {
    blob.push_back(static_cast<uint8_t>(c.kind));
    blob.push_back(c.final_block ? 1 : 0);
    if (c.kind == Chunk::kBits) {
      PutU64(&blob, c.nbits);
      PutU64(&blob, c.bits.size());
      blob.insert(blob.end(), c.bits.begin(), c.bits.end());
    } else {
      PutU64(&blob, 0);
      PutU64(&blob, c.end - c.start);
      const unsigned char* src = c.raw.empty() ? in + c.start : c.raw.data();
      blob.insert(blob.end(), src, src + (c.end - c.start));
    }
  }
  return blob;
}

bool DeserializeChunks(const unsigned char* blob, size_t size, std::vector<Chunk>* chunks) {
  if (size < 8) return false;
  const uint64_t n = GetU64(blob);
  size_t off = 8;
  for (uint64_t i = 0; i < n; ++i) {
    if (off + 18 > size) return false;
    Chunk c;
    c.kind = static_cast<Chunk::Kind>(blob[off]);
    c.final_block = blob[off + 1] != 0;
    const uint64_t a = GetU64(blob + off + 2), b = GetU64(blob + off + 10);
    off += 18;
    if (c.kind == Chunk::kBits) {
      if (off + b > size || (a + 7) / 8 > b) return false;
      c.nbits = a;
      c.bits.assign(blob + off, blob + off + b);
      off += b;
    } else if (c.kind == Chunk::kStored) {
      if (off + b > size) return false;
      c.start = 0;
      c.end = b;
      c.raw.assign(blob + off, blob + off + b);
      if (b == 0) c.raw.clear();
      off += b;
    } else {
      return false;
    }
    chunks->push_back(std::move(c));
  }
  return off == size;
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
  if (cap > oldcap || *out == nullptr) {
    void* p = std::realloc(*out, cap);
    if (!p) std::exit(-1);  // the reference also exits on allocation failure
    *out = static_cast<unsigned char*>(p);
  }
  std::memcpy(*out + *outsize, data, n);
  *outsize = newsize;
}

}  // namespace zamd

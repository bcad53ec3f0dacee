// TEST-ONLY implementation of the zmx_* device layer on top of oracle/.
//
// Linked (by tests/hostlib/Makefile) with the product's host sources into
// tests/_build/libzopfli_hosttest.so so that the host logic (block splitter,
// block cost model, iteration control, encoder, containers) can be checked
// against the real reference on machines without a GPU.  The product library
// libzopfli_amd.so never links this file: its zmx_* symbols come from the HIP
// device layer only.
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "zopfli_amd.h"
#include "block_cost.h"
#include "block_split.h"
#include "checksum.h"
#include "huffman.h"
#include "lz77_store.h"
extern "C" {
#include "zopfli_oracle.h"
}

struct zmx_ctx {
  std::vector<unsigned char> input;
};

struct BlockData {
  zmx_block blk;
  zo_table* table = nullptr;
  std::vector<uint16_t> litlens[2], dists[2];
  size_t nsym[2] = {0, 0};
  std::vector<uint16_t> length_array;
};

struct zmx_tables {
  std::vector<BlockData> blocks;
};

static thread_local std::string g_err;

extern "C" {

// (ZOPFLI_HOSTTEST_DEVICES pretends to that many devices: the multi-device sharding of api.cc on CPU)
int zmx_device_count(void) {
  const char* e = std::getenv("ZOPFLI_HOSTTEST_DEVICES");
  return e ? std::atoi(e) : 1;
}
const char* zmx_last_error(void) { return g_err.c_str(); }
int zmx_has_experiments(void) { return 0; }
void zmx_set_kernel_timing(int) {}
int zmx_last_error_class(void) { return g_err.empty() ? ZMX_ERR_NONE : ZMX_ERR_DEVICE; }
void zmx_internal_set_error(const char* msg) { g_err = msg; }

int zmx_ctx_create(int, zmx_ctx** ctx) {
  *ctx = new zmx_ctx();
  return 0;
}
void zmx_ctx_destroy(zmx_ctx* ctx) { delete ctx; }

int zmx_set_input(zmx_ctx* ctx, const unsigned char* in, size_t insize) {
  ctx->input.assign(in, in + insize);
  return 0;
}
size_t zmx_internal_input_size(zmx_ctx* ctx) { return ctx->input.size(); }
const unsigned char* zmx_internal_input_host(zmx_ctx* ctx) { return ctx->input.data(); }
void zmx_internal_kernel_stats(double* a, double* b, int) { a[0] = a[1] = a[2] = 0; *b = 0; }
void zmx_internal_seg_stats(double* a, int) { for (int i = 0; i < 8; ++i) a[i] = 0; }
void zmx_internal_match_stats(double* a, int) { for (int i = 0; i < 4; ++i) a[i] = 0; }
void zmx_internal_match5_stats(double* a, int) { for (int i = 0; i < 3; ++i) a[i] = 0; }
void zmx_internal_stats_take(double* a) { for (int i = 0; i < 19; ++i) a[i] = 0; }
void zmx_internal_stats_add(const double*) {}
int zmx_hash_links_download(zmx_ctx*, zmx_tables*, size_t, uint16_t*, uint16_t*, uint16_t*) { g_err = "not in the host test library"; return -1; }
int zmx_match_digest(zmx_ctx*, zmx_tables*, uint64_t*) { g_err = "not in the host test library"; return -1; }
int zmx_set_match_kernel(int) { return 0; }   // (no kernels here)
void zmx_set_oom_hook(zmx_oom_hook_t) {}
int zmx_ctx_set_share(zmx_ctx*, unsigned) { return 0; }
int zmx_ctx_trim_cache(zmx_ctx*) { return 0; }
int zmx_ctx_set_priority(zmx_ctx*, int) { return 0; }
int zmx_png_filter_types(zmx_ctx*, const unsigned char*, size_t, size_t, size_t, unsigned char*, unsigned char*) { g_err = "not in the host test library"; return -1; }
// (the RCCL gather of dist.cc is not part of the host-logic test library)
int zmx_dist_unique_id(unsigned char*) { g_err = "no RCCL in the host test library"; return -1; }
int zmx_dist_init(zmx_ctx*, int, int, const unsigned char*, zmx_dist**) { g_err = "no RCCL in the host test library"; return -1; }
void zmx_dist_destroy(zmx_dist*) {}
int zmx_dist_comm_count(zmx_dist*) { return -1; }
int zmx_dist_gather(zmx_dist*, const unsigned char*, size_t, unsigned char**, size_t*) { g_err = "no RCCL in the host test library"; return -1; }

int zmx_tables_build(zmx_ctx* ctx, const zmx_block* blocks, size_t nblocks, zmx_tables** tables) {
  zmx_tables* t = new zmx_tables();
  t->blocks.resize(nblocks);
  for (size_t b = 0; b < nblocks; ++b) {
    BlockData& d = t->blocks[b];
    d.blk = blocks[b];
    d.table = zo_table_build(ctx->input.data(), blocks[b].instart, blocks[b].inend);
    const size_t B = blocks[b].inend - blocks[b].instart;
    for (int s = 0; s < 2; ++s) {
      d.litlens[s].resize(B + 1);
      d.dists[s].resize(B + 1);
    }
    d.length_array.resize(B + 1);
  }
  *tables = t;
  return 0;
}

int zmx_tables_trim(zmx_ctx*, zmx_tables*) { return 0; }

int zmx_tables_build_matches(zmx_ctx* ctx, const zmx_block* blocks, size_t nblocks, zmx_tables** tables) {
  return zmx_tables_build(ctx, blocks, nblocks, tables);
}

int zmx_tables_build_from(zmx_ctx* ctx, zmx_tables*, const zmx_block* blocks, size_t nblocks, zmx_tables** tables) {
  return zmx_tables_build(ctx, blocks, nblocks, tables);
}

void zmx_tables_free(zmx_ctx*, zmx_tables* t) {
  if (!t) return;
  for (auto& d : t->blocks) zo_table_free(d.table);
  delete t;
}

int zmx_lz77_greedy(zmx_ctx*, zmx_tables* t, int slot, uint32_t* nsym, uint32_t* hist) {
  for (size_t b = 0; b < t->blocks.size(); ++b) {
    BlockData& d = t->blocks[b];
    d.nsym[slot] = zo_greedy(d.table, d.litlens[slot].data(), d.dists[slot].data());
    nsym[b] = static_cast<uint32_t>(d.nsym[slot]);
    zo_histogram(d.litlens[slot].data(), d.dists[slot].data(), d.nsym[slot], hist + b * ZMX_HIST);
  }
  return 0;
}

int zmx_squeeze_run(zmx_ctx*, zmx_tables* t, const double* cost, const double* mincost, const int32_t* slot,
                    uint32_t* nsym, uint32_t* hist) {
  for (size_t b = 0; b < t->blocks.size(); ++b) {
    BlockData& d = t->blocks[b];
    const int s = slot[b];
    const double* ll = cost + b * ZMX_HIST;
    zo_get_best_lengths(d.table, ll, ll + ZMX_NUM_LL, mincost[b], d.length_array.data());
    d.nsym[s] = zo_trace_follow(d.table, d.length_array.data(), d.litlens[s].data(), d.dists[s].data());
    nsym[b] = static_cast<uint32_t>(d.nsym[s]);
    zo_histogram(d.litlens[s].data(), d.dists[s].data(), d.nsym[s], hist + b * ZMX_HIST);
  }
  return 0;
}

int zmx_store_download(zmx_ctx*, zmx_tables* t, size_t block, int slot, uint16_t* litlens, uint16_t* dists,
                       size_t nsym) {
  BlockData& d = t->blocks[block];
  if (nsym > d.nsym[slot]) return -1;
  std::memcpy(litlens, d.litlens[slot].data(), nsym * 2);
  std::memcpy(dists, d.dists[slot].data(), nsym * 2);
  return 0;
}

int zmx_store_download_batch(zmx_ctx* c, zmx_tables* t, size_t n, const size_t* block, const int32_t* slot,
                             const size_t* nsym, uint16_t* const* litlens, uint16_t* const* dists) {
  for (size_t i = 0; i < n; ++i) {
    const int rc = zmx_store_download(c, t, block[i], slot[i], litlens[i], dists[i], nsym[i]);
    if (rc) return rc;
  }
  return 0;
}

// (CPU stand-in for the device's verify pass)
int zmx_verify_stores(zmx_ctx* c, zmx_tables* t, size_t n, const size_t* block, const int32_t* slot, const size_t* nsym) {
  for (size_t i = 0; i < n; ++i) {
    const BlockData& d = t->blocks[block[i]];
    size_t pos = d.blk.instart;
    for (size_t k = 0; k < nsym[i]; ++k) {
      const unsigned litlen = d.litlens[slot[i]][k], dist = d.dists[slot[i]][k];
      if (dist == 0) {
        if (pos >= d.blk.inend || c->input[pos] != litlen) { g_err = "zmx_verify_stores: literal"; return -1; }
        pos += 1;
      } else {
        if (litlen < 3 || litlen > 258 || dist > pos || pos + litlen > d.blk.inend) { g_err = "zmx_verify_stores: range"; return -1; }
        for (unsigned q = 0; q < litlen; ++q) {
          if (c->input[pos + q] != c->input[pos + q - dist]) { g_err = "zmx_verify_stores: bytes"; return -1; }
        }
        pos += litlen;
      }
    }
    if (pos != d.blk.inend) { g_err = "zmx_verify_stores: coverage"; return -1; }
  }
  return 0;
}

// (CPU stand-in for k_checksum: the same pieces, lanes and tree, walked one after the other, so that the
// CPU suite exercises the arithmetic of host/checksum.cc the way the device feeds it)
int zmx_checksum(zmx_ctx* c, int kind, size_t begin, size_t end, uint32_t* value) {
  using namespace zamd;
  if (kind != ZMX_CRC32 && kind != ZMX_ADLER32) { g_err = "zmx_checksum: unknown kind"; return -1; }
  if (begin > end || end > c->input.size()) { g_err = "zmx_checksum: range outside the resident input"; return -1; }
  const size_t n = end - begin, npieces = (n + kChecksumPieceBytes - 1) / kChecksumPieceBytes;
  uint32_t xpow[8];
  ChecksumTreePowers(xpow);
  std::vector<ChecksumPiece> pieces(npieces);
  for (size_t w = 0; w < npieces; ++w) {
    const long long wg_end = static_cast<long long>(end) - static_cast<long long>(w) * kChecksumPieceBytes;
    uint32_t cr[256], sa[256], sb[256];
    for (int t = 0; t < 256; ++t) {
      const long long hi = wg_end - static_cast<long long>(255 - t) * kChecksumLaneBytes;
      long long p = hi - kChecksumLaneBytes;
      if (p < static_cast<long long>(begin)) p = static_cast<long long>(begin);
      uint32_t crc = 0, sum = 0, wsum = 0;
      for (; p < hi; ++p) {
        const uint32_t d = c->input[static_cast<size_t>(p)];
        crc ^= d;
        for (int k = 0; k < 8; ++k) crc = (crc & 1) ? (crc >> 1) ^ kCrcPoly : crc >> 1;
        sum += d;
        wsum += static_cast<uint32_t>(hi - p) * d;
      }
      cr[t] = crc; sa[t] = sum % kAdlerBase; sb[t] = wsum % kAdlerBase;
    }
    for (uint32_t s = 1, k = 0; s < 256; s <<= 1, ++k) {
      for (uint32_t t = 0; t < 256; t += 2 * s) {
        cr[t] = Gf2MulMod(cr[t], xpow[k]) ^ cr[t + s];
        sb[t] = static_cast<uint32_t>((sb[t] + static_cast<uint64_t>((s * kChecksumLaneBytes) % kAdlerBase) * sa[t] + sb[t + s]) % kAdlerBase);
        sa[t] = (sa[t] + sa[t + s]) % kAdlerBase;
      }
    }
    pieces[w] = ChecksumPiece{cr[0], sa[0], sb[0]};
  }
  *value = kind == ZMX_CRC32 ? FinishCrc32(pieces.data(), npieces, n) : FinishAdler32(pieces.data(), npieces, n);
  return 0;
}

// (CPU stand-in for the device bit writer: the same contract, symbol by symbol)
int zmx_encode_blocks(zmx_ctx*, zmx_tables* t, size_t njobs, const zmx_enc_job* jobs, const uint32_t* codes,
                      unsigned char* const* out) {
  for (size_t j = 0; j < njobs; ++j) {
    const zmx_enc_job& q = jobs[j];
    const BlockData& d = t->blocks[q.block];
    if (q.nsym > d.nsym[q.slot]) return -1;
    const uint32_t* cd = codes + j * 320;
    unsigned char* o = out[j];
    uint64_t pos = q.bit_start;
    auto put = [&](uint64_t v, unsigned n) {
      for (unsigned i = 0; i < n; ++i, ++pos) o[pos >> 3] |= static_cast<unsigned char>(((v >> i) & 1u) << (pos & 7));
    };
    auto flog2 = [](unsigned v) { return 31 - __builtin_clz(v); };
    for (size_t i = 0; i < q.nsym; ++i) {
      const unsigned litlen = d.litlens[q.slot][i], dist = d.dists[q.slot][i];
      if (dist == 0) {
        put(cd[litlen] & 0xffffu, cd[litlen] >> 16);
        continue;
      }
      unsigned ls, le;
      if (litlen < 11) { ls = 254 + litlen; le = 0; }
      else if (litlen == 258) { ls = 285; le = 0; }
      else { le = static_cast<unsigned>(flog2(litlen - 3) - 2); ls = 261 + 4 * le + (((litlen - 3) >> le) & 3); }
      unsigned ds, de;
      if (dist < 5) { ds = dist - 1; de = 0; }
      else { const int l = flog2(dist - 1); ds = static_cast<unsigned>(2 * l) + (((dist - 1) >> (l - 1)) & 1); de = static_cast<unsigned>(l - 1); }
      put(cd[ls] & 0xffffu, cd[ls] >> 16);
      put((litlen - 3) & ((1u << le) - 1), le);
      put(cd[288 + ds] & 0xffffu, cd[288 + ds] >> 16);
      put((dist - 1) & ((1u << de) - 1), de);
    }
    put(cd[256] & 0xffffu, cd[256] >> 16);
    if (pos - q.bit_start != q.nbits) return -1;
  }
  return 0;
}

int zmx_find_longest_match(zmx_ctx*, zmx_tables* t, size_t block, size_t pos, uint16_t* sublen,
                           uint16_t* distance, uint16_t* length) {
  zo_find_longest_match(t->blocks[block].table, pos, sublen, distance, length);
  return 0;
}

int zmx_length_array_download(zmx_ctx*, zmx_tables* t, size_t block, uint16_t* out) {
  const auto& la = t->blocks[block].length_array;
  std::memcpy(out, la.data(), la.size() * 2);
  return 0;
}

// ---- f-1's device entry points, TEST-ONLY stand-ins: the sequences as host stores, the sizes by the host's own block-cost
// code (so that DeflateParts' device-split path — queries, the mixing of device and host rounds — runs on CPU)
struct zmx_cost_stores {
  std::vector<zamd::Lz77Store> stores;
};
int zmx_cost_stores_create(zmx_ctx*, zmx_tables* t, size_t nstores, const size_t* piece_first, const size_t* block,
                           const int32_t* slot, const size_t* nsym, zmx_cost_stores** out) {
  zmx_cost_stores* s = new zmx_cost_stores();
  s->stores.resize(nstores);
  for (size_t q = 0; q < nstores; ++q) {
    for (size_t p = piece_first[q]; p < piece_first[q + 1]; ++p) {
      const BlockData& d = t->blocks[block[p]];
      if (nsym[p] > d.nsym[slot[p]]) { delete s; g_err = "zmx_cost_stores_create: nsym exceeds the store"; return -1; }
      s->stores[q].Append(d.litlens[slot[p]].data(), d.dists[slot[p]].data(), nsym[p], d.blk.instart);
    }
  }
  *out = s;
  return 0;
}
int zmx_cost_stores_create_host(zmx_ctx*, size_t nstores, const uint16_t* const* litlens, const uint16_t* const* dists,
                                const size_t* nsym, zmx_cost_stores** out) {
  zmx_cost_stores* s = new zmx_cost_stores();
  s->stores.resize(nstores);
  for (size_t q = 0; q < nstores; ++q) s->stores[q].Append(litlens[q], dists[q], nsym[q], 0);
  *out = s;
  return 0;
}
void zmx_cost_stores_free(zmx_ctx*, zmx_cost_stores* s) { delete s; }
int zmx_cost_positions(zmx_ctx*, zmx_cost_stores* s, size_t n, const uint32_t* pairs, uint64_t* bytes) {
  for (size_t i = 0; i < n; ++i) {
    if (pairs[2 * i] >= s->stores.size() || pairs[2 * i + 1] > s->stores[pairs[2 * i]].size()) { g_err = "zmx_cost_positions: an index outside its sequence"; return -1; }
    bytes[i] = s->stores[pairs[2 * i]].ByteRange(0, pairs[2 * i + 1]);
  }
  return 0;
}
int zmx_block_costs(zmx_ctx*, zmx_cost_stores* s, size_t n, const uint32_t* r, double* cost) {
  // (ZOPFLI_HOSTTEST_COSTS_FAIL_AFTER=k: the k-th call and every later one fails — the host's fall-back in mid-search)
  static const long fail_after = [] { const char* e = std::getenv("ZOPFLI_HOSTTEST_COSTS_FAIL_AFTER"); return e ? std::atol(e) : -1L; }();
  static std::atomic<long> calls{0};
  if (fail_after >= 0 && calls.fetch_add(1) >= fail_after) { g_err = "zmx_block_costs: injected failure"; return -1; }
  for (size_t i = 0; i < n; ++i) {
    if (r[3 * i] >= s->stores.size() || r[3 * i + 1] > r[3 * i + 2] || r[3 * i + 2] > s->stores[r[3 * i]].size()) { g_err = "zmx_block_costs: a range outside its sequence"; return -1; }
    cost[i] = zamd::CalculateBlockSizeAutoType(s->stores[r[3 * i]], r[3 * i + 1], r[3 * i + 2]);
  }
  return 0;
}

// ---- hooks for unit tests of the product's host arithmetic against the real reference
// (tests/test_cpu_oracle_vs_reference.py)
__attribute__((visibility("default"))) int zamd_test_code_lengths(const size_t* freq, int n, int maxbits, unsigned* lengths) {
  return zamd::LengthLimitedCodeLengths(freq, n, maxbits, lengths) ? 0 : 1;
}

__attribute__((visibility("default"))) double zamd_test_block_size_auto(const uint16_t* litlens, const uint16_t* dists, size_t n,
                                                                          size_t lstart, size_t lend) {
  zamd::Lz77Store s(nullptr);
  s.Append(litlens, dists, n, 0);
  return zamd::CalculateBlockSizeAutoType(s, lstart, lend);
}

__attribute__((visibility("default"))) size_t zamd_test_block_split(const uint16_t* litlens, const uint16_t* dists, size_t n,
                                                                      size_t maxblocks, size_t* points, size_t cap) {
  zamd::Lz77Store s(nullptr);
  s.Append(litlens, dists, n, 0);
  std::vector<size_t> pts;
  zamd::BlockSplitLz77(s, maxblocks, &pts);
  for (size_t i = 0; i < pts.size() && i < cap; ++i) points[i] = pts[i];
  return pts.size();
}

// BlockSplitLz77Batch over `copies` views of the sequence cut at different lengths (n, n - n / 7, n - 2 n / 7, ...):
// the split points of view v go to points[v * cap ..], their number to counts[v]
__attribute__((visibility("default"))) void zamd_test_block_split_batch(const uint16_t* litlens, const uint16_t* dists, size_t n, size_t copies,
                                                                          size_t maxblocks, size_t* points, size_t* counts, size_t cap) {
  std::vector<zamd::Lz77Store> stores;
  stores.reserve(copies);
  std::vector<const zamd::Lz77Store*> ptrs;
  for (size_t v = 0; v < copies; ++v) {
    stores.emplace_back(nullptr);
    stores.back().Append(litlens, dists, n - v * (n / 7), 0);
    ptrs.push_back(&stores.back());
  }
  std::vector<std::vector<size_t>> pts;
  zamd::BlockSplitLz77Batch(ptrs, maxblocks, &pts);
  for (size_t v = 0; v < copies; ++v) {
    counts[v] = pts[v].size();
    for (size_t i = 0; i < pts[v].size() && i < cap; ++i) points[v * cap + i] = pts[v][i];
  }
}

}  // extern "C"

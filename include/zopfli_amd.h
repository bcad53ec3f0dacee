/*
 * zopfli_amd — MI355X-native drop-in for the LZ77 optimal-parse hot path of
 * google/zopfli.  C ABI of libzopfli_amd.so.
 *
 * Part 1 is the reference's own public surface (same names, same struct
 * layout, same ownership rules), so existing callers of libzopfli — the CLI
 * (zopfli_bin.c:112), the cgo binding (go/zopfli/zopfli.go:46) and zopflipng's
 * CustomPNGDeflate (zopflipng_lib.cc:60) — link against this library unchanged.
 *
 * Part 2 is the thin device layer the host code calls into ("zmx_*"): plain
 * pointers and sizes, opaque handles, int status (0 = ok), no exceptions.
 * Each entry point names the reference function whose work it replaces.
 * There is no CPU fallback: every zmx_* call fails (non-zero) and the
 * Zopfli* calls abort with a message if no gfx950 device is usable.
 */
#ifndef ZOPFLI_AMD_H_
#define ZOPFLI_AMD_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* libzopfli_amd.so is built with -fvisibility=hidden: exactly the functions declared here are
 * exported (plus one test hook). */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

/* Errors.  The reference's entry points return void; like the reference (which exit()s when malloc
 * fails, util.h:135-155) the Zopfli* functions below print a message to stderr and exit(EXIT_FAILURE)
 * when the device side fails: no usable gfx950 device, a HIP error such as an allocation that does
 * not fit in HBM, or a single deflate block whose DP edges do not fit 32-bit row offsets (more than
 * 2^32 - 65536 match-length candidates in one block: some 150 - 300 MB of input in ONE block, which only
 * ZopfliDeflatePart with blocksplitting = 0 can ask for — ZopfliDeflate and ZopfliCompress cut their
 * input into 1 000 000-byte master blocks first).  The zmx_* functions of part 2 return non-zero instead
 * and leave a message for zmx_last_error().
 *
 * Devices and threads.  The Zopfli* functions run on ONE device unless told otherwise: ZOPFLI_AMD_DEVICE,
 * else LOCAL_RANK, else device 0; ZOPFLI_AMD_DEVICES = "all" | a count | a list of indices deals the master
 * blocks of a call over several.  They may be called from several threads at once (as the reference's may):
 * a device has ZOPFLI_AMD_LANES contexts (default 3), a call takes what it needs and later callers wait. */

/* ------------------------------------------------------------------ Part 1 */

/* reference: src/zopfli/zopfli.h:33-64 (six ints, this order) */
typedef struct ZopfliOptions {
  int verbose;
  int verbose_more;
  int numiterations;
  int blocksplitting;
  int blocksplittinglast; /* unused, kept for layout compatibility */
  int blocksplittingmax;
} ZopfliOptions;

/* reference: src/zopfli/zopfli.h:70-74 */
typedef enum {
  ZOPFLI_FORMAT_GZIP,
  ZOPFLI_FORMAT_ZLIB,
  ZOPFLI_FORMAT_DEFLATE
} ZopfliFormat;

/* reference: src/zopfli/util.c:28 — numiterations 15, blocksplitting 1, max 15 */
void ZopfliInitOptions(ZopfliOptions* options);

/* reference: src/zopfli/zopfli_lib.c:28.  Appends to the malloc'ed array
 * (*out, *outsize); caller frees.  Capacity keeps the reference's
 * power-of-two invariant (util.h:135-155) so callers may keep appending. */
void ZopfliCompress(const ZopfliOptions* options, ZopfliFormat output_type,
                    const unsigned char* in, size_t insize,
                    unsigned char** out, size_t* outsize);

/* reference: src/zopfli/deflate.c:908 (deflate.h:58).  `bp` = bits already used
 * in the last output byte (0..7), carried between calls. */
void ZopfliDeflate(const ZopfliOptions* options, int btype, int final,
                   const unsigned char* in, size_t insize,
                   unsigned char* bp, unsigned char** out, size_t* outsize);

/* reference: src/zopfli/deflate.c:811 (deflate.h:67).  Reads only
 * in[max(0,instart-32768) .. inend). */
void ZopfliDeflatePart(const ZopfliOptions* options, int btype, int final,
                       const unsigned char* in, size_t instart, size_t inend,
                       unsigned char* bp, unsigned char** out, size_t* outsize);

/* reference: src/zopfli/gzip_container.c:84 */
void ZopfliGzipCompress(const ZopfliOptions* options,
                        const unsigned char* in, size_t insize,
                        unsigned char** out, size_t* outsize);

/* reference: src/zopfli/zlib_container.c:50 */
void ZopfliZlibCompress(const ZopfliOptions* options,
                        const unsigned char* in, size_t insize,
                        unsigned char** out, size_t* outsize);

/* reference: src/zopfli/lz77.h:44-62 — the LZ77 symbol sequence of the reference's own API, as far as the two
 * functions below read it: litlens / dists (lz77.c:98: a literal is (byte, 0), a match (length, distance)), their
 * number, and pos (the byte position of every symbol in the input).  The cumulative histograms the reference keeps
 * beside them (ll_symbol ... d_counts) are not read here. */
typedef struct ZopfliLZ77Store {
  unsigned short* litlens;
  unsigned short* dists;
  size_t size;
  const unsigned char* data;
  size_t* pos;
  unsigned short* ll_symbol;
  unsigned short* d_symbol;
  size_t* ll_counts;
  size_t* d_counts;
} ZopfliLZ77Store;

/* reference: src/zopfli/deflate.c:584-608 (deflate.h:79).  Size in bits of symbols [lstart, lend) as one block of
 * type `btype` (0 stored, 1 fixed, 2 dynamic: tree included).  Host arithmetic (the block-cost code the block
 * splitter and the iteration driver use: host/block_cost.cc); no device call. */
double ZopfliCalculateBlockSize(const ZopfliLZ77Store* lz77, size_t lstart, size_t lend, int btype);

/* reference: src/zopfli/deflate.c:610-621 (deflate.h:85): the smallest of the three. */
double ZopfliCalculateBlockSizeAutoType(const ZopfliLZ77Store* lz77, size_t lstart, size_t lend);

/* ------------------------------------------------------------------ Part 2 */

typedef struct zmx_ctx zmx_ctx;       /* one HIP device + stream + resident input */
typedef struct zmx_tables zmx_tables; /* match tables + LZ77 stores of a batch of blocks */

/* One deflate block: bytes [instart, inend) of the resident input.  Matches may
 * reach back to max(0, instart-32768) and never extend past inend
 * (lz77.c:551-552, squeeze.c:229-230). */
typedef struct zmx_block {
  uint64_t instart;
  uint64_t inend;
} zmx_block;

#define ZMX_NUM_LL 288
#define ZMX_NUM_D 32
#define ZMX_HIST (ZMX_NUM_LL + ZMX_NUM_D) /* litlen bins then dist bins */

int zmx_device_count(void);
/* The message of the last failed zmx_* call on the calling thread. */
const char* zmx_last_error(void);
/* ... and what kind of failure it was.  Callers that decide whether doing the work again elsewhere can help (api.cc: a
 * failed shard is done again on another context) go by this code, never by the message's text. */
#define ZMX_ERR_NONE 0
#define ZMX_ERR_DEVICE 1         /* a HIP call failed or the device misbehaved: another context / device may fare better */
#define ZMX_ERR_OUT_OF_MEMORY 2  /* hipErrorOutOfMemory: likewise (another context may have the room) */
#define ZMX_ERR_REFUSED 3        /* the request itself is refused — bad arguments, a size limit, a pool that still
                                    overflows after the device layer's own retries: it would fail the same way anywhere */
int zmx_last_error_class(void);
/* The host arrays of a call (symbol stores, bit buffers) take their memory from a cache of large blocks the library keeps
 * between calls (at most ZOPFLI_AMD_HOST_CACHE_MB, default 1024; zopfli_amd/csrc/host/block_cache.h) instead of changing
 * the host process's malloc settings.  This frees everything cached and returns the number of bytes given back. */
size_t zmx_host_cache_trim(void);
/* 1 in a build with -DZMX_EXPERIMENTS: the kernels that lost their measurement (k_bucket + k_match3 / k_match4, the four-wave
 * run task k_dp6_spec) are compiled in and selectable (zmx_set_match_kernel 3 / 4, ZOPFLI_AMD_COOP=1); 0 in the shipped
 * library, which does not contain them. */
int zmx_has_experiments(void);

int zmx_ctx_create(int device, zmx_ctx** ctx);
void zmx_ctx_destroy(zmx_ctx* ctx);

/* Copies the whole input to HBM once; the kernels work on that copy.  The host side keeps reading a
 * few bytes of `in` itself (the ends of blocks when match tables are reused, the bytes of stored
 * blocks): `in` must stay valid and unchanged until the next zmx_set_input or zmx_ctx_destroy on
 * this context.  The calling thread's current HIP device is left as it was. */
int zmx_set_input(zmx_ctx* ctx, const unsigned char* in, size_t insize);

/* Kernel A.  For every block: the static hash arrays (hash.c:100-137 val/same/
 * prev links as pure functions of the data and inend) and, for every position,
 * the result of ZopfliFindLongestMatch(limit 258, sublen) (lz77.c:407-542):
 * longest length, its distance and the sublen step function.  Replaces the
 * hash replay + longest-match cache (cache.c) of the reference. */
int zmx_tables_build(zmx_ctx* ctx, const zmx_block* blocks, size_t nblocks, zmx_tables** tables);

/* The same without what only zmx_squeeze_run reads (the DP rows and their weight codes: two bytes for each of the up to
 * 258 edges of a position): for zmx_lz77_greedy, zmx_store_download / zmx_verify of its store, and as `parent` of
 * zmx_tables_build_from — the greedy pass over a master block that is about to be split (blocksplitter.c:296).
 * zmx_squeeze_run on such tables fails. */
int zmx_tables_build_matches(zmx_ctx* ctx, const zmx_block* blocks, size_t nblocks, zmx_tables** tables);

/* The same, for blocks that lie inside blocks of `parent` (both lists ascending), e.g. the deflate
 * blocks a master block was split into (deflate.c:854-869) after the greedy pass over the master
 * block (blocksplitter.c:296): ZopfliFindLongestMatch depends on the block only through its end
 * (lz77.c:448-450, hash.c:107-108,121), so the parent's results are reused for every position but
 * those near the new block ends.  `parent` stays valid for zmx_tables_free only.  parent = NULL
 * is zmx_tables_build. */
int zmx_tables_build_from(zmx_ctx* ctx, zmx_tables* parent, const zmx_block* blocks, size_t nblocks,
                          zmx_tables** tables);
void zmx_tables_free(zmx_ctx* ctx, zmx_tables* tables);

/* Gives back everything of a table set but its two LZ77 stores: afterwards only zmx_store_download*, zmx_encode_blocks
 * and zmx_tables_free work on it (the others fail with a message).  For the caller that keeps the tables of a finished
 * batch around for the device's bit writer while it builds the next ones (deflate.c:760: the fixed-tree re-parses). */
int zmx_tables_trim(zmx_ctx* ctx, zmx_tables* tables);

/* ZopfliLZ77Greedy (lz77.c:544-630) on every block, into store slot `slot`
 * (0 or 1).  nsym[b] = symbols emitted, hist[b*ZMX_HIST..] = their histogram
 * (litlen bins 0..287 then dist bins 0..31, end symbol not counted). */
int zmx_lz77_greedy(zmx_ctx* ctx, zmx_tables* tables, int slot, uint32_t* nsym, uint32_t* hist);

/* One LZ77OptimalRun (squeeze.c:429): GetBestLengths (:217) forward DP with the
 * given per-block cost model, TraceBackwards (:317), FollowPath (:338).
 * cost[b*ZMX_HIST..] = ll_symbols[288] then d_symbols[32] in bits, mincost[b] =
 * GetCostModelMinCost (:163).  slot[b] selects the store written for block b. */
int zmx_squeeze_run(zmx_ctx* ctx, zmx_tables* tables, const double* cost, const double* mincost,
                    const int32_t* slot, uint32_t* nsym, uint32_t* hist);

/* Copies store `slot` of block `block` (nsym entries) to the host: litlens[i] is
 * a literal byte when dists[i]==0, else a match length (lz77.h:44-49). */
int zmx_store_download(zmx_ctx* ctx, zmx_tables* tables, size_t block, int slot,
                       uint16_t* litlens, uint16_t* dists, size_t nsym);

/* The same for n stores at once (one device-to-host transfer set, one synchronisation):
 * store slot[i] of block block[i], nsym[i] entries, into litlens[i] / dists[i]. */
int zmx_store_download_batch(zmx_ctx* ctx, zmx_tables* tables, size_t n, const size_t* block,
                             const int32_t* slot, const size_t* nsym, uint16_t* const* litlens,
                             uint16_t* const* dists);

/* ZopfliVerifyLenDist (lz77.c:270-295) over whole stores, on the device: every literal is the input byte at its
 * position, every (length, distance) is in range and copies equal bytes, and the symbols cover the block exactly.
 * The reference asserts this for each symbol it stores (lz77.c:115, debug builds); here it is a separate pass,
 * run by the Zopfli* entry points on every parse they keep when ZOPFLI_AMD_VERIFY is set.  Returns 0 or fails
 * with the first offending block and symbol in zmx_last_error. */
int zmx_verify_stores(zmx_ctx* ctx, zmx_tables* tables, size_t n, const size_t* block, const int32_t* slot,
                      const size_t* nsym);

/* The deflate bit writer on the device: AddLZ77Data + the end symbol (deflate.c:297-333, :735-737) of a compressed
 * block whose LZ77 symbols are a whole store of `tables` — only the bits come down, not the symbols.
 * codes[j * 320 + s] = (Huffman code of symbol s, bit-reversed as the stream wants it) | (code length << 16) for
 * the 288 litlen symbols, then the 32 distance symbols, of job j (LengthsToSymbols, tree.c:50-69, on the host).
 * Job j's symbols start at bit `bit_start` of out[j] (the caller puts the block header and the tree in front: those
 * bits come back zero), out[j] has room for (bit_start + nbits + 7) / 8 bytes; `nbits` = what the caller expects
 * the symbols and the end symbol to take (from the histogram) and is checked. */
typedef struct zmx_enc_job {
  uint32_t block;      /* block of the table set */
  int32_t slot;        /* which of its two stores */
  uint32_t nsym;       /* symbols in that store */
  uint32_t bit_start;
  uint64_t nbits;
} zmx_enc_job;
int zmx_encode_blocks(zmx_ctx* ctx, zmx_tables* tables, size_t njobs, const zmx_enc_job* jobs,
                      const uint32_t* codes, unsigned char* const* out);

/* CRC-32 (gzip_container.c:75, the value ZopfliGzipCompress writes at :107-110) or Adler-32
 * (zlib_container.c:29, written at :71-74) of bytes [begin, end) of the resident input
 * (zmx_set_input), computed on the device; f-2 of SURVEY 8.  `value` is the finished checksum of
 * the range on its own.  Ranges that lie on different devices or ranks are put together with
 * zmx_checksum_combine(kind, checksum of A, checksum of B, bytes in B) = checksum of A followed
 * by B (host arithmetic only, no context needed). */
#define ZMX_CRC32 0
#define ZMX_ADLER32 1
int zmx_checksum(zmx_ctx* ctx, int kind, size_t begin, size_t end, uint32_t* value);
uint32_t zmx_checksum_combine(int kind, uint32_t a, uint32_t b, uint64_t len_b);

/* Parity probe: the ZopfliFindLongestMatch result for one position of one
 * block, expanded to the reference's sublen[259] convention. */
int zmx_find_longest_match(zmx_ctx* ctx, zmx_tables* tables, size_t block, size_t pos,
                           uint16_t* sublen, uint16_t* distance, uint16_t* length);

/* -------- zopflipng's per-row filter search
 *
 * The PNG filter type (0 None, 1 Sub, 2 Up, 3 Average, 4 Paeth) LodePNG's encoder picks for each of `height` scanlines
 * of `linebytes` bytes under its LFS_MINSUM and LFS_ENTROPY strategies (lodepng.cpp:5444-5570; the search zopflipng
 * runs per strategy trial, zopflipng_lib.cc:160-305), computed on the device from the raw (already colour-converted,
 * non-interlaced) image; `bytewidth` = bytes per pixel, 1 below 8 bits per pixel.  Either output may be null.  The
 * types go back to LodePNG as LFS_PREDEFINED (`predefined_filters`): the scanlines it then writes are the ones its own
 * search would have produced.  zmx_png_filter_types_pooled is the same on one of the contexts of the Zopfli* entry
 * points (what libzopflipng_amd.so calls). */
int zmx_png_filter_types(zmx_ctx* ctx, const unsigned char* image, size_t linebytes, size_t height, size_t bytewidth,
                         unsigned char* minsum_types, unsigned char* entropy_types);
int zmx_png_filter_types_pooled(const unsigned char* image, size_t linebytes, size_t height, size_t bytewidth,
                                unsigned char* minsum_types, unsigned char* entropy_types);

/* Parity probe over whole tables: two 64-bit sums over all positions of a hash of the logical content of the match
 * records (block, position, length, distance, same, literal, every change point of sublen).  Equal digests of two
 * table sets over the same blocks = the same ZopfliFindLongestMatch results at every position. */
int zmx_match_digest(zmx_ctx* ctx, zmx_tables* tables, uint64_t* out2);

/* Parity probe: the static hash arrays of one block for positions max(0, instart - 32768) .. inend - 1
 * (inend - windowstart entries each): same[] (hash.c:116-126) and the distances to the previous
 * position of the same hash / of the same second hash (hash.c:110-114, 129-135; 0 = none).
 * Fails for tables built with zmx_tables_build_from and a parent: those hold the hash arrays only
 * where the recomputed positions at the block ends read them. */
int zmx_hash_links_download(zmx_ctx* ctx, zmx_tables* tables, size_t block, uint16_t* same,
                            uint16_t* prev1, uint16_t* prev2);

/* Parity probe: length_array[0..blocksize] of the last zmx_squeeze_run. */
int zmx_length_array_download(zmx_ctx* ctx, zmx_tables* tables, size_t block, uint16_t* out);

/* -------- f-1 of SURVEY 8: the block-split search's cost function on the device
 *
 * ZopfliCalculateBlockSizeAutoType (deflate.c:610-621) of MANY ranges of LZ77 symbol sequences at once: what every
 * probe of ZopfliBlockSplitLZ77 evaluates twice (blocksplitter.c:103-135: EstimateCost, SplitCost).  The sizes are
 * integers; they come back as doubles as the reference returns them, equal to the host's
 * ZopfliCalculateBlockSizeAutoType value for value (tests/test_gpu_parity.py::test_block_costs).
 *
 * A zmx_cost_stores object is a set of symbol sequences ("stores", lz77.h:44-62) resident on the device with their
 * sampled prefix histograms (lz77.c:98-149's ll_counts / d_counts, here every 1024 symbols):
 *   zmx_cost_stores_create       sequence s = the concatenation of the device stores (block[p], slot[p], nsym[p]) of
 *                                `tables` for p in [piece_first[s], piece_first[s + 1]) — the greedy store of a master
 *                                block (blocksplitter.c:296), or the optimal parses of its blocks joined
 *                                (deflate.c:866); nothing crosses the bus
 *   zmx_cost_stores_create_host  the same from host arrays (lz77.h:44-49 convention)
 *   zmx_block_costs              cost[i] = ZopfliCalculateBlockSizeAutoType(sequence ranges[3 i], lstart = ranges[3 i + 1],
 *                                lend = ranges[3 i + 2])
 * A sequence holds fewer than 2^22 symbols (a master block has at most 1 000 000). */
typedef struct zmx_cost_stores zmx_cost_stores;
int zmx_cost_stores_create(zmx_ctx* ctx, zmx_tables* tables, size_t nstores, const size_t* piece_first,
                           const size_t* block, const int32_t* slot, const size_t* nsym, zmx_cost_stores** out);
int zmx_cost_stores_create_host(zmx_ctx* ctx, size_t nstores, const uint16_t* const* litlens,
                                const uint16_t* const* dists, const size_t* nsym, zmx_cost_stores** out);
void zmx_cost_stores_free(zmx_ctx* ctx, zmx_cost_stores* stores);
int zmx_block_costs(zmx_ctx* ctx, zmx_cost_stores* stores, size_t n, const uint32_t* ranges, double* cost);
/* ZopfliLZ77GetByteRange (lz77.c:160-166) from the start of a sequence: bytes[i] = the input bytes that symbols
 * [0, pairs[2 i + 1]) of sequence pairs[2 i] stand for — where a split point lies (blocksplitter.c:303-314). */
int zmx_cost_positions(zmx_ctx* ctx, zmx_cost_stores* stores, size_t n, const uint32_t* pairs, uint64_t* bytes);

/* -------- whole-stream entry points on a resident input (bench, multi-GPU) */

/* ZopfliDeflate (deflate.c:908) of bytes [instart, inend) of the resident input,
 * cut into 1 000 000-byte master blocks counted from instart (util.h:60), bytes
 * before instart serving as dictionary — i.e. the consecutive ZopfliDeflatePart
 * calls of deflate.c:916-923 for that range.  The result is serialised as
 * position-independent chunks (compressed blocks as bit strings, stored blocks
 * as raw bytes) so that ranges compressed on different GPUs can be joined by
 * zmx_chunks_merge.  `final` marks the last block of the range as BFINAL.
 * *blob is malloc'ed; caller frees. */
int zmx_deflate_range(zmx_ctx* ctx, const ZopfliOptions* options, size_t instart, size_t inend,
                      int final, unsigned char** blob, size_t* blobsize);

/* Concatenates chunk blobs (in stream order) into a deflate bit stream appended
 * to (*out,*outsize) at bit pointer *bp, exactly as the consecutive
 * ZopfliDeflatePart calls would have produced it. */
int zmx_chunks_merge(const unsigned char* const* blobs, const size_t* blobsizes, size_t nblobs,
                     unsigned char* bp, unsigned char** out, size_t* outsize);

/* -------- one process per GPU: the gather of the ranks' blobs over RCCL (xGMI inside a node)
 *
 * ZopfliDeflate's master blocks are independent (deflate.c:916-923): rank r of `world` runs
 * zmx_deflate_range on its contiguous range of master blocks, the blobs are gathered to rank 0 and
 * merged there with zmx_chunks_merge.  librccl is loaded with dlopen on first use.  (Inside ONE
 * process the Zopfli* entry points of part 1 already deal the master blocks over every visible
 * device: ZOPFLI_AMD_DEVICES.) */
typedef struct zmx_dist zmx_dist;

/* Rank 0: a fresh ncclUniqueId (128 bytes) that the launcher hands to every rank (file, environment,
 * MPI, torch.distributed ...). */
int zmx_dist_unique_id(unsigned char* id128);
/* Collective: the communicator of `world` ranks, this process being `rank` on ctx's device. */
int zmx_dist_init(zmx_ctx* ctx, int rank, int world, const unsigned char* id128, zmx_dist** dist);
void zmx_dist_destroy(zmx_dist* dist);
/* The number of ranks RCCL itself reports for the communicator (ncclCommCount), or -1. */
int zmx_dist_comm_count(zmx_dist* dist);
/* Collective: every rank contributes `size` bytes.  On rank 0 *gathered is a malloc'ed buffer holding
 * the blobs of rank 0, 1, ... back to back and sizes[r] their lengths (sizes has `world` entries);
 * elsewhere *gathered = NULL and sizes is not written. */
int zmx_dist_gather(zmx_dist* dist, const unsigned char* blob, size_t size, unsigned char** gathered,
                    size_t* sizes);

/* Cost-aware dealing of master blocks over GPUs (host only, no device call; zopfli_amd/csrc/host/deal.cc).  Master blocks
 * are independent (deflate.c:916-923) but not equally expensive: one of long runs of equal bytes costs several times one
 * of text.  cost[b] = the estimated cost of master block b of in[0, insize) relative to a master block of text, a function
 * of the bytes alone (64-byte probes every 1024) so that every rank computes the same numbers; returns the number of master
 * blocks (ncost must be at least that).  zmx_deal_master_blocks: contiguous ranges balanced by cost,
 * first[s] .. first[s + 1] for shard s (first has shards + 1 entries).  ZopfliCompress / ZopfliDeflate deal a call over
 * their contexts this way; one-process-per-GPU launchers (zopfli_amd/sharding.py, bench.py) call these two. */
int zmx_master_block_costs(const unsigned char* in, size_t insize, double* cost, size_t ncost);
int zmx_deal_master_blocks(const double* cost, size_t nblocks, size_t shards, size_t* first);

/* (The kernel, match and task statistics below are sums over the last Zopfli* / zmx_deflate_range call of the CALLING
 * THREAD — its shard threads' numbers included — reset when the call starts: concurrent callers read their own.)
 *
 * Timing breakdown of the last Zopfli* / zmx_deflate_range call on this
 * thread: seconds spent in [0] match tables [1] greedy [2] squeeze runs
 * [3] host cost model [4] block split [5] encode [6] the chain kernels of the squeeze runs (k_dp5_spec,
 * k_dpcheck, k_dpscan, k_dp4_fix: GetBestLengths; HIP events on the launch stream) [7] squeeze runs launched.
 * For bench.py's roofline object. */
int zmx_last_timing(double* out8);

/* Kernel-only seconds (HIP events) of the squeeze runs since the last Zopfli* /
 * zmx_deflate_range call started: [0] k_wtab + k_badscan (the run's weight tables) [1] the chain kernels
 * [2] k_trace_* (exits + link + emit) [3] squeeze runs launched. */
int zmx_last_kernel_timing(double* out4);
/* The phase times of the squeeze runs (above; zmx_last_timing's dp_kernel) cost four event records and three readings a
 * run, a third of a run's runtime calls: they are taken only when asked for — this call, ZOPFLI_AMD_KERNEL_TIMING=1 or
 * ZOPFLI_AMD_PROF.  (The match-table times of zmx_last_match_timing are always taken: once per table build.) */
void zmx_set_kernel_timing(int on);

/* Host tail of the last call on this thread, seconds: [0] best LZ77 stores device -> host
 * [1] chunk serialisation (zmx_deflate_range). */
int zmx_last_host_timing(double* out2);

/* Match-table builds since the last Zopfli* / zmx_deflate_range call started (HIP events):
 * [0] seconds in the match kernel (k_match2; ZOPFLI_AMD_MATCH = 3 / 4: k_match3 / k_match4) [1] seconds in k_same +
 * k_chain (or k_bucket) [2] table builds [3] positions whose record
 * the match kernel computed (the others were copied from the parent tables). */
int zmx_last_match_timing(double* out4);
/* The skip-walk (k_match5) of the table builds since the last Zopfli* call started / zmx_deflate_range: out3 = entries in
 * flight summed over lanes and wave iterations (a position's walk touches ~10 entries whatever its chains' length), wave
 * iterations, positions the kernel walked (blocks k_hits sent to it).  Zero when every block took k_match2. */
int zmx_last_match_walk(double* out3);

/* Several contexts on one device (the Zopfli* entry points keep up to three per device).  The budgets are the DEVICE's,
 * not a context's: what the pools of all its contexts keep cached between batches counts against one third of its memory
 * together, what one batch's DP edges may take is a third; a table build that cannot allocate (the other contexts are
 * busy) returns -2, "come back with fewer blocks", like one beyond that budget.  zmx_ctx_trim_cache gives an IDLE
 * context's cached arrays back to the device; the hook (one per process, may be null) is called with the device index when
 * an allocation still fails after the failing context dropped its own cache — the owner of the contexts trims the idle
 * ones there — and the allocation is tried once more.  zmx_ctx_set_share tells a context how many share its device. */
typedef void (*zmx_oom_hook_t)(int device);
void zmx_set_oom_hook(zmx_oom_hook_t hook);
int zmx_ctx_set_share(zmx_ctx* ctx, unsigned contexts_on_device);
int zmx_ctx_trim_cache(zmx_ctx* ctx);
/* Which streams the context's next calls run on: 0 = the pair it was created with, +1 / -1 = a pair of the highest /
 * lowest priority the device has.  Only while the context is idle.  The Zopfli* entry points give the contexts a call with
 * block splitting is dealt over three different priorities (ZOPFLI_AMD_STREAM_PRIO=0: never). */
int zmx_ctx_set_priority(zmx_ctx* ctx, int level);

/* Which match-table kernel the table builds that START after this call use (the ZOPFLI_AMD_MATCH environment
 * variable sets the initial choice): 0 = per block, k_match5 where k_hits estimates long chains, k_match2 elsewhere
 * (the default), 2 = k_chain + k_match2, 3 / 4 = k_bucket + k_match3 / k_match4, 5 = k_chain + k_levels + k_rank2 +
 * k_match5 (the exact skip-walk) for every block.  All produce the same records; an A/B and test hook. */
int zmx_set_match_kernel(int kernel);

/* The chain's tasks (GetBestLengths cut into verified stretches, zmx_dp4.h) since the last Zopfli* /
 * zmx_deflate_range call started: [0] tasks [1] accepted as computed [2] re-run because the entry
 * state differed [3] because the guessed level left the binade [4] because a weight could tie
 * [5] positions re-run serially [6] re-run because the entry values differed by more than a shift
 * [7] block positions of all squeeze runs. */
int zmx_last_seg_stats(double* out8);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif

#ifdef __cplusplus
}
#endif

#endif /* ZOPFLI_AMD_H_ */

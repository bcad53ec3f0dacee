// Deflate driver: master blocks -> block split -> optimal parse (device) ->
// second split attempt -> block type choice -> bit chunks.
//
// Same decisions as ZopfliDeflatePart / AddLZ77BlockAutoType of the reference
// (deflate.c:811-906, :747-800), reorganised so that every device call is
// batched over all master blocks of a request and every host step is a
// parallel loop over independent master blocks.  Compressed blocks are
// produced as position-independent bit chunks; only stored blocks depend on
// the running bit pointer, so they are materialised at merge time.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "bit_writer.h"
#include "zopfli_amd.h"

namespace zamd {

struct Chunk {
  enum Kind : uint8_t { kBits = 0, kStored = 1 };
  Kind kind = kBits;
  bool final_block = false;     // kStored only: BFINAL of its last piece
  CVec<uint8_t> bits;           // kBits: packed from bit 0 (memory from the library's block cache)
  size_t nbits = 0;
  size_t start = 0, end = 0;    // kStored: raw input range ...
  std::vector<uint8_t> raw;     // ... or, after (de)serialisation, the bytes themselves
  // A deserialised chunk does not copy: it points into the blob it came from (which must outlive it).
  const uint8_t* view = nullptr;   // kBits: the packed bits; kStored: the raw bytes
  size_t view_bytes = 0;

  // ZopfliOptions::verbose: the reference's stderr lines around this block, produced when the chunk
  // is put into the stream (the byte counts it prints depend on the bit position, deflate.c:719-744).
  // log_pre: printed before the block (block split points, iteration lines of the part's blocks);
  // a compressed block then prints "treesize" (dynamic only) and "compressed block size".
  std::string log_pre;
  bool log_block = false;
  int log_btype = 0;
  size_t log_tree_bits = 0, log_unc = 0;

  const uint8_t* BitData() const { return view ? view : bits.data(); }
  size_t BitBytes() const { return view ? view_bytes : bits.size(); }
};

struct Part {
  size_t instart, inend;
  bool final_part;  // last deflate block of this part carries BFINAL
};

// Compresses each part independently (one ZopfliDeflatePart each); chunks come
// out in stream order.  Positions refer to the input resident in `ctx`.
int DeflateParts(zmx_ctx* ctx, const ZopfliOptions& options, int btype, const std::vector<Part>& parts,
                 std::vector<Chunk>* chunks);

// Appends chunks at (*out, *outsize, *bp), reference conventions (deflate.h:50-53, util.h:135-155);
// `in` is the base of the resident input (stored chunks that still refer to it).
void MergeChunks(const std::vector<Chunk>& chunks, const unsigned char* in, unsigned char* bp,
                 unsigned char** out, size_t* outsize, bool verbose = false);

// Stored chunks are serialised with their raw bytes (taken from `in`).  Returns a malloc'ed blob
// (nullptr when out of memory).
unsigned char* SerializeChunks(const std::vector<Chunk>& chunks, const unsigned char* in, size_t* size);
// The chunks are views into `blob`.
bool DeserializeChunks(const unsigned char* blob, size_t size, std::vector<Chunk>* chunks);

// Makes room for `n` more bytes after *outsize (capacity rule of ZOPFLI_APPEND_DATA); the new
// bytes are zeroed unless zero = false.  *outsize is not changed.
void ReserveOutput(size_t n, unsigned char** out, size_t* outsize, bool zero = true);

// Appends `n` bytes to a reference-style growable array (util.h:135-155).
void AppendToOutput(const uint8_t* data, size_t n, unsigned char** out, size_t* outsize);

}  // namespace zamd

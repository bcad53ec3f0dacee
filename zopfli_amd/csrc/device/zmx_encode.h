// The deflate bit writer on the device (SURVEY f-2): AddLZ77Data + the end symbol of a compressed block
// (deflate.c:297-333, :735-737) for whole LZ77 stores that are already in HBM, so that neither the symbols
// (4 bytes each) nor a host loop over them are needed to produce the stream — only the bits come down.
// Included only by zmx_hip.hip, after zmx_kernels.h.
//
// A job = one block of a table set, one of its two stores, the block's Huffman codes (from the host, which
// builds the trees from the histogram it already has: tree.c / katajainen.c stay on the host with libm) and
// the bit of the block's output the symbols start at (the host puts the 3 header bits and the tree there).
// Three kernels over tiles of ENC_TILE symbols:
//   k_enc_len   bits of every tile (sum of code length + extra bits; the last tile adds the end symbol)
//   k_enc_scan  one wave per job: exclusive scan of its tiles' bits -> where each tile starts; the total is
//               checked against what the host computed from the histogram
//   k_enc_emit  a tile's symbols are ORed into an LDS bit buffer at their scanned places (a symbol is at most
//               48 bits: three 32-bit LDS atomics), the buffer is shifted to the tile's bit offset and written
//               out: whole words stored, the two words shared with the neighbours ORed atomically into the
//               zeroed output.
#pragma once

#define ENC_THREADS 256u
#define ENC_PER 8u
#define ENC_TILE (ENC_THREADS * ENC_PER)            // symbols per tile
#define ENC_WORDS (ENC_TILE * 48u / 32u + 8u)       // LDS words of a tile's bits (48 bits per symbol at most)

struct EncJob {
  u64 sym_off;      // first symbol of the store in store[slot] (block's pos_off + store_begin)
  u64 out_word;     // first 32-bit word of the block's output
  u64 nbits;        // symbols + end symbol, as the host expects them
  u32 nsym;
  u32 slot;
  u32 bit_start;    // the symbols start at this bit of the block's output
  u32 tile0;        // first tile of the job in the tile arrays
  u32 code;         // index of the job's code table (codes + code * 320)
  u32 pad;
};

struct EncParams {
  const EncJob* jobs;
  const u32* tile_job;      // [tiles] job of each tile
  const u32* codes;         // [jobs][320] reversed code | length << 16: 288 litlen symbols, then 32 dist symbols
  const u32* store[2];
  u32* tile_bits;           // [tiles]
  u64* tile_off;            // [tiles] bit offset of the tile in its job's output (bit_start included)
  u32* out;                 // zeroed
  u32* flags;               // bit 2: a job's bits are not what the host expected
  u32 njobs;
};

// bits of one LZ77 symbol: value (LSB first) and length
__device__ __forceinline__ void enc_symbol(u32 sym, const u32* s_codes, u64& v, u32& n) {
  const u32 litlen = sym & 0xffffu, dist = sym >> 16;
  if (dist == 0) {
    const u32 c = s_codes[litlen];
    v = c & 0xffffu;
    n = c >> 16;
    return;
  }
  const u32 cl = s_codes[dev_length_symbol(litlen)];
  const u32 le = (u32)dev_length_extra_bits(litlen);
  const u32 cd = s_codes[288 + dev_dist_symbol(dist)];
  const u32 de = (u32)dev_dist_extra_bits(dist);
  const u32 lv = le ? (litlen - 3u) & ((1u << le) - 1u) : 0u;        // symbols.h:161 (258: no extra bits)
  const u32 dv = de ? (dist - 1u) & ((1u << de) - 1u) : 0u;          // symbols.h:61
  u32 p = cl >> 16;
  v = cl & 0xffffu;
  v |= (u64)lv << p; p += le;
  v |= (u64)(cd & 0xffffu) << p; p += cd >> 16;
  v |= (u64)dv << p; p += de;
  n = p;
}

__global__ __launch_bounds__(ENC_THREADS) void k_enc_len(EncParams P) {
  __shared__ u32 s_codes[320];
  __shared__ u32 s_sum[ENC_THREADS / 64];
  const u32 tile = blockIdx.x;
  const EncJob J = P.jobs[P.tile_job[tile]];
  for (u32 i = threadIdx.x; i < 320; i += ENC_THREADS) s_codes[i] = P.codes[(u64)J.code * 320 + i];
  __syncthreads();
  const u32 first = (tile - J.tile0) * ENC_TILE;
  const u32* st = P.store[J.slot] + J.sym_off;
  u32 bits = 0;
#pragma unroll
  for (u32 k = 0; k < ENC_PER; ++k) {
    const u32 i = first + threadIdx.x * ENC_PER + k;
    if (i < J.nsym) {
      u64 v; u32 n;
      enc_symbol(st[i], s_codes, v, n);
      bits += n;
    }
  }
  bits = wave_scan_add(bits);
  if ((threadIdx.x & 63) == 63) s_sum[threadIdx.x >> 6] = bits;
  __syncthreads();
  if (threadIdx.x == 0) {
    u32 t = 0;
    for (u32 w = 0; w < ENC_THREADS / 64; ++w) t += s_sum[w];
    if (tile - J.tile0 == J.nsym / ENC_TILE) t += s_codes[256] >> 16;     // the end symbol closes the last tile
    P.tile_bits[tile] = t;
  }
}

__global__ __launch_bounds__(64) void k_enc_scan(EncParams P) {
  const EncJob J = P.jobs[blockIdx.x];
  const u32 ntiles = J.nsym / ENC_TILE + 1;        // (a job always has a tile for the end symbol)
  u64 run = J.bit_start;
  for (u32 t0 = 0; t0 < ntiles; t0 += 64) {
    const u32 t = t0 + threadIdx.x;
    const u32 b = t < ntiles ? P.tile_bits[J.tile0 + t] : 0u;
    const u32 incl = wave_scan_add(b);
    if (t < ntiles) P.tile_off[J.tile0 + t] = run + incl - b;
    run += (u32)__builtin_amdgcn_readlane((int)incl, 63);
  }
  if (threadIdx.x == 0 && run - J.bit_start != J.nbits) atomicOr(P.flags, 4u);
}

__global__ __launch_bounds__(ENC_THREADS) void k_enc_emit(EncParams P) {
  __shared__ u32 s_codes[320];
  __shared__ u32 s_sum[ENC_THREADS / 64];
  __shared__ u32 s_bits[ENC_WORDS];
  const u32 tile = blockIdx.x;
  const EncJob J = P.jobs[P.tile_job[tile]];
  for (u32 i = threadIdx.x; i < 320; i += ENC_THREADS) s_codes[i] = P.codes[(u64)J.code * 320 + i];
  for (u32 i = threadIdx.x; i < ENC_WORDS; i += ENC_THREADS) s_bits[i] = 0;
  __syncthreads();
  const u32 first = (tile - J.tile0) * ENC_TILE;
  const u32* st = P.store[J.slot] + J.sym_off;
  u64 v[ENC_PER];
  u32 n[ENC_PER];
  u32 bits = 0;
#pragma unroll
  for (u32 k = 0; k < ENC_PER; ++k) {
    const u32 i = first + threadIdx.x * ENC_PER + k;
    v[k] = 0; n[k] = 0;
    if (i < J.nsym) enc_symbol(st[i], s_codes, v[k], n[k]);
    bits += n[k];
  }
  const u32 incl = wave_scan_add(bits);
  if ((threadIdx.x & 63) == 63) s_sum[threadIdx.x >> 6] = incl;
  __syncthreads();
  u32 pos = incl - bits, total = 0;
  for (u32 w = 0; w < ENC_THREADS / 64; ++w) {
    const u32 s = s_sum[w];
    if (w < (threadIdx.x >> 6)) pos += s;
    total += s;
  }
  // the thread's symbols, one after the other, into the tile's bit buffer
#pragma unroll
  for (u32 k = 0; k < ENC_PER; ++k) {
    if (n[k]) {
      const u32 w = pos >> 5, sh = pos & 31u;
      const u64 lo = v[k] << sh;
      atomicOr(&s_bits[w], (u32)lo);
      if (sh + n[k] > 32) atomicOr(&s_bits[w + 1], (u32)(lo >> 32));
      if (sh + n[k] > 64) atomicOr(&s_bits[w + 2], (u32)(v[k] >> (64u - sh)));
      pos += n[k];
    }
  }
  const bool last_tile = tile - J.tile0 == J.nsym / ENC_TILE;
  if (last_tile && threadIdx.x == 0) {     // the end symbol, after the tile's last symbol
    const u32 c = s_codes[256];
    const u32 w = total >> 5, sh = total & 31u;
    const u64 lo = (u64)(c & 0xffffu) << sh;
    atomicOr(&s_bits[w], (u32)lo);
    if (sh + (c >> 16) > 32) atomicOr(&s_bits[w + 1], (u32)(lo >> 32));
  }
  if (last_tile) total += s_codes[256] >> 16;
  __syncthreads();
  // out: the buffer shifted to the tile's bit offset
  const u64 g = P.tile_off[tile];
  u32* out = P.out + J.out_word + (g >> 5);
  const u32 sh = (u32)(g & 31u);
  const u32 nw = (total + sh + 31u) >> 5;
  for (u32 w = threadIdx.x; w < nw; w += ENC_THREADS) {
    const u32 a = s_bits[w], b = w ? s_bits[w - 1] : 0u;
    const u32 x = sh ? (a << sh) | (b >> (32u - sh)) : a;
    if (w == 0 || w + 1 == nw) { if (x) atomicOr(&out[w], x); }
    else out[w] = x;
  }
}

// ---------------------------------------------------------------------------------------------
// ZopfliVerifyLenDist (lz77.c:270-295; the reference asserts it for every symbol it stores, lz77.c:115) as a
// debug pass over whole stores (ZOPFLI_AMD_VERIFY): one workgroup per job walks its store 256 symbols at a
// time, a scan of the symbols' lengths gives their positions, every thread compares its match byte by byte.
// ---------------------------------------------------------------------------------------------
struct VerifyJob {
  u64 sym_off;      // first symbol in store[slot]
  u64 instart;      // first byte of the block
  u64 inend;
  u32 nsym;
  u32 slot;
};
struct VerifyParams {
  const VerifyJob* jobs;
  const u8* in;
  const u32* store[2];
  u32* bad;         // [jobs][2]: 1 + index of a symbol that fails, and what was wrong (1: length / distance out of
                    // range, 2: bytes differ, 3: the symbols do not add up to the block)
};

__global__ __launch_bounds__(256) void k_verify(VerifyParams P) {
  __shared__ u32 s_sum[4];
  const VerifyJob J = P.jobs[blockIdx.x];
  const u32* st = P.store[J.slot] + J.sym_off;
  u64 base = J.instart;
  for (u32 i0 = 0; i0 < J.nsym; i0 += 256) {
    const u32 i = i0 + threadIdx.x;
    u32 litlen = 0, dist = 0, len = 0;
    if (i < J.nsym) {
      const u32 s = st[i];
      litlen = s & 0xffffu;
      dist = s >> 16;
      len = dist ? litlen : 1u;
    }
    const u32 incl = wave_scan_add(len);
    if ((threadIdx.x & 63) == 63) s_sum[threadIdx.x >> 6] = incl;
    __syncthreads();
    u64 pos = base + incl - len;
    u32 tot = 0;
    for (u32 w = 0; w < 4; ++w) {
      if (w < (threadIdx.x >> 6)) pos += s_sum[w];
      tot += s_sum[w];
    }
    if (i < J.nsym) {
      u32 why = 0;
      if (dist == 0) {
        if (litlen > 255 || pos >= J.inend || P.in[pos] != litlen) why = litlen > 255 ? 1u : 2u;
      } else if (litlen < 3 || litlen > 258 || dist > 32768 || dist > pos || pos + litlen > J.inend) {
        why = 1;
      } else {
        for (u32 k = 0; k < litlen; ++k) {
          if (P.in[pos + k] != P.in[pos + k - dist]) { why = 2; break; }
        }
      }
      // (symbol and reason in ONE word, so that the reason reported is the reported symbol's: (i + 1) << 2 | why)
      if (why) atomicMax(&P.bad[2 * blockIdx.x], ((i + 1) << 2) | why);
    }
    base += tot;
    __syncthreads();
  }
  if (threadIdx.x == 0 && base != J.inend) atomicMax(&P.bad[2 * blockIdx.x], ((J.nsym + 1) << 2) | 3u);
}

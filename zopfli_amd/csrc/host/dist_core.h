// The gather of the ranks' chunk blobs, without the wires: sizes first (an all-gather of one u64 per rank), a
// second all-gather of every rank's status after its local preparations (staging buffers may fail to allocate on
// ONE rank: everybody learns of it and fails together instead of the others waiting in a collective for ever),
// then the payload — every rank but 0 sends its blob, rank 0 receives them behind its own into one malloc'ed
// buffer.  dist.cc plugs RCCL in (device staging buffers, ncclAllGather, grouped ncclSend / ncclRecv over xGMI);
// the CPU tests plug a socket transport in and run this very code with two ranks (tests/hostlib).
#pragma once
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace zamd {

struct GatherTransport {
  void* self = nullptr;
  int rank = 0, world = 1;
  // collective: all[r] = rank r's `mine`
  int (*all_gather_u64)(void* self, uint64_t mine, uint64_t* all) = nullptr;
  // local: get ready to send `size` bytes (rank != 0) or to receive `total - size` (rank 0); 0 = ok
  int (*prepare)(void* self, size_t size, size_t total) = nullptr;
  // collective: rank != 0 sends blob[0, size); rank 0 receives rank r's all[r] bytes at host + off[r], r = 1 ..
  int (*exchange)(void* self, const unsigned char* blob, size_t size, const uint64_t* all, const size_t* off,
                  unsigned char* host) = nullptr;
  std::string* error = nullptr;   // where a failing call leaves its message
};

// On rank 0 *gathered is a malloc'ed buffer holding the blobs of rank 0, 1, ... back to back and sizes[r] their
// lengths; elsewhere *gathered = nullptr and sizes is not written.  Non-zero: *error says why; every rank returns
// non-zero when any rank's preparation failed.
inline int GatherBlobs(const GatherTransport& t, const unsigned char* blob, size_t size, unsigned char** gathered,
                       size_t* sizes) {
  const size_t world = static_cast<size_t>(t.world);
  *gathered = nullptr;
  std::vector<uint64_t> all(world, 0);
  if (t.all_gather_u64(t.self, size, all.data()) != 0) return -1;
  std::vector<size_t> off(world + 1, 0);
  size_t total = 0;
  for (size_t r = 0; r < world; ++r) {
    off[r] = total;
    total += static_cast<size_t>(all[r]);
  }
  off[world] = total;
  unsigned char* host = nullptr;
  uint64_t ok = t.prepare(t.self, size, total) == 0 ? 1 : 0;
  if (ok && t.rank == 0) {
    host = static_cast<unsigned char*>(std::malloc(total ? total : 1));
    if (!host) {
      ok = 0;
      if (t.error) *t.error = "zmx_dist_gather: out of memory";
    }
  }
  std::vector<uint64_t> oks(world, 0);
  if (t.all_gather_u64(t.self, ok, oks.data()) != 0) {
    std::free(host);
    return -1;
  }
  for (size_t r = 0; r < world; ++r) {
    if (!oks[r]) {
      std::free(host);
      if (ok && t.error) *t.error = "zmx_dist_gather: rank " + std::to_string(r) + " could not prepare its buffers";
      return -1;
    }
  }
  if (t.rank == 0 && size) std::memcpy(host, blob, size);   // rank 0's own blob does not travel
  if (t.exchange(t.self, blob, size, all.data(), off.data(), host) != 0) {
    std::free(host);
    return -1;
  }
  if (t.rank == 0) {
    *gathered = host;
    for (size_t r = 0; r < world; ++r) sizes[r] = static_cast<size_t>(all[r]);
  }
  return 0;
}

}  // namespace zamd

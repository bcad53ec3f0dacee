// One process per GPU: the gather of the ranks' chunk blobs to rank 0 over RCCL (xGMI inside a node).
//
// ZopfliDeflate cuts its input into master blocks that are compressed independently
// (deflate.c:916-923), so a stream shards by master block with no data-path collective: rank r runs
// zmx_deflate_range on its contiguous range, and the only exchange is this variable-size gather of
// the serialised bit chunks (~0.3 bytes per input byte) before zmx_chunks_merge on rank 0.  Sizes
// first (ncclAllGather of one u64 per rank), then the payload as grouped ncclSend / ncclRecv.
//
// librccl is loaded at run time (dlopen), so a single-GPU user of libzopfli_amd.so needs no RCCL.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "zopfli_amd.h"

extern "C" int zmx_internal_device(zmx_ctx* ctx);
extern "C" void zmx_internal_set_error(const char* msg);

namespace {

struct Rccl {
  void* handle = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  std::string error;
};

Rccl* LoadRccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      r.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (r.handle) break;
    }
    if (!r.handle) {
      r.error = std::string("cannot load librccl: ") + dlerror();
      return;
    }
#define ZMX_SYM(field, name)                                             \
  r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.handle, name)); \
  if (!r.field && r.error.empty()) r.error = std::string("librccl lacks ") + name;
    ZMX_SYM(GetUniqueId, "ncclGetUniqueId")
    ZMX_SYM(CommInitRank, "ncclCommInitRank")
    ZMX_SYM(CommDestroy, "ncclCommDestroy")
    ZMX_SYM(AllGather, "ncclAllGather")
    ZMX_SYM(GroupStart, "ncclGroupStart")
    ZMX_SYM(GroupEnd, "ncclGroupEnd")
    ZMX_SYM(Send, "ncclSend")
    ZMX_SYM(Recv, "ncclRecv")
    ZMX_SYM(GetErrorString, "ncclGetErrorString")
#undef ZMX_SYM
  });
  return &r;
}

int Fail(const std::string& m) {
  zmx_internal_set_error(m.c_str());
  return -1;
}

}  // namespace

struct zmx_dist {
  Rccl* rccl = nullptr;
  ncclComm_t comm = nullptr;
  hipStream_t stream = nullptr;
  int device = 0, rank = 0, world = 1;
  unsigned char* d_send = nullptr;
  size_t send_cap = 0;
  unsigned char* d_recv = nullptr;   // rank 0: all payloads back to back
  size_t recv_cap = 0;
  uint64_t* d_sizes = nullptr;       // [1 + world]: own size, then everybody's
};

#define RCCLCHK(d, expr)                                                                         \
  do {                                                                                           \
    ncclResult_t r_ = (expr);                                                                    \
    if (r_ != ncclSuccess) return Fail(std::string(#expr) + ": " + (d)->rccl->GetErrorString(r_)); \
  } while (0)
#define HIPCHK2(expr)                                                                \
  do {                                                                               \
    hipError_t e_ = (expr);                                                          \
    if (e_ != hipSuccess) return Fail(std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)

extern "C" {

int zmx_dist_unique_id(unsigned char* id128) {
  Rccl* r = LoadRccl();
  if (!r->error.empty()) return Fail(r->error);
  ncclUniqueId id;
  const ncclResult_t rc = r->GetUniqueId(&id);
  if (rc != ncclSuccess) return Fail(std::string("ncclGetUniqueId: ") + r->GetErrorString(rc));
  static_assert(sizeof(id) == 128, "ncclUniqueId is 128 bytes");
  std::memcpy(id128, &id, sizeof(id));
  return 0;
}

int zmx_dist_init(zmx_ctx* ctx, int rank, int world, const unsigned char* id128, zmx_dist** out) {
  Rccl* r = LoadRccl();
  if (!r->error.empty()) return Fail(r->error);
  if (world < 1 || rank < 0 || rank >= world) return Fail("zmx_dist_init: bad rank / world");
  zmx_dist* d = new zmx_dist();
  d->rccl = r;
  d->rank = rank;
  d->world = world;
  d->device = zmx_internal_device(ctx);
  int old = -1;
  (void)hipGetDevice(&old);
  HIPCHK2(hipSetDevice(d->device));
  ncclUniqueId id;
  std::memcpy(&id, id128, sizeof(id));
  RCCLCHK(d, r->CommInitRank(&d->comm, world, id, rank));
  HIPCHK2(hipStreamCreate(&d->stream));
  HIPCHK2(hipMalloc(reinterpret_cast<void**>(&d->d_sizes), sizeof(uint64_t) * (1 + static_cast<size_t>(world))));
  if (old >= 0) (void)hipSetDevice(old);
  *out = d;
  return 0;
}

void zmx_dist_destroy(zmx_dist* d) {
  if (!d) return;
  int old = -1;
  (void)hipGetDevice(&old);
  (void)hipSetDevice(d->device);
  if (d->comm) (void)d->rccl->CommDestroy(d->comm);
  (void)hipFree(d->d_send);
  (void)hipFree(d->d_recv);
  (void)hipFree(d->d_sizes);
  if (d->stream) (void)hipStreamDestroy(d->stream);
  if (old >= 0) (void)hipSetDevice(old);
  delete d;
}

int zmx_dist_gather(zmx_dist* d, const unsigned char* blob, size_t size, unsigned char** gathered,
                    size_t* sizes) {
  int old = -1;
  (void)hipGetDevice(&old);
  HIPCHK2(hipSetDevice(d->device));
  const size_t world = static_cast<size_t>(d->world);
  // ---- sizes
  const uint64_t mine = size;
  HIPCHK2(hipMemcpyAsync(d->d_sizes, &mine, sizeof(mine), hipMemcpyHostToDevice, d->stream));
  RCCLCHK(d, d->rccl->AllGather(d->d_sizes, d->d_sizes + 1, 1, ncclUint64, d->comm, d->stream));
  std::vector<uint64_t> all(world);
  HIPCHK2(hipMemcpyAsync(all.data(), d->d_sizes + 1, world * sizeof(uint64_t), hipMemcpyDeviceToHost, d->stream));
  HIPCHK2(hipStreamSynchronize(d->stream));
  // ---- payload
  if (size > d->send_cap) {
    (void)hipFree(d->d_send);
    d->d_send = nullptr;
    d->send_cap = 0;
    HIPCHK2(hipMalloc(reinterpret_cast<void**>(&d->d_send), size + size / 4));
    d->send_cap = size + size / 4;
  }
  if (size) HIPCHK2(hipMemcpyAsync(d->d_send, blob, size, hipMemcpyHostToDevice, d->stream));
  size_t total = 0;
  std::vector<size_t> off(world + 1, 0);
  for (size_t r = 0; r < world; ++r) {
    off[r] = total;
    total += all[r];
  }
  off[world] = total;
  if (d->rank == 0 && total > d->recv_cap) {
    (void)hipFree(d->d_recv);
    d->d_recv = nullptr;
    d->recv_cap = 0;
    HIPCHK2(hipMalloc(reinterpret_cast<void**>(&d->d_recv), total + total / 4));
    d->recv_cap = total + total / 4;
  }
  RCCLCHK(d, d->rccl->GroupStart());
  if (d->rank != 0) {
    if (size) RCCLCHK(d, d->rccl->Send(d->d_send, size, ncclUint8, 0, d->comm, d->stream));
  } else {
    for (size_t r = 1; r < world; ++r) {
      if (all[r]) RCCLCHK(d, d->rccl->Recv(d->d_recv + off[r], all[r], ncclUint8, static_cast<int>(r), d->comm, d->stream));
    }
  }
  RCCLCHK(d, d->rccl->GroupEnd());
  if (d->rank == 0) {
    unsigned char* host = static_cast<unsigned char*>(std::malloc(total ? total : 1));
    if (!host) return Fail("zmx_dist_gather: out of memory");
    if (size) std::memcpy(host, blob, size);   // rank 0's own blob does not travel
    if (total > size) HIPCHK2(hipMemcpyAsync(host + off[1], d->d_recv + off[1], total - size, hipMemcpyDeviceToHost, d->stream));
    HIPCHK2(hipStreamSynchronize(d->stream));
    *gathered = host;
    for (size_t r = 0; r < world; ++r) sizes[r] = all[r];
  } else {
    HIPCHK2(hipStreamSynchronize(d->stream));
    *gathered = nullptr;
  }
  if (old >= 0) (void)hipSetDevice(old);
  return 0;
}

}  // extern "C"

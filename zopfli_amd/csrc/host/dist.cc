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

#include "dist_core.h"
#include "zopfli_amd.h"

extern "C" int zmx_internal_device(zmx_ctx* ctx);
extern "C" void zmx_internal_set_error(const char* msg);

namespace {

struct Rccl {
  void* handle = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclCommCount) CommCount = nullptr;
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
    ZMX_SYM(CommCount, "ncclCommCount")
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

namespace {
// the caller's current HIP device, put back whichever way a function is left
struct DeviceScope {
  int old = -1;
  DeviceScope() { if (hipGetDevice(&old) != hipSuccess) old = -1; }
  ~DeviceScope() { if (old >= 0) (void)hipSetDevice(old); }
};
}  // namespace

void zmx_dist_destroy(zmx_dist* d);

int zmx_dist_init(zmx_ctx* ctx, int rank, int world, const unsigned char* id128, zmx_dist** out) {
  Rccl* r = LoadRccl();
  if (!r->error.empty()) return Fail(r->error);
  if (world < 1 || rank < 0 || rank >= world) return Fail("zmx_dist_init: bad rank / world");
  DeviceScope scope;
  zmx_dist* d = new zmx_dist();
  d->rccl = r;
  d->rank = rank;
  d->world = world;
  d->device = zmx_internal_device(ctx);
  // (every failure below frees what exists so far: zmx_dist_destroy copes with a half-built object)
  const auto fail = [&](const std::string& m) { zmx_dist_destroy(d); return Fail(m); };
  hipError_t e = hipSetDevice(d->device);
  if (e != hipSuccess) return fail(std::string("hipSetDevice: ") + hipGetErrorString(e));
  ncclUniqueId id;
  std::memcpy(&id, id128, sizeof(id));
  const ncclResult_t rc = r->CommInitRank(&d->comm, world, id, rank);
  if (rc != ncclSuccess) { d->comm = nullptr; return fail(std::string("ncclCommInitRank: ") + r->GetErrorString(rc)); }
  e = hipStreamCreate(&d->stream);
  if (e != hipSuccess) { d->stream = nullptr; return fail(std::string("hipStreamCreate: ") + hipGetErrorString(e)); }
  e = hipMalloc(reinterpret_cast<void**>(&d->d_sizes), sizeof(uint64_t) * (1 + static_cast<size_t>(world)));
  if (e != hipSuccess) { d->d_sizes = nullptr; return fail(std::string("hipMalloc: ") + hipGetErrorString(e)); }
  *out = d;
  return 0;
}

void zmx_dist_destroy(zmx_dist* d) {
  if (!d) return;
  DeviceScope scope;
  (void)hipSetDevice(d->device);
  if (d->comm) (void)d->rccl->CommDestroy(d->comm);
  (void)hipFree(d->d_send);
  (void)hipFree(d->d_recv);
  (void)hipFree(d->d_sizes);
  if (d->stream) (void)hipStreamDestroy(d->stream);
  delete d;
}

int zmx_dist_comm_count(zmx_dist* d) {
  if (!d || !d->comm) return -1;
  int n = -1;
  if (d->rccl->CommCount(d->comm, &n) != ncclSuccess) return -1;
  return n;
}

namespace {
// ---- the RCCL transport of zamd::GatherBlobs (dist_core.h): device staging buffers, the wires are xGMI
int RcclAllGather(void* self, uint64_t mine, uint64_t* all) {
  zmx_dist* d = static_cast<zmx_dist*>(self);
  HIPCHK2(hipMemcpyAsync(d->d_sizes, &mine, sizeof(mine), hipMemcpyHostToDevice, d->stream));
  RCCLCHK(d, d->rccl->AllGather(d->d_sizes, d->d_sizes + 1, 1, ncclUint64, d->comm, d->stream));
  HIPCHK2(hipMemcpyAsync(all, d->d_sizes + 1, static_cast<size_t>(d->world) * sizeof(uint64_t), hipMemcpyDeviceToHost, d->stream));
  HIPCHK2(hipStreamSynchronize(d->stream));
  return 0;
}

int RcclPrepare(void* self, size_t size, size_t total) {
  zmx_dist* d = static_cast<zmx_dist*>(self);
  if (d->rank != 0 && size > d->send_cap) {
    (void)hipFree(d->d_send);
    d->d_send = nullptr;
    d->send_cap = 0;
    HIPCHK2(hipMalloc(reinterpret_cast<void**>(&d->d_send), size + size / 4));
    d->send_cap = size + size / 4;
  }
  if (d->rank == 0 && total - size > d->recv_cap) {
    (void)hipFree(d->d_recv);
    d->d_recv = nullptr;
    d->recv_cap = 0;
    const size_t cap = (total - size) + (total - size) / 4;
    HIPCHK2(hipMalloc(reinterpret_cast<void**>(&d->d_recv), cap));
    d->recv_cap = cap;
  }
  return 0;
}

int RcclExchange(void* self, const unsigned char* blob, size_t size, const uint64_t* all, const size_t* off,
                 unsigned char* host) {
  zmx_dist* d = static_cast<zmx_dist*>(self);
  const size_t world = static_cast<size_t>(d->world);
  if (d->rank != 0 && size) HIPCHK2(hipMemcpyAsync(d->d_send, blob, size, hipMemcpyHostToDevice, d->stream));
  const size_t base = world > 1 ? off[1] : 0;          // rank 0's own bytes are not staged
  RCCLCHK(d, d->rccl->GroupStart());
  if (d->rank != 0) {
    if (size) RCCLCHK(d, d->rccl->Send(d->d_send, size, ncclUint8, 0, d->comm, d->stream));
  } else {
    for (size_t r = 1; r < world; ++r) {
      if (all[r]) RCCLCHK(d, d->rccl->Recv(d->d_recv + (off[r] - base), all[r], ncclUint8, static_cast<int>(r), d->comm, d->stream));
    }
  }
  RCCLCHK(d, d->rccl->GroupEnd());
  if (d->rank == 0 && world > 1 && off[world] > base) {
    HIPCHK2(hipMemcpyAsync(host + base, d->d_recv, off[world] - base, hipMemcpyDeviceToHost, d->stream));
  }
  HIPCHK2(hipStreamSynchronize(d->stream));
  return 0;
}
}  // namespace

int zmx_dist_gather(zmx_dist* d, const unsigned char* blob, size_t size, unsigned char** gathered,
                    size_t* sizes) {
  DeviceScope scope;
  HIPCHK2(hipSetDevice(d->device));
  std::string err;
  zamd::GatherTransport t;
  t.self = d;
  t.rank = d->rank;
  t.world = d->world;
  t.all_gather_u64 = RcclAllGather;
  t.prepare = RcclPrepare;
  t.exchange = RcclExchange;
  t.error = &err;
  const int rc = zamd::GatherBlobs(t, blob, size, gathered, sizes);
  if (rc != 0 && !err.empty()) zmx_internal_set_error(err.c_str());
  return rc;
}

}  // extern "C"

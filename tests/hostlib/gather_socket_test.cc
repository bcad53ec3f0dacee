// TEST INFRASTRUCTURE: the product's gather logic (zopfli_amd/csrc/host/dist_core.h, the code dist.cc runs over
// RCCL) over a socket transport, with `world` ranks as forked processes around a hub at rank 0.  Exercises what a
// one-GPU box cannot: sizes of several ranks, offsets, an empty blob in the middle, and the agreed failure when
// one rank cannot prepare its buffers.
#include <sys/socket.h>
#include <sys/wait.h>
#include <unistd.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "dist_core.h"

namespace {

struct SockRank {
  int rank, world, fail_rank;
  std::vector<int> fd;   // rank 0: fd[r] to rank r; others: fd[0] to rank 0
};

bool WriteAll(int fd, const void* p, size_t n) {
  const char* c = static_cast<const char*>(p);
  while (n) {
    const ssize_t k = write(fd, c, n);
    if (k <= 0) return false;
    c += k;
    n -= static_cast<size_t>(k);
  }
  return true;
}
bool ReadAll(int fd, void* p, size_t n) {
  char* c = static_cast<char*>(p);
  while (n) {
    const ssize_t k = read(fd, c, n);
    if (k <= 0) return false;
    c += k;
    n -= static_cast<size_t>(k);
  }
  return true;
}

int SockAllGather(void* self, uint64_t mine, uint64_t* all) {
  SockRank* s = static_cast<SockRank*>(self);
  const size_t w = static_cast<size_t>(s->world);
  if (s->rank == 0) {
    all[0] = mine;
    for (size_t r = 1; r < w; ++r) if (!ReadAll(s->fd[r], &all[r], 8)) return -1;
    for (size_t r = 1; r < w; ++r) if (!WriteAll(s->fd[r], all, 8 * w)) return -1;
  } else {
    if (!WriteAll(s->fd[0], &mine, 8) || !ReadAll(s->fd[0], all, 8 * w)) return -1;
  }
  return 0;
}
int SockPrepare(void* self, size_t, size_t) {
  SockRank* s = static_cast<SockRank*>(self);
  return s->rank == s->fail_rank ? -1 : 0;
}
int SockExchange(void* self, const unsigned char* blob, size_t size, const uint64_t* all, const size_t* off,
                 unsigned char* host) {
  SockRank* s = static_cast<SockRank*>(self);
  if (s->rank != 0) return size == 0 || WriteAll(s->fd[0], blob, size) ? 0 : -1;
  for (size_t r = 1; r < static_cast<size_t>(s->world); ++r) {
    if (all[r] && !ReadAll(s->fd[r], host + off[r], static_cast<size_t>(all[r]))) return -1;
  }
  return 0;
}

std::vector<unsigned char> BlobOf(int rank) {
  const size_t n = rank == 2 ? 0 : 1000u * static_cast<size_t>(rank + 1) + static_cast<size_t>(rank);   // rank 2: nothing to send
  std::vector<unsigned char> b(n);
  for (size_t i = 0; i < n; ++i) b[i] = static_cast<unsigned char>(rank * 31 + i * 7);
  return b;
}

int RunRank(SockRank& me, unsigned char** gathered, size_t* sizes) {
  std::string err;
  zamd::GatherTransport t;
  t.self = &me;
  t.rank = me.rank;
  t.world = me.world;
  t.all_gather_u64 = SockAllGather;
  t.prepare = SockPrepare;
  t.exchange = SockExchange;
  t.error = &err;
  const std::vector<unsigned char> blob = BlobOf(me.rank);
  return zamd::GatherBlobs(t, blob.data(), blob.size(), gathered, sizes);
}

}  // namespace

// 0 = every rank behaved: without a failing rank, rank 0 holds every blob in rank order; with fail_rank in
// [0, world), every rank's gather returned an error (nobody hung, nobody believed it had worked).
extern "C" __attribute__((visibility("default"))) int zamd_test_gather(int world, int fail_rank) {
  std::vector<int> hub(static_cast<size_t>(world), -1), leaf(static_cast<size_t>(world), -1);
  for (int r = 1; r < world; ++r) {
    int sv[2];
    if (socketpair(AF_UNIX, SOCK_STREAM, 0, sv) != 0) return 100;
    hub[static_cast<size_t>(r)] = sv[0];
    leaf[static_cast<size_t>(r)] = sv[1];
  }
  std::vector<pid_t> kids;
  for (int r = 1; r < world; ++r) {
    const pid_t pid = fork();
    if (pid < 0) return 101;
    if (pid == 0) {
      for (int q = 1; q < world; ++q) { close(hub[static_cast<size_t>(q)]); if (q != r) close(leaf[static_cast<size_t>(q)]); }
      SockRank me{r, world, fail_rank, {leaf[static_cast<size_t>(r)]}};
      unsigned char* g = nullptr;
      std::vector<size_t> sizes(static_cast<size_t>(world));
      const int rc = RunRank(me, &g, sizes.data());
      _exit(rc == 0 ? (g == nullptr ? 0 : 5) : 3);
    }
    kids.push_back(pid);
  }
  for (int r = 1; r < world; ++r) close(leaf[static_cast<size_t>(r)]);
  SockRank me{0, world, fail_rank, hub};
  unsigned char* g = nullptr;
  std::vector<size_t> sizes(static_cast<size_t>(world));
  const int rc = RunRank(me, &g, sizes.data());
  int verdict = 0;
  const bool expect_fail = fail_rank >= 0 && fail_rank < world;
  if (expect_fail ? rc == 0 : rc != 0) verdict = 10;
  if (!expect_fail && rc == 0) {
    size_t at = 0;
    for (int r = 0; r < world && verdict == 0; ++r) {
      const std::vector<unsigned char> want = BlobOf(r);
      if (sizes[static_cast<size_t>(r)] != want.size()) verdict = 11;
      for (size_t i = 0; i < want.size() && verdict == 0; ++i) if (g[at + i] != want[i]) verdict = 12;
      at += want.size();
    }
  }
  std::free(g);
  for (pid_t pid : kids) {
    int st = 0;
    waitpid(pid, &st, 0);
    const int code = WIFEXITED(st) ? WEXITSTATUS(st) : 99;
    if (code != (expect_fail ? 3 : 0) && verdict == 0) verdict = 20 + code;
  }
  for (int r = 1; r < world; ++r) close(hub[static_cast<size_t>(r)]);
  return verdict;
}

// TEST INFRASTRUCTURE — a stand-in for librccl.so that moves HOST buffers over Unix sockets (star through rank 0), so that the N > 1 exchange of
// `tortoise --devices N --exchange rccl` (csrc/cli_rccl.h: rank pairing, size all-gather, send / receive to rank 0) executes on a box without GPUs:
//   TTS_RCCL_LIB=<this .so> tortoise --dry-run 1 --devices 4 --exchange rccl ...      (tests/test_distributed_cpu.py builds it with g++)
// Implements exactly the ten entry points cli_rccl.h resolves. Every call is blocking and executed at once (the `stream` argument is ignored); ranks
// must issue their calls in the same order, as NCCL requires. Send / Recv are supported between rank 0 and any other rank (all the CLI uses).
#define __HIP_PLATFORM_AMD__ 1
#include <rccl/rccl.h>
#include <sys/socket.h>
#include <sys/un.h>
#include <unistd.h>
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <vector>

struct ncclComm {
  int rank = 0, n = 1;
  std::vector<int> peer; // rank 0: socket of every other rank; rank r: peer[0] = its socket to rank 0
  int listen_fd = -1;
  char path[sizeof(((sockaddr_un *)nullptr)->sun_path)] = {};
};

static bool wr(int fd, const void *p, size_t n) {
  const char *c = (const char *)p;
  while (n) {
    const ssize_t k = write(fd, c, n);
    if (k < 0) { if (errno == EINTR) continue; return false; }
    c += k; n -= (size_t)k;
  }
  return true;
}
// FAKE_RCCL_HANG_ON_PEER_LOSS=1: a rank whose peer went away blocks for ever instead of failing — what the real library does when a rank dies inside a collective;
// the parent process has to end such ranks (csrc/cli_main.cpp: SIGTERM to the workers still alive when one exits with an error)
static bool peer_lost() {
  if (getenv("FAKE_RCCL_HANG_ON_PEER_LOSS")) for (;;) pause();
  return false;
}
static bool rd(int fd, void *p, size_t n) {
  char *c = (char *)p;
  while (n) {
    const ssize_t k = read(fd, c, n);
    if (k == 0) return peer_lost();
    if (k < 0) { if (errno == EINTR) continue; return false; }
    c += k; n -= (size_t)k;
  }
  return true;
}
static size_t tsize(ncclDataType_t t) {
  switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: case ncclBfloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
    default: return 0;
  }
}

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
  memset(id, 0, sizeof *id);
  const char *dir = getenv("TMPDIR");
  snprintf(id->internal, sizeof id->internal, "%s/fake_rccl_%d_%ld.sock", dir && *dir ? dir : "/tmp", (int)getpid(), (long)time(nullptr));
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *out, int n, ncclUniqueId id, int rank) {
  if (n < 1 || rank < 0 || rank >= n) return ncclInvalidArgument;
  ncclComm *c = new ncclComm;
  c->rank = rank; c->n = n;
  sockaddr_un a{};
  a.sun_family = AF_UNIX;
  id.internal[sizeof id.internal - 1] = 0;
  strncpy(a.sun_path, id.internal, sizeof a.sun_path - 1);
  strncpy(c->path, a.sun_path, sizeof c->path - 1);
  if (rank == 0) {
    c->peer.assign(n, -1);
    c->listen_fd = socket(AF_UNIX, SOCK_STREAM, 0);
    unlink(a.sun_path);
    if (c->listen_fd < 0 || bind(c->listen_fd, (sockaddr *)&a, sizeof a) || listen(c->listen_fd, n)) { delete c; return ncclSystemError; }
    for (int k = 1; k < n; k++) {
      const int fd = accept(c->listen_fd, nullptr, nullptr);
      int32_t r = -1;
      if (fd < 0 || !rd(fd, &r, 4) || r < 1 || r >= n || c->peer[r] >= 0) { delete c; return ncclSystemError; }
      c->peer[r] = fd;
    }
  } else {
    const int fd = socket(AF_UNIX, SOCK_STREAM, 0);
    bool ok = false;
    for (int tries = 0; tries < 3000 && !ok; tries++) { // rank 0 may not be listening yet
      ok = connect(fd, (sockaddr *)&a, sizeof a) == 0;
      if (!ok) usleep(10000);
    }
    const int32_t r = rank;
    if (!ok || !wr(fd, &r, 4)) { delete c; return ncclSystemError; }
    c->peer.assign(1, fd);
  }
  *out = c;
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t c) {
  if (!c) return ncclSuccess;
  for (int fd : c->peer) if (fd >= 0) close(fd);
  if (c->listen_fd >= 0) { close(c->listen_fd); unlink(c->path); }
  delete c;
  return ncclSuccess;
}

ncclResult_t ncclBroadcast(const void *send, void *recv, size_t count, ncclDataType_t t, int root, ncclComm_t c, hipStream_t) {
  const size_t bytes = count * tsize(t);
  if (!tsize(t) || root < 0 || root >= c->n) return ncclInvalidArgument;
  if (c->rank == 0) {
    if (root == 0) { if (recv != send) memmove(recv, send, bytes); }
    else if (!rd(c->peer[root], recv, bytes)) return ncclSystemError;
    for (int r = 1; r < c->n; r++)
      if (r != root && !wr(c->peer[r], recv, bytes)) return ncclSystemError;
  } else if (c->rank == root) {
    if (!wr(c->peer[0], send, bytes)) return ncclSystemError;
    if (recv != send) memmove(recv, send, bytes);
  } else if (!rd(c->peer[0], recv, bytes)) return ncclSystemError;
  return ncclSuccess;
}

ncclResult_t ncclAllGather(const void *send, void *recv, size_t count, ncclDataType_t t, ncclComm_t c, hipStream_t) {
  const size_t bytes = count * tsize(t);
  if (!tsize(t)) return ncclInvalidArgument;
  char *out = (char *)recv;
  if (c->rank == 0) {
    memmove(out, send, bytes);
    for (int r = 1; r < c->n; r++)
      if (!rd(c->peer[r], out + (size_t)r * bytes, bytes)) return ncclSystemError;
    for (int r = 1; r < c->n; r++)
      if (!wr(c->peer[r], out, bytes * c->n)) return ncclSystemError;
  } else {
    if (!wr(c->peer[0], send, bytes) || !rd(c->peer[0], out, bytes * c->n)) return ncclSystemError;
  }
  return ncclSuccess;
}

ncclResult_t ncclSend(const void *send, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t) {
  if (!tsize(t) || peer < 0 || peer >= c->n || peer == c->rank) return ncclInvalidArgument;
  if (c->rank != 0 && peer != 0) return ncclInvalidUsage; // the star has no edge between two non-zero ranks
  return wr(c->rank == 0 ? c->peer[peer] : c->peer[0], send, count * tsize(t)) ? ncclSuccess : ncclSystemError;
}

ncclResult_t ncclRecv(void *recv, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t) {
  if (!tsize(t) || peer < 0 || peer >= c->n || peer == c->rank) return ncclInvalidArgument;
  if (c->rank != 0 && peer != 0) return ncclInvalidUsage;
  return rd(c->rank == 0 ? c->peer[peer] : c->peer[0], recv, count * tsize(t)) ? ncclSuccess : ncclSystemError;
}

ncclResult_t ncclGroupStart() { return ncclSuccess; }
ncclResult_t ncclGroupEnd() { return ncclSuccess; }
const char *ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : r == ncclInvalidUsage ? "fake rccl: invalid usage" : "fake rccl: error"; }

} // extern "C"

// RCCL exchange between the worker processes of `tortoise --devices N --exchange rccl` (one process per GPU):
//   broadcast of the conditioning (text ids + the 4 KB voice latent) from rank 0, all-gather of the per-rank result sizes / CLVP scores,
//   send / receive of the audio to rank 0, which writes every WAV file. Candidates never interact, so nothing sits inside the data path:
//   these are KB .. MB messages at the start and the end of an utterance (latency-bound on xGMI).
// librccl.so (0.5 GB) is opened with dlopen only when this mode is asked for: a single-GPU run does not depend on it.
// The reference is single-device (main.cpp:651 picks ONE backend): there is no reference code to cite.
#pragma once
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

struct RcclApi {
  void *h = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclBroadcast) Broadcast = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  bool open() {
    // TTS_RCCL_LIB=<path>: another library with the same ten entry points (tests/fake_rccl.cpp: host buffers over a Unix socket, so that the
    // N > 1 pairing / size logic below executes on a box without GPUs)
    if (const char *over = getenv("TTS_RCCL_LIB")) {
      if (!(h = dlopen(over, RTLD_NOW | RTLD_LOCAL))) { fprintf(stderr, "rccl: cannot open TTS_RCCL_LIB=%s: %s\n", over, dlerror()); return false; }
    } else {
      for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
        if ((h = dlopen(name, RTLD_NOW | RTLD_LOCAL))) break;
    }
    if (!h) { fprintf(stderr, "rccl: cannot open librccl.so: %s\n", dlerror()); return false; }
#define RCCL_SYM(f) if (!(f = (decltype(f))dlsym(h, "nccl" #f))) { fprintf(stderr, "rccl: symbol nccl" #f " missing\n"); return false; }
    RCCL_SYM(GetUniqueId) RCCL_SYM(CommInitRank) RCCL_SYM(CommDestroy) RCCL_SYM(Broadcast) RCCL_SYM(AllGather) RCCL_SYM(Send) RCCL_SYM(Recv)
    RCCL_SYM(GroupStart) RCCL_SYM(GroupEnd) RCCL_SYM(GetErrorString)
#undef RCCL_SYM
    return true;
  }
};

inline std::string rccl_id_to_hex(const ncclUniqueId &id) {
  static const char *d = "0123456789abcdef";
  std::string s;
  for (size_t i = 0; i < sizeof(id.internal); i++) { const unsigned char c = (unsigned char)id.internal[i]; s += d[c >> 4]; s += d[c & 15]; }
  return s;
}
inline bool rccl_id_from_hex(const std::string &s, ncclUniqueId &id) {
  if (s.size() != 2 * sizeof(id.internal)) return false;
  auto v = [](char c) { return c >= '0' && c <= '9' ? c - '0' : c >= 'a' && c <= 'f' ? c - 'a' + 10 : -1; };
  for (size_t i = 0; i < sizeof(id.internal); i++) {
    const int a = v(s[2 * i]), b = v(s[2 * i + 1]);
    if (a < 0 || b < 0) return false;
    id.internal[i] = (char)(a * 16 + b);
  }
  return true;
}

// One communicator over the worker processes. Every call is collective and blocking (stream-synchronised): the messages are tiny.
struct RcclWorld {
  RcclApi api;
  ncclComm_t comm = nullptr;
  hipStream_t stream = nullptr;
  int rank = 0, n = 1;
  std::string err;
  bool fail(const char *what, ncclResult_t r) { err = std::string(what) + ": " + (api.GetErrorString ? api.GetErrorString(r) : "?"); return false; }
  bool failh(const char *what, hipError_t e) { err = std::string(what) + ": " + hipGetErrorString(e); return false; }
  // host_only (tortoise --dry-run: no device in the process): the staging buffers are host memory and no stream exists — only meaningful with a
  // TTS_RCCL_LIB stand-in that moves host buffers; the real librccl would reject them
  bool host_only = false;
  hipError_t dmalloc(void **p, size_t bytes) {
    if (!host_only) return hipMalloc(p, bytes);
    *p = malloc(bytes ? bytes : 1);
    return *p ? hipSuccess : hipErrorOutOfMemory;
  }
  hipError_t copy(void *dst, const void *src, size_t bytes, hipMemcpyKind kind) {
    if (!host_only) return hipMemcpyAsync(dst, src, bytes, kind, stream);
    memcpy(dst, src, bytes);
    return hipSuccess;
  }
  hipError_t sync() { return host_only ? hipSuccess : hipStreamSynchronize(stream); }
  bool init(const std::string &hex_id, int rank_, int n_, bool host_only_ = false) { // the calling process has already selected its device (tts_create)
    rank = rank_; n = n_; host_only = host_only_;
    // host staging buffers and no stream: only a stand-in library that moves host memory can take them; the real librccl would dereference them as device pointers
    if (host_only && !getenv("TTS_RCCL_LIB")) { err = "--dry-run with --exchange rccl needs TTS_RCCL_LIB=<stand-in library> (tests/fake_rccl.cpp): librccl cannot take host buffers"; return false; }
    ncclUniqueId id;
    if (!rccl_id_from_hex(hex_id, id)) { err = "bad --rccl-id"; return false; }
    if (!api.open()) { err = "librccl.so not available"; return false; }
    hipError_t e = host_only ? hipSuccess : hipStreamCreate(&stream);
    if (e != hipSuccess) return failh("hipStreamCreate", e);
    ncclResult_t r = api.CommInitRank(&comm, n, id, rank);
    if (r != ncclSuccess) return fail("ncclCommInitRank (distinct GPUs per rank required)", r);
    return true;
  }
  ~RcclWorld() {
    if (comm) api.CommDestroy(comm);
    if (stream) (void)hipStreamDestroy(stream);
  }
  struct Dev { // staging buffer (device; host under host_only)
    void *p = nullptr;
    bool host;
    explicit Dev(const RcclWorld &w) : host(w.host_only) {}
    ~Dev() { if (p) { if (host) free(p); else (void)hipFree(p); } }
  };
  bool broadcast(void *host, size_t bytes, int root) {
    Dev d(*this);
    hipError_t e = dmalloc(&d.p, bytes);
    if (e != hipSuccess) return failh("hipMalloc", e);
    if (rank == root && (e = copy(d.p, host, bytes, hipMemcpyHostToDevice)) != hipSuccess) return failh("hipMemcpyAsync", e);
    ncclResult_t r = api.Broadcast(d.p, d.p, bytes, ncclChar, root, comm, stream);
    if (r != ncclSuccess) return fail("ncclBroadcast", r);
    if ((e = copy(host, d.p, bytes, hipMemcpyDeviceToHost)) != hipSuccess) return failh("hipMemcpyAsync", e);
    if ((e = sync()) != hipSuccess) return failh("hipStreamSynchronize", e);
    return true;
  }
  // every rank contributes `bytes`; out = n x bytes in rank order
  bool all_gather(const void *mine, size_t bytes, std::vector<char> &out) {
    Dev s(*this), d(*this);
    hipError_t e;
    if ((e = dmalloc(&s.p, bytes)) != hipSuccess || (e = dmalloc(&d.p, bytes * n)) != hipSuccess) return failh("hipMalloc", e);
    if ((e = copy(s.p, mine, bytes, hipMemcpyHostToDevice)) != hipSuccess) return failh("hipMemcpyAsync", e);
    ncclResult_t r = api.AllGather(s.p, d.p, bytes, ncclChar, comm, stream);
    if (r != ncclSuccess) return fail("ncclAllGather", r);
    out.resize(bytes * n);
    if ((e = copy(out.data(), d.p, bytes * n, hipMemcpyDeviceToHost)) != hipSuccess) return failh("hipMemcpyAsync", e);
    if ((e = sync()) != hipSuccess) return failh("hipStreamSynchronize", e);
    return true;
  }
  // rank `from` sends counts[from] floats to rank `to` (every rank calls it with the same arguments; the others do nothing)
  bool send_floats(const float *mine, const std::vector<int64_t> &counts, int from, int to, std::vector<float> &out_at_to) {
    if (from == to) { if (rank == to) out_at_to.assign(mine, mine + counts[from]); return true; }
    if (rank != from && rank != to) return true;
    Dev d(*this);
    const size_t bytes = (size_t)counts[from] * 4;
    hipError_t e = dmalloc(&d.p, bytes ? bytes : 4);
    if (e != hipSuccess) return failh("hipMalloc", e);
    ncclResult_t r;
    if (rank == from) {
      if ((e = copy(d.p, mine, bytes, hipMemcpyHostToDevice)) != hipSuccess) return failh("hipMemcpyAsync", e);
      if ((r = api.Send(d.p, (size_t)counts[from], ncclFloat, to, comm, stream)) != ncclSuccess) return fail("ncclSend", r);
    } else {
      if ((r = api.Recv(d.p, (size_t)counts[from], ncclFloat, from, comm, stream)) != ncclSuccess) return fail("ncclRecv", r);
      out_at_to.resize((size_t)counts[from]);
      if ((e = copy(out_at_to.data(), d.p, bytes, hipMemcpyDeviceToHost)) != hipSuccess) return failh("hipMemcpyAsync", e);
    }
    if ((e = sync()) != hipSuccess) return failh("hipStreamSynchronize", e);
    return true;
  }
};

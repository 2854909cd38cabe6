// Shared internals of libtortoise_mi355x.so (host side). gfx950 only; no CPU fallback anywhere:
// every stage entry point fails with TTS_ERR_HIP if the device path is unavailable.
#pragma once
#include "../../include/tortoise_mi355x.h"
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <random>
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

namespace tts {

struct HostTensor {
  std::vector<float> data;  // empty for a tensor left in the file (file_off >= 0): see read_weight_file's lazy_from
  int64_t ne[4] = {1, 1, 1, 1};
  int n_dims = 0;
  int64_t file_off = -1;    // byte offset of the payload in the file
  int64_t nelem() const { return ne[0] * ne[1] * ne[2] * ne[3]; }
};
// Name-keyed legacy-ggml container (format: main.cpp:811-888).
struct WeightFile {
  std::map<std::string, HostTensor> t;
  int fd = -1; // open while tensors are left in the file
  WeightFile() = default;
  WeightFile(const WeightFile &) = delete;
  WeightFile &operator=(const WeightFile &) = delete;
  ~WeightFile();
  bool has(const std::string &n) const { return t.count(n) != 0; }
  // payload of a tensor that was left in the file, straight into dst (e.g. pinned staging memory); false on a short read
  bool read_payload(const HostTensor &ht, void *dst) const;
};
// lazy_from: payloads of at least this many bytes are NOT read (HostTensor::data stays empty, file_off says where they are; WeightFile::fd stays open) — the AR loader
// reads them straight into pinned staging memory on its worker threads instead of through 1.6 GB of freshly faulted-in host vectors.
int read_weight_file(const char *path, WeightFile &out, std::string &err, size_t lazy_from = (size_t)-1);

// Grow-only device buffer.
struct DevBuf {
  void *p = nullptr;
  size_t cap = 0;
  ~DevBuf() { if (p) (void)hipFree(p); }
  DevBuf() = default;
  DevBuf(const DevBuf &) = delete;
  DevBuf &operator=(const DevBuf &) = delete;
  hipError_t reserve(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    if (p) (void)hipFree(p);
    p = nullptr; cap = 0;
    hipError_t e = hipMalloc(&p, bytes);
    if (e == hipSuccess) cap = bytes;
    return e;
  }
  template <class T> T *as() const { return (T *)p; }
};

struct PinnedBuf { // page-locked host memory that lives with its owner (asynchronous copies need it)
  void *p = nullptr;
  size_t cap = 0;
  ~PinnedBuf() { if (p) (void)hipHostFree(p); }
  PinnedBuf() = default;
  PinnedBuf(const PinnedBuf &) = delete;
  PinnedBuf &operator=(const PinnedBuf &) = delete;
  hipError_t reserve(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    if (p) (void)hipHostFree(p);
    p = nullptr; cap = 0;
    hipError_t e = hipHostMalloc(&p, bytes, hipHostMallocDefault);
    if (e == hipSuccess) cap = bytes;
    return e;
  }
  template <class T> T *as() const { return (T *)p; }
};

struct ProfEntry { double ms = 0, work = 0; int64_t launches = 0, seen = 0; std::vector<std::pair<hipEvent_t, hipEvent_t>> pending; };

struct ArState;
struct DiffState;
struct VocState;
struct ClvpState;
struct VoiceEncState;
struct DiffCondEncState;
struct Tokenizer;
struct SamplerPool;
void sampler_pool_free(SamplerPool *p);

} // namespace tts

struct tts_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  hipStream_t load_stream = nullptr; // non-blocking: the loaders' uploads (a legacy-stream copy on a second thread is an error while `stream` captures a graph: the CLI loads the diffusion and vocoder models beside the AR stage)
  int stream_cus = 0; // option stream_cus: CUs per XCD this context's stream is confined to (> 0) / excluded from (< 0); 0 = whole chip
  std::string err;
  // options
  float gn_eps = 1e-6f;
  int ggml_lut = 0;
  // RNG: the reference's three globals (main.cpp:47-50)
  std::mt19937 generator;
  std::uniform_real_distribution<float> distribution{0.0, 1.0};
  std::normal_distribution<double> normal_distribution{0.0, 1.0};
  uint32_t seed_value = 0;
  // stages
  tts::ArState *ar = nullptr;
  tts::DiffState *diff = nullptr;
  tts::VocState *voc = nullptr;
  tts::ClvpState *clvp = nullptr; // candidate re-ranker (extras.hip; not in the reference)
  tts::VoiceEncState *venc = nullptr; // voice-conditioning encoder (extras.hip; not in the reference)
  tts::DiffCondEncState *dcond = nullptr; // diffusion conditioning encoder (extras.hip; not in the reference)
  tts::Tokenizer *tok = nullptr;
  tts::SamplerPool *sampler_pool = nullptr; // worker threads for the per-candidate sampler scans (host_logic.cpp)
  int device_topk = 1;                      // option "device_topk": tts_autoregressive's loop samples from the device prefilter's lists
  int time_mlp_retries = 0;                 // evaluations of the diffusion time MLP that disagreed with their repetition (diffusion.hip: precompute_time)
  int topk_fallbacks = 0;                   // candidates x steps of the last tts_autoregressive call that needed their full row
  int sampler_threads = -1;                 // -1: min(7, hardware threads - 1); option "sampler_threads"
  bool share_uncond = true;                 // option "share_uncond": see DiffState::share_integ (diffusion.hip)
  // candidate-parallel sharding: this context runs candidates [rng_shard_offset, +B) of a batch of rng_shard_total
  // (0 = unsharded). The sampler skips the other ranks' uniforms; device noise streams are keyed by the global candidate id.
  int rng_shard_offset = 0, rng_shard_total = 0;
  std::vector<int32_t> stop_schedule; // tts_ar_set_stop_schedule: forced stop iteration per candidate (empty = none)
  std::vector<int32_t> ar_stopped; // last tts_autoregressive call, per candidate: 1 = it sampled the stop token, 0 = cut at max_steps (tts_ar_stop_status)
  // profiling: per kernel family, HIP event pairs recorded on the ctx stream around every launch and
  // resolved lazily (no host sync inside the timed region)
  bool prof_on = false;
  std::vector<std::string> prof_filter; // empty = every family; "prof_only:<family>" = 1 adds one, = 0 clears the list
  int prof_stride = 1;     // option "prof_stride": every Nth launch of a family is bracketed (an event pair drains the pipeline)
  int ar_weights = 0;      // option "ar_weights": 0 = f32 weights in the decode step (reference numerics), 1 = fp16 weights, 2 = OCP fp8 (e4m3) weights with a power-of-two scale per output column (set before tts_load_ar)
  int dec_f32_mfma = 0;    // option "dec_f32_mfma": the decode step's LayerNorm-GEMVs use exact-f32 MFMA products instead of split fp16 (set before tts_load_ar; +1 us per launch)
  int attn_f32 = 0;        // option "attn_f32": the diffusion AttentionBlock in reference precision (F32 QK^T / softmax / PV / proj_out, main.cpp:3848-3875) via split-fp16 MFMA operands; 0 = fp16 operands (throughput mode)
  int proj_dual_b = 1;     // option "proj_dual_b" (developer A/B): the split-weight proj_out GEMM stages both weight halves per activation tile (1) or runs two K segments (0)
  int attn_f32_drop = 0;   // option "attn_f32_drop" (developer ablation inside attn_f32 = 1): bit 0 q/k, bit 1 v, bit 2 attention output lose their low halves (= the fp16 rounding of the default mode, one operand at a time)
  int lc_attn_f32 = 1;     // option "lc_attn_f32": the latent conditioner's AttentionBlocks (once per utterance; their output enters every step) in reference precision whatever attn_f32 says; 0 = follow attn_f32 (rounds 1-4)
  int attn_proj_f16 = 0;   // option "attn_proj_f16": 1 = proj_out's weight as ONE fp16 operand (the all-fp16 AttentionBlock of rounds 1-4, A/B only); 0 default = split pair W_hi + W_lo (F32-accurate weight, round 5)
  int fp16_check = 0;      // option "fp16_check": scan every fp16 operand the diffusion stage writes for non-finite / saturated values (tts_diffusion_fp16_check)
  int64_t fp16_bad_weights[2] = {0, 0}; // the same two counts over the split-precision weights packed by the last tts_load_diffusion
  void *fp16_counts = nullptr;           // device: int64[2]
  int rng_fast_normal = 1; // option "rng_fast_normal": 0 = every normal draw through std::normal_distribution::operator() (A/B and the tests' reference for the fast form)
  int noise_pipeline = 1;  // option "noise_pipeline": TTS_NOISE_REFERENCE with one candidate draws a step's noise on the host while the device runs the previous steps (same draws in the same order)
  int load_device_pack = 1; // option "load_device_pack": tts_load_ar builds its decode layouts with kernels from the uploaded file tensors (0: on the host threads)
  int load_threads = 0;    // option "load_threads": host threads of the tts_load_* calls (0 = min(16, hardware threads); 1 = single-threaded)
  int attn_q64 = 0; // option "attn_q64": diffusion attention with 64-query workgroups: 0 never (default: measured, no gain), 1 always, 2 = when the 128-query grid has at most 256 workgroups (bit-identical)
  int hoist_integrator = 1; // option "hoist_integrator": small diffusion batches evaluate the conditioning_timestep_integrator layers (which never see x_t) for all sampling steps before the loop, in benchmark-sized batches (bit-identical; 0 = inside every step)
  int latency_mode = 0;    // option "latency_mode": small diffusion batches (<= 2 048 packed rows) take the GroupNorm statistics from the producing GEMM's epilogue (diffusion.hip: gn_apply_kernel); not bit-identical to the batch path
  bool capturing = false;  // a hipGraph is being captured on the stream: ProfScope records nothing (event records would become graph nodes)
  int diff_graph = 1;      // option "diff_graph": the diffusion step is captured once per call and replayed (0: every step launched eagerly)
  int prof_eager_every = 8; // while a diff_* family is profiled, every Nth diffusion step runs eagerly with its event pairs; the others replay the graph
  std::map<std::string, tts::ProfEntry> prof;
  std::vector<hipEvent_t> ev_pool;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
};

namespace tts {

int fail(tts_ctx *ctx, int code, const char *fmt, ...);

// n draws of the context's normal distribution (sample_normal_noise, main.cpp:4695-4701: std::normal_distribution<double> over std::mt19937, values narrowed to float) into
// dst, leaving generator and distribution in exactly the state n single draws leave them in. Large counts take a two-phase form of the same algorithm (host_logic.cpp).
void rng_normal_fill(tts_ctx *ctx, float *dst, int64_t n);

// Host staging for the loaders' uploads: a pageable hipMemcpy goes through the runtime's own bounce buffer at ~5 GB/s, and tts_load_ar moves 4.6 GB (every matrix in two
// or three layouts); from pinned memory the same copies run at PCIe speed. A few pinned buffers, taken and given back by the load workers, freed when the load returns.
struct PinnedPool {
  // at most `max_bufs` buffers of at least `min_cap` bytes exist at any time (pinning costs ~0.4 ms per MB, and so does the release): a worker that finds none idle waits
  // for one — an upload holds its buffer for a host memcpy + a DMA, a few ms against the ~20 ms of packing between two uploads
  explicit PinnedPool(size_t min_cap_ = (size_t)18 << 20, int max_bufs_ = 4) : min_cap(min_cap_), max_bufs(max_bufs_) {}
  size_t min_cap;
  int max_bufs, made = 0;
  std::mutex mu;
  std::condition_variable cv;
  std::vector<std::pair<void *, size_t>> idle;
  ~PinnedPool() { for (auto &b : idle) (void)hipHostFree(b.first); }
  std::pair<void *, size_t> take(size_t n) {
    std::unique_lock<std::mutex> lk(mu);
    for (;;) {
      for (size_t i = 0; i < idle.size(); i++)
        if (idle[i].second >= n) { auto b = idle[i]; idle.erase(idle.begin() + (long)i); return b; }
      if (made < max_bufs) { made++; break; }                                                  // room for one more
      if (!idle.empty()) { (void)hipHostFree(idle.back().first); idle.pop_back(); break; }     // all too small: replace one
      cv.wait(lk);
    }
    lk.unlock();
    const size_t cap = std::max(n, min_cap);
    void *p = nullptr;
    if (hipHostMalloc(&p, cap, hipHostMallocDefault) != hipSuccess) {
      lk.lock(); made--; lk.unlock(); cv.notify_one();
      return {nullptr, 0};
    }
    return {p, cap};
  }
  void give(std::pair<void *, size_t> b) {
    if (!b.first) return;
    { std::lock_guard<std::mutex> lk(mu); idle.push_back(b); }
    cv.notify_one();
  }
  // dst (device) <- src (pageable host), complete on return; on stream s when given (never the legacy stream then); falls back to the plain copy when no pinned
  // memory can be had
  static hipError_t copy_now(void *dst, const void *src, size_t bytes, hipStream_t s) {
    if (!s) return hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice);
    const hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, s);
    return e != hipSuccess ? e : hipStreamSynchronize(s);
  }
  hipError_t upload(void *dst, const void *src, size_t bytes, hipStream_t s = nullptr) {
    auto b = take(bytes);
    if (!b.first) return copy_now(dst, src, bytes, s);
    memcpy(b.first, src, bytes);
    const hipError_t e = copy_now(dst, b.first, bytes, s);
    give(b);
    return e;
  }
};

// Load-time host work (layout transforms and uploads of independent tensors: the reference's loaders are one pass over the file, main.cpp:811-888; here every layer's
// decode slabs / fp16 copies are built on the host) spread over a few threads: body(i) for i in [0, n), first non-zero status wins, no exception leaves a worker.
// Option "load_threads": 0 = min(16, hardware threads), 1 = the single-threaded loaders of rounds 1-5. Everything body() shares must be guarded by the caller
// (fail() serialises the error text itself).
inline int run_parallel(tts_ctx *ctx, int n, const std::function<int(int)> &body) {
  int nt = ctx->load_threads > 0 ? ctx->load_threads : (int)std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
  nt = std::min(nt, n);
  std::atomic<int> next{0}, rc{0};
  auto work = [&]() {
    if (ctx->device >= 0) (void)hipSetDevice(ctx->device); // the current device is per thread
    for (;;) {
      const int i = next.fetch_add(1);
      if (i >= n || rc.load()) break;
      int r;
      try {
        r = body(i);
      } catch (const std::bad_alloc &) {
        r = fail(ctx, TTS_ERR_LIMIT, "out of host memory");
      } catch (...) {
        r = fail(ctx, TTS_ERR_STATE, "internal error in a load worker");
      }
      if (r) { int z = 0; rc.compare_exchange_strong(z, r); }
    }
  };
  std::vector<std::thread> th;
  for (int t = 1; t < nt; t++) th.emplace_back(work);
  work();
  for (auto &t : th) t.join();
  return rc.load();
}

// Global id of this context's candidate 0: the ONE place that interprets rng_shard_offset / rng_shard_total, used by the
// sampler's stream partition and by the device noise streams of the diffusion and vocoder stages alike. total == 0 = unsharded:
// the offset is ignored everywhere.
inline int shard_base(const tts_ctx *c) { return c->rng_shard_total > 0 ? c->rng_shard_offset : 0; }
// B candidates must fit the declared batch; TTS_OK or TTS_ERR_ARG (message set).
inline int shard_check(tts_ctx *c, int B) {
  if (c->rng_shard_total > 0 && c->rng_shard_offset + B > c->rng_shard_total)
    return fail(c, TTS_ERR_ARG, "candidates [%d, %d) exceed rng_shard_total %d", c->rng_shard_offset, c->rng_shard_offset + B, c->rng_shard_total);
  return TTS_OK;
}

#define TTS_HIP(ctx, expr)                                                                      \
  do {                                                                                          \
    hipError_t _e = (expr);                                                                     \
    if (_e != hipSuccess)                                                                       \
      return tts::fail(ctx, TTS_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                       __FILE__, __LINE__);                                                     \
  } while (0)

// Brackets a kernel (family) with HIP events on the ctx stream when profiling is enabled.
hipEvent_t prof_event(tts_ctx *c);
struct ProfScope {
  tts_ctx *c; const char *fam; hipEvent_t a = nullptr;
  // work: algorithmic FLOPs (MFMA-bound families) or bytes (HBM-bound families) of this launch
  ProfScope(tts_ctx *ctx, const char *family, double work = 0) : c(ctx), fam(family) {
    bool want = c->prof_on && !c->capturing && c->prof_filter.empty();
    if (c->prof_on && !c->capturing && !want)
      for (const std::string &f : c->prof_filter) want |= (f == fam) || (strncmp(fam, f.c_str(), f.size()) == 0 && fam[f.size()] == '_'); // a name also selects its sub-families
    if (want) {
      ProfEntry &e = c->prof[fam];
      // Which launches are bracketed: a hash of the family's launch counter, 1 in prof_stride on average. (Rounds 2-4 took every prof_stride-th launch of the
      // family; round 5's default arithmetic made the QKV projection exactly 13 launches per sampling step and the stride of 13 then bracketed the SAME layer —
      // one of the three smaller integrator launches — in every step: a biased family average. A hashed decimation has no period to resonate with.)
      uint32_t hsh = (uint32_t)e.seen++ * 2654435761u;
      hsh ^= hsh >> 15; hsh *= 0x2c1b3c6du; hsh ^= hsh >> 12;
      if (c->prof_stride <= 1 || hsh % (uint32_t)c->prof_stride == 0) { // work and time are accumulated over the bracketed launches only
        a = prof_event(c);
        (void)hipEventRecord(a, c->stream);
        e.work += work;
      }
    }
  }
  ~ProfScope() {
    if (a) {
      hipEvent_t b = prof_event(c);
      (void)hipEventRecord(b, c->stream);
      c->prof[fam].pending.emplace_back(a, b);
    }
  }
};

// host_logic.cpp
struct Tokenizer {
  std::map<std::string, int> vocab;
  bool load(const char *path);
  std::vector<int> encode(const std::string &message) const;
};
void sample_candidates(tts_ctx *ctx, const float *logits, const int32_t *ids, int ids_per_cand, int B,
                       int32_t *out);
// Device top-k prefilter of the decode step (ar.hip: sample_prefilter_kernel; option "device_topk"): per candidate TTS_PF_WORDS
// 32-bit words {n, 0, 0, 0, idx[TTS_PF_MAX], logit bits[TTS_PF_MAX]} = every logit >= a threshold that keeps TTS_PF_MIN..TTS_PF_MAX
// of the 8194, in index order (n = -1: no such threshold, the host samples from the full row).
enum { TTS_PF_MIN = 64, TTS_PF_MAX = 128, TTS_PF_WORDS = 4 + 2 * TTS_PF_MAX };
int sample_candidates_list(tts_ctx *ctx, const int32_t *lists, const int32_t *ids, int ids_per_cand, int B, int32_t *out,
                           const std::function<const float *(int)> &full_row, int *n_fallbacks, const char *retired = nullptr);
int host_prefilter_row(const float *row, int keep, int32_t *list);
int sample_one_row(const float *row, const int32_t *ids, int ids_per_cand, float uniform);
int sample_one_from_list(const int32_t *list, const int32_t *ids, int ids_per_cand, float uniform);
void pad_codes(std::vector<int> &codes);            // apply_padding
int trimmed_latent_rows(const int32_t *codes502);   // trim_latents row count
struct DiffSchedule {
  int n = 0;
  std::vector<int> timestep_map;
  // per sampled step t (already cast the way the reference casts them for the update)
  std::vector<float> max_log, min_log, cfk, sqrt_recip, sqrt_recipm1, coef1, coef2;
  void build(int n_steps);
};
void timestep_embedding(int t, float *out1024);
int rel_bucket(int i, int c);

// stage entry points implemented in the .hip files
int ar_load(tts_ctx *ctx, const char *path);
void ar_free(ArState *);
int diff_load(tts_ctx *ctx, const char *path);
void diff_free(DiffState *);
int voc_load(tts_ctx *ctx, const char *path);
// frames of context tts_vocoder_chunk adds on either side of a window; vocoder.hip static_asserts that it covers the receptive field
// computed from the architecture constants its loader enforces
#define TTS_VOC_CHUNK_HALO 24
int voc_halo_frames();
void voc_free(VocState *);
int diff_cond_enc_load(tts_ctx *ctx, const char *path);
void diff_cond_enc_free(DiffCondEncState *);
int diff_cond_enc_latent(tts_ctx *ctx, const float *mel, const int32_t *frames, int n_clips, float *out2048);
int diff_fp16_check(tts_ctx *ctx, int64_t counts[2]);            // diffusion.hip
int diff_set_cond_latent(tts_ctx *ctx, const float *latent2048); // diffusion.hip: overrides the weight file's diffusion_conditioning_latent
int voice_enc_load(tts_ctx *ctx, const char *path);
void voice_enc_free(VoiceEncState *);
int voice_enc_latent(tts_ctx *ctx, const float *mel, const int32_t *frames, int n_clips, float *out1024);
int clvp_load(tts_ctx *ctx, const char *path);
void clvp_free(ClvpState *);
int clvp_score(tts_ctx *ctx, const int32_t *text_ids, int n_text, const int32_t *codes, const int32_t *code_len, int n_candidates, int code_stride,
               float *scores_out);

} // namespace tts

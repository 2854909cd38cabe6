// C ABI of libtortoise_mi355x.so (see include/tortoise_mi355x.h for the reference call sites).
#include "common.h"
#include <sched.h>
#include <cctype>
#include <algorithm>
#include <chrono>
#include <fstream>
#include <new>

namespace tts {
int ar_begin(tts_ctx *, const int32_t *, int, const float *, int, int);
int ar_prefill(tts_ctx *, float *);
int ar_step(tts_ctx *, const int32_t *, int, float *, int mode);
const int32_t *ar_host_lists(tts_ctx *);
const float *ar_fetch_logits_row(tts_ctx *, int);
int ar_batch(const tts_ctx *);
int ar_latents(tts_ctx *, const int32_t *, int, int, float *);
int ar_layers(const tts_ctx *);
float *ar_host_logits(tts_ctx *);
int diff_layers(const tts_ctx *);
int diff_forward(tts_ctx *, const float *, int, const float *, int, int, float *);
int diff_sample(tts_ctx *, const float *, const int32_t *, int, int, const float *, int, float *);
int voc_run(tts_ctx *, const float *, const int32_t *, int, const float *, int, float *);
} // namespace tts

using namespace tts;

hipEvent_t tts::prof_event(tts_ctx *c) {
  if (!c->ev_pool.empty()) { hipEvent_t e = c->ev_pool.back(); c->ev_pool.pop_back(); return e; }
  hipEvent_t e = nullptr;
  (void)hipEventCreate(&e);
  return e;
}

// No exception crosses the C ABI: allocation failures and anything unexpected become a status + tts_last_error text.
template <class F>
static int guarded(tts_ctx *c, F body) {
  try {
    return body();
  } catch (const std::bad_alloc &) {
    return fail(c, TTS_ERR_LIMIT, "out of host memory");
  } catch (const std::exception &e) {
    return fail(c, TTS_ERR_STATE, "internal error: %s", e.what());
  } catch (...) {
    return fail(c, TTS_ERR_STATE, "internal error");
  }
}

extern "C" {

tts_ctx *tts_create(int device) {
  if (device == -1) { // host-only context: tokenizer / RNG / sampler; every device stage fails loudly
    tts_ctx *c = new tts_ctx();
    c->device = -1;
    c->seed_value = (uint32_t)std::chrono::duration_cast<std::chrono::milliseconds>(
                        std::chrono::system_clock::now().time_since_epoch()).count();
    c->generator.seed(c->seed_value);
    return c;
  }
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) return nullptr; // no CPU fallback
  if (hipSetDevice(device) != hipSuccess) return nullptr;
  tts_ctx *c = new tts_ctx();
  c->device = device;
  if (hipStreamCreate(&c->stream) != hipSuccess || hipStreamCreateWithFlags(&c->load_stream, hipStreamNonBlocking) != hipSuccess || hipEventCreate(&c->ev0) != hipSuccess ||
      hipEventCreate(&c->ev1) != hipSuccess) {
    delete c;
    return nullptr;
  }
  // the reference seeds with wall-clock ms unless --seed is given (main.cpp:39-47)
  c->seed_value = (uint32_t)std::chrono::duration_cast<std::chrono::milliseconds>(
                      std::chrono::system_clock::now().time_since_epoch()).count();
  c->generator.seed(c->seed_value);
  return c;
}

// NUMA node of the context's GPU and that node's CPU list (sysfs); -1 / "" when unknown
static int device_numa(const tts_ctx *c, std::string &cpulist) {
  cpulist.clear();
  if (!c || c->device < 0) return -1;
  char bus[64] = {};
  if (hipDeviceGetPCIBusId(bus, (int)sizeof bus, c->device) != hipSuccess) return -1;
  for (char *p = bus; *p; p++) *p = (char)tolower(*p);
  FILE *f = fopen((std::string("/sys/bus/pci/devices/") + bus + "/numa_node").c_str(), "r");
  int node = -1;
  if (f) { if (fscanf(f, "%d", &node) != 1) node = -1; fclose(f); }
  if (node < 0) return -1;
  f = fopen(("/sys/devices/system/node/node" + std::to_string(node) + "/cpulist").c_str(), "r");
  if (f) {
    char buf[1024] = {};
    if (fgets(buf, sizeof buf, f)) { cpulist = buf; while (!cpulist.empty() && isspace((unsigned char)cpulist.back())) cpulist.pop_back(); }
    fclose(f);
  }
  return node;
}
int tts_device_numa_node(const tts_ctx *c, char *out, int cap) {
  std::string cl;
  const int node = device_numa(c, cl);
  if (out && cap > 0) snprintf(out, (size_t)cap, "%s", cl.c_str());
  return node;
}
int tts_pin_to_device_numa_node(tts_ctx *c) {
  std::string cl;
  if (device_numa(c, cl) < 0 || cl.empty()) return 0;
  cpu_set_t set;
  CPU_ZERO(&set);
  int n = 0;
  for (size_t i = 0; i < cl.size();) { // "a-b,c,d-e"
    char *end = nullptr;
    const long a = strtol(cl.c_str() + i, &end, 10);
    long b = a;
    if (*end == '-') b = strtol(end + 1, &end, 10);
    for (long k = a; k <= b && k < CPU_SETSIZE; k++) { CPU_SET((int)k, &set); n++; }
    i = (size_t)(end - cl.c_str());
    if (i < cl.size() && cl[i] == ',') i++;
    else if (i < cl.size() && !isdigit((unsigned char)cl[i])) break;
  }
  if (n == 0 || sched_setaffinity(0, sizeof set, &set) != 0) return 0;
  if (c->sampler_pool) { sampler_pool_free(c->sampler_pool); c->sampler_pool = nullptr; } // its threads are re-created (inside the mask) at the next use
  return n;
}

void tts_destroy(tts_ctx *c) {
  if (!c) return;
  if (c->sampler_pool) sampler_pool_free(c->sampler_pool);
  if (c->device < 0) { delete c->tok; delete c; return; }
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  if (c->ar) ar_free(c->ar);
  if (c->diff) diff_free(c->diff);
  if (c->voc) voc_free(c->voc);
  if (c->clvp) clvp_free(c->clvp);
  if (c->venc) voice_enc_free(c->venc);
  if (c->dcond) diff_cond_enc_free(c->dcond);
  if (c->fp16_counts) (void)hipFree(c->fp16_counts);
  delete c->tok;
  for (auto &kv : c->prof)
    for (auto &pr : kv.second.pending) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
  for (hipEvent_t e : c->ev_pool) (void)hipEventDestroy(e);
  if (c->ev0) (void)hipEventDestroy(c->ev0);
  if (c->ev1) (void)hipEventDestroy(c->ev1);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  if (c->load_stream) (void)hipStreamDestroy(c->load_stream);
  delete c;
}

int tts_version(void) { return TTS_API_VERSION; }

const char *tts_last_error(const tts_ctx *c) { return c ? c->err.c_str() : "no context (no HIP device?)"; }

int tts_set_option(tts_ctx *c, const char *key, double value) {
  if (!c || !key) return TTS_ERR_ARG;
  std::string k(key);
  if (k == "gn_eps") c->gn_eps = (float)value;
  else if (k == "ggml_lut") c->ggml_lut = value != 0;
  else if (k.rfind("prof_only:", 0) == 0) { // value 1: add the family to the list of profiled families; 0: back to "all"
    if (value != 0) c->prof_filter.push_back(k.substr(10));
    else c->prof_filter.clear();
  }
  else if (k == "device_topk") c->device_topk = value != 0; // 1 default: see tts_ar_step_sample
  else if (k == "sampler_threads") { // worker threads for the per-candidate sampler scans (0 = run them on the caller)
    if (c->sampler_pool) { sampler_pool_free(c->sampler_pool); c->sampler_pool = nullptr; }
    c->sampler_threads = value < 0 ? -1 : (int)value;
  }
  else if (k == "share_uncond") c->share_uncond = value != 0;
  else if (k == "prof_stride") c->prof_stride = value < 1 ? 1 : (int)value;
  else if (k == "diff_graph") c->diff_graph = value != 0;
  else if (k == "dec_f32_mfma") c->dec_f32_mfma = value != 0;
  else if (k == "attn_f32") c->attn_f32 = value != 0;
  else if (k == "attn_proj_f16") c->attn_proj_f16 = value != 0;
  else if (k == "proj_dual_b") c->proj_dual_b = value != 0;
  else if (k == "lc_attn_f32") c->lc_attn_f32 = value != 0;
  else if (k == "latency_mode") c->latency_mode = value != 0;
  else if (k == "fp16_check") c->fp16_check = value != 0;
  else if (k == "rng_fast_normal") c->rng_fast_normal = value != 0;
  else if (k == "noise_pipeline") c->noise_pipeline = value != 0;
  else if (k == "load_device_pack") c->load_device_pack = value != 0;
  else if (k == "load_threads") c->load_threads = value < 0 ? 0 : value > 64 ? 64 : (int)value;
  else if (k == "attn_q64") c->attn_q64 = value < 0 ? 0 : value > 2 ? 2 : (int)value; // 0 never, 1 always, 2 auto (grids of at most one 128-query workgroup per CU)
  else if (k == "hoist_integrator") c->hoist_integrator = value < 0 ? 0 : (int)value; // 0 off, 1 on for small layouts, n > 1: on for layouts of at most n packed rows (A/B)
  else if (k == "attn_f32_drop") c->attn_f32_drop = (int)value & 7;
  else if (k == "ar_weights") {
    if (value != 0 && value != 1 && value != 2) return fail(c, TTS_ERR_ARG, "ar_weights: 0 (f32), 1 (fp16) or 2 (fp8 e4m3)");
    c->ar_weights = (int)value;
  }
  else if (k == "prof_eager_every") c->prof_eager_every = value < 1 ? 1 : (int)value;
  else if (k == "stream_cus") {
    // Partition of the chip between two contexts of one process (INTEGRATION.md "two-context pipeline"): value n > 0 re-creates this
    // context's stream on the n lowest CUs of every XCD, n < 0 on all BUT those, 0 on the whole chip again. The AR stage is a chain of
    // 151 short dependent kernels per decode step that uses a fraction of the chip and cannot be interleaved with another stream's
    // 1000-workgroup kernels (measured: its kernels then wait for the running GEMM to drain, 228 ms -> 1-1.9 s; stream priority
    // changes nothing); on its own CUs it runs undisturbed while the other context's diffusion stage has the rest.
    // hipExtStreamCreateWithCUMask on gfx950: bit i = XCD i % 8, CU slot i / 8 (tools/cu_mask_probe.hip); an XCD with no bit set is
    // unrestricted, so every XCD keeps at least one CU. Only before any model is loaded / graph captured on the old stream.
    if (c->device < 0) return fail(c, TTS_ERR_HIP, "host-only context: no stream");
    if (c->ar || c->diff || c->voc || c->clvp || c->venc || c->dcond) return fail(c, TTS_ERR_STATE, "stream_cus must be set before the models are loaded");
    (void)hipSetDevice(c->device);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, c->device) != hipSuccess) return fail(c, TTS_ERR_HIP, "hipGetDeviceProperties failed");
    const int xcds = 8, per_xcd = prop.multiProcessorCount / xcds, n = (int)(value < 0 ? -value : value);
    if (prop.multiProcessorCount % xcds || per_xcd > 32 || n >= per_xcd) return fail(c, TTS_ERR_ARG, "stream_cus %d: the device has %d CUs per XCD", (int)value, per_xcd);
    hipStream_t s = nullptr;
    if (n == 0) {
      if (hipStreamCreate(&s) != hipSuccess) return fail(c, TTS_ERR_HIP, "hipStreamCreate failed");
    } else {
      uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (int slot = 0; slot < per_xcd; slot++)
        if ((slot < n) == (value > 0))
          for (int x = 0; x < xcds; x++) { const int bit = slot * xcds + x; mask[bit >> 5] |= 1u << (bit & 31); }
      if (hipExtStreamCreateWithCUMask(&s, (uint32_t)((per_xcd * xcds + 31) / 32), mask) != hipSuccess)
        return fail(c, TTS_ERR_HIP, "hipExtStreamCreateWithCUMask failed");
    }
    (void)hipStreamSynchronize(c->stream);
    (void)hipStreamDestroy(c->stream);
    c->stream = s;
    c->stream_cus = (int)value;
  }
  else if (k == "rng_shard_offset") { if (value < 0) return fail(c, TTS_ERR_ARG, "rng_shard_offset < 0"); c->rng_shard_offset = (int)value; }
  else if (k == "rng_shard_total") { if (value < 0) return fail(c, TTS_ERR_ARG, "rng_shard_total < 0"); c->rng_shard_total = (int)value; }
  else return fail(c, TTS_ERR_ARG, "unknown option '%s'", key);
  return TTS_OK;
}

#define NEED_CTX(c)                                                                              \
  do {                                                                                         \
    if (!(c)) return TTS_ERR_ARG;                                                              \
    if ((c)->device < 0) return fail((c), TTS_ERR_HIP, "host-only context: no HIP device, no CPU fallback"); \
    (void)hipSetDevice((c)->device);                                                           \
  } while (0)

int tts_load_ar(tts_ctx *c, const char *path) { NEED_CTX(c); return guarded(c, [&] { return ar_load(c, path); }); }
int tts_load_diffusion(tts_ctx *c, const char *path) { NEED_CTX(c); return guarded(c, [&] { return diff_load(c, path); }); }
int tts_load_vocoder(tts_ctx *c, const char *path) { NEED_CTX(c); return guarded(c, [&] { return voc_load(c, path); }); }
int tts_load_diffusion_conditioning_encoder(tts_ctx *c, const char *path) { NEED_CTX(c); return guarded(c, [&] { return diff_cond_enc_load(c, path); }); }
int tts_diffusion_conditioning_latent(tts_ctx *c, const float *mel, const int32_t *frames, int n_clips, float *out2048) {
  NEED_CTX(c);
  return guarded(c, [&] { return diff_cond_enc_latent(c, mel, frames, n_clips, out2048); });
}
int tts_set_diffusion_conditioning_latent(tts_ctx *c, const float *latent2048) { NEED_CTX(c); return guarded(c, [&] { return diff_set_cond_latent(c, latent2048); }); }
int tts_load_voice_encoder(tts_ctx *c, const char *path) { NEED_CTX(c); return guarded(c, [&] { return voice_enc_load(c, path); }); }
int tts_voice_latent(tts_ctx *c, const float *mel, const int32_t *frames, int n_clips, float *out1024) {
  NEED_CTX(c);
  return guarded(c, [&] { return voice_enc_latent(c, mel, frames, n_clips, out1024); });
}
int tts_load_clvp(tts_ctx *c, const char *path) { NEED_CTX(c); return guarded(c, [&] { return clvp_load(c, path); }); }
int tts_clvp_score(tts_ctx *c, const int32_t *text_ids, int n_text, const int32_t *codes, const int32_t *code_len, int n_candidates,
                   int code_stride, float *scores_out) {
  NEED_CTX(c);
  return guarded(c, [&] { return clvp_score(c, text_ids, n_text, codes, code_len, n_candidates, code_stride, scores_out); });
}
int tts_ar_layers(const tts_ctx *c) { return c ? ar_layers(c) : 0; }
int tts_diffusion_layers(const tts_ctx *c) { return c ? diff_layers(c) : 0; }

void tts_seed(tts_ctx *c, uint32_t seed) {
  if (!c) return;
  c->seed_value = seed;
  c->generator.seed(seed);
  c->distribution.reset();
  c->normal_distribution.reset();
}
int tts_rng_load_state(tts_ctx *c, const char *path) {
  if (!c) return TTS_ERR_ARG;
  std::ifstream fin(path);
  if (!fin) return fail(c, TTS_ERR_IO, "cannot open '%s'", path);
  fin >> c->generator;
  c->distribution.reset();
  c->normal_distribution.reset();
  return fin ? TTS_OK : fail(c, TTS_ERR_FORMAT, "bad RNG state file '%s'", path);
}
int tts_rng_save_state(tts_ctx *c, const char *path) {
  if (!c || !path) return TTS_ERR_ARG;
  std::ofstream fout(path);
  if (!fout) return fail(c, TTS_ERR_IO, "cannot open '%s' for writing", path);
  fout << c->generator;
  return fout ? TTS_OK : fail(c, TTS_ERR_IO, "cannot write '%s'", path);
}
float tts_rng_uniform(tts_ctx *c) { return c ? c->distribution(c->generator) : 0.f; }
void tts_rng_normal(tts_ctx *c, float *out, int64_t n) { // sample_normal_noise, main.cpp:4695-4701
  if (!c || !out || n <= 0) return;
  rng_normal_fill(c, out, n);
}

int tts_tokenizer_load(tts_ctx *c, const char *path) {
  if (!c || !path) return TTS_ERR_ARG;
  return guarded(c, [&] {
    std::unique_ptr<Tokenizer> t(new Tokenizer());
    if (!t->load(path)) return fail(c, TTS_ERR_IO, "Failed to open %s (missing, truncated or malformed)", path);
    delete c->tok;
    c->tok = t.release();
    return (int)c->tok->vocab.size();
  });
}
int tts_tokenize(tts_ctx *c, const char *message, int32_t *out, int cap) {
  if (!c || !c->tok) return c ? fail(c, TTS_ERR_STATE, "tokenizer not loaded") : TTS_ERR_ARG;
  if (!message || !out || cap < 0) return fail(c, TTS_ERR_ARG, "tts_tokenize: bad argument");
  return guarded(c, [&] {
    std::vector<int> ids = c->tok->encode(message);
    for (int i = 0; i < (int)ids.size() && i < cap; i++) out[i] = ids[i];
    return (int)ids.size();
  });
}

int tts_ar_begin(tts_ctx *c, const int32_t *ids, int n, const float *voice, int B, int max_steps) {
  NEED_CTX(c);
  return guarded(c, [&] { return ar_begin(c, ids, n, voice, B, max_steps); });
}
int tts_ar_prefill(tts_ctx *c, float *logits) { NEED_CTX(c); return guarded(c, [&] { return ar_prefill(c, logits); }); }
int tts_ar_step(tts_ctx *c, const int32_t *prev, int i, float *logits) {
  NEED_CTX(c);
  return guarded(c, [&] { return ar_step(c, prev, i, logits, 0); });
}
// decode step + sampler in one call: the device prefilter's lists cross PCIe instead of the logits (ar.hip: sample_prefilter_kernel)
static int step_sample(tts_ctx *c, const int32_t *prev, int i, bool mask_stop, int32_t *out, int *fallbacks) {
  if (int rc = ar_step(c, prev, i, nullptr, mask_stop ? 2 : 1)) return rc;
  const int B = ar_batch(c);
  if (sample_candidates_list(c, ar_host_lists(c), prev, 1, B, out, [&](int b) { return ar_fetch_logits_row(c, b); }, fallbacks))
    return fail(c, TTS_ERR_HIP, "tts_ar_step_sample: fetching a logits row failed");
  return TTS_OK;
}
int tts_ar_step_sample(tts_ctx *c, const int32_t *prev, int i, unsigned flags, int32_t *samples_out) {
  NEED_CTX(c);
  if (!prev || !samples_out) return TTS_ERR_ARG;
  return guarded(c, [&] {
    if (ar_batch(c) < 1) return fail(c, TTS_ERR_STATE, "tts_ar_begin not called"); // before the shard check: with no AR state the batch is 0
    if (int rc = shard_check(c, ar_batch(c))) return rc;
    c->topk_fallbacks = 0;
    return step_sample(c, prev, i, (flags & TTS_AR_MASK_STOP) != 0, samples_out, &c->topk_fallbacks);
  });
}
int tts_ar_topk_fallbacks(const tts_ctx *c) { return c ? c->topk_fallbacks : -1; }
int tts_diffusion_time_mlp_retries(const tts_ctx *c) { return c ? c->time_mlp_retries : -1; }
int tts_diffusion_fp16_check(tts_ctx *c, int64_t counts[2]) {
  if (!c || !counts) return TTS_ERR_ARG;
  return guarded(c, [&] { return tts::diff_fp16_check(c, counts); });
}
int tts_host_sample_row(const float *row, const int32_t *ids, int ids_per_cand, float uniform) {
  if (!row || !ids || ids_per_cand < 1) return -1;
  return sample_one_row(row, ids, ids_per_cand, uniform);
}
int tts_host_sample_prefiltered(const float *row, const int32_t *ids, int ids_per_cand, float uniform, int keep) {
  if (!row || !ids || ids_per_cand < 1 || keep < 1 || keep > TTS_PF_MAX) return -2;
  std::vector<int32_t> list(TTS_PF_WORDS);
  if (host_prefilter_row(row, keep, list.data()) < 0) return -1;
  return sample_one_from_list(list.data(), ids, ids_per_cand, uniform);
}
int tts_ar_latents(tts_ctx *c, const int32_t *codes, int nb, int n_mel, float *out) {
  NEED_CTX(c);
  return guarded(c, [&] { return ar_latents(c, codes, nb, n_mel, out); });
}
int tts_sample(tts_ctx *c, const float *logits, const int32_t *ids, int ids_per_cand, int B, int32_t *out) {
  if (!c || !logits || !ids || !out || B < 1 || ids_per_cand < 1) return TTS_ERR_ARG;
  for (int i = 0; i < B * ids_per_cand; i++)
    if (ids[i] < 0 || ids[i] >= TTS_VOCAB_MEL) return fail(c, TTS_ERR_ARG, "penalty id out of range");
  if (int rc = shard_check(c, B)) return rc;
  return guarded(c, [&] { sample_candidates(c, logits, ids, ids_per_cand, B, out); return (int)TTS_OK; });
}

// autoregressive(), main.cpp:5042-5367.
static int autoregressive_impl(tts_ctx *c, const int32_t *text_ids, int n_text, const float *voice, int B, int max_steps,
                               unsigned flags, int32_t *codes_out, int32_t *rows_out, float *latents_out, int32_t *steps_out) {
  static const bool timing = getenv("TTS_TIMING") != nullptr; // developer aid: host-side breakdown of the stage on stderr
  auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double t_begin = now(), t_sample = 0, t_step = 0;
  if (!codes_out || !rows_out) return fail(c, TTS_ERR_ARG, "tts_autoregressive: null output");
  if (max_steps > 500) return fail(c, TTS_ERR_LIMIT, "max_steps %d exceeds the 500 codes apply_padding accepts", max_steps);
  // The stop schedule (tts_ar_set_stop_schedule: a bench / test device, not a reference feature) is checked BEFORE any device work, applies only to calls that pass
  // TTS_AR_MASK_STOP | TTS_AR_RETIRE (the combination it is documented for: a strict call is never truncated by a forgotten schedule) and says so on stderr once.
  const bool sched = !c->stop_schedule.empty() && (flags & TTS_AR_MASK_STOP) && (flags & TTS_AR_RETIRE);
  if (!c->stop_schedule.empty() && !sched) {
    static bool warned = false;
    if (!warned) { fprintf(stderr, "tts_autoregressive: a stop schedule is set but the call does not pass TTS_AR_MASK_STOP | TTS_AR_RETIRE: ignored\n"); warned = true; }
  }
  if (sched && (int)c->stop_schedule.size() != B) return fail(c, TTS_ERR_ARG, "tts_autoregressive: the stop schedule holds %d candidates, the call %d", (int)c->stop_schedule.size(), B);
  int rc = ar_begin(c, text_ids, n_text, voice, B, max_steps);
  if (rc) return rc;
  const int V = TTS_VOCAB_MEL;
  std::vector<float> logits0((size_t)B * V);
  const double t_after_begin = now();
  if ((rc = ar_prefill(c, logits0.data()))) return rc;
  float *logits = logits0.data(); // after the first step: the pinned buffer the decode graph copies into (no extra host copy)
  const double t_after_prefill = now();
  // mel_transformer_inputs_vector: [1 ... 1, 8192] per candidate at step 0 (5095-5105), afterwards
  // the previous samples (5208-5217).
  const int P = n_text + 2;
  std::vector<int32_t> ids((size_t)P * B, 1);
  for (int b = 0; b < B; b++) ids[(size_t)b * P + P - 1] = 8192;
  int ids_per_cand = P;
  std::vector<std::vector<int>> seq(B);
  std::vector<int32_t> samples(B);
  // Stop rule. Reference (strict, always for B == 1): the loop ends only in an iteration where ALL B samples are 8193
  // (main.cpp:5214-5222), a candidate's sequence freezes at its first 8193 (5210-5213). TTS_AR_RETIRE (throughput mode,
  // SURVEY 8e): a candidate retires at its first 8193 — from then on its input token is forced to 8193 and its samples are
  // ignored — the loop ends when every candidate has retired, and reaching max_steps pads and returns instead of failing.
  // The uniforms are consumed exactly as in strict mode (two per candidate and step, candidate order), so every
  // sequence is the one strict mode would have produced.
  const bool retire = (flags & TTS_AR_RETIRE) != 0;
  if ((rc = shard_check(c, B))) return rc;
  std::vector<char> done(B, 0);
  std::vector<int32_t> next; // the samples of the coming iteration when the device-top-k step already produced them
  bool have_next = false;
  c->topk_fallbacks = 0;
  int i = 0;
  for (;;) {
    double t0 = now();
    if (have_next) { samples = next; have_next = false; }
    else {
      if (flags & TTS_AR_MASK_STOP)
        for (int b = 0; b < B; b++) logits[(size_t)b * V + 8193] = -1e30f;
      sample_candidates(c, logits, ids.data(), ids_per_cand, B, samples.data());
      t_sample += now() - t0;
    }
    int stops = 0;
    for (int b = 0; b < B; b++) {
      if (sched && i == c->stop_schedule[b]) samples[b] = 8193; // tts_ar_set_stop_schedule
      if (retire && done[b]) { samples[b] = 8193; stops++; continue; }
      if (!(seq[b].size() > 0 && seq[b].back() == 8193)) seq[b].push_back(samples[b]);
      if (samples[b] == 8193) { stops++; done[b] = 1; }
    }
    ids.assign(samples.begin(), samples.end());
    ids_per_cand = 1;
    i++;
    if (stops == B) break;
    if (i >= max_steps) {
      if ((flags & TTS_AR_MASK_STOP) || retire) break;
      return fail(c, TTS_ERR_LIMIT, "no stop token within %d steps", max_steps);
    }
    t0 = now();
    if (!c->device_topk) {
      if ((rc = ar_step(c, samples.data(), i - 1, nullptr, 0))) return rc;
      logits = ar_host_logits(c);
      t_step += now() - t0;
      continue;
    }
    // device top-k: the step hands back each candidate's ~64..128 largest logits and the sampler runs on those (same uniforms, same ids)
    if ((rc = ar_step(c, samples.data(), i - 1, nullptr, (flags & TTS_AR_MASK_STOP) ? 2 : 1))) return rc;
    t_step += now() - t0;
    t0 = now();
    next.resize(B);
    if (sample_candidates_list(c, ar_host_lists(c), ids.data(), 1, B, next.data(), [&](int b) { return ar_fetch_logits_row(c, b); }, &c->topk_fallbacks,
                               retire ? done.data() : nullptr))
      return fail(c, TTS_ERR_HIP, "tts_autoregressive: fetching a logits row failed");
    t_sample += now() - t0;
    have_next = true;
  }
  const double t_after_loop = now();
  if (steps_out) *steps_out = i;
  c->ar_stopped.assign(B, 0); // who was cut at max_steps (TTS_AR_RETIRE / TTS_AR_MASK_STOP): tts_ar_stop_status
  for (int b = 0; b < B; b++) c->ar_stopped[b] = (!seq[b].empty() && seq[b].back() == 8193) ? 1 : 0;
  int max_rows = 0;
  for (int b = 0; b < B; b++) {
    if (seq[b].size() > 500) seq[b].resize(500); // the reference asserts (main.cpp:4517)
    pad_codes(seq[b]);
    std::copy(seq[b].begin(), seq[b].end(), codes_out + (size_t)b * 502);
    rows_out[b] = trimmed_latent_rows(codes_out + (size_t)b * 502);
    max_rows = std::max(max_rows, rows_out[b]);
  }
  if (!latents_out) return TTS_OK;
  // latent pass over the mel prefix that trim_latents keeps (causal: rows beyond it cannot matter)
  const int n_mel = std::min(502, max_rows + 1);
  const int n_out = std::min(500, n_mel);
  std::vector<float> lat((size_t)B * n_out * TTS_DMODEL);
  const double t_before_lat = now();
  if ((rc = ar_latents(c, codes_out, B, n_mel, lat.data()))) return rc;
  if (timing)
    fprintf(stderr, "[tts timing] AR: begin %.1f ms, prefill %.1f, loop %.1f (steps %.1f in %d, sampler %.1f), latents %.1f, total %.1f\n",
            t_after_begin - t_begin, t_after_prefill - t_after_begin, t_after_loop - t_after_prefill, t_step, i - 1, t_sample,
            now() - t_before_lat, now() - t_begin);
  size_t off = 0;
  for (int b = 0; b < B; b++) {
    std::copy(lat.begin() + (size_t)b * n_out * TTS_DMODEL, lat.begin() + ((size_t)b * n_out + rows_out[b]) * TTS_DMODEL,
              latents_out + off);
    off += (size_t)rows_out[b] * TTS_DMODEL;
  }
  return TTS_OK;
}

int tts_autoregressive(tts_ctx *c, const int32_t *text_ids, int n_text, const float *voice, int B, int max_steps,
                       unsigned flags, int32_t *codes_out, int32_t *rows_out, float *latents_out, int32_t *steps_out) {
  NEED_CTX(c);
  return guarded(c, [&] {
    return autoregressive_impl(c, text_ids, n_text, voice, B, max_steps, flags, codes_out, rows_out, latents_out, steps_out);
  });
}

int tts_ar_set_stop_schedule(tts_ctx *c, const int32_t *stop_at, int n_candidates) {
  if (!c || n_candidates < 0) return TTS_ERR_ARG;
  return guarded(c, [&]() -> int { // (vector::assign may throw: nothing crosses the C ABI)
    if (!stop_at || n_candidates == 0) { c->stop_schedule.clear(); return TTS_OK; }
    for (int b = 0; b < n_candidates; b++)
      if (stop_at[b] < 1) return fail(c, TTS_ERR_ARG, "tts_ar_set_stop_schedule: candidate %d would stop before its first code", b);
    c->stop_schedule.assign(stop_at, stop_at + n_candidates);
    return TTS_OK;
  });
}

int tts_ar_stop_status(tts_ctx *c, int32_t *stopped_out, int n_candidates) {
  if (!c || !stopped_out || n_candidates < 1) return TTS_ERR_ARG;
  if ((int)c->ar_stopped.size() != n_candidates) return fail(c, TTS_ERR_STATE, "tts_ar_stop_status: the last tts_autoregressive call had %d candidates", (int)c->ar_stopped.size());
  std::copy(c->ar_stopped.begin(), c->ar_stopped.end(), stopped_out);
  return TTS_OK;
}

int tts_diffusion_frames(int L) { return L * 4 * 24000 / 22050; }
int tts_diffusion_forward(tts_ctx *c, const float *latents, int L, const float *x_t, int timestep, int cond_free, float *out) {
  NEED_CTX(c);
  return guarded(c, [&] { return diff_forward(c, latents, L, x_t, timestep, cond_free, out); });
}
int tts_diffusion(tts_ctx *c, const float *latents, const int32_t *rows, int B, int n_steps, const float *noise,
                  int noise_mode, float *mel_out) {
  NEED_CTX(c);
  // the DEVICE noise streams are keyed by the global candidate id; host / reference noise does not look at the shard options
  if (B >= 1 && !noise && noise_mode == TTS_NOISE_DEVICE) if (int rc = shard_check(c, B)) return rc;
  return guarded(c, [&] { return diff_sample(c, latents, rows, B, n_steps, noise, noise_mode, mel_out); });
}
int tts_vocoder_samples(int T) { return (T + 10) * 256 - 6; }
int tts_vocoder(tts_ctx *c, const float *mel, const int32_t *frames, int B, const float *noise, int noise_mode, float *audio) {
  NEED_CTX(c);
  if (B >= 1 && !noise && noise_mode == TTS_NOISE_DEVICE) if (int rc = shard_check(c, B)) return rc;
  return guarded(c, [&] { return voc_run(c, mel, frames, B, noise, noise_mode, audio); });
}

// Streaming form of the vocoder for first-audio latency: the UnivNet stack is fully convolutional, so the samples of frames
// [frame0, frame0 + n_frames) depend only on mel / noise frames within a bounded halo (conv_pre k7, kernel predictor k5 + 6 x k3 + k3,
// transposed convs, the dilated 1/3/9/27 convs of the 8-sample stage ~ 6 frames, conv_post k7): a window of the sequence with
// VOC_HALO frames on either side reproduces them exactly; windows that touch a sequence end keep the reference's boundary treatment
// (reflect pad / zero pad / the 10 silent frames) because they coincide with it. The halo is tied to the loaded architecture in
// vocoder.hip (voc_receptive_frames: 20 frames for UnivNet c32, static_assert against TTS_VOC_CHUNK_HALO). Cost: every chunk evaluates
// up to 2 x 24 + 10 extra frames (a 100-frame chunk costs ~1.6x its share of the one-shot call).
int tts_vocoder_chunk(tts_ctx *c, const float *mel, int T, const float *noise, int frame0, int n_frames, float *audio_out,
                      int *n_samples_out) {
  NEED_CTX(c);
  if (!mel || !noise || !audio_out || T < 1 || n_frames < 1 || frame0 < 0 || frame0 >= T + 10)
    return fail(c, TTS_ERR_ARG, "tts_vocoder_chunk: bad argument");
  return guarded(c, [&] {
    const int Tm = T + 10, f1 = std::min(Tm, frame0 + n_frames);
    const int halo = voc_halo_frames();
    int w0 = std::max(0, frame0 - halo), w1 = f1 + halo;
    if (w1 >= T) w1 = Tm;                                   // the window reaches the silent pad frames: take the true end
    const int wt = (w1 == Tm ? T : w1) - w0;                // mel frames handed to the vocoder (it appends 10 pad frames itself)
    std::vector<float> wm((size_t)100 * wt), wn((size_t)64 * (wt + 10)), wa((size_t)(wt + 10) * 256);
    for (int ch = 0; ch < 100; ch++) memcpy(&wm[(size_t)ch * wt], mel + (size_t)ch * T + w0, (size_t)wt * 4);
    for (int ch = 0; ch < 64; ch++)
      for (int t = 0; t < wt + 10; t++) wn[(size_t)ch * (wt + 10) + t] = noise[(size_t)ch * Tm + std::min(w0 + t, Tm - 1)];
    const int32_t frames = wt;
    int rc = voc_run(c, wm.data(), &frames, 1, wn.data(), TTS_NOISE_REFERENCE, wa.data());
    if (rc) return rc;
    const int64_t total = (int64_t)Tm * 256 - 6, s0 = (int64_t)frame0 * 256, s1 = std::min<int64_t>((int64_t)f1 * 256, total);
    const int64_t n = std::max<int64_t>(0, s1 - s0);
    memcpy(audio_out, wa.data() + (s0 - (int64_t)w0 * 256), (size_t)n * 4);
    if (n_samples_out) *n_samples_out = (int)n;
    return (int)TTS_OK;
  });
}

static void prof_resolve(tts_ctx *c) {
  (void)hipStreamSynchronize(c->stream);
  for (auto &kv : c->prof) {
    for (auto &pr : kv.second.pending) {
      float ms = 0;
      if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) { kv.second.ms += ms; kv.second.launches++; }
      c->ev_pool.push_back(pr.first);
      c->ev_pool.push_back(pr.second);
    }
    kv.second.pending.clear();
  }
}
int tts_prof_reset(tts_ctx *c, int enable) {
  if (!c) return TTS_ERR_ARG;
  if (c->device >= 0) prof_resolve(c);
  c->prof.clear();
  c->prof_on = enable != 0;
  return TTS_OK;
}
int tts_prof_get(tts_ctx *c, const char *family, double *ms, int64_t *launches, double *work) {
  if (!c || !family) return TTS_ERR_ARG;
  if (c->device >= 0) prof_resolve(c);
  // a family name also names its sub-families ("diff_gemm" = diff_gemm_qkv + diff_gemm_k3 + ...: the GEMM launches are recorded per shape class)
  double tms = 0, tw = 0;
  int64_t tl = 0;
  const std::string f(family), pre = f + "_";
  for (auto &kv : c->prof)
    if (kv.first == f || kv.first.rfind(pre, 0) == 0) { tms += kv.second.ms; tl += kv.second.launches; tw += kv.second.work; }
  if (ms) *ms = tms;
  if (launches) *launches = tl;
  if (work) *work = tw;
  return TTS_OK;
}

} // extern "C"

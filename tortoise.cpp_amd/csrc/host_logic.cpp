// Host-side logic of the drop-in: weight container reader, tokenizer, sampler, diffusion schedule,
// sequence bookkeeping, WAV writer. Reference behaviour is cited per function (file:line in
// /root/reference); tests/test_host_parity.py checks each against the real reference code.
#include "common.h"
#include <unistd.h>
#include <iomanip>
#include <random>
#include <sstream>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <algorithm>
#include <cmath>
#include <fstream>
#include <functional>
#include <limits>
#include <regex>

namespace tts {

int fail(tts_ctx *ctx, int code, const char *fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (ctx) {
    static std::mutex mu; // load workers may fail at the same time (common.h: run_parallel)
    std::lock_guard<std::mutex> lk(mu);
    ctx->err = buf;
  }
  return code;
}

// ---------------------------------------------------------------------------------------------
// Reference-order normal noise in bulk. The reference draws every noise value with ONE call of std::normal_distribution<double>::operator()(std::mt19937 &)
// (main.cpp:4695-4701; 81 x 100 T values per utterance in diffusion(), 64 (T + 10) in vocoder()), ~38 ns each on the GPU box's host: 0.27 s for one utterance, more than
// the device needs for the whole sampling loop. libstdc++'s algorithm (bits/random.tcc; restated in oracle/orc_math.cpp and pinned there against the reference's compiled
// lines) is Marsaglia's polar method over generate_canonical<double, 53>:
//     u = (g() + g() * 2^32) / 2^64  (clamped below 1);   x = 2 u1 - 1, y = 2 u2 - 1, r2 = x x + y y, repeat while r2 > 1 or r2 == 0;
//     m = sqrt(-2 log(r2) / r2);   return y m now, x m at the next call;   every returned value v goes through v * stddev + mean.
// Only the generator calls and the accept test are sequential; log / sqrt / divide depend on one accepted pair each. rng_normal_fill runs the sequential part on the
// calling thread (the generator object itself: it ends in the state single draws leave it in) and the per-pair arithmetic on a few threads — the same operations in the
// same order per value, so every float equals the one operator() returns (tests/test_host_parity.py compares the two forms draw for draw, and the fast form against
// the reference's compiled lines). The distribution object's cached second value is read and written through its stream operators (the library's own state interface).
// ---------------------------------------------------------------------------------------------
namespace {
// (the reference's build has no fused multiply-add: contraction is switched off in every function below that multiplies and adds)
inline double canonical53(std::mt19937 &g) {
#pragma clang fp contract(off)
  const double lo = (double)g();
  const double hi = (double)g();
  double ret = (lo + hi * 4294967296.0) / 18446744073709551616.0;
  if (ret >= 1.0) ret = std::nextafter(1.0, 0.0);
  return ret;
}
struct PolarPair { double x, y, r2; };
inline void polar_finish(const PolarPair &p, double &first, double &second) {
#pragma clang fp contract(off)
  const double m = std::sqrt(-2 * std::log(p.r2) / p.r2);
  first = p.y * m;
  second = p.x * m;
}
inline float as_returned(double v) {
#pragma clang fp contract(off)
  return (float)(v * 1.0 + 0.0);
} // operator() ends with ret * stddev + mean (turns -0 into +0, like the original)
bool normal_saved(const std::normal_distribution<double> &d, double &saved) {
  std::ostringstream os;
  os << d; // "mean stddev saved_available [saved]" at max_digits10
  std::istringstream is(os.str());
  double mean = 0, sd = 0;
  int avail = 0;
  is >> mean >> sd >> avail;
  if (avail) is >> saved;
  return avail != 0;
}
void normal_set_saved(std::normal_distribution<double> &d, bool avail, double saved) {
  if (!avail) { d.reset(); return; }
  std::ostringstream os;
  os.precision(std::numeric_limits<double>::max_digits10);
  os << std::scientific << d.mean() << ' ' << d.stddev() << ' ' << 1 << ' ' << saved;
  std::istringstream is(os.str());
  is >> d;
}
} // namespace

void rng_normal_fill(tts_ctx *ctx, float *dst, int64_t n) {
#pragma clang fp contract(off)
  if (n <= 0) return;
  std::normal_distribution<double> &nd = ctx->normal_distribution;
  if (!ctx->rng_fast_normal || n < 4096 || nd.mean() != 0.0 || nd.stddev() != 1.0) {
    for (int64_t i = 0; i < n; i++) dst[i] = nd(ctx->generator);
    return;
  }
  // (nothing has touched the generator yet: if the pair buffer cannot be had, the single-draw form does the whole job)
  std::vector<PolarPair> pp;
  try {
    pp.resize((size_t)(n + 1) / 2);
  } catch (...) {
    for (int64_t i = 0; i < n; i++) dst[i] = nd(ctx->generator);
    return;
  }
  int64_t at = 0;
  double saved = 0;
  if (normal_saved(nd, saved)) dst[at++] = as_returned(saved); // the cached second value of an earlier pair comes first
  const int64_t left = n - at, pairs = (left + 1) / 2;
  std::mt19937 &g = ctx->generator;
  for (int64_t i = 0; i < pairs; i++) { // the sequential part: generator calls and the accept test
    double x, y, r2;
    do {
      x = 2.0 * canonical53(g) - 1.0;
      y = 2.0 * canonical53(g) - 1.0;
      r2 = x * x + y * y;
    } while (r2 > 1.0 || r2 == 0.0);
    pp[(size_t)i] = PolarPair{x, y, r2};
  }
  float *out = dst + at;
  double last_second = 0;
  auto finish = [&](int64_t a, int64_t b) {
#pragma clang loop vectorize(disable) // scalar libm log / sqrt, as in the original
    for (int64_t i = a; i < b; i++) {
      double v0, v1;
      polar_finish(pp[(size_t)i], v0, v1);
      out[2 * i] = as_returned(v0);
      if (2 * i + 1 < left) out[2 * i + 1] = as_returned(v1);
      else last_second = v1; // only the last pair of an odd count gets here (one thread)
    }
  };
  const int nt = (int)std::min<int64_t>(std::min(8u, std::max(1u, std::thread::hardware_concurrency())), pairs / 8192);
  if (nt <= 1) finish(0, pairs);
  else {
    std::vector<std::thread> th;
    int started = 1; // ranges [pairs t / nt, pairs (t + 1) / nt): range 0 is this thread's; a thread that cannot be created leaves its range (and the rest) to this one
    try {
      th.reserve((size_t)nt);
      for (; started < nt; started++) th.emplace_back(finish, pairs * started / nt, pairs * (started + 1) / nt);
    } catch (...) {
    }
    finish(0, pairs / nt);
    if (started < nt) finish(pairs * started / nt, pairs);
    for (auto &t : th) t.join();
  }
  normal_set_saved(nd, (left & 1) != 0, last_second);
}

// ---------------------------------------------------------------------------------------------
// Weight container. Reference loaders: main.cpp:811-888 (AR), 1545-1625 (diffusion), 1932-2012
// (vocoder): u32 magic 'ggml', then {i32 n_dims, i32 name_len, i32 ttype, i32 ne[n_dims], name,
// data} until EOF. Only F32 (ttype 0) occurs in the published files.
// ---------------------------------------------------------------------------------------------
WeightFile::~WeightFile() { if (fd >= 0) close(fd); }
bool WeightFile::read_payload(const HostTensor &ht, void *dst) const {
  if (fd < 0 || ht.file_off < 0) return false;
  char *p = (char *)dst;
  size_t left = (size_t)ht.nelem() * sizeof(float);
  long long off = ht.file_off;
  while (left > 0) {
    const ssize_t g = pread(fd, p, left, (off_t)off);
    if (g <= 0) return false;
    p += g; off += g; left -= (size_t)g;
  }
  return true;
}

int read_weight_file(const char *path, WeightFile &out, std::string &err, size_t lazy_from) {
  // Two passes (round 6): the record headers are walked with seeks, then the tensor payloads (1.6 GB for the AR model) are read by a few threads with pread() —
  // the single fread() pass of rounds 1-5 spent as long zero-filling and copying as the page cache took to deliver.
  FILE *f = fopen(path, "rb");
  if (!f) { err = std::string("failed to open '") + path + "'"; return TTS_ERR_IO; }
  fseeko(f, 0, SEEK_END);
  const long long file_size = ftello(f);
  fseeko(f, 0, SEEK_SET);
  uint32_t magic = 0;
  if (fread(&magic, 4, 1, f) != 1 || magic != 0x67676d6cu) {
    fclose(f);
    err = std::string("invalid model file '") + path + "' (bad magic)";
    return TTS_ERR_FORMAT;
  }
  struct Pending { HostTensor *t; long long off; std::string name; };
  std::vector<Pending> pend;
  for (;;) {
    int32_t hdr[3];
    size_t got = fread(hdr, 4, 3, f);
    if (got == 0) break; // clean EOF
    if (got != 3) { fclose(f); err = "truncated record header"; return TTS_ERR_IO; }
    int n_dims = hdr[0], name_len = hdr[1], ttype = hdr[2];
    if (n_dims < 1 || n_dims > 4 || name_len < 1 || name_len > 1024) {
      fclose(f); err = "corrupt record header"; return TTS_ERR_FORMAT;
    }
    if (ttype != 0) { fclose(f); err = "unsupported tensor type (only F32)"; return TTS_ERR_FORMAT; }
    HostTensor t;
    t.n_dims = n_dims;
    for (int i = 0; i < n_dims; i++) {
      int32_t v;
      if (fread(&v, 4, 1, f) != 1 || v < 1) { fclose(f); err = "corrupt shape"; return TTS_ERR_FORMAT; }
      t.ne[i] = v;
    }
    std::string name(name_len, '\0');
    if (fread(&name[0], 1, name_len, f) != (size_t)name_len) { fclose(f); err = "truncated name"; return TTS_ERR_IO; }
    // a corrupt shape must not turn into a giant allocation (no exception may cross the C ABI)
    unsigned long long want = 4;
    for (int i = 0; i < n_dims; i++) {
      want *= (unsigned long long)t.ne[i];
      if (want > (unsigned long long)file_size) break;
    }
    const long long off = ftello(f);
    if (want > (unsigned long long)(file_size - off)) { fclose(f); err = "tensor '" + name + "' truncated"; return TTS_ERR_IO; }
    if (fseeko(f, (off_t)want, SEEK_CUR) != 0) { fclose(f); err = "tensor '" + name + "' truncated"; return TTS_ERR_IO; }
    t.file_off = off;
    auto ins = out.t.emplace(name, std::move(t)); // a repeated name keeps its first record, as the single-pass reader did
    if (ins.second && want < (unsigned long long)lazy_from) pend.push_back(Pending{&ins.first->second, off, std::move(name)});
    else if (ins.second && out.fd < 0) out.fd = dup(fileno(f));
  }
  const int fd = fileno(f);
  const int n = (int)pend.size();
  int nt = (int)std::min(8u, std::max(1u, std::thread::hardware_concurrency()));
  nt = std::min(nt, std::max(1, n));
  std::atomic<int> next{0}, bad{-1}, oom{0};
  auto work = [&]() {
    for (;;) {
      const int i = next.fetch_add(1);
      if (i >= n || bad.load() >= 0 || oom.load()) break;
      Pending &p = pend[i];
      try {
        p.t->data.resize((size_t)p.t->nelem());
      } catch (...) { oom.store(1); break; }
      char *dst = (char *)p.t->data.data();
      size_t left = p.t->data.size() * sizeof(float);
      long long off = p.off;
      while (left > 0) {
        const ssize_t g = pread(fd, dst, left, (off_t)off);
        if (g <= 0) { int z = -1; bad.compare_exchange_strong(z, i); break; }
        dst += g; off += g; left -= (size_t)g;
      }
    }
  };
  std::vector<std::thread> th;
  for (int t = 1; t < nt; t++) th.emplace_back(work);
  work();
  for (auto &t : th) t.join();
  fclose(f);
  if (oom.load()) { err = "out of host memory"; return TTS_ERR_LIMIT; }
  if (bad.load() >= 0) { err = "tensor '" + pend[bad.load()].name + "' truncated"; return TTS_ERR_IO; }
  return TTS_OK;
}

// ---------------------------------------------------------------------------------------------
// Tokenizer. The reference scrapes tokenizer.json with a character state machine rather than a
// JSON parser (common.cpp:166-255) and the resulting map — quirks included, e.g. the first key of
// every nested object is swallowed — *is* the vocabulary, so the same machine is restated here.
// ---------------------------------------------------------------------------------------------
static void subst(std::string &s, const char *from, const char *to) {
  const size_t fl = strlen(from), tl = strlen(to);
  for (size_t p = s.find(from); p != std::string::npos; p = s.find(from, p + tl)) s.replace(p, fl, to);
}

bool Tokenizer::load(const char *path) {
  std::ifstream in(path);
  if (!in) return false;
  const std::string js((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
  vocab.clear();
  if (js.empty() || js[0] != '{') return true;
  enum { OUTSIDE, IN_KEY, IN_STRVAL } st = OUTSIDE;
  std::string key, val;
  const int n = (int)js.size();
  auto commit = [&]() {
    subst(key, "\\u0120", " ");
    subst(key, "\\u010a", "\n");
    subst(key, "\\\"", "\"");
    try { vocab[key] = std::stoi(val); } catch (...) { /* non-integer value: ignored */ }
    key.clear();
    val.clear();
    st = OUTSIDE;
  };
  for (int i = 1; i < n; ++i) {
    const char ch = js[i];
    if (st == OUTSIDE) {
      if (ch == '"') st = IN_KEY; // everything else between tokens is skipped
      continue;
    }
    std::string &cur = (st == IN_KEY) ? key : val;
    if (ch == '\\' && i + 1 < n) { // escapes are kept verbatim
      cur += ch;
      cur += js[++i];
      continue;
    }
    if (ch != '"') { cur += ch; continue; }
    if (st == IN_STRVAL) { commit(); continue; }
    // closing quote of a key: expect [spaces] ':' [spaces] value. Every scan is bounded: a truncated or malformed file
    // is a load failure (status + last_error), never a read past the buffer.
    ++i;
    while (i < n && js[i] == ' ') ++i;
    ++i;
    while (i < n && js[i] == ' ') ++i;
    if (i >= n) return false;
    if (js[i] == '"') { st = IN_STRVAL; continue; }
    while (i < n && js[i] != ',' && js[i] != '}') val += js[i++];
    if (i >= n) return false; // value runs into the end of the file
    commit();
  }
  return st == OUTSIDE; // a key or string value still open at EOF: truncated file
}

// gpt_split_words + greedy longest match (common.cpp:268-339); main.cpp:6559-6567 wraps the
// result with 255 ... 0 after replacing " " by "[SPACE]".
std::vector<int> Tokenizer::encode(const std::string &message) const {
  std::string text = message;
  subst(text, " ", "[SPACE]");
  static const std::regex word_re(
      R"(\[SPACE\]|\[UNK\]|\[STOP\]|'s|'t|'re|'ve|'m|'ll|'d| ?[[:alpha:]]+| ?[[:digit:]]+| ?[^\s\[\][:alpha:][:digit:]]+|\s+(?!\S)|\s+)");
  std::vector<int> ids{255};
  std::smatch m;
  while (std::regex_search(text, m, word_re)) {
    const std::string word = m.str(0);
    size_t i = 0;
    while (i < word.size()) {
      size_t len = word.size() - i;
      for (; len > 0; --len) {
        auto it = vocab.find(word.substr(i, len));
        if (it != vocab.end()) { ids.push_back(it->second); break; }
      }
      if (len == 0) {
        fprintf(stderr, "gpt_tokenize: unknown token '%s'\n", word.substr(i, 1).c_str());
        len = 1;
      }
      i += len;
    }
    text = m.suffix();
  }
  ids.push_back(0);
  return ids;
}

// ---------------------------------------------------------------------------------------------
// Sampler: process_logits_and_sample (main.cpp:4753-4806) and helpers (4562-4720).
// The reference sorts all 8194 logits three times per candidate; here only the top-k survivors
// are processed, in the same float operation order, so the sampled ids and the RNG consumption
// (two uniforms per candidate, the second used) are identical. Ties among survivors fall back to
// the literal formulation so std::sort's tie order is inherited rather than re-invented.
// ---------------------------------------------------------------------------------------------
static inline float exp_like_reference(float v) { return (float)::exp((double)v); } // exp(float) -> ::exp(double)

static int multinomial_literal(std::vector<float> &l, float sample) {
  const int V = (int)l.size();
  const float LOWEST = std::numeric_limits<float>::lowest();
  std::vector<std::pair<float, int>> pairs(V);
  for (int i = 0; i < V; i++) pairs[i] = {l[i], i};
  std::sort(pairs.begin(), pairs.end(),
            [](const std::pair<float, int> &a, const std::pair<float, int> &b) { return a.first < b.first; });
  std::vector<float> sl(V);
  float sum = 0;
  for (int i = 0; i < V; i++) { sl[i] = exp_like_reference(pairs[i].first); sum += sl[i]; }
  for (int i = 0; i < V; i++) sl[i] /= sum;
  for (int i = 1; i < V; i++) sl[i] += sl[i - 1];
  for (int i = 0; i < V - 1; i++)
    if (sl[i] <= 0.2) l[pairs[i].second] = LOWEST;
  sum = 0;
  for (int i = 0; i < V; i++) { l[i] = exp_like_reference(l[i]); sum += l[i]; }
  float cum = 0;
  for (int i = 0; i < V; i++) {
    cum += l[i] / sum;
    if (cum >= sample) return i;
  }
  return V - 1;
}

struct Surv { float v; int idx; float e; };

// Tail shared by the full-row scan and the device-prefiltered list: `s` = the top-k survivors in INDEX order with their tempered
// values. top-p over the ascending order, final softmax + multinomial in index order. Returns -1 when two survivors tie (the
// caller falls back to the literal formulation to inherit std::sort's tie order).
static int sample_survivors(std::vector<Surv> &s, std::vector<Surv> &asc, float sample) {
  const int V = TTS_VOCAB_MEL;
  const float LOWEST = std::numeric_limits<float>::lowest();
  asc = s;
  std::sort(asc.begin(), asc.end(), [](const Surv &a, const Surv &b) { return a.v < b.v; });
  for (size_t i = 1; i < asc.size(); i++)
    if (asc[i].v == asc[i - 1].v) return -1;
  // top-p over the ascending survivors (non-survivors contribute exp(lowest) = +0)
  float sum = 0;
  for (auto &a : asc) { a.e = exp_like_reference(a.v); sum += a.e; }
  float cum = 0;
  for (size_t i = 0; i < asc.size(); i++) {
    cum += asc[i].e / sum;
    if (i + 1 < asc.size() && cum <= 0.2)
      for (auto &b : s) if (b.idx == asc[i].idx) b.v = LOWEST; // cut
  }
  // final softmax + multinomial in index order
  sum = 0;
  for (auto &a : s) { a.e = (a.v == LOWEST) ? 0.f : exp_like_reference(a.v); sum += a.e; }
  if (!(0.0f < sample)) return 0; // cumulative(=0) >= sample already at index 0
  cum = 0;
  for (auto &a : s) {
    cum += a.e / sum;
    if (cum >= sample) return a.idx;
  }
  return V - 1;
}

// One candidate, given its uniform draw: pure function of its arguments (runs on the sampler pool's threads).
static int sample_one(const float *src, const int32_t *ids, int ids_per_cand, float sample) {
  const int V = TTS_VOCAB_MEL, TOPK = 50;
  const float LOWEST = std::numeric_limits<float>::lowest();
  const float temp = 0.8;
  thread_local std::vector<Surv> s, asc;
  thread_local std::vector<float> l; // only materialised for the (rare) literal fallback
  // gather -> apply_penalty(2.0) -> scatter touches at most a few distinct ids (the prompt-shaped
  // [1 ... 1, 8192] at step 0, the previous sample afterwards): keep them as overrides
  int pid[4]; float pval[4]; int np = 0;
  bool many = false;
  for (int j = 0; j < ids_per_cand; j++) {
    const int id = ids[j];
    bool seen = false;
    for (int q = 0; q < np; q++) seen |= (pid[q] == id);
    if (seen) continue;
    if (np == 4) { many = true; break; }
    const float g = src[id];
    pid[np] = id; pval[np] = (g < 0) ? g * 2.0f : g / 2.0f; np++;
  }
  auto literal = [&]() {
    l.assign(src, src + V);
    for (int j = 0; j < ids_per_cand; j++) {
      const int id = ids[j];
      const float g = src[id];
      l[id] = (g < 0) ? g * 2.0f : g / 2.0f;
    }
    for (int i = 0; i < V; i++) l[i] /= temp;
    std::vector<float> tmp(l);
    std::nth_element(tmp.begin(), tmp.begin() + (V - TOPK), tmp.end());
    const float kth = tmp[V - TOPK];
    for (int i = 0; i < V; i++) if (l[i] < kth) l[i] = LOWEST;
    return multinomial_literal(l, sample);
  };
  if (many) return literal();
  auto val = [&](int i) { for (int q = 0; q < np; q++) if (pid[q] == i) return pval[q]; return src[i]; };
  // k-th largest penalised logit: min-heap of the 50 largest seen so far (almost every element fails
  // the single compare against the heap minimum)
  float heap[TOPK];
  for (int i = 0; i < TOPK; i++) heap[i] = val(i);
  std::make_heap(heap, heap + TOPK, std::greater<float>());
  float hmin = heap[0];
  for (int i = TOPK; i < V; i++) {
    float x = src[i];
    if (x <= hmin) continue;              // (penalised values are <= their source unless negative*2, handled by val)
    x = val(i);
    if (x <= hmin) continue;
    std::pop_heap(heap, heap + TOPK, std::greater<float>());
    heap[TOPK - 1] = x;
    std::push_heap(heap, heap + TOPK, std::greater<float>());
    hmin = heap[0];
  }
  // a penalised NEGATIVE logit is x*2 < x, a positive one x/2 < x: overrides never exceed src, so the scan
  // above cannot miss them. Threshold on the tempered values: ties (also those created by the division's
  // rounding: at most two adjacent floats share a quotient) survive, as in val_where_below_thresh.
  const float kth = hmin / temp;
  float cut = hmin;
  for (int q = 0; q < 4; q++) cut = std::nextafter(cut, LOWEST);
  s.clear();
  for (int i = 0; i < V; i++) {
    if (src[i] < cut) continue;
    const float v = val(i) / temp;
    if (v >= kth) s.push_back({v, i, 0.f});
  }
  const int pick = sample_survivors(s, asc, sample);
  return pick < 0 ? literal() : pick; // a tie: inherit std::sort's tie order from the literal formulation
}

// The same candidate from the decode step's device prefilter (ar.hip: sample_prefilter_kernel): `n` (index, logit) pairs in index
// order holding EVERY logit >= the smallest one listed, 54 <= n. Returns the id sample_one would return on the full row, or -1 when
// that cannot be guaranteed from the list alone (the caller then fetches the row):
//   - penalised values never exceed their source (x * 2 < x < 0, x / 2 <= x otherwise), so an unlisted logit stays below
//     thr = min(list) after the penalty, while at least n - 4 >= 50 listed ones keep their value >= thr: the 50th largest penalised
//     value `hmin` is >= thr and is found among the listed ones;
//   - sample_one keeps i when src[i] >= cut (= hmin - 4 ulps) and val(i) / temp >= hmin / temp: with thr <= cut every such i is listed.
//     thr > cut (the ~14 logits between rank 50 and rank n within 4 ulps of each other) -> -1;
//   - more than 4 distinct penalty ids or a tie among the survivors need the literal formulation over the full row -> -1.
static int sample_one_list(int n, const int32_t *idx, const float *lv, const int32_t *ids, int ids_per_cand, float sample) {
  const int TOPK = 50;
  const float LOWEST = std::numeric_limits<float>::lowest();
  const float temp = 0.8;
  thread_local std::vector<Surv> s, asc;
  thread_local std::vector<float> pv;
  if (n < TOPK + 4) return -1;
  int pid[4]; int np = 0;
  for (int j = 0; j < ids_per_cand; j++) {
    bool seen = false;
    for (int q = 0; q < np; q++) seen |= (pid[q] == ids[j]);
    if (seen) continue;
    if (np == 4) return -1;
    pid[np++] = ids[j];
  }
  pv.resize(n);
  float thr = lv[0];
  for (int i = 0; i < n; i++) {
    const float g = lv[i];
    thr = std::min(thr, g);
    bool pen = false;
    for (int q = 0; q < np; q++) pen |= (pid[q] == idx[i]);
    pv[i] = pen ? ((g < 0) ? g * 2.0f : g / 2.0f) : g;
  }
  asc.resize(n); // scratch: the 50th largest penalised value
  for (int i = 0; i < n; i++) asc[i].v = pv[i];
  std::nth_element(asc.begin(), asc.begin() + (TOPK - 1), asc.end(), [](const Surv &a, const Surv &b) { return a.v > b.v; });
  const float hmin = asc[TOPK - 1].v;
  const float kth = hmin / temp;
  float cut = hmin;
  for (int q = 0; q < 4; q++) cut = std::nextafter(cut, LOWEST);
  if (!(thr <= cut)) return -1;
  s.clear();
  for (int i = 0; i < n; i++) {
    if (lv[i] < cut) continue;
    const float v = pv[i] / temp;
    if (v >= kth) s.push_back({v, idx[i], 0.f});
  }
  return sample_survivors(s, asc, sample);
}

// Small persistent pool for the per-candidate sampler work (16 independent scans of 8194 logits between two
// decode steps). Workers spin for a short while after each job — the decode loop calls back within ~1 ms — and
// sleep on a condition variable otherwise.
struct SamplerPool {
  std::vector<std::thread> th;
  std::mutex m;
  std::condition_variable cv;
  // (job generation << 32) | next item. Items are claimed by compare-exchange on the whole word, so a worker that is
  // still leaving job g can neither claim an item of job g+1 nor disturb its counters: its exchange fails as soon as the
  // generation has moved on. `remaining` and `n_items` are published BEFORE the ticket (release) and only read after a
  // ticket load (acquire) that showed the matching generation.
  std::atomic<uint64_t> ticket{0};
  std::atomic<int> remaining{0}, n_items{0};
  std::function<void(int)> fn;
  std::atomic<bool> stop{false};
  explicit SamplerPool(int n) {
    for (int i = 0; i < n; i++) th.emplace_back([this] { loop(); });
  }
  ~SamplerPool() {
    { std::lock_guard<std::mutex> lk(m); stop.store(true); ticket.fetch_add(1ull << 32, std::memory_order_release); }
    cv.notify_all();
    for (auto &t : th) t.join();
  }
  static uint32_t gen_of(uint64_t t) { return (uint32_t)(t >> 32); }
  void drain(uint32_t g) {
    for (;;) {
      uint64_t t = ticket.load(std::memory_order_acquire);
      if (gen_of(t) != g) return;                 // job g is over (or was never ours)
      const int i = (int)(uint32_t)t;
      if (i >= n_items.load(std::memory_order_relaxed)) return;
      if (!ticket.compare_exchange_weak(t, t + 1, std::memory_order_acq_rel)) continue;
      fn(i);                                      // job g cannot complete before this item does: fn is still job g's
      remaining.fetch_sub(1, std::memory_order_release);
    }
  }
  void loop() {
    uint32_t seen = 0;
    for (;;) {
      // spin up to 2 ms for the next job, then block
      auto t0 = std::chrono::steady_clock::now();
      while (gen_of(ticket.load(std::memory_order_acquire)) == seen) {
        if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(2000)) {
          std::unique_lock<std::mutex> lk(m);
          cv.wait(lk, [&] { return gen_of(ticket.load(std::memory_order_acquire)) != seen; });
          break;
        }
      }
      seen = gen_of(ticket.load(std::memory_order_acquire));
      if (stop.load()) return;
      drain(seen);
    }
  }
  void run(int n, std::function<void(int)> f) {
    uint32_t g;
    {
      std::lock_guard<std::mutex> lk(m);
      fn = std::move(f);
      n_items.store(n, std::memory_order_relaxed);
      remaining.store(n, std::memory_order_relaxed);
      g = gen_of(ticket.load(std::memory_order_relaxed)) + 1;
      ticket.store((uint64_t)g << 32, std::memory_order_release); // publishes fn / n_items / remaining, next item = 0
    }
    cv.notify_all();
    drain(g);
    while (remaining.load(std::memory_order_acquire) > 0) std::this_thread::yield();
  }
};
void sampler_pool_free(SamplerPool *p) { delete p; }

// The RNG is consumed in candidate order exactly as the reference does; the scans then run in parallel.
// Sharded batch (options "rng_shard_offset" / "rng_shard_total"): this context holds candidates
// [offset, offset + B) of a batch of `total`; the used uniform of (step s, global candidate c) is output 2 (s total + c) + 1
// of the one mt19937 stream, so the draws of the other ranks' candidates are skipped (each uniform is one 32-bit output).
static void draw_uniforms(tts_ctx *ctx, int B, std::vector<float> &samples) {
  const int total = ctx->rng_shard_total > 0 ? ctx->rng_shard_total : B, c0 = shard_base(ctx);
  samples.resize(B);
  if (c0 > 0) ctx->generator.discard(2ull * c0);
  for (int c = 0; c < B; c++) {
    float sample = ctx->distribution(ctx->generator); // first draw discarded (main.cpp:4708-4709)
    sample = ctx->distribution(ctx->generator);
    samples[c] = sample;
  }
  if (total - c0 - B > 0) ctx->generator.discard(2ull * (total - c0 - B));
}

static void run_on_pool(tts_ctx *ctx, int B, const std::function<void(int)> &one) {
  if (B < 4 || ctx->sampler_threads == 0) { for (int c = 0; c < B; c++) one(c); return; }
  if (!ctx->sampler_pool) {
    const int hw = (int)std::thread::hardware_concurrency();
    const int n = ctx->sampler_threads > 0 ? ctx->sampler_threads : std::max(1, std::min(7, hw - 1));
    ctx->sampler_pool = new SamplerPool(n);
  }
  ctx->sampler_pool->run(B, one);
}

void sample_candidates(tts_ctx *ctx, const float *logits, const int32_t *ids, int ids_per_cand, int B,
                       int32_t *out) {
  const int V = TTS_VOCAB_MEL;
  std::vector<float> samples;
  draw_uniforms(ctx, B, samples);
  run_on_pool(ctx, B, [&](int c) { out[c] = sample_one(logits + (size_t)c * V, ids + (size_t)c * ids_per_cand, ids_per_cand, samples[c]); });
}

// The decode loop's sampler over the device prefilter's lists (ar.hip: [B][TTS_PF_WORDS] = {n, 0, 0, 0, idx[128], logit[128]}), same
// uniforms, same ids. A candidate whose list cannot decide (n < 0: the device found no threshold keeping 64..128 logits; or
// sample_one_list's -1) is sampled from its full row, fetched through `full_row` (which applies the stop mask itself).
// `retired` (may be null): candidates whose sequence has ended (TTS_AR_RETIRE). Their uniforms are drawn like everybody's — the stream stays the reference's — but
// nothing is sampled for them (out = 8193): round 5's ragged bench pass spent 27 ms per utterance evaluating lists and fetching full logits rows for candidates whose
// sample the loop then threw away.
int sample_candidates_list(tts_ctx *ctx, const int32_t *lists, const int32_t *ids, int ids_per_cand, int B, int32_t *out,
                           const std::function<const float *(int)> &full_row, int *n_fallbacks, const char *retired) {
  std::vector<float> samples;
  draw_uniforms(ctx, B, samples);
  auto one = [&](int c) {
    if (retired && retired[c]) { out[c] = 8193; return; }
    const int32_t *l = lists + (size_t)c * TTS_PF_WORDS;
    const int n = l[0];
    out[c] = (n < 1 || n > TTS_PF_MAX) ? -1
                                       : sample_one_list(n, l + 4, (const float *)(l + 4 + TTS_PF_MAX), ids + (size_t)c * ids_per_cand, ids_per_cand, samples[c]);
  };
  if (B < 8 || ctx->sampler_threads == 0) { for (int c = 0; c < B; c++) one(c); } // ~2 us per list: the pool's wake-up costs more below 8
  else run_on_pool(ctx, B, one);
  int fb = 0;
  for (int c = 0; c < B; c++) {
    if (out[c] >= 0) continue;
    const float *row = full_row(c);
    if (!row) return -1;
    out[c] = sample_one(row, ids + (size_t)c * ids_per_cand, ids_per_cand, samples[c]);
    fb++;
  }
  if (n_fallbacks) *n_fallbacks += fb;
  return 0;
}

// Host restatement of the device prefilter for tests (tts_host_sample_prefiltered): the `keep` largest logits and every tie of the
// smallest of them, in index order; n = -1 when that exceeds the list capacity.
int host_prefilter_row(const float *row, int keep, int32_t *list) {
  const int V = TTS_VOCAB_MEL;
  std::vector<float> tmp(row, row + V);
  std::nth_element(tmp.begin(), tmp.begin() + (keep - 1), tmp.end(), std::greater<float>());
  const float thr = tmp[keep - 1];
  int n = 0;
  for (int i = 0; i < V; i++) n += row[i] >= thr;
  std::fill(list, list + TTS_PF_WORDS, 0);
  if (n > TTS_PF_MAX) { list[0] = -1; return -1; }
  list[0] = n;
  float *lv = (float *)(list + 4 + TTS_PF_MAX);
  int k = 0;
  for (int i = 0; i < V; i++)
    if (row[i] >= thr) { list[4 + k] = i; lv[k] = row[i]; k++; }
  return n;
}

int sample_one_row(const float *row, const int32_t *ids, int ids_per_cand, float uniform) { return sample_one(row, ids, ids_per_cand, uniform); }
int sample_one_from_list(const int32_t *list, const int32_t *ids, int ids_per_cand, float uniform) {
  const int n = list[0];
  if (n < 1 || n > TTS_PF_MAX) return -1;
  return sample_one_list(n, list + 4, (const float *)(list + 4 + TTS_PF_MAX), ids, ids_per_cand, uniform);
}

// apply_padding, main.cpp:4510-4532 (the 8139 is the reference's literal, not 8193)
void pad_codes(std::vector<int> &codes) {
  while (!codes.empty() && codes.back() == 8139) codes.pop_back();
  codes.resize(500, 83);
  codes[497] = 45;
  codes[498] = 45;
  codes[499] = 248;
  codes.push_back(8193);
  codes.insert(codes.begin(), 8192);
}

// trim_latents, main.cpp:4873-4915: rows kept until more than 8 consecutive 83s.
int trimmed_latent_rows(const int32_t *codes502) {
  int run = 0;
  for (int c = 0; c < 500; c++) {
    run = (codes502[1 + c] == 83) ? run + 1 : 0;
    if (run > 8) return c;
  }
  return 500;
}

// get_relative_position_buckets, main.cpp:4722-4749 (i = query, c = key)
int rel_bucket(int i, int c) {
  const int dist = std::abs(c - i);
  int b = (i < c) ? 16 : 0;
  if (dist < 8) return b + dist;
  int big = 8 + (int)(::log((double)(float(dist) / 8)) / ::log(64.0 / 8.0) * (16.0 - 8.0));
  return b + std::min(big, 15);
}

// generate_timestep_embedding, main.cpp:5496-5521 (dim 1024, max_period 10000; cos half first)
void timestep_embedding(int t, float *out) {
  for (int i = 0; i < 512; ++i) {
    float freq = ::exp(-::log((double)10000) * static_cast<float>(i) / 512);
    float arg = static_cast<float>(t) * freq;
    out[i] = (float)::cos((double)arg);
    out[512 + i] = (float)::sin((double)arg);
  }
}

// Schedule: main.cpp:5370-5493 + 5641-5716, then the per-step scalars of 5988-6015 in the exact
// types the reference uses (double tables, float at the point of use).
void DiffSchedule::build(int n_steps) {
  n = n_steps;
  timestep_map.resize(n);
  for (int i = 0; i < n; i++) timestep_map[i] = (int)std::lround((double)i * 3999.0 / (n - 1)); // == literal table for n=80
  const int NT = 4000;
  const double scale = 1000.0 / NT, beta_start = scale * 0.0001, beta_end = scale * 0.02;
  std::vector<double> acp4000(NT);
  double prod = 1.0;
  for (int i = 0; i < NT; ++i) {
    double beta = beta_start + i * (float)(beta_end - beta_start) / (NT - 1);
    double alpha = 1.0f - beta;
    prod = (i == 0) ? alpha : prod * alpha;
    acp4000[i] = prod;
  }
  std::vector<double> beta(n), acp(n), prev(n), pvar(n), plv(n);
  float last = 1.0; // float on purpose (main.cpp:5663)
  for (int k = 0; k < n; k++) {
    beta[k] = 1 - (acp4000[timestep_map[k]] / last);
    last = acp4000[timestep_map[k]];
  }
  prod = 1.0;
  for (int k = 0; k < n; k++) {
    double alpha = 1.0f - beta[k];
    prod = (k == 0) ? alpha : prod * alpha;
    acp[k] = prod;
    prev[k] = (k == 0) ? 1.0f : acp[k - 1];
  }
  for (int k = 0; k < n; k++) pvar[k] = beta[k] * (1.0 - prev[k]) / (1.0 - acp[k]);
  plv[0] = std::log(pvar[1]);
  for (int k = 1; k < n; k++) plv[k] = std::log(pvar[k]);
  max_log.resize(n); min_log.resize(n); cfk.resize(n); sqrt_recip.resize(n); sqrt_recipm1.resize(n);
  coef1.resize(n); coef2.resize(n);
  const float base_k = 2.0;
  for (int t = 0; t < n; t++) {
    max_log[t] = std::log(beta[t]);
    min_log[t] = plv[t];
    cfk[t] = base_k * (1 - (float)t / float(n));
    sqrt_recip[t] = std::sqrt(1.0f / acp[t]);
    sqrt_recipm1[t] = std::sqrt(1.0f / acp[t] - 1);
    coef1[t] = beta[t] * std::sqrt(prev[t]) / (1.0 - acp[t]);
    coef2[t] = (1.0 - prev[t]) * std::sqrt(1.0 - beta[t]) / (1.0 - acp[t]);
  }
}

} // namespace tts

// ---- host-logic probes: the host-side pieces of the stage drivers, callable without a GPU (tests/test_host_parity.py) ----
extern "C" int tts_host_schedule(int n_steps, int32_t *timestep_map, float *max_log, float *min_log, float *cfk, float *sqrt_recip,
                                 float *sqrt_recipm1, float *coef1, float *coef2) {
  if (n_steps < 2) return TTS_ERR_ARG;
  tts::DiffSchedule s;
  s.build(n_steps);
  for (int t = 0; t < n_steps; t++) {
    timestep_map[t] = s.timestep_map[t];
    max_log[t] = s.max_log[t]; min_log[t] = s.min_log[t]; cfk[t] = s.cfk[t];
    sqrt_recip[t] = s.sqrt_recip[t]; sqrt_recipm1[t] = s.sqrt_recipm1[t];
    coef1[t] = s.coef1[t]; coef2[t] = s.coef2[t];
  }
  return TTS_OK;
}
extern "C" void tts_host_timestep_embedding(int t, float *out1024) { tts::timestep_embedding(t, out1024); }
extern "C" int tts_host_rel_bucket(int query, int key) { return tts::rel_bucket(query, key); }
extern "C" int tts_host_pad_codes(const int32_t *codes, int n, int32_t *out502) {
  if (n < 0 || n > 500) return TTS_ERR_ARG;
  std::vector<int> v(codes, codes + n);
  tts::pad_codes(v);
  if (v.size() != 502) return TTS_ERR_LIMIT;
  std::copy(v.begin(), v.end(), out502);
  return TTS_OK;
}
extern "C" int tts_host_trimmed_rows(const int32_t *codes502) { return tts::trimmed_latent_rows(codes502); }

// ------------------------------------------------------------------------------------------------------------------------------
// Mel front-end of the two voice-conditioning encoders (upstream tortoise-tts; the reference has no audio INPUT path at all).
// STFT with a periodic Hann window, centre = true (reflect padding of n_fft / 2), frames = n / hop + 1; triangular mel filterbank with
// Slaney area normalisation on the Slaney ("librosa") or HTK mel scale.
//   TacotronSTFT(1024, 256, 1024, 100, 24000, 0, 12000) -> magnitude, librosa filterbank, log(clamp 1e-5), normalised to [-1, 1] with the
//     constants the vocoder driver de-normalises with (main.cpp:6044-6060)                                   = tts_host_mel_diffusion100 (normalize = 1; upstream feeds the conditioning encoder the UN-normalised log-mel: normalize = 0)
//   torchaudio MelSpectrogram(n_fft 1024, hop 256, power 2, sample_rate 22050, f_max 8000, n_mels 80, norm "slaney", mel_scale "htk"),
//     log(clamp 1e-5), divided by the per-band mel_norms of the upstream data directory (optional here)       = tts_host_mel_voice80
// ------------------------------------------------------------------------------------------------------------------------------
namespace tts {
static void fft_inplace(std::vector<double> &re, std::vector<double> &im) { // radix-2, size = power of two
  const size_t n = re.size();
  for (size_t i = 1, j = 0; i < n; i++) {
    size_t bit = n >> 1;
    for (; j & bit; bit >>= 1) j ^= bit;
    j ^= bit;
    if (i < j) { std::swap(re[i], re[j]); std::swap(im[i], im[j]); }
  }
  for (size_t len = 2; len <= n; len <<= 1) {
    const double ang = -2.0 * M_PI / (double)len;
    for (size_t i = 0; i < n; i += len)
      for (size_t k = 0; k < len / 2; k++) {
        const double wr = std::cos(ang * (double)k), wi = std::sin(ang * (double)k);
        const double ur = re[i + k], ui = im[i + k];
        const double vr = re[i + k + len / 2] * wr - im[i + k + len / 2] * wi, vi = re[i + k + len / 2] * wi + im[i + k + len / 2] * wr;
        re[i + k] = ur + vr; im[i + k] = ui + vi;
        re[i + k + len / 2] = ur - vr; im[i + k + len / 2] = ui - vi;
      }
  }
}
static double hz_to_mel(double f, bool htk) {
  if (htk) return 2595.0 * std::log10(1.0 + f / 700.0);
  const double f_sp = 200.0 / 3.0, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp, logstep = std::log(6.4) / 27.0;
  return f >= min_log_hz ? min_log_mel + std::log(f / min_log_hz) / logstep : f / f_sp;
}
static double mel_to_hz(double m, bool htk) {
  if (htk) return 700.0 * (std::pow(10.0, m / 2595.0) - 1.0);
  const double f_sp = 200.0 / 3.0, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp, logstep = std::log(6.4) / 27.0;
  return m >= min_log_mel ? min_log_hz * std::exp(logstep * (m - min_log_mel)) : f_sp * m;
}
// [n_mels][n_fft / 2 + 1], triangles between n_mels + 2 points equally spaced on the mel scale, each divided by half its width in Hz (Slaney)
static std::vector<double> mel_filterbank(int n_mels, int n_fft, double sr, double f_min, double f_max, bool htk) {
  const int nb = n_fft / 2 + 1;
  std::vector<double> pts(n_mels + 2), fb((size_t)n_mels * nb, 0.0);
  const double m0 = hz_to_mel(f_min, htk), m1 = hz_to_mel(f_max, htk);
  for (int i = 0; i < n_mels + 2; i++) pts[i] = mel_to_hz(m0 + (m1 - m0) * i / (n_mels + 1), htk);
  for (int m = 0; m < n_mels; m++) {
    const double lo = pts[m], ce = pts[m + 1], hi = pts[m + 2], enorm = 2.0 / (hi - lo);
    for (int k = 0; k < nb; k++) {
      const double f = sr * k / n_fft, up = (f - lo) / (ce - lo), down = (hi - f) / (hi - ce);
      fb[(size_t)m * nb + k] = std::max(0.0, std::min(up, down)) * enorm;
    }
  }
  return fb;
}
// mel_out [n_mels][frames]; power 1 = magnitude, 2 = power spectrum; returns the frame count (n / hop + 1), < 0 on bad arguments
static int mel_spectrogram(const float *audio, int64_t n, int n_fft, int hop, int n_mels, double sr, double f_min, double f_max, bool htk, int power,
                           std::vector<double> &mel_out) {
  if (!audio || n <= n_fft / 2 || (n_fft & (n_fft - 1)) || hop < 1 || n_mels < 1) return TTS_ERR_ARG; // reflect padding needs n > n_fft / 2
  const int nb = n_fft / 2 + 1, pad = n_fft / 2, frames = (int)(n / hop) + 1;
  const std::vector<double> fb = mel_filterbank(n_mels, n_fft, sr, f_min, f_max, htk);
  std::vector<double> win(n_fft), re(n_fft), im(n_fft), spec(nb);
  for (int i = 0; i < n_fft; i++) win[i] = 0.5 - 0.5 * std::cos(2.0 * M_PI * i / n_fft); // periodic Hann
  mel_out.assign((size_t)n_mels * frames, 0.0);
  for (int t = 0; t < frames; t++) {
    for (int i = 0; i < n_fft; i++) {
      int64_t j = (int64_t)t * hop + i - pad; // reflect (no edge repeat): -1 -> 1, n -> n - 2
      if (j < 0) j = -j;
      if (j >= n) j = 2 * (n - 1) - j;
      re[i] = (double)audio[j] * win[i]; im[i] = 0.0;
    }
    fft_inplace(re, im);
    for (int k = 0; k < nb; k++) { const double p = re[k] * re[k] + im[k] * im[k]; spec[k] = power == 2 ? p : std::sqrt(p); }
    for (int m = 0; m < n_mels; m++) {
      double a = 0;
      for (int k = 0; k < nb; k++) a += fb[(size_t)m * nb + k] * spec[k];
      mel_out[(size_t)m * frames + t] = a;
    }
  }
  return frames;
}
} // namespace tts
extern "C" int tts_host_mel_frames(int64_t n_samples) { return n_samples < 0 ? TTS_ERR_ARG : (int)(n_samples / 256) + 1; }
// normalize = 0: log(clamp(mel, 1e-5)) as upstream's get_conditioning_latents feeds the contextual_embedder (wav_to_univnet_mel(...,
// do_normalization=False)): the input of tts_diffusion_conditioning_latent. normalize = 1: mapped to [-1, 1] with the constants the vocoder
// driver de-normalises with (normalize_tacotron_mel; the inverse is main.cpp:6044-6060): the scale of the diffusion stage's OUTPUT.
extern "C" int tts_host_mel_diffusion100(const float *audio24k, int64_t n, int normalize, float *mel_out) {
  std::vector<double> mel;
  const int frames = tts::mel_spectrogram(audio24k, n, 1024, 256, 100, 24000.0, 0.0, 12000.0, /*htk=*/false, /*power=*/1, mel);
  if (frames < 0) return frames;
  const double mel_max = 2.3143386840820312, mel_min = -11.512925148010254;
  for (size_t i = 0; i < mel.size(); i++) {
    const double lm = std::log(std::max(mel[i], 1e-5));
    mel_out[i] = (float)(normalize ? 2.0 * ((lm - mel_min) / (mel_max - mel_min)) - 1.0 : lm);
  }
  return frames;
}
extern "C" int tts_host_mel_voice80(const float *audio22k, int64_t n, const float *mel_norms80, float *mel_out) {
  std::vector<double> mel;
  const int frames = tts::mel_spectrogram(audio22k, n, 1024, 256, 80, 22050.0, 0.0, 8000.0, /*htk=*/true, /*power=*/2, mel);
  if (frames < 0) return frames;
  for (int m = 0; m < 80; m++)
    for (int t = 0; t < frames; t++)
      mel_out[(size_t)m * frames + t] = (float)(std::log(std::max(mel[(size_t)m * frames + t], 1e-5)) / (mel_norms80 ? (double)mel_norms80[m] : 1.0));
  return frames;
}

// writeWav, main.cpp:4821-4868
extern "C" int tts_write_wav(const char *path, const float *samples, int64_t n, int sample_rate) {
  FILE *f = fopen(path, "wb");
  if (!f) return TTS_ERR_IO;
  const int32_t channels = 1, bits = 32;
  const int32_t byte_rate = sample_rate * channels * bits / 8, block_align = channels * bits / 8;
  const int32_t data_size = (int32_t)(n * sizeof(float)), file_size = 36 + data_size, fmt_size = 16;
  const int32_t format = 3;
  fwrite("RIFF", 1, 4, f); fwrite(&file_size, 4, 1, f); fwrite("WAVE", 1, 4, f);
  fwrite("fmt ", 1, 4, f); fwrite(&fmt_size, 4, 1, f);
  fwrite(&format, 2, 1, f); fwrite(&channels, 2, 1, f); fwrite(&sample_rate, 4, 1, f);
  fwrite(&byte_rate, 4, 1, f); fwrite(&block_align, 2, 1, f); fwrite(&bits, 2, 1, f);
  fwrite("data", 1, 4, f); fwrite(&data_size, 4, 1, f);
  fwrite(samples, sizeof(float), (size_t)n, f);
  fclose(f);
  return TTS_OK;
}

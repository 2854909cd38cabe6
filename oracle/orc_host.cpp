// TEST INFRASTRUCTURE — oracle restatement of the reference's host-side math.
// Pinned against oracle/_ref/libref.so (the real reference lines) by tests/test_oracle_host.py.
#include "orc_common.h"
#include "orc_host.h"
#include <algorithm>
#include <fstream>
#include <limits>
#include <regex>
#include <sstream>

namespace orc {

// ---------------------------------------------------------------------------------------------
// RNG. main.cpp:47-50: std::mt19937 + uniform_real_distribution<float>(0,1) +
// normal_distribution<double>(0,1). libstdc++ algorithms restated by hand:
//   mt19937: MT19937 (32-bit), init_genrand seeding.
//   uniform float: generate_canonical<float,24> = float(u32) / 2^32, clamped below 1.
//   normal double: Marsaglia polar on generate_canonical<double,53> (two u32 per double, low word
//   first); returns y*m first and caches x*m for the next call.
// ---------------------------------------------------------------------------------------------
void Rng::seed(uint32_t s) {
  mt[0] = s;
  for (int i = 1; i < 624; i++) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
  idx = 624;
  saved_available = false;
}

uint32_t Rng::next_u32() {
  if (idx >= 624) {
    for (int k = 0; k < 624; k++) {
      uint32_t y = (mt[k] & 0x80000000u) | (mt[(k + 1) % 624] & 0x7fffffffu);
      mt[k] = mt[(k + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    idx = 0;
  }
  uint32_t y = mt[idx++];
  y ^= (y >> 11);
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= (y >> 18);
  return y;
}

float Rng::uniform() {
  float sum = (float)next_u32();
  float ret = sum / 4294967296.0f;
  if (ret >= 1.0f) ret = std::nextafter(1.0f, 0.0f);
  return ret * (1.0f - 0.0f) + 0.0f;
}

double Rng::canonical_double() {
  double sum = 0, tmp = 1;
  for (int k = 0; k < 2; k++) {
    sum += (double)next_u32() * tmp;
    tmp *= 4294967296.0;
  }
  double ret = sum / tmp;
  if (ret >= 1.0) ret = std::nextafter(1.0, 0.0);
  return ret;
}

double Rng::normal() {
  double ret;
  if (saved_available) {
    saved_available = false;
    ret = saved;
  } else {
    double x, y, r2;
    do {
      x = 2.0 * canonical_double() - 1.0;
      y = 2.0 * canonical_double() - 1.0;
      r2 = x * x + y * y;
    } while (r2 > 1.0 || r2 == 0.0);
    double mult = std::sqrt(-2 * std::log(r2) / r2);
    saved = x * mult;
    saved_available = true;
    ret = y * mult;
  }
  return ret * 1.0 + 0.0;
}

// libstdc++ operator>>(istream, mersenne_twister_engine): 624 state words then the index.
bool Rng::load_state_text(const char *path) {
  std::ifstream fin(path);
  if (!fin) return false;
  for (int i = 0; i < 624; i++) {
    uint64_t v;
    fin >> v;
    mt[i] = (uint32_t)v;
  }
  uint64_t p;
  fin >> p;
  idx = (int)p;
  saved_available = false;
  return (bool)fin;
}

// ---------------------------------------------------------------------------------------------
// Sampler, main.cpp:4562-4720 + 4770-4802.
// ---------------------------------------------------------------------------------------------
static void softmax_inplace_ref(std::vector<float> &src) { // 4641-4654: no max subtraction
  float sum = 0;
  for (size_t i = 0; i < src.size(); i++) {
    src[i] = (float)::exp((double)src[i]); // unqualified exp(float) binds to ::exp(double) with <cmath> only
    sum += src[i];
  }
  for (size_t j = 0; j < src.size(); j++) src[j] /= sum;
}

void sample_batch(const float *logits, const int *ids, int ids_total, int B, Rng &rng,
                  int *out_samples, float *out_probs) {
  const int V = 8194;
  std::vector<float> work(logits, logits + (size_t)B * V);
  // gather -> penalty 2.0 -> scatter (4571-4613, 4770-4775)
  int seq_len = ids_total / B;
  std::vector<float> gathered(ids_total);
  for (int i = 0; i < ids_total; i++) gathered[i] = logits[(size_t)(i / seq_len) * V + ids[i]];
  for (int i = 0; i < ids_total; i++)
    gathered[i] = (gathered[i] < 0) ? gathered[i] * 2.0f : gathered[i] / 2.0f;
  for (int i = 0; i < ids_total; i++) work[(size_t)(i / seq_len) * V + ids[i]] = gathered[i];

  const float LOWEST = std::numeric_limits<float>::lowest();
  for (int c = 0; c < B; c++) {
    std::vector<float> l(work.begin() + (size_t)c * V, work.begin() + (size_t)(c + 1) * V);
    // temperature 0.8 (4615-4619, 4791)
    float temp = 0.8;
    for (auto &v : l) v /= temp;
    // top-k 50: threshold = k-th largest value, ties survive (4629-4639)
    {
      std::vector<float> s(l);
      std::sort(s.begin(), s.end());
      float kth = s[s.size() - 50];
      for (auto &v : l)
        if (v < kth) v = LOWEST;
    }
    // top-p (4656-4693): ascending sort, un-max-subtracted softmax, cumsum, cut <= 0.2 except
    // the last sorted element
    {
      std::vector<std::pair<float, int>> pairs;
      pairs.reserve(V);
      for (int i = 0; i < V; i++) pairs.push_back(std::make_pair(l[i], i));
      std::sort(pairs.begin(), pairs.end(),
                [](const std::pair<float, int> &a, const std::pair<float, int> &b) {
                  return a.first < b.first;
                });
      std::vector<float> sorted_logits(V);
      for (int i = 0; i < V; i++) sorted_logits[i] = pairs[i].first;
      softmax_inplace_ref(sorted_logits);
      for (int i = 1; i < V; i++) sorted_logits[i] += sorted_logits[i - 1];
      for (int i = 0; i < V - 1; i++)
        if (sorted_logits[i] <= 0.2) l[pairs[i].second] = LOWEST;
    }
    softmax_inplace_ref(l);
    // multinomial (4703-4720): two draws, second used
    float sample = rng.uniform();
    sample = rng.uniform();
    int pick = V - 1;
    float cumulative = 0;
    for (int i = 0; i < V; i++) {
      cumulative += l[i];
      if (cumulative >= sample) { pick = i; break; }
    }
    out_samples[c] = pick;
    if (out_probs) std::memcpy(out_probs + (size_t)c * V, l.data(), sizeof(float) * V);
  }
}

// ---------------------------------------------------------------------------------------------
// T5-style relative position buckets, main.cpp:4722-4749.
// ---------------------------------------------------------------------------------------------
int bucket_of(int i, int c) {
  int rel = std::abs(c - i);
  int b = (i < c) ? 16 : 0;
  int val_if_large = 8 + (int)(::log((double)(float(rel) / 8)) / ::log(64.0 / 8.0) * (16.0 - 8.0));
  if (val_if_large > 15) val_if_large = 15;
  return b + (rel < 8 ? rel : val_if_large);
}
void buckets(int len, int *out) {
  for (int i = 0; i < len; i++)
    for (int c = 0; c < len; c++) out[(size_t)i * len + c] = bucket_of(i, c);
}

// ---------------------------------------------------------------------------------------------
// Diffusion schedule, main.cpp:5370-5493 + 5641-5716.
// ---------------------------------------------------------------------------------------------
void Schedule::build(const std::vector<int> &timestep_map) {
  const int NT = 4000;
  double scale = 1000.0 / NT;
  double beta_start = scale * 0.0001, beta_end = scale * 0.02;
  std::vector<double> b4000(NT), acp4000(NT);
  for (int i = 0; i < NT; i++) b4000[i] = beta_start + i * (float)(beta_end - beta_start) / (NT - 1);
  double prod = 1.0;
  for (int i = 0; i < NT; i++) {
    double alpha = 1.0f - b4000[i];
    prod = (i == 0) ? alpha : prod * alpha;
    acp4000[i] = prod;
  }
  n = (int)timestep_map.size();
  betas.resize(n); acp.resize(n); acp_prev.resize(n); post_var.resize(n); post_logvar.resize(n);
  coef1.resize(n); coef2.resize(n); sqrt_recip.resize(n); sqrt_recipm1.resize(n);
  float last = 1.0;
  for (int k = 0; k < n; k++) {
    betas[k] = 1 - (acp4000[timestep_map[k]] / last);
    last = acp4000[timestep_map[k]];
  }
  prod = 1.0;
  for (int k = 0; k < n; k++) {
    double alpha = 1.0f - betas[k];
    prod = (k == 0) ? alpha : prod * alpha;
    acp[k] = prod;
  }
  acp_prev[0] = 1.0f;
  for (int k = 1; k < n; k++) acp_prev[k] = acp[k - 1];
  for (int k = 0; k < n; k++) {
    sqrt_recip[k] = std::sqrt(1.0f / acp[k]);
    sqrt_recipm1[k] = std::sqrt(1.0f / acp[k] - 1);
    post_var[k] = betas[k] * (1.0 - acp_prev[k]) / (1.0 - acp[k]);
    coef1[k] = betas[k] * std::sqrt(acp_prev[k]) / (1.0 - acp[k]);
    coef2[k] = (1.0 - acp_prev[k]) * std::sqrt(1.0 - betas[k]) / (1.0 - acp[k]);
  }
  post_logvar[0] = std::log(post_var[1]);
  for (int k = 1; k < n; k++) post_logvar[k] = std::log(post_var[k]);
}

std::vector<int> default_timestep_map(int steps) { // literal table 5641-5648 == round(i*3999/79)
  std::vector<int> m(steps);
  for (int i = 0; i < steps; i++) m[i] = (int)std::lround((double)i * 3999.0 / (steps - 1));
  return m;
}

// main.cpp:5496-5521
void timestep_embedding(int t, float *out) {
  const int dim = 1024, half = 512, max_period = 10000;
  for (int i = 0; i < half; i++) {
    float freq = ::exp(-::log((double)max_period) * static_cast<float>(i) / half);
    float arg = static_cast<float>(t) * freq;
    out[i] = (float)::cos((double)arg); // unqualified cos/sin(float) -> ::cos(double)
    out[half + i] = (float)::sin((double)arg);
  }
  (void)dim;
}

// One ancestral step, main.cpp:5970-6030 (t = n-1-diffusion_index).
void diffusion_update(const Schedule &s, int t, int n_steps, const float *out_cond,
                      const float *out_uncond, float *x, const float *noise, int T) {
  int N = 100 * T;
  float max_log = std::log(s.betas[t]);
  float min_log = s.post_logvar[t];
  float base_k = 2.0;
  float cfk = base_k * (1 - (float)(t) / float(n_steps));
  float sr = s.sqrt_recip[t], srm1 = s.sqrt_recipm1[t], c1 = s.coef1[t], c2 = s.coef2[t];
  for (int i = 0; i < N; i++) {
    // calculate_model_variance called with (min_log, max_log) swapped into (max_log, min_log)
    float frac = (out_cond[N + i] + 1) / 2;
    float model_log_variance = frac * min_log + (1 - frac) * max_log;
    float eps = (1 + cfk) * out_cond[i] - cfk * out_uncond[i];
    float x0 = sr * x[i] - srm1 * eps;
    if (x0 > 1.0) x0 = 1.0;
    if (x0 < -1.0) x0 = -1.0;
    float mean = c1 * x0 + c2 * x[i];
    if (t != 0) x[i] = mean + std::exp(0.5 * model_log_variance) * noise[i];
    else x[i] = mean;
  }
}

// ---------------------------------------------------------------------------------------------
// Sequence bookkeeping, main.cpp:4510-4532 and 4873-4915.
// ---------------------------------------------------------------------------------------------
void apply_padding(std::vector<int> &vec) {
  while (!vec.empty() && vec.back() == 8139) vec.pop_back(); // sic: 8139 (typo for 8193)
  for (size_t i = vec.size(); i < 500; ++i) vec.push_back(83);
  vec[vec.size() - 3] = 45;
  vec[vec.size() - 2] = 45;
  vec[vec.size() - 1] = 248;
  vec.push_back(8193);
  vec.insert(vec.begin(), 8192);
}

int trimmed_rows(const int *codes502) { // codes502 includes leading 8192 / trailing 8193
  int calm = 0, rows = 0;
  for (int c = 0; c < 500; c++) {
    if (codes502[1 + c] == 83) calm++;
    else calm = 0;
    if (calm > 8) break;
    rows++;
  }
  return rows;
}

// ---------------------------------------------------------------------------------------------
// Tokenizer: common.cpp:166-255 (json scrape), 268-339 (regex split + greedy longest match),
// main.cpp:6559-6567 (" " -> "[SPACE]", wrap 255 ... 0).
// ---------------------------------------------------------------------------------------------
static std::string replace_all(std::string s, const std::string &from, const std::string &to) {
  size_t pos = 0;
  while ((pos = s.find(from, pos)) != std::string::npos) {
    s.replace(pos, from.length(), to);
    pos += to.length();
  }
  return s;
}

bool Tokenizer::load(const char *path) {
  std::ifstream ifs(path);
  if (!ifs) return false;
  std::string json((std::istreambuf_iterator<char>(ifs)), (std::istreambuf_iterator<char>()));
  vocab.clear();
  if (json.empty() || json[0] != '{') return true;
  bool has_key = false, in_token = false;
  std::string str_key, str_val;
  int n = (int)json.size();
  for (int i = 1; i < n; ++i) {
    if (!in_token) {
      if (json[i] == ' ') continue;
      if (json[i] == '"') { in_token = true; continue; }
    } else {
      if (json[i] == '\\' && i + 1 < n) {
        if (!has_key) str_key += json[i]; else str_val += json[i];
        ++i;
      } else if (json[i] == '"') {
        if (!has_key) {
          has_key = true;
          ++i;
          while (json[i] == ' ') ++i;
          ++i; // ':'
          while (json[i] == ' ') ++i;
          if (json[i] != '\"') {
            while (json[i] != ',' && json[i] != '}') str_val += json[i++];
            has_key = false;
          } else {
            in_token = true;
            continue;
          }
        } else {
          has_key = false;
        }
        str_key = replace_all(str_key, "\\u0120", " ");
        str_key = replace_all(str_key, "\\u010a", "\n");
        str_key = replace_all(str_key, "\\\"", "\"");
        try { vocab[str_key] = std::stoi(str_val); } catch (...) {}
        str_key = "";
        str_val = "";
        in_token = false;
        continue;
      }
      if (!has_key) str_key += json[i]; else str_val += json[i];
    }
  }
  return true;
}

std::vector<int> Tokenizer::encode(const std::string &message) const {
  std::string text = replace_all(message, " ", "[SPACE]");
  std::vector<std::string> words;
  {
    static const std::regex re(
        R"(\[SPACE\]|\[UNK\]|\[STOP\]|'s|'t|'re|'ve|'m|'ll|'d| ?[[:alpha:]]+| ?[[:digit:]]+| ?[^\s\[\][:alpha:][:digit:]]+|\s+(?!\S)|\s+)");
    std::string str = text;
    std::smatch m;
    while (std::regex_search(str, m, re)) {
      for (auto x : m) words.push_back(x);
      str = m.suffix();
    }
  }
  std::vector<int> tokens;
  tokens.push_back(255);
  for (const auto &word : words) {
    for (int i = 0; i < (int)word.size();) {
      for (int j = (int)word.size() - 1; j >= i; j--) {
        auto it = vocab.find(word.substr(i, j - i + 1));
        if (it != vocab.end()) {
          tokens.push_back(it->second);
          i = j + 1;
          break;
        } else if (j == i) {
          i++;
        }
      }
    }
  }
  tokens.push_back(0);
  return tokens;
}

} // namespace orc

// ---------------------------------------------------------------------------------------------
// C ABI (ctypes)
// ---------------------------------------------------------------------------------------------
using namespace orc;
extern "C" {
void *orc_rng_new(uint32_t seed) { Rng *r = new Rng(); r->seed(seed); return r; }
void orc_rng_free(void *r) { delete (Rng *)r; }
void orc_rng_seed(void *r, uint32_t s) { ((Rng *)r)->seed(s); }
int orc_rng_load_state(void *r, const char *path) { return ((Rng *)r)->load_state_text(path) ? 0 : -1; }
uint32_t orc_rng_u32(void *r) { return ((Rng *)r)->next_u32(); }
float orc_rng_uniform(void *r) { return ((Rng *)r)->uniform(); }
void orc_rng_normal_fill(void *r, float *out, int64_t n) {
  Rng *g = (Rng *)r;
  for (int64_t i = 0; i < n; i++) out[i] = (float)g->normal();
}
void orc_sample(const float *logits, const int *ids, int ids_total, int B, void *rng,
                int *out_samples, float *out_probs) {
  sample_batch(logits, ids, ids_total, B, *(Rng *)rng, out_samples, out_probs);
}
void orc_buckets(int len, int *out) { buckets(len, out); }
void orc_timestep_embedding(int t, float *out) { timestep_embedding(t, out); }
void orc_schedule(const int *tm, int n, double *betas, double *acp, double *plv, double *c1,
                  double *c2, double *sr, double *srm1) {
  Schedule s;
  s.build(std::vector<int>(tm, tm + n));
  for (int k = 0; k < n; k++) {
    betas[k] = s.betas[k]; acp[k] = s.acp[k]; plv[k] = s.post_logvar[k];
    c1[k] = s.coef1[k]; c2[k] = s.coef2[k]; sr[k] = s.sqrt_recip[k]; srm1[k] = s.sqrt_recipm1[k];
  }
}
void orc_default_timestep_map(int steps, int *out) {
  std::vector<int> m = default_timestep_map(steps);
  std::memcpy(out, m.data(), sizeof(int) * steps);
}
void orc_diffusion_update(const int *tm, int n_steps, int t, const float *out_cond,
                          const float *out_uncond, float *x, const float *noise, int T) {
  Schedule s;
  s.build(std::vector<int>(tm, tm + n_steps));
  diffusion_update(s, t, n_steps, out_cond, out_uncond, x, noise, T);
}
void orc_apply_padding(const int *codes, int n, int *out502) {
  std::vector<int> v(codes, codes + n);
  apply_padding(v);
  std::memcpy(out502, v.data(), sizeof(int) * 502);
}
int orc_trimmed_rows(const int *codes502) { return trimmed_rows(codes502); }
void *orc_tokenizer_new(const char *json_path) {
  Tokenizer *t = new Tokenizer();
  if (!t->load(json_path)) { delete t; return nullptr; }
  return t;
}
void orc_tokenizer_free(void *t) { delete (Tokenizer *)t; }
int orc_tokenizer_vocab_size(void *t) { return (int)((Tokenizer *)t)->vocab.size(); }
int orc_tokenize(void *t, const char *msg, int *out, int cap) {
  std::vector<int> ids = ((Tokenizer *)t)->encode(msg);
  for (int i = 0; i < (int)ids.size() && i < cap; i++) out[i] = ids[i];
  return (int)ids.size();
}
void orc_set_flags(float gn_eps, int lut) { g_flags.gn_eps = gn_eps; g_flags.lut = lut; }
float orc_f16_round(float x) { return f16r(x); }
}

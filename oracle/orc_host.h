// TEST INFRASTRUCTURE — oracle host-side declarations (see orc_host.cpp).
#pragma once
#include <cstdint>
#include <map>
#include <string>
#include <vector>

namespace orc {

struct Rng {
  uint32_t mt[624];
  int idx = 624;
  bool saved_available = false;
  double saved = 0;
  void seed(uint32_t s);
  uint32_t next_u32();
  float uniform();
  double canonical_double();
  double normal();
  bool load_state_text(const char *path);
};

void sample_batch(const float *logits, const int *ids, int ids_total, int B, Rng &rng,
                  int *out_samples, float *out_probs);
int bucket_of(int i, int c);
void buckets(int len, int *out);

struct Schedule {
  int n = 0;
  std::vector<double> betas, acp, acp_prev, post_var, post_logvar, coef1, coef2, sqrt_recip,
      sqrt_recipm1;
  void build(const std::vector<int> &timestep_map);
};
std::vector<int> default_timestep_map(int steps);
void timestep_embedding(int t, float *out);
void diffusion_update(const Schedule &s, int t, int n_steps, const float *out_cond,
                      const float *out_uncond, float *x, const float *noise, int T);
void apply_padding(std::vector<int> &vec);
int trimmed_rows(const int *codes502);

struct Tokenizer {
  std::map<std::string, int> vocab;
  bool load(const char *path);
  std::vector<int> encode(const std::string &message) const;
};

} // namespace orc

// TEST INFRASTRUCTURE — not part of the product.
//
// C-ABI shim appended (by oracle/build_ref.sh) to the ggml-free line ranges of the
// reference's main.cpp, which are piped straight from /root/reference into the
// compiler (no reference source is written into this repo). Together with the
// reference's common.cpp this becomes oracle/_ref/libref.so: the REAL reference
// code for tokenizer, RNG, sampler, rel-pos buckets, diffusion schedule and
// sequence bookkeeping, used to pin oracle/ (the hand restatement) and to generate
// the golden vectors under tests/golden/ (tests/golden/make_golden.py).
//
// Everything below only *calls* reference functions in the same order the
// reference's drivers do (file:line cited per function); it adds no arithmetic.
//
// Symbols provided by the extracted ranges (main.cpp):
//   39-57    time_seed, generator, distribution, normal_distribution, localAssert
//   4510-4532 apply_padding
//   4562-4749 apply_penalty, gather, scatter, temp_inplace, top_k_inplace, top_p_inplace,
//             softmax_inplace, sample_normal_noise, multinomial, get_relative_position_buckets
//   4809-4817 replaceAll
//   4873-4915 trim_latents
//   5369-5612 schedule helpers, generate_timestep_embedding, calculate_model_variance, ...
// and by common.cpp: gpt_vocab_init, gpt_tokenize.

#include <sstream>

extern "C" {

// main.cpp:6546  generator.seed(std::stoi(argv[i + 1]));
void ref_seed(unsigned s) {
  generator.seed(s);
  distribution.reset();
  normal_distribution.reset();
}

// main.cpp:6256-6266 / 6471-6480: `fin >> generator` from a text state file.
int ref_load_rng_state(const char *path) {
  std::ifstream fin(path);
  if (!fin) return -1;
  fin >> generator;
  distribution.reset();
  normal_distribution.reset();
  return 0;
}

// raw engine output (for pinning the restated mt19937)
unsigned ref_raw_u32() { return (unsigned)generator(); }

// main.cpp:4708  distribution(generator)
float ref_uniform() { return distribution(generator); }

// main.cpp:4695-4701
void ref_normal_fill(float *out, int n) {
  std::vector<float> v = sample_normal_noise(n);
  std::memcpy(out, v.data(), sizeof(float) * n);
}

// Composition of main.cpp:4770-4802 (process_logits_and_sample minus the ggml fetch).
//   logits: [B][8194]; ids: [B][ids_per_cand] (mel_transformer_inputs_vector);
//   out_samples: [B]; out_probs (optional): [B][8194]
void ref_process_logits_and_sample(const float *logits, const int *ids, int ids_total, int B,
                                   int *out_samples, float *out_probs) {
  std::vector<float> next_token_logits_vector(logits, logits + (size_t)B * 8194);
  std::vector<int> mel_transformer_inputs_vector(ids, ids + ids_total);
  std::vector<float> gather_result =
      gather(next_token_logits_vector, mel_transformer_inputs_vector, B);
  gather_result = apply_penalty(gather_result, 2.0);
  std::vector<float> transformed =
      scatter(next_token_logits_vector, gather_result, mel_transformer_inputs_vector, B);
  for (int i = 0; i < B; i++) {
    std::vector<float> l(transformed.begin() + (size_t)i * 8194,
                         transformed.begin() + (size_t)(i + 1) * 8194);
    temp_inplace(l, 0.8);
    top_k_inplace(l, 50);
    top_p_inplace(l);
    softmax_inplace(l);
    out_samples[i] = multinomial(l);
    if (out_probs) std::memcpy(out_probs + (size_t)i * 8194, l.data(), sizeof(float) * 8194);
  }
}

// main.cpp:4722-4749
void ref_buckets(int len, int *out) {
  std::vector<int> b = get_relative_position_buckets(len);
  std::memcpy(out, b.data(), sizeof(int) * (size_t)len * len);
}

// main.cpp:5496-5521 with dim 1024, max_period 10000 (5815-5816)
void ref_timestep_embedding(int t, float *out) {
  std::vector<int> ts = {t};
  std::vector<float> e = generate_timestep_embedding(ts, 1024, 10000);
  std::memcpy(out, e.data(), sizeof(float) * 1024);
}

// Composition of main.cpp:5641-5716 for an arbitrary timestep_map (the reference hard-codes
// the 80-entry table; it equals round(i*3999/79)). All outputs have n entries (double).
void ref_schedule(const int *timestep_map, int n, double *betas, double *acp,
                  double *post_logvar_clipped, double *coef1, double *coef2,
                  double *sqrt_recip_acp, double *sqrt_recipm1_acp) {
  std::vector<double> beta_schedule(0);
  get_beta_schedule(4000, beta_schedule);
  std::vector<double> alpha_cumulative_products = get_alphas_cumulative_product(beta_schedule);
  float last_alpha_cumulative_product = 1.0;
  beta_schedule.clear();
  for (int k = 0; k < n; k++) {
    int i = timestep_map[k];
    beta_schedule.push_back(1 - (alpha_cumulative_products[i] / last_alpha_cumulative_product));
    last_alpha_cumulative_product = alpha_cumulative_products[i];
  }
  alpha_cumulative_products = get_alphas_cumulative_product(beta_schedule);
  std::vector<double> prev(alpha_cumulative_products.size());
  prev.front() = 1.0f;
  std::copy(alpha_cumulative_products.begin(), alpha_cumulative_products.end() - 1,
            prev.begin() + 1);
  std::vector<double> sra = sqrt(reciprocal(alpha_cumulative_products));
  std::vector<double> srm1 = sqrt(reciprocal_minus_one(alpha_cumulative_products));
  std::vector<double> pv, plv, c1, c2;
  calculate_posterior_variance(pv, beta_schedule, prev, alpha_cumulative_products);
  calculate_posterior_log_variance_clipped(plv, pv);
  calculate_posterior_mean_coef1(c1, beta_schedule, prev, alpha_cumulative_products);
  calculate_posterior_mean_coef2(c2, prev, alpha_cumulative_products, beta_schedule);
  for (int k = 0; k < n; k++) {
    betas[k] = beta_schedule[k];
    acp[k] = alpha_cumulative_products[k];
    post_logvar_clipped[k] = plv[k];
    coef1[k] = c1[k];
    coef2[k] = c2[k];
    sqrt_recip_acp[k] = sra[k];
    sqrt_recipm1_acp[k] = srm1[k];
  }
}

// One host update of the sampling loop, main.cpp:5970-6030, for t = 79 - diffusion_index given
// the schedule scalars (already cast the way the reference casts them: float arguments).
//   out_cond/out_uncond: [200*T] network outputs; x: [100*T] in/out; noise: [100*T]
void ref_diffusion_update(const float *out_cond, const float *out_uncond, float *x,
                          const float *noise_in, int T, float max_log, float min_log,
                          float conditioning_free_k, float sqrt_recip, float sqrt_recipm1,
                          float coef1, float coef2, int is_last) {
  int split_index = 100 * T;
  std::vector<float> model_output_means(out_cond, out_cond + split_index);
  std::vector<float> model_output_vars(out_cond + split_index, out_cond + 2 * split_index);
  std::vector<float> nc_means(out_uncond, out_uncond + split_index);
  std::vector<float> xv(x, x + split_index);
  std::vector<float> model_log_variance;
  // NB: the reference passes (min_log, max_log) into (max_log, min_log) parameters (5998-5999)
  calculate_model_variance(model_output_vars, model_log_variance, min_log, max_log);
  blend_output_with_unconditioned_output(model_output_means, nc_means, conditioning_free_k);
  std::vector<float> x_start_pred =
      predict_xstart_from_eps(sqrt_recip, sqrt_recipm1, xv, model_output_means);
  std::vector<float> final_model_mean = q_posterior_mean(coef1, coef2, xv, x_start_pred);
  std::vector<float> sample_noise(noise_in, noise_in + split_index);
  std::vector<float> model_sample;
  if (!is_last)
    model_sample = sample_function(final_model_mean, model_log_variance, sample_noise);
  else
    model_sample = final_model_mean;
  std::memcpy(x, model_sample.data(), sizeof(float) * split_index);
}

// main.cpp:5575-5584
void ref_denormalize_mel(float *mel, int n) {
  std::vector<float> v(mel, mel + n);
  denormalize_tacotron_mel(v);
  std::memcpy(mel, v.data(), sizeof(float) * n);
}

// main.cpp:4510-4532; in: codes[n] (n<=500); out: 502 ints
void ref_apply_padding(const int *codes, int n, int *out502) {
  std::vector<int> v(codes, codes + n);
  apply_padding(v);
  std::memcpy(out502, v.data(), sizeof(int) * 502);
}

// main.cpp:4873-4915; latents [B][500][1024], codes [B][502]; returns per-candidate row counts,
// rows written back-to-back into out (capacity B*500*1024).
void ref_trim_latents(const float *latents, const int *codes502, int B, float *out, int *rows) {
  std::vector<float> lat(latents, latents + (size_t)B * 500 * 1024);
  std::vector<std::vector<int>> mc(B);
  for (int i = 0; i < B; i++) mc[i].assign(codes502 + i * 502, codes502 + (i + 1) * 502);
  std::streambuf *old = std::cout.rdbuf();
  std::ostringstream sink;
  std::cout.rdbuf(sink.rdbuf()); // trim_latents prints sizes
  std::vector<std::vector<float>> tl = trim_latents(lat, mc);
  std::cout.rdbuf(old);
  size_t off = 0;
  for (int i = 0; i < B; i++) {
    rows[i] = (int)(tl[i].size() / 1024);
    std::memcpy(out + off, tl[i].data(), sizeof(float) * tl[i].size());
    off += tl[i].size();
  }
}

// main.cpp:6550-6567: vocab init, " "->"[SPACE]", tokenize, wrap with 255 ... 0.
static gpt_vocab g_vocab;
static bool g_vocab_ok = false;
int ref_tokenizer_init(const char *json_path) {
  std::streambuf *old = std::cout.rdbuf();
  FILE *devnull = fopen("/dev/null", "w");
  FILE *saved = stdout;
  if (devnull) stdout = devnull; // gpt_vocab_init printf()s
  g_vocab = gpt_vocab();
  gpt_vocab_init(json_path, g_vocab);
  if (devnull) { stdout = saved; fclose(devnull); }
  (void)old;
  g_vocab_ok = true;
  return (int)g_vocab.token_to_id.size();
}
int ref_tokenize(const char *message_c, int *out, int cap) {
  if (!g_vocab_ok) return -1;
  std::string message(message_c);
  replaceAll(message, " ", "[SPACE]");
  std::vector<gpt_vocab::id> tokens = ::gpt_tokenize(g_vocab, message);
  tokens.insert(tokens.begin(), 255);
  tokens.push_back(0);
  int n = (int)tokens.size();
  for (int i = 0; i < n && i < cap; i++) out[i] = tokens[i];
  return n;
}

} // extern "C"

// TEST INFRASTRUCTURE — oracle restatement of the autoregressive stage.
//   autoregressive_graph (prefill + decode with KV cache)   main.cpp:2545-3040
//   autoregressive_latent_graph                              main.cpp:2053-2519
//   autoregressive() driver                                  main.cpp:5042-5367
// Layer count is discovered from the weight file (the reference hard-codes 30, main.cpp:701);
// every other dimension is the reference's (d=1024, 16 heads x 64, MLP 4096, vocab 8194).
#include "orc_common.h"
#include "orc_host.h"
#include <algorithm>

namespace orc {

static const int D = 1024, NH = 16, HD = 64, FF = 4096, V = 8194;

struct ArLayer {
  const float *ln1_g, *ln1_b, *ln2_g, *ln2_b;
  const float *w_attn, *b_attn; // [1024][3072] (in,out) HF Conv1D
  const float *w_proj, *b_proj; // [1024][1024]
  const float *w_fc, *b_fc;     // [1024][4096]
  const float *w_fc2, *b_fc2;   // [4096][1024]
};

struct Ar {
  const Model *m;
  int n_layers;
  std::vector<ArLayer> L;
  const float *text_emb, *text_pos, *mel_emb, *mel_pos, *lnf_g, *lnf_b, *lmh_g, *lmh_b, *lm_b;
  std::vector<float> lm_wt; // [1024][8194] transposed lm_head.1.weight
  // state
  int B = 0, n_text = 0, P = 0, max_pos = 0;
  std::vector<int> tokens;
  std::vector<float> voice;
  std::vector<float> kc, vc; // [layer][pos][cand][1024], values already f16-rounded

  explicit Ar(const Model *model) : m(model) {
    const std::string h = "inference_model.transformer.h.";
    n_layers = m->count_layers(h, ".ln_1.weight");
    L.resize(n_layers);
    for (int i = 0; i < n_layers; i++) {
      std::string p = h + std::to_string(i);
      L[i] = {m->p(p + ".ln_1.weight"),      m->p(p + ".ln_1.bias"),
              m->p(p + ".ln_2.weight"),      m->p(p + ".ln_2.bias"),
              m->p(p + ".attn.c_attn.weight"), m->p(p + ".attn.c_attn.bias"),
              m->p(p + ".attn.c_proj.weight"), m->p(p + ".attn.c_proj.bias"),
              m->p(p + ".mlp.c_fc.weight"),  m->p(p + ".mlp.c_fc.bias"),
              m->p(p + ".mlp.c_proj.weight"), m->p(p + ".mlp.c_proj.bias")};
    }
    text_emb = m->p("text_embedding.weight");
    text_pos = m->p("text_pos_embedding.emb.weight");
    mel_emb = m->p("mel_embedding.weight");
    mel_pos = m->p("mel_pos_embedding.emb.weight");
    lnf_g = m->p("inference_model.transformer.ln_f.weight");
    lnf_b = m->p("inference_model.transformer.ln_f.bias");
    lmh_g = m->p("inference_model.lm_head.0.weight");
    lmh_b = m->p("inference_model.lm_head.0.bias");
    lm_b = m->p("inference_model.lm_head.1.bias");
    lm_wt.resize((size_t)D * V);
    transpose(m->p("inference_model.lm_head.1.weight"), V, D, lm_wt.data());
  }

  // One transformer pass over `rows` = S positions x nb candidates laid out [cand][pos].
  // Self-attention is causal inside the S new positions and sees `n_past` cached positions.
  // use_cache=false: latent pass (keys/values only from this call).
  void forward(std::vector<float> &x, int S, int nb, int n_past, bool use_cache) {
    int rows = S * nb;
    std::vector<float> res(x), qkv((size_t)rows * 3 * D), att((size_t)rows * D),
        ff((size_t)rows * FF), tmp((size_t)rows * D);
    for (int l = 0; l < n_layers; l++) {
      const ArLayer &w = L[l];
      res = x;
      layernorm_rows(x.data(), rows, D, 1e-5f, w.ln1_g, w.ln1_b);
      gemm_kn(rows, 3 * D, D, x.data(), D, w.w_attn, 3 * D, qkv.data(), 3 * D, w.b_attn);
      // ggml_cpy F32->F16->F32 of the whole qkv (main.cpp:2789-2790)
#pragma omp parallel for schedule(static)
      for (int64_t i = 0; i < (int64_t)rows * 3 * D; i++) qkv[i] = f16r(qkv[i]);
      if (use_cache) {
        for (int c = 0; c < nb; c++)
          for (int s = 0; s < S; s++) {
            const float *src = qkv.data() + ((size_t)c * S + s) * 3 * D;
            size_t dst = (((size_t)l * max_pos + n_past + s) * B + c) * D;
            std::memcpy(&kc[dst], src + D, sizeof(float) * D);
            std::memcpy(&vc[dst], src + 2 * D, sizeof(float) * D);
          }
      }
      int ctx = n_past + S;
#pragma omp parallel for collapse(2) schedule(dynamic)
      for (int c = 0; c < nb; c++)
        for (int hh = 0; hh < NH; hh++) {
          std::vector<float> sc(ctx);
          for (int s = 0; s < S; s++) {
            const float *q = qkv.data() + ((size_t)c * S + s) * 3 * D + hh * HD;
            int nvis = n_past + s + 1; // diag_mask_inf(n_past): key j visible iff j <= n_past+s
            for (int j = 0; j < nvis; j++) {
              const float *k;
              if (use_cache) k = &kc[(((size_t)l * max_pos + j) * B + c) * D + hh * HD];
              else k = qkv.data() + ((size_t)c * S + j) * 3 * D + D + hh * HD;
              float dot = 0;
              for (int d = 0; d < HD; d++) dot += q[d] * k[d];
              sc[j] = dot * (1.0f / sqrtf(float(64)));
            }
            softmax_row(sc.data(), nvis);
            float *o = att.data() + ((size_t)c * S + s) * D + hh * HD;
            for (int d = 0; d < HD; d++) o[d] = 0;
            for (int j = 0; j < nvis; j++) {
              const float *v;
              if (use_cache) v = &vc[(((size_t)l * max_pos + j) * B + c) * D + hh * HD];
              else v = qkv.data() + ((size_t)c * S + j) * 3 * D + 2 * D + hh * HD;
              float p = sc[j];
              for (int d = 0; d < HD; d++) o[d] += p * v[d];
            }
          }
        }
      gemm_kn(rows, D, D, att.data(), D, w.w_proj, D, tmp.data(), D, w.b_proj);
      for (size_t i = 0; i < (size_t)rows * D; i++) x[i] = tmp[i] + res[i];
      res = x;
      layernorm_rows(x.data(), rows, D, 1e-5f, w.ln2_g, w.ln2_b);
      gemm_kn(rows, FF, D, x.data(), D, w.w_fc, FF, ff.data(), FF, w.b_fc);
#pragma omp parallel for schedule(static)
      for (int64_t i = 0; i < (int64_t)rows * FF; i++) ff[i] = gelu_f(ff[i]);
      gemm_kn(rows, D, FF, ff.data(), FF, w.w_fc2, D, tmp.data(), D, w.b_fc2);
      for (size_t i = 0; i < (size_t)rows * D; i++) x[i] = tmp[i] + res[i];
    }
  }

  // ln_f (affine) then lm_head.0 LayerNorm (affine) then lm_head.1 linear on given rows.
  void head_logits(const float *h, int rows, float *logits) {
    std::vector<float> t(h, h + (size_t)rows * D);
    layernorm_rows(t.data(), rows, D, 1e-5f, lnf_g, lnf_b);
    layernorm_rows(t.data(), rows, D, 1e-5f, lmh_g, lmh_b);
    gemm_kn(rows, V, D, t.data(), D, lm_wt.data(), V, logits, V, lm_b);
  }

  void start(const int *toks, int n, const float *voice1024, int batch, int max_positions) {
    tokens.assign(toks, toks + n);
    n_text = n;
    P = n + 2;
    B = batch;
    max_pos = max_positions;
    voice.assign(voice1024, voice1024 + D);
    kc.assign((size_t)n_layers * max_pos * B * D, 0.f);
    vc.assign((size_t)n_layers * max_pos * B * D, 0.f);
  }

  // Prefill (main.cpp:2586-2665, 5136-5186): [voice, text_emb+pos(0..n-1), mel_emb(8192)+mel_pos(0)],
  // identical for every candidate => computed once, K/V replicated into every candidate's slots.
  void prefill(float *logits_out) {
    std::vector<float> x((size_t)P * D);
    std::memcpy(x.data(), voice.data(), sizeof(float) * D);
    for (int i = 0; i < n_text; i++)
      for (int d = 0; d < D; d++)
        x[(size_t)(1 + i) * D + d] = text_emb[(size_t)tokens[i] * D + d] + text_pos[(size_t)i * D + d];
    for (int d = 0; d < D; d++)
      x[(size_t)(P - 1) * D + d] = mel_emb[(size_t)8192 * D + d] + mel_pos[d];
    int savedB = B;
    // run as a single candidate writing candidate slot 0, then replicate
    std::vector<float> kc1, vc1;
    {
      B = 1;
      kc1.swap(kc);
      vc1.swap(vc);
      kc.assign((size_t)n_layers * max_pos * D, 0.f);
      vc.assign((size_t)n_layers * max_pos * D, 0.f);
      forward(x, P, 1, 0, true);
      B = savedB;
      for (int l = 0; l < n_layers; l++)
        for (int p = 0; p < P; p++)
          for (int c = 0; c < B; c++) {
            std::memcpy(&kc1[(((size_t)l * max_pos + p) * B + c) * D],
                        &kc[((size_t)l * max_pos + p) * D], sizeof(float) * D);
            std::memcpy(&vc1[(((size_t)l * max_pos + p) * B + c) * D],
                        &vc[((size_t)l * max_pos + p) * D], sizeof(float) * D);
          }
      kc.swap(kc1);
      vc.swap(vc1);
    }
    std::vector<float> lg(V);
    head_logits(x.data() + (size_t)(P - 1) * D, 1, lg.data());
    for (int c = 0; c < B; c++) std::memcpy(logits_out + (size_t)c * V, lg.data(), sizeof(float) * V);
  }

  // Decode step i (main.cpp:2667-2693, 5227-5247): token emb + mel_pos[i+2], n_past = P + i.
  void step(const int *toks, int i, float *logits_out) {
    std::vector<float> x((size_t)B * D);
    for (int c = 0; c < B; c++)
      for (int d = 0; d < D; d++)
        x[(size_t)c * D + d] = mel_emb[(size_t)toks[c] * D + d] + mel_pos[(size_t)(i + 2) * D + d];
    forward(x, 1, B, P + i, true);
    head_logits(x.data(), B, logits_out);
  }

  // Latent pass (main.cpp:2053-2519, 5280-5352). codes502: [B][502]. n_mel<=502 positions are
  // evaluated (causal => rows < n_mel do not depend on later ones); out: [B][n_out][1024] with
  // n_out = min(500, n_mel). Mel positions are 0..501 per candidate (the reference's fill is
  // only right for B=4, main.cpp:5326-5333).
  void latents(const int *codes502, int nb, int n_mel, float *out) {
    int S = 1 + n_text + n_mel;
    std::vector<float> x((size_t)nb * S * D);
    for (int c = 0; c < nb; c++) {
      float *xc = x.data() + (size_t)c * S * D;
      std::memcpy(xc, voice.data(), sizeof(float) * D);
      for (int i = 0; i < n_text; i++)
        for (int d = 0; d < D; d++)
          xc[(size_t)(1 + i) * D + d] = text_emb[(size_t)tokens[i] * D + d] + text_pos[(size_t)i * D + d];
      for (int j = 0; j < n_mel; j++)
        for (int d = 0; d < D; d++)
          xc[(size_t)(1 + n_text + j) * D + d] =
              mel_emb[(size_t)codes502[c * 502 + j] * D + d] + mel_pos[(size_t)j * D + d];
    }
    int savedB = B;
    B = nb;
    forward(x, S, nb, 0, false);
    B = savedB;
    int n_out = std::min(500, n_mel);
    layernorm_rows(x.data(), nb * S, D, 1e-5f, lnf_g, lnf_b);
    layernorm_rows(x.data(), nb * S, D, 1e-5f, lmh_g, lmh_b);
    for (int c = 0; c < nb; c++)
      std::memcpy(out + (size_t)c * n_out * D, x.data() + ((size_t)c * S + 1 + n_text) * D,
                  sizeof(float) * n_out * D);
  }
};

} // namespace orc

using namespace orc;
extern "C" {
void *orc_model_load(const char *path) {
  std::string err;
  Model *m = load_model(path, err);
  if (!m) fprintf(stderr, "orc_model_load: %s\n", err.c_str());
  return m;
}
void orc_model_free(void *m) { delete (Model *)m; }
int orc_model_get(void *m, const char *name, float *out, int64_t cap) {
  Model *mm = (Model *)m;
  if (!mm->has(name)) return -1;
  const Tensor &t = mm->get(name);
  if (out) std::memcpy(out, t.data.data(), sizeof(float) * std::min<int64_t>(cap, t.nelem()));
  return (int)t.nelem();
}
void *orc_ar_new(void *model) { return new Ar((Model *)model); }
void orc_ar_free(void *a) { delete (Ar *)a; }
int orc_ar_layers(void *a) { return ((Ar *)a)->n_layers; }
void orc_ar_start(void *a, const int *toks, int n, const float *voice, int B, int max_pos) {
  ((Ar *)a)->start(toks, n, voice, B, max_pos);
}
void orc_ar_prefill(void *a, float *logits) { ((Ar *)a)->prefill(logits); }
void orc_ar_step(void *a, const int *toks, int i, float *logits) { ((Ar *)a)->step(toks, i, logits); }
void orc_ar_latents(void *a, const int *codes502, int nb, int n_mel, float *out) {
  ((Ar *)a)->latents(codes502, nb, n_mel, out);
}

// Whole autoregressive() driver (main.cpp:5042-5367). max_steps bounds the loop (the reference
// has no bound besides its 404-position KV cache); mask_stop=1 sets logit[8193]=-inf-like lowest
// before sampling so exactly max_steps codes are produced (bench workload).
//   out_codes: [B][502]; out_steps: number of sampling iterations executed.
//   returns 0, or -1 if max_steps was hit without the reference's all-stopped condition.
int orc_autoregressive(void *a_, const int *toks, int n, const float *voice, int B, void *rng_,
                       int max_steps, int mask_stop, int *out_codes, int *out_steps,
                       int *out_first_ids /* optional [B][max_steps] raw samples */) {
  Ar *a = (Ar *)a_;
  Rng &rng = *(Rng *)rng_;
  a->start(toks, n, voice, B, n + 2 + max_steps + 1);
  std::vector<float> logits((size_t)B * V);
  a->prefill(logits.data());
  std::vector<int> ids((size_t)(n + 2) * B);
  for (size_t i = 0; i < ids.size(); i++) ids[i] = (i % (n + 2) == (size_t)(n + 1)) ? 8192 : 1;
  std::vector<std::vector<int>> seq(B);
  std::vector<int> samples(B);
  int i = 0, rc = 0;
  while (true) {
    if (mask_stop)
      for (int c = 0; c < B; c++) logits[(size_t)c * V + 8193] = -1e30f;
    sample_batch(logits.data(), ids.data(), (int)ids.size(), B, rng, samples.data(), nullptr);
    int stops = 0;
    ids.clear();
    for (int c = 0; c < B; c++) {
      if (!(seq[c].size() > 0 && seq[c].back() == 8193)) seq[c].push_back(samples[c]);
      if (samples[c] == 8193) stops++;
      ids.push_back(samples[c]);
      if (out_first_ids) out_first_ids[(size_t)c * max_steps + i] = samples[c];
    }
    i++;
    if (stops == B) break;
    if (i >= max_steps) { rc = mask_stop ? 0 : -1; break; }
    a->step(ids.data(), i - 1, logits.data()); // reference also runs one wasted step after the break
  }
  *out_steps = i;
  for (int c = 0; c < B; c++) {
    std::vector<int> v = seq[c];
    if (v.size() > 500) v.resize(500);
    apply_padding(v);
    std::memcpy(out_codes + (size_t)c * 502, v.data(), sizeof(int) * 502);
  }
  return rc;
}
}

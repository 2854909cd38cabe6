// TEST INFRASTRUCTURE — oracle restatement of the diffusion stage.
//   diffusion_graph    main.cpp:3066-4044   (one eps/variance prediction)
//   diffusion() driver main.cpp:5614-6042   (80 x {cond, uncond, ancestral update})
// Activations are kept [T][C] (C contiguous); the reference's [T fastest, C] tensors are the
// transpose, so API-visible tensors (x_t [100][T], output [200][T]) are converted at the edges.
// Main-layer count is discovered from the file (reference: 10 + 3, main.cpp:3656, 3890).
#include "orc_common.h"
#include "orc_host.h"
#include <algorithm>

namespace orc {

static const int C = 1024;

struct AttnW {
  const float *norm_g, *norm_b, *qkv_w, *qkv_b, *proj_w, *proj_b, *relpos; // relpos [32][16]
};
struct ResW {
  const float *in_g, *in_b, *in_w, *in_bias, *emb_w, *emb_b, *out_g, *out_b, *out_w, *out_bias;
};

static AttnW attn_w(const Model &m, const std::string &p) {
  return {m.p(p + ".norm.weight"), m.p(p + ".norm.bias"), m.p(p + ".qkv.weight"),
          m.p(p + ".qkv.bias"), m.p(p + ".proj_out.weight"), m.p(p + ".proj_out.bias"),
          m.p(p + ".relative_pos_embeddings.relative_attention_bias.weight")};
}
static ResW res_w(const Model &m, const std::string &p) {
  return {m.p(p + ".in_layers.0.weight"), m.p(p + ".in_layers.0.bias"),
          m.p(p + ".in_layers.2.weight"), m.p(p + ".in_layers.2.bias"),
          m.p(p + ".emb_layers.1.weight"), m.p(p + ".emb_layers.1.bias"),
          m.p(p + ".out_layers.0.weight"), m.p(p + ".out_layers.0.bias"),
          m.p(p + ".out_layers.3.weight"), m.p(p + ".out_layers.3.bias")};
}

struct Diff {
  const Model *m;
  std::vector<AttnW> lc_attn;               // latent_conditioner.1..4
  std::vector<ResW> integ_res, main_res, tail_res;
  std::vector<AttnW> integ_attn, main_attn;

  explicit Diff(const Model *model) : m(model) {
    for (int i = 1; m->has("latent_conditioner." + std::to_string(i) + ".norm.weight"); i++)
      lc_attn.push_back(attn_w(*m, "latent_conditioner." + std::to_string(i)));
    for (int i = 0; m->has("conditioning_timestep_integrator." + std::to_string(i) + ".resblk.in_layers.0.weight"); i++) {
      std::string p = "conditioning_timestep_integrator." + std::to_string(i);
      integ_res.push_back(res_w(*m, p + ".resblk"));
      integ_attn.push_back(attn_w(*m, p + ".attn"));
    }
    int i = 0;
    for (; m->has("layers." + std::to_string(i) + ".resblk.in_layers.0.weight"); i++) {
      std::string p = "layers." + std::to_string(i);
      main_res.push_back(res_w(*m, p + ".resblk"));
      main_attn.push_back(attn_w(*m, p + ".attn"));
    }
    for (; m->has("layers." + std::to_string(i) + ".in_layers.0.weight"); i++)
      tail_res.push_back(res_w(*m, "layers." + std::to_string(i)));
  }

  // AttentionBlock (main.cpp:3184-3288 / 3491-3608 / 3785-3886): GN32 -> qkv conv k1 (f16) ->
  // 16 heads, channel = h*192 + {q:0..63,k:64..127,v:128..191} -> softmax(q.k/8 + 8*relbias) ->
  // proj_out (F32 linear) -> + residual.
  void attention(std::vector<float> &x, int T, const AttnW &w) const {
    std::vector<float> h((size_t)T * C), qkv((size_t)T * 3 * C), a((size_t)T * C), o((size_t)T * C);
    groupnorm_tc(x.data(), T, C, 32, g_flags.gn_eps, w.norm_g, w.norm_b, h.data());
    conv1d_f16(h.data(), T, C, w.qkv_w, 1, 3 * C, w.qkv_b, 0, 1, qkv.data());
#pragma omp parallel for collapse(2) schedule(dynamic)
    for (int hh = 0; hh < 16; hh++)
      for (int i = 0; i < T; i++) {
        std::vector<float> sc(T);
        const float *q = qkv.data() + (size_t)i * 3 * C + hh * 192;
        for (int j = 0; j < T; j++) {
          const float *k = qkv.data() + (size_t)j * 3 * C + hh * 192 + 64;
          float dot = 0;
          for (int d = 0; d < 64; d++) dot += q[d] * k[d];
          float bias = w.relpos[(size_t)bucket_of(i, j) * 16 + hh] * 8.0f;
          sc[j] = bias + dot * (1.0f / sqrtf(float(64)));
        }
        softmax_row(sc.data(), T);
        float *out = a.data() + (size_t)i * C + hh * 64;
        for (int d = 0; d < 64; d++) out[d] = 0;
        for (int j = 0; j < T; j++) {
          const float *v = qkv.data() + (size_t)j * 3 * C + hh * 192 + 128;
          float p = sc[j];
          for (int d = 0; d < 64; d++) out[d] += p * v[d];
        }
      }
    gemm_nk(T, C, C, a.data(), C, w.proj_w, C, o.data(), C, w.proj_b);
    for (size_t i = 0; i < (size_t)T * C; i++) x[i] = o[i] + x[i];
  }

  // ResBlock (main.cpp:3659-3782): GN->SiLU->conv k1 ; emb: SiLU->Linear 1024->2048 = scale|shift;
  // GN*g+b -> *(scale + 1) + shift -> SiLU -> conv k3 pad1 -> + skip.
  void resblock(std::vector<float> &x, int T, const ResW &w, const float *emb) const {
    std::vector<float> h((size_t)T * C), h2((size_t)T * C);
    groupnorm_tc(x.data(), T, C, 32, g_flags.gn_eps, w.in_g, w.in_b, h.data());
    for (auto &v : h) v = silu_f(v);
    conv1d_f16(h.data(), T, C, w.in_w, 1, C, w.in_bias, 0, 1, h2.data());
    std::vector<float> se(C), ss(2 * C);
    for (int i = 0; i < C; i++) se[i] = silu_f(emb[i]);
    gemm_nk(1, 2 * C, C, se.data(), C, w.emb_w, C, ss.data(), 2 * C, w.emb_b);
    groupnorm_tc(h2.data(), T, C, 32, g_flags.gn_eps, w.out_g, w.out_b, h.data());
    const float offset = 1.0f; // conditioning_scale_offset (main.cpp:5778)
    for (int t = 0; t < T; t++)
      for (int c = 0; c < C; c++) {
        float v = h[(size_t)t * C + c] * (ss[c] + offset);
        v = v + ss[C + c];
        h[(size_t)t * C + c] = silu_f(v);
      }
    conv1d_f16(h.data(), T, C, w.out_w, 3, C, w.out_bias, 1, 1, h2.data());
    for (size_t i = 0; i < (size_t)T * C; i++) x[i] = h2[i] + x[i];
  }

  // Conditioned code embedding (main.cpp:3156-3321), timestep independent: conv k3 -> 4 attn ->
  // GN(code_norm) -> *(1+scale)+shift from diffusion_conditioning_latent -> nearest upsample L->T.
  void code_embedding(const float *latents, int L, int T, float *out) const {
    std::vector<float> x((size_t)L * C);
    conv1d_f16(latents, L, C, m->p("latent_conditioner.0.weight"), 3, C,
               m->p("latent_conditioner.0.bias"), 1, 1, x.data());
    for (const auto &w : lc_attn) attention(x, L, w);
    std::vector<float> h((size_t)L * C);
    groupnorm_tc(x.data(), L, C, 32, g_flags.gn_eps, m->p("code_norm.weight"),
                 m->p("code_norm.bias"), h.data());
    const float *cl = m->p("diffusion_conditioning_latent");
    for (int t = 0; t < L; t++)
      for (int c = 0; c < C; c++)
        h[(size_t)t * C + c] = h[(size_t)t * C + c] * (cl[c] + 1.0f) + cl[C + c];
    // ggml_upscale_ext nearest: src = (int)(dst / ((float)T / L))
    const float sf = (float)T / (float)L;
    for (int t = 0; t < T; t++) {
      int s = (int)(t / sf);
      if (s > L - 1) s = L - 1;
      std::memcpy(out + (size_t)t * C, h.data() + (size_t)s * C, sizeof(float) * C);
    }
  }

  // One network evaluation. code_emb: [T][1024] (cond) or nullptr (unconditioned_embedding repeated).
  // x_t: [100][T]; out: [200][T] (reference layouts).
  void forward(const float *code_emb, const float *x_t, int T, int timestep, float *out) const {
    std::vector<float> te(1024), e1(1024), emb(1024);
    timestep_embedding(timestep, te.data());
    gemm_nk(1, C, C, te.data(), C, m->p("time_embed.0.weight"), C, e1.data(), C, m->p("time_embed.0.bias"));
    for (auto &v : e1) v = silu_f(v);
    gemm_nk(1, C, C, e1.data(), C, m->p("time_embed.2.weight"), C, emb.data(), C, m->p("time_embed.2.bias"));

    std::vector<float> ce((size_t)T * C);
    if (code_emb) std::memcpy(ce.data(), code_emb, sizeof(float) * T * C);
    else {
      const float *u = m->p("unconditioned_embedding");
      for (int t = 0; t < T; t++) std::memcpy(ce.data() + (size_t)t * C, u, sizeof(float) * C);
    }
    for (size_t i = 0; i < integ_res.size(); i++) {
      resblock(ce, T, integ_res[i], emb.data());
      attention(ce, T, integ_attn[i]);
    }
    // inp_block conv k3 100->1024 on x_t, concat [x | code_emb] -> integrating conv k1 2048->1024
    std::vector<float> xt((size_t)T * 100), xin((size_t)T * 2 * C), x((size_t)T * C);
    for (int c = 0; c < 100; c++)
      for (int t = 0; t < T; t++) xt[(size_t)t * 100 + c] = x_t[(size_t)c * T + t];
    std::vector<float> xi((size_t)T * C);
    conv1d_f16(xt.data(), T, 100, m->p("inp_block.weight"), 3, C, m->p("inp_block.bias"), 1, 1, xi.data());
    for (int t = 0; t < T; t++) {
      std::memcpy(xin.data() + (size_t)t * 2 * C, xi.data() + (size_t)t * C, sizeof(float) * C);
      std::memcpy(xin.data() + (size_t)t * 2 * C + C, ce.data() + (size_t)t * C, sizeof(float) * C);
    }
    conv1d_f16(xin.data(), T, 2 * C, m->p("integrating_conv.weight"), 1, C,
               m->p("integrating_conv.bias"), 0, 1, x.data());
    for (size_t i = 0; i < main_res.size(); i++) {
      resblock(x, T, main_res[i], emb.data());
      attention(x, T, main_attn[i]);
    }
    for (size_t i = 0; i < tail_res.size(); i++) resblock(x, T, tail_res[i], emb.data());
    std::vector<float> h((size_t)T * C), o((size_t)T * 200);
    groupnorm_tc(x.data(), T, C, 32, g_flags.gn_eps, m->p("out.0.weight"), m->p("out.0.bias"), h.data());
    for (auto &v : h) v = silu_f(v);
    conv1d_f16(h.data(), T, C, m->p("out.2.weight"), 3, 200, m->p("out.2.bias"), 1, 1, o.data());
    for (int c = 0; c < 200; c++)
      for (int t = 0; t < T; t++) out[(size_t)c * T + t] = o[(size_t)t * 200 + c];
  }
};

} // namespace orc

using namespace orc;
extern "C" {
void *orc_diff_new(void *model) { return new Diff((Model *)model); }
void orc_diff_free(void *d) { delete (Diff *)d; }
int orc_diff_T(int L) { return L * 4 * 24000 / 22050; } // main.cpp:5616-5617
void orc_diff_code_embedding(void *d, const float *latents, int L, int T, float *out) {
  ((Diff *)d)->code_embedding(latents, L, T, out);
}
void orc_diff_forward(void *d, const float *code_emb, const float *x_t, int T, int timestep,
                      float *out) {
  ((Diff *)d)->forward(code_emb, x_t, T, timestep, out);
}
// diffusion() (main.cpp:5614-6042). noise: [(n_steps+1)][100*T] (x_T then one vector per step, the
// last one unused) or NULL to draw from rng in the reference's order. mel_out: [100][T].
void orc_diffusion(void *d_, const float *latents, int L, int n_steps, void *rng_,
                   const float *noise, float *mel_out) {
  Diff *d = (Diff *)d_;
  int T = orc_diff_T(L), N = 100 * T;
  std::vector<int> tm = default_timestep_map(n_steps);
  Schedule s;
  s.build(tm);
  std::vector<float> x(N), nz(N), ce((size_t)T * C), oc(2 * N), ou(2 * N);
  if (noise) std::memcpy(x.data(), noise, sizeof(float) * N);
  else for (int i = 0; i < N; i++) x[i] = (float)((Rng *)rng_)->normal();
  d->code_embedding(latents, L, T, ce.data());
  for (int idx = 0; idx < n_steps; idx++) {
    int t = n_steps - 1 - idx;
    d->forward(ce.data(), x.data(), T, tm[t], oc.data());
    d->forward(nullptr, x.data(), T, tm[t], ou.data());
    if (noise) std::memcpy(nz.data(), noise + (size_t)(idx + 1) * N, sizeof(float) * N);
    else for (int i = 0; i < N; i++) nz[i] = (float)((Rng *)rng_)->normal();
    diffusion_update(s, t, n_steps, oc.data(), ou.data(), x.data(), nz.data(), T);
  }
  std::memcpy(mel_out, x.data(), sizeof(float) * N);
}
}

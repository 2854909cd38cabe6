// TEST INFRASTRUCTURE — oracle restatement of the UnivNet vocoder stage.
//   vocoder_graph    main.cpp:4068-4483
//   vocoder() driver main.cpp:6044-6127 (+ denormalize_tacotron_mel 5575-5584)
// Activations are [len][C] (C contiguous); reference tensors are [len fastest, C].
#include "orc_common.h"
#include "orc_host.h"
#include <algorithm>

namespace orc {

static inline float leaky(float v) { return v > 0 ? v : 0.2f * v; } // ggml_leaky_relu(.,0.2)

struct Voc {
  const Model *m;
  explicit Voc(const Model *model) : m(model) {}

  // mel_denorm: [100][T] (already denormalised); noise: [64][Tm] (reference layout [Tm fastest,64]);
  // audio: [Tm*256 - 6]
  void forward(const float *mel_denorm, int T, const float *noise, float *audio) const {
    const int Tm = T + 10;
    // padded mel [Tm][100]: mel then 10 frames of -11.5129 (main.cpp:6051-6054, 4106-4112)
    std::vector<float> pm((size_t)Tm * 100);
    for (int t = 0; t < Tm; t++)
      for (int c = 0; c < 100; c++)
        pm[(size_t)t * 100 + c] = (t < T) ? mel_denorm[(size_t)c * T + t] : -11.5129f;
    // reflect pad 3 + conv_pre k7 64->32 (4114-4130)
    std::vector<float> z((size_t)(Tm + 6) * 64);
    for (int t = 0; t < Tm + 6; t++) {
      int s = t - 3;
      if (s < 0) s = -s;
      if (s >= Tm) s = 2 * (Tm - 1) - s;
      for (int c = 0; c < 64; c++) z[(size_t)t * 64 + c] = noise[(size_t)c * Tm + s];
    }
    std::vector<float> cur((size_t)Tm * 32);
    conv1d_f16(z.data(), Tm + 6, 64, m->p("conv_pre.weight"), 7, 32, m->p("conv_pre.bias"), 0, 1, cur.data());
    int len = Tm;
    const int strides[3] = {8, 8, 4}, hops[3] = {8, 64, 256};
    for (int i = 0; i < 3; i++) {
      std::string rs = "res_stack." + std::to_string(i);
      const int s = strides[i], K = 2 * s, hop = hops[i];
      // leaky -> conv_transpose_1d (F32 kernel [K][Cout][Cin]) -> crop s/2 each side -> + bias
      // (4145-4167)
      const float *wt = m->p(rs + ".convt_pre.1.weight");
      const float *bt = m->p(rs + ".convt_pre.1.bias");
      int full = (len - 1) * s + K, nlen = len * s;
      std::vector<float> up((size_t)full * 32, 0.f);
      for (int t = 0; t < len; t++)
        for (int ci = 0; ci < 32; ci++) {
          float xv = leaky(cur[(size_t)t * 32 + ci]);
          for (int co = 0; co < 32; co++)
            for (int k = 0; k < K; k++)
              up[(size_t)(t * s + k) * 32 + co] += xv * wt[((size_t)ci * 32 + co) * K + k];
        }
      std::vector<float> x((size_t)nlen * 32);
      for (int t = 0; t < nlen; t++)
        for (int co = 0; co < 32; co++) x[(size_t)t * 32 + co] = up[(size_t)(t + s / 2) * 32 + co] + bt[co];
      len = nlen;
      // kernel predictor on the padded mel (4169-4324)
      std::vector<float> c0((size_t)Tm * 64), c1((size_t)Tm * 64), c2((size_t)Tm * 64);
      conv1d_f16(pm.data(), Tm, 100, m->p(rs + ".kernel_predictor.input_conv.0.weight"), 5, 64,
                 m->p(rs + ".kernel_predictor.input_conv.0.bias"), 2, 1, c0.data());
      for (auto &v : c0) v = leaky(v);
      for (int r = 0; r < 3; r++) {
        std::string rp = rs + ".kernel_predictor.residual_convs." + std::to_string(r);
        conv1d_f16(c0.data(), Tm, 64, m->p(rp + ".1.weight"), 3, 64, m->p(rp + ".1.bias"), 1, 1, c1.data());
        for (auto &v : c1) v = leaky(v);
        conv1d_f16(c1.data(), Tm, 64, m->p(rp + ".3.weight"), 3, 64, m->p(rp + ".3.bias"), 1, 1, c2.data());
        for (size_t q = 0; q < c0.size(); q++) c0[q] = c0[q] + leaky(c2[q]);
      }
      std::vector<float> kern((size_t)Tm * 24576), kb((size_t)Tm * 256);
      conv1d_f16(c0.data(), Tm, 64, m->p(rs + ".kernel_predictor.kernel_conv.weight"), 3, 24576,
                 m->p(rs + ".kernel_predictor.kernel_conv.bias"), 1, 1, kern.data());
      conv1d_f16(c0.data(), Tm, 64, m->p(rs + ".kernel_predictor.bias_conv.weight"), 3, 256,
                 m->p(rs + ".kernel_predictor.bias_conv.bias"), 1, 1, kb.data());
      // 4 LVC layers (4337-4456)
      const int dil[4] = {1, 3, 9, 27};
      for (int c = 0; c < 4; c++) {
        std::string cb = rs + ".conv_blocks." + std::to_string(c) + ".1";
        std::vector<float> a((size_t)len * 32), y((size_t)len * 32);
        for (size_t q = 0; q < a.size(); q++) a[q] = leaky(x[q]);
        conv1d_f16(a.data(), len, 32, m->p(cb + ".weight"), 3, 32, m->p(cb + ".bias"), dil[c], dil[c], y.data());
        for (auto &v : y) v = leaky(v);
        // location-variable conv: out[o][l*hop+s] = b[l][o] + sum_i sum_k ypad[i][l*hop+s+k] * W_l[i][o][k]
        // kernel channel index = ((c*32 + i)*64 + o)*3 + k ; bias channel = c*64 + o. F32 math,
        // input channels reduced in ascending order (4404-4419).
#pragma omp parallel for schedule(static)
        for (int l = 0; l < Tm; l++) {
          const float *Wl = kern.data() + (size_t)l * 24576 + (size_t)c * 6144;
          const float *bl = kb.data() + (size_t)l * 256 + c * 64;
          for (int sidx = 0; sidx < hop; sidx++) {
            int pos = l * hop + sidx;
            float o[64];
            for (int oo = 0; oo < 64; oo++) o[oo] = 0.f;
            for (int ic = 0; ic < 32; ic++) {
              float xv[3];
              for (int k = 0; k < 3; k++) {
                int p = pos + k - 1;
                xv[k] = (p < 0 || p >= len) ? 0.f : y[(size_t)p * 32 + ic];
              }
              for (int oo = 0; oo < 64; oo++) {
                const float *wk = Wl + ((size_t)ic * 64 + oo) * 3;
                float part = xv[0] * wk[0] + xv[1] * wk[1] + xv[2] * wk[2];
                o[oo] = (ic == 0) ? part : o[oo] + part;
              }
            }
            for (int oo = 0; oo < 32; oo++) {
              float sg = o[oo] + bl[oo], th = o[32 + oo] + bl[32 + oo];
              float g = 1.0f / (1.0f + expf(-sg)) * tanhf(th);
              x[(size_t)pos * 32 + oo] = x[(size_t)pos * 32 + oo] + g;
            }
          }
        }
      }
      cur.swap(x);
    }
    // leaky -> conv_post k7 32->1 pad 0, no tanh (4459-4478)
    std::vector<float> a((size_t)len * 32);
    for (size_t q = 0; q < a.size(); q++) a[q] = leaky(cur[q]);
    conv1d_f16(a.data(), len, 32, m->p("conv_post.1.weight"), 7, 1, m->p("conv_post.1.bias"), 0, 1, audio);
  }
};

} // namespace orc

using namespace orc;
extern "C" {
void *orc_voc_new(void *model) { return new Voc((Model *)model); }
void orc_voc_free(void *v) { delete (Voc *)v; }
int orc_voc_audio_len(int T) { return (T + 10) * 256 - 6; }
void orc_denormalize_mel(float *mel, int64_t n) { // main.cpp:5575-5584
  const float TACOTRON_MEL_MAX = 2.3143386840820312;
  const float TACOTRON_MEL_MIN = -11.512925148010254;
  for (int64_t i = 0; i < n; i++)
    mel[i] = ((mel[i] + 1) / 2) * (TACOTRON_MEL_MAX - TACOTRON_MEL_MIN) + TACOTRON_MEL_MIN;
}
void orc_voc_forward(void *v, const float *mel_denorm, int T, const float *noise, float *audio) {
  ((Voc *)v)->forward(mel_denorm, T, noise, audio);
}
// vocoder() (main.cpp:6044-6127): mel [100][T] normalised; noise [64][Tm] or NULL (drawn from rng).
void orc_vocoder(void *v, const float *mel, int T, void *rng_, const float *noise, float *audio) {
  std::vector<float> mm(mel, mel + (size_t)100 * T);
  orc_denormalize_mel(mm.data(), (int64_t)mm.size());
  int Tm = T + 10;
  std::vector<float> nz((size_t)Tm * 64);
  if (noise) std::memcpy(nz.data(), noise, sizeof(float) * nz.size());
  else for (auto &x : nz) x = (float)((Rng *)rng_)->normal();
  ((Voc *)v)->forward(mm.data(), T, nz.data(), audio);
}
}

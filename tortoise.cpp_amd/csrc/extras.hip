// Upstream tortoise-tts pieces the reference leaves out, on MI355X (gfx950):
//   1. CLVP candidate re-ranking                      tts_load_clvp / tts_clvp_score
//   2. the voice-conditioning encoder (AR side)        tts_load_voice_encoder / tts_voice_latent
//   3. the diffusion conditioning encoder              tts_load_diffusion_conditioning_encoder / tts_diffusion_conditioning_latent
//
// 1. The reference has no CLVP: main.cpp:6575 writes candidate 0. Upstream tortoise-tts scores every autoregressive candidate with CLVP
// (tortoise/models/clvp.py, use_xformers=True) and keeps the best; this file is that scorer, for a weight file in the reference's container
// format whose tensor names are the upstream state dict's (tortoise.cpp_amd/synth_weights.py: write_clvp lists them). Checked against
// oracle.Clvp (numpy, pinned against a torch restatement): no upstream weights or fixtures exist offline, so its parity is "unpinned" in the
// sense of the task statement. Model: two encoders (text, speech codes) of `depth` x [RMSNorm -> attention (bias-free q/k/v, rotary on the
// first 32 of 64 head dims, softmax(q k^T / 8) v, to_out + bias) -> residual; RMSNorm -> GEGLU feed-forward (ff = 2 dim) -> residual], final
// LayerNorm, mean over the sequence, bias-free latent projection, L2 normalise; score = <text latent, speech latent> exp(temperature).
//
// 2 / 3. The reference READS both voice latents (--voice: main.cpp:5179-5184; `diffusion_conditioning_latent`, a weight of
// ggml-diffusion-model.bin: main.cpp:1557-1560); README.md:54-72 gives an offline PyTorch recipe. Here: upstream's
// UnifiedVoice.get_conditioning and DiffusionTts.get_conditioning (see the sections below), from the mel spectrograms of the reference clips.
//
// Device layout (all three): the sequences of a call are packed into ONE row-major activation matrix x[rows][dim] (f32 residual stream,
// rows padded to a multiple of 128); every projection / convolution is one launch of the fp16-MFMA GEMM of gemm_f16.h (fp16 operands, f32
// accumulate, bias and residual fused into the epilogue) over all rows; attention runs per (sequence, head) with K/V staged through LDS.
// These run once per utterance (CLVP: 13 ms for 16 candidates) or once per voice: the kernels are written for clarity, not for the roofline.
#include "common.h"
#include "gemm_f16.h"
#include <cmath>

namespace tts {

namespace {
constexpr int CLVP_DH = 64; // head dim of CLVP and of the voice encoder (the first 32 carry CLVP's rotary embedding)

__global__ __launch_bounds__(256) void clvp_embed_kernel(const float *__restrict__ emb, const int *__restrict__ tok, int dim, float *__restrict__ x) {
  const float *src = emb + (size_t)tok[blockIdx.x] * dim;
  for (int c = threadIdx.x; c < dim; c += 256) x[(size_t)blockIdx.x * dim + c] = src[c];
}

__device__ __forceinline__ float block_sum(float v, float *red) { // 256 threads
  for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

// y = x / max(|x| dim^-1/2, 1e-8) * g, rounded to fp16 (the A operand of the next GEMM)
__global__ __launch_bounds__(256) void clvp_rmsnorm_kernel(const float *__restrict__ x, const float *__restrict__ g, int dim, __half *__restrict__ y) {
  __shared__ float red[4];
  const float *xr = x + (size_t)blockIdx.x * dim;
  float ss = 0.f;
  for (int c = threadIdx.x; c < dim; c += 256) ss += xr[c] * xr[c];
  ss = block_sum(ss, red);
  const float inv = 1.0f / fmaxf(sqrtf(ss) * rsqrtf((float)dim), 1e-8f);
  for (int c = threadIdx.x; c < dim; c += 256) y[(size_t)blockIdx.x * dim + c] = __float2half_rn(xr[c] * inv * g[c]);
}

// One workgroup = 256 queries of one (sequence, head); a thread owns one query (q and the output row in registers, f32), the keys stream
// through LDS in chunks of 128 (K rotated on the way in). qkv: fp16 [row][3 inner] = q | k | v, head h at columns h*64.
// ROT: CLVP's rotary embedding on the first 32 head dims. DH: head dim (64, or 128 for the diffusion conditioning encoder). BIAS: relative
// position bias by signed key - query distance (T5 buckets expanded per distance at load, saturated at +-63, already multiplied by sqrt(DH)):
// bias_tab[head][(query < key) * 64 + min(|key - query|, 63)].
template <bool ROT, int DH, bool BIAS>
__global__ __launch_bounds__(256) void clvp_attn_kernel(const __half *__restrict__ qkv, const int *__restrict__ seq_start, const int *__restrict__ seq_len,
                                                        int inner, __half *__restrict__ out, const float *__restrict__ bias_tab) {
  constexpr int KC = 8192 / DH, PARTS = DH / 32; // keys per LDS stage (32 KB for K and V together), 32-dim parts per key
  __shared__ __half sk[KC][DH], sv[KC][DH];
  __shared__ float stab[128];
  const int s = blockIdx.x, h = blockIdx.y, n = seq_len[s], r0 = seq_start[s];
  const int qi = blockIdx.z * 256 + threadIdx.x;
  if (blockIdx.z * 256 >= n) return;
  const int ld = 3 * inner;
  auto rot = [](float *t, int pos) { // rotary embedding on dims 0..31: pairs (d, d + 16), angle pos * 10000^(-2 d / 32)
#pragma unroll
    for (int d = 0; d < 16; d++) {
      const float ang = (float)pos * exp2f(-(float)d * (13.287712379549449f / 16.0f)); // 10000^(-d/16)
      const float c = cosf(ang), sn = sinf(ang), a = t[d], b = t[d + 16];
      t[d] = a * c - b * sn;
      t[d + 16] = b * c + a * sn;
    }
  };
  if (BIAS && threadIdx.x < 128) stab[threadIdx.x] = bias_tab[h * 128 + threadIdx.x];
  float q[DH], acc[DH];
  const bool live = qi < n;
  {
    const __half *qp = qkv + (size_t)(r0 + (live ? qi : 0)) * ld + h * DH;
#pragma unroll
    for (int d = 0; d < DH; d++) { q[d] = __half2float(qp[d]); acc[d] = 0.f; }
    if (ROT) rot(q, live ? qi : 0);
  }
  const float scale = DH == 64 ? 0.125f : 0.08838834764831845f; // 1 / sqrt(DH)
  float m = -INFINITY, l = 0.f;
  for (int k0 = 0; k0 < n; k0 += KC) {
    __syncthreads();
    { // 256 threads stage KC keys: thread t -> key t / PARTS, dims (t % PARTS) * 32 .. + 31 (the rotated dims 0..31 belong to ONE thread)
      const int kj = threadIdx.x / PARTS, part = threadIdx.x % PARTS, key = k0 + kj;
      if (key < n) {
        const __half *kp = qkv + (size_t)(r0 + key) * ld + inner + h * DH + part * 32;
        const __half *vp = qkv + (size_t)(r0 + key) * ld + 2 * inner + h * DH + part * 32;
        float t[32], u[32];
#pragma unroll
        for (int d = 0; d < 32; d++) { t[d] = __half2float(kp[d]); u[d] = __half2float(vp[d]); }
        // the x-transformers copy vendored in upstream tortoise-tts rotates the first 32 dims of q, k AND v (Attention.forward:
        // `ql, kl, vl = map(lambda t: apply_rotary_pos_emb(t, rotary_pos_emb), (ql, kl, vl))`), v at its own (key) position
        if (ROT && part == 0) { rot(t, key); rot(u, key); }
#pragma unroll
        for (int d = 0; d < 32; d++) { sk[kj][part * 32 + d] = __float2half_rn(t[d]); sv[kj][part * 32 + d] = __float2half_rn(u[d]); }
      }
    }
    __syncthreads();
    const int kn = min(KC, n - k0);
    for (int j = 0; j < kn; j++) {
      float sc = 0.f;
#pragma unroll
      for (int d = 0; d < DH; d++) sc += q[d] * __half2float(sk[j][d]);
      sc *= scale;
      if (BIAS) {
        const int dist = k0 + j - qi, ad = dist < 0 ? -dist : dist;
        sc += stab[(dist > 0 ? 64 : 0) + (ad < 63 ? ad : 63)];
      }
      const float mn = fmaxf(m, sc), a = expf(m - mn), p = expf(sc - mn);
      l = l * a + p;
#pragma unroll
      for (int d = 0; d < DH; d++) acc[d] = acc[d] * a + p * __half2float(sv[j][d]);
      m = mn;
    }
  }
  if (live) {
    const float inv = 1.0f / l;
    __half *op = out + (size_t)(r0 + qi) * inner + h * DH;
#pragma unroll
    for (int d = 0; d < DH; d++) op[d] = __float2half_rn(acc[d] * inv);
  }
}

// GEGLU: y = u[:, :ff] * gelu(u[:, ff:]) (erf GELU), fp16
__global__ __launch_bounds__(256) void clvp_geglu_kernel(const float *__restrict__ u, int ff, __half *__restrict__ y) {
  const float *ur = u + (size_t)blockIdx.x * 2 * ff;
  for (int c = threadIdx.x; c < ff; c += 256) {
    const float g = ur[ff + c];
    y[(size_t)blockIdx.x * ff + c] = __float2half_rn(ur[c] * (0.5f * g * (1.0f + erff(g * 0.70710678118654752f))));
  }
}

// final LayerNorm of every row of a sequence, then the mean over its rows: pooled[seq][dim]
__global__ __launch_bounds__(256) void clvp_pool_kernel(const float *__restrict__ x, const int *__restrict__ seq_start, const int *__restrict__ seq_len,
                                                        const float *__restrict__ w, const float *__restrict__ b, int dim, float *__restrict__ pooled) {
  __shared__ float red[4];
  const int s = blockIdx.x, n = seq_len[s], r0 = seq_start[s];
  float acc[4] = {0.f, 0.f, 0.f, 0.f}; // dim <= 1024: columns threadIdx.x + 256 i
  for (int r = 0; r < n; r++) {
    const float *xr = x + (size_t)(r0 + r) * dim;
    float sum = 0.f;
    for (int c = threadIdx.x; c < dim; c += 256) sum += xr[c];
    const float mean = block_sum(sum, red) / dim;
    float sq = 0.f;
    for (int c = threadIdx.x; c < dim; c += 256) sq += (xr[c] - mean) * (xr[c] - mean);
    const float rstd = rsqrtf(block_sum(sq, red) / dim + 1e-5f);
    for (int i = 0, c = threadIdx.x; c < dim; c += 256, i++) acc[i] += (xr[c] - mean) * rstd * w[c] + b[c];
  }
  for (int i = 0, c = threadIdx.x; c < dim; c += 256, i++) pooled[(size_t)s * dim + c] = acc[i] / n;
}
// ---- voice-conditioning encoder kernels ----
// mel [80][T] of every clip (concatenated) -> fp16 rows [row][128] (channels 80..127 zero): the A operand of the k = 1 init convolution
__global__ __launch_bounds__(128) void venc_mel_rows_kernel(const float *__restrict__ mel, const int *__restrict__ seq_start, const int *__restrict__ seq_len,
                                                            const long long *__restrict__ mel_off, __half *__restrict__ a16) {
  const int s = blockIdx.y, t = blockIdx.x, T = seq_len[s];
  if (t >= T) return;
  const int c = threadIdx.x;
  a16[(size_t)(seq_start[s] + t) * 128 + c] = __float2half_rn(c < 80 ? mel[mel_off[s] + (size_t)c * T + t] : 0.f);
}

// GroupNorm(32 groups of D / 32 channels, eps 1e-5, statistics over the whole clip) -> fp16; one workgroup per (clip, group)
template <int D>
__global__ __launch_bounds__(256) void venc_groupnorm_kernel(const float *__restrict__ x, const int *__restrict__ seq_start, const int *__restrict__ seq_len,
                                                             const float *__restrict__ g, const float *__restrict__ b, __half *__restrict__ y) {
  constexpr int G = D / 32, SW = 256 / G; // channels per group, rows per sweep
  __shared__ float red[4];
  const int s = blockIdx.y, grp = blockIdx.x, T = seq_len[s], r0 = seq_start[s];
  const int c = grp * G + (threadIdx.x % G), t0 = threadIdx.x / G;
  float sum = 0.f;
  for (int t = t0; t < T; t += SW) sum += x[(size_t)(r0 + t) * D + c];
  const float mean = block_sum(sum, red) / ((float)G * T);
  float sq = 0.f;
  for (int t = t0; t < T; t += SW) { const float d = x[(size_t)(r0 + t) * D + c] - mean; sq += d * d; }
  const float rstd = rsqrtf(block_sum(sq, red) / ((float)G * T) + 1e-5f);
  const float gg = g[c], bb = b[c];
  for (int t = t0; t < T; t += SW) y[(size_t)(r0 + t) * D + c] = __float2half_rn((x[(size_t)(r0 + t) * D + c] - mean) * rstd * gg + bb);
}

// im2col of a k = 3, stride 2, padding 1 convolution over the frames of every clip: output row (clip, t) = [x[2t-1] | x[2t] | x[2t+1]] (zeros outside
// the clip), fp16, row length kpad >= 3 cin (zero padded). CM: the input is the caller's channel-major mel [cin][T] per clip (mel_off),
// otherwise f32 rows [row][cin] laid out by (in_start, in_len).
template <bool CM>
__global__ __launch_bounds__(256) void dcond_im2col_kernel(const float *__restrict__ x, const long long *__restrict__ mel_off, const int *__restrict__ in_start,
                                                           const int *__restrict__ in_len, const int *__restrict__ out_start, const int *__restrict__ out_len,
                                                           int cin, int kpad, __half *__restrict__ a16) {
  const int s = blockIdx.y, t = blockIdx.x;
  if (t >= out_len[s]) return;
  const int Tin = in_len[s];
  __half *dst = a16 + (size_t)(out_start[s] + t) * kpad;
  for (int k = threadIdx.x; k < kpad; k += 256) {
    const int tap = k / cin, c = k - tap * cin, ti = 2 * t + tap - 1;
    float v = 0.f;
    if (tap < 3 && ti >= 0 && ti < Tin) v = CM ? x[mel_off[s] + (size_t)c * Tin + ti] : x[(size_t)(in_start[s] + ti) * cin + c];
    dst[k] = __float2half_rn(v);
  }
}
} // namespace

struct ClvpLayer {
  float *g_attn = nullptr, *g_ff = nullptr, *b_out = nullptr, *b_ff1 = nullptr, *b_ff2 = nullptr;
  __half *w_qkv = nullptr, *w_out = nullptr, *w_ff1 = nullptr, *w_ff2 = nullptr;
};
struct ClvpState {
  int dim = 0, depth = 0, inner = 0, ff = 0, n_text = 0, n_speech = 0;
  float temperature = 0.f;
  float *emb[2] = {nullptr, nullptr}, *norm_w[2] = {nullptr, nullptr}, *norm_b[2] = {nullptr, nullptr};
  std::vector<ClvpLayer> L[2];
  std::vector<float> proj[2]; // to_text_latent / to_speech_latent [latent][dim] (host: 16 x dim x latent multiply-adds per call)
  int latent = 0;
  std::vector<void *> owned;
  DevBuf x, y16, qkv16, att16, u32, pooled, meta;
  ~ClvpState() { for (void *p : owned) (void)hipFree(p); }
};
void clvp_free(ClvpState *s) { delete s; }

static int clvp_up(tts_ctx *ctx, ClvpState *st, const std::vector<float> &src, float **dst) {
  void *p = nullptr;
  TTS_HIP(ctx, hipMalloc(&p, src.size() * 4));
  st->owned.push_back(p);
  TTS_HIP(ctx, hipMemcpy(p, src.data(), src.size() * 4, hipMemcpyHostToDevice));
  *dst = (float *)p;
  return TTS_OK;
}
static int clvp_up16(tts_ctx *ctx, ClvpState *st, const std::vector<const std::vector<float> *> &parts, __half **dst) {
  std::vector<__half> h;
  for (auto *v : parts)
    for (float f : *v) h.push_back(__float2half_rn(f));
  void *p = nullptr;
  TTS_HIP(ctx, hipMalloc(&p, h.size() * 2));
  st->owned.push_back(p);
  TTS_HIP(ctx, hipMemcpy(p, h.data(), h.size() * 2, hipMemcpyHostToDevice));
  *dst = (__half *)p;
  return TTS_OK;
}

int clvp_load(tts_ctx *ctx, const char *path) {
  WeightFile wf;
  std::string err;
  int rc = read_weight_file(path, wf, err);
  if (rc != TTS_OK) return fail(ctx, rc, "clvp_load: %s", err.c_str());
  std::unique_ptr<ClvpState> st(new ClvpState());
  auto need = [&](const std::string &name, int64_t n0, int64_t n1) -> const HostTensor * {
    auto it = wf.t.find(name);
    if (it == wf.t.end()) { fail(ctx, TTS_ERR_FORMAT, "tensor '%s' missing from CLVP model file", name.c_str()); return nullptr; }
    if (it->second.ne[0] != n0 || it->second.ne[1] != n1 || it->second.nelem() != n0 * n1) {
      fail(ctx, TTS_ERR_FORMAT, "tensor '%s' has wrong shape in CLVP model file: got [%d, %d], expected [%d, %d]", name.c_str(),
           (int)it->second.ne[0], (int)it->second.ne[1], (int)n0, (int)n1);
      return nullptr;
    }
    return &it->second;
  };
  if (!wf.has("text_emb.weight") || !wf.has("speech_emb.weight")) return fail(ctx, TTS_ERR_FORMAT, "'%s' is not a CLVP model file", path);
  st->dim = (int)wf.t.at("text_emb.weight").ne[0];
  st->n_text = (int)wf.t.at("text_emb.weight").ne[1];
  st->n_speech = (int)wf.t.at("speech_emb.weight").ne[1];
  const std::string lp = ".transformer.attn_layers.layers.";
  while (wf.has("text_transformer" + lp + std::to_string(2 * st->depth) + ".0.g")) st->depth++;
  if (st->depth == 0) return fail(ctx, TTS_ERR_FORMAT, "no encoder layers in '%s'", path);
  st->inner = (int)wf.t.at("text_transformer" + lp + "0.1.to_q.weight").ne[1];
  st->ff = (int)wf.t.at("text_transformer" + lp + "1.1.net.3.weight").ne[0];
  st->latent = (int)wf.t.at("to_text_latent.weight").ne[1];
  const int d = st->dim, in = st->inner, ff = st->ff;
  if (d % 128 || d > 1024 || in % 128 || ff % 128 || in % CLVP_DH)
    return fail(ctx, TTS_ERR_FORMAT, "CLVP dims %d / %d / %d: multiples of 128 (dim <= 1024) expected", d, in, ff);
  const HostTensor *t;
  if (!(t = need("temperature", 1, 1))) return TTS_ERR_FORMAT;
  st->temperature = t->data[0];
  const char *encs[2] = {"text_transformer", "speech_transformer"}, *embs[2] = {"text_emb.weight", "speech_emb.weight"},
             *projs[2] = {"to_text_latent.weight", "to_speech_latent.weight"};
  size_t known = 5; // embeddings, projections, temperature
  for (int e = 0; e < 2; e++) {
    if (!(t = need(embs[e], d, e ? st->n_speech : st->n_text))) return TTS_ERR_FORMAT;
    if ((rc = clvp_up(ctx, st.get(), t->data, &st->emb[e]))) return rc;
    if (!(t = need(projs[e], d, st->latent))) return TTS_ERR_FORMAT;
    st->proj[e] = t->data;
    st->L[e].resize(st->depth);
    for (int i = 0; i < st->depth; i++) {
      ClvpLayer &l = st->L[e][i];
      const std::string a = std::string(encs[e]) + lp + std::to_string(2 * i) + ".", f = std::string(encs[e]) + lp + std::to_string(2 * i + 1) + ".";
      const HostTensor *q, *k, *v;
      if (!(t = need(a + "0.g", d, 1)) || (rc = clvp_up(ctx, st.get(), t->data, &l.g_attn))) return rc ? rc : TTS_ERR_FORMAT;
      if (!(q = need(a + "1.to_q.weight", d, in)) || !(k = need(a + "1.to_k.weight", d, in)) || !(v = need(a + "1.to_v.weight", d, in))) return TTS_ERR_FORMAT;
      if ((rc = clvp_up16(ctx, st.get(), {&q->data, &k->data, &v->data}, &l.w_qkv))) return rc; // [3 inner][dim]: one projection launch
      if (!(t = need(a + "1.to_out.weight", in, d)) || (rc = clvp_up16(ctx, st.get(), {&t->data}, &l.w_out))) return rc ? rc : TTS_ERR_FORMAT;
      if (!(t = need(a + "1.to_out.bias", d, 1)) || (rc = clvp_up(ctx, st.get(), t->data, &l.b_out))) return rc ? rc : TTS_ERR_FORMAT;
      if (!(t = need(f + "0.g", d, 1)) || (rc = clvp_up(ctx, st.get(), t->data, &l.g_ff))) return rc ? rc : TTS_ERR_FORMAT;
      if (!(t = need(f + "1.net.0.proj.weight", d, 2 * ff)) || (rc = clvp_up16(ctx, st.get(), {&t->data}, &l.w_ff1))) return rc ? rc : TTS_ERR_FORMAT;
      if (!(t = need(f + "1.net.0.proj.bias", 2 * ff, 1)) || (rc = clvp_up(ctx, st.get(), t->data, &l.b_ff1))) return rc ? rc : TTS_ERR_FORMAT;
      if (!(t = need(f + "1.net.3.weight", ff, d)) || (rc = clvp_up16(ctx, st.get(), {&t->data}, &l.w_ff2))) return rc ? rc : TTS_ERR_FORMAT;
      if (!(t = need(f + "1.net.3.bias", d, 1)) || (rc = clvp_up(ctx, st.get(), t->data, &l.b_ff2))) return rc ? rc : TTS_ERR_FORMAT;
      known += 11;
    }
    if (!(t = need(std::string(encs[e]) + ".transformer.norm.weight", d, 1)) || (rc = clvp_up(ctx, st.get(), t->data, &st->norm_w[e]))) return rc ? rc : TTS_ERR_FORMAT;
    if (!(t = need(std::string(encs[e]) + ".transformer.norm.bias", d, 1)) || (rc = clvp_up(ctx, st.get(), t->data, &st->norm_b[e]))) return rc ? rc : TTS_ERR_FORMAT;
    known += 2;
  }
  // every tensor in the file must be known (the loaders of the three reference models do the same, main.cpp:834-838); the rotary
  // inv_freq buffers of the upstream state dict are recomputed here and may be present
  size_t extra = 0;
  for (auto &kv : wf.t)
    if (kv.first.find("rotary_pos_emb.inv_freq") != std::string::npos) extra++;
  if (wf.t.size() != known + extra) return fail(ctx, TTS_ERR_FORMAT, "unknown tensors in CLVP model file '%s' (%d tensors, %d expected)", path, (int)wf.t.size(), (int)(known + extra));
  if (ctx->clvp) clvp_free(ctx->clvp);
  ctx->clvp = st.release();
  return TTS_OK;
}

// pooled latent of every sequence of one encoder: lens[n_seq] token counts, tokens concatenated
static int clvp_encode(tts_ctx *ctx, ClvpState *st, int e, const std::vector<int> &tokens, const std::vector<int> &lens, std::vector<float> &lat_out) {
  const int d = st->dim, in = st->inner, ff = st->ff, nseq = (int)lens.size();
  int rows = 0, maxlen = 0;
  std::vector<int> meta(2 * nseq);
  for (int s = 0; s < nseq; s++) { meta[s] = rows; meta[nseq + s] = lens[s]; rows += lens[s]; maxlen = std::max(maxlen, lens[s]); }
  const int M = (rows + 127) / 128 * 128;
  TTS_HIP(ctx, st->x.reserve((size_t)M * d * 4));
  TTS_HIP(ctx, st->y16.reserve((size_t)M * std::max(d, ff) * 2));
  TTS_HIP(ctx, st->qkv16.reserve((size_t)M * 3 * in * 2));
  TTS_HIP(ctx, st->att16.reserve((size_t)M * in * 2));
  TTS_HIP(ctx, st->u32.reserve((size_t)M * 2 * ff * 4));
  TTS_HIP(ctx, st->pooled.reserve((size_t)nseq * d * 4));
  TTS_HIP(ctx, st->meta.reserve((size_t)(2 * nseq + rows) * 4));
  int *d_start = st->meta.as<int>(), *d_len = d_start + nseq, *d_tok = d_len + nseq;
  TTS_HIP(ctx, hipMemcpyAsync(d_start, meta.data(), meta.size() * 4, hipMemcpyHostToDevice, ctx->stream));
  TTS_HIP(ctx, hipMemcpyAsync(d_tok, tokens.data(), (size_t)rows * 4, hipMemcpyHostToDevice, ctx->stream));
  // rows past the last token are multiplied like the others (the GEMM works on 128-row tiles) and never read back: keep them finite
  TTS_HIP(ctx, hipMemsetAsync(st->x.p, 0, (size_t)M * d * 4, ctx->stream));
  TTS_HIP(ctx, hipMemsetAsync(st->y16.p, 0, (size_t)M * std::max(d, ff) * 2, ctx->stream));
  TTS_HIP(ctx, hipMemsetAsync(st->att16.p, 0, (size_t)M * in * 2, ctx->stream));
  TTS_HIP(ctx, hipMemsetAsync(st->u32.p, 0, (size_t)M * 2 * ff * 4, ctx->stream));
  float *x = st->x.as<float>(), *u = st->u32.as<float>();
  __half *y = st->y16.as<__half>(), *qkv = st->qkv16.as<__half>(), *att = st->att16.as<__half>();
  clvp_embed_kernel<<<rows, 256, 0, ctx->stream>>>(st->emb[e], d_tok, d, x);
  auto gemm = [&](const __half *A, int K, const __half *W, int N, const float *bias) {
    GemmArgs g{};
    for (int i = 0; i < 3; i++) { g.A[i] = A; g.row_off[i] = 0; }
    g.nseg = 1; g.kseg = K; g.lda = K; g.W = W; g.M = M; g.N = N; g.bias = bias;
    return g;
  };
  for (int i = 0; i < st->depth; i++) {
    const ClvpLayer &l = st->L[e][i];
    ProfScope ps(ctx, "clvp_layer", 2.0 * rows * ((double)d * 3 * in + (double)in * d + (double)d * 2 * ff + (double)ff * d));
    clvp_rmsnorm_kernel<<<rows, 256, 0, ctx->stream>>>(x, l.g_attn, d, y);
    { GemmArgs g = gemm(y, d, l.w_qkv, 3 * in, nullptr); g.mode = GEMM_OUT_F16; g.outH = qkv; g.ldh = 3 * in; TTS_HIP(ctx, launch_gemm_f16(g, ctx->stream)); }
    clvp_attn_kernel<true, 64, false><<<dim3(nseq, in / CLVP_DH, (maxlen + 255) / 256), 256, 0, ctx->stream>>>(qkv, d_start, d_len, in, att, nullptr);
    { GemmArgs g = gemm(att, in, l.w_out, d, l.b_out); g.mode = GEMM_OUT_F32; g.outF = x; g.ldo = d; g.resid = x; TTS_HIP(ctx, launch_gemm_f16(g, ctx->stream)); }
    clvp_rmsnorm_kernel<<<rows, 256, 0, ctx->stream>>>(x, l.g_ff, d, y);
    { GemmArgs g = gemm(y, d, l.w_ff1, 2 * ff, l.b_ff1); g.mode = GEMM_OUT_F32; g.outF = u; g.ldo = 2 * ff; TTS_HIP(ctx, launch_gemm_f16(g, ctx->stream)); }
    clvp_geglu_kernel<<<rows, 256, 0, ctx->stream>>>(u, ff, y);
    { GemmArgs g = gemm(y, ff, l.w_ff2, d, l.b_ff2); g.mode = GEMM_OUT_F32; g.outF = x; g.ldo = d; g.resid = x; TTS_HIP(ctx, launch_gemm_f16(g, ctx->stream)); }
  }
  clvp_pool_kernel<<<nseq, 256, 0, ctx->stream>>>(x, d_start, d_len, st->norm_w[e], st->norm_b[e], d, st->pooled.as<float>());
  TTS_HIP(ctx, hipGetLastError());
  std::vector<float> pooled((size_t)nseq * d);
  TTS_HIP(ctx, hipMemcpyAsync(pooled.data(), st->pooled.p, pooled.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
  TTS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  // latent projection (bias-free) + L2 normalisation on the host: nseq x latent x dim multiply-adds
  lat_out.assign((size_t)nseq * st->latent, 0.f);
  for (int s = 0; s < nseq; s++) {
    double nn = 0;
    std::vector<double> z(st->latent);
    for (int o = 0; o < st->latent; o++) {
      double a = 0;
      const float *w = &st->proj[e][(size_t)o * d], *p = &pooled[(size_t)s * d];
      for (int c = 0; c < d; c++) a += (double)w[c] * p[c];
      z[o] = a; nn += a * a;
    }
    const double inv = 1.0 / std::max(std::sqrt(nn), 1e-12);
    for (int o = 0; o < st->latent; o++) lat_out[(size_t)s * st->latent + o] = (float)(z[o] * inv);
  }
  return TTS_OK;
}

int clvp_score(tts_ctx *ctx, const int32_t *text_ids, int n_text, const int32_t *codes, const int32_t *code_len, int n_candidates, int code_stride,
               float *scores_out) {
  ClvpState *st = ctx->clvp;
  if (!st) return fail(ctx, TTS_ERR_STATE, "tts_load_clvp not called");
  if (n_text < 1 || n_candidates < 1 || !text_ids || !codes || !code_len || !scores_out) return fail(ctx, TTS_ERR_ARG, "tts_clvp_score: bad arguments");
  std::vector<int> tt(text_ids, text_ids + n_text), tl{n_text}, st_tok, sl;
  for (int t : tt)
    if (t < 0 || t >= st->n_text) return fail(ctx, TTS_ERR_ARG, "text id %d out of range", t);
  for (int c = 0; c < n_candidates; c++) {
    if (code_len[c] < 1 || code_len[c] > code_stride) return fail(ctx, TTS_ERR_ARG, "candidate %d: %d codes (1 .. %d expected)", c, code_len[c], code_stride);
    for (int j = 0; j < code_len[c]; j++) {
      const int v = codes[(size_t)c * code_stride + j];
      if (v < 0 || v >= st->n_speech) return fail(ctx, TTS_ERR_ARG, "candidate %d: mel code %d out of range (start / stop tokens are not scored)", c, v);
      st_tok.push_back(v);
    }
    sl.push_back(code_len[c]);
  }
  std::vector<float> zt, zs;
  int rc = clvp_encode(ctx, st, 0, tt, tl, zt);
  if (rc) return rc;
  if ((rc = clvp_encode(ctx, st, 1, st_tok, sl, zs))) return rc;
  const double temp = std::exp((double)st->temperature);
  for (int c = 0; c < n_candidates; c++) {
    double a = 0;
    for (int o = 0; o < st->latent; o++) a += (double)zt[o] * zs[(size_t)c * st->latent + o];
    scores_out[c] = (float)(a * temp);
  }
  return TTS_OK;
}

// ------------------------------------------------------------------------------------------------------------------------------
// voice-conditioning encoder
// ------------------------------------------------------------------------------------------------------------------------------
struct VencBlock {
  float *g = nullptr, *b = nullptr, *b_qkv = nullptr, *b_proj = nullptr;
  __half *w_qkv = nullptr, *w_proj = nullptr;
};
struct VoiceEncState {
  float *b_init = nullptr;
  __half *w_init = nullptr; // [1024][128]: K padded 80 -> 128
  std::vector<VencBlock> blk;
  std::vector<void *> owned;
  DevBuf x, y16, qkv16, att16, a16, meta, mel;
  ~VoiceEncState() { for (void *p : owned) (void)hipFree(p); }
};
void voice_enc_free(VoiceEncState *s) { delete s; }

template <class T> static int venc_up(tts_ctx *ctx, VoiceEncState *st, const std::vector<T> &src, T **dst) {
  void *p = nullptr;
  TTS_HIP(ctx, hipMalloc(&p, src.size() * sizeof(T)));
  st->owned.push_back(p);
  TTS_HIP(ctx, hipMemcpy(p, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
  *dst = (T *)p;
  return TTS_OK;
}

int voice_enc_load(tts_ctx *ctx, const char *path) {
  WeightFile wf;
  std::string err;
  int rc = read_weight_file(path, wf, err);
  if (rc != TTS_OK) return fail(ctx, rc, "voice_encoder_load: %s", err.c_str());
  constexpr int D = 1024, H = 16;
  auto get = [&](const std::string &name, int64_t nelem) -> const HostTensor * {
    auto it = wf.t.find(name);
    if (it == wf.t.end()) { fail(ctx, TTS_ERR_FORMAT, "tensor '%s' missing from the conditioning-encoder file", name.c_str()); return nullptr; }
    if (it->second.nelem() != nelem) { fail(ctx, TTS_ERR_FORMAT, "tensor '%s' has %d elements, %d expected", name.c_str(), (int)it->second.nelem(), (int)nelem); return nullptr; }
    return &it->second;
  };
  if (!wf.has("conditioning_encoder.init.weight")) return fail(ctx, TTS_ERR_FORMAT, "'%s' is not a conditioning-encoder file", path);
  std::unique_ptr<VoiceEncState> st(new VoiceEncState());
  const HostTensor *t, *tb;
  if (!(t = get("conditioning_encoder.init.weight", D * 80)) || !(tb = get("conditioning_encoder.init.bias", D))) return TTS_ERR_FORMAT;
  {
    std::vector<__half> w((size_t)D * 128, __float2half_rn(0.f));
    for (int n = 0; n < D; n++)
      for (int k = 0; k < 80; k++) w[(size_t)n * 128 + k] = __float2half_rn(t->data[(size_t)n * 80 + k]);
    if ((rc = venc_up(ctx, st.get(), w, &st->w_init)) || (rc = venc_up(ctx, st.get(), tb->data, &st->b_init))) return rc;
  }
  size_t known = 2;
  for (int i = 0; wf.has("conditioning_encoder.attn." + std::to_string(i) + ".norm.weight"); i++) {
    const std::string p = "conditioning_encoder.attn." + std::to_string(i) + ".";
    VencBlock b;
    const HostTensor *g, *gb, *qw, *qb, *pw, *pb;
    if (!(g = get(p + "norm.weight", D)) || !(gb = get(p + "norm.bias", D)) || !(qw = get(p + "qkv.weight", 3 * D * D)) || !(qb = get(p + "qkv.bias", 3 * D)) ||
        !(pw = get(p + "proj_out.weight", D * D)) || !(pb = get(p + "proj_out.bias", D)))
      return TTS_ERR_FORMAT;
    // QKVAttentionLegacy: output channel = head * 192 + {q, k, v} * 64 + d. The rows are re-ordered to q | k | v blocks with head h at
    // columns h * 64 (the layout clvp_attn_kernel reads): new row t * 1024 + h * 64 + d  <-  old row h * 192 + t * 64 + d
    std::vector<__half> w((size_t)3 * D * D);
    std::vector<float> bq(3 * D);
    for (int h = 0; h < H; h++)
      for (int tq = 0; tq < 3; tq++)
        for (int d = 0; d < 64; d++) {
          const int o = h * 192 + tq * 64 + d, n = tq * D + h * 64 + d;
          bq[n] = qb->data[o];
          for (int k = 0; k < D; k++) w[(size_t)n * D + k] = __float2half_rn(qw->data[(size_t)o * D + k]);
        }
    std::vector<__half> wp((size_t)D * D);
    for (size_t k = 0; k < wp.size(); k++) wp[k] = __float2half_rn(pw->data[k]);
    if ((rc = venc_up(ctx, st.get(), g->data, &b.g)) || (rc = venc_up(ctx, st.get(), gb->data, &b.b)) || (rc = venc_up(ctx, st.get(), w, &b.w_qkv)) ||
        (rc = venc_up(ctx, st.get(), bq, &b.b_qkv)) || (rc = venc_up(ctx, st.get(), wp, &b.w_proj)) || (rc = venc_up(ctx, st.get(), pb->data, &b.b_proj)))
      return rc;
    st->blk.push_back(b);
    known += 6;
  }
  if (st->blk.empty()) return fail(ctx, TTS_ERR_FORMAT, "no attention blocks in '%s'", path);
  if (wf.t.size() != known) return fail(ctx, TTS_ERR_FORMAT, "unknown tensors in conditioning-encoder file '%s' (%d tensors, %d expected)", path, (int)wf.t.size(), (int)known);
  if (ctx->venc) voice_enc_free(ctx->venc);
  ctx->venc = st.release();
  return TTS_OK;
}

int voice_enc_latent(tts_ctx *ctx, const float *mel, const int32_t *frames, int n_clips, float *out1024) {
  VoiceEncState *st = ctx->venc;
  if (!st) return fail(ctx, TTS_ERR_STATE, "tts_load_voice_encoder not called");
  if (!mel || !frames || n_clips < 1 || !out1024) return fail(ctx, TTS_ERR_ARG, "tts_voice_latent: bad arguments");
  constexpr int D = 1024;
  int rows = 0, maxlen = 0;
  std::vector<int> meta(2 * n_clips);
  std::vector<long long> moff(n_clips);
  long long off = 0;
  for (int s = 0; s < n_clips; s++) {
    if (frames[s] < 1 || frames[s] > 16384) return fail(ctx, TTS_ERR_ARG, "clip %d: %d mel frames (1 .. 16384 expected)", s, frames[s]);
    meta[s] = rows; meta[n_clips + s] = frames[s]; moff[s] = off;
    rows += frames[s]; off += (long long)80 * frames[s]; maxlen = std::max(maxlen, frames[s]);
  }
  const int M = (rows + 127) / 128 * 128;
  TTS_HIP(ctx, st->x.reserve((size_t)M * D * 4));
  TTS_HIP(ctx, st->y16.reserve((size_t)M * D * 2));
  TTS_HIP(ctx, st->qkv16.reserve((size_t)M * 3 * D * 2));
  TTS_HIP(ctx, st->att16.reserve((size_t)M * D * 2));
  TTS_HIP(ctx, st->a16.reserve((size_t)M * 128 * 2));
  TTS_HIP(ctx, st->meta.reserve((size_t)2 * n_clips * 4 + (size_t)n_clips * 8 + 16));
  TTS_HIP(ctx, st->mel.reserve((size_t)off * 4));
  int *d_start = st->meta.as<int>(), *d_len = d_start + n_clips;
  long long *d_moff = (long long *)(st->meta.as<char>() + (((size_t)2 * n_clips * 4 + 7) & ~(size_t)7));
  TTS_HIP(ctx, hipMemcpyAsync(d_start, meta.data(), meta.size() * 4, hipMemcpyHostToDevice, ctx->stream));
  TTS_HIP(ctx, hipMemcpyAsync(d_moff, moff.data(), moff.size() * 8, hipMemcpyHostToDevice, ctx->stream));
  TTS_HIP(ctx, hipMemcpyAsync(st->mel.p, mel, (size_t)off * 4, hipMemcpyHostToDevice, ctx->stream));
  TTS_HIP(ctx, hipMemsetAsync(st->a16.p, 0, (size_t)M * 128 * 2, ctx->stream)); // rows past the last frame: finite, never read back
  TTS_HIP(ctx, hipMemsetAsync(st->y16.p, 0, (size_t)M * D * 2, ctx->stream));
  TTS_HIP(ctx, hipMemsetAsync(st->att16.p, 0, (size_t)M * D * 2, ctx->stream));
  float *x = st->x.as<float>();
  __half *y = st->y16.as<__half>(), *qkv = st->qkv16.as<__half>(), *att = st->att16.as<__half>(), *a16 = st->a16.as<__half>();
  venc_mel_rows_kernel<<<dim3(maxlen, n_clips), 128, 0, ctx->stream>>>(st->mel.as<float>(), d_start, d_len, d_moff, a16);
  auto gemm = [&](const __half *A, int K, const __half *W, int N, const float *bias) {
    GemmArgs g{};
    for (int i = 0; i < 3; i++) { g.A[i] = A; g.row_off[i] = 0; }
    g.nseg = 1; g.kseg = K; g.lda = K; g.W = W; g.M = M; g.N = N; g.bias = bias;
    return g;
  };
  { GemmArgs g = gemm(a16, 128, st->w_init, D, st->b_init); g.mode = GEMM_OUT_F32; g.outF = x; g.ldo = D; TTS_HIP(ctx, launch_gemm_f16(g, ctx->stream)); }
  for (const VencBlock &b : st->blk) {
    venc_groupnorm_kernel<1024><<<dim3(32, n_clips), 256, 0, ctx->stream>>>(x, d_start, d_len, b.g, b.b, y);
    { GemmArgs g = gemm(y, D, b.w_qkv, 3 * D, b.b_qkv); g.mode = GEMM_OUT_F16; g.outH = qkv; g.ldh = 3 * D; TTS_HIP(ctx, launch_gemm_f16(g, ctx->stream)); }
    clvp_attn_kernel<false, 64, false><<<dim3(n_clips, D / CLVP_DH, (maxlen + 255) / 256), 256, 0, ctx->stream>>>(qkv, d_start, d_len, D, att, nullptr);
    { GemmArgs g = gemm(att, D, b.w_proj, D, b.b_proj); g.mode = GEMM_OUT_F32; g.outF = x; g.ldo = D; g.resid = x; TTS_HIP(ctx, launch_gemm_f16(g, ctx->stream)); }
  }
  TTS_HIP(ctx, hipGetLastError());
  // position 0 of every clip, mean over the clips
  std::vector<float> first((size_t)n_clips * D);
  for (int s = 0; s < n_clips; s++)
    TTS_HIP(ctx, hipMemcpyAsync(&first[(size_t)s * D], x + (size_t)meta[s] * D, D * 4, hipMemcpyDeviceToHost, ctx->stream));
  TTS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  for (int c = 0; c < D; c++) {
    double a = 0;
    for (int s = 0; s < n_clips; s++) a += first[(size_t)s * D + c];
    out1024[c] = (float)(a / n_clips);
  }
  return TTS_OK;
}

// ------------------------------------------------------------------------------------------------------------------------------
// diffusion conditioning encoder (the other half of SURVEY 8 f3): upstream DiffusionTts.get_conditioning. 100-band mel of the reference clips
// -> the 2048 floats the reference reads as the weight `diffusion_conditioning_latent` of ggml-diffusion-model.bin (main.cpp:1557-1560).
// Two k = 3 / stride 2 convolutions (im2col + GEMM), five attention blocks with 2048 channels (16 heads of 128, relative position bias),
// mean over the frames of all clips.
// ------------------------------------------------------------------------------------------------------------------------------
struct DcondBlock {
  float *g = nullptr, *b = nullptr, *b_qkv = nullptr, *b_proj = nullptr, *bias_tab = nullptr;
  __half *w_qkv = nullptr, *w_proj = nullptr;
};
struct DiffCondEncState {
  __half *w0 = nullptr, *w1 = nullptr; // [1024][320] (K = 3 x 100 padded), [2048][3072]
  float *b0 = nullptr, *b1 = nullptr;
  std::vector<DcondBlock> blk;
  std::vector<void *> owned;
  DevBuf h1, x, y16, qkv16, att16, a16, meta, mel;
  ~DiffCondEncState() { for (void *p : owned) (void)hipFree(p); }
};
void diff_cond_enc_free(DiffCondEncState *s) { delete s; }

template <class T> static int dcond_up(tts_ctx *ctx, DiffCondEncState *st, const std::vector<T> &src, T **dst) {
  void *p = nullptr;
  TTS_HIP(ctx, hipMalloc(&p, src.size() * sizeof(T)));
  st->owned.push_back(p);
  TTS_HIP(ctx, hipMemcpy(p, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
  *dst = (T *)p;
  return TTS_OK;
}
// Conv1d weight [cout][cin][3] -> GEMM weight [cout][kpad], k = tap * cin + c (the im2col order), fp16
static std::vector<__half> dcond_conv_weight(const std::vector<float> &w, int cout, int cin, int kpad) {
  std::vector<__half> o((size_t)cout * kpad, __float2half_rn(0.f));
  for (int n = 0; n < cout; n++)
    for (int c = 0; c < cin; c++)
      for (int tap = 0; tap < 3; tap++) o[(size_t)n * kpad + tap * cin + c] = __float2half_rn(w[((size_t)n * cin + c) * 3 + tap]);
  return o;
}

int diff_cond_enc_load(tts_ctx *ctx, const char *path) {
  WeightFile wf;
  std::string err;
  int rc = read_weight_file(path, wf, err);
  if (rc != TTS_OK) return fail(ctx, rc, "diffusion_conditioning_encoder_load: %s", err.c_str());
  constexpr int D = 2048, H = 16, DH = 128;
  auto get = [&](const std::string &name, int64_t nelem) -> const HostTensor * {
    auto it = wf.t.find(name);
    if (it == wf.t.end()) { fail(ctx, TTS_ERR_FORMAT, "tensor '%s' missing from the diffusion-conditioning-encoder file", name.c_str()); return nullptr; }
    if (it->second.nelem() != nelem) { fail(ctx, TTS_ERR_FORMAT, "tensor '%s' has %d elements, %d expected", name.c_str(), (int)it->second.nelem(), (int)nelem); return nullptr; }
    return &it->second;
  };
  if (!wf.has("contextual_embedder.0.weight")) return fail(ctx, TTS_ERR_FORMAT, "'%s' is not a diffusion-conditioning-encoder file", path);
  std::unique_ptr<DiffCondEncState> st(new DiffCondEncState());
  const HostTensor *w0, *b0, *w1, *b1;
  if (!(w0 = get("contextual_embedder.0.weight", 1024 * 100 * 3)) || !(b0 = get("contextual_embedder.0.bias", 1024)) ||
      !(w1 = get("contextual_embedder.1.weight", (int64_t)D * 1024 * 3)) || !(b1 = get("contextual_embedder.1.bias", D)))
    return TTS_ERR_FORMAT;
  if ((rc = dcond_up(ctx, st.get(), dcond_conv_weight(w0->data, 1024, 100, 320), &st->w0)) || (rc = dcond_up(ctx, st.get(), b0->data, &st->b0)) ||
      (rc = dcond_up(ctx, st.get(), dcond_conv_weight(w1->data, D, 1024, 3072), &st->w1)) || (rc = dcond_up(ctx, st.get(), b1->data, &st->b1)))
    return rc;
  size_t known = 4;
  for (int i = 0; wf.has("contextual_embedder." + std::to_string(2 + i) + ".norm.weight"); i++) {
    const std::string p = "contextual_embedder." + std::to_string(2 + i) + ".";
    DcondBlock b;
    const HostTensor *g, *gb, *qw, *qb, *pw, *pb, *rp;
    if (!(g = get(p + "norm.weight", D)) || !(gb = get(p + "norm.bias", D)) || !(qw = get(p + "qkv.weight", (int64_t)3 * D * D)) || !(qb = get(p + "qkv.bias", 3 * D)) ||
        !(pw = get(p + "proj_out.weight", (int64_t)D * D)) || !(pb = get(p + "proj_out.bias", D)) ||
        !(rp = get(p + "relative_pos_embeddings.relative_attention_bias.weight", 32 * H)))
      return TTS_ERR_FORMAT;
    // QKVAttentionLegacy rows (head * 3 DH + {q, k, v} * DH + d) -> q | k | v blocks with head h at columns h * DH
    std::vector<__half> w((size_t)3 * D * D);
    std::vector<float> bq(3 * D);
    for (int h = 0; h < H; h++)
      for (int tq = 0; tq < 3; tq++)
        for (int d = 0; d < DH; d++) {
          const int o = h * 3 * DH + tq * DH + d, n = tq * D + h * DH + d;
          bq[n] = qb->data[o];
          for (int k = 0; k < D; k++) w[(size_t)n * D + k] = __float2half_rn(qw->data[(size_t)o * D + k]);
        }
    std::vector<__half> wp((size_t)D * D);
    for (size_t k = 0; k < wp.size(); k++) wp[k] = __float2half_rn(pw->data[k]);
    // bias by signed distance (rel_bucket(i = query, c = key): main.cpp:4722-4749 is the same T5 rule), x sqrt(head dim) as upstream's RelativePositionBias scale
    std::vector<float> tab((size_t)H * 128);
    for (int h = 0; h < H; h++)
      for (int sgn = 0; sgn < 2; sgn++)
        for (int ad = 0; ad < 64; ad++)
          tab[(size_t)h * 128 + sgn * 64 + ad] = rp->data[(size_t)rel_bucket(sgn ? 0 : ad, sgn ? ad : 0) * H + h] * 11.313708498984761f;
    if ((rc = dcond_up(ctx, st.get(), g->data, &b.g)) || (rc = dcond_up(ctx, st.get(), gb->data, &b.b)) || (rc = dcond_up(ctx, st.get(), w, &b.w_qkv)) ||
        (rc = dcond_up(ctx, st.get(), bq, &b.b_qkv)) || (rc = dcond_up(ctx, st.get(), wp, &b.w_proj)) || (rc = dcond_up(ctx, st.get(), pb->data, &b.b_proj)) ||
        (rc = dcond_up(ctx, st.get(), tab, &b.bias_tab)))
      return rc;
    st->blk.push_back(b);
    known += 7;
  }
  if (st->blk.empty()) return fail(ctx, TTS_ERR_FORMAT, "no attention blocks in '%s'", path);
  if (wf.t.size() != known) return fail(ctx, TTS_ERR_FORMAT, "unknown tensors in diffusion-conditioning-encoder file '%s' (%d tensors, %d expected)", path, (int)wf.t.size(), (int)known);
  if (ctx->dcond) diff_cond_enc_free(ctx->dcond);
  ctx->dcond = st.release();
  return TTS_OK;
}

int diff_cond_enc_latent(tts_ctx *ctx, const float *mel, const int32_t *frames, int n_clips, float *out2048) {
  DiffCondEncState *st = ctx->dcond;
  if (!st) return fail(ctx, TTS_ERR_STATE, "tts_load_diffusion_conditioning_encoder not called");
  if (!mel || !frames || n_clips < 1 || !out2048) return fail(ctx, TTS_ERR_ARG, "tts_diffusion_conditioning_latent: bad arguments");
  constexpr int D = 2048;
  // three row layouts: the mel frames (T), after the first convolution (T1 = (T - 1) / 2 + 1), after the second (T2)
  std::vector<int> meta(6 * n_clips);
  std::vector<long long> moff(n_clips);
  int r1 = 0, r2 = 0, max0 = 0;
  long long off = 0;
  for (int s = 0; s < n_clips; s++) {
    if (frames[s] < 1 || frames[s] > 16384) return fail(ctx, TTS_ERR_ARG, "clip %d: %d mel frames (1 .. 16384 expected)", s, frames[s]);
    const int T1 = (frames[s] - 1) / 2 + 1, T2 = (T1 - 1) / 2 + 1;
    meta[s] = 0; meta[n_clips + s] = frames[s];             // input of conv 0 (channel-major mel: start unused)
    meta[2 * n_clips + s] = r1; meta[3 * n_clips + s] = T1; // rows of h1
    meta[4 * n_clips + s] = r2; meta[5 * n_clips + s] = T2; // rows of x
    moff[s] = off;
    off += (long long)100 * frames[s]; r1 += T1; r2 += T2; max0 = std::max(max0, frames[s]);
  }
  const int max1 = (max0 - 1) / 2 + 1, max2 = (max1 - 1) / 2 + 1;
  const int M1 = (r1 + 127) / 128 * 128, M2 = (r2 + 127) / 128 * 128;
  TTS_HIP(ctx, st->h1.reserve((size_t)M1 * 1024 * 4));
  TTS_HIP(ctx, st->x.reserve((size_t)M2 * D * 4));
  TTS_HIP(ctx, st->a16.reserve((size_t)std::max((size_t)M1 * 320, (size_t)M2 * 3072) * 2));
  TTS_HIP(ctx, st->y16.reserve((size_t)M2 * D * 2));
  TTS_HIP(ctx, st->qkv16.reserve((size_t)M2 * 3 * D * 2));
  TTS_HIP(ctx, st->att16.reserve((size_t)M2 * D * 2));
  TTS_HIP(ctx, st->meta.reserve((size_t)6 * n_clips * 4 + (size_t)n_clips * 8 + 16));
  TTS_HIP(ctx, st->mel.reserve((size_t)off * 4));
  int *dm = st->meta.as<int>();
  long long *d_moff = (long long *)(st->meta.as<char>() + (((size_t)6 * n_clips * 4 + 7) & ~(size_t)7));
  TTS_HIP(ctx, hipMemcpyAsync(dm, meta.data(), meta.size() * 4, hipMemcpyHostToDevice, ctx->stream));
  TTS_HIP(ctx, hipMemcpyAsync(d_moff, moff.data(), moff.size() * 8, hipMemcpyHostToDevice, ctx->stream));
  TTS_HIP(ctx, hipMemcpyAsync(st->mel.p, mel, (size_t)off * 4, hipMemcpyHostToDevice, ctx->stream));
  float *h1 = st->h1.as<float>(), *x = st->x.as<float>();
  __half *a16 = st->a16.as<__half>(), *y = st->y16.as<__half>(), *qkv = st->qkv16.as<__half>(), *att = st->att16.as<__half>();
  auto gemm = [&](const __half *A, int K, const __half *W, int M, int N, const float *bias) {
    GemmArgs g{};
    for (int i = 0; i < 3; i++) { g.A[i] = A; g.row_off[i] = 0; }
    g.nseg = 1; g.kseg = K; g.lda = K; g.W = W; g.M = M; g.N = N; g.bias = bias;
    return g;
  };
  // rows past the last frame of a layout are multiplied like the others and never read back: keep them finite
  TTS_HIP(ctx, hipMemsetAsync(a16, 0, (size_t)M1 * 320 * 2, ctx->stream));
  dcond_im2col_kernel<true><<<dim3(max1, n_clips), 256, 0, ctx->stream>>>(st->mel.as<float>(), d_moff, dm, dm + n_clips, dm + 2 * n_clips, dm + 3 * n_clips, 100, 320, a16);
  { GemmArgs g = gemm(a16, 320, st->w0, M1, 1024, st->b0); g.mode = GEMM_OUT_F32; g.outF = h1; g.ldo = 1024; TTS_HIP(ctx, launch_gemm_f16(g, ctx->stream)); }
  TTS_HIP(ctx, hipMemsetAsync(a16, 0, (size_t)M2 * 3072 * 2, ctx->stream));
  dcond_im2col_kernel<false><<<dim3(max2, n_clips), 256, 0, ctx->stream>>>(h1, d_moff, dm + 2 * n_clips, dm + 3 * n_clips, dm + 4 * n_clips, dm + 5 * n_clips, 1024, 3072, a16);
  { GemmArgs g = gemm(a16, 3072, st->w1, M2, D, st->b1); g.mode = GEMM_OUT_F32; g.outF = x; g.ldo = D; TTS_HIP(ctx, launch_gemm_f16(g, ctx->stream)); }
  TTS_HIP(ctx, hipMemsetAsync(y, 0, (size_t)M2 * D * 2, ctx->stream));
  TTS_HIP(ctx, hipMemsetAsync(att, 0, (size_t)M2 * D * 2, ctx->stream));
  const int *d_start = dm + 4 * n_clips, *d_len = dm + 5 * n_clips;
  for (const DcondBlock &b : st->blk) {
    venc_groupnorm_kernel<2048><<<dim3(32, n_clips), 256, 0, ctx->stream>>>(x, d_start, d_len, b.g, b.b, y);
    { GemmArgs g = gemm(y, D, b.w_qkv, M2, 3 * D, b.b_qkv); g.mode = GEMM_OUT_F16; g.outH = qkv; g.ldh = 3 * D; TTS_HIP(ctx, launch_gemm_f16(g, ctx->stream)); }
    clvp_attn_kernel<false, 128, true><<<dim3(n_clips, 16, (max2 + 255) / 256), 256, 0, ctx->stream>>>(qkv, d_start, d_len, D, att, b.bias_tab);
    { GemmArgs g = gemm(att, D, b.w_proj, M2, D, b.b_proj); g.mode = GEMM_OUT_F32; g.outF = x; g.ldo = D; g.resid = x; TTS_HIP(ctx, launch_gemm_f16(g, ctx->stream)); }
  }
  TTS_HIP(ctx, hipGetLastError());
  std::vector<float> hx((size_t)r2 * D); // the layouts have no gaps: rows 0 .. r2 - 1 are the frames of all clips
  TTS_HIP(ctx, hipMemcpyAsync(hx.data(), x, hx.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
  TTS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  for (int c = 0; c < D; c++) {
    double a = 0;
    for (int r = 0; r < r2; r++) a += hx[(size_t)r * D + c];
    out2048[c] = (float)(a / r2);
  }
  return TTS_OK;
}

} // namespace tts

// TEST INFRASTRUCTURE — oracle math kernels (plain C++/OpenMP, f32). See orc_common.h.
// Op semantics follow SURVEY.md §3.7 (ggml CPU backend of early 2024, [ggml-unverified]).
#include "orc_common.h"
#include <algorithm>
#include <fstream>

namespace orc {

Flags g_flags;

const Tensor &Model::get(const std::string &name) const {
  auto it = t.find(name);
  if (it == t.end()) {
    fprintf(stderr, "oracle: missing tensor '%s'\n", name.c_str());
    abort();
  }
  return it->second;
}

int Model::count_layers(const std::string &prefix, const std::string &suffix) const {
  int n = 0;
  while (has(prefix + std::to_string(n) + suffix)) n++;
  return n;
}

// Legacy ggml container as read by main.cpp:811-888: u32 magic 0x67676d6c, then records
// {i32 n_dims, i32 name_len, i32 ttype, i32 ne[n_dims], name, raw data} until EOF.
Model *load_model(const char *path, std::string &err) {
  std::ifstream fin(path, std::ios::binary);
  if (!fin) { err = std::string("cannot open ") + path; return nullptr; }
  uint32_t magic = 0;
  fin.read((char *)&magic, 4);
  if (magic != 0x67676d6cu) { err = "bad magic"; return nullptr; }
  Model *m = new Model();
  while (true) {
    int32_t n_dims, length, ttype;
    fin.read((char *)&n_dims, 4);
    fin.read((char *)&length, 4);
    fin.read((char *)&ttype, 4);
    if (fin.eof()) break;
    if (n_dims < 1 || n_dims > 4 || length <= 0 || length > 4096 || ttype != 0) {
      err = "bad record header";
      delete m;
      return nullptr;
    }
    Tensor t;
    t.n_dims = n_dims;
    for (int i = 0; i < n_dims; i++) {
      int32_t v;
      fin.read((char *)&v, 4);
      t.ne[i] = v;
    }
    std::string name(length, 0);
    fin.read(&name[0], length);
    t.data.resize(t.nelem());
    fin.read((char *)t.data.data(), sizeof(float) * t.nelem());
    if (!fin) { err = "truncated tensor " + name; delete m; return nullptr; }
    m->t[name] = std::move(t);
  }
  return m;
}

void transpose(const float *in, int rows, int cols, float *out) {
#pragma omp parallel for schedule(static)
  for (int c = 0; c < cols; c++)
    for (int r = 0; r < rows; r++) out[(size_t)c * rows + r] = in[(size_t)r * cols + c];
}

void gemm_kn(int M, int N, int K, const float *A, int lda, const float *Bt, int ldb, float *C,
             int ldc, const float *bias) {
  const int MB = 8, NB = 512;
  int mblocks = (M + MB - 1) / MB, nblocks = (N + NB - 1) / NB;
#pragma omp parallel for collapse(2) schedule(dynamic)
  for (int mb = 0; mb < mblocks; mb++)
    for (int nb = 0; nb < nblocks; nb++) {
      int m0 = mb * MB, m1 = std::min(M, m0 + MB);
      int n0 = nb * NB, n1 = std::min(N, n0 + NB);
      float acc[MB][NB];
      for (int i = 0; i < m1 - m0; i++)
        for (int j = 0; j < n1 - n0; j++) acc[i][j] = 0.f;
      for (int k = 0; k < K; k++) {
        const float *brow = Bt + (size_t)k * ldb + n0;
        for (int i = 0; i < m1 - m0; i++) {
          float a = A[(size_t)(m0 + i) * lda + k];
          float *ar = acc[i];
#pragma omp simd
          for (int j = 0; j < n1 - n0; j++) ar[j] += a * brow[j];
        }
      }
      for (int i = 0; i < m1 - m0; i++)
        for (int j = 0; j < n1 - n0; j++)
          C[(size_t)(m0 + i) * ldc + n0 + j] = acc[i][j] + (bias ? bias[n0 + j] : 0.f);
    }
}

void gemm_nk(int M, int N, int K, const float *A, int lda, const float *B, int ldb, float *C,
             int ldc, const float *bias) {
  std::vector<float> bt((size_t)K * N);
#pragma omp parallel for schedule(static)
  for (int k = 0; k < K; k++)
    for (int n = 0; n < N; n++) bt[(size_t)k * N + n] = B[(size_t)n * ldb + k];
  gemm_kn(M, N, K, A, lda, bt.data(), N, C, ldc, bias);
}

// ggml_norm: mean and variance with double accumulators, y = (x-mean)/sqrt(var+eps), then g,b.
void layernorm_rows(float *x, int rows, int C, float eps, const float *g, const float *b) {
#pragma omp parallel for schedule(static)
  for (int r = 0; r < rows; r++) {
    float *p = x + (size_t)r * C;
    double sum = 0;
    for (int i = 0; i < C; i++) sum += (double)p[i];
    float mean = (float)(sum / C);
    double sum2 = 0;
    for (int i = 0; i < C; i++) {
      float v = p[i] - mean;
      p[i] = v;
      sum2 += (double)(v * v);
    }
    float variance = (float)(sum2 / C);
    float scale = 1.0f / sqrtf(variance + eps);
    for (int i = 0; i < C; i++) {
      float y = p[i] * scale;
      if (g) y = y * g[i];
      if (b) y = y + b[i];
      p[i] = y;
    }
  }
}

// ggml_group_norm over x[W=T,H=1,C] with G groups: statistics over T*(C/G) elements.
void groupnorm_tc(const float *x, int T, int C, int G, float eps, const float *g, const float *b,
                  float *y) {
  int cpg = C / G;
#pragma omp parallel for schedule(static)
  for (int grp = 0; grp < G; grp++) {
    double sum = 0;
    for (int t = 0; t < T; t++)
      for (int c = grp * cpg; c < (grp + 1) * cpg; c++) sum += (double)x[(size_t)t * C + c];
    float mean = (float)(sum / ((double)T * cpg));
    double sum2 = 0;
    for (int t = 0; t < T; t++)
      for (int c = grp * cpg; c < (grp + 1) * cpg; c++) {
        float v = x[(size_t)t * C + c] - mean;
        sum2 += (double)(v * v);
      }
    float variance = (float)(sum2 / ((double)T * cpg));
    float scale = 1.0f / sqrtf(variance + eps);
    for (int t = 0; t < T; t++)
      for (int c = grp * cpg; c < (grp + 1) * cpg; c++) {
        float v = (x[(size_t)t * C + c] - mean) * scale;
        if (g) v = v * g[c];
        if (b) v = v + b[c];
        y[(size_t)t * C + c] = v;
      }
  }
}

void softmax_row(float *x, int n) {
  float mx = -INFINITY;
  for (int i = 0; i < n; i++) mx = std::max(mx, x[i]);
  double sum = 0;
  for (int i = 0; i < n; i++) {
    float v;
    if (x[i] == -INFINITY) v = 0.f;
    else if (g_flags.lut) v = f16r(expf(f16r(x[i] - mx)));
    else v = expf(x[i] - mx);
    x[i] = v;
    sum += (double)v;
  }
  float inv = (float)(1.0 / sum);
  for (int i = 0; i < n; i++) x[i] *= inv;
}

float gelu_f(float x) {
  const float GELU_COEF_A = 0.044715f, SQRT_2_OVER_PI = 0.79788456080286535587989211986876f;
  if (g_flags.lut) {
    float xr = f16r(x);
    return f16r(0.5f * xr * (1.0f + tanhf(SQRT_2_OVER_PI * xr * (1.0f + GELU_COEF_A * xr * xr))));
  }
  return 0.5f * x * (1.0f + tanhf(SQRT_2_OVER_PI * x * (1.0f + GELU_COEF_A * x * x)));
}

float silu_f(float x) {
  if (g_flags.lut) {
    float xr = f16r(x);
    return f16r(xr / (1.0f + expf(-xr)));
  }
  return x / (1.0f + expf(-x));
}

void conv1d_f16(const float *x, int T, int Cin, const float *w, int K, int Cout,
                const float *bias, int pad, int dil, float *y) {
  int Tout = T + 2 * pad - dil * (K - 1);
  int KK = Cin * K;
  // im2col (values rounded to f16): col[t][k*Cin + ci] = x[t + k*dil - pad][ci]
  std::vector<float> col((size_t)Tout * KK);
#pragma omp parallel for schedule(static)
  for (int t = 0; t < Tout; t++)
    for (int k = 0; k < K; k++) {
      int ts = t + k * dil - pad;
      float *dst = col.data() + (size_t)t * KK + (size_t)k * Cin;
      if (ts < 0 || ts >= T) {
        for (int ci = 0; ci < Cin; ci++) dst[ci] = 0.f;
      } else {
        const float *src = x + (size_t)ts * Cin;
        for (int ci = 0; ci < Cin; ci++) dst[ci] = f16r(src[ci]);
      }
    }
  // weights rounded to f16, rearranged to [k*Cin+ci][co]
  std::vector<float> wt((size_t)KK * Cout);
#pragma omp parallel for schedule(static)
  for (int co = 0; co < Cout; co++)
    for (int ci = 0; ci < Cin; ci++)
      for (int k = 0; k < K; k++)
        wt[((size_t)k * Cin + ci) * Cout + co] = f16r(w[((size_t)co * Cin + ci) * K + k]);
  gemm_kn(Tout, Cout, KK, col.data(), KK, wt.data(), Cout, y, Cout, bias);
}

} // namespace orc

// ---- op-level probes (tests/test_oracle_vs_torch.py: every op against torch's) -------------------------------------
using namespace orc;
extern "C" {
void orc_op_conv1d_f16(const float *x, int T, int Cin, const float *w, int K, int Cout, const float *bias, int pad, int dil,
                       float *y) { conv1d_f16(x, T, Cin, w, K, Cout, bias, pad, dil, y); }
void orc_op_groupnorm(const float *x, int T, int C, int G, float eps, const float *g, const float *b, float *y) {
  groupnorm_tc(x, T, C, G, eps, g, b, y);
}
void orc_op_layernorm(float *x, int rows, int C, float eps, const float *g, const float *b) { layernorm_rows(x, rows, C, eps, g, b); }
void orc_op_unary(int which, float *x, int64_t n) { // 0 gelu (tanh form), 1 silu
  for (int64_t i = 0; i < n; i++) x[i] = which == 0 ? gelu_f(x[i]) : silu_f(x[i]);
}
void orc_op_softmax(float *x, int n) { softmax_row(x, n); }
void orc_op_gemm_kn(int M, int N, int K, const float *A, const float *Bt, const float *bias, float *C) {
  gemm_kn(M, N, K, A, K, Bt, N, C, N, bias);
}
}

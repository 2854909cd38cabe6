// TEST INFRASTRUCTURE — CPU restatement ("oracle") of balisujohn/tortoise.cpp's hot path.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library;
// the product (tortoise.cpp_amd/) never links or calls it.
//
// PARITY STATUS: host-side pieces (RNG, sampler, buckets, schedule, tokenizer, padding/trim) are
// PINNED against the real reference code (oracle/_ref/libref.so, built from /root/reference) and
// against the reference's RNG fixtures. The three network stages follow the reference's ggml
// graphs (main.cpp:2053-4483) plus the ggml op semantics listed in SURVEY.md §3.7, but ggml
// itself and the trained weights are absent from the checkout => "parity unpinned" AGAINST THE REFERENCE for every
// weight-dependent tensor. What is pinned instead (round 2, tests/test_oracle_vs_torch.py): every op and the
// three whole graphs against PyTorch on the CPU, so only the ggml fork's constants (GroupNorm eps, fp16 activation tables) remain guesses.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

namespace orc {

// ---- fp16 round trip (ggml_cpy F32->F16->F32; ggml's im2col F16 output) -------------------
static inline uint16_t f32_to_f16_bits(float f) {
  uint32_t x;
  std::memcpy(&x, &f, 4);
  uint32_t sign = (x >> 16) & 0x8000u;
  uint32_t ax = x & 0x7fffffffu;
  if (ax >= 0x7f800000u) return (uint16_t)(sign | (ax > 0x7f800000u ? 0x7e00u : 0x7c00u));
  if (ax >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u); // rounds to >= 65520 -> inf
  if (ax < 0x33000001u) return (uint16_t)sign;               // < 2^-25 (and ==2^-25 ties to 0)
  int e = (int)(ax >> 23) - 127;
  uint32_t m = (ax & 0x7fffffu) | 0x800000u;
  int shift;
  uint32_t base;
  if (e < -14) { shift = 13 + (-14 - e); base = 0; }
  else { shift = 13; base = (uint32_t)(e + 15) << 10; m &= 0x7fffffu; }
  uint32_t r = m >> shift;
  uint32_t rem = m & ((1u << shift) - 1);
  uint32_t half = 1u << (shift - 1);
  if (rem > half || (rem == half && (r & 1))) r++;
  return (uint16_t)(sign | (base + r));
}
static inline float f16_bits_to_f32(uint16_t h) {
  uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
  uint32_t e = (h >> 10) & 0x1f, m = h & 0x3ff, x;
  if (e == 0) {
    if (m == 0) x = sign;
    else {
      int s = 0;
      while (!(m & 0x400)) { m <<= 1; s++; }
      m &= 0x3ff;
      x = sign | ((uint32_t)(127 - 15 - s + 1) << 23) | (m << 13);
    }
  } else if (e == 31) x = sign | 0x7f800000u | (m << 13);
  else x = sign | ((e + 112) << 23) | (m << 13);
  float f;
  std::memcpy(&f, &x, 4);
  return f;
}
static inline float f16r(float f) { return f16_bits_to_f32(f32_to_f16_bits(f)); }

// ---- runtime switches for the [ggml-unverified] op details ---------------
struct Flags {
  float gn_eps = 1e-6f; // ggml_group_norm's hard-coded eps of that era (PyTorch: 1e-5)
  int lut = 0;          // 1: emulate ggml CPU fp16 lookup tables for gelu/silu/soft_max exp
};
extern Flags g_flags;

// ---- tensor container + legacy ggml file (main.cpp:811-888) -------------------------------
struct Tensor {
  std::vector<float> data;
  int64_t ne[4] = {1, 1, 1, 1};
  int n_dims = 0;
  int64_t nelem() const { return ne[0] * ne[1] * ne[2] * ne[3]; }
};
struct Model {
  std::map<std::string, Tensor> t;
  const Tensor &get(const std::string &name) const;
  const float *p(const std::string &name) const { return get(name).data.data(); }
  bool has(const std::string &name) const { return t.count(name) != 0; }
  int count_layers(const std::string &prefix, const std::string &suffix) const;
};
Model *load_model(const char *path, std::string &err);

// ---- math helpers (orc_math.cpp) ---------------------------------------------------------
// C[M,N] = A[M,K] * Bt[K,N] (+ bias[N] if non-null); f32, k-ordered accumulation.
void gemm_kn(int M, int N, int K, const float *A, int lda, const float *Bt, int ldb, float *C,
             int ldc, const float *bias);
// C[M,N] = A[M,K] * B[N,K]^T (+bias). Internally transposes B once.
void gemm_nk(int M, int N, int K, const float *A, int lda, const float *B, int ldb, float *C,
             int ldc, const float *bias);
void transpose(const float *in, int rows, int cols, float *out); // out[cols][rows]
void layernorm_rows(float *x, int rows, int C, float eps, const float *g, const float *b);
// GroupNorm over [T][C] (C contiguous) for one sequence, G groups, then *g + b.
void groupnorm_tc(const float *x, int T, int C, int G, float eps, const float *g, const float *b,
                  float *y);
void softmax_row(float *x, int n); // ggml soft_max semantics (max-subtracted, double sum)
float gelu_f(float x);
float silu_f(float x);
// conv1d, ggml semantics: weights rounded to f16, input rounded to f16 (im2col), f32 accumulate.
//   x [T][Cin] -> y [Tout][Cout]; w in file layout ne=[K,Cin,Cout] i.e. w[(co*Cin+ci)*K + k].
void conv1d_f16(const float *x, int T, int Cin, const float *w, int K, int Cout,
                const float *bias, int pad, int dil, float *y);

} // namespace orc

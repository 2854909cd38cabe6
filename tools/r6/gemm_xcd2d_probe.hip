// Developer tool (round 6): ONE traffic experiment on the two largest GEMM shapes of the benchmark (M = 28 032 packed rows): the product's tile walk
// (an XCD owns 1/8 of the rows and every column tile, column chunks outermost) against a SPATIAL two-dimensional partition — R row ranges x C column ranges, one per XCD —
// in which an XCD's share of the weight matrix (conv3: 6.3 MB / C) can stay resident in its 4 MB L2 while its activations stream once per column tile of the range.
// Same kernels, explicit per-XCD tile tables (GemmArgs::tiles): results must be bit-identical. Prints us / launch; run under
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE   for the bytes (variants are launched in the printed order, REPS launches each).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -I tortoise.cpp_amd/csrc -I include tools/r6/gemm_xcd2d_probe.hip -o tools/bin/gemm_xcd2d_probe
#include "gemm_f16.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
using namespace tts;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
static float frand(float s) { return ((rand() % 20001) - 10000) / 10000.f * s; }

// R x C partition (R * C == 8): XCD x = r * C + c owns row tiles [219 r / R, 219 (r + 1) / R) and column tiles [NT c / C, NT (c + 1) / C); row tile outermost
// (col_major = false) or column tile outermost
static std::vector<int4> make_table(int M, int NT, int R, int C, bool col_major, int &tab_len) {
  const int MT = M / 128;
  std::vector<std::vector<int4>> per(8);
  for (int x = 0; x < 8; x++) {
    const int r = x / C, c = x % C;
    const int t0 = MT * r / R, t1 = MT * (r + 1) / R, c0 = NT * c / C, c1 = NT * (c + 1) / C;
    if (!col_major) { for (int t = t0; t < t1; t++) for (int n = c0; n < c1; n++) per[x].push_back(make_int4(t * 128, 8, n * 128, 0)); }
    else { for (int n = c0; n < c1; n++) for (int t = t0; t < t1; t++) per[x].push_back(make_int4(t * 128, 8, n * 128, 0)); }
  }
  tab_len = 0;
  for (auto &v : per) tab_len = std::max(tab_len, (int)v.size());
  std::vector<int4> tab((size_t)8 * tab_len, make_int4(0, 0, 0, 0));
  for (int x = 0; x < 8; x++) std::copy(per[x].begin(), per[x].end(), tab.begin() + (size_t)x * tab_len);
  return tab;
}

int main(int argc, char **argv) {
  const int M = 28032, C1 = 1024, REPS = argc > 1 ? atoi(argv[1]) : 20;
  srand(3);
  std::vector<__half> A((size_t)(M + 2) * C1), W((size_t)3072 * 3 * C1);
  for (auto &v : A) v = __float2half(frand(1.f));
  for (size_t i = 0; i < (size_t)C1; i++) { A[i] = __float2half(0.f); A[(size_t)(M + 1) * C1 + i] = __float2half(0.f); }
  for (auto &v : W) v = __float2half(frand(0.05f));
  std::vector<float> R((size_t)M * C1), bias(3072);
  for (auto &v : R) v = frand(1.f);
  for (auto &v : bias) v = frand(0.2f);
  std::vector<int> seq(M, 0);
  for (int i = 0; i < M; i += 876) seq[i] = -1;
  __half *dA, *dW, *dQK, *dVt; float *dR, *dB, *dOut; int *dSeq; int4 *dTab;
  CK(hipMalloc(&dA, A.size() * 2)); CK(hipMalloc(&dW, W.size() * 2)); CK(hipMalloc(&dR, R.size() * 4)); CK(hipMalloc(&dB, 3072 * 4)); CK(hipMalloc(&dOut, (size_t)M * C1 * 4));
  CK(hipMalloc(&dQK, (size_t)(M + 128) * 2048 * 2)); CK(hipMalloc(&dVt, (size_t)C1 * (M + 128) * 2)); CK(hipMalloc(&dSeq, M * 4)); CK(hipMalloc(&dTab, 8 * 1024 * 16));
  CK(hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dW, W.data(), W.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(dR, R.data(), R.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, bias.data(), 3072 * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dSeq, seq.data(), M * 4, hipMemcpyHostToDevice));
  hipStream_t s; CK(hipStreamCreate(&s));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  struct Shape { const char *name; int N, nseg, mode; };
  const Shape shapes[] = {{"conv3 + resid  N1024 K3x1024", 1024, 3, GEMM_OUT_F32}, {"qkv            N3072 K1024  ", 3072, 1, GEMM_OUT_QKV}, {"k1 (in_layers) N1024 K1024  ", 1024, 1, GEMM_OUT_F32}};
  struct Var { const char *name; int R, C; bool col_major; };
  const Var vars[] = {{"product walk (8 row ranges x all columns, L2-sized column chunks)", 0, 0, false}, {"4 row ranges x 2 column halves, row tile outermost", 4, 2, false},
                      {"4 x 2, column tile outermost", 4, 2, true}, {"2 row ranges x 4 column quarters, row tile outermost", 2, 4, false}, {"2 x 4, column tile outermost", 2, 4, true},
                      {"8 x 1 through a table (control: the product's partition without column chunks)", 8, 1, false}};
  printf("# M = %d, %d launches per variant, in this order (for the counter pass)\n", M, REPS);
  for (const Shape &sh : shapes) {
    std::vector<float> base; std::vector<__half> baseqk;
    for (const Var &v : vars) {
      GemmArgs g{};
      for (int i = 0; i < 3; i++) { g.A[i] = dA + C1; g.row_off[i] = sh.nseg == 3 ? i - 1 : 0; }
      g.nseg = sh.nseg; g.kseg = C1; g.lda = C1; g.W = dW; g.M = M; g.N = sh.N; g.bias = dB; g.row_seq = dSeq; g.mode = sh.mode;
      g.outF = dOut; g.ldo = C1; g.resid = sh.nseg == 3 ? dR : nullptr; g.outH = dQK; g.ldh = 2048; g.outVt = dVt; g.ldvt = M + 128;
      if (v.R) {
        int tl; const std::vector<int4> tab = make_table(M, sh.N / 128, v.R, v.C, v.col_major, tl);
        if (tl > 1024) { printf("table too long\n"); return 1; }
        CK(hipMemcpy(dTab, tab.data(), tab.size() * 16, hipMemcpyHostToDevice));
        g.tiles = dTab; g.tab_len = tl;
      }
      CK(hipMemsetAsync(dOut, 0xff, (size_t)M * C1 * 4, s)); CK(hipMemsetAsync(dQK, 0xff, (size_t)(M + 128) * 2048 * 2, s));
      CK(launch_gemm_f16(g, s)); CK(hipStreamSynchronize(s));
      std::vector<float> o((size_t)M * C1); std::vector<__half> oqk((size_t)M * 2048);
      CK(hipMemcpy(o.data(), dOut, o.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(oqk.data(), dQK, oqk.size() * 2, hipMemcpyDeviceToHost));
      if (!v.R) { base = o; baseqk = oqk; }
      const bool same = sh.mode == GEMM_OUT_QKV ? !memcmp(oqk.data(), baseqk.data(), oqk.size() * 2) : !memcmp(o.data(), base.data(), o.size() * 4);
      for (int i = 0; i < 3; i++) CK(launch_gemm_f16(g, s));
      CK(hipEventRecord(e0, s));
      for (int i = 0; i < REPS; i++) CK(launch_gemm_f16(g, s));
      CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      printf("%s | %-82s %8.1f us  %s\n", sh.name, v.name, 1000.0 * ms / REPS, same ? "bit-identical" : "DIFFERS");
      fflush(stdout);
    }
  }
  return 0;
}

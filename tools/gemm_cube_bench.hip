// Developer tool (round 4): calibration of the fp16 GEMM kernels on CUBES. The product's 128 x 128 x 64 kernel
// (csrc/gemm_f16.h) and the 256-column 8-phase kernel of round 3 (tools/gemm_f16_big.h) on 4096^3 and 8192^3 with uniform random [-1, 1)
// operands, printed beside the guide's reference table (cdna_hip_programming.md "Reference targets": 128^2 + XCD swizzle 912 / 948 TF on
// zero-filled operands, the 256^2 8-phase template ~1330 / ~1470 TF on random operands), and on the benchmark's own shapes, so that
// "my shapes are hard" (K = 1024, N = 1024, f32 epilogue) and "my 8-phase kernel is slow" can be told apart.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -I tortoise.cpp_amd/csrc -I tools -I tools/attic tools/gemm_cube_bench.hip -o tools/bin/gemm_cube_bench
//   tools/bin/gemm_cube_bench [zero]        ("zero": zero-filled operands as well, the guide's headline fill)
#include "gemm_f16.h"
#include "gemm_f16_big.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
using namespace tts;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

struct Shape { const char *name; int M, N, K, nseg, mode; };

int main(int argc, char **argv) {
  const bool with_zero = argc > 1 && !strcmp(argv[1], "zero");
  const std::vector<Shape> shapes = {
      {"cube 4096^3 f32 out", 4096, 4096, 4096, 1, GEMM_OUT_F32},   {"cube 4096^3 f16 out", 4096, 4096, 4096, 1, GEMM_OUT_F16},
      {"cube 8192^3 f32 out", 8192, 8192, 8192, 1, GEMM_OUT_F32},   {"cube 8192^3 f16 out", 8192, 8192, 8192, 1, GEMM_OUT_F16},
      {"M28672 N1024 K1024 f32 out (in_layers)", 28672, 1024, 1024, 1, GEMM_OUT_F32},
      {"M28672 N1024 K1024 f16 out", 28672, 1024, 1024, 1, GEMM_OUT_F16},
      {"M28672 N1024 K8192 f32 out (same M, N; long K)", 28672, 1024, 8192, 1, GEMM_OUT_F32},
      {"M28672 N3072 K1024 f16 out (QKV shape, plain store)", 28672, 3072, 1024, 1, GEMM_OUT_F16},
      {"M28672 N1024 K3x1024 f32 out (conv3 as 3 segments)", 28672, 1024, 1024, 3, GEMM_OUT_F32},
      {"M8192 N8192 K1024 f32 out (cube-like M, N; K = 1024)", 8192, 8192, 1024, 1, GEMM_OUT_F32},
  };
  size_t maxA = 0, maxW = 0, maxC = 0;
  for (auto &s : shapes) {
    maxA = std::max(maxA, (size_t)(s.M + 2) * s.K * s.nseg); maxW = std::max(maxW, (size_t)s.N * s.K * s.nseg); maxC = std::max(maxC, (size_t)s.M * s.N);
  }
  std::vector<__half> hA(maxA), hW(maxW);
  srand(1);
  for (auto &v : hA) v = __float2half((rand() % 20001 - 10000) / 10000.f);
  for (auto &v : hW) v = __float2half((rand() % 20001 - 10000) / 10000.f);
  __half *dA, *dW, *dH; float *dC, *dC2;
  CK(hipMalloc(&dA, maxA * 2)); CK(hipMalloc(&dW, maxW * 2)); CK(hipMalloc(&dC, maxC * 4)); CK(hipMalloc(&dC2, maxC * 4)); CK(hipMalloc(&dH, maxC * 2));
  hipStream_t s; CK(hipStreamCreate(&s));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("# guide table (bf16, cdna_hip_programming.md): 128^2 + XCD swizzle 912 @4k / 948 @8k TF (zero-filled operands; random operands run ~15-21 %% lower);\n"
         "#   256^2 8-phase + st_16x32 swizzle 1563 / 1728 zero-filled = ~1330 / ~1470 on uniform random [-1, 1)\n");
  for (int fill = 0; fill < (with_zero ? 2 : 1); fill++) {
    if (fill == 0) { CK(hipMemcpy(dA, hA.data(), maxA * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dW, hW.data(), maxW * 2, hipMemcpyHostToDevice)); }
    else { CK(hipMemset(dA, 0, maxA * 2)); CK(hipMemset(dW, 0, maxW * 2)); }
    printf("## operands: %s\n", fill == 0 ? "uniform random [-1, 1)" : "zero-filled");
    for (const Shape &sh : shapes) {
      auto mk = [&](float *outF) {
        GemmArgs g{};
        const int lda = sh.K * (sh.nseg == 3 ? 1 : 1);
        for (int i = 0; i < 3; i++) { g.A[i] = dA + lda; g.row_off[i] = sh.nseg == 3 ? i - 1 : 0; }
        g.nseg = sh.nseg; g.kseg = sh.K; g.lda = lda; g.W = dW; g.M = sh.M; g.N = sh.N; g.bias = nullptr; g.row_seq = nullptr;
        g.mode = sh.mode; g.outF = outF; g.ldo = sh.N; g.resid = nullptr; g.outH = dH; g.ldh = sh.N;
        return g;
      };
      const double fl = 2.0 * sh.M * sh.N * (double)sh.K * sh.nseg;
      // correctness: the two kernels against each other (f32 out), and a few entries against an f64 host sum
      double worst = 0; size_t differ = 0;
      if (fill == 0 && sh.mode == GEMM_OUT_F32) {
        GemmArgs g1 = mk(dC), g2 = mk(dC2);
        CK(launch_gemm_f16(g1, s)); CK(launch_gemm_f16_big(g2, s)); CK(hipStreamSynchronize(s));
        std::vector<float> c1((size_t)sh.M * sh.N), c2(c1.size());
        CK(hipMemcpy(c1.data(), dC, c1.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(c2.data(), dC2, c2.size() * 4, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < c1.size(); i++) differ += fabsf(c1[i] - c2[i]) > 1e-3f * (1.f + fabsf(c1[i]));
        for (int t = 0; t < 64; t++) {
          const int m = (t * 7919) % sh.M, n = (t * 104729) % sh.N;
          double acc = 0;
          for (int seg = 0; seg < sh.nseg; seg++)
            for (int k = 0; k < sh.K; k++)
              acc += (double)__half2float(hA[(size_t)(m + 1 + (sh.nseg == 3 ? seg - 1 : 0)) * sh.K + k]) * (double)__half2float(hW[(size_t)n * sh.K * sh.nseg + (size_t)seg * sh.K + k]);
          worst = std::max(worst, fabs(acc - c1[(size_t)m * sh.N + n]) / (1.0 + fabs(acc)));
          worst = std::max(worst, fabs(acc - c2[(size_t)m * sh.N + n]) / (1.0 + fabs(acc)));
        }
      }
      double us[2][5];
      for (int r = 0; r < 5; r++)
        for (int v = 0; v < 2; v++) {
          GemmArgs g = mk(dC2);
          auto go = [&]() { return v == 0 ? launch_gemm_f16(g, s) : launch_gemm_f16_big(g, s); };
          const int iters = fl > 5e11 ? 6 : 20;
          for (int i = 0; i < 2; i++) CK(go());
          CK(hipEventRecord(e0, s));
          for (int i = 0; i < iters; i++) CK(go());
          CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
          float ms; CK(hipEventElapsedTime(&ms, e0, e1));
          us[v][r] = 1000.0 * ms / iters;
        }
      printf("== %-58s %8.1f GFLOP", sh.name, fl * 1e-9);
      if (fill == 0 && sh.mode == GEMM_OUT_F32) printf("   [check: %zu entries differ between the kernels, worst rel err vs f64 %.1e]", differ, worst);
      printf("\n");
      for (int v = 0; v < 2; v++) {
        std::sort(us[v], us[v] + 5);
        printf("   %-28s med %8.1f us  min %8.1f us   %7.1f TF/s (med)  %7.1f (best)\n", v == 0 ? "128x128x64 product kernel" : "256-column 8-phase (big.h)", us[v][2], us[v][0],
               fl / (us[v][2] * 1e-6) / 1e12, fl / (us[v][0] * 1e-6) / 1e12);
      }
    }
  }
  return 0;
}

// Developer tool: small-problem GEMMs (a single utterance: M = 1 792 rows; 2 and 4 candidates) on the large-problem kernels (64-row tiles) versus
// the deep-ring kernel (gemm_f16_deep_kernel, 3 or 6 stages). Prints us / launch and checks that the deep kernel's output is BIT-IDENTICAL to the
// large-problem kernels' (a candidate's result must not depend on the batch it runs in) and close to a naive kernel.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -I tortoise.cpp_amd/csrc -I tools tools/gemm_small_diag.hip -o tools/bin/gemm_small_diag
#define TTS_GEMM_DEEP 1
#include "gemm_f16_onetile.h" // the round-2 one-tile-per-workgroup kernels these tools were written against
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
using namespace tts;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void naive_kernel(const __half *A, int lda, const __half *W, int ldw, int nseg, int kseg, int M, int N, const float *resid, float *C) {
  int n = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y;
  if (n >= N) return;
  float acc = 0;
  for (int s = 0; s < nseg; s++)
    for (int k = 0; k < kseg; k++)
      acc += __half2float(A[(size_t)(m + (nseg == 3 ? s - 1 : 0)) * lda + k]) * __half2float(W[(size_t)n * ldw + s * kseg + k]);
  C[(size_t)m * N + n] = acc + (resid ? resid[(size_t)m * N + n] : 0.f);
}

__global__ void touch_kernel(const uint4 *p, size_t n, unsigned *sink) {
  unsigned a = 0;
  for (size_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a ^= p[i].x;
  if (a == 0x12345678u) *sink = a;
}

struct Shape { const char *name; int N, K, nseg, mode, resid; };

int main() {
  const int Mmax = 7168, Kmax = 1024;
  const std::vector<Shape> shapes = {
      {"in_layers  k1 N1024 K1024       ", 1024, 1024, 1, GEMM_OUT_F32, 0}, {"proj_out   k1 N1024 K1024 +resid", 1024, 1024, 1, GEMM_OUT_F32, 1},
      {"qkv        k1 N3072 K1024       ", 3072, 1024, 1, GEMM_OUT_QKV, 0}, {"out_layers k3 N1024 K3x1024 +res", 1024, 1024, 3, GEMM_OUT_F32, 1},
      {"cond       k3 N1024 K3x1024 f16 ", 1024, 1024, 3, GEMM_OUT_F16, 0},
  };
  std::vector<__half> hA((size_t)(Mmax + 130) * Kmax), hW((size_t)3072 * 3 * Kmax);
  srand(1);
  for (auto &v : hA) v = __float2half((rand() % 2001 - 1000) / 1000.f);
  for (auto &v : hW) v = __float2half((rand() % 2001 - 1000) / 4000.f);
  __half *dA, *dW, *dH, *dVt; float *dC, *dR, *dRes, *dBias; int *dSeq;
  const size_t nC = (size_t)Mmax * 1024, nH = (size_t)(Mmax + 128) * 2048, nVt = (size_t)1024 * (Mmax + 128);
  CK(hipMalloc(&dA, hA.size() * 2)); CK(hipMalloc(&dW, hW.size() * 2)); CK(hipMalloc(&dC, nC * 4)); CK(hipMalloc(&dR, nC * 4));
  CK(hipMalloc(&dRes, nC * 4)); CK(hipMalloc(&dBias, 3072 * 4)); CK(hipMalloc(&dH, nH * 2)); CK(hipMalloc(&dVt, nVt * 2)); CK(hipMalloc(&dSeq, Mmax * 4));
  CK(hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dW, hW.data(), hW.size() * 2, hipMemcpyHostToDevice));
  std::vector<float> res(nC), bias(3072);
  for (auto &v : res) v = (rand() % 2001 - 1000) / 500.f;
  for (auto &v : bias) v = (rand() % 2001 - 1000) / 2000.f;
  CK(hipMemcpy(dRes, res.data(), nC * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dBias, bias.data(), 3072 * 4, hipMemcpyHostToDevice));
  char *dFlush; unsigned *dSink; CK(hipMalloc(&dFlush, (size_t)1 << 30)); CK(hipMalloc(&dSink, 4));
  hipStream_t s; CK(hipStreamCreate(&s));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  struct Var { const char *name; int deep; };
  const Var vars[4] = {{"large-kernels", -1}, {"deep auto", 0}, {"deep S=3", 3}, {"deep S=6", 6}};
  printf("%-34s %6s %-14s %9s %9s  %s\n", "shape", "M", "kernel", "us/launch", "TF/s", "check");
  for (int M : {1792}) {
    std::vector<int> seq(M);
    for (int i = 0; i < M; i++) seq[i] = (i % 896 == 895) ? -1 : i / 896; // a guard row per sequence
    CK(hipMemcpy(dSeq, seq.data(), M * 4, hipMemcpyHostToDevice));
    for (const Shape &sh : shapes) {
      std::vector<float> refC; std::vector<__half> refH, refVt;
      for (const Var &v : vars) {
        gemm_deep_mode() = v.deep;
        GemmArgs g{};
        for (int i = 0; i < 3; i++) { g.A[i] = dA + 64 * Kmax; g.row_off[i] = sh.nseg == 3 ? i - 1 : 0; }
        g.nseg = sh.nseg; g.kseg = sh.K; g.lda = Kmax; g.W = dW; g.M = M; g.N = sh.N; g.bias = dBias; g.row_seq = dSeq;
        g.mode = sh.mode; g.outF = dC; g.ldo = sh.N; g.resid = sh.resid ? dRes : nullptr;
        g.outH = dH; g.ldh = sh.mode == GEMM_OUT_QKV ? 2048 : 1024; g.outVt = dVt; g.ldvt = Mmax + 128;
        CK(hipMemsetAsync(dC, 0xff, nC * 4, s)); CK(hipMemsetAsync(dH, 0xff, nH * 2, s)); CK(hipMemsetAsync(dVt, 0xff, nVt * 2, s));
        CK(launch_gemm_f16(g, s));
        CK(hipStreamSynchronize(s));
        char chk[96] = "";
        std::vector<float> c(nC); std::vector<__half> h(nH), vt(nVt);
        CK(hipMemcpy(c.data(), dC, nC * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(h.data(), dH, nH * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(vt.data(), dVt, nVt * 2, hipMemcpyDeviceToHost));
        if (v.deep < 0) {
          refC = c; refH = h; refVt = vt;
          if (sh.mode == GEMM_OUT_F32) {
            naive_kernel<<<dim3((sh.N + 255) / 256, M), 256, 0, s>>>(dA + 64 * Kmax, Kmax, dW, sh.nseg * sh.K, sh.nseg, sh.K, M, sh.N, sh.resid ? dRes : nullptr, dR);
            std::vector<float> r(nC);
            CK(hipMemcpy(r.data(), dR, nC * 4, hipMemcpyDeviceToHost));
            double maxd = 0;
            for (int m = 0; m < M; m++)
              if (seq[m] >= 0) for (int n = 0; n < sh.N; n++) maxd = fmax(maxd, fabs(c[(size_t)m * sh.N + n] - (r[(size_t)m * sh.N + n] + bias[n])));
            snprintf(chk, sizeof chk, "vs naive maxdiff %.2g", maxd);
          }
        } else {
          const bool same = !memcmp(c.data(), refC.data(), nC * 4) && !memcmp(h.data(), refH.data(), nH * 2) && !memcmp(vt.data(), refVt.data(), nVt * 2);
          snprintf(chk, sizeof chk, "%s", same ? "bit-identical to large-kernels" : "DIFFERS from large-kernels");
        }
        // in-situ condition of the single-utterance diffusion step: the activations were just written (L2 / MALL warm), the weights were last
        // touched a whole step ago (0.36 GB of fp16 weights cycle through a 256 MB memory-side cache: cold). Flush everything with a 1 GB fill,
        // re-touch A, then time ONE launch between events; median of 15.
        double cold_us = 0;
        if (getenv("TTS_COLD")) {
          std::vector<float> ts;
          for (int it = 0; it < 15; it++) {
            CK(hipMemsetAsync(dFlush, it, (size_t)1 << 30, s));
            touch_kernel<<<512, 256, 0, s>>>((const uint4 *)(dA + 64 * Kmax), (size_t)M * Kmax / 8, dSink);
            if (sh.resid) touch_kernel<<<512, 256, 0, s>>>((const uint4 *)dRes, (size_t)M * 1024 / 4, dSink);
            CK(hipEventRecord(e0, s));
            CK(launch_gemm_f16(g, s));
            CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
            float ms1; CK(hipEventElapsedTime(&ms1, e0, e1));
            ts.push_back(ms1 * 1000.f);
          }
          std::sort(ts.begin(), ts.end());
          cold_us = ts[ts.size() / 2];
        }
        const int iters = 50;
        for (int i = 0; i < 5; i++) CK(launch_gemm_f16(g, s));
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < iters; i++) CK(launch_gemm_f16(g, s));
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double fl = 2.0 * M * sh.N * (double)sh.K * sh.nseg, us = 1000.0 * ms / iters;
        printf("%-34s %6d %-14s %9.1f %9.1f  cold-W %6.1f us  %s\n", sh.name, M, v.name, us, fl / (us * 1e-6) / 1e12, cold_us, chk);
      }
    }
  }
  return 0;
}

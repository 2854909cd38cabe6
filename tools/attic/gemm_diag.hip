// Developer tool: where does a diffusion GEMM launch spend its time?  Times the one-tile-per-workgroup kernels and the balanced persistent kernels on
// the benchmark's shapes and, in the trace build, prints per-workgroup phase statistics from in-kernel wall-clock stamps
// (gemm_f16.h, TTS_GEMM_TRACE): setup | first K tile (DMA round trip) | rest of the K loop | epilogue issue | store drain, plus the
// dispatch timeline (workgroup starts per 5 us).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -I tortoise.cpp_amd/csrc -I tools [-DTTS_GEMM_TRACE] [-DTTS_GEMM_DIAG_NOEPI] tools/gemm_diag.hip -o gemm_diag
//   hipcc ... -DTTS_GEMM_VARIANT=6 -I tools tools/gemm_diag.hip -o gemm_diag_v6      (the 8-phase 256^2 experiment kernel)
#define TTS_GEMM_DIAG 1
#include "gemm_f16_onetile.h" // the round-2 one-tile-per-workgroup kernels these tools were written against
#include <algorithm>
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

struct Shape { const char *name; int M, N, K, nseg, mode, resid; };

int main(int argc, char **argv) {
  const int Mmax = 28032;
  std::vector<Shape> shapes = {
      {"in_layers  k1 N1024 K1024        ", 28032, 1024, 1024, 1, GEMM_OUT_F32, 0},
      {"proj_out   k1 N1024 K1024 +resid ", 28032, 1024, 1024, 1, GEMM_OUT_F32, 1},
      {"qkv        k1 N3072 K1024        ", 28032, 3072, 1024, 1, GEMM_OUT_QKV, 0},
      {"out_layers k3 N1024 K3x1024 +res ", 28032, 1024, 1024, 3, GEMM_OUT_F32, 1},
      {"slope      k1 N1024 K2048        ", 28032, 1024, 2048, 1, GEMM_OUT_F32, 0},
      {"slope      k1 N1024 K4096        ", 28032, 1024, 4096, 1, GEMM_OUT_F32, 0},
      {"integrator k1 N1024 K1024 M14848 ", 14848, 1024, 1024, 1, GEMM_OUT_F32, 0},
      {"integrator k3 N1024 K3x1024 M14848", 14848, 1024, 1024, 3, GEMM_OUT_F32, 1},
  };
  const int Kmax = 4096;
  std::vector<__half> hA((size_t)(Mmax + 2) * Kmax), hW((size_t)3072 * Kmax);
  srand(1);
  for (auto &v : hA) v = __float2half((rand() % 2001 - 1000) / 1000.f);
  for (auto &v : hW) v = __float2half((rand() % 2001 - 1000) / 4000.f);
  __half *dA, *dW, *dH, *dVt; float *dC, *dR, *dRes, *dBias; int *dSeq;
  CK(hipMalloc(&dA, hA.size() * 2)); CK(hipMalloc(&dW, hW.size() * 2)); CK(hipMalloc(&dC, (size_t)Mmax * 1024 * 4));
  CK(hipMalloc(&dR, (size_t)256 * 1024 * 4)); CK(hipMalloc(&dRes, (size_t)Mmax * 1024 * 4)); CK(hipMalloc(&dBias, 3072 * 4));
  CK(hipMalloc(&dH, (size_t)(Mmax + 128) * 2048 * 2)); CK(hipMalloc(&dVt, (size_t)1024 * (Mmax + 128) * 2)); CK(hipMalloc(&dSeq, Mmax * 4));
  CK(hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dW, hW.data(), hW.size() * 2, hipMemcpyHostToDevice));
  {
    std::vector<float> r((size_t)Mmax * 1024);
    for (auto &v : r) v = (rand() % 2001 - 1000) / 500.f;
    CK(hipMemcpy(dRes, r.data(), r.size() * 4, hipMemcpyHostToDevice));
  }
  CK(hipMemset(dBias, 0, 3072 * 4)); CK(hipMemset(dSeq, 0, Mmax * 4));
  hipStream_t s; CK(hipStreamCreate(&s));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  // variants: classic, then the round stagger sweep (delay us, div)
  struct Var { const char *name; int bal; float us; int div; int drip = 1; };
  const std::vector<Var> vars = {{"classic", 0, 0.f, 1}, {"balanced", 1, 0.f, 1},      {"stag20/32", 0, 20.f, 32}};
#ifdef TTS_GEMM_VARIANT
  const int npers = 1;
#else
  const int npers = (int)vars.size();
#endif
  printf("%-36s %-10s %9s %9s %s\n", "shape", "kernel", "us/launch", "TF/s", "check");
  for (const Shape &sh : shapes) {
    for (int pers = 0; pers < npers; pers++) {
      gemm_balanced_flag() = vars[pers].bal;
      gemm_stagger_us() = vars[pers].us;
      gemm_stagger_div() = vars[pers].div;
      GemmArgs g{};
      const int lda = sh.K;
      for (int i = 0; i < 3; i++) { g.A[i] = dA + lda; g.row_off[i] = sh.nseg == 3 ? i - 1 : 0; }
      g.nseg = sh.nseg; g.kseg = sh.K; g.lda = lda; g.W = dW; g.M = sh.M; g.N = sh.N; g.bias = dBias; g.row_seq = dSeq;
      g.mode = sh.mode; g.outF = dC; g.ldo = sh.N; g.resid = sh.resid ? dRes : nullptr;
      g.outH = dH; g.ldh = 2048; g.outVt = dVt; g.ldvt = Mmax + 128;
      CK(launch_gemm_f16(g, s));
      CK(hipStreamSynchronize(s));
      char chk[64] = "-";
#ifndef TTS_GEMM_DIAG_NOEPI
      if (sh.mode == GEMM_OUT_F32) { // three bands of 256 rows (start, an XCD boundary of the row partition, end) against a naive kernel
        const int MC = 256, ldw = sh.nseg * sh.K;
        double maxd = 0, maxr = 0;
        const int bands[3] = {0, (sh.M / 8 / 32) * 32 - 128, sh.M - MC};
        for (int b0 : bands) {
          naive_kernel<<<dim3((sh.N + 255) / 256, MC), 256, 0, s>>>(dA + lda + (size_t)b0 * lda, lda, dW, ldw, sh.nseg, sh.K, MC, sh.N,
                                                                   sh.resid ? dRes + (size_t)b0 * sh.N : nullptr, dR);
          std::vector<float> c((size_t)MC * sh.N), r((size_t)MC * sh.N);
          CK(hipMemcpy(c.data(), dC + (size_t)b0 * sh.N, c.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(r.data(), dR, r.size() * 4, hipMemcpyDeviceToHost));
          for (size_t i = 0; i < c.size(); i++) { maxd = fmax(maxd, fabs(c[i] - r[i])); maxr = fmax(maxr, fabs(r[i])); }
        }
        snprintf(chk, sizeof chk, "maxdiff %.2g / %.2g", maxd, maxr);
      }
#endif
      const int iters = 20;
      for (int i = 0; i < 3; i++) CK(launch_gemm_f16(g, s));
      CK(hipEventRecord(e0, s));
      for (int i = 0; i < iters; i++) CK(launch_gemm_f16(g, s));
      CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      const double fl = 2.0 * sh.M * sh.N * (double)sh.K * sh.nseg, us = 1000.0 * ms / iters;
      printf("%-36s %-10s %9.1f %9.1f %s\n", sh.name, vars[pers].name, us, fl / (us * 1e-6) / 1e12, chk);
#ifdef TTS_GEMM_TRACE
      { // one more launch with a clean trace buffer
        static std::vector<unsigned long long> tr(65536 * 8);
        std::fill(tr.begin(), tr.end(), 0ull);
        CK(hipMemcpyToSymbol(HIP_SYMBOL(tts_gemm_trace), tr.data(), tr.size() * 8));
        CK(launch_gemm_f16(g, s));
        CK(hipStreamSynchronize(s));
        CK(hipMemcpyFromSymbol(tr.data(), HIP_SYMBOL(tts_gemm_trace), tr.size() * 8));
        std::vector<double> d[5];
        unsigned long long tmin = ~0ull, tmax = 0;
        std::vector<unsigned long long> starts;
        for (size_t w = 0; w < 65536; w++) {
          const unsigned long long *p = &tr[w * 8];
          if (!p[0] || !p[4]) continue;
          const unsigned long long tend = p[5] ? p[5] : p[4];
          tmin = std::min(tmin, p[0]); tmax = std::max(tmax, tend);
          starts.push_back(p[0]);
          d[0].push_back((p[1] - p[0]) * 0.01); d[1].push_back((p[2] - p[1]) * 0.01); d[2].push_back((p[3] - p[2]) * 0.01);
          d[3].push_back((p[4] - p[3]) * 0.01); d[4].push_back(p[5] ? (p[5] - p[4]) * 0.01 : 0.0);
        }
        const char *nm[5] = {"setup", "first K tile", "K loop rest", "epilogue issue", "store drain"};
        printf("    traced tiles %zu, span %.1f us\n", starts.size(), (tmax - tmin) * 0.01);
        for (int q = 0; q < 5; q++) {
          if (d[q].empty()) continue;
          std::sort(d[q].begin(), d[q].end());
          double sum = 0; for (double v : d[q]) sum += v;
          printf("    %-15s mean %7.2f  p10 %7.2f  p50 %7.2f  p90 %7.2f  max %7.2f us\n", nm[q], sum / d[q].size(), d[q][d[q].size() / 10],
                 d[q][d[q].size() / 2], d[q][d[q].size() * 9 / 10], d[q].back());
        }
        std::vector<int> hist((size_t)((tmax - tmin) / 500) + 1, 0);
        for (auto t : starts) hist[(size_t)((t - tmin) / 500)]++;
        if (pers == 0 && &sh == &shapes[0]) { // where do consecutive workgroups of one XCD land? (HW_ID: cu_id [11:8], se_id [15:13]; XCC_ID [3:0])
          printf("    placement of blockIdx 0, 8, 16, ... (XCD 0 by construction): (xcc se cu simd):");
          for (int i = 0; i < 48; i++) {
            const unsigned hw = (unsigned)tr[(size_t)i * 8 * 8 + 6], xc = (unsigned)tr[(size_t)i * 8 * 8 + 7];
            printf(" %u.%u.%u.%u", xc & 15, (hw >> 13) & 7, (hw >> 8) & 15, (hw >> 4) & 3);
          }
          printf("\n");
        }
        printf("    tile starts per 5 us:");
        for (int h : hist) printf(" %d", h);
        printf("\n");
      }
#endif
    }
  }
  return 0;
}

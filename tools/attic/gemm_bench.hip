// Developer tool: times gemm_f16_kernel variants on the diffusion shapes and checks them against a naive
// kernel.  hipcc --offload-arch=gfx950 -O3 -std=c++17 -I tortoise.cpp_amd/csrc -I tools tools/gemm_bench.hip -o /tmp/gemm_bench
#include "gemm_f16_onetile.h" // the round-2 one-tile-per-workgroup kernels these tools were written against
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace tts;

__global__ void naive_kernel(const __half *A, int lda, const __half *W, int ldw, int nseg, int kseg, int M, int N, float *C) {
  int n = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y;
  if (n >= N) return;
  float acc = 0;
  for (int s = 0; s < nseg; s++)
    for (int k = 0; k < kseg; k++)
      acc += __half2float(A[(size_t)(m + (nseg == 3 ? s - 1 : 0)) * lda + k]) * __half2float(W[(size_t)n * ldw + s * kseg + k]);
  C[(size_t)m * N + n] = acc;
}

int main(int argc, char **argv) {
  int M = argc > 1 ? atoi(argv[1]) : 28288, N = argc > 2 ? atoi(argv[2]) : 1024, K = argc > 3 ? atoi(argv[3]) : 1024, nseg = argc > 4 ? atoi(argv[4]) : 1;
  int ldw = nseg * K;
  std::vector<__half> hA((size_t)(M + 2) * K), hW((size_t)N * ldw);
  srand(1);
  for (auto &v : hA) v = __float2half((rand() % 2001 - 1000) / 1000.f);
  for (auto &v : hW) v = __float2half((rand() % 2001 - 1000) / 4000.f);
  __half *dA, *dW; float *dC, *dR, *dRes;
  hipMalloc(&dA, hA.size() * 2); hipMalloc(&dW, hW.size() * 2); hipMalloc(&dC, (size_t)M * N * 4); hipMalloc(&dR, (size_t)M * N * 4);
  hipMalloc(&dRes, (size_t)M * N * 4);
  hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dW, hW.data(), hW.size() * 2, hipMemcpyHostToDevice);
  hipMemset(dRes, 0, (size_t)M * N * 4);
  GemmArgs g{};
  for (int i = 0; i < 3; i++) { g.A[i] = dA + K; g.row_off[i] = nseg == 3 ? i - 1 : 0; }
  g.nseg = nseg; g.kseg = K; g.lda = K; g.W = dW; g.M = M; g.N = N; g.bias = nullptr; g.row_seq = nullptr;
  g.mode = GEMM_OUT_F32; g.outF = dC; g.ldo = N; g.resid = argc > 5 ? dRes : nullptr;
  hipStream_t s; hipStreamCreate(&s);
  if (launch_gemm_f16(g, s) != hipSuccess) { printf("launch failed\n"); return 1; }
  hipStreamSynchronize(s);
  // check a band of rows
  int MC = 256;
  naive_kernel<<<dim3((N + 255) / 256, MC), 256, 0, s>>>(dA + K, K, dW, ldw, nseg, K, MC, N, dR);
  std::vector<float> c((size_t)MC * N), r((size_t)MC * N);
  hipMemcpy(c.data(), dC, c.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(r.data(), dR, r.size() * 4, hipMemcpyDeviceToHost);
  double maxd = 0, maxr = 0;
  for (size_t i = 0; i < c.size(); i++) { maxd = fmax(maxd, fabs(c[i] - r[i])); maxr = fmax(maxr, fabs(r[i])); }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  int iters = 20;
  for (int i = 0; i < 3; i++) launch_gemm_f16(g, s);
  hipEventRecord(e0, s);
  for (int i = 0; i < iters; i++) launch_gemm_f16(g, s);
  hipEventRecord(e1, s); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double fl = 2.0 * M * N * (double)K * nseg;
  printf("M=%d N=%d K=%d nseg=%d resid=%d: %.1f us  %.1f TF/s  maxdiff %.3g (ref max %.3g)\n", M, N, K, nseg, argc > 5, 1000 * ms / iters,
         fl / (ms / iters * 1e-3) / 1e12, maxd, maxr);
  return 0;
}

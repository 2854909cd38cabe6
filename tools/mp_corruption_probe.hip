// Developer probe, independent of the engine: does a trivial matrix-vector kernel give the same answer every time while OTHER PROCESSES use the GPU?
//   tools/bin/mp_corruption_probe <seconds> [tag] [matrices]     (run two or three copies at once, or beside an engine process)
// matrices > 1: the kernel walks through that many different 8 MB matrices, so every launch reads COLD weights (as the engine's time MLP does once per call)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
// out[n] = sum_k x[k] * W[n][k]: one wave per output, 16-byte loads, shuffle reduction (the shape of the engine's time-MLP kernel)
__global__ __launch_bounds__(256) void matvec(const float *__restrict__ x, const float *__restrict__ W, int K, int N, float *__restrict__ out) {
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (n >= N) return;
  float acc = 0.f;
  for (int k = lane * 4; k < K; k += 256) {
    const float4 w = *(const float4 *)(W + (size_t)n * K + k), xv = *(const float4 *)(x + k);
    acc = fmaf(xv.x, w.x, acc); acc = fmaf(xv.y, w.y, acc); acc = fmaf(xv.z, w.z, acc); acc = fmaf(xv.w, w.w, acc);
  }
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if (lane == 0) out[n] = acc;
}
__global__ void scale(float *x, int n, float s) { int i = blockIdx.x * 256 + threadIdx.x; if (i < n) x[i] *= s; }
int main(int argc, char **argv) {
  const double secs = argc > 1 ? atof(argv[1]) : 10.0;
  const char *tag = argc > 2 ? argv[2] : "p";
  const int K = 1024, N = 2048, NW = argc > 3 ? atoi(argv[3]) : 1;
  std::vector<float> hw((size_t)N * K), hx(K);
  unsigned s = 12345u + (unsigned)tag[0];
  for (auto &v : hw) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 65536.f - 0.5f; }
  for (auto &v : hx) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 65536.f - 0.5f; }
  float *W, *x, *out, *big; CK(hipMalloc(&W, (size_t)NW * hw.size() * 4)); CK(hipMalloc(&x, K * 4)); CK(hipMalloc(&out, N * 4)); CK(hipMalloc(&big, 256u << 20));
  for (int m = 0; m < NW; m++) { hw[m] += 1.0f; CK(hipMemcpy(W + (size_t)m * hw.size(), hw.data(), hw.size() * 4, hipMemcpyHostToDevice)); }
  hipStream_t st; CK(hipStreamCreate(&st));
  std::vector<std::vector<float>> refs(NW, std::vector<float>(N));
  std::vector<float> got(N);
  long iters = 0, bad_iters = 0, bad_vals = 0;
  const auto t0 = std::chrono::steady_clock::now();
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs) {
    CK(hipMemcpyAsync(x, hx.data(), K * 4, hipMemcpyHostToDevice, st));
    CK(hipStreamSynchronize(st));
    const int m = (int)(iters % NW);
    std::vector<float> &ref = refs[m];
    matvec<<<N / 4, 256, 0, st>>>(x, W + (size_t)m * N * K, K, N, out);
    CK(hipMemcpyAsync(got.data(), out, N * 4, hipMemcpyDeviceToHost, st));
    CK(hipStreamSynchronize(st));
    if (iters < NW) ref = got;
    else {
      int nb = 0, first = -1;
      for (int i = 0; i < N; i++) if (memcmp(&got[i], &ref[i], 4)) { if (first < 0) first = i; nb++; }
      if (nb) { bad_iters++; bad_vals += nb; if (bad_iters <= 5) printf("[%s] iteration %ld: %d of %d outputs differ from the first iteration, first at %d (%.9g vs %.9g)\n", tag, iters, nb, N, first, got[first], ref[first]); }
    }
    // some bandwidth traffic of our own between the probes (the engine's forwards do the same)
    scale<<<(64 << 20) / 256, 256, 0, st>>>(big, 64 << 20, 1.0001f);
    iters++;
  }
  CK(hipStreamSynchronize(st));
  printf("[%s] %ld iterations, %ld with wrong outputs (%ld values)\n", tag, iters, bad_iters, bad_vals);
  return 0;
}

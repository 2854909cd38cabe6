// Developer microbench for the diffusion attention kernel: bench-shaped batch (32 sequences x 870 rows, 16 heads),
// steady-state time per launch and per-tile phase timestamps (shader cycles) of wave 0 of one workgroup.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -DTTS_ATT_TRACE=100 -I include -I tortoise.cpp_amd/csrc -I tools/attic \
//         tools/attn_bench.hip tortoise.cpp_amd/csrc/host_logic.cpp -o tools/attn_bench_bin   (TTS_ATT_TRACE = traced workgroup id)
#include "../tortoise.cpp_amd/csrc/diffusion.hip"
#include "attn_r3_kernel.h"
#include <algorithm>
#include <cstdio>
#include <cmath>
#include <cstring>
#include <vector>
using namespace tts;
#ifndef ATT_NR
#define ATT_NR 2
#endif
hipEvent_t tts::prof_event(tts_ctx *) { return nullptr; }
extern "C" int32_t tts_diffusion_frames(int32_t rows) { return rows * 4 * 24000 / 22050; } // api.cpp is not linked here
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
__global__ void clock_probe(long long *out) { // shader-cycle counter vs the constant 100 MHz wall clock
  long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
  float x = threadIdx.x;
  for (int i = 0; i < 200000; i++) x = x * 1.0001f + 0.5f;
  long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
  if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; out[2] = (long long)x; }
}
int main() {
  const int ns = 32, T = 870, per = 884, rows = ns * per, ldvt = rows + 128;
  std::vector<int> st(ns), ln(ns);
  for (int s = 0; s < ns; s++) { st[s] = 8 + s * per; ln[s] = T; }
  __half *qk, *vt, *out; int *dst, *dln; float *tab;
  CK(hipMalloc(&qk, (size_t)(rows + 256) * 2048 * 2)); CK(hipMalloc(&vt, (size_t)1024 * ldvt * 2)); CK(hipMalloc(&out, (size_t)(rows + 256) * 1024 * 2));
  CK(hipMalloc(&dst, ns * 4)); CK(hipMalloc(&dln, ns * 4)); CK(hipMalloc(&tab, 16 * 128 * 4));
  std::vector<__half> h((size_t)(rows + 256) * 2048);
  for (size_t i = 0; i < h.size(); i++) h[i] = __float2half((float)((i * 2654435761u >> 9) & 1023) / 1024.f - 0.5f);
  CK(hipMemcpy(qk, h.data(), h.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(vt, h.data(), (size_t)1024 * ldvt * 2 < h.size() * 2 ? (size_t)1024 * ldvt * 2 : h.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(dst, st.data(), ns * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dln, ln.data(), ns * 4, hipMemcpyHostToDevice));
  CK(hipMemset(tab, 0, 16 * 128 * 4));
  const int nq = (T + 127) / 128;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  // random-ish bias table (the product's is 8 x the T5 bucket table): the near-diagonal path must be exercised with non-zero values
  { std::vector<float> tb(16 * 128); for (size_t i = 0; i < tb.size(); i++) tb[i] = (float)((i * 40503u >> 3) & 255) / 64.f - 2.f; CK(hipMemcpy(tab, tb.data(), tb.size() * 4, hipMemcpyHostToDevice)); }
  __half *out2; CK(hipMalloc(&out2, (size_t)(rows + 256) * 1024 * 2));
  CK(hipMemset(out, 0, (size_t)(rows + 256) * 1024 * 2)); CK(hipMemset(out2, 0, (size_t)(rows + 256) * 1024 * 2));
  diff_attn_kernel<ATT_NR><<<nq * 16 * ns, 256, att_lds<ATT_NR>(), 0>>>(qk, vt, ldvt, dst, dln, tab, out, nq);
  diff_attn_r3_kernel<ATT_NR><<<nq * 16 * ns, 256, att_lds<ATT_NR>(), 0>>>(qk, vt, ldvt, dst, dln, tab, out2, nq);
  CK(hipDeviceSynchronize());
  {
    std::vector<__half> a((size_t)(rows + 256) * 1024), b(a.size());
    CK(hipMemcpy(a.data(), out, a.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), out2, b.size() * 2, hipMemcpyDeviceToHost));
    size_t bad = 0, nz = 0;
    float worst = 0.f, range = 0.f;
    for (size_t i = 0; i < a.size(); i++) {
      bad += memcmp(&a[i], &b[i], 2) != 0; nz += __half2float(a[i]) != 0.f;
      worst = std::max(worst, fabsf(__half2float(a[i]) - __half2float(b[i]))); range = std::max(range, fabsf(__half2float(b[i])));
    }
    printf("product kernel vs round-3 kernel: %zu of %zu outputs differ (%zu non-zero), max abs difference %.2e of range %.2f\n", bad, a.size(), nz, worst, range);
  }
  double us[2][5];
  for (int r = 0; r < 5; r++)
    for (int v = 0; v < 2; v++) {
      auto go = [&]() {
        if (v == 0) diff_attn_kernel<ATT_NR><<<nq * 16 * ns, 256, att_lds<ATT_NR>(), 0>>>(qk, vt, ldvt, dst, dln, tab, out, nq);
        else diff_attn_r3_kernel<ATT_NR><<<nq * 16 * ns, 256, att_lds<ATT_NR>(), 0>>>(qk, vt, ldvt, dst, dln, tab, out2, nq);
      };
      for (int i = 0; i < 3; i++) go();
      CK(hipEventRecord(e0, 0));
      for (int i = 0; i < 20; i++) go();
      CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
      float ms1; CK(hipEventElapsedTime(&ms1, e0, e1));
      us[v][r] = 1e3 * ms1 / 20;
    }
  double fl = 4.0 * T * (double)T * 64 * 16 * ns;
  for (int v = 0; v < 2; v++) {
    std::sort(us[v], us[v] + 5);
    printf("%-28s med %.1f us/launch (min %.1f), %.0f TF/s\n", v == 0 ? "attention (product, round 4):" : "attention (round-3 kernel):", us[v][2], us[v][0], fl / (us[v][2] * 1e-6) / 1e12);
  }
  float ms = (float)(us[0][2] * 20 / 1e3);
  (void)ms;
  {
    long long *d, hres[3];
    CK(hipMalloc(&d, 24));
    clock_probe<<<1, 64>>>(d);
    CK(hipMemcpy(hres, d, 24, hipMemcpyDeviceToHost));
    printf("cycle counter: %lld ticks in %.1f us -> %.0f MHz\n", hres[0], hres[1] * 0.01, hres[0] / (hres[1] * 0.01));
  }
#ifdef TTS_ATT_TRACE
  std::vector<long long> t(64 * 8);
  CK(hipMemcpyFromSymbol(t.data(), HIP_SYMBOL(tts_att_trace), t.size() * 8));
  printf("traced workgroup: %lld cycles in %.2f us -> %.0f MHz\n", t[63 * 8 + 2] - t[63 * 8], (t[63 * 8 + 3] - t[63 * 8 + 1]) * 0.01,
         (t[63 * 8 + 2] - t[63 * 8]) / ((t[63 * 8 + 3] - t[63 * 8 + 1]) * 0.01));
  printf("tile: wait barrier stage [scores-issue] softmax PV | total (shader cycles)\n");
  for (int kb = 0; kb < 14; kb++) {
    long long *p = &t[kb * 8];
    printf("  %2d: %6lld %6lld %6lld %6lld %6lld %6lld | %6lld\n", kb, p[1] - p[0], p[2] - p[1], p[3] - p[2], p[4] - p[3], p[5] - p[4], p[6] - p[5],
           kb + 1 < 14 ? t[(kb + 1) * 8] - p[0] : p[6] - p[0]);
  }
#endif
  return 0;
}

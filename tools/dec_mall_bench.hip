// Developer microbench (round 3): does the 256 MB Infinity Cache (MALL) serve the decode step's weight slabs faster than HBM does?
// The four GEMV kernels of a decode layer are timed cycling through L distinct weight sets: L = 30 (504 MB of fc slabs: the real step, every
// slab comes from HBM), L = 8 / 4 (134 / 67 MB: beyond the 32 MB of L2, inside the MALL), L = 1 (one 17 MB slab re-read every launch).
// If the MALL-resident cycles are much faster, a background prefetch of the next layers' slabs would pay; if not, it cannot.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I tortoise.cpp_amd/csrc tools/dec_mall_bench.hip tortoise.cpp_amd/csrc/host_logic.cpp -o tools/bin/dec_mall_bench
#include "../tortoise.cpp_amd/csrc/ar.hip"
#include <cstdio>
#include <vector>
using namespace tts;
hipEvent_t tts::prof_event(tts_ctx *) { return nullptr; } // profiling is off in this harness
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main() {
  const int LMAX = 30, B = 16;
  float *h, *ff, *att, *bvec, *q;
  CK(hipMalloc(&h, B * 1024 * 4)); CK(hipMalloc(&ff, B * 4096 * 4)); CK(hipMalloc(&att, B * 1024 * 4)); CK(hipMalloc(&q, B * 1024 * 4));
  CK(hipMalloc(&bvec, 8256 * 4));
  std::vector<float> hv(B * 4096);
  for (size_t i = 0; i < hv.size(); i++) hv[i] = (float)((i * 2654435761u >> 8) & 1023) / 512.f - 1.f;
  CK(hipMemcpy(h, hv.data(), B * 1024 * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(ff, hv.data(), B * 4096 * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(att, hv.data(), B * 1024 * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(bvec, hv.data(), 8256 * 4, hipMemcpyHostToDevice));
  std::vector<float *> wfc(LMAX), wfc2(LMAX), wproj(LMAX), wqkv(LMAX);
  std::vector<float> wv(1024 * 4096);
  for (size_t i = 0; i < wv.size(); i++) wv[i] = (float)((i * 40503u >> 4) & 255) / 4096.f;
  for (int l = 0; l < LMAX; l++) {
    CK(hipMalloc(&wfc[l], 1024 * 4096 * 4)); CK(hipMalloc(&wfc2[l], 1024 * 4096 * 4)); CK(hipMalloc(&wproj[l], 1024 * 1024 * 4)); CK(hipMalloc(&wqkv[l], 1024 * 3072 * 4));
    CK(hipMemcpy(wfc[l], wv.data(), 1024 * 4096 * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(wfc2[l], wv.data(), 1024 * 4096 * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(wproj[l], wv.data(), 1024 * 1024 * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(wqkv[l], wv.data(), 1024 * 3072 * 4, hipMemcpyHostToDevice));
  }
  __half *kc, *vc; CK(hipMalloc(&kc, (size_t)B * 256 * 1024 * 2)); CK(hipMalloc(&vc, (size_t)B * 256 * 1024 * 2));
  StepState *ss; CK(hipMalloc(&ss, sizeof(StepState)));
  StepState hs{20, 3}; CK(hipMemcpy(ss, &hs, sizeof(hs), hipMemcpyHostToDevice));
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("%-34s", "kernel \\ distinct weight sets L");
  const int Ls[6] = {30, 12, 8, 4, 2, 1};
  for (int L : Ls) printf(" %8d", L);
  printf("   (us per launch)\n");
  auto row = [&](const char *tag, auto launch) {
    printf("%-34s", tag);
    for (int L : Ls) {
      for (int i = 0; i < 60; i++) launch(i % L);
      (void)hipEventRecord(e0, st);
      const int reps = 300;
      for (int i = 0; i < reps; i++) launch(i % L);
      (void)hipEventRecord(e1, st);
      (void)hipStreamSynchronize(st);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      printf(" %8.2f", 1e3 * ms / reps);
    }
    printf("\n");
  };
  row("dec_ln_gemv<GELU> fc 16.8 MB nt", [&](int l) {
    DecLnArgs a{h, nullptr, nullptr, wfc[l], (const __half *)wfc[l], bvec, B, 4096, 0, 0, ff, nullptr, nullptr, ss, 0, 0};
    dec_ln_gemv_kernel<DEC_GELU, 1, true><<<dim3(256, 1), 256, 0, st>>>(a); });
  row("dec_ln_gemv<GELU> fc 16.8 MB plain", [&](int l) {
    DecLnArgs a{h, nullptr, nullptr, wfc[l], (const __half *)wfc[l], bvec, B, 4096, 0, 0, ff, nullptr, nullptr, ss, 0, 0};
    dec_ln_gemv_kernel<DEC_GELU, 1, false><<<dim3(256, 1), 256, 0, st>>>(a); });
  row("dec_ln_gemv<QKV> 12.6 MB nt", [&](int l) {
    DecLnArgs a{h, nullptr, nullptr, wqkv[l], (const __half *)wqkv[l], bvec, B, 3072, 1024, 0, q, kc, vc, ss, 256, 0};
    dec_ln_gemv_kernel<DEC_QKV, 1, true><<<dim3(192, 1), 256, 0, st>>>(a); });
  row("dec_gemv_resid<4,512> fc2 16.8 MB nt", [&](int l) { dec_gemv_resid_kernel<4, 512, false, true><<<dim3(256, 1), 512, 0, st>>>(ff, B, wfc2[l], bvec, h); });
  row("dec_gemv_resid<4,512> fc2 plain", [&](int l) { dec_gemv_resid_kernel<4, 512, false, false><<<dim3(256, 1), 512, 0, st>>>(ff, B, wfc2[l], bvec, h); });
  row("dec_gemv_resid<1> proj 4.2 MB nt", [&](int l) { dec_gemv_resid_kernel<1, 256, false, true><<<dim3(256, 1), 256, 0, st>>>(att, B, wproj[l], bvec, h); });
  // the whole layer's four GEMVs back to back (the dependent chain of the real step without attention)
  row("layer: qkv+proj+fc+fc2 nt", [&](int l) {
    DecLnArgs a{h, nullptr, nullptr, wqkv[l], (const __half *)wqkv[l], bvec, B, 3072, 1024, 0, q, kc, vc, ss, 256, 0};
    dec_ln_gemv_kernel<DEC_QKV, 1, true><<<dim3(192, 1), 256, 0, st>>>(a);
    dec_gemv_resid_kernel<1, 256, false, true><<<dim3(256, 1), 256, 0, st>>>(att, B, wproj[l], bvec, h);
    DecLnArgs b{h, nullptr, nullptr, wfc[l], (const __half *)wfc[l], bvec, B, 4096, 0, 0, ff, nullptr, nullptr, ss, 0, 0};
    dec_ln_gemv_kernel<DEC_GELU, 1, true><<<dim3(256, 1), 256, 0, st>>>(b);
    dec_gemv_resid_kernel<4, 512, false, true><<<dim3(256, 1), 512, 0, st>>>(ff, B, wfc2[l], bvec, h); });
  return 0;
}

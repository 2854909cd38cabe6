// Developer microbench for the decode-step kernels: cycles through 30 distinct weight slabs (as the 30 layers
// do), times the steady-state launch and prints per-phase timestamps (TTS_DEC_TRACE) of a few workgroups.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DTTS_DEC_TRACE -I include -I tortoise.cpp_amd/csrc tools/dec_bench.hip \
//         tortoise.cpp_amd/csrc/host_logic.cpp -o tools/dec_bench_bin
#include "../tortoise.cpp_amd/csrc/ar.hip"
#include <cstdio>
#include <vector>
using namespace tts;
hipEvent_t tts::prof_event(tts_ctx *) { return nullptr; } // profiling is off in this harness

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

static void dump_trace(const char *tag, int nblocks, int nph) {
  std::vector<long long> t(8 * 4096);
  hipError_t e = hipMemcpyFromSymbol(t.data(), HIP_SYMBOL(tts::tts_dec_trace), t.size() * 8);
  if (e != hipSuccess) printf("memcpyFromSymbol: %s\n", hipGetErrorString(e));
  long long t0 = t[0];
  for (int b = 0; b < nblocks; b++) t0 = std::min(t0, t[b * 8]);
  printf("%s phases (us since first workgroup start; 100 MHz clock):\n", tag);
  int show[6] = {0, 1, nblocks / 2, nblocks - 2, nblocks - 1, 7};
  for (int s = 0; s < 6; s++) {
    int b = show[s];
    if (b < 0 || b >= nblocks) continue;
    printf("  wg %4d:", b);
    for (int i = 0; i < nph; i++) printf(" %6.2f", (t[b * 8 + i] - t0) * 0.01);
    printf("\n");
  }
  long long last = 0;
  for (int b = 0; b < nblocks; b++) last = std::max(last, t[b * 8 + nph - 1]);
  printf("  last workgroup's final stamp: %.2f us\n", (last - t0) * 0.01);
}

int main(int argc, char **argv) {
  const int L = argc > 1 ? atoi(argv[1]) : 30, B = 16; // L = 1: same slab every launch (is it still in L2 / MALL?)
  float *h, *ff, *att, *g, *bvec, *q;
  CK(hipMalloc(&h, B * 1024 * 4)); CK(hipMalloc(&ff, B * 4096 * 4)); CK(hipMalloc(&att, B * 1024 * 4)); CK(hipMalloc(&q, B * 1024 * 4));
  CK(hipMalloc(&g, 8256 * 4)); CK(hipMalloc(&bvec, 8256 * 4));
  std::vector<float> hv(B * 4096);
  for (size_t i = 0; i < hv.size(); i++) hv[i] = (float)((i * 2654435761u >> 8) & 1023) / 512.f - 1.f;
  CK(hipMemcpy(h, hv.data(), B * 1024 * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(ff, hv.data(), B * 4096 * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(att, hv.data(), B * 1024 * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(g, hv.data(), 8256 * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(bvec, hv.data(), 8256 * 4, hipMemcpyHostToDevice));
  std::vector<float *> wfc(L), wfc2(L), wproj(L), wqkv(L);
  for (int l = 0; l < L; l++) {
    CK(hipMalloc(&wfc[l], 1024 * 4096 * 4)); CK(hipMalloc(&wfc2[l], 1024 * 4096 * 4));
    CK(hipMalloc(&wproj[l], 1024 * 1024 * 4)); CK(hipMalloc(&wqkv[l], 1024 * 3072 * 4));
    CK(hipMemset(wfc[l], 0, 1024 * 4096 * 4)); CK(hipMemset(wfc2[l], 0, 1024 * 4096 * 4));
    CK(hipMemset(wproj[l], 0, 1024 * 1024 * 4)); CK(hipMemset(wqkv[l], 0, 1024 * 3072 * 4));
  }
  __half *kc, *vc; CK(hipMalloc(&kc, (size_t)B * 256 * 1024 * 2)); CK(hipMalloc(&vc, (size_t)B * 256 * 1024 * 2));
  StepState *ss; CK(hipMalloc(&ss, sizeof(StepState)));
  StepState hs{20, 3}; CK(hipMemcpy(ss, &hs, sizeof(hs), hipMemcpyHostToDevice));
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timeit = [&](const char *tag, auto launch, double bytes) {
    for (int i = 0; i < 30; i++) launch(i % L);
    (void)hipEventRecord(e0, st);
    const int reps = 300;
    for (int i = 0; i < reps; i++) launch(i % L);
    (void)hipEventRecord(e1, st);
    (void)hipStreamSynchronize(st);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s %7.2f us/launch  %6.2f TB/s\n", tag, 1e3 * ms / reps, bytes / (ms / reps * 1e-3) / 1e12);
  };
  timeit("dec_ln_gemv<GELU> fc", [&](int l) {
    DecLnArgs a{h, nullptr, nullptr, wfc[l], (const __half *)wfc[l], bvec, B, 4096, 0, 0, ff, nullptr, nullptr, ss, 0, 0};
    dec_ln_gemv_kernel<DEC_GELU, 1><<<dim3(256, 1), 256, 0, st>>>(a); }, 16.8e6);
  dump_trace("fc", 256, 5);
  timeit("dec_ln_gemv<QKV>", [&](int l) {
    DecLnArgs a{h, nullptr, nullptr, wqkv[l], (const __half *)wqkv[l], bvec, B, 3072, 1024, 0, q, kc, vc, ss, 256, 0};
    dec_ln_gemv_kernel<DEC_QKV, 1><<<dim3(192, 1), 256, 0, st>>>(a); }, 12.6e6);
  dump_trace("qkv", 192, 5);
  timeit("dec_gemv_resid<4> fc2", [&](int l) { dec_gemv_resid_kernel<4><<<dim3(256, 1), 256, 0, st>>>(ff, B, wfc2[l], bvec, h); }, 16.8e6);
  timeit("dec_gemv_resid<4,512> fc2", [&](int l) { dec_gemv_resid_kernel<4, 512><<<dim3(256, 1), 512, 0, st>>>(ff, B, wfc2[l], bvec, h); }, 16.8e6);
  dump_trace("fc2", 256, 4);
  timeit("dec_gemv_resid<1> proj", [&](int l) { dec_gemv_resid_kernel<1><<<dim3(256, 1), 256, 0, st>>>(att, B, wproj[l], bvec, h); }, 4.2e6);
  dump_trace("proj", 256, 4);
  for (int np : {20, 100, 164, 250}) {
    StepState hs2{np, 3}; CK(hipMemcpy(ss, &hs2, sizeof(hs2), hipMemcpyHostToDevice));
    char tag[64];
    snprintf(tag, sizeof tag, "attn_decode (exact) nk=%d", np + 1);
    timeit(tag, [&](int l) { attn_decode_kernel<<<dim3(B, 16), 256, 0, st>>>(q, kc, vc, ss, 256, att, 0); }, 0);
    snprintf(tag, sizeof tag, "attn_decode_fast nk=%d", np + 1);
    timeit(tag, [&](int l) { attn_decode_fast_kernel<<<dim3(B, 16), 256, 0, st>>>(q, kc, vc, ss, 256, att); }, 0);
  }
  CK(hipMemcpy(ss, &hs, sizeof(hs), hipMemcpyHostToDevice));
  return 0;
}

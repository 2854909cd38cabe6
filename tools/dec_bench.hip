// Developer microbench for the decode-step kernels: cycles through 30 distinct weight slabs (as the 30 layers
// do), times the steady-state launch and prints per-phase timestamps (TTS_DEC_TRACE) of a few workgroups.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DTTS_DEC_TRACE -I include -I tortoise.cpp_amd/csrc tools/dec_bench.hip \
//         tortoise.cpp_amd/csrc/host_logic.cpp -o tools/dec_bench_bin
#include "../tortoise.cpp_amd/csrc/ar.hip"
#include <algorithm>
#include <type_traits>
#include <cstdio>
#include <vector>
using namespace tts;
hipEvent_t tts::prof_event(tts_ctx *) { return nullptr; } // profiling is off in this harness

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

static void dump_trace(const char *tag, int nblocks, int nph, int slot) {
#ifdef TTS_DEC_TRACE
  std::vector<long long> all(6 * 1024 * 8);
  hipError_t e = hipMemcpyFromSymbol(all.data(), HIP_SYMBOL(tts::tts_dec_trace), all.size() * 8);
  if (e != hipSuccess) printf("memcpyFromSymbol: %s\n", hipGetErrorString(e));
  std::vector<long long> t(all.begin() + (size_t)slot * 1024 * 8, all.begin() + (size_t)(slot + 1) * 1024 * 8);
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
#else
  (void)tag; (void)nblocks; (void)nph; (void)slot;
#endif
}

#ifdef TTS_DEC_TRACE
static const char *kBuild = "TRACED";
#else
static const char *kBuild = "plain";
#endif
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
  dump_trace("fc", 256, 5, 3);
  timeit("dec_ln_gemv<QKV>", [&](int l) {
    DecLnArgs a{h, nullptr, nullptr, wqkv[l], (const __half *)wqkv[l], bvec, B, 3072, 1024, 0, q, kc, vc, ss, 256, 0};
    dec_ln_gemv_kernel<DEC_QKV, 1><<<dim3(192, 1), 256, 0, st>>>(a); }, 12.6e6);
  dump_trace("qkv", 192, 5, 0);
  timeit("dec_gemv_resid<4> fc2", [&](int l) { dec_gemv_resid_kernel<4><<<dim3(256, 1), 256, 0, st>>>(ff, B, wfc2[l], bvec, h); }, 16.8e6);
  timeit("dec_gemv_resid<4,512> fc2", [&](int l) { dec_gemv_resid_kernel<4, 512><<<dim3(256, 1), 512, 0, st>>>(ff, B, wfc2[l], bvec, h); }, 16.8e6);
  dump_trace("fc2", 256, 4, 4);
  timeit("dec_gemv_resid<1> proj", [&](int l) { dec_gemv_resid_kernel<1><<<dim3(256, 1), 256, 0, st>>>(att, B, wproj[l], bvec, h); }, 4.2e6);
  dump_trace("proj", 256, 4, 2);
  for (int np : {20, 100, 164, 250}) {
    StepState hs2{np, 3}; CK(hipMemcpy(ss, &hs2, sizeof(hs2), hipMemcpyHostToDevice));
    char tag[64];
    snprintf(tag, sizeof tag, "attn_decode (exact) nk=%d", np + 1);
    timeit(tag, [&](int l) { attn_decode_kernel<<<dim3(B, 16), 256, 0, st>>>(q, kc, vc, ss, 256, att, 0); }, 0);
    snprintf(tag, sizeof tag, "attn_decode_fast nk=%d", np + 1);
    timeit(tag, [&](int l) { attn_decode_fast_kernel<<<dim3(B, 16), 256, 0, st>>>(q, kc, vc, ss, 256, att); }, 0);
  }
  CK(hipMemcpy(ss, &hs, sizeof(hs), hipMemcpyHostToDevice));

  // ---- the decode step as it runs: 30 layers x {LN1+QKV, attention, projection, LN2+FC, MLP projection} + head, captured in a hipGraph ----
  // Per kernel of the LAST layer (every layer overwrites its slot): first workgroup start, the median workgroup's phase stamps, last workgroup end,
  // relative to the layer's first stamp; "gap" = first start of this kernel - last end of its predecessor (the dependent kernel boundary).
  auto run_chain = [&](auto ht_c) -> int {
    constexpr bool HT = decltype(ht_c)::value;
    CK(hipMemcpy(ss, &hs, sizeof(hs), hipMemcpyHostToDevice)); // n_past = 20: keys of the attention
    StepState hs3{164, 3}; CK(hipMemcpy(ss, &hs3, sizeof(hs3), hipMemcpyHostToDevice)); // mid-sequence context (P + 96)
    hipGraph_t graph; hipGraphExec_t exec;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int l = 0; l < L; l++) {
      { DecLnArgs a{h, nullptr, nullptr, wqkv[l], (const __half *)wqkv[l], bvec, B, 3072, 1024, 0, q, kc, vc, ss, 256, 0};
        dec_ln_gemv_kernel<DEC_QKV, 1, true, HT><<<dim3(192, 1), 256, 0, st>>>(a); }
      attn_decode_fast_kernel<<<dim3(B, 16), 256, 0, st>>>(q, kc, vc, ss, 256, att);
      dec_gemv_resid_kernel<1, 256, 0, true, HT><<<dim3(256, 1), 256, 0, st>>>(att, B, wproj[l], bvec, h);
      { DecLnArgs a{h, nullptr, nullptr, wfc[l], (const __half *)wfc[l], bvec, B, 4096, 0, 0, ff, nullptr, nullptr, ss, 0, 0};
        dec_ln_gemv_kernel<DEC_GELU, 1, true, HT><<<dim3(256, 1), 256, 0, st>>>(a); }
      dec_gemv_resid_kernel<4, 512, 0, true, HT><<<dim3(256, 1), 512, 0, st>>>(ff, B, wfc2[l], bvec, h);
    }
    CK(hipStreamEndCapture(st, &graph));
    CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    for (int i = 0; i < 5; i++) CK(hipGraphLaunch(exec, st));
    CK(hipEventRecord(e0, st));
    const int reps = 50;
    for (int i = 0; i < reps; i++) CK(hipGraphLaunch(exec, st));
    CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("\nlayer chain in a hipGraph (%d layers x 5 launches, B = %d, %d keys, %s build, residual stream %s): %.1f us per replay = %.2f us per layer\n", L, B, hs3.n_past + 1, kBuild, HT ? "in the h4 layout (round 4)" : "as [row][1024] (round 3)",
           1e3 * ms / reps, 1e3 * ms / reps / L);
#ifdef TTS_DEC_TRACE
    std::vector<long long> all(6 * 1024 * 8);
    CK(hipMemcpyFromSymbol(all.data(), HIP_SYMBOL(tts::tts_dec_trace), all.size() * 8));
    struct K { const char *name; int slot, nwg, nph; const char *phases; };
    const K ks[5] = {{"LN1 + QKV (dec_ln_gemv)", 0, 192, 6, "start | loads issued | LayerNorm done (x arrived, 2 barriers) | weights arrived + MFMAs | cross-wave barrier | stored"},
                     {"attention (attn_decode_fast)", 1, B * 16, 4, "start | K/V round trip + softmax + PV | barrier | stored"},
                     {"attention projection (dec_gemv_resid<1>)", 2, 256, 5, "start | weights + activations arrived, FMAs | butterfly | barrier | stored"},
                     {"LN2 + FC (dec_ln_gemv)", 3, 256, 6, "start | loads issued | LayerNorm done | weights arrived + MFMAs | cross-wave barrier | stored"},
                     {"MLP projection (dec_gemv_resid<4,512>)", 4, 256, 5, "start | weights + activations arrived, FMAs | butterfly | barrier | stored"}};
    long long t0 = all[(size_t)ks[0].slot * 1024 * 8];
    for (int b = 0; b < ks[0].nwg; b++) t0 = std::min(t0, all[((size_t)ks[0].slot * 1024 + b) * 8]);
    long long prev_end = 0;
    for (int k = 0; k < 5; k++) {
      std::vector<std::vector<long long>> ph(ks[k].nph);
      long long first = 1ll << 62, last = 0;
      for (int b = 0; b < ks[k].nwg; b++) {
        const long long *p = &all[((size_t)ks[k].slot * 1024 + b) * 8];
        first = std::min(first, p[0]); last = std::max(last, p[ks[k].nph - 1]);
        for (int i = 0; i < ks[k].nph; i++) ph[i].push_back(p[i] - p[0]);
      }
      printf("  %-42s first start %6.2f us  (gap to predecessor's last end: %5.2f us)   last end %6.2f us   [kernel span %5.2f us]\n", ks[k].name, (first - t0) * 0.01,
             k ? (first - prev_end) * 0.01 : 0.0, (last - t0) * 0.01, (last - first) * 0.01);
      printf("      median workgroup, us since its own start: ");
      for (int i = 0; i < ks[k].nph; i++) { std::sort(ph[i].begin(), ph[i].end()); printf(" %5.2f", ph[i][ph[i].size() / 2] * 0.01); }
      printf("    (%s)\n", ks[k].phases);
      prev_end = last;
    }
#endif
    return 0;
  };
  if (run_chain(std::false_type{})) return 1;
  if (run_chain(std::true_type{})) return 1;
  return 0;
}

// Developer tool (round 3): the product GEMM kernels (csrc/gemm_f16.h) against the round-2 one-tile-per-workgroup kernels (tools/gemm_f16_onetile.h) on the
// benchmark's shapes. Policies: `arith=<h>` = the product's arithmetic tile walk (h = tile height in 16-row blocks, 0 = automatic), `<name>=h,h,..[/cn][/order]` = an
// explicit per-XCD tile table (tools/gemm_tile_tables.h) of cyclic heights, `big=0` = the rejected 256-column 8-phase kernel (tools/gemm_f16_big.h; build with
// -DBIG_STAGGER=0/1 -DBIG_PREFETCH=0/1 for its variants). Every output is compared bit for bit with the round-2 kernel's; timings are interleaved rounds in one
// process (median and min). (The per-tile phase stamps and the K-loop ablation builds of round 3 — profiles/r3_gemm_epilogue.txt, r3_gemm_kloop_ablation.txt — needed instrumentation
// inside csrc/gemm_f16.h; it was removed from the product header in round 4, the results stay in profiles/.)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -I tortoise.cpp_amd/csrc -I tools -I tools/attic tools/gemm_tab_bench.hip -o tools/bin/gemm_tab_bench
//   tools/bin/gemm_tab_bench [shape-filter] [policy ...]      e.g.  tools/bin/gemm_tab_bench conv3 arith=0 u7=7 mix=8,6/8/0 big=0
//   results: profiles/r3_gemm_tile_tables.txt, r3_gemm_epilogue.txt, r3_gemm_256col_kernel.txt
#define tts tts_r2 // the round-2 one-tile-per-workgroup kernels, for reference output and timing
#include "gemm_f16_onetile.h"
#undef tts
#include "gemm_f16.h"
#include "gemm_tile_tables.h"
#include "gemm_f16_big.h" // the rejected 256-column kernel: policy name "big"
#include "gemm_f16_wide.h" // 128 x 256 tiles with 8 waves on the single-stage K loop: policy name "wide"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <map>
using namespace tts;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void touch_kernel(const uint4 *p, size_t n, unsigned *sink) {
  unsigned a = 0;
  for (size_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a ^= p[i].x;
  if (a == 0x12345678u) *sink = a;
}

struct Shape { const char *name; int M, N, K, nseg, mode, resid; };
struct Policy { std::string name; GemmPlanSpec sp; };

static Policy parse_policy(const char *s) {
  Policy p;
  std::string str(s);
  size_t eq = str.find('=');
  p.name = str.substr(0, eq);
  std::string rest = str.substr(eq + 1);
  std::vector<std::string> parts;
  size_t pos = 0;
  while (true) { size_t q = rest.find('/', pos); parts.push_back(rest.substr(pos, q == std::string::npos ? q : q - pos)); if (q == std::string::npos) break; pos = q + 1; }
  p.sp.nh = 0;
  { const std::string &hs = parts[0]; size_t a = 0; while (a < hs.size()) { size_t b = hs.find(',', a); p.sp.h[p.sp.nh++] = atoi(hs.substr(a, b == std::string::npos ? b : b - a).c_str()); if (b == std::string::npos) break; a = b + 1; } }
  if (parts.size() > 1) p.sp.cn = atoi(parts[1].c_str());
  if (parts.size() > 2) p.sp.order = atoi(parts[2].c_str());
  return p;
}

int main(int argc, char **argv) {
  const int Mmax = 28672, Kmax = 1024;
  std::vector<Shape> shapes = {
      {"in_layers k1 N1024 K1024", 28032, 1024, 1024, 1, GEMM_OUT_F32, 0},
      {"proj_out  k1 N1024 K1024 +resid", 28032, 1024, 1024, 1, GEMM_OUT_F32, 1},
      {"qkv       k1 N3072 K1024", 28032, 3072, 1024, 1, GEMM_OUT_QKV, 0},
      {"conv3     k3 N1024 K3x1024 +resid", 28032, 1024, 1024, 3, GEMM_OUT_F32, 1},
      {"pad_in_layers k1 N1024 K1024 M28672", 28672, 1024, 1024, 1, GEMM_OUT_F32, 0},
      {"pad_proj_out k1 N1024 K1024 M28672 +resid", 28672, 1024, 1024, 1, GEMM_OUT_F32, 1},
      {"pad_qkv k1 N3072 K1024 M28672", 28672, 3072, 1024, 1, GEMM_OUT_QKV, 0},
      {"integ k1  N1024 K1024 M14848", 14848, 1024, 1024, 1, GEMM_OUT_F32, 0},
      {"integ k3  N1024 K3x1024 M14848 +resid", 14848, 1024, 1024, 3, GEMM_OUT_F32, 1},
      {"single k1 N1024 K1024 M1792", 1792, 1024, 1024, 1, GEMM_OUT_F32, 1},
      {"single k3 N1024 K3x1024 M1792 +resid", 1792, 1024, 1024, 3, GEMM_OUT_F32, 1},
      {"single qkv N3072 K1024 M1792", 1792, 3072, 1024, 1, GEMM_OUT_QKV, 0},
  };
  const char *filter = argc > 1 ? argv[1] : "";
  std::vector<Policy> pols;
  for (int i = 2; i < argc; i++) pols.push_back(parse_policy(argv[i]));
  if (pols.empty()) {
    const char *def[] = {"arith=0", "u8=8", "u7=7", "m86=8,6"};
    for (const char *d : def) pols.push_back(parse_policy(d));
  }
  std::vector<__half> hA((size_t)(Mmax + 2) * Kmax), hW((size_t)3072 * 3 * Kmax);
  srand(1);
  for (auto &v : hA) v = __float2half((rand() % 2001 - 1000) / 1000.f);
  for (auto &v : hW) v = __float2half((rand() % 2001 - 1000) / 4000.f);
  __half *dA, *dW, *dH, *dVt, *dH2, *dVt2; float *dC, *dC2, *dRes, *dBias; int *dSeq;
  const size_t hbytes = (size_t)(Mmax + 128) * 2048 * 2, vbytes = (size_t)1024 * (Mmax + 128) * 2, cbytes = (size_t)Mmax * 1024 * 4;
  CK(hipMalloc(&dA, hA.size() * 2)); CK(hipMalloc(&dW, hW.size() * 2)); CK(hipMalloc(&dC, cbytes)); CK(hipMalloc(&dC2, cbytes));
  CK(hipMalloc(&dRes, cbytes)); CK(hipMalloc(&dBias, 3072 * 4));
  CK(hipMalloc(&dH, hbytes)); CK(hipMalloc(&dVt, vbytes)); CK(hipMalloc(&dH2, hbytes)); CK(hipMalloc(&dVt2, vbytes)); CK(hipMalloc(&dSeq, Mmax * 4));
  CK(hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dW, hW.data(), hW.size() * 2, hipMemcpyHostToDevice));
  {
    std::vector<float> r((size_t)Mmax * 1024), b(3072);
    for (auto &v : r) v = (rand() % 2001 - 1000) / 500.f;
    for (auto &v : b) v = (rand() % 2001 - 1000) / 1000.f;
    CK(hipMemcpy(dRes, r.data(), r.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dBias, b.data(), b.size() * 4, hipMemcpyHostToDevice));
    std::vector<int> seq(Mmax, 0);
    for (int i = 0; i < Mmax; i += 877) seq[i] = -1; // some guard rows
    CK(hipMemcpy(dSeq, seq.data(), Mmax * 4, hipMemcpyHostToDevice));
  }
  char *dFlush; unsigned *dSink; CK(hipMalloc(&dFlush, (size_t)1 << 30)); CK(hipMalloc(&dSink, 4));
  hipStream_t s; CK(hipStreamCreate(&s));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  int4 *dTab; CK(hipMalloc(&dTab, (size_t)8 * 65536 * sizeof(int4)));
  std::vector<std::vector<int4>> tabs;
  const int rounds = 3, iters = 12;
  for (const Shape &sh : shapes) {
    if (filter[0] && strcmp(filter, "all") && !strstr(sh.name, filter)) continue;
    auto mk = [&](float *outF, __half *outH, __half *outVt) {
      GemmArgs g{};
      const int lda = sh.K;
      for (int i = 0; i < 3; i++) { g.A[i] = dA + lda; g.row_off[i] = sh.nseg == 3 ? i - 1 : 0; }
      g.nseg = sh.nseg; g.kseg = sh.K; g.lda = lda; g.W = dW; g.M = sh.M; g.N = sh.N; g.bias = dBias; g.row_seq = dSeq;
      g.mode = sh.mode; g.outF = outF; g.ldo = sh.N; g.resid = sh.resid ? dRes : nullptr;
      g.outH = outH; g.ldh = 2048; g.outVt = outVt; g.ldvt = Mmax + 128;
      return g;
    };
    const double fl = 2.0 * sh.M * sh.N * (double)sh.K * sh.nseg;
    printf("== %s  (%.1f GFLOP)\n", sh.name, fl * 1e-9);
    // reference output from the one-tile kernels
    CK(hipMemset(dC, 0, cbytes)); CK(hipMemset(dH, 0, hbytes)); CK(hipMemset(dVt, 0, vbytes));
    auto mk_r2 = [&](float *outF, __half *outH, __half *outVt) {
      tts_r2::GemmArgs g{};
      const int lda = sh.K;
      for (int i = 0; i < 3; i++) { g.A[i] = dA + lda; g.row_off[i] = sh.nseg == 3 ? i - 1 : 0; }
      g.nseg = sh.nseg; g.kseg = sh.K; g.lda = lda; g.W = dW; g.M = sh.M; g.N = sh.N; g.bias = dBias; g.row_seq = dSeq;
      g.mode = sh.mode; g.outF = outF; g.ldo = sh.N; g.resid = sh.resid ? dRes : nullptr;
      g.outH = outH; g.ldh = 2048; g.outVt = outVt; g.ldvt = Mmax + 128;
      return g;
    };
    { tts_r2::GemmArgs gref = mk_r2(dC, dH, dVt); CK(tts_r2::launch_gemm_f16(gref, s)); CK(hipStreamSynchronize(s)); }
    std::vector<float> refC; std::vector<__half> refH, refV;
    if (sh.mode == GEMM_OUT_F32) { refC.resize((size_t)sh.M * sh.N); CK(hipMemcpy(refC.data(), dC, refC.size() * 4, hipMemcpyDeviceToHost)); }
    else { refH.resize(hbytes / 2); refV.resize(vbytes / 2); CK(hipMemcpy(refH.data(), dH, hbytes, hipMemcpyDeviceToHost)); CK(hipMemcpy(refV.data(), dVt, vbytes, hipMemcpyDeviceToHost)); }
    // plans
    std::vector<GemmPlan> plans(pols.size());
    std::vector<std::string> status(pols.size());
    for (size_t pi = 0; pi < pols.size(); pi++) {
      const bool arith = pols[pi].name.rfind("arith", 0) == 0 || pols[pi].name == "big" || pols[pi].name == "wide"; // arithmetic tile walk (th = first height, 0 = automatic) / 256-column kernel
      if (!arith) {
        gemm_plan_build(plans[pi], sh.M, sh.N, pols[pi].sp);
        CK(hipMalloc(&plans[pi].dev, plans[pi].host.size() * sizeof(int4)));
        CK(hipMemcpy(plans[pi].dev, plans[pi].host.data(), plans[pi].host.size() * sizeof(int4), hipMemcpyHostToDevice));
      }
      CK(hipMemset(dC2, 0xff, cbytes)); CK(hipMemset(dH2, 0, hbytes)); CK(hipMemset(dVt2, 0, vbytes));
      GemmArgs g = mk(dC2, dH2, dVt2);
      g.tiles = plans[pi].dev; g.tab_len = plans[pi].len; g.th = arith ? pols[pi].sp.h[0] : 0;
      if (pols[pi].name == "big") { g.th = 0; if (!gemm_use_big(g)) { status[pi] = "n/a (problem too small for the 256-column kernel)"; continue; } CK(launch_gemm_f16_big(g, s)); }
      else if (pols[pi].name == "wide") { g.th = 0; if (!gemm_use_wide(g)) { status[pi] = "n/a (M % 1024 != 0)"; continue; } CK(launch_gemm_f16_wide(g, s)); }
      else CK(launch_gemm_f16(g, s));
      CK(hipStreamSynchronize(s));
      size_t bad = 0;
      if (sh.mode == GEMM_OUT_F32) {
        std::vector<float> c(refC.size());
        CK(hipMemcpy(c.data(), dC2, c.size() * 4, hipMemcpyDeviceToHost));
        bad = memcmp(c.data(), refC.data(), c.size() * 4) ? 1 : 0;
        if (bad) { bad = 0; for (size_t i = 0; i < c.size(); i++) bad += memcmp(&c[i], &refC[i], 4) != 0; }
      } else {
        std::vector<__half> h(refH.size()), v(refV.size());
        CK(hipMemcpy(h.data(), dH2, hbytes, hipMemcpyDeviceToHost)); CK(hipMemcpy(v.data(), dVt2, vbytes, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < h.size(); i++) bad += memcmp(&h[i], &refH[i], 2) != 0;
        for (size_t i = 0; i < v.size(); i++) bad += memcmp(&v[i], &refV[i], 2) != 0;
      }
      char buf[96];
      snprintf(buf, sizeof buf, "%s (%d tiles, %d per XCD)", bad ? "MISMATCH" : "bit-identical", plans[pi].tiles, plans[pi].len);
      if (bad) snprintf(buf + strlen(buf), sizeof buf - strlen(buf), " %zu elements differ", bad);
      status[pi] = buf;
    }
    // timing: variant -1 = one-tile kernels
    std::vector<std::vector<double>> us(pols.size() + 1);
    for (int r = 0; r < rounds; r++)
      for (int v = -1; v < (int)pols.size(); v++) {
        GemmArgs g = mk(dC2, dH2, dVt2);
        tts_r2::GemmArgs g2 = mk_r2(dC2, dH2, dVt2);
        if (v >= 0) { g.tiles = plans[v].dev; g.tab_len = plans[v].len; g.th = pols[v].name.rfind("arith", 0) == 0 ? pols[v].sp.h[0] : 0; }
        const bool big = v >= 0 && pols[v].name == "big" && gemm_use_big(g);
        const bool wide = v >= 0 && pols[v].name == "wide" && gemm_use_wide(g);
        if (wide) { g.tiles = nullptr; }
        auto go = [&]() { return v < 0 ? tts_r2::launch_gemm_f16(g2, s) : big ? launch_gemm_f16_big(g, s) : wide ? launch_gemm_f16_wide(g, s) : launch_gemm_f16(g, s); };
        for (int i = 0; i < 2; i++) CK(go());
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < iters; i++) CK(go());
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        us[v + 1].push_back(1000.0 * ms / iters);
      }
    // TTS_COLD=1: the in-situ condition of the single-utterance diffusion step — activations just written (re-touched after the flush),
    // weights last read a whole step ago (flushed out of L2 and the memory-side cache by a 1 GB fill); ONE launch between events, median of 15
    std::vector<double> cold(pols.size() + 1, 0.0);
    if (getenv("TTS_COLD"))
      for (int v = -1; v < (int)pols.size(); v++) {
        GemmArgs g = mk(dC2, dH2, dVt2);
        tts_r2::GemmArgs g2 = mk_r2(dC2, dH2, dVt2);
        if (v >= 0) { g.tiles = plans[v].dev; g.tab_len = plans[v].len; g.th = pols[v].name.rfind("arith", 0) == 0 ? pols[v].sp.h[0] : 0; }
        const bool big = v >= 0 && pols[v].name == "big" && gemm_use_big(g);
        const bool wide = v >= 0 && pols[v].name == "wide" && gemm_use_wide(g);
        if (wide) { g.tiles = nullptr; }
        auto go = [&]() { return v < 0 ? tts_r2::launch_gemm_f16(g2, s) : big ? launch_gemm_f16_big(g, s) : wide ? launch_gemm_f16_wide(g, s) : launch_gemm_f16(g, s); };
        std::vector<float> ts;
        for (int it = 0; it < 15; it++) {
          CK(hipMemsetAsync(dFlush, it, (size_t)1 << 30, s));
          touch_kernel<<<512, 256, 0, s>>>((const uint4 *)dA, (size_t)(sh.M + 2) * sh.K / 8, dSink);
          if (sh.resid) touch_kernel<<<512, 256, 0, s>>>((const uint4 *)dRes, (size_t)sh.M * 1024 / 4, dSink);
          CK(hipEventRecord(e0, s));
          CK(go());
          CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
          float ms1; CK(hipEventElapsedTime(&ms1, e0, e1));
          ts.push_back(ms1 * 1000.f);
        }
        std::sort(ts.begin(), ts.end());
        cold[v + 1] = ts[ts.size() / 2];
      }
    for (int v = -1; v < (int)pols.size(); v++) {
      auto &u = us[v + 1];
      std::sort(u.begin(), u.end());
      const double med = u[u.size() / 2];
      printf("  %-12s med %8.1f us  min %8.1f  %7.1f TF/s   cold-W %6.1f us   %s\n", v < 0 ? "round-2" : pols[v].name.c_str(), med, u[0], fl / (med * 1e-6) / 1e12,
             cold[v + 1], v < 0 ? "" : status[v].c_str());
    }
    for (auto &p : plans) if (p.dev) CK(hipFree(p.dev));
  }
  return 0;
}

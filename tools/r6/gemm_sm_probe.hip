// Developer tool (round 6): the latency-mode GEMM family (tortoise.cpp_amd/csrc/gemm_f16_sm.h) against straightforward reference kernels and against the
// batch kernels of gemm_f16.h, at the single-utterance shape (2 sequences x 870 frames = 1 792 packed rows). For each of the five shapes of the diffusion
// network: result check (operand transform incl. GroupNorm statistics from fixed-point sums, GEMM, epilogue, output statistics), then us / launch warm
// (back to back) and with cold weights (1 GB fill between launches, activations re-touched: the in-situ condition of the sampling step).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -I tortoise.cpp_amd/csrc -I tools/r6 -I include tools/r6/gemm_sm_probe.hip -o tools/bin/gemm_sm_probe
#include "gemm_f16_sm.h"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>
using namespace tts;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

static constexpr int C = 1024;

// reference: fp16 operand of a GroupNorm'd f32 tensor (same op order as gn_reg_kernel / the sm transform), mean / rstd given
__global__ void ref_gn_apply(const float *x, const int *row_seq, const float2 *mr, const float *ga, const float *be, const float *sc, const float *sh, int silu, __half *y) {
  const int r = blockIdx.x, s = row_seq[r];
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float u = 0.f;
    if (s >= 0) {
      const float2 m = mr[s * 32 + (c >> 5)];
      u = (x[(size_t)r * C + c] - m.x) * m.y;
      u = u * ga[c]; u = u + be[c]; u = u * (sc[c] + 1.0f); u = u + sh[c];
      if (silu) u = u * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(u * -1.44269504088896f));
    }
    y[(size_t)r * C + c] = __float2half_rn(u);
  }
}
// reference GEMM: C[m][n] = sum_tap sum_k A[m + tap - (taps == 3)][k] W[n][tap K + k] (rows outside [0, M) are zero), two A segments when nseg == 2
__global__ void ref_gemm(const __half *A0, const __half *A1, int lda, int nseg, int K, int taps, const __half *W, int ldw, int M, int N, float *out) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y;
  if (n >= N) return;
  float acc = 0.f;
  for (int t = 0; t < taps; t++) {
    const int r = m + t - (taps == 3 ? 1 : 0);
    if (r < 0 || r >= M) continue;
    for (int sgm = 0; sgm < nseg; sgm++) {
      const __half *a = (sgm ? A1 : A0) + (size_t)r * lda;
      const __half *w = W + (size_t)n * ldw + t * K + sgm * K;
      for (int k = 0; k < K; k++) acc += __half2float(a[k]) * __half2float(w[k]);
    }
  }
  out[(size_t)m * N + n] = acc;
}
__global__ void touch_kernel(const uint4 *p, size_t n, unsigned *sink) {
  unsigned a = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a ^= p[i].x;
  if (a == 0x12345678u) *sink = a;
}

static float frand(float s) { return ((rand() % 20001) - 10000) / 10000.f * s; }

int main(int argc, char **argv) {
  const int T = argc > 1 ? atoi(argv[1]) : 870, NS = argc > 2 ? atoi(argv[2]) : 2;
  // packed layout (diffusion.hip: Layout::build)
  std::vector<int> start(NS), len(NS, T);
  int r = 8;
  for (int s = 0; s < NS; s++) { start[s] = r; r = (r + len[s] + 1 + 7) & ~7; }
  const int M = (r + 127) & ~127;
  std::vector<int> row_seq(M, -1), chunk_seq(M / 8, -1);
  for (int s = 0; s < NS; s++) for (int t = 0; t < len[s]; t++) { row_seq[start[s] + t] = s; chunk_seq[(start[s] + t) >> 3] = s; }
  std::vector<int4> tile_seqs(M / 64);
  for (int t = 0; t < M / 64; t++) {
    int v[4] = {-1, -1, -1, -1}, n = 0;
    for (int rr = std::max(t * 64 - 1, 0); rr < std::min(t * 64 + 66, M); rr++) {
      const int s = row_seq[rr];
      if (s >= 0 && (n == 0 || v[n - 1] != s)) { if (n == 4) { printf("more than 4 sequences in a tile window\n"); return 1; } v[n++] = s; }
    }
    tile_seqs[t] = make_int4(v[0], v[1], v[2], v[3]);
  }
  printf("# T = %d, %d sequences: M = %d packed rows (%d row tiles of 64)\n", T, NS, M, M / 64);
  srand(7);
  std::vector<float> X((size_t)M * C, 0.f), R((size_t)M * C, 0.f), ga(C), be(C), sc(C), sh(C), zeros(C, 0.f), bias(3 * C);
  for (int m = 0; m < M; m++) if (row_seq[m] >= 0) for (int c = 0; c < C; c++) { X[(size_t)m * C + c] = frand(2.0f) + 0.7f * ((c >> 5) % 5 - 2); R[(size_t)m * C + c] = frand(1.5f); }
  for (int c = 0; c < C; c++) { ga[c] = 1.0f + frand(0.5f); be[c] = frand(0.3f); sc[c] = frand(0.4f); sh[c] = frand(0.4f); }
  for (auto &v : bias) v = frand(0.2f);
  // fixed-point statistics of X, as a producing epilogue would have left them (per 8-row chunk partials in f32, then fx)
  std::vector<long long> st((size_t)NS * 32 * 4, 0);
  std::vector<float2> mr((size_t)NS * 32);
  for (int s = 0; s < NS; s++)
    for (int gq = 0; gq < 32; gq++) {
      double S = 0, Q = 0;
      for (int t = 0; t < len[s]; t++) for (int c = 0; c < 32; c++) { const double v = X[(size_t)(start[s] + t) * C + gq * 32 + c]; S += v; Q += v * v; }
      auto put = [&](long long *d, double v) { const double h = nearbyint(v * 256.0); d[0] = (long long)h; d[1] = (long long)nearbyint((v - h / 256.0) * 1152921504606846976.0); };
      put(&st[(size_t)(s * 32 + gq) * 4], S); put(&st[(size_t)(s * 32 + gq) * 4 + 2], Q);
      const double n = len[s] * 32.0, mean = S / n, var = std::max(Q / n - mean * mean, 0.0);
      mr[s * 32 + gq] = make_float2((float)mean, (float)(1.0 / sqrt(var + 1e-6)));
    }
  std::vector<__half> W((size_t)3 * C * 3 * C), Wsp((size_t)C * 2 * C), A16a((size_t)(M + 2) * C, __float2half(0.f)), A16b((size_t)(M + 2) * C, __float2half(0.f));
  for (auto &v : W) v = __float2half(frand(0.05f));
  std::vector<float> Wf((size_t)C * C);
  for (auto &v : Wf) v = frand(0.05f);
  for (int n = 0; n < C; n++) for (int k = 0; k < C; k++) { const float w = Wf[(size_t)n * C + k] * 64.f; const __half hi = __float2half_rn(w); Wsp[(size_t)n * 2 * C + k] = hi; Wsp[(size_t)n * 2 * C + C + k] = __float2half_rn(w - __half2float(hi)); }
  for (int m = 0; m < M; m++) if (row_seq[m] >= 0) for (int c = 0; c < C; c++) { A16a[(size_t)(m + 1) * C + c] = __float2half(frand(1.f)); A16b[(size_t)(m + 1) * C + c] = __float2half(frand(1.f)); }

  float *dX, *dR, *dga, *dbe, *dsc, *dsh, *dz, *dbias, *dOut, *dRef; float2 *dmr; long long *dst, *dsto; int *drs, *dcs, *dlen; int4 *dts;
  __half *dW, *dWsp, *dA16a, *dA16b, *dOp, *dQK, *dVt; char *dFlush; unsigned *dSink;
  const int ldvt = M + 128;
  CK(hipMalloc(&dX, X.size() * 4)); CK(hipMalloc(&dR, R.size() * 4)); CK(hipMalloc(&dga, C * 4)); CK(hipMalloc(&dbe, C * 4)); CK(hipMalloc(&dsc, C * 4)); CK(hipMalloc(&dsh, C * 4));
  CK(hipMalloc(&dz, C * 4)); CK(hipMalloc(&dbias, 3 * C * 4)); CK(hipMalloc(&dOut, (size_t)M * 3 * C * 4)); CK(hipMalloc(&dRef, (size_t)M * 3 * C * 4)); CK(hipMalloc(&dmr, mr.size() * 8));
  CK(hipMalloc(&dst, st.size() * 8)); CK(hipMalloc(&dsto, st.size() * 8)); CK(hipMalloc(&drs, M * 4)); CK(hipMalloc(&dcs, M / 8 * 4)); CK(hipMalloc(&dlen, NS * 4)); CK(hipMalloc(&dts, M / 64 * 16));
  CK(hipMalloc(&dW, W.size() * 2)); CK(hipMalloc(&dWsp, Wsp.size() * 2)); CK(hipMalloc(&dA16a, A16a.size() * 2)); CK(hipMalloc(&dA16b, A16b.size() * 2)); CK(hipMalloc(&dOp, (size_t)(M + 2) * C * 2));
  CK(hipMalloc(&dQK, (size_t)(M + 128) * 2048 * 2)); CK(hipMalloc(&dVt, (size_t)C * ldvt * 2)); CK(hipMalloc(&dFlush, (size_t)1 << 30)); CK(hipMalloc(&dSink, 4));
  CK(hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dR, R.data(), R.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dga, ga.data(), C * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dbe, be.data(), C * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dsc, sc.data(), C * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dsh, sh.data(), C * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dz, zeros.data(), C * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dbias, bias.data(), 3 * C * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dmr, mr.data(), mr.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dst, st.data(), st.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(drs, row_seq.data(), M * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dcs, chunk_seq.data(), M / 8 * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dlen, len.data(), NS * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dts, tile_seqs.data(), M / 64 * 16, hipMemcpyHostToDevice));
  CK(hipMemcpy(dW, W.data(), W.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dWsp, Wsp.data(), Wsp.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(dA16a, A16a.data(), A16a.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dA16b, A16b.data(), A16b.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemset(dOp, 0, (size_t)(M + 2) * C * 2));
  hipStream_t s; CK(hipStreamCreate(&s));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));

  struct Shape { const char *name; int kind, N, taps, nseg, gn, use_ss, silu, resid; };
  const Shape shapes[] = {
      {"in_layers  gn+silu k1 N1024 K1024        ", SM_K1_GN, 1024, 1, 1, 1, 0, 1, 0},
      {"out_layers gn+ss+silu k3 N1024 K3x1024 +r", SM_K3_GN, 1024, 3, 1, 1, 1, 1, 1},
      {"qkv        gn k1 N3072 K1024             ", SM_QKV_GN, 3072, 1, 1, 1, 0, 0, 0},
      {"proj_out   f16 dual-B N1024 K1024 +r     ", SM_PROJ_DUALB, 1024, 1, 1, 0, 0, 0, 1},
      {"integ conv f16 2 seg N1024 K2x1024       ", SM_K1_F16, 1024, 1, 2, 0, 0, 0, 0},
  };
  printf("%-44s %-6s %10s %10s   %s\n", "shape", "depth", "warm us", "cold us", "checks");
  for (const Shape &sh_ : shapes) {
    struct Var { int depth; std::function<hipError_t(const GemmSmArgs &)> launch; };
    std::vector<Var> vars;
#define V(D, ...) vars.push_back({D, [&](const GemmSmArgs &a) { return launch_gemm_sm_t<__VA_ARGS__, D>(a, s); }})
    if (sh_.kind == SM_K1_GN) { V(1, GEMM_OUT_F32, 2, 1, false, true, false, true); V(2, GEMM_OUT_F32, 2, 1, false, true, false, true); V(3, GEMM_OUT_F32, 2, 1, false, true, false, true); }
    if (sh_.kind == SM_K3_GN) { V(2, GEMM_OUT_F32, 2, 3, true, true, false, true); V(4, GEMM_OUT_F32, 2, 3, true, true, false, true); V(5, GEMM_OUT_F32, 2, 3, true, true, false, true); }
    if (sh_.kind == SM_QKV_GN) { V(1, GEMM_OUT_QKV, 3, 2, false, true, false, false); V(2, GEMM_OUT_QKV, 3, 2, false, true, false, false); V(3, GEMM_OUT_QKV, 3, 2, false, true, false, false); }
    if (sh_.kind == SM_PROJ_DUALB) { V(1, GEMM_OUT_F32_SCALED, 2, 1, false, false, true, true); V(2, GEMM_OUT_F32_SCALED, 2, 1, false, false, true, true); }
    if (sh_.kind == SM_K1_F16) { V(1, GEMM_OUT_F32, 2, 1, false, false, false, true); V(3, GEMM_OUT_F32, 2, 1, false, false, false, true); V(5, GEMM_OUT_F32, 2, 1, false, false, false, true); }
#undef V
    const int N = sh_.N;
    GemmSmArgs g{};
    g.A16[0] = dA16a + C; g.A16[1] = dA16b + C; g.A32 = dX; g.lda = C; g.nseg = sh_.nseg; g.kseg = C;
    g.st_in = dst; g.gamma = dga; g.beta = dbe; g.scale = sh_.use_ss ? dsc : dz; g.shift = sh_.use_ss ? dsh : dz; g.seq_len = dlen; g.tile_seqs = dts; g.eps = 1e-6f; g.silu = sh_.silu;
    g.W = sh_.kind == SM_PROJ_DUALB ? dWsp : dW; g.ldw = sh_.kind == SM_PROJ_DUALB ? 2 * C : sh_.taps * sh_.nseg * C; g.w_lo_off = C;
    g.M = M; g.N = N; g.bias = dbias; g.row_seq = drs;
    g.outF = dOut; g.ldo = N; g.resid = sh_.resid ? dR : nullptr; g.alpha = 1.0f / 64.0f;
    g.outH = dQK; g.ldh = 2048; g.outVt = dVt; g.ldvt = ldvt;
    g.st_out = dsto; g.chunk_seq = dcs;
    // batch-path equivalent (fp16 operand prepared by the reference GroupNorm: its GroupNorm launch is NOT in the batch timing)
    const __half *opA = sh_.gn ? dOp + C : dA16a + C;
    if (sh_.gn) ref_gn_apply<<<M, 256, 0, s>>>(dX, drs, dmr, dga, dbe, g.scale, g.shift, sh_.silu, dOp + C);
    GemmArgs b{};
    for (int i = 0; i < 3; i++) { b.A[i] = opA; b.row_off[i] = sh_.taps == 3 ? i - 1 : 0; }
    if (sh_.nseg == 2) b.A[1] = dA16b + C;
    b.nseg = sh_.taps == 3 ? 3 : sh_.nseg; b.kseg = C; b.lda = C; b.W = g.W; b.M = M; b.N = N; b.bias = dbias; b.row_seq = drs;
    b.mode = sh_.kind == SM_QKV_GN ? GEMM_OUT_QKV : sh_.kind == SM_PROJ_DUALB ? GEMM_OUT_F32_SCALED : GEMM_OUT_F32;
    b.outF = dRef; b.ldo = N; b.resid = sh_.resid ? dR : nullptr; b.alpha = 1.0f / 64.0f; b.outH = dQK; b.ldh = 2048; b.outVt = dVt; b.ldvt = ldvt;
    if (sh_.kind == SM_PROJ_DUALB) { b.nseg = 2; b.custom_w = 1; b.ldw_ = 2 * C; b.w_off_[0] = 0; b.w_off_[1] = C; b.dual_b = 1; }
    // ---- timing
    auto time_warm = [&](auto &&launch) {
      for (int i = 0; i < 5; i++) launch();
      CK(hipEventRecord(e0, s));
      for (int i = 0; i < 50; i++) launch();
      CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      return 1000.0 * ms / 50;
    };
    auto time_cold = [&](auto &&launch) {
      std::vector<float> ts;
      for (int it = 0; it < 11; it++) {
        CK(hipMemsetAsync(dFlush, it, (size_t)1 << 30, s));
        touch_kernel<<<512, 256, 0, s>>>((const uint4 *)dX, (size_t)M * C / 4, dSink);
        touch_kernel<<<512, 256, 0, s>>>((const uint4 *)dR, (size_t)M * C / 4, dSink);
        touch_kernel<<<512, 256, 0, s>>>((const uint4 *)dOp, (size_t)(M + 2) * C / 8, dSink);
        touch_kernel<<<512, 256, 0, s>>>((const uint4 *)dA16a, (size_t)(M + 2) * C / 8, dSink);
        touch_kernel<<<512, 256, 0, s>>>((const uint4 *)dA16b, (size_t)(M + 2) * C / 8, dSink);
        CK(hipEventRecord(e0, s));
        launch();
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms1; CK(hipEventElapsedTime(&ms1, e0, e1));
        ts.push_back(ms1 * 1000.f);
      }
      std::sort(ts.begin(), ts.end());
      return (double)ts[ts.size() / 2];
    };
    for (const Var &var : vars) {
      char chk[320];
    // ---- checks
    {
      CK(hipMemsetAsync(dOut, 0xff, (size_t)M * N * 4, s)); CK(hipMemsetAsync(dQK, 0xff, (size_t)(M + 128) * 2048 * 2, s)); CK(hipMemsetAsync(dVt, 0xff, (size_t)C * ldvt * 2, s));
      CK(hipMemsetAsync(dsto, 0, st.size() * 8, s));
      CK(var.launch(g));
      // reference product in f32 (for the dual-B shape: against the hi and lo halves as two segments of K, alpha applied on the host)
      if (sh_.kind == SM_PROJ_DUALB) ref_gemm<<<dim3((N + 255) / 256, M), 256, 0, s>>>(opA, opA, C, 2, C, 1, dWsp, 2 * C, M, N, dRef);
      else ref_gemm<<<dim3((N + 255) / 256, M), 256, 0, s>>>(opA, dA16b + C, C, sh_.nseg, C, sh_.taps, dW, g.ldw, M, N, dRef);
      CK(hipStreamSynchronize(s));
      std::vector<float> ref((size_t)M * N), out((size_t)M * N);
      CK(hipMemcpy(ref.data(), dRef, ref.size() * 4, hipMemcpyDeviceToHost));
      double maxd = 0, maxref = 0; size_t bad_guard = 0;
      std::vector<float> want((size_t)M * N, 0.f);
      for (int m = 0; m < M; m++)
        for (int n = 0; n < N; n++) {
          float w = 0.f;
          if (row_seq[m] >= 0) w = ref[(size_t)m * N + n] * (sh_.kind == SM_PROJ_DUALB ? 1.0f / 64.0f : 1.0f) + bias[n] + (sh_.resid ? R[(size_t)m * C + n] : 0.f);
          want[(size_t)m * N + n] = w;
        }
      if (sh_.kind == SM_QKV_GN) {
        std::vector<__half> qk((size_t)(M + 128) * 2048), vt((size_t)C * ldvt);
        CK(hipMemcpy(qk.data(), dQK, qk.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(vt.data(), dVt, vt.size() * 2, hipMemcpyDeviceToHost));
        for (int m = 0; m < M; m++)
          for (int n = 0; n < N; n++) {
            const int h = n / 192, w = n % 192;
            const float got = w < 128 ? __half2float(qk[(size_t)m * 2048 + h * 128 + w]) : __half2float(vt[(size_t)(h * 64 + w - 128) * ldvt + m]);
            const float wv = want[(size_t)m * N + n];
            maxd = std::max(maxd, (double)fabsf(got - wv) - 1e-3 * fabsf(wv)); maxref = std::max(maxref, (double)fabsf(wv));
            if (row_seq[m] < 0 && got != 0.f) bad_guard++;
          }
        snprintf(chk, sizeof chk, "max |err| beyond 1e-3 rel %.2e (|ref| <= %.1f), nonzero guard outputs %zu", std::max(maxd, 0.0), maxref, bad_guard);
      } else {
        CK(hipMemcpy(out.data(), dOut, out.size() * 4, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < out.size(); i++) { maxd = std::max(maxd, (double)fabsf(out[i] - want[i])); maxref = std::max(maxref, (double)fabsf(want[i])); if (row_seq[i / N] < 0 && out[i] != 0.f) bad_guard++; }
        // output statistics against double sums of the stored values
        std::vector<long long> so(st.size());
        CK(hipMemcpy(so.data(), dsto, so.size() * 8, hipMemcpyDeviceToHost));
        double srel = 0;
        for (int q = 0; q < NS; q++)
          for (int gq = 0; gq < 32; gq++) {
            double S = 0, Q = 0;
            for (int t = 0; t < len[q]; t++) for (int c = 0; c < 32; c++) { const double v = out[(size_t)(start[q] + t) * N + gq * 32 + c]; S += v; Q += v * v; }
            const long long *p = &so[(size_t)(q * 32 + gq) * 4];
            srel = std::max(srel, fabs(fx_value(p[0], p[1]) - S) / sqrt(len[q] * 32.0 * Q)); // error of the mean in units of the rms
            srel = std::max(srel, fabs(fx_value(p[2], p[3]) - Q) / Q);
          }
        snprintf(chk, sizeof chk, "max |err| %.2e (|ref| <= %.1f), nonzero guard outputs %zu, output stats rel err %.1e", maxd, maxref, bad_guard, srel);
      }
      // determinism: a second run must give the same bits (outputs and statistics)
      std::vector<float> out2((size_t)M * N); std::vector<long long> so1(st.size()), so2(st.size());
      CK(hipMemcpy(so1.data(), dsto, so1.size() * 8, hipMemcpyDeviceToHost));
      CK(hipMemsetAsync(dsto, 0, st.size() * 8, s));
      CK(var.launch(g));
      CK(hipStreamSynchronize(s));
      CK(hipMemcpy(out2.data(), dOut, out2.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(so2.data(), dsto, so2.size() * 8, hipMemcpyDeviceToHost));
      if (sh_.kind != SM_QKV_GN && (memcmp(out.data(), out2.data(), out.size() * 4) || memcmp(so1.data(), so2.data(), so1.size() * 8))) strcat(chk, " NOT-REPRODUCIBLE");
    }
      auto l_sm = [&] { CK(var.launch(g)); };
      printf("%-44s D = %-2d %10.1f %10.1f   %s\n", sh_.name, var.depth, time_warm(l_sm), time_cold(l_sm), chk);
#ifdef SM_TRACE
      {
        CK(hipStreamSynchronize(s));
        long long tr[64 * 8];
        CK(hipMemcpyFromSymbol(tr, HIP_SYMBOL(sm_trace), sizeof tr));
        const int nphs = std::min(64, sh_.nseg * 16 * (sh_.kind == SM_K3_GN ? 3 : sh_.kind == SM_QKV_GN ? 2 : 1));
        double acc7[7] = {0};
        for (int p = 1; p < nphs; p++) { for (int i = 0; i < 6; i++) acc7[i] += double(tr[p * 8 + i + 1] - tr[p * 8 + i]); acc7[6] += double(tr[p * 8] - tr[(p - 1) * 8]); }
        printf("    trace (cycles / phase, wave 0 of workgroup 0, last warm launch): count %.0f | vmcnt+lgkm wait %.0f | barrier %.0f | issue %.0f | frag+mfma %.0f | transform %.0f || phase period %.0f\n",
               acc7[0] / (nphs - 1), acc7[1] / (nphs - 1), acc7[2] / (nphs - 1), acc7[3] / (nphs - 1), acc7[4] / (nphs - 1), acc7[5] / (nphs - 1), acc7[6] / (nphs - 1));
      }
#endif
      fflush(stdout);
    }
    // the batch kernel, K tiles per barrier pair (ku) x tile height in 16-row blocks (th; 0 = the launcher's choice): bit-identical results required
    std::vector<float> base((size_t)M * N);
    std::vector<__half> baseqk((size_t)(M + 128) * 2048);
    for (int ku : {1, 2, 4})
      for (int th : {0, 2, 4, 8}) {
        if ((sh_.taps == 3 && ku > 1) || (sh_.kind == SM_PROJ_DUALB && ku > 2)) continue;
        GemmArgs bb = b; bb.ku = ku; bb.th = th;
        CK(hipMemsetAsync(dRef, 0xff, (size_t)M * N * 4, s)); CK(hipMemsetAsync(dQK, 0xff, (size_t)(M + 128) * 2048 * 2, s));
        CK(launch_gemm_f16(bb, s)); CK(hipStreamSynchronize(s));
        std::vector<float> o((size_t)M * N); std::vector<__half> oqk((size_t)(M + 128) * 2048);
        CK(hipMemcpy(o.data(), dRef, o.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(oqk.data(), dQK, oqk.size() * 2, hipMemcpyDeviceToHost));
        const bool qkv = sh_.kind == SM_QKV_GN;
        if (ku == 1 && th == 0) { base = o; baseqk = oqk; }
        const bool same = qkv ? !memcmp(oqk.data(), baseqk.data(), (size_t)M * 2048 * 2) : !memcmp(o.data(), base.data(), o.size() * 4);
        auto l_b = [&] { CK(launch_gemm_f16(bb, s)); };
        printf("%-44s batch ku=%d th=%d %8.1f %10.1f   %s\n", sh_.name, ku, th, time_warm(l_b), time_cold(l_b), same ? "bit-identical to ku=1 th=auto" : "DIFFERS");
        fflush(stdout);
      }
  }
  return 0;
}

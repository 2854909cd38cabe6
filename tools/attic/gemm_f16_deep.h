// Measured-and-rejected (round 2): deep-ring GEMM kernel for small problems. NOT part of the product: only tools/gemm_small_diag.hip defines
// TTS_GEMM_DEEP and includes this (from inside namespace tts in gemm_f16.h, after the product kernels).
// Result on MI355X (profiles/r2_gemm_small_problems.txt): bit-identical to the product kernels and NOT faster — at M = 1 792 (one utterance) a k = 1
// GEMM takes 13.8 us on the single-stage kernel and 13.9-14.3 us with 2 or 5 K tiles in flight; the k = 3 kernel 27.0 vs 32.3 us. A small launch is
// bound by its fixed costs (dispatch, prologue, epilogue), not by the K loop's load latency.

// Small problems (a single utterance: M = 1 792 rows -> 224 tiles of 64 x 128 on 256 CUs). With one workgroup per CU nothing overlaps the
// single-stage kernel's load -> wait -> multiply sequence, and the k = 3 kernel prefetches only one tap ahead: a K tile costs a full
// L2 / HBM round trip (16-48 of them per launch). This kernel keeps S - 1 K tiles in flight in an S-deep LDS ring (LDS-DMA, counted vmcnt,
// ONE raw barrier per K tile): the stage consumed in iteration t-1 is refilled right after iteration t's barrier. Every wave issues
// MI + 4 pieces per tile, and past the last tile the issue stream re-requests the last tile into the free stage, so the vmcnt arithmetic
// is the same constant in every iteration. Accumulation order = the product kernels' (segment-major, or chunk-major with the taps
// innermost for the k = 3 convolution: g.kmajor), the residual enters where theirs does (accumulators for plain segments, epilogue for
// k = 3) — results are bit-identical to the large-problem kernels, so a candidate's output does not depend on the batch it runs in.
template <int N> __device__ __forceinline__ void gemm_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int MI, int S> constexpr int deep_lds() { return S * (32 * MI * 128 + 16384); }
template <int MODE, int MI, int S>
static __global__ __launch_bounds__(256) void gemm_f16_deep_kernel(GemmArgs g) {
  constexpr int BM = 32 * MI, STAGE = BM * 128 + 16384, P = MI + 4;
  static_assert((S - 2) * P < 64, "vmcnt is a 6-bit counter");
  extern __shared__ __attribute__((aligned(16))) char smem_dyn[];
  char *smem = smem_dyn;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int MT = (g.M + BM - 1) / BM, NT = g.N >> 7;
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int mq = MT >> 3, mr = MT & 7;
  const int mcount = mq + (xcd < mr ? 1 : 0), mfirst = xcd * mq + (xcd < mr ? xcd : mr);
  if (idx >= mcount * NT) return;
  const int cn = g.cn > 0 ? g.cn : NT, per_chunk = mcount * cn;
  const int chunk = idx / per_chunk, rem = idx - chunk * per_chunk;
  const int m0 = (mfirst + rem / cn) * BM, n0 = (chunk * cn + rem % cn) << 7;
  const int tps = g.kseg >> 6, KT = g.nseg * tps;
  const int ldw = g.custom_w ? g.ldw_ : g.nseg * g.kseg;
  const int prow = lane >> 3, pslot = lane & 7;
  int aoff[MI], boff[4];
#pragma unroll
  for (int i = 0; i < MI; i++) {
    const int row = (wave * MI + i) * 8 + prow;
    aoff[i] = min(m0 + row, g.M - 1) * g.lda + (pslot ^ ((row >> 1) & 7)) * 8;
  }
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int row = (wave * 4 + i) * 8 + prow;
    boff[i] = (n0 + row) * ldw + (pslot ^ ((row >> 1) & 7)) * 8;
  }
  const int fr = lane & 15, fq = lane >> 4;
  const bool resid_first = MODE == GEMM_OUT_F32 && g.resid != nullptr && !g.kmajor;
  floatx4 acc[MI][4];
  if (resid_first) gemm_acc_from_resid<MI>(g, acc, m0, n0, wm, wn, fr, fq); // older than every DMA piece: retired first by the in-order vmcnt
  else {
#pragma unroll
    for (int i = 0; i < MI; i++)
#pragma unroll
      for (int j = 0; j < 4; j++) acc[i][j] = (floatx4){0.f, 0.f, 0.f, 0.f};
  }
  // issue stream: (is_seg, is_kt) = the next K tile to request, ist = the ring stage it goes to
  int is_seg = 0, is_kt = 0, ist = 0;
  auto issue = [&]() {
    const __half *ab = g.A[is_seg] + (ptrdiff_t)g.row_off[is_seg] * g.lda + (is_kt << 6);
    const __half *wb = g.W + (g.custom_w ? g.w_off_[is_seg] : is_seg * g.kseg) + (is_kt << 6);
    char *sa = smem + ist * STAGE, *sb = sa + BM * 128;
#pragma unroll
    for (int i = 0; i < MI; i++) __builtin_amdgcn_global_load_lds((gptr_t)(ab + aoff[i]), (lptr_t)(sa + (wave * MI + i) * 1024), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; i++) __builtin_amdgcn_global_load_lds((gptr_t)(wb + boff[i]), (lptr_t)(sb + (wave * 4 + i) * 1024), 16, 0, 0);
    ist = (ist + 1 == S) ? 0 : ist + 1;
    if (g.kmajor) {
      if (is_seg + 1 < g.nseg) is_seg++;
      else if (is_kt + 1 < tps) { is_seg = 0; is_kt++; }
    } else {
      if (is_kt + 1 < tps) is_kt++;
      else if (is_seg + 1 < g.nseg) { is_kt = 0; is_seg++; }
    } // (the last tile is re-requested from then on)
  };
  const bool natural = (MODE == GEMM_OUT_QKV) && (((n0 + wn * 64) % 192) >= 128);
  auto kloop = [&](auto nat) {
    constexpr bool NAT = decltype(nat)::value;
#pragma unroll
    for (int p = 0; p < S - 1; p++) issue();
    int cst = 0;
    for (int t = 0; t < KT; t++) {
      gemm_wait_vmcnt<(S - 2) * P>(); // all but the S-2 newest tiles: tile t has landed (this wave's pieces)
      __builtin_amdgcn_s_barrier();    // ... everyone's pieces; and every wave has left iteration t-1, whose stage is refilled now
      issue();
      const char *sa = smem + cst * STAGE, *sb = sa + BM * 128;
      cst = (cst + 1 == S) ? 0 : cst + 1;
#pragma unroll
      for (int ks = 0; ks < 2; ks++) {
        half8 af[MI], bf[4];
#pragma unroll
        for (int i = 0; i < MI; i++) af[i] = *(const half8 *)(sa + lds_off(wm * (16 * MI) + i * 16 + fr, ks * 4 + fq));
#pragma unroll
        for (int i = 0; i < 4; i++) bf[i] = *(const half8 *)(sb + lds_off(wn * 64 + i * 16 + fr, ks * 4 + fq));
#pragma unroll
        for (int i = 0; i < MI; i++)
#pragma unroll
          for (int j = 0; j < 4; j++) {
            if (NAT) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
            else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[j], af[i], acc[i][j], 0, 0, 0);
          }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // this stage's fragment reads are complete before the next barrier releases it
    }
    gemm_wait_vmcnt<0>(); // the re-requested trailing pieces must land before the LDS is released
  };
  if (MODE == GEMM_OUT_QKV && natural) kloop(std::true_type{});
  else kloop(std::false_type{});
  if (resid_first) gemm_epilogue<MODE, MI, true>(g, acc, m0, n0, wm, wn, fr, fq, g.M);
  else gemm_epilogue<MODE, MI>(g, acc, m0, n0, wm, wn, fr, fq, g.M);
}

static inline int &gemm_deep_mode() { // -1: off (TTS_GEMM_NODEEP), 0: automatic, 3 / 6: that ring depth (TTS_GEMM_DEEP_S); tools/gemm_diag toggles it
  static int v = getenv("TTS_GEMM_NODEEP") ? -1 : getenv("TTS_GEMM_DEEP_S") ? atoi(getenv("TTS_GEMM_DEEP_S")) : 0;
  return v;
}
template <int MODE, int S>
static inline void launch_gemm_deep(const GemmArgs &gg, int grid, hipStream_t s) {
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void *)gemm_f16_deep_kernel<MODE, 2, S>, hipFuncAttributeMaxDynamicSharedMemorySize, deep_lds<2, S>());
    attr = true;
  }
  gemm_f16_deep_kernel<MODE, 2, S><<<grid, 256, deep_lds<2, S>(), s>>>(gg);
}


static inline bool launch_gemm_deep_if_small(const GemmArgs &g, GemmArgs &gg, int mi, bool conv3, int grid1, int cus_per_xcd, hipStream_t s, hipError_t *err) {
  // Small problems: the deep-ring kernel (see gemm_f16_deep_kernel). 6 stages = 144 KB = one workgroup per CU when the grid is about one
  // workgroup per CU anyway, 3 stages = 72 KB = two per CU when there are more. TTS_GEMM_NODEEP / TTS_GEMM_DEEP_S=3|6 are the A/B switches (gemm_deep_mode).
  const int deep_s = gemm_deep_mode();
  if (mi == 2 && deep_s >= 0) {
    gg.kmajor = conv3 ? 1 : 0;
    const int S = (deep_s == 3 || deep_s == 6) ? deep_s : (grid1 <= cus_per_xcd * 10 ? 6 : 3);
#define TTS_LAUNCH_DEEP(S_)                                                                         \
    do {                                                                                              \
      if (g.mode == GEMM_OUT_F32) launch_gemm_deep<GEMM_OUT_F32, S_>(gg, grid1, s);                   \
      else if (g.mode == GEMM_OUT_F16) launch_gemm_deep<GEMM_OUT_F16, S_>(gg, grid1, s);              \
      else launch_gemm_deep<GEMM_OUT_QKV, S_>(gg, grid1, s);                                          \
    } while (0)
    if (S == 6) TTS_LAUNCH_DEEP(6);
    else TTS_LAUNCH_DEEP(3);
#undef TTS_LAUNCH_DEEP
    *err = hipGetLastError();
    return true;
  }
  return false;
}

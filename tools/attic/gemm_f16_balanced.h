// Measured-and-rejected: balanced persistent GEMM kernels (round 2; DESIGN.md section 5, profiles/r2_gemm_tile_phases.txt). NOT part of the
// product: only tools/gemm_diag.hip defines TTS_GEMM_DIAG, which makes gemm_f16.h include this file from inside namespace tts (after
// GemmArgs / lds_off / gemm_epilogue / gemm_acc_from_resid / conv3_lds). Kept so that the negative result stays reproducible.
#pragma once
// ------------------------------------------------------------------------------------------------------------------------------
// Balanced persistent kernels. What the per-tile traces (tools/gemm_diag.hip) showed of the one-tile-per-workgroup grids above:
// workgroups run in lockstep ROUNDS (every tile takes the same time), 1752 tiles on 768 / 1024 slots leave a last round that is 28 % /
// 71 % full but costs a full round (a lone workgroup is latency-bound: it finishes its tile barely faster than in a crowded CU),
// and all epilogues of a round hit HBM together (5.5 TB/s bursts with the matrix pipe idle: 25 % of the launch).
// Here every workgroup instead owns a contiguous run of rows of ONE n-tile column, sized so that all workgroups get the same number of
// 32-row units (+-1): its rows are processed as chunks of 128 rows plus one chunk of 32 / 64 / 96 — all workgroups finish together,
// no partial round. Odd row groups take their short chunk first, so the mid-launch epilogue bursts of neighbouring groups fall at
// different times and overlap the other group's K loop.
//   decomposition: XCD x owns a contiguous range of units (as before: its rows and the n-chunk's weights stay in its L2); its S
//   workgroups form S / cn row groups x cn n-tiles; n-chunks (cn n-tiles, weights <= ~2.5 MB) are walked in sequence.
// ------------------------------------------------------------------------------------------------------------------------------
template <int MODE, int MI>
__device__ __forceinline__ void glds_chunk(const GemmArgs &g, int m0, int m_end, int n0, int ldw, int tiles_per_seg, int lane, int wave, int trace_slot) {
  constexpr int BM = 32 * MI;
  // the LDS stage is named here, not passed in: a pointer handed through the caller's lambda becomes a generic pointer and the
  // generic -> LDS cast in front of every DMA trips the backend ("Illegal instruction ... src_shared_base")
  extern __shared__ __attribute__((aligned(16))) char smem_dyn[];
  char *smem = smem_dyn;
  GEMM_TR_DECL;
  GEMM_TR(0);
  // a fresh (opaque) lane id per chunk: otherwise the address arithmetic of all four chunk heights is hoisted out of the persistent loop
  // and kept live across it (230 VGPRs spilled)
  asm volatile("" : "+v"(lane));
  const int wm = wave >> 1, wn = wave & 1;
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
  const bool resid_first = MODE == GEMM_OUT_F32 && g.resid != nullptr;
  floatx4 acc[MI][4];
  if (resid_first) gemm_acc_from_resid<MI>(g, acc, m0, n0, wm, wn, fr, fq);
  else {
#pragma unroll
    for (int i = 0; i < MI; i++)
#pragma unroll
      for (int j = 0; j < 4; j++) acc[i][j] = (floatx4){0.f, 0.f, 0.f, 0.f};
  }
  char *sa = smem, *sb = smem + BM * 128;
  const bool natural = (MODE == GEMM_OUT_QKV) && (((n0 + wn * 64) % 192) >= 128);
  GEMM_TR(1);
  auto kloop = [&](auto nat) {
    constexpr bool NAT = decltype(nat)::value;
    for (int seg = 0; seg < g.nseg; seg++) {
      const __half *aseg = g.A[seg] + (ptrdiff_t)g.row_off[seg] * g.lda;
      const __half *wseg = g.W + (g.custom_w ? g.w_off_[seg] : seg * g.kseg);
      for (int kt = 0; kt < tiles_per_seg; kt++) {
        const __half *abase = aseg + (kt << 6), *wbase = wseg + (kt << 6);
#pragma unroll
        for (int i = 0; i < MI; i++)
          __builtin_amdgcn_global_load_lds((gptr_t)(abase + aoff[i]), (lptr_t)(sa + (wave * MI + i) * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; i++)
          __builtin_amdgcn_global_load_lds((gptr_t)(wbase + boff[i]), (lptr_t)(sb + (wave * 4 + i) * 1024), 16, 0, 0);
        __syncthreads(); // waits vmcnt(0) for the DMA (and for the previous chunk's stores), then barrier
#ifdef TTS_GEMM_TRACE
        if (seg == 0 && kt == 0) GEMM_TR(2);
#endif
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
        __syncthreads();
      }
    }
  };
  if (MODE == GEMM_OUT_QKV && natural) kloop(std::true_type{});
  else kloop(std::false_type{});
  GEMM_TR(3);
  if (resid_first) gemm_epilogue<MODE, MI, true>(g, acc, m0, n0, wm, wn, fr, fq, m_end);
  else gemm_epilogue<MODE, MI>(g, acc, m0, n0, wm, wn, fr, fq, m_end);
  GEMM_TR(4);
  GEMM_TR_FLUSH(trace_slot);
  (void)trace_slot;
}

// row range of this workgroup in 32-row units, and its n-tile inside an n-chunk (see the comment block above)
struct BalWork { int u0, u1, nn, nchunks; };
__device__ __forceinline__ BalWork bal_work(const GemmArgs &g) {
  const int U = (g.M + 31) >> 5, NT = g.N >> 7;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, S = gridDim.x >> 3;
  const int uq = U >> 3, ur = U & 7;
  const int ucount = uq + (xcd < ur ? 1 : 0), ufirst = xcd * uq + (xcd < ur ? xcd : ur);
  const int cn = g.cn, G = S / cn, gi = slot / cn;
  BalWork w;
  w.nn = slot - gi * cn;
  w.u0 = ufirst + (gi * ucount) / G;
  w.u1 = ufirst + ((gi + 1) * ucount) / G;
  w.nchunks = NT / cn;
  return w;
}

template <int MODE, int WGS>
static __global__ __launch_bounds__(256, WGS) void gemm_f16_bal_kernel(GemmArgs g) { // 32 KB of dynamic LDS
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const BalWork w = bal_work(g);
  if (w.u0 >= w.u1) return;
  const int tiles_per_seg = g.kseg >> 6;
  const int ldw = g.custom_w ? g.ldw_ : g.nseg * g.kseg;
  const bool small_first = ((blockIdx.x >> 3) / g.cn) & 1;
  int tslot = blockIdx.x; // trace build: one record per chunk
  auto run = [&](int k, int u, int n0) {
    const int m0 = u << 5, m_end = min((u + k) << 5, g.M);
    tslot += gridDim.x;
    if (k == 4) glds_chunk<MODE, 4>(g, m0, m_end, n0, ldw, tiles_per_seg, lane, wave, tslot);
    else if (k == 3) glds_chunk<MODE, 3>(g, m0, m_end, n0, ldw, tiles_per_seg, lane, wave, tslot);
    else if (k == 2) glds_chunk<MODE, 2>(g, m0, m_end, n0, ldw, tiles_per_seg, lane, wave, tslot);
    else glds_chunk<MODE, 1>(g, m0, m_end, n0, ldw, tiles_per_seg, lane, wave, tslot);
  };
  for (int c = 0; c < w.nchunks; c++) {
    const int n0 = (c * g.cn + w.nn) << 7;
    int u = w.u0;
    const int rem = (w.u1 - w.u0) & 3;
    if (small_first && rem) { run(rem, u, n0); u += rem; }
    while (w.u1 - u >= 4) { run(4, u, n0); u += 4; }
    if (u < w.u1) run(w.u1 - u, u, n0);
  }
}

template <int MODE, int MI>
__device__ __forceinline__ void conv3_chunk(const GemmArgs &g, int m0, int m_end, int n0, int lane, int wave, int trace_slot) {
  constexpr int BM = 32 * MI, SLAB = (BM + 8) * 128;
  extern __shared__ __attribute__((aligned(16))) char smem_dyn[]; // see glds_chunk
  char *smem = smem_dyn;
  GEMM_TR_DECL;
  GEMM_TR(0);
  asm volatile("" : "+v"(lane)); // see glds_chunk
  const int wm = wave >> 1, wn = wave & 1;
  const int nchunks = g.kseg >> 6, ldw = 3 * g.kseg, nph = 3 * nchunks;
  const int prow = lane >> 3, pslot = lane & 7;
  const int fr = lane & 15, fq = lane >> 4;
  floatx4 acc[MI][4];
#pragma unroll
  for (int i = 0; i < MI; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = (floatx4){0.f, 0.f, 0.f, 0.f};
  char *sa = smem, *sb = smem + SLAB;
  const __half *abase = g.A[0] + (ptrdiff_t)(m0 - 1) * g.lda;
  const __half *wbase = g.W + (size_t)n0 * ldw;
  int aoff[MI + 1], boff[4];
#pragma unroll
  for (int i = 0; i <= MI; i++) {
    const int row = (i < MI ? wave * MI + i : 4 * MI) * 8 + prow;
    const int c = pslot ^ ((row >> 1) & 7);
    aoff[i] = min(row, g.M - m0 + 1) * g.lda + c * 8;
  }
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int row = (wave * 4 + i) * 8 + prow;
    const int c = pslot ^ ((row >> 1) & 7);
    boff[i] = row * ldw + c * 8;
  }
  auto stageA = [&](int kc) {
    const __half *src = abase + (min(kc, nchunks - 1) << 6);
#pragma unroll
    for (int i = 0; i < MI; i++) __builtin_amdgcn_global_load_lds((gptr_t)(src + aoff[i]), (lptr_t)(sa + (wave * MI + i) * 1024), 16, 0, 0);
    if (wave == 0) __builtin_amdgcn_global_load_lds((gptr_t)(src + aoff[MI]), (lptr_t)(sa + 4 * MI * 1024), 16, 0, 0);
  };
  auto stageB = [&](int p) {
    p = min(p, nph - 1);
    const int kc = p / 3, tap = p - kc * 3;
    const __half *src = wbase + tap * g.kseg + (kc << 6);
    char *dst = sb + (p & 1) * 16384;
#pragma unroll
    for (int i = 0; i < 4; i++) __builtin_amdgcn_global_load_lds((gptr_t)(src + boff[i]), (lptr_t)(dst + (wave * 4 + i) * 1024), 16, 0, 0);
  };
  // the previous chunk's epilogue stores are still in the in-order vmcnt queue: retire them here, or the counted waits below would
  // count them instead of this chunk's DMA pieces
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  stageA(0);
  stageB(0);
  GEMM_TR(1);
  for (int kc = 0, p = 0; kc < nchunks; kc++) {
#pragma unroll
    for (int tap = 0; tap < 3; tap++, p++) {
      stageB(p + 1);
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      __builtin_amdgcn_s_barrier();
#ifdef TTS_GEMM_TRACE
      if (p == 0) GEMM_TR(2);
#endif
      const char *sbp = sb + (p & 1) * 16384;
#pragma unroll
      for (int ks = 0; ks < 2; ks++) {
        half8 af[MI], bf[4];
#pragma unroll
        for (int i = 0; i < MI; i++) af[i] = *(const half8 *)(sa + lds_off(wm * (16 * MI) + i * 16 + fr + tap, ks * 4 + fq));
#pragma unroll
        for (int i = 0; i < 4; i++) bf[i] = *(const half8 *)(sbp + lds_off(wn * 64 + i * 16 + fr, ks * 4 + fq));
#pragma unroll
        for (int i = 0; i < MI; i++)
#pragma unroll
          for (int j = 0; j < 4; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[j], af[i], acc[i][j], 0, 0, 0);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (tap == 2) stageA(kc + 1);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // trailing (clamped) pieces must land before the LDS is reused / released
  GEMM_TR(3);
  gemm_epilogue<MODE, MI>(g, acc, m0, n0, wm, wn, fr, fq, m_end);
  GEMM_TR(4);
  GEMM_TR_FLUSH(trace_slot);
  (void)trace_slot;
}

template <int MODE>
static __global__ __launch_bounds__(256, 3) void gemm_f16_conv3_bal_kernel(GemmArgs g) { // conv3_lds<4>() of dynamic LDS
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const BalWork w = bal_work(g);
  if (w.u0 >= w.u1) return;
  const bool small_first = ((blockIdx.x >> 3) / g.cn) & 1;
  int tslot = blockIdx.x; // trace build: one record per chunk
  auto run = [&](int k, int u, int n0) {
    const int m0 = u << 5, m_end = min((u + k) << 5, g.M);
    tslot += gridDim.x;
    if (k == 4) conv3_chunk<MODE, 4>(g, m0, m_end, n0, lane, wave, tslot);
    else if (k == 3) conv3_chunk<MODE, 3>(g, m0, m_end, n0, lane, wave, tslot);
    else if (k == 2) conv3_chunk<MODE, 2>(g, m0, m_end, n0, lane, wave, tslot);
    else conv3_chunk<MODE, 1>(g, m0, m_end, n0, lane, wave, tslot);
  };
  for (int c = 0; c < w.nchunks; c++) {
    const int n0 = (c * g.cn + w.nn) << 7;
    int u = w.u0;
    const int rem = (w.u1 - w.u0) & 3;
    if (small_first && rem) { run(rem, u, n0); u += rem; }
    while (w.u1 - u >= 4) { run(4, u, n0); u += 4; }
    if (u < w.u1) run(w.u1 - u, u, n0);
  }
}


// A/B switch (TTS_GEMM_BAL=1 enables; tools/gemm_diag.hip toggles it): balanced persistent kernels (gemm_f16_bal_kernel,
// gemm_f16_conv3_bal_kernel) where the problem is large enough to give every workgroup rows. OFF by default: measured equal to 11 %
// slower than one tile per workgroup (profiles/r2_gemm_tile_phases.txt) — a chunk's K loop costs the same whether it is 32 or 128
// rows high (a workgroup's K iteration is bound by the CU's fixed per-iteration work: two barriers, the B tile's DMA and fragment
// reads), so shorter chunks do not buy time and equal ROW counts are not equal TIME.
static inline int &gemm_balanced_flag() {
  static int v = 0;
  return v;
}

// returns true when the balanced path took the launch
static inline bool launch_gemm_balanced(const GemmArgs &g, GemmArgs &gg, bool conv3, bool force_mi, int NTt, int cus_per_xcd, hipStream_t s, hipError_t *err) {
  if (gemm_balanced_flag() && !force_mi) {
    // S workgroups per XCD = G row groups x cn n-tiles; cn: the largest divisor of NT that divides S and keeps the n-chunk's weights
    // within ~2.5 MB of L2 (narrow outputs, NT <= 8, are not chunked: see the measurement above)
    const int wgs = conv3 ? 3 : 4, S = cus_per_xcd * wgs, ktot = g.nseg * g.kseg;
    int cnb = 0;
    for (int c = NTt; c >= 1; c--)
      if (NTt % c == 0 && S % c == 0 && (NTt <= 8 || (size_t)c * 128 * ktot * 2 <= (size_t)2560 * 1024)) { cnb = c; break; }
    const int U = (g.M + 31) >> 5;
    // (the fp16-output k = 3 convolution — inp_block, K = 3 x 128 — stays on gemm_f16_conv3_kernel: it is a few microseconds, and hipcc
    //  7.2 fails to compile its balanced instantiation: "Illegal instruction ... V_CMP_NE_U32 0, $src_shared_base")
    if (cnb > 0 && (U >> 3) >= 2 * (S / cnb) && !(conv3 && g.mode == GEMM_OUT_F16)) { // every row group gets at least two 32-row units
      gg.cn = cnb;
      if (conv3) {
        static bool attr = false;
        if (!attr) {
          (void)hipFuncSetAttribute((const void *)gemm_f16_conv3_bal_kernel<GEMM_OUT_F32>, hipFuncAttributeMaxDynamicSharedMemorySize, conv3_lds<4>());
          attr = true;
        }
        gemm_f16_conv3_bal_kernel<GEMM_OUT_F32><<<8 * S, 256, conv3_lds<4>(), s>>>(gg);
      } else if (g.mode == GEMM_OUT_F32) gemm_f16_bal_kernel<GEMM_OUT_F32, 4><<<8 * S, 256, 32768, s>>>(gg);
      else if (g.mode == GEMM_OUT_F16) gemm_f16_bal_kernel<GEMM_OUT_F16, 4><<<8 * S, 256, 32768, s>>>(gg);
      else gemm_f16_bal_kernel<GEMM_OUT_QKV, 4><<<8 * S, 256, 32768, s>>>(gg);
      *err = hipGetLastError();
      return true;
    }
  }
  return false;
}

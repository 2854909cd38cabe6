// Measured-and-rejected GEMM variants ("tried and measured slower"). NOT part of the
// product: only tools/gemm_bench.hip includes this file (define TTS_GEMM_VARIANT = 0, 2, 3, 4, 5 or 6 before
// including gemm_f16.h). Kept so that the negative results stay reproducible.
//   0: register-staged 2 x (A 16 KB + B 16 KB) double buffer            200 TF/s
//   2/3: 128^2 tile, 2-/3-deep LDS-DMA ring, counted vmcnt               slower than the single-stage kernel
//   4: 256 x 128 tile, 8 waves, 3-deep ring                               870 TF/s at 8192^3, slower on M = 28288
//   5: 256 x 256 tile, 8 waves, 2 stages, one barrier per K tile          1002 TF/s at 8192^3, 724-746 on M = 28288
//   6: 256 x 256 tile, 8-phase schedule with two staggered wave groups    1064 TF/s at 8192^3, 754-770 on M = 28288
#pragma once
// (included by gemm_f16.h from inside namespace tts, after GemmArgs / lds_off / gemm_epilogue)

static __global__ __launch_bounds__(256) void gemm_f16_kernel(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[]; // 2 x (A 16 KB + B 16 KB)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  // XCD-aware tile order: blocks b, b+8, b+16 ... (same XCD) walk consecutive n-tiles of one m-tile
  const int ntn = g.N >> 7, ntiles = (g.M >> 7) * ntn;
  int bid = blockIdx.x;
  {
    const int q = ntiles >> 3, r = ntiles & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int m0 = (bid / ntn) << 7, n0 = (bid % ntn) << 7;
  const int tiles_per_seg = g.kseg >> 6, nk = g.nseg * tiles_per_seg;
  const int ldw = g.custom_w ? g.ldw_ : g.nseg * g.kseg;

  const int lc = tid & 7, lr = tid >> 3; // staging: chunk lc of rows lr, lr+32, lr+64, lr+96
  uint4 ra[4], rb[4];
  auto gload = [&](int kt) {
    const int seg = kt / tiles_per_seg, kk = (kt - seg * tiles_per_seg) << 6;
    const __half *ap = g.A[seg] + (size_t)(m0 + g.row_off[seg] + lr) * g.lda + kk + lc * 8;
    const __half *wp = g.W + (size_t)(n0 + lr) * ldw + (g.custom_w ? g.w_off_[seg] : seg * g.kseg) + kk + lc * 8;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      ra[i] = *(const uint4 *)(ap + (size_t)i * 32 * g.lda);
      rb[i] = *(const uint4 *)(wp + (size_t)i * 32 * ldw);
    }
  };
  auto lstore = [&](int buf) {
    char *sa = smem + buf * 32768, *sb = sa + 16384;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      *(uint4 *)(sa + lds_off(lr + 32 * i, lc)) = ra[i];
      *(uint4 *)(sb + lds_off(lr + 32 * i, lc)) = rb[i];
    }
  };

  floatx4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = (floatx4){0.f, 0.f, 0.f, 0.f};

  gload(0);
  lstore(0);
  __syncthreads();
  const int fr = lane & 15, fq = lane >> 4;
  for (int kt = 0; kt < nk; kt++) {
    const int buf = kt & 1;
    if (kt + 1 < nk) gload(kt + 1);
    const char *sa = smem + buf * 32768, *sb = sa + 16384;
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {
      half8 af[4], bf[4];
#pragma unroll
      for (int i = 0; i < 4; i++) {
        af[i] = *(const half8 *)(sa + lds_off(wm * 64 + i * 16 + fr, ks * 4 + fq));
        bf[i] = *(const half8 *)(sb + lds_off(wn * 64 + i * 16 + fr, ks * 4 + fq));
      }
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) lstore(buf ^ 1);
    __syncthreads();
  }

  // epilogue: acc[i][j][r] = C[m0 + wm*64 + i*16 + fq*4 + r][n0 + wn*64 + j*16 + fr]
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int rbase = m0 + wm * 64 + i * 16 + fq * 4;
    bool guard[4];
#pragma unroll
    for (int r = 0; r < 4; r++) guard[r] = g.row_seq ? (g.row_seq[rbase + r] < 0) : false;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int col = n0 + wn * 64 + j * 16 + fr;
      const float bv = g.bias ? g.bias[col] : 0.f;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; r++) v[r] = acc[i][j][r] + bv;
      if (g.mode == GEMM_OUT_F32) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
          float o = v[r];
          if (g.resid) o += g.resid[(size_t)(rbase + r) * g.ldo + col];
          g.outF[(size_t)(rbase + r) * g.ldo + col] = guard[r] ? 0.f : o;
        }
      } else if (g.mode == GEMM_OUT_F16) {
#pragma unroll
        for (int r = 0; r < 4; r++)
          g.outH[(size_t)(rbase + r) * g.ldh + col] = __float2half_rn(guard[r] ? 0.f : v[r]);
      } else { // QKV: col = h*192 + {q 0..63 | k 64..127 | v 128..191}
        const int h = col / 192, w = col - h * 192;
        if (w < 128) {
#pragma unroll
          for (int r = 0; r < 4; r++)
            g.outH[(size_t)(rbase + r) * g.ldh + h * 128 + w] = __float2half_rn(guard[r] ? 0.f : v[r]);
        } else {
          __half2 p0 = __floats2half2_rn(guard[0] ? 0.f : v[0], guard[1] ? 0.f : v[1]);
          __half2 p1 = __floats2half2_rn(guard[2] ? 0.f : v[2], guard[3] ? 0.f : v[3]);
          uint2 u;
          u.x = *(unsigned *)&p0;
          u.y = *(unsigned *)&p1;
          *(uint2 *)(g.outVt + (size_t)(h * 64 + (w - 128)) * g.ldvt + rbase) = u;
        }
      }
    }
  }
}

// Variant 2/3: NST-deep LDS ring (NST x 32 KB, dynamic LDS). Tile kt+NST-1 is requested while tile kt is
// multiplied; DMA pieces stay in flight across the (raw) barrier and are retired with a counted vmcnt.
template <int MODE, int NST>
static __global__ __launch_bounds__(256) void gemm_f16_ring_kernel(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem_dyn[];
  char *smem = smem_dyn;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int ntn = g.N >> 7, ntiles = (g.M >> 7) * ntn;
  int bid = blockIdx.x;
  {
    const int q = ntiles >> 3, r = ntiles & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int m0 = (bid / ntn) << 7, n0 = (bid % ntn) << 7;
  const int tiles_per_seg = g.kseg >> 6, nk = g.nseg * tiles_per_seg;
  const int ldw = g.custom_w ? g.ldw_ : g.nseg * g.kseg;
  const int prow = lane >> 3, pslot = lane & 7;
  floatx4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = (floatx4){0.f, 0.f, 0.f, 0.f};
  const int fr = lane & 15, fq = lane >> 4;
  const bool natural = (MODE == GEMM_OUT_QKV) && (((n0 + wn * 64) % 192) >= 128);
  auto stage = [&](int kt, int buf) {
    const int seg = kt / tiles_per_seg, kk = (kt - seg * tiles_per_seg) << 6;
    const __half *abase = g.A[seg] + (size_t)(m0 + g.row_off[seg]) * g.lda + kk;
    const __half *wbase = g.W + (size_t)n0 * ldw + (g.custom_w ? g.w_off_[seg] : seg * g.kseg) + kk;
    char *sa = smem + buf * 32768, *sb = sa + 16384;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int row = (wave * 4 + i) * 8 + prow;
      const int c = pslot ^ ((row >> 1) & 7);
      __builtin_amdgcn_global_load_lds((gptr_t)(abase + (size_t)row * g.lda + c * 8), (lptr_t)(sa + (wave * 4 + i) * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)(wbase + (size_t)row * ldw + c * 8), (lptr_t)(sb + (wave * 4 + i) * 1024), 16, 0, 0);
    }
  };
  auto kloop = [&](auto nat) {
    constexpr bool NAT = decltype(nat)::value;
#pragma unroll
    for (int p = 0; p < NST - 1; p++)
      if (p < nk) stage(p, p);
    for (int kt = 0; kt < nk; kt++) {
      // tiles kt+1 .. kt+NST-2 may remain in flight (8 DMA pieces each)
      const int ahead = min(NST - 2, nk - 1 - kt);
      if (NST == 3 && ahead == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (kt + NST - 1 < nk) stage(kt + NST - 1, (kt + NST - 1) % NST);
      const char *sa = smem + (kt % NST) * 32768, *sb = sa + 16384;
#pragma unroll
      for (int ks = 0; ks < 2; ks++) {
        half8 af[4], bf[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
          af[i] = *(const half8 *)(sa + lds_off(wm * 64 + i * 16 + fr, ks * 4 + fq));
          bf[i] = *(const half8 *)(sb + lds_off(wn * 64 + i * 16 + fr, ks * 4 + fq));
        }
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
          for (int j = 0; j < 4; j++) {
            if (NAT) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
            else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[j], af[i], acc[i][j], 0, 0, 0);
          }
      }
    }
  };
  if (MODE == GEMM_OUT_QKV && natural) kloop(std::true_type{});
  else kloop(std::false_type{});
#ifdef TTS_GEMM_DIAG_NOEPI
  if (g.ldo != -12345) return;
#endif
  gemm_epilogue<MODE, 4>(g, acc, m0, n0, wm, wn, fr, fq, g.M);
}

template <int NST>
static inline hipError_t launch_gemm_ring(const GemmArgs &g, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void *)gemm_f16_ring_kernel<GEMM_OUT_F32, NST>, hipFuncAttributeMaxDynamicSharedMemorySize, NST * 32768);
    (void)hipFuncSetAttribute((const void *)gemm_f16_ring_kernel<GEMM_OUT_F16, NST>, hipFuncAttributeMaxDynamicSharedMemorySize, NST * 32768);
    (void)hipFuncSetAttribute((const void *)gemm_f16_ring_kernel<GEMM_OUT_QKV, NST>, hipFuncAttributeMaxDynamicSharedMemorySize, NST * 32768);
    attr_set = true;
  }
  const int ntiles = (g.M >> 7) * (g.N >> 7);
  if (g.mode == GEMM_OUT_F32) gemm_f16_ring_kernel<GEMM_OUT_F32, NST><<<ntiles, 256, NST * 32768, s>>>(g);
  else if (g.mode == GEMM_OUT_F16) gemm_f16_ring_kernel<GEMM_OUT_F16, NST><<<ntiles, 256, NST * 32768, s>>>(g);
  else gemm_f16_ring_kernel<GEMM_OUT_QKV, NST><<<ntiles, 256, NST * 32768, s>>>(g);
  return hipGetLastError();
}

// Variant 4: 256(M) x 128(N) x 64 tile, 8 waves (4 x 2, each 64x64), 3-deep LDS ring (3 x 48 KB = 144 KB,
// one workgroup per CU, two waves per SIMD). Tile kt+2 is requested while tile kt is multiplied; the DMA
// pieces stay in flight across the single raw barrier per K tile (counted vmcnt).
template <int MODE>
static __global__ __launch_bounds__(512) void gemm_f16_big_kernel(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem_dyn[];
  char *smem = smem_dyn;
  constexpr int STAGE = 49152, NST = 3;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int ntn = g.N >> 7, ntiles = ((g.M + 255) >> 8) * ntn;
  int bid = blockIdx.x;
  {
    const int q = ntiles >> 3, r = ntiles & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int m0 = (bid / ntn) << 8, n0 = (bid % ntn) << 7;
  const int tiles_per_seg = g.kseg >> 6, nk = g.nseg * tiles_per_seg;
  const int ldw = g.custom_w ? g.ldw_ : g.nseg * g.kseg;
  const int prow = lane >> 3, pslot = lane & 7;
  floatx4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = (floatx4){0.f, 0.f, 0.f, 0.f};
  const int fr = lane & 15, fq = lane >> 4;
  const bool natural = (MODE == GEMM_OUT_QKV) && (((n0 + wn * 64) % 192) >= 128);
  const int mlast = g.M - 1; // rows beyond M (M % 256 == 128) are clamped: their results are never stored
  auto stage = [&](int kt, int buf) {
    const int seg = kt / tiles_per_seg, kk = (kt - seg * tiles_per_seg) << 6;
    const __half *abase = g.A[seg] + (size_t)g.row_off[seg] * g.lda + kk;
    const __half *wbase = g.W + (size_t)n0 * ldw + (g.custom_w ? g.w_off_[seg] : seg * g.kseg) + kk;
    char *sa = smem + buf * STAGE, *sb = sa + 32768;
#pragma unroll
    for (int i = 0; i < 4; i++) { // A: 32 pieces of 8 rows
      const int row = (wave * 4 + i) * 8 + prow;
      const int c = pslot ^ ((row >> 1) & 7);
      const int grow = min(m0 + row, mlast);
      __builtin_amdgcn_global_load_lds((gptr_t)(abase + (size_t)grow * g.lda + c * 8), (lptr_t)(sa + (wave * 4 + i) * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 2; i++) { // B: 16 pieces
      const int row = (wave * 2 + i) * 8 + prow;
      const int c = pslot ^ ((row >> 1) & 7);
      __builtin_amdgcn_global_load_lds((gptr_t)(wbase + (size_t)row * ldw + c * 8), (lptr_t)(sb + (wave * 2 + i) * 1024), 16, 0, 0);
    }
  };
  auto kloop = [&](auto nat) {
    constexpr bool NAT = decltype(nat)::value;
    stage(0, 0);
    if (nk > 1) stage(1, 1);
    for (int kt = 0; kt < nk; kt++) {
      if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); // tile kt landed; tile kt+1 (6 pieces) may fly
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (kt + 2 < nk) stage(kt + 2, (kt + 2) % NST);
      const char *sa = smem + (kt % NST) * STAGE, *sb = sa + 32768;
#pragma unroll
      for (int ks = 0; ks < 2; ks++) {
        half8 af[4], bf[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
          af[i] = *(const half8 *)(sa + lds_off(wm * 64 + i * 16 + fr, ks * 4 + fq));
          bf[i] = *(const half8 *)(sb + lds_off(wn * 64 + i * 16 + fr, ks * 4 + fq));
        }
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
          for (int j = 0; j < 4; j++) {
            if (NAT) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
            else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[j], af[i], acc[i][j], 0, 0, 0);
          }
      }
    }
  };
  if (MODE == GEMM_OUT_QKV && natural) kloop(std::true_type{});
  else kloop(std::false_type{});
#ifdef TTS_GEMM_DIAG_NOEPI
  if (g.ldo != -12345) return;
#endif
  if (m0 + wm * 64 < g.M) gemm_epilogue<MODE, 4>(g, acc, m0, n0, wm, wn, fr, fq, g.M);
}

static inline hipError_t launch_gemm_big(const GemmArgs &g, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void *)gemm_f16_big_kernel<GEMM_OUT_F32>, hipFuncAttributeMaxDynamicSharedMemorySize, 147456);
    (void)hipFuncSetAttribute((const void *)gemm_f16_big_kernel<GEMM_OUT_F16>, hipFuncAttributeMaxDynamicSharedMemorySize, 147456);
    (void)hipFuncSetAttribute((const void *)gemm_f16_big_kernel<GEMM_OUT_QKV>, hipFuncAttributeMaxDynamicSharedMemorySize, 147456);
    attr_set = true;
  }
  const int ntiles = ((g.M + 255) >> 8) * (g.N >> 7);
  if (g.mode == GEMM_OUT_F32) gemm_f16_big_kernel<GEMM_OUT_F32><<<ntiles, 512, 147456, s>>>(g);
  else if (g.mode == GEMM_OUT_F16) gemm_f16_big_kernel<GEMM_OUT_F16><<<ntiles, 512, 147456, s>>>(g);
  else gemm_f16_big_kernel<GEMM_OUT_QKV><<<ntiles, 512, 147456, s>>>(g);
  return hipGetLastError();
}

// Variant 5: 256 x 256 x 64 tile (128 FLOP per operand byte: half the L2->LDS traffic of the 128^2 tile),
// 8 waves as 2(M) x 4(N), each 128 x 64 = 8 x 4 MFMA tiles; two 64 KB LDS stages (A 32 KB | B 32 KB), one raw
// barrier per K tile, tile kt+1 requested right after the barrier and retired (vmcnt 0) before the next one.
// Fragments are loaded per 64-row half of the wave's rows to stay inside 256 VGPRs.
template <int MODE>
static __global__ __launch_bounds__(512) void gemm_f16_256_kernel(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem_dyn[];
  char *smem = smem_dyn;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3;
  const int MT = (g.M + 255) >> 8, NT = g.N >> 8;
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int mq = MT >> 3, mr = MT & 7;
  const int mcount = mq + (xcd < mr ? 1 : 0), mfirst = xcd * mq + (xcd < mr ? xcd : mr);
  if (idx >= mcount * NT) return;
  const int m0 = (mfirst + idx / NT) << 8, n0 = (idx % NT) << 8;
  const int tiles_per_seg = g.kseg >> 6, nk = g.nseg * tiles_per_seg;
  const int ldw = g.custom_w ? g.ldw_ : g.nseg * g.kseg;
  const int prow = lane >> 3, pslot = lane & 7;
  const int fr = lane & 15, fq = lane >> 4;
  const int mlast = g.M - 1;
  floatx4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = (floatx4){0.f, 0.f, 0.f, 0.f};
  auto stage = [&](int kt, int buf) {
    const int seg = kt / tiles_per_seg, kk = (kt - seg * tiles_per_seg) << 6;
    const __half *abase = g.A[seg] + (size_t)g.row_off[seg] * g.lda + kk;
    const __half *wbase = g.W + (size_t)n0 * ldw + (g.custom_w ? g.w_off_[seg] : seg * g.kseg) + kk;
    char *sa = smem + buf * 65536, *sb = sa + 32768;
#pragma unroll
    for (int i = 0; i < 4; i++) { // 32 pieces of 8 rows for A and for B; wave w moves pieces 4w .. 4w+3
      const int row = (wave * 4 + i) * 8 + prow;
      const int c = pslot ^ ((row >> 1) & 7);
      const int grow = min(m0 + row, mlast);
      __builtin_amdgcn_global_load_lds((gptr_t)(abase + (size_t)grow * g.lda + c * 8), (lptr_t)(sa + (wave * 4 + i) * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)(wbase + (size_t)row * ldw + c * 8), (lptr_t)(sb + (wave * 4 + i) * 1024), 16, 0, 0);
    }
  };
  stage(0, 0);
  for (int kt = 0; kt < nk; kt++) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (kt + 1 < nk) stage(kt + 1, (kt + 1) & 1);
    const char *sa = smem + (kt & 1) * 65536, *sb = sa + 32768;
    half8 bf[4][2];
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
      for (int ks = 0; ks < 2; ks++) bf[j][ks] = *(const half8 *)(sb + lds_off(wn * 64 + j * 16 + fr, ks * 4 + fq));
#pragma unroll
    for (int mh = 0; mh < 2; mh++) {
      half8 af[4][2];
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int ks = 0; ks < 2; ks++) af[i][ks] = *(const half8 *)(sa + lds_off(wm * 128 + mh * 64 + i * 16 + fr, ks * 4 + fq));
#pragma unroll
      for (int ks = 0; ks < 2; ks++)
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
          for (int j = 0; j < 4; j++)
            acc[mh * 4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[j][ks], af[i][ks], acc[mh * 4 + i][j], 0, 0, 0);
    }
  }
  // epilogue (F32 / F16 outputs; swapped operand order: lane = 4 consecutive columns of one row)
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const int row = m0 + wm * 128 + i * 16 + fr;
    if (row >= g.M) continue;
    const bool guard = g.row_seq ? (g.row_seq[row] < 0) : false;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int col = n0 + wn * 64 + j * 16 + fq * 4;
      float4 v = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
      if (g.bias) {
        const float4 b = *(const float4 *)(g.bias + col);
        v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
      }
      if (MODE == GEMM_OUT_F32) {
        if (g.resid) {
          const float4 rr = *(const float4 *)(g.resid + (size_t)row * g.ldo + col);
          v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
        }
        if (guard) v = make_float4(0.f, 0.f, 0.f, 0.f);
        *(float4 *)(g.outF + (size_t)row * g.ldo + col) = v;
      } else {
        if (guard) v = make_float4(0.f, 0.f, 0.f, 0.f);
        __half2 p0 = __floats2half2_rn(v.x, v.y), p1 = __floats2half2_rn(v.z, v.w);
        uint2 u;
        u.x = *(unsigned *)&p0;
        u.y = *(unsigned *)&p1;
        *(uint2 *)(g.outH + (size_t)row * g.ldh + col) = u;
      }
    }
  }
}

static inline hipError_t launch_gemm_256(const GemmArgs &g, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void *)gemm_f16_256_kernel<GEMM_OUT_F32>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    (void)hipFuncSetAttribute((const void *)gemm_f16_256_kernel<GEMM_OUT_F16>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    attr_set = true;
  }
  const int MT = (g.M + 255) >> 8, NT = g.N >> 8;
  const int grid = 8 * ((MT >> 3) + ((MT & 7) ? 1 : 0)) * NT;
  if (g.mode == GEMM_OUT_F32) gemm_f16_256_kernel<GEMM_OUT_F32><<<grid, 512, 131072, s>>>(g);
  else gemm_f16_256_kernel<GEMM_OUT_F16><<<grid, 512, 131072, s>>>(g);
  return hipGetLastError();
}

// Variant 6: 256 x 256 x 64 tile, 8 waves (2 M x 4 N, 128 x 64 per wave = 8 x 4 MFMA tiles), two 64 KB LDS
// buffers, each split into four 16 KB half-tiles that are staged (2 LDS-DMA pieces per wave) and retired
// independently. A K tile is four phases; a phase is
//     ds_read subtile | stage one half-tile | [counted vmcnt] | barrier | 16 MFMA (one C quadrant x K=64) | barrier
// and the two wave groups (wr = 0 / 1) run one barrier apart, so one group's MFMAs overlap the other's LDS
// reads and DMA issue. Half-tiles are interleaved so that each is needed by ALL waves in ONE phase:
//     A-h{h}: tile rows h*64+[0,64) and 128+h*64+[0,64)   (the mh = h rows of both wave groups)
//     B-h{h}: tile n-rows wc*64+h*32+[0,32), wc = 0..3    (the nh = h columns of all four wave columns)
//   phase 0: read B-h0 (4) then A-h0 (8), quadrant (0,0); stage A-h1 of tile t+1
//   phase 1: read B-h1 (4),               quadrant (0,1); stage B-h0 of tile t+2
//   phase 2: read A-h1 (8),               quadrant (1,1); stage A-h0 of tile t+2
//   phase 3: -                            quadrant (1,0); stage B-h1 of tile t+2; s_waitcnt vmcnt(6)
// vmcnt(6) leaves the three newest half-tiles in flight and retires everything staged up to phase 0 of this
// tile, i.e. all of tile t+1, which is first read one phase (two barriers) later. A half-tile is restaged
// two phases after its last ds_read (one phase for B-h0, whose reads are retired by lgkmcnt(8) before the
// first barrier of phase 0).
template <int MODE>
static __global__ __launch_bounds__(512) void gemm_f16_8ph_kernel(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem_dyn[];
  char *smem = smem_dyn;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 2, wc = wave & 3;
  const int fr = lane & 15, fq = lane >> 4;
  const int MT = (g.M + 255) >> 8, NT = g.N >> 8;
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int mq = MT >> 3, mr = MT & 7;
  const int mcount = mq + (xcd < mr ? 1 : 0), mfirst = xcd * mq + (xcd < mr ? xcd : mr);
  if (idx >= mcount * NT) return;
  const int m0 = (mfirst + idx / NT) << 8, n0 = (idx % NT) << 8;
  const int tiles_per_seg = g.kseg >> 6, nk = g.nseg * tiles_per_seg;
  const int ldw = g.custom_w ? g.ldw_ : g.nseg * g.kseg;
  // DMA roles: wave w moves pieces 2w, 2w+1 (8 local rows x 128 B each) of every half-tile
  int aoff[2][2], boff[2][2];
#pragma unroll
  for (int h = 0; h < 2; h++)
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const int lr = (2 * wave + i) * 8 + (lane >> 3);
      const int c = (lane & 7) ^ ((lr >> 1) & 7);
      const int arow = lr < 64 ? h * 64 + lr : 128 + h * 64 + (lr - 64);
      aoff[h][i] = min(m0 + arow, g.M - 1) * g.lda + c * 8;
      const int brow = (lr >> 5) * 64 + h * 32 + (lr & 31);
      boff[h][i] = (n0 + brow) * ldw + c * 8;
    }
  auto stageA = [&](int t, int h, int buf) {
    t = min(t, nk - 1); // past the end: harmless re-stage of the last tile keeps the vmcnt arithmetic uniform
    const int seg = t / tiles_per_seg, kk = (t - seg * tiles_per_seg) << 6;
    const __half *base = g.A[seg] + (ptrdiff_t)g.row_off[seg] * g.lda + kk;
    char *dst = smem + buf * 65536 + h * 16384 + (2 * wave) * 1024;
    __builtin_amdgcn_global_load_lds((gptr_t)(base + aoff[h][0]), (lptr_t)dst, 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gptr_t)(base + aoff[h][1]), (lptr_t)(dst + 1024), 16, 0, 0);
  };
  auto stageB = [&](int t, int h, int buf) {
    t = min(t, nk - 1);
    const int seg = t / tiles_per_seg, kk = (t - seg * tiles_per_seg) << 6;
    const __half *base = g.W + (g.custom_w ? g.w_off_[seg] : seg * g.kseg) + kk;
    char *dst = smem + buf * 65536 + 32768 + h * 16384 + (2 * wave) * 1024;
    __builtin_amdgcn_global_load_lds((gptr_t)(base + boff[h][0]), (lptr_t)dst, 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gptr_t)(base + boff[h][1]), (lptr_t)(dst + 1024), 16, 0, 0);
  };
  floatx4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = (floatx4){0.f, 0.f, 0.f, 0.f};
  // fragment read offsets inside a half-tile
  int ard[4][2], brd[2][2];
#pragma unroll
  for (int ks = 0; ks < 2; ks++) {
#pragma unroll
    for (int i = 0; i < 4; i++) ard[i][ks] = lds_off(wr * 64 + i * 16 + fr, ks * 4 + fq);
#pragma unroll
    for (int j = 0; j < 2; j++) brd[j][ks] = lds_off(wc * 32 + j * 16 + fr, ks * 4 + fq);
  }
  const bool natural = (MODE == GEMM_OUT_QKV) && (((n0 + wc * 64) % 192) >= 128);
  // prologue: all of tile 0, then B-h0, A-h0, B-h1 of tile 1 (the slots phases 1-3 of "tile -1" would have filled)
  stageA(0, 0, 0); stageB(0, 0, 0); stageB(0, 1, 0); stageA(0, 1, 0);
  stageB(1, 0, 1); stageA(1, 0, 1); stageB(1, 1, 1);
  asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier(); // stagger: group 1 runs one barrier behind group 0
  auto kloop = [&](auto nat) {
    constexpr bool NAT = decltype(nat)::value;
    half8 af[2][4][2], bf[2][2][2];
#define MMA_QUAD(MH, NH)                                                                                         \
  __builtin_amdgcn_s_barrier();                                                                                  \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                             \
  __builtin_amdgcn_sched_barrier(0);                                                                             \
  __builtin_amdgcn_s_setprio(1);                                                                                 \
  _Pragma("unroll") for (int ks = 0; ks < 2; ks++)                                                               \
  _Pragma("unroll") for (int i = 0; i < 4; i++)                                                                  \
  _Pragma("unroll") for (int j = 0; j < 2; j++) {                                                                \
    if (NAT) acc[MH * 4 + i][NH * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[MH][i][ks], bf[NH][j][ks], acc[MH * 4 + i][NH * 2 + j], 0, 0, 0); \
    else acc[MH * 4 + i][NH * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[NH][j][ks], af[MH][i][ks], acc[MH * 4 + i][NH * 2 + j], 0, 0, 0);     \
  }                                                                                                              \
  __builtin_amdgcn_s_setprio(0);                                                                                 \
  __builtin_amdgcn_sched_barrier(0);                                                                             \
  __builtin_amdgcn_s_barrier();                                                                                  \
  asm volatile("" ::: "memory");
    auto ktile = [&](int t, auto bufc) {
      constexpr int BUF = decltype(bufc)::value;
      const char *bA0 = smem + BUF * 65536, *bA1 = bA0 + 16384, *bB0 = bA0 + 32768, *bB1 = bA0 + 49152;
      // phase 0
#pragma unroll
      for (int ks = 0; ks < 2; ks++)
#pragma unroll
        for (int j = 0; j < 2; j++) bf[0][j][ks] = *(const half8 *)(bB0 + brd[j][ks]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 0; ks < 2; ks++)
#pragma unroll
        for (int i = 0; i < 4; i++) af[0][i][ks] = *(const half8 *)(bA0 + ard[i][ks]);
      stageA(t + 1, 1, BUF ^ 1);
      asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory"); // the 4 B reads (issued first) are retired: B-h0 may be restaged next phase
      MMA_QUAD(0, 0)
      // phase 1
#pragma unroll
      for (int ks = 0; ks < 2; ks++)
#pragma unroll
        for (int j = 0; j < 2; j++) bf[1][j][ks] = *(const half8 *)(bB1 + brd[j][ks]);
      stageB(t + 2, 0, BUF);
      MMA_QUAD(0, 1)
      // phase 2
#pragma unroll
      for (int ks = 0; ks < 2; ks++)
#pragma unroll
        for (int i = 0; i < 4; i++) af[1][i][ks] = *(const half8 *)(bA1 + ard[i][ks]);
      stageA(t + 2, 0, BUF);
      MMA_QUAD(1, 1)
      // phase 3
      stageB(t + 2, 1, BUF);
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); // all of tile t+1 has landed (this wave's pieces)
      MMA_QUAD(1, 0)
    };
    for (int t = 0; t < nk; t += 2) {
      ktile(t, std::integral_constant<int, 0>{});
      if (t + 1 < nk) ktile(t + 1, std::integral_constant<int, 1>{});
    }
#undef MMA_QUAD
  };
  if (MODE == GEMM_OUT_QKV && natural) kloop(std::true_type{});
  else kloop(std::false_type{});
  if (wr == 0) __builtin_amdgcn_s_barrier(); // re-align the two groups
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // trailing (dummy) DMA pieces must land before the LDS is released
  // epilogue: acc[i][j] = C[m0 + wr*128 + i*16 + ..][n0 + wc*64 + j*16 + ..]
  if (MODE == GEMM_OUT_QKV && natural) {
    const int c0 = n0 + wc * 64, h = c0 / 192; // V columns: lane = 4 consecutive rows of one column
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const int rbase = m0 + wr * 128 + i * 16 + fq * 4;
      if (rbase >= g.M) continue;
      bool guard[4];
#pragma unroll
      for (int r = 0; r < 4; r++) guard[r] = g.row_seq ? (g.row_seq[rbase + r] < 0) : false;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int d = j * 16 + fr;
        const float bv = g.bias ? g.bias[c0 + d] : 0.f;
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; r++) v[r] = guard[r] ? 0.f : acc[i][j][r] + bv;
        __half2 p0 = __floats2half2_rn(v[0], v[1]), p1 = __floats2half2_rn(v[2], v[3]);
        uint2 u;
        u.x = *(unsigned *)&p0;
        u.y = *(unsigned *)&p1;
        *(uint2 *)(g.outVt + (size_t)(h * 64 + d) * g.ldvt + rbase) = u;
      }
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const int row = m0 + wr * 128 + i * 16 + fr;
    if (row >= g.M) continue;
    const bool guard = g.row_seq ? (g.row_seq[row] < 0) : false;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int col = n0 + wc * 64 + j * 16 + fq * 4;
      float4 v = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
      if (g.bias) {
        const float4 b = *(const float4 *)(g.bias + col);
        v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
      }
      if (MODE == GEMM_OUT_F32) {
        if (g.resid) {
          const float4 rr = *(const float4 *)(g.resid + (size_t)row * g.ldo + col);
          v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
        }
        if (guard) v = make_float4(0.f, 0.f, 0.f, 0.f);
        *(float4 *)(g.outF + (size_t)row * g.ldo + col) = v;
      } else {
        if (guard) v = make_float4(0.f, 0.f, 0.f, 0.f);
        __half2 p0 = __floats2half2_rn(v.x, v.y), p1 = __floats2half2_rn(v.z, v.w);
        uint2 u;
        u.x = *(unsigned *)&p0;
        u.y = *(unsigned *)&p1;
        if (MODE == GEMM_OUT_QKV) { // Q or K columns
          const int hh = col / 192, w0 = col - hh * 192;
          *(uint2 *)(g.outH + (size_t)row * g.ldh + hh * 128 + w0) = u;
        } else {
          *(uint2 *)(g.outH + (size_t)row * g.ldh + col) = u;
        }
      }
    }
  }
}

static inline hipError_t launch_gemm_8ph(const GemmArgs &g, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void *)gemm_f16_8ph_kernel<GEMM_OUT_F32>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    (void)hipFuncSetAttribute((const void *)gemm_f16_8ph_kernel<GEMM_OUT_F16>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    (void)hipFuncSetAttribute((const void *)gemm_f16_8ph_kernel<GEMM_OUT_QKV>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    attr_set = true;
  }
  const int MT = (g.M + 255) >> 8, NT = g.N >> 8;
  const int grid = 8 * ((MT >> 3) + ((MT & 7) ? 1 : 0)) * NT;
  if (g.mode == GEMM_OUT_F32) gemm_f16_8ph_kernel<GEMM_OUT_F32><<<grid, 512, 131072, s>>>(g);
  else if (g.mode == GEMM_OUT_F16) gemm_f16_8ph_kernel<GEMM_OUT_F16><<<grid, 512, 131072, s>>>(g);
  else gemm_f16_8ph_kernel<GEMM_OUT_QKV><<<grid, 512, 131072, s>>>(g);
  return hipGetLastError();
}

static inline hipError_t launch_gemm_experiment(const GemmArgs &g, hipStream_t s) {
#if TTS_GEMM_VARIANT == 6
  return launch_gemm_8ph(g, s);
#elif TTS_GEMM_VARIANT == 5
  return launch_gemm_256(g, s);
#elif TTS_GEMM_VARIANT == 4
  return launch_gemm_big(g, s);
#elif TTS_GEMM_VARIANT == 2
  return launch_gemm_ring<2>(g, s);
#elif TTS_GEMM_VARIANT == 3
  return launch_gemm_ring<3>(g, s);
#else
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void *)gemm_f16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  gemm_f16_kernel<<<(g.M >> 7) * (g.N >> 7), 256, 65536, s>>>(g);
  return hipGetLastError();
#endif
}


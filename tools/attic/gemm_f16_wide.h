// Developer experiment (round 3): the single-stage K loop of csrc/gemm_f16.h on a 128 x 256 tile with 8 waves (2 x 4, each 64 x 64 as in the
// product kernel), 48 KB of LDS, 2 workgroups per CU. Same waves per CU, same work per wave, but 48 KB of operand DMA per 2 x (128 x 128 x 64)
// instead of 64 KB. Why: ablation builds of the product kernel (TTS_GEMM_ABLATE, profiles/r3_gemm_kloop_ablation.txt) show that the operand DMA
// ALONE (no fragment reads, no MFMAs) takes 160 us of the QKV projection's 205-220: with one 32 KB K tile per workgroup in flight the stream is
// bound by bytes in flight / loaded latency, so fewer bytes per FLOP is the lever. Full 128-row tiles only (M % 1024 == 0): tools/gemm_tab_bench.hip.
#pragma once
namespace tts {
static constexpr int GEMM_WIDE_LDS = 16384 + 32768;
template <int MODE>
static __global__ __launch_bounds__(512, 2) void gemm_f16_wide_kernel(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem_dyn[];
  char *smem = smem_dyn;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int mq = g.M >> 10, NT = g.N >> 8;
  if (idx >= mq * NT) return;
  const int per_chunk = mq * g.cn, chunk = idx / per_chunk, rem = idx - chunk * per_chunk, t = rem / g.cn;
  const int m0 = (xcd * mq + t) << 7, n0 = (chunk * g.cn + rem - t * g.cn) << 8;
  const int tiles_per_seg = g.kseg >> 6;
  const int ldw = g.custom_w ? g.ldw_ : g.nseg * g.kseg;
  const int prow = lane >> 3, pslot = lane & 7;
  int aoff[2], boff[4];
#pragma unroll
  for (int i = 0; i < 2; i++) {
    const int row = (wave + 8 * i) * 8 + prow;
    aoff[i] = (m0 + row) * g.lda + (pslot ^ lds_swz(row)) * 8;
  }
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int row = (wave * 4 + i) * 8 + prow;
    boff[i] = (n0 + row) * ldw + (pslot ^ lds_swz(row)) * 8;
  }
  const int fr = lane & 15, fq = lane >> 4;
  const bool resid_first = MODE == GEMM_OUT_F32 && g.resid != nullptr;
  floatx4 acc[4][4];
  if (resid_first) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int row = m0 + vh_blk(wm, i) * 16 + fr;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const float4 rr = *(const float4 *)(g.resid + (size_t)row * g.ldo + n0 + wn * 64 + j * 16 + fq * 4);
        acc[i][j] = (floatx4){rr.x, rr.y, rr.z, rr.w};
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int j = 0; j < 4; j++) acc[i][j] = (floatx4){0.f, 0.f, 0.f, 0.f};
  }
  char *sa = smem, *sb = smem + 16384;
  const bool natural = (MODE == GEMM_OUT_QKV) && (((n0 + wn * 64) % 192) >= 128);
  auto kloop = [&](auto nat) {
    constexpr bool NAT = decltype(nat)::value;
    for (int seg = 0; seg < g.nseg; seg++) {
      const __half *aseg = g.A[seg] + (ptrdiff_t)g.row_off[seg] * g.lda;
      const __half *wseg = g.W + (g.custom_w ? g.w_off_[seg] : seg * g.kseg);
      for (int kt = 0; kt < tiles_per_seg; kt++) {
        const __half *abase = aseg + (kt << 6), *wbase = wseg + (kt << 6);
#pragma unroll
        for (int i = 0; i < 2; i++) __builtin_amdgcn_global_load_lds((gptr_t)(abase + aoff[i]), (lptr_t)(sa + (wave + 8 * i) * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; i++) __builtin_amdgcn_global_load_lds((gptr_t)(wbase + boff[i]), (lptr_t)(sb + (wave * 4 + i) * 1024), 16, 0, 0);
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
          half8 af[4], bf[4];
#pragma unroll
          for (int i = 0; i < 4; i++) af[i] = *(const half8 *)(sa + lds_off(vh_blk(wm, i) * 16 + fr, ks * 4 + fq));
#pragma unroll
          for (int i = 0; i < 4; i++) bf[i] = *(const half8 *)(sb + lds_off(wn * 64 + i * 16 + fr, ks * 4 + fq));
#pragma unroll
          for (int i = 0; i < 4; i++)
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
  if (resid_first) gemm_epilogue_vh<MODE, 4, EPI_RESID_IN_ACC>(g, acc, m0, n0, wm, wn, fr, fq);
  else gemm_epilogue_vh<MODE, 4, EPI_NO_RESID>(g, acc, m0, n0, wm, wn, fr, fq);
}

static inline bool gemm_use_wide(const GemmArgs &g) { return g.M % 1024 == 0 && g.N % 256 == 0; }
static inline hipError_t launch_gemm_f16_wide(const GemmArgs &g, hipStream_t s) {
  GemmArgs gg = g;
  const int NT = g.N >> 8, ktot = g.nseg * g.kseg;
  int cn = NT;
  if (NT > 4)
    for (cn = NT; cn > 1; cn--)
      if (NT % cn == 0 && (size_t)cn * 256 * ktot * 2 <= (size_t)2560 * 1024) break;
  gg.cn = cn;
  const int grid = 8 * (g.M >> 10) * NT;
  if (g.mode == GEMM_OUT_F32) gemm_f16_wide_kernel<GEMM_OUT_F32><<<grid, 512, GEMM_WIDE_LDS, s>>>(gg);
  else if (g.mode == GEMM_OUT_F16) gemm_f16_wide_kernel<GEMM_OUT_F16><<<grid, 512, GEMM_WIDE_LDS, s>>>(gg);
  else gemm_f16_wide_kernel<GEMM_OUT_QKV><<<grid, 512, GEMM_WIDE_LDS, s>>>(gg);
  return hipGetLastError();
}
} // namespace tts

// Small-problem ("latency mode") GEMM family of the diffusion stage for gfx950: one utterance = 2 sequences x 870 frames = 1 792 packed rows,
// which the batch kernels of gemm_f16.h run at a tenth of the matrix peak (one 4-wave workgroup alone on a CU walking a 16..48-step chain of
// exposed DMA round trips, and a GroupNorm launch in front of every GEMM: 125 launches per sampling step).
//
//   C[m][n] = sum_tap sum_k  op(A)[m + tap - (TAPS == 3)][k] * W[n][tap * K + k]      (+ bias, + residual, guard rows forced to zero)
//
// One workgroup = 8 waves (2 x 4) on a 64 x (64 NJ) output tile, ONE workgroup per CU (the grid of a 1 792-row problem offers no more), so what
// the batch kernels get from four co-resident workgroups has to come from inside the workgroup:
//  * every operand is double-buffered in LDS and requested one phase (one 64-deep K tile x one tap) ahead by global_load_lds_dwordx4; a phase is
//    [s_waitcnt vmcnt(0); s_barrier; issue the next phase's loads; fragment reads + MFMAs (+ the operand transform for the next K chunk)]:
//    one barrier per phase, the loads of phase p + 1 in flight under the MFMAs of phase p, two waves per SIMD;
//  * AGN: the A operand is the f32 residual stream itself. Its raw 64 (+ 2 halo rows for the k = 3 taps) x 64-channel tile is DMA'd into LDS two K
//    chunks ahead and turned into the fp16 MFMA image by the workgroup — GroupNorm(32) normalise, gamma / beta, the timestep's scale / shift, SiLU,
//    fp16 round: the arithmetic of gn_reg_kernel, element for element in the same order — while the matrix pipe works on the previous chunk. The
//    GroupNorm launches (43 of the 125 per step) disappear; their statistics come from the PRODUCING GEMM's epilogue:
//  * STATS: the epilogue of every f32 output reduces sum / sum of squares of the stored values per (8-row chunk, 32-channel group) — sequences start at
//    multiples of 8 rows, so a chunk belongs to one sequence — and adds them to a per-(sequence, group) accumulator in 128-bit fixed point (two int64
//    atomics per quantity: units 2^-8 and 2^-60): integer addition is associative, so the statistics do not depend on the order the workgroups
//    finish in — the mode is deterministic run to run, although not bit-identical to the batch path (different K order, statistics from
//    E[x^2] - E[x]^2 in f64 instead of the two-pass form).
//  * QKV (NJ = 6: 64 x 384 = two heads per tile, 224 workgroups for N = 3 072): q | k columns with swapped operands (16-byte stores), V columns in
//    natural order (stored transposed), chosen per 16-column block at compile time; DUALB: proj_out on the split-precision weight (both halves of
//    a K tile staged, every A fragment feeds two MFMAs).
// MEASURED AND REJECTED (round 6, profiles/r6_small_gemm.txt): correct at every ring depth, 2-3x slower than the batch kernels at the single-utterance shape (a phase
// period of 1 900-3 700 cycles around ~130 cycles of MFMA per wave: per-phase fixed costs, and the GroupNorm transform re-done by every column-tile workgroup). Kept as a
// developer header (tools/r6/gemm_sm_probe.hip); the product's option latency_mode uses the batch kernels with GEMM_OUT_*_STATS epilogues + gn_apply_kernel instead.
#pragma once
#include "gemm_f16.h"

namespace tts {

struct GemmSmArgs {
  // A operand. AGN == 0: fp16 rows [-1 .. M][lda] (zero halo rows), TAPS == 1: up to two K segments (channel concat); AGN == 1: f32 rows [M][lda]
  const __half *A16[2];
  const float *A32;
  int lda, nseg, kseg; // kseg % 64 == 0
  // GroupNorm of the A operand (AGN)
  const long long *st_in;                    // [ns][32][4]: {sum hi, sum lo, sum of squares hi, lo} fixed point (see fx_add)
  const float *gamma, *beta, *scale, *shift; // [K]; scale / shift point at zeros when the block has no timestep conditioning
  const int *seq_len;                        // [ns]
  const int4 *tile_seqs;                     // [M / 64]: the (at most 4) sequences owning rows of the window [m0 - 1, m0 + 66), -1 = none
  float eps;
  int silu;
  // weights [N][ldw] fp16; TAPS == 3: tap-major K (ldw = 3 kseg); DUALB: the low half of the split pair starts at column w_lo_off
  const __half *W;
  int ldw, w_lo_off;
  int M, N; // M % 64 == 0, N % (64 NJ) == 0
  const float *bias;  // [N]
  const int *row_seq; // [M]: < 0 = guard / padding row (output forced to zero, A operand zero)
  // GEMM_OUT_F32 / GEMM_OUT_F32_SCALED: out = alpha * acc + bias + resid
  float *outF; int ldo; const float *resid; float alpha;
  // GEMM_OUT_QKV
  __half *outH; int ldh; __half *outVt; int ldvt;
  // STATS of the output
  long long *st_out;    // [ns][32][4]
  const int *chunk_seq; // [M / 8]: owning sequence of an aligned 8-row chunk, -1 = all guard rows
};

// s_waitcnt vmcnt(n) for a wave-uniform runtime n (the count of DMA pieces this wave has issued after the ones it needs: depends on the wave and, near the
// ends of the K loop, on the phase)
__device__ __forceinline__ void sm_wait_vm(int n) {
#define SM_W(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
  switch (n) {
    SM_W(0) SM_W(1) SM_W(2) SM_W(3) SM_W(4) SM_W(5) SM_W(6) SM_W(7) SM_W(8) SM_W(9) SM_W(10) SM_W(11) SM_W(12) SM_W(13) SM_W(14) SM_W(15)
    SM_W(16) SM_W(17) SM_W(18) SM_W(19) SM_W(20) SM_W(21) SM_W(22) SM_W(23) SM_W(24) SM_W(25) SM_W(26) SM_W(27) SM_W(28) SM_W(29) SM_W(30) SM_W(31)
    SM_W(32) SM_W(33) SM_W(34) SM_W(35) SM_W(36) SM_W(37) SM_W(38) SM_W(39) SM_W(40) SM_W(41) SM_W(42) SM_W(43) SM_W(44) SM_W(45) SM_W(46) SM_W(47)
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break; // more than the counter's reach: drain
  }
#undef SM_W
}

// Geometry of one instantiation. A K chunk (64 channels) is processed in PH phases that share the chunk's A image:
//   SHIFT (k = 3 convolution): phase = tap, the image is read PH - 1 rows further down each time, one accumulator set, weight columns tap * K + ..;
//   !SHIFT, PH > 1 (QKV): phase = a 64 NJ-column slice of the tile (one head), one accumulator set per slice.
// D = phases of weight tiles in flight (ring of D + 1 stages); the A operand is requested DX chunks ahead, early enough to be OLDER in the wave's
// in-order DMA queue than the weight tile of the phase that first needs it (DX PH >= D + 1 for the raw tile: it is transformed one phase early).
template <int NJ, int PH, bool SHIFT, bool AGN, bool DUALB, int D> struct SmGeo {
  static constexpr int NACC = SHIFT ? 1 : PH;
  static constexpr int BN = NJ * 64 * NACC;                 // tile columns
  static constexpr int AROWS = SHIFT ? 72 : 64;             // image rows (the k = 3 slab: rows m0 - 1 .. m0 + 70, 66 used)
  static constexpr int AUSED = SHIFT ? 64 + PH - 1 : 64;
  static constexpr int DX = AGN ? (D + 1 + PH - 1) / PH : (D + PH - 1) / PH;
  static constexpr int NIMG = AGN ? 2 : DX + 1, NRAW = AGN ? DX : 0;
  static constexpr int IMG_B = AROWS * 128, RAW_B = AROWS * 256, PAR_B = 1024, MR_B = 1024;
  static constexpr int BHALF = NJ * 64 * 128, BST_B = BHALF * (DUALB ? 2 : 1);
  static constexpr int O_RAW = NIMG * IMG_B, O_PAR = O_RAW + NRAW * RAW_B, O_MR = O_PAR + (AGN ? NRAW * PAR_B : 0), O_B = O_MR + (AGN ? MR_B : 0);
  static constexpr int LDS = O_B + (D + 1) * BST_B;
  static_assert(LDS <= 163840, "LDS");
};

#ifdef SM_TRACE // developer build (tools/r6/gemm_sm_probe.hip): phase timestamps of wave 0 of workgroup 0
__device__ long long sm_trace[64 * 8];
#define SM_T(i) do { if (blockIdx.x == 0 && tid == 0 && p < 64) sm_trace[p * 8 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define SM_T(i)
#endif
template <int MODE, int NJ, int PH, bool SHIFT, bool AGN, bool DUALB, bool STATS, int D>
static __global__ __launch_bounds__(512, 2) void gemm_f16_sm_kernel(GemmSmArgs g) {
  using G = SmGeo<NJ, PH, SHIFT, AGN, DUALB, D>;
  static_assert(!(AGN && DUALB), "not instantiated");
  constexpr int BN = G::BN, AUSED = G::AUSED, DX = G::DX, NACC = G::NACC;
  constexpr int IMG_B = G::IMG_B, RAW_B = G::RAW_B, PAR_B = G::PAR_B, BHALF = G::BHALF, BST_B = G::BST_B;
  constexpr int O_RAW = G::O_RAW, O_PAR = G::O_PAR, O_MR = G::O_MR, O_B = G::O_B;
  extern __shared__ __attribute__((aligned(16))) char smem_dyn[]; // ONE LDS object (a second one makes hipcc drain the DMA queue in front of every fragment read)
  char *smem = smem_dyn;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3, fr = lane & 15, fq = lane >> 4;
  // tile walk: workgroup b runs on XCD b % 8; an XCD owns a contiguous range of 64-row tiles and every column tile of them (its activations stay in its L2)
  const int MT = g.M >> 6, NT = g.N / BN;
  int m0, n0, tile;
  {
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int t0 = MT * xcd >> 3, t1 = MT * (xcd + 1) >> 3;
    if (idx >= (t1 - t0) * NT) return;
    tile = t0 + idx / NT;
    m0 = tile << 6;
    n0 = (idx % NT) * BN;
  }
  const int cps = g.kseg >> 6, nchunks = g.nseg * cps, nph = nchunks * PH;
  const int prow = lane >> 3, pslot = lane & 7;
  // per-lane DMA source offsets
  int boff[NJ];
#pragma unroll
  for (int i = 0; i < NJ; i++) {
    const int row = (wave * NJ + i) * 8 + prow;
    boff[i] = (n0 + row) * g.ldw + (pslot ^ lds_swz(row)) * 8;
  }
  int aoff[2] = {0, 0};    // fp16 A: piece `wave` (8 image rows), wave 0 also piece 8 (the halo rows of the k = 3 slab)
  int roff[3] = {0, 0, 0}; // raw f32 A: pieces 2 wave, 2 wave + 1 (4 rows each), wave 0 also piece 16
  const float *parp = nullptr;
  if (!AGN) {
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const int s = (wave + 8 * i) * 8 + prow;
      aoff[i] = min(m0 - (SHIFT ? 1 : 0) + s, g.M) * g.lda + (pslot ^ lds_swz(s)) * 8;
    }
  } else {
#pragma unroll
    for (int i = 0; i < 3; i++) {
      const int s = (i < 2 ? wave * 2 + i : 16) * 4 + (lane >> 4);
      roff[i] = min(max(m0 - (SHIFT ? 1 : 0) + s, 0), g.M - 1) * g.lda + (lane & 15) * 4;
    }
    const int which = lane >> 4;
    parp = (which == 0 ? g.gamma : which == 1 ? g.beta : which == 2 ? g.scale : g.shift) + (lane & 15) * 4;
  }
  // DMA pieces per wave: a weight tile, the A operand of a chunk
  const int NB = NJ * (DUALB ? 2 : 1), NA = AGN ? 3 + (SHIFT && wave == 0 ? 1 : 0) : 1 + (SHIFT && wave == 0 ? 1 : 0);
  auto issueB = [&](int p) {
    const int kc = p / PH, ph = p - kc * PH;
    const __half *src = g.W + (SHIFT ? ph * g.kseg : ph * NJ * 64 * g.ldw) + (kc << 6);
    char *dst = smem + O_B + (p % (D + 1)) * BST_B + wave * NJ * 1024;
#pragma unroll
    for (int i = 0; i < NJ; i++) __builtin_amdgcn_global_load_lds((gptr_t)(src + boff[i]), (lptr_t)(dst + i * 1024), 16, 0, 0);
    if (DUALB) {
#pragma unroll
      for (int i = 0; i < NJ; i++) __builtin_amdgcn_global_load_lds((gptr_t)(src + g.w_lo_off + boff[i]), (lptr_t)(dst + BHALF + i * 1024), 16, 0, 0);
    }
  };
  auto issueA = [&](int kc) {
    if (AGN) {
      const float *src = g.A32 + (kc << 6);
      char *dst = smem + O_RAW + (kc % DX) * RAW_B;
      __builtin_amdgcn_global_load_lds((gptr_t)(src + roff[0]), (lptr_t)(dst + (wave * 2) * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)(src + roff[1]), (lptr_t)(dst + (wave * 2 + 1) * 1024), 16, 0, 0);
      if (SHIFT && wave == 0) __builtin_amdgcn_global_load_lds((gptr_t)(src + roff[2]), (lptr_t)(dst + 16 * 1024), 16, 0, 0);
      // gamma | beta | scale | shift of the chunk: every wave requests the same 1 KB (same bytes to the same place) so that all waves count alike
      __builtin_amdgcn_global_load_lds((gptr_t)(parp + (kc << 6)), (lptr_t)(smem + O_PAR + (kc % DX) * PAR_B), 16, 0, 0);
    } else {
      const int seg = SHIFT ? 0 : kc / cps;
      const __half *src = g.A16[seg] + ((kc - seg * cps) << 6);
      char *dst = smem + (kc % (DX + 1)) * IMG_B;
      __builtin_amdgcn_global_load_lds((gptr_t)(src + aoff[0]), (lptr_t)(dst + wave * 1024), 16, 0, 0);
      if (SHIFT && wave == 0) __builtin_amdgcn_global_load_lds((gptr_t)(src + aoff[1]), (lptr_t)(dst + 8 * 1024), 16, 0, 0);
    }
  };
  // what phase q issues after its barrier: the A operand DX chunks ahead (first phase of a chunk), then the weight tile D phases ahead
  auto issued_in = [&](int q) { return ((q % PH) == 0 && q / PH + DX < nchunks ? NA : 0) + (q + D < nph ? NB : 0); };
  // operand transform of one K chunk: raw f32 tile -> GroupNorm / affine / scale-shift / SiLU -> fp16 image (the arithmetic of gn_reg_kernel)
  int myslot[3] = {-1, -1, -1};
  auto transform = [&](int kc) {
    const char *raw = smem + O_RAW + (kc % DX) * RAW_B, *par = smem + O_PAR + (kc % DX) * PAR_B;
    char *img = smem + (kc & 1) * IMG_B;
    const int quad = tid & 15, row0 = tid >> 4, grp = kc * 2 + (quad >> 3);
    const float4 ga = *(const float4 *)(par + quad * 16), be = *(const float4 *)(par + 256 + quad * 16);
    const float4 sc = *(const float4 *)(par + 512 + quad * 16), sh = *(const float4 *)(par + 768 + quad * 16);
    const float ge[4] = {ga.x, ga.y, ga.z, ga.w}, bb[4] = {be.x, be.y, be.z, be.w};
    const float s1[4] = {sc.x + 1.0f, sc.y + 1.0f, sc.z + 1.0f, sc.w + 1.0f}, s0[4] = {sh.x, sh.y, sh.z, sh.w};
#pragma unroll
    for (int k = 0; k < (AUSED > 64 ? 3 : 2); k++) {
      const int s = row0 + 32 * k;
      if (k < 2 || s < AUSED) {
        const float4 x = *(const float4 *)(raw + s * 256 + quad * 16);
        const int sl = myslot[k];
        const float2 mr = *(const float2 *)(smem + O_MR + ((max(sl, 0) * 32 + grp) << 3));
        float e[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
        for (int i = 0; i < 4; i++) {
          float u = (e[i] - mr.x) * mr.y;
          u = u * ge[i];
          u = u + bb[i];
          u = u * s1[i];
          u = u + s0[i];
          if (g.silu) u = u * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(u * -1.44269504088896f));
          e[i] = sl < 0 ? 0.f : u;
        }
        *(uint2 *)(img + lds_off(s, quad >> 1) + (quad & 1) * 8) = pack_half4(e[0], e[1], e[2], e[3]);
      }
    }
  };
  // acc[a][i][j][r] = C[m0 + wm*32 + i*16 + fr][n0 + a*64 NJ + wn*16 NJ + j*16 + fq*4 + r] (swapped operand order)
  floatx4 acc[NACC][2][NJ];
#pragma unroll
  for (int a = 0; a < NACC; a++)
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int j = 0; j < NJ; j++) acc[a][i][j] = (floatx4){0.f, 0.f, 0.f, 0.f};
  if (AGN) {
    const int4 ts = g.tile_seqs[tile];
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const int s = (tid >> 4) + 32 * k, grow = m0 - (SHIFT ? 1 : 0) + s;
      const int sq = (s < AUSED && grow >= 0 && grow < g.M) ? g.row_seq[grow] : -1;
      myslot[k] = sq < 0 ? -1 : sq == ts.x ? 0 : sq == ts.y ? 1 : sq == ts.z ? 2 : sq == ts.w ? 3 : -1;
    }
    if (tid < 128) { // mean / rstd of (slot, group) from the producer's fixed-point sums
      const int slot = tid >> 5, grp = tid & 31;
      const int seq = slot == 0 ? ts.x : slot == 1 ? ts.y : slot == 2 ? ts.z : ts.w;
      float2 mr = make_float2(0.f, 0.f);
      if (seq >= 0) {
        const long long *sp = g.st_in + (size_t)(seq * 32 + grp) * 4;
        const double n = (double)g.seq_len[seq] * 32.0;
        const double mean = fx_value(sp[0], sp[1]) / n;
        const double var = fmax(fx_value(sp[2], sp[3]) / n - mean * mean, 0.0);
        mr = make_float2((float)mean, (float)(1.0 / sqrt(var + (double)g.eps)));
      }
      *(float2 *)(smem + O_MR + (tid << 3)) = mr;
    }
    // the plain loads above are used (= waited for) HERE, before the first DMA: hipcc drains the whole DMA queue in front of any use of an ordinary load
    asm volatile("" ::"v"(myslot[0]), "v"(myslot[1]), "v"(myslot[2]));
  }
  // prologue: the A operand of the first DX chunks, then the weight tiles of the first D phases (in that order: see SmGeo)
  for (int kc = 0; kc < DX && kc < nchunks; kc++) issueA(kc);
  for (int p = 0; p < D && p < nph; p++) issueB(p);
  if (AGN) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    transform(0);
  }
  float4 rres[2][NJ]; // residual tile, requested in the last phase (no DMA is in flight any more: its wait costs nothing)
  // one phase. NAT_FROM: 16-column blocks j >= NAT_FROM use the natural operand order (V columns of the QKV projection)
  auto phase = [&](int kc, auto ph_c, auto nat_c, auto last_c) {
    constexpr int ph = decltype(ph_c)::value, NAT_FROM = decltype(nat_c)::value;
    constexpr bool LAST = decltype(last_c)::value;
    const int p = kc * PH + ph;
    // DMA pieces of this wave that are YOUNGER than the weight tile of phase p (and the A operand the phase needs, requested before it)
    SM_T(0);
    int n = NB * max(0, min(D, nph) - 1 - p);
    for (int q = max(0, p - D + 1); q < p; q++) n += issued_in(q);
    SM_T(1);
    sm_wait_vm(n);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // this wave's fragment reads / image writes of the previous phase
    SM_T(2);
    __builtin_amdgcn_s_barrier();
    SM_T(3);
    if (ph == 0 && kc + DX < nchunks) issueA(kc + DX);
    if (p + D < nph) issueB(p + D);
    SM_T(4);
    if (LAST && gemm_mode_f32(MODE) && g.resid != nullptr) {
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < NJ; j++) rres[i][j] = *(const float4 *)(g.resid + (size_t)(m0 + wm * 32 + i * 16 + fr) * g.ldo + n0 + wn * NJ * 16 + fq * 4 + j * 16);
    }
    const char *img = smem + (AGN ? (kc & 1) : kc % (DX + 1)) * IMG_B, *bst = smem + O_B + (p % (D + 1)) * BST_B;
    constexpr int a = SHIFT ? 0 : ph;
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {
      half8 af[2], bf[NJ], bl[DUALB ? NJ : 1];
#pragma unroll
      for (int i = 0; i < 2; i++) af[i] = *(const half8 *)(img + lds_off(wm * 32 + i * 16 + fr + (SHIFT ? ph : 0), ks * 4 + fq));
#pragma unroll
      for (int j = 0; j < NJ; j++) bf[j] = *(const half8 *)(bst + lds_off(wn * NJ * 16 + j * 16 + fr, ks * 4 + fq));
      if (DUALB) {
#pragma unroll
        for (int j = 0; j < NJ; j++) bl[j] = *(const half8 *)(bst + BHALF + lds_off(wn * NJ * 16 + j * 16 + fr, ks * 4 + fq));
      }
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < NJ; j++) {
          if (DUALB) acc[a][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl[j], af[i], acc[a][i][j], 0, 0, 0); // the small term first
          if (j >= NAT_FROM) acc[a][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i], bf[j], acc[a][i][j], 0, 0, 0);
          else acc[a][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[j], af[i], acc[a][i][j], 0, 0, 0);
        }
    }
    SM_T(5);
    if (AGN && ph == PH - 1 && kc + 1 < nchunks) transform(kc + 1);
    SM_T(6);
  };
  auto run = [&](auto nat_c) {
    for (int kc = 0; kc < nchunks - 1; kc++) {
      phase(kc, std::integral_constant<int, 0>{}, nat_c, std::false_type{});
      if (PH > 1) phase(kc, std::integral_constant<int, (PH > 1 ? 1 : 0)>{}, nat_c, std::false_type{});
      if (PH > 2) phase(kc, std::integral_constant<int, (PH > 2 ? 2 : 0)>{}, nat_c, std::false_type{});
    }
    const int kc = nchunks - 1;
    if (PH == 1) phase(kc, std::integral_constant<int, 0>{}, nat_c, std::true_type{});
    else phase(kc, std::integral_constant<int, 0>{}, nat_c, std::false_type{});
    if (PH == 2) phase(kc, std::integral_constant<int, (PH > 1 ? 1 : 0)>{}, nat_c, std::true_type{});
    else if (PH > 2) phase(kc, std::integral_constant<int, (PH > 1 ? 1 : 0)>{}, nat_c, std::false_type{});
    if (PH > 2) phase(kc, std::integral_constant<int, (PH > 2 ? 2 : 0)>{}, nat_c, std::true_type{});
  };
  static_assert(PH <= 3, "phases per chunk");
  // QKV, one head (192 columns = q 64 | k 64 | v 64) per phase: wave wn covers columns wn * 48 .. + 47 -> V blocks are j = 2 of wave 2 and all of wave 3
  if (MODE == GEMM_OUT_QKV && wn == 2) run(std::integral_constant<int, 2>{});
  else if (MODE == GEMM_OUT_QKV && wn == 3) run(std::integral_constant<int, 0>{});
  else run(std::integral_constant<int, NJ>{});

  // ---- epilogue: every load first, then the stores
  const bool hb = g.bias != nullptr;
  const float *bp = hb ? g.bias : (const float *)g.W;
  int sq[2];
#pragma unroll
  for (int i = 0; i < 2; i++) sq[i] = g.row_seq[m0 + wm * 32 + i * 16 + fr];
  if (MODE == GEMM_OUT_QKV) {
    static_assert(MODE != GEMM_OUT_QKV || (NJ == 3 && !SHIFT), "QKV: one head per phase");
    float4 b4[NACC][NJ];
    float bn[NACC][NJ];
    int4 sqn[2];
#pragma unroll
    for (int a = 0; a < NACC; a++)
#pragma unroll
      for (int j = 0; j < NJ; j++) {
        const int cj = n0 + a * 192 + wn * 48 + j * 16;
        b4[a][j] = *(const float4 *)(bp + cj + fq * 4);
        bn[a][j] = bp[cj + fr];
      }
#pragma unroll
    for (int i = 0; i < 2; i++) sqn[i] = *(const int4 *)(g.row_seq + m0 + wm * 32 + i * 16 + fq * 4);
#pragma unroll
    for (int a = 0; a < NACC; a++)
#pragma unroll
      for (int j = 0; j < NJ; j++) {
        const int h = n0 / 192 + a, w = wn * 48 + j * 16; // 192 | n0
        if (w >= 128) { // natural order: acc[r] = C[row = .. + fq*4 + r][col = .. + fr] -> V^T[h*64 + w - 128 + fr][row]
#pragma unroll
          for (int i = 0; i < 2; i++) {
            const int rbase = m0 + wm * 32 + i * 16 + fq * 4;
            const bool gd[4] = {sqn[i].x < 0, sqn[i].y < 0, sqn[i].z < 0, sqn[i].w < 0};
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; r++) v[r] = gd[r] ? 0.f : acc[a][i][j][r] + (hb ? bn[a][j] : 0.f);
            *(uint2 *)(g.outVt + (size_t)(h * 64 + w - 128 + fr) * g.ldvt + rbase) = pack_half4(v[0], v[1], v[2], v[3]);
          }
        } else {
#pragma unroll
          for (int i = 0; i < 2; i++) {
            const float4 b = hb ? b4[a][j] : make_float4(0.f, 0.f, 0.f, 0.f);
            float4 v = make_float4(acc[a][i][j][0] + b.x, acc[a][i][j][1] + b.y, acc[a][i][j][2] + b.z, acc[a][i][j][3] + b.w);
            if (sq[i] < 0) v = make_float4(0.f, 0.f, 0.f, 0.f);
            *(uint2 *)(g.outH + (size_t)(m0 + wm * 32 + i * 16 + fr) * g.ldh + h * 128 + w + fq * 4) = pack_half4(v.x, v.y, v.z, v.w);
          }
        }
      }
    return;
  }
  static_assert(MODE == GEMM_OUT_QKV || NACC == 1, "f32 outputs: one accumulator set");
  const int col0 = n0 + wn * NJ * 16 + fq * 4;
  const bool hr = g.resid != nullptr;
  float4 b4[NJ];
#pragma unroll
  for (int j = 0; j < NJ; j++) b4[j] = *(const float4 *)(bp + col0 + j * 16);
  int cseq[2] = {-1, -1};
  if (STATS) {
#pragma unroll
    for (int i = 0; i < 2; i++) cseq[i] = g.chunk_seq[((m0 + wm * 32 + i * 16) >> 3) + (fr >> 3)];
  }
#pragma unroll
  for (int i = 0; i < 2; i++) {
    float *op = g.outF + (size_t)(m0 + wm * 32 + i * 16 + fr) * g.ldo + col0;
    float4 v[NJ];
#pragma unroll
    for (int j = 0; j < NJ; j++) {
      floatx4 c = acc[0][i][j];
      if (MODE == GEMM_OUT_F32_SCALED) c *= g.alpha;
      const float4 b = hb ? b4[j] : make_float4(0.f, 0.f, 0.f, 0.f);
      const float4 rr = hr ? rres[i][j] : make_float4(0.f, 0.f, 0.f, 0.f);
      v[j] = make_float4((c[0] + b.x) + rr.x, (c[1] + b.y) + rr.y, (c[2] + b.z) + rr.z, (c[3] + b.w) + rr.w);
      if (sq[i] < 0) v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      *(float4 *)(op + j * 16) = v[j];
    }
    if (STATS) {
#pragma unroll
      for (int jp = 0; jp < NJ / 2; jp++) { // 32 columns = one GroupNorm group
        const float4 x = v[2 * jp], y = v[2 * jp + 1];
        float s = ((x.x + x.y) + (x.z + x.w)) + ((y.x + y.y) + (y.z + y.w));
        float q = ((x.x * x.x + x.y * x.y) + (x.z * x.z + x.w * x.w)) + ((y.x * y.x + y.y * y.y) + (y.z * y.z + y.w * y.w));
        s = red_half_block(s);
        q = red_half_block(q);
        if ((lane & 0x37) == 0 && cseq[i] >= 0) { // lanes 0 and 8: rows 0-7 / 8-15 of the block
          long long *dst = g.st_out + (size_t)(cseq[i] * 32 + ((n0 + wn * NJ * 16) >> 5) + jp) * 4;
          fx_add(dst, s);
          fx_add(dst + 2, q);
        }
      }
    }
  }
}

template <int MODE, int NJ, int PH, bool SHIFT, bool AGN, bool DUALB, bool STATS, int D>
static inline hipError_t launch_gemm_sm_t(const GemmSmArgs &g, hipStream_t s) {
  using G = SmGeo<NJ, PH, SHIFT, AGN, DUALB, D>;
  // at least 84 KB: never two of these workgroups on one CU (the grid is sized for one per CU; a pair would leave another CU idle)
  constexpr int LDS = G::LDS > 86016 ? G::LDS : 86016;
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute((const void *)gemm_f16_sm_kernel<MODE, NJ, PH, SHIFT, AGN, DUALB, STATS, D>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) return e;
    attr = true;
  }
  const int MT = g.M >> 6, NT = g.N / G::BN;
  int mx = 0;
  for (int x = 0; x < 8; x++) mx = std::max(mx, (MT * (x + 1) >> 3) - (MT * x >> 3));
  gemm_f16_sm_kernel<MODE, NJ, PH, SHIFT, AGN, DUALB, STATS, D><<<8 * mx * NT, 512, LDS, s>>>(g);
  return hipGetLastError();
}

// The shapes of the diffusion network (diffusion.hip: network_forward_lat)
enum { SM_K1_GN = 0,      // in_layers: silu(gn(x)) . W, f32 out + stats
       SM_K3_GN = 1,      // out_layers: k = 3 conv of silu(gn(h)(1 + scale) + shift), + residual, f32 out + stats
       SM_QKV_GN = 2,     // AttentionBlock norm + qkv projection
       SM_PROJ_DUALB = 3, // proj_out on the split-precision weight, + residual, f32 out + stats
       SM_K1_F16 = 4 };   // fp16 operands (integrating conv over [inp | code_emb]), f32 out + stats
#ifndef SM_DEPTHS
#define SM_DEPTHS 3, 4, 3, 2, 5
#endif
static inline hipError_t launch_gemm_sm(int kind, const GemmSmArgs &g, hipStream_t s) {
  constexpr int dd[5] = {SM_DEPTHS};
  switch (kind) {
    case SM_K1_GN: return launch_gemm_sm_t<GEMM_OUT_F32, 2, 1, false, true, false, true, dd[0]>(g, s);
    case SM_K3_GN: return launch_gemm_sm_t<GEMM_OUT_F32, 2, 3, true, true, false, true, dd[1]>(g, s);
    case SM_QKV_GN: return launch_gemm_sm_t<GEMM_OUT_QKV, 3, 2, false, true, false, false, dd[2]>(g, s);
    case SM_PROJ_DUALB: return launch_gemm_sm_t<GEMM_OUT_F32_SCALED, 2, 1, false, false, true, true, dd[3]>(g, s);
    case SM_K1_F16: return launch_gemm_sm_t<GEMM_OUT_F32, 2, 1, false, false, false, true, dd[4]>(g, s);
  }
  return hipErrorInvalidValue;
}

} // namespace tts

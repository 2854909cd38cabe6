// Measured-and-rejected (round 3): the 256-column 8-phase GEMM kernel. NOT part of the product: only tools/gemm_tab_bench.hip includes this file
// (after csrc/gemm_f16.h, inside no namespace) to keep the negative result reproducible: profiles/r3_gemm_256col_kernel.txt.
#pragma once
namespace tts {
// ------------------------------------------------------------------------------------------------------------------------------
// Large problems: (16 h) x 256 x 64 tiles, h = 8..16, 8 waves (2 x 4, each up to 128 x 64 = 8 x 4 MFMA tiles), ONE workgroup per CU,
// 128 KB of LDS as two K-tile buffers of four 16 KB half-tiles, the 8-phase schedule of cdna_hip_programming.md ("256^2 8-phase
// template"). Why a second geometry: the 128-column kernels above move 96 KB through a CU's LDS (32 KB of DMA writes + 64 KB of fragment
// reads) and 32 KB through its load path per 512 matrix-pipe cycles — the three are balanced, so none of them gets past ~60 % busy
// (K loop of a 128 x 128 tile: 1.78 us per K tile at 4 workgroups per CU = 845 cycles per K tile against 512 of MFMA work). A wave
// tile of 128 x 64 reads 12 fragments per 32 MFMAs instead of 8 per 16 and the 256-wide tile stages 64 KB per 2048 MFMA cycles:
// LDS and load path drop to ~85 % / 50 % of the matrix pipe's time.
//   half-tiles (each needed by ALL waves in ONE phase):  A-h0 = tile rows 0..127, A-h1 = rows 128..255 (blocks 8..h-1 are staged),
//                                                        B-h{x} = weight rows wc*64 + x*32 + [0,32), wc = 0..3
//   a wave (wr = wave >> 2, wc = wave & 3) owns the 16-row blocks wr, wr + 2, .. (the epilogue's interleaved mapping) and columns wc*64..
//   K tile t in buffer t & 1, four phases, each  ds_read subtile | stage one half-tile | [counted vmcnt] | barrier | MFMAs | barrier:
//     phase 0: read B-h0 (4) then A-h0 (8), quadrant (rows i < 4, cols nh 0); stage A-h1 of tile t+1
//     phase 1: read B-h1 (4),               quadrant (i < 4,  nh 1);          stage B-h0 of tile t+2
//     phase 2: read A-h1 (2 MI2),           quadrant (i >= 4, nh 1);          stage A-h0 of tile t+2
//     phase 3: -                            quadrant (i >= 4, nh 0);          stage B-h1 of tile t+2; s_waitcnt vmcnt(6)
//   vmcnt(6) leaves the three newest half-tiles (all of tile t+2's B-h0, A-h0, B-h1: every wave issues those) in flight and retires
//   everything staged up to phase 0, i.e. all of tile t+1, first read one phase (two barriers) later. The two wave groups run one
//   barrier apart (group 1 executes one extra barrier up front), so one group's MFMAs overlap the other's LDS reads and DMA issue.
// No scalar loads and no divisions inside the K loop: a K tile's segment is found by two compares and its base pointers are kept in SGPRs
// (the experiment kernel of round 2 re-fetched the segment's pointers from the kernel arguments in every phase — an s_waitcnt lgkmcnt(0)
// in front of each DMA issue that also drained the fragment reads; tools/gemm_f16_experiments.h, variant 6: 1064 TFLOP/s at 8192^3).
// ------------------------------------------------------------------------------------------------------------------------------
static constexpr int GEMM_BIG_LDS = 131072;
#ifdef TTS_GEMM_TRACE // developer build: shader-clock stamps of two K tiles of one workgroup, kept in 4 KB of LDS behind the operand buffers
__device__ unsigned tts_big_trace[8 * 64];
#define BIG_TR(i) do { if (tr_on) { const unsigned c_ = (unsigned)__builtin_readcyclecounter(); if (lane == 0) ((unsigned *)(smem + 131072))[wave * 64 + tr_slot + (i)] = c_; } } while (0)
#define BIG_TR_NEXT do { tr_slot += 5; } while (0)
#define GEMM_BIG_LDS_ALLOC (131072 + 4096)
#else
#define BIG_TR(i)
#define BIG_TR_NEXT
#define GEMM_BIG_LDS_ALLOC 131072
#endif
template <int MODE, int MI2> // MI2 = 16-row blocks of this wave beyond its first four (0..4)
__device__ __forceinline__ void gemm_big_body(const GemmArgs &g, int m0, int n0, int nblk, int lane, int wave) {
  extern __shared__ __attribute__((aligned(16))) char smem_dyn[];
  char *smem = smem_dyn;
  constexpr int MI = 4 + MI2;
  const int wr = wave >> 2, wc = wave & 3;
  const int fr = lane & 15, fq = lane >> 4;
  const int tps = g.kseg >> 6, nk = g.nseg * tps;
  const int ldw = g.custom_w ? g.ldw_ : g.nseg * g.kseg;
  // Segment bases as OFFSETS from segment 0 (wave-uniform 64-bit scalars, static indices only: a runtime-indexed g.A[seg] makes hipcc
  // copy the argument struct to scratch and fetch the pointers with vector loads in front of every DMA issue)
  const __half *A0 = g.A[0] + (ptrdiff_t)g.row_off[0] * g.lda;
  const long long dA1 = (g.A[1] + (ptrdiff_t)g.row_off[1] * g.lda) - A0, dA2 = (g.A[2] + (ptrdiff_t)g.row_off[2] * g.lda) - A0;
  const int w0 = g.custom_w ? g.w_off_[0] : 0;
  const int dW1 = (g.custom_w ? g.w_off_[1] : g.kseg) - w0, dW2 = (g.custom_w ? g.w_off_[2] : 2 * g.kseg) - w0;
  const __half *W0 = g.W + w0;
  // DMA roles: wave w moves pieces 2w, 2w+1 (8 LDS rows x 128 B each) of every half-tile; both pieces of an A half-tile lie in block h*8 + w
  const bool a1_valid = 8 + wave < nblk; // this wave's two pieces of A-h1 carry rows of the tile (wave-uniform)
  int aoff[2][2], boff[2][2];
#pragma unroll
  for (int h = 0; h < 2; h++)
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const int lr = (2 * wave + i) * 8 + (lane >> 3);
      const int c = (lane & 7) ^ lds_swz(lr);
      // A-h1 pieces past the tile's last block re-read the wave's A-h0 rows (valid memory; their LDS rows are never multiplied): no branch
      // around a DMA, every wave issues the same count
      aoff[h][i] = (m0 + (h == 1 && !a1_valid ? 0 : h * 128) + lr) * g.lda + c * 8;
      boff[h][i] = (n0 + (lr >> 5) * 64 + h * 32 + (lr & 31)) * ldw + c * 8;
    }
  // operand pointers of a K tile, advanced incrementally: +64 halves inside a segment, a jump to the next segment's offset at its first tile;
  // past the last tile the pointers stay (harmless re-stage that keeps the vmcnt arithmetic uniform)
  struct TP { long long a; int w; int t; };
  auto advance = [&](TP p) {
    const int tn = p.t + 1;
    if (tn >= nk) return p;
    TP r;
    r.t = tn;
    if (tn == tps) { r.a = dA1; r.w = dW1; }
    else if (tn == 2 * tps) { r.a = dA2; r.w = dW2; }
    else { r.a = p.a + 64; r.w = p.w + 64; }
    return r;
  };
  auto stageA = [&](const TP &tp, int h, int buf) {
    char *dst = smem + buf * 65536 + h * 16384 + (2 * wave) * 1024;
    const __half *src = A0 + tp.a;
    __builtin_amdgcn_global_load_lds((gptr_t)(src + aoff[h][0]), (lptr_t)dst, 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gptr_t)(src + aoff[h][1]), (lptr_t)(dst + 1024), 16, 0, 0);
  };
  auto stageB = [&](const TP &tp, int h, int buf) {
    char *dst = smem + buf * 65536 + 32768 + h * 16384 + (2 * wave) * 1024;
    const __half *src = W0 + tp.w;
    __builtin_amdgcn_global_load_lds((gptr_t)(src + boff[h][0]), (lptr_t)dst, 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gptr_t)(src + boff[h][1]), (lptr_t)(dst + 1024), 16, 0, 0);
  };
  const bool resid_first = MODE == GEMM_OUT_F32 && g.resid != nullptr;
  floatx4 acc[MI][4];
  if (resid_first) {
#pragma unroll
    for (int i = 0; i < MI; i++) {
      const int row = m0 + vh_blk(wr, i) * 16 + fr;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const float4 rr = *(const float4 *)(g.resid + (size_t)row * g.ldo + n0 + wc * 64 + j * 16 + fq * 4);
        acc[i][j] = (floatx4){rr.x, rr.y, rr.z, rr.w};
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // ordinary loads must not sit in the counted DMA queue below
  } else {
#pragma unroll
    for (int i = 0; i < MI; i++)
#pragma unroll
      for (int j = 0; j < 4; j++) acc[i][j] = (floatx4){0.f, 0.f, 0.f, 0.f};
  }
  // fragment read offsets inside a half-tile: A block i < 4 -> rows (2 i + wr) * 16 of A-h0, i >= 4 -> rows (2 (i - 4) + wr) * 16 of A-h1
  int ard[4][2], brd[2][2];
#pragma unroll
  for (int ks = 0; ks < 2; ks++) {
#pragma unroll
    for (int i = 0; i < 4; i++) ard[i][ks] = lds_off((2 * i + wr) * 16 + fr, ks * 4 + fq);
#pragma unroll
    for (int j = 0; j < 2; j++) brd[j][ks] = lds_off(wc * 32 + j * 16 + fr, ks * 4 + fq);
  }
  const bool natural = (MODE == GEMM_OUT_QKV) && (((n0 + wc * 64) % 192) >= 128);
  // prologue: all of tile 0, then B-h0, A-h0, B-h1 of tile 1 (the slots phases 1-3 of "tile -1" would have filled)
  TP p1; // pointers of tile t+1 (carried across K tiles); tile t+2's are derived from them in every K tile
  {
    TP p0;
    p0.a = 0; p0.w = 0; p0.t = 0;
    p1 = advance(p0);
    stageA(p0, 0, 0); stageB(p0, 0, 0); stageB(p0, 1, 0); stageA(p0, 1, 0);
    stageB(p1, 0, 1); stageA(p1, 0, 1); stageB(p1, 1, 1);
  }
  asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  __builtin_amdgcn_s_barrier();
#ifndef BIG_STAGGER
#define BIG_STAGGER 1
#endif
#if BIG_STAGGER
  if (wr == 1) __builtin_amdgcn_s_barrier(); // stagger: group 1 runs one barrier behind group 0
#endif
#ifdef TTS_GEMM_TRACE
  bool tr_on = false;
  int tr_slot = 0;
#endif
  auto kloop = [&](auto nat) {
    constexpr bool NAT = decltype(nat)::value;
    half8 af[2][4][2], bf[2][2][2];
    auto quad = [&](auto mh_c, auto nh_c) {
      constexpr int MH = decltype(mh_c)::value, NH = decltype(nh_c)::value, NI = MH == 0 ? 4 : MI2;
      BIG_TR(0); // load part of the phase issued
      __builtin_amdgcn_s_barrier();
      BIG_TR(1);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      BIG_TR(2); // fragments have arrived
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ks = 0; ks < 2; ks++)
#pragma unroll
        for (int i = 0; i < NI; i++)
#pragma unroll
          for (int j = 0; j < 2; j++) {
            if (NAT) acc[MH * 4 + i][NH * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[MH][i][ks], bf[NH][j][ks], acc[MH * 4 + i][NH * 2 + j], 0, 0, 0);
            else acc[MH * 4 + i][NH * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[NH][j][ks], af[MH][i][ks], acc[MH * 4 + i][NH * 2 + j], 0, 0, 0);
          }
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      BIG_TR(3); // MFMAs issued
      __builtin_amdgcn_s_barrier();
      BIG_TR(4);
      asm volatile("" ::: "memory");
    };
    auto ktile = [&](int t, auto bufc) {
      constexpr int BUF = decltype(bufc)::value;
#ifdef TTS_GEMM_TRACE
      tr_on = blockIdx.x == 64 && t >= 4 && t < 6; tr_slot = (t - 4) * 20;
#endif
      const char *bA0 = smem + BUF * 65536, *bA1 = bA0 + 16384, *bB0 = bA0 + 32768, *bB1 = bA0 + 49152;
      const TP p2 = advance(p1);
      // phase 0
#pragma unroll
      for (int ks = 0; ks < 2; ks++)
#pragma unroll
        for (int j = 0; j < 2; j++) bf[0][j][ks] = *(const half8 *)(bB0 + brd[j][ks]);
      asm volatile("" ::: "memory"); // the B reads stay FIRST in program order: lgkmcnt(8) below retires exactly them
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 0; ks < 2; ks++)
#pragma unroll
        for (int i = 0; i < 4; i++) af[0][i][ks] = *(const half8 *)(bA0 + ard[i][ks]);
      asm volatile("" ::: "memory");
      stageA(p1, 1, BUF ^ 1);
      asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory"); // the 4 B reads (issued first) are retired: B-h0 may be restaged next phase
      quad(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
      BIG_TR_NEXT;
      // phase 1
#pragma unroll
      for (int ks = 0; ks < 2; ks++)
#pragma unroll
        for (int j = 0; j < 2; j++) bf[1][j][ks] = *(const half8 *)(bB1 + brd[j][ks]);
      stageB(p2, 0, BUF);
      quad(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
      BIG_TR_NEXT;
      // phase 2
#pragma unroll
      for (int ks = 0; ks < 2; ks++)
#pragma unroll
        for (int i = 0; i < MI2; i++) af[1][i][ks] = *(const half8 *)(bA1 + ard[i][ks]);
      stageA(p2, 0, BUF);
      quad(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
      BIG_TR_NEXT;
      // phase 3
      stageB(p2, 1, BUF);
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); // all of tile t+1 has landed (this wave's pieces)
      quad(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{});
      p1 = p2;
    };
#ifndef BIG_PREFETCH
#define BIG_PREFETCH 0
#endif
    // BIG_PREFETCH: the fragments of phase p + 1 are read while phase p's MFMAs run (into the register halves those MFMAs do not use), so only the 4 reads of
    // bf[0] stay in front of a barrier: phase 0 prefetches bf[1](t), phase 1 af[1](t), phase 2 af[0](t+1) — whose half-tile is retired by an extra vmcnt(6) in phase 1.
    auto quad_pf = [&](auto mh_c, auto nh_c, auto prefetch) {
      constexpr int MH = decltype(mh_c)::value, NH = decltype(nh_c)::value, NI = MH == 0 ? 4 : MI2;
      __builtin_amdgcn_s_barrier();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // this phase's operands (read one phase ago, or bf[0] just now)
      __builtin_amdgcn_sched_barrier(0);
      prefetch();
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ks = 0; ks < 2; ks++)
#pragma unroll
        for (int i = 0; i < NI; i++)
#pragma unroll
          for (int j = 0; j < 2; j++) {
            if (NAT) acc[MH * 4 + i][NH * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[MH][i][ks], bf[NH][j][ks], acc[MH * 4 + i][NH * 2 + j], 0, 0, 0);
            else acc[MH * 4 + i][NH * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[NH][j][ks], af[MH][i][ks], acc[MH * 4 + i][NH * 2 + j], 0, 0, 0);
          }
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    };
    auto ktile_pf = [&](int t, auto bufc) {
      constexpr int BUF = decltype(bufc)::value;
      const char *bA1 = smem + BUF * 65536 + 16384, *bB0 = smem + BUF * 65536 + 32768, *bB1 = bB0 + 16384;
      const char *nA0 = smem + (BUF ^ 1) * 65536; // A-h0 of tile t+1
      const TP p2 = advance(p1);
      // phase 0: bf[0](t) in front of the barrier; bf[1](t) behind it
#pragma unroll
      for (int ks = 0; ks < 2; ks++)
#pragma unroll
        for (int j = 0; j < 2; j++) bf[0][j][ks] = *(const half8 *)(bB0 + brd[j][ks]);
      asm volatile("" ::: "memory");
      stageA(p1, 1, BUF ^ 1);
      quad_pf(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, [&] {
#pragma unroll
        for (int ks = 0; ks < 2; ks++)
#pragma unroll
          for (int j = 0; j < 2; j++) bf[1][j][ks] = *(const half8 *)(bB1 + brd[j][ks]);
      });
      // phase 1
      stageB(p2, 0, BUF);
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); // A-h0 / B-h0 / B-h1 of tile t+1 have landed (this wave's pieces): A-h0(t+1) is read in phase 2
      quad_pf(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}, [&] {
#pragma unroll
        for (int ks = 0; ks < 2; ks++)
#pragma unroll
          for (int i = 0; i < MI2; i++) af[1][i][ks] = *(const half8 *)(bA1 + ard[i][ks]);
      });
      // phase 2
      stageA(p2, 0, BUF);
      quad_pf(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{}, [&] {
#pragma unroll
        for (int ks = 0; ks < 2; ks++)
#pragma unroll
          for (int i = 0; i < 4; i++) af[0][i][ks] = *(const half8 *)(nA0 + ard[i][ks]);
      });
      // phase 3
      stageB(p2, 1, BUF);
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); // all of tile t+1 has landed
      quad_pf(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{}, [&] {});
      p1 = p2;
    };
#if BIG_PREFETCH
    { // af[0] of tile 0 (its half-tile was retired by the prologue's wait + barrier)
#pragma unroll
      for (int ks = 0; ks < 2; ks++)
#pragma unroll
        for (int i = 0; i < 4; i++) af[0][i][ks] = *(const half8 *)(smem + ard[i][ks]);
    }
    for (int t = 0; t < nk; t += 2) {
      ktile_pf(t, std::integral_constant<int, 0>{});
      if (t + 1 < nk) ktile_pf(t + 1, std::integral_constant<int, 1>{});
    }
#else
    for (int t = 0; t < nk; t += 2) {
      ktile(t, std::integral_constant<int, 0>{});
      if (t + 1 < nk) ktile(t + 1, std::integral_constant<int, 1>{});
    }
#endif
  };
  if (MODE == GEMM_OUT_QKV && natural) kloop(std::true_type{});
  else kloop(std::false_type{});
#if BIG_STAGGER
  if (wr == 0) __builtin_amdgcn_s_barrier(); // re-align the two groups
#endif
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // trailing (dummy) DMA pieces must land before the LDS is released
#ifdef TTS_GEMM_TRACE
  if (blockIdx.x == 64 && lane < 40) tts_big_trace[wave * 64 + lane] = ((unsigned *)(smem + 131072))[wave * 64 + lane];
#endif
  if (resid_first) gemm_epilogue_vh<MODE, MI, EPI_RESID_IN_ACC>(g, acc, m0, n0, wr, wc, fr, fq);
  else gemm_epilogue_vh<MODE, MI, EPI_NO_RESID>(g, acc, m0, n0, wr, wc, fr, fq);
}

// Tile walk of the 256-column kernel: XCD x owns a contiguous range of 16-row blocks cut into tiles of g.th blocks (8..16); a remainder
// of fewer than 8 blocks is merged into the last tile (<= 16 blocks) or the last two tiles share it evenly, so every tile has 8..16 blocks
// (each wave group then owns at least four). Chunks of g.cn column tiles outermost, as in gemm_vh_tile.
__host__ __device__ __forceinline__ int gemm_big_mtiles(int cnt, int th, int *last2) {
  const int q = cnt / th, r = cnt - q * th;
  *last2 = 0;
  if (r == 0) return q;
  if (r >= 8) return q + 1;
  if (th + r <= 16) return q;      // the last tile takes the remainder
  *last2 = th + r;                 // the last TWO tiles share th + r blocks
  return q + 1;
}
__device__ __forceinline__ bool gemm_big_tile(const GemmArgs &g, int &m0, int &n0, int &nblk) {
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int nb = g.M >> 4, NT = g.N >> 8;
  const int b0 = (int)((long long)nb * xcd >> 3), b1 = (int)((long long)nb * (xcd + 1) >> 3), cnt = b1 - b0;
  int last2;
  const int mt = gemm_big_mtiles(cnt, g.th, &last2);
  if (idx >= mt * NT) return false;
  const int per_chunk = mt * g.cn, chunk = idx / per_chunk, rem = idx - chunk * per_chunk;
  const int t = rem / g.cn;
  int blk0 = b0 + t * g.th, h = g.th;
  if (last2) {
    const int ha = (last2 + 1) >> 1;
    if (t == mt - 2) h = ha;
    else if (t == mt - 1) { blk0 = b0 + (mt - 2) * g.th + ha; h = last2 - ha; }
  } else if (t == mt - 1) h = b1 - blk0;
  m0 = blk0 << 4;
  nblk = h;
  n0 = (chunk * g.cn + rem - t * g.cn) << 8;
  return true;
}

template <int MODE>
static __global__ __launch_bounds__(512, 2) void gemm_f16_big_kernel(GemmArgs g) {
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int m0, n0, nblk;
  if (!gemm_big_tile(g, m0, n0, nblk)) return;
  const int my_mi2 = ((nblk - (wave >> 2) + 1) >> 1) - 4; // blocks of this wave beyond the first four
  if (my_mi2 == 4) gemm_big_body<MODE, 4>(g, m0, n0, nblk, lane, wave);
  else if (my_mi2 == 3) gemm_big_body<MODE, 3>(g, m0, n0, nblk, lane, wave);
  else if (my_mi2 == 2) gemm_big_body<MODE, 2>(g, m0, n0, nblk, lane, wave);
  else if (my_mi2 == 1) gemm_big_body<MODE, 1>(g, m0, n0, nblk, lane, wave);
  else gemm_big_body<MODE, 0>(g, m0, n0, nblk, lane, wave);
}

// Large problems go to the 256-column kernel: at least 64 sixteen-row blocks per XCD (M >= 8192 rows: one workgroup per CU needs whole rounds of
// tall tiles; the AR multi-row passes and a single utterance stay on the 128-column kernels) and N a multiple of 256.
// g.th > 0 pins the 128-column kernels (developer tools); TTS_GEMM_BIG=0 is the A/B switch.
static inline bool gemm_use_big(const GemmArgs &g) { return (g.N & 255) == 0 && g.N >= 512 && (g.M >> 4) / 8 >= 64; }
static inline hipError_t launch_gemm_f16_big(const GemmArgs &g, hipStream_t s) {
  GemmArgs gg = g;
  const int NT = g.N >> 8, ktot = g.nseg * g.kseg, nb = g.M >> 4;
  int cn = NT; // L2 chunk: the largest divisor of NT whose weight rows fit ~2.5 MB (N = 3072, K = 1024: 4 column tiles = 2 MB)
  if (NT > 4)
    for (cn = NT; cn > 1; cn--)
      if (NT % cn == 0 && (size_t)cn * 256 * ktot * 2 <= (size_t)2560 * 1024) break;
  gg.cn = cn;
  // tile height (8..16 blocks): one workgroup per CU means rounds ARE rounds here (no co-resident workgroup speeds up when a CU's
  // neighbour slot is empty) -> minimise rounds x height over the XCD with the most blocks
  static int cus_per_xcd = 0;
  if (!cus_per_xcd) {
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    cus_per_xcd = cus / 8 > 0 ? cus / 8 : 32;
  }
  int cnt_max = 0;
  for (int x = 0; x < 8; x++) cnt_max = std::max(cnt_max, (int)((long long)nb * (x + 1) >> 3) - (int)((long long)nb * x >> 3));
  int best_th = 16, best_cost = 1 << 30, last2;
  for (int th = 16; th >= 8; th--) {
    const int tiles = gemm_big_mtiles(cnt_max, th, &last2) * NT, rounds = (tiles + cus_per_xcd - 1) / cus_per_xcd;
    const int cost = rounds * th;
    if (cost < best_cost) { best_cost = cost; best_th = th; }
  }
  gg.th = best_th;
  const int grid = 8 * gemm_big_mtiles(cnt_max, best_th, &last2) * NT;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void *)gemm_f16_big_kernel<GEMM_OUT_F32>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_BIG_LDS_ALLOC);
    (void)hipFuncSetAttribute((const void *)gemm_f16_big_kernel<GEMM_OUT_F16>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_BIG_LDS_ALLOC);
    (void)hipFuncSetAttribute((const void *)gemm_f16_big_kernel<GEMM_OUT_QKV>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_BIG_LDS_ALLOC);
    attr = true;
  }
  if (g.mode == GEMM_OUT_F32) gemm_f16_big_kernel<GEMM_OUT_F32><<<grid, 512, GEMM_BIG_LDS_ALLOC, s>>>(gg);
  else if (g.mode == GEMM_OUT_F16) gemm_f16_big_kernel<GEMM_OUT_F16><<<grid, 512, GEMM_BIG_LDS_ALLOC, s>>>(gg);
  else gemm_f16_big_kernel<GEMM_OUT_QKV><<<grid, 512, GEMM_BIG_LDS_ALLOC, s>>>(gg);
  return hipGetLastError();
}


} // namespace tts

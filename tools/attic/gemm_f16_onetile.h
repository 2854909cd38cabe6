// fp16 x fp16 -> f32 MFMA GEMM for gfx950 (v_mfma_f32_16x16x32_f16), used by the diffusion and
// vocoder stages for every convolution (the reference's conv1d IS an fp16 im2col GEMM with f32
// accumulation), for the attention projections, and (with split-precision operands) by the
// multi-row passes of the autoregressive stage.
//
//   C[m][n] = sum_seg sum_k A_seg[m + row_off_seg][k] * W[n][seg*kseg + k]
//
// A "segment" is one convolution tap (same activation buffer, row offset -1/0/+1: sequences are
// packed along M with a zero guard row between them, so a k=3 convolution needs no im2col and no
// boundary masking) or one half of a channel concat (two buffers, offset 0).
//
// Two product kernels, both 128 x 128 x 64 tiles, 4 waves (2 x 2), each wave 4 x 4 MFMA tiles, operands staged
// by direct global->LDS DMA (global_load_lds_dwordx4) into a lane-linear image whose 16-byte chunks are
// XOR-swizzled by (row>>1)&7 on the SOURCE address, so ds_read_b128 fragment reads spread over all banks:
//   gemm_f16_glds_kernel   any segment structure; one LDS stage, 4 workgroups per CU overlap each other
//   gemm_f16_conv3_kernel  the k=3 convolution: one activation slab shared by the three taps, weight tiles
//                          double-buffered
// launch_gemm_f16 picks between them. Measured-and-rejected variants: tools/gemm_f16_experiments.h (only tools/gemm_bench.hip
// defines TTS_GEMM_VARIANT and adds -I tools).
#pragma once
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <type_traits>

namespace tts {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

enum { GEMM_OUT_F32 = 0, GEMM_OUT_F16 = 1, GEMM_OUT_QKV = 2 };

struct GemmArgs {
  const __half *A[3];  // per segment base (row 0 of the packed layout)
  int row_off[3];
  int nseg, kseg;      // kseg % 64 == 0
  int lda;             // halves
  const __half *W;     // [N][ldw]; segment seg starts at column w_off[seg] (defaults: ldw = nseg*kseg, w_off = seg*kseg)
  int ldw_, w_off_[3], custom_w; // set custom_w = 1 to use ldw_/w_off_ (e.g. split-precision: hi|lo halves reused)
  int M, N;            // multiples of 128 (buffers are padded)
  int cn;              // n-tiles per L2 chunk (0 = all; chosen by launch_gemm_f16)
  const float *bias;   // [N] or nullptr
  const int *row_seq;  // [M]: sequence id, <0 for guard/padding rows (output forced to 0); may be null
  // GEMM_OUT_F32
  float *outF; int ldo; const float *resid; // resid may alias outF
  // GEMM_OUT_F16 (n_valid columns written) / GEMM_OUT_QKV
  __half *outH; int ldh;
  __half *outVt; int ldvt; // QKV: V channels transposed [h*64+d][row]
  int mode;
  // Round stagger (set by launch_gemm_f16): of the workgroups resident in the first round (index inside the XCD < stagger_slots) those
  // with an odd (index / stagger_div) start stagger_ticks (100 MHz wall clock) late. All tiles take the same time, so the two
  // populations stay half a tile apart for the whole launch: one's epilogue burst (HBM-bound, matrix pipe idle) falls into the other's K loop.
  int stagger_ticks, stagger_slots, stagger_div;
  int kmajor; // tools/gemm_f16_deep.h only: K tiles visited chunk-major with the segments (taps) innermost — the k = 3 kernel's order
};

__device__ __forceinline__ void gemm_round_stagger(const GemmArgs &g, int idx_in_xcd) {
  if (g.stagger_ticks > 0 && idx_in_xcd < g.stagger_slots && ((idx_in_xcd / g.stagger_div) & 1)) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < (unsigned long long)g.stagger_ticks) __builtin_amdgcn_s_sleep(16);
  }
}

#ifdef TTS_GEMM_TRACE // developer build (tools/gemm_diag.hip): phase timestamps (100 MHz wall clock) of every workgroup / tile
__device__ unsigned long long tts_gemm_trace[65536 * 8];
#define GEMM_TR_DECL unsigned long long tr_[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define GEMM_TR(i) do { tr_[i] = wall_clock64(); } while (0)
#define GEMM_TR_FLUSH(slot) do { if (threadIdx.x == 0 && (slot) < 65536) { tr_[6] = __builtin_amdgcn_s_getreg((4 /*HW_ID*/) | (0 << 6) | (31 << 11)); \
  tr_[7] = __builtin_amdgcn_s_getreg((20 /*XCC_ID*/) | (0 << 6) | (31 << 11)); for (int q_ = 0; q_ < 8; q_++) tts_gemm_trace[(size_t)(slot) * 8 + q_] = tr_[q_]; } } while (0)
#else
#define GEMM_TR_DECL
#define GEMM_TR(i)
#define GEMM_TR_FLUSH(slot)
#endif

__device__ __forceinline__ int lds_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

// Single 32 KB LDS stage filled by direct global->LDS DMA (global_load_lds_dwordx4): no
// staging VGPRs, no ds_write pass; 3-4 workgroups per CU overlap each other's load/compute phases.
// The LDS image of a DMA is lane-linear (base + lane*16), so the XOR swizzle is applied to the per-lane
// SOURCE address and again on the fragment read.
typedef const void __attribute__((address_space(1))) *gptr_t;
typedef void __attribute__((address_space(3))) *lptr_t;

// Epilogues. For F32/F16 outputs the MFMA operands are swapped (A-operand = weight rows, B-operand =
// activation rows), so a lane's 4 accumulator registers are 4 CONSECUTIVE output columns of one row:
// residual loads and stores are 16 bytes per lane instead of 4. The QKV mode keeps the natural order
// (a lane holds 4 consecutive rows of one column) because V is stored transposed.
// m_lim: rows >= m_lim are not stored (g.M for a whole tile; the end of the chunk for the balanced kernels)
template <int MODE, int MI, bool RESID_IN_ACC = false>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs &g, floatx4 (&acc)[MI][4], int m0, int n0, int wm, int wn, int fr, int fq, int m_lim) {
  if (MODE == GEMM_OUT_QKV) {
    // col = h*192 + {q 0..63 | k 64..127 | v 128..191}; a wave's 64-column span is entirely q, k or v.
    const int c0 = n0 + wn * 64, h = c0 / 192, w0 = c0 - h * 192;
    if (w0 >= 128) { // V, natural operand order: lane = 4 consecutive rows of one column -> 8-byte transposed store
#pragma unroll
      for (int i = 0; i < MI; i++) {
        const int rbase = m0 + wm * (16 * MI) + i * 16 + fq * 4;
        if (rbase >= m_lim) continue; // limits are multiples of 4: a lane's 4 rows are in or out together
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
    } else { // Q or K, swapped operand order: lane = 4 consecutive columns of one row -> 8-byte store
#pragma unroll
      for (int i = 0; i < MI; i++) {
        const int row = m0 + wm * (16 * MI) + i * 16 + fr;
        if (row >= m_lim) continue;
        const bool guard = g.row_seq ? (g.row_seq[row] < 0) : false;
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const int d = j * 16 + fq * 4;
          float4 v = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
          if (g.bias) {
            const float4 b = *(const float4 *)(g.bias + c0 + d);
            v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
          }
          if (guard) v = make_float4(0.f, 0.f, 0.f, 0.f);
          __half2 p0 = __floats2half2_rn(v.x, v.y), p1 = __floats2half2_rn(v.z, v.w);
          uint2 u;
          u.x = *(unsigned *)&p0;
          u.y = *(unsigned *)&p1;
          *(uint2 *)(g.outH + (size_t)row * g.ldh + h * 128 + w0 + d) = u;
        }
      }
    }
  } else {
    // acc[i][j][r] = C[m0 + wm*16*MI + i*16 + fr][n0 + wn*64 + j*16 + fq*4 + r]
#pragma unroll
    for (int i = 0; i < MI; i++) {
      const int row = m0 + wm * (16 * MI) + i * 16 + fr;
      if (row >= m_lim) continue;
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
          if (g.resid && !RESID_IN_ACC) {
            const float4 rr = *(const float4 *)(g.resid + (size_t)row * g.ldo + col);
            v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
          }
          if (guard) v = make_float4(0.f, 0.f, 0.f, 0.f);
          *(float4 *)(g.outF + (size_t)row * g.ldo + col) = v; // (non-temporal stores measured: diffusion 916-922 vs 906-907 ms, call c15)
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
}

// Accumulators start from the residual (F32 outputs, swapped operand order: acc[i][j] = 4 consecutive columns of
// one row): the residual read is issued with the first operand tile and hides behind it, instead of being a
// dependent HBM round trip in front of the stores when the K loop is over. The sum is the same set of f32 adds in
// a different order (the product sums land on the residual one MFMA at a time).
template <int MI>
__device__ __forceinline__ void gemm_acc_from_resid(const GemmArgs &g, floatx4 (&acc)[MI][4], int m0, int n0, int wm, int wn, int fr, int fq) {
#pragma unroll
  for (int i = 0; i < MI; i++) {
    const int row = min(m0 + wm * (16 * MI) + i * 16 + fr, g.M - 1);
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const float4 rr = *(const float4 *)(g.resid + (size_t)row * g.ldo + n0 + wn * 64 + j * 16 + fq * 4);
      acc[i][j] = (floatx4){rr.x, rr.y, rr.z, rr.w};
    }
  }
}

// MI = 16-row MFMA tiles per wave along M: the workgroup tile is (32 MI) x 128. MI = 4 (128 rows) is the default;
// MI = 5 (160 rows) is chosen by launch_gemm_f16 when it turns a 2.3-round grid into fewer, fuller rounds.
// WGS: workgroups per CU the register allocation is bounded for. The F32/F16 modes fit 128 VGPRs (4 per CU) unforced;
// the QKV mode needs 138 and gets 4 per CU with WGS = 4 at the price of 9 VGPRs spilled around (not inside) the K loop.
template <int MODE, int MI, int WGS = 3>
static __global__ __launch_bounds__(256, WGS) void gemm_f16_glds_kernel(GemmArgs g) {
  constexpr int BM = 32 * MI;
  __shared__ __attribute__((aligned(16))) char smem[BM * 128 + 16384];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6); // scalar: LDS-DMA bases stay in SGPRs
  const int wm = wave >> 1, wn = wave & 1;
  // L2-aware tile order. Workgroup b runs on XCD b % 8 (each XCD has its own 4 MB L2). An XCD owns a
  // contiguous range of m-tiles and walks them once per chunk of `cn` n-tiles, chunk outermost: the chunk's
  // weight rows (cn * 128 * K * 2 B <= ~2.5 MB) stay L2-resident while the activations stream through.
  // (With plain m-major order the 6 MB QKV weight thrashed L2: 417 MB fetched per launch for 64 MB of operands.)
  const int MT = (g.M + BM - 1) / BM, NT = g.N >> 7;
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int mq = MT >> 3, mr = MT & 7;
  const int mcount = mq + (xcd < mr ? 1 : 0), mfirst = xcd * mq + (xcd < mr ? xcd : mr);
  if (idx >= mcount * NT) return; // grid is padded to 8 * max tiles per XCD
  gemm_round_stagger(g, idx);
  GEMM_TR_DECL;
  GEMM_TR(0);
  const int cn = g.cn > 0 ? g.cn : NT, per_chunk = mcount * cn;
  const int chunk = idx / per_chunk, rem = idx - chunk * per_chunk;
  const int m0 = (mfirst + rem / cn) * BM, n0 = (chunk * cn + rem % cn) << 7;
  const int tiles_per_seg = g.kseg >> 6;
  const int ldw = g.custom_w ? g.ldw_ : g.nseg * g.kseg;
  // DMA roles: wave w, piece i covers rows (w*MI+i)*8 .. +7 of the A tile; (w*4+i)*8 .. +7 of the B tile
  const int prow = lane >> 3, pslot = lane & 7;
  int aoff[MI], boff[4];
#pragma unroll
  for (int i = 0; i < MI; i++) {
    const int row = (wave * MI + i) * 8 + prow;
    aoff[i] = min(m0 + row, g.M - 1) * g.lda + (pslot ^ ((row >> 1) & 7)) * 8; // rows past M re-read the last row (never stored)
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
  // operand order (see gemm_epilogue): natural only for the V columns of a QKV projection (wave-uniform)
  const bool natural = (MODE == GEMM_OUT_QKV) && (((n0 + wn * 64) % 192) >= 128);
  GEMM_TR(1);
  // the K loop is instantiated once per operand order so the choice costs nothing inside it
  auto kloop = [&](auto nat) {
    constexpr bool NAT = decltype(nat)::value;
    // segment loop outside, K tiles inside: the segment's base pointers are fetched from the kernel arguments once,
    // not by a scalar load (and an integer division) in front of every tile's DMA issue
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
      __syncthreads(); // waits vmcnt(0) for the DMA, then barrier
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
#ifdef TTS_GEMM_DIAG_NOEPI // tools/gemm_diag.hip: the K loop alone (the runtime condition keeps the accumulators live)
  if (g.ldo != -12345) { GEMM_TR(4); GEMM_TR_FLUSH(blockIdx.x); return; }
#endif
  if (resid_first) gemm_epilogue<MODE, MI, true>(g, acc, m0, n0, wm, wn, fr, fq, g.M);
  else gemm_epilogue<MODE, MI>(g, acc, m0, n0, wm, wn, fr, fq, g.M);
  GEMM_TR(4);
#ifdef TTS_GEMM_TRACE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // stamp 5: the epilogue's stores have been acknowledged
  GEMM_TR(5);
#endif
  GEMM_TR_FLUSH(blockIdx.x);
}

// k = 3 convolution as ONE GEMM with a shared activation slab. The three taps are three row-shifted GEMM
// segments over the SAME activation rows (row_off = -1, 0, +1), so per 64-channel chunk the kernel stages the
// 130 (136) activation rows m0-1 .. once and multiplies them three times, each time against that tap's weight
// tile and read from LDS one row further down. Operand traffic through the CU's load path per chunk:
// 17 + 3 x 16 = 65 DMA pieces instead of 3 x 32 = 96 — the 128^2 tile is bound by exactly that path
// (64 B/clk/CU feeds at most one 32 KB K tile per 512 MFMA cycles).
// The weight tiles alternate between two LDS buffers: tap p+1's tile is requested before tap p's is waited for
// (counted vmcnt + raw barrier), so only the slab load at the start of a chunk is exposed.
template <int MI> constexpr int conv3_lds() { return (32 * MI + 8) * 128 + 2 * 16384; }
// (Residual handling, measured: read in the epilogue as below 203-216 us; accumulators started from it 237-249 us — the pending loads sit in
//  front of the pipelined weight DMA in the in-order vmcnt queue; "dripped" into the accumulators block by block inside the K loop by inline-asm
//  loads with two phases of latency budget 210-225 us — the epilogue halves (20 -> 11 us per tile) but the K loop grows by 15 us: a true HBM read
//  in the in-order queue gates the retirement of the L2-hit weight tiles behind it. profiles/r2_gemm_tile_phases.txt, c8/c9.)
template <int MODE, int MI>
static __global__ __launch_bounds__(256, 3) void gemm_f16_conv3_kernel(GemmArgs g) {
  constexpr int BM = 32 * MI, SLAB = (BM + 8) * 128;
  extern __shared__ __attribute__((aligned(16))) char smem_dyn[]; // A slab BM+8 rows | B tile x 2
  char *smem = smem_dyn;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6); // scalar: LDS-DMA bases stay in SGPRs
  const int wm = wave >> 1, wn = wave & 1;
  const int MT = (g.M + BM - 1) / BM, NT = g.N >> 7;
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int mq = MT >> 3, mr = MT & 7;
  const int mcount = mq + (xcd < mr ? 1 : 0), mfirst = xcd * mq + (xcd < mr ? xcd : mr);
  if (idx >= mcount * NT) return;
  gemm_round_stagger(g, idx);
  const int cn = g.cn > 0 ? g.cn : NT, per_chunk = mcount * cn;
  const int chunk = idx / per_chunk, rem = idx - chunk * per_chunk;
  const int m0 = (mfirst + rem / cn) * BM, n0 = (chunk * cn + rem % cn) << 7;
  const int nchunks = g.kseg >> 6, ldw = 3 * g.kseg, nph = 3 * nchunks;
  GEMM_TR_DECL;
  GEMM_TR(0);
  const int prow = lane >> 3, pslot = lane & 7;
  const int fr = lane & 15, fq = lane >> 4;
  floatx4 acc[MI][4];
#pragma unroll
  for (int i = 0; i < MI; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = (floatx4){0.f, 0.f, 0.f, 0.f};
  char *sa = smem, *sb = smem + SLAB;
  // slab row s = activation row m0 - 1 + s (the buffer has its guard rows, as for the plain segments)
  const __half *abase = g.A[0] + (ptrdiff_t)(m0 - 1) * g.lda;
  const __half *wbase = g.W + (size_t)n0 * ldw;
  int aoff[MI + 1], boff[4];
#pragma unroll
  for (int i = 0; i <= MI; i++) {
    const int row = (i < MI ? wave * MI + i : 4 * MI) * 8 + prow;
    const int c = pslot ^ ((row >> 1) & 7);
    aoff[i] = min(row, g.M - m0 + 1) * g.lda + c * 8; // rows past the buffer end are never multiplied: clamp
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
  auto stageB = [&](int p) { // phase p = chunk p / 3, tap p % 3 (clamped past the end: uniform vmcnt arithmetic)
    p = min(p, nph - 1);
    const int kc = p / 3, tap = p - kc * 3;
    const __half *src = wbase + tap * g.kseg + (kc << 6);
    char *dst = sb + (p & 1) * 16384;
#pragma unroll
    for (int i = 0; i < 4; i++) __builtin_amdgcn_global_load_lds((gptr_t)(src + boff[i]), (lptr_t)(dst + (wave * 4 + i) * 1024), 16, 0, 0);
  };
  // one phase; TAP = p % 3 is compile-time (the caller unrolls the three taps of a K chunk)
  auto phase = [&](int kc, int p, auto tap_c) {
    constexpr int TAP = decltype(tap_c)::value;
    stageB(p + 1); // its buffer was last read in phase p-1, which every wave has left (trailing barrier)
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); // all but the 4 pieces just issued: B(p) and the slab have landed
    __builtin_amdgcn_s_barrier();
#ifdef TTS_GEMM_TRACE
    if (p == 0) GEMM_TR(2);
#endif
    const char *sbp = sb + (p & 1) * 16384;
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {
      half8 af[MI], bf[4];
#pragma unroll
      for (int i = 0; i < MI; i++) af[i] = *(const half8 *)(sa + lds_off(wm * (16 * MI) + i * 16 + fr + TAP, ks * 4 + fq));
#pragma unroll
      for (int i = 0; i < 4; i++) bf[i] = *(const half8 *)(sbp + lds_off(wn * 64 + i * 16 + fr, ks * 4 + fq));
#pragma unroll
      for (int i = 0; i < MI; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[j], af[i], acc[i][j], 0, 0, 0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier(); // every wave is done reading the slab and B(p)
    if (TAP == 2) stageA(kc + 1);
  };
  stageA(0);
  stageB(0);
  GEMM_TR(1);
  for (int kc = 0; kc < nchunks; kc++) {
    const int p = 3 * kc;
    phase(kc, p, std::integral_constant<int, 0>{});
    phase(kc, p + 1, std::integral_constant<int, 1>{});
    phase(kc, p + 2, std::integral_constant<int, 2>{});
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // trailing (clamped) pieces must land before the LDS is released
  GEMM_TR(3);
#ifdef TTS_GEMM_DIAG_NOEPI
  if (g.ldo != -12345) { GEMM_TR(4); GEMM_TR_FLUSH(blockIdx.x); return; }
#endif
  gemm_epilogue<MODE, MI>(g, acc, m0, n0, wm, wn, fr, fq, g.M);
  GEMM_TR(4);
#ifdef TTS_GEMM_TRACE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  GEMM_TR(5);
#endif
  GEMM_TR_FLUSH(blockIdx.x);
}

#ifdef TTS_GEMM_DIAG // tools/gemm_diag.hip only (compiled with -I tools): the measured-and-rejected balanced persistent kernels
#include "gemm_f16_balanced.h"
#endif
#ifdef TTS_GEMM_VARIANT // tools/gemm_bench.hip only (compiled with -I tools)
#include "gemm_f16_experiments.h"
#endif
#ifdef TTS_GEMM_DEEP // tools/gemm_small_diag.hip only (compiled with -I tools)
#include "gemm_f16_deep.h"
#endif

static inline float &gemm_stagger_us() {
  static float v = getenv("TTS_GEMM_STAGGER_US") ? (float)atof(getenv("TTS_GEMM_STAGGER_US")) : -1.f; // < 0: per-mode default
  return v;
}
static inline int &gemm_stagger_div() {
  static int v = getenv("TTS_GEMM_STAGGER_DIV") ? atoi(getenv("TTS_GEMM_STAGGER_DIV")) : 0; // 0: default (32)
  return v;
}

static inline hipError_t launch_gemm_f16(const GemmArgs &g, hipStream_t s) {
  GemmArgs gg = g;
  {
    const int NT = g.N >> 7, ktot = g.nseg * g.kseg;
    int cn = NT;
    static const bool no_chunk = getenv("TTS_GEMM_NOCHUNK") != nullptr; // A/B switch for tools/gemm_bench
    // measured (tools/gemm_bench): chunking pays for wide outputs (N=3072: 248-268 vs 275-279 us) and costs ~3% when
    // the activations would have to stream twice for a narrow one (N=1024, K=3072) -> only chunk when NT > 8
    while (!no_chunk && NT > 8 && cn > 1 && (cn % 2 == 0) && (size_t)cn * 128 * ktot * 2 > (size_t)2560 * 1024) cn /= 2;
    gg.cn = cn;
  }
#ifdef TTS_GEMM_VARIANT // tools/gemm_bench.hip: measured-and-rejected variants live in tools/gemm_f16_experiments.h
  if (TTS_GEMM_VARIANT != 1) return launch_gemm_experiment(g, s);
#endif
  // Tile height: 128 rows. 160-row tiles (MI = 5: 1416 instead of 1768 workgroups on 768 slots) were measured 4-6 %
  // SLOWER on all three shapes (tools/gemm_bench, TTS_GEMM_MI=5): workgroups are dispatched continuously, not in
  // rounds, so there is no 2.3 -> 3 round quantisation to win back. The instantiation is kept for the A/B switch.
  static const char *force_mi = getenv("TTS_GEMM_MI");
  const int NTt = g.N >> 7;
  // Small problems (a single utterance: M = 1 792 rows -> 14 x 8 tiles of 128 rows on 256 CUs): 64-row tiles put twice as
  // many workgroups on the chip. Threshold measured at one candidate (diffusion stage): 256 tiles 183 ms, 512 tiles 174 ms,
  // 1024 tiles 175 ms; 16 candidates are above all of them. TTS_GEMM_MI=4 keeps the 128-row tile everywhere (A/B switch).
  const int mt4 = (g.M + 127) / 128;
  static const int mi2_tiles = getenv("TTS_GEMM_MI2_TILES") ? atoi(getenv("TTS_GEMM_MI2_TILES")) : 512;
  int mi = (mt4 * NTt < mi2_tiles) ? 2 : 4;
  if (force_mi) { const int f = atoi(force_mi); if (f == 5 || f == 4 || f == 2) mi = f; }
  const int bm = 32 * mi, MTt = (g.M + bm - 1) / bm, grid1 = 8 * ((MTt >> 3) + ((MTt & 7) ? 1 : 0)) * NTt;
  // k = 3 convolution (three row-shifted segments of one activation buffer, tap-major weights): shared-slab kernel
  static const bool no_conv3 = getenv("TTS_GEMM_NOCONV3") != nullptr; // A/B switch for tools/gemm_bench
  const bool conv3 = !no_conv3 && g.nseg == 3 && !g.custom_w && g.A[0] == g.A[1] && g.A[1] == g.A[2] && g.row_off[0] == -1 &&
                     g.row_off[1] == 0 && g.row_off[2] == 1 && g.mode != GEMM_OUT_QKV;
  // A/B switch: f32-output GEMMs at 3 workgroups per CU (the per-tile trace shows 768 tiles resident with the (256, 3) launch bound although
  // the kernel needs only 128 VGPRs; with (256, 4) 1024 are, and M = 28 032 x N = 1024 is 1752 tiles: 2 rounds instead of 3)
  static const bool wgs3 = getenv("TTS_GEMM_WGS3") != nullptr;
  static const bool qkv3 = getenv("TTS_GEMM_QKV3") != nullptr; // A/B switch: QKV projection at 3 workgroups per CU (138 VGPRs, no spills)
  // A/B switches for the round stagger: TTS_GEMM_STAGGER_US (delay in microseconds for every mode; 0 = off; unset = per-mode default),
  // TTS_GEMM_STAGGER_DIV (1: alternate workgroups, N: alternate groups of N consecutive workgroups of an XCD)
  static int cus_per_xcd = 0;
  if (!cus_per_xcd) {
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    cus_per_xcd = cus / 8 > 0 ? cus / 8 : 32;
  }
  {
    // OFF by default. Stand-alone (tools/gemm_diag, back-to-back launches of one shape) the QKV projection gained 8 % with groups of 32
    // workgroups 20 us apart (241.6 -> 221.6 us) and the f32-output GEMMs lost 2-10 % with any stagger; inside the diffusion step the QKV
    // default measured nothing (bench 129.9 vs 130.3-130.6 audio-s/s without): kept as a switch, documented in DESIGN.md.
    const float us = gemm_stagger_us() >= 0 ? gemm_stagger_us() : 0.f;
    const int wgs_res = conv3 ? 3 : (g.mode == GEMM_OUT_QKV && qkv3) ? 3 : (g.mode == GEMM_OUT_F16 || (g.mode == GEMM_OUT_F32 && wgs3)) ? 3 : 4;
    gg.stagger_ticks = us > 0 ? (int)(us * 100.0f) : 0;
    gg.stagger_slots = cus_per_xcd * wgs_res;
    gg.stagger_div = gemm_stagger_div() > 0 ? gemm_stagger_div() : 32;
    // only launches with more tiles than resident slots have rounds to stagger
    if (grid1 / 8 <= gg.stagger_slots) gg.stagger_ticks = 0;
  }
#ifdef TTS_GEMM_DIAG
  { hipError_t e_; if (launch_gemm_balanced(g, gg, conv3, force_mi != nullptr, NTt, cus_per_xcd, s, &e_)) return e_; }
#endif
#ifdef TTS_GEMM_DEEP // tools/gemm_small_diag.hip only: the measured-and-rejected deep-ring kernel for small problems
  { hipError_t e_; if (launch_gemm_deep_if_small(g, gg, mi, conv3, grid1, cus_per_xcd, s, &e_)) return e_; }
#endif
#define TTS_LAUNCH_MI(MI_)                                                                                              \
  do {                                                                                                                  \
    if (conv3) {                                                                                                        \
      static bool attr = false;                                                                                         \
      if (!attr) {                                                                                                      \
        (void)hipFuncSetAttribute((const void *)gemm_f16_conv3_kernel<GEMM_OUT_F32, MI_>, hipFuncAttributeMaxDynamicSharedMemorySize, conv3_lds<MI_>()); \
        (void)hipFuncSetAttribute((const void *)gemm_f16_conv3_kernel<GEMM_OUT_F16, MI_>, hipFuncAttributeMaxDynamicSharedMemorySize, conv3_lds<MI_>()); \
        attr = true;                                                                                                    \
      }                                                                                                                 \
      if (g.mode == GEMM_OUT_F32) gemm_f16_conv3_kernel<GEMM_OUT_F32, MI_><<<grid1, 256, conv3_lds<MI_>(), s>>>(gg); \
      else gemm_f16_conv3_kernel<GEMM_OUT_F16, MI_><<<grid1, 256, conv3_lds<MI_>(), s>>>(gg);                            \
    } else if (g.mode == GEMM_OUT_F32 && wgs3) gemm_f16_glds_kernel<GEMM_OUT_F32, MI_><<<grid1, 256, 0, s>>>(gg);        \
    else if (g.mode == GEMM_OUT_F32) gemm_f16_glds_kernel<GEMM_OUT_F32, MI_, (MI_ <= 4 ? 4 : 3)><<<grid1, 256, 0, s>>>(gg); \
    else if (g.mode == GEMM_OUT_F16) gemm_f16_glds_kernel<GEMM_OUT_F16, MI_><<<grid1, 256, 0, s>>>(gg);                  \
    else if (qkv3) gemm_f16_glds_kernel<GEMM_OUT_QKV, MI_><<<grid1, 256, 0, s>>>(gg);                                    \
    else gemm_f16_glds_kernel<GEMM_OUT_QKV, MI_, 4><<<grid1, 256, 0, s>>>(gg);                                           \
  } while (0)
  if (mi == 5) TTS_LAUNCH_MI(5);
  else if (mi == 2) TTS_LAUNCH_MI(2);
  else TTS_LAUNCH_MI(4);
#undef TTS_LAUNCH_MI
  return hipGetLastError();
}

} // namespace tts

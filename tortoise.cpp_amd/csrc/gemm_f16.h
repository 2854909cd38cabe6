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
// Two kernels, both (16 h) x 128 x 64 tiles (h = 1..8 sixteen-row blocks, a launch parameter), 4 waves (2 x 2), operands staged
// by direct global->LDS DMA (global_load_lds_dwordx4) into a lane-linear image whose 16-byte chunks are
// XOR-swizzled by (row>>1)&7 on the SOURCE address, so ds_read_b128 fragment reads spread over all banks:
//   gemm_f16_vh_kernel        any segment structure; one LDS stage, 4 workgroups per CU overlap each other
//   gemm_f16_conv3_vh_kernel  the k=3 convolution: one activation slab shared by the three taps, weight tiles
//                             double-buffered (counted vmcnt + raw barriers)
// launch_gemm_f16 picks between them and chooses h. Round 3 rewrote both around two measurements
// (profiles/r3_gemm_tile_tables.txt, profiles/r3_gemm_epilogue.txt):
//  * tile-height / dispatch-order policies (tables of mixed heights, tallest-first, whole rounds filled exactly) change nothing:
//    a partially filled round runs proportionally faster, the launch is bound by total work, not by rounds;
//  * the 8-10 us "epilogue burst" of every tile was not HBM bandwidth but SIXTEEN SERIALISED memory round trips: a bias / guard /
//    residual load under a runtime `ptr ? load : 0` select is branched around by hipcc and waited for with vmcnt(0) — which also
//    waits for the previous block's stores (one in-order counter). Every epilogue load is now issued up front, unconditionally,
//    and the stores follow back to back.
// Measured-and-rejected variants: tools/ (gemm_f16_onetile.h = the round-2 kernels, gemm_f16_experiments.h, gemm_f16_big.h = the 256-column
// 8-phase kernel of round 3: one workgroup per CU, two wave groups in strict alternation — 94 vs 78 us on in_layers, 238 vs 204 on the QKV
// projection, per-phase timeline in profiles/r3_gemm_256col_kernel.txt).
#pragma once
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <type_traits>

namespace tts {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

// GEMM_OUT_QKV_SPLIT: the QKV projection of the reference-precision AttentionBlock (option attn_f32): every f32 result x is stored as the
// fp16 pair hi = fp16(x), lo = fp16(x - hi) (hi + lo = x to 2^-22) in outH / outH2 and outVt / outVt2.
// GEMM_OUT_F32_SCALED: out = alpha * acc + bias + resid (split-precision operands whose weights were scaled by 1 / alpha at load).
// GEMM_OUT_F32_STATS / GEMM_OUT_F32_SCALED_STATS (round 6, option latency_mode): the f32 output + the GroupNorm statistics of the stored values, accumulated per
// (sequence, 32-channel group) in fixed point (fx_add) — the GroupNorm that consumes the output needs no reduction pass of its own (diffusion.hip: gn_apply_kernel).
enum { GEMM_OUT_F32 = 0, GEMM_OUT_F16 = 1, GEMM_OUT_QKV = 2, GEMM_OUT_QKV_SPLIT = 3, GEMM_OUT_F32_SCALED = 4, GEMM_OUT_F32_STATS = 5, GEMM_OUT_F32_SCALED_STATS = 6 };
constexpr bool gemm_mode_qkv(int mode) { return mode == GEMM_OUT_QKV || mode == GEMM_OUT_QKV_SPLIT; }
constexpr bool gemm_mode_f32(int mode) { return mode == GEMM_OUT_F32 || mode == GEMM_OUT_F32_SCALED || mode == GEMM_OUT_F32_STATS || mode == GEMM_OUT_F32_SCALED_STATS; }
constexpr bool gemm_mode_scaled(int mode) { return mode == GEMM_OUT_F32_SCALED || mode == GEMM_OUT_F32_SCALED_STATS; }
constexpr bool gemm_mode_stats(int mode) { return mode == GEMM_OUT_F32_STATS || mode == GEMM_OUT_F32_SCALED_STATS; }

struct GemmArgs {
  const __half *A[3];  // per segment base (row 0 of the packed layout)
  int row_off[3];
  int nseg, kseg;      // kseg % 64 == 0
  int lda;             // halves
  const __half *W;     // [N][ldw]; segment seg starts at column w_off[seg] (defaults: ldw = nseg*kseg, w_off = seg*kseg)
  int ldw_, w_off_[3], custom_w; // set custom_w = 1 to use ldw_/w_off_ (e.g. split-precision: hi|lo halves reused)
  int M, N;            // multiples of 128 (buffers are padded)
  const float *bias;   // [N] or nullptr
  const int *row_seq;  // [M]: sequence id, <0 for guard/padding rows (output forced to 0); may be null
  // GEMM_OUT_F32
  float *outF; int ldo; const float *resid; // resid may alias outF
  // GEMM_OUT_F16 (n_valid columns written) / GEMM_OUT_QKV
  __half *outH; int ldh;
  __half *outVt; int ldvt; // QKV: V channels transposed [h*64+d][row]
  __half *outH2, *outVt2;  // GEMM_OUT_QKV_SPLIT: the low halves (same leading dimensions)
  float alpha;             // GEMM_OUT_F32_SCALED
  long long *st_out;       // GEMM_OUT_*_STATS: [FX_STRIPES][st_stripe_ll]: per stripe [sequences][32 groups][4] fixed-point {sum hi, sum lo, sum of squares hi, lo}
  int st_stripe_ll;        //                   (fx_add), accumulated into; a workgroup adds to stripe blockIdx % FX_STRIPES, the reader sums the stripes (exact)
  const int *chunk_seq;    //                   [M / 8]: owning sequence of an aligned 8-row chunk (sequences start at multiples of 8 rows), -1 = guard rows only
  int dual_b;              // 1: the two segments are the hi / lo halves of ONE weight over ONE activation operand -> gemm_f16_vh_dualb_kernel (GEMM_OUT_F32_SCALED only)
  int mode;
  // tile walk, set by launch_gemm_f16: th = 16-row blocks per tile (0: chosen from the problem size), cn = column tiles per L2 chunk
  int th, cn;
  int ku; // K tiles per barrier pair of gemm_f16_vh_kernel (0 / 1: one; 2, 4: small problems, set by launch_gemm_f16 or a tool)
  // developer tools only (tools/gemm_tab_bench.hip): explicit per-XCD tile lists [8][tab_len] of {first row, blocks, first column, 0}
  const int4 *tiles; int tab_len;
};


// LDS image of an operand tile: 128-byte rows (64 halves), the eight 16-byte chunks of row r stored at chunk ^ lds_swz(r). A ds_read_b128 is
// served in groups of 16 lanes = 16 CONSECUTIVE rows, eight of them reading chunk c (fq even) and eight chunk c + 1 (fq odd): the two sets differ in
// chunk bit 0, which the swizzle never touches, and inside a set the four rows of one parity have four different values of bits 1-2 — for ANY first
// row. (Round 2 xor-ed all three chunk bits with (row >> 1) & 7: conflict-free only for a first row that is a multiple of 4, i.e. not for the k = 3
// kernel's tap-shifted reads at rows +1 and +2: SQ_LDS_BANK_CONFLICT was 25 % of its LDS-active cycles, profiles/r3_pmc_lds.json.)
__device__ __forceinline__ int lds_swz(int row) { return ((row >> 1) & 3) << 1; }
__device__ __forceinline__ int lds_off(int row, int chunk) { return row * 128 + ((chunk ^ lds_swz(row)) << 4); }

// LDS stages are filled by direct global->LDS DMA (global_load_lds_dwordx4): no staging VGPRs, no ds_write pass.
// The LDS image of a DMA is lane-linear (base + lane*16), so the XOR swizzle is applied to the per-lane
// SOURCE address and again on the fragment read.
typedef const void __attribute__((address_space(1))) *gptr_t;
typedef void __attribute__((address_space(3))) *lptr_t;

// A wave of the 2 x 2 layout owns the 16-row blocks wm, wm + 2, wm + 4, wm + 6 of the tile (interleaved: a 7-block tile
// splits 4 / 3) and 64 columns: acc[i][j] = block wm + 2 i, columns wn * 64 + j * 16 ..
__device__ __forceinline__ int vh_blk(int wm, int i) { return wm + 2 * i; }

// low half of the split-precision pair of x: x - fp16(x), exact in f32, then rounded to fp16
__device__ __forceinline__ float split_lo(float x) { return x - __half2float(__float2half_rn(x)); }
__device__ __forceinline__ uint2 pack_half4(float a, float b, float c, float d) {
  const __half2 p0 = __floats2half2_rn(a, b), p1 = __floats2half2_rn(c, d);
  uint2 u;
  u.x = *(const unsigned *)&p0;
  u.y = *(const unsigned *)&p1;
  return u;
}

// Statistics accumulators are striped: one utterance sends ~440 atomics to each (sequence, group) record per GEMM, and same-line atomics are served one after the
// other by that line's L2 channel (~12 ns each: 5 us behind the kernel). 8 stripes and one set of atomics per wave instead of per chunk -> a dozen per line; the consumer adds the stripes (integers: still exact).
static constexpr int FX_STRIPES = 8;
// 128-bit fixed-point accumulation of an f32 partial sum: hi in units of 2^-8 (|p| < 2^55), the exact remainder in units of 2^-60. Integer addition is associative: the
// totals do not depend on the order in which workgroups arrive, and a per-chunk partial is computed by one wave in a fixed lane tree -> the statistics are reproducible
// run to run and independent of what else is in the batch.
__device__ __forceinline__ void fx_add(long long *dst, float p) {
  const float h = rintf(p * 256.0f);
  const float rem = p - h * (1.0f / 256.0f); // exact: |rem| <= 2^-9, or 0 when ulp(p) >= 2^-8
  atomicAdd((unsigned long long *)dst, (unsigned long long)(long long)h);
  atomicAdd((unsigned long long *)dst + 1, (unsigned long long)(long long)rintf(rem * 4503599627370496.0f)); // 2^52
}
__device__ __forceinline__ void fx_split(float p, long long &hi, long long &lo) {
  const float h = rintf(p * 256.0f);
  hi = (long long)h;
  lo = (long long)rintf((p - h * (1.0f / 256.0f)) * 4503599627370496.0f);
}
__host__ __device__ __forceinline__ double fx_value(long long hi, long long lo) { return (double)hi * (1.0 / 256.0) + (double)lo * (1.0 / 1152921504606846976.0); }
template <int CTRL> __device__ __forceinline__ float gemm_dpp(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, false));
}
// sum over the lanes that differ in bits 0-2 and 4-5 (the 8 rows of a half block x the 4 column quads of a swapped-order 16x16 accumulator): lanes 0 and 8 of the wave end
// up with the totals of rows 0-7 / 8-15 of the block. Fixed tree.
__device__ __forceinline__ float red_half_block(float x) {
  x += gemm_dpp<0xB1>(x);  // quad_perm [1,0,3,2]: lane ^ 1
  x += gemm_dpp<0x4E>(x);  // quad_perm [2,3,0,1]: lane ^ 2
  x += gemm_dpp<0x141>(x); // row_half_mirror: quads are uniform now, 7 - l swaps the two quads of a half row: lane ^ 4
  auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  x = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

// Epilogues. For F32/F16 outputs the MFMA operands are swapped (A-operand = weight rows, B-operand =
// activation rows), so a lane's 4 accumulator registers are 4 CONSECUTIVE output columns of one row:
// residual loads and stores are 16 bytes per lane instead of 4. The V columns of the QKV mode keep the natural order
// (a lane holds 4 consecutive rows of one column) because V is stored transposed.
// EVERY load (bias, guard flags, residual) is issued before the first store, inside at most one wave-uniform branch per
// operand kind: a load under a per-element `ptr ? .. : ..` select costs a vmcnt(0) round trip per element AND drains the stores
// issued before it (see the header of this file).
enum { EPI_NO_RESID = 0, EPI_RESID_IN_ACC = 1, EPI_RESID_LOAD = 2 };
template <int MODE, int MI, int RESID>
__device__ __forceinline__ void gemm_epilogue_vh(const GemmArgs &g, floatx4 (&acc)[MI > 0 ? MI : 1][4], int m0, int n0, int wm, int wn, int fr, int fq) {
  constexpr int MA = MI > 0 ? MI : 1;
  if (MI == 0) return;
  if (gemm_mode_qkv(MODE)) {
    // col = h*192 + {q 0..63 | k 64..127 | v 128..191}; a wave's 64-column span is entirely q, k or v.
    const int c0 = n0 + wn * 64, h = c0 / 192, w0 = c0 - h * 192;
    if (w0 >= 128) { // V, natural operand order: lane = 4 consecutive rows of one column -> 8-byte transposed store
      // unconditional loads + selects (a null pointer reads a valid dummy address and the value is dropped): ONE round trip
      const bool hb = g.bias != nullptr, hs = g.row_seq != nullptr;
      const float *bp = hb ? g.bias : (const float *)g.W;   // W holds >= 128 N bytes
      const int *sp = hs ? g.row_seq : (const int *)g.A[0]; // A holds >= 128 M bytes
      float bv[4];
      int4 sq[MA];
#pragma unroll
      for (int j = 0; j < 4; j++) bv[j] = bp[c0 + j * 16 + fr];
#pragma unroll
      for (int i = 0; i < MI; i++) sq[i] = *(const int4 *)(sp + m0 + vh_blk(wm, i) * 16 + fq * 4);
#pragma unroll
      for (int j = 0; j < 4; j++) bv[j] = hb ? bv[j] : 0.f;
#pragma unroll
      for (int i = 0; i < MI; i++) sq[i] = hs ? sq[i] : make_int4(0, 0, 0, 0);
#pragma unroll
      for (int i = 0; i < MI; i++) {
        const int rbase = m0 + vh_blk(wm, i) * 16 + fq * 4;
        const bool gd[4] = {sq[i].x < 0, sq[i].y < 0, sq[i].z < 0, sq[i].w < 0};
#pragma unroll
        for (int j = 0; j < 4; j++) {
          float v[4];
#pragma unroll
          for (int r = 0; r < 4; r++) v[r] = gd[r] ? 0.f : acc[i][j][r] + bv[j];
          *(uint2 *)(g.outVt + (size_t)(h * 64 + j * 16 + fr) * g.ldvt + rbase) = pack_half4(v[0], v[1], v[2], v[3]);
          if (MODE == GEMM_OUT_QKV_SPLIT)
            *(uint2 *)(g.outVt2 + (size_t)(h * 64 + j * 16 + fr) * g.ldvt + rbase) = pack_half4(split_lo(v[0]), split_lo(v[1]), split_lo(v[2]), split_lo(v[3]));
        }
      }
      return;
    }
  }
  // swapped operand order: acc[i][j][r] = C[m0 + blk(i)*16 + fr][n0 + wn*64 + j*16 + fq*4 + r]
  const int col0 = n0 + wn * 64 + fq * 4;
  const bool hb = g.bias != nullptr, hs = g.row_seq != nullptr;
  const float *bp = hb ? g.bias : (const float *)g.W;   // see above: unconditional loads, values selected afterwards
  const int *sp = hs ? g.row_seq : (const int *)g.A[0];
  float4 b4[4];
  int sq[MA];
  float4 rr[2][4]; // RESID == EPI_RESID_LOAD: residual of a pair of blocks
  auto load_pair = [&](int p) {
#pragma unroll
    for (int q = 0; q < 2; q++) {
      const int i = min(2 * p + q, MI - 1); // odd MI: the last pair re-reads its first block (never used)
      const float *rp = g.resid + (size_t)(m0 + vh_blk(wm, i) * 16 + fr) * g.ldo + col0;
#pragma unroll
      for (int j = 0; j < 4; j++) rr[q][j] = *(const float4 *)(rp + j * 16);
    }
  };
  if (RESID == EPI_RESID_LOAD) load_pair(0); // the HBM reads first, bias and guards (L2 hits) behind them: one round trip
#pragma unroll
  for (int j = 0; j < 4; j++) b4[j] = *(const float4 *)(bp + col0 + j * 16);
#pragma unroll
  for (int i = 0; i < MI; i++) sq[i] = sp[m0 + vh_blk(wm, i) * 16 + fr];
#pragma unroll
  for (int j = 0; j < 4; j++) b4[j] = hb ? b4[j] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int i = 0; i < MI; i++) sq[i] = hs ? sq[i] : 0;
  auto finish = [&](int i, int j) { // bias, guard
    if (gemm_mode_scaled(MODE)) acc[i][j] *= g.alpha;
    float4 v = make_float4(acc[i][j][0] + b4[j].x, acc[i][j][1] + b4[j].y, acc[i][j][2] + b4[j].z, acc[i][j][3] + b4[j].w);
    if (sq[i] < 0) v = make_float4(0.f, 0.f, 0.f, 0.f);
    return v;
  };
  if (gemm_mode_f32(MODE)) {
    // GroupNorm statistics of the stored values (GEMM_OUT_*_STATS): a lane's partial sums over its 2 x 4 values of a (block, 32-column group), reduced over the 8 rows of
    // each half block after the stores
    float ssum[MA][2], ssq[MA][2];
    auto stat_add = [&](int i, int j, const float4 &v) {
      const float a = (v.x + v.y) + (v.z + v.w), b = (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
      if (j & 1) { ssum[i][j >> 1] += a; ssq[i][j >> 1] += b; }
      else { ssum[i][j >> 1] = a; ssq[i][j >> 1] = b; }
    };
    auto stat_flush = [&]() {
      // Every 8-row chunk's partial is converted to fixed point on its own (lanes 0 and 8 of the wave) — the unit whose value does not depend on the tiling — and
      // the integers of a lane's chunks that belong to one sequence are added up before the atomics: 4 atomics per lane and group instead of 4 per chunk.
      const int lane = fq * 16 + fr;
      int cs[MA];
#pragma unroll
      for (int i = 0; i < MI; i++) cs[i] = g.chunk_seq[((m0 + vh_blk(wm, i) * 16) >> 3) + (fr >> 3)];
#pragma unroll
      for (int jp = 0; jp < 2; jp++) {
        long long tot[4] = {0, 0, 0, 0};
        int tseq = -1;
        auto flush = [&]() {
          if (tseq >= 0) {
            long long *dst = g.st_out + (size_t)(blockIdx.x % FX_STRIPES) * g.st_stripe_ll + (size_t)(tseq * 32 + ((n0 + wn * 64) >> 5) + jp) * 4;
#pragma unroll
            for (int k = 0; k < 4; k++) atomicAdd((unsigned long long *)dst + k, (unsigned long long)tot[k]);
          }
        };
#pragma unroll
        for (int i = 0; i < MI; i++) { // lane 0 adds up the upper halves (rows 0-7) of the wave's blocks, lane 8 the lower ones
          const float sv = red_half_block(ssum[i][jp]), qv = red_half_block(ssq[i][jp]);
          if ((lane & 0x37) == 0 && cs[i] >= 0) {
            long long f[4];
            fx_split(sv, f[0], f[1]);
            fx_split(qv, f[2], f[3]);
            if (cs[i] != tseq) { flush(); tseq = cs[i]; tot[0] = tot[1] = tot[2] = tot[3] = 0; }
#pragma unroll
            for (int k = 0; k < 4; k++) tot[k] += f[k];
          }
        }
        if ((lane & 0x37) == 0) flush();
      }
    };
    if (RESID == EPI_RESID_LOAD) {
      // residual read here (k = 3 kernel): blocks in pairs, the next pair's 8 loads are in flight while this pair is stored
      constexpr int NP = (MI + 1) / 2;
#pragma unroll
      for (int p = 0; p < NP; p++) {
        float4 v[2][4];
#pragma unroll
        for (int q = 0; q < 2; q++) {
          const int i = 2 * p + q;
          if (i < MI) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
              // same f32 adds in the same order as before: (acc + bias) + resid
              if (gemm_mode_scaled(MODE)) acc[i][j] *= g.alpha;
              float4 t = make_float4(acc[i][j][0] + b4[j].x, acc[i][j][1] + b4[j].y, acc[i][j][2] + b4[j].z, acc[i][j][3] + b4[j].w);
              t.x += rr[q][j].x; t.y += rr[q][j].y; t.z += rr[q][j].z; t.w += rr[q][j].w;
              if (sq[i] < 0) t = make_float4(0.f, 0.f, 0.f, 0.f);
              v[q][j] = t;
              if (gemm_mode_stats(MODE)) stat_add(i, j, t);
            }
          }
        }
        if (p + 1 < NP) load_pair(p + 1);
#pragma unroll
        for (int q = 0; q < 2; q++) {
          const int i = 2 * p + q;
          if (i < MI) {
            float *op = g.outF + (size_t)(m0 + vh_blk(wm, i) * 16 + fr) * g.ldo + col0;
#pragma unroll
            for (int j = 0; j < 4; j++) *(float4 *)(op + j * 16) = v[q][j];
          }
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < MI; i++) {
        float *op = g.outF + (size_t)(m0 + vh_blk(wm, i) * 16 + fr) * g.ldo + col0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const float4 v = finish(i, j);
          *(float4 *)(op + j * 16) = v;
          if (gemm_mode_stats(MODE)) stat_add(i, j, v);
        }
      }
    }
    if (gemm_mode_stats(MODE)) stat_flush();
  } else if (MODE == GEMM_OUT_F16) {
#pragma unroll
    for (int i = 0; i < MI; i++) {
      __half *op = g.outH + (size_t)(m0 + vh_blk(wm, i) * 16 + fr) * g.ldh + col0;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const float4 v = finish(i, j);
        *(uint2 *)(op + j * 16) = pack_half4(v.x, v.y, v.z, v.w);
      }
    }
  } else { // Q or K columns of the QKV projection
    const int c0 = n0 + wn * 64, h = c0 / 192, w0 = c0 - h * 192;
#pragma unroll
    for (int i = 0; i < MI; i++) {
      __half *op = g.outH + (size_t)(m0 + vh_blk(wm, i) * 16 + fr) * g.ldh + h * 128 + w0 + fq * 4;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const float4 v = finish(i, j);
        *(uint2 *)(op + j * 16) = pack_half4(v.x, v.y, v.z, v.w);
        if (MODE == GEMM_OUT_QKV_SPLIT)
          *(uint2 *)(g.outH2 + (op - g.outH) + j * 16) = pack_half4(split_lo(v.x), split_lo(v.y), split_lo(v.z), split_lo(v.w));
      }
    }
  }
}

// Any segment structure (k = 1 convolutions, projections, channel concats, split-precision operands): one 32 KB LDS stage filled by
// LDS-DMA, 4 workgroups per CU overlap each other's load / compute phases.
// The body is instantiated per number of 16-row blocks of the calling WAVE (0..4): the waves of a workgroup may run different
// instantiations; all of them issue the same DMA pieces and pass the same two barriers per K tile.
// KU (round 6): K tiles per barrier pair. KU = 2 stages two 64-deep tiles side by side (64 KB, 2 workgroups per CU) and multiplies them between ONE pair of barriers: a
// small problem (one utterance: at most two workgroups per CU) pays a DMA round trip + two barriers per pair instead of per tile. Same products in the same order:
// bit-identical to KU = 1.
template <int MODE, int MI, int KU = 1>
__device__ __forceinline__ void gemm_vh_body(const GemmArgs &g, int m0, int n0, int nblk, int lane, int wave) {
  extern __shared__ __attribute__((aligned(16))) char smem_dyn[]; // named here: an LDS pointer passed in would become a generic pointer
  char *smem = smem_dyn;
  constexpr int MA = MI > 0 ? MI : 1;
  const int wm = wave >> 1, wn = wave & 1;
  const int my_pa = (2 * nblk - wave + 3) >> 2; // 8-row DMA pieces wave, wave + 4, .. of the A tile moved by this wave
  const int tiles_per_seg = g.kseg >> 6;
  const int ldw = g.custom_w ? g.ldw_ : g.nseg * g.kseg;
  const int prow = lane >> 3, pslot = lane & 7;
  int aoff[4], boff[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int row = (wave + 4 * i) * 8 + prow;
    aoff[i] = (m0 + row) * g.lda + (pslot ^ lds_swz(row)) * 8; // pieces past the tile are never issued
  }
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int row = (wave * 4 + i) * 8 + prow;
    boff[i] = (n0 + row) * ldw + (pslot ^ lds_swz(row)) * 8;
  }
  const int fr = lane & 15, fq = lane >> 4;
  // Accumulators START from the residual (F32 outputs, swapped operand order: acc[i][j] = 4 consecutive columns of one row): the
  // residual read is issued with the first operand tile and hides behind it, instead of being a dependent HBM round trip in front
  // of the stores when the K loop is over. The sum is the same set of f32 adds in a different order.
  // GEMM_OUT_F32_SCALED (out = alpha acc + bias + resid, alpha a power of two): the accumulators start from resid / alpha — exact, and scaled back
  // exactly by the epilogue.
  const bool resid_first = gemm_mode_f32(MODE) && g.resid != nullptr;
  floatx4 acc[MA][4];
  if (resid_first) {
    const float rs = gemm_mode_scaled(MODE) ? 1.0f / g.alpha : 1.0f;
#pragma unroll
    for (int i = 0; i < MI; i++) {
      const int row = m0 + vh_blk(wm, i) * 16 + fr;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const float4 rr = *(const float4 *)(g.resid + (size_t)row * g.ldo + n0 + wn * 64 + j * 16 + fq * 4);
        acc[i][j] = (floatx4){rr.x * rs, rr.y * rs, rr.z * rs, rr.w * rs};
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < MI; i++)
#pragma unroll
      for (int j = 0; j < 4; j++) acc[i][j] = (floatx4){0.f, 0.f, 0.f, 0.f};
  }
  char *sa = smem, *sb = smem + 16384;
  // operand order (see gemm_epilogue_vh): natural only for the V columns of a QKV projection (wave-uniform)
  const bool natural = gemm_mode_qkv(MODE) && (((n0 + wn * 64) % 192) >= 128);
  // the K loop is instantiated once per operand order so the choice costs nothing inside it
  auto kloop = [&](auto nat) {
    constexpr bool NAT = decltype(nat)::value;
    // segment loop outside, K tiles inside: the segment's base pointers are fetched from the kernel arguments once
    for (int seg = 0; seg < g.nseg; seg++) {
      const __half *aseg = g.A[seg] + (ptrdiff_t)g.row_off[seg] * g.lda;
      const __half *wseg = g.W + (g.custom_w ? g.w_off_[seg] : seg * g.kseg);
      for (int kt = 0; kt < tiles_per_seg; kt += KU) { // kseg % (64 KU) == 0 (checked by the launcher)
#pragma unroll
        for (int u = 0; u < KU; u++) {
          const __half *abase = aseg + ((kt + u) << 6), *wbase = wseg + ((kt + u) << 6);
#pragma unroll
          for (int i = 0; i < 4; i++)
            if (i < my_pa) __builtin_amdgcn_global_load_lds((gptr_t)(abase + aoff[i]), (lptr_t)(sa + u * 32768 + (wave + 4 * i) * 1024), 16, 0, 0);
#pragma unroll
          for (int i = 0; i < 4; i++)
            __builtin_amdgcn_global_load_lds((gptr_t)(wbase + boff[i]), (lptr_t)(sb + u * 32768 + (wave * 4 + i) * 1024), 16, 0, 0);
        }
        __syncthreads(); // waits vmcnt(0) for the DMA, then barrier
#pragma unroll
        for (int ks = 0; ks < 2 * KU; ks++) {
          const char *sau = sa + (ks >> 1) * 32768, *sbu = sb + (ks >> 1) * 32768;
          half8 af[MA], bf[4];
#pragma unroll
          for (int i = 0; i < MI; i++) af[i] = *(const half8 *)(sau + lds_off(vh_blk(wm, i) * 16 + fr, (ks & 1) * 4 + fq));
          if (MI > 0) {
#pragma unroll
            for (int i = 0; i < 4; i++) bf[i] = *(const half8 *)(sbu + lds_off(wn * 64 + i * 16 + fr, (ks & 1) * 4 + fq));
          }
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
  if (gemm_mode_qkv(MODE) && natural) kloop(std::true_type{});
  else kloop(std::false_type{});
  if (resid_first) gemm_epilogue_vh<MODE, MI, EPI_RESID_IN_ACC>(g, acc, m0, n0, wm, wn, fr, fq);
  else gemm_epilogue_vh<MODE, MI, EPI_NO_RESID>(g, acc, m0, n0, wm, wn, fr, fq);
}

// Split-precision WEIGHT, one activation operand (round 5: proj_out of the default AttentionBlock, out = A (W_hi + W_lo)^T): the two weight tiles of a K
// tile are staged side by side (16 KB A + 2 x 16 KB B = 48 KB, 3 workgroups per CU) and every A fragment read from LDS feeds both products — against two
// K segments through gemm_vh_body: 48 instead of 64 KB of DMA, 24 instead of 32 KB of fragment reads and 2 instead of 4 barriers per 64 MFMAs.
// g.W = [N][ldw_] with the hi half at column w_off_[0] and the lo half at w_off_[1] (custom_w); g.nseg == 2, g.A[0] == g.A[1], equal row offsets.
static constexpr int GEMM_DUALB_LDS = 49152;
template <int MODE, int MI, int KU = 1>
__device__ __forceinline__ void gemm_vh_dualb_body(const GemmArgs &g, int m0, int n0, int nblk, int lane, int wave) {
  extern __shared__ __attribute__((aligned(16))) char smem_dyn[];
  char *smem = smem_dyn;
  constexpr int MA = MI > 0 ? MI : 1;
  const int wm = wave >> 1, wn = wave & 1;
  const int my_pa = (2 * nblk - wave + 3) >> 2;
  const int ntiles = g.kseg >> 6, ldw = g.ldw_;
  const int prow = lane >> 3, pslot = lane & 7;
  int aoff[4], boff[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int row = (wave + 4 * i) * 8 + prow;
    aoff[i] = (m0 + row) * g.lda + (pslot ^ lds_swz(row)) * 8;
  }
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int row = (wave * 4 + i) * 8 + prow;
    boff[i] = (n0 + row) * ldw + (pslot ^ lds_swz(row)) * 8;
  }
  const int fr = lane & 15, fq = lane >> 4;
  const bool resid_first = gemm_mode_f32(MODE) && g.resid != nullptr;
  floatx4 acc[MA][4];
  if (resid_first) {
    const float rs = gemm_mode_scaled(MODE) ? 1.0f / g.alpha : 1.0f;
#pragma unroll
    for (int i = 0; i < MI; i++) {
      const int row = m0 + vh_blk(wm, i) * 16 + fr;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const float4 rr = *(const float4 *)(g.resid + (size_t)row * g.ldo + n0 + wn * 64 + j * 16 + fq * 4);
        acc[i][j] = (floatx4){rr.x * rs, rr.y * rs, rr.z * rs, rr.w * rs};
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < MI; i++)
#pragma unroll
      for (int j = 0; j < 4; j++) acc[i][j] = (floatx4){0.f, 0.f, 0.f, 0.f};
  }
  char *sa = smem, *sb0 = smem + 16384, *sb1 = smem + 32768;
  const __half *aseg = g.A[0] + (ptrdiff_t)g.row_off[0] * g.lda;
  const __half *w0 = g.W + g.w_off_[0], *w1 = g.W + g.w_off_[1];
  for (int kt = 0; kt < ntiles; kt += KU) { // KU K tiles per barrier pair (see gemm_vh_body): 48 KB of LDS each
#pragma unroll
    for (int u = 0; u < KU; u++) {
      const __half *abase = aseg + ((kt + u) << 6), *b0 = w0 + ((kt + u) << 6), *b1 = w1 + ((kt + u) << 6);
#pragma unroll
      for (int i = 0; i < 4; i++)
        if (i < my_pa) __builtin_amdgcn_global_load_lds((gptr_t)(abase + aoff[i]), (lptr_t)(sa + u * GEMM_DUALB_LDS + (wave + 4 * i) * 1024), 16, 0, 0);
#pragma unroll
      for (int i = 0; i < 4; i++) __builtin_amdgcn_global_load_lds((gptr_t)(b0 + boff[i]), (lptr_t)(sb0 + u * GEMM_DUALB_LDS + (wave * 4 + i) * 1024), 16, 0, 0);
#pragma unroll
      for (int i = 0; i < 4; i++) __builtin_amdgcn_global_load_lds((gptr_t)(b1 + boff[i]), (lptr_t)(sb1 + u * GEMM_DUALB_LDS + (wave * 4 + i) * 1024), 16, 0, 0);
    }
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < 2 * KU; ks++) {
      const int uo = (ks >> 1) * GEMM_DUALB_LDS, kk = ks & 1;
      half8 af[MA], bf[4], bl[4];
#pragma unroll
      for (int i = 0; i < MI; i++) af[i] = *(const half8 *)(sa + uo + lds_off(vh_blk(wm, i) * 16 + fr, kk * 4 + fq));
      if (MI > 0) {
#pragma unroll
        for (int i = 0; i < 4; i++) bf[i] = *(const half8 *)(sb0 + uo + lds_off(wn * 64 + i * 16 + fr, kk * 4 + fq));
#pragma unroll
        for (int i = 0; i < 4; i++) bl[i] = *(const half8 *)(sb1 + uo + lds_off(wn * 64 + i * 16 + fr, kk * 4 + fq));
      }
#pragma unroll
      for (int i = 0; i < MI; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl[j], af[i], acc[i][j], 0, 0, 0); // the small term first
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[j], af[i], acc[i][j], 0, 0, 0);
        }
    }
    __syncthreads();
  }
  if (resid_first) gemm_epilogue_vh<MODE, MI, EPI_RESID_IN_ACC>(g, acc, m0, n0, wm, wn, fr, fq);
  else gemm_epilogue_vh<MODE, MI, EPI_NO_RESID>(g, acc, m0, n0, wm, wn, fr, fq);
}

// Tile walk, shared by both kernels. L2-aware: workgroup b runs on XCD b % 8 (each XCD has its own 4 MB L2). An XCD owns a
// contiguous range of 16-row blocks, cut into tiles of g.th blocks (the last one shorter), and walks them once per chunk of g.cn
// column tiles, chunk outermost: the chunk's weight rows (cn * 128 * K * 2 B <= ~2.5 MB) stay L2-resident while the activations
// stream through. (With plain m-major order the 6 MB QKV weight thrashed L2: 417 MB fetched per launch for 64 MB of operands.)
__device__ __forceinline__ bool gemm_vh_tile(const GemmArgs &g, int &m0, int &n0, int &nblk) {
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  if (g.tiles) { // developer tools: explicit per-XCD lists
    const int4 td = g.tiles[xcd * g.tab_len + idx];
    m0 = __builtin_amdgcn_readfirstlane(td.x); nblk = __builtin_amdgcn_readfirstlane(td.y); n0 = __builtin_amdgcn_readfirstlane(td.z);
    return nblk > 0;
  }
  const int nb = g.M >> 4, NT = g.N >> 7;
  const int b0 = (int)((long long)nb * xcd >> 3), b1 = (int)((long long)nb * (xcd + 1) >> 3);
  const int mt = (b1 - b0 + g.th - 1) / g.th;
  if (idx >= mt * NT) return false; // grid is padded to 8 * max tiles per XCD
  const int per_chunk = mt * g.cn, chunk = idx / per_chunk, rem = idx - chunk * per_chunk;
  const int t = rem / g.cn, blk0 = b0 + t * g.th;
  m0 = blk0 << 4;
  nblk = min(g.th, b1 - blk0);
  n0 = (chunk * g.cn + rem - t * g.cn) << 7;
  return true;
}

static constexpr int GEMM_VH_LDS = 32768;
template <int MODE, int WGS, int KU = 1>
static __global__ __launch_bounds__(256, WGS) void gemm_f16_vh_kernel(GemmArgs g) {
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6); // scalar: LDS-DMA bases stay in SGPRs
  int m0, n0, nblk;
  if (!gemm_vh_tile(g, m0, n0, nblk)) return;
  const int my_mi = (nblk - (wave >> 1) + 1) >> 1; // 16-row blocks of this wave
  if (my_mi == 4) gemm_vh_body<MODE, 4, KU>(g, m0, n0, nblk, lane, wave);
  else if (my_mi == 3) gemm_vh_body<MODE, 3, KU>(g, m0, n0, nblk, lane, wave);
  else if (my_mi == 2) gemm_vh_body<MODE, 2, KU>(g, m0, n0, nblk, lane, wave);
  else if (my_mi == 1) gemm_vh_body<MODE, 1, KU>(g, m0, n0, nblk, lane, wave);
  else gemm_vh_body<MODE, 0, KU>(g, m0, n0, nblk, lane, wave); // 1-block tile: this wave only moves operands
}

template <int MODE, int KU = 1>
static __global__ __launch_bounds__(256, KU == 1 ? 3 : 1) void gemm_f16_vh_dualb_kernel(GemmArgs g) {
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int m0, n0, nblk;
  if (!gemm_vh_tile(g, m0, n0, nblk)) return;
  const int my_mi = (nblk - (wave >> 1) + 1) >> 1;
  if (my_mi == 4) gemm_vh_dualb_body<MODE, 4, KU>(g, m0, n0, nblk, lane, wave);
  else if (my_mi == 3) gemm_vh_dualb_body<MODE, 3, KU>(g, m0, n0, nblk, lane, wave);
  else if (my_mi == 2) gemm_vh_dualb_body<MODE, 2, KU>(g, m0, n0, nblk, lane, wave);
  else if (my_mi == 1) gemm_vh_dualb_body<MODE, 1, KU>(g, m0, n0, nblk, lane, wave);
  else gemm_vh_dualb_body<MODE, 0, KU>(g, m0, n0, nblk, lane, wave);
}

// k = 3 convolution as ONE GEMM with a shared activation slab. The three taps are three row-shifted GEMM
// segments over the SAME activation rows (row_off = -1, 0, +1), so per 64-channel chunk the kernel stages the
// 16 nblk + 2 activation rows m0-1 .. once and multiplies them three times, each time against that tap's weight
// tile and read from LDS one row further down. Operand traffic through the CU's load path per chunk:
// 17 + 3 x 16 = 65 DMA pieces instead of 3 x 32 = 96 — the 128^2 tile is bound by exactly that path
// (64 B/clk/CU feeds at most one 32 KB K tile per 512 MFMA cycles).
// The weight tiles alternate between two LDS buffers: tap p+1's tile is requested before tap p's is waited for
// (counted vmcnt + raw barrier), so only the slab load at the start of a chunk is exposed.
// LDS: 17 KB slab + 2 x 16 KB -> 3 workgroups per CU.
static constexpr int CONV3_VH_LDS = (128 + 8) * 128 + 2 * 16384;
template <int MODE, int MI>
__device__ __forceinline__ void gemm_conv3_vh_body(const GemmArgs &g, int m0, int n0, int nblk, int lane, int wave) {
  extern __shared__ __attribute__((aligned(16))) char smem_dyn[];
  char *smem = smem_dyn;
  constexpr int MA = MI > 0 ? MI : 1;
  const int wm = wave >> 1, wn = wave & 1;
  const int my_pa = (2 * nblk + 1 - wave + 3) >> 2; // slab = 2 nblk + 1 pieces of 8 rows (the last one carries row m0 + 16 nblk)
  const int nchunks = g.kseg >> 6, ldw = 3 * g.kseg, nph = 3 * nchunks;
  const int prow = lane >> 3, pslot = lane & 7;
  const int fr = lane & 15, fq = lane >> 4;
  floatx4 acc[MA][4];
#pragma unroll
  for (int i = 0; i < MI; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = (floatx4){0.f, 0.f, 0.f, 0.f};
  char *sa = smem, *sb = smem + (128 + 8) * 128;
  const __half *abase = g.A[0] + (ptrdiff_t)(m0 - 1) * g.lda; // slab row s = activation row m0 - 1 + s (the buffer has its guard rows)
  const __half *wbase = g.W + (size_t)n0 * ldw;
  int aoff[5], boff[4];
#pragma unroll
  for (int i = 0; i < 5; i++) {
    const int row = (wave + 4 * i) * 8 + prow;
    aoff[i] = min(row, g.M - m0 + 1) * g.lda + (pslot ^ lds_swz(row)) * 8; // rows past the buffer end are never multiplied: clamp
  }
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int row = (wave * 4 + i) * 8 + prow;
    boff[i] = row * ldw + (pslot ^ lds_swz(row)) * 8;
  }
  auto stageA = [&](int kc) {
    const __half *src = abase + (min(kc, nchunks - 1) << 6);
#pragma unroll
    for (int i = 0; i < 5; i++)
      if (i < my_pa) __builtin_amdgcn_global_load_lds((gptr_t)(src + aoff[i]), (lptr_t)(sa + (wave + 4 * i) * 1024), 16, 0, 0);
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
    const char *sbp = sb + (p & 1) * 16384;
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {
      half8 af[MA], bf[4];
#pragma unroll
      for (int i = 0; i < MI; i++) af[i] = *(const half8 *)(sa + lds_off(vh_blk(wm, i) * 16 + fr + TAP, ks * 4 + fq));
      if (MI > 0) {
#pragma unroll
        for (int i = 0; i < 4; i++) bf[i] = *(const half8 *)(sbp + lds_off(wn * 64 + i * 16 + fr, ks * 4 + fq));
      }
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
  for (int kc = 0; kc < nchunks; kc++) {
    const int p = 3 * kc;
    phase(kc, p, std::integral_constant<int, 0>{});
    phase(kc, p + 1, std::integral_constant<int, 1>{});
    phase(kc, p + 2, std::integral_constant<int, 2>{});
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // trailing (clamped) pieces must land before the LDS is released
  if ((MODE == GEMM_OUT_F32 || MODE == GEMM_OUT_F32_STATS) && g.resid) gemm_epilogue_vh<MODE, MI, EPI_RESID_LOAD>(g, acc, m0, n0, wm, wn, fr, fq);
  else gemm_epilogue_vh<MODE, MI, EPI_NO_RESID>(g, acc, m0, n0, wm, wn, fr, fq);
}

template <int MODE>
static __global__ __launch_bounds__(256, 3) void gemm_f16_conv3_vh_kernel(GemmArgs g) {
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int m0, n0, nblk;
  if (!gemm_vh_tile(g, m0, n0, nblk)) return;
  const int my_mi = (nblk - (wave >> 1) + 1) >> 1;
  if (my_mi == 4) gemm_conv3_vh_body<MODE, 4>(g, m0, n0, nblk, lane, wave);
  else if (my_mi == 3) gemm_conv3_vh_body<MODE, 3>(g, m0, n0, nblk, lane, wave);
  else if (my_mi == 2) gemm_conv3_vh_body<MODE, 2>(g, m0, n0, nblk, lane, wave);
  else if (my_mi == 1) gemm_conv3_vh_body<MODE, 1>(g, m0, n0, nblk, lane, wave);
  else gemm_conv3_vh_body<MODE, 0>(g, m0, n0, nblk, lane, wave);
}

// k = 3 convolution (three row-shifted segments of one activation buffer, tap-major weights): shared-slab kernel
static inline bool gemm_is_conv3(const GemmArgs &g) {
  return g.nseg == 3 && !g.custom_w && g.A[0] == g.A[1] && g.A[1] == g.A[2] && g.row_off[0] == -1 && g.row_off[1] == 0 && g.row_off[2] == 1 &&
         !gemm_mode_qkv(g.mode) && !gemm_mode_scaled(g.mode);
}

static inline hipError_t launch_gemm_f16(const GemmArgs &g, hipStream_t s) {
  GemmArgs gg = g;
  const int NT = g.N >> 7, ktot = g.nseg * g.kseg, nb = g.M >> 4;
  if (gemm_mode_scaled(g.mode)) { // the accumulators start from resid / alpha and are scaled back by alpha: exact only for a non-zero power of two
    int e;
    if (!(g.alpha != 0.f) || std::fabs(std::frexp(g.alpha, &e)) != 0.5f) return hipErrorInvalidValue;
  }
  if (gemm_mode_stats(g.mode) && (!g.st_out || !g.chunk_seq || g.st_stripe_ll <= 0)) return hipErrorInvalidValue;
  // dual_b names ONE kernel: a caller whose arguments do not describe "hi | lo halves of one weight over one activation operand" gets an error, not another kernel
  if (g.dual_b && !(gemm_mode_scaled(g.mode) && g.nseg == 2 && g.custom_w && g.A[0] == g.A[1] && g.row_off[0] == g.row_off[1])) return hipErrorInvalidValue;
  int ku = (g.ku == 2 || g.ku == 4) && (g.kseg % (64 * g.ku)) == 0 ? g.ku : 1;
  int grid;
  if (g.tiles) grid = 8 * g.tab_len;
  else {
    // L2 chunking of the column tiles: pays for wide outputs (N = 3072: 200 us with chunks of 8 column tiles against 232 unchunked and 207-219
    // with chunks of 6) and costs ~3 % when the activations would have to stream twice for a narrow one (N = 1024, K = 3072)
    // -> only chunk when NT > 8; the chunk is the largest divisor of NT whose weight rows fit ~2.5 MB
    int cn = NT;
    if (NT > 8)
      for (cn = NT; cn > 1; cn--)
        if (NT % cn == 0 && (size_t)cn * 128 * ktot * 2 <= (size_t)2560 * 1024) break;
    gg.cn = cn;
    // Tile height: 128 rows unless the problem is small. Mixed heights, exactly filled rounds and tallest-first orders were measured
    // through explicit tile tables (tools/gemm_tab_bench.hip, profiles/r3_gemm_tile_tables.txt): no policy beats uniform 128-row
    // tiles at the benchmark's sizes. Small problems (a single utterance: M = 1 792 rows) want more, shorter tiles — halve the height
    // while the grid stays within one round of the ~1024 resident slots.
    const int maxb = (nb + 7) / 8 + 1; // blocks of the largest XCD range (upper bound)
    auto tiles_at = [&](int h) { return 8 * ((maxb + h - 1) / h) * NT; };
    int th = g.th;
    if (th <= 0) { th = 2; while (th < 8 && tiles_at(th) > 1024) th *= 2; }
    // One utterance (M = 1 792 rows, N = 1 024: 224 tiles of 64 rows = at most one workgroup per CU): four K tiles per barrier pair at 64-row tiles — a lone
    // workgroup pays its DMA round trip and two barriers per 256 of K instead of per 64 (profiles/r6_small_gemm.txt: k = 1 12.6 -> 11.6 us warm, 22.8 -> 17.9 us
    // with cold weights; the K = 2 048 integrating conv 25.7 / 36.2 -> 20.3 / 27.8). Same products in the same order: bit-identical to every other tiling.
    // (the split-weight proj_out, 48 KB per K tile: two per barrier pair, 21.5 -> 19.1 us warm / 28.3 -> 25.6 cold; the k = 3 kernel has its own pipeline and only takes
    // the 64-row tiles: 27.3 -> 25.5 / 35.6 -> 32.2)
    if (g.th <= 0 && g.ku == 0 && NT <= 8 && tiles_at(4) <= 256 && (g.kseg % 256) == 0) {
      th = 4;
      if (!gemm_is_conv3(g)) ku = g.dual_b ? 2 : 4;
    }
    gg.th = th;
    int mt_max = 0;
    for (int x = 0; x < 8; x++) {
      const int b0 = (int)((long long)nb * x >> 3), b1 = (int)((long long)nb * (x + 1) >> 3);
      mt_max = std::max(mt_max, (b1 - b0 + th - 1) / th);
    }
    grid = 8 * mt_max * NT;
  }
  if (gemm_is_conv3(g)) {
    static bool attr = false;
    if (!attr) {
      (void)hipFuncSetAttribute((const void *)gemm_f16_conv3_vh_kernel<GEMM_OUT_F32>, hipFuncAttributeMaxDynamicSharedMemorySize, CONV3_VH_LDS);
      (void)hipFuncSetAttribute((const void *)gemm_f16_conv3_vh_kernel<GEMM_OUT_F32_STATS>, hipFuncAttributeMaxDynamicSharedMemorySize, CONV3_VH_LDS);
      (void)hipFuncSetAttribute((const void *)gemm_f16_conv3_vh_kernel<GEMM_OUT_F16>, hipFuncAttributeMaxDynamicSharedMemorySize, CONV3_VH_LDS);
      attr = true;
    }
    if (g.mode == GEMM_OUT_F32) gemm_f16_conv3_vh_kernel<GEMM_OUT_F32><<<grid, 256, CONV3_VH_LDS, s>>>(gg);
    else if (g.mode == GEMM_OUT_F32_STATS) gemm_f16_conv3_vh_kernel<GEMM_OUT_F32_STATS><<<grid, 256, CONV3_VH_LDS, s>>>(gg);
    else gemm_f16_conv3_vh_kernel<GEMM_OUT_F16><<<grid, 256, CONV3_VH_LDS, s>>>(gg);
  } else if (g.dual_b && gemm_mode_scaled(g.mode) && g.nseg == 2 && g.custom_w && g.A[0] == g.A[1] && g.row_off[0] == g.row_off[1]) {
    // ku: 2 or 3 K tiles per barrier pair for a grid of at most one workgroup per CU (96 / 144 KB of LDS)
#define TTS_DUALB_LAUNCH(MODE_)                                                                                                                     \
  do {                                                                                                                                              \
    if (ku == 2) {                                                                                                                                  \
      static bool a2 = false;                                                                                                                       \
      if (!a2) { (void)hipFuncSetAttribute((const void *)gemm_f16_vh_dualb_kernel<MODE_, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * GEMM_DUALB_LDS); a2 = true; } \
      gemm_f16_vh_dualb_kernel<MODE_, 2><<<grid, 256, 2 * GEMM_DUALB_LDS, s>>>(gg);                                                                  \
    } else gemm_f16_vh_dualb_kernel<MODE_, 1><<<grid, 256, GEMM_DUALB_LDS, s>>>(gg);                                                                \
  } while (0)
    if (g.mode == GEMM_OUT_F32_SCALED) TTS_DUALB_LAUNCH(GEMM_OUT_F32_SCALED);
    else TTS_DUALB_LAUNCH(GEMM_OUT_F32_SCALED_STATS);
#undef TTS_DUALB_LAUNCH
  } else {
    // KU > 1: 64 / 128 KB of LDS -> 2 / 1 workgroups per CU (the occupancy bound of the launch is a compile-time promise: WGS)
#define TTS_VH_LAUNCH(MODE_)                                                                                                                      \
  do {                                                                                                                                            \
    if (ku == 1) gemm_f16_vh_kernel<MODE_, 4><<<grid, 256, GEMM_VH_LDS, s>>>(gg);                                                                 \
    else if (ku == 2) {                                                                                                                           \
      static bool a2 = false;                                                                                                                     \
      if (!a2) { (void)hipFuncSetAttribute((const void *)gemm_f16_vh_kernel<MODE_, 2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * GEMM_VH_LDS); a2 = true; } \
      gemm_f16_vh_kernel<MODE_, 2, 2><<<grid, 256, 2 * GEMM_VH_LDS, s>>>(gg);                                                                     \
    } else {                                                                                                                                      \
      static bool a4 = false;                                                                                                                     \
      if (!a4) { (void)hipFuncSetAttribute((const void *)gemm_f16_vh_kernel<MODE_, 1, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * GEMM_VH_LDS); a4 = true; } \
      gemm_f16_vh_kernel<MODE_, 1, 4><<<grid, 256, 4 * GEMM_VH_LDS, s>>>(gg);                                                                     \
    }                                                                                                                                             \
  } while (0)
    if (g.mode == GEMM_OUT_F32) TTS_VH_LAUNCH(GEMM_OUT_F32);
    else if (g.mode == GEMM_OUT_F16) TTS_VH_LAUNCH(GEMM_OUT_F16);
    else if (g.mode == GEMM_OUT_QKV) TTS_VH_LAUNCH(GEMM_OUT_QKV);
    else if (g.mode == GEMM_OUT_QKV_SPLIT) TTS_VH_LAUNCH(GEMM_OUT_QKV_SPLIT);
    else if (g.mode == GEMM_OUT_F32_SCALED) TTS_VH_LAUNCH(GEMM_OUT_F32_SCALED);
    else if (g.mode == GEMM_OUT_F32_STATS) TTS_VH_LAUNCH(GEMM_OUT_F32_STATS);
    else TTS_VH_LAUNCH(GEMM_OUT_F32_SCALED_STATS);
#undef TTS_VH_LAUNCH
  }
  return hipGetLastError();
}

} // namespace tts

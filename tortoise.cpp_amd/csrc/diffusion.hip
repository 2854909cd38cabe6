// Diffusion decoder stage on gfx950.
//
// Replaces diffusion_model_load (main.cpp:931-1634), diffusion_graph (3066-4044) and the diffusion()
// driver (5614-6042).
//
// Design (vs. the reference, which rebuilds/re-uploads everything for each of the 160 forwards):
//  * all candidates and both guidance branches run as ONE batch: 2*B sequences packed along the GEMM
//    M dimension ([rows][channels], channels contiguous) with zero guard rows between sequences, so a
//    k=3 convolution is three row-shifted GEMM segments — no im2col, no boundary code;
//  * every convolution is an fp16 x fp16 -> f32 MFMA GEMM — exactly the reference's conv1d numerics
//    (fp16 weights, fp16 im2col, f32 accumulation). proj_out and attention are F32 in the
//    reference and run here on fp16 MFMA inputs with f32 accumulation (north-star choice; gated by the
//    parity tests at 1e-3 of the output range);
//  * the timestep-independent latent conditioner is evaluated once per utterance, the time-embedding
//    MLP and all 16 emb_layers once per run for all steps;
//  * x_t never leaves the device: the ancestral update runs in a kernel, noise comes either from the
//    host (reference stream) or from a counter-based device generator.
#include "common.h"
#include "gemm_f16.h"
#include <algorithm>
#include <chrono>
#include <cmath>

namespace tts {

static constexpr int C = 1024, NHEAD = 16, XTC = 128 /* x_t channels padded 100 -> 128 */;
static constexpr int LAT_MAX_ROWS = 2048; // option latency_mode applies to packed layouts of at most this many rows: ONE utterance with both guidance branches (measured: -4.8 % there, +5.7 % at two utterances)
static constexpr int HOIST_MAX_ROWS = 16384; // option hoist_integrator: layouts of at most this many rows (eight utterances: -8.3 % at one, -5.6 % at two, -3.1 % at four, -1.4 % at eight, +2 % at sixteen) evaluate the integrator layers before the loop

// ------------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------------
// SiLU. The default path is x * rcp(1 + exp2(-x log2 e)) on the hardware exp2/rcp (about 1e-7 relative, the result is
// rounded to an fp16 GEMM operand right after): libm's expf plus an IEEE division cost ~30 VALU instructions per
// element and made the GroupNorm kernel VALU-bound instead of HBM-bound (seen in its ISA: 2100 instructions per
// thread). lut = 1 (option "ggml_lut") keeps the exact fp16-table emulation.
// lut = 2 (option "attn_f32", the reference-precision mode): x / (1 + expf(-x)) with libm's expf and an IEEE division, the reference's own
// f32 formula (the fast form is 2-3 ulp off). Measured on the 80-step loop: no effect on the distance from the oracle (mean abs 4.67e-5 / 5.70e-5 /
// 6.05e-5 on the small / mid / full-size problems with either form) — kept in the parity mode because it is the reference's arithmetic, not because it
// buys anything.
__device__ __forceinline__ float silu_dev(float x, int lut) {
  if (lut == 1) {
    float xr = __half2float(__float2half_rn(x));
    return __half2float(__float2half_rn(xr / (1.0f + expf(-xr))));
  }
  if (lut == 2) return x / (1.0f + expf(-x));
  return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.44269504088896f));
}

// GroupNorm statistics, 32 groups of 32 channels over the T rows of one sequence (ggml_group_norm on
// [T,1,1024]; eps inside rstd). grid (32, ns), block 256: 8 threads x float4 per row.
__global__ __launch_bounds__(256) void gn_stats_kernel(const float *__restrict__ x, const int *__restrict__ seq_start,
                                                       const int *__restrict__ seq_len, float eps, float2 *__restrict__ stats) {
  // single pass: sums of (x - p) and (x - p)^2 around a pivot p (the group's first element) — the shift
  // keeps E[d^2] - E[d]^2 well conditioned in f32 when |mean| >> std.
  __shared__ float sh[8];
  const int grp = blockIdx.x, s = blockIdx.y, T = seq_len[s];
  const float *g0 = x + (size_t)seq_start[s] * C + grp * 32;
  const float pivot = g0[0];
  const float *base = g0 + (threadIdx.x & 7) * 4;
  float sum = 0.f, sq = 0.f;
#pragma unroll 4
  for (int t = threadIdx.x >> 3; t < T; t += 32) {
    float4 v = *(const float4 *)(base + (size_t)t * C);
    v.x -= pivot; v.y -= pivot; v.z -= pivot; v.w -= pivot;
    sum += (v.x + v.y) + (v.z + v.w);
    sq += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { sum += __shfl_xor(sum, o); sq += __shfl_xor(sq, o); }
  if ((threadIdx.x & 63) == 0) { sh[threadIdx.x >> 6] = sum; sh[4 + (threadIdx.x >> 6)] = sq; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const float n = (float)T * 32.f;
    const float md = ((sh[0] + sh[1]) + (sh[2] + sh[3])) / n;
    const float var = fmaxf(((sh[4] + sh[5]) + (sh[6] + sh[7])) / n - md * md, 0.f);
    stats[s * 32 + grp] = make_float2(pivot + md, 1.0f / sqrtf(var + eps));
  }
}

// Fused GroupNorm: statistics AND normalise/affine/[scale-shift]/[SiLU]/fp16 in one launch. One block owns
// two adjacent groups (64 channels) of one sequence: pass 1 reads the [T][64] slab (256 B per row) and
// reduces sum / sum of squares around a pivot, pass 2 re-reads it (L2-resident: T*256 B) and writes the fp16
// operand. Saves one full HBM read of the activation and one launch per GroupNorm. Guard rows of the output
// are zeroed by the same blocks (rows between this sequence's end and the next sequence's start).
template <int USE_LDS>
__global__ __launch_bounds__(256) void gn_fused_kernel(const float *__restrict__ x, const int *__restrict__ seq_start,
                                                       const int *__restrict__ seq_len, int rows_total, int ns, float eps,
                                                       const float *__restrict__ g, const float *__restrict__ b,
                                                       const float *__restrict__ ss, int do_silu, int lut, __half *__restrict__ y,
                                                       const int *__restrict__ seq_step, int ss_step_stride) {
  // one block = one (sequence, group of 32 channels): the [T][32] slab (128 B per row) is read from HBM once,
  // kept in LDS (USE_LDS: T*128 B <= 144 KB) and normalised from there; longer sequences re-read it.
  extern __shared__ __attribute__((aligned(16))) float slab[];
  __shared__ float sh[8];
  const int grp = blockIdx.x, s = blockIdx.y, T = seq_len[s], r0 = seq_start[s];
  const int q = threadIdx.x & 7, c = grp * 32 + q * 4; // 8 threads x float4 per row, 32 rows per sweep
  const float *base = x + (size_t)r0 * C + c;
  const float pivot = x[(size_t)r0 * C + grp * 32];
  float sum = 0.f, sq = 0.f;
#pragma unroll 8
  for (int t = threadIdx.x >> 3; t < T; t += 32) {
    float4 v = *(const float4 *)(base + (size_t)t * C);
    if (USE_LDS) *(float4 *)(slab + t * 32 + q * 4) = v;
    v.x -= pivot; v.y -= pivot; v.z -= pivot; v.w -= pivot;
    sum += (v.x + v.y) + (v.z + v.w);
    sq += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { sum += __shfl_xor(sum, o); sq += __shfl_xor(sq, o); }
  if ((threadIdx.x & 63) == 0) { sh[threadIdx.x >> 6] = sum; sh[4 + (threadIdx.x >> 6)] = sq; }
  __syncthreads();
  const float n = (float)T * 32.f;
  const float md = ((sh[0] + sh[1]) + (sh[2] + sh[3])) / n;
  const float mean = pivot + md, rstd = 1.0f / sqrtf(fmaxf(((sh[4] + sh[5]) + (sh[6] + sh[7])) / n - md * md, 0.f) + eps);
  const float4 gg = *(const float4 *)(g + c), bb = *(const float4 *)(b + c);
  float sc4[4] = {1.f, 1.f, 1.f, 1.f}, sh4[4] = {0.f, 0.f, 0.f, 0.f};
  if (ss) {
    if (seq_step) ss += (size_t)seq_step[s] * ss_step_stride; // this sequence's timestep (the integrator evaluated for many timesteps in one batch)
#pragma unroll
    for (int i = 0; i < 4; i++) { sc4[i] = ss[c + i] + 1.0f; sh4[i] = ss[C + c + i]; }
  }
  const float ge[4] = {gg.x, gg.y, gg.z, gg.w}, be[4] = {bb.x, bb.y, bb.z, bb.w};
#pragma unroll 4
  for (int t = threadIdx.x >> 3; t < T; t += 32) {
    const float4 v = USE_LDS ? *(const float4 *)(slab + t * 32 + q * 4) : *(const float4 *)(base + (size_t)t * C);
    float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; i++) {
      float u = (e[i] - mean) * rstd;
      u = u * ge[i];
      u = u + be[i];
      if (ss) { u = u * sc4[i]; u = u + sh4[i]; }
      if (do_silu) u = silu_dev(u, lut);
      e[i] = u;
    }
    const __half2 p0 = __floats2half2_rn(e[0], e[1]), p1 = __floats2half2_rn(e[2], e[3]);
    uint2 o;
    o.x = *(const unsigned *)&p0;
    o.y = *(const unsigned *)&p1;
    *(uint2 *)(y + (size_t)(r0 + t) * C + c) = o;
  }
  // zero the guard/padding rows that follow this sequence (and those before the first one)
  const int gend = (s + 1 < ns) ? seq_start[s + 1] : rows_total;
  for (int r = r0 + T + (threadIdx.x >> 3); r < gend; r += 32) *(uint2 *)(y + (size_t)r * C + c) = make_uint2(0u, 0u);
  if (s == 0)
    for (int r = threadIdx.x >> 3; r < r0; r += 32) *(uint2 *)(y + (size_t)r * C + c) = make_uint2(0u, 0u);
}

// Same normalisation but f32 output with the conditioning-latent scale/shift: code_norm at the end of
// the latent conditioner (main.cpp:3291-3319).
__global__ __launch_bounds__(256) void gn_apply_f32_kernel(const float *__restrict__ x, const int *__restrict__ row_seq,
                                                           const float2 *__restrict__ stats, const float *__restrict__ g,
                                                           const float *__restrict__ b, const float *__restrict__ ss,
                                                           float *__restrict__ y) {
  const int r = blockIdx.x, c = threadIdx.x * 4, s = row_seq[r];
  float4 out = make_float4(0.f, 0.f, 0.f, 0.f);
  if (s >= 0) {
    const float2 st = stats[s * 32 + (c >> 5)];
    float4 v = *(const float4 *)(x + (size_t)r * C + c);
    float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; i++) {
      float t = (e[i] - st.x) * st.y;
      t = t * g[c + i];
      t = t + b[c + i];
      t = t * (ss[c + i] + 1.0f);
      t = t + ss[C + c + i];
      e[i] = t;
    }
    out = make_float4(e[0], e[1], e[2], e[3]);
  }
  *(float4 *)(y + (size_t)r * C + c) = out;
}

// f32 -> fp16 copy of [rows][1024] with guard rows zeroed.
__global__ __launch_bounds__(256) void to_f16_kernel(const float *__restrict__ x, const int *__restrict__ row_seq,
                                                     __half *__restrict__ y) {
  const int r = blockIdx.x, c = threadIdx.x * 4;
  uint2 o = make_uint2(0u, 0u);
  if (row_seq[r] >= 0) {
    float4 v = *(const float4 *)(x + (size_t)r * C + c);
    __half2 p0 = __floats2half2_rn(v.x, v.y), p1 = __floats2half2_rn(v.z, v.w);
    o.x = *(unsigned *)&p0;
    o.y = *(unsigned *)&p1;
  }
  *(uint2 *)(y + (size_t)r * C + c) = o;
}

// Option fp16_check: count non-finite values and values beyond 60000 in an fp16 operand buffer (counts: int64[2]).
__global__ __launch_bounds__(256) void fp16_scan_kernel(const __half *__restrict__ p, size_t n, unsigned long long *__restrict__ counts) {
  unsigned long long bad = 0, big = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float v = __half2float(p[i]);
    if (!(fabsf(v) <= 65504.0f)) bad++;
    else if (fabsf(v) > 60000.0f) big++;
  }
  if (bad) atomicAdd(counts, bad);
  if (big) atomicAdd(counts + 1, big);
}

// f32 -> fp16 copy of rows gathered from another layout: y[r] = x[src_row[r]], zero where src_row[r] < 0 (guard rows).
__global__ __launch_bounds__(256) void gather_f16_kernel(const float *__restrict__ x, const int *__restrict__ src_row,
                                                         __half *__restrict__ y) {
  const int r = blockIdx.x, c = threadIdx.x * 4, sr = src_row[r];
  uint2 o = make_uint2(0u, 0u);
  if (sr >= 0) {
    float4 v = *(const float4 *)(x + (size_t)sr * C + c);
    __half2 p0 = __floats2half2_rn(v.x, v.y), p1 = __floats2half2_rn(v.z, v.w);
    o.x = *(unsigned *)&p0;
    o.y = *(unsigned *)&p1;
  }
  *(uint2 *)(y + (size_t)r * C + c) = o;
}

// f32 row gather: y[r] = x[src_row[r]] (zero where src_row[r] < 0).
__global__ __launch_bounds__(256) void gather_f32_kernel(const float *__restrict__ x, const int *__restrict__ src_row, float *__restrict__ y) {
  const int r = blockIdx.x, c = threadIdx.x * 4, sr = src_row[r];
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (sr >= 0) v = *(const float4 *)(x + (size_t)sr * C + c);
  *(float4 *)(y + (size_t)r * C + c) = v;
}
// The current sampling step's slice of a per-step array (the hoisted integrator's code-embedding operand) -> the fixed address the step's kernels read.
__global__ __launch_bounds__(256) void select_step_slice_kernel(const uint4 *__restrict__ all, size_t n16_per_step, const int *__restrict__ ctr, uint4 *__restrict__ dst) {
  const uint4 *src = all + (size_t)(*ctr) * n16_per_step;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16_per_step; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}

// Multi-head attention with T5 relative-position bias (AttentionBlock, main.cpp:3232-3275).
// One block = 128 queries of one (sequence, head): 4 waves x 32 queries; keys stream through a double
// buffered LDS stage (global->LDS DMA) in tiles of 64. Everything is computed TRANSPOSED so that a lane owns
// ONE query and a quarter of the keys: S^T = K Q^T (D[key][query]) and O^T += V^T P^T (D[d][query]).
//  * the softmax row reduction is 15 in-lane ops + two cross-lane steps (xor 16, 32) per query,
//  * P^T is already in the MFMA B-operand layout of the second product (with the key order inside each
//    32-key step permuted identically for V^T), so P never goes through LDS,
//  * the output is 4 consecutive channels per lane (8-byte stores).
// bias = tab[(q<k)*64 + min(|k-q|,63)] (32-bucket table x8, expanded per distance at load time); blocks
// of keys at least 63 away from every query of the wave use the saturated constant.
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
static constexpr int ATT_TAB = 320;
template <int NR> constexpr int att_lds() { return NR * 16384 + ATT_TAB * 4; }
#ifdef TTS_ATT_TRACE // developer build (tools/attn_bench.hip): per-tile phase timestamps of wave 0 of workgroup 0
__device__ long long tts_att_trace[64 * 8];
#define ATT_CLK(j) do { if (blockIdx.x == TTS_ATT_TRACE && threadIdx.x == 0) { tts_att_trace[63 * 8 + 2 * (j)] = __builtin_readcyclecounter(); tts_att_trace[63 * 8 + 2 * (j) + 1] = wall_clock64(); } } while (0)
#define ATT_T(i) do { if (blockIdx.x == TTS_ATT_TRACE && threadIdx.x == 0 && kb < 64) tts_att_trace[kb * 8 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define ATT_T(i)
#define ATT_CLK(j)
#endif
__device__ __forceinline__ int attn_off(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }

// all-reduce over the four 16-lane rows of a wave (lanes l, l^16, l^32, l^48) with the gfx950 half/row swap
// instructions (v_permlane32_swap / v_permlane16_swap) instead of ds_bpermute round trips through LDS.
__device__ __forceinline__ float rows4_max(float x) {
  auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  x = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
  auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float rows4_sum(float x) {
  auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  x = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

// The key tiles of a wave fall into four classes that depend only on (tile, wave): far below the diagonal (every key at least 63 before
// every query of the wave: the bias is ONE constant), near the diagonal (bias by distance from the LDS table), far above it (the other
// constant), and the last tile of a sequence whose length is not a multiple of 64 (masked). Round 4: the tile loop is split into those four
// ranges with one straight-line body each (`tile<MODE>`), instead of one body with three data paths and a per-tile branch: the branch joins
// cost 44 register copies per tile (v_mov of the 32 score registers: the MFMA results are register tuples, the near path rewrote them, the
// far path did not) and the PMC pass says the kernel is bound by VALU issue, not by the matrix pipe (profiles/r4_pmc_attention.json:
// VALU pipe 66 % busy, matrix pipe 35 %, 6.6 VALU instructions per MFMA). Same arithmetic in the same order: bit-identical output.
enum { ATT_FAR = 0, ATT_NEAR = 1, ATT_TAIL = 2 };
// NI: 16-query blocks per wave. 2 = 128 queries per workgroup (the batch shape). 1 = 64 queries per workgroup (round 6): for grids that would otherwise put at most one
// workgroup on a CU (one utterance: 2 sequences x 16 heads x 7 query blocks = 224) twice as many, half as long workgroups give every SIMD a second wave. Every query's
// arithmetic is the same in the same key order: bit-identical (tests/test_latency_mode_gpu.py). MEASURED, NO GAIN: one utterance's diffusion stage 138.1 / 136.0 ms with 128-query
// workgroups, 136.2 / 135.9 with 64; two utterances 212.1 / 210.1 against 213.6 / 213.3 (profiles/r6_small_batch.txt) — the kernel's tile loop is a dependent chain per
// wave (scores -> max -> exp -> PV) that a second wave per SIMD does not shorten at this size. Kept behind option attn_q64 (default 0) for A/B.
template <int NR, int NI = 2> // NR: K/V ring depth: 3 = two tiles in flight, 3 workgroups per CU; 2 = one tile in flight, 4 workgroups per CU
__global__ __launch_bounds__(256, NR == 2 ? 4 : 3) void diff_attn_kernel(const __half *__restrict__ qk, const __half *__restrict__ vt, int ldvt,
                                                        const int *__restrict__ seq_start, const int *__restrict__ seq_len,
                                                        const float *__restrict__ bias_tab, __half *__restrict__ out, int nq) {
  // ONE LDS object: with a second __shared__ variable hipcc puts an s_waitcnt vmcnt(0) in front of the first
  // ds_read of every tile, which drains the DMA prefetch (seen in the ISA; cdna_hip_programming.md §5 trap (a)).
  // (dynamic LDS: with a static array the DMA writes and the fragment reads alias for the waitcnt pass as well)
  extern __shared__ __attribute__((aligned(16))) char smem[]; // NR x (K tile 8 KB | V^T tile 8 KB) + bias table
  float *tab = (float *)(smem + NR * 16384);
  // XCD-aware block order: workgroup id b runs on XCD b % 8, so all q-blocks of one (sequence, head) pair
  // get ids congruent mod 8 and reuse that pair's K/V tiles from one L2 (16 heads => pairs % 8 == 0).
  const int xcd = blockIdx.x & 7, tt = blockIdx.x >> 3;
  const int pair = (tt / nq) * 8 + xcd, h = pair & 15, s = pair >> 4;
  const int T = seq_len[s], r0 = seq_start[s], q0 = (tt % nq) * (64 * NI);
  if (q0 >= T) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), fr = lane & 15, fq = lane >> 4;
  const float L2E = 1.44269504088896f;
  // Bias by SIGNED key-query distance d in [-160, 160), saturated outside +-63, in raw-score units (added to q.k
  // before the 1/8 * log2e scaling): a lane's 16 keys of a tile sit at compile-time offsets from one base distance,
  // so the near-diagonal path is one LDS read at an immediate offset + one add per score.
  const float SC = 0.125f * L2E; // 1/sqrt(64) in log2 units (softmax via exp2)
  for (int j = tid; j < ATT_TAB; j += 256) {
    const int d = j - ATT_TAB / 2, ad = d < 0 ? -d : d;
    tab[j] = bias_tab[h * 128 + (d > 0 ? 64 : 0) + (ad < 63 ? ad : 63)] * (L2E / SC);
  }
  constexpr int QW = 16 * NI; // queries per wave
  const int qw = q0 + wave * QW;
  half8 qf[NI][2]; // Q[query = qw + i*16 + fr][d = ks*32 + fq*8 ..+7]
#pragma unroll
  for (int i = 0; i < NI; i++)
#pragma unroll
    for (int ks = 0; ks < 2; ks++)
      qf[i][ks] = *(const half8 *)(qk + (size_t)(r0 + qw + i * 16 + fr) * 2048 + h * 128 + ks * 32 + fq * 8);
  // Retire the Q loads HERE (a use makes hipcc place its vmcnt(0) now): vmcnt is an in-order counter, so a Q
  // load still pending at the loop would force vmcnt(0) in front of the first MFMA of every tile and drain the
  // K/V prefetch (seen in the ISA as `s_waitcnt vmcnt(0) lgkmcnt(0)` after the ds_reads).
  if constexpr (NI == 2) asm volatile("" ::"v"(qf[0][0]), "v"(qf[0][1]), "v"(qf[1][0]), "v"(qf[1][1]));
  else asm volatile("" ::"v"(qf[0][0]), "v"(qf[0][1]));
  floatx4 o[NI][4]; // O^T[d = dt*16 + fq*4 + r][query = qw + i*16 + fr]
  floatx4 lacc[NI]; // row sums of P from the matrix pipe: (all-ones A tile) . P^T, every register = l[query fr]
  float mrow[NI];
#pragma unroll
  for (int i = 0; i < NI; i++) {
#pragma unroll
    for (int j = 0; j < 4; j++) o[i][j] = (floatx4){0.f, 0.f, 0.f, 0.f};
    lacc[i] = (floatx4){0.f, 0.f, 0.f, 0.f};
    mrow[i] = -INFINITY;
  }
  const int nkb = (T + 63) >> 6;
  const int prow = lane >> 3, pslot = lane & 7;
  const __half *kbase = qk + (size_t)r0 * 2048 + h * 128 + 64;
  const __half *vbase = vt + (size_t)(h * 64) * ldvt + r0;
  // K and V^T tiles live in an NR-deep ring of (K 8 KB | V^T 8 KB) slots filled by LDS-DMA NR - 1 tiles ahead.
  // Wave w moves rows w*16 .. w*16+15 of both tiles; swizzle on the source chunk.
  // The K tile is stored with its key rows permuted: LDS row jt*16 + x holds key SIG(jt, x) =
  // (jt>>1)*32 + (x>>2)*8 + (jt&1)*4 + (x&3). The score accumulator (jt, fq, r) then belongs to key
  // (jt>>1)*32 + fq*8 + (jt&1)*4 + r, so the 8 P values a lane feeds to PV step ks2 are the 8 CONSECUTIVE keys
  // 32 ks2 + 8 fq .. +7 and its V^T fragment is one 16-byte LDS read (no half-fragment shuffles).
  // Tile indices past the end are clamped (harmless re-stage) so that the vmcnt arithmetic stays uniform.
  int koff[2], voff[2]; // per-lane source offsets (halves) inside a tile, fixed for the whole kernel
#pragma unroll
  for (int i = 0; i < 2; i++) {
    const int row = wave * 16 + i * 8 + prow, c = pslot ^ (row & 7);
    const int jt = row >> 4, x = row & 15;
    const int key = (jt >> 1) * 32 + (x >> 2) * 8 + (jt & 1) * 4 + (x & 3);
    koff[i] = key * 2048 + c * 8;
    voff[i] = row * ldvt + c * 8;
  }
  auto stage = [&](int kb, int slot) {
    kb = min(kb, nkb - 1);
    const __half *ksrc = kbase + (size_t)kb * (64 * 2048), *vsrc = vbase + kb * 64; // wave-uniform
    char *ks_ = smem + slot * 16384 + wave * 2048, *vs_ = ks_ + 8192;
    __builtin_amdgcn_global_load_lds((gptr_t)(ksrc + koff[0]), (lptr_t)ks_, 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gptr_t)(ksrc + koff[1]), (lptr_t)(ks_ + 1024), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gptr_t)(vsrc + voff[0]), (lptr_t)vs_, 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gptr_t)(vsrc + voff[1]), (lptr_t)(vs_ + 1024), 16, 0, 0);
  };
  // S^T of one key tile: sc[i][jt][r] = S[query i*16+fr][key SIG(jt, fq*4 + r)]
  auto scores = [&](const char *Ks, floatx4 (&sc)[NI][4]) {
#pragma unroll
    for (int jt = 0; jt < 4; jt++) {
      const half8 kf0 = *(const half8 *)(Ks + attn_off(jt * 16 + fr, fq));
      const half8 kf1 = *(const half8 *)(Ks + attn_off(jt * 16 + fr, 4 + fq));
#pragma unroll
      for (int i = 0; i < NI; i++) {
        floatx4 a = (floatx4){0.f, 0.f, 0.f, 0.f};
        a = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf0, qf[i][0], a, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf1, qf[i][1], a, 0, 0, 0);
        sc[i][jt] = a;
      }
    }
  };
  ATT_CLK(0);
  stage(0, 0);
  if (NR == 3) stage(1, 1);
  half8 ones;
#pragma unroll
  for (int e = 0; e < 8; e++) ones[e] = (_Float16)1.0f;
  // one key tile; MODE is compile-time, cidx = table index of the constant bias of a far tile (read AFTER the tile's barrier: the table is
  // filled by the whole workgroup in front of the loop)
  auto tile = [&](int kb, auto mode_c, int cidx) {
    constexpr int MODE = decltype(mode_c)::value;
    // Tile kb must have landed; the 4 DMA pieces of tile kb+1 may stay in flight across the barrier
    // (counted vmcnt + raw s_barrier: __syncthreads() would drain the prefetch).
    ATT_T(0);
    if (NR == 3) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ATT_T(1);
    __builtin_amdgcn_s_barrier();
    ATT_T(2);
    // every wave has passed the barrier => nobody still reads tile kb-1, whose slot receives the next tile to stage
    stage(kb + NR - 1, (kb + NR - 1) % NR);
    const char *Ks = smem + (kb % NR) * 16384, *Vs = Ks + 8192;
    ATT_T(3);
    floatx4 sc[NI][4];
    scores(Ks, sc);
    ATT_T(4);
    const int kmin = kb * 64;
    half8 pf[NI][2]; // P^T in B-operand layout: slot e of step ks2 = key 32 ks2 + 8 fq + e
#pragma unroll
    for (int i = 0; i < NI; i++) {
      const int qi = qw + i * 16 + fr;
      float mx = -INFINITY, boff = 0.f; // v = sc*SC + bias; far tiles: bias is one constant (folded below)
      if (MODE == ATT_FAR) {
        boff = tab[cidx] * SC;
#pragma unroll
        for (int jt = 0; jt < 4; jt++) { // two v_max3 per accumulator register quad
          mx = fmaxf(fmaxf(mx, sc[i][jt][0]), sc[i][jt][1]);
          mx = fmaxf(fmaxf(mx, sc[i][jt][2]), sc[i][jt][3]);
        }
        mx = fmaf(mx, SC, boff);
      } else if (MODE == ATT_NEAR) {
        // key of (jt, r) = kmin + fq*8 + off, off = (jt>>1)*32 + (jt&1)*4 + r  =>  d = (kmin + fq*8 - qi) + off
        const float *tp = tab + (kmin + fq * 8 - qi + ATT_TAB / 2); // in range: |d| < 160 on near-diagonal tiles
#pragma unroll
        for (int jt = 0; jt < 4; jt++)
#pragma unroll
          for (int r = 0; r < 4; r++) {
            const float v = sc[i][jt][r] + tp[(jt >> 1) * 32 + (jt & 1) * 4 + r];
            sc[i][jt][r] = v;
            mx = fmaxf(mx, v);
          }
        mx *= SC;
      } else { // last tile of the sequence: keys >= T are masked
        const int left = T - kmin - fq * 8; // keys of this lane with off < left exist
        const bool far_hi = kmin - (qw + QW - 1) >= 63, far = far_hi || qw - (kmin + 63) >= 63; // then the bias is one constant (and the table base would be out of range)
        const float cb = far_hi ? tab[ATT_TAB / 2 + 63] : tab[ATT_TAB / 2 - 63];
        const float *tp = tab + (far ? 0 : kmin + fq * 8 - qi + ATT_TAB / 2);
        float bv[4][4]; // all table reads first, unconditionally (a load under a per-element select is branched around)
#pragma unroll
        for (int jt = 0; jt < 4; jt++)
#pragma unroll
          for (int r = 0; r < 4; r++) bv[jt][r] = tp[(jt >> 1) * 32 + (jt & 1) * 4 + r];
#pragma unroll
        for (int jt = 0; jt < 4; jt++)
#pragma unroll
          for (int r = 0; r < 4; r++) {
            const int off = (jt >> 1) * 32 + (jt & 1) * 4 + r;
            const float v = off < left ? sc[i][jt][r] + (far ? cb : bv[jt][r]) : -INFINITY;
            sc[i][jt][r] = v;
            mx = fmaxf(mx, v);
          }
        mx *= SC;
      }
      mx = rows4_max(mx);
      const float mnew = fmaxf(mrow[i], mx);
      const float alpha = __builtin_amdgcn_exp2f(mrow[i] - mnew);
      // p = 2^(sc*SC + bias - mnew). (Round 5: scaling the numerators by 2^14 to keep them out of the fp16 subnormals — it cancels in o / l — was measured:
      // no change of the 80-step distance from the oracle; v_cvt_pk_f16_f32 and the MFMA both keep subnormals, tools/r5/mfma_denorm_probe.hip.)
      const float sub = boff - mnew;
      mrow[i] = mnew;
      if (!__all(alpha == 1.0f)) { // the running max settles after the first tiles: usually nothing to rescale
#pragma unroll
        for (int dt = 0; dt < 4; dt++)
#pragma unroll
          for (int r = 0; r < 4; r++) o[i][dt][r] *= alpha;
#pragma unroll
        for (int r = 0; r < 4; r++) lacc[i][r] *= alpha;
      }
#pragma unroll
      for (int ks2 = 0; ks2 < 2; ks2++)
#pragma unroll
        for (int e = 0; e < 8; e++)
          pf[i][ks2][e] = (_Float16)__builtin_amdgcn_exp2f(fmaf(sc[i][2 * ks2 + (e >> 2)][e & 3], SC, sub));
    }
    ATT_T(5);
    // O^T += V^T P^T : A = V^T[d = dt*16 + fr][keys 32 ks2 + 8 fq ..+7], B = P^T; row sums: A = ones
#pragma unroll
    for (int ks2 = 0; ks2 < 2; ks2++) {
#pragma unroll
      for (int dt = 0; dt < 4; dt++) {
        const half8 vf = *(const half8 *)(Vs + attn_off(dt * 16 + fr, 4 * ks2 + fq));
#pragma unroll
        for (int i = 0; i < NI; i++) o[i][dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf[i][ks2], o[i][dt], 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < NI; i++) lacc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ones, pf[i][ks2], lacc[i], 0, 0, 0);
    }
    ATT_T(6);
  };
  {
    // tiles [0, a): every key at least 63 before every query of the wave (qw - (kmin + 63) >= 63); [b, ..): at least 63 after (kmin - (qw + QW - 1) >= 63)
    const int last = (T & 63) ? nkb - 1 : nkb; // the masked tile, if any, is handled on its own
    const int a = min(max((qw - 126) >= 0 ? (qw - 126) / 64 + 1 : 0, 0), last), b = min((qw + QW + 62 + 63) / 64, last);
    int kb = 0;
    for (; kb < a; kb++) tile(kb, std::integral_constant<int, ATT_FAR>{}, ATT_TAB / 2 - 63);
    for (; kb < b; kb++) tile(kb, std::integral_constant<int, ATT_NEAR>{}, 0);
    for (; kb < last; kb++) tile(kb, std::integral_constant<int, ATT_FAR>{}, ATT_TAB / 2 + 63);
    if (last < nkb) tile(last, std::integral_constant<int, ATT_TAIL>{}, 0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // trailing (clamped) DMA pieces must land before the LDS is released
  ATT_CLK(1);
  #pragma unroll
  for (int i = 0; i < NI; i++) {
    const int qi = qw + i * 16 + fr;
    if (qi < T) {
      const float inv = 1.0f / lacc[i][0];
#pragma unroll
      for (int dt = 0; dt < 4; dt++) {
        __half2 p0 = __floats2half2_rn(o[i][dt][0] * inv, o[i][dt][1] * inv), p1 = __floats2half2_rn(o[i][dt][2] * inv, o[i][dt][3] * inv);
        uint2 u;
        u.x = *(unsigned *)&p0;
        u.y = *(unsigned *)&p1;
        *(uint2 *)(out + (size_t)(r0 + qi) * C + h * 64 + dt * 16 + fq * 4) = u;
      }
    }
  }
}

// Reference-precision AttentionBlock core (option attn_f32; main.cpp:3848-3875 evaluates QK^T, softmax and PV as F32 ggml_mul_mat /
// ggml_soft_max). Same decomposition as diff_attn_kernel (128 queries of one (sequence, head) per workgroup, everything transposed, K rows
// permuted so that P^T is the B operand of the second product), but every matrix product runs on SPLIT-PRECISION fp16 operands: x = hi + lo
// with hi = fp16(x), lo = fp16(x - hi), and a.b ~ a_lo.b_hi + a_hi.b_lo + a_hi.b_hi on three v_mfma_f32_16x16x32_f16 (every partial
// product is exact in f32, the dropped lo.lo term is 2^-22 relative: f32-class arithmetic, the scheme of the AR multi-row passes).
//  * s = q.k * 0.125 + bias in f32 exactly as the reference orders it, running max, p = 2^((s - m) log2 e + 8) on the hardware exp2 (the
//    factor 2^8 keeps the low halves of P out of the fp16 subnormals; it cancels in o / l),
//  * row sums are f32 VALU adds of the unrounded p (not the matrix pipe over rounded values), one cross-lane reduction at the end,
//  * the output o / l is written as a split pair for the split-precision proj_out GEMM.
// LDS: 2 ring slots x (K hi | K lo | V^T hi | V^T lo) 8 KB tiles + the bias table. ~3x the matrix work of diff_attn_kernel: the parity
// mode, not the throughput mode.
static constexpr int ATT32_SLOT = 32768, ATT32_LDS = 2 * ATT32_SLOT + 128 * 4;
__global__ __launch_bounds__(256, 2) void diff_attn_f32_kernel(const __half *__restrict__ qk_hi, const __half *__restrict__ qk_lo,
                                                               const __half *__restrict__ vt_hi, const __half *__restrict__ vt_lo, int ldvt,
                                                               const int *__restrict__ seq_start, const int *__restrict__ seq_len,
                                                               const float *__restrict__ bias_tab, __half *__restrict__ out_hi,
                                                               __half *__restrict__ out_lo, int nq) {
  extern __shared__ __attribute__((aligned(16))) char smem[]; // ONE LDS object (see diff_attn_kernel)
  float *tab = (float *)(smem + 2 * ATT32_SLOT);              // bias by signed distance d = key - query, clamped to +-63: tab[d + 64]
  const int xcd = blockIdx.x & 7, tt = blockIdx.x >> 3;
  const int pair = (tt / nq) * 8 + xcd, h = pair & 15, s = pair >> 4;
  const int T = seq_len[s], r0 = seq_start[s], q0 = (tt % nq) * 128;
  if (q0 >= T) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), fr = lane & 15, fq = lane >> 4;
  const float L2E = 1.44269504088896f;
  if (tid < 128) {
    const int d = tid - 64, ad = d < 0 ? -d : d;
    tab[tid] = bias_tab[h * 128 + (d > 0 ? 64 : 0) + (ad < 63 ? ad : 63)];
  }
  const int qw = q0 + wave * 32;
  half8 qh[2][2], ql[2][2]; // Q[query = qw + i*16 + fr][d = ks*32 + fq*8 ..+7]
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {
      const size_t off = (size_t)(r0 + qw + i * 16 + fr) * 2048 + h * 128 + ks * 32 + fq * 8;
      qh[i][ks] = *(const half8 *)(qk_hi + off);
      ql[i][ks] = *(const half8 *)(qk_lo + off);
    }
  asm volatile("" ::"v"(qh[0][0]), "v"(qh[0][1]), "v"(qh[1][0]), "v"(qh[1][1]), "v"(ql[0][0]), "v"(ql[0][1]), "v"(ql[1][0]), "v"(ql[1][1])); // retire the Q loads before the DMA queue starts
  floatx4 o[2][4]; // O^T[d = dt*16 + fq*4 + r][query = qw + i*16 + fr]
  float lsum[2], mrow[2];
#pragma unroll
  for (int i = 0; i < 2; i++) {
#pragma unroll
    for (int j = 0; j < 4; j++) o[i][j] = (floatx4){0.f, 0.f, 0.f, 0.f};
    lsum[i] = 0.f;
    mrow[i] = -INFINITY;
  }
  const int nkb = (T + 63) >> 6;
  const int prow = lane >> 3, pslot = lane & 7;
  const size_t kofs = (size_t)r0 * 2048 + h * 128 + 64, vofs = (size_t)(h * 64) * ldvt + r0;
  int koff[2], voff[2]; // per-lane source offsets (halves) inside a tile: see diff_attn_kernel (key permutation SIG, swizzle on the source chunk)
#pragma unroll
  for (int i = 0; i < 2; i++) {
    const int row = wave * 16 + i * 8 + prow, c = pslot ^ (row & 7);
    const int jt = row >> 4, x = row & 15;
    const int key = (jt >> 1) * 32 + (x >> 2) * 8 + (jt & 1) * 4 + (x & 3);
    koff[i] = key * 2048 + c * 8;
    voff[i] = row * ldvt + c * 8;
  }
  auto stage = [&](int kb, int slot) {
    kb = min(kb, nkb - 1);
    const size_t ko = kofs + (size_t)kb * (64 * 2048), vo = vofs + kb * 64;
    char *dst = smem + slot * ATT32_SLOT + wave * 2048;
#pragma unroll
    for (int i = 0; i < 2; i++) {
      __builtin_amdgcn_global_load_lds((gptr_t)(qk_hi + ko + koff[i]), (lptr_t)(dst + i * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)(qk_lo + ko + koff[i]), (lptr_t)(dst + 8192 + i * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)(vt_hi + vo + voff[i]), (lptr_t)(dst + 16384 + i * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)(vt_lo + vo + voff[i]), (lptr_t)(dst + 24576 + i * 1024), 16, 0, 0);
    }
  };
  stage(0, 0);
  for (int kb = 0; kb < nkb; kb++) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // tile kb has landed (this wave's pieces)
    __builtin_amdgcn_s_barrier();                    // ... everybody's; and nobody still reads tile kb-1, whose slot is restaged now
    stage(kb + 1, (kb + 1) & 1);
    const char *Kh = smem + (kb & 1) * ATT32_SLOT, *Kl = Kh + 8192, *Vh = Kh + 16384, *Vl = Kh + 24576;
    floatx4 sc[2][4]; // sc[i][jt][r] = q . k of query i*16+fr and key SIG(jt, fq*4 + r) = kmin + (jt>>1)*32 + fq*8 + (jt&1)*4 + r
#pragma unroll
    for (int jt = 0; jt < 4; jt++) {
      const half8 kh0 = *(const half8 *)(Kh + attn_off(jt * 16 + fr, fq)), kh1 = *(const half8 *)(Kh + attn_off(jt * 16 + fr, 4 + fq));
      const half8 kl0 = *(const half8 *)(Kl + attn_off(jt * 16 + fr, fq)), kl1 = *(const half8 *)(Kl + attn_off(jt * 16 + fr, 4 + fq));
#pragma unroll
      for (int i = 0; i < 2; i++) {
        floatx4 a = (floatx4){0.f, 0.f, 0.f, 0.f}; // small terms first
        a = __builtin_amdgcn_mfma_f32_16x16x32_f16(kl0, qh[i][0], a, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_16x16x32_f16(kl1, qh[i][1], a, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh0, ql[i][0], a, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh1, ql[i][1], a, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh0, qh[i][0], a, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh1, qh[i][1], a, 0, 0, 0);
        sc[i][jt] = a;
      }
    }
    const int kmin = kb * 64;
    half8 ph[2][2], pl[2][2]; // P^T in B-operand layout: slot e of step ks2 = key 32 ks2 + 8 fq + e
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const int qi = qw + i * 16 + fr;
      const int dbase = kmin + fq * 8 - qi; // d of (jt, r) = dbase + off, off = (jt>>1)*32 + (jt&1)*4 + r
      const int left = T - kmin - fq * 8;   // keys of this lane with off < left exist
      float mx = -INFINITY;
#pragma unroll
      for (int jt = 0; jt < 4; jt++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int off = (jt >> 1) * 32 + (jt & 1) * 4 + r;
          const float b = tab[min(max(dbase + off, -63), 63) + 64];
          float v = fmaf(sc[i][jt][r], 0.125f, b); // bias + dot * (1 / sqrt(64)) (main.cpp:3851-3870)
          v = off < left ? v : -INFINITY;
          sc[i][jt][r] = v;
          mx = fmaxf(mx, v);
        }
      mx = rows4_max(mx);
      const float mnew = fmaxf(mrow[i], mx);
      const float alpha = __builtin_amdgcn_exp2f((mrow[i] - mnew) * L2E);
      mrow[i] = mnew;
      if (!__all(alpha == 1.0f)) {
#pragma unroll
        for (int dt = 0; dt < 4; dt++)
#pragma unroll
          for (int r = 0; r < 4; r++) o[i][dt][r] *= alpha;
        lsum[i] *= alpha;
      }
      float part = 0.f;
#pragma unroll
      for (int ks2 = 0; ks2 < 2; ks2++)
#pragma unroll
        for (int e = 0; e < 8; e++) {
          const float p = __builtin_amdgcn_exp2f(fmaf(sc[i][2 * ks2 + (e >> 2)][e & 3] - mnew, L2E, 8.0f));
          part += p;
          const _Float16 hi = (_Float16)p;
          ph[i][ks2][e] = hi;
          pl[i][ks2][e] = (_Float16)(p - (float)hi);
        }
      lsum[i] += part;
    }
    // O^T += V^T P^T : A = V^T[d = dt*16 + fr][keys 32 ks2 + 8 fq ..+7], B = P^T
#pragma unroll
    for (int ks2 = 0; ks2 < 2; ks2++)
#pragma unroll
      for (int dt = 0; dt < 4; dt++) {
        const half8 vh = *(const half8 *)(Vh + attn_off(dt * 16 + fr, 4 * ks2 + fq)), vl = *(const half8 *)(Vl + attn_off(dt * 16 + fr, 4 * ks2 + fq));
#pragma unroll
        for (int i = 0; i < 2; i++) {
          o[i][dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vl, ph[i][ks2], o[i][dt], 0, 0, 0);
          o[i][dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh, pl[i][ks2], o[i][dt], 0, 0, 0);
          o[i][dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh, ph[i][ks2], o[i][dt], 0, 0, 0);
        }
      }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the trailing (clamped) DMA pieces must land before the LDS is released
#pragma unroll
  for (int i = 0; i < 2; i++) {
    const int qi = qw + i * 16 + fr;
    const float l = rows4_sum(lsum[i]);
    if (qi < T) {
      const float inv = 1.0f / l;
#pragma unroll
      for (int dt = 0; dt < 4; dt++) {
        const float v0 = o[i][dt][0] * inv, v1 = o[i][dt][1] * inv, v2 = o[i][dt][2] * inv, v3 = o[i][dt][3] * inv;
        const size_t off = (size_t)(r0 + qi) * C + h * 64 + dt * 16 + fq * 4;
        *(uint2 *)(out_hi + off) = pack_half4(v0, v1, v2, v3);
        *(uint2 *)(out_lo + off) = pack_half4(split_lo(v0), split_lo(v1), split_lo(v2), split_lo(v3));
      }
    }
  }
}

// nearest-neighbour upsample of the code embedding L -> T (ggml_upscale_ext: src = (int)(dst / ((float)T/L)))
// for the conditioned sequences, unconditioned_embedding broadcast for the others. One block per row.
__global__ __launch_bounds__(256) void build_code_emb_kernel(const float *__restrict__ lat_emb, const int *__restrict__ lat_start,
                                                             const int *__restrict__ lat_len, const float *__restrict__ uncond,
                                                             const int *__restrict__ row_seq, const int *__restrict__ row_t,
                                                             const int *__restrict__ seq_len, const int *__restrict__ seq_src /* latent seq or -1 */,
                                                             float *__restrict__ out) {
  const int r = blockIdx.x, c = threadIdx.x * 4, s = row_seq[r];
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (s >= 0) {
    const int src = seq_src[s];
    if (src < 0) v = *(const float4 *)(uncond + c);
    else {
      const int L = lat_len[src], T = seq_len[s];
      const float sf = (float)T / (float)L;
      int sr = (int)((float)row_t[r] / sf);
      sr = sr > L - 1 ? L - 1 : sr;
      v = *(const float4 *)(lat_emb + (size_t)(lat_start[src] + sr) * C + c);
    }
  }
  *(float4 *)(out + (size_t)r * C + c) = v;
}

// x_t [cand][100][T] (f32, reference layout) -> fp16 GEMM operand rows [row][128] for the conditioned
// and the unconditioned copy of the sequence. grid: rows of the cond sequences; block 128.
__global__ __launch_bounds__(128) void xt_to_rows_kernel(const float *__restrict__ x, const int64_t *__restrict__ x_off,
                                                         const int *__restrict__ row_seq, const int *__restrict__ row_t,
                                                         const int *__restrict__ seq_len, const int *__restrict__ seq_start,
                                                         int ncand, int has_uncond, __half *__restrict__ xt16) {
  const int r = blockIdx.x, s = row_seq[r], ch = threadIdx.x;
  if (s < 0 || s >= ncand) return;
  const int T = seq_len[s], t = row_t[r];
  float v = (ch < 100) ? x[x_off[s] + (size_t)ch * T + t] : 0.f;
  const __half hv = __float2half_rn(v);
  xt16[(size_t)r * XTC + ch] = hv;
  if (has_uncond) xt16[(size_t)(seq_start[s + ncand] + t) * XTC + ch] = hv;
}

// Philox4x32-10 + Box-Muller (device noise mode).
__device__ __forceinline__ void philox_round(uint32_t &c0, uint32_t &c1, uint32_t &c2, uint32_t &c3, uint32_t k0, uint32_t k1) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
  uint32_t hi0 = __umulhi(M0, c0), lo0 = M0 * c0, hi1 = __umulhi(M1, c2), lo1 = M1 * c2;
  uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
  c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
}
__device__ __forceinline__ float philox_normal(uint64_t seed, uint32_t stream, uint32_t step, uint32_t idx) {
  uint32_t c0 = idx >> 1, c1 = step, c2 = stream, c3 = 0x7715u, k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int i = 0; i < 10; i++) {
    philox_round(c0, c1, c2, c3, k0, k1);
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  const float u1 = ((float)c0 + 1.0f) * 2.3283064365386963e-10f, u2 = (float)c1 * 2.3283064365386963e-10f;
  const float rad = sqrtf(-2.0f * logf(u1));
  float sn, cs;
  sincosf(6.283185307179586f * u2, &sn, &cs);
  return (idx & 1) ? rad * sn : rad * cs;
}

// Ancestral sampling step (main.cpp:5970-6030) for every candidate, in place on x [cand][100][T].
// net: [rows][256] f32 (channels 0..99 eps, 100..199 variance logits). grid: cond rows; block 128.
struct StepScalars { float max_log, min_log, cfk, sqrt_recip, sqrt_recipm1, coef1, coef2; int is_last; };
// Everything of a sampling step that the reference keeps in host variables lives in a device table indexed by a device-side step
// counter, so that ONE captured hipGraph of the step can be replayed for every step (main.cpp:5723-6033 rebuilds and re-uploads
// its graph 160 times): the schedule scalars, the noise block of the step, the generator key.
struct StepEntry { StepScalars sc; int has_noise; long long noise_off; unsigned philox_step; unsigned pad; };

// Copies this step's [n_res][scale | shift] block to the fixed address the GroupNorm kernels read. grid: any, block 256.
__global__ __launch_bounds__(256) void step_begin_kernel(const float *__restrict__ ss_all, size_t ss_stride, const int *__restrict__ ctr,
                                                         float *__restrict__ ss_cur) {
  const float4 *src = (const float4 *)(ss_all + (size_t)(*ctr) * ss_stride);
  float4 *dst = (float4 *)ss_cur;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < ss_stride / 4; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
__global__ void step_advance_kernel(int *ctr) { *ctr += 1; }
__global__ __launch_bounds__(128) void ddpm_update_kernel(const float *__restrict__ net, float *__restrict__ x,
                                                          const int64_t *__restrict__ x_off, const int *__restrict__ row_seq,
                                                          const int *__restrict__ row_t, const int *__restrict__ seq_len,
                                                          const int *__restrict__ seq_start, int ncand, const StepEntry *__restrict__ tab,
                                                          const int *__restrict__ ctr, const float *__restrict__ noise_base /* or null */,
                                                          uint64_t seed, uint32_t stream0 /* global id of candidate 0 */) {
  const int r = blockIdx.x, s = row_seq[r], ch = threadIdx.x;
  if (s < 0 || s >= ncand || ch >= 100) return;
  const StepEntry e = tab[*ctr];
  const StepScalars sc = e.sc;
  const float *noise = e.has_noise ? noise_base + e.noise_off : nullptr; // same layout as x
  const uint32_t step = e.philox_step;
  const int T = seq_len[s], t = row_t[r];
  const size_t xi = x_off[s] + (size_t)ch * T + t;
  const float eps_c = net[(size_t)r * 256 + ch], var_c = net[(size_t)r * 256 + 100 + ch];
  const float eps_u = net[(size_t)(seq_start[s + ncand] + t) * 256 + ch];
  const float xv = x[xi];
  // Every f32 operation below is the reference's, one rounding each (main.cpp:5970-6030 is plain C++ built without FMA contraction; the oracle's copy is compiled with
  // -ffp-contract=off). hipcc would contract a * b + c * d into FMAs: a last-bit difference in x_t at EVERY step that the torch-f32 yardstick of the parity floor (which
  // shares the oracle's update) does not have, and that a chaotic 80- / 200-step loop amplifies like any other f32 difference (round 6: the 200-step loop at full depth
  // sat at 1.44-1.58 x its f32-vs-f32 floor in BOTH arithmetic modes).
  float mean, model_log_variance;
  {
#pragma clang fp contract(off)
    const float frac = (var_c + 1) / 2;
    // calculate_model_variance is called with (min_log, max_log) swapped (main.cpp:5998-5999)
    model_log_variance = frac * sc.min_log + (1 - frac) * sc.max_log;
    const float eps = (1 + sc.cfk) * eps_c - sc.cfk * eps_u;
    float x0 = sc.sqrt_recip * xv - sc.sqrt_recipm1 * eps;
    x0 = fminf(fmaxf(x0, -1.0f), 1.0f);
    mean = sc.coef1 * x0 + sc.coef2 * xv;
  }
  float outv = mean;
  if (!sc.is_last) {
    const float nz = noise ? noise[xi] : philox_normal(seed, stream0 + (uint32_t)s, step, (uint32_t)(ch * T + t));
    outv = (float)((double)mean + exp(0.5 * (double)model_log_variance) * (double)nz);
  }
  x[xi] = outv;
}

__global__ void philox_fill_kernel(float *__restrict__ x, int64_t n, uint64_t seed, uint32_t stream, uint32_t step) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] = philox_normal(seed, stream, step, (uint32_t)i);
}

// net output [rows][256] -> reference layout [200][T] for one sequence.
__global__ void rows_to_ct_kernel(const float *__restrict__ net, int row0, int T, int nch, int ld, float *__restrict__ out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nch * T) {
    int ch = i / T, t = i - ch * T;
    out[i] = net[(size_t)(row0 + t) * ld + ch];
  }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
struct AttnDev {
  float *norm_g, *norm_b, *qkv_b, *proj_b, *bias_tab;
  __half *qkv_w, *proj_w;
  __half *proj_w_split; // [C][hi(s w) | lo(s w)]: proj_out's F32 weight as a split-precision pair; s = 1 / proj_alpha keeps the low halves normal
  float proj_alpha;     // 1 / s, s = the largest power of two with max|W| s < 30000 (64 for |W| up to 468; round 6: was a fixed 64, whose hi half overflows at |W| > 1023)
};
struct ResDev { float *in_g, *in_b, *in_bias, *emb_w, *emb_b, *out_g, *out_b, *out_bias; __half *in_w, *out_w; };

// Packed row layout: sequence s occupies rows [start[s], start[s]+len[s]); start % 8 == 0; at least one
// zero guard row before and after every sequence; total padded to a multiple of 128.
struct Layout {
  int ns = 0, rows = 0;
  std::vector<int> start, len;
  DevBuf d_row_seq, d_row_t, d_start, d_len, d_chunk_seq; // chunk_seq[r / 8]: owning sequence of an aligned 8-row chunk (-1: guard rows only)
  int build(tts_ctx *ctx, const std::vector<int> &lens) {
    has_seq_step = false; ss_step_stride = 0;
    ns = (int)lens.size();
    len = lens;
    start.resize(ns);
    int r = 8;
    for (int s = 0; s < ns; s++) { start[s] = r; r = (r + lens[s] + 1 + 7) & ~7; }
    rows = (r + 127) & ~127;
    std::vector<int> rs(rows, -1), rt(rows, 0);
    for (int s = 0; s < ns; s++)
      for (int t = 0; t < lens[s]; t++) { rs[start[s] + t] = s; rt[start[s] + t] = t; }
    TTS_HIP(ctx, d_row_seq.reserve(rows * 4)); TTS_HIP(ctx, d_row_t.reserve(rows * 4));
    TTS_HIP(ctx, d_start.reserve(ns * 4)); TTS_HIP(ctx, d_len.reserve(ns * 4));
    TTS_HIP(ctx, hipMemcpy(d_row_seq.p, rs.data(), rows * 4, hipMemcpyHostToDevice));
    TTS_HIP(ctx, hipMemcpy(d_row_t.p, rt.data(), rows * 4, hipMemcpyHostToDevice));
    TTS_HIP(ctx, hipMemcpy(d_start.p, start.data(), ns * 4, hipMemcpyHostToDevice));
    TTS_HIP(ctx, hipMemcpy(d_len.p, len.data(), ns * 4, hipMemcpyHostToDevice));
    std::vector<int> cs(rows / 8, -1);
    for (int s = 0; s < ns; s++)
      for (int t = 0; t < lens[s]; t++) cs[(start[s] + t) >> 3] = s; // start % 8 == 0 and a guard row follows every sequence: one owner per chunk
    TTS_HIP(ctx, d_chunk_seq.reserve(rows / 8 * 4));
    TTS_HIP(ctx, hipMemcpy(d_chunk_seq.p, cs.data(), rows / 8 * 4, hipMemcpyHostToDevice));
    return TTS_OK;
  }
  int max_len() const { return len.empty() ? 0 : *std::max_element(len.begin(), len.end()); }
  // Sequences of DIFFERENT timesteps in one layout (the hoisted conditioning_timestep_integrator, round 6): seq_step[s] = index of sequence s's timestep; the GroupNorm
  // kernels then read sequence s's scale / shift at ss + seq_step[s] * ss_step_stride. Empty = one timestep for the whole layout.
  DevBuf d_seq_step;
  int ss_step_stride = 0;
  bool has_seq_step = false;
  const int *seq_step_ptr() const { return has_seq_step ? d_seq_step.as<int>() : nullptr; }
  int set_seq_step(tts_ctx *ctx, const std::vector<int> &step_of_seq, int stride_floats) {
    TTS_HIP(ctx, d_seq_step.reserve(step_of_seq.size() * 4));
    TTS_HIP(ctx, hipMemcpy(d_seq_step.p, step_of_seq.data(), step_of_seq.size() * 4, hipMemcpyHostToDevice));
    ss_step_stride = stride_floats; has_seq_step = true;
    return TTS_OK;
  }
};

// Activation workspace for one layout. fp16 GEMM operands carry a 1-row zero halo on both sides (the
// k=3 taps read rows -1 and `rows`) plus 128 rows of slack for the attention tiles.
struct Work {
  int rows = 0;
  DevBuf x, hbuf, a16, att16, qk16, vt16, stats;
  DevBuf att16_lo, qk16_lo, vt16_lo; // low halves of the split-precision pairs (option attn_f32)
  // option latency_mode: fixed-point GroupNorm statistics (gemm_f16.h: fx_add) of the tensor the blocks of this layout work on (X or the code embedding) and of H,
  // left by the epilogue of the GEMM that produced them; nullptr = not available (the batch path reduces them in the GroupNorm kernel)
  long long *st_x = nullptr, *st_h = nullptr;
  float *X() { return x.as<float>(); }
  float *H() { return hbuf.as<float>(); }
  __half *A16() { return a16.as<__half>() + C; }
  __half *ATT16() { return att16.as<__half>() + C; }
  int reserve(tts_ctx *ctx, int r, int ns, bool split = false) { // split: the low-half buffers of the reference-precision AttentionBlock are needed whatever attn_f32 says
    auto rz = [&](DevBuf &b, size_t bytes) -> hipError_t {
      size_t old = b.cap;
      hipError_t e = b.reserve(bytes);
      if (e == hipSuccess && b.cap != old) e = hipMemset(b.p, 0, b.cap);
      return e;
    };
    rows = r;
    TTS_HIP(ctx, rz(x, (size_t)r * C * 4));
    TTS_HIP(ctx, rz(hbuf, (size_t)r * C * 4));
    TTS_HIP(ctx, rz(a16, (size_t)(r + 2) * C * 2));
    TTS_HIP(ctx, rz(att16, (size_t)(r + 2) * C * 2));
    TTS_HIP(ctx, rz(qk16, (size_t)(r + 128) * 2048 * 2));
    TTS_HIP(ctx, rz(vt16, (size_t)C * (r + 128) * 2));
    TTS_HIP(ctx, rz(stats, (size_t)ns * 32 * sizeof(float2)));
    if (ctx->attn_f32 || split) {
      TTS_HIP(ctx, rz(att16_lo, (size_t)(r + 2) * C * 2));
      TTS_HIP(ctx, rz(qk16_lo, (size_t)(r + 128) * 2048 * 2));
      TTS_HIP(ctx, rz(vt16_lo, (size_t)C * (r + 128) * 2));
    }
    return TTS_OK;
  }
};

struct DiffState {
  int n_lc = 0, n_integ = 0, n_main = 0, n_tail = 0;
  std::vector<AttnDev> lc_attn, integ_attn, main_attn;
  std::vector<ResDev> integ_res, main_res, tail_res; // emb order: integ.., main.., tail..
  float *cond_latent = nullptr, *uncond_emb = nullptr, *lc_bias = nullptr, *code_g = nullptr, *code_b = nullptr;
  float *te0_w = nullptr, *te0_b = nullptr, *te2_w = nullptr, *te2_b = nullptr;
  float *inp_bias = nullptr, *integ_bias = nullptr, *outn_g = nullptr, *outn_b = nullptr, *out_bias = nullptr;
  __half *lc_w = nullptr, *inp_w = nullptr, *integ_w = nullptr, *out_w = nullptr;
  std::vector<void *> owned;
  // run state
  Layout lay, lat_lay, ilay;
  Work wk, lat_wk, iwk;
  // The conditioning_timestep_integrator stage sees only (code embedding, timestep): for the unconditioned branch its
  // input is the same vector at every position, so unconditioned sequences of equal length give identical results. With
  // share_integ the stage runs on ilay = [conditioned sequences | one unconditioned sequence per distinct length] and
  // its output rows are gathered into the full layout (ce_src: source row per row of `lay`).
  bool share_integ = false;
  DevBuf ce_src, iseq_src;
  DevBuf h0; // in_layers of the first integrator ResBlock applied to the code embedding: the same at every step
  // Hoisted integrator (round 6, small batches): the conditioning_timestep_integrator layers see only (code embedding, timestep) — never x_t (main.cpp:3322-3499) — so
  // their output for ALL sampling steps is evaluated before the loop, many timesteps per batch (play: [timestep][sequence] sequences with per-sequence scale / shift),
  // and a step only selects its slice of ce16_all. Same arithmetic per sequence: bit-identical to evaluating the layers inside every step.
  bool hoisted = false;
  Layout play;
  Work pwk;
  DevBuf pcode, ph0, pce, psrc, pscatter, ce16_all;
  std::vector<int> ce_src_host; // share_integ: source row in ilay of every row of lay
  // option latency_mode (small layouts): per sampling step one statistics slot per f32 GEMM output, zeroed at the start of the step; h0's slot persists
  bool lat = false;
  DevBuf gn_stats, gn_stats_h0;
  int gn_site = 0, gn_sites_max = 0;
  size_t gn_slot_ll = 0, gn_stripe_ll = 0; // long longs per slot = FX_STRIPES stripes x (sequences x 32 groups x 4)
  long long *new_stats_slot() { long long *p = gn_stats.as<long long>() + (size_t)gn_site * gn_slot_ll; gn_site++; return p; }
  DevBuf step_tab, step_ctr, ss_cur; // StepEntry[n_steps] | int step counter | this step's scale/shift block (fixed address)
  hipGraph_t step_graph = nullptr;
  hipGraphExec_t step_exec = nullptr;
  void drop_step_graph() {
    if (step_exec) (void)hipGraphExecDestroy(step_exec);
    if (step_graph) (void)hipGraphDestroy(step_graph);
    step_exec = nullptr; step_graph = nullptr;
  }
  DevBuf code_emb, ce, ce16, xt16, inp16, net, temb, e1, emb, ss_all, ss_chk, xbuf, xoff, noise, seq_src, lat_in16, out_ct;
  PinnedBuf noise_host; // reference-order noise drawn step by step beside the device loop (diff_sample)
  ~DiffState() { drop_step_graph(); for (void *p : owned) (void)hipFree(p); }
  int n_res() const { return n_integ + n_main + n_tail; }
};

void diff_free(DiffState *s) { delete s; }
int diff_layers(const tts_ctx *ctx) { return ctx->diff ? ctx->diff->n_main : 0; }
// The voice's diffusion conditioning latent is a WEIGHT of the reference's file (main.cpp:1557-1560: one voice per ggml-diffusion-model.bin);
// this replaces it in the loaded model, e.g. with the output of tts_diffusion_conditioning_latent.
int diff_set_cond_latent(tts_ctx *ctx, const float *latent2048) {
  if (!ctx->diff) return fail(ctx, TTS_ERR_STATE, "tts_load_diffusion not called");
  if (!latent2048) return fail(ctx, TTS_ERR_ARG, "null latent");
  TTS_HIP(ctx, hipMemcpyAsync(ctx->diff->cond_latent, latent2048, (size_t)2 * C * 4, hipMemcpyHostToDevice, ctx->stream));
  TTS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return TTS_OK;
}

namespace {
struct Loader { // its jobs run on several threads (common.h: run_parallel): `used`, `owned` and the bad-weight counters are touched under `mu`
  tts_ctx *ctx; DiffState *st; const WeightFile &wf; std::map<std::string, bool> used; std::mutex mu; PinnedPool pin{(size_t)9 << 20, 4};
  const HostTensor *get(const std::string &name, int64_t nelem) {
    auto it = wf.t.find(name);
    if (it == wf.t.end()) { fail(ctx, TTS_ERR_FORMAT, "tensor '%s' missing from diffusion model file", name.c_str()); return nullptr; }
    if (it->second.nelem() != nelem) {
      fail(ctx, TTS_ERR_FORMAT, "tensor '%s' has wrong size in model file: got %lld, expected %lld", name.c_str(),
           (long long)it->second.nelem(), (long long)nelem);
      return nullptr;
    }
    { std::lock_guard<std::mutex> lk(mu); used[name] = true; }
    return &it->second;
  }
  template <class T> int put(const std::vector<T> &h, T **dst) {
    void *p = nullptr;
    TTS_HIP(ctx, hipMalloc(&p, h.size() * sizeof(T)));
    { std::lock_guard<std::mutex> lk(mu); st->owned.push_back(p); }
    TTS_HIP(ctx, pin.upload(p, h.data(), h.size() * sizeof(T), ctx->load_stream));
    *dst = (T *)p;
    return TTS_OK;
  }
  int f32(const std::string &name, int64_t n, float **dst) {
    const HostTensor *t = get(name, n);
    if (!t) return TTS_ERR_FORMAT;
    return put(t->data, dst);
  }
  // conv weight file layout w[(co*cin + ci)*k + tap] -> fp16 [co_pad][tap*cin_pad + ci] (zero padded)
  int conv16(const std::string &name, int cout, int cin, int k, int cout_pad, int cin_pad, __half **dst) {
    const HostTensor *t = get(name, (int64_t)cout * cin * k);
    if (!t) return TTS_ERR_FORMAT;
    std::vector<__half> h((size_t)cout_pad * k * cin_pad, __float2half(0.f));
    const float *w = t->data.data();
    for (int co = 0; co < cout; co++)
      for (int ci = 0; ci < cin; ci++)
        for (int tap = 0; tap < k; tap++)
          h[(size_t)co * k * cin_pad + (size_t)tap * cin_pad + ci] = __float2half_rn(w[((size_t)co * cin + ci) * k + tap]);
    return put(h, dst);
  }
  int attn(const std::string &p, AttnDev &a) {
    int r;
    if ((r = f32(p + ".norm.weight", C, &a.norm_g))) return r;
    if ((r = f32(p + ".norm.bias", C, &a.norm_b))) return r;
    if ((r = conv16(p + ".qkv.weight", 3 * C, C, 1, 3 * C, C, &a.qkv_w))) return r;
    if ((r = f32(p + ".qkv.bias", 3 * C, &a.qkv_b))) return r;
    if ((r = conv16(p + ".proj_out.weight", C, C, 1, C, C, &a.proj_w))) return r;
    {
      const HostTensor *t = get(p + ".proj_out.weight", (int64_t)C * C);
      std::vector<__half> sp((size_t)C * 2 * C);
      float amax = 0.f;
      for (float v : t->data) amax = std::max(amax, std::fabs(v));
      if (!std::isfinite(amax)) { fail(ctx, TTS_ERR_FORMAT, "tensor '%s.proj_out.weight' holds a non-finite value", p.c_str()); return TTS_ERR_FORMAT; }
      // per-tensor power-of-two scale: as large as keeps the hi half far from fp16's 65504 (the low halves then sit as far above the subnormals as they can);
      // capped at 2^14 (a weight tensor of zeros would ask for infinity)
      int e = 14;
      while (e > -14 && std::ldexp(amax, e) >= 30000.0f) e--;
      const float scale = std::ldexp(1.0f, e);
      a.proj_alpha = 1.0f / scale;
      for (int n = 0; n < C; n++)
        for (int k = 0; k < C; k++) {
          const float w = t->data[(size_t)n * C + k] * scale;
          const __half hi = __float2half_rn(w);
          sp[(size_t)n * 2 * C + k] = hi;
          sp[(size_t)n * 2 * C + C + k] = __float2half_rn(w - __half2float(hi));
        }
      long long bad[2] = {0, 0};
      for (const __half &h : sp) {
        const float v = std::fabs(__half2float(h));
        if (!(v <= 65504.0f)) bad[0]++;
        else if (v > 60000.0f) bad[1]++;
      }
      if (bad[0] | bad[1]) { std::lock_guard<std::mutex> lk(mu); ctx->fp16_bad_weights[0] += bad[0]; ctx->fp16_bad_weights[1] += bad[1]; }
      if ((r = put(sp, &a.proj_w_split))) return r;
    }
    if ((r = f32(p + ".proj_out.bias", C, &a.proj_b))) return r;
    const HostTensor *t = get(p + ".relative_pos_embeddings.relative_attention_bias.weight", 32 * 16);
    if (!t) return TTS_ERR_FORMAT;
    // bias(q,k) = 8 * table[bucket(q,k)][h]; expanded per signed distance: tab[h][(q<k)*64 + min(|k-q|,63)]
    std::vector<float> tab(16 * 128);
    for (int h = 0; h < 16; h++)
      for (int sgn = 0; sgn < 2; sgn++)
        for (int d = 0; d < 64; d++) {
          int bucket = sgn ? rel_bucket(0, d) : rel_bucket(d, 0);
          if (d == 0) bucket = rel_bucket(0, 0);
          tab[h * 128 + sgn * 64 + d] = t->data[(size_t)bucket * 16 + h] * 8.0f;
        }
    return put(tab, &a.bias_tab);
  }
  int res(const std::string &p, ResDev &w) {
    int r;
    if ((r = f32(p + ".in_layers.0.weight", C, &w.in_g))) return r;
    if ((r = f32(p + ".in_layers.0.bias", C, &w.in_b))) return r;
    if ((r = conv16(p + ".in_layers.2.weight", C, C, 1, C, C, &w.in_w))) return r;
    if ((r = f32(p + ".in_layers.2.bias", C, &w.in_bias))) return r;
    if ((r = f32(p + ".emb_layers.1.weight", 2 * C * C, &w.emb_w))) return r;
    if ((r = f32(p + ".emb_layers.1.bias", 2 * C, &w.emb_b))) return r;
    if ((r = f32(p + ".out_layers.0.weight", C, &w.out_g))) return r;
    if ((r = f32(p + ".out_layers.0.bias", C, &w.out_b))) return r;
    if ((r = conv16(p + ".out_layers.3.weight", C, C, 3, C, C, &w.out_w))) return r;
    if ((r = f32(p + ".out_layers.3.bias", C, &w.out_bias))) return r;
    return TTS_OK;
  }
};
} // namespace

int diff_load(tts_ctx *ctx, const char *path) {
  ctx->fp16_bad_weights[0] = ctx->fp16_bad_weights[1] = 0;
  WeightFile wf;
  std::string err;
  int rc = read_weight_file(path, wf, err);
  if (rc != TTS_OK) return fail(ctx, rc, "diffusion_model_load: %s", err.c_str());
  std::unique_ptr<DiffState> st(new DiffState());
  Loader ld{ctx, st.get(), wf, {}};
  while (wf.has("latent_conditioner." + std::to_string(st->n_lc + 1) + ".norm.weight")) st->n_lc++;
  while (wf.has("conditioning_timestep_integrator." + std::to_string(st->n_integ) + ".resblk.in_layers.0.weight")) st->n_integ++;
  while (wf.has("layers." + std::to_string(st->n_main) + ".resblk.in_layers.0.weight")) st->n_main++;
  while (wf.has("layers." + std::to_string(st->n_main + st->n_tail) + ".in_layers.0.weight")) st->n_tail++;
  st->lc_attn.resize(st->n_lc);
  st->integ_res.resize(st->n_integ); st->integ_attn.resize(st->n_integ);
  st->main_res.resize(st->n_main); st->main_attn.resize(st->n_main);
  st->tail_res.resize(st->n_tail);
  // Every block's fp16 re-layout + upload is one job (round 6: the jobs run on several threads, tts_load_diffusion 1.2 s -> see DESIGN.md section 5)
  std::vector<std::function<int()>> jobs;
  DiffState *S = st.get();
#define J(x) jobs.emplace_back([&ld, S]() -> int { (void)S; return (x); })
  J(ld.f32("diffusion_conditioning_latent", 2 * C, &S->cond_latent));
  J(ld.f32("unconditioned_embedding", C, &S->uncond_emb));
  J(ld.conv16("latent_conditioner.0.weight", C, C, 3, C, C, &S->lc_w));
  J(ld.f32("latent_conditioner.0.bias", C, &S->lc_bias));
  for (int i = 0; i < st->n_lc; i++) jobs.emplace_back([&ld, S, i]() { return ld.attn("latent_conditioner." + std::to_string(i + 1), S->lc_attn[i]); });
  J(ld.f32("code_norm.weight", C, &S->code_g));
  J(ld.f32("code_norm.bias", C, &S->code_b));
  J(ld.f32("time_embed.0.weight", C * C, &S->te0_w)); J(ld.f32("time_embed.0.bias", C, &S->te0_b));
  J(ld.f32("time_embed.2.weight", C * C, &S->te2_w)); J(ld.f32("time_embed.2.bias", C, &S->te2_b));
  for (int i = 0; i < st->n_integ; i++) {
    jobs.emplace_back([&ld, S, i]() { return ld.res("conditioning_timestep_integrator." + std::to_string(i) + ".resblk", S->integ_res[i]); });
    jobs.emplace_back([&ld, S, i]() { return ld.attn("conditioning_timestep_integrator." + std::to_string(i) + ".attn", S->integ_attn[i]); });
  }
  J(ld.conv16("inp_block.weight", C, 100, 3, C, XTC, &S->inp_w));
  J(ld.f32("inp_block.bias", C, &S->inp_bias));
  J(ld.conv16("integrating_conv.weight", C, 2 * C, 1, C, 2 * C, &S->integ_w));
  J(ld.f32("integrating_conv.bias", C, &S->integ_bias));
  for (int i = 0; i < st->n_main; i++) {
    jobs.emplace_back([&ld, S, i]() { return ld.res("layers." + std::to_string(i) + ".resblk", S->main_res[i]); });
    jobs.emplace_back([&ld, S, i]() { return ld.attn("layers." + std::to_string(i) + ".attn", S->main_attn[i]); });
  }
  for (int i = 0; i < st->n_tail; i++) jobs.emplace_back([&ld, S, i]() { return ld.res("layers." + std::to_string(S->n_main + i), S->tail_res[i]); });
  J(ld.f32("out.0.weight", C, &S->outn_g)); J(ld.f32("out.0.bias", C, &S->outn_b));
  J(ld.conv16("out.2.weight", 200, C, 3, 256, C, &S->out_w));
  jobs.emplace_back([&ld, S]() -> int {
    const HostTensor *t = ld.get("out.2.bias", 200);
    if (!t) return TTS_ERR_FORMAT;
    std::vector<float> b(256, 0.f);
    std::copy(t->data.begin(), t->data.end(), b.begin());
    return ld.put(b, &S->out_bias);
  });
#undef J
  if (int r = run_parallel(ctx, (int)jobs.size(), [&](int i) { return jobs[i](); })) return r;
  for (auto &kv : wf.t)
    if (!ld.used.count(kv.first)) return fail(ctx, TTS_ERR_FORMAT, "unknown tensor '%s' in model file", kv.first.c_str());
  TTS_HIP(ctx, hipFuncSetAttribute((const void *)diff_attn_f32_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, ATT32_LDS)); // > 64 KB of dynamic LDS
  if (ctx->diff) diff_free(ctx->diff);
  ctx->diff = st.release();
  return TTS_OK;
}

#define CHECK(x) do { int _r = (x); if (_r) return _r; } while (0)

// algorithmic FLOPs of one launch: valid rows (no guard/pad rows) x valid columns x valid K
#ifdef TTS_DEBUG_CHECKSUM // developer build: hash of a buffer after the launch that produced it, one line per call on stderr (tools/determinism_probe*.py)
static void dbg_sum(tts_ctx *ctx, const char *tag, const void *p, size_t bytes) {
  std::vector<uint8_t> h(bytes);
  (void)hipMemcpyAsync(h.data(), p, bytes, hipMemcpyDeviceToHost, ctx->stream);
  (void)hipStreamSynchronize(ctx->stream);
  uint64_t x = 1469598103934665603ull;
  for (size_t i = 0; i < bytes; i++) x = (x ^ h[i]) * 1099511628211ull;
  fprintf(stderr, "[cks] %s %016llx\n", tag, (unsigned long long)x);
  if (!strncmp(tag, "time", 4)) { // the small f32 vectors of the time MLP: keep them to show WHERE two runs differ
    static std::map<std::string, std::vector<float>> first;
    const float *f = (const float *)h.data();
    const size_t n = bytes / 4;
    auto it = first.find(tag);
    if (it == first.end()) first[tag].assign(f, f + n);
    else {
      size_t nd = 0, i0 = 0; float worst = 0.f;
      for (size_t i = 0; i < n; i++) if (memcmp(&f[i], &it->second[i], 4)) { if (!nd) i0 = i; nd++; worst = std::max(worst, fabsf(f[i] - it->second[i])); }
      if (nd) fprintf(stderr, "[dif] %s: %zu of %zu floats differ from the first call, first at %zu (%.9g vs %.9g), largest difference %.3g\n", tag, nd, n, i0, f[i0], it->second[i0], worst);
    }
  }
}
#define DBG_SUM(tag, p, bytes) dbg_sum(ctx, tag, p, bytes)
#else
#define DBG_SUM(tag, p, bytes)
#endif
// option fp16_check: scan an fp16 operand buffer right after the launch that wrote it
static int fp16_check(tts_ctx *ctx, const void *p, size_t n_halves) {
  if (!ctx->fp16_check) return TTS_OK;
  if (!ctx->fp16_counts) {
    TTS_HIP(ctx, hipMalloc(&ctx->fp16_counts, 16));
    TTS_HIP(ctx, hipMemset(ctx->fp16_counts, 0, 16));
  }
  fp16_scan_kernel<<<512, 256, 0, ctx->stream>>>((const __half *)p, n_halves, (unsigned long long *)ctx->fp16_counts);
  TTS_HIP(ctx, hipGetLastError());
  return TTS_OK;
}
int diff_fp16_check(tts_ctx *ctx, int64_t counts[2]) {
  int64_t dev[2] = {0, 0};
  if (ctx->fp16_counts) {
    TTS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    TTS_HIP(ctx, hipMemcpy(dev, ctx->fp16_counts, 16, hipMemcpyDeviceToHost));
  }
  counts[0] = dev[0] + ctx->fp16_bad_weights[0];
  counts[1] = dev[1] + ctx->fp16_bad_weights[1];
  return TTS_OK;
}

static int gemm(tts_ctx *ctx, const char *fam, GemmArgs &g, const Layout &lay, int n_valid = 0, int k_valid = 0) {
  double mv = 0;
  for (int l : lay.len) mv += l;
  // (GroupNorm statistics fused into this epilogue were tried and measured slower: +2.7 ms/step of GEMM
  //  time for 1.0 ms/step of gn_stats saved — register pressure costs a resident workgroup per CU.)
  static const bool log_shapes = getenv("TTS_GEMM_LOG") != nullptr; // developer aid: one line per launch
  if (log_shapes) fprintf(stderr, "gemm M=%d N=%d K=%dx%d mode=%d resid=%d\n", g.M, g.N, g.nseg, g.kseg, g.mode, g.resid != nullptr);
  // profiling sub-family by shape class (bench.py's per-kernel roofline table: each class has its own MFMA / HBM bound)
  if (!strcmp(fam, "diff_gemm")) {
    if (g.mode == GEMM_OUT_QKV) fam = "diff_gemm_qkv";                                              // N = 3072, K = 1024, fp16 out
    else if (gemm_is_conv3(g) && g.N == C && g.kseg == C) fam = g.resid ? "diff_gemm_k3r" : "diff_gemm_k3"; // out_layers / latent conditioner conv
    else if (g.nseg == 1 && g.N == C && g.kseg == C && g.mode == GEMM_OUT_F32) fam = g.resid ? "diff_gemm_k1r" : "diff_gemm_k1"; // proj_out (option attn_proj_f16) / in_layers
    else if (g.custom_w && g.N == C && g.kseg == C && g.mode == GEMM_OUT_F32_SCALED && g.resid) fam = "diff_gemm_k1r";                   // proj_out on a split-precision weight (k_valid = C: the product's own FLOPs)
    else fam = "diff_gemm_misc";                                                                    // inp_block, integrating conv, out head
  }
  ProfScope ps(ctx, fam, 2.0 * mv * (n_valid ? n_valid : g.N) * (k_valid ? k_valid : g.nseg * g.kseg));
  TTS_HIP(ctx, launch_gemm_f16(g, ctx->stream));
  if (ctx->fp16_check) {
    if (gemm_mode_qkv(g.mode)) {
      CHECK(fp16_check(ctx, g.outH, (size_t)g.M * g.ldh));
      CHECK(fp16_check(ctx, g.outVt, (size_t)(g.N / 3) * g.ldvt));
    } else if (g.mode == GEMM_OUT_F16) CHECK(fp16_check(ctx, g.outH, (size_t)g.M * g.ldh));
  }
  return TTS_OK;
}

static GemmArgs gemm_base(const Layout &lay, const __half *A, int lda, int nseg, int kseg, const __half *W, int N,
                          const float *bias) {
  GemmArgs g{};
  for (int i = 0; i < 3; i++) { g.A[i] = A; g.row_off[i] = 0; }
  if (nseg == 3) { g.row_off[0] = -1; g.row_off[1] = 0; g.row_off[2] = 1; }
  g.nseg = nseg; g.kseg = kseg; g.lda = lda; g.W = W; g.M = lay.rows; g.N = N; g.bias = bias;
  g.row_seq = lay.d_row_seq.as<int>();
  return g;
}

static int gn_stats(tts_ctx *ctx, const Layout &lay, Work &wk, const float *x) {
  ProfScope ps(ctx, "diff_gn_stats");
  gn_stats_kernel<<<dim3(32, lay.ns), 256, 0, ctx->stream>>>(x, lay.d_start.as<int>(), lay.d_len.as<int>(), ctx->gn_eps,
                                                             wk.stats.as<float2>());
  TTS_HIP(ctx, hipGetLastError());
  return TTS_OK;
}
// Register-resident variant: one workgroup of NT threads = one (sequence, 32-channel group); the [T][32] slab
// (T <= NJ * NT / 8 rows) is read from HBM exactly once into NJ float4 per thread, mean and centred variance are
// reduced across the workgroup (two-pass on registers), and the normalised fp16 rows are written straight out.
template <int NT, int NJ>
__global__ __launch_bounds__(NT) void gn_reg_kernel(const float *__restrict__ x, const int *__restrict__ seq_start,
                                                    const int *__restrict__ seq_len, int rows_total, int ns, float eps,
                                                    const float *__restrict__ g, const float *__restrict__ b,
                                                    const float *__restrict__ ss, int do_silu, int lut, __half *__restrict__ y,
                                                    const char *__restrict__ pf0, int pf0_lines, const char *__restrict__ pf1, int pf1_lines,
                                                    const int *__restrict__ seq_step, int ss_step_stride) {
  constexpr int NW = NT / 64, SWEEP = NT / 8;
  __shared__ float sh[2][NW];
  __shared__ unsigned pf_sink[NW][64]; // landing zone of the weight touch below
  // workgroup b runs on XCD b % 8: give each XCD whole sequences (all 32 groups of a row = the full 4 KB row go
  // through one L2) instead of 128-byte slices of every row
  const int bid = blockIdx.y * 32 + blockIdx.x, nbk = ns * 32, item = (bid & 7) * (nbk >> 3) + (bid >> 3); // nbk % 8 == 0 (32 groups)
  const int s = item >> 5, grp = item & 31;
  const int T = seq_len[s], r0 = seq_start[s];
  const int q = threadIdx.x & 7, c = grp * 32 + q * 4, t0 = threadIdx.x >> 3;
  const float *base = x + (size_t)r0 * C + c;
  float4 v[NJ];
#pragma unroll
  for (int j = 0; j < NJ; j++) {
    const int t = t0 + j * SWEEP;
    v[j] = *(const float4 *)(base + (size_t)min(t, T - 1) * C); // clamped rows are masked out of the sums below
  }
  // Weight touch for the GEMM(s) that consume this kernel's output: one dword of every 128-byte line of their weight matrices, spread
  // over all threads of the launch, requested right behind the slab (LDS-DMA into a sink: no register is waiting for the data). The
  // 0.36 GB of fp16 weights cycle through the 256 MB memory-side cache once per sampling step, so without it every K tile of the next
  // GEMM is a miss to HBM for all of its workgroups at once (they walk K in lockstep): a k = 1 GEMM of one utterance takes 24 us with
  // cold weights against 14 us when they sit in the memory-side cache (profiles/r2_gemm_small_problems.txt, round-3 addendum).
  {
    const int gi = (blockIdx.y * 32 + blockIdx.x) * NT + threadIdx.x, tt = 32 * ns * NT;
    unsigned *sink = &pf_sink[__builtin_amdgcn_readfirstlane(threadIdx.x >> 6)][0];
    for (int l = gi; l < pf0_lines; l += tt) __builtin_amdgcn_global_load_lds((gptr_t)(pf0 + (size_t)l * 128), (lptr_t)sink, 4, 0, 0);
    for (int l = gi; l < pf1_lines; l += tt) __builtin_amdgcn_global_load_lds((gptr_t)(pf1 + (size_t)l * 128), (lptr_t)sink, 4, 0, 0);
  }
  const float4 gg = *(const float4 *)(g + c), bb = *(const float4 *)(b + c);
  float sc4[4] = {1.f, 1.f, 1.f, 1.f}, sh4[4] = {0.f, 0.f, 0.f, 0.f};
  if (ss) {
    if (seq_step) ss += (size_t)seq_step[s] * ss_step_stride; // this sequence's timestep (the integrator evaluated for many timesteps in one batch)
#pragma unroll
    for (int i = 0; i < 4; i++) { sc4[i] = ss[c + i] + 1.0f; sh4[i] = ss[C + c + i]; }
  }
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < NJ; j++)
    if (t0 + j * SWEEP < T) sum += (v[j].x + v[j].y) + (v[j].z + v[j].w);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
  if ((threadIdx.x & 63) == 0) sh[0][threadIdx.x >> 6] = sum;
  __syncthreads();
  float tot = 0.f;
#pragma unroll
  for (int w = 0; w < NW; w++) tot += sh[0][w];
  const float n = (float)T * 32.f, mean = tot / n;
  float sq = 0.f;
#pragma unroll
  for (int j = 0; j < NJ; j++) {
    v[j].x -= mean; v[j].y -= mean; v[j].z -= mean; v[j].w -= mean;
    if (t0 + j * SWEEP < T) sq += (v[j].x * v[j].x + v[j].y * v[j].y) + (v[j].z * v[j].z + v[j].w * v[j].w);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
  if ((threadIdx.x & 63) == 0) sh[1][threadIdx.x >> 6] = sq;
  __syncthreads();
  float tsq = 0.f;
#pragma unroll
  for (int w = 0; w < NW; w++) tsq += sh[1][w];
  const float rstd = 1.0f / sqrtf(tsq / n + eps);
  const float ge[4] = {gg.x, gg.y, gg.z, gg.w}, be[4] = {bb.x, bb.y, bb.z, bb.w};
  // the activation mode is workgroup-uniform: one copy of the unrolled store loop per mode, no per-element branch
  auto apply = [&](auto mode) {
    constexpr int MODE = decltype(mode)::value; // 0 none, 1 SiLU (hardware exp2 / rcp), 2 SiLU through the fp16 table emulation, 3 SiLU exact (libm expf, IEEE division)
    // Straight-line stores: a row past the sequence end was loaded from (and is written back to) row T - 1 — the same value from
    // every thread that holds it. Under `if (t < T)` hipcc's wait-count pass loses track at every exec-mask join and puts a
    // vmcnt(0) in front of each block, i.e. each of the NJ stores waited for the previous one's acknowledgement.
#pragma unroll
    for (int j = 0; j < NJ; j++) {
      const int t = min(t0 + j * SWEEP, T - 1);
      float e[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
#pragma unroll
      for (int i = 0; i < 4; i++) {
        float u = e[i] * rstd;
        u = u * ge[i];
        u = u + be[i];
        u = u * sc4[i]; // (1, 0 without scale/shift: exact no-ops)
        u = u + sh4[i];
        if (MODE) u = silu_dev(u, MODE - 1);
        e[i] = u;
      }
      const __half2 p0 = __floats2half2_rn(e[0], e[1]), p1 = __floats2half2_rn(e[2], e[3]);
      uint2 o;
      o.x = *(const unsigned *)&p0;
      o.y = *(const unsigned *)&p1;
      *(uint2 *)(y + (size_t)(r0 + t) * C + c) = o;
    }
  };
  if (!do_silu) apply(std::integral_constant<int, 0>{});
  else if (lut == 0) apply(std::integral_constant<int, 1>{});
  else if (lut == 1) apply(std::integral_constant<int, 2>{});
  else apply(std::integral_constant<int, 3>{});
  // zero the guard/padding rows that follow this sequence (and those before the first one)
  const int gend = (s + 1 < ns) ? seq_start[s + 1] : rows_total;
  for (int r = r0 + T + t0; r < gend; r += SWEEP) *(uint2 *)(y + (size_t)r * C + c) = make_uint2(0u, 0u);
  if (s == 0)
    for (int r = t0; r < r0; r += SWEEP) *(uint2 *)(y + (size_t)r * C + c) = make_uint2(0u, 0u);
  if (pf0_lines | pf1_lines) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the sink belongs to this workgroup until its last weight touch has landed
}

// stats + apply in one launch (see gn_fused_kernel). wa / wb: weight matrices (bytes) of the GEMMs that follow, touched into the
// memory-side cache by the register-resident kernel (see there); nullptr / 0: none.
static int gn_fused(tts_ctx *ctx, const Layout &lay, const float *x, const float *g, const float *b, const float *ss, int do_silu,
                    __half *y, const void *wa = nullptr, size_t wa_bytes = 0, const void *wb = nullptr, size_t wb_bytes = 0) {
  ProfScope ps(ctx, "diff_gn_fused");
  const int tmax = lay.max_len();
  // measured (round 3): one utterance 165.2 -> 156.6 ms per diffusion stage with the touch; the 16-candidate batch 864.8 -> 874.0 ms
  // (its GEMMs re-use every weight line from thousands of tiles: the touch only adds requests) -> small problems only
  if (lay.rows > 4096) { wa_bytes = 0; wb_bytes = 0; }
  const int silu_mode = ctx->ggml_lut ? 1 : ctx->attn_f32 ? 2 : 0; // see silu_dev
#define GN_ARGS x, lay.d_start.as<int>(), lay.d_len.as<int>(), lay.rows, lay.ns, ctx->gn_eps, g, b, ss, do_silu, silu_mode, y, \
                (const char *)wa, (int)(wa_bytes >> 7), (const char *)wb, (int)(wb_bytes >> 7), lay.seq_step_ptr(), lay.ss_step_stride
  if (tmax <= 14 * 64) gn_reg_kernel<512, 14><<<dim3(32, lay.ns), 512, 0, ctx->stream>>>(GN_ARGS);
  else if (tmax <= 18 * 128) gn_reg_kernel<1024, 18><<<dim3(32, lay.ns), 1024, 0, ctx->stream>>>(GN_ARGS);
  else gn_fused_kernel<0><<<dim3(32, lay.ns), 256, 0, ctx->stream>>>(x, lay.d_start.as<int>(), lay.d_len.as<int>(), lay.rows, lay.ns, ctx->gn_eps, g, b, ss,
                                                                    do_silu, silu_mode, y, lay.seq_step_ptr(), lay.ss_step_stride); // two sweeps over global memory
#undef GN_ARGS
  TTS_HIP(ctx, hipGetLastError());
  return TTS_OK;
}

// GroupNorm with the statistics already reduced (option latency_mode): the epilogue of the GEMM that produced x left sum / sum of squares per (sequence, group) in
// fixed point (gemm_f16.h: GEMM_OUT_*_STATS, fx_add), so nothing here waits for a reduction over the sequence. One workgroup = 4 packed rows x 1024 channels (half an
// aligned 8-row chunk: one sequence): 448 workgroups for one utterance against the 64 (one per sequence and group, each streaming its 111 KB slab alone) of
// gn_reg_kernel, whose 8.7 us per launch x 43 launches are 18 % of the single-utterance sampling step. Same arithmetic per element, in the same order, as
// gn_reg_kernel; mean / variance from E[x] and E[x^2] - E[x]^2 in f64 (the sums are exact integers) instead of its two-pass f32 form. Every row of the layout is
// written (guard rows: zeros). Weight touch as in gn_reg_kernel.
__global__ __launch_bounds__(256) void gn_apply_kernel(const float *__restrict__ x, const int *__restrict__ chunk_seq, const int *__restrict__ seq_start,
                                                       const int *__restrict__ seq_len, const long long *__restrict__ st, int stripe_ll, float eps, const float *__restrict__ g,
                                                       const float *__restrict__ b, const float *__restrict__ ss, int do_silu, int lut, __half *__restrict__ y,
                                                       const char *__restrict__ pf0, int pf0_lines, const char *__restrict__ pf1, int pf1_lines,
                                                       const int *__restrict__ seq_step, int ss_step_stride) {
  __shared__ float2 mr[32];
  __shared__ unsigned pf_sink[4][64];
  const int r0 = blockIdx.x * 4, s = chunk_seq[blockIdx.x >> 1], c = threadIdx.x * 4;
  float4 v[4];
#pragma unroll
  for (int r = 0; r < 4; r++) v[r] = *(const float4 *)(x + (size_t)(r0 + r) * C + c);
  {
    const int gi = blockIdx.x * 256 + threadIdx.x, tt = gridDim.x * 256;
    unsigned *sink = &pf_sink[__builtin_amdgcn_readfirstlane(threadIdx.x >> 6)][0];
    for (int l = gi; l < pf0_lines; l += tt) __builtin_amdgcn_global_load_lds((gptr_t)(pf0 + (size_t)l * 128), (lptr_t)sink, 4, 0, 0);
    for (int l = gi; l < pf1_lines; l += tt) __builtin_amdgcn_global_load_lds((gptr_t)(pf1 + (size_t)l * 128), (lptr_t)sink, 4, 0, 0);
  }
  int nvalid = 0; // rows of this workgroup inside the sequence (its start is chunk-aligned: only the end can fall inside)
  if (s >= 0) {
    nvalid = min(max(seq_start[s] + seq_len[s] - r0, 0), 4);
    if (threadIdx.x < 32) {
      long long a[4] = {0, 0, 0, 0};
      longlong2 u[FX_STRIPES][2];
#pragma unroll
      for (int k = 0; k < FX_STRIPES; k++) { // all stripes requested at once: one round trip
        const long long *sp = st + (size_t)k * stripe_ll + (size_t)(s * 32 + threadIdx.x) * 4;
        u[k][0] = *(const longlong2 *)sp; u[k][1] = *(const longlong2 *)(sp + 2);
      }
#pragma unroll
      for (int k = 0; k < FX_STRIPES; k++) { a[0] += u[k][0].x; a[1] += u[k][0].y; a[2] += u[k][1].x; a[3] += u[k][1].y; } // integer sums: exact, any order
      const double n = (double)seq_len[s] * 32.0;
      const double mean = fx_value(a[0], a[1]) / n;
      const double var = fmax(fx_value(a[2], a[3]) / n - mean * mean, 0.0);
      mr[threadIdx.x] = make_float2((float)mean, (float)(1.0 / sqrt(var + (double)eps)));
    }
  }
  __syncthreads();
  const float2 m = mr[c >> 5];
  const float4 gg = *(const float4 *)(g + c), bb = *(const float4 *)(b + c);
  float sc4[4] = {1.f, 1.f, 1.f, 1.f}, sh4[4] = {0.f, 0.f, 0.f, 0.f};
  if (ss) {
    if (seq_step && s >= 0) ss += (size_t)seq_step[s] * ss_step_stride;
#pragma unroll
    for (int i = 0; i < 4; i++) { sc4[i] = ss[c + i] + 1.0f; sh4[i] = ss[C + c + i]; }
  }
  const float ge[4] = {gg.x, gg.y, gg.z, gg.w}, be[4] = {bb.x, bb.y, bb.z, bb.w};
  auto apply = [&](auto mode) {
    constexpr int MODE = decltype(mode)::value; // as gn_reg_kernel
#pragma unroll
    for (int r = 0; r < 4; r++) {
      float e[4] = {v[r].x, v[r].y, v[r].z, v[r].w};
#pragma unroll
      for (int i = 0; i < 4; i++) {
        float u = (e[i] - m.x) * m.y;
        u = u * ge[i];
        u = u + be[i];
        u = u * sc4[i];
        u = u + sh4[i];
        if (MODE) u = silu_dev(u, MODE - 1);
        e[i] = r < nvalid ? u : 0.f;
      }
      const __half2 p0 = __floats2half2_rn(e[0], e[1]), p1 = __floats2half2_rn(e[2], e[3]);
      uint2 o;
      o.x = *(const unsigned *)&p0;
      o.y = *(const unsigned *)&p1;
      *(uint2 *)(y + (size_t)(r0 + r) * C + c) = o;
    }
  };
  if (!do_silu) apply(std::integral_constant<int, 0>{});
  else if (lut == 0) apply(std::integral_constant<int, 1>{});
  else if (lut == 1) apply(std::integral_constant<int, 2>{});
  else apply(std::integral_constant<int, 3>{});
  if (pf0_lines | pf1_lines) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// GroupNorm front end: st_x != nullptr (option latency_mode, statistics left by the producing GEMM) -> gn_apply_kernel, else the reducing kernels of gn_fused
static int gn(tts_ctx *ctx, const DiffState *st, const Layout &lay, const float *x, const long long *st_x, const float *g, const float *b, const float *ss, int do_silu,
              __half *y, const void *wa = nullptr, size_t wa_bytes = 0, const void *wb = nullptr, size_t wb_bytes = 0) {
  if (!st_x) {
    CHECK(gn_fused(ctx, lay, x, g, b, ss, do_silu, y, wa, wa_bytes, wb, wb_bytes));
    return fp16_check(ctx, y, (size_t)lay.rows * C);
  }
  {
    ProfScope ps(ctx, "diff_gn_apply");
    const int silu_mode = ctx->ggml_lut ? 1 : ctx->attn_f32 ? 2 : 0; // see silu_dev
    gn_apply_kernel<<<lay.rows / 4, 256, 0, ctx->stream>>>(x, lay.d_chunk_seq.as<int>(), lay.d_start.as<int>(), lay.d_len.as<int>(), st_x, (int)st->gn_stripe_ll, ctx->gn_eps, g, b,
                                                            ss, do_silu, silu_mode, y, (const char *)wa, (int)(wa_bytes >> 7), (const char *)wb, (int)(wb_bytes >> 7),
                                                            lay.seq_step_ptr(), lay.ss_step_stride);
    TTS_HIP(ctx, hipGetLastError());
  }
  return fp16_check(ctx, y, (size_t)lay.rows * C);
}

// in_layers of a ResBlock (GroupNorm, SiLU, conv k=1): H = conv(silu(gn(x))). No timestep dependence.
// st_x: statistics of x (nullptr: reduce them here); st_h_out: where the GEMM's epilogue leaves the statistics of H (nullptr: none) — option latency_mode
static int res_in_layers(tts_ctx *ctx, const DiffState *st, const Layout &lay, Work &wk, const float *x, const ResDev &w, float *H,
                         const long long *st_x = nullptr, long long *st_h_out = nullptr) {
  CHECK(gn(ctx, st, lay, x, st_x, w.in_g, w.in_b, nullptr, 1, wk.A16(), w.in_w, (size_t)C * C * 2));
  DBG_SUM("res.in gn", wk.A16(), (size_t)lay.rows * C * 2);
  GemmArgs g = gemm_base(lay, wk.A16(), C, 1, C, w.in_w, C, w.in_bias);
  g.mode = st_h_out ? GEMM_OUT_F32_STATS : GEMM_OUT_F32; g.outF = H; g.ldo = C; g.resid = nullptr;
  g.st_out = st_h_out; g.st_stripe_ll = (int)st->gn_stripe_ll; g.chunk_seq = lay.d_chunk_seq.as<int>();
  CHECK(gemm(ctx, "diff_gemm", g, lay));
  DBG_SUM("res.in conv", H, (size_t)lay.rows * C * 4);
  return TTS_OK;
}

// AttentionBlock on X (in place). Arithmetic modes (options attn_f32, attn_proj_f16):
//   0  default (north star: "MFMA ... for the dense fp16 GEMMs in attention"): q, k, v, P and the attention output are fp16 MFMA operands,
//      f32 accumulation; proj_out's F32 weight is multiplied as the split pair W_hi + W_lo (two MFMAs per product). Round 5: the five fp16
//      roundings of the rounds 1-4 throughput mode were ablated one at a time in the CPU emulator (tests/golden/parity_floor.json
//      "ablation", tools/regen_parity_floor.py --ablate): the fp16 rounding of the proj_out WEIGHT alone carries the whole distance to the
//      oracle above the f32-vs-f32 floor (full depth, 80 steps: mean 1.16e-4 with it alone, 6.5e-5 with the other four together, 5.6e-5 with
//      none) — it is the one rounding that is the SAME perturbation at every step and in every row; the activation roundings average out.
//      attn_proj_f16 = 1 restores the all-fp16 block of rounds 1-4 (A/B only);
//   1  reference precision (main.cpp:3848-3875: F32 QK^T, softmax, PV and proj_out): the same products on split-precision fp16 pairs
//      (hi + lo, three MFMAs per product, 2^-22 relative) — the parity mode, as ar_weights = 0 is for the AR stage.
static int attention_block(tts_ctx *ctx, DiffState *st, const Layout &lay, Work &wk, float *X, const AttnDev &w, bool force_ref = false) {
  const bool f32 = ctx->attn_f32 != 0 || force_ref;
  const bool pw16 = !f32 && ctx->attn_proj_f16; // weight touch for the two GEMMs that follow: the proj_out matrix this mode will stream
  const bool lat = st->lat && wk.st_x != nullptr; // option latency_mode: X's statistics come from (and go to) the GEMM epilogues
  CHECK(gn(ctx, st, lay, X, lat ? wk.st_x : nullptr, w.norm_g, w.norm_b, nullptr, 0, wk.A16(), w.qkv_w, (size_t)3 * C * C * 2, pw16 ? w.proj_w : w.proj_w_split,
           (size_t)C * C * (pw16 ? 2 : 4)));
  long long *st_out = lat ? st->new_stats_slot() : nullptr;
  wk.st_x = st_out;
  DBG_SUM("attn gn", wk.A16(), (size_t)lay.rows * C * 2);
  GemmArgs g = gemm_base(lay, wk.A16(), C, 1, C, w.qkv_w, 3 * C, w.qkv_b);
  g.mode = f32 ? GEMM_OUT_QKV_SPLIT : GEMM_OUT_QKV;
  g.outH = wk.qk16.as<__half>(); g.ldh = 2048; g.outVt = wk.vt16.as<__half>(); g.ldvt = wk.rows + 128;
  g.outH2 = wk.qk16_lo.as<__half>(); g.outVt2 = wk.vt16_lo.as<__half>();
  CHECK(gemm(ctx, "diff_gemm", g, lay));
  if (f32 && (ctx->attn_f32_drop & 1)) TTS_HIP(ctx, hipMemsetAsync(wk.qk16_lo.p, 0, (size_t)(wk.rows + 128) * 2048 * 2, ctx->stream)); // developer ablation: q, k as ONE fp16 value
  if (f32 && (ctx->attn_f32_drop & 2)) TTS_HIP(ctx, hipMemsetAsync(wk.vt16_lo.p, 0, (size_t)C * (wk.rows + 128) * 2, ctx->stream));    // v as one fp16 value
  DBG_SUM("attn qk", wk.qk16.p, (size_t)lay.rows * 2048 * 2);
  DBG_SUM("attn vt", wk.vt16.p, (size_t)C * (wk.rows + 128) * 2);
  {
    double aw = 0;
    for (int l : lay.len) aw += 4.0 * l * (double)l * 64 * NHEAD; // QK^T + PV
    ProfScope ps(ctx, "diff_attn", aw);
    const int nq = (lay.max_len() + 127) / 128;
    if (f32) {
      diff_attn_f32_kernel<<<nq * NHEAD * lay.ns, 256, ATT32_LDS, ctx->stream>>>(wk.qk16.as<__half>(), wk.qk16_lo.as<__half>(), wk.vt16.as<__half>(),
                                                                                 wk.vt16_lo.as<__half>(), wk.rows + 128, lay.d_start.as<int>(),
                                                                                 lay.d_len.as<int>(), w.bias_tab, wk.ATT16(), wk.att16_lo.as<__half>() + C, nq);
    } else {
      // one K/V tile in flight at 4 workgroups per CU (162.5-163.0 us per launch) beat two tiles in flight at 3 per CU (168.2-168.9 us; round 2). Round 6: the deeper ring
      // does not help one utterance either (224 workgroups, at most one per CU: 20.6 us with one tile in flight, 21.1 with two — profiles/r6_small_batch.txt)
      // 64-query workgroups (option attn_q64: 0 never = default, 1 always, 2 = when the 128-query grid would leave CUs with at most one workgroup): bit-identical, no gain
      const bool q64 = ctx->attn_q64 == 1 || (ctx->attn_q64 == 2 && nq * NHEAD * lay.ns <= 256);
      if (q64) {
        const int nq64 = (lay.max_len() + 63) / 64;
        diff_attn_kernel<2, 1><<<nq64 * NHEAD * lay.ns, 256, att_lds<2>(), ctx->stream>>>(wk.qk16.as<__half>(), wk.vt16.as<__half>(), wk.rows + 128,
                                                                                      lay.d_start.as<int>(), lay.d_len.as<int>(), w.bias_tab, wk.ATT16(), nq64);
      } else
        diff_attn_kernel<2, 2><<<nq * NHEAD * lay.ns, 256, att_lds<2>(), ctx->stream>>>(wk.qk16.as<__half>(), wk.vt16.as<__half>(), wk.rows + 128,
                                                                                    lay.d_start.as<int>(), lay.d_len.as<int>(), w.bias_tab, wk.ATT16(), nq);
    }
    TTS_HIP(ctx, hipGetLastError());
  }
  CHECK(fp16_check(ctx, wk.ATT16(), (size_t)lay.rows * C));
  if (f32 && (ctx->attn_f32_drop & 4)) TTS_HIP(ctx, hipMemsetAsync(wk.att16_lo.p, 0, (size_t)(wk.rows + 2) * C * 2, ctx->stream));     // attention output as one fp16 value
  DBG_SUM("attn out", wk.ATT16(), (size_t)lay.rows * C * 2);
  if (f32) { // att . W^T = att_hi . W_hi + att_lo . W_hi + att_hi . W_lo  (W scaled by 64 at load)
    GemmArgs p = gemm_base(lay, wk.ATT16(), C, 3, C, w.proj_w_split, C, w.proj_b);
    p.A[0] = wk.ATT16(); p.A[1] = wk.att16_lo.as<__half>() + C; p.A[2] = wk.ATT16();
    p.row_off[0] = p.row_off[1] = p.row_off[2] = 0;
    p.custom_w = 1; p.ldw_ = 2 * C; p.w_off_[0] = 0; p.w_off_[1] = 0; p.w_off_[2] = C;
    p.mode = st_out ? GEMM_OUT_F32_SCALED_STATS : GEMM_OUT_F32_SCALED; p.alpha = w.proj_alpha; p.outF = X; p.ldo = C; p.resid = X;
    p.st_out = st_out; p.st_stripe_ll = (int)st->gn_stripe_ll; p.chunk_seq = lay.d_chunk_seq.as<int>();
    return gemm(ctx, "diff_gemm", p, lay, 0, C);
  }
  if (!ctx->attn_proj_f16) { // default: att16 . (W_hi + W_lo)^T — proj_out's F32 weight to 2^-22, the attention output stays one fp16 operand
    GemmArgs p = gemm_base(lay, wk.ATT16(), C, 2, C, w.proj_w_split, C, w.proj_b);
    p.custom_w = 1; p.ldw_ = 2 * C; p.w_off_[0] = 0; p.w_off_[1] = C;
    p.mode = st_out ? GEMM_OUT_F32_SCALED_STATS : GEMM_OUT_F32_SCALED; p.alpha = w.proj_alpha; p.outF = X; p.ldo = C; p.resid = X;
    p.st_out = st_out; p.st_stripe_ll = (int)st->gn_stripe_ll; p.chunk_seq = lay.d_chunk_seq.as<int>();
    p.dual_b = ctx->proj_dual_b; // both weight halves per staged activation tile (gemm_f16_vh_dualb_kernel); 0 = two K segments (A/B)
    CHECK(gemm(ctx, "diff_gemm", p, lay, 0, C));
    DBG_SUM("attn proj", X, (size_t)lay.rows * C * 4);
    return TTS_OK;
  }
  GemmArgs p = gemm_base(lay, wk.ATT16(), C, 1, C, w.proj_w, C, w.proj_b);
  p.mode = st_out ? GEMM_OUT_F32_STATS : GEMM_OUT_F32; p.outF = X; p.ldo = C; p.resid = X;
  p.st_out = st_out; p.st_stripe_ll = (int)st->gn_stripe_ll; p.chunk_seq = lay.d_chunk_seq.as<int>();
  CHECK(gemm(ctx, "diff_gemm", p, lay));
  DBG_SUM("attn proj", X, (size_t)lay.rows * C * 4);
  return TTS_OK;
}

// ResBlock: X = Xin + out_layers(in_layers(Xin) with the step's scale/shift); Xin == nullptr: in place on X.
// ss = this step's [scale | shift] for this block (device, 2048 floats). Hpre: in_layers(Xin) computed earlier (it does
// not depend on the timestep), nullptr: computed here.
// st_hpre: the statistics of Hpre (option latency_mode; the first integrator block's h0 keeps them across the steps)
static int res_block(tts_ctx *ctx, DiffState *st, const Layout &lay, Work &wk, float *X, const ResDev &w, const float *ss,
                     const float *Xin = nullptr, const float *Hpre = nullptr, const long long *st_hpre = nullptr) {
  const float *xin = Xin ? Xin : X;
  const bool lat = st->lat && (Hpre ? st_hpre != nullptr : wk.st_x != nullptr);
  const long long *st_h = st_hpre;
  if (!Hpre) {
    long long *slot = lat ? st->new_stats_slot() : nullptr;
    CHECK(res_in_layers(ctx, st, lay, wk, xin, w, wk.H(), lat ? wk.st_x : nullptr, slot));
    Hpre = wk.H();
    st_h = slot;
  }
  DBG_SUM("res.out ss", ss, (size_t)2 * C * 4);
  DBG_SUM("res.out hpre", Hpre, (size_t)lay.rows * C * 4);
  CHECK(gn(ctx, st, lay, Hpre, lat ? st_h : nullptr, w.out_g, w.out_b, ss, 1, wk.A16(), w.out_w, (size_t)3 * C * C * 2));
  DBG_SUM("res.out gn", wk.A16(), (size_t)lay.rows * C * 2);
  GemmArgs c3 = gemm_base(lay, wk.A16(), C, 3, C, w.out_w, C, w.out_bias);
  long long *st_out = lat ? st->new_stats_slot() : nullptr;
  c3.mode = st_out ? GEMM_OUT_F32_STATS : GEMM_OUT_F32; c3.outF = X; c3.ldo = C; c3.resid = xin;
  c3.st_out = st_out; c3.st_stripe_ll = (int)st->gn_stripe_ll; c3.chunk_seq = lay.d_chunk_seq.as<int>();
  wk.st_x = st_out;
  CHECK(gemm(ctx, "diff_gemm", c3, lay));
  DBG_SUM("res.out conv", X, (size_t)lay.rows * C * 4);
  return TTS_OK;
}

// Sum over the 64 lanes of a wave, every lane gets the total. The same butterfly as `for (o = 32; o; o >>= 1) v += __shfl_xor(v, o)` — the same pairs in the
// same order, so the same bits — but on the gfx950 half/row swap instructions and DPP row rotations instead of ds_bpermute_b32 (the LDS crossbar): after
// the xor-8 step a row's values have period 8, so "lane + 4" holds what "lane ^ 4" holds, and so on down.
template <int CTRL> __device__ __forceinline__ float lnk_dpp(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float wave_sum_dpp(float x) {
  auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  x = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  x = __uint_as_float(b[0]) + __uint_as_float(b[1]);
  x += lnk_dpp<0x128>(x); // row_ror:8
  x += lnk_dpp<0x124>(x); // row_ror:4
  x += lnk_dpp<0x122>(x); // row_ror:2
  x += lnk_dpp<0x121>(x); // row_ror:1
  return x;
}

// out[r][n] = sum_k x[r][k] * W[n][k] + b[n] for small row counts (time MLP, emb_layers), K = 1024:
// one wave per output column — a lane keeps its 16 weights of the column in registers and the rows are walked one at a time (16-byte loads along K, wave
// reduction on permlane swaps + DPP rotations, same pairs in the same order as the shuffle butterfly). F32 exact (reference: F32 mul_mat).
// A workgroup (16 waves x 2 columns) owns 32 CONSECUTIVE columns = one whole 128-byte line of every output row. `act`: the time MLP's SiLU applied by the producer.
// Round 4: the previous form of this kernel (8 clamped rows per pass) compiled to v_pk_fma_f32 (packed f32 FMA with op_sel broadcasts), and THAT instruction
// returned wrong sums — in a wave or two per few hundred launches — whenever a second engine process (MFMA kernels) ran on the same GPU; the same kernel on
// v_fmac_f32 never did (profiles/r4_two_process_determinism.txt: 217 wrong launches of 38 518 against 0 of 37 952, same loads, same clamps, same reduction).
// The sums are the same f32 chains as before, bit for bit; this form has no packed f32 arithmetic (checked in the ISA).
static constexpr int LNK_COLS = 32;
static __global__ __launch_bounds__(1024) void linear_nk_kernel(const float *__restrict__ x, int ldx, int rows, const float *__restrict__ W,
                                                         int N, const float *__restrict__ b, float *__restrict__ out, int ldo,
                                                         int act /*0 none, 1 SiLU (exact expf and division), 2 SiLU through fp16 on both sides (ggml's table)*/) {
  const int lane = threadIdx.x & 63;
  for (int half = 0; half < 2; half++) {
    const int n = blockIdx.x * LNK_COLS + half * 16 + (threadIdx.x >> 6);
    if (n >= N) return;
    const float *wr = W + (size_t)n * C + lane * 4;
    float4 w[4]; // k = 256 j + 4 lane + (0..3)
#pragma unroll
    for (int j = 0; j < 4; j++) w[j] = *(const float4 *)(wr + j * 256);
    const float bn = b ? b[n] : 0.f;
    for (int r = 0; r < rows; r++) {
      const float *xr = x + (size_t)r * ldx + lane * 4;
      float acc = 0.f;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const float4 xv = *(const float4 *)(xr + j * 256);
        acc = fmaf(xv.x, w[j].x, acc); acc = fmaf(xv.y, w[j].y, acc);
        acc = fmaf(xv.z, w[j].z, acc); acc = fmaf(xv.w, w[j].w, acc);
      }
      float v = wave_sum_dpp(acc);
      if (lane == 0) {
        v += bn;
        if (act) { // the time MLP's SiLU, applied by the producer: no address of these vectors ever holds a second version of itself
          if (act == 2) v = __half2float(__float2half_rn(v));
          v = v / (1.f + expf(-v));
          if (act == 2) v = __half2float(__float2half_rn(v));
        }
        out[(size_t)r * ldo + n] = v;
      }
    }
  }
}

// Bitwise comparison of two f32 vectors: *flag = 1 if any word differs (see precompute_time).
static __global__ __launch_bounds__(256) void differ_kernel(const unsigned *__restrict__ a, const unsigned *__restrict__ b, size_t n, int *__restrict__ flag) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    if (a[i] != b[i]) *flag = 1;
}

// Timestep MLP + every emb_layers linear for `n` timesteps at once:
//   emb = W2 silu(W0 te + b0) + b2 (main.cpp:3331-3343); ss[j] = Wemb_j silu(emb) + bemb_j (3410-3428).
static int precompute_time(tts_ctx *ctx, DiffState *st, const std::vector<int> &timesteps) {
  const int n = (int)timesteps.size(), nres = st->n_res();
  std::vector<float> te((size_t)n * C);
  for (int i = 0; i < n; i++) timestep_embedding(timesteps[i], te.data() + (size_t)i * C);
  TTS_HIP(ctx, st->temb.reserve(te.size() * 4)); TTS_HIP(ctx, st->e1.reserve(te.size() * 4)); TTS_HIP(ctx, st->emb.reserve(te.size() * 4));
  TTS_HIP(ctx, st->ss_all.reserve((size_t)n * nres * 2 * C * 4));
  TTS_HIP(ctx, hipMemcpyAsync(st->temb.p, te.data(), te.size() * 4, hipMemcpyHostToDevice, ctx->stream));
  TTS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  // SiLU of both hidden vectors is applied by the kernel that produces them (emb is only used activated): three launches fewer than a separate in-place
  // activation kernel, and no address of these vectors ever holds two versions of itself within a call.
  const int act = ctx->ggml_lut ? 2 : 1;
  auto mlp = [&](float *ss_out) {
    linear_nk_kernel<<<C / LNK_COLS, 1024, 0, ctx->stream>>>(st->temb.as<float>(), C, n, st->te0_w, C, st->te0_b, st->e1.as<float>(), C, act);
    linear_nk_kernel<<<C / LNK_COLS, 1024, 0, ctx->stream>>>(st->e1.as<float>(), C, n, st->te2_w, C, st->te2_b, st->emb.as<float>(), C, act);
    for (int j = 0; j < nres; j++) {
      const ResDev &w = j < st->n_integ ? st->integ_res[j] : j < st->n_integ + st->n_main ? st->main_res[j - st->n_integ]
                                                                                          : st->tail_res[j - st->n_integ - st->n_main];
      // out row i -> ss[(i*nres + j)*2048]
      linear_nk_kernel<<<2 * C / LNK_COLS, 1024, 0, ctx->stream>>>(st->emb.as<float>(), C, n, w.emb_w, 2 * C, w.emb_b, ss_out + (size_t)j * 2 * C,
                                                                 nres * 2 * C, 0);
    }
  };
  // Evaluated TWICE and compared bit for bit; repeated until two evaluations agree. Round 4: while a second engine process used the same GPU, the previous form
  // of linear_nk_kernel returned a few wrong outputs in 10-30 % of the calls (traced to its packed f32 FMAs, see the kernel and
  // profiles/r4_two_process_determinism.txt; the present form has none and ran clean without this guard). The guard stays as a tripwire: the chain runs once per
  // utterance (~50 us), a silent 1e-3-level perturbation of the whole sampling loop becomes either the right values or an error, and
  // tts_diffusion_time_mlp_retries() says if it ever fired.
#ifdef TTS_DEBUG_NO_TIME_GUARD // developer build (tools/build_debug_lib.sh noguard): one evaluation, so that the probes see a fault itself
  mlp(st->ss_all.as<float>());
  if (false) {
#else
  {
#endif
  const size_t nss = (size_t)n * nres * 2 * C;
  TTS_HIP(ctx, st->ss_chk.reserve(nss * 4 + 4));
  int *flag = (int *)(st->ss_chk.as<float>() + nss);
  bool agreed = false;
  for (int attempt = 0; attempt < 64 && !agreed; attempt++) {
    mlp(st->ss_all.as<float>());
    mlp(st->ss_chk.as<float>());
    TTS_HIP(ctx, hipMemsetAsync(flag, 0, 4, ctx->stream));
    differ_kernel<<<64, 256, 0, ctx->stream>>>(st->ss_all.as<unsigned>(), st->ss_chk.as<unsigned>(), nss, flag);
    int h = 1;
    TTS_HIP(ctx, hipMemcpyAsync(&h, flag, 4, hipMemcpyDeviceToHost, ctx->stream));
    TTS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    agreed = h == 0;
    if (!agreed) ctx->time_mlp_retries++;
  }
  if (!agreed) return fail(ctx, TTS_ERR_HIP, "the timestep MLP did not evaluate to the same values twice in 64 attempts");
  }
  TTS_HIP(ctx, hipGetLastError());
  DBG_SUM("time temb", st->temb.p, (size_t)n * C * 4);
  DBG_SUM("time e1", st->e1.p, (size_t)n * C * 4);
  DBG_SUM("time emb", st->emb.p, (size_t)n * C * 4);
  DBG_SUM("time ss_all", st->ss_all.p, (size_t)n * nres * 2 * C * 4);
  return TTS_OK;
}

// Latent conditioner (main.cpp:3156-3319) for `lat_lens.size()` latents packed in st->lat_lay; result
// (before the upsample) in st->lat_wk.H().
static int latent_conditioner(tts_ctx *ctx, DiffState *st, const float *latents_host, const std::vector<int> &lat_lens) {
  Layout &ll = st->lat_lay;
  CHECK(ll.build(ctx, lat_lens));
  Work &wk = st->lat_wk;
  // Round 5: the conditioner's four AttentionBlocks ALWAYS run in reference precision (option lc_attn_f32, default 1). Their output — the code embedding — is
  // evaluated once per utterance and enters all 80 / 200 steps: an fp16 rounding inside it is the same perturbation at every step, exactly like the rounding
  // of the proj_out weight, and does not average out the way the per-step activation roundings of the integrator / main blocks do. Engine-side ablation
  // (profiles/r5_attention_ablation.txt): with an fp16 conditioner every single operand rounding of the per-step blocks looked 3-4x as expensive as the CPU
  // emulation (which takes the code embedding from the oracle) said. Cost: four blocks over L = 200 rows once per utterance.
  const bool lc_ref = ctx->lc_attn_f32 != 0;
  CHECK(wk.reserve(ctx, ll.rows, ll.ns, lc_ref));
  // latents -> f32 rows of X, then fp16 operand
  TTS_HIP(ctx, hipMemsetAsync(wk.X(), 0, (size_t)ll.rows * C * 4, ctx->stream));
  size_t off = 0;
  for (int s = 0; s < ll.ns; s++) {
    TTS_HIP(ctx, hipMemcpyAsync(wk.X() + (size_t)ll.start[s] * C, latents_host + off, (size_t)lat_lens[s] * C * 4,
                                hipMemcpyHostToDevice, ctx->stream));
    off += (size_t)lat_lens[s] * C;
  }
  to_f16_kernel<<<ll.rows, 256, 0, ctx->stream>>>(wk.X(), ll.d_row_seq.as<int>(), wk.A16());
  GemmArgs c3 = gemm_base(ll, wk.A16(), C, 3, C, st->lc_w, C, st->lc_bias);
  c3.mode = GEMM_OUT_F32; c3.outF = wk.X(); c3.ldo = C; c3.resid = nullptr;
  CHECK(gemm(ctx, "diff_gemm", c3, ll));
  for (int i = 0; i < st->n_lc; i++) CHECK(attention_block(ctx, st, ll, wk, wk.X(), st->lc_attn[i], lc_ref));
  CHECK(gn_stats(ctx, ll, wk, wk.X()));
  gn_apply_f32_kernel<<<ll.rows, 256, 0, ctx->stream>>>(wk.X(), ll.d_row_seq.as<int>(), wk.stats.as<float2>(), st->code_g, st->code_b,
                                                        st->cond_latent, wk.H());
  TTS_HIP(ctx, hipGetLastError());
  return TTS_OK;
}

// One network evaluation for every sequence of st->lay. Inputs: st->code_emb (f32 rows), st->xt16;
// ss = this timestep's scale/shift block [n_res][2048]. Output: st->net [rows][256].
static int network_forward(tts_ctx *ctx, DiffState *st, const float *ss) {
  Layout &lay = st->lay;
  Work &wk = st->wk;
  Layout &il = st->share_integ ? st->ilay : st->lay; // layout of the integrator stage
  Work &iw = st->share_integ ? st->iwk : st->wk;
  float *ce = st->ce.as<float>();
  if (st->n_integ == 0) TTS_HIP(ctx, hipMemcpyAsync(ce, st->code_emb.p, (size_t)il.rows * C * 4, hipMemcpyDeviceToDevice, ctx->stream));
  // option latency_mode: the statistics slots of this evaluation start empty (one memset node; every f32 GEMM of the step takes the next slot)
  st->gn_site = 0;
  iw.st_x = wk.st_x = iw.st_h = wk.st_h = nullptr;
  if (st->lat) TTS_HIP(ctx, hipMemsetAsync(st->gn_stats.p, 0, (size_t)st->gn_sites_max * st->gn_slot_ll * 8, ctx->stream));
  int j = 0;
  __half *ce16 = st->ce16.as<__half>(), *inp16 = st->inp16.as<__half>();
  if (st->hoisted) { // the integrator ran before the loop for every step (precompute_integrator): this step's slice
    j = st->n_integ;
    select_step_slice_kernel<<<256, 256, 0, ctx->stream>>>(st->ce16_all.as<uint4>(), (size_t)lay.rows * C / 8, st->step_ctr.as<int>(), (uint4 *)ce16);
  } else {
    for (int i = 0; i < st->n_integ; i++, j++) {
      // first block: reads the code embedding directly, its timestep-independent half (st->h0) comes from setup_batch
      if (i == 0) CHECK(res_block(ctx, st, il, iw, ce, st->integ_res[0], ss, st->code_emb.as<float>(), st->h0.as<float>(), st->lat ? st->gn_stats_h0.as<long long>() : nullptr));
      else CHECK(res_block(ctx, st, il, iw, ce, st->integ_res[i], ss + (size_t)j * 2 * C));
      CHECK(attention_block(ctx, st, il, iw, ce, st->integ_attn[i]));
    }
    if (st->share_integ) gather_f16_kernel<<<lay.rows, 256, 0, ctx->stream>>>(ce, st->ce_src.as<int>(), ce16);
    else to_f16_kernel<<<lay.rows, 256, 0, ctx->stream>>>(ce, lay.d_row_seq.as<int>(), ce16);
  }
  // inp_block: conv k3 100(->128) -> 1024 on x_t, output rounded to fp16 (operand of the next conv)
  GemmArgs gi = gemm_base(lay, st->xt16.as<__half>() + XTC, XTC, 3, XTC, st->inp_w, C, st->inp_bias);
  gi.mode = GEMM_OUT_F16; gi.outH = inp16; gi.ldh = C;
  DBG_SUM("ce16", ce16, (size_t)lay.rows * C * 2);
  CHECK(gemm(ctx, "diff_gemm", gi, lay, 0, 300));
  DBG_SUM("inp16", inp16, (size_t)lay.rows * C * 2);
  // integrating conv k1 over concat[inp | code_emb]
  GemmArgs gc = gemm_base(lay, inp16, C, 2, C, st->integ_w, C, st->integ_bias);
  gc.A[1] = ce16;
  wk.st_x = st->lat ? st->new_stats_slot() : nullptr;
  gc.mode = wk.st_x ? GEMM_OUT_F32_STATS : GEMM_OUT_F32; gc.outF = wk.X(); gc.ldo = C; gc.resid = nullptr;
  gc.st_out = wk.st_x; gc.st_stripe_ll = (int)st->gn_stripe_ll; gc.chunk_seq = lay.d_chunk_seq.as<int>();
  CHECK(gemm(ctx, "diff_gemm", gc, lay));
  DBG_SUM("integ conv", wk.X(), (size_t)lay.rows * C * 4);
  for (int i = 0; i < st->n_main; i++, j++) {
    CHECK(res_block(ctx, st, lay, wk, wk.X(), st->main_res[i], ss + (size_t)j * 2 * C));
    CHECK(attention_block(ctx, st, lay, wk, wk.X(), st->main_attn[i]));
  }
  for (int i = 0; i < st->n_tail; i++, j++) CHECK(res_block(ctx, st, lay, wk, wk.X(), st->tail_res[i], ss + (size_t)j * 2 * C));
  CHECK(gn(ctx, st, lay, wk.X(), st->lat ? wk.st_x : nullptr, st->outn_g, st->outn_b, nullptr, 1, wk.A16()));
  GemmArgs go = gemm_base(lay, wk.A16(), C, 3, C, st->out_w, 256, st->out_bias);
  go.mode = GEMM_OUT_F32; go.outF = st->net.as<float>(); go.ldo = 256; go.resid = nullptr;
  DBG_SUM("out gn", wk.A16(), (size_t)lay.rows * C * 2);
  CHECK(gemm(ctx, "diff_gemm", go, lay, 200, 0));
  DBG_SUM("net", st->net.p, (size_t)lay.rows * 256 * 4);
  if (st->lat && st->gn_site > st->gn_sites_max) return fail(ctx, TTS_ERR_STATE, "latency_mode: %d statistics slots used, %d reserved", st->gn_site, st->gn_sites_max);
  return TTS_OK;
}

// The conditioning_timestep_integrator layers for EVERY sampling step, before the loop (see DiffState::hoisted). Needs setup_batch (code embedding, h0) and
// precompute_time (ss_all) of this call. Timesteps are processed in chunks of about one benchmark batch of rows (28 672): the regime the GEMMs are tuned for, instead
// of 22 launches per step at 1 792 rows.
static int precompute_integrator(tts_ctx *ctx, DiffState *st, int n_steps) {
  const Layout &lay = st->lay;
  Layout &il = st->share_integ ? st->ilay : st->lay;
  const int per = il.ns, nt_max = std::max(1, 28672 / il.rows);
  const size_t ss_stride = (size_t)st->n_res() * 2 * C;
  TTS_HIP(ctx, st->ce16_all.reserve((size_t)n_steps * lay.rows * C * 2));
  // row of il that feeds row r of lay (-1: guard rows), and (sequence, t) of every il row
  std::vector<int> il_of_lay(lay.rows, -1), il_seq(il.rows, -1), il_t(il.rows, 0);
  for (int s = 0; s < il.ns; s++)
    for (int t = 0; t < il.len[s]; t++) { il_seq[il.start[s] + t] = s; il_t[il.start[s] + t] = t; }
  if (st->share_integ) il_of_lay = st->ce_src_host;
  else for (int s = 0; s < lay.ns; s++) for (int t = 0; t < lay.len[s]; t++) il_of_lay[lay.start[s] + t] = lay.start[s] + t;
  const bool lat_saved = st->lat;
  st->lat = false; // the batch path's GroupNorm kernels (statistics reduced per sequence: per-sequence scale / shift)
  int rc = TTS_OK;
  for (int idx0 = 0; idx0 < n_steps && rc == TTS_OK; idx0 += nt_max) {
    const int nt = std::min(nt_max, n_steps - idx0);
    std::vector<int> lens, step_of_seq;
    for (int k = 0; k < nt; k++)
      for (int s = 0; s < per; s++) { lens.push_back(il.len[s]); step_of_seq.push_back(idx0 + k); }
    Layout &pl = st->play;
    if ((rc = pl.build(ctx, lens))) break;
    if ((rc = pl.set_seq_step(ctx, step_of_seq, (int)ss_stride))) break;
    Work &pw = st->pwk;
    if ((rc = pw.reserve(ctx, pl.rows, pl.ns))) break;
    pw.st_x = pw.st_h = nullptr;
    std::vector<int> src(pl.rows, -1), scat((size_t)nt * lay.rows, -1);
    for (int k = 0; k < nt; k++)
      for (int s = 0; s < per; s++)
        for (int t = 0; t < il.len[s]; t++) src[pl.start[k * per + s] + t] = il.start[s] + t;
    for (int k = 0; k < nt; k++)
      for (int r = 0; r < lay.rows; r++) {
        const int ir = il_of_lay[r];
        if (ir >= 0 && il_seq[ir] >= 0) scat[(size_t)k * lay.rows + r] = pl.start[k * per + il_seq[ir]] + il_t[ir];
      }
    TTS_HIP(ctx, st->psrc.reserve(src.size() * 4)); TTS_HIP(ctx, st->pscatter.reserve(scat.size() * 4));
    TTS_HIP(ctx, hipMemcpy(st->psrc.p, src.data(), src.size() * 4, hipMemcpyHostToDevice));
    TTS_HIP(ctx, hipMemcpy(st->pscatter.p, scat.data(), scat.size() * 4, hipMemcpyHostToDevice));
    TTS_HIP(ctx, st->pcode.reserve((size_t)pl.rows * C * 4)); TTS_HIP(ctx, st->ph0.reserve((size_t)pl.rows * C * 4)); TTS_HIP(ctx, st->pce.reserve((size_t)pl.rows * C * 4));
    gather_f32_kernel<<<pl.rows, 256, 0, ctx->stream>>>(st->code_emb.as<float>(), st->psrc.as<int>(), st->pcode.as<float>());
    gather_f32_kernel<<<pl.rows, 256, 0, ctx->stream>>>(st->h0.as<float>(), st->psrc.as<int>(), st->ph0.as<float>());
    float *ce = st->pce.as<float>();
    const float *ssb = st->ss_all.as<float>(); // step 0's block; sequence s reads at + seq_step[s] * ss_stride
    for (int i = 0; i < st->n_integ && rc == TTS_OK; i++) {
      if (i == 0) rc = res_block(ctx, st, pl, pw, ce, st->integ_res[0], ssb, st->pcode.as<float>(), st->ph0.as<float>());
      else rc = res_block(ctx, st, pl, pw, ce, st->integ_res[i], ssb + (size_t)i * 2 * C);
      if (rc == TTS_OK) rc = attention_block(ctx, st, pl, pw, ce, st->integ_attn[i]);
    }
    if (rc) break;
    gather_f16_kernel<<<nt * lay.rows, 256, 0, ctx->stream>>>(ce, st->pscatter.as<int>(), st->ce16_all.as<__half>() + (size_t)idx0 * lay.rows * C);
    TTS_HIP(ctx, hipGetLastError());
    TTS_HIP(ctx, hipStreamSynchronize(ctx->stream)); // the next chunk rebuilds the layout's host-side tables
  }
  st->lat = lat_saved;
  return rc;
}

// Sets up layouts/buffers for B candidates (cond + optionally uncond copies) and the code embedding.
static int setup_batch(tts_ctx *ctx, DiffState *st, const float *latents, const std::vector<int> &L, bool cond, bool uncond) {
  const int B = (int)L.size();
  std::vector<int> lens, src;
  if (cond) for (int c = 0; c < B; c++) { lens.push_back(tts_diffusion_frames(L[c])); src.push_back(c); }
  if (uncond) for (int c = 0; c < B; c++) { lens.push_back(tts_diffusion_frames(L[c])); src.push_back(-1); }
  for (int t : lens) if (t < 1) return fail(ctx, TTS_ERR_ARG, "latent too short");
  CHECK(st->lay.build(ctx, lens));
  Layout &lay = st->lay;
  CHECK(st->wk.reserve(ctx, lay.rows, lay.ns));
  auto rz = [&](DevBuf &b, size_t bytes) -> hipError_t {
    size_t old = b.cap;
    hipError_t e = b.reserve(bytes);
    if (e == hipSuccess && b.cap != old) e = hipMemset(b.p, 0, b.cap);
    return e;
  };
  TTS_HIP(ctx, rz(st->code_emb, (size_t)lay.rows * C * 4));
  TTS_HIP(ctx, rz(st->ce, (size_t)lay.rows * C * 4));
  TTS_HIP(ctx, rz(st->ce16, (size_t)lay.rows * C * 2));
  TTS_HIP(ctx, rz(st->inp16, (size_t)lay.rows * C * 2));
  TTS_HIP(ctx, rz(st->xt16, (size_t)(lay.rows + 2) * XTC * 2));
  TTS_HIP(ctx, hipMemset(st->xt16.p, 0, st->xt16.cap));
  TTS_HIP(ctx, rz(st->net, (size_t)lay.rows * 256 * 4));
  TTS_HIP(ctx, st->seq_src.reserve(lay.ns * 4));
  TTS_HIP(ctx, hipMemcpy(st->seq_src.p, src.data(), lay.ns * 4, hipMemcpyHostToDevice));
  // integrator layout: every conditioned sequence, one unconditioned sequence per distinct length
  std::vector<int> ilens, isrc, seq_map(lay.ns);
  {
    std::vector<std::pair<int, int>> uniq; // (length, sequence of ilay)
    for (int s = 0; s < lay.ns; s++) {
      int found = -1;
      if (src[s] < 0)
        for (auto &u : uniq) if (u.first == lens[s]) found = u.second;
      if (found < 0) {
        found = (int)ilens.size();
        ilens.push_back(lens[s]); isrc.push_back(src[s]);
        if (src[s] < 0) uniq.push_back({lens[s], found});
      }
      seq_map[s] = found;
    }
  }
  st->share_integ = ctx->share_uncond && (int)ilens.size() < lay.ns; // option off: every unconditioned sequence is evaluated
  if (st->share_integ) {
    CHECK(st->ilay.build(ctx, ilens));
    CHECK(st->iwk.reserve(ctx, st->ilay.rows, st->ilay.ns));
    std::vector<int> cs(lay.rows, -1);
    for (int s = 0; s < lay.ns; s++)
      for (int t = 0; t < lens[s]; t++) cs[lay.start[s] + t] = st->ilay.start[seq_map[s]] + t;
    TTS_HIP(ctx, st->ce_src.reserve((size_t)lay.rows * 4));
    TTS_HIP(ctx, hipMemcpy(st->ce_src.p, cs.data(), (size_t)lay.rows * 4, hipMemcpyHostToDevice));
    st->ce_src_host = cs;
    TTS_HIP(ctx, st->iseq_src.reserve(st->ilay.ns * 4));
    TTS_HIP(ctx, hipMemcpy(st->iseq_src.p, isrc.data(), st->ilay.ns * 4, hipMemcpyHostToDevice));
  }
  Layout &il = st->share_integ ? st->ilay : st->lay;
  if (cond) CHECK(latent_conditioner(ctx, st, latents, L));
  else { // layout still needed by build_code_emb (never dereferenced for uncond rows)
    CHECK(st->lat_lay.build(ctx, L));
    CHECK(st->lat_wk.reserve(ctx, st->lat_lay.rows, st->lat_lay.ns, ctx->lc_attn_f32 != 0));
  }
  build_code_emb_kernel<<<il.rows, 256, 0, ctx->stream>>>(st->lat_wk.H(), st->lat_lay.d_start.as<int>(), st->lat_lay.d_len.as<int>(),
                                                          st->uncond_emb, il.d_row_seq.as<int>(), il.d_row_t.as<int>(), il.d_len.as<int>(),
                                                          (st->share_integ ? st->iseq_src : st->seq_src).as<int>(), st->code_emb.as<float>());
  TTS_HIP(ctx, hipGetLastError());
  // Option latency_mode, small layouts only (one or two utterances: the GroupNorm kernels are latency-bound there, 64 workgroups each): the f32 GEMMs of the sampling
  // step leave the GroupNorm statistics of their outputs in per-step slots and the GroupNorms become gn_apply_kernel. Not bit-identical to the batch path (variance from
  // exact sums instead of the two-pass f32 form): opt-in, the default keeps a candidate's result independent of its batch.
  st->lat = ctx->latency_mode != 0 && lay.rows <= LAT_MAX_ROWS && il.rows <= LAT_MAX_ROWS;
  if (st->lat) {
    st->gn_stripe_ll = (size_t)std::max(lay.ns, il.ns) * 32 * 4;
    st->gn_slot_ll = st->gn_stripe_ll * FX_STRIPES;
    st->gn_sites_max = 3 * (st->n_integ + st->n_main + st->n_tail) + 2;
    TTS_HIP(ctx, st->gn_stats.reserve((size_t)st->gn_sites_max * st->gn_slot_ll * 8));
    TTS_HIP(ctx, st->gn_stats_h0.reserve(st->gn_slot_ll * 8));
    TTS_HIP(ctx, hipMemsetAsync(st->gn_stats_h0.p, 0, st->gn_slot_ll * 8, ctx->stream));
  }
  if (st->n_integ > 0) {
    TTS_HIP(ctx, rz(st->h0, (size_t)il.rows * C * 4));
    // (the code embedding's own statistics are reduced by the GroupNorm kernel: once per utterance)
    CHECK(res_in_layers(ctx, st, il, st->share_integ ? st->iwk : st->wk, st->code_emb.as<float>(), st->integ_res[0], st->h0.as<float>(), nullptr,
                        st->lat ? st->gn_stats_h0.as<long long>() : nullptr));
  }
  return TTS_OK;
}

// tts_diffusion_forward: one evaluation of one branch (parity-test entry point).
int diff_forward(tts_ctx *ctx, const float *latents, int L, const float *x_t, int timestep, int cond_free, float *out) {
  DiffState *st = ctx->diff;
  if (!st) return fail(ctx, TTS_ERR_STATE, "diffusion model not loaded");
  if (!latents || !x_t || !out || L < 1) return fail(ctx, TTS_ERR_ARG, "tts_diffusion_forward: bad argument");
  std::vector<int> Ls{L};
  CHECK(setup_batch(ctx, st, latents, Ls, !cond_free, cond_free));
  st->hoisted = false;
  const int T = st->lay.len[0];
  CHECK(precompute_time(ctx, st, std::vector<int>{timestep}));
  TTS_HIP(ctx, st->xbuf.reserve((size_t)100 * T * 4));
  TTS_HIP(ctx, hipMemcpyAsync(st->xbuf.p, x_t, (size_t)100 * T * 4, hipMemcpyHostToDevice, ctx->stream));
  int64_t off0 = 0;
  TTS_HIP(ctx, st->xoff.reserve(8));
  TTS_HIP(ctx, hipMemcpyAsync(st->xoff.p, &off0, 8, hipMemcpyHostToDevice, ctx->stream));
  Layout &lay = st->lay;
  xt_to_rows_kernel<<<lay.rows, 128, 0, ctx->stream>>>(st->xbuf.as<float>(), st->xoff.as<int64_t>(), lay.d_row_seq.as<int>(),
                                                       lay.d_row_t.as<int>(), lay.d_len.as<int>(), lay.d_start.as<int>(), 1, 0,
                                                       st->xt16.as<__half>() + XTC);
  CHECK(network_forward(ctx, st, st->ss_all.as<float>()));
  TTS_HIP(ctx, st->out_ct.reserve((size_t)200 * T * 4));
  rows_to_ct_kernel<<<(200 * T + 255) / 256, 256, 0, ctx->stream>>>(st->net.as<float>(), lay.start[0], T, 200, 256, st->out_ct.as<float>());
  TTS_HIP(ctx, hipMemcpyAsync(out, st->out_ct.p, (size_t)200 * T * 4, hipMemcpyDeviceToHost, ctx->stream));
  TTS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return TTS_OK;
}

// tts_diffusion: the sampling loop for B candidates.
int diff_sample(tts_ctx *ctx, const float *latents, const int32_t *rows, int B, int n_steps, const float *noise, int noise_mode,
                float *mel_out) {
  DiffState *st = ctx->diff;
  if (!st) return fail(ctx, TTS_ERR_STATE, "diffusion model not loaded");
  if (!latents || !rows || !mel_out || B < 1 || n_steps < 2) return fail(ctx, TTS_ERR_ARG, "tts_diffusion: bad argument");
  std::vector<int> L(rows, rows + B);
  for (int l : L) if (l < 1 || l > 500) return fail(ctx, TTS_ERR_ARG, "latent rows %d out of range", l);
  static const bool timing = getenv("TTS_TIMING") != nullptr; // host-side breakdown on stderr
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
    return std::chrono::duration<double, std::milli>(b - a).count();
  };
  const auto t_begin = now();
  CHECK(setup_batch(ctx, st, latents, L, true, true));
  if (timing) (void)hipStreamSynchronize(ctx->stream);
  const auto t_setup = now();
  Layout &lay = st->lay;
  DiffSchedule sched;
  sched.build(n_steps);
  std::vector<int> ts(n_steps);
  for (int idx = 0; idx < n_steps; idx++) ts[idx] = sched.timestep_map[n_steps - 1 - idx]; // time_embedding_{idx} (5819-5825)
  CHECK(precompute_time(ctx, st, ts));
  // small batches: the integrator layers of all steps now, in benchmark-sized batches (option hoist_integrator, default 1; results are bit-identical either way)
  st->hoisted = ctx->hoist_integrator != 0 && st->n_integ > 0 && lay.rows <= (ctx->hoist_integrator > 1 ? ctx->hoist_integrator : HOIST_MAX_ROWS);
  if (st->hoisted) CHECK(precompute_integrator(ctx, st, n_steps));
  // x state [cand][100][T_c]
  std::vector<int64_t> xoff(B);
  int64_t total = 0;
  for (int c = 0; c < B; c++) { xoff[c] = total; total += (int64_t)100 * lay.len[c]; }
  TTS_HIP(ctx, st->xbuf.reserve(total * 4));
  TTS_HIP(ctx, st->xoff.reserve(B * 8));
  TTS_HIP(ctx, hipMemcpy(st->xoff.p, xoff.data(), B * 8, hipMemcpyHostToDevice));
  const bool host_noise = noise != nullptr || noise_mode == TTS_NOISE_REFERENCE;
  // One candidate in the reference's draw order (what ./tortoise runs): 81 x 100 T normal draws from the libstdc++ objects take longer on the host (~3 ms per step at
  // T = 870) than the device takes for the step. Round 6: block k + 1 is drawn and sent while the device still works on steps <= k - 1 — the same draws in the same
  // order (x_T, then step after step), the stream orders copy k + 1 in front of step k's update kernel. More candidates draw candidate after candidate
  // (main.cpp:5638, 6020-6021): their order does not allow it.
  bool pipe_noise = host_noise && !noise && B == 1 && ctx->noise_pipeline != 0;
  if (pipe_noise && st->noise_host.reserve((size_t)total * (n_steps + 1) * 4) != hipSuccess) { (void)hipGetLastError(); pipe_noise = false; }
  auto draw_block = [&](int k) { // block k of the one candidate into the pinned buffer, then on its way to the device
    float *dst = st->noise_host.as<float>() + (size_t)k * total;
    rng_normal_fill(ctx, dst, total);
    return hipMemcpyAsync(st->noise.as<float>() + (size_t)k * total, dst, (size_t)total * 4, hipMemcpyHostToDevice, ctx->stream);
  };
  std::vector<float> hn;
  if (pipe_noise) {
    TTS_HIP(ctx, st->noise.reserve((size_t)total * (n_steps + 1) * 4));
    TTS_HIP(ctx, draw_block(0));
    TTS_HIP(ctx, hipMemcpyAsync(st->xbuf.p, st->noise.p, total * 4, hipMemcpyDeviceToDevice, ctx->stream));
  } else if (host_noise) {
    // per step a [cand][100][T] block in the layout of x: block 0 = x_T, block 1+idx = step idx
    hn.resize((size_t)total * (n_steps + 1));
    if (noise) { // caller layout: per candidate (n_steps+1) consecutive vectors
      size_t src = 0;
      for (int c = 0; c < B; c++)
        for (int k = 0; k <= n_steps; k++) {
          memcpy(hn.data() + (size_t)k * total + xoff[c], noise + src, (size_t)100 * lay.len[c] * 4);
          src += (size_t)100 * lay.len[c];
        }
    } else { // the reference's draw order, candidate after candidate (main.cpp:5638, 6020-6021)
      for (int c = 0; c < B; c++)
        for (int k = 0; k <= n_steps; k++) {
          float *dst = hn.data() + (size_t)k * total + xoff[c];
          rng_normal_fill(ctx, dst, (int64_t)100 * lay.len[c]);
        }
    }
    TTS_HIP(ctx, st->noise.reserve(hn.size() * 4));
    TTS_HIP(ctx, hipMemcpy(st->noise.p, hn.data(), hn.size() * 4, hipMemcpyHostToDevice));
    TTS_HIP(ctx, hipMemcpyAsync(st->xbuf.p, st->noise.p, total * 4, hipMemcpyDeviceToDevice, ctx->stream));
  } else {
    for (int c = 0; c < B; c++) {
      int64_t n = (int64_t)100 * lay.len[c];
      philox_fill_kernel<<<(int)((n + 255) / 256), 256, 0, ctx->stream>>>(st->xbuf.as<float>() + xoff[c], n, ctx->seed_value, (uint32_t)(shard_base(ctx) + c), 0xFFFFFFFFu);
    }
  }
  if (timing) (void)hipStreamSynchronize(ctx->stream);
  const auto t_pre = now();
  const size_t ss_stride = (size_t)st->n_res() * 2 * C;
  // per-step table + device step counter (see StepEntry)
  std::vector<StepEntry> tab(n_steps);
  for (int idx = 0; idx < n_steps; idx++) {
    const int t = n_steps - 1 - idx;
    tab[idx].sc = StepScalars{sched.max_log[t], sched.min_log[t], sched.cfk[t], sched.sqrt_recip[t], sched.sqrt_recipm1[t],
                              sched.coef1[t], sched.coef2[t], t == 0 ? 1 : 0};
    tab[idx].has_noise = host_noise ? 1 : 0;
    tab[idx].noise_off = (long long)(idx + 1) * total;
    tab[idx].philox_step = (unsigned)idx;
    tab[idx].pad = 0;
  }
  TTS_HIP(ctx, st->step_tab.reserve(tab.size() * sizeof(StepEntry)));
  TTS_HIP(ctx, st->step_ctr.reserve(64));
  TTS_HIP(ctx, st->ss_cur.reserve(ss_stride * 4));
  TTS_HIP(ctx, hipMemcpyAsync(st->step_tab.p, tab.data(), tab.size() * sizeof(StepEntry), hipMemcpyHostToDevice, ctx->stream));
  TTS_HIP(ctx, hipMemsetAsync(st->step_ctr.p, 0, 64, ctx->stream));
  TTS_HIP(ctx, hipStreamSynchronize(ctx->stream)); // `tab` is a host vector: the copy must have read it before it goes out of scope
  // One sampling step, identical for every step (all per-step values are read through the device counter): launched eagerly or
  // captured once and replayed.
  auto enqueue_step = [&]() -> int {
    step_begin_kernel<<<16, 256, 0, ctx->stream>>>(st->ss_all.as<float>(), ss_stride, st->step_ctr.as<int>(), st->ss_cur.as<float>());
    xt_to_rows_kernel<<<lay.rows, 128, 0, ctx->stream>>>(st->xbuf.as<float>(), st->xoff.as<int64_t>(), lay.d_row_seq.as<int>(),
                                                         lay.d_row_t.as<int>(), lay.d_len.as<int>(), lay.d_start.as<int>(), B, 1,
                                                         st->xt16.as<__half>() + XTC);
    CHECK(network_forward(ctx, st, st->ss_cur.as<float>()));
    {
      ProfScope ps(ctx, "diff_update");
      ddpm_update_kernel<<<lay.rows, 128, 0, ctx->stream>>>(
          st->net.as<float>(), st->xbuf.as<float>(), st->xoff.as<int64_t>(), lay.d_row_seq.as<int>(), lay.d_row_t.as<int>(),
          lay.d_len.as<int>(), lay.d_start.as<int>(), B, st->step_tab.as<StepEntry>(), st->step_ctr.as<int>(),
          host_noise ? st->noise.as<float>() : nullptr, ctx->seed_value, (uint32_t)shard_base(ctx));
    }
    step_advance_kernel<<<1, 1, 0, ctx->stream>>>(st->step_ctr.as<int>());
    TTS_HIP(ctx, hipGetLastError());
    return TTS_OK;
  };
  // Event records cannot ride in the replayed graph: while a diff_* family is profiled every prof_eager_every-th step is launched
  // eagerly (with its event pairs), the others replay the graph; "diff_graph" = 0 (or TTS_NO_GRAPH, e.g. under rocprofv3) launches
  // every step eagerly.
  static const bool no_graph_env = getenv("TTS_NO_GRAPH") != nullptr;
  bool prof_diff = ctx->prof_on && ctx->prof_filter.empty();
  for (const std::string &f : ctx->prof_filter) prof_diff |= ctx->prof_on && f.rfind("diff_", 0) == 0;
  const bool use_graph = ctx->diff_graph && !no_graph_env && n_steps > 2;
  double t_capture = 0;
  if (use_graph) {
    const auto tc0 = now();
    st->drop_step_graph(); // layouts and buffers belong to this call
    ctx->capturing = true;
    hipError_t eb = hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal);
    int rc = eb == hipSuccess ? enqueue_step() : TTS_OK;
    hipError_t ee = eb == hipSuccess ? hipStreamEndCapture(ctx->stream, &st->step_graph) : eb;
    ctx->capturing = false;
    if (rc) return rc;
    TTS_HIP(ctx, ee);
    TTS_HIP(ctx, hipGraphInstantiate(&st->step_exec, st->step_graph, nullptr, nullptr, 0));
    t_capture = ms(tc0, now());
  }
  for (int idx = 0; idx < n_steps; idx++) {
    const bool eager = !use_graph || (prof_diff && idx % ctx->prof_eager_every == 0);
    if (pipe_noise) TTS_HIP(ctx, draw_block(idx + 1)); // read by this step's update kernel
    if (eager) CHECK(enqueue_step());
    else TTS_HIP(ctx, hipGraphLaunch(st->step_exec, ctx->stream));
  }
  const auto t_issued = now();
  if (timing) (void)hipStreamSynchronize(ctx->stream);
  const auto t_loop = now();
  TTS_HIP(ctx, hipMemcpyAsync(mel_out, st->xbuf.p, total * 4, hipMemcpyDeviceToHost, ctx->stream));
  TTS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (timing)
    fprintf(stderr, "[tts timing] diffusion: setup %.1f ms, time MLP + noise %.1f, %d steps %.1f (issued in %.1f, graph capture + instantiate %.1f), mel copy %.1f\n",
            ms(t_begin, t_setup), ms(t_setup, t_pre), n_steps, ms(t_pre, t_loop), ms(t_pre, t_issued), t_capture, ms(t_loop, now()));
  return TTS_OK;
}

} // namespace tts

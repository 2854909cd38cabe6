// Autoregressive stage (GPT-2, 30 x 1024, 16 heads x 64) on gfx950.
//
// Replaces: autoregressive_model_load (main.cpp:482-897), autoregressive_graph (2545-3040),
// autoregressive_latent_graph (2053-2519) and the autoregressive() driver (5042-5367).
//
// Numerics follow the reference graph: F32 weights and F32 accumulation for every weight matmul,
// QKV activations rounded to fp16 (main.cpp:2789-2790) — so the KV cache is *stored* as fp16
// without changing a bit — F32 softmax, tanh-GELU, LayerNorm eps 1e-5.
//
// Decode is HBM-bound weight streaming: all weights are read exactly once per step for
// up to 16 candidates (split-K GEMV, deterministic two-level reduction, no atomics, so token ids
// are run-to-run reproducible). Layouts are fixed once at load time (the reference re-transposes
// every weight matrix inside every graph execution, main.cpp:2769-2777).
#include "common.h"
#include "gemm_f16.h"
#include <hip/hip_fp16.h>
#include <algorithm>
#include <cmath>
#include <cstdlib>

namespace tts {

static constexpr int D = 1024, NH = 16, HD = 64, FF = 4096, V = TTS_VOCAB_MEL, VPAD = 8256;

// LayerNorm-GEMV decode kernels: split-precision fp16 MFMA (default: three products per K step, 2^-22 relative) or exact-f32 MFMA
// products (option "dec_f32_mfma" = 1, read by tts_load_ar: it selects the weight packing; ArState::f32_mfma).
// Weight slabs of the decode step are streamed once per step by exactly one workgroup: non-temporal loads keep them from displacing the
// activations / KV rows in L2 (MI355X_MICROARCH.md, row "nt-weights"; measured 1 008 vs 1 036 us per step, profiles/r3_bench_n1.json).
typedef float ntfloat4 __attribute__((ext_vector_type(4)));
template <bool NT> __device__ __forceinline__ float4 ldw4(const float4 *p) {
  if (!NT) return *p;
  const ntfloat4 v = __builtin_nontemporal_load((const ntfloat4 *)p);
  return make_float4(v[0], v[1], v[2], v[3]);
}

// ---------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float f16_round(float v) { return __half2float(__float2half_rn(v)); }

__device__ __forceinline__ float gelu_tanh(float x, int lut) {
  const float A = 0.044715f, S = 0.79788456080286535587989211986876f;
  if (lut) {
    float xr = f16_round(x);
    return f16_round(0.5f * xr * (1.0f + tanhf(S * xr * (1.0f + A * xr * xr))));
  }
  // tanh u = 1 - 2 / (1 + e^{2u}) on the hardware exp2/rcp (absolute error ~1e-7; libm's tanhf is ~40 instructions and
  // sits in the serial tail of the c_fc decode kernel)
  const float u = S * x * (1.0f + A * x * x);
  const float th = 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(u * 2.88539008177793f));
  return 0.5f * x * (1.0f + th);
}

// rows of the transformer input: out[r] = tabA[ia[r]] + tabB[ib[r]]  (ib < 0: no second term).
// tables: 0 voice(1 row), 1 text_emb, 2 mel_emb ; 0 text_pos, 1 mel_pos.
struct EmbedTables { const float *a[3]; const float *b[2]; };
__global__ __launch_bounds__(256) void embed_rows_kernel(EmbedTables t, const int4 *__restrict__ desc,
                                                         float *__restrict__ out) {
  const int r = blockIdx.x;
  const int4 d = desc[r]; // {tableA, idxA, tableB (-1 none), idxB}
  const float4 *pa = (const float4 *)(t.a[d.x] + (size_t)d.y * D);
  float4 v = pa[threadIdx.x];
  if (d.z >= 0) {
    const float4 *pb = (const float4 *)(t.b[d.z] + (size_t)d.w * D);
    float4 w = pb[threadIdx.x];
    v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
  }
  ((float4 *)(out + (size_t)r * D))[threadIdx.x] = v;
}

__device__ __forceinline__ float block_sum_256(float v, float *sh) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[w] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}

// LayerNorm over 1024 (ggml_norm eps 1e-5, then *g + b). One block (256 thr x float4) per row.
__global__ __launch_bounds__(256) void layernorm_kernel(const float *__restrict__ x, const float *__restrict__ g,
                                                        const float *__restrict__ b, float *__restrict__ y) {
  __shared__ float sh[4];
  const size_t row = blockIdx.x;
  float4 v = ((const float4 *)(x + row * D))[threadIdx.x];
  float mean = block_sum_256(v.x + v.y + v.z + v.w, sh) * (1.0f / D);
  v.x -= mean; v.y -= mean; v.z -= mean; v.w -= mean;
  float var = block_sum_256(v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w, sh) * (1.0f / D);
  float sc = 1.0f / sqrtf(var + 1e-5f);
  float4 gg = ((const float4 *)g)[threadIdx.x], bb = ((const float4 *)b)[threadIdx.x];
  v.x = v.x * sc * gg.x + bb.x; v.y = v.y * sc * gg.y + bb.y;
  v.z = v.z * sc * gg.z + bb.z; v.w = v.w * sc * gg.w + bb.w;
  ((float4 *)(y + row * D))[threadIdx.x] = v;
}

// Split-K GEMV/skinny GEMM: part[ks][rows][N] = X[rows][k-chunk] * W[k-chunk][N].
// W is [K][N] row-major (N contiguous, N % 64 == 0). Block = 64 columns x one K chunk x RT rows;
// thread (cx = tid&15, ky = tid>>4) owns 4 columns and every 16th k of the chunk, so a wave's load
// instruction covers 4 consecutive W rows x 256 B. Reduction over ky: shuffles inside a wave, LDS
// across the 4 waves; the K chunks are summed in order by the epilogue kernel (deterministic).
// PRO = 1: the input is itself a split-K partial buffer and x[r][k] = gelu(sum_s pin[s][r][k] + bin[k]) is
// formed while staging (fuses the c_fc epilogue into the c_proj GEMV); X then points at pin, ldx = its N.
template <int RT, int PRO>
__global__ __launch_bounds__(256) void gemv_kn_kernel(const float *__restrict__ X, int ldx, int rows,
                                                      const float *__restrict__ W, int N, int K, int kspan,
                                                      float *__restrict__ part, int pin_ks, const float *__restrict__ pin_bias,
                                                      int lut) {
  __shared__ float xs[256 * RT];
  __shared__ float red[4 * RT * 64];
  const int tid = threadIdx.x, cx = tid & 15, ky = tid >> 4;
  const int r0 = blockIdx.z * RT;
  float acc[RT][4];
#pragma unroll
  for (int r = 0; r < RT; r++) { acc[r][0] = acc[r][1] = acc[r][2] = acc[r][3] = 0.f; }
  // the block's K range [blockIdx.y*kspan, +kspan) is staged through LDS in chunks of <= 256
  for (int k0 = blockIdx.y * kspan, kend = k0 + kspan; k0 < kend; k0 += 256) {
    const int kchunk = min(256, kend - k0);
    __syncthreads();
    for (int idx = tid; idx < kchunk * RT; idx += 256) {
      int r = idx / kchunk, k = idx - r * kchunk;
      float xv = 0.f;
      if (r0 + r < rows) {
        xv = X[(size_t)(r0 + r) * ldx + k0 + k];
        if (PRO) {
          for (int sp = 1; sp < pin_ks; sp++) xv += X[((size_t)sp * rows + r0 + r) * ldx + k0 + k];
          xv = gelu_tanh(xv + pin_bias[k0 + k], lut);
        }
      }
      xs[k * RT + r] = xv;
    }
    __syncthreads();
    const float *wp = W + ((size_t)blockIdx.x * K + k0 + ky) * 64 + cx * 4; // strip-major [N/64][K][64]
#pragma unroll 4
    for (int k = ky; k < kchunk; k += 16) {
      const float4 w = *(const float4 *)wp;
      wp += 16 * 64;
#pragma unroll
      for (int r = 0; r < RT; r++) {
        const float xv = xs[k * RT + r];
        acc[r][0] = fmaf(xv, w.x, acc[r][0]); acc[r][1] = fmaf(xv, w.y, acc[r][1]);
        acc[r][2] = fmaf(xv, w.z, acc[r][2]); acc[r][3] = fmaf(xv, w.w, acc[r][3]);
      }
    }
  }
  const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int r = 0; r < RT; r++)
#pragma unroll
    for (int j = 0; j < 4; j++) {
      float v = acc[r][j];
      v += __shfl_xor(v, 16);
      v += __shfl_xor(v, 32);
      if (lane < 16) red[(wave * RT + r) * 64 + cx * 4 + j] = v;
    }
  __syncthreads();
  for (int idx = tid; idx < RT * 16; idx += 256) {
    int r = idx >> 4, c4 = idx & 15;
    if (r0 + r < rows) {
      float4 s;
      const float *p0 = &red[(0 * RT + r) * 64 + c4 * 4], *p1 = &red[(1 * RT + r) * 64 + c4 * 4];
      const float *p2 = &red[(2 * RT + r) * 64 + c4 * 4], *p3 = &red[(3 * RT + r) * 64 + c4 * 4];
      s.x = ((p0[0] + p1[0]) + p2[0]) + p3[0]; s.y = ((p0[1] + p1[1]) + p2[1]) + p3[1];
      s.z = ((p0[2] + p1[2]) + p2[2]) + p3[2]; s.w = ((p0[3] + p1[3]) + p2[3]) + p3[3];
      *(float4 *)&part[((size_t)blockIdx.y * rows + r0 + r) * N + blockIdx.x * 64 + c4 * 4] = s;
    }
  }
}


// ---------------------------------------------------------------------------------------------
// Split-precision MFMA path for the multi-row passes (prefill, latent pass): x*W in f32 is evaluated as
// x_hi*W_hi + x_lo*W_hi + x_hi*W_lo with fp16 hi/lo parts (hi = fp16(v), lo = fp16(v - hi): 22 significant
// bits) on the fp16 MFMA GEMM with f32 accumulation — ~16x the f32 FMA rate at ~2^-21 relative accuracy
// (the oracle gate for these passes is 1e-4). Weights are pre-scaled by 64 so the lo parts stay normal.
// ---------------------------------------------------------------------------------------------
static constexpr float W16_SCALE = 64.0f;
// W f32 strip-major [N/64][K][64] -> out fp16 [N][2K] = [hi(0..K-1) | lo(0..K-1)] of 64*W^T (tile transpose through LDS)
__global__ __launch_bounds__(256) void split_weight_kernel(const float *__restrict__ W, int K, int N, __half *__restrict__ out) {
  __shared__ float t[32][33];
  const int k0 = blockIdx.y * 32, n0 = blockIdx.x * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int r = ty; r < 32; r += 8) t[r][tx] = W[((size_t)(n0 >> 6) * K + k0 + r) * 64 + (n0 & 63) + tx] * W16_SCALE; // strip-major source
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const float v = t[tx][r];
    const __half hi = __float2half_rn(v);
    const __half lo = __float2half_rn(v - __half2float(hi));
    out[(size_t)(n0 + r) * 2 * K + k0 + tx] = hi;
    out[(size_t)(n0 + r) * 2 * K + K + k0 + tx] = lo;
  }
}

// ---- load-time re-layouts on the device (round 6) -------------------------------------------------
// The host packers above (strip_major, fold_layernorm, pack_mfma16h, pack_cols4) are index permutations plus one rounding each; tts_load_ar spent 5.9 s in them on one
// core and still 0.5 s on sixteen. The same permutations as kernels: a worker uploads a tensor as it lies in the file and the layouts are produced from that copy.
// Same arithmetic per element (f32 multiply by the LayerNorm gain, exact scaling by 64, round-to-nearest-even fp16 hi / lo, the double-precision column sums of the
// folded bias in ascending k): the buffers equal the host packers' byte for byte (tests/test_ar_gpu.py compares the logits of the two loads bit for bit).
__global__ __launch_bounds__(256) void pk_strip_major_kernel(const float *__restrict__ w, int K, int N, float *__restrict__ t) {
  const size_t total = (size_t)K * N;
  for (size_t o = (size_t)blockIdx.x * 256 + threadIdx.x; o < total; o += (size_t)gridDim.x * 256) {
    const size_t s0 = o / ((size_t)K * 64), rem = o % ((size_t)K * 64);
    t[o] = w[(rem / 64) * N + s0 * 64 + (rem % 64)];
  }
}
__global__ __launch_bounds__(256) void pk_fold_kernel(const float *__restrict__ w, int K, int N, const float *__restrict__ g, float *__restrict__ wf) {
  const size_t total = (size_t)K * N;
  for (size_t o = (size_t)blockIdx.x * 256 + threadIdx.x; o < total; o += (size_t)gridDim.x * 256) wf[o] = g[o / N] * w[o];
}
// cf[n] = float(double(c[n]) + sum_k double(b[k]) double(w[k][n])), k ascending (a product of two floats is exact in double: fused or not, the same sum)
__global__ __launch_bounds__(256) void pk_fold_bias_kernel(const float *__restrict__ w, int K, int N, const float *__restrict__ b, const float *__restrict__ c,
                                                           float *__restrict__ cf) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  double acc = 0.0;
  for (int k0 = 0; k0 < K; k0 += 16) { // 16 independent loads in flight, then the 16 additions in ascending k (a first version waited for every load: 280 us per launch)
    float wv[16];
#pragma unroll
    for (int j = 0; j < 16; j++) wv[j] = k0 + j < K ? w[(size_t)(k0 + j) * N + n] : 0.f;
#pragma unroll
    for (int j = 0; j < 16; j++)
      if (k0 + j < K) acc += (double)b[k0 + j] * (double)wv[j];
  }
  cf[n] = (float)((double)c[n] + acc);
}
// pack_mfma16h: one thread per weight (K = 1024)
__global__ __launch_bounds__(256) void pk_mfma16h_kernel(const float *__restrict__ w, int N, __half *__restrict__ t) {
  const size_t total = (size_t)1024 * N;
  for (size_t o = (size_t)blockIdx.x * 256 + threadIdx.x; o < total; o += (size_t)gridDim.x * 256) {
    const int e = (int)(o & 7), lane = (int)((o >> 3) & 63), s2 = (int)((o >> 9) & 7), wv = (int)((o >> 12) & 3);
    const size_t cb = o >> 14;
    const float v = W16_SCALE * w[(size_t)(wv * 256 + s2 * 32 + 8 * (lane >> 4) + e) * N + cb * 16 + (lane & 15)];
    const __half hi = __float2half_rn(v);
    const size_t base = ((((cb * 4 + wv) * 8 + s2) * 64 + lane) * 16);
    t[base + e] = hi;
    t[base + 8 + e] = __float2half_rn(v - __half2float(hi));
  }
}
__global__ __launch_bounds__(256) void pk_cols4_kernel(const float *__restrict__ w, int K, int N, float *__restrict__ t) {
  const size_t total = (size_t)K * N;
  const int KG = K / 1024;
  for (size_t o = (size_t)blockIdx.x * 256 + threadIdx.x; o < total; o += (size_t)gridDim.x * 256) {
    const int c = (int)(o & 3), tid = (int)((o >> 2) & 255), kk = (int)((o >> 10) & 3);
    const size_t q = o >> 12; // cb * KG + i
    const size_t cb = q / KG, i = q % KG;
    t[o] = w[(i * 1024 + 4 * tid + kk) * N + cb * 4 + c];
  }
}
// nn.Linear [V][D] -> [D][VPAD], zero padded
__global__ __launch_bounds__(256) void pk_transpose_pad_kernel(const float *__restrict__ w, int rowsV, int colsD, int vpad, float *__restrict__ wt) {
  const size_t total = (size_t)colsD * vpad;
  for (size_t o = (size_t)blockIdx.x * 256 + threadIdx.x; o < total; o += (size_t)gridDim.x * 256) {
    const size_t k = o / vpad, n = o % vpad;
    wt[o] = n < (size_t)rowsV ? w[n * colsD + k] : 0.f;
  }
}
// max |w| into *out (bits of a non-negative float order like unsigned integers; a NaN ends up above every number and fails the range check loudly)
__global__ __launch_bounds__(256) void pk_absmax_kernel(const float *__restrict__ w, size_t n, unsigned *__restrict__ out) {
  __shared__ unsigned wm[4];
  unsigned m = 0;
  for (size_t o = (size_t)blockIdx.x * 256 + threadIdx.x; o < n; o += (size_t)gridDim.x * 256) m = max(m, __float_as_uint(fabsf(w[o])));
  for (int off = 32; off; off >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, off));
  if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) { // one atomic per workgroup (32 K waves hitting one address took 340 us per launch)
    m = max(max(wm[0], wm[1]), max(wm[2], wm[3]));
    if (m) atomicMax(out, m);
  }
}
// x f32 [rows][K] -> hi, lo fp16 [rows_pad][K] (pad rows zero)
__global__ __launch_bounds__(256) void split_act_kernel(const float *__restrict__ x, int rows, int K, __half *__restrict__ hi,
                                                        __half *__restrict__ lo) {
  const int r = blockIdx.x;
  for (int c = threadIdx.x * 4; c < K; c += 1024) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < rows) v = *(const float4 *)(x + (size_t)r * K + c);
    const __half2 h0 = __floats2half2_rn(v.x, v.y), h1 = __floats2half2_rn(v.z, v.w);
    const float2 f0 = __half22float2(h0), f1 = __half22float2(h1);
    const __half2 l0 = __floats2half2_rn(v.x - f0.x, v.y - f0.y), l1 = __floats2half2_rn(v.z - f1.x, v.w - f1.y);
    uint2 uh, ul;
    uh.x = *(const unsigned *)&h0; uh.y = *(const unsigned *)&h1;
    ul.x = *(const unsigned *)&l0; ul.y = *(const unsigned *)&l1;
    *(uint2 *)(hi + (size_t)r * K + c) = uh;
    *(uint2 *)(lo + (size_t)r * K + c) = ul;
  }
}

enum { EPI_BIAS = 0, EPI_QKV = 1, EPI_GELU = 2, EPI_RESID = 3 };
struct KvDst {          // where EPI_QKV writes K/V as fp16
  __half *k, *v;        // base of this layer's cache: [cand][max_pos][1024]
  int S;                // positions per candidate in this launch (row = cand*S + s)
  int n_past, max_pos;
  int replicate;        // >0: the single candidate's rows are written for `replicate` candidates
};
// out[r][n] = epilogue(sum_s part[s][r][n] + bias[n]); one block per row, columns strided by 256.
template <int MODE>
__global__ __launch_bounds__(256) void epilogue_kernel(const float *__restrict__ part, int ks, int rows, int N,
                                                       int n_valid, const float *__restrict__ bias,
                                                       float *__restrict__ out, int ldo, KvDst kv, int lut, float pscale) {
  const int r = blockIdx.x;
  for (int n = blockIdx.y * 256 + threadIdx.x; n < n_valid; n += 256 * gridDim.y) {
    float v = part[(size_t)r * N + n];
    for (int s = 1; s < ks; s++) v += part[((size_t)s * rows + r) * N + n];
    v = v * pscale + bias[n]; // pscale = 1 (f32 path) or 1/64 (split-precision weights are pre-scaled; exact power of 2)
    if (MODE == EPI_QKV) {
      v = f16_round(v);
      out[(size_t)r * ldo + n] = v;
      if (n >= D) {
        const int c = r / kv.S, s = r - c * kv.S, pos = kv.n_past + s;
        __half hv = __float2half_rn(v);
        __half *base = (n < 2 * D) ? kv.k : kv.v;
        const int ch = (n < 2 * D) ? n - D : n - 2 * D;
        if (kv.replicate > 0) {
          for (int cc = 0; cc < kv.replicate; cc++) base[((size_t)cc * kv.max_pos + pos) * D + ch] = hv;
        } else {
          base[((size_t)c * kv.max_pos + pos) * D + ch] = hv;
        }
      }
    } else if (MODE == EPI_GELU) {
      out[(size_t)r * ldo + n] = gelu_tanh(v, lut);
    } else if (MODE == EPI_RESID) {
      out[(size_t)r * ldo + n] += v;
    } else {
      out[(size_t)r * ldo + n] = v;
    }
  }
}

// Causal attention of the multi-row passes (latent pass: 16 candidates x ~200 rows x 16 heads), default (non-LUT) numerics. One workgroup per
// (candidate, head, block of 64 rows): lane = row, and the four waves split the KEYS (wave w takes the groups of 8 keys w, w + 4, ..) so that every SIMD
// holds several waves to hide the LDS latency (one thread per row alone is < 1 wave per SIMD at 3 216 rows). The keys are staged through LDS in chunks of
// 128 (K and V rows of the head, 128 B each: 32 KB per chunk) and walked with wave-uniform LDS addresses (broadcast reads, no bank conflicts): q.k on
// v_dot2_f32_f16 (q is the fp16-rounded query, so the products are exact and the sum is f32), online softmax per group of 8 keys (one rescale of the 64
// accumulators per group), p.V on v_fma_mix_f32 (f32 weight x fp16 V, f32 accumulate). The four partial states (m, l, acc[64]) are merged through LDS.
// Same arithmetic class as attention_kernel below (f32 weights, f32 accumulation; only the summation order and exp2 vs expf differ), which needed one wave
// per (row, head) and re-read every K/V row from L2 per row: 225 us per layer at 3 216 rows.
typedef _Float16 ar_half2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void attention_rows_kernel(const float *__restrict__ qkv, const __half *__restrict__ kc,
                                                             const __half *__restrict__ vc, float *__restrict__ out, int S, int n_past,
                                                             int max_pos) {
  constexpr int CH = 128; // keys per LDS chunk: 32 KB of LDS, three workgroups per CU (the registers allow three waves per SIMD)
  constexpr float L2E = 1.4426950408889634f;
  __shared__ __attribute__((aligned(16))) __half kvs[2 * CH * HD]; // K chunk | V chunk; reused for the merge: acc[wave][32 dims][lane], two passes
  __shared__ float wm[4][64], wl[4][64];
  __half *ks = kvs, *vs = kvs + CH * HD;
  const int c = blockIdx.x, h = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int s0 = blockIdx.z * 64, s = s0 + lane;
  const bool live = s < S;
  const int nk = live ? n_past + s + 1 : 0;         // keys this row sees (ggml_diag_mask_inf(n_past))
  const int nk_block = n_past + min(S, s0 + 64);     // keys the block's last row sees
  ar_half2 q2[HD / 2];
  {
    const float *qp = qkv + (size_t)(c * S + (live ? s : S - 1)) * 3 * D + h * HD;
#pragma unroll
    for (int i = 0; i < HD / 4; i++) {
      const float4 f = ((const float4 *)qp)[i];
      q2[2 * i] = (ar_half2){(_Float16)f.x, (_Float16)f.y};
      q2[2 * i + 1] = (ar_half2){(_Float16)f.z, (_Float16)f.w};
    }
  }
  const __half *kb = kc + (size_t)c * max_pos * D + h * HD;
  const __half *vb = vc + (size_t)c * max_pos * D + h * HD;
  float m = -INFINITY, l = 0.f, acc[HD];
#pragma unroll
  for (int d = 0; d < HD; d++) acc[d] = 0.f;
  for (int j0 = 0; j0 < nk_block; j0 += CH) {
    const int nch = min(CH, nk_block - j0), nch8 = (nch + 7) & ~7;
    __syncthreads(); // the previous chunk has been consumed
    // stage K and V rows j0 .. j0 + nch: 8 threads per row, 16 bytes each (a row of the head = 128 contiguous bytes); the rows that fill the last
    // group of 8 are zeroed (their weights are exp2(-inf) = 0, and 0 x stale LDS bits could be 0 x inf)
    for (int r = tid >> 3; r < nch8; r += 32) {
      const size_t g = (size_t)(j0 + r) * D + (tid & 7) * 8;
      uint4 ku = make_uint4(0u, 0u, 0u, 0u), vu = ku;
      if (r < nch) { ku = *(const uint4 *)(kb + g); vu = *(const uint4 *)(vb + g); }
      *(uint4 *)(ks + r * HD + (tid & 7) * 8) = ku;
      *(uint4 *)(vs + r * HD + (tid & 7) * 8) = vu;
    }
    __syncthreads();
    for (int jg = wave * 8; jg < nch; jg += 32) {
      float sc[8];
      float gmax = m;
#pragma unroll
      for (int e = 0; e < 8; e++) {
        const ar_half2 *kr = (const ar_half2 *)(ks + (jg + e) * HD);
        float d0 = 0.f, d1 = 0.f;
#pragma unroll
        for (int i = 0; i < HD / 2; i += 2) {
          d0 = __builtin_amdgcn_fdot2(q2[i], kr[i], d0, false);
          d1 = __builtin_amdgcn_fdot2(q2[i + 1], kr[i + 1], d1, false);
        }
        const float v = (j0 + jg + e < nk) ? (d0 + d1) * 0.125f : -INFINITY; // 1/sqrt(64)
        sc[e] = v;
        gmax = fmaxf(gmax, v);
      }
      if (gmax == -INFINITY) continue; // nothing visible to this row yet (or a row past S)
      const float resc = __builtin_amdgcn_exp2f((m - gmax) * L2E); // m = -inf -> 0
      l *= resc;
#pragma unroll
      for (int d = 0; d < HD; d++) acc[d] *= resc;
      m = gmax;
#pragma unroll
      for (int e = 0; e < 8; e++) {
        const float pw = __builtin_amdgcn_exp2f((sc[e] - gmax) * L2E); // masked key: exp2(-inf) = 0
        l += pw;
        // acc[d] += pw * V[d] with V read as fp16 by the FMA itself (v_fma_mix_f32: no conversion instructions; hipcc emits cvt + packed FMA otherwise)
        const uint4 *vr = (const uint4 *)(vs + (jg + e) * HD);
#pragma unroll
        for (int i = 0; i < HD / 8; i++) {
          const uint4 u = vr[i];
          const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
          for (int k = 0; k < 4; k++) {
            asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,0]" : "+v"(acc[8 * i + 2 * k]) : "v"(pw), "v"(w[k]));
            asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[0,1,0]" : "+v"(acc[8 * i + 2 * k + 1]) : "v"(pw), "v"(w[k]));
          }
        }
      }
    }
  }
  // merge the four waves' partial states of every row (fixed order: wave 0 .. 3)
  __syncthreads(); // the last chunk has been consumed: its LDS becomes the exchange buffer
  float *xa = (float *)kvs; // [wave][32 dims][lane]: 32 KB, dims 0-31 then dims 32-63
  wm[wave][lane] = m; wl[wave][lane] = l;
  float o[16]; // this wave finishes dims 16 wave .. 16 wave + 15 of all 64 rows
#pragma unroll
  for (int half = 0; half < 2; half++) {
    if (half) __syncthreads();
#pragma unroll
    for (int d = 0; d < 32; d++) xa[(wave * 32 + d) * 64 + lane] = acc[half * 32 + d];
    __syncthreads();
    if ((wave >> 1) == half) { // waves 0, 1 own dims 0-31, waves 2, 3 dims 32-63
      const float M = fmaxf(fmaxf(wm[0][lane], wm[1][lane]), fmaxf(wm[2][lane], wm[3][lane])); // wave 0 holds key 0 of every live row: finite
      float f[4], tot = 0.f;
#pragma unroll
      for (int w = 0; w < 4; w++) {
        f[w] = __builtin_amdgcn_exp2f((wm[w][lane] - M) * L2E); // a wave that saw no key: exp2(-inf) = 0
        tot = fmaf(f[w], wl[w][lane], tot);
      }
      const float inv = 1.0f / tot;
#pragma unroll
      for (int d = 0; d < 16; d++) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 4; w++) t = fmaf(f[w], xa[(w * 32 + (wave & 1) * 16 + d) * 64 + lane], t);
        o[d] = t * inv;
      }
    }
  }
  if (!live) return;
  float *op = out + (size_t)(c * S + s) * D + h * HD + wave * 16;
#pragma unroll
  for (int i = 0; i < 4; i++) ((float4 *)op)[i] = make_float4(o[4 * i], o[4 * i + 1], o[4 * i + 2], o[4 * i + 3]);
}

// Causal attention for one (row, head): q from the f16-rounded qkv buffer, K/V from an fp16 cache
// [cand][max_pos][1024]. Row r = cand*S + s sees keys 0 .. n_past+s (ggml_diag_mask_inf(n_past)).
// One wave per (row, head). Scores in LDS (<= 1024 keys).
__global__ __launch_bounds__(64) void attention_kernel(const float *__restrict__ qkv, const __half *__restrict__ kc,
                                                       const __half *__restrict__ vc, float *__restrict__ out, int S,
                                                       int n_past, int max_pos, int lut) {
  __shared__ float sc[1024];
  __shared__ float qs[HD];
  const int r = blockIdx.x, h = blockIdx.y, lane = threadIdx.x;
  const int c = r / S, s = r - c * S;
  const int nk = n_past + s + 1;
  qs[lane] = qkv[(size_t)r * 3 * D + h * HD + lane];
  __syncthreads();
  const __half *kb = kc + (size_t)c * max_pos * D + h * HD;
  float mx = -INFINITY;
  for (int j = lane; j < nk; j += 64) {
    const uint4 *kp = (const uint4 *)(kb + (size_t)j * D);
    float dot = 0.f;
#pragma unroll
    for (int q = 0; q < 8; q++) {
      uint4 u = kp[q];
      const __half2 *h2 = (const __half2 *)&u;
#pragma unroll
      for (int e = 0; e < 4; e++) {
        float2 f = __half22float2(h2[e]);
        dot = fmaf(qs[q * 8 + e * 2], f.x, dot);
        dot = fmaf(qs[q * 8 + e * 2 + 1], f.y, dot);
      }
    }
    dot *= 0.125f; // 1/sqrt(64)
    sc[j] = dot;
    mx = fmaxf(mx, dot);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  float sum = 0.f;
  for (int j = lane; j < nk; j += 64) {
    float e = lut ? f16_round(expf(f16_round(sc[j] - mx))) : expf(sc[j] - mx);
    sc[j] = e;
    sum += e;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
  __syncthreads();
  const float inv = 1.0f / sum;
  const __half *vb = vc + (size_t)c * max_pos * D + h * HD + lane;
  float acc = 0.f;
  for (int j = 0; j < nk; j++) acc = fmaf(sc[j] * inv, __half2float(vb[(size_t)j * D]), acc);
  out[(size_t)r * D + h * HD + lane] = acc;
}


// ---------------------------------------------------------------------------------------------
// Decode-step kernels (one new position per candidate). The step is captured in a hipGraph: the
// position counters live in device memory so the same graph is replayed for every step.
// ---------------------------------------------------------------------------------------------
struct StepState { int n_past; int pos_id; };
// index of (candidate r, channel k) in the decode step's interleaved residual-stream layout h4 (described in front of the decode kernels)
__host__ __device__ __forceinline__ size_t h4_index(int r, int k) { return ((((size_t)(r >> 4) * 256 + (k >> 2)) * 16 + (r & 15)) << 2) + (k & 3); }

// First kernel of the decode step. `host_step` is the PINNED HOST block [tokens[B] | n_past, pos_id] the host fills before it launches the graph: the kernel
// reads it over PCIe (one ~2 us round trip inside a kernel that has nothing else to wait for) and block 0 leaves the step state in device memory for
// the 150 launches behind it — instead of a host-to-device copy node in front of the graph (a 5 us blit kernel + a launch boundary per step).
__global__ __launch_bounds__(256) void embed_step_kernel(const float *__restrict__ mel_emb, const float *__restrict__ mel_pos,
                                                         const int *__restrict__ host_step, int B, StepState *__restrict__ ss_out,
                                                         float *__restrict__ h4) {
  const int r = blockIdx.x;
  const int tok = host_step[r], n_past = host_step[B], pos_id = host_step[B + 1];
  if (r == 0 && threadIdx.x == 0) { ss_out->n_past = n_past; ss_out->pos_id = pos_id; }
  const float4 a = ((const float4 *)(mel_emb + (size_t)tok * D))[threadIdx.x];
  const float4 b = ((const float4 *)(mel_pos + (size_t)pos_id * D))[threadIdx.x];
  *(float4 *)(h4 + h4_index(r, 4 * threadIdx.x)) = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); // h4 layout, see below
}

// Round 4 — the residual stream of the DECODE step lives in an interleaved layout, `h4`:
//     element (candidate r, channel k)  ->  h4[r / 16][k / 4][r % 16][k % 4]       (16-byte groups of 4 channels, the 16 candidates of a tile adjacent)
// The LayerNorm-GEMV kernels feed the activations to the matrix pipe as the B operand: lane (m = candidate, q) holds 8 consecutive channels, so with the
// natural [r][1024] layout one wave-level load touched 16 rows x 64 B — sixteen half-used cache lines per instruction, twice each. That is the pattern
// cdna_hip_programming.md warns about ("fragment-shaped loads (16 rows x 64 B per instruction) straight to VGPRs ... TA_BUSY 2x") and the per-launch trace
// shows its price: ~3 us until the 64 KB of activations are in, against <1 us for the same 64 KB read as 1 KB-contiguous instructions by
// dec_gemv_resid_kernel (profiles/r4_decode_launch_breakdown.txt). In h4 the same instruction covers 4 x 256 contiguous bytes = 8 full lines, nothing twice;
// the projection kernels' 64 outputs per workgroup (16 candidates x 4 columns) become ONE contiguous 256-byte block. The prompt pass keeps [row][1024].

// ---- decode step: five launches per layer, no split-K partials ---------------------------------------
// Every weight matrix is packed at load for the workgroup that streams it (one contiguous slab per
// workgroup, every wave-level load instruction 1 KB of consecutive bytes) and every launch writes COMPLETE
// outputs, so LayerNorm, bias, GELU, the fp16 rounding of QKV, the KV-cache append and the residual add all
// live in the prologue/epilogue of the kernel that streams the weights:
//   dec_ln_gemv<QKV>  : LN1(h) . c_attn -> q (fp32 of the fp16-rounded value), K/V appended to the fp16 cache
//   attn_decode       : softmax(q.K/8).V over n_past+1 keys
//   dec_gemv_resid<4> : h += att . c_proj + b
//   dec_ln_gemv<GELU> : gelu(LN2(h) . c_fc + b) -> ff
//   dec_gemv_resid<16>: h += ff . c_proj2 + b
// and the head is dec_ln_gemv<LOGITS> (ln_f, lm_head LayerNorm, lm_head linear) straight to the logits.

// K = 1024 GEMV with LayerNorm prologue for 16 candidates x 16 output columns per workgroup. Default: split-precision
// operands on the fp16 MFMA (x = xh + xl split in registers, 64 W = wh + wl packed at load; wh.xh + wl.xh + wh.xl, 24
// MFMAs of 16 cycles per wave); TTS_DEC_F32MFMA=1: the fp32 MFMA (v_mfma_f32_16x16x4_f32, exact f32 products, 64 MFMAs of
// 32 cycles). Wave w covers k in [256w, 256w+256). Operands are fed swapped (A = weights,
// B = activations) so that a lane ends with 4 consecutive output columns of one candidate. The 4 k values of
// one MFMA are k0 + 4q + j for lane quarter q — a lane's activations are then one float4 of the natural
// [row][k] layout; the weights are packed to match (pack_mfma16 below). LayerNorm is evaluated on the
// register-resident operands: a row's 1024 values live in 4 lanes x 4 waves.
typedef float floatx2 __attribute__((ext_vector_type(2)));
enum { DEC_QKV = 0, DEC_GELU = 1, DEC_LOGITS = 2 };
#ifdef TTS_DEC_TRACE // developer build (tools/dec_bench.hip): phase timestamps (100 MHz wall clock) of every workgroup, one slot per kernel of a layer
__device__ long long tts_dec_trace[6 * 1024 * 8]; // slot: 0 LN1+QKV, 1 attention, 2 attention projection, 3 LN2+FC, 4 MLP projection, 5 head
#define DEC_T(i) do { if (threadIdx.x == 0) tts_dec_trace[((TSLOT) * 1024 + blockIdx.y * gridDim.x + blockIdx.x) * 8 + (i)] = wall_clock64(); } while (0)
#else
#define DEC_T(i)
#endif
struct DecLnArgs {
  const float *h;            // [rows][1024]
  const float *g1, *b1;      // DEC_LOGITS: ln_f (the LayerNorm feeding W is folded into W/bias at load)
  const float *W;            // pack_mfma16 of diag(gamma) W
  const __half *Wh;          // SPLIT 1: pack_mfma16h, the same matrix x 64 as fp16 hi|lo pairs; SPLIT 2: pack_mfma16q, fp16 hi only (option ar_weights = 1);
                             // SPLIT 3: pack_mfma16o, OCP fp8 e4m3 of W / wscale[column] (option ar_weights = 2)
  const float *bias;         // bias + beta . W
  int rows, n_valid, ldo;    // ldo: row stride of `out` for DEC_QKV (q) and DEC_LOGITS
  int prefill_B;             // DEC_QKV: 0 = decode (row = candidate, position n_past); > 0 = prompt pass (row = position,
                             // K/V replicated into the caches of prefill_B candidates)
  float *out;                // q [rows][ldo] | ff [rows][4096] | logits [rows][ldo]
  __half *kc, *vc;           // layer's caches [cand][max_pos][1024]
  const StepState *ss;
  int max_pos, lut;
  const float *wscale;       // SPLIT 3: per-output-column scale (a power of two) of the fp8 weights
};

// sum over the four 16-lane rows of a wave (lanes l, l^16, l^32, l^48): gfx950 half/row swap instructions
__device__ __forceinline__ float rows4_sum(float x) {
  auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  x = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

// sred: two [4][16] arrays (one per pass, so each pass costs one barrier). g == nullptr: the affine part has
// been folded into the weights/bias at load (fold_layernorm).
// KSTEP: k of the i-th float4 of a lane = koff + (i >> KSH) * KSTEP + (i & ((1 << KSH) - 1)) * 4
//   fp32-MFMA operand order: KSH = 0, KSTEP = 16;  fp16-MFMA (split) order: KSH = 1, KSTEP = 32
template <int KSH, int KSTEP>
__device__ __forceinline__ void dec_layernorm(float4 (&x)[16], const float *__restrict__ g, const float *__restrict__ b,
                                              int koff, float (*sred)[4][16], int wave, int m) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; i++) s += (x[i].x + x[i].y) + (x[i].z + x[i].w);
  sred[0][wave][m] = rows4_sum(s); // the 4 lanes of a row write the same value
  __syncthreads();
  const float mean = (((sred[0][0][m] + sred[0][1][m]) + sred[0][2][m]) + sred[0][3][m]) * (1.0f / D);
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < 16; i++) {
    x[i].x -= mean; x[i].y -= mean; x[i].z -= mean; x[i].w -= mean;
    ss += (x[i].x * x[i].x + x[i].y * x[i].y) + (x[i].z * x[i].z + x[i].w * x[i].w);
  }
  sred[1][wave][m] = rows4_sum(ss);
  __syncthreads();
  const float var = (((sred[1][0][m] + sred[1][1][m]) + sred[1][2][m]) + sred[1][3][m]) * (1.0f / D);
  const float sc = 1.0f / sqrtf(var + 1e-5f);
  if (g) {
#pragma unroll
    for (int i = 0; i < 16; i++) {
      const int k = koff + (i >> KSH) * KSTEP + (i & ((1 << KSH) - 1)) * 4;
      const float4 gg = *(const float4 *)(g + k), bb = *(const float4 *)(b + k);
      x[i].x = x[i].x * sc * gg.x + bb.x; x[i].y = x[i].y * sc * gg.y + bb.y;
      x[i].z = x[i].z * sc * gg.z + bb.z; x[i].w = x[i].w * sc * gg.w + bb.w;
    }
  } else {
#pragma unroll
    for (int i = 0; i < 16; i++) { x[i].x *= sc; x[i].y *= sc; x[i].z *= sc; x[i].w *= sc; }
  }
}

// SPLIT: 0 = fp32 MFMA on f32 weights, 1 = split-precision fp16 MFMA (f32-exact to 2^-22), 2 = fp16 WEIGHTS (rounded once at load:
// half the bytes streamed; the activations keep their hi + lo split) — the throughput mode of SURVEY 8d, option "ar_weights";
// 3 = OCP fp8 (e4m3) WEIGHTS with a power-of-two scale per output column (a quarter of the bytes): converted to fp16 in
// registers (exact: every e4m3 value is an fp16 value), multiplied on the fp16 MFMA against the hi + lo split activations.
template <int EPI, int SPLIT = 0, bool NTW = false, bool HT = false> // HT: a.h is in the h4 layout (decode step); otherwise [row][1024] (prompt pass)
__global__ __launch_bounds__(256) void dec_ln_gemv_kernel(DecLnArgs a_in) {
  constexpr int SP = SPLIT ? 1 : 0;
  constexpr int TSLOT = EPI == DEC_QKV ? 0 : EPI == DEC_GELU ? 3 : 5; (void)TSLOT;
  __shared__ float sred[4][4][16];
  __shared__ float4 accs[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, m = lane & 15, q = lane >> 4;
  const int cb = blockIdx.x, row = blockIdx.y * 16 + m;
  DEC_T(0);
  // EVERY kernel argument is fetched in ONE batch of scalar loads at the top (the empty asm pins the values in SGPRs here). Left to itself
  // hipcc fetches an argument right before its first use: the 104-byte argument block spans two cache lines and nothing of a fresh launch is
  // cached, so the QKV variant ran THREE dependent scalar round trips (arguments, the pointer to the step state, n_past) before its first
  // vector load was issued, and every variant fetched `out` / `lut` (the second line) after the reduction — a cold miss at the very end of
  // each of the step's 61 launches of this kernel (seen in the ISA and in the per-launch trace, profiles/r4_decode_launch_breakdown.txt).
  const DecLnArgs &a = a_in;
  // (input-only operands: the values must be in SGPRs HERE, and the pointers keep their global address space — behind a "+s" they would
  //  become generic pointers and every load a flat_load, which also counts on lgkmcnt)
  asm volatile("" ::"s"(a.h), "s"(a.g1), "s"(a.b1), "s"(a.W), "s"(a.Wh), "s"(a.bias), "s"(a.out), "s"(a.kc), "s"(a.vc), "s"(a.ss), "s"(a.wscale));
  asm volatile("" ::"s"(a.rows), "s"(a.n_valid), "s"(a.ldo), "s"(a.prefill_B), "s"(a.max_pos), "s"(a.lut));
  // Everything the epilogue reads (bias, the step's n_past) is requested FIRST: loaded after the reduction they were one or two
  // dependent L2 round trips (~0.5 us each) at the end of every launch. n_past goes through the VECTOR memory path (a zero offset the
  // compiler cannot see through): as a scalar load it is a dependent round trip that every later s_waitcnt lgkmcnt(0) waits for; as the
  // first entry of the in-order vmcnt queue it retires before the activations and nothing waits for it until the epilogue.
  const float4 bi = *(const float4 *)(a.bias + cb * 16 + 4 * q);
  float4 wsc = make_float4(1.f, 1.f, 1.f, 1.f);
  if (SPLIT == 3) wsc = *(const float4 *)(a.wscale + cb * 16 + 4 * q);
  int n_past = 0;
  if (EPI == DEC_QKV) {
    unsigned zero = 0;
    asm volatile("" : "+v"(zero));
    const int *np_ptr = a.prefill_B == 0 ? &a.ss->n_past : (const int *)a.bias; // the prompt pass has no step state (ss may be null): read a valid dummy, drop it
    n_past = *(const int *)((const char *)np_ptr + zero);
    n_past = a.prefill_B == 0 ? n_past : 0;
  }
  // activations next (L2 hits), then the weight slab (HBM): vmcnt retires in order, so the LayerNorm runs on
  // the activations while the 16 x 1 KB-per-wave weight loads are still streaming in
  const int koff = wave * 256 + (SPLIT ? 8 : 4) * q;
  float4 x[16];
  {
    if (HT) {
      // h4: the float4 of (candidate m, channels k .. k+3) sits at ((tile * 256 + k / 4) * 16 + m) * 4; a wave-level load = 4 runs of 256 contiguous
      // bytes (the 16 candidates of one channel group), 8 full cache lines. Rows past the batch are padding of the tile (allocated, never stored).
      const float *hp = a.h + (((size_t)blockIdx.y * 256 + (koff >> 2)) * 16 + m) * 4;
#pragma unroll
      for (int i = 0; i < 16; i++) x[i] = *(const float4 *)(hp + (SPLIT ? (i >> 1) * 8 + (i & 1) : i * 4) * 64);
    } else {
      // rows past the batch re-read the last candidate (branch-free); their results are never stored
      const float *hp = a.h + (size_t)min(row, a.rows - 1) * D + koff;
#pragma unroll
      for (int i = 0; i < 16; i++) x[i] = *(const float4 *)(hp + (SPLIT ? (i >> 1) * 32 + (i & 1) * 4 : i * 16));
    }
  }
  float4 w[16]; // SPLIT: step s = (w[2s] = 8 fp16 hi, w[2s+1] = 8 fp16 lo) of 64*W[k = koff + 32 s + e][col]
  {
    if (SPLIT == 2) { // 8 steps x 16 B per lane: half the slab of the split variant
      const float4 *wp = (const float4 *)a.Wh + ((size_t)(cb * 4 + wave) * 8) * 64;
#pragma unroll
      for (int i = 0; i < 8; i++) w[2 * i] = ldw4<NTW>(wp + i * 64 + lane);
    } else if (SPLIT == 3) { // 8 steps x 8 B per lane, two steps per 16-byte load: w[i] = steps 2 i, 2 i + 1
      const float4 *wp = (const float4 *)a.Wh + ((size_t)(cb * 4 + wave) * 4) * 64;
#pragma unroll
      for (int i = 0; i < 4; i++) w[i] = ldw4<NTW>(wp + i * 64 + lane);
    } else {
      const float4 *wp = (SPLIT ? (const float4 *)a.Wh : (const float4 *)a.W) + ((size_t)(cb * 4 + wave) * 16) * 64;
#pragma unroll
      for (int i = 0; i < 16; i++) w[i] = ldw4<NTW>(SPLIT ? wp + ((i >> 1) * 64 + lane) * 2 + (i & 1) : wp + i * 64 + lane);
    }
  }
  // Every request above stays in front of the first use: without this hipcc sinks the 16 weight loads below the LayerNorm's first
  // reduction (they would be issued only after `s_waitcnt vmcnt(0)` on the activations: the HBM round trip of the slab serialised
  // behind the L2 round trip of x instead of overlapping it; seen in the ISA, round 4).
  __builtin_amdgcn_sched_barrier(0);
  DEC_T(1);
  // the (last) LayerNorm's gamma/beta are folded into W/bias; the head's ln_f keeps its own
  if (EPI == DEC_LOGITS) dec_layernorm<SP, SPLIT ? 32 : 16>(x, a.g1, a.b1, koff, sred, wave, m);
  dec_layernorm<SP, SPLIT ? 32 : 16>(x, nullptr, nullptr, koff, sred + (EPI == DEC_LOGITS ? 2 : 0), wave, m);
  DEC_T(2);
  floatx4 acc;
  if (SPLIT) {
    // split precision on the fp16 MFMA: x = xh + xl, 64 W = wh + wl; wh.xh + wl.xh + wh.xl (the dropped wl.xl term is
    // 2^-22 relative), 24 MFMAs of 16 cycles instead of 64 fp32 MFMAs of 32
    floatx4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0;
#pragma unroll
    for (int s = 0; s < 8; s++) {
      const float xs[8] = {x[2 * s].x, x[2 * s].y, x[2 * s].z, x[2 * s].w, x[2 * s + 1].x, x[2 * s + 1].y, x[2 * s + 1].z, x[2 * s + 1].w};
      half8 xh, xl;
#pragma unroll
      for (int e = 0; e < 8; e++) {
        xh[e] = (_Float16)xs[e];
        xl[e] = (_Float16)(xs[e] - (float)xh[e]);
      }
      half8 wh;
      if (SPLIT == 3) {
        const uint2 u = ((const uint2 *)&w[s >> 1])[s & 1];
        const floatx2 f0 = __builtin_amdgcn_cvt_pk_f32_fp8(u.x, false), f1 = __builtin_amdgcn_cvt_pk_f32_fp8(u.x, true);
        const floatx2 f2 = __builtin_amdgcn_cvt_pk_f32_fp8(u.y, false), f3 = __builtin_amdgcn_cvt_pk_f32_fp8(u.y, true);
        wh[0] = (_Float16)f0[0]; wh[1] = (_Float16)f0[1]; wh[2] = (_Float16)f1[0]; wh[3] = (_Float16)f1[1];
        wh[4] = (_Float16)f2[0]; wh[5] = (_Float16)f2[1]; wh[6] = (_Float16)f3[0]; wh[7] = (_Float16)f3[1];
      } else wh = *(const half8 *)&w[2 * s];
      a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xh, a0, 0, 0, 0);
      if (SPLIT == 1) {
        const half8 wl = *(const half8 *)&w[2 * s + 1];
        a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, xh, a1, 0, 0, 0);
      }
      a2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xl, a2, 0, 0, 0);
    }
    if (SPLIT == 3) { acc = a2 + a0; acc[0] *= wsc.x; acc[1] *= wsc.y; acc[2] *= wsc.z; acc[3] *= wsc.w; }
    else acc = ((a1 + a2) + a0) * (1.0f / 64.0f);
  } else {
    // four independent accumulator chains keep the fp32 MFMA pipe issue-bound instead of latency-bound
    floatx4 ac0 = {0.f, 0.f, 0.f, 0.f}, ac1 = ac0, ac2 = ac0, ac3 = ac0;
#pragma unroll
    for (int i = 0; i < 16; i++) {
      ac0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w[i].x, x[i].x, ac0, 0, 0, 0);
      ac1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w[i].y, x[i].y, ac1, 0, 0, 0);
      ac2 = __builtin_amdgcn_mfma_f32_16x16x4f32(w[i].z, x[i].z, ac2, 0, 0, 0);
      ac3 = __builtin_amdgcn_mfma_f32_16x16x4f32(w[i].w, x[i].w, ac3, 0, 0, 0);
    }
    acc = (ac0 + ac1) + (ac2 + ac3);
  }
  DEC_T(3);
  accs[wave][lane] = make_float4(acc[0], acc[1], acc[2], acc[3]);
  __syncthreads();
  DEC_T(4);
  if (wave != 0 || row >= a.rows) return;
  const float4 p0 = accs[0][lane], p1 = accs[1][lane], p2 = accs[2][lane], p3 = accs[3][lane];
  const int col = cb * 16 + 4 * q; // lane: candidate `row`, columns col .. col+3
  float4 v;
  v.x = (((p0.x + p1.x) + p2.x) + p3.x) + bi.x; v.y = (((p0.y + p1.y) + p2.y) + p3.y) + bi.y;
  v.z = (((p0.z + p1.z) + p2.z) + p3.z) + bi.z; v.w = (((p0.w + p1.w) + p2.w) + p3.w) + bi.w;
  if (EPI == DEC_QKV) {
    // QKV activations are rounded to fp16 (main.cpp:2789-2790)
    const __half2 h01 = __floats2half2_rn(v.x, v.y), h23 = __floats2half2_rn(v.z, v.w);
    const int sec = col >> 10, cc = col & (D - 1);
    if (sec == 0) {
      const float2 f01 = __half22float2(h01), f23 = __half22float2(h23);
      *(float4 *)(a.out + (size_t)row * a.ldo + cc) = make_float4(f01.x, f01.y, f23.x, f23.y);
    } else {
      uint2 u;
      u.x = *(const unsigned *)&h01;
      u.y = *(const unsigned *)&h23;
      __half *base = (sec == 1 ? a.kc : a.vc) + cc;
      if (a.prefill_B == 0) *(uint2 *)(base + ((size_t)row * a.max_pos + n_past) * D) = u;
      else
        for (int c = 0; c < a.prefill_B; c++) *(uint2 *)(base + ((size_t)c * a.max_pos + row) * D) = u;
    }
  } else if (EPI == DEC_GELU) {
    v.x = gelu_tanh(v.x, a.lut); v.y = gelu_tanh(v.y, a.lut); v.z = gelu_tanh(v.z, a.lut); v.w = gelu_tanh(v.w, a.lut);
    *(float4 *)(a.out + (size_t)row * FF + col) = v;
  } else {
    float *o = a.out + (size_t)row * a.ldo + col;
    if (col + 0 < a.n_valid) o[0] = v.x;
    if (col + 1 < a.n_valid) o[1] = v.y;
    if (col + 2 < a.n_valid) o[2] = v.z;
    if (col + 3 < a.n_valid) o[3] = v.w;
  }
  DEC_T(5);
}

// h[rows][1024] += X[rows][K] . W[K][1024] + bias, K = 1024 * KG. One workgroup owns 4 output columns over the
// whole K (W packed by pack_cols4: 16 KB x KG contiguous per workgroup); thread t holds k = 1024 i + 4 t + kk.
// The 64 per-thread sums (16 candidates x 4 columns) are reduced with a 6-step exchange butterfly that
// leaves output t on lane t, then across the 4 waves through LDS — a fixed summation tree.
// NT threads per workgroup (256 or 512): with 512 the K range of a thread halves and twice as many activation loads
// are in flight per CU — the c_proj of the MLP (K = 4096) reads 256 KB of activations per workgroup and is bound by how
// many of those loads the CU keeps in flight.
// WH: the slab holds fp16 weights (pack_cols4 order, 8 bytes per (k, 4 columns): option ar_weights = 1), converted to f32 in registers.
// WH = 2: OCP fp8 (e4m3) weights, 4 bytes per (k, 4 columns), times the power-of-two wscale[column] after the reduction (option ar_weights = 2).
template <int KG, int NT = 256, int WH = 0, bool NTW = false, bool HT = false> // HT: h is in the h4 layout (decode step)
__global__ __launch_bounds__(NT) void dec_gemv_resid_kernel(const float *__restrict__ X, int rows, const float *__restrict__ W,
                                                            const float *__restrict__ bias, float *__restrict__ h,
                                                            const float *__restrict__ wscale = nullptr) {
  constexpr int K = 1024 * KG, NW = NT / 64, NG = K / (NT * 4); // NG K groups of NT*4 values per workgroup
  constexpr int TSLOT = KG == 1 ? 2 : 4; (void)TSLOT;
  __shared__ float red[NW][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cb = blockIdx.x, row0 = blockIdx.y * 16;
  DEC_T(0);
  // the residual element and the bias this thread adds at the very end (threads 0-63: output 4 * candidate + column) are requested
  // first instead of after the reduction (a dependent L2 round trip at the end of 60 launches per step)
  const int er = min(row0 + ((tid & 63) >> 2), rows - 1), ecol = cb * 4 + (tid & 3);
  // h4: the workgroup's 16 candidates x 4 columns are 64 consecutive floats (thread t < 64 owns float t of the block)
  const size_t hidx = HT ? (((size_t)blockIdx.y * 256 + cb) * 64 + (tid & 63)) : (size_t)er * D + ecol;
  const float h_old = h[hidx], b_old = bias[ecol];
  float s_old = 1.0f;
  if (WH == 2) s_old = wscale[ecol];
  float4 xa0[8]; // candidates 0-7 of K group 0: requested before the weight stream
#pragma unroll
  for (int r = 0; r < 8; r++) xa0[r] = *(const float4 *)(X + (size_t)min(row0 + r, rows - 1) * K + 4 * tid);
  float4 w[NG][4];
  {
    // pack_cols4 order: float4 index ((k / 1024) * 4 + kk) * 256 + (k % 1024) / 4 for k = g * NT * 4 + 4 * tid + kk
    if (WH == 2) {
      const unsigned *wp = (const unsigned *)W + (size_t)cb * KG * 1024 + (tid & 255);
#pragma unroll
      for (int i = 0; i < NG; i++)
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
          const unsigned u = wp[((i * (NT / 256) + (tid >> 8)) * 4 + kk) * 256];
          const floatx2 f0 = __builtin_amdgcn_cvt_pk_f32_fp8(u, false), f1 = __builtin_amdgcn_cvt_pk_f32_fp8(u, true);
          w[i][kk] = make_float4(f0[0], f0[1], f1[0], f1[1]);
        }
    } else if (WH == 1) {
      const uint2 *wp = (const uint2 *)W + (size_t)cb * KG * 1024 + (tid & 255);
#pragma unroll
      for (int i = 0; i < NG; i++)
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
          const uint2 u = wp[((i * (NT / 256) + (tid >> 8)) * 4 + kk) * 256];
          const float2 f0 = __half22float2(*(const __half2 *)&u.x), f1 = __half22float2(*(const __half2 *)&u.y);
          w[i][kk] = make_float4(f0.x, f0.y, f1.x, f1.y);
        }
    } else {
      const float4 *wp = (const float4 *)W + (size_t)cb * KG * 1024 + (tid & 255);
#pragma unroll
      for (int i = 0; i < NG; i++)
#pragma unroll
        for (int kk = 0; kk < 4; kk++) w[i][kk] = ldw4<NTW>(wp + ((i * (NT / 256) + (tid >> 8)) * 4 + kk) * 256);
    }
  }
  // Activations (L2 hits) are fetched 8 candidates at a time, double-buffered against the FMAs; K group i of the
  // weight slab is consumed for all 16 candidates as soon as it has arrived (vmcnt retires in order), so the
  // FMAs of groups 0..NG-2 overlap the rest of the weight stream. Two columns per v_pk_fma_f32.
  floatx2 acc[32];
#pragma unroll
  for (int j = 0; j < 32; j++) acc[j] = (floatx2){0.f, 0.f};
  const float *xbase = X + 4 * tid;
  int rowoff[16];
#pragma unroll
  for (int r = 0; r < 16; r++) rowoff[r] = min(row0 + r, rows - 1) * K; // rows past the batch re-read the last candidate
  float4 xa[8], xb[8];
#pragma unroll
  for (int r = 0; r < 8; r++) { xa[r] = xa0[r]; xb[r] = *(const float4 *)(xbase + rowoff[8 + r]); }
#pragma unroll
  for (int i = 0; i < NG; i++) {
#pragma unroll
    for (int half = 0; half < 2; half++) {
      // half 0 consumes xa (candidates 0-7), half 1 consumes xb (8-15); the other buffer is refilled meanwhile
      float4 (&cur)[8] = half ? xb : xa;
#pragma unroll
      for (int r = 0; r < 8; r++) {
        const float xs[4] = {cur[r].x, cur[r].y, cur[r].z, cur[r].w};
        floatx2 a01 = acc[(half * 8 + r) * 2], a23 = acc[(half * 8 + r) * 2 + 1];
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
          const floatx2 xx = {xs[kk], xs[kk]};
          a01 = __builtin_elementwise_fma(xx, (floatx2){w[i][kk].x, w[i][kk].y}, a01);
          a23 = __builtin_elementwise_fma(xx, (floatx2){w[i][kk].z, w[i][kk].w}, a23);
        }
        acc[(half * 8 + r) * 2] = a01; acc[(half * 8 + r) * 2 + 1] = a23;
      }
      if (i + 1 < NG) { // refill the buffer just consumed with the next K group
#pragma unroll
        for (int r = 0; r < 8; r++) cur[r] = *(const float4 *)(xbase + rowoff[half * 8 + r] + (i + 1) * (NT * 4));
      }
    }
  }
  float v[64];
#pragma unroll
  for (int j = 0; j < 32; j++) { v[2 * j] = acc[j][0]; v[2 * j + 1] = acc[j][1]; }
  DEC_T(1);
  // exchange butterfly: after the step with lane mask M a lane keeps the half of its values selected by its
  // bit M, summed with the partner lane's copy; 64 values -> 1, output index = lane.
#pragma unroll
  for (int i = 0; i < 32; i++) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[i]), __float_as_uint(v[i + 32]), false, false);
    v[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  }
#pragma unroll
  for (int i = 0; i < 16; i++) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v[i]), __float_as_uint(v[i + 16]), false, false);
    v[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  }
#pragma unroll
  for (int step = 2; step < 6; step++) {
    const int mask = 32 >> step, n = 32 >> step;
    const bool up = (lane & mask) != 0;
#pragma unroll
    for (int i = 0; i < n; i++) {
      const float send = up ? v[i] : v[i + n], keep = up ? v[i + n] : v[i];
      v[i] = keep + __shfl_xor(send, mask);
    }
  }
  DEC_T(2);
  red[wave][lane] = v[0]; // output index = lane = 4 * candidate + column
  __syncthreads();
  DEC_T(3);
  if (tid < 64) {
    const int r = row0 + (tid >> 2), col = cb * 4 + (tid & 3);
    if (r < rows) {
      float t = red[0][tid];
#pragma unroll
      for (int w2 = 1; w2 < NW; w2++) t += red[w2][tid];
      h[HT ? hidx : (size_t)r * D + col] = h_old + ((WH == 2 ? t * s_old : t) + b_old);
    }
  }
  DEC_T(4);
}

// max over the four 16-lane rows of a wave (see rows4_sum)
__device__ __forceinline__ float rows4_max(float x) {
  auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  x = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
  auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float x) { // lane i reads x of the lane the DPP control names (full row/bank masks)
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, false));
}
typedef _Float16 dhalf2 __attribute__((ext_vector_type(2)));

// 32 * NR keys starting at j0 for one (candidate, head): 8 lanes per key (16 B of K and of V each: whole 128 B rows per
// 8 lanes), key group kg = tid >> 3 owns keys j0 + kg + 32 r. All K and V rows of the chunk are requested before the
// first use, so the chunk costs one memory round trip. Online softmax state (m, l, acc) is per wave: m is uniform over
// the wave, l and acc are the lane's partial sums over its own keys.
template <int NR>
__device__ __forceinline__ void attn_decode_chunk(const __half *__restrict__ kb, const __half *__restrict__ vb, int j0, int nk, int kg,
                                                  int l8, const float4 &qa, const float4 &qb, float &m, float &l, float (&acc)[8]) {
  constexpr float L2E = 1.4426950408889634f;
  uint4 kq[NR], vq[NR];
#pragma unroll
  for (int r = 0; r < NR; r++) // uniform base + 32-bit byte offset (one candidate's cache is < 4 GB): saddr-form loads
    kq[r] = *(const uint4 *)((const char *)kb + ((unsigned)min(j0 + kg + 32 * r, nk - 1) * (unsigned)(D * 2) + (unsigned)l8 * 16u));
#pragma unroll
  for (int r = 0; r < NR; r++)
    vq[r] = *(const uint4 *)((const char *)vb + ((unsigned)min(j0 + kg + 32 * r, nk - 1) * (unsigned)(D * 2) + (unsigned)l8 * 16u));
  __builtin_amdgcn_sched_barrier(0); // keep every request in front of the first use (the scheduler would sink the V loads)
  // q was rounded to fp16 by the QKV epilogue: the conversion back is exact. The empty asm pins the first use of q
  // behind the requests above (the conversion is common to all chunk sizes and would be hoisted in front of them).
  float4 qc = qa, qd = qb;
  asm volatile("" : "+v"(qc.x), "+v"(qc.y), "+v"(qc.z), "+v"(qc.w), "+v"(qd.x), "+v"(qd.y), "+v"(qd.z), "+v"(qd.w));
  const dhalf2 q2[4] = {{(_Float16)qc.x, (_Float16)qc.y}, {(_Float16)qc.z, (_Float16)qc.w},
                        {(_Float16)qd.x, (_Float16)qd.y}, {(_Float16)qd.z, (_Float16)qd.w}};
  float s[NR], mc = -INFINITY;
#pragma unroll
  for (int r = 0; r < NR; r++) {
    const dhalf2 *k2 = (const dhalf2 *)&kq[r];
    float d = __builtin_amdgcn_fdot2(k2[0], q2[0], 0.f, false);
    d = __builtin_amdgcn_fdot2(k2[1], q2[1], d, false);
    d = __builtin_amdgcn_fdot2(k2[2], q2[2], d, false);
    d = __builtin_amdgcn_fdot2(k2[3], q2[3], d, false);
    d += dpp_f32<0xB1>(d);  // quad_perm [1,0,3,2]
    d += dpp_f32<0x4E>(d);  // quad_perm [2,3,0,1]
    d += dpp_f32<0x141>(d); // row_half_mirror: lane i <-> 7 - i of each 8
    s[r] = (j0 + kg + 32 * r < nk) ? d * 0.125f : -INFINITY;
    mc = fmaxf(mc, s[r]);
  }
  mc = rows4_max(fmaxf(mc, __shfl_xor(mc, 8)));
  const float mn = fmaxf(m, mc);
  const float mu = (mn == -INFINITY) ? 0.f : mn; // a wave without a valid key yet: all weights 0
  const float alpha = __builtin_amdgcn_exp2f((m - mu) * L2E);
  l *= alpha;
#pragma unroll
  for (int e = 0; e < 8; e++) acc[e] *= alpha;
#pragma unroll
  for (int r = 0; r < NR; r++) {
    const float p = __builtin_amdgcn_exp2f((s[r] - mu) * L2E);
    l += p;
    const __half2 *v2 = (const __half2 *)&vq[r];
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const float2 f = __half22float2(v2[e]);
      acc[2 * e] = fmaf(p, f.x, acc[2 * e]);
      acc[2 * e + 1] = fmaf(p, f.y, acc[2 * e + 1]);
    }
  }
  m = mn;
}

// Decode attention for one (candidate, head), the default (non-LUT) path: softmax(q.K/8) V over n_past+1 keys with the
// hardware exp2 (relative error ~1e-6 on a weight). Up to 288 keys go through one chunk = one memory round trip
// (attn_decode_chunk); the four waves keep separate online-softmax states that are merged once at the end.
__global__ __launch_bounds__(256) void attn_decode_fast_kernel(const float *__restrict__ qbuf, const __half *__restrict__ kc,
                                                               const __half *__restrict__ vc, const StepState *__restrict__ ss,
                                                               int max_pos, float *__restrict__ out) {
  constexpr float L2E = 1.4426950408889634f;
  constexpr int TSLOT = 1; (void)TSLOT;
  __shared__ float red[4][HD];
  __shared__ float wm[4], wl[4];
  const int c = blockIdx.x, h = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l8 = tid & 7, kg = tid >> 3;
  DEC_T(0);
  // all arguments in one batch of scalar loads (see dec_ln_gemv_kernel) and the one dependent fetch this kernel cannot avoid: n_past.
  // (the load of n_past stays IN FRONT of the asm: behind a volatile asm hipcc no longer proves the memory unclobbered and fetches it through
  //  the vector path, which makes the loop bounds divergent)
  const int nk = ss->n_past + 1;
  asm volatile("" ::"s"(qbuf), "s"(kc), "s"(vc), "s"(max_pos), "s"(out)); // input-only: the pointers keep their address space
  const __half *kb = kc + (size_t)c * max_pos * D + h * HD;
  const __half *vb = vc + (size_t)c * max_pos * D + h * HD;
  const float4 qa = *(const float4 *)(qbuf + (size_t)c * D + h * HD + l8 * 8), qb = *(const float4 *)(qbuf + (size_t)c * D + h * HD + l8 * 8 + 4);
  float m = -INFINITY, l = 0.f, acc[8];
#pragma unroll
  for (int e = 0; e < 8; e++) acc[e] = 0.f;
  for (int j0 = 0; j0 < nk; j0 += 288) {
    const int rem = nk - j0; // workgroup-uniform
    if (rem <= 96) attn_decode_chunk<3>(kb, vb, j0, nk, kg, l8, qa, qb, m, l, acc);
    else if (rem <= 160) attn_decode_chunk<5>(kb, vb, j0, nk, kg, l8, qa, qb, m, l, acc);
    else if (rem <= 224) attn_decode_chunk<7>(kb, vb, j0, nk, kg, l8, qa, qb, m, l, acc);
    else attn_decode_chunk<9>(kb, vb, j0, nk, kg, l8, qa, qb, m, l, acc);
  }
  // sum over the wave's 8 key groups (lane bits 3-5). l: plain reduction. acc: exchange butterfly, 8 values -> the
  // wave total of dim l8 * 8 + (lane >> 3)
  l += __shfl_xor(l, 8);
  l = rows4_sum(l);
#pragma unroll
  for (int i = 0; i < 4; i++) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[i]), __float_as_uint(acc[i + 4]), false, false);
    acc[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  }
#pragma unroll
  for (int i = 0; i < 2; i++) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[i]), __float_as_uint(acc[i + 2]), false, false);
    acc[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  }
  {
    const bool up = (lane & 8) != 0;
    const float send = up ? acc[0] : acc[1], keep = up ? acc[1] : acc[0];
    acc[0] = keep + __shfl_xor(send, 8);
  }
  DEC_T(1);
  red[wave][l8 * 8 + (lane >> 3)] = acc[0];
  if (lane == 0) { wm[wave] = m; wl[wave] = l; }
  __syncthreads();
  DEC_T(2);
  if (tid < HD) {
    const float M = fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3])); // wave 0 always holds key 0: finite
    float tot = 0.f, o = 0.f;
#pragma unroll
    for (int w = 0; w < 4; w++) {
      const float sc = __builtin_amdgcn_exp2f((wm[w] - M) * L2E); // exp2(-inf) = 0 for a wave without keys
      tot = fmaf(sc, wl[w], tot);
      o = fmaf(sc, red[w][tid], o);
    }
    out[(size_t)c * D + h * HD + tid] = o / tot;
  }
  DEC_T(3);
}

// Decode attention for one (candidate, head): q is this step's (fp16-rounded) query, K/V of the new position
// are already in the fp16 cache; softmax(q.K/8) V over n_past+1 keys. 4 waves: keys are spread over all 256
// threads for the scores and over 16 groups for PV.
__global__ __launch_bounds__(256) void attn_decode_kernel(const float *__restrict__ qbuf, const __half *__restrict__ kc,
                                                          const __half *__restrict__ vc, const StepState *__restrict__ ss,
                                                          int max_pos, float *__restrict__ out, int lut) {
  __shared__ float sc[1024];
  __shared__ float qs[HD];
  __shared__ float red[16 * HD];
  __shared__ float wred[8];
  const int c = blockIdx.x, h = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nk = ss->n_past + 1;
  const __half *kb = kc + (size_t)c * max_pos * D + h * HD;
  const __half *vb = vc + (size_t)c * max_pos * D + h * HD;
  if (tid < HD) qs[tid] = qbuf[(size_t)c * D + h * HD + tid];
  // V rows of the first PV batch do not depend on the scores: request them now, they arrive under the score phase
  const int dl = tid & 15, grp = tid >> 4;
  uint2 vv0[8];
#pragma unroll
  for (int u = 0; u < 8; u++) {
    const int j = grp + u * 16;
    vv0[u] = (j < nk) ? *(const uint2 *)(vb + (size_t)j * D + dl * 4) : make_uint2(0u, 0u);
  }
  __syncthreads();
  float mx = -INFINITY;
  for (int j = tid; j < nk; j += 256) {
    const uint4 *kp = (const uint4 *)(kb + (size_t)j * D);
    uint4 u[8];
#pragma unroll
    for (int q = 0; q < 8; q++) u[q] = kp[q];
    float dot = 0.f;
#pragma unroll
    for (int q = 0; q < 8; q++) {
      const __half2 *h2 = (const __half2 *)&u[q];
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const float2 f = __half22float2(h2[e]);
        dot = fmaf(qs[q * 8 + e * 2], f.x, dot);
        dot = fmaf(qs[q * 8 + e * 2 + 1], f.y, dot);
      }
    }
    dot *= 0.125f;
    sc[j] = dot;
    mx = fmaxf(mx, dot);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  if (lane == 0) wred[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(wred[0], wred[1]), fmaxf(wred[2], wred[3]));
  float sum = 0.f;
  for (int j = tid; j < nk; j += 256) {
    const float e = lut ? f16_round(expf(f16_round(sc[j] - mx))) : expf(sc[j] - mx);
    sc[j] = e;
    sum += e;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
  if (lane == 0) wred[4 + wave] = sum;
  __syncthreads();
  const float inv = 1.0f / (((wred[4] + wred[5]) + wred[6]) + wred[7]);
  // PV: 16 key groups x 16 lanes; a lane owns 4 consecutive dims (8-byte loads), keys j = grp, grp+16, ...
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  for (int j0 = grp; j0 < nk; j0 += 16 * 8) {
    uint2 vv[8];
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const int j = j0 + u * 16;
      if (j0 == grp) vv[u] = vv0[u];
      else vv[u] = (j < nk) ? *(const uint2 *)(vb + (size_t)j * D + dl * 4) : make_uint2(0u, 0u);
    }
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const int j = j0 + u * 16;
      const float p = (j < nk) ? sc[j] * inv : 0.f;
      const float2 f0 = __half22float2(*(const __half2 *)&vv[u].x), f1 = __half22float2(*(const __half2 *)&vv[u].y);
      a0 = fmaf(p, f0.x, a0); a1 = fmaf(p, f0.y, a1); a2 = fmaf(p, f1.x, a2); a3 = fmaf(p, f1.y, a3);
    }
  }
  red[grp * HD + dl * 4 + 0] = a0; red[grp * HD + dl * 4 + 1] = a1;
  red[grp * HD + dl * 4 + 2] = a2; red[grp * HD + dl * 4 + 3] = a3;
  __syncthreads();
  if (tid < HD) {
    float o = 0.f;
#pragma unroll
    for (int g2 = 0; g2 < 16; g2++) o += red[g2 * HD + tid];
    out[(size_t)c * D + h * HD + tid] = o;
  }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------
// Device top-k prefilter (option "device_topk"): the sampler (process_logits_and_sample, main.cpp:4753-4806) keeps the 50 largest
// penalised logits of 8194. Instead of 16 x 8194 floats per step crossing PCIe, every candidate's row is reduced on the device to the
// logits >= a threshold that keeps TTS_PF_MIN .. TTS_PF_MAX of them, in index order; the host runs the bit-exact float tail over that
// list (host_logic.cpp: sample_one_list, which also states why the list is sufficient and when the full row is fetched instead).
// One workgroup per candidate, 33 logits per thread in registers as order-preserving integer keys; the threshold is found by bisection
// on the key (one ballot-popcount pass over the registers + one barrier per probe, <= 32 probes), then an index-ordered compaction.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned pf_key(float v) {
  unsigned u = __float_as_uint(v);
  if (u == 0x80000000u) u = 0; // -0 and +0 compare equal as floats: one key
  return (u >> 31) ? ~u : (u | 0x80000000u);
}

__global__ __launch_bounds__(256) void sample_prefilter_kernel(const float *__restrict__ logits, int mask_stop, int32_t *__restrict__ out) {
  constexpr int NJ = (V + 255) / 256; // 33
  __shared__ int red[2][4];
  __shared__ int4 tab[NJ];
  const int c = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float *src = logits + (size_t)c * V;
  int32_t *o = out + (size_t)c * TTS_PF_WORDS;
  unsigned key[NJ];
#pragma unroll
  for (int j = 0; j < NJ; j++) {
    const int i = j * 256 + tid;
    float v = i < V ? src[i] : 0.f;
    if (mask_stop && i == V - 1) v = -1e30f; // TTS_AR_MASK_STOP: the stop token 8193 is never sampled
    key[j] = i < V ? pf_key(v) : 0u;          // padding: key 0 is below every probe (probes are > 0)
  }
  unsigned lo = 0u, hi = 0xffffffffu, t = 0u; // count(lo) > PF_MAX, count(hi) < PF_MIN
  int n = -1, it = 0;
  while (hi - lo > 1u) {
    const unsigned mid = lo + ((hi - lo) >> 1);
    int cw = 0;
#pragma unroll
    for (int j = 0; j < NJ; j++) cw += __popcll(__ballot(key[j] >= mid));
    if (lane == 0) red[it & 1][wave] = cw;
    __syncthreads(); // double-buffered: one barrier per probe
    const int cnt = red[it & 1][0] + red[it & 1][1] + red[it & 1][2] + red[it & 1][3];
    it++;
    if (cnt > TTS_PF_MAX) lo = mid;
    else if (cnt < TTS_PF_MIN) hi = mid;
    else { n = cnt; t = mid; break; }
  }
  if (n < 0) { // more than PF_MAX - PF_MIN + 1 logits tie at the 64th place: the host takes the full row
    if (tid == 0) o[0] = -1;
    return;
  }
  const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
  for (int j = 0; j < NJ; j++) {
    const int cw = __popcll(__ballot(key[j] >= t));
    if (lane == 0) ((int *)&tab[j])[wave] = cw;
  }
  __syncthreads();
  int base = 0;
#pragma unroll
  for (int j = 0; j < NJ; j++) {
    const int4 r = tab[j];
    const unsigned long long m = __ballot(key[j] >= t);
    if (key[j] >= t) {
      const int i = j * 256 + tid;
      const int pos = base + (wave > 0 ? r.x : 0) + (wave > 1 ? r.y : 0) + (wave > 2 ? r.z : 0) + __popcll(m & below);
      float v = src[i];
      if (mask_stop && i == V - 1) v = -1e30f;
      o[4 + pos] = i;
      o[4 + TTS_PF_MAX + pos] = __float_as_int(v);
    }
    base += r.x + r.y + r.z + r.w;
  }
  if (tid == 0) { o[0] = n; o[1] = 0; o[2] = 0; o[3] = 0; }
}


struct ArLayerDev {
  float *ln1_g, *ln1_b, *ln2_g, *ln2_b;
  float *w_attn, *b_attn, *w_proj, *b_proj, *w_fc, *b_fc, *w_fc2, *b_fc2;
  __half *s_attn = nullptr, *s_proj = nullptr, *s_fc = nullptr, *s_fc2 = nullptr; // [N][2K] hi|lo of 64*W^T
  float *d_attn = nullptr, *d_fc = nullptr;   // pack_mfma16 of diag(ln gamma) W (decode step)
  __half *dh_attn = nullptr, *dh_fc = nullptr; // pack_mfma16h of the same (split-precision variant)
  float *db_attn = nullptr, *db_fc = nullptr; // bias + ln beta . W
  float *d_proj = nullptr, *d_fc2 = nullptr;  // pack_cols4  (decode step)
  __half *q_attn = nullptr, *q_fc = nullptr, *q_proj = nullptr, *q_fc2 = nullptr; // fp16-weight decode slabs (option ar_weights = 1 at load)
  // fp8-weight decode slabs (option ar_weights = 2 at load): e4m3 bytes in the same packing orders + one power-of-two scale per output column
  uint8_t *o_attn = nullptr, *o_fc = nullptr, *o_proj = nullptr, *o_fc2 = nullptr;
  float *os_attn = nullptr, *os_fc = nullptr, *os_proj = nullptr, *os_fc2 = nullptr;
};

struct ArState {
  int n_layers = 0;
  std::vector<ArLayerDev> L;
  float *text_emb = nullptr, *text_pos = nullptr, *mel_emb = nullptr, *mel_pos = nullptr;
  float *lnf_g = nullptr, *lnf_b = nullptr, *lmh_g = nullptr, *lmh_b = nullptr;
  float *lm_w = nullptr /*[1024][VPAD] strip-major*/, *lm_b = nullptr /*[VPAD]*/, *d_lm = nullptr /*pack_mfma16, lm_head.0 folded*/, *d_lmb = nullptr;
  __half *dh_lm = nullptr; // pack_mfma16h
  __half *q_lm = nullptr;  // pack_mfma16q (option ar_weights = 1 at load)
  uint8_t *o_lm = nullptr; // pack_mfma16o (option ar_weights = 2 at load)
  float *os_lm = nullptr;
  int loaded_wmode = 0;    // the reduced-precision decode slabs this state was loaded with (0 = none, 1 = fp16, 2 = fp8)
  bool f32_mfma = false;   // option "dec_f32_mfma" at load: LayerNorm-GEMVs on v_mfma_f32_16x16x4_f32 (exact f32 products) instead of split fp16
  std::vector<void *> owned;
  // run state
  int B = 0, n_text = 0, P = 0, max_pos = 0;
  bool prefill_done = false; // the prompt's K/V rows are in the decode cache (set by ar_prefill, cleared by ar_begin)
  std::vector<int> tokens;
  DevBuf voice, kcache, vcache, lat_k, lat_v;
  DevBuf h, xn, qkv, att, ff, part, desc, logits, hn, a_hi, a_lo;
  // decode-step graph
  DevBuf d_toks;
  int32_t *h_toks = nullptr;   // pinned
  float *h_logits = nullptr;   // pinned [B][8194]
  int32_t *h_pf = nullptr;     // pinned [B][TTS_PF_WORDS]: the device prefilter's lists (step_mode != 0), written by the kernel itself
  int step_mode = 0;           // what the step hands to the host: 0 = the logits, 1 = the prefilter's lists, 2 = lists with the stop token masked
  int h_cap_B = 0;             // candidates the pinned buffers were sized for
  // one captured step per mode (a caller alternating tts_ar_step / tts_ar_step_sample, or bench's device_topk on/off loop, re-instantiated
  // the ~150-node graph on every switch)
  hipGraph_t graphs[3] = {};
  hipGraphExec_t graph_execs[3] = {};
  // everything the captured step bakes into its nodes: the graph of the previous utterance is replayed when nothing moved
  struct GraphSig {
    int B = 0, max_pos = 0, lut = 0, wmode = 0, mode = 0;
    const void *p[11] = {};
    bool operator==(const GraphSig &o) const {
      return B == o.B && max_pos == o.max_pos && lut == o.lut && wmode == o.wmode && mode == o.mode && std::equal(p, p + 11, o.p);
    }
  } graph_sigs[3];
  GraphSig current_sig(int lut, int wmode) const {
    GraphSig g;
    g.B = B; g.max_pos = max_pos; g.lut = lut; g.wmode = wmode; g.mode = step_mode;
    const void *q[11] = {h.p, qkv.p, att.p, ff.p, kcache.p, vcache.p, d_toks.p, logits.p, h_toks, h_logits, h_pf};
    std::copy(q, q + 11, g.p);
    return g;
  }
  void drop_graph(int m) {
    if (graph_execs[m]) (void)hipGraphExecDestroy(graph_execs[m]);
    if (graphs[m]) (void)hipGraphDestroy(graphs[m]);
    graph_execs[m] = nullptr; graphs[m] = nullptr;
  }
  ~ArState() {
    for (int m = 0; m < 3; m++) drop_graph(m);
    if (h_toks) (void)hipHostFree(h_toks);
    if (h_logits) (void)hipHostFree(h_logits);
    if (h_pf) (void)hipHostFree(h_pf);
    for (void *p : owned) (void)hipFree(p);
  }
};

void ar_free(ArState *s) { delete s; }

static PinnedPool *ar_pin = nullptr; // pinned staging of the running ar_load (one load at a time per process: the loaders are not re-entrant across contexts)
static hipStream_t ar_load_stream = nullptr;
static hipError_t ar_h2d(void *dst, const void *src, size_t bytes) { return ar_pin ? ar_pin->upload(dst, src, bytes, ar_load_stream) : PinnedPool::copy_now(dst, src, bytes, ar_load_stream); }
// a tensor that read_weight_file left in the file: pread() into pinned staging, DMA to dst
static int ar_file_to_device(tts_ctx *ctx, const WeightFile &wf, const HostTensor &t, const std::string &name, void *dst) {
  const size_t bytes = (size_t)t.nelem() * 4;
  std::pair<void *, size_t> b = ar_pin ? ar_pin->take(bytes) : std::pair<void *, size_t>{nullptr, 0};
  std::vector<float> tmp;
  void *host = b.first;
  if (!host) { tmp.resize((size_t)t.nelem()); host = tmp.data(); }
  const bool ok = wf.read_payload(t, host);
  const hipError_t e = ok ? PinnedPool::copy_now(dst, host, bytes, ar_load_stream) : hipSuccess;
  if (ar_pin) ar_pin->give(b);
  if (!ok) return fail(ctx, TTS_ERR_IO, "autoregressive_model_load: tensor '%s' truncated", name.c_str());
  TTS_HIP(ctx, e);
  return TTS_OK;
}
static std::mutex ar_own_mu; // ar_load builds the layers on several threads (common.h: run_parallel)
static void ar_own(ArState *st, void *p) {
  std::lock_guard<std::mutex> lk(ar_own_mu);
  st->owned.push_back(p);
}
static int upload_h(tts_ctx *ctx, ArState *st, const std::vector<__half> &src, __half **dst) {
  void *p = nullptr;
  TTS_HIP(ctx, hipMalloc(&p, src.size() * sizeof(__half)));
  ar_own(st, p);
  TTS_HIP(ctx, ar_h2d(p, src.data(), src.size() * sizeof(__half)));
  *dst = (__half *)p;
  return TTS_OK;
}

static int upload(tts_ctx *ctx, ArState *st, const std::vector<float> &src, float **dst) {
  void *p = nullptr;
  TTS_HIP(ctx, hipMalloc(&p, src.size() * sizeof(float)));
  ar_own(st, p);
  TTS_HIP(ctx, ar_h2d(p, src.data(), src.size() * sizeof(float)));
  *dst = (float *)p;
  return TTS_OK;
}

// Weight matrices [K][N] (N contiguous) are re-tiled at load into 64-column strips, [N/64][K][64], so that the
// K-chunk a GEMV workgroup streams is ONE contiguous region and a wave's load instruction covers 1 KB of
// consecutive bytes (the reference re-transposes every matrix in every graph execution; here the layout
// is chosen once).
static std::vector<float> strip_major(const float *w, int K, int N) {
  std::vector<float> t((size_t)K * N);
  for (int s0 = 0; s0 < N / 64; s0++)
    for (int k = 0; k < K; k++) memcpy(&t[((size_t)s0 * K + k) * 64], &w[(size_t)k * N + s0 * 64], 64 * sizeof(float));
  return t;
}

// Decode-step layouts (w is [K][N], N contiguous).
// pack_mfma16 (K = 1024): slab of workgroup cb (16 columns) = 4 waves x 16 groups x 64 lanes x float4, where lane
// (m = lane & 15, q = lane >> 4) of wave wv holds W[256 wv + 16 g + 4 q + j][16 cb + m], j = 0..3, for group g.
static std::vector<float> pack_mfma16(const float *w, int K, int N) {
  std::vector<float> t((size_t)K * N);
  for (int cb = 0; cb < N / 16; cb++)
    for (int wv = 0; wv < 4; wv++)
      for (int g = 0; g < 16; g++)
        for (int lane = 0; lane < 64; lane++)
          for (int j = 0; j < 4; j++)
            t[((((size_t)cb * 4 + wv) * 16 + g) * 64 + lane) * 4 + j] =
                w[(size_t)(wv * 256 + g * 16 + 4 * (lane >> 4) + j) * N + cb * 16 + (lane & 15)];
  return t;
}
// pack_mfma16h (K = 1024): the same slab for the split-precision fp16 MFMA. Lane (m, q) of wave wv holds for K step s
// (32 k) the 8 values k = 256 wv + 32 s + 8 q + e of column 16 cb + m, times 64, as 8 fp16 hi followed by 8 fp16 lo
// (hi = fp16(64 w), lo = fp16(64 w - hi)): 32 contiguous bytes per lane, 2 KB per wave and step.
static std::vector<__half> pack_mfma16h(const float *w, int K, int N) {
  std::vector<__half> t((size_t)K * N * 2);
  for (int cb = 0; cb < N / 16; cb++)
    for (int wv = 0; wv < 4; wv++)
      for (int s2 = 0; s2 < 8; s2++)
        for (int lane = 0; lane < 64; lane++)
          for (int e = 0; e < 8; e++) {
            const float v = 64.0f * w[(size_t)(wv * 256 + s2 * 32 + 8 * (lane >> 4) + e) * N + cb * 16 + (lane & 15)];
            const __half hi = __float2half_rn(v);
            const size_t base = ((((size_t)cb * 4 + wv) * 8 + s2) * 64 + lane) * 16;
            t[base + e] = hi;
            t[base + 8 + e] = __float2half_rn(v - __half2float(hi));
          }
  return t;
}
// pack_mfma16q: pack_mfma16h without the lo halves — fp16(64 w), 16 contiguous bytes per lane and K step (option ar_weights = 1).
static std::vector<__half> pack_mfma16q(const float *w, int K, int N) {
  std::vector<__half> t((size_t)K * N);
  for (int cb = 0; cb < N / 16; cb++)
    for (int wv = 0; wv < 4; wv++)
      for (int s2 = 0; s2 < 8; s2++)
        for (int lane = 0; lane < 64; lane++)
          for (int e = 0; e < 8; e++)
            t[((((size_t)cb * 4 + wv) * 8 + s2) * 64 + lane) * 8 + e] =
                __float2half_rn(64.0f * w[(size_t)(wv * 256 + s2 * 32 + 8 * (lane >> 4) + e) * N + cb * 16 + (lane & 15)]);
  return t;
}
// ---- OCP fp8 e4m3 (1-4-3, bias 7, largest finite 448, no infinities): round to nearest even, saturating ----
static uint8_t fp8_e4m3_encode(float f) {
  const uint8_t sign = std::signbit(f) ? 0x80 : 0;
  float a = std::fabs(f);
  if (!(a == a)) return sign | 0x7f;
  if (a >= 448.0f) return sign | 0x7e;
  if (a < 0.0009765625f) return sign; // below half of the smallest subnormal (2^-9 / 2): rounds to zero (ties-to-even at exactly 2^-10 -> 0)
  int e;
  (void)std::frexp(a, &e);           // a = m 2^e, m in [0.5, 1)
  int ex = e - 1;                    // a = 1.xxx 2^ex
  if (ex < -6) ex = -6;              // subnormals share the exponent of the smallest normal
  const float q = std::ldexp(1.0f, ex - 3); // spacing of representable values in this binade
  float r = std::nearbyint(a / q) * q;      // default rounding mode: to nearest even
  if (r >= 448.0f) return sign | 0x7e;
  if (r < 0.015625f) return sign | (uint8_t)std::lrint(r / 0.001953125f); // subnormal: mantissa = r / 2^-9
  (void)std::frexp(r, &e);
  ex = e - 1;
  const int mant = (int)std::lrint(r / std::ldexp(1.0f, ex - 3)) - 8;
  return sign | (uint8_t)(((ex + 7) << 3) | mant);
}
extern "C" uint8_t tts_host_fp8_e4m3(float v) { return fp8_e4m3_encode(v); }
// one power-of-two scale per output column, the smallest with max |w| / scale <= 448 (exact scaling: a matrix whose entries are
// e4m3 values times a power of two per column survives the round trip unchanged)
static std::vector<float> fp8_col_scales(const float *w, int K, int N) {
  std::vector<float> sc(N, 1.0f);
  for (int n = 0; n < N; n++) {
    float amax = 0.f;
    for (int k = 0; k < K; k++) amax = std::max(amax, std::fabs(w[(size_t)k * N + n]));
    if (amax > 0.f) {
      int e;
      (void)std::frexp(amax / 448.0f, &e); // amax / 448 = m 2^e, m in [0.5, 1)  ->  2^e >= amax / 448
      sc[n] = std::ldexp(1.0f, amax / 448.0f == std::ldexp(0.5f, e) ? e - 1 : e);
    }
  }
  return sc;
}
// pack_mfma16o: pack_mfma16q's order with one BYTE per value (e4m3 of w / scale[column]); a lane's K steps 2 i and 2 i + 1 share one
// 16-byte load: byte ((((cb 4 + wv) 4 + i) 64 + lane) 2 + (s & 1)) 8 + e
static std::vector<uint8_t> pack_mfma16o(const float *w, int K, int N, const std::vector<float> &sc) {
  std::vector<uint8_t> t((size_t)K * N);
  for (int cb = 0; cb < N / 16; cb++)
    for (int wv = 0; wv < 4; wv++)
      for (int s2 = 0; s2 < 8; s2++)
        for (int lane = 0; lane < 64; lane++)
          for (int e = 0; e < 8; e++) {
            const int n = cb * 16 + (lane & 15);
            t[(((((size_t)cb * 4 + wv) * 4 + (s2 >> 1)) * 64 + lane) * 2 + (s2 & 1)) * 8 + e] =
                fp8_e4m3_encode(w[(size_t)(wv * 256 + s2 * 32 + 8 * (lane >> 4) + e) * N + n] / sc[n]);
          }
  return t;
}
// pack_cols4o: pack_cols4's order with one byte per value: 4 bytes (the 4 columns) per (k, workgroup)
static std::vector<uint8_t> pack_cols4o(const float *w, int K, int N, const std::vector<float> &sc) {
  std::vector<uint8_t> t((size_t)K * N);
  const int KG = K / 1024;
  for (int cb = 0; cb < N / 4; cb++)
    for (int i = 0; i < KG; i++)
      for (int kk = 0; kk < 4; kk++)
        for (int tid = 0; tid < 256; tid++)
          for (int c = 0; c < 4; c++)
            t[((((size_t)cb * KG + i) * 4 + kk) * 256 + tid) * 4 + c] =
                fp8_e4m3_encode(w[(size_t)(i * 1024 + 4 * tid + kk) * N + cb * 4 + c] / sc[cb * 4 + c]);
  return t;
}
static std::vector<__half> to_half(const std::vector<float> &v) {
  std::vector<__half> t(v.size());
  for (size_t i = 0; i < v.size(); i++) t[i] = __float2half_rn(v[i]);
  return t;
}
// pack_cols4 (K = 1024 KG): slab of workgroup cb (4 columns) = KG x 4 x 256 threads x float4 (the 4 columns),
// thread t holding k = 1024 i + 4 t + kk.
static std::vector<float> pack_cols4(const float *w, int K, int N) {
  std::vector<float> t((size_t)K * N);
  const int KG = K / 1024;
  for (int cb = 0; cb < N / 4; cb++)
    for (int i = 0; i < KG; i++)
      for (int kk = 0; kk < 4; kk++)
        for (int tid = 0; tid < 256; tid++)
          memcpy(&t[((((size_t)cb * KG + i) * 4 + kk) * 256 + tid) * 4], &w[(size_t)(i * 1024 + 4 * tid + kk) * N + cb * 4], 16);
  return t;
}

// LayerNorm affine folded into the matrix it feeds: (xhat*g + b) W + c = xhat (diag(g) W) + (b W + c).
static void fold_layernorm(const float *w, int K, int N, const float *g, const float *b, const float *c,
                           std::vector<float> &wf, std::vector<float> &cf) {
  wf.resize((size_t)K * N);
  std::vector<double> acc(N, 0.0);
  for (int k = 0; k < K; k++)
    for (int n = 0; n < N; n++) {
      wf[(size_t)k * N + n] = g[k] * w[(size_t)k * N + n];
      acc[n] += (double)b[k] * (double)w[(size_t)k * N + n];
    }
  cf.resize(N);
  for (int n = 0; n < N; n++) cf[n] = (float)((double)c[n] + acc[n]);
}

static int fetch(tts_ctx *ctx, ArState *st, const WeightFile &wf, const std::string &name, int64_t ne0, int64_t ne1,
                 float **dst, bool tile = false) {
  auto it = wf.t.find(name);
  if (it == wf.t.end()) return fail(ctx, TTS_ERR_FORMAT, "tensor '%s' missing from AR model file", name.c_str());
  const HostTensor &t = it->second;
  if (t.ne[0] != ne0 || t.ne[1] != ne1 || t.nelem() != ne0 * ne1)
    return fail(ctx, TTS_ERR_FORMAT, "tensor '%s' has wrong shape in model file: got [%d, %d], expected [%d, %d]",
                name.c_str(), (int)t.ne[0], (int)t.ne[1], (int)ne0, (int)ne1);
  if (t.data.empty() && t.file_off >= 0) { // left in the file (device-packing load): file -> pinned staging -> device
    if (tile) return fail(ctx, TTS_ERR_STATE, "internal: strip-major re-layout of a tensor that was not read ('%s')", name.c_str());
    void *p = nullptr;
    TTS_HIP(ctx, hipMalloc(&p, (size_t)t.nelem() * 4));
    ar_own(st, p);
    const int r = ar_file_to_device(ctx, wf, t, name, p);
    if (r) return r;
    *dst = (float *)p;
    return TTS_OK;
  }
  if (tile) return upload(ctx, st, strip_major(t.data.data(), (int)ne1, (int)ne0), dst);
  return upload(ctx, st, t.data, dst);
}

int ar_load(tts_ctx *ctx, const char *path) {
  static const bool timing = getenv("TTS_TIMING") != nullptr; // host-side breakdown on stderr
  const auto t0 = std::chrono::steady_clock::now();
  auto since = [&](std::chrono::steady_clock::time_point a) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count(); };
  // (decided before the file is read: with device-side packing the large tensors stay in the file until a worker wants them)
  const bool dev_pack = ctx->load_device_pack != 0 && ctx->load_threads != 1 && ctx->dec_f32_mfma == 0 && ctx->ar_weights == 0;
  WeightFile wf;
  std::string err;
  int rc = read_weight_file(path, wf, err, dev_pack ? (size_t)256 << 10 : (size_t)-1);
  const double t_read = since(t0);
  if (rc != TTS_OK) return fail(ctx, rc, "autoregressive_model_load: %s", err.c_str());
  static std::mutex load_mu; // one tts_load_ar at a time per process (ar_pin is shared)
  std::lock_guard<std::mutex> load_lk(load_mu);
  std::unique_ptr<ArState> st(new ArState());
  PinnedPool pin{(size_t)18 << 20, 6};
  struct PinScope { PinnedPool *&slot; PinScope(PinnedPool *&s, PinnedPool *p) : slot(s) { slot = p; } ~PinScope() { slot = nullptr; } } pin_scope(ar_pin, ctx->load_threads == 1 ? nullptr : &pin);
  ar_load_stream = ctx->load_stream;
  st->f32_mfma = ctx->dec_f32_mfma != 0;
  const std::string hp = "inference_model.transformer.h.";
  while (wf.has(hp + std::to_string(st->n_layers) + ".ln_1.weight")) st->n_layers++;
  if (st->n_layers == 0) return fail(ctx, TTS_ERR_FORMAT, "no transformer layers in '%s'", path);
  // every tensor in the file must be known (main.cpp:834-838)
  for (auto &kv : wf.t) {
    const std::string &n = kv.first;
    bool ok = n == "text_embedding.weight" || n == "text_pos_embedding.emb.weight" || n == "mel_embedding.weight" ||
              n == "mel_pos_embedding.emb.weight" || n.rfind("inference_model.", 0) == 0;
    if (!ok) return fail(ctx, TTS_ERR_FORMAT, "unknown tensor '%s' in model file", n.c_str());
  }
#define FETCH(name, a, b, dst) do { int _r = fetch(ctx, st.get(), wf, name, a, b, dst); if (_r) return _r; } while (0)
#define FETCHT(name, a, b, dst) do { int _r = fetch(ctx, st.get(), wf, name, a, b, dst, true); if (_r) return _r; } while (0)
  // Device-side re-layouts (option load_device_pack, default 1) for the default decode arithmetic; the reduced-precision slab options and the f32-MFMA variant keep the
  // host packers (they are A/B options), and so does load_threads = 1 (the serial loader of rounds 1-5, kept as the reference for the equality test).
  std::mutex tmp_mu;
  std::vector<void *> temps; // file-order copies and folded matrices: freed once the packing kernels have run
  // temporaries come out of ONE device allocation (a few hundred hipMalloc / hipFree pairs cost 0.1 s): the file's tensors once more + the gain-folded matrices + the head
  char *arena = nullptr;
  size_t arena_cap = 0;
  std::atomic<size_t> arena_at{0};
  if (dev_pack) {
    size_t need = (size_t)st->n_layers * ((size_t)D * 3 * D + (size_t)D * FF) * 4 + (size_t)3 * D * VPAD * 4 + ((size_t)8 << 20);
    for (auto &kv : wf.t) need += (size_t)kv.second.nelem() * 4 + 256;
    if (hipMalloc((void **)&arena, need) == hipSuccess) arena_cap = need;
    else (void)hipGetLastError(); // fall back to one allocation per temporary
  }
  struct ArenaFree { char *&p; ~ArenaFree() { if (p) (void)hipFree(p); } } arena_free{arena};
  // ... and so do the layouts that stay (4.4 GB in ~400 pieces: one hipMalloc now, one hipFree in tts_destroy)
  char *keep = nullptr;
  size_t keep_cap = 0;
  std::atomic<size_t> keep_at{0};
  if (dev_pack) {
    const size_t per_layer = ((size_t)12 * D * D) * 4 /* strip-major f32 */ + ((size_t)7 * D * D) * 4 /* split-fp16 decode slabs */ + ((size_t)5 * D * D) * 4 /* 4-column slabs */ +
                             ((size_t)12 * D * D) * 4 /* hi | lo copies for the multi-row passes */ + (size_t)(3 * D + FF) * 4 + 64 * 256;
    const size_t need = (size_t)st->n_layers * per_layer + (size_t)D * VPAD * 8 + (size_t)VPAD * 4 + ((size_t)1 << 20);
    if (hipMalloc((void **)&keep, need) == hipSuccess) { keep_cap = need; ar_own(st.get(), keep); }
    else (void)hipGetLastError();
  }
  auto dalloc = [&](size_t bytes, bool temp) -> void * {
    if (temp && arena) {
      const size_t sz = (bytes + 255) & ~(size_t)255, at = arena_at.fetch_add(sz);
      if (at + sz <= arena_cap) return arena + at;
    }
    if (!temp && keep) {
      const size_t sz = (bytes + 255) & ~(size_t)255, at = keep_at.fetch_add(sz);
      if (at + sz <= keep_cap) return keep + at;
    }
    void *q = nullptr;
    if (hipMalloc(&q, bytes) != hipSuccess) { (void)hipGetLastError(); fail(ctx, TTS_ERR_HIP, "hipMalloc of %zu bytes failed while loading the AR model", bytes); return nullptr; }
    if (temp) { std::lock_guard<std::mutex> lk(tmp_mu); temps.push_back(q); }
    else ar_own(st.get(), q);
    return q;
  };
  struct TempFree { std::vector<void *> &v; ~TempFree() { for (void *q : v) (void)hipFree(q); } } temp_free{temps};
  auto pgrid = [](size_t n) { return (int)std::min<size_t>((n + 255) / 256, 8192); };
  // the tensor as it lies in the file, shape-checked like fetch(), into a temporary device buffer
  auto raw_up = [&](const std::string &name, int64_t ne0, int64_t ne1, float **dst) -> int {
    auto it = wf.t.find(name);
    if (it == wf.t.end()) return fail(ctx, TTS_ERR_FORMAT, "tensor '%s' missing from AR model file", name.c_str());
    const HostTensor &t = it->second;
    if (t.ne[0] != ne0 || t.ne[1] != ne1 || t.nelem() != ne0 * ne1)
      return fail(ctx, TTS_ERR_FORMAT, "tensor '%s' has wrong shape in model file: got [%d, %d], expected [%d, %d]", name.c_str(), (int)t.ne[0], (int)t.ne[1], (int)ne0, (int)ne1);
    void *q = dalloc((size_t)t.nelem() * 4, true);
    if (!q) return TTS_ERR_HIP;
    if (t.data.empty() && t.file_off >= 0) { const int r = ar_file_to_device(ctx, wf, t, name, q); if (r) return r; }
    else TTS_HIP(ctx, ar_h2d(q, t.data.data(), (size_t)t.nelem() * 4));
    *dst = (float *)q;
    return TTS_OK;
  };
  unsigned *d_max = nullptr; // max |w| bits: per layer the four matrices, then the two gain-folded ones
  if (dev_pack) {
    d_max = (unsigned *)dalloc((size_t)st->n_layers * 6 * 4, true);
    if (!d_max) return TTS_ERR_HIP;
    TTS_HIP(ctx, hipMemsetAsync(d_max, 0, (size_t)st->n_layers * 6 * 4, ctx->stream));
  }
  auto build_globals = [&]() -> int {
    FETCH("text_embedding.weight", D, 256, &st->text_emb);
    FETCH("text_pos_embedding.emb.weight", D, 404, &st->text_pos);
    FETCH("mel_embedding.weight", D, V, &st->mel_emb);
    FETCH("mel_pos_embedding.emb.weight", D, 608, &st->mel_pos);
    FETCH("inference_model.transformer.ln_f.weight", D, 1, &st->lnf_g);
    FETCH("inference_model.transformer.ln_f.bias", D, 1, &st->lnf_b);
    FETCH("inference_model.lm_head.0.weight", D, 1, &st->lmh_g);
    FETCH("inference_model.lm_head.0.bias", D, 1, &st->lmh_b);
    return TTS_OK;
  };
  st->L.resize(st->n_layers);
  // One layer = 50 MB of f32 weights re-tiled into the decode slabs, the strip-major copies and the split-fp16 MFMA layouts on the host (~0.2 s of one core) + their
  // uploads: layers are independent, built on several threads (round 6: tts_load_ar 5.9 s -> see DESIGN.md section 5; option load_threads = 1 restores the serial loader).
  auto build_layer = [&](int i) -> int {
    std::string p = hp + std::to_string(i);
    ArLayerDev &l = st->L[i];
    FETCH(p + ".ln_1.weight", D, 1, &l.ln1_g); FETCH(p + ".ln_1.bias", D, 1, &l.ln1_b);
    FETCH(p + ".ln_2.weight", D, 1, &l.ln2_g); FETCH(p + ".ln_2.bias", D, 1, &l.ln2_b);
    FETCH(p + ".attn.c_attn.bias", 3 * D, 1, &l.b_attn); FETCH(p + ".attn.c_proj.bias", D, 1, &l.b_proj);
    FETCH(p + ".mlp.c_fc.bias", FF, 1, &l.b_fc); FETCH(p + ".mlp.c_proj.bias", D, 1, &l.b_fc2);
    int r;
    if (dev_pack) {
      struct M { const char *name; int K, N; float **strip; float *raw; } m[4] = {
          {".attn.c_attn.weight", D, 3 * D, &l.w_attn, nullptr}, {".attn.c_proj.weight", D, D, &l.w_proj, nullptr},
          {".mlp.c_fc.weight", D, FF, &l.w_fc, nullptr}, {".mlp.c_proj.weight", FF, D, &l.w_fc2, nullptr}};
      for (int j = 0; j < 4; j++) {
        if ((r = raw_up(p + m[j].name, m[j].N, m[j].K, &m[j].raw))) return r;
        const size_t n = (size_t)m[j].K * m[j].N;
        if (!(*m[j].strip = (float *)dalloc(n * 4, false))) return TTS_ERR_HIP;
        pk_strip_major_kernel<<<pgrid(n), 256, 0, ctx->stream>>>(m[j].raw, m[j].K, m[j].N, *m[j].strip);
        pk_absmax_kernel<<<std::min(pgrid(n), 1024), 256, 0, ctx->stream>>>(m[j].raw, n, d_max + (size_t)i * 6 + j);
      }
      struct F { int j; const float *g, *b, *c; __half **dh; float **db; } f[2] = {{0, l.ln1_g, l.ln1_b, l.b_attn, &l.dh_attn, &l.db_attn},
                                                                                   {2, l.ln2_g, l.ln2_b, l.b_fc, &l.dh_fc, &l.db_fc}};
      for (int q = 0; q < 2; q++) { // LayerNorm gain folded into the matrix the normalised rows feed, then the split-fp16 decode slabs; beta . W + c as the new bias
        const int N = m[f[q].j].N;
        const size_t n = (size_t)D * N;
        float *fold = (float *)dalloc(n * 4, true);
        if (!fold || !(*f[q].dh = (__half *)dalloc(n * 2 * sizeof(__half), false)) || !(*f[q].db = (float *)dalloc((size_t)N * 4, false))) return TTS_ERR_HIP;
        pk_fold_kernel<<<pgrid(n), 256, 0, ctx->stream>>>(m[f[q].j].raw, D, N, f[q].g, fold);
        pk_absmax_kernel<<<std::min(pgrid(n), 1024), 256, 0, ctx->stream>>>(fold, n, d_max + (size_t)i * 6 + 4 + q);
        pk_mfma16h_kernel<<<pgrid(n), 256, 0, ctx->stream>>>(fold, N, *f[q].dh);
        pk_fold_bias_kernel<<<(N + 255) / 256, 256, 0, ctx->stream>>>(m[f[q].j].raw, D, N, f[q].b, f[q].c, *f[q].db);
      }
      if (!(l.d_proj = (float *)dalloc((size_t)D * D * 4, false)) || !(l.d_fc2 = (float *)dalloc((size_t)FF * D * 4, false))) return TTS_ERR_HIP;
      pk_cols4_kernel<<<pgrid((size_t)D * D), 256, 0, ctx->stream>>>(m[1].raw, D, D, l.d_proj);
      pk_cols4_kernel<<<pgrid((size_t)FF * D), 256, 0, ctx->stream>>>(m[3].raw, FF, D, l.d_fc2);
      TTS_HIP(ctx, hipGetLastError());
      return TTS_OK;
    }
    FETCHT(p + ".attn.c_attn.weight", 3 * D, D, &l.w_attn);
    FETCHT(p + ".attn.c_proj.weight", D, D, &l.w_proj);
    FETCHT(p + ".mlp.c_fc.weight", FF, D, &l.w_fc);
    FETCHT(p + ".mlp.c_proj.weight", D, FF, &l.w_fc2);
    // The split-precision layouts hold 64 W as fp16 hi | lo (W16_SCALE): a trained GPT-2 never comes near |W| = 937, but a file that does must fail loudly instead of
    // turning into an fp16 infinity inside the MFMA operands (round 6; the diffusion stage's proj_out pair picks its scale per tensor instead)
    auto split_range = [&](const std::string &name, const float *w, size_t n) -> int {
      float m = 0.f;
      for (size_t i = 0; i < n; i++) m = std::max(m, std::fabs(w[i]));
      if (!(m * W16_SCALE < 60000.0f)) return fail(ctx, TTS_ERR_FORMAT, "tensor '%s': max |w| = %g does not fit the split-precision fp16 layout (|w| < %g)", name.c_str(), m, 60000.0 / W16_SCALE);
      return TTS_OK;
    };
    for (const char *wn : {".attn.c_attn.weight", ".attn.c_proj.weight", ".mlp.c_fc.weight", ".mlp.c_proj.weight"})
      if ((r = split_range(p + wn, wf.t.at(p + wn).data.data(), wf.t.at(p + wn).data.size()))) return r;
    std::vector<float> wfold, cfold;
    fold_layernorm(wf.t.at(p + ".attn.c_attn.weight").data.data(), D, 3 * D, wf.t.at(p + ".ln_1.weight").data.data(),
                   wf.t.at(p + ".ln_1.bias").data.data(), wf.t.at(p + ".attn.c_attn.bias").data.data(), wfold, cfold);
    if ((r = split_range(p + ".attn.c_attn.weight (LayerNorm gain folded in)", wfold.data(), wfold.size()))) return r;
    if (st->f32_mfma && (r = upload(ctx, st.get(), pack_mfma16(wfold.data(), D, 3 * D), &l.d_attn))) return r;
    if ((r = upload_h(ctx, st.get(), pack_mfma16h(wfold.data(), D, 3 * D), &l.dh_attn))) return r;
    if ((r = upload(ctx, st.get(), cfold, &l.db_attn))) return r;
    fold_layernorm(wf.t.at(p + ".mlp.c_fc.weight").data.data(), D, FF, wf.t.at(p + ".ln_2.weight").data.data(),
                   wf.t.at(p + ".ln_2.bias").data.data(), wf.t.at(p + ".mlp.c_fc.bias").data.data(), wfold, cfold);
    if ((r = split_range(p + ".mlp.c_fc.weight (LayerNorm gain folded in)", wfold.data(), wfold.size()))) return r;
    if (st->f32_mfma && (r = upload(ctx, st.get(), pack_mfma16(wfold.data(), D, FF), &l.d_fc))) return r;
    if ((r = upload_h(ctx, st.get(), pack_mfma16h(wfold.data(), D, FF), &l.dh_fc))) return r;
    if ((r = upload(ctx, st.get(), cfold, &l.db_fc))) return r;
    if ((r = upload(ctx, st.get(), pack_cols4(wf.t.at(p + ".attn.c_proj.weight").data.data(), D, D), &l.d_proj))) return r;
    if ((r = upload(ctx, st.get(), pack_cols4(wf.t.at(p + ".mlp.c_proj.weight").data.data(), FF, D), &l.d_fc2))) return r;
    if (ctx->ar_weights == 2) { // fp8-weight decode slabs
      auto up8 = [&](const std::vector<uint8_t> &src, uint8_t **dst) {
        void *q = nullptr;
        TTS_HIP(ctx, hipMalloc(&q, src.size()));
        ar_own(st.get(), q);
        TTS_HIP(ctx, ar_h2d(q, src.data(), src.size()));
        *dst = (uint8_t *)q;
        return (int)TTS_OK;
      };
      std::vector<float> sc;
      fold_layernorm(wf.t.at(p + ".attn.c_attn.weight").data.data(), D, 3 * D, wf.t.at(p + ".ln_1.weight").data.data(),
                     wf.t.at(p + ".ln_1.bias").data.data(), wf.t.at(p + ".attn.c_attn.bias").data.data(), wfold, cfold);
      sc = fp8_col_scales(wfold.data(), D, 3 * D);
      if ((r = up8(pack_mfma16o(wfold.data(), D, 3 * D, sc), &l.o_attn)) || (r = upload(ctx, st.get(), sc, &l.os_attn))) return r;
      fold_layernorm(wf.t.at(p + ".mlp.c_fc.weight").data.data(), D, FF, wf.t.at(p + ".ln_2.weight").data.data(),
                     wf.t.at(p + ".ln_2.bias").data.data(), wf.t.at(p + ".mlp.c_fc.bias").data.data(), wfold, cfold);
      sc = fp8_col_scales(wfold.data(), D, FF);
      if ((r = up8(pack_mfma16o(wfold.data(), D, FF, sc), &l.o_fc)) || (r = upload(ctx, st.get(), sc, &l.os_fc))) return r;
      const float *wp = wf.t.at(p + ".attn.c_proj.weight").data.data(), *w2 = wf.t.at(p + ".mlp.c_proj.weight").data.data();
      sc = fp8_col_scales(wp, D, D);
      if ((r = up8(pack_cols4o(wp, D, D, sc), &l.o_proj)) || (r = upload(ctx, st.get(), sc, &l.os_proj))) return r;
      sc = fp8_col_scales(w2, FF, D);
      if ((r = up8(pack_cols4o(w2, FF, D, sc), &l.o_fc2)) || (r = upload(ctx, st.get(), sc, &l.os_fc2))) return r;
    }
    if (ctx->ar_weights == 1) { // fp16-weight decode slabs (same packing orders)
      fold_layernorm(wf.t.at(p + ".attn.c_attn.weight").data.data(), D, 3 * D, wf.t.at(p + ".ln_1.weight").data.data(),
                     wf.t.at(p + ".ln_1.bias").data.data(), wf.t.at(p + ".attn.c_attn.bias").data.data(), wfold, cfold);
      if ((r = upload_h(ctx, st.get(), pack_mfma16q(wfold.data(), D, 3 * D), &l.q_attn))) return r;
      fold_layernorm(wf.t.at(p + ".mlp.c_fc.weight").data.data(), D, FF, wf.t.at(p + ".ln_2.weight").data.data(),
                     wf.t.at(p + ".ln_2.bias").data.data(), wf.t.at(p + ".mlp.c_fc.bias").data.data(), wfold, cfold);
      if ((r = upload_h(ctx, st.get(), pack_mfma16q(wfold.data(), D, FF), &l.q_fc))) return r;
      if ((r = upload_h(ctx, st.get(), to_half(pack_cols4(wf.t.at(p + ".attn.c_proj.weight").data.data(), D, D)), &l.q_proj))) return r;
      if ((r = upload_h(ctx, st.get(), to_half(pack_cols4(wf.t.at(p + ".mlp.c_proj.weight").data.data(), FF, D)), &l.q_fc2))) return r;
    }
    return TTS_OK;
  };
  // lm_head.1: nn.Linear [8194][1024] -> [1024][VPAD] so that it streams like the Conv1D weights (one more job beside the layers)
  auto build_head = [&]() -> int {
    auto it = wf.t.find("inference_model.lm_head.1.weight");
    auto ib = wf.t.find("inference_model.lm_head.1.bias");
    if (it == wf.t.end() || ib == wf.t.end()) return fail(ctx, TTS_ERR_FORMAT, "lm_head.1 missing from model file");
    if (it->second.ne[0] != D || it->second.ne[1] != V || ib->second.nelem() != V)
      return fail(ctx, TTS_ERR_FORMAT, "tensor 'inference_model.lm_head.1.weight' has wrong shape in model file");
    if (dev_pack) {
      std::vector<float> bt(VPAD, 0.f);
      std::copy(ib->second.data.begin(), ib->second.data.end(), bt.begin());
      int r = upload(ctx, st.get(), bt, &st->lm_b); if (r) return r;
      float *rw = nullptr, *g0 = nullptr, *b0 = nullptr;
      if ((r = raw_up("inference_model.lm_head.1.weight", D, V, &rw)) || (r = raw_up("inference_model.lm_head.0.weight", D, 1, &g0)) ||
          (r = raw_up("inference_model.lm_head.0.bias", D, 1, &b0))) return r;
      const size_t n = (size_t)D * VPAD;
      float *wt = (float *)dalloc(n * 4, true), *fold = (float *)dalloc(n * 4, true);
      if (!wt || !fold || !(st->lm_w = (float *)dalloc(n * 4, false)) || !(st->dh_lm = (__half *)dalloc(n * 2 * sizeof(__half), false)) ||
          !(st->d_lmb = (float *)dalloc((size_t)VPAD * 4, false))) return TTS_ERR_HIP;
      pk_transpose_pad_kernel<<<pgrid(n), 256, 0, ctx->stream>>>(rw, V, D, VPAD, wt);
      pk_strip_major_kernel<<<pgrid(n), 256, 0, ctx->stream>>>(wt, D, VPAD, st->lm_w);
      pk_fold_kernel<<<pgrid(n), 256, 0, ctx->stream>>>(wt, D, VPAD, g0, fold);
      pk_mfma16h_kernel<<<pgrid(n), 256, 0, ctx->stream>>>(fold, VPAD, st->dh_lm);
      pk_fold_bias_kernel<<<(VPAD + 255) / 256, 256, 0, ctx->stream>>>(wt, D, VPAD, b0, st->lm_b, st->d_lmb);
      TTS_HIP(ctx, hipGetLastError());
      return TTS_OK;
    }
    std::vector<float> wt((size_t)D * VPAD, 0.f), bt(VPAD, 0.f);
    const float *w = it->second.data.data();
    for (int n = 0; n < V; n++)
      for (int k = 0; k < D; k++) wt[(size_t)k * VPAD + n] = w[(size_t)n * D + k];
    std::copy(ib->second.data.begin(), ib->second.data.end(), bt.begin());
    int r = upload(ctx, st.get(), strip_major(wt.data(), D, VPAD), &st->lm_w); if (r) return r;
    { // decode head: lm_head.0 LayerNorm folded into lm_head.1
      std::vector<float> wfold, cfold;
      fold_layernorm(wt.data(), D, VPAD, wf.t.at("inference_model.lm_head.0.weight").data.data(),
                     wf.t.at("inference_model.lm_head.0.bias").data.data(), bt.data(), wfold, cfold);
      if (st->f32_mfma) { r = upload(ctx, st.get(), pack_mfma16(wfold.data(), D, VPAD), &st->d_lm); if (r) return r; }
      r = upload_h(ctx, st.get(), pack_mfma16h(wfold.data(), D, VPAD), &st->dh_lm); if (r) return r;
      if (ctx->ar_weights == 1) { r = upload_h(ctx, st.get(), pack_mfma16q(wfold.data(), D, VPAD), &st->q_lm); if (r) return r; }
      if (ctx->ar_weights == 2) {
        const std::vector<float> sc = fp8_col_scales(wfold.data(), D, VPAD);
        const std::vector<uint8_t> o = pack_mfma16o(wfold.data(), D, VPAD, sc);
        void *q = nullptr;
        TTS_HIP(ctx, hipMalloc(&q, o.size()));
        st->owned.push_back(q);
        TTS_HIP(ctx, ar_h2d(q, o.data(), o.size()));
        st->o_lm = (uint8_t *)q;
        r = upload(ctx, st.get(), sc, &st->os_lm); if (r) return r;
      }
      r = upload(ctx, st.get(), cfold, &st->d_lmb); if (r) return r;
    }
    r = upload(ctx, st.get(), bt, &st->lm_b); if (r) return r;
    return TTS_OK;
  };
  const auto t1 = std::chrono::steady_clock::now();
  const int nl = st->n_layers;
  if (int r = run_parallel(ctx, nl + 2, [&](int i) { return i == 0 ? build_head() : i == 1 ? build_globals() : build_layer(i - 2); })) return r; // the head (0.2 s) first
  if (dev_pack) { // the range guard of the split-precision layouts, from the device's max |w| values, in the host path's order and words
    TTS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    std::vector<unsigned> mx((size_t)nl * 6);
    TTS_HIP(ctx, hipMemcpy(mx.data(), d_max, mx.size() * 4, hipMemcpyDeviceToHost));
    static const char *wn[6] = {".attn.c_attn.weight", ".attn.c_proj.weight", ".mlp.c_fc.weight", ".mlp.c_proj.weight", ".attn.c_attn.weight (LayerNorm gain folded in)",
                                ".mlp.c_fc.weight (LayerNorm gain folded in)"};
    static const int order[6] = {0, 1, 2, 3, 4, 5};
    for (int i = 0; i < nl; i++)
      for (int j : order) {
        float m;
        memcpy(&m, &mx[(size_t)i * 6 + j], 4);
        if (!(m * W16_SCALE < 60000.0f))
          return fail(ctx, TTS_ERR_FORMAT, "tensor '%s': max |w| = %g does not fit the split-precision fp16 layout (|w| < %g)", (hp + std::to_string(i) + wn[j]).c_str(), m, 60000.0 / W16_SCALE);
      }
  }
  const double t_layers = since(t1);
#undef FETCH
#undef FETCHT
  for (int i = 0; i < st->n_layers; i++) { // split-precision copies for the multi-row MFMA path
    ArLayerDev &l = st->L[i];
    struct { const float *w; int K, N; __half **dst; } jobs[4] = {
        {l.w_attn, D, 3 * D, &l.s_attn}, {l.w_proj, D, D, &l.s_proj}, {l.w_fc, D, FF, &l.s_fc}, {l.w_fc2, FF, D, &l.s_fc2}};
    for (auto &j : jobs) {
      void *p = dalloc((size_t)j.N * 2 * j.K * sizeof(__half), false);
      if (!p) return TTS_ERR_HIP;
      split_weight_kernel<<<dim3(j.N / 32, j.K / 32), 256, 0, ctx->stream>>>(j.w, j.K, j.N, (__half *)p);
      *j.dst = (__half *)p;
    }
  }
  TTS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  st->loaded_wmode = ctx->ar_weights;
  if (ctx->ar) ar_free(ctx->ar);
  ctx->ar = st.release();
  if (timing) fprintf(stderr, "[tts timing] AR load: file %.1f ms, %d layers %.1f, total %.1f\n", t_read, ctx->ar->n_layers, t_layers, since(t0));
  return TTS_OK;
}

// ---- launch helpers -----------------------------------------------------------------------------
static int pick_rt(int rows) { int rt = 1; while (rt < rows && rt < 16) rt <<= 1; return rt; }

// part <- X[rows][K] * W[K][N]; returns ks through *ks_out. pin_ks > 0: X is a partial buffer (see PRO).
static int launch_gemv(tts_ctx *ctx, ArState *st, const float *X, int ldx, int rows, const float *W, int N, int K,
                       int *ks_out, float *part = nullptr, int pin_ks = 0, const float *pin_bias = nullptr) {
  const int rt = pick_rt(rows);
  const int ztiles = (rows + rt - 1) / rt, strips = N / 64;
  int ks = 1;
  while (ks * 2 * strips * ztiles <= 768 && K / (ks * 2) >= 32) ks *= 2;
  const int kchunk = K / ks; // K range per block (staged through LDS in chunks of <= 256)
  if (!part) {
    TTS_HIP(ctx, st->part.reserve((size_t)ks * rows * N * sizeof(float)));
    part = st->part.as<float>();
  }
  dim3 grid(strips, ks, ztiles);
  ProfScope ps(ctx, "ar_gemv", (double)K * N * 4.0 * ztiles); // weight bytes streamed
#define GEMV_LAUNCH(RT_)                                                                                                   \
  if (pin_ks > 0) gemv_kn_kernel<RT_, 1><<<grid, 256, 0, ctx->stream>>>(X, ldx, rows, W, N, K, kchunk, part, pin_ks, pin_bias, ctx->ggml_lut); \
  else gemv_kn_kernel<RT_, 0><<<grid, 256, 0, ctx->stream>>>(X, ldx, rows, W, N, K, kchunk, part, 0, nullptr, 0)
  switch (rt) {
    case 1: GEMV_LAUNCH(1); break;
    case 2: GEMV_LAUNCH(2); break;
    case 4: GEMV_LAUNCH(4); break;
    case 8: GEMV_LAUNCH(8); break;
    default: GEMV_LAUNCH(16); break;
  }
#undef GEMV_LAUNCH
  TTS_HIP(ctx, hipGetLastError());
  *ks_out = ks;
  return TTS_OK;
}

template <int MODE>
static int launch_epilogue(tts_ctx *ctx, ArState *st, int ks, int rows, int N, int n_valid, const float *bias, float *out,
                           int ldo, KvDst kv, float pscale = 1.0f) {
  ProfScope ps(ctx, "ar_epilogue");
  epilogue_kernel<MODE><<<dim3(rows, (n_valid + 255) / 256), 256, 0, ctx->stream>>>(st->part.as<float>(), ks, rows, N, n_valid, bias,
                                                                                    out, ldo, kv, ctx->ggml_lut, pscale);
  TTS_HIP(ctx, hipGetLastError());
  return TTS_OK;
}


// part[rows][N] (scaled by 64) <- split-precision MFMA product of X[rows][K] with the layer's hi|lo weights.
static int launch_mfma_matmul(tts_ctx *ctx, ArState *st, const float *X, int rows, const __half *Wsplit, int N, int K) {
  const int rpad = (rows + 127) & ~127;
  TTS_HIP(ctx, st->a_hi.reserve((size_t)rpad * FF * sizeof(__half)));
  TTS_HIP(ctx, st->a_lo.reserve((size_t)rpad * FF * sizeof(__half)));
  TTS_HIP(ctx, st->part.reserve((size_t)rpad * N * sizeof(float)));
  split_act_kernel<<<rpad, 256, 0, ctx->stream>>>(X, rows, K, st->a_hi.as<__half>(), st->a_lo.as<__half>());
  GemmArgs g{};
  g.A[0] = st->a_hi.as<__half>(); g.A[1] = st->a_lo.as<__half>(); g.A[2] = st->a_hi.as<__half>();
  g.row_off[0] = g.row_off[1] = g.row_off[2] = 0;
  g.nseg = 3; g.kseg = K; g.lda = K; g.W = Wsplit;
  g.custom_w = 1; g.ldw_ = 2 * K; g.w_off_[0] = 0; g.w_off_[1] = 0; g.w_off_[2] = K; // hi*hi + lo*hi + hi*lo
  g.M = rpad; g.N = N; g.bias = nullptr; g.row_seq = nullptr;
  g.mode = GEMM_OUT_F32; g.outF = st->part.as<float>(); g.ldo = N; g.resid = nullptr;
  ProfScope ps(ctx, "ar_mfma_gemm", 2.0 * rows * (double)N * K);
  TTS_HIP(ctx, launch_gemm_f16(g, ctx->stream));
  return TTS_OK;
}

#define CHECK(x) do { int _r = (x); if (_r) return _r; } while (0)
#define DEC_LN_LAUNCH(EPI_, GRID_, HT_)                                                            \
  do {                                                                                             \
    if (st->f32_mfma) dec_ln_gemv_kernel<EPI_, 0, false, HT_><<<GRID_, 256, 0, ctx->stream>>>(a);   \
    else dec_ln_gemv_kernel<EPI_, 1, true, HT_><<<GRID_, 256, 0, ctx->stream>>>(a);                 \
  } while (0)


// Transformer stack over rows = n_cand_rows * S laid out [cand][pos] in st->h.
//   kc/vc: fp16 K/V destination [cand][kv_max_pos][1024] per layer (layer_stride halves apart).
static int run_layers(tts_ctx *ctx, ArState *st, int rows, int S, int n_past, __half *kc, __half *vc,
                      size_t layer_stride, int kv_max_pos, int replicate) {
  float *h = st->h.as<float>(), *xn = st->xn.as<float>(), *qkv = st->qkv.as<float>();
  float *att = st->att.as<float>(), *ff = st->ff.as<float>();
  KvDst nokv{nullptr, nullptr, 1, 0, 0, 0};
  const bool mfma = rows >= 32; // multi-row passes: split-precision MFMA GEMM; tiny row counts: exact f32 GEMV
  const float ps = mfma ? 1.0f / W16_SCALE : 1.0f;
  for (int l = 0; l < st->n_layers; l++) {
    const ArLayerDev &w = st->L[l];
    int ks = 1;
    { ProfScope ps(ctx, "ar_layernorm");
      layernorm_kernel<<<rows, 256, 0, ctx->stream>>>(h, w.ln1_g, w.ln1_b, xn); }
    if (mfma) CHECK(launch_mfma_matmul(ctx, st, xn, rows, w.s_attn, 3 * D, D));
    else CHECK(launch_gemv(ctx, st, xn, D, rows, w.w_attn, 3 * D, D, &ks));
    KvDst kv{kc + l * layer_stride, vc + l * layer_stride, S, n_past, kv_max_pos, replicate};
    CHECK(launch_epilogue<EPI_QKV>(ctx, st, ks, rows, 3 * D, 3 * D, w.b_attn, qkv, 3 * D, kv, ps));
    { ProfScope ps(ctx, "ar_attention");
      if (ctx->ggml_lut) attention_kernel<<<dim3(rows, NH), 64, 0, ctx->stream>>>(qkv, kv.k, kv.v, att, S, n_past, kv_max_pos, 1);
      else attention_rows_kernel<<<dim3(rows / S, NH, (S + 63) / 64), 256, 0, ctx->stream>>>(qkv, kv.k, kv.v, att, S, n_past, kv_max_pos); }
    if (mfma) CHECK(launch_mfma_matmul(ctx, st, att, rows, w.s_proj, D, D));
    else CHECK(launch_gemv(ctx, st, att, D, rows, w.w_proj, D, D, &ks));
    CHECK(launch_epilogue<EPI_RESID>(ctx, st, ks, rows, D, D, w.b_proj, h, D, nokv, ps));
    { ProfScope ps(ctx, "ar_layernorm");
      layernorm_kernel<<<rows, 256, 0, ctx->stream>>>(h, w.ln2_g, w.ln2_b, xn); }
    if (mfma) CHECK(launch_mfma_matmul(ctx, st, xn, rows, w.s_fc, FF, D));
    else CHECK(launch_gemv(ctx, st, xn, D, rows, w.w_fc, FF, D, &ks));
    CHECK(launch_epilogue<EPI_GELU>(ctx, st, ks, rows, FF, FF, w.b_fc, ff, FF, nokv, ps));
    if (mfma) CHECK(launch_mfma_matmul(ctx, st, ff, rows, w.s_fc2, D, FF));
    else CHECK(launch_gemv(ctx, st, ff, FF, rows, w.w_fc2, D, FF, &ks));
    CHECK(launch_epilogue<EPI_RESID>(ctx, st, ks, rows, D, D, w.b_fc2, h, D, nokv, ps));
  }
  TTS_HIP(ctx, hipGetLastError());
  return TTS_OK;
}

static int reserve_rows(tts_ctx *ctx, ArState *st, int rows) {
  TTS_HIP(ctx, st->h.reserve((size_t)rows * D * 4));
  TTS_HIP(ctx, st->xn.reserve((size_t)rows * D * 4));
  TTS_HIP(ctx, st->hn.reserve((size_t)rows * D * 4));
  TTS_HIP(ctx, st->qkv.reserve((size_t)rows * 3 * D * 4));
  TTS_HIP(ctx, st->att.reserve((size_t)rows * D * 4));
  TTS_HIP(ctx, st->ff.reserve((size_t)rows * FF * 4));
  TTS_HIP(ctx, st->desc.reserve((size_t)rows * sizeof(int4)));
  return TTS_OK;
}

static int embed(tts_ctx *ctx, ArState *st, const std::vector<int4> &desc) {
  TTS_HIP(ctx, hipMemcpyAsync(st->desc.p, desc.data(), desc.size() * sizeof(int4), hipMemcpyHostToDevice, ctx->stream));
  EmbedTables t{{st->voice.as<float>(), st->text_emb, st->mel_emb}, {st->text_pos, st->mel_pos}};
  embed_rows_kernel<<<(int)desc.size(), 256, 0, ctx->stream>>>(t, st->desc.as<int4>(), st->h.as<float>());
  TTS_HIP(ctx, hipGetLastError());
  return TTS_OK;
}

int ar_begin(tts_ctx *ctx, const int32_t *text_ids, int n_text, const float *voice, int B, int max_steps) {
  ArState *st = ctx->ar;
  if (!st) return fail(ctx, TTS_ERR_STATE, "AR model not loaded");
  if (n_text < 1 || B < 1 || max_steps < 1 || !text_ids || !voice) return fail(ctx, TTS_ERR_ARG, "tts_ar_begin: bad argument");
  if (n_text > 404) return fail(ctx, TTS_ERR_LIMIT, "text has %d ids; the model has 404 text positions", n_text);
  if (max_steps + 2 > 608) return fail(ctx, TTS_ERR_LIMIT, "max_steps %d exceeds the 608 mel positions", max_steps);
  for (int i = 0; i < n_text; i++)
    if (text_ids[i] < 0 || text_ids[i] >= 256) return fail(ctx, TTS_ERR_ARG, "text id %d out of range", text_ids[i]);
  st->B = B; st->n_text = n_text; st->P = n_text + 2;
  st->prefill_done = false;
  st->max_pos = st->P + max_steps + 1;
  if (st->max_pos > 1024) return fail(ctx, TTS_ERR_LIMIT, "context of %d positions exceeds 1024", st->max_pos);
  st->tokens.assign(text_ids, text_ids + n_text);
  TTS_HIP(ctx, st->voice.reserve(D * 4));
  TTS_HIP(ctx, hipMemcpy(st->voice.p, voice, D * 4, hipMemcpyHostToDevice));
  size_t cache = (size_t)st->n_layers * B * st->max_pos * D * sizeof(__half);
  TTS_HIP(ctx, st->kcache.reserve(cache));
  TTS_HIP(ctx, st->vcache.reserve(cache));
  TTS_HIP(ctx, st->d_toks.reserve((size_t)(B + 2) * 4)); // [tokens | n_past, pos_id]
  TTS_HIP(ctx, st->logits.reserve((size_t)B * V * 4));
  if (B > st->h_cap_B) { // pinned allocations are slow (milliseconds): keep them across utterances
    if (st->h_toks) (void)hipHostFree(st->h_toks);
    if (st->h_logits) (void)hipHostFree(st->h_logits);
    if (st->h_pf) (void)hipHostFree(st->h_pf);
    st->h_toks = nullptr; st->h_logits = nullptr; st->h_pf = nullptr; st->h_cap_B = 0;
    TTS_HIP(ctx, hipHostMalloc((void **)&st->h_toks, (size_t)(B + 2) * 4));
    TTS_HIP(ctx, hipHostMalloc((void **)&st->h_logits, (size_t)B * V * 4));
    TTS_HIP(ctx, hipHostMalloc((void **)&st->h_pf, (size_t)B * TTS_PF_WORDS * 4));
    st->h_cap_B = B;
  }
  // the decode-step graph is kept: ar_step re-captures it only if a buffer moved or the batch shape changed (GraphSig)
  // the decode step's h4 layout holds whole tiles of 16 candidates: the padding rows of the last tile are read (never stored), keep them finite
  CHECK(reserve_rows(ctx, st, std::max((B + 15) / 16 * 16, st->P)));
  TTS_HIP(ctx, hipMemsetAsync(st->h.p, 0, st->h.cap, ctx->stream));
  return TTS_OK;
}

// Prefill (main.cpp:2586-2665): [voice | text_emb+pos | mel_emb(8192)+mel_pos(0)] — identical for all
// candidates, so it is evaluated once and its K/V rows are written into every candidate's cache.
int ar_prefill(tts_ctx *ctx, float *logits_out) {
  ArState *st = ctx->ar;
  if (!st || st->B == 0) return fail(ctx, TTS_ERR_STATE, "tts_ar_begin not called");
  const int P = st->P;
  std::vector<int4> desc(P);
  desc[0] = make_int4(0, 0, -1, 0);
  for (int i = 0; i < st->n_text; i++) desc[1 + i] = make_int4(1, st->tokens[i], 0, i);
  desc[P - 1] = make_int4(2, 8192, 1, 0);
  CHECK(embed(ctx, st, desc));
  const size_t layer_stride = (size_t)st->B * st->max_pos * D;
  // The prompt pass runs on the decode-step kernels, tiled over 16 positions (exact f32, weights re-read from L2 per
  // tile): rows are positions of the one shared prompt, the QKV epilogue replicates K/V into every candidate's cache.
  const int tiles = (P + 15) / 16;
  float *h = st->h.as<float>(), *qkv = st->qkv.as<float>(), *att = st->att.as<float>(), *ff = st->ff.as<float>();
  for (int l = 0; l < st->n_layers; l++) {
    const ArLayerDev &w = st->L[l];
    __half *kc = st->kcache.as<__half>() + l * layer_stride, *vc = st->vcache.as<__half>() + l * layer_stride;
    { ProfScope ps(ctx, "ar_gemv", 3.0 * D * D * 4.0 * tiles);
      DecLnArgs a{h, nullptr, nullptr, w.d_attn, w.dh_attn, w.db_attn, P, 3 * D, 3 * D, st->B, qkv, kc, vc, nullptr, st->max_pos, ctx->ggml_lut};
      DEC_LN_LAUNCH(DEC_QKV, dim3(3 * D / 16, tiles), false); }
    { ProfScope ps(ctx, "ar_attention");
      attention_kernel<<<dim3(P, NH), 64, 0, ctx->stream>>>(qkv, kc, vc, att, P, 0, st->max_pos, ctx->ggml_lut); }
    { ProfScope ps(ctx, "ar_gemv", 1.0 * D * D * 4.0 * tiles);
      dec_gemv_resid_kernel<1><<<dim3(D / 4, tiles), 256, 0, ctx->stream>>>(att, P, w.d_proj, w.b_proj, h); }
    { ProfScope ps(ctx, "ar_gemv", 4.0 * D * D * 4.0 * tiles);
      DecLnArgs a{h, nullptr, nullptr, w.d_fc, w.dh_fc, w.db_fc, P, FF, 0, 0, ff, nullptr, nullptr, nullptr, 0, ctx->ggml_lut};
      DEC_LN_LAUNCH(DEC_GELU, dim3(FF / 16, tiles), false); }
    { ProfScope ps(ctx, "ar_gemv", 4.0 * D * D * 4.0 * tiles);
      dec_gemv_resid_kernel<4, 512><<<dim3(D / 4, tiles), 512, 0, ctx->stream>>>(ff, P, w.d_fc2, w.b_fc2, h); }
  }
  TTS_HIP(ctx, st->logits.reserve((size_t)V * 4));
  { ProfScope ps(ctx, "ar_gemv", (double)D * VPAD * 4.0);
    DecLnArgs a{h + (size_t)(P - 1) * D, st->lnf_g, st->lnf_b, st->d_lm, st->dh_lm, st->d_lmb, 1, V, V, 0, st->logits.as<float>(),
                nullptr, nullptr, nullptr, 0, ctx->ggml_lut};
    DEC_LN_LAUNCH(DEC_LOGITS, dim3(VPAD / 16, 1), false); }
  TTS_HIP(ctx, hipGetLastError());
  if (logits_out) {
    TTS_HIP(ctx, hipMemcpyAsync(logits_out, st->logits.p, (size_t)V * 4, hipMemcpyDeviceToHost, ctx->stream));
    TTS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int c = 1; c < st->B; c++) memcpy(logits_out + (size_t)c * V, logits_out, (size_t)V * 4);
  } else {
    TTS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  }
  st->prefill_done = true;
  return TTS_OK;
}

// One decode step for all candidates, enqueued on the ctx stream (captured once into a hipGraph): five
// launches per layer (see the kernels above) + embed + head; every launch streams its weights exactly once
// per tile of 16 candidates.
static int enqueue_decode_step(tts_ctx *ctx, ArState *st) {
  const int B = st->B, tiles = (B + 15) / 16;
  float *h = st->h.as<float>() /* h4 layout inside the step */, *q = st->qkv.as<float>(), *att = st->att.as<float>(), *ff = st->ff.as<float>();
  const StepState *ss = (const StepState *)(st->d_toks.as<int>() + B);
  const size_t layer_stride = (size_t)B * st->max_pos * D;
  const int wm = ctx->ar_weights; // 1 / 2: fp16 / fp8 weights, a half / a quarter of the bytes per step (throughput modes, not f32-exact); checked by ar_step
  const double wb = wm == 2 ? 1.0 : wm == 1 ? 2.0 : 4.0;
  embed_step_kernel<<<B, 256, 0, ctx->stream>>>(st->mel_emb, st->mel_pos, st->h_toks, B, (StepState *)(st->d_toks.as<int>() + B), h);
  for (int l = 0; l < st->n_layers; l++) {
    const ArLayerDev &w = st->L[l];
    __half *kc = st->kcache.as<__half>() + l * layer_stride, *vc = st->vcache.as<__half>() + l * layer_stride;
    { ProfScope ps(ctx, "ar_gemv", 3.0 * D * D * wb * tiles);
      DecLnArgs a{h, nullptr, nullptr, w.d_attn, wm == 2 ? (const __half *)w.o_attn : wm == 1 ? w.q_attn : w.dh_attn, w.db_attn, B, 3 * D, D, 0, q, kc, vc, ss,
                  st->max_pos, ctx->ggml_lut, w.os_attn};
      if (wm == 2) dec_ln_gemv_kernel<DEC_QKV, 3, true, true><<<dim3(3 * D / 16, tiles), 256, 0, ctx->stream>>>(a);
      else if (wm == 1) dec_ln_gemv_kernel<DEC_QKV, 2, false, true><<<dim3(3 * D / 16, tiles), 256, 0, ctx->stream>>>(a);
      else DEC_LN_LAUNCH(DEC_QKV, dim3(3 * D / 16, tiles), true); }
    { ProfScope ps(ctx, "ar_attention");
      if (ctx->ggml_lut) attn_decode_kernel<<<dim3(B, NH), 256, 0, ctx->stream>>>(q, kc, vc, ss, st->max_pos, att, 1);
      else attn_decode_fast_kernel<<<dim3(B, NH), 256, 0, ctx->stream>>>(q, kc, vc, ss, st->max_pos, att); }
    { ProfScope ps(ctx, "ar_gemv", 1.0 * D * D * wb * tiles);
      if (wm == 2) dec_gemv_resid_kernel<1, 256, 2, false, true><<<dim3(D / 4, tiles), 256, 0, ctx->stream>>>(att, B, (const float *)w.o_proj, w.b_proj, h, w.os_proj);
      else if (wm == 1) dec_gemv_resid_kernel<1, 256, 1, false, true><<<dim3(D / 4, tiles), 256, 0, ctx->stream>>>(att, B, (const float *)w.q_proj, w.b_proj, h);
      else dec_gemv_resid_kernel<1, 256, 0, true, true><<<dim3(D / 4, tiles), 256, 0, ctx->stream>>>(att, B, w.d_proj, w.b_proj, h); }
    { ProfScope ps(ctx, "ar_gemv", 4.0 * D * D * wb * tiles);
      DecLnArgs a{h, nullptr, nullptr, w.d_fc, wm == 2 ? (const __half *)w.o_fc : wm == 1 ? w.q_fc : w.dh_fc, w.db_fc, B, FF, 0, 0, ff, nullptr, nullptr, ss, 0,
                  ctx->ggml_lut, w.os_fc};
      if (wm == 2) dec_ln_gemv_kernel<DEC_GELU, 3, true, true><<<dim3(FF / 16, tiles), 256, 0, ctx->stream>>>(a);
      else if (wm == 1) dec_ln_gemv_kernel<DEC_GELU, 2, false, true><<<dim3(FF / 16, tiles), 256, 0, ctx->stream>>>(a);
      else DEC_LN_LAUNCH(DEC_GELU, dim3(FF / 16, tiles), true); }
    { ProfScope ps(ctx, "ar_gemv", 4.0 * D * D * wb * tiles);
      if (wm == 2) dec_gemv_resid_kernel<4, 512, 2, false, true><<<dim3(D / 4, tiles), 512, 0, ctx->stream>>>(ff, B, (const float *)w.o_fc2, w.b_fc2, h, w.os_fc2);
      else if (wm == 1) dec_gemv_resid_kernel<4, 512, 1, false, true><<<dim3(D / 4, tiles), 512, 0, ctx->stream>>>(ff, B, (const float *)w.q_fc2, w.b_fc2, h);
      else dec_gemv_resid_kernel<4, 512, 0, true, true><<<dim3(D / 4, tiles), 512, 0, ctx->stream>>>(ff, B, w.d_fc2, w.b_fc2, h); }
  }
  { ProfScope ps(ctx, "ar_gemv", (double)D * VPAD * wb * tiles);
    DecLnArgs a{h, st->lnf_g, st->lnf_b, st->d_lm, wm == 2 ? (const __half *)st->o_lm : wm == 1 ? st->q_lm : st->dh_lm, st->d_lmb, B, V, V, 0, st->logits.as<float>(),
                nullptr, nullptr, ss, 0, ctx->ggml_lut, st->os_lm};
    if (wm == 2) dec_ln_gemv_kernel<DEC_LOGITS, 3, true, true><<<dim3(VPAD / 16, tiles), 256, 0, ctx->stream>>>(a);
    else if (wm == 1) dec_ln_gemv_kernel<DEC_LOGITS, 2, false, true><<<dim3(VPAD / 16, tiles), 256, 0, ctx->stream>>>(a);
    else DEC_LN_LAUNCH(DEC_LOGITS, dim3(VPAD / 16, tiles), true); }
  if (st->step_mode == 0) {
    TTS_HIP(ctx, hipMemcpyAsync(st->h_logits, st->logits.p, (size_t)B * V * 4, hipMemcpyDeviceToHost, ctx->stream));
  } else { // the sampler's top-k prefilter on the device: 16 KB instead of 524 KB back to the host per step of 16 candidates
    ProfScope ps(ctx, "ar_prefilter", (double)B * V * 4.0);
    // the lists are written straight into pinned host memory (1 KB per candidate over PCIe from the kernel's stores): no copy node behind the graph
    sample_prefilter_kernel<<<B, 256, 0, ctx->stream>>>(st->logits.as<float>(), st->step_mode == 2, st->h_pf);
  }
  TTS_HIP(ctx, hipGetLastError());
  return TTS_OK;
}

// Decode step i (main.cpp:2667-2693, 5227-5247): mel_emb[tok] + mel_pos[i+2], n_past = P + i.
// mode 0: the logits come back ([B][8194], logits_out may be null: ar_host_logits); mode 1 / 2: the device prefilter's lists come back
// instead (ar_host_lists; 2 = stop token masked first) and the logits stay in HBM (ar_fetch_logits_row). Each mode is its own graph.
int ar_step(tts_ctx *ctx, const int32_t *prev_ids, int step_i, float *logits_out, int mode) {
  ArState *st = ctx->ar;
  if (!st || st->B == 0) return fail(ctx, TTS_ERR_STATE, "tts_ar_begin not called");
  if (mode < 0 || mode > 2 || (mode && logits_out)) return fail(ctx, TTS_ERR_ARG, "ar_step: bad mode");
  st->step_mode = mode;
  if (step_i < 0 || st->P + step_i >= st->max_pos) return fail(ctx, TTS_ERR_LIMIT, "step %d beyond the KV cache", step_i);
  for (int c = 0; c < st->B; c++) {
    if (prev_ids[c] < 0 || prev_ids[c] >= V) return fail(ctx, TTS_ERR_ARG, "mel token %d out of range", prev_ids[c]);
    st->h_toks[c] = prev_ids[c];
  }
  st->h_toks[st->B] = st->P + step_i; // n_past
  st->h_toks[st->B + 1] = step_i + 2; // mel position id (main.cpp:5244)
  // checked here, not inside enqueue_decode_step: that runs between hipStreamBeginCapture and hipStreamEndCapture
  if (ctx->ar_weights != 0 && ctx->ar_weights != st->loaded_wmode)
    return fail(ctx, TTS_ERR_STATE, "option ar_weights = %d must be set before tts_load_ar (the fp16 / fp8 slabs are packed at load)", ctx->ar_weights);
  static const bool no_graph = getenv("TTS_NO_GRAPH") != nullptr; // e.g. under rocprofv3, which crashes on graph replays here
  // event records are not captured: the step runs eagerly while one of its own kernel families ("ar_*") is profiled
  // ("ar_decode_step" brackets the whole graph replay and keeps the graph)
  bool prof_ar = ctx->prof_on && ctx->prof_filter.empty();
  for (const std::string &f : ctx->prof_filter) prof_ar |= ctx->prof_on && f.rfind("ar_", 0) == 0 && f != "ar_decode_step";
  if (prof_ar || no_graph) {
    CHECK(enqueue_decode_step(ctx, st));
  } else {
    const ArState::GraphSig sig = st->current_sig(ctx->ggml_lut, ctx->ar_weights);
    if (st->graph_execs[mode] && !(sig == st->graph_sigs[mode])) st->drop_graph(mode);
    if (!st->graph_execs[mode]) {
      st->graph_sigs[mode] = sig;
      TTS_HIP(ctx, hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
      int rc = enqueue_decode_step(ctx, st);
      hipError_t e = hipStreamEndCapture(ctx->stream, &st->graphs[mode]);
      if (rc || e != hipSuccess) { st->drop_graph(mode); if (rc) return rc; TTS_HIP(ctx, e); } // never keep a half-captured graph
      e = hipGraphInstantiate(&st->graph_execs[mode], st->graphs[mode], nullptr, nullptr, 0);
      if (e != hipSuccess) { st->drop_graph(mode); TTS_HIP(ctx, e); }
    }
    // HBM-bound step: every weight once (f32: 12 d^2 per layer + the padded head) + the fp16 K/V rows read + logits
    const double step_bytes = (ctx->ar_weights == 2 ? 1.0 : ctx->ar_weights == 1 ? 2.0 : 4.0) * ((double)st->n_layers * 12.0 * D * D + (double)D * V) +
                              (double)st->B * st->n_layers * 2.0 * (st->P + step_i + 1) * D * 2.0 + (double)st->B * V * 4.0;
    ProfScope ps(ctx, "ar_decode_step", step_bytes);
    TTS_HIP(ctx, hipGraphLaunch(st->graph_execs[mode], ctx->stream));
  }
  TTS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (logits_out) memcpy(logits_out, st->h_logits, (size_t)st->B * V * 4);
  return TTS_OK;
}

// The pinned lists of the last mode 1 / 2 step ([B][TTS_PF_WORDS]).
const int32_t *ar_host_lists(tts_ctx *ctx) { return ctx->ar ? ctx->ar->h_pf : nullptr; }

// One candidate's logits of the last step, from HBM (the sampler's fallback when a list cannot decide). Returns the pinned row or null.
const float *ar_fetch_logits_row(tts_ctx *ctx, int c) {
  ArState *st = ctx->ar;
  if (!st || c < 0 || c >= st->B) return nullptr;
  float *dst = st->h_logits + (size_t)c * V;
  if (hipMemcpyAsync(dst, st->logits.as<float>() + (size_t)c * V, (size_t)V * 4, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) return nullptr;
  if (hipStreamSynchronize(ctx->stream) != hipSuccess) return nullptr;
  if (st->step_mode == 2) dst[V - 1] = -1e30f;
  return dst;
}

// Latent pass (main.cpp:2053-2519, 5280-5352): full causal forward without the decode cache.
// dst[l][c][p][:] = src[l][0][p][:] for p < n_prompt: the prompt rows of the decode cache (identical for every candidate)
// become the first rows of every candidate in the latent pass' K/V buffers. grid (n_prompt, nb, 2 * n_layers).
__global__ __launch_bounds__(256) void copy_prompt_kv_kernel(const __half *__restrict__ ksrc, const __half *__restrict__ vsrc,
                                                             size_t src_layer_stride, __half *__restrict__ kdst, __half *__restrict__ vdst,
                                                             size_t dst_layer_stride, int S, int n_layers) {
  const int p = blockIdx.x, c = blockIdx.y, l = blockIdx.z % n_layers;
  const bool isv = (int)blockIdx.z >= n_layers;
  const __half *src = (isv ? vsrc : ksrc) + l * src_layer_stride + (size_t)p * D;
  __half *dst = (isv ? vdst : kdst) + l * dst_layer_stride + ((size_t)c * S + p) * D;
  ((uint2 *)dst)[threadIdx.x] = ((const uint2 *)src)[threadIdx.x];
}

// Latent pass (main.cpp:2053-2519): the full stack over [voice | text | mel codes at mel positions 0..] without the
// decode cache's position quirk. The 1 + n_text prompt rows are the same for every candidate and do not depend on the
// mel rows (causal mask): their K/V rows are taken from the decode cache (written by the prompt pass), so the stack runs
// over the mel rows only, with n_past = 1 + n_text.
int ar_latents(tts_ctx *ctx, const int32_t *codes502, int nb, int n_mel, float *out) {
  ArState *st = ctx->ar;
  if (!st || st->n_text == 0) return fail(ctx, TTS_ERR_STATE, "tts_ar_begin not called");
  if (nb < 1 || n_mel < 1 || n_mel > 502) return fail(ctx, TTS_ERR_ARG, "tts_ar_latents: bad argument");
  const int Sp = 1 + st->n_text, S = Sp + n_mel, rows = nb * n_mel;
  if (S > 1024) return fail(ctx, TTS_ERR_LIMIT, "latent pass of %d positions exceeds 1024", S);
  for (int c = 0; c < nb; c++)
    for (int j = 0; j < n_mel; j++) {
      const int code = codes502[c * 502 + j];
      if (code < 0 || code >= V) return fail(ctx, TTS_ERR_ARG, "mel code %d out of range", code);
    }
  if (!st->prefill_done) CHECK(ar_prefill(ctx, nullptr));
  CHECK(reserve_rows(ctx, st, rows));
  const size_t lat_stride = (size_t)nb * S * D; // per layer: [cand][S][1024]
  TTS_HIP(ctx, st->lat_k.reserve(st->n_layers * lat_stride * sizeof(__half)));
  TTS_HIP(ctx, st->lat_v.reserve(st->n_layers * lat_stride * sizeof(__half)));
  copy_prompt_kv_kernel<<<dim3(Sp, nb, 2 * st->n_layers), 256, 0, ctx->stream>>>(
      st->kcache.as<__half>(), st->vcache.as<__half>(), (size_t)st->B * st->max_pos * D, st->lat_k.as<__half>(), st->lat_v.as<__half>(),
      lat_stride, S, st->n_layers);
  std::vector<int4> desc(rows);
  for (int c = 0; c < nb; c++)
    for (int j = 0; j < n_mel; j++) desc[(size_t)c * n_mel + j] = make_int4(2, codes502[c * 502 + j], 1, j);
  CHECK(embed(ctx, st, desc));
  CHECK(run_layers(ctx, st, rows, n_mel, Sp, st->lat_k.as<__half>(), st->lat_v.as<__half>(), lat_stride, S, 0));
  float *xn = st->xn.as<float>(), *hn = st->hn.as<float>();
  layernorm_kernel<<<rows, 256, 0, ctx->stream>>>(st->h.as<float>(), st->lnf_g, st->lnf_b, xn);
  layernorm_kernel<<<rows, 256, 0, ctx->stream>>>(xn, st->lmh_g, st->lmh_b, hn);
  TTS_HIP(ctx, hipGetLastError());
  const int n_out = std::min(500, n_mel);
  for (int c = 0; c < nb; c++)
    TTS_HIP(ctx, hipMemcpyAsync(out + (size_t)c * n_out * D, hn + (size_t)c * n_mel * D, (size_t)n_out * D * 4, hipMemcpyDeviceToHost,
                                ctx->stream));
  TTS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return TTS_OK;
}

int ar_layers(const tts_ctx *ctx) { return ctx->ar ? ctx->ar->n_layers : 0; }
int ar_batch(const tts_ctx *ctx) { return ctx->ar ? ctx->ar->B : 0; }

// The pinned buffer the decode graph copies the logits into ([B][8194]); valid after ar_step returns.
float *ar_host_logits(tts_ctx *ctx) { return ctx->ar ? ctx->ar->h_logits : nullptr; }

} // namespace tts

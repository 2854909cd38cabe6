// UnivNet vocoder stage on gfx950.
//
// Replaces vocoder_model_load (main.cpp:1665-2021), vocoder_graph (4068-4483) and vocoder() (6044-6127).
//
// All candidates run as one batch. Frame-rate tensors use the packed row layout of the diffusion stage
// (zero guard rows between candidates); an audio-rate tensor at `hop` samples per frame is indexed
// [row*hop + s][channel], so a sample's frame (and therefore its candidate and its predicted kernel)
// is pos / hop. The location-variable convolution is fused with bias, sigmoid*tanh gate and residual
// add: per frame the 64x96 predicted kernel is staged once in LDS and applied to `hop` samples — the
// reference materialises [64, hop, Tm, 32] windows and reduces them with 31 adds (main.cpp:4404-4419).
// The only heavy GEMM (kernel_conv 64 -> 24576, k=3) runs on the fp16 MFMA kernel; everything else is
// small f32 VALU work (32-channel audio-rate tensors: HBM/latency bound).
// Numerics: conv1d = fp16-rounded weights x fp16-rounded inputs, f32 accumulate; conv_transpose_1d and
// the LVC einsum in F32, as in the reference graph.
#include "common.h"
#include "gemm_f16.h"
#include <algorithm>
#include <cmath>

namespace tts {

__device__ __forceinline__ float leaky02(float v) { return v > 0.f ? v : 0.2f * v; }
__device__ __forceinline__ float r16(float v) { return __half2float(__float2half_rn(v)); }
// sigmoid(a) * tanh(b) on the hardware exp2/rcp (1e-7-level relative error; libm's expf + tanhf + a division are ~70
// VALU instructions per gate, more than half of the LVC kernel's time)
__device__ __forceinline__ float gate_dev(float a, float b) {
  const float sg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(a * -1.44269504088896f));
  const float th = 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(b * 2.88539008177793f)); // tanh b = 1 - 2/(1+e^{2b})
  return sg * th;
}

// Generic direct conv1d over packed positions. x [P][Cin] f32 -> y [P][Cout].
//   w: [K][Cin][Cout] f32 holding fp16-rounded values. Position p belongs to frame p / hop; taps that
//   leave the sequence (row_seq differs) read zero, or are reflected when `reflect` (conv_pre's
//   ggml_pad_reflect_1d on the noise). pre_leaky: leaky_relu(0.2) on the input (before the fp16 round);
//   post_leaky on the output; resid: y = resid + out (kernel-predictor residual blocks).
struct ConvArgs {
  const float *x; const float *w; const float *bias; float *y; const float *resid;
  const int *row_seq, *seq_start, *seq_len; // frame layout
  int P, Cin, Cout, K, dil, pad, hop, pre_leaky, post_leaky, reflect;
};
__global__ __launch_bounds__(256) void conv_direct_kernel(ConvArgs a) {
  const int per = 256 / a.Cout;
  const int co = threadIdx.x % a.Cout, pl = threadIdx.x / a.Cout;
  const int p = blockIdx.x * per + pl;
  if (pl >= per || p >= a.P) return;
  const int s = a.row_seq[p / a.hop];
  if (s < 0) { a.y[(size_t)p * a.Cout + co] = 0.f; return; }
  const int lo = a.seq_start[s] * a.hop, n = a.seq_len[s] * a.hop; // this sequence: [lo, lo+n)
  float acc = 0.f;
  for (int k = 0; k < a.K; k++) {
    int q = p - lo + k * a.dil - a.pad;
    if (a.reflect) { q = q < 0 ? -q : q; q = q >= n ? 2 * (n - 1) - q : q; }
    if (q < 0 || q >= n) continue;
    const float *xr = a.x + (size_t)(lo + q) * a.Cin;
    const float *wr = a.w + (size_t)k * a.Cin * a.Cout + co;
    for (int ci = 0; ci < a.Cin; ci++) {
      float xv = xr[ci];
      if (a.pre_leaky) xv = leaky02(xv);
      acc = fmaf(r16(xv), wr[(size_t)ci * a.Cout], acc);
    }
  }
  float v = acc + a.bias[co];
  if (a.post_leaky) v = leaky02(v);
  if (a.resid) v = a.resid[(size_t)p * a.Cout + co] + v;
  a.y[(size_t)p * a.Cout + co] = v;
}

// mel [100][T] (normalised) -> denormalised, padded frames: rows [row][100] f32 (conv input) with the
// 10 trailing frames = -11.5129 (main.cpp:5575-5584, 6051-6054). One block per row, 128 threads.
__global__ __launch_bounds__(128) void voc_mel_rows_kernel(const float *__restrict__ mel, const int64_t *__restrict__ mel_off,
                                                           const int *__restrict__ row_seq, const int *__restrict__ row_t,
                                                           const int *__restrict__ seq_len, float *__restrict__ out) {
  const int r = blockIdx.x, ch = threadIdx.x, s = row_seq[r];
  if (ch >= 100) return;
  float v = 0.f;
  if (s >= 0) {
    const int T = seq_len[s] - 10, t = row_t[r];
    if (t < T) {
      const float MAXV = 2.3143386840820312f, MINV = -11.512925148010254f;
      float m = mel[mel_off[s] + (size_t)ch * T + t];
      v = ((m + 1) / 2) * (MAXV - MINV) + MINV;
    } else v = -11.5129f;
  }
  out[(size_t)r * 100 + ch] = v;
}

// noise [64][Tm] per candidate (reference layout) -> rows [row][64]
__global__ __launch_bounds__(64) void voc_noise_rows_kernel(const float *__restrict__ nz, const int64_t *__restrict__ nz_off,
                                                            const int *__restrict__ row_seq, const int *__restrict__ row_t,
                                                            const int *__restrict__ seq_len, float *__restrict__ out) {
  const int r = blockIdx.x, ch = threadIdx.x, s = row_seq[r];
  out[(size_t)r * 64 + ch] = (s >= 0) ? nz[nz_off[s] + (size_t)ch * seq_len[s] + row_t[r]] : 0.f;
}

// f32 rows [R][64] -> fp16 with zero guard rows (operand of the MFMA kernel/bias convs)
__global__ __launch_bounds__(64) void voc_cond_f16_kernel(const float *__restrict__ x, const int *__restrict__ row_seq,
                                                          __half *__restrict__ y) {
  const int r = blockIdx.x, ch = threadIdx.x;
  y[(size_t)r * 64 + ch] = __float2half_rn(row_seq[r] >= 0 ? x[(size_t)r * 64 + ch] : 0.f);
}

// leaky -> ConvTranspose1d(32->32, K=2s, stride s) -> crop s/2 each side -> + bias (main.cpp:4145-4167).
// in [R*hop_in][32], out [R*hop_in*s][32]; w f32 [K][Cin][Cout]. F32 (the reference keeps this kernel F32).
__global__ __launch_bounds__(256) void convt_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                    const float *__restrict__ bias, const int *__restrict__ row_seq,
                                                    const int *__restrict__ seq_start, const int *__restrict__ seq_len, int hop_in,
                                                    int s, int64_t Pout, float *__restrict__ y) {
  // (all index arithmetic in 32 bits with shifts: hop_in and s are powers of two and Pout < 2^31; three 64-bit
  //  divisions per thread cost more than the 64 FMAs they index)
  const int co = threadIdx.x & 31;
  const int p = blockIdx.x * 8 + (threadIdx.x >> 5);
  if (p >= Pout) return;
  const int ls = 31 - __clz(s), lh = 31 - __clz(hop_in * s);
  const int sq = row_seq[p >> lh];
  if (sq < 0) { y[(size_t)p * 32 + co] = 0.f; return; }
  const int lo_in = seq_start[sq] * hop_in, n_in = seq_len[sq] * hop_in;
  const int tl = p - lo_in * s; // local output index (after crop)
  const int u = tl + s / 2;     // index in the uncropped transposed-conv output
  float acc = bias[co];
  // contributions: u = t*s + k, k in [0, 2s)  ->  t = u/s (k = u%s) and t-1 (k = u%s + s)
  const int t0 = u >> ls;
  const int k0 = u - (t0 << ls);
#pragma unroll
  for (int j = 0; j < 2; j++) {
    const int t = t0 - j;
    const int k = k0 + j * s;
    if (t < 0 || t >= n_in) continue;
    const float *xr = x + (size_t)(lo_in + t) * 32;
    const float *wr = w + (size_t)k * 32 * 32 + co;
    for (int ci = 0; ci < 32; ci++) acc = fmaf(leaky02(xr[ci]), wr[ci * 32], acc);
  }
  y[(size_t)p * 32 + co] = acc;
}

// Fused location-variable convolution + gate + residual (main.cpp:4365-4455):
//   o[ch][pos] = b_l[ch] + sum_{i<32} sum_{k<3} ypad[i][pos+k-1] * W_l[i][ch][k],  ch < 64
//   x[pos][c] += sigmoid(o[c]) * tanh(o[32+c])
// kern: [rows][24576] f32, layer l at l*6144, inside a layer in lvc_col order (below); kb: [rows][256], channel layer*64+ch.
// One block per (frame, chunk of 64 samples); W_l staged in LDS as [i*3+k][64].
__global__ __launch_bounds__(256) void lvc_gate_kernel(const float *__restrict__ y, const float *__restrict__ kern,
                                                       const float *__restrict__ kb, const int *__restrict__ row_seq, int hop,
                                                       int layer, float *__restrict__ x) {
  __shared__ float W[96 * 64];
  __shared__ float Y[66 * 33];
  const int row = blockIdx.y, s = row_seq[row];
  if (s < 0) return;
  const int chunk = min(64, hop), s0 = blockIdx.x * chunk;
  const float *kl = kern + (size_t)row * 24576 + (size_t)layer * 6144;
  for (int idx = threadIdx.x; idx < 6144; idx += 256) { // idx in lvc_col order (see lvc_mfma_kernel)
    const int j = idx & 3, ln = (idx >> 2) & 63, nt = (idx >> 8) & 3, g = (idx >> 10) & 1, k = idx >> 11;
    const int i = g * 16 + 4 * (ln >> 4) + j, ch = nt * 16 + (ln & 15);
    W[(i * 3 + k) * 64 + ch] = kl[idx];
  }
  // y window: positions [row*hop + s0 - 1, +chunk+2); outside the sequence -> 0 (ggml_pad_ext 1,1)
  const int64_t base = (int64_t)row * hop + s0 - 1;
  for (int idx = threadIdx.x; idx < (chunk + 2) * 32; idx += 256) {
    const int j = idx >> 5, c = idx & 31;
    const int64_t p = base + j;
    float v = 0.f;
    if (p >= 0) {
      const int rr = (int)(p / hop);
      if (row_seq[rr] == s) v = y[p * 32 + c];
    }
    Y[j * 33 + c] = v;
  }
  __syncthreads();
  const int c = threadIdx.x & 31;
  for (int sl = threadIdx.x >> 5; sl < chunk; sl += 8) {
    float a0 = 0.f, a1 = 0.f;
    for (int i = 0; i < 32; i++) {
#pragma unroll
      for (int k = 0; k < 3; k++) {
        const float yv = Y[(sl + k) * 33 + i];
        a0 = fmaf(yv, W[(i * 3 + k) * 64 + c], a0);
        a1 = fmaf(yv, W[(i * 3 + k) * 64 + 32 + c], a1);
      }
    }
    const float *bl = kb + (size_t)row * 256 + layer * 64;
    a0 += bl[c];
    a1 += bl[32 + c];
    const float g = gate_dev(a0, a1);
    const int64_t p = (int64_t)row * hop + s0 + sl;
    x[p * 32 + c] += g;
  }
}

// Order of the predicted kernel's 6144 values per (frame, layer) — chosen at load by permuting the output
// channels of kernel_conv, so that the kernel GEMM writes them directly as fp32-MFMA A fragments:
//   lvc_col(tap, i, ch) = (((tap*2 + g)*4 + nt)*64 + lane)*4 + j,  i = 16 g + 4 (lane>>4) + j,  ch = 16 nt + (lane&15)
// (reference order: (i*64 + ch)*3 + tap).
__host__ __device__ inline int lvc_col(int tap, int i, int ch) {
  const int g = i >> 4, j = i & 3, lane = ((i & 15) >> 2) * 16 + (ch & 15), nt = ch >> 4;
  return (((tap * 2 + g) * 4 + nt) * 64 + lane) * 4 + j;
}

// Location-variable convolution + gate + residual on the fp32 MFMA (exact f32 products, as the reference's F32
// einsum), for hop % 64 == 0. One workgroup = one frame: its 64 x 96 predicted kernel is 24 float4 per lane,
// loaded once with 1 KB-per-wave loads; a wave walks hop/64 tiles of 16 samples; operands swapped so that a lane
// ends with 4 consecutive channels of one sample. y is zero on guard frames (voc_dconv_mfma_kernel), so the
// window's zero padding at the sequence ends needs no branch.
__global__ __launch_bounds__(256) void lvc_mfma_kernel(const float *__restrict__ y, const float *__restrict__ kern,
                                                       const float *__restrict__ kb, const int *__restrict__ row_seq, int hop,
                                                       int layer, float *__restrict__ x) {
  const int row = blockIdx.x;
  if (row_seq[row] < 0) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, m = lane & 15, kq = lane >> 4;
  float4 w[3][2][4];
  {
    const float4 *wp = (const float4 *)(kern + (size_t)row * 24576 + (size_t)layer * 6144) + lane;
#pragma unroll
    for (int tap = 0; tap < 3; tap++)
#pragma unroll
      for (int g = 0; g < 2; g++)
#pragma unroll
        for (int nt = 0; nt < 4; nt++) w[tap][g][nt] = wp[((tap * 2 + g) * 4 + nt) * 64];
  }
  float4 bs[4];
#pragma unroll
  for (int nt = 0; nt < 4; nt++) bs[nt] = *(const float4 *)(kb + (size_t)row * 256 + layer * 64 + nt * 16 + 4 * kq);
  const int per_wave = hop >> 6;
  for (int t = 0; t < per_wave; t++) {
    const int64_t pos = (int64_t)row * hop + (wave * per_wave + t) * 16 + m;
    floatx4 acc[4];
#pragma unroll
    for (int nt = 0; nt < 4; nt++) acc[nt] = (floatx4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tap = 0; tap < 3; tap++)
#pragma unroll
      for (int g = 0; g < 2; g++) {
        const float4 yv = *(const float4 *)(y + (pos + tap - 1) * 32 + g * 16 + 4 * kq);
#pragma unroll
        for (int nt = 0; nt < 4; nt++) {
          acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[tap][g][nt].x, yv.x, acc[nt], 0, 0, 0);
          acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[tap][g][nt].y, yv.y, acc[nt], 0, 0, 0);
          acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[tap][g][nt].z, yv.z, acc[nt], 0, 0, 0);
          acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[tap][g][nt].w, yv.w, acc[nt], 0, 0, 0);
        }
      }
#pragma unroll
    for (int nt = 0; nt < 2; nt++) { // channels c = 16 nt + 4 kq + r gate with 32 + c
      float *xp = x + pos * 32 + nt * 16 + 4 * kq;
      float4 xv = *(const float4 *)xp;
      const float lo[4] = {acc[nt][0] + bs[nt].x, acc[nt][1] + bs[nt].y, acc[nt][2] + bs[nt].z, acc[nt][3] + bs[nt].w};
      const float hi[4] = {acc[nt + 2][0] + bs[nt + 2].x, acc[nt + 2][1] + bs[nt + 2].y, acc[nt + 2][2] + bs[nt + 2].z,
                           acc[nt + 2][3] + bs[nt + 2].w};
      xv.x += gate_dev(lo[0], hi[0]);
      xv.y += gate_dev(lo[1], hi[1]);
      xv.z += gate_dev(lo[2], hi[2]);
      xv.w += gate_dev(lo[3], hi[3]);
      *(float4 *)xp = xv;
    }
  }
}

// leaky -> dilated conv k3 32->32 -> leaky at audio rate (main.cpp:4339-4365) on the fp16 MFMA, for hop % 16 == 0
// and dil < hop (a tap that leaves the sequence lands in a guard frame, where x is zero): a 16-sample tile x one
// tap = one 16x16x32 MFMA per 16 output channels, K = the tap's 32 input channels. The stage is a pure stream:
// 128 B read + 128 B written per sample.
__global__ __launch_bounds__(256) void voc_dconv_mfma_kernel(const float *__restrict__ x, const float *__restrict__ w /*[3][32][32]*/,
                                                             const float *__restrict__ bias, const int *__restrict__ row_seq, int hop_shift,
                                                             int dil, int64_t P, float *__restrict__ y) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, m = lane & 15, fq = lane >> 4;
  half8 wf[3][2]; // A-role: lane (co = 16 ct + m, fq) holds w[tap][ci = 8 fq .. +7][co]
#pragma unroll
  for (int tap = 0; tap < 3; tap++)
#pragma unroll
    for (int ct = 0; ct < 2; ct++)
#pragma unroll
      for (int e = 0; e < 8; e++) wf[tap][ct][e] = (_Float16)w[(tap * 32 + 8 * fq + e) * 32 + ct * 16 + m];
  float4 bs[2];
#pragma unroll
  for (int ct = 0; ct < 2; ct++) bs[ct] = *(const float4 *)(bias + ct * 16 + 4 * fq);
#pragma unroll
  for (int t = 0; t < 4; t++) {
    const int64_t p0 = (int64_t)blockIdx.x * 256 + (wave * 4 + t) * 16, pos = p0 + m;
    if (p0 >= P) break;
    const bool live = row_seq[(int)(p0 >> hop_shift)] >= 0; // a tile lies inside one frame (hop = 1 << hop_shift: no 64-bit division)
    floatx4 acc[2] = {(floatx4){0.f, 0.f, 0.f, 0.f}, (floatx4){0.f, 0.f, 0.f, 0.f}};
    if (live) {
#pragma unroll
      for (int tap = 0; tap < 3; tap++) {
        int64_t q = pos + (int64_t)(tap - 1) * dil;
        q = q < 0 ? 0 : (q >= P ? P - 1 : q);
        const float4 a = *(const float4 *)(x + q * 32 + 8 * fq), b = *(const float4 *)(x + q * 32 + 8 * fq + 4);
        half8 xf;
        xf[0] = (_Float16)leaky02(a.x); xf[1] = (_Float16)leaky02(a.y); xf[2] = (_Float16)leaky02(a.z); xf[3] = (_Float16)leaky02(a.w);
        xf[4] = (_Float16)leaky02(b.x); xf[5] = (_Float16)leaky02(b.y); xf[6] = (_Float16)leaky02(b.z); xf[7] = (_Float16)leaky02(b.w);
#pragma unroll
        for (int ct = 0; ct < 2; ct++) acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[tap][ct], xf, acc[ct], 0, 0, 0);
      }
    }
#pragma unroll
    for (int ct = 0; ct < 2; ct++) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (live) v = make_float4(leaky02(acc[ct][0] + bs[ct].x), leaky02(acc[ct][1] + bs[ct].y), leaky02(acc[ct][2] + bs[ct].z),
                                leaky02(acc[ct][3] + bs[ct].w));
      *(float4 *)(y + pos * 32 + ct * 16 + 4 * fq) = v;
    }
  }
}

// leaky -> conv_post k7 32->1, no padding (main.cpp:4459-4478): audio[c][j] for j < Tm*256 - 6.
__global__ __launch_bounds__(256) void conv_post_kernel(const float *__restrict__ x, const float *__restrict__ w /*[7][32]*/,
                                                        const float *__restrict__ bias, const int *__restrict__ seq_start,
                                                        const int *__restrict__ seq_len, const int64_t *__restrict__ out_off,
                                                        float *__restrict__ audio) {
  const int s = blockIdx.y;
  const int64_t n = (int64_t)seq_len[s] * 256 - 6, j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  const float *xr = x + ((int64_t)seq_start[s] * 256 + j) * 32;
  float acc = 0.f;
  for (int k = 0; k < 7; k++)
    for (int ci = 0; ci < 32; ci++) acc = fmaf(r16(leaky02(xr[k * 32 + ci])), w[k * 32 + ci], acc);
  audio[out_off[s] + j] = acc + bias[0];
}

__global__ void voc_philox_kernel(float *__restrict__ x, int64_t n, uint64_t seed, uint32_t stream);

// ------------------------------------------------------------------------------------------------
struct VocLayout { // same convention as the diffusion stage's Layout (start % 8 == 0, guard rows, pad 128)
  int ns = 0, rows = 0;
  std::vector<int> start, len;
  DevBuf d_row_seq, d_row_t, d_start, d_len;
};

struct KpDev { float *in_w, *in_b, *rw[6], *rb[6], *kc_b, *bc_b; __half *kc_w, *bc_w; };
struct VocState {
  float *pre_w = nullptr, *pre_b = nullptr, *post_w = nullptr, *post_b = nullptr;
  KpDev kp[3];
  float *ct_w[3], *ct_b[3], *cb_w[3][4], *cb_b[3][4];
  std::vector<void *> owned;
  VocLayout lay;
  DevBuf melrows, nzrows, nzsrc, c0, c1, c2, cond16, kern, kbias, xa, xb, ybuf, offs, audio, melsrc;
  ~VocState() { for (void *p : owned) (void)hipFree(p); }
};
void voc_free(VocState *s) { delete s; }

// One-sided receptive field of an output sample, in mel frames, of the stack voc_load accepts (its kernel widths, strides and dilations are the
// constants below: a weight file with any other shape is rejected at load, so this number cannot drift from the loaded model).
//   audio path: conv_pre k7 at frame rate (3 frames); per res_stack at `hop` samples per frame after its transposed conv (kernel 2 x stride:
//   one input sample either side), four blocks of {dilated k3 conv (d = 1, 3, 9, 27), location-variable conv k3}: (1 + 3 + 9 + 27) + 4 samples;
//   conv_post k7 at audio rate; the kernel of a sample comes from the frame it lies in, and that frame's kernel predictor sees
//   input_conv k5 + 6 x k3 + kernel_conv k3 = 2 + 6 + 1 frames of mel either side.
constexpr int voc_receptive_frames() {
  const int strides[3] = {8, 8, 4}, dil[4] = {1, 3, 9, 27};
  double frames = 3.0; // conv_pre
  int hop = 1;
  for (int i = 0; i < 3; i++) {
    frames += 1.0 / hop; // transposed conv: one sample of its input rate
    hop *= strides[i];
    int samples = 0;
    for (int c = 0; c < 4; c++) samples += dil[c] + 1;
    frames += (double)samples / hop;
  }
  frames += 3.0 / hop;       // conv_post
  const int predictor = 2 + 6 + 1;
  return (int)(frames + 0.999) + predictor;
}
static_assert(voc_receptive_frames() <= TTS_VOC_CHUNK_HALO, "tts_vocoder_chunk's halo no longer covers the vocoder's receptive field");
int voc_halo_frames() { return TTS_VOC_CHUNK_HALO; }

namespace {
struct VLoader {
  tts_ctx *ctx; VocState *st; const WeightFile &wf; std::map<std::string, bool> used; std::mutex mu; // the three res stacks load on their own threads
  const HostTensor *get(const std::string &name, int64_t nelem) {
    auto it = wf.t.find(name);
    if (it == wf.t.end()) { fail(ctx, TTS_ERR_FORMAT, "tensor '%s' missing from vocoder model file", name.c_str()); return nullptr; }
    if (it->second.nelem() != nelem) { fail(ctx, TTS_ERR_FORMAT, "tensor '%s' has wrong size in model file", name.c_str()); return nullptr; }
    { std::lock_guard<std::mutex> lk(mu); used[name] = true; }
    return &it->second;
  }
  template <class T> int put(const std::vector<T> &h, T **dst) {
    void *p = nullptr;
    TTS_HIP(ctx, hipMalloc(&p, h.size() * sizeof(T)));
    { std::lock_guard<std::mutex> lk(mu); st->owned.push_back(p); }
    TTS_HIP(ctx, PinnedPool::copy_now(p, h.data(), h.size() * sizeof(T), ctx->load_stream)); // not the legacy stream: this load may run beside a capturing AR stage
    *dst = (T *)p;
    return TTS_OK;
  }
  int f32(const std::string &name, int64_t n, float **dst) {
    const HostTensor *t = get(name, n);
    return t ? put(t->data, dst) : TTS_ERR_FORMAT;
  }
  // conv weight w[(co*cin+ci)*k+tap] -> [tap][ci][co] f32 holding fp16-rounded values
  int conv_kcc(const std::string &name, int cout, int cin, int k, bool round16, float **dst) {
    const HostTensor *t = get(name, (int64_t)cout * cin * k);
    if (!t) return TTS_ERR_FORMAT;
    std::vector<float> h((size_t)k * cin * cout);
    for (int co = 0; co < cout; co++)
      for (int ci = 0; ci < cin; ci++)
        for (int tap = 0; tap < k; tap++) {
          float v = t->data[((size_t)co * cin + ci) * k + tap];
          h[((size_t)tap * cin + ci) * cout + co] = round16 ? __half2float(__float2half_rn(v)) : v;
        }
    return put(h, dst);
  }
  // -> fp16 [cout][tap*cin + ci] for the MFMA GEMM
  // (perm: device output channel n holds the file's channel (*perm)[n])
  int conv_gemm(const std::string &name, int cout, int cin, int k, __half **dst, const std::vector<int> *perm = nullptr) {
    const HostTensor *t = get(name, (int64_t)cout * cin * k);
    if (!t) return TTS_ERR_FORMAT;
    std::vector<__half> h((size_t)cout * k * cin);
    for (int n = 0; n < cout; n++) {
      const int co = perm ? (*perm)[n] : n;
      for (int ci = 0; ci < cin; ci++)
        for (int tap = 0; tap < k; tap++)
          h[(size_t)n * k * cin + (size_t)tap * cin + ci] = __float2half_rn(t->data[((size_t)co * cin + ci) * k + tap]);
    }
    return put(h, dst);
  }
};
} // namespace

int voc_load(tts_ctx *ctx, const char *path) {
  WeightFile wf;
  std::string err;
  int rc = read_weight_file(path, wf, err);
  if (rc != TTS_OK) return fail(ctx, rc, "vocoder_model_load: %s", err.c_str());
  std::unique_ptr<VocState> st(new VocState());
  VLoader ld{ctx, st.get(), wf, {}};
#define R(x) do { int _r = (x); if (_r) return _r; } while (0)
  const int strides[3] = {8, 8, 4};
  auto load_stack = [&](int i) -> int { // one res stack (kernel predictor incl. its 4.7 M-weight kernel_conv, transposed conv, conv blocks)
    std::string rs = "res_stack." + std::to_string(i) + ".", kp = rs + "kernel_predictor.";
    KpDev &k = st->kp[i];
    R(ld.conv_kcc(kp + "input_conv.0.weight", 64, 100, 5, true, &k.in_w));
    R(ld.f32(kp + "input_conv.0.bias", 64, &k.in_b));
    for (int c = 0; c < 3; c++)
      for (int j = 0; j < 2; j++) {
        std::string p = kp + "residual_convs." + std::to_string(c) + "." + (j ? "3" : "1");
        R(ld.conv_kcc(p + ".weight", 64, 64, 3, true, &k.rw[c * 2 + j]));
        R(ld.f32(p + ".bias", 64, &k.rb[c * 2 + j]));
      }
    { // kernel_conv: output channels permuted into lvc_col order (the LVC kernels read MFMA fragments straight from its output)
      std::vector<int> perm(24576);
      for (int l = 0; l < 4; l++)
        for (int i2 = 0; i2 < 32; i2++)
          for (int ch = 0; ch < 64; ch++)
            for (int tap = 0; tap < 3; tap++) perm[l * 6144 + lvc_col(tap, i2, ch)] = ((l * 32 + i2) * 64 + ch) * 3 + tap;
      R(ld.conv_gemm(kp + "kernel_conv.weight", 24576, 64, 3, &k.kc_w, &perm));
      const HostTensor *tb = ld.get(kp + "kernel_conv.bias", 24576);
      if (!tb) return TTS_ERR_FORMAT;
      std::vector<float> hb(24576);
      for (int n = 0; n < 24576; n++) hb[n] = tb->data[perm[n]];
      R(ld.put(hb, &k.kc_b));
    }
    R(ld.conv_gemm(kp + "bias_conv.weight", 256, 64, 3, &k.bc_w));
    R(ld.f32(kp + "bias_conv.bias", 256, &k.bc_b));
    { // ConvTranspose1d: file layout ne=[K,Cout,Cin] -> w[(ci*32+co)*K + k]; device [k][ci][co], F32
      const int K = 2 * strides[i];
      const HostTensor *t = ld.get(rs + "convt_pre.1.weight", (int64_t)32 * 32 * K);
      if (!t) return TTS_ERR_FORMAT;
      std::vector<float> h((size_t)K * 32 * 32);
      for (int ci = 0; ci < 32; ci++)
        for (int co = 0; co < 32; co++)
          for (int kk = 0; kk < K; kk++) h[((size_t)kk * 32 + ci) * 32 + co] = t->data[((size_t)ci * 32 + co) * K + kk];
      R(ld.put(h, &st->ct_w[i]));
      R(ld.f32(rs + "convt_pre.1.bias", 32, &st->ct_b[i]));
    }
    for (int c = 0; c < 4; c++) {
      std::string p = rs + "conv_blocks." + std::to_string(c) + ".1";
      R(ld.conv_kcc(p + ".weight", 32, 32, 3, true, &st->cb_w[i][c]));
      R(ld.f32(p + ".bias", 32, &st->cb_b[i][c]));
    }
    return TTS_OK;
  };
  auto load_ends = [&]() -> int {
  R(ld.conv_kcc("conv_pre.weight", 32, 64, 7, true, &st->pre_w));
  R(ld.f32("conv_pre.bias", 32, &st->pre_b));
  { // conv_post.1.weight ne=[7,32]: w[ci*7 + k] -> [k][ci], fp16-rounded
    const HostTensor *t = ld.get("conv_post.1.weight", 7 * 32);
    if (!t) return TTS_ERR_FORMAT;
    std::vector<float> h(7 * 32);
    for (int ci = 0; ci < 32; ci++)
      for (int k = 0; k < 7; k++) h[k * 32 + ci] = __half2float(__float2half_rn(t->data[ci * 7 + k]));
    R(ld.put(h, &st->post_w));
    R(ld.f32("conv_post.1.bias", 1, &st->post_b));
  }
  return TTS_OK;
  };
#undef R
  if (int r = run_parallel(ctx, 4, [&](int j) { return j < 3 ? load_stack(j) : load_ends(); })) return r;
  for (auto &kv : wf.t)
    if (!ld.used.count(kv.first)) return fail(ctx, TTS_ERR_FORMAT, "unknown tensor '%s' in model file", kv.first.c_str());
  if (ctx->voc) voc_free(ctx->voc);
  ctx->voc = st.release();
  return TTS_OK;
}

#define CHECK(x) do { int _r = (x); if (_r) return _r; } while (0)

static int conv(tts_ctx *ctx, VocState *st, const float *x, const float *w, const float *bias, float *y, const float *resid,
                int64_t P, int Cin, int Cout, int K, int dil, int pad, int hop, int pre_leaky, int post_leaky, int reflect) {
  ConvArgs a{x, w, bias, y, resid, st->lay.d_row_seq.as<int>(), st->lay.d_start.as<int>(), st->lay.d_len.as<int>(),
             (int)P, Cin, Cout, K, dil, pad, hop, pre_leaky, post_leaky, reflect};
  const int per = 256 / Cout;
  ProfScope ps(ctx, "voc_conv");
  conv_direct_kernel<<<(int)((P + per - 1) / per), 256, 0, ctx->stream>>>(a);
  TTS_HIP(ctx, hipGetLastError());
  return TTS_OK;
}

__global__ void voc_philox_kernel(float *__restrict__ x, int64_t n, uint64_t seed, uint32_t stream) {
  // Philox4x32-10 + Box-Muller, stream tag distinct from the diffusion stage's
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
  uint32_t c0 = (uint32_t)(i >> 1), c1 = 0x766f63u, c2 = stream, c3 = 0x7716u, k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
  for (int r = 0; r < 10; r++) {
    uint32_t hi0 = __umulhi(M0, c0), lo0 = M0 * c0, hi1 = __umulhi(M1, c2), lo1 = M1 * c2;
    uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  const float u1 = ((float)c0 + 1.0f) * 2.3283064365386963e-10f, u2 = (float)c1 * 2.3283064365386963e-10f;
  const float rad = sqrtf(-2.0f * logf(u1));
  float sn, cs;
  sincosf(6.283185307179586f * u2, &sn, &cs);
  x[i] = (i & 1) ? rad * sn : rad * cs;
}

int voc_run(tts_ctx *ctx, const float *mel, const int32_t *frames, int B, const float *noise, int noise_mode, float *audio) {
  VocState *st = ctx->voc;
  if (!st) return fail(ctx, TTS_ERR_STATE, "vocoder model not loaded");
  if (!mel || !frames || !audio || B < 1) return fail(ctx, TTS_ERR_ARG, "tts_vocoder: bad argument");
  // frame layout over Tm = T + 10 frames per candidate
  VocLayout &lay = st->lay;
  lay.ns = B;
  lay.len.resize(B); lay.start.resize(B);
  int r = 8;
  std::vector<int64_t> mel_off(B), nz_off(B), out_off(B);
  int64_t mel_total = 0, nz_total = 0, out_total = 0;
  for (int c = 0; c < B; c++) {
    if (frames[c] < 1) return fail(ctx, TTS_ERR_ARG, "mel of candidate %d is empty", c);
    lay.len[c] = frames[c] + 10;
    lay.start[c] = r;
    r = (r + lay.len[c] + 1 + 7) & ~7;
    mel_off[c] = mel_total; mel_total += (int64_t)100 * frames[c];
    nz_off[c] = nz_total; nz_total += (int64_t)64 * lay.len[c];
    out_off[c] = out_total; out_total += (int64_t)lay.len[c] * 256 - 6;
  }
  lay.rows = (r + 127) & ~127;
  const int R = lay.rows;
  {
    std::vector<int> rs(R, -1), rt(R, 0);
    for (int c = 0; c < B; c++)
      for (int t = 0; t < lay.len[c]; t++) { rs[lay.start[c] + t] = c; rt[lay.start[c] + t] = t; }
    TTS_HIP(ctx, lay.d_row_seq.reserve(R * 4)); TTS_HIP(ctx, lay.d_row_t.reserve(R * 4));
    TTS_HIP(ctx, lay.d_start.reserve(B * 4)); TTS_HIP(ctx, lay.d_len.reserve(B * 4));
    TTS_HIP(ctx, hipMemcpy(lay.d_row_seq.p, rs.data(), R * 4, hipMemcpyHostToDevice));
    TTS_HIP(ctx, hipMemcpy(lay.d_row_t.p, rt.data(), R * 4, hipMemcpyHostToDevice));
    TTS_HIP(ctx, hipMemcpy(lay.d_start.p, lay.start.data(), B * 4, hipMemcpyHostToDevice));
    TTS_HIP(ctx, hipMemcpy(lay.d_len.p, lay.len.data(), B * 4, hipMemcpyHostToDevice));
  }
  const int *d_rs = lay.d_row_seq.as<int>(), *d_rt = lay.d_row_t.as<int>(), *d_st = lay.d_start.as<int>(), *d_ln = lay.d_len.as<int>();
  TTS_HIP(ctx, st->offs.reserve(3 * B * 8));
  int64_t *d_mel_off = st->offs.as<int64_t>(), *d_nz_off = d_mel_off + B, *d_out_off = d_nz_off + B;
  TTS_HIP(ctx, hipMemcpy(d_mel_off, mel_off.data(), B * 8, hipMemcpyHostToDevice));
  TTS_HIP(ctx, hipMemcpy(d_nz_off, nz_off.data(), B * 8, hipMemcpyHostToDevice));
  TTS_HIP(ctx, hipMemcpy(d_out_off, out_off.data(), B * 8, hipMemcpyHostToDevice));
  // buffers
  const int64_t Pmax = (int64_t)R * 256;
  TTS_HIP(ctx, st->melsrc.reserve(mel_total * 4)); TTS_HIP(ctx, st->nzsrc.reserve(nz_total * 4));
  TTS_HIP(ctx, st->melrows.reserve((size_t)R * 100 * 4)); TTS_HIP(ctx, st->nzrows.reserve((size_t)R * 64 * 4));
  TTS_HIP(ctx, st->c0.reserve((size_t)R * 64 * 4)); TTS_HIP(ctx, st->c1.reserve((size_t)R * 64 * 4)); TTS_HIP(ctx, st->c2.reserve((size_t)R * 64 * 4));
  {
    size_t old = st->cond16.cap;
    TTS_HIP(ctx, st->cond16.reserve((size_t)(R + 2) * 64 * 2));
    if (st->cond16.cap != old) TTS_HIP(ctx, hipMemset(st->cond16.p, 0, st->cond16.cap));
  }
  TTS_HIP(ctx, st->kern.reserve((size_t)R * 24576 * 4)); TTS_HIP(ctx, st->kbias.reserve((size_t)R * 256 * 4));
  TTS_HIP(ctx, st->xa.reserve(Pmax * 32 * 4)); TTS_HIP(ctx, st->xb.reserve(Pmax * 32 * 4)); TTS_HIP(ctx, st->ybuf.reserve(Pmax * 32 * 4));
  TTS_HIP(ctx, st->audio.reserve(out_total * 4));
  TTS_HIP(ctx, hipMemcpyAsync(st->melsrc.p, mel, mel_total * 4, hipMemcpyHostToDevice, ctx->stream));
  // noise (main.cpp:6058-6059): [64][Tm] per candidate
  std::vector<float> hn;
  if (noise) {
    TTS_HIP(ctx, hipMemcpyAsync(st->nzsrc.p, noise, nz_total * 4, hipMemcpyHostToDevice, ctx->stream));
  } else if (noise_mode == TTS_NOISE_REFERENCE) {
    hn.resize(nz_total);
    rng_normal_fill(ctx, hn.data(), nz_total);
    TTS_HIP(ctx, hipMemcpyAsync(st->nzsrc.p, hn.data(), nz_total * 4, hipMemcpyHostToDevice, ctx->stream));
  } else {
    for (int c = 0; c < B; c++) {
      int64_t n = (int64_t)64 * lay.len[c];
      voc_philox_kernel<<<(int)((n + 255) / 256), 256, 0, ctx->stream>>>(st->nzsrc.as<float>() + nz_off[c], n, ctx->seed_value, (uint32_t)(shard_base(ctx) + c)); // stream = global candidate id
    }
  }
  voc_mel_rows_kernel<<<R, 128, 0, ctx->stream>>>(st->melsrc.as<float>(), d_mel_off, d_rs, d_rt, d_ln, st->melrows.as<float>());
  voc_noise_rows_kernel<<<R, 64, 0, ctx->stream>>>(st->nzsrc.as<float>(), d_nz_off, d_rs, d_rt, d_ln, st->nzrows.as<float>());
  float *xa = st->xa.as<float>(), *xb = st->xb.as<float>(), *yb = st->ybuf.as<float>();
  // conv_pre: reflect pad 3, k7 64->32 (main.cpp:4114-4130)
  CHECK(conv(ctx, st, st->nzrows.as<float>(), st->pre_w, st->pre_b, xa, nullptr, R, 64, 32, 7, 1, 3, 1, 0, 0, 1));
  const int strides[3] = {8, 8, 4};
  int hop = 1;
  float *cur = xa, *nxt = xb;
  for (int i = 0; i < 3; i++) {
    const KpDev &k = st->kp[i];
    const int s = strides[i], hop_out = hop * s;
    const int64_t Pout = (int64_t)R * hop_out;
    {
      ProfScope ps(ctx, "voc_convt");
      convt_kernel<<<(int)((Pout + 7) / 8), 256, 0, ctx->stream>>>(cur, st->ct_w[i], st->ct_b[i], d_rs, d_st, d_ln, hop, s, Pout, nxt);
      TTS_HIP(ctx, hipGetLastError());
    }
    std::swap(cur, nxt);
    hop = hop_out;
    // kernel predictor on the padded mel (main.cpp:4169-4324)
    float *c0 = st->c0.as<float>(), *c1 = st->c1.as<float>(), *c2 = st->c2.as<float>();
    CHECK(conv(ctx, st, st->melrows.as<float>(), k.in_w, k.in_b, c0, nullptr, R, 100, 64, 5, 1, 2, 1, 0, 1, 0));
    for (int c = 0; c < 3; c++) {
      CHECK(conv(ctx, st, c0, k.rw[c * 2], k.rb[c * 2], c1, nullptr, R, 64, 64, 3, 1, 1, 1, 0, 1, 0));
      CHECK(conv(ctx, st, c1, k.rw[c * 2 + 1], k.rb[c * 2 + 1], c2, c0, R, 64, 64, 3, 1, 1, 1, 0, 1, 0));
      std::swap(c0, c2);
    }
    voc_cond_f16_kernel<<<R, 64, 0, ctx->stream>>>(c0, d_rs, st->cond16.as<__half>() + 64);
    {
      GemmArgs g{};
      for (int q = 0; q < 3; q++) { g.A[q] = st->cond16.as<__half>() + 64; g.row_off[q] = q - 1; }
      g.nseg = 3; g.kseg = 64; g.lda = 64; g.W = k.kc_w; g.M = R; g.N = 24576; g.bias = k.kc_b; g.row_seq = d_rs;
      g.mode = GEMM_OUT_F32; g.outF = st->kern.as<float>(); g.ldo = 24576; g.resid = nullptr;
      { ProfScope ps(ctx, "voc_kernel_gemm"); TTS_HIP(ctx, launch_gemm_f16(g, ctx->stream)); }
      g.W = k.bc_w; g.N = 256; g.bias = k.bc_b; g.outF = st->kbias.as<float>(); g.ldo = 256;
      { ProfScope ps(ctx, "voc_kernel_gemm"); TTS_HIP(ctx, launch_gemm_f16(g, ctx->stream)); }
    }
    const int dil[4] = {1, 3, 9, 27};
    for (int c = 0; c < 4; c++) {
      // leaky -> dilated conv k3 32->32 -> leaky (main.cpp:4339-4365)
      if (hop == 64 || hop == 256) { // audio-rate stages: matrix-pipe kernels (frames are whole 16-sample tiles, dil < hop)
        const int64_t P = (int64_t)R * hop;
        { ProfScope ps(ctx, "voc_conv");
          voc_dconv_mfma_kernel<<<(int)((P + 255) / 256), 256, 0, ctx->stream>>>(cur, st->cb_w[i][c], st->cb_b[i][c], d_rs, hop == 64 ? 6 : 8, dil[c], P, yb); }
        ProfScope ps(ctx, "voc_lvc");
        lvc_mfma_kernel<<<R, 256, 0, ctx->stream>>>(yb, st->kern.as<float>(), st->kbias.as<float>(), d_rs, hop, c, cur);
        TTS_HIP(ctx, hipGetLastError());
        continue;
      }
      CHECK(conv(ctx, st, cur, st->cb_w[i][c], st->cb_b[i][c], yb, nullptr, (int64_t)R * hop, 32, 32, 3, dil[c], dil[c], hop, 1, 1, 0));
      ProfScope ps(ctx, "voc_lvc");
      dim3 grid(std::max(1, hop / 64), R);
      lvc_gate_kernel<<<grid, 256, 0, ctx->stream>>>(yb, st->kern.as<float>(), st->kbias.as<float>(), d_rs, hop, c, cur);
      TTS_HIP(ctx, hipGetLastError());
    }
  }
  {
    int max_len = *std::max_element(lay.len.begin(), lay.len.end());
    dim3 grid((int)(((int64_t)max_len * 256 + 255) / 256), B);
    ProfScope ps(ctx, "voc_conv");
    conv_post_kernel<<<grid, 256, 0, ctx->stream>>>(cur, st->post_w, st->post_b, d_st, d_ln, d_out_off, st->audio.as<float>());
    TTS_HIP(ctx, hipGetLastError());
  }
  TTS_HIP(ctx, hipMemcpyAsync(audio, st->audio.p, out_total * 4, hipMemcpyDeviceToHost, ctx->stream));
  TTS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return TTS_OK;
}

} // namespace tts

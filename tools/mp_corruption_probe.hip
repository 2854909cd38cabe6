// Developer probe, independent of the engine: does a trivial matrix-vector kernel give the same answer every time while OTHER PROCESSES use the GPU?
//   tools/bin/mp_corruption_probe <seconds> [tag] [matrices] [engine]     (run two or three copies at once, or beside an engine process; a 4th argument: the engine's kernel)
// matrices > 1: the kernel walks through that many different 8 MB matrices, so every launch reads COLD weights (as the engine's time MLP does once per call)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
// out[n] = sum_k x[k] * W[n][k]: one wave per output, 16-byte loads, shuffle reduction (the shape of the engine's time-MLP kernel)
__global__ __launch_bounds__(256) void matvec(const float *__restrict__ x, const float *__restrict__ W, int K, int N, float *__restrict__ out) {
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (n >= N) return;
  float acc = 0.f;
  for (int k = lane * 4; k < K; k += 256) {
    const float4 w = *(const float4 *)(W + (size_t)n * K + k), xv = *(const float4 *)(x + k);
    acc = fmaf(xv.x, w.x, acc); acc = fmaf(xv.y, w.y, acc); acc = fmaf(xv.z, w.z, acc); acc = fmaf(xv.w, w.w, acc);
  }
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if (lane == 0) out[n] = acc;
}
// the engine's time-MLP kernel, verbatim in structure (tortoise.cpp_amd/csrc/diffusion.hip: linear_nk_kernel): 8 rows per pass (clamped to the last row), 16 waves x 2
// columns per workgroup, permlane / DPP wave reduction, bias, optional SiLU
template <int CTRL> __device__ __forceinline__ float p_dpp(float x) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, false)); }
__device__ __forceinline__ float p_wave_sum(float x) {
  auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  x = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  x = __uint_as_float(b[0]) + __uint_as_float(b[1]);
  x += p_dpp<0x128>(x); x += p_dpp<0x124>(x); x += p_dpp<0x122>(x); x += p_dpp<0x121>(x);
  return x;
}
// variants (4th argument = a bit mask): 1 = one row per pass instead of 8 clamped ones, 2 = no bias / activation code at all, 4 = __shfl_xor butterfly
// instead of the permlane / DPP reduction, 8 = act is always 0 (the SiLU code is compiled in but never run)
template <int NR, bool TAIL, bool DPP>
__global__ __launch_bounds__(1024) void engine_matvec(const float *__restrict__ x, int ldx, int rows, const float *__restrict__ W, int K, int N,
                                                      const float *__restrict__ b, float *__restrict__ out, int ldo, int act) {
  const int lane = threadIdx.x & 63;
  for (int half = 0; half < 2; half++) {
    const int n = blockIdx.x * 32 + half * 16 + (threadIdx.x >> 6);
    if (n >= N) return;
    const float *wr = W + (size_t)n * K;
    for (int r0 = 0; r0 < rows; r0 += NR) {
      float acc[NR];
#pragma unroll
      for (int i = 0; i < NR; i++) acc[i] = 0.f;
      for (int k = lane * 4; k < K; k += 256) {
        const float4 w = *(const float4 *)(wr + k);
#pragma unroll
        for (int i = 0; i < NR; i++) {
          const float4 xv = *(const float4 *)(x + (size_t)min(r0 + i, rows - 1) * ldx + k);
          acc[i] = fmaf(xv.x, w.x, acc[i]); acc[i] = fmaf(xv.y, w.y, acc[i]);
          acc[i] = fmaf(xv.z, w.z, acc[i]); acc[i] = fmaf(xv.w, w.w, acc[i]);
        }
      }
#pragma unroll
      for (int i = 0; i < NR; i++) {
        float v = acc[i];
        if (DPP) v = p_wave_sum(v);
        else for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if (lane == 0 && r0 + i < rows) {
          if (TAIL) {
            v += b ? b[n] : 0.f;
            if (act) v = v / (1.f + expf(-v));
          }
          out[(size_t)(r0 + i) * ldo + n] = v;
        }
      }
    }
  }
}
// variant 16: the same contraction restructured — a lane keeps its 16 weights of the column in registers, rows are walked one at a time (no clamped row group)
__global__ __launch_bounds__(1024) void engine_matvec_wreg(const float *__restrict__ x, int ldx, int rows, const float *__restrict__ W, int K /*1024*/, int N,
                                                           const float *__restrict__ b, float *__restrict__ out, int ldo, int act) {
  const int lane = threadIdx.x & 63;
  for (int half = 0; half < 2; half++) {
    const int n = blockIdx.x * 32 + half * 16 + (threadIdx.x >> 6);
    if (n >= N) return;
    const float *wr = W + (size_t)n * K + lane * 4;
    float4 w[4];
#pragma unroll
    for (int j = 0; j < 4; j++) w[j] = *(const float4 *)(wr + j * 256);
    for (int r = 0; r < rows; r++) {
      const float *xr = x + (size_t)r * ldx + lane * 4;
      float acc = 0.f;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const float4 xv = *(const float4 *)(xr + j * 256);
        acc = fmaf(xv.x, w[j].x, acc); acc = fmaf(xv.y, w[j].y, acc); acc = fmaf(xv.z, w[j].z, acc); acc = fmaf(xv.w, w[j].w, acc);
      }
      float v = p_wave_sum(acc);
      if (lane == 0) {
        v += b ? b[n] : 0.f;
        if (act) v = v / (1.f + expf(-v));
        out[(size_t)r * ldo + n] = v;
      }
    }
  }
}
// variant 32: the 8-row kernel with every multiply-add as a single v_fmac_f32 (no packed f32 FMA)
__global__ __launch_bounds__(1024) void engine_matvec_nopk(const float *__restrict__ x, int ldx, int rows, const float *__restrict__ W, int K, int N,
                                                           const float *__restrict__ b, float *__restrict__ out, int ldo, int act) {
  const int lane = threadIdx.x & 63;
  for (int half = 0; half < 2; half++) {
    const int n = blockIdx.x * 32 + half * 16 + (threadIdx.x >> 6);
    if (n >= N) return;
    const float *wr = W + (size_t)n * K;
    for (int r0 = 0; r0 < rows; r0 += 8) {
      float acc[8];
#pragma unroll
      for (int i = 0; i < 8; i++) acc[i] = 0.f;
      for (int k = lane * 4; k < K; k += 256) {
        const float4 w = *(const float4 *)(wr + k);
#pragma unroll
        for (int i = 0; i < 8; i++) {
          const float4 xv = *(const float4 *)(x + (size_t)min(r0 + i, rows - 1) * ldx + k);
          asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[i]) : "v"(xv.x), "v"(w.x));
          asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[i]) : "v"(xv.y), "v"(w.y));
          asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[i]) : "v"(xv.z), "v"(w.z));
          asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[i]) : "v"(xv.w), "v"(w.w));
        }
      }
#pragma unroll
      for (int i = 0; i < 8; i++) {
        float v = p_wave_sum(acc[i]);
        if (lane == 0 && r0 + i < rows) out[(size_t)(r0 + i) * ldo + n] = v;
      }
    }
  }
}
__global__ void scale(float *x, int n, float s) { int i = blockIdx.x * 256 + threadIdx.x; if (i < n) x[i] *= s; }
int main(int argc, char **argv) {
  const double secs = argc > 1 ? atof(argv[1]) : 10.0;
  const char *tag = argc > 2 ? argv[2] : "p";
  const int K = 1024, N = 2048, NW = argc > 3 ? atoi(argv[3]) : 1;
  std::vector<float> hw((size_t)N * K), hx(K);
  unsigned s = 12345u + (unsigned)tag[0];
  for (auto &v : hw) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 65536.f - 0.5f; }
  for (auto &v : hx) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 65536.f - 0.5f; }
  float *W, *x, *out, *big; CK(hipMalloc(&W, (size_t)NW * hw.size() * 4)); CK(hipMalloc(&x, K * 4)); CK(hipMalloc(&out, N * 4)); CK(hipMalloc(&big, 256u << 20));
  for (int m = 0; m < NW; m++) { hw[m] += 1.0f; CK(hipMemcpy(W + (size_t)m * hw.size(), hw.data(), hw.size() * 4, hipMemcpyHostToDevice)); }
  hipStream_t st; CK(hipStreamCreate(&st));
  std::vector<std::vector<float>> refs(2 * NW, std::vector<float>(N));
  std::vector<float> got(N);
  long iters = 0, bad_iters = 0, bad_vals = 0;
  const auto t0 = std::chrono::steady_clock::now();
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs) {
    CK(hipMemcpyAsync(x, hx.data(), K * 4, hipMemcpyHostToDevice, st));
    CK(hipStreamSynchronize(st));
    const int m = (int)((iters / 2) % NW) * 1 + 0; // (two consecutive iterations per matrix: act = 0 and act = 1 of the engine kernel)
    std::vector<float> &ref = refs[(m * 2 + (iters & 1)) % refs.size()];
    if (argc > 4) { // argv[4]: the engine's kernel, variant mask
      const int vm = atoi(argv[4]), act = (vm & 8) ? 0 : (int)(iters & 1);
      const float *Wm = W + (size_t)m * N * K;
#define EM(NR_, TAIL_, DPP_) engine_matvec<NR_, TAIL_, DPP_><<<N / 32, 1024, 0, st>>>(x, K, 1, Wm, K, N, W, out, N, act)
      if (vm & 16) engine_matvec_wreg<<<N / 32, 1024, 0, st>>>(x, K, 1, Wm, K, N, W, out, N, act);
      else if (vm & 32) engine_matvec_nopk<<<N / 32, 1024, 0, st>>>(x, K, 1, Wm, K, N, W, out, N, act);
      else switch (vm & 7) {
        case 0: EM(8, true, true); break;   case 1: EM(1, true, true); break;
        case 2: EM(8, false, true); break;  case 3: EM(1, false, true); break;
        case 4: EM(8, true, false); break;  case 5: EM(1, true, false); break;
        case 6: EM(8, false, false); break; default: EM(1, false, false); break;
      }
    }
    else matvec<<<N / 4, 256, 0, st>>>(x, W + (size_t)m * N * K, K, N, out);
    CK(hipMemcpyAsync(got.data(), out, N * 4, hipMemcpyDeviceToHost, st));
    CK(hipStreamSynchronize(st));
    if (iters < 2 * NW) ref = got;
    else {
      int nb = 0, first = -1;
      for (int i = 0; i < N; i++) if (memcmp(&got[i], &ref[i], 4)) { if (first < 0) first = i; nb++; }
      if (nb) { bad_iters++; bad_vals += nb; if (bad_iters <= 3) printf("[%s] iteration %ld (act %d): %d of %d outputs differ from the first iteration, first at %d (%.9g vs %.9g)\n", tag, iters, (int)(iters & 1), nb, N, first, got[first], ref[first]); }
    }
    // some bandwidth traffic of our own between the probes (the engine's forwards do the same)
    scale<<<(64 << 20) / 256, 256, 0, st>>>(big, 64 << 20, 1.0001f);
    iters++;
  }
  CK(hipStreamSynchronize(st));
  printf("[%s] %ld iterations, %ld with wrong outputs (%ld values)\n", tag, iters, bad_iters, bad_vals);
  return 0;
}

// Round 5: the CONSUMER-SIDE ceiling of a GEMM work-group on gfx950 — fragment reads from LDS + MFMAs only, operands already resident in LDS, no global
// loads, no epilogue — for the wave tile of the product kernels (64 x 64, four work-groups per CU, two barriers per K step) and for the 128 x 64 wave tile of the decoupled
// loader / consumer structure (four consumer waves, one or two work-groups per CU, zero or one barrier per K step). Whatever ring / loader / persistent-tile machinery is built
// around the consumers, it cannot run faster than this. Random and zero-filled fp16 operands (the chip is power-limited on real data: MI355X_MICROARCH.md, DVFS).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -I tortoise.cpp_amd/csrc tools/r5/mfma_lds_bound.hip -o tools/bin/mfma_lds_bound
#include "gemm_f16.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace tts;

// MI 16-row blocks x 4 16-column blocks per wave, 2 x 2 waves -> work-group tile (32 MI) x 128, K step 64, NS LDS stages walked round-robin (the stage address depends on the
// step, so the fragment reads stay inside the loop). NB barriers per K step (2 = the product kernel's single-stage loop, 1 = a ring with one rendezvous per step, 0 = none).
template <int MI, int NS, int NB, int WGS>
__global__ __launch_bounds__(256, WGS) void consumer_kernel(const __half *__restrict__ src, float *__restrict__ out, int ksteps) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int AB = 32 * MI * 128, BB = 128 * 128, ST = AB + BB; // bytes per stage
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < NS * ST / 16; i += 256) ((uint4 *)smem)[i] = ((const uint4 *)src)[(i + blockIdx.x * 977) & 65535];
  __syncthreads();
  const int wm = wave >> 1, wn = wave & 1, fr = lane & 15, fq = lane >> 4;
  floatx4 acc[MI][4];
#pragma unroll
  for (int i = 0; i < MI; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = (floatx4){0.f, 0.f, 0.f, 0.f};
  int stage = 0;
  for (int k = 0; k < ksteps; k++) {
    const char *sa = smem + stage * ST, *sb = sa + AB;
    stage = stage + 1 == NS ? 0 : stage + 1;
    if (NB >= 1) __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {
      half8 af[MI], bf[4];
#pragma unroll
      for (int i = 0; i < MI; i++) af[i] = *(const half8 *)(sa + lds_off((wm + 2 * i) * 16 + fr, ks * 4 + fq));
#pragma unroll
      for (int j = 0; j < 4; j++) bf[j] = *(const half8 *)(sb + lds_off(wn * 64 + j * 16 + fr, ks * 4 + fq));
#pragma unroll
      for (int i = 0; i < MI; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[j], af[i], acc[i][j], 0, 0, 0);
    }
    if (NB >= 2) __builtin_amdgcn_s_barrier();
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MI; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  out[blockIdx.x * 256 + tid] = s;
}

template <int MI, int NS, int NB, int WGS>
static void run(const char *name, const __half *src, float *out, int ksteps) {
  constexpr int LDS = NS * (32 * MI * 128 + 128 * 128);
  (void)hipFuncSetAttribute((const void *)consumer_kernel<MI, NS, NB, WGS>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
  int occ = 0;
  (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, consumer_kernel<MI, NS, NB, WGS>, 256, LDS);
  const int grid = 256 * WGS;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  consumer_kernel<MI, NS, NB, WGS><<<grid, 256, LDS>>>(src, out, ksteps);
  float best = 1e30f;
  for (int rep = 0; rep < 5; rep++) {
    (void)hipEventRecord(e0);
    consumer_kernel<MI, NS, NB, WGS><<<grid, 256, LDS>>>(src, out, ksteps);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
  }
  const double flop = (double)grid * (32.0 * MI) * 128.0 * 64.0 * 2.0 * ksteps;
  const double lds_bytes = (double)grid * 4 * (MI + 4) * 2 * 1024.0 * ksteps; // fragment bytes read
  printf("  %-64s occupancy %d/CU  %8.1f us  %7.1f TF/s  (%.3f of 2.5 PF)   LDS fragment reads %6.1f B/clk/CU at 2.4 GHz\n", name, occ, best * 1e3, flop / (best * 1e-3) / 1e12,
         flop / (best * 1e-3) / 2.5e15, lds_bytes / (best * 1e-3) / 256.0 / 2.4e9);
  if (hipGetLastError() != hipSuccess) printf("  !! launch error\n");
}

int main() {
  const size_t n = 65536 * 8; // halves (1 MB)
  std::vector<__half> h(n);
  __half *src; float *out;
  (void)hipMalloc(&src, n * 2); (void)hipMalloc(&out, 1024 * 256 * 4);
  for (int fill = 0; fill < 2; fill++) {
    srand(1);
    for (size_t i = 0; i < n; i++) h[i] = __float2half(fill == 0 ? (float)rand() / RAND_MAX * 2.f - 1.f : 0.f);
    (void)hipMemcpy(src, h.data(), n * 2, hipMemcpyHostToDevice);
    printf("%s operands, 2048 K steps of 64 per work-group:\n", fill == 0 ? "uniform random [-1, 1)" : "zero-filled");
    const int K = 2048;
    run<4, 1, 2, 4>("64x64 wave tile, 4 WG/CU, 1 stage, 2 barriers/step (product loop)", src, out, K);
    run<4, 1, 0, 4>("64x64 wave tile, 4 WG/CU, no barrier", src, out, K);
    run<4, 3, 1, 2>("64x64 wave tile, 2 WG/CU, 3-stage ring, 1 barrier/step", src, out, K);
    run<8, 3, 1, 1>("128x64 wave tile, 1 WG/CU (4 consumer waves), 3-stage ring, 1 barrier/step", src, out, K);
    run<8, 3, 0, 1>("128x64 wave tile, 1 WG/CU, 3-stage ring, no barrier", src, out, K);
    run<8, 1, 1, 2>("128x64 wave tile, 2 WG/CU, 1 stage, 1 barrier/step", src, out, K);
    run<8, 1, 0, 2>("128x64 wave tile, 2 WG/CU, no barrier", src, out, K);
    run<8, 1, 2, 3>("128x64 wave tile, 3 WG/CU, 1 stage, 2 barriers/step", src, out, K);
  }
  return 0;
}

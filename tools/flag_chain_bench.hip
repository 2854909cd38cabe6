// Developer tool (experiment for the next round): "flag-chained" launches. A chain of 150 dependent launches, each streaming a 64 KB
// weight slab per workgroup (256 workgroups) and reading the 64 KB vector its predecessor wrote, (a) as one hipGraph on one stream — what the decode step
// does today — and (b) alternating between two streams with no stream dependency between neighbours: launch N+1 becomes resident while N runs, requests its
// slab, then waits on the 256 completion flags N's workgroups publish (plain stores after a release fence; no atomics), acquires, and goes on.
// Spins are bounded: a replay order in which a launch waits for one that has not started sets the error flag instead of hanging.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/flag_chain_bench.hip -o tools/bin/flag_chain_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int NWG = 256, NK = 150;

__global__ __launch_bounds__(256) void chained_kernel(const float4 *__restrict__ slab, const float *act_in, float *act_out, const unsigned *flags_prev,
                                                      unsigned *flags_mine, unsigned epoch, int *err) {
  const int tid = threadIdx.x, wg = blockIdx.x;
  float4 w[16]; // this launch's weights: independent of the predecessor, requested first
#pragma unroll
  for (int i = 0; i < 16; i++) w[i] = slab[((size_t)wg * 16 + i) * 256 + tid];
  if (flags_prev) {
    int spins = 0;
    while (__hip_atomic_load(&flags_prev[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > (1 << 18)) { *err = 1; break; }
    }
    __syncthreads();
    if (tid == 0) __threadfence(); // acquire
    __syncthreads();
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; i++) {
    const float4 x = *(const float4 *)(act_in + ((i * 256 + tid) * 4));
    s += x.x * w[i].x + x.y * w[i].y + x.z * w[i].z + x.w * w[i].w;
  }
  if (tid < 64) act_out[wg * 64 + tid] = s * 1e-3f + 1.f;
  if (flags_mine && tid == 0) { // same wave as the stores above
    __threadfence(); // release
    __hip_atomic_store(&flags_mine[wg], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

int main() {
  float4 *slab; float *a0, *a1; unsigned *flags; int *err;
  const size_t slab_elems = (size_t)NWG * 16 * 256; // 16 MB per launch; 8 distinct slabs rotate (128 MB: nothing stays in L2 anyway)
  CK(hipMalloc(&slab, 8 * slab_elems * sizeof(float4))); CK(hipMemset(slab, 0, 8 * slab_elems * sizeof(float4)));
  CK(hipMalloc(&a0, 65536)); CK(hipMalloc(&a1, 65536)); CK(hipMemset(a0, 0, 65536)); CK(hipMemset(a1, 0, 65536));
  CK(hipMalloc(&flags, (size_t)NK * NWG * 4)); CK(hipMalloc(&err, 4)); CK(hipMemset(err, 0, 4));
  hipStream_t s0, s1; CK(hipStreamCreate(&s0)); CK(hipStreamCreate(&s1));
  hipEvent_t e0, e1, ef, ej; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreateWithFlags(&ef, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ej, hipEventDisableTiming));
  auto enqueue = [&](bool chained) {
    CK(hipMemsetAsync(flags, 0, (size_t)NK * NWG * 4, s0));
    if (chained) { CK(hipEventRecord(ef, s0)); CK(hipStreamWaitEvent(s1, ef, 0)); }
    for (int k = 0; k < NK; k++) {
      hipStream_t s = (chained && (k & 1)) ? s1 : s0;
      chained_kernel<<<NWG, 256, 0, s>>>(slab + (size_t)(k & 7) * slab_elems, (k & 1) ? a1 : a0, (k & 1) ? a0 : a1,
                                         chained && k > 0 ? flags + (size_t)(k - 1) * NWG : nullptr, chained ? flags + (size_t)k * NWG : nullptr, 1u, err);
    }
    if (chained) { CK(hipEventRecord(ej, s1)); CK(hipStreamWaitEvent(s0, ej, 0)); }
  };
  for (int chained = 0; chained < 2; chained++) {
    for (int graph = 1; graph >= 0; graph--) {
      hipGraphExec_t ge = nullptr;
      if (graph) {
        hipGraph_t g;
        CK(hipStreamBeginCapture(s0, hipStreamCaptureModeThreadLocal));
        enqueue(chained);
        CK(hipStreamEndCapture(s0, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      }
      float best = 1e9;
      for (int r = 0; r < 12; r++) {
        CK(hipEventRecord(e0, s0));
        if (graph) CK(hipGraphLaunch(ge, s0)); else enqueue(chained);
        CK(hipEventRecord(e1, s0)); CK(hipEventSynchronize(e1)); CK(hipStreamSynchronize(s1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
      }
      int h_err; CK(hipMemcpy(&h_err, err, 4, hipMemcpyDeviceToHost)); CK(hipMemset(err, 0, 4));
      printf("%-44s %-10s %6.2f us per launch%s\n", chained ? "two streams, flag-chained (slab prefetched)" : "one stream (kernel boundary = dependency)",
             graph ? "hipGraph" : "eager", 1e3 * best / NK, h_err ? "   [SPIN LIMIT HIT: a launch waited for one that had not run]" : "");
    }
  }
  return 0;
}

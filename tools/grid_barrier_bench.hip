// Developer tool: what does a device-wide barrier inside one persistent launch cost on MI355X (256 workgroups, one per CU), against the kernel boundary
// of a hipGraph of small launches? Decides whether a persistent decode-step kernel can beat the 151-launch graph.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/grid_barrier_bench.hip -o tools/bin/grid_barrier_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// all workgroups arrive (release), wait until the counter reaches `target` (bounded spin), acquire
__device__ __forceinline__ void grid_barrier(unsigned *ctr, unsigned target, int *err) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence(); // release: this workgroup's stores are written back from the XCD's L2
    __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int spins = 0;
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > (1 << 20)) { *err = 1; break; }
    }
    __threadfence(); // acquire: stale lines of the other XCDs' data are dropped
  }
  __syncthreads();
}

// every phase: each workgroup reads the 64 KB vector all workgroups wrote in the previous phase (the decode step's h), adds, writes its own 256 B
template <bool WORK>
__global__ __launch_bounds__(256) void persistent_kernel(unsigned *ctr, int *err, float *buf0, float *buf1, int phases) {
  const int n = gridDim.x;
  for (int p = 0; p < phases; p++) {
    if (WORK) {
      const float *src = (p & 1) ? buf1 : buf0;
      float *dst = (p & 1) ? buf0 : buf1;
      float s = 0.f;
      for (int i = threadIdx.x; i < 16384; i += 256) s += src[i];
      if (threadIdx.x < 64) dst[blockIdx.x * 64 + threadIdx.x] = s * 1e-6f + 1.f;
    }
    grid_barrier(ctr, (unsigned)(p + 1) * n, err);
  }
}
template <bool WORK>
__global__ __launch_bounds__(256) void phase_kernel(const float *src, float *dst) {
  if (WORK) {
    float s = 0.f;
    for (int i = threadIdx.x; i < 16384; i += 256) s += src[i];
    if (threadIdx.x < 64) dst[blockIdx.x * 64 + threadIdx.x] = s * 1e-6f + 1.f;
  }
}

int main() {
  unsigned *ctr; int *err; float *b0, *b1;
  CK(hipMalloc(&ctr, 4)); CK(hipMalloc(&err, 4)); CK(hipMalloc(&b0, 65536)); CK(hipMalloc(&b1, 65536));
  CK(hipMemset(err, 0, 4)); CK(hipMemset(b0, 0, 65536)); CK(hipMemset(b1, 0, 65536));
  hipStream_t s; CK(hipStreamCreate(&s));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int phases = 150, reps = 20;
  for (int work = 0; work < 2; work++) {
    for (int grid : {64, 128, 256}) {
      float best = 1e9;
      for (int r = 0; r < reps; r++) {
        CK(hipMemsetAsync(ctr, 0, 4, s));
        CK(hipEventRecord(e0, s));
        if (work) persistent_kernel<true><<<grid, 256, 0, s>>>(ctr, err, b0, b1, phases);
        else persistent_kernel<false><<<grid, 256, 0, s>>>(ctr, err, b0, b1, phases);
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
      }
      int h_err; CK(hipMemcpy(&h_err, err, 4, hipMemcpyDeviceToHost));
      printf("persistent, %s, %3d workgroups: %6.2f us per phase (barrier%s)%s\n", work ? "64 KB read + 256 B write per workgroup" : "no work", grid,
             1e3 * best / phases, work ? " + work" : " only", h_err ? "  [SPIN LIMIT HIT]" : "");
    }
    // the same phases as a hipGraph of launches
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int p = 0; p < phases; p++) {
      if (work) phase_kernel<true><<<256, 256, 0, s>>>((p & 1) ? b1 : b0, (p & 1) ? b0 : b1);
      else phase_kernel<false><<<256, 256, 0, s>>>(b0, b1);
    }
    CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    float best = 1e9;
    for (int r = 0; r < reps; r++) {
      CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      best = ms < best ? ms : best;
    }
    printf("hipGraph of %d launches, %s, 256 workgroups: %6.2f us per launch\n", phases, work ? "same work" : "no work", 1e3 * best / phases);
  }
  return 0;
}

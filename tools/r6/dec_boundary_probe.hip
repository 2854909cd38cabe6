// Developer tool (round 6): go / no-go for a persistent decode-layer engine, measured at THIS model's op sizes.
// A GPT-2 decode layer (30 x 1024, f32 weights, B candidates) is five dependent ops whose inputs are ALL of the previous op's outputs (main.cpp:2667-3040):
//   LN1 + QKV (12.6 MB of weights) -> attention (K/V rows) -> projection + residual (4.2 MB) -> LN2 + FC + GELU (16.8 MB) -> FC2 + residual (16.8 MB).
// The product runs them as five launches per layer inside one hipGraph (27.1 us per layer at B = 16: profiles/r4_decode_launch_breakdown.txt). The guide's engine
// (MI355X_MICROARCH.md, rows barrier-xcd / prefetch-credit / engine-vs-launches) keeps the layer in ONE persistent launch: what it pays per op boundary is the
// XCD-hierarchical grid barrier + re-reading the op's input vector on every CU; what it earns is the weight stream running ahead across the boundary (at most
// 2.6-2.8 us for a 3-4-slot op, 1.2-1.4 for a 1-slot op: this model's ops are 1 / 3 / 4 / 4 slots of 16 KiB per CU).
// This tool measures the first half at the real sizes: the same synthetic layer (every workgroup streams its share of the op's weights with non-temporal loads, reads
// the op's whole input vector, writes its slice of the output) as (L) five launches per layer in a hipGraph and (P) one persistent launch with the XCD-hierarchical
// barrier between ops and NO run-ahead. P - L per layer is what the run-ahead loader would have to win back.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/r6/dec_boundary_probe.hip -o tools/bin/dec_boundary_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

static constexpr int NWG = 256, D = 1024, FF = 4096;
struct Op { int w_bytes_per_wg; int in_floats; int out_floats_per_wg; };  // weights streamed per workgroup, input vector read by every workgroup, output slice written
struct Sync { unsigned cnt[8]; unsigned gen[8]; unsigned top; unsigned census[8]; unsigned flat; int err; };

__device__ __forceinline__ int xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7; } // HW_REG_XCC_ID, bits 3:0

__device__ __forceinline__ bool spin_until(const unsigned *p, unsigned target, int *err) {
  int spins = 0;
  while (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
    __builtin_amdgcn_s_sleep(2);
    if (++spins > (1 << 22)) { *err = 1; return false; }
  }
  return true;
}
// XCD-hierarchical barrier (the guide's barrier-xcd row): arrive on the XCC's counter; the XCC's last arriver writes the XCD's L2 back (release), arrives on the top
// counter, waits for all 8 XCCs, acquires and publishes the XCC's generation; everyone else polls its XCC's generation (relaxed) and acquires once.
__device__ __forceinline__ void barrier_xcd(Sync *s, unsigned phase /* 1-based */, int xcc, unsigned per_xcc, unsigned n_xcc) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // this workgroup's stores have left the CU
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned old = __hip_atomic_fetch_add(&s->cnt[xcc], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old == per_xcc * phase - 1) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_fetch_add(&s->top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      spin_until(&s->top, n_xcc * phase, &s->err);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      __hip_atomic_store(&s->gen[xcc], phase, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      spin_until(&s->gen[xcc], phase, &s->err);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
  }
  __syncthreads();
}

// one op of one workgroup: stream the weight share (nt), read the whole input vector, write the output slice
__device__ __forceinline__ void op_body(const Op &op, const uint4 *__restrict__ w, const float *__restrict__ in, float *__restrict__ out, int wg) {
  float acc = 0.f;
  typedef unsigned u4 __attribute__((ext_vector_type(4)));
  const u4 *wp = (const u4 *)w + (size_t)wg * (op.w_bytes_per_wg / 16);
  for (int i = threadIdx.x; i < op.w_bytes_per_wg / 16; i += 256) { const u4 v = __builtin_nontemporal_load(wp + i); acc += __uint_as_float(v.x ^ v.y ^ v.z ^ v.w) * 1e-30f; }
  const float4 *ip = (const float4 *)in;
  for (int i = threadIdx.x; i < op.in_floats / 4; i += 256) { const float4 v = ip[i]; acc += (v.x + v.y) + (v.z + v.w); }
  for (int i = threadIdx.x; i < op.out_floats_per_wg; i += 256) out[(size_t)wg * op.out_floats_per_wg + i] = acc * 1e-9f + 1.0f;
}

__global__ __launch_bounds__(256) void op_kernel(Op op, const uint4 *w, const float *in, float *out) { op_body(op, w, in, out, blockIdx.x); }

__global__ __launch_bounds__(256) void persistent_kernel(const Op *ops, int nops, int layers, const uint4 *w, size_t region16, float *buf0, float *buf1, Sync *s) {
  __shared__ Op sops[8];
  __shared__ unsigned sh[2];
  const int xcc = xcc_id();
  if (threadIdx.x < nops) sops[threadIdx.x] = ops[threadIdx.x];
  // census: workgroups per XCC (placement is whatever the dispatcher chose), published with a flat barrier once
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(&s->census[xcc], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(&s->flat, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    spin_until(&s->flat, gridDim.x, &s->err);
    unsigned nx = 0;
    for (int x = 0; x < 8; x++) nx += __hip_atomic_load(&s->census[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > 0;
    sh[0] = __hip_atomic_load(&s->census[xcc], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    sh[1] = nx;
  }
  __syncthreads();
  const unsigned per_xcc = sh[0], n_xcc = sh[1];
  unsigned phase = 0;
  for (int l = 0; l < layers; l++)
    for (int o = 0; o < nops; o++) {
      const float *in = (phase & 1) ? buf1 : buf0;
      float *out = (phase & 1) ? buf0 : buf1;
      op_body(sops[o], w + (size_t)(phase & 3) * region16, in, out, blockIdx.x);
      phase++;
      barrier_xcd(s, phase, xcc, per_xcc, n_xcc);
    }
}

int main() {
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  uint4 *w; float *b0, *b1; Sync *sy; Op *dops;
  const size_t wbytes = (size_t)NWG * 80 * 1024;
  CK(hipMalloc(&w, wbytes * 4)); CK(hipMemset(w, 1, wbytes * 4)); // 4 distinct regions: rotate so that the stream comes from HBM / MALL, not from L2
  CK(hipMalloc(&b0, 1 << 20)); CK(hipMalloc(&b1, 1 << 20)); CK(hipMemset(b0, 0, 1 << 20)); CK(hipMemset(b1, 0, 1 << 20));
  CK(hipMalloc(&sy, sizeof(Sync))); CK(hipMalloc(&dops, 8 * sizeof(Op)));
  const int layers = 30, reps = 10;
  for (int B : {16, 1}) {
    // per workgroup (256 of them): weights of the op / 256; the op's input vector; its slice of the output
    const int keys = 165;
    const Op ops[5] = {{3 * D * D * 4 / NWG, B * D, B * 3 * D / NWG},            // LN1 + QKV
                       {B * 2 * keys * D * 2 / NWG, B * 3 * D, B * D / NWG},     // attention: fp16 K / V rows of the candidates, q | k | v in
                       {D * D * 4 / NWG, B * D, B * D / NWG},                    // projection + residual
                       {D * FF * 4 / NWG, B * D, B * FF / NWG},                  // LN2 + FC
                       {FF * D * 4 / NWG, B * FF, B * D / NWG}};                 // FC2 + residual
    CK(hipMemcpy(dops, ops, sizeof ops, hipMemcpyHostToDevice));
    // (L) launches in a graph
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    int phase = 0;
    for (int l = 0; l < layers; l++)
      for (int o = 0; o < 5; o++, phase++) op_kernel<<<NWG, 256, 0, st>>>(ops[o], w + (size_t)(phase & 3) * (wbytes / 16), (phase & 1) ? b1 : b0, (phase & 1) ? b0 : b1);
    CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    float bestL = 1e9, bestP = 1e9;
    for (int r = 0; r < reps; r++) {
      CK(hipEventRecord(e0, st)); CK(hipGraphLaunch(ge, st)); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); bestL = std::min(bestL, ms);
    }
    // (P) one persistent launch, barrier-xcd between ops, no run-ahead
    int herr = 0;
    for (int r = 0; r < reps; r++) {
      CK(hipMemsetAsync(sy, 0, sizeof(Sync), st));
      CK(hipEventRecord(e0, st));
      persistent_kernel<<<NWG, 256, 0, st>>>(dops, 5, layers, w, wbytes / 16, b0, b1, sy);
      CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); bestP = std::min(bestP, ms);
      Sync h; CK(hipMemcpy(&h, sy, sizeof h, hipMemcpyDeviceToHost)); herr |= h.err;
    }
    // barrier alone (no op bodies): zero-byte ops
    const Op nul[5] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    CK(hipMemcpy(dops, nul, sizeof nul, hipMemcpyHostToDevice));
    float bestB = 1e9;
    for (int r = 0; r < reps; r++) {
      CK(hipMemsetAsync(sy, 0, sizeof(Sync), st));
      CK(hipEventRecord(e0, st));
      persistent_kernel<<<NWG, 256, 0, st>>>(dops, 5, layers, w, wbytes / 16, b0, b1, sy);
      CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); bestB = std::min(bestB, ms);
    }
    const double mb = (3.0 * D * D + D * D + 2.0 * D * FF) * 4 / 1e6;
    printf("B = %2d  (%.1f MB of weights per layer, input vectors %d / %d / %d / %d / %d KB)\n", B, mb, B * D * 4 / 1024, B * 3 * D * 4 / 1024, B * D * 4 / 1024, B * D * 4 / 1024, B * FF * 4 / 1024);
    printf("   (L) five launches per layer, hipGraph of %d launches:            %7.2f us per layer  (%.2f us per op)\n", layers * 5, 1e3 * bestL / layers, 1e3 * bestL / layers / 5);
    printf("   (P) one persistent launch, XCD-hierarchical barrier per op:      %7.2f us per layer  (%.2f us per op)%s\n", 1e3 * bestP / layers, 1e3 * bestP / layers / 5, herr ? "  [SPIN LIMIT HIT]" : "");
    printf("       the barrier alone (no streaming, no vectors):               %7.2f us per layer  (%.2f us per barrier)\n", 1e3 * bestB / layers, 1e3 * bestB / layers / 5);
    printf("   P - L = %+.2f us per layer: what a run-ahead weight loader has to win back (guide: at most 2.6-2.8 us per 3-4-slot op, 1.2-1.4 per 1-slot op, none for\n"
           "   the attention op: <= 9.5 us per layer with three consumer waves per CU)\n", 1e3 * (bestP - bestL) / layers);
    fflush(stdout);
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  }
  return 0;
}

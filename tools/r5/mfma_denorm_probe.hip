// Does v_mfma_f32_16x16x32_f16 honour fp16 SUBNORMAL inputs, and does the f32 -> fp16 conversion produce them?  (round 5: the softmax numerators of the
// diffusion attention kernel are fp16 MFMA operands in (0, 1]; anything below 2^-14 = 6.1e-5 is subnormal)
//   hipcc --offload-arch=gfx950 -O2 tools/r5/mfma_denorm_probe.hip -o tools/bin/mfma_denorm_probe && tools/bin/mfma_denorm_probe
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
__global__ void probe(const float *in, float *out, unsigned short *bits) {
  const int lane = threadIdx.x;
  // A = all ones (16 x 32), B[k][n]: column n carries in[n] at k = 0, zero elsewhere -> D[m][n] = fp16(in[n])
  half8 a, b;
  for (int e = 0; e < 8; e++) { a[e] = (_Float16)1.0f; b[e] = (_Float16)0.0f; }
  const _Float16 h = (_Float16)in[lane & 15];
  if ((lane >> 4) == 0) b[0] = h;
  bits[lane] = *(const unsigned short *)&h;
  floatx4 d = {0.f, 0.f, 0.f, 0.f};
  d = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, d, 0, 0, 0);
  out[lane] = d[0];
}
int main() {
  float h_in[16], h_out[64];
  unsigned short h_bits[64];
  for (int i = 0; i < 16; i++) h_in[i] = 1.0f / (float)(1u << (8 + i)) * 1.25f; // 1.25 * 2^-8 .. 2^-23: fp16 normal down to 2^-14, subnormal below
  float *d_in, *d_out; unsigned short *d_bits;
  hipMalloc(&d_in, sizeof h_in); hipMalloc(&d_out, sizeof h_out); hipMalloc(&d_bits, sizeof h_bits);
  hipMemcpy(d_in, h_in, sizeof h_in, hipMemcpyHostToDevice);
  probe<<<1, 64>>>(d_in, d_out, d_bits);
  hipMemcpy(h_out, d_out, sizeof h_out, hipMemcpyDeviceToHost);
  hipMemcpy(h_bits, d_bits, sizeof h_bits, hipMemcpyDeviceToHost);
  for (int i = 0; i < 16; i++)
    printf("x = 1.25 * 2^-%-2d = %.6e  fp16 bits 0x%04x  mfma(ones, x) = %.6e  %s\n", 8 + i, h_in[i], h_bits[i], h_out[i],
           h_out[i] == 0.f ? "FLUSHED" : (h_bits[i] & 0x7c00) ? "normal" : "subnormal kept");
  return 0;
}

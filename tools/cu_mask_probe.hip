// Developer probe: which (XCD, CU) does each bit of a hipExtStreamCreateWithCUMask mask enable on this device, and do kernels on
// two disjointly masked streams run concurrently?  hipcc --offload-arch=gfx950 -O2 tools/cu_mask_probe.hip -o tools/bin/cu_mask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <map>
#include <set>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void where_kernel(uint32_t *out, int spin) {
  uint32_t xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)); // HW_REG_XCC_ID, all 32 bits
  uint32_t hw = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_REG_HW_ID
  long long t0 = wall_clock64();
  while (wall_clock64() - t0 < spin) {}
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hw; }
}

static int placement(hipStream_t s, uint32_t *dout, std::map<int, std::set<int>> &cus) {
  const int nb = 8192;
  where_kernel<<<nb, 64, 0, s>>>(dout, 2000);
  if (hipStreamSynchronize(s) != hipSuccess) return 1;
  std::vector<uint32_t> h(2 * nb);
  (void)hipMemcpy(h.data(), dout, h.size() * 4, hipMemcpyDeviceToHost);
  cus.clear();
  for (int b = 0; b < nb; b++) {
    int xcc = h[2 * b] & 0xf, hw = h[2 * b + 1];
    int cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    cus[xcc].insert(se * 32 + sh * 16 + cu);
  }
  return 0;
}

int main() {
  uint32_t *dout;
  CK(hipMalloc(&dout, 8192 * 8));
  std::map<int, std::set<int>> cus;
  hipStream_t s0;
  CK(hipStreamCreate(&s0));
  placement(s0, dout, cus);
  int tot = 0;
  for (auto &kv : cus) tot += (int)kv.second.size();
  printf("unmasked: %d XCDs, %d CUs seen:", (int)cus.size(), tot);
  for (auto &kv : cus) printf(" xcc%d:%d", kv.first, (int)kv.second.size());
  printf("\n");
  auto show = [&](const char *name, std::vector<uint32_t> mask) -> int {
    hipStream_t s;
    hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data());
    if (e != hipSuccess) { printf("%s: create failed: %s\n", name, hipGetErrorString(e)); return 1; }
    placement(s, dout, cus);
    int t = 0;
    for (auto &kv : cus) t += (int)kv.second.size();
    printf("%-28s %3d CUs:", name, t);
    for (auto &kv : cus) { printf(" xcc%d:%d{", kv.first, (int)kv.second.size()); int k = 0; for (int c : kv.second) if (k++ < 6) printf("%d,", c); printf("}"); }
    printf("\n");
    (void)hipStreamDestroy(s);
    return 0;
  };
  show("bits 0..31", {0xffffffffu, 0, 0, 0, 0, 0, 0, 0});
  show("bits 0..7", {0xffu, 0, 0, 0, 0, 0, 0, 0});
  show("bits 0..15", {0xffffu, 0, 0, 0, 0, 0, 0, 0});
  show("bit 0", {1u, 0, 0, 0, 0, 0, 0, 0});
  show("bit 1", {2u, 0, 0, 0, 0, 0, 0, 0});
  show("bit 8", {0x100u, 0, 0, 0, 0, 0, 0, 0});
  show("bits 32..63", {0, 0xffffffffu, 0, 0, 0, 0, 0, 0});
  show("bits 224..255", {0, 0, 0, 0, 0, 0, 0, 0xffffffffu});
  show("every 8th bit", {0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u});
  show("all but bits 0..15", {0xffff0000u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u});
  show("1 word only (size 1)", {0xffffffffu});
  return 0;
}

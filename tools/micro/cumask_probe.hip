// Which physical CUs do the bits of hipExtStreamCreateWithCUMask select?  Launches a census kernel on streams with
// different masks and prints, per XCC, the set of (SE, CU) ids its workgroups ran on.
//   hipcc --offload-arch=gfx950 -O2 tools/micro/cumask_probe.hip -o /tmp/cumask_probe && /tmp/cumask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <map>
#include <set>
#include <vector>

__global__ void census(unsigned* out) {
  unsigned xcc, hw;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  // burn a little time so that the grid spreads over every CU the mask allows
  float x = threadIdx.x;
  for (int i = 0; i < 20000; ++i) x = x * 1.0001f + 0.5f;
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc & 0xf; out[2 * blockIdx.x + 1] = hw; }
  if (x == 12345.f) out[0] = 0;
}

static void run(const char* name, const std::vector<uint32_t>& mask) {
  hipStream_t st;
  if (hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data()) != hipSuccess) { printf("%s: stream creation refused\n", name); return; }
  const int n = 4096;
  unsigned* d; hipMalloc(&d, n * 8); hipMemsetAsync(d, 0xff, n * 8, st);
  hipLaunchKernelGGL(census, dim3(n), dim3(64), 0, st, d);
  std::vector<unsigned> h(2 * n);
  hipStreamSynchronize(st);
  hipMemcpy(h.data(), d, n * 8, hipMemcpyDeviceToHost);
  std::map<unsigned, std::set<unsigned>> per;
  for (int i = 0; i < n; ++i) { unsigned hw = h[2 * i + 1]; unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7; per[h[2 * i]].insert(se * 100 + sh * 16 + cu); }
  printf("%s:\n", name);
  int tot = 0;
  for (auto& kv : per) { printf("  xcc %u: %zu CUs:", kv.first, kv.second.size()); tot += kv.second.size(); for (unsigned v : kv.second) printf(" %u.%u", v / 100, v % 100); printf("\n"); }
  printf("  total distinct CUs %d\n", tot);
  hipFree(d); hipStreamDestroy(st);
}

int main() {
  std::vector<uint32_t> m(8, 0);
  for (int i = 0; i < 224; ++i) m[i / 32] |= 1u << (i % 32);
  run("bits 0..223", m);
  std::vector<uint32_t> m2(8, 0);
  for (int i = 224; i < 256; ++i) m2[i / 32] |= 1u << (i % 32);
  run("bits 224..255", m2);
  std::vector<uint32_t> m3(8, 0);
  for (int i = 0; i < 32; ++i) m3[i / 32] |= 1u << (i % 32);
  run("bits 0..31", m3);
  std::vector<uint32_t> m4(8, 0);
  for (int i = 0; i < 256; i += 8) m4[i / 32] |= 1u << (i % 32);
  run("bits 0,8,16,..", m4);
  std::vector<uint32_t> m5(8, 0xffffffffu);
  run("all", m5);
  return 0;
}

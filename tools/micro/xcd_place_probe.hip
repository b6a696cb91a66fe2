// Is workgroup -> XCD placement static round-robin, alone and under load, with and without a CU mask that enables whole XCDs?
// (lstm.hip's XCD-local recurrences rely on it.)  Prints per-XCC block counts and whether xcc(block b) == (b + c) % n.
//   hipcc --offload-arch=gfx950 -O2 tools/micro/xcd_place_probe.hip -o /tmp/xcd_place_probe && /tmp/xcd_place_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__global__ void census(unsigned* out, int spin) {
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  float x = threadIdx.x;
  for (int i = 0; i < spin; ++i) x = x * 1.0001f + 0.5f;
  if (threadIdx.x == 0) out[blockIdx.x] = xcc & 0xf;
  if (x == 12345.f) out[0] = 0;
}
__global__ __launch_bounds__(256) void filler(float* sink, int spin) {      // 70 KB of LDS: two per CU, like the conv kernels
  extern __shared__ float lds[];
  float x = threadIdx.x;
  for (int i = 0; i < spin; ++i) { x = x * 1.0001f + 0.5f; lds[threadIdx.x] = x; }
  if (x == 12345.f) sink[0] = lds[0];
}

static void report(const char* name, const std::vector<unsigned>& h, int nx) {
  std::vector<int> cnt(16, 0);
  for (unsigned v : h) cnt[v & 15]++;
  int rr = 0;
  for (size_t b = 0; b < h.size(); ++b) rr += h[b] == (h[0] + b) % nx ? 1 : 0;      // only meaningful for an unmasked grid
  int same8 = 0;
  for (size_t b = 0; b + nx < h.size(); ++b) same8 += h[b] == h[b + nx];
  printf("%-58s counts:", name);
  for (int i = 0; i < 8; ++i) printf(" %d", cnt[i]);
  printf("   xcc(b)==(xcc(0)+b)%%%d: %d/%zu   xcc(b)==xcc(b+%d): %d/%zu\n", nx, rr, h.size(), nx, same8, h.size() - nx);
}

int main() {
  hipFuncSetAttribute(reinterpret_cast<const void*>(&filler), hipFuncAttributeMaxDynamicSharedMemorySize, 70 * 1024);
  unsigned* d; hipMalloc(&d, 4096 * 4);
  float* sink; hipMalloc(&sink, 4);
  hipStream_t plain, fill, m4;
  hipStreamCreateWithFlags(&plain, hipStreamNonBlocking);
  hipStreamCreateWithFlags(&fill, hipStreamNonBlocking);
  std::vector<uint32_t> mask(8, 0);
  for (int i = 0; i < 256; ++i) if (i % 8 < 4) mask[i / 32] |= 1u << (i % 32);           // XCDs 0..3, all their CUs
  if (hipExtStreamCreateWithCUMask(&m4, 8, mask.data()) != hipSuccess) { printf("masked stream refused\n"); return 1; }
  for (int load = 0; load < 2; ++load) {
    for (int rep = 0; rep < 3; ++rep) {
      if (load) for (int k = 0; k < 6; ++k) hipLaunchKernelGGL(filler, dim3(2048), dim3(256), 70 * 1024, fill, sink, 40000);
      for (int which = 0; which < 3; ++which) {
        const int n = which == 0 ? 512 : which == 1 ? 128 : 256, nx = which == 0 ? 8 : 4;
        hipStream_t st = which == 0 ? plain : m4;
        hipMemsetAsync(d, 0xff, n * 4, st);
        hipLaunchKernelGGL(census, dim3(n), dim3(which == 2 ? 512 : 64), 0, st, d, 2000);
        std::vector<unsigned> h(n);
        hipStreamSynchronize(st);
        hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost);
        char name[128];
        snprintf(name, sizeof(name), "%s, %s, %d blocks of %d:", load ? "beside a chip-filling kernel" : "alone", which == 0 ? "no mask" : "mask = XCDs 0-3",
                 n, which == 2 ? 512 : 64);
        report(name, h, nx);
      }
      hipDeviceSynchronize();
    }
  }
  return 0;
}

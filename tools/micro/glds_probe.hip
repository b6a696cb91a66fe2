// __builtin_amdgcn_global_load_lds(global ptr (per lane), LDS ptr (wave-uniform base), 16, offset, aux): the 64 lanes'
// 16-byte pieces land lane-linearly at base + lane*16.  Checks (a) that mapping, with a per-lane permuted SOURCE, for
// two waves writing different 1-KiB pieces, and (b) a zero page as the source of out-of-range lanes.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ float zero_page[4] = {0.f, 0.f, 0.f, 0.f};

__global__ __launch_bounds__(128) void probe(const float* src, float* out) {
  __shared__ __attribute__((aligned(16))) float lds[2 * 64 * 4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 512; i += 128) lds[i] = 123.0f;
  __syncthreads();
  const int perm = (lane * 7 + 3) & 63;                                  // source piece of this lane
  const float* g = (lane % 5 == 4) ? zero_page : src + (wave * 64 + perm) * 4;
  float* base = lds + wave * 256;                                        // wave-uniform
  __builtin_amdgcn_global_load_lds(g, (__attribute__((address_space(3))) void*)base, 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = tid; i < 512; i += 128) out[i] = lds[i];
}

int main() {
  float h[512], *d, *o, r[512];
  for (int i = 0; i < 512; ++i) h[i] = 1000.f + i;
  hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(h));
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(128), 0, 0, d, o);
  hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int w = 0; w < 2; ++w)
    for (int l = 0; l < 64; ++l)
      for (int j = 0; j < 4; ++j) {
        float want = (l % 5 == 4) ? 0.f : h[(w * 64 + ((l * 7 + 3) & 63)) * 4 + j];
        if (r[(w * 64 + l) * 4 + j] != want) { if (bad < 5) printf("mismatch w%d l%d j%d got %.0f want %.0f\n", w, l, j, r[(w * 64 + l) * 4 + j], want); ++bad; }
      }
  printf("glds probe: %d mismatches\n", bad);
  return 0;
}

// Does a direct-to-LDS buffer load (buffer_load_dwordx4 ... lds) write ZEROS for lanes whose offset fails the
// descriptor's range check, or does it leave the LDS bytes untouched?  (The implicit-GEMM engine relies on the range
// check for conv halos and ragged edges.)   hipcc --offload-arch=gfx950 -O3 lds_dma_oob.hip -o lds_dma_oob
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void probe(const float* src, unsigned nbytes, float* out) {
  __shared__ __attribute__((aligned(16))) float lds[64 * 4];
  for (int i = threadIdx.x; i < 256; i += 64) lds[i] = 123.0f;
  __syncthreads();
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, nbytes, 0x00020000);
  // even lanes read in range, odd lanes out of range (offset = nbytes)
  unsigned off = (threadIdx.x & 1) ? nbytes : threadIdx.x * 16u;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds, 16, off, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 256; i += 64) out[i] = lds[i];
}

int main() {
  float h[256], *d, *o;
  for (int i = 0; i < 256; ++i) h[i] = 1000.f + i;
  (void)hipMalloc(&d, sizeof(h)); (void)hipMalloc(&o, sizeof(h));
  (void)hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, (unsigned)sizeof(h), o);
  float r[256];
  (void)hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
  printf("lane0 (in range): %.0f %.0f %.0f %.0f\n", r[0], r[1], r[2], r[3]);
  printf("lane1 (out of range): %.0f %.0f %.0f %.0f\n", r[4], r[5], r[6], r[7]);
  printf("lane2 (in range): %.0f %.0f %.0f %.0f\n", r[8], r[9], r[10], r[11]);
  printf("lane63 (out of range): %.0f %.0f %.0f %.0f\n", r[252], r[253], r[254], r[255]);
  return 0;
}

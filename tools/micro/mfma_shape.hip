// Bare fp32 MFMA loops on random operands: does the chip sustain a different rate (clock) on 16x16x4 than on
// 32x32x2?  (MI355X_MICROARCH.md "DVFS give-back" item 7.)   hipcc --offload-arch=gfx950 -O3 mfma_shape.hip -o mfma_shape
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int SHAPE>
__global__ __launch_bounds__(1024) void loop(const float* in, float* out, int iters) {
  float a[8], b[8];
  for (int i = 0; i < 8; ++i) { a[i] = in[(threadIdx.x * 8 + i) & 4095]; b[i] = in[(threadIdx.x * 8 + i + 2048) & 4095]; }
  if (SHAPE == 32) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k], b[k], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k], b[(k + 1) & 7], acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(k + 1) & 7], b[k], acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(k + 3) & 7], b[(k + 2) & 7], acc[3], 0, 0, 0);
      }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 1024 + threadIdx.x] = s;
  } else {
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int t = 0; t < 16; ++t)
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(k + t) & 7], b[(k + (t >> 2)) & 7], acc[t], 0, 0, 0);
      }
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
    out[blockIdx.x * 1024 + threadIdx.x] = s;
  }
}

int main() {
  float *in, *out;
  hipMalloc(&in, 4096 * 4); hipMalloc(&out, 2048 * 256 * 4);
  float h[4096];
  srand(1);
  for (int i = 0; i < 4096; ++i) h[i] = (rand() / (float)RAND_MAX - 0.5f) * 0.01f;
  hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  for (int waves_per_simd = 1; waves_per_simd <= 4; ++waves_per_simd) {
    int blocks = 256;                      // one workgroup per CU; its size sets the waves per SIMD
    int threads = 256 * waves_per_simd;
    for (int shape : {32, 16}) {
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (shape == 32) hipLaunchKernelGGL(loop<32>, dim3(blocks), dim3(threads), 0, 0, in, out, iters);
        else hipLaunchKernelGGL(loop<16>, dim3(blocks), dim3(threads), 0, 0, in, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        // per iteration per wave: 32 MFMAs x 4096 flop (32x32x2) or 64 x 2048 (16x16x4) = 131072 flop
        double fl = (double)blocks * (threads / 64) * iters * 131072.0;
        if (rep == 1) printf("waves/SIMD %d  shape %2d: %.2f ms  %.1f TFLOP/s\n", waves_per_simd, shape, ms, fl / ms * 1e-9);
      }
    }
  }
  return 0;
}

// Does a wavefront's own vector-ALU work hide behind its MFMAs?  One wavefront per SIMD (or two) runs groups of 8 v_mfma_f32_32x32x2_f32 on NACC
// accumulator tiles in rotation, with K other instructions behind every MFMA (inline asm: the order is the source's).  Prints cycles per MFMA
// (s_memtime over the loop, shader clock) -- 64 is the pipe's rate.     hipcc --offload-arch=gfx950 -O3 mfma_coissue.hip -o build/mfma_coissue
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define MFMA(acc) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define PKFMA(x) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(x) : "v"(y2), "v"(z2))
#define VADD(x) asm volatile("v_add_f32 %0, %1, %0" : "+v"(x) : "v"(a))
#define VMOV(x) asm volatile("v_mov_b32 %0, %1" : "=v"(x) : "v"(a))

// KIND 0: v_pk_fma_f32, 1: v_add_f32, 2: v_mov_b32, 3: s_nop 0, 4: ds_read_b128 (LDS), 5: pk_fma all on ONE register (dependent chain),
// 6: s_add_u32, 7: buffer_load_dwordx4 (L1/L2 hits), 8: ds_write_b32, 9: v_accvgpr_read_b32, 10: v_cndmask_b32 (VOP2 with vcc), 11: v_cmp_gt_f32
template <int K, int KIND, int NACC>
__global__ __launch_bounds__(512) void loop(const float* in, float* out, long long* cyc, int iters) {
  __shared__ float lds[4096];
  float a = in[threadIdx.x & 4095], b = in[(threadIdx.x + 2048) & 4095];
  lds[threadIdx.x] = a; lds[threadIdx.x + 512] = b;
  __syncthreads();
  f32x2 y2 = {a, b}, z2 = {b, a};
  f32x2 x[8];
  float s1[8];
  for (int i = 0; i < 8; ++i) { x[i] = f32x2{a * i, b}; s1[i] = a + i; }
  f32x16 acc[8];
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  f32x4 ld[4] = {};
  const unsigned laddr = (threadIdx.x & 63) * 16;
  unsigned sc = 0;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in), 0, 16384, 0x00020000);
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      MFMA(acc[g % NACC]);
#pragma unroll
      for (int k = 0; k < K; ++k) {
        if (KIND == 0) PKFMA(x[(g * K + k) & 7]);
        if (KIND == 1) VADD(s1[(g * K + k) & 7]);
        if (KIND == 2) VMOV(s1[(g * K + k) & 7]);
        if (KIND == 3) asm volatile("s_nop 0");
        if (KIND == 4) asm volatile("ds_read_b128 %0, %1" : "=v"(ld[(g * K + k) & 3]) : "v"(laddr));
        if (KIND == 5) PKFMA(x[0]);
        if (KIND == 6) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sc));
        if (KIND == 7) asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(ld[(g * K + k) & 3]) : "v"(laddr), "s"(rs));
        if (KIND == 8) asm volatile("ds_write_b32 %0, %1" : : "v"(laddr), "v"(a));
        if (KIND == 9) asm volatile("v_accvgpr_read_b32 %0, a0" : "=v"(s1[(g * K + k) & 7]));
        if (KIND == 10) asm volatile("v_cndmask_b32 %0, %1, %0, vcc" : "+v"(s1[(g * K + k) & 7]) : "v"(a));
        if (KIND == 11) asm volatile("v_cmp_gt_f32 vcc, %0, %1" : : "v"(s1[(g * K + k) & 7]), "v"(a) : "vcc");
      }
    }
    if (KIND == 4 || KIND == 8) asm volatile("s_waitcnt lgkmcnt(0)");
    if (KIND == 7) asm volatile("s_waitcnt vmcnt(0)");
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) { for (int r = 0; r < 16; ++r) s += acc[i][r]; s += x[i][0] + x[i][1] + s1[i]; }
  for (int i = 0; i < 4; ++i) s += ld[i][0];
  s += (float)sc;
  out[blockIdx.x * 512 + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}

template <int K, int KIND, int NACC>
void run(const char* what, const float* in, float* out, long long* cyc, int waves_per_simd) {
  const int iters = 4000;
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL((loop<K, KIND, NACC>), dim3(256), dim3(256 * waves_per_simd), 0, 0, in, out, cyc, iters);
    hipDeviceSynchronize();
  }
  long long h[8];
  hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  // s_memtime counts at 100 MHz on this part: convert with the measured wall time instead -> report ticks and let main() scale
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((loop<K, KIND, NACC>), dim3(256), dim3(256 * waves_per_simd), 0, 0, in, out, cyc, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double per_mfma_ns = ms * 1e6 / ((double)iters * 8 * waves_per_simd);   // per MFMA of the SIMD (the pipe's view)
  printf("%-34s K=%2d acc=%d waves/SIMD=%d : %7.2f ns per MFMA of the SIMD = %6.1f cycles at 2.4 GHz   (memtime ticks/MFMA/wave %.1f)\n", what, K, NACC,
         waves_per_simd, per_mfma_ns, per_mfma_ns * 2.4, (double)h[0] / (iters * 8.0));
}

int main() {
  float *in, *out; long long* cyc;
  hipMalloc(&in, 4096 * 4); hipMalloc(&out, 512 * 256 * 4); hipMalloc(&cyc, 64);
  float h[4096];
  srand(1);
  for (int i = 0; i < 4096; ++i) h[i] = (rand() / (float)RAND_MAX - 0.5f) * 0.01f;
  hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  for (int w = 1; w <= 2; ++w) {
    run<0, 0, 2>("mfma only, 2 tiles alternating", in, out, cyc, w);
    run<0, 0, 8>("mfma only, 8 tiles", in, out, cyc, w);
    run<1, 0, 2>("v_pk_fma_f32", in, out, cyc, w);
    run<2, 0, 2>("v_pk_fma_f32", in, out, cyc, w);
    run<4, 0, 2>("v_pk_fma_f32", in, out, cyc, w);
    run<8, 0, 2>("v_pk_fma_f32", in, out, cyc, w);
    run<12, 0, 2>("v_pk_fma_f32", in, out, cyc, w);
    run<16, 0, 2>("v_pk_fma_f32", in, out, cyc, w);
    run<4, 1, 2>("v_add_f32", in, out, cyc, w);
    run<8, 1, 2>("v_add_f32", in, out, cyc, w);
    run<12, 1, 2>("v_add_f32", in, out, cyc, w);
    run<16, 1, 2>("v_add_f32", in, out, cyc, w);
    run<8, 2, 2>("v_mov_b32", in, out, cyc, w);
    run<8, 3, 2>("s_nop", in, out, cyc, w);
    run<1, 4, 2>("ds_read_b128", in, out, cyc, w);
    run<2, 4, 2>("ds_read_b128", in, out, cyc, w);
    run<4, 5, 2>("v_pk_fma_f32 dependent chain", in, out, cyc, w);
    run<8, 0, 8>("v_pk_fma_f32, 8 tiles", in, out, cyc, w);
    run<8, 6, 2>("s_add_u32", in, out, cyc, w);
    run<1, 7, 2>("buffer_load_dwordx4", in, out, cyc, w);
    run<2, 7, 2>("buffer_load_dwordx4", in, out, cyc, w);
    run<2, 8, 2>("ds_write_b32", in, out, cyc, w);
    run<4, 8, 2>("ds_write_b32", in, out, cyc, w);
    run<8, 9, 2>("v_accvgpr_read_b32", in, out, cyc, w);
    run<8, 10, 2>("v_cndmask_b32", in, out, cyc, w);
    run<8, 11, 2>("v_cmp_gt_f32", in, out, cyc, w);
  }
  return 0;
}

// What does a wavefront that is NOT issuing MFMAs get while its SIMD partner streams them?  Workgroup of 8 wavefronts on one CU: wavefronts 0-3 (one per
// SIMD) run a pure v_mfma_f32_32x32x2_f32 stream, wavefronts 4-7 (their partners) run N instructions of one kind; both stamp s_memtime around
// their loops.  Prints the partner's cycles per instruction beside the stream and alone, and what the stream lost.
//   hipcc --offload-arch=gfx950 -O3 mfma_partner.hip -o build/mfma_partner
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// KIND 0 v_add_f32 (8 independent chains), 1 v_pk_fma_f32, 2 v_mov_b32, 3 ds_write_b32, 4 ds_read_b128, 5 s_add_u32 + s_nop, 6 v_add dependent chain,
// 7 buffer_load_dwordx4 (cache hits), 8 v_cndmask
// YIELD: what the STREAM puts behind every MFMA: 0 nothing, 1 s_nop 0, 2 s_nop 7, 3 s_sleep 0, 4 s_setprio 0, 5 v_nop, 6 s_nop 0 behind every 4th MFMA only,
// 7 two s_nop 0, 8 s_waitcnt lgkmcnt(0) (nothing outstanding), 9 an independent s_add
template <int KIND, int YIELD>
__global__ __launch_bounds__(512) void k(const float* in, float* out, long long* t, int mfmas, int others, int stream_on, int swap, int prio) {
  __shared__ float lds[4096];
  const int wave0 = threadIdx.x >> 6;
  const int wave = swap ? (wave0 ^ 4) : wave0;          // swap = 1: the YOUNGER wavefronts (4-7) stream, the older ones are the partners
  float a = in[threadIdx.x & 4095], b = in[(threadIdx.x + 2048) & 4095];
  lds[threadIdx.x] = a;
  __syncthreads();
  float s = 0.f;
  if (wave < 4) {
    if (!stream_on) return;
    f32x16 acc[2];
    unsigned ssc = 0;
    for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < mfmas / 8; ++it) {
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc[g & 1]) : "v"(a), "v"(b));
        if (YIELD == 1) asm volatile("s_nop 0");
        if (YIELD == 2) asm volatile("s_nop 7");
        if (YIELD == 3) asm volatile("s_sleep 0");
        if (YIELD == 4) asm volatile("s_setprio 0");
        if (YIELD == 5) asm volatile("v_nop");
        if (YIELD == 6 && (g & 3) == 3) asm volatile("s_nop 0");
        if (YIELD == 7) asm volatile("s_nop 0\n\ts_nop 0");
        if (YIELD == 8) asm volatile("s_waitcnt lgkmcnt(0)");
        if (YIELD == 9) asm volatile("s_add_u32 %0, %0, 1" : "+s"(ssc));
      }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    s += (float)ssc;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) { t[wave * 2] = t0; t[wave * 2 + 1] = t1; }
  } else {
    float x[8];
    f32x2 x2[8], y2 = {a, b}, z2 = {b, a};
    f32x4 ld[4] = {};
    for (int i = 0; i < 8; ++i) { x[i] = a + i; x2[i] = f32x2{a, b + i}; }
    unsigned sc = 0;
    const unsigned laddr = (threadIdx.x & 63) * 4;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in), 0, 16384, 0x00020000);
    if (prio) asm volatile("s_setprio 3");
    // let the stream get going
    for (int i = 0; i < 64; ++i) asm volatile("s_sleep 8");
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < others / 8; ++it) {
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        if (KIND == 0) asm volatile("v_add_f32 %0, %1, %0" : "+v"(x[g]) : "v"(a));
        if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(x2[g]) : "v"(y2), "v"(z2));
        if (KIND == 2) asm volatile("v_mov_b32 %0, %1" : "=v"(x[g]) : "v"(a));
        if (KIND == 3) asm volatile("ds_write_b32 %0, %1" : : "v"(laddr), "v"(a));
        if (KIND == 4) asm volatile("ds_read_b128 %0, %1" : "=v"(ld[g & 3]) : "v"(laddr * 4));
        if (KIND == 5) asm volatile("s_add_u32 %0, %0, 1\n\ts_nop 0" : "+s"(sc));
        if (KIND == 6) asm volatile("v_add_f32 %0, %1, %0" : "+v"(x[0]) : "v"(a));
        if (KIND == 7) asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(ld[g & 3]) : "v"(laddr * 4), "s"(rs));
        if (KIND == 8) asm volatile("v_cndmask_b32 %0, %1, %0, vcc" : "+v"(x[g]) : "v"(a));
      }
      if (KIND == 3 || KIND == 4) asm volatile("s_waitcnt lgkmcnt(0)");
      if (KIND == 7) asm volatile("s_waitcnt vmcnt(0)");
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < 8; ++i) s += x[i] + x2[i][0];
    for (int i = 0; i < 4; ++i) s += ld[i][0];
    s += (float)sc;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) { t[wave * 2] = t0; t[wave * 2 + 1] = t1; }
  }
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int KIND, int YIELD = 0>
void run(const char* what, const float* in, float* out, long long* t, int swap = 0, int prio = 0) {
  const int mfmas = 40000, others = 4000;
  long long h[16], hl[16] = {};
  double per[2], stream[2];
  for (int on = 1; on >= 0; --on) {
    for (int rep = 0; rep < 2; ++rep) {
      hipMemset(t, 0, 128);
      hipLaunchKernelGGL((k<KIND, YIELD>), dim3(256), dim3(512), 0, 0, in, out, t, mfmas, others, on, swap, prio);
      hipDeviceSynchronize();
    }
    hipMemcpy(h, t, sizeof(h), hipMemcpyDeviceToHost);
    if (on) for (int i = 0; i < 16; ++i) hl[i] = h[i];
    per[on] = (double)(h[9] - h[8]) / others;                 // wavefront 4 (partner of 0)
    stream[on] = on ? (double)(h[1] - h[0]) / mfmas : 0.0;
  }
  // s_memtime ticks at a fixed 100 MHz: x 24 = shader cycles at 2.4 GHz
  // (s_memtime counts shader cycles on this part: a lone stream reads 64.0 per MFMA)
  if (swap == 0 && prio == 0 && KIND == 0)
    printf("    (timeline, cycles after the stream's start: partner starts %lld, partner ends %lld, stream ends %lld)\n", hl[8] - hl[0], hl[9] - hl[0], hl[1] - hl[0]);
  printf("yield %d ", YIELD);
  printf("%-28s swap %d prio %d  partner: %7.1f cycles / instruction beside the stream, %6.1f alone;   stream: %.1f cycles / MFMA (partner busy for %.0f %% of it)\n", what, swap, prio,
         per[1], per[0], stream[1], 100.0 * per[1] * others / (stream[1] * mfmas));
}

int main() {
  float *in, *out; long long* t;
  hipMalloc(&in, 4096 * 4); hipMalloc(&out, 512 * 256 * 4); hipMalloc(&t, 128);
  float h[4096];
  srand(1);
  for (int i = 0; i < 4096; ++i) h[i] = (rand() / (float)RAND_MAX - 0.5f) * 0.01f;
  hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  run<0>("v_add_f32 (independent)", in, out, t);
  run<6>("v_add_f32 (one chain)", in, out, t);
  run<1>("v_pk_fma_f32", in, out, t);
  run<2>("v_mov_b32", in, out, t);
  run<8>("v_cndmask_b32", in, out, t);
  run<3>("ds_write_b32", in, out, t);
  run<4>("ds_read_b128", in, out, t);
  run<5>("s_add_u32 + s_nop", in, out, t);
  run<7>("buffer_load_dwordx4", in, out, t);
  run<0>("v_add_f32 (independent)", in, out, t, 1, 0);
  run<5>("s_add_u32 + s_nop", in, out, t, 1, 0);
  run<3>("ds_write_b32", in, out, t, 1, 0);
  run<0, 1>("v_add_f32 | s_nop 0", in, out, t);
  run<0, 2>("v_add_f32 | s_nop 7", in, out, t);
  run<0, 3>("v_add_f32 | s_sleep 0", in, out, t);
  run<0, 4>("v_add_f32 | s_setprio 0", in, out, t);
  run<0, 5>("v_add_f32 | v_nop", in, out, t);
  run<0, 6>("v_add_f32 | s_nop 0 every 4th", in, out, t);
  run<0, 7>("v_add_f32 | 2 x s_nop 0", in, out, t);
  run<0, 8>("v_add_f32 | s_waitcnt", in, out, t);
  run<0, 9>("v_add_f32 | s_add", in, out, t);
  run<3, 1>("ds_write_b32 | s_nop 0", in, out, t);
  run<5, 1>("s_add+s_nop | s_nop 0", in, out, t);
  run<7, 1>("buffer_load | s_nop 0", in, out, t);
  run<0>("v_add_f32 (independent)", in, out, t, 0, 1);
  run<0>("v_add_f32 (independent)", in, out, t, 1, 1);
  return 0;
}

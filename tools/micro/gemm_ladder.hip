// Where does an LDS-tiled fp32 MFMA GEMM loop lose throughput on gfx950?  A ladder of loops with the SAME MFMA
// stream as the implicit-GEMM engine's 256x64 tile (4 waves, 64x64 per wave, BK=16: 32 MFMAs per k-tile):
//   V0 bare MFMAs (register operands)          V1 + 8 ds_read_b128 fragment reads per k-tile
//   V2 + __syncthreads per k-tile              V3 + 5 ds_write_b128 per thread per k-tile (double buffer)
//   V4 + 5 global b128 loads per thread per k-tile feeding those writes (streams a 1 GB buffer)
// Run with 1..3 workgroups per CU.   hipcc --offload-arch=gfx950 -O3 gemm_ladder.hip -o gemm_ladder
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 256, BN = 64, BK = 16, LDK = 20, ASZ = BM * LDK, BSZ = BN * LDK;

template <int V>
__global__ __launch_bounds__(256) void ladder(const float* __restrict__ g, float* out, int ktiles, long gstride, int nslices) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;
  float* Bs = smem + 2 * ASZ;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, lr = lane & 31, lh = lane >> 5;
  for (int i = tid; i < 2 * (ASZ + BSZ); i += 256) smem[i] = g[i & 4095];
  __syncthreads();
  f32x16 acc[2][2];
  for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  f32x4 fa[2][2], fb[2][2];
  for (int q = 0; q < 2; ++q) for (int i = 0; i < 2; ++i) {
    fa[q][i] = *reinterpret_cast<const f32x4*>(As + ((wid * 2 + i) * 32 + lr) * LDK + q * 8 + lh * 4);
    fb[q][i] = *reinterpret_cast<const f32x4*>(Bs + (i * 32 + lr) * LDK + q * 8 + lh * 4);
  }
  const float* gp = g + ((long)blockIdx.x * 256 + tid) * 4;
  f32x4 ra[5];
  for (int i = 0; i < 5; ++i) ra[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  int cur = 0, slice = 0;
  for (int kt = 0; kt < ktiles; ++kt) {
    if (V >= 4) {
#pragma unroll
      for (int i = 0; i < 5; ++i) {
        ra[i] = *reinterpret_cast<const f32x4*>(gp + (long)slice * gstride);
        if (++slice == nslices) slice = 0;
      }
    }
    const float* Ac = As + cur * ASZ;
    const float* Bc = Bs + cur * BSZ;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      if (V >= 1) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          fa[q][i] = *reinterpret_cast<const f32x4*>(Ac + ((wid * 2 + i) * 32 + lr) * LDK + q * 8 + lh * 4);
          fb[q][i] = *reinterpret_cast<const f32x4*>(Bc + (i * 32 + lr) * LDK + q * 8 + lh * 4);
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[q][a][j], fb[q][b][j], acc[a][b], 0, 0, 0);
    }
    if (V >= 3) {
      float* An = As + (cur ^ 1) * ASZ;
      float* Bn = Bs + (cur ^ 1) * BSZ;
#pragma unroll
      for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(An + ((i * 256 + tid) >> 2) * LDK + ((i * 256 + tid) & 3) * 4) = ra[i];
      *reinterpret_cast<f32x4*>(Bn + (tid >> 2) * LDK + (tid & 3) * 4) = ra[4];
    }
    if (V >= 2) __syncthreads();
    if (V >= 3) cur ^= 1;
  }
  float s = 0.f;
  for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) s += acc[a][b][r];
  out[blockIdx.x * 256 + tid] = s;
}

template <int V>
float run(const float* g, float* out, int blocks, int ktiles, size_t lds, long gstride, int nslices) {
  hipFuncSetAttribute(reinterpret_cast<const void*>(&ladder<V>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(ladder<V>, dim3(blocks), dim3(256), lds, 0, g, out, ktiles, gstride, nslices);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  return best;
}

int main() {
  const long gfloats = 1L << 28;   // 1 GiB
  float *g, *out;
  hipMalloc(&g, gfloats * 4); hipMalloc(&out, 4096 * 256 * 4);
  float* h = (float*)malloc(1 << 22);
  srand(1);
  for (int i = 0; i < (1 << 20); ++i) h[i] = (rand() / (float)RAND_MAX - 0.5f) * 0.01f;
  for (long o = 0; o < gfloats; o += (1 << 20)) hipMemcpy(g + o, h, 1 << 22, hipMemcpyHostToDevice);
  const int ktiles = 2000;
  const size_t lds_min = (size_t)2 * (ASZ + BSZ) * 4;   // 51200 B
  for (int wg_per_cu = 1; wg_per_cu <= 3; ++wg_per_cu) {
    const int blocks = 256 * wg_per_cu;
    const size_t lds = wg_per_cu == 1 ? 160 * 1024 : (wg_per_cu == 2 ? 80 * 1024 : lds_min);   // pins the residency
    const long gstride = (long)blocks * 1024;   // floats between a thread's consecutive loads (whole-grid coalesced sweep)
    float t[5];
    for (long foot : {1L << 28, 1L << 24, 1L << 21}) {      // streamed footprint: 1 GiB (HBM), 64 MiB (MALL), 8 MiB (L2)
    int nslices = (int)(foot / gstride); if (nslices < 1) nslices = 1;
    t[0] = run<0>(g, out, blocks, ktiles, lds, gstride, nslices);
    t[1] = run<1>(g, out, blocks, ktiles, lds, gstride, nslices);
    t[2] = run<2>(g, out, blocks, ktiles, lds, gstride, nslices);
    t[3] = run<3>(g, out, blocks, ktiles, lds, gstride, nslices);
    t[4] = run<4>(g, out, blocks, ktiles, lds, gstride, nslices);
    double fl = (double)blocks * 4 * ktiles * 32 * 4096.0;
    printf("WG/CU %d footprint %4ld MiB:", wg_per_cu, foot >> 18);
    for (int v = 0; v < 5; ++v) printf("  V%d %.1f TF", v, fl / t[v] * 1e-9);
    printf("\n");
    }
  }
  return 0;
}

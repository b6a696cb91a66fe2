// FETCH_SIZE calibration for the access pattern of the convolution kernels (MI355X_MICROARCH.md, HBM section: "on gfx950
// FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming read ... other access widths are uncalibrated:
// calibrate on a known byte count in your own access pattern").  Two kernels read the SAME 1 GiB buffer exactly once:
//   stream    16 bytes per lane, lanes contiguous (the guide's calibrated case)
//   segments  buffer_load_dwordx4 of 64-byte segments: 4 lanes per NHWC pixel (16 of its 64 channels), 16 pixels per
//             wave-instruction, the four channel chunks of a pixel in four successive passes -- the staging pattern of
//             conv3x3_halo_kernel / igemm_kernel<ConvK>
// Run each under `rocprofv3 --pmc FETCH_SIZE` and compare the counter with 1 GiB.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/fetch_calib.hip -o /tmp/fetch_calib
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -d OUT -- /tmp/fetch_calib
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void stream_kernel(const f32x4* __restrict__ p, long n4, float* out) {
  f32x4 acc = {0, 0, 0, 0};
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) acc += p[i];
  if (acc[0] + acc[1] + acc[2] + acc[3] == 1.2345e30f) out[0] = 1.f;
}

__global__ __launch_bounds__(256) void segments_kernel(const float* p, unsigned nbytes, long npix, float* out) {
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, nbytes, 0x00020000);
  f32x4 acc = {0, 0, 0, 0};
  const int lane4 = threadIdx.x & 3;
  for (long px0 = (long)blockIdx.x * 64; px0 < npix; px0 += (long)gridDim.x * 64) {       // 64 pixels per workgroup pass
    const long px = px0 + (threadIdx.x >> 2);
    if (px < npix) {
#pragma unroll
      for (int c = 0; c < 4; ++c)
        acc += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)(px * 256 + c * 64 + lane4 * 16), 0, 0));
    }
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 1.2345e30f) out[0] = 1.f;
}

int main() {
  const long bytes = 1L << 30;
  float *buf, *out;
  hipMalloc(&buf, bytes); hipMalloc(&out, 4);
  hipMemset(buf, 0, bytes);
  for (int it = 0; it < 3; ++it) {
    hipLaunchKernelGGL(stream_kernel, dim3(4096), dim3(256), 0, 0, (const f32x4*)buf, bytes / 16, out);
    hipLaunchKernelGGL(segments_kernel, dim3(4096), dim3(256), 0, 0, buf, (unsigned)bytes, bytes / 256, out);
  }
  hipDeviceSynchronize();
  printf("each kernel read %ld bytes (1 GiB) exactly once per launch, 3 launches each\n", bytes);
  return 0;
}

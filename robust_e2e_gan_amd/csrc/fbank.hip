// K2: fused fbank feature extraction  y = log(max((x^2) . W, 1e-7)) [-> (y + cmvn0) * cmvn1]
// (model/feat_model.py:118-135).  The reference multiplies by a dense (257,80) matrix of which
// only 501 entries are non-zero (<=16 taps per filter); here the matrix is passed banded and the
// op is a banded gather, one thread per output element: HBM-bound (read F, write NF floats per frame).
#include "common.h"

namespace {
constexpr int MAXF = 520;      // max spectrum width held in LDS
constexpr int MAXNF = 128;
constexpr int MAXW = 32;

// per-column sum / sum of squares over valid frames (B,T,NF): one block per (column tile, utterance chunk)
__global__ void cmvn_stats_kernel(const float* __restrict__ y, const int* __restrict__ lens, int B, int T, int NF,
                                  float* __restrict__ sum_out, float* __restrict__ sumsq_out) {
  // single block, deterministic: thread j<NF loops over all valid frames in order
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= NF) return;
  float s = 0.f, q = 0.f;
  for (int b = 0; b < B; ++b) {
    int l = lens[b];
    float sb = 0.f, qb = 0.f;
    for (int t = 0; t < l; ++t) { float v = y[((long)b * T + t) * NF + j]; sb += v; qb += v * v; }
    s += sb; q += qb;
  }
  sum_out[j] = s; sumsq_out[j] = q;
}


// ---- round 3: barrier-free forms.  One thread per OUTPUT element, no LDS staging, no workgroup barrier in the loop: the 1 KB
// spectrum row of a frame is shared by the threads of its filters / bins through the L1, every thread is an independent chain of
// <= 16 (forward) or <= 3 x 16 (backward: the <= 3 filters covering a bin recompute their band power) loads and FMAs.  The LDS-tiled
// kernels above moved 34 / 61 MB at 0.8 / 1.0 TB/s alone (35 / 58 us) and, beside the filler streams on the critical path between the
// VGG backward and the enhancer's backward chain, the backward took 1.1 ms (round-3 trace): 128-thread workgroups meeting at three
// barriers per 4 frames make no progress when their waves are starved for issue slots.
__global__ __launch_bounds__(256) void fbank_fwd_flat_kernel(const float* __restrict__ x, long rows, int F, int NF,
                                                             const int* __restrict__ boff, const int* __restrict__ blen,
                                                             const float* __restrict__ bw, int maxw, float* __restrict__ y_raw,
                                                             float* __restrict__ y_norm, const float* __restrict__ cmvn) {
  const long tot = rows * NF;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += (long)gridDim.x * blockDim.x) {
    const long r = i / NF;
    const int j = (int)(i - r * NF);
    const int o = boff[j], l = blen[j];
    const float* xr = x + r * F + o;
    const float* wr = bw + (long)j * maxw;
    float s = 0.f;
    for (int t = 0; t < l; ++t) { const float v = xr[t]; s += v * v * wr[t]; }
    s = s > 1e-7f ? s : 1e-7f;
    const float lg = __logf(s);
    if (y_raw) y_raw[i] = lg;
    if (y_norm) y_norm[i] = (lg + cmvn[j]) * cmvn[NF + j];
  }
}

__global__ __launch_bounds__(256) void fbank_bwd_flat_kernel(const float* __restrict__ x, long rows, int F, int NF,
                                                             const int* __restrict__ boff, const int* __restrict__ blen,
                                                             const float* __restrict__ bw, int maxw, const float* __restrict__ dy_raw,
                                                             const float* __restrict__ dy_norm, const float* __restrict__ cmvn,
                                                             float* __restrict__ dx) {
  __shared__ int jlo[MAXF], jhi[MAXF];                 // filters covering bin f: a contiguous range of j
  __shared__ int so[MAXNF], sl[MAXNF];
  for (int f = threadIdx.x; f < F; f += blockDim.x) { jlo[f] = NF; jhi[f] = -1; }
  for (int j = threadIdx.x; j < NF; j += blockDim.x) { so[j] = boff[j]; sl[j] = blen[j]; }
  __syncthreads();
  for (int j = threadIdx.x; j < NF; j += blockDim.x)
    for (int t = 0; t < sl[j]; ++t) { atomicMin(&jlo[so[j] + t], j); atomicMax(&jhi[so[j] + t], j); }
  __syncthreads();
  const long tot = rows * F;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += (long)gridDim.x * blockDim.x) {
    const long r = i / F;
    const int f = (int)(i - r * F);
    const float* xr = x + r * F;
    float s = 0.f;
    for (int j = jlo[f]; j <= jhi[f]; ++j) {
      const int o = so[j], l = sl[j], t = f - o;
      if (t < 0 || t >= l) continue;
      const float* wr = bw + (long)j * maxw;
      float pw = 0.f;
      for (int u = 0; u < l; ++u) { const float v = xr[o + u]; pw += v * v * wr[u]; }
      const long oidx = r * NF + j;
      float gy = 0.f;
      if (dy_raw) gy += dy_raw[oidx];
      if (dy_norm) gy += dy_norm[oidx] * cmvn[NF + j];
      s += wr[t] * (pw > 1e-7f ? gy / pw : 0.f);         // in-place clamp => zero gradient (feat_model.py:130)
    }
    dx[i] = 2.f * xr[f] * s;
  }
}
}  // namespace

extern "C" int re2e_fbank_fwd(const float* x, long rows, int F, int NF, const int* band_off, const int* band_len,
                              const float* band_w, int maxw, float* y_raw, float* y_norm, const float* cmvn,
                              hipStream_t stream) {
  RE2E_CHECK_ARG(x && band_off && band_len && band_w && (y_raw || y_norm), "null arg");
  RE2E_CHECK_ARG(rows > 0 && F > 0 && F <= MAXF && NF > 0 && NF <= MAXNF && maxw > 0 && maxw <= MAXW, "shape out of range");
  RE2E_CHECK_ARG(!y_norm || cmvn, "y_norm requires cmvn");
  const long nb = (rows * NF + 255) / 256;
  hipLaunchKernelGGL(fbank_fwd_flat_kernel, dim3((unsigned)(nb < 8192 ? nb : 8192)), dim3(256), 0, stream, x, rows, F, NF, band_off, band_len,
                     band_w, maxw, y_raw, y_norm, cmvn);
  RE2E_LAUNCH_CHECK();
  return RE2E_OK;
}

extern "C" int re2e_fbank_bwd(const float* x, long rows, int F, int NF, const int* band_off, const int* band_len,
                              const float* band_w, int maxw, const float* dy_raw, const float* dy_norm, const float* cmvn,
                              float* dx, hipStream_t stream) {
  RE2E_CHECK_ARG(x && band_off && band_len && band_w && dx && (dy_raw || dy_norm), "null arg");
  RE2E_CHECK_ARG(rows > 0 && F > 0 && F <= MAXF && NF > 0 && NF <= MAXNF && maxw > 0 && maxw <= MAXW, "shape out of range");
  RE2E_CHECK_ARG(!dy_norm || cmvn, "dy_norm requires cmvn");
  const long nb = (rows * F + 255) / 256;
  hipLaunchKernelGGL(fbank_bwd_flat_kernel, dim3((unsigned)(nb < 8192 ? nb : 8192)), dim3(256), 0, stream, x, rows, F, NF, band_off, band_len,
                     band_w, maxw, dy_raw, dy_norm, cmvn, dx);
  RE2E_LAUNCH_CHECK();
  return RE2E_OK;
}

extern "C" int re2e_cmvn_stats(const float* y, const int* lens, int B, int T, int NF, float* sum_out, float* sumsq_out,
                               hipStream_t stream) {
  RE2E_CHECK_ARG(y && lens && sum_out && sumsq_out && B > 0 && T > 0 && NF > 0, "bad args");
  hipLaunchKernelGGL(cmvn_stats_kernel, dim3(cdiv(NF, 64)), dim3(64), 0, stream, y, lens, B, T, NF, sum_out, sumsq_out);
  RE2E_LAUNCH_CHECK();
  return RE2E_OK;
}


// ---- dense (trainable) filterbank, fbank_opti_type 'train' (model/feat_model.py:105-109): the matrix is a full (257,80)
// parameter, so x^2 W and its two gradients run on the GEMM engine (re2e_gemm); these two kernels are the element-wise tail of
// feat_model.py:127-134 around it: y = log(max(z, 1e-7)) [-> (y + cmvn0) * cmvn1], and dz = dy * cmvn1 / z, zero where the
// in-place clamp of :130 fired (z <= 1e-7).
__global__ void logclamp_fwd_kernel(const float* __restrict__ z, const float* __restrict__ cmvn, long rows, int N, float* __restrict__ y) {
  const long tot = rows * N;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += (long)gridDim.x * blockDim.x) {
    const int j = (int)(i % N);
    float v = logf(fmaxf(z[i], 1e-7f));
    if (cmvn) v = (v + cmvn[j]) * cmvn[N + j];
    y[i] = v;
  }
}
__global__ void logclamp_bwd_kernel(const float* __restrict__ z, const float* __restrict__ cmvn, long rows, int N,
                                    const float* __restrict__ dy, float* __restrict__ dz) {
  const long tot = rows * N;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += (long)gridDim.x * blockDim.x) {
    const int j = (int)(i % N);
    const float zz = z[i];
    float g = dy[i];
    if (cmvn) g *= cmvn[N + j];
    dz[i] = zz > 1e-7f ? g / zz : 0.f;
  }
}
extern "C" int re2e_logclamp_fwd(const float* z, const float* cmvn, long rows, int N, float* y, hipStream_t stream) {
  RE2E_CHECK_ARG(z && y && rows > 0 && N > 0, "bad args");
  const long tot = rows * N;
  hipLaunchKernelGGL(logclamp_fwd_kernel, dim3((unsigned)((tot + 255) / 256 > 8192 ? 8192 : (tot + 255) / 256)), dim3(256), 0, stream, z, cmvn, rows, N, y);
  RE2E_LAUNCH_CHECK();
  return RE2E_OK;
}
extern "C" int re2e_logclamp_bwd(const float* z, const float* cmvn, long rows, int N, const float* dy, float* dz, hipStream_t stream) {
  RE2E_CHECK_ARG(z && dy && dz && rows > 0 && N > 0, "bad args");
  const long tot = rows * N;
  hipLaunchKernelGGL(logclamp_bwd_kernel, dim3((unsigned)((tot + 255) / 256 > 8192 ? 8192 : (tot + 255) / 256)), dim3(256), 0, stream, z, cmvn, rows, N, dy, dz);
  RE2E_LAUNCH_CHECK();
  return RE2E_OK;
}

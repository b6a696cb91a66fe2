// K2: fused fbank feature extraction  y = log(max((x^2) . W, 1e-7)) [-> (y + cmvn0) * cmvn1]
// (model/feat_model.py:118-135).  The reference multiplies by a dense (257,80) matrix of which
// only 501 entries are non-zero (<=16 taps per filter); here the matrix is passed banded and the
// op is a gather over an LDS-staged frequency tile: HBM-bound (read F, write NF floats per frame).
#include "common.h"

namespace {
constexpr int RB = 8;          // frames per block iteration
constexpr int MAXF = 520;      // max spectrum width held in LDS
constexpr int MAXNF = 128;
constexpr int MAXW = 32;

__global__ __launch_bounds__(256) void fbank_fwd_kernel(const float* __restrict__ x, long rows, int F, int NF,
                                                        const int* __restrict__ boff, const int* __restrict__ blen,
                                                        const float* __restrict__ bw, int maxw, float* __restrict__ y_raw,
                                                        float* __restrict__ y_norm, const float* __restrict__ cmvn) {
  __shared__ float xs[RB][MAXF];
  __shared__ float ws[MAXNF * MAXW];
  __shared__ int so[MAXNF], sl[MAXNF];
  for (int i = threadIdx.x; i < NF * maxw; i += blockDim.x) ws[i] = bw[i];
  for (int i = threadIdx.x; i < NF; i += blockDim.x) { so[i] = boff[i]; sl[i] = blen[i]; }
  long ngroups = (rows + RB - 1) / RB;
  for (long gi = blockIdx.x; gi < ngroups; gi += gridDim.x) {
    long r0 = gi * RB;
    int nr = (int)((rows - r0) < RB ? (rows - r0) : RB);
    __syncthreads();
    for (int i = threadIdx.x; i < nr * F; i += blockDim.x) {   // contiguous rows: fully coalesced
      float v = x[r0 * F + i];
      xs[i / F][i % F] = v * v;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nr * NF; i += blockDim.x) {
      int r = i / NF, j = i % NF;
      float s = 0.f;
      int o = so[j], l = sl[j];
      for (int t = 0; t < l; ++t) s += xs[r][o + t] * ws[j * maxw + t];
      s = s > 1e-7f ? s : 1e-7f;
      float lg = __logf(s);
      long oidx = (r0 + r) * NF + j;
      if (y_raw) y_raw[oidx] = lg;
      if (y_norm) y_norm[oidx] = (lg + cmvn[j]) * cmvn[NF + j];
    }
  }
}

// Backward.  Small on purpose (128 threads, 4 frames per pass, ~14 KB of LDS, band weights read through the cache): it sits on the
// critical path between two chip-filling phases and must find room beside the filler streams' workgroups.
constexpr int RBB = 4;         // frames per pass of the backward kernel
__global__ __launch_bounds__(128) void fbank_bwd_kernel(const float* __restrict__ x, long rows, int F, int NF,
                                                        const int* __restrict__ boff, const int* __restrict__ blen,
                                                        const float* __restrict__ bw, int maxw, const float* __restrict__ dy_raw,
                                                        const float* __restrict__ dy_norm, const float* __restrict__ cmvn,
                                                        float* __restrict__ dx) {
  __shared__ float xs[RBB][MAXF];     // x (not squared)
  __shared__ float gs[RBB][MAXNF];    // dL/dP_j = gy_j / P_j (0 where clamped)
  __shared__ int so[MAXNF], sl[MAXNF];
  __shared__ int jlo[MAXF], jhi[MAXF];
  for (int i = threadIdx.x; i < NF; i += blockDim.x) { so[i] = boff[i]; sl[i] = blen[i]; }
  for (int f = threadIdx.x; f < F; f += blockDim.x) { jlo[f] = NF; jhi[f] = -1; }
  __syncthreads();
  for (int j = threadIdx.x; j < NF; j += blockDim.x)          // filters covering bin f form a contiguous j range
    for (int t = 0; t < sl[j]; ++t) { atomicMin(&jlo[so[j] + t], j); atomicMax(&jhi[so[j] + t], j); }
  // thread = (row r of the pass, lane q of 32): no integer division anywhere, every global access a 128-byte row segment
  const int r = threadIdx.x >> 5, q = threadIdx.x & 31;
  long ngroups = (rows + RBB - 1) / RBB;
  for (long gi = blockIdx.x; gi < ngroups; gi += gridDim.x) {
    long r0 = gi * RBB;
    int nr = (int)((rows - r0) < RBB ? (rows - r0) : RBB);
    const bool on = r < nr;
    __syncthreads();
    if (on)
      for (int f = q; f < F; f += 32) xs[r][f] = x[(r0 + r) * F + f];
    __syncthreads();
    if (on)
      for (int j = q; j < NF; j += 32) {
        float s = 0.f;
        int o = so[j], l = sl[j];
        for (int t = 0; t < l; ++t) { float v = xs[r][o + t]; s += v * v * bw[j * maxw + t]; }
        long oidx = (r0 + r) * NF + j;
        float gy = 0.f;
        if (dy_raw) gy += dy_raw[oidx];
        if (dy_norm) gy += dy_norm[oidx] * cmvn[NF + j];
        gs[r][j] = s > 1e-7f ? gy / s : 0.f;     // in-place clamp => zero gradient (feat_model.py:130)
      }
    __syncthreads();
    if (on)
      for (int f = q; f < F; f += 32) {
        float s = 0.f;
        for (int j = jlo[f]; j <= jhi[f]; ++j) {
          int t = f - so[j];
          if (t >= 0 && t < sl[j]) s += bw[j * maxw + t] * gs[r][j];
        }
        dx[(r0 + r) * F + f] = 2.f * xs[r][f] * s;
      }
  }
}

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
}  // namespace

extern "C" int re2e_fbank_fwd(const float* x, long rows, int F, int NF, const int* band_off, const int* band_len,
                              const float* band_w, int maxw, float* y_raw, float* y_norm, const float* cmvn,
                              hipStream_t stream) {
  RE2E_CHECK_ARG(x && band_off && band_len && band_w && (y_raw || y_norm), "null arg");
  RE2E_CHECK_ARG(rows > 0 && F > 0 && F <= MAXF && NF > 0 && NF <= MAXNF && maxw > 0 && maxw <= MAXW, "shape out of range");
  RE2E_CHECK_ARG(!y_norm || cmvn, "y_norm requires cmvn");
  long ngroups = (rows + RB - 1) / RB;
  int grid = (int)(ngroups < 2048 ? ngroups : 2048);
  hipLaunchKernelGGL(fbank_fwd_kernel, dim3(grid), dim3(256), 0, stream, x, rows, F, NF, band_off, band_len, band_w, maxw, y_raw,
                     y_norm, cmvn);
  RE2E_LAUNCH_CHECK();
  return RE2E_OK;
}

extern "C" int re2e_fbank_bwd(const float* x, long rows, int F, int NF, const int* band_off, const int* band_len,
                              const float* band_w, int maxw, const float* dy_raw, const float* dy_norm, const float* cmvn,
                              float* dx, hipStream_t stream) {
  RE2E_CHECK_ARG(x && band_off && band_len && band_w && dx && (dy_raw || dy_norm), "null arg");
  RE2E_CHECK_ARG(rows > 0 && F > 0 && F <= MAXF && NF > 0 && NF <= MAXNF && maxw > 0 && maxw <= MAXW, "shape out of range");
  RE2E_CHECK_ARG(!dy_norm || cmvn, "dy_norm requires cmvn");
  long ngroups = (rows + RBB - 1) / RBB;
  int grid = (int)(ngroups < 2048 ? ngroups : 2048);
  hipLaunchKernelGGL(fbank_bwd_kernel, dim3(grid), dim3(128), 0, stream, x, rows, F, NF, band_off, band_len, band_w, maxw, dy_raw,
                     dy_norm, cmvn, dx);
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

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


// ---- round 3: the banded product on the matrix cores.  The LDS-tiled kernels of rounds 1-2 (35 / 58 us alone) spent 0.3-1.1 ms on
// the critical path between the VGG backward and the enhancer's backward chain: beside the register-full filler kernels a
// vector-instruction-heavy kernel gets a fraction of the issue slots, and a thread-per-element form (tried first this round) is
// worse still (181 us alone, ~1 ms in the step).  Here ONE wavefront owns 32 frames and issues almost nothing but MFMAs:
//
//   forward   D[frame][filter] = sum_bin x^2[frame][bin] W[bin][filter], one wave per (32 frames, 32-filter tile), only the bins the
//             tile's filters cover (8 bins per iteration: lane (r, h) of the wave loads x[frame r][8q + 4h + 0..3] and the taps of
//             filter r at those bins -- 4 k-steps of v_mfma_f32_32x32x2_f32); log / CMVN on the accumulators; 128-byte row stores.
//   backward  D2[frame][bin] = sum_filter g[frame][filter] W[bin][filter] with g = dy / band power (0 where the clamp fired), one wave
//             per (32 frames, 32-bin tile), only the filters covering the tile (8 per iteration).  The band power is the forward's
//             (pw_out, saved by the caller): the A operand is computed on the fly from three 16-byte loads, the B operand comes
//             from the TRANSPOSED band table (filters covering a bin); dx = 2 x D2, 128-byte row loads / stores.
// No LDS, no barrier, 64-thread workgroups, <= 64 registers: beside two resident 216-register workgroups of the weight-gradient
// engine (or the 224-register Winograd kernel) a SIMD has 64-80 registers left, and a critical-path kernel that needs more waits for
// the co-resident launch to END -- the first form of this backward (one wave doing forward + backward for its frames in 174
// registers) took 1.29 ms in the step for that reason, 53 us alone.  Out-of-range taps / bins / frames are buffer loads at the
// out-of-range offset (zero, no branch).
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr unsigned FOOB = 0x80000000u;

__device__ __forceinline__ float bload(__amdgpu_buffer_rsrc_t rs, unsigned off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, off, 0, 0));
}
__device__ __forceinline__ int wave_min(int v) {
#pragma unroll
  for (int s = 32; s > 0; s >>= 1) v = min(v, __shfl_xor(v, s, 64));
  return __builtin_amdgcn_readfirstlane(v);
}
__device__ __forceinline__ int wave_max(int v) {
#pragma unroll
  for (int s = 32; s > 0; s >>= 1) v = max(v, __shfl_xor(v, s, 64));
  return __builtin_amdgcn_readfirstlane(v);
}

// band power of one 32-filter tile for 32 frames.  FRAME_ROWS: D[frame][filter] (lane = filter column), else D[filter][frame].
template <bool FRAME_ROWS>
__device__ __forceinline__ f32x16 band_power_tile(__amdgpu_buffer_rsrc_t rsX, __amdgpu_buffer_rsrc_t rsW, int F, int NF, int maxw, int t,
                                                  int lr, int lh, const int* __restrict__ boff, const int* __restrict__ blen) {
  const int j = 32 * t + lr;
  const bool jv = j < NF;
  const int off = jv ? boff[j] : 0, len = jv ? blen[j] : 0;
  const int lo = wave_min(len > 0 ? off : 0x7fffffff), hi = wave_max(len > 0 ? off + len : 0);
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  const unsigned xrow = (unsigned)(lr * F), wrow = (unsigned)(j * maxw);
  // the operands of iteration q + 1 are requested before the MFMAs of iteration q are issued (beyond the tile's last bin the tap
  // lookups are out of range by construction: zeros)
  auto fetch = [&](int q, float (&xv)[4], float (&wv)[4]) {
    const int bin0 = 8 * q + 4 * lh;
#pragma unroll
    for (int i = 0; i < 4; ++i) xv[i] = bload(rsX, bin0 + i < F ? (xrow + bin0 + i) * 4u : FOOB);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = bin0 + i - off;
      wv[i] = bload(rsW, (unsigned)idx < (unsigned)len ? (wrow + idx) * 4u : FOOB);
    }
  };
  float xv[4], wv[4], xn[4], wn[4];
  int q = lo >> 3;
  if (q * 8 < hi) fetch(q, xv, wv);
  for (; q * 8 < hi; ++q) {
    fetch(q + 1, xn, wn);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float pw = xv[i] * xv[i];
      acc = FRAME_ROWS ? __builtin_amdgcn_mfma_f32_32x32x2f32(pw, wv[i], acc, 0, 0, 0) : __builtin_amdgcn_mfma_f32_32x32x2f32(wv[i], pw, acc, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) { xv[i] = xn[i]; wv[i] = wn[i]; }
  }
  return acc;
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t group_rsrc(const float* base, long fg, long rows, int width) {
  const long left = rows - fg * 32;
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base) + fg * 32 * width, 0, (int)((left < 32 ? left : 32) * width * 4), 0x00020000);
}

__global__ __launch_bounds__(64) void fbank_fwd_mfma_kernel(const float* __restrict__ x, long rows, int F, int NF, int nt,
                                                            const int* __restrict__ boff, const int* __restrict__ blen,
                                                            const float* __restrict__ bw, int maxw, float* __restrict__ y_raw,
                                                            float* __restrict__ y_norm, const float* __restrict__ cmvn,
                                                            float* __restrict__ pw_out) {
  const int lr = threadIdx.x & 31, lh = threadIdx.x >> 5;
  const long fg = blockIdx.x / nt;
  const int t = (int)(blockIdx.x - fg * nt);
  const __amdgpu_buffer_rsrc_t rsX = group_rsrc(x, fg, rows, F);
  const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(bw), 0, NF * maxw * 4, 0x00020000);
  const f32x16 acc = band_power_tile<true>(rsX, rsW, F, NF, maxw, t, lr, lh, boff, blen);
  const int j = 32 * t + lr;
  if (j >= NF) return;
  const float c0 = y_norm ? cmvn[j] : 0.f, c1 = y_norm ? cmvn[NF + j] : 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const long frame = fg * 32 + (i & 3) + 8 * (i >> 2) + 4 * lh;
    if (frame >= rows) continue;
    const float lg = __logf(acc[i] <= 1e-7f ? 1e-7f : acc[i]);     // feat_model.py:130 `out[out <= 1e-7] = 1e-7`: a NaN stays a NaN (`acc > c ? acc : c` and fmaxf drop it -- round 4:
                                                                  // a NaN enhancer output came out of here as log(1e-7) and the NaN gate never saw it)
    if (y_raw) y_raw[frame * NF + j] = lg;
    if (y_norm) y_norm[frame * NF + j] = (lg + c0) * c1;
    if (pw_out) pw_out[frame * NF + j] = acc[i];
  }
}

template <bool V4>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(8, 8))) void fbank_bwd_mfma_kernel(const float* __restrict__ x, long rows, int F, int NF, int nmt,
                                                            const int* __restrict__ toff, const int* __restrict__ tlen,
                                                            const float* __restrict__ tw, int maxc, const float* __restrict__ pw,
                                                            const float* __restrict__ dy_raw, const float* __restrict__ dy_norm,
                                                            const float* __restrict__ cmvn, float* __restrict__ dx) {
  const int lr = threadIdx.x & 31, lh = threadIdx.x >> 5;
  const long fg = blockIdx.x / nmt;
  const int mt = (int)(blockIdx.x - fg * nmt);
  const __amdgpu_buffer_rsrc_t rsX = group_rsrc(x, fg, rows, F);
  const __amdgpu_buffer_rsrc_t rsD = group_rsrc(dx, fg, rows, F);
  const __amdgpu_buffer_rsrc_t rsT = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(tw), 0, F * maxc * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsP = group_rsrc(pw, fg, rows, NF);
  const __amdgpu_buffer_rsrc_t rsR = group_rsrc(dy_raw ? dy_raw : x, dy_raw ? fg : 0, dy_raw ? rows : 0, NF);   // absent: every load returns 0
  const __amdgpu_buffer_rsrc_t rsN = group_rsrc(dy_norm ? dy_norm : x, dy_norm ? fg : 0, dy_norm ? rows : 0, NF);
  const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dy_norm ? cmvn : x), 0, dy_norm ? 2 * NF * 4 : 0, 0x00020000);
  const int bin = 32 * mt + lr;
  const bool bv = bin < F;
  const int to = bv ? toff[bin] : 0, tl = bv ? tlen[bin] : 0;
  const int jmin = wave_min(tl > 0 ? to : 0x7fffffff), jmax = wave_max(tl > 0 ? to + tl : 0);
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  const unsigned trow = (unsigned)(bin * maxc), grow = (unsigned)(lr * NF);
  // k-step b of an iteration contracts filters {base + b, base + 4 + b}: lane (r, h) supplies g[frame r][base + 4h + b] as the A operand
  // and W[bin r][base + 4h + b] as the B operand
  auto fetch = [&](int base, f32x4& gr, f32x4& gn, f32x4& pp, f32x4& c1, f32x4& wv) {
    const int j0 = base + 4 * lh;
    if (V4) {                                            // NF % 4 == 0: rows of (frames, NF) tensors are 16-byte aligned
      const unsigned o = j0 < NF ? (grow + j0) * 4u : FOOB;
      gr = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsR, o, 0, 0));
      gn = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsN, o, 0, 0));
      pp = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsP, o, 0, 0));
      c1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsC, j0 < NF ? (unsigned)(NF + j0) * 4u : FOOB, 0, 0));
    } else {
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const unsigned o = j0 + b < NF ? (grow + j0 + b) * 4u : FOOB;
        gr[b] = bload(rsR, o); gn[b] = bload(rsN, o); pp[b] = bload(rsP, o);
        c1[b] = bload(rsC, j0 + b < NF ? (unsigned)(NF + j0 + b) * 4u : FOOB);
      }
    }
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int idx = j0 + b - to;
      wv[b] = bload(rsT, (unsigned)idx < (unsigned)tl ? (trow + idx) * 4u : FOOB);
    }
  };
  f32x4 gr, gn, pp, c1, wv;
  int base = jmin & ~7;
  if (base < jmax) fetch(base, gr, gn, pp, c1, wv);
  for (; base < jmax; base += 8) {
    f32x4 g, w = wv;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const float gy = gr[b] + gn[b] * c1[b];
      g[b] = pp[b] <= 1e-7f ? 0.f : gy / pp[b];          // in-place clamp => zero gradient (feat_model.py:130); a NaN was not clamped
    }
    fetch(base + 8, gr, gn, pp, c1, wv);                 // in flight under the MFMAs; beyond jmax the tap lookups are out of range: zeros
#pragma unroll
    for (int b = 0; b < 4; ++b) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(g[b], w[b], acc, 0, 0, 0);
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int fr = (i & 3) + 8 * (i >> 2) + 4 * lh;
    const unsigned o = bv ? (unsigned)(fr * F + bin) * 4u : FOOB;
    const float xv = bload(rsX, o);
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, 2.f * xv * acc[i]), rsD, o, 0, 0);
  }
}
}  // namespace

extern "C" int re2e_fbank_fwd(const float* x, long rows, int F, int NF, const int* band_off, const int* band_len,
                              const float* band_w, int maxw, float* y_raw, float* y_norm, const float* cmvn, float* pw_out,
                              hipStream_t stream) {
  RE2E_CHECK_ARG(x && band_off && band_len && band_w && (y_raw || y_norm || pw_out), "null arg");
  RE2E_CHECK_ARG(rows > 0 && F > 0 && F <= MAXF && NF > 0 && NF <= MAXNF && maxw > 0 && maxw <= MAXW, "shape out of range");
  RE2E_CHECK_ARG(!y_norm || cmvn, "y_norm requires cmvn");
  const int nt = (NF + 31) / 32;
  const long nwg = (rows + 31) / 32 * nt;
  RE2E_CHECK_ARG(nwg < 0x7fffffffL, "too many frames for one launch");
  hipLaunchKernelGGL(fbank_fwd_mfma_kernel, dim3((unsigned)nwg), dim3(64), 0, stream, x, rows, F, NF, nt, band_off, band_len, band_w, maxw,
                     y_raw, y_norm, cmvn, pw_out);
  RE2E_LAUNCH_CHECK();
  return RE2E_OK;
}

extern "C" int re2e_fbank_bwd(const float* x, long rows, int F, int NF, const int* bin_off, const int* bin_len, const float* bin_w, int maxc,
                              const float* pw, const float* dy_raw, const float* dy_norm, const float* cmvn, float* dx,
                              hipStream_t stream) {
  RE2E_CHECK_ARG(x && bin_off && bin_len && bin_w && pw && dx && (dy_raw || dy_norm), "null arg");
  RE2E_CHECK_ARG(rows > 0 && F > 0 && F <= MAXF && NF > 0 && NF <= MAXNF && maxc > 0 && maxc <= MAXNF, "shape out of range");
  RE2E_CHECK_ARG(!dy_norm || cmvn, "dy_norm requires cmvn");
  const int nmt = (F + 31) / 32;
  const long nwg = (rows + 31) / 32 * nmt;
  RE2E_CHECK_ARG(nwg < 0x7fffffffL, "too many frames for one launch");
  auto aligned16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  const bool v4 = NF % 4 == 0 && aligned16(pw) && (!dy_raw || aligned16(dy_raw)) && (!dy_norm || (aligned16(dy_norm) && aligned16(cmvn)));
  if (v4)
    hipLaunchKernelGGL(fbank_bwd_mfma_kernel<true>, dim3((unsigned)nwg), dim3(64), 0, stream, x, rows, F, NF, nmt, bin_off, bin_len, bin_w, maxc, pw,
                       dy_raw, dy_norm, cmvn, dx);
  else
    hipLaunchKernelGGL(fbank_bwd_mfma_kernel<false>, dim3((unsigned)nwg), dim3(64), 0, stream, x, rows, F, NF, nmt, bin_off, bin_len, bin_w, maxc, pw,
                       dy_raw, dy_norm, cmvn, dx);
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
    float v = logf(z[i] <= 1e-7f ? 1e-7f : z[i]);         // `out[out <= 1e-7] = 1e-7`: a NaN stays a NaN
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
    dz[i] = zz <= 1e-7f ? 0.f : g / zz;
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

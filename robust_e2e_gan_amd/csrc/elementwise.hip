// HBM-bound helpers: layout transposes, activation backward, column sums, enhancer mask backward,
// row gather/scatter, length masking, K1 batch pack/pad, K10 loss reductions, K5 pooling and VGG
// output packing, K9 BatchNorm2d(+LeakyReLU), K11 optimizer.  All reductions are two-stage with a
// fixed tree (bitwise reproducible).  Coalescing rule everywhere: consecutive lanes touch
// consecutive floats of the innermost (feature / channel) axis.
#include "common.h"

namespace {
constexpr int TPB = 256;
inline int grid_for(long n, int per = TPB, int cap = 8192) {
  long g = (n + per - 1) / per;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}
}  // namespace

// ------------------------------------------------------------------------------------------
__global__ void transpose01_kernel(const float* __restrict__ in, float* __restrict__ out, int D0, int D1, int W) {
  long tot = (long)D0 * D1 * W;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += (long)gridDim.x * blockDim.x) {
    int w = (int)(i % W); long r = i / W; int d0 = (int)(r % D0); int d1 = (int)(r / D0);   // out index (d1,d0,w)
    out[i] = in[((long)d0 * D1 + d1) * W + w];
  }
}
extern "C" int re2e_transpose01(const float* in, float* out, int D0, int D1, int W, hipStream_t stream) {
  RE2E_CHECK_ARG(in && out && D0 > 0 && D1 > 0 && W > 0, "bad args");
  long tot = (long)D0 * D1 * W;
  hipLaunchKernelGGL(transpose01_kernel, dim3(grid_for(tot)), dim3(TPB), 0, stream, in, out, D0, D1, W);
  RE2E_LAUNCH_CHECK();
  return RE2E_OK;
}

__global__ void act_bwd_kernel(const float* dy, const float* __restrict__ y, float* dz, long n, int act) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float v = y[i], g = dy[i];
    switch (act) {
      case RE2E_ACT_TANH: g *= (1.f - v * v); break;
      case RE2E_ACT_RELU: g = v > 0.f ? g : 0.f; break;
      case RE2E_ACT_LRELU: g = v > 0.f ? g : 0.2f * g; break;
      case RE2E_ACT_SIGMOID: g *= v * (1.f - v); break;
      default: break;
    }
    dz[i] = g;
  }
}
extern "C" int re2e_act_bwd(const float* dy, const float* y, float* dz, long n, int act, hipStream_t stream) {
  RE2E_CHECK_ARG(dy && y && dz && n > 0, "bad args");
  hipLaunchKernelGGL(act_bwd_kernel, dim3(grid_for(n)), dim3(TPB), 0, stream, dy, y, dz, n, act);
  RE2E_LAUNCH_CHECK();
  return RE2E_OK;
}

// ---- column sums: partial[chunk][N] then fixed-order sum over chunks --------------------------
__global__ void colsum_partial_kernel(const float* __restrict__ A, int M, int N, long lda, int rows_per_chunk,
                                      float* __restrict__ part) {
  __shared__ float red[4][64];
  int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  int col = blockIdx.x * 64 + cx;
  int r0 = blockIdx.y * rows_per_chunk, r1 = min(M, r0 + rows_per_chunk);
  float s = 0.f;
  if (col < N)
    for (int r = r0 + ry; r < r1; r += 4) s += A[(long)r * lda + col];
  red[ry][cx] = s;
  __syncthreads();
  if (ry == 0 && col < N) part[(long)blockIdx.y * N + col] = (red[0][cx] + red[1][cx]) + (red[2][cx] + red[3][cx]);
}
// final stage: CW columns x (256 / CW) chunk lanes per workgroup, lanes combined through LDS in a fixed order.  256-thread workgroups:
// the 800-chunk bias gradients of the BLSTMP projections sit on the critical path of the backward, BESIDE the weight-gradient
// stream's chip-filling tiles, and a 1024-thread workgroup has to wait for a CU with 16 free wave slots and their registers
// (round-3 traces: 205-215 us for 1.6 MB of partials, with 8 as with 32 such workgroups); CW = 64 for short partial lists, 16 for
// long ones (more workgroups, fewer dependent passes per thread).
template <int CW>
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ part, int chunks, int N, float* out, float beta) {
  constexpr int CL = 256 / CW;
  __shared__ float red[CL][CW];
  const int cx = threadIdx.x % CW, cl = threadIdx.x / CW;
  const int col = blockIdx.x * CW + cx;
  float s = 0.f;
  if (col < N) {
    // 8 independent loads in flight per thread (latency-bound otherwise)
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int c = cl;
    for (; c + 7 * CL < chunks; c += 8 * CL) {
#pragma unroll
      for (int u = 0; u < 8; ++u) a[u] += part[(long)(c + CL * u) * N + col];
    }
    for (; c < chunks; c += CL) a[0] += part[(long)c * N + col];
    s = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
  }
  red[cl][cx] = s;
  __syncthreads();
  if (cl != 0 || col >= N) return;
  s = 0.f;
#pragma unroll
  for (int q = 0; q < CL; ++q) s += red[q][cx];
  out[col] = (beta != 0.f ? beta * out[col] : 0.f) + s;
}
static void colsum_final_launch(const float* part, int chunks, int N, float* out, float beta, hipStream_t stream) {
  if (chunks >= 128) hipLaunchKernelGGL(colsum_final_kernel<16>, dim3(cdiv(N, 16)), dim3(256), 0, stream, part, chunks, N, out, beta);
  else hipLaunchKernelGGL(colsum_final_kernel<64>, dim3(cdiv(N, 64)), dim3(256), 0, stream, part, chunks, N, out, beta);
}

// Vectorised partial stage, optionally fused with the activation backward: dz = dy * act'(y) is written and
// its column sums (the bias gradient) accumulated in the same pass, so dz is not re-read from HBM.
// Workgroup = one tile of `ct` columns (ct/4 float4 lanes, a power of two <= 256) x one chunk of rows.
template <bool FUSED>
__global__ __launch_bounds__(256) void colsum_vec_kernel(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ dz,
                                                         int M, int N, int ct, int rows_per_chunk, int act, float* __restrict__ part) {
  __shared__ __attribute__((aligned(16))) float red[256 * 4];
  const int c4t = ct >> 2, c4 = threadIdx.x % c4t, rl = threadIdx.x / c4t, rpb = 256 / c4t;
  const int col = blockIdx.x * ct + c4 * 4;
  const int r0 = blockIdx.y * rows_per_chunk, r1 = min(M, r0 + rows_per_chunk);
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  int r = r0 + rl;
  if (!FUSED) {
    // the plain column sum (bias gradients of the recurrent layers: 105 MB per direction and layer) with EIGHT 16-byte loads in flight per lane
    // and four independent partial sums: at one row per pass (N >= 1024) the two-deep loop below kept 32 bytes per lane in flight and the
    // launches ran at 2.4 TB/s (profiles/r05_bench_nooverlap_kernel_stats.csv: 44 us per 105 MB).  The summation order changes with it
    // (rows r, r + 4 rpb, ... per partial): still a fixed order, bitwise reproducible run to run.
    f32x4 s1 = s, s2 = s, s3 = s;
    const long st = (long)rpb * N;
    for (; r + 7 * rpb < r1; r += 8 * rpb) {
      const float* q = dy + (long)r * N + col;
      const f32x4 g0 = *reinterpret_cast<const f32x4*>(q), g1 = *reinterpret_cast<const f32x4*>(q + st);
      const f32x4 g2 = *reinterpret_cast<const f32x4*>(q + 2 * st), g3 = *reinterpret_cast<const f32x4*>(q + 3 * st);
      const f32x4 g4 = *reinterpret_cast<const f32x4*>(q + 4 * st), g5 = *reinterpret_cast<const f32x4*>(q + 5 * st);
      const f32x4 g6 = *reinterpret_cast<const f32x4*>(q + 6 * st), g7 = *reinterpret_cast<const f32x4*>(q + 7 * st);
      s += g0; s1 += g1; s2 += g2; s3 += g3;
      s += g4; s1 += g5; s2 += g6; s3 += g7;
    }
    s = (s + s1) + (s2 + s3);
  }
#pragma unroll 2
  for (; r < r1; r += rpb) {
    const long off = (long)r * N + col;
    f32x4 g = *reinterpret_cast<const f32x4*>(dy + off);
    if (FUSED) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(y + off);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        switch (act) {
          case RE2E_ACT_TANH: g[j] *= (1.f - v[j] * v[j]); break;
          case RE2E_ACT_RELU: g[j] = v[j] > 0.f ? g[j] : 0.f; break;
          case RE2E_ACT_LRELU: g[j] = v[j] > 0.f ? g[j] : 0.2f * g[j]; break;
          case RE2E_ACT_SIGMOID: g[j] *= v[j] * (1.f - v[j]); break;
          default: break;
        }
      }
      *reinterpret_cast<f32x4*>(dz + off) = g;
    }
    s += g;
  }
  *reinterpret_cast<f32x4*>(red + threadIdx.x * 4) = s;
  __syncthreads();
  if (rl == 0) {
    for (int q = 1; q < rpb; ++q) s += *reinterpret_cast<const f32x4*>(red + (q * c4t + c4) * 4);
    *reinterpret_cast<f32x4*>(part + (long)blockIdx.y * N + col) = s;
  }
}

static inline bool aligned16e(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
// column tile of the vectorised path, or 0 when the shape needs the scalar kernels
static inline int colsum_tile(int N) {
  if (N % 4) return 0;
  if (N <= 1024) { int c4 = N / 4; return (c4 & (c4 - 1)) == 0 ? N : 0; }
  return N % 1024 == 0 ? 1024 : 0;
}
static inline int colsum_chunks(int M, int N) {
  int ct = colsum_tile(N);
  if (!ct) { int c = cdiv(M, 128); return c > 512 ? 512 : (c < 1 ? 1 : c); }
  int rpb = 256 / (ct / 4);
  long c = cdiv(M, (long)rpb * 8);           // >= 8 passes per workgroup
  long cap = 1024 / (N / ct);               // ~1024 workgroups in total (the one-workgroup final stage reads chunks x N partials)
  if (cap < 16) cap = 16;
  return (int)(c < 1 ? 1 : (c > cap ? cap : c));
}
extern "C" size_t re2e_colsum_workspace_bytes(int M, int N) { return (size_t)colsum_chunks(M, N) * N * sizeof(float); }

static void colsum_launch(const float* dy, const float* y, float* dz, int M, int N, long lda, int act, float* out, float beta,
                          float* part, hipStream_t stream) {
  const int chunks = colsum_chunks(M, N);
  const int rpc = cdiv(M, chunks);
  const int ct = colsum_tile(N);
  const bool fused = y != nullptr;
  const bool vec = ct && lda == N && aligned16e(dy) && (!fused || (aligned16e(y) && aligned16e(dz)));
  if (vec) {
    if (fused) hipLaunchKernelGGL(colsum_vec_kernel<true>, dim3(N / ct, chunks), dim3(256), 0, stream, dy, y, dz, M, N, ct, rpc, act, part);
    else hipLaunchKernelGGL(colsum_vec_kernel<false>, dim3(N / ct, chunks), dim3(256), 0, stream, dy, y, dz, M, N, ct, rpc, act, part);
  } else {
    if (fused) {
      long n = (long)M * N;
      hipLaunchKernelGGL(act_bwd_kernel, dim3(grid_for(n)), dim3(TPB), 0, stream, dy, y, dz, n, act);
      dy = dz;
    }
    hipLaunchKernelGGL(colsum_partial_kernel, dim3(cdiv(N, 64), chunks), dim3(256), 0, stream, dy, M, N, lda, rpc, part);
  }
  colsum_final_launch(part, chunks, N, out, beta, stream);
}

extern "C" int re2e_colsum(const float* A, int M, int N, long lda, float* out, float beta, void* workspace,
                           size_t workspace_bytes, hipStream_t stream) {
  RE2E_CHECK_ARG(A && out && workspace && M > 0 && N > 0, "bad args");
  RE2E_CHECK_ARG(workspace_bytes >= (size_t)colsum_chunks(M, N) * N * sizeof(float), "workspace too small");
  colsum_launch(A, nullptr, nullptr, M, N, lda, RE2E_ACT_NONE, out, beta, (float*)workspace, stream);
  RE2E_LAUNCH_CHECK();
  return RE2E_OK;
}

extern "C" int re2e_act_bwd_colsum(const float* dy, const float* y, float* dz, int M, int N, int act, float* out, float beta,
                                   void* workspace, size_t workspace_bytes, hipStream_t stream) {
  RE2E_CHECK_ARG(dy && y && dz && out && workspace && M > 0 && N > 0, "bad args");
  RE2E_CHECK_ARG(act > RE2E_ACT_NONE && act <= RE2E_ACT_SIGMOID, "bad activation");
  RE2E_CHECK_ARG(workspace_bytes >= (size_t)colsum_chunks(M, N) * N * sizeof(float), "workspace too small");
  colsum_launch(dy, y, dz, M, N, N, act, out, beta, (float*)workspace, stream);
  RE2E_LAUNCH_CHECK();
  return RE2E_OK;
}

__global__ void mask_mul_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ mix,
                                    const float* __restrict__ mask, float* __restrict__ dlin, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float m = mask[i];
    dlin[i] = dout[i] * mix[i] * m * (1.f - m);
  }
}
// the same with the result's rows padded to ldd >= N floats (padding written as zeros): an odd width (N = 257) then feeds the
// engine's 16-byte-load path in both products of the layer's backward
__global__ void mask_mul_bwd_ld_kernel(const float* __restrict__ dout, const float* __restrict__ mix, const float* __restrict__ mask,
                                       float* __restrict__ dlin, long rows, int N, int ldd) {
  const long tot = rows * ldd;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += (long)gridDim.x * blockDim.x) {
    const long r = i / ldd;
    const int c = (int)(i - r * ldd);
    float v = 0.f;
    if (c < N) { const long j = r * N + c; const float m = mask[j]; v = dout[j] * mix[j] * m * (1.f - m); }
    dlin[i] = v;
  }
}
extern "C" int re2e_mask_mul_bwd_ld(const float* dout, const float* mix, const float* mask, float* dlin, long rows, int N, int ldd,
                                    hipStream_t stream) {
  RE2E_CHECK_ARG(dout && mix && mask && dlin && rows > 0 && N > 0 && ldd >= N, "bad args");
  hipLaunchKernelGGL(mask_mul_bwd_ld_kernel, dim3(grid_for(rows * ldd)), dim3(TPB), 0, stream, dout, mix, mask, dlin, rows, N, ldd);
  RE2E_LAUNCH_CHECK();
  return RE2E_OK;
}
extern "C" int re2e_mask_mul_bwd(const float* dout, const float* mix, const float* mask, float* dlin, long n,
                                 hipStream_t stream) {
  RE2E_CHECK_ARG(dout && mix && mask && dlin && n > 0, "bad args");
  hipLaunchKernelGGL(mask_mul_bwd_kernel, dim3(grid_for(n)), dim3(TPB), 0, stream, dout, mix, mask, dlin, n);
  RE2E_LAUNCH_CHECK();
  return RE2E_OK;
}

__global__ void axpby_kernel(float a, const float* __restrict__ x, float b, float* y, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    y[i] = a * x[i] + (b != 0.f ? b * y[i] : 0.f);
}
extern "C" int re2e_axpby(float a, const float* x, float b, float* y, long n, hipStream_t stream) {
  RE2E_CHECK_ARG(x && y && n > 0, "bad args");
  hipLaunchKernelGGL(axpby_kernel, dim3(grid_for(n)), dim3(TPB), 0, stream, a, x, b, y, n);
  RE2E_LAUNCH_CHECK();
  return RE2E_OK;
}

// ---- stand-alone activation (the U-Net blocks apply LeakyReLU / ReLU / Sigmoid as separate modules: enhance_model.py:277-291)
__global__ void act_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long n, int act) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) y[i] = apply_act(x[i], act);
}
extern "C" int re2e_act_fwd(const float* x, float* y, long n, int act, hipStream_t stream) {
  RE2E_CHECK_ARG(x && y && n > 0, "bad args");
  RE2E_CHECK_ARG(act > RE2E_ACT_NONE && act <= RE2E_ACT_SIGMOID, "bad activation");
  hipLaunchKernelGGL(act_fwd_kernel, dim3(grid_for(n)), dim3(TPB), 0, stream, x, y, n, act);
  RE2E_LAUNCH_CHECK();
  return RE2E_OK;
}

// ---- dropout (F.dropout / nn.LSTM(dropout=) / nn.Dropout: e2e_ctc.py:51, e2e_encoder.py:156-157, enhance_model.py:298) ----
// Counter-based mask: element i keeps its value when word (i & 3) of Philox4x32-10(counter = {i >> 2 (lo), i >> 34 (hi), call, 0},
// key = {seed lo, seed hi}) is >= threshold = floor(p * 2^32); kept values are scaled by 1 / (1 - p).  Nothing is stored:
// the backward pass regenerates the mask from (seed, call).  oracle/philox.py is the same generator in numpy.
__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1, unsigned (&out)[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
__global__ void dropout_kernel(const float* __restrict__ x, float* __restrict__ y, long n, unsigned thr, float scale, unsigned k0, unsigned k1,
                               unsigned call) {
  const long n4 = (n + 3) >> 2;
  for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < n4; q += (long)gridDim.x * blockDim.x) {
    unsigned w[4];
    philox4x32_10((unsigned)q, (unsigned)((unsigned long long)q >> 32), call, 0u, k0, k1, w);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const long i = q * 4 + j;
      if (i < n) y[i] = w[j] >= thr ? x[i] * scale : 0.f;
    }
  }
}
extern "C" int re2e_dropout(const float* x, float* y, long n, float p, unsigned long long seed, unsigned call, hipStream_t stream) {
  RE2E_CHECK_ARG(x && y && n > 0, "bad args");
  RE2E_CHECK_ARG(p >= 0.f && p < 1.f, "dropout rate must be in [0, 1)");
  const double t = (double)p * 4294967296.0;
  const unsigned thr = t >= 4294967295.0 ? 0xFFFFFFFFu : (unsigned)t;
  hipLaunchKernelGGL(dropout_kernel, dim3(grid_for((n + 3) >> 2)), dim3(TPB), 0, stream, x, y, n, thr, 1.0f / (1.0f - p), (unsigned)seed,
                     (unsigned)(seed >> 32), call);
  RE2E_LAUNCH_CHECK();
  return RE2E_OK;
}

__global__ void mul_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) out[i] = a[i] * b[i];
}
extern "C" int re2e_mul(const float* a, const float* b, float* out, long n, hipStream_t stream) {
  RE2E_CHECK_ARG(a && b && out && n > 0, "bad args");
  hipLaunchKernelGGL(mul_kernel, dim3(grid_for(n)), dim3(TPB), 0, stream, a, b, out, n);
  RE2E_LAUNCH_CHECK();
  return RE2E_OK;
}

// out[r][j] = (x[r][j] + c0[j]) * c1[j]   (c0 == NULL: x*c1 -- the backward form)
__global__ void affine_cols_kernel(const float* __restrict__ x, const float* __restrict__ c0, const float* __restrict__ c1,
                                   float* __restrict__ out, long rows, int N) {
  long tot = rows * N;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += (long)gridDim.x * blockDim.x) {
    int j = (int)(i % N);
    float v = x[i];
    if (c0) v += c0[j];
    out[i] = v * c1[j];
  }
}
extern "C" int re2e_affine_cols(const float* x, const float* c0, const float* c1, float* out, long rows, int N, hipStream_t stream) {
  RE2E_CHECK_ARG(x && c1 && out && rows > 0 && N > 0, "bad args");
  hipLaunchKernelGGL(affine_cols_kernel, dim3(grid_for(rows * N)), dim3(TPB), 0, stream, x, c0, c1, out, rows, N);
  RE2E_LAUNCH_CHECK();
  return RE2E_OK;
}

__global__ void gather_rows_kernel(const float* __restrict__ src, const int* __restrict__ idx, float* __restrict__ dst,
                                   int nrows, int W, int scatter) {
  long tot = (long)nrows * W;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += (long)gridDim.x * blockDim.x) {
    int w = (int)(i % W); int r = (int)(i / W);
    if (scatter) dst[(long)idx[r] * W + w] = src[i];
    else dst[i] = src[(long)idx[r] * W + w];
  }
}
extern "C" int re2e_gather_rows(const float* src, const int* idx, float* dst, int nrows, int W, hipStream_t stream) {
  RE2E_CHECK_ARG(src && idx && dst && nrows > 0 && W > 0, "bad args");
  hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_for((long)nrows * W)), dim3(TPB), 0, stream, src, idx, dst, nrows, W, 0);
  RE2E_LAUNCH_CHECK();
  return RE2E_OK;
}
extern "C" int re2e_scatter_rows(const float* src, const int* idx, float* dst, int nrows, int W, hipStream_t stream) {
  RE2E_CHECK_ARG(src && idx && dst && nrows > 0 && W > 0, "bad args");
  hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_for((long)nrows * W)), dim3(TPB), 0, stream, src, idx, dst, nrows, W, 1);
  RE2E_LAUNCH_CHECK();
  return RE2E_OK;
}

__global__ void mask_rows_kernel(const float* __restrict__ in, float* __restrict__ out, const int* __restrict__ lens,
                                 int B, int T, int W) {
  long tot = (long)B * T * W;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += (long)gridDim.x * blockDim.x) {
    long r = i / W; int t = (int)(r % T); int b = (int)(r / T);
    out[i] = t < lens[b] ? in[i] : 0.f;
  }
}
extern "C" int re2e_mask_rows(const float* in, float* out, const int* lens, int B, int T, int W, hipStream_t stream) {
  RE2E_CHECK_ARG(in && out && lens && B > 0 && T > 0 && W > 0, "bad args");
  hipLaunchKernelGGL(mask_rows_kernel, dim3(grid_for((long)B * T * W)), dim3(TPB), 0, stream, in, out, lens, B, T, W);
  RE2E_LAUNCH_CHECK();
  return RE2E_OK;
}

__global__ void pack_pad_kernel(const float* __restrict__ src, const int* __restrict__ off, const int* __restrict__ lens,
                                int B, int Tmax, int F, float* __restrict__ dst) {
  long tot = (long)B * Tmax * F;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += (long)gridDim.x * blockDim.x) {
    int f = (int)(i % F); long r = i / F; int t = (int)(r % Tmax); int b = (int)(r / Tmax);
    dst[i] = t < lens[b] ? src[((long)off[b] + t) * F + f] : 0.f;
  }
}
extern "C" int re2e_pack_pad(const float* src_flat, const int* offsets, const int* lens, int B, int Tmax, int F, float* dst,
                             hipStream_t stream) {
  RE2E_CHECK_ARG(src_flat && offsets && lens && dst && B > 0 && Tmax > 0 && F > 0, "bad args");
  hipLaunchKernelGGL(pack_pad_kernel, dim3(grid_for((long)B * Tmax * F)), dim3(TPB), 0, stream, src_flat, offsets, lens, B,
                     Tmax, F, dst);
  RE2E_LAUNCH_CHECK();
  return RE2E_OK;
}

// ---- K10 loss reductions ---------------------------------------------------------------------
__device__ __forceinline__ float loss_elem(float d, int kind) {
  if (kind == RE2E_LOSS_L2) return d * d;
  float ad = fabsf(d);
  if (kind == RE2E_LOSS_L1) return ad;
  return ad < 1.f ? 0.5f * d * d : ad - 0.5f;
}
__device__ __forceinline__ float loss_grad(float d, int kind) {
  if (kind == RE2E_LOSS_L2) return 2.f * d;
  if (kind == RE2E_LOSS_L1) return d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
  return fabsf(d) < 1.f ? d : (d > 0.f ? 1.f : -1.f);
}
// RE2E_LOSS_BCE: nn.BCELoss on probabilities (gan_model.py:157-160 with --no_lsgan): -(t log a + (1-t) log(1-a)) with both
// logarithms clamped at -100, gradient (a - t) / max(a (1-a), 1e-12) -- ATen's binary_cross_entropy forward / backward.
__device__ __forceinline__ float loss_elem2(float a, float t, int kind) {
  if (kind != RE2E_LOSS_BCE) return loss_elem(a - t, kind);
  return -(t * fmaxf(logf(a), -100.f) + (1.f - t) * fmaxf(logf(1.f - a), -100.f));
}
__device__ __forceinline__ float loss_grad2(float a, float t, int kind) {
  if (kind != RE2E_LOSS_BCE) return loss_grad(a - t, kind);
  return (a - t) / fmaxf(a * (1.f - a), 1e-12f);
}
static inline int reduce_blocks(long n) { long g = (n + 4095) / 4096; return (int)(g < 1 ? 1 : (g > 1024 ? 1024 : g)); }
extern "C" size_t re2e_reduce_workspace_bytes(long n) { return (size_t)reduce_blocks(n) * sizeof(float); }

__global__ void loss_partial_kernel(const float* __restrict__ a, const float* __restrict__ b, float target, long n,
                                    int kind, float* __restrict__ part) {
  __shared__ float red[16];
  float s = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    s += loss_elem2(a[i], b ? b[i] : target, kind);
  s = block_sum(s, red);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}
__global__ void sum_final_kernel(const float* __restrict__ part, int np, float scale, float* out) {
  __shared__ float red[16];
  float s = 0.f;
  for (int i = threadIdx.x; i < np; i += blockDim.x) s += part[i];
  s = block_sum(s, red);
  if (threadIdx.x == 0) out[0] = s * scale;
}
extern "C" int re2e_loss_fwd(const float* a, const float* b, float target, long n, int kind, float* out, void* workspace,
                             size_t workspace_bytes, hipStream_t stream) {
  RE2E_CHECK_ARG(a && out && workspace && n > 0, "bad args");
  int nb = reduce_blocks(n);
  RE2E_CHECK_ARG(workspace_bytes >= (size_t)nb * sizeof(float), "workspace too small");
  hipLaunchKernelGGL(loss_partial_kernel, dim3(nb), dim3(256), 0, stream, a, b, target, n, kind, (float*)workspace);
  hipLaunchKernelGGL(sum_final_kernel, dim3(1), dim3(256), 0, stream, (const float*)workspace, nb, 1.0f / (float)n, out);
  RE2E_LAUNCH_CHECK();
  return RE2E_OK;
}
__global__ void loss_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b, float target, long n, int kind,
                                const float* __restrict__ gscale, float scale, float* da, float beta) {
  float g = (gscale ? gscale[0] : 1.f) * scale / (float)n;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float v = g * loss_grad2(a[i], b ? b[i] : target, kind);
    da[i] = (beta != 0.f ? beta * da[i] : 0.f) + v;
  }
}
extern "C" int re2e_loss_bwd(const float* a, const float* b, float target, long n, int kind, const float* gscale, float scale,
                             float* da, float beta, hipStream_t stream) {
  RE2E_CHECK_ARG(a && da && n > 0, "bad args");
  hipLaunchKernelGGL(loss_bwd_kernel, dim3(grid_for(n)), dim3(TPB), 0, stream, a, b, target, n, kind, gscale, scale, da, beta);
  RE2E_LAUNCH_CHECK();
  return RE2E_OK;
}
// 16-byte loads, four of them in flight per thread (the 117 MB ASR gradient buffer: 21 us per launch = 2.2 TB/s with one
// 4-byte load in flight, round 2); fixed grid and fixed per-thread order: deterministic
__global__ void sumsq_partial_kernel(const float* __restrict__ x, long n, float* __restrict__ part) {
  __shared__ float red[16];
  const long n4 = ((reinterpret_cast<uintptr_t>(x) & 15) == 0) ? n >> 2 : 0;
  const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
  const long stride = (long)gridDim.x * blockDim.x;
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
  for (; i + 3 * stride < n4; i += 4 * stride) {
    const f32x4 v0 = x4[i], v1 = x4[i + stride], v2 = x4[i + 2 * stride], v3 = x4[i + 3 * stride];
    a0 += v0 * v0; a1 += v1 * v1; a2 += v2 * v2; a3 += v3 * v3;
  }
  for (; i < n4; i += stride) { const f32x4 v = x4[i]; a0 += v * v; }
  const f32x4 a = (a0 + a1) + (a2 + a3);
  float s = (a[0] + a[1]) + (a[2] + a[3]);
  for (long j = 4 * n4 + (long)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += stride) s += x[j] * x[j];     // tail (or unaligned)
  s = block_sum(s, red);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}
extern "C" int re2e_sumsq(const float* x, long n, float* out, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  RE2E_CHECK_ARG(x && out && workspace && n > 0, "bad args");
  int nb = reduce_blocks(n);
  RE2E_CHECK_ARG(workspace_bytes >= (size_t)nb * sizeof(float), "workspace too small");
  hipLaunchKernelGGL(sumsq_partial_kernel, dim3(nb), dim3(256), 0, stream, x, n, (float*)workspace);
  hipLaunchKernelGGL(sum_final_kernel, dim3(1), dim3(256), 0, stream, (const float*)workspace, nb, 1.0f, out);
  RE2E_LAUNCH_CHECK();
  return RE2E_OK;
}

// ---- K5 max pool 2x2 stride 2 ceil mode over NHWC -------------------------------------------
// relu_in: the input is the output of a ReLU whose backward this pool's backward also performs -- a window whose maximum is <= 0 gets
// index 4, which no position matches, so its gradient is dropped exactly as dy * (y > 0) would drop it
__global__ void maxpool2_fwd_kernel(const float* __restrict__ in, int NI, int H, int W, int C, float* __restrict__ out,
                                    unsigned char* __restrict__ idx, int relu_in) {
  int OH = (H + 1) / 2, OW = (W + 1) / 2;
  long tot = (long)NI * OH * OW * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += (long)gridDim.x * blockDim.x) {
    int c = (int)(i % C); long r = i / C; int ox = (int)(r % OW); r /= OW; int oy = (int)(r % OH); int n = (int)(r / OH);
    float best = -3.0e38f; int bi = 0;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      int iy = oy * 2 + (d >> 1), ix = ox * 2 + (d & 1);
      if (iy < H && ix < W) {
        float v = in[(((long)n * H + iy) * W + ix) * C + c];
        if (v > best) { best = v; bi = d; }     // first max wins (PyTorch order: row-major scan)
      }
    }
    out[i] = best; idx[i] = (unsigned char)((relu_in && !(best > 0.f)) ? 4 : bi);
  }
}
// four channels per thread (C % 4 == 0, 16-byte aligned tensors): float4 loads / stores, one index word per thread,
// a quarter of the index arithmetic -- these kernels move 0.7-1.2 GB per call at config 4
typedef unsigned char uchar4v __attribute__((ext_vector_type(4)));
__global__ void maxpool2_fwd_vec_kernel(const float* __restrict__ in, int NI, int H, int W, int C4, float* __restrict__ out,
                                        unsigned char* __restrict__ idx, int relu_in) {
  const int OH = (H + 1) / 2, OW = (W + 1) / 2;
  const long tot = (long)NI * OH * OW * C4;
  const f32x4* in4 = reinterpret_cast<const f32x4*>(in);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += (long)gridDim.x * blockDim.x) {
    int c = (int)(i % C4); long r = i / C4; int ox = (int)(r % OW); r /= OW; int oy = (int)(r % OH); int n = (int)(r / OH);
    f32x4 best = {-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f};
    int bi[4] = {0, 0, 0, 0};
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      int iy = oy * 2 + (d >> 1), ix = ox * 2 + (d & 1);
      if (iy < H && ix < W) {
        const f32x4 v = in4[(((long)n * H + iy) * W + ix) * C4 + c];
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (v[k] > best[k]) { best[k] = v[k]; bi[k] = d; }     // first max wins (PyTorch order: row-major scan)
      }
    }
    reinterpret_cast<f32x4*>(out)[i] = best;
    if (relu_in) {
#pragma unroll
      for (int k = 0; k < 4; ++k) bi[k] = best[k] > 0.f ? bi[k] : 4;
    }
    uchar4v b = {(unsigned char)bi[0], (unsigned char)bi[1], (unsigned char)bi[2], (unsigned char)bi[3]};
    reinterpret_cast<uchar4v*>(idx)[i] = b;
  }
}
__global__ void maxpool2_bwd_vec_kernel(const float* __restrict__ dout, const unsigned char* __restrict__ idx, int NI, int H,
                                        int W, int C4, float* __restrict__ din) {
  const int OH = (H + 1) / 2, OW = (W + 1) / 2;
  const long tot = (long)NI * H * W * C4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += (long)gridDim.x * blockDim.x) {
    int c = (int)(i % C4); long r = i / C4; int ix = (int)(r % W); r /= W; int iy = (int)(r % H); int n = (int)(r / H);
    const int oy = iy >> 1, ox = ix >> 1, d = ((iy & 1) << 1) | (ix & 1);
    const long o = (((long)n * OH + oy) * OW + ox) * C4 + c;
    const f32x4 g = reinterpret_cast<const f32x4*>(dout)[o];
    const uchar4v b = reinterpret_cast<const uchar4v*>(idx)[o];
    f32x4 v;
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = (b[k] == d) ? g[k] : 0.f;
    reinterpret_cast<f32x4*>(din)[i] = v;
  }
}
static inline bool pool_vec_ok(const void* a, const void* b, const void* c, int C) {
  return C % 4 == 0 && ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) & 15) == 0 && (reinterpret_cast<uintptr_t>(c) & 3) == 0;
}
extern "C" int re2e_maxpool2_fwd(const float* in, int NI, int H, int W, int C, float* out, unsigned char* idx, int relu_in,
                                 hipStream_t stream) {
  RE2E_CHECK_ARG(in && out && idx && NI > 0 && H > 0 && W > 0 && C > 0, "bad args");
  long tot = (long)NI * ((H + 1) / 2) * ((W + 1) / 2) * C;
  if (pool_vec_ok(in, out, idx, C)) {
    hipLaunchKernelGGL(maxpool2_fwd_vec_kernel, dim3(grid_for(tot / 4)), dim3(TPB), 0, stream, in, NI, H, W, C / 4, out, idx, relu_in);
    RE2E_LAUNCH_CHECK();
    return RE2E_OK;
  }
  hipLaunchKernelGGL(maxpool2_fwd_kernel, dim3(grid_for(tot)), dim3(TPB), 0, stream, in, NI, H, W, C, out, idx, relu_in);
  RE2E_LAUNCH_CHECK();
  return RE2E_OK;
}
__global__ void maxpool2_bwd_kernel(const float* __restrict__ dout, const unsigned char* __restrict__ idx, int NI, int H,
                                    int W, int C, float* __restrict__ din) {
  int OH = (H + 1) / 2, OW = (W + 1) / 2;
  long tot = (long)NI * H * W * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += (long)gridDim.x * blockDim.x) {
    int c = (int)(i % C); long r = i / C; int ix = (int)(r % W); r /= W; int iy = (int)(r % H); int n = (int)(r / H);
    int oy = iy >> 1, ox = ix >> 1; int d = ((iy & 1) << 1) | (ix & 1);
    long o = (((long)n * OH + oy) * OW + ox) * C + c;
    din[i] = (idx[o] == d) ? dout[o] : 0.f;
  }
}
extern "C" int re2e_maxpool2_bwd(const float* dout, const unsigned char* idx, int NI, int H, int W, int C, float* din,
                                 hipStream_t stream) {
  RE2E_CHECK_ARG(dout && idx && din && NI > 0 && H > 0 && W > 0 && C > 0, "bad args");
  long tot = (long)NI * H * W * C;
  if (pool_vec_ok(dout, din, idx, C)) {
    hipLaunchKernelGGL(maxpool2_bwd_vec_kernel, dim3(grid_for(tot / 4)), dim3(TPB), 0, stream, dout, idx, NI, H, W, C / 4, din);
    RE2E_LAUNCH_CHECK();
    return RE2E_OK;
  }
  hipLaunchKernelGGL(maxpool2_bwd_kernel, dim3(grid_for(tot)), dim3(TPB), 0, stream, dout, idx, NI, H, W, C, din);
  RE2E_LAUNCH_CHECK();
  return RE2E_OK;
}

// NHWC (NI,T,Fq,C) -> rows [n_off, n_off+NI) of a time-major (T,NI_total,C*Fq) tensor: feature index c*Fq+f
// (e2e_encoder.py:274-276), zero for t>=lens[n].  The batch offset lets two branches computed on different
// streams land in one (T, 2B, .) tensor for the shared BLSTMP.
__global__ void vgg_pack_kernel(const float* __restrict__ src, const int* __restrict__ lens, int NI, int T, int Fq, int C,
                                float* __restrict__ dst, int backward, int NItot, int noff) {
  long tot = (long)T * NI * C * Fq;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += (long)gridDim.x * blockDim.x) {
    if (!backward) {       // i indexes (t, n, c, f) of this branch: coalesced writes
      int f = (int)(i % Fq); long r = i / Fq; int c = (int)(r % C); r /= C; int n = (int)(r % NI); int t = (int)(r / NI);
      dst[(((long)t * NItot + noff + n) * C + c) * Fq + f] = t < lens[n] ? src[(((long)n * T + t) * Fq + f) * C + c] : 0.f;
    } else {               // i indexes din (n, t, f, c): coalesced writes
      int c = (int)(i % C); long r = i / C; int f = (int)(r % Fq); r /= Fq; int t = (int)(r % T); int n = (int)(r / T);
      dst[i] = t < lens[n] ? src[(((long)t * NItot + noff + n) * C + c) * Fq + f] : 0.f;
    }
  }
}
// The same through an LDS tile: one workgroup per (t, n) transposes that frame's (Fq, C) block, so that BOTH the global read and
// the global write are contiguous runs of Fq*C floats (the element-wise form reads 4 bytes every C floats).
__global__ __launch_bounds__(256) void vgg_pack_tile_kernel(const float* __restrict__ src, const int* __restrict__ lens, int NI, int T,
                                                            int Fq, int C, float* __restrict__ dst, int backward, int NItot, int noff) {
  extern __shared__ float tile[];            // [Fq][C + 1]
  const int n = blockIdx.x % NI, t = blockIdx.x / NI;
  const int FC = Fq * C, CP = C + 1;
  const bool live = t < lens[n];
  const long nhwc = ((long)n * T + t) * FC;                       // (n, t, f, c)
  const long tm = ((long)t * NItot + noff + n) * FC;              // (t, n, c, f)
  if (!backward) {
    if (live)
      for (int e = threadIdx.x; e < FC; e += 256) tile[(e / C) * CP + (e % C)] = src[nhwc + e];
    __syncthreads();
    for (int e = threadIdx.x; e < FC; e += 256) dst[tm + e] = live ? tile[(e % Fq) * CP + (e / Fq)] : 0.f;
  } else {
    if (live)
      for (int e = threadIdx.x; e < FC; e += 256) tile[(e % Fq) * CP + (e / Fq)] = src[tm + e];
    __syncthreads();
    for (int e = threadIdx.x; e < FC; e += 256) dst[nhwc + e] = live ? tile[(e / C) * CP + (e % C)] : 0.f;
  }
}
static bool vgg_pack_tiled(const float* src, const int* lens, int NI, int T, int Fq, int C, float* dst, int backward, int NItot, int noff,
                           hipStream_t stream) {
  const size_t lds = (size_t)Fq * (C + 1) * sizeof(float);
  if (lds > 48 * 1024 || (long)T * NI > 2000000000L) return false;
  hipLaunchKernelGGL(vgg_pack_tile_kernel, dim3(T * NI), dim3(256), lds, stream, src, lens, NI, T, Fq, C, dst, backward, NItot, noff);
  return true;
}

extern "C" int re2e_vgg_pack_fwd(const float* in, const int* lens, int NI, int T, int Fq, int C, float* out, int NI_total, int n_off,
                                 hipStream_t stream) {
  RE2E_CHECK_ARG(in && lens && out && NI > 0 && T > 0 && Fq > 0 && C > 0 && n_off >= 0 && n_off + NI <= NI_total, "bad args");
  if (!vgg_pack_tiled(in, lens, NI, T, Fq, C, out, 0, NI_total, n_off, stream))
    hipLaunchKernelGGL(vgg_pack_kernel, dim3(grid_for((long)T * NI * C * Fq)), dim3(TPB), 0, stream, in, lens, NI, T, Fq, C, out, 0, NI_total, n_off);
  RE2E_LAUNCH_CHECK();
  return RE2E_OK;
}
extern "C" int re2e_vgg_pack_bwd(const float* dout, const int* lens, int NI, int T, int Fq, int C, float* din, int NI_total, int n_off,
                                 hipStream_t stream) {
  RE2E_CHECK_ARG(dout && lens && din && NI > 0 && T > 0 && Fq > 0 && C > 0 && n_off >= 0 && n_off + NI <= NI_total, "bad args");
  if (!vgg_pack_tiled(dout, lens, NI, T, Fq, C, din, 1, NI_total, n_off, stream))
    hipLaunchKernelGGL(vgg_pack_kernel, dim3(grid_for((long)T * NI * C * Fq)), dim3(TPB), 0, stream, dout, lens, NI, T, Fq, C, din, 1, NI_total, n_off);
  RE2E_LAUNCH_CHECK();
  return RE2E_OK;
}

// ---- K9 BatchNorm2d (train) + LeakyReLU(slope) over rows [P][C]; slope = 0.2 in the discriminator (gan_model.py:76-88),
// 1.0 = plain BatchNorm2d (the U-Net blocks: the next block applies its own pre-activation, enhance_model.py:281-296)
// partial sums over row chunks: mode 0: sum(x - center) ; mode 1: sum((x-center)^2)
// mode 2 (backward): s0 = sum dz, s1 = sum dz*xhat with dz = dy * lrelu'(bn(x))
static inline int bn_chunks(long P) { long c = (P + 255) / 256; return (int)(c > 512 ? 512 : (c < 1 ? 1 : c)); }
extern "C" size_t re2e_bn_workspace_bytes(long P, int C) { return (size_t)bn_chunks(P) * 2 * C * sizeof(float) + 4 * C * sizeof(float); }

__global__ void bn_partial_kernel(const float* __restrict__ x, const float* __restrict__ dy, long P, int C, long rows_per_chunk,
                                  int mode, const float* __restrict__ mean, const float* __restrict__ invstd,
                                  const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ part, float slope) {
  __shared__ float r0[4][64], r1[4][64];
  int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  int c = blockIdx.x * 64 + cx;
  long p0 = (long)blockIdx.y * rows_per_chunk, p1 = p0 + rows_per_chunk; if (p1 > P) p1 = P;
  float s0 = 0.f, s1 = 0.f;
  if (c < C) {
    float mu = mean ? mean[c] : 0.f;
    float is = invstd ? invstd[c] : 1.f, ga = gamma ? gamma[c] : 1.f, be = beta ? beta[c] : 0.f;
    for (long p = p0 + ry; p < p1; p += 4) {
      float v = x[p * C + c] - mu;
      if (mode == 0) s0 += v;
      else if (mode == 1) s0 += v * v;
      else {
        float xh = v * is; float y = xh * ga + be; float dz = dy[p * C + c]; dz = y > 0.f ? dz : slope * dz;
        s0 += dz; s1 += dz * xh;
      }
    }
  }
  r0[ry][cx] = s0; r1[ry][cx] = s1;
  __syncthreads();
  if (ry == 0 && c < C) {
    part[((long)blockIdx.y * 2 + 0) * C + c] = (r0[0][cx] + r0[1][cx]) + (r0[2][cx] + r0[3][cx]);
    part[((long)blockIdx.y * 2 + 1) * C + c] = (r1[0][cx] + r1[1][cx]) + (r1[2][cx] + r1[3][cx]);
  }
}
// the same for C % 4 == 0: four channels per thread (float4), 16 row lanes per workgroup, two rows in flight per thread
__global__ __launch_bounds__(256) void bn_partial_vec_kernel(const float* __restrict__ x, const float* __restrict__ dy, long P, int C,
                                                             long rows_per_chunk, int mode, const float* __restrict__ mean,
                                                             const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float* __restrict__ part, float slope) {
  __shared__ __attribute__((aligned(16))) float r0[16][64], r1[16][64];
  const int c4 = threadIdx.x & 15, ry = threadIdx.x >> 4;
  const int c = blockIdx.x * 64 + c4 * 4;
  long p0 = (long)blockIdx.y * rows_per_chunk, p1 = p0 + rows_per_chunk; if (p1 > P) p1 = P;
  f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
  if (c < C) {
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f}, one = {1.f, 1.f, 1.f, 1.f};
    const f32x4 mu = mean ? *reinterpret_cast<const f32x4*>(mean + c) : zero;
    const f32x4 is = invstd ? *reinterpret_cast<const f32x4*>(invstd + c) : one;
    const f32x4 ga = gamma ? *reinterpret_cast<const f32x4*>(gamma + c) : one;
    const f32x4 be = beta ? *reinterpret_cast<const f32x4*>(beta + c) : zero;
#define RE2E_BN_ROW(xv, dv)                                                                   \
  do {                                                                                        \
    const f32x4 v_ = (xv) - mu;                                                               \
    if (mode == 0) s0 += v_;                                                                  \
    else if (mode == 1) s0 += v_ * v_;                                                        \
    else {                                                                                    \
      const f32x4 xh_ = v_ * is;                                                              \
      const f32x4 y_ = xh_ * ga + be;                                                         \
      f32x4 dz_;                                                                              \
      dz_[0] = y_[0] > 0.f ? (dv)[0] : slope * (dv)[0]; dz_[1] = y_[1] > 0.f ? (dv)[1] : slope * (dv)[1]; \
      dz_[2] = y_[2] > 0.f ? (dv)[2] : slope * (dv)[2]; dz_[3] = y_[3] > 0.f ? (dv)[3] : slope * (dv)[3]; \
      s0 += dz_; s1 += dz_ * xh_;                                                             \
    }                                                                                         \
  } while (0)
    long p = p0 + ry;
    for (; p + 16 < p1; p += 32) {
      const f32x4 xa = *reinterpret_cast<const f32x4*>(x + p * C + c), xb = *reinterpret_cast<const f32x4*>(x + (p + 16) * C + c);
      f32x4 da = zero, db = zero;
      if (mode == 2) { da = *reinterpret_cast<const f32x4*>(dy + p * C + c); db = *reinterpret_cast<const f32x4*>(dy + (p + 16) * C + c); }
      RE2E_BN_ROW(xa, da); RE2E_BN_ROW(xb, db);
    }
    for (; p < p1; p += 16) {
      const f32x4 xa = *reinterpret_cast<const f32x4*>(x + p * C + c);
      const f32x4 da = mode == 2 ? *reinterpret_cast<const f32x4*>(dy + p * C + c) : zero;
      RE2E_BN_ROW(xa, da);
    }
#undef RE2E_BN_ROW
  }
  *reinterpret_cast<f32x4*>(&r0[ry][c4 * 4]) = s0;
  *reinterpret_cast<f32x4*>(&r1[ry][c4 * 4]) = s1;
  __syncthreads();
  const int cx = threadIdx.x;
  if (cx < 64 && blockIdx.x * 64 + cx < C) {
    float a0 = 0.f, a1 = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) { a0 += r0[q][cx]; a1 += r1[q][cx]; }
    part[((long)blockIdx.y * 2 + 0) * C + blockIdx.x * 64 + cx] = a0;
    part[((long)blockIdx.y * 2 + 1) * C + blockIdx.x * 64 + cx] = a1;
  }
}
static inline bool bn_vec_ok(const float* x, const float* dy, int C) {
  return C % 4 == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy)) & 15) == 0;
}
// stats[0..C) = result of slot 0 summed over chunks * scale0 ; stats[C..2C) = slot 1 * scale1
// 64 channels x 16 chunk lanes per workgroup, lanes combined through LDS in a fixed order (the serial loop over up to
// 512 chunks made this tiny kernel 60 us)
__global__ __launch_bounds__(1024) void bn_combine_kernel(const float* __restrict__ part, int chunks, int C, float scale0, float scale1,
                                                          float* out0, float* out1) {
  __shared__ float r0[16][64], r1[16][64];
  const int cx = threadIdx.x & 63, cl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cx;
  float s0 = 0.f, s1 = 0.f;
  if (c < C) {
    // 4 chunk rows x 2 slots in flight per thread: one or two workgroups read all the partials, so the loop is latency-bound
    float a0[4] = {0.f, 0.f, 0.f, 0.f}, a1[4] = {0.f, 0.f, 0.f, 0.f};
    int k = cl;
    for (; k + 3 * 16 < chunks; k += 4 * 16) {
#pragma unroll
      for (int u = 0; u < 4; ++u) { a0[u] += part[((long)(k + 16 * u) * 2 + 0) * C + c]; a1[u] += part[((long)(k + 16 * u) * 2 + 1) * C + c]; }
    }
    for (; k < chunks; k += 16) { a0[0] += part[((long)k * 2 + 0) * C + c]; a1[0] += part[((long)k * 2 + 1) * C + c]; }
    s0 = (a0[0] + a0[1]) + (a0[2] + a0[3]);
    s1 = (a1[0] + a1[1]) + (a1[2] + a1[3]);
  }
  r0[cl][cx] = s0; r1[cl][cx] = s1;
  __syncthreads();
  if (cl != 0 || c >= C) return;
  s0 = 0.f; s1 = 0.f;
#pragma unroll
  for (int q = 0; q < 16; ++q) { s0 += r0[q][cx]; s1 += r1[q][cx]; }
  if (out0) out0[c] = s0 * scale0;
  if (out1) out1[c] = s1 * scale1;
}
__global__ void bn_finalize_kernel(const float* __restrict__ mean, const float* __restrict__ var_b, long P, int C, float momentum,
                                   float eps, float* running_mean, float* running_var, float* save_mean, float* save_invstd) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float m = mean[c], v = var_b[c];
  save_mean[c] = m; save_invstd[c] = rsqrtf(v + eps);
  float unb = P > 1 ? v * ((float)P / (float)(P - 1)) : v;
  // Batch statistics that are not finite (features poisoned by a persistent recurrence that gave up, csrc/lstm.hip) leave the running
  // statistics alone: the trainer repeats such a step, and a NaN written here would outlive the repeat (torch would write it; with a
  // finite batch the arithmetic is torch's)
  if (!(fabsf(m) < 3.0e38f) || !(fabsf(unb) < 3.0e38f)) return;
  running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * m;
  running_var[c] = (1.f - momentum) * running_var[c] + momentum * unb;
}
__global__ void bn_eval_stats_kernel(const float* rm, const float* rv, int C, float eps, float* save_mean, float* save_invstd) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  save_mean[c] = rm[c]; save_invstd[c] = rsqrtf(rv[c] + eps);
}
__global__ void bn_apply_kernel(const float* __restrict__ x, long P, int C, const float* __restrict__ mean,
                                const float* __restrict__ invstd, const float* __restrict__ gamma, const float* __restrict__ beta,
                                float* __restrict__ y, float slope) {
  long tot = P * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += (long)gridDim.x * blockDim.x) {
    int c = (int)(i % C);
    float v = (x[i] - mean[c]) * invstd[c] * gamma[c] + beta[c];
    y[i] = v > 0.f ? v : slope * v;
  }
}
// float4 forms of the two apply kernels (C % 4 == 0): the channel vectors are loaded once per thread iteration
__global__ void bn_apply_vec_kernel(const float* __restrict__ x, long P, int C4, const float* __restrict__ mean,
                                    const float* __restrict__ invstd, const float* __restrict__ gamma, const float* __restrict__ beta,
                                    float* __restrict__ y, float slope) {
  const long tot = P * C4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4);
    const f32x4 m = reinterpret_cast<const f32x4*>(mean)[c], is = reinterpret_cast<const f32x4*>(invstd)[c];
    const f32x4 ga = reinterpret_cast<const f32x4*>(gamma)[c], be = reinterpret_cast<const f32x4*>(beta)[c];
    f32x4 v = (reinterpret_cast<const f32x4*>(x)[i] - m) * is * ga + be;
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = v[k] > 0.f ? v[k] : slope * v[k];
    reinterpret_cast<f32x4*>(y)[i] = v;
  }
}
__global__ void bn_bwd_apply_vec_kernel(const float* __restrict__ dy, const float* __restrict__ x, long P, int C4,
                                        const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ gamma,
                                        const float* __restrict__ beta, const float* __restrict__ sdz, const float* __restrict__ sdzx,
                                        float* __restrict__ dx, float slope) {
  const long tot = P * C4;
  const float invP = 1.0f / (float)P;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4);
    const f32x4 m = reinterpret_cast<const f32x4*>(mean)[c], is = reinterpret_cast<const f32x4*>(invstd)[c];
    const f32x4 ga = reinterpret_cast<const f32x4*>(gamma)[c], be = reinterpret_cast<const f32x4*>(beta)[c];
    const f32x4 a = reinterpret_cast<const f32x4*>(sdz)[c], bq = reinterpret_cast<const f32x4*>(sdzx)[c];
    const f32x4 xh = (reinterpret_cast<const f32x4*>(x)[i] - m) * is;
    const f32x4 yv = xh * ga + be;
    f32x4 dz = reinterpret_cast<const f32x4*>(dy)[i];
#pragma unroll
    for (int k = 0; k < 4; ++k) dz[k] = yv[k] > 0.f ? dz[k] : slope * dz[k];
    reinterpret_cast<f32x4*>(dx)[i] = ga * is * (dz - a * invP - xh * bq * invP);
  }
}
static inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

extern "C" int re2e_bn_lrelu_fwd(const float* x, long P, int C, const float* gamma, const float* beta, float* running_mean,
                                 float* running_var, float momentum, float eps, int train, float slope, float* y, float* save_mean,
                                 float* save_invstd, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  RE2E_CHECK_ARG(x && gamma && beta && running_mean && running_var && y && save_mean && save_invstd && workspace, "null arg");
  RE2E_CHECK_ARG(P > 0 && C > 0, "bad shape");
  RE2E_CHECK_ARG(workspace_bytes >= re2e_bn_workspace_bytes(P, C), "workspace too small");
  int chunks = bn_chunks(P);
  long rpc = (P + chunks - 1) / chunks;
  float* part = (float*)workspace;
  float* tmp = part + (size_t)chunks * 2 * C;    // [mean | var]
  dim3 g(cdiv(C, 64), chunks);
  if (train) {
    const bool vec = bn_vec_ok(x, nullptr, C);
    if (vec) hipLaunchKernelGGL(bn_partial_vec_kernel, g, dim3(256), 0, stream, x, (const float*)nullptr, P, C, rpc, 0, (const float*)nullptr,
                                (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, part, slope);
    else hipLaunchKernelGGL(bn_partial_kernel, g, dim3(256), 0, stream, x, (const float*)nullptr, P, C, rpc, 0, (const float*)nullptr,
                       (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, part, slope);
    hipLaunchKernelGGL(bn_combine_kernel, dim3(cdiv(C, 64)), dim3(1024), 0, stream, (const float*)part, chunks, C, 1.0f / (float)P,
                       0.f, tmp, (float*)nullptr);
    if (vec) hipLaunchKernelGGL(bn_partial_vec_kernel, g, dim3(256), 0, stream, x, (const float*)nullptr, P, C, rpc, 1, (const float*)tmp,
                                (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, part, slope);
    else hipLaunchKernelGGL(bn_partial_kernel, g, dim3(256), 0, stream, x, (const float*)nullptr, P, C, rpc, 1, (const float*)tmp,
                       (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, part, slope);
    hipLaunchKernelGGL(bn_combine_kernel, dim3(cdiv(C, 64)), dim3(1024), 0, stream, (const float*)part, chunks, C, 1.0f / (float)P,
                       0.f, tmp + C, (float*)nullptr);
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(C, 256)), dim3(256), 0, stream, (const float*)tmp, (const float*)(tmp + C), P, C,
                       momentum, eps, running_mean, running_var, save_mean, save_invstd);
  } else {
    hipLaunchKernelGGL(bn_eval_stats_kernel, dim3(cdiv(C, 256)), dim3(256), 0, stream, (const float*)running_mean,
                       (const float*)running_var, C, eps, save_mean, save_invstd);
  }
  if (C % 4 == 0 && al16(x) && al16(y) && al16(save_mean) && al16(save_invstd) && al16(gamma) && al16(beta))
    hipLaunchKernelGGL(bn_apply_vec_kernel, dim3(grid_for(P * C / 4)), dim3(TPB), 0, stream, x, P, C / 4, (const float*)save_mean,
                       (const float*)save_invstd, gamma, beta, y, slope);
  else
    hipLaunchKernelGGL(bn_apply_kernel, dim3(grid_for(P * C)), dim3(TPB), 0, stream, x, P, C, (const float*)save_mean,
                       (const float*)save_invstd, gamma, beta, y, slope);
  RE2E_LAUNCH_CHECK();
  return RE2E_OK;
}
__global__ void bn_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ x, long P, int C,
                                    const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ gamma,
                                    const float* __restrict__ beta, const float* __restrict__ sdz, const float* __restrict__ sdzx,
                                    float* __restrict__ dx, float slope) {
  long tot = P * C;
  float invP = 1.0f / (float)P;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += (long)gridDim.x * blockDim.x) {
    int c = (int)(i % C);
    float xh = (x[i] - mean[c]) * invstd[c];
    float yv = xh * gamma[c] + beta[c];
    float dz = dy[i]; dz = yv > 0.f ? dz : slope * dz;
    dx[i] = gamma[c] * invstd[c] * (dz - sdz[c] * invP - xh * sdzx[c] * invP);
  }
}
__global__ void bn_param_grad_kernel(const float* sdz, const float* sdzx, int C, float* dgamma, float* dbeta, float gbeta) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  dgamma[c] = (gbeta != 0.f ? gbeta * dgamma[c] : 0.f) + sdzx[c];
  dbeta[c] = (gbeta != 0.f ? gbeta * dbeta[c] : 0.f) + sdz[c];
}
extern "C" int re2e_bn_lrelu_bwd(const float* dy, const float* x, long P, int C, const float* gamma, const float* beta,
                                 const float* save_mean, const float* save_invstd, float slope, float* dx, float* dgamma, float* dbeta,
                                 float gbeta, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  RE2E_CHECK_ARG(dy && x && gamma && beta && save_mean && save_invstd && dx && workspace, "null arg");
  RE2E_CHECK_ARG(workspace_bytes >= re2e_bn_workspace_bytes(P, C), "workspace too small");
  int chunks = bn_chunks(P);
  long rpc = (P + chunks - 1) / chunks;
  float* part = (float*)workspace;
  float* tmp = part + (size_t)chunks * 2 * C;
  dim3 g(cdiv(C, 64), chunks);
  if (bn_vec_ok(x, dy, C)) hipLaunchKernelGGL(bn_partial_vec_kernel, g, dim3(256), 0, stream, x, dy, P, C, rpc, 2, save_mean, save_invstd, gamma, beta, part, slope);
  else hipLaunchKernelGGL(bn_partial_kernel, g, dim3(256), 0, stream, x, dy, P, C, rpc, 2, save_mean, save_invstd, gamma, beta, part, slope);
  hipLaunchKernelGGL(bn_combine_kernel, dim3(cdiv(C, 64)), dim3(1024), 0, stream, (const float*)part, chunks, C, 1.0f, 1.0f, tmp, tmp + C);
  if (C % 4 == 0 && al16(dy) && al16(x) && al16(dx) && al16(save_mean) && al16(save_invstd) && al16(gamma) && al16(beta) && al16(tmp))
    hipLaunchKernelGGL(bn_bwd_apply_vec_kernel, dim3(grid_for(P * C / 4)), dim3(TPB), 0, stream, dy, x, P, C / 4, save_mean, save_invstd,
                       gamma, beta, (const float*)tmp, (const float*)(tmp + C), dx, slope);
  else
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(grid_for(P * C)), dim3(TPB), 0, stream, dy, x, P, C, save_mean, save_invstd, gamma,
                       beta, (const float*)tmp, (const float*)(tmp + C), dx, slope);
  if (dgamma && dbeta)
    hipLaunchKernelGGL(bn_param_grad_kernel, dim3(cdiv(C, 256)), dim3(256), 0, stream, (const float*)tmp, (const float*)(tmp + C), C,
                       dgamma, dbeta, gbeta);
  RE2E_LAUNCH_CHECK();
  return RE2E_OK;
}

// ---- synchronised BatchNorm (data-parallel runs that shard ONE global batch: bench.py --scaling strong): the same kernels, cut at the
// points where the caller all-reduces -- (1) sum x, (2) sum (x - mean)^2 with the GLOBAL mean, (3) sum dz / sum dz*xhat in the backward.
// out = sum over the LOCAL rows of (x - mean) (square = 0; mean may be null) or of its square (square = 1)
extern "C" int re2e_bn_sync_partial(const float* x, long P, int C, const float* mean, int square, float* out, void* workspace, size_t workspace_bytes,
                                    hipStream_t stream) {
  RE2E_CHECK_ARG(x && out && workspace && P > 0 && C > 0, "bad args");
  RE2E_CHECK_ARG(workspace_bytes >= re2e_bn_workspace_bytes(P, C), "workspace too small");
  const int chunks = bn_chunks(P);
  const long rpc = (P + chunks - 1) / chunks;
  float* part = (float*)workspace;
  dim3 g(cdiv(C, 64), chunks);
  if (bn_vec_ok(x, nullptr, C) && (!mean || al16(mean)))
    hipLaunchKernelGGL(bn_partial_vec_kernel, g, dim3(256), 0, stream, x, (const float*)nullptr, P, C, rpc, square ? 1 : 0, mean, (const float*)nullptr,
                       (const float*)nullptr, (const float*)nullptr, part, 1.f);
  else
    hipLaunchKernelGGL(bn_partial_kernel, g, dim3(256), 0, stream, x, (const float*)nullptr, P, C, rpc, square ? 1 : 0, mean, (const float*)nullptr,
                       (const float*)nullptr, (const float*)nullptr, part, 1.f);
  hipLaunchKernelGGL(bn_combine_kernel, dim3(cdiv(C, 64)), dim3(1024), 0, stream, (const float*)part, chunks, C, 1.0f, 0.f, out, (float*)nullptr);
  RE2E_LAUNCH_CHECK();
  return RE2E_OK;
}
// GLOBAL mean / biased variance over Ptotal rows -> save_mean, save_invstd, running statistics (unbiased variance with Ptotal)
extern "C" int re2e_bn_sync_finalize(const float* mean, const float* var, long Ptotal, int C, float momentum, float eps, float* running_mean,
                                     float* running_var, float* save_mean, float* save_invstd, hipStream_t stream) {
  RE2E_CHECK_ARG(mean && var && running_mean && running_var && save_mean && save_invstd && Ptotal > 0 && C > 0, "bad args");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(C, 256)), dim3(256), 0, stream, mean, var, Ptotal, C, momentum, eps, running_mean, running_var,
                     save_mean, save_invstd);
  RE2E_LAUNCH_CHECK();
  return RE2E_OK;
}
// y = lrelu((x - mean) * invstd * gamma + beta)
extern "C" int re2e_bn_apply(const float* x, long P, int C, const float* save_mean, const float* save_invstd, const float* gamma, const float* beta,
                             float slope, float* y, hipStream_t stream) {
  RE2E_CHECK_ARG(x && save_mean && save_invstd && gamma && beta && y && P > 0 && C > 0, "bad args");
  if (C % 4 == 0 && al16(x) && al16(y) && al16(save_mean) && al16(save_invstd) && al16(gamma) && al16(beta))
    hipLaunchKernelGGL(bn_apply_vec_kernel, dim3(grid_for(P * C / 4)), dim3(TPB), 0, stream, x, P, C / 4, save_mean, save_invstd, gamma, beta, y, slope);
  else
    hipLaunchKernelGGL(bn_apply_kernel, dim3(grid_for(P * C)), dim3(TPB), 0, stream, x, P, C, save_mean, save_invstd, gamma, beta, y, slope);
  RE2E_LAUNCH_CHECK();
  return RE2E_OK;
}
// out[0:C] = sum dz, out[C:2C] = sum dz * xhat over the LOCAL rows (dz = dy through the LeakyReLU)
extern "C" int re2e_bn_sync_bwd_partial(const float* dy, const float* x, long P, int C, const float* gamma, const float* beta, const float* save_mean,
                                        const float* save_invstd, float slope, float* out, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  RE2E_CHECK_ARG(dy && x && gamma && beta && save_mean && save_invstd && out && workspace && P > 0 && C > 0, "bad args");
  RE2E_CHECK_ARG(workspace_bytes >= re2e_bn_workspace_bytes(P, C), "workspace too small");
  const int chunks = bn_chunks(P);
  const long rpc = (P + chunks - 1) / chunks;
  float* part = (float*)workspace;
  dim3 g(cdiv(C, 64), chunks);
  if (bn_vec_ok(x, dy, C)) hipLaunchKernelGGL(bn_partial_vec_kernel, g, dim3(256), 0, stream, x, dy, P, C, rpc, 2, save_mean, save_invstd, gamma, beta, part, slope);
  else hipLaunchKernelGGL(bn_partial_kernel, g, dim3(256), 0, stream, x, dy, P, C, rpc, 2, save_mean, save_invstd, gamma, beta, part, slope);
  hipLaunchKernelGGL(bn_combine_kernel, dim3(cdiv(C, 64)), dim3(1024), 0, stream, (const float*)part, chunks, C, 1.0f, 1.0f, out, out + C);
  RE2E_LAUNCH_CHECK();
  return RE2E_OK;
}
// dx from sums the caller has all-reduced AND scaled by P / Ptotal (the kernels divide by the local row count P)
extern "C" int re2e_bn_sync_bwd_apply(const float* dy, const float* x, long P, int C, const float* gamma, const float* beta, const float* save_mean,
                                      const float* save_invstd, float slope, const float* sums, float* dx, hipStream_t stream) {
  RE2E_CHECK_ARG(dy && x && gamma && beta && save_mean && save_invstd && sums && dx && P > 0 && C > 0, "bad args");
  if (C % 4 == 0 && al16(dy) && al16(x) && al16(dx) && al16(save_mean) && al16(save_invstd) && al16(gamma) && al16(beta) && al16(sums))
    hipLaunchKernelGGL(bn_bwd_apply_vec_kernel, dim3(grid_for(P * C / 4)), dim3(TPB), 0, stream, dy, x, P, C / 4, save_mean, save_invstd, gamma, beta,
                       sums, sums + C, dx, slope);
  else
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(grid_for(P * C)), dim3(TPB), 0, stream, dy, x, P, C, save_mean, save_invstd, gamma, beta, sums, sums + C,
                       dx, slope);
  RE2E_LAUNCH_CHECK();
  return RE2E_OK;
}

// ---- K11 optimizer ------------------------------------------------------------------------
__global__ void clip_coef_kernel(const float* sumsq, float max_norm, float* stats) {
  float n = sqrtf(sumsq[0]);
  float c = max_norm / (n + 1e-6f);
  stats[0] = n;
  stats[1] = c < 1.f ? c : 1.f;
  stats[2] = (n == n && fabsf(n) < 3.0e38f) ? 1.f : 0.f;     // torch only tests isnan; inf norm => coef 0 anyway
  if (!(n == n)) stats[2] = 0.f;
  // second triple: "gate only" view (coefficient 1) for optimizers that share the NaN guard but
  // are not clipped (enhancer in joint_train.py:188-193)
  stats[3] = n; stats[4] = 1.f; stats[5] = stats[2];
}
extern "C" int re2e_clip_coef(const float* sumsq, float max_norm, float* stats, hipStream_t stream) {
  RE2E_CHECK_ARG(sumsq && stats, "null arg");
  hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(1), 0, stream, sumsq, max_norm, stats);
  RE2E_LAUNCH_CHECK();
  return RE2E_OK;
}
__global__ void adadelta_kernel(float* p, const float* __restrict__ g, float* sq, float* acc, long n, float rho, float eps, float lr,
                                const float* __restrict__ stats) {
  float coef = 1.f;
  if (stats) { if (stats[2] == 0.f) return; coef = stats[1]; }
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float gr = g[i] * coef;
    float v = sq[i] * rho + gr * gr * (1.f - rho);
    float u = acc[i];
    float delta = sqrtf(u + eps) / sqrtf(v + eps) * gr;
    sq[i] = v;
    acc[i] = u * rho + delta * delta * (1.f - rho);
    p[i] -= lr * delta;
  }
}
extern "C" int re2e_adadelta_step(float* p, const float* g, float* sq_avg, float* acc_delta, long n, float rho, float eps, float lr,
                                  const float* stats, hipStream_t stream) {
  RE2E_CHECK_ARG(p && g && sq_avg && acc_delta && n > 0, "bad args");
  hipLaunchKernelGGL(adadelta_kernel, dim3(grid_for(n)), dim3(TPB), 0, stream, p, g, sq_avg, acc_delta, n, rho, eps, lr, stats);
  RE2E_LAUNCH_CHECK();
  return RE2E_OK;
}
__global__ void adam_kernel(float* p, const float* __restrict__ g, float* m, float* v, long n, float lr, float b1, float b2, float eps,
                            float bc1, float bc2, const float* __restrict__ stats) {
  float coef = 1.f;
  if (stats) { if (stats[2] == 0.f) return; coef = stats[1]; }
  float step_size = lr / bc1;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float gr = g[i] * coef;
    float mi = m[i] * b1 + (1.f - b1) * gr;
    float vi = v[i] * b2 + (1.f - b2) * gr * gr;
    m[i] = mi; v[i] = vi;
    float denom = sqrtf(vi) / sqrtf(bc2) + eps;
    p[i] -= step_size * mi / denom;
  }
}
extern "C" int re2e_adam_step(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2, float eps,
                              int step, const float* stats, hipStream_t stream) {
  RE2E_CHECK_ARG(p && g && m && v && n > 0 && step >= 1, "bad args");
  float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
  hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n)), dim3(TPB), 0, stream, p, g, m, v, n, lr, beta1, beta2, eps, bc1, bc2, stats);
  RE2E_LAUNCH_CHECK();
  return RE2E_OK;
}

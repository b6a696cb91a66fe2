// Thin-channel convolutions for gfx950: direct (non-MFMA) kernels for Cin == 1 or Cout == 1.
//
// The VGG front-end's first conv (1 -> 64, model/e2e_encoder.py:171) and the discriminator's first / last
// convs (1 -> ndf and 8*ndf -> 1, model/gan_model.py:60-90) have a GEMM dimension of 1 (or KH*KW <= 16):
// an MFMA tile would be >= 97 % padding and the work is HBM-bound anyway (one pass over the wide tensor).
// These kernels stream the wide tensor once with 16-byte accesses and keep the thin operand in LDS/registers
// (the Cin = 1 FORWARD stays on the implicit GEMM: it is bound by its 64-channel output write either way):
//   * conv_cout1_kernel : out[pix]     = act(b + sum_{tap,ci} in[pix+tap][ci] * w[tap][ci])   (forward, Cout = 1;
//                                         also the data gradient of a Cin = 1 conv)
//   * wgrad_cin1_kernel : dW[tap][co]  = sum_pix in[pix+tap] * dout[pix][co]                  (Cin = 1)
//   * wgrad_cout1_kernel: dW[tap][ci]  = sum_pix dout[pix] * in[pix+tap][ci]                  (Cout = 1)
// Weight gradients are written as per-workgroup partial slabs [slab][tap*C+ci][co] and summed by the
// implicit-GEMM engine's deterministic split-K reduce (fixed order => run-to-run bitwise stable).
#include <stdlib.h>

#include "common.h"

namespace {

__device__ __forceinline__ void decode_pixel(const ConvGeom& g, int m, int& n, int& iy0, int& ix0) {
  int px = m % g.PW; int t = m / g.PW; int py = t % g.PH; n = t / g.PH;
  iy0 = py * g.SY + g.OY0; ix0 = px * g.SX + g.OX0;
}

// ---------------------------------------------------------------------------------------------
// Cout == 1 forward.  L lanes share one pixel (each owns C/4/L float4 channel groups per tap), 256/L pixels
// per pass; the K = KH*KW*C weights sit in LDS.  A workgroup walks a CONTIGUOUS range of passes so that the
// rows its taps re-read stay in its XCD's L2.  KH_T/KW_T/CH_T > 0 fix the tap and channel-chunk counts at
// compile time (all loads of a pixel issued back to back); 0 = run-time loops.
// ---------------------------------------------------------------------------------------------
template <int L, int KH_T, int KW_T, int CH_T>
__global__ __launch_bounds__(256) void conv_cout1_kernel(ConvGeom g, const float* __restrict__ wg, int K, int M, OutMap o,
                                                         const float* __restrict__ bias, int act, float beta) {
  extern __shared__ __attribute__((aligned(16))) float wsm[];
  for (int i = threadIdx.x; i < K / 4; i += 256) reinterpret_cast<f32x4*>(wsm)[i] = reinterpret_cast<const f32x4*>(wg)[i];
  __syncthreads();
  constexpr int PPB = 256 / L;
  const int sub = threadIdx.x % L, pl = threadIdx.x / L;
  const int c4n = g.C >> 2;
  const unsigned nbytes = (unsigned)((long)g.NI * g.H * g.W * g.C * 4);
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.in), 0, nbytes, 0x00020000);
  const float b = bias ? bias[0] : 0.f;
  const int npass = (M + PPB - 1) / PPB;
  const int per = (npass + gridDim.x - 1) / gridDim.x;
  const int pass_end = min(npass, ((int)blockIdx.x + 1) * per);
  for (int pass = blockIdx.x * per; pass < pass_end; ++pass) {
    const int m = pass * PPB + pl;
    const bool live = m < M;
    int n, iy0, ix0;
    decode_pixel(g, live ? m : 0, n, iy0, ix0);
    float acc = 0.f;
    if constexpr (KH_T > 0) {
      // compile-time tap / channel-chunk counts: every load of the pixel is issued before the first FMA
      constexpr int NT = KH_T * KW_T * CH_T;
      f32x4 v[NT];
#pragma unroll
      for (int kh = 0; kh < KH_T; ++kh) {
        const int iy = iy0 + kh * g.DY;
        const bool oky = live & ((unsigned)iy < (unsigned)g.H);
#pragma unroll
        for (int kw = 0; kw < KW_T; ++kw) {
          const int ix = ix0 + kw * g.DX;
          const bool ok = oky & ((unsigned)ix < (unsigned)g.W);
          const unsigned base = (unsigned)(((((long)n * g.H + iy) * g.W + ix) * g.C) * 4);
#pragma unroll
          for (int ch = 0; ch < CH_T; ++ch)
            v[(kh * KW_T + kw) * CH_T + ch] =
                __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, ok ? base + (ch * L + sub) * 16u : nbytes, 0, 0));
        }
      }
#pragma unroll
      for (int t = 0; t < KH_T * KW_T; ++t) {
#pragma unroll
        for (int ch = 0; ch < CH_T; ++ch) {
          const f32x4 w = reinterpret_cast<const f32x4*>(wsm + t * g.C)[ch * L + sub];
          const f32x4 x = v[t * CH_T + ch];
          acc = fmaf(x[0], w[0], acc); acc = fmaf(x[1], w[1], acc); acc = fmaf(x[2], w[2], acc); acc = fmaf(x[3], w[3], acc);
        }
      }
    } else {
      for (int kh = 0; kh < g.KH; ++kh) {
        const int iy = iy0 + kh * g.DY;
        const bool oky = live & ((unsigned)iy < (unsigned)g.H);
        for (int kw = 0; kw < g.KW; ++kw) {
          const int ix = ix0 + kw * g.DX;
          const bool ok = oky & ((unsigned)ix < (unsigned)g.W);
          const unsigned base = (unsigned)(((((long)n * g.H + iy) * g.W + ix) * g.C) * 4);
          const f32x4* wv = reinterpret_cast<const f32x4*>(wsm + (kh * g.KW + kw) * g.C);
          for (int c = sub; c < c4n; c += L) {
            f32x4 x = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, ok ? base + c * 16u : nbytes, 0, 0));
            f32x4 w = wv[c];
            acc = fmaf(x[0], w[0], acc); acc = fmaf(x[1], w[1], acc); acc = fmaf(x[2], w[2], acc); acc = fmaf(x[3], w[3], acc);
          }
        }
      }
    }
#pragma unroll
    for (int s = L / 2; s > 0; s >>= 1) acc += __shfl_xor(acc, s, 64);
    if (live && sub == 0) {
      const long off = o.off(m);
      float v = apply_act(acc + b, act);
      if (beta != 0.f) v += o.out[off];
      o.out[off] = v;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Cout == 1, 64 channels, 3x3 / stride 1 / pad 1 (the data gradient of VGG conv1_1: 1 GB of dz at config 4 -> one value per pixel).
// conv_cout1_kernel reads every pixel's 64 channels once per tap (9x, served by L2: ~1 TB/s of useful input).  Here the product is
// split the other way round: a workgroup owns TH output rows of one image, FIRST projects every input pixel of its TH + 2 rows onto
// the 9 tap vectors into a [row][column][9] table in LDS, THEN every output pixel adds its 9 table entries.  HBM traffic =
// (TH + 2) / TH of the input.  The projection reads the rows as the contiguous byte range they are and runs on the matrix cores (below).
// ---------------------------------------------------------------------------------------------

template <int TH>
__global__ __launch_bounds__(256) void conv_cout1_rows3x3_kernel(ConvGeom g, const float* __restrict__ wg, OutMap o, const float* __restrict__ bias,
                                                                 int act, float beta) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* tab = sm;                                        // [(TH + 2)][W + 2][9], column 0 and W + 1 = outside the image (zero)
  const int W2 = g.W + 2, tid = threadIdx.x;
  const int tiles_y = (g.H + TH - 1) / TH;
  const int n = blockIdx.x / tiles_y, y0 = (blockIdx.x - n * tiles_y) * TH;
  for (int i = tid; i < (TH + 2) * 18; i += 256) {        // the two border columns; everything else is written by the projection
    const int r = i / 18, j = i - r * 18;
    tab[(r * W2 + (j < 9 ? 0 : g.W + 1)) * 9 + (j < 9 ? j : j - 9)] = 0.f;
  }
  // projection on the matrix cores: v_mfma_f32_16x16x4_f32 with M = 16 consecutive pixels of the workgroup's contiguous pixel range, N = the 9
  // taps (padded to 16), K = the 64 channels in 16 k-steps.  Lane (p, kq) loads pixel p's channels 16j + 4kq .. + 3 as one 16-byte load per
  // j (A operand of k-steps (j, 0..3)); the weights of tap (lane & 15) at the same channels sit in 16 registers (B operand); the accumulator
  // holds tap (lane & 15) of pixels 4kq .. 4kq + 3: stored straight into the table, no cross-lane reduction (the vector form -- 36 FMAs +
  // 36 DPP adds + 9 selects per 4 pixels -- kept the SIMDs busier than the 2.2 TB/s it streamed: 240 us in the step, 0.28 of the HBM peak).
  typedef float f32x4v __attribute__((ext_vector_type(4)));
  const int lane = tid & 63, wave = tid >> 6, pp = lane & 15, kq = lane >> 4;
  f32x4v wr[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    wr[j] = pp < 9 ? *reinterpret_cast<const f32x4v*>(wg + pp * 64 + 16 * j + 4 * kq) : f32x4v{0.f, 0.f, 0.f, 0.f};
  }
  const unsigned img_bytes = (unsigned)g.H * g.W * 256u;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.in) + (long)n * g.H * g.W * 64, 0, img_bytes, 0x00020000);
  const int npx = (TH + 2) * g.W, ngrp = (npx + 15) >> 4;
  const int lin0 = (y0 - 1) * g.W;                         // flattened pixel index of the range's first pixel (negative above the image)
  const int hw = g.H * g.W;
  int q0 = wave * 16 + 4 * kq;                             // first of the 4 pixels this lane's accumulator holds
  int rr = q0 / g.W, xx = q0 - rr * g.W;
  constexpr int U = 3;                                     // pixel groups in flight per wavefront (12 x 16-byte loads per lane)
  for (int grp = wave; grp < ngrp; grp += 4 * U) {
    f32x4v v[U][4];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int q = (grp + 4 * u) * 16 + pp, lin = lin0 + q;
      const bool ok = q < npx && (unsigned)lin < (unsigned)hw;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        v[u][j] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(rs, ok ? (unsigned)(lin * 256 + j * 64 + kq * 16) : 0x80000000u, 0, 0));
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      f32x4v acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(v[u][j][i], wr[j][i], acc, 0, 0, 0);
      if (pp < 9) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          int x2 = xx + i, r2 = rr;
          if (x2 >= g.W) { x2 -= g.W; ++r2; }
          if (q0 + i < npx) tab[(r2 * W2 + x2 + 1) * 9 + pp] = acc[i];
        }
      }
      q0 += 64; xx += 64;
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (xx >= g.W) { xx -= g.W; ++rr; }
    }
  }
  __syncthreads();
  const float b = bias ? bias[0] : 0.f;
  for (int q = tid; q < TH * g.W; q += 256) {
    const int ry = q / g.W, x = q - ry * g.W, py = y0 + ry;
    if (py >= g.H) break;
    float acc = b;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int r = ry + 1 + kh * g.DY + g.OY0;           // table row of input row py + kh*DY + OY0
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) acc += tab[(r * W2 + x + 1 + kw * g.DX + g.OX0) * 9 + kh * 3 + kw];
    }
    const long off = ((long)n * g.H + py) * g.W + x;       // dense single-channel output
    float val = apply_act(acc, act);
    if (beta != 0.f) val += o.out[off];
    o.out[off] = val;
  }
}

// ---------------------------------------------------------------------------------------------
// Cin == 1 weight gradient.  A workgroup owns a contiguous range of output rows (n, py).  Per row the KH
// input rows it touches are staged in LDS (zero-filled outside the image), then c4n = Cout/4 lanes cover
// the output channels of one pixel (256/c4n pixels per pass, two passes in flight) and every thread
// accumulates acc[tap][4].  Slab layout [tap][Cout].
// ---------------------------------------------------------------------------------------------
constexpr int MAXTAPS = 16;

__global__ __launch_bounds__(256) void wgrad_cin1_kernel(ConvGeom g, const float* __restrict__ dout, int Cout, int span,
                                                         float* __restrict__ slabs) {
  extern __shared__ __attribute__((aligned(16))) float dyn[];   // [KH][span] input rows, then the reduce buffer
  __shared__ int s_off[MAXTAPS];
  const int taps = g.KH * g.KW;
  if (threadIdx.x < MAXTAPS) {
    int t = threadIdx.x < taps ? threadIdx.x : 0;
    s_off[threadIdx.x] = (t / g.KW) * span + (t % g.KW) * g.DX;
  }
  float* rows = dyn;
  float* red = dyn + ((g.KH * span + 3) & ~3);                   // [4 waves][taps][Cout]
  const int c4n = Cout >> 2;           // power of two, <= 32
  const int ppb = 256 / c4n;
  const int c4 = threadIdx.x % c4n, pl = threadIdx.x / c4n;
  f32x4 acc[MAXTAPS];
#pragma unroll
  for (int t = 0; t < MAXTAPS; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nrows = g.NI * g.PH;
  const int per = (nrows + gridDim.x - 1) / gridDim.x;
  const int r_end = min(nrows, ((int)blockIdx.x + 1) * per);
  for (int r = blockIdx.x * per; r < r_end; ++r) {
    const int n = r / g.PH, py = r - n * g.PH;
    __syncthreads();                                             // previous row's readers are done
    for (int i = threadIdx.x; i < g.KH * span; i += 256) {
      const int kh = i / span, xs = i - kh * span;
      const int iy = py * g.SY + kh * g.DY + g.OY0, ix = xs + g.OX0;
      const bool ok = ((unsigned)iy < (unsigned)g.H) & ((unsigned)ix < (unsigned)g.W);
      rows[i] = ok ? g.in[((long)n * g.H + iy) * g.W + ix] : 0.f;
    }
    __syncthreads();
    const float* drow = dout + (long)r * g.PW * Cout + c4 * 4;
    for (int px = pl; px < g.PW; px += 2 * ppb) {
      const int px2 = px + ppb;
      const bool two = px2 < g.PW;
      const f32x4 d0 = *reinterpret_cast<const f32x4*>(drow + (long)px * Cout);
      const f32x4 d1 = two ? *reinterpret_cast<const f32x4*>(drow + (long)px2 * Cout) : f32x4{0.f, 0.f, 0.f, 0.f};
      const float* s0 = rows + px * g.SX;
      const float* s1 = rows + (two ? px2 : px) * g.SX;
#pragma unroll
      for (int t = 0; t < MAXTAPS; ++t) {
        if (t < taps) {
          const float v0 = s0[s_off[t]], v1 = s1[s_off[t]];
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[t][j] = fmaf(v1, d1[j], fmaf(v0, d0[j], acc[t][j]));
        }
      }
    }
  }
  // pixel lanes of one wave (lane bits above c4n), then the 4 waves through LDS
  const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
#pragma unroll
  for (int t = 0; t < MAXTAPS; ++t) {
    if (t < taps) {
      for (int s = c4n; s < 64; s <<= 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[t][j] += __shfl_xor(acc[t][j], s, 64);
      }
      if (lane < c4n) *reinterpret_cast<f32x4*>(red + ((wid * taps + t) * Cout + lane * 4)) = acc[t];
    }
  }
  __syncthreads();
  float* slab = slabs + (long)blockIdx.x * taps * Cout;
  for (int i = threadIdx.x; i < taps * Cout; i += 256) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) s += red[w * taps * Cout + i];
    slab[i] = s;
  }
}

// ---------------------------------------------------------------------------------------------
// Cout == 1 weight gradient.  Workgroup = (pixel chunk, tap); threads cover the C/4 channel groups
// (c4n lanes) x 256/c4n pixel lanes.  Slab layout [tap*C + ci] (Cout = 1).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void wgrad_cout1_kernel(ConvGeom g, const float* __restrict__ dout, int P, int chunk,
                                                          float* __restrict__ slabs) {
  __shared__ __attribute__((aligned(16))) float red[256 * 4];
  const int tap = blockIdx.y, kh = tap / g.KW, kw = tap - kh * g.KW;
  const int c4n = g.C >> 2;                      // power of two
  const int lanes = c4n < 256 ? c4n : 256;       // channel-group lanes per pixel
  const int ppb = 256 / lanes;
  const int cl = threadIdx.x % lanes, pl = threadIdx.x / lanes;
  const int p0 = blockIdx.x * chunk, p1 = min(P, p0 + chunk);
  const unsigned nbytes = (unsigned)((long)g.NI * g.H * g.W * g.C * 4);
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.in), 0, nbytes, 0x00020000);
  float* slab = slabs + ((long)blockIdx.x * g.KH * g.KW + tap) * g.C;
  for (int cbase = 0; cbase < c4n; cbase += lanes) {
    const int c4 = cbase + cl;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int p = p0 + pl; p < p1; p += ppb) {
      const float d = dout[p];
      int n, iy0, ix0;
      decode_pixel(g, p, n, iy0, ix0);
      const int iy = iy0 + kh * g.DY, ix = ix0 + kw * g.DX;
      const bool ok = ((unsigned)iy < (unsigned)g.H) & ((unsigned)ix < (unsigned)g.W);
      const unsigned off = ok ? (unsigned)((((((long)n * g.H + iy) * g.W + ix) * g.C) + c4 * 4) * 4) : nbytes;
      const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
      acc[0] = fmaf(d, v[0], acc[0]); acc[1] = fmaf(d, v[1], acc[1]); acc[2] = fmaf(d, v[2], acc[2]); acc[3] = fmaf(d, v[3], acc[3]);
    }
    __syncthreads();
    *reinterpret_cast<f32x4*>(red + threadIdx.x * 4) = acc;
    __syncthreads();
    if (pl == 0) {
      f32x4 s = acc;
      for (int q = 1; q < ppb; ++q) s += *reinterpret_cast<const f32x4*>(red + (q * lanes + cl) * 4);
      *reinterpret_cast<f32x4*>(slab + c4 * 4) = s;
    }
  }
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
inline bool pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

constexpr int COUT1_CHUNK = 128;   // pixels per (chunk, tap) workgroup of the Cout == 1 weight gradient

inline int cin1_span(const ConvGeom& g) { return (g.PW - 1) * g.SX + (g.KW - 1) * g.DX + 1; }

template <int L, int KH_T, int KW_T, int CH_T>
void launch_cout1(const ConvGeom& g, const float* wg, int K, int M, const OutMap& o, const float* bias, int act, float beta,
                  int grid, hipStream_t st) {
  const size_t lds = (size_t)K * 4;
  auto kern = conv_cout1_kernel<L, KH_T, KW_T, CH_T>;
  if (lds > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, g, wg, K, M, o, bias, act, beta);
}

// ---- Cin == 1 forward (VGG conv1_1, discriminator conv1): out[m][co] = act(b[co] + sum_taps in[pixel + tap] W[co][tap]) ----
// K is 9 or 16, so an MFMA tile is mostly padding and the kernel is bound by writing the output (1 GB at config 4).
// Thread = (4 output channels, pixel lane): its 4 x taps weights live in registers, pixels advance by the number of pixel lanes
// with an incremental (n, py, px) update -- no division in the loop, one float4 store per pixel.
template <int KH, int KW>
__global__ __launch_bounds__(256) void conv_cin1_fwd_kernel(ConvGeom g, const float* __restrict__ wg, int Cout, long M, float* __restrict__ out,
                                                            const float* __restrict__ bias, int act) {
  constexpr int TAPS = KH * KW;
  const int c4n = Cout >> 2, c4 = threadIdx.x % c4n, pl = threadIdx.x / c4n, npl = 256 / c4n;
  float w[TAPS][4];
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int j = 0; j < 4; ++j) w[t][j] = wg[(4 * c4 + j) * TAPS + t];
  f32x4 bv = {0.f, 0.f, 0.f, 0.f};
  if (bias) bv = f32x4{bias[4 * c4], bias[4 * c4 + 1], bias[4 * c4 + 2], bias[4 * c4 + 3]};
  const long stride = (long)gridDim.x * npl;
  long m = (long)blockIdx.x * npl + pl;
  if (m >= M) return;
  int px = (int)(m % g.PW); long r = m / g.PW; int py = (int)(r % g.PH); int n = (int)(r / g.PH);
  const int dpx = (int)(stride % g.PW); const long rr = stride / g.PW; const int dpy = (int)(rr % g.PH), dn = (int)(rr / g.PH);
  for (; m < M; m += stride) {
    f32x4 acc = bv;
    const float* ib = g.in + (long)n * g.H * g.W;
#pragma unroll
    for (int kh = 0; kh < KH; ++kh) {
      const int iy = py * g.SY + kh * g.DY + g.OY0;
#pragma unroll
      for (int kw = 0; kw < KW; ++kw) {
        const int ix = px * g.SX + kw * g.DX + g.OX0;
        const float v = (iy >= 0 && iy < g.H && ix >= 0 && ix < g.W) ? ib[(long)iy * g.W + ix] : 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] += v * w[kh * KW + kw][j];
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = apply_act(acc[j], act);
    *reinterpret_cast<f32x4*>(out + m * Cout + 4 * c4) = acc;
    px += dpx; py += dpy; n += dn;                         // (n, py, px) += stride, with carries
    if (px >= g.PW) { px -= g.PW; ++py; }
    if (py >= g.PH) { py -= g.PH; ++n; }
  }
}

}  // namespace

bool thin_conv_forward(const ConvGeom& g, const float* wg, int Cout, const OutMap& o, const float* bias, int act, float beta,
                       hipStream_t st) {
  const int M = g.NI * g.PH * g.PW;
  const int K = g.KH * g.KW * g.C;
  if (g.C == 1 && Cout % 4 == 0 && Cout <= 256 && pow2(Cout / 4) && !o.remap && o.ldc == Cout && beta == 0.f && aligned16(o.out) &&
      ((g.KH == 3 && g.KW == 3) || (g.KH == 4 && g.KW == 4))) {
    const int npl = 256 / (Cout / 4);
    const int grid = (int)min((long)cdiv(M, npl * 8), 8192L);     // >= 8 pixels per thread
    if (g.KH == 3) hipLaunchKernelGGL((conv_cin1_fwd_kernel<3, 3>), dim3(grid), dim3(256), 0, st, g, wg, Cout, (long)M, o.out, bias, act);
    else hipLaunchKernelGGL((conv_cin1_fwd_kernel<4, 4>), dim3(grid), dim3(256), 0, st, g, wg, Cout, (long)M, o.out, bias, act);
    return true;
  }
  if (!(Cout == 1 && g.C % 4 == 0 && K <= 16384 && aligned16(g.in) && aligned16(wg))) return false;
  // 64 channels, 3x3, stride 1, taps within one pixel of the output position, dense single-channel output: row-tile kernel
  static const bool no_rows = exp_env("RE2E_NO_COUT1_ROWS") != nullptr;
  if (!no_rows && g.C == 64 && g.KH == 3 && g.KW == 3 && g.SY == 1 && g.SX == 1 && g.PH == g.H && g.PW == g.W && !o.remap && o.ldc == 1 &&
      (g.DY == 1 || g.DY == -1) && (g.DX == 1 || g.DX == -1) && g.OY0 == -g.DY && g.OX0 == -g.DX && g.W >= 16 && g.W <= 256 &&
      (long)g.H * g.W * 256 < 0x7fffffffL) {
    constexpr int TH = 16;
    const size_t lds = (size_t)(TH + 2) * (g.W + 2) * 9 * sizeof(float);
    auto kern = conv_cout1_rows3x3_kernel<TH>;
    static LdsLimit lim;
    lim.ensure(reinterpret_cast<const void*>(kern), lds);
    hipLaunchKernelGGL(kern, dim3((unsigned)(g.NI * cdiv(g.H, TH))), dim3(256), lds, st, g, wg, o, bias, act, beta);
    return true;
  }
  const int c4n = g.C / 4;
  int L = 1;
  while (L < 64 && c4n % (L * 2) == 0) L *= 2;
  const int npass = cdiv(M, 256 / L);
  const int grid = (int)min((long)cdiv(npass, 8), 4096L);      // >= 8 consecutive passes per workgroup
  const int taps = g.KH * 16 + g.KW;
#define RE2E_ARGS g, wg, K, M, o, bias, act, beta, grid, st
  // the shapes of the training step get compile-time tap / chunk counts; anything else the run-time loops
  if (L == 16 && c4n == 16 && taps == 3 * 16 + 3) launch_cout1<16, 3, 3, 1>(RE2E_ARGS);        // VGG conv1_1 data gradient
  else if (L == 16 && c4n == 16 && taps == 2 * 16 + 2) launch_cout1<16, 2, 2, 1>(RE2E_ARGS);   // D conv1 data gradient (parity class)
  else if (L == 64 && c4n == 128 && taps == 4 * 16 + 4) launch_cout1<64, 4, 4, 2>(RE2E_ARGS);  // D conv5 forward
  else switch (L) {
    case 1: launch_cout1<1, 0, 0, 0>(RE2E_ARGS); break;
    case 2: launch_cout1<2, 0, 0, 0>(RE2E_ARGS); break;
    case 4: launch_cout1<4, 0, 0, 0>(RE2E_ARGS); break;
    case 8: launch_cout1<8, 0, 0, 0>(RE2E_ARGS); break;
    case 16: launch_cout1<16, 0, 0, 0>(RE2E_ARGS); break;
    case 32: launch_cout1<32, 0, 0, 0>(RE2E_ARGS); break;
    default: launch_cout1<64, 0, 0, 0>(RE2E_ARGS); break;
  }
#undef RE2E_ARGS
  return true;
}

// rows = NI*PH output rows (Cin == 1 path), P = pixels
int thin_wgrad_slabs(int C, int Cout, int KH, int KW, long P, long rows) {
  if (C == 1 && Cout % 4 == 0 && pow2(Cout / 4) && Cout <= 128 && KH * KW <= MAXTAPS) {
    long nb = (rows + 7) / 8;                                   // >= 8 rows per workgroup
    return (int)(nb < 1 ? 1 : (nb > 1024 ? 1024 : nb));
  }
  if (Cout == 1 && C % 4 == 0 && pow2(C / 4)) return cdiv(P, COUT1_CHUNK);
  return 0;
}

void thin_wgrad(const ConvGeom& g, const float* dout, int Cout, float* slabs, int nslab, hipStream_t st) {
  const int P = g.NI * g.PH * g.PW;
  if (g.C == 1) {
    const int span = cin1_span(g);
    const size_t lds = (size_t)(((g.KH * span + 3) & ~3) + 4 * g.KH * g.KW * Cout) * 4;
    hipLaunchKernelGGL(wgrad_cin1_kernel, dim3(nslab), dim3(256), lds, st, g, dout, Cout, span, slabs);
  } else {
    hipLaunchKernelGGL(wgrad_cout1_kernel, dim3(nslab, g.KH * g.KW), dim3(256), 0, st, g, dout, P, COUT1_CHUNK, slabs);
  }
}

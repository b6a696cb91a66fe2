// 4x4 / stride-1 convolution as Winograd F(2x2, 4x4) (the discriminator's conv4: networks.py NLayerDiscriminator, 256 -> 512
// channels on a 100 x 10 map, and its data gradient -- 5.9 ms of direct products per config-4 step at 100-114 TFLOP/s).
//
// A 2x2 output tile needs 5x5 = 25 element-wise products per (input channel, output channel) instead of 2*2*16 = 64 multiply-adds:
// 2.56x fewer matrix-core FLOPs.  Unlike the 3x3 layers (winograd.hip: one fused kernel) the map is small and the channel counts
// large, so the three stages are separate launches around the GEMM engine:
//
//   w44_filter_kernel  U[o][pos][c]    = (G k G^T)[pos],  k = w[o][c] (forward) or w[c][o] rotated by 180 degrees (data gradient)
//   w44_input_kernel   V[tile][pos][c] = (B^T d B)[pos],  d = the 5x5 input patch of the tile (zero outside the image)
//   gemm_kslices       M[pos][tile][o] = sum_c V[tile][pos][c] U[o][pos][c]: ONE launch of the engine's x W^T kernel over the
//                      concatenated K axis (25 * C), its split-K mechanism asked for exactly 25 slices -- slice `pos` is the product of
//                      position `pos`, and the partial slabs ARE the batched result (no reduce pass)
//   w44_output_kernel  y[2x2 tile][o]  = A^T M A
//
// Interpolation points (0, 1, -1, 2, inf); matrices and their fp32 error (1.3e-6 relative at C = 256, 8x a direct fp32 sum) from
// tools/wino_f24_matrices.py.  V and M cost 615 MB of HBM round trip per call (~0.2 ms), the GEMM 52 instead of 120 GFLOP.
#include <hip/hip_runtime.h>
#include "../../include/re2e.h"
#include "common.h"

namespace {

// B^T rows applied to a 5-vector (d0..d4)
template <class T>
__device__ __forceinline__ void bt5(const T& d0, const T& d1, const T& d2, const T& d3, const T& d4, T (&r)[5]) {
  r[0] = 2.f * d0 - d1 - 2.f * d2 + d3;
  r[1] = -2.f * d1 - d2 + d3;
  r[2] = 2.f * d1 - 3.f * d2 + d3;
  r[3] = d3 - d1;
  r[4] = 2.f * d1 - d2 - 2.f * d3 + d4;
}
// G rows applied to a 4-vector
__device__ __forceinline__ void g4(float k0, float k1, float k2, float k3, float (&r)[5]) {
  r[0] = 0.5f * k0;
  r[1] = -0.5f * (k0 + k1 + k2 + k3);
  r[2] = (-k0 + k1 - k2 + k3) * (1.f / 6.f);
  r[3] = (k0 + 2.f * k1 + 4.f * k2 + 8.f * k3) * (1.f / 6.f);
  r[4] = k3;
}

// U[o'][pos][c'] from w[Cout][Cin][4][4].  forward: o' = o, c' = c, k = w[o][c]; data gradient: o' = c, c' = o, k[a][b] = w[o][c][3-a][3-b].
__global__ __launch_bounds__(256) void w44_filter_kernel(const float* __restrict__ w, int Cout, int Cin, int dgrad, float* __restrict__ U) {
  const int Cp = dgrad ? Cout : Cin, Op = dgrad ? Cin : Cout;          // contraction / output channels of this product
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)Op * Cp) return;
  const int cp = (int)(i % Cp), op = (int)(i / Cp);
  const float* kp = w + ((long)(dgrad ? cp : op) * Cin + (dgrad ? op : cp)) * 16;
  float k[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) k[a][b] = dgrad ? kp[(3 - a) * 4 + (3 - b)] : kp[a * 4 + b];
  float t[5][4];                                                       // G k
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    float r[5];
    g4(k[0][b], k[1][b], k[2][b], k[3][b], r);
#pragma unroll
    for (int x = 0; x < 5; ++x) t[x][b] = r[x];
  }
  float* up = U + (long)op * 25 * Cp + cp;
#pragma unroll
  for (int x = 0; x < 5; ++x) {
    float r[5];
    g4(t[x][0], t[x][1], t[x][2], t[x][3], r);                         // (G k) G^T
#pragma unroll
    for (int y = 0; y < 5; ++y) up[(long)(x * 5 + y) * Cp] = r[y];
  }
}

// V[tile][pos][c]: one thread per (tile, 4 channels); a wavefront covers 256 consecutive channels of one tile (1 KiB loads / stores)
// Element (tile p, position, channel c) goes to V[p * tile_stride + position * pos_stride + c]: tile-major [tile][pos][c] for the forward /
// data-gradient products (K = channels), position-major [pos][tile][c] for the weight gradient (K = tiles; tiles P .. Ppad-1 are zero rows).
__global__ __launch_bounds__(256) void w44_input_kernel(const float* __restrict__ in, int NI, int H, int W, int C, int pad, int ty, int tx,
                                                        float* __restrict__ V, long tile_stride, long pos_stride, long Ppad) {
  const int c4n = C >> 2;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const long P = (long)NI * ty * tx;
  if (i >= Ppad * c4n) return;
  const int c4 = (int)(i % c4n);
  const long p = i / c4n;
  const int x = (int)(p % tx), y = (int)((p / tx) % ty), n = p < P ? (int)(p / ((long)tx * ty)) : -1;
  const int iy0 = 2 * y - pad, ix0 = 2 * x - pad;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  f32x4 t[5][5];                                                       // B^T d, column by column
#pragma unroll
  for (int b = 0; b < 5; ++b) {
    f32x4 d[5];
    const int ix = ix0 + b;
#pragma unroll
    for (int a = 0; a < 5; ++a) {
      const int iy = iy0 + a;
      const bool ok = n >= 0 && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
      d[a] = ok ? *reinterpret_cast<const f32x4*>(in + (((long)n * H + iy) * W + ix) * C + c4 * 4) : zero;
    }
    f32x4 r[5];
    bt5(d[0], d[1], d[2], d[3], d[4], r);
#pragma unroll
    for (int a = 0; a < 5; ++a) t[a][b] = r[a];
  }
  float* vp = V + p * tile_stride + c4 * 4;
#pragma unroll
  for (int a = 0; a < 5; ++a) {
    f32x4 r[5];
    bt5(t[a][0], t[a][1], t[a][2], t[a][3], t[a][4], r);               // (B^T d) B
#pragma unroll
    for (int b = 0; b < 5; ++b) *reinterpret_cast<f32x4*>(vp + (long)(a * 5 + b) * pos_stride) = r[b];
  }
}

// y = A^T M A per tile; A^T = [1 1 1 1 0; 0 1 -1 2 1]
__global__ __launch_bounds__(256) void w44_output_kernel(const float* __restrict__ M, long P, int Cout, int OH, int OW, int ty, int tx,
                                                         float* __restrict__ out) {
  const int o4n = Cout >> 2;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= P * o4n) return;
  const int o4 = (int)(i % o4n);
  const long p = i / o4n;
  const int x = (int)(p % tx), y = (int)((p / tx) % ty), n = (int)(p / ((long)tx * ty));
  const float* mp = M + p * Cout + o4 * 4;
  const long ps = P * Cout;
  f32x4 s[2][5];                                                       // A^T M
#pragma unroll
  for (int b = 0; b < 5; ++b) {
    f32x4 m[5];
#pragma unroll
    for (int a = 0; a < 5; ++a) m[a] = *reinterpret_cast<const f32x4*>(mp + (long)(a * 5 + b) * ps);
    s[0][b] = m[0] + m[1] + m[2] + m[3];
    s[1][b] = m[1] - m[2] + 2.f * m[3] + m[4];
  }
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    const int oy = 2 * y + a;
    if (oy >= OH) continue;
    const f32x4 y0 = s[a][0] + s[a][1] + s[a][2] + s[a][3];
    const f32x4 y1 = s[a][1] - s[a][2] + 2.f * s[a][3] + s[a][4];
    float* op = out + (((long)n * OH + oy) * OW + 2 * x) * Cout + o4 * 4;
    *reinterpret_cast<f32x4*>(op) = y0;
    if (2 * x + 1 < OW) *reinterpret_cast<f32x4*>(op + Cout) = y1;
  }
}

// weight gradient, output side: dM[pos][tile][o] = (A dY A^T)[pos] of the tile's 2x2 output gradients (zero outside the image and for the
// zero-row tiles P .. Ppad-1); A rows: (1,0) (1,1) (1,-1) (1,2) (0,1)
template <class T>
__device__ __forceinline__ void a5(const T& y0, const T& y1, T (&r)[5]) {
  r[0] = y0; r[1] = y0 + y1; r[2] = y0 - y1; r[3] = y0 + 2.f * y1; r[4] = y1;
}
__global__ __launch_bounds__(256) void w44_dout_kernel(const float* __restrict__ dy, int NI, int OH, int OW, int Cout, int ty, int tx, long P, long Ppad,
                                                       float* __restrict__ dM) {
  const int o4n = Cout >> 2;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= Ppad * o4n) return;
  const int o4 = (int)(i % o4n);
  const long p = i / o4n;
  const int x = (int)(p % tx), y = (int)((p / tx) % ty), n = p < P ? (int)(p / ((long)tx * ty)) : -1;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  f32x4 g[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int oy = 2 * y + a, ox = 2 * x + b;
      g[a][b] = (n >= 0 && oy < OH && ox < OW) ? *reinterpret_cast<const f32x4*>(dy + (((long)n * OH + oy) * OW + ox) * Cout + o4 * 4) : zero;
    }
  f32x4 t0[5], t1[5];                                                 // A dY, column by column
  a5(g[0][0], g[1][0], t0);
  a5(g[0][1], g[1][1], t1);
  float* mp = dM + p * Cout + o4 * 4;
  const long ps = Ppad * Cout;
#pragma unroll
  for (int a = 0; a < 5; ++a) {
    f32x4 r[5];
    a5(t0[a], t1[a], r);                                               // (A dY) A^T
#pragma unroll
    for (int b = 0; b < 5; ++b) *reinterpret_cast<f32x4*>(mp + (long)(a * 5 + b) * ps) = r[b];
  }
}

// dW[o][c][a][b] (+)= sum_xy G[x][a] G[y][b] dU[x*5+y][c][o], dU[pos] = the sum of the position's `sub` K-slices of the engine launch
__global__ __launch_bounds__(256) void w44_wgrad_final_kernel(const float* __restrict__ slabs, int C, int Cout, int sub, float* __restrict__ gw, float beta) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)C * Cout) return;
  const int o = (int)(i % Cout), c = (int)(i / Cout);
  const long ss = (long)C * Cout;
  float u[5][5];
#pragma unroll
  for (int pos = 0; pos < 25; ++pos) {
    float v = 0.f;
    for (int z = 0; z < sub; ++z) v += slabs[((long)pos * sub + z) * ss + i];
    u[pos / 5][pos % 5] = v;
  }
  // G^T rows: column a of G = (1/2, -1/2, -1/6, 1/6, 0), (0, -1/2, 1/6, 1/3, 0), (0, -1/2, -1/6, 2/3, 0), (0, -1/2, 1/6, 4/3, 1)
  auto gt = [](float v0, float v1, float v2, float v3, float v4, float (&r)[4]) {
    const float s6 = 1.f / 6.f;
    r[0] = 0.5f * v0 - 0.5f * v1 - s6 * v2 + s6 * v3;
    r[1] = -0.5f * v1 + s6 * v2 + 2.f * s6 * v3;
    r[2] = -0.5f * v1 - s6 * v2 + 4.f * s6 * v3;
    r[3] = -0.5f * v1 + s6 * v2 + 8.f * s6 * v3 + v4;
  };
  float t[4][5];                                                       // G^T dU
#pragma unroll
  for (int y = 0; y < 5; ++y) {
    float r[4];
    gt(u[0][y], u[1][y], u[2][y], u[3][y], u[4][y], r);
#pragma unroll
    for (int a = 0; a < 4; ++a) t[a][y] = r[a];
  }
  float* wp = gw + ((long)o * C + c) * 16;
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    float r[4];
    gt(t[a][0], t[a][1], t[a][2], t[a][3], t[a][4], r);                // (G^T dU) G
#pragma unroll
    for (int b = 0; b < 4; ++b) wp[a * 4 + b] = beta != 0.f ? wp[a * 4 + b] + r[b] : r[b];
  }
}

struct W44Plan { int OH, OW, ty, tx; long P; size_t u_off, v_off, m_off, total; };
W44Plan w44_plan(int NI, int H, int W, int C, int Cout, int pad) {
  W44Plan p;
  p.OH = H + 2 * pad - 3; p.OW = W + 2 * pad - 3;
  p.ty = (p.OH + 1) / 2; p.tx = (p.OW + 1) / 2;
  p.P = (long)NI * p.ty * p.tx;
  auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
  p.u_off = 0;
  p.v_off = up((size_t)25 * C * Cout * 4);
  p.m_off = p.v_off + up((size_t)p.P * 25 * C * 4);
  p.total = p.m_off + up((size_t)p.P * 25 * Cout * 4);
  return p;
}
}  // namespace

extern "C" size_t re2e_conv4x4_wino_workspace_bytes(int NI, int H, int W, int C, int Cout, int pad) {
  if (NI <= 0 || H <= 0 || W <= 0 || C <= 0 || Cout <= 0 || pad < 0 || H + 2 * pad < 4 || W + 2 * pad < 4) return 0;
  return w44_plan(NI, H, W, C, Cout, pad).total;
}

extern "C" int re2e_conv4x4_wino(const float* in, int NI, int H, int W, int C, const float* w, int Cout, int pad, int dgrad, float* out,
                                 void* workspace, size_t workspace_bytes, hipStream_t stream) {
  RE2E_CHECK_ARG(in && w && out && workspace, "null operand");
  RE2E_CHECK_ARG(NI > 0 && H > 0 && W > 0 && C > 0 && Cout > 0 && pad >= 0 && H + 2 * pad >= 4 && W + 2 * pad >= 4, "bad geometry");
  if (C % 16 || Cout % 4) { re2e_set_error("re2e_conv4x4_wino: C must be a multiple of 16 and Cout of 4 (got %d, %d)", C, Cout); return RE2E_EUNSUPPORTED; }
  const W44Plan p = w44_plan(NI, H, W, C, Cout, pad);
  RE2E_CHECK_ARG(workspace_bytes >= p.total, "workspace too small (re2e_conv4x4_wino_workspace_bytes)");
  RE2E_CHECK_ARG((reinterpret_cast<uintptr_t>(in) & 15) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0 &&
                 (reinterpret_cast<uintptr_t>(workspace) & 15) == 0, "in / out / workspace must be 16-byte aligned");
  if (p.P * 25 * C * 4 >= 0xFFFFFFF0L || (long)Cout * 25 * C * 4 >= 0xFFFFFFF0L || p.P >= 0x7fffffffL) {
    re2e_set_error("re2e_conv4x4_wino: transformed operand larger than 4 GiB");
    return RE2E_EUNSUPPORTED;
  }
  float* U = reinterpret_cast<float*>(static_cast<char*>(workspace) + p.u_off);
  float* V = reinterpret_cast<float*>(static_cast<char*>(workspace) + p.v_off);
  float* M = reinterpret_cast<float*>(static_cast<char*>(workspace) + p.m_off);
  // the weight tensor is (Cout_fwd, Cin_fwd, 4, 4) in both directions: forward Cout_fwd = Cout, Cin_fwd = C; data gradient the reverse
  hipLaunchKernelGGL(w44_filter_kernel, dim3((unsigned)cdiv((long)C * Cout, 256)), dim3(256), 0, stream, w, dgrad ? C : Cout, dgrad ? Cout : C, dgrad, U);
  hipLaunchKernelGGL(w44_input_kernel, dim3((unsigned)cdiv(p.P * (C / 4), 256)), dim3(256), 0, stream, in, NI, H, W, C, pad, p.ty, p.tx, V, 25L * C, (long)C, p.P);
  static const bool log_calls = getenv("RE2E_IGEMM_LOG") != nullptr;   // tools/igemm_table.py: ONE line for the engine launch, with the convolution's direct-form shape
  if (log_calls)
    fprintf(stderr, "[igemm] A=W44%s B=DenseK tile=256x128x16 vec=1 M=%ld N=%d K=%d splits=25\n", dgrad ? "D" : "F", (long)NI * p.OH * p.OW, Cout, 16 * C);
  const int rc = gemm_kslices((int)p.P, Cout, 25 * C, 25, V, 25L * C, U, 25L * C, M, stream, 1);
  if (rc != RE2E_OK) return rc;
  hipLaunchKernelGGL(w44_output_kernel, dim3((unsigned)cdiv(p.P * (Cout / 4), 256)), dim3(256), 0, stream, M, p.P, Cout, p.OH, p.OW, p.ty, p.tx, out);
  RE2E_LAUNCH_CHECK();
  return RE2E_OK;
}

namespace {
struct W44WgradPlan { int OH, OW, ty, tx, sub; long P, Ppad; size_t v_off, m_off, s_off, total; };
W44WgradPlan w44_wgrad_plan(int NI, int H, int W, int C, int Cout, int pad) {
  W44WgradPlan p;
  p.OH = H + 2 * pad - 3; p.OW = W + 2 * pad - 3;
  p.ty = (p.OH + 1) / 2; p.tx = (p.OW + 1) / 2;
  p.P = (long)NI * p.ty * p.tx;
  // K-slices per position: enough workgroups (C/128 x Cout/128 tiles x 25 x sub) for two rounds of the chip, >= 64 k-tiles each
  const long tiles = (long)cdiv(C, 128) * cdiv(Cout, 128) * 25;
  int sub = (int)((1024 + tiles - 1) / tiles);
  const long maxsub = p.P / (16 * 64);
  if (sub > maxsub) sub = (int)maxsub;
  if (sub < 1) sub = 1;
  p.sub = sub;
  const long q = 16L * sub;
  p.Ppad = (p.P + q - 1) / q * q;
  auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
  p.v_off = 0;
  p.m_off = up((size_t)25 * p.Ppad * C * 4);
  p.s_off = p.m_off + up((size_t)25 * p.Ppad * Cout * 4);
  p.total = p.s_off + up((size_t)25 * sub * C * Cout * 4);
  return p;
}
}  // namespace

extern "C" size_t re2e_conv4x4_wino_wgrad_workspace_bytes(int NI, int H, int W, int C, int Cout, int pad) {
  if (NI <= 0 || H <= 0 || W <= 0 || C <= 0 || Cout <= 0 || pad < 0 || H + 2 * pad < 4 || W + 2 * pad < 4) return 0;
  return w44_wgrad_plan(NI, H, W, C, Cout, pad).total;
}

extern "C" int re2e_conv4x4_wino_wgrad(const float* in, int NI, int H, int W, int C, const float* dout, int Cout, int pad, float* gw, float beta,
                                       void* workspace, size_t workspace_bytes, hipStream_t stream) {
  RE2E_CHECK_ARG(in && dout && gw && workspace, "null operand");
  RE2E_CHECK_ARG(NI > 0 && H > 0 && W > 0 && C > 0 && Cout > 0 && pad >= 0 && H + 2 * pad >= 4 && W + 2 * pad >= 4, "bad geometry");
  RE2E_CHECK_ARG(beta == 0.f || beta == 1.f, "beta must be 0 or 1");
  if (C % 4 || Cout % 4) { re2e_set_error("re2e_conv4x4_wino_wgrad: C and Cout must be multiples of 4 (got %d, %d)", C, Cout); return RE2E_EUNSUPPORTED; }
  const W44WgradPlan p = w44_wgrad_plan(NI, H, W, C, Cout, pad);
  RE2E_CHECK_ARG(workspace_bytes >= p.total, "workspace too small (re2e_conv4x4_wino_wgrad_workspace_bytes)");
  RE2E_CHECK_ARG((reinterpret_cast<uintptr_t>(in) & 15) == 0 && (reinterpret_cast<uintptr_t>(dout) & 15) == 0 &&
                 (reinterpret_cast<uintptr_t>(workspace) & 15) == 0, "in / dout / workspace must be 16-byte aligned");
  if (25 * p.Ppad * (long)(C > Cout ? C : Cout) * 4 >= 0xFFFFFFF0L) { re2e_set_error("re2e_conv4x4_wino_wgrad: transformed operand larger than 4 GiB"); return RE2E_EUNSUPPORTED; }
  float* V = reinterpret_cast<float*>(static_cast<char*>(workspace) + p.v_off);
  float* dM = reinterpret_cast<float*>(static_cast<char*>(workspace) + p.m_off);
  float* S = reinterpret_cast<float*>(static_cast<char*>(workspace) + p.s_off);
  hipLaunchKernelGGL(w44_input_kernel, dim3((unsigned)cdiv(p.Ppad * (C / 4), 256)), dim3(256), 0, stream, in, NI, H, W, C, pad, p.ty, p.tx, V, (long)C,
                     p.Ppad * C, p.Ppad);
  hipLaunchKernelGGL(w44_dout_kernel, dim3((unsigned)cdiv(p.Ppad * (Cout / 4), 256)), dim3(256), 0, stream, dout, NI, p.OH, p.OW, Cout, p.ty, p.tx, p.P,
                     p.Ppad, dM);
  static const bool log_calls = getenv("RE2E_IGEMM_LOG") != nullptr;
  if (log_calls)
    fprintf(stderr, "[igemm] A=W44W B=DenseM tile=128x128x16 vec=1 M=%d N=%d K=%ld splits=%d\n", 16 * C, Cout, (long)NI * p.OH * p.OW, 25 * p.sub);
  const int rc = gemm_kslices_tn(C, Cout, (int)(25 * p.Ppad), 25 * p.sub, V, (long)C, dM, (long)Cout, S, stream, 1);
  if (rc != RE2E_OK) return rc;
  hipLaunchKernelGGL(w44_wgrad_final_kernel, dim3((unsigned)cdiv((long)C * Cout, 256)), dim3(256), 0, stream, S, C, Cout, p.sub, gw, beta);
  RE2E_LAUNCH_CHECK();
  return RE2E_OK;
}

// K4: bidirectional LSTM recurrence with packed-sequence semantics, forward and BPTT, and the
// K8 LSTMCell pointwise kernels (decoder).
//
// Replaces the cuDNN/ATen nn.LSTM calls at model/e2e_encoder.py:128-132 (BLSTMP) and :168-170
// (BLSTM) and nn.LSTMCell at model/e2e_decoder.py:131.  The input projection x W_ih^T + b is a
// plain GEMM done by the caller (re2e_gemm); this file owns the sequential part.
//
// One launch per time step (a dependent kernel boundary costs ~1.5 us on MI355X, cheaper than any
// in-kernel grid barrier -- MI355X_MICROARCH.md price list), both directions in the same launch
// (blockIdx.z).  h_{t-1} W_hh^T is computed with v_mfma_f32_32x32x2_f32: a workgroup owns 8
// hidden units x 4 gates (= 32 gate columns) for 32 utterances, its waves split K=H and reduce
// through LDS, then the same workgroup applies the cell non-linearity -- gates never leave the
// CU between the matmul and the pointwise update.  Time-major buffers padded by one zero block at
// each end remove every boundary special case (h_{-1}=c_{-1}=0, reverse start).
#include "common.h"

namespace {

__device__ __forceinline__ void store_acc(float* red, const f32x16& acc, int lane) {
  int lr = lane & 31, lh = lane >> 5;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
    red[row * 33 + lr] = acc[r];
  }
}

template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void lstm_fwd_step(float* xg_f, float* xg_r, const float* __restrict__ whh_f,
                                                            const float* __restrict__ whh_r, float* ybuf, float* cbuf,
                                                            const int* __restrict__ lens, int T, int B, int H, int s) {
  extern __shared__ __attribute__((aligned(16))) float red[];   // [WAVES][32][33]
  const int dir = blockIdx.z;
  const int t = dir ? T - 1 - s : s;
  float* xg = dir ? xg_r : xg_f;
  const float* whh = dir ? whh_r : whh_f;
  const int j0 = blockIdx.x * 8, b0 = blockIdx.y * 32;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, lr = lane & 31, lh = lane >> 5;
  const int KC = H / WAVES, kc0 = wid * KC;
  const int prev_blk = dir ? t + 2 : t;
  const long H2 = 2L * H;
  const bool bvalid = (b0 + lr) < B;
  const float* hrow = ybuf + ((long)prev_blk * B + (bvalid ? b0 + lr : 0)) * H2 + dir * H + kc0 + lh * 4;
  const float* wrow = whh + (long)((lr >> 3) * H + j0 + (lr & 7)) * H + kc0 + lh * 4;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (int q = 0; q < KC / 8; ++q) {
    f32x4 a4 = {0.f, 0.f, 0.f, 0.f};
    if (bvalid) a4 = *reinterpret_cast<const f32x4*>(hrow + q * 8);
    f32x4 b4 = *reinterpret_cast<const f32x4*>(wrow + q * 8);
#pragma unroll
    for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[j], b4[j], acc, 0, 0, 0);
  }
  store_acc(red + wid * (32 * 33), acc, lane);
  __syncthreads();
  for (int idx = tid; idx < 256; idx += WAVES * 64) {
    int bm = idx >> 3, jj = idx & 7;
    int b = b0 + bm;
    if (b >= B) continue;
    int j = j0 + jj;
    float pre[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float v = 0.f;
      for (int w = 0; w < WAVES; ++w) v += red[w * (32 * 33) + bm * 33 + g * 8 + jj];
      pre[g] = v + xg[((long)t * B + b) * 4 * H + g * H + j];
    }
    float gi = sigmoidf_(pre[0]), gf = sigmoidf_(pre[1]), gg = tanhf_(pre[2]), go = sigmoidf_(pre[3]);
    float cp = cbuf[((long)prev_blk * B + b) * H2 + dir * H + j];
    float c = gf * cp + gi * gg;
    float h = go * tanhf_(c);
    if (t >= lens[b]) { c = 0.f; h = 0.f; }      // packed semantics: padded outputs 0, state stays 0
    float* go_ = xg + ((long)t * B + b) * 4 * H + j;
    go_[0] = gi; go_[H] = gf; go_[2 * H] = gg; go_[3 * H] = go;
    cbuf[((long)(t + 1) * B + b) * H2 + dir * H + j] = c;
    ybuf[((long)(t + 1) * B + b) * H2 + dir * H + j] = h;
  }
}

template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void lstm_bwd_step(float* g_f, float* g_r, const float* __restrict__ whhT_f,
                                                            const float* __restrict__ whhT_r, const float* __restrict__ dy,
                                                            const float* __restrict__ cbuf, float* dc_state,
                                                            const int* __restrict__ lens, int T, int B, int H, int s) {
  extern __shared__ __attribute__((aligned(16))) float red[];   // [WAVES][32][33]
  const int dir = blockIdx.z;
  const int t = dir ? s : T - 1 - s;            // reverse order of the forward pass
  const int tnext = dir ? t - 1 : t + 1;        // step whose dgates were produced by the previous launch
  float* G = dir ? g_r : g_f;
  const float* whhT = dir ? whhT_r : whhT_f;
  const int j0 = blockIdx.x * 32, b0 = blockIdx.y * 32;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, lr = lane & 31, lh = lane >> 5;
  const int K4 = 4 * H, KC = K4 / WAVES, kc0 = wid * KC;
  const long H2 = 2L * H;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  if (s > 0) {
    const bool bvalid = (b0 + lr) < B, jvalid = (j0 + lr) < H;
    const float* arow = G + ((long)tnext * B + (bvalid ? b0 + lr : 0)) * K4 + kc0 + lh * 4;
    const float* brow = whhT + (long)(jvalid ? j0 + lr : 0) * K4 + kc0 + lh * 4;
    for (int q = 0; q < KC / 8; ++q) {
      f32x4 a4 = {0.f, 0.f, 0.f, 0.f}, b4 = {0.f, 0.f, 0.f, 0.f};
      if (bvalid) a4 = *reinterpret_cast<const f32x4*>(arow + q * 8);
      if (jvalid) b4 = *reinterpret_cast<const f32x4*>(brow + q * 8);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[j], b4[j], acc, 0, 0, 0);
    }
  }
  store_acc(red + wid * (32 * 33), acc, lane);
  __syncthreads();
  const int prev_blk = dir ? t + 2 : t;
  for (int idx = tid; idx < 1024; idx += WAVES * 64) {
    int bm = idx >> 5, jc = idx & 31;
    int b = b0 + bm, j = j0 + jc;
    if (b >= B || j >= H) continue;
    float dh = dy[((long)t * B + b) * H2 + dir * H + j];
    for (int w = 0; w < WAVES; ++w) dh += red[w * (32 * 33) + bm * 33 + jc];
    float* gp = G + ((long)t * B + b) * K4 + j;
    float gi = gp[0], gf = gp[H], gg = gp[2 * H], go = gp[3 * H];
    float c = cbuf[((long)(t + 1) * B + b) * H2 + dir * H + j];
    float cp = cbuf[((long)prev_blk * B + b) * H2 + dir * H + j];
    float dc = dc_state[(long)b * H2 + dir * H + j];
    float di = 0.f, df = 0.f, dg = 0.f, dout = 0.f, dcp = dc;
    if (t < lens[b]) {
      float tc = tanhf_(c);
      float dct = dh * go * (1.f - tc * tc) + dc;
      dout = dh * tc * go * (1.f - go);
      di = dct * gg * gi * (1.f - gi);
      df = dct * cp * gf * (1.f - gf);
      dg = dct * gi * (1.f - gg * gg);
      dcp = dct * gf;
    }
    gp[0] = di; gp[H] = df; gp[2 * H] = dg; gp[3 * H] = dout;
    dc_state[(long)b * H2 + dir * H + j] = dcp;
  }
}

__global__ void zero_kernel(float* p, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = 0.f;
}

int pick_waves(int K, int min_kc) {
  for (int w = 16; w >= 1; w >>= 1)
    if (K % (8 * w) == 0 && K / w >= min_kc) return w;
  return K % 8 == 0 ? 1 : 0;
}

template <int W>
void launch_fwd(dim3 grid, hipStream_t st, float* xg_f, float* xg_r, const float* whh_f, const float* whh_r, float* ybuf,
                float* cbuf, const int* lens, int T, int B, int H) {
  size_t lds = (size_t)W * 32 * 33 * sizeof(float);
  static bool done = false;
  if (!done) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&lstm_fwd_step<W>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); done = true; }
  for (int s = 0; s < T; ++s)
    hipLaunchKernelGGL((lstm_fwd_step<W>), grid, dim3(W * 64), lds, st, xg_f, xg_r, whh_f, whh_r, ybuf, cbuf, lens, T, B, H, s);
}
template <int W>
void launch_bwd(dim3 grid, hipStream_t st, float* g_f, float* g_r, const float* wt_f, const float* wt_r, const float* dy,
                const float* cbuf, float* dc, const int* lens, int T, int B, int H) {
  size_t lds = (size_t)W * 32 * 33 * sizeof(float);
  static bool done = false;
  if (!done) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&lstm_bwd_step<W>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); done = true; }
  for (int s = 0; s < T; ++s)
    hipLaunchKernelGGL((lstm_bwd_step<W>), grid, dim3(W * 64), lds, st, g_f, g_r, wt_f, wt_r, dy, cbuf, dc, lens, T, B, H, s);
}

}  // namespace

extern "C" int re2e_lstm_seq_fwd(float* xg_f, float* xg_r, const float* whh_f, const float* whh_r, float* ybuf, float* cbuf,
                                 const int* lens_dev, int T, int B, int H, hipStream_t stream) {
  RE2E_CHECK_ARG(xg_f && xg_r && whh_f && whh_r && ybuf && cbuf && lens_dev, "null arg");
  RE2E_CHECK_ARG(T > 0 && B > 0 && H > 0, "bad shape");
  if (H % 8 != 0) { re2e_set_error("re2e_lstm_seq_fwd: hidden size must be a multiple of 8 (got %d)", H); return RE2E_EUNSUPPORTED; }
  int w = pick_waves(H, 32);
  dim3 grid(H / 8, cdiv(B, 32), 2);
  switch (w) {
    case 16: launch_fwd<16>(grid, stream, xg_f, xg_r, whh_f, whh_r, ybuf, cbuf, lens_dev, T, B, H); break;
    case 8: launch_fwd<8>(grid, stream, xg_f, xg_r, whh_f, whh_r, ybuf, cbuf, lens_dev, T, B, H); break;
    case 4: launch_fwd<4>(grid, stream, xg_f, xg_r, whh_f, whh_r, ybuf, cbuf, lens_dev, T, B, H); break;
    case 2: launch_fwd<2>(grid, stream, xg_f, xg_r, whh_f, whh_r, ybuf, cbuf, lens_dev, T, B, H); break;
    default: launch_fwd<1>(grid, stream, xg_f, xg_r, whh_f, whh_r, ybuf, cbuf, lens_dev, T, B, H); break;
  }
  RE2E_LAUNCH_CHECK();
  return RE2E_OK;
}

extern "C" int re2e_lstm_seq_bwd(float* g_f, float* g_r, const float* whhT_f, const float* whhT_r, const float* dy,
                                 const float* ybuf, const float* cbuf, float* dc_state, const int* lens_dev, int T, int B, int H,
                                 hipStream_t stream) {
  (void)ybuf;
  RE2E_CHECK_ARG(g_f && g_r && whhT_f && whhT_r && dy && cbuf && dc_state && lens_dev, "null arg");
  RE2E_CHECK_ARG(T > 0 && B > 0 && H > 0, "bad shape");
  if (H % 8 != 0) { re2e_set_error("re2e_lstm_seq_bwd: hidden size must be a multiple of 8 (got %d)", H); return RE2E_EUNSUPPORTED; }
  long nz = (long)B * 2 * H;
  hipLaunchKernelGGL(zero_kernel, dim3(cdiv(nz, 256)), dim3(256), 0, stream, dc_state, nz);
  int w = pick_waves(4 * H, 64);
  dim3 grid(cdiv(H, 32), cdiv(B, 32), 2);
  switch (w) {
    case 16: launch_bwd<16>(grid, stream, g_f, g_r, whhT_f, whhT_r, dy, cbuf, dc_state, lens_dev, T, B, H); break;
    case 8: launch_bwd<8>(grid, stream, g_f, g_r, whhT_f, whhT_r, dy, cbuf, dc_state, lens_dev, T, B, H); break;
    case 4: launch_bwd<4>(grid, stream, g_f, g_r, whhT_f, whhT_r, dy, cbuf, dc_state, lens_dev, T, B, H); break;
    case 2: launch_bwd<2>(grid, stream, g_f, g_r, whhT_f, whhT_r, dy, cbuf, dc_state, lens_dev, T, B, H); break;
    default: launch_bwd<1>(grid, stream, g_f, g_r, whhT_f, whhT_r, dy, cbuf, dc_state, lens_dev, T, B, H); break;
  }
  RE2E_LAUNCH_CHECK();
  return RE2E_OK;
}

// ---- LSTMCell pointwise (decoder) --------------------------------------------------------------
__global__ void lstm_cell_fwd_kernel(float* gates, const float* __restrict__ c_prev, float* c_out, float* h_out, int B, int H) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * H) return;
  int b = i / H, j = i % H;
  float* g = gates + (long)b * 4 * H + j;
  float gi = sigmoidf_(g[0]), gf = sigmoidf_(g[H]), gg = tanhf_(g[2 * H]), go = sigmoidf_(g[3 * H]);
  float c = gf * c_prev[i] + gi * gg;
  g[0] = gi; g[H] = gf; g[2 * H] = gg; g[3 * H] = go;
  c_out[i] = c;
  h_out[i] = go * tanhf_(c);
}
extern "C" int re2e_lstm_cell_fwd(float* gates, const float* c_prev, float* c_out, float* h_out, int B, int H, hipStream_t stream) {
  RE2E_CHECK_ARG(gates && c_prev && c_out && h_out && B > 0 && H > 0, "bad args");
  hipLaunchKernelGGL(lstm_cell_fwd_kernel, dim3(cdiv((long)B * H, 256)), dim3(256), 0, stream, gates, c_prev, c_out, h_out, B, H);
  RE2E_LAUNCH_CHECK();
  return RE2E_OK;
}
__global__ void lstm_cell_bwd_kernel(float* gates, const float* __restrict__ c_prev, const float* __restrict__ c_cur,
                                     const float* __restrict__ dh, const float* __restrict__ dc_in, float* dc_prev_out, int B, int H) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * H) return;
  int b = i / H, j = i % H;
  float* g = gates + (long)b * 4 * H + j;
  float gi = g[0], gf = g[H], gg = g[2 * H], go = g[3 * H];
  float tc = tanhf_(c_cur[i]);
  float d = dh[i];
  float dct = d * go * (1.f - tc * tc) + (dc_in ? dc_in[i] : 0.f);
  g[0] = dct * gg * gi * (1.f - gi);
  g[H] = dct * c_prev[i] * gf * (1.f - gf);
  g[2 * H] = dct * gi * (1.f - gg * gg);
  g[3 * H] = d * tc * go * (1.f - go);
  dc_prev_out[i] = dct * gf;
}
extern "C" int re2e_lstm_cell_bwd(float* gates, const float* c_prev, const float* c_cur, const float* dh, const float* dc_in,
                                  float* dc_prev_out, int B, int H, hipStream_t stream) {
  RE2E_CHECK_ARG(gates && c_prev && c_cur && dh && dc_prev_out && B > 0 && H > 0, "bad args");
  hipLaunchKernelGGL(lstm_cell_bwd_kernel, dim3(cdiv((long)B * H, 256)), dim3(256), 0, stream, gates, c_prev, c_cur, dh, dc_in,
                     dc_prev_out, B, H);
  RE2E_LAUNCH_CHECK();
  return RE2E_OK;
}

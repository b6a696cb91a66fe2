// K7: fused location-aware attention step (AttLoc.forward, model/e2e_attention.py:258-297) and
// its backward.  One workgroup (512 threads = 8 wavefronts) per utterance keeps the whole step on
// one CU: previous attention weights, the location convolution (k = 2*filts+1 taps), the energy
// vector and the softmax live in LDS; softmax / dot-product reductions use wavefront shuffles;
// pre_compute_enc_h and enc_h rows are streamed with coalesced (row-contiguous) loads.
//
//   w = softmax_t( 2 * ( gvec . tanh( W_att conv(att_prev)[t] + pre[b,t] + W_dec z[b] ) + gb ) )
//   c = sum_t w[t] * enc[b,t]          (softmax over ALL T frames incl. padding, Appendix A.9)
#include "common.h"

namespace {
constexpr int NT = 512;
constexpr int NW = NT / 64;
constexpr int CG = 5;       // channels per conv work item

struct Lds {
  float *zs, *dp, *ap, *conv, *e, *w, *scr, *red;
};
__device__ __forceinline__ int cpad(int C) { return (C + 3) & ~3; }

// common prologue: z -> LDS, att_prev (or uniform init) -> padded LDS, dec_proj, location conv
__device__ void prologue(const Lds& L, const float* z, const float* att_prev, int hl, const float* w_decT, const float* w_conv,
                         int b, int T, int D, int A, int C, int F) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int Kf = 2 * F + 1, CP = cpad(C);
  for (int d = tid; d < D; d += NT) L.zs[d] = z ? z[(long)b * D + d] : 0.f;
  for (int i = tid; i < T + 2 * F; i += NT) {
    int t = i - F;
    float v = 0.f;
    if (t >= 0 && t < T) v = att_prev ? att_prev[(long)b * T + t] : (t < hl ? 1.0f / (float)hl : 0.f);
    L.ap[i] = v;
  }
  __syncthreads();
  // dec_proj[a] = sum_d W_dec[a][d] z[d]; w_decT is W_dec transposed (D, A): lanes run over a => coalesced,
  // independent loads (8 in flight per lane)
  for (int a = tid; a < A; a += NT) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int d = 0;
    for (; d + 8 <= D; d += 8) {
      float w0 = w_decT[(long)(d + 0) * A + a], w1 = w_decT[(long)(d + 1) * A + a], w2 = w_decT[(long)(d + 2) * A + a],
            w3 = w_decT[(long)(d + 3) * A + a], w4 = w_decT[(long)(d + 4) * A + a], w5 = w_decT[(long)(d + 5) * A + a],
            w6 = w_decT[(long)(d + 6) * A + a], w7 = w_decT[(long)(d + 7) * A + a];
      s0 += w0 * L.zs[d] + w4 * L.zs[d + 4]; s1 += w1 * L.zs[d + 1] + w5 * L.zs[d + 5];
      s2 += w2 * L.zs[d + 2] + w6 * L.zs[d + 6]; s3 += w3 * L.zs[d + 3] + w7 * L.zs[d + 7];
    }
    for (; d < D; ++d) s0 += w_decT[(long)d * A + a] * L.zs[d];
    L.dp[a] = (s0 + s1) + (s2 + s3);
  }
  // conv[t][c] = sum_k w_conv[c][k] * ap[t+k]; a wavefront owns one channel group so the filter
  // taps are wave-uniform (scalar loads), lanes run over t
  const int ncg = (C + CG - 1) / CG;
  const int tchunks = (T + 63) / 64;
  for (int item = wid; item < ncg * tchunks; item += NW) {
    int cg = item % ncg, tc = item / ncg;
    int c0 = cg * CG;
    int t = tc * 64 + lane;
    float acc[CG];
#pragma unroll
    for (int cc = 0; cc < CG; ++cc) acc[cc] = 0.f;
    int tt = t < T ? t : T - 1;
#pragma unroll 8
    for (int k = 0; k < Kf; ++k) {
      float x = L.ap[tt + k];
#pragma unroll
      for (int cc = 0; cc < CG; ++cc)
        if (c0 + cc < C) acc[cc] += w_conv[(c0 + cc) * Kf + k] * x;
    }
    if (t < T) {
#pragma unroll
      for (int cc = 0; cc < CG; ++cc)
        if (c0 + cc < C) L.conv[t * CP + c0 + cc] = acc[cc];
    }
  }
  __syncthreads();
}

template <int CMAX, int AIMAX>
__global__ __launch_bounds__(NT) void attloc_fwd_kernel(const float* __restrict__ pre, const float* __restrict__ enc,
                                                        const float* __restrict__ z, const float* __restrict__ att_prev,
                                                        const int* __restrict__ hlens, const float* __restrict__ w_decT,
                                                        const float* __restrict__ w_att, const float* __restrict__ w_conv,
                                                        const float* __restrict__ gvec, const float* __restrict__ gvec_b, int B,
                                                        int T, int E, int D, int A, int C, int F, float* __restrict__ w_out,
                                                        float* __restrict__ c_out, long ldc_out) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int CP = cpad(C);
  Lds L;
  float* p = sm;
  L.zs = p; p += (D + 3) & ~3;
  L.dp = p; p += (A + 3) & ~3;
  L.ap = p; p += (T + 2 * F + 3) & ~3;
  L.conv = p; p += T * CP;
  L.e = p; p += (T + 3) & ~3;
  L.w = p; p += (T + 3) & ~3;
  L.red = p; p += 32;
  L.scr = p;     // [ntg][E] context partials
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hl = hlens[b];
  prologue(L, z, att_prev, hl, w_decT, w_conv, b, T, D, A, C, F);

  // ---- energies: one wavefront per frame, lanes over the attention dimension ----
  float wa[AIMAX][CMAX], gv[AIMAX], dpv[AIMAX];
#pragma unroll
  for (int i = 0; i < AIMAX; ++i) {
    int a = lane + 64 * i;
    gv[i] = a < A ? gvec[a] : 0.f;
    dpv[i] = a < A ? L.dp[a] : 0.f;
#pragma unroll
    for (int c = 0; c < CMAX; ++c) wa[i][c] = (a < A && c < C) ? w_att[a * C + c] : 0.f;
  }
  const float gb = gvec_b[0];
  for (int tb = wid; tb < T; tb += 2 * NW) {       // two frames per iteration: 2*AIMAX independent row loads in flight
    const int t0 = tb, t1 = tb + NW;
    const bool h1 = t1 < T;
    float p0[AIMAX], p1[AIMAX];
    const float* pr0 = pre + ((long)b * T + t0) * A;
    const float* pr1 = pre + ((long)b * T + (h1 ? t1 : t0)) * A;
#pragma unroll
    for (int i = 0; i < AIMAX; ++i) {
      int a = lane + 64 * i;
      p0[i] = a < A ? pr0[a] : 0.f;
      p1[i] = a < A ? pr1[a] : 0.f;
    }
    float cv0[CMAX], cv1[CMAX];
#pragma unroll
    for (int c = 0; c < CMAX; ++c) { cv0[c] = c < C ? L.conv[t0 * CP + c] : 0.f; cv1[c] = (c < C && h1) ? L.conv[t1 * CP + c] : 0.f; }
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int i = 0; i < AIMAX; ++i) {
      int a = lane + 64 * i;
      if (a < A) {
        float x0 = p0[i] + dpv[i], x1 = p1[i] + dpv[i];
#pragma unroll
        for (int c = 0; c < CMAX; ++c) { x0 += wa[i][c] * cv0[c]; x1 += wa[i][c] * cv1[c]; }
        s0 += gv[i] * tanhf_(x0);
        s1 += gv[i] * tanhf_(x1);
      }
    }
    s0 = wave_sum(s0);
    s1 = wave_sum(s1);
    if (lane == 0) { L.e[t0] = s0 + gb; if (h1) L.e[t1] = s1 + gb; }
  }
  __syncthreads();
  // ---- softmax(2e) over all T frames ----
  float m = -3.0e38f;
  for (int t = tid; t < T; t += NT) m = fmaxf(m, 2.f * L.e[t]);
  m = block_max(m, L.red);
  float sum = 0.f;
  for (int t = tid; t < T; t += NT) { float v = __expf(2.f * L.e[t] - m); L.w[t] = v; sum += v; }
  sum = block_sum(sum, L.red);
  float inv = 1.0f / sum;
  __syncthreads();
  for (int t = tid; t < T; t += NT) { float v = L.w[t] * inv; L.w[t] = v; w_out[(long)b * T + t] = v; }
  __syncthreads();
  // ---- context c = sum_t w[t] enc[b,t,:] ; float4 columns, frame groups reduced through LDS ----
  const int per = E / 4;          // host guarantees per <= NT
  const int ntg = NT / per;
  if (tid < ntg * per) {
    const int tg = tid / per, d4 = tid % per;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const f32x4* er = reinterpret_cast<const f32x4*>(enc + (long)b * T * E) + d4;
    int t = tg;
    for (; t + 3 * ntg < T; t += 4 * ntg) {          // 4 independent row loads in flight
      f32x4 v0 = er[(long)t * per], v1 = er[(long)(t + ntg) * per], v2 = er[(long)(t + 2 * ntg) * per], v3 = er[(long)(t + 3 * ntg) * per];
      acc += v0 * L.w[t] + v1 * L.w[t + ntg] + v2 * L.w[t + 2 * ntg] + v3 * L.w[t + 3 * ntg];
    }
    for (; t < T; t += ntg) acc += er[(long)t * per] * L.w[t];
    *reinterpret_cast<f32x4*>(L.scr + (long)tg * E + d4 * 4) = acc;
  }
  __syncthreads();
  for (int d = tid; d < E; d += NT) {
    float s = 0.f;
    for (int g = 0; g < ntg; ++g) s += L.scr[(long)g * E + d];
    c_out[(long)b * ldc_out + d] = s;
  }
}

template <int CMAX, int AIMAX>
__global__ __launch_bounds__(NT) void attloc_bwd_kernel(
    const float* __restrict__ pre, const float* __restrict__ enc, const float* __restrict__ z, const float* __restrict__ att_prev,
    const float* __restrict__ w_cur, const int* __restrict__ hlens, const float* __restrict__ w_decT, const float* __restrict__ w_att,
    const float* __restrict__ w_conv, const float* __restrict__ gvec, const float* __restrict__ dc, long ld_dc,
    const float* __restrict__ dw_in, int B, int T, int E, int D, int A, int C, int F, float* d_pre, float* d_enc,
    float* __restrict__ d_att_prev, float* __restrict__ d_decproj, float* partials) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int CP = cpad(C), Kf = 2 * F + 1;
  Lds L;
  float* p = sm;
  L.zs = p; p += (D + 3) & ~3;
  L.dp = p; p += (A + 3) & ~3;
  L.ap = p; p += (T + 2 * F + 3) & ~3;
  L.conv = p; p += T * CP;
  L.e = p; p += (T + 3) & ~3;       // dw / de
  L.w = p; p += (T + 3) & ~3;
  L.red = p; p += 32;
  float* dcs = p; p += (E + 3) & ~3;                 // dc[b,:]
  float* dcp = p; p += (T + 2 * F) * CP;             // d_conv, zero padded by F frames each side
  float* accum = p; p += A * (CMAX + 2);             // [A][ddp | dgv | dwa[C]]
  float* scr = p;                                    // [2][C*Kf] dw_conv halves / [ncg][T] d_att_prev partials
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hl = hlens[b];
  prologue(L, z, att_prev, hl, w_decT, w_conv, b, T, D, A, C, F);
  for (int d = tid; d < E; d += NT) dcs[d] = dc[(long)b * ld_dc + d];
  for (int t = tid; t < T; t += NT) L.w[t] = w_cur[(long)b * T + t];
  for (int i = tid; i < (T + 2 * F) * CP; i += NT) dcp[i] = 0.f;
  for (int i = tid; i < A * (CMAX + 2); i += NT) accum[i] = 0.f;
  __syncthreads();
  // ---- 1. dw[t] = dw_in[t] + dc . enc[t] ; d_enc[t] += w[t] dc ----
  {
    const int per = E / 4;
    for (int tb = wid; tb < T; tb += 2 * NW) {     // two frames per iteration
      const int t0 = tb, t1 = tb + NW;
      const bool h1 = t1 < T;
      const f32x4* er0 = reinterpret_cast<const f32x4*>(enc + ((long)b * T + t0) * E);
      const f32x4* er1 = reinterpret_cast<const f32x4*>(enc + ((long)b * T + (h1 ? t1 : t0)) * E);
      f32x4* der0 = reinterpret_cast<f32x4*>(d_enc + ((long)b * T + t0) * E);
      f32x4* der1 = reinterpret_cast<f32x4*>(d_enc + ((long)b * T + (h1 ? t1 : t0)) * E);
      const float wt0 = L.w[t0], wt1 = h1 ? L.w[t1] : 0.f;
      float s0 = 0.f, s1 = 0.f;
      for (int d4 = lane; d4 < per; d4 += 64) {
        f32x4 v0 = er0[d4], v1 = er1[d4], o0 = der0[d4], o1 = der1[d4];
        f32x4 g = *reinterpret_cast<const f32x4*>(dcs + d4 * 4);
        s0 += v0[0] * g[0] + v0[1] * g[1] + v0[2] * g[2] + v0[3] * g[3];
        s1 += v1[0] * g[0] + v1[1] * g[1] + v1[2] * g[2] + v1[3] * g[3];
        der0[d4] = o0 + g * wt0;
        if (h1) der1[d4] = o1 + g * wt1;
      }
      s0 = wave_sum(s0);
      s1 = wave_sum(s1);
      if (lane == 0) {
        L.e[t0] = s0 + (dw_in ? dw_in[(long)b * T + t0] : 0.f);
        if (h1) L.e[t1] = s1 + (dw_in ? dw_in[(long)b * T + t1] : 0.f);
      }
    }
  }
  __syncthreads();
  // ---- 2. softmax backward (scaling 2) ----
  float sd = 0.f;
  for (int t = tid; t < T; t += NT) sd += L.w[t] * L.e[t];
  sd = block_sum(sd, L.red);
  __syncthreads();
  for (int t = tid; t < T; t += NT) L.e[t] = 2.f * L.w[t] * (L.e[t] - sd);     // de[t]
  __syncthreads();
  // ---- 3. energy backward ----
  float wa[AIMAX][CMAX], gv[AIMAX], dpv[AIMAX], ddp[AIMAX], dgv[AIMAX], dwa[AIMAX][CMAX];
#pragma unroll
  for (int i = 0; i < AIMAX; ++i) {
    int a = lane + 64 * i;
    gv[i] = a < A ? gvec[a] : 0.f;
    dpv[i] = a < A ? L.dp[a] : 0.f;
    ddp[i] = 0.f; dgv[i] = 0.f;
#pragma unroll
    for (int c = 0; c < CMAX; ++c) { wa[i][c] = (a < A && c < C) ? w_att[a * C + c] : 0.f; dwa[i][c] = 0.f; }
  }
  float dgb = 0.f;
  for (int t = wid; t < T; t += NW) {
    float cv[CMAX], dcv[CMAX];
#pragma unroll
    for (int c = 0; c < CMAX; ++c) { cv[c] = c < C ? L.conv[t * CP + c] : 0.f; dcv[c] = 0.f; }
    const float* pr = pre + ((long)b * T + t) * A;
    float* dpr = d_pre + ((long)b * T + t) * A;
    const float det = L.e[t];
    dgb += det;
#pragma unroll
    for (int i = 0; i < AIMAX; ++i) {
      int a = lane + 64 * i;
      if (a < A) {
        float x = pr[a] + dpv[i];
#pragma unroll
        for (int c = 0; c < CMAX; ++c) x += wa[i][c] * cv[c];
        float u = tanhf_(x);
        float du = det * gv[i] * (1.f - u * u);
        dpr[a] += du;
        ddp[i] += du;
        dgv[i] += det * u;
#pragma unroll
        for (int c = 0; c < CMAX; ++c) { dwa[i][c] += du * cv[c]; dcv[c] += du * wa[i][c]; }
      }
    }
#pragma unroll
    for (int c = 0; c < CMAX; ++c) {
      if (c < C) {
        float v = wave_sum(dcv[c]);
        if (lane == 0) dcp[(t + F) * CP + c] = v;
      }
    }
  }
  // fixed-order cross-wavefront reduction of the per-lane partial sums
  for (int w = 0; w < NW; ++w) {
    if (wid == w) {
#pragma unroll
      for (int i = 0; i < AIMAX; ++i) {
        int a = lane + 64 * i;
        if (a < A) {
          float* q = accum + a * (CMAX + 2);
          q[0] += ddp[i]; q[1] += dgv[i];
#pragma unroll
          for (int c = 0; c < CMAX; ++c) if (c < C) q[2 + c] += dwa[i][c];
        }
      }
      if (lane == 0) L.red[16 + w] = dgb;
    }
    __syncthreads();
  }
  const int P_GB = A, P_WATT = A + 1, P_WCONV = A + 1 + A * C;
  float* part = partials + (long)b * (A + 1 + A * C + C * Kf);
  for (int a = tid; a < A; a += NT) {
    const float* q = accum + a * (CMAX + 2);
    d_decproj[(long)b * A + a] = q[0];
    part[a] += q[1];
    for (int c = 0; c < C; ++c) part[P_WATT + a * C + c] += q[2 + c];
  }
  if (tid == 0) {
    float s = 0.f;
    for (int w = 0; w < NW; ++w) s += L.red[16 + w];
    part[P_GB] += s;
  }
  // ---- 4a. d_att_prev[t'] = sum_{c,k} w_conv[c][k] * d_conv[t'-k+F][c] ----
  const int ncg = (C + CG - 1) / CG;
  const int tchunks = (T + 63) / 64;
  if (d_att_prev) {
    for (int item = wid; item < ncg * tchunks; item += NW) {
      int cg = item % ncg, tc = item / ncg;
      int c0 = cg * CG;
      int t = tc * 64 + lane;
      int tt = t < T ? t : T - 1;
      float acc = 0.f;
      for (int k = 0; k < Kf; ++k) {
        const float* q = dcp + (tt - k + 2 * F) * CP + c0;
#pragma unroll
        for (int cc = 0; cc < CG; ++cc)
          if (c0 + cc < C) acc += w_conv[(c0 + cc) * Kf + k] * q[cc];
      }
      if (t < T) scr[cg * T + t] = acc;
    }
    __syncthreads();
    for (int t = tid; t < T; t += NT) {
      float s = 0.f;
      for (int g = 0; g < ncg; ++g) s += scr[g * T + t];
      d_att_prev[(long)b * T + t] = s;
    }
    __syncthreads();
  }
  // ---- 4b. dw_conv[c][k] = sum_t d_conv[t][c] * ap[t+k] ; two threads per tap split the frames ----
  {
    const int half = (T + 1) / 2;
    for (int item = tid; item < 2 * Kf; item += NT) {
      int k = item % Kf, hf = item / Kf;
      int t0 = hf * half, t1 = min(T, t0 + half);
      float acc[CMAX];
#pragma unroll
      for (int c = 0; c < CMAX; ++c) acc[c] = 0.f;
      for (int t = t0; t < t1; ++t) {
        float x = L.ap[t + k];
        const float* q = dcp + (t + F) * CP;
#pragma unroll
        for (int c = 0; c < CMAX; ++c) if (c < C) acc[c] += q[c] * x;
      }
#pragma unroll
      for (int c = 0; c < CMAX; ++c) if (c < C) scr[hf * (C * Kf) + c * Kf + k] = acc[c];
    }
    __syncthreads();
    for (int i = tid; i < C * Kf; i += NT) part[P_WCONV + i] += scr[i] + scr[C * Kf + i];
  }
}

size_t fwd_lds_floats(int T, int E, int D, int A, int C, int F) {
  int CP = (C + 3) & ~3;
  int per = E / 4;
  int ntg = NT / per;
  return (size_t)((D + 3) & ~3) + ((A + 3) & ~3) + ((T + 2 * F + 3) & ~3) + (size_t)T * CP + 2 * ((T + 3) & ~3) + 32 + (size_t)ntg * E + 16;
}
size_t bwd_lds_floats(int T, int E, int D, int A, int C, int F, int CMAX) {
  int CP = (C + 3) & ~3, Kf = 2 * F + 1;
  size_t scr = (size_t)2 * C * Kf;
  size_t scr2 = (size_t)((C + CG - 1) / CG) * T;
  if (scr2 > scr) scr = scr2;
  return (size_t)((D + 3) & ~3) + ((A + 3) & ~3) + ((T + 2 * F + 3) & ~3) + (size_t)T * CP + 2 * ((T + 3) & ~3) + 32 + ((E + 3) & ~3) +
         (size_t)(T + 2 * F) * CP + (size_t)A * (CMAX + 2) + scr + 16;
}
}  // namespace

extern "C" size_t re2e_attloc_partial_floats(int adim, int chans, int filts) {
  return (size_t)adim + 1 + (size_t)adim * chans + (size_t)chans * (2 * filts + 1);
}

extern "C" int re2e_attloc_fwd(const float* pre, const float* enc, const float* z, const float* att_prev, const int* hlens,
                               const float* w_decT, const float* w_att, const float* w_conv, const float* gvec, const float* gvec_b,
                               int B, int T, int eprojs, int dunits, int adim, int chans, int filts, float* w_out, float* c_out,
                               long ldc_out, hipStream_t stream) {
  RE2E_CHECK_ARG(pre && enc && hlens && w_decT && w_att && w_conv && gvec && gvec_b && w_out && c_out, "null arg");
  RE2E_CHECK_ARG(B > 0 && T > 0 && eprojs > 0 && eprojs % 4 == 0 && eprojs <= 2048 && dunits > 0 && adim > 0 && chans > 0 && filts >= 0, "bad shape");
  if (chans > 16 || adim > 512) { re2e_set_error("re2e_attloc_fwd: chans<=16 and adim<=512 supported"); return RE2E_EUNSUPPORTED; }
  size_t lds = fwd_lds_floats(T, eprojs, dunits, adim, chans, filts) * sizeof(float);
  if (lds > 160 * 1024) { re2e_set_error("re2e_attloc_fwd: T=%d needs %zu bytes of LDS (>160 KiB)", T, lds); return RE2E_EUNSUPPORTED; }
  if (chans <= 12 && adim <= 320) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attloc_fwd_kernel<12, 5>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((attloc_fwd_kernel<12, 5>), dim3(B), dim3(NT), lds, stream, pre, enc, z, att_prev, hlens, w_decT, w_att, w_conv,
                       gvec, gvec_b, B, T, eprojs, dunits, adim, chans, filts, w_out, c_out, ldc_out);
  } else {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attloc_fwd_kernel<16, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((attloc_fwd_kernel<16, 8>), dim3(B), dim3(NT), lds, stream, pre, enc, z, att_prev, hlens, w_decT, w_att, w_conv,
                       gvec, gvec_b, B, T, eprojs, dunits, adim, chans, filts, w_out, c_out, ldc_out);
  }
  RE2E_LAUNCH_CHECK();
  return RE2E_OK;
}

extern "C" int re2e_attloc_bwd(const float* pre, const float* enc, const float* z, const float* att_prev, const float* w_cur,
                               const int* hlens, const float* w_decT, const float* w_att, const float* w_conv, const float* gvec,
                               const float* dc, long ld_dc, const float* dw_in, int B, int T, int eprojs, int dunits, int adim,
                               int chans, int filts, float* d_pre, float* d_enc, float* d_att_prev, float* d_decproj,
                               float* partials, hipStream_t stream) {
  RE2E_CHECK_ARG(pre && enc && w_cur && hlens && w_decT && w_att && w_conv && gvec && dc && d_pre && d_enc && d_decproj && partials, "null arg");
  RE2E_CHECK_ARG(B > 0 && T > 0 && eprojs > 0 && eprojs % 4 == 0 && eprojs <= 2048 && dunits > 0 && adim > 0 && chans > 0 && filts >= 0, "bad shape");
  if (chans > 16 || adim > 512) { re2e_set_error("re2e_attloc_bwd: chans<=16 and adim<=512 supported"); return RE2E_EUNSUPPORTED; }
  bool small = chans <= 12 && adim <= 320;
  size_t lds = bwd_lds_floats(T, eprojs, dunits, adim, chans, filts, small ? 12 : 16) * sizeof(float);
  if (lds > 160 * 1024) { re2e_set_error("re2e_attloc_bwd: T=%d needs %zu bytes of LDS (>160 KiB)", T, lds); return RE2E_EUNSUPPORTED; }
  if (small) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attloc_bwd_kernel<12, 5>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((attloc_bwd_kernel<12, 5>), dim3(B), dim3(NT), lds, stream, pre, enc, z, att_prev, w_cur, hlens, w_decT, w_att,
                       w_conv, gvec, dc, ld_dc, dw_in, B, T, eprojs, dunits, adim, chans, filts, d_pre, d_enc, d_att_prev, d_decproj,
                       partials);
  } else {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attloc_bwd_kernel<16, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((attloc_bwd_kernel<16, 8>), dim3(B), dim3(NT), lds, stream, pre, enc, z, att_prev, w_cur, hlens, w_decT, w_att,
                       w_conv, gvec, dc, ld_dc, dw_in, B, T, eprojs, dunits, adim, chans, filts, d_pre, d_enc, d_att_prev, d_decproj,
                       partials);
  }
  RE2E_LAUNCH_CHECK();
  return RE2E_OK;
}

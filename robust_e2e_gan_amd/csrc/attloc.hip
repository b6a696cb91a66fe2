// K7: location-aware attention step (AttLoc.forward, model/e2e_attention.py:258-297) and its backward.
//
//   w = softmax_t( 2 * ( gvec . tanh( W_att conv(att_prev)[t] + pre[b,t] + W_dec z[b] ) + gb ) )
//   c = sum_t w[t] * enc[b,t]          (softmax over ALL T frames incl. padding, Appendix A.9)
//
// A single CU streams only ~30 GB/s from beyond its L2 (latency-bound), and one decoder step touches
// ~0.7 MB (forward) / ~2 MB (backward) per utterance, so the step is cut along the frame axis into
// workgroups of 32 frames (grid = frame-chunks x utterances: 224 workgroups at T'=200, B=32) and a
// small per-utterance kernel for the parts that need all frames:
//   forward : attloc_energy (chunk x b)  -> e[b,t], conv[b,t,:], dec_proj[b,:]
//             attloc_context (E/64 x b)  -> softmax over T (recomputed per workgroup, 200 values),
//                                           w[b,:], c[b, 64-column slice]
//   backward: attloc_bwd_frames (chunk x b): softmax + energy backward for its frames.  The softmax
//             normaliser sum_t w[t] dw[t] = w . dw_in + dc . c needs no cross-workgroup reduction
//             because the forward context c is saved.  Writes d_conv, accumulates d_pre, emits
//             per-chunk partial sums of the weight gradients.
//             attloc_bwd_conv (b): transposed location conv -> d att_prev, dW_conv, and the fixed-order
//             sum of the chunk partials -> d dec_proj, dgvec, dgb, dW_att.
//   after the loop: attloc_denc (chunk x b): d_enc[b,t,:] = sum_i w_i[b,t] dc_i[b,:] in one pass
//             instead of a 13 MB read-modify-write per step.
// Wavefront shuffles do the dot products / softmax reductions; pre/enc rows are read with
// row-contiguous (coalesced) loads, two frames in flight per wavefront.
#include "common.h"

namespace {
constexpr int TCH = 32;        // frames per workgroup
constexpr int NTH = 256;       // threads per workgroup (4 wavefronts)
constexpr int NWV = NTH / 64;
constexpr int FPW = TCH / NWV; // frames per wavefront
constexpr int CG = 5;          // conv channels per work item
constexpr int CMAX = 12;       // max conv channels held in registers
constexpr int AIMAX = 5;       // max ceil(adim/64)

__device__ __forceinline__ int cpad(int C) { return (C + 3) & ~3; }

// ---------------------------------------------------------------------------------------------
// forward, part 0: dec_proj[b][a] = sum_d W_dec[a][d] z[b][d].  grid (ceil(A/64), B); 64 outputs x 4 slices of d per
// workgroup, every thread keeps a whole slice of loads in flight (the product is latency-, not bandwidth-bound)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NTH) void attloc_decproj_kernel(const float* __restrict__ z, const float* __restrict__ w_decT, int D, int A,
                                                             float* __restrict__ dp_out) {
  __shared__ float zs[1024];
  __shared__ float part[4][64];
  const int b = blockIdx.y, a = blockIdx.x * 64 + (threadIdx.x & 63), dq = threadIdx.x >> 6;
  for (int d = threadIdx.x; d < D; d += NTH) zs[d] = z ? z[(long)b * D + d] : 0.f;
  __syncthreads();
  const int per = (D + 3) / 4, d0 = dq * per, d1 = min(D, d0 + per);
  float s0 = 0.f, s1 = 0.f;
  if (a < A) {
    int d = d0;
    for (; d + 16 <= d1; d += 16) {
      float w[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) w[u] = w_decT[(long)(d + u) * A + a];
#pragma unroll
      for (int u = 0; u < 16; u += 2) { s0 += w[u] * zs[d + u]; s1 += w[u + 1] * zs[d + u + 1]; }
    }
    for (; d < d1; ++d) s0 += w_decT[(long)d * A + a] * zs[d];
  }
  part[dq][threadIdx.x & 63] = s0 + s1;
  __syncthreads();
  if (dq == 0 && a < A) dp_out[(long)b * A + a] = (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
}

// ---------------------------------------------------------------------------------------------
// forward, part 1: energies for one chunk of frames
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NTH) void attloc_energy_kernel(const float* __restrict__ pre, const float* __restrict__ dp_in,
                                                            const float* __restrict__ att_prev, const int* __restrict__ hlens,
                                                            const float* __restrict__ w_att, const float* __restrict__ w_conv,
                                                            const float* __restrict__ gvec, const float* __restrict__ gvec_b, int B, int T,
                                                            int A, int C, int F, float* __restrict__ e_out, float* __restrict__ conv_out) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int CP = cpad(C), Kf = 2 * F + 1;
  float* ap = sm;                                 // [TCH + 2F]
  float* cv = ap + ((TCH + 2 * F + 3) & ~3);      // [TCH][CP]
  float* wcs = cv + TCH * CP;                     // [C][Kf]  filter taps
  float* cpart = wcs + ((C * Kf + 3) & ~3);       // [NWV][TCH][CP] per-wavefront partial conv tiles
  const int b = blockIdx.y, t0 = blockIdx.x * TCH;
  const int nt = min(TCH, T - t0);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hl = hlens[b];
  // pre rows of this wavefront's 8 frames: issued first, in flight while the location conv runs
  float pv[FPW][AIMAX];
#pragma unroll
  for (int k = 0; k < FPW; ++k) {
    const int l = wid + NWV * k;
    const float* pr = pre + ((long)b * T + t0 + (l < nt ? l : 0)) * A;
#pragma unroll
    for (int i = 0; i < AIMAX; ++i) { int a = lane + 64 * i; pv[k][i] = a < A ? pr[a] : 0.f; }
  }
  for (int i = tid; i < C * Kf; i += NTH) wcs[i] = w_conv[i];
  for (int i = tid; i < TCH + 2 * F; i += NTH) {
    int t = t0 + i - F;
    float v = 0.f;
    if (t >= 0 && t < T) v = att_prev ? att_prev[(long)b * T + t] : (t < hl ? 1.0f / (float)hl : 0.f);
    ap[i] = v;
  }
  __syncthreads();
  // location conv for this chunk: conv[t][c] = sum_k att_prev[t + k - F] * w_conv[c][k] -- a 32 (frames) x C x Kf product with a
  // Toeplitz left operand, done on the matrix core: v_mfma_f32_32x32x2 takes A[t][k] = ap[t + k] and B[k][c] = w_conv[c][k]
  // straight from the LDS copies (one ds_read each per lane and k-pair), the four wavefronts interleave the k-pairs and their
  // partial tiles are summed through LDS.  (The scalar tap loop this replaces was 17 of the kernel's 31 us: six dependent
  // LDS reads per tap, two of four wavefronts idle.)
  {
    const int tl = lane & 31, lh = lane >> 5;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int nks = (Kf + 1) / 2;
    for (int s2 = wid; s2 < nks; s2 += NWV) {
      const int k = 2 * s2 + lh;
      const float av = k < Kf ? ap[tl + k] : 0.f;
      const float bv = (k < Kf && tl < C) ? wcs[tl * Kf + k] : 0.f;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
    }
    if (tl < C) {                                  // lane holds column c = tl, rows (r&3) + 8 (r>>2) + 4 lh
      float* pp = cpart + (wid * TCH) * CP + tl;
#pragma unroll
      for (int r = 0; r < 16; ++r) pp[((r & 3) + 8 * (r >> 2) + 4 * lh) * CP] = acc[r];
    }
  }
  __syncthreads();
  for (int i = tid; i < TCH * CP; i += NTH) {
    const int c = i % CP;
    cv[i] = c < C ? (cpart[i] + cpart[TCH * CP + i]) + (cpart[2 * TCH * CP + i] + cpart[3 * TCH * CP + i]) : 0.f;
  }
  __syncthreads();
  for (int i = tid; i < nt * C; i += NTH) {
    int t = i / C, c = i % C;
    conv_out[((long)b * T + t0 + t) * C + c] = cv[t * CP + c];
  }
  // energies: a wavefront takes its 8 frames one after the other, lanes over the attention dimension
  float wa[AIMAX][CMAX], gv[AIMAX], dpv[AIMAX];
#pragma unroll
  for (int i = 0; i < AIMAX; ++i) {
    int a = lane + 64 * i;
    gv[i] = a < A ? gvec[a] : 0.f;
    dpv[i] = a < A ? dp_in[(long)b * A + a] : 0.f;
#pragma unroll
    for (int c = 0; c < CMAX; ++c) wa[i][c] = (a < A && c < C) ? w_att[a * C + c] : 0.f;
  }
  const float gb = gvec_b[0];
#pragma unroll
  for (int k = 0; k < FPW; ++k) {
    const int l = wid + NWV * k;
    float cvv[CMAX];
#pragma unroll
    for (int c = 0; c < CMAX; ++c) cvv[c] = (c < C && l < nt) ? cv[l * CP + c] : 0.f;
    float sv = 0.f;
#pragma unroll
    for (int i = 0; i < AIMAX; ++i) {
      int a = lane + 64 * i;
      if (a < A) {
        float x = pv[k][i] + dpv[i];
#pragma unroll
        for (int c = 0; c < CMAX; ++c) x += wa[i][c] * cvv[c];
        sv += gv[i] * tanhf_(x);
      }
    }
    sv = wave_sum(sv);
    if (lane == 0 && l < nt) e_out[(long)b * T + t0 + l] = sv + gb;
  }
}

// ---------------------------------------------------------------------------------------------
// forward, part 2: softmax over all T (recomputed per workgroup) + context for a 64-column slice
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NTH) void attloc_context_kernel(const float* __restrict__ e, const float* __restrict__ enc, int B, int T,
                                                             int E, float* __restrict__ w_out, float* __restrict__ c_out, long ldc_out) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* w = sm;                         // [T]
  float* red = w + ((T + 3) & ~3);       // [32]
  float* scr = red + 32;                 // [16][64]
  const int b = blockIdx.y, d0 = blockIdx.x * 64;
  const int tid = threadIdx.x;
  float m = -3.0e38f;
  for (int t = tid; t < T; t += NTH) { float v = 2.f * e[(long)b * T + t]; w[t] = v; m = fmaxf(m, v); }
  m = block_max(m, red);
  float sum = 0.f;
  for (int t = tid; t < T; t += NTH) { float v = __expf(w[t] - m); w[t] = v; sum += v; }
  sum = block_sum(sum, red);
  const float inv = 1.0f / sum;
  __syncthreads();
  for (int t = tid; t < T; t += NTH) {
    float v = w[t] * inv;
    w[t] = v;
    if (blockIdx.x == 0) w_out[(long)b * T + t] = v;
  }
  __syncthreads();
  // 16 float4 columns x 16 frame groups
  const int c4 = tid & 15, tg = tid >> 4;
  const int ncol = min(64, E - d0);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (c4 * 4 < ncol) {
    const f32x4* er = reinterpret_cast<const f32x4*>(enc + (long)b * T * E + d0) + c4;
    const int per = E / 4;
    int t = tg;
    for (; t + 48 < T; t += 64) {
      f32x4 v0 = er[(long)t * per], v1 = er[(long)(t + 16) * per], v2 = er[(long)(t + 32) * per], v3 = er[(long)(t + 48) * per];
      acc += v0 * w[t] + v1 * w[t + 16] + v2 * w[t + 32] + v3 * w[t + 48];
    }
    for (; t < T; t += 16) acc += er[(long)t * per] * w[t];
  }
  *reinterpret_cast<f32x4*>(scr + tg * 64 + c4 * 4) = acc;
  __syncthreads();
  if (tid < ncol) {
    float s = 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) s += scr[g * 64 + tid];
    c_out[(long)b * ldc_out + d0 + tid] = s;
  }
}

// ---------------------------------------------------------------------------------------------
// backward, part 1: per frame chunk.  ONLY what the recurrence needs leaves this kernel every step:
//   de[b,t]      (saved: attloc_dpre recomputes everything that depends on it after the loop)
//   d_conv[b,t,c] = sum_a du[t][a] W_att[a][c]      -> transposed location conv -> d att_prev
//   slab[b,chunk][a] = sum_{t in chunk} du[t][a]    -> d dec_proj
// with du[t][a] = de[t] gvec[a] (1 - tanh^2(x[t][a])).  The 16 MB read-modify-write of d_pre and the weight-gradient
// sums that the first version carried on this latency-critical path are done ONCE after the loop (attloc_dpre).
// Every global load of the kernel is issued before the first dependent use (pre rows of all 8 frames of a wavefront at
// kernel entry, enc rows four frames at a time); du goes through LDS so that both contractions read it without a
// single cross-lane reduction.
// ---------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ int du_stride(int A) { return A | 1; }      // odd: column reads hit 32 different banks

__global__ __launch_bounds__(NTH) void attloc_bwd_frames_kernel(
    const float* __restrict__ pre, const float* __restrict__ enc, const float* __restrict__ w_cur, const float* __restrict__ dw_in,
    const float* __restrict__ dc, long ld_dc, const float* __restrict__ cx, const float* __restrict__ conv_in,
    const float* __restrict__ dp_in, const float* __restrict__ w_att, const float* __restrict__ gvec, int B, int T, int E, int A, int C,
    float* __restrict__ de_out, float* __restrict__ d_conv, float* __restrict__ slabs) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int AS = du_stride(A);
  float* dcs = sm;                              // [E]
  float* de = dcs + ((E + 3) & ~3);             // [TCH]
  float* red = de + TCH;                        // [32]
  float* cvs = red + 32;                        // [TCH][CMAX]   conv_in rows of the chunk
  float* was = cvs + TCH * CMAX;                // [A][CMAX]     W_att
  float* du = was + A * CMAX;                   // [TCH][AS]
  float* pc = du + TCH * AS;                    // [8][TCH][CMAX] partial d_conv per a-range
  const int b = blockIdx.y, chunk = blockIdx.x, t0 = chunk * TCH;
  const int nt = min(TCH, T - t0);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  // pre rows of this wavefront's frames: in flight while the softmax terms are computed
  float pv[FPW][AIMAX];
#pragma unroll
  for (int k = 0; k < FPW; ++k) {
    const int l = wid + NWV * k;
    const float* pr = pre + ((long)b * T + t0 + (l < nt ? l : 0)) * A;
#pragma unroll
    for (int i = 0; i < AIMAX; ++i) { int a = lane + 64 * i; pv[k][i] = a < A ? pr[a] : 0.f; }
  }
  // softmax normaliser  s = sum_t w[t] dw[t] = w . dw_in + dc . c
  float part = 0.f;
  for (int d = tid; d < E; d += NTH) { float g = dc[(long)b * ld_dc + d]; dcs[d] = g; part += g * cx[(long)b * E + d]; }
  if (dw_in)
    for (int t = tid; t < T; t += NTH) part += w_cur[(long)b * T + t] * dw_in[(long)b * T + t];
  for (int i = tid; i < TCH * CMAX; i += NTH) {
    int l = i / CMAX, c = i % CMAX;
    cvs[i] = (l < nt && c < C) ? conv_in[((long)b * T + t0 + l) * C + c] : 0.f;
  }
  for (int i = tid; i < A * CMAX; i += NTH) { int a = i / CMAX, c = i % CMAX; was[i] = c < C ? w_att[a * C + c] : 0.f; }
  const float sdot = block_sum(part, red);     // (contains the __syncthreads that publish dcs / cvs / was)
  // de[t] = 2 w[t] (dw_in[t] + dc . enc[t] - s): a wavefront takes four of its frames at a time
  const int per = E / 4;
#pragma unroll
  for (int k0 = 0; k0 < FPW; k0 += 4) {
    float sacc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int d4 = lane; d4 < per; d4 += 64) {
      f32x4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int l = wid + NWV * (k0 + u);
        v[u] = reinterpret_cast<const f32x4*>(enc + ((long)b * T + t0 + (l < nt ? l : 0)) * E)[d4];
      }
      const f32x4 g = *reinterpret_cast<const f32x4*>(dcs + d4 * 4);
#pragma unroll
      for (int u = 0; u < 4; ++u) sacc[u] += v[u][0] * g[0] + v[u][1] * g[1] + v[u][2] * g[2] + v[u][3] * g[3];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int l = wid + NWV * (k0 + u);
      const float sv = wave_sum(sacc[u]);
      if (lane == 0 && l < nt) {
        const long i0 = (long)b * T + t0 + l;
        const float v = 2.f * w_cur[i0] * (sv + (dw_in ? dw_in[i0] : 0.f) - sdot);
        de[l] = v;
        de_out[i0] = v;
      }
    }
  }
  __syncthreads();
  // du[t][a] for the frames of this wavefront -> LDS
  {
    float gv[AIMAX], dpv[AIMAX];
#pragma unroll
    for (int i = 0; i < AIMAX; ++i) {
      int a = lane + 64 * i;
      gv[i] = a < A ? gvec[a] : 0.f;
      dpv[i] = a < A ? dp_in[(long)b * A + a] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < FPW; ++k) {
      const int l = wid + NWV * k;
      const float det = l < nt ? de[l] : 0.f;
      float cvv[CMAX];
#pragma unroll
      for (int c = 0; c < CMAX; ++c) cvv[c] = cvs[l * CMAX + c];
#pragma unroll
      for (int i = 0; i < AIMAX; ++i) {
        int a = lane + 64 * i;
        if (a < A) {
          float x = pv[k][i] + dpv[i];
#pragma unroll
          for (int c = 0; c < CMAX; ++c) x += was[a * CMAX + c] * cvv[c];
          float u = tanhf_(x);
          du[l * AS + a] = det * gv[i] * (1.f - u * u);
        }
      }
    }
  }
  __syncthreads();
  // d_conv[t][c] = sum_a du[t][a] W_att[a][c]: thread = (frame, one of 8 a-ranges), then a fixed-order sum of the 8 parts
  {
    const int l = tid & 31, p = tid >> 5;
    const int ap = (A + 7) / 8, a0 = p * ap, a1 = min(A, a0 + ap);
    float acc[CMAX];
#pragma unroll
    for (int c = 0; c < CMAX; ++c) acc[c] = 0.f;
    for (int a = a0; a < a1; ++a) {
      const float d = du[l * AS + a];
#pragma unroll
      for (int c = 0; c < CMAX; ++c) acc[c] += d * was[a * CMAX + c];
    }
#pragma unroll
    for (int c = 0; c < CMAX; ++c) pc[(p * TCH + l) * CMAX + c] = acc[c];
  }
  // slab[a] = sum_t du[t][a]
  float* slab = slabs + ((long)b * gridDim.x + chunk) * A;
  for (int a = tid; a < A; a += NTH) {
    float s = 0.f;
    for (int l = 0; l < nt; ++l) s += du[l * AS + a];
    slab[a] = s;
  }
  __syncthreads();
  for (int i = tid; i < nt * C; i += NTH) {
    const int l = i / C, c = i % C;
    float s = 0.f;
#pragma unroll
    for (int p = 0; p < 8; ++p) s += pc[(p * TCH + l) * CMAX + c];
    d_conv[((long)b * T + t0 + l) * C + c] = s;
  }
}

// ---------------------------------------------------------------------------------------------
// after the decoder loop: everything of the energy backward that no later step needs, for ALL steps at once.
//   d_pre[b,t,a]  = sum_i du_i[t][a]
//   slab[b,chunk] = [ dgvec(A) = sum de_i[t] tanh(x_i) | dW_att(A*C) = sum du_i[t][a] conv_i[t][c] | dgb = sum de_i[t] ]
// x_i is recomputed from pre, the saved dec_proj_i and conv_i (10 FMAs + one tanh per element and step) instead of
// streaming 16 MB through d_pre on every one of the L1 steps.
// ---------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ int dpre_slab_floats(int A, int C) { return A * (1 + C) + 1; }

constexpr int TCD = 16;          // frames per workgroup; thread = one attention unit a, all TCD frames in registers
__global__ __launch_bounds__(64 * AIMAX) void attloc_dpre_kernel(const float* __restrict__ pre, const float* __restrict__ conv_all,
                                                                 const float* __restrict__ dp_all, const float* __restrict__ de_all,
                                                                 const float* __restrict__ w_att, const float* __restrict__ gvec, int L1, int B,
                                                                 int T, int A, int C, float* __restrict__ d_pre, float* __restrict__ slabs) {
  __shared__ float des[TCD];
  __shared__ float cvs[TCD * CMAX];
  const int b = blockIdx.y, chunk = blockIdx.x, t0 = chunk * TCD;
  const int nt = min(TCD, T - t0);
  const int tid = threadIdx.x, nth = blockDim.x;
  const int a = tid;
  const bool on = a < A;
  float pv[TCD], acc[TCD], wa[CMAX], dwa[CMAX];
#pragma unroll
  for (int l = 0; l < TCD; ++l) {
    pv[l] = (on && l < nt) ? pre[((long)b * T + t0 + l) * A + a] : 0.f;
    acc[l] = 0.f;
  }
#pragma unroll
  for (int c = 0; c < CMAX; ++c) { wa[c] = (on && c < C) ? w_att[a * C + c] : 0.f; dwa[c] = 0.f; }
  const float gv = on ? gvec[a] : 0.f;
  float dgv = 0.f, dgb = 0.f;
  for (int s = 0; s < L1; ++s) {
    const float dpv = on ? dp_all[((long)s * B + b) * A + a] : 0.f;
    __syncthreads();
    if (tid < TCD) des[tid] = tid < nt ? de_all[((long)s * B + b) * T + t0 + tid] : 0.f;
    for (int i = tid; i < TCD * CMAX; i += nth) {
      int l = i / CMAX, c = i % CMAX;
      cvs[i] = (l < nt && c < C) ? conv_all[(((long)s * B + b) * T + t0 + l) * C + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int l = 0; l < TCD; ++l) {
      const float det = des[l];                   // 0 beyond the last frame
      float x = pv[l] + dpv;
#pragma unroll
      for (int c = 0; c < CMAX; ++c) x += wa[c] * cvs[l * CMAX + c];
      const float u = tanhf_(x);
      const float d = det * gv * (1.f - u * u);
      acc[l] += d;
      dgv += det * u;
      dgb += det;
#pragma unroll
      for (int c = 0; c < CMAX; ++c) dwa[c] += d * cvs[l * CMAX + c];
      __builtin_amdgcn_sched_barrier(0);          // one frame at a time: hoisting all 16 x 12 LDS reads costs 192 registers
    }
  }
  float* slab = slabs + ((long)b * gridDim.x + chunk) * dpre_slab_floats(A, C);
  if (on) {
#pragma unroll
    for (int l = 0; l < TCD; ++l)
      if (l < nt) d_pre[((long)b * T + t0 + l) * A + a] = acc[l];
    slab[a] = dgv;
    for (int c = 0; c < C; ++c) slab[A + a * C + c] = dwa[c];
  }
  if (tid == 0) slab[A * (1 + C)] = dgb;
}

// partials[b][gvec | gvec_b | w_att] += fixed-order sum of the chunk slabs of attloc_dpre
__global__ __launch_bounds__(NTH) void attloc_dpre_reduce_kernel(const float* __restrict__ slabs, int nchunk, int A, int C, int npart,
                                                                 float* partials) {
  const int b = blockIdx.x, SF = dpre_slab_floats(A, C);
  const float* sl = slabs + (long)b * nchunk * SF;
  float* part = partials + (long)b * npart;
  for (int i = threadIdx.x; i < SF; i += NTH) {
    float s = 0.f;
    for (int k = 0; k < nchunk; ++k) s += sl[(long)k * SF + i];
    if (i < A) part[i] += s;                              // gvec
    else if (i < A * (1 + C)) part[A + 1 + (i - A)] += s; // w_att
    else part[A] += s;                                    // gvec_b
  }
}

// ---------------------------------------------------------------------------------------------
// backward, part 2 (per utterance): transposed conv, dW_conv, reduction of the chunk slabs
// partials layout per utterance: [gvec(A) | gvec_b(1) | w_att(A*C) | w_conv(C*Kf)]
// ---------------------------------------------------------------------------------------------
constexpr int NTC = 512;      // threads of the per-utterance backward kernel
// grid (B, 3): blockIdx.y selects one of three independent jobs so that three CUs share an utterance:
//   0: fixed-order sum of the chunk slabs -> d dec_proj
//   1: transposed location conv           -> d att_prev
//   2: filter gradient                    -> dW_conv partials
__global__ __launch_bounds__(NTC) void attloc_bwd_conv_kernel(const float* __restrict__ att_prev, const int* __restrict__ hlens,
                                                              const float* __restrict__ w_conv, const float* __restrict__ d_conv,
                                                              const float* __restrict__ slabs, int nchunk, int B, int T, int A, int C,
                                                              int F, float* __restrict__ d_att_prev, float* __restrict__ d_decproj,
                                                              float* partials) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int Kf = 2 * F + 1, TP = T + 2 * F + 1;
  float* ap = sm;                                 // [T + 2F]
  float* dct = ap + ((T + 2 * F + 3) & ~3);       // [C][TP]   d_conv TRANSPOSED (frame index fastest), zero padded by F each side
  float* wcs = dct + C * TP;                      // [C][Kf]
  float* scr = wcs + C * Kf;                      // [C][T]
  const int b = blockIdx.x, job = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int NWC = NTC / 64;
  const int P_WCONV = A + 1 + A * C;
  float* part = partials + (long)b * (A + 1 + A * C + C * Kf);
  if (job == 0) {
    const float* sl = slabs + (long)b * nchunk * A;
    for (int i = tid; i < A; i += NTC) {
      float s = 0.f;
      for (int k = 0; k < nchunk; ++k) s += sl[(long)k * A + i];
      d_decproj[(long)b * A + i] = s;
    }
    return;
  }
  if (job == 1 && !d_att_prev) return;
  const int hl = hlens[b];
  if (job == 1) for (int i = tid; i < C * Kf; i += NTC) wcs[i] = w_conv[i];
  if (job == 2) {
    for (int i = tid; i < T + 2 * F; i += NTC) {
      int t = i - F;
      float v = 0.f;
      if (t >= 0 && t < T) v = att_prev ? att_prev[(long)b * T + t] : (t < hl ? 1.0f / (float)hl : 0.f);
      ap[i] = v;
    }
  }
  for (int i = tid; i < C * TP; i += NTC) dct[i] = 0.f;
  __syncthreads();
  for (int i = tid; i < T * C; i += NTC) {        // coalesced read of d_conv[b] (T, C), transposed scatter into LDS
    int t = i / C, c = i % C;
    dct[c * TP + t + F] = d_conv[(long)b * T * C + i];
  }
  __syncthreads();
  if (job == 1) {
    // d_att_prev[t'] = sum_{c,k} w_conv[c][k] * d_conv[t'-k+F][c] : a wavefront owns one channel and 64 frames
    const int tchunks = (T + 63) / 64;
    for (int item = wid; item < C * tchunks; item += NWC) {
      const int c = item % C, tc = item / C;
      const int t = tc * 64 + lane;
      const int tt = t < T ? t : T - 1;
      const float* q = dct + c * TP + tt + 2 * F;          // q[-k] = d_conv[tt - k + F][c]
      const float* wk = wcs + c * Kf;
      float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
      int k = 0;
      for (; k + 8 <= Kf; k += 8) {
        a0 += wk[k] * q[-k] + wk[k + 4] * q[-k - 4]; a1 += wk[k + 1] * q[-k - 1] + wk[k + 5] * q[-k - 5];
        a2 += wk[k + 2] * q[-k - 2] + wk[k + 6] * q[-k - 6]; a3 += wk[k + 3] * q[-k - 3] + wk[k + 7] * q[-k - 7];
      }
      for (; k < Kf; ++k) a0 += wk[k] * q[-k];
      if (t < T) scr[c * T + t] = (a0 + a1) + (a2 + a3);
    }
    __syncthreads();
    for (int t = tid; t < T; t += NTC) {
      float s = 0.f;
      for (int c = 0; c < C; ++c) s += scr[c * T + t];
      d_att_prev[(long)b * T + t] = s;
    }
    return;
  }
  // job 2: dw_conv[c][k] = sum_t d_conv[t][c] * att_prev[t+k-F] : thread = (tap k, channel group), frames serial (4 in flight)
  {
    const int ncg = (C + CG - 1) / CG;
    for (int item = tid; item < ncg * Kf; item += NTC) {
      const int k = item % Kf, cg = item / Kf, c0 = cg * CG;
      float acc[CG];
#pragma unroll
      for (int cc = 0; cc < CG; ++cc) acc[cc] = 0.f;
      int t = 0;
      for (; t + 4 <= T; t += 4) {
        const float x0 = ap[t + k], x1 = ap[t + k + 1], x2 = ap[t + k + 2], x3 = ap[t + k + 3];
#pragma unroll
        for (int cc = 0; cc < CG; ++cc)
          if (c0 + cc < C) {
            const float* q = dct + (c0 + cc) * TP + t + F;
            acc[cc] += q[0] * x0 + q[1] * x1 + q[2] * x2 + q[3] * x3;
          }
      }
      for (; t < T; ++t) {
        const float x = ap[t + k];
#pragma unroll
        for (int cc = 0; cc < CG; ++cc)
          if (c0 + cc < C) acc[cc] += dct[(c0 + cc) * TP + t + F] * x;
      }
#pragma unroll
      for (int cc = 0; cc < CG; ++cc)
        if (c0 + cc < C) part[P_WCONV + (c0 + cc) * Kf + k] += acc[cc];
    }
  }
}

// d_enc[b,t,:] = beta*d_enc + sum_i w[i,b,t] * dc[i,b,:]   (after the decoder loop)
__global__ __launch_bounds__(NTH) void attloc_denc_kernel(const float* __restrict__ w_all, const float* __restrict__ dc_all, int L1, int B,
                                                          int T, int E, float* d_enc, float beta) {
  extern __shared__ __attribute__((aligned(16))) float sm[];     // w chunk [L1][TCH]
  const int b = blockIdx.y, t0 = blockIdx.x * TCH;
  const int nt = min(TCH, T - t0);
  const int tid = threadIdx.x;
  for (int i = tid; i < L1 * TCH; i += NTH) {
    int s = i / TCH, l = i % TCH;
    sm[i] = l < nt ? w_all[((long)s * B + b) * T + t0 + l] : 0.f;
  }
  __syncthreads();
  for (int d = tid; d < E; d += NTH) {
    float acc[TCH];
#pragma unroll
    for (int l = 0; l < TCH; ++l) acc[l] = 0.f;
    for (int s = 0; s < L1; ++s) {
      float g = dc_all[((long)s * B + b) * E + d];
#pragma unroll
      for (int l = 0; l < TCH; ++l) acc[l] += sm[s * TCH + l] * g;
    }
#pragma unroll
    for (int l = 0; l < TCH; ++l)
      if (l < nt) {
        float* q = d_enc + ((long)b * T + t0 + l) * E + d;
        *q = (beta != 0.f ? beta * (*q) : 0.f) + acc[l];
      }
  }
}

inline int nchunks(int T) { return (T + TCH - 1) / TCH; }
}  // namespace

extern "C" size_t re2e_attloc_partial_floats(int adim, int chans, int filts) {
  return (size_t)adim + 1 + (size_t)adim * chans + (size_t)chans * (2 * filts + 1);
}

extern "C" size_t re2e_attloc_workspace_bytes(int B, int T, int adim, int chans) {
  // d_conv [B][T][C] + per-step chunk slabs [B][nchunk][A] + attloc_dpre chunk slabs [B][ceil(T/16)][A*(1+C)+1]
  return ((size_t)B * T * chans + (size_t)B * nchunks(T) * adim + (size_t)B * ((T + TCD - 1) / TCD) * ((size_t)adim * (1 + chans) + 1)) * sizeof(float);
}

static int check_dims(const char* fn, int B, int T, int E, int D, int A, int C, int F) {
  if (!(B > 0 && T > 0 && E > 0 && E % 4 == 0 && D > 0 && A > 0 && C > 0 && F >= 0)) { re2e_set_error("%s: bad shape", fn); return RE2E_EINVAL; }
  if (C > CMAX || A > 64 * AIMAX) {
    re2e_set_error("%s: aconv_chans <= %d and adim <= %d supported (got %d, %d)", fn, CMAX, 64 * AIMAX, C, A);
    return RE2E_EUNSUPPORTED;
  }
  return RE2E_OK;
}

extern "C" int re2e_attloc_fwd(const float* pre, const float* enc, const float* z, const float* att_prev, const int* hlens,
                               const float* w_decT, const float* w_att, const float* w_conv, const float* gvec, const float* gvec_b,
                               int B, int T, int eprojs, int dunits, int adim, int chans, int filts, float* w_out, float* c_out,
                               long ldc_out, float* conv_out, float* dp_out, float* e_scratch, hipStream_t stream) {
  RE2E_CHECK_ARG(pre && enc && hlens && w_decT && w_att && w_conv && gvec && gvec_b && w_out && c_out && conv_out && dp_out && e_scratch,
                 "null arg");
  int rc = check_dims("re2e_attloc_fwd", B, T, eprojs, dunits, adim, chans, filts);
  if (rc) return rc;
  const int CP = (chans + 3) & ~3;
  RE2E_CHECK_ARG(dunits <= 1024, "dunits > 1024 not supported");
  hipLaunchKernelGGL(attloc_decproj_kernel, dim3((adim + 63) / 64, B), dim3(NTH), 0, stream, z, w_decT, dunits, adim, dp_out);
  size_t lds1 = (size_t)(((TCH + 2 * filts + 3) & ~3) + TCH * CP + ((chans * (2 * filts + 1) + 3) & ~3) + NWV * TCH * CP + 16) * sizeof(float);
  if (lds1 > 64 * 1024) { re2e_set_error("re2e_attloc_fwd: aconv_filts=%d needs %zu bytes of LDS (>64 KiB)", filts, lds1); return RE2E_EUNSUPPORTED; }
  hipLaunchKernelGGL(attloc_energy_kernel, dim3(nchunks(T), B), dim3(NTH), lds1, stream, pre, (const float*)dp_out, att_prev, hlens, w_att, w_conv,
                     gvec, gvec_b, B, T, adim, chans, filts, e_scratch, conv_out);
  size_t lds2 = (size_t)(((T + 3) & ~3) + 32 + 16 * 64 + 16) * sizeof(float);
  if (lds2 > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attloc_context_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
  hipLaunchKernelGGL(attloc_context_kernel, dim3((eprojs + 63) / 64, B), dim3(NTH), lds2, stream, (const float*)e_scratch, enc, B, T, eprojs, w_out,
                     c_out, ldc_out);
  RE2E_LAUNCH_CHECK();
  return RE2E_OK;
}

extern "C" int re2e_attloc_bwd(const float* pre, const float* enc, const float* att_prev, const float* w_cur, const int* hlens,
                               const float* w_att, const float* w_conv, const float* gvec, const float* conv_in, const float* dp_in,
                               const float* cx_in, const float* dc, long ld_dc, const float* dw_in, int B, int T, int eprojs, int adim,
                               int chans, int filts, float* de_out, float* d_att_prev, float* d_decproj, float* partials, void* workspace,
                               size_t workspace_bytes, hipStream_t stream) {
  RE2E_CHECK_ARG(pre && enc && w_cur && hlens && w_att && w_conv && gvec && conv_in && dp_in && cx_in && dc && de_out && d_decproj && partials &&
                     workspace, "null arg");
  int rc = check_dims("re2e_attloc_bwd", B, T, eprojs, 1, adim, chans, filts);
  if (rc) return rc;
  RE2E_CHECK_ARG(workspace_bytes >= re2e_attloc_workspace_bytes(B, T, adim, chans), "workspace too small");
  const int Kf = 2 * filts + 1, nch = nchunks(T);
  float* d_conv = (float*)workspace;
  float* slabs = d_conv + (size_t)B * T * chans;
  size_t lds1 = (size_t)(((eprojs + 3) & ~3) + TCH + 32 + TCH * CMAX + adim * CMAX + TCH * du_stride(adim) + 8 * TCH * CMAX + 16) * sizeof(float);
  if (lds1 > 160 * 1024) { re2e_set_error("re2e_attloc_bwd: eprojs=%d adim=%d need %zu bytes of LDS (>160 KiB)", eprojs, adim, lds1); return RE2E_EUNSUPPORTED; }
  if (lds1 > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attloc_bwd_frames_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1);
  hipLaunchKernelGGL(attloc_bwd_frames_kernel, dim3(nch, B), dim3(NTH), lds1, stream, pre, enc, w_cur, dw_in, dc, ld_dc, cx_in, conv_in, dp_in, w_att,
                     gvec, B, T, eprojs, adim, chans, de_out, d_conv, slabs);
  size_t lds2 = (size_t)(((T + 2 * filts + 3) & ~3) + (size_t)chans * (T + 2 * filts + 1) + (size_t)chans * Kf + (size_t)chans * T + 16) *
                sizeof(float);
  if (lds2 > 160 * 1024) { re2e_set_error("re2e_attloc_bwd: T=%d needs %zu bytes of LDS (>160 KiB)", T, lds2); return RE2E_EUNSUPPORTED; }
  if (lds2 > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attloc_bwd_conv_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
  hipLaunchKernelGGL(attloc_bwd_conv_kernel, dim3(B, 3), dim3(NTC), lds2, stream, att_prev, hlens, w_conv, (const float*)d_conv, (const float*)slabs, nch,
                     B, T, adim, chans, filts, d_att_prev, d_decproj, partials);
  RE2E_LAUNCH_CHECK();
  return RE2E_OK;
}

extern "C" int re2e_attloc_dpre(const float* pre, const float* conv_all, const float* dp_all, const float* de_all, const float* w_att,
                                const float* gvec, int L1, int B, int T, int adim, int chans, int filts, float* d_pre, float* partials,
                                void* workspace, size_t workspace_bytes, hipStream_t stream) {
  RE2E_CHECK_ARG(pre && conv_all && dp_all && de_all && w_att && gvec && d_pre && partials && workspace && L1 > 0, "bad args");
  int rc = check_dims("re2e_attloc_dpre", B, T, 4, 1, adim, chans, filts);
  if (rc) return rc;
  RE2E_CHECK_ARG(workspace_bytes >= re2e_attloc_workspace_bytes(B, T, adim, chans), "workspace too small");
  const int nch = nchunks(T), ncd = (T + TCD - 1) / TCD;
  float* slabs = (float*)workspace + (size_t)B * T * chans + (size_t)B * nch * adim;
  hipLaunchKernelGGL(attloc_dpre_kernel, dim3(ncd, B), dim3(64 * ((adim + 63) / 64)), 0, stream, pre, conv_all, dp_all, de_all, w_att, gvec, L1, B, T,
                     adim, chans, d_pre, slabs);
  hipLaunchKernelGGL(attloc_dpre_reduce_kernel, dim3(B), dim3(NTH), 0, stream, (const float*)slabs, ncd, adim, chans,
                     (int)re2e_attloc_partial_floats(adim, chans, filts), partials);
  RE2E_LAUNCH_CHECK();
  return RE2E_OK;
}

extern "C" int re2e_attloc_denc(const float* w_all, const float* dc_all, int L1, int B, int T, int eprojs, float* d_enc, float beta,
                                hipStream_t stream) {
  RE2E_CHECK_ARG(w_all && dc_all && d_enc && L1 > 0 && B > 0 && T > 0 && eprojs > 0, "bad args");
  size_t lds = (size_t)L1 * TCH * sizeof(float);
  RE2E_CHECK_ARG(lds <= 64 * 1024, "too many decoder steps for the LDS weight tile");
  hipLaunchKernelGGL(attloc_denc_kernel, dim3(nchunks(T), B), dim3(NTH), lds, stream, w_all, dc_all, L1, B, T, eprojs, d_enc, beta);
  RE2E_LAUNCH_CHECK();
  return RE2E_OK;
}

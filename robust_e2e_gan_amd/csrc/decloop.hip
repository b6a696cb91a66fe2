// K7 + K8 (round 4): the decoder's teacher-forced loop as ONE persistent launch.
//
// Replaces, for the whole loop of model/e2e_decoder.py:113-152 (AttLoc.forward model/e2e_attention.py:259-299 + LSTMCell per output
// token), the launch-per-step sequence decproj / energy / context / gates+cell of attloc.hip + lstm.hip: 4 dependent launches of
// 5-17 us each per token (41 tokens at config 4, 151 at config 5), every one of which reloaded its operands -- pre (B,T,A), the
// encoder states (B,T,E), 3 MB of decoder weights -- from beyond the CU.  The utterances of a batch are independent except for
// the shared weights, so the loop is cut into two kinds of resident workgroups that keep THEIR operands for all L1 steps and
// hand only vectors to each other (flagged hand-off "R1" of MI355X_MICROARCH.md, as in lstm.hip: sc1 stores, drain, one flag word
// per workgroup on its own 128-byte line, sc1 loads behind the poll):
//
//   gate workgroup x (cdiv(D, 8) of them): 8 decoder units = 32 gate rows of [W_ih[:, Dd:] | W_hh] in registers as MFMA operands
//       (v_mfma_f32_32x32x2_f32, utterances as M, K split over the 4 waves), the cell state of its (utterance, unit) cells in a register.
//       step i:  acc  = z_i W_hh[rows]^T                         when z_i is complete        (off the critical path)
//                acc += cx_i W_ctx[rows]^T                       when every context piece is there
//                cell -> z_{i+1}, c_{i+1}, activated gates;  publish z_{i+1}[:, units]
//   attention workgroup (b, ch, fc): utterance b, slice ch (64 columns of the attention dimension and the matching share of the projection
//       columns), frame chunk fc (<= 256 frames).  Resident: pre[b, frames, slice] in registers IN THE ACCUMULATOR LAYOUT of v_mfma_f32_16x16x4 with
//       the frames as N (lane = frame of a 16-frame tile and four columns of every 16-column block), enc[b, frames, slice] in LDS, mlp_dec[slice, :]
//       and the W_att / gvec operands in registers, the conv taps in LDS.
//       step i:  (before z_i arrives) location conv of w_{i-1} for its frames on the matrix core, TRANSPOSED (taps as the A operand, a Toeplitz B
//                operand straight from LDS): its accumulators are the B operand of u^T = pre^T + W_att conv^T as they are; u stays in registers
//                dp[slice] = W_dec[slice, :] z_i[b]              when z_i is complete
//                e_part[t] = sum_{a in slice} gvec[a] tanh(u[t][a] + dp[a]): 16 columns per lane and tile, a frame's four lanes summed by two
//                shuffles -> published; the slice axis is what is exchanged, so no workgroup needs another's dp
//                e[t] = sum_ch e_part + gb, softmax over ALL T frames (every workgroup of b redundantly: no w exchange)
//                cx_part[slice] = sum_{t in chunk} w[t] enc[t][slice] -> published
// One wave per SIMD hides no LDS or memory round trip, so every operand batch is fetched explicitly ahead of its use (first build, with the
// compiler's load -> wait -> use per element: 24 us per token; this form: 18).
// Per token the critical path is three hand-offs (z, e_part, cx_part: ~1 us each, measured) and the phases between them instead of four launches.
// Everything must be co-resident (one workgroup per CU fits: checked against the CU count), spins are bounded, an abort poisons
// z with NaN and is counted in re2e_lstm_abort_count.  Scheduled sampling (a data-dependent token per step) and shapes outside
// re2e_dec_loop_workspace_bytes' limits stay on the launch-per-step path (ops.DecoderLoopFn), which is also the parity twin in
// tests/test_kernels_gpu.py.
#include "common.h"

namespace {
constexpr int NT = 256;             // threads per workgroup
constexpr int KGZ = 10;             // k-groups of 8 per wave of the recurrent half: D <= 320
constexpr int KGC = 16;             // ... of the context half: E <= 512
constexpr int DQ = 80;              // decoder units per K quarter of the projection: D <= 320
constexpr int TPW = 4;               // 16-frame tiles per wave: <= 256 frames per attention workgroup
constexpr int CK = 13;              // conv taps per lane fetched together
constexpr int CP = 12;              // conv channels padded
constexpr int NSMAX = 8;            // 64-column slices: A, E <= 512
constexpr int NKSMAX = 52;          // location-conv taps / 4: 2 Fh + 1 <= 208
constexpr unsigned kSpinLimit = 1u << 24;

__device__ unsigned g_dec_aborts = 0;

struct DecFwdArgs {
  const float *pre, *enc;
  const int* hlens;
  const float *w_decT, *w_att, *w_conv, *gvec, *gvec_b, *w_ctx;
  long ldw;
  const float* w_hh;
  float *gates, *z, *c, *w, *cx, *conv, *dpj;
  int L1, B, T, E, D, A, C, Fh;
  int NG, NS, NFC, TC, ES, esw;
  unsigned *err, *zflag, *eflag, *cflag;
  float *ebuf, *cxp;
  unsigned long long* stamps;     // RE2E_EXPERIMENTS: [workgroup][16 steps from kStamp0][16 phases] of the 100 MHz chip-wide clock
};

typedef __attribute__((address_space(1))) unsigned gu32;
typedef float f32x2 __attribute__((ext_vector_type(2)));

#ifdef RE2E_EXPERIMENTS
constexpr int kStamp0 = 8;
#define DEC_STAMP(ph)                                                                                                   \
  do {                                                                                                                  \
    if (a.stamps && threadIdx.x == 0 && i >= kStamp0 && i < kStamp0 + 16)                                               \
      a.stamps[((long)blockIdx.x * 16 + (i - kStamp0)) * 16 + (ph)] = __builtin_amdgcn_s_memrealtime();                 \
  } while (0)
#else
#define DEC_STAMP(ph)
#endif

__device__ __forceinline__ void store4_sc1(float* p, float v) {
  asm volatile("global_store_dword %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}

// tanh as 1 - 2 / (1 + exp(2|x|)) with v_exp_f32 / v_rcp_f32 (1 ulp each): 6 instructions instead of the 16 of common.h's IEEE-division
// form -- 12 800 of them per workgroup and token are the longest phase of the critical path.  Absolute error ~1e-7, as that form's.
__device__ __forceinline__ float tanh_fast(float x) {
  const float e = __builtin_amdgcn_exp2f(fabsf(x) * 2.885390081777927f);      // exp(2|x|); inf for |x| > 44 -> rcp 0 -> 1
  const float t = 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + e);
  return copysignf(t, x);
}

// Every wave polls the n flags itself (the wave that polled is the wave that loads: no workgroup barrier between flag and data).
__device__ __forceinline__ bool wait_flags(const unsigned* flags, int n, unsigned want, unsigned* err, int lane) {
  for (int base = 0; base < n; base += 64) {
    const int idx = base + lane;
    const gu32* fl = (const gu32*)(flags + (long)(idx < n ? idx : 0) * 32);
    for (unsigned spins = 0;; ++spins) {
      const bool good = idx >= n || __hip_atomic_load(fl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want;
      if (__all(good)) break;
      if ((spins & 1023u) == 1023u) {                   // a peer gave up: follow it at once instead of spinning to the bound
        if (__hip_atomic_load((const gu32*)err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return false;
      }
      if (spins > kSpinLimit) {
        if (lane == 0 && atomicExch(err, 1u) == 0u) atomicAdd(&g_dec_aborts, 1u);
        return false;
      }
      __builtin_amdgcn_s_sleep(1);
    }
  }
  return true;
}

// ---------------------------------------------------------------------------------------------------------------------------------
__device__ void gate_role(const DecFwdArgs& a, const int x, float* sm) {
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6), lr = lane & 31, lh = lane >> 5;
  const int B = a.B, D = a.D, E = a.E, L1 = a.L1;
  const int u0 = x * 8;
  const int g = lr >> 3, u = u0 + (lr & 7);
  const bool cok = u < D, rok = lr < B;
  const long wrow = (long)g * D + (cok ? u : 0);
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  const int nqz = (D + 7) / 8, nqc = (E + 7) / 8;
  f32x4 wz[KGZ], wc[KGC];
#pragma unroll
  for (int t = 0; t < KGZ; ++t) {
    const int q = wid + 4 * t, k = 8 * q + 4 * lh;
    wz[t] = (q < nqz && k < D && cok) ? *reinterpret_cast<const f32x4*>(a.w_hh + wrow * D + k) : zero;
  }
#pragma unroll
  for (int t = 0; t < KGC; ++t) {
    const int q = wid + 4 * t, k = 8 * q + 4 * lh;
    wc[t] = (q < nqc && k < E && cok) ? *reinterpret_cast<const f32x4*>(a.w_ctx + wrow * a.ldw + k) : zero;
  }
  float (*red)[32][33] = reinterpret_cast<float (*)[32][33]>(sm);
  const int bm = tid >> 3, jj = tid & 7, uu = u0 + jj;
  const bool cell_ok = bm < B && uu < D;
  float cprev = cell_ok ? a.c[(long)bm * D + uu] : 0.f;
  float pg[4] = {0.f, 0.f, 0.f, 0.f};
  if (cell_ok) {
#pragma unroll
    for (int q = 0; q < 4; ++q) pg[q] = a.gates[(long)bm * 4 * D + (long)q * D + uu];
  }
  const __amdgpu_buffer_rsrc_t z_rs = __builtin_amdgcn_make_buffer_rsrc(a.z, 0, (int)((long)(L1 + 1) * B * D * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t c_rs = __builtin_amdgcn_make_buffer_rsrc(a.cxp, 0, (int)(2L * B * a.NFC * a.ES * 4), 0x00020000);
  const int NA = B * a.NS * a.NFC;
  bool aborted = false;
  for (int i = 0; i < L1; ++i) {
    f32x16 acc;
    DEC_STAMP(0);
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if (i > 0) {                                     // recurrent half: as soon as z_i is complete (z_0 = 0)
      if (!aborted && !wait_flags(a.zflag, a.NG, (unsigned)i, a.err, lane)) aborted = true;
      DEC_STAMP(1);
      f32x4 av[KGZ];
#pragma unroll
      for (int t = 0; t < KGZ; ++t) {
        const int q = wid + 4 * t, k = 8 * q + 4 * lh;
        av[t] = (q < nqz && k < D && rok) ? __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(z_rs, (unsigned)((((long)i * B + lr) * D + k) * 4), 0, 16)) : zero;
      }
#pragma unroll
      for (int t = 0; t < KGZ; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t][j], wz[t][j], acc, 0, 0, 0);
    }
    DEC_STAMP(2);
    // context half: the pieces of every attention workgroup
    if (!aborted && !wait_flags(a.cflag, NA, (unsigned)(i + 1), a.err, lane)) aborted = true;
    DEC_STAMP(3);
    {
      const int par = i & 1;
      f32x4 av[KGC];
#pragma unroll
      for (int t = 0; t < KGC; ++t) av[t] = zero;
      for (int fc = 0; fc < a.NFC; ++fc) {            // all of a chunk's loads in flight together (out-of-range ones return 0)
        f32x4 ld[KGC];
#pragma unroll
        for (int t = 0; t < KGC; ++t) {
          const int q = wid + 4 * t, k = 8 * q + 4 * lh;
          const unsigned off = (q < nqc && k < E && rok) ? (unsigned)(((((long)par * B + lr) * a.NFC + fc) * a.ES + k) * 4) : 0xFFFFFFF0u;
          ld[t] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(c_rs, off, 0, 16));
        }
#pragma unroll
        for (int t = 0; t < KGC; ++t) av[t] += ld[t];
      }
      if (x == 0) {                                   // saved for the backward
#pragma unroll
        for (int t = 0; t < KGC; ++t) {
          const int q = wid + 4 * t, k = 8 * q + 4 * lh;
          if (q < nqc && k < E && rok) *reinterpret_cast<f32x4*>(a.cx + ((long)i * B + lr) * E + k) = av[t];
        }
      }
#pragma unroll
      for (int t = 0; t < KGC; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t][j], wc[t][j], acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wid][(r & 3) + 8 * (r >> 2) + 4 * lh][lr] = acc[r];
    __syncthreads();
    DEC_STAMP(4);
    if (cell_ok) {
      float v[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = ((red[0][bm][q * 8 + jj] + red[1][bm][q * 8 + jj]) + (red[2][bm][q * 8 + jj] + red[3][bm][q * 8 + jj])) + pg[q];
      const float gi = sigmoidf_(v[0]), gf = sigmoidf_(v[1]), gg = tanhf_(v[2]), go = sigmoidf_(v[3]);
      const float cn = gf * cprev + gi * gg;
      float h = go * tanhf_(cn);
      if (aborted) h = __uint_as_float(0x7fc00000u);            // a peer never published: make the failure visible downstream
      float* gp = a.gates + ((long)i * B + bm) * 4 * D + uu;
      gp[0] = gi; gp[D] = gf; gp[2L * D] = gg; gp[3L * D] = go;
      a.c[((long)(i + 1) * B + bm) * D + uu] = cn;
      store4_sc1(a.z + ((long)(i + 1) * B + bm) * D + uu, h);
      cprev = cn;
      if (i + 1 < L1) {
#pragma unroll
        for (int q = 0; q < 4; ++q) pg[q] = a.gates[((long)(i + 1) * B + bm) * 4 * D + (long)q * D + uu];
      }
    }
    DEC_STAMP(5);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    DEC_STAMP(6);
    if (tid == 0) __hip_atomic_store(a.zflag + (long)x * 32, (unsigned)(i + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// LDS floats of an attention workgroup (one place: the kernel's carve and the host's plan)
struct AttLds {
  int NKP, apn, Tp, EC4;
  int o_enc, o_wl, o_ap, o_wcs, o_dps, o_zs, o_red, o_scr, total;
  __host__ __device__ AttLds(int T, int TC, int Fh, int esw) {
    const int Kf = 2 * Fh + 1, nksr = (Kf + 3) / 4;
    NKP = CK * ((nksr + CK - 1) / CK);
    Tp = (T + 3) & ~3;
    EC4 = esw / 4;
    apn = ((T + 15) & ~15) + 16 * 4 * TPW + 3 * nksr + NKP + 8;     // (a wave fetches the operands of TPW tiles, 64 frames apart, used or not)
    int o = 0;
    o_enc = o; o += TC * esw;                 // [TC][EC4] float4: enc[b, frames, column slice]
    o_wl = o; o += Tp;                        // [Tp] softmax weights of the last token
    o_ap = o; o += (apn + 3) & ~3;            // [apn] w_{i-1} with Fh zeros in front
    o_wcs = o; o += 16 * 4 * NKP;             // [16 channels][4 lane groups][NKP] conv taps, tap = group * nksr + s (zero beyond)
    o_dps = o; o += 64;                       // dp of this slice
    o_zs = o; o += 4 * DQ;                    // z_i[b] (zero beyond D)
    o_red = o; o += 32;
    o_scr = o; o += 8 * 128;                  // [8 frame groups][esw <= 128]
    total = o;
  }
};

__device__ void att_role(const DecFwdArgs& a, const int q, float* sm) {
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int B = a.B, T = a.T, E = a.E, D = a.D, A = a.A, C = a.C, Fh = a.Fh, L1 = a.L1, NS = a.NS, TC = a.TC, esw = a.esw;
  const int per_b = NS * a.NFC;
  const int b = q / per_b, rr = q % per_b, ch = rr % NS, fc = rr / NS;
  const int t0 = fc * TC, nt = min(TC, T - t0);
  const int a0 = 64 * ch, e0 = esw * ch;
  const int na = max(0, min(64, A - a0)), ne = max(0, min(esw, E - e0));
  const int Kf = 2 * Fh + 1, nksr = (Kf + 3) / 4;
  const AttLds L(T, TC, Fh, esw);
  const int NKP = L.NKP, apn = L.apn, EC4 = L.EC4, ntile = (nt + 15) >> 4;
  f32x4* encs = reinterpret_cast<f32x4*>(sm + L.o_enc);
  float* wl = sm + L.o_wl;
  float* ap = sm + L.o_ap;
  float* wcs = sm + L.o_wcs;
  float* dps = sm + L.o_dps;
  float* zs = sm + L.o_zs;
  float* red = sm + L.o_red;
  float* scr = sm + L.o_scr;
  const int hl = a.hlens[b];
  const float gb = a.gvec_b[0];
  const int m = lane & 15, kq = lane >> 4;
  // ---- resident operands.  Wave w owns the 16-frame tiles w, w + 4, ... of the chunk (for the conv, for u and for the energies); of a
  // tile, lane (m, kq) is frame m and -- in the accumulator layout of v_mfma_f32_16x16x4 with the frames as N -- rows 4 kq .. 4 kq + 3 of
  // every 16-row block: conv channels 4 kq + r, attention columns 16 at + 4 kq + r.  The conv accumulators are therefore already the B
  // operand of u^T = pre^T + W_att conv^T, u stays in registers until the energies, and a frame's energy is a sum inside 4 lanes.
  f32x4 ptT[TPW][4];                              // pre[b, frame, a0 + 16 at + 4 kq + r]: the C operand of u^T
  f32x4 wA[4];                                    // W_att[a0 + 16 at + m][4 kq + ks]: A operand of u^T, k-step ks
  f32x4 gvr[4];                                   // gvec[a0 + 16 at + 4 kq + r]
#pragma unroll
  for (int j = 0; j < TPW; ++j) {
    const int f = 16 * (wid + 4 * j) + m;
#pragma unroll
    for (int at = 0; at < 4; ++at)
      ptT[j][at] = (f < nt && 16 * at + 4 * kq < na) ? *reinterpret_cast<const f32x4*>(a.pre + ((long)b * T + t0 + f) * A + a0 + 16 * at + 4 * kq) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int at = 0; at < 4; ++at) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) wA[at][ks] = (16 * at + m < na && 4 * kq + ks < C) ? a.w_att[(long)(a0 + 16 * at + m) * C + 4 * kq + ks] : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) gvr[at][r] = 16 * at + 4 * kq + r < na ? a.gvec[a0 + 16 * at + 4 * kq + r] : 0.f;
  }
  // mlp_dec[slice, :] for dp: lane (al = 16 w + m, quarter kq of the decoder units)
  const int al = 16 * wid + m, d0 = DQ * kq;
  float wdec[DQ];
#pragma unroll
  for (int u = 0; u < DQ; ++u) wdec[u] = (d0 + u < D && al < na) ? a.w_decT[(long)(d0 + u) * A + a0 + al] : 0.f;
  for (int idx = tid; idx < nt * EC4; idx += NT) {
    const int tt = idx / EC4, c4 = idx % EC4;
    encs[idx] = 4 * c4 < ne ? *reinterpret_cast<const f32x4*>(a.enc + ((long)b * T + t0 + tt) * E + e0 + 4 * c4) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  for (int idx = tid; idx < 16 * 4 * NKP; idx += NT) {
    const int c = idx / (4 * NKP), g = (idx / NKP) & 3, s2 = idx % NKP, k = g * nksr + s2;
    wcs[idx] = (c < C && s2 < nksr && k < Kf) ? a.w_conv[c * Kf + k] : 0.f;
  }
  if (tid < 64) dps[tid] = 0.f;
  for (int idx = tid; idx < 4 * DQ; idx += NT) zs[idx] = 0.f;
  for (int idx = tid; idx < L.Tp; idx += NT) wl[idx] = (idx < hl && idx < T) ? 1.0f / (float)hl : 0.f;       // att_prev of token 0: uniform over the valid frames
  __syncthreads();
  const __amdgpu_buffer_rsrc_t z_rs = __builtin_amdgcn_make_buffer_rsrc(a.z, 0, (int)((long)(L1 + 1) * B * D * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t e_rs = __builtin_amdgcn_make_buffer_rsrc(a.ebuf, 0, (int)(2L * B * NS * T * 4), 0x00020000);
  bool aborted = false;
  for (int i = 0; i < L1; ++i) {
    const int par = i & 1;
    DEC_STAMP(0);
    // ---- location conv of w_{i-1} (in wl) for this wave's tiles: conv^T[c][t] = sum_k w_conv[c][k] w[t + k - Fh], the taps as the A operand, a
    // (taps x frames) Toeplitz B operand straight from LDS.  Tap k = kq * nksr + s: a lane's taps are consecutive and the operands of CK steps
    // are fetched together (one wave per SIMD: nothing else hides an LDS round trip); the tiles of a wave share the tap operand. ----
    for (int idx = tid; idx < apn; idx += NT) {
      const int t = idx - Fh;
      ap[idx] = (t >= 0 && t < T) ? wl[t] : 0.f;
    }
    __syncthreads();
    f32x4 cvT[TPW];
#pragma unroll
    for (int j = 0; j < TPW; ++j) cvT[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    {
      const float* apb = ap + t0 + 16 * wid + m + kq * nksr;
      const float* wb = wcs + (m * 4 + kq) * NKP;
      for (int s0 = 0; s0 < NKP; s0 += CK) {
        float tv[CK], fv[TPW][CK];
#pragma unroll
        for (int s2 = 0; s2 < CK; ++s2) {
          tv[s2] = wb[s0 + s2];
#pragma unroll
          for (int j = 0; j < TPW; ++j) fv[j][s2] = apb[64 * j + s0 + s2];
        }
#pragma unroll
        for (int s2 = 0; s2 < CK; ++s2)
#pragma unroll
          for (int j = 0; j < TPW; ++j)
            if (wid + 4 * j < ntile) cvT[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(tv[s2], fv[j][s2], cvT[j], 0, 0, 0);
      }
    }
    DEC_STAMP(1);
    // ---- u^T = pre^T + W_att conv^T: the conv accumulators are the B operand as they are ----
    f32x4 u[TPW][4];
#pragma unroll
    for (int j = 0; j < TPW; ++j)
#pragma unroll
      for (int at = 0; at < 4; ++at) {
        f32x4 acc = ptT[j][at];
        if (wid + 4 * j < ntile) {
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wA[at][ks], cvT[j][ks], acc, 0, 0, 0);
        }
        u[j][at] = acc;
      }
    if (ch == 0) {                                  // saved for the backward
#pragma unroll
      for (int j = 0; j < TPW; ++j) {
        const int f = 16 * (wid + 4 * j) + m;
        if (f < nt) {
          float* co = a.conv + (((long)i * B + b) * T + t0 + f) * C + 4 * kq;
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (4 * kq + r < C) co[r] = cvT[j][r];
        }
      }
    }
    DEC_STAMP(2);
    // ---- dp[slice] = W_dec[slice, :] z_i[b]  (z_0 = 0) ----
    if (i > 0) {
      if (!aborted && !wait_flags(a.zflag, a.NG, (unsigned)i, a.err, lane)) aborted = true;
      DEC_STAMP(3);
      for (int d = tid; d < D; d += NT) zs[d] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(z_rs, (unsigned)((((long)i * B + b) * D + d) * 4), 0, 16));
      __syncthreads();
      float s0 = 0.f, s1 = 0.f;
#pragma unroll
      for (int h = 0; h < 2; ++h) {                 // two batches of 16-byte reads
        f32x4 zv[DQ / 8];
#pragma unroll
        for (int j = 0; j < DQ / 8; ++j) zv[j] = *reinterpret_cast<const f32x4*>(zs + d0 + (DQ / 2) * h + 4 * j);
#pragma unroll
        for (int j = 0; j < DQ / 8; ++j) {
          const int u4 = (DQ / 2) * h + 4 * j;
          s0 += wdec[u4] * zv[j][0]; s1 += wdec[u4 + 1] * zv[j][1];
          s0 += wdec[u4 + 2] * zv[j][2]; s1 += wdec[u4 + 3] * zv[j][3];
        }
      }
      float sd = s0 + s1;
      sd += __shfl_xor(sd, 16, 64);
      sd += __shfl_xor(sd, 32, 64);
      if (kq == 0) dps[al] = sd;
      __syncthreads();
    }
    DEC_STAMP(4);
    if (fc == 0 && tid < na) a.dpj[((long)i * B + b) * A + a0 + tid] = dps[tid];
    // ---- partial energies of this slice: 16 columns per lane and tile, a frame's four lanes summed by two shuffles ----
    {
      f32x4 dpr[4];
#pragma unroll
      for (int at = 0; at < 4; ++at) dpr[at] = *reinterpret_cast<const f32x4*>(dps + 16 * at + 4 * kq);
#pragma unroll
      for (int j = 0; j < TPW; ++j) {
        float ev = 0.f;
        if (wid + 4 * j < ntile) {
#pragma unroll
          for (int at = 0; at < 4; ++at)
#pragma unroll
            for (int r = 0; r < 4; ++r) ev += gvr[at][r] * tanh_fast(u[j][at][r] + dpr[at][r]);
        }
        ev += __shfl_xor(ev, 16, 64);
        ev += __shfl_xor(ev, 32, 64);
        const int f = 16 * (wid + 4 * j) + m;
        if (kq == 0 && f < nt) store4_sc1(a.ebuf + (((long)par * B + b) * NS + ch) * T + t0 + f, ev);
      }
    }
    DEC_STAMP(5);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    DEC_STAMP(6);
    if (tid == 0) __hip_atomic_store(a.eflag + (long)q * 32, (unsigned)(i + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // ---- all partial energies of utterance b -> softmax over ALL T frames (incl. padding, as upstream) ----
    if (!aborted && !wait_flags(a.eflag + (long)b * per_b * 32, per_b, (unsigned)(i + 1), a.err, lane)) aborted = true;
    DEC_STAMP(7);
    float mx = -3.0e38f;
    for (int t = tid; t < T; t += NT) {
      float pv[NSMAX];
#pragma unroll
      for (int c2 = 0; c2 < NSMAX; ++c2)             // (a slice that does not exist: offset out of range, the buffer load returns 0)
        pv[c2] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(e_rs, c2 < NS ? (unsigned)(((((long)par * B + b) * NS + c2) * T + t) * 4) : 0xFFFFFFF0u, 0, 16));
      float s = 0.f;
#pragma unroll
      for (int c2 = 0; c2 < NSMAX; ++c2) s += pv[c2];
      const float v = 2.f * (s + gb);
      wl[t] = v;
      mx = fmaxf(mx, v);
    }
    mx = block_max(mx, red);
    float sum = 0.f;
    for (int t = tid; t < T; t += NT) { const float v = __expf(wl[t] - mx); wl[t] = v; sum += v; }
    sum = block_sum(sum, red);
    const float inv = 1.0f / sum;
    for (int t = tid; t < T; t += NT) {             // (a thread normalises the entries it wrote)
      const float v = wl[t] * inv;
      wl[t] = v;
      if (rr == 0) a.w[((long)i * B + b) * T + t] = v;
    }
    __syncthreads();
    DEC_STAMP(8);
    // ---- context piece: this slice's columns over this chunk's frames (32 float4 columns x 8 frame groups) ----
    {
      const int c4 = tid & 31, tg = tid >> 5;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      if (c4 < EC4) {
        int tt = tg;
        for (; tt + 24 < nt; tt += 32) {
          const f32x4 v0 = encs[tt * EC4 + c4], v1 = encs[(tt + 8) * EC4 + c4], v2 = encs[(tt + 16) * EC4 + c4], v3 = encs[(tt + 24) * EC4 + c4];
          acc += v0 * wl[t0 + tt] + v1 * wl[t0 + tt + 8] + v2 * wl[t0 + tt + 16] + v3 * wl[t0 + tt + 24];
        }
        for (; tt < nt; tt += 8) acc += encs[tt * EC4 + c4] * wl[t0 + tt];
      }
      *reinterpret_cast<f32x4*>(scr + tg * 128 + c4 * 4) = acc;
    }
    __syncthreads();
    DEC_STAMP(9);
    if (tid < esw) {
      float sv[8];
#pragma unroll
      for (int g2 = 0; g2 < 8; ++g2) sv[g2] = scr[g2 * 128 + tid];
      float s = ((sv[0] + sv[1]) + (sv[2] + sv[3])) + ((sv[4] + sv[5]) + (sv[6] + sv[7]));
      if (aborted) s = __uint_as_float(0x7fc00000u);
      store4_sc1(a.cxp + (((long)par * B + b) * a.NFC + fc) * a.ES + e0 + tid, s);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (esw > 64) __syncthreads();                  // (two storing waves: both drained before the flag)
    DEC_STAMP(10);
    if (tid == 0) __hip_atomic_store(a.cflag + (long)q * 32, (unsigned)(i + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

__global__ __launch_bounds__(NT) void dec_loop_fwd_kernel(DecFwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float dsm[];
  const int wg = blockIdx.x;
  if (wg < a.NG) gate_role(a, wg, dsm);
  else att_role(a, wg - a.NG, dsm);
}

struct DecPlan { int NG, NS, NFC, TC, ES, esw, NA; size_t lds, ws; size_t o_z, o_e, o_c, o_eb, o_cx; };

int cu_count() {
  static const int n = [] { int dev = 0, v = 256; if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev); return v; }();
  return n;
}

bool dec_plan(int L1, int B, int T, int E, int D, int A, int C, int Fh, DecPlan& p) {
  if (L1 < 1 || B < 1 || B > 32 || T < 1 || T > 2048 || E < 4 || D < 4 || A < 4 || C < 1 || Fh < 0) return false;
  if (D > 8 * 4 * KGZ || E > 8 * 4 * KGC || (D + 3) / 4 > DQ || C > CP || (E & 3) || (D & 3) || (A & 3)) return false;
  p.NG = cdiv(D, 8);
  p.NS = cdiv(A, 64);                                            // 64 attention columns per slice; the projection columns in as many slices
  p.esw = 4 * cdiv(cdiv(E, p.NS), 4);
  if (p.esw > 128) return false;
  p.ES = p.NS * p.esw;
  const int Kf = 2 * Fh + 1, nks = (Kf + 3) / 4;
  if (nks > NKSMAX) return false;
  for (p.NFC = cdiv(T, 64 * TPW);; ++p.NFC) {                    // frame chunks per utterance: the fewest whose resident slices fit the CU's LDS
    p.TC = cdiv(T, p.NFC);
    const AttLds L(T, p.TC, Fh, p.esw);
    size_t fl = (size_t)L.total, gate = 4 * 32 * 33;
    p.lds = (fl > gate ? fl : gate) * 4 + 16;
    if (p.lds <= 160 * 1024) break;
    if (p.TC <= 16) return false;
  }
  p.NA = B * p.NS * p.NFC;
  if (p.NS > NSMAX || p.NS * p.NFC > 64) return false;
  if (p.NG + p.NA > cu_count()) return false;                    // every workgroup must be resident, one per CU
  size_t o = 128;                                                // [err]
  p.o_z = o; o += (size_t)p.NG * 128;
  p.o_e = o; o += (size_t)p.NA * 128;
  p.o_c = o; o += (size_t)p.NA * 128;
  p.o_eb = o; o += (size_t)2 * B * p.NS * T * 4;
  o = (o + 127) & ~(size_t)127;
  p.o_cx = o; o += (size_t)2 * B * p.NFC * p.ES * 4;
  p.ws = (o + 127) & ~(size_t)127;
  return true;
}

LdsLimit g_dec_lim;
}  // namespace

const unsigned* re2e_dec_abort_counter_ptr_() {      // device address of the loop's give-up counter, for the step gate (lstm.hip)
  static const unsigned* p = [] { void* q = nullptr; return hipGetSymbolAddress(&q, HIP_SYMBOL(g_dec_aborts)) == hipSuccess ? (const unsigned*)q : (const unsigned*)nullptr; }();
  return p;
}

int re2e_dec_abort_count_() {
  unsigned n = 0;
  if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_dec_aborts), sizeof(n)) != hipSuccess) return -1;
  return (int)n;
}

extern "C" size_t re2e_dec_loop_workspace_bytes(int L1, int B, int T, int E, int D, int A, int C, int Fh) {
  static const bool off = [] { const char* e = getenv("RE2E_DEC_PERSIST"); return e && atoi(e) == 0; }();
  DecPlan p;
  if (off || !dec_plan(L1, B, T, E, D, A, C, Fh, p)) return 0;
  return p.ws;
}

extern "C" int re2e_dec_loop_fwd(const float* pre, const float* enc, const int* hlens, const float* w_decT, const float* w_att, const float* w_conv,
                                 const float* gvec, const float* gvec_b, const float* w_ctx, long ldw, const float* w_hh, float* gates, float* z,
                                 float* c, float* w, float* cx, float* conv, float* dpj, int L1, int B, int T, int E, int D, int A, int C, int Fh,
                                 void* ws, size_t ws_bytes, hipStream_t stream) {
  DecPlan p;
  if (!dec_plan(L1, B, T, E, D, A, C, Fh, p)) {
    re2e_set_error("re2e_dec_loop_fwd: shape outside the persistent loop's limits (L1=%d B=%d T=%d E=%d D=%d A=%d C=%d Fh=%d)", L1, B, T, E, D, A, C, Fh);
    return RE2E_EUNSUPPORTED;
  }
  RE2E_CHECK_ARG(ws && ws_bytes >= p.ws, "workspace too small (re2e_dec_loop_workspace_bytes)");
  RE2E_CHECK_ARG(!(ldw & 3), "ldw must be a multiple of 4");
  RE2E_CHECK_ARG(!((reinterpret_cast<uintptr_t>(pre) | reinterpret_cast<uintptr_t>(enc) | reinterpret_cast<uintptr_t>(w_ctx) | reinterpret_cast<uintptr_t>(w_hh) |
                    reinterpret_cast<uintptr_t>(z) | reinterpret_cast<uintptr_t>(cx) | reinterpret_cast<uintptr_t>(ws)) & 15), "operands must be 16-byte aligned");
  char* base = reinterpret_cast<char*>(ws);
  DecFwdArgs a;
  a.pre = pre; a.enc = enc; a.hlens = hlens; a.w_decT = w_decT; a.w_att = w_att; a.w_conv = w_conv; a.gvec = gvec; a.gvec_b = gvec_b;
  a.w_ctx = w_ctx; a.ldw = ldw; a.w_hh = w_hh; a.gates = gates; a.z = z; a.c = c; a.w = w; a.cx = cx; a.conv = conv; a.dpj = dpj;
  a.L1 = L1; a.B = B; a.T = T; a.E = E; a.D = D; a.A = A; a.C = C; a.Fh = Fh;
  a.NG = p.NG; a.NS = p.NS; a.NFC = p.NFC; a.TC = p.TC; a.ES = p.ES; a.esw = p.esw;
  a.err = reinterpret_cast<unsigned*>(base);
  a.zflag = reinterpret_cast<unsigned*>(base + p.o_z);
  a.eflag = reinterpret_cast<unsigned*>(base + p.o_e);
  a.cflag = reinterpret_cast<unsigned*>(base + p.o_c);
  a.ebuf = reinterpret_cast<float*>(base + p.o_eb);
  a.cxp = reinterpret_cast<float*>(base + p.o_cx);
  a.stamps = exp_env("RE2E_DEC_STAMPS") ? (unsigned long long*)strtoull(exp_env("RE2E_DEC_STAMPS"), nullptr, 16) : nullptr;
  (void)hipMemsetAsync(ws, 0, p.o_eb, stream);                    // error word and flags start at zero, every call
  // Only the LDS it needs (104 of 160 KB at config 4): unlike the 64-workgroup recurrences (lstm.hip) the loop does not gain from owning its
  // CUs -- what runs beside it are the CTC / CORAL heads, which the critical stream joins right behind the loop (step 52.1 -> 52.0, and
  // 51.4 -> 51.2 with the recurrences' rule; RE2E_DEC_OWN_CU=1 in the experiments build asks for the whole CU)
  const size_t lds = (exp_env("RE2E_DEC_OWN_CU") && atoi(exp_env("RE2E_DEC_OWN_CU")) == 1) ? (size_t)160 * 1024 : p.lds;
  g_dec_lim.ensure(reinterpret_cast<const void*>(&dec_loop_fwd_kernel), lds);
  hipLaunchKernelGGL(dec_loop_fwd_kernel, dim3(p.NG + p.NA), dim3(NT), lds, stream, a);
  RE2E_LAUNCH_CHECK();
  return RE2E_OK;
}

// =================================================================================================================================
// The loop's BACKWARD as one persistent launch (the reverse of e2e_decoder.py:113-152 / e2e_attention.py:259-299 per output token):
//   (a) dh = dz + dZ[i];  LSTMCell backward -> dgates_i                          "unit" workgroups (16 decoder units each)
//   (b) d cx_i = dgates_i W_ih[:, Dd:],  dz' = dgates_i W_hh                      "column" workgroups (16 projection columns each) / the unit workgroups
//   (c) attention backward of utterance b: softmax, energies, location conv -> d dec_proj_i, d att_prev (= d w_{i-1}), de_i, d conv_i
//   (d) dz = dz' + d dec_proj_i mlp_dec                                            the unit workgroups, in front of (a) of token i - 1
// replacing the 5-6 launches per token of ops.DecoderLoopFn.backward (cell / skinny2 / attloc_bwd_frames / attloc_bwd_conv / skinny: 70 us of
// kernels per token at config 4 plus their launch gaps, 41 or 151 tokens).  The attention workgroup here owns a chunk of <= 48 FRAMES of
// one utterance with ALL attention columns (the forward's owns a column slice of all frames): dc . enc[t], d conv[t] and the energy
// gradient of a frame are then complete inside the workgroup; what crosses workgroups per token is dgates (all-gather, read as MFMA operand),
// d cx, the per-chunk partial sums of d dec_proj (reduced by the utterance's own workgroups), the d conv rows of the neighbouring chunks
// (transposed location conv) and one scalar per chunk (the softmax normaliser's w . d w term).  Same flagged hand-off as the forward.
// Only what the recurrence needs is done per token; d W_conv is summed after the loop from the saved d conv rows (dwconv_all_kernel), d pre /
// d W_att / d gvec by attloc_dpre, d enc by attloc_denc as before.
#ifdef RE2E_EXPERIMENTS
#define BWD_STAMP(ph)                                                                                                   \
  do {                                                                                                                  \
    if (a.stamps && threadIdx.x == 0 && it >= kStamp0 && it < kStamp0 + 16)                                             \
      a.stamps[((long)blockIdx.x * 16 + (it - kStamp0)) * 16 + (ph)] = __builtin_amdgcn_s_memrealtime();                \
  } while (0)
#else
#define BWD_STAMP(ph)
#endif
namespace {
constexpr int FPA = 48;             // frames per attention workgroup (3 tiles of 16)
constexpr int TPA = 3;
constexpr int ATW = 5;              // 16-column blocks of the attention dimension per wave: A <= 320
constexpr int GKU = 19;             // groups of 16 gate rows per wave: 4 D <= 1216
constexpr int GKA = 5;              // groups of 16 attention columns per wave of the mlp_dec product
constexpr int NFRMAX = 16;

struct DecBwdArgs {
  const float *pre, *enc, *cx, *z, *c, *w, *conv, *dpj, *dZ;
  const int* hlens;
  const float *w_ctx; long ldw;
  const float *w_hh, *w_dec, *w_att, *w_conv, *gvec;
  float *gates, *d_cx, *de_all, *ddp, *d_conv, *d_pre;
  int L1, B, T, E, D, A, C, Fh;
  int NU, NC, NFR, FR;              // unit / column workgroups, frame chunks per utterance, frames per chunk
  unsigned *err, *f1, *f2, *f4, *f5, *f6a, *f6;
  float *scal, *ddpp;               // [2][B][NFR], [2][B][NFR][AP]
  float* xg;                        // [2][4 D / 16 + 1][32 utterances][16]: d(gates) in the order the gather reads it (one k-group = 2 KB contiguous)
  int AP, ARW;
  unsigned long long* stamps;
};

__device__ __forceinline__ void store16f_sc1(float* p, const f32x4& v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}

// sum over the 16 lanes of a DPP row (all 16 end with the total): two quad permutes, the half-row and the row mirror
__device__ __forceinline__ float row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));     // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));     // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));    // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));    // row_mirror
  return v;
}

// The product both GEMM roles run per token: out^T[col][b] = sum_k W[k][col] dgates[b][k] over this wave's quarter of the k-groups
// (k = 16 g + 4 kq + j, g = wave + 4 gi: one 16-byte load of dgates per lane and four MFMAs).  acc[tile]: 16 utterances each.
template <int NG_, class AOP, class BOFF>
__device__ __forceinline__ void gather_mfma(const __amdgpu_buffer_rsrc_t rs, const BOFF& boff, const int ntb, const AOP& aop, f32x4 (&acc)[2]) {
  f32x4 b0[NG_], b1[NG_];
#pragma unroll
  for (int gi = 0; gi < NG_; ++gi) {
    b0[gi] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, boff(gi, 0), 0, 16));
    b1[gi] = (ntb > 1) ? __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, boff(gi, 1), 0, 16)) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int gi = 0; gi < NG_; ++gi) {
    const f32x4 wv4 = aop(gi);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv4[j], b0[gi][j], acc[0], 0, 0, 0);
      if (ntb > 1) acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv4[j], b1[gi][j], acc[1], 0, 0, 0);
    }
  }
}

__device__ void bwd_unit_role(const DecBwdArgs& a, const int x, float* sm, const bool is_col) {
  // unit workgroup: 16 decoder units (cell backward + dz); column workgroup (is_col): 16 columns of d cx
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6), m = lane & 15, kq = lane >> 4;
  const int B = a.B, D = a.D, E = a.E, A = a.A, L1 = a.L1, K = 4 * D;
  const int ntb = B > 16 ? 2 : 1;
  const int c0 = 16 * x, ncol = is_col ? E : D;
  const bool colok = c0 + m < ncol;
  // A operands W[k][c0 + m]: the 4 D x 16 slice of the gate-row product in LDS as [group of 16 k][kq][m][4] (a lane's four k of a group are one
  // 16-byte read, issued before the flags are polled); mlp_dec's 16 columns (unit workgroups) in registers
  float* wl = sm + 4 * 2 * 64 * 4;
  for (int idx = tid; idx < (K / 16 + 1) * 256; idx += NT) {
    const int g = idx >> 8, rem = idx & 255, k = 16 * g + 4 * (rem >> 6) + (rem & 3), mm = (rem >> 2) & 15;
    wl[idx] = (k < K && c0 + mm < ncol) ? (is_col ? a.w_ctx[(long)k * a.ldw + c0 + mm] : a.w_hh[(long)k * D + c0 + mm]) : 0.f;
  }
  f32x4 wd[GKA];
#pragma unroll
  for (int gi = 0; gi < GKA; ++gi)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = 16 * (wid + 4 * gi) + 4 * kq + j;
      wd[gi][j] = (!is_col && k < A && colok) ? a.w_dec[(long)k * D + c0 + m] : 0.f;
    }
  __syncthreads();
  const int ngrp = K / 16;
  auto a_lds = [&](int gi) { const int g = wid + 4 * gi; return *reinterpret_cast<const f32x4*>(wl + (((g <= ngrp ? g : ngrp) * 4 + kq) * 16 + m) * 4); };
  auto a_reg = [&](int gi) { return wd[gi]; };
  f32x4* red = reinterpret_cast<f32x4*>(sm);                     // [4 waves][2 tiles][64 lanes]
  const int XGP = (K / 16 + 1) * 32 * 16;                        // floats per parity of the packed d(gates)
  const __amdgpu_buffer_rsrc_t g_rs = __builtin_amdgcn_make_buffer_rsrc(a.xg, 0, 2 * XGP * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t p_rs = __builtin_amdgcn_make_buffer_rsrc(a.ddp, 0, (int)((long)L1 * B * A * 4), 0x00020000);
  const bool ok0 = m < B, ok1 = 16 + m < B;
  const int NA = B * a.NFR;
  // the thread's two outputs after the cross-wave sum: utterance bm, columns jj and jj + 8
  const int bm = tid >> 3, jj = tid & 7;
  const bool bok = bm < B;
  float dcr[2] = {0.f, 0.f};
  f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  bool aborted = false;
  for (int it = 0; it < L1; ++it) {
    const int i = L1 - 1 - it;
    BWD_STAMP(0);
    if (!is_col) {
      // ---- (d) + (a): dz = dz' (in acc since the last token) + d dec_proj_{i+1} mlp_dec, then the cell backward ----
      float gv[2][4], cp[2], cc[2], dzo[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {                                // operands of the cell: in flight during the wait
        const int u = c0 + jj + 8 * h;
        const bool ok = bok && u < D;
#pragma unroll
        for (int g = 0; g < 4; ++g) gv[h][g] = ok ? a.gates[((long)i * B + bm) * K + (long)g * D + u] : 0.f;
        cp[h] = ok ? a.c[((long)i * B + bm) * D + u] : 0.f;
        cc[h] = ok ? a.c[((long)(i + 1) * B + bm) * D + u] : 0.f;
        dzo[h] = ok ? a.dZ[((long)i * B + bm) * D + u] : 0.f;
      }
      if (it > 0) {
        if (!aborted && !wait_flags(a.f6, NA, (unsigned)it, a.err, lane)) aborted = true;
        BWD_STAMP(1);
        const unsigned row_b = (unsigned)((((long)(i + 1) * B + m) * A) * 4);
        auto off_p = [&](int gi, int tl) {
          const int k = 16 * (wid + 4 * gi) + 4 * kq;
          return (k < A && (tl ? ok1 : ok0)) ? row_b + (unsigned)(k + 16 * tl * A) * 4u : 0xFFFFFFF0u;
        };
        gather_mfma<GKA>(p_rs, off_p, ntb, a_reg, acc);
      }
      red[(wid * 2 + 0) * 64 + lane] = acc[0];
      red[(wid * 2 + 1) * 64 + lane] = acc[1];
      __syncthreads();
      BWD_STAMP(2);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int ul = jj + 8 * h, u = c0 + ul;
        const int rl = (ul >> 2) * 16 + (bm & 15), rt = bm >> 4, rr = ul & 3;
        float dz = 0.f;
#pragma unroll
        for (int w4 = 0; w4 < 4; ++w4) dz += red[(w4 * 2 + rt) * 64 + rl][rr];
        if (bok && u < D) {
          const float gi = gv[h][0], gf = gv[h][1], gg = gv[h][2], go = gv[h][3];
          const float tc = tanhf_(cc[h]);
          float d = dz + dzo[h];
          if (aborted) d = __uint_as_float(0x7fc00000u);
          const float dct = d * go * (1.f - tc * tc) + dcr[h];
          float* gp = a.gates + ((long)i * B + bm) * K + u;      // plain layout: what the weight-gradient products read after the loop
          float* xp = a.xg + (long)(it & 1) * XGP;               // packed copy: what the gather of this token reads
          const float dgv[4] = {dct * gg * gi * (1.f - gi), dct * cp[h] * gf * (1.f - gf), dct * gi * (1.f - gg * gg), d * tc * go * (1.f - go)};
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            const int k = g4 * D + u;
            gp[(long)g4 * D] = dgv[g4];
            store4_sc1(xp + ((k >> 4) * 32 + bm) * 16 + (k & 15), dgv[g4]);
          }
          dcr[h] = dct * gf;
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      BWD_STAMP(3);
      if (tid == 0) __hip_atomic_store(a.f1 + (long)x * 32, (unsigned)(it + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      acc[0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (i == 0) break;                                           // nothing consumes dz of the initial state
    }
    // ---- (b): all of dgates_i, times this workgroup's 16 columns ----
    if (!aborted && !wait_flags(a.f1, a.NU, (unsigned)(it + 1), a.err, lane)) aborted = true;
    BWD_STAMP(4);
    {
      const unsigned pb = (unsigned)((it & 1) * XGP);
      auto off_g = [&](int gi, int tl) {                        // k-group g of the packed copy: [g][utterance][16], lane (m, kq) -> 16 bytes at k = 4 kq
        const int g = wid + 4 * gi;
        return (g < ngrp && (tl ? ok1 : ok0)) ? (pb + (unsigned)((g * 32 + 16 * tl + m) * 16 + 4 * kq)) * 4u : 0xFFFFFFF0u;
      };
      gather_mfma<GKU>(g_rs, off_g, ntb, a_lds, acc);
    }
    BWD_STAMP(5);
    if (is_col) {
      red[(wid * 2 + 0) * 64 + lane] = acc[0];
      red[(wid * 2 + 1) * 64 + lane] = acc[1];
      acc[0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
      __syncthreads();
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int ul = jj + 8 * h, u = c0 + ul;
        const int rl = (ul >> 2) * 16 + (bm & 15), rt = bm >> 4, rr = ul & 3;
        float v = 0.f;
#pragma unroll
        for (int w4 = 0; w4 < 4; ++w4) v += red[(w4 * 2 + rt) * 64 + rl][rr];
        if (aborted) v = __uint_as_float(0x7fc00000u);
        if (bok && u < E) store4_sc1(a.d_cx + ((long)i * B + bm) * E + u, v);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      BWD_STAMP(6);
      if (tid == 0) __hip_atomic_store(a.f2 + (long)(x) * 32, (unsigned)(it + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// LDS floats of a backward attention workgroup
struct AttBwdLds {
  int ES, W;
  int o_enc, o_dcs, o_gp, o_de, o_wv, o_dwl, o_dcw, o_wcs, o_rc, o_sc2, o_red, total;
  __host__ __device__ AttBwdLds(int E, int C, int Fh) {
    ES = E + 4;                               // enc row pitch: conflict-free 16-byte reads of consecutive rows
    W = FPA + 2 * Fh + 4;
    int o = 0;
    o_enc = o; o += FPA * ES;                 // enc[b, frames, :]
    o_dcs = o; o += (E + 3) & ~3;             // d cx_i[b]
    o_gp = o; o += 4 * 64;                    // [4 column quarters][frames] partial dc . enc
    o_de = o; o += 64;
    o_wv = o; o += 64;                        // w_i of the own frames
    o_dwl = o; o += 64;                       // d w_i of the own frames (from the transposed conv of token i + 1)
    o_dcw = o; o += 16 * W;                   // [channel][window] d conv rows of the frames within Fh of the own ones
    o_wcs = o; o += 16 * (2 * Fh + 1);        // [channel][tap]
    o_rc = o; o += 4 * TPA * 64 * 4;          // per-wave d conv partials
    o_sc2 = o; o += 2 * 16 * 64;              // [tap half][channel][frame] of the transposed conv
    o_red = o; o += 32;
    total = o;
    (void)C;
  }
};

__device__ void bwd_att_role(const DecBwdArgs& a, const int q, float* sm) {
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6), m = lane & 15, kq = lane >> 4;
  const int B = a.B, T = a.T, E = a.E, A = a.A, C = a.C, Fh = a.Fh, L1 = a.L1, NFR = a.NFR, AP = a.AP;
  const int b = q / NFR, fr = q % NFR;
  const int t0 = fr * a.FR, nt = min(a.FR, T - t0);
  const int Kf = 2 * Fh + 1;
  const AttBwdLds L(E, C, Fh);
  const int ES = L.ES, W = L.W;
  float* encs = sm + L.o_enc;
  float* dcs = sm + L.o_dcs;
  float* gp = sm + L.o_gp;
  float* de = sm + L.o_de;
  float* wv = sm + L.o_wv;
  float* dwl = sm + L.o_dwl;
  float* dcw = sm + L.o_dcw;
  float* wcs = sm + L.o_wcs;
  f32x4* rc = reinterpret_cast<f32x4*>(sm + L.o_rc);
  float* sc2 = sm + L.o_sc2;
  float* red = sm + L.o_red;
  // ---- resident operands: lane (m, kq) of wave w is frame m of every tile and columns 16 at + 4 kq + r of the blocks at = w + 4 qq ----
  f32x4 ptT[TPA][ATW], wAT[ATW];
#pragma unroll
  for (int qq = 0; qq < ATW; ++qq) {
    const int at = wid + 4 * qq;
#pragma unroll
    for (int j = 0; j < TPA; ++j) {
      const int f = 16 * j + m;
      ptT[j][qq] = (f < nt && 16 * at + 4 * kq < A) ? *reinterpret_cast<const f32x4*>(a.pre + ((long)b * T + t0 + f) * A + 16 * at + 4 * kq) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
      wAT[qq][r] = (16 * at + 4 * kq + r < A && m < C) ? a.w_att[(long)(16 * at + 4 * kq + r) * C + m] : 0.f;            // d conv^T: A[channel m][column slot kq]
  }
  for (int idx = tid; idx < FPA * (E / 4); idx += NT) {
    const int f = idx / (E / 4), c4 = idx % (E / 4);
    *reinterpret_cast<f32x4*>(encs + f * ES + 4 * c4) = f < nt ? *reinterpret_cast<const f32x4*>(a.enc + ((long)b * T + t0 + f) * E + 4 * c4) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  for (int idx = tid; idx < 16 * Kf; idx += NT) wcs[idx] = idx < C * Kf ? a.w_conv[idx] : 0.f;
  if (tid < 64) { dwl[tid] = 0.f; de[tid] = 0.f; wv[tid] = 0.f; }
  for (int idx = tid; idx < 16 * W; idx += NT) dcw[idx] = 0.f;
  __syncthreads();
  const __amdgpu_buffer_rsrc_t x_rs = __builtin_amdgcn_make_buffer_rsrc(a.d_cx, 0, (int)((long)L1 * B * E * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t s_rs = __builtin_amdgcn_make_buffer_rsrc(a.scal, 0, (int)(2L * B * NFR * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t pp_rs = __builtin_amdgcn_make_buffer_rsrc(a.ddpp, 0, (int)(2L * B * NFR * AP * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t dv_rs = __builtin_amdgcn_make_buffer_rsrc(a.d_conv, 0, (int)((long)L1 * B * T * C * 4), 0x00020000);
  const int arw = a.ARW;                                          // columns of d dec_proj this workgroup reduces
  f32x4 dpa[TPA][ATW];                                             // d pre of the own frames: the sum of du over the tokens
#pragma unroll
  for (int j = 0; j < TPA; ++j)
#pragma unroll
    for (int qq = 0; qq < ATW; ++qq) dpa[j][qq] = f32x4{0.f, 0.f, 0.f, 0.f};
  bool aborted = false;
  for (int it = 0; it < L1; ++it) {
    const int i = L1 - 1 - it, par = it & 1;
    BWD_STAMP(0);
    // ---- before d cx_i is needed: x = pre + dp + W_att conv_i recomputed from the saved conv rows, dtg = gvec (1 - tanh^2 x) ----
    f32x4 dtg[TPA][ATW];
    {
      f32x4 cvT[TPA], dpr[ATW], gvr[ATW], wA[ATW];               // (gvec and the u^T operand of W_att: re-read per token, registers are short)
#pragma unroll
      for (int qq = 0; qq < ATW; ++qq) {
        const int at = wid + 4 * qq;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          gvr[qq][r] = 16 * at + 4 * kq + r < A ? a.gvec[16 * at + 4 * kq + r] : 0.f;
          wA[qq][r] = (16 * at + m < A && 4 * kq + r < C) ? a.w_att[(long)(16 * at + m) * C + 4 * kq + r] : 0.f;          // u^T: A[column m][channel 4 kq + ks]
        }
      }
#pragma unroll
      for (int j = 0; j < TPA; ++j) {
        const int f = 16 * j + m;
#pragma unroll
        for (int r = 0; r < 4; ++r) cvT[j][r] = (f < nt && 4 * kq + r < C) ? a.conv[(((long)i * B + b) * T + t0 + f) * C + 4 * kq + r] : 0.f;
      }
#pragma unroll
      for (int qq = 0; qq < ATW; ++qq) {
        const int at = wid + 4 * qq;
        dpr[qq] = 16 * at + 4 * kq < A ? *reinterpret_cast<const f32x4*>(a.dpj + ((long)i * B + b) * A + 16 * at + 4 * kq) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
      if (tid < FPA) wv[tid] = tid < nt ? a.w[((long)i * B + b) * T + t0 + tid] : 0.f;
#pragma unroll
      for (int j = 0; j < TPA; ++j)
#pragma unroll
        for (int qq = 0; qq < ATW; ++qq) {
          f32x4 u = ptT[j][qq];
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) u = __builtin_amdgcn_mfma_f32_16x16x4f32(wA[qq][ks], cvT[j][ks], u, 0, 0, 0);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float th = tanh_fast(u[r] + dpr[qq][r]);
            dtg[j][qq][r] = gvr[qq][r] * (1.f - th * th);
          }
        }
    }
    BWD_STAMP(1);
    // the normaliser's w . dw term: one scalar per chunk of the utterance, published at the end of the last token
    float sdot = 0.f;
    if (it > 0) {
      if (!aborted && !wait_flags(a.f5 + (long)b * NFR * 32, NFR, (unsigned)it, a.err, lane)) aborted = true;
      float sv = lane < NFR ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(s_rs, (unsigned)((((long)(par ^ 1) * B + b) * NFR + lane) * 4), 0, 16)) : 0.f;
      sdot = wave_sum(sv);
    }
    const float cx0 = tid < E ? a.cx[((long)i * B + b) * E + tid] : 0.f, cx1 = tid + NT < E ? a.cx[((long)i * B + b) * E + tid + NT] : 0.f;
    // ---- d cx_i[b] ----
    BWD_STAMP(2);
    if (!aborted && !wait_flags(a.f2, a.NC, (unsigned)(it + 1), a.err, lane)) aborted = true;
    BWD_STAMP(3);
    {
      const float d0v = tid < E ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(x_rs, (unsigned)((((long)i * B + b) * E + tid) * 4), 0, 16)) : 0.f;
      const float d1v = tid + NT < E ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(x_rs, (unsigned)((((long)i * B + b) * E + tid + NT) * 4), 0, 16)) : 0.f;
      if (tid < E) dcs[tid] = d0v;
      if (tid + NT < E) dcs[tid + NT] = d1v;
      sdot += block_sum(d0v * cx0 + d1v * cx1, red);            // (its barriers publish dcs)
    }
    BWD_STAMP(4);
    // g[f] = dc . enc[f]: thread = (frame f, quarter p of the columns), two batches of 16-byte reads
    {
      const int f = lane, p = wid, e4 = E / 16;                   // float4 per quarter
      float s = 0.f;
      if (f < FPA) {
        const f32x4* er = reinterpret_cast<const f32x4*>(encs + f * ES) + p * e4;
        const f32x4* dr = reinterpret_cast<const f32x4*>(dcs) + p * e4;
        for (int j0 = 0; j0 < e4; j0 += 8) {
          f32x4 ev[8], dv[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) { ev[j] = j0 + j < e4 ? er[j0 + j] : f32x4{0.f, 0.f, 0.f, 0.f}; dv[j] = j0 + j < e4 ? dr[j0 + j] : f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
          for (int j = 0; j < 8; ++j) s += (ev[j][0] * dv[j][0] + ev[j][1] * dv[j][1]) + (ev[j][2] * dv[j][2] + ev[j][3] * dv[j][3]);
        }
      }
      gp[p * 64 + f] = s;
    }
    __syncthreads();
    if (tid < FPA) {
      const float g = (gp[tid] + gp[64 + tid]) + (gp[128 + tid] + gp[192 + tid]);
      float v = tid < nt ? 2.f * wv[tid] * (dwl[tid] + g - sdot) : 0.f;
      if (aborted) v = __uint_as_float(0x7fc00000u);
      de[tid] = v;
      if (tid < nt) a.de_all[((long)i * B + b) * T + t0 + tid] = v;
    }
    __syncthreads();
    BWD_STAMP(5);
    // ---- du = de dtg;  d dec_proj partial = sum over the own frames;  d conv^T = W_att^T du ----
    {
      f32x4 dsum[ATW], dcv[TPA];
#pragma unroll
      for (int qq = 0; qq < ATW; ++qq) dsum[qq] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < TPA; ++j) {
        const float dej = de[16 * j + m];
        dcv[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int qq = 0; qq < ATW; ++qq) {
          const f32x4 du = dtg[j][qq] * dej;
          dsum[qq] += du;
          dpa[j][qq] += du;
#pragma unroll
          for (int r = 0; r < 4; ++r) dcv[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wAT[qq][r], du[r], dcv[j], 0, 0, 0);
        }
        rc[(wid * TPA + j) * 64 + lane] = dcv[j];
      }
#pragma unroll
      for (int qq = 0; qq < ATW; ++qq) {
#pragma unroll
        for (int r = 0; r < 4; ++r) dsum[qq][r] = row16_sum(dsum[qq][r]);
        const int at = wid + 4 * qq;
        if (m == 0 && 16 * at + 4 * kq < AP) store16f_sc1(a.ddpp + (((long)par * B + b) * NFR + fr) * AP + 16 * at + 4 * kq, dsum[qq]);
      }
    }
    __syncthreads();
    BWD_STAMP(6);
    if (wid < TPA) {                                             // wave j sums the four waves' partials of tile j: lane (frame m, channels 4 kq + r)
      const f32x4 v = (rc[(0 * TPA + wid) * 64 + lane] + rc[(1 * TPA + wid) * 64 + lane]) + (rc[(2 * TPA + wid) * 64 + lane] + rc[(3 * TPA + wid) * 64 + lane]);
      const int f = 16 * wid + m;
      if (f < nt) {
        float* dst = a.d_conv + (((long)i * B + b) * T + t0 + f) * C + 4 * kq;
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (4 * kq + r < C) store4_sc1(dst + r, v[r]);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    BWD_STAMP(7);
    if (tid == 0) {
      __hip_atomic_store(a.f6a + (long)q * 32, (unsigned)(it + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(a.f4 + (long)q * 32, (unsigned)(it + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // ---- this workgroup's columns of d dec_proj_i: the sum of the utterance's chunk partials ----
    if (!aborted && !wait_flags(a.f6a + (long)b * NFR * 32, NFR, (unsigned)(it + 1), a.err, lane)) aborted = true;
    BWD_STAMP(8);
    for (int cl = tid; cl < arw; cl += NT) {
      const int col = fr * arw + cl;
      float pv[NFRMAX];
#pragma unroll
      for (int k2 = 0; k2 < NFRMAX; ++k2)
        pv[k2] = (k2 < NFR && col < AP) ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(pp_rs, (unsigned)(((((long)par * B + b) * NFR + k2) * AP + col) * 4), 0, 16)) : 0.f;
      float s = 0.f;
#pragma unroll
      for (int k2 = 0; k2 < NFRMAX; ++k2) s += pv[k2];
      if (aborted) s = __uint_as_float(0x7fc00000u);
      if (col < A) store4_sc1(a.ddp + ((long)i * B + b) * A + col, s);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    BWD_STAMP(9);
    if (tid == 0) __hip_atomic_store(a.f6 + (long)q * 32, (unsigned)(it + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (i == 0) break;
    // ---- (off the critical path) d w_{i-1} of the own frames: transposed location conv over the d conv rows within Fh frames ----
    if (!aborted && !wait_flags(a.f4 + (long)b * NFR * 32, NFR, (unsigned)(it + 1), a.err, lane)) aborted = true;
    BWD_STAMP(10);
    {
      // window of frames [o, o + wn) = [t0 - Fh, t0 + nt + Fh); its part inside [0, T) is one contiguous range of (frame, channel) floats: all of
      // a thread's loads in flight together, transposed into LDS (the positions outside the range were zeroed once, before the loop)
      const int o = t0 - Fh, tlo = o > 0 ? o : 0, thi = min(T, t0 + nt + Fh), nld = (thi - tlo) * C;
      const unsigned gb0 = (unsigned)((((long)i * B + b) * T + tlo) * C * 4);
      for (int u0 = 0; u0 < nld; u0 += 10 * NT) {
        float tv[10];
#pragma unroll
        for (int u = 0; u < 10; ++u) {
          const int idx = u0 + u * NT + tid;
          tv[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(dv_rs, idx < nld ? gb0 + (unsigned)idx * 4u : 0xFFFFFFF0u, 0, 16));
        }
#pragma unroll
        for (int u = 0; u < 10; ++u) {
          const int idx = u0 + u * NT + tid;
          if (idx < nld) dcw[(idx % C) * W + (tlo - o) + idx / C] = tv[u];
        }
      }
      __syncthreads();
      // d w[t0 + f] = sum_c sum_k w_conv[c][k] d conv[t0 + f - k + Fh][c]: item = (half of the taps, channel, 4 consecutive frames); four taps per
      // round from 4 new window values (the other 3 of the 7 it touches stay in registers) and 4 taps
      const int KH = (Kf + 1) / 2, nitem = 2 * C * (FPA / 4);
      for (int item = tid; item < nitem; item += NT) {
        const int h = item / (C * (FPA / 4)), rem = item % (C * (FPA / 4)), cch = rem / (FPA / 4), f0 = 4 * (rem % (FPA / 4));
        const int k0 = h * KH, k1 = min(Kf, k0 + KH);
        const float* bq = dcw + cch * W + f0 + 2 * Fh;                  // bq[j - k] = d conv[t0 + f0 + j - k + Fh][cch]
        const float* wk2 = wcs + cch * Kf;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        int k = k0;
        float v4 = bq[1 - k], v5 = bq[2 - k], v6 = bq[3 - k];            // v[m] = bq[-k - 3 + m]
        for (; k + 4 <= k1; k += 4) {
          const float v0 = bq[-k - 3], v1 = bq[-k - 2], v2 = bq[-k - 1], v3 = bq[-k];
          const float w0 = wk2[k], w1 = wk2[k + 1], w2 = wk2[k + 2], w3 = wk2[k + 3];
          // x_j(k + t) = v[j - t + 3]
          a0 += w0 * v3 + w1 * v2 + w2 * v1 + w3 * v0;
          a1 += w0 * v4 + w1 * v3 + w2 * v2 + w3 * v1;
          a2 += w0 * v5 + w1 * v4 + w2 * v3 + w3 * v2;
          a3 += w0 * v6 + w1 * v5 + w2 * v4 + w3 * v3;
          v4 = v0; v5 = v1; v6 = v2;
        }
        for (; k < k1; ++k) {
          const float w0 = wk2[k];
          a0 += w0 * bq[-k]; a1 += w0 * bq[1 - k]; a2 += w0 * bq[2 - k]; a3 += w0 * bq[3 - k];
        }
        float* dst = sc2 + (h * 16 + cch) * 64 + f0;
        dst[0] = a0; dst[1] = a1; dst[2] = a2; dst[3] = a3;
      }
      __syncthreads();
      if (tid < 64) {
        float s = 0.f;
        if (tid < FPA)
          for (int cch = 0; cch < C; ++cch) s += sc2[cch * 64 + tid] + sc2[(16 + cch) * 64 + tid];
        if (tid >= nt) s = 0.f;
        dwl[tid] = s;
        const float wprev = tid < nt ? a.w[((long)(i - 1) * B + b) * T + t0 + tid] : 0.f;
        const float ps = wave_sum(s * wprev);
        if (tid == 0) {
          store4_sc1(a.scal + ((long)par * B + b) * NFR + fr, ps);
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __hip_atomic_store(a.f5 + (long)q * 32, (unsigned)(it + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      __syncthreads();
      BWD_STAMP(11);
    }
  }
  if (a.d_pre) {
#pragma unroll
    for (int j = 0; j < TPA; ++j) {
      const int f = 16 * j + m;
#pragma unroll
      for (int qq = 0; qq < ATW; ++qq) {
        const int col = 16 * (wid + 4 * qq) + 4 * kq;
        if (f < nt && col < A) *reinterpret_cast<f32x4*>(a.d_pre + ((long)b * T + t0 + f) * A + col) = dpa[j][qq];
      }
    }
  }
}

__global__ __launch_bounds__(NT) void dec_loop_bwd_kernel(DecBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float dsm2[];
  const int wg = blockIdx.x;
  if (wg < a.NU) bwd_unit_role(a, wg, dsm2, false);
  else if (wg < a.NU + a.NC) bwd_unit_role(a, wg - a.NU, dsm2, true);
  else bwd_att_role(a, wg - a.NU - a.NC, dsm2);
}

// d W_conv of the whole loop from the saved d conv rows: partials[b][off + c Kf + k] += sum_i sum_t d conv_i[b][t][c] att_prev_i[b][t + k - Fh]
// (att_prev_0 = uniform over the valid frames, att_prev_i = w_{i-1}).  grid (B, C): thread = tap.
__global__ __launch_bounds__(NT) void dwconv_all_kernel(const float* __restrict__ w_all, const int* __restrict__ hlens, const float* __restrict__ d_conv, int L1,
                                                        int B, int T, int C, int Fh, int off, int npart, float* partials) {
  extern __shared__ __attribute__((aligned(16))) float sm3[];
  const int b = blockIdx.x, c = blockIdx.y, tid = threadIdx.x, Kf = 2 * Fh + 1, hl = hlens[b];
  float* ap = sm3;                                  // [T + 2 Fh + 4]
  float* dq = ap + ((T + 2 * Fh + 7) & ~3);         // [T + 4]
  float acc = 0.f;
  for (int i = 0; i < L1; ++i) {
    for (int idx = tid; idx < T + 2 * Fh + 4; idx += NT) {
      const int t = idx - Fh;
      float v = 0.f;
      if (t >= 0 && t < T) v = i > 0 ? w_all[((long)(i - 1) * B + b) * T + t] : (t < hl ? 1.0f / (float)hl : 0.f);
      ap[idx] = v;
    }
    for (int t = tid; t < T + 4; t += NT) dq[t] = t < T ? d_conv[(((long)i * B + b) * T + t) * C + c] : 0.f;
    __syncthreads();
    if (tid < Kf) {
      float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
      const float* x = ap + tid;
      for (int t = 0; t < T; t += 4) { a0 += dq[t] * x[t]; a1 += dq[t + 1] * x[t + 1]; a2 += dq[t + 2] * x[t + 2]; a3 += dq[t + 3] * x[t + 3]; }
      acc += (a0 + a1) + (a2 + a3);
    }
    __syncthreads();
  }
  if (tid < Kf) partials[(long)b * npart + off + c * Kf + tid] += acc;
}

struct DecBwdPlan { int NU, NC, NFR, FR, NA, AP, ARW; size_t lds, ws; size_t o_f1, o_f2, o_f4, o_f5, o_f6a, o_f6, o_sc, o_pp, o_xg, o_dv; };

bool dec_bwd_plan(int L1, int B, int T, int E, int D, int A, int C, int Fh, DecBwdPlan& p) {
  if (L1 < 1 || B < 1 || B > 32 || T < 1 || E < 16 || D < 4 || A < 4 || C < 1 || C > 16 || Fh < 0) return false;
  if ((E & 15) || (D & 3) || (A & 3) || E > 512 || 4 * D > 16 * 4 * GKU || A > 16 * 4 * GKA || A > 64 * ATW) return false;
  p.NU = cdiv(D, 16);
  p.NC = E / 16;
  p.NFR = cdiv(T, FPA);
  if (p.NFR > NFRMAX) return false;
  p.FR = cdiv(T, p.NFR);
  p.NA = B * p.NFR;
  p.AP = 64 * cdiv(A, 64);
  p.ARW = 4 * cdiv(cdiv(p.AP, p.NFR), 4);
  if (2 * Fh + 1 > 255) return false;                             // (dwconv_all_kernel: one thread per tap)
  if (p.NU + p.NC + p.NA > cu_count()) return false;
  const AttBwdLds L(E, C, Fh);
  size_t fl = (size_t)L.total, gemm = 4 * 2 * 64 * 4 + (size_t)(4 * D / 16 + 1) * 256;
  p.lds = (fl > gemm ? fl : gemm) * 4 + 16;
  if (p.lds > 160 * 1024) return false;
  size_t o = 128;
  p.o_f1 = o; o += (size_t)p.NU * 128;
  p.o_f2 = o; o += (size_t)p.NC * 128;
  p.o_f4 = o; o += (size_t)p.NA * 128;
  p.o_f5 = o; o += (size_t)p.NA * 128;
  p.o_f6a = o; o += (size_t)p.NA * 128;
  p.o_f6 = o; o += (size_t)p.NA * 128;
  p.o_sc = o; o += (size_t)2 * B * p.NFR * 4; o = (o + 127) & ~(size_t)127;
  p.o_pp = o; o += (size_t)2 * B * p.NFR * p.AP * 4; o = (o + 127) & ~(size_t)127;
  p.o_xg = o; o += (size_t)2 * (4 * D / 16 + 1) * 32 * 16 * 4; o = (o + 127) & ~(size_t)127;
  p.o_dv = o; o += (size_t)L1 * B * T * C * 4;
  p.ws = (o + 127) & ~(size_t)127;
  return true;
}

LdsLimit g_dec_bwd_lim;
}  // namespace

extern "C" size_t re2e_dec_loop_bwd_workspace_bytes(int L1, int B, int T, int E, int D, int A, int C, int Fh) {
  static const bool off = [] { const char* e = getenv("RE2E_DEC_PERSIST"); return e && (atoi(e) == 0 || atoi(e) == 2); }();      // 2: forward only
  DecBwdPlan p;
  if (off || !dec_bwd_plan(L1, B, T, E, D, A, C, Fh, p)) return 0;
  return p.ws;
}

extern "C" int re2e_dec_loop_bwd(const float* pre, const float* enc, const float* cx, const float* z, const float* c, const float* w, const float* conv,
                                 const float* dpj, const float* dZ, const int* hlens, const float* w_ctx, long ldw, const float* w_hh, const float* mlp_dec,
                                 const float* w_att, const float* w_conv, const float* gvec, float* gates, float* d_cx_all, float* de_all, float* ddp,
                                 float* d_pre, int L1, int B, int T, int E, int D, int A, int C, int Fh, void* ws, size_t ws_bytes, hipStream_t stream) {
  DecBwdPlan p;
  if (!dec_bwd_plan(L1, B, T, E, D, A, C, Fh, p)) {
    re2e_set_error("re2e_dec_loop_bwd: shape outside the persistent loop's limits (L1=%d B=%d T=%d E=%d D=%d A=%d C=%d Fh=%d)", L1, B, T, E, D, A, C, Fh);
    return RE2E_EUNSUPPORTED;
  }
  RE2E_CHECK_ARG(ws && ws_bytes >= p.ws, "workspace too small (re2e_dec_loop_bwd_workspace_bytes)");
  RE2E_CHECK_ARG(!((reinterpret_cast<uintptr_t>(pre) | reinterpret_cast<uintptr_t>(enc) | reinterpret_cast<uintptr_t>(gates) | reinterpret_cast<uintptr_t>(ddp) |
                    reinterpret_cast<uintptr_t>(dpj) | reinterpret_cast<uintptr_t>(ws)) & 15), "operands must be 16-byte aligned");
  char* base = reinterpret_cast<char*>(ws);
  DecBwdArgs a;
  a.pre = pre; a.enc = enc; a.cx = cx; a.z = z; a.c = c; a.w = w; a.conv = conv; a.dpj = dpj; a.dZ = dZ; a.hlens = hlens;
  a.w_ctx = w_ctx; a.ldw = ldw; a.w_hh = w_hh; a.w_dec = mlp_dec; a.w_att = w_att; a.w_conv = w_conv; a.gvec = gvec;
  a.gates = gates; a.d_cx = d_cx_all; a.de_all = de_all; a.ddp = ddp; a.d_pre = d_pre;
  a.L1 = L1; a.B = B; a.T = T; a.E = E; a.D = D; a.A = A; a.C = C; a.Fh = Fh;
  a.NU = p.NU; a.NC = p.NC; a.NFR = p.NFR; a.FR = p.FR; a.AP = p.AP; a.ARW = p.ARW;
  a.err = reinterpret_cast<unsigned*>(base);
  a.f1 = reinterpret_cast<unsigned*>(base + p.o_f1); a.f2 = reinterpret_cast<unsigned*>(base + p.o_f2);
  a.f4 = reinterpret_cast<unsigned*>(base + p.o_f4); a.f5 = reinterpret_cast<unsigned*>(base + p.o_f5);
  a.f6a = reinterpret_cast<unsigned*>(base + p.o_f6a); a.f6 = reinterpret_cast<unsigned*>(base + p.o_f6);
  a.scal = reinterpret_cast<float*>(base + p.o_sc); a.ddpp = reinterpret_cast<float*>(base + p.o_pp);
  a.d_conv = reinterpret_cast<float*>(base + p.o_dv);
  a.xg = reinterpret_cast<float*>(base + p.o_xg);
  a.stamps = exp_env("RE2E_DEC_STAMPS") ? (unsigned long long*)strtoull(exp_env("RE2E_DEC_STAMPS"), nullptr, 16) : nullptr;
  (void)hipMemsetAsync(ws, 0, p.o_sc, stream);
  const size_t lds = (exp_env("RE2E_DEC_OWN_CU") && atoi(exp_env("RE2E_DEC_OWN_CU")) == 1) ? (size_t)160 * 1024 : p.lds;
  g_dec_bwd_lim.ensure(reinterpret_cast<const void*>(&dec_loop_bwd_kernel), lds);
  hipLaunchKernelGGL(dec_loop_bwd_kernel, dim3(p.NU + p.NC + p.NA), dim3(NT), lds, stream, a);
  RE2E_LAUNCH_CHECK();
  return RE2E_OK;
}

extern "C" int re2e_dec_loop_dwconv(const float* w, const int* hlens, const void* ws, size_t ws_bytes, float* partials, int partial_floats, int wconv_offset,
                                    int L1, int B, int T, int E, int D, int A, int C, int Fh, hipStream_t stream) {
  DecBwdPlan p;
  if (!dec_bwd_plan(L1, B, T, E, D, A, C, Fh, p)) { re2e_set_error("re2e_dec_loop_dwconv: shape outside the persistent loop's limits"); return RE2E_EUNSUPPORTED; }
  RE2E_CHECK_ARG(w && hlens && ws && partials && ws_bytes >= p.ws, "bad args");
  const float* d_conv = reinterpret_cast<const float*>(reinterpret_cast<const char*>(ws) + p.o_dv);
  const size_t lds3 = (size_t)(((T + 2 * Fh + 7) & ~3) + T + 8) * sizeof(float);
  hipLaunchKernelGGL(dwconv_all_kernel, dim3(B, C), dim3(NT), lds3, stream, w, hlens, d_conv, L1, B, T, C, Fh, wconv_offset, partial_floats, partials);
  RE2E_LAUNCH_CHECK();
  return RE2E_OK;
}

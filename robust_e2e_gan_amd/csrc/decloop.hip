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
//   attention workgroup (b, ch, fc): utterance b, 64-column slice ch of the attention / projection dimension, frame chunk fc
//       (<= 256 frames, one per thread).  Resident: pre[b, frames, slice] (registers), enc[b, frames, slice] (LDS), mlp_dec[slice, :]
//       (registers), the conv / mlp_att / gvec weights (LDS).
//       step i:  (before z_i arrives) location conv of w_{i-1} for its frames on the matrix core (Toeplitz operand from LDS,
//                v_mfma_f32_16x16x4_f32), u = pre + W_att conv
//                dp[slice] = W_dec[slice, :] z_i[b]              when z_i is complete
//                e_part[t] = sum_{a in slice} gvec[a] tanh(u[t][a] + dp[a])  -> published; the slice axis is what is exchanged, so
//                no workgroup needs another's dp
//                e[t] = sum_ch e_part + gb, softmax over ALL T frames (every workgroup of b redundantly: no w exchange)
//                cx_part[slice] = sum_{t in chunk} w[t] enc[t][slice] -> published
// Per step the critical path is three hand-offs (z, e_part, cx_part) and ~3 us of arithmetic instead of four launches.
// Workgroups ask for the whole CU's LDS (as the recurrences do): everything must be co-resident, spins are bounded, an abort poisons
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
  const size_t lds = 160 * 1024;                                  // the whole CU: nothing else is co-resident with a loop workgroup (see lstm.hip)
  g_dec_lim.ensure(reinterpret_cast<const void*>(&dec_loop_fwd_kernel), lds);
  hipLaunchKernelGGL(dec_loop_fwd_kernel, dim3(p.NG + p.NA), dim3(NT), lds, stream, a);
  RE2E_LAUNCH_CHECK();
  return RE2E_OK;
}

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
#include <stdlib.h>

#include <atomic>

#include "common.h"

namespace {

// RE2E_EXPERIMENTS builds only (tools/lstm_stamps.py): s_memtime stamps of the phases of 16 consecutive steps of a persistent
// recurrence, per wavefront: stamps[((workgroup * 16 + wave) * 16 + slot) * 12 + phase]; entries 10 / 11 = s_memrealtime (100 MHz,
// chip-wide) at the top of the step and at the publish.  The shipped build contains none of it.
#ifdef RE2E_EXPERIMENTS
__device__ unsigned long long* g_lstm_stamps = nullptr;
constexpr int kStampStep0 = 64;
#define LSTM_STAMP_DECL unsigned long long* const stamp_p = g_lstm_stamps
#define LSTM_STAMP(ph)                                                                                                              \
  do {                                                                                                                              \
    if (stamp_p && (threadIdx.x & 63) == 0 && s >= kStampStep0 && s < kStampStep0 + 16)                                             \
      stamp_p[((((long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 16 + (threadIdx.x >> 6)) * 192 + (s - kStampStep0) * 12 + (ph)] = \
          (ph) >= 10 ? __builtin_amdgcn_s_memrealtime() : __builtin_amdgcn_s_memtime();                                             \
  } while (0)
static void lstm_stamps_arm() {
  const char* v = getenv("RE2E_LSTM_STAMPS");
  unsigned long long* p = v ? (unsigned long long*)strtoull(v, nullptr, 16) : nullptr;
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_lstm_stamps), &p, sizeof(p));
}
#else
#define LSTM_STAMP_DECL
#define LSTM_STAMP(ph)
static void lstm_stamps_arm() {}
#endif

__device__ __forceinline__ void store_acc(float* red, const f32x16& acc, int lane) {
  int lr = lane & 31, lh = lane >> 5;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
    red[row * 33 + lr] = acc[r];
  }
}

// acc += sum_q A_q B_q with ALL fragment loads of a batch of 8 k-groups issued before the first MFMA:
// the step kernels are latency-bound, a load -> MFMA dependency per k-group costs one L2 round trip each.
__device__ __forceinline__ void mfma_chain(const f32x4* __restrict__ ap, const f32x4* __restrict__ bp, int QN, f32x16& acc,
                                           int bstride = 64, bool bvalid = true) {
  constexpr int CH = 8;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  for (int q0 = 0; q0 < QN; q0 += CH) {
    f32x4 a[CH], b[CH];
#pragma unroll
    for (int u = 0; u < CH; ++u) {
      int q = q0 + u < QN ? q0 + u : QN - 1;
      a[u] = ap[(long)q * 64];
      b[u] = bvalid ? bp[(long)q * bstride] : zero;
    }
#pragma unroll
    for (int u = 0; u < CH; ++u) {
      if (q0 + u < QN) {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][j], b[u][j], acc, 0, 0, 0);
      }
    }
  }
}

// ---- MFMA-fragment-ordered operand copies -------------------------------------------------------
// A lane of v_mfma_f32_32x32x2_f32 owns one operand row and, with the k assignment
// k = 8Q + 4*(lane>>5) + i, four consecutive k per group Q.  Reading those four floats straight
// from a row-major matrix makes every lane of a load instruction touch a different 4 KB-strided
// row (64 cache lines per instruction, L1 thrash).  Instead the step kernels read operands that
// were stored in fragment order: [tile][Q][lane][4] for the weights (packed once per sequence) and
// [m-tile][k/4][b%32][4] for the recurrent state (written by the previous step's epilogue), so
// that one wave-instruction reads 1 KB (weights) / 2 x 512 B (state) of contiguous memory.
__global__ void pack_w_fwd_kernel(const float* __restrict__ whh, float* __restrict__ wf, int H) {
  // wf[((x*(H/8) + Q)*64 + lane)*4 + i] = whh[((n>>3)*H + 8x + (n&7))*H + 8Q + 4h + i],  n = lane&31, h = lane>>5
  long tot = (long)4 * H * H;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (long)gridDim.x * blockDim.x) {
    int i = (int)(e & 3); int lane = (int)((e >> 2) & 63); long r = e >> 8; int Q = (int)(r % (H / 8)); int x = (int)(r / (H / 8));
    int n = lane & 31, h = lane >> 5;
    wf[e] = whh[((long)((n >> 3) * H + 8 * x + (n & 7))) * H + 8 * Q + 4 * h + i];
  }
}
__global__ void pack_w_bwd_kernel(const float* __restrict__ whh, float* __restrict__ wb, int H, int UW) {
  // backward: workgroup x owns the gate rows of its 8 UW units as its K slice (m = 32 u + 8 gate + unit, u < UW) and produces
  // partial dh over ALL j.  wb[(((x*njt + jt)*4UW + Q)*64 + lane)*4 + i] = whh[row(x, 8Q+4h+i)*H + 32jt + (lane&31)],
  // row(x, m) = ((m>>3)&3)*H + 8 UW x + 8 (m>>5) + (m&7)
  const int njt = (H + 31) / 32, NQ = 4 * UW;
  long tot = (long)(H / (8 * UW)) * njt * NQ * 256;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (long)gridDim.x * blockDim.x) {
    int i = (int)(e & 3); int lane = (int)((e >> 2) & 63); long r = e >> 8; int Q = (int)(r % NQ); r /= NQ; int jt = (int)(r % njt); int x = (int)(r / njt);
    int m = 8 * Q + 4 * (lane >> 5) + i;
    int j = 32 * jt + (lane & 31);
    wb[e] = j < H ? whh[((long)(((m >> 3) & 3) * H + 8 * UW * x + 8 * (m >> 5) + (m & 7))) * H + j] : 0.f;
  }
}

template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void lstm_fwd_step(float* xg_f, float* xg_r, const float* __restrict__ wfrag, float* ybuf,
                                                            float* cbuf, float* hfrag, const int* __restrict__ lens, int T, int B,
                                                            int H, int s) {
  extern __shared__ __attribute__((aligned(16))) float red[];   // [WAVES][32][33]
  constexpr int NTH = WAVES * 64;
  constexpr int ITER = (256 + NTH - 1) / NTH;
  const int dir = blockIdx.z, mt = blockIdx.y, x = blockIdx.x, MT = gridDim.y;
  const int t = dir ? T - 1 - s : s;
  float* xg = dir ? xg_r : xg_f;
  const int j0 = x * 8, b0 = mt * 32;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, lh = lane >> 5;
  const int QN = H / 8 / WAVES;                     // k-groups of 8 per wave
  const int prev_blk = dir ? t + 2 : t;
  const long H2 = 2L * H;
  const long hf_sz = (long)2 * MT * H * 32;          // floats per parity buffer: [dir][mt][H/4][32][4]
  const float* hf_rd = hfrag + ((s & 1) ^ 1) * hf_sz + (long)(dir * MT + mt) * H * 32;
  float* hf_wr = hfrag + (s & 1) * hf_sz + (long)(dir * MT + mt) * H * 32;
  // ---- prefetch the pointwise operands (independent of the matmul) ----
  float pre[ITER][4], cp[ITER];
  int ln[ITER];
#pragma unroll
  for (int it = 0; it < ITER; ++it) {
    int idx = tid + it * NTH;
    int bm = idx >> 3, jj = idx & 7, b = b0 + bm, j = j0 + jj;
    bool ok = idx < 256 && b < B;
    const float* gp = xg + ((long)t * B + (ok ? b : 0)) * 4 * H + j;
#pragma unroll
    for (int g = 0; g < 4; ++g) pre[it][g] = ok ? gp[g * H] : 0.f;
    cp[it] = ok ? cbuf[((long)prev_blk * B + b) * H2 + dir * H + j] : 0.f;
    ln[it] = ok ? lens[b] : 0;
  }
  // ---- h_{t-1} W_hh^T for 32 utterances x (8 units x 4 gates) ----
  const f32x4* wp = reinterpret_cast<const f32x4*>(wfrag) + ((long)(dir * (H / 8) + x) * (H / 8) + wid * QN) * 64 + lane;
  const f32x4* hp = reinterpret_cast<const f32x4*>(hf_rd) + ((long)(2 * wid * QN + lh)) * 32 + (lane & 31);
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  if (s > 0) mfma_chain(hp, wp, QN, acc);
  store_acc(red + wid * (32 * 33), acc, lane);
  __syncthreads();
#pragma unroll
  for (int it = 0; it < ITER; ++it) {
    int idx = tid + it * NTH;
    int bm = idx >> 3, jj = idx & 7, b = b0 + bm, j = j0 + jj;
    if (idx >= 256 || b >= B) continue;
    float v[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float a = 0.f;
      for (int w = 0; w < WAVES; ++w) a += red[w * (32 * 33) + bm * 33 + g * 8 + jj];
      v[g] = a + pre[it][g];
    }
    float gi = sigmoidf_(v[0]), gf = sigmoidf_(v[1]), gg = tanhf_(v[2]), go = sigmoidf_(v[3]);
    float c = gf * cp[it] + gi * gg;
    float h = go * tanhf_(c);
    if (t >= ln[it]) { c = 0.f; h = 0.f; }      // packed semantics: padded outputs 0, state stays 0
    float* go_ = xg + ((long)t * B + b) * 4 * H + j;
    go_[0] = gi; go_[H] = gf; go_[2 * H] = gg; go_[3 * H] = go;
    cbuf[((long)(t + 1) * B + b) * H2 + dir * H + j] = c;
    ybuf[((long)(t + 1) * B + b) * H2 + dir * H + j] = h;
    hf_wr[((long)(j >> 2) * 32 + bm) * 4 + (j & 3)] = h;
  }
}

// ---- persistent forward: the whole sequence in ONE launch ----------------------------------------------------------
// Same decomposition as lstm_fwd_step (workgroup = 8 hidden units x 4 gates x 32 utterances, waves split K = H), but the
// workgroups stay resident for all T steps: the recurrent weights live in registers (QN float4 per lane), the cell state
// in a register of the thread that owns it, the next step's gate pre-activations are prefetched one step ahead, and
// h_t travels between workgroups as 8-byte {step tag, value} granules -- one write-through (sc1) store each, swept with
// sc1 loads until every tag matches (MI355X_MICROARCH.md "R2": the data is the flag, no fence, placement-independent).
// What this removes per step: the kernel boundary, the grid ramp / drain, and re-reading 32 KB of weights per workgroup
// from an L2 that the concurrent filler kernels keep evicting.  Requires every workgroup to be resident (checked by the
// launcher: grid <= CUs); spins are bounded and poison the output instead of hanging.
typedef unsigned long long u64;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) u64 gu64;
__device__ unsigned g_persist_aborts = 0;      // sequences given up because a peer workgroup never published (re2e_lstm_abort_count)
// Polls (~0.7-2 us each) before a workgroup gives up on a peer: ~10-30 s.  The bound exists so that a protocol bug hangs nothing; it is
// NOT a scheduling deadline -- a peer workgroup that is not resident yet (a co-running kernel, e.g. a ring all-reduce waiting for a slow
// rank that is writing a checkpoint, holds the CU it needs) arrives when that kernel ends, and waiting for it is the right thing to do
// (rounds 1-3 gave up after ~0.4 s: one slow replica would have ended an 8-GPU job).
constexpr unsigned kSpinLimit = 1u << 24;

// (Round 4, measured and rejected for THIS kernel: lstm_fwd2's exchange -- 4-byte values with a one-bit tag, one piece per producer polled,
// then one sweep, 16-byte stores -- in place of the granules: half the bytes through the fabric and the same rate alone at 64 workgroups (3.27
// against 3.24 us per step at H = 256 / B = 32; 5.0 against 6.2 at H = 512 / B = 64, where lstm_fwd2 runs anyway), but the training step
// loses 0.4 ms with it (56.3 against 55.9, two A/B rounds); consuming the k-groups in arrival order on top of it: 3.55-3.72 alone.)
// UW = 2: a workgroup owns 16 hidden units (two 32-column gate tiles): it sweeps the recurrent state ONCE for both, runs two MFMA
// chains on it, and 512 of its threads apply the cell.  Half as many workgroups sweep (the swept traffic through the fabric
// halves) and the 512-wide layers' grid fits half of the chip, so it can reserve its CUs like the narrower layers do (the
// launcher's 160 KB LDS request): what slows a chain beside the filler streams is sharing SIMDs with their MFMA streams.
template <int WAVES, int QN, int UW>
__global__ __launch_bounds__(WAVES * 64) void lstm_fwd_persist(float* xg_f, float* xg_r, const float* __restrict__ wfrag, float* ybuf,
                                                               float* cbuf, u64* hx_, unsigned* err, const int* __restrict__ lens,
                                                               int T, int B, int H) {
  extern __shared__ __attribute__((aligned(16))) float red[];   // [UW][WAVES][32][33] | abort flag
  static_assert(WAVES >= 4 * UW, "256 threads per 8 units own the 32 x 8 cell updates");
  const int dir = blockIdx.z, mt = blockIdx.y, MT = gridDim.y, NX = gridDim.x * UW;     // NX = logical producers (8 units each)
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, lr = lane & 31, lh = lane >> 5;
  const int usub = (tid >> 8) < UW ? (tid >> 8) : 0;                 // which of the workgroup's 8-unit groups this thread's cell belongs to
  const int x = blockIdx.x * UW + usub;
  float* xg = dir ? xg_r : xg_f;
  const int j0 = x * 8, b0 = mt * 32;
  constexpr int kAbort = UW * WAVES * 32 * 33;            // red[kAbort] != 0: a sweep timed out
  if (tid == 0) red[kAbort] = 0.f;
  const long H2 = 2L * H;
  // recurrent weights of this wave's K range: registers for the whole sequence
  f32x4 w[UW][QN];
#pragma unroll
  for (int u = 0; u < UW; ++u) {
    const f32x4* wp = reinterpret_cast<const f32x4*>(wfrag) + ((long)(dir * NX + blockIdx.x * UW + u) * NX + wid * QN) * 64 + lane;
#pragma unroll
    for (int q = 0; q < QN; ++q) w[u][q] = wp[q * 64];
  }
  // exchange buffer: [parity][dir][mt][producer x'][lh][b][i] granules; producer x' == k-group Q of the consumers
  gu64* hx = (gu64*)hx_;
  const int par_sz = 2 * MT * NX * 256;                 // granules per parity buffer (< 2^31: checked by the launcher's workspace size)
  const __amdgpu_buffer_rsrc_t hx_rs = __builtin_amdgcn_make_buffer_rsrc(hx_, 0, (int)(2 * par_sz * 8), 0x00020000);
  const int grp = (dir * MT + mt) * NX * 256;
  const int rd_off = grp + (wid * QN) * 256 + (lh * 32 + lr) * 4;
  const bool pw = tid < 256 * UW;
  const int bm = (tid >> 3) & 31, jj = tid & 7, b = b0 + bm, j = j0 + jj;
  const bool ok = pw && b < B;
  const int ln = ok ? lens[b] : 0;
  const int wr_off = grp + x * 256 + ((jj >> 2) * 32 + bm) * 4 + (jj & 3);
  float c = 0.f, pre[4] = {0.f, 0.f, 0.f, 0.f};
  if (ok) {
    const float* gp = xg + ((long)(dir ? T - 1 : 0) * B + b) * 4 * H + j;
#pragma unroll
    for (int g = 0; g < 4; ++g) pre[g] = gp[g * H];
  }
  LSTM_STAMP_DECL;
  __syncthreads();
  for (int s = 0; s < T; ++s) {
    const int t = dir ? T - 1 - s : s;
    LSTM_STAMP(0); LSTM_STAMP(10);
    f32x16 acc[UW];
#pragma unroll
    for (int u = 0; u < UW; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[u][r] = 0.f;
    if (s > 0) {
      const gu64* src = hx + ((s & 1) ^ 1) * par_sz + rd_off;
      float hv[QN][4];
      // cheap wait first: ONE granule per wave-pass until the first producer of this wave's range has published (all
      // producers finish a step within a fraction of a microsecond of each other), then the full validated sweep --
      // 16 waves sweeping 1 KB each in a loop are what makes two resident workgroups on one CU 1.7x slower than one
      // the validated sweep: 16-byte sc1 buffer loads, two per k-group = the lane's four {value, tag} granules (8-byte loads
      // run at 0.54-0.70x the 16-byte rate, and the sweep is the largest part of a step: 2.2 of 4.9 us alone on the chip)
      const unsigned rd_b = (unsigned)((((s & 1) ^ 1) * par_sz + rd_off) * 8);
      for (unsigned spins = 0;; ++spins) {
        asm volatile("" ::: "memory");             // the loads below must be re-issued every pass
        bool good = true;
#pragma unroll
        for (int q = 0; q < QN; ++q) {
          const u32x4 lo = __builtin_amdgcn_raw_buffer_load_b128(hx_rs, rd_b + (unsigned)q * 2048u, 0, 16);
          const u32x4 hi = __builtin_amdgcn_raw_buffer_load_b128(hx_rs, rd_b + (unsigned)q * 2048u + 16u, 0, 16);
          hv[q][0] = __uint_as_float(lo[0]); hv[q][1] = __uint_as_float(lo[2]);
          hv[q][2] = __uint_as_float(hi[0]); hv[q][3] = __uint_as_float(hi[2]);
          good &= (lo[1] == (unsigned)s) & (lo[3] == (unsigned)s) & (hi[1] == (unsigned)s) & (hi[3] == (unsigned)s);
        }
        if (__all(good)) break;
        if (spins > kSpinLimit) { if (lane == 0) { red[kAbort] = 1.f; if (atomicExch(err, 1u) == 0u) atomicAdd(&g_persist_aborts, 1u); } break; }
        __builtin_amdgcn_s_sleep(1);
      }
      LSTM_STAMP(1);
#pragma unroll
      for (int q = 0; q < QN; ++q)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int u = 0; u < UW; ++u) acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(hv[q][i], w[u][q][i], acc[u], 0, 0, 0);
    }
#pragma unroll
    for (int u = 0; u < UW; ++u) store_acc(red + (u * WAVES + wid) * (32 * 33), acc[u], lane);
    LSTM_STAMP(2);
    __syncthreads();
    LSTM_STAMP(3);
    if (red[kAbort] != 0.f) break;
    if (pw) {
      float v[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float a = 0.f;
#pragma unroll
        for (int wv = 0; wv < WAVES; ++wv) a += red[(usub * WAVES + wv) * (32 * 33) + bm * 33 + g * 8 + jj];
        v[g] = a + pre[g];
      }
      float gi = sigmoidf_(v[0]), gf = sigmoidf_(v[1]), gg = tanhf_(v[2]), go = sigmoidf_(v[3]);
      float cn = gf * c + gi * gg;
      float h = go * tanhf_(cn);
      if (t >= ln) { cn = 0.f; h = 0.f; }        // packed semantics (also rows b >= B: ln = 0)
      c = cn;
      // next step's pre-activations: loaded a whole step before the cell needs them, in front of the publish (round 4: issued before the
      // barrier above, as rounds 1-3 did, the cell's s_waitcnt vmcnt(0) waited ~0.5 us per step for these very loads -- see lstm_fwd2)
      if (ok && s + 1 < T) {
        const float* gp = xg + ((long)(dir ? t - 1 : t + 1) * B + b) * 4 * H + j;
#pragma unroll
        for (int g = 0; g < 4; ++g) pre[g] = gp[g * H];
      }
      if (s + 1 < T)
        __hip_atomic_store(hx + (s & 1) * par_sz + wr_off, ((u64)(unsigned)(s + 1) << 32) | (u64)__float_as_uint(h), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
      LSTM_STAMP(4); LSTM_STAMP(11);
      if (ok) {
        float* go_ = xg + ((long)t * B + b) * 4 * H + j;
        go_[0] = gi; go_[H] = gf; go_[2 * H] = gg; go_[3 * H] = go;
        cbuf[((long)(t + 1) * B + b) * H2 + dir * H + j] = cn;
        ybuf[((long)(t + 1) * B + b) * H2 + dir * H + j] = h;
      }
    }
    LSTM_STAMP(5);
    __syncthreads();
    LSTM_STAMP(6);
  }
  if (red[kAbort] != 0.f && ok) {                         // a peer never published: make the failure visible downstream
    for (int t = 0; t < T; ++t) ybuf[((long)(t + 1) * B + b) * H2 + dir * H + j] = __uint_as_float(0x7fc00000u);
  }
}

// ---- round 4: the forward recurrence rebuilt around what its step is made of (profiles/r04_chain_budget.md) ---------------
// Stamps of the kernel above at H = 256 / B = 32 (3.4 us per step): 0.8-1.5 us waiting for h(t-1) (the bare exchange of 64 KB of
// {value, tag} granules per workgroup is 2.7 us of the step, tools/micro/handoff_probe.hip; a CU pulls handed-off bytes from the
// memory side at ~50 GB/s, and a sweep that comes too early is paid again), 0.7-0.8 MFMA, 0.7 for summing the eight K-split
// partial tiles out of LDS + the cell (0.5 of it waiting for the NEXT step's pre-activations: s_waitcnt vmcnt(0) behind loads
// issued just in front), two workgroup barriers.  This form removes what can be removed:
//  * h(t) travels as the 4-byte value itself: |h| <= 1, so bit 30 of its fp32 pattern is always clear and carries a ONE-BIT step
//    tag (it flips with every rewrite of a parity buffer; the buffers start zeroed = tag 0, the first write carries 1).  Half the
//    bytes of the 8-byte granules through the fabric, published as 16-byte sc1 stores, swept with one 16-byte sc1 load per lane
//    and 16 k.  A NaN h is published as the pattern of 1.5 (impossible otherwise) and turned back into a NaN by the consumer, so
//    non-finite inputs still poison exactly what they poison in nn.LSTM.
//  * a workgroup owns 16 UTTERANCES (not 32) x 4 TILES units: what it has to pull per step halves again (16 KB at H = 256,
//    32 KB at H = 512) and the matrix work of a step spreads over twice the CUs.
//  * v_mfma_f32_16x16x4_f32 with the GATE ROWS as M and the utterances as N: a wavefront owns 16 utterances x (TILES x 4 units x
//    4 gates) over its quarter of K; a lane's four accumulator registers are the four gates of ONE (unit, utterance) cell, so the
//    cell non-linearity runs on the accumulators: no 32 x 33 tiles through LDS, no 32 LDS reads per cell.  What crosses waves is
//    one 16-byte partial per lane, tile and K quarter.
//  * the wait for h(t-1) is a cheap poll -- ONE 16-byte piece of every producer in the wave's K quarter, one load instruction --
//    and the state is swept ONCE, when the poll says it is there; chunks of 16 k are then consumed in arrival order (the MFMAs
//    of chunk j issue while chunks j+1.. are in flight); a chunk that is stale anyway is polled alone.
//  * ONE workgroup barrier per step (the partials are double-buffered by step parity); the next pre-activations are loaded a
//    whole step before the cell needs them.
//  * W_hh stays in registers as before, but is read straight from the row-major matrix (16-byte loads, once per sequence): no pack kernel.
// The launch-per-step twin (PERSIST = false) runs the SAME instruction sequence on plain loads, so the two are bitwise equal.
__device__ __forceinline__ void store16u_sc1(unsigned* p, const u32x4& v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
constexpr unsigned kTagBit = 0x40000000u, kNanPub = 0x3FC00000u;
__device__ __forceinline__ bool tags_are(const u32x4& v, unsigned want) {
  return (((v[0] & v[1] & v[2] & v[3]) ^ want) & kTagBit) == 0u && (((v[0] | v[1] | v[2] | v[3]) ^ want) & kTagBit) == 0u;
}

// TT (round 6): utterance tiles per workgroup.  TT = 2: ONE workgroup carries TWO independent 16-utterance tiles (same units, same W_hh registers,
// own tags / parity buffers / cell state per tile) and alternates between them inside a step: publish tile A's h(t) -> poll, MFMA and cell of
// tile B -> poll A, meant to fill the ~1-1.5 us a workgroup waits for its peers' h(t) (profiles/r04_chain_budget.md section 2) with the other
// tile's ~0.8 us of work.  Per tile the instruction sequence is the TT = 1 one: bitwise-equal results.  Measured slower and not selected
// (fwd2_config: why); experiments build only, RE2E_LSTM_FWD2_TT=2.
template <int TILES, int NJ, bool PERSIST, int TT = 1>
__global__ __launch_bounds__(256) void lstm_fwd2(float* xg_f, float* xg_r, const float* __restrict__ whh_f, const float* __restrict__ whh_r,
                                                 float* ybuf, float* cbuf, unsigned* hx_, unsigned* err, const int* __restrict__ lens, int T,
                                                 int B, int H, int s_arg, int mode) {
  static_assert(TT == 1 || PERSIST, "two tiles per workgroup: the persistent form only");
  constexpr int KS = 4;                                         // K quarters = waves
  extern __shared__ __attribute__((aligned(16))) float lds2[];  // TT x part[2][TILES][KS][64] f32x4 | abort word
  f32x4* part_all = reinterpret_cast<f32x4*>(lds2);
  int* abortw = reinterpret_cast<int*>(lds2 + TT * 2 * TILES * KS * 64 * 4);
  const int dir = blockIdx.z, MT = gridDim.y * TT, x = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, ks = tid >> 6, n = lane & 15, kk = lane >> 4;
  const int ft = ks < TILES ? ks : -1;                           // the tile this wave finishes (gates, cell, publish), if any
  float* xg = dir ? xg_r : xg_f;
  const float* whh = dir ? whh_r : whh_f;
  const long H2 = 2L * H;
  if (PERSIST && tid == 0) *abortw = 0;
  // recurrent weights of this wave's rows and K quarter: registers for the whole sequence.  A operand of 16x16x4: lane holds
  // A[m = lane & 15][k = lane >> 4]; row m = 4 * unit + gate.
  f32x4 w[TILES][NJ];
#pragma unroll
  for (int t = 0; t < TILES; ++t) {
    const long row = (long)(n & 3) * H + 4 * TILES * x + 4 * t + (n >> 2);
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj) w[t][jj] = *reinterpret_cast<const f32x4*>(whh + row * H + 16 * (ks * NJ + jj) + 4 * kk);
  }
  // exchange buffer (words): [parity][dir][mt][k / 4][16 utterances][4]
  const int par_w = 2 * MT * H * 16;
  const __amdgpu_buffer_rsrc_t hx_rs = __builtin_amdgcn_make_buffer_rsrc(hx_, 0, (int)(2 * par_w * 4), 0x00020000);
  const bool fin = ft >= 0;
  const int j = 4 * TILES * x + 4 * (fin ? ft : 0) + kk;                               // the unit whose cell this lane owns
  const int s0 = PERSIST ? 0 : s_arg, s1 = PERSIST ? T : s_arg + 1;
  // ---- per utterance tile u of this workgroup ----
  int b_[TT], ln_[TT];
  bool ok_[TT];
  unsigned rd_w_[TT], poll_w_[TT], wr_w_[TT];
  float c_[TT], pre_[TT][4];
  u32x4 ld_[TT][NJ];
#pragma unroll
  for (int u = 0; u < TT; ++u) {
    const int mt = blockIdx.y * TT + u;
    const int grp = (dir * MT + mt) * H * 16;
    b_[u] = 16 * mt + n;
    rd_w_[u] = (unsigned)(grp + ((4 * ks * NJ + kk) * 16 + n) * 4);          // + jj * 256 words
    poll_w_[u] = (unsigned)(grp + ((4 * ks * NJ + (lane < 4 * NJ ? lane : 0)) * 16) * 4);     // utterance 0 of piece lane of this quarter
    ok_[u] = fin && b_[u] < B;
    ln_[u] = ok_[u] ? lens[b_[u]] : 0;
    wr_w_[u] = (unsigned)(grp + ((TILES * x + (fin ? ft : 0)) * 16 + n) * 4);
    c_[u] = 0.f;
#pragma unroll
    for (int g = 0; g < 4; ++g) pre_[u][g] = 0.f;
    if (ok_[u]) {
      const int t0 = dir ? T - 1 - s0 : s0;
      const float* gp = xg + ((long)t0 * B + b_[u]) * 4 * H + j;
#pragma unroll
      for (int g = 0; g < 4; ++g) pre_[u][g] = gp[g * H];
      if (!PERSIST) c_[u] = cbuf[((long)(dir ? t0 + 2 : t0) * B + b_[u]) * H2 + dir * H + j];
    }
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj) ld_[u][jj] = u32x4{0u, 0u, 0u, 0u};
    if (!PERSIST && s0 > 0) {                       // plain loads: the previous launch wrote them
      const u32x4* src = reinterpret_cast<const u32x4*>(hx_ + ((s0 & 1) ^ 1) * par_w) + rd_w_[u] / 4;
#pragma unroll
      for (int jj = 0; jj < NJ; ++jj) ld_[u][jj] = src[jj * 64];
    }
  }
  LSTM_STAMP_DECL;
  if (PERSIST) __syncthreads();
  bool aborted = false;
  for (int s = s0; s < s1 && !aborted; ++s) {
    const int t = dir ? T - 1 - s : s;
#pragma unroll
   for (int u = 0; u < TT; ++u) {
    const int b = b_[u], ln = ln_[u];
    const bool ok = ok_[u];
    const unsigned rd_w = rd_w_[u], poll_w = poll_w_[u], wr_w = wr_w_[u];
    f32x4* part = part_all + u * (2 * TILES * KS * 64);
    LSTM_STAMP(0); LSTM_STAMP(10);
    constexpr int NACC = TILES == 1 ? 2 : TILES;            // one tile: two chains (even / odd k) hide the 40-cycle dependent latency
    f32x4 acc[NACC];
#pragma unroll
    for (int tt = 0; tt < NACC; ++tt) acc[tt] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (s > 0) {
      const unsigned want = PERSIST ? (unsigned)((((s - 1) >> 1) + 1) & 1) << 30 : 0u;
      const unsigned rd_b = (unsigned)((((s & 1) ^ 1) * par_w) * 4) + rd_w * 4u;
      const unsigned pl_b = (unsigned)((((s & 1) ^ 1) * par_w) * 4) + poll_w * 4u;
      // cheap poll: one 16-byte piece of every producer of this K quarter (lanes 0 .. 4 NJ - 1), one load instruction per pass
      auto wait_producers = [&]() {
        for (unsigned spins = 0;; ++spins) {
          asm volatile("" ::: "memory");
          const u32x4 pv = __builtin_amdgcn_raw_buffer_load_b128(hx_rs, pl_b, 0, 16);
          if (__all(tags_are(pv, want))) break;
          if (spins > kSpinLimit) { if (lane == 0) { *abortw = 1; if (atomicExch(err, 1u) == 0u) atomicAdd(&g_persist_aborts, 1u); } break; }
        }
        asm volatile("" ::: "memory");
      };
      if (PERSIST) {
        if (mode & 1) {                         // poll first, sweep once
          wait_producers();
#pragma unroll
          for (int jj = 0; jj < NJ; ++jj) ld_[u][jj] = __builtin_amdgcn_raw_buffer_load_b128(hx_rs, rd_b + (unsigned)jj * 1024u, 0, 16);
        }
        LSTM_STAMP(7);
      }
#pragma unroll
      for (int jj = 0; jj < NJ; ++jj) {
        u32x4 v = ld_[u][jj];
        if (PERSIST) {
          if (!__all(tags_are(v, want))) {
            // the sweep came before this chunk's producer had published: wait for all producers (cheap poll), read this and the
            // later chunks again; should the chunk still be stale (the poll looks at one piece per producer), poll it alone
            wait_producers();
#pragma unroll
            for (int j2 = jj; j2 < NJ; ++j2) ld_[u][j2] = __builtin_amdgcn_raw_buffer_load_b128(hx_rs, rd_b + (unsigned)j2 * 1024u, 0, 16);
            v = ld_[u][jj];
            for (unsigned spins = 0; !__all(tags_are(v, want)); ++spins) {
              asm volatile("" ::: "memory");
              v = __builtin_amdgcn_raw_buffer_load_b128(hx_rs, rd_b + (unsigned)jj * 1024u, 0, 16);
              if (spins > kSpinLimit) { if (lane == 0) { *abortw = 1; if (atomicExch(err, 1u) == 0u) atomicAdd(&g_persist_aborts, 1u); } break; }
            }
          }
        }
        float hv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          unsigned uu = PERSIST ? (v[i] & ~kTagBit) : v[i];
          if (PERSIST) uu = uu == kNanPub ? 0x7fc00000u : uu;
          hv[i] = __uint_as_float(uu);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int tt = 0; tt < TILES; ++tt) {
            const int a = TILES == 1 ? (i & 1) : tt;
            acc[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[tt][jj][i], hv[i], acc[a], 0, 0, 0);
          }
      }
      if (TILES == 1) acc[0] += acc[1];
    }
    LSTM_STAMP(1);
    // K-quarter partials of the tiles this wave does not finish -> LDS (double-buffered by step parity: one barrier per step)
    f32x4* pp = part + (s & 1) * (TILES * KS * 64);
#pragma unroll
    for (int tt = 0; tt < TILES; ++tt)
      if (tt != ft) pp[(tt * KS + ks) * 64 + lane] = acc[tt];
    LSTM_STAMP(2);
    __syncthreads();
    LSTM_STAMP(3);
    if (PERSIST && *abortw) { aborted = true; break; }
    float og[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};      // activated gates, c, h of this lane's cell: written after the next sweep is on its way
    if (fin) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      f32x4 mine = acc[0];
#pragma unroll
      for (int tt = 1; tt < TILES; ++tt) mine = ft == tt ? acc[tt] : mine;
#pragma unroll
      for (int k2 = 0; k2 < KS; ++k2) v += k2 == ks ? mine : pp[(ft * KS + k2) * 64 + lane];      // fixed order
      float gi, gf, gg, go, cn, h;
#ifdef RE2E_EXPERIMENTS
      if (mode & 16) {                           // v_rcp_f32 (1 ulp) instead of the IEEE division: experiment
        auto sg = [](float x_) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x_)); };
        auto th = [](float x_) { float e = __expf(-2.0f * fabsf(x_)); return copysignf((1.0f - e) * __builtin_amdgcn_rcpf(1.0f + e), x_); };
        gi = sg(v[0] + pre_[u][0]); gf = sg(v[1] + pre_[u][1]); gg = th(v[2] + pre_[u][2]); go = sg(v[3] + pre_[u][3]);
        cn = gf * c_[u] + gi * gg;
        h = go * th(cn);
      } else
#endif
      {
        gi = sigmoidf_(v[0] + pre_[u][0]); gf = sigmoidf_(v[1] + pre_[u][1]); gg = tanhf_(v[2] + pre_[u][2]); go = sigmoidf_(v[3] + pre_[u][3]);
        cn = __builtin_fmaf(gf, c_[u], gi * gg);     // spelled out: the persistent kernel and its launch-per-step twin must not contract differently
        h = go * tanhf_(cn);
      }
      if (t >= ln) { cn = 0.f; h = 0.f; }        // packed semantics (also rows b >= B: ln = 0)
      c_[u] = cn;
      // The next step's pre-activations are loaded HERE: a whole step before the cell needs them, and in front of the publish.  The
      // compiler guards a load into registers it cannot prove idle with s_waitcnt vmcnt(0) (the poll loops have exits it cannot count
      // across): behind the publish that wait is the write-through latency of the sc1 store (0.7 us per step at H = 512), just in front
      // of the cell (rounds 1-3: loaded before the barrier) it is the HBM latency of these very loads (0.5 us); here nothing recent is
      // outstanding.
      if (PERSIST && ok && s + 1 < T) {
        const float* gp = xg + ((long)(dir ? t - 1 : t + 1) * B + b) * 4 * H + j;
#pragma unroll
        for (int g = 0; g < 4; ++g) pre_[u][g] = gp[g * H];
      }
      if (!PERSIST || s + 1 < T) {
        // the four units of a 16-byte piece sit in lanes n, n + 16, n + 32, n + 48: gather them into lane n, one store per piece
        unsigned hb = __float_as_uint(h);
        if (PERSIST) hb = (h != h ? kNanPub : hb) | ((unsigned)(((s >> 1) + 1) & 1) << 30);
        u32x4 pv;
        pv[0] = hb;
        pv[1] = (unsigned)__builtin_amdgcn_ds_bpermute(((lane + 16) & 63) * 4, (int)hb);
        pv[2] = (unsigned)__builtin_amdgcn_ds_bpermute(((lane + 32) & 63) * 4, (int)hb);
        pv[3] = (unsigned)__builtin_amdgcn_ds_bpermute(((lane + 48) & 63) * 4, (int)hb);
        if (kk == 0) {
          if (PERSIST) store16u_sc1(hx_ + (s & 1) * par_w + wr_w, pv);
          else *reinterpret_cast<u32x4*>(hx_ + (s & 1) * par_w + wr_w) = pv;
        }
      }
      LSTM_STAMP(4); LSTM_STAMP(11);
      og[0] = gi; og[1] = gf; og[2] = gg; og[3] = go; og[4] = cn; og[5] = h;
    }
    // (Measured and rejected: the speculative sweep timed by a per-wave delay behind the publish that hovers at the edge of being too
    // early -- one late wave anywhere delays its workgroup's publish and with it EVERY peer's next sweep, so with 512 waves a
    // per-wave miss rate of 1 in 64 is a miss on every step: 2.85 against 2.93 us at H = 256, no gain at 512, 43 us at B = 8.)
    // Behind the publish: (mode 0: the next step's sweep, then) this step's outputs -- stores write no register, so no wait guards them.
    auto next_sweep = [&]() {
      const unsigned rd_n = (unsigned)(((s & 1) * par_w) * 4) + rd_w * 4u;
      asm volatile("" ::: "memory");
#pragma unroll
      for (int jj = 0; jj < NJ; ++jj) ld_[u][jj] = __builtin_amdgcn_raw_buffer_load_b128(hx_rs, rd_n + (unsigned)jj * 1024u, 0, 16);
    };
    if (PERSIST && (mode & 1) == 0 && s + 1 < T) next_sweep();
    LSTM_STAMP(5);
    if (ok) {
      float* go_ = xg + ((long)t * B + b) * 4 * H + j;
      go_[0] = og[0]; go_[H] = og[1]; go_[2 * H] = og[2]; go_[3 * H] = og[3];
      cbuf[((long)(t + 1) * B + b) * H2 + dir * H + j] = og[4];
      ybuf[((long)(t + 1) * B + b) * H2 + dir * H + j] = og[5];
    }
    LSTM_STAMP(6);
   }
  }
  if (PERSIST && *abortw) {                           // a peer never published: make the failure visible downstream
#pragma unroll
    for (int u = 0; u < TT; ++u)
      if (ok_[u])
        for (int t = 0; t < T; ++t) ybuf[((long)(t + 1) * B + b_[u]) * H2 + dir * H + j] = __uint_as_float(0x7fc00000u);
  }
}

// BPTT step.  Workgroup x owns hidden units j in [8x, 8x+8) for 32 utterances, exactly like the forward:
//   consume: dh_rec[b][j] = sum over ALL workgroups x' of the partial slabs P_x'[b][j] written by the previous launch,
//            then the cell backward -> d(gates) for its 32 gate columns n (kept in LDS, written to G);
//   produce: P_x[b][:] = dG_x[32 b x 32 n] . W_hh[n in x][all j]  (its K slice of dh_{t-1} = dG W_hh), written in the
//            consumer's order slab[j/8][x][b][j%8] so that the next launch reads 1 KB contiguous runs.
// Splitting K (= 4H gate columns) over the workgroups is what spreads the MFMA work over H/8 CUs per direction
// instead of H/32; the cross-workgroup sum rides on the kernel boundary that the recurrence needs anyway.
template <int JT>       // N tiles (of 32 hidden units) per wave; block = 4 waves; H = 128 * JT ... handled by a loop
__global__ __launch_bounds__(256) void lstm_bwd_step(float* g_f, float* g_r, const float* __restrict__ wb, const float* __restrict__ dy,
                                                     const float* __restrict__ cbuf, float* dc_state, float* slabs,
                                                     const int* __restrict__ lens, int T, int B, int H, int s) {
  __shared__ __attribute__((aligned(16))) float dgs[32 * 36];     // d(gates) tile [b][n], padded rows
  const int dir = blockIdx.z, mt = blockIdx.y, x = blockIdx.x, MT = gridDim.y, NX = gridDim.x;
  const int t = dir ? s : T - 1 - s;            // reverse order of the forward pass
  float* G = dir ? g_r : g_f;
  const int j0 = x * 8, b0 = mt * 32;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, lr = lane & 31, lh = lane >> 5;
  const int K4 = 4 * H;
  const long H2 = 2L * H;
  const int prev_blk = dir ? t + 2 : t;
  const long sl_sz = (long)2 * MT * NX * NX * 256;          // floats per parity buffer: [dir][mt][x''][x][32][8]
  const float* sl_rd = slabs + ((s & 1) ^ 1) * sl_sz + (long)(dir * MT + mt) * NX * NX * 256;
  float* sl_wr = slabs + (s & 1) * sl_sz + (long)(dir * MT + mt) * NX * NX * 256;
  // ---- consume + cell backward: thread = (utterance bm, unit jj) ----
  {
    const int bm = tid >> 3, jj = tid & 7, b = b0 + bm, j = j0 + jj;
    const bool ok = b < B;
    const long bb = ok ? b : 0;
    float dh = dy[((long)t * B + bb) * H2 + dir * H + j];
    const float* gp0 = G + ((long)t * B + bb) * K4 + j;
    const float gi = gp0[0], gf = gp0[H], gg = gp0[2 * H], go = gp0[3 * H];
    const float c = cbuf[((long)(t + 1) * B + bb) * H2 + dir * H + j];
    const float cp = cbuf[((long)prev_blk * B + bb) * H2 + dir * H + j];
    const float dc = dc_state[bb * H2 + dir * H + j];
    const int ln = lens[bb];
    if (s > 0) {
      const float* q = sl_rd + (long)x * NX * 256 + tid;
      float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
      int xx = 0;
      for (; xx + 8 <= NX; xx += 8) {
        float v0 = q[(long)(xx + 0) * 256], v1 = q[(long)(xx + 1) * 256], v2 = q[(long)(xx + 2) * 256], v3 = q[(long)(xx + 3) * 256];
        float v4 = q[(long)(xx + 4) * 256], v5 = q[(long)(xx + 5) * 256], v6 = q[(long)(xx + 6) * 256], v7 = q[(long)(xx + 7) * 256];
        a0 += v0 + v4; a1 += v1 + v5; a2 += v2 + v6; a3 += v3 + v7;
      }
      for (; xx < NX; ++xx) a0 += q[(long)xx * 256];
      dh += (a0 + a1) + (a2 + a3);
    }
    float di = 0.f, df = 0.f, dg = 0.f, dout = 0.f, dcp = dc;
    if (ok && t < ln) {
      float tc = tanhf_(c);
      float dct = dh * go * (1.f - tc * tc) + dc;
      dout = dh * tc * go * (1.f - go);
      di = dct * gg * gi * (1.f - gi);
      df = dct * cp * gf * (1.f - gf);
      dg = dct * gi * (1.f - gg * gg);
      dcp = dct * gf;
    }
    if (ok) {
      float* gp = G + ((long)t * B + b) * K4 + j;
      gp[0] = di; gp[H] = df; gp[2 * H] = dg; gp[3 * H] = dout;
      dc_state[(long)b * H2 + dir * H + j] = dcp;
    }
    dgs[bm * 36 + jj] = di; dgs[bm * 36 + 8 + jj] = df; dgs[bm * 36 + 16 + jj] = dg; dgs[bm * 36 + 24 + jj] = dout;
  }
  __syncthreads();
  if (s == T - 1) return;                        // nothing consumes the last partials
  // ---- produce: P_x = dG_x . W_hh[n in x][:] ; waves take N tiles round-robin ----
  f32x4 a4[4];
#pragma unroll
  for (int Q = 0; Q < 4; ++Q) a4[Q] = *reinterpret_cast<const f32x4*>(dgs + lr * 36 + 8 * Q + 4 * lh);
  const int njt = (H + 31) / 32;
  const f32x4* wp = reinterpret_cast<const f32x4*>(wb) + ((long)(dir * NX + x) * njt) * 256 + lane;
  for (int jt0 = wid * JT; jt0 < njt; jt0 += 4 * JT) {
    f32x4 b4[JT][4];
    f32x16 acc[JT];
#pragma unroll
    for (int u = 0; u < JT; ++u) {
      const int jt = jt0 + u < njt ? jt0 + u : njt - 1;
#pragma unroll
      for (int Q = 0; Q < 4; ++Q) b4[u][Q] = wp[(long)(jt * 4 + Q) * 64];
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[u][r] = 0.f;
    }
#pragma unroll
    for (int Q = 0; Q < 4; ++Q)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int u = 0; u < JT; ++u) acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[Q][i], b4[u][Q][i], acc[u], 0, 0, 0);
#pragma unroll
    for (int u = 0; u < JT; ++u) {
      const int jt = jt0 + u;
      if (jt < njt && 32 * jt + lr < H) {
        // column j = 32 jt + lr -> consumer x'' = 4 jt + (lr >> 3), slot lr & 7 ; row b = (r&3) + 8 (r>>2) + 4 lh
        float* dst = sl_wr + ((long)(4 * jt + (lr >> 3)) * NX + x) * 256 + (lr & 7);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
          dst[row * 8] = acc[u][r];
        }
      }
    }
  }
}

// ---- persistent backward: the whole BPTT sequence in ONE launch -------------------------------------------------------
// Same K-split decomposition as lstm_bwd_step (workgroup x = 32 gate rows = 8 hidden units x 4 gates for 32 utterances;
// consume the sum of everybody's partials for its own 8 units, cell backward, produce its partial dG_x . W_hh[x,:] for ALL
// units), but resident for all T steps: W_hh[x,:] lives in registers, dc in a register of the thread that owns it, the
// next step's operands (gates, dy, c) are prefetched one step ahead.  The partials are 32 KB per workgroup and step, so
// they are handed over in the flagged form (MI355X_MICROARCH.md hand-off "R1"): 16-byte write-through (sc1) stores,
// every storing wave drains (s_waitcnt vmcnt(0)), workgroup barrier, ONE lane publishes the step tag in the workgroup's
// flag word (own 128-byte line); the consumer's first wave polls the NX flags (one lane each, sc1), workgroup barrier,
// then every load of the partials is an sc1 load.  Two parity buffers; flags zeroed per call; bounded spins.
// (Round 3, measured and rejected: the partials as {value, step tag} granules like the forward's h_t -- no drain, barrier or flag word,
// one store-to-load round trip per step instead of two.  Unlike h_t, which every consumer reads from the same 64 KB, the partials are an
// all-to-all: doubling their bytes costs more than the saved round trip -- 5.2 -> 7.7 us per step at H = 256, 10.0 -> 16.7 at H = 512.)
typedef __attribute__((address_space(1))) unsigned gu32;

__device__ __forceinline__ void store16_sc1(float* p, const f32x4& v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}

template <int TPW, int UW>       // TPW: 32-unit output tiles per wave (4 waves, H <= 128 TPW); UW: 8-unit groups per workgroup
__global__ __launch_bounds__(256) void lstm_bwd_persist(float* g_f, float* g_r, const float* __restrict__ wb, const float* __restrict__ dy,
                                                        const float* __restrict__ cbuf, float* dc_state, float* xbuf, unsigned* flags_,
                                                        unsigned* err, const int* __restrict__ lens, int T, int B, int H) {
  // UW = 2 (wide layers): a workgroup owns 16 units = 64 gate rows, so half as many workgroups exchange partial slabs of the
  // same size -- the slab traffic through the fabric (what bounds H = 512) halves, the MFMA work per workgroup doubles.
  constexpr int UN = 8 * UW;              // hidden units per workgroup
  constexpr int KG = 32 * UW;             // gate rows per workgroup = K of the partial product
  constexpr int LDG = KG + 4;             // padded row of the d(gates) tile
  constexpr int BLK = 256 * UW;           // floats per (consumer, producer) block: [UN units][32 b]
  __shared__ __attribute__((aligned(16))) float dgs[32 * LDG];    // d(gates) tile [b][n]
  __shared__ __attribute__((aligned(16))) float psum[4 * 256];    // per-wave sums of the producers' partial blocks
  __shared__ int aborted;
  const int dir = blockIdx.z, mt = blockIdx.y, x = blockIdx.x, MT = gridDim.y, NX = gridDim.x;
  float* G = dir ? g_r : g_f;
  const int j0 = x * UN, b0 = mt * 32;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, lr = lane & 31, lh = lane >> 5;
  const int K4 = 4 * H;
  const long H2 = 2L * H;
  const int njt = (H + 31) / 32;
  if (tid == 0) aborted = 0;
  // W_hh rows of this workgroup, tiles of this wave: registers for the whole sequence
  f32x4 wreg[TPW][4 * UW];
  {
    const f32x4* wp = reinterpret_cast<const f32x4*>(wb) + ((long)(dir * NX + x) * njt) * (4 * UW) * 64 + lane;
#pragma unroll
    for (int u = 0; u < TPW; ++u) {
      const int jt = min(wid + 4 * u, njt - 1);
#pragma unroll
      for (int Q = 0; Q < 4 * UW; ++Q) wreg[u][Q] = wp[(long)(jt * 4 * UW + Q) * 64];
    }
  }
  const long grp = (long)(dir * MT + mt);
  const long x_par = (long)2 * MT * NX * NX * BLK;             // floats per parity buffer: [dir][mt][consumer][producer][UN j][32 b]
  float* xg = xbuf + grp * NX * NX * BLK;
  const __amdgpu_buffer_rsrc_t x_rs = __builtin_amdgcn_make_buffer_rsrc(xbuf, 0, (int)(2 * x_par * 4), 0x00020000);
  gu32* flags = (gu32*)flags_;
  const long f_par = (long)2 * MT * NX * 32;                    // words per parity: one 128-byte line per producer
  const long f_grp = grp * NX * 32;
  // cell-backward ownership: thread = (utterance bm, unit jj of each of the UW groups)
  const int bm = tid >> 3, jj = tid & 7, b = b0 + bm;
  const bool ok = b < B;
  const long bb = ok ? b : 0;
  const int ln = lens[bb];
  float dcr[UW];                                                // d(cell state) carried across steps
  float n_dy[UW], n_g[UW][4], n_c[UW], n_cp[UW];
#pragma unroll
  for (int u = 0; u < UW; ++u) {                                // operands of the first step
    const int j = j0 + 8 * u + jj;
    const int t = dir ? 0 : T - 1;
    const int prev_blk = dir ? t + 2 : t;
    dcr[u] = 0.f;
    n_dy[u] = dy[((long)t * B + bb) * H2 + dir * H + j];
    const float* gp0 = G + ((long)t * B + bb) * K4 + j;
#pragma unroll
    for (int g = 0; g < 4; ++g) n_g[u][g] = gp0[g * H];
    n_c[u] = cbuf[((long)(t + 1) * B + bb) * H2 + dir * H + j];
    n_cp[u] = cbuf[((long)prev_blk * B + bb) * H2 + dir * H + j];
  }
  LSTM_STAMP_DECL;
  __syncthreads();
  for (int s = 0; s < T; ++s) {
    const int t = dir ? s : T - 1 - s;
    LSTM_STAMP(0); LSTM_STAMP(10);
    float dh[UW], gi[UW], gf[UW], gg[UW], go[UW], c[UW], cp[UW];
#pragma unroll
    for (int u = 0; u < UW; ++u) {
      dh[u] = n_dy[u]; gi[u] = n_g[u][0]; gf[u] = n_g[u][1]; gg[u] = n_g[u][2]; go[u] = n_g[u][3]; c[u] = n_c[u]; cp[u] = n_cp[u];
    }
    if (s + 1 < T) {                               // next step's operands: in flight during this step
      const int tn = dir ? t + 1 : t - 1;
      const int pbn = dir ? tn + 2 : tn;
#pragma unroll
      for (int u = 0; u < UW; ++u) {
        const int j = j0 + 8 * u + jj;
        n_dy[u] = dy[((long)tn * B + bb) * H2 + dir * H + j];
        const float* gpn = G + ((long)tn * B + bb) * K4 + j;
#pragma unroll
        for (int g = 0; g < 4; ++g) n_g[u][g] = gpn[g * H];
        n_c[u] = cbuf[((long)(tn + 1) * B + bb) * H2 + dir * H + j];
        n_cp[u] = cbuf[((long)pbn * B + bb) * H2 + dir * H + j];
      }
    }
    if (s > 0) {
      // ---- each wavefront waits for ITS producers of step s-1 (wave w sums part w % UW of producers w / UW, w / UW + 4 / UW, ...) and loads
      // their blocks: the wave that polled is the wave that loads, so no workgroup barrier stands between the flags and the data.  Of a
      // producer only the wave that computed this consumer's 32-unit tile has to have drained: its flag word is the one polled. ----
      {
        constexpr int XS = 4 / UW;                 // producer stride of a wavefront
        const int part = wid % UW, xp0 = wid / UW;
        const int nmine = (NX - xp0 + XS - 1) / XS;                                   // producers of this wave (<= 64: NX <= 128 checked by the launcher)
        const int fw = ((x * UN) >> 5) & 3;                                          // the producer wave that owns tile (x * UN) / 32
        const gu32* fl = flags + ((s & 1) ^ 1) * f_par + f_grp + (long)(lane < nmine ? xp0 + XS * lane : 0) * 32 + fw * 8;
        for (unsigned spins = 0;; ++spins) {
          const bool good = lane >= nmine || __hip_atomic_load(fl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)s;
          if (__all(good)) break;
          if (spins > kSpinLimit) { if (lane == 0) { aborted = 1; if (atomicExch(err, 1u) == 0u) atomicAdd(&g_persist_aborts, 1u); } break; }
          __builtin_amdgcn_s_sleep(1);
        }
        LSTM_STAMP(1);
        // partials of this workgroup's units from every producer: 16-byte sc1 buffer loads of [32 b][UN units] blocks (UW x 1 KB);
        // the wave sums are combined through LDS in a fixed order
        const unsigned blk_b = (unsigned)((((s & 1) ^ 1) * x_par + grp * NX * NX * BLK + (long)x * NX * BLK) * 4) + (unsigned)part * 1024u + (unsigned)lane * 16u;
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
        int xp = xp0;
        for (; xp + 7 * XS < NX; xp += 8 * XS) {
          f32x4 v[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) v[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(x_rs, blk_b + (unsigned)(xp + XS * u) * (unsigned)(BLK * 4), 0, 16));
          a += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
        }
        for (; xp < NX; xp += XS) a += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(x_rs, blk_b + (unsigned)xp * (unsigned)(BLK * 4), 0, 16));
        *reinterpret_cast<f32x4*>(psum + wid * 256 + lane * 4) = a;
      }
      LSTM_STAMP(2);
      __syncthreads();
      LSTM_STAMP(3);
      if (aborted) break;
      {
        // block float index of (utterance bm, unit u * 8 + jj): bm * UN + 8 u + jj; part p = the 1 KB it lies in, held by waves p, p + UW, ...
#pragma unroll
        for (int u = 0; u < UW; ++u) {
          const int e = bm * UN + 8 * u + jj;
          if (UW == 1) dh[u] += (psum[e] + psum[256 + e]) + (psum[512 + e] + psum[768 + e]);
          else dh[u] += psum[(e >> 8) * 256 + (e & 255)] + psum[((e >> 8) + 2) * 256 + (e & 255)];
        }
      }
    }
    // ---- cell backward ----
#pragma unroll
    for (int u = 0; u < UW; ++u) {
      const int j = j0 + 8 * u + jj;
      float di = 0.f, df = 0.f, dg = 0.f, dout = 0.f, dcp = dcr[u];
      if (ok && t < ln) {
        float tc = tanhf_(c[u]);
        float dct = dh[u] * go[u] * (1.f - tc * tc) + dcr[u];
        dout = dh[u] * tc * go[u] * (1.f - go[u]);
        di = dct * gg[u] * gi[u] * (1.f - gi[u]);
        df = dct * cp[u] * gf[u] * (1.f - gf[u]);
        dg = dct * gi[u] * (1.f - gg[u] * gg[u]);
        dcp = dct * gf[u];
      }
      dcr[u] = dcp;
      if (ok) {
        float* gp = G + ((long)t * B + b) * K4 + j;
        gp[0] = di; gp[H] = df; gp[2 * H] = dg; gp[3 * H] = dout;
      }
      float* dq = dgs + bm * LDG + 32 * u + jj;
      dq[0] = di; dq[8] = df; dq[16] = dg; dq[24] = dout;
    }
    LSTM_STAMP(4);
    __syncthreads();
    LSTM_STAMP(5);
    if (s == T - 1) break;                         // nothing consumes the last partials
    // ---- produce: P_x^T = W_hh[n in x][:]^T . dG_x^T : the unit axis is M, the utterances are N, so a lane (utterance lr) holds 4 consecutive
    // units per accumulator quad and ONE store instruction writes a whole (consumer, producer) block [32 b][8 units] = 1 KB contiguous
    // (round 1-3 had the utterances as M: 64 separate 16-byte pieces per store instruction, 32-byte write-through fragments).
    // The stores of tile u-1 are issued between the MFMAs of tile u (a dependent chain leaves the wave ~60 idle cycles per MFMA).
    f32x4 a4[4 * UW];
#pragma unroll
    for (int Q = 0; Q < 4 * UW; ++Q) a4[Q] = *reinterpret_cast<const f32x4*>(dgs + lr * LDG + 8 * Q + 4 * lh);
    float* xw = xg + (s & 1) * x_par;
    f32x16 acc[TPW];
    auto store_quad = [&](int u, int k) {
      const int jt = wid + 4 * u;
      const int jc0 = 32 * jt + 8 * k + 4 * lh;                  // first of this lane's 4 consecutive units
      if (jt < njt && jc0 < H) {
        float* dst = xw + ((long)(jc0 / UN) * NX + x) * BLK + lr * UN + (jc0 % UN);
        f32x4 v = {acc[u][4 * k], acc[u][4 * k + 1], acc[u][4 * k + 2], acc[u][4 * k + 3]};
        store16_sc1(dst, v);
      }
    };
#pragma unroll
    for (int u = 0; u < TPW; ++u) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[u][r] = 0.f;
#pragma unroll
      for (int Q = 0; Q < 4 * UW; ++Q)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(wreg[u][Q][i], a4[Q][i], acc[u], 0, 0, 0);
          if (u > 0 && Q < 4 && i == 1) {           // one quad of the previous tile behind MFMAs 2, 6, 10, 14
            __builtin_amdgcn_sched_barrier(0);
            store_quad(u - 1, Q);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) store_quad(TPW - 1, k);
    LSTM_STAMP(6);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // this wave's stores are through: its flag word may say so (no barrier:
    LSTM_STAMP(7);                                                // a flag signals for the stores of its own wave only)
    LSTM_STAMP(8); LSTM_STAMP(11);
    if (lane == 0)
      __hip_atomic_store(flags + (s & 1) * f_par + f_grp + (long)x * 32 + wid * 8, (unsigned)(s + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
#pragma unroll
  for (int u = 0; u < UW; ++u) {
    const int j = j0 + 8 * u + jj;
    if (ok) dc_state[(long)b * H2 + dir * H + j] = dcr[u];
    if (aborted && ok) {                            // a peer never published: make the failure visible downstream
      for (int t = 0; t < T; ++t) G[((long)t * B + b) * K4 + j] = __uint_as_float(0x7fc00000u);
    }
  }
}

// ---- round 4: the persistent backward on 16-UTTERANCE tiles (lstm_bwd3) ---------------------------------------------------------
// Same K-split decomposition and flagged hand-off as lstm_bwd_persist, but a workgroup owns UN units x 16 (not 32) utterances:
// twice the workgroups, each with half the matrix work and half the bytes to publish and to pull per step -- what the stamps name
// as the step's two largest parts (profiles/r04_chain_budget.md: H = 512 / B = 64 ran 128 workgroups at 4.2 us of MFMA and 1.3 us
// of block loads per step of 9.0).  v_mfma_f32_16x16x4_f32 with the UNITS as M and the utterances as N: a lane holds 4 consecutive
// units of one utterance per tile, one store instruction writes a whole [16 b][16 units] block (1 KB contiguous).  The K order
// inside a workgroup is kappa = 4-quad-major (lane quad kq reads d(gates)[b][kq * UN + ks]), so the B operand of all UN k-steps
// is UN / 4 ds_read_b128; W_hh[x] is packed to match (pack_w_bwd3_kernel).  Two tiles are accumulated in turns (the 16x16x4
// chain has 40 cycles of dependent latency at 32 of issue); the stores of the previous pair go between the MFMAs.
__global__ void pack_w_bwd3_kernel(const float* __restrict__ whh, float* __restrict__ wb, int H, int UN) {
  // wb[(((x * (H/16) + jt) * UN + ks) * 64 + lane)] = whh[row(x, kappa) * H + 16 jt + (lane & 15)],  kappa = (lane >> 4) * UN + ks,
  // row(x, kappa) = (kappa / UN) * H + x * UN + kappa % UN   (d(gates) tile columns: gate-major, [gate][unit])
  const long tot = (long)(H / UN) * (H / 16) * UN * 64;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (long)gridDim.x * blockDim.x) {
    const int lane = (int)(e & 63); long r = e >> 6; const int ks = (int)(r % UN); r /= UN; const int jt = (int)(r % (H / 16)); const int x = (int)(r / (H / 16));
    const int kappa = (lane >> 4) * UN + ks;
    wb[e] = whh[((long)(kappa / UN) * H + (long)x * UN + kappa % UN) * H + 16 * jt + (lane & 15)];
  }
}

template <int UN, int TPW>       // UN: hidden units per workgroup (8 | 16); TPW: 16-unit output tiles per wave (H = 64 TPW)
__global__ __launch_bounds__(256) void lstm_bwd3(float* g_f, float* g_r, const float* __restrict__ wb, const float* __restrict__ dy,
                                                 const float* __restrict__ cbuf, float* dc_state, float* xbuf, unsigned* flags_, unsigned* err,
                                                 const int* __restrict__ lens, int T, int B, int H, float* __restrict__ dbp) {
  constexpr int KG = 4 * UN;              // gate rows per workgroup = K of the partial product
  constexpr int LDG = KG + 4;             // padded row of the d(gates) tile
  constexpr int BLK = 16 * UN;            // floats per (consumer, producer) block: [16 b][UN units]
  constexpr int BPI = 1024 / (BLK * 4);   // blocks one 1 KB load instruction of a wave covers (1 | 2)
  __shared__ __attribute__((aligned(16))) float dgs[16 * LDG];    // d(gates) tile [b][gate * UN + unit]
  __shared__ __attribute__((aligned(16))) float psum[4 * 256];    // per-wave sums of the producers' partial blocks
  __shared__ int aborted;
  const int dir = blockIdx.z, mt = blockIdx.y, x = blockIdx.x, MT = gridDim.y, NX = gridDim.x;
  float* G = dir ? g_r : g_f;
  const int j0 = x * UN, b0 = mt * 16;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, n = lane & 15, kq = lane >> 4;
  const int K4 = 4 * H;
  const long H2 = 2L * H;
  if (tid == 0) aborted = 0;
  // W_hh rows of this workgroup, tiles of this wave (jt = wid + 4 u): registers for the whole sequence
  float wreg[TPW][UN];
  {
    const float* wp = wb + ((long)(dir * NX + x) * (H / 16)) * UN * 64 + lane;
#pragma unroll
    for (int u = 0; u < TPW; ++u)
#pragma unroll
      for (int ks = 0; ks < UN; ++ks) wreg[u][ks] = wp[((long)(wid + 4 * u) * UN + ks) * 64];
  }
  const long grp = (long)(dir * MT + mt);
  const long x_par = (long)2 * MT * NX * NX * BLK;             // floats per parity buffer: [dir][mt][consumer][producer][16 b][UN]
  float* xg = xbuf + grp * NX * NX * BLK;
  const __amdgpu_buffer_rsrc_t x_rs = __builtin_amdgcn_make_buffer_rsrc(xbuf, 0, (int)(2 * x_par * 4), 0x00020000);
  gu32* flags = (gu32*)flags_;
  const long f_par = (long)2 * MT * NX * 32;                    // words per parity: one 128-byte line per producer, a word per wave at 32 B
  const long f_grp = grp * NX * 32;
  // cell-backward ownership: thread = (utterance bm, unit jj), 16 * UN threads
  const bool cell = tid < 16 * UN;
  const int bm = cell ? tid / UN : 0, jj = tid % UN, b = b0 + bm, j = j0 + jj;
  const bool ok = cell && b < B;
  const long bb = ok ? b : 0;
  const int ln = lens[bb];
  float dcr = 0.f;                                              // d(cell state) carried across steps
  // the bias gradient rides along (round 6): sum over the steps of this thread's four d(gates) -- the column sums of d(gates) the caller used to
  // take in a pass of its own over the (T B, 4H) tensor (105 MB per direction and layer at config 4); summed over the tile's utterances at the end
  float sbi = 0.f, sbf = 0.f, sbg = 0.f, sbo = 0.f;
  float n_dy, n_g[4], n_c, n_cp;
  {
    const int t = dir ? 0 : T - 1;
    const int prev_blk = dir ? t + 2 : t;
    n_dy = dy[((long)t * B + bb) * H2 + dir * H + j];
    const float* gp0 = G + ((long)t * B + bb) * K4 + j;
#pragma unroll
    for (int g = 0; g < 4; ++g) n_g[g] = gp0[g * H];
    n_c = cbuf[((long)(t + 1) * B + bb) * H2 + dir * H + j];
    n_cp = cbuf[((long)prev_blk * B + bb) * H2 + dir * H + j];
  }
  // consumer side: wave w sums producers w, w + 4, ...; with 512-byte blocks one load instruction covers two of them (lanes >= 32: the next)
  const int xsub = BPI == 2 ? (lane >> 5) : 0;                   // which of the instruction's blocks this lane reads
  const int nprod = (NX - wid + 3) / 4;                          // producers of this wave
  const int fw = ((x * UN) >> 4) & 3;                            // the producer wave that owns this consumer's 16-unit tile
  LSTM_STAMP_DECL;
  __syncthreads();
  for (int s = 0; s < T; ++s) {
    const int t = dir ? s : T - 1 - s;
    LSTM_STAMP(0); LSTM_STAMP(10);
    float dh = n_dy;
    const float gi = n_g[0], gf = n_g[1], gg = n_g[2], go = n_g[3], c = n_c, cp = n_cp;
    if (s + 1 < T) {                               // next step's operands: in flight during this step
      const int tn = dir ? t + 1 : t - 1;
      const int pbn = dir ? tn + 2 : tn;
      n_dy = dy[((long)tn * B + bb) * H2 + dir * H + j];
      const float* gpn = G + ((long)tn * B + bb) * K4 + j;
#pragma unroll
      for (int g = 0; g < 4; ++g) n_g[g] = gpn[g * H];
      n_c = cbuf[((long)(tn + 1) * B + bb) * H2 + dir * H + j];
      n_cp = cbuf[((long)pbn * B + bb) * H2 + dir * H + j];
    }
    if (s > 0) {
      {
        const gu32* fl = flags + ((s & 1) ^ 1) * f_par + f_grp + (long)(lane < nprod ? wid + 4 * lane : 0) * 32 + fw * 8;
        for (unsigned spins = 0;; ++spins) {
          const bool good = lane >= nprod || __hip_atomic_load(fl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)s;
          if (__all(good)) break;
          if (spins > kSpinLimit) { if (lane == 0) { aborted = 1; if (atomicExch(err, 1u) == 0u) atomicAdd(&g_persist_aborts, 1u); } break; }
          __builtin_amdgcn_s_sleep(1);
        }
        LSTM_STAMP(1);
        // this consumer's blocks: [x][producer][16 b][UN]; lane reads 16 bytes of block (wid + 4 (BPI i + xsub))
        const unsigned blk_b = (unsigned)((((s & 1) ^ 1) * x_par + grp * NX * NX * BLK + (long)x * NX * BLK) * 4) + (unsigned)(lane & (64 / BPI - 1)) * 16u;
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
        const int ninstr = (nprod + BPI - 1) / BPI;
        auto ldblk = [&](int i) {
          const int xp = wid + 4 * (BPI * i + xsub);
          return xp < NX ? __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(x_rs, blk_b + (unsigned)xp * (unsigned)(BLK * 4), 0, 16)) : f32x4{0.f, 0.f, 0.f, 0.f};
        };
        int i = 0;                                 // every batch issues all its loads before the first add (fixed order: deterministic)
        for (; i + 8 <= ninstr; i += 8) {
          f32x4 v[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) v[u] = ldblk(i + u);
          a += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
        }
        if (i + 4 <= ninstr) {
          f32x4 v[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) v[u] = ldblk(i + u);
          a += (v[0] + v[1]) + (v[2] + v[3]);
          i += 4;
        }
        if (i + 2 <= ninstr) {
          const f32x4 v0 = ldblk(i), v1 = ldblk(i + 1);
          a += v0 + v1;
          i += 2;
        }
        if (i < ninstr) a += ldblk(i);
        *reinterpret_cast<f32x4*>(psum + wid * 256 + lane * 4) = a;
      }
      LSTM_STAMP(2);
      __syncthreads();
      LSTM_STAMP(3);
      if (aborted) break;
      if (cell) {
        const int e = bm * UN + jj;               // float index inside a block
        if (BPI == 1) dh += (psum[e] + psum[256 + e]) + (psum[512 + e] + psum[768 + e]);
        else dh += ((psum[e] + psum[128 + e]) + (psum[256 + e] + psum[384 + e])) + ((psum[512 + e] + psum[640 + e]) + (psum[768 + e] + psum[896 + e]));
      }
    }
    // ---- cell backward ----
    if (cell) {
      float di = 0.f, df = 0.f, dg = 0.f, dout = 0.f, dcp = dcr;
      if (ok && t < ln) {
        float tc = tanhf_(c);
        float dct = dh * go * (1.f - tc * tc) + dcr;
        dout = dh * tc * go * (1.f - go);
        di = dct * gg * gi * (1.f - gi);
        df = dct * cp * gf * (1.f - gf);
        dg = dct * gi * (1.f - gg * gg);
        dcp = dct * gf;
      }
      dcr = dcp;
      sbi += di; sbf += df; sbg += dg; sbo += dout;
      if (ok) {
        float* gp = G + ((long)t * B + b) * K4 + j;
        gp[0] = di; gp[H] = df; gp[2 * H] = dg; gp[3 * H] = dout;
      }
      float* dq = dgs + bm * LDG + jj;
      dq[0] = di; dq[UN] = df; dq[2 * UN] = dg; dq[3 * UN] = dout;
    }
    LSTM_STAMP(4);
    __syncthreads();
    LSTM_STAMP(5);
    if (s == T - 1) break;                         // nothing consumes the last partials
    // ---- produce: P_x^T = W_hh[rows of x][:]^T . dG_x^T, tile pairs, stores of the previous pair between the MFMAs ----
    float bv[UN];
#pragma unroll
    for (int q = 0; q < UN / 4; ++q) {
      const f32x4 t4 = *reinterpret_cast<const f32x4*>(dgs + n * LDG + kq * UN + 4 * q);
      bv[4 * q] = t4[0]; bv[4 * q + 1] = t4[1]; bv[4 * q + 2] = t4[2]; bv[4 * q + 3] = t4[3];
    }
    float* xw = xg + (s & 1) * x_par;
    f32x4 acc[TPW];
    auto store_tile = [&](int u) {
      const int jc0 = 16 * (wid + 4 * u) + 4 * kq;               // first of this lane's 4 consecutive units
      float* dst = xw + ((long)(jc0 / UN) * NX + x) * BLK + n * UN + (jc0 % UN);
      store16_sc1(dst, acc[u]);
    };
#pragma unroll
    for (int u0 = 0; u0 < TPW; u0 += 2) {
      acc[u0] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (u0 + 1 < TPW) acc[u0 + 1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < UN; ++ks) {
        acc[u0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[u0][ks], bv[ks], acc[u0], 0, 0, 0);
        if (u0 + 1 < TPW) acc[u0 + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[u0 + 1][ks], bv[ks], acc[u0 + 1], 0, 0, 0);
        if (u0 >= 2 && (ks == 1 || ks == 3)) {
          __builtin_amdgcn_sched_barrier(0);
          store_tile(u0 - 2 + (ks >> 1));
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    if (TPW % 2 == 0) store_tile(TPW - 2);       // (an odd count's last pair was stored beside the single last tile)
    store_tile(TPW - 1);
    LSTM_STAMP(6);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // this wave's stores are through: its flag word may say so
    LSTM_STAMP(7); LSTM_STAMP(8); LSTM_STAMP(11);
    if (lane == 0)
      __hip_atomic_store(flags + (s & 1) * f_par + f_grp + (long)x * 32 + wid * 8, (unsigned)(s + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (ok) dc_state[(long)b * H2 + dir * H + j] = dcr;
  if (dbp) {
    // dbp[dir][mt][gate * H + unit] = sum over this tile's 16 utterances (fixed order) of the per-thread sums over time
    __syncthreads();
    if (cell) {
      float* dq = dgs + bm * LDG + jj;
      dq[0] = sbi; dq[UN] = sbf; dq[2 * UN] = sbg; dq[3 * UN] = sbo;
    }
    __syncthreads();
    if (tid < KG) {
      float v = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) v += dgs[r * LDG + tid];
      dbp[(grp * 4 + tid / UN) * H + j0 + tid % UN] = v;
    }
  }
  if (aborted && ok) {                              // a peer never published: make the failure visible downstream
    for (int t = 0; t < T; ++t) G[((long)t * B + b) * K4 + j] = __uint_as_float(0x7fc00000u);
  }
}

// out[dir][c] = sum over the utterance tiles (fixed order) of lstm_bwd3's partial bias sums dbp[dir][mt][c], c < 4H
__global__ void lstm_dbias_reduce_kernel(const float* __restrict__ dbp, int MT, int H4, float* __restrict__ out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x, dir = blockIdx.y;
  if (c >= H4) return;
  float v = 0.f;
  for (int mt = 0; mt < MT; ++mt) v += dbp[((long)dir * MT + mt) * H4 + c];
  out[(long)dir * H4 + c] = v;
}
// the same sums for the paths without the fused form (launch-per-step kernels, the round-1..3 persistent backward): column sums of d(gates) over
// all T B rows, one workgroup per 64 columns and direction, rows in a fixed order (deterministic; these paths are not the training step's)
__global__ __launch_bounds__(256) void gates_colsum_kernel(const float* __restrict__ g_f, const float* __restrict__ g_r, long M, int H4, float* __restrict__ out) {
  __shared__ float red[4][64];
  const int dir = blockIdx.y, c = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
  const float* G = dir ? g_r : g_f;
  float v = 0.f;
  if (c < H4)
    for (long r = rl; r < M; r += 4) v += G[r * H4 + c];
  red[rl][threadIdx.x & 63] = v;
  __syncthreads();
  if (rl == 0 && c < H4) out[(long)dir * H4 + c] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

__global__ void zero_kernel(float* p, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = 0.f;
}

int pick_waves(int K, int min_kc, const char* env = nullptr) {
  if (env) {                       // tuning override (tools/bench_lstm.py): RE2E_LSTM_WAVES_{FWD,BWD}
    const char* v = exp_env(env);
    if (v) { int w = atoi(v); if (w >= 1 && w <= 16 && (w & (w - 1)) == 0 && K % (8 * w) == 0) return w; }
  }
  for (int w = 16; w >= 1; w >>= 1)
    if (K % (8 * w) == 0 && K / w >= min_kc) return w;
  return K % 8 == 0 ? 1 : 0;
}

// workspace (floats): fwd  = wfrag[2][4H*H] | hfrag[2][2][MT][H*32]
//                     bwd  = wtfrag[2][nx*(H/2)*256] | gfrag[2][2][MT][4H*32]
size_t fwd_ws_floats(int B, int H) { long MT = (B + 31) / 32; return (size_t)2 * 4 * H * H + (size_t)2 * 2 * MT * H * 32; }
// persistent forward: + hx header (16 B: error word) + granules [2][2][MT][H/8][256] x 8 B
// (the round-4 form needs [2][2][cdiv(B, 16)][H / 4][16] x 16 B, at most half of this)
size_t fwd_hx_bytes(int B, int H) { long MT = (B + 31) / 32; return 16 + (size_t)2 * 2 * MT * (H / 8) * 256 * 8; }
size_t bwd_ws_floats(int B, int H) {
  long MT = (B + 31) / 32, NX = H / 8, njt = (H + 31) / 32;
  return (size_t)2 * NX * njt * 1024 + (size_t)2 * 2 * MT * NX * NX * 256;
}
// persistent backward: + header (16 B: error word) + flag lines [2][2][MT][NX] x 128 B
// (the 16-utterance-tile form has cdiv(B, 16) tiles of up to H / 8 producers)
size_t bwd_flag_bytes(int B, int H) { long MT = (B + 15) / 16, NX = H / 8; return 16 + (size_t)2 * 2 * MT * NX * 128; }

template <int W>
void launch_fwd(hipStream_t st, float* xg_f, float* xg_r, const float* wfrag, float* ybuf, float* cbuf, float* hfrag, const int* lens,
                int T, int B, int H) {
  size_t lds = (size_t)W * 32 * 33 * sizeof(float);
  static LdsLimit lim;
  lim.ensure(reinterpret_cast<const void*>(&lstm_fwd_step<W>), lds);
  dim3 grid(H / 8, cdiv(B, 32), 2);
  for (int s = 0; s < T; ++s)
    hipLaunchKernelGGL((lstm_fwd_step<W>), grid, dim3(W * 64), lds, st, xg_f, xg_r, wfrag, ybuf, cbuf, hfrag, lens, T, B, H, s);
}
template <int JT>
void launch_bwd(hipStream_t st, float* g_f, float* g_r, const float* wb, const float* dy, const float* cbuf, float* dc, float* slabs,
                const int* lens, int T, int B, int H) {
  dim3 grid(H / 8, cdiv(B, 32), 2);
  for (int s = 0; s < T; ++s)
    hipLaunchKernelGGL((lstm_bwd_step<JT>), grid, dim3(256), 0, st, g_f, g_r, wb, dy, cbuf, dc, slabs, lens, T, B, H, s);
}

int cu_count() {
  static const int n = [] { int dev = 0; hipDeviceProp_t p; return (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) ? p.multiProcessorCount : 1; }();
  return n;
}

// dynamic-LDS limits of the persistent kernels, one per instantiation (raised on first use, or all at once by re2e_warmup)
template <int W, int QN, int UW> LdsLimit& fwd_lim() { static LdsLimit l; return l; }
template <int TPW, int UW> LdsLimit& bwd_lim() { static LdsLimit l; return l; }
constexpr size_t kOwnCuLds = 160 * 1024;

template <int W, int QN, int UW = 1>
bool launch_fwd_persist(hipStream_t st, float* xg_f, float* xg_r, const float* wfrag, float* ybuf, float* cbuf, void* hxmem, size_t hxbytes,
                        const int* lens, int T, int B, int H) {
  size_t lds = (size_t)UW * W * 32 * 33 * sizeof(float) + 16;
  dim3 grid(H / (8 * UW), cdiv(B, 32), 2);
  if ((long)grid.x * grid.y * grid.z > cu_count()) return false;          // every workgroup must be resident
  // A chain of up to HALF the chip's CUs (RE2E_LSTM_OWN_CU_FRAC = 2 since the end of round 4; rounds 2-3 measured whole = half within 0.2 ms and
  // shipped whole, with the decoder loop persistent the 256-workgroup chains of the 512-wide layers are better off NOT excluding the
  // weight-gradient stream from the chip they only use 5-25 % of: 52.1 -> 51.4 ms, profiles/r04_own_cu_ab.txt; a quarter: 54.3)
  // asks for (almost) a whole CU's LDS per workgroup: nothing else can then be
  // co-resident on its CUs, so its MFMA pipe and memory queue are its own while the filler streams keep the other CUs.
  // (RE2E_LSTM_OWN_CU=0 turns it off, =n asks for n KB -- it has to be the whole CU: with 120 KB, which still admits a small filler
  // workgroup, the step is 74.2 instead of 71.8 ms.  Step 77.9 -> 75.2 ms when introduced: enhancer forward 13.6 -> 9.6, backward 16.7 -> 13.0 ms.)
  static const int hog = exp_env("RE2E_LSTM_OWN_CU") ? atoi(exp_env("RE2E_LSTM_OWN_CU")) : 160;
  static const int frac = exp_env("RE2E_LSTM_OWN_CU_FRAC") ? atoi(exp_env("RE2E_LSTM_OWN_CU_FRAC")) : 2;
  if (hog && (long)grid.x * grid.y * grid.z <= cu_count() / frac) lds = (size_t)hog * 1024;
  fwd_lim<W, QN, UW>().ensure(reinterpret_cast<const void*>(&lstm_fwd_persist<W, QN, UW>), lds);
  lstm_stamps_arm();
  (void)hipMemsetAsync(hxmem, 0, hxbytes, st);                             // tags and the error word start at zero, every call
  unsigned* err = (unsigned*)hxmem;
  u64* hx = (u64*)((char*)hxmem + 16);
  hipLaunchKernelGGL((lstm_fwd_persist<W, QN, UW>), grid, dim3(W * 64), lds, st, xg_f, xg_r, wfrag, ybuf, cbuf, hx, err, lens, T, B, H);
  return true;
}

template <int TPW, int UW>
bool launch_bwd_persist(hipStream_t st, float* g_f, float* g_r, const float* wb, const float* dy, const float* cbuf, float* dc, float* slabs,
                        void* flagmem, size_t flagbytes, const int* lens, int T, int B, int H) {
  dim3 grid(H / (8 * UW), cdiv(B, 32), 2);
  lstm_stamps_arm();
  (void)hipMemsetAsync(flagmem, 0, flagbytes, st);
  unsigned* err = (unsigned*)flagmem;
  unsigned* flags = (unsigned*)((char*)flagmem + 16);
  static const int hog = exp_env("RE2E_LSTM_OWN_CU") ? atoi(exp_env("RE2E_LSTM_OWN_CU")) : 160;      // see launch_fwd_persist
  size_t lds = 0;
  static const int frac = exp_env("RE2E_LSTM_OWN_CU_FRAC") ? atoi(exp_env("RE2E_LSTM_OWN_CU_FRAC")) : 2;
  if (hog && (long)grid.x * grid.y * grid.z <= cu_count() / frac) lds = (size_t)(hog - 16) * 1024;       // + ~13 KB static
  bwd_lim<TPW, UW>().ensure(reinterpret_cast<const void*>(&lstm_bwd_persist<TPW, UW>), lds);
  hipLaunchKernelGGL((lstm_bwd_persist<TPW, UW>), grid, dim3(256), lds, st, g_f, g_r, wb, dy, cbuf, dc, slabs, flags, err, lens, T, B, H);
  return true;
}

// ---- round-4 forward (lstm_fwd2): configuration, launch of the persistent kernel and of its launch-per-step twin ----------
struct Fwd2Cfg { int tiles, nj, tt; };      // tt: utterance tiles per workgroup (2: the interleaved form, persistent launches only)
template <int TILES, int NJ, bool P, int TT = 1> LdsLimit& fwd2_lim() { static LdsLimit l; return l; }

// RE2E_LSTM_FWD2=0 keeps the round-1..3 forward kernels (also used for widths this form is not instantiated for: H % 64 != 0)
bool fwd2_config(int B, int H, const float* whh_f, const float* whh_r, Fwd2Cfg& c) {
  const char* v = getenv("RE2E_LSTM_FWD2");
  if (v && atoi(v) == 0) return false;
  if (exp_env("RE2E_LSTM_FWD2_MAXH") && H > atoi(exp_env("RE2E_LSTM_FWD2_MAXH"))) return false;
  if (exp_env("RE2E_LSTM_FWD2_MINH") && H < atoi(exp_env("RE2E_LSTM_FWD2_MINH"))) return false;
  if ((reinterpret_cast<uintptr_t>(whh_f) | reinterpret_cast<uintptr_t>(whh_r)) & 15) return false;
  if (H % 64 != 0) return false;
  // Where it runs.  In the training step a chain is paid for in CUs x time (it owns its CUs; behind the forward half the step is bound
  // by the sum of all streams' work): at H = 256 / B = 32 this form is faster alone (2.9 against 3.2 us per step) on TWICE the CUs
  // (128 against 64) and the step loses 1.7 ms with it (57.4 against 55.7, two A/B rounds); on as many CUs (64: 16 units per workgroup)
  // it is slower than the round-1..3 kernel (3.7).  It is the better kernel where that one wastes its tile or is bound by its matrix
  // work: <= 16 utterances (half of a 32-row tile empty: 2.5 against 3.2 us, same CUs) and wide layers (H = 512, B = 64: 5.2 against 6.2).
  // Round 6, built and measured, NOT selected (RE2E_LSTM_FWD2_TT=2 in the experiments build turns it on): TWO utterance tiles per workgroup
  // (lstm_fwd2<.., TT = 2>), a workgroup working on one tile while the other's h(t) is on its way, so that B = 32 / H = 256 would keep this form's
  // 2.9 us per step on the round-1..3 kernel's 64 CUs.  Bitwise equal to the one-tile form -- and 5.14 us per step against 3.22 (the step 51.4
  // against 47.7 ms, profiles/r06_chain_two_tiles.txt): vmcnt retires in order and counts stores, so the finishing waves' poll of tile B cannot
  // return before the write-through (sc1) publish of tile A and its output stores are acknowledged by the memory side (~0.7 us) -- the hand-off
  // latency the interleave was to hide is paid once per TILE instead of once per step.
  c.nj = H / 64;
  c.tt = 1;
  static const int tt_env = exp_env("RE2E_LSTM_FWD2_TT") ? atoi(exp_env("RE2E_LSTM_FWD2_TT")) : 1;
  const bool two = tt_env == 2 && B > 16 && H < 384 && cdiv(B, 16) % 2 == 0 && (c.nj == 2 || c.nj == 4 || c.nj == 5);
  if (!exp_env("RE2E_LSTM_FWD2_MINH") && !exp_env("RE2E_LSTM_FWD2_MAXH") && B > 16 && H < 384 && !two) return false;
  if (!(c.nj == 1 || c.nj == 2 || c.nj == 4 || c.nj == 5 || c.nj == 8)) return false;
  if (two) {
    // units per workgroup = 4 x tiles: the smallest that keeps the grid within a quarter of the chip (the chains own their CUs)
    const long groups = (long)(cdiv(B, 16) / 2) * 2;
    c.tiles = 0;
    for (int tl = 1; tl <= 4; tl *= 2)
      if (H % (4 * tl) == 0 && (long)(H / (4 * tl)) * groups <= cu_count() / 4) { c.tiles = tl; break; }
    if (c.tiles) { c.tt = 2; return true; }
    if (!exp_env("RE2E_LSTM_FWD2_MINH") && !exp_env("RE2E_LSTM_FWD2_MAXH")) return false;
  }
  // units per workgroup = 4 x tiles: the smallest that keeps the grid within half of the chip (the rest stays with the filler
  // streams), else within the chip
  const long per = (long)cdiv(B, 16) * 2;
  const int te = exp_env("RE2E_LSTM_FWD2_TILES") ? atoi(exp_env("RE2E_LSTM_FWD2_TILES")) : 0;
  c.tiles = 0;
  for (int lim = cu_count() / 2; lim <= cu_count() && !c.tiles; lim *= 2)
    for (int tl = 1; tl <= 4; tl *= 2)
      if (H % (4 * tl) == 0 && (long)(H / (4 * tl)) * per <= lim) { c.tiles = tl; break; }
  if (te == 1 || te == 2 || te == 4) c.tiles = te;
  if (!c.tiles) c.tiles = 4;                          // does not fit: only the launch-per-step twin can run it
  return true;
}

template <int TILES, int NJ, int TT = 1>
bool launch_fwd2(bool persist, hipStream_t st, float* xg_f, float* xg_r, const float* whh_f, const float* whh_r, float* ybuf, float* cbuf, void* hxmem,
                 size_t hxbytes, const int* lens, int T, int B, int H) {
  if (TT > 1 && !persist) return launch_fwd2<TILES, NJ, 1>(false, st, xg_f, xg_r, whh_f, whh_r, ybuf, cbuf, hxmem, hxbytes, lens, T, B, H);
  size_t lds = (size_t)TT * TILES * 8192 + 16;
  dim3 grid(H / (4 * TILES), cdiv(B, 16) / TT, 2);
  unsigned* err = (unsigned*)hxmem;
  unsigned* hx = (unsigned*)((char*)hxmem + 16);
  if (!persist) {
    fwd2_lim<TILES, NJ, false>().ensure(reinterpret_cast<const void*>(&lstm_fwd2<TILES, NJ, false>), lds);
    for (int s = 0; s < T; ++s)
      hipLaunchKernelGGL((lstm_fwd2<TILES, NJ, false>), grid, dim3(256), lds, st, xg_f, xg_r, whh_f, whh_r, ybuf, cbuf, hx, err, lens, T, B, H, s, 0);
    return true;
  }
  if ((long)grid.x * grid.y * grid.z > cu_count()) return false;          // every workgroup must be resident
  static const int hog = exp_env("RE2E_LSTM_OWN_CU") ? atoi(exp_env("RE2E_LSTM_OWN_CU")) : 160;      // see launch_fwd_persist
  // mode 1 (default): poll one piece per producer, then sweep once.  mode 0: sweep right behind the publish and fall back to the poll when it
  // came too early -- within the run-to-run spread of mode 1 where a sweep is small (H = 256: 3.14 / 3.43 against 3.18 / 3.17 us per step in two
  // sessions), worse where it is not (H = 512, B = 64: 6.05 against 5.39)
  static const int frac = exp_env("RE2E_LSTM_OWN_CU_FRAC") ? atoi(exp_env("RE2E_LSTM_OWN_CU_FRAC")) : 2;
  static const int mode_env = exp_env("RE2E_LSTM_FWD2_MODE") ? atoi(exp_env("RE2E_LSTM_FWD2_MODE")) : -1;
  const int mode = mode_env >= 0 ? mode_env : 1;
  if (hog && (long)grid.x * grid.y * grid.z <= cu_count() / frac) lds = (size_t)hog * 1024;
  fwd2_lim<TILES, NJ, true, TT>().ensure(reinterpret_cast<const void*>(&lstm_fwd2<TILES, NJ, true, TT>), lds);
  lstm_stamps_arm();
  (void)hipMemsetAsync(hxmem, 0, hxbytes, st);                             // tags and the error word start at zero, every call
  hipLaunchKernelGGL((lstm_fwd2<TILES, NJ, true, TT>), grid, dim3(256), lds, st, xg_f, xg_r, whh_f, whh_r, ybuf, cbuf, hx, err, lens, T, B, H, 0, mode);
  return true;
}

// RE2E_FWD2_ALL(M): M(TILES, NJ) for every instantiation
#define RE2E_FWD2_NJ(M, TL) M(TL, 1) M(TL, 2) M(TL, 4) M(TL, 5) M(TL, 8)
#define RE2E_FWD2_ALL(M) RE2E_FWD2_NJ(M, 1) RE2E_FWD2_NJ(M, 2) RE2E_FWD2_NJ(M, 4)

bool try_fwd2(const Fwd2Cfg& c, bool persist, hipStream_t st, float* xg_f, float* xg_r, const float* whh_f, const float* whh_r, float* ybuf, float* cbuf,
              void* hxmem, size_t hxbytes, const int* lens, int T, int B, int H) {
#ifdef RE2E_EXPERIMENTS
  if (c.tt == 2) {        // the interleaved form (not selected, experiments build only): instantiated for H in 128 / 256 / 320
#define RE2E_F2T(TL, NJV) \
    if (c.tiles == TL && c.nj == NJV) return launch_fwd2<TL, NJV, 2>(persist, st, xg_f, xg_r, whh_f, whh_r, ybuf, cbuf, hxmem, hxbytes, lens, T, B, H);
    RE2E_F2T(1, 2) RE2E_F2T(2, 2) RE2E_F2T(4, 2) RE2E_F2T(1, 4) RE2E_F2T(2, 4) RE2E_F2T(4, 4) RE2E_F2T(1, 5) RE2E_F2T(2, 5) RE2E_F2T(4, 5)
#undef RE2E_F2T
    return false;
  }
#else
  if (c.tt == 2) return false;
#endif
#define RE2E_F2(TL, NJV) \
  if (c.tiles == TL && c.nj == NJV) return launch_fwd2<TL, NJV>(persist, st, xg_f, xg_r, whh_f, whh_r, ybuf, cbuf, hxmem, hxbytes, lens, T, B, H);
  RE2E_FWD2_ALL(RE2E_F2)
#undef RE2E_F2
  return false;
}

// round-4 backward on 16-utterance tiles: units per workgroup (0: not applicable -> the round-1..3 kernels).  RE2E_LSTM_BWD3=0 turns it off.
int bwd3_units(int T, int B, int H) {
  const char* v = getenv("RE2E_LSTM_BWD3");
  const char* pv = getenv("RE2E_LSTM_PERSIST_BWD");
  if ((v && atoi(v) == 0) || (pv && atoi(pv) == 0) || T < 2 || H % 64 != 0 || H / 64 > 8) return 0;
  if (exp_env("RE2E_LSTM_BWD3_MAXH") && H > atoi(exp_env("RE2E_LSTM_BWD3_MAXH"))) return 0;
  if (exp_env("RE2E_LSTM_BWD3_MINH") && H < atoi(exp_env("RE2E_LSTM_BWD3_MINH"))) return 0;
  const long per = (long)cdiv(B, 16) * 2;
  const int ue = exp_env("RE2E_LSTM_BWD3_UN") ? atoi(exp_env("RE2E_LSTM_BWD3_UN")) : 0;
  if (ue == 8 || ue == 16) return (long)(H / ue) * per <= cu_count() ? ue : 0;
  // 8 units while that fits a quarter of the chip, else 16 (half the workgroups, half the slab traffic): at H = 256 / B = 32 the 128-workgroup
  // form is faster alone (3.35 against 3.66 us per step) and costs the training step 1.1 ms (57.4 against 56.3: CUs x time, see fwd2_config)
  if ((long)(H / 8) * per <= cu_count() / 4) return 8;
  if ((long)(H / 16) * per <= cu_count()) return 16;
  return 0;
}
template <int UN, int TPW> LdsLimit& bwd3_lim() { static LdsLimit l; return l; }
template <int UN, int TPW>
bool launch_bwd3(hipStream_t st, float* g_f, float* g_r, const float* wb, const float* dy, const float* cbuf, float* dc, float* slabs, void* flagmem,
                 size_t flagbytes, const int* lens, int T, int B, int H, float* dbp) {
  dim3 grid(H / UN, cdiv(B, 16), 2);
  lstm_stamps_arm();
  (void)hipMemsetAsync(flagmem, 0, flagbytes, st);
  unsigned* err = (unsigned*)flagmem;
  unsigned* flags = (unsigned*)((char*)flagmem + 16);
  static const int hog = exp_env("RE2E_LSTM_OWN_CU") ? atoi(exp_env("RE2E_LSTM_OWN_CU")) : 160;      // see launch_fwd_persist
  static const int frac = exp_env("RE2E_LSTM_OWN_CU_FRAC") ? atoi(exp_env("RE2E_LSTM_OWN_CU_FRAC")) : 2;
  size_t lds = hog && (long)grid.x * grid.y * grid.z <= cu_count() / frac ? (size_t)(hog - 16) * 1024 : 0;     // + ~9 KB static
  bwd3_lim<UN, TPW>().ensure(reinterpret_cast<const void*>(&lstm_bwd3<UN, TPW>), lds);
  hipLaunchKernelGGL((lstm_bwd3<UN, TPW>), grid, dim3(256), lds, st, g_f, g_r, wb, dy, cbuf, dc, slabs, flags, err, lens, T, B, H, dbp);
  return true;
}
#define RE2E_BWD3_ALL(M) M(8, 1) M(8, 2) M(8, 3) M(8, 4) M(8, 5) M(8, 6) M(8, 7) M(8, 8) M(16, 1) M(16, 2) M(16, 3) M(16, 4) M(16, 5) M(16, 6) M(16, 7) M(16, 8)
bool try_bwd3(int un, hipStream_t st, float* g_f, float* g_r, const float* wb, const float* dy, const float* cbuf, float* dc, float* slabs, void* flagmem,
              size_t flagbytes, const int* lens, int T, int B, int H, float* dbp) {
#define RE2E_B3(U, TP) if (un == U && H == 64 * TP) return launch_bwd3<U, TP>(st, g_f, g_r, wb, dy, cbuf, dc, slabs, flagmem, flagbytes, lens, T, B, H, dbp);
  RE2E_BWD3_ALL(RE2E_B3)
#undef RE2E_B3
  return false;
}

// 0: launch per step; 1 / 2: persistent kernel with 8 / 16 hidden units per workgroup (the weights must be packed for that width)
int bwd_persist_width(int T, int B, int H) {
  const char* v = getenv("RE2E_LSTM_PERSIST_BWD");
  const char* mh = exp_env("RE2E_LSTM_PERSIST_BWD_MAXH");
  if ((v && atoi(v) == 0) || T < 2 || H > (mh ? atoi(mh) : 1024) || (H + 31) / 32 > 16) return 0;
  const char* wv = getenv("RE2E_LSTM_BWD_UW");      // tuning override
  int uw = wv ? atoi(wv) : (H >= 512 ? 2 : 1);     // wide layers are bound by the slab traffic: half as many, twice as wide workgroups
  if (uw == 2 && H % 16 != 0) uw = 1;
  if (uw != 1 && uw != 2) uw = 1;
  if ((long)(H / (8 * uw)) * cdiv(B, 32) * 2 > cu_count() || H / (8 * uw) > 128) return 0;      // every workgroup must be resident
  return uw;
}

bool try_bwd_persist(int uw, hipStream_t st, float* g_f, float* g_r, const float* wb, const float* dy, const float* cbuf, float* dc, float* slabs,
                     void* flagmem, size_t flagbytes, const int* lens, int T, int B, int H) {
  const int njt = (H + 31) / 32, tpw = (njt + 3) / 4;
#define RE2E_BWD(TP) \
  case TP: return uw == 2 ? launch_bwd_persist<TP, 2>(st, g_f, g_r, wb, dy, cbuf, dc, slabs, flagmem, flagbytes, lens, T, B, H) \
                          : launch_bwd_persist<TP, 1>(st, g_f, g_r, wb, dy, cbuf, dc, slabs, flagmem, flagbytes, lens, T, B, H);
  switch (tpw) {
    RE2E_BWD(1) RE2E_BWD(2) RE2E_BWD(3) RE2E_BWD(4)
    default: return false;
  }
#undef RE2E_BWD
}

// RE2E_LSTM_PERSIST=0 keeps the launch-per-step form (also used for shapes the persistent kernel does not cover)
bool try_fwd_persist(hipStream_t st, float* xg_f, float* xg_r, const float* wfrag, float* ybuf, float* cbuf, void* hxmem, size_t hxbytes,
                     const int* lens, int T, int B, int H) {
  const char* v = getenv("RE2E_LSTM_PERSIST");
  if ((v && atoi(v) == 0) || T < 2) return false;
  const int NX = H / 8;
  // 512-wide layers: RE2E_LSTM_FWD_UW=2 selects 16 units per workgroup (8 wavefronts x 8 k-groups: the 16-wavefront form of it
  // needs 130 registers per lane, 2 more than 1024 threads leave)
  static const int uw = exp_env("RE2E_LSTM_FWD_UW") ? atoi(exp_env("RE2E_LSTM_FWD_UW")) : 1;
  if (NX == 64 && uw == 2) return launch_fwd_persist<8, 8, 2>(st, xg_f, xg_r, wfrag, ybuf, cbuf, hxmem, hxbytes, lens, T, B, H);
#ifdef RE2E_EXPERIMENTS
  // 256-wide layers on HALF the workgroups (32 instead of 64 owned CUs for the enhancer's forward chains): RE2E_LSTM_FWD_UW256=2
  static const int uw256 = exp_env("RE2E_LSTM_FWD_UW256") ? atoi(exp_env("RE2E_LSTM_FWD_UW256")) : 1;
  if (NX == 32 && uw256 == 2) return launch_fwd_persist<8, 4, 2>(st, xg_f, xg_r, wfrag, ybuf, cbuf, hxmem, hxbytes, lens, T, B, H);
#endif
  // 512-wide layers, 8 units per workgroup: 8 wavefronts x 8 k-groups (140 registers, 2 waves per SIMD = 288 of a SIMD's 512) rather than
  // 16 x 4 (94 registers, 4 waves per SIMD = 384).  Its 256 workgroups sit on every CU of the chip for the whole sequence, and what
  // they leave free decides which filler workgroups can be co-resident: 224 registers per SIMD admit a 4-wave engine tile (152),
  // 128 admit none of the engine's tiles.  Alone 7.4 instead of 7.0 us per step, in the training step 72.94 -> 72.69 ms (3 + 3 runs,
  // one GPU session; and again 71.73 against 71.97 with the chain owning its CUs, RE2E_LSTM_OWN_CU_FRAC = 1, where no engine workgroup
  // is co-resident any more: half as many waves sweep and meet at the barrier).  RE2E_LSTM_FWD_W8=0 selects the 16-wave form.
  static const int w8 = exp_env("RE2E_LSTM_FWD_W8") ? atoi(exp_env("RE2E_LSTM_FWD_W8")) : 1;
  if (NX == 64 && w8) return launch_fwd_persist<8, 8, 1>(st, xg_f, xg_r, wfrag, ybuf, cbuf, hxmem, hxbytes, lens, T, B, H);
#define RE2E_TRY(W, Q) if (NX == (W) * (Q)) return launch_fwd_persist<W, Q>(st, xg_f, xg_r, wfrag, ybuf, cbuf, hxmem, hxbytes, lens, T, B, H)
  RE2E_TRY(16, 4); RE2E_TRY(8, 5); RE2E_TRY(8, 4); RE2E_TRY(8, 3); RE2E_TRY(4, 4); RE2E_TRY(4, 2); RE2E_TRY(4, 1);
#undef RE2E_TRY
  return false;
}

}  // namespace

// The first launch of a persistent recurrence used to take ~28 ms (rocprofv3: max 28.5 against min 2.69 ms for the enhancer's
// forward layer): raising the kernel's dynamic-LDS limit loads its code object and reconfigures the function.  re2e_warmup does that
// for every persistent instantiation up front (no launch, no stream, idempotent), so that the first sequence of a run -- smoke(),
// short jobs -- runs at the speed of the thousandth.
// ---- test hooks -------------------------------------------------------------------------------------------------------------
namespace {
std::atomic<int> g_force_abort{0};            // host side: persistent forward sequences still to be "given up" (re2e_debug_force_abort)
// The two test hooks answer only when RE2E_DEBUG_HOOKS=1 is in the environment (tests/ set it): a shipped process cannot be told to
// poison its own sequences or to park kernels on its CUs by a stray call.
bool debug_hooks_enabled() {
  static const bool on = [] { const char* e = getenv("RE2E_DEBUG_HOOKS"); return e && atoi(e) == 1; }();
  return on;
}
__global__ void forced_abort_kernel(float* ybuf, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) ybuf[i] = __uint_as_float(0x7fc00000u);
  if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&g_persist_aborts, 1u);
}
bool forced_abort(hipStream_t st, float* ybuf, int T, int B, int H) {
  int left = g_force_abort.load(std::memory_order_relaxed);
  do {
    if (left <= 0) return false;
  } while (!g_force_abort.compare_exchange_weak(left, left - 1, std::memory_order_relaxed));
  const long n = (long)T * B * 2 * H;
  hipLaunchKernelGGL(forced_abort_kernel, dim3(1024), dim3(256), 0, st, ybuf + (long)B * 2 * H, n);
  return true;
}
__global__ __launch_bounds__(256) void occupy_kernel(unsigned long long ticks, float* sink) {
  extern __shared__ float occ_lds[];
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();          // 100 MHz
  float x = 0.f;
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) { occ_lds[threadIdx.x] = x; x += 1.f; __builtin_amdgcn_s_sleep(8); }
  if (x < 0.f) sink[0] = occ_lds[0];
}
}  // namespace

extern "C" int re2e_debug_force_abort(int n) {
  RE2E_CHECK_ARG(n >= 0, "n must be >= 0");
  if (!debug_hooks_enabled()) { re2e_set_error("re2e_debug_force_abort: test hook, needs RE2E_DEBUG_HOOKS=1 in the environment"); return RE2E_EUNSUPPORTED; }
  g_force_abort.store(n, std::memory_order_relaxed);
  return RE2E_OK;
}

extern "C" int re2e_debug_occupy(int workgroups, int lds_bytes, int usec, hipStream_t stream) {
  RE2E_CHECK_ARG(workgroups > 0 && workgroups <= 4096 && lds_bytes >= 1024 && lds_bytes <= 160 * 1024 && usec > 0 && usec <= 1000000, "bad argument");
  if (!debug_hooks_enabled()) { re2e_set_error("re2e_debug_occupy: test hook, needs RE2E_DEBUG_HOOKS=1 in the environment"); return RE2E_EUNSUPPORTED; }
  static LdsLimit lim;
  lim.ensure(reinterpret_cast<const void*>(&occupy_kernel), (size_t)lds_bytes);
  hipLaunchKernelGGL(occupy_kernel, dim3(workgroups), dim3(256), (size_t)lds_bytes, stream, (unsigned long long)usec * 100ull, (float*)nullptr);
  RE2E_LAUNCH_CHECK();
  return RE2E_OK;
}

extern "C" int re2e_warmup(void) {
#define RE2E_WF(W, Q, U) fwd_lim<W, Q, U>().ensure(reinterpret_cast<const void*>(&lstm_fwd_persist<W, Q, U>), kOwnCuLds)
  RE2E_WF(8, 8, 2); RE2E_WF(8, 8, 1); RE2E_WF(16, 4, 1); RE2E_WF(8, 5, 1); RE2E_WF(8, 4, 1); RE2E_WF(8, 3, 1); RE2E_WF(4, 4, 1); RE2E_WF(4, 2, 1);
  RE2E_WF(4, 1, 1);
#undef RE2E_WF
#define RE2E_WB(T) bwd_lim<T, 1>().ensure(reinterpret_cast<const void*>(&lstm_bwd_persist<T, 1>), kOwnCuLds - 16 * 1024); \
                   bwd_lim<T, 2>().ensure(reinterpret_cast<const void*>(&lstm_bwd_persist<T, 2>), kOwnCuLds - 16 * 1024)
  RE2E_WB(1); RE2E_WB(2); RE2E_WB(3); RE2E_WB(4);
#undef RE2E_WB
#define RE2E_W3(U, TP) bwd3_lim<U, TP>().ensure(reinterpret_cast<const void*>(&lstm_bwd3<U, TP>), kOwnCuLds - 16 * 1024);
  RE2E_BWD3_ALL(RE2E_W3)
#undef RE2E_W3
#define RE2E_W2(TL, NJV) fwd2_lim<TL, NJV, true>().ensure(reinterpret_cast<const void*>(&lstm_fwd2<TL, NJV, true>), kOwnCuLds);
  RE2E_FWD2_ALL(RE2E_W2)
#undef RE2E_W2
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { re2e_set_error("re2e_warmup: %s", hipGetErrorString(e)); return RE2E_EHIP; }
  return RE2E_OK;
}

// the backward's partial bias sums [2][cdiv(B, 16)][4H] floats, behind its flag lines
size_t bwd_dbp_bytes(int B, int H) { return (size_t)2 * ((B + 15) / 16) * 4 * H * sizeof(float); }
extern "C" size_t re2e_lstm_workspace_bytes(int B, int H) {
  size_t a = fwd_ws_floats(B, H) * sizeof(float) + fwd_hx_bytes(B, H), b = bwd_ws_floats(B, H) * sizeof(float) + bwd_flag_bytes(B, H) + bwd_dbp_bytes(B, H);
  return a > b ? a : b;
}

int re2e_dec_abort_count_();      // decloop.hip: the persistent decoder loop's give-ups are counted with the recurrences'

// ---- the trainer's step gate (joint_train.py:188-193 + the give-up protocol of the persistent kernels) ---------------------------------------
// One 1-thread kernel at the end of a training step, no host round trip:
//   delta     = give-ups counted on this device since the trainer's last acknowledgement (*base); ack != 0 first sets *base = the count
//   hold_next = 1.0 if delta == 0, NaN otherwise: the NEXT step multiplies its losses by it, so a step that was enqueued before the host has
//               repeated the aborted one computes NaN gradients (on every replica, through the gradient average) and applies nothing
//   stats_main (the ASR optimizer's [norm, coef, finite | norm, 1, finite], re2e_clip_coef): the update is also refused -- finite := 0, norm :=
//               NaN so that the host's one-step-late read-back sees it -- when delta != 0 or when extra_sumsq (the ENHANCER's gradient sum of
//               squares: a give-up in its backward chain leaves the ASR norm finite) is not finite
//   stats_d (the discriminator optimizer's triple): finite := 0 when delta != 0 (model NaNs alone do not stop D: upstream's D-step is independent),
//               or when *d_requires (optional: another gate's finite flag) is 0 -- data-parallel runs pass the main gate's flag, which every
//               replica agrees on, because delta is a per-device number
const unsigned* re2e_dec_abort_counter_ptr_();      // decloop.hip
namespace {
__global__ void step_gate_kernel(const unsigned* dec_cnt, int* base, int ack, const float* extra_sumsq, float* stats_main, float* stats_d,
                                 const float* d_requires, float* hold_next, float* delta_out) {
  const int c = (int)(__hip_atomic_load(&g_persist_aborts, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) +
                      (dec_cnt ? __hip_atomic_load(dec_cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u));
  if (ack) *base = c;
  const int delta = c - *base;
  if (delta_out) *delta_out = (float)delta;
  if (hold_next) *hold_next = delta == 0 ? 1.f : __uint_as_float(0x7fc00000u);
  bool ok = delta == 0;
  if (extra_sumsq) { const float e = *extra_sumsq; ok = ok && e == e && fabsf(e) < 3.0e38f; }
  if (stats_main && !ok) { stats_main[2] = 0.f; stats_main[5] = 0.f; stats_main[0] = __uint_as_float(0x7fc00000u); stats_main[3] = stats_main[0]; }
  if (stats_d && (delta != 0 || (d_requires && *d_requires == 0.f))) stats_d[2] = 0.f;
}
}  // namespace

extern "C" int re2e_step_gate(int* base_dev, int ack, const float* extra_sumsq, float* stats_main, float* stats_d, const float* d_requires,
                              float* hold_next, float* delta_out, hipStream_t stream) {
  RE2E_CHECK_ARG(base_dev, "null base");
  hipLaunchKernelGGL(step_gate_kernel, dim3(1), dim3(1), 0, stream, re2e_dec_abort_counter_ptr_(), base_dev, ack, extra_sumsq, stats_main, stats_d,
                     d_requires, hold_next, delta_out);
  RE2E_LAUNCH_CHECK();
  return RE2E_OK;
}

extern "C" int re2e_lstm_abort_count(void) {
  unsigned n = 0;
  if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_persist_aborts), sizeof(n)) != hipSuccess) return -1;     // (synchronises the device)
  const int d = re2e_dec_abort_count_();
  return d < 0 ? -1 : (int)n + d;
}

extern "C" int re2e_lstm_seq_fwd(float* xg_f, float* xg_r, const float* whh_f, const float* whh_r, float* ybuf, float* cbuf,
                                 const int* lens_dev, int T, int B, int H, void* workspace, size_t workspace_bytes,
                                 hipStream_t stream) {
  RE2E_CHECK_ARG(xg_f && xg_r && whh_f && whh_r && ybuf && cbuf && lens_dev && workspace, "null arg");
  RE2E_CHECK_ARG(T > 0 && B > 0 && H > 0, "bad shape");
  if (H % 8 != 0) { re2e_set_error("re2e_lstm_seq_fwd: hidden size must be a multiple of 8 (got %d)", H); return RE2E_EUNSUPPORTED; }
  RE2E_CHECK_ARG(workspace_bytes >= fwd_ws_floats(B, H) * sizeof(float) + fwd_hx_bytes(B, H), "workspace too small");
  float* wfrag = (float*)workspace;
  float* hfrag = wfrag + (size_t)2 * 4 * H * H;
  void* hxmem = (char*)workspace + fwd_ws_floats(B, H) * sizeof(float);
  long wn = (long)4 * H * H, hn = (long)2 * 2 * cdiv(B, 32) * H * 32;
  {
    const char* pv0 = getenv("RE2E_LSTM_PERSIST");                   // test hook: only where a persistent kernel would have run
    if (!(pv0 && atoi(pv0) == 0) && T >= 2 && forced_abort(stream, ybuf, T, B, H)) { RE2E_LAUNCH_CHECK(); return RE2E_OK; }
  }
  Fwd2Cfg f2;
  if (fwd2_config(B, H, whh_f, whh_r, f2)) {
    const char* pv = getenv("RE2E_LSTM_PERSIST");
    const bool want_persist = !(pv && atoi(pv) == 0) && T >= 2;
    if (want_persist && try_fwd2(f2, true, stream, xg_f, xg_r, whh_f, whh_r, ybuf, cbuf, hxmem, fwd_hx_bytes(B, H), lens_dev, T, B, H)) {
      RE2E_LAUNCH_CHECK();
      return RE2E_OK;
    }
    if (try_fwd2(f2, false, stream, xg_f, xg_r, whh_f, whh_r, ybuf, cbuf, hxmem, fwd_hx_bytes(B, H), lens_dev, T, B, H)) {
      RE2E_LAUNCH_CHECK();
      return RE2E_OK;
    }
  }
  hipLaunchKernelGGL(pack_w_fwd_kernel, dim3(cdiv(wn, 256) > 2048 ? 2048 : cdiv(wn, 256)), dim3(256), 0, stream, whh_f, wfrag, H);
  hipLaunchKernelGGL(pack_w_fwd_kernel, dim3(cdiv(wn, 256) > 2048 ? 2048 : cdiv(wn, 256)), dim3(256), 0, stream, whh_r, wfrag + wn, H);
  if (try_fwd_persist(stream, xg_f, xg_r, wfrag, ybuf, cbuf, hxmem, fwd_hx_bytes(B, H), lens_dev, T, B, H)) {
    RE2E_LAUNCH_CHECK();
    return RE2E_OK;
  }
  hipLaunchKernelGGL(zero_kernel, dim3(cdiv(hn, 256) > 1024 ? 1024 : cdiv(hn, 256)), dim3(256), 0, stream, hfrag, hn);
  int w = pick_waves(H, 32, "RE2E_LSTM_WAVES_FWD");
  switch (w) {
    case 16: launch_fwd<16>(stream, xg_f, xg_r, wfrag, ybuf, cbuf, hfrag, lens_dev, T, B, H); break;
    case 8: launch_fwd<8>(stream, xg_f, xg_r, wfrag, ybuf, cbuf, hfrag, lens_dev, T, B, H); break;
    case 4: launch_fwd<4>(stream, xg_f, xg_r, wfrag, ybuf, cbuf, hfrag, lens_dev, T, B, H); break;
    case 2: launch_fwd<2>(stream, xg_f, xg_r, wfrag, ybuf, cbuf, hfrag, lens_dev, T, B, H); break;
    default: launch_fwd<1>(stream, xg_f, xg_r, wfrag, ybuf, cbuf, hfrag, lens_dev, T, B, H); break;
  }
  RE2E_LAUNCH_CHECK();
  return RE2E_OK;
}

extern "C" int re2e_lstm_seq_bwd(float* g_f, float* g_r, const float* whh_f, const float* whh_r, const float* dy, const float* ybuf,
                                 const float* cbuf, float* dc_state, const int* lens_dev, int T, int B, int H, float* dbias, void* workspace,
                                 size_t workspace_bytes, hipStream_t stream) {
  (void)ybuf;
  RE2E_CHECK_ARG(g_f && g_r && whh_f && whh_r && dy && cbuf && dc_state && lens_dev && workspace, "null arg");
  RE2E_CHECK_ARG(T > 0 && B > 0 && H > 0, "bad shape");
  if (H % 8 != 0) { re2e_set_error("re2e_lstm_seq_bwd: hidden size must be a multiple of 8 (got %d)", H); return RE2E_EUNSUPPORTED; }
  RE2E_CHECK_ARG(workspace_bytes >= bwd_ws_floats(B, H) * sizeof(float) + bwd_flag_bytes(B, H) + (dbias ? bwd_dbp_bytes(B, H) : 0), "workspace too small");
  // dbias (optional, [2][4H]): the column sums of d(gates) per direction = the gradient of b_ih and of b_hh.  The 16-utterance-tile kernel
  // accumulates them beside the recurrence (per-tile partials behind the flag lines, summed here); the other paths take them in a pass over
  // d(gates) behind the recurrence.
  const long M_rows = (long)T * B;
  auto dbias_fallback = [&]() {
    if (dbias) hipLaunchKernelGGL(gates_colsum_kernel, dim3(cdiv(4 * H, 64), 2), dim3(256), 0, stream, g_f, g_r, M_rows, 4 * H, dbias);
  };
  long wn = (long)(H / 8) * ((H + 31) / 32) * 1024;
  float* wb = (float*)workspace;
  float* slabs = wb + 2 * wn;
  if (const int un = bwd3_units(T, B, H)) {                     // round-4 form: 16-utterance tiles (same workspace regions, its own layouts)
    const long w3 = (long)4 * H * H;                            // = wn: [x][H / 16][UN][64] floats per direction
    hipLaunchKernelGGL(pack_w_bwd3_kernel, dim3(cdiv(w3, 256) > 2048 ? 2048 : cdiv(w3, 256)), dim3(256), 0, stream, whh_f, wb, H, un);
    hipLaunchKernelGGL(pack_w_bwd3_kernel, dim3(cdiv(w3, 256) > 2048 ? 2048 : cdiv(w3, 256)), dim3(256), 0, stream, whh_r, wb + w3, H, un);
    long nz3 = (long)B * 2 * H;
    hipLaunchKernelGGL(zero_kernel, dim3(cdiv(nz3, 256)), dim3(256), 0, stream, dc_state, nz3);
    void* flagmem3 = (char*)workspace + bwd_ws_floats(B, H) * sizeof(float);
    float* dbp = dbias ? (float*)((char*)flagmem3 + bwd_flag_bytes(B, H)) : nullptr;
    if (try_bwd3(un, stream, g_f, g_r, wb, dy, cbuf, dc_state, slabs, flagmem3, bwd_flag_bytes(B, H), lens_dev, T, B, H, dbp)) {
      if (dbias) hipLaunchKernelGGL(lstm_dbias_reduce_kernel, dim3(cdiv(4 * H, 256), 2), dim3(256), 0, stream, dbp, cdiv(B, 16), 4 * H, dbias);
      RE2E_LAUNCH_CHECK();
      return RE2E_OK;
    }
  }
  const int uw = bwd_persist_width(T, B, H);
  hipLaunchKernelGGL(pack_w_bwd_kernel, dim3(cdiv(wn, 256) > 2048 ? 2048 : cdiv(wn, 256)), dim3(256), 0, stream, whh_f, wb, H, uw ? uw : 1);
  hipLaunchKernelGGL(pack_w_bwd_kernel, dim3(cdiv(wn, 256) > 2048 ? 2048 : cdiv(wn, 256)), dim3(256), 0, stream, whh_r, wb + wn, H, uw ? uw : 1);
  long nz = (long)B * 2 * H;
  hipLaunchKernelGGL(zero_kernel, dim3(cdiv(nz, 256)), dim3(256), 0, stream, dc_state, nz);
  void* flagmem = (char*)workspace + bwd_ws_floats(B, H) * sizeof(float);
  if (uw && try_bwd_persist(uw, stream, g_f, g_r, wb, dy, cbuf, dc_state, slabs, flagmem, bwd_flag_bytes(B, H), lens_dev, T, B, H)) {
    dbias_fallback();
    RE2E_LAUNCH_CHECK();
    return RE2E_OK;
  }
  if (H / 32 >= 8) launch_bwd<2>(stream, g_f, g_r, wb, dy, cbuf, dc_state, slabs, lens_dev, T, B, H);
  else launch_bwd<1>(stream, g_f, g_r, wb, dy, cbuf, dc_state, slabs, lens_dev, T, B, H);
  dbias_fallback();
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
// Decoder step, fused: gates += [ctx | z_prev] [W_ih[:, Dd:] | W_hh]^T, then the LSTMCell non-linearity (e2e_decoder.py:131) in the
// same launch -- three launches per decoder step (two skinny GEMMs + the cell kernel) become one.  A workgroup owns the four
// gates of 8 hidden units (32 gate columns) for up to 32 utterances; its 8 wavefronts split K = E + D, the partial tiles
// meet in LDS (fixed order: deterministic) and 256 threads apply the cell: activated gates back to `gates` (saved for the
// backward), c and h out.  Operand fragments come straight from global memory as 16-byte loads (E, D, ldw multiples of 4).
__global__ __launch_bounds__(512) void dec_gates_cell_kernel(const float* __restrict__ cx, const float* __restrict__ zp, const float* __restrict__ w_ctx,
                                                             long ldw, const float* __restrict__ w_hh, float* gates,
                                                             const float* __restrict__ c_prev, float* c_out, float* h_out, int B, int E, int D) {
  __shared__ float red[8][32][33];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, lr = lane & 31, lh = lane >> 5;
  const int u0 = blockIdx.x * 8, b0 = blockIdx.y * 32;
  const int g = lr >> 3, u = u0 + (lr & 7);                  // this lane's B-operand row: gate g of unit u
  const bool cok = u < D, rok = b0 + lr < B;
  const long wrow = (long)g * D + (cok ? u : 0);
  const int brow = rok ? b0 + lr : 0;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int seg = 0; seg < 2; ++seg) {
    const int K = seg ? D : E;
    const float* ap = (seg ? zp + (long)brow * D : cx + (long)brow * E);
    const float* bp = (seg ? w_hh + wrow * D : w_ctx + wrow * ldw);
    const int kq = (K + 7) / 8;
    for (int qb = wid * 4; qb < kq; qb += 32) {             // wavefront w takes k-groups 4w .. 4w+3 of every 32
      f32x4 av[4], bv[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int k = 8 * (qb + t) + 4 * lh;
        const bool kok = qb + t < kq && k + 3 < K;
        av[t] = (kok && rok) ? *reinterpret_cast<const f32x4*>(ap + k) : zero;
        bv[t] = (kok && cok) ? *reinterpret_cast<const f32x4*>(bp + k) : zero;
      }
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t][j], bv[t][j], acc, 0, 0, 0);
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) red[wid][(r & 3) + 8 * (r >> 2) + 4 * lh][lr] = acc[r];
  __syncthreads();
  if (tid < 256) {
    const int bm = tid >> 3, jj = tid & 7, b = b0 + bm, uu = u0 + jj;
    if (b < B && uu < D) {
      float* gp = gates + (long)b * 4 * D + uu;
      float v[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float a = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) a += red[w][bm][q * 8 + jj];
        v[q] = a + gp[(long)q * D];
      }
      const float gi = sigmoidf_(v[0]), gf = sigmoidf_(v[1]), gg = tanhf_(v[2]), go = sigmoidf_(v[3]);
      const float c = gf * c_prev[(long)b * D + uu] + gi * gg;
      gp[0] = gi; gp[D] = gf; gp[2L * D] = gg; gp[3L * D] = go;
      c_out[(long)b * D + uu] = c;
      h_out[(long)b * D + uu] = go * tanhf_(c);
    }
  }
}
extern "C" int re2e_dec_gates_cell_fwd(const float* cx, const float* z_prev, const float* w_ctx, long ldw, const float* w_hh, float* gates,
                                       const float* c_prev, float* c_out, float* h_out, int B, int E, int D, hipStream_t stream) {
  RE2E_CHECK_ARG(cx && z_prev && w_ctx && w_hh && gates && c_prev && c_out && h_out, "null operand");
  RE2E_CHECK_ARG(B > 0 && E > 0 && D > 0, "bad shape");
  if (E % 4 || D % 4 || ldw % 4 || ((reinterpret_cast<uintptr_t>(cx) | reinterpret_cast<uintptr_t>(z_prev) | reinterpret_cast<uintptr_t>(w_ctx) |
                                      reinterpret_cast<uintptr_t>(w_hh)) & 15)) {
    re2e_set_error("re2e_dec_gates_cell_fwd: E, D, ldw must be multiples of 4 and the operands 16-byte aligned");
    return RE2E_EUNSUPPORTED;
  }
  hipLaunchKernelGGL(dec_gates_cell_kernel, dim3(cdiv(D, 8), cdiv(B, 32)), dim3(512), 0, stream, cx, z_prev, w_ctx, ldw, w_hh, gates, c_prev, c_out,
                     h_out, B, E, D);
  RE2E_LAUNCH_CHECK();
  return RE2E_OK;
}

extern "C" int re2e_lstm_cell_fwd(float* gates, const float* c_prev, float* c_out, float* h_out, int B, int H, hipStream_t stream) {
  RE2E_CHECK_ARG(gates && c_prev && c_out && h_out && B > 0 && H > 0, "bad args");
  hipLaunchKernelGGL(lstm_cell_fwd_kernel, dim3(cdiv((long)B * H, 256)), dim3(256), 0, stream, gates, c_prev, c_out, h_out, B, H);
  RE2E_LAUNCH_CHECK();
  return RE2E_OK;
}
__global__ void lstm_cell_bwd_kernel(float* gates, const float* __restrict__ c_prev, const float* __restrict__ c_cur,
                                     const float* __restrict__ dh, const float* __restrict__ dh2, const float* __restrict__ dc_in,
                                     float* dc_prev_out, int B, int H) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * H) return;
  int b = i / H, j = i % H;
  float* g = gates + (long)b * 4 * H + j;
  float gi = g[0], gf = g[H], gg = g[2 * H], go = g[3 * H];
  float tc = tanhf_(c_cur[i]);
  float d = dh[i] + (dh2 ? dh2[i] : 0.f);       // recurrent + direct gradient of h (the decoder adds dZ[i] to the carried dz)
  float dct = d * go * (1.f - tc * tc) + (dc_in ? dc_in[i] : 0.f);
  g[0] = dct * gg * gi * (1.f - gi);
  g[H] = dct * c_prev[i] * gf * (1.f - gf);
  g[2 * H] = dct * gi * (1.f - gg * gg);
  g[3 * H] = d * tc * go * (1.f - go);
  dc_prev_out[i] = dct * gf;
}
extern "C" int re2e_lstm_cell_bwd(float* gates, const float* c_prev, const float* c_cur, const float* dh, const float* dh2,
                                  const float* dc_in, float* dc_prev_out, int B, int H, hipStream_t stream) {
  RE2E_CHECK_ARG(gates && c_prev && c_cur && dh && dc_prev_out && B > 0 && H > 0, "bad args");
  hipLaunchKernelGGL(lstm_cell_bwd_kernel, dim3(cdiv((long)B * H, 256)), dim3(256), 0, stream, gates, c_prev, c_cur, dh, dh2, dc_in,
                     dc_prev_out, B, H);
  RE2E_LAUNCH_CHECK();
  return RE2E_OK;
}

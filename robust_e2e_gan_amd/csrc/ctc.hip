// K6: CTC loss (replaces the warp-ctc call at model/e2e_ctc.py:63; warp-ctc is a third-party
// dependency that is not vendored in the reference -- restated here as the textbook CTC negative
// log-likelihood: softmax inside, blank = 0, loss = sum_b nll_b / B).
//
//   ctc_lse_gather : one wavefront per (t,b) row of the (T,B,V) logits -- max / log-sum-exp over V
//                    with wavefront shuffles, then gathers log p at the 2L+1 extended-label states
//   ctc_alpha_beta : one workgroup per utterance, alpha/beta recursions with the previous column
//                    staged in LDS; log domain
//   ctc_grad       : one wavefront per row, dlogits = scale * (softmax - occupancy)
// HBM traffic: logits are read twice (lse, grad) and dlogits written once.
#include "common.h"

namespace {
constexpr float NEG = -1.0e30f;

__device__ __forceinline__ float lse2(float a, float b) {
  float m = fmaxf(a, b);
  if (m <= 0.5f * NEG) return NEG;
  return m + logf(expf(a - m) + expf(b - m));
}
__device__ __forceinline__ float lse3(float a, float b, float c) {
  float m = fmaxf(a, fmaxf(b, c));
  if (m <= 0.5f * NEG) return NEG;
  return m + logf(expf(a - m) + expf(b - m) + expf(c - m));
}

// workspace layout (floats): lse[T*B] | lp[T*B*S] | alpha[T*B*S] | beta[T*B*S],  S = 2*Lmax+1
__global__ __launch_bounds__(256) void ctc_lse_gather(const float* __restrict__ logits, int T, int B, int V,
                                                      const int* __restrict__ hlens, const int* __restrict__ labels,
                                                      const int* __restrict__ loff, const int* __restrict__ llen, int S,
                                                      float* __restrict__ lse, float* __restrict__ lp) {
  int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  int lane = threadIdx.x & 63;
  if (row >= T * B) return;
  int t = row / B, b = row % B;
  if (t >= hlens[b]) return;
  const float* x = logits + (long)row * V;
  float m = -3.0e38f;
  for (int v = lane; v < V; v += 64) m = fmaxf(m, x[v]);
  m = wave_max(m);
  float s = 0.f;
  for (int v = lane; v < V; v += 64) s += expf(x[v] - m);
  s = wave_sum(s);
  float l = m + logf(s);
  if (lane == 0) lse[row] = l;
  int Sb = 2 * llen[b] + 1;
  const int* lab = labels + loff[b];
  for (int sidx = lane; sidx < Sb; sidx += 64) {
    int v = (sidx & 1) ? lab[sidx >> 1] : 0;
    lp[(long)row * S + sidx] = x[v] - l;
  }
}

// The same with the row held in registers (NV values per lane, V <= 64 NV): the logits are read from memory ONCE, every load of a
// row is in flight before the first is used (the two-pass kernel above re-reads the 17 KB row from L2 and keeps one load in flight
// per lane: 2.0 TB/s on the 108 MB of config 4).
template <int NV>
__global__ __launch_bounds__(256) void ctc_lse_gather_reg(const float* __restrict__ logits, int T, int B, int V,
                                                          const int* __restrict__ hlens, const int* __restrict__ labels,
                                                          const int* __restrict__ loff, const int* __restrict__ llen, int S,
                                                          float* __restrict__ lse, float* __restrict__ lp) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= T * B) return;
  const int t = row / B, b = row % B;
  if (t >= hlens[b]) return;
  const float* x = logits + (long)row * V;
  float v[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) { const int c = lane + 64 * i; v[i] = c < V ? x[c] : -3.0e38f; }
  float m = -3.0e38f;
#pragma unroll
  for (int i = 0; i < NV; ++i) m = fmaxf(m, v[i]);
  m = wave_max(m);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) s += (lane + 64 * i < V) ? expf(v[i] - m) : 0.f;
  s = wave_sum(s);
  const float l = m + logf(s);
  if (lane == 0) lse[row] = l;
  const int Sb = 2 * llen[b] + 1;
  const int* lab = labels + loff[b];
  for (int sidx = lane; sidx < Sb; sidx += 64) {
    const int c = (sidx & 1) ? lab[sidx >> 1] : 0;
    lp[(long)row * S + sidx] = x[c] - l;
  }
}

__global__ void ctc_alpha_beta(int T, int B, const int* __restrict__ hlens, const int* __restrict__ labels,
                               const int* __restrict__ loff, const int* __restrict__ llen, int S,
                               const float* __restrict__ lp, float* __restrict__ alpha, float* __restrict__ beta,
                               float* __restrict__ nll) {
  extern __shared__ float sm[];      // prev[2][S+2]
  int b = blockIdx.x;
  int s = threadIdx.x;
  int L = llen[b], Sb = 2 * L + 1, Tb = hlens[b];
  const int* lab = labels + loff[b];
  float* p0 = sm;
  float* p1 = sm + (S + 2);
  // skip transition allowed into state s from s-2 ?
  bool skip_in = false, skip_out = false;
  if (s < Sb && (s & 1)) {
    if (s >= 3) skip_in = lab[s >> 1] != lab[(s >> 1) - 1];
    if (s + 2 < Sb) skip_out = lab[s >> 1] != lab[(s >> 1) + 1];
  }
  // ---- alpha ----
  float a = NEG;
  if (s < Sb && Tb > 0) {
    if (s == 0) a = lp[((long)0 * B + b) * S + 0];
    else if (s == 1 && Sb > 1) a = lp[((long)0 * B + b) * S + 1];
    alpha[((long)0 * B + b) * S + s] = a;
  }
  for (int t = 1; t < Tb; ++t) {
    float* cur = (t & 1) ? p0 : p1;
    if (s < Sb) cur[s + 2] = a;
    if (s < 2) cur[s] = NEG;
    __syncthreads();
    if (s < Sb) {
      float v = skip_in ? lse3(cur[s + 2], cur[s + 1], cur[s]) : lse2(cur[s + 2], cur[s + 1]);
      a = v + lp[((long)t * B + b) * S + s];
      if (v <= 0.5f * NEG) a = NEG;
      alpha[((long)t * B + b) * S + s] = a;
    }
  }
  __syncthreads();
  if (Tb > 0) {
    float* cur = p0;
    if (s < Sb) cur[s] = a;
    __syncthreads();
    if (s == 0) nll[b] = -(Sb > 1 ? lse2(cur[Sb - 1], cur[Sb - 2]) : cur[Sb - 1]);
  } else if (s == 0) nll[b] = 0.f;
  __syncthreads();
  // ---- beta (includes lp at its own frame) ----
  float be = NEG;
  if (s < Sb && Tb > 0) {
    if (s == Sb - 1 || s == Sb - 2) be = lp[((long)(Tb - 1) * B + b) * S + s];
    beta[((long)(Tb - 1) * B + b) * S + s] = be;
  }
  for (int t = Tb - 2; t >= 0; --t) {
    float* cur = (t & 1) ? p0 : p1;
    if (s < Sb) cur[s] = be;
    if (s < 2) cur[Sb + s] = NEG;
    __syncthreads();
    if (s < Sb) {
      float v = skip_out ? lse3(cur[s], cur[s + 1], cur[s + 2]) : lse2(cur[s], cur[s + 1]);
      be = v + lp[((long)t * B + b) * S + s];
      if (v <= 0.5f * NEG) be = NEG;
      beta[((long)t * B + b) * S + s] = be;
    }
  }
}

__global__ void ctc_loss_final(const float* __restrict__ nll, int B, float* loss) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += nll[b];
    loss[0] = s / (float)B;
  }
}

__global__ __launch_bounds__(256) void ctc_grad(const float* __restrict__ logits, int T, int B, int V,
                                                const int* __restrict__ hlens, const int* __restrict__ labels,
                                                const int* __restrict__ loff, const int* __restrict__ llen, int S,
                                                const float* __restrict__ lse, const float* __restrict__ lp,
                                                const float* __restrict__ alpha, const float* __restrict__ beta,
                                                const float* __restrict__ nll, const float* __restrict__ gscale,
                                                float* __restrict__ dlogits, int ldd) {
  extern __shared__ float occ[];     // [4][S]
  int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  int row = blockIdx.x * 4 + wv;
  bool live = row < T * B;
  int t = live ? row / B : 0, b = live ? row % B : 0;
  bool active = live && t < hlens[b];
  float scale = (gscale ? gscale[0] : 1.f) / (float)B;
  float* o = occ + wv * S;
  int Sb = active ? 2 * llen[b] + 1 : 0;
  const int* lab = labels + (live ? loff[b] : 0);
  float blank = 0.f;
  if (active) {
    float nl = nll[b];
    for (int s = lane; s < Sb; s += 64) {
      long i = (long)row * S + s;
      float a = alpha[i], be = beta[i];
      float v = (a <= 0.5f * NEG || be <= 0.5f * NEG) ? 0.f : expf(a + be - lp[i] + nl);
      o[s] = v;
      if (!(s & 1)) blank += v;
    }
    blank = wave_sum(blank);
  }
  float* dst = dlogits + (long)row * ldd;              // rows of ldd >= V floats; the padding columns are written as zeros
  if (live) {
    for (int v = V + lane; v < ldd; v += 64) dst[v] = 0.f;
    if (!active) {
      for (int v = lane; v < V; v += 64) dst[v] = 0.f;
    } else {
      const float* x = logits + (long)row * V;
      float l = lse[row];
      for (int v = lane; v < V; v += 64) {
        float y = expf(x[v] - l);
        if (v == 0) y -= blank;
        dst[v] = scale * y;
      }
    }
  }
  __syncthreads();     // block-wide visibility of the row just written (and of occ[])
  if (active) {
    for (int s = 1 + 2 * lane; s < Sb; s += 128) {     // odd states: one label each
      int l = lab[s >> 1];
      bool first = true;
      for (int s2 = 1; s2 < s; s2 += 2) if (lab[s2 >> 1] == l) { first = false; break; }
      if (first) {
        float tot = 0.f;
        for (int s2 = s; s2 < Sb; s2 += 2) if (lab[s2 >> 1] == l) tot += o[s2];
        dst[l] -= scale * tot;
      }
    }
  }
}
}  // namespace

extern "C" size_t re2e_ctc_workspace_bytes(int T, int B, int Lmax) {
  size_t S = 2 * (size_t)Lmax + 1;
  return ((size_t)T * B * (1 + 3 * S)) * sizeof(float);
}

extern "C" int re2e_ctc_fwd(const float* logits, int T, int B, int V, const int* hlens, const int* labels, const int* loff,
                            const int* llen, int Lmax, float* loss_out, float* nll_per_utt, void* workspace,
                            size_t workspace_bytes, hipStream_t stream) {
  RE2E_CHECK_ARG(logits && hlens && labels && loff && llen && loss_out && nll_per_utt && workspace, "null arg");
  RE2E_CHECK_ARG(T > 0 && B > 0 && V > 1 && Lmax >= 0, "bad shape");
  int S = 2 * Lmax + 1;
  RE2E_CHECK_ARG(S <= 1024, "label sequence too long (2L+1 must be <= 1024)");
  RE2E_CHECK_ARG(workspace_bytes >= re2e_ctc_workspace_bytes(T, B, Lmax), "workspace too small");
  float* lse = (float*)workspace;
  float* lp = lse + (size_t)T * B;
  float* alpha = lp + (size_t)T * B * S;
  float* beta = alpha + (size_t)T * B * S;
  const dim3 lg(cdiv((long)T * B, 4)), lb(256);
  const int nv = cdiv(V, 64);
  if (nv <= 8) hipLaunchKernelGGL(ctc_lse_gather_reg<8>, lg, lb, 0, stream, logits, T, B, V, hlens, labels, loff, llen, S, lse, lp);
  else if (nv <= 24) hipLaunchKernelGGL(ctc_lse_gather_reg<24>, lg, lb, 0, stream, logits, T, B, V, hlens, labels, loff, llen, S, lse, lp);
  else if (nv <= 72) hipLaunchKernelGGL(ctc_lse_gather_reg<72>, lg, lb, 0, stream, logits, T, B, V, hlens, labels, loff, llen, S, lse, lp);
  else hipLaunchKernelGGL(ctc_lse_gather, lg, lb, 0, stream, logits, T, B, V, hlens, labels, loff, llen, S, lse, lp);
  int threads = ((S + 63) / 64) * 64;
  hipLaunchKernelGGL(ctc_alpha_beta, dim3(B), dim3(threads), (size_t)2 * (S + 2) * sizeof(float), stream, T, B, hlens, labels, loff,
                     llen, S, (const float*)lp, alpha, beta, nll_per_utt);
  hipLaunchKernelGGL(ctc_loss_final, dim3(1), dim3(64), 0, stream, (const float*)nll_per_utt, B, loss_out);
  RE2E_LAUNCH_CHECK();
  return RE2E_OK;
}

extern "C" int re2e_ctc_bwd(const float* logits, int T, int B, int V, const int* hlens, const int* labels, const int* loff,
                            const int* llen, int Lmax, const float* nll_per_utt, const float* gscale, float* dlogits, int ldd,
                            const void* workspace, hipStream_t stream) {
  RE2E_CHECK_ARG(logits && hlens && labels && loff && llen && nll_per_utt && dlogits && workspace, "null arg");
  RE2E_CHECK_ARG(T > 0 && B > 0 && V > 1 && Lmax >= 0 && ldd >= V, "bad shape");
  int S = 2 * Lmax + 1;
  const float* lse = (const float*)workspace;
  const float* lp = lse + (size_t)T * B;
  const float* alpha = lp + (size_t)T * B * S;
  const float* beta = alpha + (size_t)T * B * S;
  hipLaunchKernelGGL(ctc_grad, dim3(cdiv((long)T * B, 4)), dim3(256), (size_t)4 * S * sizeof(float), stream, logits, T, B, V, hlens,
                     labels, loff, llen, S, lse, lp, alpha, beta, nll_per_utt, gscale, dlogits, ldd);
  RE2E_LAUNCH_CHECK();
  return RE2E_OK;
}


// ============================================================================================
// N3: CTC prefix scores for joint CTC / attention beam search (CTCPrefixScore, model/e2e_ctc.py:78-155, called per
// hypothesis at model/e2e_decoder.py:231-263) -- one launch per output position for ALL live hypotheses.
// One workgroup per hypothesis:
//   1. the top `ctc_beam` labels of its attention log-probabilities (torch.topk order: descending, ties -> lower label),
//      by `ctc_beam` rounds of a workgroup arg-max over the row held in LDS;
//   2. one thread per candidate label runs Algorithm 2's recursion over the T frames: the hypothesis' previous forward
//      variables r_prev (T,2) are staged in LDS (shared by all its candidates), the candidate's own r_t^n, r_t^b live in two
//      registers and are streamed out as the candidate's new state;
//   3. local score att_weight * att + ctc_weight * (ctc - ctc_prev) per candidate (e2e_decoder.py:246-248).
// Only cand / local / ctc_score (ctc_beam values each per hypothesis) leave the GPU; the states stay in `r_new` and the
// survivors are gathered by index for the next position.
// ============================================================================================
namespace {
constexpr float CTC_LOGZERO = -10000000000.0f;
__device__ __forceinline__ float logaddexp_(float x, float y) {      // numpy's float32 logaddexp
  if (x == y) return x + 0.69314718055994530942f;
  const float d = x - y;
  return d > 0.f ? x + log1pf(expf(-d)) : y + log1pf(expf(d));
}

// CTCPrefixScore.__call__ for ONE candidate label c of one hypothesis (e2e_ctc.py:113-155): new forward variables -> rn (T, 2), prefix score.
__device__ __forceinline__ void prefix_recursion(const float* __restrict__ lpz, int T, int V, const float* rp, const float* rsum, int c, float c_att,
                                                 int n, int last, float prev, float att_weight, float ctc_weight, int blank, int eos,
                                                 float* __restrict__ rn, int* cand_o, float* local_o, float* ctc_o) {
  // a hypothesis longer than the T frames (recog maxlenratio > 1): upstream's r[start - 1] raises IndexError there and the host
  // caller raises it too (beam_search.py); here the state writes are clamped to the candidate's own (T, 2) block
  const int start = min(n > 1 ? n : 1, T);
  float rn_n, rn_b;                                   // r[t-1][0], r[t-1][1]
  for (int t = 0; t < start - 1; ++t) { rn[2 * t] = CTC_LOGZERO; rn[2 * t + 1] = CTC_LOGZERO; }     // never read (numpy leaves them unset)
  if (n == 0) { rn_n = lpz[c]; rn_b = CTC_LOGZERO; } else { rn_n = CTC_LOGZERO; rn_b = CTC_LOGZERO; }
  rn[2 * (start - 1)] = rn_n; rn[2 * (start - 1) + 1] = rn_b;
  const bool rep = n > 0 && c == last;                // a repeated label needs a blank in between
  float log_psi = rn_n;
  for (int t = start; t < T; ++t) {
    const float phi = rep ? rp[2 * (t - 1) + 1] : rsum[t - 1];
    const float xs = lpz[(long)t * V + c], xb = lpz[(long)t * V + blank];
    const float nn = logaddexp_(rn_n, phi) + xs;
    const float nb = logaddexp_(rn_n, rn_b) + xb;
    log_psi = logaddexp_(log_psi, phi + xs);
    rn_n = nn; rn_b = nb;
    rn[2 * t] = nn; rn[2 * t + 1] = nb;
  }
  if (c == eos) log_psi = rsum[T - 1];
  *cand_o = c;
  *ctc_o = log_psi;
  *local_o = att_weight * c_att + ctc_weight * (log_psi - prev);
}

__global__ __launch_bounds__(256) void ctc_prefix_kernel(const float* __restrict__ lpz, int T, int V, const float* __restrict__ att, const float* __restrict__ r_prev,
                                                         const int* __restrict__ last_label, const int* __restrict__ out_len,
                                                         const float* __restrict__ prev_score, int ctc_beam, float att_weight, float ctc_weight, int blank, int eos,
                                                         int* __restrict__ cand_out, float* __restrict__ local_out, float* __restrict__ ctc_out,
                                                         float* __restrict__ r_new) {
  extern __shared__ float sm[];
  float* row = sm;                       // [V] attention log-probabilities, selected entries overwritten with -inf
  float* rp = sm + V;                    // [T][2] previous forward variables
  float* rsum = rp + 2 * T;              // [T]
  __shared__ float red_v[4];
  __shared__ int red_i[4];
  __shared__ int cand[64];
  __shared__ float cand_att[64];
  const int h = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  // NaN scores are ranked as -inf (torch.topk would rank them first; a NaN row means the decoder has already diverged) and an
  // entry that has been selected is marked with NaN and skipped afterwards, so the ctc_beam candidates are always ctc_beam
  // DISTINCT, valid labels -- also for rows with fewer than ctc_beam entries above -inf (ties -> lower label, as torch.topk)
  for (int i = tid; i < V; i += 256) { const float v = att[(long)h * V + i]; row[i] = v == v ? v : -INFINITY; }
  for (int i = tid; i < 2 * T; i += 256) rp[i] = r_prev[(long)h * 2 * T + i];
  __syncthreads();
  for (int t = tid; t < T; t += 256) rsum[t] = logaddexp_(rp[2 * t], rp[2 * t + 1]);
  constexpr int NONE = 0x7fffffff;
  for (int k = 0; k < ctc_beam; ++k) {
    float bv = -INFINITY; int bi = NONE;
    for (int i = tid; i < V; i += 256) { const float v = row[i]; if (v == v && (bi == NONE || v > bv)) { bv = v; bi = i; } }   // ascending i: first maximum wins
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(bv, o, 64); const int oi = __shfl_xor(bi, o, 64);
      if (oi != NONE && (bi == NONE || ov > bv || (ov == bv && oi < bi))) { bv = ov; bi = oi; }
    }
    if (lane == 0) { red_v[wid] = bv; red_i[wid] = bi; }
    __syncthreads();
    if (tid == 0) {
      for (int w = 1; w < 4; ++w) if (red_i[w] != NONE && (bi == NONE || red_v[w] > bv || (red_v[w] == bv && red_i[w] < bi))) { bv = red_v[w]; bi = red_i[w]; }
      cand[k] = bi; cand_att[k] = bv; row[bi] = __uint_as_float(0x7fc00000u);     // ctc_beam <= V: an unselected entry always exists
    }
    __syncthreads();
  }
  if (tid < ctc_beam)
    prefix_recursion(lpz, T, V, rp, rsum, cand[tid], cand_att[tid], out_len[h], last_label[h], prev_score[h], att_weight, ctc_weight, blank, eos,
                     r_new + ((long)h * ctc_beam + tid) * 2 * T, cand_out + (long)h * ctc_beam + tid, local_out + (long)h * ctc_beam + tid,
                     ctc_out + (long)h * ctc_beam + tid);
}

// The same scores for a GIVEN candidate list (ctc_weight == 1.0 scores all V labels, in the order of their attention scores: the list
// is a device-side stable sort of the attention row): grid (ceil(ncand / 256), hypotheses), one thread per candidate.
__global__ __launch_bounds__(256) void ctc_prefix_cands_kernel(const float* __restrict__ lpz, int T, int V, const float* __restrict__ att,
                                                               const float* __restrict__ r_prev, const int* __restrict__ last_label,
                                                               const int* __restrict__ out_len, const float* __restrict__ prev_score,
                                                               const int* __restrict__ cand_in, int ncand, float att_weight, float ctc_weight,
                                                               int blank, int eos, float* __restrict__ local_out, float* __restrict__ ctc_out,
                                                               float* __restrict__ r_new) {
  extern __shared__ float sm[];
  float* rp = sm;                        // [T][2]
  float* rsum = rp + 2 * T;              // [T]
  const int h = blockIdx.y, tid = threadIdx.x, k = blockIdx.x * 256 + tid;
  for (int i = tid; i < 2 * T; i += 256) rp[i] = r_prev[(long)h * 2 * T + i];
  __syncthreads();
  for (int t = tid; t < T; t += 256) rsum[t] = logaddexp_(rp[2 * t], rp[2 * t + 1]);
  __syncthreads();
  if (k >= ncand) return;
  const int c = cand_in[(long)h * ncand + k];
  int dummy;
  prefix_recursion(lpz, T, V, rp, rsum, c, att[(long)h * V + c], out_len[h], last_label[h], prev_score[h], att_weight, ctc_weight, blank, eos,
                   r_new + ((long)h * ncand + k) * 2 * T, &dummy, local_out + (long)h * ncand + k, ctc_out + (long)h * ncand + k);
}
}  // namespace

extern "C" int re2e_ctc_prefix_score(const float* lpz, int T, int V, const float* att_lsm, int nh, const float* r_prev, const int* last_label_dev,
                                     const int* out_len_dev, const float* prev_score_dev, int ctc_beam, float att_weight, float ctc_weight, int blank,
                                     int eos, int* cand_out, float* local_out, float* ctc_score_out, float* r_new, hipStream_t stream) {
  RE2E_CHECK_ARG(lpz && att_lsm && r_prev && last_label_dev && out_len_dev && prev_score_dev && cand_out && local_out && ctc_score_out && r_new,
                 "null operand");
  RE2E_CHECK_ARG(T > 0 && V > 0 && nh > 0 && blank >= 0 && blank < V && eos >= 0 && eos < V, "bad geometry");
  if (ctc_beam < 1 || ctc_beam > 64 || ctc_beam > V) { re2e_set_error("re2e_ctc_prefix_score: ctc_beam must be in [1, min(64, V)]"); return RE2E_EUNSUPPORTED; }
  const size_t lds = ((size_t)V + 3 * (size_t)T) * sizeof(float);
  if (lds > 150 * 1024) { re2e_set_error("re2e_ctc_prefix_score: V + 3T floats exceed the LDS"); return RE2E_EUNSUPPORTED; }
  static LdsLimit lim;
  lim.ensure(reinterpret_cast<const void*>(&ctc_prefix_kernel), lds);
  hipLaunchKernelGGL(ctc_prefix_kernel, dim3(nh), dim3(256), lds, stream, lpz, T, V, att_lsm, r_prev, last_label_dev, out_len_dev, prev_score_dev,
                     ctc_beam, att_weight, ctc_weight, blank, eos, cand_out, local_out, ctc_score_out, r_new);
  RE2E_LAUNCH_CHECK();
  return RE2E_OK;
}

extern "C" int re2e_ctc_prefix_score_cands(const float* lpz, int T, int V, const float* att_lsm, int nh, const float* r_prev, const int* last_label_dev,
                                           const int* out_len_dev, const float* prev_score_dev, const int* cand_dev, int ncand, float att_weight,
                                           float ctc_weight, int blank, int eos, float* local_out, float* ctc_score_out, float* r_new, hipStream_t stream) {
  RE2E_CHECK_ARG(lpz && att_lsm && r_prev && last_label_dev && out_len_dev && prev_score_dev && cand_dev && local_out && ctc_score_out && r_new, "null operand");
  RE2E_CHECK_ARG(T > 0 && V > 0 && nh > 0 && ncand > 0 && ncand <= V && blank >= 0 && blank < V && eos >= 0 && eos < V, "bad geometry");
  const size_t lds = 3 * (size_t)T * sizeof(float);
  if (lds > 150 * 1024) { re2e_set_error("re2e_ctc_prefix_score_cands: 3T floats exceed the LDS"); return RE2E_EUNSUPPORTED; }
  static LdsLimit lim;
  lim.ensure(reinterpret_cast<const void*>(&ctc_prefix_cands_kernel), lds);
  hipLaunchKernelGGL(ctc_prefix_cands_kernel, dim3(cdiv(ncand, 256), nh), dim3(256), lds, stream, lpz, T, V, att_lsm, r_prev, last_label_dev, out_len_dev,
                     prev_score_dev, cand_dev, ncand, att_weight, ctc_weight, blank, eos, local_out, ctc_score_out, r_new);
  RE2E_LAUNCH_CHECK();
  return RE2E_OK;
}

// K8 (tail of the decoder): embedding gather / deterministic embedding gradient, and the
// cross-entropy-with-ignore + argmax accuracy over the (B*(L+1), V) decoder logits
// (model/e2e_decoder.py:117,155-161; th_accuracy model/e2e_common.py:198-205).
// One wavefront per logits row: max / log-sum-exp / argmax with wavefront shuffles, logits are
// read once in the forward and once in the backward (HBM-bound).
#include "common.h"

namespace {
__global__ void embedding_fwd_kernel(const float* __restrict__ table, const int* __restrict__ ids, int n, int D,
                                     float* __restrict__ out, long ldo) {
  long tot = (long)n * D;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += (long)gridDim.x * blockDim.x) {
    int d = (int)(i % D); int r = (int)(i / D);
    out[(long)r * ldo + d] = table[(long)ids[r] * D + d];
  }
}
// one block per vocabulary row.  Wavefront 0 compacts the positions whose id is this row into an LDS list IN ORDER (ballot +
// prefix count), then every thread sums its columns over that short list => bitwise reproducible, and the id list is
// scanned once per row instead of once per column group.
__global__ __launch_bounds__(128) void embedding_bwd_kernel(const float* __restrict__ dout, long ldo, const int* __restrict__ ids, int n, int D,
                                                            float* __restrict__ dtable, float beta) {
  extern __shared__ int hits[];        // [n] worst case
  __shared__ int nhit;
  const int v = blockIdx.x, lane = threadIdx.x & 63;
  if (threadIdx.x < 64) {
    int cnt = 0;
    for (int base = 0; base < n; base += 64) {
      const int i = base + lane;
      const bool m = i < n && ids[i] == v;
      const unsigned long long mask = __ballot(m);
      if (m) hits[cnt + __popcll(mask & ((1ull << lane) - 1ull))] = i;
      cnt += __popcll(mask);
    }
    if (lane == 0) nhit = cnt;
  }
  __syncthreads();
  const int nh = nhit;
  for (int d = threadIdx.x; d < D; d += blockDim.x) {
    float acc = 0.f;
    for (int k = 0; k < nh; ++k) acc += dout[(long)hits[k] * ldo + d];
    float* q = dtable + (long)v * D + d;
    *q = (beta != 0.f ? beta * (*q) : 0.f) + acc;
  }
}

// workspace: rowloss[R] | rowvalid[R] | rowcorrect[R]
__global__ __launch_bounds__(256) void ce_rows_kernel(const float* __restrict__ logits, const int* __restrict__ targets, int R,
                                                      int V, float* __restrict__ lse, float* __restrict__ ws) {
  int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  int lane = threadIdx.x & 63;
  if (row >= R) return;
  const float* x = logits + (long)row * V;
  float m = -3.0e38f; int am = 0x7fffffff;
  for (int v = lane; v < V; v += 64) { float xv = x[v]; if (xv > m) { m = xv; am = v; } }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    float m2 = __shfl_xor(m, o, 64); int a2 = __shfl_xor(am, o, 64);
    if (m2 > m || (m2 == m && a2 < am)) { m = m2; am = a2; }
  }
  float s = 0.f;
  for (int v = lane; v < V; v += 64) s += expf(x[v] - m);
  s = wave_sum(s);
  float l = m + logf(s);
  if (lane == 0) {
    int tg = targets[row];
    lse[row] = l;
    ws[row] = tg >= 0 ? l - x[tg] : 0.f;
    ws[R + row] = tg >= 0 ? 1.f : 0.f;
    ws[2 * R + row] = (tg >= 0 && am == tg) ? 1.f : 0.f;
  }
}
__global__ void ce_final_kernel(const float* __restrict__ ws, int R, float scale, float* out) {
  __shared__ float red[16];
  float a = 0.f, b = 0.f, c = 0.f;
  for (int i = threadIdx.x; i < R; i += blockDim.x) { a += ws[i]; b += ws[R + i]; c += ws[2 * R + i]; }
  a = block_sum(a, red); b = block_sum(b, red); c = block_sum(c, red);
  if (threadIdx.x == 0) { out[0] = scale * a / b; out[1] = c; out[2] = b; }
}
__global__ __launch_bounds__(256) void ce_bwd_kernel(const float* __restrict__ logits, const int* __restrict__ targets,
                                                     const float* __restrict__ lse, const float* __restrict__ fwd_out, int R, int V,
                                                     float scale, const float* __restrict__ gscale, float* __restrict__ dlogits) {
  int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  int lane = threadIdx.x & 63;
  if (row >= R) return;
  float* d = dlogits + (long)row * V;
  int tg = targets[row];
  if (tg < 0) { for (int v = lane; v < V; v += 64) d[v] = 0.f; return; }
  float g = (gscale ? gscale[0] : 1.f) * scale / fwd_out[2];
  const float* x = logits + (long)row * V;
  float l = lse[row];
  for (int v = lane; v < V; v += 64) {
    float y = expf(x[v] - l);
    if (v == tg) y -= 1.f;
    d[v] = g * y;
  }
}
// label-smoothing regulariser (e2e_decoder.py:162-166): reg = -(1/nutt) sum_r sum_v log_softmax(y)[r][v] * dist[v] over ALL rows
// (padded ones included, as upstream).  Row value: lse_r * S - sum_v y[r][v] dist[v],  S = sum_v dist[v].
__global__ __launch_bounds__(256) void lsm_rows_kernel(const float* __restrict__ logits, const float* __restrict__ dist, int R, int V,
                                                       float* __restrict__ rowval) {
  int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  int lane = threadIdx.x & 63;
  if (row >= R) return;
  const float* x = logits + (long)row * V;
  float m = -3.0e38f;
  for (int v = lane; v < V; v += 64) m = fmaxf(m, x[v]);
  m = wave_max(m);
  float s = 0.f, dot = 0.f, S = 0.f;
  for (int v = lane; v < V; v += 64) { float d = dist[v]; s += expf(x[v] - m); dot += x[v] * d; S += d; }
  s = wave_sum(s); dot = wave_sum(dot); S = wave_sum(S);
  if (lane == 0) rowval[row] = (m + logf(s)) * S - dot;
}
__global__ void lsm_final_kernel(const float* __restrict__ rowval, int R, float inv_nutt, float* out) {
  __shared__ float red[16];
  float a = 0.f;
  for (int i = threadIdx.x; i < R; i += blockDim.x) a += rowval[i];
  a = block_sum(a, red);
  if (threadIdx.x == 0) out[0] = a * inv_nutt;
}
// d reg / d y[r][v] = (softmax[r][v] * S - dist[v]) / nutt, times the incoming gradient
__global__ __launch_bounds__(256) void lsm_bwd_kernel(const float* __restrict__ logits, const float* __restrict__ dist, int R, int V, float inv_nutt,
                                                      const float* __restrict__ gscale, float* __restrict__ dlogits) {
  int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  int lane = threadIdx.x & 63;
  if (row >= R) return;
  const float* x = logits + (long)row * V;
  float m = -3.0e38f;
  for (int v = lane; v < V; v += 64) m = fmaxf(m, x[v]);
  m = wave_max(m);
  float s = 0.f, S = 0.f;
  for (int v = lane; v < V; v += 64) { s += expf(x[v] - m); S += dist[v]; }
  s = wave_sum(s); S = wave_sum(S);
  const float g = (gscale ? gscale[0] : 1.f) * inv_nutt, l = m + logf(s);
  float* d = dlogits + (long)row * V;
  for (int v = lane; v < V; v += 64) d[v] = g * (expf(x[v] - l) * S - dist[v]);
}
}  // namespace

extern "C" int re2e_lsm_fwd(const float* logits, const float* dist, int R, int V, int nutt, float* out, void* workspace, size_t workspace_bytes,
                            hipStream_t stream) {
  RE2E_CHECK_ARG(logits && dist && out && workspace && R > 0 && V > 0 && nutt > 0, "bad args");
  RE2E_CHECK_ARG(workspace_bytes >= (size_t)R * sizeof(float), "workspace too small (R floats)");
  hipLaunchKernelGGL(lsm_rows_kernel, dim3(cdiv(R, 4)), dim3(256), 0, stream, logits, dist, R, V, (float*)workspace);
  hipLaunchKernelGGL(lsm_final_kernel, dim3(1), dim3(256), 0, stream, (const float*)workspace, R, 1.0f / nutt, out);
  RE2E_LAUNCH_CHECK();
  return RE2E_OK;
}
extern "C" int re2e_lsm_bwd(const float* logits, const float* dist, int R, int V, int nutt, const float* gscale, float* dlogits,
                            hipStream_t stream) {
  RE2E_CHECK_ARG(logits && dist && dlogits && R > 0 && V > 0 && nutt > 0, "bad args");
  hipLaunchKernelGGL(lsm_bwd_kernel, dim3(cdiv(R, 4)), dim3(256), 0, stream, logits, dist, R, V, 1.0f / nutt, gscale, dlogits);
  RE2E_LAUNCH_CHECK();
  return RE2E_OK;
}

extern "C" int re2e_embedding_fwd(const float* table, const int* ids, int n, int D, float* out, long ldo, hipStream_t stream) {
  RE2E_CHECK_ARG(table && ids && out && n > 0 && D > 0 && ldo >= D, "bad args");
  long tot = (long)n * D;
  long g = (tot + 255) / 256; if (g > 4096) g = 4096;
  hipLaunchKernelGGL(embedding_fwd_kernel, dim3((int)g), dim3(256), 0, stream, table, ids, n, D, out, ldo);
  RE2E_LAUNCH_CHECK();
  return RE2E_OK;
}
extern "C" int re2e_embedding_bwd(const float* dout, long ldo, const int* ids, int n, int D, int V, float* dtable, float beta,
                                  hipStream_t stream) {
  RE2E_CHECK_ARG(dout && ids && dtable && n > 0 && D > 0 && V > 0, "bad args");
  RE2E_CHECK_ARG((size_t)n * sizeof(int) <= 60000, "too many tokens for the LDS id list");
  hipLaunchKernelGGL(embedding_bwd_kernel, dim3(V), dim3(128), (size_t)n * sizeof(int), stream, dout, ldo, ids, n, D, dtable, beta);
  RE2E_LAUNCH_CHECK();
  return RE2E_OK;
}
// arg-max of each row (lowest index on ties): the token fed back by scheduled sampling / greedy attention passes
__global__ __launch_bounds__(256) void argmax_rows_kernel(const float* __restrict__ x, int R, int V, long ldx, int* __restrict__ out) {
  int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  int lane = threadIdx.x & 63;
  if (row >= R) return;
  const float* xr = x + (long)row * ldx;
  float m = -3.0e38f; int am = 0x7fffffff;
  for (int v = lane; v < V; v += 64) { float xv = xr[v]; if (xv > m) { m = xv; am = v; } }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    float m2 = __shfl_xor(m, o, 64); int a2 = __shfl_xor(am, o, 64);
    if (m2 > m || (m2 == m && a2 < am)) { m = m2; am = a2; }
  }
  if (lane == 0) out[row] = am;
}
// log-softmax of each row (decoding: local attention scores and the CTC frame posteriors lpz)
__global__ __launch_bounds__(256) void log_softmax_rows_kernel(const float* __restrict__ x, int R, int V, long ldx, float* __restrict__ out) {
  int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  int lane = threadIdx.x & 63;
  if (row >= R) return;
  const float* xr = x + (long)row * ldx;
  float m = -3.0e38f;
  for (int v = lane; v < V; v += 64) m = fmaxf(m, xr[v]);
  m = wave_max(m);
  float s = 0.f;
  for (int v = lane; v < V; v += 64) s += expf(xr[v] - m);
  s = wave_sum(s);
  const float l = m + logf(s);
  float* o = out + (long)row * V;
  for (int v = lane; v < V; v += 64) o[v] = xr[v] - l;
}
extern "C" int re2e_log_softmax_rows(const float* x, int R, int V, long ldx, float* out, hipStream_t stream) {
  RE2E_CHECK_ARG(x && out && R > 0 && V > 0 && ldx >= V, "bad args");
  hipLaunchKernelGGL(log_softmax_rows_kernel, dim3(cdiv(R, 4)), dim3(256), 0, stream, x, R, V, ldx, out);
  RE2E_LAUNCH_CHECK();
  return RE2E_OK;
}
extern "C" int re2e_argmax_rows(const float* x, int R, int V, long ldx, int* out_ids, hipStream_t stream) {
  RE2E_CHECK_ARG(x && out_ids && R > 0 && V > 0 && ldx >= V, "bad args");
  hipLaunchKernelGGL(argmax_rows_kernel, dim3(cdiv(R, 4)), dim3(256), 0, stream, x, R, V, ldx, out_ids);
  RE2E_LAUNCH_CHECK();
  return RE2E_OK;
}
extern "C" int re2e_ce_fwd(const float* logits, const int* targets, int R, int V, float scale, float* out, float* lse, void* workspace,
                           size_t workspace_bytes, hipStream_t stream) {
  RE2E_CHECK_ARG(logits && targets && out && lse && workspace && R > 0 && V > 0, "bad args");
  RE2E_CHECK_ARG(workspace_bytes >= (size_t)3 * R * sizeof(float), "workspace too small (3*R floats)");
  hipLaunchKernelGGL(ce_rows_kernel, dim3(cdiv(R, 4)), dim3(256), 0, stream, logits, targets, R, V, lse, (float*)workspace);
  hipLaunchKernelGGL(ce_final_kernel, dim3(1), dim3(256), 0, stream, (const float*)workspace, R, scale, out);
  RE2E_LAUNCH_CHECK();
  return RE2E_OK;
}
extern "C" int re2e_ce_bwd(const float* logits, const int* targets, const float* lse, const float* fwd_out, int R, int V, float scale,
                           const float* gscale, float* dlogits, hipStream_t stream) {
  RE2E_CHECK_ARG(logits && targets && lse && fwd_out && dlogits && R > 0 && V > 0, "bad args");
  hipLaunchKernelGGL(ce_bwd_kernel, dim3(cdiv(R, 4)), dim3(256), 0, stream, logits, targets, lse, fwd_out, R, V, scale, gscale, dlogits);
  RE2E_LAUNCH_CHECK();
  return RE2E_OK;
}

// Weight gradient of a 3x3 / stride-1 / pad-1 convolution as Winograd F(2x2,3x3) (the VGG conv1_2 / conv2_1 / conv2_2 weight gradients:
// 755 GFLOP of direct products per config-4 step, 6.1 ms at 120-125 TFLOP/s on the weight-gradient stream, which is the stream that
// finishes last).  In the transformed domain the sum over pixels becomes a sum over 2x2 output tiles of 16 outer products:
//
//     dW[o][c] = G^T ( sum_tiles  (B^T d B)[c]  (x)  (A dY A^T)[o] ) G         -- 16 instead of 4 * 9 multiply-adds per tile, 2.25x fewer
//
// One kernel does transform + product, two small ones finish (sum of the split-K slabs; G^T . G):
//   * a workgroup owns a 64-channel x 32-output-channel block of all 16 positions and a range of 16x8-pixel patches (32 tiles each);
//     per patch the raw input patch (+ 1-pixel halo, zero outside the image) and the dy patch are staged in LDS with 16-byte loads,
//     no prefetch: the second workgroup of the CU covers the load phase (prefetching patch k+1 into registers under the products of
//     patch k changed nothing: 1875 vs 1861 us at the conv1_2 shape);
//   * wavefront i owns row i of the transformed tile.  A k-step of v_mfma_f32_32x32x2_f32 contracts TWO tiles: lane (r, h) builds, for
//     tile 2s + h, row i of B^T d B for channels 2r and 2r + 1 (8 ds_read_b64 + 16 vector ops) and row i of A dY A^T for output channel r
//     (4 ds_read_b32 + 7 vector ops); 8 MFMAs per k-step, accumulators [4 positions][2 channel tiles] x 16 registers;
//   * LDS addresses are a lane constant + compile-time offsets (the k-step loop is unrolled; the two tiles of a k-step lie in the two
//     halves of the patch and walk along their tile rows), no bounds logic in the loop.
#include <hip/hip_runtime.h>
#include "../../include/re2e.h"
#include "common.h"

namespace {
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr unsigned WW_OOB = 0x80000000u;
constexpr int WW_CB = 64, WW_OB = 32;      // channel / output-channel block of a workgroup

struct WwArgs {
  const float* x; const float* dy; float* slabs;
  int NI, H, W, C, Cout;
  int px, py, npatch, nsplit, ncb, nblk;   // patches per image row / column, total, patch ranges, channel blocks, (c,o) blocks
  unsigned x_bytes, dy_bytes;
  const int* row_lim;   // optional, per image: patches that start at a row >= row_lim[n] are skipped (dy is zero there: re2e_conv3x3_wino_wgrad_rows)
  int flat;     // RE2E_EXPERIMENTS builds (RE2E_WW_FLAT=1): workgroup order of rounds 3-5
  int dbg;      // RE2E_EXPERIMENTS builds (RE2E_WW_DBG): 1 = every patch load re-reads the range's first patch (cache hits), 2 = only the first patch is loaded
};

template <int TXW>      // tiles per patch row: 8 (16 x 8 pixel patch) or 4 (8 x 16)
__global__ __launch_bounds__(256, 2) void wino_wgrad_kernel(WwArgs p) {
  constexpr int TYH = 32 / TXW, PW = 2 * TXW, PH = 2 * TYH;
  constexpr int XW = PW + 2, XH = PH + 2, XPIX = XW * XH;
  constexpr int XIT = (XPIX * 16 + 255) / 256, DIT = PW * PH * 8 / 256;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* xs = smem;                    // [XH][XW][64]
  float* ds = smem + XPIX * WW_CB;     // [PH][PW][32]
  const int tid = threadIdx.x, lane = tid & 63, lr = lane & 31, lh = lane >> 5;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  // the (channel, output-channel) blocks of one patch range read the same patches: they sit on ONE XCD (workgroups are dealt round-robin over the
  // 8 XCDs), next to each other in its dispatch order, so that the range's pixels come out of that XCD's L2 for all but the first of them
  // (p.flat, experiments build: neighbours in launch order = one block per XCD, rounds 3-5)
  const int xk = blockIdx.x >> 3;
  const int blk = p.flat ? blockIdx.x % p.nblk : xk % p.nblk, split = p.flat ? blockIdx.x / p.nblk : (xk / p.nblk) * 8 + (blockIdx.x & 7);
  if (split >= p.nsplit) return;
  const int cb = blk % p.ncb, ob = blk / p.ncb;
  const int c0 = cb * WW_CB, o0 = ob * WW_OB;
  const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsD = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dy), 0, p.dy_bytes, 0x00020000);
  const int per = (p.npatch + p.nsplit - 1) / p.nsplit;
  const int pbeg = split * per, pend = min(p.npatch, pbeg + per);

  f32x4 xr[XIT], dr[DIT];
  auto load_patch = [&](int pi) {
    const int n = pi / (p.py * p.px), rem = pi - n * (p.py * p.px);
    const int pyi = rem / p.px, pxi = rem - pyi * p.px;
    const int y0 = pyi * PH, x0 = pxi * PW;
#pragma unroll
    for (int it = 0; it < XIT; ++it) {
      const int item = it * 256 + tid, pix = item >> 4, c4 = item & 15;
      const int yy = pix / XW, xx = pix - yy * XW, iy = y0 - 1 + yy, ix = x0 - 1 + xx;
      const bool ok = pix < XPIX && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
      const unsigned off = (unsigned)((((n * p.H + iy) * p.W + ix) * p.C + c0 + c4 * 4) * 4);
      xr[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsX, ok ? off : WW_OOB, 0, 0));
    }
#pragma unroll
    for (int it = 0; it < DIT; ++it) {
      const int item = it * 256 + tid, pix = item >> 3, o4 = item & 7;
      const int yy = pix / PW, xx = pix - yy * PW, oy = y0 + yy, ox = x0 + xx;
      const bool ok = oy < p.H && ox < p.W;
      const unsigned off = (unsigned)((((n * p.H + oy) * p.W + ox) * p.Cout + o0 + o4 * 4) * 4);
      dr[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsD, ok ? off : WW_OOB, 0, 0));
    }
  };
  auto store_patch = [&]() {
#pragma unroll
    for (int it = 0; it < XIT; ++it) {
      const int item = it * 256 + tid;
      if (item < XPIX * 16) *reinterpret_cast<f32x4*>(xs + item * 4) = xr[it];
    }
#pragma unroll
    for (int it = 0; it < DIT; ++it) *reinterpret_cast<f32x4*>(ds + (it * 256 + tid) * 4) = dr[it];
  };

  // row i of B^T d: input rows (rA, rB) with sign sg;  row i of A dY: dy rows with coefficients (a0, a1)
  const int rA = wid == 0 ? 0 : (wid == 2 ? 2 : 1), rB = wid == 3 ? 3 : (wid == 2 ? 1 : 2);
  const float sg = wid == 1 ? 1.f : -1.f;
  const float a0 = wid == 3 ? 0.f : 1.f, a1 = wid == 0 ? 0.f : (wid == 1 ? 1.f : -1.f);
  // the two tiles of a k-step: tile row (TYH / 2) * lh + s / TXW, tile column s % TXW -- the k-steps walk along a tile row in each half
  // of the patch, so two of the four input columns of a tile are the previous k-step's (4 instead of 8 LDS reads and row combinations)
  constexpr int HR = TYH / 2;
  const float* xA = xs + ((rA + 2 * HR * lh) * XW) * WW_CB + 2 * lr;
  const float* xB = xs + ((rB + 2 * HR * lh) * XW) * WW_CB + 2 * lr;
  const float* dB = ds + (2 * HR * lh * PW) * WW_OB + lr;

  f32x16 acc[4][2];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][ct][r] = 0.f;

  for (int pi = pbeg; pi < pend; ++pi) {
    if (p.row_lim) {                       // ragged batch: a patch in rows where dy is zero adds nothing (uniform over the workgroup)
      const int n = pi / (p.py * p.px), pyi = (pi - n * (p.py * p.px)) / p.px;
      if (pyi * PH >= p.row_lim[n]) continue;
    }
    load_patch(pi);
    store_patch();
    __syncthreads();
    // The operands of k-step s + 1 (LDS reads + transforms) are independent of the MFMAs of k-step s: both are written into the loop
    // body and the scheduling groups below spread the former BETWEEN the latter -- a vector / LDS instruction outside a wave's MFMA
    // sequence is issued about once per MFMA of the SIMD's other wave (winograd.hip), inside it it is free.
    f32x2 t[4], v[2][4];
    float m[2][4];
    auto prep = [&](const int s, f32x2 (&vv)[4], float (&mm)[4]) {
      const int tyl = s / TXW, txl = s % TXW;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (txl != 0 && j < 2) { t[j] = t[j + 2]; continue; }
        const f32x2 da = *reinterpret_cast<const f32x2*>(xA + ((2 * tyl) * XW + 2 * txl + j) * WW_CB);
        const f32x2 db = *reinterpret_cast<const f32x2*>(xB + ((2 * tyl) * XW + 2 * txl + j) * WW_CB);
        t[j] = da + sg * db;
      }
      vv[0] = t[0] - t[2]; vv[1] = t[1] + t[2]; vv[2] = t[2] - t[1]; vv[3] = t[1] - t[3];
      const float e00 = dB[((2 * tyl) * PW + 2 * txl) * WW_OB], e01 = dB[((2 * tyl) * PW + 2 * txl + 1) * WW_OB];
      const float e10 = dB[((2 * tyl + 1) * PW + 2 * txl) * WW_OB], e11 = dB[((2 * tyl + 1) * PW + 2 * txl + 1) * WW_OB];
      const float s0 = a0 * e00 + a1 * e10, s1 = a0 * e01 + a1 * e11;
      mm[0] = s0; mm[1] = s0 + s1; mm[2] = s0 - s1; mm[3] = -s1;
    };
    prep(0, v[0], m[0]);
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      if (s + 1 < 16) prep(s + 1, v[(s + 1) & 1], m[(s + 1) & 1]);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[j][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[s & 1][j][0], m[s & 1][j], acc[j][0], 0, 0, 0);
        acc[j][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[s & 1][j][1], m[s & 1][j], acc[j][1], 0, 0, 0);
      }
      if (s + 1 < 16) {
#pragma unroll
        for (int g = 0; g < 8; ++g) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // one MFMA
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // one LDS read
          __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);   // four vector instructions
        }
      }
    }
    __syncthreads();
  }
  // slabs[split][pos][C][Cout]: lane = output channel o0 + lr, rows (r&3) + 8(r>>2) + 4lh of channel tile ct = channels c0 + 2 row + ct
  float* sl = p.slabs + ((long)split * 16 + 4 * wid) * p.C * p.Cout;
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int c = c0 + 2 * ((r & 3) + 8 * (r >> 2) + 4 * lh) + ct;
        sl[((long)j * p.C + c) * p.Cout + o0 + lr] = acc[j][ct][r];
      }
}

#ifdef RE2E_EXPERIMENTS
#include "experiments/wino_wgrad_dma.hip"    // the rejected LDS-DMA / packed-transform form (RE2E_WW_DMA=1): experiments build only
#endif

// red[e] = sum over the patch ranges of slabs[split][e], e < 16 * C * Cout: 16 slab lanes per element, combined through LDS in a fixed
// order (deterministic).  A thread per (c, o) walking 768 slabs on its own made the finishing pass as long as the products for the
// 64-channel layers (4096 threads, one dependent round trip per slab).
__global__ __launch_bounds__(1024) void wino_wgrad_reduce_kernel(const float* __restrict__ slabs, int nsplit, long tot, float* __restrict__ red) {
  __shared__ float part[16][64];
  const int e = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const long i = (long)blockIdx.x * 64 + e;
  float s = 0.f;
  if (i < tot)
    for (int z = sl; z < nsplit; z += 16) s += slabs[(long)z * tot + i];
  part[sl][e] = s;
  __syncthreads();
  if (sl != 0 || i >= tot) return;
  s = 0.f;
#pragma unroll
  for (int q = 0; q < 16; ++q) s += part[q][e];
  red[i] = s;
}

// gw[o][c][a][b] (+)= sum_ij G[i][a] G[j][b] sum_splits slabs[split][4i+j][c][o];  G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]
__global__ __launch_bounds__(256) void wino_wgrad_final_kernel(const float* __restrict__ slabs, int nsplit, int C, int Cout, float* __restrict__ gw,
                                                               float beta) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const long CO = (long)C * Cout;
  if (i >= CO) return;
  const int o = (int)(i % Cout), c = (int)(i / Cout);
  float u[4][4];
#pragma unroll
  for (int pos = 0; pos < 16; ++pos) u[pos >> 2][pos & 3] = 0.f;
  for (int s = 0; s < nsplit; ++s) {
    const float* sp = slabs + (long)s * 16 * CO + i;
#pragma unroll
    for (int pos = 0; pos < 16; ++pos) u[pos >> 2][pos & 3] += sp[pos * CO];
  }
  float t[3][4];                                                       // G^T u
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    t[0][j] = u[0][j] + 0.5f * (u[1][j] + u[2][j]);
    t[1][j] = 0.5f * (u[1][j] - u[2][j]);
    t[2][j] = 0.5f * (u[1][j] + u[2][j]) + u[3][j];
  }
  float* wp = gw + ((long)o * C + c) * 9;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float r0 = t[a][0] + 0.5f * (t[a][1] + t[a][2]), r1 = 0.5f * (t[a][1] - t[a][2]), r2 = 0.5f * (t[a][1] + t[a][2]) + t[a][3];
    if (beta != 0.f) { wp[a * 3] += r0; wp[a * 3 + 1] += r1; wp[a * 3 + 2] += r2; }
    else { wp[a * 3] = r0; wp[a * 3 + 1] = r1; wp[a * 3 + 2] = r2; }
  }
}

struct WwPlan { bool wide; int px, py, npatch, nsplit, ncb, nblk; size_t bytes; };
WwPlan ww_plan(int NI, int H, int W, int C, int Cout) {
  WwPlan q;
  const long pad_w = (long)cdiv(H, 8) * 8 * cdiv(W, 16) * 16, pad_n = (long)cdiv(H, 16) * 16 * cdiv(W, 8) * 8;
  q.wide = pad_w <= pad_n;
  q.px = cdiv(W, q.wide ? 16 : 8); q.py = cdiv(H, q.wide ? 8 : 16);
  q.npatch = NI * q.py * q.px;
  q.ncb = C / WW_CB; q.nblk = q.ncb * (Cout / WW_OB);
  // patch ranges: about three rounds of the 512 resident workgroups, at least 8 patches each
  int s = cdiv(1536, q.nblk);
  if (s > q.npatch / 8) s = q.npatch / 8;
  if (s < 1) s = 1;
  q.nsplit = s;
  q.bytes = (size_t)(s + 1) * 16 * C * Cout * sizeof(float);      // the slabs + their sum
  return q;
}

template <int TXW>
void launch_ww(const WwArgs& a, hipStream_t st) {
  constexpr int PW = 2 * TXW, PH = 64 / TXW;
#ifdef RE2E_EXPERIMENTS
  static const bool dma = exp_env("RE2E_WW_DMA") && atoi(exp_env("RE2E_WW_DMA")) == 1;
  if (dma) {
    constexpr size_t lds2 = (size_t)2 * (192 * WW_CB + PW * PH * WW_OB) * sizeof(float);
    static LdsLimit lim2;
    lim2.ensure(reinterpret_cast<const void*>(&wino_wgrad_dma_kernel<TXW>), lds2);
    hipLaunchKernelGGL((wino_wgrad_dma_kernel<TXW>), dim3((unsigned)(cdiv(a.nsplit, 8) * 8 * a.nblk)), dim3(256), lds2, st, a);
    return;
  }
#endif
  constexpr size_t lds = (size_t)((PW + 2) * (PH + 2) * WW_CB + PW * PH * WW_OB) * sizeof(float);
  static LdsLimit lim;
  lim.ensure(reinterpret_cast<const void*>(&wino_wgrad_kernel<TXW>), lds);
  hipLaunchKernelGGL((wino_wgrad_kernel<TXW>), dim3((unsigned)(cdiv(a.nsplit, 8) * 8 * a.nblk)), dim3(256), lds, st, a);
}
}  // namespace

extern "C" size_t re2e_conv3x3_wino_wgrad_workspace_bytes(int NI, int H, int W, int C, int Cout) {
  if (NI <= 0 || H <= 0 || W <= 0 || C <= 0 || Cout <= 0 || C % WW_CB || Cout % WW_OB) return 0;
  return ww_plan(NI, H, W, C, Cout).bytes;
}

static int ww_impl(const float* in, int NI, int H, int W, int C, const float* dout, int Cout, float* gw, float beta, const int* row_lim, void* workspace,
                   size_t workspace_bytes, hipStream_t stream) {
  RE2E_CHECK_ARG(in && dout && gw && workspace, "null operand");
  RE2E_CHECK_ARG(NI > 0 && H > 0 && W > 0 && C > 0 && Cout > 0, "bad geometry");
  RE2E_CHECK_ARG(beta == 0.f || beta == 1.f, "beta must be 0 or 1");
  if (C % WW_CB || Cout % WW_OB) { re2e_set_error("re2e_conv3x3_wino_wgrad: C must be a multiple of 64 and Cout of 32 (got %d, %d)", C, Cout); return RE2E_EUNSUPPORTED; }
  const long x_bytes = (long)NI * H * W * C * 4, dy_bytes = (long)NI * H * W * Cout * 4;
  if (x_bytes >= 0x7FFFFF00L || dy_bytes >= 0x7FFFFF00L) { re2e_set_error("re2e_conv3x3_wino_wgrad: tensors must be < 2 GiB"); return RE2E_EUNSUPPORTED; }
  RE2E_CHECK_ARG((reinterpret_cast<uintptr_t>(in) & 15) == 0 && (reinterpret_cast<uintptr_t>(dout) & 15) == 0, "in / dout must be 16-byte aligned");
  const WwPlan q = ww_plan(NI, H, W, C, Cout);
  RE2E_CHECK_ARG(workspace_bytes >= q.bytes, "workspace too small (re2e_conv3x3_wino_wgrad_workspace_bytes)");
  WwArgs a;
  a.x = in; a.dy = dout; a.slabs = static_cast<float*>(workspace);
  a.NI = NI; a.H = H; a.W = W; a.C = C; a.Cout = Cout;
  a.px = q.px; a.py = q.py; a.npatch = q.npatch; a.nsplit = q.nsplit; a.ncb = q.ncb; a.nblk = q.nblk;
  a.x_bytes = (unsigned)x_bytes; a.dy_bytes = (unsigned)dy_bytes;
  a.row_lim = row_lim;
  a.dbg = exp_env("RE2E_WW_DBG") ? atoi(exp_env("RE2E_WW_DBG")) : 0;
  a.flat = exp_env("RE2E_WW_FLAT") ? atoi(exp_env("RE2E_WW_FLAT")) : 0;
  static const bool log_calls = getenv("RE2E_IGEMM_LOG") != nullptr;   // tools/igemm_table.py joins this with a kernel trace
  if (log_calls)
    fprintf(stderr, "[igemm] A=WinoW B=DenseM tile=64x32x2 vec=1 M=%d N=%d K=%ld splits=%d\n", 9 * C, Cout, (long)NI * H * W, q.nsplit);
  if (q.wide) launch_ww<8>(a, stream); else launch_ww<4>(a, stream);
  const long tot = 16L * C * Cout;
  float* red = a.slabs + (long)q.nsplit * tot;
  hipLaunchKernelGGL(wino_wgrad_reduce_kernel, dim3((unsigned)cdiv(tot, 64)), dim3(1024), 0, stream, a.slabs, q.nsplit, tot, red);
  hipLaunchKernelGGL(wino_wgrad_final_kernel, dim3((unsigned)cdiv((long)C * Cout, 256)), dim3(256), 0, stream, red, 1, C, Cout, gw, beta);
  RE2E_LAUNCH_CHECK();
  return RE2E_OK;
}

extern "C" int re2e_conv3x3_wino_wgrad(const float* in, int NI, int H, int W, int C, const float* dout, int Cout, float* gw, float beta, void* workspace,
                                       size_t workspace_bytes, hipStream_t stream) {
  return ww_impl(in, NI, H, W, C, dout, Cout, gw, beta, nullptr, workspace, workspace_bytes, stream);
}

// The weight gradient of a RAGGED image batch: dout is zero in rows y >= row_lim[n] of image n (the caller's promise), patches that start there
// are skipped -- neither operand is read there.
extern "C" int re2e_conv3x3_wino_wgrad_rows(const float* in, int NI, int H, int W, int C, const float* dout, int Cout, float* gw, float beta,
                                            const int* row_lim, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  RE2E_CHECK_ARG(row_lim, "null row limits");
  return ww_impl(in, NI, H, W, C, dout, Cout, gw, beta, row_lim, workspace, workspace_bytes, stream);
}

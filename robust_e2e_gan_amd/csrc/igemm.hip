// K3 / K5 / K9: fp32 MFMA (v_mfma_f32_32x32x2_f32) tiled GEMM engine for gfx950.
//
// One mainloop serves the dense Linear GEMMs (NT / NN / TN), the implicit-GEMM convolution
// forward + data-gradient (im2col gather as the A operand, NHWC activations) and the
// convolution weight-gradient (im2col gather as the A operand over the pixel (=K) axis,
// split-K partial slabs + deterministic reduce).  Wave = 64 lanes computes TMxTN tiles of 32x32;
// the k index inside a group of 8 is assigned k = 8q + 4*(lane>>5) + j so that a lane fetches
// its 4 k-values for 4 consecutive MFMAs with ONE ds_read_b128 (LDS rows padded to BK+4 floats:
// conflict-free for the b128 lane groups).  Thin problems (Cin or Cout of 1) are
// routed to the direct kernels of thinconv.hip; the stride-2 data gradient runs its four output
// parity classes as blockIdx.z of one launch; workgroups are ordered XCD-aware (see the kernel).
//
// Replaces (reference, stock ATen ops): torch.nn.Linear / torch.mm call sites in
// model/e2e_encoder.py:145-147,173-174, model/e2e_ctc.py:51, model/e2e_attention.py:256,
// model/e2e_decoder.py:131,150, model/enhance_model.py:108-114 and nn.Conv2d in
// model/e2e_encoder.py:234-237 (VGG2L) and model/gan_model.py:63-90 (discriminator).
#include <stdlib.h>
#include <type_traits>

#include "common.h"

namespace {

template <int WM_, int WN_, int TM_, int TN_, int BK_>
struct Cfg {
  static constexpr int WM = WM_, WN = WN_, TM = TM_, TN = TN_, BK = BK_;
  static constexpr int BM = WM_ * TM_ * 32, BN = WN_ * TN_ * 32;
  static constexpr int THREADS = WM_ * WN_ * 64;
  static constexpr int LDK = BK_ + 4;   // padded k stride of a k-major LDS tile (floats): conflict-free b128 reads
};

// ------------------------------------------------------------------------------------------
// Operand loaders.  A loader describes a logical matrix X[row][k] (rows = M for the A operand,
// N for the B operand) and returns the BYTE OFFSET of an element group inside its tensor, or
// `nbytes` (= one past the end) when the group is out of range.  The kernel fetches through a raw
// buffer descriptor (buffer_load_dwordx4 ... offen): the hardware range check returns 0 for the
// out-of-range offset, so there is no branch and no select between a load and its LDS store --
// a tile's loads are issued back to back and stay in flight under the MFMA block.
// KMAJ loaders fetch along k (LDS tile [row][LDK]): per-item row state + ONE per-thread k state.
// MMAJ loaders fetch along rows (LDS tile [k][ROWS+4]): ONE per-thread row state + per-item k state.
// ------------------------------------------------------------------------------------------
struct DenseK {   // X[row*ld + k]
  static constexpr bool KMAJ = true;
  static constexpr const char* NAME = "DenseK";
  static constexpr bool IS_CONVK = false;
  const float* p; unsigned nbytes; long ld; int rows, K;
  struct Row { unsigned base; };     // row*ld*4, or nbytes when the row is out of range
  struct Kst { int k; };
  __device__ void init_row(Row& r, int row) const { r.base = row < rows ? (unsigned)((long)row * ld * 4) : nbytes; }
  __device__ void init_k(Kst& s, int k) const { s.k = k; }
  __device__ void advance(Kst& s, int bk) const { s.k += bk; }
  __device__ unsigned off(const Row& r, const Kst& s, int j) const {     // element k+j (j=0 for a float4 group)
    return (r.base != nbytes && s.k + j < K) ? r.base + (unsigned)(s.k + j) * 4u : nbytes;
  }
};

struct DenseM {   // X[k*ld + row]
  static constexpr bool KMAJ = false;
  static constexpr const char* NAME = "DenseM";
  static constexpr bool IS_CONVK = false;
  const float* p; unsigned nbytes; long ld; int rows, K;
  struct Row { int r0; };
  struct Kst { int k; };
  __device__ void init_row(Row& r, int row0) const { r.r0 = row0; }
  __device__ void init_k(Kst& s, int k) const { s.k = k; }
  __device__ void advance(Kst& s, int bk) const { s.k += bk; }
  __device__ unsigned off(const Row& r, const Kst& s, int j) const {
    return (s.k < K && r.r0 + j < rows) ? (unsigned)(((long)s.k * ld + r.r0 + j) * 4) : nbytes;
  }
};

// DenseM over a ROW MAP (re2e_gemm_tn_rows): logical k-row k is physical row map[k] of the tensor -- the contraction of a weight gradient
// runs over the valid (t, b) rows of a ragged time-major batch only.  The physical row of the NEXT k-tile is fetched one tile ahead (a load
// per thread and k-tile that has a whole MFMA block to land), so the operand loads never wait for the table.
struct DenseMG {   // X[map[k]*ld + row]
  static constexpr bool KMAJ = false;
  static constexpr const char* NAME = "DenseMG";
  static constexpr int LOOKAHEAD = 16;   // init_k fetches the physical row LOOKAHEAD k-rows ahead: the kernel's k-tile must be exactly this (static_assert there)
  static constexpr bool IS_CONVK = false;
  const float* p; unsigned nbytes; long ld; int rows, K; const int* map;
  int ident;      // map[k] == k for k < ident (every utterance is at least that long): no look-up there
  struct Row { int r0; };
  struct Kst { int k; unsigned koff; int nxt; };
  __device__ int phys(int k) const { return k < ident ? k : (k < K ? map[k] : 0); }
  __device__ void init_row(Row& r, int row0) const { r.r0 = row0; }
  __device__ void init_k(Kst& s, int k) const {
    s.k = k;
    s.koff = (unsigned)((long)phys(k) * ld * 4);
    s.nxt = phys(k + LOOKAHEAD);                        // = the k-tile advance() is called with
  }
  __device__ void advance(Kst& s, int bk) const {
    s.k += bk;
    s.koff = (unsigned)((long)s.nxt * ld * 4);
    s.nxt = phys(s.k + bk);
  }
  __device__ unsigned off(const Row& r, const Kst& s, int j) const {
    return (s.k < K && r.r0 + j < rows) ? s.koff + (unsigned)(r.r0 + j) * 4u : nbytes;
  }
};

// Conv loaders keep their addresses as 32-bit BYTE offsets split into a row part and a k part that are each
// updated incrementally, so the per-load work of the 16-byte path is one add, two bound checks and a select
// (tensors are < 4 GiB, checked by the callers; the split parts wrap mod 2^32 and their sum is exact whenever
// the coordinate is inside the image).
struct ConvK {    // rows = pixels, k = (kh, kw, ci) with ci fastest
  static constexpr bool KMAJ = true;
  static constexpr const char* NAME = "ConvK";
  static constexpr bool IS_CONVK = true;
  ConvGeom g; const float* p; unsigned nbytes; int rows, K;
  struct Row { int n, iy0, ix0; unsigned base; };        // base = ((n*H + iy0)*W + ix0)*C*4
  struct Kst { int k, ci, kh, kw, dy, dx; unsigned koff; };   // koff = ((dy*W + dx)*C + ci)*4
  __device__ void set_base(Row& r) const { r.base = (unsigned)((((long)r.n * g.H + r.iy0) * g.W + r.ix0) * g.C * 4); }
  __device__ void init_row(Row& r, int row) const {
    if (row >= rows) { r.n = -1; r.iy0 = -(1 << 28); r.ix0 = 0; r.base = 0; }   // iy never in range
    else {
      int px = row % g.PW; int t = row / g.PW; int py = t % g.PH; r.n = t / g.PH;
      r.iy0 = py * g.SY + g.OY0; r.ix0 = px * g.SX + g.OX0;
      set_base(r);
    }
  }
  __device__ void shift_row(Row& r, int dy, int dx) const { if (r.n >= 0) { r.iy0 += dy; r.ix0 += dx; set_base(r); } }
  __device__ void set_k(Kst& s) const {
    s.dy = s.kh * g.DY; s.dx = s.kw * g.DX;
    s.koff = (unsigned)(((s.dy * g.W + s.dx) * g.C + s.ci) * 4);
  }
  __device__ void init_k(Kst& s, int k) const {
    s.k = k; s.ci = k % g.C; int tap = k / g.C; s.kw = tap % g.KW; s.kh = tap / g.KW;
    set_k(s);
  }
  __device__ void advance(Kst& s, int bk) const {
    s.k += bk; s.ci += bk;
    while (s.ci >= g.C) { s.ci -= g.C; if (++s.kw == g.KW) { s.kw = 0; ++s.kh; } }
    set_k(s);
  }
  __device__ unsigned off(const Row& r, const Kst& s, int j) const {
    if (j == 0) {                                          // the 16-byte path only ever asks for j = 0
      const int iy = r.iy0 + s.dy, ix = r.ix0 + s.dx;
      const bool ok = (s.k < K) & ((unsigned)iy < (unsigned)g.H) & ((unsigned)ix < (unsigned)g.W);
      return ok ? r.base + s.koff : nbytes;
    }
    int ci = s.ci + j, kw = s.kw, kh = s.kh;               // scalar path: element k+j may sit in the next tap
    while (ci >= g.C) { ci -= g.C; if (++kw == g.KW) { kw = 0; ++kh; } }
    const int iy = r.iy0 + kh * g.DY, ix = r.ix0 + kw * g.DX;
    const bool ok = (r.n >= 0) & (s.k + j < K) & ((unsigned)iy < (unsigned)g.H) & ((unsigned)ix < (unsigned)g.W);
    return ok ? (unsigned)(((((long)r.n * g.H + iy) * g.W + ix) * g.C + ci) * 4) : nbytes;
  }
};

struct ConvM {    // rows = (kh, kw, ci) (ci fastest), k = pixel  (weight-gradient A operand)
  static constexpr bool KMAJ = false;
  static constexpr const char* NAME = "ConvM";
  static constexpr bool IS_CONVK = false;
  ConvGeom g; const float* p; unsigned nbytes; int rows, K;
  struct Row { int r0, ci, kh, kw, dy, dx; unsigned roff; };   // roff = ((dy*W + dx)*C + ci)*4
  struct Kst { int k, n, py, px, iy0, ix0; unsigned base; };    // base = ((n*H + iy0)*W + ix0)*C*4
  __device__ void init_row(Row& r, int row0) const {
    r.r0 = row0;
    int rr = row0 < rows ? row0 : 0;
    r.ci = rr % g.C; int tap = rr / g.C; r.kw = tap % g.KW; r.kh = tap / g.KW;
    r.dy = r.kh * g.DY; r.dx = r.kw * g.DX;
    r.roff = (unsigned)(((r.dy * g.W + r.dx) * g.C + r.ci) * 4);
  }
  __device__ void set_k(Kst& s) const {
    s.iy0 = s.py * g.SY + g.OY0; s.ix0 = s.px * g.SX + g.OX0;
    s.base = (unsigned)((((long)s.n * g.H + s.iy0) * g.W + s.ix0) * g.C * 4);
  }
  __device__ void init_k(Kst& s, int k) const {
    s.k = k; s.px = k % g.PW; int t = k / g.PW; s.py = t % g.PH; s.n = t / g.PH;
    set_k(s);
  }
  __device__ void advance(Kst& s, int bk) const {       // incremental: adds only, no multiplies on the common path
    s.k += bk; s.px += bk;
    s.ix0 += bk * g.SX;
    s.base += (unsigned)(bk * g.SX * g.C * 4);
    while (s.px >= g.PW) {
      s.px -= g.PW;
      s.ix0 -= g.PW * g.SX;
      s.iy0 += g.SY;
      s.base += (unsigned)((g.SY * g.W - g.PW * g.SX) * g.C * 4);
      if (++s.py == g.PH) {
        s.py = 0; ++s.n;
        s.iy0 -= g.PH * g.SY;
        s.base += (unsigned)(((g.H - g.PH * g.SY) * g.W) * g.C * 4);
      }
    }
  }
  __device__ unsigned off(const Row& r, const Kst& s, int j) const {
    if (j == 0) {
      const int iy = s.iy0 + r.dy, ix = s.ix0 + r.dx;
      const bool ok = (s.k < K) & (r.r0 < rows) & ((unsigned)iy < (unsigned)g.H) & ((unsigned)ix < (unsigned)g.W);
      return ok ? s.base + r.roff : nbytes;
    }
    int ci = r.ci + j, kw = r.kw, kh = r.kh;
    while (ci >= g.C) { ci -= g.C; if (++kw == g.KW) { kw = 0; ++kh; } }
    const int iy = s.py * g.SY + kh * g.DY + g.OY0, ix = s.px * g.SX + kw * g.DX + g.OX0;
    const bool ok = (s.k < K) & (r.r0 + j < rows) & ((unsigned)iy < (unsigned)g.H) & ((unsigned)ix < (unsigned)g.W);
    return ok ? (unsigned)(((((long)s.n * g.H + iy) * g.W + ix) * g.C + ci) * 4) : nbytes;
  }
};

template <bool VEC, class L>
__device__ __forceinline__ f32x4 fetch(const L& l, __amdgpu_buffer_rsrc_t rs, const typename L::Row& r, const typename L::Kst& s) {
  if constexpr (VEC) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, l.off(r, s, 0), 0, 0));
  } else {
    f32x4 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, l.off(r, s, j), 0, 0));
    return v;
  }
}

// ------------------------------------------------------------------------------------------
// Epilogue description
// ------------------------------------------------------------------------------------------
struct Epi {
  float* C; long ldc; int M, N;
  const float* bias; const float* bias2;   // per-column, optional
  int act; float beta;                     // beta in {0,1}: C = act(acc + bias) + beta*C_old
  // ACT_SIGMOID_MASK_MUL extras (enhancer fc epilogue, enhance_model.py:156-164)
  const float* mul; float* mask_out; const int* lens; int T;
  // output row remap for the stride-2 data-gradient parity classes: row m=(n,i,j) ->
  // ((n*OHF + i*osy+ooy)*OWF + j*osx+oox)*ldc ; remap==0 => m*ldc
  int remap, PH, PW, OHF, OWF, osy, osx, ooy, oox;
  // split-K: when nsplit>1 raw accumulators go to ws[z][M][N]
  float* ws; int nsplit;
  // merged stride-2 data gradient: blockIdx.z = output parity class (ph,pw); per-class input offsets and
  // gathered-weight sets (re2e_conv_dgrad_s2)
  int ncls; int cls_oy0[2], cls_ox0[2]; long cls_wstride;
  int nomem;
  int zx;           // split-K: map K slices to XCDs (see the kernel)
  int nolog;        // the caller prints its own RE2E_IGEMM_LOG line (wino44.hip: direct-equivalent shape of the whole convolution)
};

template <class LA, class LB, class CF, bool VEC>
__global__ __launch_bounds__(CF::THREADS) void igemm_kernel(LA la, LB lb, Epi ep, int K) {
  constexpr int BM = CF::BM, BN = CF::BN, TH = CF::THREADS, BK = CF::BK, LDK = CF::LDK;
  constexpr int LDA_M = BM + 8, LDB_M = BN + 8;   // row strides of row-major (MMAJ) LDS tiles: 4*stride = 32 mod 64 banks, so the two k-halves of a b32 fragment read (lanes 0-31 / 32-63) use disjoint banks
  // TS ("transposing stage"): in the weight-gradient kernels BOTH operands are row-major (k = pixel is the slow axis).
  // There a float4 (4 consecutive rows at one k) is transposed while it is staged -- 4 ds_write_b32 at (row+j)*LDK + k with
  // lane -> (k = lane % BK, row quad = lane / BK), which spreads a wavefront's 64 writes over all 64 banks -- so that the
  // LDS tiles are k-major and the MFMA fragments are read with ds_read_b128 like the k-major operands (+3..5 % on dy^T x
  // and the conv weight gradients).  It is not used for the row-major B of dy W (its lanes would walk k through the small
  // weight matrix at a row-pitch stride: -2..12 %) nor on the scalar-load path.
  constexpr bool TS = VEC && !LA::KMAJ && !LB::KMAJ;
  constexpr bool AKL = LA::KMAJ || TS, BKL = LB::KMAJ || TS;            // operand tile is k-major in LDS
  constexpr int ASZ = AKL ? BM * LDK : BK * LDA_M;
  constexpr int BSZ = BKL ? BN * LDK : BK * LDB_M;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                 // [2][ASZ]
  float* Bs = smem + 2 * ASZ;       // [2][BSZ]

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / CF::WN, wn = wid % CF::WN;
  // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (private 4 MiB L2 each),
  // so give every XCD a contiguous chunk of the tile sequence (bijective remap), and walk the
  // sequence in groups of 8 M-tiles so that co-resident workgroups share A and B panels in L2.
  int tile_m, tile_n;
  int zsplit = blockIdx.z;
  {
    const int ntm = gridDim.x, ntn = gridDim.y, nwg = ntm * ntn;
    const int orig = blockIdx.y * ntm + blockIdx.x;
    int pid;
    if (ep.zx) {
      // split-K with a multiple of 8 slices (weight gradients): XCD = K slice.  Workgroups are dealt round-robin over the 8 XCDs in launch
      // order, so workgroup L of the launch (z-major) sits on XCD L % 8: give it slice (L % 8) + 8 * ((L / 8) / nwg) and tile (L / 8) % nwg.
      // All tiles of a slice then share one L2 and walk the same k-rows at the same time: every k-row of both operands is fetched once per
      // slice, i.e. once in all (the plain order keeps a TILE on one XCD for all slices: each XCD pulls its tiles' operand panels over the
      // whole K, 3-4x the operands' bytes in all).
      const int L = (int)blockIdx.z * nwg + orig;
      const int idx = L >> 3;
      zsplit = (L & 7) + 8 * (idx / nwg);
      pid = idx % nwg;
    } else {
      const int q8 = nwg >> 3, r8 = nwg & 7, xcd = orig & 7;
      pid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
    }
    const int GROUP = 8;
    const int per_group = GROUP * ntn;
    const int gid = pid / per_group;
    const int first_m = gid * GROUP;
    const int gsz = min(ntm - first_m, GROUP);
    tile_m = first_m + (pid % per_group) % gsz;
    tile_n = (pid % per_group) / gsz;
  }
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  // NB: the by-value kernel arguments (la, lb, ep) are never written: a modified argument struct is demoted
  // to scratch memory.  Per-class values live in scalars instead.
  int ooy = ep.ooy, oox = ep.oox, cls_dy = 0, cls_dx = 0;
  const float* pB = lb.p;
  if constexpr (LA::IS_CONVK) {
    if (ep.ncls) {
      const int cls = blockIdx.z, ph = cls >> 1, pw = cls & 1;
      cls_dy = ph ? ep.cls_oy0[1] : ep.cls_oy0[0]; cls_dx = pw ? ep.cls_ox0[1] : ep.cls_ox0[0];
      pB += cls * ep.cls_wstride;
      ooy = ph; oox = pw;
      zsplit = 0;
    }
  }
  // ep.nomem (RE2E_IGEMM_NOMEM=1, timing experiments only): zero-record descriptors -- every load is dropped by the
  // range check and returns 0, the instruction stream is unchanged: prices the memory side of the loop.
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(la.p), 0, ep.nomem ? 0 : la.nbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(pB), 0, ep.nomem ? 0 : lb.nbytes, 0x00020000);
  // split-K range
  const int nkt_total = (K + BK - 1) / BK;
  const int kt_per = (nkt_total + ep.nsplit - 1) / ep.nsplit;
  const int kt_begin = zsplit * kt_per;
  const int kt_end = min(nkt_total, kt_begin + kt_per);
  const int kbegin = kt_begin * BK;

  // float4 items per thread per tile
  constexpr int KQ = BK / 4;
  constexpr int AIT = BM * KQ / TH, BIT = BN * KQ / TH;
  static_assert(AIT >= 1 && BIT >= 1 && (BM * KQ) % TH == 0 && (BN * KQ) % TH == 0, "tile / thread-count mismatch");
  static_assert(TH % KQ == 0 && TH % BK == 0, "per-thread k state must be item-invariant");
  static_assert((!std::is_same<LA, DenseMG>::value && !std::is_same<LB, DenseMG>::value) || BK == DenseMG::LOOKAHEAD,
                "DenseMG prefetches the physical row of the next k-tile: the tile must advance by DenseMG::LOOKAHEAD");
  static_assert(AKL || TH % (BM / 4) == 0, "per-thread row state of a row-major A tile must be item-invariant");
  static_assert(BKL || TH % (BN / 4) == 0, "per-thread row state of a row-major B tile must be item-invariant");
  constexpr int NRA = AKL ? AIT : 1, NKA = AKL ? 1 : AIT;               // k-major staging: per-item rows + ONE k state per thread
  constexpr int NRB = BKL ? BIT : 1, NKB = BKL ? 1 : BIT;
  typename LA::Row ra_row[NRA];
  typename LA::Kst ra_k[NKA];
  typename LB::Row rb_row[NRB];
  typename LB::Kst rb_k[NKB];
  int a_lds[AIT], b_lds[BIT];
#pragma unroll
  for (int i = 0; i < AIT; ++i) {
    int idx = i * TH + tid;
    if (LA::KMAJ) {
      int r = idx / KQ, kq = idx % KQ;
      la.init_row(ra_row[AKL ? i : 0], m0 + r);
      if constexpr (LA::IS_CONVK) { if (ep.ncls) la.shift_row(ra_row[i], cls_dy, cls_dx); }
      if (i == 0) la.init_k(ra_k[0], kbegin + kq * 4);
      a_lds[i] = r * LDK + kq * 4;
    } else if (TS) {
      int kk = idx % BK, rq = idx / BK;
      la.init_row(ra_row[AKL ? i : 0], m0 + rq * 4);
      if (i == 0) la.init_k(ra_k[0], kbegin + kk);
      a_lds[i] = rq * 4 * LDK + kk;
    } else {
      int kk = idx / (BM / 4), rq = idx % (BM / 4);
      if (i == 0) la.init_row(ra_row[0], m0 + rq * 4);
      la.init_k(ra_k[AKL ? 0 : i], kbegin + kk);
      a_lds[i] = kk * LDA_M + rq * 4;
    }
  }
#pragma unroll
  for (int i = 0; i < BIT; ++i) {
    int idx = i * TH + tid;
    if (LB::KMAJ) {
      int r = idx / KQ, kq = idx % KQ;
      lb.init_row(rb_row[BKL ? i : 0], n0 + r);
      if (i == 0) lb.init_k(rb_k[0], kbegin + kq * 4);
      b_lds[i] = r * LDK + kq * 4;
    } else if (TS) {
      int kk = idx % BK, rq = idx / BK;
      lb.init_row(rb_row[BKL ? i : 0], n0 + rq * 4);
      if (i == 0) lb.init_k(rb_k[0], kbegin + kk);
      b_lds[i] = rq * 4 * LDK + kk;
    } else {
      int kk = idx / (BN / 4), rq = idx % (BN / 4);
      if (i == 0) lb.init_row(rb_row[0], n0 + rq * 4);
      lb.init_k(rb_k[BKL ? 0 : i], kbegin + kk);
      b_lds[i] = kk * LDB_M + rq * 4;
    }
  }

  f32x16 acc[CF::TM][CF::TN];
#pragma unroll
  for (int i = 0; i < CF::TM; ++i)
#pragma unroll
    for (int j = 0; j < CF::TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // Pipeline (one register set, two LDS buffers): tile t+1 is written to LDS right AFTER the barrier that opens iteration
  // t, the loads of tile t+2 are issued immediately behind it, then tile t is computed -- so every global load has a whole
  // MFMA block plus a barrier to land before its LDS store needs it.
  f32x4 ra[AIT], rb[BIT];
  auto fetch_tile = [&]() {
#pragma unroll
    for (int i = 0; i < AIT; ++i) ra[i] = fetch<VEC>(la, rsA, ra_row[AKL ? i : 0], ra_k[AKL ? 0 : i]);
#pragma unroll
    for (int i = 0; i < BIT; ++i) rb[i] = fetch<VEC>(lb, rsB, rb_row[BKL ? i : 0], rb_k[BKL ? 0 : i]);
  };
  auto advance_k = [&]() {
#pragma unroll
    for (int i = 0; i < NKA; ++i) la.advance(ra_k[i], BK);
#pragma unroll
    for (int i = 0; i < NKB; ++i) lb.advance(rb_k[i], BK);
  };
  auto store_tile = [&](int buf) {
    float* An = As + buf * ASZ;
    float* Bn = Bs + buf * BSZ;
#pragma unroll
    for (int i = 0; i < AIT; ++i) {
      if (!TS) *reinterpret_cast<f32x4*>(An + a_lds[i]) = ra[i];
      else {
#pragma unroll
        for (int j = 0; j < 4; ++j) An[a_lds[i] + j * LDK] = ra[i][j];
      }
    }
#pragma unroll
    for (int i = 0; i < BIT; ++i) {
      if (!TS) *reinterpret_cast<f32x4*>(Bn + b_lds[i]) = rb[i];
      else {
#pragma unroll
        for (int j = 0; j < 4; ++j) Bn[b_lds[i] + j * LDK] = rb[i][j];
      }
    }
  };
  if (kt_begin < kt_end) {
    fetch_tile();
    store_tile(0);
    if (kt_begin + 1 < kt_end) { advance_k(); fetch_tile(); }
  }

  const int lr = lane & 31, lh = lane >> 5;
  // (measured and rejected: skipping the MFMA block of a wave whose slab lies outside a ragged matrix edge -- the
  //  per-iteration branch costs every kernel 10-20 %)
  // PIPE (x W^T with 16-byte loads: both operands k-major): the loop body is ONE basic block -- tile kt+1 is staged and tile
  // kt+2 fetched unconditionally (beyond K the loaders return the out-of-range offset: zeros, no traffic; the spare LDS buffer
  // of the last iteration is never read) -- and its source order plus the scheduling directives at its end put the address
  // arithmetic, the loads and the second k-group's fragment reads BETWEEN the MFMAs.  Left to itself the compiler emits all of
  // them in front of the first MFMA, and with one 8-wave workgroup per CU both waves of a SIMD sit behind the same barrier:
  // the matrix pipe idles for the prologue of every k-step.  Same GPU session, 256x128 tile: 12800x2048x2560 100.2 -> 104.1,
  // 25600x1024x512 93.4 -> 96.1, 6400x4233x512 95.8 -> 99.7 TFLOP/s.  NOT for a row-major B (x W: 4 ds_read_b32 per fragment
  // value, 112.8 -> 95.4), the transposing stage of the weight gradients (+-1 %) or the im2col A operand (its address arithmetic
  // made branch-free for the purpose: D conv2 forward 116.7 -> 115.8, its data gradient 96.2 -> 93.3).
  constexpr bool PIPE = VEC && BK == 16 && std::is_same<LA, DenseK>::value && std::is_same<LB, DenseK>::value;
  int cur = 0;
  auto read_frags = [&](const float* Ac, const float* Bc, int q, f32x4 (&fa)[CF::TM], f32x4 (&fb)[CF::TN]) {
#pragma unroll
    for (int i = 0; i < CF::TM; ++i) {
      int r = (wm * CF::TM + i) * 32 + lr;
      if (AKL) fa[i] = *reinterpret_cast<const f32x4*>(Ac + r * LDK + q * 8 + lh * 4);
      else {
#pragma unroll
        for (int j = 0; j < 4; ++j) fa[i][j] = Ac[(q * 8 + lh * 4 + j) * LDA_M + r];
      }
    }
#pragma unroll
    for (int i = 0; i < CF::TN; ++i) {
      int r = (wn * CF::TN + i) * 32 + lr;
      if (BKL) fb[i] = *reinterpret_cast<const f32x4*>(Bc + r * LDK + q * 8 + lh * 4);
      else {
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[i][j] = Bc[(q * 8 + lh * 4 + j) * LDB_M + r];
      }
    }
  };
  auto mfma_group = [&](const f32x4 (&fa)[CF::TM], const f32x4 (&fb)[CF::TN]) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int a = 0; a < CF::TM; ++a)
#pragma unroll
        for (int b = 0; b < CF::TN; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a][j], fb[b][j], acc[a][b], 0, 0, 0);
  };
  for (int kt = kt_begin; kt < kt_end; ++kt) {
    __syncthreads();                                   // tile kt is complete in buffer cur; buffer cur^1 is free
    const float* Ac = As + cur * ASZ;
    const float* Bc = Bs + cur * BSZ;
    if constexpr (PIPE) {
      // source order = the issue order wanted (LDS accesses may alias for the compiler, so it keeps them in this order):
      // fragments of the first k-group, THEN the staging of tile kt+1 / the address arithmetic / the loads of tile kt+2 and
      // the later k-groups' fragment reads, all of which the directives below place between the MFMAs of the first k-group
      static_assert(BK / 8 == 2, "two k-groups per tile");
      f32x4 fa0[CF::TM], fb0[CF::TN], fa1[CF::TM], fb1[CF::TN];
      read_frags(Ac, Bc, 0, fa0, fb0);
      store_tile(cur ^ 1);
      advance_k();
      fetch_tile();
      read_frags(Ac, Bc, 1, fa1, fb1);
      mfma_group(fa0, fb0);
      mfma_group(fa1, fb1);
      constexpr int NM = CF::TM * CF::TN * (BK / 2);
      constexpr int NDW = (AIT + BIT) * (TS ? 4 : 1), NVM = AIT + BIT;
      constexpr int FR = (AKL ? CF::TM : 4 * CF::TM) + (BKL ? CF::TN : 4 * CF::TN);      // fragment reads per k-group
      __builtin_amdgcn_sched_group_barrier(0x100, FR, 0);
#pragma unroll
      for (int m = 0; m < NM; ++m) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (m < NDW) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        if (m >= 1 && m < 1 + 2 * NVM) __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
        if (m >= 2 && m < 2 + 2 * NVM && (m & 1) == 0) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        if (m >= NM / 4 && m < NM / 4 + FR) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
    } else {
      if (kt + 1 < kt_end) {
        store_tile(cur ^ 1);
        if (kt + 2 < kt_end) { advance_k(); fetch_tile(); }
      }
#pragma unroll
      for (int q = 0; q < BK / 8; ++q) {
        f32x4 fa[CF::TM], fb[CF::TN];
        read_frags(Ac, Bc, q, fa, fb);
        mfma_group(fa, fb);
      }
    }
    cur ^= 1;
  }

  // ---------------------------------- epilogue ----------------------------------
  // A lane holds, per 32x32 tile, ONE output column (lr) and 16 rows ((r&3) + 8*(r>>2) + 4*lh).  The variants below are separate
  // loop nests selected by wave-uniform branches OUTSIDE the loops: a per-element `if (beta) v += C[off]` next to the store of
  // the same array made the compiler wait for every store before the next element's possible load (64 serialised global
  // round trips per lane, the largest single cost of a short-K tile), and a per-element activation switch + row remap
  // division tripled the code (20k lines of ISA per instance).  Row offsets are computed once per row, not per element.
  if (ep.nsplit > 1) {
    float* W = ep.ws + (long)zsplit * ep.M * ep.N;
#pragma unroll
    for (int a = 0; a < CF::TM; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + (wm * CF::TM + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        float* wr = W + (long)row * ep.N;
#pragma unroll
        for (int b = 0; b < CF::TN; ++b) {
          const int col = n0 + (wn * CF::TN + b) * 32 + lr;
          if (row < ep.M && col < ep.N) wr[col] = acc[a][b][r];
        }
      }
    return;
  }
  float bv[CF::TN];
  int colv[CF::TN];
#pragma unroll
  for (int b = 0; b < CF::TN; ++b) {
    colv[b] = n0 + (wn * CF::TN + b) * 32 + lr;
    float v = 0.f;
    if (colv[b] < ep.N) {
      if (ep.bias) v += ep.bias[colv[b]];
      if (ep.bias2) v += ep.bias2[colv[b]];
    }
    bv[b] = v;
  }
  // offset of a row's first element, or -1 when the row does not exist (ragged M; phantom row of a parity class)
  auto row_base = [&](int row) -> long {
    if (row >= ep.M) return -1;
    if (!ep.remap) return (long)row * ep.ldc;
    const int j = row % ep.PW; const int t = row / ep.PW; const int i = t % ep.PH; const int n = t / ep.PH;
    const int oy = i * ep.osy + ooy, ox = j * ep.osx + oox;
    if (oy >= ep.OHF || ox >= ep.OWF) return -1;
    return (((long)n * ep.OHF + oy) * ep.OWF + ox) * ep.ldc;
  };
  if (ep.act == RE2E_ACT_SIGMOID_MASK_MUL) {            // enhancer fc epilogue (never split, never remapped, beta 0)
    // three distinct tensors: without the no-alias promise every `mul` load waits for the two stores in front of it
    const float* __restrict__ mulp = ep.mul;
    float* __restrict__ maskp = ep.mask_out;
    float* __restrict__ outp = ep.C;
#pragma unroll
    for (int a = 0; a < CF::TM; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + (wm * CF::TM + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (row >= ep.M) continue;
        const int bi = row / ep.T, t = row - bi * ep.T;
        const bool live = t < ep.lens[bi];
        const long rb = (long)row * ep.ldc;
#pragma unroll
        for (int b = 0; b < CF::TN; ++b) {
          if (colv[b] >= ep.N) continue;
          const float sg = live ? sigmoidf_(acc[a][b][r] + bv[b]) : 0.f;
          maskp[rb + colv[b]] = sg;
          outp[rb + colv[b]] = sg * mulp[rb + colv[b]];
        }
      }
    return;
  }
  if (ep.beta != 0.f) {                                  // accumulate: the 16 old values of a tile row block are loaded before any store
#pragma unroll
    for (int a = 0; a < CF::TM; ++a)
#pragma unroll
      for (int b = 0; b < CF::TN; ++b) {
        float old[16];
        long offs[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const long rb = row_base(m0 + (wm * CF::TM + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh);
          offs[r] = (rb >= 0 && colv[b] < ep.N) ? rb + colv[b] : -1;
          old[r] = offs[r] >= 0 ? ep.C[offs[r]] : 0.f;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (offs[r] >= 0) ep.C[offs[r]] = apply_act(acc[a][b][r] + bv[b], ep.act) + old[r];
      }
    return;
  }
  if (ep.act == RE2E_ACT_TANH || ep.act == RE2E_ACT_SIGMOID) {
    const bool th = ep.act == RE2E_ACT_TANH;
#pragma unroll
    for (int a = 0; a < CF::TM; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const long rb = row_base(m0 + (wm * CF::TM + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh);
        if (rb < 0) continue;
#pragma unroll
        for (int b = 0; b < CF::TN; ++b) {
          const float v = acc[a][b][r] + bv[b];
          if (colv[b] < ep.N) ep.C[rb + colv[b]] = th ? tanhf_(v) : sigmoidf_(v);
        }
      }
    return;
  }
  {
    // none / ReLU / LeakyReLU(0.2) as one select: v < 0 ? slope * v : v with slope 1 / 0 / 0.2
    const float slope = ep.act == RE2E_ACT_RELU ? 0.f : (ep.act == RE2E_ACT_LRELU ? 0.2f : 1.f);
    const bool relu = ep.act == RE2E_ACT_RELU;
#pragma unroll
    for (int a = 0; a < CF::TM; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const long rb = row_base(m0 + (wm * CF::TM + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh);
        if (rb < 0) continue;
#pragma unroll
        for (int b = 0; b < CF::TN; ++b) {
          const float v = acc[a][b][r] + bv[b];
          if (colv[b] < ep.N) ep.C[rb + colv[b]] = v < 0.f ? (relu ? 0.f : slope * v) : v;      // torch semantics: NaN stays NaN, ReLU(-inf) = 0 (fmaxf(v, 0 * v) gave -inf)
        }
      }
  }
}

// Deterministic split-K reduce: C[perm(m,n)] = act(sum_z ws[z][m][n] + bias) + beta*C.  perm: plain
// (m*ldc+n) or the conv weight layout (rows m=(kh,kw,ci), cols n=co -> W[co][ci][kh][kw]).
__global__ void splitk_reduce_kernel(const float* ws, int nsplit, int M, int N, float* C, long ldc, float beta,
                                     int conv_perm, int Cin, int KHW, const float* bias, const float* bias2, int act) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long tot = (long)M * N;
  if (i >= tot) return;
  float s = 0.f;
  for (int z = 0; z < nsplit; ++z) s += ws[(long)z * tot + i];
  int m = (int)(i / N), n = (int)(i % N);
  long off;
  if (conv_perm) { int ci = m % Cin; int tap = m / Cin; off = ((long)n * Cin + ci) * KHW + tap; }
  else off = (long)m * ldc + n;
  if (bias) s += bias[n];
  if (bias2) s += bias2[n];
  s = apply_act(s, act);
  if (beta != 0.f) s += C[off];
  C[off] = s;
}

// Same reduce for FEW outputs and MANY slabs (thin-channel weight gradients: 576 outputs x 1024 slabs):
// 16 slab lanes per output, combined through LDS in a fixed order (still deterministic).
__global__ __launch_bounds__(1024) void splitk_reduce_wide_kernel(const float* ws, int nsplit, int M, int N, float* C, long ldc,
                                                                  float beta, int conv_perm, int Cin, int KHW) {
  __shared__ float part[16][64];
  const int e = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const long tot = (long)M * N;
  const long i = (long)blockIdx.x * 64 + e;
  float s = 0.f;
  if (i < tot)
    for (int z = sl; z < nsplit; z += 16) s += ws[(long)z * tot + i];
  part[sl][e] = s;
  __syncthreads();
  if (sl != 0 || i >= tot) return;
  s = 0.f;
#pragma unroll
  for (int q = 0; q < 16; ++q) s += part[q][e];
  int m = (int)(i / N), n = (int)(i % N);
  long off;
  if (conv_perm) { int ci = m % Cin; int tap = m / Cin; off = ((long)n * Cin + ci) * KHW + tap; }
  else off = (long)m * ldc + n;
  if (beta != 0.f) s += C[off];
  C[off] = s;
}

template <class LA, class LB, class CF, bool VEC>
int launch_igemm(const LA& la, const LB& lb, Epi ep, int K, hipStream_t st) {
  constexpr bool TS = VEC && !LA::KMAJ && !LB::KMAJ;      // see igemm_kernel
  constexpr int ASZ = (LA::KMAJ || TS) ? CF::BM * CF::LDK : CF::BK * (CF::BM + 8);
  constexpr int BSZ = (LB::KMAJ || TS) ? CF::BN * CF::LDK : CF::BK * (CF::BN + 8);
  size_t lds = (size_t)2 * (ASZ + BSZ) * sizeof(float);
  static const size_t lds_floor = exp_env("RE2E_IGEMM_LDS_FLOOR") ? (size_t)atol(exp_env("RE2E_IGEMM_LDS_FLOOR")) : 0;   // occupancy experiments
  if (lds < lds_floor) lds = lds_floor;
  static LdsLimit lim;
  lim.ensure(reinterpret_cast<const void*>(&igemm_kernel<LA, LB, CF, VEC>), lds);
  dim3 grid(cdiv(ep.M, CF::BM), cdiv(ep.N, CF::BN), ep.ncls ? ep.ncls : ep.nsplit);
  static const bool nomem = exp_env("RE2E_IGEMM_NOMEM") != nullptr;
  ep.nomem = nomem ? 1 : 0;
  {
    const char* zx = exp_env("RE2E_TN_XCD_KSLICE");      // experiments build: 0 = tiles keep their XCD for all slices (rounds 1-4)
    ep.zx = (ep.nsplit > 1 && (ep.nsplit & 7) == 0 && !ep.ncls && !(zx && atoi(zx) == 0)) ? 1 : 0;
  }
  static const bool log_calls = getenv("RE2E_IGEMM_LOG") != nullptr;   // tools/igemm_table.py joins this with a kernel trace
  if (log_calls && !ep.nolog)
    fprintf(stderr, "[igemm] A=%s B=%s tile=%dx%dx%d vec=%d M=%d N=%d K=%d splits=%d\n", LA::NAME, LB::NAME, CF::BM, CF::BN,
            CF::BK, (int)VEC, ep.ncls ? ep.M * ep.ncls : ep.M, ep.N, K, ep.nsplit);   // M = rows of ALL parity classes of the launch
  hipLaunchKernelGGL((igemm_kernel<LA, LB, CF, VEC>), grid, dim3(CF::THREADS), lds, st, la, lb, ep, K);
  return 0;
}

constexpr int BKD = 16;              // k-tile: 16 keeps LDS <= 46 KB per block => 2-3 blocks per CU
using C128 = Cfg<2, 2, 2, 2, BKD>;    // 128 x 128
using C256x64 = Cfg<4, 1, 2, 2, BKD>;
using C192x64 = Cfg<2, 2, 3, 1, BKD>;   // 576-row weight gradients (9 taps x 64 channels): 3 exact tiles instead of 2.25 of 256
using C256x32 = Cfg<4, 1, 2, 1, 32>;   // BN=32 needs BK=32 to give every thread a B item
using C32x128 = Cfg<1, 4, 1, 1, 32>;   // skinny (M<=32) GEMMs of the decoder loop, always split-K
// tuning variants (tools/bench_gemm.py, env RE2E_IGEMM_VARIANT): 1 = BK 32, 2 = 8-wave 256x128 tile
using C128b = Cfg<2, 2, 2, 2, 32>;
using C256x64b = Cfg<4, 1, 2, 2, 32>;
using C256x128 = Cfg<4, 2, 2, 2, 16>;
using C64 = Cfg<2, 2, 1, 1, BKD>;      // 64 x 64: the row tail of a Linear forward whose last round of 256x128 tiles would be mostly empty

inline int igemm_variant() {
  static const int v = exp_env("RE2E_IGEMM_VARIANT") ? atoi(exp_env("RE2E_IGEMM_VARIANT")) : 0;
  return v;
}

// big-tile launch.  WIDE = the 8-wave 256x128 tile is the measured default for this operand combination
// (tools/bench_gemm.py on MI355X: +5..35 % for Linear-forward / weight-gradient GEMMs and conv forward /
// data-gradient with Cout >= 128; the 4-wave 128x128 tile stays better for NN and conv weight-gradient).
// RE2E_IGEMM_VARIANT = 1 (BK 32) / 2 (force wide) / 3 (force 128x128) override for tuning.  On a FILLER stream the 4-wave tile is
// used whatever its stand-alone rate: training step 72.61 -> 72.32 ms (3 + 3 runs, one GPU session).
// the 8-wave tile everywhere but on streams marked as fillers (core.hip: re2e_stream_role)
inline bool wide_allowed(hipStream_t st) { return !re2e_stream_is_filler(st); }

template <class LA, class LB, bool V, bool WIDE>
void launch_big(const LA& la, const LB& lb, Epi& ep, int K, hipStream_t st) {
  if constexpr (V) {
    int v = igemm_variant();
    if constexpr (!std::is_same<LA, DenseMG>::value && !std::is_same<LB, DenseMG>::value) {      // (DenseMG looks one 16-row k-tile ahead: BK 16 only)
      if (v == 1) { launch_igemm<LA, LB, C128b, V>(la, lb, ep, K, st); return; }
    }
    bool wide = v == 2 ? true : (v == 3 ? false : (WIDE && ep.M >= 2048 && wide_allowed(st)));
    if (wide) { launch_igemm<LA, LB, C256x128, V>(la, lb, ep, K, st); return; }
  }
  launch_igemm<LA, LB, C128, V>(la, lb, ep, K, st);
}
template <class LA, class LB, bool V>
void launch_n64(const LA& la, const LB& lb, Epi& ep, int K, hipStream_t st) {
  if constexpr (V) {
    if (igemm_variant() == 1) { launch_igemm<LA, LB, C256x64b, V>(la, lb, ep, K, st); return; }
  }
  launch_igemm<LA, LB, C256x64, V>(la, lb, ep, K, st);
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// Workgroups of one tile shape that are resident at once: (workgroups per CU the register file admits -- 146 VGPRs => 3 waves per
// SIMD for the 8-wave 256x128 tile, i.e. ONE workgroup; 2 x 4 waves for 128x128 and 256x64; 3 x 4 for the 112-register
// 192x64 and the 256x32 tiles: csrc/build/igemm.resources) x CUs.  A device query would be exact, but the split count must be
// the same in re2e_*_workspace_bytes and in the launch, so it is this table.
inline long tile_slots(int bm, int bn) {
  static const int cus = [] { int dev = 0, n = 256; if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n; }();
  const int per_cu = (bm == 256 && bn == 128) ? 1 : ((bm == 192 && bn == 64) || (bm == 256 && bn == 32)) ? 3 : 2;
  return (long)per_cu * cus;
}

// Split-K count of a product with few output tiles: the number of K slices (each >= 16 k-tiles) that minimises
//   compute time / (fill of the resident slots over the rounds of workgroups)  +  slab traffic (one write + one read per slice)
// -- not "about 768 workgroups": 9 tiles x 86 slices = 774 workgroups on 512 slots ran two rounds with the second 51 % empty
// (conv2_2 weight gradient 3.0 -> 2.5 ms), 160 x 5 = 800 on 256 slots 3.1 rounds (BLSTMP layer-0 weight gradient 1.36 -> 1.14 ms).
int pick_splits(int M, int N, int K, int bm, int bn) {
  const long tiles = (long)cdiv(M, bm) * cdiv(N, bn);
  const int nkt = cdiv(K, BKD);
  const long slots = tile_slots(bm, bn);
  if (tiles >= slots * 3 / 4 || nkt < 32) return 1;
  long maxs = nkt / 16;          // >= 16 k-tiles (256 k) per split
  if (maxs > 512) maxs = 512;
  if (maxs < 1) maxs = 1;
  const double compute = 2.0 * M * N * (double)K / 110e12;            // seconds at the engine's typical rate
  long best_s = 1;
  double best = 1e30;
  for (long sp = 1; sp <= maxs; ++sp) {
    const long w = tiles * sp, rounds = (w + slots - 1) / slots;
    const double eff = (double)w / (double)(rounds * slots);
    const double t = compute / eff + (sp > 1 ? (double)sp * M * N * 8.0 / 3e12 : 0.0);
    if (t < best - 1e-12) { best = t; best_s = sp; }
  }
  static const int cap_kt = exp_env("RE2E_SPLIT_MAXKT") ? atoi(exp_env("RE2E_SPLIT_MAXKT")) : 0;   // experiment: bound a workgroup's lifetime
  if (cap_kt > 0 && nkt / best_s > cap_kt) {
    const long rounds = (nkt / best_s + cap_kt - 1) / cap_kt;         // keep the fill of the best choice: whole multiples of its workgroup count
    long sp = best_s * rounds;
    if (sp > maxs) sp = maxs;
    best_s = sp;
  }
  return (int)best_s;
}

// skinny path (M <= 32): latency-bound, so spread K over many workgroups (>= 2 k-tiles of 32 each)
int pick_splits_skinny(int N, int K) {
  long tiles = cdiv(N, 128);
  int nkt = cdiv(K, 32);
  long want = (192 + tiles - 1) / tiles;
  long maxs = nkt / 2;
  long s = want < maxs ? want : maxs;
  if (s < 1) s = 1;
  if (s > 64) s = 64;
  return (int)s;
}

inline bool use_skinny(int transa, int M, int N, int K) { return !transa && M <= 32 && K >= 128; }

int gemm_splits(int transa, int transb, int M, int N, int K) {
  if (use_skinny(transa, M, N, K)) return pick_splits_skinny(N, K);
  if (transa && !transb) return pick_splits(M, N, K, M >= 2048 ? 256 : 128, 128);
  // x W^T / dy W with few output tiles and a long K (decoder output layer gradient: 1312 x 300 x 4233)
  if ((long)cdiv(M, 128) * cdiv(N, 128) <= 64) return pick_splits(M, N, K, 128, 128);
  // many rows, few columns, long K (round 4: the CTC projection's input gradient 6400 x 512 x 4240 is 100 tiles of 256x128 on 256 CUs -- 39 % of
  // the chip at 51 TFLOP/s, 0.55 ms on the path the main stream waits for at the encoder output): K slices fill the round
  if (M >= 2048 && K >= 1024 && (long)cdiv(M, 256) * cdiv(N, 128) * 20 <= tile_slots(256, 128) * 11) return pick_splits(M, N, K, 256, 128);
  return 1;
}

// ---- skinny GEMM (M <= 32) with the K split INSIDE the workgroup -------------------------------------------------------
// The decoder loop's GEMMs (32 x 1200 x 512, 32 x 512 x 1200, ...) are latency-bound: the tiled engine ran them as split-K over
// workgroups plus a reduce launch (6 + 1.7 + 6 us per product, 205 products per training step).  Here one workgroup owns a
// 32 x 32 output tile, its 8 wavefronts take an eighth of K each (operand fragments straight from global memory, all loads of a
// pass issued before its MFMAs), the eight partial tiles meet in LDS and the epilogue (bias, bias2, beta) runs in the same launch.
// TB: B is (N, K) row-major (x W^T); otherwise (K, N).  Fixed summation order: deterministic.
template <bool TB, bool VEC>      // VEC: 16-byte aligned operands, leading dimensions and K multiples of 4 -> float4 fragment loads
__global__ __launch_bounds__(512) void skinny_gemm_kernel(const float* __restrict__ A, long lda, const float* __restrict__ B, long ldb,
                                                          float* C, long ldc, int M, int N, int K, const float* __restrict__ bias,
                                                          const float* __restrict__ bias2, float beta) {
  __shared__ float red[8][32][33];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, lr = lane & 31, lh = lane >> 5;
  const int n0 = blockIdx.x * 32;
  const int kq = (K + 7) / 8;                       // k-groups of 8
  const int per = (kq + 7) / 8;                     // k-groups per wavefront
  const int q0 = wid * per, q1 = min(kq, q0 + per);
  const int row = lr, col = n0 + lr;
  const bool rok = row < M, cok = col < N;
  const float* ap = A + (long)(rok ? row : 0) * lda;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (int qb = q0; qb < q1; qb += 4) {             // four k-groups per pass
    float av[4][4], bv[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int k = 8 * (qb + u) + 4 * lh;
      const bool qok = qb + u < q1;
      if (VEC) {
        const bool kok = qok && k < K;
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        const f32x4 a4 = (kok && rok) ? *reinterpret_cast<const f32x4*>(ap + k) : zero;
        f32x4 b4 = zero;
        if (TB) { if (kok && cok) b4 = *reinterpret_cast<const f32x4*>(B + (long)col * ldb + k); }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          av[u][j] = a4[j];
          bv[u][j] = TB ? b4[j] : ((kok && cok) ? B[(long)(k + j) * ldb + col] : 0.f);
        }
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const bool kok = qok && k + j < K;
          av[u][j] = (kok && rok) ? ap[k + j] : 0.f;
          bv[u][j] = (kok && cok) ? (TB ? B[(long)col * ldb + k + j] : B[(long)(k + j) * ldb + col]) : 0.f;
        }
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][j], bv[u][j], acc, 0, 0, 0);
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) red[wid][(r & 3) + 8 * (r >> 2) + 4 * lh][lr] = acc[r];
  __syncthreads();
  for (int i = tid; i < 32 * 32; i += 512) {
    const int m = i >> 5, n = n0 + (i & 31);
    if (m < M && n < N) {
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) v += red[w][m][i & 31];
      if (bias) v += bias[n];
      if (bias2) v += bias2[n];
      float* c = C + (long)m * ldc + n;
      *c = (beta != 0.f ? *c : 0.f) + v;
    }
  }
}

// Two products that share the skinny left operand in ONE launch: C1[M,N1] = A[M,K] B1[K,N1] and C2[M,N2] = A B2[K,N2] (both B
// row-major (K,N)).  The decoder's backward step needs d(ctx) = dgates W_ih[:, Dd:] and d(z) = dgates W_hh from the same
// dgates (e2e_decoder.py:131 backward): two launches of 16 and 10 workgroups become one of 26.  Same in-workgroup K split and
// fixed summation order as skinny_gemm_kernel.
__global__ __launch_bounds__(512) void skinny_gemm2_kernel(const float* __restrict__ A, long lda, int M, int K, const float* __restrict__ B1, long ldb1,
                                                           int N1, float* C1, long ldc1, const float* __restrict__ B2, long ldb2, int N2,
                                                           float* C2, long ldc2) {
  __shared__ float red[8][32][33];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, lr = lane & 31, lh = lane >> 5;
  const int nt1 = (N1 + 31) / 32;
  const bool second = (int)blockIdx.x >= nt1;
  const float* B = second ? B2 : B1;
  const long ldb = second ? ldb2 : ldb1;
  const int N = second ? N2 : N1;
  float* C = second ? C2 : C1;
  const long ldc = second ? ldc2 : ldc1;
  const int n0 = (second ? (int)blockIdx.x - nt1 : (int)blockIdx.x) * 32;
  const int kq = (K + 7) / 8, per = (kq + 7) / 8;
  const int q0 = wid * per, q1 = min(kq, q0 + per);
  const int col = n0 + lr;
  const bool rok = lr < M, cok = col < N;
  const float* ap = A + (long)(rok ? lr : 0) * lda;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  // A's rows are one per lane: element-wise loads of it touch 32 cache lines per instruction for 4 useful bytes each (16 such
  // instructions per pass; round-3 trace: 26 us per launch at K = 1200, a link of the decoder's backward chain).  With K and lda
  // multiples of 4 the lane's four consecutive k are ONE 16-byte load.
  const bool avec = (K & 3) == 0 && (lda & 3) == 0 && (reinterpret_cast<uintptr_t>(A) & 15) == 0;
  for (int qb = q0; qb < q1; qb += 4) {
    float av[4][4], bv[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int k = 8 * (qb + u) + 4 * lh;
      const bool qok = qb + u < q1;
      if (avec) {
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        const f32x4 a4 = (qok && k < K && rok) ? *reinterpret_cast<const f32x4*>(ap + k) : zero;
#pragma unroll
        for (int j = 0; j < 4; ++j) av[u][j] = a4[j];
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bool kok = qok && k + j < K;
        if (!avec) av[u][j] = (kok && rok) ? ap[k + j] : 0.f;
        bv[u][j] = (kok && cok) ? B[(long)(k + j) * ldb + col] : 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][j], bv[u][j], acc, 0, 0, 0);
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) red[wid][(r & 3) + 8 * (r >> 2) + 4 * lh][lr] = acc[r];
  __syncthreads();
  for (int i = tid; i < 32 * 32; i += 512) {
    const int m = i >> 5, n = n0 + (i & 31);
    if (m < M && n < N) {
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) v += red[w][m][i & 31];
      C[(long)m * ldc + n] = v;
    }
  }
}

}  // namespace

// ============================================================================================
// C ABI
// ============================================================================================
extern "C" int re2e_gemm_skinny2(int M, int K, const float* A, long lda, const float* B1, long ldb1, int N1, float* C1, long ldc1,
                                 const float* B2, long ldb2, int N2, float* C2, long ldc2, hipStream_t stream) {
  RE2E_CHECK_ARG(A && B1 && B2 && C1 && C2, "null operand");
  RE2E_CHECK_ARG(M > 0 && M <= 32 && K > 0 && N1 > 0 && N2 > 0, "needs 1 <= M <= 32 rows");
  hipLaunchKernelGGL(skinny_gemm2_kernel, dim3(cdiv(N1, 32) + cdiv(N2, 32)), dim3(512), 0, stream, A, lda, M, K, B1, ldb1, N1, C1, ldc1, B2, ldb2, N2,
                     C2, ldc2);
  RE2E_LAUNCH_CHECK();
  return RE2E_OK;
}

extern "C" size_t re2e_gemm_workspace_bytes(int transa, int transb, int M, int N, int K) {
  if (transa && transb) return 0;
  int s = gemm_splits(transa, transb, M, N, K);
  size_t b = s > 1 ? (size_t)s * M * N * sizeof(float) : 0;
  if (!transa && transb) { const size_t b2 = gemm_nt2_workspace_bytes(M, N, K); if (b2 > b) b = b2; }
  if (transa && !transb) { const size_t b2 = gemm_tn2_workspace_bytes(M, N, K); if (b2 > b) b = b2; }
  return b;
}

// bytes spanned by a (outer x inner) row-major view with leading dimension ld
static inline unsigned kbytes(long outer, long ld, long inner) { return (unsigned)(((outer - 1) * ld + inner) * 4); }
static inline bool fits32(long outer, long ld, long inner) { return ((outer - 1) * ld + inner) * 4 < 0xFFFFFFF0L; }

// Rows that stay with the 256x128 tiles of a Linear forward (see gemm_dispatch); M = no split.
static int row_tail_split(int M, int N) {
  if (M < 2048) return M;                                   // launch_big uses the wide tile from 2048 rows
  const long tn = cdiv(N, 128), tm = cdiv(M, 256), slots = tile_slots(256, 128);
  const long total = tm * tn, full = total / slots, rem = total % slots;
  if (full < 1 || rem == 0 || rem * 2 >= slots) return M;
  const long tm_main = full * slots / tn;                   // whole tile rows inside the full rounds
  if (tm_main * 256 < 2048 || tm_main >= tm) return M;     // (launch_big keeps the wide tile from 2048 rows)
  const long m1 = tm_main * 256, tail_wgs = (long)cdiv(M - (int)m1, 64) * cdiv(N, 64);
  // a 64x64 workgroup does 1/8 of a big tile's work at about half its rate: a round of them costs ~0.25 big rounds
  const double t_split = (double)cdiv((int)(tm_main * tn), (int)slots) + 0.25 * (double)cdiv((int)tail_wgs, 4 * (int)slots);   // >= 4 of them per CU
  return t_split + 0.2 < (double)(full + 1) ? (int)m1 : M;
}

template <bool V>
static void gemm_dispatch(int transa, int transb, int M, int N, int K, const float* A, long lda, const float* B, long ldb, Epi& ep,
                          hipStream_t st) {
  const bool skinny = use_skinny(transa, M, N, K);
  if (!transa && transb) {          // C = A[M,K] * B[N,K]^T   (Linear forward)
    DenseK la{A, kbytes(M, lda, K), lda, M, K};
    DenseK lb{B, kbytes(N, ldb, K), ldb, N, K};
    if (skinny) launch_igemm<DenseK, DenseK, C32x128, V>(la, lb, ep, K, st);
    else {
      // Round-filling row split: 12800 x 2048 is 800 tiles of 256x128 on 256 resident workgroups -- 3.1 rounds, the 4th one
      // 12 % full.  The rows of the last, mostly empty round go to a second launch of 64x64 tiles instead (256 small
      // workgroups: every CU gets one), the launch of the big tiles ends on a full round.
      static const bool no_tail = exp_env("RE2E_NO_ROW_TAIL") != nullptr;
      const int m1 = (V && !no_tail && igemm_variant() == 0 && ep.nsplit == 1 && ep.act != RE2E_ACT_SIGMOID_MASK_MUL && wide_allowed(st)) ? row_tail_split(M, N) : M;
      if (m1 < M) {
        Epi e1 = ep, e2 = ep;
        e1.M = m1;
        DenseK la1{A, kbytes(m1, lda, K), lda, m1, K};
        launch_big<DenseK, DenseK, V, true>(la1, lb, e1, K, st);
        e2.M = M - m1; e2.C = ep.C + (long)m1 * ep.ldc;
        DenseK la2{A + (long)m1 * lda, kbytes(M - m1, lda, K), lda, M - m1, K};
        launch_igemm<DenseK, DenseK, C64, V>(la2, lb, e2, K, st);
      } else launch_big<DenseK, DenseK, V, true>(la, lb, ep, K, st);
    }
  } else if (!transa && !transb) {  // C = A[M,K] * B[K,N]     (input gradient)
    DenseK la{A, kbytes(M, lda, K), lda, M, K};
    DenseM lb{B, kbytes(K, ldb, N), ldb, N, K};
    if (skinny) launch_igemm<DenseK, DenseM, C32x128, V>(la, lb, ep, K, st);
    else launch_big<DenseK, DenseM, V, false>(la, lb, ep, K, st);
  } else {                          // C = A[K,M]^T * B[K,N]   (weight gradient)
    DenseM la{A, kbytes(K, lda, M), lda, M, K};
    DenseM lb{B, kbytes(K, ldb, N), ldb, N, K};
    launch_big<DenseM, DenseM, V, true>(la, lb, ep, K, st);
  }
}

// K-sliced x W^T product: out[z][M][N] = A[:, z*K/ns : (z+1)*K/ns] . B[:, same]^T for z < ns -- the engine's split-K launch with the
// caller's slice count and WITHOUT the reduce pass: the slabs are the result (a batch of ns products whose operands are interleaved
// along K; wino44.hip).  K / ns must be a multiple of the k-tile.
int gemm_kslices(int M, int N, int K, int ns, const float* A, long lda, const float* B, long ldb, float* out, hipStream_t st, int nolog) {
  if (ns < 2 || K % ns || (K / ns) % BKD || !aligned16(A) || !aligned16(B) || lda % 4 || ldb % 4 || !fits32(M, lda, K) || !fits32(N, ldb, K)) {
    re2e_set_error("gemm_kslices: unsupported slicing (M=%d N=%d K=%d ns=%d)", M, N, K, ns);
    return RE2E_EUNSUPPORTED;
  }
  if (gemm_nt2_kslices(M, N, K / ns, ns, A, lda, B, ldb, out, st, nolog)) return RE2E_OK;      // gemm_nt.hip: the LDS-DMA pipeline, tiles counted over the slices
  Epi ep;
  memset(&ep, 0, sizeof(ep));
  ep.C = out; ep.ldc = N; ep.M = M; ep.N = N; ep.act = RE2E_ACT_NONE; ep.ws = out; ep.nsplit = ns; ep.nolog = nolog;
  DenseK la{A, kbytes(M, lda, K), lda, M, K};
  DenseK lb{B, kbytes(N, ldb, K), ldb, N, K};
  launch_big<DenseK, DenseK, true, true>(la, lb, ep, K, st);
  return RE2E_OK;
}

// The same for A^T B (weight-gradient form): out[z][M][N] = A[zK/ns : (z+1)K/ns, :M]^T . B[same rows, :N]; A (K, M), B (K, N) row-major.
int gemm_kslices_tn(int M, int N, int K, int ns, const float* A, long lda, const float* B, long ldb, float* out, hipStream_t st, int nolog) {
  if (ns < 2 || K % ns || (K / ns) % BKD || M % 4 || N % 4 || !aligned16(A) || !aligned16(B) || lda % 4 || ldb % 4 || !fits32(K, lda, M) ||
      !fits32(K, ldb, N)) {
    re2e_set_error("gemm_kslices_tn: unsupported slicing (M=%d N=%d K=%d ns=%d)", M, N, K, ns);
    return RE2E_EUNSUPPORTED;
  }
  Epi ep;
  memset(&ep, 0, sizeof(ep));
  ep.C = out; ep.ldc = N; ep.M = M; ep.N = N; ep.act = RE2E_ACT_NONE; ep.ws = out; ep.nsplit = ns; ep.nolog = nolog;
  DenseM la{A, kbytes(K, lda, M), lda, M, K};
  DenseM lb{B, kbytes(K, ldb, N), ldb, N, K};
  launch_big<DenseM, DenseM, true, true>(la, lb, ep, K, st);
  return RE2E_OK;
}

extern "C" int re2e_gemm(int transa, int transb, int M, int N, int K, const float* A, long lda, const float* B,
                         long ldb, float* C, long ldc, const float* bias, const float* bias2, int act, float beta,
                         const float* mul, float* mask_out, const int* lens_dev, int T, void* workspace,
                         size_t workspace_bytes, hipStream_t stream) {
  RE2E_CHECK_ARG(M > 0 && N > 0 && K > 0, "M,N,K must be positive");
  RE2E_CHECK_ARG(A && B && C, "null operand");
  RE2E_CHECK_ARG(beta == 0.f || beta == 1.f, "beta must be 0 or 1");
  RE2E_CHECK_ARG(act >= 0 && act <= RE2E_ACT_SIGMOID_MASK_MUL, "bad activation");
  if (act == RE2E_ACT_SIGMOID_MASK_MUL) RE2E_CHECK_ARG(mul && mask_out && lens_dev && T > 0, "mask epilogue needs mul/mask_out/lens/T");
  if (transa && transb) {
    re2e_set_error("re2e_gemm: transa && transb is not supported");
    return RE2E_EUNSUPPORTED;
  }
  RE2E_CHECK_ARG(fits32(transa ? K : M, lda, transa ? M : K) && fits32(transb ? N : K, ldb, transb ? K : N), "operand larger than 4 GiB");
  static const bool no_skinny_kernel = exp_env("RE2E_NO_SKINNY_GEMM") != nullptr;
  if (!no_skinny_kernel && !transa && M <= 32 && K >= 64 && K <= 8192 && act == RE2E_ACT_NONE) {
    const bool v = K % 4 == 0 && aligned16(A) && lda % 4 == 0 && (!transb || (aligned16(B) && ldb % 4 == 0));
    const dim3 g(cdiv(N, 32)), t(512);
    if (transb && v) hipLaunchKernelGGL((skinny_gemm_kernel<true, true>), g, t, 0, stream, A, lda, B, ldb, C, ldc, M, N, K, bias, bias2, beta);
    else if (transb) hipLaunchKernelGGL((skinny_gemm_kernel<true, false>), g, t, 0, stream, A, lda, B, ldb, C, ldc, M, N, K, bias, bias2, beta);
    else if (v) hipLaunchKernelGGL((skinny_gemm_kernel<false, true>), g, t, 0, stream, A, lda, B, ldb, C, ldc, M, N, K, bias, bias2, beta);
    else hipLaunchKernelGGL((skinny_gemm_kernel<false, false>), g, t, 0, stream, A, lda, B, ldb, C, ldc, M, N, K, bias, bias2, beta);
    RE2E_LAUNCH_CHECK();
    return RE2E_OK;
  }
  if (!transa && transb && gemm_nt2(M, N, K, A, lda, B, ldb, C, ldc, bias, bias2, act, beta, mul, mask_out, lens_dev, T, workspace, workspace_bytes, stream)) {
    RE2E_LAUNCH_CHECK();
    return RE2E_OK;
  }
  if (transa && !transb && gemm_tn2(M, N, K, A, lda, B, ldb, C, ldc, bias, bias2, act, beta, workspace, workspace_bytes, stream)) {
    RE2E_LAUNCH_CHECK();
    return RE2E_OK;
  }
  Epi ep;
  memset(&ep, 0, sizeof(ep));
  ep.C = C; ep.ldc = ldc; ep.M = M; ep.N = N; ep.bias = bias; ep.bias2 = bias2; ep.act = act; ep.beta = beta;
  ep.mul = mul; ep.mask_out = mask_out; ep.lens = lens_dev; ep.T = T; ep.nsplit = 1;
  const int s = act == RE2E_ACT_SIGMOID_MASK_MUL ? 1 : gemm_splits(transa, transb, M, N, K);   // mask epilogue: never split
  if (s > 1) {
    RE2E_CHECK_ARG(workspace && workspace_bytes >= (size_t)s * M * N * sizeof(float), "workspace too small");
    ep.ws = (float*)workspace; ep.nsplit = s;
  }
  // 16-byte vector loads need aligned bases, leading dimensions that are multiples of 4 and no
  // float4 straddling a bound (k-contiguous operands: K % 4; row-contiguous operands: rows % 4)
  bool va = aligned16(A) && lda % 4 == 0 && (transa ? M % 4 == 0 : K % 4 == 0);
  bool vb = aligned16(B) && ldb % 4 == 0 && (transb ? K % 4 == 0 : N % 4 == 0);
  if (va && vb) gemm_dispatch<true>(transa, transb, M, N, K, A, lda, B, ldb, ep, stream);
  else gemm_dispatch<false>(transa, transb, M, N, K, A, lda, B, ldb, ep, stream);
  if (s > 1) {
    long tot = (long)M * N;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(cdiv(tot, 256)), dim3(256), 0, stream, (const float*)workspace, s,
                       M, N, C, ldc, beta, 0, 0, 0, bias, bias2, act);
  }
  RE2E_LAUNCH_CHECK();
  return RE2E_OK;
}

// ---- weight gradient over the valid rows of a ragged time-major batch ------------------------------------------------------------------
// C[M,N] = sum_{r < Kv} A[map[r]][:M]^T B[map[r]][:N] + beta C: A (dy) and B (x) in the padded (phys_rows x .) layout, the contraction over
// the rows of the map only (the padded rows of dy are zero: the sum is the same, 15 % shorter for a config-4 batch).  16-byte-loadable
// operands only (RE2E_EUNSUPPORTED otherwise: the caller contracts over all rows).  Workspace as re2e_gemm(1, 0, M, N, Kv).
extern "C" int re2e_gemm_tn_rows(int M, int N, int Kv, const float* A, long lda, const float* B, long ldb, float* C, long ldc, float beta,
                                 const int* rowmap, int ident_rows, int phys_rows, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  RE2E_CHECK_ARG(ident_rows >= 0 && ident_rows <= Kv, "ident_rows out of range");
  RE2E_CHECK_ARG(M > 0 && N > 0 && Kv > 0 && phys_rows >= Kv, "bad sizes");
  RE2E_CHECK_ARG(A && B && C && rowmap, "null operand");
  RE2E_CHECK_ARG(beta == 0.f || beta == 1.f, "beta must be 0 or 1");
  RE2E_CHECK_ARG(fits32(phys_rows, lda, M) && fits32(phys_rows, ldb, N), "operand larger than 4 GiB");
  if (!(aligned16(A) && lda % 4 == 0 && M % 4 == 0 && aligned16(B) && ldb % 4 == 0 && N % 4 == 0) || Kv < 64) {
    re2e_set_error("re2e_gemm_tn_rows: operands not 16-byte loadable (M=%d N=%d Kv=%d)", M, N, Kv);
    return RE2E_EUNSUPPORTED;
  }
  Epi ep;
  memset(&ep, 0, sizeof(ep));
  ep.C = C; ep.ldc = ldc; ep.M = M; ep.N = N; ep.act = RE2E_ACT_NONE; ep.beta = beta; ep.nsplit = 1;
  const int s = gemm_splits(1, 0, M, N, Kv);
  if (s > 1) {
    RE2E_CHECK_ARG(workspace && workspace_bytes >= (size_t)s * M * N * sizeof(float), "workspace too small");
    ep.ws = (float*)workspace; ep.nsplit = s;
  }
  DenseMG la{A, kbytes(phys_rows, lda, M), lda, M, Kv, rowmap, ident_rows};
  DenseMG lb{B, kbytes(phys_rows, ldb, N), ldb, N, Kv, rowmap, ident_rows};
  launch_big<DenseMG, DenseMG, true, true>(la, lb, ep, Kv, stream);
  if (s > 1) {
    const long tot = (long)M * N;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(cdiv(tot, 256)), dim3(256), 0, stream, (const float*)workspace, s, M, N, C, ldc, beta, 0, 0, 0,
                       (const float*)nullptr, (const float*)nullptr, (int)RE2E_ACT_NONE);
  }
  RE2E_LAUNCH_CHECK();
  return RE2E_OK;
}

// ---- convolution (NHWC activations, weights pre-gathered by re2e_conv_weight_gather) ----------
// Weight gather: dst[r][a][b][c] laid out for the implicit GEMM B operand from the PyTorch layout
// W[Cout][Cin][KH][KW].  transpose=0: r=co, c=ci (forward) ; transpose=1: r=ci, c=co (data gradient).
// Tap (a,b) of the destination reads source tap (kh0 + a*kstep, kw0 + b*kstep).
__global__ void weight_gather_kernel(const float* W, float* dst, int Cout, int Cin, int KH, int KW, int transpose, int TA,
                                     int TB, int kh0, int kw0, int kstep, int cls_pad) {
  int R = transpose ? Cin : Cout, Cc = transpose ? Cout : Cin;
  long tot = (long)R * TA * TB * Cc;
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= tot) return;
  if (cls_pad >= 0) {   // blockIdx.y = output parity class of the stride-2 data gradient; one set per class
    const int ph = blockIdx.y >> 1, pw = blockIdx.y & 1;
    kh0 = (ph + cls_pad) & 1; kw0 = (pw + cls_pad) & 1;
    dst += (long)blockIdx.y * tot;
  }
  int c = (int)(i % Cc); long t = i / Cc; int b = (int)(t % TB); t /= TB; int a = (int)(t % TA); int r = (int)(t / TA);
  int co = transpose ? c : r, ci = transpose ? r : c;
  int kh = kh0 + a * kstep, kw = kw0 + b * kstep;
  dst[i] = W[(((long)co * Cin + ci) * KH + kh) * KW + kw];
}

// RE2E_NO_THIN=1 routes the Cin == 1 / Cout == 1 convolutions through the implicit GEMM (A/B measurements)
static bool thin_enabled() {
  static const bool v = exp_env("RE2E_NO_THIN") == nullptr;
  return v;
}

template <bool V>
static void conv_dispatch(const ConvGeom& g, int M, int K, const float* wg, int Cout, Epi& ep, hipStream_t st) {
  ConvK la{g, g.in, (unsigned)((long)g.NI * g.H * g.W * g.C * 4), M, K};
  DenseK lb{wg, kbytes(Cout, K, K), (long)K, Cout, K};
  if (ep.N <= 32) launch_igemm<ConvK, DenseK, C256x32, V>(la, lb, ep, K, st);
  else if (ep.N <= 64) launch_n64<ConvK, DenseK, V>(la, lb, ep, K, st);
  else launch_big<ConvK, DenseK, V, true>(la, lb, ep, K, st);
}

// Forward / data-gradient implicit GEMM:
//   out[pix(n,py,px)][co] = act( sum_{kh,kw,ci} in[n][py*SY+kh*DY+OY0][px*SX+kw*DX+OX0][ci] * w[co][kh][kw][ci] + bias[co] )
// written at out[((n*OHF + py*osy+ooy)*OWF + px*osx+oox)*Cout + co].
// out = mask > 0 ? out : 0 (four channels per thread when the channel count allows): the separate pass of re2e_conv_igemm_masked
// for geometries the halo-patch kernel does not cover
__global__ void relu_mask_kernel(float* __restrict__ out, const float* __restrict__ mask, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) out[i] = mask[i] > 0.f ? out[i] : 0.f;
}

static int conv_igemm_impl(const float* in, int NI, int H, int W, int C, const float* wg, int Cout, int KH, int KW,
                           int PH, int PW, int SY, int SX, int DY, int DX, int OY0, int OX0, float* out, int OHF,
                           int OWF, int osy, int osx, int ooy, int oox, const float* bias, int act, float beta, const float* mask,
                           hipStream_t stream) {
  RE2E_CHECK_ARG(in && wg && out, "null operand");
  RE2E_CHECK_ARG(NI > 0 && H > 0 && W > 0 && C > 0 && Cout > 0 && PH > 0 && PW > 0, "bad geometry");
  RE2E_CHECK_ARG(act >= 0 && act <= RE2E_ACT_SIGMOID, "bad activation");
  RE2E_CHECK_ARG((long)NI * PH * PW < 2147483647L, "too many pixels");
  RE2E_CHECK_ARG((long)NI * H * W * C * 4 < 0xFFFFFFF0L, "input tensor larger than 4 GiB");
  ConvGeom g{in, NI, H, W, C, PH, PW, KH, KW, SY, SX, DY, DX, OY0, OX0};
  int M = NI * PH * PW, K = KH * KW * C;
  Epi ep;
  memset(&ep, 0, sizeof(ep));
  ep.C = out; ep.ldc = Cout; ep.M = M; ep.N = Cout; ep.bias = bias; ep.act = act; ep.beta = beta; ep.nsplit = 1;
  ep.remap = 1; ep.PH = PH; ep.PW = PW; ep.OHF = OHF; ep.OWF = OWF; ep.osy = osy; ep.osx = osx; ep.ooy = ooy; ep.oox = oox;
  if (osy == 1 && osx == 1 && ooy == 0 && oox == 0 && OHF == PH && OWF == PW) ep.remap = 0;
  if ((Cout == 1 || C == 1) && thin_enabled() && !mask) {
    OutMap om{out, (long)Cout, ep.remap, PH, PW, OHF, OWF, osy, osx, ooy, oox};
    if (thin_conv_forward(g, wg, Cout, om, bias, act, beta, stream)) {
      RE2E_LAUNCH_CHECK();
      return RE2E_OK;
    }
  }
  if (!ep.remap && halo_conv3x3(g, wg, Cout, out, bias, act, beta, mask, stream)) {
    RE2E_LAUNCH_CHECK();
    return RE2E_OK;
  }
  if (!mask) {
    const int oy0[1] = {OY0}, ox0[1] = {OX0}, oo_y[1] = {ooy}, oo_x[1] = {oox};
    if (conv_nt2(g, M, wg, Cout, out, Cout, bias, act, beta, 1, oy0, ox0, 0, ep.remap, OHF, OWF, osy, osx, oo_y, oo_x, stream)) {
      RE2E_LAUNCH_CHECK();
      return RE2E_OK;
    }
  }
  if (C % 4 == 0 && aligned16(in) && aligned16(wg)) conv_dispatch<true>(g, M, K, wg, Cout, ep, stream);
  else conv_dispatch<false>(g, M, K, wg, Cout, ep, stream);
  if (mask) {
    const long n = (long)NI * OHF * OWF * Cout;
    const long nb = (n + 255) / 256;
    hipLaunchKernelGGL(relu_mask_kernel, dim3((unsigned)(nb < 65535 * 8 ? nb : 65535 * 8)), dim3(256), 0, stream, out, mask, n);
  }
  RE2E_LAUNCH_CHECK();
  return RE2E_OK;
}

extern "C" int re2e_conv_igemm(const float* in, int NI, int H, int W, int C, const float* wg, int Cout, int KH, int KW,
                               int PH, int PW, int SY, int SX, int DY, int DX, int OY0, int OX0, float* out, int OHF,
                               int OWF, int osy, int osx, int ooy, int oox, const float* bias, int act, float beta,
                               hipStream_t stream) {
  return conv_igemm_impl(in, NI, H, W, C, wg, Cout, KH, KW, PH, PW, SY, SX, DY, DX, OY0, OX0, out, OHF, OWF, osy, osx, ooy, oox, bias, act,
                         beta, nullptr, stream);
}

extern "C" int re2e_conv3x3_relu_pool(const float* in, int NI, int H, int W, int C, const float* wg, int Cout, const float* bias,
                                      float* pooled, unsigned char* idx_u8, hipStream_t stream) {
  RE2E_CHECK_ARG(in && wg && pooled && idx_u8, "null operand");
  RE2E_CHECK_ARG(NI > 0 && H > 0 && W > 0 && C > 0 && Cout > 0, "bad geometry");
  ConvGeom g{in, NI, H, W, C, H, W, 3, 3, 1, 1, 1, 1, -1, -1};
  if (!halo_conv3x3(g, wg, Cout, nullptr, bias, RE2E_ACT_RELU, 0.f, nullptr, stream, pooled, idx_u8)) {
    re2e_set_error("re2e_conv3x3_relu_pool: needs C %% 16 == 0, Cout %% 64 == 0, 16-byte aligned tensors < 2 GiB");
    return RE2E_EUNSUPPORTED;
  }
  RE2E_LAUNCH_CHECK();
  return RE2E_OK;
}

extern "C" int re2e_conv_igemm_masked(const float* in, int NI, int H, int W, int C, const float* wg, int Cout, int KH, int KW,
                                      int PH, int PW, int SY, int SX, int DY, int DX, int OY0, int OX0, float* out, int OHF,
                                      int OWF, int osy, int osx, int ooy, int oox, const float* relu_out, hipStream_t stream) {
  RE2E_CHECK_ARG(relu_out, "null mask");
  RE2E_CHECK_ARG(osy == 1 && osx == 1 && ooy == 0 && oox == 0 && OHF == PH && OWF == PW, "the mask variant covers dense outputs only");
  return conv_igemm_impl(in, NI, H, W, C, wg, Cout, KH, KW, PH, PW, SY, SX, DY, DX, OY0, OX0, out, OHF, OWF, osy, osx, ooy, oox, nullptr,
                         RE2E_ACT_NONE, 0.f, relu_out, stream);
}

// Stride-2 data gradient (transposed convolution) in ONE launch: blockIdx.z walks the four output parity
// classes (ph,pw); class taps a -> kh = 2a + ((ph+pad) & 1).  wt_ws holds the four gathered weight sets
// [cls][Cin][KH/2][KW/2][Cout] (= Cin*KH*KW*Cout floats), written here.
extern "C" int re2e_conv_dgrad_s2(const float* dz, int N, int OH, int OW, int Cout, const float* W, int Cin, int KH, int KW,
                                  int H, int Wd, int pad, float* dx, float* wt_ws, hipStream_t stream) {
  RE2E_CHECK_ARG(dz && W && dx && wt_ws, "null operand");
  RE2E_CHECK_ARG(N > 0 && OH > 0 && OW > 0 && Cout > 0 && Cin > 0 && H > 0 && Wd > 0 && pad >= 0, "bad geometry");
  if (KH % 2 || KW % 2) { re2e_set_error("re2e_conv_dgrad_s2: kernel sizes must be even"); return RE2E_EUNSUPPORTED; }
  const int TA = KH / 2, TB = KW / 2, PH = (H + 1) / 2, PW = (Wd + 1) / 2;
  RE2E_CHECK_ARG((long)N * PH * PW < 2147483647L, "too many pixels");
  RE2E_CHECK_ARG((long)N * OH * OW * Cout * 4 < 0xFFFFFFF0L, "gradient tensor larger than 4 GiB");
  const long wtot = (long)Cin * TA * TB * Cout;
  hipLaunchKernelGGL(weight_gather_kernel, dim3(cdiv(wtot, 256), 4), dim3(256), 0, stream, W, wt_ws, Cout, Cin, KH, KW, 1, TA, TB,
                     0, 0, 2, pad);
  ConvGeom g{dz, N, OH, OW, Cout, PH, PW, TA, TB, 1, 1, -1, -1, 0, 0};
  const int M = N * PH * PW, K = TA * TB * Cout;
  Epi ep;
  memset(&ep, 0, sizeof(ep));
  ep.C = dx; ep.ldc = Cin; ep.M = M; ep.N = Cin; ep.act = RE2E_ACT_NONE; ep.nsplit = 1;
  ep.remap = 1; ep.PH = PH; ep.PW = PW; ep.OHF = H; ep.OWF = Wd; ep.osy = 2; ep.osx = 2;
  ep.ncls = 4; ep.cls_wstride = wtot;
  for (int p = 0; p < 2; ++p) {   // oh = (2i + p + pad - kh)/2 = i + (p + pad - kh0)/2 - a
    const int k0 = (p + pad) & 1;
    ep.cls_oy0[p] = (p + pad - k0) / 2; ep.cls_ox0[p] = ep.cls_oy0[p];
  }
  {
    // class (ph, pw) = cls >> 1, cls & 1: input offsets cls_oy0[ph] / cls_ox0[pw], output positions (2i + ph, 2j + pw)
    int oy0[4], ox0[4], oo_y[4], oo_x[4];
    for (int c = 0; c < 4; ++c) { oy0[c] = ep.cls_oy0[c >> 1]; ox0[c] = ep.cls_ox0[c & 1]; oo_y[c] = c >> 1; oo_x[c] = c & 1; }
    if (conv_nt2(g, M, wt_ws, Cin, dx, Cin, nullptr, RE2E_ACT_NONE, 0.f, 4, oy0, ox0, wtot, 1, H, Wd, 2, 2, oo_y, oo_x, stream)) {
      RE2E_LAUNCH_CHECK();
      return RE2E_OK;
    }
  }
  if (Cout % 4 == 0 && aligned16(dz) && aligned16(wt_ws)) conv_dispatch<true>(g, M, K, wt_ws, Cin, ep, stream);
  else conv_dispatch<false>(g, M, K, wt_ws, Cin, ep, stream);
  RE2E_LAUNCH_CHECK();
  return RE2E_OK;
}

static int wgrad_splits_mfma(int Mrows, int Cout, long P) {
  int bm, bn;
  if (Cout <= 32) { bm = 256; bn = 32; } else if (Cout <= 64) { bm = (Mrows % 192 == 0 && Mrows % 256 != 0) ? 192 : 256; bn = 64; } else if (Mrows % 192 == 0 && Mrows % 128 != 0) { bm = 192; bn = 64; } else { bm = 128; bn = 128; }
  return pick_splits(Mrows, Cout, (int)P, bm, bn);
}

static int wgrad_splits(int Mrows, int Cout, long P, long rows, int C, int KH, int KW, bool* thin = nullptr) {
  if (thin) *thin = false;
  if ((C == 1 || Cout == 1) && thin_enabled()) {
    int s = thin_wgrad_slabs(C, Cout, KH, KW, P, rows);
    if (s > 0) { if (thin) *thin = true; return s; }
  }
  return wgrad_splits_mfma(Mrows, Cout, P);
}

template <bool V>
static void wgrad_dispatch(const ConvGeom& g, int Mrows, int P, const float* dout, int Cout, Epi& ep, hipStream_t st) {
  ConvM la{g, g.in, (unsigned)((long)g.NI * g.H * g.W * g.C * 4), Mrows, P};
  DenseM lb{dout, kbytes(P, Cout, Cout), (long)Cout, Cout, P};
  if (Cout <= 32) launch_igemm<ConvM, DenseM, C256x32, V>(la, lb, ep, P, st);
  else if (Cout <= 64) {
    if constexpr (V) {
      if (Mrows % 192 == 0 && Mrows % 256 != 0) { launch_igemm<ConvM, DenseM, C192x64, V>(la, lb, ep, P, st); return; }
    }
    launch_n64<ConvM, DenseM, V>(la, lb, ep, P, st);
  }
  else {
    if constexpr (V) {
      if (Mrows % 192 == 0 && Mrows % 128 != 0) { launch_igemm<ConvM, DenseM, C192x64, V>(la, lb, ep, P, st); return; }   // conv2_1: 576 x 128
    }
    launch_big<ConvM, DenseM, V, false>(la, lb, ep, P, st);
  }
}

extern "C" size_t re2e_conv_wgrad_workspace_bytes(int NI, int PH, int PW, int C, int Cout, int KH, int KW) {
  int Mrows = KH * KW * C;
  long P = (long)NI * PH * PW;
  int s = wgrad_splits(Mrows, Cout, P, (long)NI * PH, C, KH, KW);
  return (size_t)s * Mrows * Cout * sizeof(float);   // always reduce through the workspace (layout permute)
}

// Weight gradient: dW[co][ci][kh][kw] (+)= sum_pix dout[pix][co] * in[n][py*SY+kh+OY0][px*SX+kw+OX0][ci]
extern "C" int re2e_conv_wgrad(const float* in, int NI, int H, int W, int C, const float* dout, int Cout, int KH, int KW,
                               int PH, int PW, int SY, int SX, int OY0, int OX0, float* dW, float beta,
                               void* workspace, size_t workspace_bytes, hipStream_t stream) {
  RE2E_CHECK_ARG(in && dout && dW && workspace, "null operand");
  ConvGeom g{in, NI, H, W, C, PH, PW, KH, KW, SY, SX, 1, 1, OY0, OX0};
  int Mrows = KH * KW * C;
  long P = (long)NI * PH * PW;
  RE2E_CHECK_ARG(P < 2147483647L, "too many pixels");
  RE2E_CHECK_ARG((long)NI * H * W * C * 4 < 0xFFFFFFF0L && P * Cout * 4 < 0xFFFFFFF0L, "tensor larger than 4 GiB");
  bool thin = false;
  int s = wgrad_splits(Mrows, Cout, P, (long)NI * PH, C, KH, KW, &thin);
  if (thin && !(aligned16(in) && aligned16(dout))) { thin = false; s = wgrad_splits_mfma(Mrows, Cout, P); }
  RE2E_CHECK_ARG(workspace_bytes >= (size_t)s * Mrows * Cout * sizeof(float), "workspace too small");
  Epi ep;
  memset(&ep, 0, sizeof(ep));
  ep.M = Mrows; ep.N = Cout; ep.nsplit = s; ep.ws = (float*)workspace;
  if (s == 1) {   // still go through the slab so that the reduce kernel applies the layout permute
    ep.C = (float*)workspace; ep.ldc = Cout;
  }
  if (thin) thin_wgrad(g, dout, Cout, (float*)workspace, s, stream);
  else if (C % 4 == 0 && Cout % 4 == 0 && aligned16(in) && aligned16(dout)) wgrad_dispatch<true>(g, Mrows, (int)P, dout, Cout, ep, stream);
  else wgrad_dispatch<false>(g, Mrows, (int)P, dout, Cout, ep, stream);
  long tot = (long)Mrows * Cout;
  if (s >= 64 && tot <= 65536)
    hipLaunchKernelGGL(splitk_reduce_wide_kernel, dim3(cdiv(tot, 64)), dim3(1024), 0, stream, (const float*)workspace, s, Mrows,
                       Cout, dW, (long)Cout, beta, 1, C, KH * KW);
  else
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(cdiv(tot, 256)), dim3(256), 0, stream, (const float*)workspace, s, Mrows,
                       Cout, dW, (long)Cout, beta, 1, C, KH * KW, (const float*)nullptr, (const float*)nullptr, RE2E_ACT_NONE);
  RE2E_LAUNCH_CHECK();
  return RE2E_OK;
}

extern "C" int re2e_conv_weight_gather(const float* W, float* dst, int Cout, int Cin, int KH, int KW, int transpose,
                                       int TA, int TB, int kh0, int kw0, int kstep, hipStream_t stream) {
  RE2E_CHECK_ARG(W && dst, "null operand");
  RE2E_CHECK_ARG(kh0 + (TA - 1) * kstep < KH && kw0 + (TB - 1) * kstep < KW && kh0 >= 0 && kw0 >= 0, "tap out of range");
  long tot = (long)Cout * Cin * TA * TB;
  hipLaunchKernelGGL(weight_gather_kernel, dim3(cdiv(tot, 256)), dim3(256), 0, stream, W, dst, Cout, Cin, KH, KW, transpose,
                     TA, TB, kh0, kw0, kstep, -1);
  RE2E_LAUNCH_CHECK();
  return RE2E_OK;
}

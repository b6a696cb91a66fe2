// K3 (round 5): the dense x W^T product (both operands k-major) as an LDS-DMA pipelined, work-centric kernel for gfx950.
//
// What it changes against igemm.hip's mainloop for the same product (kept there for every other operand form):
//   * operand tiles travel global -> LDS by `buffer_load_dwordx4 ... lds` (no staging registers, no ds_write pass, no
//     per-k-step address arithmetic: per-lane offsets are kernel constants, the k position is the instruction's scalar
//     offset), ST stages deep with COUNTED vmcnt waits and raw s_barrier, so a tile has ST-1 whole k-steps to land;
//     the LDS image is unpadded [row][BK] with the 16-byte chunks XOR-swizzled on the SOURCE side (an LDS-DMA piece lands
//     lane-linearly), which keeps every ds_read_b128 fragment read conflict-free;
//   * the workgroup barrier of a k-step sits in FRONT of its last MFMA group: the first fragments of the next k-step are
//     fetched under that group, so the two waves a SIMD holds do not both wait for LDS right behind a barrier;
//   * schedule = whole tiles for the full rounds + a stream-K tail: the tiles of the last (partial) round(s) are cut
//     along K into equal unit ranges, one per CU; a tile that was cut is finished by the workgroup that arrives LAST at
//     the tile's ticket counter, which sums the parts' slabs in k order (bitwise reproducible whoever arrives last).
//     12800 x 512 outputs = 200 tiles of 256x128 on 256 CUs ran at 78 % fill; here every CU gets 200/256 of a tile.
//   * hand-off of the partial slabs: 16-byte write-through (sc1) stores, every wave drains (vmcnt(0)), workgroup barrier,
//     one lane's agent-scope ticket add; the finisher reads every slab with sc1 loads (MI355X_MICROARCH.md: hand-off R1 in
//     its counter form; no fence, no wait, no co-residency assumption: a workgroup never waits for another).
//
// Replaces (reference, stock ATen): torch.nn.Linear forward and input gradient in model/e2e_encoder.py:145-147,173-176,
// model/enhance_model.py:108-114, model/e2e_ctc.py:51, model/e2e_decoder.py:150 (via re2e_gemm (0,1)).
#include <stdlib.h>

#include "common.h"

namespace {

typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
constexpr unsigned OOB = 0x80000000u;    // a voffset no descriptor admits: the load returns / stages zeros, no memory traffic

template <int WM_, int WN_, int TM_, int TN_, int BK_, int ST_, int WPS_>
struct NtCfg {
  static constexpr int WM = WM_, WN = WN_, TM = TM_, TN = TN_, BK = BK_, ST = ST_, WPS = WPS_;
  static constexpr int BM = WM_ * TM_ * 32, BN = WN_ * TN_ * 32;
  static constexpr int NW = WM_ * WN_, THREADS = NW * 64;
  static constexpr int RPP = 256 / BK_;            // rows of one 1-KiB LDS-DMA piece
  static constexpr int CPR = BK_ / 4;              // 16-byte chunks per row
  static constexpr int RPL = 64 / BK_;             // rows per 256-byte LDS bank line
  static constexpr int PA = BM / RPP, PB = BN / RPP;
  static constexpr int NPW = (PA + PB) / NW;       // pieces per wave and stage
  static constexpr int NPA = PA / NW;              // of which A pieces
  static constexpr int STAGE = (BM + BN) * BK_;    // floats
  static constexpr int NQ = BK_ / 8;               // k-groups of 8 (one ds_read_b128 per fragment row feeds 4 MFMAs)
  static constexpr size_t LDS_BYTES = (size_t)ST_ * STAGE * 4;
  static_assert(PA % NW == 0 && PB % NW == 0, "pieces must divide over the waves per operand");
  static_assert(BK_ == 16 || BK_ == 32 || BK_ == 64, "BK");
  static_assert((ST_ - 1) * NPW < 64, "vmcnt is a 6-bit counter");
};

struct NtArgs {
  const float* A; const float* B; float* C;
  unsigned a_bytes, b_bytes;
  int lda, ldb; long ldc;
  int M, N, K;
  const float* bias; const float* bias2; int act; float beta;
  int ntm, ntn;      // tile grid
  int g_sk;          // workgroups [0, g_sk) share the k-tile units of tiles [n_dp, ntm*ntn) equally (stream-K tail, dispatched first)
  int n_dp;          // workgroups [g_sk, g_sk + n_dp) compute one whole tile each (tiles [0, n_dp) of the XCD-aware order)
  int nkt;           // k-tiles per tile
  float* slabs;      // [g_sk][2][BM*BN] partial accumulators of cut tiles
  int* counters;     // [ntm*ntn - n_dp] arrival tickets: zero on entry, put back to zero by each tile's last arriver
  // MODE 0, row map (re2e_gemm_nt_rows): logical row r of the product is physical row rowmap[r] of BOTH A and C (M counts logical rows; the
  // bounds a_bytes / ldc cover the physical tensors).  Time-major (T, B, .) activations of ragged batches: only the (t, b) with t < len_b.
  const int* rowmap;
  int ident_rows;    // rowmap[r] == r for r < ident_rows (every utterance is at least that long): tiles below need no look-up
  int nomem;
  // MODE 2 (implicit-GEMM convolution; csrc/common.h ConvGeom semantics).  A = the NHWC image, M = pixels of ONE class, K = KH*KW*C.
  int cH, cW, cC, cPH, cPW, cKH, cKW, cSY, cSX, cDY, cDX;
  int ncls;                       // 1, or 4 output parity classes: tile index / (ntm*ntn) = class
  int cOY0[4], cOX0[4];           // per class: input row / column of tap (0,0) at pixel (0,0)
  long cls_wstride;               // per class: float offset of its weight set in B
  int remap, OHF, OWF, osy, osx, ooy[4], oox[4];      // output row of pixel (n,py,px): ((n*OHF + py*osy+ooy)*OWF + px*osx+oox)*ldc ; remap = 0: m*ldc
  unsigned a_shift;               // bytes the image descriptor starts in FRONT of the tensor (so that every per-lane offset is >= 0)
  // MODE 0, K-sliced batch (gemm_kslices: Winograd F(2x2,4x4) positions): ncls slices, slice z contracts k in [z*batch_k, (z+1)*batch_k) and writes
  // C + z*batch_c; K = batch_k, tiles count over the slices like the convolution's classes
  int batch_k; long batch_c;
};

template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void wg_barrier() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
__device__ __forceinline__ void store16_sc1(float* p, const f32x4& v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}

// contiguous chunk of a sequence of n items per XCD (workgroups are dealt round-robin over the 8 XCDs): bijective
__device__ __forceinline__ int xcd_chunk(int orig, int n) {
  const int q8 = n >> 3, r8 = n & 7, xcd = orig & 7;
  return (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
}

// TNF = false: C = A[M,K] . B[N,K]^T, both operands k-major (x W^T).
// TNF = true:  C = A[K,M]^T . B[K,N], both operands ROW-major along the contraction (dy^T x, the weight gradient): a k-row of a tile is
//   contiguous, so the LDS image [BK][BM] | [BK][BN] is the memory image (no swizzle, no transposing stage) and the matrix-core operands come
//   straight out of it: the wave's TM x TN tiles are INTERLEAVED in both directions -- lane lr of the A operand stands for rows m = TM*lr + a,
//   lane lr of the B operand for columns n = TN*lr + b -- so ONE ds_read of TM (TN) consecutive floats of k-row 2s + lh is the lane's operand for
//   all its tiles (first form: a ds_read_b32 per B tile -- 12 LDS instructions per 16 MFMAs instead of 4 -- lost 15 % to instruction issue).
//   The MFMAs run un-swapped: a lane holds, per tile row, the TN consecutive columns of its lr: the store is TN floats wide.  Conflict-free reads.
// MODE 2: MODE 0 with the A operand gathered from an NHWC image (implicit GEMM: rows = output pixels, k = (tap, channel), a k-tile = BK channels
//   of ONE tap = 64 contiguous bytes of a pixel): the per-lane offset is the pixel (a tile constant), the tap + channel position goes through the
//   scalar offset, the halo through a per-lane bit mask over the taps (invalid -> out-of-range offset: zeros, no traffic).  Tiles count over the
//   output parity classes of a stride-2 data gradient (re2e_conv_dgrad_s2): class = own input offsets, weight set and output positions.
template <class CF, int MODE>
__global__ __launch_bounds__(CF::THREADS, CF::WPS) void gemm_nt2_kernel(NtArgs p) {
  constexpr bool TNF = MODE == 1, CONV = MODE == 2;
  constexpr int BM = CF::BM, BN = CF::BN, BK = CF::BK, ST = CF::ST, TM = CF::TM, TN = CF::TN, NW = CF::NW, TH = CF::THREADS;
  constexpr int NPW = CF::NPW, NPA = CF::NPA, NQ = CF::NQ, STAGE = CF::STAGE, CPR = CF::CPR, RPL = CF::RPL, RPP = CF::RPP;
  extern __shared__ __attribute__((aligned(1024))) float smem[];     // the ONE LDS object of the kernel (a second one makes hipcc drain vmcnt before every ds_read)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid / CF::WN, wn = wid % CF::WN;
  const int lr = lane & 31, lh = lane >> 5;

  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.A), 0, p.nomem ? 0 : p.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.B), 0, p.nomem ? 0 : p.b_bytes, 0x00020000);

  // ---- work range ---------------------------------------------------------------------------------------------------
  // DP workgroup: one tile, all its k-tiles.  SK workgroup j of g_sk: units [j*U/g, (j+1)*U/g) of the sequence (tile, kt) over
  // the tiles n_dp.., tiles in the same XCD-aware order, so consecutive unit ranges (same XCD) walk neighbouring tiles.
  const int T = p.ntm * p.ntn;
  const int nkt = p.nkt;
  const bool is_sk = (int)blockIdx.x < p.g_sk;      // the cut tiles FIRST: their hand-offs and slab sums then run beside the whole tiles of the other resident workgroup
  long u0, u1;           // unit range in the SK sequence (or the one tile of a DP workgroup, expressed in the same terms)
  int tile_base;         // first tile of the sequence the units count from
  int my_sk = 0;
  const long U = (long)(T - p.n_dp) * nkt;
  if (!is_sk) {
    tile_base = xcd_chunk((int)blockIdx.x - p.g_sk, p.n_dp);      // (g_sk is a multiple of 8: the XCD of a workgroup is blockIdx % 8 either way)
    u0 = 0; u1 = nkt;
  } else {
    my_sk = xcd_chunk((int)blockIdx.x, p.g_sk);
    tile_base = p.n_dp;
    u0 = (long)my_sk * U / p.g_sk; u1 = (long)(my_sk + 1) * U / p.g_sk;
  }
  const long range0 = u0;

  // ---- per-lane constants of the staging pieces and of the fragment reads ---------------------------------------------
  // x W^T:  piece = RPP rows x BK floats; prow_t = the lane's row in the operand tile, pc16 = byte offset of its (swizzled) 16-byte chunk in the row
  // dy^T x: piece = 256 consecutive floats of the [BK][BM] / [BK][BN] image; prow_t = the lane's k-row in the tile, pc16 = byte offset of its 4 columns
  const int prow = lane / CPR, cpos = lane % CPR;
  int prow_t[NPW];
  unsigned pc16[NPW];
#pragma unroll
  for (int i = 0; i < NPW; ++i) {
    const int pi = (i < NPA ? i : i - NPA) * NW + wid;       // piece index inside the operand
    if constexpr (!TNF) {
      prow_t[i] = pi * RPP + prow;
      pc16[i] = (unsigned)((cpos ^ ((prow_t[i] / RPL) % CPR)) * 16);
    } else {
      const int width = i < NPA ? BM : BN, f = pi * 256 + lane * 4;
      prow_t[i] = f / width;
      pc16[i] = (unsigned)((f % width) * 4);
    }
  }
  const int xs = lh ^ ((lr / RPL) % CPR);
  const int a_row0 = TNF ? wm * TM * 32 + TM * lr : (wm * TM) * 32 + lr, b_row0 = TNF ? wn * TN * 32 + TN * lr : (wn * TN) * 32 + lr;
  const bool ragged_k = (p.K % BK) != 0;
  const int krem_bytes = (p.K - (nkt - 1) * BK) * 4;          // valid bytes of a row in the last k-tile (x W^T); / 4 = its valid k-rows (dy^T x)

  while (u0 < u1) {
    const int trel = (int)(u0 / nkt);
    const int kb = (int)(u0 - (long)trel * nkt);
    const int ke = (int)min((long)nkt, kb + (u1 - u0));
    int tile = tile_base + trel;
    int cls = 0;
    if (CONV || p.ncls > 1) { const int tpc = p.ntm * p.ntn; cls = tile / tpc; tile -= cls * tpc; }
    int tile_m, tile_n;
    {
      const int GROUP = 8, per_group = GROUP * p.ntn;
      const int gid = tile / per_group, first_m = gid * GROUP;
      const int gsz = min(p.ntm - first_m, GROUP);
      tile_m = first_m + (tile % per_group) % gsz;
      tile_n = (tile % per_group) / gsz;
    }
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    unsigned voff[NPW], voff_t[NPW];
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
      const bool isA = i < NPA;
      if constexpr (CONV) {
        if (isA) {
          // pixel of this lane's row, the position of its tap (0,0) moved to the most negative tap so that scalar tap offsets are >= 0
          const int row = m0 + prow_t[i];
          const int px = row % p.cPW, t2 = row / p.cPW, py = t2 % p.cPH, n = t2 / p.cPH;
          const int iy0 = py * p.cSY + p.cOY0[cls], ix0 = px * p.cSX + p.cOX0[cls];
          const int ylo = p.cDY < 0 ? (p.cKH - 1) * p.cDY : 0, xlo = p.cDX < 0 ? (p.cKW - 1) * p.cDX : 0;
          unsigned mask = 0;
          if (row < p.M) {
#pragma unroll 1
            for (int t = 0; t < p.cKH * p.cKW; ++t) {
              const int iy = iy0 + (t / p.cKW) * p.cDY, ix = ix0 + (t % p.cKW) * p.cDX;
              if ((unsigned)iy < (unsigned)p.cH && (unsigned)ix < (unsigned)p.cW) mask |= 1u << t;
            }
          }
          voff[i] = (unsigned)((((long)n * p.cH + iy0 + ylo) * p.cW + ix0 + xlo) * p.cC * 4 + (long)p.a_shift) + pc16[i];
          voff_t[i] = mask;       // (MODE 2 has no ragged k: C % BK == 0)
          continue;
        }
      }
      if constexpr (!TNF) {
        int row = (isA ? m0 : n0) + prow_t[i];
        const bool ok = row < (isA ? p.M : p.N);
        if constexpr (MODE == 0) { if (isA && ok && p.rowmap && m0 + BM > p.ident_rows) row = p.rowmap[row]; }
        voff[i] = ok ? (unsigned)row * (unsigned)(isA ? p.lda : p.ldb) * 4u + pc16[i] : OOB;
        voff_t[i] = (ok && (int)pc16[i] < krem_bytes) ? voff[i] : OOB;
      } else {
        const int col = (isA ? m0 : n0) * 4 + (int)pc16[i];           // byte offset of the lane's 4 columns in a k-row (M, N multiples of 4)
        const bool ok = col < (isA ? p.M : p.N) * 4;
        voff[i] = ok ? (unsigned)prow_t[i] * (unsigned)(isA ? p.lda : p.ldb) * 4u + (unsigned)col : OOB;
        voff_t[i] = (ok && prow_t[i] * 4 < krem_bytes) ? voff[i] : OOB;
      }
    }
    auto issue = [&](int slot, int kt) {
      const bool dead = kt >= ke;
      const bool tail = ragged_k && kt == nkt - 1;
      const unsigned kb4 = dead ? 0u : (unsigned)kt * (unsigned)(BK * 4);
      unsigned soffA = TNF ? kb4 * (unsigned)p.lda : kb4, soffB = TNF ? kb4 * (unsigned)p.ldb : kb4;
      if constexpr (MODE == 0) { const unsigned zk = (unsigned)(cls * p.batch_k) * 4u; soffA += zk; soffB += zk; }
      unsigned tapbit = 0;
      if constexpr (CONV) {
        const int cpt = p.cC / BK, tap = dead ? 0 : kt / cpt, c0 = dead ? 0 : (kt - tap * cpt) * BK;       // scalar
        const int a = tap / p.cKW, b = tap - a * p.cKW;
        const int ya = a * p.cDY - (p.cDY < 0 ? (p.cKH - 1) * p.cDY : 0), xb = b * p.cDX - (p.cDX < 0 ? (p.cKW - 1) * p.cDX : 0);      // >= 0
        soffA = (unsigned)(((ya * p.cW + xb) * p.cC + c0) * 4);
        soffB += (unsigned)(cls * p.cls_wstride * 4);
        tapbit = 1u << tap;
      }
      float* base = smem + slot * STAGE + wid * 256;
#pragma unroll
      for (int i = 0; i < NPW; ++i) {
        unsigned v = dead ? OOB : (tail ? voff_t[i] : voff[i]);
        if constexpr (CONV) { if (i < NPA) v = (!dead && (voff_t[i] & tapbit)) ? voff[i] : OOB; }
        __builtin_amdgcn_raw_ptr_buffer_load_lds(i < NPA ? rsA : rsB, (__attribute__((address_space(3))) void*)(base + i * NW * 256), 16, v,
                                                 i < NPA ? soffA : soffB, 0, 0);
      }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int b = 0; b < TN; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    f32x4 fa[2][TM], fb[2][TN];
    auto read_frags = [&](int slot, int q, int buf) {
      const float* S = smem + slot * STAGE;
      if constexpr (!TNF) {
        const int co = ((2 * q) ^ xs) << 2;
#pragma unroll
        for (int a = 0; a < TM; ++a) fa[buf][a] = *reinterpret_cast<const f32x4*>(S + (a_row0 + a * 32) * BK + co);
#pragma unroll
        for (int b = 0; b < TN; ++b) fb[buf][b] = *reinterpret_cast<const f32x4*>(S + BM * BK + (b_row0 + b * 32) * BK + co);
      } else {
        typedef float fTM __attribute__((ext_vector_type(TM)));
        typedef float fTN __attribute__((ext_vector_type(TN)));
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int k = 8 * q + 2 * j + lh;
          const fTM av = *reinterpret_cast<const fTM*>(S + k * BM + a_row0);
          const fTN bv = *reinterpret_cast<const fTN*>(S + BK * BM + k * BN + b_row0);
#pragma unroll
          for (int a = 0; a < TM; ++a) fa[buf][a][j] = av[a];
#pragma unroll
          for (int b = 0; b < TN; ++b) fb[buf][b][j] = bv[b];
        }
      }
    };
    auto mfma_group = [&](int buf) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
          for (int b = 0; b < TN; ++b)
            acc[a][b] = TNF ? __builtin_amdgcn_mfma_f32_32x32x2f32(fa[buf][a][j], fb[buf][b][j], acc[a][b], 0, 0, 0)
                            : __builtin_amdgcn_mfma_f32_32x32x2f32(fb[buf][b][j], fa[buf][a][j], acc[a][b], 0, 0, 0);
        // (Measured and rejected, round 5: a v_nop behind every 1st / 2nd / 4th group.  A wavefront that issues MFMA after MFMA keeps its SIMD's issue grant
        // and a partner -- a recurrent chain's wavefront, say -- issues nothing until the stream pauses (tools/micro/mfma_partner.hip); a vector-ALU instruction
        // in the stream releases the grant.  Step 50.25 -> 50.4-50.9 / 50.5 / 50.2 ms (profiles/r05_ab_nt2_yield.txt): the barrier per k-tile is pause enough.)
      }
    };

    // ---- pipeline ----
#pragma unroll
    for (int s = 0; s < ST; ++s) issue(s, kb + s);
    wait_vmcnt<(ST - 1) * NPW>();
    wg_barrier();
    read_frags(0, 0, 0);
    int slot = 0;
    for (int kt = kb; kt < ke; ++kt) {
      const int nslot = slot + 1 == ST ? 0 : slot + 1;
#pragma unroll
      for (int q = 0; q < NQ - 1; ++q) {
        read_frags(slot, q + 1, (q + 1) & 1);
        mfma_group(q & 1);
      }
      wait_lgkm0();                          // this wave's reads of the stage are complete: its slot may be refilled behind the barrier
      wait_vmcnt<(ST - 2) * NPW>();          // this wave's pieces of the next stage have landed
      wg_barrier();                          // ... and everybody's
      issue(slot, kt + ST);
      read_frags(nslot, 0, NQ & 1);
      mfma_group((NQ - 1) & 1);
      slot = nslot;
    }
    wait_vmcnt<0>();                         // trailing (dead) pieces: nothing may still be landing when the LDS is reused
    wg_barrier();

    // ---- cut tile: publish the partial, take a ticket; the last arriver sums all parts in k order ----
    const bool whole = kb == 0 && ke == nkt;
    bool finish = true;
    if (!whole) {
      float* slab = p.slabs + ((long)my_sk * 2 + (u0 == range0 ? 0 : 1)) * (BM * BN);
      // the stores below are inline asm (write-through form): hipcc pads no hazard for them, and an MFMA result needs its passes before a
      // vector-memory instruction may read it -- the barrier above usually covers that, these wait states always do
      asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" ::: "memory");
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            f32x4 v = {acc[a][b][4 * g], acc[a][b][4 * g + 1], acc[a][b][4 * g + 2], acc[a][b][4 * g + 3]};
            store16_sc1(slab + (((a * TN + b) * 4 + g) * TH + tid) * 4, v);
          }
      wait_vmcnt<0>();
      wg_barrier();
      // parts of this tile: the workgroups whose unit ranges meet [trel*nkt, (trel+1)*nkt)
      const long t0 = (long)trel * nkt, t1 = t0 + nkt;
      const int g_first = (int)(((t0 + 1) * p.g_sk - 1) / U), g_last = (int)((t1 * p.g_sk - 1) / U);
      const int nparts = g_last - g_first + 1;
      // Ordering of the hand-off (why the ticket can be a RELAXED agent-scope add, with no buffer_wbl2 / buffer_inv around it -- a write-back of
      // the XCD's whole L2 per cut tile would cost more than the tail saves):
      //   release side: every byte a finisher will read was written by an sc1 (write-through) store; the s_waitcnt vmcnt(0) above returns only
      //     when those stores are acknowledged by the memory side, every wave of the workgroup has passed that wait before the barrier
      //     releases lane 0, and only then is the add issued -- it cannot be observed before the data it announces;
      //   acquire side: the finisher's slab reads are issued after the add has RETURNED (its value decides the branch they sit in) and are sc1
      //     loads, served by the memory side, never by this XCD's L2 or a CU's vector cache: there is no stale copy they could hit;
      //   the add itself executes at the memory side (agent scope), so the tickets of the parts of a tile are totally ordered.
      // tests/test_kernels_gpu.py::test_gemm_nt_stream_k_tail_under_load and tools/stress_gemm_nt_tail.py (many parts per tile, a busy second
      // stream, mapped rows) check sums and bitwise repeatability.
      if (tid == 0) {
        const int old = __hip_atomic_fetch_add(p.counters + trel, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        reinterpret_cast<volatile int*>(smem)[0] = old;
      }
      wg_barrier();
      const int old = reinterpret_cast<volatile int*>(smem)[0];
      wg_barrier();
      finish = old == nparts - 1;
      if (finish) {
        if (tid == 0) p.counters[trel] = 0;           // the slice of the ticket pool goes back clean (the next launch that draws it starts behind a kernel boundary)
        const __amdgpu_buffer_rsrc_t rsS = __builtin_amdgcn_make_buffer_rsrc(p.slabs, 0, 0xFFFFFFF0u, 0x00020000);
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
          for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
        for (int g = g_first; g <= g_last; ++g) {
          // the part of workgroup g inside this tile is its first segment when its range starts inside the tile (or at its start)
          const long gs = (long)g * U / p.g_sk;
          const unsigned sb = (unsigned)((((long)g * 2 + (gs >= t0 ? 0 : 1)) * (BM * BN)) * 4) + (unsigned)tid * 16u;
#pragma unroll
          for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b) {
              f32x4 v[4];
#pragma unroll
              for (int q = 0; q < 4; ++q)
                v[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsS, sb + (unsigned)(((a * TN + b) * 4 + q) * TH * 16), 0, 16));
#pragma unroll
              for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[a][b][4 * q + e] += v[q][e];
            }
        }
      }
    }

    // ---- epilogue.  The MFMAs ran with the operands swapped (D = W-tile . x-tile^T), so a lane holds, per 32x32 tile, ONE output
    // row (lr) and four runs of 4 consecutive output columns (8g + 4lh ..): every store is 16 bytes, its per-lane offset is a kernel
    // constant, the tile / run position goes through the instruction's scalar offset (excluded from the range check), and only
    // edge tiles pay a select per store (invalid -> out-of-range offset, dropped by the hardware). ----
    if constexpr (TNF) {
      // dy^T x: acc[a][b][r] = C[m0 + wm*TM*32 + TM*rho + a][n0 + wn*TN*32 + TN*lr + b], rho = (r & 3) + 8 (r >> 2) + 4 lh: per (a, r) ONE store of
      // the lane's TN consecutive columns (32 lanes = TN*128 contiguous bytes of a row)
      if (finish) {
        static_assert(!TNF || TN == 4 || TN == 2, "store width");
        typedef float fTN __attribute__((ext_vector_type(TN)));
        const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(p.C, 0, 0x7FFFFFF0u, 0x00020000);
        const int col = n0 + wn * TN * 32 + TN * lr;
        const bool col_ok = col < p.N;                                 // N % 4 == 0 and TN | 4: a run is whole or absent
        const unsigned vlane = col_ok ? ((unsigned)(TM * 4 * lh) * (unsigned)p.ldc + (unsigned)col) * 4u : OOB;
        fTN bq;
#pragma unroll
        for (int b = 0; b < TN; ++b) bq[b] = 0.f;
        if (col_ok) {
          if (p.bias) bq += *reinterpret_cast<const fTN*>(p.bias + col);
          if (p.bias2) bq += *reinterpret_cast<const fTN*>(p.bias2 + col);
        }
        const bool rows_in = m0 + BM <= p.M;
        const bool acc_old = p.beta != 0.f;
        const int act = p.act;
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int rowq = m0 + wm * TM * 32 + TM * ((r & 3) + 8 * (r >> 2)) + a;     // scalar; the lane's row is rowq + TM * 4 * lh
            const unsigned soff = (unsigned)rowq * (unsigned)p.ldc * 4u;
            const unsigned vo = (rows_in || rowq + TM * 4 * lh < p.M) ? vlane : OOB;
            fTN v;
#pragma unroll
            for (int b = 0; b < TN; ++b) v[b] = apply_act(acc[a][b][r] + bq[b], act);
            if constexpr (TN == 4) {
              if (acc_old) v += __builtin_bit_cast(fTN, __builtin_amdgcn_raw_buffer_load_b128(rsC, vo, soff, 0));
              __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), rsC, vo, soff, 0);
            } else {
              typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
              if (acc_old) v += __builtin_bit_cast(fTN, __builtin_amdgcn_raw_buffer_load_b64(rsC, vo, soff, 0));
              __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_t, v), rsC, vo, soff, 0);
            }
          }
      }
    } else
    if (finish) {
      const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(p.C + (MODE == 0 ? cls * p.batch_c : 0L), 0, 0x7FFFFFF0u, 0x00020000);     // below OOB; the scalar offset is not range-checked
      const unsigned vlane = ((unsigned)(TNF ? TM * lr : lr) * (unsigned)p.ldc + 4u * (unsigned)lh) * 4u;
      const bool interior = !CONV && m0 + BM <= p.M && n0 + BN <= p.N;
      unsigned vrow[TM];
      const bool mapped = MODE == 0 && p.rowmap != nullptr && m0 + BM > p.ident_rows;       // (a tile of identity rows stores like an unmapped one)
      if constexpr (MODE == 0) {
        if (mapped) {
#pragma unroll
          for (int a = 0; a < TM; ++a) {
            const int row = m0 + (wm * TM + a) * 32 + lr;
            vrow[a] = row < p.M ? (unsigned)(((long)p.rowmap[row] * p.ldc + 4 * lh) * 4) : OOB;
          }
        }
      }
      if constexpr (CONV) {
#pragma unroll
        for (int a = 0; a < TM; ++a) {
          const int row = m0 + (wm * TM + a) * 32 + lr;
          bool ok = row < p.M;
          long off = (long)row * p.ldc;
          if (p.remap) {
            const int px = row % p.cPW, t2 = row / p.cPW, py = t2 % p.cPH, n = t2 / p.cPH;
            const int oy = py * p.osy + p.ooy[cls], ox = px * p.osx + p.oox[cls];
            ok = ok && oy < p.OHF && ox < p.OWF;
            off = (((long)n * p.OHF + oy) * p.OWF + ox) * p.ldc;
          }
          vrow[a] = ok ? (unsigned)((off + 4 * lh) * 4) : OOB;
        }
      }
      const bool has_bias = p.bias != nullptr, has_bias2 = p.bias2 != nullptr;
      const int act = p.act;
      const bool acc_old = p.beta != 0.f;
#pragma unroll
      for (int b = 0; b < TN; ++b)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int colq = n0 + (wn * TN + b) * 32 + 8 * g;              // scalar; this lane's run starts at colq + 4*lh
          f32x4 bq = {0.f, 0.f, 0.f, 0.f};
          const bool col_ok = colq + 4 * lh < p.N;                        // N % 4 == 0: a run is whole or absent
          if (has_bias && col_ok) bq += *reinterpret_cast<const f32x4*>(p.bias + colq + 4 * lh);
          if (has_bias2 && col_ok) bq += *reinterpret_cast<const f32x4*>(p.bias2 + colq + 4 * lh);
#pragma unroll
          for (int a = 0; a < TM; ++a) {
            const int rowq = TNF ? m0 + wm * TM * 32 + a : m0 + (wm * TM + a) * 32;      // scalar; this lane's row is rowq + lr (dy^T x: + TM * lr)
            unsigned soff = ((unsigned)rowq * (unsigned)p.ldc + (unsigned)colq) * 4u;
            unsigned vo = (interior || (col_ok && rowq + (TNF ? TM * lr : lr) < p.M)) ? vlane : OOB;
            if constexpr (CONV) { soff = (unsigned)colq * 4u; vo = col_ok ? vrow[a] : OOB; }
            if constexpr (MODE == 0) { if (mapped) { soff = (unsigned)colq * 4u; vo = col_ok ? vrow[a] : OOB; } }
            f32x4 v = {acc[a][b][4 * g], acc[a][b][4 * g + 1], acc[a][b][4 * g + 2], acc[a][b][4 * g + 3]};
            v += bq;
            if (act == RE2E_ACT_TANH) {
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = tanhf_(v[e]);
            } else if (act == RE2E_ACT_SIGMOID) {
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = sigmoidf_(v[e]);
            } else if (act == RE2E_ACT_RELU) {
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = v[e] < 0.f ? 0.f : v[e];       // torch semantics: NaN stays NaN, ReLU(-inf) = 0
            } else if (act == RE2E_ACT_LRELU) {
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = v[e] < 0.f ? 0.2f * v[e] : v[e];
            }
            if (acc_old) v += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsC, vo, soff, 0));
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), rsC, vo, soff, 0);
          }
        }
    }
    u0 += ke - kb;
  }
}

// ---- configurations ------------------------------------------------------------------------------------------------------
using N256x128k32s3 = NtCfg<4, 2, 2, 2, 32, 3, 2>;    // 8 waves, 144 KB LDS: one workgroup per CU
using N256x128k16s3 = NtCfg<4, 2, 2, 2, 16, 3, 4>;    // 8 waves, 72 KB, <= 128 registers: two workgroups per CU
using N128x128k32s2 = NtCfg<2, 2, 2, 2, 32, 2, 2>;    // 4 waves, 64 KB: two per CU
using N128x128k16s3 = NtCfg<2, 2, 2, 2, 16, 3, 3>;    // 4 waves, 48 KB: three per CU
using N128x64k16s4 = NtCfg<2, 2, 2, 1, 16, 4, 3>;     // 4 waves of 64x32, 48 KB: three per CU (few-column products: 320 columns = 5 tiles, not 2.5)
using N256x64k16s3 = NtCfg<4, 1, 2, 2, 16, 3, 2>;     // 4 waves of 64x64, 60 KB: two per CU
using T128x128k16s3 = NtCfg<2, 2, 2, 2, 16, 3, 3>;    // dy^T x: 4 waves of 64x64

struct NtVariant { int id, bm, bn, bk, wg_per_cu; double tflops; };
// tflops: what the variant sustains on a chip-filling product with whole rounds (tools/bench_gemm2.py, MI355X), the cost model's rate
const NtVariant VARIANTS[] = {{1, 256, 128, 32, 1, 130.0}, {3, 256, 128, 16, 2, 138.0}, {5, 128, 128, 32, 2, 130.0},
                              {6, 128, 128, 16, 3, 137.0}, {8, 128, 64, 16, 3, 136.0},  {9, 256, 64, 16, 2, 128.0}};

struct NtPlan { int variant; int wg_per_cu; int bm, bn, bk; int ntm, ntn, nkt, n_dp, g_sk; size_t bytes; double est; };

inline int num_cus() {
  static const int cus = [] { int dev = 0, n = 256; if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n; }();
  return cus;
}

// Schedule of one variant: whole tiles (DP) + stream-K tail, and the model's time for it (calibrated on profiles/r05_gemm_nt_variants.txt).
// Balance is per CU, whatever the residency: a CU's matrix pipe runs one tile per tile-time with one resident workgroup or with three
// (768 tiles on 256 CUs x 2 resident = three tiles per CU: no tail).  So the tail is what T leaves over whole rounds of CUs: those R tiles
// are cut along K into ONE unit range per CU (more parts only add slab traffic: 768 parts of a 128x128 tile ran 20 % SLOWER than whole tiles,
// 256 parts of a 256x128 tile 14 % faster).
NtPlan nt2_plan_variant(const NtVariant& v, int M, int N, int K, int sk) {
  NtPlan pl;
  memset(&pl, 0, sizeof(pl));
  pl.variant = v.id; pl.bm = v.bm; pl.bn = v.bn; pl.bk = v.bk; pl.wg_per_cu = v.wg_per_cu;
  pl.ntm = cdiv(M, pl.bm); pl.ntn = cdiv(N, pl.bn); pl.nkt = cdiv(K, pl.bk);
  const long T = (long)pl.ntm * pl.ntn, cus = num_cus();
  const long R = T % cus;
  const double slab = (double)v.bm * v.bn * 4.0;
  const double t_tile = 2.0 * v.bm * v.bn * (double)K / (v.tflops * 1e12 / cus);      // one CU, one tile
  const double t_over = 1.0e-6 + slab / 25e9;                                          // a tile's fill + epilogue as its CU sees them
  pl.n_dp = (int)T; pl.g_sk = 0;
  pl.est = (double)cdiv(T, cus) * (t_tile + t_over);
  if (sk && R != 0) {
    const long U = R * pl.nkt;
    // one unit range per CU -- except when the WHOLE product is the tail and the workgroup has 4 waves (RE2E_NT2_TAILWG, experiments: a lone
    // 4-wave workgroup leaves three quarters of a CU's wave slots empty for the whole launch): then two per CU
    long g = cus;
    if (T == R && v.bm * v.bn <= 128 * 256) { const char* e = exp_env("RE2E_NT2_TAILWG"); g = cus * (e ? atoi(e) : 1); }
    const long min_units = 128 / pl.bk;                  // a workgroup's share of the tail: at least 128 k
    if (U / g < min_units) g = U / min_units;
    if (g >= 8) g &= ~7L;                                // whole rounds of XCDs (the whole-tile workgroups behind keep blockIdx % 8 = XCD)
    // the tail runs one workgroup to a CU (~7 % under the two-resident rate); a cut tile's hand-off: its slab out write-through, the
    // parts back in by the last arriver; beside whole tiles of the other resident workgroups part of it hides
    const double parts = (double)g / (double)R + 1.0;          // parts a cut tile falls into: its last arriver reads them one after the other
    const double t_fix = (7e-6 + (double)g * slab / 2.4e12) * (T > R ? 0.7 : 1.0) + (parts > 3.0 ? (parts - 3.0) * slab / 80e9 : 0.0);
    const double est_sk = (double)(T / cus) * (t_tile + t_over) + (double)R / (double)g * t_tile / 0.93 + t_fix;
    if (g >= 2 && (sk == 2 || est_sk < pl.est)) { pl.n_dp = (int)(T - R); pl.g_sk = (int)g; pl.est = est_sk; }
  }
  pl.bytes = pl.g_sk ? (size_t)pl.g_sk * 2 * pl.bm * pl.bn * 4 : 0;
  return pl;
}

// RE2E_NT2 (experiments build, read at every call so that one process can compare variants):
//   "old" = the engine of igemm.hip;  "<variant>[,<sk>]": variant id of the table above (0 = choose), sk = 0 whole tiles only, 1 stream-K
//   tail where the model says it pays (default), 2 stream-K tail wherever there is a partial round
NtPlan nt2_plan(int M, int N, int K, bool filler, bool tn = false, int sk_override = -1) {
  int variant = 0, sk = 1;
  if (const char* e = exp_env(tn ? "RE2E_TN2" : "RE2E_NT2")) {
    if (e[0] == 'o') { NtPlan none; memset(&none, 0, sizeof(none)); return none; }
    variant = atoi(e);
    if (const char* c = strchr(e, ',')) sk = atoi(c + 1);
  }
  if (sk_override >= 0) sk = sk_override;
  NtPlan best;
  memset(&best, 0, sizeof(best));
  for (const NtVariant& v : VARIANTS) {
    if (variant ? v.id != variant : (tn ? v.id != 6 : (v.id == 1 || v.id == 5 || v.id == 9))) continue;      // candidates of the automatic choice: 3, 6, 8 (dy^T x: 6)
    // on a FILLER stream (work that runs beside resident recurrences, core.hip re2e_stream_role) only 4-wave tiles: they fit the registers
    // and LDS a recurrence workgroup leaves free on its CU
    if (!variant && filler && v.id == 3) continue;
    const NtPlan pl = nt2_plan_variant(v, M, N, K, sk);
    if (!best.variant || pl.est < best.est) best = pl;
  }
  return best;
}

// Ticket counters of cut tiles: slices of one zeroed pool per device, handed out round-robin; a slice is zero again when the launch that drew it
// has finished (each tile's last arriver resets its counter), so no memset node sits in front of a product.  256 slices: more launches with a
// stream-K tail than that would have to be in flight at once for two of them to share one.
constexpr int POOL_SLICES = 256, POOL_SLICE_INTS = 4096;
int* ticket_slice() {
  static std::mutex mu;
  static int* pool[16] = {nullptr};
  static unsigned next[16] = {0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 16) return nullptr;
  std::lock_guard<std::mutex> g(mu);
  if (!pool[dev]) {
    if (hipMalloc(&pool[dev], (size_t)POOL_SLICES * POOL_SLICE_INTS * sizeof(int)) != hipSuccess) { pool[dev] = nullptr; return nullptr; }
    if (hipMemset(pool[dev], 0, (size_t)POOL_SLICES * POOL_SLICE_INTS * sizeof(int)) != hipSuccess) return nullptr;
  }
  return pool[dev] + (size_t)(next[dev]++ % POOL_SLICES) * POOL_SLICE_INTS;
}

template <class CF, int MODE>
void nt2_launch(const NtArgs& a, const NtPlan& pl, hipStream_t st) {
  static LdsLimit lim;
  size_t lds = CF::LDS_BYTES;
  // residency is set through the LDS request: exactly wg_per_cu workgroups fit a CU's 160 KB
  // (RE2E_NT2_MAXWG, experiments build: fewer -- what a co-resident persistent recurrence workgroup needs is registers: 240 per SIMD for the 512-wide chains)
  int per_cu = pl.wg_per_cu;
  if (const char* e = exp_env("RE2E_NT2_MAXWG")) { const int m = atoi(e); if (m >= 1 && m < per_cu) per_cu = m; }
  const size_t want = (size_t)(160 * 1024) / (per_cu + 1) + 1024;
  if (lds < want) lds = want;
  lim.ensure(reinterpret_cast<const void*>(&gemm_nt2_kernel<CF, MODE>), lds);
  hipLaunchKernelGGL((gemm_nt2_kernel<CF, MODE>), dim3(pl.n_dp + pl.g_sk), dim3(CF::THREADS), lds, st, a);
}

}  // namespace

size_t gemm_nt2_workspace_bytes(int M, int N, int K) {
  if (K % 4) return 0;
  const size_t a = nt2_plan(M, N, K, false).bytes, b = nt2_plan(M, N, K, true).bytes;      // the stream's role is not known here
  return a > b ? a : b;
}

// Implicit-GEMM convolution forward / data gradient on the same pipeline (MODE 2).  g: geometry of ONE class (g.in = the image); ncls = 1, or 4 with
// per-class offsets / weight sets / output positions (stride-2 data gradient).  Returns 1 when launched here, 0 when left to igemm.hip's gather engine.
int conv_nt2(const ConvGeom& g, int M, const float* wg, int Cout, float* out, long ldc, const float* bias, int act, float beta, int ncls,
             const int* cls_oy0, const int* cls_ox0, long cls_wstride, int remap, int OHF, int OWF, int osy, int osx, const int* ooy, const int* oox,
             hipStream_t st) {
  if (const char* e = exp_env("RE2E_CONV_NT2")) { if (atoi(e) == 0) return 0; }      // (experiments build, read per call: same-session A/B against igemm.hip's gather engine)
  const int K = g.KH * g.KW * g.C;
  if (g.C % 16 || Cout % 4 || g.KH * g.KW > 32 || g.KH * g.KW < 2 || ncls < 1 || ncls > 4) return 0;
  if ((reinterpret_cast<uintptr_t>(g.in) & 15) || (reinterpret_cast<uintptr_t>(wg) & 15) || (reinterpret_cast<uintptr_t>(out) & 15) || ldc % 4) return 0;
  if (bias && (reinterpret_cast<uintptr_t>(bias) & 15)) return 0;
  if (act == RE2E_ACT_SIGMOID_MASK_MUL || (long)M * ncls > 0x7FFFFFFFL || M < 256) return 0;
  // the most negative tap position over all classes fixes how far in front of the tensor the descriptor starts
  const int ylo = g.DY < 0 ? (g.KH - 1) * g.DY : 0, xlo = g.DX < 0 ? (g.KW - 1) * g.DX : 0;
  int pady = 0, padx = 0;
  for (int c = 0; c < ncls; ++c) {
    const int oy = (cls_oy0 ? cls_oy0[c] : g.OY0) + ylo, ox = (cls_ox0 ? cls_ox0[c] : g.OX0) + xlo;
    if (-oy > pady) pady = -oy;
    if (-ox > padx) padx = -ox;
  }
  const long shift = ((long)pady * g.W + padx) * g.C * 4;
  const long in_bytes = (long)g.NI * g.H * g.W * g.C * 4;
  // rows past the last pixel (phantom rows of the last tile) carry an empty tap mask; their offsets may be anything, but stay 31-bit
  if (in_bytes + shift + (long)(g.KH * g.W + g.KW) * g.C * 4 * 4 >= 0x7FFFFFF0L) return 0;
  if ((long)ncls * Cout * K * 4 >= 0x7FFFFFF0L) return 0;
  // whole tiles only (the convolution entry points carry no workspace for partial slabs; the step's shapes fill their rounds: 500 / 250 / 4000 tiles)
  NtPlan q = nt2_plan(M, Cout, K, re2e_stream_is_filler(st), false, 0);
  if (!q.variant || g.C % q.bk) return 0;
  q.n_dp = ncls * q.ntm * q.ntn; q.g_sk = 0;
  NtArgs a;
  memset(&a, 0, sizeof(a));
  a.A = reinterpret_cast<const float*>(reinterpret_cast<const char*>(g.in) - shift); a.B = wg; a.C = out;
  a.a_bytes = (unsigned)(in_bytes + shift); a.b_bytes = (unsigned)((long)ncls * Cout * K * 4);
  a.lda = 0; a.ldb = K; a.ldc = ldc; a.M = M; a.N = Cout; a.K = K;
  a.bias = bias; a.act = act; a.beta = beta;
  a.ntm = q.ntm; a.ntn = q.ntn; a.n_dp = q.n_dp; a.g_sk = q.g_sk; a.nkt = q.nkt;
  a.cH = g.H; a.cW = g.W; a.cC = g.C; a.cPH = g.PH; a.cPW = g.PW; a.cKH = g.KH; a.cKW = g.KW; a.cSY = g.SY; a.cSX = g.SX; a.cDY = g.DY; a.cDX = g.DX;
  a.ncls = ncls; a.cls_wstride = cls_wstride; a.remap = remap; a.OHF = OHF; a.OWF = OWF; a.osy = osy; a.osx = osx;
  for (int c = 0; c < 4; ++c) {
    a.cOY0[c] = cls_oy0 ? cls_oy0[c < ncls ? c : 0] : g.OY0; a.cOX0[c] = cls_ox0 ? cls_ox0[c < ncls ? c : 0] : g.OX0;
    a.ooy[c] = ooy ? ooy[c < ncls ? c : 0] : 0; a.oox[c] = oox ? oox[c < ncls ? c : 0] : 0;
  }
  a.a_shift = (unsigned)shift;
  static const bool log_calls = getenv("RE2E_IGEMM_LOG") != nullptr;
  if (log_calls)
    fprintf(stderr, "[igemm] A=ConvK B=DenseK tile=%dx%dx%d vec=1 M=%d N=%d K=%d splits=%d\n", q.bm, q.bn, q.bk, M * ncls, Cout, K, q.g_sk ? -q.g_sk : 1);
  if (exp_env("RE2E_NT2_LOG")) fprintf(stderr, "[conv2] %dx%dx%d cls %d variant %d tiles %ld dp %d sk %d\n", M, Cout, K, ncls, q.variant, (long)ncls * q.ntm * q.ntn, q.n_dp, q.g_sk);
  const NtPlan& pl2 = q;
  switch (pl2.variant) {
    case 1: nt2_launch<N256x128k32s3, 2>(a, pl2, st); break;
    case 3: nt2_launch<N256x128k16s3, 2>(a, pl2, st); break;
    case 5: nt2_launch<N128x128k32s2, 2>(a, pl2, st); break;
    case 6: nt2_launch<N128x128k16s3, 2>(a, pl2, st); break;
    case 8: nt2_launch<N128x64k16s4, 2>(a, pl2, st); break;
    case 9: nt2_launch<N256x64k16s3, 2>(a, pl2, st); break;
    default: return 0;
  }
  return 1;
}

// K-sliced batch of x W^T products (igemm.hip gemm_kslices; wino44.hip's 25 positions): out[z][M][N] = A[:, z*Ks : (z+1)*Ks] . B[:, same]^T, whole tiles.
int gemm_nt2_kslices(int M, int N, int Ks, int ns, const float* A, long lda, const float* B, long ldb, float* out, hipStream_t st, int nolog) {
  if (const char* e = exp_env("RE2E_NT2")) { if (e[0] == 'o') return 0; }
  if (Ks % 16 || lda % 4 || ldb % 4 || N % 4 || (reinterpret_cast<uintptr_t>(A) & 15) || (reinterpret_cast<uintptr_t>(B) & 15) || (reinterpret_cast<uintptr_t>(out) & 15)) return 0;
  const long K = (long)Ks * ns;
  if (((long)(M - 1) * lda + K) * 4 >= 0x7FFFFFF0L || ((long)(N - 1) * ldb + K) * 4 >= 0x7FFFFFF0L || (long)M * N * 4 >= 0x7FFFFFF0L || M < 256 || ns > 4096) return 0;
  NtPlan q = nt2_plan(M, N, Ks, re2e_stream_is_filler(st), false, 0);
  if (!q.variant) return 0;
  q.n_dp = ns * q.ntm * q.ntn; q.g_sk = 0;
  NtArgs a;
  memset(&a, 0, sizeof(a));
  a.A = A; a.B = B; a.C = out;
  a.a_bytes = (unsigned)(((long)(M - 1) * lda + K) * 4); a.b_bytes = (unsigned)(((long)(N - 1) * ldb + K) * 4);
  a.lda = (int)lda; a.ldb = (int)ldb; a.ldc = N; a.M = M; a.N = N; a.K = Ks;
  a.act = RE2E_ACT_NONE;
  a.ntm = q.ntm; a.ntn = q.ntn; a.n_dp = q.n_dp; a.g_sk = 0; a.nkt = q.nkt;
  a.ncls = ns; a.batch_k = Ks; a.batch_c = (long)M * N;
  static const bool log_calls = getenv("RE2E_IGEMM_LOG") != nullptr;
  if (log_calls && !nolog)
    fprintf(stderr, "[igemm] A=DenseK B=DenseK tile=%dx%dx%d vec=1 M=%d N=%d K=%ld splits=%d\n", q.bm, q.bn, q.bk, M, N, K, ns);
  switch (q.variant) {
    case 1: nt2_launch<N256x128k32s3, 0>(a, q, st); break;
    case 3: nt2_launch<N256x128k16s3, 0>(a, q, st); break;
    case 5: nt2_launch<N128x128k32s2, 0>(a, q, st); break;
    case 6: nt2_launch<N128x128k16s3, 0>(a, q, st); break;
    case 8: nt2_launch<N128x64k16s4, 0>(a, q, st); break;
    case 9: nt2_launch<N256x64k16s3, 0>(a, q, st); break;
    default: return 0;
  }
  return 1;
}

// The dy^T x form (MODE 1) is NOT what the shipped library runs.  Measured against igemm.hip's transposing-stage kernel + split-K + reduce launch
// (profiles/r05_gemm_tn_variants.txt: first form, a ds_read_b32 per B tile; profiles/r05_gemm_tn_variants_v2.txt: both operands one wide read):
// 2048x2560x12800 128 vs 122 TFLOP/s, but 2048x512x12800 104 vs 110 and everything with fewer tiles far behind -- a weight gradient has few output
// tiles and a long K, so the WHOLE product is the stream-K tail: 4 to 32 parts per tile, summed by one last arriver each, where the old path's
// reduce launch spreads that sum over the chip.  Compiled only into the experiments build (RE2E_TN2 selects it there), 128x128 tile only: the
// 128x256 tile (four interleaved column tiles per wave, 232 registers) reached 134 TFLOP/s on 2048x2560x12800 but returned wrong, non-repeatable
// sums on products with many parts per tile (tools/stress_gemm_nt_tail.py, profiles/r05_stress_streamk_tail.txt: every x W^T variant and the
// 128x128 dy^T x form pass it, bitwise repeatable, 1-3 unit ranges per CU beside an unevenly loaded chip) -- not run down, removed.
#ifdef RE2E_EXPERIMENTS
#include "experiments/gemm_tn2.hip"     // RE2E_TN2: the rejected dy^T x launcher, experiments build only
#else
size_t gemm_tn2_workspace_bytes(int, int, int) { return 0; }
int gemm_tn2(int, int, int, const float*, long, const float*, long, float*, long, const float*, const float*, int, float, void*, size_t, hipStream_t) { return 0; }
#endif

// returns 1 when the product was launched here, 0 when the shape / alignment is left to igemm.hip
// rowmap / phys_rows (optional): M logical rows, row r = physical row rowmap[r] < phys_rows of A and of C (see NtArgs)
int gemm_nt2(int M, int N, int K, const float* A, long lda, const float* B, long ldb, float* C, long ldc, const float* bias, const float* bias2,
             int act, float beta, const float* mul, float* mask_out, const int* lens, int T, void* ws, size_t wsb, hipStream_t st,
             const int* rowmap, int phys_rows, int ident_rows) {
  const int Mp = rowmap ? phys_rows : M;           // rows the bounds are taken over
  if (K % 4 || lda % 4 || ldb % 4 || (reinterpret_cast<uintptr_t>(A) & 15) || (reinterpret_cast<uintptr_t>(B) & 15)) return 0;
  if (((long)(Mp - 1) * lda + K) * 4 >= 0x7FFFFFF0L || ((long)(N - 1) * ldb + K) * 4 >= 0x7FFFFFF0L) return 0;     // 31-bit offsets: OOB is bit 31
  if (M < 256 || Mp < M || act == RE2E_ACT_SIGMOID_MASK_MUL) return 0;
  (void)mul; (void)mask_out; (void)lens; (void)T;
  if (N % 4 || ldc % 4 || (reinterpret_cast<uintptr_t>(C) & 15) || ((long)(Mp - 1) * ldc + N) * 4 >= 0x7FFFFFF0L) return 0;      // 16-byte stores of 4 output columns
  if ((bias && (reinterpret_cast<uintptr_t>(bias) & 15)) || (bias2 && (reinterpret_cast<uintptr_t>(bias2) & 15))) return 0;
  // (mapped products: the 4-wave tiles only -- their row counts are whatever the batch's lengths give, and on such counts the cost model's
  //  256x128 choices miss: 21760 x 1024 x 512 ran 236 us on the 8-wave tile with a stream-K tail it priced at 196, 182 on 128x128 tiles)
  const NtPlan pl = nt2_plan(M, N, K, re2e_stream_is_filler(st) || rowmap != nullptr);
  if (!pl.variant) return 0;
  NtArgs a;
  memset(&a, 0, sizeof(a));
  if (pl.g_sk) {
    if (!ws || wsb < pl.bytes || (long)pl.ntm * pl.ntn - pl.n_dp > POOL_SLICE_INTS) return 0;
    a.counters = ticket_slice();
    if (!a.counters) return 0;
    a.slabs = (float*)ws;
  }
  a.A = A; a.B = B; a.C = C;
  a.a_bytes = (unsigned)(((long)(Mp - 1) * lda + K) * 4); a.b_bytes = (unsigned)(((long)(N - 1) * ldb + K) * 4);
  a.lda = (int)lda; a.ldb = (int)ldb; a.ldc = ldc; a.M = M; a.N = N; a.K = K;
  a.bias = bias; a.bias2 = bias2; a.act = act; a.beta = beta;
  a.ntm = pl.ntm; a.ntn = pl.ntn; a.n_dp = pl.n_dp; a.g_sk = pl.g_sk; a.nkt = pl.nkt;
  a.rowmap = rowmap; a.ident_rows = rowmap ? ident_rows : 0;
  static const bool nomem = exp_env("RE2E_IGEMM_NOMEM") != nullptr;
  a.nomem = nomem ? 1 : 0;

  static const bool log_calls = getenv("RE2E_IGEMM_LOG") != nullptr;
  if (log_calls)
    fprintf(stderr, "[igemm] A=DenseK B=DenseK tile=%dx%dx%d vec=1 M=%d N=%d K=%d splits=%d\n", pl.bm, pl.bn, pl.bk, M, N, K, pl.g_sk ? -pl.g_sk : 1);
  if (exp_env("RE2E_NT2_LOG")) fprintf(stderr, "[nt2] %dx%dx%d variant %d tiles %ld dp %d sk %d est %.1f us\n", M, N, K, pl.variant, (long)pl.ntm * pl.ntn, pl.n_dp, pl.g_sk, pl.est * 1e6);
  switch (pl.variant) {
    case 1: nt2_launch<N256x128k32s3, 0>(a, pl, st); break;
    case 3: nt2_launch<N256x128k16s3, 0>(a, pl, st); break;
    case 5: nt2_launch<N128x128k32s2, 0>(a, pl, st); break;
    case 6: nt2_launch<N128x128k16s3, 0>(a, pl, st); break;
    case 8: nt2_launch<N128x64k16s4, 0>(a, pl, st); break;
    case 9: nt2_launch<N256x64k16s3, 0>(a, pl, st); break;
    default: return 0;
  }
  return 1;
}

// ---- x W^T over the VALID rows of a ragged time-major batch ------------------------------------------------------------------------------
// C[map[r]][:] = act(A[map[r]][:] . B^T + bias + bias2) + beta C[map[r]][:]  for r < Mv: the product runs over Mv logical rows that sit at
// physical rows map[r] (ascending or not) of A (phys_rows x K, lda) and C (phys_rows x N, ldc).  The reference packs its sequences
// (pack_padded_sequence, e2e_encoder.py:129-131, enhance_model.py:120-123) and never computes the padded (t, b) rows; in the padded (T, B, .)
// layout the recurrences run on, those rows are 15 % of a config-4 batch (lengths 0.7 T .. T).  The per-lane row offsets of the LDS-DMA
// pipeline make the gather free: a lane's offset is map[row] * lda instead of row * lda, once per tile.  Rows not in the map are not touched:
// re2e_fill_rows puts zeros there where a consumer reads all rows.  RE2E_EUNSUPPORTED when the pipeline does not take the shape (the caller
// then runs the product over all physical rows).
extern "C" int re2e_gemm_nt_rows(int Mv, int N, int K, const float* A, long lda, const float* B, long ldb, float* C, long ldc, const float* bias,
                                 const float* bias2, int act, float beta, const int* rowmap, int ident_rows, int phys_rows, void* workspace,
                                 size_t workspace_bytes, hipStream_t stream) {
  RE2E_CHECK_ARG(Mv > 0 && N > 0 && K > 0 && phys_rows >= Mv && ident_rows >= 0 && ident_rows <= Mv, "bad sizes");
  RE2E_CHECK_ARG(A && B && C && rowmap, "null operand");
  RE2E_CHECK_ARG(beta == 0.f || beta == 1.f, "beta must be 0 or 1");
  RE2E_CHECK_ARG(act >= 0 && act < RE2E_ACT_SIGMOID_MASK_MUL, "bad activation");
  if (!gemm_nt2(Mv, N, K, A, lda, B, ldb, C, ldc, bias, bias2, act, beta, nullptr, nullptr, nullptr, 0, workspace, workspace_bytes, stream, rowmap,
                phys_rows, ident_rows)) {
    re2e_set_error("re2e_gemm_nt_rows: shape not taken by the pipeline (Mv=%d N=%d K=%d)", Mv, N, K);
    return RE2E_EUNSUPPORTED;
  }
  RE2E_LAUNCH_CHECK();
  return RE2E_OK;
}

namespace {
__global__ void fill_rows_kernel(float* __restrict__ C, long ldc, int N4, const int* __restrict__ rows, int nrows, float value,
                                 const float* __restrict__ row_vec, int act) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)nrows * N4) return;
  const int r = (int)(i / N4), c = (int)(i - (long)r * N4);
  f32x4 v = {value, value, value, value};
  if (row_vec) {
    // what x W^T + b with the activation leaves in a row whose x is zero: act(b) -- the same device function as the product's epilogue
    const f32x4 b = *reinterpret_cast<const f32x4*>(row_vec + 4 * c);
    v = f32x4{apply_act(b[0], act), apply_act(b[1], act), apply_act(b[2], act), apply_act(b[3], act)};
  }
  *reinterpret_cast<f32x4*>(C + (long)rows[r] * ldc + 4 * c) = v;
}
}  // namespace

// C[rows[i]][0 .. N) = row_vec ? act(row_vec[0 .. N)) : value  for i < nrows (N % 4 == 0, 16-byte aligned rows): the padded rows of a ragged batch
// behind re2e_gemm_nt_rows.  With row_vec = the bias of the product they hold what the reference's Linear + activation over the zero-padded
// frames leaves there (e2e_encoder.py:145-147,173-176: tanh(bias)).
extern "C" int re2e_fill_rows(float* C, long ldc, int N, const int* rows, int nrows, float value, const float* row_vec, int act, hipStream_t stream) {
  RE2E_CHECK_ARG(C && (rows || nrows == 0) && nrows >= 0 && N > 0, "bad argument");
  RE2E_CHECK_ARG(N % 4 == 0 && ldc % 4 == 0 && (reinterpret_cast<uintptr_t>(C) & 15) == 0, "rows must be 16-byte aligned multiples of 4 floats");
  RE2E_CHECK_ARG(!row_vec || ((reinterpret_cast<uintptr_t>(row_vec) & 15) == 0 && act >= 0 && act <= RE2E_ACT_SIGMOID), "row_vec must be 16-byte aligned, act a plain activation");
  if (nrows == 0) return RE2E_OK;
  const long tot = (long)nrows * (N / 4);
  hipLaunchKernelGGL(fill_rows_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, stream, C, ldc, N / 4, rows, nrows, value, row_vec, act);
  RE2E_LAUNCH_CHECK();
  return RE2E_OK;
}

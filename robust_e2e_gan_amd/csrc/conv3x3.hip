// K5: 3x3 / stride-1 / pad-1 convolution (forward and data gradient) over NHWC activations as a HALO-TILE
// implicit GEMM on the fp32 matrix core (v_mfma_f32_32x32x2_f32), gfx950 only.
//
// The general engine (igemm.hip) re-stages the im2col operand once per tap: every 16-channel k-step gathers a fresh
// 256 x 16 tile from global memory, 9 times per channel chunk, with per-lane tap/bounds arithmetic between the loads
// (profiles/r01_conv1_2_pmc_*.json: FETCH 3.4x the input, matrix pipe 69 % busy).  Here a workgroup owns a TH x TW patch of
// output pixels (256 of them) x 64 output channels and walks the input channels in chunks of 16:
//
//   per chunk   the (TH+2) x (TW+2) input patch WITH its halo (16 channels = one 64-byte segment per pixel) and the
//               9 x 64 x 16 weight slice are staged in LDS ONCE (register-staged: the loads of chunk c+1 are issued before
//               the matrix work of chunk c and written to LDS behind it),
//   9 taps      are then 9 constant LDS address offsets on the same patch: 288 MFMAs per wavefront between two barriers
//               (the engine: 32), no address arithmetic, no bounds checks, no global traffic inside the matrix block.
//
// Out-of-image halo pixels read as zero through the buffer descriptor's range check (offset 2^31), so padding costs no
// branch.  LDS rows are padded to 20 floats: one ds_read_b128 per fragment feeds 4 MFMAs (k = 8q + 4*(lane>>5) + j, the
// engine's mapping), conflict-free.  73.3 KB of LDS and <= 256 registers: two workgroups per CU, so one workgroup's staging /
// epilogue runs under the other's matrix block.  Workgroups are ordered XCD-aware (each XCD's L2 sees a contiguous run of
// patches, the two 64-channel halves of a 128-channel layer back to back).
//
// DIR = +1: forward (tap (a,b) reads pixel (y+a-1, x+b-1)); DIR = -1: data gradient (reads (y+1-a, x+1-b) of dy with the
// transposed gathered weights) -- the same two geometries re2e_conv_igemm receives from ops.Conv2dFn / ops.conv_dgrad.
//
// Replaces (reference): nn.Conv2d 3x3 of VGG2L, model/e2e_encoder.py:234-237,258-266 (conv1_2, conv2_1, conv2_2) and their
// autograd data gradients.
#include <stdlib.h>

#include "common.h"

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int CK = 16;        // input channels per chunk (one 64-byte segment per pixel)
constexpr int LDC = CK + 4;   // padded LDS row (floats)
constexpr int NT = 64;        // output channels per workgroup
constexpr unsigned OOB = 0x80000000u;   // byte offset beyond any tensor this path accepts (< 2 GiB): the load returns 0

struct HaloArgs {
  const float* in; const float* wg; float* out; const float* bias;
  const float* mask;                      // optional, shape of out: out = mask > 0 ? value : 0 (the ReLU derivative of the layer in front)
  float* pool_out; unsigned char* pool_idx;   // optional: ONLY the 2x2 / stride-2 ceil-mode max pool of relu(out) is written (+ its index bytes)
  int NI, H, W, C, Cout, act; float beta;
  int tiles_x, tiles_y, ngn, nitems;      // patches per row / column, 64-channel groups, work items = patches x groups
  int ipw;                                // items per workgroup; 0 = persistent workgroups
  unsigned in_bytes, wg_bytes, out_bytes;
#ifdef RE2E_HALO_STAMPS
  unsigned long long* stamps;     // diagnostic build only (tools/micro/conv3x3_probe.hip): 16 s_memtime stamps per workgroup
#endif
};

#ifdef RE2E_HALO_STAMPS
#define HALO_STAMP(k) do { if (tid == 0 && (k) < 16) { p.stamps[(size_t)blockIdx.x * 16 + (k)] = __builtin_amdgcn_s_memtime(); \
    if ((k) == 0 || (k) == 15) p.stamps[(size_t)gridDim.x * 16 + (size_t)blockIdx.x * 2 + ((k) ? 1 : 0)] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define HALO_STAMP(k) do { } while (0)
#endif

template <int TH, int TW, int DIR, bool RELU>
__global__ __launch_bounds__(256, 2) void conv3x3_halo_kernel(HaloArgs p) {
  constexpr int HPW = TW + 2, HPH = TH + 2, HP = HPW * HPH;
  constexpr int AIT = (HP * 4 + 255) / 256;            // float4 items per thread, input patch
  constexpr int ASZ = AIT * 64 * LDC;                  // floats (rows beyond HP are scratch for the surplus items)
  constexpr int RPS = 32 / TW;                         // patch rows per 32-pixel MFMA sub-tile
  constexpr int LDO = NT + 4;                          // row stride of the epilogue's turn-around buffer
  static_assert(TH * TW == 256 && 32 % TW == 0, "patch must hold 8 sub-tiles of 32 pixels");
  static_assert(4 * 64 * LDO <= ASZ + 9 * 64 * LDC, "the epilogue buffer reuses the staging buffers");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;
  float* Bs = smem + ASZ;                              // [9][64][LDC]

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, lr = lane & 31, lh = lane >> 5;
  // Work item = (patch, 64-channel group).  A workgroup takes `ipw` CONSECUTIVE items (the first fetch of item k+1 flies under
  // the last matrix block of item k).  ipw = 0: persistent workgroups, two per CU, XCD x owning the contiguous item range
  // [x*per, (x+1)*per).  Persistent workgroups were the fastest form alone on the chip while an item's set-up cost ~300 vector
  // instructions (13 % of the slots were empty with one item per workgroup) but hold every CU for the whole launch, and the
  // recurrent chains on the high-priority stream cannot place theirs (enhancer forward 10.0 -> 11.2 ms); with the set-up reduced
  // to scalar arithmetic one item per workgroup is as fast (launch_halo) and is the default.  Workgroups are ordered XCD-aware
  // (each XCD's L2 sees a contiguous run of patches).
  int item, item_end, nj;
  if (p.ipw == 0) {
    const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
    const int per = (p.nitems + 7) >> 3;
    nj = (gridDim.x + 7 - xcd) >> 3;
    item_end = min(p.nitems, (xcd + 1) * per);
    item = xcd * per + jx;
  } else {
    const int nwg = gridDim.x, orig = blockIdx.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = orig & 7;
    const int pid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
    nj = 1;
    item = pid * p.ipw;
    item_end = min(p.nitems, item + p.ipw);
  }

#ifdef RE2E_HALO_NOLOAD        // diagnostic builds only: zero-record descriptors, every load returns 0 without memory traffic
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0, 0, 0x00020000);
#else
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0, p.in_bytes, 0x00020000);
#endif
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wg), 0, p.wg_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc(p.out ? p.out : p.pool_out, 0, p.out ? p.out_bytes : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsM = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.mask ? p.mask : p.in), 0, p.mask ? p.out_bytes : 0, 0x00020000);

  // ---- staging: a wave-instruction moves 16 rows (pixels / output channels) x 64 bytes, 4 lanes per row.  (Measured and
  // rejected, same GPU session: a lane order that makes the ds_write_b128 of the 20-float rows bank-conflict-free -- 8
  // contiguous lanes = the same 16-byte piece of 8 rows -- shortens the staging by 0.5k cycles per item but its global
  // loads touch 4 lines per lane quad instead of 1 and the matrix block they fly under grows by 4k cycles.)
  const int srow = tid >> 2, skq = tid & 3;
  // Addresses.  Everything that depends on the LANE is computed once per kernel; everything that depends on the ITEM is
  // wave-uniform and goes through the buffer instructions' scalar offset, so an interior patch costs no vector instruction
  // per load or store.  (A wavefront outside its matrix block shares its SIMD with the other workgroup's MFMA stream and
  // gets about one VALU issue per MFMA: the kernel's matrix-pipe utilisation is set by how FEW vector instructions the
  // prologue / epilogue need, not by their latency.)
  //   a_rel[i]   byte offset of this lane's 16-byte piece of halo pixel hp = i*64 + srow relative to the halo's corner
  //              pixel (y0-1, x0-1); OOB for the surplus rows of the last pass
  //   a_off[i]   what the loads of the item being FETCHED use: a_rel (interior patch, scalar offset a_s = corner pixel) or
  //              absolute offsets with OOB for out-of-image pixels (border patch, a_s = 0)
  unsigned a_rel[AIT], a_off[AIT], a_s = 0, b_s = 0;
#pragma unroll
  for (int i = 0; i < AIT; ++i) {
    const int hp = i * 64 + srow, hy = hp / HPW, hx = hp - hy * HPW;
    a_rel[i] = hp < HP ? (unsigned)(((hy * p.W + hx) * p.C + skq * 4) * 4) : OOB;
  }
  const unsigned b_rel = (unsigned)(((srow * 9) * p.C + skq * 4) * 4);      // + scalar (group*64*9*C + tap*C + c0)*4
  // the item being fetched, decomposed; consecutive items advance it without divisions
  struct Item { int nblk, tx, ty, n; };
  auto decompose = [&](int it) {
    Item r;
    r.nblk = it % p.ngn;
    const int tile = it / p.ngn, t2 = tile / p.tiles_x;
    r.tx = tile - t2 * p.tiles_x;
    r.n = t2 / p.tiles_y;
    r.ty = t2 - r.n * p.tiles_y;
    return r;
  };
  auto advance = [&](Item& r, int it) {
    if (nj != 1) { r = decompose(it); return; }
    if (++r.nblk < p.ngn) return;
    r.nblk = 0;
    if (++r.tx < p.tiles_x) return;
    r.tx = 0;
    if (++r.ty < p.tiles_y) return;
    r.ty = 0; ++r.n;
  };
  auto set_item = [&](const Item& r) {
    const int y0 = r.ty * TH, x0 = r.tx * TW;
    b_s = (unsigned)(r.nblk * NT * 9 * p.C * 4);
    if (y0 >= 1 && x0 >= 1 && y0 + TH + 1 <= p.H && x0 + TW + 1 <= p.W) {
      a_s = (unsigned)((((r.n * p.H + y0 - 1) * p.W + x0 - 1) * p.C) * 4);
#pragma unroll
      for (int i = 0; i < AIT; ++i) a_off[i] = a_rel[i];
    } else {
      a_s = 0;
#pragma unroll
      for (int i = 0; i < AIT; ++i) {
        const int hp = i * 64 + srow;
        const int hy = hp / HPW, hx = hp - hy * HPW;
        const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
        const bool ok = (hp < HP) & ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W);
        a_off[i] = ok ? (unsigned)((((r.n * p.H + iy) * p.W + ix) * p.C + skq * 4) * 4) : OOB;   // < 2^31: checked by the launcher
      }
    }
  };
  const int a_dst = srow * LDC + skq * 4;                                   // + i*64*LDC
  const int b_dst = srow * LDC + skq * 4;                                   // + tap*64*LDC

  // ---- fragment addresses
  int a_frag[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int s = 2 * wid + i;
    const int py = s * RPS + lr / TW, px = lr % TW;
    a_frag[i] = ((py + 1) * HPW + (px + 1)) * LDC + 4 * lh;
  }
  const int b_frag = lr * LDC + 4 * lh;                                     // + j*32*LDC + tap*64*LDC + 8q

  f32x4 ra[AIT], rb[9];
  auto fetch = [&](int c0) {
    const unsigned cb = (unsigned)c0 * 4u;
#pragma unroll
    for (int i = 0; i < AIT; ++i) ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, a_off[i], a_s + cb, 0));
#pragma unroll
    for (int t = 0; t < 9; ++t)
      rb[t] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB, b_rel, b_s + (unsigned)(t * p.C * 4) + cb, 0));
  };
  auto stage = [&]() {
#pragma unroll
    for (int i = 0; i < AIT; ++i) *reinterpret_cast<f32x4*>(As + a_dst + i * 64 * LDC) = ra[i];
#pragma unroll
    for (int t = 0; t < 9; ++t) *reinterpret_cast<f32x4*>(Bs + b_dst + t * 64 * LDC) = rb[t];
  };

  HALO_STAMP(0);
  Item fi = {0, 0, 0, 0};
  if (item < item_end) { fi = decompose(item); set_item(fi); fetch(0); }
  // output: this lane's 16 row-wide stores of the epilogue, relative to the patch's first pixel / the group's first channel
  const int c4 = (lane & 15) * 4, prow = lane >> 4;
  unsigned o_rel[16];
#pragma unroll
  for (int it = 0; it < 16; ++it) {
    const int m = it * 4 + prow, s = 2 * wid + (m >> 5), q = m & 31;
    o_rel[it] = (unsigned)((((s * RPS + q / TW) * p.W + q % TW) * p.Cout + c4) * 4);
  }
  while (item < item_end) {
    const int n = fi.n, y0 = fi.ty * TH, x0 = fi.tx * TW, n0 = fi.nblk * NT;
    const int next = item + nj;
    // the bias is folded into the accumulators' initial value (a column of the tile = one output channel = one lane)
    f32x16 acc[2][2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const float bv = p.bias ? p.bias[n0 + j * 32 + lr] : 0.f;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = bv;
    }
    stage();                                        // chunk 0 of this item (fetched under the previous item's last chunk)
    __syncthreads();
    HALO_STAMP(1);
    for (int c0 = 0; c0 < p.C; c0 += CK) {
      const bool more = c0 + CK < p.C;
      // the next chunk -- or chunk 0 of the NEXT item -- is in flight under this chunk's 288 MFMAs
      if (more) fetch(c0 + CK);
      else if (next < item_end) { advance(fi, next); set_item(fi); fetch(0); }
      // 18 blocks of 16 MFMAs (9 taps x 2 channel octets); the fragments of block b+1 are read from LDS before the MFMAs of
      // block b are issued (two register sets), so no MFMA ever waits for an LDS read issued right in front of it
      f32x4 fa[2][2], fb[2][2];
      auto frags = [&](int blk, int set) {
        const int t = blk >> 1, q = blk & 1, a = t / 3, b = t % 3;
        const int tapoff = (DIR * (a - 1) * HPW + DIR * (b - 1)) * LDC;
#pragma unroll
        for (int i = 0; i < 2; ++i) fa[set][i] = *reinterpret_cast<const f32x4*>(As + a_frag[i] + tapoff + 8 * q);
#pragma unroll
        for (int j = 0; j < 2; ++j) fb[set][j] = *reinterpret_cast<const f32x4*>(Bs + b_frag + (t * 64 + j * 32) * LDC + 8 * q);
      };
      frags(0, 0);
#pragma unroll
      for (int blk = 0; blk < 18; ++blk) {
        const int set = blk & 1;
#ifdef RE2E_HALO_NOFRAG        // diagnostic builds only: one fragment read per chunk instead of 18
        if (blk == 0) frags(1, 1);
#else
        if (blk + 1 < 18) frags(blk + 1, set ^ 1);
#endif
        __builtin_amdgcn_sched_barrier(0);          // keep the reads IN FRONT of the block (the scheduler sinks them behind it otherwise)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set][i][jj], fb[set][j][jj], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      HALO_STAMP(2 + 3 * (c0 / CK));
      __syncthreads();                              // every wave is done reading this chunk
      HALO_STAMP(3 + 3 * (c0 / CK));
      if (more) {
        stage();
        __syncthreads();
      }
      HALO_STAMP(4 + 3 * (c0 / CK));
    }

    // ---- epilogue.  The accumulators hold, per register, 32 consecutive output channels of ONE pixel (lanes 0-31) and of
    // the pixel 4 rows of the sub-tile further (lanes 32-63); stored from there a wave-instruction moves 2 x 128 bytes.  Each
    // wavefront turns its 64 pixel x 64 channel block around in LDS instead (the staging buffers are free behind the last
    // barrier; row stride 68 floats) and stores whole 256-byte channel rows, 16 bytes per lane, 4 pixels per instruction.
    {
      float* Os = smem + wid * (64 * LDO);               // 17 KB per wavefront
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int m = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;       // pixel of this wavefront's 64
            Os[m * LDO + j * 32 + lr] = RELU ? (acc[i][j][r] < 0.f ? 0.f : acc[i][j][r]) : acc[i][j][r];      // torch.relu: NaN stays NaN (fmaxf would drop it)
          }
      // (no barrier: a wavefront reads back only what it wrote itself, and one wavefront's LDS operations execute in order)
      if (p.pool_out) {
        // conv -> ReLU -> 2x2 max pool: a wavefront's 64 pixels are 64 / TW full rows of the patch (TW = 16: 4 x 16, TW = 8: 8 x 8), i.e.
        // 16 whole pooling windows; lane = (4 channels, window quarter), 4 passes.  The full-resolution output never reaches memory.
        // Index byte: first maximum in row-major order, 4 when the maximum is <= 0 (maxpool2_fwd_kernel with relu_in).
        constexpr int WR = 64 / TW, PC = TW / 2;
        const int PH2 = (p.H + 1) >> 1, PW2 = (p.W + 1) >> 1;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int t = it * 4 + prow, pr = t / PC, pc = t - pr * PC;
          const int yy = y0 + wid * WR + 2 * pr, xx = x0 + 2 * pc;
          if (yy < p.H && xx < p.W) {
            f32x4 best = {-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f};
            int bi[4] = {0, 0, 0, 0};
#pragma unroll
            for (int d = 0; d < 4; ++d) {
              if (yy + (d >> 1) < p.H && xx + (d & 1) < p.W) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(Os + ((2 * pr + (d >> 1)) * TW + 2 * pc + (d & 1)) * LDO + c4);
#pragma unroll
                for (int k = 0; k < 4; ++k)
                  if (v[k] > best[k]) { best[k] = v[k]; bi[k] = d; }
              }
            }
            const long po = ((((long)n * PH2 + (yy >> 1)) * PW2 + (xx >> 1)) * p.Cout + n0 + c4);
            *reinterpret_cast<f32x4*>(p.pool_out + po) = best;
            typedef unsigned char uchar4h __attribute__((ext_vector_type(4)));
            uchar4h b4;
#pragma unroll
            for (int k = 0; k < 4; ++k) b4[k] = (unsigned char)(best[k] > 0.f ? bi[k] : 4);
            *reinterpret_cast<uchar4h*>(p.pool_idx + po) = b4;
          }
        }
      } else if (p.beta == 0.f && y0 + TH <= p.H && x0 + TW <= p.W) {
        // whole patch inside the image, plain store: lane-constant offsets + one scalar offset per item
        const unsigned o_s = (unsigned)(((((long)n * p.H + y0) * p.W + x0) * p.Cout + n0) * 4);
        if (p.mask) {
          // data gradient through the ReLU of the layer in front: four mask rows in flight at a time
#pragma unroll
          for (int i4 = 0; i4 < 16; i4 += 4) {
            f32x4 mk[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) mk[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsM, o_rel[i4 + u], o_s, 0));
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              f32x4 v = *reinterpret_cast<const f32x4*>(Os + ((i4 + u) * 4 + prow) * LDO + c4);
#pragma unroll
              for (int k = 0; k < 4; ++k) v[k] = mk[u][k] > 0.f ? v[k] : 0.f;
              __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsO, o_rel[i4 + u], o_s, 0);
            }
          }
        } else {
#pragma unroll
          for (int it = 0; it < 16; ++it) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(Os + (it * 4 + prow) * LDO + c4);
#ifndef RE2E_HALO_NOSTORE      // diagnostic builds only (tools/micro/conv3x3_probe.hip)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsO, o_rel[it], o_s, 0);
#else
            if (v[0] == 1.2345e30f) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsO, o_rel[it], o_s, 0);
#endif
          }
        }
      } else {
#pragma unroll
        for (int it = 0; it < 16; ++it) {
          const int m = it * 4 + prow;                                     // 0..63: sub-tile m>>5, pixel m&31 of it
          const int s = 2 * wid + (m >> 5), q = m & 31;
          const int y = y0 + s * RPS + q / TW, x = x0 + q % TW;
          if (y < p.H && x < p.W) {
            f32x4 v = *reinterpret_cast<const f32x4*>(Os + m * LDO + c4);
            float* dst = p.out + ((((long)n * p.H + y) * p.W + x) * p.Cout + n0 + c4);
            if (p.mask) {
              const f32x4 mk = *reinterpret_cast<const f32x4*>(p.mask + (dst - p.out));
#pragma unroll
              for (int k = 0; k < 4; ++k) v[k] = mk[k] > 0.f ? v[k] : 0.f;
            }
            if (p.beta != 0.f) {
              const f32x4 o = *reinterpret_cast<const f32x4*>(dst);
              v += o;
            }
#ifndef RE2E_HALO_NOSTORE
            *reinterpret_cast<f32x4*>(dst) = v;
#else
            if (v[0] == 1.2345e30f) *reinterpret_cast<f32x4*>(dst) = v;
#endif
          }
        }
      }
      __syncthreads();                                  // the turn-around buffer is the next item's staging buffer
    }
    HALO_STAMP(15);
    item = next;
  }
}

template <int TH, int TW, int DIR, bool RELU>
void launch_halo(const HaloArgs& a, hipStream_t st) {
  constexpr int HP = (TW + 2) * (TH + 2);
  constexpr int AIT = (HP * 4 + 255) / 256;
  constexpr size_t lds = (size_t)(AIT * 64 * LDC + 9 * 64 * LDC) * sizeof(float);
  static LdsLimit lim;
  lim.ensure(reinterpret_cast<const void*>(&conv3x3_halo_kernel<TH, TW, DIR, RELU>), lds);
  static const int slots = [] {          // two workgroups per CU (LDS and registers both allow exactly two)
    if (exp_env("RE2E_HALO_SLOTS")) return atoi(exp_env("RE2E_HALO_SLOTS"));   // occupancy experiments
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    return 2 * cus;
  }();
  // items per workgroup: RE2E_HALO_IPW = 0 persistent, n > 0 fixed.  Default 1: since an item's set-up is a handful of scalar
  // instructions (lane-constant addresses), a fresh workgroup per item costs nothing measurable alone on the chip (conv1_2 0.851
  // of peak against 0.838 / 0.852 / 0.850 with 2 / 3 / persistent, conv2_2 0.811 against 0.770 / 0.768 / 0.820: no half-empty
  // last round) and its 70 us lifetime interleaves best with the other streams of the training step (72.41 -> 72.07 ms against
  // the former 2-3 items; 3 + 3 runs, one GPU session).
  static const int ipw_env = exp_env("RE2E_HALO_IPW") ? atoi(exp_env("RE2E_HALO_IPW")) : -1;
  HaloArgs b = a;
  b.ipw = ipw_env >= 0 ? ipw_env : 1;
  const int nwg = b.ipw == 0 ? (a.nitems < slots ? a.nitems : slots) : (a.nitems + b.ipw - 1) / b.ipw;
  static const bool log_calls = getenv("RE2E_IGEMM_LOG") != nullptr;   // tools/igemm_table.py joins this with a kernel trace
  if (log_calls)
    fprintf(stderr, "[igemm] A=Halo%s B=DenseK tile=%dx%dx%d vec=1 M=%d N=%d K=%d splits=1\n", DIR > 0 ? "F" : "D", TH * TW, NT, CK,
            a.NI * a.H * a.W, a.Cout, 9 * a.C);
  hipLaunchKernelGGL((conv3x3_halo_kernel<TH, TW, DIR, RELU>), dim3((unsigned)nwg), dim3(256), lds, st, b);
}

}  // namespace

// Returns true when the geometry is a 3x3 / stride-1 / pad-1 convolution (forward or data-gradient form) this kernel
// covers and the launch was enqueued; false -> the caller uses the general engine.
bool halo_conv3x3(const ConvGeom& g, const float* wg, int Cout, float* out, const float* bias, int act, float beta, const float* mask,
                  hipStream_t st, float* pool_out, unsigned char* pool_idx) {
  static const bool off = exp_env("RE2E_NO_HALO") != nullptr;     // A/B measurements against the general engine
  if (off) return false;
  if (g.KH != 3 || g.KW != 3 || g.SY != 1 || g.SX != 1 || g.PH != g.H || g.PW != g.W) return false;
  int dir;
  if (g.DY == 1 && g.DX == 1 && g.OY0 == -1 && g.OX0 == -1) dir = 1;
  else if (g.DY == -1 && g.DX == -1 && g.OY0 == 1 && g.OX0 == 1) dir = -1;
  else return false;
  if (g.C % CK || Cout % NT) return false;
  if (act != RE2E_ACT_NONE && act != RE2E_ACT_RELU) return false;
  if ((reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(mask) | reinterpret_cast<uintptr_t>(pool_out)) & 15) return false;
  if (pool_out && (act != RE2E_ACT_RELU || beta != 0.f || mask || dir != 1 || (reinterpret_cast<uintptr_t>(pool_idx) & 3))) return false;
  const long in_bytes = (long)g.NI * g.H * g.W * g.C * 4, wg_bytes = (long)Cout * 9 * g.C * 4;
  const long out_bytes = (long)g.NI * g.H * g.W * Cout * 4;
  if (in_bytes >= 0x7FFFFF00L || wg_bytes >= 0x7FFFFF00L || out_bytes >= 0x7FFFFF00L) {
    // the buffer descriptors address 2 GiB: a geometry this kernel would otherwise cover goes to the general engine (~15 % slower
    // at the VGG shapes) -- say so once instead of losing the time silently (e.g. B = 64 per GPU at T = 800: 2.1 GB activations)
    static std::once_flag warned;
    std::call_once(warned, [&] {
      fprintf(stderr, "[re2e] note: 3x3 convolution %dx%dx%d, %d->%d channels has a tensor of >= 2 GiB: the halo-patch kernel declines it, "
                      "the general implicit-GEMM engine runs it (slower; split the batch to stay below 2 GiB per tensor)\n",
              g.NI, g.H, g.W, g.C, Cout);
    });
    return false;
  }
  if ((reinterpret_cast<uintptr_t>(g.in) | reinterpret_cast<uintptr_t>(wg)) & 15) return false;
  HaloArgs a;
  a.in = g.in; a.wg = wg; a.out = pool_out ? nullptr : out; a.bias = bias; a.mask = mask; a.pool_out = pool_out; a.pool_idx = pool_idx;
  a.NI = g.NI; a.H = g.H; a.W = g.W; a.C = g.C; a.Cout = Cout; a.act = act; a.beta = beta;
  a.ngn = Cout / NT; a.in_bytes = (unsigned)in_bytes; a.wg_bytes = (unsigned)wg_bytes; a.out_bytes = (unsigned)out_bytes;
  // patch shape: 16 x 16 (smaller halo) unless 32 x 8 wastes fewer padded pixels (W = 40: 416 x 40 against 400 x 48)
  const long pad16 = (long)cdiv(g.H, 16) * 16 * cdiv(g.W, 16) * 16, pad8 = (long)cdiv(g.H, 32) * 32 * cdiv(g.W, 8) * 8;
  const bool wide = pad16 <= pad8;
  const int TH = wide ? 16 : 32, TW = wide ? 16 : 8;
  a.tiles_x = cdiv(g.W, TW); a.tiles_y = cdiv(g.H, TH);
  const long nitems = (long)a.NI * a.tiles_x * a.tiles_y * a.ngn;
  if (nitems >= 0x7FFFFFF0L) return false;
  a.nitems = (int)nitems;
  const bool relu = act == RE2E_ACT_RELU;
#define HALO_GO(TH_, TW_) do { \
    if (dir > 0) { if (relu) launch_halo<TH_, TW_, 1, true>(a, st); else launch_halo<TH_, TW_, 1, false>(a, st); } \
    else { if (relu) launch_halo<TH_, TW_, -1, true>(a, st); else launch_halo<TH_, TW_, -1, false>(a, st); } } while (0)
  if (wide) HALO_GO(16, 16); else HALO_GO(32, 8);
#undef HALO_GO
  return true;
}

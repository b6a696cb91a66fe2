// K5: 3x3 / stride-1 / pad-1 convolution (forward and data gradient) over NHWC activations as a FUSED Winograd F(2x2, 3x3)
// on the fp32 matrix core (v_mfma_f32_32x32x2_f32), gfx950 only.
//
// A 2x2 block of output pixels is  Y = A^T [ (G g G^T) . (B^T d B) ] A  with d the 4x4 input tile around it: 16 element-wise
// products per (input channel, output channel) instead of 36 -- the convolution becomes 16 independent GEMMs (one per position
// of the transformed tile) with 2.25x fewer matrix instructions than the direct form (conv3x3.hip).  The transforms only add
// and halve (B and A hold 0 / +-1, G holds 0 / 1 / +-0.5), so in fp32 the result differs from the direct sum by a few ulp.
//
// Everything between the global loads and the output transform lives in REGISTERS, and the main loop has no barrier:
//
//   workgroup   = 32 tiles (a 16 x 8 or 8 x 16 pixel patch) x 64 output channels; wavefront i owns ROW i of the transformed
//                 tile: positions (i, 0..3) = 4 GEMMs of 32 tiles x 64 channels (8 accumulator tiles, 128 registers);
//   A operand   : lane (tile, k-half) loads the two input rows that B^T's row i touches (8 pixels x 4 channels, 16 bytes each),
//                 transforms them in registers (8 packed adds per 4 channels) and feeds the MFMAs directly -- the im2col
//                 operand never exists, neither in LDS nor in memory;
//   B operand   : the transformed weights U[pos][cout][c] are prepared once per call in MFMA fragment order
//                 ([group][chunk][pos][n-tile][lane][4], one contiguous KB per fragment) and stream L2 -> registers;
//   pipeline    : input channels in chunks of 8 (32 MFMAs per wavefront and chunk); the loads of chunk c+1 are issued as soon
//                 as the registers of chunk c are free (the raw pixels right after the transform, each position's weights right
//                 after its MFMAs), so a load has a whole chunk (~2 000 cycles) to arrive;
//   epilogue    : each wavefront applies the column half of A^T . A to its own accumulators (4 -> 2 values), the four rows meet
//                 in LDS (68 KB) and are combined into the 2x2 outputs; bias, ReLU, the ReLU mask of the layer in front (data
//                 gradient) or the 2x2 max pool -- a tile IS a pooling window -- are applied there and whole 256-byte channel
//                 rows are stored.
//
// Out-of-image pixels read as zero through the buffer descriptor's range check.  <= 256 registers and 68 KB of LDS: two
// workgroups per CU -- they take turns on a SIMD rather than overlap (a wavefront streaming MFMAs keeps the issue grant, and vector-ALU
// instructions never run beside fp32 MFMAs: DESIGN.md 4.2), the second one covers the first one's stalls.
// Workgroups are ordered XCD-aware.
//
// Replaces (reference): nn.Conv2d 3x3 of VGG2L, model/e2e_encoder.py:234-237,258-266 (conv1_2, conv2_1, conv2_2) and their
// autograd data gradients.
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int WCK = 8;                  // input channels per chunk
constexpr int WNT = 64;                 // output channels per workgroup
constexpr int LDR = WNT + 4;            // row stride (floats) of the epilogue's exchange buffer: [row i][b][32 tiles][64 channels + pad]
constexpr unsigned WOOB = 0x80000000u;  // byte offset beyond any tensor this path accepts (< 2 GiB): the load returns 0

struct WinoArgs {
  const float* in; const float* ufrag; float* out; const float* bias;
  const float* mask;                          // optional, shape of out: out = mask > 0 ? value : 0
  float* pool_out; unsigned char* pool_idx;   // optional: only maxpool2(relu(out)) and its index bytes are written
  int NI, H, W, C, Cout, relu;
  int tiles_x, tiles_y, ngn_shift, per_image; // patches per row / column, log2(64-channel groups), work items per image
  unsigned long long* stamps;                 // RE2E_EXPERIMENTS builds only: 8 s_memtime stamps per wavefront of the first 4096 workgroups
  int dbg;                                    // RE2E_EXPERIMENTS builds only: 1 = weight loads always hit the same 2 KB, 2 = pixel loads always read chunk 0
  unsigned in_bytes, u_bytes, out_bytes, pool_bytes;
  int total, ipw;                             // pipelined form: work items of the launch (per_image * NI) and consecutive items per workgroup
  const int* row_lim;                         // optional, per image: output rows >= row_lim[n] are not computed (patches that start there exit at once)
};

// U[g][chunk][pos][nt][lane][e] = sum_{a,b} G[i][a] G[j][b] k[o][c][a][b],  pos = 4i + j, o = 64 g + 32 nt + (lane & 31),
// c = 8 chunk + 4 (lane >> 5) + e.  k = the correlation kernel of THIS call: forward k[o][c][a][b] = w[o][c][a][b];
// data gradient (in = dy, out = dx) k[o][c][a][b] = w[c][o][2-a][2-b]  (w in PyTorch's (Cout_f, Cin_f, 3, 3) layout).
__global__ void wino_weights_kernel(const float* __restrict__ w, int Cout, int C, int dgrad, float* __restrict__ uf) {
  const long total = (long)16 * Cout * C;
  const int nch = C / WCK;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int e4 = (int)(e & 3), lane = (int)((e >> 2) & 63), nt = (int)((e >> 8) & 1), pos = (int)((e >> 9) & 15);
    const long r = e >> 13;
    const int chunk = (int)(r % nch), g = (int)(r / nch);
    const int o = g * WNT + nt * 32 + (lane & 31), c = chunk * WCK + 4 * (lane >> 5) + e4;
    float k[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b)
        k[a][b] = dgrad ? w[(((long)c * Cout + o) * 3 + (2 - a)) * 3 + (2 - b)] : w[(((long)o * C + c) * 3 + a) * 3 + b];
    const int i = pos >> 2, j = pos & 3;
    // row transform with G row i, then column transform with G row j;  G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]
    float t[3];
#pragma unroll
    for (int b = 0; b < 3; ++b)
      t[b] = i == 0 ? k[0][b] : (i == 1 ? 0.5f * (k[0][b] + k[1][b] + k[2][b]) : (i == 2 ? 0.5f * (k[0][b] - k[1][b] + k[2][b]) : k[2][b]));
    const float u = j == 0 ? t[0] : (j == 1 ? 0.5f * (t[0] + t[1] + t[2]) : (j == 2 ? 0.5f * (t[0] - t[1] + t[2]) : t[2]));
    uf[e] = i == 2 ? -u : u;          // the kernel's row-2 input transform is d1 - d2 = -(B^T d)_2: the sign moves into the weights
  }
}

#ifdef RE2E_EXPERIMENTS
#define WSTAMP(k) do { if (p.stamps && lane == 0 && blockIdx.y == 0 && blockIdx.x < 4096) \
    p.stamps[((size_t)blockIdx.x * 4 + wid) * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define WSTAMP(k) do { } while (0)
#endif

// LDSIN (round 6, C % 64 == 0): the input patch travels global -> LDS by LDS-DMA, 32 channels at a time, and the lanes read their tile's pixels out of
// LDS.  Before, every lane loaded its own 16 bytes of 8 pixels per chunk straight from memory: 64 separate cache lines per load instruction, every
// pixel fetched by all four wavefronts and by both tiles that share it -- 0.88 vector-L1 accesses per cycle and CU, the L1's limit: the kernel was bound
// by the ACCESS RATE of the L1, not by issue or latency (TCP_TOTAL_CACHE_ACCESSES, and the diagnostic that read 1 KB contiguous per instruction instead:
// conv1_2 1.47 -> 1.26 ms with the same bytes and instructions; profiles/r06_wino_attribution.md).  Staged, a (PH + 2) x (PW + 2) patch of 32 channels
// is 24 load instructions per workgroup instead of 128, and each pixel is fetched once.
//   LDS image of a half (32 channels): [2 planes of 16 channels][192 slots][64 bytes], slot(py, px) = py * IPW + (px & 1) * IPW / 2 + px / 2 (even
//   columns first: the 8 tiles of a patch row read consecutive slots), the four 16-byte granules of a slot ROTATED by slot / 2 -- so that
//   (a) a request instruction covers 16 pixels x 64 contiguous bytes (4 lanes per pixel: 16 cache lines per instruction; the first form,
//   16-byte granule planes, asked 64 lines per instruction and cost 5 % of the six launches: profiles/r06_wino_attribution.md), and (b) the
//   eight lanes a ds_read_b128 serves together hit eight different 16-byte bank groups (conflict-free).  The rotation makes a granule's position
//   lane-dependent, so a lane keeps TWO read addresses per pixel (the two chunks of a plane); plane and buffer are instruction immediates:
//   no address arithmetic in the loop.  Two halves are resident (2 x 24 KB, aliased with the epilogue's exchange buffer); half h + 2 is
//   requested when the last chunk of half h has left LDS (one workgroup barrier per half).
template <int TXW, bool LDSIN, bool C64 = false>      // TXW = tiles per patch row: 8 (16 x 8 pixel patch) or 4 (8 x 16); C64: LDSIN with exactly 64 input channels
__global__ __launch_bounds__(256, 2) void wino_conv3x3_kernel(WinoArgs p) {
  static_assert(LDSIN || !C64, "C64 is a variant of the LDS-staged kernel");
  constexpr int TYH = 32 / TXW;
  constexpr int PW = 2 * TXW, PH = 2 * TYH;
  constexpr int IPW = PW + 2, IPH = PH + 2, NPIX = IPW * IPH;        // the input patch: 18 x 10 or 10 x 18 pixels
  constexpr int NSLOT = 192, PLANE = NSLOT * 64, HALFB = 2 * PLANE;  // bytes; 2 planes x 192 slots x 64 bytes = 24 LDS-DMA instructions of 64 lanes
  static_assert(NPIX <= NSLOT && IPW % 2 == 0, "patch geometry");
  extern __shared__ __attribute__((aligned(16))) float smem[];     // [4 rows i][2 b][32 tiles][LDR]

  const int tid = threadIdx.x, lane = tid & 63, lr = lane & 31, lh = lane >> 5;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform, and the compiler must know it (scalar offsets, scalar branches)
  // ---- work item: blockIdx.y = image, blockIdx.x = (patch, 64-channel group) of it in XCD-aware order (each XCD's L2 sees a
  // contiguous run of patches).  Measured and rejected (round 3, same GPU session): 2-8 consecutive items per workgroup with the
  // first loads of item k+1 issued under the last chunk and the epilogue of item k (the per-workgroup time per item fell 20 %, the
  // kernel's did not move: what an item costs beyond its 256 MFMAs per wavefront is instruction issue beside the other workgroup's
  // matrix stream, not latency); starting the second workgroup of every CU half an item late (no effect either way).
  int item;
  {
    const int nwg = gridDim.x, orig = blockIdx.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = orig & 7;
    item = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
  }
  const int n = blockIdx.y;

  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0, p.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsU = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.ufrag), 0, p.u_bytes, 0x00020000);
  // output side: 32-bit offsets through descriptors as well; a pixel outside the image gets the out-of-range offset and its store
  // is dropped (its mask load returns 0) -- no branch, no 64-bit address arithmetic in the epilogue
  const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc(p.pool_out ? p.pool_out : p.out, 0, p.pool_out ? p.pool_bytes : p.out_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsM = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.mask ? p.mask : p.in), 0, p.mask ? p.out_bytes : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsI = __builtin_amdgcn_make_buffer_rsrc(p.pool_idx ? (void*)p.pool_idx : (void*)p.in, 0, p.pool_idx ? p.pool_bytes / 4 : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.bias ? p.bias : p.in), 0, p.bias ? (unsigned)p.Cout * 4u : 0u, 0x00020000);

  // ---- a lane's 8 input pixels: tile (tyi, txi) = lr, rows (ra, rb) of its 4x4 input tile, 4 columns, channels 4 lh .. 4 lh + 3
  const int tyi = lr / TXW, txi = lr - tyi * TXW;
  const int ra = wid == 0 ? 0 : 1, rb = wid == 3 ? 3 : 2;            // B^T row i touches input rows (0,2) (1,2) (1,2) (1,3)
  const int rowb = p.W * p.C * 4, colb = p.C * 4;
  const unsigned a_s = (unsigned)(n * p.H * p.W * p.C * 4);          // the image; < 2^31: checked by the launcher
  const int lane_base = (2 * tyi - 1) * rowb + (2 * txi - 1) * colb + 16 * lh;
  const int nch = p.C / WCK;
  const unsigned u_lane = (unsigned)lane * 16u;

  struct Geo { int y0, x0, nblk; };
  auto geo_of = [&](int it) {
    Geo g;
    g.nblk = it & ((1 << p.ngn_shift) - 1);
    const int tile_id = it >> p.ngn_shift;
    const int ty = __builtin_amdgcn_readfirstlane(tile_id / p.tiles_x);     // (vector-ALU quotient: tell the compiler it is uniform)
    g.y0 = ty * PH;
    g.x0 = (tile_id - ty * p.tiles_x) * PW;
    return g;
  };
  unsigned a_off[8];
  unsigned u_s0;         // fragment (chunk, position 4 wid + j, n-tile nt) of group nblk: ((((nblk nch + chunk) 16 + 4 wid + j) 2 + nt) KB
  auto set_item = [&](const Geo& g) {
    u_s0 = (unsigned)(((g.nblk * nch) * 16 + 4 * wid) * 2) * 1024u;
    const int base = lane_base + g.y0 * rowb + g.x0 * colb;
    if (g.y0 >= 1 && g.x0 >= 1 && g.y0 + PH + 1 <= p.H && g.x0 + PW + 1 <= p.W) {
#pragma unroll
      for (int k = 0; k < 8; ++k) a_off[k] = (unsigned)(base + (k < 4 ? ra : rb) * rowb + (k & 3) * colb);
    } else {
      const int iy0 = g.y0 - 1 + 2 * tyi, ix0 = g.x0 - 1 + 2 * txi;
      const bool rok0 = (unsigned)(iy0 + ra) < (unsigned)p.H, rok1 = (unsigned)(iy0 + rb) < (unsigned)p.H;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const bool ok = (k < 4 ? rok0 : rok1) & ((unsigned)(ix0 + (k & 3)) < (unsigned)p.W);
        a_off[k] = ok ? (unsigned)(base + (k < 4 ? ra : rb) * rowb + (k & 3) * colb) : WOOB;
      }
    }
  };

  // ---- LDSIN: per-lane constants of the staging requests (6 per wavefront and half) and of the fragment reads (8 pixels)
  unsigned st_off[3];       // byte offset of the lane's 16 bytes inside the image (channel half 0, plane 0) for its three slots, or out of range
  unsigned rd_off[8][2];    // LDS byte offset of pixel k's granule (this lane's k-half) for the first / second chunk of a 16-channel plane
  const char* const sbytes = reinterpret_cast<const char*>(smem);
  // Request instruction q = wid + 4 i (i < 6) of a half covers plane q / 12, slots [16 (q % 12), + 16), four lanes per slot.  q % 12 = wid, wid + 4,
  // wid + 8 for i = 0, 1, 2 and AGAIN for i = 3, 4, 5 (the second plane: + 64 bytes, through the instruction's scalar offset): a lane has three
  // slots, 64 apart, and since 64 / 2 is a multiple of 4 the same granule rotation in all of them -- one division per work item, the rest are adds
  // (the first form of this set-up cost 0.85 vector instructions per MFMA: profiles/r06_conv1_2_wino_pmc_sq.json against r05).
  auto set_item_lds = [&](const Geo& g) {
    u_s0 = (unsigned)(((g.nblk * nch) * 16 + 4 * wid) * 2) * 1024u;
    const bool interior = g.y0 >= 1 && g.x0 >= 1 && g.y0 + PH + 1 <= p.H && g.x0 + PW + 1 <= p.W;
    int slot = wid * 16 + (lane >> 2);                              // < 64
    int py = slot / IPW, rem = slot - py * IPW;
    const int gr16 = (((lane & 3) - (slot >> 1)) & 3) * 16;         // LDS position lane & 3 of a slot holds granule (position - slot / 2) mod 4
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int px = rem < IPW / 2 ? 2 * rem : 2 * (rem - IPW / 2) + 1;
      const int iy = g.y0 - 1 + py, ix = g.x0 - 1 + px;
      const bool ok = slot < NPIX && (interior || ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W));
      st_off[i] = ok ? (unsigned)(iy * rowb + ix * colb + gr16) : WOOB;
#ifdef RE2E_EXPERIMENTS
      // diagnostic (RE2E_WINO_DBG bit 128, timing only): every request instruction reads 1 KB CONTIGUOUS (wrong pixels)
      if (p.dbg & 128) st_off[i] = (unsigned)((g.y0 * rowb + g.x0 * colb) + (wid + 4 * i) * 1024 + lane * 16);
#endif
      slot += 64; py += 64 / IPW; rem += 64 % IPW;
      if (rem >= IPW) { rem -= IPW; ++py; }
    }
  };
  auto stage = [&](int half, int buf) {                            // half `half` of the channels -> LDS buffer `buf`
#ifdef RE2E_EXPERIMENTS
    if (p.dbg & 64) return;                                        // diagnostic (timing only): no pixel traffic at all
#endif
    const unsigned so = a_s + (unsigned)half * 128u;
    float* base = smem + buf * (HALFB / 4) + wid * 256;
#pragma unroll
    for (int i = 0; i < 6; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(base + i * 1024), 16, st_off[i % 3], so + (unsigned)(i / 3) * 64u, 0, 0);
  };
  if constexpr (LDSIN) {
    const int base_lane = 2 * tyi * IPW + txi;                    // slot of the tile's pixel (0, 0); pixel (r, c) adds a wave-uniform constant
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int r = k < 4 ? ra : rb, c = k & 3;
      const int slot = base_lane + (r * IPW + (c & 1) * (IPW / 2) + (c >> 1));
      rd_off[k][0] = (unsigned)(slot * 64 + ((lh + (slot >> 1)) & 3) * 16);            // granule lh (chunk 0 of the plane) ...
      rd_off[k][1] = rd_off[k][0] ^ 32u;                                               // ... and granule 2 + lh (chunk 1): the position + 2 mod 4
    }
  }
  f32x4 raw[8], uf[4][2];
  auto read_raw = [&](int buf, int cc) {                           // chunk position cc (8 channels: plane cc / 2, its first or second chunk) of the half in `buf`
#pragma unroll
    for (int k = 0; k < 8; ++k) raw[k] = *reinterpret_cast<const f32x4*>(sbytes + rd_off[k][cc & 1] + (buf * HALFB + (cc >> 1) * PLANE));
  };
#ifdef RE2E_EXPERIMENTS
  // diagnostic (RE2E_WINO_DBG bit 16): the eight pixel loads of a chunk read 1 KB CONTIGUOUS each (lane-linear, wrong pixels) instead of 64 separate
  // 16-byte pieces 512 bytes apart: same instruction stream, same bytes, 1/4 .. 1/8 of the vector-L1 accesses -- is the kernel bound by the L1's
  // access rate (TCP_TOTAL_CACHE_ACCESSES: 0.88 per cycle and CU, profiles/r06_wino_attribution.md)?
  const bool coalesced_dbg = (p.dbg & 16) != 0;
#endif
  auto fetch_raw = [&](int chunk) {
    unsigned cb = a_s + (unsigned)(chunk * WCK * 4);
#ifdef RE2E_EXPERIMENTS
    if (p.dbg & 2) cb = a_s;                          // diagnostic: every chunk re-reads chunk 0 (cache hits): is the pixel traffic the limit?
#endif
#pragma unroll
    for (int k = 0; k < 8; ++k) raw[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, a_off[k], cb, 0));
  };
  auto fetch_u = [&](int chunk, int j) {
    unsigned s = u_s0 + (unsigned)(chunk * 16 * 2 + j * 2) * 1024u;
#ifdef RE2E_EXPERIMENTS
    if (p.dbg & 1) s = 0;                             // diagnostic: every fragment load reads the same 2 KB: is the weight traffic the limit?
#endif
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) uf[j][nt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsU, u_lane, s + (unsigned)nt * 1024u, 0));
  };

  // Row transform as ONE fused multiply-add per value: T[b] = d[ra][b] + tsign * d[rb][b] with tsign = -1, +1, -1, -1 for rows
  // i = 0..3 of B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]; row 2 then carries the opposite sign (d1 - d2 instead of d2 - d1),
  // which the weight transform compensates by negating U[2][.] (wino_weights_kernel).
  const float tsign = wid == 1 ? 1.f : -1.f;
  float m1 = -1.f;
  asm volatile("" : "+v"(m1));      // opaque: x + m1 * y stays ONE packed fused multiply-add per register pair (a visible -1 becomes two unpacked subtractions)

  WSTAMP(0);
#ifdef RE2E_EXPERIMENTS
  // diagnostic (RE2E_WINO_DBG bits 4 / 8): the phases outside the matrix loop at a higher / lower issue priority than the loop -- do two co-resident
  // workgroups phase-lock because the one that streams MFMAs keeps the issue grant?
  if (p.dbg & 4) __builtin_amdgcn_s_setprio(3);
  if (p.dbg & 8) __builtin_amdgcn_s_setprio(0);
#endif
  Geo cur = geo_of(item);
  // ragged image batches (re2e_conv3x3_wino_rows): rows the caller never reads (beyond an utterance's end + the stack's reach) are left alone
  if (p.row_lim && cur.y0 >= p.row_lim[n]) return;
  if constexpr (LDSIN) {
    // half 0 is requested first, the first weights behind it; "at most the 8 weight loads outstanding" = half 0 has landed (vmcnt retires in order)
    set_item_lds(cur);
    __builtin_amdgcn_sched_barrier(0);
    stage(0, 0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      fetch_u(0, j);
      __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    stage(1, 1);                                                    // (C % 64 == 0: there is a half 1)
    __builtin_amdgcn_sched_barrier(0);
    read_raw(0, 0);
    __builtin_amdgcn_sched_barrier(0);
  } else {
    set_item(cur);
#ifdef RE2E_EXPERIMENTS
    if (coalesced_dbg) {
#pragma unroll
      for (int k = 0; k < 8; ++k) a_off[k] = (unsigned)(lane * 16 + k * 1024 + (cur.y0 * rowb + cur.x0 * colb));
    }
#endif
    // (same issue order as in the loop -- pixels first, then the weights position by position -- or the wait at the loop's head
    // has to cover the prologue's order as well and degenerates to vmcnt(0))
    __builtin_amdgcn_sched_barrier(0);
    fetch_raw(0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      fetch_u(0, j);
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  WSTAMP(1);
  {
    const int y0 = cur.y0, x0 = cur.x0, n0 = cur.nblk * WNT;
    f32x16 acc[4][2];
    // 64-channel layers on the LDS-staged path (their own instantiation: ONE pass of eight chunks): no clears -- the first MFMA of every accumulator
    // takes the constant 0 as C (128 vector moves fewer per work item of 256 MFMAs).  For the other channel counts the same needs more copies of the
    // unrolled pass in one function, and its register allocation spills.
    if constexpr (!C64) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[j][nt][r] = 0.f;
    }

    // The chunk loop has no branch -- the very last chunk of the workgroup re-fetches itself instead of fetching nothing -- so the
    // compiler's s_waitcnt counts stay exact: a wait only covers the loads it needs, never the ones issued a few instructions
    // earlier.  Measured and rejected, same GPU session each: a two-chunk-deep pipeline that transforms chunk k + 1's pixels between the third
    // and fourth position group of chunk k (241 registers, conv1_2 1.44 -> 1.53 ms, conv2_2 1.28 -> 1.32 / 1.42 ms); never clearing the
    // accumulators (first chunk as a second copy of the body whose first MFMA per tile takes the constant 0 as C: round 4 1.422 -> 1.445 ms, and
    // again in round 5 on top of the do-while form below, which removed the second set of 128 clears the compiler kept for a loop that may
    // run zero times: no change on any of the six shapes -- the ~ 250 vector instructions a work item lost this way do not bound the kernel).
    int chunk = 0;
#ifdef RE2E_EXPERIMENTS
    if (p.dbg & 4) __builtin_amdgcn_s_setprio(0);
    if (p.dbg & 8) __builtin_amdgcn_s_setprio(3);
#endif
    if constexpr (LDSIN) {
      // Eight chunks (two 32-channel halves: LDS buffers 0 and 1) per pass, unrolled, so that buffer and chunk position are instruction
      // immediates.  At the last chunk of a half every wavefront holds that chunk in registers: behind ONE workgroup barrier the half's buffer is free
      // for half h + 2, and the barrier also publishes half h + 1 (each wavefront has waited for its own requests: at most the 8 weight loads of the
      // chunk may still be in flight -- they are younger).  The LAST pass is a copy of the body without requests (there is no half to ask for, and an
      // out-of-range LDS-DMA request would write zeros into space the epilogue's exchange buffer is about to use): the load counts the compiler waits
      // with are exact in both copies.
      const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      auto pass_body = [&](auto only_tag, auto last_tag) {
        constexpr bool ONLY = decltype(only_tag)::value, LAST = decltype(last_tag)::value;     // ONLY: the single pass of a 64-channel layer
#pragma unroll
        for (int c8 = 0; c8 < 8; ++c8) {
          const int cc = c8 & 3, bf = c8 >> 2;
          f32x4 T[4], V[4];
#pragma unroll
          for (int b = 0; b < 4; ++b) T[b] = raw[b] + tsign * raw[4 + b];
          V[0] = T[0] + m1 * T[2];
          V[1] = T[1] + T[2];
          V[2] = T[2] + m1 * T[1];
          V[3] = T[1] + m1 * T[3];
          const int nxt = chunk + 1 < nch ? chunk + 1 : chunk;
          __builtin_amdgcn_sched_barrier(0);
          if (cc == 3) {
#ifdef RE2E_EXPERIMENTS
            if (!(p.dbg & 32))       // diagnostic (timing only, results may be wrong): no wait / barrier per half -- what do they cost?
#endif
            {
              asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
              __builtin_amdgcn_s_barrier();
              asm volatile("" ::: "memory");
            }
            if (!LAST) {
              stage((chunk >> 2) + 2, bf);                         // the half after next: it exists in every pass but the last
              __builtin_amdgcn_sched_barrier(0);
            }
          }
          if (!(LAST && c8 == 7)) read_raw(((c8 + 1) >> 2) & 1, (c8 + 1) & 3);     // next chunk's pixels out of LDS
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
#pragma unroll
              for (int nt = 0; nt < 2; ++nt)
                acc[j][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[j][jj], uf[j][nt][jj], (ONLY && c8 == 0 && jj == 0) ? zero16 : acc[j][nt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (!(LAST && c8 == 7)) {
              fetch_u(nxt, j);
              __builtin_amdgcn_sched_barrier(0);
            }
          }
          ++chunk;
        }
      };
      if constexpr (C64) {
        pass_body(std::true_type{}, std::true_type{});
      } else {
        for (int pass = 8; pass < nch; pass += 8) pass_body(std::false_type{}, std::false_type{});
        pass_body(std::false_type{}, std::true_type{});
      }
    } else {
      do {
        f32x4 T[4], V[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) T[b] = raw[b] + tsign * raw[4 + b];
        V[0] = T[0] + m1 * T[2];                        // subtractions as packed fused multiply-adds (there is no packed subtract)
        V[1] = T[1] + T[2];
        V[2] = T[2] + m1 * T[1];
        V[3] = T[1] + m1 * T[3];
        const int nxt = chunk + 1 < nch ? chunk + 1 : chunk;
        // (sched_barrier: the scheduler otherwise sinks every load to the end of the body, right in front of the wait that needs it)
        __builtin_amdgcn_sched_barrier(0);
        fetch_raw(nxt);                                 // the raw registers are free: next chunk's pixels fly under this chunk's MFMAs
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
          for (int jj = 0; jj < 4; ++jj)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
              acc[j][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[j][jj], uf[j][nt][jj], acc[j][nt], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          fetch_u(nxt, j);                              // ... and position j's weights of the next chunk behind its last use
          __builtin_amdgcn_sched_barrier(0);
        }
      } while (++chunk < nch);
    }

    WSTAMP(2);
#ifdef RE2E_EXPERIMENTS
    if (p.dbg & 4) __builtin_amdgcn_s_setprio(3);
    if (p.dbg & 8) __builtin_amdgcn_s_setprio(0);
#endif
    // ---- output transform, column half, in registers: R[b] = sum_j M[i][j] A[j][b],  A^T = [1 1 1 0; 0 1 -1 -1]
    float* Rs = smem + (wid * 2) * (32 * LDR);
    // (two accumulator registers per packed instruction: 64 vector instructions instead of 128 -- outside its matrix block a wavefront gets about
    // one issue slot per MFMA of its SIMD partner, so the count is what this phase costs)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const f32x2 a0 = {acc[0][nt][r], acc[0][nt][r + 1]}, a1 = {acc[1][nt][r], acc[1][nt][r + 1]};
        const f32x2 a2 = {acc[2][nt][r], acc[2][nt][r + 1]}, a3 = {acc[3][nt][r], acc[3][nt][r + 1]};
        const f32x2 sm = (a0 + a1) + a2, df = (a1 + m1 * a2) + m1 * a3;
        const int m = (r & 3) + 8 * (r >> 2) + 4 * lh;                  // tile of this wavefront's 32 (r + 1: the next one)
        Rs[m * LDR + nt * 32 + lr] = sm[0];
        Rs[(m + 1) * LDR + nt * 32 + lr] = sm[1];
        Rs[32 * LDR + m * LDR + nt * 32 + lr] = df[0];
        Rs[32 * LDR + (m + 1) * LDR + nt * 32 + lr] = df[1];
      }
    // Behind the LDS writes the accumulators are dead: the epilogue's own loads -- the bias of this item's channels and, for a data
    // gradient, the ReLU mask of its output pixels -- are issued HERE, in front of the barrier, so that their latency hides behind
    // the wait for the slowest wavefront and the LDS reads (fetched where they are used they would expose it whole: vmcnt counts
    // in order and they would be the youngest loads).  Output offsets are 32-bit; a pixel outside the image gets the out-of-range
    // offset: its mask reads 0, its store is dropped.
    const int c4 = (lane & 15) * 4;
    const f32x4 bv = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB, (unsigned)((n0 + c4) * 4), 0, 0));
    const int PH2 = (p.H + 1) >> 1, PW2 = (p.W + 1) >> 1;
    unsigned o_off[2][4];
    f32x4 mk[2][4];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int tl = 8 * wid + 4 * it + (lane >> 4);
      const int ty2 = tl / TXW, tx2 = tl - ty2 * TXW;
      const int oy = y0 + 2 * ty2, ox = x0 + 2 * tx2;
      if (p.pool_out) {
        o_off[it][0] = (oy < p.H && ox < p.W) ? (unsigned)((((n * PH2 + (oy >> 1)) * PW2 + (ox >> 1)) * p.Cout + n0 + c4) * 4) : WOOB;
      } else {
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          const int y = oy + (d >> 1), x = ox + (d & 1);
          o_off[it][d] = (y < p.H && x < p.W) ? (unsigned)((((n * p.H + y) * p.W + x) * p.Cout + n0 + c4) * 4) : WOOB;
          if (p.mask) mk[it][d] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsM, o_off[it][d], 0, 0));
        }
      }
    }
    WSTAMP(3);
    __syncthreads();
    WSTAMP(4);
    // ---- row half + bias / ReLU / mask / pool + store: wavefront w takes tiles 8w .. 8w+7; lane = (tile of 4, 4 channels), 2 passes;
    // a store instruction writes four pixels' 64 channels (4 x 256 contiguous bytes)
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int tl = 8 * wid + 4 * it + (lane >> 4);
      f32x4 R[4][2];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int b = 0; b < 2; ++b)
          R[i][b] = *reinterpret_cast<const f32x4*>(smem + ((i * 2 + b) * 32 + tl) * LDR + c4);
      f32x4 Y[2][2];
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const f32x4 s12 = R[1][b] + R[2][b], d12 = R[1][b] + m1 * R[2][b];
        Y[0][b] = R[0][b] + s12 + bv;
        Y[1][b] = d12 + m1 * R[3][b] + bv;
      }
      if (p.pool_out) {
        // a tile is one 2x2 / stride-2 pooling window (y0, x0 even).  max and ReLU commute: the ReLU is applied to the pooled value.
        // Index byte: first maximum in row-major order among the pixels inside the image (ceil mode), 4 when the maximum is <= 0
        // (re2e_maxpool2_fwd with relu_in)
        const int ty2 = tl / TXW, tx2 = tl - ty2 * TXW;
        const int oy = y0 + 2 * ty2, ox = x0 + 2 * tx2;
        f32x4 best = Y[0][0];
        int bi[4] = {0, 0, 0, 0};
#pragma unroll
        for (int d = 1; d < 4; ++d) {
          if (oy + (d >> 1) < p.H && ox + (d & 1) < p.W) {
            const f32x4 v = Y[d >> 1][d & 1];
#pragma unroll
            for (int k = 0; k < 4; ++k)
              if (v[k] > best[k]) { best[k] = v[k]; bi[k] = d; }
          }
        }
        unsigned b4 = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) { b4 |= (unsigned)(best[k] > 0.f ? bi[k] : 4) << (8 * k); best[k] = best[k] > 0.f ? best[k] : 0.f; }
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, best), rsO, o_off[it][0], 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(b4, rsI, o_off[it][0] == WOOB ? WOOB : o_off[it][0] >> 2, 0, 0);
      } else {
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          f32x4 v = Y[d >> 1][d & 1];
          if (p.relu) {
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = v[k] < 0.f ? 0.f : v[k];       // torch.relu: NaN stays NaN
          }
          if (p.mask) {
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = mk[it][d][k] > 0.f ? v[k] : 0.f;
          }
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsO, o_off[it][d], 0, 0);
        }
      }
    }
    WSTAMP(5);
    WSTAMP(6);
  }
}


#ifdef RE2E_EXPERIMENTS
#include "experiments/wino_pipe.hip"      // the rejected pipelined form (RE2E_WINO_PIPE=1): experiments build only
#endif

template <int TXW, bool LDSIN, bool C64 = false>
void launch_wino(const WinoArgs& a, hipStream_t st) {
  size_t lds = (size_t)8 * 32 * LDR * sizeof(float);      // the exchange buffer (LDSIN: aliases the two staged halves, 2 x 24 KB)
  static const char* lds_env = exp_env("RE2E_WINO_LDS_KB");       // experiments: a larger request = one workgroup per CU (one wavefront per SIMD)
  if (lds_env && (size_t)atoi(lds_env) * 1024 > lds) lds = (size_t)atoi(lds_env) * 1024;
  static LdsLimit lim;
  lim.ensure(reinterpret_cast<const void*>(&wino_conv3x3_kernel<TXW, LDSIN, C64>), lds);
  hipLaunchKernelGGL((wino_conv3x3_kernel<TXW, LDSIN, C64>), dim3((unsigned)a.per_image, (unsigned)a.NI), dim3(256), lds, st, a);
}

}  // namespace

extern "C" size_t re2e_conv3x3_wino_workspace_bytes(int C, int Cout) {
  return C > 0 && Cout > 0 ? (size_t)16 * C * Cout * sizeof(float) : 0;
}

static int wino_impl(const float* in, int NI, int H, int W, int C, const float* w, int Cout, int dgrad, const float* bias, int relu,
                     const float* mask, float* out, float* pool_out, unsigned char* pool_idx, const int* row_lim, void* workspace,
                     size_t workspace_bytes, hipStream_t stream) {
  RE2E_CHECK_ARG(in && w && workspace && (out || pool_out), "null operand");
  RE2E_CHECK_ARG(NI > 0 && H > 0 && W > 0 && C > 0 && Cout > 0, "bad geometry");
  RE2E_CHECK_ARG(!pool_out || (pool_idx && relu && !mask && !dgrad), "pool_out needs pool_idx, relu = 1, no mask, forward direction");
  if (C % WCK || Cout % WNT) { re2e_set_error("re2e_conv3x3_wino: C must be a multiple of 8 and Cout of 64 (got %d, %d)", C, Cout); return RE2E_EUNSUPPORTED; }
  const long in_bytes = (long)NI * H * W * C * 4, out_bytes = (long)NI * H * W * Cout * 4, u_bytes = (long)16 * C * Cout * 4;
  if (in_bytes >= 0x7FFFFF00L || out_bytes >= 0x7FFFFF00L || u_bytes >= 0x7FFFFF00L) {
    re2e_set_error("re2e_conv3x3_wino: tensors must be < 2 GiB");
    return RE2E_EUNSUPPORTED;
  }
  const uintptr_t al = reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(mask) |
                       reinterpret_cast<uintptr_t>(pool_out) | reinterpret_cast<uintptr_t>(bias) | reinterpret_cast<uintptr_t>(workspace);
  if ((al & 15) || (reinterpret_cast<uintptr_t>(pool_idx) & 3)) { re2e_set_error("re2e_conv3x3_wino: operands must be 16-byte aligned"); return RE2E_EUNSUPPORTED; }
  RE2E_CHECK_ARG(workspace_bytes >= (size_t)u_bytes, "workspace too small");
  if ((Cout / WNT) & (Cout / WNT - 1)) { re2e_set_error("re2e_conv3x3_wino: Cout / 64 must be a power of two (got Cout = %d)", Cout); return RE2E_EUNSUPPORTED; }
  float* uf = (float*)workspace;
  const long total = (long)16 * C * Cout;
  hipLaunchKernelGGL(wino_weights_kernel, dim3((unsigned)(cdiv(total, 256) > 2048 ? 2048 : cdiv(total, 256))), dim3(256), 0, stream, w, Cout, C, dgrad, uf);
  WinoArgs a;
  a.in = in; a.ufrag = uf; a.out = out; a.bias = bias; a.mask = mask; a.pool_out = pool_out; a.pool_idx = pool_idx;
  a.NI = NI; a.H = H; a.W = W; a.C = C; a.Cout = Cout; a.relu = relu;
  a.row_lim = row_lim;
  const int ngn = Cout / WNT;
  if (ngn & (ngn - 1)) { re2e_set_error("re2e_conv3x3_wino: Cout / 64 must be a power of two (got Cout = %d)", Cout); return RE2E_EUNSUPPORTED; }
  a.ngn_shift = 0;
  while ((1 << a.ngn_shift) < ngn) ++a.ngn_shift;
  a.in_bytes = (unsigned)in_bytes; a.u_bytes = (unsigned)u_bytes; a.out_bytes = (unsigned)out_bytes;
  a.pool_bytes = (unsigned)((long)NI * ((H + 1) / 2) * ((W + 1) / 2) * Cout * 4);
  // patch shape: 16 x 8 pixels (8 x 4 tiles) unless 8 x 16 wastes fewer padded pixels (W = 40: 8-wide patches fit exactly)
  const long pad_w = (long)cdiv(H, 8) * 8 * cdiv(W, 16) * 16, pad_n = (long)cdiv(H, 16) * 16 * cdiv(W, 8) * 8;
  const bool wide = pad_w <= pad_n;
  const int PWp = wide ? 16 : 8, PHp = wide ? 8 : 16;
  a.tiles_x = cdiv(W, PWp); a.tiles_y = cdiv(H, PHp);
  const long per_image = (long)a.tiles_x * a.tiles_y * ngn;
  if (per_image >= 0x7FFFFFF0L || NI > 65535) { re2e_set_error("re2e_conv3x3_wino: too many work items"); return RE2E_EUNSUPPORTED; }
  a.per_image = (int)per_image;
  static const int dbg_env = exp_env("RE2E_WINO_DBG") ? atoi(exp_env("RE2E_WINO_DBG")) : 0;
  a.dbg = dbg_env;
  static const char* st_env = exp_env("RE2E_WINO_STAMPS");      // device address (hex) of a 4096 x 4 x 8 x 8-byte buffer
  a.stamps = st_env ? (unsigned long long*)strtoull(st_env, nullptr, 16) : nullptr;
  static const bool log_calls = getenv("RE2E_IGEMM_LOG") != nullptr;   // tools/igemm_table.py joins this with a kernel trace
  if (log_calls)
    fprintf(stderr, "[igemm] A=Wino%s B=DenseK tile=128x64x8 vec=1 M=%ld N=%d K=%d splits=1\n", dgrad ? "D" : "F", (long)NI * H * W, Cout, 9 * C);
#ifdef RE2E_EXPERIMENTS
  if (wino_pipe_try(a, per_image, NI, C, wide, stream)) { RE2E_LAUNCH_CHECK(); return RE2E_OK; }
#endif
  // the input patch through LDS (round 6) wherever the channel count allows it (whole pairs of 32-channel halves); RE2E_WINO_LDSIN=0 (experiments
  // build): the per-lane loads of rounds 3-5
  static const bool ldsin_env = !(exp_env("RE2E_WINO_LDSIN") && atoi(exp_env("RE2E_WINO_LDSIN")) == 0);
  const bool ldsin = ldsin_env && C % 64 == 0 && !a.stamps && !(a.dbg & ~(1 | 32 | 64 | 128));
  if (ldsin && C == 64) { if (wide) launch_wino<8, true, true>(a, stream); else launch_wino<4, true, true>(a, stream); }
  else if (ldsin) { if (wide) launch_wino<8, true>(a, stream); else launch_wino<4, true>(a, stream); }
  else { if (wide) launch_wino<8, false>(a, stream); else launch_wino<4, false>(a, stream); }
  RE2E_LAUNCH_CHECK();
  return RE2E_OK;
}

extern "C" int re2e_conv3x3_wino(const float* in, int NI, int H, int W, int C, const float* w, int Cout, int dgrad, const float* bias, int relu,
                                 const float* mask, float* out, float* pool_out, unsigned char* pool_idx, void* workspace,
                                 size_t workspace_bytes, hipStream_t stream) {
  return wino_impl(in, NI, H, W, C, w, Cout, dgrad, bias, relu, mask, out, pool_out, pool_idx, nullptr, workspace, workspace_bytes, stream);
}

// The same over a RAGGED image batch: output rows y >= row_lim[n] of image n (full-resolution rows, also with the fused pool) are not computed
// and not written -- patches that start at or beyond the limit exit at once, a patch that straddles it is computed whole.  The VGG front end
// convolves zero-padded utterances and cuts every one at its pooled length afterwards (e2e_encoder.py:259-278): what it computes more than
// the stack's reach beyond an utterance's end is never read.  The caller owns the limits' consistency (VGG2L.conv_stack) and puts zeros where a
// consumer reads all rows (re2e_fill_image_rows).
extern "C" int re2e_conv3x3_wino_rows(const float* in, int NI, int H, int W, int C, const float* w, int Cout, int dgrad, const float* bias, int relu,
                                      const float* mask, float* out, float* pool_out, unsigned char* pool_idx, const int* row_lim, void* workspace,
                                      size_t workspace_bytes, hipStream_t stream) {
  RE2E_CHECK_ARG(row_lim, "null row limits");
  return wino_impl(in, NI, H, W, C, w, Cout, dgrad, bias, relu, mask, out, pool_out, pool_idx, row_lim, workspace, workspace_bytes, stream);
}

namespace {
__global__ void fill_image_rows_kernel(float* __restrict__ t, int H, long row_f4, const int* __restrict__ lim, int div, float value) {
  const int n = blockIdx.y;
  int r0 = (lim[n] + div - 1) / div;
  if (r0 >= H) return;
  const long tot = (long)(H - r0) * row_f4;
  f32x4* base = reinterpret_cast<f32x4*>(t) + ((long)n * H + r0) * row_f4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += (long)gridDim.x * blockDim.x) base[i] = f32x4{value, value, value, value};
}
}  // namespace

// t: (N, H, row_floats) images; rows r >= ceil(lim[n] / div) of image n = value (div = 2: t is the 2x2-pooled tensor of a layer whose limits
// count full-resolution rows).  max_tail_rows: an upper bound of H - lim / div over the batch (the host knows the lengths): sizes the launch.
extern "C" int re2e_fill_image_rows(float* t, int N, int H, long row_floats, const int* lim, int div, int max_tail_rows, float value, hipStream_t stream) {
  RE2E_CHECK_ARG(t && lim && N > 0 && H > 0 && row_floats > 0 && (div == 1 || div == 2), "bad argument");
  RE2E_CHECK_ARG(row_floats % 4 == 0 && (reinterpret_cast<uintptr_t>(t) & 15) == 0, "rows must be whole float4s");
  if (max_tail_rows <= 0) return RE2E_OK;
  long blocks = ((long)max_tail_rows * (row_floats / 4) + 1023) / 1024;       // 4 float4 per thread
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(fill_image_rows_kernel, dim3((unsigned)blocks, (unsigned)N), dim3(256), 0, stream, t, H, row_floats / 4, lim, div, value);
  RE2E_LAUNCH_CHECK();
  return RE2E_OK;
}

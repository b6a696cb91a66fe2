// EXPERIMENTS BUILD ONLY (make EXPERIMENTS=1; textually included by ../winograd.hip inside its anonymous namespace): the pipelined form of
// the fused Winograd F(2x2,3x3) kernel (round 5, RE2E_WINO_PIPE=1) -- built, measured, not kept (profiles/r05_wino_pipelined.txt).  The shipped
// library does not contain it.
// ---- The pipelined form (round 5, EXPERIMENTS BUILD ONLY: RE2E_WINO_PIPE=1 -- built, measured, not kept).
// Two workgroups per CU walk IPW consecutive work items each; the last chunk of item k fetches the first pixels and weights of item k + 1 (the
// geometry advances by scalar increments: no division after the first item), the first MFMA of every accumulator tile takes the constant 0 as its
// C operand (nothing is cleared), the output stage of item k runs at the head of item k + 1 where its accumulators do not exist yet (128 free
// registers), the first item's stage processes a dummy whose stores the descriptor's range check drops (one straight-line body, exact s_waitcnt
// counts), the epilogue variant (plain / pool / mask) is a template parameter.  Same arithmetic in the same order: bitwise-equal results.
// What it was built to remove -- workgroup start, address set-up, the latency of the first loads, the accumulator clears: 8 000 cycles per
// 16 400 of matrix work when a wavefront is alone on its SIMD (s_memtime stamps) -- it removes; what it gets is the same rate
// (profiles/r05_wino_pipelined.txt: conv1_2 +3 %, conv2_1 +5 %, the 128-channel layers -1 %, the step +0.45 ms), because of what the
// two probes written for it show (tools/micro/mfma_coissue.hip, mfma_partner.hip; profiles/r05_mfma_coissue.txt, r05_mfma_partner.txt):
//   * fp32 MFMAs and vector-ALU instructions of a SIMD do NOT overlap, from one wavefront or from two: every v_* instruction (packed or not, a
//     v_mov or a v_cndmask as well) takes 4-8 cycles out of the matrix stream wherever it is placed; only scalar, LDS and memory instructions
//     are free.  A work item here is 256 MFMAs x 64 cycles + ~ 520-870 vector instructions x ~ 5 = 19 000-20 700 cycles: 0.79-0.86 is the bound
//     of ANY arrangement of this arithmetic, before a single stall;
//   * a wavefront that issues MFMA after MFMA keeps the SIMD's issue grant: its partner issues NOTHING (not even scalar instructions) until the
//     stream has a gap or issues a vector-ALU instruction -- so two resident wavefronts take turns rather than overlap, and the second workgroup
//     per CU hides stalls but adds no issue capacity;
//   * s_memtime counts shader cycles: a lone MFMA stream reads exactly 64.0 per MFMA, one wavefront per SIMD or two; the wall clock of the probe's
//     first launches (31 ns per MFMA, 2.05 GHz) is the clock ramping up, sustained it is 27.2 ns = 2.35 GHz (the bare loop's 154-155 TFLOP/s).
// Measured per item and SIMD: 23 400 cycles (this form, two workgroups per CU), 24 900 (this form, one), 27 200 (the kernel above).
enum { EPI_PLAIN = 0, EPI_POOL = 1, EPI_MASK = 2 };

template <int TXW, int NCH, int EPI>
__global__ __launch_bounds__(256, 2) void wino_pipe_kernel(WinoArgs p) {
  constexpr int TYH = 32 / TXW;
  constexpr int PW = 2 * TXW, PH = 2 * TYH;
  extern __shared__ __attribute__((aligned(16))) float smem[];     // [4 rows i][2 b][32 tiles][LDR]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  // ---- this workgroup's run of work items: XCD x (= blockIdx.x & 7, the hardware's round robin) owns a contiguous eighth of the launch's
  // items (its L2 sees neighbouring patches and one set of images), workgroup l of it the items [l * ipw, (l + 1) * ipw) of that eighth
  const int xcd = blockIdx.x & 7, wl = blockIdx.x >> 3;
  const int q8 = p.total >> 3, r8 = p.total & 7;
  const int xs = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8, xc = q8 + (xcd < r8 ? 1 : 0);
  const int f0 = xs + wl * p.ipw;
  const int f1 = f0 + p.ipw < xs + xc ? f0 + p.ipw : xs + xc;
  if (f0 >= f1) return;

  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0, p.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsU = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.ufrag), 0, p.u_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc(EPI == EPI_POOL ? p.pool_out : p.out, 0, EPI == EPI_POOL ? p.pool_bytes : p.out_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsM = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(EPI == EPI_MASK ? p.mask : p.in), 0, EPI == EPI_MASK ? p.out_bytes : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsI = __builtin_amdgcn_make_buffer_rsrc(EPI == EPI_POOL ? (void*)p.pool_idx : (void*)p.in, 0, EPI == EPI_POOL ? p.pool_bytes / 4 : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.bias ? p.bias : p.in), 0, p.bias ? (unsigned)p.Cout * 4u : 0u, 0x00020000);

  const int ra = wid == 0 ? 0 : 1, rb = wid == 3 ? 3 : 2;
  const int rowb = p.W * p.C * 4, colb = p.C * 4;
  const int img_bytes = p.H * rowb;
  const unsigned u_lane = (unsigned)lane * 16u;
  // (what depends on the lane only and is needed once per item is RE-derived from the lane id where it is used -- the opaque copy keeps the compiler from
  // holding a dozen such values in registers across the matrix loop, where 128 + 80 + 16 are taken)
  auto lane_now = [&]() { int l = lane; asm volatile("" : "+v"(l)); return l; };
  const int ngn = 1 << p.ngn_shift;

  struct Geo { int n, nblk, ty, tx; };            // image, 64-channel group, patch row / column: all wave-uniform
  Geo cur;
  {
    const int n = f0 / p.per_image, local = f0 - n * p.per_image;
    const int tile_id = local >> p.ngn_shift;
    const int ty = tile_id / p.tiles_x;
    cur.n = __builtin_amdgcn_readfirstlane(n);
    cur.nblk = local & (ngn - 1);
    cur.ty = __builtin_amdgcn_readfirstlane(ty);
    cur.tx = tile_id - cur.ty * p.tiles_x;
  }
  auto advance = [&](Geo g) {
    if (++g.nblk == ngn) {
      g.nblk = 0;
      if (++g.tx == p.tiles_x) {
        g.tx = 0;
        if (++g.ty == p.tiles_y) { g.ty = 0; ++g.n; }
      }
    }
    return g;
  };

  unsigned a_off[8];
  unsigned a_s, u_s0;
  auto set_item = [&](const Geo& g) {
    const int y0 = g.ty * PH, x0 = g.tx * PW;
    a_s = (unsigned)(g.n * img_bytes);
    const int ln = lane_now(), lr_ = ln & 31, lh_ = ln >> 5;
    const int tyi = lr_ / TXW, txi = lr_ - tyi * TXW;
    const int base = (2 * tyi - 1) * rowb + (2 * txi - 1) * colb + 16 * lh_ + y0 * rowb + x0 * colb;
    // (no interior / border branch: the body stays one basic block)
    const int iy0 = y0 - 1 + 2 * tyi, ix0 = x0 - 1 + 2 * txi;
    const bool rok0 = (unsigned)(iy0 + ra) < (unsigned)p.H, rok1 = (unsigned)(iy0 + rb) < (unsigned)p.H;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const bool ok = (k < 4 ? rok0 : rok1) & ((unsigned)(ix0 + (k & 3)) < (unsigned)p.W);
      a_off[k] = ok ? (unsigned)(base + (k < 4 ? ra : rb) * rowb + (k & 3) * colb) : WOOB;
    }
  };

  f32x4 raw[8], uf[4][2];
  auto fetch_raw = [&](int chunk) {
    unsigned cb = a_s + (unsigned)(chunk * WCK * 4);
#pragma unroll
    for (int k = 0; k < 8; ++k) raw[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, a_off[k], cb, 0));
  };
  auto fetch_u = [&](int chunk, int j) {
    unsigned s = u_s0 + (unsigned)(chunk * 16 * 2 + j * 2) * 1024u;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) uf[j][nt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsU, u_lane, s + (unsigned)nt * 1024u, 0));
  };

  const float tsign = wid == 1 ? 1.f : -1.f;
  float m1 = -1.f;
  asm volatile("" : "+v"(m1));
  const float relu_lo = p.relu ? 0.f : -__builtin_inff();
  const int PH2 = (p.H + 1) >> 1, PW2 = (p.W + 1) >> 1;

  // ---- the output stage of the PREVIOUS item in four pieces (prev_ok = 0: the dummy in front of the first item -- every offset out of range).
  // Piece (it, b): 4 tiles x 16 lanes of this wavefront's 8 tiles (it), output column b of their 2x2 blocks: the four R rows of that column come
  // from LDS, become the two pixels (0, b) and (1, b), and -- plain / mask -- leave at once; the pool variant keeps column 0's two values for the
  // piece of column 1, which owns the window.  16 + 8 registers in flight per piece, 8 carried to the next.
  int pv_n = 0, pv_n0 = 0, pv_y0 = 0, pv_x0 = 0, prev_ok = 0;
  f32x4 bv, keep[2], mkall[2][2][2];
  auto out_off = [&](int it, int a, int b) {
    const int ln = lane_now(), c4 = (ln & 15) * 4;
    const int tl = 8 * wid + 4 * it + (ln >> 4);
    const int ty2 = tl / TXW, tx2 = tl - ty2 * TXW;
    const int y = pv_y0 + 2 * ty2 + a, x = pv_x0 + 2 * tx2 + b;
    return (prev_ok && y < p.H && x < p.W) ? (unsigned)((((pv_n * p.H + y) * p.W + x) * p.Cout + pv_n0 + c4) * 4) : WOOB;
  };
  auto stage_requests = [&]() {           // the stage's own global loads, all at once: the bias of the item's channels, the ReLU mask of its pixels
    const int c4 = (lane_now() & 15) * 4;
    bv = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB, (unsigned)((pv_n0 + c4) * 4), 0, 0));
    if (EPI == EPI_MASK) {
#pragma unroll
      for (int it = 0; it < 2; ++it)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b)
            mkall[it][a][b] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsM, out_off(it, a, b), 0, 0));
    }
  };
  auto piece = [&](int it, int b) {
    const int ln = lane_now(), c4 = (ln & 15) * 4;
    const int tl = 8 * wid + 4 * it + (ln >> 4);
    const int ty2 = tl / TXW, tx2 = tl - ty2 * TXW;
    const int oy = pv_y0 + 2 * ty2, ox = pv_x0 + 2 * tx2;
    f32x4 R[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) R[i] = *reinterpret_cast<const f32x4*>(smem + ((i * 2 + b) * 32 + tl) * LDR + c4);
    const f32x4 s12 = R[1] + R[2], d12 = R[1] + m1 * R[2];
    f32x4 Y[2];
    Y[0] = R[0] + s12 + bv;
    Y[1] = d12 + m1 * R[3] + bv;
    if (EPI == EPI_POOL) {
      if (b == 0) {
        keep[0] = Y[0]; keep[1] = Y[1];
      } else {
        // a tile is one 2x2 / stride-2 pooling window (y0, x0 even).  max and ReLU commute: the ReLU is applied to the pooled value.  Index byte: first
        // maximum in row-major order among the pixels inside the image (ceil mode), 4 when the maximum is <= 0 (re2e_maxpool2_fwd with relu_in)
        const unsigned o = (prev_ok && oy < p.H && ox < p.W) ? (unsigned)((((pv_n * PH2 + (oy >> 1)) * PW2 + (ox >> 1)) * p.Cout + pv_n0 + c4) * 4) : WOOB;
        f32x4 best = keep[0];
        int bi[4] = {0, 0, 0, 0};
#pragma unroll
        for (int d = 1; d < 4; ++d) {
          const bool in = (oy + (d >> 1) < p.H) & (ox + (d & 1) < p.W);
          const f32x4 v = (d & 1) ? Y[d >> 1] : keep[d >> 1];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const bool take = in & (v[k] > best[k]);
            best[k] = take ? v[k] : best[k];
            bi[k] = take ? d : bi[k];
          }
        }
        unsigned b4 = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) { b4 |= (unsigned)(best[k] > 0.f ? bi[k] : 4) << (8 * k); best[k] = best[k] > 0.f ? best[k] : 0.f; }
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, best), rsO, o, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(b4, rsI, o == WOOB ? WOOB : o >> 2, 0, 0);
      }
    } else {
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        f32x4 v = Y[a];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = v[k] < relu_lo ? relu_lo : v[k];       // torch.relu: NaN stays NaN (relu_lo = -inf: no ReLU)
        if (EPI == EPI_MASK) {
#pragma unroll
          for (int k = 0; k < 4; ++k) v[k] = mkall[it][a][b][k] > 0.f ? v[k] : 0.f;
        }
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsO, out_off(it, a, b), 0, 0);
      }
    }
  };
  // The stage runs at the head of the FOLLOWING item's first chunk: its accumulators do not exist yet (the first MFMA of every tile takes the constant
  // 0), so 128 registers are free -- anywhere later the stage's 40-70 registers do not fit beside 128 + 80 of accumulators and prefetch.
  auto deferred = [&](int chunk) {
    if (chunk == 0) {
      stage_requests();
      __builtin_amdgcn_s_barrier();
      piece(0, 0);
      piece(0, 1);
      piece(1, 0);
      piece(1, 1);
    }
  };

  set_item(cur);
  u_s0 = (unsigned)(((cur.nblk * NCH) * 16 + 4 * wid) * 2) * 1024u;
  __builtin_amdgcn_sched_barrier(0);
  fetch_raw(0);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    fetch_u(0, j);
    __builtin_amdgcn_sched_barrier(0);
  }

  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int f = f0; f < f1; ++f) {
    const Geo nxt = f + 1 < f1 ? advance(cur) : cur;            // (the last item re-fetches its own first chunk: no branch in the body)
    const unsigned u_cur = u_s0;
    f32x16 acc[4][2];
#pragma unroll
    for (int chunk = 0; chunk < NCH; ++chunk) {
      f32x4 T[4], V[4];
#pragma unroll
      for (int b = 0; b < 4; ++b) T[b] = raw[b] + tsign * raw[4 + b];
      V[0] = T[0] + m1 * T[2];
      V[1] = T[1] + T[2];
      V[2] = T[2] + m1 * T[1];
      V[3] = T[1] + m1 * T[3];
      __builtin_amdgcn_sched_barrier(0);
      // a piece of the previous item's output stage, while the pixel registers are free (vector instructions are never hidden behind fp32 MFMAs --
      // tools/micro/mfma_coissue.hip: every one costs the matrix pipe 3-7 cycles wherever it sits -- so where a piece goes is a matter of
      // registers only)
      deferred(chunk);
      __builtin_amdgcn_sched_barrier(0);
      if (chunk == NCH - 1) set_item(nxt);
      __builtin_amdgcn_sched_barrier(0);
      fetch_raw(chunk == NCH - 1 ? 0 : chunk + 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt)
            acc[j][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[j][jj], uf[j][nt][jj], chunk == 0 && jj == 0 ? zero16 : acc[j][nt], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        u_s0 = chunk == NCH - 1 ? (unsigned)(((nxt.nblk * NCH) * 16 + 4 * wid) * 2) * 1024u : u_cur;
        fetch_u(chunk == NCH - 1 ? 0 : chunk + 1, j);           // position j's weights of the next chunk (of the next item) behind their last use
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    u_s0 = (unsigned)(((nxt.nblk * NCH) * 16 + 4 * wid) * 2) * 1024u;
    // ---- column half of the output transform on this item's accumulators -> LDS (every wavefront has read the previous item's R values long
    // ago -- slots 2 and 6 -- the barrier only makes that a guarantee)
    __builtin_amdgcn_s_barrier();
    float* Rs = smem + (wid * 2) * (32 * LDR);
    const int lnw = lane_now(), lr = lnw & 31, lh = lnw >> 5;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const f32x2 a0 = {acc[0][nt][r], acc[0][nt][r + 1]}, a1 = {acc[1][nt][r], acc[1][nt][r + 1]};
        const f32x2 a2 = {acc[2][nt][r], acc[2][nt][r + 1]}, a3 = {acc[3][nt][r], acc[3][nt][r + 1]};
        const f32x2 sm = (a0 + a1) + a2, df = (a1 + m1 * a2) + m1 * a3;
        const int m = (r & 3) + 8 * (r >> 2) + 4 * lh;
        Rs[m * LDR + nt * 32 + lr] = sm[0];
        Rs[(m + 1) * LDR + nt * 32 + lr] = sm[1];
        Rs[32 * LDR + m * LDR + nt * 32 + lr] = df[0];
        Rs[32 * LDR + (m + 1) * LDR + nt * 32 + lr] = df[1];
      }
    pv_n = cur.n; pv_n0 = cur.nblk * WNT; pv_y0 = cur.ty * PH; pv_x0 = cur.tx * PW; prev_ok = 1;
    cur = nxt;
    __builtin_amdgcn_sched_barrier(0);
  }
  // ---- drain: the last item's output stage
  stage_requests();
  __syncthreads();
  piece(0, 0);
  piece(0, 1);
  piece(1, 0);
  piece(1, 1);
}

inline int device_cus() {
  static const int n = [] { int dev = 0; hipDeviceProp_t pr; return (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) ? pr.multiProcessorCount : 256; }();
  return n;
}

template <int TXW, int NCH, int EPI>
void launch_wino_pipe(const WinoArgs& a, hipStream_t st) {
  constexpr size_t lds = (size_t)8 * 32 * LDR * sizeof(float);
  static LdsLimit lim;
  lim.ensure(reinterpret_cast<const void*>(&wino_pipe_kernel<TXW, NCH, EPI>), lds);
  const int per_xcd = (a.total + 7) / 8;
  hipLaunchKernelGGL((wino_pipe_kernel<TXW, NCH, EPI>), dim3((unsigned)(8 * cdiv(per_xcd, a.ipw))), dim3(256), lds, st, a);
}

template <int TXW, int NCH>
void launch_wino_pipe_epi(const WinoArgs& a, hipStream_t st) {
  if (a.pool_out) launch_wino_pipe<TXW, NCH, EPI_POOL>(a, st);
  else if (a.mask) launch_wino_pipe<TXW, NCH, EPI_MASK>(a, st);
  else launch_wino_pipe<TXW, NCH, EPI_PLAIN>(a, st);
}


// RE2E_WINO_PIPE=1: the pipelined form (channel counts 64 and 128); consecutive items per workgroup: about 14, rounded so that an XCD's workgroups
// fill whole rounds of its CUs (RE2E_WINO_IPW overrides).  Returns false when this launch is not its case.
inline bool wino_pipe_try(WinoArgs& a, long per_image, int NI, int C, bool wide, hipStream_t stream) {
  static const char* pipe_env = exp_env("RE2E_WINO_PIPE");
  static const char* ipw_env = exp_env("RE2E_WINO_IPW");
  const long items = per_image * NI;
  if (!((C == 64 || C == 128) && items < 0x7FFFFFF0L && pipe_env && atoi(pipe_env) == 1 && !a.stamps)) return false;
  a.total = (int)items;
  const int per_xcd = cdiv(items, 8), slots_xcd = 2 * (device_cus() / 8 > 0 ? device_cus() / 8 : 32);     // two workgroups per CU
  int rounds = (per_xcd + slots_xcd * 7) / (slots_xcd * 14);
  if (rounds < 1) rounds = 1;
  a.ipw = cdiv(per_xcd, (long)slots_xcd * rounds);
  if (ipw_env && atoi(ipw_env) > 0) a.ipw = atoi(ipw_env);
  if (C == 64) { if (wide) launch_wino_pipe_epi<8, 8>(a, stream); else launch_wino_pipe_epi<4, 8>(a, stream); }
  else { if (wide) launch_wino_pipe_epi<8, 16>(a, stream); else launch_wino_pipe_epi<4, 16>(a, stream); }
  return true;
}

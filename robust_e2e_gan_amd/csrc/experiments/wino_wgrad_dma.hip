// EXPERIMENTS BUILD ONLY (make EXPERIMENTS=1; textually included by ../wino_wgrad.hip inside its anonymous namespace).  The shipped library
// does not contain it.
// ---- EXPERIMENTS BUILD ONLY (RE2E_WW_DMA=1): built and measured in round 5, not kept (profiles/r05_wino_wgrad_dma.txt) ----------------------
// One workgroup per CU, the patches travelling global -> LDS by `buffer_load_dwordx4 ... lds` into TWO 64 KB stages (no staging registers, no
// ds_write pass, one barrier per patch), the k-steps as a three-deep pipeline held in place by scheduling barriers (LDS reads of step s + 2,
// transforms of step s + 1, MFMAs of step s) and the transforms as PACKED instructions written in inline asm.  Bitwise-equal sums; same speed as
// the kernel above (conv1_2 1.42 / 1.43 ms, conv2_1 0.72 / 0.75, conv2_2 1.34 / 1.40).  What the variants measured on the way say:
//   * no patch loads at all (RE2E_WW_DBG=2): 0.715 of the matrix peak -- the k-step loop itself, 128 MFMAs + 184 vector instructions + 68 LDS reads
//     per patch, costs 9 900 cycles where the MFMAs alone are 8 192; with the compiler's unpacked transforms (300 vector instructions) 0.66: a
//     vector instruction costs this loop ~8 cycles of matrix time, consistent with tools/micro/mfma_coissue.hip;
//   * the loads cost 0.22 ms of 1.42 whichever way they travel: 0.12 with every load hitting L2 (RE2E_WW_DBG=1: issue instructions, LDS write port),
//     0.10 more from memory;
//   * sinking LDS reads next to their uses (what the scheduler does without the barriers) or not made no difference at one workgroup per CU.
// Packed fp32 operations as inline assembly: hipcc splits every <2 x float> operation whose results are consumed element-wise (the MFMA operands
// are scalars) into two unpacked ones -- also behind an empty asm that takes the pair -- and a vector instruction costs the matrix stream 8-10
// cycles here.  The compiler pads no hazard for an instruction it did not emit: a vector-ALU result needs two wait states before an MFMA may read
// it (gfx90a+; without them the sums are wrong).  The k-step pipeline below guarantees them by construction: what step s computes with these is
// read by the MFMAs of step s + 1, behind a scheduling barrier and an s_nop 1.
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { f32x2 d; asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }
__device__ __forceinline__ f32x2 pk_mul(f32x2 a, f32x2 b) { f32x2 d; asm("v_pk_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ f32x2 pk_add(f32x2 a, f32x2 b) { f32x2 d; asm("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ f32x2 pk_sub(f32x2 a, f32x2 b) { f32x2 d; asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b)); return d; }
template <int TXW>
__global__ __launch_bounds__(256, 1) void wino_wgrad_dma_kernel(WwArgs p) {
  constexpr int TYH = 32 / TXW, PW = 2 * TXW, PH = 2 * TYH;
  constexpr int XW = PW + 2, XH = PH + 2, XPIX = XW * XH;
  constexpr int XIT = ((XPIX + 3) / 4 + 3) / 4;          // wave-instructions per wave for x: 4 pixels each, 4 waves (12)
  constexpr int XPAD = XIT * 16;                          // pixels of the x region incl. the dead tail (192)
  constexpr int DIT = PW * PH / 32;                       // ... for dy: 8 pixels each (4)
  constexpr int STAGE = XPAD * WW_CB + PW * PH * WW_OB;   // floats (16 384 = 64 KB)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, lr = lane & 31, lh = lane >> 5;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int xk = blockIdx.x >> 3;           // (XCD-aware order: see the kernel above)
  const int blk = p.flat ? blockIdx.x % p.nblk : xk % p.nblk, split = p.flat ? blockIdx.x / p.nblk : (xk / p.nblk) * 8 + (blockIdx.x & 7);
  if (split >= p.nsplit) return;
  const int cb = blk % p.ncb, ob = blk / p.ncb;
  const int c0 = cb * WW_CB, o0 = ob * WW_OB;
  const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsD = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dy), 0, p.dy_bytes, 0x00020000);
  const int per = (p.npatch + p.nsplit - 1) / p.nsplit;
  const int pbeg = split * per, pend = min(p.npatch, pbeg + per);

  // this lane's pieces: x instruction i of wave w covers pixels 4 (w + 4 i) .. + 3 of the (XH x XW) halo patch, 16 bytes of 4 channels per lane;
  // dy instruction i covers pixels 8 (w + 4 i) .. + 7 of the (PH x PW) patch, 4 output channels per lane
  int xrel[XIT], xyx[XIT], drel[DIT], dyx[DIT];
#pragma unroll
  for (int i = 0; i < XIT; ++i) {
    const int pix = 4 * (wid + 4 * i) + (lane >> 4), yy = pix / XW, xx = pix - yy * XW;
    xrel[i] = ((yy * p.W + xx) * p.C + (lane & 15) * 4) * 4;
    xyx[i] = pix < XPIX ? (yy << 8 | xx) : 0x7f7f;            // dead tail: never inside an image
  }
#pragma unroll
  for (int i = 0; i < DIT; ++i) {
    const int pix = 8 * (wid + 4 * i) + (lane >> 3), yy = pix / PW, xx = pix - yy * PW;
    drel[i] = ((yy * p.W + xx) * p.Cout + (lane & 7) * 4) * 4;
    dyx[i] = yy << 8 | xx;
  }
  auto issue = [&](int stage, int pi) {
    const int ppi = p.py * p.px;
    const int n = __builtin_amdgcn_readfirstlane(pi / ppi), rem = pi - n * ppi;
    const int pyi = __builtin_amdgcn_readfirstlane(rem / p.px), pxi = rem - pyi * p.px;
    const int y0 = pyi * PH, x0 = pxi * PW;
    const int xbase = (((n * p.H + y0 - 1) * p.W + x0 - 1) * p.C + c0) * 4;       // may be negative; + xrel of a pixel inside the image is not
    const int dbase = (((n * p.H + y0) * p.W + x0) * p.Cout + o0) * 4;
    float* S = smem + stage * STAGE + wid * 256;
    if (y0 >= 1 && x0 >= 1 && y0 + PH + 1 <= p.H && x0 + PW + 1 <= p.W) {       // every halo pixel inside the image: one add per piece
#pragma unroll
      for (int i = 0; i < XIT; ++i) {
        unsigned v = (unsigned)(xbase + xrel[i]);
        if (4 * (3 + 4 * i) + 3 >= XPIX) v = xyx[i] != 0x7f7f ? v : WW_OOB;         // (only the last piece of a wave can lie in the dead tail)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (__attribute__((address_space(3))) void*)(S + i * 1024), 16, v, 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int i = 0; i < XIT; ++i) {
        const int yy = xyx[i] >> 8, xx = xyx[i] & 255;
        const bool ok = (unsigned)(y0 - 1 + yy) < (unsigned)p.H && (unsigned)(x0 - 1 + xx) < (unsigned)p.W;
        const unsigned v = ok ? (unsigned)(xbase + xrel[i]) : WW_OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (__attribute__((address_space(3))) void*)(S + i * 1024), 16, v, 0, 0, 0);
      }
    }
    float* D = S + XPAD * WW_CB;
    if (y0 + PH <= p.H && x0 + PW <= p.W) {
#pragma unroll
      for (int i = 0; i < DIT; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsD, (__attribute__((address_space(3))) void*)(D + i * 1024), 16, (unsigned)(dbase + drel[i]), 0, 0, 0);
    } else {
#pragma unroll
      for (int i = 0; i < DIT; ++i) {
        const int yy = dyx[i] >> 8, xx = dyx[i] & 255;
        const bool ok = y0 + yy < p.H && x0 + xx < p.W;
        const unsigned v = ok ? (unsigned)(dbase + drel[i]) : WW_OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsD, (__attribute__((address_space(3))) void*)(D + i * 1024), 16, v, 0, 0, 0);
      }
    }
  };

  const int rA = wid == 0 ? 0 : (wid == 2 ? 2 : 1), rB = wid == 3 ? 3 : (wid == 2 ? 1 : 2);
  const float sg = wid == 1 ? 1.f : -1.f;
  const float a0 = wid == 3 ? 0.f : 1.f, a1 = wid == 0 ? 0.f : (wid == 1 ? 1.f : -1.f);
  const f32x2 sg2 = {sg, sg}, a02 = {a0, a0}, a12 = {a1, a1};
  constexpr int HR = TYH / 2;
  const int xA_o = ((rA + 2 * HR * lh) * XW) * WW_CB + 2 * lr;
  const int xB_o = ((rB + 2 * HR * lh) * XW) * WW_CB + 2 * lr;
  const int dB_o = XPAD * WW_CB + (2 * HR * lh * PW) * WW_OB + lr;

  f32x16 acc[4][2];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][ct][r] = 0.f;

  if (pbeg < pend) issue(0, pbeg);
  int cur = 0;
  for (int pi = pbeg; pi < pend; ++pi) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's pieces of patch pi have landed (issued a whole patch of products ago)
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();                        // ... everybody's; and everybody is done reading the other stage (patch pi - 1)
    asm volatile("" ::: "memory");
    if (pi + 1 < pend && !(p.dbg & 2)) issue(cur ^ 1, (p.dbg & 1) ? pbeg : pi + 1);
    const float* S = smem + cur * STAGE;
    const float* xA = S + xA_o;
    const float* xB = S + xB_o;
    const float* dB = S + dB_o;
    // Three-deep software pipeline over the 16 k-steps: LDS reads of step s + 2, transforms of step s + 1, MFMAs of step s.  With the reads
    // and the transforms of ONE step side by side (the register-staged kernel above) a transform waits for an LDS read issued a few
    // instructions earlier, and a wave is in-order: its MFMA stream stops with it (~830 cycles per k-step of 8 MFMAs where 512 + the vector
    // instructions' own ~100 would do).  Subtractions are written as x + m1 * y so that they stay packed (there is no packed subtract).
    f32x2 ra[3][4], rb[3][4], re[3][2];        // raw: input rows A / B of up to 4 tile columns; dy rows 0 / 1 of the tile's 2 columns
    f32x2 t[4], v[2][4];
    float m[2][4];
    auto rd = [&](const int s, f32x2 (&a)[4], f32x2 (&b)[4], f32x2 (&e)[2]) {
      const int tyl = s / TXW, txl = s % TXW;
#pragma unroll
      for (int j = (txl != 0 ? 2 : 0); j < 4; ++j) {
        a[j] = *reinterpret_cast<const f32x2*>(xA + ((2 * tyl) * XW + 2 * txl + j) * WW_CB);
        b[j] = *reinterpret_cast<const f32x2*>(xB + ((2 * tyl) * XW + 2 * txl + j) * WW_CB);
      }
      e[0] = f32x2{dB[((2 * tyl) * PW + 2 * txl) * WW_OB], dB[((2 * tyl) * PW + 2 * txl + 1) * WW_OB]};
      e[1] = f32x2{dB[((2 * tyl + 1) * PW + 2 * txl) * WW_OB], dB[((2 * tyl + 1) * PW + 2 * txl + 1) * WW_OB]};
    };
    // 10-12 vector instructions per k-step (the register-staged kernel: 21): 2-4 row combinations, 4 column combinations, 4 for the dy side
    auto xf = [&](const int s, const f32x2 (&a)[4], const f32x2 (&b)[4], const f32x2 (&e)[2], f32x2 (&vv)[4], float (&mm)[4]) {
      const int txl = s % TXW;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (txl != 0 && j < 2) { t[j] = t[j + 2]; continue; }
        t[j] = pk_fma(sg2, b[j], a[j]);
      }
      vv[0] = pk_sub(t[0], t[2]); vv[1] = pk_add(t[1], t[2]); vv[2] = pk_sub(t[2], t[1]); vv[3] = pk_sub(t[1], t[3]);
      const f32x2 s01 = pk_fma(a12, e[1], pk_mul(a02, e[0]));          // (s0, s1) = row i of A dY for the tile's two dy columns
      mm[0] = s01[0]; mm[1] = s01[0] + s01[1]; mm[2] = s01[0] - s01[1]; mm[3] = s01[1];      // position 3 takes -s1: its accumulators change sign once, below
    };
    rd(0, ra[0], rb[0], re[0]);
    rd(1, ra[1], rb[1], re[1]);
    xf(0, ra[0], rb[0], re[0], v[0], m[0]);
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      // (sched_barrier: left to itself the scheduler sinks every LDS read to just in front of its first use -- a wait for the whole LDS
      //  latency with one or two MFMAs in flight, two or three times per k-step)
      __builtin_amdgcn_sched_barrier(0);
      if (s + 2 < 16) rd(s + 2, ra[(s + 2) % 3], rb[(s + 2) % 3], re[(s + 2) % 3]);
      asm volatile("s_nop 1");                 // the inline-asm results of step s - 1 (or of the prologue) are two wait states old before an MFMA reads them
      __builtin_amdgcn_sched_barrier(0);
      if (s + 1 < 16) xf(s + 1, ra[(s + 1) % 3], rb[(s + 1) % 3], re[(s + 1) % 3], v[(s + 1) & 1], m[(s + 1) & 1]);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[j][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[s & 1][j][0], m[s & 1][j], acc[j][0], 0, 0, 0);
        acc[j][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[s & 1][j][1], m[s & 1][j], acc[j][1], 0, 0, 0);
      }
      if (s + 1 < 16) {
#pragma unroll
        for (int g = 0; g < 8; ++g) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // one MFMA
          __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);   // two vector instructions (of step s + 1)
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's reads of the stage are complete before it passes the next barrier
    cur ^= 1;
  }
  float* sl = p.slabs + ((long)split * 16 + 4 * wid) * p.C * p.Cout;
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int c = c0 + 2 * ((r & 3) + 8 * (r >> 2) + 4 * lh) + ct;
        sl[((long)j * p.C + c) * p.Cout + o0 + lr] = j == 3 ? -acc[j][ct][r] : acc[j][ct][r];
      }
}

// The bare hand-off of a recurrent chain: what ONE step costs when the participants do nothing but exchange their state.
// This is the chain kernels' (csrc/lstm.hip) latency roofline: a step cannot be shorter than the exchange it contains.
//
// A "group" is P workgroups (one per CU); per step each publishes V = 32 * U floats (U hidden units x 32 utterances) and
// must see what its peers published before it may publish again.  Patterns and protocols:
//   0  all-gather, 8-byte {value, 32-bit step tag} granules, 8-byte sc1 stores, 16-byte sc1 sweep   (round-1..3 forward)
//   1  all-gather, 4-byte values that carry a ONE-BIT tag in bit 30 (free for |v| < 2: an LSTM's h), 16-byte sc1 stores + sweep
//   2  as 1 with two sweeps in flight, half a round trip apart (pipelined polling)
//   3  all-to-all of P blocks of V floats per workgroup (the BPTT partial slabs): 16-byte sc1 stores, every wave drains,
//      barrier, one flag store per producer; consumer polls P flags, barrier, 16-byte sc1 loads               (round-1..3 backward)
//   4  all-to-all, 16-byte {3 values, 32-bit tag} granules: one hop instead of two, 4/3 of the bytes
//   5  all-gather with the flag form of 3 (V floats per producer)
// Participants per group P, groups G (independent chains running side by side: directions x utterance tiles), same-XCD
// placement (blocks dealt round-robin over the 8 XCDs: a launch of 8P blocks keeps those with blockIdx % 8 == 0).
// Every consumer checks every word it receives (integer checksum against the host's).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/handoff_probe.hip -o /tmp/handoff_probe && /tmp/handoff_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;
constexpr unsigned kSpin = 1u << 16;
constexpr int MAXIT = 16;

__device__ __host__ inline unsigned payload(int s, int prod, int e) {          // float bits in [0, 1): bit 30 clear
  unsigned m = (unsigned)(prod * 131 + e * 7 + s * 29) & 1023u;
  float f = (float)m * (1.0f / 1024.0f);
#ifdef __HIP_DEVICE_COMPILE__
  return __float_as_uint(f);
#else
  unsigned u; memcpy(&u, &f, 4); return u;
#endif
}

__device__ __forceinline__ void store16_sc1(void* p, u32x4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory"); }

template <int PROTO, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void exch(char* buf, unsigned* flags, u64* sums, unsigned* err, int T, int P, int U, int xs) {
  __shared__ int aborted;
  __shared__ u64 wsum[WAVES];
  if (blockIdx.x % xs) return;
  const int x = blockIdx.x / xs, g = blockIdx.y, G = gridDim.y;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  constexpr int NTH = WAVES * 64;
  const int V = 32 * U;
  const bool a2a = PROTO == 3 || PROTO == 4;
  // units of 16 bytes a consumer receives per step, and a producer's share of them
  const int upp = PROTO == 0 ? V / 2 : PROTO == 4 ? (V + 2) / 3 : V / 4;          // 16-byte units per (producer[, consumer]) block
  const int L = P * upp;
  const long par_b = (long)G * (a2a ? P : 1) * L * 16;                              // bytes per parity buffer
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(buf, 0, (int)(2 * par_b), 0x00020000);
  const long rd_base = a2a ? ((long)g * P + x) * L * 16 : (long)g * L * 16;
  if (tid == 0) aborted = 0;
  u64 sum = 0;
  __syncthreads();
  for (int s = 0; s < T; ++s) {
    const int p = s & 1;
    const unsigned tag1 = (unsigned)(((s >> 1) + 1) & 1) << 30;
    // ---- publish ----
    if (PROTO == 0) {
      for (int e = tid; e < V; e += NTH) {
        u64 gv = ((u64)(unsigned)(s + 1) << 32) | payload(s, x, e);
        __hip_atomic_store((u64*)(buf + p * par_b + rd_base + ((long)x * V + e) * 8), gv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    } else if (PROTO == 1 || PROTO == 2) {
      for (int u = tid; u < upp; u += NTH) {
        u32x4 v;
        for (int i = 0; i < 4; ++i) v[i] = payload(s, x, 4 * u + i) | tag1;
        store16_sc1(buf + p * par_b + rd_base + ((long)x * upp + u) * 16, v);
      }
    } else if (PROTO == 5) {
      for (int u = tid; u < upp; u += NTH) {
        u32x4 v;
        for (int i = 0; i < 4; ++i) v[i] = payload(s, x, 4 * u + i);
        store16_sc1(buf + p * par_b + rd_base + ((long)x * upp + u) * 16, v);
      }
    } else {                                                                   // all-to-all: block (consumer c, producer x)
      for (int l = tid; l < L; l += NTH) {
        const int c = l / upp, r = l % upp;
        u32x4 v;
        if (PROTO == 3) { for (int i = 0; i < 4; ++i) v[i] = payload(s, x, (4 * r + i + c) & 0xffff); }
        else { for (int i = 0; i < 3; ++i) v[i] = payload(s, x, (3 * r + i + c) & 0xffff); v[3] = (unsigned)(s + 1); }
        store16_sc1(buf + p * par_b + (((long)g * P + c) * L + (long)x * upp + r) * 16, v);
      }
    }
    if (PROTO == 3 || PROTO == 5) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) __hip_atomic_store(flags + ((long)(p * G + g) * P + x) * 32, (unsigned)(s + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (wid == 0) {
        for (unsigned spins = 0;; ++spins) {
          bool good = true;
          for (int q = lane; q < P; q += 64)
            good &= __hip_atomic_load(flags + ((long)(p * G + g) * P + q) * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(s + 1);
          if (__all(good)) break;
          if (spins > kSpin) { if (lane == 0) { aborted = 1; atomicExch(err, 1u); } break; }
          __builtin_amdgcn_s_sleep(1);
        }
      }
      __syncthreads();
      if (aborted) break;
    }
    // ---- sweep: 16-byte sc1 loads of this consumer's L units ----
    const unsigned rb = (unsigned)(p * par_b + rd_base) + (unsigned)tid * 16u;
    u32x4 a[MAXIT];
    auto issue = [&](u32x4* d) {
#pragma unroll
      for (int it = 0; it < MAXIT; ++it)
        if (it * NTH < L) d[it] = __builtin_amdgcn_raw_buffer_load_b128(rs, rb + (unsigned)(it * NTH) * 16u, 0, 16);
    };
    auto check = [&](const u32x4* d) {
      bool good = true;
#pragma unroll
      for (int it = 0; it < MAXIT; ++it) {
        if (it * NTH < L && it * NTH + tid < L) {
          if (PROTO == 0) good &= (d[it][1] == (unsigned)(s + 1)) & (d[it][3] == (unsigned)(s + 1));
          else if (PROTO == 4) good &= d[it][3] == (unsigned)(s + 1);
          else if (PROTO == 1 || PROTO == 2) good &= ((d[it][0] & d[it][1] & d[it][2] & d[it][3] & 0x40000000u) == tag1) & (((d[it][0] | d[it][1] | d[it][2] | d[it][3]) & 0x40000000u) == tag1);
        }
      }
      return good;
    };
    if (PROTO == 2) {
      u32x4 b[MAXIT];
      issue(a);
      __builtin_amdgcn_s_sleep(6);
      for (unsigned spins = 0;; ++spins) {
        asm volatile("" ::: "memory");
        issue(b);
        if (__all(check(a))) break;
        asm volatile("" ::: "memory");
        issue(a);
        if (__all(check(b))) {
#pragma unroll
          for (int it = 0; it < MAXIT; ++it) a[it] = b[it];
          break;
        }
        if (spins > kSpin) { if (lane == 0) { aborted = 1; atomicExch(err, 1u); } break; }
      }
    } else {
      for (unsigned spins = 0;; ++spins) {
        asm volatile("" ::: "memory");
        issue(a);
        if (__all(check(a))) break;
        if (spins > kSpin) { if (lane == 0) { aborted = 1; atomicExch(err, 1u); } break; }
        __builtin_amdgcn_s_sleep(1);
      }
    }
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
      if (it * NTH < L && it * NTH + tid < L) {
        if (PROTO == 0) sum += (u64)a[it][0] + a[it][2];
        else if (PROTO == 4) sum += (u64)a[it][0] + a[it][1] + a[it][2];
        else sum += (u64)(a[it][0] & 0xBFFFFFFFu) + (a[it][1] & 0xBFFFFFFFu) + (a[it][2] & 0xBFFFFFFFu) + (a[it][3] & 0xBFFFFFFFu);
      }
    }
    __syncthreads();
    if (aborted) break;
  }
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
  if (lane == 0) wsum[wid] = sum;
  __syncthreads();
  if (tid == 0) { u64 t = 0; for (int w = 0; w < WAVES; ++w) t += wsum[w]; sums[g * P + x] = t; }
}

template <int PROTO, int WAVES>
static double run(int P, int U, int G, int same, int T, bool verbose) {
  const int V = 32 * U;
  const bool a2a = PROTO == 3 || PROTO == 4;
  const int upp = PROTO == 0 ? V / 2 : PROTO == 4 ? (V + 2) / 3 : V / 4;
  const long L = (long)P * upp;
  if (L > (long)MAXIT * WAVES * 64) return -1;
  const long par_b = (long)G * (a2a ? P : 1) * L * 16;
  char* buf; unsigned* flags; u64* sums; unsigned* err;
  hipMalloc(&buf, 2 * par_b); hipMalloc(&flags, (size_t)2 * G * P * 128); hipMalloc(&sums, (size_t)G * P * 8); hipMalloc(&err, 4);
  const int xs = same ? 8 : 1;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  double best = 1e30;
  for (int rep = 0; rep < 4; ++rep) {
    hipMemset(buf, 0, 2 * par_b); hipMemset(flags, 0, (size_t)2 * G * P * 128); hipMemset(err, 0, 4); hipMemset(sums, 0, (size_t)G * P * 8);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((exch<PROTO, WAVES>), dim3(P * xs, G), dim3(WAVES * 64), 0, 0, buf, flags, sums, err, T, P, U, xs);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (rep) best = ms * 1e3 / T < best ? ms * 1e3 / T : best;
  }
  unsigned herr; hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost);
  std::vector<u64> hs((size_t)G * P); hipMemcpy(hs.data(), sums, hs.size() * 8, hipMemcpyDeviceToHost);
  // expected checksum of consumer x
  int bad = 0;
  for (int x = 0; x < P && x < 4; ++x) {
    u64 ex = 0;
    for (int s = 0; s < T; ++s)
      for (int pr = 0; pr < P; ++pr) {
        if (!a2a) { for (int e = 0; e < V; ++e) ex += payload(s, pr, e); }
        else if (PROTO == 3) { for (int r = 0; r < upp; ++r) for (int i = 0; i < 4; ++i) ex += payload(s, pr, (4 * r + i + x) & 0xffff); }
        else { for (int r = 0; r < upp; ++r) for (int i = 0; i < 3; ++i) ex += payload(s, pr, (3 * r + i + x) & 0xffff); }
      }
    for (int g = 0; g < G; ++g) bad += hs[(size_t)g * P + x] != ex;
  }
  if (verbose)
    printf("proto %d  P=%3d U=%2d (H=%4d) G=%d %-8s waves=%2d  KB in per WG %6.1f : %6.2f us/step%s%s\n", PROTO, P, U, P * U, G, same ? "same-XCD" : "spread", WAVES,
           L * 16 / 1024.0, best, herr ? "  ABORTED" : "", bad ? "  CHECKSUM MISMATCH" : "");
  fflush(stdout);
  hipFree(buf); hipFree(flags); hipFree(sums); hipFree(err);
  return best;
}

template <int PROTO>
static void sweep_cfg(int T) {
  // (P, U, G, same-XCD)
  const int cfg[][4] = {{4, 8, 1, 0},  {4, 8, 1, 1},  {8, 8, 1, 0},   {8, 8, 1, 1},   {8, 32, 2, 0}, {8, 32, 2, 1}, {16, 16, 2, 0}, {16, 16, 2, 1},
                        {32, 8, 2, 0}, {32, 8, 1, 1}, {64, 4, 2, 0},  {16, 32, 4, 0}, {32, 16, 4, 0}, {64, 8, 4, 0}, {64, 8, 1, 0}};
  for (auto& c : cfg) {
    if (c[0] * c[2] * (c[3] ? 8 : 1) > 2048) continue;
    run<PROTO, 8>(c[0], c[1], c[2], c[3], T, true);
    if (c[0] * c[1] <= 256) run<PROTO, 4>(c[0], c[1], c[2], c[3], T, true);
  }
}

int main(int argc, char** argv) {
  const int T = argc > 1 ? atoi(argv[1]) : 2000;
  printf("hand-off probe: T=%d steps per launch, best of 3; a group = one chain direction x 32 utterances\n", T);
  sweep_cfg<0>(T);
  sweep_cfg<1>(T);
  sweep_cfg<2>(T);
  sweep_cfg<5>(T);
  sweep_cfg<3>(T);
  sweep_cfg<4>(T);
  return 0;
}

// Diagnostic build of the halo-patch convolution (csrc/conv3x3.hip compiled with RE2E_HALO_STAMPS): where do a workgroup's
// cycles go?  s_memtime stamps of thread 0 of every workgroup: 0 start, 1 first patch staged, then per channel chunk
// (2+3c) matrix block done, (3+3c) barrier passed, (4+3c) next chunk staged, 15 epilogue done.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DRE2E_HALO_STAMPS tools/micro/conv3x3_probe.hip -o /tmp/conv3x3_probe && /tmp/conv3x3_probe
#include <algorithm>
#include <stdarg.h>
#include <vector>
#include "../../robust_e2e_gan_amd/csrc/conv3x3.hip"

void re2e_set_error(const char*, ...) {}

int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 32, H = 800, W = 80, C = 64, K = 64;
  const size_t nin = (size_t)N * H * W * C, nout = (size_t)N * H * W * K, nw = (size_t)K * 9 * C;
  float *in, *out, *wg, *bias;
  hipMalloc(&in, nin * 4); hipMalloc(&out, nout * 4); hipMalloc(&wg, nw * 4); hipMalloc(&bias, K * 4);
  std::vector<float> h(nin);
  unsigned s = 12345;
  for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xFFFF) / 65536.0f - 0.5f; }
  hipMemcpy(in, h.data(), nin * 4, hipMemcpyHostToDevice);
  hipMemcpy(wg, h.data(), nw * 4, hipMemcpyHostToDevice);
  hipMemset(bias, 0, K * 4);
  HaloArgs a;
  a.in = in; a.wg = wg; a.out = out; a.bias = bias; a.mask = nullptr; a.pool_out = nullptr; a.pool_idx = nullptr; a.NI = N; a.H = H; a.W = W; a.C = C; a.Cout = K; a.act = RE2E_ACT_RELU; a.beta = 0.f;
  a.ngn = K / NT; a.in_bytes = (unsigned)(nin * 4); a.wg_bytes = (unsigned)(nw * 4); a.out_bytes = (unsigned)(nout * 4); a.tiles_x = cdiv(W, 16); a.tiles_y = cdiv(H, 16);
  a.nitems = N * a.tiles_x * a.tiles_y * a.ngn; a.ipw = 0;
  const int slots = getenv("RE2E_HALO_SLOTS") ? atoi(getenv("RE2E_HALO_SLOTS")) : 512;
  const long nwg = a.nitems < slots ? a.nitems : slots;
  hipMalloc(&a.stamps, nwg * 18 * 8);
  hipMemset(a.stamps, 0, nwg * 18 * 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int it = 0; it < 3; ++it) launch_halo<16, 16, 1, true>(a, 0);
  hipEventRecord(e0);
  for (int it = 0; it < 10; ++it) launch_halo<16, 16, 1, true>(a, 0);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
  printf("N=%d: %.3f ms per launch, %.1f TFLOP/s (%ld persistent workgroups, %d items)\n", N, ms, 2.0 * 9 * C * K * N * H * W / ms / 1e9, nwg, a.nitems);
  std::vector<unsigned long long> st(nwg * 18);
  hipMemcpy(st.data(), a.stamps, nwg * 18 * 8, hipMemcpyDeviceToHost);
  const char* names[] = {"stage chunk 0 (fetched under the previous item)", "matrix block", "barrier", "stage next chunk", "epilogue"};
  std::vector<double> acc(5, 0.0); std::vector<std::vector<unsigned long long>> all(5);
  unsigned long long tmin = ~0ull, tmax = 0;
  for (long w = 0; w < nwg; ++w) {
    const unsigned long long* t = &st[w * 16];
    tmin = std::min(tmin, t[0]); tmax = std::max(tmax, t[15]);
    all[0].push_back(0);
    unsigned long long mb = 0, br = 0, sg = 0;
    for (int c = 0; c < 4; ++c) { mb += t[2 + 3 * c] - (c ? t[4 + 3 * (c - 1)] : t[1]); br += t[3 + 3 * c] - t[2 + 3 * c]; sg += t[4 + 3 * c] - t[3 + 3 * c]; }
    all[1].push_back(mb); all[2].push_back(br); all[3].push_back(sg); all[4].push_back(t[15] - t[13]);
  }
  std::vector<double> clk;
  for (long w = 0; w < nwg; ++w) {
    const double dr = (double)(st[nwg * 16 + w * 2 + 1] - st[nwg * 16 + w * 2]);      // 100 MHz
    if (dr > 0) clk.push_back((double)(st[w * 16 + 15] - st[w * 16]) / dr * 0.1);
  }
  std::sort(clk.begin(), clk.end());
  printf("in-kernel shader clock (s_memtime / s_memrealtime over a workgroup's lifetime): median %.3f GHz, p5 %.3f, p95 %.3f\n",
         clk[clk.size() / 2], clk[clk.size() / 20], clk[clk.size() * 19 / 20]);
  {   // occupancy over the launch from the 100 MHz real-time stamps (one clock for the whole chip)
    unsigned long long r0 = ~0ull, r1 = 0; double sum = 0;
    for (long w = 0; w < nwg; ++w) {
      const unsigned long long a0 = st[nwg * 16 + w * 2], a1 = st[nwg * 16 + w * 2 + 1];
      r0 = std::min(r0, a0); r1 = std::max(r1, a1); sum += (double)(a1 - a0);
    }
    printf("last launch: first start -> last end %.1f us; sum of workgroup lifetimes / that span = %.1f workgroups resident on average (512 slots)\n",
           (r1 - r0) / 100.0, sum / (double)(r1 - r0));
    // resident workgroups sampled every 20 us (or 1/100 of the span)
    const unsigned long long dt = std::max<unsigned long long>(2000, (r1 - r0) / 100);     // at most 100 samples
    for (unsigned long long t = r0; t < r1; t += dt) {
      int c = 0;
      for (long w = 0; w < nwg; ++w) c += (st[nwg * 16 + w * 2] <= t && t < st[nwg * 16 + w * 2 + 1]);
      printf(" %d", c);
    }
    printf("\n");
  }
  double tot = 0;
  for (int k = 0; k < 5; ++k) { std::sort(all[k].begin(), all[k].end()); for (auto v : all[k]) acc[k] += v; tot += acc[k]; }
  printf("s_memtime ticks of the LAST item of each workgroup (thread 0), mean / median / p95, share:\n");
  for (int k = 0; k < 5; ++k)
    printf("  %-45s %9.0f %9llu %9llu   %5.1f %%\n", names[k], acc[k] / nwg, all[k][nwg / 2], all[k][nwg * 95 / 100], 100.0 * acc[k] / tot);
  {
    double life = 0; for (long w = 0; w < nwg; ++w) life += (double)(st[w * 16 + 15] - st[w * 16]);
    const double items_per_wg = (double)a.nitems / nwg;
    printf("  workgroup lifetime mean %.0f ticks for %.2f items = %.0f per item; ideal matrix block = 73728 SIMD cycles per item, %d workgroups per CU\n"
           "  => matrix pipe busy %.1f %% of the resident time\n", life / nwg, items_per_wg, life / nwg / items_per_wg, slots / 256,
           100.0 * (slots / 256) * 73728.0 / (life / nwg / items_per_wg));
  }
  return 0;
}

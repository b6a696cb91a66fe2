"""Per-call table of the implicit-GEMM engine: shape, duration, TFLOP/s.

Run on the GPU box:
    RE2E_IGEMM_LOG=1 RE2E_NO_OVERLAP=1 rocprofv3 --kernel-trace --output-format csv -d OUT -- \
        python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline 2> OUT/log.txt
    python3 tools/igemm_table.py OUT/log.txt OUT/*/*_kernel_trace.csv

The k-th "[igemm]" log line is the k-th engine launch (igemm_kernel, gemm_nt2_kernel, the halo / Winograd kernels; single stream => launch
order = trace order).  spl < 0: the product ran with a stream-K tail of -spl workgroups (csrc/gemm_nt.hip).
"""
import csv
import re
import sys
from collections import defaultdict


def main(log_path, trace_path):
    calls = []
    rows_m = None          # "[igemm-rows] M=<pixels>" (ops._note_row_limits): the pixels the NEXT Winograd launch computes (per-image row limits)
    for line in open(log_path, errors="replace"):
        mr = re.search(r"\[igemm-rows\] M=(\d+)", line)
        if mr:
            rows_m = int(mr.group(1))
            continue
        m = re.search(r"\[igemm\] A=(\w+) B=(\w+) tile=(\d+)x(\d+)x(\d+) vec=(\d) M=(\d+) N=(\d+) K=(\d+) splits=(-?\d+)", line)
        if m:
            a, b = m.group(1), m.group(2)
            vals = [int(x) for x in m.groups()[2:]]
            if a.startswith("Wino") and rows_m is not None:
                if a == "WinoW":
                    vals[6] = min(vals[6], rows_m)       # weight gradient: the pixels are the contraction (K)
                else:
                    vals[4] = min(vals[4], rows_m)       # forward / data gradient: the pixels are M
                rows_m = None
            calls.append((a, b) + tuple(vals))
    kern = []
    for r in csv.DictReader(open(trace_path)):
        if "igemm_kernel" in r["Kernel_Name"] or "gemm_nt2_kernel" in r["Kernel_Name"] or "conv3x3_halo_kernel" in r["Kernel_Name"] or "conv3x3_wgrad_kernel" in r["Kernel_Name"] or "wino_conv3x3_kernel" in r["Kernel_Name"] or "wino_wgrad_kernel" in r["Kernel_Name"]:
            kern.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    kern.sort()
    if len(kern) != len(calls):
        print(f"warning: {len(calls)} log lines vs {len(kern)} kernels; joining the common tail", file=sys.stderr)
    n = min(len(kern), len(calls))
    calls, kern = calls[-n:], kern[-n:]
    durs = defaultdict(list)
    for c, (t0, t1) in zip(calls, kern):
        a, b, bm, bn, bk, vec, M, N, K, s = c
        durs[(a, b, f"{bm}x{bn}x{bk}", vec, M, N, K, s)].append((t1 - t0) * 1e-3)
    agg = {}
    for key, d in durs.items():   # median per call: one-off first-launch outliers (code load) would skew a mean
        d.sort()
        med = d[len(d) // 2]
        agg[key] = [len(d), med * len(d), 2.0 * key[4] * key[5] * key[6] * len(d)]
    rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
    tot_us = sum(v[1] for _, v in rows)
    tot_fl = sum(v[2] for _, v in rows)
    print(f"{'A':7}{'B':7}{'tile':11}{'v':2}{'M':>9}{'N':>6}{'K':>9}{'spl':>4}{'calls':>6}{'us/call':>9}{'TF/s':>7}{'share':>7}")
    for (a, b, tile, vec, M, N, K, s), (cnt, us, fl) in rows:
        print(f"{a:7}{b:7}{tile:11}{vec:<2}{M:9d}{N:6d}{K:9d}{s:4d}{cnt:6d}{us / cnt:9.1f}{fl / us * 1e-6:7.1f}{us / tot_us:7.1%}")
    wino_fl = sum(v[2] for k, v in rows if k[0].startswith("Wino"))
    w44_fl = sum(v[2] for k, v in rows if k[0].startswith("W44"))
    exe_fl = tot_fl - wino_fl * (1.0 - 1.0 / 2.25) - w44_fl * (1.0 - 25.0 / 64.0)
    print(f"total {tot_us * 1e-3:.2f} ms, {tot_fl * 1e-12:.3f} TFLOP, {tot_fl / tot_us * 1e-6:.1f} TFLOP/s average")
    print(f"(Wino / W44 rows: direct-equivalent FLOPs of 3x3 / 4x4 convolutions run as Winograd F(2x2,3x3) / F(2x2,4x4), which execute 1/2.25 / 25/64 of them "
          f"(W44 rows: the engine launch only, the transform launches around it are not engine kernels); "
          f"executed by the matrix cores: {exe_fl * 1e-12:.3f} TFLOP = {exe_fl / tot_us * 1e-6:.1f} TFLOP/s average)")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])

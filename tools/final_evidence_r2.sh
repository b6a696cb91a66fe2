# Round-2 evidence run on the GPU box:  bash tools/final_evidence_r2.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2_final
rm -rf $O; mkdir -p $O
cd $R
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log; tail -2 $O/smoke.log
python bench.py > $O/bench_default.json 2> $O/bench_default.log; echo "bench rc=$?"; tail -4 $O/bench_default.log
for i in 1 2 3; do python bench.py --no-cpu-baseline --no-roofline --steps 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('repeat', d['ms_per_step'], d['value'])"; done | tee $O/bench_repeats.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- python3 bench.py --no-cpu-baseline > $O/prof_bench.json 2> $O/prof_bench.log
cp $O/prof_bench/*/*_kernel_stats.csv $O/bench_kernel_stats.csv
rm -f $O/prof_bench/*/*_kernel_trace.csv $O/prof_bench/*/*.db
bash tools/pmc_conv_r2.sh > $O/pmc.log 2>&1
# single-stream run with the engine's call log: per-call table + kernel stats for the HBM table
rm -rf $O/noov; mkdir -p $O/noov
RE2E_IGEMM_LOG=1 RE2E_NO_OVERLAP=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/noov -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > $O/noov/out.json 2> $O/noov/log.txt
python3 tools/igemm_table.py $O/noov/log.txt $O/noov/*/*_kernel_trace.csv > $O/igemm_calls_nooverlap.txt 2>&1; tail -1 $O/igemm_calls_nooverlap.txt
cp $O/noov/*/*_kernel_stats.csv $O/bench_nooverlap_kernel_stats.csv
rm -f $O/noov/*/*_kernel_trace.csv $O/noov/*/*.db
python3 tools/hbm_table.py $O/bench_nooverlap_kernel_stats.csv 7 > $O/hbm_kernels.md 2> $O/hbm_kernels.err
python tools/step_timeline.py 2>&1 | grep -v amdgpu.ids > $O/step_timeline.txt
BENCH_CONV_CHILD=1 python tools/bench_conv3x3.py 2>&1 | grep -v amdgpu.ids > $O/bench_conv3x3.txt
python tools/bench_gemm.py 2>&1 | grep -v amdgpu.ids > $O/bench_gemm.txt
python tools/bench_lstm_persist.py 2>&1 | grep -v amdgpu.ids > $O/bench_lstm_persist.txt
python -m pytest tests -m gpu -q --durations=8 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; grep -E "passed|failed|rc=" $O/pytest_gpu.log | tail -3

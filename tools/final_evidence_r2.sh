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
python -m pytest tests -m gpu -q --durations=8 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -14 $O/pytest_gpu.log

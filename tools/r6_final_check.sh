# last check of the final tree: smoke, every GPU test, three bench repeats
export GPU_MAX_HW_QUEUES=8
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_check; rm -rf $O; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log | cut -c1-120
python -m pytest tests -m gpu -q --durations=5 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest_gpu.log | tail -2
for i in 1 2 3; do python bench.py --no-cpu-baseline --no-roofline --no-other-configs --no-input-side --steps 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('repeat', d['ms_per_step'], d['value'])"; done | tee $O/bench_repeats.txt

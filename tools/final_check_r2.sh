# last check of the committed tree on the GPU box:  bash tools/final_check_r2.sh
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2_check; rm -rf $O; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"
python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest_gpu.log | tail -2
python bench.py > $O/bench_default.json 2> $O/bench_default.log; echo "bench rc=$?"
for i in 1 2 3; do python bench.py --no-cpu-baseline --no-roofline --steps 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('repeat', d['ms_per_step'], d['value'])"; done | tee $O/bench_repeats.txt

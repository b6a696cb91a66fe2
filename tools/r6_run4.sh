# Round-6 run 4: bias gradients accumulated inside the backward recurrence (re2e_lstm_seq_bwd dbias): parity tests + step time
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6_run4
rm -rf $O; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_abi.py tests/test_kernels_gpu.py tests/test_modules_gpu.py tests/test_trainers_gpu.py tests/test_dp_gpu.py -m gpu -x -q 2>&1 | tail -6 | tee $O/pytest.txt
for i in 1 2 3; do python bench.py --no-cpu-baseline --no-roofline --no-other-configs --no-input-side --steps 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('repeat', d['ms_per_step'], d['value'], d['step_executed_tflop'])"; done | tee $O/bench_repeats.txt

# Round-6 check run 1: the changed paths (fill rows with act(bias), guarded validation, stream-K stress, encoder at production width), full-size
# parity at the tightened tolerance with the worst ratios printed, then a first bench line with the executed-FLOP figures
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6_run1
rm -rf $O; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_abi.py tests/test_kernels_gpu.py tests/test_modules_gpu.py tests/test_trainers_gpu.py -m gpu -x -q 2>&1 | tail -15 | tee $O/pytest_part.txt
RE2E_PRINT_WORST=1 timeout 1500 python -m pytest tests/test_fullsize_gpu.py -m gpu -x -q -s -k "config4_full_size or config5_full_step" 2>&1 | grep -v amdgpu.ids | tail -12 | tee $O/pytest_fullsize.txt
timeout 900 python bench.py --no-cpu-baseline --no-other-configs --no-roofline --steps 20 > $O/bench_quick.json 2> $O/bench_quick.log; tail -c 3000 $O/bench_quick.json

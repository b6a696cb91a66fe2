cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r6_soak
timeout 900 python -m pytest tests/test_modules_gpu.py tests/test_trainers_gpu.py tests/test_dp_gpu.py tests/test_kernels_gpu.py -m gpu -x -q 2>&1 | tail -3
timeout 800 python tools/soak_step.py 300 2>&1 | grep -v amdgpu.ids > gpurun_out/r6_soak/soak.txt; tail -14 gpurun_out/r6_soak/soak.txt
SOAK_NO_LATE_READ=1 timeout 800 python tools/soak_step.py 200 2>&1 | grep -v amdgpu.ids > gpurun_out/r6_soak/soak_free_running.txt; tail -9 gpurun_out/r6_soak/soak_free_running.txt
for i in 1 2; do python bench.py --no-cpu-baseline --no-roofline --no-other-configs --no-input-side --steps 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('repeat', d['ms_per_step'], d['value'])"; done

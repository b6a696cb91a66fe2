# round 6: the soak in its three read-back modes + the tests around fit
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r6_soak
timeout 900 python -m pytest tests/test_trainers_gpu.py tests/test_dp_gpu.py tests/test_modules_gpu.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error" | tail -3
timeout 800 python tools/soak_step.py 300 2>&1 | grep -v amdgpu.ids > gpurun_out/r6_soak/soak.txt; tail -7 gpurun_out/r6_soak/soak.txt
SOAK_DRAINING_READ=1 timeout 800 python tools/soak_step.py 200 2>&1 | grep -v amdgpu.ids > gpurun_out/r6_soak/soak_draining_read.txt; tail -5 gpurun_out/r6_soak/soak_draining_read.txt
SOAK_NO_LATE_READ=1 timeout 800 python tools/soak_step.py 200 2>&1 | grep -v amdgpu.ids > gpurun_out/r6_soak/soak_free_running.txt; tail -5 gpurun_out/r6_soak/soak_free_running.txt

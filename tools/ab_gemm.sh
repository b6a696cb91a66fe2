#!/bin/bash
# same-session A/B of two builds of the library on the engine micro-benchmark and the step
mkdir -p gpurun_out/ab
python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm or linear or conv" > gpurun_out/ab/tests_gemm.txt 2>&1; tail -3 gpurun_out/ab/tests_gemm.txt
for r in 1 2; do
  echo "== new"; python tools/bench_gemm.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/ab/gemm_new.txt
  echo "== old"; RE2E_LIB=$PWD/ab/libre2e_hip_old.so python tools/bench_gemm.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/ab/gemm_old.txt
done
for r in 1 2 3; do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('new', d['ms_per_step'], d['roofline']['frac'])"
  RE2E_LIB=$PWD/ab/libre2e_hip_old.so python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('old', d['ms_per_step'], d['roofline']['frac'])"
done

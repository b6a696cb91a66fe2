#!/bin/bash
# same-session A/B of two builds of the library (RE2E_LIB=ab/libre2e_hip_old.so = the previous commit's)
mkdir -p gpurun_out/ab
python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "conv or halo" > gpurun_out/ab/tests.txt 2>&1; tail -3 gpurun_out/ab/tests.txt
for r in 1 2; do
  echo "== new"; BENCH_CONV_CHILD=1 python tools/bench_conv3x3.py 2>&1 | tee -a gpurun_out/ab/conv_new.txt
  echo "== old"; RE2E_LIB=$PWD/ab/libre2e_hip_old.so BENCH_CONV_CHILD=1 python tools/bench_conv3x3.py 2>&1 | tee -a gpurun_out/ab/conv_old.txt
done
for r in 1 2; do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('new', d['ms_per_step'], d['roofline']['frac'])"
  RE2E_LIB=$PWD/ab/libre2e_hip_old.so python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('old', d['ms_per_step'], d['roofline']['frac'])"
done

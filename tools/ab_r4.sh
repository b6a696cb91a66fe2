#!/bin/bash
# several one-variable variants of the step against the default, interleaved in one GPU session; args: env assignments
# ("A=1,B=2" sets two variables).  RE2E_LIB of the caller applies to every run.
for r in 1 2; do
  python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-input-side 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('default', d['ms_per_step'])"
  for v in "$@"; do
    env ${v//,/ } python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-input-side 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'])"
  done
done

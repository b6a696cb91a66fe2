#!/bin/bash
# several one-variable variants of the step against the default, interleaved (same GPU session); args: env assignments
for r in 1 2 3; do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('default', d['ms_per_step'])"
  for v in "$@"; do
    env $v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'])"
  done
done

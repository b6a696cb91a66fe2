#!/bin/bash
# same-session A/B of the step: "$1" = env assignment that selects the variant B (e.g. RE2E_NO_NT_INPUT_GRAD=1)
for r in 1 2 3; do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('A', d['ms_per_step'], d['roofline']['frac'])"
  env $1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B $1', d['ms_per_step'], d['roofline']['frac'])"
done

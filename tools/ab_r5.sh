#!/bin/bash
# same-session A/B of the training step (experiments build): each argument is one arm = a comma-separated list of VAR=value (or "base")
#   bash tools/ab_r5.sh base RE2E_NT2_MAXWG=2 RE2E_TN2=6,,1+RE2E_NT2_MAXWG=2      ('+' separates assignments, ',,' is a literal comma)
export RE2E_EXPERIMENTS=1 RE2E_LIB=$PWD/robust_e2e_gan_amd/libre2e_hip_exp.so
REPS=${REPS:-2}
for r in $(seq $REPS); do
  for arm in "$@"; do
    envs=""
    if [ "$arm" != "base" ]; then envs=$(echo "$arm" | sed 's/,,/\x01/g; s/+/ /g; s/\x01/,/g'); fi
    env $envs timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-input-side --no-other-configs 2>/dev/null | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-60s %8.3f ms  %8.2f utt/s' % ('$arm', d['ms_per_step'], d['value']))"
  done
done

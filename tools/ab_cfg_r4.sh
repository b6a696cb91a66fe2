for r in 1 2; do
for c in 2 5; do
  for v in "X=0" "RE2E_LSTM_OWN_CU_FRAC=4" "RE2E_LSTM_OWN_CU_FRAC=1" "RE2E_DEC_OWN_CU=1"; do
    env $v python bench.py --config $c --steps 8 --warmup 2 --no-cpu-baseline --no-roofline --no-input-side 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg$c $v', d['ms_per_step'])"
  done
done
done

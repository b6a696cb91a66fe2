# round 6: the N1 trainers' multi-stream step -- tests, then configs 1-3 of the bench with and without it (RE2E_NO_OVERLAP=1)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_n1; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_trainers_gpu.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|error" | tail -5 | tee $O/tests.txt
timeout 900 python -m pytest tests/test_fullsize_gpu.py -m gpu -x -q -k "config1 or config2 or config3" 2>&1 | grep -E "passed|failed|Error|error" | tail -5 | tee -a $O/tests.txt
for r in 1 2; do
  for c in ${CONFIGS:-2 3}; do
    for e in 0 1; do
      RE2E_NO_OVERLAP=$e python bench.py --config $c --no-cpu-baseline --no-roofline --no-input-side --steps 20 2>/dev/null | tail -1 | \
        python -c "import sys,json; d=json.loads(sys.stdin.read()); print('config $c  RE2E_NO_OVERLAP=$e  %8.3f ms  %8.1f utt/s' % (d['ms_per_step'], d['value']))" | tee -a $O/ab.txt
    done
  done
done

# generic round-3 GPU job:  bash tools/r3_run.sh <tag> "<pytest args or empty>" "<env for variant B or empty>" [repeats]
cd $GRAFT_REPO_ROOT
TAG=$1; PT="$2"; VB="$3"; N=${4:-2}
O=gpurun_out/r3_$TAG; rm -rf $O; mkdir -p $O
if [ -n "$PT" ]; then python -m pytest $PT -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest.log; fi
for r in $(seq $N); do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>$O/benchA_$r.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('A', d['ms_per_step'], d.get('parity'))" | tee -a $O/ab.txt
  if [ -n "$VB" ]; then env $VB python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>$O/benchB_$r.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B $VB', d['ms_per_step'])" | tee -a $O/ab.txt; fi
done

# config-5 A/B of python-side switches:  bash tools/r3_cfg5.sh "ENV1" "ENV2" ...
cd $GRAFT_REPO_ROOT
run() { env $1 python bench.py --config 5 --steps 8 --warmup 3 --no-cpu-baseline --no-roofline --no-input-side 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'])"; }
run "X=0"
for e in "$@"; do run "RE2E_EXPERIMENTS=1 $e"; done
run "X=0"

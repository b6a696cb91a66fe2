# overlapped kernel stats of the default step for a few kernels:  bash tools/r3_stats.sh "<name substrings separated by |>"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3_stats; rm -rf $O; mkdir -p $O; cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $O/p -- python3 bench.py --no-cpu-baseline --no-roofline --no-input-side --steps 10 --warmup 3 > $O/out.json 2> $O/log.txt
cp $O/p/*/*_kernel_stats.csv $O/kernel_stats.csv; rm -rf $O/p
python3 - "$1" <<'PY'
import csv, sys, json, os
O = os.path.join(os.environ['GRAFT_REPO_ROOT'], 'gpurun_out', 'r3_stats')
print('ms_per_step', json.loads(open(O + '/out.json').read().strip().splitlines()[-1])['ms_per_step'])
for r in csv.DictReader(open(O + '/kernel_stats.csv')):
    if any(k in r['Name'] for k in sys.argv[1].split('|')):
        print('%-60s calls %4s avg %8.1f min %8.1f max %8.1f us' % (r['Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:60], r['Calls'], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3, float(r['MaxNs']) / 1e3))
PY

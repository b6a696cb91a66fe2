# kernel trace of one step -> per-2ms stream bins + main/side lists:  bash tools/r3_trace.sh <tag> [env...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; shift
O=$R/gpurun_out/r3_trace_$TAG; rm -rf $O; mkdir -p $O
cd $R
for kv in "$@"; do export $kv; done
rocprofv3 --kernel-trace --output-format rocpd -d $O/t -- python3 bench.py --no-cpu-baseline --no-roofline --steps 3 --warmup 3 > $O/out.json 2> $O/log.txt
DB=$(ls $O/t/*/*.db | head -1)
python3 tools/trace_bins.py $DB 1 > $O/bins.txt 2>&1
python3 tools/trace_list.py $DB 0 90 0 0 > $O/main_list.txt 2>&1
python3 tools/trace_list.py $DB 0 90 1 0 > $O/side_list.txt 2>&1
python3 tools/trace_list.py $DB 0 90 2 0 > $O/wgrad_list.txt 2>&1
RE2E_TIMELINE=1 python3 tools/step_timeline.py > $O/timeline.txt 2>&1
rm -rf $O/t

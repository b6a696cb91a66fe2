# kernel trace of one step -> per-ms stream bins + main/side/wgrad lists + timeline:  bash tools/r4_trace.sh <tag> [env...]
export GPU_MAX_HW_QUEUES=8
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; shift
O=$R/gpurun_out/r4_trace_$TAG; rm -rf $O; mkdir -p $O
cd $R
for kv in "$@"; do export $kv; done
rocprofv3 --kernel-trace --output-format rocpd -d $O/t -- python3 bench.py --no-cpu-baseline --no-roofline --no-input-side --no-other-configs --steps 3 --warmup 3 > $O/out.json 2> $O/log.txt
DB=$(ls $O/t/*/*.db | head -1)
python3 tools/trace_bins.py $DB 1 > $O/bins.txt 2>&1
python3 tools/trace_list.py $DB 0 90 0 0 > $O/main_list.txt 2>&1
python3 tools/trace_list.py $DB 0 90 1 0 > $O/side_list.txt 2>&1
python3 tools/trace_list.py $DB 0 90 2 0 > $O/wgrad_list.txt 2>&1
RE2E_TIMELINE=1 python3 tools/step_timeline.py > $O/timeline.txt 2>&1
rm -rf $O/t

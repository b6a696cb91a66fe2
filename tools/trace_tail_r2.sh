cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2_tail; rm -rf $O; mkdir -p $O
cd $R
rocprofv3 --kernel-trace --output-format rocpd -d $O/t -- python3 bench.py --no-cpu-baseline --no-roofline --steps 3 --warmup 3 > $O/out.json 2> $O/log.txt
DB=$(ls $O/t/*/*.db | head -1)
for r in 0 1 2; do python3 tools/trace_window.py $DB 55 80 $r; done > $O/window_tail.txt 2>&1
for r in 0 1 2; do python3 tools/trace_window.py $DB 0 12 $r; done > $O/window_head.txt 2>&1
python3 tools/trace_bins.py $DB 2 > $O/bins.txt 2>&1
rm -rf $O/t
cat $O/window_tail.txt

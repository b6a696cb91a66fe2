# round-3 evidence: no-overlap kernel stats + per-call engine table + overlapped stream bins + timeline
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r3_prof; rm -rf $O; mkdir -p $O
RE2E_IGEMM_LOG=1 RE2E_NO_OVERLAP=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/noov -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > $O/noov_out.json 2> $O/noov_log.txt
python3 tools/igemm_table.py $O/noov_log.txt $O/noov/*/*_kernel_trace.csv > $O/igemm_table.txt 2>&1
cp $O/noov/*/*_kernel_stats.csv $O/noov_kernel_stats.csv
rm -rf $O/noov
rocprofv3 --kernel-trace --output-format rocpd -d $O/t -- python3 bench.py --no-cpu-baseline --no-roofline --steps 3 --warmup 3 > $O/out.json 2> $O/log.txt
DB=$(ls $O/t/*/*.db | head -1)
python3 tools/trace_bins.py $DB 2 > $O/bins.txt 2>&1
python3 tools/trace_list.py $DB 0 90 0 150 > $O/main_list.txt 2>&1
rm -rf $O/t
RE2E_TIMELINE=1 python3 tools/step_timeline.py > $O/timeline.txt 2>&1
tail -3 $O/igemm_table.txt

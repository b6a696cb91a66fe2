# per-call engine table of a single-stream step:  bash tools/r3_table.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3_table; rm -rf $O; mkdir -p $O; cd $R
RE2E_IGEMM_LOG=1 RE2E_NO_OVERLAP=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/noov -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-input-side > $O/noov_out.json 2> $O/noov_log.txt
python3 tools/igemm_table.py $O/noov_log.txt $O/noov/*/*_kernel_trace.csv > $O/igemm_table.txt 2>&1
cp $O/noov/*/*_kernel_stats.csv $O/noov_kernel_stats.csv
rm -rf $O/noov
tail -3 $O/igemm_table.txt

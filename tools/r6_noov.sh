# Round-6: single-stream step, kernel stats + per-call engine table (part (2) of the evidence run)
export GPU_MAX_HW_QUEUES=8
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6_noov
rm -rf $O; mkdir -p $O
cd $R
RE2E_IGEMM_LOG=1 RE2E_NO_OVERLAP=1 timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/noov -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-input-side --no-other-configs > $O/noov_out.json 2> $O/noov_log.txt
python3 tools/igemm_table.py $O/noov_log.txt $O/noov/*/*_kernel_trace.csv > $O/igemm_calls_nooverlap.txt 2>&1; tail -2 $O/igemm_calls_nooverlap.txt
cp $O/noov/*/*_kernel_stats.csv $O/bench_nooverlap_kernel_stats.csv; rm -rf $O/noov
python3 tools/hbm_table.py $O/bench_nooverlap_kernel_stats.csv > $O/hbm_kernels.md 2> $O/hbm_kernels.err

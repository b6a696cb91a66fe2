# round 6: kernel statistics of configurations 2 / 3 / 5 (single-stream run: stand-alone kernel durations) -- where their steps go
export GPU_MAX_HW_QUEUES=8
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6_other; rm -rf $O; mkdir -p $O
cd $R
for c in 2 3 5; do
  RE2E_NO_OVERLAP=1 timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c$c -- python3 bench.py --config $c --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --no-input-side > $O/config${c}_nooverlap.json 2> $O/config${c}.log
  cp $O/c$c/*/*_kernel_stats.csv $O/config${c}_nooverlap_kernel_stats.csv; rm -rf $O/c$c
  python bench.py --config $c --no-cpu-baseline --no-roofline --no-input-side 2>/dev/null | tail -1 > $O/config${c}.json
done
ls -la $O

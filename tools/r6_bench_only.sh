# the default bench line alone (after a change of bench.py that does not touch the timed path)
export GPU_MAX_HW_QUEUES=8
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6_final
python bench.py > gpurun_out/r6_final/bench_default.json 2> gpurun_out/r6_final/bench_default.log; echo "bench rc=$?"; tail -2 gpurun_out/r6_final/bench_default.log

# Round-6: kernel trace of the overlapped step -> stream bins + idle gaps of every stream
export GPU_MAX_HW_QUEUES=8
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6_trace
rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 600 rocprofv3 --kernel-trace --output-format rocpd -d $O/t -- python3 bench.py --no-cpu-baseline --no-roofline --no-input-side --no-other-configs --steps 3 --warmup 3 > $O/trace_out.json 2> $O/trace_log.txt
DB=$(ls $O/t/*/*.db | head -1)
python3 tools/trace_bins.py $DB 2 > $O/step_bins_2ms.txt 2>&1
python3 tools/trace_gaps.py $DB > $O/step_gaps.txt 2>&1
cat $O/step_gaps.txt
rm -rf $O/t

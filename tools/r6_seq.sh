export GPU_MAX_HW_QUEUES=8
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6_seq
rm -rf $O; mkdir -p $O
cd $R
RE2E_NO_OVERLAP=1 timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d $O/t -- python3 bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-roofline --no-input-side --no-other-configs > $O/out.json 2> $O/log.txt
python3 tools/trace_sequence.py $O/t/*/*_kernel_trace.csv 0 > $O/sequence_all.txt 2>&1
python3 tools/trace_sequence.py $O/t/*/*_kernel_trace.csv 30 > $O/sequence_30us.txt 2>&1
tail -1 $O/sequence_all.txt
rm -rf $O/t

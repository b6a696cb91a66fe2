# Round-6: vector-L1 access rate of every kernel of the single-stream step
export GPU_MAX_HW_QUEUES=8
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6_l1
rm -rf $O; mkdir -p $O
cd $R
RE2E_NO_OVERLAP=1 timeout -k 10 600 rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD --output-format csv -d $O/pass -- python3 bench.py --steps 3 --no-cpu-baseline --no-roofline --no-input-side --no-other-configs > $O/out.json 2> $O/log.txt
python3 tools/kernel_l1_rate.py $O/pass > $O/kernel_l1_rate.txt 2>&1
cat $O/kernel_l1_rate.txt
rm -rf $O/pass

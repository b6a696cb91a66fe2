# main-stream kernel list of the decoder window of one step (rocprofv3 kernel trace):  bash tools/trace_decoder_r3.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3_dec; rm -rf $O; mkdir -p $O
cd $R
python bench.py --no-cpu-baseline --no-roofline --steps 20 2>/dev/null | tail -1 > $O/bench_base.json
rocprofv3 --kernel-trace --output-format rocpd -d $O/t -- python3 bench.py --no-cpu-baseline --no-roofline --steps 3 --warmup 3 > $O/out.json 2> $O/log.txt
DB=$(ls $O/t/*/*.db | head -1)
python3 tools/trace_list.py $DB 14 48 0 0 > $O/main_list_mid_all.txt 2>&1
python3 tools/trace_bins.py $DB 2 > $O/bins.txt 2>&1
rm -rf $O/t

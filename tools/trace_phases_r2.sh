# main-stream kernel tables of the phases of one step from a rocprofv3 kernel trace:  bash tools/trace_phases_r2.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2_phases; rm -rf $O; mkdir -p $O
cd $R
rocprofv3 --kernel-trace --output-format rocpd -d $O/t -- python3 bench.py --no-cpu-baseline --no-roofline --steps 3 --warmup 3 > $O/out.json 2> $O/log.txt
DB=$(ls $O/t/*/*.db | head -1)
for w in "0 11.6" "11.6 16" "16 24.5" "24.5 28.2" "28.2 35.5" "35.5 47.6" "47.6 57.7" "57.7 73"; do python3 tools/trace_window.py $DB $w 0; done > $O/main_phases.txt 2>&1
python3 tools/trace_list.py $DB 0 12 0 40 > $O/main_list_head.txt 2>&1
python3 tools/trace_list.py $DB 0 12 1 40 > $O/side_list_head.txt 2>&1
python3 tools/trace_list.py $DB 56 74 0 40 > $O/main_list_tail.txt 2>&1
python3 tools/trace_list.py $DB 24 48 0 150 > $O/main_list_mid.txt 2>&1
python3 tools/trace_bins.py $DB 2 > $O/bins.txt 2>&1
rm -rf $O/t

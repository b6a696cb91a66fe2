set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
python bench.py --no-cpu-baseline > gpurun_out/r2_bench2.json 2> gpurun_out/r2_bench2.log; tail -3 gpurun_out/r2_bench2.log; cat gpurun_out/r2_bench2.json
rm -rf gpurun_out/r2_noov; mkdir -p gpurun_out/r2_noov
RE2E_IGEMM_LOG=1 RE2E_NO_OVERLAP=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2_noov -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > gpurun_out/r2_noov/out.json 2> gpurun_out/r2_noov/log.txt
tail -2 gpurun_out/r2_noov/out.json
python3 tools/igemm_table.py gpurun_out/r2_noov/log.txt gpurun_out/r2_noov/*/*_kernel_trace.csv > gpurun_out/r2_igemm_table.txt 2>&1
head -30 gpurun_out/r2_igemm_table.txt; tail -2 gpurun_out/r2_igemm_table.txt
cp gpurun_out/r2_noov/*/*_kernel_stats.csv gpurun_out/r2_noov_kernel_stats.csv
rm -f gpurun_out/r2_noov/*/*_kernel_trace.csv gpurun_out/r2_noov/*/*.db

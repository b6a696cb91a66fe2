# Round-6 evidence run on the GPU box:  bash tools/final_evidence_r6.sh [skip-bench|bench] [tests]
# Everything lands in gpurun_out/r6_final; tools/collect_evidence_r6.py copies the summaries into profiles/r06_*.
# GPU_MAX_HW_QUEUES as a plain shell export: under rocprofv3 the runtime is up before python starts, bench.py's own setdefault comes too late
export GPU_MAX_HW_QUEUES=8
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6_final
rm -rf $O; mkdir -p $O
cd $R
if [ "$1" != "skip-bench" ]; then
  python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log; tail -2 $O/smoke.log
  python bench.py > $O/bench_default.json 2> $O/bench_default.log; echo "bench rc=$?"; tail -3 $O/bench_default.log
fi
for i in 1 2 3; do python bench.py --no-cpu-baseline --no-roofline --no-other-configs --steps 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('repeat', d['ms_per_step'], d['value'], d.get('input_side'))"; done | tee $O/bench_repeats.txt
# (1) overlapped run: kernel stats of the default command
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- python3 bench.py --no-cpu-baseline --no-input-side --no-other-configs --no-roofline > $O/prof_bench.json 2> $O/prof_bench.log
cp $O/prof_bench/*/*_kernel_stats.csv $O/bench_kernel_stats.csv; rm -rf $O/prof_bench
# (2) single-stream run with the engine's call log: per-call table + kernel stats for the HBM table
RE2E_IGEMM_LOG=1 RE2E_NO_OVERLAP=1 timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/noov -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-input-side --no-other-configs > $O/noov_out.json 2> $O/noov_log.txt
python3 tools/igemm_table.py $O/noov_log.txt $O/noov/*/*_kernel_trace.csv > $O/igemm_calls_nooverlap.txt 2>&1; tail -2 $O/igemm_calls_nooverlap.txt
cp $O/noov/*/*_kernel_stats.csv $O/bench_nooverlap_kernel_stats.csv; rm -rf $O/noov
python3 tools/hbm_table.py $O/bench_nooverlap_kernel_stats.csv > $O/hbm_kernels.md 2> $O/hbm_kernels.err
# (3) whole-step counters: three separate --pmc passes over the single-stream step (MI355X_MICROARCH.md: never mix FETCH/WRITE with SQ)
for P in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "FETCH_SIZE" "WRITE_SIZE"; do
  T=$(echo $P | tr ' ' '_')
  RE2E_NO_OVERLAP=1 timeout -k 10 600 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $O/step_$T -- python3 bench.py --steps 3 --no-cpu-baseline --no-roofline --no-input-side --no-other-configs > $O/step_$T.json 2> $O/step_$T.log
done
python3 tools/step_pmc.py $O > $O/step_pmc.json 2> $O/step_pmc.err; head -c 1500 $O/step_pmc.json
python3 tools/kernel_clock.py $O/step_SQ_VALU_MFMA_BUSY_CYCLES_SQ_BUSY_CYCLES > $O/kernel_clock.txt 2> $O/kernel_clock.err; head -30 $O/kernel_clock.txt
rm -rf $O/step_*/
# (4) the roofline kernel alone (bench.py conv_roofline = tools/roofline_conv.py): traffic + SQ passes of THIS round, then launch statistics
for P in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_F32" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE" "SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA"; do
  T=$(echo $P | cut -d' ' -f1)
  timeout -k 10 600 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $O/conv_$T -- python3 tools/roofline_conv.py > $O/conv_$T.log 2>&1
done
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/conv_stats -- python3 tools/roofline_conv.py > $O/conv_stats.log 2>&1
cp $O/conv_stats/*/*_kernel_stats.csv $O/roofline_conv_kernel_stats.csv
python3 tools/conv_pmc.py $O > $O/conv_pmc.json 2> $O/conv_pmc.err; head -c 600 $O/conv_pmc.json
rm -rf $O/conv_*/
# (5) the overlapped step: stream bins, main-stream list, host timeline
timeout -k 10 600 rocprofv3 --kernel-trace --output-format rocpd -d $O/t -- python3 bench.py --no-cpu-baseline --no-roofline --no-input-side --no-other-configs --steps 3 --warmup 3 > $O/trace_out.json 2> $O/trace_log.txt
DB=$(ls $O/t/*/*.db | head -1)
python3 tools/trace_bins.py $DB 2 > $O/step_bins_2ms.txt 2>&1
rm -rf $O/t
RE2E_TIMELINE=1 python3 tools/step_timeline.py 2>&1 | grep -v amdgpu.ids > $O/step_timeline.txt
python tools/bench_kernels.py 2>&1 | grep -v amdgpu.ids > $O/bench_kernels.txt
RE2E_EXPERIMENTS=1 RE2E_LIB=$R/robust_e2e_gan_amd/libre2e_hip_exp.so timeout 900 python tools/bench_gemm2.py 0:3,0:3,2:6,0:6,2:8,0:8,2 0:6,2 2>/dev/null > $O/gemm_nt_variants.txt
python tools/bench_chain.py --all 2>&1 | grep -v amdgpu.ids > $O/chain_rates_alone.txt
for D in 0 128 64 97; do RE2E_WINO_DBG=$D RE2E_EXPERIMENTS=1 RE2E_LIB=$R/robust_e2e_gan_amd/libre2e_hip_exp.so timeout 300 python tools/bench_wino_ab.py lds_dbg$D 2>/dev/null >> $O/wino_final.txt; done
RE2E_WINO_LDSIN=0 RE2E_EXPERIMENTS=1 RE2E_LIB=$R/robust_e2e_gan_amd/libre2e_hip_exp.so timeout 300 python tools/bench_wino_ab.py per_lane 2>/dev/null >> $O/wino_final.txt
for D in 16 17; do RE2E_WINO_LDSIN=0 RE2E_WINO_DBG=$D RE2E_EXPERIMENTS=1 RE2E_LIB=$R/robust_e2e_gan_amd/libre2e_hip_exp.so timeout 300 python tools/bench_wino_ab.py per_lane_dbg$D 2>/dev/null >> $O/wino_final.txt; done
python tools/bench_decoder.py 2>&1 | grep -v amdgpu.ids > $O/decoder_loop_alone.txt
RE2E_EXPERIMENTS=1 RE2E_LIB=$R/robust_e2e_gan_amd/libre2e_hip_exp.so python tools/dec_stamps.py 2>&1 | grep -v amdgpu.ids > $O/decoder_loop_budget.txt
RE2E_EXPERIMENTS=1 RE2E_LIB=$R/robust_e2e_gan_amd/libre2e_hip_exp.so python tools/dec_stamps.py bwd 2>&1 | grep -v amdgpu.ids >> $O/decoder_loop_budget.txt
for c in 2 3 5; do python bench.py --config $c --no-cpu-baseline --no-roofline --no-input-side > $O/bench_config$c.json 2>/dev/null; done
if [ "$2" = "tests" ]; then python -m pytest tests -m gpu -q --durations=8 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; grep -E "passed|failed|rc=" $O/pytest_gpu.log | tail -3; fi
ls -la $O

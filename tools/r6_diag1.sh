# Round-6 diagnostics, part 1 (GPU box): where does the fused Winograd kernel wait?
#   (a) the diagnostic switches of the experiments build: weight loads always hit the same 2 KB (1), pixel loads always re-read chunk 0 (2), both (3),
#       two workgroups per CU and one (RE2E_WINO_LDS_KB=100)
#   (b) counter passes over bench.py's roofline launches (tools/roofline_conv.py), one --pmc set per run
export GPU_MAX_HW_QUEUES=8
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6_diag1
rm -rf $O; mkdir -p $O
cd $R
rocprofv3 -L > $O/counters_avail.txt 2>&1
export RE2E_EXPERIMENTS=1 RE2E_LIB=$R/robust_e2e_gan_amd/libre2e_hip_exp.so
for D in 0 1 2 3; do
  RE2E_WINO_DBG=$D timeout 300 python tools/bench_wino_ab.py dbg$D 2>/dev/null | grep -v wgrad >> $O/wino_dbg.txt
done
for D in 0 3; do
  RE2E_WINO_LDS_KB=100 RE2E_WINO_DBG=$D timeout 300 python tools/bench_wino_ab.py one_wg_dbg$D 2>/dev/null | grep -v wgrad >> $O/wino_dbg.txt
done
cat $O/wino_dbg.txt
unset RE2E_EXPERIMENTS RE2E_LIB
i=0
for P in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" \
         "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
         "SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES" \
         "GRBM_GUI_ACTIVE GRBM_COUNT" \
         "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
         "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
         "TA_BUSY_avr TA_TA_BUSY_sum TD_TD_BUSY_sum TCP_GATE_EN1_sum"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $O/pmc_$i -- python3 tools/roofline_conv.py > $O/pmc_$i.log 2>&1
  echo "pass $i ($P) rc=$?"
  F=$(ls $O/pmc_$i/*/*counter_collection.csv 2>/dev/null | head -1)
  if [ -n "$F" ]; then python3 tools/r6_pmc_sum.py $F wino_conv3x3 > $O/pmc_$i.txt 2>&1; cat $O/pmc_$i.txt; fi
  rm -rf $O/pmc_$i
done
ls -la $O

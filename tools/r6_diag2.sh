# Round-6 diagnostics, part 2: issue priority of the phases outside the Winograd matrix loop (experiments build, RE2E_WINO_DBG bits 4 / 8)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6_diag2
rm -rf $O; mkdir -p $O
cd $R
export RE2E_EXPERIMENTS=1 RE2E_LIB=$R/robust_e2e_gan_amd/libre2e_hip_exp.so
for D in 0 4 8 0 4 8; do
  RE2E_WINO_DBG=$D timeout 300 python tools/bench_wino_ab.py dbg$D 2>/dev/null | grep -v wgrad >> $O/wino_prio.txt
done
cat $O/wino_prio.txt
unset RE2E_EXPERIMENTS RE2E_LIB
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_modules_gpu.py -m gpu -x -q -k "valid_rows or padded_rows or gemm_nt or e2e or bilstm" 2>&1 | tail -5

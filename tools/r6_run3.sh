# Round-6 run 3: the fused Winograd kernel with the input patch staged through LDS (LDS-DMA): parity tests, the six launches alone (A/B against the
# per-lane loads, RE2E_WINO_LDSIN=0), the step A/B
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6_run3
rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "wino or conv or vgg" 2>&1 | tail -6 | tee $O/pytest_conv.txt
export RE2E_EXPERIMENTS=1 RE2E_LIB=$R/robust_e2e_gan_amd/libre2e_hip_exp.so
for i in 1 2; do
  timeout 300 python tools/bench_wino_ab.py lds_staged 2>/dev/null | grep -v wgrad >> $O/wino_ldsin.txt
  RE2E_WINO_LDSIN=0 timeout 300 python tools/bench_wino_ab.py per_lane 2>/dev/null | grep -v wgrad >> $O/wino_ldsin.txt
done
cat $O/wino_ldsin.txt
REPS=3 bash tools/ab_r5.sh base RE2E_WINO_LDSIN=0 2>&1 | tee $O/ab_wino_ldsin.txt

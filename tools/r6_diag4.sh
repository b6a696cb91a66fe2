# Round-6 diagnostics, part 4 (experiments build): what the LDS-staged Winograd kernel pays for its per-half barriers (bit 32) and for the pixel traffic (bit 64)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6_diag4
rm -rf $O; mkdir -p $O
cd $R
export RE2E_EXPERIMENTS=1 RE2E_LIB=$R/robust_e2e_gan_amd/libre2e_hip_exp.so
for D in 0 128 64 0 128; do
  RE2E_WINO_DBG=$D timeout 300 python tools/bench_wino_ab.py dbg$D 2>/dev/null | grep -v wgrad >> $O/wino_lds_dbg.txt
done
cat $O/wino_lds_dbg.txt

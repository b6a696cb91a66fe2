# Round-6 diagnostics, part 3: is the fused Winograd kernel bound by the vector L1's access rate?  RE2E_WINO_DBG bit 16: the pixel loads of a chunk read
# 1 KB contiguous each (wrong pixels, same bytes and instruction stream, 1/4 .. 1/8 of the L1 accesses)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6_diag3
rm -rf $O; mkdir -p $O
cd $R
export RE2E_EXPERIMENTS=1 RE2E_LIB=$R/robust_e2e_gan_amd/libre2e_hip_exp.so
for D in 0 16 17 0 16; do
  RE2E_WINO_DBG=$D timeout 300 python tools/bench_wino_ab.py dbg$D 2>/dev/null | grep -v wgrad >> $O/wino_coalesced.txt
done
cat $O/wino_coalesced.txt

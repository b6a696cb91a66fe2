# Round-6: robustness of the final build -- the stream-K tail and the persistent recurrences under load (experiments build for the tail's switches)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_stress; rm -rf $O; mkdir -p $O
RE2E_EXPERIMENTS=1 RE2E_LIB=$PWD/robust_e2e_gan_amd/libre2e_hip_exp.so timeout 900 python tools/stress_gemm_nt_tail.py 2>/dev/null | tail -4 | tee $O/streamk_tail.txt
timeout 900 python tools/stress_lstm_persist.py 2>/dev/null | tail -6 | tee $O/lstm_persist.txt

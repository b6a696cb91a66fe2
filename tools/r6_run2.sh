# Round-6 run 2: the two-utterance-tiles-per-workgroup forward recurrence (lstm_fwd2<.., TT = 2>): parity tests, the chain alone, the step A/B
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6_run2
rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "lstm or bilstm or colsum" 2>&1 | tail -6 | tee $O/pytest_lstm.txt
export RE2E_EXPERIMENTS=1 RE2E_LIB=$R/robust_e2e_gan_amd/libre2e_hip_exp.so
for i in 1 2; do
  echo "== two tiles per workgroup (shipped selection)" >> $O/chain_alone.txt
  python tools/bench_chain.py 2>&1 | grep -v amdgpu.ids >> $O/chain_alone.txt
  echo "== RE2E_LSTM_FWD2_TT=1 (round-5 selection: the round-1..3 kernel at H=256 / B=32)" >> $O/chain_alone.txt
  RE2E_LSTM_FWD2_TT=1 python tools/bench_chain.py 2>&1 | grep -v amdgpu.ids >> $O/chain_alone.txt
done
cat $O/chain_alone.txt
REPS=3 bash tools/ab_r5.sh base RE2E_LSTM_FWD2_TT=1 2>&1 | tee $O/ab_fwd2_tt.txt

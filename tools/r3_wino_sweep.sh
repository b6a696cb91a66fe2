cd $GRAFT_REPO_ROOT
for d in 0 2 3 4; do echo stagger=$d; RE2E_LIB=$PWD/robust_e2e_gan_amd/libre2e_hip_exp.so RE2E_EXPERIMENTS=1 RE2E_WINO_STAGGER=$d python tools/bench_wino.py 2>&1 | grep -v amdgpu | cut -c1-150 | head -3; done

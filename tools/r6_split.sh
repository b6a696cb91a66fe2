# round 6: the branch split (apply profiles/r06_branch_split.patch first: the experiment is not in the product) -- parity first, then a same-session A/B against the switch that turns it on, then the two timelines
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6_split
timeout 900 python -m pytest tests/test_fullsize_gpu.py -m gpu -x -q -k "branch_split or config4_architecture_vs or config4_full_size_vs" 2>&1 | tail -15 | tee gpurun_out/r6_split/tests.txt
timeout 600 python -m pytest tests/test_modules_gpu.py -m gpu -x -q -k "joint" 2>&1 | tail -5 | tee -a gpurun_out/r6_split/tests.txt
REPS=${REPS:-3} bash tools/ab_r5.sh RE2E_BRANCH_SPLIT=1 base 2>&1 | tee gpurun_out/r6_split/ab.txt
export RE2E_EXPERIMENTS=1 RE2E_LIB=$PWD/robust_e2e_gan_amd/libre2e_hip_exp.so
RE2E_BRANCH_SPLIT=1 timeout 300 python3 tools/step_timeline.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r6_split/tl_split.txt
timeout 300 python3 tools/step_timeline.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r6_split/tl_nosplit.txt

# same-session step A/B of round-6 switches (experiments build): bash tools/r6_ab.sh <arm> <arm> ...
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6_ab
REPS=${REPS:-3} bash tools/ab_r5.sh "$@" 2>&1 | tee gpurun_out/r6_ab/ab_$(date +%s).txt

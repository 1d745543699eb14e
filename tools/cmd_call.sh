# GPU-box call: one command ($2...) with its output under gpurun_out/$1.log
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; tag=$1; shift
timeout 900 "$@" 2>&1 | grep -v "^/opt" | tail -80 > gpurun_out/${tag}.log

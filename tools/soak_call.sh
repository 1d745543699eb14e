# GPU-box call: repeat a pytest selection N times, stop at the first failure: bash tools/soak_call.sh TAG N "pytest args"
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; : > gpurun_out/${1}_soak.txt
for i in $(seq $2); do
  timeout 600 python -m pytest $3 -q -x 2>&1 | tail -3 >> gpurun_out/${1}_soak.txt || break
  grep -q "failed" gpurun_out/${1}_soak.txt && break
done
grep -c "passed" gpurun_out/${1}_soak.txt >> gpurun_out/${1}_soak.txt

# GPU-box call: the tile-rescue tests, optionally the occupancy probe ($2 = probe) and the step with the rescue form off / on ($3 = ab)
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_tile_rescue_gpu.py -q -x -s 2>&1 | tail -60 > $O/${1}_rescue_tests.log
if [ "$2" = probe ]; then timeout 600 python tools/occupancy_hazard_probe.py 2>&1 | grep -v "^/opt" > $O/${1}_occupancy_probe.txt; fi
if [ "$3" = ab ]; then bash tools/rescue_ab.sh $1; fi

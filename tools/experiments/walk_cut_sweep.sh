#!/bin/bash
# usage (GPU box): bash tools/experiments/walk_cut_sweep.sh — times C3 in both black-path modes for several values of RMD_WALK_CUT (= K + 1: a walk call
# leaves the walks of its last K rays to the wave's next call; 1 = every call finishes every walk) and runs the full-size C3 parity tests.
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04_cut
for k in 1 5 3 9 13; do
  echo "== RMD_WALK_CUT=$k" | tee -a gpurun_out/r04_cut/sweep.log
  RMD_WALK_CUT=$k timeout -k 10 120 python tools/quick_time.py C3-end 200 3 2>&1 | tee -a gpurun_out/r04_cut/sweep.log
  RMD_WALK_CUT=$k timeout -k 10 120 python tools/quick_time.py C3 200 3 2>&1 | tee -a gpurun_out/r04_cut/sweep.log
done
timeout -k 10 600 python -m pytest tests/test_gpu_fullsize.py -x -q -k "C3" 2>&1 | tail -5 | tee -a gpurun_out/r04_cut/sweep.log

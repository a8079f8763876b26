#!/bin/bash
# usage (GPU box, repo root): tools/diag_ablate.sh WORKLOAD SPP — DIAG=1 build in /tmp; kernel time with RMD_DEBUG = 0 (everything), 1 (no triangle tests: every
# walk runs to its end, no mesh hits) and 2 (no grid walks at all).  Timing ablations: 1 and 2 render wrong frames.
set -e
wl=$1; spp=$2
rm -rf /tmp/repo_diag && mkdir -p /tmp/repo_diag && cp -r include raymond_amd /tmp/repo_diag/
make -s -C /tmp/repo_diag/raymond_amd/csrc clean
make -s -j8 -C /tmp/repo_diag/raymond_amd/csrc DIAG=1 2>&1 | grep -E "error" || true
for dbg in 0 1 2; do
  echo "== RMD_DEBUG=$dbg"
  RAYMOND_HIP_LIB=/tmp/repo_diag/raymond_amd/csrc/libraymond_hip.so RMD_DEBUG=$dbg python3 tools/quick_time.py $wl $spp 2>&1 | tail -3
done

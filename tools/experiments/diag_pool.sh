#!/bin/bash
# NEEDS tools/experiments/walk_pool.patch applied.  usage (GPU box, repo root): tools/experiments/diag_pool.sh SPP — one DIAG=1 build in /tmp; event counters (RMD_DEBUG=8) and per-phase time stamps (16) of the C3 frame with the
# walk pool on and off
spp=${1:-50}
rm -rf /tmp/repo_diag && mkdir -p /tmp/repo_diag && cp -r include raymond_amd /tmp/repo_diag/
make -s -C /tmp/repo_diag/raymond_amd/csrc clean
make -s -j16 -C /tmp/repo_diag/raymond_amd/csrc DIAG=1 libraymond_hip.so 2>&1 | grep -E "error" || true
for pool in 1 0; do for dbg in ${RMD_DIAG_LIST:-8 16}; do
  echo "== RMD_WALK_POOL=$pool (1 = off) RMD_DEBUG=$dbg"
  RAYMOND_HIP_LIB=/tmp/repo_diag/raymond_amd/csrc/libraymond_hip.so RMD_WALK_POOL=$pool RMD_DEBUG=$dbg timeout -k 10 200 python3 tools/quick_time.py C3 $spp 2>&1 | tail -4
done; done

#!/bin/bash
# usage (GPU box, repo root): tools/diag_counts.sh WORKLOAD SPP [EXTRA make flags] — builds a DIAG=1 copy of the library in /tmp and prints the walk / main-loop event counters (RMD_DIAG_FLAGS=16: time stamps per phase instead)
set -e
wl=$1; spp=$2; extra=$3
rm -rf /tmp/repo_diag && mkdir -p /tmp/repo_diag && cp -r include raymond_amd /tmp/repo_diag/
make -s -C /tmp/repo_diag/raymond_amd/csrc clean
make -s -j8 -C /tmp/repo_diag/raymond_amd/csrc DIAG=1 EXTRA="$extra" 2>&1 | grep -E "error" || true
RAYMOND_HIP_LIB=/tmp/repo_diag/raymond_amd/csrc/libraymond_hip.so RMD_DEBUG=${RMD_DIAG_FLAGS:-8} python3 tools/quick_time.py $wl $spp 2>&1 | tail -14

#!/bin/bash
# usage (GPU box, repo root): tools/ab_variants.sh WORKLOAD SPP "EXTRA flags A" "EXTRA flags B" ...
# Builds a copy of the library per variant under /tmp (the in-tree build stays as shipped) and times it with tools/quick_time.py.
# A variant may carry environment settings in front of a '|':  "RMD_WALK_BATCH=40|-DRMD_WALK_MAX_WAIT=6"
wl=$1; spp=$2; shift 2
rm -rf /tmp/repo_ab && mkdir -p /tmp/repo_ab && cp -r include raymond_amd /tmp/repo_ab/
for v in "$@"; do
  envs=""; extra="$v"
  if [[ "$v" == *"|"* ]]; then envs="${v%%|*}"; extra="${v#*|}"; fi
  make -s -C /tmp/repo_ab/raymond_amd/csrc clean
  make -s -j8 -C /tmp/repo_ab/raymond_amd/csrc EXTRA="$extra" 2>&1 | grep -E "error" || true
  echo "== env '$envs' EXTRA '$extra'"
  env $envs RAYMOND_HIP_LIB=/tmp/repo_ab/raymond_amd/csrc/libraymond_hip.so timeout -k 10 120 python3 tools/quick_time.py $wl $spp | sed -n 2,3p
done

#!/bin/bash
# usage (GPU box, repo root): tools/ab_multi.sh "C2:500 C3:500" "EXTRA flags A" "EXTRA flags B" ...
# Like ab_variants.sh, but builds each variant ONCE and times every workload of the first argument with it (tools/quick_time.py,
# best of its 3 frames).  A variant may carry environment settings in front of a '|'.
wls=$1; shift 1
rm -rf /tmp/repo_ab && mkdir -p /tmp/repo_ab && cp -r include raymond_amd /tmp/repo_ab/
for v in "$@"; do
  envs=""; extra="$v"
  if [[ "$v" == *"|"* ]]; then envs="${v%%|*}"; extra="${v#*|}"; fi
  make -s -C /tmp/repo_ab/raymond_amd/csrc clean
  make -s -j16 -C /tmp/repo_ab/raymond_amd/csrc EXTRA="$extra" libraymond_hip.so 2>&1 | grep -E "error" || true
  for wl in $wls; do
    w=${wl%%:*}; spp=${wl##*:}
    best=$(env $envs RAYMOND_HIP_LIB=/tmp/repo_ab/raymond_amd/csrc/libraymond_hip.so timeout -k 10 300 python3 tools/quick_time.py $w $spp | grep kernel | sed -E 's/.*kernel ([0-9.]+) ms.*/\1/' | sort -n | head -1)
    echo "== env '$envs' EXTRA '$extra' $w spp=$spp best kernel ms: $best"
  done
done

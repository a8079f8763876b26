#!/bin/bash
# usage (GPU box, repo root): plain sample stores + agent-scope release against write-through stores, same box: full C2 frame at several item sizes and the N = 8 tile share
rm -rf /tmp/repo_ab && mkdir -p /tmp/repo_ab && cp -r include raymond_amd /tmp/repo_ab/
for v in "-DRMD_SAMPLE_STORE_WT=0" "-DRMD_SAMPLE_STORE_WT=1"; do
  make -s -C /tmp/repo_ab/raymond_amd/csrc clean; make -s -j16 -C /tmp/repo_ab/raymond_amd/csrc EXTRA="$v" libraymond_hip.so 2>&1 | grep -E "error" || true
  for k in 0 4 7 12; do
    best=$(RMD_SAMPLE_SPLIT=$k RAYMOND_HIP_LIB=/tmp/repo_ab/raymond_amd/csrc/libraymond_hip.so python3 tools/quick_time.py C2 500 | grep kernel | sed -E 's/.*kernel ([0-9.]+) ms.*/\1/' | sort -n | head -1)
    echo "EXTRA '$v' split $k: $best ms"
  done
  RAYMOND_HIP_LIB=/tmp/repo_ab/raymond_amd/csrc/libraymond_hip.so python3 tools/shard_split.py C2 8 | sed "s/^/EXTRA '$v' /"
done

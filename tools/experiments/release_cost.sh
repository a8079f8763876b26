#!/bin/bash
# usage (GPU box, repo root): what the per-item release of the spheres kernel costs — the library built with RMD_EXPERIMENT_NO_RELEASE (timing only:
# frames may be wrong across XCDs) against the shipped one, at several item sizes
rm -rf /tmp/repo_ab && mkdir -p /tmp/repo_ab && cp -r include raymond_amd /tmp/repo_ab/
for v in "" "-DRMD_EXPERIMENT_NO_RELEASE=1"; do
  make -s -C /tmp/repo_ab/raymond_amd/csrc clean; make -s -j16 -C /tmp/repo_ab/raymond_amd/csrc EXTRA="$v" libraymond_hip.so 2>&1 | grep -E "error" || true
  for k in 4 9 16; do
    best=$(RMD_SAMPLE_SPLIT=$k RAYMOND_HIP_LIB=/tmp/repo_ab/raymond_amd/csrc/libraymond_hip.so python3 tools/quick_time.py C2 500 | grep kernel | sed -E 's/.*kernel ([0-9.]+) ms.*/\1/' | sort -n | head -1)
    echo "EXTRA '$v' split $k: $best ms"
  done
done

#!/bin/bash
# usage (GPU box, repo root): tools/test_variants.sh "EXTRA flags A" "EXTRA flags B" ...
# Builds a full copy of the repo per variant under /tmp (compile-time fallbacks such as -DRMD_WALK_ASM_LOOP=0, -DRMD_WALK_ASM_STEP=0,
# -DRMD_TRIP_RELOAD=0) and runs the GPU parity tests against it: every variant must give the oracle's results like the shipped build.
for v in "$@"; do
  rm -rf /tmp/repo_var && mkdir -p /tmp/repo_var && cp -r include raymond_amd oracle tests tools __graft_entry__.py bench.py /tmp/repo_var/ 2>/dev/null
  make -s -C /tmp/repo_var/raymond_amd/csrc clean
  make -s -j8 -C /tmp/repo_var/raymond_amd/csrc EXTRA="$v" 2>&1 | grep -E "error" || true
  echo "== EXTRA '$v'"
  (cd /tmp/repo_var && timeout -k 10 600 python3 -m pytest tests/test_gpu_golden.py tests/test_gpu_parity.py tests/test_gpu_reference_pins.py -x -q -m gpu 2>&1 | tail -2)
done

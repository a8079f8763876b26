#!/bin/bash
# usage (GPU box, repo root): tools/queue_sweep.sh [WORKLOAD SPP] — scheduling knobs of the queued mesh kernel, one quick_time run each (no rebuild: environment tunables)
wl=${1:-C3}; spp=${2:-200}
run() { best=$(env "$@" timeout -k 10 120 python3 tools/quick_time.py $wl $spp | grep kernel | sed -E 's/.*kernel ([0-9.]+) ms.*/\1/' | sort -n | head -1); echo "$* -> $best ms"; }
run RMD_NOOP=1
for k in 1 3 7 9 13 17 25; do run RMD_WALK_CUT=$k; done
for m in 2 4 8 16 32 64; do run RMD_SPLIT_MIN_SAMPLES=$m; done
for s in 8 16 24 48 64; do run RMD_SAMPLE_SPLIT=$s; done

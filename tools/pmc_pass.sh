#!/bin/bash
# usage: tools/pmc_pass.sh OUTDIR WORKLOAD SPP "COUNTER LIST" — one rocprofv3 --pmc pass over tools/quick_time.py
set -e
out=$1; wl=$2; spp=$3; ctrs=$4
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $ctrs --output-format csv -d "$out" -- python3 "$GRAFT_REPO_ROOT/tools/quick_time.py" "$wl" "$spp" > "$out.log" 2>&1

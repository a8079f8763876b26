#!/bin/bash
# usage: tools/ab_variants.sh WORKLOAD SPP "EXTRA flags A" "EXTRA flags B" ... — rebuilds the library per variant (on the GPU box) and times it
wl=$1; spp=$2; shift 2
for extra in "$@"; do
  touch raymond_amd/csrc/device_core.hpp raymond_amd/csrc/grid_walk.hpp raymond_amd/csrc/kernels.hip
  make -s -C raymond_amd/csrc EXTRA="$extra" -j8 2>&1 | grep -E "error" 
  echo "== EXTRA='$extra'"
  python tools/quick_time.py $wl $spp | sed -n 2,3p
done

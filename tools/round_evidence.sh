#!/bin/bash
# usage (GPU box, one gpurun call): tools/round_evidence.sh ROUND_TAG PROFILE_NAME   e.g. tools/round_evidence.sh r05 r05_v1
# One call collects what DESIGN.md / README.md quote for a round, at the sources of the snapshot: the -m gpu suite (with the off-pixel accounting of
# the whole-frame tests), the 1,000,000-sample parity campaign, the pixels that differ between the black-path modes at full sample counts, the
# one-GPU scaling proxy for every configuration BASELINE.json names, the host-API timings and tools/profile_round.sh (kernel trace, PMC passes, bench line).
set -e
cd $GRAFT_REPO_ROOT
tag=$1; prof=$2
out=gpurun_out/${tag}_final
mkdir -p $out
rm -f gpurun_out/off_pixels.jsonl
# (unbuffered and verbose into a file: a quiet call is taken to be hung after seven minutes)
PYTHONUNBUFFERED=1 timeout -k 10 900 python -u -m pytest tests -x -v -m gpu > $out/pytest_gpu_full.log 2>&1
tail -3 $out/pytest_gpu_full.log | tee $out/pytest_gpu.log
cp gpurun_out/off_pixels.jsonl $out/off_pixels.jsonl 2>/dev/null || true
timeout -k 10 400 python tools/parity_campaign.py 200000 2>&1 | tee $out/parity_campaign.txt
timeout -k 10 400 python tools/count_mode_diffs.py $out/black_path_pixel_counts.json 2>&1 | tee $out/black_path_pixel_counts.log
(timeout -k 10 200 python tools/shard_proxy.py C2 500; timeout -k 10 300 python tools/shard_proxy.py C3 500; timeout -k 10 300 python tools/shard_proxy.py C4 100; timeout -k 10 300 python tools/shard_proxy.py C5 200) 2>&1 | tee $out/shard_proxy.txt
for a in "spheres 1920 1080 500 5 0" "spheres 1920 1080 500 5 8" "dragon 1920 1080 500 5 0" "dragon 1920 1080 500 5 8" "dragon 1920 1080 500 5 16"; do timeout -k 10 200 raymond_amd/host/raymond_cli hostapi $a; done 2>&1 | tee $out/host_api.jsonl
# (the profile round is a call of its own — gpurun's limit is 1200 s a call: `tools/profile_round.sh $prof`; pass a third argument to run it here anyway)
if [ -n "$3" ]; then timeout -k 10 900 tools/profile_round.sh $prof 2>&1 | tail -1 | cut -c1-200; fi

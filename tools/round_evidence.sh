set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04_final
timeout -k 10 600 python -m pytest tests -x -q -m gpu 2>&1 | tail -3 | tee gpurun_out/r04_final/pytest_gpu.log
timeout -k 10 400 python tools/parity_campaign.py 200000 2>&1 | tee gpurun_out/r04_final/parity_campaign.txt
timeout -k 10 400 python tools/count_mode_diffs.py gpurun_out/r04_final/black_path_pixel_counts.json 2>&1 | tee gpurun_out/r04_final/black_path_pixel_counts.log
(timeout -k 10 200 python tools/shard_proxy.py C2 500; timeout -k 10 300 python tools/shard_proxy.py C3 500) 2>&1 | tee gpurun_out/r04_final/shard_proxy.txt
timeout -k 10 600 tools/profile_round.sh r04_v9 2>&1 | tail -1 | cut -c1-200

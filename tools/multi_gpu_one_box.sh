#!/bin/bash
# usage (GPU box, from the repo root): tools/multi_gpu_one_box.sh > gpurun_out/multi_gpu_one_box.txt
# What one GPU can measure of the 8-rank job: the rebuild kernel alone, every rank's share of chair 4K
# under the even deal, and rank 0 (render + rebuild) against rank 1 under the candidate deals, 4K and 8K.
# Launches of 32 frames, three contexts taking launches in turn.  Each figure is one short run: +-3 %.
set -e
export SLICE_BATCH=32 SLICE_K=3
run() { timeout -k 10 120 python3 tools/slice_rate.py "$@" 2>/dev/null | tr '\n' ' ' | sed -E 's/GPU_MAX_HW_QUEUES=[a-z0-9]+  //; s/ +\([0-9.]+ us of host time per launch\)//'; echo; }
echo "## rm_assemble_luminance_batch alone (tools/rebuild_rate.py)"
timeout -k 10 120 python3 tools/rebuild_rate.py 2>/dev/null
echo; echo "## one GPU whole frames, then the whole frame through the band path, then rank 0's share of 2, 4, 8 ranks"
run 1 chair 3840 2160
SLICE_WEIGHTS=1,0 run 2 chair 3840 2160
run 2 chair 3840 2160
run 4 chair 3840 2160
echo; echo "## every rank's share, even deal of 8-row bands, chair 4K"
for r in 0 1 2 3 4 5 6 7; do SLICE_RANK=$r run 8 chair 3840 2160; done
for size in "3840 2160" "7680 4320"; do
  echo; echo "## rank 0 (render + rebuild of every frame) and rank 1 under candidate deals, $size"
  for wts in 1,1,1,1,1,1,1,1 3,5,5,5,5,5,5,5 1,2,2,2,2,2,2,2 1,4,4,4,4,4,4,4 1,8,8,8,8,8,8,8; do
    SLICE_WEIGHTS=$wts SLICE_RANK=0 SLICE_REBUILD=1 run 8 chair $size
    SLICE_WEIGHTS=$wts SLICE_RANK=1 run 8 chair $size
  done
done

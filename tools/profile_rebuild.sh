#!/bin/bash
# usage (GPU box, repo root): tools/profile_rebuild.sh TAG [WORKLOAD]
# rank 0's side of the N>1 path on one GPU (RM_BENCH_SOLO_GATHER=1: bands, luminance planes, three contexts in turn,
# an RCCL gather of the one part to itself, rm_assemble_luminance_batch rebuilding every frame): a kernel trace and
# the two HBM counter passes, each a rocprofv3 run of its own with the program straight after `--`; summarised by
# tools/rebuild_summary.py into gpurun_out/profiles_TAG/TAG.txt
tag=$1
wl=${2:-chair_4k}
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export RM_BENCH_SOLO_GATHER=1
set -e
timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d $root/gpurun_out/${tag}_trace -o t -- python3 $root/bench.py --gpus 1 --workload $wl --steps 64 --warmup 32 --repeats 3 --no-cpu-baseline > $root/gpurun_out/${tag}_trace_bench.json
for pass in FETCH_SIZE WRITE_SIZE; do
    timeout -s KILL 200 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $root/gpurun_out/${tag}_pmc_$pass -o pmc -- python3 $root/bench.py --gpus 1 --workload $wl --steps 64 --warmup 32 --warmup-seconds 0 --repeats 1 --no-cpu-baseline > /dev/null
done
cd $root && python3 tools/rebuild_summary.py $tag $wl --out gpurun_out/profiles_$tag > /dev/null

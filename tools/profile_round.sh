#!/bin/bash
# usage (GPU box, from the repo root): tools/profile_round.sh TAG [WORKLOAD]
# kernel trace + four counter passes of bench.py (whole launches of up to 32 frames), each its own rocprofv3 run;
# raw output under gpurun_out/TAG_*, summarised here (the box has the data) into profiles/-shaped files under
# gpurun_out/profiles_TAG/ by tools/profile_summary.py
tag=$1
wl=${2:-chair_4k}
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
set -e
timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d $root/gpurun_out/${tag}_trace -o t -- python3 $root/bench.py --workload $wl --steps 96 --warmup 16 --repeats 3 --no-cpu-baseline --no-single-frame > $root/gpurun_out/${tag}_trace_bench.json
for pass in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" \
            "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_TRANS SQ_INSTS_BRANCH SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE GRBM_COUNT" \
            "FETCH_SIZE" "WRITE_SIZE"; do
    name=$(echo $pass | cut -d' ' -f1)
    timeout -s KILL 150 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $root/gpurun_out/${tag}_pmc_$name -o pmc -- python3 $root/bench.py --workload $wl --steps 64 --warmup 32 --warmup-seconds 0 --repeats 1 --no-cpu-baseline --no-single-frame > /dev/null
done
cd $root && python3 tools/profile_summary.py $tag $wl --out gpurun_out/profiles_$tag > /dev/null

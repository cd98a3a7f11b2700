#!/bin/bash
# usage (GPU box, repo root): tools/pmc.sh TAG PROGRAM ARGS...   - two counter passes of `python3 PROGRAM ARGS`,
# per-kernel sums printed by tools/pmc_summary.py
tag=$1; shift
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for pass in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" \
            "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_INSTS_LDS GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD"; do
    name=$(echo $pass | cut -d' ' -f1)
    timeout -s KILL 150 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $root/gpurun_out/${tag}_pmc_$name -o pmc -- python3 "$@" > /dev/null || exit 1
done
cd $root && python3 tools/pmc_summary.py $tag

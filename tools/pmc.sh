#!/bin/bash
# usage: tools/pmc.sh OUTDIR "COUNTERS..." -- program args   (one counter pass; run on the GPU box)
out=$1; shift; ctrs=$1; shift; shift
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -s KILL 120 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d "$out" -o pmc -- "$@"

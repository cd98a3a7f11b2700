#!/bin/bash
# usage (GPU box, repo root): tools/gpu_steps.sh NAME:SECONDS:COMMAND...   (one quoted argument per step)
# Runs the steps in order, each under its own time limit, stdout+stderr to gpurun_out/NAME.log.  An ordinary
# failure (exit 1: a failed assertion, a bench that raised) is recorded and the next step runs; a time limit,
# a kill, an abort or a segmentation fault (124, 137, 134, 139) ends the script there: nothing more goes to
# the GPU after a run that may have left it in a bad state.
mkdir -p gpurun_out
for step in "$@"; do
    name=${step%%:*}; rest=${step#*:}; secs=${rest%%:*}; cmd=${rest#*:}
    echo "== $name (limit ${secs}s)"
    timeout -k 10 "$secs" bash -c "$cmd" > "gpurun_out/$name.log" 2>&1
    rc=$?
    echo "== $name rc=$rc"
    tail -n 4 "gpurun_out/$name.log"
    case $rc in 124|137|134|139|135|136) echo "stopping: rc=$rc"; exit $rc;; esac
done
exit 0

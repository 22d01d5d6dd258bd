#!/bin/bash
# tools/g.sh <log name> <command...>: run a command on the GPU box with its output kept under gpurun_out/r05/ (the directory does not exist on a fresh box)
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r05
LOG=gpurun_out/r05/$1.log; shift
"$@" > $LOG 2>&1; rc=$?
tail -${G_TAIL:-60} $LOG
exit $rc

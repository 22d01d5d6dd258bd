#!/bin/bash
# tools/gq.sh <timeout> <command for the GPU box>: gpurun, retried while no box / slot is free (exit code 3: nothing was charged)
T=$1; shift
for i in 1 2 3 4 5 6 7 8 9 10 11 12; do
  /usr/local/graft/bin/gpurun --timeout $T -- "$@"; rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 45
done
exit 3

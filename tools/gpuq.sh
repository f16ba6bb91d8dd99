#!/bin/bash
# Submit one gpurun call; when no GPU slot / box is free (exit code 3: nothing ran, nothing charged) wait and submit again.
# usage: tools/gpuq.sh TIMEOUT 'command'
T=$1; shift
for i in $(seq 1 40); do
    /usr/local/graft/bin/gpurun --timeout "$T" -- "$@"
    rc=$?
    [ $rc -ne 3 ] && exit $rc
    sleep 120
done
exit 3

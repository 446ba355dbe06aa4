#!/bin/bash
# retry a gpurun call while the pod has no free GPU slot (exit code 3: nothing charged); usage: tools/gpurun_retry.sh <timeout_s> '<command>'
T=$1; shift
for i in $(seq 1 40); do
    /usr/local/graft/bin/gpurun --timeout "$T" -- "$@"
    rc=$?
    if [ $rc -ne 3 ]; then exit $rc; fi
    echo "[retry] no slot (attempt $i), sleeping 90 s"
    sleep 90
done
exit 3

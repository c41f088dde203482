#!/bin/bash
# gpurun with a wait for a free GPU slot: retries ONLY the "no box or slot free" answer (exit code 3: nothing ran, nothing
# charged), never a command that ran and failed.  Usage: tools/gpurun_when_free.sh <timeout-seconds> '<command>'
T=$1; shift
for i in $(seq 1 30); do
  /usr/local/graft/bin/gpurun --timeout "$T" -- "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  echo "[gpurun_when_free] no slot free (try $i), waiting 90 s" >&2
  sleep 90
done
exit 3

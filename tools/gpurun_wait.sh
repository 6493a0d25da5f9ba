#!/bin/bash
# dev aid: gpurun, waiting while no GPU slot is free (exit code 3 = nothing ran, nothing was charged).  Never retries a run that started.
for i in $(seq 1 20); do
  /usr/local/graft/bin/gpurun "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 90
done
exit 3

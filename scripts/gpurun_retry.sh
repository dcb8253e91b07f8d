#!/bin/bash
# gpurun with a wait for a free slot: exit code 3 means "no box or slot free, nothing charged" -- wait and ask again.
# (Never a retry of a GPU command that ran: any other exit code ends the loop.)
LOG=$1; shift
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun "$@" > "$LOG" 2>&1
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 60
done
exit 3

#!/bin/bash
# Run GPU-box steps one after the other: usage  tools/run_steps.sh <log dir> "<seconds> <name> <command...>" ...
# An ordinary failure (non-zero exit) is logged and the next step still runs; a step that TIMES OUT or is KILLED ends the
# call (no further GPU step after a hang).  Each step's stdout+stderr -> <log dir>/<name>.log
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
D=$1; shift; mkdir -p "$D"
for spec in "$@"; do
  set -- $spec; secs=$1; name=$2; shift 2
  echo "[$(date +%H:%M:%S)] $name: $*" | tee -a "$D/steps.log"
  timeout -k 10 "$secs" bash -c "$*" > "$D/$name.log" 2>&1; rc=$?
  echo "[$(date +%H:%M:%S)] $name rc=$rc" | tee -a "$D/steps.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step $name timed out / was killed: stopping" | tee -a "$D/steps.log"; exit 1; fi
done
exit 0

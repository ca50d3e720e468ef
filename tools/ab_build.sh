#!/bin/bash
# Same-box A/B of BUILD-FLAG variants of libpfpn_hip.so: usage  tools/ab_build.sh <out dir> "<probe command>" "<flags A>" "<flags B>" ...
# ("" = the default build).  The variants take turns for ROUNDS rounds (default 3: a box drifts by more than most effects, so
# only alternating runs inside one session compare).  Leaves the DEFAULT library behind.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=$1; CMD=$2; shift 2
mkdir -p "$O"; : > "$O/ab.log"
trap 'rm -rf pfpn_amd/build && python3 -m pfpn_amd.build > /dev/null 2>&1' EXIT
for r in $(seq 1 ${ROUNDS:-3}); do
  for f in "$@"; do
    rm -rf pfpn_amd/build && PFPN_EXTRA_FLAGS="$f" python3 -m pfpn_amd.build > "$O/build.log" 2>&1 || { echo "build failed [$f]" >> "$O/ab.log"; continue; }
    echo "round $r [$f]:" >> "$O/ab.log"
    timeout -k 10 300 bash -c "$CMD" 2>&1 | grep -v amdgpu.ids >> "$O/ab.log"
  done
done
cat "$O/ab.log"

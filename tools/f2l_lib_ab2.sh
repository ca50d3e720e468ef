#!/bin/bash
# tools/f2l_lib_ab.sh over several shapes:  tools/f2l_lib_ab2.sh "<probe args>;<probe args>;..." tag tag ...
IFS=';' read -ra SH <<< "$1"; shift
for a in "${SH[@]}"; do echo "=== $a"; PROBE_ARGS="$a" ROUNDS=${ROUNDS:-2} bash tools/f2l_lib_ab.sh "$@" | grep -E "round|2L fused"; done

#!/bin/bash
# Same-box A/B of prebuilt libraries (tools/bin/libpfpn_hip_<tag>.so, built here before the call): they take turns as
# pfpn_amd/libpfpn_hip.so for ROUNDS rounds; the default library is put back at the end.   usage: tools/f2l_lib_ab.sh v12 v14 ...
cd "$GRAFT_REPO_ROOT"
cp pfpn_amd/libpfpn_hip.so /tmp/libpfpn_default.so
trap 'cp /tmp/libpfpn_default.so pfpn_amd/libpfpn_hip.so' EXIT
for r in $(seq 1 ${ROUNDS:-3}); do
  for t in "$@"; do
    cp tools/bin/libpfpn_hip_$t.so pfpn_amd/libpfpn_hip.so
    echo "round $r [$t]:"
    timeout -k 10 300 python3 tools/f2l_probe.py ${PROBE_ARGS} 2>&1 | grep -E 'us |parity'
  done
done

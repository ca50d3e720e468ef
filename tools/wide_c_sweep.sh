#!/bin/bash
# the wide backward against its chunk count (PFPN_WIDE_C), alternating, ROUNDS rounds: tools/wide_c_sweep.sh "<build flags>" C1 C2 ...
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
F=$1; shift
rm -rf pfpn_amd/build && PFPN_EXTRA_FLAGS="$F" python3 -m pfpn_amd.build > /dev/null 2>&1 || exit 1
trap 'rm -rf pfpn_amd/build && python3 -m pfpn_amd.build > /dev/null 2>&1' EXIT
for r in $(seq 1 ${ROUNDS:-2}); do for c in "$@"; do echo -n "round $r C=$c: "; PFPN_WIDE_C=$c python3 tools/bwd_probe.py 2>&1 | grep "^{"; done; done

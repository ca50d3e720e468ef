#!/bin/bash
# step-level A/B of build-flag sets: bench.py headline (no extras) with each flag set given as arguments 2.., same box
O=gpurun_out/${1:-stepab}; shift
mkdir -p $O; : > $O/out.log
for f in "$@"; do
  rm -rf pfpn_amd/build && PFPN_EXTRA_FLAGS="$f" python -m pfpn_amd.build > $O/build.log 2>&1
  for i in 1 2; do
    python bench.py --no-cpu-baseline --steps 400 > $O/b.log 2>&1
    python - "$f" $O/b.log >> $O/out.log <<'PY'
import json, sys
for l in open(sys.argv[2]):
    if l.startswith("{"):
        d = json.loads(l); print("[%-30s] %.1f steps/s  %.2f us/step" % (sys.argv[1], d["value"], d["ms_per_step"] * 1e3))
PY
  done
done
cat $O/out.log

# leave the DEFAULT library behind: later bench / test runs must not silently measure a diagnostic or flagged build
rm -rf pfpn_amd/build && python3 -m pfpn_amd.build > /dev/null 2>&1

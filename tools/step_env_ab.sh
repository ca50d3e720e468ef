#!/bin/bash
# step-level A/B of environment switches: bench.py headline (no extras) under each "VAR=val ..." argument, same box
O=gpurun_out/${1:-envab}; shift
mkdir -p $O; : > $O/out.log
for e in "$@"; do
  for i in 1 2; do
    env $e python bench.py --no-cpu-baseline --steps 400 > $O/b.log 2>&1
    python - "$e" $O/b.log >> $O/out.log <<'PY'
import json, sys
for l in open(sys.argv[2]):
    if l.startswith("{"):
        d = json.loads(l); print("%-40s %.1f steps/s  %.2f us/step" % (sys.argv[1], d["value"], d["ms_per_step"] * 1e3))
PY
  done
done
cat $O/out.log

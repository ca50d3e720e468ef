#!/bin/bash
# step-level A/B of a build flag set: bench.py (no CPU baseline) with the default build and with $2
O=gpurun_out/${1:-stepab}
mkdir -p $O
python bench.py --no-cpu-baseline > $O/bench_default.log 2>&1
rm -rf pfpn_amd/build && PFPN_EXTRA_FLAGS="$2" python -m pfpn_amd.build > $O/build.log 2>&1 && python bench.py --no-cpu-baseline > $O/bench_flag.log 2>&1
python - <<PY
import json
for n in ("default","flag"):
    for l in open("$O/bench_%s.log" % n):
        if l.startswith("{"):
            d=json.loads(l); print(n, d["value"], d["ms_per_step"], d["roofline"]["us_per_launch"], d["roofline"]["frac"], d["roofline"].get("us_in_step_event_pairs"))
PY

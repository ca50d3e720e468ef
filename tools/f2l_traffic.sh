#!/bin/bash
# HBM traffic of the one-launch PPO pass (k_mix_wide_f2l) on the 8192 x 128 x 256 shape: FETCH_SIZE and WRITE_SIZE in separate
# rocprofv3 --pmc passes of tools/stress_probe.py (MI355X_MICROARCH.md: FETCH_SIZE doubled on gfx950; KB -> bytes).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/f2l_pmc; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/sfetch -- python3 tools/stress_probe.py > $O/sfetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/swrite -- python3 tools/stress_probe.py > $O/swrite.log 2>&1
python3 - <<'PY'
import csv, glob, json
def mean_counter(d, name):
    vals = {}
    for f in glob.glob("gpurun_out/f2l_pmc/%s/**/*counter_collection.csv" % d, recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name:
                k = r["Kernel_Name"]
                key = "k_mix_wide_f2l" if "k_mix_wide_f2l<" in k else ("k_mix_wide<BWD>" if "k_mix_wide<1, 1" in k else
                                                                          "k_mix_wide<FWD>" if "k_mix_wide<1, 0" in k else None)
                if key:
                    vals.setdefault(key, []).append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in vals.items()}
f, w = mean_counter("sfetch", "FETCH_SIZE"), mean_counter("swrite", "WRITE_SIZE")
out = {}
for k in sorted(set(f) | set(w)):
    rb = f.get(k, (0, 0))[0] * 1024 * 2
    wb = w.get(k, (0, 0))[0] * 1024
    out[k] = {"read_bytes": rb, "write_bytes": wb, "traffic_bytes": rb + wb, "launches": f.get(k, (0, 0))[1]}
print(json.dumps(out, indent=1))
json.dump(out, open("gpurun_out/f2l_pmc/traffic.json", "w"), indent=1)
PY
rm -rf $O/sfetch $O/swrite

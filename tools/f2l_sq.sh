#!/bin/bash
# SQ counters of the wide mixture kernels on the 8192 x 128 x 256 shape (forward, backward, the one-launch PPO pass): two
# rocprofv3 --pmc passes of tools/stress_probe.py, per-launch means -> gpurun_out/f2l_sq.json
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/f2l_sq; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_ANY --output-format csv -d $O/sq1 -- python3 tools/stress_probe.py > $O/sq1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC --output-format csv -d $O/sq2 -- python3 tools/stress_probe.py > $O/sq2.log 2>&1
python3 - <<'PY'
import csv, glob, json
acc = {}
for d in ("sq1", "sq2"):
    for f in glob.glob("gpurun_out/f2l_sq/%s/**/*counter_collection.csv" % d, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            key = "k_mix_wide_f2l" if "k_mix_wide_f2l<" in k else ("k_mix_wide<BWD>" if "k_mix_wide<1, 1" in k else
                                                                      "k_mix_wide<FWD>" if "k_mix_wide<1, 0" in k else None)
            if key:
                acc.setdefault(key, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
out = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}
for k, c in out.items():
    w = c.get("SQ_WAVES", 0)
    if w and c.get("SQ_WAVE_CYCLES"):
        c["derived"] = {"valu_instructions_per_wave": c.get("SQ_INSTS_VALU", 0) / w,
                        "trans_instructions_per_wave": c.get("SQ_INSTS_VALU_TRANS", 0) / w,
                        "salu_instructions_per_wave": c.get("SQ_INSTS_SALU", 0) / w,
                        # (SQ_WAVE_CYCLES, SQ_ACTIVE_INST_* and SQ_WAIT_* all count 4-clock quads on gfx950: wave cycles x 4 / clock
                        # = the kernel's duration)
                        "valu_active_share_of_wave_cycles": c.get("SQ_ACTIVE_INST_VALU", 0) / c["SQ_WAVE_CYCLES"],
                        "wait_any_share_of_wave_cycles": c.get("SQ_WAIT_ANY", 0) / c["SQ_WAVE_CYCLES"]}
print(json.dumps(out, indent=1))
json.dump(out, open("gpurun_out/f2l_sq.json", "w"), indent=1)
PY
rm -rf $O/sq1 $O/sq2

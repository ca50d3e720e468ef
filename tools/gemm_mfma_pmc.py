#!/usr/bin/env python3
"""MFMA-busy share of the GEMM kernels that ARE the learner step, from one `rocprofv3 --pmc` pass over the headline command.

  cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv \
            -d gpurun_out/gemm_mfma -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline
  python3 tools/gemm_mfma_pmc.py gpurun_out/gemm_mfma profiles/r04_gemm_mfma.json

Units (MI355X_MICROARCH.md, constants table): SQ_VALU_MFMA_BUSY_CYCLES counts shader cycles summed over the chip's 1024
SIMDs (= 64 x N for `v_mfma_f32_32x32x2_f32`, 32 x N for `v_mfma_f32_16x16x4_f32`: checked below against SQ_INSTS_MFMA);
SQ_BUSY_CYCLES is summed over the 32 SQ instances (8 XCDs x 4 shader engines); GRBM_GUI_ACTIVE over the 8 XCDs.
mfma_busy = (MFMA busy cycles / 1024 SIMDs) / (kernel cycles), kernel cycles = SQ_BUSY_CYCLES / 32.
A counter pass serialises kernels and adds set-up time around each, so durations here are NOT the step's; the share is.
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

raw, out = sys.argv[1:3]
MIN_LAUNCHES = int(sys.argv[3]) if len(sys.argv) > 3 else 100
files = sorted(glob.glob(os.path.join(raw, "**", "*counter_collection.csv"), recursive=True))
acc = defaultdict(lambda: defaultdict(list))
dur = defaultdict(list)
seen = set()
for f in files:
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if not k.startswith("Cijk_") and "k_mix_tile" not in k:
            continue
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        key = (f, r["Dispatch_Id"])
        if key not in seen:
            seen.add(key)
            dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)


def mean(v):
    return sum(v) / max(1, len(v))


kernels = {}
for k, cs in acc.items():
    c = {n: mean(v) for n, v in cs.items()}
    launches = len(dur[k])
    mf, busy, insts, gui = (c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), c.get("SQ_BUSY_CYCLES", 0.0),
                            c.get("SQ_INSTS_MFMA", 0.0), c.get("GRBM_GUI_ACTIVE", 0.0))
    mi = re.search(r"_MI(\d+)x(\d+)x(\d+)", k)
    e = {"launches": launches, "counters_per_launch": c, "us_per_launch_in_the_counter_pass": mean(dur[k])}
    if insts:
        e["mfma_cycles_per_inst"] = mf / insts
        e["mfma_busy_cycles_per_simd"] = mf / 1024.0
        if busy:
            e["kernel_cycles_sq_busy"] = busy / 32.0
            e["mfma_busy"] = (mf / 1024.0) / (busy / 32.0)
        if gui:
            e["kernel_cycles_gui_active"] = gui / 8.0
            e["mfma_busy_of_gui_active"] = (mf / 1024.0) / (gui / 8.0)
        if mi:
            e["mfma_tile_in_kernel_name"] = "MI%sx%sx%s" % mi.groups()
        # every fp32 MFMA delivers 64 flop per busy cycle of its SIMD (32x32x2: 4096 flop / 64 cycles, 16x16x4: 2048 / 32)
        e["flop_from_mfma_busy_cycles"] = mf * 64.0
    if launches >= MIN_LAUNCHES:               # (TunableOp's candidates in the warm-up run ~40 times each: not the step's kernels)
        kernels[re.sub(r"\(.*", "", k)[:160]] = e

order = sorted(kernels, key=lambda n: -kernels[n].get("mfma_busy_cycles_per_simd", 0) * kernels[n]["launches"])
tot_mf = sum(kernels[n].get("mfma_busy_cycles_per_simd", 0) * kernels[n]["launches"] for n in order)
tot_cy = sum(kernels[n].get("kernel_cycles_sq_busy", 0) * kernels[n]["launches"] for n in order if "mfma_busy" in kernels[n])
res = {"note": __doc__.strip().split("\n\n")[-1].replace("\n", " "),
       "command": "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE -- "
                  "python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline",
       "mfma_busy_all_gemm_kernels": tot_mf / tot_cy if tot_cy else None,
       "kernels": {n: kernels[n] for n in order}}
json.dump(res, open(out, "w"), indent=1)
for n in order:
    e = kernels[n]
    print("%5d x  mfma_busy %s  (%s, %.1f cyc/inst)  %s" % (e["launches"], ("%.3f" % e["mfma_busy"]) if "mfma_busy" in e else "  -  ",
                                                            e.get("mfma_tile_in_kernel_name", "-"), e.get("mfma_cycles_per_inst", 0.0), n[:110]))
print("all GEMM kernels, launch-weighted:", res["mfma_busy_all_gemm_kernels"])

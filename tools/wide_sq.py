"""VALU-issue share of the wide mixture kernels from a rocprofv3 --pmc pass over tools/stress_probe.py.
usage: python tools/wide_sq.py <counter_collection.csv>"""
import csv, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "k_mix_wide" not in k:
        continue
    name = "fwd" if "<1, 0" in k or ", 0, " in k.split("k_mix_wide")[1][:12] else "bwd"
    acc[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in acc.items():
    m = {n: sum(v) / len(v) for n, v in c.items()}
    print(k)
    for n, v in sorted(m.items()):
        print("   %-24s %.4g" % (n, v))
    if "SQ_ACTIVE_INST_VALU" in m and "SQ_BUSY_CYCLES" in m:
        # quad-cycles -> cycles; 1024 SIMDs
        print("   VALU issue cycles per SIMD %.0f ; SQ busy cycles (per SE sum / 32 ...) %.0f" % (m["SQ_ACTIVE_INST_VALU"] * 4 / 1024, m["SQ_BUSY_CYCLES"]))
    if "SQ_ACTIVE_INST_VALU" in m and "SQ_WAVE_CYCLES" in m and "SQ_WAVES" in m:
        print("   VALU issue share of wave lifetime (avg) %.3f ; waves %.0f ; wave cycles per wave %.0f ; VALU insts per wave %.0f" % (
            m["SQ_ACTIVE_INST_VALU"] / m["SQ_WAVE_CYCLES"] * (m["SQ_WAVES"] / 1024 if False else 1), m["SQ_WAVES"], m["SQ_WAVE_CYCLES"] * 4 / m["SQ_WAVES"],
            m.get("SQ_INSTS_VALU", 0) / m["SQ_WAVES"]))

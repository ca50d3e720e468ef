"""Turn the raw rocprofv3 outputs of tools/profile_round.sh into the small files committed under profiles/:
  <tag>_bench_kernel_stats.csv  per-kernel stats table of the eager step
  <tag>_step_timeline.txt       one learner step of the default run, kernel by kernel
  <tag>_pmc_traffic.json        HBM bytes per launch of the tile / wide mixture kernels (FETCH_SIZE x2 + WRITE_SIZE)
  <tag>_mix_tile_sq.json        SQ counters of k_mix_tile<35,PPO>: the measured VALU-issue share
  <tag>_mix_tile_stamps.txt     in-kernel phase stamps of block 0 (tools/stamp_probe.py, stamped build)
  <tag>_mix_probe_kernel_durations.json   rocprofv3 kernel-trace durations of the probe's mixture kernels (begin to end)
  <tag>_bench_line.json         the bench JSON line
usage: python tools/profile_collect.py <raw dir> <out dir> <tag>"""
import csv, glob, json, os, re, subprocess, sys
from collections import defaultdict

raw, out, tag = sys.argv[1:4]
os.makedirs(out, exist_ok=True)
HERE = os.path.dirname(os.path.abspath(__file__))


def one(pattern):
    fs = sorted(glob.glob(os.path.join(raw, pattern), recursive=True))
    return fs[0] if fs else None


st = one("stats/**/*kernel_stats.csv")
if st:
    open(os.path.join(out, tag + "_bench_kernel_stats.csv"), "w").write(open(st).read())
tl = one("timeline/**/*kernel_trace.csv")
if tl:
    r = subprocess.run([sys.executable, os.path.join(HERE, "step_timeline.py"), tl], stdout=subprocess.PIPE)
    open(os.path.join(out, tag + "_step_timeline.txt"), "wb").write(r.stdout)


def counters(path):
    acc = defaultdict(lambda: defaultdict(list))
    if not path:
        return acc
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        m = re.search(r"k_mix_tile<(\d+), (\d+)", k)
        name = None
        if m:
            name = "k_mix_tile<%s,%s>" % (m.group(1), {0: "FWD", 1: "BWD", 2: "PPO"}[int(m.group(2))])
        else:
            m = re.search(r"k_mix_wide<(\d+), (\d+)", k)
            if m:
                name = "k_mix_wide<%s,%s>" % (m.group(1), {0: "FWD", 1: "BWD", 2: "PPO"}[int(m.group(2))])
            elif "k_mix_wide_f2l<" in k:
                name = "k_mix_wide_f2l"                      # the one-launch (2L) PPO pass of the 256-particle shapes
        if name:
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


def mean(v):
    return sum(v) / len(v)


traffic = {"note": "rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE in separate runs) of tools/mix_probe.py (4096x36x35) and "
                   "tools/stress_probe.py (8192x128x256); FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half the "
                   "bytes of wide coalesced reads); KB -> bytes", "kernels": {}}
for f, w in (("fetch", "write"), ("sfetch", "swrite")):
    fs, ws = counters(one(f + "/**/*counter_collection.csv")), counters(one(w + "/**/*counter_collection.csv"))
    for k in sorted(set(fs) | set(ws)):
        rb = mean(fs[k]["FETCH_SIZE"]) * 1024 * 2 if "FETCH_SIZE" in fs.get(k, {}) else 0.0
        wb = mean(ws[k]["WRITE_SIZE"]) * 1024 if "WRITE_SIZE" in ws.get(k, {}) else 0.0
        traffic["kernels"][k] = {"read_bytes": rb, "write_bytes": wb, "traffic_bytes": rb + wb,
                                 "launches": len(fs.get(k, {}).get("FETCH_SIZE", []))}
json.dump(traffic, open(os.path.join(out, tag + "_pmc_traffic.json"), "w"), indent=1)

sq = {}
for d in ("sq1", "sq2"):
    for k, cs in counters(one(d + "/**/*counter_collection.csv")).items():
        sq.setdefault(k, {}).update({c: mean(v) for c, v in cs.items()})
k = "k_mix_tile<35,PPO>"
if k in sq:
    c = sq[k]
    waves, simds = c.get("SQ_WAVES", 0), 256 * 4
    note = {"kernel": k, "counters_per_launch": c,
            "derived": {
                "waves": waves,
                "valu_insts_per_wave": c.get("SQ_INSTS_VALU", 0) / max(1, waves),
                "valu_issue_cycles_per_simd": 4 * c.get("SQ_ACTIVE_INST_VALU", 0) / simds,
                "wave_cycles_per_wave": 4 * c.get("SQ_WAVE_CYCLES", 0) / max(1, waves),
                "valu_issue_share_of_wave_lifetime_avg_simd": (c.get("SQ_ACTIVE_INST_VALU", 0) / simds) /
                                                              max(1.0, c.get("SQ_WAVE_CYCLES", 0) / max(1, waves)),
                "wait_any_share": c.get("SQ_WAIT_ANY", 0) / max(1.0, c.get("SQ_WAVE_CYCLES", 1)),
            },
            "note": "SQ_ACTIVE_INST_* / SQ_WAVE_CYCLES / SQ_WAIT_* count quad-cycles (MI355X_MICROARCH.md); 1024 SIMDs; the "
                    "SIMD that hosts three of the nine compute waves carries 3 / 2.25 of the average VALU issue"}
    json.dump(note, open(os.path.join(out, tag + "_mix_tile_sq.json"), "w"), indent=1)
bl = os.path.join(raw, "bench.log")
if os.path.exists(bl):
    lines = [l for l in open(bl).read().splitlines() if l.startswith("{")]
    if lines:
        open(os.path.join(out, tag + "_bench_line.json"), "w").write(lines[-1] + "\n")
print("collected into", out, sorted(os.listdir(out)))

sp = os.path.join(raw, "stamps.txt")
if os.path.exists(sp):
    open(os.path.join(out, tag + "_mix_tile_stamps.txt"), "w").write(open(sp).read())
pt = one("probe_trace/**/*kernel_trace.csv")
if pt:
    dur = defaultdict(list)
    for r in csv.DictReader(open(pt)):
        k = r["Kernel_Name"]
        if "k_mix_tile" in k or "k_finalize" in k:
            dur[re.sub(r"\(.*", "", k.replace("void pfpn::", ""))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    json.dump({"note": "rocprofv3 --kernel-trace of tools/mix_probe.py (4096x36x35): kernel begin-to-end in us, all launches",
               "kernels": {k: {"launches": len(v), "mean_us": mean(v), "min_us": min(v), "max_us": max(v)} for k, v in dur.items()}},
              open(os.path.join(out, tag + "_mix_probe_kernel_durations.json"), "w"), indent=1)

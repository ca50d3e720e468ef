"""Timeline of ONE learner step from a rocprofv3 --kernel-trace CSV: per kernel start offset, duration and the idle gap
to the previous kernel's end.  Usage: python tools/step_timeline.py <kernel_trace.csv | results.db> [anchor-kernel-substring]
The step is delimited by consecutive launches of the anchor kernel (default: k_state_prologue)."""
import csv, sys
path = sys.argv[1]
anchor = sys.argv[2] if len(sys.argv) > 2 else "k_state_prologue"
if path.endswith(".db"):                      # rocprofv3's default rocpd database
    import sqlite3
    rows = list(sqlite3.connect(path).execute("select start, end, name from kernels order by start"))
else:                                          # --output-format csv
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(path))]
    rows.sort()
# (inside a multi-step capture the prologue of step k+1 rides in the optimiser launch of step k -- k_clip_adam_prologue:
# that launch ENDS a step, the next one starts behind it)
idx = sorted([i for i, r in enumerate(rows) if anchor in r[2]] +
             ([i + 1 for i, r in enumerate(rows) if "k_clip_adam_prologue" in r[2] and i + 1 < len(rows)]
              if anchor == "k_state_prologue" else []))
if len(idx) < 3:
    sys.exit("anchor kernel not found often enough")
# steps = spans between consecutive anchors.  A bench run holds warm-up steps, the timed graph replays and an eager
# loop of the same step (slower under the profiler): show a typical FAST step -- the 25th percentile of duration among
# the steps with the most common launch count
from collections import Counter
def rider(i):
    return 0 < i <= len(rows) and "k_clip_adam_prologue" in rows[i - 1][2]
inner = [i for i in range(len(idx) - 1) if rider(idx[i]) and rider(idx[i + 1])]      # steps inside a multi-step replay
cand = inner if (anchor == "k_state_prologue" and inner) else list(range(len(idx) - 1))
count = Counter(idx[i + 1] - idx[i] for i in cand).most_common(1)[0][0]
spans = sorted((i for i in cand if idx[i + 1] - idx[i] == count),
               key=lambda i: rows[idx[i + 1]][0] - rows[idx[i]][0])
pick = spans[int(0.25 * (len(spans) - 1))]
a, b = idx[pick], idx[pick + 1]
t0 = rows[a][0]
prev_end = t0
busy = 0
print("%8s %8s %7s  %s" % ("start_us", "dur_us", "gap_us", "kernel"))
for s, e, n in rows[a:b]:
    print("%8.1f %8.2f %7.2f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, n[:110]))
    busy += e - s
    prev_end = max(prev_end, e)
print("step %.1f us, kernels busy %.1f us, idle %.1f us, %d launches" % ((rows[b][0] - t0) / 1e3, busy / 1e3, (rows[b][0] - t0 - busy) / 1e3, b - a))

"""is the 5-9 % each kernel loses inside the alternating fwd / bwd sequence a cost of the TRANSITION (first launch after the other
kernel) or of the mix?  Patterns f f b b (pairs) and f b with a spin of ~100 us on the stream in between."""
import os, sys, json, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "ppo_compose_probe.py")).read().split("res = {}")[0])
def run(pattern, reps=12, gap=False):
    ops = {"f": fwd, "b": bwd}
    for _ in range(2):
        for ch in pattern: ops[ch]()
    torch.cuda.synchronize()
    t = {i: [] for i in range(len(pattern))}
    pad = torch.empty(1 << 22, device=dev)
    for r in range(reps):
        evs = []
        for ch in pattern:
            if gap: pad.add_(1.0)                      # a ~10-20 us elementwise kernel between the two
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); ops[ch](); e1.record(); evs.append((e0, e1))
        torch.cuda.synchronize()
        for i, (e0, e1) in enumerate(evs): t[i].append(e0.elapsed_time(e1) * 1e3)
    return {"%d:%s" % (i, pattern[i]): round(statistics.median(v), 1) for i, v in t.items()}
print(json.dumps({"fb": run("fb"), "ffbb": run("ffbb"), "fffbbb": run("fffbbb"), "fb_gap": run("fb", gap=True)}))

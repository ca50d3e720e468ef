"""fwd and bwd of the 1 GiB stress shape timed INSIDE an alternating fwd, bwd, fwd, bwd ... sequence (events between launches)"""
import os, sys, json, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "ppo_compose_probe.py")).read().split("res = {}")[0])
for _ in range(3): fwd(); bwd()
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(41)]
ev[0].record()
for i in range(20):
    fwd(); ev[2 * i + 1].record(); bwd(); ev[2 * i + 2].record()
torch.cuda.synchronize()
f = [ev[2 * i].elapsed_time(ev[2 * i + 1]) * 1e3 for i in range(20)]
b = [ev[2 * i + 1].elapsed_time(ev[2 * i + 2]) * 1e3 for i in range(20)]
alone_f, alone_b = timeit(fwd), timeit(bwd)
print(json.dumps({"fwd_in_sequence_median": statistics.median(f), "bwd_in_sequence_median": statistics.median(b),
                  "fwd_alone": alone_f, "bwd_alone": alone_b, "bwd_in_sequence": [round(v, 1) for v in b[:8]]}))

"""alternating A/B: composed PPO pass with the head inside the forward launch (2 launches) vs forward, loss head, backward (3)"""
import os, sys, json, statistics, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "ppo_compose_probe.py")).read().split("res = {}")[0])
two = lambda: kernels.ppo_mixture_fwd_bwd(logits, x, loc, logstd, lp_old, adv, zv, zv, stats, out=out)
three = lambda: (fwd(), bwd(loss()))
r2, r3 = [], []
for r in range(12):
    r3.append(timeit(three)); r2.append(timeit(two))
print(json.dumps({"three_launches": r3, "two_launches": r2, "median3": statistics.median(r3), "median2": statistics.median(r2)}))

"""The floor of a chain of dependent launches replayed from a hipGraph on this box: N trivial kernels (one float4 each)
back to back, us per launch -- what a launch-bound learner step of L launches cannot go below."""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from pfpn_amd import kernels
dev = torch.device("cuda", 0)
t, s = torch.zeros(4, device=dev), torch.ones(4, device=dev)
x = torch.zeros(1, device=dev)
for name, fn in (("pfpn_soft_update on 4 floats", lambda: kernels.soft_update(t, s, 0.5)), ("torch add_ on 1 float", lambda: x.add_(1.0))):
    for n in (16, 160):
        for _ in range(3): fn()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(n): fn()
        for _ in range(5): g.replay()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        R = 200
        for _ in range(R): g.replay()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / R
        print("%-30s %4d launches per replay: %.1f us per replay, %.2f us per launch" % (name, n, dt * 1e6, dt / n * 1e6), flush=True)

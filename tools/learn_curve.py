"""Learning curves of the assembled learner through the real launcher (tests/_learning.py), written as JSON lines.

  python tools/learn_curve.py dppo:34114:4000 dppo:34114:300:frozen sac:33406:6000 a3c:3:3000 --out gpurun_out/learn.jsonl \
      [--net resample_interval=150] [--worker lr_actor=3e-4] [--particles 128]

Every positional is family:seed:max_iterations[:frozen].  One line per run: the settings, every episode's return
(20-episode means as `curve`), every resample tick that fired with its naturally dead particles and the mass check,
the particles' distance from the initial grid.
"""
import argparse
import json
import os
import sys
import time

R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))


def kv(items):
    out = {}
    for it in items or []:
        k, v = it.split("=")
        out[k] = int(v) if v.lstrip("-").isdigit() else float(v)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("runs", nargs="+")
    ap.add_argument("--out", default=os.path.join(R, "gpurun_out", "learn_curve.jsonl"))
    ap.add_argument("--net", action="append")
    ap.add_argument("--worker", action="append")
    ap.add_argument("--group", type=int, default=20)
    ap.add_argument("--particles", type=int, default=35)     # (>= 64: the wide kernels and the one-launch PPO pass)
    args, _ = ap.parse_known_args()
    import _learning
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    for spec in args.runs:
        parts = spec.split(":")
        family, seed, iters = parts[0], int(parts[1]), int(parts[2])
        frozen = len(parts) > 3 and parts[3] == "frozen"
        t0 = time.time()
        worker, rec = _learning.launch(family, seed, iters, worker_over=kv(args.worker), net_over=kv(args.net),
                                       frozen=frozen, quiet=False, particles=args.particles)
        r = rec.returns()
        g = args.group
        line = {"run": spec, "particles": args.particles, "net": kv(args.net), "worker": kv(args.worker), "global_step": int(worker.global_step),
                "env_steps": int(worker.total_samples), "seconds": round(time.time() - t0, 1), "finite": rec.finite,
                "curve": [round(float(r[i:i + g].mean()), 2) for i in range(0, len(r) - g + 1, g)],
                "summary": rec.summary(), "ticks": rec.ticks,
                "episodes": [(e[0], e[2], round(e[3], 1)) for e in (rec.episodes if len(rec.episodes) <= 120 else
                                                                    rec.episodes[:40] + rec.episodes[-40:])],
                "grid_shift": getattr(rec, "grid_shift", None), "grid_moved": getattr(rec, "grid_moved", None)}
        print(json.dumps(line), flush=True)
        with open(args.out, "a") as f:
            f.write(json.dumps(line) + "\n")


if __name__ == "__main__":
    main()

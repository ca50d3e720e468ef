"""Round 5 probe: the clipped-PPO pass of the stress shape 8192 x 128 x 256 as ONE 2L launch
(pfpn_ppo_wide_fused: the 128 waves of a row chunk meet through memory once per four rows) against the shipped 3L composition
([forward + PPO head] -> backward, two launches).  Parity first (per-row outputs and sums against the 3L path), then time:
each form replayed from a hipGraph, HIP events around the replays.  Kill criterion of the review: the 2L form must run
forward + head + backward in < 520 us (shipped: 568-601) or the idea is recorded and dropped.

  python tools/f2l_probe.py [--rows 8192] [--dbias]          (PFPN_WIDE_C=<n> limits the chunk count of the fused form)
"""
import argparse
import os
import sys

import numpy as np
import torch

R_ = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R_)
from pfpn_amd import kernels  # noqa: E402


def graph_time(call, per_graph=10, replays=5):
    for _ in range(2):
        call()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        call()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=side):
            for _ in range(per_graph):
                call()
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(3):
        ea.record()
        for _ in range(replays):
            g.replay()
        eb.record()
        torch.cuda.synchronize()
        ts.append(ea.elapsed_time(eb) * 1e3 / (per_graph * replays))
    return min(ts), max(ts)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=8192)
    ap.add_argument("--dbias", action="store_true")
    ap.add_argument("--no-time", action="store_true")
    ap.add_argument("--dims", type=int, default=128)
    ap.add_argument("--particles", type=int, default=256)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    B, A, N = args.rows, args.dims, args.particles
    g = torch.Generator(device=dev).manual_seed(11)
    logits, dlogits = kernels.stream_pair(B, A * N, dev)
    logits.normal_(generator=g)
    x = torch.rand(B, A, device=dev, generator=g) * 2.0 - 1.0
    grid = torch.linspace(-1, 1, N, device=dev)[None, :].expand(A, N)
    loc = (grid + 0.003 * torch.randn(A, N, device=dev, generator=g)).contiguous()
    logstd = (float(np.log(2.0 / (N - 1))) + 0.1 * torch.randn(A, N, device=dev, generator=g)).contiguous()
    lp0 = kernels.logprob_fwd(logits, x, loc, logstd)
    lp_old = (lp0 + 0.15 * torch.randn(B, device=dev, generator=g)).contiguous()
    adv, v, v_old = (torch.randn(B, device=dev, generator=g) for _ in range(3))
    stats = kernels.adv_stats(adv)

    def outs(dl):
        return {"logp": torch.empty(B, device=dev), "dlogits": dl, "dloc": torch.empty(A, N, device=dev),
                "dlogstd": torch.empty(A, N, device=dev), "dvalue": torch.empty(B, device=dev), "loss3": torch.empty(3, device=dev)}
    o3 = outs(dlogits)
    db3 = torch.empty(A * N, device=dev) if args.dbias else None
    kernels.WIDE_FUSED = False                               # the reference form: [forward + head] -> backward, two launches
    kernels.ppo_mixture_fwd_bwd(logits, x, loc, logstd, lp_old, adv, v, v_old, stats, 0.2, 0.5, out=o3, dbias=db3)
    ref = {k: t.clone() for k, t in o3.items() if k != "dlogits"}
    ref_dlogp = kernels._dlogp_scratch(B, dev).clone()
    rows = torch.arange(0, B, max(1, B // 512), device=dev)
    ref_dl_rows = o3["dlogits"][rows].clone()
    ref_colsum = o3["dlogits"].double().sum(0)
    ref_db = db3.clone() if db3 is not None else None
    o3["dlogits"].zero_()
    o2 = outs(dlogits)
    o2["dlogp"] = torch.empty(B, device=dev)
    db2 = torch.empty(A * N, device=dev) if args.dbias else None
    _, chunks, err = kernels.ppo_wide_fused(logits, x, loc, logstd, lp_old, adv, v, v_old, stats, 0.2, 0.5, out=o2, dbias=db2)
    torch.cuda.synchronize()
    print("fused 2L launch: %d chunks of %d waves, wait-expired flag %d" % (chunks, A, int(err.item())), flush=True)
    if int(err.item()):
        sys.exit("a bounded wait expired: the grid was not co-resident")

    def rel(a, b):
        return float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30))
    print("parity vs the 3L composition (max |diff| / max |ref|): logp %.2e  dvalue %.2e  dlogp %.2e  loss3 %.2e  dloc %.2e  "
          "dlogstd %.2e  dlogits(%d rows) %.2e  colsum(dlogits) %.2e%s"
          % (rel(o2["logp"], ref["logp"]), rel(o2["dvalue"], ref["dvalue"]), rel(o2["dlogp"], ref_dlogp),
             rel(o2["loss3"], ref["loss3"]), rel(o2["dloc"], ref["dloc"]), rel(o2["dlogstd"], ref["dlogstd"]),
             len(rows), rel(o2["dlogits"][rows], ref_dl_rows), rel(o2["dlogits"].double().sum(0), ref_colsum),
             ("  dbias %.2e" % rel(db2, ref_db)) if args.dbias else ""), flush=True)
    same_clip = bool(torch.equal(o2["dlogp"] != 0, ref_dlogp != 0))
    again = outs(dlogits)
    again["dlogp"] = torch.empty(B, device=dev)
    kernels.ppo_wide_fused(logits, x, loc, logstd, lp_old, adv, v, v_old, stats, 0.2, 0.5, out=again)
    torch.cuda.synchronize()
    print("same clip decisions: %s; run to run bitwise: logp %s dloc %s loss3 %s"
          % (same_clip, torch.equal(again["logp"], o2["logp"]), torch.equal(again["dloc"], o2["dloc"]),
             torch.equal(again["loss3"], o2["loss3"])), flush=True)
    if args.no_time:
        return
    L, X, V, P = 4 * B * A * N, 4 * B * A, 4 * B, 8 * A * N
    kernels.WIDE_FUSED = False
    t3 = graph_time(lambda: kernels.ppo_mixture_fwd_bwd(logits, x, loc, logstd, lp_old, adv, v, v_old, stats, 0.2, 0.5, out=o3,
                                                        dbias=db3))
    kernels.WIDE_FUSED = True
    t2 = graph_time(lambda: kernels.ppo_wide_fused(logits, x, loc, logstd, lp_old, adv, v, v_old, stats, 0.2, 0.5, out=o2,
                                                   dbias=db2))
    b3, b2 = 3 * L + 2 * X + 8 * V + 3 * P, 2 * L + X + 6 * V + 2 * P
    for name, t, by in (("3L composition (two launches)", t3, b3), ("2L fused launch", t2, b2)):
        print("%-32s %.1f - %.1f us   %.3f of 8 TB/s on its own algorithmic bytes (%.2f GB); %.3f on the 3L bytes"
              % (name, t[0], t[1], by / t[0] / 1e3 / 8000.0, by / 1e9, b3 / t[0] / 1e3 / 8000.0), flush=True)
    print("kill criterion (< 520 us): %s" % ("MET" if t2[0] < 520.0 else "NOT met"), flush=True)


if __name__ == "__main__":
    main()

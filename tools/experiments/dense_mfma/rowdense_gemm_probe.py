"""The six dense launches of a small-row DPPO learner step one by one (csrc/rowdense_kernels.hip): device time per launch
replayed from a hipGraph of 20 launches, per tile arrangement, against the fp32 MFMA peak, and the PyTorch-ROCm GEMMs of the
same products (no epilogues) for reference.  Usage: python tools/rowdense_gemm_probe.py [rows ...]   (PROBE_CFGS="0 1 2 -1")"""
import os, sys
import torch
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R)
from pfpn_amd import kernels as K
dev = torch.device("cuda", 0)
S, H0, H1, AN = 197, 1024, 512, 1260
cfgs = [int(c) for c in os.environ.get("PROBE_CFGS", "0 1 2 -1").split()]


def timeit(fn, per_graph=20, replays=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        fn(); torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=side):
            for _ in range(per_graph):
                fn()
    torch.cuda.synchronize(); g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(replays):
        g.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / (per_graph * replays)


for B in [int(a) for a in sys.argv[1:]] or [32, 256, 512, 1024]:
    r = lambda *s: torch.randn(*s, device=dev)
    x, w0, b0 = r(B, S), r(S, 2 * H0) * 0.05, r(2 * H0)
    act0 = r(B, 2 * H0).clamp(0, 6); a0 = act0.view(B, 2, H0).transpose(0, 1)
    w1, b1, act1 = r(2, H0, H1) * 0.05, r(2, H1), r(2, B, H1).clamp(0, 6)
    wp, bp, logits, dlog = r(H1, AN) * 0.05, r(AN), r(B, AN), r(B, AN)
    w3, b3, v, dv = r(H1), r(1), r(B), r(B)
    part = torch.empty(B, H1 // 32, device=dev)
    dh, d0 = r(2, B, H1), r(B, 2 * H0); d0v = d0.view(B, 2, H0).transpose(0, 1)
    g0, gb0, g1, gb1 = r(S, 2 * H0), r(2 * H0), r(2, H0, H1), r(2, H1)
    gp, gw3, gb3 = r(H1, AN), r(H1), r(1)
    o0 = torch.empty_like(act0); o1 = torch.empty_like(act1)
    launches = {
        "L0 fwd": (lambda c: [K.dense_job_nn(x, w0, o0, bias=b0, relu6=True, cfg=c)], 2.0 * B * S * 2 * H0),
        "L1 fwd": (lambda c: [K.dense_job_nn(a0, w1, o1, bias=b1, relu6=True, head=(1, w3, part), cfg=c)], 2.0 * B * 2 * H0 * H1),
        "head fwd": (lambda c: [K.dense_job_nn(act1[0], wp, logits, bias=bp, cfg=c), K.dense_job_head_final(part, b3, v)],
                     2.0 * B * H1 * AN),
        "head bwd": (lambda c: [K.dense_job_tn(act1[0], dlog, gp, cfg=c), K.dense_job_nt(dlog, wp, dh[0], mask=act1[0], cfg=c),
                                K.dense_job_rank1(dv, w3, act1[1], dh[1]), K.dense_job_head_wgrad(dv, act1[1], gw3, gb3)],
                     4.0 * B * H1 * AN),
        "L1 bwd": (lambda c: [K.dense_job_tn(a0, dh, g1, colsum=gb1, cfg=c), K.dense_job_nt(dh, w1, d0v, mask=a0, cfg=c)],
                   4.0 * B * 2 * H0 * H1),
        "L0 bwd": (lambda c: [K.dense_job_tn(x, d0, g0, colsum=gb0, cfg=c)], 2.0 * B * S * 2 * H0),
    }
    lib = {
        "L0 fwd": lambda: torch.mm(x, w0, out=o0), "L1 fwd": lambda: torch.bmm(a0, w1, out=o1),
        "head fwd": lambda: torch.addmm(bp, act1[0], wp, out=logits),
        "head bwd": lambda: (torch.mm(act1[0].t(), dlog, out=gp), torch.mm(dlog, wp.t(), out=dh[0])),
        "L1 bwd": lambda: (torch.bmm(a0.transpose(1, 2), dh, out=g1), torch.bmm(dh, w1.transpose(1, 2), out=d0v)),
        "L0 bwd": lambda: torch.mm(x.t(), d0, out=g0),
    }
    print("rows %d" % B)
    tot = {c: 0.0 for c in cfgs}; tot["lib"] = 0.0
    for name, (mk, flops) in launches.items():
        line = "  %-9s" % name
        for c in cfgs:
            jobs = mk(c)
            us = timeit(lambda: K.dense_jobs(jobs))
            tot[c] += us
            line += "  cfg%2d %6.1f us %5.1f TF" % (c, us, flops / us / 1e6)
        us = timeit(lib[name]); tot["lib"] += us
        line += "  | library GEMMs only %6.1f us" % us
        print(line, flush=True)
    print("  total    " + "  ".join("cfg%2d %6.1f us" % (c, tot[c]) for c in cfgs) + "  | library GEMMs only %6.1f us" % tot["lib"])

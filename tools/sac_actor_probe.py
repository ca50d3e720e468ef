"""The two actor-side launches of the SAC step alone (20 launches per hipGraph, replayed between events), against the
piece-wise launches they stand for, at replay batches 256 / 1024 (forward over 2B rows)."""
import os, sys
import numpy as np, torch
R_ = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R_)
from pfpn_amd import kernels
dev = torch.device("cuda", 0)


def time_in_graph(call, per_graph=20, replays=5):
    for _ in range(3):
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
    ea.record()
    for _ in range(replays):
        g.replay()
    eb.record()
    torch.cuda.synchronize()
    return ea.elapsed_time(eb) * 1e3 / (per_graph * replays)


def main():
    A, N, S = 36, 35, 197
    rng = np.random.default_rng(0)
    for B in (256, 1024):
        R = 2 * B
        t = lambda a: torch.from_numpy(a.astype(np.float32)).to(dev)
        loc = t(np.tile(np.linspace(-1.6, 1.6, N), (A, 1)))
        logstd = t(np.log(np.full((A, N), 3.2 / N)))
        logits, gumbel, eps = t(rng.normal(0, 1.5, (R, A * N))), t(-np.log(-np.log(rng.uniform(1e-10, 1, (R, A * N))))), t(rng.normal(0, 1, (R, A * N)))
        x, a_hist = t(rng.normal(0, 1, (R, S))), t(np.tanh(rng.normal(0, 1, (B, A))))
        xq_both, xq_next = torch.empty(2 * B, S + A, device=dev), torch.empty(B, S + A, device=dev)
        xa, xh, xn = (torch.empty(B, S + A, device=dev) for _ in range(3))
        f_new = time_in_graph(lambda: kernels.sac_actor_fwd(logits, loc, logstd, gumbel, eps, x=x, a_hist=a_hist, xq_both=xq_both, xq_next=xq_next))
        f_noq = time_in_graph(lambda: kernels.sac_actor_fwd(logits, loc, logstd, gumbel, eps))

        def f_old():
            a0, s0, k0 = kernels.rsample_fwd(logits, loc, logstd, gumbel, eps)
            kernels.logprob_fwd(logits, s0, loc, logstd, True)
            kernels.sac_q_inputs(x[:B], x[B:], a0[:B], a_hist, a0[B:], xa, xh, xn)
        f_o = time_in_graph(f_old)
        a0, s0, k0 = kernels.rsample_fwd(logits[:B], loc, logstd, gumbel[:B], eps[:B])
        dlogp, da = t(rng.normal(0, 0.01, B)), t(rng.normal(0, 0.02, (B, A)))
        ws = kernels.sac_actor_workspace(B, A, N, dev)
        dl = torch.empty(B, A * N, device=dev)
        b_new = time_in_graph(lambda: kernels.sac_actor_bwd(logits[:B], loc, logstd, gumbel[:B], eps[:B], s0, dlogp, da, dl, ws))
        wm, wr, wp = kernels.mixture_workspace(B, A, N, dev), kernels.rollout_workspace(B, A, N, dev), kernels.learner_workspace(B, A * N, dev)
        dloc, dls = torch.empty(A, N, device=dev), torch.empty(A, N, device=dev)

        def b_old():
            _, _, _, dxs = kernels.logprob_bwd(logits[:B], s0, loc, logstd, dlogp, True, True, out={"dlogits": dl, "dloc": dloc, "dlogstd": dls}, slabs_ws=wm)
            kernels.rsample_bwd(logits[:B], loc, logstd, gumbel[:B], eps[:B], da, dxs, dl, dloc, dls, slabs_ws=wr)
            kernels.act_bwd_partial(None, dl, 0, wp, relu6=False)
        b_o = time_in_graph(b_old)
        print("B %4d  forward: one launch %.2f us (without the critic inputs %.2f), three launches %.2f us | backward: one launch %.2f us, four launches %.2f us"
              % (B, f_new, f_noq, f_o, b_new, b_o), flush=True)


if __name__ == "__main__":
    main()

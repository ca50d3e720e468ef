#!/usr/bin/env python3
"""bench.py -- DPPO-PFPN learner steps/sec on MI355X (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one learner update on one 4096-transition synthetic batch PER GPU (config 2 of
BASELINE.json: DeepMimicWalk DPPO-PFPN, S=197, A=36, N=35, MLPs 197->1024->512): state-norm(+EMA) ->
actor & critic MLP forward -> fused particle-mixture log-prob + PPO loss + backward (HIP) -> MLP
backward -> global-norm clip -> [N>1: one RCCL all-reduce of the clipped grads + statistics] ->
TF-Adam -> resample tick.  Inputs are resident in HBM before the timed region.  `value` counts
4096-transition learner batches per second over ALL ranks (weak scaling: per-GPU work fixed).

Rank 0 prints ONE JSON line; besides the contract keys it carries
  roofline      -- the dominant hand-written kernel (fused mixture fwd+PPO+bwd) vs the HBM roof,
  roofline_gemm -- the FC GEMMs (PyTorch-ROCm) vs the fp32 MFMA roof,
  cpu_baseline  -- the NumPy oracle (op-for-op restatement of the reference's TF1 CPU path, NOT TF1)
                   timed on this box's host cores on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

S, A, NP, B = 197, 36, 35, 4096
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TF = 157.3       # MI355X_MICROARCH.md: fp32 matrix peak
SEED = 34114                   # benchmark.sh:13 first seed

NET_OPTS = dict(common_net_shape=[], critic_net_shape=[1024, 512], actor_net_shape=[1024, 512],
                value_loss_coef=0.5, gamma=0.95, normalize_state=True, clip_state=5.0,
                normalize_value=False, clip_value=None, normalize_advantage=True, clip_advantage=None,
                particles=NP, resample=-1, resample_interval=368)


def synth_batch(rng, rows):
    """SURVEY 8d synthetic inputs."""
    return {"state": (rng.normal(0, 1, (rows, S)) * 2 + 0.3).astype(np.float32),
            "action": rng.uniform(-1.1, 1.1, (rows, A)).astype(np.float32),
            "advantage": rng.normal(0, 1, rows).astype(np.float32)}


def algorithmic_bytes_fused(rows):
    L, X, V, P = 4 * rows * A * NP, 4 * rows * A, 4 * rows, 8 * A * NP
    return 2 * L + X + 6 * V + 2 * P        # read logits, write dlogits, action, 6 [B] vectors, bank in + grads out


def gemm_flops(rows):
    mac_fwd = S * 2048 + 2 * 1024 * 512 + 512 * A * NP      # per row (value head GEMV excluded)
    mac_bwd = S * 2048 + 2 * 2 * 1024 * 512 + 2 * 512 * A * NP
    return 2.0 * rows * (mac_fwd + mac_bwd)


def cpu_baseline(budget_s=20.0):
    """NumPy oracle learner step on a 4096-row batch on the host cores: all BLAS threads (the headline `value`, `cores` =
    the BLAS pool size actually used) and, for SURVEY 8d's like-for-like ratio, one thread."""
    from oracle import pfpn_oracle as orc
    try:
        from threadpoolctl import threadpool_info, threadpool_limits
        pools = [p.get("num_threads", 1) for p in threadpool_info() if p.get("user_api") == "blas"]
        blas_threads = max(pools) if pools else 1
    except Exception:
        threadpool_limits, blas_threads = None, os.cpu_count()
    rng = np.random.default_rng(SEED)
    st = orc.DPPOState(rng, S, A, NP)
    b = synth_batch(rng, B)
    x = orc.state_normalize(b["state"], st.norm_mean, st.norm_std, 5.0)
    logits, _ = orc.actor_forward(st.P, x, A, NP)
    b["log_prob"], _ = orc.mixture_log_prob(logits, b["action"], st.P["actor/samples"], st.P["actor/samples_std"])
    b["value"], _ = orc.critic_forward(st.P, x)
    orc.dppo_learner_step(st, [b])          # warm-up

    def timed(max_steps, budget):
        n, t0 = 0, time.perf_counter()
        while True:
            orc.dppo_learner_step(st, [b])
            n += 1
            el = time.perf_counter() - t0
            if el > budget or n >= max_steps:
                return n, el

    n, el = timed(20, budget_s * 0.6)
    runs = [{"value": n / el, "cores": blas_threads, "sample": "%d steps, %d BLAS threads" % (n, blas_threads)}]
    if threadpool_limits is not None and blas_threads > 1:
        with threadpool_limits(limits=1):
            n1, el1 = timed(20, budget_s * 0.4)
        runs.append({"value": n1 / el1, "cores": 1, "sample": "%d steps, 1 BLAS thread" % n1})
    best = max(runs, key=lambda r: r["value"])          # the baseline is the FASTER arrangement on this host
    out = {"value": best["value"], "unit": "steps/s", "cores": best["cores"], "host_cpus": os.cpu_count(), "kind": "port",
           "sample": "oracle learner steps on one 4096-transition batch (NumPy restatement; NumPy's elementwise part is "
                     "single-threaded, the GEMMs use the BLAS pool): " + best["sample"] + "; not TF1",
           "runs": runs}
    return out


def _flush_c_stdio():
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replays")
    ap.add_argument("--steps-per-graph", type=int, default=8,
                    help="learner steps captured per hipGraph replay (1: one replay per step).  Replays of a one-step "
                         "graph idle the GPU ~10 us between them; the remainder of --steps and anything across a "
                         "resample tick runs on a one-step graph")
    ap.add_argument("--no-cpu-baseline", action="store_true",
                    help="skip the informational extras (CPU baseline, 256-row reference-schedule step)")
    ap.add_argument("--no-tune", action="store_true", help="skip PyTorch TunableOp GEMM selection")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for --gpus > 1 (nccl = RCCL); "
                                                      "`gloo` + --one-device smoke-tests the multi-rank path on one GPU")
    ap.add_argument("--one-device", action="store_true", help="every rank on cuda:0 (control-flow smoke test only)")
    ap.add_argument("--force-exchange", action="store_true",
                    help="N=1 only: run the data-parallel step (clip in place, RCCL all-reduce in a world-size-1 group, "
                         "Adam with 1/W) instead of the fused single-replica step -- its cost without the links")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from pfpn_amd.models import SyncReplicasApplier
    from pfpn_amd.models.distributed_model import init_process_group
    from pfpn_amd.networks import ParticleFilteringClipPPONetwork
    from pfpn_amd import kernels, graphs

    tuned = None
    if not args.no_tune:
        # the GEMM selection found offline on an MI355X (tools/tune_gemms.py); shapes it lacks are tuned in the warm-up
        from pfpn_amd.tuning import enable_tuned_gemms
        tuned = enable_tuned_gemms(tune_missing=True)
    if args.one_device:
        os.environ["LOCAL_RANK"] = "0"
    rank, world, local_rank = init_process_group(args.backend if (args.gpus > 1 or args.force_exchange) else None,
                                                 single_rank_group=args.force_exchange)
    assert world == args.gpus, "launch with torchrun --nproc-per-node %d (WORLD_SIZE=%d)" % (args.gpus, world)
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    net = ParticleFilteringClipPPONetwork(trainable=True, state_shape=[S], action_shape=[A],
                                          action_lower_bound=-np.ones(A), action_upper_bound=np.ones(A),
                                          device=dev, seed=SEED, **NET_OPTS)
    net.init()
    opt = SyncReplicasApplier(net, [{"names": None, "lr": 1e-4}], norm_clip=1.0, world_size=world,
                              force_exchange=args.force_exchange)

    rng = np.random.default_rng(SEED + rank)      # every rank owns a different 4096-transition shard
    hb = synth_batch(rng, B)
    state = torch.from_numpy(hb["state"]).to(dev)
    action = torch.from_numpy(hb["action"]).to(dev)
    adv = torch.from_numpy(hb["advantage"]).to(dev)
    with torch.no_grad():      # value_old / log_prob_old = outputs of the (frozen) initial net: ratio ~ 1
        policy, ha, hc = net._forward(state)
        logp_old = policy.log_prob(action).clone()
        value_old = net._value(hc).clone()

    use_graph = (world == 1) and not args.no_graph and not args.force_exchange
    spg = max(1, args.steps_per_graph) if use_graph else 1
    if use_graph:
        step = graphs.GraphedLearnerStep(net, opt, state, action, value_old, logp_old, adv)
        step_k = graphs.GraphedLearnerStep(net, opt, state, action, value_old, logp_old, adv,
                                           steps_per_replay=spg) if spg > 1 else None
    else:
        step_k = None

        def step():
            return net.train_device(opt, state, action, value_old, logp_old, adv)

    def run_steps(n):
        """exactly n learner steps: K-step replays while they fit (and no resample tick falls inside), else single steps"""
        while n > 0:
            if step_k is not None and n >= spg and step_k.can_replay():
                step_k()
                n -= spg
            else:
                step()
                n -= 1

    if use_graph:            # the FIRST replay of a hipGraph uploads it: keep that out of the timed region even when
        step()               # --warmup is smaller than one multi-step replay (these steps are extra, untimed)
        if step_k is not None and step_k.can_replay():
            step_k()
    run_steps(args.warmup)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_steps(args.steps)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([el], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    loss3 = net.train_out["loss3"].cpu().numpy()
    assert np.all(np.isfinite(loss3)), "non-finite loss in the benchmark loop"

    out = None
    if rank == 0:
        ms = el / args.steps * 1e3
        out = {
            "metric": "DPPO learner steps/sec (4096 trans, 36 act-dim x 35 particles)",
            "value": world * args.steps / el, "unit": "steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "DeepMimicWalk-v0 DPPO-PFPN 35 particles, 4096-transition batch per GPU "
                                   "(S=197, A=36, N=35, MLP 1024/512), random-init weights",
                       "global_batch": B * world, "per_gpu_batch": B,
                       "parallelism": "dp%d" % world, "launch": ("hipGraph, %d learner steps per replay" % spg if spg > 1 else "hipGraph") if use_graph else "eager",
                       **({"exchange": "forced on one replica (RCCL world size 1)"} if args.force_exchange else {}),
                       "gemm_selection": ("shipped TunableOp file" if tuned else "TunableOp in warm-up") if not args.no_tune
                       else "library default",
                       "step": "state-norm, MLP fwd (+critic head), fused mixture logp+PPO+bwd, MLP bwd, one deferred "
                               "reduction launch (bias grads, particle grads, loss, state-norm EMA), clip, "
                               + ("RCCL all-reduce, " if (world > 1 or args.force_exchange) else "") + "TF-Adam, resample tick"},
        }

    # ---- roofline of the dominant hand-written kernel, and of the GEMMs (rank 0, N=1 only) ----------
    if rank == 0 and world == 1:
        logits = net.logits
        stats = net.adv_stats
        reps = 200

        def time_fused(finalize):
            for _ in range(10):
                kernels.ppo_mixture_fwd_bwd(logits, action, net.loc, net.logstd, logp_old, adv, net.value_buf, value_old,
                                            stats, 0.2, 0.5, out=net.train_out, dbias=net.store.g("actor/fc_policy/bias"),
                                            finalize=finalize)
            ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ea.record()
            for _ in range(reps):
                kernels.ppo_mixture_fwd_bwd(logits, action, net.loc, net.logstd, logp_old, adv, net.value_buf, value_old,
                                            stats, 0.2, 0.5, out=net.train_out, dbias=net.store.g("actor/fc_policy/bias"),
                                            finalize=finalize)
            eb.record()
            torch.cuda.synchronize()
            return ea.elapsed_time(eb) * 1e3 / reps

        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        # the dominant kernel alone (its launch back to back, the slab finalize skipped) -- the number rocprofv3's
        # per-kernel average must agree with -- and the whole op (main kernel + the ~4 us slab-reduction launch)
        us_b2b = time_fused(False)
        us_op = time_fused(True)

        # the kernel's duration INSIDE learner steps: HIP events on the launch stream around every call of the fused
        # kernel during 100 eager steps of the same workload (a captured graph cannot carry timing events); this is
        # the number rocprofv3's per-kernel average of this command agrees with
        pairs = []
        real_fused = kernels.ppo_mixture_fwd_bwd

        def timed_fused(*a, **k):
            ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ea.record()
            r = real_fused(*a, **k)
            eb.record()
            pairs.append((ea, eb))
            return r

        kernels.ppo_mixture_fwd_bwd = timed_fused
        try:
            for i in range(110):
                if i == 10:
                    pairs.clear()
                net.train_device(opt, state, action, value_old, logp_old, adv)
            torch.cuda.synchronize()
        finally:
            kernels.ppo_mixture_fwd_bwd = real_fused
        us_in_step = sum(a.elapsed_time(b) for a, b in pairs) * 1e3 / len(pairs)
        # an event pair around ONE ~19 us launch also times the dispatch gap behind the first marker (~3 us); the
        # back-to-back loop amortises it and is the figure that agrees with rocprofv3's per-kernel average
        us = us_b2b
        ab = algorithmic_bytes_fused(B)
        ach = ab / (us * 1e-6) / 1e9
        traffic = None
        try:      # HBM bytes per launch from the committed rocprofv3 --pmc passes (FETCH_SIZE x2 + WRITE_SIZE)
            with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")) as f:
                traffic = json.load(f)["kernels"]["k_mix_packed<35,PPO>"]["traffic_bytes"]
        except Exception:
            pass
        out["roofline"] = {"kernel": "k_mix_packed<35,PPO>", "bound": "hbm", "achieved": ach,
                           "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
                           "algorithmic_bytes": ab, "us_per_launch": us, "launches_timed": reps,
                           "us_in_step_event_pairs": us_in_step, "us_with_slab_finalize_launch": us_op,
                           "note": "us_per_launch: HIP events around 200 back-to-back launches of the kernel alone (in "
                                   "the step its slab sums are finished by the single deferred reduce launch); "
                                   "us_in_step_event_pairs: an event pair around each call inside 100 eager learner "
                                   "steps (includes the dispatch gap); "
                                   "42 MB per launch: start-up and VALU-issue bound at this size (DESIGN.md section 4); "
                                   "the 1 GiB shape of BASELINE.json config 5 is in roofline_stress"}
        # the same mixture math on the 1 GiB stress shape (BASELINE.json config 5, one GPU's shard)
        try:
            sB, sA, sN = 8192, 128, 256
            g = torch.Generator(device=dev); g.manual_seed(1)
            sl = torch.randn(sB, sA * sN, device=dev, generator=g)
            sx = torch.rand(sB, sA, device=dev, generator=g) * 2.2 - 1.1
            sloc = torch.linspace(-1, 1, sN, device=dev).repeat(sA, 1).contiguous()
            sls = torch.full((sA, sN), float(np.log(2.0 / (sN - 1))), device=dev)
            sg = torch.randn(sB, device=dev, generator=g)
            L_, X_, V_, P_ = 4 * sB * sA * sN, 4 * sB * sA, 4 * sB, 8 * sA * sN
            res = {}
            for name, fn, nbytes in (("fwd", lambda: kernels.logprob_fwd(sl, sx, sloc, sls), L_ + X_ + V_ + P_),
                                     ("bwd", lambda: kernels.logprob_bwd(sl, sx, sloc, sls, sg), 2 * L_ + X_ + 2 * V_ + 2 * P_)):
                fn(); fn()
                torch.cuda.synchronize()
                e0.record()
                for _ in range(10):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                u = e0.elapsed_time(e1) * 1e3 / 10
                res[name] = {"kernel": "k_mix_wide<1,%s> (+reduce)" % name.upper(), "us_per_launch": u,
                             "algorithmic_bytes": nbytes, "achieved": nbytes / u / 1e3, "unit": "GB/s",
                             "peak": HBM_PEAK_GBS, "frac": nbytes / u / 1e3 / HBM_PEAK_GBS}
            out["roofline_stress"] = {"shape": [sB, sA, sN], "bound": "hbm", **res}
            del sl, sx, sg
        except Exception as ex:      # never fail the headline line on the auxiliary measurement
            out["roofline_stress"] = {"error": repr(ex)}
        # GEMM-only pass: forward + backward dense layers of one step
        def gemms():
            x = net.xn
            h = net.trunk.forward(x)
            torch.mm(h[0], net.policy_weight, out=net.logits)
            d = net.train_out["dlogits"]
            torch.mm(h[0].t(), d, out=net.store.g("actor/fc_policy/weight"))
            torch.mm(d, net.policy_weight.t(), out=net.dh[0])
            net.trunk.backward(net.dh)
        for _ in range(5):
            gemms()
        e0.record()
        for _ in range(50):
            gemms()
        e1.record()
        torch.cuda.synchronize()
        gus = e0.elapsed_time(e1) * 1e3 / 50
        tf = gemm_flops(B) / (gus * 1e-6) / 1e12
        out["roofline_gemm"] = {"kernel": "11 fp32 GEMMs via PyTorch-ROCm (+bias/ReLU6 epilogue kernels)",
                                "bound": "mfma", "achieved": tf, "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s",
                                "frac": tf / MFMA_F32_PEAK_TF, "us_per_step": gus, "flops": gemm_flops(B)}
        # the reference's own schedule is 16 optimiser steps of global batch 256 (8 workers x 32 rows) per 4096
        # transitions (deepmimic_dppo_base.py) -- one 256-row step on this GPU, captured like the headline step
        try:
            if args.no_cpu_baseline:         # profiling runs: keep the kernel statistics to the headline shape
                raise RuntimeError("skipped with --no-cpu-baseline")
            rb = synth_batch(np.random.default_rng(SEED + 1), 256)
            rt = [torch.from_numpy(rb[k]).to(dev) for k in ("state", "action", "advantage")]
            with torch.no_grad():
                rp, _, rhc = net._forward(rt[0])
                r_lp = rp.log_prob(rt[1]).clone()
                r_v = net._value(rhc).clone()
            rstep = graphs.GraphedLearnerStep(net, opt, rt[0], rt[1], r_v, r_lp, rt[2])
            for _ in range(10):
                rstep()
            torch.cuda.synchronize()
            tr = time.perf_counter()
            for _ in range(200):
                rstep()
            torch.cuda.synchronize()
            rus = (time.perf_counter() - tr) / 200 * 1e6
            out["reference_schedule"] = {"global_batch": 256, "steps_per_4096_transitions": 16, "us_per_step": rus,
                                         "transitions_per_s": 256 / (rus * 1e-6),
                                         "note": "the headline step consumes 4096 transitions per step: %.0f transitions/s"
                                                 % (B * out["value"])}
        except Exception as e:          # informational only
            out["reference_schedule"] = {"error": repr(e)}
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
    # The JSON line must be the LAST thing on stdout.  RCCL leaves a version banner in the C stdio buffer of every rank,
    # which would otherwise reach the pipe at process exit -- after a line printed from Python.  So every rank flushes
    # its C streams, the ranks meet at a barrier, the group is torn down, and only then rank 0 prints.
    if dist.is_initialized():
        dist.barrier()
        _flush_c_stdio()
        dist.barrier()                      # every rank has flushed its banner
        dist.destroy_process_group()
    _flush_c_stdio()
    if rank == 0:
        sys.stdout.write(json.dumps(out) + "\n")
        sys.stdout.flush()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""bench.py -- DPPO-PFPN learner steps/sec on MI355X (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one learner update on one 4096-transition synthetic batch (config 2 of BASELINE.json: DeepMimicWalk
DPPO-PFPN, S=197, A=36, N=35, MLPs 197->1024->512): state-norm(+EMA) -> actor & critic MLP forward -> fused
particle-mixture log-prob + PPO loss + backward (HIP) -> MLP backward -> global-norm clip -> [N>1: one RCCL all-reduce
of the clipped grads + statistics] -> TF-Adam -> resample tick.  Inputs are resident in HBM before the timed region.
The metric is quoted on a 4096-transition batch at 1/2/4/8 GPUs ("B sharded"), so with N GPUs the SAME
global batch is split 4096/N rows per rank ("scaling": "strong") and `value` counts global learner steps per second;
`--weak` keeps 4096 rows per rank instead (also measured as the extra key `weak_scaling` on N > 1 runs).

Rank 0 prints ONE JSON line; besides the contract keys it carries
  roofline      -- the dominant hand-written kernel (fused mixture fwd+PPO+bwd) vs the HBM roof,
  roofline_gemm -- the FC GEMMs (PyTorch-ROCm) vs the fp32 MFMA roof,
  strong_shard  -- (N = 1) the data-parallel step of ONE rank at 2048 / 1024 / 512 rows over a world-size-1 RCCL group:
                   what a rank of a 2 / 4 / 8-GPU run spends per step before any byte crosses a link,
  weak_scaling  -- rows per GPU constant (config 4's natural mode): 4096 rows on every rank,
  sac_step      -- (N = 1) the SAC-PFPN learner step of BASELINE.json config 3 (replay batches 1024 and 256) with its roofline,
  config5_step  -- (N = 1) BASELINE.json config 5 as a learner step on one GPU's 1024-row shard, with its roofline,
  cpu_baseline  -- restatements of the reference's TF1 CPU path (NOT TF1) timed on this box's host cores on a bounded
                   sample, in a child process: all-core and single-thread 4096-row steps, and the reference's own
                   8-worker x 32-row schedule (oracle/cpu_baseline.py).
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


def latest_profile(suffix):
    """the newest committed profiles/rNN_<suffix> (path, round tag) or (None, None)"""
    import glob
    fs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_" + suffix)))
    return (fs[-1], os.path.basename(fs[-1])[:3]) if fs else (None, None)


def rocprof_kernel_us(kernel_substring):
    """average duration (us) of a kernel in the newest committed `rocprofv3 --kernel-trace --stats` summary of this command
    (profiles/rNN_bench_kernel_stats.csv) -> (us, calls, file) or None"""
    import csv
    path, _ = latest_profile("bench_kernel_stats.csv")
    if not path:
        return None
    try:
        for r in csv.DictReader(open(path)):
            if kernel_substring in r.get("Name", ""):
                return float(r["AverageNs"]) / 1e3, int(r["Calls"]), "profiles/" + os.path.basename(path)
    except Exception:
        return None
    return None


def stress_traffic(kernel):
    """HBM bytes per launch of a stress-shape kernel from the newest committed PMC passes (profiles/rNN_pmc_traffic.json) or None"""
    path, _ = latest_profile("pmc_traffic.json")
    try:
        return json.load(open(path))["kernels"][kernel]["traffic_bytes"]
    except Exception:
        return None


def timeline_launches(suffix):
    """launch count of one step in the newest committed rocprofv3 timeline profiles/rNN_<suffix> -> (n, file) or (None, None)"""
    import re
    path, _ = latest_profile(suffix)
    if not path:
        return None, None
    m = re.search(r"(\d+) launches", open(path).read().strip().splitlines()[-1])
    return (int(m.group(1)), "profiles/" + os.path.basename(path)) if m else (None, None)


def dense_macs(S_in, hidden, head):
    """multiply-accumulates per row of an MLP S_in -> hidden... -> head: (forward, dW, dX below the first layer)"""
    dims = [S_in] + list(hidden) + [head]
    fwd = sum(a * b for a, b in zip(dims[:-1], dims[1:]))
    return fwd, fwd, fwd - dims[0] * dims[1]


def cpu_baseline(budget_s=24.0):
    """oracle/cpu_baseline.py in a CHILD process (it forks worker processes and pins BLAS threads; this process holds
    the GPU).  Returns its JSON: value = the fastest 4096-row CPU arrangement, runs = every leg."""
    import subprocess
    env = dict(os.environ)
    env.pop("OPENBLAS_NUM_THREADS", None)
    env["CUDA_VISIBLE_DEVICES"] = env["HIP_VISIBLE_DEVICES"] = ""         # the child never touches the GPU
    r = subprocess.run([sys.executable, "-m", "oracle.cpu_baseline", "--budget", str(budget_s)], cwd=ROOT, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=max(180.0, 8 * budget_s))
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
    if r.returncode != 0 or not lines:
        return {"error": "cpu_baseline child failed (rc %d): %s" % (r.returncode, r.stderr.decode()[-400:])}
    return json.loads(lines[-1])


def _flush_c_stdio():
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()


# ---- the printed line ---------------------------------------------------------------------------------------------------
# The driver keeps ~8.6 KB of stdout + stderr; round 4's 13.5-KB line lost its front (roofline_stress among it).  The line
# printed is therefore the NUMBERS: prose (`note`, `*_source`, long descriptions) lives in profiles/README.md and in the full
# line this run also writes to gpurun_out/bench_line_full.json; floats carry 5-7 significant digits; a `digest` of the
# figures a reader looks for first is the LAST key, so even a short tail of stdout holds them.
LINE_BUDGET = 7600
_DROP_ORDER = ("reference_schedule", "roofline_more_rows", "weak_scaling", "exchange_two_graphs", "exchange_sharded",
               "strong_shard", "config5_step", "sac_step")          # least essential first, if a line is still too long


_KEEP_TEXT = ("workload", "metric", "error", "sample", "launch", "headline_form")     # descriptions the line keeps whole


def _sig(x, n):
    return float("%.*g" % (n, x)) if x == x and abs(x) != float("inf") else x


def _compact(o, depth=0, parent=None):
    if isinstance(o, dict):
        out = {}
        for k, v in o.items():
            if k == "note" or k.endswith("_source") or k.endswith("_per_kernel") or k == "box_copy":
                continue
            if k == "launch" and parent not in (None, "config", "exchange_two_graphs", "exchange_sharded", "weak_scaling"):
                continue
            if isinstance(v, str):
                if k == "kernel" and " (" in v:
                    v = v[:v.index(" (")]                       # the parenthetical is in profiles/README.md
                if len(v) > 96 and k not in _KEEP_TEXT:
                    continue
            out[k] = _compact(v, depth + 1, k)
        return out
    if isinstance(o, float):
        return _sig(o, 7 if depth <= 1 else 5)
    if isinstance(o, (list, tuple)):
        return [_compact(v, depth + 1, parent) for v in o]
    return o


def _dig(o, *path):
    for k in path:
        if not isinstance(o, dict) or k not in o:
            return None
        o = o[k]
    return _sig(o, 4) if isinstance(o, float) else o


def wall_ms(call, n, windows=3):
    """Wall-clock ms per call of an informational block (NOT the headline, whose K steps are timed once as the contract says):
    the quietest of three windows of n calls, so that one host hiccup does not end up in the line."""
    import torch
    best = float("inf")
    for _ in range(windows):
        t_s = time.perf_counter()
        for _ in range(n):
            call()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t_s) / n * 1e3)
    return best


def digest(out):
    """<= 600 bytes: the figures the review reads first"""
    d = {"stress_fwd": _dig(out, "roofline_stress", "fwd", "frac"), "stress_bwd": _dig(out, "roofline_stress", "bwd", "frac"),
         "stress_ppo": _dig(out, "roofline_stress", "ppo", "frac"),
         "stress_ppo_on_3l_bytes": _dig(out, "roofline_stress", "ppo", "frac_on_3l_bytes"),
         "stress_ppo_3l_form": _dig(out, "roofline_stress", "ppo_3l", "frac"),
         "stress_ppo_us": _dig(out, "roofline_stress", "ppo", "us_per_launch"),
         "tile_frac": _dig(out, "roofline", "frac"), "tile_frac_events": _dig(out, "roofline", "frac_events"),
         "sac_ms_1024": _dig(out, "sac_step", "ms_per_step"), "sac_ms_256": _dig(out, "sac_step", "batch_256", "ms_per_step"),
         "config5_ms": _dig(out, "config5_step", "ms_per_step"), "value_200": _dig(out, "value_200"),
         "ms_min": _dig(out, "ms_per_step_min"), "ms_median": _dig(out, "ms_per_step_median"),
         "ms_max": _dig(out, "ms_per_step_max"),
         "shard_us_512": _dig(out, "strong_shard", "rows_per_gpu", "512", "us_per_step"),
         "mfma_busy": _dig(out, "roofline_gemm", "mfma_busy"), "gemm_frac": _dig(out, "roofline_gemm", "frac"),
         "cpu_steps_s": _dig(out, "cpu_baseline", "value"), "value": _dig(out, "value")}
    return {k: v for k, v in d.items() if v is not None}


def compact_line(out, budget=LINE_BUDGET):
    """the full result dict -> the ONE line printed (a str, <= `budget` bytes): prose dropped, floats rounded, digest last;
    whole blocks leave (least essential first, named in `dropped`) only if that is still not enough"""
    c = _compact(out)
    c.pop("digest", None)
    dropped = []

    def render():
        line = dict(c)
        if dropped:
            line["dropped"] = dropped
        line["digest"] = digest(out)
        return json.dumps(line, separators=(",", ":"))
    s = render()
    keep = {"config", "roofline", "cpu_baseline", "roofline_stress", "rccl"}          # the contract's objects + the stress target
    order = [k for k in _DROP_ORDER if k in c]
    order += sorted((k for k, v in c.items() if isinstance(v, (dict, list)) and k not in keep and k not in order),
                    key=lambda k: -len(json.dumps(c[k])))
    for k in order + ["roofline_stress"]:
        if len(s) <= budget:
            break
        if k in c:
            del c[k]
            dropped.append(k)
            s = render()
    return s


def print_line(out):
    """rank 0: the full line to gpurun_out/bench_line_full.json (best effort), the compact one to stdout"""
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "bench_line_full.json"), "w") as f:
            json.dump(out, f)
    except OSError:
        pass
    sys.stdout.write(compact_line(out) + "\n")
    sys.stdout.flush()


class ExtrasDeadline(object):
    """Multi-rank runs only.  The informational extras behind the headline hold collectives of their own (a second
    learner on the other scaling mode / exchange form); if ONE rank fails inside them, the others would wait in a
    collective for ever and the headline -- already measured -- would be lost with the job.  So: once the headline is
    known a timer starts on every rank; a rank whose extra raised parks here instead of walking into the next
    collective; when the timer fires rank 0 prints the headline line it has (marked `extras_timed_out`) and every rank
    leaves with exit code 0."""

    def __init__(self, seconds, emit):
        import threading
        self.lock, self.done, self.emit = threading.Lock(), False, emit
        self.timer = threading.Timer(seconds, self.fire)
        self.timer.daemon = True
        self.timer.start()

    def fire(self):
        with self.lock:
            if self.done:
                return
            self.done = True
        self.emit()
        os._exit(0)

    def park(self):
        """a rank that cannot take part in the remaining collectives waits for the timer"""
        while True:
            time.sleep(1.0)

    def finish(self):
        """-> True when the caller owns the final print (the timer has not fired and will not)"""
        with self.lock:
            if self.done:
                return False
            self.done = True
        self.timer.cancel()
        return True


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
    ap.add_argument("--weak", action="store_true",
                    help="weak scaling: 4096 rows on EVERY rank (global batch 4096 x N) instead of the metric's "
                         "4096-row global batch split over the ranks")
    ap.add_argument("--force-exchange", action="store_true",
                    help="N=1 only: run the data-parallel step (clip in place, RCCL all-reduce in a world-size-1 group, "
                         "Adam with 1/W) instead of the fused single-replica step -- its cost without the links")
    ap.add_argument("--two-graph-exchange", action="store_true",
                    help="data-parallel step as two hipGraphs around an eagerly enqueued all-reduce (the automatic fall-back "
                         "form) instead of the default: the all-reduce captured inside the step's graph")
    ap.add_argument("--extras-deadline", type=float, default=240.0,
                    help="N > 1: seconds the informational extras may take before rank 0 prints the headline it has")
    ap.add_argument("--watchdog", type=float, default=0.0,
                    help="diagnostic: every rank dumps its Python stacks and exits after this many seconds")
    ap.add_argument("--fail-extra", default="", help="rehearsal of the extras' deadline: `<extra name>:<rank>` raises there")
    ap.add_argument("--alloc-shift", type=int, default=0,
                    help="diagnostic (tools/placement_probe.sh): a dummy allocation of this many bytes in front of everything")
    args = ap.parse_args()
    if args.watchdog > 0:
        import faulthandler
        faulthandler.dump_traceback_later(args.watchdog, exit=True)

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
    if args.alloc_shift:
        # a dummy allocation first moves every buffer of the step relative to the 2 MiB segments the allocator carves --
        # does the headline depend on where things land?
        _shift = torch.empty(args.alloc_shift, dtype=torch.uint8, device=dev)

    def make_learner(rows, seed, exchange_group=False, exchange_mode=None):
        """a network + optimiser + `rows` synthetic transitions resident in HBM + the step callable"""
        net_ = ParticleFilteringClipPPONetwork(trainable=True, state_shape=[S], action_shape=[A],
                                               action_lower_bound=-np.ones(A), action_upper_bound=np.ones(A),
                                               device=dev, seed=SEED, **NET_OPTS)
        net_.init()
        opt_ = SyncReplicasApplier(net_, [{"names": None, "lr": 1e-4}], norm_clip=1.0, world_size=world,
                                   force_exchange=args.force_exchange or exchange_group, exchange_mode=exchange_mode)
        hb = synth_batch(np.random.default_rng(seed), rows)
        t = {k: torch.from_numpy(hb[k]).to(dev) for k in ("state", "action", "advantage")}
        with torch.no_grad():      # value_old / log_prob_old = outputs of the (frozen) initial net: ratio ~ 1
            policy, ha, hc = net_._forward(t["state"])
            t["log_prob"] = policy.log_prob(t["action"]).clone()
            t["value"] = net_._value(hc).clone()
        return net_, opt_, (t["state"], t["action"], t["value"], t["log_prob"], t["advantage"])

    def make_stepper(net_, opt_, inputs, allow_multi=True, capture_exchange=None):
        """-> (run_steps(n), launch description).  hipGraph replays of `--steps-per-graph` learner steps, the remainder
        n % K as ONE replay of a K'-step graph (captured by run_steps.prepare(n), outside every timed region); data parallel
        over RCCL: the same, the all-reduce captured inside (pfpn_amd/graphs.py; it falls back to two hipGraphs around an
        eager collective by itself, every rank together, if the capture fails anywhere); --no-graph: eager."""
        if capture_exchange is None and args.two_graph_exchange:
            capture_exchange = False
        if args.no_graph:
            f = lambda n: [net_.train_device(opt_, *inputs) for _ in range(n)]
            f.prepare = lambda n: None
            return f, "eager"
        one = graphs.GraphedLearnerStep(net_, opt_, *inputs, capture_exchange=capture_exchange)
        spg_ = max(1, args.steps_per_graph) if (allow_multi and not one.split) else 1
        sized = {1: one}
        one()                    # the FIRST replay of a hipGraph uploads it: keep that out of every timed region

        def graph_of(k):
            g = sized.get(k)
            if g is None:
                g = sized[k] = graphs.GraphedLearnerStep(net_, opt_, *inputs, steps_per_replay=k,
                                                         capture_exchange=capture_exchange)
                if g.can_replay():
                    g()
            return g

        def prepare(n):
            """capture (and upload) every graph a run of n steps will replay"""
            if spg_ > 1:
                if n >= spg_:
                    graph_of(spg_)
                if n % spg_ > 1:
                    graph_of(n % spg_)

        def run_steps(n):
            """exactly n learner steps: K-step replays while they fit, the rest as one K'-step replay (one-step replays
            only up to a resample tick that falls inside a multi-step replay)"""
            while n > 0:
                k = min(n, spg_)
                g = sized.get(k) if k > 1 else None
                if g is not None and g.can_replay():
                    g()
                    n -= k
                else:
                    one()
                    n -= 1
        run_steps.prepare = prepare
        lib_ = "RCCL" if (not dist.is_initialized() or dist.get_backend() == "nccl") else dist.get_backend()
        what = ("two hipGraphs around the %s" % ("sharded exchange (all-to-all | owner Adam | all-gather)"
                                                 if getattr(opt_, "sharded", False) else "%s all-reduce" % lib_)) if one.split else \
            ("hipGraph, %d learner steps per replay (remainder as one shorter replay)" % spg_ if spg_ > 1 else "hipGraph")
        if one.in_graph_exchange:
            what += ", the RCCL all-reduce captured inside"
        if one.exchange_fallback:
            what += " (FALL-BACK: capturing the exchange failed: %s)" % one.exchange_fallback[:120]
        return run_steps, what

    def timed_steps(run_steps, n_steps, n_warm):
        """the contract's timed region: barrier + synchronize on both sides, MAX over ranks."""
        run_steps.prepare(n_warm)
        run_steps.prepare(n_steps)
        run_steps(n_warm)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run_steps(n_steps)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        el_ = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el_], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el_ = float(t.item())
        return el_

    def prelude(run_steps, seconds=0.25):
        """untimed, disclosed clock-ramp: a fresh box starts its first kernels at a low clock and a 20-step window
        (~10 ms) would otherwise be measured on the ramp.  Returns its length in ms."""
        run_steps.prepare(8)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if world > 1:
            # the steps hold collectives: every rank must run the SAME number of them, so the count is agreed (from the
            # slowest rank's time for one replay) instead of each rank watching its own clock
            run_steps(8)
            torch.cuda.synchronize()
            t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            for _ in range(int(min(max(seconds / max(float(t.item()), 1e-6) - 1, 0), 4096))):
                run_steps(8)
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) * 1e3
        while time.perf_counter() - t0 < seconds:
            run_steps(8)
            torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3

    assert B % world == 0, "--gpus must divide 4096"
    strong = not args.weak
    rows = B // world if strong else B            # rank r owns rows [r * B/W, (r+1) * B/W) of the global batch
    net, opt, inputs = make_learner(rows, SEED + rank)
    state, action, value_old, logp_old, adv = inputs
    use_graph = not args.no_graph
    holder = {"out": None}                   # what rank 0 prints (also from the deadline's timer thread)

    def headline(run_steps_, launch_, spread=True):
        """the contract's timed window (+ its spread over 5 windows and the same figure over 200 steps) -> the line's
        headline fields.  Every rank runs it (the steps hold collectives); rank 0 keeps the dict."""
        # every hipGraph the windows below replay is captured (and uploaded) FIRST: a capture idles the GPU for tens of
        # milliseconds, and captures sitting between the clock-ramp prelude and the contract's window left that window --
        # the one `value` comes from -- 0.6-0.9 % slower than the four behind it (BENCH_r05: 0.4573 vs 0.4533-0.4544 ms)
        for n_ in (8, args.warmup, args.steps) + ((200,) if (spread and args.steps != 200) else ()):
            run_steps_.prepare(n_)
        prelude_ms = prelude(run_steps_)
        el = timed_steps(run_steps_, args.steps, args.warmup)          # the contract's window: `value`
        windows = [el] + ([timed_steps(run_steps_, args.steps, 0) for _ in range(4)] if spread else [])
        el_200 = timed_steps(run_steps_, 200, 0) if (spread and args.steps != 200) else el
        loss3 = net.train_out["loss3"].cpu().numpy()
        assert np.all(np.isfinite(loss3)), "non-finite loss in the benchmark loop"
        if rank != 0:
            return None
        ms = el / args.steps * 1e3
        return {
            "metric": "DPPO learner steps/sec (4096 trans, 36 act-dim x 35 particles)",
            "value": (1 if strong else world) * args.steps / el, "unit": "steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True,
            "prelude_ms": prelude_ms, "windows": len(windows),
            "ms_per_step_min": min(windows) / args.steps * 1e3,
            "ms_per_step_median": sorted(windows)[len(windows) // 2] / args.steps * 1e3,
            "ms_per_step_max": max(windows) / args.steps * 1e3,
            "value_200": (1 if strong else world) * 200 / el_200,
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "DeepMimicWalk-v0 DPPO-PFPN 35 particles, 4096-transition batch%s "
                                   "(S=197, A=36, N=35, MLP 1024/512), random-init weights"
                                   % (" sharded %d rows per GPU" % rows if (strong and world > 1) else
                                      (" per GPU" if world > 1 else "")),
                       "global_batch": rows * world, "per_gpu_batch": rows,
                       "parallelism": "dp%d" % world, "launch": launch_,
                       **({"exchange": "forced on one replica (RCCL world size 1)"} if args.force_exchange else {}),
                       "gemm_selection": ("shipped TunableOp file" if tuned else "TunableOp in warm-up") if not args.no_tune
                       else "library default",
                       "step": "state-norm, MLP fwd (+critic head), fused mixture logp+PPO+bwd, MLP bwd, one deferred "
                               "reduction launch (bias grads, particle grads, loss, state-norm EMA), clip, "
                               + (("sharded exchange (all-to-all, owner Adam, all-gather), " if getattr(opt, "sharded", False)
                                   else "%s all-reduce, " % ("RCCL" if dist.get_backend() == "nccl" else dist.get_backend()))
                                  if (world > 1 or args.force_exchange) else "")
                               + "TF-Adam, resample tick"},
        }

    # ---- behind a deadline: once a headline is measured, nothing that follows may cost the line (see ExtrasDeadline) ----
    deadline = None

    def emit_headline():
        if rank == 0 and holder["out"] is not None:
            holder["out"]["extras_timed_out"] = True
            _flush_c_stdio()
            print_line(holder["out"])

    # Data parallel over RCCL: the default step captures the all-reduce INSIDE its graph -- the form no multi-GPU node has run
    # yet.  A capture that RAISES falls back by itself (pfpn_amd/graphs.py); one that HANGS cannot.  So the two-graph form --
    # the one every earlier round measured -- is timed FIRST and already stands as the line when the default form starts; the
    # default form's numbers replace it when they exist.
    exchanging = world > 1 or args.force_exchange
    safe_first = exchanging and use_graph and not args.two_graph_exchange and not getattr(opt, "sharded", False) and \
        dist.is_initialized() and dist.get_backend() == "nccl" and os.environ.get("PFPN_GRAPH_EXCHANGE", "1") != "0"
    safe = None
    if safe_first:
        run_safe, launch_safe = make_stepper(net, opt, inputs, capture_exchange=False)
        holder["out"] = headline(run_safe, launch_safe, spread=False)
        if rank == 0:
            holder["out"]["headline_form"] = "two hipGraphs around an eager all-reduce (the default form had not finished)"
            safe = {"value": holder["out"]["value"], "unit": "steps/s", "ms_per_step": holder["out"]["ms_per_step"],
                    "launch": launch_safe, "note": "timed first: the form that stands if the default one cannot finish"}
        deadline = ExtrasDeadline(args.extras_deadline, emit_headline)
        try:
            if args.fail_extra == "default_form:%d" % rank:         # rehearsal: this rank dies inside the default form
                raise RuntimeError("injected failure")
            run_steps, launch = make_stepper(net, opt, inputs)
            better = headline(run_steps, launch)
            if rank == 0:
                better["exchange_two_graphs"] = safe
                holder["out"] = better
        except Exception as e:
            sys.stderr.write("bench.py rank %d: the default exchange form failed: %r\n" % (rank, e))
            deadline.park()
    else:
        run_steps, launch = make_stepper(net, opt, inputs)
        holder["out"] = headline(run_steps, launch)
        if world > 1:
            deadline = ExtrasDeadline(args.extras_deadline, emit_headline)
    out = holder["out"]

    def guarded(fn):
        """an extra on every rank: its result, or -- when THIS rank failed in it -- no way on into the next collective"""
        try:
            if args.fail_extra == "%s:%d" % (fn.__name__, rank):       # rehearsal of the deadline
                raise RuntimeError("injected failure")
            return fn()
        except Exception as e:
            sys.stderr.write("bench.py rank %d: extra %s failed: %r\n" % (rank, fn.__name__, e))
            if deadline is None:
                return {"error": repr(e)}
            deadline.park()

    def rccl_block():
        # evidence from the LIVE process group that N ranks took part: a sum of ones and every rank's device index
        tdev = dev if dist.get_backend() == "nccl" else torch.device("cpu")
        ones = torch.ones(1, device=tdev)
        dist.all_reduce(ones)
        devs = [torch.zeros(1, dtype=torch.int64, device=tdev) for _ in range(world)]
        dist.all_gather(devs, torch.tensor([torch.cuda.current_device()], dtype=torch.int64, device=tdev))
        try:
            ver = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:
            ver = None
        coll = None
        if dist.get_backend() == "nccl":
            # what the links give the step's collectives on their own (20 back-to-back calls each, max over ranks): the
            # bucket all-reduce of the default exchange; the all-to-all + in-place all-gather of the sharded one
            nb_ = net.store.bucket.numel()
            npad_ = (nb_ + 256 * world - 1) // (256 * world) * (256 * world)
            bufs = (torch.zeros(nb_, device=dev), torch.zeros(npad_, device=dev), torch.zeros(npad_, device=dev))
            calls = {"all_reduce_bucket": lambda: dist.all_reduce(bufs[0]),
                     "all_to_all_bucket": lambda: dist.all_to_all_single(bufs[2], bufs[1]),
                     "all_gather_in_place": lambda: dist.all_gather_into_tensor(
                         bufs[1], bufs[1][rank * (npad_ // world):(rank + 1) * (npad_ // world)])}
            coll = {}
            for name_, fn_ in calls.items():
                for _ in range(3):
                    fn_()
                torch.cuda.synchronize()
                dist.barrier()
                t0_ = time.perf_counter()
                for _ in range(20):
                    fn_()
                torch.cuda.synchronize()
                t_ = torch.tensor([(time.perf_counter() - t0_) / 20 * 1e6], dtype=torch.float64, device=dev)
                dist.all_reduce(t_, op=dist.ReduceOp.MAX)
                coll[name_] = float(t_.item())
            coll["MB"] = nb_ * 4 / 1e6
        return {"world": dist.get_world_size(), "ranks_seen": int(ones.item()), "backend": dist.get_backend(),
                "collective_us": coll,
                "rccl_version": ver, "devices": [int(d.item()) for d in devs],
                "bucket_MB": net.store.bucket.numel() * 4 / 1e6,
                "exchange": ("sharded: all_to_all_single of %.2f MB bucket slices + all_gather of %d clip factors + "
                             "in-place all_gather_into_tensor of the parameter slices"
                             % (net.store.npad * 4 / 1e6, world)) if getattr(opt, "sharded", False)
                else "one all_reduce(SUM) of the bucket [clipped grads | statistics]"}

    def weak_block():                # the other scaling mode as an extra key: 4096 rows on every rank
        wnet, wopt, winputs = make_learner(B, SEED + rank)
        wrun, _ = make_stepper(wnet, wopt, winputs)
        wel = timed_steps(wrun, args.steps, args.warmup)
        return {"value": world * args.steps / wel, "unit": "steps/s of 4096 transitions (4096 rows per GPU)",
                "ms_per_step": wel / args.steps * 1e3, "global_batch": B * world}

    def sharded_block():
        # the other exchange form on the same shard (PFPN_EXCHANGE=sharded makes it the headline): all-to-all of unscaled
        # bucket slices, owner-side clip factors + Adam on 1/W of the buffer, in-place all-gather
        xnet, xopt, xin = make_learner(rows, SEED + rank, exchange_mode="sharded")
        xrun, xlaunch = make_stepper(xnet, xopt, xin)
        xel = timed_steps(xrun, args.steps, args.warmup)
        return {"value": args.steps / xel, "unit": "steps/s", "ms_per_step": xel / args.steps * 1e3, "launch": xlaunch}

    def two_graph_block():
        # the fall-back form of the exchange (--two-graph-exchange makes it the headline): two hipGraphs around an eager all-reduce
        gnet, gopt, gin = make_learner(rows, SEED + rank, exchange_mode="allreduce")
        grun, glaunch = make_stepper(gnet, gopt, gin, capture_exchange=False)
        gel = timed_steps(grun, args.steps, args.warmup)
        return {"value": args.steps / gel, "unit": "steps/s", "ms_per_step": gel / args.steps * 1e3, "launch": glaunch}

    def extra(key, fn):
        res = guarded(fn)
        if rank == 0:
            out[key] = res                                          # at once: a later deadline still prints it

    if world > 1 or args.force_exchange:
        extra("rccl", rccl_block)                                   # two tiny collectives: first, so it is never lost
    if world == 1 and rank == 0 and strong:
        out["weak_scaling"] = {"value": out["value"], "unit": "steps/s of 4096 transitions (4096 rows per GPU)",
                               "ms_per_step": out["ms_per_step"], "global_batch": B,
                               "note": "one GPU: the headline IS the weak-scaling point (4096 rows per GPU)"}
    if world > 1 and strong:
        extra("weak_scaling", weak_block)
        if not getattr(opt, "sharded", False):
            extra("exchange_sharded", sharded_block)
        if dist.get_backend() == "nccl" and "captured inside" in launch and safe is None:
            extra("exchange_two_graphs", two_graph_block)

    # ---- N > 1: the roofline object of rank 0's shard (local launches only: no collective; the other ranks wait at the
    # closing barrier meanwhile).  The N = 1 line carries the full set (traffic, rocprof agreement, stress shape, GEMMs).
    if rank == 0 and world > 1:
        try:
            def shard_call():
                kernels.ppo_mixture_fwd_bwd(net.logits, action, net.loc, net.logstd, logp_old, adv, net.value_buf, value_old,
                                            net.adv_stats, 0.2, 0.5, out=net.train_out,
                                            dbias=net.store.g("actor/fc_policy/bias"), finalize=False)
            for _ in range(3):
                shard_call()
            torch.cuda.synchronize()
            side_ = torch.cuda.Stream()
            side_.wait_stream(torch.cuda.current_stream())
            g_ = torch.cuda.CUDAGraph()
            with torch.cuda.stream(side_):
                shard_call()
                torch.cuda.synchronize()
                with torch.cuda.graph(g_, stream=side_, **graphs._capture_mode()):
                    for _ in range(20):
                        shard_call()
            torch.cuda.synchronize()
            g_.replay()
            torch.cuda.synchronize()
            ea_, eb_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ea_.record()
            for _ in range(10):
                g_.replay()
            eb_.record()
            torch.cuda.synchronize()
            us_ = ea_.elapsed_time(eb_) * 1e3 / 200
            ab_ = algorithmic_bytes_fused(rows)
            out["roofline"] = {"kernel": "k_mix_tile<35,PPO>", "bound": "hbm", "achieved": ab_ / us_ / 1e3,
                               "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ab_ / us_ / 1e3 / HBM_PEAK_GBS, "traffic": None,
                               "algorithmic_bytes": ab_, "us_per_launch": us_, "launches_timed": 200, "rows": rows}
            del g_
        except Exception as e:
            out["roofline"] = {"error": repr(e)}

    # ---- roofline of the dominant hand-written kernel, and of the GEMMs (rank 0, N=1 only) ----------
    if rank == 0 and world == 1:
        logits = net.logits
        stats = net.adv_stats
        reps = 200

        def time_in_graph(call, per_graph=20, replays=5):
            """device time per call: `per_graph` calls captured in one hipGraph, replayed between two HIP events on the replay
            stream (eager back-to-back calls measure the host: a Python wrapper call costs more than most kernels here)"""
            for _ in range(3):
                call()
            torch.cuda.synchronize()
            side_ = torch.cuda.Stream()
            side_.wait_stream(torch.cuda.current_stream())
            g_ = torch.cuda.CUDAGraph()
            with torch.cuda.stream(side_):
                call()
                torch.cuda.synchronize()
                with torch.cuda.graph(g_, stream=side_, **graphs._capture_mode()):
                    for _ in range(per_graph):
                        call()
            torch.cuda.synchronize()
            g_.replay()
            torch.cuda.synchronize()
            ea_, eb_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ea_.record()
            for _ in range(replays):
                g_.replay()
            eb_.record()
            torch.cuda.synchronize()
            return ea_.elapsed_time(eb_) * 1e3 / (per_graph * replays)

        def time_fused(finalize):
            """device time per launch: 20 launches captured in one hipGraph and replayed 10 times between two HIP events on
            the replay stream.  (Eager back-to-back launches measure the HOST: one call of the Python wrapper costs
            ~16 us, more than the kernel since round 2.)"""
            def call():
                kernels.ppo_mixture_fwd_bwd(logits, action, net.loc, net.logstd, logp_old, adv, net.value_buf, value_old,
                                            stats, 0.2, 0.5, out=net.train_out, dbias=net.store.g("actor/fc_policy/bias"),
                                            finalize=finalize)
            for _ in range(5):
                call()
            torch.cuda.synchronize()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            g = torch.cuda.CUDAGraph()
            per_graph, replays = 20, reps // 20
            with torch.cuda.stream(side):
                call()
                torch.cuda.synchronize()
                with torch.cuda.graph(g, stream=side, **graphs._capture_mode()):
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
        traffic, traffic_source = None, None
        try:                                  # HBM bytes per launch from the committed rocprofv3 --pmc passes (FETCH_SIZE + WRITE_SIZE)
            tpath, rnd = latest_profile("pmc_traffic.json")
            with open(tpath) as f:
                traffic = json.load(f)["kernels"]["k_mix_tile<35,PPO>"]["traffic_bytes"]
            traffic_source = "profiles/%s_pmc_traffic.json: separate rocprofv3 --pmc passes of round %s over this " \
                             "command (tools/pmc_traffic.py), NOT measured in this run" % (rnd, rnd[1:])
        except Exception:
            pass
        # `frac` must not read higher than the rocprofv3 summary of this command says: the events time launches replayed
        # back to back from a hipGraph (a launch's start-up overlaps its predecessor's drain, no tracer attached), the
        # summary a traced kernel from begin to end (+~0.5 us of tracer).  Both are printed; `frac` is the LOWER one.
        rp = rocprof_kernel_us("k_mix_tile<35, 2>")
        frac_events = ach / HBM_PEAK_GBS
        frac_rocprof = (ab / (rp[0] * 1e-6) / 1e9 / HBM_PEAK_GBS) if rp else None
        frac = min(frac_events, frac_rocprof) if frac_rocprof else frac_events
        out["roofline"] = {"kernel": "k_mix_tile<35,PPO>", "bound": "hbm", "achieved": frac * HBM_PEAK_GBS,
                           "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": frac, "frac_events": frac_events,
                           "frac_rocprof": frac_rocprof,
                           "rocprof": ({"us_per_launch": rp[0], "calls": rp[1], "source": rp[2] + " (committed rocprofv3 "
                                        "--kernel-trace --stats summary of this command; NOT measured in this run)"} if rp else None),
                           "traffic": traffic,
                           "traffic_source": traffic_source, "algorithmic_bytes": ab, "us_per_launch": us, "launches_timed": reps,
                           "us_in_step_event_pairs": us_in_step, "us_with_slab_finalize_launch": us_op,
                           "note": "us_per_launch: HIP events around 200 launches of the kernel alone, replayed from a hipGraph "
                                   "so that the host's launch rate is out of the measurement (in "
                                   "the step its slab sums are finished by the single deferred reduce launch); "
                                   "us_in_step_event_pairs: an event pair around each call inside 100 eager learner "
                                   "steps (includes the dispatch gap); "
                                   "42 MB per launch, four tiles per block (~22 k cycles): ~5 k until the first tile has landed, ~3 k "
                                   "per tile -- VALU issue on the SIMD that hosts three of the nine compute waves -- and "
                                   "~2 k of tail (DESIGN.md section 4); rocprofv3 adds ~0.5 us to a traced kernel; "
                                   "the 1 GiB shape of BASELINE.json config 5 is in roofline_stress"}
        # the SAME kernel on four and sixteen times the rows (same A x N): how much of the 0.46 at 4096 rows is the fixed cost
        # of a block's start-up and tail (four tiles per block at 4096 rows, 16 / 64 at these sizes)
        try:
            if args.no_cpu_baseline:         # profiling runs: keep the kernel statistics to the headline shape
                raise RuntimeError("skipped with --no-cpu-baseline")
            scaled = {}
            for rows_ in (16384, 65536):
                gq = torch.Generator(device=dev); gq.manual_seed(3)
                lq = torch.randn(rows_, A * NP, device=dev, generator=gq)
                xq = torch.rand(rows_, A, device=dev, generator=gq) * 2.2 - 1.1
                lpq = kernels.logprob_fwd(lq, xq, net.loc, net.logstd).clone()
                aq = torch.randn(rows_, device=dev, generator=gq)
                vq = torch.randn(rows_, device=dev, generator=gq)
                sq_ = kernels.adv_stats(aq)
                oq = kernels.ppo_mixture_fwd_bwd(lq, xq, net.loc, net.logstd, lpq, aq, vq, vq, sq_, finalize=False)

                def callq():
                    kernels.ppo_mixture_fwd_bwd(lq, xq, net.loc, net.logstd, lpq, aq, vq, vq, sq_, out=oq, finalize=False)
                for _ in range(3):
                    callq()
                torch.cuda.synchronize()
                sideq = torch.cuda.Stream()
                sideq.wait_stream(torch.cuda.current_stream())
                gqr = torch.cuda.CUDAGraph()
                with torch.cuda.stream(sideq):
                    callq()
                    torch.cuda.synchronize()
                    with torch.cuda.graph(gqr, stream=sideq, **graphs._capture_mode()):
                        for _ in range(10):
                            callq()
                torch.cuda.synchronize()
                gqr.replay()
                torch.cuda.synchronize()
                e0.record()
                for _ in range(5):
                    gqr.replay()
                e1.record()
                torch.cuda.synchronize()
                uq = e0.elapsed_time(e1) * 1e3 / 50
                abq = algorithmic_bytes_fused(rows_)
                scaled[str(rows_)] = {"us_per_launch": uq, "algorithmic_bytes": abq, "achieved": abq / uq / 1e3, "unit": "GB/s",
                                      "frac": abq / uq / 1e3 / HBM_PEAK_GBS}
                del lq, xq, oq, gqr
            out["roofline_more_rows"] = {"kernel": "k_mix_tile<35,PPO>", "bound": "hbm", "peak": HBM_PEAK_GBS, "rows": scaled,
                                         "note": "the roofline kernel on 4x / 16x the metric's rows (same 36 x 35 mixture): the "
                                                 "block's start-up and tail are paid once per launch"}
        except Exception as ex:
            out["roofline_more_rows"] = {"error": repr(ex)}
        # the same mixture math on the 1 GiB stress shape (BASELINE.json config 5, one GPU's shard)
        try:
            sB, sA, sN = 8192, 128, 256
            g = torch.Generator(device=dev); g.manual_seed(1)
            # logits / dlogits placed as the network places them at this size (kernels.stream_pair: one allocation, the
            # second stream a fixed distance behind the first -- independent allocations time 380-485 us by luck of placement)
            sl, s_dl = kernels.stream_pair(sB, sA * sN, dev)
            sl.normal_(generator=g)
            s_bw = {"dlogits": s_dl, "dloc": torch.empty(sA, sN, device=dev), "dlogstd": torch.empty(sA, sN, device=dev)}
            sx = torch.rand(sB, sA, device=dev, generator=g) * 2.2 - 1.1
            sloc = torch.linspace(-1, 1, sN, device=dev).repeat(sA, 1).contiguous()
            sls = torch.full((sA, sN), float(np.log(2.0 / (sN - 1))), device=dev)
            sg = torch.randn(sB, device=dev, generator=g)
            L_, X_, V_, P_ = 4 * sB * sA * sN, 4 * sB * sA, 4 * sB, 8 * sA * sN
            res = {}
            for name, fn, nbytes in (("fwd", lambda: kernels.logprob_fwd(sl, sx, sloc, sls), L_ + X_ + V_ + P_),
                                     ("bwd", lambda: kernels.logprob_bwd(sl, sx, sloc, sls, sg, out=s_bw), 2 * L_ + X_ + 2 * V_ + 2 * P_)):
                fn(); fn()
                torch.cuda.synchronize()
                e0.record()
                for _ in range(30):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                u = e0.elapsed_time(e1) * 1e3 / 30
                res[name] = {"kernel": "k_mix_wide<1,%s> (its reduction inside the launch)" % name.upper(), "us_per_launch": u,
                             "algorithmic_bytes": nbytes, "achieved": nbytes / u / 1e3, "unit": "GB/s",
                             "peak": HBM_PEAK_GBS, "frac": nbytes / u / 1e3 / HBM_PEAK_GBS}
            # the PPO composition at this shape: forward -> loss head on [B] -> backward (3L: the row log-prob couples
            # all 128 segments of a row and neither a 128-KiB row plus the 393-KB bank nor 1 GiB of logits stays on chip
            # between the two passes)
            s_lp_old = kernels.logprob_fwd(sl, sx, sloc, sls).clone()
            s_stats = kernels.adv_stats(sg)
            s_out = {"logp": torch.empty(sB, device=dev), "dlogits": s_dl,
                     "dloc": torch.empty(sA, sN, device=dev), "dlogstd": torch.empty(sA, sN, device=dev),
                     "dvalue": torch.empty(sB, device=dev), "loss3": torch.empty(3, device=dev)}
            s_v = torch.zeros(sB, device=dev)

            def ppo_composed():
                kernels.ppo_mixture_fwd_bwd(sl, sx, sloc, sls, s_lp_old, sg, s_v, s_v, s_stats, out=s_out)

            def time_ppo():
                ppo_composed(); ppo_composed()
                torch.cuda.synchronize()
                e0.record()
                for _ in range(10):
                    ppo_composed()
                e1.record()
                torch.cuda.synchronize()
                return e0.elapsed_time(e1) * 1e3 / 10
            nb3, nb2 = 3 * L_ + 2 * X_ + 8 * V_ + 3 * P_, 2 * L_ + X_ + 6 * V_ + 2 * P_
            fused_default = kernels.WIDE_FUSED
            # the two-launch composition (3L: the backward reads the logits again) ...
            kernels.WIDE_FUSED = False
            try:
                u3 = time_ppo()
            finally:
                kernels.WIDE_FUSED = fused_default
            res["ppo_3l"] = {"kernel": "k_mix_wide<FWD + PPO head> + k_mix_wide<BWD> (two launches, results in the caller's buffers)",
                             "us_per_launch": u3, "algorithmic_bytes": nb3, "achieved": nb3 / u3 / 1e3, "unit": "GB/s",
                             "peak": HBM_PEAK_GBS, "frac": nb3 / u3 / 1e3 / HBM_PEAK_GBS}
            # ... and what the learner composition runs (round 5): ONE launch, the logits read once (2L), the waves of a row
            # chunk exchanging their per-dimension log p through the XCD's L2 once per four rows
            kernels.WIDE_FUSED = fused_default
            if fused_default:
                u2 = time_ppo()
                res["ppo"] = {"kernel": "k_mix_wide_f2l (forward + PPO head + backward in one launch, logits read once)",
                              "us_per_launch": u2, "algorithmic_bytes": nb2, "achieved": nb2 / u2 / 1e3, "unit": "GB/s",
                              "peak": HBM_PEAK_GBS, "frac": nb2 / u2 / 1e3 / HBM_PEAK_GBS,
                              "frac_on_3l_bytes": nb3 / u2 / 1e3 / HBM_PEAK_GBS, "speedup_vs_3l_form": u3 / u2,
                              "traffic": stress_traffic("k_mix_wide_f2l"),
                              "traffic_source": "profiles/rNN_pmc_traffic.json (tools/f2l_traffic.sh; NOT measured in this run)",
                              "note": "frac: on its own (2L) algorithmic bytes; frac_on_3l_bytes: the bytes of the two-launch "
                                      "form it replaces over this launch's time"}
            else:
                res["ppo"] = dict(res["ppo_3l"])
            del s_out
            # this box's roof for the two access patterns, in the same run, from the build's OWN streaming kernels
            # (pfpn_stream_copy: float4 grid-stride; the few launch shapes that won tools/copy_probe.py's sweep, best taken):
            # read + write of the 1 GiB logits (what the backward streams) and the read-only stream (the forward)
            def best_of(fn_of_cfg, cfgs, nbytes, reps=8):
                best_ = None
                for cfg in cfgs:
                    f_ = fn_of_cfg(*cfg)
                    f_(); f_()
                    torch.cuda.synchronize()
                    e0.record()
                    for _ in range(reps):
                        f_()
                    e1.record()
                    torch.cuda.synchronize()
                    u_ = e0.elapsed_time(e1) * 1e3 / reps
                    if best_ is None or u_ < best_[0]:
                        best_ = (u_, cfg)
                return {"us_per_launch": best_[0], "algorithmic_bytes": nbytes, "achieved": nbytes / best_[0] / 1e3,
                        "unit": "GB/s", "peak": HBM_PEAK_GBS, "frac": nbytes / best_[0] / 1e3 / HBM_PEAK_GBS,
                        "config": {"unroll": best_[1][0], "nontemporal": best_[1][1], "blocks": best_[1][2],
                                   **({"destination": ("independent allocation", "paired buffer")[best_[1][3]]}
                                      if len(best_[1]) > 3 else {})}}
            # (destination: an independent allocation AND the paired buffer the backward writes -- placement moves a copy
            # by ~10 % too, and not the way it moves the backward; the calibration takes its own best)
            s_cp = torch.empty_like(sl)
            dsts = (s_cp.view(-1), s_dl.view(-1))
            res["device_copy"] = dict(best_of(lambda u_, nt_, bl_, d_: (lambda: kernels.stream_copy(dsts[d_], sl, u_, nt_, bl_)),
                                              [(u_, nt_, bl_, d_) for d_ in (0, 1) for (u_, nt_, bl_) in
                                               ((1, 1, 32768), (1, 1, 1024), (2, 1, 16384), (1, 0, 1024), (4, 0, 2048))], 2 * L_),
                                      kernel="pfpn_stream_copy (own float4 grid-stride copy of the same logits: read L + write L)")
            del s_cp, dsts
            sinks = {bl_: torch.empty(bl_ * 256, device=dev) for bl_ in (2048, 8192, 32768)}
            res["device_read"] = dict(best_of(lambda u_, nt_, bl_: (lambda: kernels.stream_copy(sinks[bl_], sl, 0, nt_, bl_)),
                                              [(0, 1, 2048), (0, 1, 8192), (0, 1, 32768), (0, 0, 8192)], L_),
                                      kernel="pfpn_stream_copy, read-only (own float4 grid-stride sum of the same logits: read L)")
            for k_, ref_ in (("fwd", "device_read"), ("bwd", "device_copy"), ("ppo", "device_copy"), ("ppo_3l", "device_copy")):
                res[k_]["frac_of_box_copy"] = res[k_]["achieved"] / res[ref_]["achieved"]
                res[k_]["box_copy"] = ref_
            res["note"] = ("frac: of the 8 TB/s spec; frac_of_box_copy: of what this box's memory system moves for the same "
                           "access pattern with NO arithmetic (device_read for the forward, device_copy for the backward and "
                           "the PPO passes).  fwd is bound by its arithmetic -- two v_exp_f32 per particle: with the rows "
                           "already in LDS (LDS-DMA ring, tools/bw_probe8.hip) the same math takes ~203 us")
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
        try:      # counter evidence: share of the GEMM kernels' cycles in which the matrix pipe was busy (one --pmc pass, committed)
            mpath, mrnd = latest_profile("gemm_mfma.json")
            with open(mpath) as f:
                mj = json.load(f)
            per = {k[:72]: round(v["mfma_busy"], 3) for k, v in mj["kernels"].items() if "mfma_busy" in v}
            out["roofline_gemm"]["mfma_busy"] = mj["mfma_busy_all_gemm_kernels"]
            out["roofline_gemm"]["mfma_busy_per_kernel"] = per
            out["roofline_gemm"]["mfma_busy_source"] = (
                "profiles/%s_gemm_mfma.json: rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA over this "
                "command (tools/gemm_mfma_pmc.py): MFMA-busy cycles per SIMD / kernel cycles, launch-weighted over the step's "
                "GEMM kernels; NOT measured in this run.  frac is flops / time / 157.3 TFLOP/s (the peak at the 2.4 GHz the "
                "chip does not hold under MFMA load); mfma_busy is cycle-based, so the clock drops out" % mrnd)
        except Exception:
            pass
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
            rrun, rlaunch = make_stepper(net, opt, (rt[0], rt[1], r_v, r_lp, rt[2]))      # K steps per replay, like the headline
            rrun.prepare(16); rrun.prepare(208)
            rrun(16)
            torch.cuda.synchronize()
            tr = time.perf_counter()
            rrun(208)
            torch.cuda.synchronize()
            rus = (time.perf_counter() - tr) / 208 * 1e6
            out["reference_schedule"] = {"global_batch": 256, "steps_per_4096_transitions": 16, "us_per_step": rus,
                                         "transitions_per_s": 256 / (rus * 1e-6), "launch": rlaunch,
                                         "note": "the headline step consumes 4096 transitions per step: %.0f transitions/s"
                                                 % (B * out["value"])}
        except Exception as e:          # informational only
            out["reference_schedule"] = {"error": repr(e)}
        extras = not args.no_cpu_baseline
        # ---- strong_shard: what ONE rank of a 2 / 4 / 8-GPU run of this metric spends per step (the data-parallel step
        # on 2048 / 1024 / 512 rows: per-replica clip, bucket all-reduce through RCCL in a world-size-1 group, Adam with
        # 1/W, captured as two hipGraphs) -- the compute + launch side of the scaling curve, no link time in it
        if extras and not args.force_exchange:
            try:
                if not dist.is_initialized():
                    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                    os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 2000))
                    dist.init_process_group(backend=args.backend, rank=0, world_size=1)
                shard = {}
                for n_gpu in (1, 2, 4, 8):
                    entry = {"gpus_of_the_run_it_stands_for": n_gpu}
                    for mode in ("allreduce", "sharded", "allreduce_two_graphs"):
                        if mode == "allreduce_two_graphs" and dist.get_backend() != "nccl":
                            continue
                        snet, sopt, sin = make_learner(B // n_gpu, SEED, exchange_group=True, exchange_mode=mode.split("_")[0])
                        srun, sl_ = make_stepper(snet, sopt, sin, capture_exchange=False if mode == "allreduce_two_graphs" else None)
                        srun(20)
                        torch.cuda.synchronize()
                        t_s = time.perf_counter()
                        srun(200)
                        torch.cuda.synchronize()
                        us_s = (time.perf_counter() - t_s) / 200 * 1e6
                        if mode == "allreduce":
                            slaunch = sl_
                            entry.update({"us_per_step": us_s, "steps_per_s_without_links": 1e6 / us_s})
                        elif mode == "allreduce_two_graphs":
                            # (the fall-back form: two hipGraphs around an eagerly enqueued all-reduce)
                            entry["us_per_step_two_graph_exchange"] = us_s
                        else:
                            # (world size 1: the owner's slice is the whole buffer -- the launch structure and the three
                            # collectives' host cost, not the 1/W Adam of a real run)
                            entry["us_per_step_sharded_exchange"] = us_s
                        del snet, sopt, sin, srun
                    shard[str(B // n_gpu)] = entry
                bucket_mb = (net.store.bucket.numel() * 4) / 1e6
                base_ = shard[str(B)]["us_per_step"]
                out["strong_shard"] = {"rows_per_gpu": shard, "launch": slaunch, "bucket_MB": bucket_mb,
                                       "compute_side_speedup_at_8_gpus": base_ / shard[str(B // 8)]["us_per_step"],
                                       "note": "one rank's step with the RCCL call in it but nothing on the links (world size "
                                               "1); an N-GPU strong-scaling step = the N-row entry + the all-reduce of the "
                                               "%.2f MB bucket over xGMI.  compute_side_speedup_at_8_gpus is the ceiling of "
                                               "1 -> 8 GPU strong scaling BEFORE any link time: >= 3.5x is arithmetically out "
                                               "of reach for a 0.46-ms step (DESIGN.md section 6); weak_scaling is the mode "
                                               "that scales" % bucket_mb}
            except Exception as e:
                out["strong_shard"] = {"error": repr(e)}
        # ---- sac_step: BASELINE.json config 3 (SAC-PFPN, replay batch 1024; 256 = the reference's own batch,
        # settings/deepmimic/deepmimic_sac_base.py:4-12) as one hipGraph; parity configs, timed here so their rates and
        # rooflines are bench evidence and not prose
        if extras:
            try:
                from pfpn_amd.networks import ParticleFilteringSACNetwork
                sopts = dict(NET_OPTS, normalize_advantage=False, resample_interval=12000)
                AN_ = A * NP
                # multiply-accumulates per replay row (MLP 1024 / 512): the actor runs forward over [s; s'] (2 rows per replay
                # row) and backward over s; the twin live critics forward over [(s, a_pi); (s, a_hist)], the twin targets over
                # (s', a'); the live critics' dX chain over both halves (first layer: the 36 action columns of the policy half
                # only), their dW from the value-loss half
                a_f, a_dw, a_dx = dense_macs(S, (1024, 512), AN_)
                q_f, q_dw, q_dx = dense_macs(S + A, (1024, 512), 1)
                sac_mac = 2 * a_f + (a_dw + a_dx) + 2 * (3 * q_f) + 2 * (2 * q_dx + A * 1024 + q_dw)
                sac = {}
                for SB in (1024, 256):
                    snet = ParticleFilteringSACNetwork(trainable=True, state_shape=[S], action_shape=[A],
                                                       action_lower_bound=-np.ones(A), action_upper_bound=np.ones(A),
                                                       device=dev, seed=SEED, **sopts)
                    snet.init()
                    for op in snet.init_ops:
                        op()
                    sopt = SyncReplicasApplier(snet, [{"names": snet.critic_param_names(), "lr": 1e-4},
                                                      {"names": snet.actor_param_names(), "lr": 1e-4}], norm_clip=1.0, world_size=1)
                    srng = np.random.default_rng(SEED + 3)
                    sb = [torch.from_numpy(a_).to(dev) for a_ in (
                        (srng.normal(0, 1, (SB, S)) * 2 + 0.3).astype(np.float32), np.tanh(srng.normal(0, 1, (SB, A))).astype(np.float32),
                        srng.random(SB).astype(np.float32), (srng.random(SB) < 0.98).astype(np.float32),
                        (srng.normal(0, 1, (SB, S)) * 2 + 0.3).astype(np.float32))]
                    SK = max(1, args.steps_per_graph)      # learner steps per replay, like the headline
                    sstep = graphs.GraphedLearnerStep(snet, sopt, *sb, noise=snet.make_train_noise(SB), debug_nodes=True,
                                                      steps_per_replay=SK)
                    for _ in range(4):
                        sstep()
                    torch.cuda.synchronize()
                    sms = wall_ms(sstep, 10) / SK
                    # the same step as ONE step per replay with its noise drawn by a launch of its own between replays -- the
                    # form the off-policy worker runs (one learner step per environment step, `workers/ddpg.py`)
                    s1 = graphs.GraphedLearnerStep(snet, sopt, *sb, noise=snet.make_train_noise(SB))
                    for _ in range(5):
                        s1()
                    torch.cuda.synchronize()
                    sms1 = wall_ms(s1, 40)
                    del s1
                    s_nodes = sstep.kernel_nodes()         # (None where the runtime cannot print a captured graph: this image)
                    s_launches, s_src = (s_nodes / SK, "kernel nodes of the captured %d-step hipGraph / %d" % (SK, SK)) \
                        if s_nodes else timeline_launches("sac_step_timeline.txt")
                    if not s_nodes and s_src:
                        s_src += " (committed rocprofv3 --kernel-trace of tools/sac_trace.py at replay batch 1024; NOT counted in this run)"
                    sl3 = sstep.loss3.cpu().numpy()
                    sflops = 2.0 * SB * sac_mac
                    # the step's hand-written actor-side launches on their own (replayed from a hipGraph): sample + log-prob +
                    # critic inputs over the 2B rows [s; s'] (reads logits and Gumbel draws: 4L, L = one B-row tensor) and
                    # their backward over B rows (reads logits, Gumbel, normal draws, writes dlogits: 4L)
                    Ls = 4 * SB * AN_
                    nz_ = sstep.noise["_buf"]
                    us_rf = time_in_graph(lambda: kernels.sac_actor_fwd(
                        snet.logits_both, snet.loc, snet.logstd, nz_[0:2].view(2 * SB, -1), nz_[2:4].view(2 * SB, -1),
                        x=snet.xn_both, a_hist=sb[1], xq_both=snet.xq_both, xq_next=snet.xa2))
                    da_ = torch.zeros(SB, A, device=dev)
                    s_smp = snet.last["s_"].contiguous()
                    us_rb = time_in_graph(lambda: kernels.sac_actor_bwd(snet.logits, snet.loc, snet.logstd, nz_[0], nz_[2], s_smp,
                                                                        snet.sac_out["dlogp"], da_, snet.dlogits, snet.ws_actor))
                    sac[SB] = {"steps_per_s": 1e3 / sms, "ms_per_step": sms, "ms_per_step_one_step_per_replay": sms1,
                               "finite_loss": bool(np.all(np.isfinite(sl3))),
                               "roofline": {"bound": "mfma", "gemm_flops": sflops, "achieved": sflops / (sms * 1e-3) / 1e12,
                                            "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s", "frac": sflops / (sms * 1e-3) / 1e12 / MFMA_F32_PEAK_TF,
                                            "launches_per_step": s_launches,
                                            "launches_source": s_src,
                                            "noise_and_next_prologue_ride_in_the_optimiser_launch": sstep.noise_in_graph,
                                            "note": "frac = the step's GEMM flops / its WHOLE time / the fp32 MFMA peak: a step of ~%d "
                                                    "launches at ~5 us each is launch-bound, not MFMA-bound, at these row counts"
                                                    % round(s_launches or 28),
                                            "hbm_kernels": {
                                                "k_sac_actor_fwd (2B rows)": {"us": us_rf, "algorithmic_bytes": 4 * Ls,
                                                                              "frac": 4 * Ls / us_rf / 1e3 / HBM_PEAK_GBS},
                                                "k_sac_actor_bwd": {"us": us_rb, "algorithmic_bytes": 4 * Ls,
                                                                    "frac": 4 * Ls / us_rb / 1e3 / HBM_PEAK_GBS},
                                                "note": "latency-bound single-pass launches (stage, compute, write): 7-13 us each, "
                                                        "not HBM-bound at these sizes"}}}
                    del snet, sopt, sstep
                out["sac_step"] = dict(sac[1024], workload="DeepMimicKick-v0 SAC-PFPN 35 particles, replay batch 1024 (BASELINE.json "
                                                            "config 3), synthetic replay rows",
                                       launch="hipGraph, %d learner steps per replay, noise drawn inside the graph" % SK,
                                       batch_256=sac[256])
            except Exception as e:
                out["sac_step"] = {"error": repr(e)}
        # ---- config5_step: BASELINE.json config 5 as a learner step on ONE GPU's shard (8192 rows over 8 GPUs = 1024 rows
        # per GPU): 128 action dims x 256 particles, head 512 -> 32768, as one hipGraph
        if extras:
            try:
                cA, cN, cB = 128, 256, 1024
                copts = dict(NET_OPTS, particles=cN)
                cnet = ParticleFilteringClipPPONetwork(trainable=True, state_shape=[S], action_shape=[cA],
                                                       action_lower_bound=-np.ones(cA), action_upper_bound=np.ones(cA),
                                                       device=dev, seed=SEED, **copts)
                cnet.init()
                copt = SyncReplicasApplier(cnet, [{"names": None, "lr": 1e-4}], norm_clip=1.0, world_size=1)
                crng = np.random.default_rng(SEED + 5)
                ct = {"state": torch.from_numpy((crng.normal(0, 1, (cB, S)) * 2 + 0.3).astype(np.float32)).to(dev),
                      "action": torch.from_numpy(crng.uniform(-1, 1, (cB, cA)).astype(np.float32)).to(dev),
                      "advantage": torch.from_numpy(crng.normal(0, 1, cB).astype(np.float32)).to(dev)}
                with torch.no_grad():
                    cp, _, chc = cnet._forward(ct["state"])
                    c_lp = cp.log_prob(ct["action"]).clone()
                    c_v = cnet._value(chc).clone()
                cargs = (ct["state"], ct["action"], c_v, c_lp, ct["advantage"])
                for _ in range(5):
                    cnet.train_device(copt, *cargs)
                torch.cuda.synchronize()
                CK = max(1, args.steps_per_graph)       # learner steps per replay, like the headline
                claunch = "hipGraph, %d learner steps per replay" % CK
                try:
                    cstep = graphs.GraphedLearnerStep(cnet, copt, *[t.clone() for t in cargs], debug_nodes=True,
                                                      steps_per_replay=CK)
                except Exception:                       # (capture refused: time the eager launches)
                    CK, claunch, cstep = 1, "eager", (lambda: cnet.train_device(copt, *cargs))
                for _ in range(3):
                    cstep()
                torch.cuda.synchronize()
                cms = wall_ms(cstep, 4) / CK
                cl3 = cnet.train_out["loss3"].cpu().numpy()
                mac = S * 2048 + 2 * 1024 * 512 + 512 * cA * cN
                flops = 2.0 * cB * (mac + S * 2048 + 2 * 2 * 1024 * 512 + 2 * 512 * cA * cN)
                # the step's HBM-bound kernels on their own at this shard's shape (128 MiB of logits: inside the infinity cache)
                cL, cX, cV, cP = 4 * cB * cA * cN, 4 * cB * cA, 4 * cB, 8 * cA * cN
                us_cf = time_in_graph(lambda: kernels.logprob_fwd(cnet.logits, ct["action"], cnet.loc, cnet.logstd,
                                                                  out=cnet.train_out["logp"]), per_graph=10)
                cdl = torch.randn(cB, device=dev)
                us_cb = time_in_graph(lambda: kernels.logprob_bwd(cnet.logits, ct["action"], cnet.loc, cnet.logstd, cdl,
                                                                  out=cnet.train_out), per_graph=10)
                # ... and the one-launch PPO pass the step actually runs at this shape (the memset node in front of it included)
                us_c2 = None
                if kernels.WIDE_FUSED:
                    c_stats = kernels.adv_stats(ct["advantage"])
                    us_c2 = time_in_graph(lambda: kernels.ppo_mixture_fwd_bwd(
                        cnet.logits, ct["action"], cnet.loc, cnet.logstd, c_lp, ct["advantage"], c_v, c_v,
                        c_stats, out=cnet.train_out, dbias=cnet.store.g("actor/fc_policy/bias")), per_graph=10)
                # launches per step: the runtime's dump of the captured graph where it can print one, else the committed
                # rocprofv3 timeline of tools/config5_trace.py (profiles/rNN_config5_step_timeline.txt; NOT counted in this run)
                c5_launches = ((cstep.kernel_nodes() or 0) / CK or None) if claunch != "eager" else None
                c5_src = "hipGraph node dump of this run"
                if c5_launches is None:
                    c5_launches, c5_src = timeline_launches("config5_step_timeline.txt")
                out["config5_step"] = {"workload": "128 action dims x 256 particles, %d rows (one GPU's shard of the 8192-row "
                                                   "batch over 8 GPUs), head 512 -> 32768" % cB,
                                       "ms_per_step": cms, "steps_per_s": 1e3 / cms, "launch": claunch, "gemm_flops": flops,
                                       "finite_loss": bool(np.all(np.isfinite(cl3))),
                                       "roofline": {"bound": "mfma", "gemm_flops": flops, "achieved": flops / (cms * 1e-3) / 1e12,
                                                    "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s",
                                                    "frac": flops / (cms * 1e-3) / 1e12 / MFMA_F32_PEAK_TF,
                                                    "launches_per_step": c5_launches, "launches_source": c5_src,
                                                    "note": "frac = the step's GEMM flops / its WHOLE time / the fp32 MFMA peak",
                                                    "hbm_kernels": {
                                                        "k_mix_wide<FWD> (row sums inside)": {
                                                            "us": us_cf, "algorithmic_bytes": cL + cX + cV + cP,
                                                            "frac": (cL + cX + cV + cP) / us_cf / 1e3 / HBM_PEAK_GBS},
                                                        "k_mix_wide<BWD> (slab sums inside)": {
                                                            "us": us_cb, "algorithmic_bytes": 2 * cL + cX + 2 * cV + 2 * cP,
                                                            "frac": (2 * cL + cX + 2 * cV + 2 * cP) / us_cb / 1e3 / HBM_PEAK_GBS},
                                                        **({"k_mix_wide_f2l (forward + PPO head + backward, one launch)": {
                                                            "us": us_c2, "algorithmic_bytes": 2 * cL + cX + 6 * cV + 2 * cP,
                                                            "frac": (2 * cL + cX + 6 * cV + 2 * cP) / us_c2 / 1e3 / HBM_PEAK_GBS}}
                                                           if us_c2 else {}),
                                                        "note": "128 MiB of logits fit the 256 MiB infinity cache: a fraction above "
                                                                "the HBM figure of the 1 GiB shape is cache, not HBM"}}}
                del cnet, copt
            except Exception as e:
                out["config5_step"] = {"error": repr(e)}
        if extras:
            out["cpu_baseline"] = cpu_baseline()
            rs = out["cpu_baseline"].get("runs", {}).get("reference_schedule", {})
            if "transitions_per_s" in rs and "transitions_per_s" in out.get("reference_schedule", {}):
                out["reference_schedule"]["cpu_transitions_per_s"] = rs["transitions_per_s"]
                out["reference_schedule"]["cpu_cores"] = rs["cores"]
    # The JSON line must be the LAST thing on stdout.  RCCL leaves a version banner in the C stdio buffer of every rank,
    # which would otherwise reach the pipe at process exit -- after a line printed from Python.  So every rank flushes
    # its C streams, the ranks meet at a barrier, the group is torn down, and only then rank 0 prints.
    if dist.is_initialized():
        dist.barrier()
        _flush_c_stdio()
        dist.barrier()                      # every rank has flushed its banner
        from pfpn_amd.models.distributed_model import shutdown_process_group
        shutdown_process_group()            # hipGraphs that captured collectives first, then the group (pfpn_amd/graphs.py)
    _flush_c_stdio()
    if deadline is not None and not deadline.finish():
        return                              # the deadline fired meanwhile: it prints (rank 0) and ends the process
    if rank == 0:
        print_line(out)


if __name__ == "__main__":
    main()

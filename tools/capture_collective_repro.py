#!/usr/bin/env python3
"""Minimal repro of the round-3 finding "a captured all_to_all_single / all_gather_into_tensor hangs destroy_process_group"
(pfpn_amd/graphs.py) on a world-size-1 RCCL group, and of what cures it.

Every case runs in its OWN child process with a hard timeout (a hang is a host-side wait inside ncclCommDestroy /
the process group's shutdown, never a GPU hang):

  capture <collective> into a hipGraph -> replay it 3 times -> synchronize -> [cure] -> destroy_process_group -> exit 0

  cures:  none          the graph object is still alive at destroy
          del_graph     the CUDAGraph is deleted (hipGraphExecDestroy + hipGraphDestroy: RCCL's persistent-plan references
                        are dropped by the graph's user-object destructors) + gc + synchronize, THEN destroy
          no_destroy    never call destroy_process_group (plain interpreter exit)

usage (GPU box):  python tools/capture_collective_repro.py            # the table
                  python tools/capture_collective_repro.py --case all_to_all none     # one case, in-process
"""
import os
import subprocess
import sys
import time

CASES = [(c, cure) for c in ("all_reduce", "all_to_all", "all_gather") for cure in ("none", "del_graph", "no_destroy")]


def one(coll, cure):
    import gc
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29600 + os.getpid() % 2000))
    dist.init_process_group(backend="nccl", rank=0, world_size=1)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    n = 2_107_648                                    # the learner's padded bucket (8.43 MB)
    a, b = torch.ones(n, device=dev), torch.zeros(n, device=dev)
    fn = {"all_reduce": lambda: dist.all_reduce(a),
          "all_to_all": lambda: dist.all_to_all_single(b, a),
          "all_gather": lambda: dist.all_gather_into_tensor(a, a[:n])}[coll]
    fn()                                             # communicator + first-use set-up outside the capture
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
        fn()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    print("captured + replayed", coll, flush=True)
    if cure == "del_graph":
        del g
        gc.collect()
        torch.cuda.synchronize()
    if cure != "no_destroy":
        t0 = time.perf_counter()
        dist.destroy_process_group()
        print("destroy_process_group returned after %.2f s" % (time.perf_counter() - t0), flush=True)
    print("done", flush=True)


def main():
    if len(sys.argv) >= 4 and sys.argv[1] == "--case":
        one(sys.argv[2], sys.argv[3])
        return
    limit = float(os.environ.get("REPRO_TIMEOUT", "75"))
    print("%-12s %-11s %s" % ("collective", "cure", "outcome"))
    for coll, cure in CASES:
        t0 = time.perf_counter()
        p = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--case", coll, cure],
                             stdout=subprocess.PIPE, stderr=subprocess.STDOUT, start_new_session=True)
        try:
            out, _ = p.communicate(timeout=limit)
            tail = [l for l in out.decode().splitlines() if l.strip()][-1:] or [""]
            res = "exit %d after %.1f s (%s)" % (p.returncode, time.perf_counter() - t0, tail[0][:80])
        except subprocess.TimeoutExpired:
            os.killpg(p.pid, 9)                      # exactly the process group this script started
            out, _ = p.communicate()
            seen = [l for l in out.decode().splitlines() if l.strip()][-1:] or [""]
            res = "HUNG (> %.0f s; last line: %s)" % (limit, seen[0][:80])
        print("%-12s %-11s %s" % (coll, cure, res), flush=True)


if __name__ == "__main__":
    main()

"""What state does a FAILED stream capture leave behind?  (R5.15: one full-suite run lost test_sharded_exchange_over_rccl_single_rank to a
hipBLASLt call that failed inside a capture; the fall-back's first device operation then failed too.)  A capture is broken on purpose
(an illegal call inside it), with and without an RCCL collective captured before the failure; then the process tries to go on."""
import os, sys
import torch
import torch.distributed as dist

def attempt(tag, with_collective, breaker):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    x = torch.zeros(1024, device="cuda")
    try:
        with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
            x += 1
            if with_collective:
                dist.all_reduce(x)
            breaker(x)
            x += 1
        print(tag, ": capture did NOT fail", flush=True)
    except Exception as e:
        print(tag, ": capture failed:", str(e).split("\n")[0], flush=True)
    for name, op in (("tensor on the device", lambda: torch.tensor([1], device="cuda")),
                     ("synchronize", torch.cuda.synchronize),
                     ("kernel on the default stream", lambda: x.add_(1)),
                     ("eager all-reduce", (lambda: dist.all_reduce(x)) if with_collective else (lambda: None)),
                     ("synchronize again", torch.cuda.synchronize)):
        try:
            op()
            print(tag, ":   ", name, "ok", flush=True)
        except Exception as e:
            print(tag, ":   ", name, "FAILED:", str(e).split("\n")[0], flush=True)

os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("nccl", rank=0, world_size=1)
torch.cuda.set_device(0)
w = torch.zeros(8, device="cuda"); dist.all_reduce(w); torch.cuda.synchronize()
which = sys.argv[1] if len(sys.argv) > 1 else "sync"
breakers = {"sync": lambda x: torch.cuda.synchronize(), "item": lambda x: x[0].item(),
            "malloc": lambda x: torch.cuda.caching_allocator_alloc(1 << 20) and torch.cuda.memory.empty_cache(),
            "raise": lambda x: (_ for _ in ()).throw(RuntimeError("python exception inside the capture"))}
attempt(which + ", no collective", False, breakers[which])
attempt(which + ", collective captured first", True, breakers[which])
print("done", flush=True)
dist.destroy_process_group()

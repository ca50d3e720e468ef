"""RCCL smoke test on ONE GPU: world_size-1 process group over the nccl backend, the learner's bucket all-reduce and a
barrier.  (Two ranks cannot share a GPU under RCCL, so the W > 1 arithmetic is covered by the gloo tests; this checks
that the backend initialises and runs on the box.)  Launch: python -m torch.distributed.run --standalone
--local-addr 127.0.0.1 --nproc-per-node 1 tools/rccl_smoke.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1"); os.environ.setdefault("LOCAL_RANK", "0")
dist.init_process_group("nccl", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
dev = torch.device("cuda", int(os.environ["LOCAL_RANK"])); torch.cuda.set_device(dev)
bucket = torch.randn(2104517 + 2 * 197 + 2 * 1260 + 3, device=dev)
ref = bucket.clone()
for _ in range(3):
    dist.all_reduce(bucket)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    dist.all_reduce(bucket)
torch.cuda.synchronize()
print("rccl all_reduce(world=1) of %.2f MB: %.1f us per call, unchanged=%s" % (
    bucket.numel() * 4 / 1e6, (time.perf_counter() - t0) / 20 * 1e6, torch.equal(bucket, ref)))
dist.barrier(); dist.destroy_process_group()

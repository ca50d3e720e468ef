"""what does the runtime's dump of a captured hipGraph look like? (feeds GraphedLearnerStep.kernel_nodes)"""
import torch, tempfile, os
x = torch.zeros(1024, device="cuda"); y = torch.zeros(1024, device="cuda")
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
g = torch.cuda.CUDAGraph(); g.enable_debug_mode()
with torch.cuda.stream(s):
    x.add_(1); torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        x.add_(1); y.copy_(x); y.mul_(2)
d = tempfile.mkdtemp(); p = os.path.join(d, "g.dot")
g.debug_dump(p)
print(open(p).read()[:3000])

"""hipGraph capture of the launch-bound learner step.

One DPPO learner step is ~40 short launches (GEMMs + HIP kernels); at a 4096-row batch the host
dispatch cost is comparable to the device time, so the whole device part of the step
(compute_gradients + clip + TF-Adam + statistics) is captured once into a hipGraph and replayed.
Host-side bookkeeping (global_step, the resample tick every `resample_interval` steps) stays outside
the graph: the resample kernels run eagerly between replays.
"""
import torch


class GraphedLearnerStep(object):
    def __init__(self, net, optimizer, state, action, value, log_prob, advantage, warmup=3):
        assert optimizer.world == 1, "graph capture covers the single-GPU step (RCCL runs eagerly)"
        self.net, self.opt = net, optimizer
        self.inputs = (state, action, value, log_prob, advantage)
        st = net.store
        # warm up on a side stream (lazy hipBLASLt/workspace init), then restore every piece of state
        saved = [t.clone() for t in self._state_tensors()]
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(warmup):
                net.compute_gradients(*self.inputs)
                optimizer.apply_device(net)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        self.capture_stream = s
        for t, sv in zip(self._state_tensors(), saved):
            t.copy_(sv)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, stream=s):      # same stream as the warm-up: per-stream workspaces exist
            self.loss3 = net.compute_gradients(*self.inputs)
            optimizer.apply_device(net)
        # capture does not execute; state is untouched

    def _state_tensors(self):
        net, opt, st = self.net, self.opt, self.net.store
        ts = [st.flat, st.m, st.v, st.bucket, net.step_dev] + [g["beta_pow"] for g in opt.groups]
        if net.state_normalizer is not None:
            ts += [net.state_normalizer.mean, net.state_normalizer.std]
        if net.max_active is not None:
            ts += [net.max_active, net.sum_active]
        return ts

    def __call__(self, batch=None):
        if batch is not None:
            for dst, src in zip(self.inputs, batch):
                dst.copy_(src, non_blocking=True)
        self.graph.replay()
        self.opt.apply_host(self.net)
        return self.loss3

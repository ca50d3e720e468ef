"""Measured error of the HIP learner step against the graph-evaluated golden fixtures (tests/golden/graph_*.npz):
per tensor, max |err| / max |ref| and the largest per-element relative error.  Feeds the tolerances written in
tests/test_gpu_graph_golden.py.  Usage (GPU box): python tools/parity_report.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import fixture_inputs as fi          # noqa: E402
import test_gpu_graph_golden as gg   # noqa: E402


def stat(name, got, ref):
    got = got.detach().cpu().numpy() if isinstance(got, torch.Tensor) else np.asarray(got)
    got, ref = np.asarray(got, np.float64).reshape(np.shape(ref)), np.asarray(ref, np.float64)
    fin = np.isfinite(ref)
    err = np.abs(got[fin] - ref[fin])
    scale = np.abs(ref[fin]).max()
    with np.errstate(divide="ignore", invalid="ignore"):
        rel = np.where(np.abs(ref[fin]) > 0, err / np.abs(ref[fin]), 0.0)
    n_bad = int((err > 5e-5 * scale).sum())
    print("  %-44s scaled %.2e   per-elem rel max %.2e   >5e-5*scale: %d / %d" % (name, err.max() / scale, rel.max(), n_bad, err.size))


def main():
    dev = torch.device("cuda:0")
    tags = sys.argv[1:] or ("step_b32", "step_b256", "guard_b32", "resample_b32", "step_b4096")
    for tag in tags:
        fx = gg.load(tag)
        net, opt = gg.build(dev, fx)
        if bool(fx["fired"]):
            u_res = torch.from_numpy(fx["resample:u_resample"]).to(dev)
            u_noise = torch.from_numpy(fx["resample:u_noise_slot"]).to(dev)
            orig = net.resample_now
            net.resample_now = lambda: orig(u_res, u_noise)
        b = fi.batch_of(fx, 0)
        loss, _ = net.train(None, opt, [], b["state"], b["action"], b["value"], b["log_prob"], b["advantage"])
        print(tag)
        stat("logp", net.train_out["logp"], fx["out0:logp"])
        stat("value", net.value_buf, fx["out0:value"])
        stat("losses", [loss[0], loss[2], loss[3]], [fx["out0:loss"], fx["out0:policy_loss"], fx["out0:value_loss"]])
        G = net.gradients()
        Pn, Mn, Vn = net.variables(), net.variables(net.store.views_of(net.store.m)), net.variables(net.store.views_of(net.store.v))
        for k in fi.TRAINABLES:
            stat("grad " + k, fi.subsample(k, G[k].cpu().numpy().reshape(fi.SHAPES[k])), fx["out0:grad:" + k])
        for k in fi.TRAINABLES:
            stat("P " + k, fi.subsample(k, Pn[k].cpu().numpy().reshape(fi.SHAPES[k])), fx["P:" + k])
        for k in fi.TRAINABLES:
            stat("V " + k, fi.subsample(k, Vn[k].cpu().numpy().reshape(fi.SHAPES[k])), fx["V:" + k])
        stat("norm mean", net.state_normalizer.mean, fx["post:state_normalizer/mean"])
        stat("norm std", net.state_normalizer.std, fx["post:state_normalizer/std"])


if __name__ == "__main__":
    main()

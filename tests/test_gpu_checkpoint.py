"""GPU: a network + optimiser saved as a TensorFlow checkpoint (reference variable names) restores into a fresh
replica that then trains bit-identically; the file passes the format checks of tests/test_checkpoint_cpu.py."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
F = np.float32


def test_save_restore_continue(dev, tmp_path):
    from pfpn_amd import checkpoint as ck
    from pfpn_amd.models import SyncReplicasApplier
    from test_gpu_dppo_step import make_net, make_batch
    rng = np.random.default_rng(0)

    def step(net, opt, batch):
        return net.train(None, opt, [], batch["state"], batch["action"], batch["value"], batch["log_prob"],
                         batch["advantage"])[0]

    net = make_net(dev, seed=1, resample_interval=1000)
    opt = SyncReplicasApplier(net, [{"names": None, "lr": 1e-4}], norm_clip=1.0, world_size=1)
    batches = []
    for _ in range(4):
        b = make_batch(rng, 64)
        b["log_prob"] = np.full(64, -40.0, F)
        batches.append(b)
    for _ in range(20):                                   # some activation statistics too
        net.run(None, rng.normal(0, 1, 197).astype(F))
    step(net, opt, batches[0]); step(net, opt, batches[1])
    prefix = str(tmp_path / "model.ckpt-2")
    entries = ck.save(prefix, net, opt, n_workers=2, total_samples=128)
    # reference inventory: 14 trainables x (value, Adam, Adam_1) + powers + statistics + counters + local copies
    assert entries["global_net/actor/fc_policy/weight"].shape == (512, 1260)
    assert entries["global_net/actor/fc1/weight/Adam_1"].shape == (197, 1024)
    assert entries["step/global_step"].dtype == 9 and "worker_1/critic/fc3/bias" in entries
    rd = ck.TFCheckpointReader(prefix)                    # verifies every block and tensor checksum
    assert int(rd.get("step/global_step")) == 2 and int(rd.get("global_net/samples/samples")) == 128
    np.testing.assert_array_equal(rd.get("global_net/actor/samples"), net.loc.cpu().numpy())

    net2 = make_net(dev, seed=99, resample_interval=1000)            # different initial weights
    opt2 = SyncReplicasApplier(net2, [{"names": None, "lr": 1e-4}], norm_clip=1.0, world_size=1)
    assert not torch.equal(net2.store.flat, net.store.flat)
    ck.restore(prefix, net2, opt2)
    assert net2.global_step == 2 and net2.train_flag == net.train_flag
    for a, b in ((net.store.flat, net2.store.flat), (net.store.m, net2.store.m), (net.store.v, net2.store.v),
                 (net.state_normalizer.mean, net2.state_normalizer.mean), (net.max_active, net2.max_active),
                 (net.sum_active, net2.sum_active), (opt.groups[0]["beta_pow"], opt2.groups[0]["beta_pow"])):
        assert torch.equal(a, b)
    for b in batches[2:]:
        l1, l2 = step(net, opt, b), step(net2, opt2, b)
        assert l1 == l2
    assert torch.equal(net.store.flat, net2.store.flat) and torch.equal(net.store.v, net2.store.v)

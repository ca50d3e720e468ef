"""CPU, world_size 2 over gloo: the data-parallel exchange of SyncReplicasApplier (one all-reduce of
the bucket [per-rank clipped grads | local statistics]) reproduces the reference's SyncReplicasOptimizer
semantics as restated by the oracle: MEAN of per-worker-clipped gradients, MEAN of the 4 pushed statistic
tensors (including the reference's quirk that max_active is averaged, not max-reduced)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

F = np.float32


class CpuNet(object):
    """a network-shaped holder of a CPU ParamStore + the stat slots of the PFPN network."""
    def __init__(self, sizes, S, AN):
        from pfpn_amd.networks.ops import ParamStore
        self.store = ParamStore(torch.device("cpu"))
        for n, shp in sizes:
            self.store.declare(n, shp)
        self.store.allocate(extra=2 * S + 2 * AN, with_optimizer=False)
        self.S, self.AN = S, AN
        self.mean, self.std = torch.zeros(S), torch.ones(S)
        self.max_active, self.sum_active = torch.zeros(AN), torch.zeros(AN)
        self.trainable, self.normalize_state = True, True

    def stat_slots(self):
        S, AN = self.S, self.AN
        return [(self.mean, 0, S, True), (self.std, S, S, True),
                (self.max_active, 2 * S, AN, False), (self.sum_active, 2 * S + AN, AN, False)]

    # the real methods, borrowed from the product class
    from pfpn_amd.networks.actor_critic.actor_critic import ActorCriticNetwork as _A
    pack_local_stats = _A.pack_local_stats
    unpack_synced_stats = _A.unpack_synced_stats


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank_main(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from oracle import pfpn_oracle as orc
        from pfpn_amd.models.sync_model import SyncReplicasApplier
        S, A, N, B = 13, 3, 35, 24
        rng = np.random.default_rng(0)                       # same on both ranks: the global state + all shards
        st = orc.DPPOState(rng, S, A, N, hidden=(16, 8))
        shards = []
        for r in range(world):
            b = {"state": rng.normal(0, 1, (B, S)).astype(F), "action": rng.uniform(-1, 1, (B, A)).astype(F),
                 "value": rng.normal(0, 1, B).astype(F), "log_prob": rng.normal(-3, 0.3, B).astype(F),
                 "advantage": rng.normal(0, 1, B).astype(F),
                 "max_active": rng.random((A, N)).astype(F), "sum_active": (rng.random((A, N)) * 9).astype(F)}
            shards.append(b)
        mine = shards[rank]
        # what THIS rank computes locally (checker arithmetic stands in for the GPU kernels)
        (_, _, _), G, _ = orc.dppo_grads(st.P, st.norm_mean, st.norm_std, mine, A, N)
        clipped, _ = orc.clip_by_global_norm([G[n] for n in orc.PARAM_ORDER], 1.0)
        lm, ls = orc.state_norm_update(mine["state"], st.norm_mean, st.norm_std, 0)
        net = CpuNet([(n, st.P[n].shape) for n in orc.PARAM_ORDER], S, A * N)
        for n, g in zip(orc.PARAM_ORDER, clipped):
            net.store.g(n).copy_(torch.from_numpy(g.reshape(net.store.g(n).shape)))
        net.store.tail[:S].copy_(torch.from_numpy(lm)); net.store.tail[S:2 * S].copy_(torch.from_numpy(ls))
        net.max_active.copy_(torch.from_numpy(mine["max_active"].reshape(-1)))
        net.sum_active.copy_(torch.from_numpy(mine["sum_active"].reshape(-1)))
        ap = SyncReplicasApplier.__new__(SyncReplicasApplier)
        ap.pg, ap.world = None, world
        ap.reduce(net)                                              # the product's exchange step
        net.unpack_synced_stats(net.store.tail, 1.0 / world)
        gmean = {n: (net.store.g(n) / world).numpy().copy() for n in orc.PARAM_ORDER}
        # reference semantics from the oracle (mean of per-worker clipped grads and stats)
        ref_clipped = []
        ref_stats = []
        for b in shards:
            (_, _, _), Gb, _ = orc.dppo_grads(st.P, st.norm_mean, st.norm_std, b, A, N)
            c, _ = orc.clip_by_global_norm([Gb[n] for n in orc.PARAM_ORDER], 1.0)
            ref_clipped.append(c)
            m_, s_ = orc.state_norm_update(b["state"], st.norm_mean, st.norm_std, 0)
            ref_stats.append((m_, s_, b["max_active"], b["sum_active"]))
        ok = True
        for i, n in enumerate(orc.PARAM_ORDER):
            ref = sum(c[i] for c in ref_clipped) / F(world)
            ok &= np.allclose(gmean[n].reshape(ref.shape), ref, rtol=1e-6, atol=1e-8)
        for got, j in ((net.mean, 0), (net.std, 1), (net.max_active, 2), (net.sum_active, 3)):
            ref = sum(s[j] for s in ref_stats) / F(world)
            ok &= np.allclose(got.numpy().reshape(ref.shape), ref, rtol=1e-6, atol=1e-7)
        # max_active is AVERAGED, not max-reduced (reference quirk, sync_model.py:85-87)
        mx = np.maximum(shards[0]["max_active"], shards[1]["max_active"]).reshape(-1)
        ok &= not np.allclose(net.max_active.numpy(), mx)
        # the stepper's fall-back agreement (pfpn_amd/graphs.py): ONE replica whose capture failed takes every replica
        # to the two-graph form; nobody failing leaves everybody where they are
        from pfpn_amd.graphs import any_replica
        ok &= any_replica(rank == world - 1) is True and any_replica(False) is False
        # a replica whose capture the runtime invalidated cannot vote through a device collective: it publishes that it is lost
        # (host side, the rendezvous store) and raises; the others get ReplicaLost from their agreement call at once instead
        # of waiting in an all-reduce for a rank that will never arrive
        from pfpn_amd.graphs import ReplicaLost
        if rank == world - 1:
            ok &= any_replica(True, fatal=True) is True
        else:
            try:
                any_replica(False, timeout_s=60)
                ok &= False
            except ReplicaLost as e:
                ok &= str(world - 1) in str(e)
        dist.barrier()
        q.put((rank, bool(ok), float(net.store.bucket.sum())))
    except Exception as e:                 # report instead of leaving the parent waiting for its timeout
        import traceback
        traceback.print_exc()
        q.put((rank, False, repr(e)))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_bucket_allreduce_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_main, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
    assert all(r[1] for r in res), res
    assert res[0][2] == res[1][2], "ranks hold different buckets after the all-reduce"


def _sharded_rank_main(rank, world, port, q):
    """the SHARDED exchange's transport on CPU tensors: all-to-all of unscaled bucket slices, all-gather of the clip
    factors, (the owner's combine, checker arithmetic standing in for pfpn_shard_adam), in-place all-gather."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from oracle import pfpn_oracle as orc
        from pfpn_amd.models.sync_model import SyncReplicasApplier
        S, A, N, B = 13, 3, 35, 24
        rng = np.random.default_rng(0)
        st = orc.DPPOState(rng, S, A, N, hidden=(16, 8))
        shards = []
        for r in range(world):
            shards.append({"state": rng.normal(0, 1, (B, S)).astype(F), "action": rng.uniform(-1, 1, (B, A)).astype(F),
                           "value": rng.normal(0, 1, B).astype(F), "log_prob": rng.normal(-3, 0.3, B).astype(F),
                           "advantage": (rng.normal(0, 1, B) * (3.0 if r else 0.05)).astype(F),    # one replica clips
                           "max_active": rng.random((A, N)).astype(F), "sum_active": (rng.random((A, N)) * 9).astype(F)})
        mine = shards[rank]
        (_, _, _), G, _ = orc.dppo_grads(st.P, st.norm_mean, st.norm_std, mine, A, N)
        _, gnorm = orc.clip_by_global_norm([G[n] for n in orc.PARAM_ORDER], 1.0)
        s_mine = F(1.0) * min(F(1.0) / F(gnorm), F(1.0))
        lm, ls = orc.state_norm_update(mine["state"], st.norm_mean, st.norm_std, 0)
        net = CpuNet([(n, st.P[n].shape) for n in orc.PARAM_ORDER], S, A * N)
        stn = net.store
        assert stn.npad % (4 * world) == 0 and stn.bucketx.numel() == stn.npad == stn.flatx.numel()
        for n in orc.PARAM_ORDER:                                     # UNSCALED gradients travel
            stn.g(n).copy_(torch.from_numpy(G[n].reshape(stn.g(n).shape)))
        stn.tail[:S].copy_(torch.from_numpy(lm)); stn.tail[S:2 * S].copy_(torch.from_numpy(ls))
        net.max_active.copy_(torch.from_numpy(mine["max_active"].reshape(-1)))
        net.sum_active.copy_(torch.from_numpy(mine["sum_active"].reshape(-1)))
        net.pack_local_stats(stn.tail)
        ap = SyncReplicasApplier.__new__(SyncReplicasApplier)
        ap.pg, ap.world, ap.rank, ap.sharded = None, world, rank, True
        L = ap.shard_len = stn.npad // world
        ap.recv, ap.scales = torch.zeros(stn.npad), torch.zeros(world)
        ap.norm_scale = torch.tensor([float(gnorm), float(s_mine)])
        ap.reduce(net)                                                # all-to-all + factor all-gather
        recv = ap.recv.view(world, L).numpy()
        sc = ap.scales.numpy()
        lo = rank * L
        own = np.zeros(L, F)
        for i in range(L):                                            # the owner's combine, replica order
            idx = lo + i
            if idx < stn.size:
                own[i] = -(sum(F(sc[w]) * recv[w, i] for w in range(world)) / F(world))
            elif idx < stn.size + stn.extra:
                own[i] = sum(recv[w, i] for w in range(world)) / F(world)
        stn.flatx[lo:lo + L].copy_(torch.from_numpy(own))
        ap.finish_reduce(net)                                         # in-place all-gather
        ok = True
        ref_scales = []
        for i, n in enumerate(orc.PARAM_ORDER):
            ref = 0
            for b in shards:
                (_, _, _), Gb, _ = orc.dppo_grads(st.P, st.norm_mean, st.norm_std, b, A, N)
                c, nb_ = orc.clip_by_global_norm([Gb[m] for m in orc.PARAM_ORDER], 1.0)
                ref = ref + c[i]
                if i == 0:
                    ref_scales.append(min(1.0 / nb_, 1.0))
            ok &= np.allclose(-stn[n].numpy().reshape(ref.shape), ref / F(world), rtol=2e-6, atol=1e-8)
        # every replica clips by its OWN factor (they differ): the owner's weighted sum is what makes this exact
        ok &= np.allclose(sc, ref_scales, rtol=1e-6) and max(ref_scales) < 1.0 and len(set(ref_scales)) == len(ref_scales)
        tail = stn.flat_tail.numpy()
        for j, (off, n) in enumerate(((0, S), (S, S), (2 * S, A * N), (2 * S + A * N, A * N))):
            vals = []
            for b in shards:
                m_, s_ = orc.state_norm_update(b["state"], st.norm_mean, st.norm_std, 0)
                vals.append((m_, s_, b["max_active"].reshape(-1), b["sum_active"].reshape(-1))[j])
            ok &= np.allclose(tail[off:off + n], sum(vals) / F(world), rtol=1e-6, atol=1e-7)
        q.put((rank, bool(ok), float(stn.flatx.double().sum())))
    except Exception as e:
        import traceback
        traceback.print_exc()
        q.put((rank, False, repr(e)))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 4, 8])
def test_sharded_exchange_transport_gloo(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sharded_rank_main, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
    assert all(r[1] for r in res), res
    assert all(r[2] == res[0][2] for r in res), "ranks hold different parameters after the all-gather"

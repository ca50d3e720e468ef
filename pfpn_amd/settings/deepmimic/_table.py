"""All DeepMimic settings of the PFPN path as ONE declarative table.

The reference keeps one small module per setting (settings/deepmimic/deepmimic_*.py) that main.py imports by
name and reads as attributes (network_opts, worker_opts, workers, model, worker, network, max_iterations, ...).
Here the values live in the table below -- each entry cites the reference file it restates -- and the modules of
the same names are three-line stubs that ask `publish()` to fill their namespace, so `--setting
settings.deepmimic.deepmimic_dppo_pfpn` keeps working.  Command-line overrides (`--particles`, `--max_samples`,
`--soft`) are read from sys.argv exactly where the reference reads them.
"""
import os
import sys


def _flag(name, default, cast=int):
    return cast(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else default


def _ceil(x, y):
    return -(-x // y)


# deepmimic_base.py:3-24 -- the TF callables of the reference become names / None
NETWORK_BASE = dict(common_net_shape=[], critic_net_shape=[1024, 512], actor_net_shape=[1024, 512],
                    weight_initializer=None,      # truncated_normal(0.0, 0.01)
                    activator="relu6",            # tf.nn.relu6
                    value_loss_coef=0.5, gamma=0.95, normalize_state=True, clip_state=5., normalize_value=False,
                    clip_value=None, normalize_advantage=False, clip_advantage=None)
WORKER_BASE = dict(lr_critic=1e-4, lr_actor=5e-6)

# family -> (network overrides, workers, worker overrides, model, worker class); reference file in the comment
FAMILIES = {
    # deepmimic_dppo_base.py:4-19
    "dppo": (dict(normalize_advantage=True), 8,
             dict(norm_clip=1., opt_epochs=1, batch_size=32, unroll_length=512), "SyncModel", "DPPOWorker"),
    # deepmimic_sac_base.py:4-17
    "sac": (dict(), 1,
            dict(norm_clip=1, opt_epochs=None, batch_size=256, unroll_length=1, buffer_capacity=int(1e6),
                 observations=0), "SyncModel", "SyncSACWorker"),
    # deepmimic_a3c_base.py:4-21 -- the reference's AsyncModel / AsyncA2CWorker (16 asynchronous workers over a
    # parameter server) become synchronous replicas: the same A2C learner step, run as batched A2C
    "a3c": (dict(entropy_beta=0.00025), 16,
            dict(norm_clip=0.5, opt_epochs=1, batch_size=32, unroll_length=32), "SyncModel", "SyncA2CWorker"),
    # deepmimic_impala_base.py:4-19 -- LearnerModel / IMPALAWorker (actor / learner queues) become synchronous
    # replicas: every rank collects batch_size segments of unroll_length steps and takes the V-trace step
    "impala": (dict(entropy_beta=0.00025), 32,
               dict(norm_clip=.5, opt_epochs=1, batch_size=4, unroll_length=64), "SyncModel", "SyncVTraceWorker"),
}


def _max_iterations(family, workers, w):
    samples = _flag("--max_samples", 15000000)
    if family == "dppo":          # deepmimic_dppo_base.py:15-17
        return _ceil(samples, w["unroll_length"] * workers) * w["unroll_length"] / w["batch_size"] * w["opt_epochs"]
    if family == "sac":           # deepmimic_sac_base.py:15
        return samples
    if family == "a3c":           # deepmimic_a3c_base.py:16-18
        return _ceil(samples, w["batch_size"])
    return 10000000               # deepmimic_impala_base.py:16


def _pfpn(family, net, w, workers):
    """the *_pfpn.py modules: particle count from the command line, resampling on, actor learning rate 1e-4."""
    net["particles"] = _flag("--particles", 35)
    net["resample"] = -1
    w["lr_actor"] = 1e-4
    if family == "dppo":          # deepmimic_dppo_pfpn.py:9-17
        rolls = round(600 * 20 / w["unroll_length"])                      # episode length x resample interval
        per_roll = w["unroll_length"] / w["batch_size"] * w["opt_epochs"]
        net["resample_interval"] = int(rolls * w["unroll_length"]) if "--soft" in sys.argv else int(rolls * per_roll)
        return "ParticleFilteringClipPPONetwork"
    if family == "sac":           # deepmimic_sac_pfpn.py:4-10
        net["resample_interval"] = 12000
        return "ParticleFilteringSACNetwork"
    if family == "a3c":           # deepmimic_a3c_pfpn.py:4-9
        net["resample_interval"] = int(12000 * workers / w["batch_size"])
        return "ParticleFilteringA2CNetwork"
    net["resample_interval"] = int(12000 * workers / w["batch_size"] / w["unroll_length"])   # deepmimic_impala_pfpn.py:4-9
    return "ParticleFilteringVTraceNetwork"


def publish(namespace, family=None, pfpn=False):
    """fill a settings module's namespace: publish(globals(), "dppo", pfpn=True)."""
    net, w = dict(NETWORK_BASE), dict(WORKER_BASE)
    net["common_net_shape"], net["critic_net_shape"], net["actor_net_shape"] = [], [1024, 512], [1024, 512]
    out = {"network_opts": net, "worker_opts": w}
    if family is not None:
        net_over, workers, w_over, model, worker = FAMILIES[family]
        # the replica count that the schedules below assume is the one torchrun actually started
        world = int(os.environ.get("WORLD_SIZE", "0"))
        if world and world != workers:
            sys.stderr.write("[settings] %s: %d replicas running (setting's default %d): max_iterations / "
                             "resample_interval follow the running count\n" % (family, world, workers))
            workers = world
        net.update(net_over)
        w.update(w_over)
        out.update(workers=workers, model=model, worker=worker, max_iterations=_max_iterations(family, workers, w))
        if family == "impala":
            out["max_samples"] = _flag("--max_samples", 15000000)
        if pfpn:
            out["network"] = _pfpn(family, net, w, workers)
    namespace.update(out)

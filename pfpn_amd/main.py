"""Launcher with the reference's command line (reference main.py:14-135, benchmark.sh:18):

  python -m pfpn_amd.main --env DeepMimicWalk-v0 --seed 34114 \
      --setting settings.deepmimic.deepmimic_dppo_pfpn --particles 35 [--train]

One process per GPU: start N ranks with torchrun (RANK/WORLD_SIZE/LOCAL_RANK) for the reference's
`workers = N` synchronous data-parallel run; without torchrun a single worker runs on cuda:0.
`--port`, `--device`, `--save_checkpoint_interval` are accepted for command-line compatibility.
"""
import argparse
import importlib
import sys


def parse(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument("--setting", type=str, required=True)
    parser.add_argument("--env", type=str, required=True)
    parser.add_argument("--seed", type=int, default=1)
    parser.add_argument("--port", type=int, default=2425)
    parser.add_argument("--train", action="store_true", default=False)
    parser.add_argument("--debug", action="store_true", default=False)
    parser.add_argument("--suffix", type=str, default="")
    parser.add_argument("--device", type=str, default="GPU:0")
    parser.add_argument("--save_checkpoint_interval", type=int, default=1000)
    parser.add_argument("--particles", type=int, default=None)
    parser.add_argument("--max_samples", type=int, default=None)
    parser.add_argument("--max_iterations", type=int, default=None)
    settings, _ = parser.parse_known_args(argv)
    name = settings.setting
    if not name.startswith("pfpn_amd."):
        name = "pfpn_amd." + name
    mod = importlib.import_module(name)
    for k in ("network_opts", "worker_opts", "workers", "worker", "network", "model", "max_iterations"):
        if hasattr(mod, k) and getattr(settings, k, None) is None:
            setattr(settings, k, getattr(mod, k))
    settings.log_dir = settings.setting.split(".")[-1]
    if settings.suffix:
        settings.suffix = "_" + settings.suffix
    if settings.particles is not None:                       # main.py:48-50
        settings.network_opts["particles"] = settings.particles
        settings.suffix = "_particle" + str(settings.particles) + settings.suffix
    settings.checkpoint_dir = "ckpt_{}/{}{}/{}".format(settings.env, settings.log_dir, settings.suffix, settings.seed)
    settings.log_dir = "log_{}/{}{}/{}".format(settings.env, settings.log_dir, settings.suffix, settings.seed)
    return settings


def build(settings):
    import os
    from . import models, networks
    from .envs import make_env
    if os.environ.get("PFPN_TUNED_GEMMS", "1") != "0":       # shipped hipBLASLt / rocBLAS selection for the MLP GEMMs
        from .tuning import enable_tuned_gemms
        enable_tuned_gemms(tune_missing=True)
    worker = getattr(models, settings.worker)                # resolved by string name (main.py:56-58)
    network = getattr(networks, settings.network)
    model = getattr(models, settings.model)

    def env_wrapper(name, render):
        return make_env(settings.env, render, settings.seed)

    def network_wrapper(env, trainable):
        kwargs = {"trainable": trainable, "state_shape": env.observation_space.shape,
                  "action_shape": env.action_space.shape, "seed": settings.seed}
        if hasattr(env.action_space, "low"):
            kwargs["action_lower_bound"] = env.action_space.low
            kwargs["action_upper_bound"] = env.action_space.high
        return network(**kwargs, **settings.network_opts)

    def worker_wrapper(**kwargs):
        return worker(**kwargs, **settings.worker_opts)

    return model(worker_wrapper, env_wrapper, network_wrapper, checkpoint_dir=settings.checkpoint_dir,
                 save_checkpoint_interval=settings.save_checkpoint_interval,
                 max_iterations=settings.max_iterations, device=settings.device, seed=settings.seed,
                 log_dir=settings.log_dir, max_samples=settings.max_samples)


def main(argv=None):
    settings = parse(argv)
    model = build(settings)
    try:
        model.start({"worker": []} if settings.train else None)
    finally:
        from .models.distributed_model import shutdown_process_group
        shutdown_process_group()


if __name__ == "__main__":
    main(sys.argv[1:])

"""reference settings/deepmimic/deepmimic_dppo_pfpn.py:4-21 (resample interval arithmetic :9-17)."""
import sys

from .deepmimic_dppo_base import *  # noqa: F401,F403
from .deepmimic_dppo_base import network_opts, worker_opts

network_opts["particles"] = int(sys.argv[sys.argv.index("--particles") + 1]) if "--particles" in sys.argv else 35
network_opts["resample"] = -1

resample_interval = 20
episode_length = 600
iterations_per_roll = worker_opts["unroll_length"] / worker_opts["batch_size"] * worker_opts["opt_epochs"]
unroll_length = worker_opts["unroll_length"]
rolls = round(episode_length * resample_interval / unroll_length)
if "--soft" in sys.argv:
    network_opts["resample_interval"] = int(rolls * worker_opts["unroll_length"])
else:
    network_opts["resample_interval"] = int(rolls * iterations_per_roll)

worker_opts["lr_actor"] = 1e-4

network = "ParticleFilteringClipPPONetwork"

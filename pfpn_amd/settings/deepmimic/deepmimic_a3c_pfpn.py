"""reference settings/deepmimic/deepmimic_a3c_pfpn.py:4-9."""
import sys

from .deepmimic_a3c_base import *  # noqa: F401,F403
from .deepmimic_a3c_base import network_opts, worker_opts, workers

network_opts["particles"] = int(sys.argv[sys.argv.index("--particles") + 1]) if "--particles" in sys.argv else 35
network_opts["resample"] = -1
network_opts["resample_interval"] = int(12000 * workers / worker_opts["batch_size"])
worker_opts["lr_actor"] = 1e-4

network = "ParticleFilteringA2CNetwork"

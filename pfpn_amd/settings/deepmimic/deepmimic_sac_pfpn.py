"""reference settings/deepmimic/deepmimic_sac_pfpn.py:4-10."""
import sys

from .deepmimic_sac_base import *  # noqa: F401,F403
from .deepmimic_sac_base import network_opts, worker_opts

network_opts["particles"] = int(sys.argv[sys.argv.index("--particles") + 1]) if "--particles" in sys.argv else 35
network_opts["resample"] = -1
network_opts["resample_interval"] = 12000

worker_opts["lr_actor"] = 1e-4

network = "ParticleFilteringSACNetwork"

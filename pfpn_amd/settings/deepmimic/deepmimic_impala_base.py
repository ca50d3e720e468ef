"""reference settings/deepmimic/deepmimic_impala_base.py:4-19.  The reference's actor/learner queues
(`LearnerModel` / `IMPALAWorker`) become synchronous replicas: every rank collects `batch_size` segments of
`unroll_length` steps and takes the V-trace learner step on them."""
import copy
import sys

from . import deepmimic_base as _base

network_opts = copy.deepcopy(_base.network_opts)
worker_opts = copy.deepcopy(_base.worker_opts)

network_opts["entropy_beta"] = 0.00025

workers = 32
worker_opts.update({
    "norm_clip": .5,
    "opt_epochs": 1,
    "batch_size": 4,
    "unroll_length": 64,
})

max_samples = int(sys.argv[sys.argv.index("--max_samples") + 1]) if "--max_samples" in sys.argv else 15000000
max_iterations = 10000000

model = "SyncModel"
worker = "SyncVTraceWorker"

"""reference settings/deepmimic/deepmimic_a3c_base.py:4-21.  The reference trains this setting with 16
asynchronous workers over a parameter server (`AsyncModel` / `AsyncA2CWorker`); here the replicas are
synchronous (one RCCL all-reduce per step), i.e. the same A2C learner step run as batched A2C."""
import copy
import sys

from . import deepmimic_base as _base

network_opts = copy.deepcopy(_base.network_opts)
worker_opts = copy.deepcopy(_base.worker_opts)

network_opts["entropy_beta"] = 0.00025

workers = 16
worker_opts.update({
    "norm_clip": 0.5,
    "opt_epochs": 1,
    "batch_size": 32,
    "unroll_length": 32,
})

_ceil = lambda x, y: -(-x // y)
_max_samples = int(sys.argv[sys.argv.index("--max_samples") + 1]) if "--max_samples" in sys.argv else 15000000
max_iterations = _ceil(_max_samples, worker_opts["batch_size"])

model = "SyncModel"
worker = "SyncA2CWorker"

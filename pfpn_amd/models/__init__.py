"""Models and workers of the PFPN path, under the names the settings resolve (reference models/__init__.py):
`SyncModel` (one process per GPU, RCCL all-reduce) and the synchronous workers."""
from .sync_model import SyncModel, SyncReplicasApplier  # noqa: F401
from .workers import (BatchedA2CWorker, Buffer, DPPOWorker, IMPALAWorker, SyncA2CWorker, SyncDDPGWorker,  # noqa: F401
                      SyncPPOWorker, SyncSACWorker, SyncVTraceWorker, VTraceLearner)

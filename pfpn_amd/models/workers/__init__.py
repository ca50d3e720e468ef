"""Workers (class names as the settings name them)."""
from .a2c import BatchedA2CWorker, SyncA2CWorker
from .ddpg import Buffer, SyncDDPGWorker
from .ppo import DPPOWorker, SyncPPOWorker
from .sac import SyncSACWorker
from .vtrace import IMPALAWorker, SyncVTraceWorker, VTraceLearner

"""reference models/workers/sac.py:1-12: the SAC worker is the DDPG worker."""
from .ddpg import SyncDDPGWorker

__all__ = ["SyncSACWorker"]

SyncSACWorker = SyncDDPGWorker

from .sync_model import SyncModel, SyncReplicasApplier
from .workers import *

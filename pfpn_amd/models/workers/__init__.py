from .ppo import *
from .ddpg import *
from .sac import *

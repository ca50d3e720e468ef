from .ppo import *
from .ddpg import *
from .sac import *
from .a2c import *
from .vtrace import *

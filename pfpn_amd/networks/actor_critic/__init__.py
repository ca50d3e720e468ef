from .a2c import *
from .ppo import *
from .sac import *
from .vtrace import *

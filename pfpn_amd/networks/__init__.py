from .actor_critic import *

"""Actor-critic networks (reference networks/actor_critic/__init__.py exports the same class names)."""
from .a2c import ParticleFilteringA2CNetwork, particle_grid
from .ddpg import ParticleFilteringDDPGNetwork
from .ppo import ParticleFilteringClipPPONetwork
from .sac import ParticleFilteringSACNetwork
from .vtrace import ParticleFilteringVTraceNetwork, v_trace_network_wrapper

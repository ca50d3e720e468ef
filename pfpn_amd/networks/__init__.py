"""Policy networks of the PFPN path (names as main.py resolves them: reference networks/__init__.py)."""
from .actor_critic import (ParticleFilteringA2CNetwork, ParticleFilteringClipPPONetwork, ParticleFilteringDDPGNetwork,
                           ParticleFilteringSACNetwork,
                           ParticleFilteringVTraceNetwork, particle_grid, v_trace_network_wrapper)

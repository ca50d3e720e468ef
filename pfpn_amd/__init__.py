"""pfpn_amd -- MI355X-native (gfx950) implementation of the PFPN particle-policy hot path.

Kernels live in csrc/*.hip behind the C ABI of include/pfpn_hip.h (libpfpn_hip.so); the host side
mirrors the reference's Python policy/worker API (networks/, models/)."""
__version__ = "0.1.0"

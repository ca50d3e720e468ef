#!/bin/bash
# A/B builds of the one-launch PPO pass (tools/ab_build.sh; alternating rounds on one box):  tools/f2l_diag.sh "<flags>" "<flags>" ...
#   ""                     the shipped kernel
# DIAGNOSTIC knobs (results WRONG by design; only the times count):
#   -DPFPN_F2L_NOX=1       nobody waits for the chunk, nobody polls;   =2  polls issued, nobody waits
#   -DPFPN_F2L_NOSTORE=1   dlogits stored for one row in 64 (the read side + the arithmetic alone)
# PROBE_ARGS="--rows 2048 --dbias" etc. selects the shape (tools/f2l_probe.py).
ROUNDS=${ROUNDS:-2} bash tools/ab_build.sh gpurun_out/f2l_diag "python3 tools/f2l_probe.py ${PROBE_ARGS} 2>&1 | grep -E 'us |chunks|parity|polls|bitwise'" "$@"

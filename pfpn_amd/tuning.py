"""GEMM algorithm selection for the dense layers (PyTorch-ROCm fp32 GEMMs: hipBLASLt / rocBLAS through TunableOp).

`tuned/tunableop_gfx950.csv` is the selection `tools/tune_gemms.py` found on an MI355X for the GEMM shapes of the shipped
settings (4096-row learner batches, the 32..1024-row minibatches / replay batches).  Loading it makes every process --
every data-parallel rank included -- run the same kernels without spending its first steps tuning; shapes the file does
not know are tuned on first use (`tune_missing`).  The file is only honoured when TunableOp's validators (PyTorch, HIP,
hipBLASLt, rocBLAS versions, GCN arch) match the running stack; otherwise everything is tuned on first use.
"""
import os
import tempfile

import torch

# (PFPN_TUNED_FILE: another selection file, for A/B runs of a re-tuned selection against the shipped one)
TUNED_FILE = os.environ.get("PFPN_TUNED_FILE") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuned",
                                                                 "tunableop_gfx950.csv")


def enable_tuned_gemms(tune_missing=True, max_duration_ms=30, max_iterations=20):
    """-> True when the shipped selection was accepted.  New results go to a per-process scratch file, never into the
    package.  Safe to call more than once; a no-op (returns False) when this PyTorch has no TunableOp."""
    try:
        tunable = torch.cuda.tunable
        tunable.enable(True)
        tunable.set_filename(os.path.join(tempfile.gettempdir(), "pfpn_tunableop_%d.csv" % os.getpid()))
        tunable.set_max_tuning_duration(int(max_duration_ms))
        tunable.set_max_tuning_iterations(int(max_iterations))
        loaded = bool(os.path.exists(TUNED_FILE) and tunable.read_file(TUNED_FILE))
        tunable.tuning_enable(bool(tune_missing))
        return loaded
    except Exception:
        return False

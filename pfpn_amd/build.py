"""Build libpfpn_hip.so (gfx950) in-tree with hipcc.  No torch, no cmake: one shared library
with a plain C ABI (include/pfpn_hip.h)."""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libpfpn_hip.so")
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-ffp-contract=off",
         "-Wno-unused-result", "-I" + os.path.join(HERE, "..", "include")]
FLAGS += os.environ.get("PFPN_EXTRA_FLAGS", "").split()      # tuning experiments (-DTK_DEPTH=3 ...)
if os.environ.get("PFPN_BUILD_STAMPS"):       # diagnostic build for tools/stamp_probe.py (remove build/ first)
    FLAGS.append("-DPFPN_ENABLE_STAMPS")


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    mt = os.path.getmtime(target)
    return any(os.path.getmtime(s) > mt for s in sources)


def build(force=False, verbose=False):
    """Compile every csrc/*.hip to an object and link libpfpn_hip.so.  Returns the path."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    hdrs = sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [os.path.join(HERE, "..", "include", "pfpn_hip.h")]
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    objs, procs = [], []
    for s in srcs:
        o = os.path.join(objdir, os.path.basename(s)[:-4] + ".o")
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            cmd = [hipcc] + FLAGS + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd))
            procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode())
            raise RuntimeError("hipcc failed on " + s)
    if force or procs or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB]
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))

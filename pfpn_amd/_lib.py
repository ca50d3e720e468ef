"""ctypes binding of libpfpn_hip.so (the C ABI in include/pfpn_hip.h).

The product path has NO CPU fallback: if the shared library is missing or an entry point is
absent, importing/using the kernels raises.  Prototypes are parsed from include/pfpn_hip.h so the
header stays the single source of truth for the ABI.
"""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpfpn_hip.so")
HEADER_PATH = os.path.join(_HERE, "..", "include", "pfpn_hip.h")

_CT = {
    "const float*": ctypes.c_void_p, "float*": ctypes.c_void_p, "void*": ctypes.c_void_p,
    "const void*": ctypes.c_void_p, "const double*": ctypes.c_void_p, "double*": ctypes.c_void_p,
    "const int*": ctypes.c_void_p, "int*": ctypes.c_void_p, "const int32_t*": ctypes.c_void_p,
    "int32_t*": ctypes.c_void_p, "const int64_t*": ctypes.c_void_p, "int64_t*": ctypes.c_void_p,
    "const uint8_t*": ctypes.c_void_p, "uint8_t*": ctypes.c_void_p,
    "const uint64_t*": ctypes.c_void_p, "uint64_t*": ctypes.c_void_p, "uint32_t*": ctypes.c_void_p,
    "const long long*": ctypes.c_void_p, "long long*": ctypes.c_void_p, "const PfpnReduceJob*": ctypes.c_void_p,
    "const PfpnStatePrologue*": ctypes.c_void_p, "const PfpnSacLoss*": ctypes.c_void_p,
    "const PfpnEpilogueJob*": ctypes.c_void_p,
    "const void* const*": ctypes.c_void_p, "void* const*": ctypes.c_void_p,
    "int": ctypes.c_int, "float": ctypes.c_float, "double": ctypes.c_double,
    "long long": ctypes.c_longlong, "size_t": ctypes.c_size_t, "uint64_t": ctypes.c_uint64, "int64_t": ctypes.c_int64,
    "uint32_t": ctypes.c_uint32,
    "void": None,
}


def parse_header(path=HEADER_PATH):
    """-> {name: (restype, [argtypes], [argnames])} for every prototype in the header."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"//[^\n]*", "", src)
    protos = {}
    for m in re.finditer(r"\b(int|size_t|void|uint32_t)\s+(pfpn_\w+)\s*\(([^)]*)\)\s*;", src):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        argtypes, argnames = [], []
        if args and args != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                mm = re.match(r"(.*?)(\w+)$", a)
                ty = mm.group(1).strip().replace(" *", "*")
                argtypes.append(ty)
                argnames.append(mm.group(2))
        protos[name] = (ret, argtypes, argnames)
    return protos


class PfpnLibraryError(RuntimeError):
    pass


_lib = None


def load():
    """Load libpfpn_hip.so and attach prototypes.  Raises PfpnLibraryError when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PfpnLibraryError(
            "libpfpn_hip.so not found at %s -- build it with `python -m pfpn_amd.build` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback." % LIB_PATH)
    # PyTorch bundles its own libamdhip64 (same SONAME as /opt/rocm's).  The kernels share streams and
    # device pointers with torch, so they MUST bind to the runtime instance torch uses: import torch
    # first, then the dynamic loader resolves our NEEDED libamdhip64.so.7 to the already-loaded copy.
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, (ret, argtypes, _) in parse_header().items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            raise PfpnLibraryError("libpfpn_hip.so does not export %s (stale build?)" % name)
        fn.restype = _CT[ret]
        fn.argtypes = [_CT[t] for t in argtypes]
    _lib = lib
    return lib


_ERR = {1: "PFPN_ERR_ARG (null pointer / unsupported shape)", 2: "PFPN_ERR_LAUNCH (HIP launch failed)",
        3: "PFPN_ERR_WORKSPACE (workspace missing or too small)"}


def check(status, what):
    if status != 0:
        raise PfpnLibraryError("%s failed: %s" % (what, _ERR.get(status, "status %d" % status)))

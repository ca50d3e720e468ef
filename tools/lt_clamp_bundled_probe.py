"""Round 5, one cheap re-check: does the hipBLASLt that lives INSIDE the PyTorch process (torch/lib/
libhipblaslt.so, resolved -- not loaded a second time -- through ctypes) accept HIPBLASLT_EPILOGUE_CLAMP_BIAS_EXT with
ACT_ARG0/1 = 0, 6 (BiasAdd + ReLU6 inside the GEMM, reference networks/ops.py:105-117) for the layer-0 / layer-1 forward
shapes of the 4096-row step?  Prints every status; "heuristic: 0 algorithms" or a non-zero status = no."""
import ctypes
import os

import torch

lib_path = os.path.join(os.path.dirname(torch.__file__), "lib", "libhipblaslt.so")
torch.zeros(1, device="cuda")                                   # the process has its BLAS libraries mapped
lt = ctypes.CDLL(lib_path, mode=getattr(os, "RTLD_NOLOAD", 4) | ctypes.RTLD_GLOBAL)
ver = ctypes.c_int(0)
h = ctypes.c_void_p()
print("library", lib_path, "create", lt.hipblasLtCreate(ctypes.byref(h)))
try:
    print("version status", lt.hipblasLtGetVersion(h, ctypes.byref(ver)), "version", ver.value)
except AttributeError:
    print("no hipblasLtGetVersion")
HIPBLAS_COMPUTE_32F, HIP_R_32F, OP_N = 2, 0, 111
EPILOGUE, BIAS_POINTER, ACT0, ACT1 = 2, 3, 102, 103
CLAMP_BIAS = 131076
for (M, K, N) in ((4096, 197, 2048), (4096, 1024, 512)):
    desc = ctypes.c_void_p()
    print("shape", (M, K, N), "desc", lt.hipblasLtMatmulDescCreate(ctypes.byref(desc), HIPBLAS_COMPUTE_32F, HIP_R_32F))
    ep = ctypes.c_uint32(CLAMP_BIAS)
    print("  set epilogue CLAMP_BIAS_EXT ->", lt.hipblasLtMatmulDescSetAttribute(desc, EPILOGUE, ctypes.byref(ep), 4))
    bias = torch.zeros(N, device="cuda")
    bp = ctypes.c_void_p(bias.data_ptr())
    print("  set bias pointer ->", lt.hipblasLtMatmulDescSetAttribute(desc, BIAS_POINTER, ctypes.byref(bp), 8))
    lo, hi = ctypes.c_float(0.0), ctypes.c_float(6.0)
    print("  set ACT_ARG0_EXT ->", lt.hipblasLtMatmulDescSetAttribute(desc, ACT0, ctypes.byref(lo), 4))
    print("  set ACT_ARG1_EXT ->", lt.hipblasLtMatmulDescSetAttribute(desc, ACT1, ctypes.byref(hi), 4))
    la, lb, ld = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
    lt.hipblasLtMatrixLayoutCreate(ctypes.byref(la), HIP_R_32F, ctypes.c_uint64(N), ctypes.c_uint64(K), ctypes.c_int64(N))
    lt.hipblasLtMatrixLayoutCreate(ctypes.byref(lb), HIP_R_32F, ctypes.c_uint64(K), ctypes.c_uint64(M), ctypes.c_int64(K))
    lt.hipblasLtMatrixLayoutCreate(ctypes.byref(ld), HIP_R_32F, ctypes.c_uint64(N), ctypes.c_uint64(M), ctypes.c_int64(N))
    pref = ctypes.c_void_p()
    lt.hipblasLtMatmulPreferenceCreate(ctypes.byref(pref))
    res = (ctypes.c_byte * (32 * 1024))()
    n = ctypes.c_int(0)
    st = lt.hipblasLtMatmulAlgoGetHeuristic(h, desc, la, lb, ld, ld, pref, 8, res, ctypes.byref(n))
    print("  heuristic status", st, "algorithms", n.value)

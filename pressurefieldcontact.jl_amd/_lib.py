"""ctypes binding of csrc/libpfc_hip.so (the C ABI of include/pfc.h).

There is no CPU fallback: if the shared library is missing or no HIP device is usable, the calls raise.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_PATH = os.path.join(CSRC, "libpfc_hip.so")      # the product build; build() never writes anywhere else
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
HIPCC_FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC", "-shared"]

OK, ERR_NONFINITE, ERR_OVERFLOW, ERR_BAD_ARG, ERR_NOMEM, ERR_HIP, ERR_STATE, ERR_INVERTED_TET = range(8)
STATUS_NAMES = {0: "PFC_OK", 1: "PFC_ERR_NONFINITE", 2: "PFC_ERR_OVERFLOW", 3: "PFC_ERR_BAD_ARG", 4: "PFC_ERR_NOMEM",
                5: "PFC_ERR_HIP", 6: "PFC_ERR_STATE", 7: "PFC_ERR_INVERTED_TET"}
REGULARIZED, BRISTLE = 0, 1

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)

# every symbol include/pfc.h declares: (restype, argtypes)
SIGNATURES = {
    "pfc_version": (C.c_int, []),
    "pfc_build_info": (C.c_int, []),
    "pfc_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "pfc_create_multi": (C.c_int, [_ip, C.c_int, C.POINTER(C.c_void_p)]),
    "pfc_last_shards": (C.c_int, [C.c_void_p]),
    "pfc_destroy": (None, [C.c_void_p]),
    "pfc_last_error": (C.c_char_p, [C.c_void_p]),
    "pfc_add_mesh": (C.c_int, [C.c_void_p, C.c_int, _dp, C.c_int, _ip, C.c_int, _ip, _dp, C.c_double, C.c_int,
                               _dp, _dp, _dp, _ip, _ip]),
    "pfc_add_instruction": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int, _dp]),
    "pfc_finalize": (C.c_int, [C.c_void_p]),
    "pfc_eval": (C.c_int, [C.c_void_p, C.c_int, _ip, _dp, _dp, _dp, _dp, _dp, _ip]),
    "pfc_eval_device": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_void_p, C.c_void_p]),
    "pfc_check": (C.c_int, [C.c_void_p]),
    "pfc_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_longlong]),
    "pfc_get_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_longlong)]),
    "pfc_get_stage_ms": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    "pfc_debug_pairs": (C.c_int, [C.c_void_p, C.c_int, _ip, _ip, C.c_int]),
    "pfc_debug_tractions": (C.c_int, [C.c_void_p, C.c_int, _dp, C.c_int]),
    "pfc_debug_stiffness": (C.c_int, [C.c_void_p, C.c_int, _dp, _dp, _dp, _dp]),
    "pfc_last_parts": (C.c_int, [C.c_void_p]),
    "pfc_last_team": (C.c_int, [C.c_void_p]),
    "pfc_last_dual_reused": (C.c_int, [C.c_void_p]),
    "pfc_eval_dual": (C.c_int, [C.c_void_p, C.c_int, C.c_int, _ip, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _ip]),
    "pfc_eval_dual_bp": (C.c_int, [C.c_void_p, C.c_int, C.c_int, _ip, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _ip]),
    "pfc_eval_dual_device": (C.c_int, [C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 13),
    "pfc_eval_dual_device_bp": (C.c_int, [C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 14),
    "pfc_eval_dual_device_more": (C.c_int, [C.c_void_p, C.c_int] + [C.c_void_p] * 6),
    "pfc_build_tree": (C.c_int, [C.c_int, _dp, C.c_int, C.c_int, _ip, _dp, C.c_int, _dp, _dp, _dp, _ip, _ip]),
    "pfc_tree_last_error": (C.c_char_p, []),
    "pfc_scatter_generalized": (C.c_int, [C.c_void_p, C.c_int, _dp, _dp, _ip, _ip, _ip, C.c_int, C.c_int, C.c_int, _dp, _dp]),
    "pfc_scatter_generalized_device": (C.c_int, [C.c_void_p, C.c_int] + [C.c_void_p] * 5 + [C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                                  C.c_int, C.c_void_p]),
    "pfc_debug_stamps": (C.c_int, [C.c_void_p, C.POINTER(C.c_longlong)]),
    "pfc_selftest_math": (C.c_int, [C.c_void_p, C.c_int, _dp, _dp, _dp]),
}


class PFCError(RuntimeError):
    def __init__(self, status: int, message: str = ""):
        self.status = status
        super().__init__(f"{STATUS_NAMES.get(status, status)}: {message}")


def build(force: bool = False) -> str:
    """Compile csrc/pfc_hip.hip for gfx950 with hipcc (cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC, f) for f in ("pfc_hip.hip", "pfc_tree.cpp", "pfc_sort.hip", "pfc_kernels.h", "pfc_bp.h", "pfc_np.h", "pfc_br.h", "pfc_dual.h", "pfc_fused.h", "pfc_clip.h", "pfc_multi.h", "pfc_sort.h")]
    srcs.append(os.path.join(os.path.dirname(HERE), "include", "pfc.h"))
    srcs.append(os.path.abspath(__file__))      # the compiler flags live here
    stale = (not os.path.exists(LIB_PATH)) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs)
    if force or stale:
        cmd = [HIPCC] + HIPCC_FLAGS + ["-o", LIB_PATH, srcs[0], srcs[1], srcs[2]]
        subprocess.run(cmd, check=True, cwd=CSRC)
    return LIB_PATH


_lib = None


def lib():
    """Load libpfc_hip.so; raises (never falls back) if it is absent."""
    global _lib
    if _lib is None:
        # Diagnostic variants (scripts/elimination.sh, scripts/build_stamps.sh) are built to build/variants/*.so (outside the package) and selected with
        # PFC_LIB=<path>; the product library is never overwritten by them.
        path = os.environ.get("PFC_LIB") or LIB_PATH
        if not os.path.exists(path):
            raise ImportError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(hipcc --offload-arch=gfx950); there is no CPU fallback for the contact hot path")
        L = C.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)       # AttributeError if the library does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        info = L.pfc_build_info()
        if info != 0 and os.environ.get("PFC_ALLOW_DIAGNOSTIC") != "1":
            raise ImportError(f"{path} is a diagnostic build (pfc_build_info() = {info:#x}: stamps / elimination variant, "
                              "results may be wrong); set PFC_ALLOW_DIAGNOSTIC=1 to load it on purpose")
        # The two calls a simulation makes thousands of times per second, bound a second time with plain address
        # arguments: building a typed ctypes pointer per array costs ~2.5 us, its integer address ~1 us (seven arrays
        # per pfc_eval -- a third of what a small scene's evaluation takes on the device).
        L.pfc_eval_addr = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, *([C.c_void_p] * 7))(("pfc_eval", L))
        L.pfc_eval_dual_addr = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int, *([C.c_void_p] * 12))(("pfc_eval_dual", L))
        L.pfc_eval_dual_bp_addr = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int, *([C.c_void_p] * 13))(("pfc_eval_dual_bp", L))
        L.pfc_loaded_path = path
        _lib = L
    return _lib

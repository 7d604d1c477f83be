"""ctypes binding of libjmid_hip.so (include/jmid_hip.h).  Fails loudly when the library is missing:
there is no Python/CPU fallback for the compute path."""
from __future__ import annotations

import ctypes as C
import os

from .build import library_path

c_float_p = C.POINTER(C.c_float)
Handle = C.c_void_p

NET_IMID, NET_JMID = 0, 1
MEM_HOST, MEM_DEVICE = 0, 1
PREC_F32, PREC_F16X3, PREC_F16, PREC_F16X2, PREC_F16MX = 0, 1, 2, 3, 4
PRECISIONS = {"f32": PREC_F32, "f16x3": PREC_F16X3, "f16": PREC_F16, "f16x2": PREC_F16X2, "f16mx": PREC_F16MX}

ERR_NAMES = {-1: "JMID_EINVAL", -2: "JMID_ENOWEIGHT", -3: "JMID_EHIP", -4: "JMID_ENOMEM", -5: "JMID_ERANGE", -6: "JMID_ETIMEOUT"}

# name -> (restype, argtypes): every symbol declared in include/jmid_hip.h
SIGNATURES = {
    "jmid_version": (C.c_char_p, []),
    "jmid_device_count": (C.c_int, []),
    "jmid_create": (C.c_int, [C.POINTER(Handle), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "jmid_destroy": (C.c_int, [Handle]),
    "jmid_last_error": (C.c_char_p, [Handle]),
    "jmid_load_weight": (C.c_int, [Handle, C.c_char_p, C.c_void_p, C.c_size_t]),
    "jmid_finalize_weights": (C.c_int, [Handle]),
    "jmid_set_ddim_table": (C.c_int, [Handle, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "jmid_set_ddpm_table": (C.c_int, [Handle, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "jmid_denoise_ddpm": (C.c_int, [Handle, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_int]),
    "jmid_encode": (C.c_int, [Handle, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "jmid_denoise": (C.c_int, [Handle, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                               C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_int]),
    "jmid_net_eval": (C.c_int, [Handle, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                C.c_int, C.c_void_p, C.c_int]),
    "jmid_episode_metrics": (C.c_int, [Handle, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_int]),
    "jmid_topk": (C.c_int, [Handle, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                            C.c_void_p, C.c_int]),
    "jmid_predict": (C.c_int, [Handle, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                               C.c_void_p, C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "jmid_set_chunk_episodes": (C.c_int, [Handle, C.c_int]),
    "jmid_set_tuning": (C.c_int, [Handle, C.c_char_p, C.c_int]),
    "jmid_set_caller_stream": (C.c_int, [Handle, C.c_void_p]),
    "jmid_graph_replays": (C.c_int64, [Handle]),
    "jmid_erange_count": (C.c_int64, [Handle]),
    "jmid_timeout_count": (C.c_int64, [Handle]),
    "jmid_profile_enable": (C.c_int, [Handle, C.c_uint32]),
    "jmid_profile_reset": (C.c_int, [Handle]),
    "jmid_profile_get": (C.c_int, [Handle, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_double)]),
    "jmid_kernel_class_count": (C.c_int, []),
    "jmid_kernel_class_name": (C.c_char_p, [C.c_int]),
    "jmid_synchronize": (C.c_int, [Handle]),
}

# exported by the diagnostics flavour only (-DJMID_DIAGNOSTICS, csrc/libjmid_hip_diag.so; include/jmid_hip.h)
DIAG_SIGNATURES = {
    "jmid_dbg_gemm": (C.c_int, [Handle, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                C.c_int, C.c_void_p]),
    "jmid_dbg_attention": (C.c_int, [Handle, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "jmid_dbg_gemm_ln_mx": (C.c_int, [Handle, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_int]),
    "jmid_dbg_add_layernorm": (C.c_int, [Handle, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "jmid_dbg_plan_chunks": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int]),
    "jmid_dbg_plan_chunks_mode": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int]),
}


class JmidError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"{ERR_NAMES.get(code, code)}: {message}")
        self.code = code


_LIBS = {}


def load_library(path: str = None) -> C.CDLL:
    """Load (once per path) the in-tree HIP library - ``path`` = None: the product's (``build.library_path()``: csrc/libjmid_hip.so
    unless JMID_LIB names another build).  Raises if it has not been built.  (Tests load both flavours side by side.)"""
    key = os.path.abspath(path or library_path())
    if key in _LIBS:
        return _LIBS[key]
    # torch first: it bundles its own HIP runtime (libamdhip64) and must be the one instance in the process,
    # otherwise a second runtime loaded from /opt/rocm sees no device and device pointers cannot be shared
    import torch  # noqa: F401

    path = key
    if not os.path.exists(path):
        raise ImportError(
            f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  The predictor has no CPU fallback.")
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    lib.has_diagnostics = b"diagnostics" in lib.jmid_version()
    if lib.has_diagnostics:
        for name, (res, args) in DIAG_SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
    _LIBS[key] = lib
    return lib

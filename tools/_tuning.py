"""Experiment scripts steer GEMM / depthwise launches issued from model code through PROCESS-WIDE switches, which the
product library does not have (no mutable globals). They run against libsei_hip_tuning.so instead:

    make -C scale-equivariant-imaging_amd/csrc tuning

`use()` points the ctypes binding at that build and registers its extra entry points; call it before any kernel."""
import ctypes
import os


def use():
    import _native
    path = os.path.join(os.path.dirname(_native.LIB_PATH), "libsei_hip_tuning.so")
    if not os.path.exists(path):
        raise SystemExit(f"{path} is missing: make -C {os.path.join(os.path.dirname(path), 'csrc')} tuning")
    _native.LIB_PATH = path
    _native.SIGNATURES["sei_debug_set_nt_tile"] = [ctypes.c_int]
    _native.SIGNATURES["sei_debug_tr_probe"] = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                                ctypes.c_void_p]
    _native._lib = None
    return _native.lib()

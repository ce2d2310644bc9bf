"""The C-ABI library builds, loads without a GPU, and exports exactly what include/sei_hip.h declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "sei_hip.h")


def declared():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"#ifdef SEI_TUNING.*?#endif", "", text, flags=re.S)     # tools-only build, not the product
    decls = {}
    for m in re.finditer(r"\b(?:int|size_t)\s+(sei_\w+)\s*\(([^;]*?)\)\s*;", text, flags=re.S):
        args = [a.strip() for a in m.group(2).split(",")]
        decls[m.group(1)] = 0 if args == ["void"] else len(args)
    return decls


def test_header_declares_entry_points():
    d = declared()
    assert len(d) >= 10 and "sei_blur_sep_circ" in d and "sei_abi_version" in d


def test_library_exports_every_declared_symbol():
    import _native
    assert os.path.exists(_native.LIB_PATH), "run `python -c 'import __graft_entry__ as g; g.build()'` first"
    handle = ctypes.CDLL(_native.LIB_PATH)       # loads on a GPU-less host: no HIP call at load time
    missing = [name for name in declared() if not hasattr(handle, name)]
    assert not missing, f"declared in sei_hip.h but not exported: {missing}"
    assert handle.sei_abi_version() == _native.ABI_VERSION == 12
    buf = ctypes.create_string_buffer(16)
    assert handle.sei_build_target(buf, 16) == 0 and buf.value == b"gfx950"


def test_product_library_has_no_debug_state():
    """re-entrant, no mutable globals. Schedule overrides are per-call arguments of the _ex entry
    points; the process-wide tuning switches and hardware probes exist only in the tools build (-DSEI_TUNING)."""
    import subprocess
    import _native
    handle = ctypes.CDLL(_native.LIB_PATH)
    for name in ("sei_debug_set_nt_tile", "sei_debug_set_dw_seg", "sei_debug_tr_probe"):
        assert not hasattr(handle, name), name
    syms = subprocess.run(["nm", "-D", "--defined-only", _native.LIB_PATH], capture_output=True, text=True).stdout
    exported = {ln.split()[-1] for ln in syms.splitlines() if ln.strip()}
    assert "sei_gemm_bf16nt_ex" in exported
    assert not [s_ for s_ in exported if "debug" in s_ or "gemm_bf16pp" in s_], exported
    # no writable override variable of any visibility survives in the product library
    allsyms = subprocess.run(["nm", _native.LIB_PATH], capture_output=True, text=True).stdout
    writable = [ln.split()[-1] for ln in allsyms.splitlines() if len(ln.split()) == 3 and ln.split()[1] in "bBdD"]
    assert not [w for w in writable if "force" in w or "tuning" in w or "g_dw" in w], writable


def test_python_binding_table_matches_header():
    import _native
    d = declared()
    sized = _native.SIZE_QUERIES                          # size queries: no stream argument, size_t result
    assert set(_native.SIGNATURES) | set(sized) == set(d)
    for name, nargs in d.items():
        table = sized if name in sized else _native.SIGNATURES
        assert len(table[name]) == nargs, name


def test_argument_errors_are_reported_not_launched():
    import _native
    L = _native.lib()
    # NULL pointers / bad sizes are rejected on the host before any HIP call (safe without a GPU)
    assert L.sei_blur_sep_circ(None, None, None, None, 13, 13, 1, 8, 8, 0, None) == 10001
    assert L.sei_axpy(None, None, 1.0, None, 4, None) == 10001
    assert L.sei_scale_resample_fwd(None, None, None, None, 1, 3, 8, 8, 8, 8, None) == 10001
    # ABI 12: the step prologue, the glue kernels and the split-bf16 passes refuse bad arguments on the host too
    assert L.sei_proposed_draws(1, 0, None, 2, 3, 48, 48, 6, None, 2, None, None, None, None) == 10001
    assert L.sei_proposed_draws(1, 2, 8, 2, 3, 48, 48, 6, 8, 2, 8, 8, 8, None) == 10001          # offset % 4 != 0
    assert L.sei_proposed_draws(1, 0, 8, 2, 3, 12, 12, 6, 8, 2, 8, 8, 8, None) == 10001         # margin swallows the image
    assert L.sei_proposed_draws(1, 0, 8, 4096, 3, 48, 48, 6, 8, 2, 8, 8, 8, None) == 10001      # beyond one element per thread
    assert L.sei_proposed_draws_max_numel() == 256 * 2048
    assert L.sei_crop_window(None, None, 6, 64, 64, 0, 0, 48, None) == 10001
    assert L.sei_stack_axpy(None, None, 0.01, None, 16, None) == 10001
    assert L.sei_concat2_f32(None, 16, None, 0, None, None) == 10001
    assert L.sei_scale_dev_f32(None, None, None, 16, None) == 10001
    assert L.sei_add_scalars(None, None, None, None) == 10001
    assert L.sei_split_bf16x2(None, None, 16, None) == 10001 and L.sei_split_bf16x2(16, 16, 6, None) == 10001   # n % 4
    assert L.sei_split_bf16x3(16, 16, 16, 2, None) == 10001                                     # pattern is 0 or 1
    assert L.sei_gelu_f32(None, None, 16, None) == 10001 and L.sei_mul_dgelu_f32(None, None, 16, None) == 10001


def test_product_refuses_cpu_tensors():
    """No CPU fallback: the product path fails loudly off-GPU."""
    import torch
    import _native
    import physics
    op = physics.BlurV2(kernel=physics.get_kernel("Gaussian_R2")[None, None])
    with pytest.raises(_native.NativeLibraryError):
        op.A(torch.rand(1, 3, 16, 16))


def test_gemm_plan_query_runs_without_a_gpu():
    """sei_gemm_bf16nt_plan is host arithmetic: the schedules of the timed batch's bottleneck GEMMs (B = 32: 576 / 288
    rows) and of the reference's default batch (B = 8: 144 / 72 rows), as tests/test_loss_gpu.py pins them on the GPU."""
    import _native
    BIAS_GELU, BIAS_RES, MUL_DGELU = 2, 3, 4
    assert _native.gemm_plan(0, 0, True, False, 576, 32768, 8192, BIAS_GELU) == ("pq", 288, 256, 1)
    assert _native.gemm_plan(0, 0, True, False, 288, 32768, 8192, BIAS_GELU)[:3] == ("pq", 288, 128)
    fam, bm, bn, sk = _native.gemm_plan(0, 0, True, False, 576, 8192, 32768, BIAS_RES)
    assert (fam, bm) == ("pq", 288) and sk > 1
    assert _native.gemm_plan(0, 1, False, True, 288, 32768, 8192, MUL_DGELU)[:2] == ("pq", 288)
    assert _native.gemm_plan(0, 0, True, False, 144, 32768, 8192, BIAS_GELU)[:3] == ("nt", 128, 128)
    with pytest.raises(_native.NativeLibraryError):
        _native.gemm_plan(0, 0, True, False, 0, 128, 64, 0)

"""ctypes binding of libsei_hip.so (C ABI declared in include/sei_hip.h).

This is the only place the Python host layer touches native code. There is NO fallback: if the
library is missing, or a tensor is not a contiguous float32 tensor on an AMD GPU, the call raises.
PyTorch is used for device memory and streams only: every call passes raw device pointers and the
current HIP stream, so the launches are captured by `torch.cuda.graph` like any other kernel.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# (SEI_HIP_LIBRARY: another build of the same ABI, for same-box A/B runs of tools/ and bench.py; never a fallback)
LIB_PATH = os.environ.get("SEI_HIP_LIBRARY") or os.path.join(_HERE, "libsei_hip.so")

SEI_REDUCE_BLOCKS = 256

_c = ctypes
_P, _I, _F, _Z, _L = _c.c_void_p, _c.c_int, _c.c_float, _c.c_size_t, _c.c_longlong

# name -> argument ctypes (all return int). Kept in the order of include/sei_hip.h;
# tests/test_abi.py checks this table against the header and against the built library.
SIGNATURES = {
    "sei_abi_version": [],
    "sei_event_create": [_P],
    "sei_event_destroy": [_P],
    "sei_event_record_external": [_P, _P],
    "sei_stream_wait_event": [_P, _P],
    "sei_graph_node_counts": [_P, _P, _P],
    "sei_build_target": [_c.c_char_p, _I],
    "sei_blur_sep_circ": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "sei_blur_dense_circ": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "sei_resample_sepband": [_P, _P, _I, _I, _I, _I, _I, _P, _P, _I, _I, _P, _P, _I, _I, _P],
    "sei_scale_params": [_P, _P, _P, _I, _I, _P, _P, _P],
    "sei_scale_resample_fwd": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "sei_scale_resample_bwd": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "sei_rotate_nearest_fwd": [_P, _P, _I, _I, _I, _F, _F, _F, _F, _P],
    "sei_rotate_nearest_bwd": [_P, _P, _I, _I, _I, _F, _F, _F, _F, _P],
    "sei_axpy": [_P, _P, _F, _P, _Z, _P],
    "sei_stack_axpy": [_P, _P, _F, _P, _Z, _P],
    "sei_proposed_draws": [_c.c_ulonglong, _c.c_ulonglong, _P, _I, _I, _I, _I, _I, _P, _I, _P, _P, _P, _P],
    "sei_crop_window": [_P, _P, _I, _I, _I, _I, _I, _I, _P],
    "sei_concat2_f32": [_P, _Z, _P, _Z, _P, _P],
    "sei_scale_dev_f32": [_P, _P, _P, _Z, _P],
    "sei_add_scalars": [_P, _P, _P, _P],
    "sei_zero_ranges": [_P, _P, _I, _P],
    "sei_sure_terms": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _F, _F, _P, _P, _P, _P, _P],
    "sei_mse_terms": [_P, _P, _Z, _F, _P, _P, _P, _P],
    "sei_sure_loss": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _F, _F, _F, _P, _P, _P, _P, _P],
    "sei_mse_loss": [_P, _P, _Z, _F, _F, _P, _P, _P, _P],
    "sei_luma_sqerr": [_P, _P, _Z, _P, _P, _P],
    "sei_conv3x3_fwd": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "sei_conv3x3_bwd_weight": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "sei_conv3x3_bwd_weight_parts": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "sei_dwconv7_fwd": [_P, _P, _P, _P, _F, _P, _I, _I, _I, _I, _I, _P],
    "sei_dwconv7_bwd_weight": [_P, _P, _P, _P, _I, _I, _I, _I, _P, _Z, _P],
    "sei_dwconv7_fwd_ex": [_P, _P, _P, _P, _F, _P, _I, _I, _I, _I, _I, _I, _P],
    "sei_dwconv7_bwd_weight_ex": [_P, _P, _P, _P, _I, _I, _I, _I, _P, _Z, _I, _P],
    "sei_dwconv7_ln_fwd": [_P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _I, _I, _I, _I, _F, _P],
    "sei_dwconv7_ln_fwd_ex": [_P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _I, _I, _I, _I, _F, _I, _P],
    "sei_ln_fwd": [_P, _P, _P, _P, _P, _P, _Z, _I, _F, _P],
    "sei_ln_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _Z, _I, _P, _Z, _P],
    "sei_ln_bwd_res": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _Z, _I, _P, _Z, _P],
    "sei_gemm_f32": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P],
    "sei_gemm_f32_ex": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _I, _L, _L, _L, _I, _P],
    "sei_gemm_bf16_ex": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _I, _L, _L, _L, _I, _P],
    "sei_gemm_bf16_mixed": [_P, _I, _P, _I, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _I, _L, _L, _L, _I, _P],
    "sei_cast_bf16": [_P, _P, _Z, _P],
    "sei_weight_shadow_bf16": [_P, _P, _P, _I, _I, _P],
    "sei_ln_fwd_bf16": [_P, _P, _P, _P, _P, _P, _Z, _I, _F, _P],
    "sei_colsum_bf16": [_P, _P, _Z, _I, _P],
    "sei_split_bf16x2": [_P, _P, _Z, _P],
    "sei_split_bf16x3": [_P, _P, _Z, _I, _P],
    "sei_gelu_f32": [_P, _P, _Z, _P],
    "sei_mul_dgelu_f32": [_P, _P, _Z, _P],
    "sei_cast_transpose_bf16": [_P, _I, _P, _P, _I, _I, _I, _P, _P],
    "sei_cast_bf16_colsum_weighted": [_P, _P, _P, _P, _I, _I, _P],
    "sei_cast_bf16_colsum_parts": [_P, _P, _P, _P, _I, _I, _P],
    "sei_gemm_bf16nt": [_P, _I, _I, _P, _I, _I, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P],
    "sei_gemm_bf16nt_colsum": [_P, _I, _I, _P, _I, _I, _P, _I, _I, _I, _I, _P, _P, _P],
    "sei_gemm_bf16nt_ws": [_P, _I, _I, _P, _I, _I, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _Z, _I, _I, _I, _P],
    "sei_fold_many": [_P, _I, _P],
    "sei_transpose_bf16_many": [_P, _I, _P],
    "sei_gemm_bf16nt_dw2": [_P, _P, _I, _P, _P, _I, _P, _I, _I, _I, _I, _I, _P],
    "sei_gemm_bf16nt_ex": [_P, _I, _I, _P, _I, _I, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _I, _I, _P],
    "sei_gemm_bf16nt_dw2_ex": [_P, _P, _I, _P, _P, _I, _P, _I, _I, _I, _I, _I, _I, _P],
    "sei_gemm_bf16nt_dw2_adam": [_P, _P, _I, _P, _P, _I, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "sei_gemm_bf16nt_dw2_adam_ex": [_P, _P, _I, _P, _P, _I, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "sei_gemm_bf16nt_dw2_bf16out": [_P, _P, _I, _P, _P, _I, _P, _I, _I, _I, _I, _P],
    "sei_gemm_bf16nt_dw2_taps": [_P, _P, _I, _P, _P, _I, _P, _I, _I, _I, _I, _I, _I, _P, _L, _P],
    "sei_sepmap2": [_P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _Z, _P],
    "sei_sepmap2_big_pack": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "sei_sepmap2_big": [_P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P],
    "sei_sepmap2_packed": [_P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P, _Z, _P],
    "sei_sepmap2_bf16": [_P, _P, _I, _I, _I, _I, _I, _I, _P, _P],
    "sei_sepmap2_bf16_pack": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "sei_sepmap2_small": [_P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P],
    "sei_sepmap2_bf16_out16": [_P, _P, _I, _I, _I, _I, _I, _I, _P, _P],
    "sei_colsum_f32": [_P, _P, _Z, _I, _P],
    "sei_colsum_weighted_f32": [_P, _P, _P, _Z, _I, _P],
    "sei_mlp_fused_fwd": [_P, _P, _P, _P, _P, _P, _F, _P, _I, _I, _P],
    "sei_mlp_fused_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P],
    "sei_swin_attn_fwd": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _F, _P],
    "sei_swin_attn_bwd": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _F, _P],
    "sei_swin_attn_fwd_bf16": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _P],
    "sei_swin_attn_bwd_bf16": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _P],
    "sei_pad_nhwc": [_P, _P, _I, _I, _I, _I, _I, _P],
    "sei_unpad_nhwc": [_P, _P, _P, _I, _I, _I, _I, _I, _P],
    "sei_rowscale": [_P, _P, _P, _P, _Z, _I, _P],
    "sei_pack": [_P, _P, _P, _Z, _I, _P],
    "sei_unpack_add": [_P, _P, _P, _Z, _P],
    "sei_ln_fwd_bf16_pad": [_P, _P, _P, _P, _P, _P, _Z, _I, _I, _F, _I, _P],
    "sei_ln_bwd_pad": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _Z, _I, _I, _P, _Z, _P],
    "sei_cast_pad_bf16": [_P, _P, _P, _P, _Z, _I, _I, _P, _Z, _P],
    "sei_pad_nhwc_bf16": [_P, _P, _I, _I, _I, _I, _I, _I, _P],
    "sei_pad_nhwc_bf16_ones": [_P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "sei_gemm_bf16nt_conv": [_P, _I, _P, _P, _I, _P, _P, _I, _I, _I, _P, _P],
    "sei_gemm_bf16nt_conv_unpad": [_P, _I, _P, _P, _I, _P, _P, _I, _I, _I, _I, _P, _I, _P],
    "sei_rowgemm_bf16": [_P, _I, _P, _I, _P, _I, _P, _I, _L, _I, _I, _I, _I, _P, _P, _P, _I, _P],
    "sei_rowgemm_lnbwd_bf16": [_P, _I, _P, _I, _L, _I, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _I, _P, _P, _Z, _P],
    "sei_rowgemm_dgelu_bf16": [_P, _I, _P, _I, _P, _I, _P, _I, _P, _I, _P, _I, _L, _I, _I, _P],
    "sei_rowgemm_ln_bf16": [_P, _I, _P, _I, _L, _I, _I, _P, _P, _P, _P, _P, _P, _F, _I, _P, _I, _P, _P, _P],
    "sei_rowgemm_gelu_bf16": [_P, _I, _P, _I, _P, _I, _P, _I, _L, _I, _I, _I, _P],
    "sei_tokgrad_bf16_blocks": [_P, _I, _L, _L, _P],
    "sei_tokgrad_bf16": [_P, _P, _I, _P, _P, _I, _P, _I, _I, _I, _L, _L, _P],
    "sei_dwstream_bf16_jobs": [_P, _I, _P],
    "sei_adam_fused": [_P, _P, _I, _P, _P, _Z, _F, _F, _F, _F, _F, _I, _F, _P, _P],
    "sei_adam_scalars": [_F, _F, _F, _F, _F, _I, _P, _P],
    "sei_adam_scalars_to_device": [_F, _F, _F, _F, _F, _I, _P, _P],
}

_lib = None


class TokGradBlock(_c.Structure):
    """SeiTokGradBlock of include/sei_hip.h (one 192 x 192 block of a token-streamed weight gradient)."""
    _fields_ = [("Y1", _P), ("Y2", _P), ("X1", _P), ("X2", _P), ("ldy", _I), ("ldx", _I), ("y0", _I), ("x0", _I),
                ("D", _P), ("ldd", _I)]


TOKGRAD_MAX_BLOCKS = 16      # SEI_TOKGRAD_MAX_BLOCKS


class DwStreamJob(_c.Structure):
    """SeiDwStreamJob of include/sei_hip.h (one weight gradient of a sei_dwstream_bf16_jobs table)."""
    _fields_ = [("Y1", _P), ("Y2", _P), ("X1", _P), ("X2", _P), ("ldy", _I), ("ldx", _I), ("Mo", _I), ("Ni", _I),
                ("D", _P), ("ldd", _I), ("reserved", _I), ("K1", _L), ("K2", _L), ("gbias", _P)]


DWSTREAM_MAX_JOBS = 64


class FoldJob(_c.Structure):
    """SeiFoldJob of include/sei_hip.h (one destination of sei_fold_many with the partial sums of up to three launches)."""
    _fields_ = [("a", _P), ("b", _P), ("c", _P), ("ncol", _I), ("split", _I), ("kind", _I), ("nseg", _I),
                ("part", _P * 3), ("groups", _I * 3), ("reserved", _I)]


FOLD_SPLIT, FOLD_DWCONV7, FOLD_MAX_JOBS = 0, 1, 48


class TransposeJob(_c.Structure):
    """SeiTransposeJob of include/sei_hip.h (one matrix of a sei_transpose_bf16_many table)."""
    _fields_ = [("src", _P), ("dst", _P), ("R", _I), ("C", _I)]


TRANSPOSE_MAX_JOBS = 16


# size queries: return size_t, take no stream
SIZE_QUERIES = {
    "sei_proposed_draws_max_numel": [],
    "sei_dwconv7_bwd_weight_workspace": [_I, _I, _I, _I],
    "sei_dwconv7_bwd_weight_workspace_ex": [_I, _I, _I, _I, _I],
    "sei_ln_bwd_workspace": [_Z, _I],
    "sei_dwconv7_ln_fwd_launches": [_I, _I, _I, _I],
    "sei_sepmap2_bf16_eligible": [_I, _I, _I, _I, _I, _I],
    "sei_sepmap2_bf16_pack_elems": [_I, _I, _I, _I],
    "sei_sepmap2_big_eligible": [_I, _I, _I, _I, _I, _I],
    "sei_sepmap2_big_pack_elems": [_I, _I, _I, _I],
    "sei_sepmap2_big_work_elems": [_I, _I, _I, _I, _I, _I],
    "sei_swin_partials_floats": [_I],
    "sei_tokgrad_bf16_eligible": [_I, _I, _I, _I, _L, _L],
    "sei_dwstream_bf16_eligible": [_I, _I, _I, _I, _L, _L],
    "sei_mlp_fused_eligible": [_L, _I],
    "sei_rowgemm_bf16_eligible": [_L, _I, _I, _I, _I],
    "sei_rowgemm_lnbwd_bf16_eligible": [_L, _I, _I],
    "sei_rowgemm_lnbwd_work_floats": [_I],
    "sei_ln_bwd_part_offset": [_Z, _I],
    "sei_ln_bwd_part_count": [_Z, _I],
    "sei_rowgemm_dgelu_bf16_eligible": [_L, _I, _I],
    "sei_rowgemm_ln_bf16_eligible": [_L, _I, _I],
    "sei_gemm_bf16nt_plan": [_I, _I, _I, _I, _I, _I, _I, _I],
    "sei_gemm_bf16nt_plan_ws": [_I, _I, _I, _I, _I, _I, _I, _I, _Z],
    "sei_conv3x3_bwd_weight_parts_count": [_I, _I, _I, _I, _I, _I, _I],
    "sei_sepmap2_small_eligible": [_I, _I, _I, _I, _I, _I],
    "sei_cast_bf16_colsum_parts_count": [_I, _I],
}
ABI_VERSION = 12      # SEI_ABI_VERSION of include/sei_hip.h this table was written against


class NativeLibraryError(RuntimeError):
    pass


def lib():
    """The loaded library; raises NativeLibraryError (never falls back) if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NativeLibraryError(
                f"{LIB_PATH} is missing: build it with `make -C {os.path.join(_HERE, 'csrc')}` "
                "(or `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback.")
        handle = ctypes.CDLL(LIB_PATH)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(handle, name, None)
            if fn is None:
                continue            # reported by tests/test_abi.py; a call would raise AttributeError
            fn.argtypes = argtypes
            fn.restype = _I
        for name, argtypes in SIZE_QUERIES.items():
            fn = getattr(handle, name)
            fn.argtypes = argtypes
            fn.restype = _Z
        if handle.sei_abi_version() != ABI_VERSION:
            raise NativeLibraryError("libsei_hip.so was built from a different include/sei_hip.h")
        _lib = handle
    return _lib


def gemm_plan(a_rmajor, b_rmajor, out_f32, out_bf16, M, Nn, K, epilogue, ws_bytes=0):
    """(family, tile rows, tile columns, K splits) of the launch sei_gemm_bf16nt would make (sei_gemm_bf16nt_plan: nothing
    is launched); family "nt" = gemm_bf16nt_kernel, "pq" = gemm_bf16pq_kernel. ws_bytes > 0: the launch sei_gemm_bf16nt_ws
    would make with a split-K workspace of that size, and a fifth element says whether the K slices meet in slabs.
    Raises on shapes the entry point refuses."""
    if ws_bytes:
        code = lib().sei_gemm_bf16nt_plan_ws(int(a_rmajor), int(b_rmajor), int(out_f32), int(out_bf16), M, Nn, K, epilogue,
                                             int(ws_bytes))
    else:
        code = lib().sei_gemm_bf16nt_plan(int(a_rmajor), int(b_rmajor), int(out_f32), int(out_bf16), M, Nn, K, epilogue)
    if code == 0:
        raise NativeLibraryError("sei_gemm_bf16nt_plan: arguments sei_gemm_bf16nt would refuse")
    plan = {1: "nt", 2: "pq"}[code >> 48], (code >> 32) & 0xFFFF, (code >> 16) & 0xFFFF, code & 0x7FFF
    return plan + (bool(code & 0x8000),) if ws_bytes else plan


def stream():
    return torch.cuda.current_stream().cuda_stream


def check_tensor(t, name="tensor", dtype=torch.float32):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a torch.Tensor, got {type(t).__name__}")
    if not t.is_cuda:
        raise NativeLibraryError(
            f"{name} is on {t.device}: this build runs on MI355X only (HIP kernels, no CPU path); "
            "use the CPU oracle under oracle/ for host-side checks")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name}: expected a contiguous tensor")
    return t


def ptr(t):
    return None if t is None else t.data_ptr()


_CALL_LOG = None


def record_calls(enable):
    """Measurement aid (bench.py): while enabled, every entry-point call is logged as (name, args) so that the
    launches of one step can be re-issued back to back between HIP events, family by family. Returns the log
    collected so far."""
    global _CALL_LOG
    log, _CALL_LOG = _CALL_LOG, ([] if enable else None)
    return log


def call(name, *args):
    """Invoke an entry point with the current stream appended; raise on a non-zero status."""
    if _CALL_LOG is not None:
        _CALL_LOG.append((name, args))
    rc = getattr(lib(), name)(*args, stream())
    if rc != 0:
        if rc == 10001:
            what = "SEI_ERR_BAD_ARG"
        elif rc == 10002:
            what = "SEI_ERR_TOO_LARGE"
        else:
            what = f"hipError_t {rc}"
        raise NativeLibraryError(f"{name} failed: {what}")


def scale_by(x, scalar):
    """x * scalar for a 0-dim device `scalar` (the incoming gradient of a loss value): one own launch where it applies."""
    if x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and x.numel() % 4 == 0 and isinstance(scalar, torch.Tensor) \
            and scalar.is_cuda and scalar.dtype == torch.float32 and scalar.numel() == 1:
        out = torch.empty_like(x)
        call("sei_scale_dev_f32", x.data_ptr(), scalar.data_ptr(), out.data_ptr(), x.numel())
        return out
    return x * scalar


def copy_into(dst, a, b=None):
    """dst <- a (b None) or the concatenation [a | b] along the leading dimension, as one own launch where it applies."""
    ok = all(t is None or (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.numel() % 4 == 0)
             for t in (dst, a, b))
    if ok and dst.numel() == a.numel() + (0 if b is None else b.numel()):
        call("sei_concat2_f32", a.data_ptr(), a.numel(), ptr(b), 0 if b is None else b.numel(), dst.data_ptr())
        return dst
    if b is None:
        dst.copy_(a.reshape(dst.shape))
    else:
        dst[:a.shape[0]].copy_(a)
        dst[a.shape[0]:].copy_(b)
    return dst


def graph_kernel_nodes(graph):
    """(kernel nodes, all nodes) of a torch.cuda.CUDAGraph captured with keep_graph=True."""
    kernels, total = _c.c_int(0), _c.c_int(0)
    rc = lib().sei_graph_node_counts(_c.c_void_p(int(graph.raw_cuda_graph())), _c.byref(kernels), _c.byref(total))
    if rc != 0:
        raise NativeLibraryError(f"sei_graph_node_counts failed: hipError_t {rc}")
    return kernels.value, total.value


class ExternalEvent:
    """A HIP event that may be recorded inside a captured hipGraph and waited for from another stream
    (what torch.cuda.Event(external=True) would be; PyTorch refuses that on ROCm)."""

    def __init__(self):
        handle = _c.c_void_p()
        rc = lib().sei_event_create(_c.byref(handle))
        if rc != 0:
            raise NativeLibraryError(f"sei_event_create failed: {rc}")
        self.handle = handle

    def record(self, torch_stream=None):
        s = (torch_stream or torch.cuda.current_stream()).cuda_stream
        rc = lib().sei_event_record_external(self.handle, s)
        if rc != 0:
            raise NativeLibraryError(f"sei_event_record_external failed: hipError_t {rc}")

    def wait(self, torch_stream=None):
        """Make `torch_stream` (default: the current one) wait for the most recent record."""
        s = (torch_stream or torch.cuda.current_stream()).cuda_stream
        rc = lib().sei_stream_wait_event(s, self.handle)
        if rc != 0:
            raise NativeLibraryError(f"sei_stream_wait_event failed: hipError_t {rc}")

    def __del__(self):
        try:
            if self.handle:
                lib().sei_event_destroy(self.handle)
        except Exception:
            pass

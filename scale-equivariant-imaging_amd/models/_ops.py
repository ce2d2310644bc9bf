"""Autograd functions of the U-Net blocks, each a short sequence of HIP kernel launches.

Activations are NHWC float32 tensors (B, H, W, C); a 1x1 convolution is then a row-major GEMM over
the (B*H*W, C) view. Parameter gradients are not returned to autograd: every backward ACCUMULATES
straight into `param.grad` (a view of the model's flat gradient bucket when the model has been
flattened), which is what lets the kernels fuse "grad += ..." and keeps one contiguous buffer for the
fused Adam step and the RCCL all-reduce. `optimizer.zero_grad()` (either flavour) is honoured.
"""
import os

import torch

import _native as N
from . import _mats

EPI_NONE, EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_RES, EPI_MUL_DGELU, EPI_ACCUM, EPI_BIAS_ROWSCALE = range(7)
LN_EPS = 1e-6


# ---------------------------------------------------------------------------------------------
# gradient buckets
# ---------------------------------------------------------------------------------------------
def grad_of(p):
    """The tensor to accumulate p's gradient into (zeroed on first touch after zero_grad)."""
    if p.grad is None:
        view = getattr(p, "_sei_grad_view", None)
        if view is None or view.shape != p.shape or view.device != p.device:
            view = torch.zeros_like(p, memory_format=torch.contiguous_format)
        else:
            view.zero_()
        p.grad = view
    return p.grad


# ---------------------------------------------------------------------------------------------
# activations of a RECORDED model call come from the recorder's arena (models/_joint.py: the step's two model calls
# share one backward pass over 3B-row tensors); everything else is torch.empty
# ---------------------------------------------------------------------------------------------
_ARENA = None            # the models._joint.Recorder of the model call being recorded, or None


def _alloc(shape, dtype, device):
    if _ARENA is None:
        return torch.empty(tuple(shape), dtype=dtype, device=device)
    return _ARENA.alloc(shape, dtype, device)


def _tape(fn, ctx):
    if _ARENA is not None:
        _ARENA.record(fn, ctx)


def _no_joint_form():
    """Called by layer functions that have no joint backward (the float32 path, stand-alone resamplers / LayerNorms)."""
    if _ARENA is not None:
        _ARENA.unsupported()


class recording:
    """`with recording(recorder):` around a model call: its layer functions allocate from the arena and fill the tape."""

    def __init__(self, rec):
        self.rec = rec

    def __enter__(self):
        global _ARENA
        self.prev, _ARENA = _ARENA, self.rec
        return self.rec

    def __exit__(self, *exc):
        global _ARENA
        _ARENA = self.prev
        return False


# ---------------------------------------------------------------------------------------------
# thin launch helpers (pointers + sizes only; shapes are checked here, on the host)
# ---------------------------------------------------------------------------------------------
_GEMM_PROFILE = None     # bench.py: list of (flops, entry point, ctypes args) recorded while enabled
# "bf16x3": the float32 layer functions with every GEMM evaluated as three bf16 MFMA products (gemm_x3 below)
_GEMM_ENTRY = {"f32": "sei_gemm_f32_ex", "bf16": "sei_gemm_bf16_ex", "bf16x3": None}
_COMPUTE_DTYPE = "f32"


_DTYPE_SCOPE = []            # innermost compute_dtype_scope (a model call or a backward function of one) wins


def set_compute_dtype(name):
    """The PROCESS DEFAULT of the arithmetic type of the 1x1-convolution GEMMs: "f32" (exact-f32 MFMA; the parity mode),
    "bf16" (bf16 MFMA with f32 accumulation; GEMM-only activations stored in bf16) or "bf16x3" (float32 storage and layer
    functions as "f32"; each GEMM as three bf16 MFMA products of bf16 head / remainder operands, f32 accumulation: 16
    mantissa bits per operand, the second parity mode). Everything else is f32 either way. A backbone
    may carry its own (`backbone.compute_dtype = "f32" | "bf16"`, None = the default): its forward pass and the backward
    functions it recorded run under it (`compute_dtype_scope`), so two models of different modes can live in one process."""
    global _COMPUTE_DTYPE
    if name not in _GEMM_ENTRY:
        raise ValueError(f"compute dtype must be one of {sorted(_GEMM_ENTRY)}, got {name!r}")
    previous, _COMPUTE_DTYPE = _COMPUTE_DTYPE, name
    return previous


def get_compute_dtype(owner=None):
    """The mode in effect: `owner`'s own (a backbone), else the innermost scope's, else the process default."""
    own = getattr(owner, "compute_dtype", None) if owner is not None else None
    if own is not None:
        return own
    return _DTYPE_SCOPE[-1] if _DTYPE_SCOPE else _COMPUTE_DTYPE


class compute_dtype_scope:
    """`with compute_dtype_scope(backbone_or_name):` -- the GEMMs issued inside use that backbone's mode (or the named
    one); a backbone without its own mode, or None, leaves the mode in effect unchanged."""

    def __init__(self, owner_or_name):
        name = owner_or_name if isinstance(owner_or_name, str) or owner_or_name is None \
            else getattr(owner_or_name, "compute_dtype", None)
        if name is not None and name not in _GEMM_ENTRY:
            raise ValueError(f"compute dtype must be one of {sorted(_GEMM_ENTRY)}, got {name!r}")
        self.name = name

    def __enter__(self):
        if self.name is not None:
            _DTYPE_SCOPE.append(self.name)
        return self

    def __exit__(self, *exc):
        if self.name is not None:
            _DTYPE_SCOPE.pop()
        return False


def _in_forward_mode(fn):
    """Decorator for the backward of an autograd Function whose forward stored `ctx.dtype = get_compute_dtype()`."""
    def backward(ctx, *grads):
        with compute_dtype_scope(getattr(ctx, "dtype", None)):
            return fn(ctx, *grads)
    return staticmethod(backward)


def profile_gemms(enable):
    """Record every GEMM launch (entry point + arguments + algorithmic FLOPs) while enabled, so that
    bench.py can re-issue exactly those launches back to back between HIP events (roofline leg)."""
    global _GEMM_PROFILE
    records, _GEMM_PROFILE = _GEMM_PROFILE, ([] if enable else None)
    return records


def _gemm_call(flops, entry, *args):
    if _GEMM_PROFILE is not None:
        _GEMM_PROFILE.append((flops, entry, args))
    N.call(entry, *args)


def gemm(A, Bm, M, Nn, K, ta, tb, epi, out=None, bias=None, R1=None, R2=None, D2=None, allow_splitk=True):
    if out is None:
        out = torch.empty((M, Nn), dtype=torch.float32, device=A.device)
    if get_compute_dtype() == "bf16x3" and _x3_ok(A, Bm, M, Nn, K, ta, tb, epi):
        return gemm_x3(A, Bm, M, Nn, K, ta, tb, epi, out, bias, R1, R2, D2)
    _gemm_call(2.0 * M * Nn * K, _GEMM_ENTRY[get_compute_dtype()] or "sei_gemm_f32_ex", A.data_ptr(), Bm.data_ptr(), out.data_ptr(), M, Nn, K,
               ta, tb, epi, N.ptr(bias), N.ptr(R1), N.ptr(R2), N.ptr(D2), 1, 0, 0, 0, int(allow_splitk))
    return out


_JOINT_SPLIT = None      # (B1, B2) while one backward pass serves the step's two model calls (joint_rows), else None

# Split-K workspace of the quadrant GEMM (sei_gemm_bf16nt_ws, include/sei_hip.h): tile counters in its first 16 KiB (zero
# between launches: every launch leaves them zero) + the slabs the K slices meet in. The ticket protocol indexes counters
# and slabs by tile ordinal alone, so two launches in flight at once must never share one: a workspace belongs to ONE
# (device, stream) -- launches on a stream are a chain. Eager launches find theirs in a small registry (created on the
# stream's first eager use; the least recently used one is dropped when a device has SPLITK_WS_STREAMS of them). A stream
# that is CAPTURING and has none gets none (its 256-MiB zero fill would be replayed with the graph): those launches take
# the float-atomics path (sei_gemm_bf16nt / _colsum). graphs.GraphedLossStep therefore warms up and captures on one side
# stream with a workspace OF ITS OWN (`own_splitk_workspace` ... `release_splitk_workspace`: registered for that stream
# while the step is being built, kept alive by the graph's owner afterwards): its replays -- serialised by the graph
# itself -- never share it with anybody, whatever stream they are launched from, and torch handing the same pooled stream
# to somebody else later cannot alias it. SEI_SPLITK_WS_MIB = 0 switches the workspace off everywhere (round-1..5 path).
SPLITK_WS_MIB = int(os.environ.get("SEI_SPLITK_WS_MIB", "256"))
SPLITK_WS_STREAMS = 4
SPLITK_COUNTER_BYTES = 16 << 10
_SPLITK_WS = {}              # (device index, stream handle) -> uint8 tensor; insertion order = least recently used first


def _splitk_key(device):
    dev = torch.device(device).index
    if dev is None:
        dev = torch.cuda.current_device()
    return dev, torch.cuda.current_stream(dev).cuda_stream


def splitk_workspace(device):
    """(pointer, bytes) of the split-K workspace of (`device`, its current stream), or (None, 0): switched off, or the
    stream is capturing without one."""
    if SPLITK_WS_MIB <= 0:
        return None, 0
    key = _splitk_key(device)
    ws = _SPLITK_WS.pop(key, None)
    if ws is None:
        if torch.cuda.is_current_stream_capturing():
            return None, 0
        mine = [k for k in _SPLITK_WS if k[0] == key[0]]
        if len(mine) >= SPLITK_WS_STREAMS:
            del _SPLITK_WS[mine[0]]              # (freed in stream order by the caching allocator: its launches are queued)
        ws = torch.zeros(SPLITK_WS_MIB << 20, dtype=torch.uint8, device=f"cuda:{key[0]}")
    _SPLITK_WS[key] = ws                         # most recently used last
    return ws.data_ptr(), ws.numel()


def own_splitk_workspace(device):
    """A NEW workspace registered for (`device`, its current stream), returned to the caller, who keeps it alive for as
    long as launches recorded on this stream may run (a captured graph) and calls release_splitk_workspace when it has
    finished recording. None when switched off."""
    if SPLITK_WS_MIB <= 0:
        return None
    key = _splitk_key(device)
    ws = _SPLITK_WS[key] = torch.zeros(SPLITK_WS_MIB << 20, dtype=torch.uint8, device=f"cuda:{key[0]}")
    return ws


def release_splitk_workspace(device, stream, ws):
    """Take `ws` (from own_splitk_workspace on `stream`) out of the registry: later eager launches on that stream handle
    get a workspace of their own."""
    dev = torch.device(device).index
    if dev is None:
        dev = torch.cuda.current_device()
    key = (dev, stream.cuda_stream)
    if ws is not None and _SPLITK_WS.get(key) is ws:
        del _SPLITK_WS[key]


def reset_splitk_counters(device):
    """Zero the tile counters of every workspace of `device`, each on the stream that owns it (a launch that never
    finished -- a fault survived by the process -- would leave tickets behind, and no slice would ever draw the last one).
    Not under capture. graphs.GraphedLossStep calls it before its warm-up."""
    dev = torch.device(device).index
    if dev is None:
        dev = torch.cuda.current_device()
    for (d, handle), ws in _SPLITK_WS.items():
        if d == dev:
            with torch.cuda.stream(torch.cuda.ExternalStream(handle, device=f"cuda:{dev}") if handle else
                                   torch.cuda.default_stream(dev)):
                ws[:SPLITK_COUNTER_BYTES].zero_()


def gemm_nt16(A16, B16, M, Nn, K, epi, out32=None, out16=None, bias=None, R1=None, R2=None, D2_16=None,
              lda=None, ldb=None, a_rmajor=False, b_rmajor=False, tile=0, band=0, flops=None, _whole=False, colsum=None):
    """D[M,N] = op(A16) op(B16) on the direct-to-LDS bf16 kernel. A16 is (M,K) [or (K,M) when a_rmajor],
    B16 is (N,K) [or (K,N) when b_rmajor]; K % 8 == 0. Outputs as given. tile / band: an explicit schedule
    choice (sei_gemm_bf16nt_ex; tests and tools), 0 = the library's dispatch. flops: the algorithmic FLOP count
    to book for the roofline leg when the operands are zero-padded (default 2 M N K). colsum (float32, (N,)): += the column
    sums of the bf16 result out16 (sei_gemm_bf16nt_colsum: a bias gradient riding in the data gradient's epilogue)."""
    # (the CONTRACTING data gradients gh2 = gh3 W2 -- float32 out, K = 4 N, split K -- run faster on the joint rows: their
    # K splits fill the rounds whatever the row count. tools/exp_joint_rows.py, merged vs the two launches: 3456 x 2048 x 8192
    # 153 vs 104 + 71 us, 864 x 8192 x 32768 479 vs 310 + 187, 13824 x 512 x 2048 44 vs 37 + 30; the expanding ones with
    # GELU' lose merged: 3456 x 8192 x 2048 147 vs 85 + 51, 864 x 32768 x 8192 492 vs 287 + 169.)
    contracting = (out32 is not None and out16 is None and epi in (EPI_NONE, EPI_ACCUM) and b_rmajor and K == 4 * Nn
                   and (M >= 3456 or K >= 8192))               # (also 864 x 2048 x 8192: 61 vs 53 + 39, exp_joint_rows2.py)
    if _JOINT_SPLIT is not None and not _whole and not a_rmajor and max(Nn, K) >= 2048 and not (tile or band) and lda is None \
            and not contracting:
        # One backward pass over the rows of both model calls (models/_joint.py) -- but the deep levels' GEMMs are tuned to
        # the row counts of the separate calls (2304 / 1152 and 576 / 288 rows are whole rounds of 288-row tiles on 256 CUs;
        # 3456 and 864 rows are 1.5 rounds: measured 168 us against 79 + 58): their rows go as the two launches they were
        B1, B2 = _JOINT_SPLIT
        M1 = M * B1 // (B1 + B2)
        if 0 < M1 < M and M1 % 8 == 0 and (M - M1) % 8 == 0:
            # an epilogue operand is cut with the rows when it is per-row data: an (M, Nn) matrix in any view with M * Nn
            # elements, or BIAS_ROWSCALE's M-vector -- never by a leading dimension that happens to equal M
            def cut(t, lo, hi):
                if t is None:
                    return None
                if t.dim() == 1 and t.numel() == M:
                    return t[lo:hi]
                if t.numel() == M * Nn:
                    return t.reshape(M, Nn)[lo:hi]
                return t
            for lo, hi in ((0, M1), (M1, M)):
                gemm_nt16(A16[lo:hi], B16, hi - lo, Nn, K, epi, out32=cut(out32, lo, hi), out16=cut(out16, lo, hi), bias=bias,
                          R1=cut(R1, lo, hi), R2=cut(R2, lo, hi), D2_16=cut(D2_16, lo, hi),
                          ldb=ldb, b_rmajor=b_rmajor, flops=None if flops is None else flops * (hi - lo) / M, _whole=True,
                          colsum=colsum)
            return
    if lda is None:
        lda = M if a_rmajor else K
    if ldb is None:
        ldb = Nn if b_rmajor else K
    args = (A16.data_ptr(), lda, int(a_rmajor), B16.data_ptr(), ldb, int(b_rmajor), N.ptr(out32), N.ptr(out16), M, Nn,
            K, epi, N.ptr(bias), N.ptr(R1), N.ptr(R2), N.ptr(D2_16))
    fl = 2.0 * M * Nn * K if flops is None else float(flops)
    ws, ws_bytes = (None, 0) if (tile or band or a_rmajor) else splitk_workspace(A16.device)
    if colsum is not None:
        if out32 is not None or out16 is None or epi not in (EPI_NONE, EPI_MUL_DGELU) or tile or band or bias is not None \
                or R2 is not None or D2_16 is not None:
            raise ValueError("gemm_nt16(colsum=): a bf16 result with EPI_NONE / EPI_MUL_DGELU on the automatic dispatch")
        if ws is not None:
            _gemm_call(fl, "sei_gemm_bf16nt_ws", *args, colsum.data_ptr(), ws, ws_bytes, 0, 0, 0)
            return
        _gemm_call(fl, "sei_gemm_bf16nt_colsum", A16.data_ptr(), lda, int(a_rmajor), B16.data_ptr(), ldb, int(b_rmajor),
                   out16.data_ptr(), M, Nn, K, epi, N.ptr(R1), colsum.data_ptr())
        return
    if tile or band:
        _gemm_call(fl, "sei_gemm_bf16nt_ex", *args, int(tile), int(band))
    elif ws is not None:
        # (K slices of the quadrant kernel meet in slabs of the workspace: no zero fill, no float atomics)
        _gemm_call(fl, "sei_gemm_bf16nt_ws", *args, None, ws, ws_bytes, 0, 0, 0)
    else:
        _gemm_call(fl, "sei_gemm_bf16nt", *args)


# ---------------------------------------------------------------------------------------------
# --compute_dtype bf16x3: a float32 GEMM as three bf16 MFMA products (csrc/bf16x3.hip has the arithmetic and the error
# bound). Operands are split once into bf16 head / remainder PLANES ((2, rows, cols): either plane is a dense operand of
# sei_gemm_bf16nt in whatever orientation the float32 GEMM read the tensor), weights once per optimizer step. The three
# launches accumulate into the float32 result, small terms first; additive epilogues (bias, residuals, the running
# gradient) ride on the first launch, GELU / GELU' follow as element-wise passes.
# ---------------------------------------------------------------------------------------------
def _x3_ok(A, Bm, M, Nn, K, ta, tb, epi):
    """Shapes sei_gemm_bf16nt takes (K % 8, 16-byte rows of the reduction-major operands); anything else -- the 3-channel
    ends of the network never come here -- stays on the float32 GEMM."""
    if not (A.is_cuda and A.dtype == torch.float32 and Bm.dtype == torch.float32 and K % 8 == 0 and Nn % 4 == 0):
        return False
    if (ta and M % 8) or (not tb and Nn % 8) or A.numel() % 4 or Bm.numel() % 4 or not A.is_contiguous() \
            or not Bm.is_contiguous():
        return False
    return epi in (EPI_NONE, EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_RES, EPI_MUL_DGELU, EPI_ACCUM, EPI_BIAS_ROWSCALE)


def split_x2(t):
    """(2, *t.shape) bf16: head and remainder planes of a float32 tensor. A parameter's planes are cached until its
    values change (this model's optimizer kernel / load_state_dict: `_generation`, torch's version counter) -- and rebuilt
    once inside a capture, so that every replay splits the weights of ITS step."""
    def fresh():
        planes = torch.empty((2,) + tuple(t.shape), dtype=torch.bfloat16, device=t.device)
        N.call("sei_split_bf16x2", t.data_ptr(), planes.data_ptr(), t.numel())
        return planes
    if not isinstance(t, torch.nn.Parameter):
        return fresh()
    capturing = torch.cuda.is_current_stream_capturing()
    key = (_generation(getattr(t, "_sei_plain_state", None)), t._version, t.data_ptr(), capturing)
    hit = getattr(t, "_sei_split", None)
    if hit is None or hit[0] != key:
        hit = (key, fresh())
        t._sei_split = hit
    return hit[1]


def gemm_x3(A, Bm, M, Nn, K, ta, tb, epi, out, bias=None, R1=None, R2=None, D2=None):
    """gemm()'s contract (float32 operands as stored: A (M, K) or (K, M) when ta, Bm (N, K) when tb else (K, N))."""
    arm, brm = bool(ta), not tb
    if arm and brm and epi == EPI_ACCUM:
        # a weight gradient (both operands reduction-major, a large output that is read-modify-written): the three
        # products in ONE launch whose reduction runs over the stacked planes, K' = 3 K
        a3 = torch.empty((3 * K, M), dtype=torch.bfloat16, device=A.device)
        b3 = torch.empty((3 * K, Nn), dtype=torch.bfloat16, device=A.device)
        N.call("sei_split_bf16x3", A.data_ptr(), a3.data_ptr(), A.numel(), 0)
        N.call("sei_split_bf16x3", Bm.data_ptr(), b3.data_ptr(), Bm.numel(), 1)
        gemm_nt16(a3, b3, M, Nn, 3 * K, EPI_ACCUM, out32=out, a_rmajor=True, b_rmajor=True, flops=2.0 * M * Nn * K)
        return out
    a2, b2 = split_x2(A).view(2, -1), split_x2(Bm).view(2, -1)
    (a_hi, a_lo), (b_hi, b_lo) = a2, b2
    third = 2.0 * M * Nn * K / 3.0                    # (algorithmic FLOPs are booked once over the three launches)
    first = {EPI_BIAS_GELU: EPI_BIAS, EPI_MUL_DGELU: EPI_NONE}.get(epi, epi)
    gemm_nt16(a_lo, b_hi, M, Nn, K, first, out32=out, bias=bias, R1=R1 if first != EPI_NONE else None,
              R2=R2 if first == EPI_BIAS_RES else None, a_rmajor=arm, b_rmajor=brm, flops=third)
    gemm_nt16(a_hi, b_lo, M, Nn, K, EPI_ACCUM, out32=out, a_rmajor=arm, b_rmajor=brm, flops=third)
    gemm_nt16(a_hi, b_hi, M, Nn, K, EPI_ACCUM, out32=out, a_rmajor=arm, b_rmajor=brm, flops=third)
    if epi == EPI_BIAS_GELU:
        N.call("sei_gelu_f32", out.data_ptr(), D2.data_ptr(), out.numel())
    elif epi == EPI_MUL_DGELU:
        N.call("sei_mul_dgelu_f32", out.data_ptr(), R1.data_ptr(), out.numel())
    return out


def layer_norm(x2d, gamma, beta):
    rows, C = x2d.shape
    y = _alloc((rows, C), torch.float32, x2d.device)
    mean = _alloc((rows,), torch.float32, x2d.device)
    rstd = _alloc((rows,), torch.float32, x2d.device)
    N.call("sei_ln_fwd", x2d.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), mean.data_ptr(),
           rstd.data_ptr(), rows, C, LN_EPS)
    return y, mean, rstd


def layer_norm_bwd(x2d, gamma, mean, rstd, gy, ggamma, gbeta, res=None):
    """res: a second gradient of the LayerNorm's input (rows, C), added to gx in the same kernel (the skip connection's
    gradient where a level's output feeds the downsampler and the decoder: no autograd add kernel)."""
    rows, C = x2d.shape
    gx = torch.empty_like(x2d)
    need = N.lib().sei_ln_bwd_workspace(rows, C)
    work = torch.empty(max(need, 1), dtype=torch.float32, device=x2d.device)
    parts = N.lib().sei_ln_bwd_part_count(rows, C)
    deferred = parts > 0 and defer_fold(ggamma, gbeta, None, 2 * C, C, N.FOLD_SPLIT, work,
                                        N.lib().sei_ln_bwd_part_offset(rows, C), parts)
    fused_res = res is not None and parts > 0
    N.call("sei_ln_bwd_res", x2d.data_ptr(), gamma.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gy.data_ptr(),
           res.data_ptr() if fused_res else None, gx.data_ptr(), None if deferred else ggamma.data_ptr(),
           None if deferred else gbeta.data_ptr(), rows, C, work.data_ptr(), need)
    if res is not None and not fused_res:
        gx += res.view(rows, C)
    return gx


# ---------------------------------------------------------------------------------------------
# Leaf launches beside the chain (round 5). A backward pass is a CHAIN of launches (each layer's data gradient feeds the
# next) with LEAVES hanging off it: parameter gradients that nothing else in the pass reads -- bias column sums, the
# depthwise and 3x3 weight gradients. At this model's sizes both are latency-bound launches of 10-60 us that leave most of
# the chip idle. Leaves are issued on a side stream that waits for everything enqueued so far (their operands) and is
# joined when the pass ends (the engine callback that flushes parked pairs and deferred folds, `flush_weight_grads`): in
# the captured step they become a parallel branch of the hipGraph and run under the chain. Operands are kept alive until
# the join (the caching allocator may not hand their memory to the chain meanwhile).
# MEASURED AND OFF BY DEFAULT (SEI_LEAF_STREAM=1 switches it on): same box, same run, the captured step took 13.22 ms with
# the ~20 leaves (0.5 ms of launches) on the branch against 12.97 ms in line -- as with the Adam-epilogue GEMMs on a second
# branch in rounds 2 and 4, what the branch gains in idle CUs the fork / join edges and the shared memory system take back.
# ---------------------------------------------------------------------------------------------
LEAF_STREAM = os.environ.get("SEI_LEAF_STREAM") == "1"
_LEAF_SIDE = {}


def leaf_call(grad, name, *args, keep=()):
    """N.call(name, *args) for a launch that only produces (part of) the parameter gradient `grad`."""
    _DW = _state_for(grad.data_ptr()) if grad.is_cuda else None
    if not LEAF_STREAM or _DW is None or not _queue_flush(_DW):
        N.call(name, *args)                     # (no backward pass running: nobody would join the side stream)
        return
    dev = grad.device
    side = _LEAF_SIDE.get(dev)
    if side is None:
        side = _LEAF_SIDE[dev] = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        N.call(name, *args)
    _DW.setdefault("leaf", []).append((side, keep))


def _join_leaves(_DW):
    leaves = _DW.pop("leaf", None)
    if leaves:
        torch.cuda.current_stream(leaves[0][0].device).wait_stream(leaves[0][0])


def colsum_into(acc, x2d, row_weight=None, leaf=True):
    """acc[n] += sum_m x2d[m, n] (times row_weight[m] if given). `acc` is a parameter (bias) gradient of the U-Net: a leaf
    launch. leaf=False (the Swin blocks, whose staged gradients are unpacked inside the pass): in line."""
    M, Nn = x2d.shape
    call = leaf_call if leaf else (lambda _g, name, *a, keep=(): N.call(name, *a))
    if row_weight is None:
        call(acc, "sei_colsum_f32", x2d.data_ptr(), acc.data_ptr(), M, Nn, keep=(x2d,))
    else:
        call(acc, "sei_colsum_weighted_f32", x2d.data_ptr(), row_weight.data_ptr(), acc.data_ptr(), M, Nn,
             keep=(x2d, row_weight))


def colsum_into_inline(acc, x2d, row_weight=None):
    colsum_into(acc, x2d, row_weight, leaf=False)


def dwconv7(x, w, bias, flip=False, res=None, res_scale=1.0, seg=0):
    """seg > 0: the generic kernels with that segment width (sei_dwconv7_fwd_ex; tests), 0 = chosen by shape."""
    B, H, W, C = x.shape
    y = torch.empty_like(x)
    args = (x.data_ptr(), w.data_ptr(), N.ptr(bias), N.ptr(res), float(res_scale), y.data_ptr(), B, H, W, C, int(flip))
    if seg:
        N.call("sei_dwconv7_fwd_ex", *args, int(seg))
    else:
        N.call("sei_dwconv7_fwd", *args)
    return y


def dwconv7_ln(x, w, bias, gamma, beta, out16, fuse=0):
    """ConvBlock.conv1 -> LayerNorm (reference convolutional.py:36-39) through sei_dwconv7_ln_fwd: one fused launch
    at the first level (C = 32), the depthwise kernel + the stand-alone LayerNorm elsewhere (fuse: 1 / 2 force either
    form; tests). Returns h1 (f32, NHWC), h2 ((M, C) bf16 when out16 else f32), mean, rstd."""
    B, H, W, C = x.shape
    M = B * H * W
    h1 = _alloc((B, H, W, C), torch.float32, x.device)
    h2 = _alloc((M, C), torch.bfloat16 if out16 else torch.float32, x.device)
    mean = _alloc((M,), torch.float32, x.device)
    rstd = _alloc((M,), torch.float32, x.device)
    args = (x.data_ptr(), w.data_ptr(), N.ptr(bias), gamma.data_ptr(), beta.data_ptr(), h1.data_ptr(), h2.data_ptr(),
            int(out16), mean.data_ptr(), rstd.data_ptr(), B, H, W, C, LN_EPS)
    if fuse:
        N.call("sei_dwconv7_ln_fwd_ex", *args, int(fuse))
    else:
        N.call("sei_dwconv7_ln_fwd", *args)
    return h1, h2, mean, rstd


def dwconv7_weight_grad(x, gy, gw, gb, seg=0):
    B, H, W, C = x.shape
    need = N.lib().sei_dwconv7_bwd_weight_workspace_ex(B, H, W, C, int(seg))
    work = torch.empty(need, dtype=torch.float32, device=x.device)
    deferred = need > 0 and defer_fold(gw, gb, None, 50 * C, C, N.FOLD_DWCONV7, work, 0, need // (50 * C))
    leaf_call(gw, "sei_dwconv7_bwd_weight_ex", x.data_ptr(), gy.data_ptr(), None if deferred else gw.data_ptr(),
              None if deferred else N.ptr(gb), B, H, W, C, work.data_ptr(), need, int(seg), keep=(x, gy, work))


def sepmap2(x, mats, Ho, Wo):
    """mats: (L1, R1, L2, R2[, RW, LH]) -- the packed pair comes with `_mats.resample_matrices`; a bare 4-tuple
    (tests, experiments) is packed here."""
    B, Hi, Wi, C = x.shape
    y = _alloc((B, Ho, Wo, C), torch.float32, x.device)
    work = torch.empty(2 * B * Hi * Wo * C, dtype=torch.float32, device=x.device)
    RW, LH = (mats[4], mats[5]) if len(mats) >= 6 else _mats.pack_for_kernel(mats[:4], x.device)
    N.call("sei_sepmap2_packed", x.data_ptr(), y.data_ptr(), B, Hi, Wi, Ho, Wo, C, RW.data_ptr(), LH.data_ptr(),
           work.data_ptr(), work.numel())
    return y


# SEI_SEPMAP_F32=1 keeps the resamplers of the bf16 mode on the f32 FMA kernels (A/B runs)
_SEPMAP_MFMA = __import__("os").environ.get("SEI_SEPMAP_F32") != "1"
_SEPMAP_SMALL = __import__("os").environ.get("SEI_NO_SEPMAP_SMALL") != "1"     # (A/B runs: the two-launch f32 kernels instead)


def sepmap2_16(x, mats, Ho, Wo, out16=False):
    """sepmap2 in the bf16 throughput mode: on the matrix cores where the shape is eligible (sei_sepmap2_bf16:
    activations rounded to bf16, matrices as bf16 head + remainder, f32 accumulation), else the f32 kernels.
    out16: the caller wants the result as a bf16 GEMM operand; the kernels that can write it directly (the one-pass kernel of
    the deepest levels, the matrix-core kernel of the 24 - 64-pixel extents) return a bfloat16 tensor -- the float32
    accumulator rounded once, exactly what a cast pass would have produced -- the others float32 (the caller casts)."""
    B, Hi, Wi, C = x.shape
    if _SEPMAP_SMALL and x.is_cuda and N.lib().sei_sepmap2_small_eligible(B, Hi, Wi, Ho, Wo, C):
        # the deep levels' 6- and 3-pixel images: one float32 pass through LDS, no HBM intermediate (round 5)
        y = _alloc((B, Ho, Wo, C), torch.bfloat16 if out16 else torch.float32, x.device)
        L1, R1, L2, R2 = mats[:4]
        N.call("sei_sepmap2_small", x.data_ptr(), y.data_ptr(), int(out16), B, Hi, Wi, Ho, Wo, C, L1.data_ptr(),
               R1.data_ptr(), L2.data_ptr(), R2.data_ptr())
        return y
    small = _SEPMAP_MFMA and x.is_cuda and max(Hi, Wi, Ho, Wo) <= 64 and N.lib().sei_sepmap2_bf16_eligible(B, Hi, Wi, Ho, Wo, C)
    big = not small and _SEPMAP_MFMA and x.is_cuda and N.lib().sei_sepmap2_big_eligible(B, Hi, Wi, Ho, Wo, C)
    if not big and _SEPMAP_MFMA and x.is_cuda and N.lib().sei_sepmap2_bf16_eligible(B, Hi, Wi, Ho, Wo, C):
        y = _alloc((B, Ho, Wo, C), torch.bfloat16 if out16 else torch.float32, x.device)
        N.call("sei_sepmap2_bf16_out16" if out16 else "sei_sepmap2_bf16", x.data_ptr(), y.data_ptr(), B, Hi, Wi, Ho, Wo, C,
               _packed16(mats).data_ptr())
        return y
    if big:
        # extents beyond one workgroup's LDS (the x4 network's 96- / 192-pixel levels, 256-pixel inputs): two launches of
        # the constant-matrix GEMM kernel with a bf16 intermediate
        y = _alloc((B, Ho, Wo, C), torch.float32, x.device)
        work = torch.empty(N.lib().sei_sepmap2_big_work_elems(B, Hi, Wi, Ho, Wo, C), dtype=torch.int16, device=x.device)
        N.call("sei_sepmap2_big", x.data_ptr(), y.data_ptr(), B, Hi, Wi, Ho, Wo, C, _packed16(mats, big=True).data_ptr(),
               work.data_ptr())
        return y
    return sepmap2(x, mats, Ho, Wo)


_PACKED16 = {}


def _packed16(mats, big=False):
    """The map's matrices in sei_sepmap2_bf16's (big: sei_sepmap2_big's) image (bf16 head + remainder), packed once per
    matrix set."""
    L1, R1, L2, R2 = mats[:4]
    key = (L1.data_ptr(), R1.data_ptr(), L2.data_ptr(), R2.data_ptr(), tuple(L1.shape), tuple(R1.shape), big)
    hit = _PACKED16.get(key)
    if hit is None:
        (Ho, Hi), (Wo, Wi) = L1.shape, R1.shape
        kind = "sei_sepmap2_big" if big else "sei_sepmap2_bf16"
        out = torch.empty(getattr(N.lib(), kind + "_pack_elems")(Hi, Wi, Ho, Wo), dtype=torch.int16, device=L1.device)
        N.call(kind + "_pack", L1.data_ptr(), R1.data_ptr(), L2.data_ptr(), R2.data_ptr(), out.data_ptr(), Hi, Wi,
               Ho, Wo)
        hit = _PACKED16[key] = (out, L1, R1, L2, R2)          # (keeps the sources alive: the key holds their addresses)
    return hit[0]


def _nhwc(x):
    N.check_tensor(x, "activation")
    if x.dim() != 4:
        raise ValueError("expected an NHWC activation (B, H, W, C)")
    return x


# ---------------------------------------------------------------------------------------------
# ConvBlock: x + conv3(gelu(conv2(LN(dwconv7(x)))))   (reference convolutional.py:33-51)
# `twice` adds the block input a second time: the encoder's inner residual x + xb with xb == x
# (convolutional.py:226-231) fused into the last GEMM's epilogue.
# ---------------------------------------------------------------------------------------------
class ConvBlockFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w1, b1, gamma, beta, w2, b2, w3, b3, twice):
        _no_joint_form()
        ctx.dtype = get_compute_dtype()
        x = _nhwc(x)
        B, H, W, C = x.shape
        M = B * H * W
        h1, h2, mean, rstd = dwconv7_ln(x, w1, b1, gamma, beta, out16=False)
        h4 = torch.empty((M, 4 * C), dtype=torch.float32, device=x.device)
        h3 = gemm(h2, w2, M, 4 * C, C, 0, 1, EPI_BIAS_GELU, bias=b2, D2=h4)
        out = gemm(h4, w3, M, C, 4 * C, 0, 1, EPI_BIAS_RES, bias=b3, R1=x, R2=x if twice else None)
        ctx.save_for_backward(x, h1, mean, rstd, h2, h3, h4)
        ctx.params = (w1, b1, gamma, beta, w2, b2, w3, b3)
        ctx.twice = twice
        return out.view(B, H, W, C)

    @_in_forward_mode
    def backward(ctx, go):
        x, h1, mean, rstd, h2, h3, h4 = ctx.saved_tensors
        w1, b1, gamma, beta, w2, b2, w3, b3 = ctx.params
        B, H, W, C = x.shape
        M = B * H * W
        go = go.contiguous()
        go2 = go.view(M, C)
        # conv3
        colsum_into(grad_of(b3), go2)
        gemm(go2, h4, C, 4 * C, M, 1, 0, EPI_ACCUM, out=grad_of(w3).view(C, 4 * C))
        gh3 = gemm(go2, w3, M, 4 * C, C, 0, 0, EPI_MUL_DGELU, R1=h3)          # (go W3) * gelu'(h3)
        # conv2
        colsum_into(grad_of(b2), gh3)
        gemm(gh3, h2, 4 * C, C, M, 1, 0, EPI_ACCUM, out=grad_of(w2).view(4 * C, C))
        gh2 = gemm(gh3, w2, M, C, 4 * C, 0, 0, EPI_NONE)
        # LayerNorm, depthwise conv
        gh1 = layer_norm_bwd(h1.view(M, C), gamma, mean, rstd, gh2, grad_of(gamma), grad_of(beta))
        gh1 = gh1.view(B, H, W, C)
        dwconv7_weight_grad(x, gh1, grad_of(w1), grad_of(b1))
        gx = None
        if ctx.needs_input_grad[0]:
            gx = dwconv7(gh1, w1, None, flip=True, res=go, res_scale=2.0 if ctx.twice else 1.0)
        return (gx,) + (None,) * 9


# ---------------------------------------------------------------------------------------------
# Downsample: LN -> 1x1 conv C -> 4C -> ideal downsample    (reference convolutional.py:136-150)
# ---------------------------------------------------------------------------------------------
class DownsampleFn(torch.autograd.Function):
    """LN -> 1x1 conv C->Co -> ideal downsample, evaluated as LN -> ideal downsample -> 1x1 conv.

    The resampler is linear and acts per channel, the convolution is linear and acts per pixel, so they
    commute exactly; only the bias needs care: it comes out of the resampler as bias[c] * s[pixel], s = the
    resampler's response to a constant image (`_mats.constant_response`), which is the BIAS_ROWSCALE epilogue.
    The resampler then runs on C channels instead of Co = 4C, the three GEMMs on a quarter of the rows, and the
    full-resolution Co-channel tensor (75 MB per level at B = 32) never exists."""

    @staticmethod
    def forward(ctx, x, gamma, beta, w, b, rate, with_skip=False):
        """with_skip: also return x itself as a second output (the U-Net's skip connection): its gradient then arrives
        HERE, next to the downsampler's, and is added inside the LayerNorm backward instead of by an autograd add."""
        _no_joint_form()
        ctx.dtype = get_compute_dtype()
        ctx.set_materialize_grads(False)
        x = _nhwc(x)
        B, H, W, C = x.shape
        M, Co = B * H * W, w.shape[0]
        h, mean, rstd = layer_norm(x.view(M, C), gamma, beta)
        fwd, bwd = _mats.resample_matrices("down", H, W, rate, x.device)
        Ho, Wo = fwd[0].shape[0], fwd[1].shape[0]
        u = sepmap2(h.view(B, H, W, C), fwd, Ho, Wo)
        Mo = B * Ho * Wo
        s = _mats.constant_response("down", H, W, rate, x.device, B)
        out = gemm(u.view(Mo, C), w, Mo, Co, C, 0, 1, EPI_BIAS_ROWSCALE, bias=b, R1=s)
        ctx.save_for_backward(x, mean, rstd, u, s)
        ctx.params, ctx.mats_t, ctx.hw = (gamma, beta, w, b), bwd, (H, W, Ho, Wo)
        out = out.view(B, Ho, Wo, Co)
        return (out, x) if with_skip else out

    @_in_forward_mode
    def backward(ctx, go, gskip=None):
        x, mean, rstd, u, s = ctx.saved_tensors
        gamma, beta, w, b = ctx.params
        B, H, W, C = x.shape
        Ho, Wo = ctx.hw[2], ctx.hw[3]
        M, Mo, Co = B * H * W, B * Ho * Wo, w.shape[0]
        go2 = go.contiguous().view(Mo, Co)
        colsum_into(grad_of(b), go2, row_weight=s)
        gemm(go2, u.view(Mo, C), Co, C, Mo, 1, 0, EPI_ACCUM, out=grad_of(w).view(Co, C))
        gu = gemm(go2, w, Mo, C, Co, 0, 0, EPI_NONE)
        gh = sepmap2(gu.view(B, Ho, Wo, C), ctx.mats_t, H, W).view(M, C)
        res = None if gskip is None else gskip.contiguous().view(M, C)
        gx = layer_norm_bwd(x.view(M, C), gamma, mean, rstd, gh, grad_of(gamma), grad_of(beta), res=res).view(B, H, W, C)
        return (gx if ctx.needs_input_grad[0] else None), None, None, None, None, None, None


# ---------------------------------------------------------------------------------------------
# Upsample: ideal upsample -> LN -> 1x1 conv (+ skip)       (reference convolutional.py:95-110,236-240)
# ---------------------------------------------------------------------------------------------
class UpsampleFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, skip, gamma, beta, w, b, rate):
        _no_joint_form()
        ctx.dtype = get_compute_dtype()
        x = _nhwc(x)
        B, H, W, C = x.shape
        Co = w.shape[0]
        fwd, bwd = _mats.resample_matrices("up", H, W, rate, x.device)
        Ho, Wo = fwd[0].shape[0], fwd[1].shape[0]
        u = sepmap2(x, fwd, Ho, Wo)
        M = B * Ho * Wo
        h, mean, rstd = layer_norm(u.view(M, C), gamma, beta)
        if skip is not None:
            skip = _nhwc(skip)
            if tuple(skip.shape) != (B, Ho, Wo, Co):
                raise ValueError("skip connection shape mismatch")
            out = gemm(h, w, M, Co, C, 0, 1, EPI_BIAS_RES, bias=b, R1=skip)
        else:
            out = gemm(h, w, M, Co, C, 0, 1, EPI_BIAS, bias=b)
        ctx.save_for_backward(u, mean, rstd, h)
        ctx.params, ctx.mats_t, ctx.in_hw = (gamma, beta, w, b), bwd, (H, W)
        return out.view(B, Ho, Wo, Co)

    @_in_forward_mode
    def backward(ctx, go):
        u, mean, rstd, h = ctx.saved_tensors
        gamma, beta, w, b = ctx.params
        B, Ho, Wo, C = u.shape
        M, Co = B * Ho * Wo, w.shape[0]
        go = go.contiguous()
        go2 = go.view(M, Co)
        colsum_into(grad_of(b), go2)
        gemm(go2, h, Co, C, M, 1, 0, EPI_ACCUM, out=grad_of(w).view(Co, C))
        gh = gemm(go2, w, M, C, Co, 0, 0, EPI_NONE)
        gu = layer_norm_bwd(u.view(M, C), gamma, mean, rstd, gh, grad_of(gamma), grad_of(beta))
        gx = None
        if ctx.needs_input_grad[0]:
            gx = sepmap2(gu.view(B, Ho, Wo, C), ctx.mats_t, *ctx.in_hw)
        gskip = go if ctx.needs_input_grad[1] else None
        return gx, gskip, None, None, None, None, None


# =============================================================================================
# bf16 throughput mode (--compute_dtype bf16): GEMM-only activations are STORED in bf16, weights get
# bf16 shadows (plain + transposed) refreshed once per optimizer step, and every GEMM whose operands can
# be K-contiguous runs on the direct-to-LDS kernel (sei_gemm_bf16nt). Weight gradients use the
# register-staged kernel on the bf16 tensors and accumulate into the f32 gradient bucket. Everything
# else (depthwise conv, LayerNorm statistics and backward, resamplers, residuals, Adam) stays f32.
# =============================================================================================
_GLOBAL_GENERATION = 0      # bumped by weights_updated() without a model: every model's cached bf16 copies are suspect


def _new_plain_state():
    """Validity of a model's bf16 bucket (`flat_shadow`), tracked PER MODEL: "wgen" = this model's weight generation
    (bumped whenever its parameters change behind torch's version counters: its own optimizer kernel, its own
    load_state_dict), "gen" = the (global, own) generation pair the bf16 bucket was last written for, "version" = torch's
    version counter of each parameter at that time, "stale" = (start, stop) of the bucket whose float32 masters are OUT OF
    DATE on this rank (sharded optimizer step: only the bf16 copies of other ranks' shares were gathered) or None."""
    return {"gen": None, "wgen": 0, "version": {}, "stale": None}


def _generation(plain):
    return (_GLOBAL_GENERATION, plain["wgen"] if plain is not None else 0)


def weights_updated(backbone=None, plain_shadow_written=False):
    """Parameters changed outside torch's version counters. With a `backbone` only THAT model's cached bf16 copies are
    invalidated (another model's optimizer step or load_state_dict must not make this one recast its weights: under a
    sharded optimizer step the float32 masters of other ranks' shares are stale and a recast would overwrite good bf16
    weights with old values); without one, every model's. `plain_shadow_written`: the optimizer kernel also refreshed
    `backbone.flat_shadow` (the bf16 copy of every parameter), so that copy is current for the new generation."""
    global _GLOBAL_GENERATION
    if backbone is None:
        _GLOBAL_GENERATION += 1
        return
    plain = backbone._sei_plain_state
    plain["wgen"] += 1
    if plain_shadow_written:
        plain["gen"] = _generation(plain)


def plain_shadow_is_current(backbone):
    return backbone._sei_plain_state["gen"] == _generation(backbone._sei_plain_state)


def set_stale_masters(backbone, span):
    """optim.FlatAdam (sharded step): float32 parameters inside bucket range `span` are stale on this rank until
    `consolidate()`; None clears it. While set, nothing may rebuild bf16 copies of that range from the masters."""
    backbone._sei_plain_state["stale"] = None if span is None else (int(span[0]), int(span[1]))


def _stale_error():
    return RuntimeError("the float32 weights of other ranks' shares are out of date on this rank (sharded optimizer step: "
                        "only their bf16 copies were all-gathered) and something asked for bf16 copies to be rebuilt from "
                        "them; call optimizer.consolidate() on every rank before changing or re-reading the weights")


def refresh_plain_shadow(backbone):
    """Cast the whole flat parameter bucket to its bf16 copy (what the fused Adam does as a side output)."""
    if getattr(backbone, "flat_shadow", None) is None:
        return
    if backbone._sei_plain_state["stale"] is not None:
        raise _stale_error()
    N.call("sei_cast_bf16", backbone.flat_params.data_ptr(), backbone.flat_shadow.data_ptr(),
           backbone.flat_params.numel())
    plain = backbone._sei_plain_state
    plain["gen"] = _generation(plain)


def shadow(p):
    """bf16 copy w16 (R,C) of a 1x1-conv weight p (R,C,1,1): a view of the owning model's flat bf16 bucket,
    which the fused Adam kernel rewrites every step; cast here only when that copy is not current.
    (No transposed copy exists: the data-gradient GEMM reads w16 reduction-major.)"""
    plain = getattr(p, "_sei_plain_state", None)
    key = (_generation(plain), p._version, p.data_ptr())
    st = getattr(p, "_sei_shadow", None)
    if st is None or st[0] != key:
        R, C = p.shape[0], p.shape[1]
        flat16 = getattr(p, "_sei_shadow_view", None)
        if st is not None and st[1].device == p.device:
            w16 = st[1]
        else:
            w16 = flat16.view(R, C) if flat16 is not None else torch.empty((R, C), dtype=torch.bfloat16, device=p.device)
        # the bucket copy is current when it was written for this generation and torch has not changed p since it was
        # last looked at here
        plain_current = (flat16 is not None and plain is not None and plain["gen"] == key[0]
                         and plain["version"].get(id(p)) == p._version)
        if not plain_current:
            stale = plain["stale"] if plain is not None else None
            if stale is not None:
                off = getattr(p, "_sei_bucket_offset", None)
                if off is None or (off < stale[1] and stale[0] < off + p.numel()):
                    raise _stale_error()
            N.call("sei_cast_bf16", p.data_ptr(), w16.data_ptr(), p.numel())
        if plain is not None:
            plain["version"][id(p)] = p._version
        st = (key, w16)
        p._sei_shadow = st
    return st[1]


def nt16_ok(K):
    return K % 8 == 0


def to_bf16(x):
    y = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    N.call("sei_cast_bf16", x.data_ptr(), y.data_ptr(), x.numel())
    return y


def layer_norm16(x2d, gamma, beta):
    rows, C = x2d.shape
    y = _alloc((rows, C), torch.bfloat16, x2d.device)
    mean = _alloc((rows,), torch.float32, x2d.device)
    rstd = _alloc((rows,), torch.float32, x2d.device)
    N.call("sei_ln_fwd_bf16", x2d.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), mean.data_ptr(),
           rstd.data_ptr(), rows, C, LN_EPS)
    return y, mean, rstd


def colsum16_into(acc, x16, leaf=True):
    M, Nn = x16.shape
    if leaf:
        leaf_call(acc, "sei_colsum_bf16", x16.data_ptr(), acc.data_ptr(), M, Nn, keep=(x16,))
    else:
        N.call("sei_colsum_bf16", x16.data_ptr(), acc.data_ptr(), M, Nn)


def colsum16_into_inline(acc, x16):
    colsum16_into(acc, x16, leaf=False)


def gemm_mixed(A, Bm, M, Nn, K, ta, tb, epi, out=None, bias=None, R1=None, R2=None, D2=None, allow_splitk=True):
    """Register-staged bf16-MFMA GEMM; each operand may be stored as float32 or bfloat16."""
    if out is None:
        out = torch.empty((M, Nn), dtype=torch.float32, device=A.device)
    _gemm_call(2.0 * M * Nn * K, "sei_gemm_bf16_mixed", A.data_ptr(), int(A.dtype == torch.bfloat16), Bm.data_ptr(),
               int(Bm.dtype == torch.bfloat16), out.data_ptr(), M, Nn, K, ta, tb, epi, N.ptr(bias), N.ptr(R1),
               N.ptr(R2), N.ptr(D2), 1, 0, 0, 0, int(allow_splitk))
    return out


CAST_COLSUM_PARTS = os.environ.get("SEI_CAST_COLSUM_ATOMICS") != "1"      # (A/B: the round-1..5 atomics form)


def cast16(x2d, colsum_into_=None, row_weight=None):
    """f32 (R, C) -> bf16 copy; colsum_into_: accumulate the column sums (a bias gradient) in the same pass, each row
    times row_weight[r] when given (the downsampler's convolution: DownsampleFn16)."""
    R, C = x2d.shape
    x16 = _alloc((R, C), torch.bfloat16, x2d.device)
    if colsum_into_ is not None and CAST_COLSUM_PARTS and C % 4 == 0:
        # the column sums leave as per-row-block partial sums and join the pass's deferred folds (no atomics: the grid is
        # sized for bandwidth); outside a backward pass, or switched off, the atomics form below
        parts = N.lib().sei_cast_bf16_colsum_parts_count(R, C)
        ms = _state_for(colsum_into_.data_ptr())["milestone"]
        # (a bias gradient between the two early-released weight gradients must be final when their event fires: such a
        # destination -- none in the U-Net, whose only bias there comes from a GEMM epilogue -- keeps the atomics)
        early = ms is not None and min(ms[0]) <= colsum_into_.data_ptr() <= max(ms[0])
        if parts > 0 and not early:
            work = torch.empty(parts * C, dtype=torch.float32, device=x2d.device)
            if defer_fold(colsum_into_, None, None, C, C, N.FOLD_SPLIT, work, 0, parts):
                N.call("sei_cast_bf16_colsum_parts", x2d.data_ptr(), x16.data_ptr(), N.ptr(row_weight), work.data_ptr(), R, C)
                return x16
    if row_weight is not None and colsum_into_ is not None and C % 4 == 0:
        N.call("sei_cast_bf16_colsum_weighted", x2d.data_ptr(), x16.data_ptr(), row_weight.data_ptr(), colsum_into_.data_ptr(),
               R, C)
        return x16
    if row_weight is not None and colsum_into_ is not None:
        colsum_into(colsum_into_, x2d, row_weight=row_weight)
        colsum_into_ = None
    N.call("sei_cast_transpose_bf16", x2d.data_ptr(), 0, x16.data_ptr(), None, R, C, R, N.ptr(colsum_into_))
    return x16


# ---------------------------------------------------------------------------------------------
# weight gradients of the 1x1 convolutions, merged across the model calls of one step
#
# ProposedLoss calls the model twice per step (the fused SURE pass on 2B crops and the EI pass on B); each
# call's backward used to read-modify-write every weight gradient, which at the deep levels (268 M weights,
# 1.07 GB of float32 gradient each) is HBM traffic, not arithmetic. Instead the first backward to reach a
# weight parks its (gy, x) pair, and the second one issues ONE GEMM whose reduction runs over both pairs
# (sei_gemm_bf16nt_dw2). Pairs still parked when autograd finishes are flushed by an engine callback, so
# `p.grad` is complete whenever .backward() returns, whoever consumes it.
#
# "Store" mode (opt-in, GraphedLossStep): the first launch of a step into a weight gradient stores instead
# of accumulating, and zero_grad skips those gradients -- valid because the captured step writes every one
# of them exactly this way on every replay.
# ---------------------------------------------------------------------------------------------
def _fresh_state():
    return {"uses": 0, "arrivals": {}, "parked": {}, "written": set(), "store": False, "store_min": 0,
            "flush_queued": False, "merge": True, "seen": {}, "milestone": None, "milestone_done": False,
            "adam": None, "adam_launched": set(), "direct16": None, "direct16_launched": set(), "flops_per_row": {},
            "taps": {}, "merged_ok": {}, "folds": {}, "dwjobs": [], "bias_of": {}}


class WeightGradState:
    """The bookkeeping below, PER BACKBONE: model calls of the step, parked pairs, store / fused-Adam / direct-bf16
    tables. Two models alive in one process (a frozen copy beside the fine-tuned one, two networks trained side by
    side) each merge and store their own weight gradients. A backbone's state is found from the address of the
    gradient a backward function writes (`register_gradient_range`: the flat gradient bucket, SwinPack's staging);
    gradients outside every registered range (kernel-level tests) share the default state."""

    def __init__(self):
        self.d = _fresh_state()


_DEFAULT_STATE = WeightGradState()
_RANGES = []                      # [(first byte, end byte, weakref to the WeightGradState)], newest last


def state_of(owner=None):
    """The state dict of `owner` (a backbone with a flat bucket; created on first use), or the default one."""
    if owner is None:
        return _DEFAULT_STATE.d
    st = owner.__dict__.get("_sei_dw_state")
    if st is None:
        st = WeightGradState()
        owner.__dict__["_sei_dw_state"] = st
    return st.d


def register_gradient_range(owner, tensor):
    """Gradients written inside `tensor` (the owner's flat gradient bucket, a staging buffer) belong to `owner`."""
    import weakref
    state_of(owner)
    lo = tensor.data_ptr()
    _RANGES[:] = [r for r in _RANGES if r[2]() is not None and not (r[0] < lo + tensor.numel() * tensor.element_size()
                                                                    and lo < r[1])]
    _RANGES.append((lo, lo + tensor.numel() * tensor.element_size(), weakref.ref(owner.__dict__["_sei_dw_state"])))


def _state_for(ptr):
    for lo, hi, ref in reversed(_RANGES):
        if lo <= ptr < hi:
            st = ref()
            if st is not None:
                return st.d
    return _DEFAULT_STATE.d


def note_forward(owner=None):
    """A model call that autograd will differentiate (ConvolutionalModel.forward / SwinIR.forward pass themselves)."""
    if torch.is_grad_enabled():
        state_of(owner)["uses"] += 1


def begin_step(store=False, store_min=0, owner=None):
    """Start of a step (zero_grad): nothing parked, no model call counted, no gradient written yet. store: the
    first launch of the step into a weight gradient of at least store_min elements stores (it was not zeroed)."""
    _DW = state_of(owner)
    flush_weight_grads(owner)
    rec = owner.__dict__.get("_sei_joint") if owner is not None else None
    if rec is not None:
        rec.reset()
    _DW["uses"] = 0
    _DW["arrivals"].clear()
    _DW["written"].clear()
    _DW["milestone_done"] = False
    _DW["store"] = bool(store)
    _DW["store_min"] = int(store_min)


def set_weight_grad_merging(enabled, owner=None):
    _DW = state_of(owner)
    previous, _DW["merge"] = _DW["merge"], bool(enabled)
    return previous


def weight_grad_views(reset=False, owner=None):
    """{data_ptr: numel} of every gradient view written through weight_grad16 since the last reset."""
    _DW = state_of(owner)
    seen = dict(_DW["seen"])
    if reset:
        _DW["seen"].clear()
        _DW["taps"].clear()
        _DW["merged_ok"].clear()
        _DW["flops_per_row"].clear()
    return seen


def set_weight_grad_milestone(keys, event, owner=None):
    """Record `event` on the current stream right after the LAST of the gradients `keys` (data_ptrs) has been
    launched in a step (GraphedLossStep: an external event inside the captured backward, after which the
    bottleneck block's gradients -- most of the bucket -- are final and their all-reduce may start)."""
    state_of(owner)["milestone"] = (frozenset(keys), event) if keys else None


# The token-streaming kernels of token_gemm.hip (Swin blocks) against the tiled GEMMs they replace: on unless
# SEI_SWIN_TILED=1 (tests flip the flag to run the same step on both paths; shapes the streaming kernels do not take --
# token counts that are not multiples of 64 -- use the tiled ones anyway).
TOKEN_STREAMING = os.environ.get("SEI_SWIN_TILED") != "1"
CONV_TOKGRAD = os.environ.get("SEI_NO_CONV_TOKGRAD") != "1"    # the 192-channel convolutions' tap gradients as token-streamed blocks


def _check_milestone(_DW, key):
    ms = _DW["milestone"]
    if ms is not None and key in ms[0] and ms[0] <= _DW["written"] and not _DW["milestone_done"]:
        # every gradient of the milestone has had its (single, merged) launch of this step
        if all(_DW["arrivals"].get(k, 0) >= max(_DW["uses"], 1) for k in ms[0]):
            # a milestone gradient that was only QUEUED for the streamed launch (flush at the end of the backward pass)
            # must be on the stream before the event that releases it to the reducer
            if any(job[1][4].data_ptr() in ms[0] for job in _DW["dwjobs"]):
                flush_dwstream(_DW)
            ms[1].record()
            _DW["milestone_done"] = True


def _launch_weight_grad(_DW, grad2d, pairs):
    key = grad2d.data_ptr()
    store = _DW["store"] and key not in _DW["written"] and grad2d.numel() >= _DW["store_min"]
    _DW["written"].add(key)
    try:
        _launch_weight_grad_inner(_DW, grad2d, pairs, store)
    finally:
        _check_milestone(_DW, key)


def _launch_weight_grad_group(_DW, segs):
    """The weight gradients of several layers over the same tokens -- segs: one list of (gy16, x16, grad2d, flops per
    row) per model call of the step, the layers in the same order -- as ONE token-streamed launch
    (sei_tokgrad_bf16_blocks: every 192 x 192 block of every gradient on its share of the CUs) when the shapes allow
    it and every gradient accumulates; one launch per layer otherwise."""
    first = segs[0]
    rows = [seg[0][0].shape[0] for seg in segs]
    blocks, ok = [], TOKEN_STREAMING and len(segs) <= 2 and all(r % 64 == 0 for r in rows)
    for i, (_, _, grad2d, _) in enumerate(first):
        key = grad2d.data_ptr()
        ok = ok and grad2d.dim() == 2 and grad2d.is_contiguous() and grad2d.dtype == torch.float32
        ok = ok and not (_DW["store"] and key not in _DW["written"] and grad2d.numel() >= _DW["store_min"])
        ok = ok and not (_DW["adam"] is not None and key in _DW["adam"][0])
        ok = ok and not (_DW["direct16"] is not None and key in _DW["direct16"]) and key not in _DW["taps"]
        for s, seg in enumerate(segs):
            gy, x = seg[i][0], seg[i][1]
            ok = ok and gy.dtype == torch.bfloat16 and x.dtype == torch.bfloat16 and gy.is_contiguous() and x.is_contiguous()
            ok = ok and gy.shape == (rows[s], grad2d.shape[0]) and x.shape == (rows[s], grad2d.shape[1])
        if ok:
            ok = N.lib().sei_tokgrad_bf16_eligible(grad2d.shape[0], grad2d.shape[1], grad2d.shape[0], grad2d.shape[1],
                                                   rows[0], rows[1] if len(rows) == 2 else 0) != 0
        if not ok:
            break
        Mo, Ni = grad2d.shape
        a, b = segs[0][i], segs[-1][i]
        for gy in range(Mo // 192):
            for gx in range(Ni // 192):
                blocks.append(N.TokGradBlock(a[0].data_ptr(), b[0].data_ptr(), a[1].data_ptr(), b[1].data_ptr(), Mo, Ni,
                                             192 * gy, 192 * gx, grad2d.data_ptr() + 4 * (192 * gy * Ni + 192 * gx), Ni))
    if not ok or len(blocks) > 8:
        for i, (_, _, grad2d, _) in enumerate(first):
            _launch_weight_grad(_DW, grad2d, [(seg[i][0], seg[i][1]) for seg in segs])
        return
    flops = 0.0
    for _, _, grad2d, fpr in first:
        key = grad2d.data_ptr()
        _DW["written"].add(key)
        _DW["merged_ok"][key] = len(segs) == 2 and len(segs) == _DW["uses"]
        flops += (fpr or 2.0 * grad2d.shape[0] * grad2d.shape[1]) * sum(rows)
    arr = (N.TokGradBlock * len(blocks))(*blocks)
    try:
        _gemm_call(flops, "sei_tokgrad_bf16_blocks", arr, len(blocks), rows[0], rows[1] if len(rows) == 2 else 0)
    finally:
        for _, _, grad2d, _ in first:
            _check_milestone(_DW, grad2d.data_ptr())


def set_fused_adam(table, hyper, owner=None):
    """Optimizer step inside the weight-gradient GEMM (optim.FlatAdam.fuse_weight_updates): `table` maps the
    data_ptr of a gradient view to the (param, exp_avg, exp_avg_sq, bf16 shadow or None) views of the same shape,
    `hyper` is the device array of the step's six Adam scalars. The step's single, merged, storing launch into such a
    gradient applies the update instead of writing the gradient (sei_gemm_bf16nt_dw2_adam); None switches it off."""
    _DW = state_of(owner)
    _DW["adam"] = (dict(table), hyper) if table else None
    _DW["adam_launched"] = set()


def set_direct_bf16_grads(table, owner=None):
    """Several GPUs, bf16-compressed exchange (parallel.FlatGradientReducer): `table` maps the data_ptr of a gradient
    view to the bf16 view of the exchange buffer with the same shape. The step's single, merged, storing launch into
    such a gradient writes bf16 there (sei_gemm_bf16nt_dw2_bf16out) and nothing into the float32 bucket, which the
    reducer then does not cast for those ranges; None switches it off."""
    _DW = state_of(owner)
    _DW["direct16"] = dict(table) if table else None
    _DW["direct16_launched"] = set()


def direct_bf16_launches(owner=None):
    return set(state_of(owner)["direct16_launched"])


def fused_adam_launches(owner=None):
    """data_ptrs whose update was applied inside a GEMM since set_fused_adam."""
    return set(state_of(owner)["adam_launched"])


def merged_weight_grads(owner=None):
    """data_ptrs of the gradients whose last launch carried the step's COMPLETE gradient as one two-segment GEMM (the
    condition for applying the optimizer step, or writing bf16, in that launch: graphs.GraphedLossStep reads this after
    its warm-up steps -- a batch whose pixel counts do not add up to a multiple of 8 rows is served by other launches)."""
    return {k for k, ok in state_of(owner)["merged_ok"].items() if ok}


def _launch_weight_grad_inner(_DW, grad2d, pairs, store):
    """(weight_grad16(..., bias=): the bias gradient = gy's column sums rides in the streamed launch where that serves the
    weight, else it is summed here, one column-sum launch per pair.)"""
    bias = _DW["bias_of"].pop(grad2d.data_ptr(), None)
    if bias is not None and not (not store and _dwstream_ok(grad2d, pairs)):
        for gy16, _ in pairs:
            colsum16_into(bias, gy16, leaf=False)
        bias = None
    _launch_weight_grad_inner2(_DW, grad2d, pairs, store, bias)


def _launch_weight_grad_inner2(_DW, grad2d, pairs, store, bias):
    key = grad2d.data_ptr()
    Np, Kp = grad2d.shape[-2:]
    _DW["merged_ok"][key] = (len(pairs) == 2 and len(pairs) == _DW["uses"]
                             and (pairs[0][0].shape[0] + pairs[1][0].shape[0]) % 8 == 0)
    fused = _DW["adam"]
    if fused is not None and key in fused[0]:
        # the update replaces the stored gradient only when this launch IS the step's whole gradient
        complete = store and len(pairs) == 2 and len(pairs) == _DW["uses"]
        (g1, x1), (g2, x2) = pairs if len(pairs) == 2 else (pairs[0], pairs[0])
        K1, K2 = g1.shape[0], g2.shape[0]
        if not complete or (K1 + K2) % 8 != 0 or key in _DW["adam_launched"]:
            raise RuntimeError("fused optimizer step: a weight registered with set_fused_adam did not receive its "
                               "gradient as one merged storing GEMM (different loss / batch / call count than planned)")
        prm, m1, v1, sh = fused[0][key]
        _DW["adam_launched"].add(key)
        _gemm_call(2.0 * Np * Kp * (K1 + K2), "sei_gemm_bf16nt_dw2_adam", g1.data_ptr(), g2.data_ptr(), Np,
                   x1.data_ptr(), x2.data_ptr(), Kp, prm.data_ptr(), m1.data_ptr(), v1.data_ptr(), N.ptr(sh),
                   fused[1].data_ptr(), Np, Kp, K1, K2)
        return
    direct = _DW["direct16"]
    if direct is not None and key in direct:
        complete = store and len(pairs) == 2 and len(pairs) == _DW["uses"]
        (g1, x1), (g2, x2) = pairs if len(pairs) == 2 else (pairs[0], pairs[0])
        K1, K2 = g1.shape[0], g2.shape[0]
        if not complete or (K1 + K2) % 8 != 0 or key in _DW["direct16_launched"]:
            raise RuntimeError("bf16 gradients written into the exchange buffer: a registered weight did not receive its "
                               "gradient as one merged storing GEMM (different loss / batch / call count than planned)")
        _DW["direct16_launched"].add(key)
        _gemm_call(2.0 * Np * Kp * (K1 + K2), "sei_gemm_bf16nt_dw2_bf16out", g1.data_ptr(), g2.data_ptr(), Np,
                   x1.data_ptr(), x2.data_ptr(), Kp, direct[key].data_ptr(), Np, Kp, K1, K2)
        return
    taps = _DW["taps"].get(key)
    if taps is not None:                               # (T, N', K') gradient: every tap in one launch
        T, Np, Kp = grad2d.shape
        (g1, x1), (g2, x2) = pairs if len(pairs) == 2 else (pairs[0], pairs[0])
        K1, K2 = g1.shape[0], (g2.shape[0] if len(pairs) == 2 else 0)
        per_row = _DW["flops_per_row"].get(key) or 2.0 * T * Np * Kp
        if (K1 + K2) % 8 != 0:
            raise ValueError("weight_grad16 with taps: the reduction length must be a multiple of 8 rows")
        if TOKEN_STREAMING and CONV_TOKGRAD and not store and Np == 192 and Kp == 192 and T <= N.TOKGRAD_MAX_BLOCKS \
                and K1 % 64 == 0 and K2 % 64 == 0 \
                and g1.stride(0) == Np and x1.stride(0) == Kp and grad2d.is_contiguous():
            # 192-channel convolutions (the body of the SwinIR network): the taps are blocks of ONE token-streamed launch --
            # the same gy rows against the input rows shifted by each tap's offset, every block with its share of the CUs,
            # the two operands crossing an XCD's L2 once for all nine (sei_tokgrad_bf16_blocks; the tiled kernel re-stages
            # both operands for every 128 x 128 tile of every tap)
            blocks = [N.TokGradBlock(g1.data_ptr(), g2.data_ptr(), x1.data_ptr() + 2 * Kp * int(taps[t]),
                                     x2.data_ptr() + 2 * Kp * int(taps[t]), Np, Kp, 0, 0,
                                     grad2d.data_ptr() + 4 * t * Np * Kp, Kp) for t in range(T)]
            arr = (N.TokGradBlock * T)(*blocks)
            _gemm_call(per_row * (K1 + K2), "sei_tokgrad_bf16_blocks", arr, T, K1, K2)
            return
        _gemm_call(per_row * (K1 + K2), "sei_gemm_bf16nt_dw2_taps", g1.data_ptr(), g2.data_ptr(), Np, x1.data_ptr(),
                   x2.data_ptr(), Kp, grad2d.data_ptr(), Np, Kp, K1, K2, 0 if store else 1, T, taps, Np * Kp)
        return
    per_row = _DW["flops_per_row"].get(key) or 2.0 * Np * Kp
    if not store and _queue_dwstream(_DW, grad2d, pairs, per_row, bias):
        return
    assert bias is None
    if len(pairs) == 2:
        (g1, x1), (g2, x2) = pairs
        K1, K2 = g1.shape[0], g2.shape[0]
        if (K1 + K2) % 8 == 0:
            _gemm_call(per_row * (K1 + K2), "sei_gemm_bf16nt_dw2", g1.data_ptr(), g2.data_ptr(), Np,
                       x1.data_ptr(), x2.data_ptr(), Kp, grad2d.data_ptr(), Np, Kp, K1, K2, 0 if store else 1)
            return
    for gy16, x16 in pairs:
        rows = gy16.shape[0]
        if rows % 8 == 0:
            gemm_nt16(gy16, x16, Np, Kp, rows, EPI_NONE if store else EPI_ACCUM, out32=grad2d, a_rmajor=True,
                      b_rmajor=True, flops=per_row * rows)
        else:       # a pixel count the LDS-DMA kernel cannot chunk (e.g. 9 bottleneck pixels x batch 2): staged kernel
            gemm_mixed(gy16, x16, Np, Kp, rows, 1, 0, EPI_NONE if store else EPI_ACCUM, out=grad2d)
        store = False


# Streamed weight gradients of the shallow levels (csrc/dw_stream.hip): the accumulating weight gradients whose shapes
# sei_dwstream_bf16_eligible takes -- conv2 / conv3 of the C = 32 and C = 128 blocks, the 1x1 convolutions between the
# 32-, 128- and 512-channel levels -- are not launched one by one on the tiled GEMM (a 128 x 128 output tile under up to
# 256 K-splits) but collected, operands kept alive, and issued as ONE job table when the backward pass ends (the engine
# callback that flushes parked pairs and deferred folds), or at once outside a backward pass. SEI_NO_DWSTREAM=1: the GEMMs.
DWSTREAM = os.environ.get("SEI_NO_DWSTREAM") != "1"


def _dwstream_ok(grad2d, pairs):
    """The streamed launch serves this weight gradient (shapes, layouts, pixel counts)."""
    if not DWSTREAM or not grad2d.is_cuda or grad2d.dim() != 2 or grad2d.dtype != torch.float32 or len(pairs) > 2:
        return False
    Np, Kp = grad2d.shape
    if grad2d.stride(1) != 1:
        return False
    for gy, x in pairs:
        if not (gy.dtype == x.dtype == torch.bfloat16 and gy.is_contiguous() and x.is_contiguous()
                and gy.dim() == 2 and x.dim() == 2 and gy.shape[1] == Np and x.shape[1] == Kp and gy.shape[0] == x.shape[0]):
            return False
    K1 = pairs[0][0].shape[0]
    K2 = pairs[1][0].shape[0] if len(pairs) == 2 else 0
    return N.lib().sei_dwstream_bf16_eligible(Np, Kp, Np, Kp, K1, K2) != 0


def _queue_dwstream(_DW, grad2d, pairs, per_row, bias=None):
    if not _dwstream_ok(grad2d, pairs):
        return False
    Np, Kp = grad2d.shape
    K1 = pairs[0][0].shape[0]
    K2 = pairs[1][0].shape[0] if len(pairs) == 2 else 0
    (g1, x1), (g2, x2) = pairs[0], pairs[-1]
    job = N.DwStreamJob(g1.data_ptr(), g2.data_ptr(), x1.data_ptr(), x2.data_ptr(), Np, Kp, Np, Kp, grad2d.data_ptr(),
                        grad2d.stride(0), 0, K1, K2, N.ptr(bias))
    _DW["dwjobs"].append((job, (g1, x1, g2, x2, grad2d, bias), per_row * (K1 + K2)))
    if len(_DW["dwjobs"]) == N.DWSTREAM_MAX_JOBS or not _queue_flush(_DW):
        flush_dwstream(_DW)
    return True


def flush_dwstream(_DW):
    jobs, _DW["dwjobs"] = _DW["dwjobs"], []
    if not jobs:
        return
    arr = (N.DwStreamJob * len(jobs))(*[j[0] for j in jobs])
    _gemm_call(sum(j[2] for j in jobs), "sei_dwstream_bf16_jobs", arr, len(jobs))   # (the operands in `jobs` live until here)


def flush_weight_grads(owner=None, _state=None):
    """Issue every parked weight gradient of `owner` (default state when None) on its own (no partner arrived)."""
    _DW = _state if _state is not None else state_of(owner)
    _DW["flush_queued"] = False
    _join_leaves(_DW)                                  # (before the deferred folds below, which read the leaves' partial sums)
    parked, _DW["parked"] = _DW["parked"], {}
    for entry in parked.values():
        if isinstance(entry, list):                    # a parked group (weight_grad16_group)
            _launch_weight_grad_group(_DW, [entry])
        else:
            gy16, x16, grad2d = entry
            _launch_weight_grad(_DW, grad2d, [(gy16, x16)])
    flush_dwstream(_DW)
    flush_folds(_DW)


def weight_grad16(gy16, x16, grad2d, flops_per_row=None, tap_rows=None, bias=None):
    """grad (N', K') += gy^T x, gy16 (M, N') and x16 (M, K') bf16 as stored: both read reduction-major.
    May park the pair until the step's other model call reaches the same weight (see above). flops_per_row: the
    algorithmic FLOPs per reduction row to book for the roofline leg when the operands are zero-padded (2 N' K').
    tap_rows (a ctypes int array of T row offsets): grad2d is (T, N', K') and slice t is gy^T x[rows shifted by
    tap_rows[t]] -- the taps of a 3x3 convolution's weight gradient in one launch (sei_gemm_bf16nt_dw2_taps); x16 is
    the un-shifted window of a grid with guard rows on both sides."""
    key = grad2d.data_ptr()
    _DW = _state_for(key)
    _DW["seen"][key] = grad2d.numel()
    _DW["flops_per_row"][key] = flops_per_row
    if bias is not None:                   # (M,)-shaped float32 gradient: += gy's column sums, with the weight's launch
        _DW["bias_of"][key] = bias
    if tap_rows is not None:
        _DW["taps"][key] = tap_rows
    else:
        _DW["taps"].pop(key, None)          # the address may have belonged to a freed model's tap-major gradient
    split = _DW.get("joint")
    if split is not None and tap_rows is None:
        # one backward pass for the step's two model calls (models/_joint.py): the operands hold both calls' rows -- the
        # two row segments that two backward functions would otherwise have brought one after the other
        M1 = gy16.shape[0] * split[0] // (split[0] + split[1])
        _DW["arrivals"][key] = _DW["arrivals"].get(key, 0) + 2
        _launch_weight_grad(_DW, grad2d, [(gy16[:M1], x16[:M1]), (gy16[M1:], x16[M1:])])
        return
    n = _DW["arrivals"].get(key, 0) + 1
    _DW["arrivals"][key] = n
    partner = _DW["parked"].pop(key, None)
    if partner is not None:
        _launch_weight_grad(_DW, grad2d, [partner[:2], (gy16, x16)])
    elif _DW["merge"] and n < _DW["uses"] and _queue_flush(_DW):
        _DW["parked"][key] = (gy16, x16, grad2d)
    else:
        _launch_weight_grad(_DW, grad2d, [(gy16, x16)])


def weight_grad16_group(items):
    """weight_grad16 for several layers whose operands cover the SAME tokens (the four linear layers of a Swin block):
    items = [(gy16, x16, grad2d, flops_per_row)]. Parked and merged across the step's model calls like single pairs; the
    launch is one token-streamed kernel for all of them (_launch_weight_grad_group)."""
    items = list(items)
    head = items[0][2].data_ptr()
    _DW = _state_for(head)
    for gy16, x16, grad2d, fpr in items:
        key = grad2d.data_ptr()
        _DW["seen"][key] = grad2d.numel()
        _DW["flops_per_row"][key] = fpr
        _DW["taps"].pop(key, None)
        _DW["arrivals"][key] = _DW["arrivals"].get(key, 0) + 1
    n = _DW["arrivals"][head]
    gkey = ("group", head)
    partner = _DW["parked"].pop(gkey, None)
    if partner is not None:
        _launch_weight_grad_group(_DW, [partner, items])
    elif _DW["merge"] and n < _DW["uses"] and _queue_flush(_DW):
        _DW["parked"][gkey] = items
    else:
        _launch_weight_grad_group(_DW, [items])


# ---------------------------------------------------------------------------------------------
# Deferred folds. The reducing kernels of a backward pass (LayerNorm parameter gradients, depthwise weight gradients,
# the LayerNorm epilogue of SwinIR's data-gradient GEMMs) leave per-workgroup partial sums; instead of one ~5-us fold
# launch behind each of them (52 per U-Net step, ~146 per SwinIR step) the partial sums are kept alive and ONE
# sei_fold_many launch per <= 40 destinations adds them up when the backward pass ends (the engine callback that also
# flushes parked weight gradients) -- same slices, same order, launches of one destination one after the other:
# bit-identical gradients. Outside a backward pass (kernel-level tests calling the helpers directly) nothing is
# deferred. SEI_NO_DEFERRED_FOLDS=1 restores the fold per launch.
# ---------------------------------------------------------------------------------------------
DEFERRED_FOLDS = __import__("os").environ.get("SEI_NO_DEFERRED_FOLDS") != "1"


def defer_fold(a, b, c, ncol, split, kind, work, offset, groups):
    """Register `work[offset:]` ([groups][ncol] partial sums, kept alive here) to be folded into a (| b | c) when the
    running backward pass ends. False: not deferred (switched off / no backward pass running) -- the caller folds."""
    if not DEFERRED_FOLDS:
        return False
    _DW = _state_for(a.data_ptr())
    if not _queue_flush(_DW):
        return False
    key = a.data_ptr()
    meta = (N.ptr(b), N.ptr(c), int(ncol), int(split), int(kind))
    job = _DW["folds"].get(key)
    if job is not None and (job["meta"] != meta or len(job["segs"]) == 3):
        flush_folds(_DW)
        job = None
    if job is None:
        if len(_DW["folds"]) == N.FOLD_MAX_JOBS:
            flush_folds(_DW)
        job = _DW["folds"][key] = {"meta": meta, "segs": []}       # (flush_folds starts a new table)
    job["segs"].append((work, work.data_ptr() + 4 * int(offset), int(groups)))
    return True


def flush_folds(_DW):
    jobs, _DW["folds"] = _DW["folds"], {}
    if not jobs:
        return
    arr = (N.FoldJob * len(jobs))()
    for j, (a_ptr, job) in zip(arr, jobs.items()):
        b_ptr, c_ptr, ncol, split, kind = job["meta"]
        j.a, j.b, j.c, j.ncol, j.split, j.kind, j.nseg = a_ptr, b_ptr, c_ptr, ncol, split, kind, len(job["segs"])
        for k, (_, ptr, groups) in enumerate(job["segs"]):
            j.part[k] = ptr
            j.groups[k] = groups
    N.call("sei_fold_many", arr, len(jobs))           # (the partial-sum tensors in `jobs` live until here)


def _queue_flush(_DW):
    """Ask autograd to flush parked pairs when the running backward ends; False outside a backward pass
    (then nothing may be parked: nobody would flush it)."""
    if not _DW["flush_queued"]:
        try:
            torch.autograd.Variable._execution_engine.queue_callback(lambda: flush_weight_grads(_state=_DW))
        except RuntimeError:
            return False
        _DW["flush_queued"] = True
    return True


def use_bf16_blocks(C):
    """A block takes the bf16-storage path when the mode is bf16 and its GEMMs fit the NT kernel."""
    return get_compute_dtype() == "bf16" and C % 32 == 0


# Levels whose ConvBlock MLP runs as the fused kernel. Measured on MI355X at batch 32 (2B pass): C = 32 wins
# (forward 40 vs 56 us for the two GEMMs, backward 49 vs 76 us incl. the cast); C = 128 loses (forward 63 vs 51 us,
# backward 139 vs 70 us): 36,864 pixels are 288 four-wave tiles, one per CU, and nothing overlaps the 16 serial
# weight slices of a tile. SEI_FUSED_MLP=32,128 / SEI_FUSED_MLP= (empty) override for A/B runs.
# Round 4: the 128-channel level has a kernel of its own (csrc/mlp128.hip: nine-wave workgroups of 144 pixels, weights
# through an LDS-DMA ring) for pixel counts that are multiples of 144 -- sei_mlp_fused_eligible says where the fused form is
# the faster one; everything else keeps the GEMMs.
FUSED_MLP_CHANNELS = tuple(int(v) for v in __import__("os").environ.get("SEI_FUSED_MLP", "32,128").split(",") if v)


def fused_mlp_ok(M, C):
    return C in FUSED_MLP_CHANNELS and N.lib().sei_mlp_fused_eligible(M, C) != 0


def _transposed16_cached(p, w16):
    """_transposed16 of a weight's bf16 copy, rebuilt only when the copy changed (one optimizer step = one rebuild, not
    one per backward function: the step's two model calls share it). Every weight that has ever asked is remembered per
    model; when one of them is stale, ALL stale ones are rebuilt by one sei_transpose_bf16_many launch (the 8 matrices of the
    two fused levels: one launch per step instead of 8 of ~9 us each)."""
    plain = getattr(p, "_sei_plain_state", None)
    capturing = torch.cuda.is_current_stream_capturing()
    key = (_generation(plain), p._version, w16.data_ptr())
    hit = getattr(p, "_sei_shadow_t", None)
    if hit is not None and hit[0] == key and hit[2] == capturing:
        return hit[1]
    group = plain.setdefault("transposed", {}) if plain is not None else {}
    group[id(p)] = (p, w16)
    stale = []
    for q, q16 in group.values():
        qkey = (_generation(plain), q._version, q16.data_ptr())
        qhit = getattr(q, "_sei_shadow_t", None)
        if qhit is None or qhit[0] != qkey or qhit[2] != capturing:
            # another weight rides along only while its bf16 copy in the bucket is known to be current (`shadow`'s own
            # test); anything else is rebuilt when its layer asks, after `shadow` has had its look
            if q is p or (plain is not None and plain["gen"] == qkey[0] and plain["version"].get(id(q)) == q._version):
                stale.append((q, q16, qkey))
    if len(stale) == 1 or not p.is_cuda:
        hit = (key, _transposed16(w16), capturing)
        p._sei_shadow_t = hit
        return hit[1]
    for k in range(0, len(stale), N.TRANSPOSE_MAX_JOBS):
        part = stale[k:k + N.TRANSPOSE_MAX_JOBS]
        outs = [torch.empty((q16.shape[1], q16.shape[0]), dtype=torch.bfloat16, device=q16.device) for _, q16, _ in part]
        jobs = (N.TransposeJob * len(part))(*[N.TransposeJob(q16.data_ptr(), o.data_ptr(), q16.shape[0], q16.shape[1])
                                             for (_, q16, _), o in zip(part, outs)])
        N.call("sei_transpose_bf16_many", jobs, len(part))
        for (q, _, qkey), o in zip(part, outs):
            q._sei_shadow_t = (qkey, o, capturing)
    return p._sei_shadow_t[1]


def _transposed16(w16):
    """(R, C) bf16 -> (C, R) bf16 copy (data movement; the fused MLP backward reads both weights transposed)."""
    R, C = w16.shape
    wt = torch.empty((C, R), dtype=torch.bfloat16, device=w16.device)
    N.call("sei_cast_transpose_bf16", w16.data_ptr(), 1, None, wt.data_ptr(), R, C, R, None)
    return wt


class ConvBlockFn16(torch.autograd.Function):
    """ConvBlockFn with bf16 storage of h2 / h4 / gh3 and the direct-to-LDS GEMMs (C % 64 == 0). At the shallow
    levels (C in FUSED_MLP_CHANNELS) conv2 -> GELU -> conv3 + residual is ONE launch whose 4C-wide hidden activation
    never reaches HBM (sei_mlp_fused_fwd); the backward recomputes it (sei_mlp_fused_bwd)."""

    @staticmethod
    def forward(ctx, x, w1, b1, gamma, beta, w2, b2, w3, b3, twice):
        ctx.dtype = get_compute_dtype()
        x = _nhwc(x)
        B, H, W, C = x.shape
        M = B * H * W
        h1, h2, mean, rstd = dwconv7_ln(x, w1, b1, gamma, beta, out16=True)
        w2_16, w3_16 = shadow(w2), shadow(w3)
        ctx.fused = fused_mlp_ok(M, C)
        _tape(ConvBlockFn16, ctx)
        if ctx.fused:
            out = _alloc((M, C), torch.float32, x.device)
            N.call("sei_mlp_fused_fwd", h2.data_ptr(), w2_16.data_ptr(), b2.data_ptr(), w3_16.data_ptr(), b3.data_ptr(),
                   x.data_ptr(), 2.0 if twice else 1.0, out.data_ptr(), M, C)
            if _GEMM_PROFILE is not None:               # counted with the GEMM family (roofline leg): 2 GEMMs of M x 4C x C
                _GEMM_PROFILE.append((4.0 * M * 4 * C * C, "sei_mlp_fused_fwd",
                                      (h2.data_ptr(), w2_16.data_ptr(), b2.data_ptr(), w3_16.data_ptr(), b3.data_ptr(),
                                       x.data_ptr(), 2.0 if twice else 1.0, out.data_ptr(), M, C)))
            ctx.save_for_backward(x, h1, mean, rstd, h2)
            ctx.params = (w1, b1, gamma, beta, w2, b2, w3, b3)
            ctx.twice = twice
            return out.view(B, H, W, C)
        h3 = _alloc((M, 4 * C), torch.float32, x.device)
        h4 = _alloc((M, 4 * C), torch.bfloat16, x.device)
        gemm_nt16(h2, w2_16, M, 4 * C, C, EPI_BIAS_GELU, out32=h3, bias=b2, D2_16=h4)
        out = _alloc((M, C), torch.float32, x.device)
        gemm_nt16(h4, w3_16, M, C, 4 * C, EPI_BIAS_RES, out32=out, bias=b3, R1=x, R2=x if twice else None)
        ctx.save_for_backward(x, h1, mean, rstd, h2, h3, h4)
        ctx.params = (w1, b1, gamma, beta, w2, b2, w3, b3)
        ctx.twice = twice
        return out.view(B, H, W, C)

    @_in_forward_mode
    def backward(ctx, go):
        if ctx.fused:
            return ConvBlockFn16._backward_fused(ctx, go)
        x, h1, mean, rstd, h2, h3, h4 = ctx.saved_tensors
        w1, b1, gamma, beta, w2, b2, w3, b3 = ctx.params
        B, H, W, C = x.shape
        M = B * H * W
        go = go.contiguous()
        go2 = go.view(M, C)
        # Each weight's data gradient goes BEFORE its weight gradient: with the optimizer step fused into the weight-
        # gradient GEMM (set_fused_adam) that launch rewrites the weight's bf16 shadow, which the data gradient reads.
        go16 = cast16(go2, colsum_into_=grad_of(b3))
        gh3 = torch.empty((M, 4 * C), dtype=torch.bfloat16, device=x.device)
        # (go W3) gelu'(h3), with conv2's bias gradient = its column sums riding in the epilogue (no pass over gh3)
        gemm_nt16(go16, shadow(w3), M, 4 * C, C, EPI_MUL_DGELU, out16=gh3, R1=h3, b_rmajor=True, colsum=grad_of(b2))
        weight_grad16(go16, h4, grad_of(w3).view(C, 4 * C))
        gh2 = torch.empty((M, C), dtype=torch.float32, device=x.device)
        gemm_nt16(gh3, shadow(w2), M, C, 4 * C, EPI_NONE, out32=gh2, b_rmajor=True)
        weight_grad16(gh3, h2, grad_of(w2).view(4 * C, C))
        gh1 = layer_norm_bwd(h1.view(M, C), gamma, mean, rstd, gh2, grad_of(gamma), grad_of(beta)).view(B, H, W, C)
        dwconv7_weight_grad(x, gh1, grad_of(w1), grad_of(b1))
        gx = None
        if ctx.needs_input_grad[0]:
            gx = dwconv7(gh1, w1, None, flip=True, res=go, res_scale=2.0 if ctx.twice else 1.0)
        return (gx,) + (None,) * 9


    @staticmethod
    def _backward_fused(ctx, go):
        x, h1, mean, rstd, h2 = ctx.saved_tensors
        w1, b1, gamma, beta, w2, b2, w3, b3 = ctx.params
        B, H, W, C = x.shape
        M = B * H * W
        go = go.contiguous()
        dev = x.device
        w2_16, w3_16 = shadow(w2), shadow(w3)
        gh2 = torch.empty((M, C), dtype=torch.float32, device=dev)
        go16 = torch.empty((M, C), dtype=torch.bfloat16, device=dev)
        h4 = torch.empty((M, 4 * C), dtype=torch.bfloat16, device=dev)
        gh3 = torch.empty((M, 4 * C), dtype=torch.bfloat16, device=dev)
        w3t, w2t = _transposed16_cached(w3, w3_16), _transposed16_cached(w2, w2_16)
        args = (go.data_ptr(), h2.data_ptr(), w2_16.data_ptr(), b2.data_ptr(), w3t.data_ptr(), w2t.data_ptr(),
                gh2.data_ptr(), go16.data_ptr(), h4.data_ptr(), gh3.data_ptr(), M, C)
        N.call("sei_mlp_fused_bwd", *args)
        if _GEMM_PROFILE is not None:                   # booked as the two data-gradient GEMMs it replaces
            _GEMM_PROFILE.append((4.0 * M * 4 * C * C, "sei_mlp_fused_bwd", args))
        if DWSTREAM and N.lib().sei_dwstream_bf16_eligible(C, 4 * C, C, 4 * C, M, 0):
            # the bias gradients = column sums of go16 / gh3 ride in the streamed weight-gradient launch
            weight_grad16(go16, h4, grad_of(w3).view(C, 4 * C), bias=grad_of(b3))
            weight_grad16(gh3, h2, grad_of(w2).view(4 * C, C), bias=grad_of(b2))
        else:                                           # ragged pixel counts: the column-sum kernels (go in float32)
            colsum_into(grad_of(b3), go.view(M, C))
            colsum16_into(grad_of(b2), gh3)
            weight_grad16(go16, h4, grad_of(w3).view(C, 4 * C))
            weight_grad16(gh3, h2, grad_of(w2).view(4 * C, C))
        gh1 = layer_norm_bwd(h1.view(M, C), gamma, mean, rstd, gh2, grad_of(gamma), grad_of(beta)).view(B, H, W, C)
        dwconv7_weight_grad(x, gh1, grad_of(w1), grad_of(b1))
        gx = None
        if ctx.needs_input_grad[0]:
            gx = dwconv7(gh1, w1, None, flip=True, res=go, res_scale=2.0 if ctx.twice else 1.0)
        return (gx,) + (None,) * 9


class DownsampleFn16(torch.autograd.Function):
    """DownsampleFn (resampler before the convolution) with the three GEMMs on bf16 operands."""

    @staticmethod
    def forward(ctx, x, gamma, beta, w, b, rate, with_skip=False):
        ctx.dtype = get_compute_dtype()
        ctx.set_materialize_grads(False)
        x = _nhwc(x)
        B, H, W, C = x.shape
        M, Co = B * H * W, w.shape[0]
        h, mean, rstd = layer_norm(x.view(M, C), gamma, beta)
        fwd, bwd = _mats.resample_matrices("down", H, W, rate, x.device)
        Ho, Wo = fwd[0].shape[0], fwd[1].shape[0]
        u = sepmap2_16(h.view(B, H, W, C), fwd, Ho, Wo, out16=True)
        Mo = B * Ho * Wo
        s = _mats.constant_response("down", H, W, rate, x.device, B)
        u16 = u.view(Mo, C) if u.dtype == torch.bfloat16 else cast16(u.view(Mo, C))     # (straight from the resampler where it can)
        out = _alloc((Mo, Co), torch.float32, x.device)
        gemm_nt16(u16, shadow(w), Mo, Co, C, EPI_BIAS_ROWSCALE, out32=out, bias=b, R1=s)
        ctx.save_for_backward(x, mean, rstd, u16, s)
        ctx.params, ctx.mats_t, ctx.hw = (gamma, beta, w, b), bwd, (H, W, Ho, Wo)
        ctx.with_skip, ctx.rate = with_skip, rate
        _tape(DownsampleFn16, ctx)
        out = out.view(B, Ho, Wo, Co)
        return (out, x) if with_skip else out

    @_in_forward_mode
    def backward(ctx, go, gskip=None):
        x, mean, rstd, u16, s = ctx.saved_tensors
        gamma, beta, w, b = ctx.params
        B, H, W, C = x.shape
        Ho, Wo = ctx.hw[2], ctx.hw[3]
        M, Mo, Co = B * H * W, B * Ho * Wo, w.shape[0]
        go2 = go.contiguous().view(Mo, Co)
        go16 = cast16(go2, colsum_into_=grad_of(b), row_weight=s)           # (the bias gradient from the cast's own pass)
        gu = torch.empty((Mo, C), dtype=torch.float32, device=x.device)
        gemm_nt16(go16, shadow(w), Mo, C, Co, EPI_NONE, out32=gu, b_rmajor=True)
        weight_grad16(go16, u16, grad_of(w).view(Co, C))           # after the data gradient: see ConvBlockFn16.backward
        gh = sepmap2_16(gu.view(B, Ho, Wo, C), ctx.mats_t, H, W).view(M, C)
        res = None if gskip is None else gskip.contiguous().view(M, C)
        gx = layer_norm_bwd(x.view(M, C), gamma, mean, rstd, gh, grad_of(gamma), grad_of(beta), res=res).view(B, H, W, C)
        return (gx if ctx.needs_input_grad[0] else None), None, None, None, None, None, None


class UpsampleFn16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, skip, gamma, beta, w, b, rate):
        ctx.dtype = get_compute_dtype()
        x = _nhwc(x)
        B, H, W, C = x.shape
        Co = w.shape[0]
        fwd, bwd = _mats.resample_matrices("up", H, W, rate, x.device)
        Ho, Wo = fwd[0].shape[0], fwd[1].shape[0]
        u = sepmap2_16(x, fwd, Ho, Wo)
        M = B * Ho * Wo
        h, mean, rstd = layer_norm16(u.view(M, C), gamma, beta)
        w16 = shadow(w)
        out = _alloc((M, Co), torch.float32, x.device)
        _tape(UpsampleFn16, ctx)
        if skip is not None:
            skip = _nhwc(skip)
            if tuple(skip.shape) != (B, Ho, Wo, Co):
                raise ValueError("skip connection shape mismatch")
            gemm_nt16(h, w16, M, Co, C, EPI_BIAS_RES, out32=out, bias=b, R1=skip)
        else:
            gemm_nt16(h, w16, M, Co, C, EPI_BIAS, out32=out, bias=b)
        ctx.save_for_backward(u, mean, rstd, h)
        ctx.params, ctx.mats_t, ctx.in_hw = (gamma, beta, w, b), bwd, (H, W)
        return out.view(B, Ho, Wo, Co)

    @_in_forward_mode
    def backward(ctx, go):
        u, mean, rstd, h = ctx.saved_tensors
        gamma, beta, w, b = ctx.params
        B, Ho, Wo, C = u.shape
        M, Co = B * Ho * Wo, w.shape[0]
        go = go.contiguous()
        go2 = go.view(M, Co)
        go16 = cast16(go2, colsum_into_=grad_of(b))
        gh = torch.empty((M, C), dtype=torch.float32, device=u.device)
        gemm_nt16(go16, shadow(w), M, C, Co, EPI_NONE, out32=gh, b_rmajor=True)
        weight_grad16(go16, h, grad_of(w).view(Co, C))             # after the data gradient: see ConvBlockFn16.backward
        gu = layer_norm_bwd(u.view(M, C), gamma, mean, rstd, gh, grad_of(gamma), grad_of(beta))
        gx = None
        if ctx.needs_input_grad[0]:
            gx = sepmap2_16(gu.view(B, Ho, Wo, C), ctx.mats_t, *ctx.in_hw)
        gskip = go if ctx.needs_input_grad[1] else None
        return gx, gskip, None, None, None, None, None


# ---------------------------------------------------------------------------------------------
# 3x3 convolutions at the ends of the U-Net                 (reference convolutional.py:174-176)
# ---------------------------------------------------------------------------------------------
class Conv3x3Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, res, nchw_in, nchw_out):
        N.check_tensor(x, "conv3x3 input")
        Co, Ci = w.shape[0], w.shape[1]
        if nchw_in:
            B, _, H, W = x.shape
        else:
            B, H, W, _ = x.shape
        y = _alloc((B, Co, H, W) if nchw_out else (B, H, W, Co), torch.float32, x.device)
        _tape(Conv3x3Fn, ctx)
        if res is not None:
            N.check_tensor(res, "conv3x3 residual")
            if res.shape != y.shape:
                raise ValueError("conv3x3: residual shape mismatch")
        N.call("sei_conv3x3_fwd", x.data_ptr(), w.data_ptr(), b.data_ptr(), N.ptr(res), y.data_ptr(), B, H, W,
               Ci, Co, int(nchw_in), int(nchw_out), 0)
        ctx.save_for_backward(x)
        ctx.params, ctx.cfg = (w, b), (B, H, W, Ci, Co, nchw_in, nchw_out)
        return y

    @staticmethod
    def backward(ctx, go):
        (x,) = ctx.saved_tensors
        w, b = ctx.params
        B, H, W, Ci, Co, nchw_in, nchw_out = ctx.cfg
        go = go.contiguous()
        parts = N.lib().sei_conv3x3_bwd_weight_parts_count(B, H, W, Ci, Co, int(nchw_in), int(nchw_out)) if x.is_cuda else 0
        ncol = Co * Ci * 9 + Co
        work = torch.empty(parts * ncol, dtype=torch.float32, device=x.device) if parts else None
        if parts and defer_fold(grad_of(w), grad_of(b), None, ncol, Co * Ci * 9, N.FOLD_SPLIT, work, 0, parts):
            # the end convolutions on the matrix cores, per-workgroup sums folded with the pass's other partial sums
            N.call("sei_conv3x3_bwd_weight_parts", x.data_ptr(), go.data_ptr(), work.data_ptr(), B, H, W, Ci, Co,
                   int(nchw_in), int(nchw_out))
        else:
            leaf_call(grad_of(w), "sei_conv3x3_bwd_weight", x.data_ptr(), go.data_ptr(), grad_of(w).data_ptr(),
                      grad_of(b).data_ptr(), B, H, W, Ci, Co, int(nchw_in), int(nchw_out), keep=(x, go))
        gx = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty_like(x)
            N.call("sei_conv3x3_fwd", go.data_ptr(), w.data_ptr(), None, None, gx.data_ptr(), B, H, W, Co, Ci,
                   int(nchw_out), int(nchw_in), 1)
        gres = go if ctx.needs_input_grad[3] else None
        return gx, None, None, gres, None, None


# ---------------------------------------------------------------------------------------------
# One backward pass for the step's two model calls (models/_joint.py): joint ctx of each layer function, the walk
# ---------------------------------------------------------------------------------------------
class joint_rows:
    """`with joint_rows(backbone, (B1, B2)):` -- weight_grad16 splits its operands' rows B1 : B2 into the two segments
    of the step's two model calls (None: a single call walked alone, nothing to split)."""

    def __init__(self, backbone, batches):
        self.state, self.batches = state_of(backbone), batches

    def __enter__(self):
        global _JOINT_SPLIT
        self.prev = (self.state.get("joint"), _JOINT_SPLIT)
        self.state["joint"] = _JOINT_SPLIT = self.batches
        return self

    def __exit__(self, *exc):
        global _JOINT_SPLIT
        self.state["joint"], _JOINT_SPLIT = self.prev
        return False


def joint_ctx(fn, c1, c2, rec):
    """The ctx of a layer's backward over both calls: saved activations as 3B-row tensors (rec.joint raises where two
    tensors are not the two parts of one arena buffer)."""
    from ._joint import JointCtx
    s1, s2 = c1.saved_tensors, c2.saved_tensors
    if fn is Conv3x3Fn:
        B, H, W, Ci, Co, nchw_in, nchw_out = c1.cfg
        return JointCtx(c1, [rec.joint(s1[0], s2[0])], cfg=(B + c2.cfg[0], H, W, Ci, Co, nchw_in, nchw_out))
    if fn is ConvBlockFn16:
        if c1.fused != c2.fused or c1.twice != c2.twice:
            from ._joint import _NotJoint
            raise _NotJoint()
        return JointCtx(c1, [rec.joint(a, b) for a, b in zip(s1, s2)])
    if fn is DownsampleFn16:
        x = rec.joint(s1[0], s2[0])
        H, W = c1.hw[0], c1.hw[1]
        s = _mats.constant_response("down", H, W, c1.rate, x.device, x.shape[0])
        return JointCtx(c1, [x, rec.joint(s1[1], s2[1]), rec.joint(s1[2], s2[2]), rec.joint(s1[3], s2[3]), s])
    if fn is UpsampleFn16:
        return JointCtx(c1, [rec.joint(a, b) for a, b in zip(s1, s2)])
    from ._joint import _NotJoint
    raise _NotJoint()


def walk_backward(fns, ctxs, go):
    """Play a model call's tape (the layer functions in forward order, with their ctx -- or joint ctx --) backwards from
    the gradient of the model output. The U-Net is a chain plus skip connections nested like brackets: an Upsample's skip
    gradient waits on a stack for the Downsample that handed the skip on."""
    skips = []
    g = go
    for fn, ctx in zip(reversed(fns), reversed(ctxs)):
        if g is None:
            break
        if fn is UpsampleFn16:
            outs = fn.backward(ctx, g)
            g = outs[0]
            skips.append(outs[1])
        elif fn is DownsampleFn16:
            gskip = skips.pop() if ctx.with_skip and skips else None
            g = fn.backward(ctx, g, gskip)[0]
        else:
            g = fn.backward(ctx, g)[0]

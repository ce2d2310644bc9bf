"""Linear image operators backed by the HIP kernels, with autograd.

Every operator here is linear, so one autograd Function serves them all: the backward of `op` is
`op` transposed (and vice versa), which also makes double-backward and `A_adjoint` trivial.
"""
import numpy as np
import torch

import _native as N
from . import _bands


class _Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, op, transpose, out_hw):
        ctx.op, ctx.transpose, ctx.in_hw = op, transpose, tuple(x.shape[-2:])
        return op.run(x, transpose, out_hw)

    @staticmethod
    def backward(ctx, g):
        return _Linear.apply(g.contiguous(), ctx.op, not ctx.transpose, ctx.in_hw), None, None, None


def apply_linear(op, x, transpose=False, out_hw=None):
    """op(x), or op^T(x) when transpose. `out_hw` pins the output size of a transposed resampling
    (the size of the image the forward consumed) where it is not implied by the input size."""
    return _Linear.apply(x, op, transpose, out_hw)


def _as_planes(x):
    if x.dim() < 2:
        raise ValueError("expected an image tensor (..., H, W)")
    N.check_tensor(x, "image")
    H, W = x.shape[-2:]
    planes = x.numel() // (H * W)
    return planes, H, W


class CircularBlurOp:
    """y = k (*) x, circular (reference BlurV2.A, src/physics/blur/__init__.py:205-223).
    Rank-1 kernels run the separable kernel; anything else the dense one."""

    def __init__(self, kernel2d):
        k = kernel2d.detach().to("cpu", torch.float64).reshape(kernel2d.shape[-2], kernel2d.shape[-1]).numpy()
        self.shape = k.shape
        total = k.sum()
        tv, th = k.sum(1), k.sum(0)
        sep = np.outer(tv, th) / total if total != 0 else None
        self.separable = sep is not None and np.abs(sep - k).max() <= 1e-12 * np.abs(k).max()
        if self.separable:
            self._host = (tv / total, th)
        else:
            self._host = (k,)
        self._dev = {}

    def _taps(self, device):
        if device not in self._dev:
            self._dev[device] = tuple(torch.tensor(a, dtype=torch.float32, device=device).contiguous()
                                      for a in self._host)
        return self._dev[device]

    def run(self, x, transpose, out_hw=None):
        planes, H, W = _as_planes(x)
        y = torch.empty_like(x)
        kv, kh = self.shape
        t = self._taps(x.device)
        if self.separable:
            N.call("sei_blur_sep_circ", x.data_ptr(), y.data_ptr(), t[0].data_ptr(), t[1].data_ptr(),
                   kv, kh, planes, H, W, int(transpose))
        else:
            N.call("sei_blur_dense_circ", x.data_ptr(), y.data_ptr(), t[0].data_ptr(), kv, kh, planes,
                   H, W, int(transpose))
        return y


class SeparableResampleOp:
    """y = Wv x Wh^T with Wv, Wh built per input size by `matrix_fn(n_in) -> dense (n_out, n_in)`.
    transpose=True applies Wv^T . Wh (the exact adjoint), taking the *output*-sized image."""

    def __init__(self, matrix_fn, inverse_length=None):
        self.matrix_fn = matrix_fn
        self.inverse_length = inverse_length   # n_out -> n_in, needed to transpose without a forward
        self._cache = {}

    def _axis(self, n_in, device, transpose):
        key = (n_in, device, transpose)
        if key not in self._cache:
            dense = self.matrix_fn(n_in)
            if transpose:
                dense = dense.T
            w, lo, nb, step = _bands.to_band(np.ascontiguousarray(dense))
            self._cache[key] = (torch.from_numpy(w).to(device), torch.from_numpy(lo).to(device), nb, step,
                                dense.shape[0], dense.shape[1])
        return self._cache[key]

    def run(self, x, transpose, out_hw=None):
        planes, H, W = _as_planes(x)
        if transpose:
            if out_hw is not None:
                hi, wi = out_hw
            elif self.inverse_length is not None:
                hi, wi = self.inverse_length(H), self.inverse_length(W)
            else:
                raise RuntimeError("this resampling operator cannot be transposed without its input size")
        else:
            hi, wi = H, W
        wv, lov, nbv, sv, ho, need_h = self._axis(hi, x.device, transpose)
        wh, loh, nbh, sh, wo, need_w = self._axis(wi, x.device, transpose)
        if (need_h, need_w) != (H, W):
            raise ValueError(f"resampling: image is {H}x{W} but the operator expects {need_h}x{need_w}")
        y = torch.empty(x.shape[:-2] + (ho, wo), dtype=x.dtype, device=x.device)
        N.call("sei_resample_sepband", x.data_ptr(), y.data_ptr(), planes, H, W, ho, wo,
               wv.data_ptr(), lov.data_ptr(), nbv, sv, wh.data_ptr(), loh.data_ptr(), nbh, sh)
        return y


_RESIZE_AXES = {}


def resample_to_size(x, out_h, out_w, antialias=True):
    """Bicubic resize of (.., H, W) float32 GPU images to (out_h, out_w) in one launch of the banded separable
    resampler: antialiased (the GroundTruthDataset resize, datasets/ground_truth.py) or plain
    F.interpolate(size=.., mode="bicubic", align_corners=False) (the HOMOGENEOUS_SWINIR measurement upsampling,
    src/datasets/synthetic_dataset.py:43-53). No autograd."""
    planes, H, W = _as_planes(x)
    build = _bands.aa_bicubic_matrix_to_size if antialias else _bands.plain_bicubic_matrix_to_size

    def axis(n_in, n_out):
        key = (n_in, n_out, str(x.device), bool(antialias))
        if key not in _RESIZE_AXES:
            w, lo, nb, step = _bands.to_band(build(n_in, n_out))
            _RESIZE_AXES[key] = (torch.from_numpy(w).to(x.device), torch.from_numpy(lo).to(x.device), nb, step)
        return _RESIZE_AXES[key]

    wv, lov, nbv, sv = axis(H, out_h)
    wh, loh, nbh, sh = axis(W, out_w)
    y = torch.empty(x.shape[:-2] + (out_h, out_w), dtype=x.dtype, device=x.device)
    N.call("sei_resample_sepband", x.data_ptr(), y.data_ptr(), planes, H, W, out_h, out_w,
           wv.data_ptr(), lov.data_ptr(), nbv, sv, wh.data_ptr(), loh.data_ptr(), nbh, sh)
    return y


class _Axpy(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, alpha):
        N.check_tensor(a, "a")
        N.check_tensor(b, "b")
        if a.shape != b.shape:
            raise ValueError("axpy: shape mismatch")
        ctx.alpha = alpha
        out = torch.empty_like(a)
        N.call("sei_axpy", a.data_ptr(), b.data_ptr(), alpha, out.data_ptr(), a.numel())
        return out

    @staticmethod
    def backward(ctx, g):
        gb = g * ctx.alpha if ctx.needs_input_grad[1] else None
        return g, gb, None


def axpy(a, b, alpha):
    """a + alpha*b (measurement noise, SURE probe)."""
    return _Axpy.apply(a, b, float(alpha))

#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the reference itself.

Runs ONLY in the build container (needs /root/reference). The reference's pure-torch modules are
imported read-only (PYTHONDONTWRITEBYTECODE, nothing is written under /root/reference); `deepinv`
is not installed, so an in-memory shell of the three names the physics package imports
(`LinearPhysics`, `GaussianNoise`, `adjoint_function`) is registered first. None of the stub's
arithmetic is exercised by the goldens below except `adjoint_function`, which is a plain
`torch.autograd.functional.vjp` (flagged "vjp" in the fixture names).

What is written is DATA ONLY: seeded inputs and the outputs the reference produced for them
(G12 noise2inverse; G13 the in-tree R2R / EI loss; G14 CropPair on batches; G15 the loss layer end to end). The reference cannot travel to the GPU box; these files can.

    python tools/gen_golden.py            # writes tests/golden/*.npz + manifest json
"""
import importlib
import importlib.util
import json
import os
import sys
import types

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True

import numpy as np
import torch

REF = "/root/reference"
REF_SRC = os.path.join(REF, "src")
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


# --------------------------------------------------------------------------------------------
# deepinv shell (names only; see module docstring)
# --------------------------------------------------------------------------------------------
def _install_deepinv_shell():
    dinv = types.ModuleType("deepinv")
    phys = types.ModuleType("deepinv.physics")
    fwd = types.ModuleType("deepinv.physics.forward")

    class LinearPhysics(torch.nn.Module):
        def __init__(self, **kwargs):
            super().__init__()
            self.noise_model = lambda x: x

        def forward(self, x):
            return self.noise_model(self.A(x))

    class GaussianNoise(torch.nn.Module):
        def __init__(self, sigma=0.1):
            super().__init__()
            self.sigma = sigma

        def forward(self, x):
            return x + torch.randn_like(x) * self.sigma

    def adjoint_function(A, input_size, device="cpu", dtype=torch.float):
        x = torch.ones(input_size, device=device, dtype=dtype)
        (_, vjpfunc) = torch.func.vjp(A, x)

        def adj(y):
            return vjpfunc(y)[0]

        return adj

    phys.LinearPhysics = LinearPhysics
    phys.GaussianNoise = GaussianNoise
    phys.adjoint_function = adjoint_function
    fwd.LinearPhysics = LinearPhysics
    dinv.physics = phys
    phys.forward = fwd
    sys.modules["deepinv"] = dinv
    sys.modules["deepinv.physics"] = phys
    sys.modules["deepinv.physics.forward"] = fwd


def _load_by_path(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _np(t):
    return t.detach().cpu().numpy()


def _save(name, **arrays):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"  {name}.npz  {os.path.getsize(path) / 1024:.1f} KiB  ({len(arrays)} arrays)")


def _rand(shape, seed, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(shape, generator=g, dtype=torch.float32).to(dtype)


def _randn(shape, seed, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g, dtype=torch.float32).to(dtype)


def _fwd_vjp(fn, x, seed):
    """forward + vjp against a seeded cotangent, at x's dtype."""
    x = x.clone().requires_grad_(True)
    y = fn(x)
    ct = _randn(tuple(y.shape), seed, y.dtype)
    (gx,) = torch.autograd.grad(y, x, ct)
    return y.detach(), ct, gx.detach()


def gen_noise2inverse():
    """G12: src/noise2inverse.py (numpy / torch only, imported by path).  Row slices + FFT inverse filter
    (ImageSlices, deblurring), the X:1 training pair of Noise2InverseTransform (numpy seed 3) and the summed
    reconstruction of Noise2InverseModel around a fixed pointwise backbone."""
    os.makedirs(OUT, exist_ok=True)
    n2i = _load_by_path("ref_noise2inverse", os.path.join(REF_SRC, "noise2inverse.py"))
    kernels = _load_by_path("ref_kernels", os.path.join(REF_SRC, "physics", "kernels.py"))
    kernel = kernels.get_kernel("Gaussian_R2")[None, None].to(torch.float32)
    y = _rand((2, 3, 16, 20), 31)
    arrs = {"y": _np(y), "kernel": _np(kernel)}
    slicer = n2i.ImageSlices(num_splits=4, task="deblurring", physics_filter=kernel, degradation_inverse_fn=None)
    for j, t in enumerate(slicer(y)):
        arrs[f"slice{j}"] = _np(t)
    tr = n2i.Noise2InverseTransform(task="deblurring", physics_filter=kernel, degradation_inverse_fn=None)
    np.random.seed(3)
    tgt, inp = tr(None, y)
    arrs["pair.tgt"], arrs["pair.inp"] = _np(tgt), _np(inp)
    np.random.seed(3)
    arrs["pair.index"] = np.asarray(np.random.randint(0, 4))
    backbone = lambda v: 0.25 * v + 0.1 * v * v
    model = n2i.Noise2InverseModel(backbone=backbone, task="deblurring", physics_filter=kernel,
                                   degradation_inverse_fn=None)
    arrs["model.x_hat"] = _np(model(y))
    # the non-deblurring branch: the caller's pseudo-inverse stands in for the FFT filter
    up = n2i.ImageSlices(num_splits=4, task="sr", physics_filter=None, degradation_inverse_fn=lambda v: 2.0 * v)
    arrs["sr.slice1"] = _np(up(y)[1])
    _save("g12_noise2inverse", **arrs)


def gen_r2r():
    """G13: src/losses/r2r.py (in-tree; imported by path).  `R2RLoss` (:7-23) and `R2REILoss` (:26-57), whose
    `ei_loss` is the reference authors' own near-copy of deepinv's EILoss ("slightly modified for consistent input
    noise"): it pins, by the reference's code, T under no_grad, y2 = A(x2), the noise placement and the metric.
    The file's only third-party symbol is `deepinv.loss.metric.mse` (:4): a one-class shell (mean squared error over
    all elements, flagged "mse shell") stands in for it.  Everything else is the reference: the h8s3 U-Net
    (models/convolutional.py), BlurV2 / Downsampling (physics), ScalingTransform("padded", antialias=False)
    (transforms.py, unmodified: its rand draws come from manual_seed(5)).  The backbone is called as
    `model(y, physics)`; src/models/__init__.py:148-149 (`Model.forward(x, *args)`, not importable: deepinv at the
    top) drops the extras, and so does the wrapper here.  torch.randn_like is patched to hand out the three saved
    normal draws in call order (pert, epsilon1, epsilon2)."""
    os.makedirs(OUT, exist_ok=True)
    _install_deepinv_shell()
    loss_mod = types.ModuleType("deepinv.loss")
    metric_mod = types.ModuleType("deepinv.loss.metric")

    class mse(torch.nn.Module):                      # "mse shell"
        def forward(self, x, y):
            return torch.nn.functional.mse_loss(x, y)

    metric_mod.mse = mse
    loss_mod.metric = metric_mod
    sys.modules["deepinv"].loss = loss_mod
    sys.modules["deepinv.loss"] = loss_mod
    sys.modules["deepinv.loss.metric"] = metric_mod
    if REF_SRC not in sys.path:
        sys.path.insert(0, REF_SRC)
    physics = importlib.import_module("physics")
    transforms = importlib.import_module("transforms")
    conv = _load_by_path("ref_convolutional", os.path.join(REF_SRC, "models", "convolutional.py"))
    r2r = _load_by_path("ref_r2r", os.path.join(REF_SRC, "losses", "r2r.py"))
    torch.set_num_threads(8)
    sigma = 5 / 255
    print("G13 r2r / ei_loss")
    for tag, rate in [("deblur", 1), ("sr2", 2)]:
        torch.manual_seed(0)
        m32 = conv.ConvolutionalModel(in_channels=3, upsampling_rate=rate, residual=True, inner_residual=True,
                                      num_conv_blocks=1, hidden_channels=8, inout_convs=True, scales=3)
        with torch.no_grad():
            for n, p in m32.named_parameters():
                if ".ln." in n:
                    p.add_(0.05 * torch.randn_like(p))
        if rate == 1:
            phys_op = physics.BlurV2(kernel=physics.BlurKernel("Gaussian_R2").to_tensor("cpu"))
            S = 48
        else:
            phys_op = physics.Downsampling(rate=rate, antialias=True)
            S = 24
        B = 2
        y32 = _rand((B, 3, S, S), 130)
        n0, n1, n2 = (_randn((B, 3, S, S), 131 + i) for i in range(3))
        arrs = {"y": _np(y32), "n0": _np(n0), "n1": _np(n1), "n2": _np(n2)}
        for k2, v in m32.state_dict().items():
            arrs[f"sd.{k2}"] = _np(v)
        import copy
        for dt, dn in [(torch.float64, "f64"), (torch.float32, "f32")]:
            m = copy.deepcopy(m32).to(dt)
            y = y32.to(dt)
            net = lambda v, *ignored, m=m: m(v)
            for no_grad in (True, False):
                ng = "" if no_grad else "grad_through_T."
                seen = {}
                T = transforms.ScalingTransform(kind="padded", antialias=False)

                class Recorder(torch.nn.Module):        # passes x through the reference transform, keeps in/out
                    def forward(self, x):
                        out = T(x)
                        seen["x1"], seen["x2"] = x.detach().clone(), out.detach().clone()
                        return out

                lf = r2r.R2REILoss(transform=Recorder(), sigma=sigma, no_grad=no_grad, metric=None)
                draws = iter([n0.to(dt), n1.to(dt), n2.to(dt)])
                saved = torch.randn_like
                torch.randn_like = lambda t, **kw: next(draws).clone()
                try:
                    torch.manual_seed(5)                   # ScalingTransform's rand(B), rand(B, 2)
                    l_r2r = lf.r2r_loss(y=y, physics=phys_op, model=net)
                    l_ei = lf.ei_loss(y=y, physics=phys_op, model=net)
                finally:
                    torch.randn_like = saved
                total = l_r2r + l_ei
                # the forward() of the class is the same sum: check it on the same draws
                draws = iter([n0.to(dt), n1.to(dt), n2.to(dt)])
                torch.randn_like = lambda t, **kw: next(draws).clone()
                try:
                    torch.manual_seed(5)
                    assert torch.equal(lf(y=y, physics=phys_op, model=net), total)
                finally:
                    torch.randn_like = saved
                torch.manual_seed(5)
                r, c = transforms.sample_downsampling_parameters(B, "cpu", dt, [0.75, 0.5])
                m.zero_grad()
                total.backward()
                p = f"{dn}.{ng}"
                arrs[p + "rate"], arrs[p + "center"] = _np(r), _np(c.view(-1, 2))
                arrs[p + "loss_r2r"], arrs[p + "loss_ei"], arrs[p + "loss"] = _np(l_r2r), _np(l_ei), _np(total)
                if no_grad:
                    arrs[p + "x1"], arrs[p + "x2"] = _np(seen["x1"]), _np(seen["x2"])
                # full gradients for the float64 default (no_grad) run; per-tensor norms for the other three
                for k2, q in m.named_parameters():
                    if dn == "f64" and no_grad:
                        arrs[p + f"grad.{k2}"] = _np(q.grad).astype(np.float32)
                    else:
                        arrs[p + f"gradnorm.{k2}"] = _np(q.grad.norm())
        _save(f"g13_r2r_{tag}", **arrs)


def gen_crop():
    """G14: src/crop.py (imported by path) -- CropPair / MinSizePadding on BATCHES, i.e. the batched-crop quirk (SURVEY a8:
    MinSizePadding reads x.shape[1], x.shape[2] as height and width, which on a 4-D batch are C and H). Its only third-party
    symbols are torchvision.transforms.functional.pad and .crop: a two-function shell stands in (flagged "TF shell": pad =
    constant F.pad in torchvision's [left, top, right, bottom] order, crop = the in-bounds slice -- every crop the reference
    makes here is in bounds). The two CPU randint draws come from manual_seed(k), k = 0..5."""
    os.makedirs(OUT, exist_ok=True)
    tv = types.ModuleType("torchvision")
    tvt = types.ModuleType("torchvision.transforms")
    tf = types.ModuleType("torchvision.transforms.functional")

    def pad(img, padding, fill=0, padding_mode="constant"):          # "TF shell"
        assert padding_mode == "constant" and len(padding) == 4
        left, top, right, bottom = padding
        return torch.nn.functional.pad(img, (left, right, top, bottom), value=fill)

    def crop(img, top, left, height, width):                         # "TF shell"
        assert top >= 0 and left >= 0 and top + height <= img.shape[-2] and left + width <= img.shape[-1]
        return img[..., top:top + height, left:left + width]

    tf.pad, tf.crop = pad, crop
    tvt.functional = tf
    tv.transforms = tvt
    sys.modules.update({"torchvision": tv, "torchvision.transforms": tvt, "torchvision.transforms.functional": tf})
    ref_crop = _load_by_path("ref_crop", os.path.join(REF_SRC, "crop.py"))
    print("G14 crop")
    arrs = {}
    # (values = 1 + the element's flat index: a crop then shows which element went where, zeros are padding; small crop
    # sizes keep the fixture small -- the quirk scales with them: size - 3 zero rows are appended to a 3-channel batch)
    for tag, xs, ys, size, ratio in [("deblur", (2, 3, 20, 17), (2, 3, 20, 17), 12, 1), ("deblur_small", (2, 3, 10, 14), (2, 3, 10, 14), 12, 1),
                                     ("sr2", (2, 3, 24, 24), (2, 3, 12, 12), 12, 2), ("sr4", (1, 3, 64, 64), (1, 3, 16, 16), 16, 4),
                                     ("item3d", (3, 30, 26), (3, 15, 13), 12, 2), ("deblur48", (1, 3, 64, 56), (1, 3, 64, 56), 48, 1)]:
        x = (1 + torch.arange(int(np.prod(xs)), dtype=torch.float32)).view(xs)
        y = (1 + torch.arange(int(np.prod(ys)), dtype=torch.float32)).view(ys)
        arrs[f"{tag}.xshape"], arrs[f"{tag}.yshape"] = np.array(xs), np.array(ys)
        arrs[f"{tag}.cfg"] = np.array([size, ratio])
        for k in range(6 if size < 48 else 2):
            torch.manual_seed(k)
            xc, yc = ref_crop.CropPair("random", size)(x, y, xy_size_ratio=ratio)
            arrs[f"{tag}.seed{k}.xc"], arrs[f"{tag}.seed{k}.yc"] = _np(xc).astype(np.int32), _np(yc).astype(np.int32)
        xc, yc = ref_crop.CropPair("center", size)(x, y, xy_size_ratio=ratio)
        arrs[f"{tag}.center.xc"], arrs[f"{tag}.center.yc"] = _np(xc).astype(np.int32), _np(yc).astype(np.int32)
    _save("g14_crop", **arrs)


def gen_loss_glue():
    """G15: the reference's OWN loss layer end to end -- src/losses/__init__.py (get_loss :210-266 with demo/train.py's
    default flags :35-61, Loss.forward :203-207, ProposedLoss :67-142) around src/crop.py, src/losses/sure.py,
    src/transforms.py, src/physics (get_physics) and the h8s3 U-Net of src/models/convolutional.py, on the CPU generator
    (manual_seed(k): the crop's two randint draws, SURE's randn, the transform's rand(B) / rand(B, 2), the measurement
    noise's randn_like, in the order the reference's code consumes them).
    Shells (deepinv / torchvision are absent), all flagged: `deepinv.loss.EILoss` and `SupLoss` as their documented
    v0.2.0 forward [recollection; the EI arithmetic itself is pinned by G13, the reference's in-tree near-copy],
    `deepinv.loss.metric.mse`, `deepinv.physics.GaussianNoise` (x + sigma randn_like(x)), `LinearPhysics.__call__`
    (noise_model(A(x))), placeholders for `deepinv.transform.Rotate` / `Shift` (not exercised), torchvision's TF.pad /
    TF.crop (G14's shell). What is pinned is the reference's glue: the margin rule of get_loss, crop-then-method,
    the loss list and its keyword protocol, the summation, and the ORDER in which the generator is consumed."""
    import argparse
    os.makedirs(OUT, exist_ok=True)
    _install_deepinv_shell()
    dinv = sys.modules["deepinv"]
    loss_mod, metric_mod, tr_mod = (types.ModuleType(n) for n in ("deepinv.loss", "deepinv.loss.metric", "deepinv.transform"))

    class mse(torch.nn.Module):                      # "mse shell"
        def forward(self, x, y):
            return torch.nn.functional.mse_loss(x, y)

    class SupLoss(torch.nn.Module):                  # "SupLoss shell"
        def __init__(self, metric=None):
            super().__init__()
            self.metric = metric if metric is not None else torch.nn.MSELoss()

        def forward(self, x_net, x, **kwargs):
            return self.metric(x_net, x)

    class EILoss(torch.nn.Module):                   # "EILoss shell" (deepinv v0.2.0's forward, [recollection]; see G13)
        def __init__(self, transform, metric=None, apply_noise=True, weight=1.0, no_grad=False):
            super().__init__()
            self.T, self.metric, self.noise, self.weight, self.no_grad = transform, metric, apply_noise, weight, no_grad

        def forward(self, x_net, physics, model, **kwargs):
            if self.no_grad:
                with torch.no_grad():
                    x2 = self.T(x_net)
            else:
                x2 = self.T(x_net)
            y = physics(x2) if self.noise else physics.A(x2)
            x3 = model(y, physics)
            return self.weight * self.metric(x3, x2)

    class _Unused(torch.nn.Module):
        def forward(self, x):
            raise RuntimeError("placeholder: not exercised by G15")

    metric_mod.mse = mse
    loss_mod.metric, loss_mod.SupLoss, loss_mod.EILoss = metric_mod, SupLoss, EILoss
    tr_mod.Rotate, tr_mod.Shift = _Unused, _Unused
    dinv.loss, dinv.transform = loss_mod, tr_mod
    sys.modules.update({"deepinv.loss": loss_mod, "deepinv.loss.metric": metric_mod, "deepinv.transform": tr_mod})
    tv, tvt, tf = (types.ModuleType(n) for n in ("torchvision", "torchvision.transforms", "torchvision.transforms.functional"))

    def pad(img, padding, fill=0, padding_mode="constant"):          # "TF shell" (as G14)
        left, top, right, bottom = padding
        return torch.nn.functional.pad(img, (left, right, top, bottom), value=fill)

    def crop(img, top, left, height, width):
        assert top >= 0 and left >= 0 and top + height <= img.shape[-2] and left + width <= img.shape[-1]
        return img[..., top:top + height, left:left + width]

    tf.pad, tf.crop = pad, crop
    tvt.functional, tv.transforms = tf, tvt
    sys.modules.update({"torchvision": tv, "torchvision.transforms": tvt, "torchvision.transforms.functional": tf})
    if REF_SRC not in sys.path:
        sys.path.insert(0, REF_SRC)
    physics = importlib.import_module("physics")
    losses = importlib.import_module("losses")
    transforms = importlib.import_module("transforms")
    conv = _load_by_path("ref_convolutional", os.path.join(REF_SRC, "models", "convolutional.py"))
    torch.set_num_threads(8)
    print("G15 loss glue")
    for tag, task, rate in [("deblur", "deblurring", None), ("sr2", "sr", 2)]:
        # src/settings.py's and demo/train.py's defaults for the flags get_physics / get_loss read
        args = argparse.Namespace(task=task, kernel="Gaussian_R2" if task == "deblurring" else None, sr_factor=rate, noise_level=5,
                                  physics_v2=True, physics_true_adjoint=False, method="proposed", partial_sure=True,
                                  sure_margin=None, partial_sure_sr=False, sure_cropped_div=True, sure_averaged_cst=None,
                                  Loss__crop_training_pairs=True, Loss__crop_size=48, ProposedLoss__stop_gradient=True,
                                  ProposedLoss__sure_alternative=None, ProposedLoss__alpha_tradeoff=1.0,
                                  ProposedLoss__transforms="Scaling_Transforms", ScalingTransform__kind="padded",
                                  ScalingTransform__antialias=False)
        phys = physics.get_physics(args, "cpu")
        lf = losses.get_loss(args, phys)
        inner = lf.loss
        assert type(inner).__name__ == "ProposedLoss" and [type(f).__name__ for f in inner.loss_fns] == ["SureGaussianLoss", "EILoss"]
        # the other methods behind the same surface, through the reference's own SURELoss / SupervisedLoss
        others = {}
        for method in ("sure", "supervised"):
            a2 = argparse.Namespace(**vars(args))
            a2.method = method
            others[method] = losses.get_loss(a2, phys)
        torch.manual_seed(0)
        m32 = conv.ConvolutionalModel(in_channels=3, upsampling_rate=rate or 1, residual=True, inner_residual=True,
                                      num_conv_blocks=1, hidden_channels=8, inout_convs=True, scales=3)
        with torch.no_grad():
            for n, p in m32.named_parameters():
                if ".ln." in n:
                    p.add_(0.05 * torch.randn_like(p))
        r = rate or 1
        B, S = 2, 48
        x32 = _rand((B, 3, 64 * r, 64 * r), 150)
        y32 = _rand((B, 3, 64, 64), 151)
        arrs = {"x": _np(x32), "y": _np(y32), "sure_margin": np.array(inner.loss_fns[0].margin),
                "cropped_div": np.array(int(inner.loss_fns[0].cropped_div)), "xy_size_ratio": np.array(lf.xy_size_ratio)}
        for k2, v in m32.state_dict().items():
            arrs[f"sd.{k2}"] = _np(v)
        import copy
        for dt, dn in [(torch.float64, "f64"), (torch.float32, "f32")]:
            m = copy.deepcopy(m32).to(dt)

            class Net(torch.nn.Module):                  # src/models/__init__.py:148-149: Model.forward(x, *args) drops the extras
                def forward(self, v, *ignored):
                    return m(v)

            for seed in (0, 1):
                torch.manual_seed(seed)
                val = lf(x=x32.to(dt), y=y32.to(dt), model=Net())
                m.zero_grad()
                val.backward()
                p = f"{dn}.seed{seed}."
                arrs[p + "loss"] = _np(val)
                for k2, q in m.named_parameters():
                    if dn == "f64" and seed == 0:
                        arrs[p + f"grad.{k2}"] = _np(q.grad).astype(np.float32)
                    else:
                        arrs[p + f"gradnorm.{k2}"] = _np(q.grad.norm())
                for method, lf2 in others.items():     # (sure: crop, then its randn; supervised: crop only)
                    torch.manual_seed(seed)
                    v2 = lf2(x=x32.to(dt), y=y32.to(dt), model=Net())
                    m.zero_grad()
                    v2.backward()
                    arrs[p + f"{method}.loss"] = _np(v2)
                    arrs[p + f"{method}.gradnorm"] = _np(torch.stack([q.grad.norm() for q in m.parameters()]))
                # the numbers that seed hands out, in the reference's order (for a path that must have them injected)
                torch.manual_seed(seed)
                hy = y32.shape[-2] + max(0, S - 3)                    # MinSizePadding on a batch: size - C rows appended
                wy = y32.shape[-1] + max(0, S - y32.shape[-2])
                i = torch.randint(0, hy - S + 1, size=(1,)).item()
                j = torch.randint(0, wy - S + 1, size=(1,)).item()
                mg = int(inner.loss_fns[0].margin)
                b = torch.randn(B, 3, S - 2 * mg, S - 2 * mg, dtype=dt) if mg else torch.randn(B, 3, S, S, dtype=dt)
                rr, cc = transforms.sample_downsampling_parameters(B, "cpu", dt, [0.75, 0.5])
                nn_ = torch.randn(B, 3, S, S, dtype=dt)
                arrs[p + "ij"] = np.array([i, j])
                arrs[p + "b"], arrs[p + "rate"], arrs[p + "center"], arrs[p + "n"] = _np(b), _np(rr), _np(cc.view(-1, 2)), _np(nn_)
        _save(f"g15_loss_glue_{tag}", **arrs)


def main():
    os.makedirs(OUT, exist_ok=True)
    _install_deepinv_shell()
    sys.path.insert(0, REF_SRC)
    physics = importlib.import_module("physics")
    transforms = importlib.import_module("transforms")
    scheduler = importlib.import_module("scheduler")
    kernels = importlib.import_module("physics.kernels")
    sure = _load_by_path("ref_sure", os.path.join(REF_SRC, "losses", "sure.py"))
    conv = _load_by_path("ref_convolutional", os.path.join(REF_SRC, "models", "convolutional.py"))
    torch.set_num_threads(8)

    # ---------------- G1: kernel table --------------------------------------------------------
    print("G1 kernels")
    _save("g1_kernels", **{n: _np(kernels.get_kernel(n)) for n in kernels._table})

    # ---------------- G2: BlurV2 (FFT circular blur) fwd + vjp; legacy Blur --------------------
    print("G2 blur")
    arrs = {}
    for kname in ["Gaussian_R2", "Box_R3", "Gaussian_R1"]:
        k = physics.BlurKernel(kname).to_tensor("cpu")
        op = physics.BlurV2(kernel=k)
        for tag, shape, seed in [("sq", (2, 3, 48, 48), 11), ("rect", (1, 3, 40, 56), 12),
                                 ("tiny", (1, 1, 16, 20), 13)]:
            for dt, dn in [(torch.float32, "f32"), (torch.float64, "f64")]:
                x = _rand(shape, seed, dt)
                y, ct, gx = _fwd_vjp(op.A, x, seed + 100)
                p = f"{kname}.{tag}.{dn}."
                if dn == "f32":
                    arrs[p + "x"] = _np(x)
                    arrs[p + "ct"] = _np(ct)
                arrs[p + "y"] = _np(y)
                arrs[p + "gx"] = _np(gx)
        # legacy operator (--no-physics_v2): same math by a different route
        x = _rand((2, 3, 24, 24), 14)
        leg = physics.Blur(filter=k.float(), padding="circular", device="cpu")
        arrs[f"{kname}.legacy.x"] = _np(x)
        arrs[f"{kname}.legacy.y"] = _np(leg.A(x))
        arrs[f"{kname}.legacy.adj"] = _np(leg.A_adjoint(x))
    _save("g2_blur", **arrs)

    # ---------------- G3: Downsampling (AA bicubic) fwd + vjp + deprecated adjoint ------------
    print("G3 downsampling")
    arrs = {}
    for rate in [2, 3, 4]:
        op = physics.Downsampling(rate=rate, antialias=True)
        for tag, shape, seed in [("sq", (2, 3, 96, 96), 21), ("rect", (1, 3, 48, 72), 22)]:
            for dt, dn in [(torch.float32, "f32"), (torch.float64, "f64")]:
                x = _rand(shape, seed, dt)
                y, ct, gx = _fwd_vjp(op.A, x, seed + 100)
                p = f"r{rate}.{tag}.{dn}."
                if dn == "f32":
                    arrs[p + "x"] = _np(x)
                    arrs[p + "ct"] = _np(ct)
                arrs[p + "y"] = _np(y)
                arrs[p + "gx"] = _np(gx)
        # A_adjoint default (plain bicubic upsample, a=-0.75) and true adjoint
        yy = _rand((1, 3, 12, 16), 23)
        arrs[f"r{rate}.adj.y"] = _np(yy)
        arrs[f"r{rate}.adj.plain"] = _np(op.A_adjoint(yy))
        op_t = physics.Downsampling(rate=rate, antialias=True, true_adjoint=True)
        arrs[f"r{rate}.adj.true"] = _np(op_t.A_adjoint(yy))
    _save("g3_downsampling", **arrs)

    # ---------------- G4/G5: scale transform ---------------------------------------------------
    print("G4/G5 scale transform")
    arrs = {}
    # RNG draw table (a12): manual_seed(k) -> rates, centers, for B=4, CPU f32
    for k in range(4):
        torch.manual_seed(k)
        r, c = transforms.sample_downsampling_parameters(4, "cpu", torch.float32, [0.75, 0.5])
        arrs[f"draw.seed{k}.rate"] = _np(r)
        arrs[f"draw.seed{k}.center"] = _np(c)
    cases = [
        ("b4s48", (4, 3, 48, 48), 31, [0.75, 0.5, 0.75, 0.5],
         [[-0.3852, 0.2682], [-0.0198, 0.7929], [0.95, -0.97], [-1.0, 1.0]]),
        ("b2s96", (2, 3, 96, 96), 32, [0.5, 0.75], [[0.25, -0.6], [0.0, 0.0]]),
        ("b1s20", (1, 2, 20, 20), 33, [0.5], [[0.999, 0.999]]),
    ]
    for tag, shape, seed, rates, centers in cases:
        for dt, dn in [(torch.float32, "f32"), (torch.float64, "f64")]:
            x = _rand(shape, seed, dt)
            # f32-rounded parameters in both precisions, so the f64 run sees the same inputs
            r = torch.tensor(rates, dtype=torch.float32).to(dt)
            c = torch.tensor(centers, dtype=torch.float32).to(dt).view(len(rates), 1, 1, 2)
            fn = lambda t: transforms.padded_downsampling_transform(
                t, downsampling_rate=r, center=c, mode="bicubic", padding_mode="reflection",
                antialiased=False)
            y, ct, gx = _fwd_vjp(fn, x, seed + 100)
            p = f"{tag}.{dn}."
            if dn == "f32":
                arrs[p + "x"] = _np(x)
                arrs[p + "rate"] = _np(r)
                arrs[p + "center"] = _np(c.view(-1, 2))
                arrs[p + "ct"] = _np(ct)
            arrs[p + "y"] = _np(y)
            arrs[p + "gx"] = _np(gx)
    # antialiased variant (a15): all rates equal so torch.stack succeeds
    x = _rand((2, 3, 48, 48), 34)
    r = torch.tensor([0.5, 0.5])
    c = torch.tensor([[0.1, -0.2], [0.4, 0.3]]).view(2, 1, 1, 2)
    arrs["aa.x"] = _np(x)
    arrs["aa.rate"] = _np(r)
    arrs["aa.center"] = _np(c.view(-1, 2))
    arrs["aa.y"] = _np(transforms.padded_downsampling_transform(
        x, downsampling_rate=r, center=c, mode="bicubic", padding_mode="reflection", antialiased=True))
    # G5: the grid itself
    g = transforms.get_downsampling_grid((2, 3, 6, 6), torch.tensor([0.75, 0.5]).double(),
                                         torch.tensor([[0.2, -0.4], [0.0, 0.5]]).double().view(2, 1, 1, 2),
                                         torch.float64, "cpu")
    arrs["grid.b2s6"] = _np(g)
    # normal kind (a16): deterministic given the rate
    x = _rand((2, 3, 24, 24), 35)
    arrs["normal.x"] = _np(x)
    for rr in [0.75, 0.5]:
        for aa in [False, True]:
            arrs[f"normal.r{rr}.aa{int(aa)}"] = _np(
                transforms.normal_downsampling_transform(x, rr, "bicubic", aa))
    _save("g4_scale_transform", **arrs)

    # ---------------- G6: SURE with a linear stand-in model and injected randn ----------------
    print("G6 sure")
    arrs = {}
    k = physics.BlurKernel("Gaussian_R2").to_tensor("cpu")
    blur = physics.BlurV2(kernel=k)
    y = _rand((2, 3, 48, 48), 41)
    w = torch.tensor(0.9, requires_grad=True)
    model = lambda t: w * t + 0.05 * t * t
    for margin in [0, 6]:
        bfull = _randn((2, 3, 48 - 2 * margin, 48 - 2 * margin), 42 + margin)
        saved = (torch.randn, torch.randn_like)
        torch.randn = lambda *a, **kw: bfull.clone()
        torch.randn_like = lambda t, **kw: bfull.clone()
        try:
            for cst in [False, True]:
                lf = sure.SureGaussianLoss(sigma=5 / 255, margin=margin, cropped_div=True,
                                           averaged_cst=cst)
                x_net = model(y)
                val = lf(y=y, x_net=x_net, physics=blur, model=model)
                (gw,) = torch.autograd.grad(val, w)
                arrs[f"m{margin}.cst{int(cst)}.loss"] = _np(val)
                arrs[f"m{margin}.cst{int(cst)}.gw"] = _np(gw)
        finally:
            torch.randn, torch.randn_like = saved
        arrs[f"m{margin}.b"] = _np(bfull)
    arrs["y"] = _np(y)
    _save("g6_sure", **arrs)

    # ---------------- G7: the U-Net ------------------------------------------------------------
    print("G7 unet")

    def sd_np(m):
        return {k: _np(v) for k, v in m.state_dict().items()}

    def grads_np(m):
        return {k: _np(p.grad) for k, p in m.named_parameters()}

    # per-layer goldens
    arrs = {}
    torch.manual_seed(7)
    for H in [48, 24, 12, 6, 20]:
        x = _rand((2, 3, H, H), 50 + H)
        arrs[f"ideal_down2.{H}.x"] = _np(x)
        arrs[f"ideal_down2.{H}.y"] = _np(conv.IdealDownsample(2)(x))
        arrs[f"ideal_down2.{H}.y64"] = _np(conv.IdealDownsample(2)(x.double()))
    x = _rand((1, 2, 16, 32), 59)
    arrs["ideal_down2.rect.x"] = _np(x)
    arrs["ideal_down2.rect.y"] = _np(conv.IdealDownsample(2)(x))
    for H in [3, 6, 12, 24, 10]:
        x = _rand((2, 3, H, H), 60 + H)
        arrs[f"ideal_up2.{H}.x"] = _np(x)
        arrs[f"ideal_up2.{H}.y"] = _np(conv.IdealUpsample(2)(x))
        arrs[f"ideal_up2.{H}.y64"] = _np(conv.IdealUpsample(2)(x.double()))
    for rate, H in [(4, 12), (4, 48), (3, 18)]:
        x = _rand((1, 3, H, H), 70 + H)
        arrs[f"ideal_up{rate}.{H}.x"] = _np(x)
        arrs[f"ideal_up{rate}.{H}.y"] = _np(conv.IdealUpsample(rate)(x))
    # reference quirk: an odd rate with a width = 0 (mod 4) cannot embed its spectrum and raises
    try:
        conv.IdealUpsample(3)(_rand((1, 3, 16, 16), 1))
        arrs["ideal_up3.16.raises"] = np.array(0)
    except RuntimeError:
        arrs["ideal_up3.16.raises"] = np.array(1)
    x = _rand((1, 2, 16, 32), 79)
    arrs["ideal_up2.rect.x"] = _np(x)
    arrs["ideal_up2.rect.y"] = _np(conv.IdealUpsample(2)(x))
    _save("g7_ideal_resamplers", **arrs)

    arrs = {}
    torch.manual_seed(8)
    for name, mod, shape in [
        ("convblock16", conv.ConvBlock(16), (2, 16, 12, 12)),
        ("convblock8", conv.ConvBlock(8), (1, 8, 5, 7)),
        ("layernorm12", conv.LayerNorm(12, eps=1e-6), (2, 12, 6, 6)),
        ("downsample8", conv.Downsample(in_channels=8), (2, 8, 12, 12)),
        ("upsample32", conv.Upsample(in_channels=32, rate=2), (2, 32, 6, 6)),
        ("upsample3x4", conv.Upsample(in_channels=3, out_channels=3, rate=4), (1, 3, 12, 12)),
    ]:
        # perturb the norm affine params so that they are not the identity
        with torch.no_grad():
            for n, p in mod.named_parameters():
                if ".ln." in n or n.startswith("ln."):
                    p.add_(0.1 * torch.randn_like(p))
        x = _randn(shape, 80 + shape[1]).requires_grad_(True)
        y = mod(x)
        ct = _randn(tuple(y.shape), 90 + shape[1])
        mod.zero_grad()
        (gx,) = torch.autograd.grad(y, x, ct, retain_graph=True)
        y.backward(ct)
        arrs[f"{name}.x"] = _np(x)
        arrs[f"{name}.y"] = _np(y)
        arrs[f"{name}.ct"] = _np(ct)
        arrs[f"{name}.gx"] = _np(gx)
        for k2, v in sd_np(mod).items():
            arrs[f"{name}.sd.{k2}"] = v
        for k2, v in grads_np(mod).items():
            arrs[f"{name}.grad.{k2}"] = v
    _save("g7_layers", **arrs)

    # whole-model goldens, small configs
    for tag, kw, xshape in [
        ("h8s3_deblur", dict(in_channels=3, upsampling_rate=1, residual=True, inner_residual=True,
                             num_conv_blocks=1, hidden_channels=8, inout_convs=True, scales=3),
         (2, 3, 48, 48)),
        ("h8s3_sr2", dict(in_channels=3, upsampling_rate=2, residual=True, inner_residual=True,
                          num_conv_blocks=1, hidden_channels=8, inout_convs=True, scales=3),
         (2, 3, 24, 24)),
        ("h2s4_pad", dict(in_channels=3, upsampling_rate=1, residual=True, inner_residual=True,
                          num_conv_blocks=1, hidden_channels=2, inout_convs=True, scales=4),
         (1, 3, 45, 50)),
        ("h8s2_nb2", dict(in_channels=3, upsampling_rate=1, residual=False, inner_residual=False,
                          num_conv_blocks=2, hidden_channels=8, inout_convs=True, scales=2),
         (1, 3, 16, 16)),
        ("h2s4_sr4", dict(in_channels=3, upsampling_rate=4, residual=True, inner_residual=True,
                          num_conv_blocks=1, hidden_channels=2, inout_convs=True, scales=4),
         (1, 3, 12, 12)),
    ]:
        torch.manual_seed(0)
        m = conv.ConvolutionalModel(**kw)
        with torch.no_grad():
            for n, p in m.named_parameters():
                if ".ln." in n:
                    p.add_(0.05 * torch.randn_like(p))
        x = _rand(xshape, 100).requires_grad_(True)
        y = m(x)
        ct = _randn(tuple(y.shape), 101)
        m.zero_grad()
        (gx,) = torch.autograd.grad(y, x, ct, retain_graph=True)
        y.backward(ct)
        arrs = {"x": _np(x), "y": _np(y), "ct": _np(ct), "gx": _np(gx),
                "cfg": np.frombuffer(json.dumps(kw).encode(), dtype=np.uint8)}
        for k2, v in sd_np(m).items():
            arrs[f"sd.{k2}"] = v
        # float64 run of the same reference module on the same (f32-valued) inputs: the tight pin.
        # Parameter gradients are stored rounded to f32 (6e-8 relative) to keep the fixture small.
        import copy
        m64 = copy.deepcopy(m).double()
        x64 = x.detach().double().requires_grad_(True)
        y64 = m64(x64)
        m64.zero_grad()
        (gx64,) = torch.autograd.grad(y64, x64, ct.double(), retain_graph=True)
        y64.backward(ct.double())
        arrs["y64"] = _np(y64)
        arrs["gx64"] = _np(gx64)
        for k2, v in grads_np(m64).items():
            arrs[f"grad.{k2}"] = v.astype(np.float32)
        _save(f"g7_unet_{tag}", **arrs)

    # ---------------- G8: default-config manifest ----------------------------------------------
    print("G8 manifest")
    manifest = {}
    for tag, up in [("deblur", 1), ("sr2", 2), ("sr4", 4)]:
        with torch.device("meta"):
            m = conv.ConvolutionalModel(in_channels=3, upsampling_rate=up, residual=True,
                                        inner_residual=True, num_conv_blocks=1, hidden_channels=32,
                                        inout_convs=True, scales=5)
        sd = m.state_dict()
        manifest[tag] = {
            "num_parameters": int(sum(p.numel() for p in m.parameters())),
            "keys": [[k2, list(v.shape)] for k2, v in sd.items()],
        }
        print(f"  {tag}: {manifest[tag]['num_parameters']} parameters, {len(sd)} tensors")
    with open(os.path.join(OUT, "g8_state_dict_manifest.json"), "w") as f:
        json.dump(manifest, f, indent=0)

    # ---------------- G9: LR schedule ----------------------------------------------------------
    print("G9 scheduler")
    out = {}
    for kind in ["delayed_linear_decay", "multi_step_decay"]:
        for epochs in [10, 20]:
            p = torch.nn.Parameter(torch.zeros(1))
            opt = torch.optim.Adam([p], lr=1e-4)
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                sch = scheduler.get_lr_scheduler(opt, epochs, kind)
                lrs = []
                for _ in range(epochs):
                    lrs.append(opt.param_groups[0]["lr"])
                    opt.step()
                    sch.step()
            out[f"{kind}.{epochs}"] = lrs
    with open(os.path.join(OUT, "g9_lr_schedule.json"), "w") as f:
        json.dump(out, f, indent=0)

    # ---------------- G11: composite proposed loss ---------------------------------------------
    # Reference pieces (U-Net, BlurV2 / Downsampling, SureGaussianLoss, padded_downsampling_transform)
    # composed by the EI glue restated from deepinv v0.2.0's documented EILoss behaviour
    # ([recollection]): x2 = T(x_net) under no_grad; y2 = A(x2) + sigma*n; x3 = model(y2);
    # loss_ei = alpha * mean((x3 - x2)^2).  All randomness injected and saved.
    print("G11 composite loss")
    for tag, task, rate in [("deblur", "deblurring", 1), ("sr2", "sr", 2)]:
        torch.manual_seed(0)
        m = conv.ConvolutionalModel(in_channels=3, upsampling_rate=rate, residual=True,
                                    inner_residual=True, num_conv_blocks=1, hidden_channels=8,
                                    inout_convs=True, scales=3)
        sd32 = sd_np(m)
        m = m.double()          # float64 throughout: this golden pins formulas, not rounding
        if task == "deblurring":
            phys_op = blur
            margin = 6
        else:
            phys_op = physics.Downsampling(rate=rate, antialias=True)
            margin = 0
        B, S = 2, 24 if task == "deblurring" else 16
        y = _rand((B, 3, S, S), 110).double()
        b = _randn((B, 3, S - 2 * margin, S - 2 * margin), 111).double()
        n = _randn((B, 3, S, S), 112).double()
        r = torch.tensor([0.75, 0.5]).double()
        c = torch.tensor([[0.3, -0.2], [-0.5, 0.6]]).double().view(B, 1, 1, 2)
        sigma = 5 / 255
        saved = (torch.randn, torch.randn_like)
        torch.randn = lambda *a, **kw: b.clone()
        torch.randn_like = lambda t, **kw: b.clone()
        try:
            x_net = m(y)
            lf = sure.SureGaussianLoss(sigma=sigma, margin=margin, cropped_div=True, averaged_cst=None)
            l_sure = lf(y=y, x_net=x_net, physics=phys_op, model=m)
        finally:
            torch.randn, torch.randn_like = saved
        with torch.no_grad():
            x2 = transforms.padded_downsampling_transform(
                x_net, downsampling_rate=r, center=c, mode="bicubic", padding_mode="reflection",
                antialiased=False)
        y2 = phys_op.A(x2) + sigma * n
        x3 = m(y2)
        l_ei = torch.nn.functional.mse_loss(x3, x2)
        total = l_sure + 1.0 * l_ei
        m.zero_grad()
        total.backward()
        arrs = {"y": _np(y), "b": _np(b), "n": _np(n), "rate": _np(r), "center": _np(c.view(-1, 2)),
                "x_net": _np(x_net), "x2": _np(x2), "x3": _np(x3),
                "loss_sure": _np(l_sure), "loss_ei": _np(l_ei), "loss": _np(total)}
        arrs = {k2: (v.astype(np.float32) if k2 in ("y", "b", "n", "rate", "center") else v)
                for k2, v in arrs.items()}
        for k2, v in sd32.items():
            arrs[f"sd.{k2}"] = v
        for k2, v in grads_np(m).items():
            arrs[f"grad.{k2}"] = v.astype(np.float32)
        _save(f"g11_proposed_{tag}", **arrs)

    print("done ->", OUT)


if __name__ == "__main__":
    if sys.argv[1:] == ["--only", "g12"]:
        gen_noise2inverse()
    elif sys.argv[1:] == ["--only", "g13"]:
        gen_r2r()
    elif sys.argv[1:] == ["--only", "g14"]:
        gen_crop()
    elif sys.argv[1:] == ["--only", "g15"]:
        gen_loss_glue()
    else:
        main()
        gen_noise2inverse()
        gen_r2r()
        gen_crop()
        gen_loss_glue()

"""Training losses (reference call surface: src/losses/__init__.py).

`get_loss(args, physics)` -> module with forward(x, y, model) -> 0-dim tensor. The hot path is
method "proposed": crop the batch to 48x48 (Loss.forward), then SURE + scale-equivariant loss
(ProposedLoss). `supervised`, `css`, `noise2inverse` and `sure` keep their surface (thin model(y) + one
loss term); the R2R alternative and the Rotate / Shift transforms sit behind the same ProposedLoss surface.

One build-side optimisation, on by default and exactly equivalent per image: the two network passes
that do not depend on each other -- model(y) and model(y + tau*b) -- run as ONE batch of 2B images
(`fuse_passes`), doubling the row count of every weight-bandwidth-bound GEMM. `fuse_passes=False`
(env SEI_NO_FUSED_PASSES=1) issues the reference's literal call sequence.
"""
from os import environ

import torch
from torch.nn import Module
from torch.nn.functional import l1_loss

from crop import CropPair
from physics._ops import axpy
from transforms import CombinedTransform, Rotate, ScalingTransform, Shift
from .ei import EILoss, SupLoss, mse
from .r2r import R2REILoss
from .sure import SureGaussianLoss, draw_probe


def _stochastic_depth_masks(model, batch):
    """The per-sample stochastic-depth masks of one model call (SwinIR in training mode), else None."""
    if model is None:
        return None
    backbone = model.get_backbone() if hasattr(model, "get_backbone") else model
    draw = getattr(backbone, "draw_drop_masks", None)
    return draw(batch) if draw is not None else None


def _concat_masks(first, second):
    """Two B-sized mask sets -> one for the pass of 2B images that runs both calls at once."""
    if first is None or second is None:
        return first if second is None else second
    if hasattr(first, "buffer") and hasattr(second, "buffer"):          # models.swinir.DropMasks: one concatenation
        return first.rebuilt_on(torch.cat([first.buffer, second.buffer], dim=1))
    return [None if a is None else tuple(torch.cat([u, v]) for u, v in zip(a, b)) for a, b in zip(first, second)]


def _with_masks(model, masks):
    """`model` as a callable that hands `masks` to the backbone (extra positional arguments ignored, as Model)."""
    if masks is None:
        return model
    return lambda v, *ignored: model(v, drop_masks=masks)


class _ModelPlusOneLoss(Module):
    """x_net = model(y); one deepinv-style loss term (reference :13-64).

    `graph_safe`: the step consumes no host-side randomness and issues no host sync, so graphs.GraphedLossStep
    may capture it; `needs_x`: the ground truth enters the loss (the graph then needs a static x as well)."""
    graph_safe = True
    needs_x = False

    def __init__(self, physics, loss):
        super().__init__()
        self.physics = physics
        self.loss = loss

    def draw(self, y, model=None):
        """The step's device-side random numbers, in the order the step consumes them."""
        masks = _stochastic_depth_masks(model, y.shape[0])
        return None if masks is None else {"drop": [masks]}

    def forward(self, x, y, model, draws=None):
        if draws is None:
            draws = self.draw(y, model)
        call = model if draws is None else _with_masks(model, draws["drop"][0])
        return self.loss(x=x, x_net=call(y), y=y, physics=self.physics, model=model)


class SupervisedLoss(_ModelPlusOneLoss):
    needs_x = True

    def __init__(self, physics):
        metric = mse()
        if "SUPERVISED_L1" in environ:
            print("SUPERVISED_L1")
            metric = l1_loss
        super().__init__(physics, SupLoss(metric=metric))


class CSSLoss(_ModelPlusOneLoss):
    needs_x = True

    def __init__(self, physics):
        super().__init__(physics, SupLoss(metric=mse()))


class Noise2InverseLoss(_ModelPlusOneLoss):
    needs_x = True

    def __init__(self, physics):
        super().__init__(physics, SupLoss(metric=mse()))


class SURELoss(_ModelPlusOneLoss):
    def __init__(self, noise_level, cropped_div, averaged_cst, margin, physics):
        super().__init__(physics, SureGaussianLoss(sigma=noise_level / 255, cropped_div=cropped_div,
                                                   averaged_cst=averaged_cst, margin=margin))

    def draw(self, y, model=None):
        first = _stochastic_depth_masks(model, y.shape[0])                       # model(y)
        draws = {"b": draw_probe(y, self.loss.div_margin)}
        second = _stochastic_depth_masks(model, y.shape[0])                      # model(y + tau b)
        if first is not None:
            draws["drop"] = [first, second]
        return draws

    def forward(self, x, y, model, draws=None):
        y = y.contiguous()
        if draws is None:
            draws = self.draw(y, model)
        drop = draws.get("drop", [None, None])
        return self.loss(x=x, x_net=_with_masks(model, drop[0])(y), y=y, physics=self.physics,
                         model=_with_masks(model, drop[1]), b=draws["b"])


class _SumLossTerms(torch.autograd.Function):
    """a + b for two 0-dim loss terms on the GPU as one own launch (the backward hands the incoming gradient to both)."""

    @staticmethod
    def forward(ctx, a, b):
        import _native as N
        out = torch.empty((), dtype=torch.float32, device=a.device)
        N.call("sei_add_scalars", a.data_ptr(), b.data_ptr(), out.data_ptr())
        return out

    @staticmethod
    def backward(ctx, go):
        return go, go


def _sum_terms(a, b):
    if isinstance(a, torch.Tensor) and isinstance(b, torch.Tensor) and a.is_cuda and b.is_cuda and a.dim() == 0 and b.dim() == 0 \
            and a.dtype == torch.float32 and b.dtype == torch.float32:
        return _SumLossTerms.apply(a, b)
    return a + b


def _stacked_probe_input(y, b, tau):
    """[y, y + tau b] along the batch: the input of the fused 2B pass (src/losses/sure.py:24 beside the model call on y)."""
    if y.is_cuda and y.dtype == torch.float32 and b.dtype == torch.float32 and y.is_contiguous() and b.is_contiguous() \
            and y.shape == b.shape and y.numel() % 4 == 0 and not (y.requires_grad or b.requires_grad):
        import _native as N
        out = torch.empty((2 * y.shape[0],) + tuple(y.shape[1:]), dtype=y.dtype, device=y.device)
        N.call("sei_stack_axpy", y.data_ptr(), b.data_ptr(), float(tau), out.data_ptr(), y.numel())
        return out
    return torch.cat([y, axpy(y, b, tau)], dim=0)


class ProposedLoss(Module):
    needs_x = False             # the ground truth never enters the proposed loss
    keep_outputs = False        # diagnostics: keep the restored images of the (fused) first pass in `self.kept`

    def __init__(self, blueprint, sure_alternative, noise_level, stop_gradient, sure_cropped_div,
                 sure_averaged_cst, sure_margin, alpha_tradeoff, transforms, physics, fuse_passes=None):
        super().__init__()
        self.physics = physics
        self.graph_safe = False
        if transforms == "Scaling_Transforms":
            ei_transform = ScalingTransform(**blueprint[ScalingTransform.__name__])
        elif transforms == "Shifts":
            ei_transform = Shift()
        elif transforms == "Rotations+Shifts":
            ei_transform = CombinedTransform([Rotate(), Shift()])
        elif transforms == "Rotations":
            ei_transform = Rotate()
        else:
            raise ValueError(f"Unknown transforms: {transforms}")
        assert sure_alternative in [None, "r2r"]
        if sure_alternative == "r2r":
            self.loss_fns = [R2REILoss(transform=ei_transform, sigma=noise_level / 255, no_grad=stop_gradient,
                                       metric=mse())]
            self.compute_x_net = False
            self.fuse_passes = False
            return
        self.sure = SureGaussianLoss(sigma=noise_level / 255, cropped_div=sure_cropped_div,
                                     averaged_cst=sure_averaged_cst, margin=sure_margin)
        self.ei = EILoss(metric=mse(), transform=ei_transform, no_grad=stop_gradient, weight=alpha_tradeoff)
        self.loss_fns = [self.sure, self.ei]
        self.compute_x_net = True
        if fuse_passes is None:
            fuse_passes = "SEI_NO_FUSED_PASSES" not in environ
        self.fuse_passes = fuse_passes
        # Every random number of the default step is a device-side draw whose shape is known from y alone
        # (probe b, then the per-image rate and centre, then the measurement noise of the EI branch): `draw`
        # can hoist them in front of the step, which is what lets graphs.GraphedLossStep replay the step with
        # the numbers an eager step would have drawn. The `normal` kind and the antialiased variant read
        # their rates on the host (.item() / .tolist()), the Shift transform draws on the host.
        padded = isinstance(ei_transform, ScalingTransform) and ei_transform.kind == "padded"
        self.graph_safe = padded and not ei_transform.antialias

    def draw(self, y, model=None):
        """{"b", "rate", "center", "noise"[, "drop"]} in the order an eager step consumes them: the probe
        (src/losses/sure.py:13-22), the transform's rates and centres (src/transforms.py:15-24, inside EILoss), the
        measurement noise of A(T x_net) (y's shape); with a backbone that uses stochastic depth (SwinIR in training
        mode) also one set of per-sample masks per model call, drawn where that call happens. None when this
        configuration draws on the host or in data-dependent shapes."""
        if not self.graph_safe:
            return None
        B = y.shape[0]
        first = _stochastic_depth_masks(model, B)                               # model(y)
        if first is None and self._fused_draws_ok(y):
            # no stochastic-depth masks between the draws: the four of them in one launch, the same numbers (draw_into)
            draws = {"b": torch.zeros_like(y), "rate": torch.empty(B, dtype=y.dtype, device=y.device),
                     "center": torch.empty((B, 1, 1, 2), dtype=y.dtype, device=y.device), "noise": torch.empty_like(y)}
            if self.draw_into(draws, y, model):          # (first is None: this backbone draws no masks at all)
                return draws
        draws = {"b": draw_probe(y, self.sure.div_margin)}
        second = _stochastic_depth_masks(model, B)                              # model(y + tau b)
        # the reference's order whatever the pass structure (masks, probe, masks): a seeded run consumes the device
        # generator exactly as the literal call sequence does; the fused 2B pass takes the two sets side by side
        drop = [_concat_masks(first, second)] if self.fuse_passes else [first, second]
        draws["rate"], draws["center"] = self.ei.T.sample(B, y.device, y.dtype)
        draws["noise"] = torch.randn_like(y)
        drop.append(_stochastic_depth_masks(model, B))                          # model(y2) of the EI branch
        if any(d is not None for d in drop):
            draws["drop"] = drop
        return draws

    def _fused_draws_ok(self, y):
        """sei_proposed_draws serves this step: float32 on the GPU, tensors in torch's one-element-per-thread regime, the
        padded scaling transform's rate table; SEI_TORCH_DRAWS=1 keeps torch's own calls (tests compare the two)."""
        import _native as N
        return (y.is_cuda and y.dtype == torch.float32 and y.dim() == 4 and environ.get("SEI_TORCH_DRAWS") != "1"
                and y.numel() <= N.lib().sei_proposed_draws_max_numel() and y.numel() % 4 == 0
                and hasattr(self.ei.T, "transform") and hasattr(self.ei.T.transform, "downsampling_rates")
                and not torch.cuda.is_current_stream_capturing())

    def draw_into(self, static, y, model=None):
        """`draw` with the numbers landing in the buffers of `static` (an earlier result of `draw`): the same generator
        calls in the same order -- randn of the probe's interior, the transform's two uniform draws, randn of the
        measurement noise -- without the fresh tensors and the copies (graphs.GraphedLossStep, before every replay).
        False: not available for this configuration (stochastic depth masks ride along: the generic path copies)."""
        if not self.graph_safe or "drop" in static or not hasattr(self.ei.T, "sample_into"):
            return False
        m = self.sure.div_margin
        b = static["b"]
        if self._fused_draws_ok(y) and b.is_contiguous() and static["noise"].is_contiguous():
            # csrc/draws.hip: what the four torch calls below would draw from the device generator's (seed, offset),
            # in one launch; the generator is advanced as those calls advance it (4 Philox outputs per thread each)
            import _native as N
            from transforms import _table
            gen = torch.cuda.default_generators[y.device.index if y.device.index is not None else torch.cuda.current_device()]
            seed, offset = gen.initial_seed(), gen.get_offset()
            rates = self.ei.T.transform.downsampling_rates
            table = _table(rates, y.device, y.dtype)
            N.call("sei_proposed_draws", seed & 0xFFFFFFFFFFFFFFFF, offset, b.data_ptr(), y.size(0), y.size(1), y.size(2), y.size(3),
                   m, table.data_ptr(), len(rates), static["rate"].data_ptr(), static["center"].data_ptr(),
                   static["noise"].data_ptr())
            gen.set_offset(offset + 16)
            return True
        if m == 0:
            torch.randn(tuple(y.shape), dtype=y.dtype, device=y.device, out=b)
        else:                                       # the border of `b` is zero and stays zero
            b[:, :, m:-m, m:-m].copy_(torch.randn(y.size(0), y.size(1), y.size(2) - 2 * m, y.size(3) - 2 * m,
                                                  device=y.device, dtype=y.dtype))
        self.ei.T.sample_into(static["rate"], static["center"])
        torch.randn(tuple(y.shape), dtype=y.dtype, device=y.device, out=static["noise"])
        return True

    def forward(self, x, y, model, draws=None):
        y = y.contiguous()
        if draws is None:
            draws = self.draw(y, model) if self.compute_x_net else None
        ei_kw = {} if draws is None else {"transform_params": (draws["rate"], draws["center"]),
                                          "noise": draws["noise"]}
        drop = draws.get("drop") if draws is not None else None
        calls = [model] * 3 if drop is None else [_with_masks(model, m) for m in drop]
        if not self.fuse_passes:
            x_net = calls[0](y) if self.compute_x_net else None
            loss = 0
            for loss_fn in self.loss_fns:
                if loss_fn is getattr(self, "ei", None):
                    loss = loss + loss_fn(x=x, x_net=x_net, y=y, physics=self.physics, model=calls[2], **ei_kw)
                elif loss_fn is getattr(self, "sure", None):
                    kw = {"b": draws["b"]} if draws is not None else {}
                    loss = loss + loss_fn(x=x, x_net=x_net, y=y, physics=self.physics, model=calls[1], **kw)
                else:
                    loss = loss + loss_fn(x=x, x_net=x_net, y=y, physics=self.physics, model=model)
            return loss
        # Same arithmetic and the same RNG draw order (`draw`: masks of model(y), probe b, masks of model(y + tau b)),
        # but model(y) and model(y + tau b) share one pass of 2B images.
        B = y.shape[0]
        b = draws["b"] if draws is not None else draw_probe(y, self.sure.div_margin)
        if self.ei.no_grad and drop is None:
            # the second model call's input (A T x_net + noise) is a constant: the two calls' backward passes are
            # independent and the backbone may run them as one (models/_joint.py)
            backbone = model.get_backbone() if hasattr(model, "get_backbone") else model
            if getattr(backbone, "flat_grads", None) is not None:
                from models import _joint
                _joint.recorder_of(backbone).expect_pair()
        both = calls[0](_stacked_probe_input(y, b, self.sure.tau))
        y12 = self.physics.A(both)
        x_net = both[:B]
        loss = self.sure(y=y, x_net=x_net, physics=self.physics, model=model, b=b, y12=y12)
        if self.keep_outputs:
            self.kept = {"x_net": x_net.detach()}
        return _sum_terms(loss, self.ei(x=x, x_net=x_net, y=y, physics=self.physics, model=calls[-1], **ei_kw))


class Loss(Module):
    def __init__(self, physics, blueprint, noise_level, sure_cropped_div, sure_averaged_cst, sure_margin,
                 method, crop_training_pairs, crop_size):
        super().__init__()
        if method == "supervised":
            self.loss = SupervisedLoss(physics=physics)
        elif method == "css":
            self.loss = CSSLoss(physics=physics)
        elif method == "noise2inverse":
            self.loss = Noise2InverseLoss(physics=physics)
        elif method == "sure":
            self.loss = SURELoss(physics=physics, noise_level=noise_level, cropped_div=sure_cropped_div,
                                 averaged_cst=sure_averaged_cst, margin=sure_margin)
        elif method == "proposed":
            self.loss = ProposedLoss(physics=physics, blueprint=blueprint, noise_level=noise_level,
                                     sure_cropped_div=sure_cropped_div, sure_averaged_cst=sure_averaged_cst,
                                     sure_margin=sure_margin, **blueprint[ProposedLoss.__name__])
        else:
            raise ValueError(f"Unknwon method: {method}")

        if crop_training_pairs:
            self.xy_size_ratio = physics.rate if hasattr(physics, "rate") else 1
            self.crop_fn = CropPair(location="random", size=crop_size)
        else:
            self.crop_fn = None
        if "HOMOGENEOUS_SWINIR" in environ:
            self.crop_fn = None

    def forward(self, x, y, model, draws=None):
        """draws: the step's device-side random numbers (`self.loss.draw(y_cropped)`), injected by tests and by
        graphs.GraphedLossStep; drawn here, in the reference's order, when absent."""
        if self.crop_fn is not None:
            x, y = self.crop_fn(x, y, xy_size_ratio=self.xy_size_ratio)
        if draws is None:
            return self.loss(x=x, y=y, model=model)
        return self.loss(x=x, y=y, model=model, draws=draws)


def get_loss(args, physics):
    if args.partial_sure:
        if args.sure_margin is not None:
            sure_margin = args.sure_margin
        elif args.task == "deblurring":
            assert physics.task == "deblurring"
            kernel = physics.filter
            sure_margin = (max(kernel.shape[-2], kernel.shape[-1]) - 1) // 2
        elif args.task == "sr":
            sure_margin = 2 if args.partial_sure_sr else 0
    else:
        assert args.sure_margin is None
        sure_margin = 0

    blueprint = {
        Loss.__name__: {
            "crop_training_pairs": args.Loss__crop_training_pairs,
            "crop_size": args.Loss__crop_size,
        },
        ProposedLoss.__name__: {
            "stop_gradient": args.ProposedLoss__stop_gradient,
            "sure_alternative": args.ProposedLoss__sure_alternative,
            "alpha_tradeoff": args.ProposedLoss__alpha_tradeoff,
            "transforms": args.ProposedLoss__transforms,
        },
        ScalingTransform.__name__: {
            "kind": args.ScalingTransform__kind,
            "antialias": args.ScalingTransform__antialias,
        },
    }
    return Loss(physics=physics, blueprint=blueprint, method=args.method, noise_level=args.noise_level,
                sure_cropped_div=args.sure_cropped_div, sure_averaged_cst=args.sure_averaged_cst,
                sure_margin=sure_margin, **blueprint[Loss.__name__])

"""Model factory (reference call surface: src/models/__init__.py).

`get_model(args, physics, device)` returns a `Model` whose forward ignores extra positional arguments,
with `get_backbone() / get_weights() / load_weights()` working on the backbone's state_dict exactly as
the reference's (same keys), so checkpoints interchange.

In scope: kind "Proposed" with architecture "Convolutional" (the in-tree U-Net) or "Transformer" (the
reference's default: deepinv's SwinIR with the arguments at src/models/__init__.py:51-74, rebuilt in
models/swinir.py from the published architecture -- deepinv is not part of the reference tree, parity
unpinned), and the trivial "Identity". The test-time baselines (DIP, PnP, BM3D, DiffPIR, DPS, TV, ...)
raise a clear error instead of silently running something else.
"""
from os import environ

from torch import nn
from torch.nn import Module

from .convolutional import ConvolutionalModel
from .swinir import SwinIR

_OUT_OF_SCOPE = ("DeepImagePrior", "PlugAndPlay", "BM3D", "DiffPIR_DRUNet", "DiffPIR_DiffUNet", "DPS", "TV",
                 "Upsample")


class Identity(Module):
    def forward(self, y):
        return y


class InverseFilter(Module):
    """The least-squares pseudo-inverse of the physics as a "model" (reference :22-28): physics.A_dagger(y)."""

    def __init__(self, physics):
        super().__init__()
        self.physics = physics

    def forward(self, y):
        return self.physics.A_dagger(y)


class ProposedModel(Module):
    def __init__(self, blueprint, architecture, sampling_rate):
        super().__init__()
        if architecture == "Convolutional":
            self.model = ConvolutionalModel(in_channels=3, upsampling_rate=sampling_rate,
                                            **blueprint[ConvolutionalModel.__name__])
        elif architecture == "Transformer":
            if sampling_rate > 1:
                upsampling_rate, upsampler = sampling_rate, "pixelshuffle"
                if "HOMOGENEOUS_SWINIR" in environ:                 # reference :43-47
                    print("\nUsing homogeneous SwinIR\n")
                    upsampling_rate, upsampler = 1, None
            else:
                upsampling_rate, upsampler = 1, None
            self.model = SwinIR(upscale=upsampling_rate, upsampler=upsampler, img_size=48, patch_size=1, in_chans=3,
                                embed_dim=180, depths=[6, 6, 6, 6, 6, 6], num_heads=[6, 6, 6, 6, 6, 6], window_size=8,
                                mlp_ratio=2, qkv_bias=True, qk_scale=None, drop_rate=0.0, attn_drop_rate=0.0,
                                drop_path_rate=0.1, norm_layer=nn.LayerNorm, ape=False, patch_norm=True,
                                use_checkpoint=False, img_range=1.0, resi_connection="1conv", pretrained=None)
        else:
            raise ValueError(f"Unknown model kind: {architecture}")

    def forward(self, y, **kwargs):
        return self.model(y, **kwargs)

    def get_backbone(self):
        return self.model


class Model(Module):
    def __init__(self, blueprint, kind, physics, task, sr_factor, device, noise_level, data_parallel_devices):
        super().__init__()
        sampling_rate = sr_factor if task == "sr" else 1
        if kind == "Proposed":
            self.model = ProposedModel(blueprint=blueprint, sampling_rate=sampling_rate,
                                       **blueprint[ProposedModel.__name__])
        elif kind == "Identity":
            self.model = Identity()
        elif kind == "InverseFilter":
            self.model = InverseFilter(physics=physics)
        elif kind in _OUT_OF_SCOPE:
            raise NotImplementedError(f"model kind {kind!r} is an evaluation baseline outside the training "
                                      "hot path this build implements")
        else:
            raise ValueError(f"Unknown model kind: {kind}")
        if data_parallel_devices is not None:
            # reference :142-145,178-182: nn.DataParallel over these device ids, re-replicating the weights on
            # every forward. Here one process per GPU owns a replica; the same devices are used by:
            n = len(data_parallel_devices)
            ids = ",".join(data_parallel_devices)
            raise NotImplementedError(
                f"--data_parallel_devices {ids}: single-process nn.DataParallel is replaced by one process per "
                f"GPU with an RCCL gradient all-reduce (parallel.py). Equivalent launch on the same devices:\n"
                f"  HIP_VISIBLE_DEVICES={ids} python -m torch.distributed.run --nnodes=1 --nproc-per-node {n} "
                f"--master-addr 127.0.0.1 train.py <the same flags without --data_parallel_devices> "
                f"--batch_size <batch_size / {n}>")

    def forward(self, x, *args, **kwargs):
        """Extra positional arguments are ignored, as upstream (:148-149); keyword arguments (the SwinIR backbone's
        injected stochastic-depth masks) go to the backbone."""
        return self.model(x, **kwargs)

    def get_backbone(self):
        model = self.model
        return model.get_backbone() if isinstance(model, ProposedModel) else model

    def get_weights(self):
        return self.get_backbone().state_dict()

    def load_weights(self, state_dict):
        self.get_backbone().load_state_dict(state_dict)


def get_model(args, physics, device):
    data_parallel_devices = (args.data_parallel_devices.split(",")
                             if args.data_parallel_devices is not None else None)
    blueprint = {
        ConvolutionalModel.__name__: {
            "residual": args.ConvolutionalModel__residual,
            "inner_residual": args.ConvolutionalModel__inner_residual,
            "num_conv_blocks": args.ConvolutionalModel__num_conv_blocks,
            "inout_convs": args.ConvolutionalModel__inout_convs,
            "hidden_channels": args.ConvolutionalModel__hidden_channels,
            "scales": args.ConvolutionalModel__scales,
        },
        Model.__name__: {
            "task": args.task,
            "sr_factor": args.sr_factor,
            "noise_level": args.noise_level,
            "kind": args.model_kind,
        },
        ProposedModel.__name__: {"architecture": args.ProposedModel__architecture},
    }
    return Model(blueprint=blueprint, physics=physics, device=device,
                 data_parallel_devices=data_parallel_devices, **blueprint[Model.__name__])

"""Model factory (reference call surface: src/models/__init__.py).

`get_model(args, physics, device)` returns a `Model` whose forward ignores extra positional arguments,
with `get_backbone() / get_weights() / load_weights()` working on the backbone's state_dict exactly as
the reference's (same keys), so checkpoints interchange.

In scope this round: kind "Proposed" with architecture "Convolutional" (the in-tree U-Net) and the
trivial "Identity". The reference's default architecture "Transformer" is deepinv's SwinIR, which is
not part of the reference tree; it and the test-time baselines (DIP, PnP, BM3D, DiffPIR, DPS, TV, ...)
raise a clear error instead of silently running something else.
"""
from torch.nn import Module

from .convolutional import ConvolutionalModel

_OUT_OF_SCOPE = ("DeepImagePrior", "PlugAndPlay", "BM3D", "DiffPIR_DRUNet", "DiffPIR_DiffUNet", "DPS", "TV",
                 "InverseFilter", "Upsample")


class Identity(Module):
    def forward(self, y):
        return y


class ProposedModel(Module):
    def __init__(self, blueprint, architecture, sampling_rate):
        super().__init__()
        if architecture == "Convolutional":
            self.model = ConvolutionalModel(in_channels=3, upsampling_rate=sampling_rate,
                                            **blueprint[ConvolutionalModel.__name__])
        elif architecture == "Transformer":
            raise NotImplementedError(
                "--ProposedModel__architecture Transformer is deepinv's SwinIR (not vendored by the "
                "reference); this build implements the in-tree U-Net: pass "
                "--ProposedModel__architecture Convolutional")
        else:
            raise ValueError(f"Unknown model kind: {architecture}")

    def forward(self, y):
        return self.model(y)

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

    def forward(self, x, *args):
        return self.model(x)

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

"""Degradation operators (reference call surface: src/physics/__init__.py).

`get_physics(args, device)` returns an object obeying the deepinv LinearPhysics protocol the
reference's losses, datasets and scripts rely on: A, A_adjoint, __call__ = noise(A(.)), attributes
noise_model (.sigma), task, filter/kernel or rate, and the literally-named "__manager" attribute
exposing randomly_degrade(x, seed). The arithmetic runs in the HIP kernels of libsei_hip.so.
"""
from os.path import exists

import torch

from rng import fork_rng
from ._base import GaussianNoise, LinearPhysics
from .blur import Blur, BlurV2
from .downsampling import Downsampling
from .kernels import get_kernel


class BlurKernel:
    """A kernel named in the table, or a path to a torch.save'd 2-D tensor (reference :16-26)."""

    def __init__(self, kernel_path):
        self.kernel_path = kernel_path

    def to_tensor(self, device):
        if exists(self.kernel_path):
            kernel = torch.load(self.kernel_path)
        else:
            kernel = get_kernel(name=self.kernel_path)
        return kernel[None, None].to(device)


class PhysicsManager:
    def __init__(self, blueprint, task, device, noise_level, v2):
        if task == "deblurring":
            kernel = BlurKernel(**blueprint[BlurKernel.__name__]).to_tensor(device)
            physics = BlurV2(kernel=kernel) if v2 else Blur(filter=kernel, padding="circular", device=device)
        elif task == "sr":
            physics = Downsampling(antialias=True, **blueprint[Downsampling.__name__])
        elif task == "invert_a_tomography_like_filter":
            raise ValueError("task 'invert_a_tomography_like_filter' (CTLikeFilter) is outside the "
                             "hot path this build implements")
        else:
            raise ValueError(f"Unknown task: {task}")

        physics.noise_model = GaussianNoise(sigma=noise_level / 255)
        self.task = task
        physics.task = task
        setattr(physics, "__manager", self)
        self.physics = physics

    def get_physics(self):
        return self.physics

    def randomly_degrade(self, x, seed):
        """noise(A(x)); with a seed, under a forked RNG seeded with it (reference :65-74)."""
        with fork_rng(enabled=seed is not None):
            if seed is not None:
                torch.manual_seed(seed)
            return self.physics.noise_model(self.physics.A(x))


def get_physics(args, device):
    blueprint = {
        PhysicsManager.__name__: {"task": args.task, "noise_level": args.noise_level, "v2": args.physics_v2},
        BlurKernel.__name__: {"kernel_path": args.kernel},
        Downsampling.__name__: {"rate": args.sr_factor, "true_adjoint": args.physics_true_adjoint},
    }
    manager = PhysicsManager(blueprint=blueprint, device=device, **blueprint[PhysicsManager.__name__])
    return manager.get_physics()

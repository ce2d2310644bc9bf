"""Times the matrix-core resampler (sei_sepmap2_bf16) against the f32 FMA kernels at the shapes of a bench step."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scale-equivariant-imaging_amd"))
from models import _mats, _ops as ops  # noqa: E402


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps


for kind, B, H, C in [("down", 64, 48, 32), ("up", 64, 24, 128), ("down", 64, 24, 128), ("up", 64, 12, 512),
                      ("down", 64, 12, 512), ("up", 32, 24, 128), ("down", 32, 48, 32), ("down", 96, 48, 32), ("up", 96, 24, 128),
                      ("down", 96, 24, 128)]:
    fwd, bwd = _mats.resample_matrices(kind, H, H, 2, "cuda")
    Ho = fwd[0].shape[0]
    x = torch.randn((B, H, H, C), device="cuda")
    g = torch.randn((B, Ho, Ho, C), device="cuda")
    f32 = timed(lambda: ops.sepmap2(x, fwd, Ho, Ho))
    m16 = timed(lambda: ops.sepmap2_16(x, fwd, Ho, Ho))
    f32b = timed(lambda: ops.sepmap2(g, bwd, H, H))
    m16b = timed(lambda: ops.sepmap2_16(g, bwd, H, H))
    print(f"{kind} {B}x{H}x{H}x{C} -> {Ho}: forward f32 {f32:.1f} us, matrix cores {m16:.1f} us | transposed f32 {f32b:.1f}, "
          f"matrix cores {m16b:.1f}", flush=True)

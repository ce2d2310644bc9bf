"""Times the pipelined depthwise kernel (seg 66) against the first tiled kernel (seg 65), and the fused
conv1 -> LayerNorm launch against depthwise + stand-alone LayerNorm, at the shapes of a bench step (2B = 64 crops)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scale-equivariant-imaging_amd"))
from models import _ops as ops  # noqa: E402


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


for (B, H, W, C) in [(64, 48, 48, 32), (64, 24, 24, 128), (64, 12, 12, 512), (32, 48, 48, 32), (32, 24, 24, 128),
                     (64, 192, 192, 32), (64, 96, 96, 128)]:
    x = torch.randn((B, H, W, C), device="cuda")
    r = torch.randn((B, H, W, C), device="cuda")
    w, b = torch.randn((C, 1, 7, 7), device="cuda") * 0.1, torch.randn(C, device="cuda")
    gamma, beta = torch.randn(C, device="cuda"), torch.randn(C, device="cuda")
    t65 = timed(lambda: ops.dwconv7(x, w, b, seg=65))
    t66 = timed(lambda: ops.dwconv7(x, w, b, seg=66))
    d65 = timed(lambda: ops.dwconv7(x, w, None, flip=True, res=r, res_scale=2.0, seg=65))
    d66 = timed(lambda: ops.dwconv7(x, w, None, flip=True, res=r, res_scale=2.0, seg=66))
    M = B * H * W

    def unfused():
        h1 = ops.dwconv7(x, w, b, seg=65)
        return ops.layer_norm16(h1.view(M, C), gamma, beta)

    tu = timed(unfused)
    tf = timed(lambda: ops.dwconv7_ln(x, w, b, gamma, beta, out16=True))
    mb = 8 * M * C / 1e6
    print(f"{B}x{H}x{W}x{C} ({mb:.1f} MB in+out): fwd tiled {t65:.1f} us, pipelined {t66:.1f} us | dX tiled {d65:.1f}, "
          f"pipelined {d66:.1f} | conv1+LN: tiled + LN kernel {tu:.1f} us, sei_dwconv7_ln_fwd {tf:.1f} us", flush=True)

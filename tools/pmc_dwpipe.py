"""One launch each of the first tiled (seg 65) and the pipelined (seg 66) depthwise kernel at a bench shape, for
rocprofv3 --pmc passes (tools/pmc_summary.py reads the csv)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scale-equivariant-imaging_amd"))
from models import _ops as ops  # noqa: E402

B, H, W, C = (int(v) for v in (sys.argv[1:5] if len(sys.argv) >= 5 else (64, 192, 192, 32)))
x = torch.randn((B, H, W, C), device="cuda")
w, b = torch.randn((C, 1, 7, 7), device="cuda") * 0.1, torch.randn(C, device="cuda")
for _ in range(3):
    ops.dwconv7(x, w, b, seg=65)
    ops.dwconv7(x, w, b, seg=66)
torch.cuda.synchronize()

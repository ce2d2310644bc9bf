"""One launch group of the fused pointwise MLP (sei_mlp_fused_fwd / _bwd) at C = 128, M = 36,864 (the 2B pass of level 1),
for rocprofv3 --pmc passes."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scale-equivariant-imaging_amd"))
import _native as N  # noqa: E402

M, C = 36864, 128
h2 = torch.randn((M, C), device="cuda").bfloat16()
W2 = (torch.randn((4 * C, C), device="cuda") * C ** -0.5).bfloat16()
W3 = (torch.randn((C, 4 * C), device="cuda") * (4 * C) ** -0.5).bfloat16()
b2, b3 = torch.randn(4 * C, device="cuda"), torch.randn(C, device="cuda")
x = torch.randn((M, C), device="cuda")
out = torch.empty((M, C), device="cuda")
go = torch.randn((M, C), device="cuda")
gh2 = torch.empty((M, C), device="cuda")
go16 = torch.empty((M, C), device="cuda", dtype=torch.bfloat16)
h4 = torch.empty((M, 4 * C), device="cuda", dtype=torch.bfloat16)
gh3 = torch.empty((M, 4 * C), device="cuda", dtype=torch.bfloat16)
W3T, W2T = W3.t().contiguous(), W2.t().contiguous()
for _ in range(3):
    N.call("sei_mlp_fused_fwd", h2.data_ptr(), W2.data_ptr(), b2.data_ptr(), W3.data_ptr(), b3.data_ptr(), x.data_ptr(), 1.0,
           out.data_ptr(), M, C)
    N.call("sei_mlp_fused_bwd", go.data_ptr(), h2.data_ptr(), W2.data_ptr(), b2.data_ptr(), W3T.data_ptr(), W2T.data_ptr(),
           gh2.data_ptr(), go16.data_ptr(), h4.data_ptr(), gh3.data_ptr(), M, C)
torch.cuda.synchronize()

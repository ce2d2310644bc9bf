"""Times the fused pointwise MLP (sei_mlp_fused_fwd / _bwd) at the shapes of a bench step against the two GEMMs
(+ the cast) it replaces."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scale-equivariant-imaging_amd"))
import _native as N  # noqa: E402
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


for M, C in [(147456, 32), (73728, 32), (36864, 128), (18432, 128)]:
    h2 = torch.randn((M, C), device="cuda").bfloat16()
    W2 = (torch.randn((4 * C, C), device="cuda") * C ** -0.5).bfloat16()
    W3 = (torch.randn((C, 4 * C), device="cuda") * (4 * C) ** -0.5).bfloat16()
    b2, b3 = torch.randn(4 * C, device="cuda"), torch.randn(C, device="cuda")
    x, out, go, gh2 = (torch.randn((M, C), device="cuda") for _ in range(4))
    go16 = torch.empty((M, C), device="cuda", dtype=torch.bfloat16)
    h3 = torch.empty((M, 4 * C), device="cuda")
    h4 = torch.empty((M, 4 * C), device="cuda", dtype=torch.bfloat16)
    gh3 = torch.empty((M, 4 * C), device="cuda", dtype=torch.bfloat16)
    W3T, W2T = W3.t().contiguous(), W2.t().contiguous()
    ff = timed(lambda: N.call("sei_mlp_fused_fwd", h2.data_ptr(), W2.data_ptr(), b2.data_ptr(), W3.data_ptr(), b3.data_ptr(),
                              x.data_ptr(), 1.0, out.data_ptr(), M, C))
    fb = timed(lambda: N.call("sei_mlp_fused_bwd", go.data_ptr(), h2.data_ptr(), W2.data_ptr(), b2.data_ptr(), W3T.data_ptr(),
                              W2T.data_ptr(), gh2.data_ptr(), go16.data_ptr(), h4.data_ptr(), gh3.data_ptr(), M, C))

    def unf_fwd():
        ops.gemm_nt16(h2, W2, M, 4 * C, C, ops.EPI_BIAS_GELU, out32=h3, bias=b2, D2_16=h4)
        ops.gemm_nt16(h4, W3, M, C, 4 * C, ops.EPI_BIAS_RES, out32=out, bias=b3, R1=x)

    def unf_bwd():
        g16 = ops.cast16(go)
        ops.gemm_nt16(g16, W3, M, 4 * C, C, ops.EPI_MUL_DGELU, out16=gh3, R1=h3, b_rmajor=True)
        ops.gemm_nt16(gh3, W2, M, C, 4 * C, ops.EPI_NONE, out32=gh2, b_rmajor=True)

    uf, ub = timed(unf_fwd), timed(unf_bwd)
    print(f"M = {M}, C = {C}: forward fused {ff:.1f} us, two GEMMs {uf:.1f} | backward fused {fb:.1f} us, cast + two GEMMs "
          f"{ub:.1f}", flush=True)

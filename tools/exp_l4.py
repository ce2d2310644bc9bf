"""The bottleneck level's weight-streaming GEMMs (M = 576 / 288 rows of the 2B / B pass against 8192 x 32768 weights) on
the quadrant schedule's tile choices: automatic dispatch against 288x256 (code 31) and 288x128 (code 32) tiles."""
import os, sys, statistics, torch
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "scale-equivariant-imaging_amd"))
from models import _ops
def once(fn, iters=5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
cases = [(M, N, K, kind) for M in (576, 288) for (N, K, kind) in ((8192, 32768, "none_kr"), (8192, 32768, "res"), (32768, 8192, "dgelu_kr"), (32768, 8192, "gelu"))]
for (M, N, K, kind) in cases:
    kr = kind.endswith("_kr")
    A = torch.randn((M, K), device="cuda").bfloat16(); B = (0.02 * torch.randn((K, N) if kr else (N, K), device="cuda")).bfloat16()
    out = torch.empty((M, N), device="cuda"); o16 = torch.empty((M, N), device="cuda", dtype=torch.bfloat16)
    bias = torch.randn(N, device="cuda"); R1 = torch.randn((M, N), device="cuda")
    def f(tile):
        if kind == "none_kr": _ops.gemm_nt16(A, B, M, N, K, _ops.EPI_NONE, out32=out, b_rmajor=True, tile=tile)
        elif kind == "dgelu_kr": _ops.gemm_nt16(A, B, M, N, K, _ops.EPI_MUL_DGELU, out16=o16, R1=R1, b_rmajor=True, tile=tile)
        elif kind == "gelu": _ops.gemm_nt16(A, B, M, N, K, _ops.EPI_BIAS_GELU, out32=out, bias=bias, D2_16=o16, tile=tile)
        else: _ops.gemm_nt16(A, B, M, N, K, _ops.EPI_BIAS_RES, out32=out, bias=bias, R1=R1, tile=tile)
    codes = [0, 31, 32]
    times = {c: [] for c in codes}
    for rnd in range(4):
        for c in codes:
            f(c); torch.cuda.synchronize()
            times[c].append(once(lambda: f(c)))
    print(f"{M}x{N}x{K} {kind:9s}: " + "  ".join(f"tile {c}: {statistics.median(t):6.0f} us {2.0*M*N*K/statistics.median(t)/1e6:5.0f} TF" for c, t in times.items()), flush=True)

"""The bottleneck level's GEMMs at the reference's default batch (8: 144 rows for the 2B crops, 72 for the B
measurements) under the tile shapes of gemm_bf16nt.hip: pure weight streaming (537 MB per launch). Tools build."""
import os, sys, statistics, torch
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "scale-equivariant-imaging_amd"))
sys.path.insert(1, os.path.dirname(os.path.abspath(__file__)))
import _tuning; _tuning.use()
from models import _ops
import _native
lib = _native.lib()
def once(fn, iters=6):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for M in (144, 72, 288):
    for (N, K, kind) in [(32768, 8192, "gelu"), (32768, 8192, "dgelu_kr"), (8192, 32768, "res"), (8192, 32768, "none_kr")]:
        kr = kind.endswith("_kr")
        A = torch.randn((M, K), device="cuda").bfloat16()
        B = (torch.randn((K, N) if kr else (N, K), device="cuda") * K ** -0.5).bfloat16()
        o32 = torch.empty((M, N), device="cuda"); o16 = torch.empty((M, N), device="cuda", dtype=torch.bfloat16)
        bias = torch.randn(N, device="cuda"); r = torch.randn((M, N), device="cuda")
        if kind == "gelu": f = lambda: _ops.gemm_nt16(A, B, M, N, K, _ops.EPI_BIAS_GELU, out32=o32, D2_16=o16, bias=bias)
        elif kind == "dgelu_kr": f = lambda: _ops.gemm_nt16(A, B, M, N, K, _ops.EPI_MUL_DGELU, out16=o16, R1=r, b_rmajor=True)
        elif kind == "res": f = lambda: _ops.gemm_nt16(A, B, M, N, K, _ops.EPI_BIAS_RES, out32=o32, bias=bias, R1=r)
        else: f = lambda: _ops.gemm_nt16(A, B, M, N, K, _ops.EPI_NONE, out32=o32, b_rmajor=True)
        times = {}
        for t in (0, 1, 3, 31, 32):
            lib.sei_debug_set_nt_tile(t)
            try:
                f(); torch.cuda.synchronize()
            except RuntimeError:
                continue
            times[t] = statistics.median(once(f) for _ in range(3))
        lib.sei_debug_set_nt_tile(0)
        print(f"{M}x{N}x{K} {kind}: " + "  ".join(f"t{t} {v:.0f}" for t, v in times.items()), flush=True)

"""Quadrant schedule (tile codes 30-33) against the automatic choice on the K-contiguous GEMMs of the deep levels,
with the epilogues the model uses: interleaved rounds in one process, median times."""
import os, sys, statistics, torch
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "scale-equivariant-imaging_amd"))
sys.path.insert(1, os.path.dirname(os.path.abspath(__file__)))
import _tuning; _tuning.use()    # process-wide tile switches live in the tools-only build
from models import _ops
import _native
NAMES = {0: "auto", 30: "256x256", 31: "288x256", 32: "288x128", 33: "256x128", 34: "256^2-dma", 35: "256^2 mfma"}
def once(fn, iters=5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
shapes = [(2304, 8192, 64, "gelu"), (2304, 8192, 2048, "gelu"), (2304, 8192, 64, "dgelu_kr"), (2304, 8192, 2048, "dgelu_kr"),
          (9216, 2048, 64, "gelu"), (9216, 2048, 512, "gelu"), (36864, 512, 64, "gelu"), (36864, 512, 128, "gelu")]
for (M, N, K, kind) in shapes:
    kr = kind.endswith("_kr")
    A = torch.randn((M, K), device="cuda").bfloat16(); B = torch.randn((K, N) if kr else (N, K), device="cuda").bfloat16()
    out = torch.empty((M, N), device="cuda"); o16 = torch.empty((M, N), device="cuda", dtype=torch.bfloat16)
    bias = torch.randn(N, device="cuda"); R1 = torch.randn((M, N), device="cuda")
    if kind == "none": f = lambda: _ops.gemm_nt16(A, B, M, N, K, _ops.EPI_NONE, out32=out)
    elif kind == "none_kr": f = lambda: _ops.gemm_nt16(A, B, M, N, K, _ops.EPI_NONE, out32=out, b_rmajor=True)
    elif kind == "dgelu_kr": f = lambda: _ops.gemm_nt16(A, B, M, N, K, _ops.EPI_MUL_DGELU, out16=o16, R1=R1, b_rmajor=True)
    elif kind == "gelu": f = lambda: _ops.gemm_nt16(A, B, M, N, K, _ops.EPI_BIAS_GELU, out32=out, bias=bias, D2_16=o16)
    else: f = lambda: _ops.gemm_nt16(A, B, M, N, K, _ops.EPI_BIAS_RES, out32=out, bias=bias, R1=R1)
    codes = [31, 0]
    NAMES[1] = "128x128"; NAMES[15] = "128x128 1-stage"
    times = {c: [] for c in codes}
    for rnd in range(5):
        for code in codes:
            _native.lib().sei_debug_set_nt_tile(code)
            f(); torch.cuda.synchronize()
            times[code].append(once(f))
    print(f"{M}x{N}x{K} {kind}: " + "  ".join(f"{NAMES[c]} {statistics.median(t):.0f}us/{2.0*M*N*K/statistics.median(t)/1e6:.0f}TF" for c, t in times.items()), flush=True)
_native.lib().sei_debug_set_nt_tile(0)

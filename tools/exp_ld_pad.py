"""Does a power-of-two leading dimension (K = 32768 bf16 = 64 KB row stride) cost the K-contiguous GEMMs of the bottleneck
level anything? Same GEMM with the operands' rows padded by 64 elements."""
import os, sys, statistics, torch
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "scale-equivariant-imaging_amd"))
from models import _ops
def once(fn, iters=5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for (M, N, K) in ((576, 8192, 32768), (288, 8192, 32768), (576, 32768, 8192), (2304, 2048, 8192)):
    line = f"{M}x{N}x{K} NT bias+res:"
    for pad in (0, 64, 0, 64):
        A = torch.randn((M, K + pad), device="cuda").bfloat16(); B = torch.randn((N, K + pad), device="cuda").bfloat16()
        out = torch.empty((M, N), device="cuda"); bias = torch.randn(N, device="cuda"); R1 = torch.randn((M, N), device="cuda")
        f = lambda: _ops.gemm_nt16(A, B, M, N, K, _ops.EPI_BIAS_RES, out32=out, bias=bias, R1=R1, lda=K + pad, ldb=K + pad)
        f(); torch.cuda.synchronize()
        t = statistics.median(once(f) for _ in range(5))
        line += f"  pad {pad}: {t:6.0f} us ({2.0*M*N*K/t/1e6:5.0f} TF)"
    print(line, flush=True)

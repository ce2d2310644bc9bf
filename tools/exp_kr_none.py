import os, sys, torch
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "scale-equivariant-imaging_amd"))
from models import _ops
import _native
def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for (M, N, K) in [(147456, 128, 32), (147456, 32, 128), (147456, 128, 128), (147456, 128, 64)]:
    A = torch.randn((M, K), device="cuda").bfloat16(); B = torch.randn((K, N), device="cuda").bfloat16()
    out = torch.empty((M, N), device="cuda"); o16 = torch.empty((M, N), dtype=torch.bfloat16, device="cuda"); R1 = torch.randn((M, N), device="cuda")
    t1 = timeit(lambda: _ops.gemm_nt16(A, B, M, N, K, _ops.EPI_NONE, out32=out, b_rmajor=True))
    t2 = timeit(lambda: _ops.gemm_nt16(A, B, M, N, K, _ops.EPI_NONE, out16=o16, b_rmajor=True))
    t3 = timeit(lambda: _ops.gemm_nt16(A, B, M, N, K, _ops.EPI_MUL_DGELU, out16=o16, R1=R1, b_rmajor=True))
    Bk = B.t().contiguous()
    t4 = timeit(lambda: _ops.gemm_nt16(A, Bk, M, N, K, _ops.EPI_NONE, out32=out))
    print(f"{M}x{N}x{K}: kr f32-out {t1:6.1f} us | kr bf16-out {t2:6.1f} | kr dgelu bf16-out {t3:6.1f} | kk f32-out {t4:6.1f}")

import os, sys, torch
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "scale-equivariant-imaging_amd"))
sys.path.insert(1, os.path.dirname(os.path.abspath(__file__)))
import _tuning; _tuning.use()    # process-wide tile switches live in the tools-only build
from models import _ops
import _native
def timeit(fn, iters=8):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for (M, N, K) in [(576, 32768, 8192), (576, 8192, 32768), (288, 32768, 8192), (288, 8192, 32768), (304, 520, 2104)]:
    A = torch.randn((M, K), device="cuda").bfloat16(); B = torch.randn((K, N), device="cuda").bfloat16()
    ref = A.float() @ B.float()
    line = f"{M}x{N}x{K} kr:"
    for tile in (1, 0, 1, 0):
        _native.lib().sei_debug_set_nt_tile(tile)
        out = torch.full((M, N), float("nan"), device="cuda")
        _ops.gemm_nt16(A, B, M, N, K, _ops.EPI_NONE, out32=out, b_rmajor=True)
        err = float((out - ref).abs().max() / ref.abs().max())
        t = timeit(lambda: _ops.gemm_nt16(A, B, M, N, K, _ops.EPI_NONE, out32=out, b_rmajor=True))
        line += f"  {'128x128' if tile else 'auto   '} {t:5.0f} us err {err:.0e} |"
    print(line)
_native.lib().sei_debug_set_nt_tile(0)

"""K sweep of one 256x256 tile through the ping-pong kernel: fixed vs per-k-tile cost (kernel time via events)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "scale-equivariant-imaging_amd"))
sys.path.insert(1, os.path.dirname(os.path.abspath(__file__)))
import _tuning; _tuning.use()    # process-wide tile switches live in the tools-only build
from models import _ops
import _native
def timeit(fn, iters=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for tile in (20, 1):
    _native.lib().sei_debug_set_nt_tile(tile)
    for (M, N) in ((256, 256), (2048, 2048 * 8)):
        line = f"tile {tile} {M}x{N}:"
        for K in (64, 128, 256, 512, 1024, 4096):
            A = torch.randn((M, K), device="cuda").bfloat16(); B = torch.randn((N, K), device="cuda").bfloat16()
            out = torch.empty((M, N), device="cuda")
            t = timeit(lambda: _ops.gemm_nt16(A, B, M, N, K, _ops.EPI_BIAS, out32=out, bias=out[0]))
            line += f"  K={K}: {t:6.1f} us"
        print(line)
_native.lib().sei_debug_set_nt_tile(0)

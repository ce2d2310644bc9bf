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
names = {0: "auto", 1: "128x128", 2: "128x256", 3: "192x256", 5: "96x256"}
for (M, N, K) in [(576, 32768, 8192), (576, 8192, 32768), (288, 32768, 8192), (288, 8192, 32768), (2304, 8192, 2048), (1152, 8192, 2048), (1152, 2048, 8192)]:
    A = torch.randn((M, K), device="cuda").bfloat16(); B = torch.randn((N, K), device="cuda").bfloat16()
    out = torch.zeros((M, N), device="cuda"); bias = torch.zeros(N, device="cuda")
    line = f"{M}x{N}x{K}:"
    for tile in (0, 1, 2, 3, 5):
        _native.lib().sei_debug_set_nt_tile(tile)
        t = timeit(lambda: _ops.gemm_nt16(A, B, M, N, K, _ops.EPI_BIAS, out32=out, bias=bias))
        line += f"  {names[tile]} {t:5.0f} us ({2.0*M*N*K/t/1e6:4.0f} TF)"
    print(line)
_native.lib().sei_debug_set_nt_tile(0)

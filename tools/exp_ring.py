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
names = {0: "auto", 1: "128x128 s2", 15: "128x128 s1x3"}
for (M, N, K) in [(2304, 8192, 2048), (2304, 2048, 8192), (9216, 2048, 512), (36864, 512, 128), (4096, 4096, 4096), (8192, 8192, 1024)]:
    A = torch.randn((M, K), device="cuda").bfloat16(); B = torch.randn((N, K), device="cuda").bfloat16()
    out = torch.zeros((M, N), device="cuda"); ref = None
    for tile in (0, 1, 15, 1, 15):
        _native.lib().sei_debug_set_nt_tile(tile)
        t = timeit(lambda: _ops.gemm_nt16(A, B, M, N, K, _ops.EPI_NONE, out32=out))
        if ref is None: ref = out.clone()
        err = float((out - ref).abs().max() / ref.abs().max())
        print(f"{M}x{N}x{K} {names[tile]:12s}: {t:8.0f} us {2.0*M*N*K/t/1e6:7.1f} TF  maxdiff {err:.1e}")
_native.lib().sei_debug_set_nt_tile(0)

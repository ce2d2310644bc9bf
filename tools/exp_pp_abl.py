"""Ablations of the 256x256 ping-pong kernel (timing only): which of fragment reads / DMA / barriers costs what."""
import os, sys, statistics, torch
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "scale-equivariant-imaging_amd"))
sys.path.insert(1, os.path.dirname(os.path.abspath(__file__)))
import _tuning; _tuning.use()    # process-wide tile switches live in the tools-only build
from models import _ops
import _native
NAMES = {0: "auto", 20: "pp", 21: "-reads", 22: "-dma", 23: "-reads-dma", 24: "-bar", 25: "-reads-bar", 26: "-dma-bar", 27: "mfma only"}
def once(fn, iters=6):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for (M, N, K) in ((4096, 4096, 4096), (8192, 8192, 4096), (2304, 8192, 2048)):
    A = torch.randn((M, K), device="cuda").bfloat16(); B = torch.randn((N, K), device="cuda").bfloat16()
    out = torch.empty((M, N), device="cuda")
    times = {c: [] for c in NAMES}
    for rnd in range(5):
        for code in NAMES:
            _native.lib().sei_debug_set_nt_tile(code)
            f = lambda: _ops.gemm_nt16(A, B, M, N, K, _ops.EPI_NONE, out32=out)
            f(); torch.cuda.synchronize()
            times[code].append(once(f))
    print(f"{M}x{N}x{K}: " + "  ".join(f"{NAMES[c]} {statistics.median(t):.0f}us/{2.0*M*N*K/statistics.median(t)/1e6:.0f}TF" for c, t in times.items()))
_native.lib().sei_debug_set_nt_tile(0)

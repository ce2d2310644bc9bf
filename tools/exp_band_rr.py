"""Band width of the tile order for the bottleneck weight gradients (both orientations), 128x128 loop."""
import os, sys, statistics, torch
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "scale-equivariant-imaging_amd"))
sys.path.insert(1, os.path.dirname(os.path.abspath(__file__)))
import _tuning; _tuning.use()    # process-wide tile switches live in the tools-only build
import _native
from _native import call
def once(fn, iters=4):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for (M, N, K1, K2) in ((8192, 32768, 288, 576), (32768, 8192, 288, 576)):
    A1 = torch.randn((K1, M), device="cuda").bfloat16(); A2 = torch.randn((K2, M), device="cuda").bfloat16()
    B1 = torch.randn((K1, N), device="cuda").bfloat16(); B2 = torch.randn((K2, N), device="cuda").bfloat16()
    D = torch.empty((M, N), device="cuda")
    f = lambda: call("sei_gemm_bf16nt_dw2", A1.data_ptr(), A2.data_ptr(), M, B1.data_ptr(), B2.data_ptr(), N, D.data_ptr(), M, N, K1, K2, 0)
    bands = [0, 2, 4, 6, 8, 12, 16, 24, 32, 64]
    times = {b: [] for b in bands}
    for rnd in range(4):
        for b in bands:
            _native.lib().sei_debug_set_nt_tile(100 + b)            # 100 = automatic
            f(); torch.cuda.synchronize()
            times[b].append(once(f))
    print(f"{M}x{N}: " + "  ".join(f"band {b or 'auto'} {statistics.median(t):.0f}us" for b, t in times.items()), flush=True)
_native.lib().sei_debug_set_nt_tile(100)

"""Weight-gradient GEMMs (both operands reduction-major, two K segments) of the deep levels: automatic choice
(quadrant schedule) against the 128x128 loop (tile code 1 / 15)."""
import os, sys, statistics, torch
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "scale-equivariant-imaging_amd"))
sys.path.insert(1, os.path.dirname(os.path.abspath(__file__)))
import _tuning; _tuning.use()    # process-wide tile switches live in the tools-only build
import _native
from _native import call
def once(fn, iters=5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for (M, N, K1, K2) in ((2048, 8192, 1152, 2304), (8192, 32768, 288, 576), (8192, 32768, 32, 32), (8192, 32768, 1152, 2304)):
    A1 = torch.randn((K1, M), device="cuda").bfloat16(); A2 = torch.randn((K2, M), device="cuda").bfloat16()
    B1 = torch.randn((K1, N), device="cuda").bfloat16(); B2 = torch.randn((K2, N), device="cuda").bfloat16()
    D = torch.empty((M, N), device="cuda")
    f = lambda: call("sei_gemm_bf16nt_dw2", A1.data_ptr(), A2.data_ptr(), M, B1.data_ptr(), B2.data_ptr(), N, D.data_ptr(), M, N, K1, K2, 0)
    times = {0: [], 30: [], 36: [], 37: []}
    for rnd in range(5):
        for code in times:
            _native.lib().sei_debug_set_nt_tile(code)
            f(); torch.cuda.synchronize()
            times[code].append(once(f))
    fl = 2.0 * M * N * (K1 + K2)
    print(f"{M}x{N}x({K1}+{K2}): " + "  ".join(f"code {c} {statistics.median(t):.0f}us/{fl/statistics.median(t)/1e6:.0f}TF" for c, t in times.items()), flush=True)
_native.lib().sei_debug_set_nt_tile(0)

"""Do the reduction-major operands of the weight gradients suffer from their power-of-two row pitch? Same GEMM with
the rows padded by 64 / 256 elements (lda / ldb arguments), quadrant schedule (30) and 128x128 loop (0)."""
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
for (M, N, K1, K2) in ((8192, 32768, 288, 576), (2048, 8192, 1152, 2304)):
    D = torch.empty((M, N), device="cuda")
    line = f"{M}x{N}x({K1}+{K2}):"
    for pad in (0, 64, 256):
        A1 = torch.randn((K1, M + pad), device="cuda").bfloat16(); A2 = torch.randn((K2, M + pad), device="cuda").bfloat16()
        B1 = torch.randn((K1, N + pad), device="cuda").bfloat16(); B2 = torch.randn((K2, N + pad), device="cuda").bfloat16()
        f = lambda: call("sei_gemm_bf16nt_dw2", A1.data_ptr(), A2.data_ptr(), M + pad, B1.data_ptr(), B2.data_ptr(), N + pad, D.data_ptr(), M, N, K1, K2, 0)
        for code in (0, 30, 36):
            _native.lib().sei_debug_set_nt_tile(code)
            ts = []
            for rnd in range(3):
                f(); torch.cuda.synchronize(); ts.append(once(f))
            line += f"  pad {pad} code {code}: {statistics.median(ts):.0f}us"
    print(line, flush=True)
_native.lib().sei_debug_set_nt_tile(0)

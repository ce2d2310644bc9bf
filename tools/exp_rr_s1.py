import os, sys, torch
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "scale-equivariant-imaging_amd"))
sys.path.insert(1, os.path.dirname(os.path.abspath(__file__)))
import _tuning; _tuning.use()    # process-wide tile switches live in the tools-only build
from models import _ops
import _native
def timeit(fn, iters=6):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for (M, N, K) in [(8192, 32768, 864), (32768, 8192, 864), (2048, 8192, 3456), (512, 2048, 13824), (128, 512, 55296)]:
    A = torch.randn((K, M), device="cuda").bfloat16(); B = torch.randn((K, N), device="cuda").bfloat16()
    out = torch.zeros((M, N), device="cuda")
    line = f"{M}x{N}x{K} rr (store):"
    for tile in (0, 15, 16, 17, 0, 15, 16, 17):
        _native.lib().sei_debug_set_nt_tile(tile)
        chk = torch.full((M, N), float("nan"), device="cuda")
        _ops.gemm_nt16(A, B, M, N, K, _ops.EPI_NONE, out32=chk, a_rmajor=True, b_rmajor=True)
        if tile == 0: ref = chk.clone()
        else: assert float((chk - ref).abs().max()) <= 1e-3 * float(ref.abs().max()), "mismatch"
        t = timeit(lambda: _ops.gemm_nt16(A, B, M, N, K, _ops.EPI_NONE, out32=out, a_rmajor=True, b_rmajor=True))
        nm = {0: "128x128 s2", 15: "128x128 s1", 16: "128x256 s2", 17: "128x256 s1"}[tile]
        line += f"  {nm} {t:6.0f} us ({2.0*M*N*K/t/1e6:4.0f} TF) |"
    print(line)
_native.lib().sei_debug_set_nt_tile(0)

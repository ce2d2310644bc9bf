"""256x256 ping-pong schedule (tile code 20) against the automatic choice: correctness and time."""
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
shapes = [  # (M, N, K, a_rmajor, b_rmajor)
    (4096, 4096, 4096, 0, 0), (2304, 8192, 2048, 0, 0), (2304, 2048, 8192, 0, 0), (576, 32768, 8192, 0, 0),
    (2304, 8192, 2048, 0, 1), (2304, 2048, 8192, 0, 1), (576, 8192, 32768, 0, 1),
    (2048, 8192, 3456, 1, 1), (8192, 32768, 864, 1, 1), (512, 2048, 13824, 1, 1), (300, 520, 200, 0, 0), (264, 776, 72, 1, 1)]
if len(sys.argv) > 1: shapes = shapes[:int(sys.argv[1])]
for (M, N, K, ar, br) in shapes:
    A = torch.randn((K, M) if ar else (M, K), device="cuda").bfloat16()
    B = torch.randn((K, N) if br else (N, K), device="cuda").bfloat16()
    ref = (A.float().t() if ar else A.float()) @ (B.float() if br else B.float().t())
    line = f"{M}x{N}x{K} {'r' if ar else 'k'}{'r' if br else 'k'}:"
    for tile in (0, 20):
        _native.lib().sei_debug_set_nt_tile(tile)
        out = torch.full((M, N), float("nan"), device="cuda")
        _ops.gemm_nt16(A, B, M, N, K, _ops.EPI_NONE, out32=out, a_rmajor=bool(ar), b_rmajor=bool(br))
        torch.cuda.synchronize()
        err = float((out - ref).abs().max() / ref.abs().max())
        t = timeit(lambda: _ops.gemm_nt16(A, B, M, N, K, _ops.EPI_NONE, out32=out, a_rmajor=bool(ar), b_rmajor=bool(br)))
        line += f"  {'auto' if tile == 0 else 'pp256'} {t:7.0f} us {2.0*M*N*K/t/1e6:7.1f} TF err {err:.1e} |"
    print(line)
_native.lib().sei_debug_set_nt_tile(0)

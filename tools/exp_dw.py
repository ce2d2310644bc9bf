import os, sys, torch
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "scale-equivariant-imaging_amd"))
sys.path.insert(1, os.path.dirname(os.path.abspath(__file__)))
import _tuning; _tuning.use()    # process-wide tile switches live in the tools-only build
from models import _ops
def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
import _native
names = {1: "128x128", 2: "128x256", 3: "192x256", 4: "256x256", 5: " 96x256"}
for (Mo, No) in [(8192, 32768), (576, 32768), (2304, 8192)]:
    out = torch.zeros((Mo, No), device="cuda")
    for K in ((576, 2304) if Mo == 8192 else (8192,) if Mo == 576 else (2048,)):
        A = torch.randn((Mo, K), device="cuda").bfloat16(); B = torch.randn((No, K), device="cuda").bfloat16()
        for tile in (1, 2, 3, 4, 5):
            _native.lib().sei_debug_set_nt_tile(tile)
            for epi, name in ((_ops.EPI_NONE, "store"), (_ops.EPI_ACCUM, "accum")):
                t = timeit(lambda: _ops.gemm_nt16(A, B, Mo, No, K, epi, out32=out))
                print(f"out {Mo}x{No} K={K:5d} tile {names[tile]} {name}: {t:8.0f} us  {2.0*Mo*No*K/t/1e6:7.1f} TF")
_native.lib().sei_debug_set_nt_tile(0)
# pure streaming reference: out += 1 over the same 1 GiB buffer
t = timeit(lambda: out.add_(1.0)); print(f"torch add_ (RMW 2x{out.numel()*4/1e9:.2f} GB): {t:.0f} us -> {2*out.numel()*4/t/1e6:.2f} TB/s")
t = timeit(lambda: out.zero_()); print(f"torch zero_: {t:.0f} us -> {out.numel()*4/t/1e6:.2f} TB/s")

"""sei_sepmap2_small's wave kernel at the deep levels' shapes of configs[1] (2B = 64, B = 32, 3B = 96 images)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "scale-equivariant-imaging_amd"))
import _native as N
from models import _ops, _mats
def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
tot = 0.0
for (B, hin, hout, C) in ((64, 6, 3, 2048), (64, 3, 6, 8192), (64, 6, 12, 2048), (32, 6, 3, 2048), (32, 3, 6, 8192), (32, 6, 12, 2048),
                          (96, 6, 3, 8192), (96, 12, 6, 2048), (96, 3, 6, 2048), (96, 6, 12, 512), (64, 12, 6, 512)):
    kind = "down" if hout < hin else "up"
    mats = _mats.resample_matrices(kind, hin, hin, 2, "cuda")[0]
    x = torch.randn((B, hin, hin, C), device="cuda"); y = torch.empty((B, hout, hout, C), device="cuda")
    L1, R1, L2, R2 = mats[:4]
    def f(): N.call("sei_sepmap2_small", x.data_ptr(), y.data_ptr(), 0, B, hin, hin, hout, hout, C, L1.data_ptr(), R1.data_ptr(), L2.data_ptr(), R2.data_ptr())
    f(); torch.cuda.synchronize()
    ref = _ops.sepmap2(x, mats, hout, hout)
    t = timeit(f); tot += t
    print(f"{B} x {hin}x{hin} -> {hout}x{hout} x {C}: {t:6.1f} us  bit-identical to the two-launch kernels: {bool(torch.equal(y, ref))}", flush=True)
print(f"total {tot:.1f} us")

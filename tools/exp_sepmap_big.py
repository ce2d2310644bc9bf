"""sei_sepmap2_big (constant-matrix GEMM kernel, bf16 intermediate) against the f32 FMA kernels on the resampler calls of
the x4 network (64 crops at 192 / 96 pixels) and of the un-cropped 256-pixel series: us per call, GB/s of algorithmic bytes."""
import os, sys, statistics, torch
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "scale-equivariant-imaging_amd"))
from models import _ops, _mats
def once(fn, iters=5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); torch.cuda.synchronize()
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for kind, B, H, C in [("down", 64, 192, 32), ("up", 64, 96, 32), ("down", 64, 96, 128), ("up", 64, 48, 128),
                      ("down", 32, 192, 32), ("down", 32, 256, 32), ("up", 32, 128, 32), ("down", 32, 128, 128),
                      ("down", 32, 64, 512), ("up", 32, 32, 512), ("down", 32, 32, 2048), ("up", 32, 16, 2048)]:
    fwd, bwd = _mats.resample_matrices(kind, H, H, 2, "cuda")
    x = torch.randn((B, H, H, C), device="cuda")
    Ho = fwd[0].shape[0]
    for tag, mats, xin, ho in (("fwd", fwd, x, Ho), ("bwd", bwd, torch.randn((B, Ho, Ho, C), device="cuda"), H)):
        t16 = once(lambda: _ops.sepmap2_16(xin, mats, ho, ho))
        t32 = once(lambda: _ops.sepmap2(xin, mats, ho, ho))
        nbytes = 4 * B * C * (xin.shape[1] ** 2 + ho * ho)
        print(f"{kind} {tag} B{B} {xin.shape[1]}->{ho} C{C}: matrix cores {t16:.0f} us ({nbytes / t16 / 1e3:.0f} GB/s)  f32 {t32:.0f} us", flush=True)

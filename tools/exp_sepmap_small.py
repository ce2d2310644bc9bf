"""sei_sepmap2_small (one pass through LDS, round 5) against the two-launch float32 kernels on the deep levels' maps of the
benchmarked step (batch 64 / 32 / 96 images of 12 x 12, 6 x 6, 3 x 3 pixels): median us per launch, GB/s of x in + y out."""
import os, sys, statistics, torch
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "scale-equivariant-imaging_amd"))
from models import _mats, _ops


def once(fn, iters=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for B in (64, 96):
    for kind, H, C in (("down", 12, 512), ("down", 6, 2048), ("up", 3, 8192), ("up", 6, 2048), ("up", 12, 512)):
        fwd, bwd = _mats.resample_matrices(kind, H, H, 2, "cuda")
        Ho = fwd[0].shape[0]
        x = torch.randn((B, H, H, C), device="cuda")
        nbytes = 4 * B * C * (H * H + Ho * Ho)
        t_new, t_old, t_st = [], [], []
        for _ in range(5):
            t_new.append(once(lambda: _ops.sepmap2_16(x, fwd, Ho, Ho)))
            t_st.append(float("nan"))      # (the staged kernel for 3 x 3 / 6 x 6 inputs was an experiment-build switch; see sepmap_small.hip)
            t_old.append(once(lambda: _ops.sepmap2(x, fwd, Ho, Ho)))
        a, b, c = statistics.median(t_new), statistics.median(t_old), statistics.median(t_st)
        print(f"B={B} {kind} {H}->{Ho} C={C}: one pass {a:.1f} us ({nbytes / a / 1e3:.0f} GB/s)   staged {c:.1f} us   two launches {b:.1f} us",
              flush=True)

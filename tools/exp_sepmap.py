"""Time sei_sepmap2 at the shapes of the default network (2B = 64 images of 48x48, hidden 32, 5 scales)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "scale-equivariant-imaging_amd"))
from models import _ops, _mats
def timeit(fn, iters=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
B = 64
for lvl in range(4):
    H = 48 >> lvl; C = 32 << (2 * lvl)
    for kind in ("down", "up", "up_bwd"):
        if kind == "down":
            Hi, Ho, Cc = H, H // 2, C              # Downsample: LayerNorm, the map on C channels, then the 1x1 conv
        elif kind == "up":
            Hi, Ho, Cc = H // 2, H, 4 * C          # Upsample: the map on the deeper level's 4C channels, then LN + conv
        else:
            Hi, Ho, Cc = H, H // 2, 4 * C          # its data gradient: the transposed map
        x = torch.randn((B, Hi, Hi, Cc), device="cuda")
        mats = [torch.randn((Ho, Hi), device="cuda") for _ in range(4)]
        mats = tuple(mats) + _mats.pack_for_kernel(mats, "cuda")
        t = timeit(lambda: _ops.sepmap2(x, mats, Ho, Ho))
        mac = B * Cc * (Hi * Ho * 2 * Hi + Ho * Ho * 2 * Hi)
        mb = (x.numel() + B * Ho * Ho * Cc) * 4 / 1e6
        print(f"lvl{lvl} {kind:4s} {Hi}->{Ho} C={Cc}: {t:7.1f} us  {2*mac/t/1e6:6.2f} TF  in+out {mb:.1f} MB -> {mb/t*1e3:.0f} GB/s")

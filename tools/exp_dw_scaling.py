"""How do the depthwise launches of the step scale with the batch? Fixed cost vs per-image cost of sei_dwconv7_fwd (flipped taps +
residual: the data gradient), sei_dwconv7_ln_fwd and sei_dwconv7_bwd_weight_ex at the fine levels.
    python tools/exp_dw_scaling.py"""
import os, sys, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "scale-equivariant-imaging_amd"))
import _native as N


def timeit(fn, iters=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for (H, C) in ((48, 32), (24, 128), (12, 512), (6, 2048), (3, 8192)):
    line = f"{H:2d} x {H:2d} x {C:4d}:"
    for B in (8, 16, 32, 64, 96, 192, 384):
        x = torch.randn(B, H, H, C, device="cuda"); w = torch.randn(C, 49, device="cuda"); b = torch.randn(C, device="cuda")
        res = torch.randn(B, H, H, C, device="cuda"); y = torch.empty_like(x)
        g, be = torch.randn(C, device="cuda"), torch.randn(C, device="cuda")
        h1 = torch.empty_like(x); h2 = torch.empty((B * H * H, C), device="cuda", dtype=torch.bfloat16)
        mean, rstd = torch.empty(B * H * H, device="cuda"), torch.empty(B * H * H, device="cuda")
        need = N.lib().sei_dwconv7_bwd_weight_workspace_ex(B, H, H, C, 0)
        work = torch.empty(max(need, 1), device="cuda")
        t_f = timeit(lambda: N.call("sei_dwconv7_fwd", x.data_ptr(), w.data_ptr(), None, res.data_ptr(), 1.0, y.data_ptr(), B, H, H, C, 1))
        t_l = timeit(lambda: N.call("sei_dwconv7_ln_fwd", x.data_ptr(), w.data_ptr(), b.data_ptr(), g.data_ptr(), be.data_ptr(), h1.data_ptr(),
                                    h2.data_ptr(), 1, mean.data_ptr(), rstd.data_ptr(), B, H, H, C, 1e-6))
        t_w = timeit(lambda: N.call("sei_dwconv7_bwd_weight_ex", x.data_ptr(), y.data_ptr(), None, None, B, H, H, C, work.data_ptr(), need, 0))
        line += f"  B {B:3d}: {t_f:5.1f} / {t_l:5.1f} / {t_w:5.1f}"
    print(line + "   (us: dX / conv+LN fwd / dW)", flush=True)

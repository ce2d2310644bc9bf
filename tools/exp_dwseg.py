"""Time the depthwise kernels at the default network's five levels (2B = 64 crops of 48x48, hidden 32)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "scale-equivariant-imaging_amd"))
import _native as N
def timeit(fn, iters=30):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
B = 64
segs = [0, -1]            # 0 = generic kernels, -1 = automatic (tiled / whole-image)
for lvl in range(5):
    H = max(48 >> lvl, 3); C = 32 << (2 * lvl)
    x = torch.randn((B, H, H, C), device="cuda"); y = torch.empty_like(x); r = torch.randn_like(x)
    w = torch.randn((C, 49), device="cuda"); b = torch.randn((C,), device="cuda")
    gw = torch.zeros_like(w); gb = torch.zeros_like(b)
    mb = x.numel() * 4 / 1e6
    line = f"lvl{lvl} H={H} C={C} ({mb:.1f} MB/tensor):"
    for seg in segs:
        N.lib().sei_debug_set_dw_seg(seg)
        need = N.lib().sei_dwconv7_bwd_weight_workspace(B, H, H, C)
        work = torch.empty(need, device="cuda")
        tf = timeit(lambda: N.call("sei_dwconv7_fwd", x.data_ptr(), w.data_ptr(), b.data_ptr(), None, 0.0, y.data_ptr(), B, H, H, C, 0))
        tb = timeit(lambda: N.call("sei_dwconv7_fwd", x.data_ptr(), w.data_ptr(), None, r.data_ptr(), 1.0, y.data_ptr(), B, H, H, C, 1))
        tw = timeit(lambda: N.call("sei_dwconv7_bwd_weight", x.data_ptr(), r.data_ptr(), gw.data_ptr(), gb.data_ptr(), B, H, H, C, work.data_ptr(), need))
        line += f"  {'generic' if seg == 0 else 'auto'}: fwd {tf:5.1f} bwdx {tb:5.1f} wgrad {tw:5.1f} us |"
    print(line)
N.lib().sei_debug_set_dw_seg(-1)

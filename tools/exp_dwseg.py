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
segs = [16, 0]            # seg argument of the _ex entry points: 16 = generic kernels, 0 = automatic (tiled / whole-image)
for lvl in range(5):
    H = max(48 >> lvl, 3); C = 32 << (2 * lvl)
    x = torch.randn((B, H, H, C), device="cuda"); y = torch.empty_like(x); r = torch.randn_like(x)
    w = torch.randn((C, 49), device="cuda"); b = torch.randn((C,), device="cuda")
    gw = torch.zeros_like(w); gb = torch.zeros_like(b)
    mb = x.numel() * 4 / 1e6
    line = f"lvl{lvl} H={H} C={C} ({mb:.1f} MB/tensor):"
    for seg in segs:
        need = N.lib().sei_dwconv7_bwd_weight_workspace_ex(B, H, H, C, seg)
        work = torch.empty(need, device="cuda")
        tf = timeit(lambda: N.call("sei_dwconv7_fwd_ex", x.data_ptr(), w.data_ptr(), b.data_ptr(), None, 0.0, y.data_ptr(), B, H, H, C, 0, seg))
        tb = timeit(lambda: N.call("sei_dwconv7_fwd_ex", x.data_ptr(), w.data_ptr(), None, r.data_ptr(), 1.0, y.data_ptr(), B, H, H, C, 1, seg))
        tw = timeit(lambda: N.call("sei_dwconv7_bwd_weight_ex", x.data_ptr(), r.data_ptr(), gw.data_ptr(), gb.data_ptr(), B, H, H, C, work.data_ptr(), need, seg))
        line += f"  {'generic' if seg else 'auto'}: fwd {tf:5.1f} bwdx {tb:5.1f} wgrad {tw:5.1f} us |"
    print(line)

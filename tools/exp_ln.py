"""Time LayerNorm forward (bf16 out) / backward at the default network's five levels. COLD: every launch works on
another of 24 tensor sets (1.4 GB in all), so nothing is served from L2 / MALL as it is when one set is re-run."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "scale-equivariant-imaging_amd"))
import _native as N              # (SEI_HIP_LIBRARY=<path>: another build of the library, for A/B runs)
def timeit(fn, iters=30):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
B = 64
H0 = int(sys.argv[1]) if len(sys.argv) > 1 else 48      # 192: the x4 network's grids
SETS = 24 if H0 <= 48 else 6
for lvl in range(5):
    H = max(H0 >> lvl, 3); C = 32 << (2 * lvl); rows = B * H * H
    xs = [torch.randn((rows, C), device="cuda") for _ in range(SETS)]; gys = [torch.randn((rows, C), device="cuda") for _ in range(SETS)]
    gx = torch.empty((rows, C), device="cuda")
    y16 = torch.empty((rows, C), dtype=torch.bfloat16, device="cuda"); y = torch.empty((rows, C), device="cuda")
    g = torch.randn(C, device="cuda"); b = torch.randn(C, device="cuda")
    mean = torch.empty(rows, device="cuda"); rstd = torch.empty(rows, device="cuda")
    gg = torch.zeros(C, device="cuda"); gb = torch.zeros(C, device="cuda")
    need = N.lib().sei_ln_bwd_workspace(rows, C); work = torch.empty(max(need, 1), device="cuda")
    k = [0]
    def nxt():
        k[0] = (k[0] + 1) % SETS
        return xs[k[0]], gys[k[0]]
    def f32():
        x, _ = nxt(); N.call("sei_ln_fwd", x.data_ptr(), g.data_ptr(), b.data_ptr(), y.data_ptr(), mean.data_ptr(), rstd.data_ptr(), rows, C, 1e-6)
    def f16():
        x, _ = nxt(); N.call("sei_ln_fwd_bf16", x.data_ptr(), g.data_ptr(), b.data_ptr(), y16.data_ptr(), mean.data_ptr(), rstd.data_ptr(), rows, C, 1e-6)
    def bwd():
        x, gy = nxt(); N.call("sei_ln_bwd", x.data_ptr(), g.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gy.data_ptr(), gx.data_ptr(), gg.data_ptr(), gb.data_ptr(), rows, C, work.data_ptr(), need)
    tf, tf16, tb = timeit(f32, 48), timeit(f16, 48), timeit(bwd, 48)
    print(f"lvl{lvl} rows={rows} C={C}: fwd f32 {tf:5.1f} us, fwd bf16 {tf16:5.1f} us, bwd {tb:5.1f} us (workspace {need*4/1e6:.2f} MB)")

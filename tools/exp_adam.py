"""Does staggering the base addresses of the Adam streams (p, g, m, v, bf16 copy) change its HBM rate?"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "scale-equivariant-imaging_amd"))
import _native as N
n = 645_063_043 // 64 * 64
def timeit(fn, iters=6):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for trial in range(2):
  for grid in ("library default",):
    for stagger in (0, 1024):
        pad = 4 * max(stagger, 1) + 64
        bufs = [torch.zeros(n + pad, device="cuda") for _ in range(4)]
        p, g, m, v = (b[k * stagger:k * stagger + n] for k, b in enumerate(bufs))
        sh = torch.zeros(n + 2 * pad, dtype=torch.bfloat16, device="cuda")[2 * stagger:2 * stagger + n]
        g.normal_()
        t = timeit(lambda: N.call("sei_adam_fused", p.data_ptr(), g.data_ptr(), 0, m.data_ptr(), v.data_ptr(), n,
                                  1e-4, 0.9, 0.999, 1e-8, 0.0, 3, 1.0, sh.data_ptr()))
        print(f"trial {trial} grid {grid} stagger {stagger:6d} floats: {t:6.3f} ms  {30.0 * n / t / 1e9:6.2f} TB/s", flush=True)
        del bufs, p, g, m, v, sh

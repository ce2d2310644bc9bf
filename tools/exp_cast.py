"""cast + column sums (sei_cast_transpose_bf16 plain form, sei_cast_bf16_colsum_weighted) at the step's shapes; run against
library builds with another grid cap through SEI_HIP_LIBRARY."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "scale-equivariant-imaging_amd"))
import _native as N
def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
tot = 0.0
for (R, C) in ((221184, 32), (55296, 128), (13824, 512), (3456, 2048), (864, 8192)):
    x = torch.randn((R, C), device="cuda"); y = torch.empty((R, C), device="cuda", dtype=torch.bfloat16)
    cs = torch.zeros(C, device="cuda"); w = torch.rand(R, device="cuda")
    t1 = timeit(lambda: N.call("sei_cast_transpose_bf16", x.data_ptr(), 0, y.data_ptr(), None, R, C, R, cs.data_ptr()))
    t2 = timeit(lambda: N.call("sei_cast_bf16_colsum_weighted", x.data_ptr(), y.data_ptr(), w.data_ptr(), cs.data_ptr(), R, C))
    tot += t1 + t2
    print(f"{R} x {C}: cast+colsum {t1:6.1f} us   weighted {t2:6.1f} us")
print(f"{os.environ.get('SEI_HIP_LIBRARY', 'default')}: total {tot:.1f} us")

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
_ops.set_compute_dtype("bf16") if hasattr(_ops, "set_compute_dtype") else None
for (B, C, out16, which) in ((96, 2048, False, "bwd_of_up"), (64, 512, True, "down"), (32, 512, True, "down"), (96, 512, False, "bwd_of_up")):
    x = torch.randn((B, 12, 12, C), device="cuda")
    if which == "down":
        fwd, bwd = _mats.resample_matrices("down", 12, 12, 2, x.device); mats = fwd
    else:
        fwd, bwd = _mats.resample_matrices("up", 6, 6, 2, x.device); mats = bwd
    Ho, Wo = mats[0].shape[0], mats[1].shape[0]
    assert (Ho, Wo) == (6, 6), (Ho, Wo)
    L1, R1, L2, R2 = mats[:4]
    y_s = torch.empty((B, 6, 6, C), device="cuda", dtype=torch.bfloat16 if out16 else torch.float32)
    y_m = torch.empty_like(y_s)
    def small(): N.call("sei_sepmap2_small", x.data_ptr(), y_s.data_ptr(), int(out16), B, 12, 12, 6, 6, C, L1.data_ptr(), R1.data_ptr(), L2.data_ptr(), R2.data_ptr())
    pk = _ops._packed16(mats)
    def mfma(): N.call("sei_sepmap2_bf16_out16" if out16 else "sei_sepmap2_bf16", x.data_ptr(), y_m.data_ptr(), B, 12, 12, 6, 6, C, pk.data_ptr())
    small(); mfma(); torch.cuda.synchronize()
    ref = _ops.sepmap2(x, mats, 6, 6)
    es = float((y_s.float() - ref).abs().max() / ref.abs().max()); em = float((y_m.float() - ref).abs().max() / ref.abs().max())
    print(f"B={B} C={C} out16={out16} {which}: small {timeit(small):6.1f} us (err {es:.1e})   mfma {timeit(mfma):6.1f} us (err {em:.1e})", flush=True)

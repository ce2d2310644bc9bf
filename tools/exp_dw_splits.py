"""Stored weight gradients of the 512-channel level on the 128 x 128 loop: K slices (forced through the tuning build) with the
zero fill + atomics (store) and without the zero fill (accumulate) -- what a slab combine could at most remove."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "scale-equivariant-imaging_amd"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _tuning; lib = _tuning.use()
import _native as N
def timeit(fn, iters=6):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
g = torch.Generator(device="cuda").manual_seed(5)
for (M, Nn, K1, K2) in ((512, 2048, 9216, 4608), (2048, 512, 9216, 4608), (2048, 512, 2304, 1152)):
    A1, A2 = ((0.05 * torch.randn((k, M), device="cuda", generator=g)).bfloat16() for k in (K1, K2))
    B1, B2 = (torch.randn((k, Nn), device="cuda", generator=g).bfloat16() for k in (K1, K2))
    D = torch.zeros((M, Nn), device="cuda")
    line = f"{M} x {Nn} x ({K1}+{K2}):"
    for acc in (0, 1):
        line += f"\n   {'accumulate (no zero fill)' if acc else 'store (zero fill + atomics)'}:"
        for sk in (0, 1, 2, 3, 4, 6, 8, 12, 16, 24):
            lib.sei_debug_set_nt_tile(1000 + sk)
            t = timeit(lambda: N.call("sei_gemm_bf16nt_dw2", A1.data_ptr(), A2.data_ptr(), M, B1.data_ptr(), B2.data_ptr(), Nn, D.data_ptr(), M, Nn, K1, K2, acc))
            line += f"  S{sk if sk else 'auto'} {t:5.1f}"
    lib.sei_debug_set_nt_tile(1000)
    print(line, flush=True)

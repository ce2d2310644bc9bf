import os, sys, torch
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "scale-equivariant-imaging_amd"))
import _native as N
def timeit(fn, iters=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for (M, Nn, K1, K2) in ((8192, 32768, 9216, 4608), (2048, 8192, 36864, 18432), (512, 2048, 147456, 73728), (8192, 32768, 576, 288)):
    g = torch.Generator(device="cuda").manual_seed(1)
    A1 = (0.05 * torch.randn((K1, M), device="cuda", generator=g)).bfloat16(); A2 = (0.05 * torch.randn((K2, M), device="cuda", generator=g)).bfloat16()
    B1 = torch.randn((K1, Nn), device="cuda", generator=g).bfloat16(); B2 = torch.randn((K2, Nn), device="cuda", generator=g).bfloat16()
    D = torch.empty((M, Nn), device="cuda")
    fl = 2.0 * M * Nn * (K1 + K2)
    out = []
    for tile in (0, 30, 33, 16):
        try:
            t = timeit(lambda: N.call("sei_gemm_bf16nt_dw2_ex", A1.data_ptr(), A2.data_ptr(), M, B1.data_ptr(), B2.data_ptr(), Nn, D.data_ptr(), M, Nn, K1, K2, 0, tile))
            out.append(f"tile {tile}: {t:8.0f} us {fl / t / 1e6:5.0f} TF")
        except Exception as e:
            out.append(f"tile {tile}: {type(e).__name__}")
    print(f"{M}x{Nn}x({K1}+{K2}): " + "   ".join(out), flush=True)

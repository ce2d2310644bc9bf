import os, sys, torch
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "scale-equivariant-imaging_amd"))
import _native as N
def timeit(fn, iters=6):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
ws_bytes = 768 << 20
ws = torch.zeros(ws_bytes, dtype=torch.uint8, device="cuda")
g = torch.Generator(device="cuda").manual_seed(3)
for (M, Nn, K, brm, epi) in ((288, 8192, 32768, 0, 3), (576, 8192, 32768, 0, 3), (864, 8192, 32768, 1, 0)):
    A = (torch.randn((M, K), device="cuda", generator=g) * 0.5).bfloat16()
    B = (torch.randn((K, Nn) if brm else (Nn, K), device="cuda", generator=g) * 0.05).bfloat16()
    bias = torch.randn(Nn, device="cuda", generator=g); R1 = torch.randn((M, Nn), device="cuda", generator=g)
    D = torch.empty((M, Nn), device="cuda")
    def call(tile, sk):
        N.call("sei_gemm_bf16nt_ws", A.data_ptr(), K, 0, B.data_ptr(), Nn if brm else K, brm, D.data_ptr(), None, M, Nn, K, epi,
               bias.data_ptr() if epi == 3 else None, R1.data_ptr() if epi == 3 else None, None, None, None, ws.data_ptr(), ws_bytes, tile, 0, sk)
    line = f"{M} x {Nn} x {K}: auto {timeit(lambda: call(0, 0)):6.1f}"
    for tile in (31, 32):
        line += f"\n   tile {tile}:"
        for sk in (2, 3, 4, 5, 6, 7, 8, 10, 12, 16):
            line += f"  S{sk} {timeit(lambda: call(tile, sk)):6.1f}"
    print(line, flush=True)

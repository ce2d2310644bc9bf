"""Band width of the quadrant kernel's tile order (sei_gemm_bf16nt_ws, band argument) on the bottleneck-level GEMMs, whose
weight (537 MB) streams from HBM: with band b an XCD's contiguous range of the order covers b tile columns for every tile
row before it moves on, so with band 1 all row tiles of a column tile sit on ONE XCD and read its weight slab through one
L2. Times per (shape, band); tile and slice count automatic."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "scale-equivariant-imaging_amd"))
import _native as N
def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
ws_bytes = 256 << 20
ws = torch.zeros(ws_bytes, dtype=torch.uint8, device="cuda")
g = torch.Generator(device="cuda").manual_seed(3)
for (M, Nn, K, brm, kind) in ((576, 8192, 32768, 0, "res"), (864, 8192, 32768, 1, "none32"), (288, 8192, 32768, 0, "res"),
                              (576, 32768, 8192, 0, "gelu"), (576, 32768, 8192, 1, "dgelu"), (288, 32768, 8192, 0, "gelu"),
                              (2304, 2048, 8192, 0, "res"), (3456, 2048, 8192, 1, "none32"), (2304, 8192, 2048, 0, "gelu"),
                              (1152, 8192, 2048, 0, "gelu"), (1152, 2048, 8192, 0, "res")):
    A = (torch.randn((M, K), device="cuda", generator=g) * 0.5).bfloat16()
    B = (torch.randn((K, Nn) if brm else (Nn, K), device="cuda", generator=g) * 0.05).bfloat16()
    bias = torch.randn(Nn, device="cuda", generator=g); R1 = torch.randn((M, Nn), device="cuda", generator=g)
    o32 = torch.empty((M, Nn), device="cuda"); o16 = torch.empty((M, Nn), device="cuda", dtype=torch.bfloat16)
    d2 = torch.empty((M, Nn), device="cuda", dtype=torch.bfloat16); cs = torch.zeros(Nn, device="cuda")
    epi = {"res": 3, "none32": 0, "gelu": 2, "dgelu": 4}[kind]
    D32 = None if kind == "dgelu" else o32
    D16 = o16 if kind == "dgelu" else None
    def call(band):
        N.call("sei_gemm_bf16nt_ws", A.data_ptr(), K, 0, B.data_ptr(), Nn if brm else K, brm, N.ptr(D32), N.ptr(D16), M, Nn, K, epi,
               bias.data_ptr() if kind in ("res", "gelu") else None, R1.data_ptr() if kind in ("res", "dgelu") else None, None,
               d2.data_ptr() if kind == "gelu" else None, cs.data_ptr() if kind == "dgelu" else None, ws.data_ptr(), ws_bytes, 0, band, 0)
    ref = None
    line = f"{M} x {Nn} x {K} brm={brm} {kind}:"
    for band in (0, 1, 2, 3, 4, 8, 16):
        call(band); torch.cuda.synchronize()
        out = (D32 if D32 is not None else D16).float().clone()
        if ref is None: ref = out
        ok = float((out - ref).abs().max()) <= 1e-2 * float(ref.abs().max())
        line += f"  b{band} {timeit(lambda: call(band)):6.1f}{'' if ok else '(!)'}"
    print(line, flush=True)

"""Weight-gradient GEMMs of the two deepest levels (both operands reduction-major, two K segments, float32 store): the
128 x 128 loop (automatic choice) against the quadrant schedule's 256 x 256 / 256 x 128 tiles (tile codes 30 / 33 of
sei_gemm_bf16nt_dw2_ex), after round 5's inline-asm LDS-DMA (the quadrant kernel's reduction-major variants had been
measured with the compiler's vmcnt(0) in front of every transposing read). Results checked against a float32 matmul."""
import os, sys, statistics, torch
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "scale-equivariant-imaging_amd"))
from _native import call


def once(fn, iters=5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for (M, N, K1, K2) in ((2048, 8192, 1152, 2304), (8192, 2048, 1152, 2304), (8192, 32768, 288, 576), (32768, 8192, 288, 576)):
    A1 = torch.randn((K1, M), device="cuda").bfloat16()
    A2 = torch.randn((K2, M), device="cuda").bfloat16()
    B1 = torch.randn((K1, N), device="cuda").bfloat16()
    B2 = torch.randn((K2, N), device="cuda").bfloat16()
    D = torch.empty((M, N), device="cuda")
    f = lambda code: call("sei_gemm_bf16nt_dw2_ex", A1.data_ptr(), A2.data_ptr(), M, B1.data_ptr(), B2.data_ptr(), N,
                          D.data_ptr(), M, N, K1, K2, 0, code)
    ref = (A1[:, :512].float().t() @ B1[:, :512].float()) + (A2[:, :512].float().t() @ B2[:, :512].float())
    times = {0: [], 30: [], 33: []}
    for code in times:
        D.zero_()
        f(code)
        torch.cuda.synchronize()
        err = float((D[:512, :512] - ref).abs().max() / ref.abs().max())
        assert err < 1e-5, (M, N, code, err)
    for rnd in range(5):
        for code in times:
            f(code)
            torch.cuda.synchronize()
            times[code].append(once(lambda: f(code)))
    fl = 2.0 * M * N * (K1 + K2)
    print(f"{M}x{N}x({K1}+{K2}): " + "  ".join(
        f"code {c} {statistics.median(t):.0f}us/{fl / statistics.median(t) / 1e6:.0f}TF/{4.0 * M * N / statistics.median(t) / 1e6:.2f}TB/s-out"
        for c, t in times.items()), flush=True)

"""Quadrant schedule, 128-column tiles: THREE LDS stages (tile codes 32 / 33, round 5) against two (codes 38 / 39, round 4's
kernel) on the skinny weight-streaming launches of the deep levels, with the epilogues the model uses. Product library
(sei_gemm_bf16nt_ex takes the tile code per call); interleaved rounds in one process, median times; results checked
against a float32 matmul of the same bf16 operands first (short reductions included: 1, 2, 3, 4 k-tiles)."""
import os, sys, statistics, torch
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "scale-equivariant-imaging_amd"))
from models import _ops


def once(fn, iters=5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def build(M, N, K, kind):
    kr = kind.endswith("_kr")
    A = torch.randn((M, K), device="cuda").bfloat16()
    B = (torch.randn((K, N) if kr else (N, K), device="cuda") / K ** 0.5).bfloat16()
    out = torch.empty((M, N), device="cuda")
    o16 = torch.empty((M, N), device="cuda", dtype=torch.bfloat16)
    bias = torch.randn(N, device="cuda")
    R1 = torch.randn((M, N), device="cuda")
    if kind == "none_kr":
        f = lambda tile: _ops.gemm_nt16(A, B, M, N, K, _ops.EPI_NONE, out32=out, b_rmajor=True, tile=tile)
        ref = lambda: A.float() @ B.float()
        got = lambda: out
    elif kind == "dgelu_kr":
        f = lambda tile: _ops.gemm_nt16(A, B, M, N, K, _ops.EPI_MUL_DGELU, out16=o16, R1=R1, b_rmajor=True, tile=tile)
        def ref():
            x = R1.double()
            phi = 0.5 * (1 + torch.erf(x / 2 ** 0.5))
            pdf = torch.exp(-0.5 * x * x) / (2 * torch.pi) ** 0.5
            return ((A.float() @ B.float()).double() * (phi + x * pdf)).float()
        got = lambda: o16.float()
    elif kind == "gelu":
        f = lambda tile: _ops.gemm_nt16(A, B, M, N, K, _ops.EPI_BIAS_GELU, out32=out, bias=bias, D2_16=o16, tile=tile)
        ref = lambda: A.float() @ B.float().t() + bias
        got = lambda: out
    else:
        f = lambda tile: _ops.gemm_nt16(A, B, M, N, K, _ops.EPI_BIAS_RES, out32=out, bias=bias, R1=R1, tile=tile)
        ref = lambda: A.float() @ B.float().t() + bias + R1
        got = lambda: out
    return f, ref, got


def check(M, N, K, kind, codes):
    f, ref, got = build(M, N, K, kind)
    r = ref()
    for code in codes:
        got().zero_()
        f(code)
        torch.cuda.synchronize()
        err = float((got() - r).abs().max() / r.abs().max())
        bar = 2e-2 if kind == "dgelu_kr" else 2e-5 * max(1, K // 512) + (1e-3 if False else 0)
        status = "ok" if err < bar else "WRONG"
        print(f"check {M}x{N}x{K} {kind} tile {code}: rel err {err:.2e} {status}", flush=True)
        assert err < bar, (M, N, K, kind, code, err)


if __name__ == "__main__":
    NAMES = {0: "auto", 31: "288x256", 32: "288x128/3st", 38: "288x128/2st", 33: "256x128/3st", 39: "256x128/2st"}
    for K in (64, 128, 192, 256, 448, 2048):
        for kind in ("gelu", "dgelu_kr", "res", "none_kr"):
            check(576, 1024, K, kind, (32, 38))
        check(512, 1024, K, "gelu", (33, 39))
        check(512, 1024, K, "none_kr", (33, 39))
    check(288, 8192, 32768, "res", (32,))
    shapes = [(288, 32768, 8192, "gelu"), (288, 32768, 8192, "dgelu_kr"), (288, 8192, 32768, "res"), (288, 8192, 32768, "none_kr"),
              (1152, 8192, 2048, "gelu"), (1152, 8192, 2048, "dgelu_kr"), (1152, 8192, 2048, "none_kr"),
              (2304, 2048, 8192, "res"), (2304, 2048, 8192, "none_kr"), (1152, 2048, 8192, "res"), (1152, 2048, 8192, "none_kr"),
              (576, 32768, 8192, "gelu"), (576, 8192, 32768, "res"), (36864, 512, 128, "gelu"), (9216, 2048, 512, "gelu")]
    for (M, N, K, kind) in shapes:
        f, _, _ = build(M, N, K, kind)
        codes = [32, 38, 0]
        times = {c: [] for c in codes}
        for rnd in range(5):
            for code in codes:
                f(code)
                torch.cuda.synchronize()
                times[code].append(once(lambda: f(code)))
        print(f"{M}x{N}x{K} {kind}: " + "  ".join(
            f"{NAMES[c]} {statistics.median(t):.0f}us/{2.0 * M * N * K / statistics.median(t) / 1e6:.0f}TF" for c, t in times.items()),
            flush=True)

"""Split-K through slabs (sei_gemm_bf16nt_ws) against the launches the step makes today (float atomics or unsplit), shape by
shape: results against a float64 product of the same bf16 operands, then times for tile x slice-count choices.

    python tools/exp_splitk_slabs.py [--quick]
Shapes: the deep-level forward / data-gradient GEMMs of configs[1] at batch 32 (tools/step_timeline.py)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "scale-equivariant-imaging_amd"))
import _native as N  # noqa: E402

EPI_NONE, EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_RES, EPI_MUL_DGELU = 0, 1, 2, 3, 4


def epi_codes():
    import re
    hdr = open(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "include", "sei_hip.h")).read()
    return {m.group(1): int(m.group(2)) for m in re.finditer(r"#define (SEI_EPI_\w+)\s+(\d+)", hdr)}


def timeit(fn, iters=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    o = ap.parse_args()
    E = epi_codes()
    dev = "cuda"
    ws_bytes = 512 << 20
    ws = torch.zeros(ws_bytes, dtype=torch.uint8, device=dev)
    g = torch.Generator(device=dev).manual_seed(3)
    # (M, N, K, b_rmajor, kind): kind "res" = f32 out with bias + residual (contracting conv3), "none32" = f32 out (gh2),
    # "gelu" = bias + GELU with f32 + bf16 outputs (expanding conv2), "dgelu" = bf16 out x gelu'(h3) + column sums (gh3)
    shapes = [(2304, 2048, 8192, 0, "res"), (1152, 2048, 8192, 0, "res"), (3456, 2048, 8192, 1, "none32"),
              (576, 8192, 32768, 0, "res"), (288, 8192, 32768, 0, "res"), (864, 8192, 32768, 1, "none32"),
              (288, 32768, 8192, 0, "gelu"), (288, 32768, 8192, 1, "dgelu"), (576, 32768, 8192, 0, "gelu"),
              (864, 2048, 8192, 1, "none32"), (576, 8192, 2048, 0, "res"), (288, 8192, 2048, 0, "res"),
              (9216, 512, 2048, 0, "res"), (4608, 512, 2048, 0, "res"), (13824, 512, 2048, 1, "none32")]
    if o.quick:
        shapes = shapes[:2] + shapes[6:8]
    for (M, Nn, K, brm, kind) in shapes:
        A = (torch.randn((M, K), device=dev, generator=g) * 0.5).bfloat16()
        B = (torch.randn((K, Nn) if brm else (Nn, K), device=dev, generator=g) * 0.05).bfloat16()
        bias = torch.randn(Nn, device=dev, generator=g)
        R1 = torch.randn((M, Nn), device=dev, generator=g)
        out32 = torch.empty((M, Nn), device=dev)
        out16 = torch.empty((M, Nn), device=dev, dtype=torch.bfloat16)
        d2 = torch.empty((M, Nn), device=dev, dtype=torch.bfloat16)
        colsum = torch.zeros(Nn, device=dev)
        ref = A.double() @ (B.double() if brm else B.double().t())
        if kind == "res":
            epi, D32, D16, D2, cs, want = E["SEI_EPI_BIAS_RES"], out32, None, None, None, ref + bias.double() + R1.double()
        elif kind == "none32":
            epi, D32, D16, D2, cs, want = E["SEI_EPI_NONE"], out32, None, None, None, ref
        elif kind == "gelu":
            epi, D32, D16, D2, cs = E["SEI_EPI_BIAS_GELU"], out32, None, d2, None
            want = ref + bias.double()
        else:
            epi, D32, D16, D2, cs = E["SEI_EPI_MUL_DGELU"], None, out16, None, colsum
            x = R1.double()
            cdf = 0.5 * (1 + torch.erf(x / 2 ** 0.5))
            pdf = torch.exp(-0.5 * x * x) / (2 * torch.pi) ** 0.5
            want = ref * (cdf + x * pdf)

        def call(ws_t, tile, sk, band=0):
            N.call("sei_gemm_bf16nt_ws", A.data_ptr(), K, 0, B.data_ptr(), Nn if brm else K, brm, N.ptr(D32), N.ptr(D16), M, Nn, K,
                   epi, bias.data_ptr() if kind in ("res", "gelu") else None, R1.data_ptr() if kind in ("res", "dgelu") else None,
                   None, N.ptr(D2), N.ptr(cs), None if ws_t is None else ws_t.data_ptr(), 0 if ws_t is None else ws_bytes, tile, band, sk)

        def err():
            got = (D32 if D32 is not None else D16).double()
            return float((got - want).abs().max() / want.abs().max())

        def plan(wsb):
            code = N.lib().sei_gemm_bf16nt_plan_ws(0, brm, int(D32 is not None), int(D16 is not None), M, Nn, K, epi, wsb)
            return f"{ {1: 'nt', 2: 'pq'}.get(code >> 48)} {(code >> 32) & 0xFFFF}x{(code >> 16) & 0xFFFF} S={code & 0x7FFF}{'slab' if code & 0x8000 else ''}"

        call(None, 0, 0)
        torch.cuda.synchronize()
        e0 = err()
        t0 = timeit(lambda: call(None, 0, 0))
        print(f"{M} x {Nn} x {K} brm={brm} {kind}: today [{plan(0)}] {t0:7.1f} us (err {e0:.1e}); auto with ws [{plan(ws_bytes)}]", flush=True)
        rows = []
        for tile in ((31, 32) if M % 288 == 0 else (30, 33)):
            for sk in (1, 2, 3, 4, 6, 8):
                if K // 64 < 4 * sk:
                    continue
                if cs is not None:
                    colsum.zero_()
                (D32 if D32 is not None else D16).fill_(7.0)
                try:
                    call(ws, tile, sk)
                except N.NativeLibraryError as ex:
                    rows.append(f"   tile {tile} S={sk}: {ex}")
                    continue
                torch.cuda.synchronize()
                e = err()
                cs_err = ""
                if cs is not None:
                    cs_want = out16.double().sum(0)
                    cs_err = f" colsum err {float((colsum.double() - cs_want).abs().max() / cs_want.abs().max()):.1e}"
                call(ws, tile, sk)                                # a second launch on the same counters
                torch.cuda.synchronize()
                e2 = err()
                t = timeit(lambda: call(ws, tile, sk))
                flag = "" if max(e, e2) < 2e-2 else "   <-- WRONG"
                rows.append(f"   tile {tile} S={sk}: {t:7.1f} us  err {e:.1e} / {e2:.1e}{cs_err}{flag}")
        print("\n".join(rows), flush=True)
        assert int(ws[:16384].view(torch.int32).abs().sum()) == 0, "tile counters not back to zero"
    print("done")


if __name__ == "__main__":
    main()

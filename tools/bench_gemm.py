#!/usr/bin/env python3
"""Per-shape GEMM timing on the U-Net's actual shapes (batch 32): sei_gemm_f32 / sei_gemm_bf16 vs
torch.matmul (rocBLAS / hipBLASLt) as the known-good reference on the same device.
    python tools/bench_gemm.py [--batch 32] [--fused]      (run on the GPU box)
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scale-equivariant-imaging_amd"))
from models import _ops  # noqa: E402


def timeit(fn, iters=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--rows-mult", type=int, default=2, help="2 = the fused 2B pass, 1 = the B pass")
    opt = ap.parse_args()
    shapes = []
    for s in range(5):
        C = 32 * 4**s
        M = opt.batch * opt.rows_mult * 2304 // 4**s
        shapes += [(f"s{s} conv2 fwd NT", M, 4 * C, C, 0, 1), (f"s{s} conv3 fwd NT", M, C, 4 * C, 0, 1),
                   (f"s{s} conv3 dX  NN", M, 4 * C, C, 0, 0), (f"s{s} conv2 dX  NN", M, C, 4 * C, 0, 0),
                   (f"s{s} conv3 dW  TN", C, 4 * C, M, 1, 0), (f"s{s} conv2 dW  TN", 4 * C, C, M, 1, 0)]
    print(f"{'shape':18s} {'M':>7s} {'N':>6s} {'K':>6s} | {'f32 TF':>7s} {'bf16 TF':>8s} | {'rocblas f32':>11s} {'rocblas bf16':>12s} | us f32/bf16/rb16")
    tot = {"f32": 0.0, "bf16": 0.0, "rb32": 0.0, "rb16": 0.0, "nt16": 0.0}
    for name, M, N, K, ta, tb in shapes:
        A = torch.randn((K, M) if ta else (M, K), device="cuda")
        B = torch.randn((N, K) if tb else (K, N), device="cuda")
        out = torch.zeros((M, N), device="cuda")
        epi = _ops.EPI_ACCUM if ta else _ops.EPI_NONE
        res = {}
        for dt in ("f32", "bf16"):
            _ops.set_compute_dtype(dt)
            res[dt] = timeit(lambda: _ops.gemm(A, B, M, N, K, ta, tb, epi, out=out))
        _ops.set_compute_dtype("f32")
        Am, Bm = (A.t() if ta else A), (B.t() if tb else B)
        res["rb32"] = timeit(lambda: torch.matmul(Am, Bm, out=out))
        A16, B16 = Am.bfloat16(), Bm.bfloat16()
        o16 = torch.empty((M, N), device="cuda", dtype=torch.bfloat16)
        res["rb16"] = timeit(lambda: torch.matmul(A16, B16, out=o16))
        res["nt16"] = float("nan")
        if K % 8 == 0 and M % 8 == 0 and N % 8 == 0:      # operands as the model stores them (no transposes)
            A16, B16 = A.bfloat16(), B.bfloat16()
            epi16 = _ops.EPI_ACCUM if ta else _ops.EPI_NONE
            res["nt16"] = timeit(lambda: _ops.gemm_nt16(A16, B16, M, N, K, epi16, out32=out, a_rmajor=bool(ta),
                                                        b_rmajor=not tb))
        fl = 2.0 * M * N * K
        for k in tot:
            tot[k] += res[k] if res[k] == res[k] else res["bf16"]
        print(f"{name:18s} {M:7d} {N:6d} {K:6d} | {fl / res['f32'] / 1e12:7.1f} {fl / res['bf16'] / 1e12:8.1f} | "
              f"{fl / res['rb32'] / 1e12:11.1f} {fl / res['rb16'] / 1e12:12.1f} | "
              f"{res['f32'] * 1e6:8.0f} {res['bf16'] * 1e6:8.0f} {res['rb16'] * 1e6:8.0f} | nt16 {fl / res['nt16'] / 1e12:7.1f} TF {res['nt16'] * 1e6:7.0f} us")
    print("total ms:", {k: round(v * 1e3, 2) for k, v in tot.items()})


if __name__ == "__main__":
    main()

"""sei_rowgemm_bf16 (row-streaming GEMM with the layer's matrix in registers, token_gemm.hip) against sei_gemm_bf16nt on
the eight linear-layer GEMMs of a Swin block (forward and data gradient), at the bench's token counts."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "scale-equivariant-imaging_amd"))
import _native as N
def timeit(fn, iters=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
Ms = [147456, 73728] if len(sys.argv) < 2 else [int(a) for a in sys.argv[1:]]
EPI = dict(NONE=0, BIAS=1, GELU=2, RES=3, DGELU=4, SCALE_RES=7)
# name, N, K, nv, epilogue, out16
cases = [("qkv fwd", 576, 192, 576, "BIAS", 1), ("proj fwd", 192, 192, 180, "RES", 0), ("proj fwd drop", 192, 192, 180, "SCALE_RES", 0),
         ("fc1 fwd", 384, 192, 384, "GELU", 0), ("fc2 fwd", 192, 384, 180, "RES", 0), ("fc2 fwd drop", 192, 384, 180, "SCALE_RES", 0),
         ("fc2 dgrad", 384, 192, 384, "DGELU", 1), ("fc1 dgrad", 192, 384, 192, "NONE", 0), ("proj dgrad", 192, 192, 192, "NONE", 1),
         ("qkv dgrad", 192, 576, 192, "NONE", 0)]
g = torch.Generator(device="cuda").manual_seed(1)
for M in Ms:
    tot_old = tot_new = 0.0
    for name, Nn, K, nv, epi, out16 in cases:
        a = torch.randn((M, K), device="cuda", generator=g).bfloat16()
        w = (0.1 * torch.randn((Nn, K), device="cuda", generator=g)).bfloat16()
        w[nv:] = 0
        bias = torch.randn(nv, device="cuda", generator=g)
        res = torch.randn((M, nv), device="cuda", generator=g)
        drop = (torch.rand(M, device="cuda", generator=g) > 0.1).float() / 0.9
        e = EPI[epi]
        def outs():
            d32 = torch.zeros((M, nv), device="cuda") if (not out16 or epi == "GELU") and epi != "DGELU" else None
            d16 = torch.zeros((M, Nn), device="cuda", dtype=torch.bfloat16) if out16 or epi in ("GELU", "DGELU") else None
            return d32, d16
        n32, n16 = outs(); o32, o16 = outs()
        R1 = drop if epi == "SCALE_RES" else (res if epi in ("RES", "DGELU") else None)
        R2 = res if epi == "SCALE_RES" else None
        hasb = epi in ("BIAS", "GELU", "RES", "SCALE_RES")
        new = lambda: N.call("sei_rowgemm_bf16", a.data_ptr(), K, w.data_ptr(), K, N.ptr(n32), nv, N.ptr(n16), Nn, M, Nn, K, nv, e,
                             bias.data_ptr() if hasb else None, N.ptr(R1), N.ptr(R2), nv)
        # the tiled kernel writes N = nv columns with ld = nv: bf16 outputs of the padded width only when nv == N
        def old():
            if epi == "GELU":
                N.call("sei_gemm_bf16nt", a.data_ptr(), K, 0, w.data_ptr(), K, 0, o32.data_ptr(), None, M, nv, K, e, bias.data_ptr(), None, None, o16.data_ptr())
            else:
                N.call("sei_gemm_bf16nt", a.data_ptr(), K, 0, w.data_ptr(), K, 0, N.ptr(o32), N.ptr(o16) if nv == Nn else None, M, nv, K, e,
                       bias.data_ptr() if hasb else None, N.ptr(R1), N.ptr(R2), None)
        new(); old(); torch.cuda.synchronize()
        acc = a.float() @ w.float().T
        if hasb: acc[:, :nv] += bias
        if epi == "RES": ref32 = acc[:, :nv] + res
        elif epi == "SCALE_RES": ref32 = res + drop[:, None] * acc[:, :nv]
        elif epi == "DGELU":
            x = res.double(); cdf = 0.5 * (1 + torch.erf(x / 2 ** 0.5)); pdf = torch.exp(-0.5 * x * x) / (2 * torch.pi) ** 0.5
            ref32 = (acc[:, :nv].double() * (cdf + x * pdf)).float()
        else: ref32 = acc[:, :nv]
        errs = []
        if n32 is not None: errs.append(((n32 - ref32).abs().max() / ref32.abs().max()).item())
        if n16 is not None:
            r16 = torch.nn.functional.gelu(ref32) if epi == "GELU" else ref32
            errs.append(((n16[:, :nv].float() - r16).abs().max() / r16.abs().max()).item())
            if nv < Nn: errs.append(n16[:, nv:].float().abs().max().item())
        if o32 is not None and n32 is not None: errs.append(((n32 - o32).abs().max() / ref32.abs().max()).item())
        byt = M * K * 2 + (M * nv * 4 if n32 is not None else 0) + (M * Nn * 2 if n16 is not None else 0) + (M * nv * 4 if R1 is not None and epi != "SCALE_RES" or R2 is not None else 0)
        t0 = timeit(old); t1 = timeit(new)
        tot_old += t0; tot_new += t1
        print(f"M={M} {name:14s} N={Nn} K={K}: {byt/1e6:6.1f} MB  nt {t0:6.1f} us ({byt/t0/1e6:.2f} TB/s)  rowgemm {t1:6.1f} us ({byt/t1/1e6:.2f} TB/s)  err " + " ".join(f"{x:.1e}" for x in errs), flush=True)
    print(f"M={M}: sum nt {tot_old:.0f} us, rowgemm {tot_new:.0f} us")

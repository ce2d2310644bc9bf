"""fc1's forward of a Swin block (sei_rowgemm_gelu_bf16: bias + GELU, bf16 output with a ones column; bias + GELU in the
accumulator layout, two workgroups per CU) against the float32-patch form of the same product (sei_rowgemm_bf16 with
SEI_EPI_BIAS_GELU and a float32 pre-activation output), both checked against float64 on the same bf16 operands."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "scale-equivariant-imaging_amd"))
import _native as N


def timeit(fn, iters=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


g = torch.Generator(device="cuda").manual_seed(1)
K, Nn, one_at = 192, 384, 360
for M in (147456, 73728):
    a = torch.randn((M, K), device="cuda", generator=g).bfloat16()
    w = (0.1 * torch.randn((Nn, K), device="cuda", generator=g)).bfloat16()
    w[360:] = 0
    bias = torch.randn(Nn, device="cuda", generator=g)
    bias[360:] = 0
    outs = {}
    for mode in ("two workgroups per CU", "float32 patch"):
        d16 = torch.zeros((M, Nn), device="cuda", dtype=torch.bfloat16)
        d32 = torch.zeros((M, Nn), device="cuda")
        if mode == "float32 patch":
            fn = lambda: N.call("sei_rowgemm_bf16", a.data_ptr(), K, w.data_ptr(), K, d32.data_ptr(), Nn, d16.data_ptr(), Nn, M, Nn, K, Nn,
                                2, bias.data_ptr(), None, None, Nn)
        else:
            fn = lambda: N.call("sei_rowgemm_gelu_bf16", a.data_ptr(), K, w.data_ptr(), K, bias.data_ptr(), Nn, d16.data_ptr(), Nn, M, Nn,
                                K, one_at)
        t = timeit(fn)
        outs[mode] = d16
        ref = torch.nn.functional.gelu((a[:4096].double() @ w.double().T) + bias.double())
        if mode != "float32 patch":
            ref[:, one_at] = 1.0
        err = ((d16[:4096].double() - ref).abs().max() / ref.abs().max()).item()
        print(f"M={M} {mode:22s}: {t:6.1f} us  ({(M * K * 2 + M * Nn * 2) / t / 1e6:.2f} TB/s)  err {err:.1e}", flush=True)
    same = outs["two workgroups per CU"].clone()
    same[:, one_at] = outs["float32 patch"][:, one_at]
    print("   bit-identical (ones column apart):", torch.equal(same, outs["float32 patch"]))

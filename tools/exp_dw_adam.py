"""Bottleneck weight gradient (8192 x 32768, K = 288 + 576): stored gradient + separate Adam over the same range
against the GEMM whose epilogue applies Adam (sei_gemm_bf16nt_dw2_adam).  Same results, times of both."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "scale-equivariant-imaging_amd"))
import _native as N
def timeit(fn, iters=6):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for (M, Nn, K1, K2) in ((8192, 32768, 288, 576), (32768, 8192, 288, 576), (2048, 8192, 1152, 2304)):
    g = torch.Generator(device="cuda").manual_seed(1)
    A1 = (0.05 * torch.randn((K1, M), device="cuda", generator=g)).bfloat16(); A2 = (0.05 * torch.randn((K2, M), device="cuda", generator=g)).bfloat16()
    B1 = torch.randn((K1, Nn), device="cuda", generator=g).bfloat16(); B2 = torch.randn((K2, Nn), device="cuda", generator=g).bfloat16()
    p0 = 0.02 * torch.randn((M, Nn), device="cuda", generator=g)
    host = (ctypes.c_float * 6)()
    N.call("sei_adam_scalars", 1e-4, 0.9, 0.999, 1e-8, 0.0, 3, ctypes.cast(host, ctypes.c_void_p))
    hyper = torch.tensor(list(host), device="cuda")
    def fresh():
        return p0.clone(), torch.full_like(p0, 1e-3), torch.full_like(p0, 1e-5), torch.empty((M, Nn), device="cuda", dtype=torch.bfloat16)
    pa, ma, va, sa = fresh(); grad = torch.empty((M, Nn), device="cuda")
    def separate():
        N.call("sei_gemm_bf16nt_dw2", A1.data_ptr(), A2.data_ptr(), M, B1.data_ptr(), B2.data_ptr(), Nn, grad.data_ptr(), M, Nn, K1, K2, 0)
        N.call("sei_adam_fused", pa.data_ptr(), grad.data_ptr(), 0, ma.data_ptr(), va.data_ptr(), M * Nn, 1e-4, 0.9, 0.999, 1e-8, 0.0, 3, 1.0, sa.data_ptr())
    pb, mb, vb, sb = fresh()
    def fused():
        N.call("sei_gemm_bf16nt_dw2_adam", A1.data_ptr(), A2.data_ptr(), M, B1.data_ptr(), B2.data_ptr(), Nn, pb.data_ptr(), mb.data_ptr(), vb.data_ptr(), sb.data_ptr(), hyper.data_ptr(), M, Nn, K1, K2)
    separate(); fused(); torch.cuda.synchronize()
    same = all(torch.equal(a, b) for a, b in ((pa, pb), (ma, mb), (va, vb), (sa, sb)))
    moved = float((pa - p0).abs().max())
    tg = timeit(lambda: N.call("sei_gemm_bf16nt_dw2", A1.data_ptr(), A2.data_ptr(), M, B1.data_ptr(), B2.data_ptr(), Nn, grad.data_ptr(), M, Nn, K1, K2, 0))
    o16 = torch.empty((M, Nn), device="cuda", dtype=torch.bfloat16)
    t16 = timeit(lambda: N.call("sei_gemm_bf16nt_dw2_bf16out", A1.data_ptr(), A2.data_ptr(), M, B1.data_ptr(), B2.data_ptr(), Nn, o16.data_ptr(), M, Nn, K1, K2))
    tc = timeit(lambda: N.call("sei_cast_bf16", grad.data_ptr(), o16.data_ptr(), M * Nn))
    print(f"    bf16-output GEMM {t16:6.0f} us (f32 store {tg:6.0f} us + cast pass {tc:6.0f} us)")
    ts, tf = timeit(separate), timeit(fused)
    print(f"{M}x{Nn}x({K1}+{K2}): bit-identical {same} (max |dp| {moved:.2e})  GEMM alone {tg:6.0f} us  GEMM + Adam {ts:6.0f} us  fused {tf:6.0f} us", flush=True)

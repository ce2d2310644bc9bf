"""The weight-gradient GEMMs whose epilogue applies the Adam step (sei_gemm_bf16nt_dw2_adam_ex): the 128 x 128 loop (tile 1)
against the quadrant schedule's 256 x 256 / 256 x 128 tiles (30 / 33) on the shapes of the two deepest levels. Every
variant steps the SAME state once first and the results are compared (parameter, both moments, bf16 shadow), then the
launches are timed (they keep stepping their own copies: timing only)."""
import os, sys, statistics, torch
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "scale-equivariant-imaging_amd"))
import _native
from _native import call


def once(fn, iters=3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


hyper = torch.empty(6, device="cuda")
rc = _native.lib().sei_adam_scalars_to_device(1e-4, 0.9, 0.999, 1e-8, 0.0, 1, hyper.data_ptr(), _native.stream())
assert rc == 0
for (M, N, K1, K2) in ((8192, 32768, 288, 576), (32768, 8192, 288, 576), (2048, 8192, 1152, 2304), (8192, 2048, 1152, 2304)):
    A1 = (torch.randn((K1, M), device="cuda") * 0.05).bfloat16()
    A2 = (torch.randn((K2, M), device="cuda") * 0.05).bfloat16()
    B1 = torch.randn((K1, N), device="cuda").bfloat16()
    B2 = torch.randn((K2, N), device="cuda").bfloat16()
    p0 = torch.randn((M, N), device="cuda") * 0.02
    m0 = torch.randn((M, N), device="cuda") * 1e-3
    v0 = torch.rand((M, N), device="cuda") * 1e-4
    state = {}
    for code in (1, 30, 33):
        p, m, v = p0.clone(), m0.clone(), v0.clone()
        p16 = torch.zeros((M, N), device="cuda", dtype=torch.bfloat16)
        f = lambda p=p, m=m, v=v, p16=p16, code=code: call(
            "sei_gemm_bf16nt_dw2_adam_ex", A1.data_ptr(), A2.data_ptr(), M, B1.data_ptr(), B2.data_ptr(), N, p.data_ptr(),
            m.data_ptr(), v.data_ptr(), p16.data_ptr(), hyper.data_ptr(), M, N, K1, K2, code)
        f()
        torch.cuda.synchronize()
        state[code] = (p, m, v, p16, f)
    ref = state[1]
    for code in (30, 33):
        got = state[code]
        for name, a, b in zip(("param", "exp_avg", "exp_avg_sq"), got[:3], ref[:3]):
            err = float((a - b).abs().max() / b.abs().max())
            assert err < 2e-5, (M, N, code, name, err)
        assert float((got[3].float() != ref[3].float()).float().mean()) < 1e-3       # bf16 ties on the last bit of the sum
        assert torch.equal(got[3], got[0].bfloat16())
    # against torch: the step of element block [:256, :256]
    g = (A1[:, :256].float().t() @ B1[:, :256].float()) + (A2[:, :256].float().t() @ B2[:, :256].float())
    m_ref = m0[:256, :256] + (g - m0[:256, :256]) * (1 - 0.9)
    assert float((ref[1][:256, :256] - m_ref).abs().max() / m_ref.abs().max()) < 1e-4
    times = {1: [], 30: [], 33: []}
    for rnd in range(5):
        for code in times:
            times[code].append(once(state[code][4]))
    nbytes = M * N * 26 + 2 * (K1 + K2) * (M + N)
    print(f"{M}x{N}x({K1}+{K2}): " + "  ".join(
        f"tile {c} {statistics.median(t):.0f}us/{nbytes / statistics.median(t) / 1e6:.2f}TB/s/{2.0 * M * N * (K1 + K2) / statistics.median(t) / 1e6:.0f}TF"
        for c, t in times.items()), flush=True)
    del state

"""Weight gradient of a 180 -> 180 3x3 convolution on the padded grid (B images of 50 x 50 grid rows, 192 padded
channels): nine accumulating launches (one per tap) against one batched launch (sei_gemm_bf16nt_dw2_taps)."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "scale-equivariant-imaging_amd"))
import _native as N
def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
Wp, C = 50, 192
offs = [(ky - 1) * Wp + (kx - 1) for ky in range(3) for kx in range(3)]
rows_c = (ctypes.c_int * 9)(*offs)
for (B1, B2) in ((64, 32), (64, 0)):
    guard = Wp + 9
    def grid(Bn):
        R = Bn * Wp * Wp; R8 = (R + 7) // 8 * 8
        return R8, (0.1 * torch.randn((R8, C), device="cuda")).bfloat16(), torch.randn((R8 + 2 * guard, C), device="cuda").bfloat16()
    K1, g1, x1 = grid(B1)
    K2, g2, x2 = grid(B2) if B2 else (0, g1, x1)
    out_a = torch.zeros((9, C, C), device="cuda"); out_b = torch.zeros((9, C, C), device="cuda")
    def nine():
        for t, o in enumerate(offs):
            if K2:
                N.call("sei_gemm_bf16nt_dw2", g1.data_ptr(), g2.data_ptr(), C, x1[guard + o:].data_ptr(), x2[guard + o:].data_ptr(), C, out_a[t].data_ptr(), C, C, K1, K2, 1)
            else:
                N.call("sei_gemm_bf16nt_ex", g1.data_ptr(), C, 1, x1[guard + o:].data_ptr(), C, 1, out_a[t].data_ptr(), None, C, C, K1, 5, None, None, None, None, 0, 0)
    def one():
        N.call("sei_gemm_bf16nt_dw2_taps", g1.data_ptr(), g2.data_ptr(), C, x1[guard:].data_ptr(), x2[guard:].data_ptr(), C, out_b.data_ptr(), C, C, K1, K2, 1, 9, ctypes.cast(rows_c, ctypes.c_void_p), C * C)
    nine(); one(); torch.cuda.synchronize()
    err = float((out_a - out_b).abs().max() / out_a.abs().max())
    fl = 2.0 * (K1 + K2) * C * C * 9
    tn, to = timeit(nine), timeit(one)
    print(f"rows {K1}+{K2}: nine launches {tn:7.0f} us ({fl/tn/1e6:5.0f} TF)   one batched launch {to:7.0f} us ({fl/to/1e6:5.0f} TF)   max rel diff {err:.1e}", flush=True)

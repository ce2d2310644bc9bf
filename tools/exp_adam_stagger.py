"""Weight-gradient GEMMs with the Adam epilogue (128 x 128 loop, two workgroups per CU): a start-up phase offset for the
second workgroup of every CU, so that one's 26-byte-per-element epilogue runs under the other's main loop instead of both
doing the same thing at the same time. SEI_ADAM_STAGGER=mode,sleeps (experiment build): mode 1 = blocks 256-511 start late,
2 = every other block of an XCD, 3 = every second group of 256 blocks; sleeps x s_sleep(64) (~1.7 us each).

RESULT (round 5): the delay only adds to the launch time (see gemm_bf16nt.hip, pq_adam_auto); the kernel hook was removed
again, so against the product library every setting below times the same launch."""
import ctypes, os, sys, statistics, torch
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "scale-equivariant-imaging_amd"))
import _native as N


def timeit(fn, iters=5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for (M, Nn, K1, K2) in ((2048, 8192, 1152, 2304), (8192, 2048, 1152, 2304), (512, 2048, 4608, 9216), (8192, 32768, 288, 576)):
    g = torch.Generator(device="cuda").manual_seed(1)
    A1 = (0.05 * torch.randn((K1, M), device="cuda", generator=g)).bfloat16()
    A2 = (0.05 * torch.randn((K2, M), device="cuda", generator=g)).bfloat16()
    B1 = torch.randn((K1, Nn), device="cuda", generator=g).bfloat16()
    B2 = torch.randn((K2, Nn), device="cuda", generator=g).bfloat16()
    host = (ctypes.c_float * 6)()
    N.call("sei_adam_scalars", 1e-4, 0.9, 0.999, 1e-8, 0.0, 3, ctypes.cast(host, ctypes.c_void_p))
    hyper = torch.tensor(list(host), device="cuda")
    p = 0.02 * torch.randn((M, Nn), device="cuda", generator=g)
    m, v = torch.full_like(p, 1e-3), torch.full_like(p, 1e-5)
    s16 = torch.empty((M, Nn), device="cuda", dtype=torch.bfloat16)

    def fused():
        N.call("sei_gemm_bf16nt_dw2_adam", A1.data_ptr(), A2.data_ptr(), M, B1.data_ptr(), B2.data_ptr(), Nn, p.data_ptr(),
               m.data_ptr(), v.data_ptr(), s16.data_ptr(), hyper.data_ptr(), M, Nn, K1, K2)

    settings = ["0,0"] + [f"{mode},{sl}" for mode in (1, 2, 3) for sl in (4, 8, 12, 16, 24)]
    times = {s: [] for s in settings}
    for rnd in range(3):
        for s in settings:
            os.environ["SEI_ADAM_STAGGER"] = s
            fused()
            torch.cuda.synchronize()
            times[s].append(timeit(fused))
    print(f"{M}x{Nn}x({K1}+{K2}): " + "  ".join(f"[{s}] {statistics.median(t):.0f}" for s, t in times.items()), flush=True)

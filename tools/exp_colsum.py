import os, sys, torch
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "scale-equivariant-imaging_amd"))
import _native as N
def timeit(fn, iters=30):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for lvl in range(5):
    M = 64 * (48 >> lvl if lvl < 4 else 3) ** 2; Nn = 128 << (2 * lvl)
    X = torch.randn((M, Nn), device="cuda").bfloat16(); out = torch.zeros(Nn, device="cuda")
    t = timeit(lambda: N.call("sei_colsum_bf16", X.data_ptr(), out.data_ptr(), M, Nn))
    ref = X.double().sum(0); out.zero_(); N.call("sei_colsum_bf16", X.data_ptr(), out.data_ptr(), M, Nn)
    err = float((out.double() - ref).abs().max() / ref.abs().max())
    print(f"M={M} N={Nn}: {t:6.1f} us  {M*Nn*2/t/1e6:6.2f} TB/s  err {err:.1e}")
print("cast16 + colsum (f32 -> bf16):")
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "scale-equivariant-imaging_amd"))
for lvl in range(5):
    M = 64 * (48 >> lvl if lvl < 4 else 3) ** 2
    for mult in (1, 4):
        C = (32 << (2 * lvl)) * mult
        X = torch.randn((M, C), device="cuda"); Y = torch.empty((M, C), dtype=torch.bfloat16, device="cuda"); cs = torch.zeros(C, device="cuda")
        t1 = timeit(lambda: N.call("sei_cast_transpose_bf16", X.data_ptr(), 0, Y.data_ptr(), None, M, C, M, cs.data_ptr()))
        t2 = timeit(lambda: N.call("sei_cast_transpose_bf16", X.data_ptr(), 0, Y.data_ptr(), None, M, C, M, None))
        print(f"M={M} C={C}: with colsum {t1:6.1f} us ({M*C*6/t1/1e6:5.2f} TB/s), cast only {t2:6.1f} us")
print("f32 column sums (plain and row-weighted):")
for lvl in range(5):
    M = 64 * (48 >> lvl if lvl < 4 else 3) ** 2; C = 32 << (2 * lvl)
    X = torch.randn((M, C), device="cuda"); out = torch.zeros(C, device="cuda"); w = torch.rand(M, device="cuda")
    t1 = timeit(lambda: N.call("sei_colsum_f32", X.data_ptr(), out.data_ptr(), M, C))
    t2 = timeit(lambda: N.call("sei_colsum_weighted_f32", X.data_ptr(), w.data_ptr(), out.data_ptr(), M, C))
    out.zero_(); N.call("sei_colsum_weighted_f32", X.data_ptr(), w.data_ptr(), out.data_ptr(), M, C)
    ref = (X.double() * w.double()[:, None]).sum(0)
    err = float((out.double() - ref).abs().max() / ref.abs().max())
    print(f"M={M} C={C}: plain {t1:6.1f} us ({M*C*4/t1/1e6:5.2f} TB/s), weighted {t2:6.1f} us, err {err:.1e}")

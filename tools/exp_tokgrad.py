"""sei_tokgrad_bf16 (token-streaming weight gradient, token_gemm.hip) against sei_gemm_bf16nt_dw2 on SwinIR's four
linear layers at the bench's token counts (2B and B passes of batch 32 at 48 x 48), one launch per weight and ONE launch
for the eight 192 x 192 blocks of a Swin block."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "scale-equivariant-imaging_amd"))
import _native as N
def timeit(fn, iters=30):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
K1, K2 = 147456, 73728
if len(sys.argv) > 1: K1, K2 = int(sys.argv[1]), int(sys.argv[2])
blocks, olds, checks, total_mb, t_old = [], [], [], 0.0, 0.0
for name, Mo, Ni in [("qkv", 576, 192), ("proj", 192, 192), ("fc1", 384, 192), ("fc2", 192, 384)]:
    y1 = torch.randn((K1, Mo), device="cuda").bfloat16(); y2 = torch.randn((K2, Mo), device="cuda").bfloat16()
    x1 = torch.randn((K1, Ni), device="cuda").bfloat16(); x2 = torch.randn((K2, Ni), device="cuda").bfloat16()
    d = torch.zeros((Mo, Ni), device="cuda"); d0 = torch.zeros((Mo, Ni), device="cuda"); dg = torch.zeros((Mo, Ni), device="cuda")
    new = lambda: N.call("sei_tokgrad_bf16", y1.data_ptr(), y2.data_ptr(), Mo, x1.data_ptr(), x2.data_ptr(), Ni, d.data_ptr(), Ni, Mo, Ni, K1, K2)
    old = lambda: N.call("sei_gemm_bf16nt_dw2", y1.data_ptr(), y2.data_ptr(), Mo, x1.data_ptr(), x2.data_ptr(), Ni, d0.data_ptr(), Mo, Ni, K1, K2, 1)
    new(); old(); torch.cuda.synchronize()
    ref = y1.float().T @ x1.float() + y2.float().T @ x2.float()
    e1 = ((d - ref).abs().max() / ref.abs().max()).item(); e0 = ((d0 - ref).abs().max() / ref.abs().max()).item()
    mb = (K1 + K2) * (Mo + Ni) * 2 / 1e6
    t0 = timeit(old); t1 = timeit(new)
    t_old += t0; total_mb += mb
    print(f"{name:5s} {Mo}x{Ni}: operands {mb:6.1f} MB  dw2 {t0:6.1f} us ({mb/t0:.2f} TB/s)  tokgrad {t1:6.1f} us ({mb/t1:.2f} TB/s)  rel err {e0:.1e} / {e1:.1e}", flush=True)
    for gy in range(Mo // 192):
        for gx in range(Ni // 192):
            blocks.append(N.TokGradBlock(y1.data_ptr(), y2.data_ptr(), x1.data_ptr(), x2.data_ptr(), Mo, Ni, 192 * gy, 192 * gx,
                                         dg.data_ptr() + 4 * (192 * gy * Ni + 192 * gx), Ni))
    checks.append((name, dg, ref, (y1, y2, x1, x2)))
arr = (N.TokGradBlock * len(blocks))(*blocks)
grouped = lambda: N.call("sei_tokgrad_bf16_blocks", ctypes.addressof(arr), len(blocks), K1, K2)
grouped(); torch.cuda.synchronize()
errs = " ".join(f"{n} {((dg - ref).abs().max() / ref.abs().max()).item():.1e}" for n, dg, ref, _ in checks)
tg = timeit(grouped)
print(f"one launch, {len(blocks)} blocks: {tg:6.1f} us ({total_mb/tg:.2f} TB/s of {total_mb:.0f} MB) against {t_old:6.1f} us for the four dw2 launches; rel err {errs}")

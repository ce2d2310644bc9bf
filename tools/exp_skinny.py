"""The skinny 1x1 convolutions between the deep levels (Downsample's convolution behind the resampler and its data
gradient: M = 288 ... 2304 rows against 2048 x 8192 / 512 x 2048 weights) under every tile shape and forced K-split:
which schedule the dispatch of gemm_bf16nt.hip should take for them. Tools build (make tuning). Interleaved rounds."""
import os, sys, statistics, torch
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "scale-equivariant-imaging_amd"))
sys.path.insert(1, os.path.dirname(os.path.abspath(__file__)))
import _tuning; _tuning.use()
from models import _ops
import _native
lib = _native.lib()
def once(fn, iters=8):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
shapes = [(576, 8192, 2048, "rowscale"), (288, 8192, 2048, "rowscale"), (2304, 2048, 512, "rowscale"), (1152, 2048, 512, "rowscale"),
          (576, 2048, 8192, "none_kr"), (288, 2048, 8192, "none_kr"), (2304, 512, 2048, "none_kr"), (1152, 512, 2048, "none_kr"),
          (9216, 128, 512, "none_kr"), (4608, 128, 512, "none_kr")]
if len(sys.argv) > 1:
    shapes = [s for s in shapes if s[3] == sys.argv[1]]
for (M, N, K, kind) in shapes:
    kr = kind.endswith("_kr")
    A = torch.randn((M, K), device="cuda").bfloat16(); B = torch.randn((K, N) if kr else (N, K), device="cuda").bfloat16()
    out = torch.empty((M, N), device="cuda"); bias = torch.randn(N, device="cuda"); s = torch.rand(M, device="cuda")
    if kind == "none_kr": f = lambda: _ops.gemm_nt16(A, B, M, N, K, _ops.EPI_NONE, out32=out, b_rmajor=True)
    else: f = lambda: _ops.gemm_nt16(A, B, M, N, K, _ops.EPI_BIAS_ROWSCALE, out32=out, bias=bias, R1=s)
    ref = (A.float() @ (B.float() if kr else B.float().T)) + (0 if kr else bias[None, :] * s[:, None])
    tiles = [0, 1, 3, 31, 32] if M % 288 == 0 else [0, 1, 3, 30, 33]
    configs = [(t, sk) for t in tiles for sk in (0, 1, 2, 4, 8, 16) if not (t == 0 and sk)]
    times = {c: [] for c in configs}
    bad = {}
    for rnd in range(3):
        for c in configs:
            lib.sei_debug_set_nt_tile(c[0]); lib.sei_debug_set_nt_tile(1000 + c[1])
            try:
                f(); torch.cuda.synchronize()
            except RuntimeError as e:
                bad[c] = "refused"; continue
            if rnd == 0:
                err = float((out - ref).abs().max() / ref.abs().max())
                if err > 2e-2: bad[c] = f"err {err:.1e}"
            times[c].append(once(f))
    lib.sei_debug_set_nt_tile(0); lib.sei_debug_set_nt_tile(1000)
    res = sorted((statistics.median(t), c) for c, t in times.items() if t and c not in bad)
    auto = statistics.median(times[(0, 0)])
    print(f"{M}x{N}x{K} {kind}: auto {auto:.1f}us | best " + "  ".join(f"t{c[0]}/sk{c[1]} {t:.1f}" for t, c in res[:6])
          + (f" | bad {bad}" if bad else ""), flush=True)

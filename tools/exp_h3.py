"""Pre-GELU activation stored in f32 (today) or bf16: the expanding 1x1 convolution's launch time per level."""
import os, sys, statistics, torch
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "scale-equivariant-imaging_amd"))
from models import _ops
def once(fn, iters=5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for (M, N, K) in ((147456, 128, 32), (36864, 512, 128), (9216, 2048, 512), (2304, 8192, 2048), (576, 32768, 8192)):
    A = torch.randn((M, K), device="cuda").bfloat16(); B = torch.randn((N, K), device="cuda").bfloat16()
    o32 = torch.empty((M, N), device="cuda"); o16 = torch.empty((M, N), device="cuda", dtype=torch.bfloat16)
    h4 = torch.empty((M, N), device="cuda", dtype=torch.bfloat16); bias = torch.randn(N, device="cuda")
    f32 = lambda: _ops.gemm_nt16(A, B, M, N, K, _ops.EPI_BIAS_GELU, out32=o32, bias=bias, D2_16=h4)
    f16 = lambda: _ops.gemm_nt16(A, B, M, N, K, _ops.EPI_BIAS_GELU, out16=o16, bias=bias, D2_16=h4)
    t = {"f32": [], "bf16": []}
    for rnd in range(5):
        for name, f in (("f32", f32), ("bf16", f16)):
            f(); torch.cuda.synchronize()
            t[name].append(once(f))
    print(f"{M}x{N}x{K}: " + "  ".join(f"h3 {k} {statistics.median(v):.1f}us" for k, v in t.items()), flush=True)

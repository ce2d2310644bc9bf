"""The contracting GEMMs of the step (K = 4N, float32 out, split K through slabs) fetch 2.4-4x their operands beyond L2
(profiles/r06_i): does another band width of the tile order (a squarer patch of tiles per XCD) pay in TIME now?
sei_gemm_bf16nt_ws's band argument against the automatic choice, tile and K slices left to the dispatcher.
    python tools/exp_band_contracting.py"""
import os, sys, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "scale-equivariant-imaging_amd"))
import _native as N
from models import _ops
EPI_NONE, EPI_BIAS_RES = 0, 3
SHAPES = [(2304, 2048, 8192, 0, EPI_BIAS_RES, 3), (1152, 2048, 8192, 0, EPI_BIAS_RES, 3), (576, 8192, 32768, 0, EPI_BIAS_RES, 1),
          (288, 8192, 32768, 0, EPI_BIAS_RES, 1), (3456, 2048, 8192, 1, EPI_NONE, 2), (864, 8192, 32768, 1, EPI_NONE, 1),
          (864, 2048, 8192, 1, EPI_NONE, 1), (9216, 512, 2048, 0, EPI_BIAS_RES, 3), (13824, 512, 2048, 1, EPI_NONE, 2)]


def timeit(fn, iters=12):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


ws, ws_bytes = _ops.splitk_workspace("cuda:0")
gain = 0.0
for M, Nn, K, brm, epi, count in SHAPES:
    g = torch.Generator(device="cuda").manual_seed(M + Nn + K)
    A = torch.randn((M, K), device="cuda", generator=g).bfloat16()
    B = (0.05 * torch.randn((K, Nn) if brm else (Nn, K), device="cuda", generator=g)).bfloat16()
    bias = torch.randn(Nn, device="cuda", generator=g)
    R1 = torch.randn((M, Nn), device="cuda", generator=g) if epi == EPI_BIAS_RES else None
    out = torch.empty((M, Nn), device="cuda")

    def run(band):
        N.call("sei_gemm_bf16nt_ws", A.data_ptr(), K, 0, B.data_ptr(), Nn if brm else K, brm, out.data_ptr(), None, M, Nn, K, epi,
               N.ptr(bias) if epi == EPI_BIAS_RES else None, N.ptr(R1), None, None, None, ws, ws_bytes, 0, band, 0)

    auto = timeit(lambda: run(0))
    ref = out.clone()
    row = []
    for band in (1, 2, 3, 4, 6, 8, 16):
        t = timeit(lambda: run(band))
        assert float((out - ref).abs().max() / ref.abs().max()) < 1e-3
        row.append((t, band))
    auto = min(auto, timeit(lambda: run(0)))
    best = min(row)
    gain += count * max(0.0, auto - best[0])
    print(f"{M:6d} x {Nn:5d} x {K:5d} brm {brm} x{count}: auto {auto:7.1f} us | " + "  ".join(f"band {b}: {t:6.1f}" for t, b in row), flush=True)
print(f"best band per shape instead of the automatic one: -{gain:.0f} us per step")

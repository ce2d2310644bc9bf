"""Beyond-L2 traffic of every distinct forward / data-gradient GEMM launch of the timed step, one shape after the other, for
`rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes (one counter per pass). Every shape is launched REPS times in the
order of tools/exp_tile_sweep.py's SHAPES; tools/pmc_step_gemms_summary.py pairs the dispatches with the shapes.
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/X/fetch -o pmc -- python3 tools/pmc_step_gemms.py"""
import os, sys, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "scale-equivariant-imaging_amd"))
sys.path.insert(1, os.path.join(ROOT, "tools"))
import _native as N
from models import _ops
import importlib.util
spec = importlib.util.spec_from_file_location("sweep_shapes", os.path.join(ROOT, "tools", "exp_tile_sweep_shapes.py"))
mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
REPS = 3
EPI_NONE, EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_RES, EPI_MUL_DGELU, EPI_ACCUM, EPI_ROWSCALE = range(7)
ws, ws_bytes = _ops.splitk_workspace("cuda:0")
torch.cuda.synchronize()
for M, Nn, K, brm, epi, count in mod.SHAPES:
    g = torch.Generator(device="cuda").manual_seed(M + Nn + K)
    A = torch.randn((M, K), device="cuda", generator=g).bfloat16()
    B = (0.05 * torch.randn((K, Nn) if brm else (Nn, K), device="cuda", generator=g)).bfloat16()
    bias = torch.randn(Nn, device="cuda", generator=g)
    R1 = torch.randn((M, Nn), device="cuda", generator=g) if epi in (EPI_BIAS_RES, EPI_MUL_DGELU) else \
        (torch.rand(M, device="cuda", generator=g) if epi == EPI_ROWSCALE else None)
    to16 = epi == EPI_MUL_DGELU
    out32 = None if to16 else torch.empty((M, Nn), device="cuda")
    out16 = torch.empty((M, Nn), device="cuda", dtype=torch.bfloat16) if to16 else None
    d2 = torch.empty((M, Nn), device="cuda", dtype=torch.bfloat16) if epi == EPI_BIAS_GELU else None
    colsum = torch.zeros(Nn, device="cuda") if to16 else None
    torch.cuda.synchronize()
    N.call("sei_axpy", bias.data_ptr(), bias.data_ptr(), 0.0, bias.data_ptr(), 4)        # marker dispatch between shapes
    for _ in range(REPS):
        N.call("sei_gemm_bf16nt_ws", A.data_ptr(), K, 0, B.data_ptr(), Nn if brm else K, brm, N.ptr(out32), N.ptr(out16), M, Nn, K,
               epi, N.ptr(bias) if epi in (EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_RES, EPI_ROWSCALE) else None, N.ptr(R1), None,
               N.ptr(d2), N.ptr(colsum), ws, ws_bytes, 0, 0, 0)
    torch.cuda.synchronize()
print("done", len(mod.SHAPES), "shapes x", REPS)

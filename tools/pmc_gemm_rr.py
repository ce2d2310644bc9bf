"""Run one reduction-major (weight-gradient) GEMM shape a few times, for rocprofv3 --pmc passes."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "scale-equivariant-imaging_amd"))
sys.path.insert(1, os.path.dirname(os.path.abspath(__file__)))
import _tuning; _tuning.use()    # process-wide tile switches live in the tools-only build
from models import _ops
import _native
M, N, K = (int(a) for a in sys.argv[1:4])
band = int(sys.argv[4]) if len(sys.argv) > 4 else 0
_native.lib().sei_debug_set_nt_tile(100 + band)
A = torch.randn((K, M), device="cuda").bfloat16(); B = torch.randn((K, N), device="cuda").bfloat16()
out = torch.zeros((M, N), device="cuda")
for _ in range(4):
    _ops.gemm_nt16(A, B, M, N, K, _ops.EPI_NONE, out32=out, a_rmajor=True, b_rmajor=True)
torch.cuda.synchronize()

"""Run a single NT GEMM shape a few times (for `rocprofv3 --pmc` passes on one kernel)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "scale-equivariant-imaging_amd"))
sys.path.insert(1, os.path.dirname(os.path.abspath(__file__)))
import _tuning; _tuning.use()    # process-wide tile switches live in the tools-only build
from models import _ops
import _native
M, N, K, tile = (int(a) for a in sys.argv[1:5])
_native.lib().sei_debug_set_nt_tile(tile)
A = torch.randn((M, K), device="cuda").bfloat16(); B = torch.randn((N, K), device="cuda").bfloat16()
out = torch.zeros((M, N), device="cuda")
for _ in range(6):
    _ops.gemm_nt16(A, B, M, N, K, _ops.EPI_NONE, out32=out)
torch.cuda.synchronize()

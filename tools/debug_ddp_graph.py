import os, sys, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "scale-equivariant-imaging_amd")); sys.path.insert(1, ROOT)
import parallel, bench
rank, local, world = parallel.init_from_env()
dev = "cuda:0"; torch.cuda.set_device(dev)
from losses import get_loss; from models import _ops, get_model; from optim import FlatAdam; from physics import get_physics
from graphs import GraphedLossStep
args = bench.reference_args(dev, 8, 3); _ops.set_compute_dtype("bf16"); torch.manual_seed(0)
p = get_physics(args, dev); model = get_model(args, p, dev); model.to(dev).train(); bb = model.get_backbone()
parallel.broadcast_parameters(bb.flat_params)
lf = get_loss(args, p); red = parallel.FlatGradientReducer(bb.flat_grads) if world > 1 else None
opt = FlatAdam(model, lr=1e-4, reducer=red)
x = torch.rand(8, 3, 256, 256, device=dev); torch.cuda.manual_seed(7 + rank); y = p(x)
def fin(t): return bool(torch.isfinite(t).all())
print(rank, "params finite", fin(bb.flat_params), flush=True)
g = GraphedLossStep(lf, model, opt, (8, 3, 48, 48))
print(rank, "after capture: grads finite", fin(bb.flat_grads), "params", fin(bb.flat_params), flush=True)
for it in range(3):
    l = g(x, y); torch.cuda.synchronize()
    print(rank, it, "loss", float(l), "grads finite", fin(bb.flat_grads), "gnorm", float(bb.flat_grads.norm()), flush=True)
    if red is not None: red.reduce_async(); red.wait_all(); torch.cuda.synchronize(); print(rank, it, "reduced finite", fin(red.comm), float(red.comm.float().norm()), flush=True)
    opt.step(); torch.cuda.synchronize()
    print(rank, it, "params finite", fin(bb.flat_params), "shadow finite", fin(bb.flat_shadow.float()), flush=True)

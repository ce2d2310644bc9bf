import os, sys, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "scale-equivariant-imaging_amd")); sys.path.insert(1, ROOT)
import bench
dev = "cuda:0"
from losses import get_loss; from models import _ops, get_model; from optim import FlatAdam; from physics import get_physics
from graphs import GraphedLossStep
mode = sys.argv[1]
for seed in (7, 8, 9, 10):
    args = bench.reference_args(dev, 8, 3); _ops.set_compute_dtype("bf16"); torch.manual_seed(0)
    p = get_physics(args, dev); model = get_model(args, p, dev); model.to(dev).train(); bb = model.get_backbone()
    lf = get_loss(args, p); opt = FlatAdam(model, lr=1e-4)
    x = torch.rand(8, 3, 256, 256, device=dev); torch.cuda.manual_seed(seed); y = p(x)
    g = GraphedLossStep(lf, model, opt, (8, 3, 48, 48)) if mode == "graph" else None
    out = []
    for it in range(4):
        if g is not None:
            l = g(x, y)
        else:
            opt.zero_grad(); l = lf(x=x, y=y, model=model); l.backward()
        torch.cuda.synchronize()
        # per-parameter norms to locate garbage
        bad = [(n, float(q.grad.norm())) for n, q in bb.named_parameters() if not torch.isfinite(q.grad).all() or q.grad.norm() > 1e6]
        out.append((round(float(l), 4), f"{float(bb.flat_grads.norm()):.3e}", bad[:3]))
        opt.step()
    print(mode, "seed", seed, out, flush=True)

import os, sys, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "scale-equivariant-imaging_amd")); sys.path.insert(1, ROOT)
import bench
dev = "cuda:0"
from losses import get_loss; from models import _ops, get_model; from optim import FlatAdam; from physics import get_physics
from graphs import GraphedLossStep
args = bench.reference_args(dev, 8, 3); _ops.set_compute_dtype("bf16"); torch.manual_seed(0)
p = get_physics(args, dev); model = get_model(args, p, dev); model.to(dev).train(); bb = model.get_backbone()
lf = get_loss(args, p); opt = FlatAdam(model, lr=1e-4)
x = torch.rand(8, 3, 256, 256, device=dev); torch.cuda.manual_seed(7); y = p(x)
g = GraphedLossStep(lf, model, opt, (8, 3, 48, 48))
for it in range(3):
    l = g(x, y); torch.cuda.synchronize()
    print("it", it, "loss", float(l), "gnorm", float(bb.flat_grads.norm()))
    for n, q in bb.named_parameters():
        st = getattr(q, "_sei_shadow", None)
        if st is not None:
            w16, wt16 = st[1], st[2]
            R, C = q.shape[0], q.shape[1]
            e1 = float((w16.float() - q.detach().view(R, C).bfloat16().float()).abs().max())
            e2 = float((wt16.float() - q.detach().view(R, C).t().bfloat16().float()).abs().max())
            print("   ", n, tuple(q.shape), "plain err", e1, "transposed err", e2, "ptr in flat", w16.data_ptr() - bb.flat_shadow.data_ptr())
        bad = not torch.isfinite(q.grad).all() or q.grad.norm() > 1e5
        if bad: print("    BAD grad", n, float(q.grad.norm()))
    opt.step(); torch.cuda.synchronize()

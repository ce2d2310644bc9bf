"""Single-GPU validation of the early-release event: is the external event, recorded inside the captured
backward, really waited for by a side stream (poisoned region must be clean in the side stream's copy), and how
long before the end of the graph does it fire?"""
import os, sys, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scale-equivariant-imaging_amd"))
import bench
import physics, models
from graphs import GraphedLossStep
from losses import get_loss
from models import _ops
from optim import FlatAdam
_ops.set_compute_dtype("bf16")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
args = bench.reference_args("cuda")
torch.manual_seed(0)
p = physics.get_physics(args, "cuda")
model = models.get_model(args, p, "cuda").to("cuda")
bb = model.get_backbone()
lf = get_loss(args, p)
opt = FlatAdam(model, lr=1e-4)
x = torch.rand(B, 3, 256, 256, device="cuda"); y = p(x)
g = GraphedLossStep(lf, model, opt, (B, 3, 48, 48), early_release=True)
print("early_grads:", None if g.early_grads is None else (g.early_grads[1], g.early_grads[2], bb.flat_grads.numel()))
assert g.early_grads is not None
ev, lo, hi = g.early_grads
side = torch.cuda.Stream()
buf = torch.empty(hi - lo, device="cuda")
for it in range(3):
    bb.flat_grads[lo:hi].fill_(float("nan"))
    t0, t_side, t_end = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    t0.record()
    g(x, y)                                   # enqueue the replay on the current stream
    with torch.cuda.stream(side):
        ev.wait(side)
        buf.copy_(bb.flat_grads[lo:hi])
        t_side.record(side)
    t_end.record()
    torch.cuda.synchronize()
    clean = bool(torch.isfinite(buf).all())
    same = bool(torch.equal(buf, bb.flat_grads[lo:hi]))
    print(f"iter {it}: side copy clean={clean} equal_to_final={same}  side done at {t0.elapsed_time(t_side):6.2f} ms, graph done at {t0.elapsed_time(t_end):6.2f} ms")
    opt.step()

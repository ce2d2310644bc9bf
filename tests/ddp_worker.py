"""Worker of tests/test_ddp_gpu.py: one rank of a data-parallel run (launched by torch.distributed.run, or alone
with WORLD_SIZE unset as the single-process reference on the concatenated batch).

Every process builds the same model and the same GLOBAL batch, crop offsets and random draws from fixed CPU
generators; rank r trains on images [r*B/W, (r+1)*B/W) with its slice of the draws. Writes
{out}/rank{r}.pt = {"params", "grads_step0", "losses"}.
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scale-equivariant-imaging_amd"))
sys.path.insert(1, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--global-batch", type=int, default=8)
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--graph", type=int, default=1)
    ap.add_argument("--mode", default="all_reduce")
    ap.add_argument("--comm", default="f32")
    ap.add_argument("--hidden", type=int, default=8, help="32: GEMM weights that the bf16 mode reads through bf16 copies only")
    ap.add_argument("--shard", type=int, default=1, help="rs_ag: the sharded optimizer step (FlatAdam's default)")
    ap.add_argument("--direct-min", type=int, default=0,
                    help=">0: weight gradients of at least this many elements are stored (store_min_numel=0) and, with "
                         "--comm bf16, written into the exchange buffer as bf16 by their GEMM")
    opt = ap.parse_args()

    import parallel                                     # before any GPU call: spawns nothing, reads the env
    rank, local, world = parallel.init_from_env()
    # the ranks of the rehearsal share the one GPU (gloo exchange); over RCCL (no SEI_DIST_BACKEND) each takes its own
    torch.cuda.set_device(local if world > 1 and torch.distributed.get_backend() == "nccl" else 0)
    import bench
    import physics
    import models
    from graphs import GraphedLossStep
    from losses import get_loss
    from losses.sure import embed_probe
    from models import _ops
    from optim import FlatAdam

    _ops.set_compute_dtype(opt.dtype)
    args = bench.reference_args("cuda", opt.hidden, 3)
    torch.manual_seed(0)
    p = physics.get_physics(args, "cuda")
    model = models.get_model(args, p, "cuda").to("cuda")
    bb = model.get_backbone()
    if world > 1:
        parallel.broadcast_parameters(bb.flat_params)
    lf = get_loss(args, p)
    comm = torch.bfloat16 if opt.comm == "bf16" else torch.float32
    reducer = parallel.FlatGradientReducer(bb.flat_grads, comm_dtype=comm, chunk_mib=1, mode=opt.mode) \
        if world > 1 else None
    optim = FlatAdam(model, lr=1e-4, reducer=reducer, shard_step=bool(opt.shard))
    sharded = reducer is not None and reducer.mode == "sharded"
    assert sharded == (world > 1 and opt.mode == "rs_ag" and bool(opt.shard))

    G, B = opt.global_batch, opt.global_batch // world
    lo = rank * B
    gen = torch.Generator().manual_seed(77)
    x = torch.rand((G, 3, 256, 256), generator=gen)
    y = x + 5 / 255 * torch.randn((G, 3, 256, 256), generator=gen)
    xs, ys = x[lo:lo + B].cuda(), y[lo:lo + B].cuda()
    kw = dict(store_min_numel=0, fuse_min_numel=opt.direct_min) if opt.direct_min > 0 else dict(direct_bf16_grads=False)
    graphed = GraphedLossStep(lf, model, optim, (B, 3, 48, 48), **kw) if opt.graph else None
    direct = graphed is not None and bool(graphed.direct_views)
    if opt.direct_min > 0 and opt.comm == "bf16" and world > 1 and opt.dtype == "bf16" and opt.hidden == 8:
        assert len(graphed.direct_views) == 2, len(graphed.direct_views)
    losses, grads0 = [], None
    for step in range(opt.steps):
        b = torch.randn((G, 3, 36, 36), generator=gen)
        rate = torch.tensor([0.75, 0.5])[torch.randint(0, 2, (G,), generator=gen)]
        center = 2 * torch.rand((G, 2), generator=gen) - 1
        noise = torch.randn((G, 3, 48, 48), generator=gen)
        draws = {"b": embed_probe(torch.empty(B, 3, 48, 48, device="cuda"), b[lo:lo + B].cuda(), 6),
                 "rate": rate[lo:lo + B].cuda(), "center": center[lo:lo + B].cuda().view(B, 1, 1, 2),
                 "noise": noise[lo:lo + B].cuda()}
        torch.manual_seed(1000 + step)                  # the crop offsets: one draw per (global) batch
        if graphed is not None:
            val = graphed(xs, ys, draws=draws)
        else:
            optim.zero_grad()
            val = lf(x=xs, y=ys, model=model, draws=draws)
            val.backward()
        if reducer is not None:
            reducer.reduce_async(direct=direct)
            if not (sharded and step > 0):
                reducer.wait_all()
        if step == 0:
            if sharded:                                 # each rank holds its shares: put the bucket together (test aid)
                g = reducer.gathered_gradient().float() / world
            else:
                g = (reducer.comm if reducer is not None else bb.flat_grads).float() / world
            grads0 = g.cpu().clone()
        optim.step()
        losses.append(float(parallel.all_reduce_mean_scalar(val.detach().clone())))
    stale = bool(getattr(optim, "_master_stale", False))
    shadow = bb.flat_shadow.cpu().clone() if opt.dtype == "bf16" else None
    optim.consolidate()                                 # sharded step: float32 masters + moments of the other shares
    torch.cuda.synchronize()
    os.makedirs(opt.out, exist_ok=True)
    st = optim.state[bb.flat_params]
    torch.save({"params": bb.flat_params.cpu(), "grads_step0": grads0, "losses": losses, "stale": stale, "shadow": shadow,
                "exp_avg": st["exp_avg"].cpu(), "sharded": sharded},
               os.path.join(opt.out, f"rank{rank}.pt"))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()

"""Worker of tests/test_dist1_gpu.py: the N > 1 step -- process group, FlatGradientReducer (reduce-scatter / all-gather or
all-reduce, early release from inside the captured graph, optional bf16 exchange), sharded FlatAdam -- executed at
WORLD_SIZE = 1 on the real backend (RCCL: `nccl`) with SEI_FORCE_EXCHANGE=1, next to the plain single-process optimizer
step on a twin model fed the SAME gradients. Replaces the hook at /root/reference/src/models/__init__.py:142-145
(nn.DataParallel) for configs[3]; what this proves on one GPU: the collectives, the process group's stream ordering
against the compute stream and the side stream, and the sharded step's bookkeeping produce bit-identical parameters,
moments and bf16 copies.

Prints one JSON line. Run by the test as a fresh process (the environment decides the backend before any GPU call).
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scale-equivariant-imaging_amd"))
sys.path.insert(1, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--hidden", type=int, default=32)
    ap.add_argument("--scales", type=int, default=3)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--mode", default="rs_ag")
    ap.add_argument("--comm", default="f32")
    ap.add_argument("--direct", type=int, default=0, help="bf16 exchange: the big weight gradients written as bf16 by their GEMM")
    ap.add_argument("--early", type=int, default=1)
    opt = ap.parse_args()

    import parallel
    assert parallel.force_exchange(), "run with SEI_FORCE_EXCHANGE=1"
    rank, local, world = parallel.init_from_env()              # before any GPU call
    import torch.distributed as dist
    assert world == 1 and dist.is_initialized() and parallel.exchange_active()
    backend = dist.get_backend()
    torch.cuda.set_device(0)

    calls = {"reduce_scatter_tensor": 0, "all_gather_into_tensor": 0, "all_reduce": 0, "broadcast": 0}
    for name in calls:                                          # count what really reaches torch.distributed
        inner = getattr(dist, name)

        def counted(*a, _inner=inner, _name=name, **k):
            calls[_name] += 1
            return _inner(*a, **k)
        setattr(dist, name, counted)

    import bench
    import models
    import physics
    from graphs import GraphedLossStep
    from losses import get_loss
    from losses.sure import embed_probe
    from models import _ops
    from optim import FlatAdam

    _ops.set_compute_dtype(opt.dtype)
    args = bench.reference_args("cuda", opt.hidden, opt.scales)
    B = opt.batch

    def build():
        torch.manual_seed(0)
        p = physics.get_physics(args, "cuda")
        m = models.get_model(args, p, "cuda").to("cuda")
        m.train()
        return p, m

    phys, model = build()
    _, twin = build()
    bb, tb = model.get_backbone(), twin.get_backbone()
    assert torch.equal(bb.flat_params, tb.flat_params)
    parallel.broadcast_parameters(bb.flat_params)
    lf = get_loss(args, phys)
    comm = torch.bfloat16 if opt.comm == "bf16" else torch.float32
    reducer = parallel.FlatGradientReducer(bb.flat_grads, comm_dtype=comm, chunk_mib=1, mode=opt.mode)
    optim = FlatAdam(model, lr=1e-4, reducer=reducer)
    plain = FlatAdam(twin, lr=1e-4)                             # the unsharded, unfused single-process step
    sharded = reducer.mode == "sharded"
    assert sharded == (opt.mode == "rs_ag")

    kw = dict(store_min_numel=0, fuse_min_numel=60000) if opt.direct else dict(direct_bf16_grads=False)
    graphed = GraphedLossStep(lf, model, optim, (B, 3, 48, 48), early_release=bool(opt.early), fuse_optimizer=True, **kw)
    assert not graphed.fused_views, "the optimizer step must not be fused into the GEMMs when gradients are exchanged"
    early_event = None
    if opt.early and graphed.early_grads is not None:
        early_event = graphed.early_grads[0]
        reducer.set_early_range(graphed.early_grads[1:])
    direct = bool(graphed.direct_views)
    n_sharded = sum(reducer.is_sharded(k) for k in range(len(reducer.bounds)))

    gen = torch.Generator().manual_seed(77)
    x = torch.rand((B, 3, 256, 256), generator=gen)
    y = (x + 5 / 255 * torch.randn((B, 3, 256, 256), generator=gen)).cuda()
    x = x.cuda()
    out = {"backend": backend, "sharded": sharded, "chunks": len(reducer.bounds), "sharded_chunks": n_sharded,
           "early": early_event is not None, "direct": direct, "steps": []}
    st_a, st_b = optim.state[bb.flat_params], plain.state[tb.flat_params]
    for step in range(opt.steps):
        b = torch.randn((B, 3, 36, 36), generator=gen)
        rate = torch.tensor([0.75, 0.5])[torch.randint(0, 2, (B,), generator=gen)]
        center = 2 * torch.rand((B, 2), generator=gen) - 1
        noise = torch.randn((B, 3, 48, 48), generator=gen)
        draws = {"b": embed_probe(torch.empty(B, 3, 48, 48, device="cuda"), b.cuda(), 6), "rate": rate.cuda(),
                 "center": center.cuda().view(B, 1, 1, 2), "noise": noise.cuda()}
        torch.manual_seed(1000 + step)
        val = graphed(x, y, draws=draws)
        reducer.reduce_async(early=early_event, direct=direct)
        rec = {}
        if direct or step == 0:
            # the complete reduced bucket as the backend delivered it (a collective of its own; world 1: sum == input)
            full = reducer.gathered_gradient() if sharded else (reducer.wait_all() or reducer.comm.clone())
            if not direct:
                want = bb.flat_grads if comm == torch.float32 else bb.flat_grads.bfloat16()
                rec["exchange_exact"] = bool(torch.equal(full, want))
            tb.flat_grads.copy_(full.float())
        elif comm == torch.float32:
            tb.flat_grads.copy_(bb.flat_grads)
        else:
            tb.flat_grads.copy_(bb.flat_grads.bfloat16().float())
        optim.step()
        plain.step()
        torch.cuda.synchronize()
        rec.update(loss=float(val), params=bool(torch.equal(bb.flat_params, tb.flat_params)),
                   exp_avg=bool(torch.equal(st_a["exp_avg"], st_b["exp_avg"])),
                   exp_avg_sq=bool(torch.equal(st_a["exp_avg_sq"], st_b["exp_avg_sq"])),
                   shadow=bool(opt.dtype != "bf16" or torch.equal(bb.flat_shadow, tb.flat_shadow)),
                   moved=float((bb.flat_params - tb.flat_params).abs().max()),
                   stale=bool(optim._master_stale))
        out["steps"].append(rec)
    optim.consolidate()
    out["stale_after_consolidate"] = bool(optim._master_stale)
    out["calls"] = calls
    out["grad_norm"] = float(tb.flat_grads.norm())
    print(json.dumps(out))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

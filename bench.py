#!/usr/bin/env python3
"""Benchmark of the proposed-loss training step (BASELINE.json metric: training images/sec).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B_per_gpu]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch: Loss.forward (random 48-crop of the 256x256 pairs)
-> SURE + scale-equivariant loss (three U-Net evaluations, two physics operators, the EI resample)
-> backward -> gradient all-reduce (N>1) -> Adam. One "image" = one 256x256 pair entering Loss.forward.
Workload: config[1] of BASELINE.json -- deblurring, Gaussian_R2, noise 5, proposed loss, the reference's
default ConvolutionalModel (hidden 32, 5 scales, 645,063,043 parameters), per-GPU batch 32 (config[3]'s
256 / 8), synthetic inputs resident in HBM, random-init weights (torch.manual_seed(0)).

Prints ONE JSON line on rank 0 (contract in the task statement), with two extra objects:
  roofline     -- the dominant kernel (sei_gemm_f32: all 1x1 convolutions and their gradients), timed
                  live with HIP events on the launch stream during the timed steps.
  cpu_baseline -- the oracle's torch-CPU restatement of the same step on the host cores (N=1 only).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "scale-equivariant-imaging_amd"))
sys.path.insert(1, ROOT)

# MI355X_MICROARCH.md, dense spec peaks: v_mfma_f32_32x32x2_f32 / v_mfma_f32_32x32x16_bf16
MFMA_PEAK_TFLOPS = {"f32": 157.3, "bf16": 2500.0}
CROP, NOISE, KERNEL = 48, 5, "Gaussian_R2"


def reference_args(device, hidden=32, scales=5, task="deblurring", sr_factor=None):
    return argparse.Namespace(
        device=device, task=task, kernel=KERNEL, sr_factor=sr_factor, noise_level=NOISE, physics_v2=True,
        physics_true_adjoint=False, model_kind="Proposed", ProposedModel__architecture="Convolutional",
        ConvolutionalModel__residual=True, ConvolutionalModel__inner_residual=True,
        ConvolutionalModel__num_conv_blocks=1, ConvolutionalModel__inout_convs=True,
        ConvolutionalModel__hidden_channels=hidden, ConvolutionalModel__scales=scales,
        data_parallel_devices=None, method="proposed", partial_sure=True, sure_margin=None,
        partial_sure_sr=False, sure_cropped_div=True, sure_averaged_cst=None, Loss__crop_training_pairs=True,
        Loss__crop_size=CROP, ProposedLoss__stop_gradient=True, ProposedLoss__sure_alternative=None,
        ProposedLoss__alpha_tradeoff=1.0, ProposedLoss__transforms="Scaling_Transforms",
        ScalingTransform__kind="padded", ScalingTransform__antialias=False)


def cpu_baseline(batch, hidden, scales):
    """The oracle's restatement of the same training step on the host CPU (kind "port")."""
    from oracle import torch_path as tp
    torch.manual_seed(0)
    threads = torch.get_num_threads()
    sd = {k: v.requires_grad_(True) for k, v in tp.unet_init_state_dict(hidden, scales).items()}
    opt = torch.optim.Adam(list(sd.values()), lr=1e-4)
    k = tp.blur_kernel(KERNEL)
    A = lambda v: tp.blur_fft(v, k)
    g = torch.Generator().manual_seed(1234)
    x = torch.rand((batch, 3, 256, 256), generator=g)
    y = tp.add_noise(A(x), NOISE / 255)
    model = lambda v: tp.unet_forward(sd, v, scales=scales)
    t0 = time.perf_counter()
    opt.zero_grad()
    _, yc = tp.crop_pair(x, y, CROP, 1)
    rate, center = tp.sample_scale_params(batch)
    loss, _ = tp.proposed_loss(yc.contiguous(), A, model, NOISE / 255, margin=6, rate=rate, center=center)
    loss.backward()
    opt.step()
    dt = time.perf_counter() - t0
    return {"value": batch / dt, "unit": "images/s", "cores": threads, "kind": "port",
            "sample": f"1 un-warmed proposed-loss step (crop {CROP}, 3 fwd + 3 bwd + Adam) of the same U-Net at "
                      f"batch {batch}, float32, torch CPU ops in the reference's order (oracle/torch_path.py), "
                      f"{dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="images per GPU per step")
    ap.add_argument("--hidden", type=int, default=32)
    ap.add_argument("--scales", type=int, default=5)
    ap.add_argument("--cpu-baseline", action=argparse.BooleanOptionalAction, default=True)
    ap.add_argument("--cpu-batch", type=int, default=2)
    ap.add_argument("--profile-gemms", action=argparse.BooleanOptionalAction, default=True)
    ap.add_argument("--grad-comm", choices=["auto", "f32", "bf16"], default="auto",
                    help="dtype of the all-reduced gradient bucket (auto: bf16 in bf16 mode, f32 in f32 mode)")
    ap.add_argument("--graph", action=argparse.BooleanOptionalAction, default=True,
                    help="replay forward+backward as one hipGraph (the eager launch path otherwise)")
    ap.add_argument("--task", choices=["deblurring", "sr"], default="deblurring",
                    help="deblurring = BASELINE configs[1] (the headline); sr = configs[2] (x4 by default), a "
                         "secondary series: pairs (48r x 48r, 48 x 48) as the reference's dataset hands them over")
    ap.add_argument("--sr-factor", type=int, default=4)
    ap.add_argument("--full256", action="store_true",
                    help="secondary series of SURVEY 8d: --no-Loss__crop_training_pairs, the network sees the whole "
                         "256x256 pair (28x the pixels of the default 48-crop); use a small --batch")
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="bf16",
                    help="arithmetic type of the 1x1-conv GEMMs (f32 = parity mode, bf16 = throughput mode)")
    opt = ap.parse_args()

    import parallel
    rank, local_rank, world = parallel.init_from_env()
    if world != opt.gpus:
        raise SystemExit(f"--gpus {opt.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    device = f"cuda:{local_rank % torch.cuda.device_count()}"     # (% only matters for shared-GPU rehearsals)
    torch.cuda.set_device(device)

    import torch.distributed as dist
    from losses import get_loss
    from models import _ops, get_model
    from optim import FlatAdam
    from physics import get_physics

    sr = opt.task == "sr"
    args = reference_args(device, opt.hidden, opt.scales, opt.task, opt.sr_factor if sr else None)
    if opt.full256:
        args.Loss__crop_training_pairs = False
    _ops.set_compute_dtype(opt.dtype)
    torch.manual_seed(0)
    physics = get_physics(args, device)
    model = get_model(args, physics, device)
    model.to(device)
    model.train()
    backbone = model.get_backbone()
    nparams = sum(p.numel() for p in backbone.parameters())
    if world > 1:
        parallel.broadcast_parameters(backbone.flat_params)
    loss_fn = get_loss(args, physics)
    comm_dtype = torch.bfloat16 if (opt.dtype == "bf16" and opt.grad_comm == "auto") or opt.grad_comm == "bf16" \
        else torch.float32
    reducer = parallel.FlatGradientReducer(backbone.flat_grads, comm_dtype=comm_dtype) if world > 1 else None
    optimizer = FlatAdam(model, lr=1e-4, betas=(0.9, 0.999), reducer=reducer)

    # synthetic 256x256 pairs, resident in HBM before the timed region
    g = torch.Generator().manual_seed(1234 + rank)
    side = CROP * opt.sr_factor if sr else 256              # SR: the dataset's _HOTFIX crop (datasets/__init__.py:84-85)
    x = torch.rand((opt.batch, 3, side, side), generator=g).to(device)
    torch.manual_seed(4321 + rank)
    torch.cuda.manual_seed(4321 + rank)
    y = physics(x)

    def eager_step():
        optimizer.zero_grad()
        loss = loss_fn(x=x, y=y, model=model)
        loss.backward()
        if reducer is not None:
            reducer.reduce_async()
        optimizer.step()
        return loss

    step = eager_step
    if opt.graph:
        from graphs import GraphedLossStep
        ys = 256 if opt.full256 else CROP
        early = reducer is not None and os.environ.get("SEI_NO_EARLY_RELEASE") != "1"
        graphed = GraphedLossStep(loss_fn, model, optimizer, (opt.batch, 3, ys, ys), early_release=early)
        early_event = None
        if early and graphed.early_grads is not None:        # the bottleneck block's gradients leave early
            early_event = graphed.early_grads[0]
            reducer.set_early_range(graphed.early_grads[1:])

        def step():
            loss = graphed(x, y)
            if reducer is not None:
                reducer.reduce_async(early=early_event)
            optimizer.step()
            return loss

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(opt.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(opt.steps):
        last = step()
    fence()
    elapsed = time.perf_counter() - t0
    loss_value = float(last.detach())

    # Roofline leg: the dominant kernel family (all GEMM launches of one step). One eager step records each
    # launch's entry point and arguments (the tensors stay alive in the autograd graph / local scope), then
    # every recorded launch is re-issued back to back between HIP events on the launch stream: device time
    # per launch with a full queue, free of host-side gaps, same shapes and data as the timed steps.
    records = None
    if opt.profile_gemms:
        import _native
        _ops.profile_gemms(True)
        backbone = model.get_backbone() if hasattr(model, "get_backbone") else model
        if opt.graph and graphed.store_weight_grads:
            # the launches of the timed (graphed) step: merged weight gradients STORE instead of accumulating
            backbone.zero_grad_flat(store_weight_grads=True)
        else:
            optimizer.zero_grad()
        keep = loss_fn(x=x, y=y, model=model)
        keep.backward(retain_graph=True)       # keeps the saved activations (GEMM operands) alive for the replay
        records = _ops.profile_gemms(False)
        torch.cuda.synchronize()
        reps, total_ms, flops = 3, 0.0, 0.0
        for fl, entry, cargs in records:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            _native.call(entry, *cargs)                 # warm
            e0.record()
            for _ in range(reps):
                _native.call(entry, *cargs)
            e1.record()
            e1.synchronize()
            total_ms += e0.elapsed_time(e1) / reps
            flops += fl
            if os.environ.get("SEI_GEMM_TABLE"):            # per-launch table for kernel work (tools/, not the bench line)
                with open(os.environ["SEI_GEMM_TABLE"], "a") as f:
                    ints = [a for a in cargs if isinstance(a, int) and 0 <= a < (1 << 24)]
                    us = 1e3 * e0.elapsed_time(e1) / reps
                    f.write(f"{entry} {ints} {us:.1f} us {fl / us / 1e6:.1f} TF\n")
        del keep
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    roofline = None
    if records:
        achieved = flops / (total_ms * 1e-3) / 1e12
        peak = MFMA_PEAK_TFLOPS[opt.dtype]
        kernels = ("gemm_bf16pq_kernel<*> (quadrant schedule, deep levels) + gemm_bf16nt_kernel<*> (128x128 loop, "
                   "everything else) + gemm_bf16_kernel<*> (K<64 layers)"
                   if opt.dtype == "bf16" else "gemm_f32_kernel<*>")
        traffic, traffic_src = None, None
        pmc_file = os.path.join(ROOT, "profiles", "r01_g_pmc_gemm.json")
        if opt.dtype == "bf16" and not sr and not opt.full256 and opt.hidden == 32 and opt.scales == 5 and opt.batch == 32 and os.path.exists(pmc_file):
            pmc = json.load(open(pmc_file))                 # PMC counters cannot be read live; see the file
            traffic = round(pmc["traffic_bytes_per_launch"])
            traffic_src = ("bytes beyond L2 per GEMM launch from committed rocprofv3 --pmc passes of this command "
                           "(profiles/r01_g_pmc_gemm.json: FETCH_SIZE x2 + WRITE_SIZE, gfx950 corrections)")
        roofline = {"bound": "mfma", "kernel": kernels, "achieved": round(achieved, 2), "peak": peak,
                    "unit": "TFLOP/s", "frac": round(achieved / peak, 4), "traffic": traffic,
                    "traffic_source": traffic_src,
                    "launches_per_step": len(records), "avg_launch_us": round(1e3 * total_ms / len(records), 2),
                    "gemm_ms_per_step": round(total_ms, 2),
                    "algorithmic_gflop_per_step": round(flops / 1e9, 1),
                    "timed_with": "HIP events on the launch stream around back-to-back re-issues of every GEMM "
                                  "launch of one step (recorded arguments), right after the timed region"}

    if rank == 0:
        images = opt.batch * world * opt.steps
        out = {
            "metric": ("training images/sec, proposed-loss super-resolution" if sr else
                       "training images/sec (256x256 crops), proposed-loss deblur"),
            "value": round(images / elapsed, 2), "unit": "images/s", "n_gpus": world, "steps": opt.steps,
            "warmup": opt.warmup, "ms_per_step": round(1e3 * elapsed / opt.steps, 2), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": opt.dtype, "data": "synthetic",
            "config": {"workload": (f"BASELINE configs[2]: super-resolution x{opt.sr_factor} noise=5, proposed loss "
                                    f"(SURE + scale-EI), pairs {side}x{side} / 48x48, ConvolutionalModel "
                                    f"hidden={opt.hidden} scales={opt.scales}") if sr else
                                   ("BASELINE configs[1]: deblurring Gaussian_R2 noise=5, proposed loss (SURE + "
                                    "scale-EI), 256x256 pairs " +
                                    ("NOT cropped (full-256 series)" if opt.full256 else "cropped to 48 in Loss.forward") +
                                    f", ConvolutionalModel hidden={opt.hidden} scales={opt.scales}"),
                       "parameters": nparams, "batch_per_gpu": opt.batch, "global_batch": opt.batch * world,
                       "parallelism": f"dp{world}", "optimizer": "Adam (fused, flat bucket)",
                       "grad_allreduce": None if world == 1 else str(comm_dtype).replace("torch.", ""),
                       "launch": ("hipGraph replay of forward+backward" + (", early gradient release" if opt.graph and world > 1 and early_event is not None else "")) if opt.graph else "eager",
                       "final_loss": loss_value},
            "roofline": roofline,
        }
        if opt.cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(opt.cpu_batch, opt.hidden, opt.scales)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

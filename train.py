#!/usr/bin/env python3
"""Training driver with the reference's command line (reference: demo/train.py).

    python train.py --device cuda --method proposed --task deblurring --kernel Gaussian_R2 \
        --ProposedModel__architecture Convolutional --dataset synthetic --out_dir runs/x --epochs 2

Same flags, defaults, epoch loop, csv log and checkpoint files as the reference; the arithmetic of the
step runs in libsei_hip.so. Differences, all build-side and documented in DESIGN.md:
  * multi-GPU = one process per GPU (`python -m torch.distributed.run --nproc-per-node N train.py ...`),
    gradients summed over RCCL; `--data_parallel_devices 0,1,..` (the reference's single-process
    nn.DataParallel) stops with the equivalent launch command. `--batch_size` is PER PROCESS: the reference's
    DataParallel splits one batch over the devices, here every rank draws its own batch, so the global batch
    is world x batch_size (pass batch_size / world to keep the reference's global batch and learning rate).
  * `--grad_comm_dtype bf16` halves the bytes of the gradient exchange (summed in bf16 across ranks; the
    default f32 keeps the exchange exact whatever the compute dtype).
  * `--fused_optimizer` (default on for Adam): one fused Adam kernel over the flat parameter bucket.
  * `--fix_batched_crop`: opt out of the reference's batched-crop padding quirk (crop.py).
  * the per-step `.item()` host sync of the reference is replaced by a device-side running mean.
"""
import csv
import os
import random
import sys
from argparse import BooleanOptionalAction
from datetime import datetime

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "scale-equivariant-imaging_amd"))

import crop  # noqa: E402
import graphs  # noqa: E402
import parallel  # noqa: E402
from datasets import get_dataset  # noqa: E402
from losses import get_loss  # noqa: E402
from models import get_model  # noqa: E402
from optim import FlatAdam  # noqa: E402
from physics import get_physics  # noqa: E402
from scheduler import get_lr_scheduler  # noqa: E402
from settings import DefaultArgParser  # noqa: E402
from training import get_weights, save_training_state  # noqa: E402


def build_parser():
    parser = DefaultArgParser()
    flag, onoff = parser.add_argument, dict(action=BooleanOptionalAction)
    flag("--method", type=str)
    flag("--Loss__crop_training_pairs", default=True, **onoff)
    flag("--Loss__crop_size", type=int, default=48)
    flag("--ProposedLoss__transforms", type=str, default="Scaling_Transforms")
    flag("--ProposedLoss__stop_gradient", default=True, **onoff)
    flag("--ProposedLoss__sure_alternative", type=str, default=None)
    flag("--ProposedLoss__alpha_tradeoff", type=float, default=1.0)
    flag("--ScalingTransform__kind", type=str, default="padded")
    flag("--ScalingTransform__antialias", default=False, **onoff)
    flag("--out_dir", type=str)
    flag("--batch_size", type=int, default=8)
    flag("--epochs", type=int, default=None)
    flag("--checkpoint_interval", type=int, default=None)
    flag("--memoize_gt", default=True, **onoff)
    flag("--partial_sure", default=True, **onoff)
    flag("--sure_cropped_div", default=True, **onoff)
    flag("--sure_averaged_cst", default=None, **onoff)
    flag("--partial_sure_sr", default=False, **onoff)
    flag("--sure_margin", type=int, default=None)
    flag("--lr_scheduler_kind", type=str, default="delayed_linear_decay")
    flag("--optimizer_beta2", type=float, default=0.999)
    flag("--SyntheticDataset__deterministic_measurements", default=True, **onoff)
    flag("--GroundTruthDataset__split", type=str, default="train")
    flag("--weights", type=str, default=None)
    flag("--lr", type=float, default=None)
    flag("--optimizer", type=str, default=None)
    flag("--fine_tuning", default=False, **onoff)
    flag("--fine_tuning_params", default=False, **onoff)
    flag("--weights_distance_loss", default=False, **onoff)
    flag("--RESUME", type=str, default=None)
    # build-side additions
    flag("--fused_optimizer", default=True, **onoff)
    flag("--fix_batched_crop", default=False, **onoff)
    flag("--max_steps", type=int, default=None, help="stop each epoch after this many steps (smoke runs)")
    flag("--compute_dtype", choices=["f32", "bf16", "bf16x3"], default="f32",
         help="1x1-conv GEMM arithmetic: f32 (reference precision), bf16 MFMA with f32 accumulation, or bf16x3 (three bf16 "
              "MFMA products of head / remainder operands per float32 GEMM: 16 mantissa bits, held to the f32 parity bars)")
    flag("--hip_graph", default=True, **onoff)
    flag("--fuse_optimizer_step", default=True, **onoff,
         help="one GPU, bf16, hipGraph replay: weights of at least 2^24 elements (the U-Net's two deepest levels) take their "
              "Adam step in the epilogue of the GEMM that computes their gradient (optim.FlatAdam.fuse_weight_updates)")
    flag("--grad_comm_dtype", choices=["f32", "bf16"], default="f32",
         help="dtype of the all-reduced gradient bucket under torch.distributed.run: f32 (exact sum, default) or "
              "bf16 (half the xGMI bytes; the sum runs in bf16, its rounding grows with the number of ranks; "
              "needs --fused_optimizer)")
    flag("--grad_comm_mode", choices=["all_reduce", "rs_ag"], default="rs_ag",
         help="gradient exchange under torch.distributed.run: rs_ag = reduce-scatter of every chunk, the fused Adam steps "
              "this rank's 1 / world share and the UPDATED weights are all-gathered (sharded optimizer step; with another "
              "optimizer: reduce-scatter + all-gather of the gradients); all_reduce = chunked all-reduce, every rank "
              "steps the whole bucket (parallel.py, optim.py)")
    flag("--device_cache", default=False, **onoff,
         help="keep every (x, y) training pair in HBM (deterministic measurements) and draw crops on the device "
              "instead of the per-item DataLoader path (datasets/device_cache.py)")
    return parser


def main(argv=None):
    rank, local_rank, world = parallel.init_from_env()
    seed = 0 + rank                                   # reference: all seeds 0 (single process)
    torch.manual_seed(0)                              # identical initial weights on every rank
    np.random.seed(seed)
    random.seed(seed)

    args = build_parser().parse_args(argv)
    if args.device == "cuda" and world > 1:
        args.device = f"cuda:{local_rank % torch.cuda.device_count()}"     # (% only matters for shared-GPU rehearsals)
        torch.cuda.set_device(args.device)
    if args.fine_tuning or args.fine_tuning_params or args.weights_distance_loss:
        raise NotImplementedError("fine-tuning options are outside the hot path of this build")
    crop.FIX_BATCHED_CROP = args.fix_batched_crop

    physics = get_physics(args, device=args.device)
    model = get_model(args=args, physics=physics, device=args.device)
    model.to(args.device)
    model.train()
    if args.weights is not None:
        model.load_weights(get_weights(args.weights, args.device))
    backbone = model.get_backbone()
    parallel.broadcast_parameters(backbone.flat_params)       # (no-op without a gradient exchange)
    torch.manual_seed(seed)
    torch.cuda.manual_seed(seed)                      # decorrelated b / noise / rates / centres per rank

    loss = get_loss(args=args, physics=physics)
    dataset = get_dataset(args=args, purpose="train", physics=physics, device=args.device,
                          _HOTFIX=(args.task == "sr"))
    sampler = None
    if world > 1:
        sampler = torch.utils.data.distributed.DistributedSampler(dataset, shuffle=True, seed=0)
    dataloader = torch.utils.data.DataLoader(dataset, batch_size=args.batch_size, shuffle=sampler is None,
                                             sampler=sampler)
    device_cache = None
    if args.device_cache:
        if args.method == "css" or not args.SyntheticDataset__deterministic_measurements:
            raise ValueError("--device_cache needs deterministic measurements (and no css re-degradation)")
        from datasets import SyntheticPairs
        from datasets.device_cache import DeviceResidentPairs
        if isinstance(dataset, SyntheticPairs):
            full = SyntheticPairs(physics, args.device, length=len(dataset))
        else:
            full = dataset.dataset.synthetic_dataset
        device_cache = DeviceResidentPairs(full, physics, crop_size=args.PrepareTrainingPairs__crop_size,
                                           crop_location=args.PrepareTrainingPairs__crop_location,
                                           hotfix_sr_crop=(args.task == "sr"), rank=rank, world=world)
        if rank == 0:
            print(f"\nDevice cache: {len(device_cache)} pairs, {device_cache.nbytes() / 2**20:.0f} MiB\n")

    epochs = args.epochs if args.epochs is not None else {"urban100": 4000, "ct": 100}.get(args.dataset, 500)
    lr = args.lr if args.lr is not None else (2e-4 if args.task == "sr" else 1e-4)
    optimizer_kind = args.optimizer if args.optimizer is not None else "Adam"
    if rank == 0:
        print(f"\nSelected learning rate: {lr:e}\n")
        print(f"\nSelected optimizer: {optimizer_kind}\n")

    from models import _ops as model_ops
    model_ops.set_compute_dtype(args.compute_dtype)
    fused_adam = optimizer_kind == "Adam" and args.fused_optimizer
    if args.grad_comm_dtype == "bf16" and not fused_adam:
        raise ValueError("--grad_comm_dtype bf16 needs the fused Adam (it reads the bf16 bucket directly)")
    comm_dtype = torch.bfloat16 if args.grad_comm_dtype == "bf16" else torch.float32
    reducer = parallel.FlatGradientReducer(backbone.flat_grads, comm_dtype=comm_dtype,
                                           mode=args.grad_comm_mode) if parallel.exchange_active() else None
    if optimizer_kind == "Adam" and args.fused_optimizer:
        optimizer = FlatAdam(model, lr=lr, betas=(0.9, args.optimizer_beta2), reducer=reducer)
    elif optimizer_kind == "Adam":
        optimizer = torch.optim.Adam(model.parameters(), lr=lr, betas=(0.9, args.optimizer_beta2))
    elif optimizer_kind == "SGD":
        optimizer = torch.optim.SGD(model.parameters(), lr=lr)
    else:
        raise ValueError(f"Unknown optimizer: {optimizer_kind}")
    scheduler = get_lr_scheduler(optimizer=optimizer, epochs=epochs, lr_scheduler_kind=args.lr_scheduler_kind)

    checkpoint_interval = args.checkpoint_interval
    if checkpoint_interval is None:
        checkpoint_interval = 400 if args.dataset == "urban100" else 50

    log = None
    if rank == 0:
        os.makedirs(args.out_dir, exist_ok=True)
        file = open(f"{args.out_dir}/training.csv", "w", newline="", buffering=1)
        log = csv.writer(file)
        log.writerow(["Epoch", "Training Loss"])

    scheduler_disabled = False
    if args.RESUME is not None:
        ckp = torch.load(args.RESUME, map_location=args.device)
        print("Loading checkpoint from epoch", ckp["epoch"])
        model.load_weights(ckp["params"])
        optimizer.load_state_dict(ckp["optimizer"])
        scheduler.load_state_dict(ckp["scheduler"])
        scheduler_disabled = True                     # as upstream: fixed, explicitly given rate
        assert args.lr is not None
        for group in optimizer.param_groups:
            group["lr"] = args.lr

    checkpoints_dir = f"{args.out_dir}/checkpoints"

    def checkpoint_name(epoch):
        return f"{checkpoints_dir}/ckp_{epoch:0{len(str(epochs))}}.pt"

    def consolidate():
        """Several GPUs with the sharded optimizer step: every rank brings its float32 weights and Adam moments up to
        date before rank 0 writes them (a collective; a no-op otherwise)."""
        if hasattr(optimizer, "consolidate"):
            optimizer.consolidate()

    consolidate()
    if rank == 0:
        save_training_state(epoch=0, model=model, optimizer=optimizer, scheduler=scheduler,
                            state_path=checkpoint_name(0))

    graphed, early_event = None, None
    trace_step_kind = rank == 0 and os.environ.get("SEI_TRACE_STEP_KIND") == "1"
    for epoch in range(epochs):
        if sampler is not None:
            sampler.set_epoch(epoch)
        loss_sum = torch.zeros((), device=args.device)
        steps = 0
        batches = dataloader if device_cache is None else device_cache.batches(args.batch_size)
        for x, y in batches:
            x, y = x.to(args.device), y.to(args.device)
            used_graph = False
            can_graph = (args.hip_graph and isinstance(optimizer, FlatAdam) and graphs.can_capture(loss)
                         and y.shape[0] == args.batch_size)
            if can_graph:
                if graphed is None:                   # capture once; short last batches run eagerly
                    from graphs import GraphedLossStep
                    early = reducer is not None and os.environ.get("SEI_NO_EARLY_RELEASE") != "1"
                    graphed = GraphedLossStep(loss, model, optimizer,
                                              (args.batch_size, y.shape[1], args.Loss__crop_size, args.Loss__crop_size),
                                              early_release=early,
                                              fuse_optimizer=reducer is None and args.fuse_optimizer_step)
                    if early and graphed.early_grads is not None:
                        early_event = graphed.early_grads[0]
                        # (a capture after sharded optimizer steps -- the first batches were short -- changes which
                        # slices a rank owns: complete every rank's masters and moments first; a collective, reached by
                        # all ranks at the same step since their shards are equal-length)
                        consolidate()
                        reducer.set_early_range(graphed.early_grads[1:])
                training_loss = graphed(x, y)
                used_graph = True
            else:
                optimizer.zero_grad()
                training_loss = loss(x=x, y=y, model=model)
                training_loss.backward()
            if trace_step_kind:                       # tests: which launch path the first step took
                print("step kind: hipGraph replay" if used_graph else "step kind: eager")
                trace_step_kind = False
            if reducer is not None:
                reducer.reduce_async(early=early_event if used_graph else None,
                                     direct=used_graph and bool(graphed.direct_views))
                if not isinstance(optimizer, FlatAdam):
                    reducer.wait_all()
                    backbone.flat_grads /= world
            optimizer.step()
            loss_sum += training_loss.detach()
            steps += 1
            if args.max_steps is not None and steps >= args.max_steps:
                break
        if not scheduler_disabled:
            scheduler.step()

        epoch_loss = parallel.all_reduce_mean_scalar(loss_sum / max(steps, 1)).item()
        if (epoch % checkpoint_interval == 0) or (epoch == epochs - 1):
            consolidate()
        if rank == 0:
            stamp = datetime.now().strftime("%Y-%m-%d %H:%M:%S")
            print(f"\t{stamp}\t[{epoch + 1:{len(str(int(epochs)))}d}/{epochs}]\tTraining_Loss: {epoch_loss:.2e}")
            log.writerow([epoch + 1, epoch_loss])
            if (epoch % checkpoint_interval == 0) or (epoch == epochs - 1):
                save_training_state(epoch, model, optimizer, scheduler, checkpoint_name(epoch + 1))

    if rank == 0:
        torch.save(model.get_weights(), f"{args.out_dir}/weights.pt")


if __name__ == "__main__":
    main()

"""Which torch (ATen) kernels does one step still launch, and from which source line? (the captured
step should contain no at::native kernel / copyBuffer.) Runs the captured step's eager twin under torch.profiler with
Python stacks and prints every device kernel that is not one of libsei_hip.so's, grouped by (kernel, innermost package
frame), with its launch count and device time per step."""
import argparse
import collections
import os
import sys

import torch

ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "scale-equivariant-imaging_amd"))
sys.path.insert(1, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--exchange", action="store_true", help="SEI_FORCE_EXCHANGE=1 must be set in the environment too")
    o = ap.parse_args()
    opt = argparse.Namespace(batch=o.batch, hidden=32, scales=5, task="deblurring", sr_factor=4, arch="unet", full256=False,
                             grad_comm="auto", grad_comm_mode="rs_ag", graph=True, fuse_optimizer=True, fuse_min_numel=1 << 24,
                             direct_bf16_grads=True)
    import parallel
    rank, local_rank, world = parallel.init_from_env()
    torch.cuda.set_device(0)
    leg = bench.Leg(opt, "bf16", "cuda:0", 0, world)
    for _ in range(2):
        leg.step()
    torch.cuda.synchronize()
    step = leg.eager_twin_step
    step()
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
        step()
        torch.cuda.synchronize()
    rows = collections.OrderedDict()
    pkg = os.path.join(ROOT, "scale-equivariant-imaging_amd")
    for ev in prof.events():
        if ev.device_type != torch.autograd.DeviceType.CPU or not ev.kernels:
            continue
        for k in ev.kernels:
            name = k.name
            if not ("at::native" in name or "rocclr" in name or "Memcpy" in name or "Memset" in name or "elementwise" in name
                    or "reduce_kernel" in name or "distribution" in name or "philox" in name.lower()):
                continue
            where = next((fr for fr in ev.stack if "scale-equivariant-imaging_amd" in fr or "bench.py" in fr), "?")
            key = (name[:70], ev.name, str(ev.input_shapes)[:60], where.replace(pkg, "")[:90])
            r = rows.setdefault(key, [0, 0.0])
            r[0] += 1
            r[1] += k.duration
    total = 0.0
    for (kname, op, shapes, where), (n, us) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
        total += us
        print(f"{us:8.1f} us  x{n:<3d} {op:<28s} {shapes:<60s} {where}\n             {kname}")
    print(f"torch kernels in one eager twin step: {sum(v[0] for v in rows.values())} launches, {total:.1f} us")


if __name__ == "__main__":
    main()

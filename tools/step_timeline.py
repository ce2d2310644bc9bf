"""The timed step as an ORDERED list of entry-point calls with the device time of each (kernel work aid, not the bench
line): one eager step of configs[1] is recorded exactly as bench.py records it for its roofline legs, then every call is
re-issued on its own between HIP events (one warm run + `reps` timed ones). Prints `index  us  entry  small-int args`
and per-entry totals; the sum is what the captured graph would take with no overlap between nodes.

    python tools/step_timeline.py [--batch 32] [--reps 3] > gpurun_out/timeline.txt
"""
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
    ap.add_argument("--reps", type=int, default=3)
    o = ap.parse_args()
    opt = argparse.Namespace(batch=o.batch, hidden=32, scales=5, task="deblurring", sr_factor=4, arch="unet", full256=False,
                             grad_comm="auto", grad_comm_mode="rs_ag", graph=True, fuse_optimizer=True, fuse_min_numel=1 << 24,
                             direct_bf16_grads=True)
    import _native
    import parallel
    rank, local_rank, world = parallel.init_from_env()
    torch.cuda.set_device(0)
    leg = bench.Leg(opt, "bf16", "cuda:0", 0, world)
    for _ in range(2):
        leg.step()
    torch.cuda.synchronize()
    keep, records, log = leg.record_one_step()
    totals = collections.OrderedDict()
    total = 0.0
    for i, (name, args) in enumerate(log):
        _native.call(name, *args)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(o.reps):
            _native.call(name, *args)
        e1.record()
        e1.synchronize()
        us = 1e3 * e0.elapsed_time(e1) / o.reps
        ints = [a for a in args if isinstance(a, int) and 0 <= a < (1 << 24)]
        print(f"{i:4d} {us:8.1f} us  {name:34s} {ints}")
        t = totals.setdefault(name, [0, 0.0])
        t[0] += 1
        t[1] += us
        total += us
    print(f"\n{len(log)} calls, {total / 1e3:.3f} ms\n")
    for name, (n, us) in sorted(totals.items(), key=lambda kv: -kv[1][1]):
        print(f"{us:9.1f} us  x{n:<3d} {name}")
    del keep


if __name__ == "__main__":
    main()

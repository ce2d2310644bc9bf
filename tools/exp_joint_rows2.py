"""More joint-row candidates (see exp_joint_rows.py): the inter-level convolutions' data gradients."""
import os, sys, statistics, torch
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "scale-equivariant-imaging_amd"))
from exp_ring3 import build, once
for (M, N, K, kind) in [(864, 2048, 8192, "none_kr"), (576, 2048, 8192, "none_kr"), (288, 2048, 8192, "none_kr"),
                        (13824, 2048, 512, "none_kr"), (9216, 2048, 512, "none_kr"), (4608, 2048, 512, "none_kr"),
                        (3456, 8192, 2048, "none_kr"), (2304, 8192, 2048, "none_kr"), (1152, 8192, 2048, "none_kr"),
                        (3456, 512, 2048, "none_kr"), (2304, 512, 2048, "none_kr"), (1152, 512, 2048, "none_kr")]:
    f, _, _ = build(M, N, K, kind)
    codes = [0, 32, 31, 33, 1]
    times = {c: [] for c in codes}
    for rnd in range(5):
        for code in codes:
            f(code); torch.cuda.synchronize()
            times[code].append(once(lambda: f(code)))
    print(f"{M}x{N}x{K} {kind}: " + "  ".join(f"tile {c} {statistics.median(t):.1f}us" for c, t in times.items()), flush=True)

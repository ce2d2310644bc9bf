import os, sys, statistics, torch
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "scale-equivariant-imaging_amd"))
from exp_ring3 import build, once
for (M, N, K, kind) in [(288, 8192, 32768, "res"), (288, 8192, 32768, "none_kr"), (576, 8192, 32768, "res"), (576, 8192, 32768, "none_kr"),
                        (2304, 2048, 8192, "res"), (2304, 2048, 8192, "none_kr"), (1152, 2048, 8192, "res"), (1152, 2048, 8192, "none_kr"),
                        (288, 32768, 8192, "gelu"), (576, 32768, 8192, "gelu")]:
    f, _, _ = build(M, N, K, kind)
    codes = [0, 31, 32, 1]
    times = {c: [] for c in codes}
    for rnd in range(5):
        for code in codes:
            f(code); torch.cuda.synchronize()
            times[code].append(once(lambda: f(code)))
    print(f"{M}x{N}x{K} {kind}: " + "  ".join(f"tile {c} {statistics.median(t):.0f}us" for c, t in times.items()), flush=True)

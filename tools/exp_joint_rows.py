import os, sys, statistics, torch
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "scale-equivariant-imaging_amd"))
from exp_ring3 import build, once
for (M, N, K, kind) in [(3456, 2048, 8192, "none_kr"), (2304, 2048, 8192, "none_kr"), (1152, 2048, 8192, "none_kr"),
                        (3456, 8192, 2048, "dgelu_kr"), (2304, 8192, 2048, "dgelu_kr"), (1152, 8192, 2048, "dgelu_kr"),
                        (864, 8192, 32768, "none_kr"), (576, 8192, 32768, "none_kr"), (288, 8192, 32768, "none_kr"),
                        (864, 32768, 8192, "dgelu_kr"), (576, 32768, 8192, "dgelu_kr"), (288, 32768, 8192, "dgelu_kr"),
                        (13824, 512, 2048, "none_kr"), (9216, 512, 2048, "none_kr"), (4608, 512, 2048, "none_kr"),
                        (13824, 2048, 512, "dgelu_kr"), (9216, 2048, 512, "dgelu_kr"), (4608, 2048, 512, "dgelu_kr")]:
    f, _, _ = build(M, N, K, kind)
    codes = [0, 32, 31]
    times = {c: [] for c in codes}
    for rnd in range(5):
        for code in codes:
            f(code); torch.cuda.synchronize()
            times[code].append(once(lambda: f(code)))
    print(f"{M}x{N}x{K} {kind}: " + "  ".join(f"tile {c} {statistics.median(t):.1f}us" for c, t in times.items()), flush=True)

"""Fixed cost of a kernel node inside a replayed hipGraph: chains of N tiny dependent kernels, time per node."""
import torch, time
x = torch.zeros(64, device="cuda")
for n in (100, 500, 2000):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3): x.add_(1.0)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): x.add_(1.0)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): g.replay()
    e1.record(); torch.cuda.synchronize()
    print(f"{n} nodes: {e0.elapsed_time(e1) / 10 / n * 1e3:.2f} us per node")

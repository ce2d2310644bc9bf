"""How does torch's HIP generator map (seed, offset, element index) to Philox4x32-10 engine calls? Compares torch.randn /
torch.rand with rocRAND's device functions per subsequence (tools/probes/philox_probe.hip, built with contraction fast / off).
    python tools/probe_torch_philox.py"""
import ctypes, os, torch
here = os.path.dirname(os.path.abspath(__file__))
torch.cuda.init()
for variant in ("fast", "off"):
    lib = ctypes.CDLL(os.path.join(here, "probes", f"libphilox_probe_{variant}.so"))
    lib.philox_probe.argtypes = [ctypes.c_ulonglong, ctypes.c_ulonglong, ctypes.c_int] + [ctypes.c_void_p] * 4
    for numel in (32, 64, 6912, 124416, 221184):
        seed = 1234
        torch.cuda.manual_seed(seed)
        gen = torch.cuda.default_generators[0]
        off0 = gen.get_offset()
        t = torch.randn(numel, device="cuda")
        off1 = gen.get_offset()
        torch.cuda.manual_seed(seed)
        u = torch.rand(numel, device="cuda")
        n = numel
        nrm = torch.empty((n, 4), device="cuda"); uni = torch.empty((n, 4), device="cuda")
        raw = torch.empty((n, 4), device="cuda", dtype=torch.int32)
        rc = lib.philox_probe(seed, off0, n, nrm.data_ptr(), uni.data_ptr(), raw.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        eq0 = float((t == nrm[:, 0]).float().mean())
        close0 = float(((t - nrm[:, 0]).abs() <= 1e-6 * nrm[:, 0].abs().clamp(min=1e-3)).float().mean())
        ueq0 = float((u == torch.where(uni[:, 0] == 1, torch.zeros_like(u), uni[:, 0])).float().mean())
        print(f"[{variant}] numel {numel}: offset {off0} -> {off1}; randn == normal4.x of subsequence i: {eq0:.4f} (within 1e-6: {close0:.4f}); "
              f"rand == uniform4.x: {ueq0:.4f}", flush=True)
        if eq0 < 0.5:
            # where does element i come from? search the first elements in the whole table (any component)
            flat = nrm.reshape(-1)
            for i in (0, 1, 2, 255, 256, 1000, numel - 1):
                if i >= numel: continue
                hit = (flat == t[i]).nonzero().flatten().tolist()[:4]
                near = ((flat - t[i]).abs() < 1e-6).nonzero().flatten().tolist()[:4]
                print(f"     element {i}: value {float(t[i]):+.7f}; exact hits (subsequence, component): {[(h // 4, h % 4) for h in hit]}; near: {[(h // 4, h % 4) for h in near]}")

import os, sys, torch
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "scale-equivariant-imaging_amd"))
import _native as N
g = torch.Generator(device="cuda").manual_seed(1)
EPI = dict(NONE=0, BIAS=1, GELU=2, RES=3, DGELU=4, SCALE_RES=7)
cases = [("qkv fwd", 576, 192, 576, "BIAS", 1), ("proj fwd", 192, 192, 180, "RES", 0), ("proj fwd drop", 192, 192, 180, "SCALE_RES", 0),
         ("fc1 fwd", 384, 192, 384, "GELU", 0), ("fc2 fwd", 192, 384, 180, "RES", 0), ("fc2 fwd drop", 192, 384, 180, "SCALE_RES", 0),
         ("fc2 dgrad", 384, 192, 384, "DGELU", 1), ("fc1 dgrad", 192, 384, 192, "NONE", 0), ("proj dgrad", 192, 192, 192, "NONE", 1),
         ("qkv dgrad", 192, 576, 192, "NONE", 0)]
M = 110592
for name, Nn, K, nv, epi, out16 in cases:
    a = torch.randn((M, K), device="cuda", generator=g).bfloat16()
    w = (0.1 * torch.randn((Nn, K), device="cuda", generator=g)).bfloat16(); w[nv:] = 0
    bias = torch.randn(nv, device="cuda", generator=g); res = torch.randn((M, nv), device="cuda", generator=g)
    drop = (torch.rand(M, device="cuda", generator=g) > 0.1).float() / 0.9
    R1 = drop if epi == "SCALE_RES" else (res if epi in ("RES", "DGELU") else None); R2 = res if epi == "SCALE_RES" else None
    hasb = epi in ("BIAS", "GELU", "RES", "SCALE_RES")
    first = None; bad = 0
    for rep in range(12):
        n32 = torch.zeros((M, nv), device="cuda") if (not out16 or epi == "GELU") and epi != "DGELU" else None
        n16 = torch.zeros((M, Nn), device="cuda", dtype=torch.bfloat16) if out16 or epi in ("GELU", "DGELU") else None
        junk = torch.randn((64, 1024, 1024), device="cuda")      # evict caches, vary timing
        N.call("sei_rowgemm_bf16", a.data_ptr(), K, w.data_ptr(), K, N.ptr(n32), nv, N.ptr(n16), Nn, M, Nn, K, nv, EPI[epi],
               bias.data_ptr() if hasb else None, N.ptr(R1), N.ptr(R2), nv)
        torch.cuda.synchronize()
        cur = (n32.clone() if n32 is not None else None, n16.clone() if n16 is not None else None)
        if first is None: first = cur
        else:
            for x, y in zip(first, cur):
                if x is not None and not torch.equal(x, y): bad += 1
    print(f"{name:14s}: {bad} of 11 repeats differ from the first run", flush=True)

"""Print what ds_read_b64_tr_b16 returns (lane -> 4 elements) for a known LDS image."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "scale-equivariant-imaging_amd"))
sys.path.insert(1, os.path.dirname(os.path.abspath(__file__)))
import _tuning; _tuning.use()    # process-wide tile switches live in the tools-only build
import _native as N
img = (torch.arange(64).view(64, 1) * 128 + torch.arange(128).view(1, 128)).to(torch.int16).cuda()
out = torch.zeros(64 * 4, dtype=torch.int16, device="cuda")
N.call("sei_debug_tr_probe", img.data_ptr(), out.data_ptr(), 8, 32)
torch.cuda.synchronize()
o = out.cpu().view(64, 4).tolist()
for lane in (0, 1, 2, 3, 4, 15, 16, 17, 31, 32, 47, 48, 63):
    print(lane, [(v // 128, v % 128) for v in o[lane]])
ok = all(o[l][e] == (8 + 4 * (l >> 4) + e) * 128 + 32 + (l & 15) for l in range(64) for e in range(4))
print("matches 'lane i of group g gets column c0+i of rows r0+4g .. r0+4g+3':", ok)

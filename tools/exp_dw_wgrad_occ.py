"""dwconv7_tiled_kernel<WEIGHT_GRAD> compiled for 3 workgroups per CU (168 VGPRs, 36 B of scratch per lane) against 2 (no
spills): the weight-gradient launches of the three fine levels at 3B = 96 crops. Run once per library:
    python tools/exp_dw_wgrad_occ.py ; SEI_HIP_LIBRARY=.../libsei_hip_occ2.so python tools/exp_dw_wgrad_occ.py"""
import os, sys, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "scale-equivariant-imaging_amd"))
import _native as N
print("library:", N.LIB_PATH)
for (B, H, C) in ((96, 48, 32), (96, 24, 128), (96, 12, 512), (64, 48, 32), (32, 48, 32)):
    x = torch.randn(B, H, H, C, device="cuda"); gy = torch.randn(B, H, H, C, device="cuda")
    need = N.lib().sei_dwconv7_bwd_weight_workspace_ex(B, H, H, C, 0)
    work = torch.empty(need, device="cuda")
    fn = lambda: N.call("sei_dwconv7_bwd_weight_ex", x.data_ptr(), gy.data_ptr(), None, None, B, H, H, C, work.data_ptr(), need, 0)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"  {B} x {H} x {H} x {C}: {e0.elapsed_time(e1) / 20 * 1e3:6.1f} us   checksum {float(work.double().sum()):.6e}")

"""Time the MFMA window-attention kernels at the SwinIR bench shape (2B = 64 images of 48x48 tokens, 6 heads)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "scale-equivariant-imaging_amd"))
import _native as N
def timeit(fn, iters=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
B, H, W, heads = 64, 48, 48, 6
M = B * H * W
qkv = (torch.randn((M, 3 * heads * 32), device="cuda") * 0.5).bfloat16()
go = torch.randn((M, heads * 32), device="cuda").bfloat16()
table = torch.randn((225, heads), device="cuda") * 0.1
out = torch.empty((M, heads * 32), device="cuda", dtype=torch.bfloat16)
dqkv = torch.empty_like(qkv); dtable = torch.zeros_like(table)
for shift in (0, 4):
    tf = timeit(lambda: N.call("sei_swin_attn_fwd_bf16", qkv.data_ptr(), table.data_ptr(), out.data_ptr(), B, H, W, heads, shift, 30 ** -0.5))
    tb = timeit(lambda: N.call("sei_swin_attn_bwd_bf16", qkv.data_ptr(), table.data_ptr(), go.data_ptr(), dqkv.data_ptr(), dtable.data_ptr(), B, H, W, heads, shift, 30 ** -0.5))
    items = B * (H // 8) * (W // 8) * heads
    print(f"shift {shift}: fwd {tf:.1f} us ({items * 0.52 / tf:.0f} GFLOP/s-ish), bwd {tb:.1f} us; {items} (window, head) items")
